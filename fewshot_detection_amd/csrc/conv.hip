// Convolution (3x3 / 1x1, stride 1, "same" padding) as an im2col-free implicit GEMM on the gfx950 matrix cores, fp32 in and
// out.  Two arithmetics (fsd_f32_gemm_mode): "split" (default) -- fp32 operands split in the staging registers into three
// bfloat16 planes, six v_mfma_f32_32x32x16_bf16 terms per product, fp32 accumulate (SPLIT below) -- and the native fp32
// matrix instruction (v_mfma_f32_32x32x2_f32, 157 TFLOP/s chip peak), which the rest of this header describes.
//
//   M = B*H*W output pixels, N = Cout, K = taps * Cin.   Activations are NHWC, weights are
//   pre-packed [Cout][tap][Cin] ("K-major" on both sides), so every 16-byte global load is a
//   run of 4 input channels of one filter tap and both operands land in LDS as [row][k] tiles.
//   LDS rows are padded 32 -> 36 floats: the ds_read_b128 fragment reads (lane = row, 16 B at a
//   144-B stride) then touch 16 distinct 16-B slots per lane group, i.e. conflict-free.
//   A lane feeds 4 consecutive k of its row to 4 MFMAs (lanes 0-31: k..k+3, lanes 32-63:
//   k+4..k+7), so one b128 read per operand tile supplies four matrix instructions.
//
// The same kernel produces: forward convs (+bias, + per-tile BatchNorm partial sums in the
// epilogue), data gradients (weights packed flipped/transposed), and the fused
// reweighting (x) 1x1 detection head with an NCHW store (operands swapped in the MFMA so the
// accumulator comes out transposed and the store stays coalesced).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include "fsdet.h"
#include "conv_common.hpp"
#include "profile.hpp"

namespace {

using namespace fsd_conv;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kBK = 32;      // k-chunk (floats) staged per LDS buffer
constexpr int kLd = 36;      // padded LDS row stride (floats)

__device__ float g_zero_page[64];   // zero-initialised, never written: source of padding for the DMA path

// SPLIT: the same GEMM on the bf16 matrix cores at fp32 accuracy.  Every fp32 operand element is split, once per
// workgroup on its way from the staging registers into LDS, into three bfloat16 planes x = x1 + x2 + x3 (x1 = bf16(x),
// x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 3 x 8 significant bits = the 24 of fp32, the residuals are exact), and a
// product a*b is accumulated (fp32, v_mfma_f32_32x32x16_bf16) from the six cross terms down to 2^-16 relative:
// a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1.  The three dropped terms are <= 2^-24 |ab| each -- below the rounding of one
// fp32 product.  Measured against a double-precision sum (tools/probes/split_gemm_probe.hip, profiles/r03_split_probe.txt):
// relative L2 error 0.98e-6 at K = 4608 against 1.20e-6 for v_mfma_f32_32x32x2_f32, identical to all nine terms.
// Six bf16 MFMAs cost 6/16 of the fp32 MFMAs they replace (2.48 PFLOP/s against 138 TFLOP/s measured issue rate).
// (Non-finite inputs: an infinite operand element gives NaN here -- inf - bf16(inf) -- where the fp32 instruction gives
// +-inf or NaN depending on its partner; finite fp32 values above bfloat16's largest, 3.3895e38, round to inf the same way.)
constexpr int kLdH = 40;     // SPLIT: padded LDS row stride of one plane (bf16 elements; 80 bytes: ds_read_b128 conflict-free)

template <int BM, int BN, int WAVES_M, int WAVES_N, bool NCHW_OUT, int STAGES, bool FASTK, bool GLDS, bool SPLIT = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void conv_gemm_kernel(ConvArgs p) {
  constexpr int NT = WAVES_M * WAVES_N * 64;      // threads per workgroup (4 or 8 waves)
  constexpr int RPP = NT / 8;                     // tile rows staged per pass (8 threads x 16 B per row)
  static_assert(NT == 256 || NT == 512, "4 or 8 waves");
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
  constexpr int TM = BM / WAVES_M / 32;
  constexpr int TN = BN / WAVES_N / 32;
  constexpr int A_PER_T = BM / RPP;   // float4 loads per thread per chunk
  constexpr int B_PER_T = BN / RPP;
  static_assert(!GLDS || (FASTK && STAGES >= 2), "direct-to-LDS staging needs the per-tap fast path and >= 2 LDS stages");
  static_assert(!(SPLIT && GLDS), "the split happens in the staging registers");
  constexpr int LD = GLDS ? kBK : kLd;          // GLDS: linear 128-B rows (XOR-swizzled), else padded rows
  constexpr int PLANE = (BM + BN) * kLdH;       // SPLIT: bf16 elements of one plane of one stage
  constexpr int STAGE = SPLIT ? 3 * PLANE / 2 : (BM + BN) * LD;     // floats
  extern __shared__ __attribute__((aligned(16))) float smem[];

  // XCD-aware order.  Per batch (default): XCD c takes the c-th eighth of every batch's tiles.  Flat (p.flat_xcd, the split
  // arithmetic's batched launches): XCD c takes the c-th eighth of the flat (batch, tile) space and walks it batch by batch,
  // so the tiles of one position share ONE L2 -- with the per-batch order every XCD re-fetches each position's whole U panel
  // (the n-tile index runs fastest).  On the native fp32 MFMA kernel the two orders measured the same (+-2 %, round 2); with a
  // third of the matrix time per tile the operand traffic shows.
  int batch = blockIdx.y, L;
  if (p.flat_xcd && gridDim.y > 1) {
    const int Lf = xcd_swizzle((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
    batch = Lf / (int)gridDim.x;
    L = Lf - batch * (int)gridDim.x;
  } else {
    L = xcd_swizzle(blockIdx.x, gridDim.x);
  }
  int kc0 = 0, kc1 = p.nk;      // this workgroup's k-chunks (split-K: one slice)
  if (gridDim.y > 1) {      // batched GEMMs: every batch has its own operand / result matrices
    if (p.ksplit > 1) {
      const int slice = batch % p.ksplit;
      batch /= p.ksplit;
      const int per = p.nk / p.ksplit;
      kc0 = slice * per;
      kc1 = kc0 + per;
      p.y += (long long)slice * p.y_ks;
    }
    p.x += (long long)batch * p.x_bs;
    p.w = static_cast<const float*>(p.w) + (long long)batch * p.w_bs;
    p.y += (long long)batch * p.y_bs;
  }
  const int mt = L / p.n_tiles, nt = L - mt * p.n_tiles;
  const int m0 = p.m_base + mt * BM, n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
  const int kq = tid & 7, g8 = tid >> 3;
  // SPLIT: a 32-lane half of a wave (what one ds_write_b64 cycle serves) stages rows {b, b+4, b+8, b+12} instead of four
  // consecutive ones: with the 80-byte plane rows their 64-byte pieces tile the 256-byte bank row exactly (offsets 0, 64, 128,
  // 192), where consecutive rows (0, 80, 160, 240) wrap onto each other -- PMC before: SQ_LDS_BANK_CONFLICT / IDX_ACTIVE 0.32.
  const int r0 = SPLIT ? ((g8 & ~15) | ((g8 >> 2) & 3) | ((g8 & 3) << 2)) : g8;
  // GLDS: the DMA writes LDS linearly (wave base + lane*16 B); bank conflicts are avoided by
  // permuting which 16-B k-group each lane FETCHES (same 128-B line) and un-permuting on the read.
  // The XOR mask is (row >> 1) & 7: a 16-lane service group of the ds_read_b128 fragment reads covers rows
  // {0-3, 12-15, 20-27} (+ three more such sets), i.e. all 16 values of row & 15, and with 128-byte rows the 256-byte
  // bank row is (row & 1, 16-byte group) -- so the group must vary with bits 1..3 of the row.  (Rounds 1-2 used
  // row & 7, which maps every pair of rows 8 apart onto one slot: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50.)
  const int kq_src = GLDS ? (kq ^ ((r0 >> 1) & 7)) : kq;

  // Per-thread im2col bookkeeping.  NHWC input and output share the pixel grid (stride 1, "same"
  // padding), so the input pixel of output pixel `pix` under tap (dy, dx) is pix + dy*W + dx.
  int a_y[A_PER_T], a_x[A_PER_T];
  unsigned a_pix[A_PER_T];
#pragma unroll
  for (int j = 0; j < A_PER_T; ++j) {
    const int pix = m0 + r0 + RPP * j;
    const int b = pix / p.HW;
    const int rem = pix - b * p.HW;
    const int yy = rem / p.W;
    a_y[j] = pix < p.M ? yy : -(1 << 20);       // out-of-range rows fail the bounds test below
    a_x[j] = rem - yy * p.W;
    a_pix[j] = (unsigned)pix;
  }
  const float* wrow = static_cast<const float*>(p.w) + (long long)(n0 + r0) * p.Kpad + kq_src * 4;
  const unsigned x_ld = (unsigned)p.x_ld;

  f32x4 ra[A_PER_T], rb[B_PER_T];
  unsigned a_mask = 0;     // bit j: ra[j] holds real data (else the tile row is zero padding)

  // FASTK (Cin % 32 == 0): a k-chunk is 32 channels of ONE filter tap, so the bounds tests and
  // row offsets are recomputed only when the tap changes (every Cin/32 chunks); per chunk the
  // address work is one add per load.  Offsets are 32-bit element indices (launcher checks range).
  unsigned a_off[A_PER_T];
  unsigned tap_mask = 0;
  int f_tap = 0, f_cc = kc0;         // (split-K launches are 1x1: one tap, the chunk index is the channel chunk)
  auto retap = [&]() {
    const int ky = f_tap / p.ks, kx = f_tap - ky * p.ks;
    const int dy = ky - p.pad, dx = kx - p.pad;
    const int shift = dy * p.W + dx;
    tap_mask = 0;
#pragma unroll
    for (int j = 0; j < A_PER_T; ++j) {
      const bool ok = (unsigned)(a_y[j] + dy) < (unsigned)p.H && (unsigned)(a_x[j] + dx) < (unsigned)p.W;
      a_off[j] = ok ? (a_pix[j] + (unsigned)shift) * x_ld + (unsigned)(kq_src * 4) : (unsigned)(kq_src * 4);
      tap_mask |= ok ? (1u << j) : 0u;
    }
  };
  if constexpr (FASTK) retap();

  auto gload_lds = [&](int kc, float* st) {
    // 16-byte global->LDS DMA per lane: no staging registers, no ds_write, no select.  Padding
    // (image border taps, rows past M) is fetched from a zero page.
    const unsigned coff = (unsigned)f_cc * kBK;
#pragma unroll
    for (int j = 0; j < A_PER_T; ++j) {
      const float* src = (tap_mask >> j) & 1u ? p.x + (a_off[j] + coff) : g_zero_page + kq * 4;
      __builtin_amdgcn_global_load_lds(src, st + (j * RPP + wave * 8) * kBK, 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_PER_T; ++j)
      __builtin_amdgcn_global_load_lds(wrow + (long long)j * RPP * p.Kpad + kc * kBK,
                                       st + (BM + j * RPP + wave * 8) * kBK, 16, 0, 0);
    if (++f_cc == p.cpt) {
      f_cc = 0;
      ++f_tap;
      retap();
    }
  };
  unsigned a_ch = 0;       // first input channel of this thread's staged float4 (activation on load)
  auto gload = [&](int kc) {
    if constexpr (FASTK) {
      const unsigned coff = (unsigned)f_cc * kBK;
      a_ch = coff + (unsigned)kq * 4u;
#pragma unroll
      for (int j = 0; j < A_PER_T; ++j) ra[j] = *reinterpret_cast<const f32x4*>(p.x + (a_off[j] + coff));
      a_mask = tap_mask;
#pragma unroll
      for (int j = 0; j < B_PER_T; ++j)
        rb[j] = *reinterpret_cast<const f32x4*>(wrow + (long long)j * RPP * p.Kpad + kc * kBK);
      if (++f_cc == p.cpt) {      // uniform branch: next chunk starts a new tap
        f_cc = 0;
        ++f_tap;
        retap();
      }
    } else {
      const int kg = kc * 8 + kq;
      const int tap = kg / p.cpg;
      const int c4 = kg - tap * p.cpg;
      const int ky = tap / p.ks, kx = tap - ky * p.ks;
      const int dy = ky - p.pad, dx = kx - p.pad;
      const bool kvalid = kg < p.kgroups;
      a_mask = 0;
#pragma unroll
      for (int j = 0; j < A_PER_T; ++j) {
        const bool ok = kvalid && (unsigned)(a_y[j] + dy) < (unsigned)p.H && (unsigned)(a_x[j] + dx) < (unsigned)p.W;
        // Always load (from a clamped, valid address); the zero-select happens in sstore(), i.e.
        // AFTER the MFMA block, so the loads stay in flight behind the matrix work.
        const unsigned off = ok ? (a_pix[j] + (unsigned)(dy * p.W + dx)) * x_ld + (unsigned)(c4 * 4) : 0u;
        ra[j] = *reinterpret_cast<const f32x4*>(p.x + off);
        a_mask |= ok ? (1u << j) : 0u;
      }
#pragma unroll
      for (int j = 0; j < B_PER_T; ++j)
        rb[j] = *reinterpret_cast<const f32x4*>(wrow + (long long)j * RPP * p.Kpad + kc * kBK);
    }
  };
  // activation on load (p.in_scale): every staged float4 of a chunk holds the same 4 input channels (a_ch); padding stays 0
  f32x4 in_sc = {1.f, 1.f, 1.f, 1.f}, in_sh = {0.f, 0.f, 0.f, 0.f};
  auto act_in = [&](const f32x4& v) -> f32x4 { return p.in_scale ? affine_act4(v, in_sc, in_sh, p.in_slope) : v; };
  auto sstore = [&](float* st) {
    if constexpr (FASTK) {
      if (p.in_scale) {
        in_sc = *reinterpret_cast<const f32x4*>(p.in_scale + a_ch);
        in_sh = *reinterpret_cast<const f32x4*>(p.in_shift + a_ch);
      }
    }
    if constexpr (SPLIT) {
      unsigned short* sp = reinterpret_cast<unsigned short*>(st);
#pragma unroll
      for (int j = 0; j < A_PER_T + B_PER_T; ++j) {
        const bool is_a = j < A_PER_T;
        const f32x4 v = is_a ? ((a_mask >> j) & 1u ? act_in(ra[is_a ? j : 0]) : f32x4{0.f, 0.f, 0.f, 0.f}) : rb[is_a ? 0 : j - A_PER_T];
        const int row = is_a ? r0 + RPP * j : BM + r0 + RPP * (j - A_PER_T);
        uint2 h, m, l;
        split3(v, h, m, l);
        *reinterpret_cast<uint2*>(sp + row * kLdH + kq * 4) = h;
        *reinterpret_cast<uint2*>(sp + PLANE + row * kLdH + kq * 4) = m;
        *reinterpret_cast<uint2*>(sp + 2 * PLANE + row * kLdH + kq * 4) = l;
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < A_PER_T; ++j) {
      const f32x4 v = (a_mask >> j) & 1u ? act_in(ra[j]) : f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(st + (r0 + RPP * j) * kLd + kq * 4) = v;
    }
#pragma unroll
    for (int j = 0; j < B_PER_T; ++j)
      *reinterpret_cast<f32x4*>(st + (BM + r0 + RPP * j) * kLd + kq * 4) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_off = GLDS ? (lane & 31) * LD : (lane & 31) * LD + (lane >> 5) * 4;
  // GLDS read-side un-swizzle: logical k-group g of row r sits at physical group g ^ ((r >> 1) & 7)
  auto kofs = [&](int k8) { return GLDS ? (((k8 * 2 + (lane >> 5)) ^ ((lane >> 1) & 7)) * 4) : k8 * 8; };
  auto compute = [&](const float* st) {
    if constexpr (SPLIT) {
      // lane = tile row (lane & 31), 8 consecutive k of the 16 of one MFMA (lane >> 5): one ds_read_b128 per plane and tile
      const unsigned short* sp = reinterpret_cast<const unsigned short*>(st);
      const unsigned short* sa = sp + (wm * TM * 32 + (lane & 31)) * kLdH + (lane >> 5) * 8;
      const unsigned short* sb = sp + (BM + wn * TN * 32 + (lane & 31)) * kLdH + (lane >> 5) * 8;
      bf16x8 af[2][3][TM], bf[2][3][TN];
      auto frags = [&](int k16, int buf) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
          for (int i = 0; i < TM; ++i) af[buf][q][i] = *reinterpret_cast<const bf16x8*>(sa + q * PLANE + i * 32 * kLdH + k16 * 16);
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[buf][q][j] = *reinterpret_cast<const bf16x8*>(sb + q * PLANE + j * 32 * kLdH + k16 * 16);
        }
      };
      frags(0, 0);
#pragma unroll
      for (int k16 = 0; k16 < kBK / 16; ++k16) {
        if (k16 + 1 < kBK / 16) frags(k16 + 1, (k16 + 1) & 1);
        // the six terms, smallest first; each term sweeps the TM x TN independent accumulators
        constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = NCHW_OUT ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[k16 & 1][TB[t]][j], af[k16 & 1][TA[t]][i], acc[i][j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[k16 & 1][TA[t]][i], bf[k16 & 1][TB[t]][j], acc[i][j], 0, 0, 0);
      }
      return;
    }
    const float* sa = st + (wm * TM * 32) * LD + frag_off;
    const float* sb = st + (BM + wn * TN * 32) * LD + frag_off;
    {
      // software-pipelined fragments: the reads of k-step k8+1 are in flight under the MFMAs of k8
      f32x4 af[2][TM], bf[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(sa + i * 32 * LD + kofs(0));
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const f32x4*>(sb + j * 32 * LD + kofs(0));
#pragma unroll
      for (int k8 = 0; k8 < kBK / 8; ++k8) {
        if (k8 + 1 < kBK / 8) {
#pragma unroll
          for (int i = 0; i < TM; ++i) af[(k8 + 1) & 1][i] = *reinterpret_cast<const f32x4*>(sa + i * 32 * LD + kofs(k8 + 1));
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[(k8 + 1) & 1][j] = *reinterpret_cast<const f32x4*>(sb + j * 32 * LD + kofs(k8 + 1));
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = NCHW_OUT
                              ? __builtin_amdgcn_mfma_f32_32x32x2f32(bf[k8 & 1][j][kk], af[k8 & 1][i][kk], acc[i][j], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_32x32x2f32(af[k8 & 1][i][kk], bf[k8 & 1][j][kk], acc[i][j], 0, 0, 0);
      }
    }
  };

  // ---- main loop: global -> registers (prefetch) -> LDS.  STAGES = 2: two LDS buffers, one
  // barrier per k-chunk.  STAGES = 1: one buffer, two barriers, half the LDS (more blocks per CU).
  if constexpr (GLDS && STAGES == 2) {
    gload_lds(0, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int kc = 0; kc < p.nk; ++kc) {
      if (kc + 1 < p.nk) gload_lds(kc + 1, smem + (cur ^ 1) * STAGE);   // buffer last read before the previous barrier
      compute(smem + cur * STAGE);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // DMA of the next tile has landed
      __syncthreads();
      cur ^= 1;
    }
  } else if constexpr (GLDS) {
    // Ring of STAGES LDS buffers, DMA running STAGES-1 chunks ahead of the MFMAs: an L2 miss
    // (MALL/HBM, ~2 us under load) in chunk k+2 no longer stalls chunk k+1.  Each wave waits, with
    // a COUNTED vmcnt, only for its own DMA of the NEXT chunk; the raw barrier then publishes
    // every wave's part.  (A plain __syncthreads() would drain the whole queue, vmcnt(0).)
    constexpr int PER = A_PER_T + B_PER_T;          // DMA instructions per wave per chunk
    static_assert(PER == 4 || PER == 6 || PER == 8 || PER == 12, "counted waits below are literal");
    gload_lds(0, smem);
    if (p.nk > 1) gload_lds(1, smem + STAGE);
    if (p.nk > 1) {
      if constexpr (PER == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if constexpr (PER == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if constexpr (PER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    int cur = 0, nxt2 = 2;                          // buffer of chunk kc, buffer for chunk kc+2
    for (int kc = 0; kc < p.nk; ++kc) {
      const bool ahead = kc + 2 < p.nk;
      if (ahead) gload_lds(kc + 2, smem + nxt2 * STAGE);   // last read in iteration kc-1, before its barrier
      compute(smem + cur * STAGE);
      if (ahead) {
        if constexpr (PER == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if constexpr (PER == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (PER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur = cur + 1 == STAGES ? 0 : cur + 1;
      nxt2 = nxt2 + 1 == STAGES ? 0 : nxt2 + 1;
    }
    __syncthreads();
  } else {
  gload(kc0);
  sstore(smem);
  __syncthreads();
  int cur = 0;
  for (int kc = kc0; kc < kc1; ++kc) {
    const bool more = kc + 1 < kc1;
    if (more) gload(kc + 1);
    __builtin_amdgcn_sched_barrier(0);   // the zero-select/LDS store of the prefetched registers stays below the MFMAs
    compute(smem + cur * STAGE);
    if constexpr (STAGES == 2) {
      if (more) sstore(smem + (cur ^ 1) * STAGE);
      __syncthreads();
      cur ^= 1;
    } else {
      __syncthreads();
      if (more) sstore(smem);
      __syncthreads();
    }
  }
  }

  conv_epilogue<BM, BN, WAVES_M, WAVES_N, TM, TN, NCHW_OUT>(p, acc, smem, m0, n0, mt, tid, lane, wm, wn);
}

// ---- 8-wave split GEMM for the batched Winograd position GEMMs (round 4) ---------------------------------------------------
// The 128x128 / 4-wave split kernel above issues 11.4 instructions per MFMA in its main loop (547 per 32-wide k-chunk for 48
// MFMAs: every workgroup re-splits 64 operand values per thread and chunk), far above the ~5 that fit beside a 32-cycle
// v_mfma_f32_32x32x16_bf16 -- PMC: MFMA-busy 0.42.  This kernel cuts the staging work per MFMA instead of re-arranging it:
//   * 256 x 128 tile on 8 waves (4 x 2, wave tile 64 x 64 as before): a thread splits 16 A + 8 B values per chunk for the
//     same 48 MFMAs per wave (2.4x less VALU, LDS-store and global-load work per MFMA);
//   * plane rows are UNPADDED (64 bytes = 32 bf16) with the 16-byte pieces XOR-permuted inside a row (piece ^ ((row >> 2) & 3):
//     the ds_read_b128 fragment reads of a 16-lane service group then cover all 16 slots of the 256-byte bank row, and the
//     8-byte plane stores of 16 consecutive lanes tile 128 contiguous bytes), so TWO stages of all three planes fit: 2 x 3 x
//     (256 + 128) x 64 B = 144 KB of the 160 KB -- one workgroup per CU, two waves per SIMD;
//   * with two stages the split + LDS stores of chunk k+1 sit BETWEEN the two MFMA groups of chunk k (one barrier per chunk).
// Plain GEMM rows only (the batched launches of conv_gemm_batched: 1x1 taps, Cin % 32 == 0, no activation on load); rows past
// M are clamped on load and never stored.
// ACT: activation on load (p.in_scale, see ConvArgs): x = leaky(y * scale[k] + shift[k]) is formed in the staging registers, one
// value per micro-step, a chunk ahead of its split.  ACT = 1: 0 <= in_slope <= 1, leaky(t) = max(t, t * slope) (the bits of
// affine_act4's select, one instruction less); ACT = 2: any slope, the select itself.
// DBG (timing experiments only, WRONG results; FSD_SPLIT8_DBG): 1 = no plane stores, 2 = no barrier in the loop, 3 = no global
// loads in the loop, 4 = no split arithmetic (raw halves stored), 5 = no fragment reads of k-step 1.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool ILV = true, int ACT = 0, bool PERSIST = false, int DBG = 0>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, 2) void conv_gemm_split8_kernel(ConvArgs p) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  static_assert(NT == 512, "8 waves");
  constexpr int RPP = NT / 8;                       // tile rows staged per pass (8 threads x 16 B of fp32 per row)
  constexpr int A_PER_T = BM / RPP, B_PER_T = BN / RPP;
  constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
  static_assert(BM % RPP == 0 && BN % RPP == 0 && TM * 32 * WAVES_M == BM && TN * 32 * WAVES_N == BN, "tile shape");
  constexpr int ROWB = 64;                          // bytes of one plane row (32 bf16)
  constexpr int PLANE_A = BM * ROWB, PLANE_B = BN * ROWB;
  constexpr int STAGE = 3 * (PLANE_A + PLANE_B);    // bytes
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* sm = reinterpret_cast<unsigned char*>(smem);

  // PERSIST (tuning aid, FSD_SPLIT8_PERSIST=1; measured +-0 and 25 more registers, which is what decides whether an HBM-bound
  // kernel of another stream fits beside this one on a SIMD -- default off): the launch has at most one workgroup per CU (the
  // kernel needs 144 KB of LDS); workgroup b walks the virtual
  // block ids b, b + G, b + 2G, ... of the flat (batch, tile) space (G % 8 == 0 keeps a workgroup on the XCD-contiguous run
  // xcd_swizzle gives its XCD).  The first chunk of the NEXT tile is fetched before the epilogue of the current one, whose
  // global stores then drain under the next tile's main loop -- a one-shot workgroup pays the first-load latency and the
  // store tail of every tile with the matrix pipe idle (one workgroup per CU: nobody else is there to fill it).
  const int tiles_pb = p.m_tiles * p.n_tiles, total = tiles_pb * p.batches;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
  const int kq = tid & 7, g8 = tid >> 3;
  const float* asrc[A_PER_T];
  const float* bsrc;
  int m0 = 0, n0 = 0, mt = 0;
  float* ybase = p.y;
  auto set_tile = [&](int v) {
    const int Lf = xcd_swizzle(v, total);
    const int batch = Lf / tiles_pb, L = Lf - batch * tiles_pb;
    mt = L / p.n_tiles;
    const int nt = L - mt * p.n_tiles;
    m0 = p.m_base + mt * BM;
    n0 = nt * BN;
    const float* xb = p.x + (long long)batch * p.x_bs;
#pragma unroll
    for (int j = 0; j < A_PER_T; ++j) {
      int row = m0 + g8 + RPP * j;
      row = row < p.M ? row : p.M - 1;
      asrc[j] = xb + (unsigned)row * (unsigned)p.x_ld + kq * 4;
    }
    bsrc = static_cast<const float*>(p.w) + (long long)batch * p.w_bs + (long long)(n0 + g8) * p.Kpad + kq * 4;
    ybase = p.y + (long long)batch * p.y_bs;
  };
  int v = blockIdx.x;
  set_tile(v);
  const long long b_step = (long long)RPP * p.Kpad;
  // byte offset of this thread's 8-byte piece inside a plane: row g8 (+ RPP * j: multiples of 64 rows leave (row >> 2) & 3 alone)
  const int st_off = g8 * ROWB + ((((kq >> 1) ^ ((g8 >> 2) & 3))) << 4) + (kq & 1) * 8;

  f32x4 ra[A_PER_T], rb[B_PER_T];
  auto gload = [&](int kc) {
#pragma unroll
    for (int j = 0; j < A_PER_T; ++j) ra[j] = *reinterpret_cast<const f32x4*>(asrc[j] + kc * kBK);
#pragma unroll
    for (int j = 0; j < B_PER_T; ++j) rb[j] = *reinterpret_cast<const f32x4*>(bsrc + j * b_step + kc * kBK);
  };
  auto sstore = [&](unsigned char* st) {
#pragma unroll
    for (int j = 0; j < A_PER_T; ++j) {
      uint2 h, m, l;
      split3(ra[j], h, m, l);
      unsigned char* d = st + st_off + j * RPP * ROWB;
      *reinterpret_cast<uint2*>(d) = h;
      *reinterpret_cast<uint2*>(d + PLANE_A) = m;
      *reinterpret_cast<uint2*>(d + 2 * PLANE_A) = l;
    }
#pragma unroll
    for (int j = 0; j < B_PER_T; ++j) {
      uint2 h, m, l;
      split3(rb[j], h, m, l);
      unsigned char* d = st + 3 * PLANE_A + st_off + j * RPP * ROWB;
      *reinterpret_cast<uint2*>(d) = h;
      *reinterpret_cast<uint2*>(d + PLANE_B) = m;
      *reinterpret_cast<uint2*>(d + 2 * PLANE_B) = l;
    }
  };

  f32x16 acc[TM][TN];

  // fragment reads: lane = tile row (lane & 31), k-half (lane >> 5) of the 16 k of one MFMA; logical 16-byte piece
  // 2 * step + half sits at piece ^ ((row >> 2) & 3)
  const int frow = lane & 31, fsw = (frow >> 2) & 3, fh = lane >> 5;
  const int fa_off = (wm * TM * 32 + frow) * ROWB, fb_off = 3 * PLANE_A + (wn * TN * 32 + frow) * ROWB;
  bf16x8 af[2][3][TM], bf[2][3][TN];
  auto frags = [&](const unsigned char* st, int s) {
    const int po = ((2 * s + fh) ^ fsw) << 4;
    // in the order the six terms first use them (A plane 0 with B plane 2, A 2 with B 0, A 1 with B 1): the first MFMAs of a
    // chunk wait for 4 reads, not 12
    constexpr int QA[3] = {0, 2, 1}, QB[3] = {2, 0, 1};
#pragma unroll
    for (int o = 0; o < 3; ++o) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[s][QA[o]][i] = *reinterpret_cast<const bf16x8*>(st + fa_off + QA[o] * PLANE_A + i * 32 * ROWB + po);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[s][QB[o]][j] = *reinterpret_cast<const bf16x8*>(st + fb_off + QB[o] * PLANE_B + j * 32 * ROWB + po);
    }
  };

  {
    static_assert(ILV, "one schedule");
    // Software pipeline with a hand-dealt issue order.  At the top of iteration kc the raw fp32 values of chunk kc+1 are
    // already in registers (loaded during iteration kc-1).  Their split + LDS stores are cut into 48 micro-steps of 2-4 VALU
    // (+ one DS) instructions, ONE after each of the 48 MFMAs of chunk kc, pinned with sched_barrier: the wave keeps issuing
    // while its MFMA runs (32 cycles; 64 with the SIMD's second wave contending), instead of running ~150 VALU instructions
    // as one block during which neither wave of the SIMD has matrix work queued (both are in the same phase of the same
    // barrier interval).  The loads of chunk kc+2 go out as soon as a staging register set is free.  Chunk indices past the
    // end are clamped (redundant loads / stores keep the body branch-free).
    constexpr int NMF = 12 * TM * TN;                      // MFMAs per chunk and wave
    constexpr int NFR = 3 * (TM + TN);                     // fragment reads of one k-step
    constexpr int NST = 8 * (A_PER_T + B_PER_T);           // split micro-steps per chunk and thread
    static_assert(NFR <= NMF / 2 && NST <= 2 * NMF, "the k-step-1 fragments fit behind the k-step-0 MFMAs, <= 2 micro-steps per MFMA");
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    auto cvt2 = [](float a, float b) -> unsigned {                     // one v_cvt_pk_bf16_f32
      const f32x2 v = {a, b};
      return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    };
    auto lo_f = [](unsigned pk) -> float { return __builtin_bit_cast(float, pk << 16); };
    auto hi_f = [](unsigned pk) -> float { return __builtin_bit_cast(float, pk & 0xffff0000u); };
    const int last = p.nk - 1;
    // ACT: scale / shift of this thread's four input channels (kq * 4 ...) of one chunk
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    const float slope = p.in_slope;
    auto load_sc = [&](int kc) {
      sc = *reinterpret_cast<const f32x4*>(p.in_scale + kc * kBK + kq * 4);
      sh = *reinterpret_cast<const f32x4*>(p.in_shift + kc * kBK + kq * 4);
    };
    auto act1 = [&](float v, int e) -> float {
      const float t = __builtin_fmaf(v, sc[e], sh[e]);
      if constexpr (ACT == 2) return t > 0.f ? t : t * slope;
      return __builtin_fmaxf(t, t * slope);
    };
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
    // static priority for the second-dispatched half of the workgroup (MI355X_MICROARCH.md, two waves per SIMD, item 4: the
    // younger wave of a SIMD loses every issue arbitration to its older partner; one s_setprio for that half, no flips)
    if (p.wide & 2) {
      if (wave >= (WAVES_M * WAVES_N) / 2) __builtin_amdgcn_s_setprio(1);
    }
    gload(0);
    for (;;) {
    if constexpr (ACT) {
      load_sc(0);
#pragma unroll
      for (int j = 0; j < A_PER_T; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[j][e] = act1(ra[j][e], e);
    }
    sstore(sm);
    gload(last < 1 ? last : 1);
    if constexpr (ACT) {           // steady state at the top of an iteration: ra[0] activated, ra[1..] raw, (sc, sh) of that chunk
      load_sc(last < 1 ? last : 1);
#pragma unroll
      for (int e = 0; e < 4; ++e) ra[0][e] = act1(ra[0][e], e);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    __syncthreads();
    for (int kc = 0; kc < p.nk; ++kc) {
      unsigned char* cur = sm + (kc & 1) * STAGE;
      unsigned char* nxt = sm + ((kc & 1) ^ 1) * STAGE;
      const int kn = kc + 2 < last ? kc + 2 : last;
      frags(cur, 0);
      unsigned h0, h1, m0_, m1_, l0, l1;
      float r0, r1, r2, r3;
      const int po1 = ((2 + fh) ^ fsw) << 4;
      auto micro = [&](const int s) {
        const int f = s / 8, step = s % 8;                               // float4 f of this thread (first the A rows, then the B rows)
        if constexpr (ACT) {
          // one value of the NEXT A float4 per even micro-step: ra[f + 1] (this chunk's successor, kc+1) under the A float4s
          // before the last, ra[0] of chunk kc+2 (fetched at the end of float4 0) under the last float4 of all; (sc, sh) move on
          // to chunk kc+2 in between (first B micro-step)
          if (s == 8 * A_PER_T) load_sc(kn);
          if (step % 2 == 0 && (f + 1 < A_PER_T || f == A_PER_T + B_PER_T - 1)) {
            const int g = f + 1 < A_PER_T ? f + 1 : 0, e = step / 2;
            ra[g][e] = act1(ra[g][e], e);
          }
        }
        const f32x4 v = f < A_PER_T ? ra[f < A_PER_T ? f : 0] : rb[f < A_PER_T ? 0 : f - A_PER_T];
        unsigned char* d = f < A_PER_T ? nxt + st_off + f * RPP * ROWB : nxt + 3 * PLANE_A + st_off + (f - A_PER_T) * RPP * ROWB;
        const int pl = f < A_PER_T ? PLANE_A : PLANE_B;
        if constexpr (DBG == 4) {                     // timing experiment: no split arithmetic
          if (step == 2) *reinterpret_cast<uint2*>(d) = make_uint2(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]));
          else if (step == 5) *reinterpret_cast<uint2*>(d + pl) = make_uint2(__builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3]));
          else if (step == 7) *reinterpret_cast<uint2*>(d + 2 * pl) = make_uint2(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[3]));
        } else {
        if (step == 0) { h0 = cvt2(v[0], v[1]); h1 = cvt2(v[2], v[3]); }
        else if (step == 1) { r0 = v[0] - lo_f(h0); r1 = v[1] - hi_f(h0); }
        else if (step == 2) { r2 = v[2] - lo_f(h1); r3 = v[3] - hi_f(h1); if constexpr (DBG != 1) *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1); }
        else if (step == 3) { m0_ = cvt2(r0, r1); m1_ = cvt2(r2, r3); }
        else if (step == 4) { r0 -= lo_f(m0_); r1 -= hi_f(m0_); }
        else if (step == 5) { r2 -= lo_f(m1_); r3 -= hi_f(m1_); if constexpr (DBG != 1) *reinterpret_cast<uint2*>(d + pl) = make_uint2(m0_, m1_); }
        else if (step == 6) { l0 = cvt2(r0, r1); l1 = cvt2(r2, r3); }
        else if constexpr (DBG == 1) { acc[0][0][0] += __builtin_bit_cast(float, l0 ^ l1 ^ h0 ^ m1_) * 1e-30f; }
        else *reinterpret_cast<uint2*>(d + 2 * pl) = make_uint2(l0, l1);
        }
        if (step == 7 && DBG != 3) {
          // this float4's registers are free: fetch its successor (chunk kc+2)
          if (f < A_PER_T) ra[f < A_PER_T ? f : 0] = *reinterpret_cast<const f32x4*>(asrc[f < A_PER_T ? f : 0] + kn * kBK);
          else rb[f < A_PER_T ? 0 : f - A_PER_T] = *reinterpret_cast<const f32x4*>(bsrc + (f - A_PER_T) * b_step + kn * kBK);
        }
      };
#pragma unroll
      for (int u = 0; u < NMF; ++u) {
        const int ks = u / (NMF / 2), t = (u % (NMF / 2)) / (TM * TN), i = (u % (TM * TN)) / TN, j = u % TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][TA[t]][i], bf[ks][TB[t]][j], acc[i][j], 0, 0, 0);
        if (u < NFR && DBG != 5) {                                       // fragments of k-step 1: one per MFMA
          const int q = u / (TM + TN), w = u % (TM + TN);
          if (w < TM) af[1][q][w] = *reinterpret_cast<const bf16x8*>(cur + fa_off + q * PLANE_A + w * 32 * ROWB + po1);
          else bf[1][q][w - TM] = *reinterpret_cast<const bf16x8*>(cur + fb_off + q * PLANE_B + (w - TM) * 32 * ROWB + po1);
        }
#pragma unroll
        for (int sidx = u * NST / NMF; sidx < (u + 1) * NST / NMF; ++sidx) micro(sidx);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (DBG != 2) __syncthreads();
    }
    // ---- end of the tile: fetch the first chunk of the next one, then store this one ----
    const int em0 = m0, en0 = n0, emt = mt;
    ConvArgs q = p;
    q.y = ybase;
    const int vn = v + (int)gridDim.x;
    const bool has_next = PERSIST && vn < total;
    if (has_next) {
      set_tile(vn);
      gload(0);
    }
    conv_epilogue<BM, BN, WAVES_M, WAVES_N, TM, TN, false>(q, acc, smem, em0, en0, emt, tid, lane, wm, wn);
    if (!has_next) break;
    v = vn;
    __syncthreads();               // the epilogue's LDS tile is read: stage 0 may be overwritten
    }
  }
}

__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin,
                                   int ks, int mode, int rows_pad, int red4, int kpad) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)rows_pad * kpad) return;
  const int row = (int)(idx / kpad), k = (int)(idx - (long long)row * kpad);
  const int taps = ks * ks;
  const int tap = k / red4, r = k - tap * red4;
  const int rows = mode == 0 ? cout : cin, red = mode == 0 ? cin : cout;
  float v = 0.f;
  if (row < rows && tap < taps && r < red) {
    const int ky = tap / ks, kx = tap - ky * ks;
    if (mode == 0)
      v = w[(((long long)row * cin + r) * ks + ky) * ks + kx];
    else   // data gradient: correlate dy with the 180-degree rotated filter, channels swapped
      v = w[(((long long)r * cin + row) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];
  }
  out[idx] = v;
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---- tile configurations -----------------------------------------------------------------------
// Measured on MI355X (profiles/r01_summary.txt, section D):
//  * kTile64     64x64, one LDS stage (18 KB -> 8 workgroups = 32 waves per CU): 104-116 TFLOP/s on every
//                layer of the path and insensitive to code placement.                      [DEFAULT]
//  * kTile128x32 128x32, one stage: layers with <= 32 output channels (a 64-wide tile idles half its MFMAs).
//  * kTile128    128x128, two stages, + 64x64 tiles for the rows of the last partial round of
//                co-resident workgroups ("tail split"): up to 120 TFLOP/s on the 13x13 layers but swings
//                between 87 and 120 with unrelated code changes.
//  * kDma*       direct global->LDS DMA staging (XOR-swizzled linear LDS image; 2 stages or a 3-deep
//                ring with counted vmcnt): same throughput as register staging on the direct layers; the
//                128x128 DMA variant is the default of the K >= 1024 Winograd batches (conv_gemm_batched:
//                MFMA pipe 83 % busy against 69 % for 64x64).
//  * kTile128x64 128x64, one stage: measured equal to 64x64 (kept as a tuning aid).
// FSD_CONV_TILE=<letter> forces one configuration (tuning aid, read once per process).
enum TileId { kTile64 = 0, kTile128x32, kTile128, kDma128, kDma128Ring, kDma64, kTile128x64, kNumTiles };
struct TileCfg { int bm, bn; };
constexpr TileCfg kCfgs[kNumTiles] = {{64, 64}, {128, 32}, {128, 128}, {128, 128}, {128, 128}, {64, 64}, {128, 64}};
constexpr int kSlots = 512;   // co-resident 128x128 workgroups on the chip (256 CUs x 2)

inline int tile_cfg(int cin, int ksize, int cout, bool nchw = false) {
  static const char* env = getenv("FSD_CONV_TILE");
  if (env && env[0] >= 'a' && env[0] < 'a' + kNumTiles) return env[0] - 'a';
  // outputs no wider than 32 channels: a 64-wide tile would idle half the MFMAs.  (Short reductions, K <= 320, used to
  // take this tile too; re-measured with the current kernel the 64x64 tile is 8-10 % faster there: 32->64 @208x208
  // 1.27 -> 1.18 ms, 128->64 1x1 @104x104 0.154 -> 0.144 ms.)
  (void)cin;
  if (cout <= 32 && !nchw) return kTile128x32;
  // split arithmetic, 3x3 into 64 channels (32 -> 64 at 208x208): 128x64 tiles 0.994 ms against 1.076 for 64x64
  if (fsd_conv::f32_split_on() && ksize == 3 && cout == 64 && !nchw) return kTile128x64;
  return kTile64;
}

// conv_gemm_split8_kernel launches: one tile per workgroup (default), or persistent (at most one workgroup per CU, a multiple
// of 8 for the XCD order, each walking total / grid tiles)
template <typename K>
int launch_split8_grid(K k, const ConvArgs& a, size_t lds, bool persist, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8)
      v = 256;
    n_cu = v;
  }
  const long long total = (long long)a.m_tiles * a.n_tiles * a.batches;
  static const char* prio_env = FSD_TUNE("FSD_SPLIT8_PRIO");            // tuning aid: 1 = s_setprio 1 for waves 4-7
  ConvArgs b = a;
  if (prio_env && prio_env[0] == '1' && b.wide) b.wide |= 2;      // (bit 1 rides on the wide-epilogue flag: still truthy)
  long long grid = total < n_cu || !persist ? total : n_cu;
  if (grid >= 8 && grid < total) grid = grid / 8 * 8;
  const double rows = (double)a.M - (double)a.m_base;
  fsd_prof::Scope prof(fsd_prof::kGemmFwd, 2.0 * rows * a.Cout * ((double)a.nk * kBK) * a.batches, stream);
  FSD_LAUNCH(k, dim3((unsigned)grid), dim3(512), lds, stream, b);
  return (int)hipGetLastError();
}

template <int BM, int ACT>
int launch_split8_t(const ConvArgs& a, size_t lds, hipStream_t stream) {
#ifdef FSD_EXPERIMENTS
  static const char* dbg = FSD_TUNE("FSD_SPLIT8_DBG");                 // timing experiments, wrong results
  if (dbg && BM == 256 && ACT == 0) {
    switch (dbg[0]) {
      case '1': return launch_split8_grid(conv_gemm_split8_kernel<256, 128, 4, 2, true, 0, false, 1>, a, lds, false, stream);
      case '2': return launch_split8_grid(conv_gemm_split8_kernel<256, 128, 4, 2, true, 0, false, 2>, a, lds, false, stream);
      case '3': return launch_split8_grid(conv_gemm_split8_kernel<256, 128, 4, 2, true, 0, false, 3>, a, lds, false, stream);
      case '4': return launch_split8_grid(conv_gemm_split8_kernel<256, 128, 4, 2, true, 0, false, 4>, a, lds, false, stream);
      case '5': return launch_split8_grid(conv_gemm_split8_kernel<256, 128, 4, 2, true, 0, false, 5>, a, lds, false, stream);
      default: break;
    }
  }
  static const char* env = FSD_TUNE("FSD_SPLIT8_PERSIST");
  if (env && env[0] == '1') return launch_split8_grid(conv_gemm_split8_kernel<BM, 128, 4, 2, true, ACT, true>, a, lds, true, stream);
#endif
  return launch_split8_grid(conv_gemm_split8_kernel<BM, 128, 4, 2, true, ACT, false>, a, lds, false, stream);
}

template <typename K>
int launch_kernel(K k, const ConvArgs& a, size_t lds, int threads, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  // issued MFMA work of this launch: every output row x column x (padded) reduction element, all batches
  const double rows = (double)a.M - (double)a.m_base;
  fsd_prof::Scope prof(fsd_prof::kGemmFwd, 2.0 * rows * a.Cout * ((double)a.nk * kBK) * a.batches, stream);
  FSD_LAUNCH(k, dim3(a.m_tiles * a.n_tiles, a.batches * a.ksplit), dim3(threads), lds, stream, a);
  return (int)hipGetLastError();
}

inline bool split_on() { return fsd_conv::f32_split_on(); }

template <int BM, int BN, int WM, int WN, int STAGES, bool GLDS = false>
int launch(const ConvArgs& a, bool nchw, hipStream_t stream) {
  const bool fast = a.cpt > 0;
  constexpr int NT = WM * WN * 64;
  const size_t tile_bytes = a.wide ? (size_t)BM * BN * sizeof(float) : 0;       // the wide epilogue's [BM][BN] tile
  if (split_on()) {
    // one stage of three planes is 240 B per tile row: 128x128 -> 60 KB (two workgroups per CU), 64x64 -> 30 KB
    constexpr int SS = (BM + BN >= 256) ? 1 : (STAGES > 2 ? 2 : STAGES);
    size_t lds_s = SS * (size_t)3 * (BM + BN) * kLdH * sizeof(unsigned short);
    if (lds_s < tile_bytes) lds_s = tile_bytes;
    if (nchw)
      return fast ? launch_kernel(conv_gemm_kernel<BM, BN, WM, WN, true, SS, true, false, true>, a, lds_s, NT, stream)
                  : launch_kernel(conv_gemm_kernel<BM, BN, WM, WN, true, SS, false, false, true>, a, lds_s, NT, stream);
    return fast ? launch_kernel(conv_gemm_kernel<BM, BN, WM, WN, false, SS, true, false, true>, a, lds_s, NT, stream)
                : launch_kernel(conv_gemm_kernel<BM, BN, WM, WN, false, SS, false, false, true>, a, lds_s, NT, stream);
  }
  if constexpr (GLDS) {
    if (fast && !a.in_scale) {   // the DMA path needs the per-tap fast path (and an input that needs no work on the way in)
      size_t lds_g = STAGES * (size_t)(BM + BN) * kBK * sizeof(float);
      if (lds_g < tile_bytes) lds_g = tile_bytes;
      return nchw ? launch_kernel(conv_gemm_kernel<BM, BN, WM, WN, true, STAGES, true, true>, a, lds_g, NT, stream)
                  : launch_kernel(conv_gemm_kernel<BM, BN, WM, WN, false, STAGES, true, true>, a, lds_g, NT, stream);
    }
  }
  constexpr int RS = STAGES > 2 ? 2 : STAGES;
  size_t lds = RS * (size_t)(BM + BN) * kLd * sizeof(float);
  if (lds < tile_bytes) lds = tile_bytes;
  if (nchw)
    return fast ? launch_kernel(conv_gemm_kernel<BM, BN, WM, WN, true, RS, true, false>, a, lds, NT, stream)
                : launch_kernel(conv_gemm_kernel<BM, BN, WM, WN, true, RS, false, false>, a, lds, NT, stream);
  return fast ? launch_kernel(conv_gemm_kernel<BM, BN, WM, WN, false, RS, true, false>, a, lds, NT, stream)
              : launch_kernel(conv_gemm_kernel<BM, BN, WM, WN, false, RS, false, false>, a, lds, NT, stream);
}

// float4 epilogue (conv_epilogue): FSD_CONV_WIDE=0 switches it off (tuning aid)
inline bool wide_ok(const float* y, long long y_ld, int cout) {
  static const char* env = FSD_TUNE("FSD_CONV_WIDE");
  return !(env && env[0] == '0') && cout % 4 == 0 && y_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
}

// Tail splitting for kTile128: rows covered by WHOLE rounds of co-resident workgroups use 128x128 tiles; the
// remaining rows (a partial round that would leave CUs idle for up to a full tile time) are cut into 64x64
// tiles, which balance ~4x finer.
struct RowPlan { int main_m_tiles; int tail_m_tiles; };
inline RowPlan plan_rows(long long pixels, int cout, int cfg) {
  const int bm = kCfgs[cfg].bm, bn = kCfgs[cfg].bn;
  const int m_tiles = (int)((pixels + bm - 1) / bm);
  RowPlan r{m_tiles, 0};
  if (cfg != kTile128) return r;
  const int n_tiles = (cout + bn - 1) / bn;
  const long long total = (long long)m_tiles * n_tiles;
  if (total <= kSlots || total % kSlots == 0) return r;
  const int main_m = (int)((total / kSlots) * kSlots / n_tiles);
  if (main_m <= 0 || main_m >= m_tiles) return r;
  r.main_m_tiles = main_m;
  r.tail_m_tiles = (int)((pixels - (long long)main_m * bm + 63) / 64);
  return r;
}

// Tile choice of the batched plain GEMMs (Winograd positions).  64x64 single-stage tiles win up to K = 256; from
// K = 512 with >= 512 output channels (the 13x13 layers) the 128x128 tile with DMA staging (global_load_lds, two
// stages) is faster: -10 % at 512 -> 1024, -8 % at 1024 / 1280 -> 1024; at 256 -> 512 @ 26x26 it loses 4 % (measured
// round 2, tools/layer_bench.py).  FSD_WINO_TILE = a|c|d|e overrides.  'a' 64x64, 'c' 128x128 register-staged,
// 'd' 128x128 DMA, 'e' 128x64.
// The 128-row tile needs rows to fill it: the reweighting net's last layers have 20-80 tile rows per position (3x3 and
// 7x7 maps of 20 supports), where a 128-row tile is 84 % padding -- those keep the 64x64 tile (4 launches per step,
// 75-140 us each with the 128-row tile).
// Split arithmetic (six bf16 MFMA terms): the 128x128 register-staged tile wins on EVERY Winograd layer of the episode
// (measured round 3, gemm ms per launch, 64x64 / 128x128 / 128x64: 104x104 64->128 0.347 / 0.299 / 0.335, 52x52 0.252 /
// 0.219 / 0.234, 26x26 0.243 / 0.221 / 0.234, 13x13 512->1024 0.303 / 0.251 / 0.274, 1024->1024 0.585 / 0.490 / 0.540):
// the matrix work of a tile is a third of the native kernel's, so the operand traffic per MFMA decides.
// 'k' (round 4, split arithmetic only): 256x128 on 8 waves, two LDS stages (conv_gemm_split8_kernel).
inline char batched_pick(long long rows, int cin, int cout) {
  static const char* env = FSD_TUNE("FSD_WINO_TILE");
  if (env) {
    if (env[0] == 'k' && !fsd_conv::f32_split_on()) return 'c';
    return env[0];
  }
  if (fsd_conv::f32_split_on()) {
    static const char* k_env = getenv("FSD_WINO_SPLIT8");      // tuning aid: 0 keeps the 4-wave tiles everywhere
    // (K <= 128: those launches are bound by the V / M traffic, where two 4-wave workgroups per CU hide each other's epilogue)
    if (!(k_env && k_env[0] == '0') && rows >= 512 && cout >= 128 && cout % 128 == 0 && cin >= 256) return 'k';
    // 'b': at most 32 rows per position (two images at 13x13, the reweighting net's 3x3 maps of 20 supports): a 32x128 tile
    // (1 x 4 waves).  Such a launch is bound by what a CU does per k-chunk -- the split of every staged value, the MFMAs, the
    // LDS round trip (see batched_ksplit) -- and half of a 64x64 tile's rows are padding: 32x128 stages 5120 values for four
    // useful 32x32 tiles where 64x64 stages 4096 for two.  FSD_WINO_TILE32=0 keeps 64x64.
    static const char* b_env = FSD_TUNE("FSD_WINO_TILE32");
    if (!(b_env && b_env[0] == '0') && rows <= 32 && cout >= 128 && cout % 128 == 0) return 'b';
    return rows >= 96 && cout > 64 ? 'c' : 'a';
  }
  return cin >= 512 && cout >= 512 && cout % 128 == 0 && rows >= 256 ? 'd' : 'a';
}

// 1x1 convolutions on the 8-wave split kernel (K >= 256, whole 128-column tiles, >= 512 rows; FSD_CONV1_SPLIT8=0 keeps the 64x64
// tiles).  Alone the kernel is 20-25 % faster than the 64x64 tiles on these launches (13x13 1024->512 0.104 -> 0.078 ms, 26x26
// 512->256 0.098 -> 0.084, 52x52 256->128 0.106 -> 0.085, incl. the activation-on-load variant); in the step that is 0.1-0.15 ms
// (same box, four runs per arm: 25.81 against 25.91 ms).  (A first A/B had shown 33.7 against 27.0 ms and was blamed on the
// 144 KB workgroups waiting behind small-LDS kernels of the side streams; it was the unbounded queue depth of section 5.0 --
// with two steps in flight the difference is the one above.)  The row tiling (and with it the number of BatchNorm partial rows,
// fsd_conv_row_tiles) does not depend on the activation arguments.
inline bool split8_1x1(long long pixels, int cin, int cout, int ksize, bool nchw) {
  static const char* env = FSD_TUNE("FSD_CONV1_SPLIT8");
  if (env && env[0] == '0') return false;
  // ... and only when the 256x128 tiles cover at least half of the 256 CUs (two images at 26x26 are 10 such tiles: 64x64 there)
  return fsd_conv::f32_split_on() && ksize == 1 && !nchw && cin % kBK == 0 && cin >= 256 && cout % 128 == 0 &&
         (pixels / 256) * (cout / 128) >= 128;
}

// How many of the `batches` positions of a batched launch go to the 256-row tiles (the rest: 128-row tiles, second launch).
// Cost model: rounds of 256 co-resident workgroups; a 128-row tile takes 0.55 of a 256-row one.  FSD_SPLIT8_TAIL=0: all 256.
inline int split8_main_positions(long long rows, int cout, int batches) {
  static const char* env = FSD_TUNE("FSD_SPLIT8_TAIL");
  if ((env && env[0] == '0') || batches < 2) return batches;
  const long long nt = (cout + 127) / 128;
  const long long tp256 = (rows + 255) / 256 * nt, tp128 = (rows + 127) / 128 * nt;
  int best = batches;
  double best_cost = (double)((batches * tp256 + 255) / 256);
  for (int pm = batches - 1; pm >= 0; --pm) {
    const double cost = (double)((pm * tp256 + 255) / 256) + 0.55 * (double)(((batches - pm) * tp128 + 255) / 256);
    if (cost < best_cost - 0.2) { best_cost = cost; best = pm; }     // (a second launch has to buy at least a fifth of a round)
  }
  return best;
}

// -1: not decided yet (first use reads FSD_F32_SPLIT; default 1 = split arithmetic, FSD_F32_SPLIT=0 = native fp32 MFMA)
std::atomic<int> g_f32_split{-1};

}  // namespace

bool fsd_conv::f32_split_on() {
  int v = g_f32_split.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* env = getenv("FSD_F32_SPLIT");
    v = (env && env[0] == '0') ? 0 : 1;
    int expected = -1;
    if (!g_f32_split.compare_exchange_strong(expected, v)) v = expected;      // somebody set the mode meanwhile: theirs wins
  }
  return v == 1;
}

extern "C" int fsd_f32_gemm_mode(int mode) {
  const int prev = fsd_conv::f32_split_on() ? 1 : 0;
  if (mode == 0 || mode == 1) g_f32_split.store(mode);
  return prev;
}

inline void batched_tile(char pick, int* bm, int* bn) {
  const int big = pick == 'c' || pick == 'd' || pick == 'k';
  *bm = pick == 'k' ? 256 : pick == 'b' ? 32 : (big || pick == 'e' || pick == 'h') ? 128 : 64;
  *bn = big || pick == 'b' ? 128 : 64;
}

int fsd_conv::conv_gemm_batched_plan(long long rows, int cin, int cout, int* bm_out, int* bn_out, int* dma_out) {
  const char pick = batched_pick(rows, cin, cout);
  int bm, bn;
  batched_tile(pick, &bm, &bn);
  if (bm_out) *bm_out = bm;
  if (bn_out) *bn_out = bn;
  if (dma_out) *dma_out = pick == 'd';
  return (int)((rows + bm - 1) / bm);
}

// Split-K of a batched launch.  Two images at 13x13 are 32 tile rows per Winograd position: 36 positions x 16 column tiles =
// 576 workgroups of 64x64, ~2 per CU, each with ONE 16 KB k-chunk in flight at a time (single LDS stage) -- the launch streams
// its 151 MB of transformed weights (1024 -> 1024) at 2.8 TB/s, 54 us.  Cutting K puts more chunks in flight; the slices land
// side by side and the output transform adds them on load (they are 4.7 MB each at that size).
// MEASURED SLOWER, so it is OPT-IN (FSD_KSPLIT=a picks the slices automatically -- launches short of workgroups, < 1024, with
// >= 8 chunks per slice -- =N forces N): two images, whole forward 0.83 -> 0.87 ms; four images 0.98 -> 1.05; train step
// unchanged.  Per kernel (rocprofv3, 13 Winograd layers of one forward): the 64x64 GEMMs 406 -> 382 us with 2 slices, 373 with
// 4 -- and the output transform that adds the slices 6.5 -> 16.5 / 35.7 us per launch.  So the launch is NOT short of bytes in
// flight, and not bound by its access pattern either (tools/probes/panel_stream_probe.hip: the same 576 workgroups reading the
// same panels the same way, one chunk ahead of a barrier, stream 151 MB in 33.7 us = 4.5 TB/s; 5.7 TB/s if a chunk's 64 x 32
// block were contiguous).  It is bound by what a CU does per chunk whatever the concurrency: the split of 16 operand values per
// thread (~160 VALU instructions = 640 cycles a wave) + 12 MFMAs (384) + the LDS round trip, serialised by the single stage --
// 72 chunk-tiles per CU x ~1500 cycles = 54 us.  Operands that are constant between calls (inference weights) would have to
// arrive already split for these launches to approach the 30 us their bytes cost.
int fsd_conv::batched_ksplit(long long rows, int cin, int cout, int batches) {
  static const char* env = FSD_TUNE("FSD_KSPLIT");
  if (!env || cin % kBK != 0) return 1;
  const int nk = cin / kBK;
  const char pick = batched_pick(rows, cin, cout);
  if (pick != 'a' && pick != 'c') return 1;                       // the register-staged 4-wave kernels (plain main loop)
  if (env[0] != 'a') {
    const int forced = atoi(env);
    return forced >= 1 && forced <= 16 && nk % forced == 0 ? forced : 1;
  }
  const int bm = pick == 'c' ? 128 : 64, bn = bm;
  const long long total = ((rows + bm - 1) / bm) * ((cout + bn - 1) / bn) * batches;
  int s = 1;
  while (s < 8 && total * s < 1024 && nk % (2 * s) == 0 && nk / (2 * s) >= 8) s *= 2;
  return s;
}

int fsd_conv::conv_gemm_batched(const float* x, long long x_ld, long long x_bs, const float* w_packed, long long w_bs,
                                float* y, long long y_ld, long long y_bs, long long rows, int cin, int cout, int batches,
                                hipStream_t stream, int ksplit, long long y_ks) {
  if (cin % kBK != 0 || rows < 1 || rows > 0x7fffffffLL - 512 || (rows + 1) * x_ld >= 0xffffffffLL) return FSD_ERR_UNSUPPORTED;
  if (ksplit < 1 || (ksplit > 1 && (ksplit != batched_ksplit(rows, cin, cout, batches) || (y_ks & 3)))) return FSD_ERR_ARG;
  ConvArgs a;
  a.ksplit = ksplit; a.y_ks = y_ks;
  a.x = x; a.w = w_packed; a.bias = nullptr; a.y = y; a.bn_partial = nullptr;
  a.x_ld = x_ld; a.y_ld = y_ld;
  a.H = 1; a.W = (int)rows; a.HW = (int)rows; a.M = (int)rows;     // one "image" of `rows` pixels, 1x1 taps
  a.Cout = cout; a.ks = 1; a.pad = 0;
  a.cpg = cin / 4;
  a.kgroups = a.cpg;
  a.Kpad = cin;
  a.nk = cin / kBK;
  a.cpt = cin / kBK;
  const char pick = batched_pick(rows, cin, cout);
  const int big = pick == 'c' || pick == 'd' || pick == 'k';
  int bm, bn;
  batched_tile(pick, &bm, &bn);
  a.m_tiles = (int)((rows + bm - 1) / bm);
  a.n_tiles = (cout + bn - 1) / bn;
  a.m_base = 0;
  a.part_base = 0;
  a.batches = batches;
  a.slope = 1.f;
  a.x_bs = x_bs; a.w_bs = w_bs; a.y_bs = y_bs;
  a.wide = wide_ok(y, y_ld, cout) && y_bs % 4 == 0;
  static const char* flat_env = FSD_TUNE("FSD_CONV_FLAT_XCD");        // tuning aid: 0 / 1 force the order
  a.flat_xcd = flat_env ? (flat_env[0] == '1') : (fsd_conv::f32_split_on() ? 1 : 0);
  a.in_scale = a.in_shift = nullptr; a.in_slope = 1.f;
  if (pick == 'k') {
    constexpr size_t lds_k = 2 * 3 * (size_t)(256 + 128) * 64;        // two stages of three planes; >= the 256x128 fp32 tile
    static_assert(lds_k >= (size_t)256 * 128 * sizeof(float), "the wide epilogue's tile must fit the staging space");
    // Tail balance: one workgroup per CU means whole ROUNDS of 256 tiles -- 36 positions x 32 tiles = 1152 = 4.5 rounds pay for
    // 5.  The last positions go to a second launch on 128-row tiles (half the time each): 32 positions = 4 rounds + 4 positions
    // x 64 half-tiles = one half-length round.
    const int p_main = split8_main_positions(rows, cout, batches);
    if (p_main < batches) {
      ConvArgs t = a;
      t.batches = batches - p_main;
      t.x = a.x + (long long)p_main * a.x_bs;
      t.w = static_cast<const float*>(a.w) + (long long)p_main * a.w_bs;
      t.y = a.y + (long long)p_main * a.y_bs;
      t.m_tiles = (int)((rows + 127) / 128);
      constexpr size_t lds_t = 2 * 3 * (size_t)(128 + 128) * 64;
      static_assert(lds_t >= (size_t)128 * 128 * sizeof(float), "the wide epilogue's tile must fit the staging space");
      if (p_main > 0) {
        a.batches = p_main;
        const int rc = launch_split8_t<256, 0>(a, lds_k, stream);
        if (rc != 0) return rc;
      }
      return launch_split8_t<128, 0>(t, lds_t, stream);
    }
    return launch_split8_t<256, 0>(a, lds_k, stream);
  }
  if (pick == 'b') return launch<32, 128, 1, 4, 1>(a, false, stream);
  if (pick == 'e') return launch<128, 64, 4, 1, 1>(a, false, stream);
  if (pick == 'h') return launch<128, 64, 2, 2, 2, true>(a, false, stream);      // 128x64 DMA, two stages (48 KB: 3 per CU)
  if (pick == 'f') return launch<64, 64, 2, 2, 2, true>(a, false, stream);       // 64x64 DMA, two stages
  if (pick == 'g') return launch<64, 64, 2, 2, 3, true>(a, false, stream);       // 64x64 DMA, 3-deep ring
  if (big) return pick == 'c' ? launch<128, 128, 2, 2, 2>(a, false, stream) : launch<128, 128, 2, 2, 2, true>(a, false, stream);
  return launch<64, 64, 2, 2, 1>(a, false, stream);
}

extern "C" size_t fsd_packed_weight_elems(int rows, int red, int ksize) {
  return (size_t)round_up(rows, 128) * (size_t)round_up(ksize * ksize * round_up(red, 4), kBK);
}

extern "C" int fsd_pack_conv_weight(const float* w_oihw, float* w_packed, int cout, int cin, int ksize,
                                    int mode, hipStream_t stream) {
  (void)hipGetLastError();   // drop a stale error left by someone else's earlier call
  if (!w_oihw || !w_packed || cout < 1 || cin < 1 || (ksize != 1 && ksize != 3) || (mode != 0 && mode != 1))
    return FSD_ERR_ARG;
  const int rows = mode == 0 ? cout : cin, red = mode == 0 ? cin : cout;
  const int rows_pad = round_up(rows, 128), red4 = round_up(red, 4);
  const int kpad = round_up(ksize * ksize * red4, kBK);
  const long long total = (long long)rows_pad * kpad;
  FSD_LAUNCH(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w_oihw,
                     w_packed, cout, cin, ksize, mode, rows_pad, red4, kpad);
  return (int)hipGetLastError();
}

extern "C" int fsd_conv_row_tiles(int batch, int height, int width, int cout, int cin, int ksize) {
  // ONE plan for this query and for the launch in fsd_conv2d_fwd_ex: halo kernel (a row per 8 x 16 block), 8-wave 1x1 tiles,
  // or the implicit-GEMM row tiles
  if (fsd_conv::halo_ok(height, width, cin, cout, ksize, false)) return batch * (height / 8) * (width / 16);
  const long long pixels = (long long)batch * height * width;
  if (split8_1x1(pixels, cin, cout, ksize, false)) return (int)(pixels / 256 + (pixels % 256 + 63) / 64);
  const RowPlan r = plan_rows(pixels, cout, tile_cfg(cin, ksize, cout));
  return r.main_m_tiles + r.tail_m_tiles;
}

extern "C" int fsd_conv2d_fwd(const float* x, long long x_ld, const float* w_packed, const float* bias,
                              float* y, long long y_ld, float* bn_partial, int batch, int height, int width,
                              int cin, int cout, int ksize, int out_nchw, hipStream_t stream) {
  return fsd_conv2d_fwd_act(x, x_ld, w_packed, bias, y, y_ld, bn_partial, batch, height, width, cin, cout, ksize, out_nchw,
                            1.f, stream);
}

extern "C" int fsd_conv2d_fwd_act(const float* x, long long x_ld, const float* w_packed, const float* bias,
                                  float* y, long long y_ld, float* bn_partial, int batch, int height, int width,
                                  int cin, int cout, int ksize, int out_nchw, float slope, hipStream_t stream) {
  return fsd_conv2d_fwd_ex(x, x_ld, w_packed, bias, y, y_ld, bn_partial, batch, height, width, cin, cout, ksize, out_nchw, slope,
                           nullptr, nullptr, 1.f, stream);
}

extern "C" int fsd_conv2d_fwd_ex(const float* x, long long x_ld, const float* w_packed, const float* bias,
                                 float* y, long long y_ld, float* bn_partial, int batch, int height, int width,
                                 int cin, int cout, int ksize, int out_nchw, float slope, const float* in_scale,
                                 const float* in_shift, float in_slope, hipStream_t stream) {
  (void)hipGetLastError();   // drop a stale error left by someone else's earlier call
  if ((in_scale == nullptr) != (in_shift == nullptr)) return FSD_ERR_ARG;
  if (in_scale && cin % kBK != 0) return FSD_ERR_UNSUPPORTED;      // activation on load: whole 32-channel chunks per tap
  if (!x || !w_packed || !y || batch < 1 || height < 1 || width < 1 || cout < 1) return FSD_ERR_ARG;
  if (slope != 1.f && (out_nchw || bn_partial)) return FSD_ERR_UNSUPPORTED;   // activation: NHWC store, no statistics
  if (ksize != 1 && ksize != 3) return FSD_ERR_UNSUPPORTED;
  if (cin < 4 || (cin & 3) || (x_ld & 3) || x_ld < cin) return FSD_ERR_ARG;
  if (!out_nchw && y_ld < cout) return FSD_ERR_ARG;
  if (out_nchw && bn_partial) return FSD_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w_packed) & 15)) return FSD_ERR_ARG;
  const long long pixels = (long long)batch * height * width;
  if (pixels > 0x7fffffffLL - 512) return FSD_ERR_UNSUPPORTED;
  if ((pixels + 1) * x_ld >= 0xffffffffLL) return FSD_ERR_UNSUPPORTED;   // 32-bit element offsets in the kernel
  ConvArgs a;
  a.x = x; a.w = w_packed; a.bias = bias; a.y = y; a.bn_partial = bn_partial;
  a.x_ld = x_ld; a.y_ld = y_ld;
  a.H = height; a.W = width; a.HW = height * width; a.M = (int)pixels;
  a.Cout = cout; a.ks = ksize; a.pad = (ksize - 1) / 2;
  a.cpg = cin / 4;
  a.kgroups = ksize * ksize * a.cpg;
  a.Kpad = round_up(ksize * ksize * cin, kBK);
  a.nk = a.Kpad / kBK;
  a.cpt = (cin % kBK == 0) ? cin / kBK : 0;
  const bool nchw = out_nchw != 0;
  if (fsd_conv::halo_ok(height, width, cin, cout, ksize, nchw)) {
    // narrow 3x3 layers (32 / 64 channels): halo patch split once per workgroup (conv_halo.hip); one BatchNorm
    // partial row per 8 x 16 block -- the count fsd_conv_row_tiles reports for these shapes (the same halo_ok decides there).
    // The implicit-GEMM tiles behind it write the same number of rows for these channel counts unless a tiling is forced:
    // then (or for an unaligned y / an activation on load) a BatchNorm launch is refused rather than left with unwritten rows.
    const bool aligned = (y_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    if (!in_scale && aligned)
      return fsd_conv::conv3x3_halo(x, x_ld, w_packed, a.Kpad, bias, y, y_ld, bn_partial, batch, height, width, cin, cout,
                                    slope, stream);
    const RowPlan r = plan_rows(pixels, cout, tile_cfg(cin, ksize, cout));
    if (bn_partial && r.main_m_tiles + r.tail_m_tiles != batch * (height / 8) * (width / 16)) return FSD_ERR_UNSUPPORTED;
  }
  if (split8_1x1(pixels, cin, cout, ksize, nchw)) {
    // 1x1 convolution = a plain GEMM over the pixel rows: the 8-wave 256x128 split kernel on the WHOLE 256-row tiles (its
    // clamped rows past M would enter the BatchNorm sums), 64x64 tiles for the remaining < 256 rows
    const int full_tiles = (int)(pixels / 256);
    a.n_tiles = (cout + 127) / 128;
    a.m_base = 0; a.part_base = 0; a.batches = 1; a.slope = slope;
    a.wide = wide_ok(y, y_ld, cout);
    a.flat_xcd = 0;
    a.in_scale = in_scale; a.in_shift = in_shift; a.in_slope = in_slope;
    a.x_bs = a.w_bs = a.y_bs = 0;
    if (pixels % 256) {
      ConvArgs t = a;
      t.m_base = full_tiles * 256;
      t.part_base = full_tiles;
      t.m_tiles = (int)((pixels - t.m_base + 63) / 64);
      t.n_tiles = (cout + 63) / 64;
      const int rc = launch<64, 64, 2, 2, 1>(t, false, stream);
      if (rc != 0) return rc;
    }
    a.m_tiles = full_tiles;
    a.M = full_tiles * 256;            // the main launch sees whole tiles only
    constexpr size_t lds_k = 2 * 3 * (size_t)(256 + 128) * 64;
    if (in_scale && in_slope >= 0.f && in_slope <= 1.f)
      return launch_split8_t<256, 1>(a, lds_k, stream);
    if (in_scale) return launch_split8_t<256, 2>(a, lds_k, stream);
    return launch_split8_t<256, 0>(a, lds_k, stream);
  }
  const int cfg = tile_cfg(cin, ksize, cout, nchw);
  const int bm = kCfgs[cfg].bm, bn = kCfgs[cfg].bn;
  const RowPlan plan = nchw ? RowPlan{(int)((pixels + bm - 1) / bm), 0} : plan_rows(pixels, cout, cfg);
  a.m_tiles = plan.main_m_tiles;
  a.n_tiles = (cout + bn - 1) / bn;
  a.m_base = 0;
  a.part_base = 0;
  a.batches = 1;
  a.slope = slope;
  a.wide = !out_nchw && wide_ok(y, y_ld, cout);
  a.flat_xcd = 0;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_slope = in_slope;
  a.x_bs = a.w_bs = a.y_bs = 0;
  if (plan.tail_m_tiles > 0) {
    ConvArgs t = a;
    t.m_base = plan.main_m_tiles * bm;
    t.part_base = plan.main_m_tiles;
    t.m_tiles = plan.tail_m_tiles;
    t.n_tiles = (cout + 63) / 64;
    const int rc = launch<64, 64, 2, 2, 2>(t, nchw, stream);
    if (rc != 0) return rc;
  }
  switch (cfg) {
    case kTile64: return launch<64, 64, 2, 2, 1>(a, nchw, stream);
    case kTile128x32: return launch<128, 32, 4, 1, 1>(a, nchw, stream);
    case kTile128: return launch<128, 128, 2, 2, 2>(a, nchw, stream);
    case kDma128: return launch<128, 128, 2, 2, 2, true>(a, nchw, stream);
    case kDma128Ring: return launch<128, 128, 2, 4, 3, true>(a, nchw, stream);
    case kTile128x64: return launch<128, 64, 4, 1, 1>(a, nchw, stream);
    default: return launch<64, 64, 2, 2, 2, true>(a, nchw, stream);
  }
}
