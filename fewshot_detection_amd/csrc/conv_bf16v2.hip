// bf16 mode of the convolution path (BASELINE configs[2] / [4]: "bf16 MFMA path"): activations, their gradients and
// the packed weights live in HBM as bfloat16, products accumulate in fp32 on v_mfma_f32_32x32x16_bf16 (2.5 PFLOP/s dense
// chip peak, 16x the fp32 matrix rate), BatchNorm statistics come from the fp32 accumulators.
//
// Forward / data gradient (conv_bf16_dma_kernel): implicit GEMM, M = B*H*W pixels, N = Cout, K = taps*Cin.
//   * Both operands are K-contiguous in HBM (NHWC activations, [Cout][tap][Cin] weights), so a k-chunk of a tile row is
//     one 128-byte (BK = 64) or 64-byte (BK = 32, the Cin = 32 layers) run: staged with global_load_lds, 16 B per lane,
//     no staging registers, no ds_write.  The DMA writes LDS linearly (wave base + lane*16 B); bank conflicts are
//     avoided by permuting which 16-byte k-group of its row each lane FETCHES (same cache line) and undoing the
//     permutation in the fragment read: group g of row r sits at g ^ ((r >> 1) & 7) (128-B rows) resp.
//     g ^ ((r >> 2) & 3) (64-B rows), which makes every 16-lane service group of the ds_read_b128 fragment reads hit 16
//     distinct 16-byte slots.  Image-border taps and rows past M fetch a zero page.
//   * The weight rows of a 64-channel block are staged EVEN channels first, then ODD (a free permutation of the DMA
//     source), so the two accumulators of a wave hold channel 2l and 2l+1 in lane l: one v_cvt_pk_bf16_f32 packs them
//     and the epilogue stores 4 bytes per lane, 128 contiguous bytes per pixel row, straight from registers.
//   * Two LDS stages; the DMA of chunk k+1 is in flight under the MFMAs of chunk k.
//
// Weight gradient (wgrad_bf16_tr_kernel): dW[co][tap][ci] = sum_pix dy[pix][co] * x[pix + tap][ci], M = Cout, N = Cin
// per tap, K = pixels.  In HBM both operands are [k = pixel][m = channel] -- transposed with respect to what an MFMA
// operand register wants (8 consecutive k per lane) -- so the tiles are staged as they lie (DMA, 16 channels x 32
// pixels per instruction) and the fragments are read with ds_read_b64_tr_b16, the LDS transpose read.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "fsdet.h"
#include "conv_common.hpp"
#include "profile.hpp"

namespace {

using namespace fsd_conv;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

__device__ __attribute__((aligned(16))) u16 g_zero_page_h[64];     // 128 zero bytes: padding source of the DMA path

struct ConvHArgs {
  const u16* x;        // activations, bf16 NHWC, pixel stride x_ld elements
  const u16* w;        // packed weights, bf16 [rows padded to 128][Kpad]
  const float* bias;   // [Cout] or null
  void* y;             // bf16 NHWC (pixel stride y_ld elements) or float NCHW
  float* bn_partial;   // [m_tiles][Cout][2] or null
  long long x_ld, y_ld;
  int H, W, HW, M, Cout, ks, pad;
  int nk;              // k-chunks
  int cpt;             // chunks per tap (Cin / BK)
  int Kpad;            // packed weight row stride (elements)
  int m_tiles, n_tiles;
  float slope;         // leaky slope applied to (acc + bias) before the bf16 store; 1 = linear
  int wide;            // bf16 NHWC epilogue through LDS with 16-byte stores (needs y 16-byte aligned, y_ld % 8 == 0, Cout % 8 == 0)
};


// global -> LDS DMA of 16 bytes per lane (LDS destination = wave-uniform base + lane * 16).  The address spaces are
// spelled out: with generic pointers the builtin's implicit conversions fail SILENTLY in some template contexts on the
// host pass (no diagnostic, the kernel's host stub is simply not emitted -> undefined symbol at load time).
__device__ __forceinline__ void dma16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  const __bf16 a = (__bf16)lo, b = (__bf16)hi;
  return (unsigned)__builtin_bit_cast(u16, a) | ((unsigned)__builtin_bit_cast(u16, b) << 16);
}

// 4 waves (256 threads, two workgroups per CU) or 8 waves (512 threads, ONE workgroup per CU = two waves per SIMD): the
// larger tiles halve the L2 -> LDS bytes per MFMA of the 128x128 tile -- at 128x128x64 a CU running the matrix pipe flat
// out needs (128+128)*128 B per 512 cycles x 2 workgroups = 64 B/clk of operand delivery, more than the ~56 B/clk/CU the
// L2 sustains (MI355X_MICROARCH.md); 256x256 needs 32 B/clk.
//
// ILV: the DMA pieces of the NEXT chunk are issued between the fragment reads and the MFMAs of the first k-steps of the
// current one (a piece costs the issuing wave 60-180 cycles of issue time, MI355X_MICROARCH.md; up front, all waves pay
// that at once with the matrix pipe idle -- spread out, the SIMD's other wave fills the pipe meanwhile): +5 % on the
// 8-wave tiles, +-1 % on 128x128.
//
// NS: LDS stages.  NS = 2: the chunk after the current one is in flight and fully drained (vmcnt(0)) before the barrier.
// NS > 2: a ring -- chunk kc + NS - 1 is issued while chunk kc is computed, and the wait before the barrier is COUNTED
// (only chunk kc + 1 has to have landed: (NS - 2) * PIECES newer DMA instructions may stay in flight).  Kept as a
// tuning aid (FSD_CONV_H_RING=1): measured 5-15 % SLOWER than two drained stages (see the tile choice below).
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool NCHW_F32_OUT, bool ILV = false, int NS = 2>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_bf16_dma_kernel(ConvHArgs p) {
  static_assert(WAVES_M * WAVES_N == 4 || WAVES_M * WAVES_N == 8, "4 or 8 waves");
  static_assert(BK == 64 || BK == 32, "k-chunk of 64 or 32 bf16");
  static_assert(NS >= 2 && NS <= 6, "2 .. 6 LDS stages");
  static_assert(NS == 2 || ILV, "the ring is written for the interleaved issue");
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int TM = BM / WAVES_M / 32;
  constexpr int TN = BN / WAVES_N / 32;
  static_assert(TM * 32 * WAVES_M == BM && TN * 32 * WAVES_N == BN, "tile = whole 32x32 accumulators per wave");
  static_assert(TN == 2 || (TN == 1 && WAVES_N == 1 && !NCHW_F32_OUT),
                "a wave owns one even/odd pair of 32-channel accumulators, or (32-channel outputs) a single one");
  constexpr int LPR = BK / 8;                 // lanes (16-byte groups) per tile row
  constexpr int RPP = NT / LPR;               // tile rows staged per pass of the workgroup
  constexpr int RPW = 64 / LPR;               // ... per wave instruction
  constexpr int A_PER_T = (BM + RPP - 1) / RPP, B_PER_T = BN / RPP;
  // the last A pass may be partial (192 rows on 128-row passes): whole WAVES skip it (a_short), never single lanes
  constexpr bool A_PARTIAL = A_PER_T * RPP != BM;
  static_assert(B_PER_T * RPP == BN && BM % RPW == 0, "whole B passes; A rows in whole wave instructions");
  constexpr int STAGE = (BM + BN) * BK;       // elements per LDS stage
  constexpr int HSH = BK == 64 ? 1 : 2, HMASK = LPR - 1;
  extern __shared__ __attribute__((aligned(16))) u16 smem_h[];

  const int L = xcd_swizzle(blockIdx.x, gridDim.x);
  const int mt = L / p.n_tiles, nt = L - mt * p.n_tiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
  const int r0 = tid / LPR, pg = tid % LPR;
  const int g_src = pg ^ ((r0 >> HSH) & HMASK);        // logical k-group this lane fetches for its physical slot

  const bool a_short = A_PARTIAL && (A_PER_T - 1) * RPP + wave * RPW >= BM;   // this wave has no rows in the last A pass
  // im2col bookkeeping of this thread's A rows
  int a_y[A_PER_T], a_x[A_PER_T];
  unsigned a_pix[A_PER_T];
#pragma unroll
  for (int j = 0; j < A_PER_T; ++j) {
    const int pix = m0 + r0 + RPP * j;
    const int b = pix / p.HW;
    const int rem = pix - b * p.HW;
    const int yy = rem / p.W;
    a_y[j] = pix < p.M ? yy : -(1 << 20);
    a_x[j] = rem - yy * p.W;
    a_pix[j] = (unsigned)pix;
  }
  // B rows: LDS row r of a 64-row block holds channel 2*(r % 32) + (r / 32) % 2 of that block
  const u16* wrow[B_PER_T];
#pragma unroll
  for (int j = 0; j < B_PER_T; ++j) {
    const int r = r0 + RPP * j;
    const int ch = TN == 2 ? (r & ~63) + 2 * (r & 31) + ((r >> 5) & 1) : r;      // TN = 1: rows in channel order
    wrow[j] = p.w + (long long)(n0 + ch) * p.Kpad + g_src * 8;
  }
  const unsigned x_ld = (unsigned)p.x_ld;
  unsigned a_off[A_PER_T];
  unsigned tap_mask = 0;
  int f_tap = 0, f_cc = 0;
  auto retap = [&]() {
    const int ky = f_tap / p.ks, kx = f_tap - ky * p.ks;
    const int dy = ky - p.pad, dx = kx - p.pad;
    const int shift = dy * p.W + dx;
    tap_mask = 0;
#pragma unroll
    for (int j = 0; j < A_PER_T; ++j) {
      const bool ok = (unsigned)(a_y[j] + dy) < (unsigned)p.H && (unsigned)(a_x[j] + dx) < (unsigned)p.W;
      a_off[j] = ok ? (a_pix[j] + (unsigned)shift) * x_ld + (unsigned)(g_src * 8) : 0u;
      tap_mask |= ok ? (1u << j) : 0u;
    }
  };
  retap();

  // one DMA piece (a wave instruction = RPW tile rows) of chunk kc: pieces [0, A_PER_T) are A rows, the rest B rows
  auto piece = [&](int q, int kc, u16* st) {
    if (q < A_PER_T) {
      if (A_PARTIAL && q == A_PER_T - 1 && a_short) return;
      const u16* src = (tap_mask >> q) & 1u ? p.x + (a_off[q] + (unsigned)f_cc * BK) : g_zero_page_h + pg * 8;
      dma16(src, st + (q * RPP + wave * RPW) * BK);
    } else {
      const int j = q - A_PER_T;
      dma16(wrow[j] + kc * BK, st + (BM + j * RPP + wave * RPW) * BK);
    }
  };
  auto advance = [&]() {                    // after the last piece of a chunk: next chunk's channel offset / tap
    if (++f_cc == p.cpt) {
      f_cc = 0;
      ++f_tap;
      retap();
    }
  };
  auto gload_lds = [&](int kc, u16* st) {
#pragma unroll
    for (int q = 0; q < A_PER_T + B_PER_T; ++q) piece(q, kc, st);
    advance();
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int hl = ((lane & 31) >> HSH) & HMASK;
  auto compute = [&](const u16* st) {
    const u16* sa = st + (wm * TM * 32 + (lane & 31)) * BK;
    const u16* sb = st + (BM + wn * TN * 32 + (lane & 31)) * BK;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      const int ko = (((2 * s + (lane >> 5)) ^ hl) & HMASK) * 8;
      bf16x8 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + i * 32 * BK + ko);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(sb + j * 32 * BK + ko);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = NCHW_F32_OUT ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };

  // compute with the next chunk's DMA pieces spread over the k-steps (two stages: the last step stays DMA-free, cover
  // before the drain; a ring has whole chunks of cover)
  auto compute_ilv = [&](const u16* st, int kc_next, u16* st_next) {
    constexpr int KS = BK / 16, PIECES = A_PER_T + B_PER_T;
    constexpr int PPS = NS > 2 ? (PIECES + KS - 1) / KS : KS > 1 ? (PIECES + KS - 2) / (KS - 1) : PIECES;
    const u16* sa = st + (wm * TM * 32 + (lane & 31)) * BK;
    const u16* sb = st + (BM + wn * TN * 32 + (lane & 31)) * BK;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int ko = (((2 * s + (lane >> 5)) ^ hl) & HMASK) * 8;
      bf16x8 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + i * 32 * BK + ko);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(sb + j * 32 * BK + ko);
      if (st_next != nullptr) {
#pragma unroll
        for (int q = s * PPS; q < (s + 1) * PPS && q < PIECES; ++q) piece(q, kc_next, st_next);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = NCHW_F32_OUT ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (st_next != nullptr) advance();
  };

  if constexpr (NS == 2) {
    gload_lds(0, smem_h);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int kc = 0; kc < p.nk; ++kc) {
      if constexpr (ILV) {
        compute_ilv(smem_h + cur * STAGE, kc + 1, kc + 1 < p.nk ? smem_h + (cur ^ 1) * STAGE : nullptr);
      } else {
        if (kc + 1 < p.nk) gload_lds(kc + 1, smem_h + (cur ^ 1) * STAGE);   // that buffer was last read before the previous barrier
        compute(smem_h + cur * STAGE);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }
  } else {
    constexpr int PIECES = A_PER_T + B_PER_T;
    constexpr int KEEP = (NS - 2) * PIECES;        // DMA instructions of the chunks after the next one
    constexpr int KEEP_S = (NS - 2) * (PIECES - 1);   // waves that skip the partial A pass issue one piece less per chunk
    static_assert(KEEP <= 63, "vmcnt is a 6-bit counter");
    auto wait_next = [&](bool ring_full) {
      if (!ring_full) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (A_PARTIAL && a_short) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP_S) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
    };
#pragma unroll
    for (int c = 0; c < NS - 1; ++c)
      if (c < p.nk) gload_lds(c, smem_h + c * STAGE);
    wait_next(p.nk >= NS - 1);
    __syncthreads();
    int cur = 0, fill = NS - 1;                    // stage being computed / stage the next issue goes to
    for (int kc = 0; kc < p.nk; ++kc) {
      const int nxt = kc + NS - 1;
      // stage `fill` was computed in iteration kc - 1: every wave is past the barrier that ended it
      compute_ilv(smem_h + cur * STAGE, nxt, nxt < p.nk ? smem_h + fill * STAGE : nullptr);
      wait_next(nxt < p.nk);                         // chunk kc + 1 has landed
      __syncthreads();
      cur = cur + 1 == NS ? 0 : cur + 1;
      fill = fill + 1 == NS ? 0 : fill + 1;
    }
  }

  // ---- epilogue ----
  const int c_lane = lane & 31, r_lane = 4 * (lane >> 5);
  if constexpr (!NCHW_F32_OUT && TN == 1) {
    // 32 output channels (the data gradient of a 32-channel layer): lane l holds channel l of 16 pixel rows.  Lane pairs
    // swap one value per register pair so that the even lane stores channels (l, l+1) of row R(r) and the odd lane
    // channels (l-1, l) of row R(r+1): 4-byte stores, 64 contiguous bytes per pixel row, no statistics (launcher).
    const int n = n0 + c_lane;
    const bool odd = lane & 1;
    const float bv = (p.bias != nullptr && n < p.Cout) ? p.bias[n] : 0.f;
    u16* yb = static_cast<u16*>(p.y);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float a = acc[i][0][r] + bv, b = acc[i][0][r + 1] + bv;
        if (p.slope != 1.f) { a = a > 0.f ? a : a * p.slope; b = b > 0.f ? b : b * p.slope; }
        const float got = __shfl_xor(odd ? a : b, 1, 64);
        const int rr = odd ? r + 1 : r;
        const int m = m0 + (wm * TM + i) * 32 + (rr & 3) + 8 * (rr >> 2) + r_lane;
        const int nn = odd ? n - 1 : n;
        if (nn < p.Cout && m < p.M)                                       // Cout is even (launcher)
          *reinterpret_cast<unsigned*>(yb + (long long)m * p.y_ld + nn) = odd ? pack2(got, b) : pack2(a, got);
      }
  } else if constexpr (!NCHW_F32_OUT) {
    // lane l holds channels 2l (accumulator 0) and 2l+1 (accumulator 1) of its wave's 64-channel block
    const int n = n0 + wn * 64 + 2 * c_lane;
    const bool n_ok = n < p.Cout;                                   // Cout is even (launcher)
    float bv0 = 0.f, bv1 = 0.f;
    if (p.bias != nullptr && n_ok) { bv0 = p.bias[n]; bv1 = p.bias[n + 1]; }
    u16* yb = static_cast<u16*>(p.y);
    if (!p.wide) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + r_lane;
          if (n_ok && m < p.M) {
            float v0 = acc[i][0][r] + bv0, v1 = acc[i][1][r] + bv1;
            if (p.slope != 1.f) { v0 = v0 > 0.f ? v0 : v0 * p.slope; v1 = v1 > 0.f ? v1 : v1 * p.slope; }
            *reinterpret_cast<unsigned*>(yb + (long long)m * p.y_ld + n) = pack2(v0, v1);
          }
        }
    }
    if (p.bn_partial != nullptr) {
      // per-tile column sums of the fp32 accumulators (rows past M are exact zeros: zero-page operands)
      float* s_stat = reinterpret_cast<float*>(smem_h);            // [WAVES_M][BN][2]
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[i][j][r];
            s += v;
            q += v * v;
          }
        s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 32, 64);
        if (lane < 32) {
          const int col = wn * 64 + 2 * c_lane + j;
          s_stat[(wm * BN + col) * 2 + 0] = s;
          s_stat[(wm * BN + col) * 2 + 1] = q;
        }
      }
      __syncthreads();
      if (tid < BN) {                                               // (BN <= 256 <= NT)
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_M; ++w) {
          s += s_stat[(w * BN + tid) * 2 + 0];
          q += s_stat[(w * BN + tid) * 2 + 1];
        }
        const int nn = n0 + tid;
        if (nn < p.Cout) {
          float* dst = p.bn_partial + ((long long)mt * p.Cout + nn) * 2;
          dst[0] = s;
          dst[1] = q;
        }
      }
      if (p.wide) __syncthreads();                                  // s_stat is read: the tile below re-uses the space
    }
    if (p.wide) {
      // Wide stores.  A lane's own values are 4 bytes per pixel row (32 store instructions per wave for a 64 x 64 wave
      // tile), and store ISSUE -- not bandwidth -- is what a short-K tile then waits for (MI355X_MICROARCH.md: an
      // epilogue of 16 narrow stores per lane costs ~9 k cycles, wide stores halve it).  The tile goes through LDS once
      // ([BM][BN] bf16, written as it lies in the accumulators, conflict-free) and leaves as 16-byte pieces: a wave
      // instruction = 4 pixel rows x 256 contiguous bytes (BN = 128), 4x fewer store instructions.  Measured at B = 64:
      // 104x104 64->128 0.251 -> 0.223 ms, 52x52 128->256 0.163 -> 0.146, 1x1 128->64 data gradient 0.082 -> 0.058.
      u16* s_tile = smem_h;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ml = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + r_lane;
          float v0 = acc[i][0][r] + bv0, v1 = acc[i][1][r] + bv1;
          if (p.slope != 1.f) { v0 = v0 > 0.f ? v0 : v0 * p.slope; v1 = v1 > 0.f ? v1 : v1 * p.slope; }
          *reinterpret_cast<unsigned*>(s_tile + ml * BN + wn * 64 + 2 * c_lane) = pack2(v0, v1);
        }
      __syncthreads();
      constexpr int PPR = BN / 8;                                   // 16-byte pieces per tile row
      for (int it = tid; it < BM * PPR; it += NT) {
        const int row = it / PPR, pc = it - row * PPR;
        const int m = m0 + row, nn = n0 + pc * 8;
        if (m < p.M && nn < p.Cout)                                 // Cout % 8 == 0 (launcher): whole pieces
          *reinterpret_cast<uint4*>(yb + (long long)m * p.y_ld + nn) = *reinterpret_cast<const uint4*>(s_tile + row * BN + pc * 8);
      }
    }
  } else {
    // operands were swapped: accumulator rows = channels (2*rho + j of the wave's block), columns = pixels
    float* yf = static_cast<float*>(p.y);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + (wm * TM + i) * 32 + c_lane;
      const int b = m / p.HW;
      const int hw = m - b * p.HW;
      const bool m_ok = m < p.M;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + wn * 64 + 2 * ((r & 3) + 8 * (r >> 2) + r_lane) + j;
          if (m_ok && n < p.Cout) {
            const float bv = p.bias != nullptr ? p.bias[n] : 0.f;
            yf[((long long)b * p.Cout + n) * p.HW + hw] = acc[i][j][r] + bv;
          }
        }
    }
  }
}

template <int BM, int BN, int BK, int WM, int WN, bool ILV = false, int NS = 2>
int launch_conv(const ConvHArgs& a, bool nchw, hipStream_t stream) {
  size_t lds = NS * (size_t)(BM + BN) * BK * sizeof(u16);
  if (a.wide && lds < (size_t)BM * BN * sizeof(u16)) lds = (size_t)BM * BN * sizeof(u16);     // the epilogue's [BM][BN] tile
  const dim3 grid(a.m_tiles * a.n_tiles), block(64 * WM * WN);
  fsd_prof::Scope prof(fsd_prof::kGemmBf16, 2.0 * (double)a.M * a.Cout * ((double)a.nk * BK), stream);
  if (nchw) {
    if constexpr (BN >= 64 && WM * WN == 4) {
      auto k = conv_bf16_dma_kernel<BM, BN, BK, WM, WN, true, ILV, NS>;
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      FSD_LAUNCH(k, grid, block, lds, stream, a);
    } else {
      return FSD_ERR_UNSUPPORTED;
    }
  } else {
    auto k = conv_bf16_dma_kernel<BM, BN, BK, WM, WN, false, ILV, NS>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    FSD_LAUNCH(k, grid, block, lds, stream, a);
  }
  return (int)hipGetLastError();
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// 128x64 tiles (4 x 1 waves) for layers with <= 64 output channels, 128x128 otherwise.  Measured on MI355X
// (tools/layer_bench.py, B = 64): forcing 128x64 on the wide layers to fill the last round of workgroups better (13x13
// maps: 680 tiles of 128x128 on 512 resident slots) LOSES 8-12 % -- the lower operand reuse costs more than the
// quantisation; a 3- or 4-deep DMA ring with counted vmcnt at 1 workgroup per CU loses 15-25 % against two stages at 2
// workgroups per CU (occupancy hides the DMA latency better than depth).
inline bool narrow_tile(long long pixels, int cout) {
  (void)pixels;
  return cout <= 64;
}

// ---- tile choice ---------------------------------------------------------------------------------------------------
// Tile ids: 0 = 128x128, 6 = 192x128 (4 waves, TWO workgroups per CU); 1 = 256x256, 2 = 192x256, 3 = 256x128 (8 waves,
// ONE workgroup per CU); 4 = 128x64, 5 = 128x32 (4 waves, narrow outputs).  The tiles past 0 / 4 / 5 need Cin % 64 == 0,
// Cout % BN == 0 and the bf16 NHWC store.  FSD_CONV_H_TILE forces one (tuning aid).
//
// Measured on MI355X at B = 64 (tools/layer_bench.py; forward / data gradient, TFLOP/s; round 3):
//   * two independent 4-wave workgroups per CU beat one 8-wave workgroup of the same wave tile (256x128: 811 vs 700 on
//     13x13 1024->1024): with one barrier domain per CU every wave issues its DMA pieces, drains and restarts in phase;
//     two domains run out of phase and fill each other's stalls.  A counted-vmcnt ring (4 stages of 32-element chunks,
//     3 stages at 256x128) LOSES 5-15 % against two drained stages: twice the barriers, and DMA latency was not the wait.
//   * what the large tiles buy is (a) fewer L2 -> LDS bytes per MFMA and (b) the round count of the 13x13 maps: 10816
//     pixels = 85 row tiles of 128 (680 tiles of 128x128 on 512 slots = two rounds for 1.33 rounds of work) but 57 of 192:
//     192x256 -> 228 tiles = ONE round on 256 CUs (1006 vs 811), 192x128 -> 456 of 512 slots (957).
//   * 192x128 (two workgroups per CU, wide stores) is the best or within 3 % of the best tile on every >= 128-channel
//     layer with >= 43 k pixels (26x26: 829 vs 722, 52x52: 703 vs 626, 104x104: 500 vs 406).
//   * round 5: ONE wave per SIMD on 128x128 / 96x128 wave tiles (256x256 and 192x256 on four waves; half the LDS bytes
//     per MFMA of the 8-wave layouts), the loop software-pipelined by hand (inline-assembly fragment reads with counted
//     lgkmcnt, next chunk's DMA spread over the k-steps, the barrier in front of the last step's MFMAs) LOSES 10-25 %:
//     13x13 1024->1024 876 / 771 TFLOP/s against 975 for the 8-wave 192x256, 26x26 256->512 668 / 563 against 805; the
//     same loop in the 256x256 weight gradient 640 against 679.  A DMA piece costs its wave 60-180 cycles of ISSUE
//     (MI355X_MICROARCH.md) and a lone wave has nobody to fill the matrix pipe meanwhile: 16 pieces per 64-MFMA chunk
//     ~ the 2400 cycles measured on top of the 2048 of the MFMAs.  Patch and numbers: tools/experiments_r05/pipe4_tiles.patch.
struct TileH { int bm, bn, per_cu; };
constexpr int kNumTiles = 7;
constexpr TileH kTiles[kNumTiles] = {{128, 128, 2}, {256, 256, 1}, {192, 256, 1}, {256, 128, 1},
                                     {128, 64, 2}, {128, 32, 2}, {192, 128, 2}};

inline double fill_of(const TileH& t, long long pixels, int cout) {      // used slots / slots of the rounds it takes
  const long long tiles = ((pixels + t.bm - 1) / t.bm) * ((cout + t.bn - 1) / t.bn);
  const long long slots = 256LL * t.per_cu;
  return (double)tiles / (double)(((tiles + slots - 1) / slots) * slots);
}

inline int pick_tile_h(long long pixels, int cin, int cout, int ksize, bool nchw, bool has_partial, int bk) {
  const char* env = getenv("FSD_CONV_H_TILE");                    // read per launch: tests / tuning flip it in-process
  static const char* n32_env = FSD_TUNE("FSD_CONV_H_N32");          // tuning aid: 0 disables the 32-channel tile
  if (!nchw && cout <= 32 && !has_partial && bk == 64 && !(n32_env && n32_env[0] == '0')) return 5;
  if (!nchw && narrow_tile(pixels, cout)) return 4;
  if (nchw || bk != 64 || cin % 64) return 0;
  if (env && env[0] >= '0' && env[0] <= '6' && env[0] != '4' && env[0] != '5') {
    const int t = env[0] - '0';
    return (cout % kTiles[t].bn == 0) ? t : 0;
  }
  if (cout % 128) return 0;
  // small maps with a long reduction (the 13x13 / 19x19 layers): an 8-wave tile if it fills one round of CUs well
  if (pixels <= 32768 && cout % 256 == 0 && (long long)ksize * ksize * cin >= 2304) {
    if (fill_of(kTiles[2], pixels, cout) >= 0.85) return 2;
    if (fill_of(kTiles[1], pixels, cout) >= 0.80) return 1;
  }
  return fill_of(kTiles[6], pixels, cout) >= fill_of(kTiles[0], pixels, cout) - 0.15 ? 6 : 0;
}

inline int bk_of(int cin, int ksize) {
  // k-chunk: 64 elements (128-byte rows) by default; 32 for Cin = 32 and for the short reductions of the 1x1 layers
  // (K <= 512: measured 104x104 128->64 0.072 -> 0.062 ms, 26x26 512->256 0.035 -> 0.029 ms, their data gradients
  // likewise; the 3x3 layers and the K = 1024 head lose 3-20 % with it).  FSD_CONV_H_BK=32|64 forces one (tuning aid).
  static const char* bk_env = FSD_TUNE("FSD_CONV_H_BK");
  int bk = (cin % 64 == 0 && !(ksize == 1 && cin <= 512)) ? 64 : 32;
  if (bk_env && cin % 64 == 0) bk = atoi(bk_env) == 32 ? 32 : 64;
  return bk;
}

inline int tile_bm(int tile) { return kTiles[tile].bm; }

}  // namespace

extern "C" int fsd_conv_row_tiles_h(long long pixels) { return (int)((pixels + 127) / 128); }

extern "C" int fsd_conv2d_h_plan(long long pixels, int cin, int cout, int ksize, int out_nchw_f32, int has_partial) {
  return pick_tile_h(pixels, cin, cout, ksize, out_nchw_f32 != 0, has_partial != 0, bk_of(cin, ksize));
}

extern "C" int fsd_conv2d_h_partial_rows(int batch, int height, int width, int cin, int cout, int ksize) {
  if (fsd_conv::halo_h_ok(height, width, cin, cout, ksize)) return fsd_conv::halo_h_rows(batch, height, width, cin, cout);
  const long long pixels = (long long)batch * height * width;
  const int tile = pick_tile_h(pixels, cin, cout, ksize, false, true, bk_of(cin, ksize));
  const int bm = tile_bm(tile);
  return (int)((pixels + bm - 1) / bm);
}

extern "C" int fsd_conv2d_h_partial_rows_at(int batch, int height, int width, int cin, int cout, int ksize, const void* x_bf16,
                                           long long x_ld, const void* y, long long y_ld) {
  // the rows of exactly the launch fsd_conv2d_fwd[_act]_h makes for THESE operands: the halo-staged kernel needs 16-byte
  // aligned rows on both sides, anything else takes the implicit-GEMM kernel (and its row tiles)
  if (fsd_conv::halo_h_ok(height, width, cin, cout, ksize) &&
      fsd_conv::halo_h_layout_ok(x_bf16, x_ld, y, y_ld, batch, height, width, cin, cout))
    return fsd_conv::halo_h_rows(batch, height, width, cin, cout);
  const long long pixels = (long long)batch * height * width;
  const int tile = pick_tile_h(pixels, cin, cout, ksize, false, true, bk_of(cin, ksize));
  const int bm = tile_bm(tile);
  return (int)((pixels + bm - 1) / bm);
}

extern "C" int fsd_conv2d_fwd_h(const void* x_bf16, long long x_ld, const void* w_packed_bf16, const float* bias, void* y,
                                long long y_ld, float* bn_partial, int batch, int height, int width, int cin, int cout,
                                int ksize, int out_nchw_f32, hipStream_t stream) {
  return fsd_conv2d_fwd_act_h(x_bf16, x_ld, w_packed_bf16, bias, y, y_ld, bn_partial, batch, height, width, cin, cout, ksize,
                              out_nchw_f32, 1.f, stream);
}

extern "C" int fsd_conv2d_fwd_act_h(const void* x_bf16, long long x_ld, const void* w_packed_bf16, const float* bias, void* y,
                                    long long y_ld, float* bn_partial, int batch, int height, int width, int cin, int cout,
                                    int ksize, int out_nchw_f32, float slope, hipStream_t stream) {
  (void)hipGetLastError();
  if (slope != 1.f && (out_nchw_f32 || bn_partial)) return FSD_ERR_UNSUPPORTED;
  if (!x_bf16 || !w_packed_bf16 || !y || batch < 1 || height < 1 || width < 1 || cout < 1) return FSD_ERR_ARG;
  if (ksize != 1 && ksize != 3) return FSD_ERR_UNSUPPORTED;
  if (cin % 32 || (x_ld & 7) || x_ld < cin) return FSD_ERR_UNSUPPORTED;      // 16-byte DMA pieces of 8 channels
  if (!out_nchw_f32 && ((cout & 1) || (y_ld & 1) || y_ld < cout)) return FSD_ERR_UNSUPPORTED;
  if (out_nchw_f32 && bn_partial) return FSD_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x_bf16) & 15) || (reinterpret_cast<uintptr_t>(w_packed_bf16) & 15)) return FSD_ERR_ARG;
  if (!out_nchw_f32 && (reinterpret_cast<uintptr_t>(y) & 3)) return FSD_ERR_ARG;
  const long long pixels = (long long)batch * height * width;
  // (odd strides / alignments stay on the implicit-GEMM kernel; bn_partial must hold fsd_conv2d_h_partial_rows_at() rows)
  if (!out_nchw_f32 && fsd_conv::halo_h_ok(height, width, cin, cout, ksize) &&
      fsd_conv::halo_h_layout_ok(x_bf16, x_ld, y, y_ld, batch, height, width, cin, cout))
    return fsd_conv::conv3x3_halo_h(x_bf16, x_ld, w_packed_bf16, round_up(ksize * ksize * cin, 64), bias, y, y_ld, bn_partial,
                                    batch, height, width, cin, cout, slope, stream);
  if (pixels > 0x7fffffffLL - 512 || (pixels + 1) * x_ld >= 0xffffffffLL) return FSD_ERR_UNSUPPORTED;
  ConvHArgs a;
  a.x = static_cast<const u16*>(x_bf16); a.w = static_cast<const u16*>(w_packed_bf16); a.bias = bias; a.y = y;
  a.bn_partial = bn_partial; a.x_ld = x_ld; a.y_ld = y_ld;
  a.slope = slope;
  static const char* wide_env = FSD_TUNE("FSD_CONV_H_WIDE");        // tuning aid: 0 = 4-byte stores straight from registers
  a.wide = (!out_nchw_f32 && cout % 8 == 0 && y_ld % 8 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
            !(wide_env && wide_env[0] == '0')) ? 1 : 0;            // (switched off below for the 8-wave tiles)
  a.H = height; a.W = width; a.HW = height * width; a.M = (int)pixels;
  a.Cout = cout; a.ks = ksize; a.pad = (ksize - 1) / 2;
  a.Kpad = round_up(ksize * ksize * cin, 64);                    // row stride of fsd_pack_conv_weight_bf16
  const int bk = bk_of(cin, ksize);
  a.nk = ksize * ksize * cin / bk;
  a.cpt = cin / bk;
  const bool nchw = out_nchw_f32 != 0;
  const int tile = pick_tile_h(pixels, cin, cout, ksize, nchw, bn_partial != nullptr, bk);
  // 8-wave tiles: the 96-128 KB tile would cross LDS behind one barrier for all eight waves; measured -15 % on the 13x13
  // layers they are picked for (long K: the store tail is a small share there)
  if (tile >= 1 && tile <= 3 && !(wide_env && wide_env[0] == '1')) a.wide = 0;
  const char* ilv_env = FSD_TUNE("FSD_CONV_H_ILV");                 // tuning aid: 0 / 1 forces the DMA interleave off / on
  const bool ilv = ilv_env ? ilv_env[0] == '1' : (tile >= 1 && tile <= 3) || tile == 6;
  const char* ring_env = FSD_TUNE("FSD_CONV_H_RING");               // tuning aid: 1 = counted-vmcnt ring (measured slower)
  const bool ring = ring_env && ring_env[0] == '1';
  (void)ilv; (void)ring;
  a.m_tiles = (int)((pixels + tile_bm(tile) - 1) / tile_bm(tile));
  switch (tile) {
    case 5:   // 32 output channels (data gradient of the 32 -> 64 layer at 208x208): a 64-wide tile would issue twice the MFMAs
      a.n_tiles = 1;
      return launch_conv<128, 32, 64, 4, 1>(a, false, stream);
    case 4:
      a.n_tiles = (cout + 63) / 64;
      return bk == 64 ? launch_conv<128, 64, 64, 4, 1>(a, false, stream) : launch_conv<128, 64, 32, 4, 1>(a, false, stream);
    case 1:
      a.n_tiles = cout / 256;
#ifdef FSD_EXPERIMENTS
      if (ring) { a.nk *= 2; a.cpt *= 2; return launch_conv<256, 256, 32, 2, 4, true, 4>(a, false, stream); }
      if (!ilv) return launch_conv<256, 256, 64, 2, 4>(a, false, stream);
#endif
      return launch_conv<256, 256, 64, 2, 4, true>(a, false, stream);
    case 2:
      a.n_tiles = cout / 256;
#ifdef FSD_EXPERIMENTS
      if (ring) { a.nk *= 2; a.cpt *= 2; return launch_conv<192, 256, 32, 2, 4, true, 4>(a, false, stream); }
      if (!ilv) return launch_conv<192, 256, 64, 2, 4>(a, false, stream);
#endif
      return launch_conv<192, 256, 64, 2, 4, true>(a, false, stream);
    case 3:
      a.n_tiles = cout / 128;
#ifdef FSD_EXPERIMENTS
      if (ring) return launch_conv<256, 128, 64, 4, 2, true, 3>(a, false, stream);
      if (!ilv) return launch_conv<256, 128, 64, 4, 2>(a, false, stream);
#endif
      return launch_conv<256, 128, 64, 4, 2, true>(a, false, stream);
    case 6:
      a.n_tiles = cout / 128;
#ifdef FSD_EXPERIMENTS
      if (!ilv) return launch_conv<192, 128, 64, 2, 2>(a, false, stream);
#endif
      return launch_conv<192, 128, 64, 2, 2, true>(a, false, stream);
    default:
      a.n_tiles = (cout + 127) / 128;
#ifdef FSD_EXPERIMENTS
      if (bk == 64 && ilv) return launch_conv<128, 128, 64, 2, 2, true>(a, nchw, stream);
#endif
      if (bk == 64) return launch_conv<128, 128, 64, 2, 2>(a, nchw, stream);
      return launch_conv<128, 128, 32, 2, 2>(a, nchw, stream);
  }
}

// ================= weight gradient: K = pixels, fragments through the LDS transpose read =====================
namespace {

struct WgradHArgs {
  const u16* dy;       // (pixels, dy_ld) bf16, columns [0, Cout)
  const u16* x;        // (pixels, x_ld) bf16, columns [0, Cin)
  float* ws;           // [splits][Cout][taps * Cin]
  long long dy_ld, x_ld;
  int H, W, M, Cout, Cin, ks, pad;
  int m_tiles, n_tiles, taps;
  int pix_per_split;   // multiple of 32
};

// Stage a [32 pixels][CH channels] bf16 tile as it lies in HBM.  CH*2-byte rows, written lane-linear by the DMA; the
// 16-byte piece a lane FETCHES is permuted inside its row (XOR below) so that the transpose reads that follow are
// bank-conflict free: a 32-lane half reads 4 consecutive pixel rows x 2 sixteen-channel blocks = 8 pieces of 32 B, which
// must land in 8 different 32-byte bank ranges.
template <int CH, int KC>
struct TileGeom {
  static constexpr int LPR = CH / 8;            // lanes per pixel row
  static constexpr int RPI = 64 / LPR;          // pixel rows per wave instruction
  static constexpr int PASSES = KC / (4 * RPI) > 0 ? KC / (4 * RPI) : 1;
  static constexpr int ROWS_PER_PASS = 4 * RPI; // 16 (CH = 128), 32 (CH = 64), 64 (CH = 32: only the first KC used)
  __device__ static __forceinline__ int swz(int row) {      // XOR mask on the 16-byte piece index of a row
    return CH == 128 ? ((row & 3) << 2) : CH == 64 ? ((row & 2) << 1) : 0;
  }
};

// KC = pixels per k-chunk (one barrier per chunk): 32, or 64 for the 128 x 128 tile (half the barriers per MFMA).
template <int BM, int BN, int WAVES_M, int WAVES_N, int KC>
__global__ __launch_bounds__(256) void wgrad_bf16_tr_kernel(WgradHArgs p) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(KC == 32 || KC == 64, "k-chunk of 32 or 64 pixels");
  constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile of at least 32 x 32");
  typedef TileGeom<BM, KC> GA;
  typedef TileGeom<BN, KC> GB;
  constexpr int STAGE = KC * (BM + BN);           // elements per LDS stage
  extern __shared__ __attribute__((aligned(16))) u16 smem_w[];

  // The N dimension is the packed (tap, ci) column space of dW, ncols = taps * Cin: a 128-wide tile is one tap's slice of
  // a wide layer, or several whole taps of a narrow one (Cin = 32: four taps per tile, so dy is re-read 3x instead of 9x).
  const int L = xcd_swizzle(blockIdx.x, gridDim.x);
  const int nt = L % p.n_tiles, mt = L / p.n_tiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int ncols = p.taps * p.Cin;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
  const long long pix0 = (long long)blockIdx.y * p.pix_per_split;
  long long pix_end = pix0 + p.pix_per_split;
  if (pix_end > p.M) pix_end = p.M;
  const int nk = (int)((pix_end - pix0 + KC - 1) / KC);

  // this lane's piece of an A row: channels m0 + a_piece*8 .. +7 (logical), rows a_row + ROWS_PER_PASS * pass
  const int a_row = wave * GA::RPI + lane / GA::LPR, a_pp = lane % GA::LPR;
  const int b_row = wave * GB::RPI + lane / GB::LPR, b_pp = lane % GB::LPR;
  // the B piece of this lane is the same in every pass (the swizzle depends on row & 3 and passes advance by >= 16 rows):
  // its tap and channel offset are lane constants
  const int b_piece = b_pp ^ GB::swz(b_row);
  const int b_col = n0 + b_piece * 8;                      // column of dW = tap * Cin + ci
  const int b_tap = b_col / p.Cin, b_ci = b_col - b_tap * p.Cin;
  const bool b_col_ok = b_col < ncols;
  const int ky = b_tap / p.ks, kx = b_tap - ky * p.ks;
  const int dyo = ky - p.pad, dxo = kx - p.pad;
  const int shift = dyo * p.W + dxo;
  // image coordinates of the B rows of this thread (they advance by 32 pixels per chunk)
  int b_y[GB::PASSES], b_x[GB::PASSES];
  const float inv_w = 1.0f / (float)p.W, inv_h = 1.0f / (float)p.H;
#pragma unroll
  for (int j = 0; j < GB::PASSES; ++j) {
    const long long pp = pix0 + b_row + GB::ROWS_PER_PASS * j;
    const long long q = pp / p.W;
    b_x[j] = (int)(pp - q * p.W);
    b_y[j] = (int)(q % p.H);
  }

  // Staging addresses are carried from chunk to chunk (a chunk = the next KC pixel rows: one 64-bit add per piece) and the
  // padding select is branch-free.  (Round 2 recomputed pix * ld with 64-bit multiplies under divergent `ok` branches for
  // every piece: ~300 scalar / vector instructions per 16 MFMAs, PMC: 45 % of wave time parked, 0.25 MFMA-busy.)
  const bool a_ch_ok = (m0 + (a_pp ^ GA::swz(a_row)) * 8) < p.Cout;       // the swizzle is the same in every pass
  const u16* a_ptr[GA::PASSES];
  const u16* b_ptr[GB::PASSES];
#pragma unroll
  for (int j = 0; j < GA::PASSES; ++j)
    a_ptr[j] = p.dy + (pix0 + a_row + GA::ROWS_PER_PASS * j) * p.dy_ld + m0 + (a_pp ^ GA::swz(a_row)) * 8;
#pragma unroll
  for (int j = 0; j < GB::PASSES; ++j)
    b_ptr[j] = p.x + (pix0 + b_row + GB::ROWS_PER_PASS * j + shift) * p.x_ld + b_ci;
  const long long a_step = (long long)KC * p.dy_ld, b_step = (long long)KC * p.x_ld;
  int a_left = (int)(pix_end - pix0) - a_row, b_left = (int)(pix_end - pix0) - b_row;     // rows left below this lane's row
  const u16* zero_src = g_zero_page_h;
  asm volatile("" : "+v"(zero_src));      // opaque: no s_getpc + s_load (and its lgkmcnt(0)) at every use, see wgrad_bf16_tr8_kernel

  auto stage = [&](int kc, u16* st) {
    (void)kc;
#pragma unroll
    for (int j = 0; j < GA::PASSES; ++j) {
      if (GA::ROWS_PER_PASS * j + wave * GA::RPI < KC) {            // CH = 32: 64 rows per pass, only KC exist
        const bool ok = a_ch_ok && (a_row + GA::ROWS_PER_PASS * j < KC) && a_left > GA::ROWS_PER_PASS * j;
        dma16(ok ? a_ptr[j] : zero_src, st + (GA::ROWS_PER_PASS * j + wave * GA::RPI) * BM);
      }
      a_ptr[j] += a_step;
    }
    a_left -= KC;
#pragma unroll
    for (int j = 0; j < GB::PASSES; ++j) {
      if (GB::ROWS_PER_PASS * j + wave * GB::RPI < KC) {
        const bool ok = b_col_ok && (b_row + GB::ROWS_PER_PASS * j < KC) && b_left > GB::ROWS_PER_PASS * j &&
                        (unsigned)(b_y[j] + dyo) < (unsigned)p.H && (unsigned)(b_x[j] + dxo) < (unsigned)p.W;
        dma16(ok ? b_ptr[j] : zero_src, st + KC * BM + (GB::ROWS_PER_PASS * j + wave * GB::RPI) * BN);
      }
      b_ptr[j] += b_step;
      // advance this row's image coordinates by one chunk (KC pixels).  (v + 0.5) / n is never within 1/(2n) of an
      // integer, so the float quotients are exact; the row index wraps by the same rule (a chunk spans up to KC image
      // rows on the 3x3 / 2x2 / 1x1 maps at the end of the reweighting net, so a fixed number of subtractions is not
      // enough -- the version with three conditional subtractions mis-tracked every map smaller than 4x4).
      int xx = b_x[j] + KC;
      const int q = (int)(((float)xx + 0.5f) * inv_w);
      xx -= q * p.W;
      int yy = b_y[j] + q;
      yy -= (int)(((float)yy + 0.5f) * inv_h) * p.H;
      b_x[j] = xx;
      b_y[j] = yy;
    }
    b_left -= KC;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transpose-read geometry: 16-lane group G = lane >> 4 reads pixel rows kb + (L >> 2), channels cb + 4 * (L & 3)
  const int G = lane >> 4, Lq = lane & 15;
  auto frag = [&](const u16* tile, int CH, int ch0, int krow, int swzmask) -> bf16x8 {
    const int row = krow + (Lq >> 2);
    const int ch = ch0 + 16 * (G & 1) + 4 * (Lq & 3);
    const int piece = (ch >> 3) ^ swzmask;
    const u16* a = tile + row * CH + piece * 8 + (ch & 7);
    // the bf16-typed form of the builtin + one shufflevector: rebuilding the 8-element operand element by element from
    // the v4i16 form (bit_cast short -> __bf16 per element) miscompiles on ROCm 7.2 (every element becomes element 0)
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a + 4 * CH));
    const bf16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return v;
  };
  auto compute = [&](const u16* st) {
    const u16* sa = st;
    const u16* sb = st + KC * BM;
#pragma unroll
    for (int s = 0; s < KC / 16; ++s) {
      const int krow = s * 16 + (G >> 1) * 8;
      const int row_lo = krow + (Lq >> 2);                        // the +4 rows of the second read share (row & 3)
      bf16x8 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = frag(sa, BM, (wm * TM + i) * 32, krow, GA::swz(row_lo));
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = frag(sb, BN, (wn * TN + j) * 32, krow, GB::swz(row_lo));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };

  // (The compiler puts `s_waitcnt vmcnt(0)` in front of the first transpose read that follows a global_load_lds in program
  // order -- it cannot prove the two do not alias.  Issuing every fragment read of the chunk BEFORE the next chunk's DMA
  // removes that wait and was measured: no change, +3 % time on the 32-pixel-chunk variants (64 fragment registers cost
  // occupancy).  The kernel is bound by operand delivery -- 512 bytes of L2 -> LDS DMA per MFMA at this tile size, i.e.
  // 64 B/clk/CU at the full matrix rate -- not by that wait.)
  if (nk > 0) {
    stage(0, smem_w);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int kc = 0; kc < nk; ++kc) {
      if (kc + 1 < nk) stage(kc + 1, smem_w + (cur ^ 1) * STAGE);
      compute(smem_w + cur * STAGE);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }
  }
  float* out = p.ws + (long long)blockIdx.y * p.Cout * ncols;
  const int c_lane = lane & 31, r_lane = 4 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + c_lane;          // column of dW = tap * Cin + ci
    if (n >= ncols) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + r_lane;
        if (m < p.Cout) out[(long long)m * ncols + n] = acc[i][j][r];
      }
  }
}

// Folding the split-K workspace, fixed summation order throughout.
// (1) many splits (the layers with few tiles): slices are first summed in groups of kFoldGroup by a 2-D grid -- a
//     single pass would leave each thread with hundreds of dependent-latency loads and a handful of workgroups;
// (2) ws[split][cout][tap * cin + ci] -> dW[cout][cin][taps] (OIHW): one workgroup per (cout, block of CB input channels)
//     reads TAPS runs of CB floats per slice and writes ONE contiguous run of CB * TAPS floats (the tap-major ->
//     tap-minor transposition goes through LDS instead of 4-byte stores at a 36-byte stride).
constexpr int kFoldGroup = 16, kFoldDirect = 24;

__global__ __launch_bounds__(256) void wgrad_h_partial_kernel(const float* __restrict__ ws, float* __restrict__ out, int splits,
                                                              long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int k0 = blockIdx.y * kFoldGroup, k1 = min(splits, k0 + kFoldGroup);
  // loads four at a time, sums in slice order (the plain loop issued one dependent-latency load per trip)
  float s = 0.f;
  int k = k0;
  for (; k + 4 <= k1; k += 4) {
    const float v0 = ws[(long long)k * total + idx], v1 = ws[(long long)(k + 1) * total + idx];
    const float v2 = ws[(long long)(k + 2) * total + idx], v3 = ws[(long long)(k + 3) * total + idx];
    s += v0; s += v1; s += v2; s += v3;
  }
  for (; k < k1; ++k) s += ws[(long long)k * total + idx];
  out[(long long)blockIdx.y * total + idx] = s;
}

template <int TAPS, int CB>
__global__ __launch_bounds__(256) void wgrad_h_fold_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits,
                                                           int cout, int cin) {
  __shared__ float stage[CB * TAPS];
  const int cblocks = (cin + CB - 1) / CB;
  const int co = blockIdx.x / cblocks, c0 = (blockIdx.x - co * cblocks) * CB;
  const int nci = min(CB, cin - c0);
  const long long total = (long long)cout * TAPS * cin;
  const float* base = ws + (long long)co * TAPS * cin + c0;
  for (int e = threadIdx.x; e < CB * TAPS; e += 256) {
    const int tap = e / CB, c = e - tap * CB;
    if (c < nci) {
      const float* p = base + (long long)tap * cin + c;
      float s = 0.f;
      int k = 0;
      for (; k + 4 <= splits; k += 4) {       // four loads in flight, summed in slice order
        const float v0 = p[(long long)k * total], v1 = p[(long long)(k + 1) * total];
        const float v2 = p[(long long)(k + 2) * total], v3 = p[(long long)(k + 3) * total];
        s += v0; s += v1; s += v2; s += v3;
      }
      for (; k < splits; ++k) s += p[(long long)k * total];
      stage[c * TAPS + tap] = s;
    }
  }
  __syncthreads();
  float* o = dw + ((long long)co * cin + c0) * TAPS;
  for (int e = threadIdx.x; e < nci * TAPS; e += 256) o[e] = stage[e];
}

// floats of workspace the fold needs BEHIND the `splits` slices the GEMM writes
inline long long wgrad_h_fold_extra_slices(int splits) { return splits > kFoldDirect ? (splits + kFoldGroup - 1) / kFoldGroup : 0; }

int launch_wgrad_h_fold(float* ws, float* dw, int splits, int cout, int cin, int taps, hipStream_t stream) {
  const long long total = (long long)cout * taps * cin;
  float* src = ws;
  float* dst = ws + (long long)splits * total;          // ping: behind the slices; pong: over the (consumed) first slices
  while (splits > kFoldDirect) {
    const int groups = (splits + kFoldGroup - 1) / kFoldGroup;
    FSD_LAUNCH(wgrad_h_partial_kernel, dim3((unsigned)((total + 255) / 256), groups), dim3(256), 0, stream, src, dst,
                       splits, total);
    float* t = src; src = dst; dst = t;
    splits = groups;
  }
  if (taps == 9) {
    const unsigned wgs = (unsigned)cout * ((cin + 63) / 64);
    FSD_LAUNCH((wgrad_h_fold_kernel<9, 64>), dim3(wgs), dim3(256), 0, stream, src, dw, splits, cout, cin);
  } else {
    const unsigned wgs = (unsigned)cout * ((cin + 255) / 256);
    FSD_LAUNCH((wgrad_h_fold_kernel<1, 256>), dim3(wgs), dim3(256), 0, stream, src, dw, splits, cout, cin);
  }
  return (int)hipGetLastError();
}

// 8-wave 256 x 256 tile of the weight gradient (round 3).  The 128 x 128 kernel above moves 512 bytes of L2 -> LDS DMA per
// MFMA -- 64 B/clk/CU at the full matrix rate, more than the L2 delivers -- and sits at 0.25 MFMA-busy whatever the issue
// order; this tile halves that.  Both operand tiles are kept as 128-channel SUB-tiles ([KC pixels][128 channels], the
// geometry and swizzle of TileGeom<128, KC>, verified by the hardware probes): 2 of dy + 2 of x per stage.  2 x 4 waves of
// 128 x 64 (TM = 4, TN = 2: 128 accumulator registers), one workgroup per CU.  Measured at B = 64 (tools/layer_bench.py
// wgrad, incl. the fold): 26x26 256->512 0.180 -> 0.153 ms, 52x52 128->256 0.196 -> 0.172, 13x13 512->1024 0.182 -> 0.153,
// 1024->1024 0.322 -> 0.313, 1280->1024 0.383 -> 0.366.  The same kernel as 256 x 128 on 4 waves (SA = 2, SB = 1, two
// workgroups per CU, 32-pixel chunks) is no faster than 128 x 128 (0.170 / 0.180 / 0.174 / 0.347 / 0.415 on those layers).
template <int KC, int SA = 2, int SB = 2, int WM = 2, int WN = 4>
__global__ __launch_bounds__(64 * WM * WN) void wgrad_bf16_tr8_kernel(WgradHArgs p) {
  static_assert(KC == 32 || KC == 64, "k-chunk of 32 or 64 pixels");
  constexpr int BM = 128 * SA, BN = 128 * SB, TM = BM / WM / 32, TN = BN / WN / 32, NT = 64 * WM * WN;
  static_assert(TM * 32 * WM == BM && TN * 32 * WN == BN, "whole 32 x 32 accumulators per wave");
  typedef TileGeom<128, KC> GT;
  constexpr int RPASS = 4 * WM * WN;                // pixel rows staged per pass of the workgroup (4 per wave instruction)
  constexpr int PASSES = KC / RPASS;
  static_assert(PASSES >= 1 && PASSES * RPASS == KC, "whole staging passes");
  constexpr int SUB = KC * 128;                     // elements of one sub-tile
  constexpr int STAGE = (SA + SB) * SUB;
  extern __shared__ __attribute__((aligned(16))) u16 smem_w8[];

  const int L = xcd_swizzle(blockIdx.x, gridDim.x);
  const int nt = L % p.n_tiles, mt = L / p.n_tiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int ncols = p.taps * p.Cin;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const long long pix0 = (long long)blockIdx.y * p.pix_per_split;
  long long pix_end = pix0 + p.pix_per_split;
  if (pix_end > p.M) pix_end = p.M;
  const int nk = (int)((pix_end - pix0 + KC - 1) / KC);

  const int row0 = wave * GT::RPI + lane / GT::LPR, pp = lane % GT::LPR;     // this lane's row (of a pass) and 16-byte slot
  const int piece = pp ^ GT::swz(row0);                                       // the piece it fetches (same in every pass)
  bool a_ok[SA], b_ok[SB];
  const u16* a_ptr[SA][PASSES];
  const u16* b_ptr[SB][PASSES];
  int b_dy[SB], b_dx[SB];
#pragma unroll
  for (int t = 0; t < SA; ++t) {
    const int ch = m0 + t * 128 + piece * 8;
    a_ok[t] = ch < p.Cout;
#pragma unroll
    for (int j = 0; j < PASSES; ++j) a_ptr[t][j] = p.dy + (pix0 + row0 + RPASS * j) * p.dy_ld + ch;
  }
#pragma unroll
  for (int t = 0; t < SB; ++t) {
    const int col = n0 + t * 128 + piece * 8;                                 // column of dW = tap * Cin + ci
    b_ok[t] = col < ncols;
    const int tap = col / p.Cin, ci = col - tap * p.Cin;
    const int ky = tap / p.ks, kx = tap - ky * p.ks;
    b_dy[t] = ky - p.pad;
    b_dx[t] = kx - p.pad;
#pragma unroll
    for (int j = 0; j < PASSES; ++j)
      b_ptr[t][j] = p.x + (pix0 + row0 + RPASS * j + b_dy[t] * p.W + b_dx[t]) * p.x_ld + ci;
  }
  int b_y[PASSES], b_x[PASSES];
  const float inv_w = 1.0f / (float)p.W, inv_h = 1.0f / (float)p.H;
#pragma unroll
  for (int j = 0; j < PASSES; ++j) {
    const long long q0 = pix0 + row0 + RPASS * j;
    const long long q = q0 / p.W;
    b_x[j] = (int)(q0 - q * p.W);
    b_y[j] = (int)(q % p.H);
  }
  const long long a_step = (long long)KC * p.dy_ld, b_step = (long long)KC * p.x_ld;
  int left = (int)(pix_end - pix0) - row0;                                    // pixel rows left below this lane's row
  const u16* zero_src = g_zero_page_h;
  // opaque from here on: the compiler otherwise re-derives the address (s_getpc + s_load) at every use, and the scalar
  // load's lgkmcnt(0) drains the transpose reads in flight
  asm volatile("" : "+v"(zero_src));

  auto stage = [&](u16* st) {
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
      const bool row_ok = left > RPASS * j;
      u16* dst = st + (RPASS * j + wave * GT::RPI) * 128;
#pragma unroll
      for (int t = 0; t < SA; ++t) {
        dma16(row_ok && a_ok[t] ? a_ptr[t][j] : zero_src, dst + t * SUB);
        a_ptr[t][j] += a_step;
      }
#pragma unroll
      for (int t = 0; t < SB; ++t) {
        const bool ok = row_ok && b_ok[t] && (unsigned)(b_y[j] + b_dy[t]) < (unsigned)p.H &&
                        (unsigned)(b_x[j] + b_dx[t]) < (unsigned)p.W;
        dma16(ok ? b_ptr[t][j] : zero_src, dst + (SA + t) * SUB);
        b_ptr[t][j] += b_step;
      }
      // advance the row's image coordinates by one chunk (exact float quotients, see wgrad_bf16_tr_kernel)
      int xx = b_x[j] + KC;
      const int q = (int)(((float)xx + 0.5f) * inv_w);
      xx -= q * p.W;
      int yy = b_y[j] + q;
      yy -= (int)(((float)yy + 0.5f) * inv_h) * p.H;
      b_x[j] = xx;
      b_y[j] = yy;
    }
    left -= KC;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int G = lane >> 4, Lq = lane & 15;
  auto frag = [&](const u16* tile, int ch0, int krow, int swzmask) -> bf16x8 {
    const int row = krow + (Lq >> 2);
    const int ch = ch0 + 16 * (G & 1) + 4 * (Lq & 3);
    const int pc = (ch >> 3) ^ swzmask;
    const u16* a = tile + row * 128 + pc * 8 + (ch & 7);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a + 4 * 128));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto compute = [&](const u16* st) {
#pragma unroll
    for (int s = 0; s < KC / 16; ++s) {
      const int krow = s * 16 + (G >> 1) * 8;
      const int sw = GT::swz(krow + (Lq >> 2));
      bf16x8 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int cb = (wm * TM + i) * 32;                                    // channel block inside the 256-row tile
        af[i] = frag(st + (cb >> 7) * SUB, cb & 127, krow, sw);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int cb = (wn * TN + j) * 32;
        bf[j] = frag(st + (SA + (cb >> 7)) * SUB, cb & 127, krow, sw);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };

  if (nk > 0) {
    stage(smem_w8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int kc = 0; kc < nk; ++kc) {
      if (kc + 1 < nk) stage(smem_w8 + (cur ^ 1) * STAGE);
      compute(smem_w8 + cur * STAGE);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }
  }
  float* out = p.ws + (long long)blockIdx.y * p.Cout * ncols;
  const int c_lane = lane & 31, r_lane = 4 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + c_lane;
    if (n >= ncols) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + r_lane;
        if (m < p.Cout) out[(long long)m * ncols + n] = acc[i][j][r];
      }
  }
}

inline int wgrad_h_kc(long long pixels);
// workgroups of the weight-gradient kernel resident on the chip at a time, for the tile wgrad_h_tiles picks
inline int wgrad_h_resident(int bm, int bn, long long pixels) { return (bm == 256 || (bm == 128 && bn == 128 && wgrad_h_kc(pixels) == 64)) ? 512 : 1536; }

inline int wgrad_h_splits(long long pixels, int tiles, int resident) {
  const long long max_s = (pixels + 1023) / 1024;     // at least 32 chunks per split
  int s;
  static const char* env = FSD_TUNE("FSD_WGRAD_H_SPLITS");          // tuning aid: force the split count of the many-tile case
  if (tiles >= 256 && env && atoi(env) > 0) {
    s = atoi(env);
  } else if (tiles >= 256) {
    // many tiles: few splits, chosen so that the last round of the 512 resident workgroups (2 per CU) is well filled;
    // every split is one more workspace slice for the fold kernel to read.  The weight per split is fitted to the sweep of
    // round 2 (tools/layer_bench.py wgrad, bf16, FSD_WGRAD_H_SPLITS): with 0.03 the 576-tile layers (1024 -> 1024) took 6
    // splits and 0.423 ms; 0.06 gives them 4 (0.340 ms) and leaves 288 tiles at 5 (0.190 ms) and 720 tiles at 2 (0.41 ms).
    double best = 1e30;
    s = 1;
    for (int c = 1; c <= 8; ++c) {
      const double wgs = (double)tiles * c, rounds = (double)((long long)((wgs + 511) / 512));
      const double cost = rounds / (wgs / 512.0) * (1.0 + 0.06 * c);
      if (cost < best - 1e-9) { best = cost; s = c; }
    }
  } else {
    // few tiles: exactly ONE round of co-resident workgroups (`resident`: 512 for the 64-pixel-chunk 128x128 kernel with its
    // 64 KB of LDS, 1536 for the others), rounded DOWN so that no handful of workgroups spills into a second round.
    // Sweep of round 2 (tools/layer_bench.py wgrad, bf16, FSD_WGRAD_H_TARGET = 512 / 1024 / 1536, repeatable to 1 %):
    //   104x104 64->128 0.229 / 0.277 / 0.303 ms    52x52 128->256 0.192 / 0.211 / 0.219 ms       (64-pixel chunks)
    //   26x26 256->512  0.210 / 0.233 / 0.193 ms    208x208 32->64 0.481 / 0.325 / 0.311 ms       (32-pixel chunks)
    static const char* env_t = FSD_TUNE("FSD_WGRAD_H_TARGET");      // tuning aid
    const int target = env_t && atoi(env_t) > 0 ? atoi(env_t) : resident;
    s = target / tiles;
  }
  if (s > max_s) s = (int)max_s;
  return s < 1 ? 1 : s > 1024 ? 1024 : s;
}

// split count for the one-workgroup-per-CU 256 x 256 tile.  Many tiles: whole rounds of 256, a small price per workspace
// slice.  Few tiles: one round, as long as a split keeps 32 chunks of 64 pixels.
inline int wgrad_h_big_splits(long long pixels, int tiles) {
  const long long max_s = (pixels + 2047) / 2048;
  int s = 1;
  if (tiles >= 128) {
    double best = 1e30;
    for (int c = 1; c <= 16; ++c) {
      const double wgs = (double)tiles * c, rounds = (double)((long long)((wgs + 255) / 256));
      const double cost = rounds / (wgs / 256.0) * (1.0 + 0.03 * c);
      if (cost < best - 1e-9) { best = cost; s = c; }
    }
  } else {
    s = 256 / tiles;
  }
  if (s > max_s) s = (int)max_s;
  return s < 1 ? 1 : s;
}

// the 8-wave 256 x 256 tile: from 256 x 256 of dW, when tiles x splits fill at least three quarters of the CUs (the 1x1
// layers and short maps do not: 26x26 512->256 1x1 has 2 tiles and 21 splits' worth of pixels).  FSD_WGRAD_H_BIG=0
// switches it off (tuning aid).
inline bool wgrad_h_big(int cout, int ncols, long long pixels) {
  const char* env = FSD_TUNE("FSD_WGRAD_H_BIG");
  if ((env && env[0] == '0') || cout < 256 || ncols < 256) return false;
  const int tiles = ((cout + 255) / 256) * ((ncols + 255) / 256);
  return (long long)tiles * wgrad_h_big_splits(pixels, tiles) >= 192;
}

inline void wgrad_h_tiles(int cout, int ncols, long long pixels, int* bm, int* bn) {
  if (wgrad_h_big(cout, ncols, pixels)) { *bm = 256; *bn = 256; return; }
  *bm = cout <= 64 ? 64 : 128;
  *bn = ncols <= 32 ? 32 : ncols <= 64 ? 64 : 128;     // ncols = taps * Cin: the packed column space of dW
  if (*bn == 32) *bm = 128;                            // the 32-wide variant runs 4 x 1 waves of 32 rows
}

template <int BM, int BN, int WM, int WN, int KC = 32>
int launch_wgrad_h(const WgradHArgs& a, int splits, hipStream_t stream) {
  const size_t lds = 2 * (size_t)KC * (BM + BN) * sizeof(u16);
  fsd_prof::Scope prof(fsd_prof::kGemmBf16, 2.0 * (double)a.M * a.Cout * ((double)a.taps * a.Cin), stream);
  auto k = wgrad_bf16_tr_kernel<BM, BN, WM, WN, KC>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  FSD_LAUNCH(k, dim3(a.m_tiles * a.n_tiles, splits), dim3(256), lds, stream, a);
  return (int)hipGetLastError();
}

// pixels per k-chunk of the 128 x 128 tile: 64 (two LDS stages of 32 KB, still two workgroups per CU, half the barriers)
// on the large maps.  Measured at B = 64 (tools/layer_bench.py wgrad, bf16): 104x104 64->128 0.331 -> 0.280 ms, 52x52
// 128->256 0.251 -> 0.227 ms; 13x13 (11 k pixels, few chunks per split) loses 4-7 %, so it keeps 32.
// FSD_WGRAD_H_KC=32|64 forces one (tuning aid).
inline int wgrad_h_kc(long long pixels) {
  static const char* env = FSD_TUNE("FSD_WGRAD_H_KC");
  if (env) return atoi(env) == 32 ? 32 : 64;
  return pixels >= 32768 ? 64 : 32;       // with the one-round split rule 26x26 (43 k pixels) gains too: 0.207 -> 0.189 ms
}

}  // namespace

extern "C" size_t fsd_conv2d_wgrad_h_workspace_bytes(int batch, int height, int width, int cin, int cout, int ksize) {
  if (fsd_conv::wgrad_halo_h_ok(height, width, cin, cout, ksize)) {
    const int slots = fsd_conv::wgrad_halo_h_slots(batch, height, width, cin);
    return (size_t)(slots + wgrad_h_fold_extra_slices(slots)) * cout * 9 * cin * sizeof(float);
  }
  int bm, bn;
  const int ncols = ksize * ksize * cin;
  wgrad_h_tiles(cout, ncols, (long long)batch * height * width, &bm, &bn);
  const int tiles = ((cout + bm - 1) / bm) * ((ncols + bn - 1) / bn);
  const int splits = (bm == 256 && bn == 256) ? wgrad_h_big_splits((long long)batch * height * width, tiles)
                               : wgrad_h_splits((long long)batch * height * width, tiles, wgrad_h_resident(bm, bn, (long long)batch * height * width));
  return (size_t)(splits + wgrad_h_fold_extra_slices(splits)) * cout * ksize * ksize * cin * sizeof(float);
}

extern "C" int fsd_conv2d_wgrad_h_plan(long long pixels, int cin, int cout, int ksize) {
  // (the halo-staged kernel of the 32 -> 64 / 64 -> 128 layers also depends on the image shape: H % 8 == 0, W % 16 == 0 --
  // this query answers for the GEMM kernels behind it)
  int bm, bn;
  wgrad_h_tiles(cout, ksize * ksize * cin, pixels, &bm, &bn);
  return bm * 1000 + bn;
}

extern "C" int fsd_conv2d_wgrad_h(const void* dy_bf16, long long dy_ld, const void* x_bf16, long long x_ld, float* dw_oihw,
                                  void* workspace, size_t workspace_bytes, int batch, int height, int width, int cin,
                                  int cout, int ksize, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dy_bf16 || !x_bf16 || !dw_oihw || !workspace || batch < 1 || height < 1 || width < 1 || cin < 1 || cout < 1)
    return FSD_ERR_ARG;
  if (ksize != 1 && ksize != 3) return FSD_ERR_UNSUPPORTED;
  if ((cin & 7) || (cout & 7) || (dy_ld & 7) || (x_ld & 7) || dy_ld < cout || x_ld < cin) return FSD_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(dy_bf16) & 15) || (reinterpret_cast<uintptr_t>(x_bf16) & 15)) return FSD_ERR_ARG;
  const long long pixels = (long long)batch * height * width;
  if (pixels > 0x7fffffffLL - 4096) return FSD_ERR_UNSUPPORTED;
  if (workspace_bytes < fsd_conv2d_wgrad_h_workspace_bytes(batch, height, width, cin, cout, ksize)) return FSD_ERR_WORKSPACE;
  if (fsd_conv::wgrad_halo_h_ok(height, width, cin, cout, ksize)) {
    // narrow layers: halo-staged blocks, one partial per persistent workgroup (wgrad_halo_h.hip)
    const int rc = fsd_conv::wgrad3x3_halo_h(dy_bf16, dy_ld, x_bf16, x_ld, static_cast<float*>(workspace), batch, height, width,
                                             cin, cout, stream);
    if (rc != 0) return rc;
    return launch_wgrad_h_fold(static_cast<float*>(workspace), dw_oihw, fsd_conv::wgrad_halo_h_slots(batch, height, width, cin),
                               cout, cin, 9, stream);
  }
  WgradHArgs a;
  a.dy = static_cast<const u16*>(dy_bf16); a.x = static_cast<const u16*>(x_bf16); a.ws = static_cast<float*>(workspace);
  a.dy_ld = dy_ld; a.x_ld = x_ld; a.H = height; a.W = width; a.M = (int)pixels; a.Cout = cout; a.Cin = cin;
  a.ks = ksize; a.pad = (ksize - 1) / 2; a.taps = ksize * ksize;
  int bm, bn;
  const int ncols = a.taps * cin;
  wgrad_h_tiles(cout, ncols, pixels, &bm, &bn);
  a.m_tiles = (cout + bm - 1) / bm;
  a.n_tiles = (ncols + bn - 1) / bn;
  const int splits = (bm == 256 && bn == 256) ? wgrad_h_big_splits(pixels, a.m_tiles * a.n_tiles)
                               : wgrad_h_splits(pixels, a.m_tiles * a.n_tiles, wgrad_h_resident(bm, bn, pixels));
  const bool kc64 = (bm == 256 && bn == 256) || (bm == 128 && bn == 128 && wgrad_h_kc(pixels) == 64);
  a.pix_per_split = round_up((int)((pixels + splits - 1) / splits), kc64 ? 64 : 32);
  int rc;
  if (bm == 256 && bn == 256) {
    const size_t lds = 2 * (size_t)64 * (256 + 256) * sizeof(u16);
    fsd_prof::Scope prof(fsd_prof::kGemmBf16, 2.0 * (double)a.M * a.Cout * ((double)a.taps * a.Cin), stream);
    auto k = wgrad_bf16_tr8_kernel<64>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    FSD_LAUNCH(k, dim3(a.m_tiles * a.n_tiles, splits), dim3(512), lds, stream, a);
    rc = (int)hipGetLastError();
  }
  else if (bn == 32) rc = launch_wgrad_h<128, 32, 4, 1>(a, splits, stream);
  else if (kc64) rc = launch_wgrad_h<128, 128, 2, 2, 64>(a, splits, stream);
  else if (bm == 128 && bn == 128) rc = launch_wgrad_h<128, 128, 2, 2>(a, splits, stream);
  else if (bm == 128 && bn == 64) rc = launch_wgrad_h<128, 64, 2, 2>(a, splits, stream);
  else if (bm == 64 && bn == 128) rc = launch_wgrad_h<64, 128, 2, 2>(a, splits, stream);
  else rc = launch_wgrad_h<64, 64, 2, 2>(a, splits, stream);
  if (rc != 0) return rc;
  return launch_wgrad_h_fold(static_cast<float*>(workspace), dw_oihw, splits, cout, cin, a.taps, stream);
}
