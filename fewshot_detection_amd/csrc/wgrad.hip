// Weight gradient of the 3x3 / 1x1 convolutions on the fp32 matrix cores.
//
//   dW[co][tap][ci] = sum over pixels p of  dy[p][co] * x[p + tap][ci]
//
// i.e. a GEMM with M = Cout, N = taps*Cin and the reduction over K = B*H*W pixels.  Both operands
// are "reduction-strided" in memory (NHWC: the channel index is contiguous, the pixel index is the
// row), so tiles are staged in LDS as [k][m] / [k][n] exactly as they lie in HBM (16-byte loads and
// stores along the channel axis) and MFMA fragments are gathered with ds_read_b32 (32 consecutive
// floats per half-wave: conflict-free).  The pixel range is split across workgroups (split-K) to
// fill the chip; each split writes its own slice of a workspace and a small second kernel sums the
// slices in a fixed order (deterministic) while scattering into the OIHW layout of the parameter.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "fsdet.h"
#include "conv_common.hpp"
#include "profile.hpp"
#include "ew_types.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

constexpr int kBK = 32;       // pixels per k-chunk
constexpr int kThreads = 256;

struct WgradArgs {
  const float* dy;     // (pixels, dy_ld), columns [0, Cout) (zero padded to a multiple of 4)
  const float* x;      // (pixels, x_ld) NHWC activations that fed the convolution
  float* ws;           // [splits][Cout][ncols]
  long long dy_ld, x_ld;
  int H, W, HW, M;     // M = pixels
  int Cout, cin4, ks, pad, ncols;
  int m_tiles, n_tiles, pix_per_split;
  long long dy_bs, x_bs, ws_bs;   // batched (gridDim.z > 1, Winograd weight gradient): strides between batches
  // activation on load (fsd_conv2d_wgrad_ex, ks == 1): x holds the raw output of the producing convolution and the operand is
  // leaky(x * x_scale[ci] + x_shift[ci]); null = x as it is
  const float* x_scale;
  const float* x_shift;
  float x_slope;
};

// TILE x TILE outputs per workgroup (2x2 waves), STAGES LDS buffers.
//   Exact fp32 MFMA (32x32x2), LDS tiles hold floats; default TILE 64, 1 stage (17 KB, 8 WGs/CU).  BF16 must be false:
//   the bf16 storage mode has its own weight-gradient kernel (conv_bf16v2.hip, LDS transpose reads).
//   DMA = true  : (fp32, PLAIN, 2 stages) the [row][channel] tiles go global -> LDS with global_load_lds: no staging
//                 registers, no ds_write, no address VALU in the loop; LDS rows are unpadded (the DMA writes 1 KB per
//                 wave instruction linearly), which is conflict-free for the 32-lane b32 fragment reads.  Needs full
//                 tiles and full 32-row chunks: the launcher sends column remainders / the last < 32 rows elsewhere.
//   SPLIT = true: the same reduction on the bf16 matrix cores at fp32 accuracy (conv.hip, conv_gemm_kernel SPLIT): every
//                 staged fp32 value is split into three bfloat16 planes on its way into LDS ([pixel][channel] bf16 tiles,
//                 one per plane, 16-byte pieces XOR-permuted inside a row), the k-contiguous MFMA fragments come out
//                 through the transposing read ds_read_b64_tr_b16, and a product is six MFMA terms.  One LDS stage.
template <int TILE, int STAGES, bool BF16, bool PLAIN = false, bool DMA = false, bool SPLIT = false>
// (at least 3 waves per SIMD: the split 128x128 variant then keeps its accumulators in VGPRs, 154 registers instead of 130 + 64
// AGPRs = two waves per SIMD; its 48 KB of LDS allow three workgroups per CU: 26x26 256->512 0.205 -> 0.197 ms, 13x13 0.474 -> 0.456)
__global__ __launch_bounds__(kThreads, SPLIT ? 3 : 1) void wgrad_kernel(WgradArgs p) {
  static_assert(!BF16, "fp32 kernel");
  static_assert(!DMA || (PLAIN && STAGES == 2), "DMA staging: fp32 plain GEMM, two LDS stages");
  static_assert(!SPLIT || (!DMA && STAGES == 1), "split operands: register staging, one LDS stage");
  typedef float lds_t;
  constexpr int LDT = DMA ? TILE : TILE + 4;   // LDS row stride in elements (keeps 16-byte alignment)
  constexpr int CQ = TILE / 4;                     // float4 column groups per tile row
  constexpr int RPP = kThreads / CQ;               // pixel rows staged per pass
  constexpr int PASSES = kBK / RPP;
  constexpr int T = TILE / 64;                     // 32x32 MFMA tiles per wave along each dimension
  constexpr int STAGE = 2 * kBK * LDT;             // A tile + B tile, in elements
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  lds_t* smem = reinterpret_cast<lds_t*>(smem_raw);
  // XCD-aware order: all tiles of one (split, batch) -- which read the same pixel rows -- land on one XCD / L2
  const int flat = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int logical = fsd_conv::xcd_swizzle(flat, gridDim.x * gridDim.y * gridDim.z);
  const int tile = logical % gridDim.x;
  const int rest = logical / gridDim.x;
  const int split = rest % gridDim.y;
  const int zb = rest / gridDim.y;
  if (gridDim.z > 1) {
    p.dy += (long long)zb * p.dy_bs;
    p.x += (long long)zb * p.x_bs;
    p.ws += (long long)zb * p.ws_bs;
  }
  const int mt = tile / p.n_tiles, nt = tile - mt * p.n_tiles;
  const int m0 = mt * TILE, n0 = nt * TILE;
  const int p_begin = split * p.pix_per_split;
  const int p_end = min(p.M, p_begin + p.pix_per_split);
  const int nk = (p_end - p_begin + kBK - 1) / kBK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int cq = tid % CQ, kr = tid / CQ;          // this thread's 4-channel column group / first pixel row

  const int a_col = m0 + cq * 4;                   // A: dy columns are fixed per thread
  const bool a_ok = a_col < p.Cout;
  const int b_col = n0 + cq * 4;                   // B: the im2col column (tap, ci) is fixed per thread
  const bool b_colok = b_col < p.ncols;
  const int tap = b_colok ? b_col / p.cin4 : 0;
  const int ci = b_col - tap * p.cin4;
  const int ky = tap / p.ks, kx = tap - ky * p.ks;
  const int dyy = ky - p.pad, dxx = kx - p.pad;
  const int shift = dyy * p.W + dxx;

  // image coordinates of this thread's pixels, advanced by kBK pixels per chunk (no divisions in the loop)
  int yy_[PASSES], xx_[PASSES];
  const int step_y = kBK / p.W, step_x = kBK - step_y * p.W;
  if constexpr (!PLAIN) {
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
      const int pix0 = p_begin + kr + RPP * j;
      const int rem = pix0 % p.HW;
      yy_[j] = rem / p.W;
      xx_[j] = rem - yy_[j] * p.W;
    }
  }
  f32x4 ra[PASSES], rb[PASSES];
  f32x4 xs = {1.f, 1.f, 1.f, 1.f}, xh = {0.f, 0.f, 0.f, 0.f};
  if (p.x_scale && b_colok) {          // this thread's four x channels never change (ks == 1: column = channel)
    xs = *reinterpret_cast<const f32x4*>(p.x_scale + ci);
    xh = *reinterpret_cast<const f32x4*>(p.x_shift + ci);
  }
  unsigned okmask = 0;
  auto gload = [&](int kc) {
    okmask = 0;
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
      const int pix = p_begin + kc * kBK + kr + RPP * j;
      const bool pv = pix < p_end;
      const bool aok = pv && a_ok;
      ra[j] = *reinterpret_cast<const f32x4*>(p.dy + (aok ? (long long)pix * p.dy_ld + a_col : 0));
      bool bok;
      if constexpr (PLAIN) {       // 1x1 taps: a plain GEMM over rows, no image geometry
        bok = pv && b_colok;
      } else {
        const int yy = yy_[j], xx = xx_[j];
        bok = pv && b_colok && (unsigned)(yy + dyy) < (unsigned)p.H && (unsigned)(xx + dxx) < (unsigned)p.W;
        xx_[j] = xx + step_x;
        yy_[j] = yy + step_y;
        while (xx_[j] >= p.W) { xx_[j] -= p.W; ++yy_[j]; }
        while (yy_[j] >= p.H) yy_[j] -= p.H;
      }
      rb[j] = *reinterpret_cast<const f32x4*>(p.x + (bok ? (long long)(pix + shift) * p.x_ld + ci : 0));
      okmask |= (aok ? 1u : 0u) << j;
      okmask |= (bok ? 256u : 0u) << j;
    }
  };
  auto put = [&](lds_t* dst, f32x4 v, bool ok) {
    *reinterpret_cast<f32x4*>(dst) = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  constexpr int PT = kBK * TILE;                   // SPLIT: bf16 elements of one plane tile ([32 pixels][TILE channels])
  auto swz = [](int row) { return TILE == 128 ? ((row & 3) << 2) : ((row & 2) << 1); };   // XOR on the 16-byte piece index
  auto sstore = [&](lds_t* st) {
    if constexpr (SPLIT) {
      u16* sp = reinterpret_cast<u16*>(st);
#pragma unroll
      for (int j = 0; j < PASSES; ++j) {
        const int row = kr + RPP * j;
        const int off = row * TILE + (((cq >> 1) ^ swz(row)) << 3) + (cq & 1) * 4;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        uint2 h, m, l;
        fsd_conv::split3((okmask >> j) & 1u ? ra[j] : zero, h, m, l);
        *reinterpret_cast<uint2*>(sp + off) = h;
        *reinterpret_cast<uint2*>(sp + PT + off) = m;
        *reinterpret_cast<uint2*>(sp + 2 * PT + off) = l;
        fsd_conv::split3((okmask >> (8 + j)) & 1u ? (p.x_scale ? fsd_conv::affine_act4(rb[j], xs, xh, p.x_slope) : rb[j]) : zero, h, m, l);
        *reinterpret_cast<uint2*>(sp + 3 * PT + off) = h;
        *reinterpret_cast<uint2*>(sp + 4 * PT + off) = m;
        *reinterpret_cast<uint2*>(sp + 5 * PT + off) = l;
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
      put(st + (kr + RPP * j) * LDT + cq * 4, ra[j], (okmask >> j) & 1u);
      put(st + (kBK + kr + RPP * j) * LDT + cq * 4, p.x_scale ? fsd_conv::affine_act4(rb[j], xs, xh, p.x_slope) : rb[j],
          (okmask >> (8 + j)) & 1u);
    }
  };

  f32x16 acc[T][T];
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](const lds_t* st) {
    if constexpr (SPLIT) {
      // transpose-read geometry: 16-lane group G = lane >> 4 reads pixel rows kb + (L >> 2), channels cb + 4 * (L & 3); two
      // reads (rows +0..3, +4..7) give the lane the 8 consecutive pixels of its channel that one 16-wide MFMA step wants
      const u16* sp = reinterpret_cast<const u16*>(st);
      const int G = lane >> 4, Lq = lane & 15;
      auto frag = [&](const u16* tile, int ch0, int krow, int mask) -> bf16x8 {
        const int row = krow + (Lq >> 2);
        const int ch = ch0 + 16 * (G & 1) + 4 * (Lq & 3);
        const u16* a = tile + row * TILE + (((ch >> 3) ^ mask) << 3) + (ch & 7);
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a + 4 * TILE));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      };
#pragma unroll
      for (int s16 = 0; s16 < kBK / 16; ++s16) {
        const int krow = s16 * 16 + (G >> 1) * 8;
        const int mask = swz(krow + (Lq >> 2));                  // the +4 rows of the second read share it
        bf16x8 af[3][T], bf[3][T];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
          for (int i = 0; i < T; ++i) af[q][i] = frag(sp + q * PT, wm * (T * 32) + i * 32, krow, mask);
#pragma unroll
          for (int j = 0; j < T; ++j) bf[q][j] = frag(sp + (3 + q) * PT, wn * (T * 32) + j * 32, krow, mask);
        }
        constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};     // the six terms, smallest first
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int i = 0; i < T; ++i)
#pragma unroll
            for (int j = 0; j < T; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA[t]][i], bf[TB[t]][j], acc[i][j], 0, 0, 0);
      }
      return;
    }
    {
      const float* sa = st + (lane >> 5) * 4 * LDT + wm * (T * 32) + (lane & 31);
      const float* sb = st + kBK * LDT + (lane >> 5) * 4 * LDT + wn * (T * 32) + (lane & 31);
      // fragments one k8-step ahead of the MFMAs (statically indexed double buffer)
      float af[2][4][T], bf[2][4][T];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int i = 0; i < T; ++i) af[0][kk][i] = sa[kk * LDT + i * 32];
#pragma unroll
        for (int j = 0; j < T; ++j) bf[0][kk][j] = sb[kk * LDT + j * 32];
      }
#pragma unroll
      for (int k8 = 0; k8 < kBK / 8; ++k8) {
        if (k8 + 1 < kBK / 8) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < T; ++i) af[(k8 + 1) & 1][kk][i] = sa[((k8 + 1) * 8 + kk) * LDT + i * 32];
#pragma unroll
            for (int j = 0; j < T; ++j) bf[(k8 + 1) & 1][kk][j] = sb[((k8 + 1) * 8 + kk) * LDT + j * 32];
          }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int i = 0; i < T; ++i)
#pragma unroll
            for (int j = 0; j < T; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k8 & 1][kk][i], bf[k8 & 1][kk][j], acc[i][j], 0, 0, 0);
      }
    }
  };

  if constexpr (DMA) {
    // one wave instruction moves 2 rows of TILE floats (64 lanes x 16 B); a wave owns kBK/4 rows of each operand
    constexpr int ROWS_PER_INSTR = 256 / TILE;                 // 2 for TILE = 128
    constexpr int PER_WAVE = kBK / 4 / ROWS_PER_INSTR;         // instructions per wave per operand per chunk
    const int r2 = lane / (TILE / 4), c4 = (lane % (TILE / 4)) * 4;
    auto dma = [&](int kc, float* st) {
      const long long row0 = (long long)p_begin + (long long)kc * kBK + wave * (kBK / 4) + r2;
#pragma unroll
      for (int j = 0; j < PER_WAVE; ++j) {
        const long long row = row0 + j * ROWS_PER_INSTR;
        float* dstA = st + (wave * (kBK / 4) + j * ROWS_PER_INSTR) * LDT;
        __builtin_amdgcn_global_load_lds(p.dy + row * p.dy_ld + m0 + c4, dstA, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(p.x + row * p.x_ld + n0 + c4, dstA + kBK * LDT, 16, 0, 0);
      }
    };
    if (nk > 0) {
      dma(0, smem);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      int cur = 0;
      for (int kc = 0; kc < nk; ++kc) {
        if (kc + 1 < nk) dma(kc + 1, smem + (cur ^ 1) * STAGE);     // buffer last read before the previous barrier
        compute(smem + cur * STAGE);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // DMA of the next chunk has landed
        __syncthreads();
        cur ^= 1;
      }
    }
  } else if (nk > 0) {
    gload(0);
    sstore(smem);
    __syncthreads();
    int cur = 0;
    for (int kc = 0; kc < nk; ++kc) {
      const bool more = kc + 1 < nk;
      if (more) gload(kc + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(smem + cur * STAGE);
      __builtin_amdgcn_sched_barrier(0);
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

  float* out = p.ws + (long long)split * p.Cout * p.ncols;
  const int c_lane = lane & 31, r_lane = 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j) {
      const int n = n0 + wn * (T * 32) + j * 32 + c_lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (T * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + r_lane;
        if (m < p.Cout && n < p.ncols) out[(long long)m * p.ncols + n] = acc[i][j][r];
      }
    }
}

// ---- 8-wave split weight-gradient GEMM for the batched Winograd reductions (round 4) ----------------------------------------
// The twin of conv_gemm_split8_kernel (conv.hip) for  ws[b][split][co][ci] = sum_rows dy[b][row][co] * x[b][row][ci]:
// 256 (co) x 128 (ci) tile on 8 waves (4 x 2, wave tile 64 x 64), 32-row chunks, TWO LDS stages of three bf16 planes.  A
// thread stages 6 float4 per chunk instead of 8 for the same 48 MFMAs per wave, and their split + LDS stores are dealt out one
// micro-step behind each MFMA (the 4-wave kernel above issues 12 instructions per MFMA in its main loop).  The operand tiles
// stay 128-channel SUB-tiles ([32 rows][128 channels] bf16, 16-byte pieces XOR-permuted by (row & 3) << 2): the swizzle and
// the ds_read_b64_tr_b16 geometry of wgrad_kernel<128, ..., SPLIT> carry over unchanged.  Whole 32-row chunks only (the
// launcher sends the last < 32 rows through the 64x64 kernel into an extra workspace slot, like the DMA variant).
// ACT (the 1x1 layers' weight gradient, fsd_conv2d_wgrad_ex): x holds the RAW output of the producing convolution; the operand
// leaky(x * x_scale[ci] + x_shift[ci]) is formed in the staging registers one value per micro-step, ahead of its split (a
// thread's four x channels never change).  ACT = 1: 0 <= slope <= 1, max(t, t * slope); ACT = 2: the select (any slope).
template <int ACT>
__global__ __launch_bounds__(512, 2) void wgrad_split8_kernel(WgradArgs p) {
  constexpr int PT = kBK * 128;                    // bf16 elements of one plane of one sub-tile
  constexpr int STAGE = 9 * PT;                    // 3 sub-tiles (A0, A1, B) x 3 planes, in elements (72 KB)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u16* sm = reinterpret_cast<u16*>(smem_raw);
  const int flat = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int logical = fsd_conv::xcd_swizzle(flat, gridDim.x * gridDim.y * gridDim.z);
  const int tile = logical % gridDim.x;
  const int rest = logical / gridDim.x;
  const int split = rest % gridDim.y;
  const int zb = rest / gridDim.y;
  p.dy += (long long)zb * p.dy_bs;
  p.x += (long long)zb * p.x_bs;
  p.ws += (long long)zb * p.ws_bs;
  const int mt = tile / p.n_tiles, nt = tile - mt * p.n_tiles;
  const int m0 = mt * 256, n0 = nt * 128;
  const int p_begin = split * p.pix_per_split;
  const int p_end = min(p.M, p_begin + p.pix_per_split);
  const int nk = (p_end - p_begin) / kBK;          // whole chunks (launcher)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int cq = tid & 31, kr = tid >> 5;          // 4-channel column group / first row (second: + 16)

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nk > 0) {
    // float4 f = 2 * s + j of this thread: sub-tile s (0, 1: dy columns m0 + 128 s ...; 2: x columns n0 ...), row kr + 16 j
    const float* src[6];
#pragma unroll
    for (int f = 0; f < 6; ++f) {
      const int sct = f >> 1, j = f & 1;
      const long long row = (long long)p_begin + kr + 16 * j;
      src[f] = sct < 2 ? p.dy + row * p.dy_ld + m0 + sct * 128 + cq * 4 : p.x + row * p.x_ld + n0 + cq * 4;
    }
    const long long a_step = (long long)kBK * p.dy_ld, b_step = (long long)kBK * p.x_ld;
    const int st_off = kr * 128 + (((cq >> 1) ^ ((kr & 3) << 2)) << 3) + (cq & 1) * 4;     // elements; row + 16: + 2048
    f32x4 rr[6];
    auto gload = [&](int kc) {
#pragma unroll
      for (int f = 0; f < 6; ++f) rr[f] = *reinterpret_cast<const f32x4*>(src[f] + kc * (f < 4 ? a_step : b_step));
    };
    auto sstore = [&](u16* st) {
#pragma unroll
      for (int f = 0; f < 6; ++f) {
        uint2 h, m, l;
        fsd_conv::split3(rr[f], h, m, l);
        u16* d = st + (f >> 1) * 3 * PT + st_off + (f & 1) * 2048;
        *reinterpret_cast<uint2*>(d) = h;
        *reinterpret_cast<uint2*>(d + PT) = m;
        *reinterpret_cast<uint2*>(d + 2 * PT) = l;
      }
    };
    // transpose-read geometry of wgrad_kernel<128, ..., SPLIT>: 16-lane group G reads rows krow + (L >> 2), channels
    // ch0 + 16 (G & 1) + 4 (L & 3); two reads (rows +0..3, +4..7) = the 8 consecutive rows of its channel one MFMA step wants
    const int G = lane >> 4, Lq = lane & 15;
    const int fmask = ((Lq >> 2) & 3) << 2;                       // swz(krow + (Lq >> 2)): krow is a multiple of 8
    const int a_sub = (wm >> 1) * 3 * PT, a_ch = (wm & 1) * 64 + 16 * (G & 1) + 4 * (Lq & 3);
    const int b_ch = wn * 64 + 16 * (G & 1) + 4 * (Lq & 3);
    auto frag_at = [&](const u16* base, int ch, int krow) -> bf16x8 {
      const u16* a = base + (krow + (Lq >> 2)) * 128 + (((ch >> 3) ^ fmask) << 3) + (ch & 7);
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a + 4 * 128));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    bf16x8 af[2][3][2], bf[2][3][2];
    // fragment w (0, 1: A blocks i; 2, 3: B blocks j) of plane q, k-step ks
    auto frag_one = [&](const u16* st, int ks, int q, int w) {
      const int krow = ks * 16 + (G >> 1) * 8;
      if (w < 2) af[ks][q][w] = frag_at(st + a_sub + q * PT, a_ch + w * 32, krow);
      else bf[ks][q][w - 2] = frag_at(st + 6 * PT + q * PT, b_ch + (w - 2) * 32, krow);
    };
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    auto cvt2 = [](float a, float b) -> unsigned {                     // one v_cvt_pk_bf16_f32
      const f32x2 v = {a, b};
      return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    };
    auto lo_f = [](unsigned pk) -> float { return __builtin_bit_cast(float, pk << 16); };
    auto hi_f = [](unsigned pk) -> float { return __builtin_bit_cast(float, pk & 0xffff0000u); };
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};     // the six terms, smallest first

    const int last = nk - 1;
    f32x4 xs = {1.f, 1.f, 1.f, 1.f}, xh = {0.f, 0.f, 0.f, 0.f};
    const float xslope = p.x_slope;
    if constexpr (ACT != 0) {
      xs = *reinterpret_cast<const f32x4*>(p.x_scale + n0 + cq * 4);
      xh = *reinterpret_cast<const f32x4*>(p.x_shift + n0 + cq * 4);
    }
    auto act1 = [&](float v, int e) -> float {
      const float t = __builtin_fmaf(v, xs[e], xh[e]);
      if constexpr (ACT == 2) return t > 0.f ? t : t * xslope;
      return __builtin_fmaxf(t, t * xslope);
    };
    gload(0);
    if constexpr (ACT != 0) {
#pragma unroll
      for (int f = 4; f < 6; ++f)
#pragma unroll
        for (int e = 0; e < 4; ++e) rr[f][e] = act1(rr[f][e], e);
    }
    sstore(sm);
    gload(last < 1 ? last : 1);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
      u16* cur = sm + (kc & 1) * STAGE;
      u16* nxt = sm + ((kc & 1) ^ 1) * STAGE;
      const int kn = kc + 2 < last ? kc + 2 : last;
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int w = 0; w < 4; ++w) frag_one(cur, 0, q, w);
      unsigned h0, h1, m0_, m1_, l0, l1;
      float r0, r1, r2, r3;
#pragma unroll
      for (int u = 0; u < 48; ++u) {
        const int ks = u / 24, t = (u % 24) / 4, i = (u % 4) / 2, j = u % 2;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][TA[t]][i], bf[ks][TB[t]][j], acc[i][j], 0, 0, 0);
        if (u < 12) frag_one(cur, 1, u / 4, u % 4);                        // fragments of k-step 1: one (two reads) per MFMA
        const int f = u / 8, step = u % 8;                               // split + store of float4 f of chunk kc+1, 8 micro-steps
        if constexpr (ACT != 0) {
          // the x float4s (f = 4, 5) of chunk kc+1 are still raw: one value per even micro-step under the float4 before them
          if ((f == 3 || f == 4) && step % 2 == 0) rr[f + 1][step / 2] = act1(rr[f + 1][step / 2], step / 2);
        }
        const f32x4 v = rr[f];
        u16* d = nxt + (f >> 1) * 3 * PT + st_off + (f & 1) * 2048;
        if (step == 0) { h0 = cvt2(v[0], v[1]); h1 = cvt2(v[2], v[3]); }
        else if (step == 1) { r0 = v[0] - lo_f(h0); r1 = v[1] - hi_f(h0); }
        else if (step == 2) { r2 = v[2] - lo_f(h1); r3 = v[3] - hi_f(h1); *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1); }
        else if (step == 3) { m0_ = cvt2(r0, r1); m1_ = cvt2(r2, r3); }
        else if (step == 4) { r0 -= lo_f(m0_); r1 -= hi_f(m0_); }
        else if (step == 5) { r2 -= lo_f(m1_); r3 -= hi_f(m1_); *reinterpret_cast<uint2*>(d + PT) = make_uint2(m0_, m1_); }
        else if (step == 6) { l0 = cvt2(r0, r1); l1 = cvt2(r2, r3); }
        else {
          *reinterpret_cast<uint2*>(d + 2 * PT) = make_uint2(l0, l1);
          rr[f] = *reinterpret_cast<const f32x4*>(src[f] + kn * (f < 4 ? a_step : b_step));   // its successor (chunk kc+2)
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
  }

  float* out = p.ws + (long long)split * p.Cout * p.ncols;
  const int c_lane = lane & 31, r_lane = 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + c_lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + r_lane;
        if (m < p.Cout && n < p.ncols) out[(long long)m * p.ncols + n] = acc[i][j][r];
      }
    }
}

// dw[co][ci][ky][kx] = sum_s ws[s][co][tap*cin4 + ci].  Threads run along the workspace's fastest
// dimension (coalesced reads); SY lanes share the split loop when there are many splits (small
// layers) and are folded through LDS.  Fixed summation order -> deterministic.
template <int SY>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits,
                                                          int cout, int cin, int cin4, int taps, int ncols) {
  constexpr int NX = 256 / SY;
  __shared__ float s_part[SY][NX];
  const int tx = threadIdx.x % NX, ty = threadIdx.x / NX;
  const int col = blockIdx.x * NX + tx;
  const int co = blockIdx.y;
  const long long slice = (long long)cout * ncols;
  float v = 0.f;
  if (col < ncols) {
    const float* src = ws + (long long)co * ncols + col;
    int k = ty;
    for (; k + 3 * SY < splits; k += 4 * SY) {
      const float v0 = src[k * slice], v1 = src[(k + SY) * slice], v2 = src[(k + 2 * SY) * slice], v3 = src[(k + 3 * SY) * slice];
      v += v0; v += v1; v += v2; v += v3;
    }
    for (; k < splits; k += SY) v += src[k * slice];
  }
  if constexpr (SY > 1) {
    s_part[ty][tx] = v;
    __syncthreads();
    if (ty != 0) return;
#pragma unroll
    for (int y = 1; y < SY; ++y) v += s_part[y][tx];
  }
  if (col < ncols) {
    const int tap = col / cin4, ci = col - tap * cin4;
    if (ci < cin) dw[((long long)co * cin + ci) * taps + tap] = v;
  }
}


// ---- first layer: 3x3 weight gradient of a <=4-channel input, BatchNorm backward fused into the operand load ----
// dW[co][ci][ky][kx] = sum_pix dy[pix][co] * x[pix + (ky-1, kx-1)][ci],   dy = c1*(dt - c2 - xhat*c3)  (bn_bwd_apply).
// The first layer needs no data gradient, so dy is formed in registers and never written (saves a 3-pass
// elementwise kernel over the largest activation of the network).  HBM-bound: dt and y are read once.
// No LDS staging: every lane loads its own MFMA operands (32x32x2: lane = (column, k) with k = one of 2 pixels):
//   A[co][k]      = dy[pix + k][co]                   co = lane & 31          (128-B coalesced rows of dt / y)
//   B[k][(tap,ci)] = x[pix + k + shift(tap)][ci]      taps 0..7 = 32 columns  (L1/L2-resident 16-B pixels)
// and the ninth tap (4 columns) is a 4-FMA side sum per lane.  One wave owns a contiguous run of pixels.
struct FirstArgs {
  const void* dt; const void* y;    // float or bf16 (template parameter of the kernel), same leading dimensions in ELEMENTS
  const float* coef; const float* mean; const float* invstd; const float* x;
  float* ws;                        // [blocks][Cout][36]
  unsigned dt_ld, y_ld, x_ld;
  int H, W, Cout;
  long long pixels;
  int ppw;                          // pixels per wave (multiple of 16)
};

// SIDE = false: 3 input channels, the 27 (tap, ci) columns fit one 32-wide MFMA tile.
// SIDE = true : 4 input channels, taps 0..7 in the tile and the ninth tap as a 4-FMA side sum per lane.
template <bool SIDE, typename T>
__global__ __launch_bounds__(256) void wgrad_first_kernel(FirstArgs p) {
  constexpr unsigned ES = sizeof(T);                         // bytes per dt / y element
  __shared__ float s_out[4][32 * 36];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int co = blockIdx.y * 32 + c;
  const float c1 = p.coef[co], c2 = p.coef[p.Cout + co], c3 = p.coef[2 * p.Cout + co];
  const float mu = p.mean[co], is = p.invstd[co];
  const int tap = SIDE ? c >> 2 : c / 3, ci = SIDE ? c & 3 : c - 3 * (c / 3);
  const bool col_ok = SIDE || c < 27;
  const int ky = tap / 3, kx = tap - 3 * ky;
  const int dyo = ky - 1, dxo = kx - 1;
  const long long p_begin = ((long long)blockIdx.x * 4 + wave) * p.ppw;
  const long long p_end = p_begin + p.ppw < p.pixels ? p_begin + p.ppw : p.pixels;
  const int len = (int)(p_end - p_begin);                    // pixels of this wave (<= ppw)
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float acc8[4] = {0.f, 0.f, 0.f, 0.f};
  if (len > 0) {
    // image coordinates of this lane's first pixel; afterwards +2 pixels per k-step
    const long long pix0 = p_begin + h;
    const int rem = (int)(pix0 % ((long long)p.H * p.W));
    int yy = rem / p.W, xx = rem - yy * p.W;
    // 32-bit BYTE offsets from the (uniform) base pointers; the launcher guarantees they fit
    const char* dt_b = reinterpret_cast<const char*>(p.dt);
    const char* y_b = reinterpret_cast<const char*>(p.y);
    const char* x_b = reinterpret_cast<const char*>(p.x);
    unsigned off_dt = ((unsigned)pix0 * p.dt_ld + co) * ES, off_y = ((unsigned)pix0 * p.y_ld + co) * ES;
    unsigned off_x = (unsigned)pix0 * p.x_ld * 4u;
    const unsigned safe_dt = ((unsigned)p_begin * p.dt_ld + co) * ES, safe_y = ((unsigned)p_begin * p.y_ld + co) * ES;
    const unsigned step_dt = 2u * ES * p.dt_ld, step_y = 2u * ES * p.y_ld, step_x = 8u * p.x_ld;      // two pixels, in bytes
    const int kb = ((dyo * p.W + dxo) * (int)p.x_ld + ci) * 4, k8 = (p.W + 1) * (int)p.x_ld * 4;
    int rel = h;                                             // pixel index of this lane within the wave's run
    // Operands of 4 k-steps (8 pixels).  Loads are unconditional from clamped (always mapped) offsets and the
    // zero-selects happen at use, so all loads of a group are in flight together, one group ahead of the MFMAs.
    struct Group { float dtv[4], yv[4], bv[4]; f32x4 x8[SIDE ? 4 : 1]; unsigned mask; };
    auto load = [&](Group& g) {
      g.mask = 0;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bool valid = rel < len;
        g.dtv[s] = fsd_ew::ld1<T>(reinterpret_cast<const T*>(dt_b + (valid ? off_dt : safe_dt)));
        g.yv[s] = fsd_ew::ld1<T>(reinterpret_cast<const T*>(y_b + (valid ? off_y : safe_y)));
        const bool okb = valid && col_ok && (unsigned)(yy + dyo) < (unsigned)p.H && (unsigned)(xx + dxo) < (unsigned)p.W;
        g.bv[s] = *reinterpret_cast<const float*>(x_b + (okb ? off_x + (unsigned)kb : 0u));
        g.mask |= (valid ? 1u : 0u) << s;
        g.mask |= (okb ? 16u : 0u) << s;
        if constexpr (SIDE) {
          const bool ok8 = valid && yy + 1 < p.H && xx + 1 < p.W;
          g.x8[s] = *reinterpret_cast<const f32x4*>(x_b + (ok8 ? off_x + (unsigned)k8 : 0u));
          g.mask |= (ok8 ? 256u : 0u) << s;
        }
        rel += 2; off_dt += step_dt; off_y += step_y; off_x += step_x;
        xx += 2;                              // W >= 2 (checked by the launcher): at most one wrap
        if (xx >= p.W) { xx -= p.W; if (++yy >= p.H) yy = 0; }
      }
    };
    auto compute = [&](const Group& g) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float av = c1 * (g.dtv[s] - c2 - (g.yv[s] - mu) * is * c3);
        const float a = ((g.mask >> s) & 1u) ? av : 0.f;
        const float b = ((g.mask >> (4 + s)) & 1u) ? g.bv[s] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        if constexpr (SIDE) {
          const float a8 = ((g.mask >> (8 + s)) & 1u) ? a : 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) acc8[j] += a8 * g.x8[s][j];
        }
      }
    };
    Group g0, g1;
    load(g0);
    for (int done = 0; done < len; done += 16) {
      load(g1);
      __builtin_amdgcn_sched_barrier(0);
      compute(g0);
      __builtin_amdgcn_sched_barrier(0);
      load(g0);
      __builtin_amdgcn_sched_barrier(0);
      compute(g1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // fold the two pixel-halves of the side sum, then the 4 waves through LDS
  float* so = s_out[wave];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;        // output channel within the 32-tile
    if (col_ok) so[row * 36 + tap * 4 + ci] = acc[r];
  }
  if constexpr (SIDE) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc8[j] += __shfl_xor(acc8[j], 32, 64);
    if (h == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) so[c * 36 + 32 + j] = acc8[j];
    }
  } else {
    if (c < 9) {                                             // unused ci = 3 columns: keep them defined
#pragma unroll
      for (int r = 0; r < 16; ++r) so[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + c * 4 + 3] = 0.f;
    }
  }
  __syncthreads();
  float* out = p.ws + ((size_t)blockIdx.x * p.Cout + (size_t)blockIdx.y * 32) * 36;
  for (int e = threadIdx.x; e < 32 * 36; e += 256) out[e] = s_out[0][e] + s_out[1][e] + s_out[2][e] + s_out[3][e];
}

inline int first_blocks(long long pixels) {
  long long b = (pixels + 4 * 256 - 1) / (4 * 256);        // at least 256 pixels per wave
  return (int)(b < 1 ? 1 : b > 2048 ? 2048 : b);
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

inline int pick_splits(long long pixels, int tiles, int tile = 64) {
  // Two rounds of the 1536 co-resident 64x64 workgroups (6 per CU), rounded DOWN so that the launch does not spill a few
  // workgroups into a third round.  Measured round 2 (tools/layer_bench.py wgrad; 2048 rounded up -> 3072 rounded down):
  // 208x208 32->64 1.487 -> 1.403 ms, 104x104 0.671 -> 0.663, 52x52 0.425 -> 0.423, 26x26 0.382 -> 0.375.
  static const char* env = FSD_TUNE("FSD_WGRAD_TARGET");          // tuning aid: target number of 64x64 workgroups
  // (128x128 tiles, split arithmetic: half the workgroups -- two per CU are co-resident; measured 26x26 256->512
  // 0.254 -> 0.205 ms against a quarter, 13x13 1280->1024 0.628 -> 0.599)
  const int target = (env && atoi(env) > 0 ? atoi(env) : 3072) / (tile == 128 ? 2 : 1);
  int s = target / tiles;
  const long long max_s = (pixels + 255) / 256;      // at least 8 k-chunks per split
  if (s > max_s) s = (int)max_s;
  if (s < 1) s = 1;
  if (s > 4096) s = 4096;
  return s;
}

// fp32 tile variants: 0 = 64x64 single stage (17 KB LDS, 8 WGs/CU), 1 = 128x128 two stages, 2 = 128x128 single stage
inline int f32_variant() {
  static const char* env = getenv("FSD_WGRAD_TILE");     // tuning aid
  if (!env) return 0;
  const int v = atoi(env);
  return v == 128 ? 1 : v == 1281 ? 2 : v == 642 ? 3 : v == 640 ? 4 : 0;     // 642: 64x64 two stages, 640: no 1x1 specialisation
}
inline int tile_of(int bf16) { return (bf16 || f32_variant() == 1 || f32_variant() == 2) ? 128 : 64; }
// tile of the fp32 path for a dW of cout x ncols: split arithmetic takes the 128x128 tile where both dimensions fill it
inline int f32_tile(int cout, int ncols) {
  if (!fsd_conv::f32_split_on()) return tile_of(0);
  static const char* env = FSD_TUNE("FSD_WGRAD_SPLIT_TILE");      // tuning aid: 64 or 128
  if (env) return atoi(env) == 128 ? 128 : 64;
  return (cout % 128 == 0 && ncols >= 128) ? 128 : 64;
}

// 1x1 weight gradients on the 8-wave split kernel: > 0 = number of row splits (few, long ones), 0 = not this shape.
// OPT-IN (FSD_WGRAD1_SPLIT8=1).  Measured: 26x26 512->256 0.100 -> 0.083 ms, 13x13 1024->512 0.101 -> 0.087 -- but the kernel
// only pays with >= 32 chunks (1024+ rows) per split, and an fp32 accumulator that long is where the round-off goes: relative
// L2 error of dW against float64 5.7e-7 against 2.9e-7 for the many short splits of the 4-wave kernels (tests/test_gpu_split.py
// holds the split arithmetic to 1.25x the native kernel's error + 2e-7: missed by 1 %).  0.06 ms per step is not worth a
// looser gate.
inline int split8_1x1_splits(long long pixels, int cin, int cout, int ksize) {
  static const char* env = FSD_TUNE("FSD_WGRAD1_SPLIT8");
  if (!(env && env[0] == '1')) return 0;
  if (!fsd_conv::f32_split_on() || f32_variant() != 0 || ksize != 1 || cout % 256 || cin % 128 || cin < 256) return 0;
  const long long chunks = pixels / kBK;
  if (chunks < 64) return 0;
  const int tiles = (cout / 256) * (cin / 128);
  long long sp = chunks / 32;                       // >= 32 chunks per split
  const long long want = (256 + tiles - 1) / tiles; // ... but no more splits than it takes to give every CU a workgroup
  if (sp > want) sp = want;
  return (int)(sp < 1 ? 1 : sp);
}

int launch_wgrad(const WgradArgs& a, int bf16, dim3 grid, hipStream_t stream, int tile = 0) {
  // issued MFMA work: dW[Cout][ncols] reduced over M pixel rows, per batch (grid.z)
  fsd_prof::Scope prof(fsd_prof::kGemmWgrad, 2.0 * a.M * (double)a.Cout * a.ncols * grid.z, stream);
  if (bf16) {
    return FSD_ERR_UNSUPPORTED;       // bf16 operands: fsd_conv2d_wgrad_h
  } else if (fsd_conv::f32_split_on()) {
    // three bf16 planes of the [32 pixels][TILE channels] tiles of both operands, one stage
    if (tile == 128) {
      const size_t lds = 6 * (size_t)kBK * 128 * sizeof(u16);
      if (a.ks == 1) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<128, 1, false, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        FSD_LAUNCH((wgrad_kernel<128, 1, false, true, false, true>), grid, dim3(kThreads), lds, stream, a);
      } else {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<128, 1, false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        FSD_LAUNCH((wgrad_kernel<128, 1, false, false, false, true>), grid, dim3(kThreads), lds, stream, a);
      }
    } else {
      const size_t lds = 6 * (size_t)kBK * 64 * sizeof(u16);
      if (a.ks == 1) FSD_LAUNCH((wgrad_kernel<64, 1, false, true, false, true>), grid, dim3(kThreads), lds, stream, a);
      else FSD_LAUNCH((wgrad_kernel<64, 1, false, false, false, true>), grid, dim3(kThreads), lds, stream, a);
    }
  } else if (f32_variant() == 1) {
    const size_t lds = 2 * (size_t)(2 * kBK * (128 + 4)) * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<128, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    FSD_LAUNCH((wgrad_kernel<128, 2, false>), grid, dim3(kThreads), lds, stream, a);
  } else if (f32_variant() == 2) {
    const size_t lds = 1 * (size_t)(2 * kBK * (128 + 4)) * sizeof(float);
    FSD_LAUNCH((wgrad_kernel<128, 1, false>), grid, dim3(kThreads), lds, stream, a);
  } else if (f32_variant() == 3) {
    const size_t lds = 2 * (size_t)(2 * kBK * (64 + 4)) * sizeof(float);
    if (a.ks == 1) FSD_LAUNCH((wgrad_kernel<64, 2, false, true>), grid, dim3(kThreads), lds, stream, a);
    else FSD_LAUNCH((wgrad_kernel<64, 2, false>), grid, dim3(kThreads), lds, stream, a);
  } else {
    const size_t lds = 1 * (size_t)(2 * kBK * (64 + 4)) * sizeof(float);
    if (a.ks == 1 && f32_variant() != 4) FSD_LAUNCH((wgrad_kernel<64, 1, false, true>), grid, dim3(kThreads), lds, stream, a);
    else FSD_LAUNCH((wgrad_kernel<64, 1, false>), grid, dim3(kThreads), lds, stream, a);
  }
  return 0;
}

int wgrad_impl(const float* dy, long long dy_ld, const float* x, long long x_ld, float* dw_oihw, void* workspace,
               size_t workspace_bytes, int batch, int height, int width, int cin, int cout, int ksize, int bf16,
               hipStream_t stream, const float* x_scale = nullptr, const float* x_shift = nullptr, float x_slope = 1.f) {
  (void)hipGetLastError();   // drop a stale error left by someone else's earlier call
  if ((x_scale == nullptr) != (x_shift == nullptr)) return FSD_ERR_ARG;
  if (x_scale && (ksize != 1 || (cin & 3))) return FSD_ERR_UNSUPPORTED;
  if (!dy || !x || !dw_oihw || !workspace || batch < 1 || height < 1 || width < 1 || cin < 1 || cout < 1) return FSD_ERR_ARG;
  if (ksize != 1 && ksize != 3) return FSD_ERR_UNSUPPORTED;
  const int cin4 = round_up(cin, 4);
  if ((dy_ld & 3) || (x_ld & 3) || dy_ld < round_up(cout, 4) || x_ld < cin4) return FSD_ERR_ARG;
  const long long pixels = (long long)batch * height * width;
  if (pixels > 0x7fffffffLL - 4096) return FSD_ERR_UNSUPPORTED;
  if (!bf16 && split8_1x1_splits(pixels, cin, cout, ksize) > 0) {
    // 1x1 layer = plain reduction GEMM over the pixel rows: 256x128 tiles on 8 waves (wgrad_split8_kernel) over the whole 32-row
    // chunks, few LONG splits (>= 32 chunks each: a split stores 128 KB), the last < 32 rows through the 64x64 kernel
    const int sp = split8_1x1_splits(pixels, cin, cout, ksize);
    const long long full = pixels / kBK * kBK;
    const int tail = (int)(pixels - full);
    const int slots = sp + (tail ? 1 : 0);
    if (workspace_bytes < (size_t)slots * cout * cin * sizeof(float)) return FSD_ERR_WORKSPACE;
    WgradArgs a;
    a.dy = dy; a.x = x; a.ws = reinterpret_cast<float*>(workspace);
    a.dy_ld = dy_ld; a.x_ld = x_ld;
    a.H = 1; a.W = (int)full; a.HW = (int)full; a.M = (int)full;
    a.Cout = cout; a.cin4 = cin; a.ks = 1; a.pad = 0; a.ncols = cin;
    a.m_tiles = cout / 256; a.n_tiles = cin / 128;
    a.pix_per_split = round_up((int)((full + sp - 1) / sp), kBK);
    a.dy_bs = a.x_bs = a.ws_bs = 0;
    a.x_scale = x_scale; a.x_shift = x_shift; a.x_slope = x_slope;
    const size_t lds = 2 * (size_t)9 * kBK * 128 * sizeof(u16);
    const dim3 grid(a.m_tiles * a.n_tiles, sp, 1);
    {
      fsd_prof::Scope prof(fsd_prof::kGemmWgrad, 2.0 * a.M * (double)cout * cin, stream);
      const int act = !x_scale ? 0 : (x_slope >= 0.f && x_slope <= 1.f) ? 1 : 2;
      hipError_t e = hipSuccess;
      if (act == 0) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_split8_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) FSD_LAUNCH(wgrad_split8_kernel<0>, grid, dim3(512), lds, stream, a);
      } else if (act == 1) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_split8_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) FSD_LAUNCH(wgrad_split8_kernel<1>, grid, dim3(512), lds, stream, a);
      } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_split8_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) FSD_LAUNCH(wgrad_split8_kernel<2>, grid, dim3(512), lds, stream, a);
      }
      if (e != hipSuccess) return (int)e;
    }
    if (tail) {
      WgradArgs t = a;
      t.dy = dy + full * dy_ld; t.x = x + full * x_ld; t.ws = a.ws + (long long)sp * cout * cin;
      t.H = 1; t.W = tail; t.HW = tail; t.M = tail;
      t.m_tiles = (cout + 63) / 64; t.n_tiles = (cin + 63) / 64;
      t.pix_per_split = kBK;
      if (int rc = launch_wgrad(t, 0, dim3(t.m_tiles * t.n_tiles, 1, 1), stream, 64)) return rc;
    }
    if (slots <= 8)
      FSD_LAUNCH(wgrad_reduce_kernel<1>, dim3((cin + 255) / 256, cout), dim3(256), 0, stream, a.ws, dw_oihw, slots, cout, cin, cin, 1, cin);
    else
      FSD_LAUNCH(wgrad_reduce_kernel<8>, dim3((cin + 31) / 32, cout), dim3(256), 0, stream, a.ws, dw_oihw, slots, cout, cin, cin, 1, cin);
    return (int)hipGetLastError();
  }
  if (!bf16 && !x_scale && f32_variant() == 0 && fsd_conv::wgrad3x3_halo_ok(height, width, cin, cout, ksize) &&
      (reinterpret_cast<uintptr_t>(dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    // 32 -> 64 channels (darknet L2, the reweighting net's second layer): 8 x 8 pixel blocks, dy and the halo patch of x
    // fetched and split once for all nine taps (wgrad_halo.hip); one 64 x 288 partial per workgroup
    const int ncols = 9 * cin;
    if (workspace_bytes < (size_t)fsd_conv::wgrad3x3_halo_slots(batch, height, width) * cout * ncols * sizeof(float))
      return FSD_ERR_WORKSPACE;
    int slots = 0;
    if (int rc = fsd_conv::wgrad3x3_halo(dy, dy_ld, x, x_ld, reinterpret_cast<float*>(workspace), batch, height, width, &slots, stream))
      return rc;
    if (slots <= 8)
      FSD_LAUNCH(wgrad_reduce_kernel<1>, dim3((ncols + 255) / 256, cout), dim3(256), 0, stream, reinterpret_cast<const float*>(workspace),
                 dw_oihw, slots, cout, cin, cin, 9, ncols);
    else
      FSD_LAUNCH(wgrad_reduce_kernel<8>, dim3((ncols + 31) / 32, cout), dim3(256), 0, stream, reinterpret_cast<const float*>(workspace),
                 dw_oihw, slots, cout, cin, cin, 9, ncols);
    return (int)hipGetLastError();
  }
  const int tile = bf16 ? tile_of(bf16) : f32_tile(cout, ksize * ksize * cin4);
  WgradArgs a;
  a.dy = dy; a.x = x; a.ws = reinterpret_cast<float*>(workspace);
  a.dy_ld = dy_ld; a.x_ld = x_ld;
  a.H = height; a.W = width; a.HW = height * width; a.M = (int)pixels;
  a.Cout = cout; a.cin4 = cin4; a.ks = ksize; a.pad = (ksize - 1) / 2;
  a.ncols = ksize * ksize * cin4;
  a.m_tiles = (cout + tile - 1) / tile;
  a.n_tiles = (a.ncols + tile - 1) / tile;
  const int splits = pick_splits(pixels, a.m_tiles * a.n_tiles, tile);
  if (workspace_bytes < (size_t)splits * cout * a.ncols * sizeof(float)) return FSD_ERR_WORKSPACE;
  a.pix_per_split = round_up((int)((pixels + splits - 1) / splits), kBK);
  a.dy_bs = a.x_bs = a.ws_bs = 0;
  a.x_scale = x_scale; a.x_shift = x_shift; a.x_slope = x_slope;
  const dim3 grid(a.m_tiles * a.n_tiles, splits);
  if (int rc = launch_wgrad(a, bf16, grid, stream, tile)) return rc;
  if (splits <= 8)
    FSD_LAUNCH(wgrad_reduce_kernel<1>, dim3((a.ncols + 255) / 256, cout), dim3(256), 0, stream, a.ws, dw_oihw,
                       splits, cout, cin, cin4, ksize * ksize, a.ncols);
  else
    FSD_LAUNCH(wgrad_reduce_kernel<8>, dim3((a.ncols + 31) / 32, cout), dim3(256), 0, stream, a.ws, dw_oihw,
                       splits, cout, cin, cin4, ksize * ksize, a.ncols);
  return (int)hipGetLastError();
}

}  // namespace

// 16 (or any number of) independent reduction GEMMs  ws[b][split][m][n] = sum_rows dy[b][row][m] * x[b][row][n]
// on the fp32 weight-gradient kernel (ks = 1).  Returns the number of splits through *splits_out.
// Plan of a batched reduction GEMM: 128x128 DMA-staged tiles over the full 32-row chunks when both channel counts
// are multiples of 128 (+ one extra workspace slot filled by the 64x64 kernel for the last < 32 rows), else 64x64.
struct BatchedPlan { bool dma; int splits; int tail_rows; int slots; bool s8; };
inline BatchedPlan batched_plan(long long rows, int cin, int cout, int batches) {
  static const char* env = FSD_TUNE("FSD_WGRAD_DMA");         // tuning aid: 0 disables the DMA variant
  const bool allow = !(env && env[0] == '0') && f32_variant() == 0 && !fsd_conv::f32_split_on();
  BatchedPlan pl;
  pl.s8 = false;
  const long long full = rows / kBK * kBK;
  static const char* env8 = getenv("FSD_WGRAD_SPLIT8");     // tuning aid: 0 keeps the 4-wave split kernel
  if (fsd_conv::f32_split_on() && f32_variant() == 0 && !(env8 && env8[0] == '0') && cout % 256 == 0 && cin % 128 == 0 &&
      cin >= 256 && full >= 8 * kBK) {     // (cin = 128: measured equal or slower than the 4-wave kernel)
    // split arithmetic: 256x128 tiles on 8 waves, one workgroup per CU (wgrad_split8_kernel); ~4 rounds of the 256 CUs
    pl.s8 = true;
    pl.dma = false;
    const int tiles = (cout / 256) * (cin / 128) * batches;
    const long long max_s = full / (8 * kBK);
    static const char* env_t = FSD_TUNE("FSD_WGRAD_S8_TARGET");
    const int target = env_t && atoi(env_t) > 0 ? atoi(env_t) : 1024;
    int sp = (target + tiles - 1) / tiles;
    if (sp > max_s) sp = (int)max_s;
    pl.splits = sp < 1 ? 1 : sp;
    pl.tail_rows = (int)(rows - full);
    pl.slots = pl.splits + (pl.tail_rows ? 1 : 0);
    return pl;
  }
  // measured (tools/layer_bench.py wgrad): +5-7 % on the 1024/1280-channel layers, +2 % at 512, -2 % at 128/256
  pl.dma = allow && cout % 128 == 0 && cin % 128 == 0 && cin >= 512 && cout >= 512 && full >= 8 * kBK;
  if (pl.dma) {
    const int tiles = (cout / 128) * (cin / 128) * batches;
    // 512 co-resident 128x128 workgroups: ~3 rounds of blocks are enough, every extra split is another workspace
    // slice for the fold kernel to read.  (Choosing the split count so that the last round of blocks is full -- 2304
    // tiles x 2 splits = 9.0 rounds instead of 4.5 -- was measured and LOSES 5-10 %: 0.721 -> 0.762 ms at 1024 -> 1024.)
    const long long max_s = full / (8 * kBK);
    static const char* env_d = FSD_TUNE("FSD_WGRAD_DMA_TARGET");      // tuning aid: target number of 128x128 workgroups
    const int target_d = env_d && atoi(env_d) > 0 ? atoi(env_d) : 1536;
    int sp = (target_d + tiles - 1) / tiles;
    if (sp > max_s) sp = (int)max_s;
    pl.splits = sp < 1 ? 1 : sp;
    pl.tail_rows = (int)(rows - full);
    pl.slots = pl.splits + (pl.tail_rows ? 1 : 0);
  } else {
    const int tile = f32_tile(cout, cin);
    const int tiles = ((cout + tile - 1) / tile) * ((cin + tile - 1) / tile) * batches;
    pl.splits = pick_splits(rows, tiles, tile);
    pl.tail_rows = 0;
    pl.slots = pl.splits;
  }
  return pl;
}

int fsd_conv::wgrad_gemm_batched(const float* dy, long long dy_ld, long long dy_bs, const float* x, long long x_ld,
                                 long long x_bs, float* ws, long long rows, int cin, int cout, int batches,
                                 int* splits_out, hipStream_t stream) {
  if (rows < 1 || rows > 0x7fffffffLL - 4096 || (cin & 3) || (dy_ld & 3) || (x_ld & 3)) return FSD_ERR_UNSUPPORTED;
  const BatchedPlan pl = batched_plan(rows, cin, cout, batches);
  WgradArgs a;
  a.dy = dy; a.x = x; a.ws = ws;
  a.dy_ld = dy_ld; a.x_ld = x_ld;
  a.Cout = cout; a.cin4 = cin; a.ks = 1; a.pad = 0;
  a.ncols = cin;
  a.dy_bs = dy_bs; a.x_bs = x_bs; a.ws_bs = (long long)pl.slots * cout * cin;
  a.x_scale = a.x_shift = nullptr; a.x_slope = 1.f;
  *splits_out = pl.slots;
  if (pl.dma || pl.s8) {
    const long long full = rows - pl.tail_rows;
    a.H = 1; a.W = (int)full; a.HW = (int)full; a.M = (int)full;
    a.m_tiles = cout / (pl.s8 ? 256 : 128);
    a.n_tiles = cin / 128;
    a.pix_per_split = round_up((int)((full + pl.splits - 1) / pl.splits), kBK);
    if (pl.s8) {
      const size_t lds = 2 * (size_t)9 * kBK * 128 * sizeof(u16);       // two stages x 3 sub-tiles x 3 planes
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_split8_kernel<0>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      fsd_prof::Scope prof(fsd_prof::kGemmWgrad, 2.0 * a.M * (double)cout * cin * batches, stream);
      FSD_LAUNCH(wgrad_split8_kernel<0>, dim3(a.m_tiles * a.n_tiles, pl.splits, batches), dim3(512), lds, stream, a);
    } else {
      const size_t lds = 2 * (size_t)(2 * kBK * 128) * sizeof(float);
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<128, 2, false, true, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      fsd_prof::Scope prof(fsd_prof::kGemmWgrad, 2.0 * a.M * (double)cout * cin * batches, stream);
      FSD_LAUNCH((wgrad_kernel<128, 2, false, true, true>), dim3(a.m_tiles * a.n_tiles, pl.splits, batches),
                         dim3(kThreads), lds, stream, a);
    }
    if (pl.tail_rows) {                                      // the last < 32 rows -> workspace slot `splits`
      WgradArgs t = a;
      t.dy = dy + full * dy_ld; t.x = x + full * x_ld; t.ws = ws + (long long)pl.splits * cout * cin;
      t.H = 1; t.W = pl.tail_rows; t.HW = pl.tail_rows; t.M = pl.tail_rows;
      t.m_tiles = (cout + 63) / 64;
      t.n_tiles = (cin + 63) / 64;
      t.pix_per_split = kBK;
      const size_t lds1 = (size_t)(2 * kBK * (64 + 4)) * sizeof(float);
      fsd_prof::Scope prof(fsd_prof::kGemmWgrad, 2.0 * t.M * (double)cout * cin * batches, stream);
      FSD_LAUNCH((wgrad_kernel<64, 1, false, true>), dim3(t.m_tiles * t.n_tiles, 1, batches), dim3(kThreads), lds1,
                         stream, t);
    }
    return (int)hipGetLastError();
  }
  a.H = 1; a.W = (int)rows; a.HW = (int)rows; a.M = (int)rows;
  const int tile = f32_tile(cout, cin);
  a.m_tiles = (cout + tile - 1) / tile;
  a.n_tiles = (cin + tile - 1) / tile;
  a.pix_per_split = round_up((int)((rows + pl.splits - 1) / pl.splits), kBK);
  if (int rc = launch_wgrad(a, 0, dim3(a.m_tiles * a.n_tiles, pl.splits, batches), stream, tile)) return rc;
  return (int)hipGetLastError();
}

int fsd_conv::wgrad_batched_splits(long long rows, int cin, int cout, int batches) {
  return batched_plan(rows, cin, cout, batches).slots;
}

int fsd_conv::wgrad_batched_plan(long long rows, int cin, int cout, int batches, int* dma, int* splits, int* tail_rows) {
  const BatchedPlan pl = batched_plan(rows, cin, cout, batches);
  if (dma) *dma = pl.dma ? 1 : 0;
  if (splits) *splits = pl.splits;
  if (tail_rows) *tail_rows = pl.tail_rows;
  return pl.slots;
}

extern "C" size_t fsd_conv3x3_wgrad_c4_bnfused_workspace_bytes(int batch, int height, int width, int cout) {
  return (size_t)first_blocks((long long)batch * height * width) * cout * 36 * sizeof(float);
}

namespace {

template <typename T>
int wgrad_first_impl(const T* dt, long long dt_ld, const T* y, long long y_ld, const float* coef, const float* mean,
                     const float* invstd, const float* x, long long x_ld, float* dw_oihw, void* workspace,
                     size_t workspace_bytes, int batch, int height, int width, int cin, int cout, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dt || !y || !coef || !mean || !invstd || !x || !dw_oihw || !workspace || batch < 1 || height < 1 || width < 1)
    return FSD_ERR_ARG;
  if (cin < 1 || cin > 4 || cout % 32 || width < 2 || x_ld < 4 || (x_ld & 3) || dt_ld < cout || y_ld < cout) return FSD_ERR_UNSUPPORTED;
  const long long pixels = (long long)batch * height * width;
  if ((pixels + width + 18) * (dt_ld > y_ld ? (dt_ld > x_ld ? dt_ld : x_ld) : (y_ld > x_ld ? y_ld : x_ld)) * 4 >= 0xffffffffLL)
    return FSD_ERR_UNSUPPORTED;                              // 32-bit byte offsets in the kernel
  if ((reinterpret_cast<uintptr_t>(x) & 15)) return FSD_ERR_ARG;
  const int blocks = first_blocks(pixels);
  if (workspace_bytes < (size_t)blocks * cout * 36 * sizeof(float)) return FSD_ERR_WORKSPACE;
  FirstArgs a;
  a.dt = dt; a.y = y; a.coef = coef; a.mean = mean; a.invstd = invstd; a.x = x;
  a.ws = reinterpret_cast<float*>(workspace);
  a.dt_ld = (unsigned)dt_ld; a.y_ld = (unsigned)y_ld; a.x_ld = (unsigned)x_ld;
  a.H = height; a.W = width; a.Cout = cout; a.pixels = pixels;
  a.ppw = round_up((int)((pixels + (long long)blocks * 4 - 1) / ((long long)blocks * 4)), 16);
  {
    fsd_prof::Scope prof(fsd_prof::kFirst, (double)pixels * (2.0 * sizeof(T) * cout + 16.0), stream);
    if (cin == 4) FSD_LAUNCH((wgrad_first_kernel<true, T>), dim3(blocks, cout / 32), dim3(256), 0, stream, a);
    else FSD_LAUNCH((wgrad_first_kernel<false, T>), dim3(blocks, cout / 32), dim3(256), 0, stream, a);
  }
  if (blocks <= 8)
    FSD_LAUNCH(wgrad_reduce_kernel<1>, dim3(1, cout), dim3(256), 0, stream, a.ws, dw_oihw, blocks, cout, cin, 4, 9, 36);
  else
    FSD_LAUNCH(wgrad_reduce_kernel<8>, dim3(2, cout), dim3(256), 0, stream, a.ws, dw_oihw, blocks, cout, cin, 4, 9, 36);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int fsd_conv3x3_wgrad_c4_bnfused(const float* dt, long long dt_ld, const float* y, long long y_ld,
                                            const float* coef, const float* mean, const float* invstd, const float* x,
                                            long long x_ld, float* dw_oihw, void* workspace, size_t workspace_bytes,
                                            int batch, int height, int width, int cin, int cout, hipStream_t stream) {
  return wgrad_first_impl<float>(dt, dt_ld, y, y_ld, coef, mean, invstd, x, x_ld, dw_oihw, workspace, workspace_bytes, batch,
                                 height, width, cin, cout, stream);
}

extern "C" int fsd_conv3x3_wgrad_c4_bnfused_h(const void* dt, long long dt_ld, const void* y, long long y_ld,
                                              const float* coef, const float* mean, const float* invstd, const float* x,
                                              long long x_ld, float* dw_oihw, void* workspace, size_t workspace_bytes,
                                              int batch, int height, int width, int cin, int cout, hipStream_t stream) {
  return wgrad_first_impl<fsd_ew::bf16_t>(static_cast<const fsd_ew::bf16_t*>(dt), dt_ld, static_cast<const fsd_ew::bf16_t*>(y),
                                          y_ld, coef, mean, invstd, x, x_ld, dw_oihw, workspace, workspace_bytes, batch,
                                          height, width, cin, cout, stream);
}

extern "C" size_t fsd_conv2d_wgrad_workspace_bytes(int batch, int height, int width, int cin, int cout, int ksize) {
  // the larger of the two tilings the fp32 path may use (64x64; 128x128 under split arithmetic): valid for either mode
  const long long pixels = (long long)batch * height * width;
  const int ncols = ksize * ksize * round_up(cin, 4);
  int splits = 0;
  for (int tile : {64, 128, tile_of(0)}) {       // every tiling f32_tile() can return, whatever FSD_WGRAD_F32 says
    const int tiles = ((cout + tile - 1) / tile) * ((ncols + tile - 1) / tile);
    const int s = pick_splits(pixels, tiles, tile);
    splits = s > splits ? s : splits;
  }
  const int s8 = split8_1x1_splits(pixels, cin, cout, ksize);
  if (s8 + 1 > splits) splits = s8 + 1;             // + the slot of the < 32-row side launch
  if (ksize == 3 && cin == 32 && cout == 64 && height % 8 == 0 && width % 8 == 0) {      // wgrad_halo.hip's partials (either mode)
    const int hs = fsd_conv::wgrad3x3_halo_slots(batch, height, width);
    if (hs > splits) splits = hs;
  }
  return (size_t)splits * cout * ncols * sizeof(float);
}

extern "C" int fsd_conv2d_wgrad_ex(const float* dy, long long dy_ld, const float* x, long long x_ld, float* dw_oihw,
                                   void* workspace, size_t workspace_bytes, int batch, int height, int width, int cin,
                                   int cout, int ksize, const float* x_scale, const float* x_shift, float x_slope,
                                   hipStream_t stream) {
  return wgrad_impl(dy, dy_ld, x, x_ld, dw_oihw, workspace, workspace_bytes, batch, height, width, cin, cout, ksize, 0, stream,
                    x_scale, x_shift, x_slope);
}

extern "C" int fsd_conv2d_wgrad(const float* dy, long long dy_ld, const float* x, long long x_ld, float* dw_oihw,
                                void* workspace, size_t workspace_bytes, int batch, int height, int width, int cin,
                                int cout, int ksize, hipStream_t stream) {
  return wgrad_impl(dy, dy_ld, x, x_ld, dw_oihw, workspace, workspace_bytes, batch, height, width, cin, cout, ksize, 0,
                    stream);
}
