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

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// TILE x TILE outputs per workgroup (2x2 waves), STAGES LDS buffers.
//   BF16 = false: exact fp32 MFMA (32x32x2), LDS tiles hold floats.            TILE 64, 1 stage (17 KB, 8 WGs/CU)
//   BF16 = true : operands rounded to bf16 (RNE) on the LDS store, 32x32x16 MFMA with fp32 accumulation
//                 (BASELINE C3/C5); fragments need 8 consecutive PIXELS per lane, which are LDS rows here, so
//                 they are gathered with 16-bit LDS reads (pairs land in the two halves of one VGPR).
template <int TILE, int STAGES, bool BF16, bool PLAIN = false>
__global__ __launch_bounds__(kThreads) void wgrad_kernel(WgradArgs p) {
  typedef typename std::conditional<BF16, unsigned short, float>::type lds_t;
  constexpr int LDT = TILE + (BF16 ? 8 : 4);       // LDS row stride in elements (keeps 16-byte alignment)
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
    if constexpr (BF16) {
      uint2 w = make_uint2(0u, 0u);
      if (ok) {
        const __bf16 b0 = (__bf16)v[0], b1 = (__bf16)v[1], b2 = (__bf16)v[2], b3 = (__bf16)v[3];
        w.x = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
        w.y = (unsigned)__builtin_bit_cast(unsigned short, b2) | ((unsigned)__builtin_bit_cast(unsigned short, b3) << 16);
      }
      *reinterpret_cast<uint2*>(dst) = w;
    } else {
      *reinterpret_cast<f32x4*>(dst) = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto sstore = [&](lds_t* st) {
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
      put(st + (kr + RPP * j) * LDT + cq * 4, ra[j], (okmask >> j) & 1u);
      put(st + (kBK + kr + RPP * j) * LDT + cq * 4, rb[j], (okmask >> (8 + j)) & 1u);
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
    if constexpr (BF16) {
      // lane (m = lane & 31, h = lane >> 5) supplies pixels s*16 + h*8 .. +7 of column m
      const unsigned short* sa = st + (lane >> 5) * 8 * LDT + wm * (T * 32) + (lane & 31);
      const unsigned short* sb = st + kBK * LDT + (lane >> 5) * 8 * LDT + wn * (T * 32) + (lane & 31);
#pragma unroll
      for (int s16 = 0; s16 < kBK / 16; ++s16) {
        bf16x8 af[T], bf[T];
#pragma unroll
        for (int i = 0; i < T; ++i) {
          unsigned short v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = sa[(s16 * 16 + e) * LDT + i * 32];
#pragma unroll
          for (int e = 0; e < 8; ++e) af[i][e] = __builtin_bit_cast(__bf16, v[e]);
        }
#pragma unroll
        for (int j = 0; j < T; ++j) {
          unsigned short v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = sb[(s16 * 16 + e) * LDT + j * 32];
#pragma unroll
          for (int e = 0; e < 8; ++e) bf[j][e] = __builtin_bit_cast(__bf16, v[e]);
        }
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
          for (int j = 0; j < T; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    } else {
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

  if (nk > 0) {
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

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

inline int pick_splits(long long pixels, int tiles) {
  int s = (4096 + tiles - 1) / tiles;
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

int launch_wgrad(const WgradArgs& a, int bf16, dim3 grid, hipStream_t stream) {
  if (bf16) {
    const size_t lds = 2 * (size_t)(2 * kBK * (128 + 8)) * sizeof(unsigned short);
    hipLaunchKernelGGL((wgrad_kernel<128, 2, true>), grid, dim3(kThreads), lds, stream, a);
  } else if (f32_variant() == 1) {
    const size_t lds = 2 * (size_t)(2 * kBK * (128 + 4)) * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<128, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((wgrad_kernel<128, 2, false>), grid, dim3(kThreads), lds, stream, a);
  } else if (f32_variant() == 2) {
    const size_t lds = 1 * (size_t)(2 * kBK * (128 + 4)) * sizeof(float);
    hipLaunchKernelGGL((wgrad_kernel<128, 1, false>), grid, dim3(kThreads), lds, stream, a);
  } else if (f32_variant() == 3) {
    const size_t lds = 2 * (size_t)(2 * kBK * (64 + 4)) * sizeof(float);
    if (a.ks == 1) hipLaunchKernelGGL((wgrad_kernel<64, 2, false, true>), grid, dim3(kThreads), lds, stream, a);
    else hipLaunchKernelGGL((wgrad_kernel<64, 2, false>), grid, dim3(kThreads), lds, stream, a);
  } else {
    const size_t lds = 1 * (size_t)(2 * kBK * (64 + 4)) * sizeof(float);
    if (a.ks == 1 && f32_variant() != 4) hipLaunchKernelGGL((wgrad_kernel<64, 1, false, true>), grid, dim3(kThreads), lds, stream, a);
    else hipLaunchKernelGGL((wgrad_kernel<64, 1, false>), grid, dim3(kThreads), lds, stream, a);
  }
  return 0;
}

int wgrad_impl(const float* dy, long long dy_ld, const float* x, long long x_ld, float* dw_oihw, void* workspace,
               size_t workspace_bytes, int batch, int height, int width, int cin, int cout, int ksize, int bf16,
               hipStream_t stream) {
  (void)hipGetLastError();   // drop a stale error left by someone else's earlier call
  if (!dy || !x || !dw_oihw || !workspace || batch < 1 || height < 1 || width < 1 || cin < 1 || cout < 1) return FSD_ERR_ARG;
  if (ksize != 1 && ksize != 3) return FSD_ERR_UNSUPPORTED;
  const int cin4 = round_up(cin, 4);
  if ((dy_ld & 3) || (x_ld & 3) || dy_ld < round_up(cout, 4) || x_ld < cin4) return FSD_ERR_ARG;
  const long long pixels = (long long)batch * height * width;
  if (pixels > 0x7fffffffLL - 4096) return FSD_ERR_UNSUPPORTED;
  const int tile = tile_of(bf16);
  WgradArgs a;
  a.dy = dy; a.x = x; a.ws = reinterpret_cast<float*>(workspace);
  a.dy_ld = dy_ld; a.x_ld = x_ld;
  a.H = height; a.W = width; a.HW = height * width; a.M = (int)pixels;
  a.Cout = cout; a.cin4 = cin4; a.ks = ksize; a.pad = (ksize - 1) / 2;
  a.ncols = ksize * ksize * cin4;
  a.m_tiles = (cout + tile - 1) / tile;
  a.n_tiles = (a.ncols + tile - 1) / tile;
  const int splits = pick_splits(pixels, a.m_tiles * a.n_tiles);
  if (workspace_bytes < (size_t)splits * cout * a.ncols * sizeof(float)) return FSD_ERR_WORKSPACE;
  a.pix_per_split = round_up((int)((pixels + splits - 1) / splits), kBK);
  a.dy_bs = a.x_bs = a.ws_bs = 0;
  const dim3 grid(a.m_tiles * a.n_tiles, splits);
  if (int rc = launch_wgrad(a, bf16, grid, stream)) return rc;
  if (splits <= 8)
    hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((a.ncols + 255) / 256, cout), dim3(256), 0, stream, a.ws, dw_oihw,
                       splits, cout, cin, cin4, ksize * ksize, a.ncols);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3((a.ncols + 31) / 32, cout), dim3(256), 0, stream, a.ws, dw_oihw,
                       splits, cout, cin, cin4, ksize * ksize, a.ncols);
  return (int)hipGetLastError();
}

}  // namespace

// 16 (or any number of) independent reduction GEMMs  ws[b][split][m][n] = sum_rows dy[b][row][m] * x[b][row][n]
// on the fp32 weight-gradient kernel (ks = 1).  Returns the number of splits through *splits_out.
int fsd_conv::wgrad_gemm_batched(const float* dy, long long dy_ld, long long dy_bs, const float* x, long long x_ld,
                                 long long x_bs, float* ws, long long rows, int cin, int cout, int batches,
                                 int* splits_out, hipStream_t stream) {
  if (rows < 1 || rows > 0x7fffffffLL - 4096 || (cin & 3) || (dy_ld & 3) || (x_ld & 3)) return FSD_ERR_UNSUPPORTED;
  WgradArgs a;
  a.dy = dy; a.x = x; a.ws = ws;
  a.dy_ld = dy_ld; a.x_ld = x_ld;
  a.H = 1; a.W = (int)rows; a.HW = (int)rows; a.M = (int)rows;
  a.Cout = cout; a.cin4 = cin; a.ks = 1; a.pad = 0;
  a.ncols = cin;
  const int tile = tile_of(0);
  a.m_tiles = (cout + tile - 1) / tile;
  a.n_tiles = (cin + tile - 1) / tile;
  const int splits = wgrad_batched_splits(rows, cin, cout, batches);
  a.pix_per_split = round_up((int)((rows + splits - 1) / splits), kBK);
  a.dy_bs = dy_bs; a.x_bs = x_bs; a.ws_bs = (long long)splits * cout * cin;
  if (int rc = launch_wgrad(a, 0, dim3(a.m_tiles * a.n_tiles, splits, batches), stream)) return rc;
  *splits_out = splits;
  return (int)hipGetLastError();
}

int fsd_conv::wgrad_batched_splits(long long rows, int cin, int cout, int batches) {
  const int tile = tile_of(0);
  const int tiles = ((cout + tile - 1) / tile) * ((cin + tile - 1) / tile) * batches;
  return pick_splits(rows, tiles);
}

extern "C" size_t fsd_conv2d_wgrad_workspace_bytes(int batch, int height, int width, int cin, int cout, int ksize) {
  // sized for the finer (fp32, 64x64) tiling, which needs the larger number of splits; valid for both modes
  const long long pixels = (long long)batch * height * width;
  const int ncols = ksize * ksize * round_up(cin, 4);
  size_t best = 0;
  for (int bf16 = 0; bf16 < 2; ++bf16) {
    const int tile = tile_of(bf16);
    const int tiles = ((cout + tile - 1) / tile) * ((ncols + tile - 1) / tile);
    const size_t need = (size_t)pick_splits(pixels, tiles) * cout * ncols * sizeof(float);
    if (need > best) best = need;
  }
  return best;
}

extern "C" int fsd_conv2d_wgrad(const float* dy, long long dy_ld, const float* x, long long x_ld, float* dw_oihw,
                                void* workspace, size_t workspace_bytes, int batch, int height, int width, int cin,
                                int cout, int ksize, hipStream_t stream) {
  return wgrad_impl(dy, dy_ld, x, x_ld, dw_oihw, workspace, workspace_bytes, batch, height, width, cin, cout, ksize, 0,
                    stream);
}

extern "C" int fsd_conv2d_wgrad_bf16(const float* dy, long long dy_ld, const float* x, long long x_ld, float* dw_oihw,
                                     void* workspace, size_t workspace_bytes, int batch, int height, int width, int cin,
                                     int cout, int ksize, hipStream_t stream) {
  return wgrad_impl(dy, dy_ld, x, x_ld, dw_oihw, workspace, workspace_bytes, batch, height, width, cin, cout, ksize, 1,
                    stream);
}
