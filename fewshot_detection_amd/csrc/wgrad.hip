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
#include "fsdet.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBK = 32;
constexpr int kThreads = 256;
// 64x64 tiles, one LDS stage (17 KB -> 8 workgroups per CU): same occupancy-first choice as the
// forward kernel (conv.hip), which measured best and most stable on MI355X.
constexpr int kBM = 64, kBN = 64;
constexpr int kStages = 1;
constexpr int kLdT = kBM + 4;        // LDS row stride (floats), keeps 16-byte alignment
constexpr int kCQ = kBM / 4;         // float4 column groups per tile row
constexpr int kRPP = kThreads / kCQ; // k rows staged per pass
constexpr int kPasses = kBK / kRPP;
constexpr int kT = kBM / 64;         // 32x32 MFMA tiles per wave along each dimension (2x2 waves)

struct WgradArgs {
  const float* dy;     // (pixels, dy_ld), columns [0, Cout) (zero padded to a multiple of 4)
  const float* x;      // (pixels, x_ld) NHWC activations that fed the convolution
  float* ws;           // [splits][Cout][ncols]
  long long dy_ld, x_ld;
  int H, W, HW, M;     // M = pixels
  int Cout, cin4, ks, pad, ncols;
  int m_tiles, n_tiles, pix_per_split;
};

__global__ __launch_bounds__(kThreads) void wgrad_kernel(WgradArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int STAGE = 2 * kBK * kLdT;          // A tile + B tile
  const int tile = blockIdx.x;
  const int mt = tile / p.n_tiles, nt = tile - mt * p.n_tiles;
  const int m0 = mt * kBM, n0 = nt * kBN;
  const int split = blockIdx.y;
  const int p_begin = split * p.pix_per_split;
  const int p_end = min(p.M, p_begin + p.pix_per_split);
  const int nk = (p_end - p_begin + kBK - 1) / kBK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int cq = tid % kCQ, kr = tid / kCQ;      // this thread's 4-channel column group / first k row

  // A: dy columns are fixed per thread
  const int a_col = m0 + cq * 4;
  const bool a_ok = a_col < p.Cout;
  // B: the im2col column (tap, ci) is fixed per thread
  const int b_col = n0 + cq * 4;
  const bool b_colok = b_col < p.ncols;
  const int tap = b_colok ? b_col / p.cin4 : 0;
  const int ci = b_col - tap * p.cin4;
  const int ky = tap / p.ks, kx = tap - ky * p.ks;
  const int dyy = ky - p.pad, dxx = kx - p.pad;
  const int shift = dyy * p.W + dxx;

  f32x4 ra[kPasses], rb[kPasses];
  unsigned okmask = 0;
  auto gload = [&](int kc) {
    okmask = 0;
#pragma unroll
    for (int j = 0; j < kPasses; ++j) {
      const int pix = p_begin + kc * kBK + kr + kRPP * j;
      const bool pv = pix < p_end;
      const bool aok = pv && a_ok;
      ra[j] = *reinterpret_cast<const f32x4*>(p.dy + (aok ? (long long)pix * p.dy_ld + a_col : 0));
      const int b = pix / p.HW;
      const int rem = pix - b * p.HW;
      const int yy = rem / p.W, xx = rem - yy * p.W;
      const bool bok = pv && b_colok && (unsigned)(yy + dyy) < (unsigned)p.H && (unsigned)(xx + dxx) < (unsigned)p.W;
      rb[j] = *reinterpret_cast<const f32x4*>(p.x + (bok ? (long long)(pix + shift) * p.x_ld + ci : 0));
      okmask |= (aok ? 1u : 0u) << j;
      okmask |= (bok ? 256u : 0u) << j;
    }
  };
  auto sstore = [&](float* st) {
#pragma unroll
    for (int j = 0; j < kPasses; ++j) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(st + (kr + kRPP * j) * kLdT + cq * 4) = (okmask >> j) & 1u ? ra[j] : z;
      *reinterpret_cast<f32x4*>(st + (kBK + kr + kRPP * j) * kLdT + cq * 4) = (okmask >> (8 + j)) & 1u ? rb[j] : z;
    }
  };

  f32x16 acc[kT][kT];
#pragma unroll
  for (int i = 0; i < kT; ++i)
#pragma unroll
    for (int j = 0; j < kT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag = (lane >> 5) * 4 * kLdT + (lane & 31);
  auto compute = [&](const float* st) {
    const float* sa = st + wm * (kT * 32) + frag;
    const float* sb = st + kBK * kLdT + wn * (kT * 32) + frag;
    // fragments one k8-step ahead of the MFMAs (statically indexed double buffer)
    float af[2][4][kT], bf[2][4][kT];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int i = 0; i < kT; ++i) af[0][kk][i] = sa[kk * kLdT + i * 32];
#pragma unroll
      for (int j = 0; j < kT; ++j) bf[0][kk][j] = sb[kk * kLdT + j * 32];
    }
#pragma unroll
    for (int k8 = 0; k8 < kBK / 8; ++k8) {
      if (k8 + 1 < kBK / 8) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
          for (int i = 0; i < kT; ++i) af[(k8 + 1) & 1][kk][i] = sa[((k8 + 1) * 8 + kk) * kLdT + i * 32];
#pragma unroll
          for (int j = 0; j < kT; ++j) bf[(k8 + 1) & 1][kk][j] = sb[((k8 + 1) * 8 + kk) * kLdT + j * 32];
        }
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < kT; ++i)
#pragma unroll
          for (int j = 0; j < kT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k8 & 1][kk][i], bf[k8 & 1][kk][j], acc[i][j], 0, 0, 0);
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
      if constexpr (kStages == 2) {
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
  for (int i = 0; i < kT; ++i)
#pragma unroll
    for (int j = 0; j < kT; ++j) {
      const int n = n0 + wn * (kT * 32) + j * 32 + c_lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (kT * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + r_lane;
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
    for (int k = ty; k < splits; k += SY) v += src[k * slice];
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

}  // namespace

extern "C" size_t fsd_conv2d_wgrad_workspace_bytes(int batch, int height, int width, int cin, int cout, int ksize) {
  const long long pixels = (long long)batch * height * width;
  const int ncols = ksize * ksize * round_up(cin, 4);
  const int tiles = ((cout + kBM - 1) / kBM) * ((ncols + kBN - 1) / kBN);
  return (size_t)pick_splits(pixels, tiles) * cout * ncols * sizeof(float);
}

extern "C" int fsd_conv2d_wgrad(const float* dy, long long dy_ld, const float* x, long long x_ld, float* dw_oihw,
                                void* workspace, size_t workspace_bytes, int batch, int height, int width, int cin,
                                int cout, int ksize, hipStream_t stream) {
  (void)hipGetLastError();   // drop a stale error left by someone else's earlier call
  if (!dy || !x || !dw_oihw || !workspace || batch < 1 || height < 1 || width < 1 || cin < 1 || cout < 1) return FSD_ERR_ARG;
  if (ksize != 1 && ksize != 3) return FSD_ERR_UNSUPPORTED;
  const int cin4 = round_up(cin, 4);
  if ((dy_ld & 3) || (x_ld & 3) || dy_ld < round_up(cout, 4) || x_ld < cin4) return FSD_ERR_ARG;
  const long long pixels = (long long)batch * height * width;
  if (pixels > 0x7fffffffLL - 4096) return FSD_ERR_UNSUPPORTED;
  if (workspace_bytes < fsd_conv2d_wgrad_workspace_bytes(batch, height, width, cin, cout, ksize)) return FSD_ERR_WORKSPACE;
  WgradArgs a;
  a.dy = dy; a.x = x; a.ws = reinterpret_cast<float*>(workspace);
  a.dy_ld = dy_ld; a.x_ld = x_ld;
  a.H = height; a.W = width; a.HW = height * width; a.M = (int)pixels;
  a.Cout = cout; a.cin4 = cin4; a.ks = ksize; a.pad = (ksize - 1) / 2;
  a.ncols = ksize * ksize * cin4;
  a.m_tiles = (cout + kBM - 1) / kBM;
  a.n_tiles = (a.ncols + kBN - 1) / kBN;
  const int splits = pick_splits(pixels, a.m_tiles * a.n_tiles);
  a.pix_per_split = round_up((int)((pixels + splits - 1) / splits), kBK);
  const size_t lds = kStages * (size_t)(2 * kBK * kLdT) * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(wgrad_kernel, dim3(a.m_tiles * a.n_tiles, splits), dim3(kThreads), lds, stream, a);
  if (splits <= 8)
    hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((a.ncols + 255) / 256, cout), dim3(256), 0, stream, a.ws, dw_oihw,
                       splits, cout, cin, cin4, ksize * ksize, a.ncols);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3((a.ncols + 31) / 32, cout), dim3(256), 0, stream, a.ws, dw_oihw,
                       splits, cout, cin, cin4, ksize * ksize, a.ncols);
  return (int)hipGetLastError();
}
