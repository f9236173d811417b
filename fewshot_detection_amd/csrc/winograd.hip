// Winograd F(2x2, 3x3) for the 3x3 / stride-1 / "same" convolutions of the path (fp32).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// 16 multiplications per 2x2 outputs instead of 36: 2.25x less matrix work on the layers that hold 93 % of
// the FLOPs.  Three launches:
//   1. wino_input_kernel   x (NHWC)  ->  V[16][tiles][Cin]      (B^T d B, zero padding handled here)
//   2. conv_gemm_batched   M[p] = V[p] * U[p]^T for the 16 positions p  (the fp32 MFMA kernel of conv.hip,
//                          grid.y = 16; U = G g G^T is packed once per weight update)
//   3. wino_output_kernel  M -> y (NHWC, + bias) and the per-block BatchNorm partial sums (sum y, sum y^2)
// The data gradient is the same pipeline with the filter rotated by 180 degrees and channels swapped (mode 1).
// Transforms are pure adds in fp32; the result differs from the direct convolution by ~1e-6 relative.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include "fsdet.h"
#include "conv_common.hpp"
#include "profile.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kTilesPerBlock = 64;     // tiles reduced by one block of the output transform

// Stride (elements) between the (tile+2)^2 position matrices of a transformed operand V / M / Wt: the matrix itself plus a
// pad.  With the bare T * C stride the 36 streams a transform thread reads or writes are a multiple of 64 KB apart on the big
// layers (43264 tiles x 128 channels x 4 B = 338 x 64 KB) -- FSD_WINO_PAD elements (a build-time constant) break that up.
#ifndef FSD_WINO_PAD
#define FSD_WINO_PAD 0
#endif
// Minimum waves per SIMD of the F(4x4) output / gradient transforms (build-time, tuning aid; 4 = at most 128 VGPRs): left alone the compiler takes 242 / 130
// VGPRs for them (every load of a tile in flight at once) -- two waves per SIMD, and no room beside an 8-wave GEMM workgroup of
// another stream (2 x 200 of the 512 registers of a SIMD lane: 112 left).
#ifdef FSD_XFORM_WAVES
#define FSD_XFORM_LB __launch_bounds__(256, FSD_XFORM_WAVES)
#else
#define FSD_XFORM_LB __launch_bounds__(256)
#endif
__host__ __device__ inline long long pos_stride(long long T, int C) { return T * C + FSD_WINO_PAD; }

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// One thread: one tile x 4 channels.
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, long long x_ld, float* __restrict__ V,
                                                        int H, int W, int TH, int TW, int C, long long T) {
  const int cg = C >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * cg) return;
  const unsigned uidx = (unsigned)idx;                 // < 2^32 (launcher check): 32-bit divisions, not 64-bit ones
  const int g = (int)(uidx % (unsigned)cg);
  const unsigned utile = uidx / (unsigned)cg;
  const long long tile = utile;
  const int tx = (int)(utile % (unsigned)TW);
  const unsigned ut2 = utile / (unsigned)TW;
  const int ty = (int)(ut2 % (unsigned)TH);
  const long long b = ut2 / (unsigned)TH;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 d[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int iy = 2 * ty - 1 + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ix = 2 * tx - 1 + j;
      const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      d[i][j] = ok ? ld4(x + ((b * H + iy) * (long long)W + ix) * x_ld + g * 4) : zero;
    }
  }
  // t = B^T d   (rows), then V = t B (columns);  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
  f32x4 t[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[0][j] = d[0][j] - d[2][j];
    t[1][j] = d[1][j] + d[2][j];
    t[2][j] = d[2][j] - d[1][j];
    t[3][j] = d[1][j] - d[3][j];
  }
  float* dst = V + tile * C + g * 4;
  const long long ps = pos_stride(T, C);      // stride between the 16 position matrices
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    st4(dst + (i * 4 + 0) * ps, t[i][0] - t[i][2]);
    st4(dst + (i * 4 + 1) * ps, t[i][1] + t[i][2]);
    st4(dst + (i * 4 + 2) * ps, t[i][2] - t[i][1]);
    st4(dst + (i * 4 + 3) * ps, t[i][1] - t[i][3]);
  }
}

// leaky activation of (output + bias) for the inference entry point (slope 1 = linear)
__device__ __forceinline__ f32x4 act4(f32x4 v, float slope) {
  if (slope != 1.f) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * slope;
  }
  return v;
}

// One block: 64 channel groups x 4 tile lanes over `tpb` tiles; writes y and the BN partial sums.
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mb, const float* __restrict__ bias,
                                                         float* __restrict__ y, long long y_ld, float* __restrict__ partial,
                                                         int H, int W, int TH, int TW, int C, long long T, int tpb, float slope) {
  __shared__ float s_red[4][64][8];
  const int gl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int g = blockIdx.y * 64 + gl;
  const int cg = C >> 2;
  const bool g_ok = g < cg;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 bv = (g_ok && bias) ? ld4(bias + g * 4) : zero;
  f32x4 s1 = zero, s2 = zero;
  const long long t0 = (long long)blockIdx.x * tpb;
  const long long ps = pos_stride(T, C);
  if (g_ok) {
    for (int it = pl; it < tpb; it += 4) {
      const long long tile = t0 + it;
      if (tile >= T) break;
      const unsigned utile = (unsigned)tile;             // < 2^31: 32-bit divisions
      const int tx = (int)(utile % (unsigned)TW);
      const unsigned ut2 = utile / (unsigned)TW;
      const int ty = (int)(ut2 % (unsigned)TH);
      const long long b = ut2 / (unsigned)TH;
      const float* src = Mb + tile * C + g * 4;
      f32x4 m[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[i][j] = ld4(src + (i * 4 + j) * ps);
      // s = A^T m (rows), Y = s A (columns);  A^T = [1 1 1 0; 0 1 -1 -1]
      f32x4 s[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[0][j] = m[0][j] + m[1][j] + m[2][j];
        s[1][j] = m[1][j] - m[2][j] - m[3][j];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int oy = 2 * ty + i;
        if (oy >= H) continue;
        const f32x4 o0 = s[i][0] + s[i][1] + s[i][2];
        const f32x4 o1 = s[i][1] - s[i][2] - s[i][3];
        const int ox = 2 * tx;
        float* dst = y + ((b * H + oy) * (long long)W + ox) * y_ld + g * 4;
        st4(dst, act4(o0 + bv, slope));
#pragma unroll
        for (int k = 0; k < 4; ++k) { s1[k] += o0[k]; s2[k] += o0[k] * o0[k]; }
        if (ox + 1 < W) {
          st4(dst + y_ld, act4(o1 + bv, slope));
#pragma unroll
          for (int k = 0; k < 4; ++k) { s1[k] += o1[k]; s2[k] += o1[k] * o1[k]; }
        }
      }
    }
  }
  if (partial == nullptr) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    s_red[pl][gl][k] = s1[k];
    s_red[pl][gl][4 + k] = s2[k];
  }
  __syncthreads();
  if (pl == 0 && g_ok) {
    float* dst = partial + ((long long)blockIdx.x * C + g * 4) * 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      dst[2 * k] = s_red[0][gl][k] + s_red[1][gl][k] + s_red[2][gl][k] + s_red[3][gl][k];
      dst[2 * k + 1] = s_red[0][gl][4 + k] + s_red[1][gl][4 + k] + s_red[2][gl][4 + k] + s_red[3][gl][4 + k];
    }
  }
}

// U[p][row][k] = (G g G^T)[p],  g = w[row][k] (mode 0) or the rotated / channel-swapped filter (mode 1);
// written in the packed layout of the GEMM kernel: [16][rows_pad][red] (rows_pad = round_up(rows, 128)).
__global__ void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int cout, int cin, int mode,
                                   int rows, int red, int rows_pad) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)rows_pad * red) return;
  const int row = (int)(idx / red), k = (int)(idx - (long long)row * red);
  float gk[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      float v = 0.f;
      if (row < rows) {
        v = mode == 0 ? w[(((long long)row * cin + k) * 3 + a) * 3 + b]
                      : w[(((long long)k * cin + row) * 3 + (2 - a)) * 3 + (2 - b)];
      }
      gk[a][b] = v;
    }
  // t = G g : rows [g0; (g0+g1+g2)/2; (g0-g1+g2)/2; g2]
  float t[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    t[0][b] = gk[0][b];
    t[1][b] = 0.5f * (gk[0][b] + gk[1][b] + gk[2][b]);
    t[2][b] = 0.5f * (gk[0][b] - gk[1][b] + gk[2][b]);
    t[3][b] = gk[2][b];
  }
  const long long ps = (long long)rows_pad * red;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    U[(a * 4 + 0) * ps + idx] = t[a][0];
    U[(a * 4 + 1) * ps + idx] = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
    U[(a * 4 + 2) * ps + idx] = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
    U[(a * 4 + 3) * ps + idx] = t[a][2];
  }
}

// Weight gradient, F(3x3, 2x2):  dW = A3^T [ (G2 dy G2^T) (.) (B^T d B) ] A3  summed over tiles.
// The input transform B^T d B is the forward one; the sign difference between the F(2,3) and F(3,2)
// B matrices is folded into G2 = [1 0; .5 .5; .5 -.5; 0 -1].
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* __restrict__ dy, long long dy_ld, float* __restrict__ Wt,
                                                     int H, int W, int TH, int TW, int C, long long T) {
  const int cg = C >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * cg) return;
  const unsigned uidx = (unsigned)idx;                 // < 2^32 (launcher check): 32-bit divisions, not 64-bit ones
  const int g = (int)(uidx % (unsigned)cg);
  const unsigned utile = uidx / (unsigned)cg;
  const long long tile = utile;
  const int tx = (int)(utile % (unsigned)TW);
  const unsigned ut2 = utile / (unsigned)TW;
  const int ty = (int)(ut2 % (unsigned)TH);
  const long long b = ut2 / (unsigned)TH;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 q[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int oy = 2 * ty + i, ox = 2 * tx + j;
      q[i][j] = (oy < H && ox < W) ? ld4(dy + ((b * H + oy) * (long long)W + ox) * dy_ld + g * 4) : zero;
    }
  f32x4 t[4][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    t[0][j] = q[0][j];
    t[1][j] = 0.5f * (q[0][j] + q[1][j]);
    t[2][j] = 0.5f * (q[0][j] - q[1][j]);
    t[3][j] = -q[1][j];
  }
  float* dst = Wt + tile * C + g * 4;
  const long long ps = pos_stride(T, C);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    st4(dst + (i * 4 + 0) * ps, t[i][0]);
    st4(dst + (i * 4 + 1) * ps, 0.5f * (t[i][0] + t[i][1]));
    st4(dst + (i * 4 + 2) * ps, 0.5f * (t[i][0] - t[i][1]));
    st4(dst + (i * 4 + 3) * ps, -t[i][1]);
  }
}

// ws[p][split][co][ci] -> dw[co][ci][3][3] = A3^T m A3,  A3^T = [1 1 1 0; 0 1 -1 0; 0 1 1 1].
// Block = 16 outputs x 16 positions: every thread folds the splits of one position (fixed order), then the
// first 16 threads apply the transform.
__global__ __launch_bounds__(256) void wino_dw_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits,
                                                     int cout, int cin) {
  __shared__ float s_m[16][17];
  const int il = threadIdx.x & 15, pp = threadIdx.x >> 4;
  const long long n = (long long)cout * cin;
  const long long idx = (long long)blockIdx.x * 16 + il;      // over (co, ci), ci fastest
  float v = 0.f;
  if (idx < n) {
    const float* src = ws + (long long)pp * splits * n + idx;
    int k = 0;
    for (; k + 3 < splits; k += 4) {
      const float v0 = src[k * n], v1 = src[(k + 1) * n], v2 = src[(k + 2) * n], v3 = src[(k + 3) * n];
      v += v0; v += v1; v += v2; v += v3;
    }
    for (; k < splits; ++k) v += src[k * n];
  }
  s_m[pp][il] = v;
  __syncthreads();
  if (threadIdx.x >= 16 || idx >= n) return;
  float m[4][4];
#pragma unroll
  for (int p = 0; p < 16; ++p) m[p >> 2][p & 3] = s_m[p][il];
  float s[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s[0][j] = m[0][j] + m[1][j] + m[2][j];
    s[1][j] = m[1][j] - m[2][j];
    s[2][j] = m[1][j] + m[2][j] + m[3][j];
  }
  float* dst = dw + idx * 9;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    dst[i * 3 + 0] = s[i][0] + s[i][1] + s[i][2];
    dst[i * 3 + 1] = s[i][1] - s[i][2];
    dst[i * 3 + 2] = s[i][1] + s[i][2] + s[i][3];
  }
}

// =====================================================================================================
// F(4x4, 3x3): 36 multiplications per 4x4 outputs (4x fewer than direct, 1.78x fewer than F(2x2,3x3)) and a
// transformed input of only 2.25x the activation (F(2x2): 4x).  Interpolation points 0, +-1, +-2, inf:
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// Weight gradient F(3x3, 4x4) shares B^T:  G4 = [1/4 0 0 0; -1/6(1 1 1 1); -1/6(1 -1 1 -1); 1/24(1 2 4 8);
//   1/24(1 -2 4 -8); 0 0 0 1],  A3^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 1].
// fp32 round-off of this variant is ~1.5e-5 of the output magnitude (F(2x2): ~1e-6), see DESIGN.md.
// a * b + c with ONE rounding, spelled out (vector types): the contraction of `a * b + c` is the compiler's choice per
// instantiation, and the two instantiations of the input transform (plain / activation on load) must produce the same bits
template <typename T>
__device__ __forceinline__ T fma_s(float a, const T& b, const T& c) { return __builtin_elementwise_fma((T)(a), b, c); }

template <typename T>
__device__ __forceinline__ void bt6(const T& d0, const T& d1, const T& d2, const T& d3, const T& d4, const T& d5, T (&r)[6]) {
  r[0] = fma_s(4.f, d0, fma_s(-5.f, d2, d4));
  r[1] = fma_s(-4.f, d1 + d2, d4 + d3);
  r[2] = fma_s(4.f, d1 - d2, d4 - d3);
  r[3] = fma_s(2.f, d3 - d1, d4 - d2);
  r[4] = fma_s(-2.f, d3 - d1, d4 - d2);
  r[5] = fma_s(4.f, d1, fma_s(-5.f, d3, d5));
}
template <typename T>
__device__ __forceinline__ void at4(const T& m0, const T& m1, const T& m2, const T& m3, const T& m4, const T& m5, T (&o)[4]) {
  const T a = m1 + m2, b = m1 - m2, c = m3 + m4, d = m3 - m4;
  o[0] = m0 + a + c;
  o[1] = b + 2.f * d;
  o[2] = a + 4.f * c;
  o[3] = b + 8.f * d + m5;
}
template <typename T>
__device__ __forceinline__ void g6(const T& g0, const T& g1, const T& g2, T (&r)[6]) {
  const T e = g0 + g2, f = g0 + 4.f * g2;
  r[0] = 0.25f * g0;
  r[1] = (-1.f / 6.f) * (e + g1);
  r[2] = (-1.f / 6.f) * (e - g1);
  r[3] = (1.f / 24.f) * (f + 2.f * g1);
  r[4] = (1.f / 24.f) * (f - 2.f * g1);
  r[5] = g2;
}
template <typename T>
__device__ __forceinline__ void g6x4(const T& q0, const T& q1, const T& q2, const T& q3, T (&r)[6]) {
  const T e = q0 + q2, o = q1 + q3, f = q0 + 4.f * q2, h = 2.f * q1 + 8.f * q3;
  r[0] = 0.25f * q0;
  r[1] = (-1.f / 6.f) * (e + o);
  r[2] = (-1.f / 6.f) * (e - o);
  r[3] = (1.f / 24.f) * (f + h);
  r[4] = (1.f / 24.f) * (f - h);
  r[5] = q3;
}
template <typename T>
__device__ __forceinline__ void a3(const T& m0, const T& m1, const T& m2, const T& m3, const T& m4, const T& m5, T (&o)[3]) {
  const T a = m1 + m2, b = m1 - m2, c = m3 + m4, d = m3 - m4;
  o[0] = m0 + a + c;
  o[1] = b + 2.f * d;
  o[2] = a + 4.f * c + m5;
}

// One thread: one 4x4-output tile (6x6 patch) x 2 channels (36 live values: two channels keep the kernel at
// ~90 VGPRs; with four it needs 166 and runs at less than half the HBM rate).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 ld2(const float* p) { return *reinterpret_cast<const f32x2*>(p); }
__device__ __forceinline__ void st2(float* p, f32x2 v) { *reinterpret_cast<f32x2*>(p) = v; }

// ACT: x is the raw output of the producing convolution; leaky(x * in_scale + in_shift) is formed on load (padding stays 0)
template <bool ACT>
__global__ __launch_bounds__(256) void wino4_input_kernel(const float* __restrict__ x, long long x_ld, float* __restrict__ V,
                                                         int H, int W, int TH, int TW, int C, long long T,
                                                         const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                         float in_slope, long long t_base, long long t_count) {
  // tiles [t_base, t_base + t_count) of the T tiles of the launch's tensor (a slab, see fsd_wino_conv3x3_fwd_ex)
  const int cg = C >> 1;
  // neighbouring tiles re-read 2 of their 6 patch rows/columns: keep runs of consecutive tiles on one XCD (own L2)
  const long long idx = (long long)fsd_conv::xcd_swizzle((int)blockIdx.x, (int)gridDim.x) * blockDim.x + threadIdx.x;
  if (idx >= t_count * cg) return;
  const unsigned uidx = (unsigned)idx;                 // < 2^32 (launcher check): 32-bit divisions, not 64-bit ones
  const int g = (int)(uidx % (unsigned)cg);
  const unsigned utile = uidx / (unsigned)cg + (unsigned)t_base;
  const long long tile = utile;
  const int tx = (int)(utile % (unsigned)TW);
  const unsigned ut2 = utile / (unsigned)TW;
  const int ty = (int)(ut2 % (unsigned)TH);
  const long long b = ut2 / (unsigned)TH;
  const f32x2 zero = {0.f, 0.f};
  f32x2 isc = {1.f, 1.f}, ish = zero;
  if constexpr (ACT) { isc = ld2(in_scale + g * 2); ish = ld2(in_shift + g * 2); }
  f32x2 d[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int iy = 4 * ty - 1 + i;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int ix = 4 * tx - 1 + j;
      const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      f32x2 v = ok ? ld2(x + ((b * H + iy) * (long long)W + ix) * x_ld + g * 2) : zero;
      if constexpr (ACT) {
        if (ok) {
#pragma unroll
          for (int k = 0; k < 2; ++k) {                    // the expression of bn_act_pool_kernel (elementwise.hip)
            const float t = __builtin_fmaf(v[k], isc[k], ish[k]);
            v[k] = t > 0.f ? t : t * in_slope;
          }
        }
      }
      d[i][j] = v;
    }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {            // columns: t = B^T d (in place)
    f32x2 r[6];
    bt6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], r);
#pragma unroll
    for (int i = 0; i < 6; ++i) d[i][j] = r[i];
  }
  float* dst = V + tile * C + g * 2;
  const long long ps = pos_stride(T, C);
#pragma unroll
  for (int i = 0; i < 6; ++i) {            // rows: V = t B
    f32x2 r[6];
    bt6(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5], r);
#pragma unroll
    for (int j = 0; j < 6; ++j) st2(dst + (i * 6 + j) * ps, r[j]);
  }
}

// vector of CPT channels per thread (2: 8-byte accesses and half the registers; 4: 16-byte accesses)
template <int CPT> struct VecOf;
template <> struct VecOf<4> {
  typedef f32x4 type;
  static __device__ __forceinline__ f32x4 splat(float v) { return f32x4{v, v, v, v}; }
};
template <> struct VecOf<2> {
  typedef f32x2 type;
  static __device__ __forceinline__ f32x2 splat(float v) { return f32x2{v, v}; }
};
template <int CPT> __device__ __forceinline__ typename VecOf<CPT>::type ldv(const float* p) {
  return *reinterpret_cast<const typename VecOf<CPT>::type*>(p);
}
template <int CPT> __device__ __forceinline__ void stv(float* p, typename VecOf<CPT>::type v) {
  *reinterpret_cast<typename VecOf<CPT>::type*>(p) = v;
}

// dy (4x4 tile) -> G4 dy G4^T  (36 positions)
// BN = 1: `dy` holds dt (gradient w.r.t. the BatchNorm output); the BatchNorm backward c1*(dt - c2 - xhat*c3) is
// applied on the way in and written back IN PLACE (tiles do not overlap), so fsd_bn_bwd_apply's separate pass -- and
// one read of its result -- disappear; the data-gradient transform then reads the finished dy as usual.
// BN = 2: the first backward pass only took the statistics (dt never written): dt is formed here from the gradient of the
// block output `gg.dz` (+ `gg.dz_full`) through the 2x2 / stride-2 maxpool (gg.pool == 1; a 4x4 tile is 2x2 whole pooling
// cells) and the leaky activation -- the expressions of act_bwd_pool2_kernel / act_bwd_kernel -- and dy is WRITTEN to `dy`.
struct DyFromG {
  const float* dz;          // gradient of the (pooled) block output (B, OH, OW, C)
  const float* dz_full;     // optional gradient of the un-pooled activation
  const float* scale;       // BatchNorm affine
  const float* shift;
  long long dz_ld, dzf_ld;
  float slope;
  int pool, OH, OW;
};

template <int BN, int CPT>
__global__ FSD_XFORM_LB void wino4_dy_kernel(float* __restrict__ dy, long long dy_ld, float* __restrict__ Wt,
                                                      int H, int W, int TH, int TW, int C, long long T,
                                                      const float* __restrict__ y, long long y_ld,
                                                      const float* __restrict__ coef, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, DyFromG gg) {
  typedef typename VecOf<CPT>::type VT;
  const int cg = C / CPT;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * cg) return;
  const unsigned uidx = (unsigned)idx;                 // < 2^32 (launcher check): 32-bit divisions, not 64-bit ones
  const int g = (int)(uidx % (unsigned)cg);
  const unsigned utile = uidx / (unsigned)cg;
  const long long tile = utile;
  const int tx = (int)(utile % (unsigned)TW);
  const unsigned ut2 = utile / (unsigned)TW;
  const int ty = (int)(ut2 % (unsigned)TH);
  const long long b = ut2 / (unsigned)TH;
  const VT zero = VecOf<CPT>::splat(0.f);
  VT c1 = zero, c2 = zero, c3 = zero, mu = zero, is = zero;
  if constexpr (BN != 0) {
    c1 = ldv<CPT>(coef + g * CPT); c2 = ldv<CPT>(coef + C + g * CPT); c3 = ldv<CPT>(coef + 2 * C + g * CPT);
    mu = ldv<CPT>(mean + g * CPT); is = ldv<CPT>(invstd + g * CPT);
  }
  VT q[4][4];
  if constexpr (BN == 2) {
    const VT one = VecOf<CPT>::splat(1.f);
    const VT sc = gg.scale ? ldv<CPT>(gg.scale + g * CPT) : one, sh = gg.shift ? ldv<CPT>(gg.shift + g * CPT) : zero;
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int cj = 0; cj < 2; ++cj) {
        const int cy = 2 * ty + ci, cx = 2 * tx + cj;            // pooling cell of the image
        VT yv[4], tv[4];
        bool in[4];
        int best[CPT] = {};
        VT bv = zero;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int oy = 2 * cy + (w >> 1), ox = 2 * cx + (w & 1);
          in[w] = oy < H && ox < W;
          yv[w] = in[w] ? ldv<CPT>(y + ((b * H + oy) * (long long)W + ox) * y_ld + g * CPT) : zero;
#pragma unroll
          for (int k = 0; k < CPT; ++k) {
            tv[w][k] = __builtin_fmaf(yv[w][k], sc[k], sh[k]);   // one rounding, like the forward pass that picked the sign / the pool winner
            const float a = tv[w][k] > 0.f ? tv[w][k] : tv[w][k] * gg.slope;
            if (w == 0 || a > bv[k]) { bv[k] = a; best[k] = w; }
          }
        }
        const bool win = gg.pool == 1 && cy < gg.OH && cx < gg.OW;
        const VT gz = win ? ldv<CPT>(gg.dz + ((b * gg.OH + cy) * (long long)gg.OW + cx) * gg.dz_ld + g * CPT) : zero;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int i = 2 * ci + (w >> 1), j = 2 * cj + (w & 1);
          if (!in[w]) { q[i][j] = zero; continue; }
          const long long pix = (b * H + (4 * ty + i)) * (long long)W + 4 * tx + j;
          VT gin = gg.pool == 0 ? ldv<CPT>(gg.dz + pix * gg.dz_ld + g * CPT) : zero;
          if (gg.dz_full) {
            const VT gf = ldv<CPT>(gg.dz_full + pix * gg.dzf_ld + g * CPT);
#pragma unroll
            for (int k = 0; k < CPT; ++k) gin[k] += gf[k];
          }
          VT v;
#pragma unroll
          for (int k = 0; k < CPT; ++k) {
            if (win && best[k] == w) gin[k] += gz[k];
            const float d = tv[w][k] > 0.f ? gin[k] : gin[k] * gg.slope;
            v[k] = c1[k] * (d - c2[k] - (yv[w][k] - mu[k]) * is[k] * c3[k]);
          }
          stv<CPT>(dy + pix * dy_ld + g * CPT, v);
          q[i][j] = v;
        }
      }
  } else {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int oy = 4 * ty + i, ox = 4 * tx + j;
      if (oy < H && ox < W) {
        const long long pix = (b * H + oy) * (long long)W + ox;
        VT v = ldv<CPT>(dy + pix * dy_ld + g * CPT);
        if constexpr (BN == 1) {
          const VT yv = ldv<CPT>(y + pix * y_ld + g * CPT);
#pragma unroll
          for (int k = 0; k < CPT; ++k) v[k] = c1[k] * (v[k] - c2[k] - (yv[k] - mu[k]) * is[k] * c3[k]);   // == bn_bwd_apply
          stv<CPT>(dy + pix * dy_ld + g * CPT, v);
        }
        q[i][j] = v;
      } else {
        q[i][j] = zero;
      }
    }
  }
  VT t[6][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    VT r[6];
    g6x4(q[0][j], q[1][j], q[2][j], q[3][j], r);
#pragma unroll
    for (int i = 0; i < 6; ++i) t[i][j] = r[i];
  }
  float* dst = Wt + tile * C + g * CPT;
  const long long ps = pos_stride(T, C);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    VT r[6];
    g6x4(t[i][0], t[i][1], t[i][2], t[i][3], r);
#pragma unroll
    for (int j = 0; j < 6; ++j) stv<CPT>(dst + (i * 6 + j) * ps, r[j]);
  }
}

// Backward of a Winograd layer in ONE pass over the gradient: the BatchNorm backward dy = c1*(dt - c2 - xhat*c3)
// (bn_bwd_apply) is formed in registers for the 6x6 patch, then BOTH transforms the layer's gradients need are written:
//   Vd = B^T dy B            (input of the data-gradient GEMMs; patch = the 4x4 tile + 1 pixel of padding all round)
//   Wt = G4 dy4 G4^T         (dy4 = the inner 4x4 of the same patch; operand of the weight-gradient GEMMs)
// dt and y are read once (+ the patch overlap from L2) instead of three elementwise passes and two transform reads.
__global__ __launch_bounds__(256) void wino4_grad_kernel(const float* __restrict__ dt, long long dt_ld,
                                                        const float* __restrict__ y, long long y_ld,
                                                        const float* __restrict__ coef, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, float* __restrict__ Vd,
                                                        float* __restrict__ Wt, int H, int W, int TH, int TW, int C,
                                                        long long T) {
  const int cg = C >> 1;
  const long long idx = (long long)fsd_conv::xcd_swizzle((int)blockIdx.x, (int)gridDim.x) * blockDim.x + threadIdx.x;
  if (idx >= T * cg) return;
  const unsigned uidx = (unsigned)idx;                 // < 2^32 (launcher check): 32-bit divisions, not 64-bit ones
  const int g = (int)(uidx % (unsigned)cg);
  const unsigned utile = uidx / (unsigned)cg;
  const long long tile = utile;
  const int tx = (int)(utile % (unsigned)TW);
  const unsigned ut2 = utile / (unsigned)TW;
  const int ty = (int)(ut2 % (unsigned)TH);
  const long long b = ut2 / (unsigned)TH;
  const f32x2 zero = {0.f, 0.f};
  const f32x2 c1 = ld2(coef + g * 2), c2 = ld2(coef + C + g * 2), c3 = ld2(coef + 2 * C + g * 2);
  const f32x2 mu = ld2(mean + g * 2), is = ld2(invstd + g * 2);
  f32x2 d[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int iy = 4 * ty - 1 + i;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int ix = 4 * tx - 1 + j;
      const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      if (ok) {
        const long long pix = (b * H + iy) * (long long)W + ix;
        const f32x2 tv = ld2(dt + pix * dt_ld + g * 2), yv = ld2(y + pix * y_ld + g * 2);
        d[i][j] = c1 * (tv - c2 - (yv - mu) * is * c3);
      } else {
        d[i][j] = zero;
      }
    }
  }
  const long long ps = pos_stride(T, C);
  {   // weight-gradient operand from the inner 4x4
    f32x2 t[6][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x2 r[6];
      g6x4(d[1][j + 1], d[2][j + 1], d[3][j + 1], d[4][j + 1], r);
#pragma unroll
      for (int i = 0; i < 6; ++i) t[i][j] = r[i];
    }
    float* dst = Wt + tile * C + g * 2;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      f32x2 r[6];
      g6x4(t[i][0], t[i][1], t[i][2], t[i][3], r);
#pragma unroll
      for (int j = 0; j < 6; ++j) st2(dst + (i * 6 + j) * ps, r[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {            // data-gradient operand: columns, then rows
    f32x2 r[6];
    bt6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], r);
#pragma unroll
    for (int i = 0; i < 6; ++i) d[i][j] = r[i];
  }
  float* dst = Vd + tile * C + g * 2;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    f32x2 r[6];
    bt6(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5], r);
#pragma unroll
    for (int j = 0; j < 6; ++j) st2(dst + (i * 6 + j) * ps, r[j]);
  }
}

// One block: GL channel pairs x 256/GL tile lanes over `tpb` tiles; writes y (+bias) and the BN partial sums.
// The 6 rows of m are streamed: row r is transformed along its columns (u = A^T-row-pass) and accumulated into the
// 4x4 output with the column weights A^T[:, r] = (1,0,0,0) (1,1,1,1) (1,-1,1,-1) (1,2,4,8) (1,-2,4,-8) (0,0,0,1),
// so only 16 outputs + one row are live (the all-at-once form needs 256 VGPRs and runs at occupancy 1).
template <int GL>
__global__ FSD_XFORM_LB void wino4_output_kernel(const float* __restrict__ Mb, const float* __restrict__ bias,
                                                          float* __restrict__ y, long long y_ld, float* __restrict__ partial,
                                                          int H, int W, int TH, int TW, int C, long long T, int tpb, float slope) {
  constexpr int NPL = 256 / GL;
  __shared__ float s_red[NPL][GL][4];
  const int gl = threadIdx.x % GL, pl = threadIdx.x / GL;
  const int g = blockIdx.y * GL + gl;
  const int cg = C >> 1;
  const bool g_ok = g < cg;
  const f32x2 zero = {0.f, 0.f};
  const f32x2 bv = (g_ok && bias) ? ld2(bias + g * 2) : zero;
  f32x2 s1 = zero, s2 = zero;
  const long long t0 = (long long)blockIdx.x * tpb;
  const long long ps = pos_stride(T, C);
  if (g_ok) {
    for (int it = pl; it < tpb; it += NPL) {
      const long long tile = t0 + it;
      if (tile >= T) break;
      const unsigned utile = (unsigned)tile;             // < 2^31: 32-bit divisions
      const int tx = (int)(utile % (unsigned)TW);
      const unsigned ut2 = utile / (unsigned)TW;
      const int ty = (int)(ut2 % (unsigned)TH);
      const long long b = ut2 / (unsigned)TH;
      const float* src = Mb + tile * C + g * 2;
      f32x2 o[4][4];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        f32x2 u[4];
        at4(ld2(src + (r * 6 + 0) * ps), ld2(src + (r * 6 + 1) * ps), ld2(src + (r * 6 + 2) * ps),
            ld2(src + (r * 6 + 3) * ps), ld2(src + (r * 6 + 4) * ps), ld2(src + (r * 6 + 5) * ps), u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (r == 0) { o[0][j] = u[j]; }
          else if (r == 1) { o[0][j] += u[j]; o[1][j] = u[j]; o[2][j] = u[j]; o[3][j] = u[j]; }
          else if (r == 2) { o[0][j] += u[j]; o[1][j] -= u[j]; o[2][j] += u[j]; o[3][j] -= u[j]; }
          else if (r == 3) { o[0][j] += u[j]; o[1][j] += 2.f * u[j]; o[2][j] += 4.f * u[j]; o[3][j] += 8.f * u[j]; }
          else if (r == 4) { o[0][j] += u[j]; o[1][j] -= 2.f * u[j]; o[2][j] += 4.f * u[j]; o[3][j] -= 8.f * u[j]; }
          else { o[3][j] += u[j]; }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int oy = 4 * ty + i;
        if (oy >= H) continue;
        float* dst = y + ((b * H + oy) * (long long)W + 4 * tx) * y_ld + g * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (4 * tx + j >= W) continue;
          f32x2 v = o[i][j] + bv;
          if (slope != 1.f) { v[0] = v[0] > 0.f ? v[0] : v[0] * slope; v[1] = v[1] > 0.f ? v[1] : v[1] * slope; }
          st2(dst + j * y_ld, v);
          s1 += o[i][j];
          s2 += o[i][j] * o[i][j];
        }
      }
    }
  }
  if (partial == nullptr) return;
  s_red[pl][gl][0] = s1[0]; s_red[pl][gl][1] = s1[1];
  s_red[pl][gl][2] = s2[0]; s_red[pl][gl][3] = s2[1];
  __syncthreads();
  if (pl == 0 && g_ok) {
    float* dst = partial + ((long long)blockIdx.x * C + g * 2) * 2;
    float a0 = 0.f, a1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int l = 0; l < NPL; ++l) { a0 += s_red[l][gl][0]; a1 += s_red[l][gl][1]; q0 += s_red[l][gl][2]; q1 += s_red[l][gl][3]; }
    dst[0] = a0; dst[1] = q0; dst[2] = a1; dst[3] = q1;
  }
}

// The same with FOUR channels per thread (16-byte loads of M and 16-byte stores of y; 64 accumulator registers): for layers
// with >= 128 output channels.  FSD_WINO_OUT4=0 keeps the two-channel kernel everywhere.
// KS: the position GEMMs were cut into `ks` K slices (fsd_conv::batched_ksplit), slice s lies `ss` floats behind slice 0: added
// on load, in slice order.
template <int GL, bool KS = false>
__global__ FSD_XFORM_LB void wino4_output4_kernel(const float* __restrict__ Mb, const float* __restrict__ bias,
                                                           float* __restrict__ y, long long y_ld, float* __restrict__ partial,
                                                           int H, int W, int TH, int TW, int C, long long T, int tpb, float slope,
                                                           int ks, long long ss, long long m_ps, long long t_base) {
  // Mb: the position matrices of tiles [t_base, T) (position stride m_ps), blockIdx.x counts blocks of tpb tiles from t_base;
  // `partial` points at this launch's first row
  constexpr int NPL = 256 / GL;
  __shared__ float s_red[NPL][GL][8];
  const int gl = threadIdx.x % GL, pl = threadIdx.x / GL;
  const int g = blockIdx.y * GL + gl;
  const int cg = C >> 2;
  const bool g_ok = g < cg;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 bv = (g_ok && bias) ? ld4(bias + g * 4) : zero;
  f32x4 s1 = zero, s2 = zero;
  const long long t0 = t_base + (long long)blockIdx.x * tpb;
  const long long ps = m_ps;
  if (g_ok) {
    for (int it = pl; it < tpb; it += NPL) {
      const long long tile = t0 + it;
      if (tile >= T) break;
      const unsigned utile = (unsigned)tile;             // < 2^31: 32-bit divisions
      const int tx = (int)(utile % (unsigned)TW);
      const unsigned ut2 = utile / (unsigned)TW;
      const int ty = (int)(ut2 % (unsigned)TH);
      const long long b = ut2 / (unsigned)TH;
      const float* src = Mb + (tile - t_base) * C + g * 4;
      auto ldm = [&](const float* q) -> f32x4 {
        f32x4 v = ld4(q);
        if constexpr (KS)
          for (int sl = 1; sl < ks; ++sl) v += ld4(q + sl * ss);
        return v;
      };
      f32x4 o[4][4];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        f32x4 u[4];
        at4(ldm(src + (r * 6 + 0) * ps), ldm(src + (r * 6 + 1) * ps), ldm(src + (r * 6 + 2) * ps),
            ldm(src + (r * 6 + 3) * ps), ldm(src + (r * 6 + 4) * ps), ldm(src + (r * 6 + 5) * ps), u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (r == 0) { o[0][j] = u[j]; }
          else if (r == 1) { o[0][j] += u[j]; o[1][j] = u[j]; o[2][j] = u[j]; o[3][j] = u[j]; }
          else if (r == 2) { o[0][j] += u[j]; o[1][j] -= u[j]; o[2][j] += u[j]; o[3][j] -= u[j]; }
          else if (r == 3) { o[0][j] += u[j]; o[1][j] += 2.f * u[j]; o[2][j] += 4.f * u[j]; o[3][j] += 8.f * u[j]; }
          else if (r == 4) { o[0][j] += u[j]; o[1][j] -= 2.f * u[j]; o[2][j] += 4.f * u[j]; o[3][j] -= 8.f * u[j]; }
          else { o[3][j] += u[j]; }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int oy = 4 * ty + i;
        if (oy >= H) continue;
        float* dst = y + ((b * H + oy) * (long long)W + 4 * tx) * y_ld + g * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (4 * tx + j >= W) continue;
          f32x4 v = o[i][j] + bv;
          if (slope != 1.f) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * slope;
          }
          st4(dst + j * y_ld, v);
          s1 += o[i][j];
          s2 += o[i][j] * o[i][j];
        }
      }
    }
  }
  if (partial == nullptr) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) { s_red[pl][gl][k] = s1[k]; s_red[pl][gl][4 + k] = s2[k]; }
  __syncthreads();
  if (pl == 0 && g_ok) {
    float* dst = partial + ((long long)blockIdx.x * C + g * 4) * 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int l = 0; l < NPL; ++l) { a += s_red[l][gl][k]; q += s_red[l][gl][4 + k]; }
      dst[2 * k] = a; dst[2 * k + 1] = q;
    }
  }
}


// U[p][row][k] = (G g G^T)[p] for the 36 positions, packed like the F(2x2) variant.  The stores are laid along the packed
// rows: a workgroup owns RB rows x KB consecutive k of U (KB = 128: 8 rows, KB = 32: 32 rows), a thread 4 consecutive k of one
// row -> for each of the 36 positions the workgroup writes RB runs of KB * 4 bytes (512 for KB = 128).  (Rounds 1-2: one
// element per thread -- 4-byte pieces into 36 position planes and, in mode 1, taps gathered at a 36 * Cin byte stride; PMC:
// 82 % of the wave time in s_waitcnt, 2 TB/s.)  The taps of the block are staged through LDS as they lie in the weight
// (mode 0: a row's KB x 9 floats are contiguous; mode 1: 9 x RB floats per k).
template <int KB>
__global__ __launch_bounds__(256) void wino4_weight_wide_kernel(const float* __restrict__ w, float* __restrict__ U, int cout,
                                                                int cin, int mode, int rows, int red, int rows_pad) {
  constexpr int LPR = KB / 4;                 // lanes per row
  constexpr int RB = 256 / LPR;               // rows per workgroup
  constexpr int LDK = KB + 4;                 // padded k extent of the LDS image [row][tap][k]
  __shared__ float s_w[RB][9][LDK];
  const int row0 = blockIdx.y * RB, k0 = blockIdx.x * KB;
  const int t = threadIdx.x;
  // all loads of a thread are issued before the first LDS store (a rolled loop would wait for each load in turn: 36 round
  // trips to HBM per workgroup)
  constexpr int PER_T = RB * KB * 9 / 256;          // 36
  float v[PER_T];
  if (mode == 0) {
    // row = co, k = ci: w[(row * cin + k) * 9 + tap], KB * 9 contiguous floats per row
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int e = t + i * 256;
      const int r = e / (KB * 9), c = e - r * (KB * 9);
      v[i] = row0 + r < rows ? w[((long long)(row0 + r) * cin + k0) * 9 + c] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int e = t + i * 256;
      const int r = e / (KB * 9), c = e - r * (KB * 9);
      const int k = c / 9, tap = c - k * 9;
      s_w[r][tap][k] = v[i];
    }
  } else {
    // row = ci, k = co, rotated filter: w[((k * cin) + row) * 9 + (8 - tap)], RB * 9 contiguous floats per k
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int e = t + i * 256;
      const int k = e / (RB * 9), c = e - k * (RB * 9);
      const int r = c / 9;
      v[i] = row0 + r < rows ? w[((long long)(k0 + k) * cin + row0) * 9 + c] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int e = t + i * 256;
      const int k = e / (RB * 9), c = e - k * (RB * 9);
      const int r = c / 9, tap = c - r * 9;
      s_w[r][8 - tap][k] = v[i];
    }
  }
  __syncthreads();
  const int row_l = t / LPR, k4 = (t % LPR) * 4;
  f32x4 tt[6][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    f32x4 r[6];
    g6(ld4(&s_w[row_l][0 * 3 + b][k4]), ld4(&s_w[row_l][1 * 3 + b][k4]), ld4(&s_w[row_l][2 * 3 + b][k4]), r);
#pragma unroll
    for (int a = 0; a < 6; ++a) tt[a][b] = r[a];
  }
  float* dst = U + (long long)(row0 + row_l) * red + k0 + k4;
  const long long ps = (long long)rows_pad * red;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    f32x4 r[6];
    g6(tt[a][0], tt[a][1], tt[a][2], r);
#pragma unroll
    for (int b = 0; b < 6; ++b) st4(dst + (a * 6 + b) * ps, r[b]);
  }
}

// ws[p][split][co][ci] (36 positions) -> dw[co][ci][3][3] = A3^T m A3.  Block = 192 threads = 16 output quads
// (64 outputs, 16-byte loads) x 12 position lanes of 3 positions each; the first 64 threads apply the transform.
__global__ __launch_bounds__(192) void wino4_dw_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits,
                                                      int cout, int cin) {
  __shared__ float s_m[36][68];
  const int il = threadIdx.x & 15, pg = threadIdx.x >> 4;          // 16 x 12
  const long long n = (long long)cout * cin;                         // multiple of 4 (cin % 4 == 0)
  const long long idx4 = (long long)blockIdx.x * 64 + il * 4;
  f32x4 v[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) v[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (idx4 < n) {
    const float* src = ws + (long long)(pg * 3) * splits * n + idx4;
    const long long pstride = (long long)splits * n;
    for (int k = 0; k < splits; ++k) {                                // fixed order 0, 1, 2, ... per position
      f32x4 t[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) t[q] = ld4(src + q * pstride + k * n);
#pragma unroll
      for (int q = 0; q < 3; ++q) v[q] += t[q];
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) s_m[pg * 3 + q][il * 4 + e] = v[q][e];
  __syncthreads();
  const long long idx = (long long)blockIdx.x * 64 + threadIdx.x;
  if (threadIdx.x >= 64 || idx >= n) return;
  const int o = threadIdx.x;
  float s[3][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float r[3];
    a3(s_m[0 * 6 + j][o], s_m[1 * 6 + j][o], s_m[2 * 6 + j][o], s_m[3 * 6 + j][o], s_m[4 * 6 + j][o], s_m[5 * 6 + j][o], r);
#pragma unroll
    for (int i = 0; i < 3; ++i) s[i][j] = r[i];
  }
  float* dst = dw + idx * 9;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float r[3];
    a3(s[i][0], s[i][1], s[i][2], s[i][3], s[i][4], s[i][5], r);
    dst[i * 3 + 0] = r[0];
    dst[i * 3 + 1] = r[1];
    dst[i * 3 + 2] = r[2];
  }
}

// tiles reduced by one block of an output transform = one BatchNorm partial row; small maps get short blocks so
// that the launch still covers the chip
inline int tiles_per_block(long long T) { return T <= 2048 ? 8 : T <= 16384 ? 16 : kTilesPerBlock; }

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline long long tiles_of(int batch, int h, int w, int m) { return (long long)batch * ((h + m - 1) / m) * ((w + m - 1) / m); }
inline bool tile_ok(int m) { return m == 2 || m == 4; }
inline int npos(int m) { return (m + 2) * (m + 2); }       // 16 or 36 transformed positions

}  // namespace

extern "C" size_t fsd_wino_packed_weight_elems(int rows, int red, int tile) {
  return (size_t)npos(tile) * round_up(rows, 128) * red;
}

extern "C" int fsd_wino_pack_weight(const float* w_oihw, float* u_packed, int cout, int cin, int mode, int tile,
                                    hipStream_t stream) {
  (void)hipGetLastError();
  if (!w_oihw || !u_packed || cout < 1 || cin < 1 || (mode != 0 && mode != 1) || !tile_ok(tile)) return FSD_ERR_ARG;
  const int rows = mode == 0 ? cout : cin, red = mode == 0 ? cin : cout;
  if (red % 32) return FSD_ERR_UNSUPPORTED;
  const int rows_pad = round_up(rows, 128);
  const long long total = (long long)rows_pad * red;
  if (tile == 2)
    FSD_LAUNCH(wino_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w_oihw, u_packed,
                       cout, cin, mode, rows, red, rows_pad);
  else if (red % 128 == 0)                 // rows_pad is a multiple of 128: whole blocks of 8 rows
    FSD_LAUNCH(wino4_weight_wide_kernel<128>, dim3(red / 128, rows_pad / 8), dim3(256), 0, stream, w_oihw, u_packed, cout,
                       cin, mode, rows, red, rows_pad);
  else                                     // red % 32 == 0 (checked above)
    FSD_LAUNCH(wino4_weight_wide_kernel<32>, dim3(red / 32, rows_pad / 32), dim3(256), 0, stream, w_oihw, u_packed, cout,
                       cin, mode, rows, red, rows_pad);
  return (int)hipGetLastError();
}

inline bool out4_on(int cout) {
  static const char* out4_env = FSD_TUNE("FSD_WINO_OUT4");
  return cout >= 128 && !(out4_env && out4_env[0] == '0');
}

// K slices of the forward / data-gradient position GEMMs (only the four-channel output transform adds slices)
inline int fwd_ksplit(long long T, int cin, int cout, int tile) {
  return tile == 4 && out4_on(cout) ? fsd_conv::batched_ksplit(T, cin, cout, npos(tile)) : 1;
}

extern "C" size_t fsd_wino_workspace_bytes(int batch, int height, int width, int cin, int cout, int tile) {
  const long long T = tiles_of(batch, height, width, tile);
  const int ks = fwd_ksplit(T, cin, cout, tile);
  return (size_t)npos(tile) * (size_t)(pos_stride(T, cin) + ks * pos_stride(T, cout)) * sizeof(float);
}

extern "C" int fsd_wino_partial_rows(int batch, int height, int width, int tile) {
  const long long T = tiles_of(batch, height, width, tile);
  const int tpb = tiles_per_block(T);
  return (int)((T + tpb - 1) / tpb);
}

extern "C" size_t fsd_wino_v_elems(int batch, int height, int width, int cin, int tile) {
  return (size_t)npos(tile) * (size_t)pos_stride(tiles_of(batch, height, width, tile), cin);
}

extern "C" int fsd_wino_conv3x3_fwd(const float* x, long long x_ld, const float* u_packed, const float* bias, float* y,
                                    long long y_ld, float* bn_partial, void* workspace, size_t workspace_bytes,
                                    float* v_keep, const float* v_in, int batch, int height, int width, int cin,
                                    int cout, int tile, hipStream_t stream) {
  return fsd_wino_conv3x3_fwd_act(x, x_ld, u_packed, bias, y, y_ld, bn_partial, workspace, workspace_bytes, v_keep, v_in,
                                  batch, height, width, cin, cout, tile, 1.f, stream);
}

extern "C" int fsd_wino_conv3x3_fwd_act(const float* x, long long x_ld, const float* u_packed, const float* bias, float* y,
                                        long long y_ld, float* bn_partial, void* workspace, size_t workspace_bytes,
                                        float* v_keep, const float* v_in, int batch, int height, int width, int cin,
                                        int cout, int tile, float slope, hipStream_t stream) {
  return fsd_wino_conv3x3_fwd_ex(x, x_ld, u_packed, bias, y, y_ld, bn_partial, workspace, workspace_bytes, v_keep, v_in, batch,
                                 height, width, cin, cout, tile, slope, nullptr, nullptr, 1.f, stream);
}

extern "C" int fsd_wino_conv3x3_fwd_ex(const float* x, long long x_ld, const float* u_packed, const float* bias, float* y,
                                       long long y_ld, float* bn_partial, void* workspace, size_t workspace_bytes,
                                       float* v_keep, const float* v_in, int batch, int height, int width, int cin,
                                       int cout, int tile, float slope, const float* in_scale, const float* in_shift,
                                       float in_slope, hipStream_t stream) {
  (void)hipGetLastError();
  if ((in_scale == nullptr) != (in_shift == nullptr)) return FSD_ERR_ARG;
  if (in_scale && (tile != 4 || v_in)) return FSD_ERR_UNSUPPORTED;
  if (slope != 1.f && bn_partial) return FSD_ERR_UNSUPPORTED;       // statistics are taken from the linear output
  if ((!x && !v_in) || !u_packed || !y || !workspace || batch < 1 || height < 1 || width < 1 || !tile_ok(tile))
    return FSD_ERR_ARG;
  if (cin % 32 || (cout & 3) || (y_ld & 3) || y_ld < cout) return FSD_ERR_UNSUPPORTED;
  if (!v_in && ((x_ld & 3) || x_ld < cin)) return FSD_ERR_UNSUPPORTED;
  if (workspace_bytes < fsd_wino_workspace_bytes(batch, height, width, cin, cout, tile)) return FSD_ERR_WORKSPACE;
  const int TH = (height + tile - 1) / tile, TW = (width + tile - 1) / tile;
  const long long T = tiles_of(batch, height, width, tile);
  if (T * (long long)(cin > cout ? cin : cout) >= 0x7fffffffLL) return FSD_ERR_UNSUPPORTED;   // 32-bit tile/channel indices
  const int P = npos(tile);
  const int rows_pad = round_up(cout, 128);
  float* Vw = v_keep ? v_keep : reinterpret_cast<float*>(workspace);    // kept for the weight gradient if asked
  float* Mb = reinterpret_cast<float*>(workspace) + (size_t)P * pos_stride(T, cin);
  const long long n_in = T * (cin / 4);
  const float* V = v_in;                                                 // already transformed (fsd_wino_grad_transforms)
  if (!V) {
    // algorithmic bytes of a transform: the activation once + the (tile+2)^2 transformed positions once
    fsd_prof::Scope prof(fsd_prof::kWinoXform, 4.0 * cin * ((double)batch * height * width + (double)P * T), stream);
    if (tile == 2)
      FSD_LAUNCH(wino_input_kernel, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, stream, x, x_ld, Vw,
                         height, width, TH, TW, cin, T);
    else if (in_scale)
      FSD_LAUNCH(wino4_input_kernel<true>, dim3((unsigned)((2 * n_in + 255) / 256)), dim3(256), 0, stream, x, x_ld, Vw,
                         height, width, TH, TW, cin, T, in_scale, in_shift, in_slope, 0LL, T);
    else
      FSD_LAUNCH(wino4_input_kernel<false>, dim3((unsigned)((2 * n_in + 255) / 256)), dim3(256), 0, stream, x, x_ld, Vw,
                         height, width, TH, TW, cin, T, (const float*)nullptr, (const float*)nullptr, 1.f, 0LL, T);
    V = Vw;
  }
  const int ks = fwd_ksplit(T, cin, cout, tile);
  const long long ss = (long long)P * pos_stride(T, cout);          // slice s of every position lies behind slice s - 1 of all
  int rc = fsd_conv::conv_gemm_batched(V, cin, pos_stride(T, cin), u_packed, (long long)rows_pad * cin, Mb, cout, pos_stride(T, cout), T, cin, cout,
                                       P, stream, ks, ss);
  if (rc != 0) return rc;
  const int tpb = tiles_per_block(T);
  const unsigned bx = (unsigned)((T + tpb - 1) / tpb);
  fsd_prof::Scope prof_out(fsd_prof::kWinoXform, 4.0 * cout * ((double)batch * height * width + (double)P * T), stream);
  if (tile == 2) {
    FSD_LAUNCH(wino_output_kernel, dim3(bx, (cout / 4 + 63) / 64), dim3(256), 0, stream, Mb, bias, y, y_ld,
                       bn_partial, height, width, TH, TW, cout, T, tpb, slope);
  } else {
    const int cg = cout / 2;                                 // channel pairs
    if (out4_on(cout))
      FSD_LAUNCH(wino4_output4_kernel<32>, dim3(bx, (cout / 4 + 31) / 32), dim3(256), 0, stream, Mb, bias, y, y_ld,
                         bn_partial, height, width, TH, TW, cout, T, tpb, slope, 1, 0LL, pos_stride(T, cout), 0LL);
    else if (cg <= 32)
      FSD_LAUNCH(wino4_output_kernel<32>, dim3(bx, (cg + 31) / 32), dim3(256), 0, stream, Mb, bias, y, y_ld,
                         bn_partial, height, width, TH, TW, cout, T, tpb, slope);
    else
      FSD_LAUNCH(wino4_output_kernel<64>, dim3(bx, (cg + 63) / 64), dim3(256), 0, stream, Mb, bias, y, y_ld,
                         bn_partial, height, width, TH, TW, cout, T, tpb, slope);
  }
  return (int)hipGetLastError();
}

extern "C" int fsd_wino_fwd_plan(int batch, int height, int width, int cin, int cout, int tile, int* plan4) {
  if (!plan4 || batch < 1 || height < 1 || width < 1 || !tile_ok(tile)) return FSD_ERR_ARG;
  const long long T = tiles_of(batch, height, width, tile);
  int bm = 0, bn = 0, dma = 0;
  const int m_tiles = fsd_conv::conv_gemm_batched_plan(T, cin, cout, &bm, &bn, &dma);
  plan4[0] = bm; plan4[1] = bn; plan4[2] = dma; plan4[3] = m_tiles;
  return 0;
}

extern "C" int fsd_wino_wgrad_plan(int batch, int height, int width, int cin, int cout, int tile, int* plan4) {
  if (!plan4 || batch < 1 || height < 1 || width < 1 || !tile_ok(tile)) return FSD_ERR_ARG;
  const long long T = tiles_of(batch, height, width, tile);
  int dma = 0, splits = 0, tail = 0;
  const int slots = fsd_conv::wgrad_batched_plan(T, cin, cout, npos(tile), &dma, &splits, &tail);
  plan4[0] = dma; plan4[1] = splits; plan4[2] = tail; plan4[3] = slots;
  return 0;
}

extern "C" size_t fsd_wino_wgrad_workspace_bytes(int batch, int height, int width, int cin, int cout, int tile) {
  const long long T = tiles_of(batch, height, width, tile);
  const int P = npos(tile);
  const int splits = fsd_conv::wgrad_batched_splits(T, cin, cout, P);
  return ((size_t)P * (pos_stride(T, cin) + pos_stride(T, cout)) + (size_t)P * splits * cout * cin) * sizeof(float);
}

extern "C" int fsd_wino_conv3x3_wgrad(const float* dy, long long dy_ld, const float* x, long long x_ld,
                                      const float* v_kept, const float* wt_in, float* dw_oihw, void* workspace,
                                      size_t workspace_bytes, int batch, int height, int width, int cin, int cout,
                                      int tile, hipStream_t stream) {
  (void)hipGetLastError();
  if ((!dy && !wt_in) || (!x && !v_kept) || !dw_oihw || !workspace || batch < 1 || height < 1 || width < 1 ||
      !tile_ok(tile))
    return FSD_ERR_ARG;
  if ((cin & 3) || (cout & 3)) return FSD_ERR_UNSUPPORTED;
  if (!wt_in && ((dy_ld & 3) || dy_ld < cout)) return FSD_ERR_UNSUPPORTED;
  if (!v_kept && ((x_ld & 3) || x_ld < cin)) return FSD_ERR_UNSUPPORTED;
  if (workspace_bytes < fsd_wino_wgrad_workspace_bytes(batch, height, width, cin, cout, tile)) return FSD_ERR_WORKSPACE;
  const int TH = (height + tile - 1) / tile, TW = (width + tile - 1) / tile;
  const long long T = tiles_of(batch, height, width, tile);
  if (T * (long long)(cin > cout ? cin : cout) >= 0x7fffffffLL) return FSD_ERR_UNSUPPORTED;   // 32-bit tile/channel indices
  const int P = npos(tile);
  float* Vw = reinterpret_cast<float*>(workspace);
  float* Wt = Vw + (size_t)P * pos_stride(T, cin);
  float* ws = Wt + (size_t)P * pos_stride(T, cout);
  const long long n_in = T * (cin / 4), n_dy = T * (cout / 4);
  const float* V = v_kept;                                   // the forward pass's B^T d B, if the caller kept it
  if (!V) {
    // algorithmic bytes of a transform: the activation once + the (tile+2)^2 transformed positions once
    fsd_prof::Scope prof(fsd_prof::kWinoXform, 4.0 * cin * ((double)batch * height * width + (double)P * T), stream);
    if (tile == 2)
      FSD_LAUNCH(wino_input_kernel, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, stream, x, x_ld, Vw,
                         height, width, TH, TW, cin, T);
    else
      FSD_LAUNCH(wino4_input_kernel<false>, dim3((unsigned)((2 * n_in + 255) / 256)), dim3(256), 0, stream, x, x_ld, Vw,
                         height, width, TH, TW, cin, T, (const float*)nullptr, (const float*)nullptr, 1.f, 0LL, T);
    V = Vw;
  }
  const float* Wg = wt_in;                                   // already transformed (fsd_wino_grad_transforms)
  if (!Wg) {
    fsd_prof::Scope prof(fsd_prof::kWinoXform, 4.0 * cout * ((double)batch * height * width + (double)P * T), stream);
    if (tile == 2)
      FSD_LAUNCH(wino_dy_kernel, dim3((unsigned)((n_dy + 255) / 256)), dim3(256), 0, stream, dy, dy_ld, Wt,
                         height, width, TH, TW, cout, T);
    else
      FSD_LAUNCH((wino4_dy_kernel<0, 4>), dim3((unsigned)((n_dy + 255) / 256)), dim3(256), 0, stream,
                         const_cast<float*>(dy), dy_ld, Wt, height, width, TH, TW, cout, T, (const float*)nullptr, 0LL,
                         (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, DyFromG{});
    Wg = Wt;
  }
  int splits = 0;
  int rc = fsd_conv::wgrad_gemm_batched(Wg, cout, pos_stride(T, cout), V, cin, pos_stride(T, cin), ws, T, cin, cout, P, &splits, stream);
  if (rc != 0) return rc;
  const long long n = (long long)cout * cin;
  if (tile == 2)
    FSD_LAUNCH(wino_dw_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, stream, ws, dw_oihw, splits, cout, cin);
  else
    FSD_LAUNCH(wino4_dw_kernel, dim3((unsigned)((n + 63) / 64)), dim3(192), 0, stream, ws, dw_oihw, splits, cout, cin);
  return (int)hipGetLastError();
}

extern "C" int fsd_wino_grad_transforms(const float* dt, long long dt_ld, const float* y, long long y_ld, const float* coef,
                                        const float* mean, const float* invstd, float* v_out, float* wt_out, int batch,
                                        int height, int width, int channels, int tile, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dt || !y || !coef || !mean || !invstd || !v_out || !wt_out || batch < 1 || height < 1 || width < 1) return FSD_ERR_ARG;
  if (tile != 4 || (channels & 3) || (dt_ld & 1) || (y_ld & 1) || dt_ld < channels || y_ld < channels) return FSD_ERR_UNSUPPORTED;
  const int TH = (height + 3) / 4, TW = (width + 3) / 4;
  const long long T = tiles_of(batch, height, width, 4);
  if (T * (long long)channels >= 0x7fffffffLL) return FSD_ERR_UNSUPPORTED;
  const long long n = T * (channels / 2);
  fsd_prof::Scope prof(fsd_prof::kWinoXform, 4.0 * channels * (2.0 * batch * height * width + 2.0 * 36 * T), stream);
  FSD_LAUNCH(wino4_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dt, dt_ld, y, y_ld, coef,
                     mean, invstd, v_out, wt_out, height, width, TH, TW, channels, T);
  return (int)hipGetLastError();
}

extern "C" int fsd_wino_dy_bn_transform(float* dt, long long dt_ld, const float* y, long long y_ld, const float* coef,
                                        const float* mean, const float* invstd, float* wt_out, int batch, int height,
                                        int width, int channels, int tile, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dt || !y || !coef || !mean || !invstd || !wt_out || batch < 1 || height < 1 || width < 1) return FSD_ERR_ARG;
  if (tile != 4 || (channels & 3) || (dt_ld & 3) || (y_ld & 3) || dt_ld < channels || y_ld < channels) return FSD_ERR_UNSUPPORTED;
  const int TH = (height + 3) / 4, TW = (width + 3) / 4;
  const long long T = tiles_of(batch, height, width, 4);
  if (T * (long long)channels >= 0x7fffffffLL) return FSD_ERR_UNSUPPORTED;
  const long long n = T * (channels / 4);
  // reads dt and y, writes dy (in place) and the 36 transformed positions
  fsd_prof::Scope prof(fsd_prof::kWinoXform, 4.0 * channels * (3.0 * batch * height * width + 36.0 * T), stream);
  FSD_LAUNCH((wino4_dy_kernel<1, 4>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dt, dt_ld, wt_out,
                     height, width, TH, TW, channels, T, y, y_ld, coef, mean, invstd, DyFromG{});
  return (int)hipGetLastError();
}

extern "C" int fsd_wino_dy_bn_transform_g(const float* dz, long long dz_ld, const float* dz_full, long long dz_full_ld,
                                          const float* y, long long y_ld, const float* scale, const float* shift, float slope,
                                          int pool, const float* coef, const float* mean, const float* invstd, float* dy,
                                          float* wt_out, int batch, int height, int width, int channels, int tile,
                                          hipStream_t stream) {
  (void)hipGetLastError();
  if (!dz || !y || !coef || !mean || !invstd || !dy || !wt_out || batch < 1 || height < 1 || width < 1) return FSD_ERR_ARG;
  if (tile != 4 || (channels & 3) || (dz_ld & 3) || (y_ld & 3) || y_ld < channels || (dz_full && (dz_full_ld & 3)))
    return FSD_ERR_UNSUPPORTED;
  if (pool != 0 && pool != 1) return FSD_ERR_UNSUPPORTED;
  const int TH = (height + 3) / 4, TW = (width + 3) / 4;
  const long long T = tiles_of(batch, height, width, 4);
  if (T * (long long)channels >= 0x7fffffffLL) return FSD_ERR_UNSUPPORTED;
  DyFromG gg;
  gg.dz = dz; gg.dz_full = dz_full; gg.scale = scale; gg.shift = shift; gg.dz_ld = dz_ld; gg.dzf_ld = dz_full_ld;
  gg.slope = slope; gg.pool = pool; gg.OH = pool ? height / 2 : height; gg.OW = pool ? width / 2 : width;
  const long long n = T * (channels / 4);
  // reads dz (+ dz_full) and y, writes dy and the 36 transformed positions
  fsd_prof::Scope prof(fsd_prof::kWinoXform, 4.0 * channels * ((double)batch * gg.OH * gg.OW + (dz_full ? 3.0 : 2.0) * batch * height * width + 36.0 * T), stream);
  // FSD_DY_CPT=2 (tuning aid): two channels per thread, 79 registers instead of 130 -- the kernel then fits beside an 8-wave GEMM
  // workgroup of the weight-gradient stream (112 registers of a SIMD lane are free there).  Measured in the step, four runs per
  // arm on one box: 25.84 against 25.83 ms -- co-residency alone buys nothing, the default stays four channels (16-byte accesses).
  static const char* cpt_env = FSD_TUNE("FSD_DY_CPT");
  if (cpt_env && cpt_env[0] == '2')
    FSD_LAUNCH((wino4_dy_kernel<2, 2>), dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, stream, dy, (long long)channels,
               wt_out, height, width, TH, TW, channels, T, y, y_ld, coef, mean, invstd, gg);
  else
    FSD_LAUNCH((wino4_dy_kernel<2, 4>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dy, (long long)channels,
               wt_out, height, width, TH, TW, channels, T, y, y_ld, coef, mean, invstd, gg);
  return (int)hipGetLastError();
}
