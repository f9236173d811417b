// HBM-bound backward companions: gradient through maxpool + leaky + BatchNorm (two passes with
// per-channel reductions in between), reorg / global-max scatter, the un-folding of the fused
// reweighting (x) head gradient, and small utilities.  NHWC, float4 along channels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "fsdet.h"
#include "profile.hpp"
#include "ew_types.hpp"

namespace {

using fsd_ew::bf16_t;
using fsd_ew::f32x4;
using fsd_ew::ld1;
using fsd_ew::ld4;
using fsd_ew::st1;
using fsd_ew::st4;
constexpr int kPixPerBlock = 256;     // pixels reduced by one block of the first pass
constexpr int kSlots = 256;

inline unsigned blocks_for(long long n, int per) {
  long long b = (n + per - 1) / per;
  return (unsigned)(b < 1 ? 1 : b);
}

template <typename T>
struct ActBwdArgsT {
  const T* dz;            // grad of the (pooled) block output, NHWC (B,OH,OW,C), stride dz_ld
  const T* dz_full;       // optional grad of the un-pooled activation (a [route] tapped it)
  const T* y;             // raw conv output (B,H,W,C), stride y_ld
  const float* scale;     // BN affine (nullable = identity)
  const float* shift;
  const float* mean;      // BN batch statistics (nullable when no BN)
  const float* invstd;
  T* dt;                  // out: grad wrt the BN output / pre-activation, dense (pixels, C); null = statistics only
  float* partial;         // out: [blocks_x][C][2]  (sum dt, sum dt*xhat)
  long long dz_ld, dzf_ld, y_ld;
  int H, W, OH, OW, C, pool;
  long long pixels;
  float slope;
  int ppb;                // pixels (cells x 4 for the pool2 kernel) covered by one block = one partial row
};

// One block: GL channel groups (4 channels each) x 256/GL pixel lanes, looping over kPixPerBlock pixels.
// GL follows the layer (8 for 32 channels ... 64 for >= 256) so that narrow layers keep every lane busy.
template <typename T, int GL>
__global__ __launch_bounds__(256) void act_bwd_kernel(ActBwdArgsT<T> p) {
  constexpr int NPL = 256 / GL;
  __shared__ float s_red[NPL][GL][8];
  const int gl = threadIdx.x % GL, pl = threadIdx.x / GL;
  const int g = blockIdx.y * GL + gl;
  const int cg = p.C >> 2;
  const bool g_ok = g < cg;
  const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 sc = (g_ok && p.scale) ? ld4(p.scale + g * 4) : one;
  const f32x4 sh = (g_ok && p.shift) ? ld4(p.shift + g * 4) : zero;
  const f32x4 mu = (g_ok && p.mean) ? ld4(p.mean + g * 4) : zero;
  const f32x4 is = (g_ok && p.invstd) ? ld4(p.invstd + g * 4) : one;
  f32x4 s1 = zero, s2 = zero;
  const long long p0 = (long long)blockIdx.x * p.ppb;
  if (g_ok) {
    for (int it = pl; it < p.ppb; it += NPL) {
      const long long pix = p0 + it;
      if (pix >= p.pixels) break;
      const int ix = (int)(pix % p.W);
      const long long t = pix / p.W;
      const int iy = (int)(t % p.H);
      const long long b = t / p.H;
      const f32x4 yv = ld4(p.y + pix * p.y_ld + g * 4);
      f32x4 tv, gin = zero;
#pragma unroll
      for (int k = 0; k < 4; ++k) tv[k] = __builtin_fmaf(yv[k], sc[k], sh[k]);
      if (p.pool == 0) {
        gin = ld4(p.dz + pix * p.dz_ld + g * 4);
      } else {
        // windows that contain this pixel: stride 2 -> one, stride 1 -> up to four
        const int oy_lo = p.pool == 1 ? iy >> 1 : (iy > 0 ? iy - 1 : 0), oy_hi = p.pool == 1 ? iy >> 1 : iy;
        const int ox_lo = p.pool == 1 ? ix >> 1 : (ix > 0 ? ix - 1 : 0), ox_hi = p.pool == 1 ? ix >> 1 : ix;
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
          for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            if (oy >= p.OH || ox >= p.OW) continue;
            const int y0 = p.pool == 1 ? 2 * oy : oy, x0 = p.pool == 1 ? 2 * ox : ox;
            // first maximum in scan order wins (like torch's max_pool2d backward)
            f32x4 best;
            int by[4], bx[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int wy = min(y0 + (q >> 1), p.H - 1), wx = min(x0 + (q & 1), p.W - 1);
              const f32x4 v = ld4(p.y + ((b * p.H + wy) * (long long)p.W + wx) * p.y_ld + g * 4);
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                float a = __builtin_fmaf(v[k], sc[k], sh[k]);
                a = a > 0.f ? a : a * p.slope;
                if (q == 0 || a > best[k]) { best[k] = a; by[k] = wy; bx[k] = wx; }
              }
            }
            const f32x4 gz = ld4(p.dz + ((b * p.OH + oy) * (long long)p.OW + ox) * p.dz_ld + g * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (by[k] == iy && bx[k] == ix) gin[k] += gz[k];
          }
      }
      if (p.dz_full) {
        const f32x4 gf = ld4(p.dz_full + pix * p.dzf_ld + g * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) gin[k] += gf[k];
      }
      f32x4 d;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        d[k] = tv[k] > 0.f ? gin[k] : gin[k] * p.slope;
        s1[k] += d[k];
        s2[k] += d[k] * ((yv[k] - mu[k]) * is[k]);
      }
      if (p.dt) st4<T>(p.dt + pix * p.C + g * 4, d);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    s_red[pl][gl][k] = s1[k];
    s_red[pl][gl][4 + k] = s2[k];
  }
  __syncthreads();
  // fold the NPL pixel lanes: one thread per (channel group, value) pair, fixed order.  (Unrolling this over all
  // lanes of the 32-channel layer -- 32 lanes x 8 values per thread -- cost 254 VGPRs and occupancy 1.)
  for (int t = threadIdx.x; t < GL * 8; t += 256) {
    const int tg = t >> 3, k8 = t & 7;
    const int gg = blockIdx.y * GL + tg;
    if (gg >= cg) continue;
    float a = 0.f;
#pragma unroll 4
    for (int l = 0; l < NPL; ++l) a += s_red[l][tg][k8];
    // k8 < 4: sum of dt for channel gg*4 + k8;  k8 >= 4: sum of dt * xhat for channel gg*4 + k8 - 4
    p.partial[((long long)blockIdx.x * p.C + gg * 4 + (k8 & 3)) * 2 + (k8 >> 2)] = a;
  }
}

// Statistics only, no pool, no un-pooled tap (the first pass of most fp32 BatchNorm layers since dt is not materialised):
// the same sums in the same order as act_bwd_kernel, without its window / image-coordinate code (99 -> ~40 registers).
template <int GL>
__global__ __launch_bounds__(256) void act_stats_kernel(ActBwdArgsT<float> p) {
  constexpr int NPL = 256 / GL;
  __shared__ float s_red[NPL][GL][8];
  const int gl = threadIdx.x % GL, pl = threadIdx.x / GL;
  const int g = blockIdx.y * GL + gl;
  const int cg = p.C >> 2;
  const bool g_ok = g < cg;
  const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 sc = (g_ok && p.scale) ? ld4(p.scale + g * 4) : one;
  const f32x4 sh = (g_ok && p.shift) ? ld4(p.shift + g * 4) : zero;
  const f32x4 mu = (g_ok && p.mean) ? ld4(p.mean + g * 4) : zero;
  const f32x4 is = (g_ok && p.invstd) ? ld4(p.invstd + g * 4) : one;
  f32x4 s1 = zero, s2 = zero;
  const long long p0 = (long long)blockIdx.x * p.ppb;
  if (g_ok) {
    long long left = p.pixels - p0;
    const int n = (int)(left < p.ppb ? left : p.ppb);
    // UB pixels of a lane in flight at a time (`#pragma unroll 4` left the loads in their guarded bodies: four dependent
    // round trips where one does).  Same pixels in the same order per lane: bit-identical sums.
    constexpr int UB = 8;
    const float* yp = p.y + p0 * p.y_ld + g * 4;
    const float* zp = p.dz + p0 * p.dz_ld + g * 4;
    for (int it0 = pl; it0 < n; it0 += NPL * UB) {
      f32x4 yr[UB], zr[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int it = it0 + u * NPL;
        const int itc = it < n ? it : it0;                                // clamped: always a mapped pixel of this lane
        yr[u] = ld4(yp + (long long)itc * p.y_ld);
        zr[u] = ld4(zp + (long long)itc * p.dz_ld);
      }
      asm volatile("" ::: "memory");        // the loads stay up here
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        if (it0 + u * NPL >= n) break;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float tv = __builtin_fmaf(yr[u][k], sc[k], sh[k]);
          const float d = tv > 0.f ? zr[u][k] : zr[u][k] * p.slope;
          s1[k] += d;
          s2[k] += d * ((yr[u][k] - mu[k]) * is[k]);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    s_red[pl][gl][k] = s1[k];
    s_red[pl][gl][4 + k] = s2[k];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < GL * 8; t += 256) {
    const int tg = t >> 3, k8 = t & 7;
    const int gg = blockIdx.y * GL + tg;
    if (gg >= cg) continue;
    float a = 0.f;
#pragma unroll 4
    for (int l = 0; l < NPL; ++l) a += s_red[l][tg][k8];
    p.partial[((long long)blockIdx.x * p.C + gg * 4 + (k8 & 3)) * 2 + (k8 >> 2)] = a;
  }
}

// bf16 storage mode twin of act_stats_kernel with EIGHT channels per lane (16-byte loads; the generic act_bwd_kernel the mode used
// before moves 8 bytes per lane and carries 64-bit image-coordinate divisions it does not need here: 0.50 ms per step over 16
// launches).  Same sums, fp32 arithmetic on the stored (bf16) values.
template <int GL>
__global__ __launch_bounds__(256) void act_stats8_kernel(ActBwdArgsT<bf16_t> p) {
  constexpr int NPL = 256 / GL;
  __shared__ float s_red[NPL][GL][16];
  const int gl = threadIdx.x % GL, pl = threadIdx.x / GL;
  const int g = blockIdx.y * GL + gl;
  const int cg = p.C >> 3;
  const bool g_ok = g < cg;
  float sc[8], sh[8], mu[8], is[8], s1[8], s2[8];
  {
    const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[8];                           // two 16-byte loads per vector (element loads under 32 separate branches before)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      v[0 + q] = (g_ok && p.scale) ? ld4(p.scale + g * 8 + 4 * q) : one;
      v[2 + q] = (g_ok && p.shift) ? ld4(p.shift + g * 8 + 4 * q) : zero;
      v[4 + q] = (g_ok && p.mean) ? ld4(p.mean + g * 8 + 4 * q) : zero;
      v[6 + q] = (g_ok && p.invstd) ? ld4(p.invstd + g * 8 + 4 * q) : one;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      sc[k] = v[0 + (k >> 2)][k & 3];
      sh[k] = v[2 + (k >> 2)][k & 3];
      mu[k] = v[4 + (k >> 2)][k & 3];
      is[k] = v[6 + (k >> 2)][k & 3];
      s1[k] = 0.f;
      s2[k] = 0.f;
    }
  }
  const long long p0 = (long long)blockIdx.x * p.ppb;
  if (g_ok) {
    long long left = p.pixels - p0;
    const int n = (int)(left < p.ppb ? left : p.ppb);
    // UB pixels of a lane in flight at a time, as raw 16-byte words (the 13x13 / 26x26 layers run ~1.3 waves per SIMD with 16
    // pixels per lane: two pixels at a time were eight dependent round trips to HBM -- 25 us for 44 MB).  Same pixels in
    // the same order per lane: the sums are bit-identical to the two-at-a-time loop.
    constexpr int UB = 8;
    const bf16_t* yp = p.y + p0 * p.y_ld + g * 8;
    const bf16_t* zp = p.dz + p0 * p.dz_ld + g * 8;
    for (int it0 = pl; it0 < n; it0 += NPL * UB) {
      uint4 yr[UB], zr[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int it = it0 + u * NPL;
        const int itc = it < n ? it : it0;                              // clamped: always a mapped pixel of this lane
        yr[u] = *reinterpret_cast<const uint4*>(yp + (long long)itc * p.y_ld);
        zr[u] = *reinterpret_cast<const uint4*>(zp + (long long)itc * p.dz_ld);
      }
      asm volatile("" ::: "memory");      // the loads stay up here (the compiler otherwise sinks each pair into its guarded use)
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        if (it0 + u * NPL >= n) break;
        const unsigned yw[4] = {yr[u].x, yr[u].y, yr[u].z, yr[u].w}, zw[4] = {zr[u].x, zr[u].y, zr[u].z, zr[u].w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float yk = __uint_as_float((k & 1) ? (yw[k >> 1] & 0xffff0000u) : (yw[k >> 1] << 16));
          const float gk = __uint_as_float((k & 1) ? (zw[k >> 1] & 0xffff0000u) : (zw[k >> 1] << 16));
          const float tv = __builtin_fmaf(yk, sc[k], sh[k]);
          const float d = tv > 0.f ? gk : gk * p.slope;
          s1[k] += d;
          s2[k] += d * ((yk - mu[k]) * is[k]);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    s_red[pl][gl][k] = s1[k];
    s_red[pl][gl][8 + k] = s2[k];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < GL * 16; t += 256) {
    const int tg = t >> 4, k16 = t & 15;
    const int gg = blockIdx.y * GL + tg;
    if (gg >= cg) continue;
    float a = 0.f;
#pragma unroll 4
    for (int l = 0; l < NPL; ++l) a += s_red[l][tg][k16];
    p.partial[((long long)blockIdx.x * p.C + gg * 8 + (k16 & 7)) * 2 + (k16 >> 3)] = a;
  }
}

// bf16 storage, pool == 1, STATISTICS ONLY (dt not materialised, no un-pooled tap -- every pooled BatchNorm layer of the
// bf16 mode's backward): one lane = one 2x2 cell x EIGHT channels, the cell's five 16-byte loads issued together.  The generic
// act_bwd_pool2_kernel moves 8 bytes per lane and ran these launches at 1.2-2.6 TB/s (0.50 ms per step).  Same sums: the same
// cells in the same order per lane as that kernel would take with the same block geometry, same float expressions.
template <int GL>
__global__ __launch_bounds__(256) void act_stats_pool8_kernel(ActBwdArgsT<bf16_t> p) {
  constexpr int NPL = 256 / GL;
  __shared__ float s_red[NPL][GL][16];
  const int gl = threadIdx.x % GL, pl = threadIdx.x / GL;
  const int g = blockIdx.y * GL + gl;
  const int cg = p.C >> 3;
  const bool g_ok = g < cg;
  float sc[8], sh[8], mu[8], is[8], s1[8], s2[8];
  {
    const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      v[0 + q] = (g_ok && p.scale) ? ld4(p.scale + g * 8 + 4 * q) : one;
      v[2 + q] = (g_ok && p.shift) ? ld4(p.shift + g * 8 + 4 * q) : zero;
      v[4 + q] = (g_ok && p.mean) ? ld4(p.mean + g * 8 + 4 * q) : zero;
      v[6 + q] = (g_ok && p.invstd) ? ld4(p.invstd + g * 8 + 4 * q) : one;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      sc[k] = v[0 + (k >> 2)][k & 3];
      sh[k] = v[2 + (k >> 2)][k & 3];
      mu[k] = v[4 + (k >> 2)][k & 3];
      is[k] = v[6 + (k >> 2)][k & 3];
      s1[k] = 0.f;
      s2[k] = 0.f;
    }
  }
  // 32-bit index arithmetic (cells and pixels fit: the launcher checks); the image coordinates of a lane's cell are carried
  // from trip to trip (+NPL cells) instead of being divided out of the cell index every time
  const unsigned CH = (unsigned)(p.H + 1) >> 1, CW = (unsigned)(p.W + 1) >> 1;
  const unsigned cells = (unsigned)(p.pixels / ((long long)p.H * p.W)) * CH * CW;
  const unsigned c0 = blockIdx.x * (unsigned)(p.ppb / 4);
  if (g_ok) {
    const int ncell = p.ppb / 4;
    unsigned cx = (c0 + pl) % CW, t0 = (c0 + pl) / CW;
    unsigned cy = t0 % CH, b = t0 / CH;
    for (int it = pl; it < ncell; it += NPL) {
      const unsigned cell = c0 + it;
      if (cell >= cells) break;
      const bool win = cy < (unsigned)p.OH && cx < (unsigned)p.OW;
      uint4 yr[4], zr;
      bool in[4];
      const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned yy = 2 * cy + (q >> 1), xx = 2 * cx + (q & 1);
        in[q] = yy < (unsigned)p.H && xx < (unsigned)p.W;
        yr[q] = in[q] ? *reinterpret_cast<const uint4*>(p.y + (long long)((b * p.H + yy) * p.W + xx) * p.y_ld + g * 8) : z4;
      }
      zr = win ? *reinterpret_cast<const uint4*>(p.dz + (long long)((b * p.OH + cy) * p.OW + cx) * p.dz_ld + g * 8) : z4;
      const unsigned zw[4] = {zr.x, zr.y, zr.z, zr.w};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float gz = __uint_as_float((k & 1) ? (zw[k >> 1] & 0xffff0000u) : (zw[k >> 1] << 16));
        float yk[4], tv[4];
        int best = 0;
        float bv = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned w = q == 0 ? (k >> 1 == 0 ? yr[0].x : k >> 1 == 1 ? yr[0].y : k >> 1 == 2 ? yr[0].z : yr[0].w)
                           : q == 1 ? (k >> 1 == 0 ? yr[1].x : k >> 1 == 1 ? yr[1].y : k >> 1 == 2 ? yr[1].z : yr[1].w)
                           : q == 2 ? (k >> 1 == 0 ? yr[2].x : k >> 1 == 1 ? yr[2].y : k >> 1 == 2 ? yr[2].z : yr[2].w)
                                    : (k >> 1 == 0 ? yr[3].x : k >> 1 == 1 ? yr[3].y : k >> 1 == 2 ? yr[3].z : yr[3].w);
          yk[q] = __uint_as_float((k & 1) ? (w & 0xffff0000u) : (w << 16));
          tv[q] = __builtin_fmaf(yk[q], sc[k], sh[k]);
          const float a = tv[q] > 0.f ? tv[q] : tv[q] * p.slope;
          if (q == 0 || a > bv) { bv = a; best = q; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (!in[q]) continue;
          const float gin = (win && best == q) ? gz : 0.f;
          const float d = tv[q] > 0.f ? gin : gin * p.slope;
          s1[k] += d;
          s2[k] += d * ((yk[q] - mu[k]) * is[k]);
        }
      }
      cx += NPL;                                    // next cell of this lane (NPL <= 32 cells further)
      while (cx >= CW) {
        cx -= CW;
        if (++cy == CH) { cy = 0; ++b; }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    s_red[pl][gl][k] = s1[k];
    s_red[pl][gl][8 + k] = s2[k];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < GL * 16; t += 256) {
    const int tg = t >> 4, k16 = t & 15;
    const int gg = blockIdx.y * GL + tg;
    if (gg >= cg) continue;
    float a = 0.f;
#pragma unroll 4
    for (int l = 0; l < NPL; ++l) a += s_red[l][tg][k16];
    p.partial[((long long)blockIdx.x * p.C + gg * 8 + (k16 & 7)) * 2 + (k16 >> 3)] = a;
  }
}

// pool == 1 (2x2 stride 2) specialisation: one thread per 2x2 CELL and 4 channels, so every y is
// read once, the argmax is decided once and the (up to) four dt values are written together.
// Cells on the odd border (no pooling window) only carry the dz_full / zero gradient.
template <typename T, int GL>
__global__ __launch_bounds__(256) void act_bwd_pool2_kernel(ActBwdArgsT<T> p) {
  constexpr int NPL = 256 / GL;
  __shared__ float s_red[NPL][GL][8];
  const int gl = threadIdx.x % GL, pl = threadIdx.x / GL;
  const int g = blockIdx.y * GL + gl;
  const int cg = p.C >> 2;
  const bool g_ok = g < cg;
  const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 sc = (g_ok && p.scale) ? ld4(p.scale + g * 4) : one;
  const f32x4 sh = (g_ok && p.shift) ? ld4(p.shift + g * 4) : zero;
  const f32x4 mu = (g_ok && p.mean) ? ld4(p.mean + g * 4) : zero;
  const f32x4 is = (g_ok && p.invstd) ? ld4(p.invstd + g * 4) : one;
  const int CH = (p.H + 1) >> 1, CW = (p.W + 1) >> 1;
  const long long cells = (long long)(p.pixels / ((long long)p.H * p.W)) * CH * CW;
  f32x4 s1 = zero, s2 = zero;
  const long long c0 = (long long)blockIdx.x * (p.ppb / 4);
  if (g_ok) {
    for (int it = pl; it < p.ppb / 4; it += NPL) {
      const long long cell = c0 + it;
      if (cell >= cells) break;
      const int cx = (int)(cell % CW);
      const long long t = cell / CW;
      const int cy = (int)(t % CH);
      const long long b = t / CH;
      const bool win = cy < p.OH && cx < p.OW;
      f32x4 yv[4], tv[4];
      bool in[4];
      int best[4] = {0, 0, 0, 0};
      f32x4 bv = zero;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int yy = 2 * cy + (q >> 1), xx = 2 * cx + (q & 1);
        in[q] = yy < p.H && xx < p.W;
        yv[q] = in[q] ? ld4(p.y + ((b * p.H + yy) * (long long)p.W + xx) * p.y_ld + g * 4) : zero;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          tv[q][k] = __builtin_fmaf(yv[q][k], sc[k], sh[k]);
          const float a = tv[q][k] > 0.f ? tv[q][k] : tv[q][k] * p.slope;
          if (q == 0 || a > bv[k]) { bv[k] = a; best[k] = q; }
        }
      }
      const f32x4 gz = win ? ld4(p.dz + ((b * p.OH + cy) * (long long)p.OW + cx) * p.dz_ld + g * 4) : zero;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!in[q]) continue;
        const int yy = 2 * cy + (q >> 1), xx = 2 * cx + (q & 1);
        const long long pix = (b * p.H + yy) * (long long)p.W + xx;
        f32x4 gin = zero;
        if (p.dz_full) gin = ld4(p.dz_full + pix * p.dzf_ld + g * 4);
        f32x4 d;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (win && best[k] == q) gin[k] += gz[k];
          d[k] = tv[q][k] > 0.f ? gin[k] : gin[k] * p.slope;
          s1[k] += d[k];
          s2[k] += d[k] * ((yv[q][k] - mu[k]) * is[k]);
        }
        if (p.dt) st4<T>(p.dt + pix * p.C + g * 4, d);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    s_red[pl][gl][k] = s1[k];
    s_red[pl][gl][4 + k] = s2[k];
  }
  __syncthreads();
  // fold the NPL pixel lanes: one thread per (channel group, value) pair, fixed order.  (Unrolling this over all
  // lanes of the 32-channel layer -- 32 lanes x 8 values per thread -- cost 254 VGPRs and occupancy 1.)
  for (int t = threadIdx.x; t < GL * 8; t += 256) {
    const int tg = t >> 3, k8 = t & 7;
    const int gg = blockIdx.y * GL + tg;
    if (gg >= cg) continue;
    float a = 0.f;
#pragma unroll 4
    for (int l = 0; l < NPL; ++l) a += s_red[l][tg][k8];
    // k8 < 4: sum of dt for channel gg*4 + k8;  k8 >= 4: sum of dt * xhat for channel gg*4 + k8 - 4
    p.partial[((long long)blockIdx.x * p.C + gg * 4 + (k8 & 3)) * 2 + (k8 >> 2)] = a;
  }
}

// [rows][two_c] float partials -> [n_slots][two_c] doubles.  Block = RL row lanes x CL columns; row lanes are
// folded through LDS in a fixed order (deterministic).
template <int RL>
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, double* __restrict__ slots,
                                                              int rows, int two_c, int n_slots) {
  constexpr int CL = 256 / RL;
  __shared__ double s_acc[RL][CL];
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int e = blockIdx.y * CL + cl;
  const int s = blockIdx.x;
  double acc = 0.0;
  if (e < two_c) {
    const long long step = (long long)n_slots * RL;
    long long t = s + (long long)n_slots * rl;
    for (; t + 3 * step < rows; t += 4 * step) {
      const float v0 = partial[t * two_c + e], v1 = partial[(t + step) * two_c + e];
      const float v2 = partial[(t + 2 * step) * two_c + e], v3 = partial[(t + 3 * step) * two_c + e];
      acc += (double)v0; acc += (double)v1; acc += (double)v2; acc += (double)v3;
    }
    for (; t < rows; t += step) acc += (double)partial[t * two_c + e];
  }
  if constexpr (RL > 1) {
    s_acc[rl][cl] = acc;
    __syncthreads();
    if (rl != 0) return;
#pragma unroll
    for (int l = 1; l < RL; ++l) acc += s_acc[l][cl];
  }
  if (e < two_c) slots[(long long)s * two_c + e] = acc;
}

inline void launch_reduce_partials(const float* partial, double* slots, int rows, int two_c, int n_slots, hipStream_t stream) {
  if (two_c <= 64)
    FSD_LAUNCH(reduce_partials_kernel<4>, dim3(n_slots, (two_c + 63) / 64), dim3(256), 0, stream, partial, slots, rows, two_c, n_slots);
  else if (two_c <= 128)
    FSD_LAUNCH(reduce_partials_kernel<2>, dim3(n_slots, (two_c + 127) / 128), dim3(256), 0, stream, partial, slots, rows, two_c, n_slots);
  else
    FSD_LAUNCH(reduce_partials_kernel<1>, dim3(n_slots, (two_c + 255) / 256), dim3(256), 0, stream, partial, slots, rows, two_c, n_slots);
}

// dgamma/dbeta (or dbias) and the per-channel coefficients of  dy = c1 * (dt - c2 - xhat * c3)
// S = double: `slots` from reduce_partials_kernel; S = float: the partial rows themselves (rows <= slots: one row per slot)
template <typename S>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const S* __restrict__ slots, int n_slots, double count, int channels,
                                       const float* __restrict__ scale, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ coef) {
  // block = 32 channels x 8 slot lanes (coalesced along channels); lanes folded through LDS in a fixed order
  __shared__ double s_part[8][32][2];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < channels) {
#pragma unroll 4
    for (int k = sl; k < n_slots; k += 8) {
      s1 += (double)slots[((long long)k * channels + c) * 2 + 0];
      s2 += (double)slots[((long long)k * channels + c) * 2 + 1];
    }
  }
  s_part[sl][cl][0] = s1;
  s_part[sl][cl][1] = s2;
  __syncthreads();
  if (sl != 0 || c >= channels) return;
#pragma unroll
  for (int l = 1; l < 8; ++l) { s1 += s_part[l][cl][0]; s2 += s_part[l][cl][1]; }
  if (dbeta) dbeta[c] = (float)s1;
  if (dgamma) dgamma[c] = (float)s2;
  if (coef) {
    coef[c] = scale ? scale[c] : 1.f;
    coef[channels + c] = (float)(s1 / count);
    coef[2 * channels + c] = (float)(s2 / count);
  }
}

template <typename T>
__global__ void bn_bwd_apply_kernel(T* __restrict__ dt, const T* __restrict__ y, long long y_ld,
                                    const float* __restrict__ coef, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, int C, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int cg = C >> 2;
  const int g = (int)(idx % cg);
  const long long pix = idx / cg;
  const f32x4 c1 = ld4(coef + g * 4), c2 = ld4(coef + C + g * 4), c3 = ld4(coef + 2 * C + g * 4);
  const f32x4 mu = ld4(mean + g * 4), is = ld4(invstd + g * 4);
  const f32x4 yv = ld4(y + pix * y_ld + g * 4);
  f32x4 d = ld4(dt + pix * C + g * 4);
#pragma unroll
  for (int k = 0; k < 4; ++k) d[k] = c1[k] * (d[k] - c2[k] - (yv[k] - mu[k]) * is[k] * c3[k]);
  st4<T>(dt + pix * C + g * 4, d);
}

// The same second pass for a FIRST pass that only took the statistics (dt never written): dt is formed again from the
// gradient of the block output -- through the 2x2 / stride-2 maxpool (first maximum in scan order, as in the first pass) and the
// leaky activation, same expressions -- and dy = c1 * (dt - c2 - xhat * c3) is written once.  Against first pass + in-place
// apply this drops one write and one read of a full-resolution tensor per layer (pooled layers read a quarter-size gradient
// instead of dt).  POOL 0: one thread per pixel and 4 channels; POOL 1: one thread per 2x2 cell and 4 channels.
template <int POOL>
__global__ __launch_bounds__(256) void bn_bwd_apply_g_kernel(const float* __restrict__ dz, long long dz_ld,
                                                             const float* __restrict__ dz_full, long long dzf_ld,
                                                             const float* __restrict__ y, long long y_ld,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             float slope, const float* __restrict__ coef,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             float* __restrict__ dy, int H, int W, int OH, int OW, int C,
                                                             long long total) {
  // POOL 0: a thread owns 4 channels of UC consecutive pixels -- the seven per-channel vectors (112 bytes, more than a
  // pixel's own data) are fetched once per thread, and the loads of all its pixels are issued before the first use
  // (bn_bwd_apply_g8_kernel: 27 -> 20 us per launch).  `total` = units x channel groups; per-element arithmetic unchanged.
  constexpr int UC = POOL == 0 ? 4 : 1;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cg = C >> 2;
  const long long units = total / cg;
  const int g = (int)(idx % cg);
  const long long unit = idx / cg * UC;                  // first pixel (POOL 0) or the cell (POOL 1)
  if (unit >= units) return;
  const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 sc = scale ? ld4(scale + g * 4) : one, sh = shift ? ld4(shift + g * 4) : zero;
  const f32x4 c1 = ld4(coef + g * 4), c2 = ld4(coef + C + g * 4), c3 = ld4(coef + 2 * C + g * 4);
  const f32x4 mu = ld4(mean + g * 4), is = ld4(invstd + g * 4);
  if constexpr (POOL == 0) {
    f32x4 yv[UC], gin[UC], gf[UC];
    bool ok[UC];
    long long pix[UC];
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      ok[u] = unit + u < units;
      pix[u] = ok[u] ? unit + u : unit;
      yv[u] = ld4(y + pix[u] * y_ld + g * 4);
      gin[u] = ld4(dz + pix[u] * dz_ld + g * 4);
      gf[u] = dz_full ? ld4(dz_full + pix[u] * dzf_ld + g * 4) : zero;
    }
    asm volatile("" ::: "memory");          // the loads stay up here
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float gk = gin[u][k];
        if (dz_full) gk += gf[u][k];
        const float tv = __builtin_fmaf(yv[u][k], sc[k], sh[k]);
        const float d = tv > 0.f ? gk : gk * slope;
        o[k] = c1[k] * (d - c2[k] - (yv[u][k] - mu[k]) * is[k] * c3[k]);
      }
      if (ok[u]) st4<float>(dy + pix[u] * C + g * 4, o);
    }
  } else {
    const int CH = (H + 1) >> 1, CW = (W + 1) >> 1;
    const int cx = (int)(unit % CW);
    const long long t = unit / CW;
    const int cy = (int)(t % CH);
    const long long b = t / CH;
    const bool win = cy < OH && cx < OW;
    f32x4 yv[4], tv[4];
    bool in[4];
    int best[4] = {0, 0, 0, 0};
    f32x4 bv = zero;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int yy = 2 * cy + (q >> 1), xx = 2 * cx + (q & 1);
      in[q] = yy < H && xx < W;
      yv[q] = in[q] ? ld4(y + ((b * H + yy) * (long long)W + xx) * y_ld + g * 4) : zero;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        tv[q][k] = __builtin_fmaf(yv[q][k], sc[k], sh[k]);
        const float a = tv[q][k] > 0.f ? tv[q][k] : tv[q][k] * slope;
        if (q == 0 || a > bv[k]) { bv[k] = a; best[k] = q; }
      }
    }
    const f32x4 gz = win ? ld4(dz + ((b * OH + cy) * (long long)OW + cx) * dz_ld + g * 4) : zero;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!in[q]) continue;
      const int yy = 2 * cy + (q >> 1), xx = 2 * cx + (q & 1);
      const long long pix = (b * H + yy) * (long long)W + xx;
      f32x4 gin = zero;
      if (dz_full) gin = ld4(dz_full + pix * dzf_ld + g * 4);
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (win && best[k] == q) gin[k] += gz[k];
        const float d = tv[q][k] > 0.f ? gin[k] : gin[k] * slope;
        o[k] = c1[k] * (d - c2[k] - (yv[q][k] - mu[k]) * is[k] * c3[k]);
      }
      st4<float>(dy + pix * C + g * 4, o);
    }
  }
}

// The bf16-storage form of bn_bwd_apply_g_kernel, 8 channels (16 bytes) per lane.  dt is formed in fp32 and never rounded to
// the storage type on the way (the two-pass form stores it as bf16 in between), dy is rounded once when it is written.
__device__ __forceinline__ void load8(const bf16_t* p, float (&a)[8]) {
  const fsd_ew::f32x8 v = fsd_ew::ld8(p);
#pragma unroll
  for (int k = 0; k < 4; ++k) { a[k] = v.lo[k]; a[4 + k] = v.hi[k]; }
}
__device__ __forceinline__ void load8f(const float* p, float (&a)[8]) {
  const f32x4 lo = ld4(p), hi = ld4(p + 4);
#pragma unroll
  for (int k = 0; k < 4; ++k) { a[k] = lo[k]; a[4 + k] = hi[k]; }
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&a)[8]) {
  fsd_ew::f32x8 v;
#pragma unroll
  for (int k = 0; k < 4; ++k) { v.lo[k] = a[k]; v.hi[k] = a[4 + k]; }
  fsd_ew::st8(p, v);
}

template <int POOL>
__global__ __launch_bounds__(256) void bn_bwd_apply_g8_kernel(const bf16_t* __restrict__ dz, long long dz_ld,
                                                              const bf16_t* __restrict__ dz_full, long long dzf_ld,
                                                              const bf16_t* __restrict__ y, long long y_ld,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float slope, const float* __restrict__ coef,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              bf16_t* __restrict__ dy, int H, int W, int OH, int OW, int C,
                                                              long long units) {
  // A thread owns 8 channels of UC consecutive units (pixels / 2x2 cells): the seven per-channel vectors (224 bytes, more
  // than a unit's own data) are fetched once per thread instead of once per unit, and the loads of all its units are issued
  // before the first use.  Per-element arithmetic as before (bit-identical results).
  constexpr int UC = POOL == 0 ? 4 : 1;      // (pooled cells: 18 loads each; two per thread measured 58 -> 65 us)
  // 32-bit index arithmetic (the launcher checks that threads and pixels fit): the 64-bit divisions and products of the first
  // version were ~250 of the pooled variant's 1000 instructions per thread
  const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned cg = (unsigned)C >> 3;
  const unsigned g = idx % cg;
  const unsigned unit0 = idx / cg * UC;
  if (unit0 >= (unsigned)units) return;
  float sc[8], sh[8], c1[8], c2[8], c3[8], mu[8], is[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { sc[k] = 1.f; sh[k] = 0.f; }
  if (scale) load8f(scale + g * 8, sc);
  if (shift) load8f(shift + g * 8, sh);
  load8f(coef + g * 8, c1); load8f(coef + C + g * 8, c2); load8f(coef + 2 * C + g * 8, c3);
  load8f(mean + g * 8, mu); load8f(invstd + g * 8, is);
  if constexpr (POOL == 0) {
    float yv[UC][8], gin[UC][8], gf[UC][8];
    bool ok[UC];
    unsigned pix[UC];
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      ok[u] = unit0 + u < (unsigned)units;
      pix[u] = ok[u] ? unit0 + u : unit0;
      load8(y + (long long)pix[u] * y_ld + g * 8, yv[u]);
      load8(dz + (long long)pix[u] * dz_ld + g * 8, gin[u]);
      if (dz_full) load8(dz_full + (long long)pix[u] * dzf_ld + g * 8, gf[u]);
    }
    asm volatile("" ::: "memory");          // the loads stay up here
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      float o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float gk = gin[u][k];
        if (dz_full) gk += gf[u][k];
        const float tv = __builtin_fmaf(yv[u][k], sc[k], sh[k]);
        const float d = tv > 0.f ? gk : gk * slope;
        o[k] = c1[k] * (d - c2[k] - (yv[u][k] - mu[k]) * is[k] * c3[k]);
      }
      if (ok[u]) store8(dy + (long long)pix[u] * C + g * 8, o);
    }
  } else {
    const unsigned unit = unit0;
    const unsigned CH = (unsigned)(H + 1) >> 1, CW = (unsigned)(W + 1) >> 1;
    const int cx = (int)(unit % CW);
    const unsigned t = unit / CW;
    const int cy = (int)(t % CH);
    const unsigned b = t / CH;
    const bool win = cy < OH && cx < OW;
    float yv[4][8], tv[4][8], bv[8], gz[8];
    bool in[4];
    int best[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int yy = 2 * cy + (q >> 1), xx = 2 * cx + (q & 1);
      in[q] = yy < H && xx < W;
      if (in[q]) {
        load8(y + (long long)((b * H + yy) * W + xx) * y_ld + g * 8, yv[q]);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) yv[q][k] = 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        tv[q][k] = __builtin_fmaf(yv[q][k], sc[k], sh[k]);
        const float a = tv[q][k] > 0.f ? tv[q][k] : tv[q][k] * slope;
        if (q == 0 || a > bv[k]) { bv[k] = a; best[k] = q; }
      }
    }
    if (win) {
      load8(dz + (long long)((b * OH + cy) * OW + cx) * dz_ld + g * 8, gz);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) gz[k] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!in[q]) continue;
      const int yy = 2 * cy + (q >> 1), xx = 2 * cx + (q & 1);
      const unsigned pix = (b * H + yy) * W + xx;
      float gin[8], o[8];
      if (dz_full) {
        load8(dz_full + (long long)pix * dzf_ld + g * 8, gin);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) gin[k] = 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float gk = gin[k];
        if (win && best[k] == q) gk += gz[k];
        const float d = tv[q][k] > 0.f ? gk : gk * slope;
        o[k] = c1[k] * (d - c2[k] - (yv[q][k] - mu[k]) * is[k] * c3[k]);
      }
      store8(dy + (long long)pix * C + g * 8, o);
    }
  }
}

// bf16 storage, 8 channels (16 bytes) per lane: with 4 channels a lane moves 8 bytes and a wave instruction 512 -- the bf16
// twins of the HBM-bound kernels then reach 3.5-4.1 TB/s where their fp32 versions (16 bytes per lane) reach 5.1-5.4
__global__ void bn_bwd_apply8_kernel(bf16_t* __restrict__ dt, const bf16_t* __restrict__ y, long long y_ld,
                                     const float* __restrict__ coef, const float* __restrict__ mean,
                                     const float* __restrict__ invstd, int C, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int cg = C >> 3;
  const int g = (int)(idx % cg);
  const long long pix = idx / cg;
  const uint4 yu = *reinterpret_cast<const uint4*>(y + pix * y_ld + g * 8);
  uint4 du = *reinterpret_cast<const uint4*>(dt + pix * C + g * 8);
  const unsigned yw[4] = {yu.x, yu.y, yu.z, yu.w};
  unsigned dw[4] = {du.x, du.y, du.z, du.w};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = g * 8 + h * 4;
    const f32x4 c1 = ld4(coef + c), c2 = ld4(coef + C + c), c3 = ld4(coef + 2 * C + c);
    const f32x4 mu = ld4(mean + c), is = ld4(invstd + c);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const unsigned yv2 = yw[h * 2 + k], dv2 = dw[h * 2 + k];
      const float y0 = __uint_as_float(yv2 << 16), y1 = __uint_as_float(yv2 & 0xffff0000u);
      const float d0 = __uint_as_float(dv2 << 16), d1 = __uint_as_float(dv2 & 0xffff0000u);
      const int e = 2 * k;
      const float r0 = c1[e] * (d0 - c2[e] - (y0 - mu[e]) * is[e] * c3[e]);
      const float r1 = c1[e + 1] * (d1 - c2[e + 1] - (y1 - mu[e + 1]) * is[e + 1] * c3[e + 1]);
      dw[h * 2 + k] = (unsigned)fsd_ew::f32_to_bf16(r0) | ((unsigned)fsd_ew::f32_to_bf16(r1) << 16);
    }
  }
  du.x = dw[0]; du.y = dw[1]; du.z = dw[2]; du.w = dw[3];
  *reinterpret_cast<uint4*>(dt + pix * C + g * 8) = du;
}

// column sums of a (rows, ld) matrix, any C: partial[blocks][C][2] with the second slot zero
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ m, long long ld, float* __restrict__ partial, int C, long long rows,
                              int ppb) {
  const int c = blockIdx.y * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const long long r0 = (long long)blockIdx.x * ppb;
  float s = 0.f;
  for (int i = 0; i < ppb && r0 + i < rows; ++i) s += ld1<T>(m + (r0 + i) * ld + c);
  partial[((long long)blockIdx.x * C + c) * 2] = s;
  partial[((long long)blockIdx.x * C + c) * 2 + 1] = 0.f;
}

template <typename T>
__global__ void reorg_bwd_kernel(const T* __restrict__ dout, long long dout_ld, T* __restrict__ dx, long long dx_ld,
                                 int H, int W, int C, int s, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int cg = C / 4;
  const int g = (int)(idx % cg);
  long long t = idx / cg;
  const int ix = (int)(t % W); t /= W;
  const int iy = (int)(t % H);
  const long long b = t / H;
  const int OH = H / s, OW = W / s;
  const int oi = iy / s, di = iy - oi * s, oj = ix / s, dj = ix - oj * s;
  const f32x4 v = ld4(dout + ((b * OH + oi) * (long long)OW + oj) * dout_ld + (di * s + dj) * C + g * 4);
  st4<T>(dx + ((b * H + iy) * (long long)W + ix) * dx_ld + g * 4, v);
}

template <typename T>
__global__ void global_max_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ argmax, T* __restrict__ dx,
                                      long long dx_ld, int HW, int C, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (b, pixel, c)
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const long long t = idx / C;
  const int pix = (int)(t % HW);
  const long long b = t / HW;
  st1<T>(dx + (b * HW + pix) * dx_ld + c, argmax[b * C + c] == pix ? dout[b * C + c] : 0.f);
}

template <typename T>
__global__ void global_avg_bwd_kernel(const float* __restrict__ dout, T* __restrict__ dx, long long dx_ld, int HW, int C,
                                      float inv_hw, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (b, pixel, c)
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const long long t = idx / C;
  st1<T>(dx + t * dx_ld + c, dout[(t / HW) * C + c] * inv_hw);
}

template <typename T>
__global__ void add_inplace_kernel(T* __restrict__ dst, long long dst_ld, const T* __restrict__ src, long long src_ld,
                                   int C, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const long long r = idx / C;
  st1<T>(dst + r * dst_ld + c, ld1<T>(dst + r * dst_ld + c) + ld1<T>(src + r * src_ld + c));
}

// d head_w[o,c] = sum_n dWeff[n*O+o, c] * dyn[n,c];  d dyn[n,c] = sum_o dWeff[n*O+o, c] * head_w[o,c]
__global__ void head_unfold_kernel(const float* __restrict__ dweff, const float* __restrict__ head_w,
                                   const float* __restrict__ dyn, float* __restrict__ d_head_w, float* __restrict__ d_dyn,
                                   int n_cls, int O, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int which = blockIdx.y;           // [0, O): one head row;  [O, O + n_cls): one class vector
  if (which < O) {
    float s = 0.f;
    for (int n = 0; n < n_cls; ++n) s += dweff[((long long)n * O + which) * C + c] * dyn[(long long)n * C + c];
    d_head_w[(long long)which * C + c] = s;
  } else {
    const int n = which - O;
    float s = 0.f;
    for (int o = 0; o < O; ++o) s += dweff[((long long)n * O + o) * C + c] * head_w[(long long)o * C + c];
    d_dyn[(long long)n * C + c] = s;
  }
}

// SGD with momentum + L2 weight decay over one flat fp32 buffer (torch.optim.SGD semantics)
__global__ void sgd_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ mom, float lr,
                           float momentum, float weight_decay, int first_step, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const float wi = w[i];
    const float d = g[i] + weight_decay * wi;
    const float b = first_step ? d : momentum * mom[i] + d;
    mom[i] = b;
    w[i] = wi - lr * b;
  }
}

}  // namespace

// small layers get shorter blocks so that the launch still covers the chip
inline int pix_per_block(long long pixels) { return pixels <= 65536 ? 64 : kPixPerBlock; }

inline int cells_per_block(long long cells) { return cells <= 16384 ? 32 : kPixPerBlock / 4; }

extern "C" int fsd_act_bwd_rows(long long pixels) {
  const int ppb = pix_per_block(pixels);
  return (int)((pixels + ppb - 1) / ppb);
}

extern "C" int fsd_bn_act_pool_bwd_rows(int batch, int height, int width, int pool) {
  if (pool == 1) {
    const long long cells = (long long)batch * ((height + 1) / 2) * ((width + 1) / 2);
    const int cpb = cells_per_block(cells);
    return (int)((cells + cpb - 1) / cpb);
  }
  return fsd_act_bwd_rows((long long)batch * height * width);
}

extern "C" size_t fsd_reduce_workspace_bytes(int channels) { return (size_t)kSlots * channels * 2 * sizeof(double); }

namespace {

template <typename T>
int bn_act_pool_bwd_impl(const T* dz, long long dz_ld, const T* dz_full, long long dz_full_ld, const T* y, long long y_ld,
                         const float* scale, const float* shift, const float* mean, const float* invstd, float slope,
                         int pool, T* dt, float* partial, int batch, int height, int width, int channels,
                         hipStream_t stream) {
  (void)hipGetLastError();
  if (!dz || !y || !partial || batch < 1 || channels < 4 || (channels & 3) || (dz_ld & 3) || (y_ld & 3))
    return FSD_ERR_ARG;
  if (dz_full && (dz_full_ld & 3)) return FSD_ERR_ARG;
  if (pool < 0 || pool > 2) return FSD_ERR_UNSUPPORTED;
  ActBwdArgsT<T> a;
  a.dz = dz; a.dz_full = dz_full; a.y = y; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd;
  a.dt = dt; a.partial = partial; a.dz_ld = dz_ld; a.dzf_ld = dz_full_ld; a.y_ld = y_ld;
  a.H = height; a.W = width; a.OH = pool == 1 ? height / 2 : height; a.OW = pool == 1 ? width / 2 : width;
  a.C = channels; a.pool = pool; a.pixels = (long long)batch * height * width; a.slope = slope;
  const int cg = channels / 4;
  const int gl = cg <= 8 ? 8 : cg <= 16 ? 16 : cg <= 32 ? 32 : 64;    // channel-group lanes per block
  // algorithmic bytes: read dz (+ dz_full) and y, write dt (unless statistics only)
  fsd_prof::Scope prof(fsd_prof::kActBwd, (double)sizeof(T) * channels * ((double)batch * a.OH * a.OW + ((dz_full ? 2.0 : 1.0) + (dt ? 1.0 : 0.0)) * a.pixels), stream);
  if (pool == 1) {
    // window-major: a block covers cells_per_block cells = one partial row (fsd_bn_act_pool_bwd_rows)
    const long long cells = (long long)batch * ((height + 1) / 2) * ((width + 1) / 2);
    a.ppb = 4 * cells_per_block(cells);
    const dim3 grid(blocks_for(cells, a.ppb / 4), (cg + gl - 1) / gl);
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (!dt && !dz_full && channels % 8 == 0 && dz_ld % 8 == 0 && y_ld % 8 == 0 && (reinterpret_cast<uintptr_t>(dz) & 15) == 0 &&
          (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
        // statistics only: eight channels per lane, the same cells per block (= the same partial rows)
        const int cg8 = channels / 8;
        const int gl8 = cg8 <= 8 ? 8 : cg8 <= 16 ? 16 : cg8 <= 32 ? 32 : 64;
        const dim3 grid8(blocks_for(cells, a.ppb / 4), (cg8 + gl8 - 1) / gl8);
        if (gl8 == 8) FSD_LAUNCH((act_stats_pool8_kernel<8>), grid8, dim3(256), 0, stream, a);
        else if (gl8 == 16) FSD_LAUNCH((act_stats_pool8_kernel<16>), grid8, dim3(256), 0, stream, a);
        else if (gl8 == 32) FSD_LAUNCH((act_stats_pool8_kernel<32>), grid8, dim3(256), 0, stream, a);
        else FSD_LAUNCH((act_stats_pool8_kernel<64>), grid8, dim3(256), 0, stream, a);
        return (int)hipGetLastError();
      }
    }
    if (gl == 8) FSD_LAUNCH((act_bwd_pool2_kernel<T, 8>), grid, dim3(256), 0, stream, a);
    else if (gl == 16) FSD_LAUNCH((act_bwd_pool2_kernel<T, 16>), grid, dim3(256), 0, stream, a);
    else if (gl == 32) FSD_LAUNCH((act_bwd_pool2_kernel<T, 32>), grid, dim3(256), 0, stream, a);
    else FSD_LAUNCH((act_bwd_pool2_kernel<T, 64>), grid, dim3(256), 0, stream, a);
    return (int)hipGetLastError();
  }
  a.ppb = pix_per_block(a.pixels);
  const dim3 grid(blocks_for(a.pixels, a.ppb), (cg + gl - 1) / gl);
  if constexpr (std::is_same<T, float>::value) {
    if (!dt && pool == 0 && !dz_full) {
      if (gl == 8) FSD_LAUNCH((act_stats_kernel<8>), grid, dim3(256), 0, stream, a);
      else if (gl == 16) FSD_LAUNCH((act_stats_kernel<16>), grid, dim3(256), 0, stream, a);
      else if (gl == 32) FSD_LAUNCH((act_stats_kernel<32>), grid, dim3(256), 0, stream, a);
      else FSD_LAUNCH((act_stats_kernel<64>), grid, dim3(256), 0, stream, a);
      return (int)hipGetLastError();
    }
  }
  if constexpr (std::is_same<T, bf16_t>::value) {
    static const char* env8 = FSD_TUNE("FSD_ACT_STATS8");       // tuning aid: 0 = the generic 4-channel kernel
    if (!dt && pool == 0 && !dz_full && !(env8 && env8[0] == '0') && channels % 8 == 0 && dz_ld % 8 == 0 && y_ld % 8 == 0 &&
        (reinterpret_cast<uintptr_t>(dz) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
      const int cg8 = channels / 8;
      const int gl8 = cg8 <= 8 ? 8 : cg8 <= 16 ? 16 : cg8 <= 32 ? 32 : 64;
      const dim3 grid8(blocks_for(a.pixels, a.ppb), (cg8 + gl8 - 1) / gl8);
      if (gl8 == 8) FSD_LAUNCH((act_stats8_kernel<8>), grid8, dim3(256), 0, stream, a);
      else if (gl8 == 16) FSD_LAUNCH((act_stats8_kernel<16>), grid8, dim3(256), 0, stream, a);
      else if (gl8 == 32) FSD_LAUNCH((act_stats8_kernel<32>), grid8, dim3(256), 0, stream, a);
      else FSD_LAUNCH((act_stats8_kernel<64>), grid8, dim3(256), 0, stream, a);
      return (int)hipGetLastError();
    }
  }
  if (gl == 8) FSD_LAUNCH((act_bwd_kernel<T, 8>), grid, dim3(256), 0, stream, a);
  else if (gl == 16) FSD_LAUNCH((act_bwd_kernel<T, 16>), grid, dim3(256), 0, stream, a);
  else if (gl == 32) FSD_LAUNCH((act_bwd_kernel<T, 32>), grid, dim3(256), 0, stream, a);
  else FSD_LAUNCH((act_bwd_kernel<T, 64>), grid, dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int fsd_bn_act_pool_bwd(const float* dz, long long dz_ld, const float* dz_full, long long dz_full_ld,
                                   const float* y, long long y_ld, const float* scale, const float* shift,
                                   const float* mean, const float* invstd, float slope, int pool, float* dt,
                                   float* partial, int batch, int height, int width, int channels,
                                   hipStream_t stream) {
  return bn_act_pool_bwd_impl<float>(dz, dz_ld, dz_full, dz_full_ld, y, y_ld, scale, shift, mean, invstd, slope, pool, dt,
                                     partial, batch, height, width, channels, stream);
}

extern "C" int fsd_bn_act_pool_bwd_h(const void* dz, long long dz_ld, const void* dz_full, long long dz_full_ld,
                                     const void* y, long long y_ld, const float* scale, const float* shift,
                                     const float* mean, const float* invstd, float slope, int pool, void* dt,
                                     float* partial, int batch, int height, int width, int channels,
                                     hipStream_t stream) {
  return bn_act_pool_bwd_impl<bf16_t>(static_cast<const bf16_t*>(dz), dz_ld, static_cast<const bf16_t*>(dz_full), dz_full_ld,
                                      static_cast<const bf16_t*>(y), y_ld, scale, shift, mean, invstd, slope, pool,
                                      static_cast<bf16_t*>(dt), partial, batch, height, width, channels, stream);
}

extern "C" int fsd_bn_bwd_finalize(const float* partial, int rows, long long count, int channels, const float* scale,
                                   float* dgamma, float* dbeta, float* coef, void* workspace, hipStream_t stream) {
  (void)hipGetLastError();
  if (!partial || !workspace || rows < 1 || channels < 1 || count < 1) return FSD_ERR_ARG;
  const int n_slots = rows < kSlots ? rows : kSlots;
  if (rows <= kSlots) {                 // one row per slot: finalize reads the partial sums directly
    FSD_LAUNCH(bn_bwd_finalize_kernel<float>, dim3((channels + 31) / 32), dim3(256), 0, stream, partial, n_slots,
                       (double)count, channels, scale, dgamma, dbeta, coef);
    return (int)hipGetLastError();
  }
  const int two_c = 2 * channels;
  launch_reduce_partials(partial, reinterpret_cast<double*>(workspace), rows, two_c, n_slots, stream);
  FSD_LAUNCH(bn_bwd_finalize_kernel<double>, dim3((channels + 31) / 32), dim3(256), 0, stream,
                     reinterpret_cast<const double*>(workspace), n_slots, (double)count, channels, scale, dgamma, dbeta,
                     coef);
  return (int)hipGetLastError();
}

namespace {

template <typename T>
int bn_bwd_apply_impl(T* dt, const T* y, long long y_ld, const float* coef, const float* mean, const float* invstd,
                      long long pixels, int channels, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dt || !y || !coef || !mean || !invstd || (channels & 3) || (y_ld & 3)) return FSD_ERR_ARG;
  fsd_prof::Scope prof(fsd_prof::kActBwd, (double)sizeof(T) * channels * 3.0 * pixels, stream);      // read dt, y; write dy
  if constexpr (std::is_same<T, bf16_t>::value) {
    static const char* env = FSD_TUNE("FSD_EW_WIDE");                  // tuning aid: 0 = 4 channels per lane
    if (channels % 8 == 0 && y_ld % 8 == 0 && !(reinterpret_cast<uintptr_t>(dt) & 15) && !(reinterpret_cast<uintptr_t>(y) & 15) &&
        !(env && env[0] == '0')) {
      const long long total8 = pixels * (channels / 8);
      FSD_LAUNCH(bn_bwd_apply8_kernel, dim3(blocks_for(total8, 256)), dim3(256), 0, stream, dt, y, y_ld, coef, mean,
                         invstd, channels, total8);
      return (int)hipGetLastError();
    }
  }
  const long long total = pixels * (channels / 4);
  FSD_LAUNCH(bn_bwd_apply_kernel<T>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, dt, y, y_ld, coef, mean,
                     invstd, channels, total);
  return (int)hipGetLastError();
}

template <typename T>
int colsum_impl(const T* m, long long ld, float* partial, long long rows, int channels, hipStream_t stream) {
  (void)hipGetLastError();
  if (!m || !partial || rows < 1 || channels < 1) return FSD_ERR_ARG;
  const int ppb = pix_per_block(rows);      // partial rows = fsd_act_bwd_rows(rows)
  FSD_LAUNCH(colsum_kernel<T>, dim3(blocks_for(rows, ppb), (channels + 255) / 256), dim3(256), 0, stream, m,
                     ld, partial, channels, rows, ppb);
  return (int)hipGetLastError();
}

template <typename T>
int reorg_bwd_impl(const T* dout, long long dout_ld, T* dx, long long dx_ld, int batch, int height, int width, int channels,
                   int stride, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dout || !dx || stride < 1 || height % stride || width % stride || (channels & 3) || (dout_ld & 3) || (dx_ld & 3))
    return FSD_ERR_ARG;
  const long long total = (long long)batch * height * width * (channels / 4);
  FSD_LAUNCH(reorg_bwd_kernel<T>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, dout, dout_ld, dx, dx_ld,
                     height, width, channels, stride, total);
  return (int)hipGetLastError();
}

template <typename T>
int global_maxpool_bwd_impl(const float* dout, const int* argmax, T* dx, long long dx_ld, int batch, int height, int width,
                            int channels, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dout || !argmax || !dx) return FSD_ERR_ARG;
  const long long total = (long long)batch * height * width * channels;
  FSD_LAUNCH(global_max_bwd_kernel<T>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, dout, argmax, dx, dx_ld,
                     height * width, channels, total);
  return (int)hipGetLastError();
}

template <typename T>
int add_inplace_impl(T* dst, long long dst_ld, const T* src, long long src_ld, long long rows, int channels,
                     hipStream_t stream) {
  (void)hipGetLastError();
  if (!dst || !src || rows < 1 || channels < 1) return FSD_ERR_ARG;
  const long long total = rows * channels;
  FSD_LAUNCH(add_inplace_kernel<T>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, dst, dst_ld, src, src_ld,
                     channels, total);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int fsd_bn_bwd_apply(float* dt, const float* y, long long y_ld, const float* coef, const float* mean,
                                const float* invstd, long long pixels, int channels, hipStream_t stream) {
  return bn_bwd_apply_impl<float>(dt, y, y_ld, coef, mean, invstd, pixels, channels, stream);
}
extern "C" int fsd_bn_bwd_apply_h(void* dt, const void* y, long long y_ld, const float* coef, const float* mean,
                                  const float* invstd, long long pixels, int channels, hipStream_t stream) {
  return bn_bwd_apply_impl<bf16_t>(static_cast<bf16_t*>(dt), static_cast<const bf16_t*>(y), y_ld, coef, mean, invstd, pixels,
                                   channels, stream);
}

extern "C" int fsd_bn_bwd_apply_g(const float* dz, long long dz_ld, const float* dz_full, long long dz_full_ld, const float* y,
                                 long long y_ld, const float* scale, const float* shift, float slope, int pool,
                                 const float* coef, const float* mean, const float* invstd, float* dy, int batch, int height,
                                 int width, int channels, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dz || !y || !coef || !mean || !invstd || !dy || batch < 1 || height < 1 || width < 1 || channels < 4 || (channels & 3) ||
      (dz_ld & 3) || (y_ld & 3) || (dz_full && (dz_full_ld & 3)))
    return FSD_ERR_ARG;
  if (pool != 0 && pool != 1) return FSD_ERR_UNSUPPORTED;
  const int OH = pool ? height / 2 : height, OW = pool ? width / 2 : width;
  const long long pixels = (long long)batch * height * width;
  // algorithmic bytes: read dz (+ dz_full) and y, write dy
  fsd_prof::Scope prof(fsd_prof::kActBwd, 4.0 * channels * ((double)batch * OH * OW + (dz_full ? 3.0 : 2.0) * pixels), stream);
  if (pool == 0) {
    const long long total = pixels * (channels / 4);                      // 4 pixels per thread (UC in the kernel)
    FSD_LAUNCH(bn_bwd_apply_g_kernel<0>, dim3(blocks_for((pixels + 3) / 4 * (channels / 4), 256)), dim3(256), 0, stream, dz, dz_ld, dz_full,
                       dz_full_ld, y, y_ld, scale, shift, slope, coef, mean, invstd, dy, height, width, OH, OW, channels, total);
  } else {
    const long long total = (long long)batch * ((height + 1) / 2) * ((width + 1) / 2) * (channels / 4);
    FSD_LAUNCH(bn_bwd_apply_g_kernel<1>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, dz, dz_ld, dz_full,
                       dz_full_ld, y, y_ld, scale, shift, slope, coef, mean, invstd, dy, height, width, OH, OW, channels, total);
  }
  return (int)hipGetLastError();
}

extern "C" int fsd_bn_bwd_apply_g_h(const void* dz, long long dz_ld, const void* dz_full, long long dz_full_ld, const void* y,
                                   long long y_ld, const float* scale, const float* shift, float slope, int pool,
                                   const float* coef, const float* mean, const float* invstd, void* dy, int batch, int height,
                                   int width, int channels, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dz || !y || !coef || !mean || !invstd || !dy || batch < 1 || height < 1 || width < 1) return FSD_ERR_ARG;
  if (channels < 8 || (channels & 7) || (dz_ld & 7) || (y_ld & 7) || (dz_full && (dz_full_ld & 7))) return FSD_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(dz) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(dy) & 15) ||
      (dz_full && (reinterpret_cast<uintptr_t>(dz_full) & 15)))
    return FSD_ERR_UNSUPPORTED;
  if (pool != 0 && pool != 1) return FSD_ERR_UNSUPPORTED;
  const int OH = pool ? height / 2 : height, OW = pool ? width / 2 : width;
  const long long pixels = (long long)batch * height * width;
  fsd_prof::Scope prof(fsd_prof::kActBwd, 2.0 * channels * ((double)batch * OH * OW + (dz_full ? 3.0 : 2.0) * pixels), stream);
  if (pixels * (channels / 8) >= 0xffffffffLL) return FSD_ERR_UNSUPPORTED;      // 32-bit thread / pixel indices in the kernels
  const bf16_t* dzh = static_cast<const bf16_t*>(dz);
  const bf16_t* dfh = static_cast<const bf16_t*>(dz_full);
  const bf16_t* yh = static_cast<const bf16_t*>(y);
  bf16_t* dyh = static_cast<bf16_t*>(dy);
  if (pool == 0) {      // a thread = 8 channels of 4 consecutive pixels / of one cell (UC in the kernel)
    const long long threads = (pixels + 3) / 4 * (channels / 8);
    FSD_LAUNCH(bn_bwd_apply_g8_kernel<0>, dim3(blocks_for(threads, 256)), dim3(256), 0, stream, dzh, dz_ld, dfh, dz_full_ld,
                       yh, y_ld, scale, shift, slope, coef, mean, invstd, dyh, height, width, OH, OW, channels, pixels);
  } else {
    const long long cells = (long long)batch * ((height + 1) / 2) * ((width + 1) / 2);
    const long long threads = cells * (channels / 8);
    FSD_LAUNCH(bn_bwd_apply_g8_kernel<1>, dim3(blocks_for(threads, 256)), dim3(256), 0, stream, dzh, dz_ld, dfh, dz_full_ld,
                       yh, y_ld, scale, shift, slope, coef, mean, invstd, dyh, height, width, OH, OW, channels, cells);
  }
  return (int)hipGetLastError();
}

extern "C" int fsd_colsum_partials(const float* m, long long ld, float* partial, long long rows, int channels,
                                   hipStream_t stream) {
  return colsum_impl<float>(m, ld, partial, rows, channels, stream);
}
extern "C" int fsd_colsum_partials_h(const void* m, long long ld, float* partial, long long rows, int channels,
                                     hipStream_t stream) {
  return colsum_impl<bf16_t>(static_cast<const bf16_t*>(m), ld, partial, rows, channels, stream);
}

extern "C" int fsd_reorg_bwd(const float* dout, long long dout_ld, float* dx, long long dx_ld, int batch, int height,
                             int width, int channels, int stride, hipStream_t stream) {
  return reorg_bwd_impl<float>(dout, dout_ld, dx, dx_ld, batch, height, width, channels, stride, stream);
}
extern "C" int fsd_reorg_bwd_h(const void* dout, long long dout_ld, void* dx, long long dx_ld, int batch, int height,
                               int width, int channels, int stride, hipStream_t stream) {
  return reorg_bwd_impl<bf16_t>(static_cast<const bf16_t*>(dout), dout_ld, static_cast<bf16_t*>(dx), dx_ld, batch, height,
                                width, channels, stride, stream);
}

extern "C" int fsd_global_maxpool_bwd(const float* dout, const int* argmax, float* dx, long long dx_ld, int batch,
                                      int height, int width, int channels, hipStream_t stream) {
  return global_maxpool_bwd_impl<float>(dout, argmax, dx, dx_ld, batch, height, width, channels, stream);
}
extern "C" int fsd_global_maxpool_bwd_h(const float* dout, const int* argmax, void* dx, long long dx_ld, int batch,
                                        int height, int width, int channels, hipStream_t stream) {
  return global_maxpool_bwd_impl<bf16_t>(dout, argmax, static_cast<bf16_t*>(dx), dx_ld, batch, height, width, channels, stream);
}

extern "C" int fsd_global_avgpool_bwd(const float* dout, void* dx, int dx_bf16, long long dx_ld, int batch, int height, int width,
                                      int channels, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dout || !dx || batch < 1 || height < 1 || width < 1 || channels < 1) return FSD_ERR_ARG;
  const int hw = height * width;
  const long long total = (long long)batch * hw * channels;
  if (dx_bf16)
    FSD_LAUNCH(global_avg_bwd_kernel<bf16_t>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, dout, static_cast<bf16_t*>(dx),
               dx_ld, hw, channels, 1.f / (float)hw, total);
  else
    FSD_LAUNCH(global_avg_bwd_kernel<float>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, dout, static_cast<float*>(dx),
               dx_ld, hw, channels, 1.f / (float)hw, total);
  return (int)hipGetLastError();
}

extern "C" int fsd_add_inplace(float* dst, long long dst_ld, const float* src, long long src_ld, long long rows,
                               int channels, hipStream_t stream) {
  return add_inplace_impl<float>(dst, dst_ld, src, src_ld, rows, channels, stream);
}
extern "C" int fsd_add_inplace_h(void* dst, long long dst_ld, const void* src, long long src_ld, long long rows,
                                 int channels, hipStream_t stream) {
  return add_inplace_impl<bf16_t>(static_cast<bf16_t*>(dst), dst_ld, static_cast<const bf16_t*>(src), src_ld, rows, channels,
                                  stream);
}

extern "C" int fsd_head_unfold_bwd(const float* dweff, const float* head_w, const float* dyn, float* d_head_w,
                                   float* d_dyn, int n_cls, int out_ch, int channels, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dweff || !head_w || !dyn || !d_head_w || !d_dyn) return FSD_ERR_ARG;
  FSD_LAUNCH(head_unfold_kernel, dim3((channels + 255) / 256, out_ch + n_cls), dim3(256), 0, stream, dweff,
                     head_w, dyn, d_head_w, d_dyn, n_cls, out_ch, channels);
  return (int)hipGetLastError();
}

extern "C" int fsd_sgd_step(float* w, const float* grad, float* momentum_buf, float lr, float momentum,
                            float weight_decay, int first_step, long long count, hipStream_t stream) {
  (void)hipGetLastError();
  if (!w || !grad || !momentum_buf || count < 0) return FSD_ERR_ARG;
  if (count == 0) return FSD_OK;
  long long blocks = (count + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  fsd_prof::Scope prof(fsd_prof::kSgd, 20.0 * count, stream);          // w, g, m read; w, m written
  FSD_LAUNCH(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, grad, momentum_buf, lr, momentum,
                     weight_decay, first_step, count);
  return (int)hipGetLastError();
}
