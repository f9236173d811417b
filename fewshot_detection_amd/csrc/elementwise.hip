// HBM-bound companions of the conv kernel: BatchNorm statistics finalisation, the fused
// affine + leaky + maxpool pass, layout transposes, reorg, global max pool and the
// channel-reweighting helpers.  All NHWC / float4-vectorised where the layout allows.
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
using fsd_ew::f32x8;
using fsd_ew::ld1;
using fsd_ew::ld4;
using fsd_ew::st1;
using fsd_ew::st4;
constexpr int kBnSlots = 256;

inline unsigned blocks_for(long long n, int per) {
  long long b = (n + per - 1) / per;
  return (unsigned)(b < 1 ? 1 : b);
}

// ---- BN statistics -----------------------------------------------------------------------
// stage 1: fold the [tiles][C][2] float partials of the conv epilogue into [slots][C][2] doubles.
// Block = RL row lanes x CL columns (CL = 256 / RL); row lanes are folded through LDS in a fixed order, so the
// result does not depend on scheduling.
template <int RL>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ partial, double* __restrict__ slots,
                                                        int tiles, int two_c, int n_slots) {
  constexpr int CL = 256 / RL;
  __shared__ double s_acc[RL][CL];
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int e = blockIdx.y * CL + cl;
  const int s = blockIdx.x;
  double acc = 0.0;
  if (e < two_c) {
    const long long step = (long long)n_slots * RL;
    long long t = s + (long long)n_slots * rl;
    for (; t + 3 * step < tiles; t += 4 * step) {
      const float v0 = partial[t * two_c + e], v1 = partial[(t + step) * two_c + e];
      const float v2 = partial[(t + 2 * step) * two_c + e], v3 = partial[(t + 3 * step) * two_c + e];
      acc += (double)v0; acc += (double)v1; acc += (double)v2; acc += (double)v3;
    }
    for (; t < tiles; t += step) acc += (double)partial[t * two_c + e];
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

inline void launch_bn_reduce(const float* partial, double* slots, int tiles, int two_c, int n_slots, hipStream_t stream) {
  if (two_c <= 64)
    FSD_LAUNCH(bn_reduce_kernel<4>, dim3(n_slots, (two_c + 63) / 64), dim3(256), 0, stream, partial, slots, tiles, two_c, n_slots);
  else if (two_c <= 128)
    FSD_LAUNCH(bn_reduce_kernel<2>, dim3(n_slots, (two_c + 127) / 128), dim3(256), 0, stream, partial, slots, tiles, two_c, n_slots);
  else
    FSD_LAUNCH(bn_reduce_kernel<1>, dim3(n_slots, (two_c + 255) / 256), dim3(256), 0, stream, partial, slots, tiles, two_c, n_slots);
}

// S = double: `slots` from bn_reduce_kernel.  S = float: the per-tile partial sums themselves, when there are no more rows
// than slots (a slot would hold exactly one row: same values, same order of summation, one launch less).
template <typename S>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const S* __restrict__ slots, int n_slots, double count, int channels,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* running_mean, float* running_var, float momentum, float eps,
                                   int training, float* __restrict__ scale, float* __restrict__ shift,
                                   float* save_mean, float* save_invstd) {
  // block = 32 channels x 8 slot lanes (coalesced along channels); lanes folded through LDS in a fixed order
  __shared__ double s_part[8][32][2];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s = 0.0, q = 0.0;
  if (training) {
    if (c < channels) {
#pragma unroll 4
      for (int k = sl; k < n_slots; k += 8) {
        s += (double)slots[((long long)k * channels + c) * 2 + 0];
        q += (double)slots[((long long)k * channels + c) * 2 + 1];
      }
    }
    s_part[sl][cl][0] = s;
    s_part[sl][cl][1] = q;
    __syncthreads();
  }
  if (sl != 0 || c >= channels) return;
  double mean, var;
  if (training) {
#pragma unroll
    for (int l = 1; l < 8; ++l) { s += s_part[l][cl][0]; q += s_part[l][cl][1]; }
    mean = s / count;
    var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
    running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
  } else {
    mean = (double)running_mean[c];
    var = (double)running_var[c];
  }
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const double g = gamma ? (double)gamma[c] : 1.0, b = beta ? (double)beta[c] : 0.0;
  scale[c] = (float)(g * invstd);
  shift[c] = (float)(b - mean * g * invstd);
  if (save_mean) save_mean[c] = (float)mean;
  if (save_invstd) save_invstd[c] = (float)invstd;
}

// ---- affine + activation + pool ------------------------------------------------------------
__device__ __forceinline__ f32x4 affine_act(f32x4 v, f32x4 sc, f32x4 sh, float slope) {
  f32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float t = __builtin_fmaf(v[k], sc[k], sh[k]);      // one rounding, spelled out: the consumers that form this
    r[k] = t > 0.f ? t : t * slope;                          // activation on load (conv_common.hpp affine_act4) must match
  }
  return r;
}

template <typename T, int POOL>
__global__ __launch_bounds__(256) void bn_act_pool_kernel(const T* __restrict__ y, long long y_ld,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float slope,
                                                          T* __restrict__ z, long long z_ld, int H, int W,
                                                          int OH, int OW, int cg, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int g = (int)(idx % cg);
  const long long opix = idx / cg;
  const f32x4 sc = scale ? *reinterpret_cast<const f32x4*>(scale + g * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
  const f32x4 sh = shift ? *reinterpret_cast<const f32x4*>(shift + g * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 out;
  if constexpr (POOL == 0) {
    out = affine_act(ld4<T>(y + opix * y_ld + g * 4), sc, sh, slope);
  } else {
    const int ox = (int)(opix % OW);
    const long long t = opix / OW;
    const int oy = (int)(t % OH);
    const long long b = t / OH;
    const int y0 = POOL == 1 ? 2 * oy : oy, x0 = POOL == 1 ? 2 * ox : ox;
    const int y1 = (y0 + 1 < H) ? y0 + 1 : H - 1, x1 = (x0 + 1 < W) ? x0 + 1 : W - 1;   // replicate pad (stride 1)
    const T* base = y + (b * H * (long long)W) * y_ld + g * 4;
    const f32x4 v00 = affine_act(ld4<T>(base + ((long long)y0 * W + x0) * y_ld), sc, sh, slope);
    const f32x4 v01 = affine_act(ld4<T>(base + ((long long)y0 * W + x1) * y_ld), sc, sh, slope);
    const f32x4 v10 = affine_act(ld4<T>(base + ((long long)y1 * W + x0) * y_ld), sc, sh, slope);
    const f32x4 v11 = affine_act(ld4<T>(base + ((long long)y1 * W + x1) * y_ld), sc, sh, slope);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = fmaxf(fmaxf(v00[k], v01[k]), fmaxf(v10[k], v11[k]));
  }
  st4<T>(z + opix * z_ld + g * 4, out);
}

// bf16 storage, 8 channels (one 16-byte access) per lane: see ew_types.hpp
template <int POOL>
__global__ __launch_bounds__(256) void bn_act_pool8_kernel(const bf16_t* __restrict__ y, long long y_ld,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, float slope,
                                                           bf16_t* __restrict__ z, long long z_ld, int H, int W,
                                                           int OH, int OW, int cg8, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int g = (int)(idx % cg8);
  const long long opix = idx / cg8;
  const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 sc0 = scale ? *reinterpret_cast<const f32x4*>(scale + g * 8) : one;
  const f32x4 sc1 = scale ? *reinterpret_cast<const f32x4*>(scale + g * 8 + 4) : one;
  const f32x4 sh0 = shift ? *reinterpret_cast<const f32x4*>(shift + g * 8) : zero;
  const f32x4 sh1 = shift ? *reinterpret_cast<const f32x4*>(shift + g * 8 + 4) : zero;
  auto act8 = [&](const bf16_t* p) {
    f32x8 v = fsd_ew::ld8(p);
    v.lo = affine_act(v.lo, sc0, sh0, slope);
    v.hi = affine_act(v.hi, sc1, sh1, slope);
    return v;
  };
  f32x8 out;
  if constexpr (POOL == 0) {
    out = act8(y + opix * y_ld + g * 8);
  } else {
    const int ox = (int)(opix % OW);
    const long long t = opix / OW;
    const int oy = (int)(t % OH);
    const long long b = t / OH;
    const int y0 = POOL == 1 ? 2 * oy : oy, x0 = POOL == 1 ? 2 * ox : ox;
    const int y1 = (y0 + 1 < H) ? y0 + 1 : H - 1, x1 = (x0 + 1 < W) ? x0 + 1 : W - 1;   // replicate pad (stride 1)
    const bf16_t* base = y + (b * H * (long long)W) * y_ld + g * 8;
    const f32x8 v00 = act8(base + ((long long)y0 * W + x0) * y_ld), v01 = act8(base + ((long long)y0 * W + x1) * y_ld);
    const f32x8 v10 = act8(base + ((long long)y1 * W + x0) * y_ld), v11 = act8(base + ((long long)y1 * W + x1) * y_ld);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      out.lo[k] = fmaxf(fmaxf(v00.lo[k], v01.lo[k]), fmaxf(v10.lo[k], v11.lo[k]));
      out.hi[k] = fmaxf(fmaxf(v00.hi[k], v01.hi[k]), fmaxf(v10.hi[k], v11.hi[k]));
    }
  }
  fsd_ew::st8(z + opix * z_ld + g * 8, out);
}

// ---- batched 2-D transpose through LDS (NCHW <-> NHWC) ---------------------------------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void transpose_kernel(const TS* __restrict__ src, long long sbs, long long srs,
                                                        TD* __restrict__ dst, long long dbs, long long drs,
                                                        int rows, int cols) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
  const TS* s = src + (long long)b * sbs;
  TD* d = dst + (long long)b * dbs;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + 8 * k][tx] = ld1<TS>(s + (long long)r * srs + c);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (r < rows && c < cols) st1<TD>(d + (long long)c * drs + r, tile[tx][ty + 8 * k]);
  }
}

// (B, C<=4, HW) planes -> (B*HW, 4) pixels, missing channels zero: one thread per pixel, coalesced plane reads and
// one 16-byte store (the 32x32 LDS transpose wastes 29 of its 32 rows on a 3-channel image)
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                            long long hw, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long b = i / hw, pix = i - b * hw;
  const float* s = src + b * C * hw + pix;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  v[0] = s[0];
  if (C > 1) v[1] = s[hw];
  if (C > 2) v[2] = s[2 * hw];
  if (C > 3) v[3] = s[3 * hw];
  *reinterpret_cast<f32x4*>(dst + i * 4) = v;
}

__global__ void fill_kernel(float* dst, float v, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += step) dst[i] = v;
}

// ---- reorg (space to depth), NHWC ------------------------------------------------------------
template <typename T>
__global__ void reorg_kernel(const T* __restrict__ x, long long x_ld, T* __restrict__ out, long long out_ld,
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
  const f32x4 v = ld4<T>(x + ((b * H + iy) * (long long)W + ix) * x_ld + g * 4);
  st4<T>(out + ((b * OH + oi) * (long long)OW + oj) * out_ld + (di * s + dj) * C + g * 4, v);
}

// ---- global max pool: (B, HW, C) -> (B, C) -----------------------------------------------------
template <typename T>
__global__ void global_max_kernel(const T* __restrict__ x, long long x_ld, float* __restrict__ out,
                                  int* __restrict__ argmax, int HW, int C, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const long long b = idx / C;
  const T* p = x + b * HW * x_ld + c;
  float best = ld1<T>(p);
  int arg = 0;
  for (int i = 1; i < HW; ++i) {
    const float v = ld1<T>(p + (long long)i * x_ld);
    if (v > best || v != v) { best = v; arg = i; }
  }
  out[idx] = best;
  if (argmax) argmax[idx] = arg;
}

// ---- global average pool: (B, HW, C) -> (B, C), F.adaptive_avg_pool2d(x, 1) of pooling.py:29-45 (fp64 sum, one rounding) ----
template <typename T>
__global__ void global_avg_kernel(const T* __restrict__ x, long long x_ld, float* __restrict__ out, int HW, int C,
                                  long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const long long b = idx / C;
  const T* p = x + b * HW * x_ld + c;      // consecutive lanes = consecutive channels of one pixel: coalesced rows
  double s = 0.0;
  for (int i = 0; i < HW; ++i) s += (double)ld1<T>(p + (long long)i * x_ld);
  out[idx] = (float)(s / (double)HW);
}

// ---- channel reweighting ------------------------------------------------------------------
__global__ void dynamic_conv_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out,
                                    int n_cls, int C, int hw, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (b, n, c, hw)
  if (idx >= total) return;
  const int p = (int)(idx % hw);
  long long t = idx / hw;
  const int c = (int)(t % C); t /= C;
  const int n = (int)(t % n_cls);
  const long long b = t / n_cls;
  out[idx] = x[(b * C + c) * hw + p] * w[(long long)n * C + c];
}

__global__ void fold_head_kernel(const float* __restrict__ head_w, const float* __restrict__ head_b,
                                 const float* __restrict__ dyn, float* __restrict__ w_eff, float* __restrict__ b_eff,
                                 int n_cls, int O, int C, int rows_pad, int kpad) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)rows_pad * kpad) return;
  const int row = (int)(idx / kpad), k = (int)(idx - (long long)row * kpad);
  float v = 0.f;
  if (row < n_cls * O && k < C) {
    const int n = row / O, o = row - n * O;
    v = head_w[(long long)o * C + k] * dyn[(long long)n * C + k];
  }
  w_eff[idx] = v;
  if (k == 0 && row < n_cls * O && b_eff) b_eff[row] = head_b ? head_b[row % O] : 0.f;
}

}  // namespace

extern "C" size_t fsd_bn_finalize_workspace_bytes(int channels) {
  return (size_t)kBnSlots * channels * 2 * sizeof(double);
}

extern "C" int fsd_bn_finalize(const float* bn_partial, int row_tiles, long long count, int channels,
                               const float* gamma, const float* beta, float* running_mean, float* running_var,
                               float momentum, float eps, int training, float* scale, float* shift,
                               float* save_mean, float* save_invstd, void* workspace, hipStream_t stream) {
  (void)hipGetLastError();   // drop a stale error left by someone else's earlier call
  if (!scale || !shift || channels < 1 || !running_mean || !running_var) return FSD_ERR_ARG;
  int n_slots = 0;
  if (training) {
    if (!bn_partial || !workspace || row_tiles < 1 || count < 1) return FSD_ERR_ARG;
    n_slots = row_tiles < kBnSlots ? row_tiles : kBnSlots;
    if (row_tiles <= kBnSlots) {          // one row per slot: finalize reads the partial sums directly
      FSD_LAUNCH(bn_finalize_kernel<float>, dim3((channels + 31) / 32), dim3(256), 0, stream, bn_partial, n_slots,
                         (double)count, channels, gamma, beta, running_mean, running_var, momentum, eps, training, scale, shift,
                         save_mean, save_invstd);
      return (int)hipGetLastError();
    }
    const int two_c = 2 * channels;
    launch_bn_reduce(bn_partial, reinterpret_cast<double*>(workspace), row_tiles, two_c, n_slots, stream);
  }
  FSD_LAUNCH(bn_finalize_kernel<double>, dim3((channels + 31) / 32), dim3(256), 0, stream,
                     reinterpret_cast<const double*>(workspace), n_slots, (double)count, channels, gamma, beta,
                     running_mean, running_var, momentum, eps, training, scale, shift, save_mean, save_invstd);
  return (int)hipGetLastError();
}

namespace {

template <typename T>
int bn_act_pool_impl(const T* y, long long y_ld, const float* scale, const float* shift, float slope, int pool, T* z,
                     long long z_ld, int batch, int height, int width, int channels, hipStream_t stream) {
  (void)hipGetLastError();   // drop a stale error left by someone else's earlier call
  if (!y || !z || batch < 1 || channels < 4 || (channels & 3) || (y_ld & 3) || (z_ld & 3)) return FSD_ERR_ARG;
  if (pool < 0 || pool > 2) return FSD_ERR_UNSUPPORTED;
  const int OH = pool == 1 ? height / 2 : height, OW = pool == 1 ? width / 2 : width;
  if (OH < 1 || OW < 1) return FSD_ERR_ARG;
  // algorithmic bytes: read y once, write z once
  fsd_prof::Scope prof(fsd_prof::kActFwd, (double)sizeof(T) * channels * ((double)batch * height * width + (double)batch * OH * OW), stream);
  if constexpr (std::is_same<T, bf16_t>::value) {
    static const char* env = FSD_TUNE("FSD_EW_WIDE");                  // tuning aid: 0 = 4 channels per lane
    if (channels % 8 == 0 && y_ld % 8 == 0 && z_ld % 8 == 0 && !(reinterpret_cast<uintptr_t>(y) & 15) &&
        !(reinterpret_cast<uintptr_t>(z) & 15) && !(env && env[0] == '0')) {
      const int cg8 = channels / 8;
      const long long total8 = (long long)batch * OH * OW * cg8;
      const dim3 grid8(blocks_for(total8, 256)), block8(256);
      if (pool == 0)
        FSD_LAUNCH((bn_act_pool8_kernel<0>), grid8, block8, 0, stream, y, y_ld, scale, shift, slope, z, z_ld, height, width, OH, OW, cg8, total8);
      else if (pool == 1)
        FSD_LAUNCH((bn_act_pool8_kernel<1>), grid8, block8, 0, stream, y, y_ld, scale, shift, slope, z, z_ld, height, width, OH, OW, cg8, total8);
      else
        FSD_LAUNCH((bn_act_pool8_kernel<2>), grid8, block8, 0, stream, y, y_ld, scale, shift, slope, z, z_ld, height, width, OH, OW, cg8, total8);
      return (int)hipGetLastError();
    }
  }
  const int cg = channels / 4;
  const long long total = (long long)batch * OH * OW * cg;
  const dim3 grid(blocks_for(total, 256)), block(256);
  if (pool == 0)
    FSD_LAUNCH((bn_act_pool_kernel<T, 0>), grid, block, 0, stream, y, y_ld, scale, shift, slope, z, z_ld, height, width, OH, OW, cg, total);
  else if (pool == 1)
    FSD_LAUNCH((bn_act_pool_kernel<T, 1>), grid, block, 0, stream, y, y_ld, scale, shift, slope, z, z_ld, height, width, OH, OW, cg, total);
  else
    FSD_LAUNCH((bn_act_pool_kernel<T, 2>), grid, block, 0, stream, y, y_ld, scale, shift, slope, z, z_ld, height, width, OH, OW, cg, total);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int fsd_bn_act_pool_fwd(const float* y, long long y_ld, const float* scale, const float* shift,
                                   float slope, int pool, float* z, long long z_ld, int batch, int height,
                                   int width, int channels, hipStream_t stream) {
  return bn_act_pool_impl<float>(y, y_ld, scale, shift, slope, pool, z, z_ld, batch, height, width, channels, stream);
}

extern "C" int fsd_bn_act_pool_fwd_h(const void* y, long long y_ld, const float* scale, const float* shift,
                                     float slope, int pool, void* z, long long z_ld, int batch, int height,
                                     int width, int channels, hipStream_t stream) {
  return bn_act_pool_impl<bf16_t>(static_cast<const bf16_t*>(y), y_ld, scale, shift, slope, pool, static_cast<bf16_t*>(z),
                                  z_ld, batch, height, width, channels, stream);
}

namespace {

template <typename TS, typename TD>
int transpose_impl(const TS* src, long long src_batch_stride, long long src_row_stride, TD* dst, long long dst_batch_stride,
                   long long dst_row_stride, int batch, int rows, int cols, hipStream_t stream) {
  (void)hipGetLastError();   // drop a stale error left by someone else's earlier call
  if (!src || !dst || batch < 1 || rows < 1 || cols < 1 || batch > 65535) return FSD_ERR_ARG;
  const dim3 grid((cols + 31) / 32, (rows + 31) / 32, batch);
  if (grid.y > 65535) return FSD_ERR_UNSUPPORTED;
  FSD_LAUNCH((transpose_kernel<TS, TD>), grid, dim3(256), 0, stream, src, src_batch_stride, src_row_stride, dst,
                     dst_batch_stride, dst_row_stride, rows, cols);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int fsd_transpose_batched(const float* src, long long src_batch_stride, long long src_row_stride,
                                     float* dst, long long dst_batch_stride, long long dst_row_stride, int batch,
                                     int rows, int cols, hipStream_t stream) {
  return transpose_impl<float, float>(src, src_batch_stride, src_row_stride, dst, dst_batch_stride, dst_row_stride, batch,
                                      rows, cols, stream);
}

extern "C" int fsd_transpose_batched_h(const void* src, int src_bf16, long long src_batch_stride, long long src_row_stride,
                                       void* dst, int dst_bf16, long long dst_batch_stride, long long dst_row_stride,
                                       int batch, int rows, int cols, hipStream_t stream) {
  if (src_bf16 && dst_bf16)
    return transpose_impl<bf16_t, bf16_t>(static_cast<const bf16_t*>(src), src_batch_stride, src_row_stride,
                                          static_cast<bf16_t*>(dst), dst_batch_stride, dst_row_stride, batch, rows, cols, stream);
  if (src_bf16)
    return transpose_impl<bf16_t, float>(static_cast<const bf16_t*>(src), src_batch_stride, src_row_stride,
                                         static_cast<float*>(dst), dst_batch_stride, dst_row_stride, batch, rows, cols, stream);
  if (dst_bf16)
    return transpose_impl<float, bf16_t>(static_cast<const float*>(src), src_batch_stride, src_row_stride,
                                         static_cast<bf16_t*>(dst), dst_batch_stride, dst_row_stride, batch, rows, cols, stream);
  return transpose_impl<float, float>(static_cast<const float*>(src), src_batch_stride, src_row_stride,
                                      static_cast<float*>(dst), dst_batch_stride, dst_row_stride, batch, rows, cols, stream);
}

extern "C" int fsd_nchw_to_nhwc4(const float* src, float* dst, int batch, int channels, long long hw, hipStream_t stream) {
  (void)hipGetLastError();
  if (!src || !dst || batch < 1 || channels < 1 || channels > 4 || hw < 1) return FSD_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(dst) & 15)) return FSD_ERR_ARG;
  const long long total = (long long)batch * hw;
  FSD_LAUNCH(nchw_to_nhwc4_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, stream, src, dst, channels, hw, total);
  return (int)hipGetLastError();
}

extern "C" int fsd_fill(float* dst, float value, long long count, hipStream_t stream) {
  (void)hipGetLastError();   // drop a stale error left by someone else's earlier call
  if (!dst || count < 0) return FSD_ERR_ARG;
  if (count == 0) return FSD_OK;
  long long blocks = (count + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  FSD_LAUNCH(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dst, value, count);
  return (int)hipGetLastError();
}

namespace {

template <typename T>
int reorg_impl(const T* x, long long x_ld, T* out, long long out_ld, int batch, int height, int width, int channels,
               int stride, hipStream_t stream) {
  (void)hipGetLastError();   // drop a stale error left by someone else's earlier call
  if (!x || !out || stride < 1 || height % stride || width % stride || (channels & 3) || (x_ld & 3) || (out_ld & 3))
    return FSD_ERR_ARG;
  const long long total = (long long)batch * height * width * (channels / 4);
  FSD_LAUNCH(reorg_kernel<T>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, x, x_ld, out, out_ld, height,
                     width, channels, stride, total);
  return (int)hipGetLastError();
}

template <typename T>
int global_maxpool_impl(const T* x, long long x_ld, float* out, int* argmax, int batch, int height, int width, int channels,
                        hipStream_t stream) {
  (void)hipGetLastError();   // drop a stale error left by someone else's earlier call
  if (!x || !out || batch < 1 || height < 1 || width < 1 || channels < 1) return FSD_ERR_ARG;
  if (height != width) return FSD_ERR_UNSUPPORTED;   // pooling.py:23-27 assumes a square map
  const long long total = (long long)batch * channels;
  FSD_LAUNCH(global_max_kernel<T>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, x, x_ld, out, argmax,
                     height * width, channels, total);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int fsd_reorg_fwd(const float* x, long long x_ld, float* out, long long out_ld, int batch, int height,
                             int width, int channels, int stride, hipStream_t stream) {
  return reorg_impl<float>(x, x_ld, out, out_ld, batch, height, width, channels, stride, stream);
}

extern "C" int fsd_reorg_fwd_h(const void* x, long long x_ld, void* out, long long out_ld, int batch, int height,
                               int width, int channels, int stride, hipStream_t stream) {
  return reorg_impl<bf16_t>(static_cast<const bf16_t*>(x), x_ld, static_cast<bf16_t*>(out), out_ld, batch, height, width,
                            channels, stride, stream);
}

extern "C" int fsd_global_maxpool_fwd(const float* x, long long x_ld, float* out, int* argmax, int batch, int height,
                                      int width, int channels, hipStream_t stream) {
  return global_maxpool_impl<float>(x, x_ld, out, argmax, batch, height, width, channels, stream);
}

extern "C" int fsd_global_maxpool_fwd_h(const void* x, long long x_ld, float* out, int* argmax, int batch, int height,
                                        int width, int channels, hipStream_t stream) {
  return global_maxpool_impl<bf16_t>(static_cast<const bf16_t*>(x), x_ld, out, argmax, batch, height, width, channels, stream);
}

extern "C" int fsd_global_avgpool_fwd(const void* x, int x_bf16, long long x_ld, float* out, int batch, int height, int width,
                                      int channels, hipStream_t stream) {
  (void)hipGetLastError();
  if (!x || !out || batch < 1 || height < 1 || width < 1 || channels < 1) return FSD_ERR_ARG;
  const long long total = (long long)batch * channels;
  if (x_bf16)
    FSD_LAUNCH(global_avg_kernel<bf16_t>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, static_cast<const bf16_t*>(x), x_ld,
               out, height * width, channels, total);
  else
    FSD_LAUNCH(global_avg_kernel<float>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, static_cast<const float*>(x), x_ld,
               out, height * width, channels, total);
  return (int)hipGetLastError();
}

extern "C" int fsd_dynamic_conv_fwd(const float* x, const float* w, float* out, int batch, int n_cls, int channels,
                                    int hw, hipStream_t stream) {
  (void)hipGetLastError();   // drop a stale error left by someone else's earlier call
  if (!x || !w || !out || batch < 1 || n_cls < 1 || channels < 1 || hw < 1) return FSD_ERR_ARG;
  const long long total = (long long)batch * n_cls * channels * hw;
  FSD_LAUNCH(dynamic_conv_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, stream, x, w, out, n_cls,
                     channels, hw, total);
  return (int)hipGetLastError();
}

extern "C" int fsd_fold_reweight_head(const float* head_w, const float* head_b, const float* dyn,
                                      float* w_eff_packed, float* bias_eff, int n_cls, int out_ch, int channels,
                                      hipStream_t stream) {
  (void)hipGetLastError();   // drop a stale error left by someone else's earlier call
  if (!head_w || !dyn || !w_eff_packed || n_cls < 1 || out_ch < 1 || channels < 4 || (channels & 3)) return FSD_ERR_ARG;
  const int rows_pad = (n_cls * out_ch + 127) / 128 * 128;
  const int kpad = (channels + 31) / 32 * 32;
  const long long total = (long long)rows_pad * kpad;
  FSD_LAUNCH(fold_head_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, stream, head_w, head_b, dyn,
                     w_eff_packed, bias_eff, n_cls, out_ch, channels, rows_pad, kpad);
  return (int)hipGetLastError();
}

namespace {
// 4-byte words; src may be PINNED HOST memory (the GPU reads it over the host link): the per-step upload of the RegionLoss
// targets as a kernel.  A hipMemcpyAsync on a stream with work pending makes the HOST wait for that work on this runtime.
__global__ __launch_bounds__(256) void upload_words_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst,
                                                            long long words) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < words; i += stride)
    dst[i] = __builtin_nontemporal_load(src + i);
}
}  // namespace

extern "C" int fsd_upload_words(const void* src, void* dst, long long words, hipStream_t stream) {
  (void)hipGetLastError();
  if (!src || !dst || words < 0) return FSD_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3) return FSD_ERR_ARG;
  if (words == 0) return 0;
  const long long want = (words + 255) / 256;
  FSD_LAUNCH(upload_words_kernel, dim3((unsigned)(want < 512 ? want : 512)), dim3(256), 0, stream,
             static_cast<const unsigned*>(src), static_cast<unsigned*>(dst), words);
  return (int)hipGetLastError();
}

extern "C" const char* fsd_version(void) { return "fsdet-hip 0.1 (gfx950)"; }
