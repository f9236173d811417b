// Backward pass of a network's FIRST conv block -- 3x3 convolution of a <= 4-channel NHWC4 input, BatchNorm, leaky,
// 2x2 / 2 max pool (darknet L0 and the reweighting net's L0) -- in ONE sweep over its activation.
//
// The unfused sequence moves the largest activation of the network five times: fsd_bn_act_pool_bwd reads dz and y and
// writes dt, the per-channel sums are reduced, fsd_conv3x3_wgrad_c4_bnfused reads dt and y again.  A first layer needs
// no data gradient, so dt is only ever consumed by per-channel sums and by the weight gradient, and the BatchNorm
// backward is affine in dt:
//       dy = c1 (dt - c2 - xhat c3),     c2 = mean(dt),  c3 = mean(dt xhat),  c1 = gamma invstd     (fsd_bn_bwd_finalize)
//   =>  dW[co][col] = sum_p dy[p,co] x[p+tap,ci] = c1[co] ( S1[co][col] - c2[co] S2[col] - c3[co] S3[co][col] )
//       S1 = sum_p dt[p,co] x[p+tap,ci]      S2 = sum_p x[p+tap,ci]      S3 = sum_p xhat[p,co] x[p+tap,ci]
// S1, S2, S3 and the two BatchNorm sums (sum dt, sum dt xhat) are all accumulated by the pass that FORMS dt from
// (dz, y): dt is never written, y is read once.  A small second kernel folds the per-workgroup partials (fp64, fixed
// order) and applies the coefficients.  HBM traffic: dz (a quarter of the activation) + y + the 16-byte input pixels,
// against 2 x dz/dt + 2 x y + dt before (416x416x32 fp32 at B = 64: 1.9 GB instead of 6.0 GB).
//
// MFMA formulation (v_mfma_f32_32x32x2_f32, no LDS staging, as in wgrad_first_kernel): a wave owns a run of pooling
// cells; a cell is two k-steps of two pixels (its top and its bottom row).  Lane (c, h): A operand = channel c of pixel
// column h, B operand = (tap, ci) column c of the same pixel.  The pool winner of a cell needs the four activations of
// the window: two are this lane's (top / bottom of column h), two the partner lane's (lane ^ 32).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "fsdet.h"
#include "profile.hpp"
#include "ew_types.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
using fsd_ew::bf16_t;


struct FirstBwdArgs {
  const void* dz; const void* y;          // float or bf16 (template parameter); leading dimensions in ELEMENTS
  const float* scale; const float* shift; const float* mean; const float* invstd; const float* x;
  float* ws;                               // [blocks][2][Cout][36]: S1, S3
  float* ws2;                              // [blocks][36]: S2
  float* partial;                          // [blocks][Cout][2]: sum dt, sum dt*xhat
  unsigned dz_ld, y_ld, x_ld;
  int H, W, OH, OW, Cout;
  long long cells;
  int cpw;                                 // cells per wave (multiple of 2 * CPG)
  float slope;
};

// SIDE = false: 3 input channels, the 27 (tap, ci) columns fit one 32-wide MFMA tile.
// SIDE = true : 4 input channels, taps 0..7 in the tile and the ninth tap as FMA side sums per lane.
// CPG = cells per load group (one group of loads is in flight ahead of the MFMAs of the previous one).
template <bool SIDE, typename T, int CPG>
__global__ __launch_bounds__(256) void first_bwd_kernel(FirstBwdArgs p) {
  constexpr int kCPG = CPG;
  constexpr unsigned ES = sizeof(T);
  __shared__ float s_out[4][32 * 36 + 36 + 64];       // per wave: a [32][36] tile, then S2[36], then BN sums [32][2]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int co = blockIdx.y * 32 + c;
  const float sc = p.scale[co], sh = p.shift[co], mu = p.mean[co], is = p.invstd[co];
  const int tap = SIDE ? c >> 2 : c / 3, ci = SIDE ? c & 3 : c - 3 * (c / 3);
  const bool col_ok = SIDE || c < 27;
  const int ky = tap / 3, kx = tap - 3 * ky;
  const int dyo = ky - 1, dxo = kx - 1;
  const long long c_begin = ((long long)blockIdx.x * 4 + wave) * p.cpw;
  const long long c_end = c_begin + p.cpw < p.cells ? c_begin + p.cpw : p.cells;
  const int len = (int)(c_end - c_begin);                    // cells of this wave (may be <= 0)
  f32x16 acc1, acc3;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc3[r] = 0.f; }
  float a81[4] = {0.f, 0.f, 0.f, 0.f}, a83[4] = {0.f, 0.f, 0.f, 0.f}, s2c8[4] = {0.f, 0.f, 0.f, 0.f};
  float s1 = 0.f, s2 = 0.f, s2c = 0.f;
  if (len > 0) {
    // window coordinates of the wave's first cell; afterwards +1 cell per step
    int cx = (int)(c_begin % p.OW);
    const long long t0 = c_begin / p.OW;
    int cy = (int)(t0 % p.OH);
    const long long b0 = t0 / p.OH;
    const unsigned pix_top0 = (unsigned)((b0 * p.H + 2 * cy) * p.W + 2 * cx);
    const char* dz_b = reinterpret_cast<const char*>(p.dz);
    const char* y_b = reinterpret_cast<const char*>(p.y);
    const char* x_b = reinterpret_cast<const char*>(p.x);
    // 32-bit BYTE offsets from the (uniform) base pointers; the launcher guarantees they fit
    unsigned off_dz = ((unsigned)c_begin * p.dz_ld + co) * ES;
    unsigned off_y = ((pix_top0 + h) * p.y_ld + co) * ES;          // top pixel of this lane's column
    unsigned off_x = (pix_top0 + h) * p.x_ld * 4u;
    const unsigned safe_dz = off_dz, safe_y = off_y;
    const unsigned row_y = (unsigned)p.W * p.y_ld * ES, row_x = (unsigned)p.W * p.x_ld * 4u;
    const unsigned step_dz = p.dz_ld * ES, step_y = 2u * p.y_ld * ES, step_x = 8u * p.x_ld;     // one cell = two pixel columns
    const int kb = ((dyo * p.W + dxo) * (int)p.x_ld + ci) * 4, k8 = (p.W + 1) * (int)p.x_ld * 4;
    int rel = 0;
    // Loads are unconditional from clamped (always mapped) offsets and the zero-selects happen at use, so all loads of
    // a group are in flight together, one group ahead of the MFMAs.
    struct Group { float yt[kCPG], yb[kCPG], gz[kCPG], bt[kCPG], bb[kCPG]; f32x4 x8t[SIDE ? kCPG : 1], x8b[SIDE ? kCPG : 1]; unsigned mask; };
    auto load = [&](Group& g) {
      g.mask = 0;
#pragma unroll
      for (int s = 0; s < kCPG; ++s) {
        const bool valid = rel < len;
        const int yy = 2 * cy, xx = 2 * cx + h;
        g.yt[s] = fsd_ew::ld1<T>(reinterpret_cast<const T*>(y_b + (valid ? off_y : safe_y)));
        g.yb[s] = fsd_ew::ld1<T>(reinterpret_cast<const T*>(y_b + (valid ? off_y + row_y : safe_y)));
        g.gz[s] = fsd_ew::ld1<T>(reinterpret_cast<const T*>(dz_b + (valid ? off_dz : safe_dz)));
        const bool x_in = (unsigned)(xx + dxo) < (unsigned)p.W;
        const bool okt = valid && col_ok && x_in && (unsigned)(yy + dyo) < (unsigned)p.H;
        const bool okb = valid && col_ok && x_in && (unsigned)(yy + 1 + dyo) < (unsigned)p.H;
        g.bt[s] = *reinterpret_cast<const float*>(x_b + (okt ? off_x + (unsigned)kb : 0u));
        g.bb[s] = *reinterpret_cast<const float*>(x_b + (okb ? off_x + row_x + (unsigned)kb : 0u));
        g.mask |= (valid ? 1u : 0u) << s;
        g.mask |= (okt ? 1u : 0u) << (4 + s);
        g.mask |= (okb ? 1u : 0u) << (8 + s);
        if constexpr (SIDE) {
          const bool ok8t = valid && xx + 1 < p.W;                        // tap (+1, +1) of the top pixel: row 2cy+1 < H
          const bool ok8b = valid && xx + 1 < p.W && yy + 2 < p.H;
          g.x8t[s] = *reinterpret_cast<const f32x4*>(x_b + (ok8t ? off_x + (unsigned)k8 : 0u));
          g.x8b[s] = *reinterpret_cast<const f32x4*>(x_b + (ok8b ? off_x + row_x + (unsigned)k8 : 0u));
          g.mask |= (ok8t ? 1u : 0u) << (12 + s);
          g.mask |= (ok8b ? 1u : 0u) << (16 + s);
        }
        ++rel; off_dz += step_dz; off_y += step_y; off_x += step_x;
        if (++cx == p.OW) {             // next window row: skip the bottom pixel row just covered
          cx = 0;
          off_y += row_y; off_x += row_x;
          if (++cy == p.OH) cy = 0;     // next image follows contiguously (H = 2 OH)
        }
      }
    };
    auto compute = [&](const Group& g) {
#pragma unroll
      for (int s = 0; s < kCPG; ++s) {
        const bool valid = (g.mask >> s) & 1u;
        const float yt = g.yt[s], yb = g.yb[s];
        const float gz = valid ? g.gz[s] : 0.f;
        const float tt = __builtin_fmaf(yt, sc, sh), tb = __builtin_fmaf(yb, sc, sh);
        const float at = tt > 0.f ? tt : tt * p.slope, ab = tb > 0.f ? tb : tb * p.slope;
        const float pt = __shfl_xor(at, 32, 64), pb = __shfl_xor(ab, 32, 64);
        // the window in scan order (first maximum wins, like torch and fsd_bn_act_pool_bwd)
        const float q0 = h ? pt : at, q1 = h ? at : pt, q2 = h ? pb : ab, q3 = h ? ab : pb;
        int bq = 0;
        float bv = q0;
        if (q1 > bv) { bv = q1; bq = 1; }
        if (q2 > bv) { bv = q2; bq = 2; }
        if (q3 > bv) { bq = 3; }
        const float gt = bq == h ? gz : 0.f, gb = bq == 2 + h ? gz : 0.f;
        const float dtop = tt > 0.f ? gt : gt * p.slope, dbot = tb > 0.f ? gb : gb * p.slope;
        const float xt = valid ? (yt - mu) * is : 0.f, xb = valid ? (yb - mu) * is : 0.f;
        s1 += dtop; s1 += dbot;
        s2 += dtop * xt; s2 += dbot * xb;
        // the weight gradient sees dt as the unfused path would have STORED it (bf16 mode: rounded)
        const float dq_t = fsd_ew::stored<T>(dtop), dq_b = fsd_ew::stored<T>(dbot);
        const float bt = ((g.mask >> (4 + s)) & 1u) ? g.bt[s] : 0.f;
        const float bb = ((g.mask >> (8 + s)) & 1u) ? g.bb[s] : 0.f;
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(dq_t, bt, acc1, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(xt, bt, acc3, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(dq_b, bb, acc1, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(xb, bb, acc3, 0, 0, 0);
        s2c += bt; s2c += bb;
        if constexpr (SIDE) {
          const bool o8t = (g.mask >> (12 + s)) & 1u, o8b = (g.mask >> (16 + s)) & 1u;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float vt = o8t ? g.x8t[s][j] : 0.f, vb = o8b ? g.x8b[s][j] : 0.f;
            a81[j] += dq_t * vt; a81[j] += dq_b * vb;
            a83[j] += xt * vt; a83[j] += xb * vb;
            s2c8[j] += vt; s2c8[j] += vb;
          }
        }
      }
    };
    Group g0, g1;
    load(g0);
    for (int done = 0; done < len; done += 2 * kCPG) {
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
  // ---- fold the four waves through LDS and write this workgroup's partials (fixed order: deterministic) ----
  float* so = s_out[wave];
  float* out = p.ws + ((size_t)blockIdx.x * 2 * p.Cout + (size_t)blockIdx.y * 32) * 36;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const f32x16& acc = which == 0 ? acc1 : acc3;
    float* a8 = which == 0 ? a81 : a83;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;        // output channel within the 32-tile
      if (col_ok) so[row * 36 + tap * 4 + ci] = acc[r];
    }
    if constexpr (SIDE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) a8[j] += __shfl_xor(a8[j], 32, 64);
      if (h == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) so[c * 36 + 32 + j] = a8[j];
      }
    } else {
      if (c < 9) {                                             // unused ci = 3 columns: keep them defined
#pragma unroll
        for (int r = 0; r < 16; ++r) so[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + c * 4 + 3] = 0.f;
      }
    }
    if (which == 1) {
      // S2 per column and the BatchNorm sums per channel ride along in the tail of the wave's LDS slice
      const float v2 = s2c + __shfl_xor(s2c, 32, 64);
      const float b1 = s1 + __shfl_xor(s1, 32, 64), b2 = s2 + __shfl_xor(s2, 32, 64);
      if (h == 0) {
        if (col_ok) so[32 * 36 + tap * 4 + ci] = v2;
        if (!SIDE && c < 9) so[32 * 36 + c * 4 + 3] = 0.f;
        so[32 * 36 + 36 + 2 * c] = b1;
        so[32 * 36 + 36 + 2 * c + 1] = b2;
      }
      if constexpr (SIDE) {
        // every lane of a half saw the same x8 values (they depend on the pixel, not on c): lane 0 / lane 32 carry them
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v8 = s2c8[j] + __shfl_xor(s2c8[j], 32, 64);
          if (lane == 0) so[32 * 36 + 32 + j] = v8;
        }
      }
    }
    __syncthreads();
    float* o = out + (size_t)which * p.Cout * 36;
    for (int e = threadIdx.x; e < 32 * 36; e += 256) o[e] = s_out[0][e] + s_out[1][e] + s_out[2][e] + s_out[3][e];
    if (which == 1) {
      if (blockIdx.y == 0 && threadIdx.x < 36) {
        const int e = 32 * 36 + threadIdx.x;
        p.ws2[(size_t)blockIdx.x * 36 + threadIdx.x] = s_out[0][e] + s_out[1][e] + s_out[2][e] + s_out[3][e];
      }
      if (threadIdx.x >= 64 && threadIdx.x < 128) {
        const int e = 32 * 36 + 36 + (threadIdx.x - 64);             // (channel, which sum) pairs
        const int cc = (threadIdx.x - 64) >> 1, w = (threadIdx.x - 64) & 1;
        p.partial[((size_t)blockIdx.x * p.Cout + blockIdx.y * 32 + cc) * 2 + w] =
            s_out[0][e] + s_out[1][e] + s_out[2][e] + s_out[3][e];
      }
    }
    __syncthreads();
  }
}

// dW[co][ci][ky][kx] = c1 (sum_b S1 - c2 sum_b S2 - c3 sum_b S3): one workgroup per (output channel, 9 of the 36 columns),
// 28 block lanes x 9 columns, fp64 sums in a fixed order.  (One workgroup per channel with 4 block lanes was a chain of
// 512 dependent-latency iterations on 32 workgroups: 141 us for 19 MB.)
constexpr int kFoldLanes = 28, kFoldCols = 9;
__global__ __launch_bounds__(256) void first_bwd_fold_kernel(const float* __restrict__ ws, const float* __restrict__ ws2,
                                                            const float* __restrict__ coef, float* __restrict__ dw,
                                                            int blocks, int cout, int cin) {
  __shared__ double s_acc[kFoldLanes][3][kFoldCols];
  const int co = blockIdx.x;
  const int cl = threadIdx.x % kFoldCols, ry = threadIdx.x / kFoldCols;
  const int col = blockIdx.y * kFoldCols + cl;
  if (ry < kFoldLanes) {
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int b = ry; b < blocks; b += kFoldLanes) {
      a1 += (double)ws[(((size_t)b * 2 + 0) * cout + co) * 36 + col];
      a3 += (double)ws[(((size_t)b * 2 + 1) * cout + co) * 36 + col];
      a2 += (double)ws2[(size_t)b * 36 + col];
    }
    s_acc[ry][0][cl] = a1; s_acc[ry][1][cl] = a2; s_acc[ry][2][cl] = a3;
  }
  __syncthreads();
  if (ry == 0) {
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int l = 0; l < kFoldLanes; ++l) { a1 += s_acc[l][0][cl]; a2 += s_acc[l][1][cl]; a3 += s_acc[l][2][cl]; }
    const double c1 = coef[co], c2 = coef[cout + co], c3 = coef[2 * cout + co];
    const int tap = col >> 2, ci = col & 3;
    if (ci < cin) dw[((size_t)co * cin + ci) * 9 + tap] = (float)(c1 * (a1 - c2 * a2 - c3 * a3));
  }
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

inline int fb_blocks(long long cells) {
  long long b = (cells + 4 * 64 - 1) / (4 * 64);              // at least 64 cells (256 pixels) per wave
  return (int)(b < 1 ? 1 : b > 2048 ? 2048 : b);
}

template <typename T>
int accum_impl(const T* dz, long long dz_ld, const T* y, long long y_ld, const float* scale, const float* shift,
               const float* mean, const float* invstd, float slope, const float* x, long long x_ld, void* workspace,
               size_t workspace_bytes, float* partial, int batch, int height, int width, int cin, int cout,
               hipStream_t stream) {
  (void)hipGetLastError();
  if (!dz || !y || !scale || !shift || !mean || !invstd || !x || !workspace || !partial || batch < 1 || height < 2 || width < 2)
    return FSD_ERR_ARG;
  if (cin < 1 || cin > 4 || cout % 32 || (height & 1) || (width & 1) || x_ld < 4 || (x_ld & 3) || dz_ld < cout || y_ld < cout)
    return FSD_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) & 15)) return FSD_ERR_ARG;
  const long long pixels = (long long)batch * height * width, cells = pixels / 4;
  const long long ld_max = dz_ld > y_ld ? (dz_ld > x_ld ? dz_ld : x_ld) : (y_ld > x_ld ? y_ld : x_ld);
  if ((pixels + 2LL * width + 18) * ld_max * 4 >= 0xffffffffLL) return FSD_ERR_UNSUPPORTED;      // 32-bit byte offsets
  const int blocks = fb_blocks(cells);
  if (workspace_bytes < fsd_first_layer_bwd_workspace_bytes(batch, height, width, cout)) return FSD_ERR_WORKSPACE;
  FirstBwdArgs a;
  a.dz = dz; a.y = y; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.x = x;
  a.ws = reinterpret_cast<float*>(workspace);
  a.ws2 = a.ws + (size_t)blocks * 2 * cout * 36;
  a.partial = partial;
  a.dz_ld = (unsigned)dz_ld; a.y_ld = (unsigned)y_ld; a.x_ld = (unsigned)x_ld;
  a.H = height; a.W = width; a.OH = height / 2; a.OW = width / 2; a.Cout = cout; a.cells = cells;
  // cells per load group: 1 (92 VGPRs, 5 waves per SIMD) measured fastest on the L0 shape -- 0.91 ms fp32 / 0.94 ms bf16
  // against 0.92 / 1.03 (2 cells, 108-168 VGPRs) and 0.96 / 1.07 (4 cells); FSD_FB_CPG = 1|2|4 is a tuning aid
  static const char* env = FSD_TUNE("FSD_FB_CPG");
  const int cpg = env ? atoi(env) : 1;
  a.cpw = round_up((int)((cells + (long long)blocks * 4 - 1) / ((long long)blocks * 4)), 8);
  a.slope = slope;
  fsd_prof::Scope prof(fsd_prof::kFirst, (double)pixels * (sizeof(T) * cout * 1.25 + 16.0), stream);
  const dim3 grid(blocks, cout / 32);
  if (cin == 4) {
    if (cpg == 1) FSD_LAUNCH((first_bwd_kernel<true, T, 1>), grid, dim3(256), 0, stream, a);
    else if (cpg == 4) FSD_LAUNCH((first_bwd_kernel<true, T, 4>), grid, dim3(256), 0, stream, a);
    else FSD_LAUNCH((first_bwd_kernel<true, T, 2>), grid, dim3(256), 0, stream, a);
  } else {
    if (cpg == 1) FSD_LAUNCH((first_bwd_kernel<false, T, 1>), grid, dim3(256), 0, stream, a);
    else if (cpg == 4) FSD_LAUNCH((first_bwd_kernel<false, T, 4>), grid, dim3(256), 0, stream, a);
    else FSD_LAUNCH((first_bwd_kernel<false, T, 2>), grid, dim3(256), 0, stream, a);
  }
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int fsd_first_layer_bwd_rows(int batch, int height, int width) {
  return fb_blocks((long long)batch * height * width / 4);
}

extern "C" size_t fsd_first_layer_bwd_workspace_bytes(int batch, int height, int width, int cout) {
  const size_t blocks = (size_t)fb_blocks((long long)batch * height * width / 4);
  return blocks * (2 * (size_t)cout * 36 + 36) * sizeof(float);
}

extern "C" int fsd_first_layer_bwd_accum(const float* dz, long long dz_ld, const float* y, long long y_ld,
                                         const float* scale, const float* shift, const float* mean, const float* invstd,
                                         float slope, const float* x, long long x_ld, void* workspace,
                                         size_t workspace_bytes, float* partial, int batch, int height, int width,
                                         int cin, int cout, hipStream_t stream) {
  return accum_impl<float>(dz, dz_ld, y, y_ld, scale, shift, mean, invstd, slope, x, x_ld, workspace, workspace_bytes,
                           partial, batch, height, width, cin, cout, stream);
}

extern "C" int fsd_first_layer_bwd_accum_h(const void* dz, long long dz_ld, const void* y, long long y_ld,
                                           const float* scale, const float* shift, const float* mean,
                                           const float* invstd, float slope, const float* x, long long x_ld,
                                           void* workspace, size_t workspace_bytes, float* partial, int batch, int height,
                                           int width, int cin, int cout, hipStream_t stream) {
  return accum_impl<bf16_t>(static_cast<const bf16_t*>(dz), dz_ld, static_cast<const bf16_t*>(y), y_ld, scale, shift, mean,
                            invstd, slope, x, x_ld, workspace, workspace_bytes, partial, batch, height, width, cin, cout,
                            stream);
}

extern "C" int fsd_first_layer_bwd_fold(const void* workspace, size_t workspace_bytes, const float* coef, float* dw_oihw,
                                        int batch, int height, int width, int cin, int cout, hipStream_t stream) {
  (void)hipGetLastError();
  if (!workspace || !coef || !dw_oihw || cin < 1 || cin > 4 || cout < 1) return FSD_ERR_ARG;
  if (workspace_bytes < fsd_first_layer_bwd_workspace_bytes(batch, height, width, cout)) return FSD_ERR_WORKSPACE;
  const int blocks = fb_blocks((long long)batch * height * width / 4);
  const float* ws = reinterpret_cast<const float*>(workspace);
  FSD_LAUNCH(first_bwd_fold_kernel, dim3(cout, 36 / kFoldCols), dim3(256), 0, stream, ws, ws + (size_t)blocks * 2 * cout * 36, coef,
                     dw_oihw, blocks, cout, cin);
  return (int)hipGetLastError();
}
