// Packed bf16 conv weights for the bf16 storage mode (conv_bf16v2.hip): [rows padded to 128][tap][channels], K-major, the
// row stride padded to a multiple of 64 elements; mode 1 = the data-gradient operand (filter rotated 180 degrees, channel
// roles swapped).  Rounding: round-to-nearest-even (v_cvt_pk_bf16_f32).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "fsdet.h"
#include "conv_common.hpp"
#include "profile.hpp"

namespace {

using namespace fsd_conv;

constexpr int kBKh = 64;     // k-chunk, in bf16 elements

__global__ void pack_weight_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int cout, int cin,
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
    else
      v = w[(((long long)r * cin + row) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];
  }
  const __bf16 b = (__bf16)v;
  out[idx] = __builtin_bit_cast(unsigned short, b);
}

// Both packings of one weight tensor in a single pass: a workgroup converts a [64 cout][64 cin][taps] block of the OIHW
// tensor (contiguous runs of 64*taps floats, float4 loads) into a bf16 LDS tile and writes it twice -- forward operand
// out0[co][tap][ci] and data-gradient operand out1[ci][taps-1-tap][co] -- as full 128-byte lines, 16 bytes per lane.
// Padding elements of out0 / out1 are never written: the caller zero-fills the buffers once.
// UPDATE: the block first takes its SGD(momentum, weight decay) step -- w, grad, momentum are read as the block is loaded, the
// new weight and momentum go back to their places, and the packed copies are made from the NEW weight (fsd_sgd_step_multi: the
// optimizer step and the per-step re-packing of the bf16 operands are one pass over the parameters).
struct SgdHyper { float lr, momentum, weight_decay; int first; };

__device__ __forceinline__ float sgd_update(float wi, float gi, float* mom, const SgdHyper& h) {
  const float d = gi + h.weight_decay * wi;                 // torch.optim.SGD: d = g + wd*w; buf = first ? d : mu*buf + d
  const float b = h.first ? d : h.momentum * *mom + d;
  *mom = b;
  return wi - h.lr * b;
}

template <int TAPS, bool UPDATE>
__device__ __forceinline__ void pack_pair_block(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ mom,
                                                unsigned short* __restrict__ out0, unsigned short* __restrict__ out1, int cout,
                                                int cin, int red4_0, int kpad0, int red4_1, int kpad1, int ci0, int co0,
                                                const SgdHyper& hy, unsigned short* tile) {
  constexpr int T = 64, RUN = T * TAPS;
  constexpr int ROW = RUN + 2;         // bf16 elements per cout row; (8 * ROW / 2) % 64 == 8: the mode-1 gathers spread over banks
  const int n_ci = min(T, cin - ci0), n_co = min(T, cout - co0);
  const bool full = n_ci == T && n_co == T && (cin & 3) == 0;
  if (full) {
#pragma unroll 9
    for (int e = threadIdx.x; e < T * RUN / 4; e += 1024) {
      const int co_l = e / (RUN / 4), r = (e - co_l * (RUN / 4)) * 4;       // r = ci_l * TAPS + tap
      const long long at = ((long long)(co0 + co_l) * cin + ci0) * TAPS + r;
      float4 v = *reinterpret_cast<const float4*>(w + at);
      if (UPDATE) {
        const float4 gv = *reinterpret_cast<const float4*>(g + at);
        float4 mv = hy.first ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(mom + at);
        v.x = sgd_update(v.x, gv.x, &mv.x, hy);
        v.y = sgd_update(v.y, gv.y, &mv.y, hy);
        v.z = sgd_update(v.z, gv.z, &mv.z, hy);
        v.w = sgd_update(v.w, gv.w, &mv.w, hy);
        *reinterpret_cast<float4*>(mom + at) = mv;
        *reinterpret_cast<float4*>(w + at) = v;
      }
      unsigned short* d = tile + co_l * ROW + r;
      d[0] = __builtin_bit_cast(unsigned short, (__bf16)v.x);
      d[1] = __builtin_bit_cast(unsigned short, (__bf16)v.y);
      d[2] = __builtin_bit_cast(unsigned short, (__bf16)v.z);
      d[3] = __builtin_bit_cast(unsigned short, (__bf16)v.w);
    }
  } else {
    for (int e = threadIdx.x; e < T * RUN; e += 1024) {
      const int co_l = e / RUN, r = e - co_l * RUN;
      float v = 0.f;
      if (co_l < n_co && r < n_ci * TAPS) {
        const long long at = ((long long)(co0 + co_l) * cin + ci0) * TAPS + r;
        v = w[at];
        if (UPDATE) {
          float mv = hy.first ? 0.f : mom[at];
          v = sgd_update(v, g[at], &mv, hy);
          mom[at] = mv;
          w[at] = v;
        }
      }
      tile[co_l * ROW + r] = __builtin_bit_cast(unsigned short, (__bf16)v);
    }
  }
  __syncthreads();
  // mode 0: (co, tap) rows of 64 ci = 8 lanes x 8 ci
#pragma unroll 5
  for (int e = threadIdx.x; e < T * TAPS * 8; e += 1024) {
    const int c8 = (e & 7) * 8, t2 = e >> 3;
    const int tap = t2 % TAPS, co_l = t2 / TAPS;
    if (co_l >= n_co || c8 >= n_ci) continue;
    const unsigned short* sp = tile + co_l * ROW + c8 * TAPS + tap;
    unsigned short* dp = out0 + (long long)(co0 + co_l) * kpad0 + tap * red4_0 + ci0 + c8;
    if (c8 + 8 <= n_ci && ((red4_0 | kpad0) & 7) == 0) {
      uint4 u;
      u.x = sp[0] | ((unsigned)sp[TAPS] << 16);
      u.y = sp[2 * TAPS] | ((unsigned)sp[3 * TAPS] << 16);
      u.z = sp[4 * TAPS] | ((unsigned)sp[5 * TAPS] << 16);
      u.w = sp[6 * TAPS] | ((unsigned)sp[7 * TAPS] << 16);
      *reinterpret_cast<uint4*>(dp) = u;
    } else {
      for (int q = 0; q < 8 && c8 + q < n_ci; ++q) dp[q] = sp[q * TAPS];
    }
  }
  // mode 1: (ci, tap) rows of 64 co = 8 lanes x 8 co
#pragma unroll 5
  for (int e = threadIdx.x; e < T * TAPS * 8; e += 1024) {
    const int r = e % RUN, o8 = (e / RUN) * 8;                  // consecutive lanes: consecutive (ci, tap), same co group
    const int ci_l = r / TAPS, tap = r - ci_l * TAPS;
    if (ci_l >= n_ci || o8 >= n_co) continue;
    const unsigned short* sp = tile + o8 * ROW + r;
    unsigned short* dp = out1 + (long long)(ci0 + ci_l) * kpad1 + (TAPS - 1 - tap) * red4_1 + co0 + o8;
    if (o8 + 8 <= n_co && ((red4_1 | kpad1) & 7) == 0) {
      uint4 u;
      u.x = sp[0] | ((unsigned)sp[ROW] << 16);
      u.y = sp[2 * ROW] | ((unsigned)sp[3 * ROW] << 16);
      u.z = sp[4 * ROW] | ((unsigned)sp[5 * ROW] << 16);
      u.w = sp[6 * ROW] | ((unsigned)sp[7 * ROW] << 16);
      *reinterpret_cast<uint4*>(dp) = u;
    } else {
      for (int q = 0; q < 8 && o8 + q < n_co; ++q) dp[q] = sp[q * ROW];
    }
  }
}

template <int TAPS>
__global__ __launch_bounds__(1024) void pack_weight_bf16_pair_kernel(const float* __restrict__ w, unsigned short* __restrict__ out0,
                                                                    unsigned short* __restrict__ out1, int cout, int cin,
                                                                    int red4_0, int kpad0, int red4_1, int kpad1) {
  extern __shared__ unsigned short tile[];
  const SgdHyper none = {0.f, 0.f, 0.f, 0};
  pack_pair_block<TAPS, false>(const_cast<float*>(w), nullptr, nullptr, out0, out1, cout, cin, red4_0, kpad0, red4_1, kpad1,
                               blockIdx.x * 64, blockIdx.y * 64, none, tile);
}

// One optimizer launch over MANY tensors of a flat parameter buffer.  Table row t (8 x int64): element offset of the tensor in
// the flat buffers, element count, cout, cin, taps (0: a plain range -- BatchNorm parameters, biases, weights without packed
// copies), first workgroup block of the tensor (prefix sum), address of the bf16 forward operand, address of the bf16
// data-gradient operand.  A plain range takes kPlainChunk elements per block; a conv weight one [64 cout][64 cin][taps] block.
constexpr int kPlainChunk = 16384;

__global__ __launch_bounds__(1024) void sgd_pack_multi_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                             float* __restrict__ mom, const long long* __restrict__ table,
                                                             int n_entries, SgdHyper hy) {
  extern __shared__ unsigned short tile[];
  // the entry this block belongs to: last row whose first block is <= blockIdx.x (binary search, wave-uniform)
  int lo = 0, hi = n_entries - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid * 8 + 5] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long long* e = table + lo * 8;
  const long long off = e[0], count = e[1];
  const int cout = (int)e[2], cin = (int)e[3], taps = (int)e[4];
  const int b = (int)((long long)blockIdx.x - e[5]);
  if (taps == 0) {
    const long long i0 = (long long)b * kPlainChunk, i1 = i0 + kPlainChunk < count ? i0 + kPlainChunk : count;
    for (long long i = i0 + threadIdx.x; i < i1; i += 1024) {
      float mv = hy.first ? 0.f : mom[off + i];
      w[off + i] = sgd_update(w[off + i], g[off + i], &mv, hy);
      mom[off + i] = mv;
    }
    return;
  }
  unsigned short* out0 = reinterpret_cast<unsigned short*>(e[6]);
  unsigned short* out1 = reinterpret_cast<unsigned short*>(e[7]);
  const int ci_blocks = (cin + 63) / 64;
  const int co0 = (b / ci_blocks) * 64, ci0 = (b % ci_blocks) * 64;
  const int red4_0 = (cin + 3) / 4 * 4, red4_1 = (cout + 3) / 4 * 4;
  const int kpad0 = (taps * red4_0 + kBKh - 1) / kBKh * kBKh, kpad1 = (taps * red4_1 + kBKh - 1) / kBKh * kBKh;
  if (taps == 9)
    pack_pair_block<9, true>(w + off, g + off, mom + off, out0, out1, cout, cin, red4_0, kpad0, red4_1, kpad1, ci0, co0, hy, tile);
  else
    pack_pair_block<1, true>(w + off, g + off, mom + off, out0, out1, cout, cin, red4_0, kpad0, red4_1, kpad1, ci0, co0, hy, tile);
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

extern "C" int fsd_pack_conv_weight_bf16_pair(const float* w_oihw, void* w_fwd_bf16, void* w_dgrad_bf16, int cout, int cin,
                                              int ksize, hipStream_t stream) {
  (void)hipGetLastError();
  if (!w_oihw || !w_fwd_bf16 || !w_dgrad_bf16 || cout < 1 || cin < 1 || (ksize != 1 && ksize != 3)) return FSD_ERR_ARG;
  const int taps = ksize * ksize;
  const int red4_0 = round_up(cin, 4), kpad0 = round_up(taps * red4_0, kBKh);
  const int red4_1 = round_up(cout, 4), kpad1 = round_up(taps * red4_1, kBKh);
  const dim3 grid((cin + 63) / 64, (cout + 63) / 64);
  if (grid.y > 65535) return FSD_ERR_UNSUPPORTED;
  const size_t lds = (size_t)64 * (64 * taps + 2) * sizeof(unsigned short);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pack_weight_bf16_pair_kernel<9>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)64 * (64 * 9 + 2) * 2));
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  unsigned short* o0 = static_cast<unsigned short*>(w_fwd_bf16);
  unsigned short* o1 = static_cast<unsigned short*>(w_dgrad_bf16);
  if (taps == 9)
    FSD_LAUNCH(pack_weight_bf16_pair_kernel<9>, grid, dim3(1024), lds, stream, w_oihw, o0, o1, cout, cin, red4_0, kpad0, red4_1, kpad1);
  else
    FSD_LAUNCH(pack_weight_bf16_pair_kernel<1>, grid, dim3(1024), lds, stream, w_oihw, o0, o1, cout, cin, red4_0, kpad0, red4_1, kpad1);
  return (int)hipGetLastError();
}

extern "C" long long fsd_sgd_multi_blocks(long long count, int cout, int cin, int taps) {
  if (taps == 0) return (count + kPlainChunk - 1) / kPlainChunk;
  return (long long)((cout + 63) / 64) * ((cin + 63) / 64);
}

extern "C" int fsd_sgd_step_multi(float* w_flat, const float* grad_flat, float* momentum_flat, const long long* table_dev,
                                  int n_entries, long long total_blocks, long long elements, float lr, float momentum,
                                  float weight_decay, int first_step, hipStream_t stream) {
  (void)hipGetLastError();
  if (!w_flat || !grad_flat || !momentum_flat || !table_dev || n_entries < 1 || total_blocks < 1) return FSD_ERR_ARG;
  if (total_blocks > 0x7fffffffLL) return FSD_ERR_UNSUPPORTED;
  const size_t lds = (size_t)64 * (64 * 9 + 2) * sizeof(unsigned short);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sgd_pack_multi_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const SgdHyper hy = {lr, momentum, weight_decay, first_step};
  fsd_prof::Scope prof(fsd_prof::kSgd, 20.0 * (double)elements, stream);          // w, g, m read; w, m written (+ the bf16 copies)
  FSD_LAUNCH(sgd_pack_multi_kernel, dim3((unsigned)total_blocks), dim3(1024), lds, stream, w_flat, grad_flat, momentum_flat,
             table_dev, n_entries, hy);
  return (int)hipGetLastError();
}

extern "C" size_t fsd_packed_weight_elems_bf16(int rows, int red, int ksize) {
  return (size_t)round_up(rows, 128) * (size_t)round_up(ksize * ksize * round_up(red, 4), kBKh);
}

extern "C" int fsd_pack_conv_weight_bf16(const float* w_oihw, void* w_packed_bf16, int cout, int cin, int ksize,
                                         int mode, hipStream_t stream) {
  (void)hipGetLastError();
  if (!w_oihw || !w_packed_bf16 || cout < 1 || cin < 1 || (ksize != 1 && ksize != 3) || (mode != 0 && mode != 1))
    return FSD_ERR_ARG;
  const int rows = mode == 0 ? cout : cin, red = mode == 0 ? cin : cout;
  const int rows_pad = round_up(rows, 128), red4 = round_up(red, 4);
  const int kpad = round_up(ksize * ksize * red4, kBKh);
  const long long total = (long long)rows_pad * kpad;
  FSD_LAUNCH(pack_weight_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w_oihw,
                     static_cast<unsigned short*>(w_packed_bf16), cout, cin, ksize, mode, rows_pad, red4, kpad);
  return (int)hipGetLastError();
}
