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

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

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
  hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w_oihw,
                     static_cast<unsigned short*>(w_packed_bf16), cout, cin, ksize, mode, rows_pad, red4, kpad);
  return (int)hipGetLastError();
}
