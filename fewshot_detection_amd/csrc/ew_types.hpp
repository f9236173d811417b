// Storage types of the activation tensors in HBM and their 4-channel load / store helpers.  The HBM-bound kernels are
// templated on the storage type T: float (fp32 mode) or bf16_t (bf16 mode: activations, their gradients and the conv
// operands live in HBM as bfloat16; all arithmetic -- BatchNorm statistics included -- stays in float registers).
#ifndef FSD_EW_TYPES_HPP_
#define FSD_EW_TYPES_HPP_
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fsd_ew {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;        // raw bfloat16 bits

__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
// round-to-nearest-even, the hardware conversion (v_cvt_pk_bf16_f32); NaN stays NaN
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  const __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ld4<bf16_t>(const bf16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  f32x4 v;
  v[0] = __uint_as_float(u.x << 16);
  v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16);
  v[3] = __uint_as_float(u.y & 0xffff0000u);
  return v;
}

template <typename T> __device__ __forceinline__ void st4(T* p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, f32x4 v) {
  uint2 u;
  u.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
  u.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
  *reinterpret_cast<uint2*>(p) = u;
}

// 8 bf16 channels = one 16-byte access per lane (the 4-channel form moves 8 bytes per lane, 512 per wave instruction)
struct f32x8 { f32x4 lo, hi; };
__device__ __forceinline__ f32x8 ld8(const bf16_t* p) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  f32x8 v;
  v.lo[0] = __uint_as_float(u.x << 16); v.lo[1] = __uint_as_float(u.x & 0xffff0000u);
  v.lo[2] = __uint_as_float(u.y << 16); v.lo[3] = __uint_as_float(u.y & 0xffff0000u);
  v.hi[0] = __uint_as_float(u.z << 16); v.hi[1] = __uint_as_float(u.z & 0xffff0000u);
  v.hi[2] = __uint_as_float(u.w << 16); v.hi[3] = __uint_as_float(u.w & 0xffff0000u);
  return v;
}
__device__ __forceinline__ void st8(bf16_t* p, const f32x8& v) {
  uint4 u;
  u.x = (unsigned)f32_to_bf16(v.lo[0]) | ((unsigned)f32_to_bf16(v.lo[1]) << 16);
  u.y = (unsigned)f32_to_bf16(v.lo[2]) | ((unsigned)f32_to_bf16(v.lo[3]) << 16);
  u.z = (unsigned)f32_to_bf16(v.hi[0]) | ((unsigned)f32_to_bf16(v.hi[1]) << 16);
  u.w = (unsigned)f32_to_bf16(v.hi[2]) | ((unsigned)f32_to_bf16(v.hi[3]) << 16);
  *reinterpret_cast<uint4*>(p) = u;
}

template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// what a value looks like after a round trip through the storage type (the backward pass must see the activation the
// forward pass STORED, not the float it was computed as)
template <typename T> __device__ __forceinline__ float stored(float v);
template <> __device__ __forceinline__ float stored<float>(float v) { return v; }
template <> __device__ __forceinline__ float stored<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }

}  // namespace fsd_ew
#endif  // FSD_EW_TYPES_HPP_
