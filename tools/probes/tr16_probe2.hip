// Probe 2: the weight-gradient kernel's transposed fragment read on a [32 rows][CH] bf16 LDS image with its XOR swizzle,
// filled (a) by plain stores, (b) by global_load_lds exactly like the kernel's stage().
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
template <int CH> __device__ int swz(int row) { return CH == 128 ? ((row & 3) << 2) : CH == 64 ? ((row & 2) << 1) : 0; }
__device__ void dma16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int CH>
__global__ void k(const u16* src, short* out, int use_dma) {
  extern __shared__ __attribute__((aligned(16))) u16 lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = CH / 8, RPI = 64 / LPR;
  if (use_dma) {
    for (int j = 0; j * 4 * RPI < 32; ++j) {
      const int row = wave * RPI + lane / LPR + 4 * RPI * j;
      if (4 * RPI * j + wave * RPI < 32) {
        const int piece = (lane % LPR) ^ swz<CH>(row);
        dma16(src + row * CH + piece * 8, lds + (4 * RPI * j + wave * RPI) * CH);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    for (int i = threadIdx.x; i < 32 * CH; i += 256) {
      const int row = i / CH, ch = i % CH;
      lds[row * CH + (((ch >> 3) ^ swz<CH>(row)) << 3) + (ch & 7)] = src[i];
    }
  }
  __syncthreads();
  if (wave != 0) return;
  const int G = lane >> 4, Lq = lane & 15;
  for (int s = 0; s < 2; ++s) {
    const int krow = s * 16 + (G >> 1) * 8;
    const int row = krow + (Lq >> 2);
    const int ch = 16 * (G & 1) + 4 * (Lq & 3);
    const int piece = (ch >> 3) ^ swz<CH>(row);
    const u16* a = lds + row * CH + piece * 8 + (ch & 7);
    s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(a));
    s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(a + 4 * CH));
    for (int e = 0; e < 4; ++e) { out[(s * 64 + lane) * 8 + e] = lo[e]; out[(s * 64 + lane) * 8 + 4 + e] = hi[e]; }
  }
}
template <int CH> int run(int use_dma) {
  u16 h[32 * 128]; short o[2 * 64 * 8];
  for (int r = 0; r < 32; ++r) for (int c = 0; c < CH; ++c) h[r * CH + c] = (u16)(r * 256 + c);   // value = row*256 + channel
  u16* d; short* dout;
  (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&dout, sizeof(o));
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k<CH>, dim3(1), dim3(256), 32 * CH * 2, 0, d, dout, use_dma);
  (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int s = 0; s < 2; ++s) for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
    const int krow = s * 16 + (l >> 5) * 8 + e, m = l & 31;
    const int want = krow * 256 + m, got = (unsigned short)o[(s * 64 + l) * 8 + e];
    if (want != got) { if (bad < 6) printf("  CH=%d dma=%d s=%d lane=%d e=%d want row %d ch %d got row %d ch %d\n", CH, use_dma, s, l, e, krow, m, got >> 8, got & 255); ++bad; }
  }
  printf("CH=%d dma=%d mismatches %d\n", CH, use_dma, bad);
  return bad;
}
int main() { run<128>(0); run<128>(1); run<64>(0); run<64>(1); run<32>(0); run<32>(1); return 0; }
