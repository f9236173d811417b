// Probe of ds_read_b64_tr_b16 lane semantics on gfx950 (used to design the bf16 weight-gradient kernel's fragment reads).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int mode) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int off;
  if (mode == 0) off = 4 * l;                                   // lane-linear 8-byte chunks
  else {                                                        // [k][m] image with row stride 128 elements
    const int G = l >> 4, L = l & 15;
    off = ((G >> 1) * 8 + (L >> 2)) * 128 + 16 * (G & 1) + 4 * (L & 3);
  }
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + off));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, 512);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
