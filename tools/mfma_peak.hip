// Same-hardware ceiling for v_mfma_f32_32x32x2_f32: pure MFMA loop, no memory traffic.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, const float* in, int iters, long long* clk) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float av = in[threadIdx.x], bv = in[256 + threadIdx.x];
  long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
  }
  long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int NACC>
void run(const char* label, int blocks, int iters, float* out, float* in, long long* clk) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<NACC><<<blocks, 256>>>(out, in, 10, clk);
  hipEventRecord(e0);
  mfma_loop<NACC><<<blocks, 256>>>(out, in, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  double flops = (double)blocks * 4 /*waves*/ * iters * 4.0 * NACC * (32.0 * 32 * 2 * 2);
  printf("%-34s blocks=%5d  %8.3f ms  %7.1f TFLOP/s   shader clk %.0f MHz (clock64/wall_clock64 @100MHz)\n", label, blocks, ms,
         flops / ms / 1e9, (double)h[0] / ((double)h[1] / 100.0));
}

int main() {
  float *out, *in; long long* clk;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, 512 * 4); hipMalloc(&clk, 16);
  float h[512];
  for (int z = 0; z < 2; ++z) {
    for (int i = 0; i < 512; ++i) h[i] = z ? 0.f : (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    printf("---- operands: %s\n", z ? "zeros" : "uniform random [-1,1)");
    run<4>("4 acc, 1 block/CU (1 wave/SIMD)", 256, 20000, out, in, clk);
    run<4>("4 acc, 2 blocks/CU (2 waves/SIMD)", 512, 20000, out, in, clk);
    run<1>("1 acc, 2 blocks/CU", 512, 40000, out, in, clk);
    run<4>("4 acc, 8 blocks/CU", 2048, 5000, out, in, clk);
  }
  return 0;
}
