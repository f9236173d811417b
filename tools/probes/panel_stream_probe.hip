// How fast does HBM serve the weight panel of a short batched GEMM?  576 workgroups (36 positions x 16 column tiles), each reading
// its own 64-row x 1024-float panel ONCE, 32 floats of every row per k-chunk, one chunk in flight ahead of a barrier -- the access
// pattern of conv_gemm_kernel<64, 64> on a 1024 -> 1024 Winograd layer at two images.  mode 0: rows 4 KB apart ([n][K] layout);
// mode 1: the 64 x 32 block of a chunk contiguous (8 KB, chunk-major layout).  depth: chunks in flight.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/panel_stream_probe.hip -o /tmp/panel_probe && /tmp/panel_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void panel_kernel(const float* __restrict__ buf, float* __restrict__ out, int nk) {
  __shared__ float s[256];
  const float* base = buf + (size_t)blockIdx.x * 64 * nk * 32;
  const int t = threadIdx.x, row = t >> 3, piece = t & 7;
  auto addr = [&](int kc, int j) {
    const int r = row + 32 * j;
    return MODE == 0 ? base + (size_t)r * nk * 32 + kc * 32 + piece * 4 : base + (size_t)kc * 2048 + r * 32 + piece * 4;
  };
  f32x4 q[DEPTH][2];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int j = 0; j < 2; ++j) q[d][j] = *reinterpret_cast<const f32x4*>(addr(d, j));
  float acc = 0.f;
  for (int kc = 0; kc < nk; kc += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const f32x4 a = q[d][0], b = q[d][1];
      if (kc + d + DEPTH < nk) {
        q[d][0] = *reinterpret_cast<const f32x4*>(addr(kc + d + DEPTH, 0));
        q[d][1] = *reinterpret_cast<const f32x4*>(addr(kc + d + DEPTH, 1));
      }
      acc += a[0] + a[1] + a[2] + a[3] + b[0] + b[1] + b[2] + b[3];
      s[t] = acc;
      __syncthreads();
      acc += s[t ^ 1] * 1e-30f;
    }
  }
  if (acc == 123.456f) out[blockIdx.x] = acc;
}

template <int MODE, int DEPTH>
double run(const std::vector<float*>& bufs, float* out, int wgs, int nk) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((panel_kernel<MODE, DEPTH>), dim3(wgs), dim3(256), 0, 0, bufs[i % bufs.size()], out, nk);
  hipEventRecord(e0);
  const int reps = 40;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((panel_kernel<MODE, DEPTH>), dim3(wgs), dim3(256), 0, 0, bufs[i % bufs.size()], out, nk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  const int nk = 32;
  for (int wgs : {576, 1152, 2304}) {
    const size_t bytes = (size_t)wgs * 64 * nk * 32 * 4;
    std::vector<float*> bufs(6);
    for (auto& b : bufs) { hipMalloc(&b, bytes); hipMemset(b, 0, bytes); }
    float* out; hipMalloc(&out, 1 << 20);
    const double a1 = run<0, 1>(bufs, out, wgs, nk), b1 = run<1, 1>(bufs, out, wgs, nk);
    const double a2 = run<0, 2>(bufs, out, wgs, nk), b2 = run<1, 2>(bufs, out, wgs, nk);
    const double a4 = run<0, 4>(bufs, out, wgs, nk), b4 = run<1, 4>(bufs, out, wgs, nk);
    printf("wgs %4d  %6.1f MB | depth 1: rows-4KB-apart %6.1f us %5.2f TB/s, chunk-major %6.1f us %5.2f TB/s | depth 2: %6.1f us %5.2f, %6.1f us %5.2f | depth 4: %6.1f us %5.2f, %6.1f us %5.2f\n",
           wgs, bytes / 1e6, a1 * 1e3, bytes / a1 / 1e9, b1 * 1e3, bytes / b1 / 1e9, a2 * 1e3, bytes / a2 / 1e9, b2 * 1e3, bytes / b2 / 1e9,
           a4 * 1e3, bytes / a4 / 1e9, b4 * 1e3, bytes / b4 / 1e9);
    for (auto& b : bufs) hipFree(b);
    hipFree(out);
  }
  return 0;
}
