// Probe (not shipped): can fp32 GEMMs run on the bf16 matrix cores at fp32 accuracy?
//   a = a1 + a2 + a3 with a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2): 3 x 8 significant bits = the 24 of fp32.
//   x9: all nine cross terms; x6: without a2*b3, a3*b2, a3*b3 (<= 2^-24 relative each); x3: two-way split (16 bits).
// Prints the error of each scheme and of the native fp32 MFMA against a double-precision host sum, and the raw issue rates
// of v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_bf16.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/split_probe tools/probes/split_gemm_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float bf_round(float x) { return (float)(__bf16)x; }

// one wave, one 32x32 tile: a[32][K], b[32][K] (K contiguous), out[scheme][32][32]
__global__ void probe_kernel(const float* a, const float* b, float* out, int K) {
  const int lane = threadIdx.x, r = lane & 31, kh = lane >> 5;
  f32x16 nat = {0}, x3 = {0}, x6 = {0}, x9 = {0}, x6s = {0}, x6l = {0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    bf16x8 a1, a2, a3, b1, b2, b3;
    for (int j = 0; j < 8; ++j) {
      const float av = a[(size_t)r * K + k0 + kh * 8 + j], bv = b[(size_t)r * K + k0 + kh * 8 + j];
      const float ah = bf_round(av), am = bf_round(av - ah), al = bf_round(av - ah - am);
      const float bh = bf_round(bv), bm = bf_round(bv - bh), bl = bf_round(bv - bh - bm);
      a1[j] = (__bf16)ah; a2[j] = (__bf16)am; a3[j] = (__bf16)al;
      b1[j] = (__bf16)bh; b2[j] = (__bf16)bm; b3[j] = (__bf16)bl;
    }
    // native fp32: 8 steps of k=2; lane holds (row r, k = kh) per step
    for (int s = 0; s < 8; ++s) {
      const float av = a[(size_t)r * K + k0 + s * 2 + kh], bv = b[(size_t)r * K + k0 + s * 2 + kh];
      nat = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, nat, 0, 0, 0);
    }
    x3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, x3, 0, 0, 0);
    x3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, x3, 0, 0, 0);
    x3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, x3, 0, 0, 0);
    // x6, one accumulator, small terms first
    x6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, x6, 0, 0, 0);
    x6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, x6, 0, 0, 0);
    x6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, x6, 0, 0, 0);
    x6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, x6, 0, 0, 0);
    x6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, x6, 0, 0, 0);
    x6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, x6, 0, 0, 0);
    // x6 with the five small terms in their own accumulator (added once at the end)
    x6s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, x6s, 0, 0, 0);
    x6s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, x6s, 0, 0, 0);
    x6s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, x6s, 0, 0, 0);
    x6s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, x6s, 0, 0, 0);
    x6s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, x6s, 0, 0, 0);
    x6l = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, x6l, 0, 0, 0);
    x9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b3, x9, 0, 0, 0);
    x9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b3, x9, 0, 0, 0);
    x9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b2, x9, 0, 0, 0);
    x9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, x9, 0, 0, 0);
    x9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, x9, 0, 0, 0);
    x9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, x9, 0, 0, 0);
    x9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, x9, 0, 0, 0);
    x9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, x9, 0, 0, 0);
    x9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, x9, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) {
    const int row = (i / 4) * 8 + kh * 4 + (i % 4), col = r;     // D[row = a's row][col = b's row]
    out[0 * 1024 + row * 32 + col] = nat[i];
    out[1 * 1024 + row * 32 + col] = x3[i];
    out[2 * 1024 + row * 32 + col] = x6[i];
    out[3 * 1024 + row * 32 + col] = x6s[i] + x6l[i];
    out[4 * 1024 + row * 32 + col] = x9[i];
  }
}

template <int KIND> __global__ void rate_kernel(float* sink, int iters) {
  f32x16 acc[4] = {{0}, {0}, {0}, {0}};
  const float s = (float)threadIdx.x * 1e-9f;
  bf16x8 va, vb;
  for (int j = 0; j < 8; ++j) { va[j] = (__bf16)(s + j); vb[j] = (__bf16)(s - j); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (KIND == 0) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s + 1.f, acc[u], 0, 0, 0);
      else acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc[u], 0, 0, 0);
    }
  }
  float t = 0.f;
  for (int u = 0; u < 4; ++u) for (int i = 0; i < 16; ++i) t += acc[u][i];
  if (t == 123.456f) sink[0] = t;
}

static void fill(std::vector<float>& v, int kind, unsigned seed) {
  srand(seed);
  for (auto& x : v) {
    const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
    const double n = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    x = kind == 0 ? (float)n : (kind == 1 ? (float)fabs(n) : (float)(n > 0 ? n : 0.1 * n));
  }
}

int main() {
  const char* names[5] = {"native fp32 mfma", "bf16 x3 (2-way) ", "bf16 x6         ", "bf16 x6 (5+1 acc)", "bf16 x9         "};
  const char* kinds[3] = {"normal", "|normal|", "leaky(normal)"};
  for (int K : {576, 4608, 9216}) {
    for (int kind = 0; kind < 3; ++kind) {
      std::vector<float> a(32 * (size_t)K), b(32 * (size_t)K), o(5 * 1024);
      fill(a, kind, 1 + kind); fill(b, kind == 1 ? 1 : 0, 77 + kind);
      float *da, *db, *dout;
      hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dout, o.size() * 4);
      hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
      hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
      probe_kernel<<<1, 64>>>(da, db, dout, K);
      hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
      std::vector<double> ref(1024), mag(1024);
      for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double s = 0, m = 0;
        for (int k = 0; k < K; ++k) { const double p = (double)a[(size_t)i * K + k] * b[(size_t)j * K + k]; s += p; m += fabs(p); }
        ref[i * 32 + j] = s; mag[i * 32 + j] = m;
      }
      printf("K=%d a=%s\n", K, kinds[kind]);
      for (int v = 0; v < 5; ++v) {
        double e2 = 0, r2 = 0, emax = 0, bias = 0;
        for (int i = 0; i < 1024; ++i) {
          const double e = o[v * 1024 + i] - ref[i];
          e2 += e * e; r2 += ref[i] * ref[i]; bias += e / mag[i];
          if (fabs(e) / mag[i] > emax) emax = fabs(e) / mag[i];
        }
        printf("  %s  rel L2 %.3e   max |e|/sum|ab| %.3e   mean e/sum|ab| %+.3e\n", names[v], sqrt(e2 / r2), emax, bias / 1024);
      }
      hipFree(da); hipFree(db); hipFree(dout);
    }
  }
  float* sink; hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int kind = 0; kind < 2; ++kind) {
    const int iters = 20000, blocks = 256 * 8;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (kind == 0) rate_kernel<0><<<blocks, 256>>>(sink, iters); else rate_kernel<1><<<blocks, 256>>>(sink, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 4 * (kind == 0 ? 4096.0 : 32768.0);
    printf("%s: %.1f TFLOP/s\n", kind == 0 ? "v_mfma_f32_32x32x2_f32 " : "v_mfma_f32_32x32x16_bf16", flops / ms / 1e9);
  }
  return 0;
}
