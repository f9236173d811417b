// Per-kernel-class HIP-event bookkeeping behind fsd_profile_enable / fsd_profile_collect (include/fsdet.h).
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <vector>
#include "fsdet.h"
#include "profile.hpp"

namespace {

struct Rec { hipEvent_t e0, e1; int cls; double work; };
std::atomic<int> g_on{0};
std::atomic<long long> g_launches{0};
std::mutex g_mu;
std::vector<Rec*> g_recs;

}  // namespace

void fsd_prof::count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }

extern "C" long long fsd_launch_count(int reset) {
  return reset ? g_launches.exchange(0, std::memory_order_relaxed) : g_launches.load(std::memory_order_relaxed);
}

bool fsd_prof::enabled() { return g_on.load(std::memory_order_relaxed) != 0; }

void fsd_prof::begin(int cls, double work, hipStream_t stream, void** token) {
  Rec* r = new Rec{nullptr, nullptr, cls, work};
  if (hipEventCreate(&r->e0) != hipSuccess || hipEventCreate(&r->e1) != hipSuccess) {
    if (r->e0) (void)hipEventDestroy(r->e0);
    delete r;
    return;
  }
  (void)hipEventRecord(r->e0, stream);
  *token = r;
}

void fsd_prof::end(void* token, hipStream_t stream) {
  Rec* r = static_cast<Rec*>(token);
  (void)hipEventRecord(r->e1, stream);
  std::lock_guard<std::mutex> lk(g_mu);
  g_recs.push_back(r);
}

extern "C" void fsd_profile_enable(int on) { g_on.store(on ? 1 : 0, std::memory_order_relaxed); }

extern "C" int fsd_profile_num_classes(void) { return fsd_prof::kNumClasses; }

extern "C" int fsd_profile_collect(double* ms, double* work, long long* launches, int n_classes) {
  if (!ms || !work || !launches || n_classes < fsd_prof::kNumClasses) return FSD_ERR_ARG;
  for (int i = 0; i < n_classes; ++i) { ms[i] = 0.0; work[i] = 0.0; launches[i] = 0; }
  std::vector<Rec*> recs;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    recs.swap(g_recs);
  }
  int rc = 0;
  for (Rec* r : recs) {
    float t = 0.f;
    if (hipEventSynchronize(r->e1) == hipSuccess && hipEventElapsedTime(&t, r->e0, r->e1) == hipSuccess &&
        r->cls >= 0 && r->cls < fsd_prof::kNumClasses) {
      ms[r->cls] += t;
      work[r->cls] += r->work;
      launches[r->cls] += 1;
    } else {
      rc = FSD_ERR_ARG;
    }
    (void)hipEventDestroy(r->e0);
    (void)hipEventDestroy(r->e1);
    delete r;
  }
  return rc;
}

// ---- clock probe: the shader clock the chip sustains under matrix-core load, from a dependent MFMA chain ------------
// One wave per SIMD issues `iters` x 16 dependent v_mfma_f32_32x32x2_f32 (64 cycles each, MI355X_MICROARCH.md): the chain
// takes iters * 16 * 64 cycles whatever the memory system does, so cycles / elapsed time is the clock.  bench.py records
// it next to the roofline (the fp32 MFMA peak of 157.3 TFLOP/s is quoted at 2.4 GHz; under sustained fp32 MFMA load the
// chip runs at 2.0-2.3 GHz) and uses it to notice a throttled GPU before it times anything.
namespace {
typedef float f32x16p __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(64) void clock_probe_kernel(float* out, int iters) {
  f32x16p acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float a = 1.0f + 1e-6f * threadIdx.x, b = 0.5f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  float v = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) v += acc[r];
  if (v == 123.456f) out[0] = v;          // never true: keeps the chain alive
}
}  // namespace

extern "C" int fsd_clock_probe(float* scratch, int iters, double* mhz_out, hipStream_t stream) {
  (void)hipGetLastError();
  if (!scratch || !mhz_out || iters < 1) return FSD_ERR_ARG;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return FSD_ERR_ARG;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1024), dim3(64), 0, stream, scratch, 64);        // warm the clocks up
  (void)hipEventRecord(e0, stream);
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1024), dim3(64), 0, stream, scratch, iters);
  (void)hipEventRecord(e1, stream);
  float ms = 0.f;
  int rc = (int)hipGetLastError();
  if (rc == 0 && (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)) rc = FSD_ERR_ARG;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc != 0 || ms <= 0.f) return rc ? rc : FSD_ERR_ARG;
  *mhz_out = (double)iters * 16.0 * 64.0 / (ms * 1e-3) / 1e6;
  return 0;
}
