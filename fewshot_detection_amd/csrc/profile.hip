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
std::mutex g_mu;
std::vector<Rec*> g_recs;

}  // namespace

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
