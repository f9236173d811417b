// Measurement aid shared by every translation unit: when enabled (fsd_profile_enable), a Scope around a kernel launch
// records HIP events on the launch stream and books the launch under a kernel class together with its work figure
// (MFMA FLOPs actually issued, or algorithmic HBM bytes).  fsd_profile_collect() synchronises, sums per class and
// clears.  Disabled (the default) it costs one relaxed load per launch.  Defined in profile.hip.
#ifndef FSD_PROFILE_HPP_
#define FSD_PROFILE_HPP_
#include <hip/hip_runtime.h>
#include <stdlib.h>

// Switches.  The default library reads a documented handful of FSD_* environment variables ("Switches") with
// getenv(); every other knob of the measurement history is a tuning aid that exists only in a build with -DFSD_EXPERIMENTS
// (`make EXTRA=-DFSD_EXPERIMENTS`): in the default build FSD_TUNE(name) is a null constant and the branch behind it folds away.
#ifdef FSD_EXPERIMENTS
#define FSD_TUNE(name) getenv(name)
#else
#define FSD_TUNE(name) (static_cast<const char*>(nullptr))
#endif


namespace fsd_prof {

enum Class {
  kGemmFwd = 0,     // conv_gemm_kernel: forward convolutions, data gradients, Winograd position GEMMs   [MFMA, FLOP]
  kGemmWgrad = 1,   // wgrad_kernel: weight-gradient reduction GEMMs (direct and Winograd)              [MFMA, FLOP]
  kWinoXform = 2,   // Winograd input / output / gradient transforms                                    [HBM, bytes]
  kActBwd = 3,      // BatchNorm / leaky / maxpool backward passes                                      [HBM, bytes]
  kActFwd = 4,      // BatchNorm + leaky + maxpool forward pass                                         [HBM, bytes]
  kRegion = 5,      // region_rows_kernel + region_class_kernel                                         [HBM, bytes]
  kSgd = 6,         // fused SGD step                                                                   [HBM, bytes]
  kFirst = 7,       // first-layer direct-operand kernels (forward, fused weight gradient)              [HBM, bytes]
  kGemmBf16 = 8,    // bf16-operand MFMA GEMM kernels (forward / data gradient / weight gradient)       [MFMA, FLOP]
  kNumClasses = 9
};

bool enabled();
void count_launch();     // one relaxed increment per kernel launch of the library (fsd_launch_count)
void begin(int cls, double work, hipStream_t stream, void** token);
void end(void* token, hipStream_t stream);

struct Scope {
  void* token = nullptr;
  hipStream_t stream;
  Scope(int cls, double work, hipStream_t s) : stream(s) {
    if (enabled()) begin(cls, work, s, &token);
  }
  ~Scope() {
    if (token) end(token, stream);
  }
  Scope(const Scope&) = delete;
  Scope& operator=(const Scope&) = delete;
};

}  // namespace fsd_prof

// every kernel launch of the library goes through this: the launch counter behind fsd_launch_count (include/fsdet.h)
#define FSD_LAUNCH(...) do { fsd_prof::count_launch(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#endif  // FSD_PROFILE_HPP_
