// Per-host-thread library state: launch stream and sticky error (see include/pwclo_ops.h, section 0).
#include <stdarg.h>
#include <string.h>

#include "common.hpp"

namespace pwclo {

static thread_local hipStream_t t_stream = nullptr;
static thread_local int t_err = 0;
static thread_local char t_msg[512] = "";

hipStream_t current_stream() { return t_stream; }

void set_error(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  char buf[448];
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (t_err == 0) {  // sticky: keep the first failure
    t_err = code;
    snprintf(t_msg, sizeof(t_msg), "%s", buf);
  }
  fprintf(stderr, "libpwclo_hip: error %d: %s\n", code, buf);
}

// ---- device-side failures -----------------------------------------------------------------------
// A kernel that detects a failure only while it runs (the cooperative sampler's bounded spins) cannot
// reach the host's sticky status directly.  The library owns ONE word of pinned, device-mapped host
// memory; such a kernel stores a PWCLO_E* code there with a system-scope atomic and ends.  The host reads
// the word without any HIP call in pwclo_last_error(): once the kernel has run, the next library call,
// or an explicit pwclo_last_error() after a synchronisation, reports it -- never silently continued
// (the reference's contract, cuda_utils.h:30-39).  Allocated on first use, outside stream capture.
static unsigned *g_dev_err_host = nullptr;   // host view of the word
static unsigned *g_dev_err_dev = nullptr;    // device view of the same word

unsigned *device_error_word() {
  if (g_dev_err_dev == nullptr) {
    void *h = nullptr, *d = nullptr;
    hipError_t e = hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess) {
      set_error((int)e, "cannot allocate the device error word: %s", hipGetErrorString(e));
      (void)hipGetLastError();
      return nullptr;
    }
    *reinterpret_cast<volatile unsigned *>(h) = 0u;
    g_dev_err_host = reinterpret_cast<unsigned *>(h);
    g_dev_err_dev = reinterpret_cast<unsigned *>(d);
  }
  return g_dev_err_dev;
}

// Moves a code posted by a kernel into the calling thread's sticky status.
static void poll_device_error() {
  if (g_dev_err_host == nullptr) return;
  if (__atomic_load_n(g_dev_err_host, __ATOMIC_ACQUIRE) == 0u) return;     // the common case: one load, no write
  // take and clear in ONE step: a code a kernel posts between a load and a separate store would be lost
  const unsigned code = __atomic_exchange_n(g_dev_err_host, 0u, __ATOMIC_ACQ_REL);
  if (code == 0u) return;
  if (code == (unsigned)PWCLO_ECOOP_TIMEOUT)
    set_error(PWCLO_ECOOP_TIMEOUT,
              "furthest_point_sampling(coop): a workgroup waited for its peers beyond the spin bound "
              "(the workgroups of a cloud were not co-resident); the output indices are incomplete");
  else
    set_error((int)code, "a kernel reported device-side failure %u", code);
}

bool check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error((int)e, "HIP kernel launch failed in %s: %s", what, hipGetErrorString(e));
    return false;
  }
  return true;
}

}  // namespace pwclo

extern "C" {

int pwclo_abi_version(void) { return 1; }
void pwclo_set_stream(void *hip_stream) { pwclo::t_stream = (hipStream_t)hip_stream; }
void *pwclo_get_stream(void) { return (void *)pwclo::t_stream; }
int pwclo_last_error(void) {
  pwclo::poll_device_error();
  return pwclo::t_err;
}
const char *pwclo_last_error_message(void) { return pwclo::t_msg; }
void pwclo_trace_enable(void *records, void *count, unsigned capacity) {
#if !defined(PWCLO_TRACE)
  if (records != nullptr)
    pwclo::set_error(PWCLO_EINVAL, "this library was built without the workgroup-trace hooks "
                                   "(python -m pwclonet_pylidarslam_amd.build --trace builds the traced variant)");
#endif
  pwclo::TraceBuf b{reinterpret_cast<pwclo::TraceRec *>(records), reinterpret_cast<unsigned *>(count),
                    records ? capacity : 0u};
  pwclo::trace_set_fused_layers(b);
  pwclo::trace_set_fused_hoisted(b);
  pwclo::trace_set_knn(b);
  pwclo::trace_set_sampling(b);
  pwclo::trace_set_warp(b);
}
void pwclo_clear_error(void) {
  pwclo::t_err = 0;
  pwclo::t_msg[0] = 0;
}

}  // extern "C"
