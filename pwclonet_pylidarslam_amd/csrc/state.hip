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
int pwclo_last_error(void) { return pwclo::t_err; }
const char *pwclo_last_error_message(void) { return pwclo::t_msg; }
void pwclo_clear_error(void) {
  pwclo::t_err = 0;
  pwclo::t_msg[0] = 0;
}

}  // extern "C"
