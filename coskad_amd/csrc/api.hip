// C-ABI plumbing: version, last-error text, launch checking.
#include "common.h"

namespace coskad {

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(COSKAD_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return COSKAD_OK;
}

}  // namespace coskad

extern "C" {
int coskad_abi_version(void) { return COSKAD_ABI_VERSION; }
const char* coskad_last_error(void) { return coskad::err_buf(); }
}
