// C-ABI plumbing: version, last-error text, launch checking.
#include "common.h"
#include <utility>
#include <vector>

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

// ---- per-kernel timing probe (bench.py's roofline leg) -------------------------------------
// coskad_probe_begin(kernel, Ci, Co) arms the probe: every later launch of that tile kernel with those
// channel counts is bracketed by HIP events ON ITS LAUNCH STREAM; coskad_probe_end() synchronises the
// events and returns the average duration.  Not thread-safe; meant for one benchmarking thread.
namespace coskad {
namespace {
struct Probe {
  int kernel = 0, ci = 0, co = 0;
  int stride = 1, seen = 0;    // every stride-th matching launch is timed (two event records cost ~2.7 us of stream time each pair)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
} g_probe;
}  // namespace

ProbeScope::ProbeScope(int kernel, int ci, int co, hipStream_t st) : st_(st) {
  if (g_probe.kernel == kernel && g_probe.ci == ci && g_probe.co == co && g_probe.evs.size() < 4096 &&
      (g_probe.seen++ % g_probe.stride) == 0) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) {
      (void)hipEventRecord(a, st);
      g_probe.evs.emplace_back(a, b);
      armed_ = true;
    }
  }
}
ProbeScope::~ProbeScope() {
  if (armed_) (void)hipEventRecord(g_probe.evs.back().second, st_);
}
}  // namespace coskad

extern "C" {
int coskad_probe_begin(int kernel, int Ci, int Co) {
  for (auto& e : coskad::g_probe.evs) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  coskad::g_probe.evs.clear();
  coskad::g_probe.kernel = kernel;
  coskad::g_probe.ci = Ci;
  coskad::g_probe.co = Co;
  coskad::g_probe.seen = 0;
  return COSKAD_OK;
}

/* Time every n-th matching launch only (n >= 1; stays until changed). */
int coskad_probe_stride(int n) {
  if (n < 1) return coskad::fail(COSKAD_ERR_ARG, "probe_stride: n=%d", n);
  coskad::g_probe.stride = n;
  return COSKAD_OK;
}

int coskad_probe_end(float* avg_ms, int* launches) {
  double tot = 0.0;
  int n = 0;
  for (auto& e : coskad::g_probe.evs) {
    float ms = 0.f;
    if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) {
      tot += ms;
      ++n;
    }
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  coskad::g_probe.evs.clear();
  coskad::g_probe.kernel = 0;
  if (avg_ms) *avg_ms = n ? (float)(tot / n) : 0.f;
  if (launches) *launches = n;
  return COSKAD_OK;
}

int coskad_abi_version(void) { return COSKAD_ABI_VERSION; }
const char* coskad_last_error(void) { return coskad::err_buf(); }
}
