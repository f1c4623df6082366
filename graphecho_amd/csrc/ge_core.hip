// Library-wide state: last-error string and version query.
#include "ge_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void ge_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static thread_local char g_kernel[160] = "";

void ge_note_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
}

static thread_local hipEvent_t g_split_event = nullptr;

void ge_record_split_event(hipStream_t st) {
  if (g_split_event) (void)hipEventRecord(g_split_event, st);
}

extern "C" {
// Measurement hook (bench.py): while an event is set, ge_conv2d_wgrad / ge_conv2d_f16_wgrad record it on their stream
// between the weight-gradient kernel and its slab reduce, so the two launches of one call can be timed apart.
void ge_set_wgrad_split_event(void* event) { g_split_event = (hipEvent_t)event; }
const char* ge_last_error(void) { return g_err; }
const char* ge_last_conv_kernel(void) { return g_kernel; }
int ge_abi_version(void) { return 1; }
// Number of HIP devices visible (0 when there is no GPU or the runtime cannot initialise).
int ge_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
}
