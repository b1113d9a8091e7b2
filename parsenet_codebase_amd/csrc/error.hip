// Thread-local error string behind the C ABI (include/parsenet_hip.h).
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void pn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* pn_last_error(void) { return g_err; }

// 2: edge-conv backward takes a workspace, mean-shift backward reduces its partial sums itself,
//    bf16 x 3 mean-shift entry points
extern "C" int pn_abi_version(void) { return 18; }
