// Optional per-kernel timing with HIP events, recorded on the stream the kernel is launched
// on.  Off by default (zero overhead beyond one branch per launch).  bench.py switches it
// on for a few steps to obtain the average launch duration of each kernel family.
#include "common.h"
#include <mutex>
#include <string>
#include <vector>

struct PnProfSpan {
  const char* name;
  hipEvent_t a, b;
  bool bad = false;
};

static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<PnProfSpan> g_spans;

struct PnProfAgg {
  std::string name;
  double ms;
  long long calls;
};
static std::vector<PnProfAgg> g_agg;

bool pn_prof_enabled() { return g_prof_on; }

void pn_prof_begin(const char* name, hipStream_t s, int* token) {
  *token = -1;
  if (!g_prof_on) return;
  PnProfSpan sp;
  sp.name = name;
  if (hipEventCreate(&sp.a) != hipSuccess) return;
  if (hipEventCreate(&sp.b) != hipSuccess || hipEventRecord(sp.a, s) != hipSuccess) {
    (void)hipEventDestroy(sp.a);      // an unrecorded span is dropped, never half-timed
    return;
  }
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_spans.push_back(sp);
  *token = (int)g_spans.size() - 1;
}

void pn_prof_end(hipStream_t s, int token) {
  if (token < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (hipEventRecord(g_spans[token].b, s) != hipSuccess) g_spans[token].bad = true;
}

static void prof_resolve() {
  for (auto& sp : g_spans) {
    float ms = 0.f;
    const bool ok = !sp.bad && hipEventSynchronize(sp.b) == hipSuccess &&
                    hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess;
    (void)hipEventDestroy(sp.a);
    (void)hipEventDestroy(sp.b);
    if (!ok) continue;                // a span whose events failed contributes nothing
    bool found = false;
    for (auto& a : g_agg)
      if (a.name == sp.name) {
        a.ms += ms;
        a.calls += 1;
        found = true;
        break;
      }
    if (!found) g_agg.push_back({sp.name, (double)ms, 1});
  }
  g_spans.clear();
}

extern "C" void pn_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
}

extern "C" void pn_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  prof_resolve();
  g_agg.clear();
}

extern "C" int pn_prof_count(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  prof_resolve();
  return (int)g_agg.size();
}

extern "C" int pn_prof_get(int i, char* name, int name_len, double* total_ms, long long* calls) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (i < 0 || i >= (int)g_agg.size()) return PN_ERR_ARG;
  snprintf(name, name_len, "%s", g_agg[i].name.c_str());
  *total_ms = g_agg[i].ms;
  *calls = g_agg[i].calls;
  return PN_OK;
}
