// core.hip — library identification, error text, device probe.
#include <string.h>

#include "pzn_common.h"

PZN_EXPORT int pzn_version(void) { return 100; /* 0.1.0 */ }

PZN_EXPORT const char* pzn_strerror(int status) {
  switch (status) {
    case PZN_OK: return "ok";
    case PZN_EINVAL: return "invalid argument (shape, null pointer or alignment)";
    case PZN_ELAUNCH: return "HIP launch failed (hipGetLastError != hipSuccess)";
    case PZN_EUNSUPPORTED: return "unsupported size for this build";
    case PZN_ENODEVICE: return "no usable gfx950 device";
    default: return "unknown pzn status";
  }
}

PZN_EXPORT int pzn_device_check(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return PZN_ENODEVICE;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return PZN_ENODEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return PZN_ENODEVICE;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? PZN_OK : PZN_ENODEVICE;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Per-kernel timer (measurement only): see PZN_LAUNCH in pzn_common.h.
#include <cxxabi.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

extern "C" __attribute__((visibility("default"))) int pzn_ktimer_is_on = 0;

namespace {
struct KRec {
  std::string name;
  hipEvent_t a, b;
};
std::mutex g_kt_mu;
std::vector<KRec*> g_kt_recs;
struct KRow {
  std::string name;
  int launches;
  double ms;
};
std::vector<KRow> g_kt_rows;

std::string kt_name(const void* fn, const char* text, hipStream_t st) {
  const char* raw = hipKernelNameRefByPtr(fn, st);
  if (raw == nullptr || raw[0] == 0) return std::string(text);
  int status = 0;
  char* dem = abi::__cxa_demangle(raw, nullptr, nullptr, &status);
  std::string s = (status == 0 && dem) ? dem : raw;
  free(dem);
  // "void (anonymous namespace)::kernel<...>(args)" -> "kernel<...>": the template arguments stay, the parameter list goes
  size_t depth = 0, cut = std::string::npos;
  for (size_t i = 0; i < s.size(); ++i) {
    if (s[i] == '<') ++depth;
    else if (s[i] == '>') --depth;
    else if (s[i] == '(' && depth == 0 && s.compare(i, 21, "(anonymous namespace)") != 0) { cut = i; break; }
  }
  if (cut != std::string::npos) s.resize(cut);
  for (const char* pre : {"void ", "(anonymous namespace)::"}) {
    size_t at;
    while ((at = s.find(pre)) != std::string::npos) s.erase(at, strlen(pre));
  }
  return s;
}
}  // namespace

pzn::KSpan::KSpan(const void* fn, const char* text, hipStream_t s) : st(s) {
  KRec* r = new KRec;
  r->name = kt_name(fn, text, s);
  (void)hipEventCreate(&r->a);
  (void)hipEventCreate(&r->b);
  (void)hipEventRecord(r->a, s);
  rec = r;
}

pzn::KSpan::~KSpan() {
  KRec* r = static_cast<KRec*>(rec);
  (void)hipEventRecord(r->b, st);
  std::lock_guard<std::mutex> lk(g_kt_mu);
  g_kt_recs.push_back(r);
}

PZN_EXPORT int pzn_ktimer_enable(int on) {
  pzn_ktimer_is_on = on ? 1 : 0;
  return PZN_OK;
}

// Waits for every recorded launch, folds the spans into rows (kernel name -> launches, summed milliseconds), clears the spans.
// -> number of rows (read them with pzn_ktimer_row), or a negative status.
PZN_EXPORT int pzn_ktimer_collect(void) {
  std::lock_guard<std::mutex> lk(g_kt_mu);
  std::map<std::string, KRow> rows;
  for (KRec* r : g_kt_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r->b) != hipSuccess || hipEventElapsedTime(&ms, r->a, r->b) != hipSuccess) ms = 0.f;
    KRow& row = rows[r->name];
    row.name = r->name;
    row.launches += 1;
    row.ms += ms;
    (void)hipEventDestroy(r->a);
    (void)hipEventDestroy(r->b);
    delete r;
  }
  g_kt_recs.clear();
  g_kt_rows.clear();
  for (auto& kv : rows) g_kt_rows.push_back(kv.second);
  return (int)g_kt_rows.size();
}

PZN_EXPORT int pzn_ktimer_row(int i, char* name, int cap, int* launches, double* ms) {
  std::lock_guard<std::mutex> lk(g_kt_mu);
  if (i < 0 || i >= (int)g_kt_rows.size() || name == nullptr || cap <= 0 || launches == nullptr || ms == nullptr) return PZN_EINVAL;
  strncpy(name, g_kt_rows[i].name.c_str(), (size_t)cap - 1);
  name[cap - 1] = 0;
  *launches = g_kt_rows[i].launches;
  *ms = g_kt_rows[i].ms;
  return PZN_OK;
}
