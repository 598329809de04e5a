// Error reporting, version and the kernel-selection option table of the C-ABI (include/idee_hip.h).
#include <stdarg.h>

#include <mutex>
#include <shared_mutex>

#include "common.h"

namespace idee {

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

// ---- kernel-selection options (idee_set_option / idee_get_option) ----
// One process-wide table, guarded by a reader / writer lock.  The environment variable of the same name supplies the INITIAL
// value only (read once, before the first access); after that the launch path never calls getenv(), so a host that flips a
// switch between launches (tests, bench.py's second timed region) does so through idee_set_option — which is safe against
// concurrent launches from autograd or loader threads, as setenv() against getenv() is not.
namespace {
struct Opt { const char* name; char val[32]; };
Opt g_opts[] = {
    {"IDEE_SWIN_MATH", ""},       // f32 (default) | f22 | split | fwd22
    {"IDEE_CONV_MATH", ""},       // f32 (default) | f22 | split | fwd22
    {"IDEE_CONV16", ""},          // regs: register-staged 16 -> 16 strip kernel
    {"IDEE_WGRAD16_REGS", ""},    // 1: register-staged 3^3 weight gradient
    {"IDEE_CONV32", ""},          // generic: 32 -> 32 on the generic kernel
    {"IDEE_DGRAD_TO16", ""},      // wide: (1,3,3) data gradient on conv_wide_kernel
    {"IDEE_CONV_BF16", ""},       // direct | lds: bf16 path kernel choice
    {"IDEE_CONV_HRUN", ""},       // test knob: rows per work unit
    {"IDEE_WGRAD_WIDE_SEG", ""},  // tuning knob: 32 | 64
    {"IDEE_WGRAD_WIDE", ""},      // regs: the register-staged exact wide weight gradient (default: LDS-DMA staged)
};
constexpr int kNumOpts = sizeof(g_opts) / sizeof(g_opts[0]);
std::shared_mutex g_opt_mu;
std::once_flag g_opt_once;

void opts_from_env() {
  for (Opt& o : g_opts) {
    const char* e = getenv(o.name);
    if (e) { strncpy(o.val, e, sizeof(o.val) - 1); o.val[sizeof(o.val) - 1] = 0; }
  }
}
Opt* find_opt(const char* name) {
  for (Opt& o : g_opts)
    if (strcmp(o.name, name) == 0) return &o;
  return nullptr;
}
}  // namespace

bool opt_is(const char* name, const char* value) {
  std::call_once(g_opt_once, opts_from_env);
  std::shared_lock<std::shared_mutex> lk(g_opt_mu);
  const Opt* o = find_opt(name);
  return o && strcmp(o->val, value) == 0;
}

int opt_int(const char* name) {
  std::call_once(g_opt_once, opts_from_env);
  std::shared_lock<std::shared_mutex> lk(g_opt_mu);
  const Opt* o = find_opt(name);
  return (o && o->val[0]) ? atoi(o->val) : 0;
}

}  // namespace idee

// A build with kernel timing-variant macros (IDEE_EXTRA_FLAGS="-DIDEE_DBG_SKIP=.." skips loads / MFMAs / stores in
// conv16_split_kernel; -DIDEE_RD_DEPTH changes a read-ahead depth) says so in its version string: the test session
// (tests/test_host_cpu.py) refuses a library whose version carries "TIMING-VARIANT".
#define IDEE_STR2(x) #x
#define IDEE_STR(x) IDEE_STR2(x)
#if defined(IDEE_DBG_SKIP) && (IDEE_DBG_SKIP != 0)
#define IDEE_VARIANT_TAG " TIMING-VARIANT IDEE_DBG_SKIP=" IDEE_STR(IDEE_DBG_SKIP) " (results invalid)"
#else
#define IDEE_VARIANT_TAG ""
#endif
#if defined(IDEE_RD_DEPTH)
#define IDEE_RD_TAG " IDEE_RD_DEPTH=" IDEE_STR(IDEE_RD_DEPTH)
#else
#define IDEE_RD_TAG ""
#endif
extern "C" const char* idee_version(void) { return "idee_hip 0.3.0 (gfx950)" IDEE_VARIANT_TAG IDEE_RD_TAG; }
extern "C" const char* idee_last_error(void) { return idee::err_buf(); }

extern "C" int idee_set_option(const char* name, const char* value) {
  using namespace idee;
  IDEE_CHECK_ARG(name != nullptr, "idee_set_option: name is NULL");
  std::call_once(g_opt_once, opts_from_env);
  std::unique_lock<std::shared_mutex> lk(g_opt_mu);
  Opt* o = find_opt(name);
  if (!o) return fail(IDEE_ERR_INVALID_ARG, "idee_set_option: unknown option '%s'", name);
  if (value && strlen(value) >= sizeof(o->val)) return fail(IDEE_ERR_INVALID_ARG, "idee_set_option: value of '%s' too long", name);
  strcpy(o->val, value ? value : "");
  return IDEE_OK;
}

extern "C" int idee_get_option(const char* name, char* buf, int buflen) {
  using namespace idee;
  IDEE_CHECK_ARG(name != nullptr && buf != nullptr && buflen > 0, "idee_get_option: NULL argument");
  std::call_once(g_opt_once, opts_from_env);
  std::shared_lock<std::shared_mutex> lk(g_opt_mu);
  const Opt* o = find_opt(name);
  if (!o) return fail(IDEE_ERR_INVALID_ARG, "idee_get_option: unknown option '%s'", name);
  strncpy(buf, o->val, buflen - 1);
  buf[buflen - 1] = 0;
  return IDEE_OK;
}

// Refuses to run anywhere but on a gfx950 device; returns IDEE_OK when device `dev` qualifies.
extern "C" int idee_check_device(int dev) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return idee::fail(IDEE_ERR_LAUNCH, "hipGetDeviceProperties(%d): %s", dev, hipGetErrorString(e));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return idee::fail(IDEE_ERR_UNSUPPORTED, "device %d is %s; this library is built for gfx950 only", dev, prop.gcnArchName);
  return IDEE_OK;
}
