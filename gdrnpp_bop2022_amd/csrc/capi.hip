// libgdrnpp_hip.so — library-level entry points (version, last error).
#include "common.hpp"
#include <cstring>

namespace gdrnpp {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace gdrnpp

extern "C" {
int gdrnpp_version(void) { return 100; /* 0.1.0 */ }
const char* gdrnpp_last_error(void) { return gdrnpp::g_err; }
}
