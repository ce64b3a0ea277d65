// Error plumbing + version for libmio_hip.so.
#include "mio_common.h"

static thread_local std::string g_mio_err;

void mio_set_error(const std::string& msg) { g_mio_err = msg; }
int mio_fail(const std::string& msg) {
  g_mio_err = msg;
  return -1;
}

extern "C" int mio_version(void) { return MIO_VERSION; }
extern "C" const char* mio_last_error(void) { return g_mio_err.c_str(); }

#ifdef MIO_DIAG
#include <atomic>
static std::atomic<int> g_mio_dbg[8];
extern "C" void mio_dbg_set(int key, int value) {
  if (key >= 0 && key < 8) g_mio_dbg[key].store(value);
}
extern "C" int mio_dbg_get(int key) { return (key >= 0 && key < 8) ? g_mio_dbg[key].load() : 0; }
#endif
