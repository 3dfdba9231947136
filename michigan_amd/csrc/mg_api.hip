// mg_api.hip -- ABI bookkeeping for libmichigan_hip.so
#include "mg_common.h"

thread_local char g_mg_err[512] = {0};

extern "C" int mg_abi_version(void) { return MG_ABI_VERSION; }
extern "C" const char* mg_last_error(void) { return g_mg_err; }
