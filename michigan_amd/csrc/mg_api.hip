// mg_api.hip -- ABI bookkeeping for libmichigan_hip.so
#include "mg_common.h"

thread_local char g_mg_err[512] = {0};

extern "C" int mg_abi_version(void) { return MG_ABI_VERSION; }
extern "C" const char* mg_last_error(void) { return g_mg_err; }
extern "C" int mg_sizeof_desc(int32_t which)
{
    return which == 0 ? (int)sizeof(mg_conv_desc) : which == 1 ? (int)sizeof(mg_wgrad_desc) : which == 2 ? (int)sizeof(mg_grad_slot) : which == 3 ? (int)sizeof(mg_pack_job) : which == 4 ? (int)sizeof(mg_sn_layer) : which == 5 ? (int)sizeof(mg_norm_apply2_desc) : which == 6 ? (int)sizeof(mg_pyramid_desc) : -1;
}
