// capi.hip -- library-level entry points and the per-thread error message.
#include "pda_common.h"

#include <string.h>

namespace pda {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace pda

PDA_API int pda_abi_version(void) { return PDA_POINTNET2_ABI_VERSION; }
PDA_API const char* pda_last_error(void) { return pda::g_err; }
PDA_API int pda_fp_contract_mode(void) { return PDA_FP_CONTRACT; }

// ellipsoid_query (pointnet2_api.cpp:16): present so the exported name set equals the reference module's; not on
// the PDA-SSD path (include/pda_pointnet2.h) and not implemented.
PDA_API int pda_ellipsoid_query(const float*, const float*, int32_t*, int, int, int, float, float, float, int, pda_stream_t) {
    pda::set_error("pda_ellipsoid_query: not implemented (no PDA-SSD configuration calls ellipsoid_query; "
                   "reference: ellipsoid_query.cpp:13)");
    return PDA_ERR_UNSUPPORTED;
}
