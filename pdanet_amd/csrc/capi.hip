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
