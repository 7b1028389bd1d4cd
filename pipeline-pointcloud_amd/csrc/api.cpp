// Error reporting and version query of the mi3dgs C-ABI (include/mi3dgs.h).
#include "common.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

int mi_set_error(const char* what, hipError_t e, const char* file, int line) {
    snprintf(g_err, sizeof(g_err), "mi3dgs: %s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    return 1;
}

int mi_set_error_msg(const char* msg) {
    snprintf(g_err, sizeof(g_err), "mi3dgs: %s", msg);
    return 2;
}

extern "C" const char* mi3dgs_last_error(void) { return g_err; }
extern "C" int mi3dgs_abi_version(void) { return 1; }
extern "C" int mi3dgs_splat_stride(void) { return SPLAT_STRIDE; }
extern "C" int mi3dgs_grad_stride(void) { return GRAD_STRIDE; }
