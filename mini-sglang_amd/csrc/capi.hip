// Error plumbing and version query of libmsgl_hip.so.
#include "msgl_common.h"

namespace msgl {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Launch errors (bad configuration, missing code object) surface here; execution faults
// surface at the caller's next synchronisation, as with any asynchronous HIP work.
int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MSGL_OK;
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return MSGL_E_LAUNCH;
}

// Compute units of the current device (cached per device id; 256 if the query fails).
int device_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

static int g_variant[MSGL_VARIANT_COUNT] = {MSGL_VARIANT_AUTO, MSGL_VARIANT_AUTO, MSGL_VARIANT_AUTO, MSGL_VARIANT_AUTO,
                                            MSGL_VARIANT_AUTO, MSGL_VARIANT_AUTO, MSGL_VARIANT_AUTO};
int variant(int which) { return __atomic_load_n(&g_variant[which], __ATOMIC_RELAXED); }

}  // namespace msgl

extern "C" int msgl_set_variant(int which, int value) {
    MSGL_REQUIRE(which >= 0 && which < MSGL_VARIANT_COUNT, "msgl_set_variant: unknown choice %d", which);
    return __atomic_exchange_n(&msgl::g_variant[which], value < 0 ? MSGL_VARIANT_AUTO : value, __ATOMIC_RELAXED);
}
extern "C" int msgl_get_variant(int which) {
    MSGL_REQUIRE(which >= 0 && which < MSGL_VARIANT_COUNT, "msgl_get_variant: unknown choice %d", which);
    return msgl::variant(which);
}

extern "C" int msgl_abi_version(void) { return 17; }
extern "C" const char* msgl_last_error(void) { return msgl::g_err; }
