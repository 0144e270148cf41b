// Shared device helpers and host-side error plumbing for libmsgl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/msgl_hip.h"

namespace msgl {

// ---- host side -------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

// Kernel attributes (an LDS size beyond the default 64 KiB) belong to a (kernel, device) pair: every launch site keeps
// one of these and configures its kernel the first time it launches on a device.  Two threads racing on that first
// launch both configure, which is harmless; the bit is set only after success.
struct PerDeviceOnce {
    unsigned long long done[4] = {0, 0, 0, 0};  // devices 0..255
    template <typename F>
    bool ensure(F&& configure) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 255) return configure();
        const unsigned long long bit = 1ull << (dev & 63);
        if (__atomic_load_n(&done[dev >> 6], __ATOMIC_ACQUIRE) & bit) return true;
        if (!configure()) return false;
        __atomic_fetch_or(&done[dev >> 6], bit, __ATOMIC_RELEASE);
        return true;
    }
};
int device_cus();
// Current override of a kernel-form choice (msgl_set_variant), MSGL_VARIANT_AUTO = the library chooses.
int variant(int which);

#define MSGL_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            ::msgl::set_error(__VA_ARGS__); \
            return MSGL_E_INVALID;         \
        }                                  \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device side -----------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

// bf16 <-> f32 on raw bits.  A bf16 is the upper half of the f32 with the same value.
__device__ __forceinline__ float bf16_lo(uint32_t packed) { return __builtin_bit_cast(float, packed << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t packed) { return __builtin_bit_cast(float, packed & 0xffff0000u); }

// Round-to-nearest-even f32 -> bf16 bits; NaN stays NaN (hardware v_cvt_pk_bf16_f32).
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    bf16x2_t r;
    r[0] = (__bf16)lo;
    r[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, r);
}
// Value of x after one rounding to bf16 (what a torch bf16 elementwise op returns).
__device__ __forceinline__ float round_bf16(float x) { return (float)(__bf16)x; }

// a.lo*b.lo + a.hi*b.hi + c on packed bf16 pairs, f32 accumulate (v_dot2c_f32_bf16).
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}

// a.lo*b.lo + a.hi*b.hi: the three-operand encoding with a literal zero addend, for the first step of a chain
// (the compiler's choice, the accumulate-in-place encoding, would need its destination zeroed first).
__device__ __forceinline__ float dot2_bf16_first(uint32_t a, uint32_t b) {
    float r;
    asm("v_dot2_f32_bf16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// DPP lane movement inside a row of 16 lanes; fuses into the consuming VALU op.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_ROR8 = 0x128;    // rotate the 16-lane row by 8: the other octet
constexpr int DPP_ROW_BCAST0 = 0x150;  // row_newbcast: lane 0 of the row to all of its lanes
constexpr int DPP_ROW_BCAST8 = 0x158;  // row_newbcast: lane 8 of the row to all of its lanes

// Sum over the 16 lanes of a DPP row; every lane of the row gets the total.
__device__ __forceinline__ float row16_sum(float x) {
    x += dpp_f32<DPP_XOR1>(x);
    x += dpp_f32<DPP_XOR2>(x);
    x += dpp_f32<DPP_ROW_HALF_MIRROR>(x);  // quads hold equal values: mirror == xor 4
    x += dpp_f32<DPP_ROW_MIRROR>(x);       // halves hold equal values: mirror == xor 8
    return x;
}

// Sum over the 8 lanes of an octet (half a DPP row); every lane of the octet gets the total.
__device__ __forceinline__ float octet_sum(float x) {
    x += dpp_f32<DPP_XOR1>(x);
    x += dpp_f32<DPP_XOR2>(x);
    x += dpp_f32<DPP_ROW_HALF_MIRROR>(x);  // quads hold equal values: mirror == xor 4
    return x;
}

// Sum over all 64 lanes of a wave; every lane gets the total.
__device__ __forceinline__ float wave_sum(float x) {
    x = row16_sum(x);
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

}  // namespace msgl
