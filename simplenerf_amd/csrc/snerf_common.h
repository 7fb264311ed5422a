// Shared host-side helpers for the C-ABI entry points (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/simplenerf_hip.h"
#include "device_once.h"

namespace snerf {

char* error_buffer();  // thread-local, 512 bytes (api.hip)

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SNERF_E_HIP, "%s: %s", what, hipGetErrorString(e));
    return SNERF_OK;
}

// Opt-in event timing of the dominant kernels (api.hip, snerf_profile_enable): brackets the enclosing scope's launches.
int profile_begin(int kind, hipStream_t stream, long long samples);
void profile_end(int slot, hipStream_t stream);
struct ProfileScope {
    int slot;
    hipStream_t stream;
    ProfileScope(int kind, hipStream_t s, long long samples) : slot(profile_begin(kind, s, samples)), stream(s) {}
    ~ProfileScope() { profile_end(slot, stream); }
    ProfileScope(const ProfileScope&) = delete;
    ProfileScope& operator=(const ProfileScope&) = delete;
};

// Raise a kernel's dynamic-LDS cap on the CURRENT device, once per device and launch site (device_once.h).
inline int raise_dynamic_lds(DeviceOnce& once, const void* kernel, int bytes, const char* what) {
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return fail(SNERF_E_HIP, "%s: hipGetDevice: %s", what, hipGetErrorString(e));
    return once.run(device, [&]() -> int {
        hipError_t err = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (err != hipSuccess) return fail(SNERF_E_HIP, "%s: hipFuncSetAttribute: %s", what, hipGetErrorString(err));
        return SNERF_OK;
    });
}

// fp16 range flag (api.hip; include/simplenerf_hip.h "Range"): device-visible pointer to the CURRENT device's word of a
// pinned host table (NULL + error message when it cannot be allocated), and the report every fp16-mode entry point makes
// before enqueuing: SNERF_E_RANGE (flag cleared) if an earlier launch on this device raised it.
constexpr int kRangeActivation = 1, kRangeWeight = 2;
int* range_flag();
int report_range(const char* what);

// Which operand formats a packed weight buffer holds: snerf_mlp_pack_for writes only what one precision reads in one
// mode and ZEROES the rest, so a buffer packed for (f16, training) read by an eval render -- or an fp32-packed one read at f16 --
// used to return bias-only output with SNERF_OK.  The library keeps, on the host, the format mask of every buffer a pack call
// has written (keyed by the buffer's address; updated at enqueue time, which is stream order for a caller that packs and
// consumes on one stream); every entry point that consumes a weight stream asks for the bits it reads and fails with
// SNERF_E_INVALID before enqueuing anything when the buffer's last pack did not write them.  A buffer the registry has never
// seen (a device-to-device copy of a packed buffer) is not checked.
constexpr unsigned kPackF16 = 1u, kPackF16Eval = 2u, kPackBf16 = 4u, kPackBf16Eval = 8u, kPackFp32 = 16u, kPackAll = 31u;
void packed_formats_record(const float* packed, unsigned formats);
int packed_formats_require(const float* packed, unsigned needed, const char* what);
// the mask an entry point reads: precision (enum snerf_precision) x (training layout | rendering layout)
unsigned packed_formats_needed(int precision, bool training);

// Grid for a grid-stride elementwise kernel: enough blocks to fill 256 CUs x 8 blocks, never more than the work.
inline unsigned stride_grid(long long work, int block) {
    long long blocks = (work + block - 1) / block;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace snerf

#define SNERF_REQUIRE(cond, ...) \
    do {                         \
        if (!(cond)) return snerf::fail(SNERF_E_INVALID, __VA_ARGS__); \
    } while (0)
