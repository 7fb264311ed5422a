// In-kernel clock stamps of the DIAGNOSTIC build (-DSNERF_CLOCK_STAMP; tools/probes/build_variant.py clock ...): per workgroup
// the shader-clock ticks (s_memtime) and the 100 MHz reference ticks (s_memrealtime) between two points of a kernel; their
// ratio x 100 MHz is the clock the chip held inside it (MI355X_MICROARCH.md, "DVFS give-back" 6).  The stamps go to a buffer
// of their own that no kernel reads and no output depends on them; in the shipped build every macro below is empty.
#pragma once
#ifdef SNERF_CLOCK_STAMP
// (a kernel header included by several translation units defines its stamps in each of them: the buffer is per unit and the
// accessor weak -- it reads the unit the linker kept, which is the kernel's own when only one unit holds the stamped kernel)
#define SNERF_STAMP_DEFINE(name)                                                                                          \
    static __device__ unsigned long long name##_stamps[2 * 8192];                                                         \
    extern "C" __attribute__((weak)) int snerf_debug_clock_stamps_##name(unsigned long long* host, int pairs) {           \
        return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(name##_stamps), sizeof(unsigned long long) * 2 * (pairs < 8192 ? pairs : 8192)); \
    }
#define SNERF_STAMP_BEGIN() \
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime()
#define SNERF_STAMP_END(name)                                                                       \
    do {                                                                                            \
        if (threadIdx.x == 0) {                                                                     \
            name##_stamps[2 * (blockIdx.x & 8191)] = __builtin_amdgcn_s_memtime() - stamp_t0;       \
            name##_stamps[2 * (blockIdx.x & 8191) + 1] = __builtin_amdgcn_s_memrealtime() - stamp_r0; \
        }                                                                                           \
    } while (0)
#else
#define SNERF_STAMP_DEFINE(name)
#define SNERF_STAMP_BEGIN() do {} while (0)
#define SNERF_STAMP_END(name) do {} while (0)
#endif
