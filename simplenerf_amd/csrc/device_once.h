// Per-device "done once" flags for kernel attributes.
//
// hipFuncAttributeMaxDynamicSharedMemorySize is an attribute of a kernel ON ONE DEVICE.  A process that drives several
// devices -- the reference's DataParallel mode runs one Python thread per device into model.forward (SURVEY 8b,
// "Threading") -- must raise it on each of them; a process-wide `static bool` would raise it on the first device only
// and the launch on the second would fail (the fp32 forward needs 84 KB of dynamic LDS, above the 64 KB default).
//
// One DeviceOnce per launch site: a bit per device ordinal, set after the setter succeeded there.  Racing threads may
// both run the setter (it is idempotent); a failed setter leaves the bit clear so that the error repeats.  No HIP types:
// the unit test (tests/native/device_once_test.cpp) drives it with fake device ids on the CPU.
#pragma once
#include <atomic>

namespace snerf {

struct DeviceOnce {
    static constexpr int kWords = 4;  // 256 device ordinals; beyond that the setter simply runs every time
    std::atomic<unsigned long long> done[kWords] = {};

    bool is_done(int device) const {
        if (device < 0 || device >= 64 * kWords) return false;
        return (done[device >> 6].load(std::memory_order_acquire) >> (device & 63)) & 1ull;
    }

    // Runs `setter()` (returns 0 on success) unless it already succeeded for `device`; returns the setter's status.
    template <class Setter>
    int run(int device, Setter&& setter) {
        if (is_done(device)) return 0;
        const int status = setter();
        if (status == 0 && device >= 0 && device < 64 * kWords)
            done[device >> 6].fetch_or(1ull << (device & 63), std::memory_order_release);
        return status;
    }
};

}  // namespace snerf
