// CPU unit test of csrc/device_once.h: the dynamic-LDS attribute of a kernel is per DEVICE, so the
// "already raised" flag must be too.  Drives DeviceOnce with fake device ordinals and a counting setter.
#include <cstdio>
#include <thread>
#include <vector>

#include "../../simplenerf_amd/csrc/device_once.h"

static int fails = 0;
#define CHECK(cond)                                                     \
    do {                                                                \
        if (!(cond)) {                                                  \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++fails;                                                    \
        }                                                               \
    } while (0)

int main() {
    snerf::DeviceOnce once;
    int calls[300] = {};
    auto setter = [&](int dev) { return [&calls, dev]() { ++calls[dev]; return 0; }; };
    // two devices driven by one process: each gets its own setter call, exactly once
    CHECK(once.run(0, setter(0)) == 0 && once.run(0, setter(0)) == 0);
    CHECK(!once.is_done(1));
    CHECK(once.run(1, setter(1)) == 0 && once.run(1, setter(1)) == 0 && once.run(0, setter(0)) == 0);
    CHECK(calls[0] == 1 && calls[1] == 1);
    // ordinals across word boundaries, and out-of-table ordinals (setter runs every time, still succeeds)
    for (int dev : {63, 64, 127, 255}) {
        CHECK(once.run(dev, setter(dev)) == 0 && once.run(dev, setter(dev)) == 0 && calls[dev] == 1 && once.is_done(dev));
    }
    CHECK(once.run(256, setter(256)) == 0 && once.run(256, setter(256)) == 0 && calls[256] == 2 && !once.is_done(256));
    CHECK(once.run(-1, setter(299)) == 0 && calls[299] == 1 && !once.is_done(-1));
    // a failing setter is retried (the error must repeat, not be swallowed by the flag)
    int attempts = 0;
    auto failing = [&]() { ++attempts; return attempts < 3 ? -3 : 0; };
    CHECK(once.run(7, failing) == -3 && !once.is_done(7));
    CHECK(once.run(7, failing) == -3 && once.run(7, failing) == 0 && once.is_done(7) && once.run(7, failing) == 0 && attempts == 3);
    // threads, one per device (the DataParallel shape): every device done, every setter ran at least once
    snerf::DeviceOnce shared;
    std::atomic<int> ran[8] = {};
    std::vector<std::thread> pool;
    for (int t = 0; t < 32; ++t)
        pool.emplace_back([&, t]() {
            for (int i = 0; i < 1000; ++i) shared.run(t % 8, [&]() { ran[t % 8].fetch_add(1); return 0; });
        });
    for (auto& th : pool) th.join();
    for (int d = 0; d < 8; ++d) CHECK(shared.is_done(d) && ran[d].load() >= 1 && ran[d].load() <= 4);
    if (fails == 0) std::printf("device_once_test: OK\n");
    return fails == 0 ? 0 : 1;
}
