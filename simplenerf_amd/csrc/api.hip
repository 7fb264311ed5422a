// ABI bookkeeping: version and the thread-local error message.
#include "snerf_common.h"

namespace snerf {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace snerf

extern "C" int snerf_abi_version(void) { return SNERF_ABI_VERSION; }
extern "C" const char* snerf_last_error(void) { return snerf::error_buffer(); }

// ---------------------------------------------------------------------------------------------------------------
// Opt-in timing of the dominant kernels (bench.py's roofline leg): while enabled, every snerf_mlp_forward[_train]
// launch (kind 0) and every snerf_mlp_backward call (kind 1) is bracketed by a pair of HIP events recorded on the
// stream it is enqueued on -- also when it is issued from inside snerf_render_forward / _backward.  Off by default;
// the hot path then pays one relaxed atomic load.  A launch enqueued while its stream is being captured into a graph
// is not timed (an event record would become a graph node and the pre-created events would be reused by every replay);
// launches that find the event table full are counted (snerf_profile_dropped) instead of vanishing silently.
#include <atomic>
#include <mutex>
#include <vector>

namespace snerf {
namespace {
struct ProfileEntry { hipEvent_t start, stop; long long samples; int kind; };
std::mutex g_profile_mutex;
std::vector<ProfileEntry> g_profile;     // pre-created event pairs
int g_profile_used = 0;
long long g_profile_dropped = 0;         // launches seen while enabled that got no event pair (table full)
std::atomic<int> g_profile_on{0};
}  // namespace

int profile_begin(int kind, hipStream_t stream, long long samples) {
    if (!g_profile_on.load(std::memory_order_relaxed)) return -1;
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capture) != hipSuccess) {
        // e.g. the legacy stream while another stream captures in global mode (hipErrorStreamCaptureImplicit): not timed, and
        // the error must not stay behind for an unrelated launch's hipGetLastError
        (void)hipGetLastError();
        return -1;
    }
    if (capture != hipStreamCaptureStatusNone) return -1;
    std::lock_guard<std::mutex> lock(g_profile_mutex);
    if (g_profile_used >= (int)g_profile.size()) { ++g_profile_dropped; return -1; }
    const int slot = g_profile_used++;
    g_profile[slot].samples = samples;
    g_profile[slot].kind = kind;
    (void)hipEventRecord(g_profile[slot].start, stream);
    return slot;
}

void profile_end(int slot, hipStream_t stream) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lock(g_profile_mutex);
    if (slot < (int)g_profile.size()) (void)hipEventRecord(g_profile[slot].stop, stream);
}
}  // namespace snerf

extern "C" int snerf_profile_enable(int capacity) {
    using namespace snerf;
    std::lock_guard<std::mutex> lock(g_profile_mutex);
    g_profile_on.store(0);
    for (ProfileEntry& e : g_profile) { (void)hipEventDestroy(e.start); (void)hipEventDestroy(e.stop); }
    g_profile.clear();
    g_profile_used = 0;
    g_profile_dropped = 0;
    if (capacity <= 0) return SNERF_OK;
    g_profile.resize((size_t)capacity);
    for (ProfileEntry& e : g_profile) {
        if (hipEventCreate(&e.start) != hipSuccess || hipEventCreate(&e.stop) != hipSuccess)
            return fail(SNERF_E_HIP, "profile_enable: hipEventCreate failed");
    }
    g_profile_on.store(1);
    return SNERF_OK;
}

extern "C" int snerf_profile_collect(int kind, float* milliseconds, long long* samples, int capacity) {
    using namespace snerf;
    std::lock_guard<std::mutex> lock(g_profile_mutex);
    int count = 0;
    for (int i = 0; i < g_profile_used; ++i) {
        const ProfileEntry& e = g_profile[i];
        if (e.kind != kind) continue;
        if (count >= capacity) break;
        float ms = 0.0f;
        if (hipEventSynchronize(e.stop) != hipSuccess || hipEventElapsedTime(&ms, e.start, e.stop) != hipSuccess)
            return fail(SNERF_E_HIP, "profile_collect: event %d not readable", i);
        if (milliseconds) milliseconds[count] = ms;
        if (samples) samples[count] = e.samples;
        ++count;
    }
    return count;
}

extern "C" int snerf_profile_reset(void) {
    std::lock_guard<std::mutex> lock(snerf::g_profile_mutex);
    snerf::g_profile_used = 0;
    snerf::g_profile_dropped = 0;
    return SNERF_OK;
}

extern "C" long long snerf_profile_dropped(void) {
    std::lock_guard<std::mutex> lock(snerf::g_profile_mutex);
    return snerf::g_profile_dropped;
}

// ---------------------------------------------------------------------------------------------------------------
// fp16 range flag: one int per device ordinal in pinned (host-coherent, device-mapped) memory.  Kernels store to it with a
// system-scope atomic when they met a non-finite fp16 operand; the host reads it without synchronising.
namespace snerf {
namespace {
std::mutex g_range_mutex;
int* g_range_table = nullptr;      // 64 ints
constexpr int kRangeDevices = 64;
}  // namespace

int* range_flag() {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= kRangeDevices) {
        (void)fail(SNERF_E_HIP, "range flag: no current HIP device (or ordinal >= %d)", kRangeDevices);
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_range_mutex);
    if (!g_range_table) {
        void* p = nullptr;
        const hipError_t e = hipHostMalloc(&p, kRangeDevices * sizeof(int), hipHostMallocPortable | hipHostMallocMapped);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)fail(SNERF_E_HIP, "range flag: hipHostMalloc: %s (the first fp16-mode call of a process must not be inside "
                                    "a graph capture)", hipGetErrorString(e));
            return nullptr;
        }
        g_range_table = static_cast<int*>(p);
        for (int i = 0; i < kRangeDevices; ++i) g_range_table[i] = 0;
    }
    return g_range_table + device;
}

int report_range(const char* what) {
    int* flag = range_flag();
    if (!flag) return SNERF_E_HIP;
    const int bits = __atomic_exchange_n(flag, 0, __ATOMIC_ACQ_REL);
    if (bits == 0) return SNERF_OK;
    return fail(SNERF_E_RANGE, "%s: an earlier fp16-mode launch on this device met %s%s%s outside the fp16 range (|v| > 65504): "
                               "its results are invalid -- use hip_precision 'fp32' for this model", what,
                (bits & kRangeActivation) ? "a hidden activation or encoded input" : "", (bits & kRangeActivation) && (bits & kRangeWeight) ? " and " : "",
                (bits & kRangeWeight) ? "a weight" : "");
}
}  // namespace snerf

extern "C" int snerf_range_status(int clear) {
    int* flag = snerf::range_flag();
    if (!flag) return SNERF_E_HIP;
    return clear ? __atomic_exchange_n(flag, 0, __ATOMIC_ACQ_REL) : __atomic_load_n(flag, __ATOMIC_ACQUIRE);
}

// ---------------------------------------------------------------------------------------------------------------
// packed-format registry (snerf_common.h)
#include <unordered_map>
namespace snerf {
namespace {
std::mutex g_formats_mutex;
std::unordered_map<const float*, unsigned> g_formats;
constexpr size_t kFormatsCapacity = 1 << 16;     // distinct packed buffers remembered; beyond it the table starts over

const char* formats_text(unsigned formats, char* buffer, size_t size) {
    snprintf(buffer, size, "%s%s%s%s%s", (formats & kPackFp32) ? "fp32 " : "", (formats & kPackF16) ? "f16-training " : "",
             (formats & kPackF16Eval) ? "f16-rendering " : "", (formats & kPackBf16) ? "bf16-training " : "",
             (formats & kPackBf16Eval) ? "bf16-rendering " : "");
    return buffer;
}
}  // namespace

void packed_formats_record(const float* packed, unsigned formats) {
    std::lock_guard<std::mutex> lock(g_formats_mutex);
    if (g_formats.size() >= kFormatsCapacity) g_formats.clear();
    g_formats[packed] = formats;
}

int packed_formats_require(const float* packed, unsigned needed, const char* what) {
    unsigned have;
    {
        std::lock_guard<std::mutex> lock(g_formats_mutex);
        auto it = g_formats.find(packed);
        if (it == g_formats.end()) return SNERF_OK;       // never packed through this library instance: cannot tell
        have = it->second;
    }
    if ((have & needed) == needed) return SNERF_OK;
    char a[96], b[96];
    return fail(SNERF_E_INVALID, "%s: this packed buffer holds the operand formats [ %s] (its last snerf_mlp_pack_for), the call reads "
                                 "[ %s]: the missing streams are zero-filled -- pack with snerf_mlp_pack, or snerf_mlp_pack_for the "
                                 "precision and mode it is used at", what, formats_text(have, a, sizeof a),
                formats_text(needed, b, sizeof b));
}

unsigned packed_formats_needed(int precision, bool training) {
    switch (precision) {
        case SNERF_PRECISION_FP32: return kPackFp32;
        case SNERF_PRECISION_F16X3: case SNERF_PRECISION_F16: case SNERF_PRECISION_F16S8:
            return kPackF16 | (training ? 0u : kPackF16Eval);
        case SNERF_PRECISION_BF16: case SNERF_PRECISION_BF16S8: return kPackBf16 | (training ? 0u : kPackBf16Eval);
        default: return 0u;
    }
}
}  // namespace snerf
