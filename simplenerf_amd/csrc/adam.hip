// O1: Adam for all parameter tensors of the model in one launch per 64 tensors.
//
// The reference steps ~90 small tensors through torch.optim.Adam (a dozen elementwise kernels per tensor on its
// single-tensor path).  The update is pure streaming -- 16 B read and 12 B written per parameter, 2.27 M parameters
// -> 63 MB per step, ~10 us at HBM speed -- so the only thing that matters is doing it in one pass and one launch.
// A block owns 1024 consecutive elements of ONE tensor; the block -> tensor map is a binary search in a kernarg table.
#include "snerf_common.h"

#include <cmath>

#include "../../include/simplenerf_train.h"

namespace {

constexpr int kTensorsPerLaunch = 64;
constexpr int kElemsPerBlock = 1024;

struct AdamTable {
    float* param[kTensorsPerLaunch];
    const float* grad[kTensorsPerLaunch];
    float* exp_avg[kTensorsPerLaunch];
    float* exp_avg_sq[kTensorsPerLaunch];
    unsigned size[kTensorsPerLaunch];
    unsigned first_block[kTensorsPerLaunch + 1];
    int count;
    float one_minus_beta1, beta2, one_minus_beta2, neg_step_size, bias2_sqrt, eps;
    const snerf_iteration* at;     // snerf_adam_step_at: the two step-dependent factors come from the device-resident record
};

__global__ void __launch_bounds__(256) adam_kernel(AdamTable t) {
    // (locals, never a store into `t`: a modified by-value table is copied to scratch by every thread -- 2.6 KB each, which
    // made this kernel 160 us instead of 10; tools/isa_stats.sh adam must show scratch 0)
    const float neg_step_size = t.at ? t.at->adam_neg_step_size : t.neg_step_size;
    const float bias2_sqrt = t.at ? t.at->adam_bias2_sqrt : t.bias2_sqrt;
    int lo = 0, hi = t.count;                     // largest i with first_block[i] <= blockIdx.x
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t.first_block[mid] <= blockIdx.x) lo = mid; else hi = mid;
    }
    float* __restrict__ p = t.param[lo];
    const float* __restrict__ g = t.grad[lo];
    float* __restrict__ m = t.exp_avg[lo];
    float* __restrict__ v = t.exp_avg_sq[lo];
    const unsigned base = (blockIdx.x - t.first_block[lo]) * kElemsPerBlock;
#pragma unroll
    for (int k = 0; k < kElemsPerBlock / 256; ++k) {
        const unsigned i = base + k * 256 + threadIdx.x;
        if (i < t.size[lo]) {
            const float grad = g[i];
            const float m1 = fmaf(t.one_minus_beta1, grad - m[i], m[i]);
            const float v1 = fmaf(t.one_minus_beta2 * grad, grad, v[i] * t.beta2);
            const float denom = sqrtf(v1) / bias2_sqrt + t.eps;
            m[i] = m1;
            v[i] = v1;
            p[i] = p[i] + (neg_step_size * m1) / denom;
        }
    }
}

}  // namespace

static int adam_impl(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                     const long long* sizes, int num_tensors, long long step, double lr, double beta1, double beta2, double eps,
                     const snerf_iteration* at, snerf_stream_t stream) {
    SNERF_REQUIRE(params && grads && exp_avg && exp_avg_sq && sizes, "adam_step: NULL table");
    SNERF_REQUIRE(num_tensors >= 0, "adam_step: negative tensor count");
    SNERF_REQUIRE(at || step >= 1, "adam_step: step must be >= 1, got %lld", step);
    SNERF_REQUIRE(beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0, "adam_step: bad hyper-parameters");
    AdamTable t;
    t.at = at;
    t.one_minus_beta1 = (float)(1.0 - beta1);
    t.beta2 = (float)beta2;
    t.one_minus_beta2 = (float)(1.0 - beta2);
    t.neg_step_size = 0.0f;
    t.bias2_sqrt = 1.0f;
    if (!at) {
        // python: bias_correction = 1 - beta ** step ; step_size = lr / bc1 ; bc2 ** 0.5   (all in double)
        const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
        t.neg_step_size = (float)(-(lr / bc1));
        t.bias2_sqrt = (float)std::pow(bc2, 0.5);
    }
    t.eps = (float)eps;
    int i = 0;
    while (i < num_tensors) {
        t.count = 0;
        unsigned blocks = 0;
        for (; i < num_tensors && t.count < kTensorsPerLaunch; ++i) {
            if (!grads[i] || sizes[i] == 0) continue;
            SNERF_REQUIRE(params[i] && exp_avg[i] && exp_avg_sq[i], "adam_step: tensor %d has a NULL pointer", i);
            SNERF_REQUIRE(sizes[i] > 0 && sizes[i] < (1LL << 31), "adam_step: tensor %d size %lld outside (0, 2^31)", i, sizes[i]);
            const int k = t.count++;
            t.param[k] = params[i];
            t.grad[k] = grads[i];
            t.exp_avg[k] = exp_avg[i];
            t.exp_avg_sq[k] = exp_avg_sq[i];
            t.size[k] = (unsigned)sizes[i];
            t.first_block[k] = blocks;
            blocks += (unsigned)((sizes[i] + kElemsPerBlock - 1) / kElemsPerBlock);
        }
        if (t.count == 0) break;
        t.first_block[t.count] = blocks;
        hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t);
        if (int st = snerf::check_launch("adam_step")) return st;
    }
    return SNERF_OK;
}

extern "C" int snerf_adam_step(float* const* params, const float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const long long* sizes, int num_tensors, long long step,
                               double lr, double beta1, double beta2, double eps, snerf_stream_t stream) {
    return adam_impl(params, grads, exp_avg, exp_avg_sq, sizes, num_tensors, step, lr, beta1, beta2, eps, nullptr, stream);
}

extern "C" int snerf_adam_step_at(float* const* params, const float* const* grads, float* const* exp_avg,
                                  float* const* exp_avg_sq, const long long* sizes, int num_tensors,
                                  const snerf_iteration* current, double beta1, double beta2, double eps, snerf_stream_t stream) {
    SNERF_REQUIRE(current, "adam_step_at: NULL iteration record");
    return adam_impl(params, grads, exp_avg, exp_avg_sq, sizes, num_tensors, 0, 0.0, beta1, beta2, eps, current, stream);
}
