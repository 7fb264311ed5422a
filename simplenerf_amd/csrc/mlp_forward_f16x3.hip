// Entry point of the split-precision format (SNERF_PRECISION_F16X3: fp16 hi/lo pairs, three MFMAs per product) of the fused
// PE + MLP forward (kernel: mlp_forward_half_kernel.h).
#include "mlp_forward_half_kernel.h"

namespace snerf {

int mlp_forward_f16_split(const MlpPlan& plan, const MlpArgs& m, bool train, hipStream_t stream) {
    return dispatch_half<3>(plan, m, train, plan.half_offset, stream);
}

}  // namespace snerf
