// Entry point of the fp16 single-product format (SNERF_PRECISION_F16) of the fused PE + MLP forward (kernel:
// mlp_forward_half_kernel.h).  The split-precision format lives in mlp_forward_f16x3.hip, the bf16 one in mlp_forward_bf16.hip:
// three translation units, so that the template instantiations compile side by side.
#include "mlp_forward_half_kernel.h"

namespace snerf {

int mlp_forward_f16_single(const MlpPlan& plan, const MlpArgs& m, bool train, hipStream_t stream) {
    return dispatch_half<1>(plan, m, train, plan.half_offset, stream);
}

}  // namespace snerf
