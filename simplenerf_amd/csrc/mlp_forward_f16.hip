// Entry point of the fp16 formats of the fused PE + MLP forward (kernel: mlp_forward_half_kernel.h).
#include "mlp_forward_half_kernel.h"

namespace snerf {

// Called by snerf_mlp_forward for SNERF_PRECISION_F16X3 (products = 3) and SNERF_PRECISION_F16 (products = 1).
int mlp_forward_f16x3(const MlpPlan& plan, const MlpArgs& m, bool train, int products, hipStream_t stream) {
    return products == 3 ? dispatch_half<3>(plan, m, train, plan.half_offset, stream)
                         : dispatch_half<1>(plan, m, train, plan.half_offset, stream);
}

}  // namespace snerf
