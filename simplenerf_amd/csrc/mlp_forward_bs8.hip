// Entry point of the training forward of SNERF_PRECISION_BF16S8 (kernel: mlp_forward_half_kernel.h, BF + S8): the bf16
// single-product forward that saves the trunk activations h_1 .. h_D-1 as fp8 e4m3 tiles.
#include "mlp_forward_half_kernel.h"

namespace snerf {

int mlp_forward_bs8_train(const MlpPlan& plan, const MlpArgs& m, hipStream_t stream) {
    return dispatch_half<5>(plan, m, true, plan.bf_offset, stream);
}

}  // namespace snerf
