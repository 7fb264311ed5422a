// Entry point of the bf16 single-product format (SNERF_PRECISION_BF16) of the fused PE + MLP forward: the kernels of
// mlp_forward_half_kernel.h with bf16 operand conversion, v_mfma_f32_32x32x16_bf16 and the compact bf16 unit stream.
#include "mlp_forward_half_kernel.h"

namespace snerf {

int mlp_forward_bf16(const MlpPlan& plan, const MlpArgs& m, bool train, hipStream_t stream) {
    return dispatch_half<2>(plan, m, train, plan.bf_offset, stream);
}

}  // namespace snerf
