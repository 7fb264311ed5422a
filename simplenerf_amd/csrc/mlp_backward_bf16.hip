// Entry point of the bf16 single-product format (SNERF_PRECISION_BF16) of the backward chain (kernel:
// mlp_backward_half_kernel.h; a.act_rows / a.grad_rows describe 16-bit rows).
#include "mlp_backward_half_kernel.h"

namespace snerf {

int mlp_backward_chain_bf16(const MlpPlan& plan, const ChainArgs& a, hipStream_t stream) {
    return dispatch_chain<2>(plan, a, plan.bf_dgrad_offset, stream);
}

}  // namespace snerf
