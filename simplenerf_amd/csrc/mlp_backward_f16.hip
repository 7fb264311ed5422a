// Entry point of the fp16 formats of the backward chain (kernel: mlp_backward_half_kernel.h).
#include "mlp_backward_half_kernel.h"

namespace snerf {

// products = 3: SNERF_PRECISION_F16X3.  products = 1: SNERF_PRECISION_F16 (a.act_rows / a.grad_rows describe 16-bit rows).
int mlp_backward_chain_f16x3(const MlpPlan& plan, const ChainArgs& a, int products, hipStream_t stream) {
    return products == 3 ? dispatch_chain<3>(plan, a, plan.half_dgrad_offset, stream)
                         : dispatch_chain<1>(plan, a, plan.half_dgrad_offset, stream);
}

}  // namespace snerf
