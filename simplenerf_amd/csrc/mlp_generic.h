// The layered MLP path (mlp_generic.hip): plan + entry points used by the C-ABI functions when build_plan (mlp_plan.h)
// reports a shape the fused kernels are not built for.
#pragma once
#include <vector>

#include "snerf_common.h"

namespace snerf {

struct GenericPlan {
    int depth = 0, width = 0, views_depth = 0, views_width = 0;
    bool view_dep = false;
    int points_degree = 0, views_degree = 0;
    int pe_full = 0, pts_in = 0, extra = 0, views_pe = 0, views_in = 0, pts_out_rows = 1;
    int num_params = 0;
    std::vector<long long> w_off, w_count;   // parameter i in the "packed" buffer (ABI order, plain copies)
    long long packed_floats = 0;
    // activation row (floats per sample) and the column of every block in it
    long long row = 0;
    int c_pe = 0, c_pev = 0, c_x5 = -1, c_v0 = 0, c_out = 0, c_vout = 0;
    std::vector<int> c_h, c_hv;
    bool skip_layer(int l) const { return l == 5 && depth > 5; }   // its input is [encoding | H_4] (:580, :662-663)
    int layer_in_dim(int l) const { return l == 0 ? pts_in : (skip_layer(l) ? pts_in + width : width); }
    int layer_in_col(int l) const { return l == 0 ? c_pe : (skip_layer(l) ? c_x5 : c_h[l - 1]); }
};

int generic_plan(const snerf_mlp_desc* desc, GenericPlan* out);
int generic_pack(const GenericPlan& p, const float* const* params, float* packed, hipStream_t stream);
size_t generic_saved_floats(const GenericPlan& p, long long total);
int generic_forward(const GenericPlan& p, const float* packed, const float* origins, const float* dirs, const float* view_dirs,
                    const float* depths, long long num_rays, int num_samples, const float* noise, float* sigma, float* rgb,
                    float* saved_acts, int precision, hipStream_t stream);
size_t generic_backward_workspace_floats(const GenericPlan& p, long long total);
int generic_backward(const GenericPlan& p, const float* packed, const float* acts, const float* sigma, const float* rgb,
                     const float* d_sigma, const float* d_rgb, long long total, float* workspace, float* const* grads, int precision,
                     int accumulate, hipStream_t stream);

// build_plan said "unsupported": is it a shape the layered path takes?  (fills *plan when so)
inline bool generic_takes(const snerf_mlp_desc* desc, GenericPlan* plan) { return generic_plan(desc, plan) == SNERF_OK; }

}  // namespace snerf
