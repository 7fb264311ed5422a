// TORCH_LIBRARY binding of the one-call render ops -- the PyTorch-side boundary SURVEY 8b ("Ownership / errors") names:
// ops registered with the dispatcher (torch.ops.snerf.*), wrapped in a C++ torch::autograd::Function, taking raw device
// pointers on the CURRENT HIP stream of the tensors' device, allocating their outputs through torch's allocator and
// throwing c10::Error (a Python RuntimeError) on a shape / dtype / device mismatch.
//
// This file contains no arithmetic and no device code: it is host glue over the C ABI (include/simplenerf_hip.h, built
// into libsimplenerf_hip.so), the same two calls the ctypes binding makes (simplenerf_amd/ops.py RenderCall) --
//   snerf::render          = snerf_render_forward  (the whole of SimpleNeRF.render_rays, src/models/SimpleNeRF01.py:108-270)
//   its autograd backward  = snerf_render_backward (what autograd replays for it)
// -- minus ~60 Python-level tensor views, pointer-struct fills and two Python autograd hand-offs per training forward.  The
// ctypes path remains the torch-free binding of the same ABI.
//
// Built by torch.utils.cpp_extension (simplenerf_amd/build.py: build_torch_extension) with the host compiler only.
#include <ATen/ATen.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <c10/util/accumulate.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/custom_class.h>
#include <torch/library.h>

#include <memory>
#include <string>
#include <vector>

#include "simplenerf_hip.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

constexpr int kLevels = SNERF_RENDER_LEVELS;
constexpr int kDescInts = 10;   // fields of snerf_mlp_desc, in declaration order
// cfg ints: ndc, white_bkgd, lindisp, num_coarse, num_fine, precision, per-sample mask (1 alpha | 2 visibility | 4 weights),
// fused (eval-mode renders of a plain coarse + fine model as one launch, snerf_render_config::fused)
constexpr int kCfgInts = 8;
// rays: rays_o, rays_d, view_dirs, rays_o_ndc, rays_d_ndc, near, far, rays_o2
enum { R_O, R_D, R_VIEW, R_O_NDC, R_D_NDC, R_NEAR, R_FAR, R_O2, kRays };
// draws: t_rand, u, sigma noise of the six levels, fine-depth override
enum { D_TRAND, D_U, D_NOISE0, D_ZFINE = D_NOISE0 + kLevels, kDraws };

void check_status(int status, const char* what) {
    if (status == SNERF_OK) return;
    const char* msg = snerf_last_error();
    TORCH_CHECK(false, what, " failed (", status, "): ", msg && msg[0] ? msg : "unknown error");
}

// A validated device operand: contiguous fp32 of the expected shape on the GPU (copying only if the caller's is strided).
Tensor operand(const std::optional<Tensor>& t, const char* name, at::IntArrayRef shape) {
    if (!t.has_value() || !t->defined()) return Tensor();
    TORCH_CHECK(t->is_cuda(), name, ": expected a tensor on the GPU (the HIP renderer has no CPU path), got one on ", t->device());
    TORCH_CHECK(t->scalar_type() == at::kFloat, name, ": expected float32, got ", t->scalar_type());
    if (t->sizes() != shape) {   // (n,1) columns and (n,S,1) noise are taken as (n) / (n,S): same memory
        TORCH_CHECK(t->numel() == c10::multiply_integers(shape) && t->dim() == (int64_t)shape.size() + 1 && t->size(-1) == 1,
                    name, ": expected shape ", shape, ", got ", t->sizes());
        return t->detach().reshape(shape).contiguous();
    }
    return t->detach().contiguous();
}
const float* ptr(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
float* mptr(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }

snerf_mlp_desc desc_from(const int64_t* p) {
    snerf_mlp_desc d;
    d.points_net_depth = (int)p[0]; d.points_net_width = (int)p[1]; d.views_net_depth = (int)p[2]; d.views_net_width = (int)p[3];
    d.points_pe_degree = (int)p[4]; d.views_pe_degree = (int)p[5]; d.sigma_pe_degree = (int)p[6]; d.use_view_dirs = (int)p[7];
    d.view_dependent_rgb = (int)p[8]; d.predict_visibility = (int)p[9];
    return d;
}

// Everything one render call owns: configuration, validated inputs, output pools and the C structs that point into them.
// Shared between the forward and the autograd node (which keeps it alive for the backward).  It holds the POOLS the outputs
// are carved from, never the output views themselves: those carry the node that owns this object, and holding them would
// close a reference cycle that keeps GBs of saved activations alive.
struct RenderCall {
    snerf_render_config cfg{};
    snerf_mlp_desc descs[kLevels]{};
    snerf_render_mlp mlps[kLevels]{};
    bool present[kLevels]{};
    snerf_render_rays rays{};
    snerf_render_outputs out{};
    long long n = 0;
    int per_sample = 0;
    int num_other = 0;
    std::vector<Tensor> held;          // inputs, pools, saved activations: the memory behind the pointers above
    std::vector<int> num_params;       // per level
    struct DiffOutput { int index, level, key; };
    std::vector<DiffOutput> diff_outputs;          // position, level and key of every differentiable output
    bool return_param_grads = false;
    c10::DeviceIndex device = 0;

    int samples(int level) const { return level < 3 ? cfg.num_coarse : cfg.num_coarse + cfg.num_fine; }
};

enum Key { K_RGB, K_ACC, K_DEPTH, K_DEPTH_VAR, K_DEPTH_NDC, K_DEPTH_VAR_NDC, K_ALPHA, K_VISIBILITY, K_WEIGHTS, K_SIGMA, K_RAW_RGB,
           K_RAW_VIS, K_RAW_VIS2, K_VIS2, kKeys };
bool differentiable(int key) { return key == K_RGB || key == K_ACC || key == K_DEPTH || key == K_DEPTH_NDC || key == K_SIGMA || key == K_RAW_RGB; }

struct Pool {
    Tensor buffer;
    int64_t used = 0;
    Tensor carve(at::IntArrayRef shape) {
        int64_t size = 1;
        std::vector<int64_t> strides(shape.size());
        for (int i = (int)shape.size() - 1; i >= 0; --i) { strides[i] = size; size *= shape[i]; }
        Tensor view = buffer.as_strided(shape, strides, used);
        used += size;
        return view;
    }
};

// The forward proper: validates, allocates, fills the C structs, enqueues snerf_render_forward on the current stream.
// Returns the output tensors: z_vals_coarse, z_vals_fine (when the model has a fine pass), then for every present level, in
// level order, the defined keys in `Key` order.  `layout` receives (level, key) of every per-level output.
std::vector<Tensor> run_forward(RenderCall& c, at::IntArrayRef cfg, at::IntArrayRef descs, const c10::List<std::optional<Tensor>>& packed,
                                const c10::List<std::optional<Tensor>>& rays, const c10::List<std::optional<Tensor>>& draws,
                                bool keep_activations, std::vector<std::pair<int, int>>* layout) {
    TORCH_CHECK((int)cfg.size() == kCfgInts, "snerf::render: cfg holds ", kCfgInts, " ints, got ", cfg.size());
    TORCH_CHECK((int)descs.size() == kLevels * kDescInts, "snerf::render: descs holds ", kLevels * kDescInts, " ints, got ", descs.size());
    TORCH_CHECK((int)packed.size() == kLevels && (int)rays.size() == kRays && (int)draws.size() == kDraws,
                "snerf::render: packed / rays / draws hold ", kLevels, " / ", (int)kRays, " / ", (int)kDraws, " entries");
    c.cfg.ndc = (int)cfg[0]; c.cfg.white_bkgd = (int)cfg[1]; c.cfg.lindisp = (int)cfg[2]; c.cfg.num_coarse = (int)cfg[3];
    c.cfg.num_fine = (int)cfg[4]; c.cfg.precision = (int)cfg[5]; c.cfg.keep_activations = keep_activations ? 1 : 0;
    c.per_sample = (int)cfg[6];
    c.cfg.fused = (int)cfg[7] && !keep_activations ? 1 : 0;
    const bool ndc = c.cfg.ndc != 0;
    const std::optional<Tensor> rays_o = rays.get(R_O);
    TORCH_CHECK(rays_o.has_value() && rays_o->defined() && rays_o->dim() == 2, "snerf::render: rays_o (n,3) is required");
    TORCH_CHECK(rays_o->is_cuda(), "rays_o: expected a tensor on the GPU (the HIP renderer has no CPU path), got one on ", rays_o->device());
    const int64_t n = rays_o->size(0);
    c.n = n;
    c.device = rays_o->device().index();
    const auto options = rays_o->options().dtype(at::kFloat).requires_grad(false);
    auto hold = [&](Tensor t) { if (t.defined()) c.held.push_back(t); return t; };

    for (int l = 0; l < kLevels; ++l) {
        c.descs[l] = desc_from(descs.data() + l * kDescInts);
        c.present[l] = c.descs[l].points_net_depth > 0;
        c.mlps[l].desc = c.present[l] ? &c.descs[l] : nullptr;
        c.mlps[l].packed = nullptr;
        if (c.present[l]) {
            const std::optional<Tensor> p = packed.get(l);
            TORCH_CHECK(p.has_value() && p->defined() && p->is_cuda() && p->scalar_type() == at::kFloat && p->is_contiguous(),
                        "snerf::render: level ", l, " needs its packed weight stream (contiguous float32 on the GPU)");
            TORCH_CHECK((size_t)p->numel() >= snerf_mlp_packed_floats(&c.descs[l]), "snerf::render: level ", l, ": packed stream too short");
            c.mlps[l].packed = hold(*p).data_ptr<float>();
        }
    }
    TORCH_CHECK(c.present[0], "snerf::render: the main coarse MLP is required");
    if (!c.present[3]) c.cfg.num_fine = 0;
    const int s_c = c.cfg.num_coarse, s_f = c.cfg.num_fine;

    snerf_render_rays& r = c.rays;
    // (presence is checked on the tensors, not the pointers: a zero-ray call has tensors without storage)
    auto given = [&](int index) { const std::optional<Tensor> t = rays.get(index); return t.has_value() && t->defined(); };
    r.rays_o = ptr(hold(operand(rays.get(R_O), "rays_o", {n, 3})));
    r.rays_d = ptr(hold(operand(rays.get(R_D), "rays_d", {n, 3})));
    TORCH_CHECK(given(R_D), "snerf::render: rays_d is required");
    bool need_dirs = false, predicts = false;
    for (int l = 0; l < kLevels; ++l) {
        need_dirs = need_dirs || (c.present[l] && c.descs[l].use_view_dirs);
        predicts = predicts || (c.present[l] && c.descs[l].predict_visibility);
    }
    r.view_dirs = need_dirs ? ptr(hold(operand(rays.get(R_VIEW), "view_dirs", {n, 3}))) : nullptr;
    TORCH_CHECK(!need_dirs || given(R_VIEW), "snerf::render: view_dirs is required (an MLP uses view directions)");
    r.rays_o_ndc = ndc ? ptr(hold(operand(rays.get(R_O_NDC), "rays_o_ndc", {n, 3}))) : nullptr;
    r.rays_d_ndc = ndc ? ptr(hold(operand(rays.get(R_D_NDC), "rays_d_ndc", {n, 3}))) : nullptr;
    TORCH_CHECK(!ndc || (given(R_O_NDC) && given(R_D_NDC)), "snerf::render: rays_o_ndc / rays_d_ndc are required when ndc");
    r.near = ptr(hold(operand(rays.get(R_NEAR), "near", {n})));
    r.far = ptr(hold(operand(rays.get(R_FAR), "far", {n})));
    TORCH_CHECK(given(R_NEAR) && given(R_FAR), "snerf::render: near / far are required");
    r.t_rand = ptr(hold(operand(draws.get(D_TRAND), "t_rand", {n, s_c})));
    r.u = s_f ? ptr(hold(operand(draws.get(D_U), "u", {n, s_f}))) : nullptr;
    for (int l = 0; l < kLevels; ++l)
        r.sigma_noise[l] = c.present[l] ? ptr(hold(operand(draws.get(D_NOISE0 + l), "sigma_noise", {n, c.samples(l)}))) : nullptr;
    Tensor fine_in = s_f ? operand(draws.get(D_ZFINE), "z_vals_fine", {n, s_c + s_f}) : Tensor();
    r.depths_fine = ptr(hold(fine_in));
    Tensor rays_o2;
    if (predicts) {
        const std::optional<Tensor> o2 = rays.get(R_O2);
        if (o2.has_value() && o2->defined()) {
            TORCH_CHECK(o2->dim() == 3, "rays_o2: expected (n, K, 3)");
            rays_o2 = operand(o2, "rays_o2", {n, o2->size(1), 3});
        }
    }
    const int k_other = rays_o2.defined() ? (int)rays_o2.size(1) : 0;
    r.rays_o2 = k_other ? ptr(hold(rays_o2)) : nullptr;
    r.num_other = k_other;
    c.num_other = k_other;

    // ---- outputs: one allocation per group (per-ray, per-sample), views carved out of them
    const bool want[3] = {(c.per_sample & 1) != 0, (c.per_sample & 2) != 0, (c.per_sample & 4) != 0};   // alpha, visibility, weights
    const int per_ray_floats = 3 + 1 + 1 + 1 + (ndc ? 2 : 0);
    int64_t ray_floats = 0, sample_floats = n * s_c + ((s_f && !fine_in.defined()) ? n * (s_c + s_f) : 0);
    for (int l = 0; l < kLevels; ++l) {
        if (!c.present[l]) continue;
        const int64_t s = c.samples(l);
        ray_floats += per_ray_floats * n;
        sample_floats += n * s * (4 + (int)want[0] + (int)want[1] + (int)want[2]);
        if (c.descs[l].predict_visibility) {
            sample_floats += n * s * (1 + 4 * k_other + ((want[2] || !k_other) ? 0 : 1));
            ray_floats += n * k_other;
        }
    }
    Pool small{at::empty({ray_floats}, options)}, big{at::empty({sample_floats}, options)};
    std::vector<Tensor> outputs;
    snerf_render_outputs& o = c.out;
    Tensor z_coarse = big.carve({n, s_c});
    o.depths_coarse = mptr(z_coarse);
    outputs.push_back(z_coarse);
    o.depths_fine = nullptr;
    if (s_f) {
        Tensor z_fine = fine_in.defined() ? fine_in : big.carve({n, s_c + s_f});
        if (!fine_in.defined()) o.depths_fine = mptr(z_fine);
        outputs.push_back(z_fine);
    }
    for (int l = 0; l < kLevels; ++l) {
        snerf_render_level_out& lo = o.level[l];
        lo = snerf_render_level_out{};
        if (!c.present[l]) continue;
        const int64_t s = c.samples(l);
        auto emit = [&](int key, const Tensor& t) { outputs.push_back(t); layout->emplace_back(l, key); return mptr(t); };
        lo.rgb = emit(K_RGB, small.carve({n, 3}));
        lo.acc = emit(K_ACC, small.carve({n}));
        lo.depth = emit(K_DEPTH, small.carve({n}));
        lo.depth_var = emit(K_DEPTH_VAR, small.carve({n}));
        if (ndc) {
            lo.depth_ndc = emit(K_DEPTH_NDC, small.carve({n}));
            lo.depth_var_ndc = emit(K_DEPTH_VAR_NDC, small.carve({n}));
        }
        if (want[0]) lo.alpha = emit(K_ALPHA, big.carve({n, s}));
        if (want[1]) lo.visibility = emit(K_VISIBILITY, big.carve({n, s}));
        if (want[2]) lo.weights = emit(K_WEIGHTS, big.carve({n, s}));
        lo.sigma = emit(K_SIGMA, big.carve({n, s, 1}));
        lo.raw_rgb = emit(K_RAW_RGB, big.carve({n, s, 3}));
        if (c.descs[l].predict_visibility) {
            lo.raw_visibility = emit(K_RAW_VIS, big.carve({n, s, 1}));
            if (k_other) {
                lo.raw_visibility2 = emit(K_RAW_VIS2, big.carve({n, s, k_other, 1}));
                lo.visibility2 = emit(K_VIS2, small.carve({n, k_other}));
                lo.view_dirs2 = mptr(big.carve({n, s, k_other, 3}));
                if (!lo.weights) lo.weights = mptr(big.carve({n, s}));   // visibility2 is composited with the level's weights
            }
        }
        if (keep_activations) {
            const int64_t floats = (int64_t)snerf_mlp_saved_floats(&c.descs[l], n, (int)s);
            lo.saved_acts = mptr(hold(at::empty({floats}, options)));
        }
    }
    TORCH_INTERNAL_ASSERT(small.used == ray_floats && big.used == sample_floats, "snerf::render: output pools mis-sized");
    hold(small.buffer);
    hold(big.buffer);
    Tensor work;
    if (s_f && !fine_in.defined() && !want[2])
        work = hold(at::empty({(int64_t)snerf_render_workspace_floats(&c.cfg, n)}, options));
    if (n > 0) {
        const c10::hip::HIPGuard guard(c.device);
        hipStream_t stream = c10::hip::getCurrentHIPStream(c.device).stream();
        check_status(snerf_render_forward(&c.cfg, c.mlps, &c.rays, n, &c.out, mptr(work), stream), "snerf_render_forward");
    }
    return outputs;
}

// The whole of render_rays as ONE autograd node.  Parameter gradients are written by the kernels straight into `p.grad`
// -- overwritten when the parameter has none yet, ADDED to it otherwise (the trainer's second sub-batch,
// src/Trainer01.py:82-96) -- unless `return_param_grads`, which restores the plain autograd contract (torch.autograd.grad).
struct RenderFunction : public torch::autograd::Function<RenderFunction> {
    struct Capsule;
    static variable_list forward(AutogradContext* ctx, at::TensorList params, std::shared_ptr<RenderCall> call,
                                 std::vector<int64_t> cfg, std::vector<int64_t> descs, c10::List<std::optional<Tensor>> packed,
                                 c10::List<std::optional<Tensor>> rays, c10::List<std::optional<Tensor>> draws, bool keep) {
        std::vector<std::pair<int, int>> layout;
        std::vector<Tensor> outputs = run_forward(*call, cfg, descs, packed, rays, draws, keep, &layout);
        const size_t first_level_output = outputs.size() - layout.size();
        variable_list plain;
        for (size_t i = 0; i < outputs.size(); ++i) {
            if (i >= first_level_output && differentiable(layout[i - first_level_output].second))
                call->diff_outputs.push_back({(int)i, layout[i - first_level_output].first, layout[i - first_level_output].second});
            else
                plain.push_back(outputs[i]);
        }
        ctx->mark_non_differentiable(plain);
        ctx->set_materialize_grads(false);
        ctx->saved_data["params"] = c10::IValue(c10::List<Tensor>(params.vec()));
        auto capsule = c10::make_intrusive<Capsule>();
        capsule->call = call;
        ctx->saved_data["call"] = c10::IValue(std::move(capsule));
        return outputs;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grad_outputs) {
        c10::intrusive_ptr<Capsule> capsule = ctx->saved_data["call"].toCustomClass<Capsule>();
        std::shared_ptr<RenderCall> call = capsule->call;
        TORCH_CHECK(call, "snerf::render: backward called twice (the saved activations were released)");
        const std::vector<Tensor> params = ctx->saved_data["params"].toTensorVector();
        RenderCall& c = *call;
        const int64_t n = c.n;
        snerf_render_level_grads grads[kLevels];
        for (auto& g : grads) g = snerf_render_level_grads{};
        std::vector<Tensor> keep;
        bool any[kLevels] = {};
        for (const RenderCall::DiffOutput& d : c.diff_outputs) {
            const Tensor& g = grad_outputs[d.index];
            if (!g.defined()) continue;
            const int l = d.level, key = d.key;
            const int64_t s = c.samples(l);
            std::vector<int64_t> shape = key == K_RGB ? std::vector<int64_t>{n, 3} : key == K_SIGMA ? std::vector<int64_t>{n, s, 1}
                                       : key == K_RAW_RGB ? std::vector<int64_t>{n, s, 3} : std::vector<int64_t>{n};
            TORCH_CHECK(g.is_cuda() && g.scalar_type() == at::kFloat && g.numel() == c10::multiply_integers(shape),
                        "snerf::render backward: unexpected gradient for output ", key, " of level ", l);
            Tensor t = g.reshape(shape).contiguous();
            keep.push_back(t);
            const float* p = t.data_ptr<float>();
            switch (key) {
                case K_RGB: grads[l].rgb = p; break;
                case K_ACC: grads[l].acc = p; break;
                case K_DEPTH: grads[l].depth = p; break;
                case K_DEPTH_NDC: grads[l].depth_ndc = p; break;
                case K_SIGMA: grads[l].sigma = p; break;
                default: grads[l].raw_rgb = p; break;
            }
            any[l] = true;
        }
        // parameter gradients: p.grad itself (accumulate when it exists) or fresh tensors handed back to autograd
        variable_list returned(params.size() + 7);        // one slot per forward argument; the seven non-tensor ones stay undefined
        std::vector<std::vector<float*>> tables(kLevels);
        std::vector<std::pair<Tensor, Tensor>> fresh;     // (parameter, new gradient) to install after the call
        size_t at = 0;
        for (int l = 0; l < kLevels; ++l) {
            if (!c.present[l]) continue;
            const int count = c.num_params[l];
            bool trainable = false;
            for (int i = 0; i < count; ++i) trainable = trainable || params[at + i].requires_grad();
            if (!any[l] || !trainable) { at += count; continue; }
            bool all_have = !c.return_param_grads;
            for (int i = 0; i < count && all_have; ++i) all_have = params[at + i].grad().defined();
            for (int i = 0; i < count; ++i) {
                const Tensor& p = params[at + i];
                Tensor g;
                if (all_have) {
                    g = p.grad();
                    TORCH_CHECK(g.is_contiguous() && g.scalar_type() == at::kFloat && g.sizes() == p.sizes(),
                                "snerf::render backward: .grad of a parameter is not a contiguous float32 tensor of its shape");
                } else {
                    g = at::empty_like(p, p.options().requires_grad(false), at::MemoryFormat::Contiguous);
                    if (c.return_param_grads) returned[at + i] = g;
                    else fresh.emplace_back(p, g);
                }
                keep.push_back(g);
                tables[l].push_back(g.data_ptr<float>());
            }
            grads[l].param_grads = tables[l].data();
            grads[l].num_params = count;
            grads[l].accumulate = all_have ? 1 : 0;
            at += count;
        }
        if (n > 0) {
            const c10::hip::HIPGuard guard(c.device);
            hipStream_t stream = c10::hip::getCurrentHIPStream(c.device).stream();
            const auto options = c.held.front().options().dtype(at::kFloat).requires_grad(false);
            Tensor work = at::empty({(int64_t)snerf_render_backward_workspace_floats(&c.cfg, c.mlps, n)}, options);
            check_status(snerf_render_backward(&c.cfg, c.mlps, &c.rays, n, &c.out, grads, work.data_ptr<float>(), stream),
                         "snerf_render_backward");
        }
        for (auto& pg : fresh) {     // direct mode, first sub-batch: the new tensor becomes (or is added to) p.grad
            if (!pg.first.requires_grad()) continue;
            Tensor& slot = pg.first.mutable_grad();
            if (slot.defined()) slot.add_(pg.second); else slot = pg.second;
        }
        capsule->call.reset();       // the saved activations are released with the node's first backward
        return returned;
    }

    // the call object travels in the context's saved_data and dies with the node
    struct Capsule : torch::CustomClassHolder {
        std::shared_ptr<RenderCall> call;
    };
};

// snerf::render(cfg, descs, packed, rays, draws, params, num_params, keep_activations, return_param_grads) -> Tensor[]
// Output order: z_vals_coarse, z_vals_fine (if the model has a fine pass), then for every present level, in level order, the
// keys it produces in the order of `Key`.  rgb, acc, depth, depth_ndc, sigma, raw_rgb carry the autograd node.
std::vector<Tensor> render(at::IntArrayRef cfg, at::IntArrayRef descs, const c10::List<std::optional<Tensor>>& packed,
                           const c10::List<std::optional<Tensor>>& rays, const c10::List<std::optional<Tensor>>& draws,
                           at::TensorList params, at::IntArrayRef num_params, bool keep_activations, bool return_param_grads) {
    auto call = std::make_shared<RenderCall>();
    TORCH_CHECK((int)num_params.size() == kLevels, "snerf::render: num_params holds one count per level");
    int64_t total = 0;
    for (int l = 0; l < kLevels; ++l) { call->num_params.push_back((int)num_params[l]); total += num_params[l]; }
    TORCH_CHECK(!keep_activations || total == (int64_t)params.size(), "snerf::render: ", params.size(), " parameters for counts summing to ", total);
    call->return_param_grads = return_param_grads;
    return RenderFunction::apply(params, call, cfg.vec(), descs.vec(), packed, rays, draws, keep_activations);
}

}  // namespace

TORCH_LIBRARY(snerf, m) {
    m.class_<RenderFunction::Capsule>("RenderCallCapsule");
    m.def("render(int[] cfg, int[] descs, Tensor?[] packed, Tensor?[] rays, Tensor?[] draws, Tensor[] params, int[] num_params, "
          "bool keep_activations, bool return_param_grads) -> Tensor[]");
    m.def("abi_version() -> int", []() -> int64_t { return snerf_abi_version(); });
}
// One kernel for the autograd AND the backend key: the function builds its own graph node when gradients are recorded and
// is a plain forward otherwise (no_grad, inference mode); it never re-dispatches.  (ROCm tensors carry the CUDA keys.)
TORCH_LIBRARY_IMPL(snerf, Autograd, m) { m.impl("render", render); }
TORCH_LIBRARY_IMPL(snerf, CUDA, m) { m.impl("render", render); }
