// Is a 64-SAMPLE wave tile worth building for the 16-bit inference kernel?  A skeleton of the kernel's trunk in
// both shapes, with the REAL register footprint, data flow and per-layer epilogue, and nothing else (no encoding, no heads, no
// weight DMA: one layer's 128 KiB of fp16 fragments sit in LDS and every layer re-reads them -- an upper bound for both):
//
//   shape 32 (what ships, csrc/mlp_forward_m16.hip): 8 waves per workgroup = two per SIMD, 32 samples per wave; a 1-KiB
//            fragment (16 out rows x 32 k) feeds TWO v_mfma_f32_16x16x32_f16; the wave keeps all 8 out tiles of a layer as
//            accumulators (128 registers) next to the layer's input operands (64) and converts them after the layer's last tile.
//   shape 32p: shape 32 with the proposal's layer structure (below) at 32 samples per wave -- is it the tile or the pipelining?
//   shape 64 (the proposal): 4 waves per workgroup = ONE per SIMD, 64 samples per wave; a fragment feeds FOUR MFMAs -- half
//            the LDS bytes per MFMA; input operands 128 registers + the next layer's 128 (a tile is converted while the next
//            tile's MFMAs issue: with one wave per SIMD nothing else can hide the epilogue) + two tiles of accumulators (64).
//
// LDS arithmetic per CU and 256 x 256 layer: shape 32: 8 waves x 128 fragments x 1 KiB = 1 MiB read per 8 x 256 MFMAs x 16
// cycles / 4 SIMDs = 8192 cycles -> 128 B/clk/CU of the 256 B/clk the LDS delivers to ds_read_b128 (MI355X_MICROARCH.md, LDS).
// shape 64: 4 waves x 128 KiB per 4 x 512 x 16 / 4 = 8192 cycles -> 64 B/clk/CU.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/m64_skeleton tools/probes/m64_skeleton.hip && /tmp/m64_skeleton [seconds]
// Prints hardware TFLOP/s (2 x MACs issued) of both skeletons, alternated, and the registers each kernel was given.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kLayers = 16;          // 256 -> 256 layers per pass (the real trunk has 8 + feature + views ~ 10)
constexpr int kPasses = 8;           // passes per launch (a persistent loop: no launch ramp in the figure)
constexpr int kLayerFrags = 128;     // 8 out tiles x 8 k-blocks x 2 row halves, 1 KiB each

// fragment reads as the kernels issue them: inline asm with counted waits (compiler-visible reads get hoisted by the dozen and
// the kernel spills: first build of this probe, 581 spilled registers) -- mlp_device_f16.h lds_read_f16x8 / mlp_forward_m16.hip
__device__ __forceinline__ f16x8 lds_read_f16x8(unsigned lds_byte_address, int byte_offset) {
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_byte_address), "i"(byte_offset) : "memory");
    return v;
}
__device__ __forceinline__ void lds_pair_landed(f16x8& a, f16x8& b, int newer) {   // `newer` folds to a constant
    if (newer >= 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a), "+v"(b)::"memory");
    else if (newer == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a), "+v"(b)::"memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)::"memory");
}

__device__ __forceinline__ void to_operand(const f32x4& r0, const f32x4& r1, f16x8& h) {   // relu + fp16, as tile_to_operand16
    const f16x2 zero = {(_Float16)0.0f, (_Float16)0.0f};
#pragma unroll
    for (int q = 0; q < 4; q += 2) {
        f16x2 a = __builtin_convertvector((f32x2){r0[q], r0[q + 1]}, f16x2);
        f16x2 b = __builtin_convertvector((f32x2){r1[q], r1[q + 1]}, f16x2);
        a = __builtin_elementwise_max(a, zero);
        b = __builtin_elementwise_max(b, zero);
        h[q] = a[0]; h[q + 1] = a[1]; h[4 + q] = b[0]; h[4 + q + 1] = b[1];
    }
}

// one 32-row out tile over the 8 k-blocks of a 256-wide input: fragment pairs requested two k-blocks ahead (seg1_m16);
// `side(c)` runs behind k-block c's MFMAs
template <int NS, class Side>
__device__ __forceinline__ void tile_kloop(unsigned base, f32x4 (&acc)[2][NS], const f16x8 (&in)[8][NS], Side&& side) {
    f16x8 a[3][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 2; ++r) a[c][r] = lds_read_f16x8(base, (2 * c + r) * 1024);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (c + 2 < 8) {
#pragma unroll
            for (int r = 0; r < 2; ++r) a[(c + 2) % 3][r] = lds_read_f16x8(base, (2 * (c + 2) + r) * 1024);
        }
        const int newer = 2 * ((7 - c) < 2 ? (7 - c) : 2);
        lds_pair_landed(a[c % 3][0], a[c % 3][1], newer);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[r][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c % 3][r], in[c][s], acc[r][s], 0, 0, 0);
        side(c);
    }
}

constexpr int kBiasBytes = 1024;     // 256 floats behind the fragments

// ------------------------------------------------------------------------------------------------ shape 32 (as shipped)
__global__ void __launch_bounds__(512, 2) skeleton32(const f16x8* __restrict__ weights, const f16x8* __restrict__ operands,
                                                     const float* __restrict__ bias_g, float* out) {
    extern __shared__ __attribute__((aligned(16))) f16x8 lds[];
    const int lane = threadIdx.x & 63, grp = lane >> 4;
    for (int i = threadIdx.x; i < kLayerFrags * 64; i += blockDim.x) lds[i] = weights[i];
    float* bias = reinterpret_cast<float*>(lds + kLayerFrags * 64);
    if (threadIdx.x < 256) bias[threadIdx.x] = bias_g[threadIdx.x];
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)lds + lane * 16;
    f16x8 xh[8][2];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int s = 0; s < 2; ++s) xh[c][s] = operands[((c * 2 + s) * 64 + lane) % (16 * 64)];
    f32x4 acc[8][2][2];
    for (int pass = 0; pass < kPasses; ++pass) {
        for (int layer = 0; layer < kLayers; ++layer) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + 32 * u + 4 * grp);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + 32 * u + 16 + 4 * grp);
                acc[u][0][0] = b0; acc[u][0][1] = b0; acc[u][1][0] = b1; acc[u][1][1] = b1;
                tile_kloop<2>(lds0 + u * 16 * 1024, acc[u], xh, [](int) {});
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int s = 0; s < 2; ++s) to_operand(acc[u][0][s], acc[u][1][s], xh[u][s]);
        }
    }
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int s = 0; s < 2; ++s) sum += (float)xh[c][s][0] + (float)xh[c][s][5];
    if (sum == 12345.678f) out[0] = sum;
}

// ------------------------------------------------------------------------------------------------ shape 32 + the weight stream's DMA
// What does ISSUING the weight stream cost (the real kernel without its LDS-DMA is 14 % faster, r05_m16_ablation.txt)?  The shipped
// shape with, per tile (= staging unit: 16 KiB for the 8 waves), two 1-KiB loads per wave of a 1.2 MB global stream:
//   MODE 1: global_load_lds_dwordx4 into a 16-KiB area beside the resident weights (never read: the timing of the DMA alone)
//   MODE 2: the same bytes by global_load_dwordx4 into registers (kept alive by an empty asm, never used): vector-memory issue and
//           L2 -> CU traffic without the LDS write
//   MODE 4: MODE 1 through buffer_load_dwordx4 ... lds
//   MODE 3: MODE 1 + the counted wait and the workgroup barrier of the ring's hand-over at every tile
template <int MODE>
__global__ void __launch_bounds__(512, 2) skeleton32_dma(const f16x8* __restrict__ weights, const f16x8* __restrict__ operands,
                                                         const float* __restrict__ bias_g, float* out, const float* __restrict__ stream) {
    extern __shared__ __attribute__((aligned(16))) f16x8 lds[];
    const int lane = threadIdx.x & 63, grp = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < kLayerFrags * 64; i += blockDim.x) lds[i] = weights[i];
    float* bias = reinterpret_cast<float*>(lds + kLayerFrags * 64);
    if (threadIdx.x < 256) bias[threadIdx.x] = bias_g[threadIdx.x];
    float* ring = bias + 256;                  // 4096 floats
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)lds + lane * 16;
    f16x8 xh[8][2];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int s = 0; s < 2; ++s) xh[c][s] = operands[((c * 2 + s) * 64 + lane) % (16 * 64)];
    f32x4 acc[8][2][2];
    int unit = 0;
    f32x4 sink = {0, 0, 0, 0};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(stream), 0, (320 * 1024 + 8192) * 4, 0x00020000);
    for (int pass = 0; pass < kPasses; ++pass) {
        for (int layer = 0; layer < kLayers; ++layer) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (MODE == 3) {
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + 32 * u + 4 * grp);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + 32 * u + 16 + 4 * grp);
                acc[u][0][0] = b0; acc[u][0][1] = b0; acc[u][1][0] = b1; acc[u][1][1] = b1;
                const int soff = __builtin_amdgcn_readfirstlane((((unit * 4096) % (300 * 1024)) + wave * 256) * 4);
                const float* src = stream + (size_t)((unit * 4096) % (300 * 1024)) + wave * 256 + lane * 4;
                float* dst = ring + wave * 256;       // (one 16-KiB slot written over and over: nobody reads it)
                tile_kloop<2>(lds0 + u * 16 * 1024, acc[u], xh, [&](int c) __attribute__((always_inline)) {
                    if (c == 0 || c == 4) {
                        const int piece = c >> 2;
                        if (MODE == 1 || MODE == 3) {
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 2048),
                                                             (__attribute__((address_space(3))) void*)(dst + piece * 2048), 16, 0, 0);
                        } else if (MODE == 4) {
                            // the same piece by buffer_load_dwordx4 ... lds: descriptor in SGPRs, the lane's 16 bytes in a constant
                            // voffset register, the piece's position in the scalar offset -- no per-piece vector address arithmetic
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + piece * 2048), 16,
                                                                 lane * 16, soff + piece * 8192, 0, 0);
                        } else if (MODE == 2) {
                            f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + piece * 2048));
                            asm volatile("" : "+v"(v));
                            sink = v;      // (overwritten, never accumulated: the load's register is free again at the next piece)
                        }
                    }
                });
                ++unit;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int s = 0; s < 2; ++s) to_operand(acc[u][0][s], acc[u][1][s], xh[u][s]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = sink[0];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int s = 0; s < 2; ++s) sum += (float)xh[c][s][0] + (float)xh[c][s][5];
    if (sum == 12345.678f) out[0] = sum;
}

// ------------------------------------------------------------------------------------------------ pipelined layers (the proposal)
// one layer: operands `in` -> `next`; tile u's epilogue is dealt out, one sample tile per 8 / NS k-blocks, behind tile u + 1's
// MFMAs (and the last tile's behind nothing: it is the layer's tail).  NS = 4: the 64-sample wave tile (one wave per SIMD);
// NS = 2: the same restructure at 32 samples per wave (two waves per SIMD) -- 64 + 64 operand registers + two tiles of
// accumulators instead of 64 + 128 accumulators.
template <int NS>
__device__ __forceinline__ void layer_pipelined(unsigned lds0, const float* bias, int grp, const f16x8 (&in)[8][NS], f16x8 (&next)[8][NS]) {
    f32x4 prev[2][NS];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        f32x4 acc[2][NS];
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + 32 * u + 4 * grp);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + 32 * u + 16 + 4 * grp);
#pragma unroll
        for (int s = 0; s < NS; ++s) { acc[0][s] = b0; acc[1][s] = b1; }
        tile_kloop<NS>(lds0 + u * 16 * 1024, acc, in, [&](int c) __attribute__((always_inline)) {
            constexpr int every = 8 / NS;
            if (u > 0 && (c % every) == every - 1) to_operand(prev[0][c / every], prev[1][c / every], next[u - 1][c / every]);
        });
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int s = 0; s < NS; ++s) prev[r][s] = acc[r][s];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) to_operand(prev[0][s], prev[1][s], next[7][s]);
}

template <int NS>
__global__ void __launch_bounds__(NS == 4 ? 256 : 512, NS == 4 ? 1 : 2) skeleton_pipelined(const f16x8* __restrict__ weights, const f16x8* __restrict__ operands,
                                                                                          const float* __restrict__ bias_g, float* out) {
    extern __shared__ __attribute__((aligned(16))) f16x8 lds[];
    const int lane = threadIdx.x & 63, grp = lane >> 4;
    for (int i = threadIdx.x; i < kLayerFrags * 64; i += blockDim.x) lds[i] = weights[i];
    float* bias = reinterpret_cast<float*>(lds + kLayerFrags * 64);
    if (threadIdx.x < 256) bias[threadIdx.x] = bias_g[threadIdx.x];
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)lds + lane * 16;
    f16x8 xa[8][NS], xb[8][NS];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int s = 0; s < NS; ++s) xa[c][s] = operands[((c * NS + s) * 64 + lane) % (16 * 64)];
    for (int pass = 0; pass < kPasses; ++pass) {
        for (int layer = 0; layer < kLayers; layer += 2) {
            layer_pipelined<NS>(lds0, bias, grp, xa, xb);
            layer_pipelined<NS>(lds0, bias, grp, xb, xa);
        }
    }
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int s = 0; s < NS; ++s) sum += (float)xa[c][s][0] + (float)xa[c][s][5];
    if (sum == 12345.678f) out[0] = sum;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    std::vector<_Float16> w(kLayerFrags * 64 * 8), x(16 * 64 * 8);
    std::vector<float> b(256);
    srand(1);
    auto rnd = [] { float u = 0; for (int i = 0; i < 4; ++i) u += (float)rand() / RAND_MAX - 0.5f; return u; };
    for (auto& v : w) v = (_Float16)(rnd() * 0.11f);      // (keeps the activations O(1) through the layers: random, non-trivial operands)
    for (auto& v : x) v = (_Float16)(rnd() > 0 ? rnd() * 0.7f : 0.0f);
    for (auto& v : b) v = rnd() * 0.1f;
    f16x8 *dw, *dx; float *db, *dout, *dstream;
    hipMalloc(&dw, w.size() * 2); hipMalloc(&dx, x.size() * 2); hipMalloc(&db, b.size() * 4); hipMalloc(&dout, 4);
    {   // the global weight stream of the DMA variants: 1.2 MB + slack, random
        std::vector<float> st(320 * 1024 + 8192);
        for (auto& v : st) v = rnd();
        hipMalloc(&dstream, st.size() * 4);
        hipMemcpy(dstream, st.data(), st.size() * 4, hipMemcpyHostToDevice);
    }
    hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    const size_t ldsb = kLayerFrags * 1024 + kBiasBytes + 16384;
    struct Variant { const char* name; const void* fn; int threads; };
    const Variant variants[] = {{"32 (as shipped: 2 waves/SIMD, tiles kept, converted after the layer)", (const void*)skeleton32, 512},
                                {"32p (2 waves/SIMD, tile u converted behind tile u+1's MFMAs)", (const void*)skeleton_pipelined<2>, 512},
                                {"64p (1 wave/SIMD, 64-sample wave tile, pipelined)", (const void*)skeleton_pipelined<4>, 256},
                                {"32 + LDS-DMA of the weight stream (2 x 1 KiB per wave and tile), never waited for", (const void*)skeleton32_dma<1>, 512},
                                {"32 + the same bytes by global_load_dwordx4 into registers", (const void*)skeleton32_dma<2>, 512},
                                {"32 + LDS-DMA + counted wait + workgroup barrier per tile", (const void*)skeleton32_dma<3>, 512},
                                {"32 + LDS-DMA by buffer_load ... lds (descriptor + scalar offset, no vector address arithmetic)", (const void*)skeleton32_dma<4>, 512}};
    for (const Variant& v : variants) {
        hipFuncSetAttribute(v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        hipFuncAttributes at;
        hipFuncGetAttributes(&at, v.fn);
        printf("registers per lane, shape %s: %d (scratch %zu B)\n", v.name, at.numRegs, (size_t)at.localSizeBytes);
    }
    const int blocks = 256;       // one workgroup per CU: 256 samples either way (8 x 32 or 4 x 64)
    // MACs per workgroup and launch: 256 samples x 256 x 256 per layer
    const double flop_per_launch = 2.0 * blocks * 256.0 * 256 * 256 * kLayers * kPasses;
    for (int round = 0; round < 3; ++round) {
        for (const Variant& v : variants) {
            auto t0 = std::chrono::steady_clock::now();
            long launches = 0;
            double elapsed = 0;
            void* args[] = {(void*)&dw, (void*)&dx, (void*)&db, (void*)&dout, (void*)&dstream};
            while (elapsed < seconds) {
                for (int i = 0; i < 10; ++i) hipLaunchKernel(v.fn, dim3(blocks), dim3(v.threads), args, ldsb, 0);
                hipDeviceSynchronize();
                launches += 10;
                elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
            printf("round %d shape %s: %.1f TFLOP/s (%.3f ms per launch of %d layers x %d passes)\n", round, v.name,
                   flop_per_launch * launches / elapsed / 1e12, elapsed / launches * 1e3, kLayers, kPasses);
            fflush(stdout);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
