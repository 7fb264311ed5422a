// Device-side building blocks of the split-precision (f16x3) kernels: the LDS unit ring, the 3-MFMA segment product,
// accumulator -> fp16 hi/lo operand conversion.  Shared by mlp_forward_f16.hip and mlp_backward_f16.hip.
#pragma once
#include "mlp_device.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kUnitBufFloats = 22 * 512;  // largest unit: 22 k-steps x 2 KiB (views layer of the points-aug MLP)
constexpr int kUnitBuffers = 3;

// ---- the two 16-bit operand formats of the single-product kernels -----------------------------------------------------------
// SNERF_PRECISION_F16 multiplies fp16 operands (11 significand bits, |v| <= 65504: the range watch below), SNERF_PRECISION_BF16
// bf16 operands (8 significand bits, fp32's exponent range: nothing to watch) -- BASELINE config 5's literal dtype.  Same MFMA
// rate, same fragment layouts, same 16-bit saved tensors; what differs is the conversion instruction, the MFMA opcode and the
// weight stream the kernel reads (mlp_plan.h: the bf16 streams hold one KiB per k-step, no lo fragments).  Operand registers
// are typed f16x8 in both modes: a container of eight 16-bit patterns; `BF` says how to read them.
template <bool BF>
__device__ __forceinline__ f32x16 mfma_32x32x16(const f16x8& a, const f16x8& b, const f32x16& c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
template <bool BF>
__device__ __forceinline__ f32x4 mfma_16x16x32(const f16x8& a, const f16x8& b, const f32x4& c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// two fp32 values -> the packed 16-bit pair (round to nearest even: v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32), ReLU optional.
// fp16: relu(fp16(v)) == fp16(relu(v)), so the maximum follows the conversion as ONE packed instruction; bf16 has no packed
// maximum, the two fp32 maxima come first.
template <bool BF, bool RELU>
__device__ __forceinline__ f16x2 pack_pair(f32x2 v) {
    if constexpr (BF) {
        if (RELU) v = {fmaxf(v[0], 0.0f), fmaxf(v[1], 0.0f)};
        return __builtin_bit_cast(f16x2, __builtin_convertvector(v, bf16x2));
    } else {
        f16x2 h = __builtin_convertvector(v, f16x2);
        if (RELU) {
            const f16x2 zero = {(_Float16)0.0f, (_Float16)0.0f};
            h = __builtin_elementwise_max(h, zero);
        }
        return h;
    }
}
template <bool BF>
__device__ __forceinline__ _Float16 pack_one(float v) {
    if constexpr (BF) return __builtin_bit_cast(_Float16, (__bf16)v);
    else return (_Float16)v;
}

#define SNERF_VMCNT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vmcnt(int n) {  // n is wave-uniform in [0, 63]; the count must be an immediate
    switch (n) {
        SNERF_VMCNT_CASE(0) SNERF_VMCNT_CASE(1) SNERF_VMCNT_CASE(2) SNERF_VMCNT_CASE(3) SNERF_VMCNT_CASE(4) SNERF_VMCNT_CASE(5)
        SNERF_VMCNT_CASE(6) SNERF_VMCNT_CASE(7) SNERF_VMCNT_CASE(8) SNERF_VMCNT_CASE(9) SNERF_VMCNT_CASE(10) SNERF_VMCNT_CASE(11)
        SNERF_VMCNT_CASE(12) SNERF_VMCNT_CASE(13) SNERF_VMCNT_CASE(14) SNERF_VMCNT_CASE(15) SNERF_VMCNT_CASE(16)
        SNERF_VMCNT_CASE(17) SNERF_VMCNT_CASE(18) SNERF_VMCNT_CASE(19) SNERF_VMCNT_CASE(20) SNERF_VMCNT_CASE(21)
        SNERF_VMCNT_CASE(22) SNERF_VMCNT_CASE(23) SNERF_VMCNT_CASE(24) SNERF_VMCNT_CASE(25) SNERF_VMCNT_CASE(26)
        SNERF_VMCNT_CASE(27) SNERF_VMCNT_CASE(28) SNERF_VMCNT_CASE(29) SNERF_VMCNT_CASE(30) SNERF_VMCNT_CASE(31)
        SNERF_VMCNT_CASE(32) SNERF_VMCNT_CASE(33) SNERF_VMCNT_CASE(34) SNERF_VMCNT_CASE(35) SNERF_VMCNT_CASE(36)
        SNERF_VMCNT_CASE(37) SNERF_VMCNT_CASE(38) SNERF_VMCNT_CASE(39) SNERF_VMCNT_CASE(40) SNERF_VMCNT_CASE(41)
        SNERF_VMCNT_CASE(42) SNERF_VMCNT_CASE(43) SNERF_VMCNT_CASE(44) SNERF_VMCNT_CASE(45) SNERF_VMCNT_CASE(46)
        SNERF_VMCNT_CASE(47) SNERF_VMCNT_CASE(48) SNERF_VMCNT_CASE(49) SNERF_VMCNT_CASE(50) SNERF_VMCNT_CASE(51)
        SNERF_VMCNT_CASE(52) SNERF_VMCNT_CASE(53) SNERF_VMCNT_CASE(54) SNERF_VMCNT_CASE(55) SNERF_VMCNT_CASE(56)
        SNERF_VMCNT_CASE(57) SNERF_VMCNT_CASE(58) SNERF_VMCNT_CASE(59) SNERF_VMCNT_CASE(60) SNERF_VMCNT_CASE(61)
        SNERF_VMCNT_CASE(62) SNERF_VMCNT_CASE(63)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
#undef SNERF_VMCNT_CASE

// Weight stream L2 -> LDS by LDS-DMA through a ring of three unit buffers, TWO units ahead of the one being consumed:
// at fp16 rates one tile's MFMAs (~0.65 us) are shorter than the DMA's issue-to-landing time, so a single unit of
// run-ahead leaves the matrix pipe waiting.  A unit of k k-steps is 2k KiB-pieces (hi + lo fragment per k-step), k even,
// so each of the 4 waves issues exactly k/2 DMA instructions per unit and can wait with a COUNTED vmcnt that leaves the
// younger unit in flight (a plain __syncthreads() would drain it: its fence waits vmcnt(0) while LDS-DMA is pending).
//
// Unit layout (mlp_pack.hip): [hi fragment of every k-step, 1 KiB each][lo fragment of every k-step].  The single-product
// kernels (P = 1, SNERF_PRECISION_F16) request only the hi half of each unit -- k KiB-pieces, rounded up to a multiple of
// four (one DMA instruction per wave; the surplus pieces are the first lo fragments, which nobody reads).
// NW = waves of the workgroup that share the ring (4, or 8 for the single-product forward: two waves per SIMD reading ONE
// weight stream -- the same occupancy as two 4-wave workgroups per CU at half the L2 -> LDS traffic, which is what bounds
// the 16-bit forward: 2 x 128 KB per layer and CU at fp16 MFMA rates is ~150 GB/s per CU, twice what a CU draws from L2).
// D = ring slots (default three, "two units ahead").  A deeper ring requests unit i + D - 1 while unit i is consumed, and --
// what matters to kernels that also STORE between units -- the counted wait at unit i then leaves everything the wave issued
// during the last D - 2 units in flight, not just the last one: vmcnt retires in order, so a store whose write acknowledgement
// takes longer than one unit's matrix work (HBM under a 3 TB/s write stream: ~2 us) otherwise stalls the next acquire.
// KF = floats per k-step in the STREAM: 512 (hi + lo fragment; the fp16 streams) or 256 (the bf16 streams: hi only).
template <int P, int NW = 4, int D = kUnitBuffers, int KF = 512>
struct UnitStreamT {
    static_assert(KF == 512 || (KF == 256 && P == 1), "compact streams are single-product");
    static_assert(NW == 4 || (NW == 8 && P == 1), "8-wave rings are built for the single-product kernels only");
    static_assert(D >= 3 && D <= 6, "ring depth");
    static constexpr int kWaves = NW;
    static constexpr int kSlots = D;
    int hist[D > 3 ? D - 3 : 1];   // vector-memory instructions of the D - 3 units before the current one, newest first
    // The stream is addressed through a buffer descriptor (round 5): `buffer_load_dwordx4 ... lds` with the lane's 16 bytes in a
    // CONSTANT offset register and the piece's position in a scalar -- no per-piece 64-bit vector address arithmetic.  In the
    // trunk's skeleton the LDS-DMA of the weight stream costs 8 % with global_load_lds and 5.4 % this way
    // (tools/probes/m64_skeleton.hip, profiles/r05_m64_skeleton.txt).
    // Measured on the kernels (same box, profiles/r05_srd_ab.txt): 16-bit rendering +2-3 %, the bf16s8 iteration -1 %, the
    // split-precision (P = 3) rendering kernel 1.3 % slower -- P = 3 keeps the global form.
    __amdgpu_buffer_rsrc_t rsrc;
    const float* stream_base;
    int lane_bytes;          // this lane's 16 bytes inside a KiB piece
    int fetch_off;           // byte offset (from the stream's first byte) of the next unit to request
    float* lds;
    int slot;                // ring slot of the unit about to be consumed
    int lane, wave;
    int slot_floats;         // size of one ring slot (kUnitBufFloats, or less when the caller sized the ring to its units)

    int pend_off;            // byte offset of the piece requested last (one DMA instruction per call of fetch_piece)
    float* pend_dst;
    int pend_left;           // DMA instructions this wave still has to issue for it
    int issued;              // DMA instructions issued for the youngest requested unit (>= its piece count)
    int younger;             // other vector-memory instructions (tile stores) this wave issued since that request was opened

    // Branch-free on purpose: a conditional here would cut the unrolled MFMA loop into basic blocks and the fragment reads
    // could no longer be scheduled a k-step ahead.  Once the unit's pieces are all requested the same (last) piece is simply
    // requested again -- idempotent, and it only happens where a unit has more k-step pairs than its second successor has
    // pieces (a few times per pass).  `issued` feeds the counted vmcnt of the next acquire().  (A select-free variant that
    // keeps reading the stream into the rest of the slot saves a third of the scalar instructions and measured 1 % slower.)
    __device__ __forceinline__ void fetch_piece() {
#ifdef SNERF_ABL_NODMA
        pend_left -= pend_left > 0 ? 1 : 0;
        return;
#endif
        const int adv = pend_left > 0 ? NW * 256 : 0;
        pend_off += adv * 4; pend_dst += adv;
        if constexpr (P == 1) {
            // (Also tried: the same instruction as one asm statement, M0 saved and restored around it, and the tiles' bias vectors
            // read one tile ahead by asm reads so that the compiler's own `s_waitcnt lgkmcnt(1)` -- it cannot see the asm fragment
            // reads in flight -- no longer sits in front of every tile's first MFMA: 192 -> 14 such waits, 4 spilled registers,
            // and no faster: 0.270 against 0.264 ms, profiles/r05_srd_ab.txt.  Not kept.)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)pend_dst, 16, lane_bytes, pend_off, 0, 0);
        } else {      // (the split-precision kernels measured 1.3 % SLOWER with the descriptor form: they keep global_load_lds)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stream_base + (pend_off >> 2) + lane * 4),
                                             (__attribute__((address_space(3))) void*)pend_dst, 16, 0, 0);
        }
        pend_left -= pend_left > 0 ? 1 : 0;
        ++issued;
    }
    // n vector-memory instructions (stores) were just issued by this wave; n must not exceed the real count
    __device__ __forceinline__ void note_vmem(int n) { younger += n; }
    __device__ __forceinline__ void finish_fetch() {
        while (pend_left > 0) fetch_piece();
    }
    // No further unit to request: the (branch-free) fetch_piece calls of the remaining k-steps re-read one valid KiB of
    // the stream into a per-wave dump area instead of touching a live buffer.
    __device__ __forceinline__ void issued_next_none() {
        pend_off = 0;
        pend_dst = lds + D * slot_floats + wave * 256;  // dump: NW KiB right after the ring
        pend_left = 0;
        issued = 0;
    }
    __device__ __forceinline__ void begin_fetch(int ksteps, int into_slot) {
        pend_off = fetch_off + (wave * 256 - NW * 256) * 4;   // fetch_piece pre-increments
        pend_dst = lds + into_slot * slot_floats + wave * 256 - NW * 256;
        pend_left = P == 3 ? ksteps >> 1 : (ksteps + NW - 1) / NW;
        issued = 0;
        fetch_off += ksteps * KF * 4;
    }
    __device__ __forceinline__ void fetch(int ksteps, int into_slot) {
        begin_fetch(ksteps, into_slot);
        finish_fetch();
    }
    __device__ __forceinline__ void start(const float* first, float* lds_base, int ks0, int ks1, int lane_, int wave_,
                                          int slot_floats_ = kUnitBufFloats) {
        // (the descriptor covers 2 GiB from the stream's first byte: the streams are a few MB; wave-uniform inputs only)
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(first), 0, 0x7ffffff0, 0x00020000);
        stream_base = first;
        fetch_off = 0; lds = lds_base; slot = 0; lane = lane_; wave = wave_; pend_left = 0; issued = 0;
        lane_bytes = lane_ * 16;
        younger = 0; slot_floats = slot_floats_;
#pragma unroll
        for (int i = 0; i < (D > 3 ? D - 3 : 1); ++i) hist[i] = 0;
        fetch(ks0, 0);
        if (ks1 > 0) fetch(ks1, 1);
        if (ks1 <= 0) issued_next_none();
    }
    // deeper rings: units 2 .. D - 2 of the initial run-ahead (call right after start(); 0 k-steps = no such unit)
    __device__ __forceinline__ void start_more(int unit, int ks) {
#pragma unroll
        for (int i = D - 4; i > 0; --i) hist[i] = hist[i - 1];
        hist[0] = issued;
        if (ks > 0) fetch(ks, unit); else issued_next_none();
    }
    // Unit i becomes readable.  `next` = k-steps of unit i+1 (still in flight afterwards), `next2` = k-steps of unit i+D-1
    // (i+2 for the three-slot ring), which is requested now into the slot unit i-1 just vacated (0 = no such unit).
    // The request for unit i+2 is only OPENED here; its DMA instructions are issued one per two k-steps from inside the
    // MFMA loop (fetch_piece) so that their issue cost (~60-100 cycles each) does not sit in front of the tile's MFMAs.
    __device__ __forceinline__ const float* acquire(int next, int next2) {
        finish_fetch();                                         // (units shorter than their successor's piece count)
        // Everything OLDER than the request for unit i+1 must have landed; what may stay in flight is that request's DMA
        // instructions plus whatever else this wave issued after it was opened (vmcnt counts loads, stores and LDS-DMA
        // together, in issue order).  Kernels that store tiles between units report those stores (note_vmem); ignoring
        // them made the wait stricter than needed: it drained the run-ahead DMA and the stores at every unit of the
        // training forward (MFMA pipe 34 % busy against 48 % without the stores).  The counter saturates at 63.
        const int cur = issued + younger;
        int in_flight = cur;
#pragma unroll
        for (int i = 0; i < D - 3; ++i) in_flight += hist[i];
#pragma unroll
        for (int i = D - 4; i > 0; --i) hist[i] = hist[i - 1];
        if (D > 3) hist[0] = cur;
        const int allowed = next > 0 ? (in_flight < 63 ? in_flight : 63) : 0;
        // (the generic 64-way dispatch compiles to a compare-and-branch tree of ~30 scalar instructions, and with one wave
        // per SIMD every instruction is an issue slot: the steady-state counts of the 256-wide trunk -- the successor's
        // k/2 resp. k/4 DMA instructions and nothing else -- are tested first)
#ifdef SNERF_PROBE_NO_VMWAIT
        // timing probe (wrong results): the DMA is issued but nobody waits for it -- what the waits cost, as against the issue
        (void)allowed;
#else
        // (... times the D - 2 units that stay in flight)
        constexpr int kSteady = (D - 2) * (P == 3 ? 8 : 16 / NW);
        if (D <= 4 && allowed == kSteady) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kSteady) : "memory");
        else wait_vmcnt(allowed);
#endif
        younger = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my LDS reads of unit i-1 are complete
#ifndef SNERF_ABL_NOBARRIER
        __builtin_amdgcn_s_barrier();
#endif
        const float* ready = lds + slot * slot_floats;
        const int vacated = slot == 0 ? D - 1 : slot - 1;
        if (next2 > 0) begin_fetch(next2, vacated); else issued_next_none();
        slot = slot == D - 1 ? 0 : slot + 1;
        return ready;
    }
};
using UnitStream = UnitStreamT<3>;

// fp16 range watch (include/simplenerf_hip.h, "Range").  An activation (or encoded input) beyond 65504 converts to +-inf
// (v_cvt_pk_f16_f32 rounds to nearest), and an infinite operand k makes EVERY pre-activation of the next layer non-finite
// for that sample: y_j = sum_k w_jk x_k holds inf . w_jk = +-inf, or NaN where w_jk == 0, for every unit j.  So it is enough
// to look at ONE pre-activation per sample and layer, before its ReLU can mask it: `probe` keeps a NaN-propagating maximum
// of their magnitudes (v_maximum3_f32 with |x| modifiers: one instruction per layer and sample slot, where watching every
// converted operand cost 32-64 and 2-5 % of the kernel).  Conversions nothing downstream multiplies (the views layer's
// activations, which only the training forward converts -- to save them) are watched directly with `see`.
// `report` runs once per wave at the end of the kernel: a non-finite probe ORs kRangeActivation into the device's pinned
// host word (system-scope atomic, executed by offending lanes only).
struct RangeWatch {
    float worst = 0.0f;
    __device__ __forceinline__ void probe(float pre_activation) {
        worst = __builtin_elementwise_maximum(worst, __builtin_fabsf(pre_activation));
        asm volatile("" : "+v"(worst));   // evaluated here, not sunk to the end of the kernel with its operand kept alive
    }
    __device__ __forceinline__ void see(const f16x8& converted) {
        float m = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) m = __builtin_elementwise_maximum(m, __builtin_fabsf((float)converted[i]));
        probe(m);
    }
    // `weight_range`: the consumed packed buffer's word (one thread of the grid forwards it)
    __device__ __forceinline__ void report(int* flag, const int* weight_range) const {
        if (!(worst <= 3.0e38f) && flag) __hip_atomic_fetch_or(flag, snerf::kRangeActivation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (flag && weight_range && blockIdx.x == 0 && threadIdx.x == 0 && *weight_range != 0)
            __hip_atomic_fetch_or(flag, snerf::kRangeWeight, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
};

// Converts accumulator registers (2i, 2i+1) of a finished tile -- ReLU optional -- into the fp16 hi/lo pair they form
// in the next layer's operand: registers 8s..8s+7 are the 8 elements of k-step s.
template <bool RELU>
struct TileSplitter {
    const f32x16* src;
    f16x8 *h0, *l0, *h1, *l1;
    bool on;
    __device__ __forceinline__ void step(int i) const {  // i = 0..7 (compile-time after unrolling)
        if (!on) return;
#ifdef SNERF_ABL_NOSPLIT
        if (i > 0) return;
#endif
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = 2 * i + e;
            float v = (*src)[r];
            if (RELU) v = fmaxf(v, 0.0f);
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            if (r < 8) { (*h0)[r] = hi; (*l0)[r] = lo; } else { (*h1)[r - 8] = hi; (*l1)[r - 8] = lo; }
        }
    }
};

// acc += W[tile rows, segment columns] . X over NKS k-steps; `p` walks the unit (lane offset already applied).
// Fragments for k-step ks+1 are requested before the MFMAs of k-step ks are issued (LDS latency hides under 96 MFMA
// cycles), and `side.step()` slots one slice of the previous tile's ReLU + hi/lo split (VALU) behind each k-step's MFMAs.
// LDS fragment reads the compiler does not see, with counted waits that carry the destination registers as a
// dependency (everything that uses them is ordered after the wait).  `byte_offset` folds to an immediate once the caller's
// k-step loop is unrolled.
__device__ __forceinline__ f16x8 lds_read_f16x8(unsigned lds_byte_address, int byte_offset) {
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_byte_address), "i"(byte_offset) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait_all_but_two(f16x8& a, f16x8& b) {
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a), "+v"(b)::"memory");
}
__device__ __forceinline__ void lds_wait_all(f16x8& a, f16x8& b) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)::"memory");
}

// Three MFMAs per k-step on the fp16 pipe (hi*hi, hi*lo, lo*hi).  The weight fragments of k-step ks+1 are requested
// BEFORE the MFMAs of k-step ks, into the other register pair, and k-step ks waits with lgkmcnt(2) -- "all but the two
// newest LDS reads".  With compiler-visible reads this software pipeline does not survive: the register allocator folds
// the two fragment pairs into one and the wait-count pass puts lgkmcnt(0) right behind each pair of reads, so every
// k-step paid the full LDS latency (~130 cycles) in front of 96 cycles of MFMAs -- the matrix pipe was 41-42 % busy
// (PMC) and a third of all wave cycles were parked at s_waitcnt.
// `unit_ks` = k-steps of the whole unit `p` points into: its lo fragments start unit_ks KiB after its hi fragments.
template <int NKS, int NB, typename Side, typename Stream>
__device__ __forceinline__ void seg_mfma(f32x16& acc, const float*& p, int unit_ks, const f16x8 (&bh)[NB],
                                         const f16x8 (&bl)[NB], const Side& side, int side_first, Stream& st) {
    static_assert(NB >= NKS, "operand array too short");
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
    const unsigned base_lo = base + unit_ks * 1024;
    f16x8 ah = lds_read_f16x8(base, 0);
    f16x8 al = lds_read_f16x8(base_lo, 0);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        f16x8 nah = ah, nal = al;
        if (ks + 1 < NKS) {
            nah = lds_read_f16x8(base, (ks + 1) * 1024);
            nal = lds_read_f16x8(base_lo, (ks + 1) * 1024);
            lds_wait_all_but_two(ah, al);
        } else {
            lds_wait_all(ah, al);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], acc, 0, 0, 0);
        if (side_first + ks < 8) side.step(side_first + ks);
        if ((ks & 1) == 0) st.fetch_piece();
        ah = nah; al = nal;
    }
    p += NKS * 256;
}

// Single-product variant (SNERF_PRECISION_F16): one fp16 MFMA per k-step (32 cycles), so a fragment has to be requested
// FOUR k-steps before its use to cover the LDS latency from one wave per SIMD; counted waits as above.
template <int N>
__device__ __forceinline__ void lds_wait_all_but(f16x8& a) {
    static_assert(N >= 0 && N <= 4, "wait count");
    if (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)::"memory");
    if (N == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a)::"memory");
    if (N == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a)::"memory");
    if (N == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a)::"memory");
    if (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a)::"memory");
}
template <int NKS, bool BF, int NB, typename Stream>
__device__ __forceinline__ void seg_mfma1(f32x16& acc, const float*& p, const f16x8 (&bh)[NB], Stream& st) {
    static_assert(NB >= NKS, "operand array too short");
    constexpr int AHEAD = 4;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
    f16x8 a[AHEAD + 1];
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
        if (i < NKS) a[i] = lds_read_f16x8(base, i * 1024);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        if (ks + AHEAD < NKS) a[(ks + AHEAD) % (AHEAD + 1)] = lds_read_f16x8(base, (ks + AHEAD) * 1024);
        constexpr int kLast = NKS - 1;
        const int newer = kLast - ks < AHEAD ? kLast - ks : AHEAD;   // reads issued after the one for k-step ks
        f16x8& cur = a[ks % (AHEAD + 1)];
        if (newer == 4) lds_wait_all_but<4>(cur);
        else if (newer == 3) lds_wait_all_but<3>(cur);
        else if (newer == 2) lds_wait_all_but<2>(cur);
        else if (newer == 1) lds_wait_all_but<1>(cur);
        else lds_wait_all_but<0>(cur);
        acc = mfma_32x32x16<BF>(cur, bh[ks], acc);
        if ((ks & (Stream::kWaves - 1)) == 0) st.fetch_piece();   // one KiB-piece per wave and kWaves k-steps
    }
    p += NKS * 256;
}

// The same with `side(ks)` slotted behind the MFMA of every k-step (ks is a compile-time constant after unrolling): VALU work
// that does not depend on this tile's accumulator issues while the matrix pipe runs the MFMA (32 cycles = eight VALU slots).
// With one wave per SIMD nothing else fills those slots -- work placed BETWEEN tiles instead runs with the matrix pipe idle.
template <int NKS, bool BF, int NB, typename Stream, typename Side>
__device__ __forceinline__ void seg_mfma1_side(f32x16& acc, const float*& p, const f16x8 (&bh)[NB], Stream& st, Side&& side) {
    static_assert(NB >= NKS, "operand array too short");
    constexpr int AHEAD = 4;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
    f16x8 a[AHEAD + 1];
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
        if (i < NKS) a[i] = lds_read_f16x8(base, i * 1024);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        if (ks + AHEAD < NKS) a[(ks + AHEAD) % (AHEAD + 1)] = lds_read_f16x8(base, (ks + AHEAD) * 1024);
        constexpr int kLast = NKS - 1;
        const int newer = kLast - ks < AHEAD ? kLast - ks : AHEAD;
        f16x8& cur = a[ks % (AHEAD + 1)];
        if (newer == 4) lds_wait_all_but<4>(cur);
        else if (newer == 3) lds_wait_all_but<3>(cur);
        else if (newer == 2) lds_wait_all_but<2>(cur);
        else if (newer == 1) lds_wait_all_but<1>(cur);
        else lds_wait_all_but<0>(cur);
        acc = mfma_32x32x16<BF>(cur, bh[ks], acc);
        if ((ks & (Stream::kWaves - 1)) == 0) st.fetch_piece();
        side(ks);
    }
    p += NKS * 256;
}

struct NoSide {
    __device__ __forceinline__ void step(int) const {}
};

// P = 3: split-precision product (hi.hi + hi.lo + lo.hi); P = 1: hi.hi only.
template <int P, int NKS, bool BF, int NB, typename Stream>
__device__ __forceinline__ void seg_product(f32x16& acc, const float*& p, int unit_ks, const f16x8 (&bh)[NB],
                                            const f16x8 (&bl)[NB], Stream& st) {
    if constexpr (P == 3) seg_mfma<NKS>(acc, p, unit_ks, bh, bl, NoSide(), 8, st);
    else seg_mfma1<NKS, BF>(acc, p, bh, st);
}

__device__ __forceinline__ void tile_bias(f32x16& acc, const float* __restrict__ bias, int half) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(bias + 8 * g + 4 * half);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[4 * g + q] = v[q];
    }
}

// Single-product kernels: only the hi operand exists, so the ReLU can follow the conversion as ONE packed fp16 maximum per
// register pair (relu(fp16(v)) == fp16(relu(v)) bit for bit: rounding is monotone and keeps the sign) instead of one fp32
// maximum per value before it -- 64 fewer VALU instructions per layer and wave.
template <bool RELU, bool BF = false>
__device__ __forceinline__ void convert_tile(const f32x16& acc, f16x8& h0, f16x8& h1) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f16x2 a = pack_pair<BF, RELU>(f32x2{acc[j], acc[j + 1]});
        const f16x2 b = pack_pair<BF, RELU>(f32x2{acc[8 + j], acc[9 + j]});
        h0[j] = a[0]; h0[j + 1] = a[1];
        h1[j] = b[0]; h1[j + 1] = b[1];
    }
}

// (ReLU and) split one finished accumulator tile into the two k-steps it feeds in the next layer.
template <bool RELU>
__device__ __forceinline__ void split_tile(const f32x16& acc, f16x8& h0, f16x8& l0, f16x8& h1, f16x8& l1) {
#ifdef SNERF_ABL_NOCONVERT
    h0[0] = (_Float16)acc[0]; l0[0] = (_Float16)acc[1]; h1[0] = (_Float16)acc[8]; l1[0] = (_Float16)acc[9];
    return;
#endif
    // two values at a time, so that each conversion is one packed instruction (v_cvt_pk_f16_f32, round to nearest even)
    // and the halves land in adjacent lanes of the fragment without separate packing moves
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        f32x2 a = {acc[j], acc[j + 1]}, b = {acc[8 + j], acc[9 + j]};
        if (RELU) {
            a = {fmaxf(a[0], 0.0f), fmaxf(a[1], 0.0f)};
            b = {fmaxf(b[0], 0.0f), fmaxf(b[1], 0.0f)};
        }
        const f16x2 ah = __builtin_convertvector(a, f16x2), bh = __builtin_convertvector(b, f16x2);
        const f16x2 al = __builtin_convertvector(a - __builtin_convertvector(ah, f32x2), f16x2);
        const f16x2 bl = __builtin_convertvector(b - __builtin_convertvector(bh, f32x2), f16x2);
        h0[j] = ah[0]; h0[j + 1] = ah[1]; l0[j] = al[0]; l0[j + 1] = al[1];
        h1[j] = bh[0]; h1[j + 1] = bh[1]; l1[j] = bl[0]; l1[j + 1] = bl[1];
    }
}

// sum over the tile's 32 features (this lane half's 16) of w[f] * relu(acc)
__device__ __forceinline__ float tile_dot_relu(const f32x16& acc, const float* __restrict__ w, int half) {
    float s = 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + 8 * g + 4 * half);
#pragma unroll
        for (int q = 0; q < 4; ++q) s = fmaf(v[q], fmaxf(acc[4 * g + q], 0.0f), s);
    }
    return s;
}

// (BF: bf16 patterns in h, l unused -- the bf16 mode is single-product)
template <int NREG, int NKS, bool BF = false>
__device__ __forceinline__ void split_encoding(const float (&pe)[NREG], f16x8 (&h)[NKS], f16x8 (&l)[NKS]) {
    static_assert(NREG == NKS * 8, "8 encoding registers per k-step");
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = pe[8 * ks + j];
            const _Float16 hi = pack_one<BF>(v);
            h[ks][j] = hi;
            l[ks][j] = BF ? (_Float16)0.0f : (_Float16)(v - (float)hi);
        }
}


// ReLU sign bits of U (even) finished accumulator tiles -> the mask words of this wave block (MlpPlan::act_mask), and
// their use in the backward chain: acc <- acc . [the forward's activation was > 0].
template <int U>
__device__ __forceinline__ void store_relu_masks(const f32x16* acc, unsigned* __restrict__ masks, int t0, int lane) {
#pragma unroll
    for (int u = 0; u < U; u += 2) {
        unsigned word = 0;
#pragma unroll
        for (int r = 31; r >= 1; r -= 2) word = positive_bits(word, acc[u + (r >> 4)][r & 15], acc[u + (r >> 4)][(r & 15) - 1]);
        __builtin_nontemporal_store(word, masks + ((t0 + u) >> 1) * 64 + lane);
    }
}
// One tile at a time (the single-product forward calls this right behind each tile's MFMAs, so the compares run in the
// shadow of the next tile's matrix work and no second pass over all accumulators is needed): the tile's 16 sign bits,
// shifted to the half of its pair's word.  `bits` collects a pair; the word is stored when its second tile arrives.
__device__ __forceinline__ void relu_mask_tile(const f32x16& acc, int u, unsigned& bits, unsigned* __restrict__ masks, int t0,
                                               int lane) {
    unsigned m = 0;
#pragma unroll
    for (int r = 15; r >= 1; r -= 2) m = positive_bits(m, acc[r], acc[r - 1]);
    if ((u & 1) == 0) {
        bits = m;
    } else {
        __builtin_nontemporal_store(bits | (m << 16), masks + ((t0 + u) >> 1) * 64 + lane);
    }
}
template <int U>
__device__ __forceinline__ void apply_relu_masks(f32x16* acc, const unsigned* __restrict__ masks, int t0, int lane) {
#pragma unroll
    for (int u = 0; u < U; u += 2) {
        const unsigned word = __builtin_nontemporal_load(masks + ((t0 + u) >> 1) * 64 + lane);
#pragma unroll
        for (int r = 0; r < 32; ++r) acc[u + (r >> 4)][r & 15] = keep_if_bit(acc[u + (r >> 4)][r & 15], word, r);
    }
}

// The same in two steps, so that the words of the NEXT layer can be requested before a long stretch of matrix work and
// their memory latency is hidden behind it (load_relu_words early, mask_with_words when the tiles are finished).
template <int U>
__device__ __forceinline__ void load_relu_words(unsigned (&words)[U / 2], const unsigned* __restrict__ masks, int t0, int lane) {
#pragma unroll
    for (int p = 0; p < U / 2; ++p) words[p] = __builtin_nontemporal_load(masks + ((t0 >> 1) + p) * 64 + lane);   // read once
}
template <int U>
__device__ __forceinline__ void mask_with_words(f32x16* acc, const unsigned (&words)[U / 2]) {
#pragma unroll
    for (int u = 0; u < U; u += 2)
#pragma unroll
        for (int r = 0; r < 32; ++r) acc[u + (r >> 4)][r & 15] = keep_if_bit(acc[u + (r >> 4)][r & 15], words[u >> 1], r);
}

// 16-bit saved tiles (SNERF_PRECISION_F16): NKS consecutive operand fragments -> NKS "pieces" of 1 KiB.  A piece is the
// fragment image of one 16-feature k-step, [slot = 2 * sample + lane half][8 x 16 bit]: sample j owns 32 consecutive
// bytes holding features 16s + {0..3, 8..11, 4..7, 12..15} (feature 8g + 4h + q sits at byte 16h + 8g + 2q) -- sample-major
// rows that the weight-gradient kernel reads back TRANSPOSED (ds_read_b64_tr_b16).  Every store writes 1 KiB contiguous.
template <int NKS, int NB>
__device__ __forceinline__ void store_pieces(const f16x8 (&frag)[NB], _Float16* __restrict__ rows, int lane) {
    static_assert(NB >= NKS, "fragment array too short");
    const int slot = 2 * (lane & 31) + (lane >> 5);
#pragma unroll
    // non-temporal: the saved tensors (GBs per pass) are read back only by the backward kernels, long after they have left
    // every cache; the plain store policy cost the storing forward 7 % and the backward 9 % (r02, A/B builds)
    for (int s = 0; s < NKS; ++s) {
#ifdef SNERF_PROBE_HALF_X      // traffic ablation (tools/probes/build_variant.py; WRONG results): the bytes an fp8 X tile would take
        if (NKS >= 8 && (s & 1)) continue;
#endif
        __builtin_nontemporal_store(frag[s], reinterpret_cast<f16x8*>(rows + s * 512 + slot * 8));
    }
}

// 8-bit saved tiles (SNERF_PRECISION_F16S8): the trunk activations h_1 .. h_D-1 are read back ONLY by the weight-gradient
// kernel, as the X operand of a contraction over ~10^6 samples -- a per-element rounding error of 2^-4 averages out there (it is
// never seen by the forward or the backward chain) -- so they are kept as plain fp8 e4m3: half the bytes of the fp16 pieces on
// the way out and on the way back in (measured bound of the saving: profiles/r04_train_f16_traffic_ablation.txt).  e4m3 spans
// 2^-9 .. 448 with four significant bits from 2^-6 up; post-ReLU activations above 448 are CLAMPED there (in this operand of the
// weight gradients only: v_cvt_scalef32_pk_fp8_f16 returns NaN above the range, tools/probes/cvt_fp8_scale.hip), values
// below 2^-10 contribute nothing -- as they would not to the sum.  A tile (two k-steps) becomes one KiB: slot = 2 * sample +
// lane half holds 16 bytes, k-step 2u's eight elements then k-step 2u+1's; tile after tile and layer after layer PACKED from
// where h_1's 16-bit rows would begin (one-KiB images with one-KiB holes between them, i.e. the 16-bit row numbers, read back
// SLOWER than the 16-bit pieces: half of every two KiB never requested).  The weight-gradient kernel reads it back
// transposed with ds_read_b64_tr_b8 and widens with v_cvt_scalef32_pk_f16_fp8 (mlp_backward.hip).
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// `frag`: NKS post-ReLU (non-negative) fp16 fragments of this wave block -- BF (SNERF_PRECISION_BF16S8): bf16 bits in the same
// registers; bf16 has no packed minimum, but non-negative values order like their bit patterns, so the clamp is an UNSIGNED
// 16-bit minimum against 448's (v_cvt_scalef32_pk_fp8_bf16 returns NaN above it too: tools/probes/cvt_fp8_bf16.hip)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
template <int NKS, bool BF = false, int NB>
__device__ __forceinline__ void store_pieces8(const f16x8 (&frag)[NB], _Float16* __restrict__ rows, int lane) {
    static_assert(NB >= NKS && NKS % 2 == 0, "whole tiles");
    const int slot = 2 * (lane & 31) + (lane >> 5);
    const f16x2 top = {(_Float16)448.0f, (_Float16)448.0f};
    const u16x2 top_bits = {0x43E0, 0x43E0};            // 448.0 as bf16
#pragma unroll
    for (int u = 0; u < NKS / 2; ++u) {
        u32x4 w;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const f16x8& f = frag[2 * u + k];
                s16x2 p = {0, 0};
                if constexpr (BF) {
                    const u16x2 lo = __builtin_elementwise_min(__builtin_bit_cast(u16x2, f16x2{f[4 * d], f[4 * d + 1]}), top_bits);
                    const u16x2 hi = __builtin_elementwise_min(__builtin_bit_cast(u16x2, f16x2{f[4 * d + 2], f[4 * d + 3]}), top_bits);
                    p = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(p, __builtin_bit_cast(bf16x2, lo), 1.0f, false);
                    p = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(p, __builtin_bit_cast(bf16x2, hi), 1.0f, true);
                } else {
                    p = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p, __builtin_elementwise_min(f16x2{f[4 * d], f[4 * d + 1]}, top), 1.0f, false);
                    p = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p, __builtin_elementwise_min(f16x2{f[4 * d + 2], f[4 * d + 3]}, top), 1.0f, true);
                }
                w[2 * k + d] = __builtin_bit_cast(unsigned, p);
            }
        __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(rows + u * 512 + slot * 8));
    }
}

// One finished accumulator tile -> rows 32u .. 32u+31 of a [feature][32-sample] fp32 tile (training: saved activations).
template <bool RELU>
__device__ __forceinline__ void store_tile_rows(const f32x16& acc, float* __restrict__ rows, int lane) {
    const int j = lane & 31, half = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = (r & 3) + 8 * (r >> 2) + 4 * half;
        __builtin_nontemporal_store(RELU ? fmaxf(acc[r], 0.0f) : acc[r], rows + f * 32 + j);
    }
}

