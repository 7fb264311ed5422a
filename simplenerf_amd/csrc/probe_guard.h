// Force-included into every translation unit of the library (build.py: -include probe_guard.h).
//
// The kernels carry diagnostic switches -- ablations that remove a stage to price it, traffic probes that skip half of an
// operand, in-kernel clock stamps.  Most of them produce WRONG results by design.  They are for tools/probes/build_variant.py,
// which builds gpurun_abl_<name>.so beside the shipped library and defines SNERF_PROBE_BUILD; a translation unit of the shipped
// library that sees one of them -- a stray -D in HIPCC_COMPILE_FLAGS_APPEND, CXXFLAGS, a wrapper script -- must not compile.
// tests/test_host_logic.py::test_every_diagnostic_switch_is_guarded holds this list to the sources.
#pragma once

#if !defined(SNERF_PROBE_BUILD)
#if defined(SNERF_ABL_NODMA) || defined(SNERF_ABL_NOBARRIER) || defined(SNERF_ABL_NOSPLIT) || defined(SNERF_ABL_NOCONVERT) || \
    defined(SNERF_ABL_F32_NODMA) || defined(SNERF_ABL_CHAIN_NOSTORE) || defined(SNERF_ABL_CHAIN_NOEPI) ||                     \
    defined(SNERF_ABL_CHAIN_NODMA) || defined(SNERF_ABL_CHAIN_NOBARRIER) || defined(SNERF_PROBE_SMALL_RING) ||                 \
    defined(SNERF_PROBE_NO_M16) || defined(SNERF_PROBE_HALF_DY) || defined(SNERF_PROBE_HALF_X) || defined(SNERF_PROBE_RING4) || \
    defined(SNERF_PROBE_NO_SMALL_FOLD) || defined(SNERF_PROBE_DEEP) || defined(SNERF_PROBE_WGRAD_NOMATH) ||                    \
    defined(SNERF_PROBE_NO_K4K5_FUSION) || defined(SNERF_PROBE_M16_NOENCODE) || defined(SNERF_PROBE_NO_SIDE_BY_SIDE) || defined(SNERF_PROBE_M16_UNPAIRED) || defined(SNERF_PROBE_GEMM_PLAIN_ORDER) || defined(SNERF_PROBE_GEMM_NO_PREFETCH) || defined(SNERF_PROBE_GEMM_NOLOAD) || defined(SNERF_PROBE_GEMM_SCALAR_STAGING) || defined(SNERF_PROBE_GENERIC_PACKED_ROW) || defined(SNERF_PROBE_GEMM_NOSTORE) || defined(SNERF_PROBE_NO_VMWAIT) || defined(SNERF_CLOCK_STAMP)
#error "a diagnostic switch (SNERF_ABL_* / SNERF_PROBE_* / SNERF_CLOCK_STAMP) is defined in a build of the shipped library: these produce wrong results by design; build variants with tools/probes/build_variant.py (defines SNERF_PROBE_BUILD, writes gpurun_abl_<name>.so)"
#endif
#endif
