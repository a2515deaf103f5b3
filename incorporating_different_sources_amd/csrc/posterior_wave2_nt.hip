// posterior_wave2_nt.hip - one instantiation of the two- / four-wave-per-window kernel: tile count TP_NT, layout TP_LEAN
// (1 = contiguous rolling windows, 0 = general / index layout; two translation units per tile count so that the build
// parallelises - the four-wave kernels of 13..15 tiles per side are the longest compiles of the library).
// See posterior_wave2_impl.h.
#include "posterior_wave2_impl.h"

#if !defined(TP_NT) || !defined(TP_LEAN)
#error "compile with -DTP_NT=<tiles per side> -DTP_LEAN=<0|1>"
#endif
#define TP_CAT2(a, b) a##b
#define TP_CAT(a, b) TP_CAT2(a, b)

// wavefronts per window: two while a wave's share of the triangle (39 tiles at 12 per side) fits the register file
// next to the row pipeline, four above (30 tiles each at 15 per side)
#ifndef TP_WAVE2_NWV
#define TP_WAVE2_NWV (TP_NT <= 12 ? 2 : 4)
#endif

#if TP_LEAN
hipError_t TP_CAT(tp_wave2_general_launch_nt, TP_NT)(const tp_kargs_t&, int, hipStream_t, tp_launch_info_t*);

hipError_t TP_CAT(tp_wave2_launch_nt, TP_NT)(const tp_kargs_t& a, int grid, hipStream_t stream, tp_launch_info_t* info, bool lean) {
    constexpr int NWV = TP_WAVE2_NWV;
    return lean ? wave2_launch_variant<TP_NT, NWV, true>(a, grid, stream, info) : TP_CAT(tp_wave2_general_launch_nt, TP_NT)(a, grid, stream, info);
}
#else
hipError_t TP_CAT(tp_wave2_general_launch_nt, TP_NT)(const tp_kargs_t& a, int grid, hipStream_t stream, tp_launch_info_t* info) {
    return wave2_launch_variant<TP_NT, TP_WAVE2_NWV, false>(a, grid, stream, info);
}
#endif
