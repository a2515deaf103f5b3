// posterior_wave2_nt.hip - one instantiation of the two- / four-wave-per-window kernel (tile count TP_NT).
// See posterior_wave2_impl.h.
#include "posterior_wave2_impl.h"

#ifndef TP_NT
#error "compile with -DTP_NT=<tiles per side>"
#endif
#define TP_CAT2(a, b) a##b
#define TP_CAT(a, b) TP_CAT2(a, b)

// wavefronts per window: two while a wave's share of the triangle (39 tiles at 12 per side) fits the register file
// next to the row pipeline, four above (30 tiles each at 15 per side)
#ifndef TP_WAVE2_NWV
#define TP_WAVE2_NWV (TP_NT <= 12 ? 2 : 4)
#endif

hipError_t TP_CAT(tp_wave2_launch_nt, TP_NT)(const tp_kargs_t& a, int grid, hipStream_t stream, tp_launch_info_t* info, bool lean) {
    constexpr int NWV = TP_WAVE2_NWV;
    return lean ? wave2_launch_variant<TP_NT, NWV, true>(a, grid, stream, info) : wave2_launch_variant<TP_NT, NWV, false>(a, grid, stream, info);
}
