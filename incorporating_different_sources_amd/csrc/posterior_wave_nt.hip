// posterior_wave_nt.hip - one instantiation of the one-wave-per-window kernel (tile count TP_NT).
// See posterior_wave_impl.h.
#include "posterior_wave_impl.h"

#ifndef TP_NT
#error "compile with -DTP_NT=<tiles per side>"
#endif
#define TP_CAT2(a, b) a##b
#define TP_CAT(a, b) TP_CAT2(a, b)

hipError_t TP_CAT(tp_wave_launch_nt, TP_NT)(const tp_kargs_t& a, int grid, hipStream_t stream, tp_launch_info_t* info, bool lean) {
    return lean ? wave_launch_variant<TP_NT, true>(a, grid, stream, info) : wave_launch_variant<TP_NT, false>(a, grid, stream, info);
}
