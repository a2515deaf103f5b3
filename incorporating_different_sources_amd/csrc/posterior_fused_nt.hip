// posterior_fused_nt.hip - one instantiation of the fused kernel (tile count TP_NT), so that the
// tile counts compile in parallel.  See posterior_fused_impl.h.
#include "posterior_fused_impl.h"
#include <stdlib.h>

#ifndef TP_NT
#error "compile with -DTP_NT=<tiles per side>"
#endif
#define TP_CAT2(a, b) a##b
#define TP_CAT(a, b) TP_CAT2(a, b)

// tile counts that also have the one-wave-per-window kernel (posterior_wave_nt.hip; the Makefile's WAVE_NTS) and the
// two- / four-wave-per-window kernel (posterior_wave2_nt.hip; WAVE2_NTS)
#define TP_WAVE_NT_MIN 1
#define TP_WAVE_NT_MAX 9
#ifndef TP_WAVE2_NT_MIN
#define TP_WAVE2_NT_MIN 10      // (the Makefile's WAVE2_NTS; 7 in A/B builds)
#endif
#define TP_WAVE2_NT_MAX 15
#if TP_NT >= TP_WAVE_NT_MIN && TP_NT <= TP_WAVE_NT_MAX
hipError_t TP_CAT(tp_wave_launch_nt, TP_NT)(const tp_kargs_t&, int, hipStream_t, tp_launch_info_t*, bool);
#define TP_WAVE_FN TP_CAT(tp_wave_launch_nt, TP_NT)
#else
#define TP_WAVE_FN nullptr
#endif
#if TP_NT >= TP_WAVE2_NT_MIN && TP_NT <= TP_WAVE2_NT_MAX
hipError_t TP_CAT(tp_wave2_launch_nt, TP_NT)(const tp_kargs_t&, int, hipStream_t, tp_launch_info_t*, bool);
#define TP_WAVE2_FN TP_CAT(tp_wave2_launch_nt, TP_NT)
#else
#define TP_WAVE2_FN nullptr
#endif

hipError_t TP_CAT(tp_fused_launch_nt, TP_NT)(const tp_kargs_t& a, int grid, hipStream_t stream, tp_launch_info_t* info,
                                             int* want_occupancy) {
    constexpr int NW = tp_waves_for_tiles(TP_NT);
    if (want_occupancy) { *want_occupancy = blocks_per_cu<TP_NT, NW>(); return hipSuccess; }
    // which register-tile kernel: tp_kopts_t::wave_kernel 0 = the multi-wave kernel, 1 = one wave per window, 2 = two /
    // four waves per window (each where it is built), -1 = automatic (tp_pick_wave_kernel)
    const tp_wave_launch_fn one = TP_WAVE_FN, two = TP_WAVE2_FN;
    tp_wave_launch_fn wave = nullptr;
    switch (tp_pick_wave_kernel(TP_NT, a.opts.wave_kernel)) {
        case 1: wave = one; break;
        case 2: wave = two ? two : one; break;
        default: break;
    }
    return launch_one<TP_NT, NW>(a, grid, stream, info, wave);
}
