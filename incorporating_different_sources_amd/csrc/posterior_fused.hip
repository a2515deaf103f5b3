// posterior_fused.hip - dispatcher of the register-tile fused kernel over the tile count
// NT = ceil((k+1)/16); the kernels themselves are instantiated in posterior_fused_nt.hip.
#include "posterior_kernels.h"

#define TP_MAX_NT 15   // NT = 16 would need 163 KiB of LDS (two 32-row staging buffers of 272 doubles)

#define TP_DECL(NT) hipError_t tp_fused_launch_nt##NT(const tp_kargs_t&, int, hipStream_t, tp_launch_info_t*, int*);
TP_DECL(1) TP_DECL(2) TP_DECL(3) TP_DECL(4) TP_DECL(5) TP_DECL(6) TP_DECL(7) TP_DECL(8)
TP_DECL(9) TP_DECL(10) TP_DECL(11) TP_DECL(12) TP_DECL(13) TP_DECL(14) TP_DECL(15)

int tp_fused_max_assets(void) { return 16 * TP_MAX_NT - 1; }

hipError_t tp_fused_launch(const tp_kargs_t& a, int grid, hipStream_t stream, tp_launch_info_t* info,
                           int* want_occupancy) {
    const int nt = (a.k + 1 + 15) / 16;
    switch (nt) {
#define TP_CASE(NT) case NT: return tp_fused_launch_nt##NT(a, grid, stream, info, want_occupancy);
        TP_CASE(1) TP_CASE(2) TP_CASE(3) TP_CASE(4) TP_CASE(5) TP_CASE(6) TP_CASE(7) TP_CASE(8)
        TP_CASE(9) TP_CASE(10) TP_CASE(11) TP_CASE(12) TP_CASE(13) TP_CASE(14) TP_CASE(15)
        default: return hipErrorInvalidValue;
    }
}
