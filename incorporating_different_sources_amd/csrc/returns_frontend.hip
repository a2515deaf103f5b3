// returns_frontend.hip - log-return panels from price panels on the device (gfx950).
//
// The reference recomputes log(P_t / P_{t-1}) per window on the host (ref:31-62 daily / resampled prices,
// ref:299-314 intraday bars; ref:LINE = /root/reference/src/portfolio_calculations.py).  Here a backtest
// uploads each PRICE panel once and this kernel forms the return panel all windows index into:
//     R[i][c] = log(P[num[i]][c] / P[den[i]][c]),      NaN -> 0, +-inf -> +-DBL_MAX
// with (num, den) = (i, i-1) for consecutive rows, (bin end, previous bin end) for weekly / monthly
// resampled windows, (date, last complete bin end) for the running bin.  Elementwise and HBM-bound:
// 16 B read + 8 B written per element.
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "posterior_kernels.h"

namespace {

__device__ __forceinline__ double log_return(double p, double q) {
    double r = log(p / q);                              // ref:44 np.log(prices / prices.shift(1))
    if (r != r) r = 0.0;                                // the packer's nan_to_num: NaN -> 0,
    else if (r > DBL_MAX) r = DBL_MAX;                  // +inf -> largest finite,
    else if (r < -DBL_MAX) r = -DBL_MAX;                // -inf -> most negative finite
    return r;
}

// One output row per blockIdx.y (grid-stride), 512 consecutive columns per workgroup: no index division, and
// 16-byte accesses (two columns per lane) when the rows are 16-byte aligned (even ld).
template <bool VEC2>
__global__ void __launch_bounds__(256) log_return_rows_kernel(const double* __restrict__ prices, int ld,
                                                              const int* __restrict__ num,
                                                              const int* __restrict__ den, long long n_out,
                                                              double* __restrict__ out) {
    const int c = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (c >= ld) return;
    for (long long i = blockIdx.y; i < n_out; i += gridDim.y) {
        const double* pn = prices + (long long)num[i] * ld + c;
        const double* pd = prices + (long long)den[i] * ld + c;
        double* po = out + i * (long long)ld + c;
        if (VEC2) {
            const double2 p = *(const double2*)pn, q = *(const double2*)pd;
            *(double2*)po = double2{log_return(p.x, q.x), log_return(p.y, q.y)};
        } else {
            po[0] = log_return(pn[0], pd[0]);
            if (c + 1 < ld) po[1] = log_return(pn[1], pd[1]);
        }
    }
}

}  // namespace

hipError_t tp_log_return_rows_launch(const double* prices, int ld, const int* num, const int* den, long long n_out,
                                     double* out, hipStream_t stream) {
    if (n_out <= 0 || ld <= 0) return hipSuccess;
    const dim3 grid((unsigned)((ld + 511) / 512), (unsigned)(n_out < 32768 ? n_out : 32768));
    const bool vec2 = (ld % 2 == 0) && ((uintptr_t)prices % 16 == 0) && ((uintptr_t)out % 16 == 0);
    if (vec2) hipLaunchKernelGGL(log_return_rows_kernel<true>, grid, dim3(256), 0, stream, prices, ld, num, den, n_out, out);
    else hipLaunchKernelGGL(log_return_rows_kernel<false>, grid, dim3(256), 0, stream, prices, ld, num, den, n_out, out);
    return hipGetLastError();
}
