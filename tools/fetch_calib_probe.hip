// fetch_calib_probe.hip - known-byte streaming reads for calibrating rocprofv3's FETCH_SIZE on gfx950 in the access
// shapes libtangency uses (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports 1/2 of the bytes of a 16 B/lane stream;
// other widths are uncalibrated).  Three kernels read the same N-byte buffer (default 1 GiB, far beyond the 256 MiB
// Infinity Cache) exactly once:
//   read8_rows   8 B per lane, 16 lanes = one 128-byte row segment, rows 800 B apart (the staging loads of the
//                posterior kernels: 16 threads per row, 100-column panel)
//   read8_flat   8 B per lane, 64 consecutive lanes = 512 contiguous bytes
//   read16_flat  16 B per lane (the shape the guide calibrated)
// Build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib_probe.hip -o tools/bin/fetch_calib_probe
// Run:   rocprofv3 --pmc FETCH_SIZE -f csv -d <out> -- tools/bin/fetch_calib_probe   (prints the bytes each kernel read)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) read8_rows(const double* __restrict__ p, long long rows, int ld, double* out) {
    // 16 threads per row, each reads columns cb, cb+16, ..., cb+16*6 (7 x 8 B: 100 of 112 padded columns are real: clamp)
    const int cb = threadIdx.x & 15, rl = threadIdx.x >> 4;
    double s = 0.0;
    for (long long r = (long long)blockIdx.x * 16 + rl; r < rows; r += (long long)gridDim.x * 16) {
        const double* row = p + r * ld;
#pragma unroll
        for (int i = 0; i < 7; ++i) { const int c = cb + 16 * i; s += row[c < ld ? c : ld - 1]; }
    }
    if (s == 123.456) out[0] = s;
}
__global__ void __launch_bounds__(256) read8_flat(const double* __restrict__ p, long long n, double* out) {
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += p[i];
    if (s == 123.456) out[0] = s;
}
__global__ void __launch_bounds__(256) read16_flat(const double2* __restrict__ p, long long n, double* out) {
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) { const double2 v = p[i]; s += v.x + v.y; }
    if (s == 123.456) out[0] = s;
}

int main(int argc, char** argv) {
    const long long bytes = (argc > 1 ? atoll(argv[1]) : 1024LL) << 20;
    const int ld = 100;
    const long long n = bytes / 8, rows = n / ld;
    double *p = nullptr, *out = nullptr;
    CK(hipMalloc(&p, (size_t)bytes));
    CK(hipMalloc(&out, 8));
    CK(hipMemset(p, 0, (size_t)bytes));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(read8_rows, dim3(4096), dim3(256), 0, 0, p, rows, ld, out);
        hipLaunchKernelGGL(read8_flat, dim3(4096), dim3(256), 0, 0, p, n, out);
        hipLaunchKernelGGL(read16_flat, dim3(4096), dim3(256), 0, 0, (const double2*)p, n / 2, out);
    }
    CK(hipDeviceSynchronize());
    // read8_rows touches 100 of every 100 columns once (the clamped padding loads re-read column 99: same cache line)
    printf("bytes_read read8_rows %lld read8_flat %lld read16_flat %lld\n", rows * ld * 8, n * 8, (n / 2) * 16);
    return 0;
}
