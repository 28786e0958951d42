// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths libyawhip uses:
// coalesced 8-byte-per-lane column reads (float64 SoA columns) and, for comparison, 16 bytes per lane.
//   hipcc -O3 --offload-arch=gfx950 -o tools/micro/fetch_calib tools/micro/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o run -- tools/micro/fetch_calib
// Each kernel reads exactly 1 GiB once; FETCH_SIZE (kB) / 2^20 is the fraction the counter reports.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void read8(const double *__restrict__ p, size_t n, double *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (; i < n; i += stride) acc += p[i];
    if (acc == 123.456) out[0] = acc;
}

__global__ void read16(const double2 *__restrict__ p, size_t n2, double *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (; i < n2; i += stride) { const double2 v = p[i]; acc += v.x + v.y; }
    if (acc == 123.456) out[0] = acc;
}

// 8-byte gathers: lane l of a wave reads element (base + perm(l)) of a 64-element block, blocks visited once
__global__ void gather8(const double *__restrict__ p, size_t n, double *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (; i < n; i += stride) {
        const size_t j = (i & ~(size_t)63) | ((i * 37) & 63);  // permutation inside the 512-byte block
        acc += p[j];
    }
    if (acc == 123.456) out[0] = acc;
}

int main() {
    const size_t n = (size_t)1 << 27;  // 2^27 doubles = 1 GiB
    double *p = nullptr, *out = nullptr;
    CHECK(hipMalloc(&p, n * sizeof(double)));
    CHECK(hipMalloc(&out, sizeof(double)));
    CHECK(hipMemset(p, 0, n * sizeof(double)));
    const dim3 grid(256 * 8), block(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read8, grid, block, 0, 0, p, n, out);
        hipLaunchKernelGGL(read16, grid, block, 0, 0, reinterpret_cast<const double2 *>(p), n / 2, out);
        hipLaunchKernelGGL(gather8, grid, block, 0, 0, p, n, out);
    }
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    printf("done: each kernel read %zu bytes\n", n * sizeof(double));
    CHECK(hipFree(p));
    CHECK(hipFree(out));
    return 0;
}
