// Micro-benchmark: FP32 dot-product pre-filter with the streamed side read through SCALAR loads
// (s_load_dwordx8 from a packed float4 record array) versus nothing else -- no LDS, no barriers.
// Build: hipcc -O3 --offload-arch=gfx950 -o sstream sstream.hip ; run: ./sstream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int R, int UNROLL>
__global__ __launch_bounds__(256) void k_sstream(const float4 *__restrict__ rec, int n_stream, const float *__restrict__ ax,
                                                 const float *__restrict__ ay, const float *__restrict__ az, int win,
                                                 unsigned long long *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    float fx[R], fy[R], fz[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = (wave_global * R + r) * 64 + lane;
        fx[r] = ax[i]; fy[r] = ay[i]; fz[r] = az[i];
    }
    // each wave streams its own window [b0, b0+win)
    const int b0 = __builtin_amdgcn_readfirstlane((wave_global * 37) % (n_stream - win));
    unsigned int hits = 0;
    for (int i = 0; i < win; i += UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const float4 b = rec[b0 + i + u];  // uniform address -> scalar load
            float best = -2.f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float d = __builtin_fmaf(fz[r], b.z, __builtin_fmaf(fy[r], b.y, fx[r] * b.x));
                best = fmaxf(best, d);
            }
            if (__builtin_amdgcn_ballot_w64(best >= b.w) != 0ull) hits += 1;
        }
    }
    if (lane == 0 && hits) atomicAdd(out, (unsigned long long)hits);
}

int main() {
    const int n_stream = 1 << 20, R = 4, win = 2048;
    const int n_waves = 256 * 4 * 5 * 8;  // 8 rounds of 5 waves/SIMD
    const int n_lane = n_waves * 8 * 64;  // sized for the largest R instantiated below (8)
    std::vector<float4> rec(n_stream);
    for (int i = 0; i < n_stream; ++i) rec[i] = make_float4(drand48(), drand48(), drand48(), 5.0f);  // never passes
    std::vector<float> a(n_lane);
    for (auto &v : a) v = drand48();
    float4 *d_rec; float *d_a; unsigned long long *d_out;
    CK(hipMalloc(&d_rec, n_stream * sizeof(float4))); CK(hipMalloc(&d_a, n_lane * sizeof(float))); CK(hipMalloc(&d_out, 8));
    CK(hipMemcpy(d_rec, rec.data(), n_stream * sizeof(float4), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_a, a.data(), n_lane * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemset(d_out, 0, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](auto kern, const char *name, int threads, int rr) {
        const int blocks = n_waves / (threads / 64);
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_rec, n_stream, d_a, d_a, d_a, win, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double pairs = (double)n_waves * rr * 64 * win;
            if (rep == 2) printf("%-28s %8.3f ms  %.3e pairs/s\n", name, ms, pairs / ms * 1e3);
        }
    };
    run(k_sstream<4, 1>, "scalar R=4 unroll1 wg256", 256, 4);
    run(k_sstream<4, 2>, "scalar R=4 unroll2 wg256", 256, 4);
    run(k_sstream<4, 4>, "scalar R=4 unroll4 wg256", 256, 4);
    run(k_sstream<4, 4>, "scalar R=4 unroll4 wg64", 64, 4);
    run(k_sstream<8, 2>, "scalar R=8 unroll2 wg256", 256, 8);
    run(k_sstream<2, 4>, "scalar R=2 unroll4 wg256", 256, 2);
    return 0;
}
