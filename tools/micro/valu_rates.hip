// valu_rates.hip -- issue cost of the vector instructions the band walk is made of, on one MI355X.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rates.hip -o tools/micro/valu_rates && tools/micro/valu_rates
// Every kernel runs ITER iterations of UNROLL independent copies of one instruction per wave; the grid puts W waves
// on every SIMD of every CU (W = 1, 2, 4, 8). Reported: cycles per wave-instruction per SIMD at the measured clock
// (s_memrealtime is 100 MHz; the shader clock comes from hipDeviceProp). Development aid, not part of the package.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                           \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

constexpr int ITER = 4096;

// 8 independent chains per instruction kind, registers chosen by the compiler through constraints
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(double *out, float seedf, double seedd) {
    double d[8], e = seedd + threadIdx.x * 1e-9;
    float f[8], g = seedf + threadIdx.x * 1e-6f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8], q = {g, g + 1.f};
    unsigned u[8];
    unsigned long long m = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = seedd * (i + 1); f[i] = seedf * (i + 1); p[i] = f2{f[i], f[i] + 0.5f}; u[i] = (unsigned)(threadIdx.x * (i + 1)); }
    __shared__ double lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = seedd * i;
    __syncthreads();
    unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) double *)lds + (threadIdx.x & 63) * 8;
    for (int it = 0; it < ITER; ++it) {
        asm volatile("" : "+v"(e), "+v"(g), "+v"(q));  // opaque per iteration: nothing below is loop invariant
        if (KIND == 0) {
#define S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e));
            REP8(S)
#undef S
        } else if (KIND == 1) {
#define S(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e));
            REP8(S)
#undef S
        } else if (KIND == 2) {
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(e));
            REP8(S)
#undef S
        } else if (KIND == 3) {
#define S(i) asm volatile("v_cmp_gt_f64 vcc, %1, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(u[i]) : "v"(d[i]), "v"(e) : "vcc");
            REP8(S)
#undef S
        } else if (KIND == 4) {
#define S(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g));
            REP8(S)
#undef S
        } else if (KIND == 5) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(g));
            REP8(S)
#undef S
        } else if (KIND == 6) {
#define S(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(q));
            REP8(S)
#undef S
        } else if (KIND == 7) {
#define S(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q));
            REP8(S)
#undef S
        } else if (KIND == 8) {
#define S(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q));
            REP8(S)
#undef S
        } else if (KIND == 9) {
#define S(i) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(u[i]) : "v"(f[i]), "v"(g) : "vcc");
            REP8(S)
#undef S
        } else if (KIND == 10) {
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            REP8(S)
#undef S
        } else if (KIND == 11) {
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 7]) : "vcc");
            REP8(S)
#undef S
        } else if (KIND == 12) {  // the exact predicate as the band walk issues it: 3 sub, 3 mul, 2 add, 2 cmp, s_and, addc
#define S(i)                                                                                                          \
    {                                                                                                                 \
        double dx = d[i] - e, dy = d[(i + 1) & 7] - e, dz = d[(i + 2) & 7] - e;                                       \
        double s = (dx * dx + dy * dy) + dz * dz;                                                                     \
        u[i] += (s > seedd) & (s <= d[(i + 3) & 7]);                                                                  \
    }
            REP8(S)
#undef S
        } else if (KIND == 13) {  // the same in float32, scalar instructions
#define S(i)                                                                                                          \
    {                                                                                                                 \
        float dx = f[i] - g, dy = f[(i + 1) & 7] - g, dz = f[(i + 2) & 7] - g;                                        \
        float s = fmaf(dz, dz, fmaf(dy, dy, dx * dx));                                                                \
        u[i] += (s > seedf) & (s <= f[(i + 3) & 7]);                                                                  \
    }
            REP8(S)
#undef S
        } else if (KIND == 18) {  // float32, two evaluations per packed instruction
#define S(i)                                                                                                          \
    {                                                                                                                 \
        f2 dx = p[i] - q.x, dy = p[(i + 1) & 7] - q.y, dz = p[(i + 2) & 7] - q.x;                                     \
        f2 s = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));                         \
        u[i] += (s.x > seedf) & (s.x <= f[(i + 3) & 7]);                                                              \
        u[(i + 4) & 7] += (s.y > seedf) & (s.y <= f[(i + 3) & 7]);                                                    \
    }
            REP8(S)
#undef S
        } else if (KIND == 19) {  // float32 packed, classification by |s - c| against two half widths
#define S(i)                                                                                                          \
    {                                                                                                                 \
        f2 dx = p[i] - q.x, dy = p[(i + 1) & 7] - q.y, dz = p[(i + 2) & 7] - q.x;                                     \
        f2 s = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx)) - q.y;                   \
        u[i] += __builtin_fabsf(s.x) < seedf;                                                                         \
        u[(i + 4) & 7] += __builtin_fabsf(s.y) < seedf;                                                               \
        m |= __builtin_amdgcn_ballot_w64((__builtin_fabsf(s.x) < g) | (__builtin_fabsf(s.y) < g));                    \
    }
            REP8(S)
#undef S
        } else if (KIND == 14) {
#define S(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[i]) : "v"(la), "n"(i * 520));
            REP8(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 15) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 r[8];
#define S(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"(la * 2), "n"(i * 1040));
            REP8(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] += r[i].x;
        } else if (KIND == 16) {
#define S(i) asm volatile("v_sub_f32 %0, %0, %1\n v_cmp_lt_f32 vcc, |%0|, %1" : "+v"(f[i]) : "v"(g) : "vcc");
            REP8(S)
#undef S
        } else if (KIND == 17) {
#define S(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
            REP8(S)
#undef S
        }
    }
    double acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += d[i] + f[i] + p[i].x + p[i].y + u[i];
    if (acc == 1.2345 || m == 77ull) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int KIND>
void run(const char *name, int per_iter, double *out, int n_cu, double ghz) {
    printf("%-34s", name);
    for (int w : {1, 2, 4, 8}) {  // waves per SIMD: a 256-thread workgroup = one wave on each SIMD of a CU
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a));
        CHECK(hipEventCreate(&b));
        hipLaunchKernelGGL(k_rate<KIND>, dim3(n_cu * w), dim3(256), 0, 0, out, 1.0001f, 1.0000001);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k_rate<KIND>, dim3(n_cu * w), dim3(256), 0, 0, out, 1.0001f, 1.0000001);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        const double cyc = ms * 1e-3 * ghz * 1e9;                // cycles of the launch
        const double inst = (double)ITER * per_iter * w;         // wave-instructions per SIMD
        printf("  W=%d: %6.2f cyc/inst", w, cyc / inst);
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz (cycles below assume this clock)\n", prop.name, n_cu, ghz);
    double *out;
    CHECK(hipMalloc(&out, sizeof(double) * 256 * n_cu * 8));
    run<0>("v_add_f64", 8, out, n_cu, ghz);
    run<1>("v_mul_f64", 8, out, n_cu, ghz);
    run<2>("v_fma_f64", 8, out, n_cu, ghz);
    run<3>("v_cmp_gt_f64 + v_addc", 16, out, n_cu, ghz);
    run<4>("v_add_f32", 8, out, n_cu, ghz);
    run<5>("v_fma_f32", 8, out, n_cu, ghz);
    run<6>("v_pk_fma_f32", 8, out, n_cu, ghz);
    run<7>("v_pk_add_f32", 8, out, n_cu, ghz);
    run<8>("v_pk_mul_f32", 8, out, n_cu, ghz);
    run<9>("v_cmp_gt_f32 + v_addc", 16, out, n_cu, ghz);
    run<10>("v_add_u32", 8, out, n_cu, ghz);
    run<11>("v_cndmask_b32", 8, out, n_cu, ghz);
    run<12>("f64 predicate (per evaluation)", 8, out, n_cu, ghz);
    run<13>("f32 predicate (per evaluation)", 8, out, n_cu, ghz);
    run<14>("ds_read_b64 (+wait per 8)", 8, out, n_cu, ghz);
    run<15>("ds_read_b128 (+wait per 8)", 8, out, n_cu, ghz);
    run<16>("v_sub_f32 + v_cmp_lt_f32 |x|", 16, out, n_cu, ghz);
    run<17>("v_cvt_f32_f64", 8, out, n_cu, ghz);
    run<18>("f32 packed predicate (per 2 evals)", 8, out, n_cu, ghz);
    run<19>("f32 packed |s-c| classes (per 2)", 8, out, n_cu, ghz);
    return 0;
}
