// Device-side ordering of a catalogue at upload: rocPRIM radix sorts (ROCm's own header library) wrapped for
// yawhip.hip. Replaces the per-job tree build of the reference (BinnedTrees.build, catalog/trees.py:483-545) and
// the host-thread sorts the first version of this library used (0.45 s per 10 M objects).
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "yawhip_sort.h"

namespace yawsort {

namespace {

__global__ void k_iota(uint32_t *p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (uint32_t)i;
}

// key = (run << 32) | position in `order`: unique keys, so the result does not depend on the sort being stable
__global__ void k_run_keys(const uint32_t *__restrict__ run, int64_t n, uint64_t *__restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = ((uint64_t)run[i] << 32) | (uint64_t)i;
}

__global__ void k_split_keys(const uint64_t *__restrict__ keys, int64_t n, uint32_t *__restrict__ run_sorted) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) run_sorted[i] = (uint32_t)(keys[i] >> 32);
}

inline unsigned grid_for(int64_t n) { return (unsigned)((n + 255) / 256); }

template <typename T>
hipError_t regrow(T *&p, size_t count) {
    if (p) (void)hipFree(p);
    p = nullptr;
    return hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
}

}  // namespace

hipError_t Workspace::reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    const size_t want = n + n / 8 + 1024;
    hipError_t e = regrow(keys_out, want);
    if (e == hipSuccess) e = regrow(k64_in, want);
    if (e == hipSuccess) e = regrow(k64_out, want);
    if (e == hipSuccess) e = regrow(iota, want);
    cap = e == hipSuccess ? want : 0;
    return e;
}

void Workspace::release() {
    if (tmp) (void)hipFree(tmp);
    if (keys_out) (void)hipFree(keys_out);
    if (k64_in) (void)hipFree(k64_in);
    if (k64_out) (void)hipFree(k64_out);
    if (iota) (void)hipFree(iota);
    *this = Workspace{};
}

static hipError_t reserve_tmp(Workspace &ws, size_t bytes) {
    if (bytes <= ws.tmp_bytes) return hipSuccess;
    if (ws.tmp) (void)hipFree(ws.tmp);
    ws.tmp = nullptr;
    ws.tmp_bytes = 0;
    const size_t want = bytes + bytes / 8 + 4096;
    hipError_t e = hipMalloc(&ws.tmp, want);
    if (e == hipSuccess) ws.tmp_bytes = want;
    return e;
}

hipError_t sort_segments(Workspace &ws, hipStream_t stream, int64_t n, const double *d_key, const int64_t *d_offsets,
                         int64_t n_segments, uint32_t *d_perm) {
    if (n <= 0) return hipSuccess;
    hipError_t e = ws.reserve((size_t)n);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_iota, dim3(grid_for(n)), dim3(256), 0, stream, ws.iota, n);
    size_t bytes = 0;
    e = rocprim::segmented_radix_sort_pairs(nullptr, bytes, d_key, ws.keys_out, ws.iota, d_perm, (unsigned int)n,
                                            (unsigned int)n_segments, d_offsets, d_offsets + 1, 0, 64, stream);
    if (e != hipSuccess) return e;
    e = reserve_tmp(ws, bytes);
    if (e != hipSuccess) return e;
    bytes = ws.tmp_bytes;
    return rocprim::segmented_radix_sort_pairs(ws.tmp, bytes, d_key, ws.keys_out, ws.iota, d_perm, (unsigned int)n,
                                               (unsigned int)n_segments, d_offsets, d_offsets + 1, 0, 64, stream);
}

hipError_t sort_runs(Workspace &ws, hipStream_t stream, int64_t n, const uint32_t *d_run, const uint32_t *d_order,
                     int run_bits, uint32_t *d_perm2, uint32_t *d_run_sorted) {
    if (n <= 0) return hipSuccess;
    hipError_t e = ws.reserve((size_t)n);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_run_keys, dim3(grid_for(n)), dim3(256), 0, stream, d_run, n, ws.k64_in);
    const unsigned end_bit = (unsigned)(32 + (run_bits < 1 ? 1 : run_bits));
    size_t bytes = 0;
    e = rocprim::radix_sort_pairs(nullptr, bytes, ws.k64_in, ws.k64_out, d_order, d_perm2, (size_t)n, 0, end_bit, stream);
    if (e != hipSuccess) return e;
    e = reserve_tmp(ws, bytes);
    if (e != hipSuccess) return e;
    bytes = ws.tmp_bytes;
    e = rocprim::radix_sort_pairs(ws.tmp, bytes, ws.k64_in, ws.k64_out, d_order, d_perm2, (size_t)n, 0, end_bit, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_split_keys, dim3(grid_for(n)), dim3(256), 0, stream, ws.k64_out, n, d_run_sorted);
    return hipGetLastError();
}

}  // namespace yawsort
