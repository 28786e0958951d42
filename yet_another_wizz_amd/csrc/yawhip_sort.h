// Internal interface between yawhip.hip and yawhip_sort.hip (device-side ordering of a catalogue at upload).
// Not part of the C ABI.
#ifndef YAWHIP_SORT_H
#define YAWHIP_SORT_H
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace yawsort {

// Workspace of the sorts (grow-only, owned by the caller's context).
struct Workspace {
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    double *keys_out = nullptr;     // [cap]
    uint64_t *k64_in = nullptr;     // [cap]
    uint64_t *k64_out = nullptr;    // [cap]
    uint32_t *iota = nullptr;       // [cap]
    size_t cap = 0;
    hipError_t reserve(size_t n);
    void release();
};

// perm[i] = index (into the input order) of the object that comes i-th when every segment
// [offsets[s], offsets[s+1]) is ordered by ascending key. Equal keys: deterministic, unspecified.
hipError_t sort_segments(Workspace &ws, hipStream_t stream, int64_t n, const double *d_key, const int64_t *d_offsets,
                         int64_t n_segments, uint32_t *d_perm);

// The strip layout: objects grouped by run id, inside a run in the order given by `order` (a permutation that is
// already sorted along the sort axis inside every patch). d_run[i] = run of the object order[i].
// Result: perm2[j] = input index of the object at position j of the strip layout; run_sorted[j] = its run.
hipError_t sort_runs(Workspace &ws, hipStream_t stream, int64_t n, const uint32_t *d_run, const uint32_t *d_order,
                     int run_bits, uint32_t *d_perm2, uint32_t *d_run_sorted);

}  // namespace yawsort
#endif
