// yawhip.hip -- MI355X (gfx950 / CDNA4) angular pair counting behind the C ABI of include/yawhip.h.
//
// Replaces the per-job loop of PatchLinkage.count_pairs (reference src/yaw/correlation/measurements.py:344-364):
// for each linked patch pair (p,q) and redshift bin k it counts, per fine angular bin e, the object
// pairs with  t[k][e] < s <= t[k][e+1],  s = ((ax-bx)^2 + (ay-by)^2) + (az-bz)^2  in float64 without
// FMA -- the predicate scipy's KDTree.count_neighbors applies behind AngularTree.count
// (src/yaw/catalog/trees.py:303-362).
//
// Design (wave64, no MFMA -- K=3 distances are not a contraction and bit parity forbids replacing the
// predicate by a dot-product form; DESIGN.md section 4 has the details and the measurements):
//   * catalogues live in HBM as SoA float64 columns x,y,z,(w) in two library-private orders, both made on the
//     device at upload (yawhip_sort.hip): (patch, z-bin, u) with a CSR offset table, u = the sort axis; and the
//     strip layout (patch, strip, u) -- strips of a global grid along a second axis, all bins together, bin id
//     per object -- whose runs can be paired across catalogues by grid index alone;
//   * k_build_items / k_build_items_strips turn the job table into work items (lane tile of c2) x (window of a
//     c1 segment or run): one thread per potential item decodes it arithmetically, binary-searches the window
//     |du| <= sqrt(t_max) and drops empty ones; one atomic per workgroup appends the rest -- for the float32 band
//     kernels into one of eight segments of the list, one per XCD (append_items);
//   * k_count (EXACT / FILTER, non-unit input): 256-thread workgroups, 256*R lane objects in registers, the
//     c1 segment streamed through LDS; 8 FP64 ops + compare per pair, or a conservative FP32 dot-product test
//     first and exact FP64 for its survivors. Per-lane private LDS histograms, fixed-order reduction;
//   * k_count_merged (SWEEP): single-wave workgroups, float32 only on chip (packed v_pk_fma_f32),
//     survivors queued per wave and evaluated 64 at a time in exact FP64; one item of the cross-correlation
//     path covers all redshift bins. AUTO uses it on layouts without strips;
//   * k_count_band32 / k_count_band32_one (BAND, what AUTO runs on strip layouts of unit vectors -- the headline;
//     csrc/yawhip_band32.inc): single-wave workgroups, the window of a lane tile staged in LDS by LDS-DMA from float32
//     images of the columns, every lane walks only the band |du| <= r of its objects, classifies every entry in
//     float32 against guard bands around the edges and decides the few inside a guard band with the exact FP64
//     predicate on the float64 columns (results identical to an all-float64 evaluation). The streamed side is read
//     from merged runs of three neighbouring strips (k_merge_triples) where such a window fits the stage;
//     k_count_band32_fine: the same for fine radial grids (separation weights); k_count_band: every entry in FP64.
// Unweighted counts: uint32 LDS histograms -> 64-bit integer atomics. Weighted sums: per-item slabs (LDS float64
// atomics private to one wave) reduced in a fixed two-level order -> bit-reproducible run to run. No floating
// point atomics in global memory.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off (see yet_another_wizz_amd/build.py).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <numeric>
#include <thread>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <system_error>
#include <utility>
#include <functional>
#include <vector>

#include "yawhip.h"
#include "yawhip_sort.h"

namespace {
// YAWHIP_TRACE=1: wall-clock marks of a call's host side, printed to stderr when the call returns (diagnostics; two
// clock reads per mark when off)
struct Trace {
    bool on = getenv("YAWHIP_TRACE") != nullptr;
    int n = 0;
    const char *name[32];
    std::chrono::steady_clock::time_point at[32];
    void mark(const char *what) {
        if (on && n < 32) { name[n] = what; at[n++] = std::chrono::steady_clock::now(); }
    }
    void flush() {
        if (on && n > 1) {
            fprintf(stderr, "[yawhip trace]");
            for (int i = 1; i < n; ++i)
                fprintf(stderr, " %s +%.1f", name[i], std::chrono::duration<double, std::micro>(at[i] - at[i - 1]).count());
            fprintf(stderr, " | total %.1f us\n", std::chrono::duration<double, std::micro>(at[n - 1] - at[0]).count());
        }
        n = 0;
    }
};
thread_local Trace g_trace;


constexpr int WG = 256;      // threads per workgroup = 4 waves of 64
constexpr int STAGE = 256;   // streamed objects per LDS stage (one per thread)
#ifndef YAW_MSTAGE
#define YAW_MSTAGE 64
#endif
#ifndef YAW_MWG
#define YAW_MWG 64
#endif
constexpr int MWG = YAW_MWG;        // threads per workgroup of the lean kernel (k_count_merged)
static_assert(MWG == 64, "the band kernels are single-wave workgroups: their lane tile is 64 * R objects");
constexpr int MSTAGE = YAW_MSTAGE;  // stage of the merged path: smaller -> less LDS -> more workgroups per CU
constexpr int MAX_EDGES = 512;
constexpr int SEG_STRIPS_MIN_RUN = 16;  // mean objects per (patch, bin, strip) run of the lane side from which mode 3 is used
constexpr int BAND_MIN_STREAM_RUN = 64;  // AUTO: objects per run of the streamed side (as the typical object sees it) from which the band kernel is used
constexpr int64_t SYNC_GRID_MIN_ITEMS = 400000;  // potential items from which the count grid is sized exactly (one host sync)
constexpr int MAX_STRIP_REACH = 12;  // strip pairing is used while sqrt(t_max) <= 12 grid spacings
constexpr int COUNT_FLUSH_MASK = (1 << 13) - 1;  // k_count: stages between flushes of the 32-bit LDS counters (see there)
constexpr int MERGED_FLUSH_MASK = (1 << 16) - 1; // k_count_merged: 256 lane objects x 64 streamed objects per stage
constexpr int SPLIT_JOBS = 1;  // internal status of count_enqueue: nothing was enqueued, the caller must split the job list
constexpr double PAD_COORD = 4.0;  // padded lanes sit >= 3 away from any unit vector: s >= 9 > max t = 4

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? YAWHIP_ERR_OOM : YAWHIP_ERR_HIP, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                         \
    } while (0)

struct CatView {
    const double *x, *y, *z, *w;  // w may be null
    const int64_t *off;           // [P*nb+1]
    int nb;
    const double *key;            // the column the segments are sorted by (x, y or z)
    int axis;                     // 0, 1, 2
};

struct alignas(16) Obj {  // one streamed object in LDS: two 16-byte broadcast reads
    double x, y, z, w;
};

struct alignas(16) ObjF {  // its float32 image for the pre-filter: one 16-byte broadcast read
    float x, y, z, pad;
};

// Pre-filter guard (see k_count): |dot32 - a.b| <= 5.000001 u for unit vectors rounded to float32 and
// a mul + 2 fma evaluation (u = 2^-24); 8 u leaves room for |a|^2 deviating from 1 by < 1e-9 and for
// the rounding of the threshold itself.
constexpr double FILTER_GUARD = 8.0 * 5.9604644775390625e-8;
constexpr double UNIT_NORM_TOL = 1e-9;

constexpr int MAX_WIN = 3;  // windows (partner runs of c1) one work item can carry
struct alignas(16) Item {  // one unit of work for a workgroup: a lane tile of c2 and up to MAX_WIN windows of c1
    int64_t a0;    // first lane object (c2 side)
    int32_t na;    // lane objects (<= 256*R, <= 64*R on the SWEEP / BAND paths)
    int32_t slot;  // output slot: job * n_bins + bin, or the job itself on the strip path (bits 0..29);
                   // bits 30..31: orientation = which of the catalogues' three strip layouts a0 / b0 index
    int32_t pot;   // index among all potential items (slab index of weighted partial sums)
    int32_t nwin;  // windows in use (>= 1 for a kept item)
    int64_t b0[MAX_WIN];  // first streamed object of every window (c1 side)
    int32_t nb[MAX_WIN];  // streamed objects of every window
    int32_t pad_;
};
static_assert(sizeof(Item) == 64, "Item layout");
constexpr int SLOT_MASK = 0x3fffffff;
__host__ __device__ inline int item_slot(const Item &it) { return it.slot & SLOT_MASK; }
__host__ __device__ inline int item_orient(const Item &it) { return (int)((unsigned)it.slot >> 30); }

// One layout of one catalogue as the kernels see it. The count kernels and the strip builder receive a table of six:
// [o] = layout of c1 for orientation o, [3 + o] = layout of c2 (plain layouts: entry 0 / 3 only).
// Orientation o = the sort axis u of the layout (0 = x, 1 = y, 2 = z); strips are cut along v = (o + 2) % 3 and the
// third axis w = (o + 1) % 3 is the one the projection drops: a job uses the orientation whose w points towards its
// two patches, where the (u, v) projection of the sphere is least compressed.
// (pointers carry the global address space: loaded from a table the compiler could not tell, and would use flat loads)
typedef const __attribute__((address_space(1))) double *gf64p;
typedef const __attribute__((address_space(1))) int32_t *gi32p;
typedef const __attribute__((address_space(1))) int64_t *gi64p;
#ifndef YAW_B32_PREFETCH
#define YAW_B32_PREFETCH 0  // next trip's LDS reads issued before this trip's arithmetic: 0.404 against 0.379 ms (registers -> 5 waves)
#endif
#ifndef YAW_B32_UNROLL
#define YAW_B32_UNROLL 1  // entries per trip of the walk loop: 1, 2 and 4 measure the same (0.362 / 0.366 / 0.374 ms at the headline)
#endif
#ifndef YAW_B32_PAIRS
#define YAW_B32_PAIRS 1  // k_count_band32_one with one object per lane evaluates two entries per trip (packed float32)
#endif
#ifndef YAW_B32_SHARE
#define YAW_B32_SHARE 1  // bands of sparse single-window items are shared out over the wave (k_count_band32)
#endif
#define YAW_STR_(x) #x
#define YAW_STR(x) YAW_STR_(x)
#ifndef YAW_B32_ITEM_PREFETCH
#define YAW_B32_ITEM_PREFETCH 0  // the record of a workgroup's next item is fetched while it counts the present one
#endif
#ifndef YAW_B32_AW_EARLY
#define YAW_B32_AW_EARLY 1  // weighted: a lane object's own weight is loaded with its coordinates instead of at the flush, the end of the
                            // item's chain of dependent memory latencies (config #4: DD 0.370 -> 0.349, DR 1.67 -> 1.56, RR 3.20 -> 3.13 ms)
#endif
#ifndef YAW_B32_WAVES_W
#define YAW_B32_WAVES_W 5  // waves per SIMD the weighted one-annulus variants are compiled for (96 VGPRs; the compiler took 97 by itself: 4 waves, 0.57 against 0.51 ms)
#endif
#ifndef YAW_B32_WAVES_W1
#define YAW_B32_WAVES_W1 5  // ... with one object per lane (DD / RR of an autocorrelation: 82 VGPRs)
#endif
#ifndef YAW_B32_WAVES_BIG
#define YAW_B32_WAVES_BIG 6  // ... the plain count with the big stage: its 6.3 KB of LDS admit 25 workgroups per CU anyway, and at 80 registers
#endif                       // nothing is spilled (headline 0.305 -> 0.286 ms; 8: 0.343)
#ifndef YAW_B32_WAVES_LT
#define YAW_B32_WAVES_LT 1   // ... the plain count with per-bin thresholds (physical scales): the compiler's choice (87 registers, 5 waves)
#endif
#ifndef YAW_B32_WAVES
#define YAW_B32_WAVES 1  // > 1: waves per SIMD every variant is compiled for (experiments). Default: 7 for the plain count (72 VGPRs
#endif                   // instead of 79: 0.359 against 0.371 ms at the headline; 8 spills: 0.405), the compiler's choice elsewhere
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) float *gf32p;
struct DevTab {
    gf64p x, y, z, w;          // columns; w may be null
    gi32p k;                   // bin id per object (merged cross-correlation layouts), else null
    gi64p off;                 // run offsets [V+1] (strip layouts) or segment offsets
    gi64p vbase, slo, tiles;   // strip layouts: first run of a group, its grid index, lane-tile prefix over runs
    const struct TileRec *tile_rec;  // strip layouts: first object, length and run of every lane tile
    const struct RunGrid *grid;      // strip layouts: per-run index along the sort axis (item builder)
    gf32p qx, qy, qz;          // strip layouts: float32 images of the columns (k_count_band32)
    gi32p idx;                 // merged triple runs (streamed side): index of an entry in the layout's own order, else null
    gi32p pos3;                // lane side of a self count on merged triple runs: place of an object in its own strip's triple, else null
    int32_t axis;              // sort axis inside a run / segment
    int32_t pad_;
};
__device__ __forceinline__ gf64p tab_key(const DevTab &t) { return t.axis == 0 ? t.x : (t.axis == 1 ? t.y : t.z); }

// Tables of the strip item builder. Every thread of the builder walks a chain of dependent loads and the chain's length
// is the kernel's run time (0.06 of the 0.55 ms of a headline call), so what the host or the layout build can precompute
// travels as one record per job, per lane tile and per run instead of being looked up table by table.
struct TileRec {   // per lane tile of a layout (one table per tile size)
    int64_t a0;    // first object
    int32_t na;    // objects (<= tile)
    int32_t run;   // run the tile belongs to
};
struct JobRec {    // per job of a call
    int64_t t_lo;      // first lane tile of the job (absolute index into the lane side's tile table)
    int64_t k_off;     // strip of the streamed group facing lane run r2 under neighbour offset d: r2 + k_off + d
    int64_t vbase1;    // first run of the streamed group
    int32_t n_strips1; // runs of the streamed group
    int32_t o;         // orientation: which pair of layouts the job runs on
};
// Per-run index along the sort axis: the key range [first, last] of a run is cut into RUN_GRID cells by
// cell(key) = clamp(floor((key - first) * inv), 0, RUN_GRID - 1), and g[c] = number of entries whose cell is < c
// (g[0] = 0, g[RUN_GRID] = run length). cell() is monotone in the key and evaluated by the same instructions when the
// table is built and when it is queried, so for any w the first entry with key >= w and the first with key > w both lie in
// [g[cell(w)], g[cell(w) + 1]] -- exactly, whatever the rounding of the product: the bisection over a run of 900 entries
// (ten dependent loads) becomes one table look-up and four steps.
#ifndef YAW_RUN_GRID
#define YAW_RUN_GRID 64
#endif
#ifndef YAW_BUILD_BISECT
#define YAW_BUILD_BISECT 32  // bisection steps of the strip builder inside a grid cell (fewer: the window is a superset, a few entries longer)
#endif
constexpr int RUN_GRID = YAW_RUN_GRID;
struct RunGrid {
    double inv;                  // RUN_GRID / (last - first), 0 for a run with one distinct key
    uint32_t g[RUN_GRID + 2];    // + 1 pad: 8-byte multiple
};
__device__ __forceinline__ int run_cell(double key, double first, double inv) {
    const double f = (key - first) * inv;
    return f >= (double)RUN_GRID ? RUN_GRID - 1 : (f >= 1.0 ? (int)f : 0);
}
// one thread per (run, cell boundary): g[c] by bisection with the predicate cell(key) < c
template <typename KeyT>
__global__ __launch_bounds__(256) void k_run_grid(int64_t n_runs, const int64_t *__restrict__ off, const KeyT *__restrict__ key,
                                                  RunGrid *__restrict__ grid) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = i / (RUN_GRID + 1);
    const int c = (int)(i - r * (RUN_GRID + 1));
    if (r >= n_runs) return;
    const int64_t b0 = off[r], b1 = off[r + 1];
    double inv = 0.0, first = 0.0;
    if (b1 > b0) {
        first = (double)key[b0];
        const double span = (double)key[b1 - 1] - first;
        inv = span > 0.0 ? (double)RUN_GRID / span : 0.0;
        if (!(inv < 1e300)) inv = 0.0;  // a denormal span: one cell
    }
    int64_t l = b0, h = b1;
    while (l < h) {
        const int64_t m = (l + h) >> 1;
        if (run_cell((double)key[m], first, inv) < c) l = m + 1; else h = m;
    }
    grid[r].g[c] = (uint32_t)(l - b0);
    if (c == 0) { grid[r].inv = inv; grid[r].g[RUN_GRID + 1] = 0; }
}

// Merged triple runs of a strip layout (streamed side of the float32 band kernels). With a grid as wide as the largest
// separation the partners of a lane tile in strip c are the strips c - 1, c, c + 1 of the other patch: three windows, three
// band searches and three walks per item, each walk as long as the longest of 64 short bands. The triple run T(group, c)
// holds the objects of those three strips MERGED along u (float32 images, weights, and the index of every entry in the
// layout's own order for the exact re-evaluation): one window, one search, one walk whose trip count is the longest of 64
// bands three times as long -- relatively more even. Every object is a member of three triples: 36 bytes of float32 images
// per object more (+ 4 for the index, + 24 with weights). c runs over [first strip - 1, last strip + 1] of the group.
// One thread per entry: its place in each of its three triples is its rank among the members (ties: lower run first) -- the
// order (key, run, position in the run) is the SAME total order of objects in every triple two objects share, which is what
// lets a self count take every unordered pair once: a lane object walks only the entries BEHIND its own place in the triple
// of its strip (pos3), and the pair (a, b) is then met from exactly one side (k_count_band32_one, half bands).
__global__ __launch_bounds__(256) void k_merge_triples(int64_t n, int64_t n_runs, const int64_t *__restrict__ off,
                                                       const int32_t *__restrict__ run_group, const int64_t *__restrict__ vbase,
                                                       const int64_t *__restrict__ off3, const double *__restrict__ key,
                                                       const float *__restrict__ q, int64_t q_stride, const double *__restrict__ w,
                                                       float *__restrict__ q3, int64_t q3_stride, double *__restrict__ w3,
                                                       int32_t *__restrict__ idx3, int32_t *__restrict__ pos3) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t lo = 0, hi = n_runs;  // run of the entry: the largest r with off[r] <= i
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    const int64_t r = lo;
    const int64_t g = run_group[r], g_lo = vbase[g], g_hi = vbase[g + 1];
    const double ki = key[i];
    const float fx = q[i], fy = q[q_stride + i], fz = q[2 * q_stride + i];
    const double wi = w ? w[i] : 0.0;
    // entries of the group's runs r - 2 .. r + 2 in front of this one
    int64_t before[5];
#pragma unroll
    for (int d = -2; d <= 2; ++d) {
        const int64_t m = r + d;
        int64_t cnt = 0;
        if (d == 0) {
            cnt = i - off[r];
        } else if (m >= g_lo && m < g_hi) {
            int64_t l = off[m], h = off[m + 1];
            const int64_t base = l;
            if (d < 0) { while (l < h) { const int64_t mid = (l + h) >> 1; if (key[mid] <= ki) l = mid + 1; else h = mid; } }
            else       { while (l < h) { const int64_t mid = (l + h) >> 1; if (key[mid] < ki) l = mid + 1; else h = mid; } }
            cnt = l - base;
        }
        before[d + 2] = cnt;
    }
#pragma unroll
    for (int d = -1; d <= 1; ++d) {  // triple centred on run r + d: members r + d - 1, r + d, r + d + 1
        const int64_t t = r + d + 1 + 2 * g;
        const int64_t dst = off3[t] + before[d + 1] + before[d + 2] + before[d + 3];
        q3[dst] = fx; q3[q3_stride + dst] = fy; q3[2 * q3_stride + dst] = fz;
        idx3[dst] = (int32_t)i;
        if (w3) w3[dst] = wi;
        if (d == 0) pos3[i] = (int32_t)dst;  // where the object stands in the triple of its OWN strip (half bands of self counts)
    }
}

// ------------------------------------------------------------------------------------------------
// Item builder: one thread per potential item (slot, lane tile).
//   SWEEP = false: the item streams the whole c1 segment; record written at its own index.
//   SWEEP = true : segments are sorted by one coordinate u (the catalogue's sort axis, z by default),
//                  so the tile spans [u(a0), u(a_last)] and only c1 objects with u in
//                  [umin - rwin, umax + rwin] can satisfy s <= t_max (s >= du^2);
//                  rwin[k] = sqrt(t_max[k]) * (1 + 1e-12) + 1e-15 absorbs every rounding in
//                  s = fl(fl(dx^2 + dy^2) + dz^2) >= dz^2 (1 - 3 eps). Items with an empty window are
//                  dropped; survivors are appended with one atomic per workgroup (order is irrelevant).
// ------------------------------------------------------------------------------------------------
constexpr int EVAL_SLOTS = 256;  // statistics counters, one 64-byte line each (a single hot address would serialise)
constexpr int BUILD_WG = 1024;       // most threads per workgroup of the item builders
// Builder workgroups: one atomic per workgroup appends its items, so few large workgroups suit long lists (16 k atomics on
// the one counter cost 0.15 ms at 4 M potential items), but 1024 threads make 300 workgroups for 256 CUs at the headline
// and half the chip waits for the CUs that got two (+0.07 ms): 256 threads while that keeps the atomics below 4096.
#ifndef YAW_BUILD_WG_SMALL
#define YAW_BUILD_WG_SMALL 256
#endif
inline int build_wg_for(int64_t n_pot) { return n_pot / 256 <= 4096 ? YAW_BUILD_WG_SMALL : BUILD_WG; }
constexpr int BUILD_PREFIX_LDS = 1024;  // job tables up to this many entries are searched in LDS by the strip builder (8 KB: no occupancy cost)

// Append the kept items of a builder workgroup to the item list and add its evaluated-pair total: ONE atomic
// per workgroup on each of the two counters. (They are single hot addresses -- with an atomic per wave the
// builders spent three quarters of their time queueing on them.) Order of the list is irrelevant.
// seg_cap > 0: the list is kept in ITEM_SEGS segments of seg_cap records, workgroup b appends to segment b % ITEM_SEGS and
// counts in that segment's own counter (ITEM_SEG_CTR): eight addresses take the atomics of a launch side by side (on one
// address the 1200 appends of the headline queue for 10 of the builder's 43 us), and a segment -- every eighth builder
// workgroup's tiles -- is the same mix of dense and sparse items as the whole list: the float32 band kernels give XCD x
// segment x.
constexpr int ITEM_SEGS = 8;
__host__ __device__ constexpr int ITEM_SEG_CTR(int seg) { return 12 + 8 * seg; }  // counters[]: one 64-byte line each
__device__ __forceinline__ void append_items(bool keep, const Item &it, unsigned long long work, Item *__restrict__ items,
                                             unsigned long long *__restrict__ counters, unsigned char *__restrict__ kept,
                                             unsigned long long seg_cap = 0) {
    if (keep && kept) kept[it.pot] = 1;  // weighted runs: this potential item will write its slab
    __shared__ unsigned int s_cnt[BUILD_WG / 64];
    __shared__ unsigned long long s_work[BUILD_WG / 64], s_base;
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 32; off > 0; off >>= 1) work += __shfl_down(work, off, 64);
    if (lane == 0) {
        s_cnt[wave] = (unsigned int)__popcll(mask);
        s_work[wave] = work;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int total = 0;
        unsigned long long wsum = 0;
        for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) {
            const unsigned int c = s_cnt[wv];
            s_cnt[wv] = total;  // exclusive prefix
            total += c;
            wsum += s_work[wv];
        }
        if (seg_cap) {
            const int seg = (int)(blockIdx.x % ITEM_SEGS);
            s_base = (unsigned long long)seg * seg_cap + (total ? atomicAdd(&counters[ITEM_SEG_CTR(seg)], (unsigned long long)total) : 0ull);
        } else {
            s_base = total ? atomicAdd(&counters[0], (unsigned long long)total) : 0ull;
        }
        if (wsum) atomicAdd(&counters[10 + 8 * (blockIdx.x & (EVAL_SLOTS - 1))], wsum);  // statistics, spread like the other totals
    }
    __syncthreads();
    if (keep) items[s_base + s_cnt[wave] + __popcll(mask & ((1ull << lane) - 1ull))] = it;
}

template <bool SWEEP>
__global__ __launch_bounds__(BUILD_WG) void k_build_items(CatView c1, CatView c2, const int32_t *__restrict__ jobs,
                                                     const int64_t *__restrict__ prefix, int n_slots, int n_bins,
                                                     int tile, const double *__restrict__ rwin, int64_t n_pot,
                                                     Item *__restrict__ items, unsigned long long *__restrict__ counters,
                                                     unsigned char *__restrict__ kept) {
    const int64_t pot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false;
    Item it{};
    unsigned long long work = 0;
    if (pot < n_pot) {
        int lo = 0, hi = n_slots;  // slot = largest s with prefix[s] <= pot
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (prefix[mid] <= pot) lo = mid; else hi = mid;
        }
        const int slot = lo, job = slot / n_bins, k = slot - job * n_bins;
        const int p = jobs[2 * job], q = jobs[2 * job + 1];
        const int k1 = c1.nb == 1 ? 0 : k, k2 = c2.nb == 1 ? 0 : k;
        int64_t b0 = c1.off[(int64_t)p * c1.nb + k1], b1 = c1.off[(int64_t)p * c1.nb + k1 + 1];
        const int64_t a_seg1 = c2.off[(int64_t)q * c2.nb + k2 + 1];
        const int64_t a0 = c2.off[(int64_t)q * c2.nb + k2] + (pot - prefix[slot]) * (int64_t)tile;
        const int64_t a1 = a0 + tile < a_seg1 ? a0 + tile : a_seg1;
        if (SWEEP) {
            const double wlo = c2.key[a0] - rwin[k], whi = c2.key[a1 - 1] + rwin[k];
            int64_t l = b0, h = b1;  // first index with z >= wlo
            while (l < h) {
                const int64_t m = (l + h) >> 1;
                if (c1.key[m] < wlo) l = m + 1; else h = m;
            }
            const int64_t first = l;
            h = b1;  // first index with z > whi
            while (l < h) {
                const int64_t m = (l + h) >> 1;
                if (c1.key[m] <= whi) l = m + 1; else h = m;
            }
            b0 = first;
            b1 = l;
        }
        keep = b1 > b0;
        it.a0 = a0; it.b0[0] = b0; it.na = (int32_t)(a1 - a0); it.nb[0] = (int32_t)(b1 - b0); it.nwin = 1; it.slot = slot; it.pot = (int32_t)pot;
        work = keep ? (unsigned long long)it.na * (unsigned long long)it.nb[0] : 0ull;
    }
    if (SWEEP) {
        append_items(keep, it, work, items, counters, kept);
    } else {
        if (pot < n_pot) items[pot] = it;  // every potential item is kept
        append_items(false, it, work, items, counters, nullptr);
        if (pot == 0) counters[0] = (unsigned long long)n_pot;
    }
}

// ------------------------------------------------------------------------------------------------
// Item builder of the strip path. A job (p, q) is cut into potential items
//   (lane tile of a (patch q, strip) run of c2)  x  (one of the 2*reach+1 neighbouring strips of patch p in c1),
// enumerated arithmetically: no per-job tables travel from the host. One thread per potential item finds its
// job (prefix over jobs), its tile (prefix of tiles over the runs of c2) and the strip of c1 on the common
// grid; window search and compaction as in k_build_items<true>.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BUILD_WG) void k_build_items_strips(const DevTab *__restrict__ tabs, const JobRec *__restrict__ jobs,
                                                            const int64_t *__restrict__ prefix, int n_jobs, int reach,
                                                            int tile, double rwin, int swap, int triple, int64_t n_pot,
                                                            Item *__restrict__ items, unsigned long long *__restrict__ counters,
                                                            unsigned char *__restrict__ kept, unsigned long long seg_cap) {
    // triple: the streamed side consists of merged triple runs (k_merge_triples) -- the host passes reach = 0 (one partner
    // run per lane tile: the triple centred on its strip) and the triples' offsets and grid index in the streamed table; their
    // sort key exists as float32 image only, so the window is widened by the rounding of a key (the count kernel searches
    // its bands in float32 with a margin of its own).
    const int64_t pot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false;
    Item it{};
    unsigned long long work = 0;
    // Every thread walks a chain of dependent loads (job -> tile -> runs -> windows); its length is the kernel's run time.
    // The job table is small: searched in LDS (one coalesced load instead of log2(jobs) round trips to L2).
    __shared__ int64_t s_prefix[BUILD_PREFIX_LDS];
    // ... and so are the six layout records: which one a thread needs depends on its job's orientation, so every pointer in
    // them would be a vector load of its own in front of the load it points to (five round trips of the chain)
    __shared__ DevTab s_tabs[6];
    static_assert(sizeof(DevTab) % 8 == 0, "DevTab is copied in 8-byte words");
    for (int e = threadIdx.x; e < (int)(6 * sizeof(DevTab) / 8); e += blockDim.x)
        reinterpret_cast<uint64_t *>(s_tabs)[e] = reinterpret_cast<const uint64_t *>(tabs)[e];
    const bool prefix_in_lds = n_jobs + 1 <= BUILD_PREFIX_LDS;
    if (prefix_in_lds)
        for (int e = threadIdx.x; e <= n_jobs; e += blockDim.x) s_prefix[e] = prefix[e];
    __syncthreads();
    if (pot < n_pot) {
        int lo = 0, hi = n_jobs;  // job = largest j with prefix[j] <= pot
        if (prefix_in_lds) {
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_prefix[mid] <= pot) lo = mid; else hi = mid;
            }
        } else {
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (prefix[mid] <= pot) lo = mid; else hi = mid;
            }
        }
        const int job = lo;
        const JobRec jr = jobs[job];
        const int o = jr.o & 3;  // orientation of the job: which pair of layouts it runs on
        it.pad_ = (jr.o >> 2) & 1;  // 1: diagonal job of a self count: lanes walk only the entries behind their own place (see k_merge_triples)
        // swap: the lane tiles come from the first catalogue of the job (the binned one), the windows from the second
#if defined(YAW_BUILD_TABS_GLOBAL)  // (A/B: the records read from the table in global memory, as before)
        const DevTab &c1 = tabs[swap ? 3 + o : o], &c2 = tabs[swap ? o : 3 + o];
#else
        const DevTab &c1 = s_tabs[swap ? 3 + o : o], &c2 = s_tabs[swap ? o : 3 + o];
#endif
        const gf64p key1d = tab_key(c1), key2 = tab_key(c2);
        const gf32p key1f = c1.axis == 0 ? c1.qx : (c1.axis == 1 ? c1.qy : c1.qz);
        auto key1 = [&](int64_t i) { return triple ? (double)key1f[i] : key1d[i]; };
        // potential items of a job in the order (lane tile, group of neighbour offsets). One item carries up to MAX_WIN
        // neighbouring strips of the streamed group (all 2 * reach + 1 = 3 of them when the grid is as wide as the
        // largest separation): the lane tile is loaded once and its histogram flushed once for all of them.
        const int nd = 2 * reach + 1, ng = (nd + MAX_WIN - 1) / MAX_WIN;
        const int64_t local = pot - (prefix_in_lds ? s_prefix[job] : prefix[job]);
        // (a call has fewer than 2^31 potential items: 32-bit division, and none in the usual case of one group)
        const uint32_t local32 = (uint32_t)local, tl = ng == 1 ? local32 : local32 / (uint32_t)ng;
        const int g = (int)(local32 - tl * (uint32_t)ng);
        const TileRec tr = c2.tile_rec[jr.t_lo + tl];
        const int64_t r2 = tr.run, a0 = tr.a0, a1 = a0 + tr.na;
        const double kpad = triple ? 1.2e-7 : 0.0;  // float32 rounding of a streamed key (|u| <= 1: 2^-24)
        // (half bands: no lane of the tile looks at an entry in front of the tile's first object -- the window starts there)
        const double wlo = key2[a0] - (it.pad_ ? 0.0 : rwin) - kpad, whi = key2[a1 - 1] + rwin + kpad;
        it.a0 = a0; it.na = tr.na; it.nwin = 0;
        it.slot = (int32_t)((unsigned)job | ((unsigned)o << 30)); it.pot = (int32_t)pot;
        // The windows of the (up to) three partner runs are searched in lockstep: three independent chains of loads per
        // thread instead of one after the other (static indices throughout: everything stays in registers).
        // (bounds relative to the run's first entry, 32 bits: half the integer work of the search)
        int64_t wb[MAX_WIN];
        uint32_t sl[MAX_WIN], sh[MAX_WIN], ul[MAX_WIN], uh[MAX_WIN];
        int32_t wn[MAX_WIN];
#pragma unroll
        for (int j = 0; j < MAX_WIN; ++j) {
            const int dd = g * MAX_WIN + j;
            const int64_t s1 = r2 + jr.k_off + (dd - reach);
            const bool valid = dd < nd && s1 >= 0 && s1 < jr.n_strips1;
            const int64_t r1 = jr.vbase1 + (valid ? s1 : 0);
            int64_t b0 = 0, b1 = 0;
            if (valid) { b0 = c1.off[r1]; b1 = c1.off[r1 + 1]; }
            // four partner runs in five lie entirely before or behind the tile's window (neighbouring patches share a
            // boundary only): two loads settle those
            const bool some = b1 > b0;
            // (loads under their condition: the texture path is what this kernel is bound by, and it charges the lanes of a
            // load that are switched on; a wave whose tiles face no live run skips them altogether)
            double kfirst = 0.0, klast = 0.0;
            if (some) { kfirst = key1(b0); klast = key1(b1 - 1); }
            const bool live = some && !(klast < wlo || kfirst > whi);
            // the run's index along the sort axis narrows both searches to one cell (RunGrid)
            uint32_t l0 = 0, l1 = 0, u0 = 0, u1 = 0;
            if (live) {
                const double inv = c1.grid[r1].inv;
                const int cl = run_cell(wlo, kfirst, inv), cu = run_cell(whi, kfirst, inv);
                const uint32_t *gr = c1.grid[r1].g;
                l0 = gr[cl]; l1 = gr[cl + 1]; u0 = gr[cu]; u1 = gr[cu + 1];
            }
            wb[j] = b0;
            sl[j] = l0; sh[j] = l1; ul[j] = u0; uh[j] = u1;  // all 0 for a dead window
        }
        // lower bounds (first index with key >= wlo) and upper bounds (first index with key > whi), all at once
        for (int step = 0; step < YAW_BUILD_BISECT &&
                           ((sl[0] < sh[0]) | (sl[1] < sh[1]) | (sl[2] < sh[2]) | (ul[0] < uh[0]) | (ul[1] < uh[1]) | (ul[2] < uh[2])); ++step) {
            double kl[MAX_WIN], ku[MAX_WIN];
            uint32_t ml[MAX_WIN], mu[MAX_WIN];
#pragma unroll
            for (int j = 0; j < MAX_WIN; ++j) {  // the six probes of a step are issued together ...
                ml[j] = sl[j] + ((sh[j] - sl[j]) >> 1);
                mu[j] = ul[j] + ((uh[j] - ul[j]) >> 1);
                kl[j] = ku[j] = 0.0;
                if (sl[j] < sh[j]) kl[j] = key1(wb[j] + ml[j]);
                if (ul[j] < uh[j]) ku[j] = key1(wb[j] + mu[j]);
            }
#pragma unroll
            for (int j = 0; j < MAX_WIN; ++j) {  // ... and consumed after one wait
                if (sl[j] < sh[j]) { if (kl[j] < wlo) sl[j] = ml[j] + 1; else sh[j] = ml[j]; }
                if (ul[j] < uh[j]) { if (ku[j] <= whi) ul[j] = mu[j] + 1; else uh[j] = mu[j]; }
            }
        }
#pragma unroll
        for (int j = 0; j < MAX_WIN; ++j) {  // (a search cut short leaves sl <= first entry of the window, uh >= its end)
            wb[j] += sl[j];
            wn[j] = (int32_t)((YAW_BUILD_BISECT < 32 ? uh[j] : ul[j]) - sl[j]);  // 0 for a dead window
        }
#pragma unroll
        for (int j = 0; j < MAX_WIN; ++j) {  // non-empty windows to the front
            if (wn[j] <= 0) continue;
            if (it.nwin == 0) { it.b0[0] = wb[j]; it.nb[0] = wn[j]; }
            else if (it.nwin == 1) { it.b0[1] = wb[j]; it.nb[1] = wn[j]; }
            else { it.b0[2] = wb[j]; it.nb[2] = wn[j]; }
            ++it.nwin;
            work += (unsigned long long)it.na * (unsigned long long)wn[j];
        }
        keep = it.nwin > 0;
    }
    append_items(keep, it, work, items, counters, kept, seg_cap);
}

// ------------------------------------------------------------------------------------------------
// Pair count kernel.
//   R         objects per lane (lane tile = 256*R objects of the c2 segment)
//   WEIGHTED  accumulate w_a*w_b in float64 (else count in uint32)
//   PRIVATE   per-lane private LDS histogram (deterministic); else one shared LDS histogram per
//             workgroup updated with LDS atomics (only used when E is too large for private ones)
//   FILTER    false: every pair is evaluated in FP64 (8 flop + 1 compare);
//             true:  every pair is first tested in FP32 as  a.b >= 1 - t_max/2 - guard  (mul + 2 fma
//                    + compare on the float32 images of the unit vectors); only survivors (a few
//                    1e-5 of the pairs) are evaluated with the exact FP64 predicate. The test is
//                    conservative: s = |a|^2 + |b|^2 - 2 a.b, so s <= t_max implies
//                    a.b >= 1 - t_max/2 - eps_norm, and the float32 dot product is within 5.000001 u
//                    of a.b; dthr[k] is that bound rounded down. A pair the filter drops therefore has
//                    s > t_max and belongs to no bin: results are bit-identical to FILTER=false.
// ------------------------------------------------------------------------------------------------
template <int R, bool WEIGHTED, bool PRIVATE, bool FILTER>
__global__ __launch_bounds__(WG) void k_count(CatView c1, CatView c2, const Item *__restrict__ items, int n_bins,
                                              int n_edges, const double *__restrict__ t,
                                              const float *__restrict__ dthr, int64_t item_base,
                                              unsigned long long *__restrict__ out_counts,
                                              double *__restrict__ partials,
                                              const unsigned long long *__restrict__ n_kept) {
    using HistT = typename std::conditional<WEIGHTED, double, unsigned int>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    Obj *stage = reinterpret_cast<Obj *>(lds_raw);                                  // [2][STAGE]
    ObjF *stagef = reinterpret_cast<ObjF *>(lds_raw + 2 * STAGE * sizeof(Obj));     // [2][STAGE]
    double *thr = reinterpret_cast<double *>(lds_raw + 2 * STAGE * (sizeof(Obj) + sizeof(ObjF)));  // [n_edges]
    HistT *hist = reinterpret_cast<HistT *>(thr + ((n_edges + 1) & ~1));            // [nf][WG] or [nf]

    const int tid = threadIdx.x;
    const int nf = n_edges - 1;
    if ((unsigned long long)(item_base + blockIdx.x) >= *n_kept) return;  // grid = potential items; the builder kept fewer
    const Item it = items[item_base + blockIdx.x];
    const int slot = it.slot;
    const int k = slot % n_bins;
    const int64_t b0 = it.b0[0], b1 = it.b0[0] + it.nb[0];  // items of k_build_items carry one window
    const int64_t a0 = it.a0, a_seg1 = it.a0 + it.na;
    const int64_t item = it.pot;

    // lane objects (c2 side) -> registers; padded lanes are parked far away
    double ax[R], ay[R], az[R], aw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = a0 + (int64_t)r * WG + tid;
        const bool ok = i < a_seg1;
        ax[r] = ok ? c2.x[i] : PAD_COORD;
        ay[r] = ok ? c2.y[i] : PAD_COORD;
        az[r] = ok ? c2.z[i] : PAD_COORD;
        aw[r] = (WEIGHTED && ok && c2.w) ? c2.w[i] : (ok ? 1.0 : 0.0);
    }
    float fx[R], fy[R], fz[R];  // float32 images; a padded lane gets NaN -> its dot product fails every comparison
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool ok = a0 + (int64_t)r * WG + tid < a_seg1;
        fx[r] = ok ? (float)ax[r] : __builtin_nanf("");
        fy[r] = ok ? (float)ay[r] : 0.f;
        fz[r] = ok ? (float)az[r] : 0.f;
    }
    const float dmin = FILTER ? dthr[3 * k] : 0.f;

    for (int e = tid; e < n_edges; e += WG) thr[e] = t[(int64_t)k * n_edges + e];
    if (PRIVATE) {
        for (int j = 0; j < nf; ++j) hist[j * WG + tid] = HistT(0);
    } else {
        for (int j = tid; j < nf; j += WG) hist[j] = HistT(0);
    }
    const double tmax = t[(int64_t)k * n_edges + n_edges - 1];

    const int64_t nb_total = b1 - b0;
    const int nstages = (int)((nb_total + STAGE - 1) / STAGE);
    // LDS histogram(s) -> result: fixed-order tree reduction of the private histograms, then 64-bit integer atomics
    // (unweighted; the histogram is cleared and counting goes on) or the item's slab (weighted, once at the end)
    auto flush_counts = [&]() {
        if (PRIVATE) {
            for (int stride = WG / 2; stride > 0; stride >>= 1) {
                if (tid < stride)
                    for (int j = 0; j < nf; ++j) hist[j * WG + tid] += hist[j * WG + tid + stride];
                __syncthreads();
            }
        }
        for (int j = tid; j < nf; j += WG) {
            const HistT v = PRIVATE ? hist[j * WG] : hist[j];
            if (WEIGHTED) partials[item * nf + j] = (double)v;
            else if (v != HistT(0)) atomicAdd(&out_counts[(int64_t)slot * nf + j], (unsigned long long)v);
        }
        __syncthreads();
        if (PRIVATE) {
            for (int j = 0; j < nf; ++j) hist[j * WG + tid] = HistT(0);
        } else {
            for (int j = tid; j < nf; j += WG) hist[j] = HistT(0);
        }
        __syncthreads();
    };

    // stage 0
    {
        const int64_t i = b0 + tid;
        Obj o;
        const bool ok = i < b1;
        o.x = ok ? c1.x[i] : 0.0; o.y = ok ? c1.y[i] : 0.0; o.z = ok ? c1.z[i] : 0.0;
        o.w = (WEIGHTED && ok && c1.w) ? c1.w[i] : 1.0;
        stage[tid] = o;
        if (FILTER) stagef[tid] = ObjF{(float)o.x, (float)o.y, (float)o.z, 0.f};
    }
    __syncthreads();

    for (int st = 0; st < nstages; ++st) {
        const Obj *cur = stage + (st & 1) * STAGE;
        // issue the next stage's global loads early; they land in registers while we compute
        Obj nxt;
        const bool have_next = st + 1 < nstages;
        if (have_next) {
            const int64_t i = b0 + (int64_t)(st + 1) * STAGE + tid;
            const bool ok = i < b1;
            nxt.x = ok ? c1.x[i] : 0.0; nxt.y = ok ? c1.y[i] : 0.0; nxt.z = ok ? c1.z[i] : 0.0;
            nxt.w = (WEIGHTED && ok && c1.w) ? c1.w[i] : 1.0;
        }
        const int64_t left = nb_total - (int64_t)st * STAGE;
        const int n = left < STAGE ? (int)left : STAGE;

        const ObjF *curf = stagef + (st & 1) * STAGE;
        // exact evaluation + histogram update of lane object r against streamed object b
        auto settle = [&](int r, const Obj &b) {
            const double dx = ax[r] - b.x;
            const double dy = ay[r] - b.y;
            const double dz = az[r] - b.z;
            const double xx = dx * dx;
            const double yy = dy * dy;
            const double zz = dz * dz;
            const double sxy = xx + yy;
            const double s = sxy + zz;
            if (s <= tmax) {
                int cnt = 0;
                for (int e = 0; e < n_edges; ++e) cnt += (s > thr[e]) ? 1 : 0;
                if (cnt > 0) {  // t[cnt-1] < s <= t[cnt]
                    const HistT v = WEIGHTED ? HistT(aw[r] * b.w) : HistT(1);
                    if (PRIVATE) hist[(cnt - 1) * WG + tid] += v;
                    else atomicAdd(&hist[cnt - 1], v);
                }
            }
        };
        if (FILTER) {
            for (int i = 0; i < n; ++i) {
                const ObjF bf = curf[i];  // wave-wide broadcast read, 16 B
                float d[R];
                float best = -2.f;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    d[r] = __builtin_fmaf(fz[r], bf.z, __builtin_fmaf(fy[r], bf.y, fx[r] * bf.x));
                    best = fmaxf(best, d[r]);
                }
                if (__builtin_amdgcn_ballot_w64(best >= dmin) != 0ull) {  // rare: a pair may be inside the outer edge
                    const Obj b = cur[i];
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (d[r] >= dmin) settle(r, b);
                }
            }
        } else {
            for (int i = 0; i < n; ++i) {
                const Obj b = cur[i];  // wave-wide broadcast read
                double s[R];
                bool any = false;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double dx = ax[r] - b.x;
                    const double dy = ay[r] - b.y;
                    const double dz = az[r] - b.z;
                    const double xx = dx * dx;
                    const double yy = dy * dy;
                    const double zz = dz * dz;
                    const double sxy = xx + yy;
                    s[r] = sxy + zz;
                    any |= (s[r] <= tmax);
                }
                if (__builtin_amdgcn_ballot_w64(any) != 0ull) {  // rare: some lane has a pair inside the outer edge
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (s[r] <= tmax) settle(r, b);
                }
            }
        }

        if (have_next) {
            stage[((st + 1) & 1) * STAGE + tid] = nxt;
            if (FILTER) stagef[((st + 1) & 1) * STAGE + tid] = ObjF{(float)nxt.x, (float)nxt.y, (float)nxt.z, 0.f};
        }
        __syncthreads();
        // 32-bit counters: 256 * R lane objects (R <= 4) x 256 streamed objects per stage reach 2^32 after 2^14 stages
        // (a c1 segment of 4.2 M objects inside one wide bin): move them to the 64-bit result before that
        if (!WEIGHTED && (st & COUNT_FLUSH_MASK) == COUNT_FLUSH_MASK && st + 1 < nstages) flush_counts();
    }
    flush_counts();
}

// ------------------------------------------------------------------------------------------------
// Culling kernel (SWEEP). Cross-correlation form: c1 binned in redshift, c2 unbinned, unit vectors, both in
// their strip layouts (runs of (patch, strip), all bins together, bin id per object on the c1 side).
// A work item = (lane tile: 64*R consecutive objects of one run of c2) x (the window of one partner run of c1
// that can hold partners of the tile, whatever their bin). One wave = one workgroup = one item.
//
// Fast path (all pairs of the window): only float32 lives on chip. Lanes keep the float32 image of
// their R objects packed in pairs; the stream is staged in LDS 64 objects at a time as 16-byte records
// (xf, yf, zf, pre-filter threshold of the object's own bin), so per-bin scales cost nothing in the loop:
// per trip (two streamed objects, R = 2: 256 pairs) 6 v_pk_mul/fma_f32 + 2 v_max + 2 v_cmp, next trip
// prefetched from LDS. Stage entries outside the wave's own window are skipped (ballot + popcount).
//
// Survivors (the real pairs and a 1e-6 fringe): the owner lane pushes a 4-byte code (r, lane, stage
// slot) on its wave's queue; when 64 are waiting (or the stage ends) every lane takes one, gathers
// the two float64 positions from global memory (L2-hot: both were just read by this workgroup) and
// evaluates the exact predicate -- full lanes, one memory latency per 64 survivors. Hits go to a
// [B][E-1] histogram in LDS:
//   unweighted: uint32 LDS atomics (exact, order independent);
//   weighted:   one float64 histogram per wave, updated with wave-private LDS float64 atomics
//               (no other wave writes it => bit-reproducible sums, see the drain).
// One item covers all B bins, so the c2 tile is read once per job instead of once per (job, bin).
// ------------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
struct MergedView {
    const double *x, *y, *z, *w;  // w may be null
    const int32_t *k;             // bin id per object
};

// MERGED = true:  the cross-correlation form described above (c1 binned, c2 unbinned, strip layouts, one item for
//                 all bins).
// MERGED = false: per-bin items (job, bin, tile): every streamed object belongs to the item's bin. Serves the
//                 binned x binned counts of an autocorrelation -- on the per-(patch, bin) strip layouts when the lane
//                 side is dense enough, else on the plain (patch, bin, u) layout -- and everything without a common
//                 strip grid.
template <int R, bool WEIGHTED, bool NF1, bool MERGED>
__device__ __forceinline__ void count_merged_body(const DevTab *__restrict__ tabs, const Item *__restrict__ items,
                                                     int n_bins, int n_edges, const double *__restrict__ t,
                                                     const float *__restrict__ dthr, const double *__restrict__ rwin_k,
                                                     int64_t item_base, unsigned long long *__restrict__ out_counts,
                                                     double *__restrict__ partials,
                                                     const unsigned long long *__restrict__ counters) {
    using HistT = typename std::conditional<WEIGHTED, double, unsigned int>::type;
    constexpr int NHIST = WEIGHTED ? MWG / 64 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ObjF *stagef = reinterpret_cast<ObjF *>(lds_raw);                                // [2][MSTAGE]
    double *thr = reinterpret_cast<double *>(stagef + 2 * MSTAGE);                   // [nkb][n_edges]
    const int nkb_l = MERGED ? n_bins : 1;
    HistT *hist = reinterpret_cast<HistT *>(thr + (size_t)nkb_l * n_edges);          // [NHIST][nkb*nf]
    float *dth = reinterpret_cast<float *>(hist + (size_t)NHIST * nkb_l * (n_edges - 1));  // [nkb]
    unsigned int *candq = reinterpret_cast<unsigned int *>(dth + nkb_l);             // [MWG/64][64] survivor codes

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nf = n_edges - 1;
    const unsigned long long ticket = item_base + blockIdx.x;
    if (ticket >= counters[0]) return;  // never taken when the host sized the grid from the builder's count
    const Item it = items[ticket];
    const int o = item_orient(it), islot = item_slot(it);
    const DevTab T1 = tabs[o], T2 = tabs[3 + o];
    const MergedView c1{(const double *)T1.x, (const double *)T1.y, (const double *)T1.z, (const double *)T1.w, (const int32_t *)T1.k};
    const CatView c2{(const double *)T2.x, (const double *)T2.y, (const double *)T2.z, (const double *)T2.w, (const int64_t *)T2.off, 1,
                     (const double *)tab_key(T2), T2.axis};
    const int kfix = MERGED ? 0 : islot % n_bins;  // the item's bin (ordinary items)
    const int nkb = MERGED ? n_bins : 1;             // bins this item can add to
    const int nslots = nkb * nf;
    const double rwin = rwin_k[kfix];
    int64_t b0 = it.b0[0], b1 = it.b0[0] + it.nb[0];  // the current window (an item of the strip builder carries up to three)
    const int64_t a0 = it.a0, a_end = it.a0 + it.na;
    // wave-contiguous assignment: wave w owns objects [w*64R, (w+1)*64R) of the z-sorted tile, so its
    // own z-window is narrower than the workgroup's
    const int64_t wa0 = a0 + (int64_t)wave * (64 * R);

    // Everything the item needs from global memory is requested here, back to back, before the first use:
    // lane objects, the key range of the wave, the first stage of the stream, thresholds (one memory latency).
    int64_t nb_total = b1 - b0;
#ifdef YAW_DIAG_SKIP_STREAM
    int nstages = 0;  // diagnostics: per-item fixed cost only (wrong counts)
#else
    int nstages = (int)((nb_total + MSTAGE - 1) / MSTAGE);
#endif
    constexpr int NPF = (MSTAGE + MWG - 1) / MWG;  // stage slots a thread fills
    struct Raw { double x, y, z; int k; bool in; };
    auto fetch_raw = [&](int64_t i) {  // streamed object i (clamped into the window: unconditional loads)
        Raw o;
        o.in = i < b1;
        const int64_t ic = o.in ? i : b0;
        o.x = c1.x[ic]; o.y = c1.y[ic]; o.z = c1.z[ic];
        o.k = MERGED ? c1.k[ic] : 0;
        return o;
    };
    auto finish = [&](const Raw &o) {  // float32 record; slots past the window get a threshold nothing passes
        return ObjF{(float)o.x, (float)o.y, (float)o.z, o.in ? dth[o.k] : 2.0f};
    };
    double lx[R], ly[R], lz[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = wa0 + (int64_t)r * 64 + lane;
        const int64_t ic = i < a_end ? i : a0;  // padded lanes read a valid object
        lx[r] = c2.x[ic]; ly[r] = c2.y[ic]; lz[r] = c2.z[ic];
    }
    int64_t wa1 = wa0 + 64 * R;
    if (wa1 > a_end) wa1 = a_end;
    const bool wave_has = wa0 < wa1;
    const double key_lo = c2.key[wave_has ? wa0 : a0], key_hi = c2.key[wave_has ? wa1 - 1 : a0];
    Raw first[NPF];
#pragma unroll
    for (int f = 0; f < NPF; ++f) first[f] = fetch_raw(b0 + f * MWG + tid);
    // threshold tables: the first MWG entries travel with the batch above (all of them in the standard
    // 30 bins x 2 edges case), the rest in ordinary loops
    const bool has_t0 = tid < nkb * n_edges, has_d0 = tid < nkb;
    const double t0 = t[(int64_t)kfix * n_edges + (has_t0 ? tid : 0)];
    const float d0 = dthr[3 * (kfix + (has_d0 ? tid : 0))];
    __builtin_amdgcn_sched_barrier(0);
    if (has_t0) thr[tid] = t0;
    if (has_d0) dth[tid] = d0;
    for (int e = tid + MWG; e < nkb * n_edges; e += MWG) thr[e] = t[(int64_t)kfix * n_edges + e];
    for (int e = tid + MWG; e < nkb; e += MWG) dth[e] = dthr[3 * (kfix + e)];
    for (int e = tid; e < NHIST * nslots; e += MWG) hist[e] = HistT(0);
    __syncthreads();

    float fx[R], fy[R], fz[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool ok = wa0 + (int64_t)r * 64 + lane < a_end;
        fx[r] = ok ? (float)lx[r] : __builtin_nanf("");  // padded lane: dot = NaN fails every comparison (thresholds
                                                          // are <= 0 for separations >= 90 degrees, so 0 would pass)
        fy[r] = ok ? (float)ly[r] : 0.f;
        fz[r] = ok ? (float)lz[r] : 0.f;
    }
    constexpr int RP = (R + 1) / 2;
    v2f px[RP], py[RP], pz[RP];  // the same, packed in pairs for v_pk_*_f32 (R = 1: the upper half idles on NaN)
#pragma unroll
    for (int q = 0; q < RP; ++q) {
        const int r1 = 2 * q + 1 < R ? 2 * q + 1 : 2 * q;
        px[q] = v2f{fx[2 * q], 2 * q + 1 < R ? fx[r1] : __builtin_nanf("")};
        py[q] = v2f{fy[2 * q], fy[r1]};
        pz[q] = v2f{fz[2 * q], fz[r1]};
    }
    // z-range of this wave's lane objects (+/- the window half width), as conservative float32 bounds
    // for comparison with the float32 z of the staged objects (monotone rounding keeps them conservative)
    float wz_lo = 4.0f, wz_hi = -4.0f;  // wave without objects: empty range
    if (wave_has) {
        const double lo = key_lo - rwin, hi = key_hi + rwin;
        wz_lo = (float)lo;
        if ((double)wz_lo > lo) wz_lo = nextafterf(wz_lo, -4.0f);
        wz_hi = (float)hi;
        if ((double)wz_hi < hi) wz_hi = nextafterf(wz_hi, 4.0f);
    }
#pragma unroll
    for (int f = 0; f < NPF; ++f)
        if (f * MWG + tid < MSTAGE) stagef[f * MWG + tid] = finish(first[f]);
    __syncthreads();

    auto flush_hist = [&]() {  // LDS histogram -> 64-bit result (unweighted: cleared, counting goes on) / the item's slab
        for (int idx = tid; idx < nslots; idx += MWG) {
            if (WEIGHTED) {
                double v = 0.0;
                for (int wv = 0; wv < NHIST; ++wv) v += reinterpret_cast<double *>(hist)[wv * nslots + idx];
                partials[(int64_t)it.pot * nslots + idx] = v;
            } else {
                const unsigned int v = reinterpret_cast<unsigned int *>(hist)[idx];
                if (v) atomicAdd(&out_counts[(int64_t)islot * nslots + idx], (unsigned long long)v);
                reinterpret_cast<unsigned int *>(hist)[idx] = 0u;
            }
        }
    };
    int qn = 0;  // entries in this wave's survivor queue (wave-uniform)
    for (int win = 0; win < it.nwin; ++win) {
    if (win > 0) {  // next window of the item: its first stage goes through the same double buffer
        b0 = win == 1 ? it.b0[1] : it.b0[2];
        b1 = b0 + (win == 1 ? it.nb[1] : it.nb[2]);
        nb_total = b1 - b0;
#ifndef YAW_DIAG_SKIP_STREAM
        nstages = (int)((nb_total + MSTAGE - 1) / MSTAGE);
#endif
#pragma unroll
        for (int f = 0; f < NPF; ++f) first[f] = fetch_raw(b0 + f * MWG + tid);
        __syncthreads();  // every lane is done with the previous window's last stage
#pragma unroll
        for (int f = 0; f < NPF; ++f)
            if (f * MWG + tid < MSTAGE) stagef[f * MWG + tid] = finish(first[f]);
        __syncthreads();
    }
    for (int st = 0; st < nstages; ++st) {
        const int cb = st & 1;
        const int64_t sb0 = b0 + (int64_t)st * MSTAGE;  // global index of stage slot 0
        Raw nxt[NPF];
        const bool have_next = st + 1 < nstages;
#pragma unroll
        for (int f = 0; f < NPF; ++f)
            if (have_next) nxt[f] = fetch_raw(sb0 + MSTAGE + f * MWG + tid);
        const int64_t left = nb_total - (int64_t)st * MSTAGE;
        const int n = left < MSTAGE ? (int)left : MSTAGE;
        const ObjF *curf = stagef + cb * MSTAGE;

        // Exact evaluation of the queued survivors, one per lane.
        auto drain = [&]() {
            if (qn == 0) return;
            int hslot = -1;
            double val = 0.0;
            if (lane < qn) {
                const unsigned int code = candq[wave * 64 + lane];
                const int64_t ia = wa0 + (code >> 8);         // (r * 64 + lane) of the owner
                const int64_t ib = sb0 + (code & 0xffu);      // stage slot
                // issue all gathers before the first use: one memory latency per drain, not three
                const double ax = c2.x[ia], ay = c2.y[ia], az = c2.z[ia];
                const double bx = c1.x[ib], by = c1.y[ib], bz = c1.z[ib];
                const int kb = MERGED ? c1.k[ib] : 0;
                double wa = 1.0, wb = 1.0;
                if (WEIGHTED) {
                    if (c2.w) wa = c2.w[ia];
                    if (c1.w) wb = c1.w[ib];
                }
                __builtin_amdgcn_sched_barrier(0);
                const double dx = ax - bx;
                const double dy = ay - by;
                const double dz = az - bz;
                const double xx = dx * dx;
                const double yy = dy * dy;
                const double zz = dz * dz;
                const double sxy = xx + yy;
                const double s = sxy + zz;
                const double *tk = thr + kb * n_edges;
                if (s > tk[0] && s <= tk[n_edges - 1]) {
                    if (NF1) {
                        hslot = kb;
                    } else {
                        int cnt = 0;
                        for (int e = 0; e < n_edges; ++e) cnt += (s > tk[e]) ? 1 : 0;
                        hslot = kb * nf + cnt - 1;  // t[cnt-1] < s <= t[cnt], cnt >= 1 because s > t[0]
                    }
                    if (WEIGHTED) val = wa * wb;
                }
            }
            if (!WEIGHTED) {
                if (hslot >= 0) atomicAdd(reinterpret_cast<unsigned int *>(hist) + hslot, 1u);
            } else {
                // One LDS float64 atomic for the whole wave (ds_add_f64). The histogram belongs to this wave
                // alone and a wave's LDS instructions execute in program order, so the only freedom is the order
                // in which the LDS unit serialises lanes of ONE instruction that hit the same slot -- a fixed
                // property of the hardware, not a race: sums are bit-reproducible from run to run
                // (tests/test_gpu_scale_properties.py checks that at 10M x 10M).
                double *wh = reinterpret_cast<double *>(hist) + wave * nslots;
                if (hslot >= 0) atomicAdd(&wh[hslot], val);
            }
            qn = 0;
        };
        // Owner lanes of the survivors of stage slot i push their code on the wave's queue.
        auto enqueue = [&](int i, const float (&d)[R], float dmin) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool pass = d[r] >= dmin;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
                if (m == 0ull) continue;  // uniform
                const int cnt = __popcll(m);
                if (qn + cnt > 64) drain();
                if (pass) {
                    const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    candq[wave * 64 + pos] = ((unsigned)(r * 64 + lane) << 8) | (unsigned)i;
                }
                qn += cnt;
            }
        };

        // this wave's sub-range of the (z-sorted) stage: entries below wz_lo / above wz_hi cannot pair with it
        int i_lo = 0, i_hi = 0;
#pragma unroll
        for (int j = 0; j < MSTAGE / 64; ++j) {
            const int e = j * 64 + lane;
            const float ze = (&curf[e].x)[c2.axis];  // float32 image of the sorted coordinate
            i_lo += __popcll(__builtin_amdgcn_ballot_w64(e < n && ze < wz_lo));
            i_hi += __popcll(__builtin_amdgcn_ballot_w64(e < n && ze <= wz_hi));
        }
        i_lo &= ~1;
        if (i_lo > MSTAGE - 2) i_lo = MSTAGE - 2;  // keeps the first read inside the stage; the loop is then empty
        // two streamed objects per trip, next pair prefetched from LDS while this one is evaluated.
        // Slots past the window hold a threshold of 2 (nothing passes), so an odd tail needs no test.
        // One trip = two streamed objects against the R lane objects, in packed float32 (v_pk_mul/fma_f32:
        // two lane objects per instruction). The loop is unrolled over two trips with alternating register
        // sets so that the LDS prefetch of the next pair needs no register moves.
        auto trip = [&](int i, const ObjF &c0, const ObjF &c1r) {
            v2f e0[RP], e1[RP];
#pragma unroll
            for (int q = 0; q < RP; ++q) {
                e0[q] = __builtin_elementwise_fma(pz[q], v2f{c0.z, c0.z}, __builtin_elementwise_fma(py[q], v2f{c0.y, c0.y}, px[q] * v2f{c0.x, c0.x}));
                e1[q] = __builtin_elementwise_fma(pz[q], v2f{c1r.z, c1r.z}, __builtin_elementwise_fma(py[q], v2f{c1r.y, c1r.y}, px[q] * v2f{c1r.x, c1r.x}));
            }
            float d0[R], d1[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                d0[r] = (r & 1) ? e0[r >> 1].y : e0[r >> 1].x;
                d1[r] = (r & 1) ? e1[r >> 1].y : e1[r >> 1].x;
            }
            float best0 = d0[0], best1 = d1[0];
#pragma unroll
            for (int r = 1; r < R; ++r) {
                best0 = fmaxf(best0, d0[r]);
                best1 = fmaxf(best1, d1[r]);
            }
            const bool p0 = best0 >= c0.pad, p1 = best1 >= c1r.pad;
            // wave masks straight from the compares (a bool that goes through a VGPR costs two VALU per test)
            const unsigned long long m0 = __builtin_amdgcn_ballot_w64(p0), m1 = __builtin_amdgcn_ballot_w64(p1);
            if ((m0 | m1) != 0ull) {
                if (m0 != 0ull) enqueue(i, d0, c0.pad);
                if (m1 != 0ull) enqueue(i + 1, d1, c1r.pad);
            }
        };
        // Slots past the window hold a threshold of 2 (nothing passes), so an odd tail needs no test.
        ObjF a0r = curf[i_lo], a1r = curf[i_lo + 1];
        for (int i = i_lo; i < i_hi; i += 4) {
            const int ib = i + 2 < MSTAGE ? i + 2 : MSTAGE - 2;
            const ObjF b0r = curf[ib], b1r = curf[ib + 1];
            trip(i, a0r, a1r);
            const int ia = i + 4 < MSTAGE ? i + 4 : MSTAGE - 2;
            a0r = curf[ia];
            a1r = curf[ia + 1];
            if (i + 2 < i_hi) trip(i + 2, b0r, b1r);
        }
        drain();  // codes refer to this stage's slots
#pragma unroll
        for (int f = 0; f < NPF; ++f)
            if (have_next && f * MWG + tid < MSTAGE) stagef[(cb ^ 1) * MSTAGE + f * MWG + tid] = finish(nxt[f]);
        __syncthreads();
        if (!WEIGHTED && (st & MERGED_FLUSH_MASK) == MERGED_FLUSH_MASK && have_next) {  // 32-bit counters: see k_count
            flush_hist();
            __syncthreads();
        }
    }
    if (!WEIGHTED && nstages > MERGED_FLUSH_MASK / 4 && win + 1 < it.nwin) {  // a long window: do not carry its counts into the next
        flush_hist();
        __syncthreads();
    }
    }

    flush_hist();
}

// The kernel proper. With one or two objects per lane the body fits 64 VGPRs without spilling, so the compiler is
// told to keep 8 waves per SIMD (80 VGPRs / 6 waves otherwise: -9 % time at the headline); with four objects per
// lane that limit would spill, the default allocation stays.
#define YAW_COUNT_MERGED_ARGS                                                                                         \
    const DevTab *__restrict__ tabs, const Item *__restrict__ items, int n_bins,                                      \
        int n_edges, const double *__restrict__ t, const float *__restrict__ dthr, const double *__restrict__ rwin_k, \
        int64_t item_base, unsigned long long *__restrict__ out_counts, double *__restrict__ partials,                \
        const unsigned long long *__restrict__ counters
#define YAW_COUNT_MERGED_PASS tabs, items, n_bins, n_edges, t, dthr, rwin_k, item_base, out_counts, partials, counters
template <int R, bool WEIGHTED, bool NF1, bool MERGED>
__global__ __launch_bounds__(MWG) void k_count_merged(YAW_COUNT_MERGED_ARGS) {
    count_merged_body<R, WEIGHTED, NF1, MERGED>(YAW_COUNT_MERGED_PASS);
}
template <int R, bool WEIGHTED, bool NF1, bool MERGED>
__global__ __launch_bounds__(MWG) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_count_merged_occ8(YAW_COUNT_MERGED_ARGS) {
    count_merged_body<R, WEIGHTED, NF1, MERGED>(YAW_COUNT_MERGED_PASS);
}

template <int R, bool WEIGHTED, bool NF1, bool MERGED>
auto pick_count_merged() -> void (*)(YAW_COUNT_MERGED_ARGS) {
    if constexpr (R <= 2) return k_count_merged_occ8<R, WEIGHTED, NF1, MERGED>;
    else return k_count_merged<R, WEIGHTED, NF1, MERGED>;
}

// ------------------------------------------------------------------------------------------------
// Band kernel (BAND, the default). Same work items as k_count_merged -- (lane tile of 64*R consecutive objects of a
// c2 run) x (the window of a c1 run that can hold their partners) -- but the window is not evaluated as a full
// rectangle. Both sides are sorted along u, so the partners of ONE lane object inside the window are the contiguous
// index range with |du| <= r_win: a band of ~2 r_win * (objects per unit u) entries, of which about half are real
// pairs when the strips are about r_win wide (disc / bounding box), against 64*R + band entries per lane object for
// the rectangle. So:
//   * the window is staged in LDS once, as float64 SoA columns (x, y, z, bin id, weight), BCAP objects at a time;
//   * every lane object finds its own band [lo, hi) by two branch-free binary searches on the staged sort-axis column;
//   * the lane walks its band: per step one staged object per lane (consecutive lanes read consecutive addresses:
//     conflict-free ds_read_b64), the exact float64 predicate, and the histogram update -- no float32 pre-filter,
//     no survivor queue, no gathers from global memory.
// At the headline density a lane object meets ~14 entries per strip instead of ~200; every evaluated entry costs
// 8 FP64 operations, which is what the parity contract asks for anyway.
// Work distribution: the grid is sized on the host from the number of POTENTIAL items (no host round trip for the
// number the builder kept); a workgroup takes the kept items v = blockIdx.x, blockIdx.x + gridDim.x, ... and maps v
// to the item list so that the workgroups of one XCD (blockIdx.x mod 8, MI355X_MICROARCH.md "Workgroup dispatch")
// walk one contiguous eighth of the list: consecutive items share c1 runs and lane tiles, which then hit in that
// XCD's L2 instead of being fetched by all eight.
//   UNI: all redshift bins share one threshold row (angular scales): edges live in registers.
// ------------------------------------------------------------------------------------------------
#ifndef YAW_BCAP
#define YAW_BCAP 192
#endif
constexpr int BCAP = YAW_BCAP;  // window objects per LDS stage. 192: the window of a 128-object lane tile at equal densities (128 +- 11
                                // entries + one band) fits in one stage; 6.2 KB -> 26 single-wave workgroups per CU. Measured
                                // 160 / 176 / 192 / 208 / 224: count kernel 0.545 / 0.529 / 0.523 / 0.535 / 0.531 ms at the headline
constexpr int N_CTR = 8 + 8 * EVAL_SLOTS;  // counters: [0] kept items, [8 + 8 i] band entries, [9 + 8 i] exact re-evaluations, [10 + 8 i] lane-tile x window pairs

// LDS image of a band workgroup. The staged window lives in float64 SoA columns filled by LDS-DMA (global_load_lds_dwordx4:
// 16 bytes per lane straight from HBM into LDS, no staging registers, no ds_write), one entry of slack per column for the
// sentinel. The columns sit in a STATIC array at fixed offsets 2064 bytes apart, with the small tables in the gaps:
//   * the walk addresses x, y, z and the bin id from one register with immediate offsets;
//   * two ds_read_b64 less than 2041 bytes, or a multiple of 512 bytes, apart would be fused into ds_read2_b64 /
//     ds_read2st64_b64, which take twice their LDS cycles.
// Weights, edge tables and histograms too large for the gaps follow in dynamic LDS.
constexpr int band_apart(int at_least, int from1, int from2) {  // next 16-byte aligned offset >= at_least that is no multiple of
    int v = (at_least + 15) & ~15;                              // 512 bytes away from from1 and from2 (and >= 2048 away: caller)
    while ((v - from1) % 512 == 0 || (v - from2) % 512 == 0) v += 16;
    return v;
}
constexpr int band_max(int a, int b) { return a > b ? a : b; }
template <int CAP>
struct BandLds {  // byte offsets inside the static LDS image of a band workgroup whose stage holds CAP entries
    static constexpr int COL = (CAP + 2) * 8;   // a float64 column: CAP entries + sentinel, 16-byte multiple
    static constexpr int KCOL = (CAP + 4) * 4;  // the bin-id column
    static constexpr int X = 0, K = COL;
    static constexpr int Y = band_apart(band_max(K + KCOL, X + 2048), X, X);
    static constexpr int H = Y + COL;           // small histogram (H_BYTES), then 64 dummy cells
    static constexpr int H_BYTES = 512;
    static constexpr int Z = band_apart(band_max(H + H_BYTES + 256, Y + 2048), Y, X);
    static constexpr int FIXED = Z + COL;
};
static_assert(BandLds<160>::Y == 2064 && BandLds<160>::Z == 4128 && BandLds<160>::FIXED == 5424, "band LDS image");
static_assert(BandLds<192>::FIXED == 6208, "band LDS image");
// Stage capacities the band kernel is compiled for. A stage should hold the whole window of a lane tile (the tile's own
// extent in streamed entries plus one band): a window cut into stages makes every stage wait for the longest clipped band
// while the lanes whose bands lie in the other stage idle (50M x 50M, bands of 216 entries: 400 trips per 256 lane objects
// with 288-entry stages against 219 in one stage).
// (416-entry stages were measured too: never ahead of 288 -- 23.3 / 23.3 ms at 50M x 50M, 0.70 / 0.59 ms at the headline.)
constexpr int BCAP_MID = 288;
inline int band_lds_fixed(int cap) { return cap == BCAP_MID ? BandLds<BCAP_MID>::FIXED : BandLds<BCAP>::FIXED; }
__host__ __device__ inline bool band_small_hist(bool weighted, int nslots, int hp) { return (size_t)nslots * hp * (weighted ? 8 : 4) <= 512; }
// dynamic LDS bytes of a band workgroup (host and device agree through this one function)
__host__ __device__ inline size_t band_lds_dynamic(bool weighted, bool need_thr, int nkb, int n_edges, int hp, int cap, int thr_rows) {
    const int nslots = nkb * (n_edges - 1);
    return (weighted ? (size_t)(cap + 2) * 8 : 0) + (need_thr ? (size_t)thr_rows * n_edges * sizeof(double) : 0) +
           (band_small_hist(weighted, nslots, hp) ? 0 : (size_t)nslots * hp * (weighted ? 8 : 4)) + 16;
}
typedef __attribute__((address_space(3))) unsigned char lds_byte;
__device__ __forceinline__ __attribute__((address_space(3))) void *lds_ptr(unsigned addr) { return (__attribute__((address_space(3))) void *)(size_t)addr; }
__device__ __forceinline__ double lds_f64(unsigned addr) { return *(const __attribute__((address_space(3))) double *)(size_t)addr; }
__device__ __forceinline__ int lds_i32(unsigned addr) { return *(const __attribute__((address_space(3))) int *)(size_t)addr; }

// Maximum of a non-negative int over the 64 lanes of the wave, as a scalar: four row shifts, two row broadcasts (DPP, in
// the VALU) and one readlane -- the shuffle form goes through the LDS crossbar six times, each with its own wait.
__device__ __forceinline__ int wave_max_nonneg(int v) {
    // update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl): lanes without a source lane keep old = 0, the identity
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));  // row_shr:8 -> lane 15 of a row holds the row's maximum
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1 and 3
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// Sum of an unsigned int over the 64 lanes, in lane 63 (same DPP steps; lanes without a source add 0).
__device__ __forceinline__ unsigned int wave_sum_lane63(unsigned int v) {
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// NE: edges per bin known at compile time (2: one fine bin; 3, 4: edges in registers when every bin -- or the item -- has one
// row of them); 0: any number, edge table in LDS.
template <int R, int CAP, bool WEIGHTED, int NE, bool MERGED, bool UNI>
__global__ __launch_bounds__(64) void k_count_band(const DevTab *__restrict__ tabs, const Item *__restrict__ items, int n_bins,
                                                   int n_edges, const double *__restrict__ t,
                                                   const double *__restrict__ rwin_k, unsigned flush_mask, int hp_shift,
                                                   int batch_log2, unsigned long long *__restrict__ out_counts,
                                                   double *__restrict__ partials,
                                                   unsigned long long *__restrict__ counters) {
    using HistT = typename std::conditional<WEIGHTED, double, unsigned int>::type;
    constexpr bool NF1 = NE == 2;
    constexpr bool REG_EDGES = NE >= 2 && (!MERGED || UNI);  // the item's edges live in registers
    constexpr bool NEED_THR = !REG_EDGES;                    // else: edge table in LDS
    constexpr int HB = WEIGHTED ? 3 : 2;                 // log2 of the bytes of a histogram cell
    using L = BandLds<CAP>;
    constexpr int LDS_X = L::X, LDS_K = L::K, LDS_Y = L::Y, LDS_H = L::H, LDS_Z = L::Z, LDS_H_BYTES = L::H_BYTES, BCOL = L::COL;
    __shared__ __attribute__((aligned(16))) unsigned char lds_fix[L::FIXED];
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
    const int nkb = MERGED ? n_bins : 1;  // bins one item can add to
    const int nf = NF1 ? 1 : n_edges - 1;
    const int nslots = nkb * nf;
    const int hp = 1 << hp_shift;  // copies of the histogram (lanes spread over them: fewer same-address LDS atomics)
    double *sx = reinterpret_cast<double *>(lds_fix + LDS_X);
    double *sy = reinterpret_cast<double *>(lds_fix + LDS_Y);
    double *sz = reinterpret_cast<double *>(lds_fix + LDS_Z);
    int *sk = reinterpret_cast<int *>(lds_fix + LDS_K);
    unsigned char *p = lds_dyn;
    double *sw = reinterpret_cast<double *>(p); if (WEIGHTED) p += BCOL;
    constexpr bool ROW1 = !(MERGED && !UNI);          // one edge row: every bin (or the item's only bin) shares it
    const int thr_rows = ROW1 ? 1 : nkb;
    double *thr = reinterpret_cast<double *>(p); if (NEED_THR) p += (size_t)thr_rows * n_edges * sizeof(double);  // [thr_rows][n_edges]
    const bool small_hist = band_small_hist(WEIGHTED, nslots, hp);
    HistT *hist = reinterpret_cast<HistT *>(small_hist ? lds_fix + LDS_H : p);                                // [nslots][hp]
    unsigned int *dummy = reinterpret_cast<unsigned int *>(lds_fix + LDS_H + LDS_H_BYTES);                    // [64] one cell per lane for misses
    const int lane = threadIdx.x;
    // LDS addresses (32 bit) the walk works with
    const unsigned a_sx = (unsigned)(size_t)(lds_byte *)(lds_fix + LDS_X);
    const unsigned a_sw = (unsigned)(size_t)(lds_byte *)reinterpret_cast<unsigned char *>(sw);
    const unsigned a_cell = (unsigned)(size_t)(lds_byte *)reinterpret_cast<unsigned char *>(hist) + ((lane & (hp - 1)) << HB);  // this lane's copy of cell 0
    const unsigned a_dummy = (unsigned)(size_t)(lds_byte *)reinterpret_cast<unsigned char *>(dummy + lane);
    const int ksh = hp_shift + HB;                  // slot number -> byte offset of its first histogram cell

    const unsigned long long n_kept = counters[0];
    const unsigned long long chunk = (n_kept + 7) >> 3;  // items per XCD
    // A workgroup takes 2^batch_log2 CONSECUTIVE items at a time (default: one) and carries its (unweighted) histogram from
    // one to the next while they add to the same output slot -- consecutive items are lane tiles of one job -- so that a
    // histogram of hundreds of cells goes to global memory once per batch. Off by default, see the launch.
    const unsigned long long n_batches = (chunk + (1ull << batch_log2) - 1) >> batch_log2;
    for (int e = lane; e < nslots * hp; e += 64) hist[e] = HistT(0);  // every flush leaves the histogram zeroed again
    unsigned int cnt1 = 0;             // NF1 && !MERGED: the only counter lives in a register
    unsigned stage_no = 0;
    int pend_slot = -1;                // output slot the histogram holds counts for (unweighted)
    int thr_k = -1;                    // bin whose edge row the LDS table holds
    auto flush_counts = [&](int slot_out) {  // LDS histogram / register counter -> global result (unweighted)
        if (NF1 && !MERGED) {
            if (lane == 0 && cnt1) atomicAdd(&out_counts[(int64_t)slot_out * nslots], (unsigned long long)cnt1);  // every lane holds the wave total
            cnt1 = 0;
            return;
        }
        __syncthreads();
        for (int idx = lane; idx < nslots; idx += 64) {
            unsigned int c = 0;
            for (int h = 0; h < hp; ++h) {
                c += (unsigned int)hist[idx * hp + h];
                hist[idx * hp + h] = HistT(0);
            }
            if (c) atomicAdd(&out_counts[(int64_t)slot_out * nslots + idx], (unsigned long long)c);
        }
    };
    for (unsigned long long v = blockIdx.x;; v += gridDim.x) {
        if ((v >> 3) >= n_batches) break;
      for (unsigned long long j = (v >> 3) << batch_log2; j < (((v >> 3) + 1) << batch_log2) && j < chunk; ++j) {
        const unsigned long long ticket = (v & 7) * chunk + j;
        if (ticket >= n_kept) break;  // short last eighth
        const Item it = items[ticket];
        const int o = item_orient(it), islot = item_slot(it);
        if (!WEIGHTED) {
            if (pend_slot >= 0 && pend_slot != islot) flush_counts(pend_slot);
            pend_slot = islot;
        }
        const DevTab c1 = tabs[o], c2 = tabs[3 + o];  // wave-uniform: scalar loads
        const int kfix = MERGED ? 0 : islot % n_bins;
        const double rwin = rwin_k[kfix];
        int64_t b0 = it.b0[0], nb_total = it.nb[0];  // the current window (an item of the strip builder carries up to three)

        __syncthreads();  // the previous item of this workgroup has left the LDS
        // stage of the window -> LDS, 16 bytes per lane and instruction; lanes past the stage stay out of it.
        // (wave-uniform 64-bit bases + 32-bit lane offsets: the loads take their base from scalar registers)
        auto stage_in = [&](int64_t first, int n) {
            const gf64p gx = c1.x + b0 + first, gy = c1.y + b0 + first, gz = c1.z + b0 + first;
#pragma unroll
            for (int c = 0; c < (CAP + 127) / 128; ++c) {
                const unsigned e = (unsigned)(c * 128 + 2 * lane);
                if (e < (unsigned)n) {
                    __builtin_amdgcn_global_load_lds(gx + e, lds_ptr(a_sx + c * 1024), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(gy + e, lds_ptr(a_sx + (LDS_Y - LDS_X) + c * 1024), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(gz + e, lds_ptr(a_sx + (LDS_Z - LDS_X) + c * 1024), 16, 0, 0);
                    if (WEIGHTED && c1.w) __builtin_amdgcn_global_load_lds(c1.w + b0 + first + e, lds_ptr(a_sw + c * 1024), 16, 0, 0);
                }
            }
            if (MERGED) {
#pragma unroll
                for (int c = 0; c < (CAP + 255) / 256; ++c) {
                    const unsigned e = (unsigned)(c * 256 + 4 * lane);
                    if (e < (unsigned)n)
                        __builtin_amdgcn_global_load_lds(c1.k + b0 + first + e, lds_ptr(a_sx + (LDS_K - LDS_X) + c * 1024), 16, 0, 0);
                }
            }
        };
        stage_in(0, (int)(nb_total < CAP ? nb_total : CAP));
        // lane objects and thresholds while the stage is in flight
        // A lane holds R NEIGHBOURING objects of the (u-sorted) tile: their bands overlap almost completely, so the lane
        // walks their union once -- one read of an entry serves R evaluations, two searches serve R objects.
        double ax[R], ay[R], az[R], aw[R];
        int n_own = (int)it.na - lane * R;  // objects of this lane (<= 0: none)
        n_own = n_own < 0 ? 0 : (n_own > R ? R : n_own);
        {
            const gf64p px = c2.x + it.a0, py = c2.y + it.a0, pz = c2.z + it.a0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool have = r < n_own;
                const unsigned ic = have ? (unsigned)(lane * R + r) : 0u;
                ax[r] = px[ic]; ay[r] = py[ic]; az[r] = pz[ic];
                aw[r] = (WEIGHTED && c2.w) ? (c2.w + it.a0)[ic] : 1.0;
            }
        }
        double ed[NE >= 2 ? NE : 1];  // edges of the item's bin (or of every bin) in registers
        if (REG_EDGES) {
#pragma unroll
            for (int q = 0; q < NE; ++q) ed[q] = t[(int64_t)kfix * n_edges + q];
        }
        if (NEED_THR && thr_k != kfix) {  // (the barriers of the first stage come before anyone reads it)
            for (int e = lane; e < thr_rows * n_edges; e += 64) thr[e] = t[(int64_t)kfix * n_edges + e];
            thr_k = kfix;
        }
        // outer edges of the one row in registers; largest power of two <= the number of inner edges (fine-bin search)
        const double e_lo = NEED_THR && ROW1 ? t[(int64_t)kfix * n_edges] : 0.0;
        const double e_hi = NEED_THR && ROW1 ? t[(int64_t)kfix * n_edges + n_edges - 1] : 0.0;
        const int edge_top = n_edges > 2 ? 1 << (31 - __builtin_clz(n_edges - 2)) : 0;
        // the lane's band: from the first object's lower to the last object's upper bound (the tile is sorted along u)
        double klo, khi;
        {
            const int last_r = n_own > 0 ? n_own - 1 : 0;
            double u_first = c2.axis == 0 ? ax[0] : (c2.axis == 1 ? ay[0] : az[0]), u_last = u_first;
#pragma unroll
            for (int r = 1; r < R; ++r)
                if (r == last_r) u_last = c2.axis == 0 ? ax[r] : (c2.axis == 1 ? ay[r] : az[r]);
            klo = u_first - rwin;
            khi = u_last + rwin;
        }
#pragma unroll
        for (int r = 0; r < R; ++r)  // slots without an object sit beyond every edge of every entry
            if (r >= n_own) { ax[r] = PAD_COORD; ay[r] = PAD_COORD; az[r] = PAD_COORD; aw[r] = 0.0; }
        unsigned int nev = 0;              // band entries this lane evaluated
        auto flush_slab = [&]() {  // weighted: LDS histogram -> the item's slab
            __syncthreads();
            for (int idx = lane; idx < nslots; idx += 64) {
                double c = 0.0;
                for (int h = 0; h < hp; ++h) {  // fixed order: reproducible weighted sums
                    c += (double)hist[idx * hp + h];
                    hist[idx * hp + h] = HistT(0);
                }
                partials[(int64_t)it.pot * nslots + idx] = c;
            }
        };

        for (int win = 0; win < it.nwin; ++win) {
        if (win > 0) {
            b0 = win == 1 ? it.b0[1] : it.b0[2];
            nb_total = win == 1 ? it.nb[1] : it.nb[2];
        }
        for (int64_t st0 = 0; st0 < nb_total; st0 += CAP, ++stage_no) {
            const int n = (int)(nb_total - st0 < CAP ? nb_total - st0 : CAP);
            if (st0 > 0 || win > 0) {
                __syncthreads();  // every lane is done with the previous stage
                stage_in(st0, n);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stage has landed: only the wave's own vmcnt orders LDS reads behind its LDS-DMA
            __syncthreads();
            if (lane == 0) {  // sentinel behind the stage: lanes whose band has ended read it; it is beyond every edge
                sx[n] = PAD_COORD; sy[n] = PAD_COORD; sz[n] = PAD_COORD;
                if (WEIGHTED) sw[n] = 0.0;
                if (MERGED) sk[n] = 0;
            }
            if (WEIGHTED && !c1.w)
                for (int e = lane; e < n; e += 64) sw[e] = 1.0;
            __syncthreads();
            // band of the lane inside this stage: [lo, hi) = entries with klo <= key <= khi
            // Both searches run on LDS byte addresses: q = address of entry (lo - 1). A probe beyond the stage is clamped
            // onto the sentinel (4.0 > every key bound), which fails both comparisons by itself -- no index checks, and
            // hipcc keeps the loop free of branches (the index form compiled to twice the instructions, with the
            // second read under an exec branch).
            const unsigned a_key = a_sx + (c2.axis == 0 ? 0u : (c2.axis == 1 ? (unsigned)(LDS_Y - LDS_X) : (unsigned)(LDS_Z - LDS_X)));
            const unsigned a_sent = a_key + ((unsigned)n << 3);
            unsigned ql = a_key - 8u, qh = ql;
#if defined(YAW_BAND_DIAG) && YAW_BAND_DIAG >= 2
            for (unsigned step8 = 0; step8 >= 8u; step8 >>= 1) {  // diagnostics: no search either
#else
            for (unsigned step8 = 8u << (31 - __builtin_clz(n)); step8 >= 8u; step8 >>= 1) {  // largest power of two <= n
#endif
                const unsigned pl = ql + step8, ph = qh + step8;
                const double kl = lds_f64(pl < a_sent ? pl : a_sent), kh = lds_f64(ph < a_sent ? ph : a_sent);
                ql = kl < klo ? pl : ql;    // entries [0, lo) have key <  klo
                qh = kh <= khi ? ph : qh;   // entries [0, hi) have key <= khi
            }
            int lo = (int)((ql + 8u - a_key) >> 3), hi = (int)((qh + 8u - a_key) >> 3);
            if (n_own == 0) lo = hi = n;  // lanes without an object walk the sentinel
            int len = hi - lo;
            nev += (unsigned int)(len * n_own);
#if defined(YAW_BAND_DIAG) && YAW_BAND_DIAG >= 1
            const int steps = 0;  // diagnostics: everything but the walk (wrong counts)
#else
            const int steps = wave_max_nonneg(len);  // the longest band of the wave: uniform trip count
#endif

            // Walk the band, one entry per trip, evaluated against the lane's R objects. A lane whose band has ended moves
            // on through the window (entries beyond a band have |du| > r_win, hence s > every upper edge: they fail the
            // predicate by themselves, as do the band's entries that lie outside the narrower band of one of the R
            // objects) and parks on the sentinel. Occupancy, not a software pipeline inside the wave, covers the LDS
            // latency of a trip (the pipeline would cost the registers that occupancy needs).
            unsigned cur = a_sx + ((unsigned)lo << 3);  // LDS address of the next entry's x; y, z, bin id at fixed distances
            const unsigned last = a_sx + ((unsigned)n << 3);
            for (int s = 0; s < steps; ++s) {
                const unsigned a8 = cur < last ? cur : last;
                cur += 8;
                const double ex = lds_f64(a8);
                const double ey = lds_f64(a8 + (LDS_Y - LDS_X));
                const double ez = lds_f64(a8 + (LDS_Z - LDS_X));
                const double ew = WEIGHTED ? lds_f64(a8 - a_sx + a_sw) : 1.0;
                const int ek = MERGED ? lds_i32(((a8 - a_sx) >> 1) + a_sx + (LDS_K - LDS_X)) : 0;
                // edge row of the entry's bin and its outer edges: once per entry, not per evaluation, and without
                // conditions around the reads (hipcc turned `a && b` over two LDS reads into exec branches with full waits)
                const double *tk = thr + ((NEED_THR && !ROW1) ? ek * n_edges : 0);
                const double t_first = (NEED_THR && !ROW1) ? tk[0] : e_lo, t_last = (NEED_THR && !ROW1) ? tk[n_edges - 1] : e_hi;
                if constexpr (WEIGHTED && NF1 && R > 1) {
                    // One fine bin: the R evaluations of an entry add to the same cell (the entry's bin), and neighbouring
                    // objects mostly hit the same entries -- their products are summed in registers and go to the LDS as ONE
                    // float64 atomic (those are slow: ~1 lane per cycle). Fixed order inside the lane: still reproducible.
                    double acc = 0.0;
                    bool any = false;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const double dx = ax[r] - ex, dy = ay[r] - ey, dz = az[r] - ez;
                        const double xx = dx * dx, yy = dy * dy, zz = dz * dz;
                        const double sxy2 = xx + yy;
                        const double sd = sxy2 + zz;
                        const bool in = REG_EDGES ? (sd > ed[0]) & (sd <= ed[NE >= 2 ? NE - 1 : 0]) : (sd > t_first) & (sd <= t_last);
                        acc += in ? aw[r] * ew : 0.0;
                        any |= in;
                    }
                    if (any)
                        (void)__hip_atomic_fetch_add((__attribute__((address_space(3))) double *)(size_t)(((unsigned)ek << ksh) + a_cell), acc,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else if constexpr (!REG_EDGES && !NF1) {
                    // Many edges (separation weights: ~50 fine bins): the fine bins of the R evaluations are searched in
                    // lockstep and for every lane -- R independent chains of LDS reads instead of one after the other under a
                    // branch; a miss ends in some valid bin and goes to the dummy cell.
                    double sdv[R];
                    bool inv[R];
                    int c[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const double dx = ax[r] - ex, dy = ay[r] - ey, dz = az[r] - ez;
                        const double xx = dx * dx, yy = dy * dy, zz = dz * dz;
                        const double sxy2 = xx + yy;
                        sdv[r] = sxy2 + zz;
                        inv[r] = (sdv[r] > t_first) & (sdv[r] <= t_last);
                        c[r] = 0;
                    }
                    for (int step = edge_top; step > 0; step >>= 1) {  // fine bin = number of inner edges below s
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const int probe = c[r] + step;
                            c[r] = sdv[r] > tk[probe < n_edges - 1 ? probe : n_edges - 1] ? probe : c[r];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const unsigned cell = ((unsigned)(ek * nf + c[r]) << ksh) + a_cell;  // t[c] < s <= t[c + 1]
                        if constexpr (WEIGHTED) {
                            if (inv[r])
                                (void)__hip_atomic_fetch_add((__attribute__((address_space(3))) double *)(size_t)cell, aw[r] * ew,
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        } else {
                            (void)__hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned int *)(size_t)(inv[r] ? cell : a_dummy),
                                                         1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    double bx[R], by[R], bz[R], bw[R];
                    int kb[R];
                    bx[r] = ex; by[r] = ey; bz[r] = ez; bw[r] = ew; kb[r] = ek;
                    const double dx = ax[r] - bx[r];
                    const double dy = ay[r] - by[r];
                    const double dz = az[r] - bz[r];
                    const double xx = dx * dx;
                    const double yy = dy * dy;
                    const double zz = dz * dz;
                    const double sxy2 = xx + yy;
                    const double sd = sxy2 + zz;
                    bool in;
                    int slot = kb[r] * nf;
                    if constexpr (REG_EDGES) {
                        in = sd > ed[0] && sd <= ed[NE - 1];
                        if constexpr (NE >= 3) slot += (sd > ed[1]) ? 1 : 0;  // inner edges: t[c-1] < s <= t[c]
                        if constexpr (NE >= 4) slot += (sd > ed[2]) ? 1 : 0;
                    } else {
                        in = (sd > t_first) & (sd <= t_last);
                    }
                    if (NF1 && !MERGED && !WEIGHTED) {
                        cnt1 += (unsigned int)__popcll(__builtin_amdgcn_ballot_w64(in));
                    } else if ((REG_EDGES || NF1) && !WEIGHTED) {  // the slot is known without a search over the edges
                        // Branch-free: a miss adds to the lane's own dummy cell. (Under a branch the compiler can no longer
                        // count the LDS operations in flight and drains them all before every evaluation.)
                        const unsigned cell = in ? ((unsigned)slot << ksh) + a_cell : a_dummy;
                        (void)__hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned int *)(size_t)cell, 1u,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else if (in) {
                        if constexpr (!REG_EDGES) {
                            // fine bin = number of inner edges below s: a branch-free binary search over the sorted row
                            // (a probe past the inner edges reads the last edge, which s does not exceed)
                            int c = 0;
                            for (int step = edge_top; step > 0; step >>= 1) {
                                const int probe = c + step;
                                c = sd > tk[probe < n_edges - 1 ? probe : n_edges - 1] ? probe : c;
                            }
                            slot += c;  // t[c] < s <= t[c + 1]
                        }
                        // The histogram belongs to this wave alone: integer adds are exact; float64 adds of ONE instruction
                        // that hit the same cell are serialised by the LDS in a fixed lane order -> reproducible sums.
                        const unsigned cell = ((unsigned)slot << ksh) + a_cell;
                        if constexpr (WEIGHTED)
                            (void)__hip_atomic_fetch_add((__attribute__((address_space(3))) double *)(size_t)cell, aw[r] * bw[r],
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else
                            (void)__hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned int *)(size_t)cell, 1u,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                }
            }
            // 64*R lane objects x BCAP entries per stage: a uint32 counter cannot wrap within flush_mask + 1 stages
            if (!WEIGHTED && (stage_no & flush_mask) == flush_mask) flush_counts(islot);
        }
        }
        if (WEIGHTED) flush_slab();
        // evaluated band entries of the item -> one of EVAL_SLOTS counters (statistics)
        for (int off = 32; off > 0; off >>= 1) nev += __shfl_down(nev, off, 64);
        if (lane == 0 && nev) atomicAdd(&counters[8 + 8 * (ticket & (EVAL_SLOTS - 1))], (unsigned long long)nev);
      }
        if (!WEIGHTED && pend_slot >= 0) {  // end of the batch
            flush_counts(pend_slot);
            pend_slot = -1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Band kernel, float32 classification (BAND on strip layouts of unit vectors).
// Same items and the same walk as k_count_band, but an entry is decided in FLOAT32 wherever float32 can decide it:
//   * both catalogues keep a float32 image of every strip layout, 16 bytes per object {x, y, z, bin id} (DevTab::q);
//     the window is staged from it by LDS-DMA (one dwordx4 per entry) and read back with one ds_read_b128;
//   * s32 = fma(dz, dz, fma(dy, dy, dx * dx)) on the float32 images, two lane objects per packed instruction
//     (v_pk_add / v_pk_mul / v_pk_fma_f32: 7 VALU for two evaluations instead of 16 FP64 operations);
//   * for unit vectors |s32 - s| <= g(t) = 2.1e-7 sqrt(t) + 5e-7 t + 1e-12 around an edge t (rounding of the images to
//     float32: 2^-25 per coordinate -- |x| <= 1, and values a hair above 1 round to 1 --, of the differences: 2^-24
//     relative, hence |error of d_i| <= 6e-8 (1 + |d_i|) and 2 |d| sqrt(3) 6e-8 on the sum of squares; of the three-term
//     sum: 3 x 2^-24 relative), so s32 < t - g proves s <= t and
//     s32 > t + g proves s > t. The host turns every edge into the two float32 bounds (thr32, build_thr32);
//   * an evaluation that lands inside a guard band (~1e-4 of them at the headline) is UNCERTAIN: the wave branches, the
//     lanes concerned fetch both objects in float64 and apply the exact predicate of the parity contract
//     (((dx*dx + dy*dy) + dz*dz) against the host's float64 thresholds). Every pair is therefore decided exactly as
//     k_count_band decides it: results are bit-identical.
// The roles are swapped against k_count_band where one side is binned (MERGED, the cross-correlation counts): the
// lanes hold the BINNED objects (c1), the window streams the unbinned side (c2). A lane then knows the bin, the edge
// row and the counter of each of its objects: hits are counted in registers (add with carry) and go to the LDS
// histogram once per item, not once per evaluation.
//   NE == 2 (one annulus): classes by |s32 - c| against two half widths (certainly inside / possibly inside);
//   NE  > 2: cumulative counters per edge (s <= t_e), fine bin j = cum[j + 1] - cum[j] at the flush.
// ------------------------------------------------------------------------------------------------
// Items -> XCDs (float32 band kernels). Workgroup v runs on XCD v & 7 and takes the items of that XCD one after the other.
// The list arrives in the order of the jobs -- diagonal jobs first (dense lane tiles), then the tiles at patch borders
// (windows that reach a few lanes only) -- so eight contiguous eighths would give six XCDs the heavy items and two the light
// ones. The list is cut into BLOCKS of 2^bs consecutive items (consecutive items are neighbouring lane tiles: their windows
// overlap, which is what an XCD's L2 is for) and the blocks are dealt round robin to the XCDs.
// A list the builder kept in segments (append_items, seg_cap > 0) needs none of this: XCD x takes segment x.
struct TicketMap {
    unsigned long long per_xcd;  // tickets an XCD walks through (multiple of the block size)
    unsigned long long n_kept;   // tickets below this are items (segmented: end of this XCD's segment)
    unsigned long long seg_base; // segmented: first record of this XCD's segment
    unsigned bs;                 // log2 of the block size
    bool segmented;
    __device__ __forceinline__ unsigned long long ticket(unsigned long long v) const {
        const unsigned long long j = v >> 3;
        if (segmented) return seg_base + j;
        return ((((j >> bs) << 3) + (v & 7)) << bs) + (j & ((1ull << bs) - 1ull));
    }
};
__device__ __forceinline__ TicketMap ticket_map(const unsigned long long *__restrict__ counters, unsigned long long seg_cap) {
    TicketMap m;
    m.segmented = seg_cap != 0;
    if (m.segmented) {  // (gridDim.x is a multiple of 8: a workgroup stays with its XCD)
        const int seg = (int)(blockIdx.x % ITEM_SEGS);
        const unsigned long long n_seg = counters[ITEM_SEG_CTR(seg)];
        m.seg_base = (unsigned long long)seg * seg_cap;
        m.per_xcd = n_seg;
        m.n_kept = m.seg_base + n_seg;
        m.bs = 0;
        return m;
    }
    const unsigned long long n_kept = counters[0];
    m.n_kept = n_kept;
    m.seg_base = 0;
    const unsigned long long want = n_kept >> 7;  // ~16 blocks per XCD
    int bs = want > 1 ? 63 - __builtin_clzll(want) : 0;
    bs = bs < 6 ? 6 : (bs > 12 ? 12 : bs);
    m.bs = (unsigned)bs;
    const unsigned long long nblocks = (n_kept + (1ull << bs) - 1ull) >> bs;
    m.per_xcd = ((nblocks + 7ull) >> 3) << bs;
    return m;
}

constexpr float PAD_COORD32 = 4.0f;
constexpr double BAND32_GUARD_SQRT = 2.1e-7;  // coefficient of sqrt(t) in the float32 guard g(t), see k_count_band32
// float32 words per bin of the threshold table: NE == 2: {c, h_in, h_out, 0}; else per edge {t - g, t + g}
__host__ __device__ constexpr int thr32_width(int ne) { return ne == 2 ? 4 : 2 * ne; }
// dynamic LDS of a k_count_band32 workgroup (host and device agree through this one function)
__host__ __device__ inline size_t band32_lds(bool weighted, int cap, int nslots, int thr_rows, int ne) {
    return (size_t)3 * (cap + 4) * 4 + (weighted ? (size_t)(cap + 4) * 8 : 0) + (size_t)nslots * (weighted ? 8 : 4) +
           (size_t)thr_rows * thr32_width(ne) * 4 + 32;
}

// The exact predicate of the parity contract on the float64 columns, for an evaluation the float32 classes left undecided
// (rare: kept out of line so that its addresses and temporaries do not live in the walk loop's registers). The float64
// thresholds the caller will compare with travel in the same round trip as the coordinates: an undecided evaluation stalls
// its wave for ONE memory latency (coordinates, then thresholds one after the other, were two to four; at 51 edges per bin
// one trip in nine of the walk meets an undecided evaluation and the stalls were a third of the kernel's time).
template <int NT>
struct ExactEval {
    double s;        // ((dx*dx + dy*dy) + dz*dz), rounded product by product
    double th[NT];   // tk[0 .. NT)
};
template <int NT>
__device__ __attribute__((noinline)) ExactEval<NT> band32_exact(gf64p lx, gf64p ly, gf64p lz, int64_t li, gf64p sx, gf64p sy, gf64p sz,
                                                               int64_t gi, const double *__restrict__ tk,
                                                               unsigned long long *__restrict__ counters) {
    atomicAdd(counters, 1ull);  // statistics: evaluations decided by the exact predicate (the caller passes one of EVAL_SLOTS
                                // counters: tens of thousands of adds per launch on ONE address cost 0.4 ms at the headline)
    ExactEval<NT> r;
    const double ax = lx[li], ay = ly[li], az = lz[li], bx = sx[gi], by = sy[gi], bz = sz[gi];
#pragma unroll
    for (int e = 0; e < NT; ++e) r.th[e] = tk[e];
    const double dx = ax - bx, dy = ay - by, dz = az - bz;
    const double xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const double sxy = xx + yy;
    r.s = sxy + zz;
    return r;
}

#ifndef YAW_B32_CAP
#define YAW_B32_CAP 320
#endif
#ifndef YAW_B32_CAP_BIG
#define YAW_B32_CAP_BIG 512
#endif
// Stage of k_count_band32 (entries, 12 bytes each + 8 with weights). 320 holds two windows of a typical lane tile (128 objects
// at equal densities: ~142 entries each) -- measured 192 / 288 / 320 / 448 at the headline: 0.367 / 0.355 / 0.350 / 0.360 ms,
// weighted 0.521 / 0.519 / 0.511 / 0.556, RR of config #4 4.65 / 4.44 / 4.41 / 4.89 (the larger the stage, the fewer workgroups
// a CU holds). The big one is for lane tiles whose single window would not fit (denser streamed side, four objects per lane).
constexpr int B32_CAP = YAW_B32_CAP;
constexpr int B32_CAP_BIG = YAW_B32_CAP_BIG;
// The kernel itself: csrc/yawhip_band32.inc, compiled twice -- k_count_band32 stages up to three windows (or pieces of one)
// in a round, k_count_band32_one a single one, for calls whose items all have one window (merged triple runs, items of
// k_build_items): the bookkeeping of two more chunks costs scalar registers (spilled) and instructions per item, 0.283 against
// 0.274 ms at the headline.

#define YAW_B32_NAME k_count_band32
#define YAW_B32_CH 3
#include "yawhip_band32.inc"
#undef YAW_B32_NAME
#undef YAW_B32_CH
#define YAW_B32_NAME k_count_band32_one
#define YAW_B32_CH 1
#include "yawhip_band32.inc"
#undef YAW_B32_NAME
#undef YAW_B32_CH

// ------------------------------------------------------------------------------------------------
// Band kernel for FINE radial grids (separation weights: `resolution` + 1 log-spaced edges per redshift bin,
// reference src/yaw/catalog/trees.py:107-117,358-360). Items, staging, band search and the float32 distance are
// k_count_band32's; what differs is how an evaluation finds its fine bin among ~50:
//   * the edges of a bin are log-spaced, so f = (log2 s32 - log2 t_0) * m puts edge j near f = j: j = round(f), clamped to
//     the table, is the NEAREST edge -- one v_log_f32, one fma, one round instead of a six-step binary search;
//   * the lane reads the float32 bounds {t_j - g, t_j + g} of that edge (g: the guard of k_count_band32) from an LDS table:
//     s32 below the lower bound is in fine bin j - 1, above the upper bound in bin j; the host admits a table only if f is
//     off by less than half a bin everywhere (build_fine32), so the edges on the other side of s32 need no look;
//   * inside the guard band (~1e-3 of the evaluations in range) the exact float64 predicate on the float64 columns decides,
//     against the host's float64 thresholds.
// Hits go to an LDS histogram [bin][fine bin] with one atomic per evaluation (a miss adds to the lane's dummy cell).
// Rows of the float32 table (fine32): {m, a = m log2 t_0, 0, 0}, then {t_j - g, t_j + g} per edge.
// (The first version guessed the bin as floor(f) and was certain only if f kept a distance eps(s32) = e0 + e1 / sqrt(s32)
// from the grid, with a second look at the table otherwise: 32 instruction slots per evaluation against 18 now, and a
// third of the walk's trips took the second look. 2.09 -> 1.25 ms then, -> see DESIGN.md for this one.)
// ------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int fine32_width(int n_edges) { return 4 + 2 * n_edges; }
__host__ __device__ inline size_t band32_fine_lds(bool weighted, int cap, int nslots, int rows, int n_edges) {
    return (size_t)3 * (cap + 4) * 4 + (weighted ? (size_t)(cap + 2) * 8 : 0) + (size_t)nslots * (weighted ? 8 : 4) + 64 * 8 +
           (size_t)rows * fine32_width(n_edges) * 4 + 48;
}

template <int R, int CAP, bool WEIGHTED, bool MERGED, bool UNI>
__global__ __launch_bounds__(64) void k_count_band32_fine(const DevTab *__restrict__ tabs, const Item *__restrict__ items, int n_bins,
                                                          int n_edges, const double *__restrict__ t, const float *__restrict__ fine32,
                                                          const double *__restrict__ rwin_k, unsigned flush_mask, int swap,
                                                          unsigned long long *__restrict__ out_counts,
                                                          double *__restrict__ partials,
                                                          unsigned long long *__restrict__ counters, unsigned long long seg_cap) {
    using HistT = typename std::conditional<WEIGHTED, double, unsigned int>::type;
    constexpr int HB = WEIGHTED ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
    const int nkb = MERGED ? n_bins : 1;
    const int nf = n_edges - 1;
    const int nslots = nkb * nf;
    const int tw = fine32_width(n_edges);
    const int rows = UNI ? 1 : n_bins;
    unsigned char *p = lds_dyn;
    constexpr unsigned COLB = (CAP + 4) * 4;
    float *stage = reinterpret_cast<float *>(p); p += (size_t)3 * COLB;
    double *sw = reinterpret_cast<double *>(p); if (WEIGHTED) p += (size_t)(CAP + 2) * 8;
    HistT *hist = reinterpret_cast<HistT *>(p); p += (size_t)nslots * sizeof(HistT);
    p = reinterpret_cast<unsigned char *>(((size_t)p + 15) & ~(size_t)15);
    double *dummy = reinterpret_cast<double *>(p); p += 64 * 8;  // one cell per lane for misses
    float *stab = reinterpret_cast<float *>(p);                  // [rows][tw]
    const int lane = threadIdx.x;
    const unsigned a_stage = (unsigned)(size_t)(lds_byte *)reinterpret_cast<unsigned char *>(stage);
    const unsigned a_sw = (unsigned)(size_t)(lds_byte *)reinterpret_cast<unsigned char *>(sw);
    const unsigned a_hist = (unsigned)(size_t)(lds_byte *)reinterpret_cast<unsigned char *>(hist);
    const unsigned a_dummy = (unsigned)(size_t)(lds_byte *)reinterpret_cast<unsigned char *>(dummy + lane);

    const TicketMap tmap = ticket_map(counters, seg_cap);
    const unsigned long long n_kept = tmap.n_kept;
    for (int e = lane; e < nslots; e += 64) hist[e] = HistT(0);  // every flush leaves the histogram zeroed again
    for (int e = lane; e < rows * tw; e += 64) stab[e] = fine32[e];
    unsigned stage_no = 0;
    for (unsigned long long v = blockIdx.x;; v += gridDim.x) {
        if ((v >> 3) >= tmap.per_xcd) break;
        const unsigned long long ticket = tmap.ticket(v);
        if (ticket >= n_kept) continue;
        const Item it = items[ticket];
        const int o = item_orient(it), islot = item_slot(it);
        const DevTab cl = tabs[swap ? o : 3 + o], cs = tabs[swap ? 3 + o : o];  // lane side, streamed side
        const int kfix = MERGED ? 0 : islot % n_bins;
        const float rwin = (float)(rwin_k[kfix] * 1.000001 + 4e-7);
        int64_t b0 = it.b0[0], nb_total = it.nb[0];

        __syncthreads();
        auto stage_in = [&](int64_t first, int n) {
            const gf32p gx = cs.qx + b0 + first, gy = cs.qy + b0 + first, gz = cs.qz + b0 + first;
#pragma unroll
            for (int c = 0; c < (CAP + 255) / 256; ++c) {
                const unsigned e = (unsigned)(c * 256 + 4 * lane);
                if (e < (unsigned)n) {
                    __builtin_amdgcn_global_load_lds(gx + e, lds_ptr(a_stage + c * 1024), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(gy + e, lds_ptr(a_stage + COLB + c * 1024), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(gz + e, lds_ptr(a_stage + 2 * COLB + c * 1024), 16, 0, 0);
                }
            }
            if (WEIGHTED && cs.w) {
#pragma unroll
                for (int c = 0; c < (CAP + 127) / 128; ++c) {
                    const unsigned e = (unsigned)(c * 128 + 2 * lane);
                    if (e < (unsigned)n) __builtin_amdgcn_global_load_lds(cs.w + b0 + first + e, lds_ptr(a_sw + c * 1024), 16, 0, 0);
                }
            }
        };
        stage_in(0, (int)(nb_total < CAP ? nb_total : CAP));
        float ax[R], ay[R], az[R];
        double aw[R];
        int kb[R];
        int n_own = (int)it.na - lane * R;
        n_own = n_own < 0 ? 0 : (n_own > R ? R : n_own);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool have = r < n_own;
            const unsigned ic = have ? (unsigned)(lane * R + r) : 0u;
            ax[r] = have ? (cl.qx + it.a0)[ic] : PAD_COORD32;
            ay[r] = have ? (cl.qy + it.a0)[ic] : PAD_COORD32;
            az[r] = have ? (cl.qz + it.a0)[ic] : PAD_COORD32;
            kb[r] = MERGED ? (have ? (cl.k + it.a0)[ic] : 0) : 0;
            aw[r] = (WEIGHTED && cl.w) ? (have ? (cl.w + it.a0)[ic] : 0.0) : (have ? 1.0 : 0.0);
        }
        float klo, khi;
        {
            const int last_r = n_own > 0 ? n_own - 1 : 0;
            float u_first = cl.axis == 0 ? ax[0] : (cl.axis == 1 ? ay[0] : az[0]), u_last = u_first;
#pragma unroll
            for (int r = 1; r < R; ++r)
                if (r == last_r) u_last = cl.axis == 0 ? ax[r] : (cl.axis == 1 ? ay[r] : az[r]);
            klo = u_first - rwin;
            khi = u_last + rwin;
        }
        unsigned int nev = 0;
        auto flush_counts = [&]() {  // LDS histogram -> global result (unweighted)
#if defined(YAW_FINE_DIAG) && YAW_FINE_DIAG == 5
            return;  // diagnostics: no flush (wrong counts)
#endif
            __syncthreads();
            for (int idx = lane; idx < nslots; idx += 64) {
                const unsigned int c = (unsigned int)hist[idx];
                hist[idx] = HistT(0);
                if (c) atomicAdd(&out_counts[(int64_t)islot * nslots + idx], (unsigned long long)c);
            }
        };
        // row of every lane object in the float32 table: model parameters in registers, LDS addresses of its edge bounds
        // and of its row of the histogram
        float pm[R], pa[R];
        unsigned a_edges[R], a_rowh[R];
        const float nf_f = (float)nf;
        const unsigned a_stab = (unsigned)(size_t)(lds_byte *)reinterpret_cast<unsigned char *>(stab);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int trow = (UNI ? 0 : (MERGED ? kb[r] : kfix)) * tw;
            pm[r] = stab[trow]; pa[r] = stab[trow + 1];
            a_edges[r] = a_stab + (unsigned)(trow + 4) * 4u;
            a_rowh[r] = a_hist + ((unsigned)(kb[r] * nf) << HB);
        }
        f32x2 ax2[R / 2 > 0 ? R / 2 : 1], ay2[R / 2 > 0 ? R / 2 : 1], az2[R / 2 > 0 ? R / 2 : 1];  // packed pairs (R even)
        if constexpr (R >= 2) {
#pragma unroll
            for (int h = 0; h < R / 2; ++h) {
                ax2[h] = f32x2{ax[2 * h], ax[2 * h + 1]}; ay2[h] = f32x2{ay[2 * h], ay[2 * h + 1]}; az2[h] = f32x2{az[2 * h], az[2 * h + 1]};
            }
        }

        for (int win = 0; win < it.nwin; ++win) {
        if (win > 0) {
            b0 = win == 1 ? it.b0[1] : it.b0[2];
            nb_total = win == 1 ? it.nb[1] : it.nb[2];
        }
        for (int64_t st0 = 0; st0 < nb_total; st0 += CAP, ++stage_no) {
            const int n = (int)(nb_total - st0 < CAP ? nb_total - st0 : CAP);
            if (st0 > 0 || win > 0) {
                __syncthreads();
                stage_in(st0, n);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stage has landed (see k_count_band32)
            __syncthreads();
            if (lane == 0) {
                stage[n] = PAD_COORD32; stage[COLB / 4 + n] = PAD_COORD32; stage[2 * (COLB / 4) + n] = PAD_COORD32;
                if (WEIGHTED) sw[n] = 0.0;
            }
            if (WEIGHTED && !cs.w)
                for (int e = lane; e < n; e += 64) sw[e] = 1.0;
            __syncthreads();
            const unsigned a_key = a_stage + COLB * (unsigned)cl.axis;
            const unsigned a_sent = a_key + ((unsigned)n << 2);
            unsigned ql = a_key - 4u, qh = ql;
            for (unsigned step = 4u << (31 - __builtin_clz(n)); step >= 4u; step >>= 1) {
                const unsigned pl = ql + step, ph = qh + step;
                const float kl = *(const __attribute__((address_space(3))) float *)(size_t)(pl < a_sent ? pl : a_sent);
                const float kh = *(const __attribute__((address_space(3))) float *)(size_t)(ph < a_sent ? ph : a_sent);
                ql = kl < klo ? pl : ql;
                qh = kh <= khi ? ph : qh;
            }
            int lo = (int)((ql + 4u - a_key) >> 2), hi = (int)((qh + 4u - a_key) >> 2);
            if (n_own == 0) lo = hi = n;
            const int len = hi - lo;
            nev += (unsigned int)(len * n_own);
#if defined(YAW_FINE_DIAG) && YAW_FINE_DIAG == 1
            const int steps = 0;  // diagnostics: everything but the walk (wrong counts)
#else
            const int steps = wave_max_nonneg(len);
#endif

            unsigned cur = a_stage + ((unsigned)lo << 2);
            const unsigned last = a_stage + ((unsigned)n << 2);
            // The walk is a chain of LDS round trips (entry -> nearest edge -> its bounds -> histogram): the next entry is
            // fetched while this one is classified, and the edge bounds of all lane objects are fetched together.
            struct Entry { float x, y, z; double w; unsigned at; };
            auto fetch = [&](unsigned at) {
                Entry e;
                e.at = at < last ? at : last;
                e.x = *(const __attribute__((address_space(3))) float *)(size_t)e.at;
                e.y = *(const __attribute__((address_space(3))) float *)(size_t)(e.at + COLB);
                e.z = *(const __attribute__((address_space(3))) float *)(size_t)(e.at + 2 * COLB);
                e.w = WEIGHTED ? lds_f64(((e.at - a_stage) << 1) + a_sw) : 1.0;
                return e;
            };
            unsigned pend[R];  // cells of the previous entry's hits (unweighted)
#pragma unroll
            for (int r = 0; r < R; ++r) pend[r] = a_dummy;
            Entry nxt = fetch(cur);
            // (consumed here, so that the compiler's wait-count bookkeeping enters the loop with nothing of the prologue pending:
            // a wait at the loop header has to satisfy every incoming edge, and "the reads just issued" of the prologue would
            // turn it into a full drain -- of the previous trip's histogram updates, every trip)
            if constexpr (WEIGHTED) asm volatile("" :: "v"(nxt.x), "v"(nxt.y), "v"(nxt.z), "v"(nxt.w));
            else asm volatile("" :: "v"(nxt.x), "v"(nxt.y), "v"(nxt.z));
            for (int s = 0; s < steps; ++s) {
                const Entry en = nxt;
                cur += 4;
                nxt = fetch(cur);  // (past the longest band: an entry beyond every band, or the sentinel -- never used)
                const unsigned a16 = en.at;
                const float ex = en.x, ey = en.y, ez = en.z;
                const double ew = en.w;
                float s32[R];
                if constexpr (R >= 2) {
#pragma unroll
                    for (int h = 0; h < R / 2; ++h) {
                        const f32x2 dx = ax2[h] - ex, dy = ay2[h] - ey, dz = az2[h] - ez;
                        const f32x2 sq = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                        s32[2 * h] = sq.x; s32[2 * h + 1] = sq.y;
                    }
                } else {
                    const float dx = ax[0] - ex, dy = ay[0] - ey, dz = az[0] - ez;
                    s32[0] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                }
                int jj[R];
                bool unc[R];
                unsigned long long any_unc_mask = 0ull;
                f32x2 tb[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    // nearest edge of the object's grid (s32 = 0: f = -inf -> edge 0) and its float32 bounds
                    const float fr = __builtin_rintf(__builtin_fmaf(__builtin_amdgcn_logf(s32[r]), pm[r], -pa[r]));
                    jj[r] = (int)__builtin_amdgcn_fmed3f(fr, 0.0f, nf_f);
                    tb[r] = *(const __attribute__((address_space(3))) f32x2 *)(size_t)(a_edges[r] + ((unsigned)jj[r] << 3));
                }
                if constexpr (!WEIGHTED) {
                    // the histogram updates of the PREVIOUS entry go out behind this entry's reads: by the time the loop comes
                    // round to anything that waits for the LDS, they have long been absorbed
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        (void)__hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned int *)(size_t)pend[r], 1u,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int j = jj[r];
                    // the side of the edge s32 lies on; "inside the guard band of edge j" = neither below nor above (spelled
                    // !below && !(s32 > hi) the compiler issues a third compare for the negation)
                    const bool below = s32[r] < tb[r].x, above = s32[r] > tb[r].y;
                    const int bin = j - (below ? 1 : 0);
                    unc[r] = !(below | above);
                    any_unc_mask |= ~(__builtin_amdgcn_ballot_w64(below) | __builtin_amdgcn_ballot_w64(above));  // scalar unit only
                    const bool hit = ((unsigned)bin < (unsigned)nf) & !unc[r];
                    const unsigned cell = a_rowh[r] + ((unsigned)bin << HB);
                    if constexpr (WEIGHTED) {
                        if (hit)
                            (void)__hip_atomic_fetch_add((__attribute__((address_space(3))) double *)(size_t)cell, aw[r] * ew,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        pend[r] = hit ? cell : a_dummy;  // a miss adds to the lane's dummy cell
                    }
                }
#if defined(YAW_FINE_DIAG) && YAW_FINE_DIAG == 6
                any_unc_mask = 0ull;  // diagnostics: no exact re-evaluation (wrong counts)
#endif
                if (any_unc_mask != 0ull) {
                    // the exact float64 predicate on the float64 columns decides, against the host's float64 thresholds (rare)
                    const unsigned eidx = (a16 - a_stage) >> 2;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        if (unc[r] && eidx < (unsigned)n && r < n_own) {
                            const int j = jj[r];
                            // (the admission rule of build_fine32 leaves edge j as the only one s can be confused with)
                            const ExactEval<1> ev = band32_exact<1>(cl.x, cl.y, cl.z, it.a0 + lane * R + r, cs.x, cs.y, cs.z,
                                                                    cs.idx ? (int64_t)cs.idx[b0 + st0 + eidx] : b0 + st0 + (int64_t)eidx,
                                                                    t + (size_t)(MERGED ? kb[r] : kfix) * n_edges + j,
                                                                    counters + 9 + 8 * (ticket & (EVAL_SLOTS - 1)));
                            const int bin = ev.s <= ev.th[0] ? j - 1 : j;  // t[bin] < s <= t[bin + 1]
                            if ((unsigned)bin < (unsigned)nf) {
                                const unsigned cell = a_rowh[r] + ((unsigned)bin << HB);
                                if constexpr (WEIGHTED)
                                    (void)__hip_atomic_fetch_add((__attribute__((address_space(3))) double *)(size_t)cell, aw[r] * ew,
                                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                else
                                    (void)__hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned int *)(size_t)cell, 1u,
                                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                }
            }
            if constexpr (!WEIGHTED) {  // the last entry's updates
#pragma unroll
                for (int r = 0; r < R; ++r)
                    (void)__hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned int *)(size_t)pend[r], 1u,
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (!WEIGHTED && (stage_no & flush_mask) == flush_mask) flush_counts();
        }
        }
        if constexpr (WEIGHTED) {
            __syncthreads();
            for (int idx = lane; idx < nslots; idx += 64) {
                partials[(int64_t)it.pot * nslots + idx] = (double)hist[idx];
                hist[idx] = HistT(0);
            }
        } else {
            flush_counts();
        }
        nev = wave_sum_lane63(nev);  // (DPP: the shuffle form is six trips through the LDS crossbar, per item, for a statistic)
        if (lane == 63 && nev) atomicAdd(&counters[8 + 8 * (ticket & (EVAL_SLOTS - 1))], (unsigned long long)nev);
    }
}

// Evaluated pairs per job (na * nb of the job's kept items): the cost the host balances over GPUs.
__global__ void k_item_work(const Item *__restrict__ items, const unsigned long long *__restrict__ counters,
                            int slots_per_job, unsigned long long *__restrict__ job_work) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= counters[0]) return;
    const Item it = items[i];
    unsigned long long streamed = 0;
    for (int w = 0; w < it.nwin; ++w) streamed += (unsigned long long)it.nb[w];
    atomicAdd(&job_work[item_slot(it) / slots_per_job], (unsigned long long)it.na * streamed);
}

// Weighted sums: every kept item left a slab of `slab` float64 values at partials[pot]. They are added per output
// slot in a fixed two-level order -- chunks of REDUCE_CHUNK consecutive potential items, then the chunks of a slot in
// order -- so the result is bit-reproducible and the reduction is parallel over (chunk, value). Dropped potential
// items (kept[pot] == 0, their slab is never written) are skipped; kept == nullptr means every item was kept.
constexpr int REDUCE_CHUNK = 32;
__global__ void k_reduce_chunks(const double *__restrict__ partials, const unsigned char *__restrict__ kept,
                                const int64_t *__restrict__ prefix, const int64_t *__restrict__ cprefix, int n_slots,
                                int slab, double *__restrict__ chunk_sums) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t g = idx / slab;
    if (g >= cprefix[n_slots]) return;
    const int e = (int)(idx - g * slab);
    int lo = 0, hi = n_slots;  // slot = largest s with cprefix[s] <= g
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cprefix[mid] <= g) lo = mid; else hi = mid;
    }
    const int64_t p0 = prefix[lo] + (g - cprefix[lo]) * REDUCE_CHUNK;
    const int64_t p1 = p0 + REDUCE_CHUNK < prefix[lo + 1] ? p0 + REDUCE_CHUNK : prefix[lo + 1];
    double acc = 0.0;
    for (int64_t pot = p0; pot < p1; ++pot)
        if (!kept || kept[pot]) acc += partials[pot * slab + e];
    chunk_sums[idx] = acc;
}

__global__ void k_reduce_slots(const double *__restrict__ chunk_sums, const int64_t *__restrict__ cprefix, int n_slots,
                               int slab, double *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n_slots * slab) return;
    const int slot = (int)(idx / slab), e = (int)(idx - (int64_t)slot * slab);
    double acc = 0.0;
    for (int64_t g = cprefix[slot]; g < cprefix[slot + 1]; ++g) acc += chunk_sums[g * slab + e];
    out[idx] = acc;
}

// ndarray.sum() of values v(0) .. v(n - 1), in numpy's order (see numpy_sum on the host side of yawhip_count_pairs_dense)
template <typename F>
__device__ double numpy_sum_dev(F v, int lo, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += v(lo + i);
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = v(lo + j);
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += v(lo + i + j);
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += v(lo + i);
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return numpy_sum_dev(v, lo, n2) + numpy_sum_dev(v, lo + n2, n - n2);
}

// Per-scale recombination of the fine bins on the device (yawhip_count_pairs_dense): out[job][bin][scale] = sum over the
// scale's fine bins of count (or weighted sum) x separation weight -- the same products and the same order of additions as
// the host epilogue; what crosses PCIe afterwards is S values per (job, bin) instead of E - 1.
__global__ void k_combine_scales(const unsigned long long *__restrict__ counts, const double *__restrict__ sums, int weighted,
                                 int64_t n_jobs, int n_bins, int nf, int n_scales, const int32_t *__restrict__ slices,
                                 const double *__restrict__ factors, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_jobs * n_bins * n_scales) return;
    const int sc = (int)(i % n_scales), k = (int)((i / n_scales) % n_bins);
    const int64_t j = i / ((int64_t)n_scales * n_bins);
    const int lo = slices[2 * (k * n_scales + sc)], hi = slices[2 * (k * n_scales + sc) + 1];
    const int64_t base = (j * n_bins + k) * (int64_t)nf;
    const double *wk = factors ? factors + (int64_t)k * nf : nullptr;
    auto value = [&](int e) {
        const double v = weighted ? sums[base + e] : (double)counts[base + e];
        return wk ? v * wk[e] : v;
    };
    out[i] = hi > lo ? numpy_sum_dev(value, lo, hi - lo) : 0.0;
}

// rows of a call's result into their place in the full [rows][row] tensor (device-resident all-reduce of the process route)
__global__ void k_scatter_rows(const double *__restrict__ in, const int32_t *__restrict__ row_index, int64_t row, int64_t n,
                               double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = i / row;
    out[(int64_t)row_index[r] * row + (i - r * row)] = in[i];
}

__global__ void k_counts_to_double(const unsigned long long *__restrict__ in, double *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}

inline DevTab make_tab(const double *x, const double *y, const double *z, const double *w, const int32_t *k, const int64_t *off,
                       const int64_t *vbase, const int64_t *slo, const int64_t *tiles, const TileRec *tile_rec, const RunGrid *grid,
                       int axis, const float *q = nullptr, int64_t q_stride = 0, const int32_t *idx = nullptr) {
    return DevTab{(gf64p)x, (gf64p)y, (gf64p)z, (gf64p)w, (gi32p)k, (gi64p)off, (gi64p)vbase, (gi64p)slo, (gi64p)tiles,
                  tile_rec, grid, (gf32p)q, (gf32p)(q ? q + q_stride : nullptr), (gf32p)(q ? q + 2 * q_stride : nullptr), (gi32p)idx,
                  (gi32p)nullptr, axis, 0};
}

template <typename T>
struct DevBuf {  // grow-only device workspace
    T *ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 64;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&ptr), want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

// Small per-call tables travel in ONE host-to-device copy from a pinned staging buffer, and the results (counters, counts,
// sums) come back in ONE copy into pinned memory: a dozen pageable copies of a few hundred bytes each cost more host
// time than the kernels of a small call take.
struct Arena {
    unsigned char *h = nullptr, *d = nullptr;  // pinned host image and device buffer of the same size
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        release();
        const size_t want = n + n / 4 + 4096;
        hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&h), want, hipHostMallocPortable);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d), want);
        if (e == hipSuccess) cap = want; else release();
        return e;
    }
    void release() {
        if (h) (void)hipHostFree(h);
        if (d) (void)hipFree(d);
        h = d = nullptr;
        cap = 0;
    }
};
inline size_t align16(size_t n) { return (n + 15) & ~(size_t)15; }

}  // namespace

template <typename T>
struct View {  // typed window into one of the context's arenas, set by every yawhip_count_pairs call
    T *ptr = nullptr;
};

// Everything ONE count call in flight owns: the tables it sent, its work items, partial sums, result block and timing
// events. A context keeps MAX_BATCH of these; the active one is the base-class part of the context (all the code below
// says ctx->d_items ...), the others are parked -- yawhip_count_pairs_dense_batch activates one per request so that
// several counts of a measurement are on the stream at once (use_slot).
struct CallBufs {
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evc0 = nullptr, evc1 = nullptr, ev_done = nullptr;
    View<int32_t> d_jobs;
    View<int64_t> d_prefix;
    View<double> d_t;
    View<float> d_dthr;
    View<float> d_thr32;
    View<double> d_rwin;
    DevBuf<Item> d_items;
    View<unsigned long long> d_ctr;
    View<unsigned long long> d_counts;
    View<double> d_sums;
    DevBuf<double> d_partials;
    DevBuf<double> d_chunk_sums;
    View<int64_t> d_cprefix;        // weighted calls: first chunk of every output slot (in the plan's device tables)
    DevBuf<unsigned char> d_kept;   // weighted runs: 1 for potential items the builder kept
    Arena out;   // results (device -> host)
    Arena comb;  // yawhip_count_pairs_dense: recombination tables in, per-scale values out
    View<DevTab> d_tabs;
    hipError_t make_events() {
        hipError_t e = hipSuccess;
        for (hipEvent_t *ev : {&ev0, &ev1, &evc0, &evc1, &ev_done})
            if (e == hipSuccess && !*ev) e = hipEventCreate(ev);
        return e;
    }
    void release_all() {
        d_items.release(); d_partials.release(); d_chunk_sums.release(); d_kept.release();
        out.release(); comb.release();
        for (hipEvent_t *ev : {&ev0, &ev1, &evc0, &evc1, &ev_done}) {
            if (*ev) (void)hipEventDestroy(*ev);
            *ev = nullptr;
        }
    }
};
constexpr int MAX_BATCH = 4;
constexpr size_t MAX_PLANS = 16;  // plans kept per context (least recently used one goes)  // counts of one measurement on the stream at once (DD, DR, RD, RR)

namespace {
struct HostPlan;  // what a call derives from its inputs on the host, kept for the next call with the same inputs (below)
}

struct yawhip_ctx : CallBufs {
    int device = 0;
    hipStream_t stream = nullptr;
    int tile_r = 0;          // 0 = auto
    int hist_copies_log2 = -1;  // band kernel: log2 of the copies of the LDS histogram (-1 = auto)
    int band_batch_log2 = -1;   // band kernel: log2 of the consecutive items a workgroup takes per visit (-1 = auto)
    int band_cap = 0;        // entries per LDS stage of the band kernel: 0 = auto, BCAP (192), BCAP_MID (288)
    int seg_strips = 1;      // binned x binned counts of dense catalogues use the per-segment strip layouts
    int seg_min_run = SEG_STRIPS_MIN_RUN;  // mean run length of the lane side from which binned x binned counts use it
    int debug_no_hits = 0;   // diagnostics only: pre-filter threshold above 1 -> no pair survives (timing of the fast path)
    int auto_orient = 1;     // every job runs on the strip layouts of the orientation that suits its patches (0: the catalogues' sort axis)
    int64_t slab_budget = 1ll << 30;  // bytes of per-item partial sums (weighted calls) above which a job list is cut in two
    int band_grid_div = 0;   // band kernel on strip items: workgroups = potential items / this (the kernel loops over the rest);
                             // 0 = auto (make_plan: 8, 16 for per-bin items, 4 on clustered catalogues)
    int flush_log2 = 17;     // band kernel: stages between flushes of the 32-bit LDS counters = 2^flush_log2
    int spin_wait = 1;       // wait for a call's results by polling the stream for the first 2 ms, then block (0: block at once)
    int item_segments = 1;   // strip builder -> float32 band kernels: the item list in eight segments, one per XCD (append_items)
    int half_bands = 1;      // self counts on merged triple runs, one object per lane: diagonal jobs take every unordered pair once (x 2)
    int triple_runs = 1;     // float32 band kernels stream merged triple runs (k_merge_triples) when the partner strips are c - 1, c, c + 1
    int band_fp32 = 1;       // band kernel on strip layouts of unit vectors: float32 classification + exact float64 for the
                             // guard bands (k_count_band32); 0: every entry in float64 (k_count_band)
    double strip_width = 0.005;  // strip grid of newly uploaded catalogues (chord units, ~17 arcmin); 0 = no strips
    int default_kernel = YAWHIP_KERNEL_AUTO;
    int lds_limit = 160 * 1024;
    int n_cu = 256;
    DevBuf<unsigned long long> d_jobwork;
    DevBuf<double> d_full;          // yawhip_count_pairs_rows_device: the full result tensor of a sharded count
    DevBuf<int32_t> d_rowidx;
    // A context made by yawhip_ctx_create_multi owns one further context per additional device: catalogues are
    // replicated on all of them and yawhip_count_pairs splits its job list over them.
    std::vector<yawhip_ctx *> peers;
    struct Plan {  // job partition of the last multi-device call (a function of its inputs only)
        uint64_t key = 0;
        std::vector<std::vector<int32_t>> parts;  // job indices per device
    } plan;
    yawsort::Workspace sort_ws;  // upload-side sorts
    CallBufs parked[MAX_BATCH];  // the slots that are not active (the active one's entry is empty)
    int slot = 0;
    uint64_t opt_gen = 1;        // bumped by every yawhip_ctx_set_option: plans are keyed on it
    uint64_t plan_clock = 0;     // least-recently-used stamp of the plans
    std::vector<HostPlan *> plans;
};
// Make slot i the active set of per-call buffers (its events are created on first use).
inline hipError_t use_slot(yawhip_ctx *ctx, int i) {
    if (i != ctx->slot) {
        std::swap(static_cast<CallBufs &>(*ctx), ctx->parked[ctx->slot]);  // park the active one
        std::swap(static_cast<CallBufs &>(*ctx), ctx->parked[i]);          // activate slot i
        ctx->slot = i;
    }
    return ctx->make_events();
}


struct StripLayout {
    bool built = false;
    double *x = nullptr, *y = nullptr, *z = nullptr, *w = nullptr;
    int32_t *k = nullptr;             // bin id per object (patch-level layout of a binned catalogue)
    float *q = nullptr;               // [3][q_stride] float32 images of x, y, z (k_count_band32)
    int64_t q_stride = 0;
    int64_t *off = nullptr;           // [V+1] offsets of the runs
    std::vector<int64_t> h_off;       // same on the host
    std::vector<int64_t> h_vbase;     // [G+1] first run of every group
    std::vector<int64_t> h_slo;       // [G]   global strip index of a group's first run
    std::vector<int64_t> h_tiles[3];  // [V+1] prefix of lane tiles over the runs, for tiles of MWG * {1, 2, 4} objects
    int64_t *d_vbase = nullptr, *d_slo = nullptr, *d_tiles[3] = {nullptr, nullptr, nullptr};
    TileRec *d_tile_rec[3] = {nullptr, nullptr, nullptr};  // [tiles] first object, length and run of every lane tile
    RunGrid *d_grid = nullptr;        // [V+1] per-run index along the sort axis (item builder)
    // merged triple runs (k_merge_triples), built when a float32 band kernel first streams this layout
    bool triples = false;
    float *q3 = nullptr;              // [3][q3_stride] float32 images in merged order (3 n entries)
    int64_t q3_stride = 0;
    double *w3 = nullptr;             // weights in merged order
    int32_t *idx3 = nullptr;          // [3 n] entry -> index in the layout's own order
    int32_t *pos3 = nullptr;          // [n] object -> its place in the triple run centred on its own strip
    int64_t *off3 = nullptr;          // [V + 2 G + 1] offsets of the triple runs: group g has its strips + 2, first one = vbase[g] + 2 g
    RunGrid *d_grid3 = nullptr;       // [V + 2 G + 1]
    int64_t n_groups = 0;
    int64_t device_bytes = 0;
    double obj_run = 0.0;             // run length seen by the typical object (sum len^2 / sum len)
    double same_bin = 0.0;            // fraction of neighbours in the layout's order that share their bin (binned patch-level layouts)
    void release() {
        for (void *ptr : {(void *)x, (void *)y, (void *)z, (void *)w, (void *)k, (void *)q, (void *)off, (void *)d_vbase, (void *)d_slo,
                          (void *)d_tiles[0], (void *)d_tiles[1], (void *)d_tiles[2], (void *)d_tile_rec[0], (void *)d_tile_rec[1],
                          (void *)d_tile_rec[2], (void *)d_grid, (void *)q3, (void *)w3, (void *)idx3, (void *)pos3, (void *)off3, (void *)d_grid3})
            if (ptr) (void)hipFree(ptr);
        q3 = nullptr; w3 = nullptr; idx3 = nullptr; pos3 = nullptr; off3 = nullptr; d_grid3 = nullptr; triples = false;
        x = y = z = w = nullptr; k = nullptr; q = nullptr; off = d_vbase = d_slo = nullptr;
        d_tiles[0] = d_tiles[1] = d_tiles[2] = nullptr;
        d_tile_rec[0] = d_tile_rec[1] = d_tile_rec[2] = nullptr;
        d_grid = nullptr;
        built = false;
    }
};

struct yawhip_catalog {
    yawhip_ctx *ctx = nullptr;
    uint64_t uid = 0;  // upload id, never reused (plans are keyed on it, not on the address)
    int64_t n = 0;
    int32_t n_patches = 0, nb = 1;
    double *x = nullptr, *y = nullptr, *z = nullptr, *w = nullptr;
    int64_t *off = nullptr;
    std::vector<int64_t> h_off;
    int64_t device_bytes = 0;
    bool unit_norm = true;  // every |a|^2 within UNIT_NORM_TOL of 1 (precondition of the FP32 pre-filter)
    int axis = 2;           // coordinate the segments are sorted by (0 = x, 1 = y, 2 = z)
    // strip layouts: the objects of every *group* cut into strips of a global grid along a second axis; inside a
    // (group, strip) run sorted along the sort axis. Partner runs of two catalogues are those whose grid indices
    // differ by at most sqrt(t_max) / spacing + 1.
    //   strips: group = patch, all redshift bins together, bin id per object (cross-correlation counts);
    //   seg:    group = (patch, bin) segment (binned x binned counts of dense catalogues; binned catalogues only).
    // One layout per orientation o = sort axis u (strips along (o + 2) % 3), built when a job first needs it (the one of
    // the catalogue's own sort axis at upload): see DevTab.
    StripLayout strips[3], seg[3];
    std::vector<yawhip_catalog *> replicas;  // copies on ctx->peers (multi-device contexts), same order
    bool has_strips = false;          // strip layouts can be built (unit vectors, n > 0)
    double strip_width = 0.0;         // grid spacing (chord units); 0 = one run per patch
    std::vector<double> h_box;        // [P][6] bounding box of every patch: min x, y, z, max x, y, z (empty patch: +4 / -4)
};

namespace {

void drop_plans(yawhip_ctx *ctx, const yawhip_catalog *c);  // (defined with HostPlan)

const double *key_of(const double *x, const double *y, const double *z, int axis) { return axis == 0 ? x : (axis == 1 ? y : z); }
CatView view_of(const yawhip_catalog *c) {
    return CatView{c->x, c->y, c->z, c->w, c->off, c->nb, key_of(c->x, c->y, c->z, c->axis), c->axis};
}

template <int R, bool W, bool P, bool F>
hipError_t launch_count(yawhip_ctx *ctx, const yawhip_catalog *c1, const yawhip_catalog *c2, int n_slots, int n_bins,
                        int n_edges, int64_t n_items, size_t lds_bytes) {
    auto kern = k_count<R, W, P, F>;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    const int64_t max_grid = (1ll << 31) / WG;  // a launch addresses at most 2^32 - 1 work-items per dimension
    for (int64_t base = 0; base < n_items; base += max_grid) {
        const int64_t g = std::min(max_grid, n_items - base);
        hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(WG), lds_bytes, ctx->stream, view_of(c1), view_of(c2),
                           ctx->d_items.ptr, n_bins, n_edges, ctx->d_t.ptr, ctx->d_dthr.ptr, base, ctx->d_counts.ptr,
                           ctx->d_partials.ptr, ctx->d_ctr.ptr);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <bool W, bool P, bool F>
hipError_t launch_count_r(int r, yawhip_ctx *ctx, const yawhip_catalog *c1, const yawhip_catalog *c2, int n_slots,
                          int n_bins, int n_edges, int64_t n_items, size_t lds) {
    switch (r) {
        case 1: return launch_count<1, W, P, F>(ctx, c1, c2, n_slots, n_bins, n_edges, n_items, lds);
        case 2: return launch_count<2, W, P, F>(ctx, c1, c2, n_slots, n_bins, n_edges, n_items, lds);
        default: return launch_count<4, W, P, F>(ctx, c1, c2, n_slots, n_bins, n_edges, n_items, lds);
    }
}

template <bool W>
hipError_t launch_count_any(bool priv, bool filter, int r, yawhip_ctx *ctx, const yawhip_catalog *c1,
                            const yawhip_catalog *c2, int n_slots, int n_bins, int n_edges, int64_t n_items, size_t lds) {
    if (priv)
        return filter ? launch_count_r<W, true, true>(r, ctx, c1, c2, n_slots, n_bins, n_edges, n_items, lds)
                      : launch_count_r<W, true, false>(r, ctx, c1, c2, n_slots, n_bins, n_edges, n_items, lds);
    return filter ? launch_count_r<W, false, true>(r, ctx, c1, c2, n_slots, n_bins, n_edges, n_items, lds)
                  : launch_count_r<W, false, false>(r, ctx, c1, c2, n_slots, n_bins, n_edges, n_items, lds);
}

// Nearest patch centre of every object (replaces scipy.cluster.vq.vq in assign_patch_centers, catalog.py:229-249):
// squared distance accumulated x, y, z in that order with separately rounded products and sums, first minimum
// wins -- the arithmetic of scipy's small-dimension vq loop, so ids are identical including exact ties.
__global__ __launch_bounds__(256) void k_assign_patches(int64_t n, const double *__restrict__ x, const double *__restrict__ y,
                                                       const double *__restrict__ z, int n_centers,
                                                       const double *__restrict__ centers, int32_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double *c = reinterpret_cast<double *>(lds_raw);  // [n_centers][3]
    for (int e = threadIdx.x; e < 3 * n_centers; e += blockDim.x) c[e] = centers[e];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double px = x[i], py = y[i], pz = z[i];
    double best = INFINITY;
    int best_j = -1;
    for (int j = 0; j < n_centers; ++j) {
        const double dx = px - c[3 * j], dy = py - c[3 * j + 1], dz = pz - c[3 * j + 2];
        const double xx = dx * dx;
        const double yy = dy * dy;
        const double zz = dz * dz;
        const double sxy = xx + yy;
        const double d = sxy + zz;
        if (d < best) {
            best = d;
            best_j = j;
        }
    }
    out[i] = best_j;
}

// ---- upload-side kernels: ordering of a catalogue on the device (the sorts themselves: yawhip_sort.hip) ----
__global__ void k_gather_columns(int64_t n, const uint32_t *__restrict__ perm, const double *__restrict__ sx,
                                 const double *__restrict__ sy, const double *__restrict__ sz, const double *__restrict__ sw,
                                 double *__restrict__ dx, double *__restrict__ dy, double *__restrict__ dz, double *__restrict__ dw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t src = perm[i];
    dx[i] = sx[src];
    dy[i] = sy[src];
    dz[i] = sz[src];
    if (sw) dw[i] = sw[src];
}

// largest s in [0, n_seg) with off[s] <= i (off[0] = 0 <= i < off[n_seg])
__device__ __forceinline__ int segment_of(const int64_t *__restrict__ off, int n_seg, int64_t i) {
    int lo = 0, hi = n_seg;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// bin id of every object of the strip layout = its (patch, bin) segment in the input order, modulo B
__global__ void k_gather_bins(int64_t n, const uint32_t *__restrict__ perm, const int64_t *__restrict__ off, int64_t n_seg,
                              int n_bins, int32_t *__restrict__ bins) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bins[i] = segment_of(off, (int)n_seg, (int64_t)perm[i]) % n_bins;
}

// float32 images of a strip layout's columns, rounded to nearest: [3][stride]
__global__ void k_make_q(int64_t n, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                         int64_t stride, float *__restrict__ q) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    q[i] = (float)x[i];
    q[stride + i] = (float)y[i];
    q[2 * stride + i] = (float)z[i];
}

// How often two neighbours of the (strip, u)-sorted order share their redshift bin: ~1/B when redshift and position are
// unrelated, towards 1 when they are not -- then the lanes of a wave keep hitting the same histogram cells.
__global__ void k_same_bin_neighbours(int64_t n, const int32_t *__restrict__ bins, unsigned long long *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool same = i + 1 < n && bins[i] == bins[i + 1];
    const unsigned long long m = __builtin_amdgcn_ballot_w64(same);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, (unsigned long long)__popcll(m));
}

// grid index floor((v + 1) / width) of every object (0 without strips) and the occupied range per patch
__global__ void k_strip_index(int64_t n, const double *__restrict__ v, double width, const int64_t *__restrict__ poff,
                              int n_patches, int32_t *__restrict__ gidx, int32_t *__restrict__ lohi) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = i < n;
    int32_t g = 0;
    int p = -1;
    if (ok) {
        g = width > 0.0 ? (int32_t)floor((v[i] + 1.0) / width) : 0;
        gidx[i] = g;
        p = segment_of(poff, n_patches, i);
    }
    // Objects of a patch are contiguous, so nearly every wave sits inside one patch: reduce there and issue one
    // atomic pair per wave (one pair per object on 2P addresses cost 97 ms for 10 M objects).
    const int p0 = __builtin_amdgcn_readfirstlane(p);
    if (__builtin_amdgcn_ballot_w64(p != p0) == 0ull) {
        if (p0 < 0) return;  // whole wave past the end
        int32_t lo = g, hi = g;
        for (int off = 32; off > 0; off >>= 1) {
            lo = min(lo, __shfl_xor(lo, off, 64));
            hi = max(hi, __shfl_xor(hi, off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&lohi[2 * p0], lo);
            atomicMax(&lohi[2 * p0 + 1], hi);
        }
    } else if (ok) {  // a wave across a patch boundary (or the ragged end)
        atomicMin(&lohi[2 * p], g);
        atomicMax(&lohi[2 * p + 1], g);
    }
}

// run id of the object that is i-th in `order` (objects of a patch are contiguous in the input)
__global__ void k_run_of(int64_t n, const uint32_t *__restrict__ order, const int32_t *__restrict__ gidx,
                         const int64_t *__restrict__ poff, int n_patches, const int64_t *__restrict__ vbase,
                         const int64_t *__restrict__ slo, uint32_t *__restrict__ run) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t src = order[i];
    const int p = segment_of(poff, n_patches, (int64_t)src);
    run[i] = (uint32_t)(vbase[p] + (int64_t)gidx[src] - slo[p]);
}

// moff[r] = first position of the (sorted) run column that holds a run >= r; moff[n_runs] = n
__global__ void k_run_offsets(const uint32_t *__restrict__ run_sorted, int64_t n, int64_t n_runs, int64_t *__restrict__ moff) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_runs) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)run_sorted[mid] < r) lo = mid + 1; else hi = mid;
    }
    moff[r] = lo;
}

// monotone map double -> uint64 (atomicMin / atomicMax on the images order like the doubles)
__host__ __device__ inline unsigned long long sortable_of(double d) {
    unsigned long long b;
    memcpy(&b, &d, sizeof b);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
inline double double_of(unsigned long long s) {
    const unsigned long long b = (s >> 63) ? (s & 0x7fffffffffffffffull) : ~s;
    double d;
    memcpy(&d, &b, sizeof d);
    return d;
}

// Bounding box of every patch ([P][6]: min x, y, z, max x, y, z as sortable images) and, in box[6 P], the number of
// waves that saw an object off the unit sphere. One atomic set per wave inside a patch (see k_strip_index).
__global__ void k_patch_boxes(int64_t n, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                              const int64_t *__restrict__ poff, int n_patches, unsigned long long *__restrict__ box) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = i < n;
    double v[3] = {0.0, 0.0, 0.0};
    int p = -1;
    bool off_sphere = false;
    if (ok) {
        v[0] = x[i]; v[1] = y[i]; v[2] = z[i];
        p = segment_of(poff, n_patches, i);
        const double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
        off_sphere = !(n2 > 1.0 - UNIT_NORM_TOL && n2 < 1.0 + UNIT_NORM_TOL);
    }
    if (__builtin_amdgcn_ballot_w64(off_sphere) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(&box[(size_t)6 * n_patches], 1ull);
    const int p0 = __builtin_amdgcn_readfirstlane(p);
    if (__builtin_amdgcn_ballot_w64(p != p0) == 0ull) {
        if (p0 < 0) return;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            unsigned long long lo = sortable_of(v[a]), hi = lo;
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long l2 = __shfl_xor(lo, off, 64), h2 = __shfl_xor(hi, off, 64);
                lo = l2 < lo ? l2 : lo;
                hi = h2 > hi ? h2 : hi;
            }
            if ((threadIdx.x & 63) == 0) {
                atomicMin(&box[(size_t)6 * p0 + a], lo);
                atomicMax(&box[(size_t)6 * p0 + 3 + a], hi);
            }
        }
    } else if (ok) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicMin(&box[(size_t)6 * p + a], sortable_of(v[a]));
            atomicMax(&box[(size_t)6 * p + 3 + a], sortable_of(v[a]));
        }
    }
}

inline int64_t seg_len(const yawhip_catalog *c, int patch, int k) {
    const int kk = c->nb == 1 ? 0 : k;
    const int64_t i = (int64_t)patch * c->nb + kk;
    return c->h_off[i + 1] - c->h_off[i];
}

// Build one strip layout of a catalogue from its resident (patch, bin, u) copy: orientation o = sort axis inside a run,
// strips of the global grid along (o + 2) % 3; seg = groups are the (patch, bin) segments instead of the patches.
int build_strip_layout(yawhip_ctx *ctx, yawhip_catalog *c, int o, bool seg) {
    StripLayout &L = seg ? c->seg[o] : c->strips[o];
    if (L.built) return YAWHIP_OK;
    const int64_t n = c->n, nseg = (int64_t)c->n_patches * c->nb;
    const int n_groups = seg ? (int)nseg : c->n_patches;
    const bool want_bins = !seg && c->nb > 1;
    const double width = c->strip_width;
    const int saxis = (o + 2) % 3;  // z -> y, y -> x, x -> z
    std::vector<int64_t> h_poff((size_t)n_groups + 1);
    for (int g = 0; g <= n_groups; ++g) h_poff[(size_t)g] = seg ? c->h_off[(size_t)g] : c->h_off[(size_t)g * c->nb];
    const size_t col = (size_t)std::max<int64_t>(n, 1) * sizeof(double) + 16;  // + 16: the band kernel's 16-byte loads may touch the bytes behind the last element
    uint32_t *perm = nullptr, *perm2 = nullptr, *run = nullptr, *run_sorted = nullptr;
    int32_t *gidx = nullptr, *lohi = nullptr;
    int64_t *poff = nullptr;
    auto bail = [&](hipError_t err, const char *what) {
        for (void *q : {(void *)perm, (void *)perm2, (void *)run, (void *)run_sorted, (void *)gidx, (void *)lohi, (void *)poff})
            if (q) (void)hipFree(q);
        if (err == hipSuccess) return (int)YAWHIP_OK;
        L.release();
        return fail(err == hipErrorOutOfMemory ? YAWHIP_ERR_OOM : YAWHIP_ERR_HIP, "strip layout (%s) failed: %s", what,
                    hipGetErrorString(err));
    };
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<int32_t> h_lohi((size_t)2 * n_groups);
    for (int g = 0; g < n_groups; ++g) { h_lohi[(size_t)2 * g] = INT32_MAX; h_lohi[(size_t)2 * g + 1] = INT32_MIN; }
    const size_t n1 = (size_t)std::max<int64_t>(n, 1);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&poff), (size_t)(n_groups + 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&lohi), (size_t)2 * n_groups * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&gidx), n1 * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&perm), n1 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&perm2), n1 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&run), n1 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&run_sorted), n1 * sizeof(uint32_t));
    if (e == hipSuccess)
        e = hipMemcpyAsync(poff, h_poff.data(), (size_t)(n_groups + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(lohi, h_lohi.data(), (size_t)2 * n_groups * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return bail(e, "strip tables");
    const unsigned ngrid = (unsigned)((n1 + 255) / 256);
    // grid index of every object, first / last occupied strip of every group
    hipLaunchKernelGGL(k_strip_index, dim3(ngrid), dim3(256), 0, ctx->stream, n, key_of(c->x, c->y, c->z, saxis), width, poff,
                       n_groups, gidx, lohi);
    e = hipMemcpyAsync(h_lohi.data(), lohi, (size_t)2 * n_groups * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return bail(e, "strip index");
    std::vector<int64_t> vbase((size_t)n_groups + 1, 0), slo((size_t)n_groups, 0);
    for (int g = 0; g < n_groups; ++g) {
        const bool any = h_poff[(size_t)g + 1] > h_poff[(size_t)g];
        slo[(size_t)g] = any ? h_lohi[(size_t)2 * g] : 0;
        vbase[(size_t)g + 1] = vbase[(size_t)g] + (any ? (int64_t)h_lohi[(size_t)2 * g + 1] - h_lohi[(size_t)2 * g] + 1 : 0);
    }
    const int64_t n_runs = vbase[(size_t)n_groups];
    if (n_runs >= (1ll << 31)) return bail(hipErrorInvalidValue, "too many strip runs");
    int run_bits = 1;
    while ((1ll << run_bits) < n_runs) ++run_bits;
    e = hipMalloc(reinterpret_cast<void **>(&L.d_vbase), (size_t)(n_groups + 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&L.d_slo), (size_t)std::max(n_groups, 1) * sizeof(int64_t));
    if (e == hipSuccess)
        e = hipMemcpyAsync(L.d_vbase, vbase.data(), (size_t)(n_groups + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(L.d_slo, slo.data(), (size_t)n_groups * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
    // order along the sort axis inside every group, then group by run (unique keys (run, rank): no reliance on
    // the stability of the sort)
    if (e == hipSuccess) e = yawsort::sort_segments(ctx->sort_ws, ctx->stream, n, key_of(c->x, c->y, c->z, o), poff, n_groups, perm);
    if (e != hipSuccess) return bail(e, "group sort");
    hipLaunchKernelGGL(k_run_of, dim3(ngrid), dim3(256), 0, ctx->stream, n, perm, gidx, poff, n_groups, L.d_vbase, L.d_slo, run);
    e = yawsort::sort_runs(ctx->sort_ws, ctx->stream, n, run, perm, run_bits, perm2, run_sorted);
    if (e != hipSuccess) return bail(e, "run sort");
    e = hipMalloc(reinterpret_cast<void **>(&L.x), col);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&L.y), col);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&L.z), col);
    if (e == hipSuccess && c->w) e = hipMalloc(reinterpret_cast<void **>(&L.w), col);
    if (e == hipSuccess && want_bins) e = hipMalloc(reinterpret_cast<void **>(&L.k), n1 * sizeof(int32_t) + 16);
    L.q_stride = (int64_t)((n1 + 3) & ~(size_t)3) + 8;  // a 16-byte load of the band kernel may run up to 12 bytes past a column
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&L.q), (size_t)3 * L.q_stride * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&L.off), (size_t)(n_runs + 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&L.d_grid), (size_t)(n_runs + 1) * sizeof(RunGrid));
    if (e == hipSuccess) e = hipMemsetAsync(L.d_grid, 0, (size_t)(n_runs + 1) * sizeof(RunGrid), ctx->stream);  // [V]: read for groups without runs
    if (e != hipSuccess) return bail(e, "strip layout");
    hipLaunchKernelGGL(k_gather_columns, dim3(ngrid), dim3(256), 0, ctx->stream, n, perm2, c->x, c->y, c->z, c->w, L.x, L.y, L.z, L.w);
    if (want_bins)
        hipLaunchKernelGGL(k_gather_bins, dim3(ngrid), dim3(256), 0, ctx->stream, n, perm2, c->off, nseg, c->nb, L.k);
    hipLaunchKernelGGL(k_make_q, dim3(ngrid), dim3(256), 0, ctx->stream, n, L.x, L.y, L.z, L.q_stride, L.q);
    hipLaunchKernelGGL(k_run_offsets, dim3((unsigned)((n_runs + 1 + 255) / 256)), dim3(256), 0, ctx->stream, run_sorted, n, n_runs,
                       L.off);
    if (n_runs > 0)
        hipLaunchKernelGGL(k_run_grid<double>, dim3((unsigned)((n_runs * (RUN_GRID + 1) + 255) / 256)), dim3(256), 0, ctx->stream, n_runs,
                           L.off, o == 0 ? L.x : (o == 1 ? L.y : L.z), L.d_grid);
    std::vector<int64_t> voff((size_t)n_runs + 1);
    e = hipMemcpyAsync(voff.data(), L.off, (size_t)(n_runs + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
    unsigned long long h_same = 0;
    if (want_bins && n > 1) {  // `run` (sorted away by now) serves as the 8-byte result cell
        unsigned long long *d_same = reinterpret_cast<unsigned long long *>(run);
        if (e == hipSuccess) e = hipMemsetAsync(d_same, 0, sizeof(unsigned long long), ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_same_bin_neighbours, dim3(ngrid), dim3(256), 0, ctx->stream, n, L.k, d_same);
            e = hipMemcpyAsync(&h_same, d_same, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return bail(e, "run offsets");
    L.same_bin = n > 1 ? (double)h_same / (double)(n - 1) : 0.0;
    for (int64_t r = 0; r < n_runs; ++r)
        if (voff[(size_t)r + 1] - voff[(size_t)r] >= (1ll << 32)) return bail(hipErrorInvalidValue, "a strip run of 2^32 objects or more");
    // small per-run tables the item builder walks on the device
    for (int ri = 0; ri < 3; ++ri) {
        const int64_t tile = (int64_t)MWG << ri;
        L.h_tiles[ri].assign((size_t)n_runs + 1, 0);
        for (int64_t r = 0; r < n_runs; ++r)
            L.h_tiles[ri][(size_t)r + 1] = L.h_tiles[ri][(size_t)r] + (voff[(size_t)r + 1] - voff[(size_t)r] + tile - 1) / tile;
        {  // record of every tile: the item builder decodes a potential item with one load instead of a search over the
           // prefix and a look-up of the run's offsets
            const int64_t n_tiles = L.h_tiles[ri][(size_t)n_runs];
            std::vector<TileRec> tile_rec((size_t)std::max<int64_t>(n_tiles, 1), TileRec{0, 0, 0});
            for (int64_t r = 0; r < n_runs; ++r)
                for (int64_t tl = L.h_tiles[ri][(size_t)r]; tl < L.h_tiles[ri][(size_t)r + 1]; ++tl) {
                    const int64_t a0 = voff[(size_t)r] + (tl - L.h_tiles[ri][(size_t)r]) * tile;
                    tile_rec[(size_t)tl] = TileRec{a0, (int32_t)std::min<int64_t>(tile, voff[(size_t)r + 1] - a0), (int32_t)r};
                }
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&L.d_tile_rec[ri]), tile_rec.size() * sizeof(TileRec));
            if (e == hipSuccess)
                e = hipMemcpy(L.d_tile_rec[ri], tile_rec.data(), tile_rec.size() * sizeof(TileRec), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&L.d_tiles[ri]), (size_t)(n_runs + 1) * sizeof(int64_t));
        if (e == hipSuccess)
            e = hipMemcpyAsync(L.d_tiles[ri], L.h_tiles[ri].data(), (size_t)(n_runs + 1) * sizeof(int64_t),
                               hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return bail(e, "tile tables");
    {  // run length as the typical OBJECT sees it (sum of squares / sum): equals the mean for uniform data, far above it
       // for clustered data, where most objects live in a few dense runs
        double sq = 0.0;
        for (int64_t r = 0; r < n_runs; ++r) {
            const double len = (double)(voff[(size_t)r + 1] - voff[(size_t)r]);
            sq += len * len;
        }
        L.obj_run = n > 0 ? sq / (double)n : 0.0;
    }
    L.h_off = std::move(voff);
    L.h_vbase = std::move(vbase);
    L.h_slo = std::move(slo);
    L.n_groups = n_groups;
    L.device_bytes = (int64_t)col * (c->w ? 4 : 3) + (want_bins ? n * (int64_t)sizeof(int32_t) : 0) + 3 * L.q_stride * (int64_t)sizeof(float) +
                     (4 * (n_runs + 1) + 2 * (int64_t)n_groups + 1) * (int64_t)sizeof(int64_t) + (n_runs + 1) * (int64_t)sizeof(RunGrid) +
                     (L.h_tiles[0][(size_t)n_runs] + L.h_tiles[1][(size_t)n_runs] + L.h_tiles[2][(size_t)n_runs]) * (int64_t)sizeof(TileRec);
    c->device_bytes += L.device_bytes;
    L.built = true;
    return bail(hipSuccess, "");
}

// Merged triple runs of a built strip layout (see k_merge_triples); built once, on first use as the streamed side of a
// float32 band kernel with partner strips c - 1, c, c + 1.
int build_triples(yawhip_ctx *ctx, yawhip_catalog *c, int o, bool seg) {
    StripLayout &L = seg ? c->seg[o] : c->strips[o];
    if (!L.built) return fail(YAWHIP_ERR_INVALID, "build_triples: layout not built");
    if (L.triples) return YAWHIP_OK;
    const int64_t n = c->n, G = L.n_groups, V = L.h_vbase[(size_t)G], V3 = V + 2 * G;
    if (3 * n >= (1ll << 31)) return fail(YAWHIP_ERR_INVALID, "build_triples: catalogue too large for 32-bit entry indices");
    std::vector<int64_t> off3((size_t)V3 + 1, 0);
    std::vector<int32_t> run_group((size_t)std::max<int64_t>(V, 1), 0);
    for (int64_t g = 0; g < G; ++g) {
        const int64_t lo = L.h_vbase[(size_t)g], hi = L.h_vbase[(size_t)g + 1];
        for (int64_t r = lo; r < hi; ++r) run_group[(size_t)r] = (int32_t)g;
        for (int64_t c_rel = 0; c_rel < hi - lo + 2; ++c_rel) {
            const int64_t t = lo + 2 * g + c_rel, rc = lo + c_rel - 1;
            int64_t len = 0;
            for (int64_t m = rc - 1; m <= rc + 1; ++m)
                if (m >= lo && m < hi) len += L.h_off[(size_t)m + 1] - L.h_off[(size_t)m];
            off3[(size_t)t + 1] = len;
        }
    }
    for (int64_t t = 0; t < V3; ++t) off3[(size_t)t + 1] += off3[(size_t)t];
    if (off3[(size_t)V3] != 3 * n) return fail(YAWHIP_ERR_HIP, "build_triples: %lld entries for %lld objects", (long long)off3[(size_t)V3], (long long)n);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n3 = (size_t)std::max<int64_t>(3 * n, 1);
    L.q3_stride = (int64_t)((n3 + 3) & ~(size_t)3) + 8;  // as q_stride: a 16-byte load may run up to 12 bytes past a column
    int32_t *d_run_group = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&L.q3), (size_t)3 * L.q3_stride * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&L.idx3), n3 * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&L.pos3), (size_t)std::max<int64_t>(n, 1) * sizeof(int32_t));
    if (e == hipSuccess && c->w) e = hipMalloc(reinterpret_cast<void **>(&L.w3), n3 * sizeof(double) + 16);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&L.off3), (size_t)(V3 + 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&L.d_grid3), (size_t)(V3 + 1) * sizeof(RunGrid));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_run_group), run_group.size() * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemsetAsync(L.q3, 0, (size_t)3 * L.q3_stride * sizeof(float), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(L.d_grid3, 0, (size_t)(V3 + 1) * sizeof(RunGrid), ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(L.off3, off3.data(), (size_t)(V3 + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_run_group, run_group.data(), run_group.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && n > 0) {
        hipLaunchKernelGGL(k_merge_triples, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, V, L.off, d_run_group,
                           L.d_vbase, L.off3, o == 0 ? L.x : (o == 1 ? L.y : L.z), L.q, L.q_stride, L.w, L.q3, L.q3_stride, L.w3, L.idx3, L.pos3);
        hipLaunchKernelGGL(k_run_grid<float>, dim3((unsigned)((V3 * (RUN_GRID + 1) + 255) / 256)), dim3(256), 0, ctx->stream, V3, L.off3,
                           L.q3 + (size_t)o * L.q3_stride, L.d_grid3);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (d_run_group) (void)hipFree(d_run_group);
    if (e != hipSuccess) {
        for (void *ptr : {(void *)L.q3, (void *)L.w3, (void *)L.idx3, (void *)L.pos3, (void *)L.off3, (void *)L.d_grid3})
            if (ptr) (void)hipFree(ptr);
        L.q3 = nullptr; L.w3 = nullptr; L.idx3 = nullptr; L.pos3 = nullptr; L.off3 = nullptr; L.d_grid3 = nullptr;
        return fail(e == hipErrorOutOfMemory ? YAWHIP_ERR_OOM : YAWHIP_ERR_HIP, "merged triple runs failed: %s", hipGetErrorString(e));
    }
    const int64_t bytes = 3 * L.q3_stride * (int64_t)sizeof(float) + (int64_t)n3 * (4 + (c->w ? 8 : 0)) + n * 4 + (V3 + 1) * (int64_t)(sizeof(int64_t) + sizeof(RunGrid));
    L.device_bytes += bytes;
    c->device_bytes += bytes;
    L.triples = true;
    return YAWHIP_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

const char *yawhip_last_error(void) { return g_last_error.c_str(); }
int yawhip_abi_version(void) { return YAWHIP_ABI_VERSION; }

int yawhip_device_count(int *n) {
    if (!n) return fail(YAWHIP_ERR_INVALID, "yawhip_device_count: n is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        return fail(YAWHIP_ERR_NO_DEVICE, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    *n = c;
    return YAWHIP_OK;
}

int yawhip_ctx_create(int device_id, yawhip_ctx **out) {
    if (!out) return fail(YAWHIP_ERR_INVALID, "yawhip_ctx_create: out is NULL");
    *out = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0)
        return fail(YAWHIP_ERR_NO_DEVICE, "no HIP device visible (the HIP path is mandatory; there is no CPU fallback)");
    if (device_id < 0 || device_id >= c)
        return fail(YAWHIP_ERR_NO_DEVICE, "device id %d out of range [0,%d)", device_id, c);
    HIP_TRY(hipSetDevice(device_id));
    yawhip_ctx *ctx = new (std::nothrow) yawhip_ctx();
    if (!ctx) return fail(YAWHIP_ERR_OOM, "host allocation failed");
    ctx->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = ctx->make_events();
    if (e != hipSuccess) {
        delete ctx;
        return fail(YAWHIP_ERR_HIP, "context setup failed: %s", hipGetErrorString(e));
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
        if (prop.sharedMemPerBlock > 0) ctx->lds_limit = (int)std::min<size_t>(prop.sharedMemPerBlock, 160 * 1024);
        if (prop.multiProcessorCount > 0) ctx->n_cu = prop.multiProcessorCount;
    }
    *out = ctx;
    return YAWHIP_OK;
}

int yawhip_ctx_destroy(yawhip_ctx *ctx) {
    if (!ctx) return YAWHIP_OK;
    for (yawhip_ctx *peer : ctx->peers) (void)yawhip_ctx_destroy(peer);
    ctx->peers.clear();
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    drop_plans(ctx, nullptr);
    ctx->release_all();
    for (CallBufs &pb : ctx->parked) pb.release_all();
    ctx->d_jobwork.release();
    ctx->d_full.release();
    ctx->d_rowidx.release();
    ctx->sort_ws.release();
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return YAWHIP_OK;
}

int yawhip_ctx_set_option(yawhip_ctx *ctx, const char *key, int64_t value) {
    if (!ctx || !key) return fail(YAWHIP_ERR_INVALID, "yawhip_ctx_set_option: NULL argument");
    for (yawhip_ctx *peer : ctx->peers) {  // every device of a multi-device context follows
        const int rc = yawhip_ctx_set_option(peer, key, value);
        if (rc != YAWHIP_OK) return rc;
    }
    ctx->plan.key = 0;  // options change the work per job
    ++ctx->opt_gen;     // ... and every decision of a plan
    drop_plans(ctx, nullptr);
    if (!strcmp(key, "half_bands")) {
        ctx->half_bands = value != 0;
        return YAWHIP_OK;
    }
    if (!strcmp(key, "tile_r")) {
        if (value != 0 && value != 1 && value != 2 && value != 4)
            return fail(YAWHIP_ERR_INVALID, "tile_r must be 0 (auto), 1, 2 or 4");
        ctx->tile_r = (int)value;
        return YAWHIP_OK;
    }
    if (!strcmp(key, "band_batch_log2")) {
        if (value < -1 || value > 6) return fail(YAWHIP_ERR_INVALID, "band_batch_log2 must be -1 (auto) or 0..6");
        ctx->band_batch_log2 = (int)value;
        return YAWHIP_OK;
    }
    if (!strcmp(key, "hist_copies_log2")) {
        if (value < -1 || value > 6) return fail(YAWHIP_ERR_INVALID, "hist_copies_log2 must be -1 (auto) or 0..6");
        ctx->hist_copies_log2 = (int)value;
        return YAWHIP_OK;
    }
    if (!strcmp(key, "triple_runs")) {  // 0: never, 1: where the merged window fits the stage, 2: wherever the partner strips are c - 1, c, c + 1
        if (value < 0 || value > 2) return fail(YAWHIP_ERR_INVALID, "triple_runs must be 0, 1 or 2");
        ctx->triple_runs = (int)value;
        return YAWHIP_OK;
    }
    if (!strcmp(key, "item_segments")) {
        ctx->item_segments = value != 0;
        return YAWHIP_OK;
    }
    if (!strcmp(key, "spin_wait")) {
        ctx->spin_wait = value != 0;
        return YAWHIP_OK;
    }
    if (!strcmp(key, "band_cap")) {
        if (value != 0 && value != BCAP && value != BCAP_MID && value != B32_CAP && value != B32_CAP_BIG)
            return fail(YAWHIP_ERR_INVALID, "band_cap must be 0 (auto), 192 or 288 (float64 / fine-grid band kernels), %d or %d (float32 band kernel)",
                        B32_CAP, B32_CAP_BIG);
        ctx->band_cap = (int)value;
        return YAWHIP_OK;
    }
    if (!strcmp(key, "strip_width_micro")) {  // strip grid spacing in units of 1e-6 (0 = off)
        if (value != 0 && (value < 1000 || value > 2000000))
            return fail(YAWHIP_ERR_INVALID, "strip_width_micro must be 0 (off) or in [1e3, 2e6]");
        ctx->strip_width = (double)value * 1e-6;
        return YAWHIP_OK;
    }
    if (!strcmp(key, "seg_strips_min_run")) {
        if (value < 1) return fail(YAWHIP_ERR_INVALID, "seg_strips_min_run must be >= 1");
        ctx->seg_min_run = (int)std::min<int64_t>(value, INT32_MAX);
        return YAWHIP_OK;
    }
    if (!strcmp(key, "seg_strips")) {
        ctx->seg_strips = value != 0;
        return YAWHIP_OK;
    }
    if (!strcmp(key, "debug_no_hits")) {
        ctx->debug_no_hits = value != 0;
        return YAWHIP_OK;
    }
    if (!strcmp(key, "auto_orient")) {
        ctx->auto_orient = value != 0;
        return YAWHIP_OK;
    }
    if (!strcmp(key, "slab_budget_bytes")) {
        if (value < 4096) return fail(YAWHIP_ERR_INVALID, "slab_budget_bytes must be >= 4096");
        ctx->slab_budget = value;
        return YAWHIP_OK;
    }
    if (!strcmp(key, "band_grid_div")) {
        if (value < 0 || value > 64) return fail(YAWHIP_ERR_INVALID, "band_grid_div must be 0 (auto) or in [1, 64]");
        ctx->band_grid_div = (int)value;
        return YAWHIP_OK;
    }
    if (!strcmp(key, "flush_stages_log2")) {
        if (value < 0 || value > 17) return fail(YAWHIP_ERR_INVALID, "flush_stages_log2 must be in [0, 17]");
        ctx->flush_log2 = (int)value;
        return YAWHIP_OK;
    }
    if (!strcmp(key, "band_fp32")) {
        ctx->band_fp32 = value != 0;
        return YAWHIP_OK;
    }
    if (!strcmp(key, "kernel")) {
        if (value < YAWHIP_KERNEL_AUTO || value > YAWHIP_KERNEL_BAND)
            return fail(YAWHIP_ERR_INVALID, "unknown kernel id %lld", (long long)value);
        ctx->default_kernel = (int)value;
        return YAWHIP_OK;
    }
    return fail(YAWHIP_ERR_INVALID, "unknown option '%s'", key);
}

int yawhip_catalog_upload(yawhip_ctx *ctx, int64_t n, const double *x, const double *y, const double *z,
                          const double *w, int32_t n_patches, int32_t n_bins_or_1, const int64_t *offsets,
                          yawhip_catalog **out) {
    return yawhip_catalog_upload_axis(ctx, n, x, y, z, w, n_patches, n_bins_or_1, offsets, 2, out);
}

int yawhip_catalog_sort_axis(const yawhip_catalog *cat, int32_t *axis) {
    if (!cat || !axis) return fail(YAWHIP_ERR_INVALID, "yawhip_catalog_sort_axis: NULL argument");
    *axis = cat->axis;
    return YAWHIP_OK;
}

int yawhip_catalog_upload_axis(yawhip_ctx *ctx, int64_t n, const double *x, const double *y, const double *z,
                               const double *w, int32_t n_patches, int32_t n_bins_or_1, const int64_t *offsets,
                               int32_t sort_axis, yawhip_catalog **out) {
    if (!out) return fail(YAWHIP_ERR_INVALID, "yawhip_catalog_upload: out is NULL");
    if (sort_axis < 0 || sort_axis > 2) return fail(YAWHIP_ERR_INVALID, "sort_axis must be 0 (x), 1 (y) or 2 (z)");
    *out = nullptr;
    if (!ctx) return fail(YAWHIP_ERR_INVALID, "yawhip_catalog_upload: ctx is NULL");
    if (n < 0 || n_patches <= 0 || n_bins_or_1 <= 0 || !offsets || (n > 0 && (!x || !y || !z)))
        return fail(YAWHIP_ERR_INVALID, "yawhip_catalog_upload: bad sizes or NULL columns");
    const int64_t nseg = (int64_t)n_patches * n_bins_or_1;
    if (n >= (1ll << 32)) return fail(YAWHIP_ERR_INVALID, "at most 2^32 - 1 objects per catalogue");
    if (offsets[0] != 0 || offsets[nseg] != n) return fail(YAWHIP_ERR_INVALID, "offsets must start at 0 and end at n");
    for (int64_t i = 0; i < nseg; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(YAWHIP_ERR_INVALID, "offsets must be non-decreasing");
    HIP_TRY(hipSetDevice(ctx->device));
    yawhip_catalog *c = new (std::nothrow) yawhip_catalog();
    static std::atomic<uint64_t> next_uid{1};
    if (c) c->uid = next_uid.fetch_add(1);
    if (!c) return fail(YAWHIP_ERR_OOM, "host allocation failed");
    c->ctx = ctx;
    c->n = n;
    c->n_patches = n_patches;
    c->nb = n_bins_or_1;
    c->axis = sort_axis;
    c->h_off.assign(offsets, offsets + nseg + 1);
    // Library-private order: the columns go to the device as they are and are ordered there (rocPRIM radix sorts,
    // yawhip_sort.hip): ascending along the sort axis inside every (patch, bin) segment. The strip layouts are derived
    // from this resident copy (build_strip_layout), the one of the catalogue's own sort axis right away.
    const size_t col = (size_t)std::max<int64_t>(n, 1) * sizeof(double) + 16;  // + 16: see build_strip_layout
    double *rx = nullptr, *ry = nullptr, *rz = nullptr, *rw = nullptr;  // raw columns (temporary)
    uint32_t *perm = nullptr;
    int64_t *poff = nullptr;
    unsigned long long *box = nullptr;  // [P][6] sortable images of min / max per axis, [6 P]: violations of the unit norm
    auto free_tmp = [&]() {
        for (void *q : {(void *)rx, (void *)ry, (void *)rz, (void *)rw, (void *)perm, (void *)poff, (void *)box})
            if (q) (void)hipFree(q);
    };
    auto bail = [&](hipError_t err, const char *what) {
        free_tmp();
        yawhip_catalog_free(c);
        return fail(err == hipErrorOutOfMemory ? YAWHIP_ERR_OOM : YAWHIP_ERR_HIP, "catalog upload (%s) failed: %s", what,
                    hipGetErrorString(err));
    };
    std::vector<int64_t> h_poff((size_t)n_patches + 1);
    for (int p = 0; p <= n_patches; ++p) h_poff[(size_t)p] = offsets[(int64_t)p * n_bins_or_1];
    std::vector<unsigned long long> h_box((size_t)6 * n_patches + 1);
    for (int p = 0; p < n_patches; ++p)
        for (int a = 0; a < 3; ++a) {
            h_box[(size_t)6 * p + a] = sortable_of(4.0);       // running minimum
            h_box[(size_t)6 * p + 3 + a] = sortable_of(-4.0);  // running maximum
        }
    h_box[(size_t)6 * n_patches] = 0ull;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&c->x), col);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->y), col);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->z), col);
    if (e == hipSuccess && w) e = hipMalloc(reinterpret_cast<void **>(&c->w), col);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->off), (size_t)(nseg + 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&rx), col);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&ry), col);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&rz), col);
    if (e == hipSuccess && w) e = hipMalloc(reinterpret_cast<void **>(&rw), col);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&perm), (size_t)std::max<int64_t>(n, 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&poff), (size_t)(n_patches + 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&box), h_box.size() * sizeof(unsigned long long));
    if (e == hipSuccess && n > 0) {
        e = hipMemcpyAsync(rx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(ry, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(rz, z, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && w) e = hipMemcpyAsync(rw, w, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(c->off, offsets, (size_t)(nseg + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(poff, h_poff.data(), (size_t)(n_patches + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(box, h_box.data(), h_box.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return bail(e, "columns");
    const unsigned ngrid = (unsigned)((std::max<int64_t>(n, 1) + 255) / 256);
    if (n > 0) {
        // bounding box of every patch (the orientation of a job follows from the boxes of its two patches) and the
        // unit-norm check of the pre-filter, both on the device
        hipLaunchKernelGGL(k_patch_boxes, dim3(ngrid), dim3(256), 0, ctx->stream, n, rx, ry, rz, poff, n_patches, box);
        e = yawsort::sort_segments(ctx->sort_ws, ctx->stream, n, key_of(rx, ry, rz, sort_axis), c->off, nseg, perm);
        if (e != hipSuccess) return bail(e, "segment sort");
        hipLaunchKernelGGL(k_gather_columns, dim3(ngrid), dim3(256), 0, ctx->stream, n, perm, rx, ry, rz, rw, c->x, c->y, c->z, c->w);
        if ((e = hipGetLastError()) != hipSuccess) return bail(e, "gather");
        e = hipMemcpyAsync(h_box.data(), box, h_box.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
        if (e != hipSuccess) return bail(e, "patch boxes");
    }
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return bail(e, "finish");
    free_tmp();
    c->unit_norm = h_box[(size_t)6 * n_patches] == 0ull;
    c->h_box.resize((size_t)6 * n_patches);
    for (size_t i = 0; i < c->h_box.size(); ++i) c->h_box[i] = double_of(h_box[i]);
    c->device_bytes = (int64_t)col * (w ? 4 : 3) + (nseg + 1) * (int64_t)sizeof(int64_t);
    c->strip_width = ctx->strip_width;
    c->has_strips = c->unit_norm && n > 0;
    if (c->has_strips) {
        const int rc = build_strip_layout(ctx, c, sort_axis, false);
        if (rc != YAWHIP_OK) {
            yawhip_catalog_free(c);
            return rc;
        }
    }
    if (ctx->sort_ws.cap > ((size_t)1 << 25)) ctx->sort_ws.release();  // ~30 bytes per object: keep only small workspaces
    for (yawhip_ctx *peer : ctx->peers) {  // multi-device context: the same catalogue on every further device
        yawhip_catalog *rep = nullptr;
        const int rc = yawhip_catalog_upload_axis(peer, n, x, y, z, w, n_patches, n_bins_or_1, offsets, sort_axis, &rep);
        if (rc != YAWHIP_OK) {
            yawhip_catalog_free(c);
            return rc;
        }
        c->replicas.push_back(rep);
    }
    *out = c;
    return YAWHIP_OK;
}

int yawhip_catalog_free(yawhip_catalog *c) {
    if (!c) return YAWHIP_OK;
    for (yawhip_catalog *rep : c->replicas) (void)yawhip_catalog_free(rep);
    c->replicas.clear();
    if (c->ctx) c->ctx->plan.key = 0;  // a later catalogue may reuse the address the plan was keyed on
    if (c->ctx) (void)hipSetDevice(c->ctx->device);
    if (c->ctx) {  // its plans hold pointers into its layouts (nothing of them is in flight: calls are blocking)
        if (c->ctx->stream) (void)hipStreamSynchronize(c->ctx->stream);
        drop_plans(c->ctx, c);
    }
    if (c->x) (void)hipFree(c->x);
    if (c->y) (void)hipFree(c->y);
    if (c->z) (void)hipFree(c->z);
    if (c->w) (void)hipFree(c->w);
    if (c->off) (void)hipFree(c->off);
    for (int o = 0; o < 3; ++o) {
        c->strips[o].release();
        c->seg[o].release();
    }
    delete c;
    return YAWHIP_OK;
}

int yawhip_catalog_device_bytes(const yawhip_catalog *cat, int64_t *bytes) {
    if (!cat || !bytes) return fail(YAWHIP_ERR_INVALID, "yawhip_catalog_device_bytes: NULL argument");
    *bytes = cat->device_bytes;
    return YAWHIP_OK;
}

}  // extern "C"

namespace {

// What count_finish needs to know about a call count_enqueue has put on a context's stream.
struct CallState {
    std::chrono::steady_clock::time_point wall0;
    bool pending = false;          // something was enqueued (false: nothing to count, outputs are zero)
    int64_t n_out = 0;
    size_t o_ctr = 0, o_counts = 0, o_sums = 0;
    bool want_counts = false, want_sums = false, band_ran = false, run_unweighted = false, run_weighted = false;
    bool segmented = false;        // the item list was kept in segments: the kept items are the sum of the segment counters
    int64_t cand = 0, abytes = 0, n_pot = 0;
    int launches = 0, kernel = 0, mode = 0, n_orient = 0, band_variant = 0, merged_triples = 0;
};

// Float32 bounds of every edge for k_count_band32 (see there): for unit vectors rounded to float32,
//   |s32 - s| <= g(t) = 2.1e-7 sqrt(t) + 5e-7 t + 1e-12 near s = t,
// so s32 < t - g proves s <= t and s32 > t + g proves s > t; in between the kernel evaluates in float64.
//   n_edges == 2: {c, h_in, h_out, 0}: |s32 - c| < h_in proves t0 < s <= t1, |s32 - c| >= h_out proves the opposite
//                 (s32 - c is what the kernel's three fused multiply-adds deliver; both widths carry their rounding);
//   else per edge {t - g rounded down, t + g rounded up}.
std::vector<float> build_thr32(const double *t, int n_bins, int n_edges) {
    auto down = [](double v) { float f = (float)v; if ((double)f > v) f = nextafterf(f, -INFINITY); return f; };
    auto up = [](double v) { float f = (float)v; if ((double)f < v) f = nextafterf(f, INFINITY); return f; };
    auto guard = [](double te) { return BAND32_GUARD_SQRT * std::sqrt(te) + 5e-7 * te + 1e-12; };
    const int tw = thr32_width(n_edges);
    std::vector<float> out((size_t)n_bins * tw, 0.f);
    for (int k = 0; k < n_bins; ++k) {
        const double *tk = t + (size_t)k * n_edges;
        float *row = &out[(size_t)k * tw];
        if (n_edges == 2) {
            const double g0 = guard(tk[0]), g1 = guard(tk[1]);
            const float c = (float)(0.5 * (tk[0] + tk[1]));
            const double cd = (double)c;
            // q = fma(dz, dz, fma(dy, dy, fma(dx, dx, -c))): three roundings of intermediate sums that stay below 2 c wherever
            // a class is claimed (|q| < h_in <= c, or s inside the annulus: s32 <= t1 + g1 <= 2 c) -> 6 x 2^-24 x c = 3.6e-7 c
            const double fold = 4e-7 * cd;
            const double h_in = (std::min(cd - (tk[0] + g0), (tk[1] - g1) - cd) - fold) * (1.0 - 1e-6);
            const double h_out = (std::max(cd - (tk[0] - g0), (tk[1] + g1) - cd) + fold) * (1.0 + 1e-6);
            row[0] = c;
            row[1] = h_in > 0.0 ? down(h_in) : 0.f;   // |q| < 0 never holds: nothing is certain
            row[2] = up(std::max(h_out, 0.0));
            row[3] = 0.f;
        } else {
            for (int e = 0; e < n_edges; ++e) {
                const double g = guard(tk[e]);
                row[2 * e] = down(tk[e] - g);
                row[2 * e + 1] = up(tk[e] + g);
            }
        }
    }
    return out;
}

// Float32 table of k_count_band32_fine (see there), one row per redshift bin: {m, a, e0, e1}, then {t_j - g, t_j + g} per edge.
// Empty when the edges of some bin do not follow the log-spaced model closely enough for float32 (the caller then counts
// with the float64 band kernel): deviation above 0.05 fine bins, a guard wider than a fifth of a fine bin, t_0 = 0.
std::vector<float> build_fine32(const double *t, int n_bins, int n_edges) {
    auto down = [](double v) { float f = (float)v; if ((double)f > v) f = nextafterf(f, -INFINITY); return f; };
    auto up = [](double v) { float f = (float)v; if ((double)f < v) f = nextafterf(f, INFINITY); return f; };
    auto guard = [](double te) { return BAND32_GUARD_SQRT * std::sqrt(te) + 5e-7 * te + 1e-12; };
    const int tw = fine32_width(n_edges), nf = n_edges - 1;
    std::vector<float> out((size_t)n_bins * tw, 0.f);
    for (int k = 0; k < n_bins; ++k) {
        const double *tk = t + (size_t)k * n_edges;
        if (!(tk[0] > 1e-12) || !(tk[nf] > tk[0])) return {};
        const double l0 = std::log2(tk[0]), l1 = std::log2(tk[nf]);
        const double m = (double)nf / (l1 - l0), a = l0 * m;
        double dev = 0.0;
        for (int j = 0; j <= nf; ++j) {
            if (j > 0 && !(tk[j] > tk[j - 1])) return {};
            dev = std::max(dev, std::fabs((std::log2(tk[j]) - l0) * m - (double)j));
        }
        // error of the device's f: hardware log2 (1 ulp of a result below 64), float32 images of m and a, the fma
        const double dev_f = m * (1e-5 + 6e-8 * 64.0) + 2.0 * 6e-8 * std::fabs(a) + 4e-5 + 6e-8 * (nf + 2);
        const double per_s = 1.05 * m / std::log(2.0);  // d f / (d s / s), with room for the second order
        // Admission: an s32 between the guard bands of edges j and j + 1 must round to one of the two, i.e. f may be off by
        // less than half a bin: the model's deviation at the edges, the device's arithmetic, and the guard (widest,
        // relative to t, at the first edge).
        if (dev + dev_f + per_s * guard(tk[0]) / tk[0] > 0.45) return {};
        float *row = &out[(size_t)k * tw];
        row[0] = (float)m; row[1] = (float)a; row[2] = 0.f; row[3] = 0.f;
        for (int j = 0; j <= nf; ++j) {
            const double g = guard(tk[j]);
            row[4 + 2 * j] = down(tk[j] - g);
            row[5 + 2 * j] = up(tk[j] + g);
        }
    }
    return out;
}

// What a count call derives from its inputs on the HOST before anything is launched -- kernel choice, layouts, tile and stage
// sizes, the job records / prefix / threshold tables of the item builder and the count kernel (uploaded once, into the plan's
// own device buffer) -- kept for the next call with the same inputs: the same catalogue pair (by upload id), job list,
// thresholds (both compared byte for byte), kernel, outputs asked for and option set. A repeated call (the next step of a
// bench, the same count of the next measurement, DR after DD with the same job list is ANOTHER plan) then marshals no
// tables at all; the item builder and the count kernels run every call. Plans die with their catalogues and options.
struct HostPlan {
    // identity
    uint64_t hash = 0, stamp = 0, c1_uid = 0, c2_uid = 0, opt_gen = 0;
    int32_t n_jobs_in = 0, n_bins_in = 0, n_edges_in = 0, kernel_in = 0;
    bool want_counts = false, want_sums = false, for_work = false;
    std::vector<int32_t> jobs_in;
    std::vector<double> t_in;
    // decisions
    bool empty = false;   // nothing to count (no output values)
    bool split = false;   // the job list has to be counted in pieces (SPLIT_JOBS)
    int grid_div = 8;  // band kernels: workgroups = potential items / this
    int R = 0, band_ne = 0, cap = 0, hp_shift = 0, lean_bins = 0, mode = 0, reach = 0, kernel = 0, nf = 0, n_orient = 0;
    bool band = false, band32 = false, band_fine = false, filter = false, lean = false, merged = false, run_unweighted = false,
         run_weighted = false, strip_items = false, swap = false, sweep = false, triple = false, uniform_t = false, weighted = false,
         weighted_any = false;
    int64_t abytes = 0, cand = 0, n_items = 0, n_out = 0, n_pslots = 0, n_sjobs = 0, n_slots = 0, slab = 0, tile = 0;
    double rwin_max = 0.0;
    size_t lds_band = 0, lds_merged = 0;
    // device tables (one allocation): jobs / job records, prefix, thresholds, pre-filter thresholds, window widths, float32
    // classes, layout table, and -- weighted calls -- the chunk prefix of the slab reduction
    unsigned char *d_in = nullptr;
    size_t o_jobs = 0, o_prefix = 0, o_t = 0, o_dthr = 0, o_rwin = 0, o_thr32 = 0, o_tabs = 0, o_cprefix = 0;
    int64_t n_chunks = 0, n_oslots = 0;
    ~HostPlan() { if (d_in) (void)hipFree(d_in); }
};

// Forget the plans that involve catalogue `c` (nullptr: all of them).
void drop_plans(yawhip_ctx *ctx, const yawhip_catalog *c) {
    for (size_t i = 0; i < ctx->plans.size();) {
        if (!c || ctx->plans[i]->c1_uid == c->uid || ctx->plans[i]->c2_uid == c->uid) {
            delete ctx->plans[i];
            ctx->plans[i] = ctx->plans.back();
            ctx->plans.pop_back();
        } else ++i;
    }
}

// The host half of a count call: validation, every decision, the tables -- into a plan (see HostPlan).
int make_plan(yawhip_ctx *ctx, const yawhip_catalog *c1, const yawhip_catalog *c2, int32_t n_jobs, const int32_t *jobs,
              int32_t n_bins, int32_t n_edges, const double *t, int32_t kernel, bool want_counts, bool want_sums, bool for_work,
              HostPlan &P) {
    const void *job_work = for_work ? static_cast<const void *>(&P) : nullptr;  // (only its truth value matters below)
    if (c1->ctx != ctx || c2->ctx != ctx) return fail(YAWHIP_ERR_MISMATCH, "catalogues belong to another context");
    if (c1->n_patches != c2->n_patches)
        return fail(YAWHIP_ERR_MISMATCH, "patch counts differ (%d vs %d)", c1->n_patches, c2->n_patches);
    if (n_jobs < 0 || n_bins <= 0 || n_edges < 2 || n_edges > MAX_EDGES || !t || (n_jobs > 0 && !jobs))
        return fail(YAWHIP_ERR_INVALID, "yawhip_count_pairs: bad sizes (n_jobs=%d n_bins=%d n_edges=%d, max edges %d)",
                    n_jobs, n_bins, n_edges, MAX_EDGES);
    if ((c1->nb != 1 && c1->nb != n_bins) || (c2->nb != 1 && c2->nb != n_bins))
        return fail(YAWHIP_ERR_MISMATCH, "catalogue bin counts (%d, %d) do not fit n_bins=%d", c1->nb, c2->nb, n_bins);
    for (int k = 0; k < n_bins; ++k)
        for (int e = 0; e < n_edges; ++e) {
            const double v = t[(size_t)k * n_edges + e];
            if (!(v >= 0.0) || (e > 0 && !(v >= t[(size_t)k * n_edges + e - 1])))
                return fail(YAWHIP_ERR_INVALID, "thresholds of bin %d are not ascending non-negative numbers", k);
        }
    for (int j = 0; j < n_jobs; ++j)
        if (jobs[2 * j] < 0 || jobs[2 * j] >= c1->n_patches || jobs[2 * j + 1] < 0 || jobs[2 * j + 1] >= c1->n_patches)
            return fail(YAWHIP_ERR_INVALID, "job %d has a patch id outside [0,%d)", j, c1->n_patches);
    if (kernel == YAWHIP_KERNEL_AUTO) kernel = ctx->default_kernel;
    const bool auto_pick = kernel == YAWHIP_KERNEL_AUTO;  // BAND or SWEEP, whichever suits the layouts (decided below)
    if (kernel == YAWHIP_KERNEL_AUTO) kernel = YAWHIP_KERNEL_BAND;
    if (kernel < YAWHIP_KERNEL_EXACT || kernel > YAWHIP_KERNEL_BAND)
        return fail(YAWHIP_ERR_INVALID, "unknown kernel id %d", kernel);
    // the FP32 pre-filter assumes unit vectors; anything else is evaluated pair by pair in FP64
    const bool unit = c1->unit_norm && c2->unit_norm;
    if (kernel == YAWHIP_KERNEL_FILTER && !unit) kernel = YAWHIP_KERNEL_EXACT;
    // the window search compares the sorted coordinate of both sides: the axes must agree
    if ((kernel == YAWHIP_KERNEL_SWEEP || kernel == YAWHIP_KERNEL_BAND) && c1->axis != c2->axis)
        kernel = unit ? YAWHIP_KERNEL_FILTER : YAWHIP_KERNEL_EXACT;
    // the band kernels park finished lanes on a sentinel at coordinate 4.0 and bound their searches by it: unit vectors only
    if (kernel == YAWHIP_KERNEL_BAND && !unit) kernel = YAWHIP_KERNEL_EXACT;
    bool band = kernel == YAWHIP_KERNEL_BAND;
    const bool sweep = kernel == YAWHIP_KERNEL_SWEEP || band;
    const bool filter = unit && kernel != YAWHIP_KERNEL_EXACT;

    const int nf = n_edges - 1;
    const int64_t n_slots = (int64_t)n_jobs * n_bins;
    const int64_t n_out = n_slots * nf;
    const bool weighted = (c1->w != nullptr) || (c2->w != nullptr);
    P.n_out = n_out;
    if (n_out == 0) { P.empty = true; return YAWHIP_OK; }
    if (n_slots > (1ll << 30)) return fail(YAWHIP_ERR_INVALID, "too many (job,bin) slots");
    HIP_TRY(hipSetDevice(ctx->device));

    // tile size: objects per lane. Larger tiles amortise the streamed-object read; small segments
    // prefer small tiles so that padded lanes do not dominate.
    // Lean path (k_count_merged): z-window culling + FP32 pre-filter + queued exact evaluation. Its merged
    // form (one item for all bins, strip layouts on both sides) serves c1 binned x c2 unbinned, i.e. every
    // count of a cross-correlation.
    const bool weighted_any = (c1->w != nullptr) || (c2->w != nullptr);
    const bool lean = sweep && (filter || band);  // single-wave workgroups on windowed items (k_count_merged / k_count_band)
    double rwin_max = 0.0;  // widest window half width over the bins
    for (int k = 0; k < n_bins; ++k)
        rwin_max = std::max(rwin_max, std::sqrt(t[(size_t)k * n_edges + n_edges - 1]) * (1.0 + 1e-12) + 1e-15);
    // strip pairing pays while a run has few partner runs; for separations far beyond the grid spacing the
    // ordinary (patch, bin) layout is used instead
    const bool strips = lean && c1->has_strips && c2->has_strips && c1->strip_width == c2->strip_width &&
                        (c1->strip_width <= 0.0 || rwin_max / c1->strip_width <= (double)MAX_STRIP_REACH);
    // mode 3: binned x binned on the per-segment strip layouts: ordinary (job, bin) items whose lane tiles and windows
    // come from (patch, bin, strip) runs -- it pays when the lane side is dense: runs of at least a few lane tiles per
    // (patch, bin, strip); estimated from the patch-level layout the upload built (B times as many runs)
    bool seg_ok = false;
    if (strips && c1->nb == n_bins && c2->nb == n_bins && n_bins > 1 && ctx->seg_strips) {
        const StripLayout &base2 = c2->strips[c2->axis];
        const int64_t seg_runs = base2.h_vbase[(size_t)base2.n_groups] * (int64_t)c2->nb;
        seg_ok = c2->n / std::max<int64_t>(seg_runs, 1) >= ctx->seg_min_run;
    }
    bool uniform_t = true;  // every bin has the same threshold row (angular scales)
    for (int k = 1; k < n_bins && uniform_t; ++k)
        uniform_t = memcmp(t, t + (size_t)k * n_edges, sizeof(double) * n_edges) == 0;
    // One item for all bins needs a histogram of B x (E - 1) cells (and B edge rows when they differ) in LDS. Where that
    // does not fit (hundreds of bins times dozens of separation-weight bins), the count falls back to ordinary
    // (job, bin) items, whose histogram has E - 1 cells.
    const size_t merged_lds = (size_t)n_bins * nf * (weighted_any ? 8 : 4) + (size_t)(uniform_t ? 1 : n_bins) * n_edges * sizeof(double) +
                              (size_t)n_bins * sizeof(float) + BandLds<BCAP_MID>::FIXED + (BCAP_MID + 2) * 8 + 2 * MSTAGE * sizeof(ObjF) + 1024;
    const bool merged_fits = merged_lds <= (size_t)ctx->lds_limit;
    const int mode = !strips ? 0 : (c1->nb > 1 && c2->nb == 1) ? (merged_fits ? 1 : 0) : (seg_ok ? 3 : 0);
    const bool merged = mode == 1;                // one item covers all bins, output slot = job
    const bool strip_items = mode != 0;           // items come from strip runs (k_build_items_strips)
    // Orientation of every job: the (u, v) projection that compresses the sphere least around its two patches, i.e.
    // the one that drops the coordinate w in which the patches lie farthest from the origin. (Projected along an
    // axis the patches are nearly perpendicular to, objects pile up in (u, v) -- density grows like 1 / |w| -- and
    // the opposite hemisphere folds onto the same cells: every u-window then holds several times the partners.)
    std::vector<int32_t> orient((size_t)n_jobs, (int32_t)c1->axis);
    const StripLayout *L1[3] = {nullptr, nullptr, nullptr}, *L2[3] = {nullptr, nullptr, nullptr};
    if (strip_items) {
        bool need[3] = {false, false, false};
        for (int j = 0; j < n_jobs; ++j) {
            if (ctx->auto_orient) {
                const double *b1 = &c1->h_box[(size_t)6 * jobs[2 * j]], *b2 = &c2->h_box[(size_t)6 * jobs[2 * j + 1]];
                double best = -1.0;
                int wax = (c1->axis + 1) % 3;
                for (int a = 0; a < 3; ++a) {
                    const double m = (b1[a] <= b1[3 + a] ? 0.5 * (b1[a] + b1[3 + a]) : 0.0) +
                                     (b2[a] <= b2[3 + a] ? 0.5 * (b2[a] + b2[3 + a]) : 0.0);
                    if (std::fabs(m) > best) { best = std::fabs(m); wax = a; }
                }
                orient[(size_t)j] = (wax + 2) % 3;  // sort axis u whose dropped axis (u + 1) % 3 is wax
            }
            need[orient[(size_t)j]] = true;
        }
        for (int o = 0; o < 3; ++o) {
            if (!need[o]) continue;
            int rc = build_strip_layout(ctx, const_cast<yawhip_catalog *>(c1), o, mode == 3);
            if (rc == YAWHIP_OK && c2 != c1) rc = build_strip_layout(ctx, const_cast<yawhip_catalog *>(c2), o, mode == 3);
            if (rc != YAWHIP_OK) return rc;
            L1[o] = mode == 3 ? &c1->seg[o] : &c1->strips[o];
            L2[o] = mode == 3 ? &c2->seg[o] : &c2->strips[o];
        }
    }
    // Float32 classification (k_count_band32) on strip layouts of unit vectors with up to four edges per bin. Where one
    // side is binned (merged items) the roles are swapped against k_count_band: lane tiles come from the binned catalogue
    // c1, the windows from the unbinned c2 (see the kernel).
    const bool want32 = band && strip_items && unit && n_edges <= 4 && ctx->band_fp32 != 0 &&
                        band32_lds(weighted_any, BCAP_MID, (merged ? n_bins : 1) * nf, merged && !uniform_t ? n_bins : 0, n_edges) <=
                            (size_t)ctx->lds_limit;
    // ... and the fine radial grids of separation weights (k_count_band32_fine), when their edges follow the log-spaced model
    std::vector<float> fine32;
    if (band && strip_items && unit && n_edges > 4 && ctx->band_fp32 != 0 &&
        band32_fine_lds(weighted_any, BCAP_MID, (merged ? n_bins : 1) * nf, uniform_t ? 1 : n_bins, n_edges) <= (size_t)ctx->lds_limit)
        fine32 = build_fine32(t, n_bins, n_edges);
    const bool want_fine = !fine32.empty();
    // (Binned x binned counts of two different catalogues keep c2 on the lanes whichever is sparser: with the 10M data on the
    // lanes and the 100M randoms streamed, DR of config #4 has 570 k items instead of 1.28 M but walks 2.3 x the entries --
    // neighbouring lane objects of a sparse run lie far apart, their common band is long -- 4.1 against 2.2 ms.)
    bool swap = (want32 || want_fine) && merged;
    const yawhip_catalog *c_lane = swap ? c1 : c2, *c_strm = swap ? c2 : c1;
    const StripLayout *const *LL = swap ? L1 : L2, *const *LS = swap ? L2 : L1;  // lane side, streamed side
    if (auto_pick && band && unit) {
        // The band kernel decides every entry of a per-object band: unbeatable while a band is a handful of entries of which
        // half are pairs (strip layouts). Without strips a band is the whole u-window of a segment, nearly all of it far away
        // along v -- the FP32 pre-filter of the sweep kernel is made for that. Sparse streamed runs (a few dozen objects: an
        // item is all fixed cost) went to the sweep kernel while the band kernel evaluated in float64; the float32 band kernel
        // has the smaller fixed cost (measured: DD of config #4 0.61 against 0.83 ms, 1M x 1M / 64 patches 0.047 against 0.102,
        // 3M x 0.3M 0.09 against 0.28; sweep stays ahead only where the two sides differ tenfold in density and the items are
        // tiny: DR of config #4 2.09 against 2.23 ms, 0.3M x 3M 0.092 against 0.119) -- so only the float64 band kernel
        // (band_fp32 = 0, more than four edges off the log grid) keeps the density rule.
        bool use_sweep = mode == 0;
        if (!use_sweep && !want32 && !want_fine) {
            double obj_run = 0.0;  // of the densest built orientation
            for (int o = 0; o < 3; ++o)
                if (LS[o]) obj_run = std::max(obj_run, LS[o]->obj_run);
            use_sweep = obj_run < (double)BAND_MIN_STREAM_RUN;
        }
        if (use_sweep) {
            kernel = YAWHIP_KERNEL_SWEEP;
            band = false;
        }
    }
    const bool band32 = want32 && band, band_fine = want_fine && band;
    if (!band32 && !band_fine && swap) {  // the sweep kernel streams c1 past lane tiles of c2
        swap = false;
        c_lane = c2; c_strm = c1; LL = L2; LS = L1;
    }
    int R = ctx->tile_r;
    double est_window = 0.0;  // band kernel: expected entries of one window
    if (R == 0) {
        int64_t max_seg = 0;
        if (strip_items) {  // lanes hold runs of a strip layout: their typical (mean) length decides
            int64_t n_runs = 1;
            for (int o = 0; o < 3; ++o)
                if (LL[o]) n_runs = std::max(n_runs, LL[o]->h_vbase[(size_t)LL[o]->n_groups]);
            max_seg = c_lane->n / std::max<int64_t>(n_runs, 1);
            if (mode == 3) max_seg = std::max<int64_t>(max_seg, 4 * MWG * 2);  // at least two objects per lane: per-bin runs are
                                                                                // sparse, the per-item cost outweighs the wider window
        } else {
            for (int j = 0; j < n_jobs; ++j)
                for (int k = 0; k < (c2->nb == 1 ? 1 : n_bins); ++k) max_seg = std::max(max_seg, seg_len(c2, jobs[2 * j + 1], k));
        }
        const int wg = lean ? MWG : WG;
        R = max_seg >= 8 * wg * 4 ? 4 : (max_seg >= 4 * wg * 2 ? 2 : 1);
        if (strip_items && R > 2) R = 2;  // on strip runs two objects per lane beat four at every size measured (10M: 2.25 / 2.5 ms, 50M: 68 / 72 ms)
        if (band && strip_items) {
            // band kernel: two neighbouring objects per lane at every density measured once a stage holds the whole
            // window (four per lane: 0.66 / 0.59 ms at the headline, 33 / 24 ms at 50M x 50M, 7.0 / 5.7 ms for RR of
            // config #4). Expected window = the tile's own extent in streamed entries + one band of
            // 2 r_win x (streamed objects of a run per unit of u).
            R = 2;
            auto per_u = [](const auto *c, const StripLayout *const *Ls) {
                int64_t runs = 1;
                for (int o = 0; o < 3; ++o)
                    if (Ls[o]) runs = std::max(runs, Ls[o]->h_vbase[(size_t)Ls[o]->n_groups]);
                double extent = 0.0;
                int n_ext = 0;
                for (int p = 0; p < c->n_patches; ++p) {
                    const double *b = &c->h_box[(size_t)6 * p];
                    double widest = 0.0;
                    for (int a = 0; a < 3; ++a) widest = std::max(widest, b[3 + a] - b[a]);
                    if (widest > 0.0) { extent += widest; ++n_ext; }
                }
                extent = n_ext ? extent / n_ext : 1.0;
                return ((double)c->n / (double)runs) / std::max(extent, 1e-6);
            };
            const double d1 = per_u(c_strm, LS), d2 = per_u(c_lane, LL);
            // binned x binned counts on per-(patch, bin) strip runs: runs are short (35 objects at 10 M, 350 at 100 M objects in
            // 30 bins), bands a handful of entries -- ONE object per lane then evaluates its own band instead of the union of
            // two (DD of config #4: 2.7e7 instead of 5.6e7 entries, 0.63 -> 0.41 ms; RR 3.74 -> 3.57), unless the streamed side
            // is much the sparser one and items are all fixed cost (DR: 2.05e6 items instead of 1.28e6, 1.86 -> 2.25 ms)
            if (mode == 3 && d1 >= 0.5 * d2) R = 1;
            est_window = 64.0 * R * d1 / std::max(d2, 1e-12) + 2.0 * rwin_max * d1;
        }
    }
    if (band && R == 0) R = 2;
    // Merged triple runs on the streamed side: one window per item instead of three (k_merge_triples), when the partner
    // strips are exactly c - 1, c, c + 1 (grid at least as wide as the largest separation) AND the merged window still goes
    // through the stage in one piece: cut in pieces it costs more than three whole windows (100M x 100M: 22.0 against 14.9 ms,
    // 50M x 50M with three scales 25.8 against 17.9). The fine-grid kernel has less room (a larger stage costs it residency:
    // 51 fine bins 1.25 against 1.06 ms in a 512-entry stage), so it merges only windows that fit the stage it uses anyway
    // (sparse streamed sides: DR of config #4 2.07 against 2.43). Weighted counts merge like unweighted ones since the kernel
    // with one chunk per round exists: the count kernel takes the same 0.48 ms at the headline in the big stage, the builder
    // searches one window per item instead of three (0.045 against 0.063 ms).
    bool triple = false;
    if ((band32 || band_fine) && strip_items && ctx->triple_runs && c_strm->n < (1ll << 31) && c1->strip_width > 0.0 &&
        (int)std::floor(rwin_max / c1->strip_width + 1e-6) + 1 == 1) {
        const double est3 = 3.0 * est_window;
        triple = ctx->triple_runs == 2 ||
                 (band32 ? est3 <= 0.88 * (B32_CAP_BIG - 4) : est3 <= 0.9 * BCAP_MID);
        for (int o = 0; o < 3 && triple; ++o) {
            if (!LS[o]) continue;
            const int rc = build_triples(ctx, const_cast<yawhip_catalog *>(c_strm), o, mode == 3);
            if (rc == YAWHIP_ERR_OOM) triple = false;  // no room for the copies: three windows per item as before
            else if (rc != YAWHIP_OK) return rc;
        }
        if (triple) est_window = est3;
    }
    if (g_trace.on) fprintf(stderr, "[yawhip trace] est_window %.1f (triple %d) R %d mode %d\n", est_window, (int)triple, R, mode);
    // stage capacity of the band kernel: the smallest compiled one that holds a whole window (see BCAP_MID)
    int cap = ctx->band_cap == BCAP || ctx->band_cap == BCAP_MID ? ctx->band_cap : 0;
    if (band && cap == 0) cap = R >= 4 || est_window > 0.95 * BCAP ? BCAP_MID : BCAP;
    if (band) {  // combinations that are compiled
        if (R == 1) cap = BCAP;
        if (R == 4 && cap == BCAP) cap = BCAP_MID;
    }
    int cap32 = ctx->band_cap == B32_CAP || ctx->band_cap == B32_CAP_BIG ? ctx->band_cap
                                                                         : (est_window > 0.75 * (B32_CAP - 4) ? B32_CAP_BIG : B32_CAP);
    // (0.75: window lengths scatter around the estimate, and a window cut in two costs more than a larger stage -- 50M x 50M
    // with windows of ~265 entries: 21.0 ms in the 320-entry stage, 18.3 ms in a 448-entry one)
    if (R == 1) cap32 = B32_CAP;      // combinations that are compiled
    if (R >= 4) cap32 = B32_CAP_BIG;
    if (band32) cap = cap32;  // (the fine-grid kernel still stages window by window, with the capacities of k_count_band)
    const int64_t tile = (int64_t)(lean ? MWG : WG) * R;
    const int lean_bins = merged ? n_bins : 1;
    const size_t lds_merged = 2 * MSTAGE * sizeof(ObjF) + (size_t)lean_bins * n_edges * sizeof(double) +
                              (size_t)lean_bins * nf * (weighted_any ? 8 * (MWG / 64) : 4) + (size_t)lean_bins * sizeof(float) +
                              (size_t)MWG * sizeof(unsigned int) + 16;
    // Copies of the LDS histogram, lanes spread over them by lane id: same-address atomics of one instruction are
    // serialised. Four copies when there are few slots and the bins of neighbouring entries are unrelated (headline:
    // 0.535 ms with four, 0.565 with eight -- the flush grows with the copies). When the histogram has only the fine bins
    // of ONE redshift bin (per-bin items: every hit of the wave lands in 1-3 cells), or when redshift follows position
    // (same_bin: neighbours of the layout's order sharing their bin; clustered survey: 108 -> 62 ms weighted cross count,
    // 31 -> 18 ms autocorrelation count), more copies pay: up to 16 within 2 KB.
    int hp_shift = lean_bins * nf <= 32 ? 2 : 0;
    if (band) {
        double coherence = merged ? 0.0 : 1.0;
        if (merged)
            for (int o = 0; o < 3; ++o)
                if (L1[o]) coherence = std::max(coherence, L1[o]->same_bin);
        if (coherence > 0.25) {
            const int cell = weighted_any ? 8 : 4;
            while (hp_shift < (merged ? 3 : 4) && ((size_t)lean_bins * nf * cell << (hp_shift + 1)) <= 2048) ++hp_shift;
        }
    }
    if (ctx->hist_copies_log2 >= 0) hp_shift = ctx->hist_copies_log2;
    const int band_ne = (!merged || uniform_t) && n_edges <= 4 ? n_edges : (nf == 1 ? 2 : 0);  // compile-time edge count of k_count_band
    const bool band_thr = !(band_ne >= 2 && (!merged || uniform_t));
    const size_t LDS_FIXED = (size_t)band_lds_fixed(cap);
    auto band_lds_for = [&](int shift) {
        return band_lds_dynamic(weighted_any, band_thr, lean_bins, n_edges, 1 << shift, cap, merged && !uniform_t ? lean_bins : 1);
    };
    while (band && hp_shift > 0 && band_lds_for(hp_shift) + LDS_FIXED > (size_t)ctx->lds_limit) --hp_shift;  // copies are a tunable, not a need
    const size_t lds_band = band_lds_for(hp_shift);
    if (lean && (band ? lds_band + LDS_FIXED : lds_merged) > (size_t)ctx->lds_limit)
        return fail(YAWHIP_ERR_INVALID, "too many bins x edges for the LDS histogram (%zu bytes)", band ? lds_band + LDS_FIXED : lds_merged);

    // item table: prefix[slot] = first item of the slot; items of a slot are its lane tiles.
    // standard path: slot = (job, bin); merged path: slot = job (one item covers all bins).
    // strip path: slot = job; its potential items = (lane tiles of patch q) x (groups of up to MAX_WIN of the 2*reach+1
    // neighbouring strips), enumerated by the builder kernel from the catalogues' run tables.
    std::vector<int64_t> prefix;
    std::vector<JobRec> job_recs;  // strip path, per job: what the builder needs of the two groups (JobRec)
    int64_t n_items = 0, cand = 0, abytes = 0;
    const int obj_bytes1 = c1->w ? 32 : 24, obj_bytes2 = c2->w ? 32 : 24;
    int reach = 0;
    const int tile_idx = R == 1 ? 0 : (R == 2 ? 1 : 2);
    // strip paths: the builder's job table. Modes 1/2: the jobs themselves (groups = patches); mode 3: one pseudo job
    // per (job, bin) between the segments (p, k) and (q, k) (groups = segments), numbered like the output slots.
    std::vector<int32_t> sjobs;
    const int64_t n_sjobs = mode == 3 ? n_slots : (int64_t)n_jobs;
    // Half bands: a catalogue counted against ITSELF meets every unordered pair of a diagonal job twice -- a as lane object with b
    // in its window, b as lane object with a in its. On merged triple runs with one object per lane the lane walks only the
    // entries BEHIND its own place in the triple of its strip (one total order of objects in all triples, k_merge_triples): every
    // pair is met once and counts twice (an exact doubling, also of weighted sums). Half the walk of DD / RR of an autocorrelation.
    const bool half_ok = band32 && triple && R == 1 && c1 == c2 && !swap && ctx->half_bands != 0 && !job_work;
    if (strip_items) {
        const double width = c1->strip_width;
        // |dv| <= rwin_max  ->  grid indices differ by at most floor(rwin_max / width) + 1
        reach = width > 0.0 ? (int)std::floor(rwin_max / width + 1e-6) + 1 : 0;
        sjobs.resize((size_t)2 * n_sjobs);
        for (int j = 0; j < n_jobs; ++j)
            for (int k = 0; k < (mode == 3 ? n_bins : 1); ++k) {
                const int64_t sj = mode == 3 ? (int64_t)j * n_bins + k : j;
                sjobs[(size_t)2 * sj] = mode == 3 ? jobs[2 * j] * n_bins + k : jobs[2 * j];
                sjobs[(size_t)2 * sj + 1] = mode == 3 ? jobs[2 * j + 1] * n_bins + k : jobs[2 * j + 1];
            }
        prefix.resize((size_t)n_sjobs + 1);
        job_recs.assign((size_t)n_sjobs, JobRec{0, 0, 0, 0, 0});
        for (int64_t j = 0; j < n_sjobs; ++j) {
            const int p = sjobs[(size_t)2 * j + (swap ? 1 : 0)], q = sjobs[(size_t)2 * j + (swap ? 0 : 1)];  // streamed, lane side
            const int o = orient[(size_t)(mode == 3 ? j / n_bins : j)];
            const StripLayout &sl1 = *LS[o], &sl2 = *LL[o];
            const std::vector<int64_t> &tiles = sl2.h_tiles[tile_idx];
            JobRec &jr = job_recs[(size_t)j];
            jr.o = o | (half_ok && p == q ? 4 : 0);
            prefix[(size_t)j] = n_items;
            // strips of q whose grid index lies within `reach` of the strips group p occupies
            const int64_t cnt1 = sl1.h_vbase[(size_t)p + 1] - sl1.h_vbase[(size_t)p], lo1 = sl1.h_slo[(size_t)p];
            const int64_t cnt2 = sl2.h_vbase[(size_t)q + 1] - sl2.h_vbase[(size_t)q], lo2 = sl2.h_slo[(size_t)q];
            const int64_t s_lo = std::max<int64_t>(lo1 - reach - lo2, 0), s_hi = std::min<int64_t>(lo1 + cnt1 - 1 + reach - lo2, cnt2 - 1);
            if (cnt1 > 0 && s_hi >= s_lo) {
                const int64_t r0 = sl2.h_vbase[(size_t)q] + s_lo;
                jr.t_lo = tiles[(size_t)r0];
                jr.k_off = lo2 - sl2.h_vbase[(size_t)q] - lo1;  // strip index of lane run r2 on the common grid, relative to group p
                jr.vbase1 = sl1.h_vbase[(size_t)p];
                jr.n_strips1 = (int32_t)cnt1;
                if (triple) {  // triple runs of group p: strips [lo1 - 1, lo1 + cnt1], the first one at vbase + 2 p
                    jr.k_off += 1;
                    jr.vbase1 += 2 * (int64_t)p;
                    jr.n_strips1 += 2;
                }
                n_items += (tiles[(size_t)(r0 + s_hi - s_lo + 1)] - tiles[(size_t)r0]) * ((2 * reach + 1 + MAX_WIN - 1) / MAX_WIN);
            }
        }
        prefix[(size_t)n_sjobs] = n_items;
    } else {
        prefix.resize((size_t)n_slots + 1);
    }
    const int64_t n_pslots = merged ? (int64_t)n_jobs : n_slots;
    auto patch_total = [](const yawhip_catalog *c, int patch) {  // objects of a patch over all its bins
        return c->h_off[(size_t)(patch + 1) * c->nb] - c->h_off[(size_t)patch * c->nb];
    };
    for (int j = 0; j < n_jobs; ++j) {
        const int p = jobs[2 * j], q = jobs[2 * j + 1];
        if (strip_items && (c1->nb == 1 || c2->nb == 1)) {
            // an unbinned side is one segment used for every bin: sum_k N1(p,k) N2(q,k) factorises
            cand += c1->nb == 1 ? patch_total(c1, p) * patch_total(c2, q) * (c2->nb == 1 ? n_bins : 1)
                                : patch_total(c1, p) * patch_total(c2, q);
        } else {
            for (int k = 0; k < n_bins; ++k) {
                const int64_t n1 = seg_len(c1, p, k), n2 = seg_len(c2, q, k);
                if (!strip_items) prefix[(size_t)j * n_bins + k] = n_items;
                if (n1 > 0 && n2 > 0) {
                    if (!strip_items) n_items += (n2 + tile - 1) / tile;
                    cand += n1 * n2;
                }
            }
        }
        // algorithmic bytes of a job = every object of the two patches once (Bobj * (N1 + N2))
        abytes += patch_total(c1, p) * obj_bytes1 + patch_total(c2, q) * obj_bytes2;
    }
    if (!strip_items) prefix[(size_t)n_pslots] = n_items;
    const int64_t slab = merged ? (int64_t)n_bins * nf : nf;  // float64 values per item of the weighted slab

    const bool run_weighted = weighted && want_sums;
    const bool run_unweighted = want_counts || (!weighted && want_sums);

    std::vector<float> dthr((size_t)3 * n_bins);  // per bin: pre-filter threshold, certain-band lower / upper bound
    auto round_down = [](double v) { float f = (float)v; if ((double)f > v) f = nextafterf(f, -4.0f); return f; };
    for (int k = 0; k < n_bins; ++k) {
        const double thi = t[(size_t)k * n_edges + n_edges - 1];
        float thr32 = ctx->debug_no_hits ? 2.0f : round_down(1.0 - 0.5 * thi - FILTER_GUARD);
        dthr[(size_t)3 * k] = thr32;
        dthr[(size_t)3 * k + 1] = 0.f;  // reserved
        dthr[(size_t)3 * k + 2] = 0.f;
    }
    std::vector<double> rwin((size_t)n_bins);
    for (int k = 0; k < n_bins; ++k) rwin[(size_t)k] = std::sqrt(t[(size_t)k * n_edges + n_edges - 1]) * (1.0 + 1e-12) + 1e-15;
    if (merged) rwin[0] = rwin_max;  // one window for all bins of the merged run
    // A weighted call keeps one slab of partial sums per potential item; long job lists of big catalogues would need
    // tens of GB (50M x 50M, three scales: 40 GB). Above the budget -- and when the items no longer fit 31 bits -- the
    // caller cuts the job list in two and counts the halves one after the other (rows of the result are independent).
    if (n_jobs > 1 && !job_work &&
        ((run_weighted && n_items * slab * (int64_t)sizeof(double) > ctx->slab_budget) || n_items >= (1ll << 31)))
        { P.split = true; return YAWHIP_OK; }
    // layout table of the call: [o] = c1, [3 + o] = c2 for orientation o (plain layouts: entries 0 and 3)
    DevTab h_tabs[6];
    memset(h_tabs, 0, sizeof h_tabs);
    if (strip_items) {
        for (int o = 0; o < 3; ++o) {
            if (!L1[o]) continue;
            const StripLayout &a = *L1[o], &b = *L2[o];
            h_tabs[o] = make_tab(a.x, a.y, a.z, a.w, merged ? a.k : nullptr, a.off, a.d_vbase, a.d_slo, a.d_tiles[tile_idx],
                                 a.d_tile_rec[tile_idx], a.d_grid, o, a.q, a.q_stride);
            h_tabs[3 + o] = make_tab(b.x, b.y, b.z, b.w, nullptr, b.off, b.d_vbase, b.d_slo, b.d_tiles[tile_idx],
                                     b.d_tile_rec[tile_idx], b.d_grid, o, b.q, b.q_stride);
            if (triple) {
                // the streamed side as merged triple runs: images, weights, offsets and grid index of the triples; the float64
                // columns stay the layout's own (reached through idx by the exact re-evaluation)
                const StripLayout &st = swap ? b : a;
                DevTab &tb = h_tabs[swap ? 3 + o : o];
                tb = make_tab(st.x, st.y, st.z, st.w3, nullptr, st.off3, st.d_vbase, st.d_slo, st.d_tiles[tile_idx],
                              st.d_tile_rec[tile_idx], st.d_grid3, o, st.q3, st.q3_stride, st.idx3);
                if (half_ok) h_tabs[3 + o].pos3 = (gi32p)b.pos3;  // (c1 == c2: the lane side's layout is the streamed one)
            }
        }
    } else {
        h_tabs[0] = make_tab(c1->x, c1->y, c1->z, c1->w, nullptr, c1->off, nullptr, nullptr, nullptr, nullptr, nullptr, c1->axis);
        h_tabs[3] = make_tab(c2->x, c2->y, c2->z, c2->w, nullptr, c2->off, nullptr, nullptr, nullptr, nullptr, nullptr, c2->axis);
    }
    // the tables of the call, packed into the pinned staging buffer and sent with one copy
    static_assert(sizeof(JobRec) == 8 * sizeof(int32_t), "JobRec is 32 bytes");
    const size_t n_jobtab = strip_items ? (size_t)8 * n_sjobs : (size_t)2 * n_jobs;  // JobRec per job, or (p, q) pairs
    size_t off_in = 0;
    auto take = [&](size_t bytes) { const size_t o = off_in; off_in = align16(off_in + bytes); return o; };
    const size_t o_jobs = take(n_jobtab * sizeof(int32_t));
    const size_t o_prefix = take(((size_t)n_pslots + 1) * sizeof(int64_t));
    const size_t o_t = take((size_t)n_bins * n_edges * sizeof(double));
    const size_t o_dthr = take((size_t)3 * n_bins * sizeof(float));
    const size_t o_rwin = take((size_t)n_bins * sizeof(double));
    const std::vector<float> thr32 = band32 ? build_thr32(t, n_bins, n_edges) : (band_fine ? fine32 : std::vector<float>());
    const size_t o_thr32 = take(thr32.size() * sizeof(float));
    const size_t o_tabs = take(sizeof h_tabs);
    // weighted calls: the two-level ordered reduction of the slabs needs the first chunk of every output slot
    const int64_t n_oslots = !lean ? n_slots : (merged ? (int64_t)n_jobs : n_slots);
    std::vector<int64_t> cprefix;
    if (run_weighted) {
        cprefix.assign((size_t)n_oslots + 1, 0);
        for (int64_t sl = 0; sl < n_oslots; ++sl)
            cprefix[(size_t)sl + 1] = cprefix[(size_t)sl] + (prefix[(size_t)sl + 1] - prefix[(size_t)sl] + REDUCE_CHUNK - 1) / REDUCE_CHUNK;
    }
    const size_t o_cprefix = take(cprefix.size() * sizeof(int64_t));
    std::vector<unsigned char> image(off_in, 0);
    if (strip_items) {
        memcpy(image.data() + o_jobs, job_recs.data(), sizeof(JobRec) * n_sjobs);
    } else {
        memcpy(image.data() + o_jobs, jobs, sizeof(int32_t) * 2 * n_jobs);
    }
    memcpy(image.data() + o_prefix, prefix.data(), sizeof(int64_t) * ((size_t)n_pslots + 1));
    memcpy(image.data() + o_t, t, sizeof(double) * n_bins * n_edges);
    memcpy(image.data() + o_dthr, dthr.data(), sizeof(float) * 3 * n_bins);
    memcpy(image.data() + o_rwin, rwin.data(), sizeof(double) * n_bins);
    if (!thr32.empty()) memcpy(image.data() + o_thr32, thr32.data(), sizeof(float) * thr32.size());
    memcpy(image.data() + o_tabs, h_tabs, sizeof h_tabs);
    if (!cprefix.empty()) memcpy(image.data() + o_cprefix, cprefix.data(), sizeof(int64_t) * cprefix.size());
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&P.d_in), std::max<size_t>(off_in, 16)));
    HIP_TRY(hipMemcpy(P.d_in, image.data(), off_in, hipMemcpyHostToDevice));  // once per plan
    P.o_jobs = o_jobs; P.o_prefix = o_prefix; P.o_t = o_t; P.o_dthr = o_dthr; P.o_rwin = o_rwin; P.o_thr32 = o_thr32;
    P.o_tabs = o_tabs; P.o_cprefix = o_cprefix;
    P.n_chunks = cprefix.empty() ? 0 : cprefix.back();
    P.n_oslots = n_oslots;
    P.n_orient = (L1[0] ? 1 : 0) + (L1[1] ? 1 : 0) + (L1[2] ? 1 : 0);
    // Workgroups of the band kernels = potential items / grid_div (the kernel loops over the rest). Uniform catalogues, whose
    // items are alike, run best with few, longer-lived workgroups: / 8, / 16 for the per-bin items of binned x binned counts, of
    // which the builder keeps a third (headline 4 / 8 / 16 -> 0.277 / 0.275 / 0.292 ms; config #4 DR 1.59 / 1.53 / 1.48, RR 3.08 /
    // 3.04 / 3.02). On CLUSTERED catalogues items differ a hundredfold and the hardware's dispatch of many short workgroups is the
    // load balancer: / 4 (clustered survey, 3M x 4M: cross count 40.1 against 43.3 ms with / 8, autocorrelation count 7.1
    // against 8.2 with / 16). Clustered = the run the typical OBJECT sits in (sum len^2 / sum len) is more than twice the mean run.
    {
        double skew = 1.0;
        for (const StripLayout *const *LX : {L1, L2})
            for (int o = 0; o < 3; ++o)
                if (LX[o] && LX[o]->built) {
                    const yawhip_catalog *cx = LX == L1 ? c1 : c2;
                    const double runs = (double)std::max<int64_t>(LX[o]->h_vbase[(size_t)LX[o]->n_groups], 1);
                    skew = std::max(skew, LX[o]->obj_run / std::max((double)cx->n / runs, 1.0));
                }
        P.grid_div = ctx->band_grid_div > 0 ? ctx->band_grid_div : (skew > 2.0 ? 4 : (mode == 3 ? 16 : 8));
        if (g_trace.on) fprintf(stderr, "[yawhip trace] run skew %.2f -> grid / %d\n", skew, P.grid_div);
    }
    P.R = R;
    P.abytes = abytes;
    P.band = band;
    P.band32 = band32;
    P.band_fine = band_fine;
    P.band_ne = band_ne;
    P.cand = cand;
    P.cap = cap;
    P.filter = filter;
    P.hp_shift = hp_shift;
    P.lds_band = lds_band;
    P.lds_merged = lds_merged;
    P.lean = lean;
    P.lean_bins = lean_bins;
    P.merged = merged;
    P.mode = mode;
    P.n_items = n_items;
    P.n_out = n_out;
    P.n_pslots = n_pslots;
    P.n_sjobs = n_sjobs;
    P.n_slots = n_slots;
    P.nf = nf;
    P.reach = reach;
    P.run_unweighted = run_unweighted;
    P.run_weighted = run_weighted;
    P.rwin_max = rwin_max;
    P.slab = slab;
    P.strip_items = strip_items;
    P.swap = swap;
    P.sweep = sweep;
    P.tile = tile;
    P.triple = triple;
    P.uniform_t = uniform_t;
    P.weighted = weighted;
    P.weighted_any = weighted_any;
    P.kernel = kernel;
    g_trace.mark("planned");
    return YAWHIP_OK;
}

// First half of yawhip_count_pairs on ONE device: everything up to and including the copy of the results into the
// context's pinned buffer is put on the context's stream; nothing waits for the device (SWEEP's grid sizing aside).
// The host side of it (make_plan) is done once per distinct set of inputs and looked up afterwards.
// job_work != nullptr: cost estimate only -- the item builder runs, evaluated pairs per job are returned, no counting.
int count_enqueue(yawhip_ctx *ctx, const yawhip_catalog *c1, const yawhip_catalog *c2, int32_t n_jobs,
                  const int32_t *jobs, int32_t n_bins, int32_t n_edges, const double *t, int32_t kernel,
                  bool want_counts, bool want_sums, int64_t *job_work, CallState &cs, bool fetch_results = true) {
    cs = CallState{};
    cs.wall0 = std::chrono::steady_clock::now();
    g_trace.mark("enqueue");
    cs.want_counts = want_counts;
    cs.want_sums = want_sums;
    if (!ctx || !c1 || !c2) return fail(YAWHIP_ERR_INVALID, "yawhip_count_pairs: NULL handle");
    if (n_jobs < 0 || n_bins <= 0 || n_edges < 2 || n_edges > MAX_EDGES || !t || (n_jobs > 0 && !jobs))
        return fail(YAWHIP_ERR_INVALID, "yawhip_count_pairs: bad sizes (n_jobs=%d n_bins=%d n_edges=%d, max edges %d)",
                    n_jobs, n_bins, n_edges, MAX_EDGES);
    HIP_TRY(hipSetDevice(ctx->device));
    // the plan of these inputs: FNV-1a over everything it depends on, then an exact comparison of job list and thresholds
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *ptr, size_t n) {
        const unsigned char *bytes = static_cast<const unsigned char *>(ptr);
        for (size_t i = 0; i < n; ++i) { h ^= bytes[i]; h *= 1099511628211ull; }
    };
    const int32_t head_key[6] = {n_jobs, n_bins, n_edges, kernel, (want_counts ? 1 : 0) | (want_sums ? 2 : 0), job_work ? 1 : 0};
    mix(head_key, sizeof head_key);
    mix(&c1->uid, sizeof c1->uid); mix(&c2->uid, sizeof c2->uid); mix(&ctx->opt_gen, sizeof ctx->opt_gen);
    mix(jobs, sizeof(int32_t) * 2 * (size_t)n_jobs);
    mix(t, sizeof(double) * (size_t)n_bins * n_edges);
    HostPlan *plan = nullptr;
    for (HostPlan *cand_plan : ctx->plans)
        if (cand_plan->hash == h && cand_plan->c1_uid == c1->uid && cand_plan->c2_uid == c2->uid && cand_plan->opt_gen == ctx->opt_gen &&
            cand_plan->n_jobs_in == n_jobs && cand_plan->n_bins_in == n_bins && cand_plan->n_edges_in == n_edges &&
            cand_plan->kernel_in == kernel && cand_plan->want_counts == want_counts && cand_plan->want_sums == want_sums &&
            cand_plan->for_work == (job_work != nullptr) &&
            memcmp(cand_plan->jobs_in.data(), jobs, sizeof(int32_t) * 2 * (size_t)n_jobs) == 0 &&
            memcmp(cand_plan->t_in.data(), t, sizeof(double) * (size_t)n_bins * n_edges) == 0) {
            plan = cand_plan;
            break;
        }
    if (!plan) {
        std::unique_ptr<HostPlan> fresh(new (std::nothrow) HostPlan());
        if (!fresh) return fail(YAWHIP_ERR_OOM, "host allocation failed");
        const int rc = make_plan(ctx, c1, c2, n_jobs, jobs, n_bins, n_edges, t, kernel, want_counts, want_sums, job_work != nullptr, *fresh);
        if (rc != YAWHIP_OK) return rc;
        fresh->hash = h; fresh->c1_uid = c1->uid; fresh->c2_uid = c2->uid; fresh->opt_gen = ctx->opt_gen;
        fresh->n_jobs_in = n_jobs; fresh->n_bins_in = n_bins; fresh->n_edges_in = n_edges; fresh->kernel_in = kernel;
        fresh->want_counts = want_counts; fresh->want_sums = want_sums; fresh->for_work = job_work != nullptr;
        fresh->jobs_in.assign(jobs, jobs + 2 * (size_t)n_jobs);
        fresh->t_in.assign(t, t + (size_t)n_bins * n_edges);
        if (ctx->plans.size() >= MAX_PLANS) {  // evict the least recently used one (nothing of it is in flight: calls are blocking,
            size_t old = 0;                    // and a batch is never longer than the plans kept)
            for (size_t i = 1; i < ctx->plans.size(); ++i)
                if (ctx->plans[i]->stamp < ctx->plans[old]->stamp) old = i;
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            delete ctx->plans[old];
            ctx->plans[old] = ctx->plans.back();
            ctx->plans.pop_back();
        }
        plan = fresh.release();
        ctx->plans.push_back(plan);
    }
    plan->stamp = ++ctx->plan_clock;
    const HostPlan &P = *plan;
    cs.n_out = P.n_out;
    if (P.empty) return YAWHIP_OK;
    if (P.split) return SPLIT_JOBS;
    if (P.run_weighted) HIP_TRY(ctx->d_partials.reserve((size_t)std::max<int64_t>(P.n_items, 1) * P.slab));
    ctx->d_jobs.ptr = reinterpret_cast<int32_t *>(P.d_in + P.o_jobs);
    ctx->d_prefix.ptr = reinterpret_cast<int64_t *>(P.d_in + P.o_prefix);
    ctx->d_t.ptr = reinterpret_cast<double *>(P.d_in + P.o_t);
    ctx->d_dthr.ptr = reinterpret_cast<float *>(P.d_in + P.o_dthr);
    ctx->d_rwin.ptr = reinterpret_cast<double *>(P.d_in + P.o_rwin);
    ctx->d_thr32.ptr = reinterpret_cast<float *>(P.d_in + P.o_thr32);
    ctx->d_tabs.ptr = reinterpret_cast<DevTab *>(P.d_in + P.o_tabs);
    ctx->d_cprefix.ptr = reinterpret_cast<int64_t *>(P.d_in + P.o_cprefix);
    g_trace.mark("plan");
    // results: [counters][counts][sums] in one device buffer, zeroed by one memset (sums are always fully written) and
    // fetched by one copy
    const size_t o_ctr = 0, o_counts = align16(N_CTR * sizeof(unsigned long long)),
                 o_sums = o_counts + align16((size_t)P.n_out * sizeof(unsigned long long));
    const size_t out_bytes = o_sums + align16((size_t)P.n_out * sizeof(double));
    HIP_TRY(ctx->out.reserve(out_bytes));
    ctx->d_ctr.ptr = reinterpret_cast<unsigned long long *>(ctx->out.d + o_ctr);
    ctx->d_counts.ptr = reinterpret_cast<unsigned long long *>(ctx->out.d + o_counts);
    ctx->d_sums.ptr = reinterpret_cast<double *>(ctx->out.d + o_sums);
    HIP_TRY(hipMemsetAsync(ctx->out.d, 0, P.n_items > 0 ? o_sums : out_bytes, ctx->stream));

    // LDS: two stages + thresholds + histogram(s)
    const size_t lds_fixed = 2 * STAGE * (sizeof(Obj) + sizeof(ObjF)) + (size_t)((n_edges + 1) & ~1) * sizeof(double);
    auto lds_for = [&](bool w, bool priv) { return lds_fixed + (size_t)P.nf * (priv ? WG : 1) * (w ? 8 : 4); };
    int launches = 0;
    const int64_t n_pot = P.n_items;
    int64_t n_items = P.n_items;  // the count grid: all potential items, or what the builder kept (SWEEP)
    unsigned long long seg_cap = 0;  // > 0: the item list is kept in ITEM_SEGS segments of this many records
    g_trace.mark("memset");
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    if (n_pot > 0) {
        if (n_pot >= (1ll << 31))
            return fail(YAWHIP_ERR_INVALID, "too many work items (%lld) in one job", (long long)n_pot);
        const int bwg = build_wg_for(n_pot);
        const unsigned bgrid = (unsigned)((n_pot + bwg - 1) / bwg);
        // item list in segments (append_items): where the float32 band kernels consume what the strip builder keeps
        if (P.strip_items && (P.band32 || P.band_fine) && !job_work && ctx->item_segments)
            seg_cap = (unsigned long long)((bgrid + ITEM_SEGS - 1) / ITEM_SEGS) * (unsigned long long)bwg;
        HIP_TRY(ctx->d_items.reserve(seg_cap ? (size_t)(seg_cap * ITEM_SEGS) : (size_t)n_pot));
        unsigned char *kept_flags = nullptr;  // weighted runs of the culling builders: which potential items write a slab
        if (P.run_weighted && P.sweep) {
            HIP_TRY(ctx->d_kept.reserve((size_t)n_pot));
            HIP_TRY(hipMemsetAsync(ctx->d_kept.ptr, 0, (size_t)n_pot, ctx->stream));
            kept_flags = ctx->d_kept.ptr;
        }
        if (P.strip_items)
            hipLaunchKernelGGL(k_build_items_strips, dim3(bgrid), dim3(bwg), 0, ctx->stream, ctx->d_tabs.ptr,
                               reinterpret_cast<const JobRec *>(ctx->d_jobs.ptr), ctx->d_prefix.ptr, (int)P.n_sjobs,
                               P.triple ? 0 : P.reach, (int)P.tile, P.rwin_max, P.swap ? 1 : 0, P.triple ? 1 : 0, n_pot, ctx->d_items.ptr,
                               ctx->d_ctr.ptr, kept_flags, seg_cap);
        else if (P.sweep)
            hipLaunchKernelGGL(k_build_items<true>, dim3(bgrid), dim3(bwg), 0, ctx->stream, view_of(c1), view_of(c2),
                               ctx->d_jobs.ptr, ctx->d_prefix.ptr, (int)P.n_pslots, n_bins, (int)P.tile,
                               ctx->d_rwin.ptr, n_pot, ctx->d_items.ptr, ctx->d_ctr.ptr, kept_flags);
        else
            hipLaunchKernelGGL(k_build_items<false>, dim3(bgrid), dim3(bwg), 0, ctx->stream, view_of(c1), view_of(c2),
                               ctx->d_jobs.ptr, ctx->d_prefix.ptr, (int)P.n_pslots, n_bins, (int)P.tile, ctx->d_rwin.ptr, n_pot,
                               ctx->d_items.ptr, ctx->d_ctr.ptr, nullptr);
        HIP_TRY(hipGetLastError());
        ++launches;
        // The count kernels are launched over all potential items and return at once for indices beyond the
        // number the builder kept (device counter): no host round trip between the two kernels.
        n_items = n_pot;
        if (P.strip_items && !P.band && n_pot > SYNC_GRID_MIN_ITEMS) {
            // SWEEP: the strip path keeps about one potential item in five; a grid over all of them spends ~0.2 ms
            // dispatching workgroups that exit at once (measured at 1.6e6 potential items, 10M x 10M), more than
            // this round trip (~0.05 ms) costs. Small calls (one GPU's share of a sharded job list) skip it.
            // (The band kernel sizes its grid from the potential items and loops: no round trip.)
            HIP_TRY(hipMemcpyAsync(ctx->out.h, ctx->d_ctr.ptr, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            n_items = (int64_t)reinterpret_cast<unsigned long long *>(ctx->out.h)[0];
        }
    }
    if (job_work) {  // cost estimate only: evaluated pairs per job from the item list, no counting
        HIP_TRY(ctx->d_jobwork.reserve((size_t)n_jobs));
        HIP_TRY(hipMemsetAsync(ctx->d_jobwork.ptr, 0, sizeof(unsigned long long) * (size_t)n_jobs, ctx->stream));
        if (n_pot > 0) {
            hipLaunchKernelGGL(k_item_work, dim3((unsigned)((n_pot + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_items.ptr,
                               ctx->d_ctr.ptr, P.merged ? 1 : n_bins, ctx->d_jobwork.ptr);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipMemcpyAsync(job_work, ctx->d_jobwork.ptr, sizeof(int64_t) * (size_t)n_jobs, hipMemcpyDeviceToHost,
                               ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return YAWHIP_OK;
    }
    // two-level ordered reduction of the weighted slabs (k_reduce_chunks / k_reduce_slots); prefix = first potential
    // item of every output slot
    auto reduce_partials = [&](int64_t n_oslots, int64_t values) -> hipError_t {
        const int64_t n_chunks = P.n_chunks;  // (chunk prefix: in the plan's device tables, ctx->d_cprefix)
        if (n_oslots != P.n_oslots) return hipErrorInvalidValue;
        hipError_t er = ctx->d_chunk_sums.reserve((size_t)std::max<int64_t>(n_chunks, 1) * values);
        if (er != hipSuccess) return er;
        const int thr = 256;
        const bool all_kept = !(P.run_weighted && P.sweep);
        if (n_chunks > 0)
            hipLaunchKernelGGL(k_reduce_chunks, dim3((unsigned)((n_chunks * values + thr - 1) / thr)), dim3(thr), 0, ctx->stream,
                               ctx->d_partials.ptr, all_kept ? nullptr : ctx->d_kept.ptr, ctx->d_prefix.ptr, ctx->d_cprefix.ptr,
                               (int)n_oslots, (int)values, ctx->d_chunk_sums.ptr);
        hipLaunchKernelGGL(k_reduce_slots, dim3((unsigned)((n_oslots * values + thr - 1) / thr)), dim3(thr), 0, ctx->stream,
                           ctx->d_chunk_sums.ptr, ctx->d_cprefix.ptr, (int)n_oslots, (int)values, ctx->d_sums.ptr);
        return hipGetLastError();
    };
    HIP_TRY(hipEventRecord(ctx->evc0, ctx->stream));
    bool band_ran = false;
    if (n_items > 0 && P.lean && P.band) {
        // Grid from the number of POTENTIAL items (known on the host); the kernel reads the number the builder kept
        // from the device counter, workgroups beyond it exit, workgroups loop if more were kept than the grid holds.
        // The strip builder keeps about one potential item in five, ordinary items are all kept.
        const int grid_div = P.grid_div;
        int64_t grid = P.strip_items && n_pot > 65536 ? n_pot / grid_div : n_pot;
        // items per workgroup visit (unweighted): batches of 4 / 8 when the histogram has hundreds of cells to flush
        const int n_cells = P.lean_bins * P.nf;
        // Batches of consecutive items (one flush of the histogram per batch) are a tunable, off by default: consecutive
        // items are tiles of the same run, so on clustered data a batch strings the heaviest items together on one
        // workgroup (measured: DD of the clustered survey with 31 fine bins 7.2 -> 20.6 ms with batches of four), and on
        // uniform data the flush they save is not what the time goes to (2.27 ms either way at the headline, 51 fine bins).
        (void)n_cells;
        const int batch_log2 = ctx->band_batch_log2 >= 0 ? ctx->band_batch_log2 : 0;
        if (!P.run_weighted) grid = std::max<int64_t>(grid >> batch_log2, 8);
        grid = std::min<int64_t>((grid + 7) & ~7ll, 1ll << 22);
        // 32-bit LDS counters: one stage adds at most 64 R x CAP to a cell, so flush at the latest every
        // 2^32 / (64 R CAP) stages (2^17 for two objects per lane and 192-entry stages, 2^15 for four and 288)
        int flush_log2 = ctx->flush_log2;
        while (flush_log2 > 0 && ((uint64_t)64 * P.R * 2 * P.cap << flush_log2) >= (1ull << 32)) --flush_log2;  // (x 2: half bands count double)
        const unsigned flush_mask = (1u << flush_log2) - 1u;
        const size_t lds_band32 = band32_lds(P.weighted_any, P.cap, P.lean_bins * P.nf, P.merged && !P.uniform_t ? n_bins : 0, n_edges);
        const bool one_chunk = P.triple || !P.strip_items;  // every item has one window
        auto launch_band32 = [&](bool wgt) -> hipError_t {
#define YAW_LAUNCH_B32_CH(RR, CC, WW, NN, MM, UU, KNAME)                                                              \
    do {                                                                                                              \
        auto kern = KNAME<RR, CC, WW, NN, MM, UU>;                                                                    \
        if (lds_band32 > 64 * 1024) {                                                                                 \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                 \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_band32);         \
            if (ea != hipSuccess) return ea;                                                                          \
        }                                                                                                             \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds_band32, ctx->stream, ctx->d_tabs.ptr, ctx->d_items.ptr, \
                           n_bins, ctx->d_t.ptr, ctx->d_thr32.ptr, ctx->d_rwin.ptr, flush_mask, P.swap ? 1 : 0, ctx->d_counts.ptr, \
                           ctx->d_partials.ptr, ctx->d_ctr.ptr, seg_cap);                                             \
    } while (0)
#define YAW_LAUNCH_B32(RR, CC, WW, NN, MM, UU)                                                                        \
    do {                                                                                                              \
        if (one_chunk) YAW_LAUNCH_B32_CH(RR, CC, WW, NN, MM, UU, k_count_band32_one);                                 \
        else YAW_LAUNCH_B32_CH(RR, CC, WW, NN, MM, UU, k_count_band32);                                               \
    } while (0)
#define YAW_LAUNCH_B32_R(WW, NN, MM, UU)                                                                              \
    do {                                                                                                              \
        if (P.R == 1) YAW_LAUNCH_B32(1, B32_CAP, WW, NN, MM, UU);                                                       \
        else if (P.R == 2 && P.cap == B32_CAP) YAW_LAUNCH_B32(2, B32_CAP, WW, NN, MM, UU);                                \
        else if (P.R == 2) YAW_LAUNCH_B32(2, B32_CAP_BIG, WW, NN, MM, UU);                                              \
        else YAW_LAUNCH_B32(4, B32_CAP_BIG, WW, NN, MM, UU);                                                          \
    } while (0)
#define YAW_LAUNCH_B32_M(WW, NN)                                                                                      \
    do {                                                                                                              \
        if (!P.merged) YAW_LAUNCH_B32_R(WW, NN, false, true);                                                           \
        else if (P.uniform_t) YAW_LAUNCH_B32_R(WW, NN, true, true);                                                     \
        else YAW_LAUNCH_B32_R(WW, NN, true, false);                                                                   \
    } while (0)
#define YAW_LAUNCH_B32_N(WW)                                                                                          \
    do {                                                                                                              \
        if (n_edges == 2) YAW_LAUNCH_B32_M(WW, 2); else if (n_edges == 3) YAW_LAUNCH_B32_M(WW, 3);                    \
        else YAW_LAUNCH_B32_M(WW, 4);                                                                                 \
    } while (0)
            if (wgt) YAW_LAUNCH_B32_N(true); else YAW_LAUNCH_B32_N(false);
#undef YAW_LAUNCH_B32_N
#undef YAW_LAUNCH_B32_M
#undef YAW_LAUNCH_B32_R
#undef YAW_LAUNCH_B32
#undef YAW_LAUNCH_B32_CH
            return hipGetLastError();
        };
        auto launch_band64 = [&](bool wgt) -> hipError_t {
#define YAW_LAUNCH_BAND(RR, CC, WW, NN, MM, UU)                                                                       \
    do {                                                                                                              \
        auto kern = k_count_band<RR, CC, WW, NN, MM, UU>;                                                             \
        if (P.lds_band > 64 * 1024) {                                                                                   \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                 \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.lds_band);           \
            if (ea != hipSuccess) return ea;                                                                          \
        }                                                                                                             \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), P.lds_band, ctx->stream, ctx->d_tabs.ptr, ctx->d_items.ptr, \
                           n_bins, n_edges, ctx->d_t.ptr, ctx->d_rwin.ptr, flush_mask, P.hp_shift, WW ? 0 : batch_log2, ctx->d_counts.ptr, \
                           ctx->d_partials.ptr, ctx->d_ctr.ptr);                                                      \
    } while (0)
#define YAW_LAUNCH_BAND_R(WW, NN, MM, UU)                                                                             \
    do {                                                                                                              \
        if (P.R == 1) YAW_LAUNCH_BAND(1, BCAP, WW, NN, MM, UU);                                                         \
        else if (P.R == 2 && P.cap == BCAP) YAW_LAUNCH_BAND(2, BCAP, WW, NN, MM, UU);                                     \
        else if (P.R == 2) YAW_LAUNCH_BAND(2, BCAP_MID, WW, NN, MM, UU);                                                \
        else YAW_LAUNCH_BAND(4, BCAP_MID, WW, NN, MM, UU);                                                            \
    } while (0)
#define YAW_LAUNCH_BAND_M(WW, NN)                                                                                     \
    do {                                                                                                              \
        if (!P.merged) YAW_LAUNCH_BAND_R(WW, NN, false, true);                                                          \
        else if (P.uniform_t) YAW_LAUNCH_BAND_R(WW, NN, true, true);                                                    \
        else YAW_LAUNCH_BAND_R(WW, NN, true, false);                                                                  \
    } while (0)
#define YAW_LAUNCH_BAND_N(WW)                                                                                         \
    do {                                                                                                              \
        if (P.band_ne == 2) YAW_LAUNCH_BAND_M(WW, 2); else if (P.band_ne == 3) YAW_LAUNCH_BAND_M(WW, 3);                  \
        else if (P.band_ne == 4) YAW_LAUNCH_BAND_M(WW, 4); else YAW_LAUNCH_BAND_M(WW, 0);                               \
    } while (0)
            if (wgt) YAW_LAUNCH_BAND_N(true); else YAW_LAUNCH_BAND_N(false);
#undef YAW_LAUNCH_BAND_N
#undef YAW_LAUNCH_BAND_M
#undef YAW_LAUNCH_BAND_R
#undef YAW_LAUNCH_BAND
            return hipGetLastError();
        };
        const size_t lds_fine = band32_fine_lds(P.weighted_any, P.cap, P.lean_bins * P.nf, P.uniform_t ? 1 : n_bins, n_edges);
        auto launch_fine = [&](bool wgt) -> hipError_t {
#define YAW_LAUNCH_FINE(RR, CC, WW, MM, UU)                                                                           \
    do {                                                                                                              \
        auto kern = k_count_band32_fine<RR, CC, WW, MM, UU>;                                                          \
        if (lds_fine > 64 * 1024) {                                                                                   \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                 \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fine);           \
            if (ea != hipSuccess) return ea;                                                                          \
        }                                                                                                             \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds_fine, ctx->stream, ctx->d_tabs.ptr, ctx->d_items.ptr, \
                           n_bins, n_edges, ctx->d_t.ptr, ctx->d_thr32.ptr, ctx->d_rwin.ptr, flush_mask, P.swap ? 1 : 0, ctx->d_counts.ptr, \
                           ctx->d_partials.ptr, ctx->d_ctr.ptr, seg_cap);                                             \
    } while (0)
#define YAW_LAUNCH_FINE_R(WW, MM, UU)                                                                                 \
    do {                                                                                                              \
        if (P.R == 1) YAW_LAUNCH_FINE(1, BCAP, WW, MM, UU);                                                             \
        else if (P.R == 2 && P.cap == BCAP) YAW_LAUNCH_FINE(2, BCAP, WW, MM, UU);                                         \
        else if (P.R == 2) YAW_LAUNCH_FINE(2, BCAP_MID, WW, MM, UU);                                                    \
        else YAW_LAUNCH_FINE(4, BCAP_MID, WW, MM, UU);                                                                \
    } while (0)
#define YAW_LAUNCH_FINE_M(WW)                                                                                         \
    do {                                                                                                              \
        if (!P.merged && P.uniform_t) YAW_LAUNCH_FINE_R(WW, false, true);                                                 \
        else if (!P.merged) YAW_LAUNCH_FINE_R(WW, false, false);                                                        \
        else if (P.uniform_t) YAW_LAUNCH_FINE_R(WW, true, true);                                                        \
        else YAW_LAUNCH_FINE_R(WW, true, false);                                                                      \
    } while (0)
            if (wgt) YAW_LAUNCH_FINE_M(true); else YAW_LAUNCH_FINE_M(false);
#undef YAW_LAUNCH_FINE_M
#undef YAW_LAUNCH_FINE_R
#undef YAW_LAUNCH_FINE
            return hipGetLastError();
        };
        auto launch_band = [&](bool wgt) -> hipError_t {
            return P.band32 ? launch_band32(wgt) : (P.band_fine ? launch_fine(wgt) : launch_band64(wgt));
        };
        if (P.run_unweighted) {
            HIP_TRY(launch_band(false));
            ++launches;
        }
        if (P.run_weighted) {
            HIP_TRY(launch_band(true));
            ++launches;
            HIP_TRY(reduce_partials(P.merged ? (int64_t)n_jobs : P.n_slots, P.slab));  // slabs are reduced per output slot
            launches += 2;
        }
        band_ran = true;
    } else if (n_items > 0 && P.lean) {
        auto launch_lean = [&](bool wgt) -> hipError_t {
            const int64_t max_grid = (1ll << 31) / MWG;  // at most 2^32 - 1 work-items per launch dimension
            for (int64_t base = 0; base < n_items; base += max_grid) {
                const unsigned g = (unsigned)std::min(max_grid, n_items - base);
#define YAW_LAUNCH_LEAN(RR, WW, NN, MM)                                                                               \
    do {                                                                                                              \
        auto kern = pick_count_merged<RR, WW, NN, MM>();                                                              \
        if (P.lds_merged > 64 * 1024) {                                                                                 \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                 \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.lds_merged);         \
            if (ea != hipSuccess) return ea;                                                                          \
        }                                                                                                             \
        hipLaunchKernelGGL(kern, dim3(g), dim3(MWG), P.lds_merged, ctx->stream, ctx->d_tabs.ptr, ctx->d_items.ptr,      \
                           n_bins, n_edges, ctx->d_t.ptr, ctx->d_dthr.ptr, ctx->d_rwin.ptr, base, ctx->d_counts.ptr,  \
                           ctx->d_partials.ptr, ctx->d_ctr.ptr);                                                      \
    } while (0)
#define YAW_LAUNCH_LEAN_R(WW, NN, MM)                                                                                 \
    do {                                                                                                              \
        if (P.R == 1) YAW_LAUNCH_LEAN(1, WW, NN, MM); else if (P.R == 2) YAW_LAUNCH_LEAN(2, WW, NN, MM); else YAW_LAUNCH_LEAN(4, WW, NN, MM); \
    } while (0)
#define YAW_LAUNCH_LEAN_M(WW, NN)                                                                                     \
    do {                                                                                                              \
        if (P.merged) YAW_LAUNCH_LEAN_R(WW, NN, true); else YAW_LAUNCH_LEAN_R(WW, NN, false);                           \
    } while (0)
                const bool nf1 = P.nf == 1;
                if (wgt) {
                    if (nf1) YAW_LAUNCH_LEAN_M(true, true); else YAW_LAUNCH_LEAN_M(true, false);
                } else {
                    if (nf1) YAW_LAUNCH_LEAN_M(false, true); else YAW_LAUNCH_LEAN_M(false, false);
                }
#undef YAW_LAUNCH_LEAN_M
#undef YAW_LAUNCH_LEAN_R
#undef YAW_LAUNCH_LEAN
                hipError_t el = hipGetLastError();
                if (el != hipSuccess) return el;
            }
            return hipSuccess;
        };
        if (P.run_unweighted) {
            HIP_TRY(launch_lean(false));
            ++launches;
        }
        if (P.run_weighted) {
            HIP_TRY(launch_lean(true));
            ++launches;
            HIP_TRY(reduce_partials(P.merged ? (int64_t)n_jobs : P.n_slots, P.slab));  // slabs are reduced per output slot
            launches += 2;
        }
    } else if (n_items > 0) {
        if (P.run_unweighted) {
            const bool priv = lds_for(false, true) <= (size_t)ctx->lds_limit;
            hipError_t e = launch_count_any<false>(priv, P.filter, P.R, ctx, c1, c2, (int)P.n_slots, n_bins, n_edges, P.n_items,
                                                   lds_for(false, priv));
            HIP_TRY(e);
            ++launches;
        }
        if (P.run_weighted) {
            const bool priv = lds_for(true, true) <= (size_t)ctx->lds_limit;
            hipError_t e = launch_count_any<true>(priv, P.filter, P.R, ctx, c1, c2, (int)P.n_slots, n_bins, n_edges, P.n_items,
                                                  lds_for(true, priv));
            HIP_TRY(e);
            ++launches;
            HIP_TRY(reduce_partials(P.n_slots, P.nf));
            launches += 2;
        }
    }
    HIP_TRY(hipEventRecord(ctx->evc1, ctx->stream));
    if (!P.weighted && want_sums) {
        const int thr = 256;
        hipLaunchKernelGGL(k_counts_to_double, dim3((unsigned)((P.n_out + thr - 1) / thr)), dim3(thr), 0, ctx->stream,
                           ctx->d_counts.ptr, ctx->d_sums.ptr, P.n_out);
        HIP_TRY(hipGetLastError());
        ++launches;
    }
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    // one copy brings back the counters and whatever was asked for, into pinned memory
    // (fetch_results = false: the caller reduces the results on the device first and fetches what is left; counters only here)
    const size_t fetch = !fetch_results ? o_counts : (want_sums ? out_bytes : (want_counts ? o_sums : o_counts));
    HIP_TRY(hipMemcpyAsync(ctx->out.h, ctx->out.d, fetch, hipMemcpyDeviceToHost, ctx->stream));
    cs.pending = true;
    cs.o_ctr = o_ctr; cs.o_counts = o_counts; cs.o_sums = o_sums;
    cs.band_ran = band_ran; cs.run_unweighted = P.run_unweighted; cs.run_weighted = P.run_weighted;
    cs.cand = P.cand; cs.abytes = P.abytes; cs.n_pot = n_pot; cs.segmented = seg_cap != 0;
    cs.launches = launches; cs.kernel = P.kernel; cs.mode = P.mode;
    cs.n_orient = P.n_orient;
    cs.band_variant = !band_ran ? 0 : (P.band32 ? 32 : (P.band_fine ? 33 : 64));
    cs.merged_triples = band_ran && P.triple ? 1 : 0;
    g_trace.mark("launched");
    return YAWHIP_OK;
}

// Second half: wait for the context's stream, hand the results (contiguous rows of the jobs given to count_enqueue) and
// the statistics over.
// row_index != nullptr: row r of this call's result goes to row row_index[r] of the caller's arrays (rows of row_len values):
// the devices of a multi-device call write their shares straight into place.
// wait_done: wait for the active slot's ev_done (recorded by the caller behind everything this call put on the stream)
// instead of the whole stream -- the requests of a batch behind it keep running.
int count_finish(yawhip_ctx *ctx, const CallState &cs, int64_t *fine_counts, double *fine_sums, yawhip_stats *stats,
                 const int32_t *row_index = nullptr, int64_t row_len = 0, bool wait_done = false) {
    if (stats) memset(stats, 0, sizeof *stats);
    const int64_t n_rows = row_index && row_len > 0 ? cs.n_out / row_len : 0;
    if (!cs.pending) {
        if (!row_index) {
            if (fine_counts) memset(fine_counts, 0, sizeof(int64_t) * (size_t)cs.n_out);
            if (fine_sums) memset(fine_sums, 0, sizeof(double) * (size_t)cs.n_out);
        } else {
            for (int64_t r = 0; r < n_rows; ++r) {
                if (fine_counts) memset(fine_counts + (size_t)row_index[r] * row_len, 0, sizeof(int64_t) * (size_t)row_len);
                if (fine_sums) memset(fine_sums + (size_t)row_index[r] * row_len, 0, sizeof(double) * (size_t)row_len);
            }
        }
        return YAWHIP_OK;
    }
    g_trace.mark("meanwhile");
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->spin_wait) {
        // poll for up to 2 ms (a headline call takes 0.5 ms; the wake-up of a blocked thread costs ~0.01 ms), then block
        hipError_t qe;
        const auto spin0 = std::chrono::steady_clock::now();
        while ((qe = wait_done ? hipEventQuery(ctx->ev_done) : hipStreamQuery(ctx->stream)) == hipErrorNotReady &&
               std::chrono::steady_clock::now() - spin0 < std::chrono::milliseconds(2))
            __builtin_ia32_pause();
        if (qe == hipErrorNotReady) qe = wait_done ? hipEventSynchronize(ctx->ev_done) : hipStreamSynchronize(ctx->stream);
        HIP_TRY(qe);
    } else {
        HIP_TRY(wait_done ? hipEventSynchronize(ctx->ev_done) : hipStreamSynchronize(ctx->stream));
    }
    g_trace.mark("waited");
    if (!row_index) {
        if (fine_counts) memcpy(fine_counts, ctx->out.h + cs.o_counts, sizeof(int64_t) * (size_t)cs.n_out);
        if (fine_sums) memcpy(fine_sums, ctx->out.h + cs.o_sums, sizeof(double) * (size_t)cs.n_out);
    } else {
        const int64_t *hc = reinterpret_cast<const int64_t *>(ctx->out.h + cs.o_counts);
        const double *hs = reinterpret_cast<const double *>(ctx->out.h + cs.o_sums);
        for (int64_t r = 0; r < n_rows; ++r) {
            if (fine_counts) memcpy(fine_counts + (size_t)row_index[r] * row_len, hc + (size_t)r * row_len, sizeof(int64_t) * (size_t)row_len);
            if (fine_sums) memcpy(fine_sums + (size_t)row_index[r] * row_len, hs + (size_t)r * row_len, sizeof(double) * (size_t)row_len);
        }
    }
    const unsigned long long *ctr = reinterpret_cast<const unsigned long long *>(ctx->out.h + cs.o_ctr);
    g_trace.mark("copied");
    if (stats) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        float cms = 0.f;
        HIP_TRY(hipEventElapsedTime(&cms, ctx->evc0, ctx->evc1));
        stats->count_ms = cms;
        stats->candidate_pairs = cs.cand;
        unsigned long long tile_pairs = 0;
        for (int i = 0; i < EVAL_SLOTS; ++i) tile_pairs += ctr[10 + 8 * (size_t)i];
        stats->evaluated_pairs = (int64_t)tile_pairs * ((cs.run_unweighted ? 1 : 0) + (cs.run_weighted ? 1 : 0));
        if (cs.band_ran) {  // band kernel: the entries its lanes really walked (both launches of a weighted + counts call)
            unsigned long long ev = 0;
            for (int i = 0; i < EVAL_SLOTS; ++i) ev += ctr[8 + 8 * (size_t)i];
            stats->evaluated_pairs = (int64_t)ev;
        }
        stats->algorithmic_bytes = cs.abytes;
        stats->n_workgroups = cs.n_pot > 0 ? (int64_t)ctr[0] : 0;
        if (cs.segmented && cs.n_pot > 0)
            for (int sg = 0; sg < ITEM_SEGS; ++sg) stats->n_workgroups += (int64_t)ctr[ITEM_SEG_CTR(sg)];
        stats->n_launches = cs.launches;
        stats->kernel_used = cs.kernel;
        stats->layout_mode = cs.mode;
        stats->n_orientations = cs.n_orient;
        stats->band_variant = cs.band_variant;
        stats->merged_triples = cs.merged_triples;
        if (cs.band_ran)
            for (int i = 0; i < EVAL_SLOTS; ++i) stats->exact_reevaluations += (int64_t)ctr[9 + 8 * (size_t)i];
        stats->kernel_ms = ms;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - cs.wall0).count();
    }
    g_trace.mark("stats");
    return YAWHIP_OK;
}

void add_stats(yawhip_stats &total, const yawhip_stats &part, bool side_by_side) {
    total.candidate_pairs += part.candidate_pairs;
    total.evaluated_pairs += part.evaluated_pairs;
    total.algorithmic_bytes += part.algorithmic_bytes;
    total.n_workgroups += part.n_workgroups;
    total.n_launches += part.n_launches;
    total.kernel_used = part.kernel_used;
    total.layout_mode = part.layout_mode;
    total.n_orientations = std::max(total.n_orientations, part.n_orientations);
    total.band_variant = part.band_variant;
    total.merged_triples = part.merged_triples;
    total.exact_reevaluations += part.exact_reevaluations;
    if (side_by_side) {  // devices of one call run at the same time: the slowest counts
        total.kernel_ms = std::max(total.kernel_ms, part.kernel_ms);
        total.count_ms = std::max(total.count_ms, part.count_ms);
    } else {             // pieces of one job list on one device run one after the other
        total.kernel_ms += part.kernel_ms;
        total.count_ms += part.count_ms;
    }
}

// One job list on one device, cut in halves as often as count_enqueue asks for (SPLIT_JOBS).
int run_single(yawhip_ctx *ctx, const yawhip_catalog *c1, const yawhip_catalog *c2, int32_t n_jobs, const int32_t *jobs,
               int32_t n_bins, int32_t n_edges, const double *t, int32_t kernel, int64_t *fine_counts, double *fine_sums,
               yawhip_stats *stats, const std::function<void()> *meanwhile = nullptr) {
    // meanwhile: host work of the caller that does not need the result, done while the device counts (once)
    CallState cs;
    int rc = count_enqueue(ctx, c1, c2, n_jobs, jobs, n_bins, n_edges, t, kernel, fine_counts != nullptr, fine_sums != nullptr,
                           nullptr, cs);
    if (meanwhile && (rc == YAWHIP_OK || rc == SPLIT_JOBS)) (*meanwhile)();
    if (rc == YAWHIP_OK) return count_finish(ctx, cs, fine_counts, fine_sums, stats);
    if (rc != SPLIT_JOBS) return rc;
    const int32_t half = n_jobs / 2;
    const size_t row = (size_t)n_bins * (size_t)(n_edges - 1);
    yawhip_stats a{}, b{};
    rc = run_single(ctx, c1, c2, half, jobs, n_bins, n_edges, t, kernel, fine_counts, fine_sums, &a);
    if (rc != YAWHIP_OK) return rc;
    rc = run_single(ctx, c1, c2, n_jobs - half, jobs + 2 * (size_t)half, n_bins, n_edges, t, kernel,
                    fine_counts ? fine_counts + (size_t)half * row : nullptr, fine_sums ? fine_sums + (size_t)half * row : nullptr, &b);
    if (rc != YAWHIP_OK) return rc;
    if (stats) {
        memset(stats, 0, sizeof *stats);
        add_stats(*stats, a, false);
        add_stats(*stats, b, false);
        stats->total_ms = a.total_ms + b.total_ms;
    }
    return YAWHIP_OK;
}

// FNV-1a over the inputs that determine the job partition of a multi-device call
uint64_t plan_key(const yawhip_catalog *c1, const yawhip_catalog *c2, int32_t n_jobs, const int32_t *jobs, int32_t n_bins,
                  int32_t n_edges, const double *t, int32_t kernel, size_t n_dev) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    };
    mix(&c1, sizeof c1); mix(&c2, sizeof c2); mix(&c1->n, sizeof c1->n); mix(&c2->n, sizeof c2->n);
    mix(&n_jobs, sizeof n_jobs); mix(jobs, sizeof(int32_t) * 2 * (size_t)n_jobs);
    mix(&n_bins, sizeof n_bins); mix(&n_edges, sizeof n_edges); mix(t, sizeof(double) * (size_t)n_bins * n_edges);
    mix(&kernel, sizeof kernel); mix(&n_dev, sizeof n_dev);
    return h;
}

}  // namespace

namespace {
// Host-side grouping of catalogue columns (no device involved): a stable counting sort by key, run by a few threads.
// Chunk c of the input counts its keys; group g then holds the entries of chunk 0, chunk 1, ... in input order, so every
// chunk knows where its entries of every group go and scatters all columns in one pass over its slice.
template <typename K>
static int group_columns(int64_t n, const K *keys, int64_t num_groups, int32_t n_cols, const double *const *in, double *const *out,
                         int64_t *sizes, int n_threads) {
    const int64_t min_chunk = 1 << 16;
    int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, (n + min_chunk - 1) / min_chunk));
    std::vector<std::vector<int64_t>> hist((size_t)T, std::vector<int64_t>((size_t)num_groups, 0));
    std::atomic<int> bad{0};
    auto bounds = [&](int c) { return std::make_pair(n * c / T, n * (c + 1) / T); };
    auto run = [&](auto &&fn) {
        if (T == 1) { fn(0); return; }
        std::vector<std::thread> th;
        for (int c = 0; c < T; ++c) th.emplace_back(fn, c);
        for (auto &t : th) t.join();
    };
    run([&](int c) {
        auto [lo, hi] = bounds(c);
        int64_t *h = hist[(size_t)c].data();
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t k = (int64_t)keys[i];
            if (k >= num_groups) { bad.store(1); return; }
            if (k >= 0) ++h[k];
        }
    });
    if (bad.load()) return fail(YAWHIP_ERR_INVALID, "yawhip_host_group_columns: key >= num_groups");
    int64_t at = 0;
    for (int64_t g = 0; g < num_groups; ++g) {
        int64_t size = 0;
        for (int c = 0; c < T; ++c) {
            const int64_t cnt = hist[(size_t)c][(size_t)g];
            hist[(size_t)c][(size_t)g] = at + size;  // first slot of chunk c in group g
            size += cnt;
        }
        sizes[g] = size;
        at += size;
    }
    if (n_cols > 0)
        run([&](int c) {
            auto [lo, hi] = bounds(c);
            int64_t *h = hist[(size_t)c].data();
            for (int64_t i = lo; i < hi; ++i) {
                const int64_t k = (int64_t)keys[i];
                if (k < 0) continue;
                const int64_t dst = h[k]++;
                for (int32_t col = 0; col < n_cols; ++col) out[col][dst] = in[col][i];
            }
        });
    return YAWHIP_OK;
}

}  // namespace

extern "C" {

int yawhip_count_pairs(yawhip_ctx *ctx, const yawhip_catalog *c1, const yawhip_catalog *c2, int32_t n_jobs,
                       const int32_t *jobs, int32_t n_bins, int32_t n_edges, const double *t, int32_t kernel,
                       int64_t *fine_counts, double *fine_sums, yawhip_stats *stats) {
    if (stats) memset(stats, 0, sizeof *stats);
    if (!ctx || !c1 || !c2) return fail(YAWHIP_ERR_INVALID, "yawhip_count_pairs: NULL handle");
    if (ctx->peers.empty() || n_jobs < 2)
        return run_single(ctx, c1, c2, n_jobs, jobs, n_bins, n_edges, t, kernel, fine_counts, fine_sums, stats);
    // ---- several devices: the independent jobs are split over them (replaces the reference's process pool,
    // src/yaw/utils/parallel.py:251-346). Every device holds both catalogues; a job's rows of the result come from
    // exactly one device, so nothing has to be reduced: the rows are copied into place.
    if (c1->ctx != ctx || c2->ctx != ctx) return fail(YAWHIP_ERR_MISMATCH, "catalogues belong to another context");
    const size_t n_dev = ctx->peers.size() + 1;
    if (c1->replicas.size() != n_dev - 1 || c2->replicas.size() != n_dev - 1)
        return fail(YAWHIP_ERR_MISMATCH, "catalogue was not uploaded to every device of the context");
    if (n_jobs < 0 || n_bins <= 0 || n_edges < 2 || !t || !jobs) return fail(YAWHIP_ERR_INVALID, "yawhip_count_pairs: bad sizes");
    const auto wall0 = std::chrono::steady_clock::now();
    // the plan: evaluated pairs per job from the item builder (device 0), longest-processing-time-first over the devices;
    // it depends on the inputs only and is kept for the next call with the same inputs
    const uint64_t key = plan_key(c1, c2, n_jobs, jobs, n_bins, n_edges, t, kernel, n_dev);
    if (ctx->plan.key != key || ctx->plan.parts.size() != n_dev) {
        std::vector<int64_t> work((size_t)n_jobs, 0);
        CallState cs;
        int rc = count_enqueue(ctx, c1, c2, n_jobs, jobs, n_bins, n_edges, t, kernel, false, false, work.data(), cs);
        if (rc != YAWHIP_OK) return rc;
        std::vector<int32_t> order((size_t)n_jobs);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return work[(size_t)a] > work[(size_t)b]; });
        std::vector<double> load(n_dev, 0.0);
        ctx->plan.parts.assign(n_dev, {});
        const double fixed = 2.0e5;  // evaluated-pair equivalent of touching a job at all
        for (int32_t j : order) {
            const size_t d = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
            ctx->plan.parts[d].push_back(j);
            load[d] += (double)work[(size_t)j] + fixed;
        }
        for (auto &part : ctx->plan.parts) std::sort(part.begin(), part.end());
        ctx->plan.key = key;
    }
    const int64_t row = (int64_t)n_bins * (n_edges - 1);
    std::vector<CallState> states(n_dev);
    std::vector<std::vector<int32_t>> sub(n_dev);
    std::vector<char> later(n_dev, 0);  // shares that have to be cut in pieces: counted after the others, one by one
    for (size_t d = 0; d < n_dev; ++d) {  // enqueue everywhere first: the devices work side by side
        for (int32_t j : ctx->plan.parts[d]) { sub[d].push_back(jobs[2 * j]); sub[d].push_back(jobs[2 * j + 1]); }
        yawhip_ctx *dc = d == 0 ? ctx : ctx->peers[d - 1];
        const yawhip_catalog *a = d == 0 ? c1 : c1->replicas[d - 1], *b = d == 0 ? c2 : c2->replicas[d - 1];
        const int rc = count_enqueue(dc, a, b, (int32_t)ctx->plan.parts[d].size(), sub[d].data(), n_bins, n_edges, t, kernel,
                                     fine_counts != nullptr, fine_sums != nullptr, nullptr, states[d]);
        if (rc == SPLIT_JOBS) {
            later[d] = 1;
        } else if (rc != YAWHIP_OK) {
            for (size_t e = 0; e < d; ++e) (void)hipStreamSynchronize((e == 0 ? ctx : ctx->peers[e - 1])->stream);
            return rc;
        }
    }
    std::vector<int64_t> rows_c;
    std::vector<double> rows_s;
    yawhip_stats total{}, part{};
    int rc_all = YAWHIP_OK;
    for (size_t d = 0; d < n_dev; ++d) {  // every device's copy into its pinned buffer is already under way: drain in turn
        yawhip_ctx *dc = d == 0 ? ctx : ctx->peers[d - 1];
        const yawhip_catalog *a = d == 0 ? c1 : c1->replicas[d - 1], *b = d == 0 ? c2 : c2->replicas[d - 1];
        const size_t nj = ctx->plan.parts[d].size();
        int rc;
        if (later[d]) {  // a share that is counted in pieces: through a temporary, then into place
            if (fine_counts) rows_c.resize(nj * (size_t)row);
            if (fine_sums) rows_s.resize(nj * (size_t)row);
            rc = run_single(dc, a, b, (int32_t)nj, sub[d].data(), n_bins, n_edges, t, kernel, fine_counts ? rows_c.data() : nullptr,
                            fine_sums ? rows_s.data() : nullptr, &part);
            if (rc == YAWHIP_OK)
                for (size_t r = 0; r < nj; ++r) {
                    const size_t j = (size_t)ctx->plan.parts[d][r];
                    if (fine_counts) memcpy(fine_counts + j * (size_t)row, rows_c.data() + r * (size_t)row, sizeof(int64_t) * (size_t)row);
                    if (fine_sums) memcpy(fine_sums + j * (size_t)row, rows_s.data() + r * (size_t)row, sizeof(double) * (size_t)row);
                }
        } else {         // rows go from the device's pinned buffer straight into the caller's arrays
            rc = count_finish(dc, states[d], fine_counts, fine_sums, &part, ctx->plan.parts[d].data(), row);
        }
        if (rc != YAWHIP_OK) { rc_all = rc; continue; }  // keep draining the other devices
        add_stats(total, part, true);
    }
    (void)hipSetDevice(ctx->device);  // leave the thread on the context's first device, as single-device calls do
    if (rc_all != YAWHIP_OK) return rc_all;
    total.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    if (stats) *stats = total;
    return YAWHIP_OK;
}

int yawhip_count_pairs_rows_device(yawhip_ctx *ctx, const yawhip_catalog *c1, const yawhip_catalog *c2, int32_t n_jobs,
                                   const int32_t *jobs, int32_t n_bins, int32_t n_edges, const double *t, int32_t kernel,
                                   int64_t n_rows_total, const int32_t *row_index, double **device_rows, yawhip_stats *stats) {
    if (stats) memset(stats, 0, sizeof *stats);
    if (!ctx || !c1 || !c2 || !device_rows) return fail(YAWHIP_ERR_INVALID, "yawhip_count_pairs_rows_device: NULL argument");
    *device_rows = nullptr;
    if (!ctx->peers.empty()) return fail(YAWHIP_ERR_INVALID, "yawhip_count_pairs_rows_device: single-device contexts only");
    if (n_jobs < 0 || n_rows_total < n_jobs || n_bins <= 0 || n_edges < 2 || (n_jobs > 0 && (!jobs || !row_index)))
        return fail(YAWHIP_ERR_INVALID, "yawhip_count_pairs_rows_device: bad sizes or NULL arrays");
    const int64_t row = (int64_t)n_bins * (n_edges - 1);
    for (int j = 0; j < n_jobs; ++j)
        if (row_index[j] < 0 || row_index[j] >= n_rows_total)
            return fail(YAWHIP_ERR_INVALID, "yawhip_count_pairs_rows_device: row index %d outside [0, %lld)", row_index[j], (long long)n_rows_total);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n_full = (size_t)n_rows_total * (size_t)row + 1;  // + 1: the caller's status element
    HIP_TRY(ctx->d_full.reserve(n_full));
    HIP_TRY(ctx->d_rowidx.reserve((size_t)std::max(n_jobs, 1)));
    CallState cs;
    // (the rows stay on the device: only the statistics counters are fetched)
    int rc = count_enqueue(ctx, c1, c2, n_jobs, jobs, n_bins, n_edges, t, kernel, false, true, nullptr, cs, /*fetch_results=*/false);
    if (rc == SPLIT_JOBS) {
        // a job list that is counted in pieces: through the host (rare: weighted slabs beyond the budget)
        std::vector<double> rows((size_t)n_jobs * (size_t)row), full(n_full, 0.0);
        rc = run_single(ctx, c1, c2, n_jobs, jobs, n_bins, n_edges, t, kernel, nullptr, rows.data(), stats);
        if (rc != YAWHIP_OK) return rc;
        for (int j = 0; j < n_jobs; ++j)
            memcpy(full.data() + (size_t)row_index[j] * row, rows.data() + (size_t)j * row, sizeof(double) * (size_t)row);
        HIP_TRY(hipMemcpy(ctx->d_full.ptr, full.data(), sizeof(double) * n_full, hipMemcpyHostToDevice));
        *device_rows = ctx->d_full.ptr;
        return YAWHIP_OK;
    }
    if (rc != YAWHIP_OK) return rc;
    HIP_TRY(hipMemsetAsync(ctx->d_full.ptr, 0, sizeof(double) * n_full, ctx->stream));
    if (cs.pending && n_jobs > 0) {
        HIP_TRY(hipMemcpyAsync(ctx->d_rowidx.ptr, row_index, sizeof(int32_t) * (size_t)n_jobs, hipMemcpyHostToDevice, ctx->stream));
        const int64_t n = (int64_t)n_jobs * row;
        hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_sums.ptr,
                           ctx->d_rowidx.ptr, row, n, ctx->d_full.ptr);
        HIP_TRY(hipGetLastError());
    }
    rc = count_finish(ctx, cs, nullptr, nullptr, stats);  // waits for the stream: the rows are in place when this returns
    if (rc != YAWHIP_OK) return rc;
    *device_rows = ctx->d_full.ptr;
    return YAWHIP_OK;
}

}  // extern "C"

namespace {
// ndarray.sum() of a contiguous float64 vector, in numpy's order (pairwise summation: plain loop below 8 values, eight
// running sums up to 128, halves above): the reference sums the fine bins of a scale this way (trees.py:134-160), so
// separation-weighted counts of unweighted catalogues come out bit for bit as the reference's.
double numpy_sum(const double *a, int64_t n) {
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return numpy_sum(a, n2) + numpy_sum(a + n2, n - n2);
}
}  // namespace

extern "C" {

int yawhip_count_pairs_dense(yawhip_ctx *ctx, const yawhip_catalog *c1, const yawhip_catalog *c2, int32_t n_jobs,
                             const int32_t *jobs, int32_t n_bins, int32_t n_edges, const double *t, int32_t kernel,
                             int32_t n_scales, const int32_t *slices, const double *fine_factors, int32_t halve_diagonal,
                             double *dense, yawhip_stats *stats) {
    yawhip_dense_request req{c1, c2, n_jobs, halve_diagonal, jobs, dense, stats};
    return yawhip_count_pairs_dense_batch(ctx, 1, &req, n_bins, n_edges, t, kernel, n_scales, slices, fine_factors);
}

}  // extern "C"

namespace {

struct DenseState {
    CallState cs;
    bool enqueued = false;        // on the stream (false: counted by the blocking route at finish time)
    bool device_combine = false;  // the per-scale values were recombined on the device (k_combine_scales)
    bool weighted = false;
    int slot = 0;
    int64_t n_comb = 0;
    size_t h_comb_off = 0;
};

int dense_check(const yawhip_dense_request &r, int32_t n_bins, int32_t n_edges, int32_t n_scales, const int32_t *slices) {
    if (!r.c1 || !r.c2) return fail(YAWHIP_ERR_INVALID, "yawhip_count_pairs_dense: NULL handle");
    if (r.n_jobs < 0 || n_bins <= 0 || n_edges < 2 || n_scales <= 0 || !slices || !r.dense || (r.n_jobs > 0 && !r.jobs))
        return fail(YAWHIP_ERR_INVALID, "yawhip_count_pairs_dense: bad sizes or NULL arrays");
    const int64_t P = r.c1->n_patches;
    for (int64_t j = 0; j < r.n_jobs; ++j)
        if (r.jobs[2 * j] < 0 || r.jobs[2 * j] >= P || r.jobs[2 * j + 1] < 0 || r.jobs[2 * j + 1] >= P)
            return fail(YAWHIP_ERR_INVALID, "job %lld has a patch id outside [0,%lld)", (long long)j, (long long)P);
    return YAWHIP_OK;
}

// Put one request on the context's stream, in the ACTIVE slot: the count, the recombination of several fine bins on the
// device (one device, E - 1 > 1: S values per (job, bin) come back instead of E - 1 -- separation weights: 51 -> 1), the copies
// into the slot's pinned buffers, and the slot's ev_done behind all of it. Nothing waits.
int dense_enqueue(yawhip_ctx *ctx, const yawhip_dense_request &r, int32_t n_bins, int32_t n_edges, const double *t, int32_t kernel,
                  int32_t n_scales, const int32_t *slices, const double *fine_factors, DenseState &ds) {
    const int nf = n_edges - 1;
    ds.weighted = r.c1->w != nullptr || r.c2->w != nullptr;
    ds.slot = ctx->slot;
    ds.device_combine = nf > 1;
    int rc = count_enqueue(ctx, r.c1, r.c2, r.n_jobs, r.jobs, n_bins, n_edges, t, kernel, !ds.weighted, ds.weighted, nullptr, ds.cs,
                           /*fetch_results=*/!ds.device_combine);
    if (rc == SPLIT_JOBS) return YAWHIP_OK;  // counted in pieces by the blocking route when its turn comes (ds.enqueued stays false)
    if (rc != YAWHIP_OK) return rc;
    if (ds.device_combine) {
        ds.n_comb = (int64_t)r.n_jobs * n_bins * n_scales;
        const size_t b_slices = align16(sizeof(int32_t) * 2 * (size_t)n_bins * n_scales);
        const size_t b_fact = fine_factors ? align16(sizeof(double) * (size_t)n_bins * nf) : 0;
        HIP_TRY(ctx->comb.reserve(b_slices + b_fact + sizeof(double) * (size_t)std::max<int64_t>(ds.n_comb, 1)));
        memcpy(ctx->comb.h, slices, sizeof(int32_t) * 2 * (size_t)n_bins * n_scales);
        if (fine_factors) memcpy(ctx->comb.h + b_slices, fine_factors, sizeof(double) * (size_t)n_bins * nf);
        HIP_TRY(hipMemcpyAsync(ctx->comb.d, ctx->comb.h, b_slices + b_fact, hipMemcpyHostToDevice, ctx->stream));
        ds.h_comb_off = b_slices + b_fact;
        double *d_comb = reinterpret_cast<double *>(ctx->comb.d + ds.h_comb_off);
        if (ds.cs.pending && ds.n_comb > 0) {
            hipLaunchKernelGGL(k_combine_scales, dim3((unsigned)((ds.n_comb + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->d_counts.ptr, ctx->d_sums.ptr, ds.weighted ? 1 : 0, (int64_t)r.n_jobs, n_bins, nf, n_scales,
                               reinterpret_cast<const int32_t *>(ctx->comb.d),
                               fine_factors ? reinterpret_cast<const double *>(ctx->comb.d + b_slices) : nullptr, d_comb);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(ctx->comb.h + ds.h_comb_off, d_comb, sizeof(double) * (size_t)ds.n_comb, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    HIP_TRY(hipEventRecord(ctx->ev_done, ctx->stream));
    ds.enqueued = true;
    return YAWHIP_OK;
}

int dense_blocking(yawhip_ctx *ctx, const yawhip_dense_request &r, int32_t n_bins, int32_t n_edges, const double *t, int32_t kernel,
                   int32_t n_scales, const int32_t *slices, const double *fine_factors);

// Wait for a request's slot and write its result tensor: the host epilogue, O(jobs x B x S), of PatchLinkage.count_pairs
// (reference src/yaw/correlation/measurements.py:354-364): halving of the doubly counted diagonal of an autocorrelation and
// the scatter into [scale][bin][patch i][patch j]; unlinked slots are 0. Values come straight from the slot's pinned buffer.
int dense_finish(yawhip_ctx *ctx, const yawhip_dense_request &r, int32_t n_bins, int32_t n_edges, const double *t, int32_t kernel,
                 int32_t n_scales, const int32_t *slices, const double *fine_factors, DenseState &ds) {
    if (!ds.enqueued) return dense_blocking(ctx, r, n_bins, n_edges, t, kernel, n_scales, slices, fine_factors);
    const int64_t P = r.c1->n_patches;
    const int32_t n_jobs = r.n_jobs;
    const int32_t *jobs = r.jobs;
    double *dense = r.dense;
    // (the result tensor is cleared while the device counts: 1 MB, 0.04 ms at the headline)
    memset(dense, 0, sizeof(double) * (size_t)n_scales * (size_t)n_bins * (size_t)(P * P));
    const int rc = count_finish(ctx, ds.cs, nullptr, nullptr, r.stats, nullptr, 0, /*wait_done=*/true);
    if (rc != YAWHIP_OK) return rc;
    if (!ds.cs.pending) return YAWHIP_OK;
    if (ds.device_combine) {
        const double *h_comb = reinterpret_cast<const double *>(ctx->comb.h + ds.h_comb_off);
        for (int k = 0; k < n_bins; ++k)
            for (int64_t j = 0; j < n_jobs; ++j) {
                const int64_t p = jobs[2 * j], q = jobs[2 * j + 1];
                const double f = (r.halve_diagonal && p == q) ? 0.5 : 1.0;
                for (int s_ = 0; s_ < n_scales; ++s_)
                    dense[(((size_t)s_ * n_bins + k) * P + p) * P + q] = h_comb[((size_t)j * n_bins + k) * n_scales + s_] * f;
            }
        return YAWHIP_OK;
    }
    // one fine bin per (job, bin): numpy's sum of one element is the element; unweighted catalogues are counted in int64 and
    // converted here (exact below 2^53, the reference's .astype(float64), trees.py:353)
    const int64_t *hc = reinterpret_cast<const int64_t *>(ctx->out.h + ds.cs.o_counts);
    const double *hs = reinterpret_cast<const double *>(ctx->out.h + ds.cs.o_sums);
    // position and factor of every job, once; then job by job: a job's B values are read in one piece, each goes to its own
    // [P, P] slice (13 200 scattered stores at the headline)
    thread_local std::vector<int64_t> cell;
    thread_local std::vector<double> half;
    cell.resize((size_t)n_jobs);
    half.resize((size_t)n_jobs);
    for (int64_t j = 0; j < n_jobs; ++j) {
        cell[(size_t)j] = (int64_t)jobs[2 * j] * P + jobs[2 * j + 1];
        half[(size_t)j] = (r.halve_diagonal && jobs[2 * j] == jobs[2 * j + 1]) ? 0.5 : 1.0;
    }
    const size_t PP = (size_t)(P * P);
    for (int s_ = 0; s_ < n_scales; ++s_) {
        double *base = dense + (size_t)s_ * n_bins * PP;
        for (int64_t j = 0; j < n_jobs; ++j) {
            double *dst = base + cell[(size_t)j];
            const double f = half[(size_t)j];
            const int64_t *cj = hc + (size_t)j * n_bins;
            const double *sj = hs + (size_t)j * n_bins;
            for (int k = 0; k < n_bins; ++k) {
                if (!(slices[2 * ((int64_t)k * n_scales + s_) + 1] > slices[2 * ((int64_t)k * n_scales + s_)])) continue;  // (cleared above)
                const double v = ds.weighted ? sj[k] : (double)cj[k];
                dst[(size_t)k * PP] = (fine_factors ? v * fine_factors[(size_t)k] : v) * f;
            }
        }
    }
    g_trace.mark("scattered");
    g_trace.flush();
    return YAWHIP_OK;
}

// The blocking route of one request: several devices in the context (the library splits the job list), or a job list that
// has to be counted in pieces (weighted slabs beyond the budget). Per-job fine values on the host, then the epilogue.
int dense_blocking(yawhip_ctx *ctx, const yawhip_dense_request &r, int32_t n_bins, int32_t n_edges, const double *t, int32_t kernel,
                   int32_t n_scales, const int32_t *slices, const double *fine_factors) {
    const yawhip_catalog *c1 = r.c1, *c2 = r.c2;
    const int32_t n_jobs = r.n_jobs;
    const int32_t *jobs = r.jobs;
    const int32_t halve_diagonal = r.halve_diagonal;
    double *dense = r.dense;
    yawhip_stats *stats = r.stats;
    const int nf = n_edges - 1;
    const int64_t P = c1->n_patches, row = (int64_t)n_bins * nf;
    const bool weighted = c1->w != nullptr || c2->w != nullptr;
    // unweighted catalogues are counted in int64 and converted here (exact below 2^53, the reference's .astype(float64),
    // trees.py:353): one kernel and half the device-to-host bytes less than asking the device for both
    const size_t n_fine = (size_t)std::max<int64_t>((int64_t)n_jobs * row, 1);
    std::unique_ptr<double[]> fine_s(weighted ? new (std::nothrow) double[n_fine] : nullptr);
    std::unique_ptr<int64_t[]> fine_c(weighted ? nullptr : new (std::nothrow) int64_t[n_fine]);
    if (!fine_s && !fine_c) return fail(YAWHIP_ERR_OOM, "yawhip_count_pairs_dense: out of host memory");
    // Host epilogue, O(jobs x B x E), of PatchLinkage.count_pairs (reference src/yaw/correlation/measurements.py:354-364 with
    // src/yaw/catalog/trees.py:358-362,134-160 applied per job): separation weights, per-scale sums of the fine bins, halving
    // of the doubly counted diagonal of an autocorrelation, scatter into [scale][bin][patch i][patch j]; unlinked slots are 0.
    // The tensor is cleared while the device counts (1 MB, 0.04 ms at the headline) when the call runs on one device.
    bool cleared = false;
    const std::function<void()> clear = [&]() {
        memset(dense, 0, sizeof(double) * (size_t)n_scales * (size_t)n_bins * (size_t)(P * P));
        cleared = true;
    };
    const int rc = ctx->peers.empty() ? run_single(ctx, c1, c2, n_jobs, jobs, n_bins, n_edges, t, kernel, fine_c.get(), fine_s.get(), stats, &clear)
                                      : yawhip_count_pairs(ctx, c1, c2, n_jobs, jobs, n_bins, n_edges, t, kernel, fine_c.get(), fine_s.get(), stats);
    if (rc != YAWHIP_OK) return rc;
    g_trace.mark("finished");
    if (!cleared) clear();
    std::vector<double> scaled((size_t)nf);
    std::vector<int64_t> cell((size_t)n_jobs);  // p * P + q and the factor of every job, once
    std::vector<double> half((size_t)n_jobs);
    for (int64_t j = 0; j < n_jobs; ++j) {
        cell[(size_t)j] = (int64_t)jobs[2 * j] * P + jobs[2 * j + 1];
        half[(size_t)j] = (halve_diagonal && jobs[2 * j] == jobs[2 * j + 1]) ? 0.5 : 1.0;
    }
    for (int k = 0; k < n_bins; ++k) {  // bin by bin: the scattered writes of one pass stay inside S slices of [P, P]
        const double *wk = fine_factors ? fine_factors + (size_t)k * nf : nullptr;
        if (nf == 1) {  // one value per (job, bin), the usual call: numpy's sum of one element is the element
            for (int s_ = 0; s_ < n_scales; ++s_) {
                const bool take = slices[2 * ((int64_t)k * n_scales + s_) + 1] > slices[2 * ((int64_t)k * n_scales + s_)];
                if (!take) continue;  // (cleared above)
                double *slice = dense + ((size_t)s_ * n_bins + k) * (size_t)(P * P);
                const double w0 = wk ? wk[0] : 1.0;
                if (weighted)
                    for (int64_t j = 0; j < n_jobs; ++j) {
                        const double v = fine_s[(size_t)j * row + (size_t)k];
                        slice[cell[(size_t)j]] = (wk ? v * w0 : v) * half[(size_t)j];
                    }
                else
                    for (int64_t j = 0; j < n_jobs; ++j) {
                        const double v = (double)fine_c[(size_t)j * row + (size_t)k];
                        slice[cell[(size_t)j]] = (wk ? v * w0 : v) * half[(size_t)j];
                    }
            }
            continue;
        }
        for (int64_t j = 0; j < n_jobs; ++j) {
            const double f = half[(size_t)j];
            const size_t at = (size_t)j * row + (size_t)k * nf;
            const double *fk = weighted ? fine_s.get() + at : scaled.data();
            if (!weighted)
                for (int e = 0; e < nf; ++e) scaled[(size_t)e] = (double)fine_c[at + (size_t)e];
            if (wk) {  // counts *= weights (trees.py:358-360), then the sums
                for (int e = 0; e < nf; ++e) scaled[(size_t)e] = fk[e] * wk[e];
                fk = scaled.data();
            }
            for (int s_ = 0; s_ < n_scales; ++s_) {
                const int lo = slices[2 * ((int64_t)k * n_scales + s_)], hi = slices[2 * ((int64_t)k * n_scales + s_) + 1];
                const double acc = hi > lo ? numpy_sum(fk + lo, hi - lo) : 0.0;
                dense[((size_t)s_ * n_bins + k) * (size_t)(P * P) + (size_t)cell[(size_t)j]] = acc * f;
            }
        }
    }
    g_trace.mark("scattered");
    g_trace.flush();
    return YAWHIP_OK;
}

}  // namespace

extern "C" {

int yawhip_count_pairs_dense_batch(yawhip_ctx *ctx, int32_t n_requests, const yawhip_dense_request *requests, int32_t n_bins,
                                   int32_t n_edges, const double *t, int32_t kernel, int32_t n_scales, const int32_t *slices,
                                   const double *fine_factors) {
    if (!ctx || n_requests < 0 || (n_requests > 0 && !requests))
        return fail(YAWHIP_ERR_INVALID, "yawhip_count_pairs_dense_batch: NULL argument");
    for (int i = 0; i < n_requests; ++i) {
        if (requests[i].stats) memset(requests[i].stats, 0, sizeof(yawhip_stats));
        const int rc = dense_check(requests[i], n_bins, n_edges, n_scales, slices);
        if (rc != YAWHIP_OK) return rc;
    }
    const int nf = n_edges - 1;
    if (n_bins > 0 && n_scales > 0 && slices)
        for (int64_t i = 0; i < (int64_t)n_bins * n_scales; ++i)
            if (slices[2 * i] < 0 || slices[2 * i + 1] > nf)
                return fail(YAWHIP_ERR_INVALID, "yawhip_count_pairs_dense: slice %lld outside [0, %d]", (long long)i, nf);
    if (n_requests == 0) return YAWHIP_OK;
    if (!ctx->peers.empty()) {  // several devices: every request is split over them by yawhip_count_pairs, one after the other
        for (int i = 0; i < n_requests; ++i) {
            const int rc = dense_blocking(ctx, requests[i], n_bins, n_edges, t, kernel, n_scales, slices, fine_factors);
            if (rc != YAWHIP_OK) return rc;
        }
        return YAWHIP_OK;
    }
    // One device: up to MAX_BATCH requests are on the stream at once, each in a slot of its own (tables, work items, partial
    // sums, result block, events). The host enqueues request k + 1 while the device counts request k, and writes the tensor of
    // request k (its epilogue) while the device counts the ones behind it; the device never waits for the host in between.
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<DenseState> st((size_t)n_requests);
    int rc_all = YAWHIP_OK, done = 0;
    auto finish_next = [&]() {
        hipError_t e = use_slot(ctx, done % MAX_BATCH);
        int rc = e == hipSuccess ? dense_finish(ctx, requests[done], n_bins, n_edges, t, kernel, n_scales, slices, fine_factors, st[(size_t)done])
                                 : fail(YAWHIP_ERR_HIP, "event creation failed: %s", hipGetErrorString(e));
        if (rc != YAWHIP_OK && rc_all == YAWHIP_OK) rc_all = rc;
        ++done;
    };
    int issued = 0;
    for (; issued < n_requests && rc_all == YAWHIP_OK; ++issued) {
        if (issued - done >= MAX_BATCH) finish_next();  // its slot is needed again
        if (rc_all != YAWHIP_OK) break;
        hipError_t e = use_slot(ctx, issued % MAX_BATCH);
        if (e != hipSuccess) { rc_all = fail(YAWHIP_ERR_HIP, "event creation failed: %s", hipGetErrorString(e)); break; }
        const int rc = dense_enqueue(ctx, requests[issued], n_bins, n_edges, t, kernel, n_scales, slices, fine_factors, st[(size_t)issued]);
        if (rc != YAWHIP_OK) { rc_all = rc; break; }
    }
    if (rc_all != YAWHIP_OK) {  // leave nothing in flight behind an error
        (void)hipStreamSynchronize(ctx->stream);
        (void)use_slot(ctx, 0);
        return rc_all;
    }
    while (done < issued) finish_next();
    (void)use_slot(ctx, 0);
    if (rc_all != YAWHIP_OK) (void)hipStreamSynchronize(ctx->stream);
    return rc_all;
}

int yawhip_ctx_create_multi(const int *device_ids, int n_devices, yawhip_ctx **out) {
    if (!out) return fail(YAWHIP_ERR_INVALID, "yawhip_ctx_create_multi: out is NULL");
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) return fail(YAWHIP_ERR_INVALID, "yawhip_ctx_create_multi: 1 to 64 device ids");
    yawhip_ctx *ctx = nullptr;
    int rc = yawhip_ctx_create(device_ids[0], &ctx);
    if (rc != YAWHIP_OK) return rc;
    for (int i = 1; i < n_devices; ++i) {
        yawhip_ctx *peer = nullptr;
        rc = yawhip_ctx_create(device_ids[i], &peer);
        if (rc != YAWHIP_OK) {
            yawhip_ctx_destroy(ctx);
            return rc;
        }
        ctx->peers.push_back(peer);
    }
    *out = ctx;
    return YAWHIP_OK;
}

int yawhip_ctx_device_count(const yawhip_ctx *ctx, int *n) {
    if (!ctx || !n) return fail(YAWHIP_ERR_INVALID, "yawhip_ctx_device_count: NULL argument");
    *n = (int)ctx->peers.size() + 1;
    return YAWHIP_OK;
}

int yawhip_assign_patches(yawhip_ctx *ctx, int64_t n, const double *x, const double *y, const double *z, int32_t n_centers,
                          const double *centers_xyz, int32_t *patch_out) {
    if (!ctx) return fail(YAWHIP_ERR_INVALID, "yawhip_assign_patches: ctx is NULL");
    if (n < 0 || n_centers <= 0 || !centers_xyz || (n > 0 && (!x || !y || !z || !patch_out)))
        return fail(YAWHIP_ERR_INVALID, "yawhip_assign_patches: bad sizes or NULL arrays");
    if ((size_t)n_centers * 3 * sizeof(double) > (size_t)ctx->lds_limit)
        return fail(YAWHIP_ERR_INVALID, "too many centres (%d) for the LDS table", n_centers);
    if (n == 0) return YAWHIP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    double *dx = nullptr, *dc = nullptr;
    int32_t *dout = nullptr;
    auto cleanup = [&]() {
        if (dx) (void)hipFree(dx);
        if (dc) (void)hipFree(dc);
        if (dout) (void)hipFree(dout);
    };
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&dx), (size_t)3 * n * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dc), (size_t)3 * n_centers * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dout), (size_t)n * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpyAsync(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dx + n, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dx + 2 * n, z, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(dc, centers_xyz, (size_t)3 * n_centers * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        const size_t lds = (size_t)3 * n_centers * sizeof(double);
        if (lds > 64 * 1024)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_assign_patches), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_assign_patches, dim3((unsigned)((n + 255) / 256)), dim3(256), lds, ctx->stream, n, dx, dx + n, dx + 2 * n,
                               n_centers, dc, dout);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipMemcpyAsync(patch_out, dout, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? YAWHIP_ERR_OOM : YAWHIP_ERR_HIP, "yawhip_assign_patches failed: %s", hipGetErrorString(e));
    return YAWHIP_OK;
}

int yawhip_host_group_columns(int64_t n, const void *keys, int32_t key_bytes, int64_t num_groups, int32_t n_cols,
                              const double *const *in, double *const *out, int64_t *sizes, int32_t n_threads) {
    if (n < 0 || num_groups <= 0 || n_cols < 0 || !sizes || (n > 0 && !keys) || (key_bytes != 4 && key_bytes != 8))
        return fail(YAWHIP_ERR_INVALID, "yawhip_host_group_columns: bad sizes or NULL arrays");
    for (int32_t c = 0; c < n_cols; ++c)
        if (n > 0 && (!in || !out || !in[c] || !out[c] || in[c] == out[c]))
            return fail(YAWHIP_ERR_INVALID, "yawhip_host_group_columns: column %d is NULL or aliases its output", c);
    if (n_threads <= 0) n_threads = (int32_t)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    try {
        return key_bytes == 4 ? group_columns(n, (const int32_t *)keys, num_groups, n_cols, in, out, sizes, n_threads)
                              : group_columns(n, (const int64_t *)keys, num_groups, n_cols, in, out, sizes, n_threads);
    } catch (const std::bad_alloc &) {
        return fail(YAWHIP_ERR_OOM, "yawhip_host_group_columns: out of host memory");
    } catch (const std::system_error &err) {
        return fail(YAWHIP_ERR_INVALID, "yawhip_host_group_columns: %s", err.what());
    }
}

int yawhip_host_scatter_rows(int64_t n_rows, int64_t row_len, double *out, int64_t n_cols, const int64_t *cols,
                             const double *vals, int64_t val_row_stride, int64_t val_col_stride, const double *col_factor) {
    if (n_rows < 0 || row_len < 0 || n_cols < 0 || (n_rows * row_len > 0 && !out) || (n_cols > 0 && (!cols || (n_rows > 0 && !vals))))
        return fail(YAWHIP_ERR_INVALID, "yawhip_host_scatter_rows: bad sizes or NULL arrays");
    for (int64_t j = 0; j < n_cols; ++j)
        if (cols[j] < 0 || cols[j] >= row_len) return fail(YAWHIP_ERR_INVALID, "yawhip_host_scatter_rows: column %lld out of range", (long long)cols[j]);
    memset(out, 0, sizeof(double) * (size_t)(n_rows * row_len));
    for (int64_t r = 0; r < n_rows; ++r) {
        double *dst = out + r * row_len;
        const double *src = vals + r * val_row_stride;
        if (col_factor)
            for (int64_t j = 0; j < n_cols; ++j) dst[cols[j]] = src[j * val_col_stride] * col_factor[j];
        else
            for (int64_t j = 0; j < n_cols; ++j) dst[cols[j]] = src[j * val_col_stride];
    }
    return YAWHIP_OK;
}

int yawhip_job_work(yawhip_ctx *ctx, const yawhip_catalog *c1, const yawhip_catalog *c2, int32_t n_jobs, const int32_t *jobs,
                    int32_t n_bins, int32_t n_edges, const double *t, int32_t kernel, int64_t *work) {
    if (!ctx || !work) return fail(YAWHIP_ERR_INVALID, "yawhip_job_work: NULL argument");
    for (int j = 0; j < n_jobs; ++j) work[j] = 0;
    CallState cs;
    return count_enqueue(ctx, c1, c2, n_jobs, jobs, n_bins, n_edges, t, kernel, false, false, work, cs);
}

}  // extern "C"
