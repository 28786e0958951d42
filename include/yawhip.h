/*
 * yawhip.h -- C ABI of libyawhip.so: MI355X (gfx950) angular pair counting for yet_another_wizz.
 *
 * The reference (jlvdb/yet_another_wizz, pure Python) has no FFI of its own; the seam this library
 * replaces is the per-job loop of
 *     PatchLinkage.count_pairs                       src/yaw/correlation/measurements.py:344-364
 * i.e. for every linked patch pair and every redshift bin one call of
 *     process_patch_pair -> AngularTree.count        measurements.py:88-128, src/yaw/catalog/trees.py:303-362
 *     -> scipy KDTree.count_neighbors                trees.py:348-353
 * The reference-side binding a maintainer would add is shown in INTEGRATION.md (a ctypes stub).
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer is caller-owned host memory unless stated otherwise,
 *     contiguous, 8-byte aligned; the library owns all device memory behind opaque handles;
 *   - one context = one GPU (yawhip_ctx_create) or several GPUs of the node (yawhip_ctx_create_multi) = one
 *     process; calls are blocking and must not be issued concurrently on the same context;
 *   - a context remembers what a count call derives from its inputs on the host (kernel choice, job / threshold tables on the
 *     device) for the next call with the same catalogue pair, job list, thresholds and options: inputs are compared by
 *     content, the item builder and the count kernels run every call;
 *   - every function returns 0 on success or a negative yawhip_status; nothing throws;
 *     yawhip_last_error() returns a thread-local, human readable message for the last failure.
 */
#ifndef YAWHIP_H
#define YAWHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YAWHIP_ABI_VERSION 5

typedef enum yawhip_status {
    YAWHIP_OK = 0,
    YAWHIP_ERR_INVALID = -1,   /* bad argument (NULL handle, negative size, unsorted thresholds ...) */
    YAWHIP_ERR_NO_DEVICE = -2, /* no usable HIP device / device id out of range */
    YAWHIP_ERR_HIP = -3,       /* a HIP runtime call failed (message has the HIP error string) */
    YAWHIP_ERR_OOM = -4,       /* device or host allocation failed */
    YAWHIP_ERR_MISMATCH = -5   /* catalogs do not fit together (patch count, bin count, context) */
} yawhip_status;

/* Which device code path counts the pairs. All of them return identical results. */
typedef enum yawhip_kernel {
    YAWHIP_KERNEL_AUTO = 0,   /* library picks the fastest exact path: BAND on every strip layout of unit vectors (its float32
                                 kernels; the float64 band kernel -- band_fp32 = 0, more than four edges off a log grid --
                                 only where the streamed runs are dense), SWEEP on layouts without strips, EXACT for
                                 input that is not unit vectors (stats->kernel_used / band_variant tell) */
    YAWHIP_KERNEL_EXACT = 1,  /* plain FP64 brute force over every candidate pair */
    YAWHIP_KERNEL_FILTER = 2, /* FP32 guard-banded pre-filter, FP64 re-evaluation of survivors */
    YAWHIP_KERNEL_SWEEP = 3,  /* FILTER + sorted-axis sweep that skips far-away tile pairs */
    YAWHIP_KERNEL_BAND = 4    /* sorted-axis windows as SWEEP, then every lane object walks only its own band |du| <= r of the
                                 window: classified in float32 against guard bands around the edges, the evaluations inside a
                                 guard band decided by the exact FP64 predicate (band_fp32 = 1, default; results identical),
                                 or every entry in FP64 (band_fp32 = 0) (ABI >= 3) */
} yawhip_kernel;

typedef struct yawhip_ctx yawhip_ctx;
typedef struct yawhip_catalog yawhip_catalog;

/* Filled by yawhip_count_pairs (may be NULL). Times are milliseconds. */
typedef struct yawhip_stats {
    int64_t candidate_pairs;   /* sum over (job, bin) of N1(p,k) * N2(q,k): the brute-force work unit   */
    int64_t evaluated_pairs;   /* pair distances the launched kernels actually evaluated (<= candidates
                                  when tile culling is active, padded lanes not included)               */
    int64_t algorithmic_bytes; /* compulsory HBM bytes: per job, every object of both patches once       */
    int64_t n_workgroups;      /* workgroups of the dominant (count) kernel                              */
    int32_t n_launches;        /* kernel launches in this call                                           */
    int32_t kernel_used;       /* yawhip_kernel actually run                                             */
    double kernel_ms;          /* HIP-event time of all launches of the call (item builder, count kernel,
                                  reduction) on the context's stream                                     */
    double total_ms;           /* host wall time of the whole call (job upload, kernels, result download)*/
    double count_ms;           /* HIP-event time of the count kernel(s) alone (ABI >= 2)                 */
    int32_t layout_mode;       /* which device layouts the items came from (ABI >= 3): 0 = (patch, bin, u) segments,
                                  1 = (patch, strip) runs with all bins merged (binned x unbinned), 3 = (patch, bin,
                                  strip) runs (binned x binned, dense catalogues)                       */
    int32_t n_orientations;    /* strip layouts: how many of the three orientations the jobs used (ABI >= 3) */
    int64_t exact_reevaluations; /* band kernel, float32 classification: evaluations that fell into a guard band of an edge
                                  and were decided by the exact float64 predicate (ABI >= 4)                  */
    int32_t band_variant;      /* which band kernel ran (ABI >= 4): 0 none, 64 every entry in float64, 32 float32 classes +
                                  exact guard bands, 33 the same for fine log-spaced radial grids             */
    int32_t merged_triples;       /* 1: the float32 band kernel streamed merged triple runs (one window per item), else 0 */
} yawhip_stats;

const char *yawhip_last_error(void);
int yawhip_abi_version(void);

/* Number of visible HIP devices. */
int yawhip_device_count(int *n);

/* Create / destroy a context on device `device_id` (creates one HIP stream). */
int yawhip_ctx_create(int device_id, yawhip_ctx **out);
int yawhip_ctx_destroy(yawhip_ctx *ctx);
/* One context over several GPUs of the node (ABI >= 3; replaces the worker pool behind `max_workers` of
 * PatchLinkage.count_pairs, src/yaw/correlation/measurements.py:344-350, src/yaw/utils/parallel.py:251-346):
 * catalogues uploaded to it are replicated on every device, yawhip_count_pairs splits its job list over the devices
 * (longest job first, by the work the item builder reports) and returns the complete result; options apply to all
 * devices. A device id may be listed more than once (several streams on one GPU; used by the tests). Everything
 * else behaves as for a single-device context; yawhip_assign_patches and yawhip_job_work use the first device. */
int yawhip_ctx_create_multi(const int *device_ids, int n_devices, yawhip_ctx **out);
/* Number of devices a context spans. */
int yawhip_ctx_device_count(const yawhip_ctx *ctx, int *n);

/* Tunables (all optional):
 *   "tile_r"            objects per lane (0 = auto, 1, 2, 4)
 *   "band_batch_log2"   band kernel: log2 of the consecutive items a workgroup takes per visit, carrying its unweighted
 *                       histogram while they add to the same output slot (-1 = default: 1 item; larger batches unbalance clustered data)
 *   "hist_copies_log2"  band kernel: log2 of the copies of its LDS histogram (-1 = auto: 4 copies for few slots, up to 16
 *                       for per-bin items or when neighbouring objects of a binned catalogue mostly share their bin; 0..6)
 *   "band_cap"          entries per LDS stage of the band kernel (0 = auto; 192 / 288: float64 and fine-grid kernels;
 *                       320 / 512: float32 kernel -- the larger one when a lane tile's window is expected to need it)
 *   "kernel"            default yawhip_kernel of yawhip_count_pairs(kernel = AUTO)
 *   "strip_width_micro" spacing, in 1e-6 chord units, of the strip grid of catalogues uploaded afterwards
 *                       (0 = no strips, default 5000). Catalogues counted against each other should share it;
 *                       otherwise the cross-correlation path falls back to ordinary (job, bin) items.
 *   "seg_strips"        binned x binned counts use the per-(patch, bin) strip layouts of dense catalogues (default 1)
 *   "seg_strips_min_run" mean objects per (patch, bin, strip) run of the lane-side catalogue from which they are used (default 16)
 *   "debug_no_hits"     diagnostics: the pre-filter rejects everything (times the filter alone; wrong counts)
 *   "auto_orient"       1 (default): every job runs on the strip layouts of the orientation (sort axis u, strips along v,
 *                       dropped axis w) that suits its two patches -- w pointing at them; layouts of further
 *                       orientations are built on first use. 0: the sort axis the catalogues were uploaded with
 *   "slab_budget_bytes" weighted calls keep a slab of partial sums per potential work item; a job list that would need more
 *                       than this many bytes (default 2^30) is counted in pieces, one after the other (same results)
 *   "band_fp32"         1 (default): on strip layouts of unit vectors the band kernel classifies every evaluation in float32
 *                       and decides the ones inside a guard band of an edge with the exact float64 predicate (same results);
 *                       0: every evaluation in float64
 *   "triple_runs"       float32 band kernels: the streamed side is read from MERGED runs of three neighbouring strips (one
 *                       window per work item instead of three) when the strip grid is as wide as the largest separation --
 *                       1 (default): where the merged window still fits one LDS stage, 2: always, 0: never (same results)
 *   "half_bands"        1 (default): a catalogue counted against ITSELF on merged triple runs with one object per lane takes every
 *                       unordered pair of a diagonal job once and counts it twice (half the walk of DD / RR of an
 *                       autocorrelation; an exact doubling, also of weighted sums); 0: both sides walk their full bands
 *   "item_segments"     1 (default): the strip builder keeps its work items in eight segments, one per XCD, each with its
 *                       own append counter; 0: one list, dealt to the XCDs in blocks (same results)
 *   "band_grid_div"     band kernels: workgroups = potential work items / this (1..64; default 0 = auto: 8, 16 for the per-bin
 *                       items of binned x binned counts, 4 on clustered catalogues); the kernel loops over the rest
 *   "spin_wait"         1 (default): the host waits for a call's results by polling the stream for the first 2 ms, then blocks;
 *                       0: it blocks at once
 *   "flush_stages_log2" band kernel: the 32-bit LDS counters of an item are flushed to the 64-bit result every
 *                       2^value stages (default 17: 128 lane objects x 192 entries x 2^17 < 2^32; tests lower it) */
int yawhip_ctx_set_option(yawhip_ctx *ctx, const char *key, int64_t value);

/*
 * Upload one catalogue (replaces: Catalog.build_trees + the per-job pickle.load of trees.pkl,
 * catalog.py:1406-1461, trees.py:365-429,597).
 *   n            objects kept (objects outside the redshift binning are already dropped, trees.py:414)
 *   x,y,z        unit vectors, float64[n], exactly the host's AngularCoordinates.to_3d() values
 *   w            float64[n] weights or NULL (unweighted)
 *   n_patches    P
 *   n_bins_or_1  B for a catalogue binned in redshift, 1 for an unbinned one (single tree re-used
 *                for every bin, trees.py:600-601)
 *   offsets      int64[P * n_bins_or_1 + 1], CSR over (patch, bin) segments; objects are sorted by
 *                (patch, bin); offsets[0] == 0, offsets[last] == n, non-decreasing
 * Order of objects inside a segment is free (it only permutes floating point summation order).
 */
int yawhip_catalog_upload(yawhip_ctx *ctx, int64_t n, const double *x, const double *y, const double *z,
                          const double *w, int32_t n_patches, int32_t n_bins_or_1, const int64_t *offsets,
                          yawhip_catalog **out);
/* Same, choosing the coordinate (0 = x, 1 = y, 2 = z) along which the library keeps each segment sorted
 * for its window culling; yawhip_catalog_upload uses z. Pick the axis most perpendicular to the survey
 * footprint (a footprint around a pole is flat in z and culls badly along it). Two catalogues counted
 * against each other should use the same axis, otherwise the culling is skipped (results unchanged). */
int yawhip_catalog_upload_axis(yawhip_ctx *ctx, int64_t n, const double *x, const double *y, const double *z,
                               const double *w, int32_t n_patches, int32_t n_bins_or_1, const int64_t *offsets,
                               int32_t sort_axis, yawhip_catalog **out);
int yawhip_catalog_sort_axis(const yawhip_catalog *cat, int32_t *axis);
int yawhip_catalog_free(yawhip_catalog *cat);
/* Device bytes held by a catalogue (for memory accounting). */
int yawhip_catalog_device_bytes(const yawhip_catalog *cat, int64_t *bytes);

/*
 * Count pairs for a list of jobs (replaces measurements.py:344-364 up to, not including, the
 * scatter into [S,B,P,P] and the rweight / scale recombination of trees.py:358-362: this entry point
 * returns the per-job fine values; yawhip_count_pairs_dense below does the epilogue as well -- the
 * recombination of several fine bins on the device, the scatter on the host).
 *   c1, c2     catalogues on the same context with equal n_patches; c1 == c2 is allowed (DD / RR
 *              of an autocorrelation): every ordered pair a != b is then counted, self pairs have
 *              s == 0 and never fall above an edge, exactly as in the reference
 *   jobs       int32[n_jobs][2] = (patch id in c1, patch id in c2)
 *   n_bins     B; a catalogue uploaded with n_bins_or_1 == 1 uses its single segment for every bin,
 *              otherwise its n_bins_or_1 must equal n_bins
 *   n_edges    E >= 2 thresholds per bin
 *   t          float64[B][E], ascending in E: t[k][e] = pow(2 sin(ang_bins[k][e] / 2), 2.0) computed
 *              on the host (trees.py:107-117, coordinates.py:277)
 *   kernel     yawhip_kernel
 * Pair (a in c1 segment (p,k), b in c2 segment (q,k)) belongs to fine bin e (0 <= e < E-1) iff
 *       t[k][e] < s <= t[k][e+1],   s = ((ax-bx)^2 + (ay-by)^2) + (az-bz)^2   in float64 without FMA.
 * Outputs (either may be NULL; on success every element is written, the caller need not clear them):
 *   fine_counts  int64[n_jobs][B][E-1]  number of pairs            (bit exact)
 *   fine_sums    float64[n_jobs][B][E-1] sum of w_a * w_b, a missing weight column counts as 1.0.
 *                No floating point atomics touch global memory: per-item partial sums are combined in a fixed
 *                order. BAND and SWEEP add into a histogram that one wave owns (LDS float64 adds in program
 *                order; adds of ONE instruction that hit the same cell are assumed to be serialised by the LDS in a
 *                fixed lane order, as observed on MI355X): bit-reproducible from run to run for a given build and tile_r. EXACT / FILTER keep
 *                per-lane private histograms, reduced in a fixed order, while (E-1) * 2 KiB fits LDS (E-1 <= 78);
 *                beyond that they share one LDS histogram between four waves, whose float64 adds are ordered
 *                by the hardware: sums then agree only to rounding (1e-10 relative is what the tests ask)
 * When both catalogues are unweighted fine_sums, if requested, is the exact conversion of fine_counts.
 */
int yawhip_count_pairs(yawhip_ctx *ctx, const yawhip_catalog *c1, const yawhip_catalog *c2, int32_t n_jobs,
                       const int32_t *jobs, int32_t n_bins, int32_t n_edges, const double *t, int32_t kernel,
                       int64_t *fine_counts, double *fine_sums, yawhip_stats *stats);

/*
 * The same count, returned as the result tensor of PatchLinkage.count_pairs (ABI >= 4; replaces measurements.py:344-364
 * INCLUDING the host epilogue: the separation weights and per-scale recombination of trees.py:358-362,134-160, the x 0.5
 * of the diagonal jobs of an autocorrelation, measurements.py:362-363, and the scatter of every job into its slot):
 *   n_scales        S
 *   slices          int32[B][S][2]: scale s of bin k sums the fine bins [first, last) of that bin (the indices of the edges
 *                   nearest to the scale's limits, trees.py:134-160)
 *   fine_factors    float64[B][E-1] multiplied into the fine bins before they are summed (separation weights,
 *                   trees.py:358-360), or NULL
 *   halve_diagonal  non-zero: jobs with equal patch ids count x 0.5 (autocorrelation)
 *   dense           float64[S][B][P][P], P = the catalogues' patch count; slot [s][k][i][j] of job (i, j), 0 elsewhere
 * The weighted-sum reproducibility note of yawhip_count_pairs applies; it further assumes that the LDS serialises float64
 * adds of ONE instruction to the same address in a fixed lane order (observed on MI355X; tested run to run).
 */
int yawhip_count_pairs_dense(yawhip_ctx *ctx, const yawhip_catalog *c1, const yawhip_catalog *c2, int32_t n_jobs,
                             const int32_t *jobs, int32_t n_bins, int32_t n_edges, const double *t, int32_t kernel,
                             int32_t n_scales, const int32_t *slices, const double *fine_factors, int32_t halve_diagonal,
                             double *dense, yawhip_stats *stats);

/*
 * Several counts of ONE measurement from one call (ABI >= 5): the reference's crosscorrelate issues DD, DR, RD, RR back to
 * back on one linkage (src/yaw/correlation/measurements.py:617-628), autocorrelate DD, DR, RR (:517-523) -- same binning,
 * thresholds and recombination, other catalogue pairs and job lists. All requests are put on the context's stream at once
 * (up to four in flight, each with its own work items and result block): the host prepares count k + 1 and writes the
 * tensor of count k while the device counts. Results are those of n_requests calls of yawhip_count_pairs_dense, bit for bit.
 *   requests   n_requests records; `dense` and `stats` of every record are written (stats may be NULL)
 * A context of several devices counts the requests one after the other (each split over the devices).
 */
typedef struct yawhip_dense_request {
    const yawhip_catalog *c1, *c2; /* catalogue pair of this count (c1 == c2: DD / RR of an autocorrelation)      */
    int32_t n_jobs;                /* linked patch pairs of this count                                            */
    int32_t halve_diagonal;        /* non-zero: jobs with equal patch ids count x 0.5 (autocorrelation)           */
    const int32_t *jobs;           /* int32[n_jobs][2]                                                            */
    double *dense;                 /* out: float64[S][B][P][P]                                                    */
    yawhip_stats *stats;           /* out, may be NULL                                                            */
} yawhip_dense_request;
int yawhip_count_pairs_dense_batch(yawhip_ctx *ctx, int32_t n_requests, const yawhip_dense_request *requests, int32_t n_bins,
                                   int32_t n_edges, const double *t, int32_t kernel, int32_t n_scales, const int32_t *slices,
                                   const double *fine_factors);

/*
 * The count of ONE rank of a job list sharded over processes (one process per GPU; ABI >= 4), left on the device for the
 * final reduce (replaces the result messages of the reference's task farm, src/yaw/utils/parallel.py:251-346):
 *   jobs, n_jobs    this rank's share
 *   n_rows_total    jobs of the whole list
 *   row_index       int32[n_jobs]: position of each of this rank's jobs in the whole list
 *   device_rows     out: DEVICE pointer to float64[n_rows_total * B * (E-1) + 1], owned by the context and valid until its
 *                   next call: this rank's rows in place (weighted sums, or the exact conversion of the counts), zero
 *                   elsewhere -- every row is non-zero on one rank only, so a sum all-reduce (RCCL) over the ranks yields the
 *                   complete tensor, exactly. The last element is zero: callers use it as a status flag in the reduce.
 * The context's stream has been waited for when the call returns. Single-device contexts only.
 */
int yawhip_count_pairs_rows_device(yawhip_ctx *ctx, const yawhip_catalog *c1, const yawhip_catalog *c2, int32_t n_jobs,
                                   const int32_t *jobs, int32_t n_bins, int32_t n_edges, const double *t, int32_t kernel,
                                   int64_t n_rows_total, const int32_t *row_index, double **device_rows, yawhip_stats *stats);

/*
 * Nearest patch centre of n objects in Euclidean xyz (replaces scipy.cluster.vq.vq in assign_patch_centers,
 * catalog/catalog.py:229-249, same arithmetic: identical ids including ties, first minimum wins).
 *   x,y,z        float64[n] unit vectors (host)
 *   centers_xyz  float64[n_centers][3] (host, row-major)
 *   patch_out    int32[n] (host)
 */
int yawhip_assign_patches(yawhip_ctx *ctx, int64_t n, const double *x, const double *y, const double *z,
                          int32_t n_centers, const double *centers_xyz, int32_t *patch_out);

/*
 * Host-only helper of the ingest path (no device, no context): stable grouping of float64 columns by an integer key --
 * what the reference does per chunk with groupby(patch_ids, chunk) (catalog/catalog.py:293, utils/misc.py:40-51) and
 * groupby(bin_idx, chunk) (catalog/trees.py:413), i.e. np.argsort(kind="stable") + a gather per column, here as one
 * threaded counting sort. Entries with key < 0 are dropped (objects outside the binning, trees.py:414).
 *   keys       int32[n] or int64[n] (key_bytes = 4 | 8), every key < num_groups
 *   in, out    n_cols pointers to float64[n] each; out[c] receives the kept entries of in[c], group after group, input
 *              order inside a group; out[c] must not alias in[c]
 *   sizes      int64[num_groups] entries per group (out)
 *   n_threads  0 = one per core, at most 16
 */
int yawhip_host_group_columns(int64_t n, const void *keys, int32_t key_bytes, int64_t num_groups, int32_t n_cols,
                              const double *const *in, double *const *out, int64_t *sizes, int32_t n_threads);

/*
 * Host-only helper of the epilogue: the dense result tensor from the per-job values -- the loop
 * counts[:, id1, id2] = result (x 0.5 on the diagonal of an autocorrelation) of PatchLinkage.count_pairs
 * (correlation/measurements.py:358-364), for all scales and bins at once.
 *   out         float64[n_rows][row_len], zero-filled, then out[r][cols[j]] = vals[r][j] * (col_factor ? col_factor[j] : 1)
 *               (n_rows = scales x bins, row_len = P x P, cols[j] = id1 * P + id2 of job j)
 *   vals        float64, element (r, j) at vals[r * val_row_stride + j * val_col_stride] (strides in elements: the per-job
 *               values arrive job-major from yawhip_count_pairs)
 */
int yawhip_host_scatter_rows(int64_t n_rows, int64_t row_len, double *out, int64_t n_cols, const int64_t *cols,
                             const double *vals, int64_t val_row_stride, int64_t val_col_stride, const double *col_factor);

/*
 * Evaluated pair distances per job, without counting anything: runs the item builder of yawhip_count_pairs for the
 * same arguments and sums lane-tile x window sizes per job (for the brute-force kernels that is N1*N2 per bin).
 * This is the cost the host balances when it shards the job list over GPUs (replaces the "largest jobs first"
 * scheduling heuristic of measurements.py:262-273). work: int64[n_jobs].
 */
int yawhip_job_work(yawhip_ctx *ctx, const yawhip_catalog *c1, const yawhip_catalog *c2, int32_t n_jobs,
                    const int32_t *jobs, int32_t n_bins, int32_t n_edges, const double *t, int32_t kernel,
                    int64_t *work);

#ifdef __cplusplus
}
#endif
#endif /* YAWHIP_H */
