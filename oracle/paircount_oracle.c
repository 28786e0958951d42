/*
 * oracle/paircount_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, brute force, O(N1*N2)) of the reference's per-tree pair count
 *   yaw.catalog.trees.AngularTree.count            /root/reference/src/yaw/catalog/trees.py:303-362
 *   -> scipy.spatial.KDTree.count_neighbors        call site trees.py:348-353 (third party, scipy 1.15.3,
 *      compiled; semantics pinned empirically in SURVEY.md section 8(a11))
 * and of the per-job loop over redshift bins
 *   yaw.correlation.measurements.process_patch_pair /root/reference/src/yaw/correlation/measurements.py:88-128
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (yet_another_wizz_amd + libyawhip.so) never links or calls it.
 *
 * Inclusion predicate (bit-exact contract):
 *     s = ((ax-bx)*(ax-bx) + (ay-by)*(ay-by)) + (az-bz)*(az-bz)      float64, every product and sum
 *                                                                     rounded separately (NO FMA)
 *     pair lies inside edge j  <=>  s <= t[j],  t[j] = pow(2 sin(ang_bins[j]/2), 2.0) computed by the caller
 * Fine bin j (0 <= j < E-1) collects pairs with t[j] < s <= t[j+1]  (trees.py:127-131 dispatch_counts:
 * cumulative -> np.diff, non-cumulative -> drop first element; both give this).
 * Self pairs (s == 0) are never > t[0] >= 0, hence never counted -- same as the reference.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no -ffast-math).
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#if defined(__FMA__) && !defined(YAW_ORACLE_ALLOW_FMA_TARGET)
/* contraction is disabled by -ffp-contract=off in the Makefile; this is a belt-and-braces reminder */
#endif

/* One tree-vs-tree count.  out_counts[E-1] (int64) is always filled with the unweighted number of
 * pairs per fine bin; out_sums[E-1] (double) with sum of w1*w2 (a missing weight array counts as 1.0,
 * matching scipy's weights=(None, w) handling). Either output pointer may be NULL. */
void yaw_oracle_count_tree(int64_t n1, const double *x1, const double *y1, const double *z1, const double *w1,
                           int64_t n2, const double *x2, const double *y2, const double *z2, const double *w2,
                           int n_edges, const double *t, int64_t *out_counts, double *out_sums)
{
    const int nf = n_edges - 1;
    if (out_counts) memset(out_counts, 0, sizeof(int64_t) * (size_t)(nf > 0 ? nf : 0));
    if (out_sums)   memset(out_sums, 0, sizeof(double) * (size_t)(nf > 0 ? nf : 0));
    if (nf <= 0) return;
    const double t_lo = t[0], t_hi = t[n_edges - 1];
    for (int64_t a = 0; a < n1; ++a) {
        const double ax = x1[a], ay = y1[a], az = z1[a];
        const double wa = w1 ? w1[a] : 1.0;
        for (int64_t b = 0; b < n2; ++b) {
            const double dx = ax - x2[b];
            const double dy = ay - y2[b];
            const double dz = az - z2[b];
            const double xx = dx * dx;
            const double yy = dy * dy;
            const double zz = dz * dz;
            const double sxy = xx + yy;
            const double s = sxy + zz;
            if (s <= t_hi && s > t_lo) {
                int j = 0;                       /* find j with t[j] < s <= t[j+1] */
                while (s > t[j + 1]) ++j;
                if (out_counts) out_counts[j] += 1;
                if (out_sums)   out_sums[j] += wa * (w2 ? w2[b] : 1.0);
            }
        }
    }
}

/* Job-level restatement (measurements.py:88-128 + :344-364 without the scatter):
 * catalogs are SoA, pre-sorted by (patch, z-bin) with CSR offsets off[c][patch*nb + bin];
 * nb == 1 means "unbinned": the single tree is re-used for every z-bin (trees.py:600-601).
 * jobs[j] = (patch id in catalog 1, patch id in catalog 2); t = double[n_bins][n_edges].
 * Outputs are [n_jobs][n_bins][n_edges-1].
 * Threads (OpenMP, when compiled with it): every (job, bin) slot is cut into blocks of ROWS_PER_ITEM
 * catalogue-1 rows; blocks are counted independently into a slab and the slabs of a slot are summed
 * in block order afterwards, so the result does not depend on the number of threads. */
#include <stdlib.h>
#define ROWS_PER_ITEM 2048

#ifdef _OPENMP
#include <omp.h>
#endif

int yaw_oracle_count_jobs(const double *x1, const double *y1, const double *z1, const double *w1,
                          int nb1, const int64_t *off1,
                          const double *x2, const double *y2, const double *z2, const double *w2,
                          int nb2, const int64_t *off2,
                          int n_jobs, const int32_t *jobs, int n_bins, int n_edges, const double *t,
                          int num_threads, int64_t *out_counts, double *out_sums)
{
    const int nf = n_edges - 1;
    const int64_t n_slots = (int64_t)n_jobs * n_bins;
    if (nf <= 0 || n_slots <= 0) return 0;
    int64_t *first = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_slots + 1));
    if (!first) return -1;
    int64_t n_items = 0;
    for (int64_t s = 0; s < n_slots; ++s) {
        const int j = (int)(s / n_bins), k = (int)(s % n_bins);
        const int p = jobs[2 * j], k1 = nb1 == 1 ? 0 : k;
        const int64_t n1 = off1[(int64_t)p * nb1 + k1 + 1] - off1[(int64_t)p * nb1 + k1];
        first[s] = n_items;
        n_items += (n1 + ROWS_PER_ITEM - 1) / ROWS_PER_ITEM;
    }
    first[n_slots] = n_items;
    int64_t *slot_of = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_items + 1));
    int64_t *pc = (int64_t *)calloc((size_t)(n_items + 1) * nf, sizeof(int64_t));
    double *ps = (double *)calloc((size_t)(n_items + 1) * nf, sizeof(double));
    if (!slot_of || !pc || !ps) { free(first); free(slot_of); free(pc); free(ps); return -1; }
    for (int64_t s = 0; s < n_slots; ++s)
        for (int64_t it = first[s]; it < first[s + 1]; ++it) slot_of[it] = s;
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int64_t it = 0; it < n_items; ++it) {
        const int64_t s = slot_of[it];
        const int j = (int)(s / n_bins), k = (int)(s % n_bins);
        const int p = jobs[2 * j], q = jobs[2 * j + 1];
        const int k1 = nb1 == 1 ? 0 : k, k2 = nb2 == 1 ? 0 : k;
        const int64_t seg0 = off1[(int64_t)p * nb1 + k1], seg1 = off1[(int64_t)p * nb1 + k1 + 1];
        const int64_t a0 = seg0 + (it - first[s]) * ROWS_PER_ITEM;
        const int64_t a1 = a0 + ROWS_PER_ITEM < seg1 ? a0 + ROWS_PER_ITEM : seg1;
        const int64_t b0 = off2[(int64_t)q * nb2 + k2], b1 = off2[(int64_t)q * nb2 + k2 + 1];
        yaw_oracle_count_tree(a1 - a0, x1 + a0, y1 + a0, z1 + a0, w1 ? w1 + a0 : NULL,
                              b1 - b0, x2 + b0, y2 + b0, z2 + b0, w2 ? w2 + b0 : NULL,
                              n_edges, t + (size_t)k * n_edges, pc + (size_t)it * nf, ps + (size_t)it * nf);
    }
    for (int64_t s = 0; s < n_slots; ++s)
        for (int e = 0; e < nf; ++e) {
            int64_t c = 0;
            double v = 0.0;
            for (int64_t it = first[s]; it < first[s + 1]; ++it) {
                c += pc[(size_t)it * nf + e];
                v += ps[(size_t)it * nf + e];
            }
            if (out_counts) out_counts[(size_t)s * nf + e] = c;
            if (out_sums) out_sums[(size_t)s * nf + e] = v;
        }
    free(first); free(slot_of); free(pc); free(ps);
    return 0;
}

int yaw_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
