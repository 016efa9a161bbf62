/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Exact brute-force k-nearest-neighbour search and the volume/weight
 * reduction of the reference hot path, in plain C:
 *   - oracle_knn_f64  restates what `nbrs.kneighbors(samples)` returns at
 *     /root/reference/MCEvidence.py:1104 (K smallest Euclidean distances per
 *     query, ascending), computed by direct differences d2 = sum (x_i-y_i)^2.
 *     The third-party routine the reference actually calls is scikit-learn
 *     NearestNeighbors (KDTree / ArgKmin); its published contract is
 *     "K smallest distances, sorted ascending", which is what is restated.
 *   - oracle_dotp_f64 restates MCEvidence.py:1107-1117 in the log domain.
 * Pinned against reference outputs by tests/test_oracle_golden.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define KMAX_ORACLE 64

/* self_mode: 0 plain; 1 plain (Y==X, self included as sklearn does);
 *            2 skip reference row (self_offset + q). */
int oracle_knn_f64(const double *X, int64_t nq, const double *Y, int64_t nr, int32_t d, int32_t K,
                   int32_t self_mode, int64_t self_offset, double *dist, int64_t *idx, int32_t nthreads)
{
    if (K < 1 || K > KMAX_ORACLE || d < 1) return -1;
    if ((self_mode == 2 ? nr - 1 : nr) < K) return -2;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t q = 0; q < nq; ++q) {
        double bd[KMAX_ORACLE];
        int64_t bi[KMAX_ORACLE];
        int n = 0;
        const double *x = X + q * (int64_t)d;
        const int64_t skip = (self_mode == 2) ? self_offset + q : -1;
        for (int64_t j = 0; j < nr; ++j) {
            if (j == skip) continue;
            const double *y = Y + j * (int64_t)d;
            double s = 0.0;
            for (int i = 0; i < d; ++i) { double t = x[i] - y[i]; s += t * t; }
            if (n < K) {
                int p = n++;
                while (p > 0 && bd[p - 1] > s) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; --p; }
                bd[p] = s; bi[p] = j;
            } else if (s < bd[K - 1]) {
                int p = K - 1;
                while (p > 0 && bd[p - 1] > s) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; --p; }
                bd[p] = s; bi[p] = j;
            }
        }
        for (int k = 0; k < K; ++k) { dist[q * K + k] = sqrt(bd[k]); if (idx) idx[q * K + k] = bi[k]; }
    }
    return 0;
}

/* dotp[k] = sum_j sign(w_j) exp(lnC_D + D ln r_jk - ln |w_j| + fs_j), k in [k0,kmax); serial, row order
   (= volume/weight * exp(fs) of MCEvidence.py:1107-1117, also for a negative weight). */
int oracle_dotp_f64(const double *dist, int64_t nq, int32_t ld, int32_t k0, int32_t kmax, int32_t d,
                    const double *w, const double *fs, double *dotp)
{
    if (kmax > ld || k0 < 0 || k0 > kmax) return -1;
    const double lnc = 0.5 * d * log(M_PI) - lgamma(1.0 + 0.5 * d);
    for (int k = k0; k < kmax; ++k) {
        double s = 0.0;
        for (int64_t j = 0; j < nq; ++j) {
            double r = dist[j * ld + k];
            s += (w[j] < 0 ? -1.0 : 1.0) * exp(lnc + d * log(r) - log(fabs(w[j])) + fs[j]);
        }
        dotp[k] = s;
    }
    return 0;
}
