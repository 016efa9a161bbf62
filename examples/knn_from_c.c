/* knn_from_c.c -- the C ABI used from plain C (no Python, no torch): evidence sums of a small
 * synthetic chain through mce_knn_dotp_f64, then the same search with distances returned.
 *
 *   gcc -O2 -Iinclude examples/knn_from_c.c -o knn_from_c -Lmcevidence_amd -lmcevidence_hip \
 *       -Wl,-rpath,$PWD/mcevidence_amd -lm
 *   ./knn_from_c            (needs an MI355X)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "mcevidence_hip.h"

static double gauss(unsigned long long *s)
{   /* Box-Muller on a 64-bit LCG: deterministic, no libc rand() */
    double u[2];
    for (int i = 0; i < 2; ++i) {
        *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
        u[i] = ((*s >> 11) + 0.5) / 9007199254740992.0;
    }
    return sqrt(-2.0 * log(u[0])) * cos(6.283185307179586 * u[1]);
}

int main(void)
{
    const int64_t n = 50000;
    const int32_t d = 5, kmax = 4, k0 = 1;
    double *X = malloc(sizeof(double) * n * d), *w = malloc(sizeof(double) * n), *fs = malloc(sizeof(double) * n);
    double *dist = malloc(sizeof(double) * n * (kmax - k0));
    int64_t *idx = malloc(sizeof(int64_t) * n * (kmax - k0));
    unsigned long long seed = 42;
    for (int64_t i = 0; i < n * d; ++i) X[i] = gauss(&seed);
    for (int64_t i = 0; i < n; ++i) { w[i] = 1.0; fs[i] = 0.0; }

    printf("abi %d, devices %d\n", mce_abi_version(), mce_device_count());
    double dotp[4] = {0, 0, 0, 0};
    int rc = mce_knn_dotp_f64(X, n, X, n, d, kmax, k0, 0, w, fs, dotp, NULL, NULL, 0);
    if (rc != MCE_OK) { fprintf(stderr, "mce_knn_dotp_f64: %d %s\n", rc, mce_last_error()); return 1; }
    printf("kernel: %s\n", mce_last_kernel());
    for (int k = k0; k < kmax; ++k) printf("dotp[%d] = %.12g\n", k, dotp[k]);

    rc = mce_knn_f64(X, n, X, n, d, kmax - k0, MCE_SELF_EXCLUDE, 0, dist, idx, 0);
    if (rc != MCE_OK) { fprintf(stderr, "mce_knn_f64: %d %s\n", rc, mce_last_error()); return 1; }
    /* the reduction from the returned distances must reproduce the fused sums */
    const double lnc = 0.5 * d * log(3.14159265358979323846) - lgamma(1.0 + 0.5 * d);
    int bad = 0;
    for (int k = k0; k < kmax; ++k) {
        double s = 0.0;
        for (int64_t j = 0; j < n; ++j) s += exp(lnc + d * log(dist[j * (kmax - k0) + (k - k0)]) - log(w[j]) + fs[j]);
        printf("from distances [%d] = %.12g  (rel diff %.2e)\n", k, s, fabs(s - dotp[k]) / dotp[k]);
        bad |= !(fabs(s - dotp[k]) <= 1e-11 * dotp[k]);
    }
    /* a wrong call is reported, not crashed on */
    rc = mce_knn_f64(X, 10, X, 10, d, 10, MCE_SELF_EXCLUDE, 0, dist, idx, 0);
    printf("K > usable rows -> rc %d: %s\n", rc, mce_last_error());
    bad |= (rc != MCE_ERR_K_RANGE);
    mce_release_device_memory();
    free(X); free(w); free(fs); free(dist); free(idx);
    printf(bad ? "FAILED\n" : "OK\n");
    return bad;
}
