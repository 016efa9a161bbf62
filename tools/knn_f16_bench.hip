// knn_f16_bench.hip -- developer microbench for knn_f16_kernel (ablation/tuning); not part of the product.
#include "../mcevidence_amd/csrc/f16_prep.hpp"
#include "../mcevidence_amd/csrc/pack_refs.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#ifndef DIM
#define DIM 27
#endif
#ifndef BKCAP
#define BKCAP 12
#endif
#ifndef BKSEL
#define BKSEL 9
#endif
#ifndef BQT
#define BQT 2      // query tiles per wave (4: the wide form of the exhaustive sweep)
#endif
using namespace mce;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 200000;
    const int rsplit = argc > 2 ? atoi(argv[2]) : 1;
    const int reps = argc > 3 ? atoi(argv[3]) : 2;
    const int seed_rows = argc > 4 ? atoi(argv[4]) : MCE_H_SEED_ROWS;      // seed phase: rows, tiles per group
    const int seed_tg = argc > 5 ? atoi(argv[5]) : MCE_H_SEED_TG;
    constexpr int D = DIM;
    constexpr int KST = f16_ksteps(D);
    constexpr int CT = f16_chunk_tiles(KST);
    const int qpb = kHWaves * BQT * 32;
    const int nqblk = (int)((n + qpb - 1) / qpb);
    const int64_t nq_pad = (int64_t)nqblk * qpb;
    const int64_t nchunk = (n + CT * 32 - 1) / (CT * 32);
    const int64_t nrow_pad = nchunk * CT * 32;
    std::vector<double> h((size_t)n * D);
    std::mt19937_64 g(1); std::normal_distribution<double> nd;
    for (auto& v : h) v = nd(g);
    double *X, *pd, *center, *msum, *params, *qinfo; int* pi; _Float16 *Yh, *Xh;
    CK(hipMalloc(&X, sizeof(double) * n * D));
    CK(hipMalloc(&Yh, 2 * nrow_pad * 16 * KST)); CK(hipMalloc(&Xh, 2 * nq_pad * 16 * KST));
    CK(hipMalloc(&qinfo, 16 * nq_pad)); const size_t pbytes = 256 + (size_t)nqblk * rsplit * 8 * 64; CK(hipMalloc(&params, pbytes)); CK(hipMalloc(&center, 3 * 512)); CK(hipMalloc(&msum, 8 * kStatStride * 256));
    const size_t nl = (size_t)rsplit * BKCAP * nq_pad;
    CK(hipMalloc(&pd, sizeof(double) * nl)); CK(hipMalloc(&pi, sizeof(int) * nl));
    CK(hipMemcpy(X, h.data(), sizeof(double) * n * D, hipMemcpyHostToDevice));
    col_stats_partial_kernel<<<kMeanBlocks, kMeanThreads>>>(X, n, D, msum);
    col_stats_final_kernel<<<1, 64>>>(msum, n, D, center, center + 64);
    CK(hipMemset(params, 0, pbytes));
    f16_scale_kernel<<<1, 64>>>(center + 64, nullptr, params);
    const int64_t rpb = 4 * (64 / (2 * KST));
    f16_pack_refs_kernel<<<(unsigned)std::min<int64_t>((nrow_pad + rpb - 1) / rpb, 2048), 256>>>(X, n, D, KST, nrow_pad, center, params, Yh);
    f16_pack_queries_kernel<<<(unsigned)((nq_pad + rpb - 1) / rpb), 256>>>(X, n, nq_pad, D, KST, center, params, Xh, qinfo);
    CK(hipDeviceSynchronize());
    constexpr size_t LDS = f16_lds_bytes(KST, BKCAP, false, BQT);
    auto kern = knn_f16_kernel<KST, BKCAP, false, false, 0, BKCAP, BQT>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        kern<<<nqblk * rsplit, kHThreads, LDS>>>(Yh, nchunk, rsplit, Xh, qinfo, params, X, X, n, n, D, nq_pad, nqblk, 1, 0, BKSEL, pd, pi,
                                                (const int*)nullptr, (const float*)nullptr, 0, (const int*)nullptr, (const int*)nullptr, (const float*)nullptr,
                                                (const float*)nullptr, (const float*)nullptr, 0, 1, (const int*)nullptr, (const double*)nullptr, (const int*)nullptr,
                                                f16_seed_cfg((nchunk + rsplit - 1) / rsplit, CT, BKSEL + 1, seed_rows, MCE_H_SEED_SHARE, seed_tg), SymParams(), (float*)nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("D=%d KST=%d BKCAP=%d K=%d QT=%d CT=%d ablate=%d lds=%zu n=%lld rsplit=%d grid=%d: %.2f ms  %.3f Mq/s  %.1f TF(f16 flops)\n", D, KST, BKCAP, BKSEL, BQT, CT, MCE_ABLATE, LDS,
               (long long)n, rsplit, nqblk * rsplit, ms, n / ms / 1e3, (double)n * n * 32.0 * KST / ms / 1e9);
    }
    {   // sanity of the lists: K-th entries finite, and a checksum to compare builds / seed settings
        std::vector<double> hk((size_t)nq_pad);
        long long bad = 0; double sum = 0.0;
        for (int sp = 0; sp < rsplit; ++sp) {
            CK(hipMemcpy(hk.data(), pd + ((size_t)sp * BKCAP + (BKSEL - 1)) * nq_pad, sizeof(double) * nq_pad, hipMemcpyDeviceToHost));
            for (int64_t q = 0; q < n; ++q) { if (!(hk[q] < 1e300)) ++bad; else sum += hk[q]; }
        }
        printf("K-th entries: %lld not finite, checksum %.17g\n", bad, sum);
    }
    return 0;
}
