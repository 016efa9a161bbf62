// knn_bench.hip -- developer microbench for knn_mfma_kernel (ablation + tuning); not part of the product.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DMCE_ABLATE=n] [-DMCE_CHUNK_KSTEPS=n] -DKS=7 -DKCAP=12 tools/knn_bench.hip -o tools/knn_bench
#include "../mcevidence_amd/csrc/knn_mfma.hpp"
#include "../mcevidence_amd/csrc/pack_refs.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#ifndef KS
#define KS 7
#endif
#ifndef KCAP
#define KCAP 12
#endif
#ifndef KSEL
#define KSEL KCAP
#endif
using namespace mce;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 200000;
    const int rsplit = argc > 2 ? atoi(argv[2]) : 1;
    const int reps = argc > 3 ? atoi(argv[3]) : 2;
    const int D = 4 * KS - 1;
    constexpr int QT = kQT;
    constexpr int CT = chunk_tiles(KS, KCAP);
    const int qpb = queries_per_block(QT);
    const int nqblk = (int)((n + qpb - 1) / qpb);
    const int64_t nq_pad = (int64_t)nqblk * qpb;
    const int64_t nchunk = (n + CT * 16 - 1) / (CT * 16);
    const int64_t nrow_pad = nchunk * CT * 16;
    std::vector<double> h((size_t)n * D);
    std::mt19937_64 g(1); std::normal_distribution<double> nd;
    for (auto& v : h) v = nd(g);
    double *X, *Yf, *pd; int* pi;
    CK(hipMalloc(&X, sizeof(double) * n * D));
    CK(hipMalloc(&Yf, sizeof(double) * nrow_pad * 4 * KS));
    const size_t nl = (size_t)rsplit * KCAP * nq_pad;
    CK(hipMalloc(&pd, sizeof(double) * nl));
    CK(hipMalloc(&pi, sizeof(int) * nl));
    CK(hipMemcpy(X, h.data(), sizeof(double) * n * D, hipMemcpyHostToDevice));
    double* center; CK(hipMalloc(&center, 64 * 8)); CK(hipMemset(center, 0, 64 * 8));
    pack_refs_kernel<<<(unsigned)((nrow_pad + 255) / 256), 256>>>(X, n, D, KS, nrow_pad, center, Yf);
    constexpr size_t LDS = lds_bytes(KS, KCAP);
    auto kern = knn_mfma_kernel<KS, KCAP>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        kern<<<nqblk * rsplit, kThreads, LDS>>>(Yf, nchunk, rsplit, X, center, n, D, nq_pad, nqblk, 1, 0, KSEL, pd, pi);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("KS=%d KCAP=%d QT=%d CT=%d ablate=%d lds=%zu n=%lld rsplit=%d grid=%d: %.2f ms  %.3f Mq/s  %.2f TFLOP/s\n", KS, KCAP, QT, CT, MCE_ABLATE, LDS,
               (long long)n, rsplit, nqblk * rsplit, ms, n / ms / 1e3, (double)n * n * 8.0 * KS / ms / 1e9);
    }
    return 0;
}
