// knn_generic.hpp -- plain exact brute-force k-nearest-neighbour kernel for shapes outside the
// MFMA kernels' register budgets (d > 63 or K > 32): one thread per query, direct fp64
// differences, reference rows staged through LDS, sorted top-K list per query in global
// memory.  Same contract and output format as the MFMA kernels (reference
// MCEvidence.py:1093-1104); throughput is not a goal here -- these shapes are rare for MCMC
// chains -- only that no input the reference accepts is refused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mce {

constexpr int kGenThreads = 128;      // queries per workgroup
constexpr int kGenTileRows = 16;      // reference rows per LDS tile
constexpr int kGenMaxDim = 1024;
constexpr int kGenMaxK = 1024;

// part_d/part_i: [1][K][nq_pad] (same layout as the MFMA kernels with rsplit = 1, KCAP = K)
__global__ __launch_bounds__(kGenThreads) void knn_generic_kernel(
    const double* __restrict__ X, int64_t nq, const double* __restrict__ Y, int64_t nr, int D, int K,
    int64_t nq_pad, int self_exclude, int64_t self_offset, double* __restrict__ part_d, int* __restrict__ part_i)
{
    extern __shared__ double ytile[];                      // kGenTileRows * D
    const int64_t q = (int64_t)blockIdx.x * kGenThreads + threadIdx.x;
    const bool live = q < nq;
    const double INF = __builtin_huge_val();
    const double* x = X + (live ? q : 0) * (int64_t)D;
    const int64_t selfj = (self_exclude && live) ? self_offset + q : -1;
    if (q < nq_pad)
        for (int k = 0; k < K; ++k) { part_d[(int64_t)k * nq_pad + q] = INF; part_i[(int64_t)k * nq_pad + q] = -1; }
    double thr = INF;
    for (int64_t j0 = 0; j0 < nr; j0 += kGenTileRows) {
        const int rows = (int)((nr - j0 < kGenTileRows) ? nr - j0 : kGenTileRows);
        __syncthreads();
        for (int e = threadIdx.x; e < rows * D; e += kGenThreads) ytile[e] = Y[j0 * D + e];
        __syncthreads();
        if (!live) continue;
        for (int r = 0; r < rows; ++r) {
            const int64_t j = j0 + r;
            if (j == selfj) continue;
            const double* y = ytile + r * D;
            double s = 0.0;
            for (int i = 0; i < D; ++i) { const double t = x[i] - y[i]; s = fma(t, t, s); }
            if (!(s < thr)) continue;                      // ties keep the earlier (smaller) row
            int p = K - 1;                                 // sorted insertion, list in global memory
            while (p > 0) {
                const double dp = part_d[(int64_t)(p - 1) * nq_pad + q];
                if (!(dp > s)) break;
                part_d[(int64_t)p * nq_pad + q] = dp;
                part_i[(int64_t)p * nq_pad + q] = part_i[(int64_t)(p - 1) * nq_pad + q];
                --p;
            }
            part_d[(int64_t)p * nq_pad + q] = s;
            part_i[(int64_t)p * nq_pad + q] = (int)j;
            thr = part_d[(int64_t)(K - 1) * nq_pad + q];
        }
    }
}

// lists [1][K][nq_pad] (already sorted, exact keys) -> dist[nq,K] (+ idx); SELF_INCLUDE moves the
// query's own row to column 0 with distance exactly 0.
__global__ __launch_bounds__(256) void generic_finalize_kernel(const double* __restrict__ part_d, const int* __restrict__ part_i,
                                                              int64_t nq, int64_t nq_pad, int K, int self_mode, int64_t self_offset,
                                                              double* __restrict__ dist, int64_t* __restrict__ idx)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const int selfj = (self_mode == 1) ? (int)(self_offset + q) : -1;
    int pself = -1;
    if (selfj >= 0)
        for (int k = 0; k < K; ++k)
            if (part_i[(int64_t)k * nq_pad + q] == selfj) { pself = k; break; }
    int o = 0;
    if (pself >= 0) {
        dist[q * (int64_t)K] = 0.0;
        if (idx) idx[q * (int64_t)K] = selfj;
        o = 1;
    }
    for (int k = 0; k < K && o < K; ++k) {
        if (k == pself) continue;
        const int i = part_i[(int64_t)k * nq_pad + q];
        dist[q * (int64_t)K + o] = (i >= 0) ? sqrt(part_d[(int64_t)k * nq_pad + q]) : __builtin_huge_val();
        if (idx) idx[q * (int64_t)K + o] = i;
        ++o;
    }
}

}  // namespace mce
