// knn_generic.hpp -- plain exact brute-force k-nearest-neighbour kernel for shapes outside the
// MFMA kernels' register budgets (d > 127 or K > 32; 64 <= d <= 127: knn_mfma.hpp): one thread per query, direct fp64
// differences, reference rows staged through LDS, sorted top-K list per query in global
// memory.  Same contract and output format as the MFMA kernels (reference
// MCEvidence.py:1093-1104).  These shapes are rare for MCMC chains; the kernel is there so that
// no input the reference accepts is refused, and (round 5) register-blocked so that stepping
// over the MFMA kernels' limit costs a factor of tens, not thousands.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mce {

constexpr int kGenThreads = 128;      // queries per workgroup
constexpr int kGenTileRows = 32;      // reference rows per tile: one accumulator per row and thread
constexpr int kGenChunk = 32;         // dimensions per pass: the thread's slice of its query row lives in registers
constexpr int kGenMaxDim = 1024;
constexpr int kGenMaxK = 1024;
__host__ __device__ constexpr size_t generic_lds_bytes() { return (size_t)kGenTileRows * kGenChunk * sizeof(double); }

// part_d/part_i: [1][K][nq_pad] (same layout as the MFMA kernels with rsplit = 1, KCAP = K)
//
// Round 5: register-blocked.  The first version re-read x[i] from memory for every (reference row, dimension) -- 128 threads,
// 128 different cache lines per load -- and ran at 0.5 TFLOP/s: 100 k x 100 k x 64 took 2.6 s next to 2 ms at d = 63 (fp16
// filter), a thousandfold cliff at the MFMA kernels' limit.  Now a thread keeps 32 dimensions of its query row in registers
// and 32 running sums, one per reference row of the tile; the tile's rows come through LDS 32 dimensions at a time
// (broadcast reads).  Every sum still adds its terms in ascending dimension order through one fma chain: the distances are
// those of the first version, bit for bit.  137.6 ms at 100 k x 100 k x 64 (was 2633).
__global__ __launch_bounds__(kGenThreads) void knn_generic_kernel(
    const double* __restrict__ X, int64_t nq, const double* __restrict__ Y, int64_t nr, int D, int K,
    int64_t nq_pad, int self_exclude, int64_t self_offset, double* __restrict__ part_d, int* __restrict__ part_i)
{
    __shared__ __attribute__((aligned(16))) double ytile[kGenTileRows * kGenChunk];
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
        double s[kGenTileRows];
#pragma unroll
        for (int r = 0; r < kGenTileRows; ++r) s[r] = 0.0;
        for (int c0 = 0; c0 < D; c0 += kGenChunk) {
            const int dims = D - c0 < kGenChunk ? D - c0 : kGenChunk;
            __syncthreads();
            // the tile's rows, dimensions [c0, c0 + 32): zero-padded (a zero difference adds nothing: fma(0, 0, s) = s)
            for (int e = threadIdx.x; e < kGenTileRows * kGenChunk; e += kGenThreads) {
                const int r = e / kGenChunk, i = e % kGenChunk;
                ytile[e] = (r < rows && i < dims) ? Y[(j0 + r) * (int64_t)D + c0 + i] : 0.0;
            }
            double xr[kGenChunk];
#pragma unroll
            for (int i = 0; i < kGenChunk; ++i) xr[i] = (i < dims) ? x[c0 + i] : 0.0;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < kGenTileRows; ++r) {
                double acc = s[r];
#pragma unroll
                for (int i = 0; i < kGenChunk; ++i) {
                    const double t = xr[i] - ytile[r * kGenChunk + i];
                    acc = fma(t, t, acc);
                }
                s[r] = acc;
            }
        }
        if (!live) continue;
#pragma unroll
        for (int r = 0; r < kGenTileRows; ++r) {
            const int64_t j = j0 + r;
            if (r >= rows || j == selfj) continue;
            const double sv = s[r];
            if (!(sv < thr)) continue;                      // ties keep the earlier (smaller) row
            int p = K - 1;                                 // sorted insertion, list in global memory
            while (p > 0) {
                const double dp = part_d[(int64_t)(p - 1) * nq_pad + q];
                if (!(dp > sv)) break;
                part_d[(int64_t)p * nq_pad + q] = dp;
                part_i[(int64_t)p * nq_pad + q] = part_i[(int64_t)(p - 1) * nq_pad + q];
                --p;
            }
            part_d[(int64_t)p * nq_pad + q] = sv;
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
