// long_prep.hpp -- preparation kernels of the long-row fp64 sweep (knn_long.hpp: 128 <= d <= 1024): column means for any D, the
// queries in MFMA B-fragment order, their squared norms.  (The references are packed by pack_refs_kernel, pack_refs.hpp.)
// Reference: the "fit" step of MCEvidence.py:1093-1101.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mce {

constexpr int kLongMeanBlocks = 256;

// ---- column means for any D (the other kernels' statistics stop at 128 columns): block b sums its row range, thread t the columns
// t, t + 256, ...; partial[b][D]; fixed order: deterministic
__global__ __launch_bounds__(256) void long_col_mean_partial_kernel(const double* __restrict__ Y, int64_t nr, int D, int nblocks, double* __restrict__ partial)
{
    const int64_t per = (nr + nblocks - 1) / nblocks;
    const int64_t r0 = (int64_t)blockIdx.x * per;
    const int64_t r1 = (r0 + per < nr) ? r0 + per : nr;
    for (int c = threadIdx.x; c < D; c += 256) {
        double s0 = 0.0, s1 = 0.0;
        int64_t r = r0;
        for (; r + 1 < r1; r += 2) { s0 += Y[r * (int64_t)D + c]; s1 += Y[(r + 1) * (int64_t)D + c]; }
        if (r < r1) s0 += Y[r * (int64_t)D + c];
        partial[(int64_t)blockIdx.x * D + c] = s0 + s1;
    }
}
__global__ __launch_bounds__(256) void long_col_mean_final_kernel(const double* __restrict__ partial, int64_t nr, int D, int nblocks, double* __restrict__ center)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= D) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[(int64_t)b * D + c];
    center[c] = s / (double)nr;
}

// ---- queries into B-fragment order: fragment (tile, ks) = 64 doubles, lane l <-> query (l & 15), dimension 4 ks + (l >> 4);
// x' = [x - centre, 1, 0..]; rows >= nq: zeros.  One thread per (query, k-step): the 16 queries of a tile write 128 contiguous bytes.
__global__ __launch_bounds__(256) void long_pack_queries_kernel(const double* __restrict__ X, int64_t nq, int D, int KSP, int64_t nq_pad,
                                                               const double* __restrict__ center, double* __restrict__ Xf)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = nq_pad * (int64_t)KSP;
    if (e >= total) return;
    // e -> (tile, ks, c): consecutive threads = consecutive queries of a tile for one k-step
    const int c = (int)(e & 15);
    const int64_t rest = e >> 4;
    const int ks = (int)(rest % KSP);
    const int64_t tile = rest / KSP;
    const int64_t q = tile * 16 + c;
    const bool live = q < nq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int dim = 4 * ks + r;
        double v = 0.0;
        if (live && dim < D) v = X[q * (int64_t)D + dim] - center[dim];
        else if (live && dim == D) v = 1.0;
        Xf[(tile * KSP + ks) * 64 + r * 16 + c] = v;
    }
}
// |x - centre|^2 per query, the terms added in ascending dimension order through one fma chain
__global__ __launch_bounds__(256) void long_query_norms_kernel(const double* __restrict__ X, int64_t nq, int D, int64_t nq_pad,
                                                              const double* __restrict__ center, double* __restrict__ xn)
{
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq_pad) return;
    double s = 0.0;
    if (q < nq)
        for (int i = 0; i < D; ++i) { const double t = X[q * (int64_t)D + i] - center[i]; s = fma(t, t, s); }
    xn[q] = s;
}

}  // namespace mce
