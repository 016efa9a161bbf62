// feeders.hpp -- device versions of the hot path's feeders:
//   get_covariance   (reference MCEvidence.py:851-882: UNWEIGHTED np.cov of the first ndim columns)
//   diagonalise_chain (:842-849: s @ eVec, column i divided by sqrt(eVal[i]); the mean is NOT removed)
// so that one upload of the chain feeds covariance -> (host: d x d eigen-system) -> whitening ->
// kNN -> reduction without the whitened matrix ever visiting the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pack_refs.hpp"

namespace mce {

constexpr int kCovBlocks = 256;
constexpr int kCovThreads = 256;
constexpr int kCovTileRows = 32;

// partial[b][p] = sum over the block's rows of (s_i - m_i)(s_j - m_j), p = packed index of (i <= j).
// Rows are staged through LDS (centred on the way); thread t owns pairs t, t+256, ...
// A launch covers the pairs [p0, p0 + kCovPairsPerLaunch): d <= 63 has at most 2016 of them, one launch; the wide feeders
// (64 <= d <= 127, round 5) take up to four.
constexpr int kCovPairsPerLaunch = 8 * kCovThreads;
__global__ __launch_bounds__(kCovThreads) void cov_partial_kernel(const double* __restrict__ S, int64_t n, int d,
                                                                  const double* __restrict__ mean,
                                                                  double* __restrict__ partial /*[kCovBlocks][npair]*/, int p0)
{
    extern __shared__ double tile[];                 // kCovTileRows * d
    const int npair_all = d * (d + 1) / 2;
    const int npair = p0 + kCovPairsPerLaunch < npair_all ? p0 + kCovPairsPerLaunch : npair_all;      // end of this launch's pairs
    const int64_t per = (n + kCovBlocks - 1) / kCovBlocks;
    const int64_t r0 = (int64_t)blockIdx.x * per;
    const int64_t r1 = (r0 + per < n) ? r0 + per : n;
    // this thread's pairs (at most 8)
    int pi_[8], pj_[8];
    double acc[8];
    int np = 0;
    for (int p = p0 + threadIdx.x; p < npair && np < 8; p += kCovThreads) {
        int i = 0, base = 0;                         // packed upper triangle, row-major: (i, j>=i)
        while (base + (d - i) <= p) { base += d - i; ++i; }
        pi_[np] = i;
        pj_[np] = i + (p - base);
        acc[np] = 0.0;
        ++np;
    }
    for (int64_t t0 = r0; t0 < r1; t0 += kCovTileRows) {
        const int rows = (int)((r1 - t0 < kCovTileRows) ? r1 - t0 : kCovTileRows);
        __syncthreads();
        for (int e = threadIdx.x; e < rows * d; e += kCovThreads) {
            const int c = e % d;
            tile[e] = S[t0 * (int64_t)d + e] - mean[c];
        }
        __syncthreads();
        for (int r = 0; r < rows; ++r) {
            const double* s = tile + r * d;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (u < np) acc[u] = fma(s[pi_[u]], s[pj_[u]], acc[u]);
        }
    }
    int u = 0;
    for (int p = p0 + threadIdx.x; p < npair && u < 8; p += kCovThreads, ++u) partial[(int64_t)blockIdx.x * npair_all + p] = acc[u];
}

// cov[i][j] = cov[j][i] = sum_b partial[b][p] / (n - 1)   (fixed order: deterministic)
__global__ __launch_bounds__(kCovThreads) void cov_final_kernel(const double* __restrict__ partial, int64_t n, int d,
                                                                double* __restrict__ cov /*[d*d]*/)
{
    const int npair = d * (d + 1) / 2;
    for (int p = threadIdx.x; p < npair; p += kCovThreads) {
        double s = 0.0;
        for (int b = 0; b < kCovBlocks; ++b) s += partial[(int64_t)b * npair + p];
        int i = 0, base = 0;
        while (base + (d - i) <= p) { base += d - i; ++i; }
        const int j = i + (p - base);
        const double v = s / (double)(n - 1);
        cov[i * d + j] = v;
        cov[j * d + i] = v;
    }
}

// out[r][c] = (sum_k S[r][k] * evec[k][c]) * scale[c],  scale = 1/sqrt(eval).  64 rows per workgroup,
// staged through LDS both ways so global reads and writes are contiguous runs.  `out` may alias `S`.
constexpr int kWhitenRows = 64;
__host__ __device__ constexpr size_t whiten_lds_bytes(int d) { return ((size_t)d * d + (size_t)2 * kWhitenRows * (d | 1)) * sizeof(double); }

__global__ __launch_bounds__(kWhitenRows) void whiten_kernel(const double* S, int64_t n, int d,
                                                             const double* __restrict__ evec /*[d][d] row-major*/,
                                                             const double* __restrict__ scale /*[d]*/, double* out)
{
    extern __shared__ double sh[];                   // evec d*d | rows in 64*(d|1) | rows out 64*(d|1)
    const int ld = d | 1;
    double* U = sh;
    double* rin = sh + d * d;
    double* rout = rin + kWhitenRows * ld;
    for (int e = threadIdx.x; e < d * d; e += kWhitenRows) U[e] = evec[e];
    const int64_t row0 = (int64_t)blockIdx.x * kWhitenRows;
    const int64_t e0 = row0 * d, e1 = ((row0 + kWhitenRows < n) ? row0 + kWhitenRows : n) * (int64_t)d;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += kWhitenRows) {
        const int r = (int)((e - e0) / d), c = (int)((e - e0) - (int64_t)r * d);
        rin[r * ld + c] = S[e];
    }
    __syncthreads();
    if (row0 + threadIdx.x < n) {
        const double* s = rin + threadIdx.x * ld;
        double* o = rout + threadIdx.x * ld;
        for (int c = 0; c < d; ++c) {
            double a = 0.0;
            for (int k = 0; k < d; ++k) a = fma(s[k], U[k * d + c], a);
            o[c] = a * scale[c];
        }
    }
    __syncthreads();
    for (int64_t e = e0 + threadIdx.x; e < e1; e += kWhitenRows) {
        const int r = (int)((e - e0) / d), c = (int)((e - e0) - (int64_t)r * d);
        out[e] = rout[r * ld + c];
    }
}


// the same for 64 <= d <= 127 (round 5): the eigenvectors (up to 126 KB) stay in global memory -- every thread of the workgroup
// reads the same element, a uniform load -- and the rows go through LDS as above (2 x 64 x 127 doubles = 127 KB).  The sums add
// their terms in the same order: the results are what the narrow kernel's would be.
__host__ __device__ constexpr size_t whiten_wide_lds_bytes(int d) { return ((size_t)2 * kWhitenRows * (d | 1)) * sizeof(double); }
__global__ __launch_bounds__(kWhitenRows) void whiten_wide_kernel(const double* S, int64_t n, int d, const double* __restrict__ evec,
                                                                  const double* __restrict__ scale, double* out)
{
    extern __shared__ double sh[];                   // rows in 64*(d|1) | rows out 64*(d|1)
    const int ld = d | 1;
    double* rin = sh;
    double* rout = rin + kWhitenRows * ld;
    const int64_t row0 = (int64_t)blockIdx.x * kWhitenRows;
    const int64_t e0 = row0 * d, e1 = ((row0 + kWhitenRows < n) ? row0 + kWhitenRows : n) * (int64_t)d;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += kWhitenRows) {
        const int r = (int)((e - e0) / d), c = (int)((e - e0) - (int64_t)r * d);
        rin[r * ld + c] = S[e];
    }
    __syncthreads();
    {
        const double* s = rin + threadIdx.x * ld;      // (rows beyond n: whatever the buffer holds; never written out)
        double* o = rout + threadIdx.x * ld;
        for (int c = 0; c < d; ++c) {
            double a = 0.0;
            for (int k = 0; k < d; ++k) a = fma(s[k], evec[k * d + c], a);
            o[c] = a * scale[c];
        }
    }
    __syncthreads();
    for (int64_t e = e0 + threadIdx.x; e < e1; e += kWhitenRows) {
        const int r = (int)((e - e0) / d), c = (int)((e - e0) - (int64_t)r * d);
        out[e] = rout[r * ld + c];
    }
}

// 64-bit fingerprint of a device buffer of 8-byte words: sum over i of mix64(word_i + (salt + i) * golden) -- the sum of
// per-word hashes is order-independent (integer adds), position-sensitive (i enters the hash), and costs one pass at HBM
// speed.  Used by the multi-rank feed (mce_evidence_feed_part_f64) to check that every rank was handed the same samples,
// weights and likelihoods -- where the host route hashed all bytes with BLAKE2b (0.3 s per GB).  `out` accumulates:
// clear it, then launch once per buffer with different salts.
constexpr int kSumThreads = 256;
__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(kSumThreads) void checksum_kernel(const unsigned long long* __restrict__ words, int64_t n,
                                                               unsigned long long salt, unsigned long long* __restrict__ out)
{
    unsigned long long h = 0;
    for (int64_t i = (int64_t)blockIdx.x * kSumThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kSumThreads)
        h += mix64(words[i] + (salt + (unsigned long long)i) * 0x9E3779B97F4A7C15ull);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) h += __shfl_xor(h, o, 64);
    if ((threadIdx.x & 63) == 0 && h != 0) atomicAdd(out, h);
}

}  // namespace mce
