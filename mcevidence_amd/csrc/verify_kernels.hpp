// verify_kernels.hpp -- an independent re-check of a finished k-NN search on a sample of its query rows (round 5).
//
// The default search is an fp16-MFMA FILTER with a bound that is rigorous on paper and measured with a 5x margin on the
// hardware; what comes out of it is claimed to be the exact fp64 result of the reference's
// `NearestNeighbors(...).kneighbors()` (MCEvidence.py:1093-1104).  This is the run-time certificate of that claim: for
// `nsample` query rows spread over the set, every reference row's squared distance is recomputed by plain fp64 differences
// -- no matrix cores, no packed operands, no lists, no bounds: nothing of the search's machinery -- and COUNTED against the
// distances the search reported.  With r_1 <= .. <= r_K the reported distances of a row, the row passes iff for every k
//     #{j : d2(q, j) <  r_k^2 (1 - tol)} <= k - 1      (nobody outside the list is closer than its k-th entry)
//     #{j : d2(q, j) <= r_k^2 (1 + tol)} >= k          (the k-th entry is not closer than the data allow)
// (the own row skipped under MCE_SELF_EXCLUDE; tol = 1e-9 relative, the parity tolerance of ln E).  A neighbour the filter
// dropped shows up in the first count.  Cost: nsample x nr x d fused multiply-adds at the fp64 vector rate + the LDS
// broadcasts of the query rows: ~2 ms for 1024 rows of C3 (1 M x 27), 0.5 ms for 256.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mce {

constexpr int kVerifyThreads = 256;
constexpr int kVerifyQB = 16;          // sampled queries per workgroup
constexpr int kVerifyMaxDim = 128;
constexpr int kVerifyMaxK = 32;
constexpr double kVerifyTol = 1e-9;

// sample s of n: rows spread evenly over [0, nq), the whole pattern shifted by `seed`
__host__ __device__ inline int64_t verify_row(int64_t s, int64_t n, int64_t nq, unsigned long long seed)
{
    const int64_t stride = nq / n > 0 ? nq / n : 1;
    return (int64_t)((seed % (unsigned long long)nq + (unsigned long long)(s * stride)) % (unsigned long long)nq);
}

__host__ __device__ constexpr size_t verify_lds_bytes(int D, int K) { return (size_t)kVerifyQB * D * 8 + (size_t)kVerifyQB * K * (8 + 8 + 4 + 4) + kVerifyQB * 8; }

// cnt [nsample][2][K]: (rows strictly inside the lower threshold of column k | rows inside the upper one)
__global__ __launch_bounds__(kVerifyThreads) void verify_scan_kernel(const double* __restrict__ X, int64_t nq, const double* __restrict__ Y, int64_t nr,
                                                                     int D, int K, int self_exclude, int64_t self_offset,
                                                                     const double* __restrict__ dist, int ld, int nsample, unsigned long long seed,
                                                                     int rsplit, int* __restrict__ cnt)
{
    extern __shared__ __attribute__((aligned(16))) char vr_raw[];
    double* const xq = reinterpret_cast<double*>(vr_raw);                       // [QB][D]
    double* const lo2 = xq + kVerifyQB * D;                                      // [QB][K]
    double* const hi2 = lo2 + kVerifyQB * K;
    int* const cl = reinterpret_cast<int*>(hi2 + kVerifyQB * K);               // [QB][K]
    int* const cu = cl + kVerifyQB * K;
    long long* const qrow = reinterpret_cast<long long*>(cu + kVerifyQB * K);  // [QB] the query's row (-1: none)
    const int s0 = (int)(blockIdx.x / rsplit) * kVerifyQB;
    const int split = (int)(blockIdx.x % rsplit);
    for (int e = threadIdx.x; e < kVerifyQB * K; e += kVerifyThreads) {
        const int s = s0 + e / K, k = e % K;
        double r = __builtin_huge_val();
        if (s < nsample) r = dist[verify_row(s, nsample, nq, seed) * (int64_t)ld + k];
        const double r2 = r * r;
        lo2[e] = r2 * (1.0 - kVerifyTol);
        hi2[e] = r2 * (1.0 + kVerifyTol);
        cl[e] = 0;
        cu[e] = 0;
    }
    for (int e = threadIdx.x; e < kVerifyQB * D; e += kVerifyThreads) {
        const int s = s0 + e / D;
        xq[e] = s < nsample ? X[verify_row(s, nsample, nq, seed) * (int64_t)D + e % D] : 0.0;
    }
    if (threadIdx.x < kVerifyQB) qrow[threadIdx.x] = s0 + (int)threadIdx.x < nsample ? (long long)verify_row(s0 + threadIdx.x, nsample, nq, seed) : -1;
    __syncthreads();
    const int64_t j_lo = nr * split / rsplit, j_hi = nr * (int64_t)(split + 1) / rsplit;
    for (int64_t j = j_lo + threadIdx.x; j < j_hi; j += kVerifyThreads) {
        const double* const y = Y + j * (int64_t)D;
        double acc[kVerifyQB];
#pragma unroll
        for (int q = 0; q < kVerifyQB; ++q) acc[q] = 0.0;
        for (int i0 = 0; i0 < D; i0 += 8) {
            double yv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) yv[u] = y[i0 + u < D ? i0 + u : D - 1];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + u < D) {
#pragma unroll
                    for (int q = 0; q < kVerifyQB; ++q) {
                        const double t = xq[q * D + i0 + u] - yv[u];
                        acc[q] = fma(t, t, acc[q]);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < kVerifyQB; ++q) {
            const double d2 = acc[q];
            if (!(d2 <= hi2[q * K + K - 1])) continue;                       // (NaN rows never count)
            if (qrow[q] < 0 || (self_exclude && j == self_offset + qrow[q])) continue;
            for (int k = 0; k < K; ++k) {
                if (d2 < lo2[q * K + k]) atomicAdd(&cl[q * K + k], 1);
                if (d2 <= hi2[q * K + k]) atomicAdd(&cu[q * K + k], 1);
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < kVerifyQB * K; e += kVerifyThreads) {
        const int s = s0 + e / K, k = e % K;
        if (s < nsample) {
            if (cl[e]) atomicAdd(&cnt[((int64_t)s * 2 + 0) * K + k], cl[e]);
            if (cu[e]) atomicAdd(&cnt[((int64_t)s * 2 + 1) * K + k], cu[e]);
        }
    }
}

// result[0] = rows checked, result[1] = rows that fail either count for some column (columns whose reported distance is
// infinite -- fewer than k usable reference rows -- are not judged)
__global__ __launch_bounds__(kVerifyThreads) void verify_check_kernel(const double* __restrict__ dist, int ld, int64_t nq, int K, int nsample, unsigned long long seed,
                                                                      const int* __restrict__ cnt, int* __restrict__ result)
{
    const int s = blockIdx.x * kVerifyThreads + threadIdx.x;
    if (s >= nsample) return;
    const int64_t row = verify_row(s, nsample, nq, seed);
    bool bad = false;
    for (int k = 0; k < K; ++k) {
        const double r = dist[row * (int64_t)ld + k];
        if (!(r < __builtin_huge_val())) { bad |= (r != r); continue; }
        bad |= cnt[((int64_t)s * 2 + 0) * K + k] > k;
        bad |= cnt[((int64_t)s * 2 + 1) * K + k] < k + 1;
    }
    atomicAdd(&result[0], 1);
    if (bad) atomicAdd(&result[1], 1);
}

}  // namespace mce
