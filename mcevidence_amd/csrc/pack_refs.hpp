// pack_refs.hpp -- one-time repack of the reference set into MFMA A-fragment order
// (the "fit" step of NearestNeighbors, reference MCEvidence.py:1093-1101: there it
// builds a KD-tree or is a no-op for brute force; here it is a 1-pass layout change).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mce {

// ---------------------------------------------------------------------------
// Column means of the reference set.  Distances are translation invariant, and the
// GEMM form |x|^2 + |y|^2 - 2x.y loses absolute accuracy ~1e-16*|x|^2, so both sets are
// centred on the reference mean before packing.  This matters for real chains: the
// reference's whitening (MCEvidence.py:842-849) rotates and rescales but does NOT
// subtract the mean, and e.g. CosmoMC's theta sits ~2000 sigma from the origin.
// Deterministic: fixed row ranges per workgroup, fixed-order final pass.
// ---------------------------------------------------------------------------
constexpr int kMeanBlocks = 256;
constexpr int kMeanThreads = 256;
constexpr int kMaxDimPad = 64;
constexpr int kStatStride = 3 * kMaxDimPad;   // per block: sum[64] | min[64] | max[64]

// Column statistics (sum, min, max) of a row-major [n, D] matrix in ONE coalesced pass:
// a wave covers RPW = 64/D consecutive rows per trip (lane -> row l/D, column l%D: the lanes read
// one contiguous run of RPW*D doubles), every lane keeps its own column's partial.
__global__ __launch_bounds__(kMeanThreads) void col_stats_partial_kernel(const double* __restrict__ Y, int64_t nr, int D,
                                                                        double* __restrict__ partial /*[kMeanBlocks][3][64]*/)
{
    __shared__ double s_sum[kMeanThreads], s_min[kMeanThreads], s_max[kMeanThreads];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int RPW = 64 / D;                       // rows per wave trip (D <= 63 -> >= 1)
    const int lr = lane / D, col = lane - lr * D;
    const bool on = lr < RPW;
    const int64_t per = (nr + kMeanBlocks - 1) / kMeanBlocks;
    const int64_t r0 = (int64_t)blockIdx.x * per;
    const int64_t r1 = (r0 + per < nr) ? r0 + per : nr;
    double sum = 0.0, mn = __builtin_huge_val(), mx = -__builtin_huge_val();
    if (on) {
        // four rows per trip and lane, each with its own partial sum (four loads in flight: one block per CU is all the
        // parallelism there is -- 0.19 -> 0.07 ms at 1M x 27); combined in a fixed order: deterministic
        const int64_t step = (int64_t)(kMeanThreads / 64) * RPW;
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
        int64_t r = r0 + wave * RPW + lr;
        for (; r + 3 * step < r1; r += 4 * step) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = Y[(r + u * step) * (int64_t)D + col];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s4[u] += v[u];
                mn = fmin(mn, v[u]);
                mx = fmax(mx, v[u]);
            }
        }
        for (; r < r1; r += step) {
            const double v = Y[r * (int64_t)D + col];
            s4[0] += v;
            mn = fmin(mn, v);
            mx = fmax(mx, v);
        }
        sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    s_sum[threadIdx.x] = sum;
    s_min[threadIdx.x] = mn;
    s_max[threadIdx.x] = mx;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = threadIdx.x;
        double a = 0.0, lo = __builtin_huge_val(), hi = -__builtin_huge_val();
        if (c < D)
            for (int w = 0; w < kMeanThreads / 64; ++w)
                for (int k = 0; k < RPW; ++k) {            // fixed order: deterministic
                    const int t = w * 64 + k * D + c;
                    a += s_sum[t];
                    lo = fmin(lo, s_min[t]);
                    hi = fmax(hi, s_max[t]);
                }
        double* o = partial + (int64_t)blockIdx.x * kStatStride;
        o[c] = a;
        o[kMaxDimPad + c] = lo;
        o[2 * kMaxDimPad + c] = hi;
    }
}

// centre[c] = column mean;  box[c] = max(|max_c - centre|, |min_c - centre|)  (0 beyond D)
__global__ __launch_bounds__(64) void col_stats_final_kernel(const double* __restrict__ partial, int64_t nr, int D,
                                                             double* __restrict__ center /*[64]*/, double* __restrict__ box /*[64]*/)
{
    const int col = threadIdx.x;
    double s = 0.0, lo = __builtin_huge_val(), hi = -__builtin_huge_val();
    for (int b = 0; b < kMeanBlocks; ++b) {
        const double* o = partial + (int64_t)b * kStatStride;
        s += o[col];
        lo = fmin(lo, o[kMaxDimPad + col]);
        hi = fmax(hi, o[2 * kMaxDimPad + col]);
    }
    const double c = (col < D) ? s / (double)nr : 0.0;
    center[col] = c;
    if (box) box[col] = (col < D && hi >= lo) ? fmax(fabs(hi - c), fabs(lo - c)) : 0.0;
}

// 64 <= D <= 128 (the fp64 sweep's wide form, round 5): column means only -- thread t sums column t % 128 over every second row of
// its block's range; partial[b][c] in the same [kMeanBlocks][kStatStride] scratch (kStatStride = 192 >= 128).  Deterministic.
__global__ __launch_bounds__(256) void col_mean_wide_partial_kernel(const double* __restrict__ Y, int64_t nr, int D, double* __restrict__ partial)
{
    __shared__ double s_sum[256];
    const int c = threadIdx.x & 127, rl = threadIdx.x >> 7;
    const int64_t per = (nr + kMeanBlocks - 1) / kMeanBlocks;
    const int64_t r0 = (int64_t)blockIdx.x * per;
    const int64_t r1 = (r0 + per < nr) ? r0 + per : nr;
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    if (c < D) {
        int64_t r = r0 + rl;
        for (; r + 6 < r1; r += 8) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s4[u] += Y[(r + 2 * u) * (int64_t)D + c];
        }
        for (; r < r1; r += 2) s4[0] += Y[r * (int64_t)D + c];
    }
    s_sum[threadIdx.x] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    __syncthreads();
    if (threadIdx.x < 128) partial[(int64_t)blockIdx.x * kStatStride + c] = s_sum[c] + s_sum[128 + c];
}
__global__ __launch_bounds__(128) void col_mean_wide_final_kernel(const double* __restrict__ partial, int64_t nr, int D, double* __restrict__ center /*[128]*/)
{
    const int c = threadIdx.x;
    double s = 0.0;
    for (int b = 0; b < kMeanBlocks; ++b) s += partial[(int64_t)b * kStatStride + c];
    center[c] = (c < D) ? s / (double)nr : 0.0;
}

// ---------------------------------------------------------------------------
// pack_refs: Y[nr, D] row-major -> Yf[tile][ks][lane], lane l <-> (row tile*16+(l&15), dim 4ks+(l>>4))
//   with yc = y - center:  dims 0..D-1 : -2*yc ; dim D : |yc|^2 ; beyond : 0 ;
//   rows >= nr : |yc|^2 = +inf (never selected)
// ---------------------------------------------------------------------------
__global__ void pack_refs_kernel(const double* __restrict__ Y, int64_t nr, int D, int KS,
                                 int64_t nrow_pad, const double* __restrict__ center, double* __restrict__ Yf)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrow_pad) return;
    const int64_t tile = row >> 4;
    const int c = (int)(row & 15);
    const bool live = row < nr;
    const double* y = Y + row * (int64_t)D;
    double nrm = 0.0;
    if (live)
        for (int i = 0; i < D; ++i) { const double t = y[i] - center[i]; nrm = fma(t, t, nrm); }
    const int DP = 4 * KS;
    for (int dim = 0; dim < DP; ++dim) {
        double v = 0.0;
        if (dim < D) v = live ? -2.0 * (y[dim] - center[dim]) : 0.0;
        else if (dim == D) v = live ? nrm : __builtin_huge_val();
        Yf[(tile * KS + (dim >> 2)) * 64 + (dim & 3) * 16 + c] = v;
    }
}

}  // namespace mce
