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

__global__ __launch_bounds__(kMeanThreads) void col_sum_partial_kernel(const double* __restrict__ Y, int64_t nr, int D,
                                                                      double* __restrict__ partial /*[kMeanBlocks][64]*/)
{
    __shared__ double red[kMeanThreads];
    const int64_t per = (nr + kMeanBlocks - 1) / kMeanBlocks;
    const int64_t r0 = (int64_t)blockIdx.x * per;
    const int64_t r1 = (r0 + per < nr) ? r0 + per : nr;
    // thread t owns column t % 64 of rows r0 + t/64, +4, ...  (coalesced along a row)
    const int col = threadIdx.x & 63;
    const int sub = threadIdx.x >> 6;
    double acc = 0.0;
    if (col < D)
        for (int64_t r = r0 + sub; r < r1; r += kMeanThreads / 64) acc += Y[r * (int64_t)D + col];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < kMeanThreads / 64; ++i) s += red[i * 64 + threadIdx.x];
        partial[(int64_t)blockIdx.x * kMaxDimPad + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(64) void col_mean_final_kernel(const double* __restrict__ partial, int64_t nr, int D,
                                                            double* __restrict__ center /*[64]*/)
{
    const int col = threadIdx.x;
    double s = 0.0;
    for (int b = 0; b < kMeanBlocks; ++b) s += partial[(int64_t)b * kMaxDimPad + col];
    center[col] = (col < D) ? s / (double)nr : 0.0;
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
