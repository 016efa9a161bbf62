// pack_refs.hpp -- one-time repack of the reference set into MFMA A-fragment order
// (the "fit" step of NearestNeighbors, reference MCEvidence.py:1093-1101: there it
// builds a KD-tree or is a no-op for brute force; here it is a 1-pass layout change).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mce {

// ---------------------------------------------------------------------------
// pack_refs: Y[nr, D] row-major -> Yf[tile][ks][lane], lane l <-> (row tile*16+(l&15), dim 4ks+(l>>4))
//   dims 0..D-1 : -2*y ; dim D : |y|^2 ; beyond : 0 ; rows >= nr : |y|^2 = +inf (never selected)
// ---------------------------------------------------------------------------
__global__ void pack_refs_kernel(const double* __restrict__ Y, int64_t nr, int D, int KS,
                                 int64_t nrow_pad, double* __restrict__ Yf)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrow_pad) return;
    const int64_t tile = row >> 4;
    const int c = (int)(row & 15);
    const bool live = row < nr;
    const double* y = Y + row * (int64_t)D;
    double nrm = 0.0;
    if (live)
        for (int i = 0; i < D; ++i) { const double t = y[i]; nrm = fma(t, t, nrm); }
    const int DP = 4 * KS;
    for (int dim = 0; dim < DP; ++dim) {
        double v = 0.0;
        if (dim < D) v = live ? -2.0 * y[dim] : 0.0;
        else if (dim == D) v = live ? nrm : __builtin_huge_val();
        Yf[(tile * KS + (dim >> 2)) * 64 + (dim & 3) * 16 + c] = v;
    }
}

}  // namespace mce
