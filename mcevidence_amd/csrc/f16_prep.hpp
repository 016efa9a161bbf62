// f16_prep.hpp -- one-pass preparation kernels of the fp16-filter search (knn_f16.hpp):
// radius -> power-of-two scale, fp16 packing of references (MFMA A-fragment order) and of
// queries, with the per-point rounding errors the rigorous filter bound needs.
// (The "fit" step, reference MCEvidence.py:1093-1101.)  Included by capi.hip only.
#pragma once
#include "knn_f16.hpp"

namespace mce {

__device__ __forceinline__ void atomic_max_pos(double* p, double v)   // v >= 0
{
    atomicMax(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v));
}

// ---------------------------------------------------------------------------
// radius of both point sets around the centre -> power-of-two scale
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void f16_radius_kernel(const double* __restrict__ X, int64_t nq,
                                                         const double* __restrict__ Y, int64_t nr, int D,
                                                         const double* __restrict__ center, double* __restrict__ params)
{
    double m = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nq + nr; r += stride) {
        const double* p = (r < nq) ? X + r * (int64_t)D : Y + (r - nq) * (int64_t)D;
        double s2 = 0.0;
        for (int i = 0; i < D; ++i) { const double t = p[i] - center[i]; s2 = fma(t, t, s2); }
        m = fmax(m, s2);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomic_max_pos(params + HP_RMAX, sqrt(m));
}

__global__ void f16_scale_kernel(double* __restrict__ params)
{
    const double r = params[HP_RMAX];
    double s = 1.0;
    if (r > 0.0 && r < __builtin_huge_val()) s = exp2(floor(log2(kHTargetRadius / r)));
    params[HP_SCALE] = s;
}

// ---------------------------------------------------------------------------
// references -> fp16 A fragments.  Packed layout (halfs):
//   Yh[((tile*KST + ks)*64 + lane)*8 + e],  lane = (row&31) + 32*h,  k = 16*ks + 8*h + e
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void f16_pack_refs_kernel(const double* __restrict__ Y, int64_t nr, int D, int KST,
                                                            int64_t nrow_pad, const double* __restrict__ center,
                                                            double* __restrict__ params, _Float16* __restrict__ Yh)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double ey = 0.0, yn = 0.0, rho = 0.0;
    if (row < nrow_pad) {
        const bool live = row < nr;
        const double s = params[HP_SCALE];
        const double* y = Y + row * (int64_t)D;
        double err2 = 0.0, n2 = 0.0;
        if (live)
            for (int i = 0; i < D; ++i) {
                const double t = (y[i] - center[i]) * s;
                const double th = (double)(_Float16)t;
                err2 = fma(t - th, t - th, err2);
                n2 = fma(th, th, n2);
            }
        const _Float16 n_hi = (_Float16)n2;
        const _Float16 n_mid = (_Float16)(n2 - (double)n_hi);
        const _Float16 n_lo = (_Float16)(n2 - (double)n_hi - (double)n_mid);
        ey = sqrt(err2);
        yn = sqrt(n2);
        rho = fabs(n2 - (double)n_hi - (double)n_mid - (double)n_lo);
        const int64_t tile = row >> 5;
        const int i32 = (int)(row & 31);
        for (int ks = 0; ks < KST; ++ks)
            for (int h = 0; h < 2; ++h) {
                v8h v;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = 16 * ks + 8 * h + e;
                    _Float16 x = (_Float16)0.0f;
                    if (live) {
                        if (k < D) x = (_Float16)(-2.0 * (double)(_Float16)((y[k] - center[k]) * s));
                        else if (k == D) x = n_hi;
                        else if (k == D + 1) x = n_mid;
                        else if (k == D + 2) x = n_lo;
                    } else if (k == D) {
                        x = (_Float16)__builtin_huge_valf();      // padding rows: A = +inf, never below a finite gate
                    }
                    v[e] = x;
                }
                *reinterpret_cast<v8h*>(Yh + (((tile * KST + ks) * 64 + i32 + 32 * h) * 8)) = v;
            }
    }
    // block maxima -> global maxima (max is order independent: deterministic)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        ey = fmax(ey, __shfl_xor(ey, o, 64));
        yn = fmax(yn, __shfl_xor(yn, o, 64));
        rho = fmax(rho, __shfl_xor(rho, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomic_max_pos(params + HP_EY, ey);
        atomic_max_pos(params + HP_YHATMAX, yn);
        atomic_max_pos(params + HP_RHO, rho);
    }
}

// ---------------------------------------------------------------------------
// queries -> fp16 rows Xh[nq_pad][16*KST] (x' = [x^, 1, 1, 1, 0..]) + qinfo[q] = {e_x, |x^|^2}
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void f16_pack_queries_kernel(const double* __restrict__ X, int64_t nq, int64_t nq_pad,
                                                               int D, int KST, const double* __restrict__ center,
                                                               const double* __restrict__ params,
                                                               _Float16* __restrict__ Xh, double* __restrict__ qinfo)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq_pad) return;
    const bool live = q < nq;
    const double s = params[HP_SCALE];
    const double* x = X + q * (int64_t)D;
    double err2 = 0.0, n2 = 0.0;
    const int KD = 16 * KST;
    for (int c0 = 0; c0 < KD; c0 += 8) {
        v8h v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = c0 + e;
            _Float16 xv = (_Float16)0.0f;
            if (live) {
                if (k < D) {
                    const double t = (x[k] - center[k]) * s;
                    xv = (_Float16)t;
                    err2 = fma(t - (double)xv, t - (double)xv, err2);
                    n2 = fma((double)xv, (double)xv, n2);
                } else if (k < D + 3) {
                    xv = (_Float16)1.0f;
                }
            }
            v[e] = xv;
        }
        *reinterpret_cast<v8h*>(Xh + q * (int64_t)KD + c0) = v;
    }
    qinfo[2 * q + 0] = sqrt(err2);
    qinfo[2 * q + 1] = n2;
}

}  // namespace mce
