// f16_prep.hpp -- one-pass preparation kernels of the fp16-filter search (knn_f16.hpp):
// radius -> power-of-two scale, fp16 packing of references (MFMA A-fragment order) and of
// queries, with the per-point rounding errors the rigorous filter bound needs.
// (The "fit" step, reference MCEvidence.py:1093-1101.)  Included by capi.hip only.
#pragma once
#include "knn_f16.hpp"
#include "pack_refs.hpp"

namespace mce {

__device__ __forceinline__ void atomic_max_pos(double* p, double v)   // v >= 0
{
    atomicMax(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v));
}

// ---------------------------------------------------------------------------
// power-of-two scale from a radius BOUND: the corner of the centred bounding box of the
// reference set (box_y, from col_stats) and, for a separate query set, of the queries (box_x,
// statistics taken about the same centre).  fp16 is floating point, so a loose bound costs no
// precision; it only has to keep |.| <= 200 (no overflow of -2y^ and of |y^|^2).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void f16_scale_kernel(const double* __restrict__ box_y, const double* __restrict__ box_x,
                                                       double* __restrict__ params)
{
    double b = box_y[threadIdx.x];
    if (box_x) b = fmax(b, box_x[threadIdx.x]);
    double r2 = b * b;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) r2 += __shfl_xor(r2, o, 64);
    if (threadIdx.x == 0) {
        const double r = sqrt(r2);
        double s = 1.0;
        if (r > 0.0 && r < __builtin_huge_val()) s = exp2(floor(log2(kHTargetRadius / r)));
        params[HP_RMAX] = r;
        params[HP_SCALE] = s;
    }
}

// box of a second point set about an EXISTING centre (cross evidence: queries != references)
__global__ __launch_bounds__(64) void f16_box_about_kernel(const double* __restrict__ partial, int D,
                                                           const double* __restrict__ center, double* __restrict__ box)
{
    const int col = threadIdx.x;
    double lo = __builtin_huge_val(), hi = -__builtin_huge_val();
    for (int b = 0; b < kMeanBlocks; ++b) {
        const double* o = partial + (int64_t)b * kStatStride;
        lo = fmin(lo, o[kMaxDimPad + col]);
        hi = fmax(hi, o[2 * kMaxDimPad + col]);
    }
    box[col] = (col < D && hi >= lo) ? fmax(fabs(hi - center[col]), fabs(lo - center[col])) : 0.0;
}

// ---------------------------------------------------------------------------
// references -> fp16 A fragments.  Packed layout (halfs):
//   Yh[((tile*KST + ks)*64 + lane)*8 + e],  lane = (row&31) + 32*h,  k = 16*ks + 8*h + e
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void f16_pack_refs_kernel(const double* __restrict__ Y, int64_t nr, int D, int KST,
                                                            int64_t nrow_pad, const double* __restrict__ center,
                                                            double* __restrict__ params, _Float16* __restrict__ Yh)
{
    // 256 rows per workgroup: the rows are read as ONE contiguous run (coalesced) into LDS, centred and
    // scaled on the way; then thread t works on row t from LDS (row stride D+1 doubles: conflict-free).
    extern __shared__ double rows[];                     // 256 * (D+1)
    const int64_t row0 = (int64_t)blockIdx.x * 256;
    const double s = params[HP_SCALE];
    const int ld = D | 1;                                // odd stride
    {
        const int64_t e0 = row0 * D;
        const int64_t e1 = ((row0 + 256 < nr) ? row0 + 256 : nr) * (int64_t)D;
        for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
            const int r = (int)((e - e0) / D), c = (int)((e - e0) - (int64_t)r * D);
            rows[r * ld + c] = (Y[e] - center[c]) * s;
        }
    }
    __syncthreads();
    const int64_t row = row0 + threadIdx.x;
    double ey = 0.0, yn = 0.0, rho = 0.0;
    if (row < nrow_pad) {
        const bool live = row < nr;
        const double* y = rows + threadIdx.x * ld;
        double err2 = 0.0, n2 = 0.0;
        if (live)
            for (int i = 0; i < D; ++i) {
                const double t = y[i];
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
                        if (k < D) x = (_Float16)(-2.0 * (double)(_Float16)y[k]);
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
