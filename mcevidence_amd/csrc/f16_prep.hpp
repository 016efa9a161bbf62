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

// d > 63 (the deep filter, knn_deep.hpp): the column statistics above hold 64 columns; here the radius is taken from the rows
// themselves -- the largest |row - centre| of a set, one thread per row, one atomic per wave -- and the scale from it
// (launch f16_radius_rows_kernel once per set, then f16_scale_from_radius_kernel)
__global__ __launch_bounds__(256) void f16_radius_rows_kernel(const double* __restrict__ Y, int64_t n, int D, const double* __restrict__ center,
                                                              double* __restrict__ params)
{
    double r2 = 0.0;
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < n; row += (int64_t)gridDim.x * 256) {
        double s2 = 0.0;
        for (int k = 0; k < D; ++k) { const double t = Y[row * (int64_t)D + k] - center[k]; s2 = fma(t, t, s2); }
        r2 = fmax(r2, s2);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) r2 = fmax(r2, __shfl_xor(r2, o, 64));
    // (NaN / inf rows: the maximum is then not a number the scale can use -- f16_scale_from_radius_kernel keeps s = 1, as f16_scale_kernel does)
    if ((threadIdx.x & 63) == 0 && r2 > 0.0) atomic_max_pos(params + HP_RMAX, sqrt(r2) * (1.0 + 1e-12));
}
__global__ void f16_scale_from_radius_kernel(double* __restrict__ params)
{
    const double r = params[HP_RMAX];
    double s = 1.0;
    if (r > 0.0 && r < __builtin_huge_val()) s = exp2(floor(log2(kHTargetRadius / r)));
    params[HP_SCALE] = s;
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
// One lane per (row, fragment f = 2*ks + h): it converts the row's elements 8f..8f+7 (a 64-byte
// segment), the G = 2*KST lanes of a row combine their error / norm partial sums, and each lane
// writes its own 16-byte fragment piece.  Lane = f*R + r with R = 64/G rows per wave, so the
// stores of a wave are G contiguous runs (same fragment slot, consecutive rows).
// ---------------------------------------------------------------------------
// Xh_same / qinfo_same (round 6; null: not asked for): the queries ARE these rows (auto evidence, one buffer) -- the same pass
// also writes their fp16 query rows x' = [x^, 1, 1, 1, 0..] and qinfo = {e_x, |x^|^2} for the rows below nq_pad_same, bit for
// bit what f16_pack_queries_kernel would write (the same converted values, the same order of the partial sums), and that
// kernel's launch is saved (0.095 ms of a C3 step).
__global__ __launch_bounds__(256) void f16_pack_refs_kernel(const double* __restrict__ Y, int64_t nr, int D, int KST,
                                                            int64_t nrow_pad, const double* __restrict__ center,
                                                            double* __restrict__ params, _Float16* __restrict__ Yh,
                                                            _Float16* __restrict__ Xh_same = nullptr, double* __restrict__ qinfo_same = nullptr,
                                                            int64_t nq_pad_same = 0)
{
    const int G = 2 * KST, R = 64 / G;
    const int lane = threadIdx.x & 63;
    const int f = lane / R, r = lane - f * R;
    const bool on = f < G;
    const double s = params[HP_SCALE];
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t ngroups = (nrow_pad + R - 1) / R;
    double ey = 0.0, yn = 0.0, rho = 0.0;                     // running maxima of this lane
    // grid-stride over groups of R rows: the three global maxima cost one atomic per WAVE at the end
    for (int64_t g = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); g < ngroups; g += nwaves) {
        const int64_t row = g * R + r;
        const bool inrange = on && row < nrow_pad;
        const bool live = inrange && row < nr;
        double th[8];
        double err2 = 0.0, n2 = 0.0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * f + e;
            double t = 0.0;
            if (live && k < D) t = (Y[row * (int64_t)D + k] - center[k]) * s;
            const double h = (double)(_Float16)t;
            th[e] = h;
            err2 = fma(t - h, t - h, err2);
            n2 = fma(h, h, n2);
        }
        // combine over the row's G lanes (fixed order: bitwise reproducible)
        double e_tot = 0.0, n_tot = 0.0;
        for (int m = 0; m < G; ++m) {
            e_tot += __shfl(err2, m * R + r, 64);
            n_tot += __shfl(n2, m * R + r, 64);
        }
        const int npieces = f16_norm_pieces(D);
        const _Float16 n_hi = (_Float16)n_tot;
        const _Float16 n_mid = npieces > 1 ? (_Float16)(n_tot - (double)n_hi) : (_Float16)0.0f;
        const _Float16 n_lo = npieces > 2 ? (_Float16)(n_tot - (double)n_hi - (double)n_mid) : (_Float16)0.0f;
        if (inrange) {
            v8h v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 8 * f + e;
                _Float16 x = (_Float16)0.0f;
                if (live) {
                    if (k < D) x = (_Float16)(-2.0 * th[e]);
                    else if (k == D) x = n_hi;
                    else if (k == D + 1 && npieces > 1) x = n_mid;
                    else if (k == D + 2 && npieces > 2) x = n_lo;
                } else if (k == D) {
                    x = (_Float16)__builtin_huge_valf();      // padding rows: A = +inf, never below a finite gate
                }
                v[e] = x;
            }
            const int64_t tile = row >> 5;
            const int i32 = (int)(row & 31);
            const int ks = f >> 1, hh = f & 1;
            *reinterpret_cast<v8h*>(Yh + (((tile * KST + ks) * 64 + i32 + 32 * hh) * 8)) = v;
            if (Xh_same && row < nq_pad_same) {
                v8h xq;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = 8 * f + e;
                    _Float16 x = (_Float16)0.0f;
                    if (live) {
                        if (k < D) x = (_Float16)th[e];                       // (th holds the converted value: the conversion back is exact)
                        else if (k < D + npieces) x = (_Float16)1.0f;
                    }
                    xq[e] = x;
                }
                *reinterpret_cast<v8h*>(Xh_same + row * (int64_t)(16 * KST) + 8 * f) = xq;
                if (f == 0) {
                    qinfo_same[2 * row + 0] = sqrt(e_tot);
                    qinfo_same[2 * row + 1] = n_tot;
                }
            }
        }
        if (live) {
            ey = fmax(ey, sqrt(e_tot));
            yn = fmax(yn, sqrt(n_tot));
            rho = fmax(rho, fabs(n_tot - (double)n_hi - (double)n_mid - (double)n_lo));
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        ey = fmax(ey, __shfl_xor(ey, o, 64));
        yn = fmax(yn, __shfl_xor(yn, o, 64));
        rho = fmax(rho, __shfl_xor(rho, o, 64));
    }
    // one set of atomics per BLOCK (the waves' maxima meet in LDS first): the three addresses are shared by the whole launch
    __shared__ double blk_max[3][4];
    const int wv = threadIdx.x >> 6;
    if (lane == 0) { blk_max[0][wv] = ey; blk_max[1][wv] = yn; blk_max[2][wv] = rho; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ey = fmax(fmax(blk_max[0][0], blk_max[0][1]), fmax(blk_max[0][2], blk_max[0][3]));
        yn = fmax(fmax(blk_max[1][0], blk_max[1][1]), fmax(blk_max[1][2], blk_max[1][3]));
        rho = fmax(fmax(blk_max[2][0], blk_max[2][1]), fmax(blk_max[2][2], blk_max[2][3]));
        atomic_max_pos(params + HP_EY, ey);
        atomic_max_pos(params + HP_YHATMAX, yn);
        atomic_max_pos(params + HP_RHO, rho);
    }
}

// ---------------------------------------------------------------------------
// queries -> fp16 rows Xh[nq_pad][16*KST] (x' = [x^, 1, 1, 1, 0..]) + qinfo[q] = {e_x, |x^|^2}
// One lane per (query, 8-element segment); lane = r*G + f, so loads and stores are contiguous.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void f16_pack_queries_kernel(const double* __restrict__ X, int64_t nq, int64_t nq_pad,
                                                               int D, int KST, const double* __restrict__ center,
                                                               const double* __restrict__ params,
                                                               _Float16* __restrict__ Xh, double* __restrict__ qinfo)
{
    const int G = 2 * KST, R = 64 / G;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int r = lane / G, f = lane - r * G;
    const bool on = r < R;
    const int64_t q = wave * R + r;
    const bool inrange = on && q < nq_pad;
    const bool live = inrange && q < nq;
    const double s = params[HP_SCALE];
    double err2 = 0.0, n2 = 0.0;
    v8h v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * f + e;
        _Float16 xv = (_Float16)0.0f;
        if (live) {
            if (k < D) {
                const double t = (X[q * (int64_t)D + k] - center[k]) * s;
                xv = (_Float16)t;
                err2 = fma(t - (double)xv, t - (double)xv, err2);
                n2 = fma((double)xv, (double)xv, n2);
            } else if (k < D + f16_norm_pieces(D)) {
                xv = (_Float16)1.0f;
            }
        }
        v[e] = xv;
    }
    double e_tot = 0.0, n_tot = 0.0;
    for (int m = 0; m < G; ++m) {
        e_tot += __shfl(err2, r * G + m, 64);
        n_tot += __shfl(n2, r * G + m, 64);
    }
    if (inrange) {
        *reinterpret_cast<v8h*>(Xh + q * (int64_t)(16 * KST) + 8 * f) = v;
        if (f == 0) {
            qinfo[2 * q + 0] = sqrt(e_tot);
            qinfo[2 * q + 1] = n_tot;
        }
    }
}

}  // namespace mce
