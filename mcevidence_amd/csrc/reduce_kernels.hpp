// reduce_kernels.hpp -- list merge, distance finalisation and the volume/weight
// reduction of the evidence estimator (reference MCEvidence.py:1107-1117).
//
//   dotp[k] = sum_j  pi^(D/2) r_jk^D / Gamma(1+D/2) / w_j * exp(fs_j)
//           = sum_j  sign(w_j) exp( lnC_D + (D/2) ln r_jk^2 - ln |w_j| + fs_j )     (log domain)
//
// The sign keeps the reference's value for a negative weight (volume / weight is a finite signed term there);
// fs_j = -inf (a row whose likelihood is 0) contributes exp(-inf) = 0, r = 0 likewise.
//
// Deterministic: per-workgroup partial sums in a fixed tree, then one fixed-order
// pass over the partials (no floating-point atomics), so results are run-to-run
// reproducible and independent of dispatch order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sym_types.hpp"

namespace mce {

constexpr int kRedThreads = 256;
constexpr int kMaxK = 32;        // == MCE_MAX_K
constexpr int kMaxLists = 16;    // reference splits merged per query

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// sum over the workgroup; result valid in thread 0.  `red` = kRedThreads/64 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double* red)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < kRedThreads / 64; ++i) s += red[i];
    }
    return s;
}

// ---------------------------------------------------------------------------
// merge_lists: one thread per query merges its L sorted per-split lists into the K best
// (keys = GEMM-form squared distances, good to ~1e-16*|x|^2 absolute), then REFINES them:
// the K selected pairs get their exact direct-difference distance sum_i (x_i - y_i)^2 from
// the original rows and are re-sorted -- so reported distances equal an exact search's
// (duplicates give exactly 0, like the reference's KD-tree path) and only the choice
// between candidates that tie to the last bits could differ -- which is why Kc = K + 2 candidates are selected and refined and
// the K nearest BY EXACT DISTANCE reported (round 6: the rows then equal an exact search's too).  (REFINE=false when the lists
// already hold exact distances: the fp16-filter path.)  Optionally feeds the evidence
// reduction directly.
//   part_d/part_i : [L][KCAP][nq_pad]
//   self_mode 1 (include): the entry whose reference row is self_offset+q is
//                          forced to distance 0 and sorts first.
// ---------------------------------------------------------------------------
template <bool WRITE_DIST, bool FUSE_DOTP, bool REFINE>
__global__ __launch_bounds__(kRedThreads) void merge_lists_kernel(
    const double* __restrict__ part_d, const int* __restrict__ part_i, int L, int KCAP,
    int64_t nq, int64_t nq_pad, const double* __restrict__ X, const double* __restrict__ Y, int D, int K, int Kc,
    int self_mode, int64_t self_offset,
    double* __restrict__ dist, int64_t* __restrict__ idx, int ld_out,
    int k0, int kmax, const double* __restrict__ w, const double* __restrict__ fs, double lnc,
    double* __restrict__ partial, const int* __restrict__ qperm, int part, int nparts, int qpb, const int* __restrict__ border, int nqblk,
    int64_t col0, int64_t col1)
{
    __shared__ double red[kRedThreads / 64];
    // nparts > 1: this launch covers the query blocks part, part + nparts, ... (qpb list columns each) of a
    // pruned search; the thread index is the compact position among them.  col0, col1: the launch covers the list
    // columns [col0, col1) (one rank's blocks of a symmetric partition; the whole set: 0, nq)
    int64_t q = col0 + (int64_t)blockIdx.x * kRedThreads + threadIdx.x;
    if (q >= col1) q = nq;
    if (nparts > 1 && border) {
        const int64_t slot = (q / qpb) * nparts + part;          // position in the dispatch order
        q = slot < nqblk ? (int64_t)border[slot] * qpb + q % qpb : nq;
    }
    const bool live = q < nq;
    // list column q belongs to caller row qo (pruned search: queries were reordered; the lists
    // already carry caller row numbers for the references)
    const int64_t qo = (live && qperm) ? (int64_t)qperm[q] : q;
    const double INF = __builtin_huge_val();

    double term[kMaxK];
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) term[k] = 0.0;

    if (live) {
        const int selfj = (self_mode == 1) ? (int)(self_offset + qo) : -1;
        double base = 0.0, sgn = 1.0;
        if (FUSE_DOTP) {
            const double wq = w[qo];
            base = lnc - log(fabs(wq)) + fs[qo];
            sgn = wq < 0.0 ? -1.0 : 1.0;
        }

        if (L == 1 && !REFINE && self_mode != 1) {
            // ONE list of exact distances per query (pruned walk, symmetric sweep, unsplit exhaustive sweep): it IS the result,
            // ascending with ties by row -- no merge, and none of the indexed scratch arrays of the general path below (272 B
            // of private memory per thread; C5's 10 M columns: 2.05 -> 0.79 ms)
            for (int k = 0; k < K; ++k) {
                const int64_t o = (int64_t)k * nq_pad + q;
                const int i = k < KCAP ? part_i[o] : -1;
                const double d2 = i >= 0 ? fmax(part_d[o], 0.0) : INF;
                if (WRITE_DIST) {
                    dist[qo * (int64_t)ld_out + k] = sqrt(d2);
                    if (idx) idx[qo * (int64_t)ld_out + k] = (int64_t)i;
                }
                if (FUSE_DOTP) {
                    const double t = sgn * exp(base + 0.5 * (double)D * log(d2));
#pragma unroll
                    for (int kk = 0; kk < kMaxK; ++kk)
                        if (kk == k) term[kk] = t;
                }
            }
        } else {
        unsigned char head[kMaxLists];
        for (int l = 0; l < L; ++l) head[l] = 0;

        // ---- select: L-way merge on the approximate keys ----------------------
        double sel_d[kMaxK];
        int sel_i[kMaxK];
        int nsel = 0;
        // SELF_INCLUDE: the own row is a reference row by contract; it is reported first, at distance 0, whether or not
        // it made it into a list (K or more exact duplicates with lower row numbers push it out of one) -- so the result
        // does not depend on how many lists the plan happened to use
        if (selfj >= 0) { sel_d[0] = -1.0; sel_i[0] = selfj; nsel = 1; }     // (-1: sentinel, sorts first, reported as 0)
        // (REFINE: Kc >= K candidates are selected on the keys and refined; the K nearest by EXACT distance are reported)
        for (int k = nsel; k < (REFINE ? Kc : K); ++k) {
            double bv = INF;
            int bi = 0x7fffffff, bl = -1;
            for (int l = 0; l < L; ++l) {
                if (head[l] < KCAP && part_i[((int64_t)l * KCAP + head[l]) * nq_pad + q] == selfj && selfj >= 0) head[l]++;      // already taken
                const int h = head[l];
                if (h >= KCAP) continue;
                const int64_t o = ((int64_t)l * KCAP + h) * nq_pad + q;
                const int i = part_i[o];
                if (i < 0) continue;                       // list exhausted
                const double v = part_d[o];
                if (v < bv || (v == bv && i < bi)) { bv = v; bi = i; bl = l; }
            }
            if (bl < 0) break;
            head[bl]++;
            sel_d[nsel] = bv;
            sel_i[nsel++] = bi;
        }
        // ---- refine: exact direct-difference distances of the selected pairs ----
        // (REFINE=false: the lists already hold exact distances -- fp16-filter path)
        const double* x = X + qo * (int64_t)D;
        for (int k = 0; REFINE && k < nsel; ++k) {
            const double* y = Y + (int64_t)sel_i[k] * D;
            double s2 = 0.0;
            for (int i = 0; i < D; ++i) { const double t = x[i] - y[i]; s2 = fma(t, t, s2); }
            sel_d[k] = (sel_i[k] == selfj) ? -1.0 : s2;      // own row: sentinel, sorts first, reported as 0
        }
        // ---- re-sort (insertion sort; K <= 32), ties by reference row; own row first ----
        for (int k = 1; REFINE && k < nsel; ++k) {
            const double dv = sel_d[k];
            const int iv = sel_i[k];
            int p = k;
            while (p > 0 && (sel_d[p - 1] > dv || (sel_d[p - 1] == dv && sel_i[p - 1] > iv))) {
                sel_d[p] = sel_d[p - 1];
                sel_i[p] = sel_i[p - 1];
                --p;
            }
            sel_d[p] = dv;
            sel_i[p] = iv;
        }
        for (int k = 0; k < K; ++k) {
            const double d2 = (k < nsel) ? fmax(sel_d[k], 0.0) : INF;
            if (WRITE_DIST) {
                dist[qo * (int64_t)ld_out + k] = sqrt(d2);
                if (idx) idx[qo * (int64_t)ld_out + k] = (k < nsel) ? (int64_t)sel_i[k] : (int64_t)-1;
            }
            if (FUSE_DOTP) {
                // column k of the K = kmax-k0 true neighbours <-> reference column k0+k
                const double t = sgn * exp(base + 0.5 * (double)D * log(d2));
#pragma unroll
                for (int kk = 0; kk < kMaxK; ++kk)
                    if (kk == k) term[kk] = t;
            }
        }
        }
    }

    if (FUSE_DOTP) {
        const int ncol = kmax - k0;
#pragma unroll
        for (int kk = 0; kk < kMaxK; ++kk) {
            if (kk < ncol) {
                const double s = block_sum(term[kk], red);
                if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * kmax + k0 + kk] = s;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// pruned walk, heavy waves (knn_f16.hpp): the extra sub-waves of the first `nheavy` waves of the launch's dispatch
// order left their lists in side arrays hv_d / hv_i [S - 1][KCAP][nheavy * qpb] (qpb = the 64 queries of a wave; border[k]
// = block * 8 + wave, so wave k's columns start at border[k] * qpb); fold them into those columns of part_d / part_i
// (ascending, ties by row -- the lists' own order).  Every sub-wave multiplied the same bootstrap tiles, so the same
// (distance, row) may sit in several lists: it is taken once.  One thread per heavy query.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kRedThreads) void prune_heavy_fold_kernel(double* __restrict__ part_d, int* __restrict__ part_i, int64_t nq_pad, int KCAP,
                                                                       const double* __restrict__ hv_d, const int* __restrict__ hv_i, int nheavy, int S, int qpb,
                                                                       const int* __restrict__ border, int qblk0, int qblk_stride)
{
    const int64_t hcols = (int64_t)nheavy * qpb;
    const int64_t hc = (int64_t)blockIdx.x * kRedThreads + threadIdx.x;
    if (hc >= hcols) return;
    const int64_t q = (int64_t)border[qblk0 + (int)(hc / qpb) * qblk_stride] * qpb + hc % qpb;
    double d[kMaxK];
    int r[kMaxK];
    for (int k = 0; k < KCAP; ++k) { d[k] = part_d[(int64_t)k * nq_pad + q]; r[k] = part_i[(int64_t)k * nq_pad + q]; }
    for (int s = 0; s < S - 1; ++s)
        for (int k = 0; k < KCAP; ++k) {
            const int iv = hv_i[((int64_t)s * KCAP + k) * hcols + hc];
            if (iv < 0) break;                                   // list exhausted
            const double dv = hv_d[((int64_t)s * KCAP + k) * hcols + hc];
            // beyond the current last entry (or equal to it): nothing to do -- and neither for the rest of this sorted list
            if (r[KCAP - 1] >= 0 && (dv > d[KCAP - 1] || (dv == d[KCAP - 1] && iv >= r[KCAP - 1]))) break;
            int p = KCAP - 1;
            bool dup = false;
            while (p > 0 && !(r[p - 1] >= 0 && (d[p - 1] < dv || (d[p - 1] == dv && r[p - 1] <= iv)))) --p;      // first position whose predecessor sorts before (dv, iv)
            if (p > 0 && r[p - 1] == iv && d[p - 1] == dv) dup = true;
            if (dup) continue;
            for (int m = KCAP - 1; m > p; --m) { d[m] = d[m - 1]; r[m] = r[m - 1]; }
            d[p] = dv;
            r[p] = iv;
        }
    for (int k = 0; k < KCAP; ++k) { part_d[(int64_t)k * nq_pad + q] = d[k]; part_i[(int64_t)k * nq_pad + q] = r[k]; }
}

// ---------------------------------------------------------------------------
// symmetric sweep (knn_f16.hpp): fold the row-side candidates of one query block -- its bucket -- into the
// block's lists.  One thread per query keeps its list in registers; the bucket is taken in batches through LDS:
// every entry is linked onto its row's chain (atomic exchange on the chain head), then every thread walks its
// chain with the same insertion network as the sweep (ascending distance, ties by caller row: the result does not
// depend on the order in which candidates arrived).  Blocks whose bucket overflowed were searched again
// exhaustively (their lists are complete) and are skipped.
//   part_d / part_i : [KCAP][nq_pad], updated in place
// ---------------------------------------------------------------------------
constexpr int kSymMergeThreads = 512;      // == queries per block of the filter kernels
constexpr int kSymMergeBatch = 2048;
template <int KCAP>
__global__ __launch_bounds__(kSymMergeThreads) void sym_merge_kernel(double* __restrict__ part_d, int* __restrict__ part_i, int64_t nq_pad,
                                                                     const int* __restrict__ bucket_cnt, const int* __restrict__ bucket_flag,
                                                                     const SymEntry* __restrict__ bucket, int cap, int b0, int bstride)
{
    __shared__ double e_d[kSymMergeBatch];
    __shared__ int e_i[kSymMergeBatch];
    __shared__ int e_nx[kSymMergeBatch];
    __shared__ int head[kSymMergeThreads];
    const int b = b0 + bstride * blockIdx.x, t = threadIdx.x;      // (b0, bstride: the launch's blocks -- one rank's range of a partition: stride 1; its every W-th block: W)
    if (bucket_flag[b]) return;
    int n = bucket_cnt[b];
    if (n <= 0) return;
    if (n > cap) n = cap;
    const double INF = __builtin_huge_val();
    const int64_t q = (int64_t)b * kSymMergeThreads + t;
    double own_d[KCAP];
    int own_i[KCAP];
#pragma unroll
    for (int k = 0; k < KCAP; ++k) {
        own_d[k] = part_d[(int64_t)k * nq_pad + q];
        own_i[k] = part_i[(int64_t)k * nq_pad + q];
    }
    const SymEntry* const mine = bucket + (int64_t)b * cap;
    for (int e0 = 0; e0 < n; e0 += kSymMergeBatch) {
        const int m = n - e0 < kSymMergeBatch ? n - e0 : kSymMergeBatch;
        head[t] = -1;
        __syncthreads();
        for (int e = t; e < m; e += kSymMergeThreads) {
            const SymEntry en = mine[e0 + e];
            e_d[e] = en.d2;
            e_i[e] = en.src;
            e_nx[e] = atomicExch(&head[en.row - b * kSymMergeThreads], e);
        }
        __syncthreads();
        for (int cur = head[t]; cur >= 0; cur = e_nx[cur]) {
            const double d2 = e_d[cur];
            const int j = e_i[cur];
            bool c_hi = (d2 < own_d[KCAP - 1]) || (d2 == own_d[KCAP - 1] && j < own_i[KCAP - 1] && d2 < INF);
#pragma unroll
            for (int k = KCAP - 1; k >= 1; --k) {
                const bool c_lo = (d2 < own_d[k - 1]) || (d2 == own_d[k - 1] && j < own_i[k - 1] && d2 < INF);
                own_d[k] = c_lo ? own_d[k - 1] : (c_hi ? d2 : own_d[k]);
                own_i[k] = c_lo ? own_i[k - 1] : (c_hi ? j : own_i[k]);
                c_hi = c_lo;
            }
            own_d[0] = c_hi ? d2 : own_d[0];
            own_i[0] = c_hi ? j : own_i[0];
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < KCAP; ++k) {
        part_d[(int64_t)k * nq_pad + q] = own_d[k];
        part_i[(int64_t)k * nq_pad + q] = own_i[k];
    }
}

// ---------------------------------------------------------------------------
// dotp from an explicit distance matrix (the unfused path, MCEvidence.py:1107-1117)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kRedThreads) void dotp_partial_kernel(
    const double* __restrict__ dist, int64_t nq, int ld, int k0, int kmax, int D, double lnc,
    const double* __restrict__ w, const double* __restrict__ fs, double* __restrict__ partial)
{
    __shared__ double red[kRedThreads / 64];
    const int64_t q = (int64_t)blockIdx.x * kRedThreads + threadIdx.x;
    const bool live = q < nq;
    double base = 0.0, sgn = 1.0;
    if (live) {
        const double wq = w[q];
        base = lnc - log(fabs(wq)) + fs[q];
        sgn = wq < 0.0 ? -1.0 : 1.0;
    }
    for (int k = k0; k < kmax; ++k) {
        double t = 0.0;
        if (live) {
            const double r = dist[q * (int64_t)ld + k];
            t = sgn * exp(base + (double)D * log(r));
        }
        const double s = block_sum(t, red);
        if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * kmax + k] = s;
    }
}

// final pass: block k sums partial[:, k] in a fixed order.
__global__ __launch_bounds__(kRedThreads) void dotp_final_kernel(
    const double* __restrict__ partial, int64_t nblocks, int k0, int kmax, double* __restrict__ dotp)
{
    __shared__ double red[kRedThreads / 64];
    const int k = blockIdx.x;
    if (k < k0) {
        if (threadIdx.x == 0) dotp[k] = 0.0;
        return;
    }
    double acc = 0.0;
    for (int64_t b = threadIdx.x; b < nblocks; b += kRedThreads) acc += partial[b * kmax + k];
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) dotp[k] = s;
}

}  // namespace mce
