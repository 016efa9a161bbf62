// knn_long.hpp -- the fp64 MFMA sweep for LONG rows: 128 <= d <= 1024, K <= 32 (round 6).
//
// Replaces `NearestNeighbors(...).fit(Y).kneighbors(X)` of the reference (MCEvidence.py:1093-1104) for the rows the other
// MFMA kernels cannot hold: knn_mfma.hpp keeps a wave's query fragments in registers for the whole kernel (4 KS registers per
// 16-query tile: d <= 127), knn_deep.hpp its fp16 ones (d <= 127).  Until round 6 these shapes ran on the vector-FMA kernel
// (knn_generic.hpp: one thread per query, 0.12 of the fp64 vector peak -- 100 k x 100 k x 128 in 278 ms next to 4.4 ms at
// d = 127).  This is its wave-cooperative tile form:
//
//  * the same algebra as knn_mfma.hpp: d2 = |x|^2 + x'.y' with x' = [x, 1, 0..], y' = [-2y, |y|^2, 0..] (both centred on the
//    reference mean), a chain of v_mfma_f64_16x16x4_f64 whose C-in is |x|^2; the lists are chosen on these GEMM-form keys, carry
//    K + 2 entries, and the merge picks the K on EXACT direct-difference distances (reduce_kernels.hpp, REFINE);
//  * the k dimension goes in BLOCKS of at most kLongKSB = 8 k-steps (32 dimensions), all of one length.  A workgroup (8 waves x 2 query tiles = 256 queries)
//    holds the accumulators of a whole CHUNK of CT reference tiles in registers (2 x CT x 8 VGPRs: 128 at CT = 8) and walks the
//    chunk's k blocks: per block, CT x 8 A fragments come through LDS (one DMA stage of CT x 4 KB, double buffered, one barrier
//    per block) and each wave loads ITS 2 x 8 query fragments of the NEXT block from global memory -- each into the register its
//    predecessor just left, right behind the MFMAs that read it -- (packed in B-fragment order by
//    long_pack_queries_kernel: 512 contiguous bytes per fragment; L2 resident -- the workgroup's 256 queries are
//    256 x 8 x (d + 1) bytes) -- 16 x CT MFMAs per wave between barriers, every A fragment feeding two of them;
//  * after the last block the chunk's 2 x CT accumulator tiles ARE squared distances and go through knn_mfma.hpp's gate
//    (integer compare of the high dword against the query's K-th best) and whole-wave list insertion, unchanged.
//
// Flop per pair: 2 * 4 * KSP with KSP = the k-steps padded to equal blocks (d = 128: 35 k-steps for 129 columns -- 5 blocks of 7).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "knn_mfma.hpp"

namespace mce {

constexpr int kLongKSB = 8;                        // k-steps (4 dimensions each) per block
constexpr int kLongQT = 2;                         // 16-query tiles per wave
constexpr int kLongQPB = kWaves * kLongQT * 16;    // 256 queries per workgroup
constexpr int kLongMinDim = 128, kLongMaxDim = 1024;
__host__ __device__ constexpr int long_ct(int KCAP) { return KCAP > 16 ? 4 : 8; }      // reference tiles (16 rows) per chunk
// k-steps of a row (D + 1 columns, four per step), the blocks they go in (at most kLongKSB each, all of one length: d = 128 is 33 steps =
// 5 blocks of 7, not 5 of 8), and the padded total
__host__ __device__ constexpr int long_ks(int D) { return (D + 1 + 3) / 4; }
__host__ __device__ constexpr int long_nkb(int D) { return (long_ks(D) + kLongKSB - 1) / kLongKSB; }
__host__ __device__ constexpr int long_ksb(int D) { return (long_ks(D) + long_nkb(D) - 1) / long_nkb(D); }
__host__ __device__ constexpr int long_ksp(int D) { return long_nkb(D) * long_ksb(D); }
__host__ __device__ constexpr int long_stage_doubles(int KCAP) { return long_ct(KCAP) * kLongKSB * 64; }
__host__ __device__ constexpr size_t long_lds_bytes(int KCAP)
{
    return (size_t)2 * long_stage_doubles(KCAP) * 8 + (size_t)kLongQPB * KCAP * 12 + 1024;
}

struct LongArgs {
    const double* Yf;        // packed references [tile][KSP][64] (pack_refs_kernel with KS = KSP)
    const double* Xf;        // packed queries    [tile][KSP][64] (long_pack_queries_kernel)
    const double* xn;        // [nq_pad] |x - centre|^2
    int64_t nchunk_total;
    int rsplit;
    int KSP;                 // k-steps per packed row = NKB blocks of KSB steps (long_ksp)
    int KSB;                 // k-steps per block, <= kLongKSB
    int64_t nq, nq_pad;
    int nqblk;
    int self_exclude;
    int64_t self_offset;
    int ksel;
    double* part_d;          // [rsplit][KCAP][nq_pad]
    int* part_i;
};

// ---------------------------------------------------------------------------
// the search kernel
//   grid.x = nqblk * rsplit ; block b -> query block b % nqblk, reference split b / nqblk
//   part_d / part_i : [rsplit][KCAP][nq_pad]   (keys = GEMM-form squared distances, int32 reference rows)
// ---------------------------------------------------------------------------
template <int KCAP>
__global__ __launch_bounds__(kThreads, 1) void knn_long_kernel(LongArgs A)
{
    constexpr int CT = long_ct(KCAP);
    constexpr int KSBMAX = kLongKSB;
    constexpr int QT = kLongQT;
    constexpr int STAGE_DOUBLES = long_stage_doubles(KCAP);       // CT * KSB fragments of 64 doubles
    constexpr int STAGE_VEC = STAGE_DOUBLES / 2;                  // 16-byte vectors
    constexpr int VPT = STAGE_VEC / kThreads;                     // vectors per thread at full blocks: 4 (CT = 8), 2 (CT = 4)
    static_assert(STAGE_VEC % kThreads == 0, "a stage is a whole number of vectors per thread");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* const list_d = lds + 2 * STAGE_DOUBLES;
    int* const list_i = reinterpret_cast<int*>(list_d + kLongQPB * KCAP);

    const double* __restrict__ Yf = A.Yf;
    const double* __restrict__ Xf = A.Xf;
    const int KSP = A.KSP;
    const int KSB = A.KSB;                    // (wave-uniform, 1..8)
    const int NKB = KSP / KSB;
    const int run_vec = KSB * 32;             // 16-byte vectors per (tile, block)
    const int stage_vec = CT * run_vec;
    const int64_t nq_pad = A.nq_pad;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qblk = blockIdx.x % A.nqblk;
    const int split = blockIdx.x / A.nqblk;

    const int64_t cps = (A.nchunk_total + A.rsplit - 1) / A.rsplit;
    const int64_t c_begin = (int64_t)split * cps;
    int64_t c_end = c_begin + cps;
    if (c_end > A.nchunk_total) c_end = A.nchunk_total;

    const double INF = __builtin_huge_val();

    // ---- this wave's 32 lists (wave-private) ----
    double* const wl_d = list_d + wave * (QT * 16 * KCAP);
    int* const wl_i = list_i + wave * (QT * 16 * KCAP);
    for (int e = lane; e < QT * 16 * KCAP; e += 64) { wl_d[e] = INF; wl_i[e] = -1; }

    const int64_t q0 = (int64_t)qblk * kLongQPB + wave * (QT * 16) + (lane & 15);
    const int64_t qtile0 = ((int64_t)qblk * kLongQPB + wave * (QT * 16)) >> 4;
    v4d xn4[QT];
    int selfj[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int64_t q = q0 + qt * 16;
        const double n = A.xn[q];                      // (q < nq_pad by construction)
        xn4[qt] = v4d{n, n, n, n};
        selfj[qt] = (A.self_exclude && q < A.nq) ? (int)(A.self_offset + q) : -1;
    }
    double thr[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) thr[qt] = INF;

    // one k block of one chunk, global -> LDS by DMA: the LDS image is [t][ks < KSB][64 doubles], contiguous; a lane's source is its own
    // address (a wave's 64 vectors may straddle two tiles' runs); vectors beyond the stage re-read the last one into padding
    auto stage_async = [&](int64_t c, int kb, int buf) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int e = tid + i * kThreads;
            const int es = e < stage_vec ? e : stage_vec - 1;
            const int t = es / run_vec;
            const int within = es - t * run_vec;
            const double* src = Yf + (((c * CT + t) * (int64_t)KSP + (int64_t)kb * KSB) * 64 + within * 2);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds + buf * STAGE_DOUBLES + (size_t)(wave * 64 + i * kThreads) * 2),
                                             16, 0, 0);
        }
    };

    const int k_last = A.ksel - 1;
    auto insert_one = [&](int ql, double vv, int jj) -> double {
        const double* ld = wl_d + ql * KCAP;
        const int* li = wl_i + ql * KCAP;
        const int i = lane;
        const double e_i = ld[i];                 // lanes >= KCAP read slack / the next list: unused
        const double e_p = ld[i - 1];             // lane 0 reads one slot below: unused
        const int id_i = li[i];
        const int id_p = li[i - 1];
        const bool c_i = (vv < e_i) || (vv == e_i && jj < id_i);
        const bool c_p = (i > 0) && ((vv < e_p) || (vv == e_p && jj < id_p));
        const double n_e = c_p ? e_p : (c_i ? vv : e_i);
        const int n_id = c_p ? id_p : (c_i ? jj : id_i);
        if (i < KCAP) {
            const_cast<double*>(ld)[i] = n_e;
            const_cast<int*>(li)[i] = n_id;
        }
        const int lo = __builtin_amdgcn_readlane(__double2loint(n_e), k_last);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(n_e), k_last);
        return __hiloint2double(hi, lo);
    };
    // threshold gate + (rare) insertion for one finished tile of one query tile; lane l holds rows jb0 + (l>>4) + 4r, query column l&15
    auto process = [&](const v4d& acc, int qt, int jb0) {
        bool pass = false;
        const int th = __double2hiint(thr[qt]);
#pragma unroll
        for (int r = 0; r < 4; ++r) pass |= __double2hiint(acc[r]) <= th;
        if (__any(pass)) {
            const int jl = jb0 + (lane >> 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = acc[r];
                if (jl + 4 * r == selfj[qt]) v = INF;
                unsigned long long m = __ballot(v < thr[qt]);
                while (m) {
                    const int src = __builtin_ctzll(m);
                    m &= m - 1;
                    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
                    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
                    const double vv = __hiloint2double(hi, lo);
                    const int tlo = __builtin_amdgcn_readlane(__double2loint(thr[qt]), src);
                    const int thi = __builtin_amdgcn_readlane(__double2hiint(thr[qt]), src);
                    if (!(vv < __hiloint2double(thi, tlo))) continue;
                    const int jj = jb0 + (src >> 4) + 4 * r;
                    const double t = insert_one(qt * 16 + (src & 15), vv, jj);
                    if ((lane & 15) == (src & 15)) thr[qt] = t;
                }
            }
        }
    };

    if (c_begin < c_end) stage_async(c_begin, 0, 0);
    int buf = 0;
    // this wave's query fragments of the CURRENT block.  They do not depend on the chunk; the fragments of the next block are
    // loaded into the same registers k-step by k-step, each right behind the MFMAs that read it (below): their latency rides under
    // the rest of the block instead of in front of the next one
    // (PREFETCH: measured, same box -- 100 k x 100 k at d = 128 / 160 / 256 with lists of 8 or 16: -2.4 / -2.2 / -6 %; d = 512 +-0, 1024
    // +1 %; with lists of 32 -- four tiles per chunk, half the MFMAs between two loads -- +2 to +8 %: those load at the top of the block)
    constexpr bool PREFETCH = (CT == 8);
    double b[QT][KSBMAX];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int ks = 0; ks < KSBMAX; ++ks) b[qt][ks] = (PREFETCH && ks < KSB) ? Xf[((qtile0 + qt) * (int64_t)KSP + ks) * 64 + lane] : 0.0;
#pragma unroll 1
    for (int64_t c = c_begin; c < c_end; ++c) {
        v4d acc[QT][CT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int t = 0; t < CT; ++t) acc[qt][t] = xn4[qt];
#pragma unroll 1
        for (int kb = 0; kb < NKB; ++kb) {
            const int kb_next = kb + 1 < NKB ? kb + 1 : 0;
            if constexpr (!PREFETCH) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                    for (int ks = 0; ks < KSBMAX; ++ks)
                        b[qt][ks] = ks < KSB ? Xf[((qtile0 + qt) * (int64_t)KSP + (int64_t)kb * KSB + ks) * 64 + lane] : 0.0;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's share of the stage has landed (and its fragments) ...
            __syncthreads();                                    // ... everybody's has; the other buffer is free
            if (kb + 1 < NKB) stage_async(c, kb + 1, buf ^ 1);
            else if (c + 1 < c_end) stage_async(c + 1, 0, buf ^ 1);
            const double* lbuf = lds + buf * STAGE_DOUBLES + lane;
#pragma unroll
            for (int ks = 0; ks < KSBMAX; ++ks) {
                if (ks < KSB) {                                 // (wave-uniform)
                    double a[CT];
#pragma unroll
                    for (int t = 0; t < CT; ++t) a[t] = lbuf[(t * KSB + ks) * 64];
#pragma unroll
                    for (int t = 0; t < CT; ++t)
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) acc[qt][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b[qt][ks], acc[qt][t], 0, 0, 0);
                    if constexpr (PREFETCH) {
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) b[qt][ks] = Xf[((qtile0 + qt) * (int64_t)KSP + (int64_t)kb_next * KSB + ks) * 64 + lane];
                    }
                }
                // (keeps the A fragments of LATER k-steps out of registers: with all 8 x CT reads hoisted the kernel spills)
                __builtin_amdgcn_sched_barrier(0);
            }
            buf ^= 1;
        }
        const int jchunk = (int)(c * (CT * 16));
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) process(acc[qt][t], qt, jchunk + t * 16);
    }

    // ---- write this wave's lists: lane -> (query lane&15, slot (lane>>4)+4i) ----
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int64_t q = q0 + qt * 16;
        const int ql = qt * 16 + (lane & 15);
        for (int k = lane >> 4; k < KCAP; k += 4) {
            const int64_t o = ((int64_t)split * KCAP + k) * nq_pad + q;
            A.part_d[o] = wl_d[ql * KCAP + k];
            A.part_i[o] = wl_i[ql * KCAP + k];
        }
    }
}

}  // namespace mce
