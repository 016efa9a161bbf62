// knn_mfma.hpp -- tiled brute-force fp64 k-nearest-neighbour search on CDNA4 (gfx950).
//
// Replaces `NearestNeighbors(...).fit(Y).kneighbors(X)` of the reference
// (MCEvidence.py:1093-1104).  Design:
//
//  * d2(x,y) = |x|^2 + s(x,y),  s = |y|^2 - 2 x.y.  |x|^2 is constant per query, so
//    the ranking is done on s.  With the augmented vectors
//        x' = [x, 1, 0..]   y' = [-2y, |y|^2, 0..]       (D+1 padded to 4*KS)
//    s is one dot product, taken through v_mfma_f64_16x16x4_f64:
//    A = 16 reference rows, B = 16 query rows, KS chained MFMAs per 16x16 tile.
//  * The reference set is pre-packed ONCE into MFMA A-fragment order
//    (pack_refs_kernel) so a fragment is 64 consecutive doubles: staging
//    global->LDS is a straight 16-byte-per-lane copy and the LDS->register read is
//    a conflict-free ds_read_b64 at base + lane*8.
//  * One workgroup = 8 waves x 2 query tiles = 256 queries; the query fragments
//    live in registers for the whole kernel; all 8 waves share the LDS-staged
//    reference chunk (double buffered, one barrier per chunk).
//  * C/D layout of the f64 MFMA: lane l holds column (l&15) = ONE query and rows
//    (l>>4)+4r = four references.  So every lane owns a private running top-K
//    (sorted, in registers) for its query over a quarter of the references;
//    a candidate is compared against the lane's current K-th best (threshold
//    gate) and the insertion network only runs when some lane of the wave passes.
//  * The 4 lane-lists (x rsplit reference splits) of a query are merged by
//    merge_lists_kernel, which also converts s -> distance (and, in the fused
//    path, feeds the volume/weight reduction directly).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mce {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int kWaves = 8;               // waves per workgroup (2 per SIMD)
constexpr int kThreads = kWaves * 64;   // 512
constexpr int kLaneLists = 4;           // lane-lists per query (l>>4)

// 16-query tiles per wave: 2 when the register budget (256 VGPRs at 2 waves/SIMD) allows.
// estimate: per tile 2*KS (query fragment) + 3*KCAP (list) + 10, plus 2*KS + 30 shared.
__host__ __device__ constexpr int pick_qt(int KS, int KCAP) { return (2 * (2 * KS + 3 * KCAP + 10) + 2 * KS + 30 <= 240) ? 2 : 1; }
__host__ __device__ constexpr int queries_per_block(int QT) { return kWaves * QT * 16; }

// reference tiles (16 rows) per LDS chunk: ~32 KB per buffer
__host__ __device__ constexpr int chunk_tiles(int KS) { return (64 / KS) < 1 ? 1 : (64 / KS); }

// ---------------------------------------------------------------------------
// sorted insertion into a register-resident ascending list (static indexing only)
// v = +inf leaves the list untouched (used as the per-lane predicate).
// ---------------------------------------------------------------------------
template <int KCAP>
__device__ __forceinline__ void list_insert(double (&d)[KCAP], int (&id)[KCAP], double v, int j)
{
    bool c_hi = v < d[KCAP - 1];
#pragma unroll
    for (int i = KCAP - 1; i >= 1; --i) {
        const bool c_lo = v < d[i - 1];
        d[i] = c_lo ? d[i - 1] : (c_hi ? v : d[i]);
        id[i] = c_lo ? id[i - 1] : (c_hi ? j : id[i]);
        c_hi = c_lo;
    }
    d[0] = c_hi ? v : d[0];
    id[0] = c_hi ? j : id[0];
}

// ---------------------------------------------------------------------------
// the search kernel
//   grid.x = nqblk * rsplit ; block b -> query block b % nqblk, reference split b / nqblk
//   part_d / part_i : [L = rsplit*4][KCAP][nq_pad]   (s-space keys, int32 reference rows)
// ---------------------------------------------------------------------------
template <int KS, int KCAP, int kQT>
__global__ __launch_bounds__(kThreads, 2) void knn_mfma_kernel(
    const double* __restrict__ Yf, int64_t nchunk_total, int rsplit,
    const double* __restrict__ X, int64_t nq, int D, int64_t nq_pad, int nqblk,
    int self_exclude, int64_t self_offset,
    double* __restrict__ part_d, int* __restrict__ part_i)
{
    constexpr int CT = chunk_tiles(KS);
    constexpr int kQPB = queries_per_block(kQT);
    constexpr int CHUNK_DOUBLES = CT * KS * 64;
    constexpr int CHUNK_VEC = CHUNK_DOUBLES / 2;                  // 16-byte vectors
    constexpr int VPT = (CHUNK_VEC + kThreads - 1) / kThreads;    // vectors per thread
    constexpr int LDS_CHUNK_DOUBLES = VPT * kThreads * 2;          // padded LDS image of a chunk
    extern __shared__ __attribute__((aligned(16))) double lds[];   // 2 * LDS_CHUNK_DOUBLES

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qblk = blockIdx.x % nqblk;
    const int split = blockIdx.x / nqblk;

    // chunk range of this reference split
    const int64_t cps = (nchunk_total + rsplit - 1) / rsplit;
    const int64_t c_begin = (int64_t)split * cps;
    int64_t c_end = c_begin + cps;
    if (c_end > nchunk_total) c_end = nchunk_total;

    // ---- query fragments (B operand), resident in registers -----------------
    const int64_t q0 = (int64_t)qblk * kQPB + wave * (kQT * 16) + (lane & 15);
    double b[kQT][KS];
    int selfj[kQT];
#pragma unroll
    for (int qt = 0; qt < kQT; ++qt) {
        const int64_t q = q0 + qt * 16;
        const bool live = q < nq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int dim = 4 * ks + (lane >> 4);
            double v = 0.0;
            if (live) v = (dim < D) ? X[q * (int64_t)D + dim] : (dim == D ? 1.0 : 0.0);
            b[qt][ks] = v;
        }
        selfj[qt] = (self_exclude && live) ? (int)(self_offset + q) : -1;
    }

    // ---- per-lane running top-K ---------------------------------------------
    const double INF = __builtin_huge_val();
    double ld[kQT][KCAP];
    int li[kQT][KCAP];
    double thr[kQT];
#pragma unroll
    for (int qt = 0; qt < kQT; ++qt) {
#pragma unroll
        for (int k = 0; k < KCAP; ++k) { ld[qt][k] = INF; li[qt][k] = -1; }
        thr[qt] = INF;
    }

    const v4d zero4 = {0.0, 0.0, 0.0, 0.0};

    // ---- staging: async global -> LDS copies (no VGPR round trip).  The LDS image of a
    // chunk is the packed global image, so lane l of wave w writes base(w,i) + l*16.
    // Tail vectors (e >= CHUNK_VEC) re-read the last vector into LDS padding.
    auto stage_async = [&](int64_t c, int buf) {
        const char* src = reinterpret_cast<const char*>(Yf + c * (int64_t)CHUNK_DOUBLES);
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int e = tid + i * kThreads;
            const int es = e < CHUNK_VEC ? e : CHUNK_VEC - 1;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + (size_t)es * 16),
                (__attribute__((address_space(3))) void*)(lds + buf * LDS_CHUNK_DOUBLES + (size_t)(wave * 64 + i * kThreads) * 2),
                16, 0, 0);
        }
    };

    if (c_begin < c_end) stage_async(c_begin, 0);
    __syncthreads();

    for (int64_t c = c_begin; c < c_end; ++c) {
        const int buf = (int)((c - c_begin) & 1);
        const bool more = (c + 1) < c_end;
        if (more) stage_async(c + 1, buf ^ 1);  // DMA in flight under the MFMAs

        const double* lbuf = lds + buf * LDS_CHUNK_DOUBLES + lane;
        const int jchunk = (int)(c * (CT * 16)) + (lane >> 4);
#pragma unroll 1
        for (int t = 0; t < CT; ++t) {
            double a[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a[ks] = lbuf[(t * KS + ks) * 64];
            v4d acc[kQT];
#pragma unroll
            for (int qt = 0; qt < kQT; ++qt) acc[qt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[qt][0], zero4, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < KS; ++ks)
#pragma unroll
                for (int qt = 0; qt < kQT; ++qt) acc[qt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[qt][ks], acc[qt], 0, 0, 0);

            bool pass = false;
#pragma unroll
            for (int qt = 0; qt < kQT; ++qt)
#pragma unroll
                for (int r = 0; r < 4; ++r) pass |= acc[qt][r] < thr[qt];
            if (__any(pass)) {
                const int jb = jchunk + t * 16;
#pragma unroll
                for (int qt = 0; qt < kQT; ++qt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = jb + 4 * r;
                        double v = acc[qt][r];
                        if (j == selfj[qt]) v = INF;
                        const bool p = v < thr[qt];
                        if (__any(p)) {
                            list_insert<KCAP>(ld[qt], li[qt], p ? v : INF, j);
                            thr[qt] = ld[qt][KCAP - 1];
                        }
                    }
                }
            }
        }

        __syncthreads();
    }

    // ---- write the lane-lists -------------------------------------------------
    const int L = split * kLaneLists + (lane >> 4);
#pragma unroll
    for (int qt = 0; qt < kQT; ++qt) {
        const int64_t q = q0 + qt * 16;   // < nq_pad by construction
#pragma unroll
        for (int k = 0; k < KCAP; ++k) {
            const int64_t o = ((int64_t)L * KCAP + k) * nq_pad + q;
            part_d[o] = ld[qt][k];
            part_i[o] = li[qt][k];
        }
    }
}

}  // namespace mce
