// knn_mfma.hpp -- tiled brute-force fp64 k-nearest-neighbour search on CDNA4 (gfx950).
//
// Replaces `NearestNeighbors(...).fit(Y).kneighbors(X)` of the reference
// (MCEvidence.py:1093-1104).  Design:
//
//  * d2(x,y) = |x|^2 + |y|^2 - 2 x.y (x, y centred on the reference mean, pack_refs.hpp).
//    With the augmented vectors
//        x' = [x, 1, 0..]   y' = [-2y, |y|^2, 0..]       (D+1 padded to 4*KS)
//    d2 = |x|^2 + x'.y' is KS chained v_mfma_f64_16x16x4_f64 whose C-in is |x|^2:
//    A = 16 reference rows, B = 16 query rows; the accumulator IS the squared distance.
//  * The reference set is pre-packed ONCE into MFMA A-fragment order
//    (pack_refs_kernel) so a fragment is 64 consecutive doubles: staging
//    global->LDS is a straight 16-byte-per-lane DMA (global_load_lds) and the
//    LDS->register read is a conflict-free ds_read_b64 at base + lane*8.
//  * One workgroup = 8 waves x 2 query tiles = 256 queries (KS > 16, i.e. 64 <= d <= 127: one tile, 128); the query fragments
//    live in registers for the whole kernel; all 8 waves share the LDS-staged
//    reference chunk (double buffered, one barrier per chunk).
//  * C/D layout of the f64 MFMA: lane l holds column (l&15) = ONE query and rows
//    (l>>4)+4r = four references.  Each lane keeps only its query's current K-th
//    best distance (the threshold) in registers; a candidate is gated by a 32-bit
//    integer compare of its high dword (fp64 VALU shares the DGEMM pipe on gfx950 --
//    tools/mfma_f64_peak.hip -- so the gate must not use it).
//  * The running top-K of every query is ONE sorted list in LDS.  Candidates that pass
//    the gate are rare (K ln(N/K) per query out of N), so they are inserted one at a
//    time by the whole wave: lane i owns list slot i, reads slots i and i-1, and
//    writes the shifted/inserted value -- ~30 instructions per accepted candidate,
//    instead of a 64-lane-wide register insertion network per event.
//  * The tile loop is software-pipelined two tiles deep: the MFMAs of tile t+1 are
//    issued before the gate of tile t, so the matrix pipe always has queued work.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mce {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int kWaves = 8;               // waves per workgroup (2 per SIMD)
constexpr int kThreads = kWaves * 64;   // 512

// 16-query tiles per wave: two; one for KS > 16 (64 <= d <= 127, round 5: the query fragments alone are 4 KS registers per tile)
__host__ __device__ constexpr int mfma_qt(int KS) { return KS > 16 ? 1 : 2; }
__host__ __device__ constexpr int queries_per_block(int qt) { return kWaves * qt * 16; }      // 256 (128)

#ifndef MCE_ABLATE
#define MCE_ABLATE 0          // tools/knn_bench.hip only: 1 = gate never passes, 2 = no gate
#endif
// A-fragments (512 B each) per LDS staging buffer: 32 KB, or 16 KB when the per-query
// lists are long (KCAP > 16) so that staging + lists fit the 160 KB LDS.
// (KS > 16: a tile is 17..32 fragments, and the workgroup holds half the lists: 64 KB buffers, 32 KB with long lists)
__host__ __device__ constexpr int chunk_fragments(int KS, int KCAP) { return KS > 16 ? (KCAP > 16 ? 64 : 128) : (KCAP > 16 ? 32 : 64); }
// reference tiles (16 rows) per chunk: always even (the tile loop is unrolled by two)
__host__ __device__ constexpr int chunk_tiles(int KS, int KCAP)
{
    return ((chunk_fragments(KS, KCAP) / KS) / 2) * 2 < 2 ? 2 : ((chunk_fragments(KS, KCAP) / KS) / 2) * 2;
}
__host__ __device__ constexpr int chunk_vpt(int KS, int KCAP) { return (chunk_tiles(KS, KCAP) * KS * 32 + kThreads - 1) / kThreads; }
// dynamic LDS: two staging buffers + per-query lists (fp64 keys, int32 rows) + read slack
__host__ __device__ constexpr size_t lds_bytes(int KS, int KCAP)
{
    return (size_t)2 * chunk_vpt(KS, KCAP) * kThreads * 16 + (size_t)queries_per_block(mfma_qt(KS)) * KCAP * 12 + 1024;
}

// ---------------------------------------------------------------------------
// the search kernel
//   grid.x = nqblk * rsplit ; block b -> query block b % nqblk, reference split b / nqblk
//   part_d / part_i : [rsplit][KCAP][nq_pad]   (keys = squared distances, int32 reference rows)
// ---------------------------------------------------------------------------
template <int KS, int KCAP>
__global__ __launch_bounds__(kThreads, 2) void knn_mfma_kernel(
    const double* __restrict__ Yf, int64_t nchunk_total, int rsplit,
    const double* __restrict__ X, const double* __restrict__ center, int64_t nq, int D, int64_t nq_pad, int nqblk,
    int self_exclude, int64_t self_offset, int ksel,
    double* __restrict__ part_d, int* __restrict__ part_i)
{
    constexpr int kQT = mfma_qt(KS);
    constexpr int kQPB = queries_per_block(kQT);
    constexpr int CT = chunk_tiles(KS, KCAP);
    static_assert(CT % 2 == 0, "tile loop is unrolled by two");
    constexpr int CHUNK_DOUBLES = CT * KS * 64;
    constexpr int CHUNK_VEC = CHUNK_DOUBLES / 2;                  // 16-byte vectors
    constexpr int VPT = chunk_vpt(KS, KCAP);                      // vectors per thread
    constexpr int LDS_CHUNK_DOUBLES = VPT * kThreads * 2;          // padded LDS image of a chunk
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // LDS map: [2 staging buffers][list keys: kQPB*KCAP doubles][list rows: kQPB*KCAP ints][slack]
    double* const list_d = lds + 2 * LDS_CHUNK_DOUBLES;
    int* const list_i = reinterpret_cast<int*>(list_d + kQPB * KCAP);

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

    const double INF = __builtin_huge_val();

    // ---- this wave's 32 lists (wave-private: no cross-wave hazards) -----------
    double* const wl_d = list_d + wave * (kQT * 16 * KCAP);
    int* const wl_i = list_i + wave * (kQT * 16 * KCAP);
    for (int e = lane; e < kQT * 16 * KCAP; e += 64) { wl_d[e] = INF; wl_i[e] = -1; }

    // ---- query fragments (B operand) and |x|^2, resident in registers --------
    const int64_t q0 = (int64_t)qblk * kQPB + wave * (kQT * 16) + (lane & 15);
    double b[kQT][KS];
    v4d xn4[kQT];
    int selfj[kQT];
#pragma unroll
    for (int qt = 0; qt < kQT; ++qt) {
        const int64_t q = q0 + qt * 16;
        const bool live = q < nq;
        double part = 0.0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int dim = 4 * ks + (lane >> 4);
            double v = 0.0;
            if (live && dim < D) { v = X[q * (int64_t)D + dim] - center[dim]; part = fma(v, v, part); }
            if (live && dim == D) v = 1.0;
            b[qt][ks] = v;
        }
        // the 4 lanes l, l^16, l^32, l^48 hold the 4 dim-residues of the same query
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        xn4[qt] = v4d{part, part, part, part};
        selfj[qt] = (self_exclude && live) ? (int)(self_offset + q) : -1;
    }

    // per-lane copy of the query's current K-th best (threshold); gate key = its high dword
    double thr[kQT];
#pragma unroll
    for (int qt = 0; qt < kQT; ++qt) thr[qt] = INF;
#if MCE_ABLATE == 1
    const int gate_off = (int)0x80000000;
#endif

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

    // one 16-reference tile: KS A-fragment reads + kQT chains of KS MFMAs, C-in = |x|^2
    auto mfma_tile = [&](const double* lp, v4d (&acc)[kQT]) {
        double a[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[ks] = lp[ks * 64];
#pragma unroll
        for (int qt = 0; qt < kQT; ++qt) acc[qt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[qt][0], xn4[qt], 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < KS; ++ks)
#pragma unroll
            for (int qt = 0; qt < kQT; ++qt) acc[qt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[qt][ks], acc[qt], 0, 0, 0);
    };

    const int k_last = ksel - 1;

    // Whole-wave sorted insertion of ONE accepted candidate (vv, jj) into list `ql`
    // (wave-local list index, wave-uniform).  Lane i owns slot i.  Returns the list's new
    // K-th best.  Ties are ordered by reference row, so the result does not depend on
    // the order candidates arrive in.
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

    // threshold gate + (rare) insertion for one finished tile; jb0 = reference row of the
    // tile's first row (wave-uniform); lane l holds rows jb0 + (l>>4) + 4r, query column l&15.
    auto process = [&](const v4d (&acc)[kQT], int jb0) {
#if MCE_ABLATE == 2
#pragma unroll
        for (int qt = 0; qt < kQT; ++qt) asm volatile("" ::"v"(acc[qt]));
        return;
#endif
        bool pass = false;
#pragma unroll
        for (int qt = 0; qt < kQT; ++qt) {
#if MCE_ABLATE == 1
            const int th = gate_off;
#else
            const int th = __double2hiint(thr[qt]);
#endif
#pragma unroll
            for (int r = 0; r < 4; ++r) pass |= __double2hiint(acc[qt][r]) <= th;
        }
        if (__any(pass)) {
            const int jl = jb0 + (lane >> 4);
#pragma unroll
            for (int qt = 0; qt < kQT; ++qt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double v = acc[qt][r];
                    if (jl + 4 * r == selfj[qt]) v = INF;
                    unsigned long long m = __ballot(v < thr[qt]);
                    while (m) {                                   // wave-uniform, usually 0-1 trips
                        const int src = __builtin_ctzll(m);
                        m &= m - 1;
                        const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
                        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
                        const double vv = __hiloint2double(hi, lo);
                        // an earlier candidate of this batch may have tightened the threshold
                        const int tlo = __builtin_amdgcn_readlane(__double2loint(thr[qt]), src);
                        const int thi = __builtin_amdgcn_readlane(__double2hiint(thr[qt]), src);
                        if (!(vv < __hiloint2double(thi, tlo))) continue;
                        const int jj = jb0 + (src >> 4) + 4 * r;
                        const double t = insert_one(qt * 16 + (src & 15), vv, jj);
                        if ((lane & 15) == (src & 15)) thr[qt] = t;
                    }
                }
            }
        }
    };

    v4d accA[kQT], accB[kQT];
    int jbA = 0, jbB = 0;
#pragma unroll
    for (int qt = 0; qt < kQT; ++qt) accB[qt] = v4d{INF, INF, INF, INF};   // "no pending tile"

    if (c_begin < c_end) stage_async(c_begin, 0);

    for (int64_t c = c_begin; c < c_end; ++c) {
        const int buf = (int)((c - c_begin) & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the DMA has landed (knn_f16.hpp: dma_barrier) ...
        __syncthreads();                                  // ... everybody's has: chunk c is in LDS; chunk c-1's buffer is free
        if ((c + 1) < c_end) stage_async(c + 1, buf ^ 1);   // DMA in flight under the MFMAs

        const double* lbuf = lds + buf * LDS_CHUNK_DOUBLES + lane;
        const int jchunk = (int)(c * (CT * 16));
#pragma unroll 1
        for (int t = 0; t < CT; t += 2) {
            mfma_tile(lbuf + (t * KS) * 64, accA);
            jbA = jchunk + t * 16;
            process(accB, jbB);
            mfma_tile(lbuf + ((t + 1) * KS) * 64, accB);
            jbB = jchunk + (t + 1) * 16;
            process(accA, jbA);
        }
    }
    process(accB, jbB);

    // ---- write this wave's lists: lane -> (query lane&15, slot (lane>>4)+4i) ----
#pragma unroll
    for (int qt = 0; qt < kQT; ++qt) {
        const int64_t q = q0 + qt * 16;   // < nq_pad by construction
        const int ql = qt * 16 + (lane & 15);
        for (int k = lane >> 4; k < KCAP; k += 4) {
            const int64_t o = ((int64_t)split * KCAP + k) * nq_pad + q;
            part_d[o] = wl_d[ql * KCAP + k];
            part_i[o] = wl_i[ql * KCAP + k];
        }
    }
}

}  // namespace mce
