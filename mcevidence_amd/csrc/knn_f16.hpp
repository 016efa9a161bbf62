// knn_f16.hpp -- exact fp64 k-nearest-neighbour search with an fp16-MFMA pre-filter.
//
// Same contract as knn_mfma.hpp (reference MCEvidence.py:1093-1104) and the same result
// lists, but the all-pairs sweep runs on v_mfma_f32_32x32x16_f16 (2.5 PFLOP/s dense, 32x
// the fp64 matrix rate) as a RIGOROUS lower-bound filter; only the few candidates that
// survive it (~K ln(N/K) per query) are evaluated exactly, in fp64, by direct differences.
// The neighbours and distances that come out are those of an exact fp64 search.
//
// Filter.  Points are centred on the reference mean and scaled by a power of two s so that
// |.| <= 200 (fp16 range; s is exact).  x^ = fp16(x~), y^ = fp16(y~) elementwise, and per
// point e_x = |x~ - x^|, e_y = |y~ - y^| are computed in fp64 from the converted values.
// Triangle inequality:   |x~ - y~|  >=  |x^ - y^| - e_x - e_y.
// The MFMA evaluates  A = |y^|^2 - 2 x^.y^  from the augmented rows
//      y' = [-2 y^_0.., n_hi, n_mid, n_lo, 0..]     x' = [x^_0.., 1, 1, 1, 0..]
// (|y^|^2 split into three fp16 pieces = 33 bits; products of two fp16 are exact in fp32;
// fp32 accumulation error <= 32*KST*2^-24 * (|x^|+max|y^|)^2 =: eps_q).  A candidate can be
// among the query's K nearest only if  |x~-y~|^2 <= s^2 thr, hence only if
//      A  <=  (s sqrt(thr) + e_x + max_j e_y)^2 - |x^|^2 + eps_q  =: G_q      (rounded up, fp32)
// Lane gate:  min over the lane's 16 accumulators  <=  G_q   (8 v_min3_f32 + 1 compare).
//
// Survivors are queued per wave in LDS and drained in batches by the whole workgroup at a
// chunk boundary: phase A, 8 lanes share one queued (query,row) pair, read the two ORIGINAL fp64
// rows in 64-byte segments, sum the exact squared distance and link the entry into its query's
// chain (24 pairs per trip, all loads in flight before the first use); phase B,
// lane l -- which OWNS wave-local query l and keeps its sorted top-K list in registers --
// walks its chain and applies a static compare/select insertion network.  Then the gates
// G_q are refreshed from the owners' K-th best.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef MCE_STATS
#define MCE_STATS 0    // tools/knn_f16_bench.hip only: per-wave clock64/event counters appended to `params`
#endif
#ifndef MCE_ABLATE
#define MCE_ABLATE 0   // tools/knn_f16_bench.hip only: 1 = gate never passes, 3 = also no barriers (results invalid)
#endif

namespace mce {

// v_min3_f32 without the NaN-canonicalising v_max the compiler adds around fminf()
__device__ __forceinline__ float min3f(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// Workgroup geometry.  MCE_H_GEOM 0 (shipped): 8 waves (2 per SIMD) x 2 query tiles.  Kept for the
// record, both measured SLOWER at C3 (tools/knn_f16_bench.hip): 1 = 4 waves (one per SIMD, 512
// registers) x 4 query tiles: sweep 83 ms vs 56 ms; 2 = 16 waves (4 per SIMD, 128 VGPRs): spills.
#ifndef MCE_H_GEOM
#define MCE_H_GEOM 0
#endif
constexpr int kHWaves = MCE_H_GEOM == 1 ? 4 : (MCE_H_GEOM == 2 ? 16 : 8);   // GEOM 2: 16 waves (4 per SIMD, <= 128 VGPRs)
constexpr int kHQT = MCE_H_GEOM == 1 ? 4 : 2;      // 32-query tiles per wave
constexpr int kHNL = kHQT / 2;                     // top-K lists per owner lane (64 queries per list set)
constexpr int kHThreads = kHWaves * 64;
#ifndef MCE_H_SETPRIO
#define MCE_H_SETPRIO 0
#endif
#ifndef MCE_H_QUEUE
#define MCE_H_QUEUE 448
#endif
#ifndef MCE_H_TRIGGER
#define MCE_H_TRIGGER 96
#endif
#ifndef MCE_H_STAGE_KB
#define MCE_H_STAGE_KB 48
#endif
constexpr int kHQueue = MCE_H_QUEUE;          // candidate queue entries per wave (16 B each in LDS)
constexpr int kHDrainTrigger = MCE_H_TRIGGER;    // a wave with this many queued candidates asks the workgroup to drain
constexpr int kHRelBits = MCE_H_GEOM == 1 ? 25 : 26;   // queue entry = query-local (6|7 bits) << kHRelBits | row - first row of the split
constexpr double kHTargetRadius = 200.0;

// device-side scalars shared by the f16 kernels (doubles; maxima kept as bit patterns)
enum { HP_RMAX = 0, HP_SCALE = 1, HP_EY = 2, HP_YHATMAX = 3, HP_RHO = 4, HP_COUNT = 8 };

__host__ __device__ constexpr int f16_ksteps(int D) { return (D + 3 + 15) / 16; }          // 16-wide k-steps
__host__ __device__ constexpr int f16_qt(int) { return kHQT; }
__host__ __device__ constexpr bool f16_supported(int D, int K) { return D >= 1 && f16_ksteps(D) <= 4 && K <= 16; }
__host__ __device__ constexpr int f16_qpb(int KCAP) { return kHWaves * f16_qt(KCAP) * 32; }
// 32-row reference tiles per LDS chunk (tile = KST KB): MCE_H_STAGE_KB per buffer, even count,
// and a whole number of 16-byte vectors per thread (CT*KST % 8 == 0)
__host__ __device__ constexpr int f16_chunk_tiles(int KST) { return MCE_H_STAGE_KB / KST; }   // 48 KB: 48, 24, 16, 12 tiles
__host__ __device__ constexpr size_t f16_lds_bytes(int KST, int KCAP)
{
    return (size_t)2 * f16_chunk_tiles(KST) * KST * 1024             // staging
           + (size_t)kHWaves * kHQueue * 16                            // queues: packed(4) + next(4) + d2(8)
           + (size_t)kHWaves * kHQT * 32 * 4 + 64;                     // chain heads + votes
}

// ---------------------------------------------------------------------------
// the filter + exact-refine search kernel
//   grid.x = nqblk * rsplit (as knn_mfma_kernel); lists out: part_d/part_i [rsplit][KCAP][nq_pad]
//   with EXACT squared distances as keys.
// ---------------------------------------------------------------------------
template <int KST, int KCAP>
__global__ __launch_bounds__(kHThreads, MCE_H_GEOM == 1 ? 1 : (MCE_H_GEOM == 2 ? 4 : 2)) void knn_f16_kernel(
    const _Float16* __restrict__ Yh, int64_t nchunk_total, int rsplit,
    const _Float16* __restrict__ Xh, const double* __restrict__ qinfo, const double* __restrict__ params,
    const double* __restrict__ X, const double* __restrict__ Y, int64_t nq, int64_t nr, int D,
    int64_t nq_pad, int nqblk, int self_exclude, int64_t self_offset, int ksel,
    double* __restrict__ part_d, int* __restrict__ part_i)
{
    constexpr int QT = f16_qt(KCAP);
    constexpr int QPW = QT * 32;                         // queries per wave
    constexpr int QPB = kHWaves * QPW;
    constexpr int CT = f16_chunk_tiles(KST);
    static_assert(CT % 2 == 0, "tile loop is unrolled by two");
    constexpr int CHUNK_BYTES = CT * KST * 1024;
    constexpr int CHUNK_VEC = CHUNK_BYTES / 16;
    constexpr int VPT = (CHUNK_VEC + kHThreads - 1) / kHThreads;
    static_assert(CHUNK_VEC % kHThreads == 0, "chunk must be a whole number of 16-byte vectors per thread");
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    // LDS map: [2 staging buffers][queue d2: W*Q f64][queue packed: W*Q i32][queue next: W*Q i32][heads W*64][votes 2]
    char* const stage0 = lds_raw;
    double* const qd2_all = reinterpret_cast<double*>(lds_raw + 2 * CHUNK_BYTES);
    int* const qpk_all = reinterpret_cast<int*>(qd2_all + kHWaves * kHQueue);
    int* const qnx_all = qpk_all + kHWaves * kHQueue;
    int* const head_all = qnx_all + kHWaves * kHQueue;

#if MCE_STATS
    const long long t_kernel0 = clock64();
    long long st_drains = 0, st_enq = 0, st_steps = 0, st_events = 0, st_tA = 0, st_tD = 0, st_tB = 0;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qblk = blockIdx.x % nqblk;
    const int split = blockIdx.x / nqblk;

    const int64_t cps = (nchunk_total + rsplit - 1) / rsplit;
    const int64_t c_begin = (int64_t)split * cps;
    int64_t c_end = c_begin + cps;
    if (c_end > nchunk_total) c_end = nchunk_total;

    const double INF = __builtin_huge_val();
    double* const wqd = qd2_all + wave * kHQueue;               // exact distance of a queued entry (phase A)
    int* const wq = qpk_all + wave * kHQueue;                   // packed (query-local, relative row)
    int* const wnx = qnx_all + wave * kHQueue;                  // next entry of the same query
    int* const whead = head_all + wave * QPW;                   // chain head per wave-local query
    volatile int* const wvote = head_all + kHWaves * QPW;       // [2] drain votes (chunk parity)
    const int jsplit0 = (int)(c_begin * (CT * 32));             // first reference row of this split
#pragma unroll
    for (int nl = 0; nl < kHNL; ++nl) whead[nl * 64 + lane] = -1;

    // lane l OWNS wave-local queries nl*64 + l (query ql = qt*32 + column): their sorted top-KCAP lists live here
    double own_d[kHNL][KCAP];
    int own_i[kHNL][KCAP];
#pragma unroll
    for (int nl = 0; nl < kHNL; ++nl)
#pragma unroll
        for (int k = 0; k < KCAP; ++k) { own_d[nl][k] = INF; own_i[nl][k] = -1; }

    const int64_t qwave0 = (int64_t)qblk * QPB + wave * QPW;     // first query of this wave

    // ---- B fragments (fp16 query rows) + per-query gate constants ---------------
    v8h b[QT][KST];
    float G[QT];
    const double s2 = params[HP_SCALE] * params[HP_SCALE];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int64_t q = qwave0 + qt * 32 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < KST; ++ks)
            b[qt][ks] = *reinterpret_cast<const v8h*>(Xh + q * (int64_t)(16 * KST) + 16 * ks + 8 * (lane >> 5));
        G[qt] = (q < nq && MCE_ABLATE != 1 && MCE_ABLATE != 3) ? __builtin_huge_valf() : -__builtin_huge_valf();     // padding queries never pass
    }
    const int k_last = ksel - 1;
    // gate of query (qt, lane&31) from its current K-th best `thr` (exact squared distance, input
    // units).  The per-query constants are re-read from qinfo (L2) here -- this runs once per drain,
    // and keeping them in registers would cost 6 VGPRs per query tile in the sweep.
    auto gate_of = [&](double thr, int qt) -> float {
        const int64_t q = qwave0 + qt * 32 + (lane & 31);
        if (!(q < nq) || MCE_ABLATE == 1 || MCE_ABLATE == 3) return -__builtin_huge_valf();
        if (!(thr < INF)) return __builtin_huge_valf();
        const double ex = qinfo[2 * q], xn = qinfo[2 * q + 1];
        const double r = sqrt(xn) + params[HP_YHATMAX];
        // + 2*sqrt(16 KST)*2^-14: even if the matrix unit flushed fp16 subnormal inputs (it does not
        // on gfx950) the bound would hold
        const double ga = (ex + params[HP_EY]) * (1.0 + 1e-9) + 2.0 * sqrt(16.0 * KST) * 0x1p-14;
        const double eps = (32.0 * KST) * 0x1p-24 * r * r * (1.0 + 1e-9) + params[HP_RHO] + 1e-30;
        const double rr = sqrt(thr * s2) * (1.0 + 1e-12) + ga;
        const double g = rr * rr * (1.0 + 1e-12) - xn + eps;
        return __double2float_ru(g);
    };

    // ---- staging (global_load_lds DMA, linear image) -----------------------------
    auto stage_async = [&](int64_t c, int buf) {
        const char* src = reinterpret_cast<const char*>(Yh) + c * (int64_t)CHUNK_BYTES;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int e = tid + i * kHThreads;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + (size_t)e * 16),
                (__attribute__((address_space(3))) void*)(stage0 + buf * CHUNK_BYTES + (size_t)(wave * 64 + i * kHThreads) * 16),
                16, 0, 0);
        }
    };

    // A fragments of one 32-row tile: KST 16-byte LDS reads per lane.  They are fetched one
    // tile AHEAD of the MFMAs that consume them, so the LDS latency is never exposed.
    auto load_a = [&](const char* lp, v8h (&a)[KST]) {
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) a[ks] = *reinterpret_cast<const v8h*>(lp + ks * 1024);
    };
    // one 32-row tile: QT chains of KST MFMAs
    auto mfma_tile = [&](const v8h (&a)[KST], v16f (&acc)[QT]) {
#if MCE_H_SETPRIO
        __builtin_amdgcn_s_setprio(1);      // the wave in its MFMA burst wins issue arbitration; its SIMD partner gates meanwhile
#endif
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            v16f z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[qt][0], z, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 1; ks < KST; ++ks)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) acc[qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], b[qt][ks], acc[qt], 0, 0, 0);
#if MCE_H_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };

    int qcount = 0;   // wave-uniform number of queued candidates

    auto drain = [&]() {
#if MCE_STATS
        const long long t_d0 = clock64();
        st_drains += 1; st_enq += qcount;
#endif
        // ---- phase A: exact distances + chain links.  8 lanes share one queued pair and read
        // the two rows in 64-byte segments (a row is fetched once, not once per element).  The
        // gather is latency-bound, so ALL loads of a group of NPASS*8 pairs are issued before the
        // first use (the compiler otherwise serialises the passes: one round trip each).
        const int sub = lane & 7;
#ifndef MCE_H_NPASS
#define MCE_H_NPASS 3
#endif
        constexpr int NPASS = MCE_H_NPASS;                         // 8*NPASS pairs, 8*NPASS loads per lane in flight (D <= 32)
        constexpr int EPL = (16 * KST + 3) / 8 > 4 ? 8 : 4;        // elements per lane: 4 covers D <= 32, 8 covers D <= 61
        for (int b0 = 0; b0 < qcount; b0 += NPASS * 8) {
            int qlp[NPASS], ep[NPASS];
            bool okp[NPASS];
            const double* xp[NPASS];
            const double* yp[NPASS];
#pragma unroll
            for (int u = 0; u < NPASS; ++u) {
                const int e = b0 + u * 8 + (lane >> 3);
                ep[u] = e;
                int ql = 0, j = 0;
                const bool valid = e < qcount;
                if (valid) {
                    const unsigned ent = (unsigned)wq[e];
                    ql = (int)(ent >> kHRelBits);
                    j = jsplit0 + (int)(ent & ((1u << kHRelBits) - 1u));
                }
                qlp[u] = ql;
                const int64_t q = qwave0 + ql;
                okp[u] = valid && j < nr && q < nq && !(self_exclude && (int64_t)j == self_offset + q);
                xp[u] = X + (okp[u] ? q : 0) * (int64_t)D;
                yp[u] = Y + (okp[u] ? (int64_t)j : 0) * D;
            }
            double xv[NPASS][EPL], yv[NPASS][EPL];
#pragma unroll
            for (int u = 0; u < NPASS; ++u)
#pragma unroll
                for (int v = 0; v < EPL; ++v) {
                    const int iv = (sub + 8 * v < D) ? sub + 8 * v : sub;     // clamped load, masked use
                    xv[u][v] = xp[u][iv];
                    yv[u][v] = yp[u][iv];
                }
#pragma unroll
            for (int u = 0; u < NPASS; ++u) {
                double a0 = 0.0;
#pragma unroll
                for (int v = 0; v < EPL; ++v) {
                    const double t = (sub + 8 * v < D) ? xv[u][v] - yv[u][v] : 0.0;
                    a0 = fma(t, t, a0);
                }
                a0 += __shfl_xor(a0, 1, 64);
                a0 += __shfl_xor(a0, 2, 64);
                a0 += __shfl_xor(a0, 4, 64);
                if (okp[u] && sub == 0) {
                    wqd[ep[u]] = a0;
                    wnx[ep[u]] = atomicExch(&whead[qlp[u]], ep[u]);      // push onto the query's chain
                }
            }
        }
#if MCE_STATS
        st_tA += clock64() - t_d0;
#endif
        // ---- phase B: every owner lane folds its chain(s) into its register list(s) ----------
#pragma unroll
        for (int nl = 0; nl < kHNL; ++nl) {
            int cur = whead[nl * 64 + lane];
            whead[nl * 64 + lane] = -1;
            while (__any(cur >= 0)) {
#if MCE_STATS
                st_steps += 1;
#endif
                const bool on = cur >= 0;
                const int ce = on ? cur : 0;
                const double d2 = on ? wqd[ce] : INF;
                const int j = jsplit0 + (int)((unsigned)wq[ce] & ((1u << kHRelBits) - 1u));
                cur = on ? wnx[ce] : -1;
                // ascending list, ties by row; d2 = +inf (idle lane) changes nothing
                bool c_hi = (d2 < own_d[nl][KCAP - 1]) || (d2 == own_d[nl][KCAP - 1] && j < own_i[nl][KCAP - 1] && d2 < INF);
#pragma unroll
                for (int k = KCAP - 1; k >= 1; --k) {
                    const bool c_lo = (d2 < own_d[nl][k - 1]) || (d2 == own_d[nl][k - 1] && j < own_i[nl][k - 1] && d2 < INF);
                    own_d[nl][k] = c_lo ? own_d[nl][k - 1] : (c_hi ? d2 : own_d[nl][k]);
                    own_i[nl][k] = c_lo ? own_i[nl][k - 1] : (c_hi ? j : own_i[nl][k]);
                    c_hi = c_lo;
                }
                own_d[nl][0] = c_hi ? d2 : own_d[nl][0];
                own_i[nl][0] = c_hi ? j : own_i[nl][0];
            }
        }
#if MCE_STATS
        st_tD += clock64() - t_d0;
#endif
        qcount = 0;
        // ---- refresh the gates: lane l needs the K-th best of queries ql = qt*32 + (l&31),
        // owned by lane ql & 63 in list ql >> 6
        double thr_own[kHNL];
#pragma unroll
        for (int nl = 0; nl < kHNL; ++nl) {
            thr_own[nl] = own_d[nl][KCAP - 1];
#pragma unroll
            for (int k = 0; k < KCAP - 1; ++k) thr_own[nl] = (k == k_last) ? own_d[nl][k] : thr_own[nl];
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) G[qt] = gate_of(__shfl(thr_own[qt >> 1], (qt & 1) * 32 + (lane & 31), 64), qt);
    };

    // gate + enqueue for one finished tile; jb0 = first reference row of the tile.
    // C layout of 32x32 f32: lane l -> query column l&31, rows (r&3) + 8*(r>>2) + 4*(l>>5)
    auto process = [&](const v16f (&acc)[QT], int jb0) {
#if MCE_ABLATE == 2
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) asm volatile("" ::"v"(acc[qt]));
        return;
#endif
        bool passq[QT];
        bool pass = false;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const v16f& c = acc[qt];
            float m0 = min3f(c[0], c[1], c[2]);
            float m1 = min3f(c[3], c[4], c[5]);
            float m2 = min3f(c[6], c[7], c[8]);
            float m3 = min3f(c[9], c[10], c[11]);
            float m4 = min3f(c[12], c[13], c[14]);
            m0 = min3f(m0, m1, m2);
            m3 = min3f(m3, m4, c[15]);
            passq[qt] = min3f(m0, m3, m3) <= G[qt];
            pass |= passq[qt];
        }
        if (__any(pass)) {
#if MCE_STATS
            st_events += 1;
#endif
            const int jrel0 = jb0 - jsplit0 + 4 * (lane >> 5);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                if (!__any(passq[qt])) continue;
                // per-lane bit mask of the accumulators under the gate (branch-free) ...
                unsigned pm = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) pm |= (acc[qt][r] <= G[qt]) ? (1u << r) : 0u;
                // ... then one queue entry per lane and trip (usually a single trip)
                unsigned long long m = __ballot(pm != 0);
                while (m) {
                    if (qcount > kHQueue - 64) drain();
                    if (pm != 0) {
                        const int r = __builtin_ctz(pm);
                        pm &= pm - 1;
                        const int slot = qcount + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                        const unsigned rel = (unsigned)(jrel0 + (r & 3) + 8 * (r >> 2));
                        wq[slot] = (int)(((unsigned)(qt * 32 + (lane & 31)) << kHRelBits) | rel);
                    }
                    qcount += __builtin_popcountll(m);
                    m = __ballot(pm != 0);
                }
            }
        }
    };

    v16f accA[QT], accB[QT];
    int jbA = 0, jbB = 0;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[qt][r] = __builtin_nanf("");       // "no pending tile": NaN never passes the gate

    if (c_begin < c_end) stage_async(c_begin, 0);
    if (tid < 2) wvote[tid] = 0;

    // Drains are taken by ALL waves of the workgroup at the same chunk boundary (a wave that
    // drained alone would hold the other seven at the next barrier): before the barrier a
    // wave whose queue is filling raises the vote of this chunk's parity; after the barrier
    // everybody reads it.  (process() still drains locally if its queue would overflow.)
    for (int64_t c = c_begin; c < c_end; ++c) {
        const int buf = (int)((c - c_begin) & 1);
        if (qcount >= kHDrainTrigger && lane == 0) wvote[buf] = 1;
#if MCE_STATS
        const long long t_b0 = clock64();
#endif
#if MCE_ABLATE != 3
        __syncthreads();
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ablation: no barrier (results invalid)
#endif
#if MCE_STATS
        st_tB += clock64() - t_b0;
#endif
        if ((c + 1) < c_end) stage_async(c + 1, buf ^ 1);
        const bool all_drain = wvote[buf] != 0;
        if (tid == 0) wvote[buf ^ 1] = 0;          // re-arm the other parity (read again only after the next barrier)
        if (all_drain) drain();
        const char* lbuf = stage0 + buf * CHUNK_BYTES + lane * 16;
        const int jchunk = (int)(c * (CT * 32));
        v8h a0[KST], a1[KST];
        load_a(lbuf, a0);
#pragma unroll 1
        for (int t = 0; t < CT; t += 2) {
            load_a(lbuf + ((t + 1) * KST) * 1024, a1);
            mfma_tile(a0, accA);
            jbA = jchunk + t * 32;
            process(accB, jbB);
            load_a(lbuf + ((t + 2 < CT ? t + 2 : t) * KST) * 1024, a0);      // (last trip: harmless re-read)
            mfma_tile(a1, accB);
            jbB = jchunk + (t + 1) * 32;
            process(accA, jbA);
        }
    }
    process(accB, jbB);
    drain();

#if MCE_STATS
    if (lane == 0) {
        double* o = const_cast<double*>(params) + 16 + ((int64_t)blockIdx.x * kHWaves + wave) * 8;
        o[0] = (double)st_drains; o[1] = (double)st_enq; o[2] = (double)st_steps; o[3] = (double)st_events;
        o[4] = (double)st_tA; o[5] = (double)st_tD; o[6] = (double)(clock64() - t_kernel0); o[7] = (double)st_tB;
    }
#endif
    // ---- write the lists: lane l owns wave-local queries nl*64 + l (coalesced over lanes) ----
#pragma unroll
    for (int nl = 0; nl < kHNL; ++nl) {
        const int64_t q = qwave0 + nl * 64 + lane;
#pragma unroll
        for (int k = 0; k < KCAP; ++k) {
            const int64_t o = ((int64_t)split * KCAP + k) * nq_pad + q;
            part_d[o] = own_d[nl][k];
            part_i[o] = own_i[nl][k];
        }
    }
}

}  // namespace mce
