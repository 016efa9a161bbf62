// knn_panel.hpp -- the PANEL sweep: the fp16-MFMA filter + exact fp64 refine of knn_f16.hpp (same bound, same lists, same
// results: reference MCEvidence.py:1093-1104) as a dedicated kernel, written around its register and instruction budget.
// It runs the symmetric sweep of auto evidence (knn_f16.hpp, "Symmetric sweep": every pair of rows multiplied once, gated
// for both of its sides) on one GPU, or one rank's share of it (sym_types.hpp, PanelGeom: the tiles of the rank's own range
// of blocks symmetrically, everybody else's column side only), with the work cut into UNITS (panel of reference rows x
// query block) that hand a block's register lists on.  (geom.sym_on = 0 -- every tile column side only, i.e. the exhaustive
// sweep of cross evidence in units -- works and was measured: no faster than knn_f16_kernel's seeded sweep, so the library
// does not take it)
//
// Why a second kernel.  knn_f16_kernel<.., SYM = 2> carries the pruned walk, the two-pass search and the seed phase in one
// body: 256 VGPRs, 104 SGPRs with scalars spilled to VGPR lanes INSIDE the tile loop, six inlined copies of the drain, and a
// per-lane 16-bit mask built with ~40 VALU instructions on every tile that has a candidate -- 13 VALU per MFMA measured
// (rocprofv3, profiles/r02_symmetric) where the gate needs 5.5.  Here (8.2 VALU per MFMA, profiles/r03_panel):
//   * arguments are read through the kernarg pointer WHERE THEY ARE USED (cold paths re-load them), so the tile loop holds a
//     handful of scalars and nothing is spilled;
//   * the two query tiles of a tile share ONE branch, so the tile's MFMAs and its gate form one basic block and interleave;
//   * a tile with a candidate is resolved with wave-uniform compares: the gate's five first-level minima + the sixteenth
//     accumulator, then the three members of a triple that fired -- 9 v_cmp, scalar branches over the rest, ~4 VALU per
//     queued pair;
//   * there is ONE drain in the code, at the end of a chunk; a queue that fills up inside a chunk defers the rest of the
//     tile to a redo list, which the chunk end works off by multiplying those tiles again (the chunk is still in LDS);
//   * the pipeline is flushed at the end of a chunk, so the accumulators are dead while the drain runs.  (Measured on one
//     box against a pipeline kept going across chunk boundaries, flushed only for a drain, redo from global memory:
//     37.5 vs 39.3 ms at C3 -- the carried tile costs the loop more than the flush.)
//   * the wait of a unit for its block's previous unit is bounded; waves that give up flag the block for the repair launch.
#pragma once
#include "knn_f16.hpp"

namespace mce {

struct PanelArgs {
    const _Float16* Yh;          // packed fp16 references (f16_pack_refs_kernel)
    const _Float16* Xh;          // fp16 query rows [nq_pad][16 KST]
    const double* qinfo;         // [nq_pad][2]: e_x, |x^|^2
    const double* params;        // HP_* scalars
    const double* X;             // fp64 query rows [nq, D] (exact refine)
    const double* Y;             // fp64 reference rows [nr, D]
    const int* rperm;            // reference row -> caller's row (nullptr: the row itself)
    double* part_d;              // lists [KCAP][nq_pad], exact squared distances
    int* part_i;
    int64_t nq, nr, nq_pad, self_offset;
    int D, ksel, self_exclude, spin_limit;
    int debug;                   // test hooks (MCE_PANEL_DEBUG): 8 every candidate through the redo list, 16 the odd waves of every block's
                                 // second unit give up waiting at once
    SymParams sym;               // .done, .panel always; the row-side state only with geom.sym_on
    PanelGeom geom;
    const PanelUnit* units = nullptr;   // non-null: the launch's units come from this table (sym_types.hpp) instead of the geometry -- the all-pairs-once partition
    const double* lo_d = nullptr;   // LOWER (second pass of a search for 16 < K <= 32 neighbours): the FIRST pass's lists, [KCAP][nq_pad] in the
    const int* lo_i = nullptr;      // same column order -- every row's own 16 nearest; this pass keeps only what lies beyond a row's 16th
};

#ifndef MCE_PANEL_STATS
#define MCE_PANEL_STATS 0       // tools/knn_sym_bench.hip: per-wave cycle / event counters appended to `params`
#endif
#ifndef MCE_PANEL_NPASS
#define MCE_PANEL_NPASS 3       // drain, phase A: 8 * NPASS pairs in flight per trip
#endif
#ifndef MCE_PANEL_STAGE_KB
#define MCE_PANEL_STAGE_KB MCE_H_STAGE_KB      // KB per staging buffer (two of them)
#endif
#ifndef MCE_PANEL_QUEUE
#define MCE_PANEL_QUEUE MCE_H_QUEUE            // candidate queue entries per wave (16 B each)
#endif
#ifndef MCE_PANEL_TRIGGER
#define MCE_PANEL_TRIGGER 192                  // a wave with this many queued candidates asks the workgroup to drain (96 -> 192: 35.4 -> 35.1 ms at C3)
#endif
#ifndef MCE_PANEL_ABL
#define MCE_PANEL_ABL 0         // tools only: 1 = the gates never pass, 2 = no gate at all (results invalid)
#endif

// 32-row reference tiles per staged chunk (tile = KST KB): even, and a whole number of 16-byte vectors per thread
__host__ __device__ constexpr int panel_chunk_tiles(int KST) { return MCE_PANEL_STAGE_KB / KST / 2 * 2; }
constexpr int kPanelQueue = MCE_PANEL_QUEUE;
__host__ __device__ constexpr size_t panel_lds_bytes(int KST)
{
    return (size_t)2 * panel_chunk_tiles(KST) * KST * 1024           // staging
           + (size_t)kHWaves * kPanelQueue * 16                        // queues: d2 (8) + packed (4) + next (4)
           + (size_t)kHWaves * kHQT * 32 * 4 + 128                     // chain heads + votes
           + (size_t)kHWaves * kHQT * 32 * 4                           // K-th bound per query as of the last drain (float, rounded up)
           + (size_t)kHWaves * panel_chunk_tiles(KST) * kHQT * 4;      // redo list
}

// pointers read out of the argument block are generic to the compiler; these say what they are (global memory), so that
// the accesses are global_load / global_store / global_atomic and not FLAT ones
template <class T> __device__ __forceinline__ const __attribute__((address_space(1))) T* gptr(const T* p)
{
    return (const __attribute__((address_space(1))) T*)p;
}
template <class T> __device__ __forceinline__ __attribute__((address_space(1))) T* gptr_w(T* p)
{
    return (__attribute__((address_space(1))) T*)p;
}

__device__ __forceinline__ float vmaxf(float a, float b)       // v_max_f32 without the canonicalising v_max fmaxf() adds
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// LOWER (round 5): the second pass of a symmetric search for 16 < K <= 32 neighbours.  The first pass (this kernel, K = 16) left every
// row's 16 nearest in lo_d / lo_i; this one finds the next K - 16: a pair (i, j) counts for row i only if (d2, caller row of j) lies
// lexicographically beyond the 16th entry of i's first list -- checked where the exact distance is known: column side in phase A,
// row side (the same test with the roles exchanged, against j's first list) in phase R.  Thresholds, slots and published bounds
// then speak of the (K - 16)-th entry beyond the cut, i.e. of the K-th neighbour; the prepass that seeds them bounds the K-th
// distance (capi_search.hpp).  The first 16 still pass the fp16 gate and are evaluated again -- the price of lists that hold 16.
template <int KST, int KCAP, bool LOWER = false>
__global__ __launch_bounds__(kHThreads, 2) void knn_panel_kernel(PanelArgs args_by_value)
{
    static_assert(kHQT == 2 && kHNL == 1, "8 waves x 2 query tiles, one list per owner lane");
    constexpr int QT = 2;
    constexpr int QPW = 64;                              // queries per wave
    constexpr int QPB = kHWaves * QPW;                   // 512
    constexpr int TPB = QPB / 32;
    constexpr int CT = panel_chunk_tiles(KST);
    constexpr int CHUNK_BYTES = CT * KST * 1024;
    constexpr int VPT = CHUNK_BYTES / 16 / kHThreads;
    static_assert(CHUNK_BYTES % (16 * kHThreads) == 0 && CT % 2 == 0, "chunk geometry");
    constexpr int QN = kPanelQueue;
    // The only explicit kernel argument is the struct, so it sits at offset 0 of the kernarg segment.  It is read through
    // this pointer, laundered before every cold use: the loads then happen THERE (scalar loads from constant memory)
    // instead of at kernel entry with the values held -- or spilled -- across the tile loop.
    (void)args_by_value;
    typedef const __attribute__((address_space(4))) PanelArgs* ArgsPtr;      // constant address space: uniform loads are scalar loads
    const ArgsPtr ap0 = (ArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
#define MCE_ARGS() ([&]() __attribute__((always_inline)) { ArgsPtr p_ = ap0; asm volatile("" : "+s"(p_)); return p_; }())

    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    char* const stage0 = lds_raw;
    double* const qd2_all = reinterpret_cast<double*>(lds_raw + 2 * CHUNK_BYTES);
    int* const qpk_all = reinterpret_cast<int*>(qd2_all + kHWaves * QN);
    int* const qnx_all = qpk_all + kHWaves * QN;
    int* const head_all = qnx_all + kHWaves * QN;
    volatile int* const wvote = head_all + kHWaves * QPW;                 // [3] drain votes (chunk index mod 3)
    float* const sthr_all = reinterpret_cast<float*>(head_all + kHWaves * QPW + 32);
    int* const redo_all = reinterpret_cast<int*>(sthr_all + kHWaves * QPW);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* const wqd = qd2_all + wave * QN;
    int* const wq = qpk_all + wave * QN;
    int* const wnx = qnx_all + wave * QN;
    int* const whead = head_all + wave * QPW;
    float* const sthr = sthr_all + wave * QPW;         // K-th bound of every wave-local query as of the last drain, rounded UP: what lies beyond never joins a chain
    int* const wredo = redo_all + wave * (CT * QT);
    const double INF = __builtin_huge_val();

#if MCE_PANEL_STATS
    const long long t_kernel0 = clock64();
    long long st_drains = 0, st_enq = 0, st_events = 0, st_redo = 0, st_tD = 0, st_tE = 0, st_tPro = 0, st_tA = 0, st_tR = 0, st_tB = 0;
#endif

    // ---- which unit ---------------------------------------------------------------------------------------------
    int sym_p, qblk, t_lo, t_hi, useq, sym_on, qb_lo;
    {
        const ArgsPtr a = MCE_ARGS();
        if (a->units) {
            // (a table made by panel_unit_table_kernel: strided blocks, several chains per block -- arithmetic this kernel is spared;
            //  with it inlined here the headline sweep ran 0.45 ms slower of 34.3 although the tile loop's code was the same)
            const auto u = (const __attribute__((address_space(1))) PanelUnit*)a->units + blockIdx.x;
            qblk = __builtin_amdgcn_readfirstlane(u->qblk);
            t_lo = __builtin_amdgcn_readfirstlane(u->t_lo);
            t_hi = __builtin_amdgcn_readfirstlane(u->t_hi);
            useq = __builtin_amdgcn_readfirstlane(u->useq);
            sym_p = 0;
            sym_on = a->geom.sym_on;
            qb_lo = a->geom.qb_lo;
        } else {
            PanelGeom g;
            g.qb_lo = a->geom.qb_lo; g.qb_hi = a->geom.qb_hi; g.tpb = a->geom.tpb; g.tpp = a->geom.tpp;
            g.ct = a->geom.ct; g.ntiles = a->geom.ntiles; g.sym_on = a->geom.sym_on;
            panel_unit_decode((int)blockIdx.x, g, sym_p, qblk);
            panel_unit_tiles(sym_p, qblk, g, t_lo, t_hi);
            useq = panel_unit_seq(sym_p, qblk, g);
            sym_on = g.sym_on;
            qb_lo = g.qb_lo;
        }
    }
    // the hand-over counter the unit waits on and bumps, and its list set: the block's, and set 0, unless the table says otherwise (read
    // where needed, here and at the very end, not carried through the tile loop)
    auto chain_of = [&](int& chain, int64_t& list_off) __attribute__((always_inline)) {
        const ArgsPtr a = MCE_ARGS();
        chain = qblk;
        list_off = 0;
        if (a->units) {
            const auto u = (const __attribute__((address_space(1))) PanelUnit*)a->units + blockIdx.x;
            chain = __builtin_amdgcn_readfirstlane(u->chain);
            list_off = (int64_t)__builtin_amdgcn_readfirstlane(u->list_set) * KCAP * a->nq_pad;
        }
    };
    const int64_t qwave0 = (int64_t)qblk * QPB + wave * QPW;     // first query of this wave

    whead[lane] = -1;
    if (tid < 3) wvote[tid] = 0;

    // lane l OWNS wave-local query l: its sorted top-KCAP list lives here
    double own_d[KCAP];
    int own_i[KCAP];
#pragma unroll
    for (int k = 0; k < KCAP; ++k) { own_d[k] = INF; own_i[k] = -1; }
    double seed_thr = INF;          // bound on the final K-th squared distance known from elsewhere (prepass, row side)

    // A block's lists travel from one of its units to the next through the list arrays.  Units are dispatched in number
    // order, panel by panel, so the previous unit of this block started a whole panel's worth of units ago and the wait
    // practically never spins.  It is BOUNDED all the same: in-order dispatch is an observation, not a guarantee (several
    // searches on different streams share the chip).  A wave that gives up starts from empty lists and flags its block:
    // the repair launch then searches that block again exhaustively -- the result never depends on the wait.  (The eight
    // waves of a unit wait independently; under load some give up and others do not -- found with
    // tools/stress_concurrent.py at a 50 us limit: whole waves of 64 queries with incomplete lists in a block only wave 0
    // would have flagged.)
    {
        const ArgsPtr a = MCE_ARGS();
        if (useq > 0) {
            int chain;
            int64_t list_off;
            chain_of(chain, list_off);
            int spins = 0;
            // (debug 16, tests: in every block's second unit the ODD waves give up at once -- the waves of a workgroup wait
            //  independently, so a give-up can be any subset of them)
            const int limit = ((a->debug & 16) && useq == 1 && (wave & 1)) ? -1 : a->spin_limit;
            bool ok = true;
            const auto done = gptr(a->sym.done);
            while (spins > limit || __hip_atomic_load(done + chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < useq) {
                if (++spins > limit) { ok = false; break; }
                __builtin_amdgcn_s_sleep(32);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            // A block in which some unit has given up is being searched by units that no longer run one after the other:
            // two of them may write the lists at the same time, and what a later unit would load can hold the same row
            // twice -- a K-th entry BELOW the true K-th distance, which would be published as a bound (and the repair launch
            // starts from the published bounds).  Such a block is flagged already (the flag is set before the sweep of the
            // unit that gave up, and this unit has seen all its predecessors finish): its units start from empty lists.
            if (ok && gptr(a->sym.bucket_flag)[qblk] != 0) ok = false;
            if (ok) {
                const int64_t q = qwave0 + lane;
                const int64_t np = a->nq_pad;
                const auto pd = gptr(a->part_d) + list_off;
                const auto pi = gptr(a->part_i) + list_off;
#pragma unroll
                for (int k = 0; k < KCAP; ++k) {
                    own_d[k] = pd[(int64_t)k * np + q];
                    own_i[k] = pi[(int64_t)k * np + q];
                }
            } else if (lane == 0) {
                // (every WAVE waits on its own and may be the only one of its workgroup to give up: each flags the block)
                gptr_w(a->sym.bucket_flag)[qblk] = 1;
            }
        }
    }

    // ---- B fragments (fp16 query rows) + per-query gate constants ---------------------------------------------------
    v8h b[QT][KST];
    float G[QT], cR[QT];
    unsigned lanew[QT];             // queue word of the lane's query: query-local << kHRelBits | 4 (lane >> 5)
    const int k_last = MCE_ARGS()->ksel - 1;
    {
        const ArgsPtr a = MCE_ARGS();
        const auto Xh = gptr(a->Xh);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int64_t q = qwave0 + qt * 32 + (lane & 31);
#pragma unroll
            for (int ks = 0; ks < KST; ++ks)
                b[qt][ks] = *(const __attribute__((address_space(1))) v8h*)(Xh + q * (int64_t)(16 * KST) + 16 * ks + 8 * (lane >> 5));
            lanew[qt] = ((unsigned)(qt * 32 + (lane & 31)) << kHRelBits) | (unsigned)(4 * (lane >> 5));
            cR[qt] = -__builtin_huge_valf();
            G[qt] = -__builtin_huge_valf();
        }
    }
    // Per-query constants of the gates, computed once and kept in registers (the drains re-read them from L2 before:
    // a round trip per refresh): for the lane's two gated queries gq_a = e_x + max e_y (+ slack) and gq_c = eps - |x^|^2
    // (-inf: padding query), for the query the lane OWNS own_a = the same sum; and the launch's scale^2 and the row gate's
    // additive term.  See knn_f16.hpp (gate_of, sym_row_gate) for the bound.
    double gq_a[QT], gq_c[QT], own_a = 0.0, s2c, rowc;
    {
        const ArgsPtr a = MCE_ARGS();
        const auto params = gptr(a->params);
        const auto qinfo = gptr(a->qinfo);
        const double p_scale = params[HP_SCALE], p_ey = params[HP_EY], p_ym = params[HP_YHATMAX], p_rho = params[HP_RHO];
        s2c = p_scale * p_scale;
        rowc = 0x1p-22 * (p_ym * p_ym + 1.0) + 1e-30;
        const double slack = 2.0 * sqrt(16.0 * KST) * 0x1p-14;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int64_t q = qwave0 + qt * 32 + (lane & 31);
            gq_a[qt] = 0.0;
            gq_c[qt] = -INF;
            if (q < a->nq && MCE_PANEL_ABL != 1) {
                const double ex = qinfo[2 * q], xn = qinfo[2 * q + 1];
                const double r = sqrt(xn) + p_ym;
                const double eps = (32.0 * KST) * 0x1p-24 * r * r * (1.0 + 0x1p-9) + p_rho + 1e-30;
                gq_a[qt] = (ex + p_ey) * (1.0 + 1e-9) + slack;
                gq_c[qt] = eps - xn;
            }
        }
        {
            const int64_t q = qwave0 + lane;
            if (q < a->nq) own_a = (qinfo[2 * q] + p_ey) * (1.0 + 1e-9) + slack;
        }
    }
    // gate of query (qt, lane & 31) from a bound `thr` on its K-th squared distance (input units)
    auto gate_of = [&](double thr, int qt) __attribute__((always_inline)) -> float {
        if (!(gq_c[qt] > -INF)) return -__builtin_huge_valf();
        if (!(thr < INF)) return __builtin_huge_valf();
        const double rr = sqrt(thr * s2c) * (1.0 + 1e-12) + gq_a[qt];
        return __double2float_ru(rr * rr * (1.0 + 1e-12) + gq_c[qt]);
    };
    // row-side gate constant R of the OWNED query from the bound on its K-th squared distance (sym_row_gate, knn_f16.hpp)
    auto own_row_gate = [&](double thr) __attribute__((always_inline)) -> float {
        if (!(thr < INF)) return __builtin_huge_valf();
        const double rr = sqrt(thr * s2c) * (1.0 + 1e-12) + own_a;
        return __double2float_ru(rr * rr * (1.0 + 1e-12) * (1.0 + 0x1p-22) + rowc);
    };
    {
        const ArgsPtr a = MCE_ARGS();
        const int64_t q = qwave0 + lane;
        double t0 = INF;
        if (sym_on) t0 = __longlong_as_double((long long)__hip_atomic_load(gptr(a->sym.thr) + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        else if (a->sym.thr) t0 = __longlong_as_double((long long)gptr(a->sym.thr)[q]);       // bounds of a prepass (plain array)
        seed_thr = t0;
        // (the list may already hold a tighter K-th: units after the first)
        double tl = own_d[KCAP - 1];
#pragma unroll
        for (int k = 0; k < KCAP - 1; ++k) tl = (k == k_last) ? own_d[k] : tl;
        t0 = fmin(t0, tl);
        sthr[lane] = __double2float_ru(t0);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) G[qt] = gate_of(__shfl(t0, qt * 32 + (lane & 31), 64), qt);
        if (sym_on) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) cR[qt] = gq_c[qt] > -INF ? __double2float_ru(gq_c[qt]) : -__builtin_huge_valf();
        }
    }

    const int dbg_flags = MCE_ARGS()->debug;
    // a hit is queued while at most this many entries are waiting (64 lanes may follow); beyond, the tile is deferred to the
    // redo list.  (debug 8, tests: -1 -- every candidate of the sweep goes through the redo list)
    const int qlimit_gate = (dbg_flags & 8) ? -1 : QN - 64;
    int qcount = 0;      // wave-uniform number of queued candidates
    int nredo = 0;       // wave-uniform number of deferred (tile, query tile) events of the current chunk

    // ---- the drain: exact distances (phase A), row side (phase R), list insertion (phase B), new gates ---------------
    auto drain = [&]() __attribute__((always_inline)) {
#if MCE_PANEL_STATS
        const long long t_d0 = clock64();
        st_drains += 1; st_enq += qcount;
#endif
        const ArgsPtr a = MCE_ARGS();
        const auto X = gptr(a->X);
        const auto Y = gptr(a->Y);
        const int D = a->D;
        const int64_t nq = a->nq, nr = a->nr;
        const int self_exclude = a->self_exclude;
        const int64_t self_offset = a->self_offset;
        const auto rperm = gptr(a->rperm);
        {
            // phase A: 8 lanes share one queued pair and read the two rows in 64-byte segments; all loads of a group of
            // NPASS * 8 pairs are issued before the first use (the gather is latency-bound)
            const int sub = lane & 7;
            constexpr int NPASS = KST > 2 ? 1 : MCE_PANEL_NPASS;
            constexpr int EPL = KST > 2 ? 8 : 4;        // elements per lane: 4 covers D <= 32, 8 covers D <= 63
            for (int b0 = 0; b0 < qcount; b0 += NPASS * 8) {
                int qlp[NPASS], ep[NPASS];
                bool okp[NPASS];
                const __attribute__((address_space(1))) double* xp[NPASS];
                const __attribute__((address_space(1))) double* yp[NPASS];
#pragma unroll
                for (int u = 0; u < NPASS; ++u) {
                    const int e = b0 + u * 8 + (lane >> 3);
                    ep[u] = e;
                    int ql = 0, j = 0;
                    const bool valid = e < qcount;
                    if (valid) {
                        const unsigned ent = (unsigned)wq[e];
                        ql = (int)(ent >> kHRelBits);
                        j = (int)(ent & ((1u << kHSymRowBits) - 1u));
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
                        const int iv = (sub + 8 * v < D) ? sub + 8 * v : (sub < D ? sub : D - 1);     // clamped INSIDE the row, masked use
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
                    // every entry gets its distance (-1: no pair behind it) for phase R; only what can still enter the
                    // query's list (K-th bound of the last drain) joins its chain
                    if (sub == 0 && ep[u] < qcount) {
                        wqd[ep[u]] = okp[u] ? a0 : -1.0;
                        bool col = okp[u] && !(a0 > (double)sthr[qlp[u]]);
                        if constexpr (LOWER) {
                            if (col) {       // beyond the 16th entry of the query's first list?  (list not full: +inf, nothing is left)
                                const int64_t o = (int64_t)(KCAP - 1) * a->nq_pad + qwave0 + qlp[u];
                                const double ld = gptr(a->lo_d)[o];
                                const int li = gptr(a->lo_i)[o];
                                const int jr = (int)((unsigned)wq[ep[u]] & ((1u << kHSymRowBits) - 1u));
                                const int jc = rperm ? rperm[jr] : jr;
                                col = a0 > ld || (a0 == ld && jc > li);
                            }
                        }
                        if (col) wnx[ep[u]] = atomicExch(&whead[qlp[u]], ep[u]);
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#if MCE_PANEL_STATS
        const long long t_dA = clock64();
        st_tA += t_dA - t_d0;
#endif
        {
            // phase R (one lane per queue entry): the ROW side of every evaluated pair whose lane passed the row gate --
            // through row j's K slots (replace the largest of the K smallest row-side distances so far, by
            // compare-and-swap; their maximum is a bound on j's K-th distance, published for everybody) and into the
            // bucket of j's block.  Then the entry's packed word is replaced by j's caller row, which the lists carry.
            const auto sp_thr = gptr_w(a->sym.thr);
            const auto sp_rrow = gptr_w(a->sym.rrow);
            const auto sp_rtile = gptr_w(a->sym.rtile);
            const auto sp_slots = gptr_w(a->sym.slots);
            const auto sp_bucket_cnt = gptr_w(a->sym.bucket_cnt);
            const auto sp_bucket_flag = gptr_w(a->sym.bucket_flag);
            const auto sp_bucket = gptr_w(a->sym.bucket);
            const int sp_cap = a->sym.cap;
            const int ksel = a->ksel;
            auto slot_insert = [&](int row, double d2) __attribute__((always_inline)) -> bool {
                const auto sl = sp_slots + (int64_t)row * KCAP;
                for (;;) {
                    double vmax = -1.0, v2 = -1.0;
                    int imax = 0;
#pragma unroll
                    for (int k = 0; k < KCAP; ++k) {
                        if (k < ksel) {
                            const double v = __longlong_as_double((long long)__hip_atomic_load(sl + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                            if (v > vmax) { v2 = vmax; vmax = v; imax = k; }
                            else if (v > v2) v2 = v;
                        }
                    }
                    if (d2 > vmax) return false;
                    if (d2 == vmax) return true;                     // a tie: the merge decides by row number
                    unsigned long long expect = (unsigned long long)__double_as_longlong(vmax);
                    if (__hip_atomic_compare_exchange_strong(sl + imax, &expect, (unsigned long long)__double_as_longlong(d2), __ATOMIC_RELAXED,
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        const double nk = fmax(v2, d2);              // the K-th smallest after the replacement, from a snapshot: an upper bound
                        if (nk < INF) {
                            const unsigned long long nb = (unsigned long long)__double_as_longlong(nk);
                            const unsigned long long ob = __hip_atomic_fetch_min(sp_thr + row, nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (nb < ob) {
                                const unsigned rb = __float_as_uint(sym_row_gate(nk, gptr(a->qinfo)[2 * (int64_t)row], a->params, KST));
                                const unsigned orb = __hip_atomic_fetch_min(sp_rrow + row, rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (rb < orb) {
                                    // the tile's largest R_j, from a snapshot (each value >= its current one): safe to store
                                    const auto rt = sp_rrow + (int64_t)(row >> 5) * 32;
                                    unsigned m = 0;
                                    for (int k = 0; k < 32; ++k) {
                                        const unsigned v = __hip_atomic_load(rt + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        m = v > m ? v : m;
                                    }
                                    __hip_atomic_store(sp_rtile + (row >> 5), __uint_as_float(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                            }
                        }
                        return true;
                    }
                }
            };
            for (int e0 = 0; e0 < qcount; e0 += 64) {
                const int e = e0 + lane;
                const bool valid = e < qcount;
                const unsigned ent = valid ? (unsigned)wq[e] : 0u;
                const int ql = (int)(ent >> kHRelBits);
                const int j = (int)(ent & ((1u << kHSymRowBits) - 1u));
                const bool rowflag = (ent >> kHSymRowBits) & 1u;        // the lane passed the row gate on this tile: only then can the pair matter to row j
                const double d2 = valid ? wqd[e] : -1.0;
                const bool ok = valid && d2 >= 0.0;
                const int oj = ok ? (rperm ? rperm[j] : j) : -1;
                bool rs = ok && rowflag;          // (the row flag is raised only on tiles that carry the row-side gate)
                if (rs) {
                    const int jb = j / QPB;
                    rs = d2 <= __longlong_as_double((long long)__hip_atomic_load(sp_thr + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if constexpr (LOWER) {
                        if (rs) {       // beyond the 16th entry of row j's first list?  (what j is offered is the QUERY's row)
                            const int64_t o = (int64_t)(KCAP - 1) * a->nq_pad + j;
                            const double ld = gptr(a->lo_d)[o];
                            const int li = gptr(a->lo_i)[o];
                            const int sc = rperm ? rperm[qwave0 + ql] : (int)(qwave0 + ql);
                            rs = d2 > ld || (d2 == ld && sc > li);
                        }
                    }
                    if (rs) rs = slot_insert(j, d2);
                    if (rs) {
                        const int slot = __hip_atomic_fetch_add(sp_bucket_cnt + jb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned)slot < (unsigned)sp_cap) {      // (unsigned: a count that is not a count ends in the repair pass, not in a wild store)
                            const auto en = sp_bucket + ((int64_t)jb * sp_cap + slot);
                            en->d2 = d2;
                            en->src = rperm ? rperm[qwave0 + ql] : (int)(qwave0 + ql);
                            en->row = j;
                        } else {
                            sp_bucket_flag[jb] = 1;
                        }
                    }
                }
                if (valid) wq[e] = oj;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#if MCE_PANEL_STATS
        const long long t_dR = clock64();
        st_tR += t_dR - t_dA;
#endif
        // ---- phase B: every owner lane folds its chain into its register list ----------
        {
            int cur = whead[lane];
            whead[lane] = -1;
            while (__any(cur >= 0)) {
                const bool on = cur >= 0;
                const int ce = on ? cur : 0;
                const double d2 = on ? wqd[ce] : INF;
                const int j = wq[ce];
                cur = on ? wnx[ce] : -1;
                // ascending list, ties by row; d2 = +inf (idle lane) changes nothing
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
        }
        qcount = 0;
        // ---- refresh the gates; symmetric sweep: publish the bound and take back what the row side knows ----------
        double thr_own = own_d[KCAP - 1];
#pragma unroll
        for (int k = 0; k < KCAP - 1; ++k) thr_own = (k == k_last) ? own_d[k] : thr_own;
        thr_own = fmin(thr_own, seed_thr);
        if (sym_on) {
            const auto sp_thr = gptr_w(a->sym.thr);
            const auto sp_rrow = gptr_w(a->sym.rrow);
            const auto sp_rtile = gptr_w(a->sym.rtile);
            const int64_t q = qwave0 + lane;
            double t = thr_own;
            float R = 0.0f;
            if (q < nq) {
                if (t < INF) {
                    // both fetch-mins in flight together: the row constant is formed from the list's own bound; whoever
                    // published a tighter bound on this row published the matching constant with it, and it comes back here
                    const unsigned long long mb = (unsigned long long)__double_as_longlong(t);
                    const unsigned rb = __float_as_uint(own_row_gate(t));
                    const unsigned long long ob = __hip_atomic_fetch_min(sp_thr + q, mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned orb = __hip_atomic_fetch_min(sp_rrow + q, rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    t = fmin(t, __longlong_as_double((long long)ob));
                    R = __uint_as_float(rb < orb ? rb : orb);
                } else {
                    t = __longlong_as_double((long long)__hip_atomic_load(sp_thr + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    const unsigned rb = __float_as_uint(own_row_gate(t));
                    const unsigned orb = __hip_atomic_fetch_min(sp_rrow + q, rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    R = __uint_as_float(rb < orb ? rb : orb);
                }
            }
            thr_own = t;
            seed_thr = t;
            float m = R;
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            if ((lane & 31) == 0) __hip_atomic_store(sp_rtile + ((qwave0 + lane) >> 5), m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        sthr[lane] = __double2float_ru(thr_own);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) G[qt] = gate_of(__shfl(thr_own, qt * 32 + (lane & 31), 64), qt);
#if MCE_PANEL_STATS
        st_tD += clock64() - t_d0;
#endif
    };

    // ---- a tile with a candidate ------------------------------------------------------------------------------------
    // c: the 16 accumulators of query tile qt (C layout of 32x32 f32: lane l -> query column l & 31, rows
    // (r & 3) + 8 (r >> 2) + 4 (l >> 5)); gq: the lane's gate; rowflag: the lane passed the ROW gate; jb0: first reference
    // row of the tile; todo: the accumulators still to be looked at (a redo passes what is left).  Wave-wide compares, scalar
    // branches over the empty ones; the lanes under the gate append (query, row) to the wave's queue.  Returns the accumulators
    // NOT handled because the queue was full.  (Static s_setprio for either half of the workgroup -- MI355X_MICROARCH.md, two
    // waves per SIMD, item 4 -- was measured in round 5: 36.9-37.1 ms either way against 36.7-37.3, nothing.  DYNAMIC priority, round 6 --
    // raised while a wave sweeps and dropped inside its candidate tiles and drains, or the other way round: +1.0 % / +0.15 %,
    // profiles/r06_mid/panel_dynamic_priority_ab.txt.)
    auto event = [&](const v16f& c, const float (&l1)[5], const int qt, const float gq, const bool rowflag, const int jb0, const unsigned todo, const int qlimit) __attribute__((always_inline)) -> unsigned {
        (void)l1;
        const unsigned wbase = (lanew[qt] + (unsigned)jb0) | (rowflag ? (1u << kHSymRowBits) : 0u);
        unsigned rem = 0;
        float g = gq;
#define MCE_HIT(R_, P_, S_)                                                                                               \
        if ((S_) != 0 && (todo & (1u << (R_)))) {                                                                         \
            if (qcount > qlimit) rem |= 1u << (R_);                                                                       \
            else {                                                                                                        \
                if (P_) wq[__builtin_amdgcn_mbcnt_hi((unsigned)((S_) >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)(S_), (unsigned)qcount))] = \
                            (int)(wbase + (unsigned)(((R_) & 3) + 8 * ((R_) >> 2)));                                      \
                qcount += __builtin_popcountll(S_);                                                                       \
            }                                                                                                             \
        }
        // the gate's own first-level minima say which triples of accumulators hold something: 6 wave-wide compares, then 3
        // for each triple that does (usually one) -- 9 instead of 16
        {
            const bool q0 = l1[0] <= g, q1 = l1[1] <= g, q2 = l1[2] <= g, q3 = l1[3] <= g, q4 = l1[4] <= g, p15 = c[15] <= g;
            const unsigned long long u0 = __ballot(q0), u1 = __ballot(q1), u2 = __ballot(q2), u3 = __ballot(q3), u4 = __ballot(q4), s15 = __ballot(p15);
#define MCE_TRIPLE(I_, U_)                                                                                                \
            if ((U_) != 0) {                                                                                              \
                const bool p0 = c[3 * (I_) + 0] <= g, p1 = c[3 * (I_) + 1] <= g, p2 = c[3 * (I_) + 2] <= g;             \
                const unsigned long long s0 = __ballot(p0), s1 = __ballot(p1), s2 = __ballot(p2);                        \
                MCE_HIT(3 * (I_) + 0, p0, s0)                                                                             \
                MCE_HIT(3 * (I_) + 1, p1, s1)                                                                             \
                MCE_HIT(3 * (I_) + 2, p2, s2)                                                                             \
            }
            MCE_TRIPLE(0, u0) MCE_TRIPLE(1, u1) MCE_TRIPLE(2, u2) MCE_TRIPLE(3, u3) MCE_TRIPLE(4, u4)
#undef MCE_TRIPLE
            MCE_HIT(15, p15, s15)
        }
#undef MCE_HIT
        return rem;
    };

    // ---- staging (global_load_lds DMA, linear image) + A fragments + MFMA -----------------------------------------
    const auto Yh_bytes = (const __attribute__((address_space(1))) char*)MCE_ARGS()->Yh;
    auto stage_async = [&](int64_t c, int buf) {
        const auto src = Yh_bytes + c * (int64_t)CHUNK_BYTES;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int e = tid + i * kHThreads;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + (size_t)e * 16),
                (__attribute__((address_space(3))) void*)(stage0 + buf * CHUNK_BYTES + (size_t)(wave * 64 + i * kHThreads) * 16),
                16, 0, 0);
        }
    };
    auto load_a = [&](const char* lp, v8h (&a)[KST]) {
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) a[ks] = *reinterpret_cast<const v8h*>(lp + ks * 1024);
    };
    // one 32-row tile = QT chains of KST MFMAs, issued k-step by k-step (the dependent ones one apart)
    auto mfma_first = [&](const v8h (&a)[KST], v16f (&acc)[QT]) __attribute__((always_inline)) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            v16f z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[qt][0], z, 0, 0, 0);
        }
    };
    auto mfma_rest = [&](const v8h (&a)[KST], v16f (&acc)[QT]) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 1; ks < KST; ++ks)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) acc[qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], b[qt][ks], acc[qt], 0, 0, 0);
    };
    // The gate reads the accumulators through inline asm (v_min3_f32), which neither the compiler's hazard recogniser nor
    // its scheduler knows to keep away from the MFMAs that write them.  The result of an 8-pass MFMA may be read 11 wait
    // states after issue; two later MFMAs in the (in-order) matrix pipe are 16.  This empty statement pins the order: what
    // reads GATED_ comes after it, and it comes after the first two MFMAs of the next tile (NEXT_); the rest of that
    // tile's MFMAs and the gate's VALU work are then free to interleave.
#if defined(__HIP_DEVICE_COMPILE__)
#define MCE_ORDER(GATED_, NEXT_) asm volatile("" : "+v"(GATED_[0]), "+v"(GATED_[1]), "+v"(NEXT_[0]), "+v"(NEXT_[1]))
#else
#define MCE_ORDER(GATED_, NEXT_) do {} while (0)
#endif
    // min of the lane's 16 accumulators (8 v_min3_f32); l1: the five first-level minima -- of the accumulators 3i .. 3i + 2 --
    // which the event path looks at first
    auto min16 = [&](const v16f& c, float (&l1)[5]) __attribute__((always_inline)) -> float {
        l1[0] = min3f(c[0], c[1], c[2]);
        l1[1] = min3f(c[3], c[4], c[5]);
        l1[2] = min3f(c[6], c[7], c[8]);
        l1[3] = min3f(c[9], c[10], c[11]);
        l1[4] = min3f(c[12], c[13], c[14]);
        const float m0 = min3f(l1[0], l1[1], l1[2]);
        const float m3 = min3f(l1[3], l1[4], c[15]);
        return min3f(m0, m3, m3);
    };
    // gate of one finished tile (both query tiles): ONE branch per tile, so that the tile's MFMAs and the gate's VALU work
    // share a basic block and interleave; tix: the tile's index in its chunk (for a redo); gq / rg: the lane's gates for
    // the PAIR of tiles this one belongs to (either side / row side)
    auto gate_tile = [&](const v16f (&acc)[QT], const int jb0, const float (&gq)[QT], const float (&rg)[QT], const int tix) __attribute__((always_inline)) {
#if MCE_PANEL_ABL == 2      // tools only: no gate at all -- the MFMA + LDS stream alone (results invalid)
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" :: "v"(acc[0]), "v"(acc[1]));
#endif
        return;
#endif
        float l1[QT][5], mm[QT];
        bool pq[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            mm[qt] = min16(acc[qt], l1[qt]);
            pq[qt] = mm[qt] <= gq[qt];
        }
        if (__any(pq[0] || pq[1])) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                if (!__any(pq[qt])) continue;
#if MCE_PANEL_STATS
                st_events += 1;
                const long long t_e0 = clock64();
#endif
                const unsigned rem = event(acc[qt], l1[qt], qt, gq[qt], mm[qt] <= rg[qt], jb0, 0xffffu, qlimit_gate);
                if (rem) {                                     // queue full: the chunk end multiplies this tile again
                    if (lane == 0) wredo[nredo] = (int)(rem | ((unsigned)qt << 16) | ((unsigned)tix << 17));
                    nredo += 1;
                }
#if MCE_PANEL_STATS
                st_tE += clock64() - t_e0;
#endif
            }
        }
    };

    // ---- the unit's chunks ------------------------------------------------------------------------------------------
    const int cfirst = t_lo / CT;
    const int ntot = (t_hi - 1) / CT - cfirst + 1;
    const auto rtile_p = gptr(MCE_ARGS()->sym.rtile);
    // lane t <- the row-side gate constant of tile t of chunk c: only the tiles of the launch's OTHER blocks below this one
    auto rt_load = [&](int c) -> float {
        const int t = c * CT + lane;
        const int tb = t / TPB;
        const bool en = sym_on && lane < CT && tb >= qb_lo && tb < qblk && MCE_PANEL_ABL != 1;
        const float r = en ? __hip_atomic_load(rtile_p + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -__builtin_huge_valf();
        // the tiles are gated in pairs (t, t + 1), t even, with the larger of the two constants: one gate refresh per pair
        // (rows sorted by distance from the mean: neighbouring tiles differ little); a block is a whole number of pairs
        return fmaxf(r, __shfl_xor(r, 1, 64));
    };
    float rt_cur = -__builtin_huge_valf(), rt_next;
    stage_async(cfirst, 0);
    rt_next = rt_load(cfirst);
#if MCE_PANEL_STATS
    st_tPro = clock64() - t_kernel0;
#endif
    v16f accA[QT], accB[QT];
    for (int k = 0; k < ntot; ++k) {
        const int buf = k & 1;
        const int c = cfirst + k;
        // drain votes: slot k % 3 collects the votes for chunk k (written before barrier k, read by everybody after it) and is
        // cleared by thread 0 after barrier k + 1 -- which every wave reaches only after its read -- and voted on again for
        // chunk k + 3, by waves that have passed barrier k + 2, i.e. after the clear: no write can overtake another.  (Two
        // slots re-armed right after the read let a late thread 0 wipe the vote of a wave already one chunk ahead.)
        const int vs = k % 3;
        if (qcount >= MCE_PANEL_TRIGGER && lane == 0) wvote[vs] = 1;
#if MCE_PANEL_STATS
        const long long t_b0 = clock64();
#endif
        dma_barrier();
#if MCE_PANEL_STATS
        st_tB += clock64() - t_b0;
#endif
        if (tid == 0) wvote[vs == 0 ? 2 : vs - 1] = 0;             // the slot of chunk k - 1: all its readers are behind this barrier
        rt_cur = rt_next;
        if (k + 1 < ntot) {
            stage_async(c + 1, buf ^ 1);
            rt_next = rt_load(c + 1);
        }
        // (readfirstlane: an LDS load is a divergent value to the compiler, and a drain under a "divergent" branch would make
        //  the queue length a per-lane quantity -- vector compares and exec masks on every use)
        bool need_drain = __builtin_amdgcn_readfirstlane(wvote[vs]) != 0 || k + 1 == ntot;         // everybody drains at the same chunk; the last chunk ends with the final drain
        const int tlo = (k == 0) ? t_lo - c * CT : 0;
        const int thi = t_hi - c * CT < CT ? t_hi - c * CT : CT;
        const char* const lbuf = stage0 + buf * CHUNK_BYTES + lane * 16;
        const int jchunk = c * (CT * 32);
        {
            // tiles in pairs, software-pipelined one tile deep: the MFMAs of a tile are issued before the gate of the one
            // before it; the gate of the pair's second tile closes the chunk (flush)
            v8h a0[KST], a1[KST];
            load_a(lbuf + (tlo * KST) * 1024, a0);
            load_a(lbuf + ((tlo + 1) * KST) * 1024, a1);
            mfma_first(a0, accA);
            mfma_rest(a0, accA);
            int t = tlo;
            float gqP[QT], rgP[QT];
            _Pragma("unroll 1") for (;;)
            {
                const float rP = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(rt_cur), t));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    rgP[qt] = rP + cR[qt];
                    gqP[qt] = vmaxf(G[qt], rgP[qt]);            // either side
                }
                const bool more = t + 2 < thi;
                load_a(lbuf + ((more ? t + 2 : t) * KST) * 1024, a0);        // (last trip: harmless re-read)
                mfma_first(a1, accB);
                MCE_ORDER(accA, accB);
                mfma_rest(a1, accB);
                gate_tile(accA, jchunk + t * 32, gqP, rgP, t);
                if (!more) break;
                load_a(lbuf + ((t + 3) * KST) * 1024, a1);
                mfma_first(a0, accA);
                MCE_ORDER(accB, accA);
                mfma_rest(a0, accA);
                gate_tile(accB, jchunk + (t + 1) * 32, gqP, rgP, t + 1);
                t += 2;
            }
            {   // the chunk's last tile: no MFMA behind it -- the wait states spelled out, then its gate (flush)
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("s_nop 15\n\ts_nop 3" : "+v"(accB[0]), "+v"(accB[1]));
#endif
                gate_tile(accB, jchunk + (t + 1) * 32, gqP, rgP, t + 1);
            }
        }
        // ---- chunk end: the deferred tiles and the drain (ONE copy of it in the code) ------------------------------
        // (the unit's last chunk must leave the queue empty: what a redo queues there is drained too)
        while (need_drain || nredo > 0 || (k + 1 == ntot && qcount > 0)) {
            drain();
            need_drain = false;
            const int n = nredo;
            nredo = 0;
            for (int i = 0; i < n; ++i) {
                const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane(wredo[i]);
                const unsigned todo = w & 0xffffu;
                const int qt = (int)((w >> 16) & 1u), t = (int)(w >> 17);
                const float Rt = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(rt_cur), t));
                v8h at[KST];
                load_a(lbuf + (t * KST) * 1024, at);
                v16f z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                v16f r_ = z;
                // (wave-uniform choice of the query tile: a branch, not sixteen selects)
                if (qt == 0) {
#pragma unroll
                    for (int ks = 0; ks < KST; ++ks) r_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(at[ks], b[0][ks], r_, 0, 0, 0);
                } else {
#pragma unroll
                    for (int ks = 0; ks < KST; ++ks) r_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(at[ks], b[1][ks], r_, 0, 0, 0);
                }
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("s_nop 15\n\ts_nop 3" : "+v"(r_));
#endif
                float l1[5];
                const float mm = min16(r_, l1);
                const float rg = Rt + (qt ? cR[1] : cR[0]);
                const float gq = vmaxf(qt ? G[1] : G[0], rg);
                unsigned rem = 0;
                if (__any(mm <= gq)) rem = event(r_, l1, qt, gq, mm <= rg, jchunk + t * 32, todo, QN - 64);
                if (rem) {
                    if (lane == 0) wredo[nredo] = (int)(rem | ((unsigned)qt << 16) | ((unsigned)t << 17));
                    nredo += 1;
                }
#if MCE_PANEL_STATS
                st_redo += 1;
#endif
            }
        }
    }

#if MCE_PANEL_STATS
    if (lane == 0) {
        const ArgsPtr a = MCE_ARGS();
        double* o = const_cast<double*>(a->params) + 16 + ((int64_t)blockIdx.x * kHWaves + wave) * 8;
        o[0] = (double)st_drains; o[1] = (double)st_enq; o[2] = (double)st_redo; o[3] = (double)st_events;
        o[4] = (double)st_tE; o[5] = (double)st_tD; o[6] = (double)(clock64() - t_kernel0); o[7] = (double)st_tPro;
        double* o2 = const_cast<double*>(a->params) + 16 + ((int64_t)gridDim.x * kHWaves) * 8 + ((int64_t)blockIdx.x * kHWaves + wave) * 8;
        o2[0] = (double)st_tA; o2[1] = (double)st_tR; o2[2] = (double)(st_tD - st_tA - st_tR); o2[3] = (double)st_tB; o2[4] = 0; o2[5] = 0; o2[6] = 0; o2[7] = 0;
    }
#endif
    // ---- write the lists (lane l owns wave-local query l: coalesced) and hand them to the block's next unit ------------
    {
        const ArgsPtr a = MCE_ARGS();
        const int64_t q = qwave0 + lane;
        const int64_t np = a->nq_pad;
        int chain;
        int64_t list_off;
        chain_of(chain, list_off);
        const auto pd = gptr_w(a->part_d) + list_off;
        const auto pi = gptr_w(a->part_i) + list_off;
#pragma unroll
        for (int k = 0; k < KCAP; ++k) {
            pd[(int64_t)k * np + q] = own_d[k];
            pi[(int64_t)k * np + q] = own_i[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(gptr_w(a->sym.done) + chain, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#undef MCE_ARGS
}

}  // namespace mce
