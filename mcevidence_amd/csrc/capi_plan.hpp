// capi_plan.hpp -- part of capi.hip: the planner.  Plan = which kernel, which geometry, which scratch; make_plan() is a pure
// function of the sizes, the modes in force on this thread and the planner hints (no device work).
#pragma once
namespace {

using mce::kMaxDevices;
constexpr int kAssumedCUs = 256;   // MI355X; only steers the reference-split heuristic

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Pruned walk over ONE set (auto evidence), heavy waves: the walk's workgroups (one wave of 64 queries each) are dispatched
// largest box first, and the first of them -- sparse cells of the k-d order, waves holding a far outlier -- walk their chunk
// list 10-30x longer than the average (C5, measured with MCE_PRUNE_TIMES: mean 2.4 ms, 1 in 1000 over 18 ms, one at 81 ms
// of a 200 ms launch).  A launch of many rounds of workgroups hides that (heaviest first); a rank of an 8-GPU run (25 ms of
// walks) or a one-GPU search of 1 M rows does not.  In a launch of fewer than kPruneHeavyMaxRounds rounds (2048 waves in
// flight) the first waves of the order (kPruneHeavyCount of a launch of up to kPruneHeavyFullRounds rounds, fewer of a
// longer one: run_search) are therefore served by kPruneHeavySplit workgroups each, every one on every S-th window of the
// list (knn_f16.hpp; MCE_PRUNE_HEAVY="<waves>,<S>" overrides, "0" = off; a split wave costs ~1.6x the work of a whole one:
// bootstrap per sub-wave, looser bounds).
// The side lists are sized for 1 / kPruneHeavyMaxShare of all waves at kPruneHeavyMaxSplit.
// Since the per-query reach test (knn_f16.hpp: query_reach -- a far outlier no longer sets the reach of its whole wave) the
// slowest wave of C5 takes 3.5 ms against a mean of 1.25 (was 81 against 2.4), and splitting only costs: one rank of eight
// 8.8 ms of walk without, 9.7 with (4 M rows with Student-t tails, nu = 3: 24.1 vs 25.2).  OFF by default now
// (kPruneHeavyDefault); MCE_PRUNE_HEAVY="<waves>,<S>" still sets it, "auto" restores the rule below.
constexpr bool kPruneHeavyDefault = false;
constexpr int kPruneHeavyMinBlocks = 64, kPruneHeavyCount = 700, kPruneHeavyMaxShare = 16, kPruneHeavyMinCount = 32, kPruneHeavyMaxRounds = 60;
constexpr double kPruneHeavyFullRounds = 10.0;
constexpr int kPruneHeavySplit = 4, kPruneHeavyMaxSplit = 8;
constexpr int kPruneWaveQueries = mce::kHQT * 32;     // list columns per workgroup of the walk
constexpr int kWideMinBlocks = 480;                    // query blocks from which the exhaustive one-k-step sweep takes its wide form (make_plan):
                                                       // ~one full round of the wider workgroups (two ranks' and four ranks' shards of C4: 977, 489 blocks)
// side lists for split waves are part of a plan only when the split can happen (44 % of the list arrays: ~630 MB at C5)
bool prune_heavy_enabled()
{
    if (t_no_heavy) return false;
    const char* e = getenv("MCE_PRUNE_HEAVY");
    return kPruneHeavyDefault || (e && *e && strcmp(e, "0") != 0);
}
static_assert(mce::kPruneWavesPerBlock == mce::kHWaves, "prune.hip orders kHWaves waves per query block");

constexpr int kRefineMargin = 2;
struct Plan {
    const mce::KnnVariant* v = nullptr;
    const mce::KnnF16Variant* vh = nullptr;   // non-null: fp16-filter path
    const mce::KnnDeepVariant* vd = nullptr;  // non-null: the DEEP fp16 filter (knn_deep.hpp: 64 <= d <= 127, K <= 32); vh and v are null then
    bool filter() const { return vh != nullptr || vd != nullptr; }      // exact lists behind an fp16 filter (no refine in the merge; certified by default)
    const mce::KnnLongVariant* vl = nullptr;  // non-null: the long-row fp64 sweep (knn_long.hpp: 128 <= d <= 1024, K <= 32); KS = its padded k-steps
    size_t off_xf = 0, off_xn = 0;            // ... its packed queries and their squared norms
    bool generic = false;                     // plain exact kernel (K > 32, or d > 127 with the long-row sweep turned off)
    int KST = 0;
    size_t off_yh = 0, off_xh = 0, off_qinfo = 0, off_params = 0;
    int KS = 0, KCAP = 0, QT = 0, CT = 0;
    int64_t nchunk = 0;      // reference chunks (CT tiles of 16 rows)
    int64_t nrow_pad = 0;    // padded reference rows
    int nqblk = 0;
    int64_t nq_pad = 0;
    int rsplit = 1;
    int L = 4;
    size_t off_yf = 0, off_pd = 0, off_pi = 0, off_center = 0, off_msum = 0, total = 0;
    double cost = 0.0;                        // the split model's estimate for this plan (cycles per SIMD; exhaustive kernels)
    int ksel = 0;                             // fp64 sweep: entries kept per list -- K + kRefineMargin (capped at MCE_MAX_K): the lists are chosen on
                                              // GEMM-form keys, the final K among them on EXACT distances (reduce_kernels.hpp, REFINE); 0: K
    bool twopass = false;                     // fp16 filter, 16 < K <= 32: two sweeps of 16-entry lists (knn_f16.hpp, LOWER)
    bool wide_ok = false;                     // the exhaustive sweep may run four query tiles per wave (knn_f16.hpp, QTT = 4): nqblk is even
    bool prune = false;                       // fp16 filter walking k-d ordered chunk lists (prune.hpp)
    bool kd_ready = false;                    // ... whose k-d order is in the workspace already (mce_prune_part_prepare_dev on every rank + the all-reduce
                                              // of the permutation: mce_knn_dotp_part_prepared_f64_dev)
    int part = 0, nparts = 1;                 // pruned walk over the waves part, part + nparts, ... of the dispatch order only; symmetric sweep: the
                                              // contiguous range of sorted blocks [sym_qb_lo, sym_qb_hi) (one rank's share)
    int sym_qb_lo = 0, sym_qb_hi = 0;         // set by run_search when the symmetric sweep ran
    bool apo = false;                         // symmetric sweep as one rank's share of the all-pairs-once partition (capi_apo.hpp): run_search stops after the sweep
    int apo_phase = 0;                        // ... 1: up to the prepass of the rank's own blocks; 2: the sweep (capi_search.hpp)
    int apo_nsplit = 1;                       // ... with this many independent chains (list sets) per block, and
    int apo_panel = 0;                        // ... panels of this many chunks (0: the default length)
    int64_t pl_nr = 0;                        // reference rows the plan was made for
    mce::PruneLayout pl;
    size_t off_prune = 0;
    size_t off_heavy = 0;                     // pruned walk over one set: side lists of the heavy waves' extra sub-waves (knn_f16.hpp)
    int heavy_max = 0;                        // ... room for this many waves at kPruneHeavyMaxSplit sub-waves
    bool sym = false;                         // workspace holds the symmetric sweep's scratch (run_search decides: X and Y must be one buffer)
    bool sym_active = false;                  // set by run_search: the lists are in sorted-row order, one split
    mce::SymLayout sl;
    size_t off_sym = 0;
    int sym_nsplit = 1;                       // symmetric sweep on one GPU: chains (list sets) per block, and
    int sym_panel = 0;                        // ... chunks per panel when that shortens them (0: the default length)
};

const mce::KnnVariant* variant_for(int KS, int kcap_idx)
{
    switch (kcap_idx) {
        case 0: return &mce::g_knn_kcap4[mce::mfma_variant_index(KS)];
        case 1: return &mce::g_knn_kcap8[mce::mfma_variant_index(KS)];
        case 2: return &mce::g_knn_kcap12[mce::mfma_variant_index(KS)];
        case 3: return &mce::g_knn_kcap16[mce::mfma_variant_index(KS)];
        case 4: return &mce::g_knn_kcap24[mce::mfma_variant_index(KS)];
        default: return &mce::g_knn_kcap32[mce::mfma_variant_index(KS)];
    }
}

// Validates (nq, nr, d, K, self_mode) and lays out the workspace.  Pure function of its
// arguments so mce_knn_workspace_bytes() and the launcher always agree.
// Seed phase of the exhaustive fp16 sweep for splits of (at least) `cps` chunks: chunks | group tiles << 16, 0 = none
// (knn_f16.hpp: f16_seed_cfg).  MCE_F16_SEED_ROWS / MCE_F16_SEED_SHARE / MCE_F16_SEED_TG override (tests, tuning).
// Small splits: a quarter of the chunks (then half) in smaller groups, as long as they hold twice the K groups a bound
// needs.  Without a seed phase a query accepts ~K ln(n/K) candidates before its list settles; measured (fused call, none ->
// seeded): 32 k x 6, K = 3: 0.61 -> 0.33 ms; 100 k x 6 (C2): 1.23 -> 0.90; 65 k x 27, K = 9: 1.51 -> 0.96; 131 k x 27: 2.60 ->
// 1.93; 197 k x 27: 5.35 -> 4.10; from ~400 k rows the row budget binds as before.
int sweep_seed_cfg(int64_t cps, int CT, int kneed)
{
    const Tuning t = read_tuning();
    if (t.f16_seed_rows >= 0 || t.f16_seed_share >= 0 || t.f16_seed_tg >= 0)
        return mce::f16_seed_cfg(cps, CT, kneed, t.f16_seed_rows >= 0 ? t.f16_seed_rows : MCE_H_SEED_ROWS, t.f16_seed_share >= 0 ? t.f16_seed_share : MCE_H_SEED_SHARE,
                                 t.f16_seed_tg >= 0 ? t.f16_seed_tg : MCE_H_SEED_TG);
    for (int share = MCE_H_SEED_SHARE; share >= 2; share /= 2)
        for (int tg = MCE_H_SEED_TG; tg >= 2; tg /= 2)
            if (const int cfg = mce::f16_seed_cfg(cps, CT, kneed, MCE_H_SEED_ROWS, share, tg)) return cfg;
    return 0;
}

int make_plan(int64_t nq, int64_t nr, int32_t d, int32_t K, int32_t self_mode, Plan& p)
{
    if (nq < 0 || nr < 1 || d < 1 || K < 1) return fail(MCE_ERR_INVALID, "invalid sizes nq=%lld nr=%lld d=%d K=%d", (long long)nq, (long long)nr, d, K);
    if (self_mode < 0 || self_mode > 2) return fail(MCE_ERR_INVALID, "invalid self_mode %d", self_mode);
    if (d > mce::kGenMaxDim) return fail(MCE_ERR_DIM_RANGE, "d=%d exceeds the supported maximum %d", d, mce::kGenMaxDim);
    if (K > mce::kGenMaxK) return fail(MCE_ERR_K_RANGE, "K=%d exceeds the supported maximum %d", K, mce::kGenMaxK);
    const int64_t usable = (self_mode == MCE_SELF_EXCLUDE) ? nr - 1 : nr;
    if (K > usable)   // sklearn raises ValueError("Expected n_neighbors <= n_samples_fit")
        return fail(MCE_ERR_K_RANGE, "Expected n_neighbors <= n_samples_fit, but n_neighbors = %d, n_samples_fit = %lld", K, (long long)usable);
    if (nr >= (int64_t)1 << 31) return fail(MCE_ERR_INVALID, "nr=%lld exceeds 2^31-1 reference rows", (long long)nr);

    // 64 <= d <= 127 (round 5): the fp64 MFMA sweep at KS = 20..32, one query tile per wave -- the vector-FMA kernel below was
    // 66x the time of d = 63 at 100 k rows (knn_generic.hpp); it keeps d > 127 and K > 32
    const bool wide_f64 = d > MCE_MAX_DIM && d <= mce::kWideMaxDim && K <= MCE_MAX_K;
    // 128 <= d <= 1024 (round 6): the fp64 MFMA sweep with the k dimension in blocks (knn_long.hpp) -- the vector-FMA kernel below ran
    // these at 0.12 of the fp64 vector peak; it keeps K > 32.  (MCE_LONG=0: comparisons.)
    static const bool long_on = [] { const char* e = getenv("MCE_LONG"); return !(e && e[0] == '0'); }();
    if (long_on && d >= mce::kLongMinDim && d <= mce::kLongMaxDim && K <= MCE_MAX_K) {
        p.ksel = std::min<int>(K + kRefineMargin, MCE_MAX_K);          // (GEMM-form keys: the merge picks the K on exact distances)
        p.vl = &mce::g_knn_long[p.ksel <= 8 ? 0 : (p.ksel <= 16 ? 1 : 2)];
        p.v = nullptr;
        p.vh = nullptr;
        p.KCAP = p.vl->kcap;
        p.CT = p.vl->ct;
        p.QT = mce::kLongQT;
        p.KS = mce::long_ksp(d);
        p.nqblk = (int)std::max<int64_t>(1, (nq + mce::kLongQPB - 1) / mce::kLongQPB);
        p.nq_pad = (int64_t)p.nqblk * mce::kLongQPB;
        const int64_t rows_per_chunk = (int64_t)p.CT * 16;
        p.nchunk = (nr + rows_per_chunk - 1) / rows_per_chunk;
        p.nrow_pad = p.nchunk * rows_per_chunk;
        // reference splits: fill the chip (one 512-thread workgroup per CU) and trim the last partial round; every split re-pays a
        // list warm-up of a few chunks' worth of insertions
        int best_r = 1;
        double best_c = 1e300;
        const int rmax = (int)std::min<int64_t>(mce::kMaxLists, p.nchunk);
        for (int r = 1; r <= rmax; ++r) {
            const double c = std::ceil((double)p.nqblk * r / kAssumedCUs) * (std::ceil((double)p.nchunk / r) + 16.0);
            if (c < best_c * 0.98) { best_c = c; best_r = r; }
        }
        p.rsplit = best_r;
        p.L = p.rsplit;
        p.cost = best_c;
        p.pl_nr = nr;
        size_t off = 0;
        p.off_pd = off;
        off = align_up(off + (size_t)p.L * p.KCAP * (size_t)p.nq_pad * sizeof(double), 256);
        p.off_pi = off;
        off = align_up(off + (size_t)p.L * p.KCAP * (size_t)p.nq_pad * sizeof(int), 256);
        p.off_center = off;
        off = align_up(off + (size_t)(d + 4) * sizeof(double), 256);
        p.off_msum = off;
        off = align_up(off + (size_t)mce::kLongMeanBlocks * (size_t)d * sizeof(double), 256);
        p.off_yf = off;
        off = align_up(off + (size_t)p.nrow_pad * (size_t)p.KS * 4 * sizeof(double), 256);
        p.off_xf = off;
        off = align_up(off + (size_t)p.nq_pad * (size_t)p.KS * 4 * sizeof(double), 256);
        p.off_xn = off;
        off = align_up(off + (size_t)p.nq_pad * sizeof(double), 256);
        p.total = off + 256;
        return MCE_OK;
    }
    if ((d > MCE_MAX_DIM && !wide_f64) || K > MCE_MAX_K) {
        // outside the MFMA kernels' register budgets: plain exact kernel, lists [1][K][nq_pad]
        p.generic = true;
        p.v = nullptr;
        p.vh = nullptr;
        p.KCAP = K;
        p.rsplit = 1;
        p.L = 1;
        p.nqblk = (int)std::max<int64_t>(1, (nq + mce::kGenThreads - 1) / mce::kGenThreads);
        p.nq_pad = (int64_t)p.nqblk * mce::kGenThreads;
        size_t off = 0;
        p.off_pd = off;
        off = align_up(off + (size_t)K * (size_t)p.nq_pad * sizeof(double), 256);
        p.off_pi = off;
        off = align_up(off + (size_t)K * (size_t)p.nq_pad * sizeof(int), 256);
        p.off_center = off;      // (generic plans: scratch for the distance matrix of the fused path)
        off = align_up(off + (size_t)nq * K * sizeof(double), 256);
        p.off_msum = off;
        p.total = off + 256;
        return MCE_OK;
    }
    p.KS = mce::mfma_ks_for(d);
    int ki = 0;
    while (ki < mce::kNumKcap - 1 && mce::kKcapList[ki] < K) ++ki;
    p.KCAP = mce::kKcapList[ki];
    const bool f16 = !wide_f64 && (eff_search_mode() != 1) && mce::f16_supported(d, K);
    // 64 <= d <= 127 (round 6): the fp16 filter with 5, 6 or 8 k-steps (knn_deep.hpp), K <= 32 (beyond 16 in two passes); search mode 1
    // keeps the fp64 sweep's wide form.  (MCE_DEEP=0: comparisons.)
    static const bool deep_on = [] { const char* e = getenv("MCE_DEEP"); return !(e && e[0] == '0'); }();
    const bool deep = wide_f64 && (eff_search_mode() != 1) && mce::deep_supported(d, K) && deep_on;
    const bool filt = f16 || deep;          // fp16 operands in the workspace, 32-row tiles, 512-query blocks
    p.vd = nullptr;
    int qpb, rows_per_tile;
    if (deep) {
        if (K > 16) {                          // 16 nearest per reference split first, then the next K - 16 beyond them (knn_deep.hpp, LOWER)
            p.twopass = true;
            ki = 3;
            p.KCAP = 16;
        }
        p.KST = mce::deep_ksteps(d);
        const mce::KnnDeepVariant* tab = ki == 0 ? mce::g_knn_deep_kcap4 : ki == 1 ? mce::g_knn_deep_kcap8 : ki == 2 ? mce::g_knn_deep_kcap12 : mce::g_knn_deep_kcap16;
        p.vd = &tab[p.KST == 5 ? 0 : (p.KST == 6 ? 1 : 2)];
        p.v = nullptr;
        p.vh = nullptr;
        p.QT = mce::kHQT;
        p.CT = p.vd->ct;
        qpb = mce::f16_qpb(p.KCAP);
        rows_per_tile = 32;
    } else if (f16) {
        if (K > 16) {                          // 16 nearest per reference split first, then the next K - 16 beyond them
            p.twopass = true;
            ki = 3;
            p.KCAP = 16;
        }
        p.KST = mce::f16_ksteps(d);
        const mce::KnnF16Variant* tab = ki == 0 ? mce::g_knn_f16_kcap4 : ki == 1 ? mce::g_knn_f16_kcap8
                                        : ki == 2 ? mce::g_knn_f16_kcap12 : mce::g_knn_f16_kcap16;
        p.vh = &tab[p.KST - 1];
        p.v = nullptr;
        p.QT = p.vh->qt;
        p.CT = p.vh->ct;
        qpb = mce::f16_qpb(p.KCAP);
        rows_per_tile = 32;
    } else {
        // The fp64 sweep SELECTS on GEMM-form keys (|x|^2 + |y|^2 - 2 x.y: good to ~1e-16 |x|^2 absolute) and the merge refines
        // the selected pairs to exact direct-difference distances.  A near-tie at the K-th place could therefore hand the
        // merge the (K + 1)-th neighbour instead of the K-th (parity footnote a).  The lists carry
        // kRefineMargin more entries than asked for, so the final K are chosen on exact distances among K + 2 candidates
        // (a wrong row would need three rows within the keys' rounding of each other at the K-th place).
        p.ksel = std::min<int>(K + kRefineMargin, MCE_MAX_K);
        ki = 0;
        while (ki < mce::kNumKcap - 1 && mce::kKcapList[ki] < p.ksel) ++ki;
        p.KCAP = mce::kKcapList[ki];
        p.v = variant_for(p.KS, ki);
        p.vh = nullptr;
        p.QT = p.v->qt;
        p.CT = p.v->ct;
        qpb = mce::queries_per_block(p.QT);
        rows_per_tile = 16;
    }
    p.nqblk = (int)std::max<int64_t>(1, (nq + qpb - 1) / qpb);
    // One-k-step sweeps of short lists over many queries: four query tiles per wave, a workgroup = TWO query blocks (knn_f16.hpp,
    // QTT = 4: C4 41.5 -> 37.7 ms).  From kWideMinBlocks blocks: the chip then still gets about a full round of the wider workgroups
    // (C2's 196 blocks would leave 60 % of the CUs idle).  Decided on the sizes alone, so that the workspace query and the call
    // agree; which sweep runs (pruned walk, symmetric, exhaustive) is settled later -- only the exhaustive one has a wide form.
    p.wide_ok = f16 && !p.twopass && p.KST == 1 && p.vh->launch_wide && p.nqblk >= kWideMinBlocks && nr <= ((int64_t)1 << 25) && read_tuning().wide;
    if (p.wide_ok && (p.nqblk & 1)) p.nqblk += 1;
    p.nq_pad = (int64_t)p.nqblk * qpb;
    const int64_t rows_per_chunk = (int64_t)p.CT * rows_per_tile;
    p.nchunk = (nr + rows_per_chunk - 1) / rows_per_chunk;
    p.nrow_pad = p.nchunk * rows_per_chunk;

    if (f16 && !p.twopass && p.KST == 1 && d <= mce::kPruneMaxDim && p.vh->launch_prune && nq > 0 &&
        p.nrow_pad <= ((int64_t)1 << mce::kHRelBits) && (int64_t)p.nqblk * p.nchunk <= mce::kPruneMaxPairs) {
        const int pm = eff_prune_mode();
        // (the k-d ordering costs ~4 ms per million reference rows whatever the number of queries, and sparse
        // query sets make large query tiles: measured at 2 M x 6, separate sets, the walk wins from nq ~ nr/10)
        p.prune = pm == 2 || (pm == 0 && kPruneAutoMinRows[d] > 0 && nr >= kPruneAutoMinRows[d] && nq >= kPruneAutoMinQueries &&
                                           nq >= nr / 8);
    }
    if (p.prune) {
        // no chunk staging in this mode: a "chunk" is just a list entry of 64 tiles (one per lane) = an aligned
        // k-d subtree of 2048 rows
        p.CT = mce::kHPruneChunkTiles;
        p.nchunk = (nr + p.CT * 32 - 1) / (p.CT * 32);
        p.nrow_pad = p.nchunk * p.CT * 32;
        if (p.nrow_pad > ((int64_t)1 << mce::kHRelBits)) p.prune = false;
        if (!p.prune) {
            p.CT = p.vh->ct;
            p.nchunk = (nr + rows_per_chunk - 1) / rows_per_chunk;
            p.nrow_pad = p.nchunk * rows_per_chunk;
        }
    }
    // reference split r: more workgroups fill the chip and trim the last partial round
    // (one 512-thread workgroup per CU), but every split re-pays the list warm-up: a query
    // accepts ~K(1+ln(n/K)) candidates while streaming n references, each a serialised
    // whole-wave insertion.  Model (cycles per SIMD, fitted on MI355X):
    //   fp64 sweep : block(r) = 256*KS*tiles16(r)      + 1000 * 32   * K (1 + ln(n_r/K))
    //   fp16 filter: block(r) = 64*QT*KST*tiles32(r)   +  300 * 32QT * K (1 + ln(n_r/K))
    //   total(r)   = ceil(nqblk*r / CUs) * block(r),   n_r = nr/r
    int best_r = 1;
    double best_c = 1e300;
    const int rmax = (int)std::min<int64_t>(p.twopass ? mce::kMaxLists / 2 : mce::kMaxLists, p.nchunk);
    const int rmin = f16 ? (int)((nr + ((int64_t)1 << mce::kHRelBits) - 1) >> mce::kHRelBits) : 1;   // queue entries hold 26-bit row offsets
    if (deep && p.nrow_pad > ((int64_t)1 << mce::kHRelBits))      // (the deep kernel's queue entries hold ABSOLUTE 26-bit rows)
        return fail(MCE_ERR_INVALID, "reference set too large for the fp16-filter path at d = %d (nr=%lld): use search mode 1", d, (long long)nr);
    if (rmin > rmax) return fail(MCE_ERR_INVALID, "reference set too large for the fp16-filter path (nr=%lld)", (long long)nr);
    for (int r = std::max(1, rmin); r <= rmax; ++r) {
        const double n_r = (double)nr / r;
        const double lnf = 1.0 + std::log(std::max(1.0, n_r / K));
        const double block = filt ? 64.0 * p.QT * p.KST * (n_r / 32.0) + 300.0 * 32.0 * p.QT * K * lnf
                                  : 256.0 * p.KS * (n_r / 16.0) + 1000.0 * 32.0 * K * lnf;
        const double rounds = std::ceil((double)p.nqblk * r / kAssumedCUs);
        const double c = rounds * block;
        if (c < best_c * 0.98) { best_c = c; best_r = r; }   // need >2% gain to take a bigger split
    }
    // Searches of at most one round of workgroups (up to ~130 k queries): the sweep of such a set is mostly candidate
    // handling, which a seed phase cuts by half or more and which parallelises over the splits -- take the largest split
    // count that still fits one round AND leaves every split enough chunks for a seed phase (tools/_tmp scans, fused call,
    // model's choice -> this: 8 k x 6, K = 3: 0.41 -> 0.20 ms; 16 k x 6: 0.47 -> 0.23; 12 k x 27, K = 9: 0.75 -> 0.41; 16 k x 45:
    // 1.04 -> 0.47; from 24 k rows both agree).  Nothing seeded: the model's choice.
    if (f16 && !p.twopass && p.nqblk <= kAssumedCUs) {
        const int kneed = K + 1;      // (whatever the self mode: the workspace query does not know it, and the layout depends on r)
        for (int r = std::min(rmax, kAssumedCUs / p.nqblk); r >= std::max(1, rmin); --r)
            if (sweep_seed_cfg(p.nchunk / r, p.CT, kneed)) { best_r = r; break; }
    }
    if (deep && p.nqblk <= kAssumedCUs) {
        // at most one round of workgroups: as above -- the largest split count that fits the round and leaves every split four
        // seed groups' worth of tiles per neighbour (knn_deep.hpp: its seed phase needs K + 1 groups)
        for (int r = std::min(rmax, kAssumedCUs / p.nqblk); r >= 1; --r)
            if ((p.nchunk / r) * p.CT >= 4 * (K + 1)) { best_r = r; break; }
    }
    if (const int r = read_tuning().rsplit) {          // tuning
        if (r >= std::max(1, rmin) && r <= rmax) best_r = r;
    }
    if (p.prune) best_r = 1;                  // every workgroup walks its own chunk list
    p.rsplit = best_r;
    {   // the model's cost of the split count actually taken (the overrides above may have left its minimum)
        const double n_r = (double)nr / best_r;
        const double lnf = 1.0 + std::log(std::max(1.0, n_r / K));
        const double block = filt ? 64.0 * p.QT * p.KST * (n_r / 32.0) + 300.0 * 32.0 * p.QT * K * lnf : 256.0 * p.KS * (n_r / 16.0) + 1000.0 * 32.0 * K * lnf;
        best_c = std::ceil((double)p.nqblk * best_r / kAssumedCUs) * block;
    }
    p.cost = best_c;
    // the workspace layout for the split count taken; a part of a search that was handed the WHOLE search's workspace
    // (mce_knn_dotp_part_f64_dev: a row shard plans more reference splits than the whole set would, and every split has its
    // own lists) takes fewer splits until it fits (t_plan_cap)
    auto layout = [&]() {
    p.L = p.twopass ? 2 * p.rsplit : p.rsplit;

    size_t off = 0;
    p.off_yf = off;
    if (filt) {
        p.off_yh = off;
        off = align_up(off + (size_t)p.nrow_pad * (size_t)(16 * p.KST) * 2, 256);
        p.off_xh = off;
        off = align_up(off + (size_t)p.nq_pad * (size_t)(16 * p.KST) * 2, 256);
        p.off_qinfo = off;
        off = align_up(off + (size_t)p.nq_pad * 2 * sizeof(double), 256);
        p.off_params = off;
        off = align_up(off + (size_t)mce::HP_COUNT * sizeof(double), 256);
    } else {
        off = align_up(off + (size_t)p.nrow_pad * (size_t)(4 * p.KS) * sizeof(double), 256);
    }
    // symmetric sweep: auto-evidence searches (the caller passes ONE buffer as X and Y; only the sizes are known here)
    // (16 < K <= 32: two symmetric passes over 16-entry lists -- capi_search.hpp -- where the list capacity has the second-pass kernels)
    if (f16 && (!p.twopass || p.vh->launch_panel_lower) && !p.prune && nq == nr && p.vh->launch_sym && p.nqblk >= 2 && p.nrow_pad <= ((int64_t)1 << mce::kHSymRowBits)) {
        const int sm = eff_sym_mode();
        p.sym = sm == 2 || (sm == 0 && p.nqblk >= kSymAutoMinBlocks[p.KST] * ((p.KST == 1 && p.KCAP == 16) ? 2 : 1));     // (1M x 15, K = 16: 62.3 vs 62.7 ms)
        // the symmetric sweep needs queries and references to be ONE buffer: known from the pointers (host entry points, and
        // the *_dev ones at call time) or said by the caller of a workspace query (mce_options.same_set); unknown: reserve
        if (g_same_set_hint == 0 || (g_same_set_hint < 0 && t_opt.same_set == 0)) p.sym = false;
    }
    p.sym_nsplit = 1;
    p.sym_panel = 0;
    if (p.sym && !p.twopass) {
        const Tuning tun = read_tuning();
        // (on ONE GPU chains do not pay: measured at d = 27, K = 9, S = 1 / 2 / 3 / 6: 200 k rows 3.09 / 3.27 / 3.35 / 3.55 ms, 400 k 8.33 /
        //  8.59 / 8.83 / 9.37 -- every chain fills a list of its own, and a block's chain is short already; only below ~130 k rows
        //  did they win, 2.20 -> 2.05 ms.  Off unless MCE_SYM_CHAINS asks; a rank of the all-pairs-once partition, whose blocks are
        //  few and long, takes them: capi_apo.hpp)
        int S = 1;
        if (tun.sym_chains >= 1) S = std::min(tun.sym_chains, kSymMaxChains);
        p.sym_nsplit = S;
        if (S > 1) p.sym_panel = (int)std::max<int64_t>(8, std::min<int64_t>(kSymPanelChunks[p.KST], p.nchunk / (2 * S)));
    }
    const int l_alloc = std::max(p.L, p.sym_nsplit);
    p.off_pd = off;
    off = align_up(off + (size_t)l_alloc * p.KCAP * (size_t)p.nq_pad * sizeof(double), 256);
    p.off_pi = off;
    off = align_up(off + (size_t)l_alloc * p.KCAP * (size_t)p.nq_pad * sizeof(int), 256);
    p.off_center = off;
    off = align_up(off + (size_t)3 * mce::kMaxDimPad * sizeof(double), 256);      // centre | box(Y) | box(X)
    p.off_msum = off;
    off = align_up(off + (size_t)mce::kMeanBlocks * mce::kStatStride * sizeof(double), 256);
    if (p.prune) {
        mce::prune_layout(nq, p.nq_pad, p.nqblk, nr, p.nrow_pad, p.nchunk, d, p.pl);
        p.pl_nr = nr;
        p.off_prune = off;
        off = align_up(off + p.pl.total, 256);
        if (nq == nr && p.nqblk >= kPruneHeavyMinBlocks && prune_heavy_enabled()) {
            p.heavy_max = std::max(p.nqblk * mce::kHWaves / kPruneHeavyMaxShare, kPruneHeavyMinCount);       // waves
            p.off_heavy = off;
            off = align_up(off + (size_t)p.heavy_max * kPruneWaveQueries * (kPruneHeavyMaxSplit - 1) * p.KCAP * (sizeof(double) + sizeof(int)), 256);
        }
    }
    if (p.sym) {
        mce::sym_layout(nr, p.nq_pad, p.nqblk, d, p.KCAP, mce::f16_qpb(p.KCAP), sym_bucket_per_row(std::min<int>(K, p.KCAP)), p.sl);      // (two passes: each fills lists of KCAP)
        p.off_sym = off;
        off = align_up(off + p.sl.total, 256);
    }
    p.total = off;
    };
    layout();
    while (t_plan_cap && p.total > t_plan_cap && p.rsplit > std::max(1, rmin)) {
        p.rsplit -= 1;
        layout();
    }
    return MCE_OK;
}

size_t dotp_ws_bytes(int64_t nq, int32_t kmax)
{
    const int64_t nb = (nq + mce::kRedThreads - 1) / mce::kRedThreads;
    return align_up((size_t)std::max<int64_t>(nb, 1) * (size_t)kmax * sizeof(double), 256);
}

}  // namespace
