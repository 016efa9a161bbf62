// capi_common.hpp -- part of capi.hip (one translation unit): error reporting, per-thread search statistics, the tuning / test
// knobs (the ONE place the library reads the environment), per-call options, planner hints and the device / pinned memory pools.
#pragma once
namespace {

thread_local char g_err[512] = "";
thread_local char g_last_kernel[256] = "";
// optional timing of the dominant kernel with HIP events on the launch stream (bench.py)
thread_local int g_prof_on = 0;
thread_local std::vector<std::pair<hipEvent_t, hipEvent_t>> g_ev_pool;   // reused brackets
thread_local size_t g_ev_used = 0;                                        // brackets since enable
thread_local size_t g_ev_calls = 0;                                       // searches since enable (a split search: two brackets)
thread_local bool g_in_tail = false;                                      // inside the tail part of a split search
thread_local int g_split_depth = 0;                                       // > 0: inside a part of a split search
thread_local double g_last_prune_geom[3] = {0, 0, 0};                     // blocks, chunks, tiles per chunk
// what the matrix cores executed in the last search on this thread (mce_last_search_stats): flops of the dominant kernel and
// of every launch of the search (prepass / seed phases included); -1: not known on the host (pruned walk: device counters)
thread_local double g_last_flops_main = 0.0, g_last_flops_all = 0.0;
thread_local std::vector<std::pair<hipEvent_t, hipEvent_t>> g_evs_pool;  // brackets around the WHOLE search (packing .. last list kernel)
thread_local size_t g_evs_used = 0;

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define MCE_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(MCE_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// search mode: 0 auto (fp16 filter + fp64 refine where supported, else fp64 MFMA),
//              1 fp64 MFMA sweep only, 2 same as 0 (explicit)
std::atomic<int> g_mode{0};
// spatial pruning (prune.hpp): 0 auto (low d, large reference sets), 1 never, 2 whenever the shape allows it
std::atomic<int> g_prune_mode{0};
// ---------------------------------------------------------------------------------------------------------------------
// Tuning and test knobs.  This is the ONE place where the library reads the environment; every knob is result-neutral
// (the tests run the same searches under different settings and compare bit for bit) and exists for tests, bisecting and
// tuning -- production callers set none of them.  Read at every call (a dozen getenv per search), so tests can change
// them between calls; the workspace LAYOUT depends on sym / sym_bucket / rsplit / the seed knobs, which must therefore not
// change between a workspace query and the call that uses the workspace.
//   MCE_SYM=0|1|2                initial value of mce_set_sym_mode (read once)
//   MCE_SYM_KERNEL=f16           the symmetric sweep on knn_f16_kernel<.., SYM = 2> (round 2) instead of knn_panel_kernel
//   MCE_SYM_SPIN_LIMIT=n         ~microseconds a unit waits for its block's previous unit before it gives up (2^21)
//   MCE_SYM_BUCKET=n             row-side candidates per row the buckets hold (6K + 24)
//   MCE_SYM_PANEL=n              chunks per panel of reference rows (256; 96 with one k-step)
//   MCE_SYM_SEED_ROWS / _SHARE / _MODE   prepass: rows (32768; 65536), at most 1/share of the chunks (2), where (by k-steps)
//   MCE_F16_SEED_ROWS / _SHARE / _TG     seed phase of the exhaustive sweep: rows (24576), share (4), tiles per group (8)
//   MCE_RSPLIT=n                 reference splits of the exhaustive sweep (the model's choice)
//   MCE_TAIL_SPLIT=0             keep a search with a nearly empty last round of workgroups in one launch
//   MCE_WIDE=0                   exhaustive one-k-step sweep: two query tiles per wave also where four would be taken (A/B)
//   MCE_PANEL_DEBUG=bits         knn_panel.hpp test hooks (8: every candidate through the redo list, 16: waves give up waiting)
//   MCE_FEED_WAVE_BYTES=n        batched feed: bytes of host data per upload wave (tests: force several waves)
//   MCE_FEED_UPLOAD=async        batched feed: uploads on the job's stream (read once)
//   MCE_PRUNE_PROF=1             print the pruned walk's per-wave cycle breakdown (builds with -DMCE_PRUNE_PROF)
// ---------------------------------------------------------------------------------------------------------------------
struct Tuning {
    bool sym_kernel_f16 = false, tail_split = true, prune_prof = false, wide = true;
    int spin_limit = 1 << 21, sym_bucket = 0, sym_panel = 0, sym_seed_rows = 0, sym_seed_share = 2, sym_seed_mode = -1;
    int f16_seed_rows = -1, f16_seed_share = -1, f16_seed_tg = -1, rsplit = 0, panel_debug = 0, sym_chains = 0;
    size_t feed_wave_bytes = 0;
};
Tuning read_tuning()
{
    Tuning t;
    auto num = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
    const char* e = getenv("MCE_SYM_KERNEL");
    t.sym_kernel_f16 = e && strcmp(e, "f16") == 0;
    t.spin_limit = num("MCE_SYM_SPIN_LIMIT", 1 << 21);
    t.sym_bucket = num("MCE_SYM_BUCKET", 0);
    t.sym_panel = num("MCE_SYM_PANEL", 0);
    t.sym_chains = num("MCE_SYM_CHAINS", 0);
    t.sym_seed_rows = num("MCE_SYM_SEED_ROWS", 0);
    t.sym_seed_share = num("MCE_SYM_SEED_SHARE", 2);
    t.sym_seed_mode = num("MCE_SYM_SEED_MODE", -1);
    t.f16_seed_rows = num("MCE_F16_SEED_ROWS", -1);
    t.f16_seed_share = num("MCE_F16_SEED_SHARE", -1);
    t.f16_seed_tg = num("MCE_F16_SEED_TG", -1);
    t.rsplit = num("MCE_RSPLIT", 0);
    t.tail_split = num("MCE_TAIL_SPLIT", 1) != 0;
    t.wide = num("MCE_WIDE", 1) != 0;
    t.panel_debug = num("MCE_PANEL_DEBUG", 0);
    t.prune_prof = getenv("MCE_PRUNE_PROF") != nullptr;
    if ((e = getenv("MCE_FEED_WAVE_BYTES"))) t.feed_wave_bytes = (size_t)std::strtoull(e, nullptr, 10);
    return t;
}
// symmetric sweep of an auto-evidence search (knn_f16.hpp): 0 auto (large sets), 1 never, 2 whenever the shape allows it.
// MCE_SYM in the environment sets the initial value.
std::atomic<int> g_sym_mode{-1};
int sym_mode()
{
    int m = g_sym_mode.load();
    if (m < 0) {
        const char* e = getenv("MCE_SYM");
        m = e ? atoi(e) : 0;
        if (m < 0 || m > 2) m = 0;
        g_sym_mode.store(m);
    }
    return m;
}
// which kernel sweeps: knn_panel_kernel (default) or the SYM = 2 instantiation of knn_f16_kernel (MCE_SYM_KERNEL=f16: kept
// for comparisons).  A unit of the panel kernel waits for its block's previous unit at most this many ~1 us sleeps
// (MCE_SYM_SPIN_LIMIT; 0 in the tests: every wait that is not already satisfied gives up, and the repair launch takes over)
bool sym_use_panel_kernel() { return !read_tuning().sym_kernel_f16; }
int sym_spin_limit() { return read_tuning().spin_limit; }
// Per-call options (mce_options, include/mcevidence_hip.h): the *_opt entry points and mce_options_push / _pop set them for
// the calls the CURRENT THREAD makes; -1 = the process-wide default of the setters above.  Threads the library starts
// itself (one per device) inherit the caller's.  The planner reads the modes through these three functions only.
struct CallOptions { int search = -1, prune = -1, sym = -1, same_set = -1, verify = -1; };
thread_local CallOptions t_opt;
thread_local std::vector<CallOptions> t_opt_stack;
int eff_search_mode() { return t_opt.search >= 0 ? t_opt.search : g_mode.load(); }
int eff_prune_mode() { return t_opt.prune >= 0 ? t_opt.prune : g_prune_mode.load(); }
int eff_sym_mode() { return t_opt.sym >= 0 ? t_opt.sym : sym_mode(); }
// Rows re-checked after a host-pointer search (mce_options.verify; verify_kernels.hpp).  An explicit value (>= 0; 0 = off) wins.
// Unset (-1): searches that took the fp16 FILTER -- whose exactness rests on a rounding model of the matrix core (derived from one
// assumption and measured: docs/design/sweep_f16_exhaustive.md), not on fp64 arithmetic throughout -- are certified on
// kVerifyDefaultRows rows by default (round 6; MCE_VERIFY=n in the environment changes the number, MCE_VERIFY=0 turns the default
// off); the fp64 sweep and the generic kernel are not.  What "by default" costs was measured (bench.py: `certificate`): 0.4 - 1.1 ms
// of a 37 ms search (C3, C4), 3.5 of 73.5 (C5: the distances of 10 M rows have to be written out for it) -- but 0.16 ms of a 0.68 ms
// call on a Planck-sized chain, which is launches and one synchronisation, not rows.  So searches of kVerifyAlwaysFrom query rows and
// more are certified on EVERY call, smaller ones on one call in kVerifySmallEvery (a per-thread counter: a loop over thousands of
// small chains still samples hundreds of searches, at 3 % instead of 24 %).
constexpr int kVerifyDefaultRows = 256;
constexpr int64_t kVerifyAlwaysFrom = 65536;
constexpr int kVerifySmallEvery = 8;
int default_verify_rows()
{
    static const int v = [] {
        const char* e = getenv("MCE_VERIFY");
        if (!e || !*e) return kVerifyDefaultRows;
        const long n = strtol(e, nullptr, 10);
        return (int)std::max<long>(0, std::min<long>(n, 1 << 20));
    }();
    return v;
}
bool verify_env_set() { static const bool s = [] { const char* e = getenv("MCE_VERIFY"); return e && *e; }(); return s; }
std::atomic<int> g_last_verify_rows{0};     // mce_last_verify_rows(): rows the certificate of the most recent host-pointer search checked
thread_local unsigned t_small_search_count = 0;
int eff_verify(bool filter_path, int64_t nq)
{
    if (t_opt.verify >= 0) return t_opt.verify;
    if (!filter_path) return 0;
    if (nq < kVerifyAlwaysFrom && !verify_env_set() && (t_small_search_count++ % kVerifySmallEvery) != 0) return 0;
    return default_verify_rows();
}
// query blocks (512 rows each) from which the automatic mode takes it, by 16-wide k-steps of the filter.  Measured with the
// panel kernel (tools/sym_crossover.py -> profiles/r03_panel/crossover.json; fused search + reduction, exhaustive -> symmetric,
// ms).  Up to ~256 blocks -- one round of workgroups -- the seeded exhaustive sweep with its reference splits is faster
// (d = 27: 0.68 vs 0.96 at 32 k rows, 1.03 vs 1.36 at 65 k, 2.03 vs 2.20 at 131 k); past that the symmetric sweep wins at once
// where the filter takes two k-steps or more (d = 27: 4.2 -> 3.0 at 197 k, 6.0 -> 4.2 at 262 k, 19.8 -> 12.8 at 524 k,
// 66.2 -> 38.3 at 1 M; d = 45: 2.06 -> 2.02 at 98 k, 2.70 -> 2.52 at 131 k, 5.7 -> 3.6 at 197 k, 96.2 -> 50.4 at 1 M) and from
// ~0.4 M rows with one k-step (d = 15 / 10 / 6: 3.56 -> 3.44 / 3.53 -> 3.79 / 3.48 -> 4.48 at 262 k, 7.1 -> 6.1 / 7.0 -> 6.2 /
// 7.0 -> 7.4 at 393 k, 11.9 -> 9.7 / 11.6 -> 9.9 / 11.5 -> 11.1 at 524 k, 40.6 -> 28.0 / 39.5 -> 28.5 / 39.0 -> 30.7 at 1 M;
// d <= 6 from 300 k rows: the pruned walk takes over before that).  The more of a search is MFMA work, the more halving the
// products pays.
constexpr int kSymAutoMinBlocks[5] = {0, 768, 257, 193, 193};
// Chains per block (round 5; PanelGeom.nsplit).  A block's units hand its register lists on one after the other; a launch whose
// blocks are few and long is as slow as its longest block's chain.  Its panels can be dealt to S chains with their own list sets,
// merged at the end: what a rank of the all-pairs-once partition does (capi_apo.hpp); on one GPU only when MCE_SYM_CHAINS = S asks
// (capi_plan.hpp has the measurement).
constexpr int kSymMaxChains = 8;
// prepass rows by k-steps: 0 spread over the sorted rows, 1 the rows nearest the mean, 2 half and half.  Fused call, ms,
// spread / nearest / half: 1M x 3 42.6 / 152.6 / 45.2; 1M x 6 36.1 / 50.1 / 37.1; 1M x 10 34.4 / 36.8 / 34.8; 1M x 15
// 33.8 / 33.4 / 33.2; 1M x 20 50.2 / 45.9 / 46.8; 1M x 27 48.8 / 44.4 / 45.5 (tools/sym_seedmode.py)
constexpr int kSymSeedMode[5] = {0, 0, 1, 1, 1};
// largest rank count for which a multi-GPU auto-evidence search is partitioned symmetrically (mce_knn_dotp_part_f64)
constexpr int kSymPartitionMaxParts = 4;
// The planner sees sizes only; the host-pointer entry points see the pointers.  They say here whether queries and
// references are one buffer, so that cross evidence with equal halves (split = True, s1frac = 0.5: nq == nr) does not
// reserve ~1 GB of scratch it can never use.  -1: unknown (the *_dev entry points: the workspace query must cover both).
thread_local int g_same_set_hint = -1;
// make_plan's workspace cap (0: none), see there
thread_local size_t t_plan_cap = 0;
struct PlanCap {
    size_t prev;
    explicit PlanCap(size_t cap) : prev(t_plan_cap) { t_plan_cap = cap; }
    ~PlanCap() { t_plan_cap = prev; }
};
// the heavy-wave side lists of a pruned same-set plan (make_plan) are reserved only while the split is enabled -- by default it
// is not (kPruneHeavyDefault) -- or asked for through MCE_PRUNE_HEAVY; t_no_heavy plans without them (a workspace that was sized
// before the variable was set: the walk then runs unsplit, hv_n = 0)
thread_local bool t_no_heavy = false;
struct NoHeavy {
    bool prev;
    NoHeavy() : prev(t_no_heavy) { t_no_heavy = true; }
    ~NoHeavy() { t_no_heavy = prev; }
};
struct SameSetHint {
    int prev;
    explicit SameSetHint(bool same) : prev(g_same_set_hint) { g_same_set_hint = same ? 1 : 0; }
    ~SameSetHint() { g_same_set_hint = prev; }
};
// 48 KB chunks per panel of reference rows (sym_types.hpp, units), by k-steps.  With the blocks of a panel dispatched from
// the last one down the length hardly matters above ~150 chunks at d = 27 (1 M rows: 64 -> 41.6 ms, 96 -> 40.8, 160 -> 40.1,
// 256 -> 39.9, 512 -> 40.1, one panel 41.1; 200 k rows: 96 -> 2.95, 384 -> 2.85); one k-step (d <= 16): 96 -> 32.5 ms,
// 256 -> 33.6 at 1 M x 15, the other way round at 262 k (4.16 vs 3.84)
constexpr int kSymPanelChunks[5] = {0, 96, 256, 256, 256};
// bucket entries per row: a row receives ~K ln(N/2 / seed rows) + K row-side candidates; MCE_SYM_BUCKET overrides (tests)
int sym_bucket_per_row(int K)
{
    const int b = read_tuning().sym_bucket;
    return b > 0 ? b : 6 * K + 24;
}
// measured on MI355X (tools/prune_sweep.sh, tools/prune_sweep_small.sh; search + preparation, K = 10):
//   d = 1: 0.3 M 3.3 vs 54 ms, 1 M 6.8 vs 474 ms      d = 2: 0.3 M 2.9 vs 11.8 ms
//   d = 3: 0.1 M 1.3 vs 1.5 ms, 1 M 17 vs 59 ms, 10 M 0.17 vs 4.0 s      d = 6: 0.2 M 4.6 vs 4.0, 0.3 M 8.5 vs 10.8,
//   1 M 28 vs 58 ms, 4 M 0.14 vs 0.69 s, 10 M 0.37 vs 3.95 s      d = 7: 1 M 48 vs 59 ms      d = 8: 2 M 197 vs 200,
//   3 M 342 vs 401, 4 M 504 vs 694 ms      d = 10: 4 M 1.43 vs 0.69 s (the boxes overlap too much)
// round 4 (tools/prune_crossover.sh, K = 4 / 9, against the automatic exhaustive / symmetric choice, after the preparation
// lost a quarter of its time and the symmetric sweep got its own kernel): d = 4: 0.1 M 1.37 vs 1.07, 0.15 M 1.56 vs 1.96 ms
//   d = 5: 0.15 M 2.48 vs 1.95, 0.2 M 2.62 vs 3.10, 0.3 M 3.4 vs 5.2      d = 6: 0.2 M 3.4 vs 3.1, 0.3 M 5.3 vs 5.2 (K = 9: 6.6 vs 6.8)
//   d = 7: 0.5 M 11.8 vs 9.8, 0.8 M 19.3 vs 20.6      d = 8: 1 M 45 vs 29, 2 M 105 vs 96 ms (the symmetric sweep moved the
//   crossover up: 3 M)      d = 9: 4 M 557 vs 349 ms
// round 4, after the per-query reach test (knn_f16.hpp: query_reach; the walk multiplies a sixth of the tiles it did) and the
// batched loads -- profiles/r04_final/prune_crossover.txt, pruned vs the automatic choice, ms: d = 3: 0.1 M 0.96 vs 1.06
//   d = 4: 0.1 M 1.02 vs 1.05      d = 5: 0.1 M 1.24 vs 1.06, 0.15 M 1.49 vs 1.97      d = 6: 0.1 M 1.50 vs 1.07, 0.15 M 1.80 vs
//   1.98, 0.3 M 3.3 vs 5.2      d = 7: 0.2 M 3.30 vs 3.08, 0.3 M 4.99 vs 5.18, 0.8 M 12.0 vs 20.5      d = 8: 0.5 M 11.3 vs 9.6,
//   1 M 23.9 vs 29.0, 2 M 53 vs 97
// smallest reference set for which the automatic mode takes the pruned walk, by dimension (0: never)
constexpr int64_t kPruneAutoMinQueries = 32768;
// (with the reach test on the chunk boxes too: d = 7: 0.2 M 3.16 vs 3.05, 0.3 M 4.45 vs 5.38      d = 8: 0.3 M 5.95 vs 5.23, 0.5 M 9.56 vs 9.58
//  (K = 9: 12.6 vs 13.8), 1 M 17.9 vs 29.0, 2 M 33 vs 97)
constexpr int64_t kPruneAutoMinRows[16] = {0, 100000, 100000, 100000, 100000, 125000, 150000, 250000, 500000, 0, 0, 0, 0, 0, 0, 0};

// Device buffers of the host-pointer entry points.  Small allocations (<= 64 MB) are kept in a
// per-thread, per-device pool between calls: the reference's typical workload is thousands of
// Planck-sized chains (planck_mcevidence.py:306-348), where seven hipMalloc/hipFree pairs per call
// would cost more than the kernels.  Larger buffers are allocated and freed per call.
// mce_release_device_memory() empties the pool.
constexpr size_t kPoolMaxBytes = (size_t)64 << 20;
constexpr int kPoolSlots = 16;
struct PoolSlot { void* p = nullptr; size_t cap = 0; int dev = -1; bool busy = false; };
struct Pool {
    PoolSlot slot[kPoolSlots];
    PoolSlot& operator[](int i) { return slot[i]; }
    void release_idle()
    {
        for (int i = 0; i < kPoolSlots; ++i)
            if (slot[i].p && !slot[i].busy) {
                int cur = 0;
                (void)hipGetDevice(&cur);
                if (slot[i].dev != cur) (void)hipSetDevice(slot[i].dev);
                (void)hipFree(slot[i].p);
                if (slot[i].dev != cur) (void)hipSetDevice(cur);
                slot[i] = PoolSlot();
            }
    }
    ~Pool() { release_idle(); }   // worker threads of the multi-device paths give their buffers back
};
thread_local Pool g_pool;

// pinned host staging for the small result copies of the feed path (grow-only, per thread)
struct PinnedArena {
    void* p = nullptr;
    size_t cap = 0;
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        release();
        const size_t c = std::max<size_t>(n + n / 2, (size_t)64 << 10);
        hipError_t e = hipHostMalloc(&p, c, hipHostMallocPortable);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = c;
        return hipSuccess;
    }
    ~PinnedArena() { release(); }
};
thread_local PinnedArena g_pinned;

thread_local const double* g_last_params = nullptr;                       // HP_* scalars of the last pruned launch (in the caller's workspace)

struct DevBuf {
    void* p = nullptr;
    int slot = -1;
    ~DevBuf()
    {
        // a host-pointer entry point is returning its scratch: the statistics of a pruned search that
        // lived in it are gone with it
        if (p && g_last_params >= static_cast<const double*>(p) &&
            reinterpret_cast<const char*>(g_last_params) < static_cast<const char*>(p) + bytes)
            g_last_params = nullptr;
        if (slot >= 0) g_pool[slot].busy = false;
        else if (p) (void)hipFree(p);
    }
    size_t bytes = 0;
    hipError_t alloc(size_t n)
    {
        if (n == 0) n = 1;
        bytes = n;
        if (n <= kPoolMaxBytes) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            int pick = -1;
            for (int i = 0; i < kPoolSlots; ++i)          // best fit among idle slots of this device
                if (!g_pool[i].busy && g_pool[i].p && g_pool[i].dev == dev && g_pool[i].cap >= n &&
                    (pick < 0 || g_pool[i].cap < g_pool[pick].cap)) pick = i;
            if (pick < 0)
                for (int i = 0; i < kPoolSlots; ++i)
                    if (!g_pool[i].busy) {                 // (re)allocate an idle slot
                        if (g_pool[i].p) { (void)hipFree(g_pool[i].p); g_pool[i].p = nullptr; g_pool[i].cap = 0; }
                        const size_t cap = n + n / 4;
                        hipError_t e = hipMalloc(&g_pool[i].p, cap);
                        if (e != hipSuccess) { g_pool[i].p = nullptr; return e; }
                        g_pool[i].cap = cap;
                        g_pool[i].dev = dev;
                        pick = i;
                        break;
                    }
            if (pick >= 0) {
                g_pool[pick].busy = true;
                slot = pick;
                p = g_pool[pick].p;
                return hipSuccess;
            }
        }
        return hipMalloc(&p, n);
    }
    template <class T> T* as() { return static_cast<T*>(p); }
};

}  // namespace
