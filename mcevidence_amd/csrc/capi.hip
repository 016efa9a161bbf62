// capi.hip -- the C ABI of libmcevidence_hip.so (see include/mcevidence_hip.h).
//
// Host-side planning, workspace carving, kernel dispatch and the host-pointer
// convenience wrappers.  No torch, no Python: plain HIP runtime calls.
#include "../../include/mcevidence_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "knn_dispatch.hpp"
#include "knn_mfma.hpp"
#include "knn_panel.hpp"
#include "pack_refs.hpp"
#include "f16_prep.hpp"
#include "knn_generic.hpp"
#include "feeders.hpp"
#include "reduce_kernels.hpp"
#include "prune.hpp"
#include "zero_fill.hpp"

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
//   MCE_PANEL_DEBUG=bits         knn_panel.hpp test hooks (8: every candidate through the redo list, 16: waves give up waiting)
//   MCE_FEED_WAVE_BYTES=n        batched feed: bytes of host data per upload wave (tests: force several waves)
//   MCE_FEED_UPLOAD=async        batched feed: uploads on the job's stream (read once)
//   MCE_PRUNE_PROF=1             print the pruned walk's per-wave cycle breakdown (builds with -DMCE_PRUNE_PROF)
// ---------------------------------------------------------------------------------------------------------------------
struct Tuning {
    bool sym_kernel_f16 = false, tail_split = true, prune_prof = false;
    int spin_limit = 1 << 21, sym_bucket = 0, sym_panel = 0, sym_seed_rows = 0, sym_seed_share = 2, sym_seed_mode = -1;
    int f16_seed_rows = -1, f16_seed_share = -1, f16_seed_tg = -1, rsplit = 0, panel_debug = 0;
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
    t.sym_seed_rows = num("MCE_SYM_SEED_ROWS", 0);
    t.sym_seed_share = num("MCE_SYM_SEED_SHARE", 2);
    t.sym_seed_mode = num("MCE_SYM_SEED_MODE", -1);
    t.f16_seed_rows = num("MCE_F16_SEED_ROWS", -1);
    t.f16_seed_share = num("MCE_F16_SEED_SHARE", -1);
    t.f16_seed_tg = num("MCE_F16_SEED_TG", -1);
    t.rsplit = num("MCE_RSPLIT", 0);
    t.tail_split = num("MCE_TAIL_SPLIT", 1) != 0;
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
struct CallOptions { int search = -1, prune = -1, sym = -1, same_set = -1; };
thread_local CallOptions t_opt;
thread_local std::vector<CallOptions> t_opt_stack;
int eff_search_mode() { return t_opt.search >= 0 ? t_opt.search : g_mode.load(); }
int eff_prune_mode() { return t_opt.prune >= 0 ? t_opt.prune : g_prune_mode.load(); }
int eff_sym_mode() { return t_opt.sym >= 0 ? t_opt.sym : sym_mode(); }
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
// smallest reference set for which the automatic mode takes the pruned walk, by dimension (0: never)
constexpr int64_t kPruneAutoMinQueries = 32768;
constexpr int64_t kPruneAutoMinRows[16] = {0, 100000, 100000, 100000, 150000, 300000, 300000, 800000, 2000000, 0, 0, 0, 0, 0, 0, 0};

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


using mce::kMaxDevices;
constexpr int kAssumedCUs = 256;   // MI355X; only steers the reference-split heuristic

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Plan {
    const mce::KnnVariant* v = nullptr;
    const mce::KnnF16Variant* vh = nullptr;   // non-null: fp16-filter path
    bool generic = false;                     // plain exact kernel (d > 63 or K > 32)
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
    bool twopass = false;                     // fp16 filter, 16 < K <= 32: two sweeps of 16-entry lists (knn_f16.hpp, LOWER)
    bool prune = false;                       // fp16 filter walking k-d ordered chunk lists (prune.hpp)
    int part = 0, nparts = 1;                 // pruned walk over query blocks part, part + nparts, ... only; symmetric sweep: the
                                              // contiguous range of sorted blocks [sym_qb_lo, sym_qb_hi) (one rank's share)
    int sym_qb_lo = 0, sym_qb_hi = 0;         // set by run_search when the symmetric sweep ran
    int64_t pl_nr = 0;                        // reference rows the plan was made for
    mce::PruneLayout pl;
    size_t off_prune = 0;
    bool sym = false;                         // workspace holds the symmetric sweep's scratch (run_search decides: X and Y must be one buffer)
    bool sym_active = false;                  // set by run_search: the lists are in sorted-row order, one split
    mce::SymLayout sl;
    size_t off_sym = 0;
};

const mce::KnnVariant* variant_for(int KS, int kcap_idx)
{
    switch (kcap_idx) {
        case 0: return &mce::g_knn_kcap4[KS - 1];
        case 1: return &mce::g_knn_kcap8[KS - 1];
        case 2: return &mce::g_knn_kcap12[KS - 1];
        case 3: return &mce::g_knn_kcap16[KS - 1];
        case 4: return &mce::g_knn_kcap24[KS - 1];
        default: return &mce::g_knn_kcap32[KS - 1];
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

    if (d > MCE_MAX_DIM || K > MCE_MAX_K) {
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
    p.KS = (d + 1 + 3) / 4;
    int ki = 0;
    while (ki < mce::kNumKcap - 1 && mce::kKcapList[ki] < K) ++ki;
    p.KCAP = mce::kKcapList[ki];
    const bool f16 = (eff_search_mode() != 1) && mce::f16_supported(d, K);
    int qpb, rows_per_tile;
    if (f16) {
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
        p.v = variant_for(p.KS, ki);
        p.vh = nullptr;
        p.QT = p.v->qt;
        p.CT = p.v->ct;
        qpb = mce::queries_per_block(p.QT);
        rows_per_tile = 16;
    }
    p.nqblk = (int)std::max<int64_t>(1, (nq + qpb - 1) / qpb);
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
    if (rmin > rmax) return fail(MCE_ERR_INVALID, "reference set too large for the fp16-filter path (nr=%lld)", (long long)nr);
    for (int r = std::max(1, rmin); r <= rmax; ++r) {
        const double n_r = (double)nr / r;
        const double lnf = 1.0 + std::log(std::max(1.0, n_r / K));
        const double block = f16 ? 64.0 * p.QT * p.KST * (n_r / 32.0) + 300.0 * 32.0 * p.QT * K * lnf
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
    if (const int r = read_tuning().rsplit) {          // tuning
        if (r >= std::max(1, rmin) && r <= rmax) best_r = r;
    }
    if (p.prune) best_r = 1;                  // every workgroup walks its own chunk list
    p.rsplit = best_r;
    {   // the model's cost of the split count actually taken (the overrides above may have left its minimum)
        const double n_r = (double)nr / best_r;
        const double lnf = 1.0 + std::log(std::max(1.0, n_r / K));
        const double block = f16 ? 64.0 * p.QT * p.KST * (n_r / 32.0) + 300.0 * 32.0 * p.QT * K * lnf : 256.0 * p.KS * (n_r / 16.0) + 1000.0 * 32.0 * K * lnf;
        best_c = std::ceil((double)p.nqblk * best_r / kAssumedCUs) * block;
    }
    p.cost = best_c;
    p.L = p.twopass ? 2 * p.rsplit : p.rsplit;

    size_t off = 0;
    p.off_yf = off;
    if (f16) {
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
    if (f16 && !p.twopass && !p.prune && nq == nr && p.vh->launch_sym && p.nqblk >= 2 && p.nrow_pad <= ((int64_t)1 << mce::kHSymRowBits)) {
        const int sm = eff_sym_mode();
        p.sym = sm == 2 || (sm == 0 && p.nqblk >= kSymAutoMinBlocks[p.KST] * ((p.KST == 1 && p.KCAP == 16) ? 2 : 1));     // (1M x 15, K = 16: 62.3 vs 62.7 ms)
        // the symmetric sweep needs queries and references to be ONE buffer: known from the pointers (host entry points, and
        // the *_dev ones at call time) or said by the caller of a workspace query (mce_options.same_set); unknown: reserve
        if (g_same_set_hint == 0 || (g_same_set_hint < 0 && t_opt.same_set == 0)) p.sym = false;
    }
    const int l_alloc = p.L;
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
    }
    if (p.sym) {
        mce::sym_layout(nr, p.nq_pad, p.nqblk, d, p.KCAP, mce::f16_qpb(p.KCAP), sym_bucket_per_row(K), p.sl);
        p.off_sym = off;
        off = align_up(off + p.sl.total, 256);
    }
    p.total = off;
    return MCE_OK;
}

size_t dotp_ws_bytes(int64_t nq, int32_t kmax)
{
    const int64_t nb = (nq + mce::kRedThreads - 1) / mce::kRedThreads;
    return align_up((size_t)std::max<int64_t>(nb, 1) * (size_t)kmax * sizeof(double), 256);
}

double ln_unit_ball(int d) { return 0.5 * d * std::log(M_PI) - std::lgamma(1.0 + 0.5 * d); }

// pack + search; leaves the lane/split lists in the workspace
int run_search(Plan& p, const double* dX, int64_t nq, const double* dY, int64_t nr, int32_t d, int32_t K,
               int32_t self_mode, int64_t self_offset, char* ws, hipStream_t st)
{
    p.sym_active = false;
    double* pd = reinterpret_cast<double*>(ws + p.off_pd);
    int* pi = reinterpret_cast<int*>(ws + p.off_pi);
    if (p.generic) {
        const size_t lds = (size_t)mce::kGenTileRows * d * sizeof(double);
        hipLaunchKernelGGL(mce::knn_generic_kernel, dim3((unsigned)p.nqblk), dim3(mce::kGenThreads), lds, st, dX, nq, dY, nr, (int)d,
                           (int)K, p.nq_pad, (self_mode == MCE_SELF_EXCLUDE) ? 1 : 0, self_offset, pd, pi);
        MCE_HIP(hipGetLastError());
        snprintf(g_last_kernel, sizeof(g_last_kernel), "knn_generic_kernel grid=%d block=%d lds=%zu", p.nqblk, mce::kGenThreads, lds);
        return MCE_OK;
    }
    double* center = reinterpret_cast<double*>(ws + p.off_center);
    double* msum = reinterpret_cast<double*>(ws + p.off_msum);
    double* box_y = center + mce::kMaxDimPad;
    double* box_x = center + 2 * mce::kMaxDimPad;
    hipLaunchKernelGGL(mce::col_stats_partial_kernel, dim3(mce::kMeanBlocks), dim3(mce::kMeanThreads), 0, st, dY, nr, (int)d, msum);
    MCE_HIP(hipGetLastError());
    hipLaunchKernelGGL(mce::col_stats_final_kernel, dim3(1), dim3(64), 0, st, msum, nr, (int)d, center, box_y);
    MCE_HIP(hipGetLastError());
    const bool prof = g_prof_on && g_ev_used < 1024;
    // bracket of the whole search: closed by the caller-visible end of run_search (SearchBracket's destructor)
    struct SearchBracket {
        hipStream_t st; bool on;
        SearchBracket(hipStream_t s, bool o) : st(s), on(o)
        {
            if (!on) return;
            if (g_evs_used == g_evs_pool.size()) {
                hipEvent_t e0, e1;
                if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { on = false; return; }
                g_evs_pool.emplace_back(e0, e1);
            }
            (void)hipEventRecord(g_evs_pool[g_evs_used].first, st);
        }
        ~SearchBracket() { if (on) { (void)hipEventRecord(g_evs_pool[g_evs_used].second, st); ++g_evs_used; } }
    } search_bracket(st, prof && g_evs_used < 1024);
    g_last_flops_main = g_last_flops_all = 0.0;
    auto prof_begin = [&]() -> int {
        if (!prof) return MCE_OK;
        if (g_ev_used == g_ev_pool.size()) {
            hipEvent_t e0, e1;
            MCE_HIP(hipEventCreate(&e0));
            MCE_HIP(hipEventCreate(&e1));
            g_ev_pool.emplace_back(e0, e1);
        }
        MCE_HIP(hipEventRecord(g_ev_pool[g_ev_used].first, st));
        return MCE_OK;
    };
    auto prof_end = [&]() -> int {
        if (!prof) return MCE_OK;
        MCE_HIP(hipEventRecord(g_ev_pool[g_ev_used].second, st));
        ++g_ev_used;
        if (!g_in_tail) ++g_ev_calls;
        return MCE_OK;
    };
    const int threads = 256;
    if (p.vh) {
        // ---- fp16 filter + exact fp64 refine ------------------------------------
        _Float16* yh = reinterpret_cast<_Float16*>(ws + p.off_yh);
        _Float16* xh = reinterpret_cast<_Float16*>(ws + p.off_xh);
        double* qinfo = reinterpret_cast<double*>(ws + p.off_qinfo);
        double* params = reinterpret_cast<double*>(ws + p.off_params);
        MCE_HIP(mce::zero_async(params, mce::HP_COUNT * sizeof(double), st));
        // queries that are literally rows of the reference buffer are inside its bounding box already
        bool separate_queries = !(dX >= dY && dX + (size_t)nq * d <= dY + (size_t)nr * d);
        const double* sX = dX;     // the rows the search reads: the caller's, or their k-d ordered copies
        const double* sY = dY;
        mce::PruneOut po;
        const bool use_sym = p.sym && dX == dY && nq == nr && self_offset == 0 && g_split_depth == 0;
        if (use_sym) {
            // rows by distance from the mean: a 32-row tile then holds rows of nearly equal K-th neighbour distance
            MCE_HIP(mce::sym_prepare(dY, nr, (int)d, center, p.nq_pad, ws + p.off_sym, p.sl, st));
            sX = sY = reinterpret_cast<const double*>(ws + p.off_sym + p.sl.Ys);
            separate_queries = false;
        }
        if (p.prune) {
            const bool same_set = (dX == dY && nq == nr);
            MCE_HIP(mce::prune_prepare(dX, nq, dY, nr, (int)d, same_set, mce::f16_qpb(p.KCAP), p.CT * 32, p.nq_pad, p.nqblk, p.nrow_pad,
                                       p.nchunk, ws + p.off_prune, p.pl, st, po));
            sX = po.Xs;
            sY = po.Ys;
            separate_queries = separate_queries && !same_set;
        }
        if (separate_queries) {
            hipLaunchKernelGGL(mce::col_stats_partial_kernel, dim3(mce::kMeanBlocks), dim3(mce::kMeanThreads), 0, st, dX, nq, (int)d, msum);
            MCE_HIP(hipGetLastError());
            hipLaunchKernelGGL(mce::f16_box_about_kernel, dim3(1), dim3(64), 0, st, msum, (int)d, center, box_x);
            MCE_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(mce::f16_scale_kernel, dim3(1), dim3(64), 0, st, box_y, separate_queries ? box_x : (const double*)nullptr, params);
        MCE_HIP(hipGetLastError());
        {
            const int64_t rows_per_block = 4 * (64 / (2 * p.KST));          // 4 waves x R rows
            const int64_t pack_blocks = std::min<int64_t>((p.nrow_pad + rows_per_block - 1) / rows_per_block, 2048);   // grid-stride
            hipLaunchKernelGGL(mce::f16_pack_refs_kernel, dim3((unsigned)pack_blocks), dim3(256), 0, st,
                               sY, nr, (int)d, p.KST, p.nrow_pad, center, params, yh);
            MCE_HIP(hipGetLastError());
            hipLaunchKernelGGL(mce::f16_pack_queries_kernel, dim3((unsigned)((p.nq_pad + rows_per_block - 1) / rows_per_block)), dim3(256), 0, st,
                               sX, nq, p.nq_pad, (int)d, p.KST, center, params, xh, qinfo);
            MCE_HIP(hipGetLastError());
        }
        mce::KnnF16Args a;
        a.Yh = yh; a.nchunk_total = p.nchunk; a.rsplit = p.rsplit; a.Xh = xh; a.qinfo = qinfo; a.params = params;
        a.X = sX; a.Y = sY; a.nq = nq; a.nr = nr; a.D = d; a.nq_pad = p.nq_pad; a.nqblk = p.nqblk;
        a.self_exclude = (self_mode == MCE_SELF_EXCLUDE) ? 1 : 0;
        a.self_offset = self_offset; a.ksel = K; a.part_d = pd; a.part_i = pi;
        if (p.prune) {
            a.clist = po.clist; a.cdist = po.cdist; a.list_len = (int)p.nchunk; a.rperm = po.rperm; a.qperm = po.qperm;
            a.tbox_r = po.tbox_r; a.tbox_q = po.tbox_q; a.cbox_r = po.cbox_r; a.border = po.border;
            if (p.nparts > 1) { a.qblk0 = p.part; a.qblk_stride = p.nparts; a.nqblk_run = (p.nqblk - p.part + p.nparts - 1) / p.nparts; }
            int rc = prof_begin();
            if (rc != MCE_OK) return rc;
            MCE_HIP(p.vh->launch_prune(a, st));
            rc = prof_end();
            if (rc != MCE_OK) return rc;
            g_last_params = params;
            g_last_flops_main = g_last_flops_all = -1.0;        // (tiles multiplied: a device counter, mce_last_prune_stats)
            g_last_prune_geom[0] = p.nqblk; g_last_prune_geom[1] = (double)p.nchunk; g_last_prune_geom[2] = p.CT;
            snprintf(g_last_kernel, sizeof(g_last_kernel), "%s pruned grid=%d block=64 lds=%zu qt=%d ct=%d chunks=%lld", p.vh->name,
                     p.nqblk * mce::kHWaves, mce::f16_prune_lds_bytes(p.KST, d, p.KCAP), p.QT, p.CT, (long long)p.nchunk);
            return MCE_OK;
        }
        // seed phase: the host picks the group size; MCE_F16_SEED_ROWS / MCE_F16_SEED_SHARE override (tests, tuning)
        // (the kernel balances the splits to within one chunk: size the seed phase for the smallest)
        auto seed_cfg = [&](int ksel) { return sweep_seed_cfg(p.nchunk / p.rsplit, p.CT, ksel + a.self_exclude); };
        if (use_sym) {
            char* const sw = ws + p.off_sym;
            a.rsplit = 1;
            a.rperm = reinterpret_cast<const int*>(sw + p.sl.perm);
            a.sym.thr = reinterpret_cast<unsigned long long*>(sw + p.sl.thr);
            a.sym.rrow = reinterpret_cast<unsigned*>(sw + p.sl.rrow);
            a.sym.rtile = reinterpret_cast<float*>(sw + p.sl.rtile);
            a.sym.slots = reinterpret_cast<unsigned long long*>(sw + p.sl.slots);
            a.sym.bucket_cnt = reinterpret_cast<int*>(sw + p.sl.bucket_cnt);
            a.sym.bucket_flag = reinterpret_cast<int*>(sw + p.sl.bucket_flag);
            a.sym.bucket = reinterpret_cast<mce::SymEntry*>(sw + p.sl.bucket);
            a.sym.cap = p.sl.cap;
            a.sym.done = reinterpret_cast<int*>(sw + p.sl.done);
            // panel = the packed rows one L2 (4 MB per XCD) serves to the units running at the same time; MCE_SYM_PANEL: chunks (tuning)
            const Tuning tun = read_tuning();
            a.sym.panel = tun.sym_panel > 0 ? tun.sym_panel : kSymPanelChunks[p.KST];
            // prepass: every row's bound before any block runs (the seed phase as its own launch)
            // about 32 k rows (one k-step: 64 k), at most half of the chunks (tools/_tmp-style scans, fused call, share 8 -> 2:
            // 49 k x 27 1.51 -> 1.32 ms, 98 k 2.18 -> 2.03, 131 k 2.58 -> 2.47, from 197 k rows the same; 393 k x 15 7.04 -> 6.88)
            const int seed_rows = tun.sym_seed_rows > 0 ? tun.sym_seed_rows : (p.KST == 1 ? 65536 : 32768);
            const int seed_share = tun.sym_seed_share;
            a.seed_cfg = mce::f16_seed_cfg(p.nchunk, p.CT, K + a.self_exclude, seed_rows, seed_share, MCE_H_SEED_TG);
            // tiny sets (forced mode): smaller groups, so that half of the chunks still hold twice the K groups a bound needs --
            // without any bound every pair would go through the row side (20 k x 27: 10.8 ms instead of 0.8)
            for (int tg = MCE_H_SEED_TG / 2; a.seed_cfg == 0 && tg >= 1; tg /= 2)
                a.seed_cfg = mce::f16_seed_cfg(p.nchunk, p.CT, K + a.self_exclude, seed_rows, seed_share, tg);
            if (a.seed_cfg) {
                a.seed_cfg |= ((tun.sym_seed_mode >= 0 ? tun.sym_seed_mode : kSymSeedMode[p.KST]) & 3) << 28;
            }
            // One rank's share of a multi-GPU partition: the contiguous range of sorted blocks [qb_lo, qb_hi).  Their tiles
            // carry the row-side gate; everybody else's rows are swept column side only (sym_types.hpp, PanelGeom) -- no
            // exchange between the ranks, each ends with complete lists for its own rows.  Only they need a prepass bound.
            const int qb_lo = p.nparts > 1 ? (int)((int64_t)p.nqblk * p.part / p.nparts) : 0;
            const int qb_hi = p.nparts > 1 ? (int)((int64_t)p.nqblk * (p.part + 1) / p.nparts) : p.nqblk;
            p.sym_qb_lo = qb_lo;
            p.sym_qb_hi = qb_hi;
            a.qblk0 = qb_lo;
            a.nqblk_run = qb_hi - qb_lo;
            if (qb_hi > qb_lo) MCE_HIP(p.vh->launch_sym_pre(a, st));
            a.qblk0 = 0;
            a.nqblk_run = 0;
            const int seed_used = a.seed_cfg;
            a.seed_cfg = 0;
            int rc = prof_begin();             // (the bracket of mce_last_kernel_ms(): the dominant kernel, as for the other searches)
            if (rc != MCE_OK) return rc;
            const bool panel_kernel = !tun.sym_kernel_f16 || p.nparts > 1;
            if (panel_kernel) {
                mce::PanelArgs pa;
                pa.Yh = yh; pa.Xh = xh; pa.qinfo = qinfo; pa.params = params; pa.X = sX; pa.Y = sY; pa.rperm = a.rperm;
                pa.part_d = pd; pa.part_i = pi; pa.nq = nq; pa.nr = nr; pa.nq_pad = p.nq_pad; pa.self_offset = 0;
                pa.D = d; pa.ksel = K; pa.self_exclude = a.self_exclude; pa.spin_limit = tun.spin_limit;
                pa.sym = a.sym;
                pa.debug = tun.panel_debug;
                pa.geom.qb_lo = qb_lo; pa.geom.qb_hi = qb_hi; pa.geom.tpb = mce::kHWaves * mce::kHQT; pa.geom.ct = p.CT;
                pa.geom.tpp = a.sym.panel * p.CT; pa.geom.sym_on = 1;
                pa.geom.ntiles = (int)((nr + 31) / 32) + (int)(((nr + 31) / 32) & 1);
                MCE_HIP(p.vh->launch_panel(pa, st));
            } else {
                MCE_HIP(p.vh->launch_sym(a, st));
            }
            rc = prof_end();
            if (rc != MCE_OK) return rc;
            MCE_HIP(p.vh->launch_sym_repair(a, st));      // blocks whose bucket overflowed (normally none: every workgroup exits at once)
            {
                const dim3 g((unsigned)std::max(1, qb_hi - qb_lo)), b(mce::kSymMergeThreads);
                static_assert(mce::kSymMergeThreads == mce::f16_qpb(4), "one merge block per query block");
                switch (p.KCAP) {
                    case 4: hipLaunchKernelGGL(mce::sym_merge_kernel<4>, g, b, 0, st, pd, pi, p.nq_pad, a.sym.bucket_cnt, a.sym.bucket_flag, a.sym.bucket, a.sym.cap, qb_lo); break;
                    case 8: hipLaunchKernelGGL(mce::sym_merge_kernel<8>, g, b, 0, st, pd, pi, p.nq_pad, a.sym.bucket_cnt, a.sym.bucket_flag, a.sym.bucket, a.sym.cap, qb_lo); break;
                    case 12: hipLaunchKernelGGL(mce::sym_merge_kernel<12>, g, b, 0, st, pd, pi, p.nq_pad, a.sym.bucket_cnt, a.sym.bucket_flag, a.sym.bucket, a.sym.cap, qb_lo); break;
                    default: hipLaunchKernelGGL(mce::sym_merge_kernel<16>, g, b, 0, st, pd, pi, p.nq_pad, a.sym.bucket_cnt, a.sym.bucket_flag, a.sym.bucket, a.sym.cap, qb_lo); break;
                }
                MCE_HIP(hipGetLastError());
            }
            p.sym_active = true;
            p.L = 1;
            int sym_units = mce::sym_unit_count(p.nqblk, mce::kHWaves * mce::kHQT, a.sym.panel * p.CT, (int)((nr + 31) / 32) + (int)(((nr + 31) / 32) & 1));
            if (panel_kernel) {
                mce::PanelGeom g;
                g.qb_lo = qb_lo; g.qb_hi = qb_hi; g.tpb = mce::kHWaves * mce::kHQT; g.ct = p.CT; g.tpp = a.sym.panel * p.CT; g.sym_on = 1;
                g.ntiles = (int)((nr + 31) / 32) + (int)(((nr + 31) / 32) & 1);
                sym_units = mce::panel_unit_count(g);
                // executed MFMA flops: every unit's tiles x 16 query tiles x (32 x 32 x 16 KST) multiply-adds
                double tiles = 0.0;
                for (int u = 0; u < sym_units; ++u) {
                    int pp, aa, lo, hi;
                    mce::panel_unit_decode(u, g, pp, aa);
                    mce::panel_unit_tiles(pp, aa, g, lo, hi);
                    tiles += hi - lo;
                }
                g_last_flops_main = tiles * 16.0 * 1024.0 * 32.0 * p.KST;
            } else {
                const double nb = p.nqblk, tpb = mce::kHWaves * mce::kHQT, T = (double)((nr + 31) / 32);
                double tiles = 0.0;
                for (int b = 0; b < p.nqblk; ++b) tiles += std::min(tpb * (b + 1), T);
                (void)nb;
                g_last_flops_main = tiles * 16.0 * 1024.0 * 32.0 * p.KST;
            }
            g_last_flops_all = g_last_flops_main + (double)(seed_used & 0xffff) * p.CT * (double)(qb_hi - qb_lo) * 16.0 * 1024.0 * 32.0 * p.KST;
            snprintf(g_last_kernel, sizeof(g_last_kernel), "%s symmetric%s grid=%d block=%d lds=%zu qt=%d ct=%d panel=%d seed=%dx%d/%d bucket=%d", p.vh->name, panel_kernel ? " panel-kernel" : "",
                     sym_units,
                     mce::kHThreads, panel_kernel ? p.vh->lds_bytes_panel : p.vh->lds_bytes_sym, p.QT, p.CT, a.sym.panel, seed_used & 0xffff, (seed_used >> 16) & 0xfff, (seed_used >> 28) & 3, p.sl.cap);
            return MCE_OK;
        }
        int rc = prof_begin();
        if (rc != MCE_OK) return rc;
        if (p.twopass) {
            // lists [2*rsplit][16][nq_pad]: pass 1 fills splits 0..rsplit-1 with each split's 16 nearest, pass 2 the
            // next K - 16 beyond them into rsplit..2*rsplit-1; the merge takes the K best of all
            a.ksel = 16;
            a.seed_cfg = seed_cfg(16);
            MCE_HIP(p.vh->launch(a, st));
            a.lo_d = pd;
            a.lo_i = pi;
            a.part_d = pd + (size_t)p.rsplit * p.KCAP * (size_t)p.nq_pad;
            a.part_i = pi + (size_t)p.rsplit * p.KCAP * (size_t)p.nq_pad;
            a.ksel = K - 16;
            a.seed_cfg = 0;
            MCE_HIP(p.vh->launch_lower(a, st));
        } else {
            a.seed_cfg = seed_cfg(K);
            MCE_HIP(p.vh->launch(a, st));
        }
        rc = prof_end();
        if (rc != MCE_OK) return rc;
        const int seed_first = p.twopass ? seed_cfg(16) : a.seed_cfg;
        g_last_flops_main = (double)p.nqblk * ((double)p.nchunk + (double)(seed_first & 0xffff) * p.rsplit) * p.CT * 16.0 * 1024.0 * 32.0 * p.KST +
                            (p.twopass ? (double)p.nqblk * (double)p.nchunk * p.CT * 16.0 * 1024.0 * 32.0 * p.KST : 0.0);
        g_last_flops_all = g_last_flops_main;
        char seed_txt[48] = "";
        if (seed_first) snprintf(seed_txt, sizeof(seed_txt), " seed=%dx%d", seed_first & 0xffff, seed_first >> 16);   // chunks x tiles per group
        snprintf(g_last_kernel, sizeof(g_last_kernel), "%s grid=%d block=%d lds=%zu qt=%d ct=%d rsplit=%d%s%s", p.vh->name,
                 p.nqblk * p.rsplit, mce::kHThreads, p.vh->lds_bytes, p.QT, p.CT, p.rsplit, p.twopass ? " two passes" : "", seed_txt);
        return MCE_OK;
    }
    // ---- fp64 MFMA sweep ----------------------------------------------------------
    double* yf = reinterpret_cast<double*>(ws + p.off_yf);
    hipLaunchKernelGGL(mce::pack_refs_kernel, dim3((unsigned)((p.nrow_pad + threads - 1) / threads)), dim3(threads), 0, st, dY, nr,
                       (int)d, p.KS, p.nrow_pad, center, yf);
    MCE_HIP(hipGetLastError());
    mce::KnnArgs a;
    a.Yf = yf;
    a.nchunk_total = p.nchunk;
    a.rsplit = p.rsplit;
    a.X = dX;
    a.center = center;
    a.nq = nq;
    a.D = d;
    a.nq_pad = p.nq_pad;
    a.nqblk = p.nqblk;
    a.self_exclude = (self_mode == MCE_SELF_EXCLUDE) ? 1 : 0;
    a.self_offset = self_offset;
    a.ksel = K;
    a.part_d = pd;
    a.part_i = pi;
    int rc = prof_begin();
    if (rc != MCE_OK) return rc;
    MCE_HIP(p.v->launch(a, st));
    rc = prof_end();
    if (rc != MCE_OK) return rc;
    snprintf(g_last_kernel, sizeof(g_last_kernel), "%s grid=%d block=%d lds=%zu qt=%d ct=%d rsplit=%d", p.v->name,
             p.nqblk * p.rsplit, mce::kThreads, p.v->lds_bytes, p.QT, p.CT, p.rsplit);
    g_last_flops_main = g_last_flops_all = (double)p.nq_pad * (double)p.nrow_pad * 2.0 * 4.0 * p.KS;
    return MCE_OK;
}

// merge (+ optional distance output, + optional fused reduction) of the per-split lists
int launch_merge(const Plan& p, bool write_dist, bool fuse, const double* dX, const double* dY, int64_t nq, int32_t d, int K,
                 int self_mode, int64_t self_offset, double* d_dist, int64_t* d_idx, int k0, int kmax,
                 const double* d_w, const double* d_fs, double* partial, char* ws, hipStream_t st)
{
    const bool same_set = (dX == dY && nq == p.pl_nr);
    // a part of a pruned search: only the list columns of its query blocks were filled; the merge threads
    // enumerate those columns compactly
    const int qpb = p.vh ? mce::f16_qpb(p.KCAP) : 1;
    int64_t ncol = nq;
    int64_t col0 = 0, col1 = INT64_MAX;
    if (p.sym_active && p.nparts > 1) {         // one rank's blocks of a symmetric partition: a contiguous range of list columns
        col0 = (int64_t)p.sym_qb_lo * qpb;
        col1 = std::min<int64_t>((int64_t)p.sym_qb_hi * qpb, nq);
        ncol = std::max<int64_t>(col1 - col0, 0);
    } else if (p.nparts > 1) ncol = (int64_t)((p.nqblk - p.part + p.nparts - 1) / p.nparts) * qpb;
    const unsigned blocks = (unsigned)std::max<int64_t>((ncol + mce::kRedThreads - 1) / mce::kRedThreads, 1);
    const double* pd = reinterpret_cast<const double*>(ws + p.off_pd);
    const int* pi = reinterpret_cast<const int*>(ws + p.off_pi);
    const bool refine = p.vh == nullptr && !p.generic;   // fp64 sweep keys are GEMM-form: refine; the others are exact
    const double lnc = fuse ? ln_unit_ball(d) : 0.0;
    // pruned search: list column q is the q-th query in k-d order; its caller row is qperm[q]
    const int* qperm = nullptr;
    if (p.prune) qperm = reinterpret_cast<const int*>(ws + p.off_prune + (same_set ? p.pl.perm_r : p.pl.perm_q));
    if (p.sym_active) qperm = reinterpret_cast<const int*>(ws + p.off_sym + p.sl.perm);     // list column = sorted position
    const int* border = p.prune ? reinterpret_cast<const int*>(ws + p.off_prune + p.pl.border) : nullptr;
#define MCE_MERGE(W, F, R)                                                                                          \
    hipLaunchKernelGGL((mce::merge_lists_kernel<W, F, R>), dim3(blocks), dim3(mce::kRedThreads), 0, st, pd, pi, p.L,  \
                       p.KCAP, nq, p.nq_pad, dX, dY, (int)d, K, self_mode, self_offset, d_dist, d_idx, K, k0, kmax, \
                       d_w, d_fs, lnc, partial, qperm, p.part, p.nparts, qpb, border, p.nqblk, col0, col1)
    if (write_dist && !fuse) { if (refine) MCE_MERGE(true, false, true); else MCE_MERGE(true, false, false); }
    else if (write_dist && fuse) { if (refine) MCE_MERGE(true, true, true); else MCE_MERGE(true, true, false); }
    else { if (refine) MCE_MERGE(false, true, true); else MCE_MERGE(false, true, false); }
#undef MCE_MERGE
    MCE_HIP(hipGetLastError());
    return MCE_OK;
}

}  // namespace

extern "C" {

int mce_abi_version(void) { return MCE_ABI_VERSION; }

const char* mce_last_error(void) { return g_err; }

const char* mce_last_kernel(void) { return g_last_kernel; }

void mce_release_device_memory(void)
{
    g_pool.release_idle();
    g_pinned.release();
}

int mce_set_search_mode(int mode)
{
    if (mode < 0 || mode > 2) return fail(MCE_ERR_INVALID, "search mode must be 0 (auto), 1 (fp64 sweep) or 2 (fp16 filter + fp64 refine)");
    g_mode.store(mode);
    return MCE_OK;
}

int mce_get_search_mode(void) { return g_mode.load(); }

int mce_set_prune_mode(int mode)
{
    if (mode < 0 || mode > 2) return fail(MCE_ERR_INVALID, "prune mode must be 0 (auto), 1 (never) or 2 (whenever the shape allows it)");
    g_prune_mode.store(mode);
    return MCE_OK;
}

int mce_get_prune_mode(void) { return g_prune_mode.load(); }

int mce_set_sym_mode(int mode)
{
    if (mode < 0 || mode > 2) return fail(MCE_ERR_INVALID, "symmetric-sweep mode must be 0 (auto), 1 (never) or 2 (whenever the shape allows it)");
    g_sym_mode.store(mode);
    return MCE_OK;
}

int mce_get_sym_mode(void) { return sym_mode(); }

int mce_options_push(const mce_options* opt)
{
    if (!opt) return fail(MCE_ERR_INVALID, "null options");
    if (opt->size < (int32_t)(4 * sizeof(int32_t))) return fail(MCE_ERR_INVALID, "mce_options.size=%d: set it to sizeof(mce_options)", opt->size);
    if (opt->search_mode < -1 || opt->search_mode > 2 || opt->prune_mode < -1 || opt->prune_mode > 2 || opt->sym_mode < -1 || opt->sym_mode > 2)
        return fail(MCE_ERR_INVALID, "mce_options: modes are -1 (default), 0, 1 or 2");
    t_opt_stack.push_back(t_opt);
    if (opt->search_mode >= 0) t_opt.search = opt->search_mode;
    if (opt->prune_mode >= 0) t_opt.prune = opt->prune_mode;
    if (opt->sym_mode >= 0) t_opt.sym = opt->sym_mode;
    if (opt->same_set >= 0) t_opt.same_set = opt->same_set ? 1 : 0;
    return MCE_OK;
}

int mce_options_pop(void)
{
    if (t_opt_stack.empty()) return fail(MCE_ERR_INVALID, "mce_options_pop without a push on this thread");
    t_opt = t_opt_stack.back();
    t_opt_stack.pop_back();
    return MCE_OK;
}

int mce_last_prune_stats(double* chunk_fraction, double* tile_fraction)
{
    if (!chunk_fraction || !tile_fraction) return fail(MCE_ERR_INVALID, "null pointer argument");
    if (!g_last_params) return fail(MCE_ERR_INVALID, "no pruned search has run on this thread");
    double hp[mce::HP_COUNT];
    MCE_HIP(hipDeviceSynchronize());
    MCE_HIP(hipMemcpy(hp, g_last_params, sizeof(hp), hipMemcpyDeviceToHost));
    const double pairs = g_last_prune_geom[0] * g_last_prune_geom[1];
    *chunk_fraction = hp[mce::HP_STAT_CHUNKS] / pairs;
    *tile_fraction = hp[mce::HP_STAT_TILES] / (pairs * mce::kHWaves * g_last_prune_geom[2]);
    if (read_tuning().prune_prof) {
        const double nw = g_last_prune_geom[0] * mce::kHWaves;
        fprintf(stderr, "[prune prof] per wave (cycles@100MHz): walk %.0f stage %.0f mul %.0f drain %.0f total %.0f  candidates drained %.0f | tiles with enqueue %.0f (process %.0f cyc each), without: process total %.0f\n", hp[8] / nw, hp[9] / nw, hp[10] / nw, hp[11] / nw, hp[12] / nw, hp[13] / nw, hp[14] / nw, hp[15] / std::max(1.0, hp[14]), hp[7] / nw);
    }
    return MCE_OK;
}

void mce_set_profiling(int on)
{
    g_prof_on = on ? 1 : 0;
    if (on) { g_ev_used = 0; g_ev_calls = 0; g_evs_used = 0; }
}

int mce_last_search_stats(double* out, int32_t n)
{
    if (!out || n < 4) return fail(MCE_ERR_INVALID, "mce_last_search_stats: out[4] expected");
    out[0] = g_last_flops_main;
    out[1] = g_last_flops_all;
    out[2] = -1.0;
    out[3] = mce_last_kernel_ms();
    if (g_evs_used > 0) {
        double sum = 0.0;
        for (size_t i = 0; i < g_evs_used; ++i) {
            float ms = 0.0f;
            if (hipEventSynchronize(g_evs_pool[i].second) != hipSuccess ||
                hipEventElapsedTime(&ms, g_evs_pool[i].first, g_evs_pool[i].second) != hipSuccess) return fail(MCE_ERR_HIP, "event timing failed");
            sum += ms;
        }
        out[2] = sum / (double)(g_ev_calls ? g_ev_calls : g_evs_used);
    }
    return MCE_OK;
}

double mce_last_kernel_ms(void)
{
    if (g_ev_used == 0) return -1.0;
    double sum = 0.0;
    for (size_t i = 0; i < g_ev_used; ++i) {
        if (hipEventSynchronize(g_ev_pool[i].second) != hipSuccess) return -1.0;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, g_ev_pool[i].first, g_ev_pool[i].second) != hipSuccess) return -1.0;
        sum += ms;
    }
    return sum / (double)(g_ev_calls ? g_ev_calls : g_ev_used);      // per search (a split search times two launches)
}

int mce_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- trailing partial round ------------------------------------------------------------------------
// The exhaustive kernels run one 512-query workgroup per CU, all of the same length, so a search is a sequence of
// rounds of 256 workgroups and the last one is as long as the others however few it holds: 782 blocks (N = 400 k)
// take four rounds for 3.05 rounds of work.  When the last round would fill at most half the chip, the queries are
// searched as two ranges instead: the full rounds as they are, and the remainder as its own search, whose plan
// then splits the reference set over the idle CUs (measured, sequential launches: 400 k x 6 13.1 -> 11.0 ms,
// 400 k x 27 17.0 -> 13.4 ms, 300 k x 15 6.7 -> 5.9 ms, 700 k x 10 22.4 -> 20.6 ms; C3's last round is 63 % full
// and is left alone: 67.6 vs 67.3 ms).  Rows, lists and partial sums of the two ranges are disjoint and laid
// out exactly as in one search, so every result is bit-identical.  Returns the rows of the first range, or 0.
// mce_last_kernel() of a split search: the first range's launch + the geometry of the second
void note_split(const char* first)
{
    char tail[sizeof(g_last_kernel)];
    snprintf(tail, sizeof(tail), "%s", g_last_kernel);
    const char* g = strstr(tail, " grid=");
    snprintf(g_last_kernel, sizeof(g_last_kernel), "%.150s + tail%.80s", first, g ? g : "");
}

int64_t tail_split_rows(const Plan& p, int64_t nq, int64_t nr, int32_t d, int32_t K, int32_t self_mode, size_t ws_avail, bool same_set)
{
    if (g_split_depth > 0 || !p.vh || p.prune || p.generic || p.rsplit != 1 || p.nqblk <= kAssumedCUs) return 0;
    if (p.sym && same_set) return 0;          // symmetric sweep: one launch over the whole set
    if (!read_tuning().tail_split) return 0;
    const int tail = p.nqblk % kAssumedCUs;
    if (tail == 0 || 2 * tail > kAssumedCUs) return 0;
    const int64_t nq_main = (int64_t)(p.nqblk - tail) * mce::f16_qpb(p.KCAP);
    static_assert(mce::f16_qpb(4) % mce::kRedThreads == 0, "the merge blocks of the two ranges must tile like one search's");
    Plan pm, pt;
    if (make_plan(nq_main, nr, d, K, self_mode, pm) != MCE_OK || make_plan(nq - nq_main, nr, d, K, self_mode, pt) != MCE_OK) return 0;
    if (!pm.vh || !pt.vh || pm.prune || pt.prune || pm.total > ws_avail || pt.total > ws_avail) return 0;
    if (pm.cost + pt.cost > 0.97 * p.cost) return 0;
    return nq_main;
}

size_t mce_knn_workspace_bytes(int64_t nq, int64_t nr, int32_t d, int32_t K)
{
    Plan p;
    if (make_plan(nq, nr, d, K, MCE_SELF_NONE, p) != MCE_OK) return 0;
    return p.total;
}

size_t mce_dotp_workspace_bytes(int64_t nq, int32_t kmax) { return dotp_ws_bytes(nq, kmax); }

int mce_knn_f64_dev(const double* dX, int64_t nq, const double* dY, int64_t nr, int32_t d, int32_t K,
                    int32_t self_mode, int64_t self_offset, double* d_dist, int64_t* d_idx, void* ws,
                    size_t ws_bytes, void* stream)
{
    if (!dX || !dY || !d_dist || !ws) return fail(MCE_ERR_INVALID, "null pointer argument");
    Plan p;
    SameSetHint hint(dX == dY && nq == nr && self_offset == 0);
    int rc = make_plan(nq, nr, d, K, self_mode, p);
    if (rc != MCE_OK) return rc;
    if (ws_bytes < p.total && p.sym) {          // a workspace sized with same_set = 0: the exhaustive plan fits it
        g_same_set_hint = 0;
        rc = make_plan(nq, nr, d, K, self_mode, p);
        if (rc != MCE_OK) return rc;
    }
    if (ws_bytes < p.total) return fail(MCE_ERR_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes, p.total);
    if (nq == 0) return MCE_OK;
    if (const int64_t nm = tail_split_rows(p, nq, nr, d, K, self_mode, ws_bytes, dX == dY && self_offset == 0)) {
        ++g_split_depth;
        rc = mce_knn_f64_dev(dX, nm, dY, nr, d, K, self_mode, self_offset, d_dist, d_idx, ws, ws_bytes, stream);
        char first[sizeof(g_last_kernel)];
        snprintf(first, sizeof(first), "%s", g_last_kernel);
        if (rc == MCE_OK) {
            g_in_tail = true;
            rc = mce_knn_f64_dev(dX + nm * (int64_t)d, nq - nm, dY, nr, d, K, self_mode, self_offset + nm, d_dist + nm * (int64_t)K,
                                 d_idx ? d_idx + nm * (int64_t)K : nullptr, ws, ws_bytes, stream);
            g_in_tail = false;
        }
        --g_split_depth;
        if (rc == MCE_OK) note_split(first);
        return rc;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = run_search(p, dX, nq, dY, nr, d, K, self_mode, self_offset, static_cast<char*>(ws), st);
    if (rc != MCE_OK) return rc;
    if (p.generic) {
        hipLaunchKernelGGL(mce::generic_finalize_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<const double*>(static_cast<char*>(ws) + p.off_pd),
                           reinterpret_cast<const int*>(static_cast<char*>(ws) + p.off_pi), nq, p.nq_pad, (int)K, (int)self_mode,
                           self_offset, d_dist, d_idx);
        MCE_HIP(hipGetLastError());
        return MCE_OK;
    }
    rc = launch_merge(p, true, false, dX, dY, nq, d, K, self_mode, self_offset, d_dist, d_idx, 0, 0, nullptr, nullptr, nullptr,
                      static_cast<char*>(ws), st);
    if (rc != MCE_OK) return rc;
    return MCE_OK;
}

int mce_dotp_f64_dev(const double* d_dist, int64_t nq, int32_t ld, int32_t k0, int32_t kmax, int32_t d,
                     const double* d_w, const double* d_fs, double* d_dotp, void* ws, size_t ws_bytes,
                     void* stream)
{
    if (!d_dist || !d_w || !d_fs || !d_dotp || !ws) return fail(MCE_ERR_INVALID, "null pointer argument");
    if (nq < 1 || d < 1 || k0 < 0 || kmax < 1 || k0 > kmax || kmax > ld)
        return fail(MCE_ERR_INVALID, "invalid sizes nq=%lld ld=%d k0=%d kmax=%d d=%d", (long long)nq, ld, k0, kmax, d);
    if (ws_bytes < dotp_ws_bytes(nq, kmax)) return fail(MCE_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned blocks = (unsigned)((nq + mce::kRedThreads - 1) / mce::kRedThreads);
    double* partial = static_cast<double*>(ws);
    hipLaunchKernelGGL(mce::dotp_partial_kernel, dim3(blocks), dim3(mce::kRedThreads), 0, st, d_dist, nq, (int)ld,
                       (int)k0, (int)kmax, (int)d, ln_unit_ball(d), d_w, d_fs, partial);
    MCE_HIP(hipGetLastError());
    hipLaunchKernelGGL(mce::dotp_final_kernel, dim3((unsigned)kmax), dim3(mce::kRedThreads), 0, st, partial,
                       (int64_t)blocks, (int)k0, (int)kmax, d_dotp);
    MCE_HIP(hipGetLastError());
    return MCE_OK;
}

int mce_knn_dotp_f64_dev(const double* dX, int64_t nq, const double* dY, int64_t nr, int32_t d, int32_t kmax,
                         int32_t k0, int64_t self_offset, const double* d_w, const double* d_fs,
                         double* d_dotp, double* d_dist_out, void* ws, size_t ws_bytes, void* stream)
{
    if (!dX || !dY || !d_w || !d_fs || !d_dotp || !ws) return fail(MCE_ERR_INVALID, "null pointer argument");
    if (k0 != 0 && k0 != 1) return fail(MCE_ERR_INVALID, "k0 must be 0 (cross) or 1 (auto), got %d", k0);
    if (kmax <= k0) return fail(MCE_ERR_INVALID, "kmax=%d must exceed k0=%d", kmax, k0);
    if (nq < 1) return fail(MCE_ERR_INVALID, "nq must be >= 1");
    const int K = kmax - k0;
    const int self_mode = k0 == 1 ? MCE_SELF_EXCLUDE : MCE_SELF_NONE;
    Plan p;
    SameSetHint hint(dX == dY && nq == nr && self_offset == 0);
    int rc = make_plan(nq, nr, d, K, self_mode, p);
    if (rc != MCE_OK) return rc;
    if (ws_bytes < p.total + dotp_ws_bytes(nq, kmax) && p.sym) {          // a workspace sized with same_set = 0: the exhaustive plan fits it
        g_same_set_hint = 0;
        rc = make_plan(nq, nr, d, K, self_mode, p);
        if (rc != MCE_OK) return rc;
    }
    const size_t need = p.total + dotp_ws_bytes(nq, kmax);
    if (ws_bytes < need) return fail(MCE_ERR_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* wsc = static_cast<char*>(ws);
    double* partial = reinterpret_cast<double*>(wsc + p.total);
    const unsigned blocks = (unsigned)((nq + mce::kRedThreads - 1) / mce::kRedThreads);
    if (const int64_t nm = tail_split_rows(p, nq, nr, d, K, self_mode, p.total, dX == dY && self_offset == 0)) {
        // two ranges, one reduction: the partial sums of the second range follow those of the first in the
        // order one search would have produced them
        Plan pm, pt;
        if (make_plan(nm, nr, d, K, self_mode, pm) != MCE_OK || make_plan(nq - nm, nr, d, K, self_mode, pt) != MCE_OK)
            return fail(MCE_ERR_INVALID, "split plan");
        ++g_split_depth;
        rc = run_search(pm, dX, nm, dY, nr, d, K, self_mode, self_offset, wsc, st);
        if (rc == MCE_OK)
            rc = launch_merge(pm, d_dist_out != nullptr, true, dX, dY, nm, d, K, self_mode, self_offset, d_dist_out, nullptr, (int)k0,
                              (int)kmax, d_w, d_fs, partial, wsc, st);
        char first[sizeof(g_last_kernel)];
        snprintf(first, sizeof(first), "%s", g_last_kernel);
        if (rc == MCE_OK) {
            g_in_tail = true;
            rc = run_search(pt, dX + nm * (int64_t)d, nq - nm, dY, nr, d, K, self_mode, self_offset + nm, wsc, st);
            g_in_tail = false;
        }
        if (rc == MCE_OK)
            rc = launch_merge(pt, d_dist_out != nullptr, true, dX + nm * (int64_t)d, dY, nq - nm, d, K, self_mode, self_offset + nm,
                              d_dist_out ? d_dist_out + nm * (int64_t)K : nullptr, nullptr, (int)k0, (int)kmax, d_w + nm, d_fs + nm,
                              partial + (nm / mce::kRedThreads) * (int64_t)kmax, wsc, st);
        --g_split_depth;
        if (rc != MCE_OK) return rc;
        note_split(first);
        hipLaunchKernelGGL(mce::dotp_final_kernel, dim3((unsigned)kmax), dim3(mce::kRedThreads), 0, st, partial,
                           (int64_t)blocks, (int)k0, (int)kmax, d_dotp);
        MCE_HIP(hipGetLastError());
        return MCE_OK;
    }
    rc = run_search(p, dX, nq, dY, nr, d, K, self_mode, self_offset, wsc, st);
    if (rc != MCE_OK) return rc;
    if (p.generic) {
        // lists -> distance matrix [nq, K] (caller's buffer or workspace scratch) -> unfused reduction
        double* dd = d_dist_out ? d_dist_out : reinterpret_cast<double*>(wsc + p.off_center);
        hipLaunchKernelGGL(mce::generic_finalize_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<const double*>(wsc + p.off_pd), reinterpret_cast<const int*>(wsc + p.off_pi), nq,
                           p.nq_pad, K, self_mode, self_offset, dd, (int64_t*)nullptr);
        MCE_HIP(hipGetLastError());
        // dotp_partial reads column (k - k0) of dd for reference column k: shift the base pointer
        hipLaunchKernelGGL(mce::dotp_partial_kernel, dim3(blocks), dim3(mce::kRedThreads), 0, st, dd - k0, nq, K, (int)k0, (int)kmax,
                           (int)d, ln_unit_ball(d), d_w, d_fs, partial);
        MCE_HIP(hipGetLastError());
    } else {
        rc = launch_merge(p, d_dist_out != nullptr, true, dX, dY, nq, d, K, self_mode, self_offset, d_dist_out, nullptr, (int)k0,
                          (int)kmax, d_w, d_fs, partial, wsc, st);
        if (rc != MCE_OK) return rc;
    }
    hipLaunchKernelGGL(mce::dotp_final_kernel, dim3((unsigned)kmax), dim3(mce::kRedThreads), 0, st, partial,
                       (int64_t)blocks, (int)k0, (int)kmax, d_dotp);
    MCE_HIP(hipGetLastError());
    return MCE_OK;
}

int mce_knn_dotp_part_f64_dev(const double* dY, int64_t nr, int32_t d, int32_t kmax, int32_t part, int32_t nparts,
                              const double* d_w, const double* d_fs, double* d_dotp, void* ws, size_t ws_bytes, void* stream)
{
    if (!dY || !d_w || !d_fs || !d_dotp || !ws) return fail(MCE_ERR_INVALID, "null pointer argument");
    if (nparts < 1 || part < 0 || part >= nparts) return fail(MCE_ERR_INVALID, "part %d of %d", part, nparts);
    if (kmax <= 1) return fail(MCE_ERR_INVALID, "kmax=%d must exceed k0=1", kmax);
    const int K = kmax - 1;
    Plan p;
    int rc = make_plan(nr, nr, d, K, MCE_SELF_EXCLUDE, p);
    if (rc != MCE_OK) return rc;
    const size_t need = p.total + dotp_ws_bytes(nr, kmax);
    if (ws_bytes < need) return fail(MCE_ERR_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!p.prune && p.sym && nparts > 1 && nparts <= kSymPartitionMaxParts) {
        // Auto evidence of a set large enough for the symmetric sweep: every rank takes a contiguous range of the SORTED
        // blocks, symmetric within its range and column side only against everybody else's rows (no exchange).
        // Per rank n^2/W (1 - 1/2W) tile products instead of the n^2/W of a query shard, at the symmetric kernel's cost per
        // product (1.08 x the exhaustive kernel's) + ~1 ms of sorting and prepass per rank: it wins up to four ranks
        // (C3, slowest rank, same box: 26.7 vs 32.5 ms at 2, 15.9 vs 16.4 at 4, 9.5 vs 8.4 at 8); beyond, query shards.
        p.part = part;
        p.nparts = nparts;
        char* wsc = static_cast<char*>(ws);
        SameSetHint hint(true);
        rc = run_search(p, dY, nr, dY, nr, d, K, MCE_SELF_EXCLUDE, 0, wsc, st);
        if (rc != MCE_OK) return rc;
        if (p.sym_active) {
            if (p.sym_qb_hi <= p.sym_qb_lo) { MCE_HIP(mce::zero_async(d_dotp, (size_t)kmax * sizeof(double), st)); return MCE_OK; }
            double* partial = reinterpret_cast<double*>(wsc + p.total);
            rc = launch_merge(p, false, true, dY, dY, nr, d, K, MCE_SELF_EXCLUDE, 0, nullptr, nullptr, 1, (int)kmax, d_w, d_fs, partial, wsc, st);
            if (rc != MCE_OK) return rc;
            const int64_t qpb = mce::f16_qpb(p.KCAP);
            const int64_t ncol = std::min<int64_t>((int64_t)p.sym_qb_hi * qpb, nr) - (int64_t)p.sym_qb_lo * qpb;
            const unsigned blocks = (unsigned)((ncol + mce::kRedThreads - 1) / mce::kRedThreads);
            hipLaunchKernelGGL(mce::dotp_final_kernel, dim3((unsigned)kmax), dim3(mce::kRedThreads), 0, st, partial, (int64_t)blocks, 1, (int)kmax, d_dotp);
            MCE_HIP(hipGetLastError());
            return MCE_OK;
        }
        return fail(MCE_ERR_INVALID, "symmetric partition: the sweep did not run");
    }
    if (!p.prune) {
        // contiguous rows: exactly the query shard of SURVEY.md section 8e
        const int64_t lo = nr * part / nparts, hi = nr * (int64_t)(part + 1) / nparts;
        if (hi == lo) { MCE_HIP(mce::zero_async(d_dotp, (size_t)kmax * sizeof(double), st)); return MCE_OK; }
        return mce_knn_dotp_f64_dev(dY + lo * (int64_t)d, hi - lo, dY, nr, d, kmax, 1, lo, d_w + lo, d_fs + lo, d_dotp, nullptr, ws, ws_bytes, stream);
    }
    // pruned walk: every nparts-th query block of the k-d order -- spatially compact work units, one shared
    // ordering, and statistically equal shares (contiguous ranges of the order differ 2x in cost)
    if (part >= p.nqblk) { MCE_HIP(mce::zero_async(d_dotp, (size_t)kmax * sizeof(double), st)); return MCE_OK; }
    p.part = part;
    p.nparts = nparts;
    char* wsc = static_cast<char*>(ws);
    rc = run_search(p, dY, nr, dY, nr, d, K, MCE_SELF_EXCLUDE, 0, wsc, st);
    if (rc != MCE_OK) return rc;
    double* partial = reinterpret_cast<double*>(wsc + p.total);
    rc = launch_merge(p, false, true, dY, dY, nr, d, K, MCE_SELF_EXCLUDE, 0, nullptr, nullptr, 1, (int)kmax, d_w, d_fs, partial, wsc, st);
    if (rc != MCE_OK) return rc;
    const int64_t ncol = (int64_t)((p.nqblk - part + nparts - 1) / nparts) * mce::f16_qpb(p.KCAP);
    const unsigned blocks = (unsigned)((ncol + mce::kRedThreads - 1) / mce::kRedThreads);
    hipLaunchKernelGGL(mce::dotp_final_kernel, dim3((unsigned)kmax), dim3(mce::kRedThreads), 0, st, partial, (int64_t)blocks, 1, (int)kmax, d_dotp);
    MCE_HIP(hipGetLastError());
    return MCE_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// host-pointer wrappers
// ---------------------------------------------------------------------------
namespace {

int select_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return fail(MCE_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(MCE_ERR_INVALID, "device %d out of range (have %d)", device, n);
    MCE_HIP(hipSetDevice(device));
    return MCE_OK;
}

}  // namespace

extern "C" {

int mce_knn_f64(const double* X, int64_t nq, const double* Y, int64_t nr, int32_t d, int32_t K,
                int32_t self_mode, int64_t self_offset, double* dist, int64_t* idx, int32_t device)
{
    if (!X || !Y || !dist) return fail(MCE_ERR_INVALID, "null pointer argument");
    SameSetHint hint(X == Y && nq == nr && self_offset == 0);
    Plan p;
    int rc = make_plan(nq, nr, d, K, self_mode, p);
    if (rc != MCE_OK) return rc;
    if (nq == 0) return MCE_OK;
    rc = select_device(device);
    if (rc != MCE_OK) return rc;
    // queries that ARE rows of the caller's reference buffer (kneighbors(Y) after fit(Y), a shard of it): one
    // upload, and the search sees one set (shared k-d order in the pruned walk)
    const bool inside = X >= Y && X + (size_t)nq * d <= Y + (size_t)nr * d && (X - Y) % d == 0;
    DevBuf dX, dY, dD, dI, ws;
    if (!inside) MCE_HIP(dX.alloc((size_t)nq * d * sizeof(double)));
    MCE_HIP(dY.alloc((size_t)nr * d * sizeof(double)));
    MCE_HIP(dD.alloc((size_t)nq * K * sizeof(double)));
    if (idx) MCE_HIP(dI.alloc((size_t)nq * K * sizeof(int64_t)));
    MCE_HIP(ws.alloc(p.total));
    if (!inside) MCE_HIP(hipMemcpy(dX.p, X, (size_t)nq * d * sizeof(double), hipMemcpyHostToDevice));
    MCE_HIP(hipMemcpy(dY.p, Y, (size_t)nr * d * sizeof(double), hipMemcpyHostToDevice));
    const double* dXp = inside ? dY.as<double>() + (X - Y) : dX.as<double>();
    rc = mce_knn_f64_dev(dXp, nq, dY.as<double>(), nr, d, K, self_mode, self_offset, dD.as<double>(),
                         idx ? dI.as<int64_t>() : nullptr, ws.p, p.total, nullptr);
    if (rc != MCE_OK) return rc;
    MCE_HIP(hipDeviceSynchronize());
    MCE_HIP(hipMemcpy(dist, dD.p, (size_t)nq * K * sizeof(double), hipMemcpyDeviceToHost));
    if (idx) MCE_HIP(hipMemcpy(idx, dI.p, (size_t)nq * K * sizeof(int64_t), hipMemcpyDeviceToHost));
    return MCE_OK;
}

int mce_dotp_f64(const double* dist, int64_t nq, int32_t ld, int32_t k0, int32_t kmax, int32_t d,
                 const double* w, const double* fs, double* dotp, int32_t device)
{
    if (!dist || !w || !fs || !dotp) return fail(MCE_ERR_INVALID, "null pointer argument");
    if (nq < 1 || d < 1 || k0 < 0 || kmax < 1 || k0 > kmax || kmax > ld)
        return fail(MCE_ERR_INVALID, "invalid sizes nq=%lld ld=%d k0=%d kmax=%d d=%d", (long long)nq, ld, k0, kmax, d);
    int rc = select_device(device);
    if (rc != MCE_OK) return rc;
    DevBuf dD, dW, dF, dO, ws;
    const size_t wsb = dotp_ws_bytes(nq, kmax);
    MCE_HIP(dD.alloc((size_t)nq * ld * sizeof(double)));
    MCE_HIP(dW.alloc((size_t)nq * sizeof(double)));
    MCE_HIP(dF.alloc((size_t)nq * sizeof(double)));
    MCE_HIP(dO.alloc((size_t)kmax * sizeof(double)));
    MCE_HIP(ws.alloc(wsb));
    MCE_HIP(hipMemcpy(dD.p, dist, (size_t)nq * ld * sizeof(double), hipMemcpyHostToDevice));
    MCE_HIP(hipMemcpy(dW.p, w, (size_t)nq * sizeof(double), hipMemcpyHostToDevice));
    MCE_HIP(hipMemcpy(dF.p, fs, (size_t)nq * sizeof(double), hipMemcpyHostToDevice));
    rc = mce_dotp_f64_dev(dD.as<double>(), nq, ld, k0, kmax, d, dW.as<double>(), dF.as<double>(), dO.as<double>(), ws.p, wsb, nullptr);
    if (rc != MCE_OK) return rc;
    MCE_HIP(hipDeviceSynchronize());
    MCE_HIP(hipMemcpy(dotp, dO.p, (size_t)kmax * sizeof(double), hipMemcpyDeviceToHost));
    return MCE_OK;
}

}  // extern "C"

namespace {

// cyclic Jacobi eigen-solver for a symmetric d x d matrix (row-major A, destroyed); eigenvalues in
// lam[d], eigenvectors in the COLUMNS of V (row-major [d][d]).  d <= 1024; converges to ~1e-15.
void jacobi_eig(std::vector<double>& A, int d, std::vector<double>& lam, std::vector<double>& V)
{
    V.assign((size_t)d * d, 0.0);
    for (int i = 0; i < d; ++i) V[(size_t)i * d + i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < d; ++i) {
            diag += A[(size_t)i * d + i] * A[(size_t)i * d + i];
            for (int j = i + 1; j < d; ++j) off += A[(size_t)i * d + j] * A[(size_t)i * d + j];
        }
        if (off <= 1e-32 * diag || off == 0.0) break;
        for (int p = 0; p < d - 1; ++p)
            for (int q = p + 1; q < d; ++q) {
                const double apq = A[(size_t)p * d + q];
                if (apq == 0.0) continue;
                const double app = A[(size_t)p * d + p], aqq = A[(size_t)q * d + q];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < d; ++k) {          // A <- A J   (columns p, q)
                    const double akp = A[(size_t)k * d + p], akq = A[(size_t)k * d + q];
                    A[(size_t)k * d + p] = c * akp - s * akq;
                    A[(size_t)k * d + q] = s * akp + c * akq;
                }
                for (int k = 0; k < d; ++k) {          // A <- J^T A (rows p, q)
                    const double apk = A[(size_t)p * d + k], aqk = A[(size_t)q * d + k];
                    A[(size_t)p * d + k] = c * apk - s * aqk;
                    A[(size_t)q * d + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < d; ++k) {          // V <- V J
                    const double vkp = V[(size_t)k * d + p], vkq = V[(size_t)k * d + q];
                    V[(size_t)k * d + p] = c * vkp - s * vkq;
                    V[(size_t)k * d + q] = s * vkp + c * vkq;
                }
            }
    }
    // canonical form: eigenvalues descending, each eigenvector's largest component positive.  Two
    // sets whitened with their OWN systems (covtype 'single' cross evidence) are then rotated
    // consistently whenever their covariances are close, whatever the sweep order did.
    std::vector<int> order(d);
    for (int i = 0; i < d; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return A[(size_t)a * d + a] > A[(size_t)b * d + b]; });
    lam.resize(d);
    std::vector<double> Vs((size_t)d * d);
    for (int c = 0; c < d; ++c) {
        const int src = order[c];
        lam[c] = A[(size_t)src * d + src];
        int big = 0;
        for (int k = 1; k < d; ++k)
            if (std::fabs(V[(size_t)k * d + src]) > std::fabs(V[(size_t)big * d + src])) big = k;
        const double sgn = V[(size_t)big * d + src] < 0.0 ? -1.0 : 1.0;
        for (int k = 0; k < d; ++k) Vs[(size_t)k * d + c] = sgn * V[(size_t)k * d + src];
    }
    V.swap(Vs);
}

// covariance (two-pass, unweighted, n-1) of the device matrix S[n, d] -> device cov[d*d]; enqueue only
int launch_covariance(const double* dS, int64_t n, int d, double* scratch_partial, double* d_mean3, double* d_cov, hipStream_t st)
{
    hipLaunchKernelGGL(mce::col_stats_partial_kernel, dim3(mce::kMeanBlocks), dim3(mce::kMeanThreads), 0, st, dS, n, d, scratch_partial);
    MCE_HIP(hipGetLastError());
    hipLaunchKernelGGL(mce::col_stats_final_kernel, dim3(1), dim3(64), 0, st, scratch_partial, n, d, d_mean3, (double*)nullptr);
    MCE_HIP(hipGetLastError());
    hipLaunchKernelGGL(mce::cov_partial_kernel, dim3(mce::kCovBlocks), dim3(mce::kCovThreads), (size_t)mce::kCovTileRows * d * sizeof(double), st,
                       dS, n, d, d_mean3, scratch_partial);
    MCE_HIP(hipGetLastError());
    hipLaunchKernelGGL(mce::cov_final_kernel, dim3(1), dim3(mce::kCovThreads), 0, st, scratch_partial, n, d, d_cov);
    MCE_HIP(hipGetLastError());
    return MCE_OK;
}

// one device's share of the fused path: queries [q_lo, q_hi)
int fused_on_device(int device, const double* X, int64_t q_lo, int64_t q_hi, const double* Y, int64_t nr, int32_t d,
                    int32_t kmax, int32_t k0, int64_t self_offset, const double* w, const double* fs,
                    double* dotp_part, double* dist_out)
{
    const int64_t nq = q_hi - q_lo;
    const int K = kmax - k0;
    int rc = select_device(device);
    if (rc != MCE_OK) return rc;
    const double* Xs = X + q_lo * (int64_t)d;
    SameSetHint hint(Xs == Y && nq == nr && self_offset + q_lo == 0);
    Plan p;
    rc = make_plan(nq, nr, d, K, k0 == 1 ? MCE_SELF_EXCLUDE : MCE_SELF_NONE, p);
    if (rc != MCE_OK) return rc;
    const size_t wsb = p.total + dotp_ws_bytes(nq, kmax);
    const bool inside = Xs >= Y && Xs + (size_t)nq * d <= Y + (size_t)nr * d && (Xs - Y) % d == 0;     // as in mce_knn_f64
    DevBuf dX, dY, dW, dF, dO, dD, ws;
    if (!inside) MCE_HIP(dX.alloc((size_t)nq * d * sizeof(double)));
    MCE_HIP(dY.alloc((size_t)nr * d * sizeof(double)));
    MCE_HIP(dW.alloc((size_t)nq * sizeof(double)));
    MCE_HIP(dF.alloc((size_t)nq * sizeof(double)));
    MCE_HIP(dO.alloc((size_t)kmax * sizeof(double)));
    if (dist_out) MCE_HIP(dD.alloc((size_t)nq * K * sizeof(double)));
    MCE_HIP(ws.alloc(wsb));
    if (!inside) MCE_HIP(hipMemcpy(dX.p, Xs, (size_t)nq * d * sizeof(double), hipMemcpyHostToDevice));
    MCE_HIP(hipMemcpy(dY.p, Y, (size_t)nr * d * sizeof(double), hipMemcpyHostToDevice));
    const double* dXp = inside ? dY.as<double>() + (Xs - Y) : dX.as<double>();
    MCE_HIP(hipMemcpy(dW.p, w + q_lo, (size_t)nq * sizeof(double), hipMemcpyHostToDevice));
    MCE_HIP(hipMemcpy(dF.p, fs + q_lo, (size_t)nq * sizeof(double), hipMemcpyHostToDevice));
    rc = mce_knn_dotp_f64_dev(dXp, nq, dY.as<double>(), nr, d, kmax, k0, self_offset + q_lo,
                              dW.as<double>(), dF.as<double>(), dO.as<double>(), dist_out ? dD.as<double>() : nullptr,
                              ws.p, wsb, nullptr);
    if (rc != MCE_OK) return rc;
    MCE_HIP(hipDeviceSynchronize());
    MCE_HIP(hipMemcpy(dotp_part, dO.p, (size_t)kmax * sizeof(double), hipMemcpyDeviceToHost));
    if (dist_out) MCE_HIP(hipMemcpy(dist_out + q_lo * (int64_t)K, dD.p, (size_t)nq * K * sizeof(double), hipMemcpyDeviceToHost));
    return MCE_OK;
}

// ---- evidence feed: covariance -> eigen-system -> whitening -> search -> reduction ---------------
// One problem is four stages; only B runs on the host:
//   A  upload the raw rows, enqueue the covariance kernels, copy cov back (async, pinned)
//   B  d x d Jacobi eigen-solve, whitening scales, Jacobian
//   C  upload eVec/scale, whiten in place, fused search + reduction, copy dotp back (async, pinned)
//   D  hand the results to the caller
// A batch is pipelined two deep in groups of kFeedGroup problems: while the device runs stage C of
// group g, the host performs the (blocking, pageable) uploads of group g+1 and that group's
// covariance kernels run beside the searches on a second stream set.  The searches themselves fill
// the device (make_plan splits the reference set of a small problem over all CUs), so the gain is
// hiding the PCIe upload, the host eigen-solves and the per-problem synchronisations.  Problems are
// processed in waves bounded by kWaveBytes of device memory.
constexpr size_t kWaveBytes = (size_t)8 << 30;
constexpr int kWaveMaxJobs = 1024;
constexpr int kFeedStreams = 4;   // per set (upload+covariance | whiten+search)
constexpr int kFeedGroup = 8;     // problems per pipeline step

struct FeedJob {
    mce_feed_problem* q = nullptr;
    int64_t index = 0;
    Plan plan;
    int k0 = 1, K = 0, rc = MCE_OK;
    int64_t nr = 0, ntot = 0;
    size_t wsb = 0, dev_bytes = 0, host_bytes = 0;
    size_t o_S = 0, o_W = 0, o_F = 0, o_O = 0, o_small = 0, o_part = 0, o_ws = 0;
    char* dbase = nullptr;    // this job's slice of the wave's device arena
    double* hbase = nullptr;  // this job's slice of the pinned host arena: cov[2] | evec[2] | scale[2] | dotp
    double jac = 0.0;
    std::vector<double> lam;  // eigenvalues of the system that defines J (s1's in 'single' mode)
    std::string err;
    hipEvent_t upload_ev = nullptr;   // orders the job's stream behind its blocking uploads
    ~FeedJob() { if (upload_ev) (void)hipEventDestroy(upload_ev); }
    FeedJob() = default;
    FeedJob(const FeedJob&) = delete;
    FeedJob& operator=(const FeedJob&) = delete;

    int d() const { return q->d; }
    double* dS1() const { return reinterpret_cast<double*>(dbase + o_S); }
    double* dS2() const { return dS1() + (size_t)q->n1 * q->d; }
    double* dW() const { return reinterpret_cast<double*>(dbase + o_W); }
    double* dF() const { return reinterpret_cast<double*>(dbase + o_F); }
    double* dO() const { return reinterpret_cast<double*>(dbase + o_O); }
    double* d_mean3() const { return reinterpret_cast<double*>(dbase + o_small); }
    double* d_cov() const { return d_mean3() + 3 * 64; }
    double* d_evec() const { return d_cov() + (size_t)q->d * q->d; }
    double* d_scale() const { return d_evec() + (size_t)q->d * q->d; }
    double* d_part() const { return reinterpret_cast<double*>(dbase + o_part); }
    char* ws() const { return dbase + o_ws; }
    double* h_cov(int i) const { return hbase + (size_t)i * q->d * q->d; }
    double* h_evec(int i) const { return hbase + (size_t)(2 + i) * q->d * q->d; }
    double* h_scale(int i) const { return hbase + (size_t)4 * q->d * q->d + (size_t)i * q->d; }
    double* h_dotp() const { return hbase + (size_t)4 * q->d * q->d + (size_t)2 * q->d; }
    bool two_systems() const { return q->cov_mode == 1 && q->S2 != nullptr; }
    void set_error(int code) { rc = code; err = g_err; }
};

// argument checks + sizes; no device work
int feed_plan(FeedJob& j)
{
    const mce_feed_problem& q = *j.q;
    if (!q.S1 || !q.w || !q.fs || !q.dotp) return fail(MCE_ERR_INVALID, "null pointer argument");
    if (q.n1 < 2 || q.d < 1 || q.ld1 < q.d || (q.S2 && (q.n2 < 1 || q.ld2 < q.d)) || (q.cov_mode != 0 && q.cov_mode != 1))
        return fail(MCE_ERR_INVALID, "invalid sizes n1=%lld ld1=%lld n2=%lld ld2=%lld d=%d cov_mode=%d", (long long)q.n1, (long long)q.ld1,
                    (long long)q.n2, (long long)q.ld2, q.d, q.cov_mode);
    if (q.d > 63) return fail(MCE_ERR_DIM_RANGE, "device feeders support d <= 63 (got %d)", q.d);
    j.k0 = q.S2 ? 0 : 1;
    j.K = q.kmax - j.k0;
    if (q.kmax <= j.k0) return fail(MCE_ERR_INVALID, "kmax=%d must exceed k0=%d", q.kmax, j.k0);
    j.nr = q.S2 ? q.n2 : q.n1;
    j.ntot = q.n1 + (q.S2 ? q.n2 : 0);
    SameSetHint hint(q.S2 == nullptr);          // auto evidence: one set; cross evidence: never the symmetric sweep
    int rc = make_plan(q.n1, j.nr, q.d, j.K, j.k0 == 1 ? MCE_SELF_EXCLUDE : MCE_SELF_NONE, j.plan);
    if (rc != MCE_OK) return rc;
    j.wsb = j.plan.total + dotp_ws_bytes(q.n1, q.kmax);
    const int d = q.d, npair = d * (d + 1) / 2;
    size_t off = 0;
    j.o_S = off;     off = align_up(off + (size_t)j.ntot * d * sizeof(double), 256);
    j.o_W = off;     off = align_up(off + (size_t)q.n1 * sizeof(double), 256);
    j.o_F = off;     off = align_up(off + (size_t)q.n1 * sizeof(double), 256);
    j.o_O = off;     off = align_up(off + (size_t)q.kmax * sizeof(double), 256);
    j.o_small = off; off = align_up(off + (size_t)(3 * 64 + 2 * d * d + d) * sizeof(double), 256);     // mean3 | cov | evec | scale
    j.o_part = off;  off = align_up(off + (size_t)std::max<int64_t>((int64_t)mce::kCovBlocks * npair, (int64_t)mce::kMeanBlocks * mce::kStatStride) * sizeof(double), 256);
    j.o_ws = off;    off = align_up(off + j.wsb, 256);
    j.dev_bytes = off;
    j.host_bytes = align_up((size_t)(4 * d * d + 2 * d + q.kmax) * sizeof(double), 64);
    return MCE_OK;
}

int feed_stage_a(FeedJob& j, hipStream_t st)
{
    const mce_feed_problem& q = *j.q;
    const int d = q.d;
    const size_t row = (size_t)d * sizeof(double);
    // Uploads: blocking copies from the caller's pageable arrays (measured faster than hipMemcpyAsync on the job's
    // non-blocking stream: 300 Planck-sized chains 0.131 vs 0.153 s), followed by an EXPLICIT dependency -- an event
    // recorded on the stream the copies ran on, waited for by the job's stream -- so the covariance kernels behind them
    // are ordered after the uploads by the API's rules, not by how this runtime happens to implement a pageable copy
    // (a blocking hipMemcpy from pageable memory only promises that the SOURCE has been consumed on return, and
    // hipStreamNonBlocking streams do not synchronise with the legacy default stream).  MCE_FEED_UPLOAD=async: the
    // copies themselves on the job's stream.
    static const bool async_upload = [] { const char* e = getenv("MCE_FEED_UPLOAD"); return e && !strcmp(e, "async"); }();
    if (async_upload || st == nullptr) {
        MCE_HIP(hipMemcpy2DAsync(j.dS1(), row, q.S1, (size_t)q.ld1 * sizeof(double), row, (size_t)q.n1, hipMemcpyHostToDevice, st));
        if (q.S2) MCE_HIP(hipMemcpy2DAsync(j.dS2(), row, q.S2, (size_t)q.ld2 * sizeof(double), row, (size_t)q.n2, hipMemcpyHostToDevice, st));
        MCE_HIP(hipMemcpyAsync(j.dW(), q.w, (size_t)q.n1 * sizeof(double), hipMemcpyHostToDevice, st));
        MCE_HIP(hipMemcpyAsync(j.dF(), q.fs, (size_t)q.n1 * sizeof(double), hipMemcpyHostToDevice, st));
    } else {
        MCE_HIP(hipMemcpy2D(j.dS1(), row, q.S1, (size_t)q.ld1 * sizeof(double), row, (size_t)q.n1, hipMemcpyHostToDevice));
        if (q.S2) MCE_HIP(hipMemcpy2D(j.dS2(), row, q.S2, (size_t)q.ld2 * sizeof(double), row, (size_t)q.n2, hipMemcpyHostToDevice));
        MCE_HIP(hipMemcpy(j.dW(), q.w, (size_t)q.n1 * sizeof(double), hipMemcpyHostToDevice));
        MCE_HIP(hipMemcpy(j.dF(), q.fs, (size_t)q.n1 * sizeof(double), hipMemcpyHostToDevice));
        if (!j.upload_ev) MCE_HIP(hipEventCreateWithFlags(&j.upload_ev, hipEventDisableTiming));
        MCE_HIP(hipEventRecord(j.upload_ev, nullptr));
        MCE_HIP(hipStreamWaitEvent(st, j.upload_ev, 0));
    }
    // "all": one eigen-system from s1 U s2; "single": s1's own, and s2's own for s2 (J stays s1's)
    int rc = launch_covariance(j.dS1(), q.cov_mode == 0 ? j.ntot : q.n1, d, j.d_part(), j.d_mean3(), j.d_cov(), st);
    if (rc != MCE_OK) return rc;
    MCE_HIP(hipMemcpyAsync(j.h_cov(0), j.d_cov(), (size_t)d * d * sizeof(double), hipMemcpyDeviceToHost, st));
    if (j.two_systems()) {
        rc = launch_covariance(j.dS2(), q.n2, d, j.d_part(), j.d_mean3(), j.d_cov(), st);
        if (rc != MCE_OK) return rc;
        MCE_HIP(hipMemcpyAsync(j.h_cov(1), j.d_cov(), (size_t)d * d * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    return MCE_OK;
}

int feed_stage_b(FeedJob& j)
{
    const int d = j.d();
    const int nsys = j.two_systems() ? 2 : 1;
    for (int sidx = 0; sidx < nsys; ++sidx) {
        std::vector<double> cov(j.h_cov(sidx), j.h_cov(sidx) + (size_t)d * d), lam, V;
        jacobi_eig(cov, d, lam, V);
        for (int i = 0; i < d; ++i) {
            if (lam[i] != lam[i] || std::isinf(lam[i])) return fail(MCE_ERR_INVALID, "samples contain NaN or infinity (non-finite covariance)");
            if (!(lam[i] > 0.0)) return fail(MCE_ERR_INVALID, "math domain error: covariance eigenvalue %d is %g (use fewer parameters, ndim)", i, lam[i]);
        }
        std::copy(V.begin(), V.end(), j.h_evec(sidx));
        for (int i = 0; i < d; ++i) j.h_scale(sidx)[i] = 1.0 / std::sqrt(lam[i]);
        if (sidx == 0) {
            double logdet = 0.0;
            for (int i = 0; i < d; ++i) logdet += std::log(lam[i]);
            j.jac = std::exp(0.5 * logdet);
            j.lam = lam;
        }
    }
    return MCE_OK;
}

int feed_whiten(FeedJob& j, int sidx, double* rows, int64_t n, hipStream_t st)
{
    const int d = j.d();
    MCE_HIP(hipMemcpyAsync(j.d_evec(), j.h_evec(sidx), (size_t)d * d * sizeof(double), hipMemcpyHostToDevice, st));
    MCE_HIP(hipMemcpyAsync(j.d_scale(), j.h_scale(sidx), (size_t)d * sizeof(double), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(mce::whiten_kernel, dim3((unsigned)((n + mce::kWhitenRows - 1) / mce::kWhitenRows)), dim3(mce::kWhitenRows),
                       mce::whiten_lds_bytes(d), st, rows, n, d, j.d_evec(), j.d_scale(), rows);
    MCE_HIP(hipGetLastError());
    return MCE_OK;
}

int feed_stage_c(FeedJob& j, hipStream_t st)
{
    const mce_feed_problem& q = *j.q;
    {
        static std::atomic<bool> attr_set[kMaxDevices];
        int dev = 0;
        MCE_HIP(hipGetDevice(&dev));
        if (dev < kMaxDevices && !attr_set[dev].load()) {
            MCE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mce::whiten_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mce::whiten_lds_bytes(63)));
            attr_set[dev].store(true);
        }
    }
    int rc;
    if (j.two_systems()) {
        rc = feed_whiten(j, 0, j.dS1(), q.n1, st);
        if (rc != MCE_OK) return rc;
        rc = feed_whiten(j, 1, j.dS2(), q.n2, st);
    } else {
        rc = feed_whiten(j, 0, j.dS1(), q.cov_mode == 0 ? j.ntot : q.n1, st);
    }
    if (rc != MCE_OK) return rc;
    SameSetHint hint(q.S2 == nullptr);          // as in feed_plan: the workspace was sized with it
    rc = mce_knn_dotp_f64_dev(j.dS1(), q.n1, q.S2 ? j.dS2() : j.dS1(), j.nr, q.d, q.kmax, j.k0, 0, j.dW(), j.dF(), j.dO(), nullptr,
                              j.ws(), j.wsb, st);
    if (rc != MCE_OK) return rc;
    MCE_HIP(hipMemcpyAsync(j.h_dotp(), j.dO(), (size_t)q.kmax * sizeof(double), hipMemcpyDeviceToHost, st));
    return MCE_OK;
}

void feed_stage_d(FeedJob& j)
{
    mce_feed_problem& q = *j.q;
    std::copy(j.h_dotp(), j.h_dotp() + q.kmax, q.dotp);
    q.jacobian = j.jac;
    if (q.eigenvalues) std::copy(j.lam.begin(), j.lam.end(), q.eigenvalues);
}

// all jobs of one device, in waves; per-job failures are recorded in the job, a failure of the
// machinery itself (allocation, stream) is returned
int feed_run_on_device(int device, std::vector<FeedJob*>& jobs)
{
    int rc = select_device(device);
    if (rc != MCE_OK) return rc;
    // two stream sets so that the covariance of the NEXT group never queues behind the searches of
    // the current one; a single problem runs on the default stream
    std::vector<hipStream_t> sa, sc;
    std::vector<hipEvent_t> events;
    struct Guard {
        std::vector<hipStream_t>&a, &c;
        std::vector<hipEvent_t>& e;
        ~Guard()
        {
            for (hipStream_t x : a) (void)hipStreamDestroy(x);
            for (hipStream_t x : c) (void)hipStreamDestroy(x);
            for (hipEvent_t x : e) (void)hipEventDestroy(x);
        }
    } guard{sa, sc, events};
    const bool piped = jobs.size() > 1;
    if (piped) {
        const int ns = (int)std::min<size_t>(kFeedStreams, jobs.size());
        for (int i = 0; i < ns; ++i) {
            hipStream_t s;
            MCE_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            sa.push_back(s);
            MCE_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            sc.push_back(s);
        }
        for (int i = 0; i < kFeedGroup; ++i) {
            hipEvent_t e;
            MCE_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            events.push_back(e);
        }
    }
    size_t wave_bytes = kWaveBytes;
    if (const size_t wb = read_tuning().feed_wave_bytes) wave_bytes = wb;   // tests: force several waves
    size_t lo = 0;
    while (lo < jobs.size()) {
        size_t hi = lo, dev_bytes = 0, host_bytes = 0;
        while (hi < jobs.size() && (hi == lo || (dev_bytes + jobs[hi]->dev_bytes <= wave_bytes && hi - lo < (size_t)kWaveMaxJobs))) {
            dev_bytes += jobs[hi]->dev_bytes;
            host_bytes += jobs[hi]->host_bytes;
            ++hi;
        }
        DevBuf arena;
        // destroyed BEFORE the arena: an early return (a failing HIP call in the loops below) must not hand the
        // arena back to the pool while kernels of other jobs are still running on the other streams
        struct Quiesce {
            bool armed = true;
            ~Quiesce() { if (armed) (void)hipDeviceSynchronize(); }
        } quiesce;
        PinnedArena& pinned = g_pinned;
        MCE_HIP(arena.alloc(dev_bytes));
        MCE_HIP(pinned.reserve(host_bytes));
        size_t doff = 0, hoff = 0;
        for (size_t i = lo; i < hi; ++i) {
            jobs[i]->dbase = static_cast<char*>(arena.p) + doff;
            jobs[i]->hbase = reinterpret_cast<double*>(static_cast<char*>(pinned.p) + hoff);
            doff += jobs[i]->dev_bytes;
            hoff += jobs[i]->host_bytes;
        }
        if (!piped) {
            FeedJob& j = *jobs[lo];
            if (j.rc == MCE_OK) {
                int r = feed_stage_a(j, nullptr);
                if (r == MCE_OK) { MCE_HIP(hipStreamSynchronize(nullptr)); r = feed_stage_b(j); }
                if (r == MCE_OK) r = feed_stage_c(j, nullptr);
                if (r == MCE_OK) { MCE_HIP(hipStreamSynchronize(nullptr)); feed_stage_d(j); }
                else { j.set_error(r); (void)hipStreamSynchronize(nullptr); }
            }
            quiesce.armed = false;
            lo = hi;
            continue;
        }
        // groups of kFeedGroup problems, two deep: while the device searches group g the host uploads
        // group g+1 (blocking pageable copies) and its covariance kernels run beside the searches
        auto stage_a_group = [&](size_t g0, size_t g1) -> int {
            for (size_t i = g0; i < g1; ++i) {
                FeedJob& j = *jobs[i];
                if (j.rc != MCE_OK) continue;
                hipStream_t st = sa[i % sa.size()];
                const int r = feed_stage_a(j, st);
                if (r != MCE_OK) { j.set_error(r); continue; }
                MCE_HIP(hipEventRecord(events[i - g0], st));
            }
            return MCE_OK;
        };
        rc = stage_a_group(lo, std::min(hi, lo + (size_t)kFeedGroup));
        if (rc != MCE_OK) return rc;
        for (size_t g0 = lo; g0 < hi; g0 += kFeedGroup) {
            const size_t g1 = std::min(hi, g0 + (size_t)kFeedGroup);
            for (size_t i = g0; i < g1; ++i) {
                FeedJob& j = *jobs[i];
                if (j.rc != MCE_OK) continue;
                MCE_HIP(hipEventSynchronize(events[i - g0]));
                int r = feed_stage_b(j);
                if (r == MCE_OK) r = feed_stage_c(j, sc[i % sc.size()]);
                if (r != MCE_OK) j.set_error(r);
            }
            if (g1 < hi) {
                rc = stage_a_group(g1, std::min(hi, g1 + (size_t)kFeedGroup));
                if (rc != MCE_OK) return rc;
            }
        }
        for (hipStream_t st : sc) MCE_HIP(hipStreamSynchronize(st));
        for (hipStream_t st : sa) MCE_HIP(hipStreamSynchronize(st));     // (jobs that failed after stage A left work there)
        quiesce.armed = false;
        for (size_t i = lo; i < hi; ++i)
            if (jobs[i]->rc == MCE_OK) feed_stage_d(*jobs[i]);
        lo = hi;
    }
    return MCE_OK;
}

}  // namespace

extern "C" {

size_t mce_feed_problem_size(void) { return sizeof(mce_feed_problem); }

int mce_evidence_feed_batch_f64(mce_feed_problem* problems, int64_t nprob, const int32_t* devices, int32_t ndev)
{
    if (nprob < 0 || (nprob > 0 && !problems)) return fail(MCE_ERR_INVALID, "invalid problem list");
    if (nprob == 0) return MCE_OK;
    std::vector<FeedJob> jobs((size_t)nprob);
    for (int64_t i = 0; i < nprob; ++i) {
        jobs[i].q = &problems[i];
        jobs[i].index = i;
        problems[i].status = MCE_OK;
        problems[i].jacobian = 0.0;
        const int r = feed_plan(jobs[i]);
        if (r != MCE_OK) jobs[i].set_error(r);
    }
    std::vector<int> devs;
    if (!devices || ndev <= 0) devs.push_back(0);
    else devs.assign(devices, devices + ndev);
    const int n = (int)std::min<int64_t>((int64_t)devs.size(), nprob);
    // greedy balance by pair count (largest first), then restore the caller's order per device
    std::vector<std::vector<FeedJob*>> per_dev(n);
    if (n == 1) {
        for (auto& j : jobs) if (j.rc == MCE_OK) per_dev[0].push_back(&j);
    } else {
        std::vector<FeedJob*> order;
        for (auto& j : jobs) if (j.rc == MCE_OK) order.push_back(&j);
        std::stable_sort(order.begin(), order.end(), [](const FeedJob* a, const FeedJob* b) {
            return (double)a->q->n1 * (double)a->nr > (double)b->q->n1 * (double)b->nr;
        });
        std::vector<double> load(n, 0.0);
        for (FeedJob* j : order) {
            const int t = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            load[t] += (double)j->q->n1 * (double)j->nr + 1e6;
            per_dev[t].push_back(j);
        }
        for (auto& v : per_dev) std::sort(v.begin(), v.end(), [](const FeedJob* a, const FeedJob* b) { return a->index < b->index; });
    }
    std::vector<int> rcs(n, MCE_OK);
    std::vector<std::string> errs(n);
    auto work = [&](int i) {
        if (per_dev[i].empty()) return;
        rcs[i] = feed_run_on_device(devs[i], per_dev[i]);
        if (rcs[i] != MCE_OK) errs[i] = g_err;
    };
    if (n == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        const CallOptions inherited = t_opt;
        for (int i = 0; i < n; ++i) th.emplace_back([&, i]() { t_opt = inherited; work(i); });
        for (auto& t : th) t.join();
    }
    for (int i = 0; i < n; ++i)
        if (rcs[i] != MCE_OK) return fail(rcs[i], "device %d: %s", devs[i], errs[i].c_str());
    int first = MCE_OK;
    for (int64_t i = 0; i < nprob; ++i) {
        problems[i].status = jobs[i].rc;
        if (jobs[i].rc != MCE_OK && first == MCE_OK) {
            first = jobs[i].rc;
            if (nprob == 1) fail(first, "%s", jobs[i].err.c_str());
            else fail(first, "problem %lld: %s", (long long)i, jobs[i].err.c_str());
        }
    }
    return first;
}

int mce_evidence_feed_f64(const double* S1, int64_t n1, int64_t ld1, const double* S2, int64_t n2, int64_t ld2,
                          int32_t d, int32_t cov_mode, int32_t kmax, const double* w, const double* fs,
                          double* dotp, double* jacobian, double* eigenvalues, int32_t device)
{
    if (!jacobian) return fail(MCE_ERR_INVALID, "null pointer argument");
    mce_feed_problem q;
    std::memset(&q, 0, sizeof(q));
    q.S1 = S1; q.n1 = n1; q.ld1 = ld1;
    q.S2 = S2; q.n2 = S2 ? n2 : 0; q.ld2 = S2 ? ld2 : 0;
    q.d = d; q.cov_mode = cov_mode; q.kmax = kmax;
    q.w = w; q.fs = fs; q.dotp = dotp; q.eigenvalues = eigenvalues;
    const int rc = mce_evidence_feed_batch_f64(&q, 1, &device, 1);
    if (rc == MCE_OK) *jacobian = q.jacobian;
    return rc;
}

int mce_knn_dotp_f64(const double* X, int64_t nq, const double* Y, int64_t nr, int32_t d, int32_t kmax,
                     int32_t k0, int64_t self_offset, const double* w, const double* fs, double* dotp,
                     double* dist_out, const int32_t* devices, int32_t ndev)
{
    if (!X || !Y || !w || !fs || !dotp) return fail(MCE_ERR_INVALID, "null pointer argument");
    if (k0 != 0 && k0 != 1) return fail(MCE_ERR_INVALID, "k0 must be 0 (cross) or 1 (auto), got %d", k0);
    if (kmax <= k0 || nq < 1) return fail(MCE_ERR_INVALID, "invalid kmax=%d k0=%d nq=%lld", kmax, k0, (long long)nq);
    {   // validate before touching any device
        Plan p;
        int rc = make_plan(nq, nr, d, kmax - k0, k0 == 1 ? MCE_SELF_EXCLUDE : MCE_SELF_NONE, p);
        if (rc != MCE_OK) return rc;
    }
    std::vector<int> devs;
    if (!devices || ndev <= 0) devs.push_back(0);
    else devs.assign(devices, devices + ndev);
    const int n = (int)std::min<int64_t>((int64_t)devs.size(), nq);
    std::vector<std::vector<double>> parts(n, std::vector<double>(kmax, 0.0));
    std::vector<int> rcs(n, MCE_OK);
    std::vector<std::string> errs(n);
    // auto evidence over one set: let each device take a library-chosen part (rows for the sweep, blocks of the
    // shared k-d order for the pruned walk) instead of a row range
    const bool whole_set = n > 1 && X == Y && nq == nr && k0 == 1 && self_offset == 0 && !dist_out;
    auto work = [&](int i) {
        const int64_t lo = nq * i / n, hi = nq * (i + 1) / n;
        if (whole_set) rcs[i] = mce_knn_dotp_part_f64(Y, nr, d, kmax, i, n, w, fs, parts[i].data(), devs[i]);
        else rcs[i] = fused_on_device(devs[i], X, lo, hi, Y, nr, d, kmax, k0, self_offset, w, fs, parts[i].data(), dist_out);
        if (rcs[i] != MCE_OK) errs[i] = g_err;   // g_err is thread-local
    };
    if (n == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        const CallOptions inherited = t_opt;
        for (int i = 0; i < n; ++i) th.emplace_back([&, i]() { t_opt = inherited; work(i); });
        for (auto& t : th) t.join();
    }
    for (int i = 0; i < n; ++i)
        if (rcs[i] != MCE_OK) return fail(rcs[i], "device %d: %s", devs[i], errs[i].c_str());
    for (int k = 0; k < kmax; ++k) {   // fixed device order -> reproducible
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += parts[i][k];
        dotp[k] = s;
    }
    return MCE_OK;
}

int mce_knn_dotp_part_f64(const double* Y, int64_t nr, int32_t d, int32_t kmax, int32_t part, int32_t nparts, const double* w,
                          const double* fs, double* dotp, int32_t device)
{
    if (!Y || !w || !fs || !dotp) return fail(MCE_ERR_INVALID, "null pointer argument");
    if (kmax <= 1 || nr < 1) return fail(MCE_ERR_INVALID, "invalid kmax=%d nr=%lld", kmax, (long long)nr);
    Plan p;
    int rc = make_plan(nr, nr, d, kmax - 1, MCE_SELF_EXCLUDE, p);
    if (rc != MCE_OK) return rc;
    rc = select_device(device);
    if (rc != MCE_OK) return rc;
    const size_t wsb = p.total + dotp_ws_bytes(nr, kmax);
    DevBuf dY, dW, dF, dO, ws;
    MCE_HIP(dY.alloc((size_t)nr * d * sizeof(double)));
    MCE_HIP(dW.alloc((size_t)nr * sizeof(double)));
    MCE_HIP(dF.alloc((size_t)nr * sizeof(double)));
    MCE_HIP(dO.alloc((size_t)kmax * sizeof(double)));
    MCE_HIP(ws.alloc(wsb));
    MCE_HIP(hipMemcpy(dY.p, Y, (size_t)nr * d * sizeof(double), hipMemcpyHostToDevice));
    MCE_HIP(hipMemcpy(dW.p, w, (size_t)nr * sizeof(double), hipMemcpyHostToDevice));
    MCE_HIP(hipMemcpy(dF.p, fs, (size_t)nr * sizeof(double), hipMemcpyHostToDevice));
    rc = mce_knn_dotp_part_f64_dev(dY.as<double>(), nr, d, kmax, part, nparts, dW.as<double>(), dF.as<double>(), dO.as<double>(), ws.p, wsb, nullptr);
    if (rc != MCE_OK) return rc;
    MCE_HIP(hipDeviceSynchronize());
    MCE_HIP(hipMemcpy(dotp, dO.p, (size_t)kmax * sizeof(double), hipMemcpyDeviceToHost));
    return MCE_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Test hook: ONE 32x32 tile of the filter's matrix product, exactly as the search kernels issue it
// (v_mfma_f32_32x32x16_f16, KST chained k-steps, C-in = 0), from caller-made fp16 rows.  The rigorous bound of
// knn_f16.hpp rests on a model of this instruction -- products of two fp16 exact in fp32, accumulation error at most
// 32 KST 2^-24 (|x'| |y'|) -- which tests/test_gpu_parity.py::test_mfma_error_model measures directly.
// ---------------------------------------------------------------------------
namespace mce {
template <int KST>
__global__ __launch_bounds__(64) void mfma_tile_probe_kernel(const _Float16* __restrict__ yp, const _Float16* __restrict__ xp, float* __restrict__ out)
{
    const int lane = threadIdx.x;
    v16f acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) {
        // A fragment: lane l <-> reference row l & 31, k = 16 ks + 8 (l >> 5) + e; B fragment: query l & 31, the same k
        const v8h a = *reinterpret_cast<const v8h*>(yp + (lane & 31) * (16 * KST) + 16 * ks + 8 * (lane >> 5));
        const v8h b = *reinterpret_cast<const v8h*>(xp + (lane & 31) * (16 * KST) + 16 * ks + 8 * (lane >> 5));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    // C layout: lane l -> query column l & 31, reference rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
}
}  // namespace mce

extern "C" int mce_debug_mfma_tile_f16(const uint16_t* yprime, const uint16_t* xprime, int32_t kst, float* out, int32_t device)
{
    if (!yprime || !xprime || !out) return fail(MCE_ERR_INVALID, "null pointer argument");
    if (kst < 1 || kst > 4) return fail(MCE_ERR_INVALID, "kst must be 1..4");
    int rc = select_device(device);
    if (rc != MCE_OK) return rc;
    const size_t nb = (size_t)32 * 16 * kst * sizeof(uint16_t);
    void *dy = nullptr, *dx = nullptr, *dout = nullptr;
    MCE_HIP(hipMalloc(&dy, nb));
    MCE_HIP(hipMalloc(&dx, nb));
    MCE_HIP(hipMalloc(&dout, 1024 * sizeof(float)));
    hipError_t e = hipMemcpy(dy, yprime, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dx, xprime, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const _Float16* y = static_cast<const _Float16*>(dy);
        const _Float16* x = static_cast<const _Float16*>(dx);
        float* o = static_cast<float*>(dout);
        switch (kst) {
            case 1: hipLaunchKernelGGL(mce::mfma_tile_probe_kernel<1>, dim3(1), dim3(64), 0, nullptr, y, x, o); break;
            case 2: hipLaunchKernelGGL(mce::mfma_tile_probe_kernel<2>, dim3(1), dim3(64), 0, nullptr, y, x, o); break;
            case 3: hipLaunchKernelGGL(mce::mfma_tile_probe_kernel<3>, dim3(1), dim3(64), 0, nullptr, y, x, o); break;
            default: hipLaunchKernelGGL(mce::mfma_tile_probe_kernel<4>, dim3(1), dim3(64), 0, nullptr, y, x, o); break;
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, 1024 * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(dy); (void)hipFree(dx); (void)hipFree(dout);
    if (e != hipSuccess) return fail(MCE_ERR_HIP, "mfma tile probe: %s", hipGetErrorString(e));
    return MCE_OK;
}

// ---------------------------------------------------------------------------
// per-call options: the three drop-in entry points with a trailing mce_options (NULL: the defaults)
// ---------------------------------------------------------------------------
namespace {
struct ScopedOptions {
    bool pushed = false;
    int rc = MCE_OK;
    explicit ScopedOptions(const mce_options* o) { if (o) { rc = mce_options_push(o); pushed = rc == MCE_OK; } }
    ~ScopedOptions() { if (pushed) (void)mce_options_pop(); }
};
}  // namespace

extern "C" {

int mce_knn_f64_opt(const double* X, int64_t nq, const double* Y, int64_t nr, int32_t d, int32_t K, int32_t self_mode,
                    int64_t self_offset, double* dist, int64_t* idx, int32_t device, const mce_options* opt)
{
    ScopedOptions so(opt);
    if (so.rc != MCE_OK) return so.rc;
    return mce_knn_f64(X, nq, Y, nr, d, K, self_mode, self_offset, dist, idx, device);
}

int mce_knn_dotp_f64_opt(const double* X, int64_t nq, const double* Y, int64_t nr, int32_t d, int32_t kmax, int32_t k0,
                         int64_t self_offset, const double* w, const double* fs, double* dotp, double* dist_out,
                         const int32_t* devices, int32_t ndev, const mce_options* opt)
{
    ScopedOptions so(opt);
    if (so.rc != MCE_OK) return so.rc;
    return mce_knn_dotp_f64(X, nq, Y, nr, d, kmax, k0, self_offset, w, fs, dotp, dist_out, devices, ndev);
}

int mce_knn_dotp_f64_dev_opt(const double* dX, int64_t nq, const double* dY, int64_t nr, int32_t d, int32_t kmax, int32_t k0,
                             int64_t self_offset, const double* d_w, const double* d_fs, double* d_dotp, double* d_dist_out,
                             void* ws, size_t ws_bytes, void* stream, const mce_options* opt)
{
    ScopedOptions so(opt);
    if (so.rc != MCE_OK) return so.rc;
    return mce_knn_dotp_f64_dev(dX, nq, dY, nr, d, kmax, k0, self_offset, d_w, d_fs, d_dotp, d_dist_out, ws, ws_bytes, stream);
}

size_t mce_knn_workspace_bytes_opt(int64_t nq, int64_t nr, int32_t d, int32_t K, const mce_options* opt)
{
    ScopedOptions so(opt);
    if (so.rc != MCE_OK) return 0;
    return mce_knn_workspace_bytes(nq, nr, d, K);
}

}  // extern "C"
