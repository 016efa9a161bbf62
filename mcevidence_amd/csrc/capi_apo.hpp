// capi_apo.hpp -- part of capi.hip: the ALL-PAIRS-ONCE partition of auto evidence over W ranks (round 5).
//
// The exchange-free partition (mce_knn_dotp_part_f64_dev) lets a rank multiply its rows against everybody else's from its side
// only; the other rank does the same from the other side: per node every pair of rows of different ranks is multiplied twice
// (ceiling of the efficiency 1 / (2 - 1/W)).  Here every pair is multiplied once per NODE (sym_types.hpp: PanelGeom.blk_stride):
// rank r owns the sorted blocks r, r + W, ... and runs the single-GPU units of those blocks -- block a against the tiles of
// the blocks 0..a, both gates on -- collects the row-side candidates of rows it does not own in the buckets of their blocks,
// and ships them to the owners; what it receives is folded into its own lists.  (A first version gave every rank a contiguous
// range of blocks and the rectangles against half of the other ranks: half of those rectangles then have the rows NEARER the
// mean as queries and the farther, sparser rows on the row side, whose bounds are loose -- 6.5 M candidates shipped from rank 0
// to rank 1 at C3 against 0.19 M the other way, and 19 ms to fold them in; profiles/r05_mid/pairs_once_contiguous.json.)
// Four calls, the collectives between them are the caller's (parallel.py: RCCL all_to_all_single):
//   mce_pairs_once_prepare_dev sort, pack, prepass of the rank's OWN blocks -> their rows' bounds (all-reduce with MIN: everybody's)
//   mce_pairs_once_sweep_dev   the sweep; counts[s] = candidates for rank s, flags[b] = overflowed foreign buckets
//   mce_pairs_once_export_dev  the candidates, densely, ordered by destination rank (16 B each)
//   mce_pairs_once_finish_dev  received candidates -> own buckets; repair launch; merge; volume / weight sums of the rank's own rows
// The workspace carries the state from call to call (the plan is a pure function of the shape).  Reference: MCEvidence.py:1093-1117.
#pragma once
#include "sym_exchange.hpp"

namespace {

// the plan of the symmetric sweep for (nr, d, kmax), or an error: the partition exists only where one GPU would run the
// one-pass symmetric sweep (large set, K <= 16, d >= the pruned walk's reach)
int pairs_once_plan(int64_t nr, int32_t d, int32_t kmax, Plan& p, bool quiet)
{
    if (kmax <= 1) return quiet ? MCE_ERR_INVALID : fail(MCE_ERR_INVALID, "kmax=%d must exceed k0=1", kmax);
    SameSetHint hint(true);
    const int rc = make_plan(nr, nr, d, kmax - 1, MCE_SELF_EXCLUDE, p);
    if (rc != MCE_OK) return rc;
    if (p.prune || !p.sym || p.twopass || !p.vh || !p.vh->launch_panel)
        return quiet ? MCE_ERR_INVALID : fail(MCE_ERR_INVALID, "pairs-once partition: this shape does not take the one-pass symmetric sweep");
    return MCE_OK;
}

// A rank's blocks are fewer than one GPU's: with fewer CHAINS of units than the chip has workgroup slots (2 per CU) the sweep is as
// slow as the longest block's chain of units (measured at C3, one chain per block: sweep kernel 19.9 / 11.3 / 8.9 ms at 2 / 4 / 8
// ranks against 17.4 / 8.7 / 4.4 for perfect shares).  So a block's panels are dealt to S independent chains, each with its own
// list set (merged at the end; the chains share the row's published bound): S such that the rank has ~3 chains per slot, the
// panels short enough that the longest block has two units per chain.  MCE_PAIRS_ONCE_SPLIT / MCE_PAIRS_ONCE_PANEL override.
struct PairsOnceShape {
    int nsplit = 1, panel = 0;
    size_t off_d = 0, off_i = 0;      // list sets [nsplit][KCAP][nq_pad] (nsplit > 1: behind the plan's workspace and the reduction's scratch)
    size_t total = 0;
};
PairsOnceShape pairs_once_shape(const Plan& p, int64_t nr, int32_t kmax, int32_t nparts)
{
    PairsOnceShape sh;
    const int nown = (p.nqblk + nparts - 1) / std::max(nparts, 1);
    // (the two overrides are read ONCE per process: the five entry points below derive the workspace layout from them again and
    //  again, and an environment that changes between the calls on one workspace would misplace the list sets -- ADVICE round 5)
    static const int env_split = [] { const char* e = getenv("MCE_PAIRS_ONCE_SPLIT"); const int v = e ? atoi(e) : 0; return (v >= 1 && v <= 8) ? v : 0; }();
    static const int env_panel = [] { const char* e = getenv("MCE_PAIRS_ONCE_PANEL"); const int v = e ? atoi(e) : 0; return v >= 1 ? v : 0; }();
    int S = std::min(8, std::max(1, (1536 + nown - 1) / std::max(nown, 1)));
    if (env_split) S = env_split;
    const int def_panel = kSymPanelChunks[p.KST];
    int panel = S > 1 ? (int)std::max<int64_t>(8, std::min<int64_t>(def_panel, p.nchunk / (2 * S))) : 0;
    if (env_panel) panel = env_panel;
    sh.nsplit = S;
    sh.panel = panel;
    size_t off = p.total + dotp_ws_bytes(nr, kmax);
    if (S > 1) {
        off = align_up(off, 256);
        sh.off_d = off;
        off = align_up(off + (size_t)S * p.KCAP * (size_t)p.nq_pad * sizeof(double), 256);
        sh.off_i = off;
        off = align_up(off + (size_t)S * p.KCAP * (size_t)p.nq_pad * sizeof(int), 256);
    } else {
        sh.off_d = p.off_pd;
        sh.off_i = p.off_pi;
    }
    sh.total = off;
    return sh;
}
// The four calls of one rank's share work on ONE workspace in a fixed order (prepare, sweep, export, finish) with the same
// arguments; this table -- keyed by the workspace pointer, host memory, no device round trip -- records where a workspace
// stands, and a call that arrives out of order or with other arguments is refused (MCE_ERR_INVALID) instead of reading
// lists and buckets that are not there.
struct ApoCall { int64_t nr; int d, kmax, part, nparts, nsplit, panel, phase; };
std::mutex g_apo_mutex;
std::unordered_map<const void*, ApoCall> g_apo_calls;
// phase_from <= recorded phase <= phase_to required (0: no record needed -- prepare); afterwards the record's phase is phase_set (-1: erased)
int apo_step(const void* ws, int64_t nr, int d, int kmax, int part, int nparts, const PairsOnceShape& sh, int phase_from, int phase_to, int phase_set,
             const char* what)
{
    std::lock_guard<std::mutex> lk(g_apo_mutex);
    auto it = g_apo_calls.find(ws);
    if (phase_from > 0) {
        if (it == g_apo_calls.end())
            return fail(MCE_ERR_INVALID, "pairs-once partition: %s on a workspace that mce_pairs_once_prepare_dev has not prepared", what);
        const ApoCall& c = it->second;
        if (c.nr != nr || c.d != d || c.kmax != kmax || c.part != part || c.nparts != nparts || c.nsplit != sh.nsplit || c.panel != sh.panel)
            return fail(MCE_ERR_INVALID, "pairs-once partition: %s with other arguments than the workspace was prepared with (prepared: nr=%lld d=%d kmax=%d part %d "
                        "of %d, %d chains, panel %d)", what, (long long)c.nr, c.d, c.kmax, c.part, c.nparts, c.nsplit, c.panel);
        if (c.phase < phase_from || c.phase > phase_to)
            return fail(MCE_ERR_INVALID, "pairs-once partition: %s out of order (the calls are prepare, sweep, export, finish; this workspace stands after step %d)",
                        what, c.phase);
    }
    if (phase_set < 0) { if (it != g_apo_calls.end()) g_apo_calls.erase(it); }
    else g_apo_calls[ws] = ApoCall{nr, d, kmax, part, nparts, sh.nsplit, sh.panel, phase_set};
    return MCE_OK;
}
// (the plan's list offsets point at the shape's list sets from here on)
void pairs_once_apply(Plan& p, const PairsOnceShape& sh)
{
    p.apo = true;
    p.apo_nsplit = sh.nsplit;
    p.apo_panel = sh.panel;
    p.off_pd = sh.off_d;
    p.off_pi = sh.off_i;
}

__global__ __launch_bounds__(256) void pairs_once_fill_kernel(unsigned long long* __restrict__ p, int64_t n, unsigned long long v)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}
// every row's row-side gate constant and every tile's maximum from the (all-reduced) bounds -- what the prepass writes for the rows
// it handles (knn_f16.hpp, SYM == 1), here for all rows
__global__ __launch_bounds__(256) void pairs_once_row_gates_kernel(unsigned long long* __restrict__ thr, unsigned* __restrict__ rrow, float* __restrict__ rtile,
                                                                   const double* __restrict__ qinfo, const double* __restrict__ params, int64_t nq, int KST)
{
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;         // (nq_pad rows: a whole number of workgroups)
    float R = 0.0f;
    if (q < nq) {
        R = mce::sym_row_gate(__longlong_as_double((long long)thr[q]), qinfo[2 * q], params, KST);
        rrow[q] = __float_as_uint(R);
    } else {
        thr[q] = 0x7FF0000000000000ull;
        rrow[q] = 0u;
    }
    float m = R;
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 31) == 0) rtile[q >> 5] = m;
}

// the argument block of the symmetric sweep's launches from a plan and its workspace (after run_search has filled it)
void pairs_once_args(const Plan& p, char* ws, int64_t nr, int32_t d, int32_t K, mce::KnnF16Args& a)
{
    char* const sw = ws + p.off_sym;
    a.Yh = reinterpret_cast<_Float16*>(ws + p.off_yh); a.nchunk_total = p.nchunk; a.rsplit = 1;
    a.Xh = reinterpret_cast<_Float16*>(ws + p.off_xh);
    a.qinfo = reinterpret_cast<double*>(ws + p.off_qinfo); a.params = reinterpret_cast<double*>(ws + p.off_params);
    a.X = a.Y = reinterpret_cast<const double*>(sw + p.sl.Ys);
    a.nq = a.nr = nr; a.D = d; a.nq_pad = p.nq_pad; a.nqblk = p.nqblk;
    a.self_exclude = 1; a.self_offset = 0; a.ksel = K;
    a.part_d = reinterpret_cast<double*>(ws + p.off_pd); a.part_i = reinterpret_cast<int*>(ws + p.off_pi);
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
    const Tuning tun = read_tuning();
    a.sym.panel = tun.sym_panel > 0 ? tun.sym_panel : kSymPanelChunks[p.KST];
}

// a result that must not be used: an entry arrived for a row this rank does not own (ranks that disagree about the partition)
__global__ void pairs_once_poison_kernel(const int* __restrict__ err, double* __restrict__ dotp, int kmax)
{
    if (*err != 0 && (int)threadIdx.x < kmax) dotp[threadIdx.x] = __builtin_nan("");
}
__global__ __launch_bounds__(256) void pairs_once_check_kernel(const mce::SymEntry* __restrict__ recv, int64_t n, int qpb, int nqblk, int part, int nparts,
                                                               int* __restrict__ err)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int jb = recv[i].row / qpb;
    if (recv[i].row < 0 || jb >= nqblk || jb % nparts != part) *err = 1;
}

}  // namespace

extern "C" {

int32_t mce_pairs_once_blocks(int64_t nr, int32_t d, int32_t kmax)
{
    Plan p;
    if (pairs_once_plan(nr, d, kmax, p, true) != MCE_OK) return 0;
    return p.nqblk;
}

size_t mce_pairs_once_workspace_bytes(int64_t nr, int32_t d, int32_t kmax, int32_t nparts)
{
    Plan p;
    if (nparts < 2 || pairs_once_plan(nr, d, kmax, p, true) != MCE_OK) return 0;
    return pairs_once_shape(p, nr, kmax, nparts).total;
}

int mce_pairs_once_prepare_dev(const double* dY, int64_t nr, int32_t d, int32_t kmax, int32_t part, int32_t nparts, size_t* bounds_offset,
                               int64_t* bounds_count, void* ws, size_t ws_bytes, void* stream)
{
    if (!dY || !ws || !bounds_offset || !bounds_count) return fail(MCE_ERR_INVALID, "null pointer argument");
    if (nparts < 2 || part < 0 || part >= nparts) return fail(MCE_ERR_INVALID, "part %d of %d (the pairs-once partition needs two ranks or more)", part, nparts);
    Plan p;
    int rc = pairs_once_plan(nr, d, kmax, p, false);
    if (rc != MCE_OK) return rc;
    if (nparts > p.nqblk) return fail(MCE_ERR_INVALID, "pairs-once partition: %d ranks for %d blocks", nparts, p.nqblk);
    const PairsOnceShape sh = pairs_once_shape(p, nr, kmax, nparts);
    if (ws_bytes < sh.total) return fail(MCE_ERR_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes, sh.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* wsc = static_cast<char*>(ws);
    rc = apo_step(ws, nr, d, kmax, part, nparts, sh, 0, 0, 1, "prepare");
    if (rc != MCE_OK) return rc;
    p.part = part;
    p.nparts = nparts;
    pairs_once_apply(p, sh);
    p.apo_phase = 1;
    // every row's bound starts at +inf, every slot empty; the prepass below fills in the rows of this rank's blocks
    const unsigned long long inf_bits = 0x7FF0000000000000ull;
    hipLaunchKernelGGL(pairs_once_fill_kernel, dim3(1024), dim3(256), 0, st, reinterpret_cast<unsigned long long*>(wsc + p.off_sym + p.sl.thr), p.nq_pad, inf_bits);
    MCE_HIP(hipGetLastError());
    hipLaunchKernelGGL(pairs_once_fill_kernel, dim3(2048), dim3(256), 0, st, reinterpret_cast<unsigned long long*>(wsc + p.off_sym + p.sl.slots),
                       p.nq_pad * (int64_t)p.KCAP, inf_bits);
    MCE_HIP(hipGetLastError());
    if (sh.nsplit > 1) {
        static_assert(mce::f16_qpb(4) == 512, "one workgroup per block of list columns");
        hipLaunchKernelGGL(mce::sym_chain_init_kernel, dim3((unsigned)mce::apo_rank_count(p.nqblk, part, nparts)), dim3(512), 0, st,
                           reinterpret_cast<double*>(wsc + sh.off_d), reinterpret_cast<int*>(wsc + sh.off_i), p.nq_pad, p.KCAP, sh.nsplit, (int)part, (int)nparts);
        MCE_HIP(hipGetLastError());
    }
    SameSetHint hint(true);
    rc = run_search(p, dY, nr, dY, nr, d, kmax - 1, MCE_SELF_EXCLUDE, 0, wsc, st);
    if (rc != MCE_OK) return rc;
    *bounds_offset = p.off_sym + p.sl.thr;
    *bounds_count = p.nq_pad;
    return MCE_OK;
}

int mce_pairs_once_sweep_dev(const double* dY, int64_t nr, int32_t d, int32_t kmax, int32_t part, int32_t nparts, int64_t* d_counts,
                             int32_t* d_flags, void* ws, size_t ws_bytes, void* stream)
{
    if (!dY || !d_counts || !d_flags || !ws) return fail(MCE_ERR_INVALID, "null pointer argument");
    if (nparts < 2 || part < 0 || part >= nparts) return fail(MCE_ERR_INVALID, "part %d of %d (the pairs-once partition needs two ranks or more)", part, nparts);
    Plan p;
    int rc = pairs_once_plan(nr, d, kmax, p, false);
    if (rc != MCE_OK) return rc;
    if (nparts > p.nqblk) return fail(MCE_ERR_INVALID, "pairs-once partition: %d ranks for %d blocks", nparts, p.nqblk);
    const PairsOnceShape sh = pairs_once_shape(p, nr, kmax, nparts);
    if (ws_bytes < sh.total) return fail(MCE_ERR_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes, sh.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* wsc = static_cast<char*>(ws);
    rc = apo_step(ws, nr, d, kmax, part, nparts, sh, 1, 1, 2, "sweep");
    if (rc != MCE_OK) return rc;
    p.part = part;
    p.nparts = nparts;
    pairs_once_apply(p, sh);
    p.apo_phase = 2;
    mce::KnnF16Args a;
    pairs_once_args(p, wsc, nr, d, kmax - 1, a);
    // the bounds are everybody's now (MIN over the ranks): the row-side gate constants follow from them
    hipLaunchKernelGGL(pairs_once_row_gates_kernel, dim3((unsigned)(p.nq_pad / 256)), dim3(256), 0, st, a.sym.thr, a.sym.rrow, a.sym.rtile, a.qinfo, a.params, nr, p.KST);
    MCE_HIP(hipGetLastError());
    SameSetHint hint(true);
    rc = run_search(p, dY, nr, dY, nr, d, kmax - 1, MCE_SELF_EXCLUDE, 0, wsc, st);
    if (rc != MCE_OK) return rc;
    if (!p.sym_active) return fail(MCE_ERR_INVALID, "pairs-once partition: the sweep did not run");
    int* offs = reinterpret_cast<int*>(wsc + p.off_sym + p.sl.keys_a);     // [nqblk + 1] offsets + 1 error word (the sort's keys: n_pad words, free by now)
    static_assert(sizeof(long long) == sizeof(int64_t), "counts");
    hipLaunchKernelGGL(mce::apo_offsets_kernel, dim3(1), dim3(mce::kApoScanThreads), 0, st, a.sym.bucket_cnt, a.sym.bucket_flag, a.sym.cap, p.nqblk, (int)part,
                       (int)nparts, offs, reinterpret_cast<long long*>(d_counts), d_flags);
    MCE_HIP(hipGetLastError());
    return MCE_OK;
}

int mce_pairs_once_export_dev(int64_t nr, int32_t d, int32_t kmax, int32_t part, int32_t nparts, void* d_send, void* ws, size_t ws_bytes, void* stream)
{
    if (!ws) return fail(MCE_ERR_INVALID, "null pointer argument");
    Plan p;
    int rc = pairs_once_plan(nr, d, kmax, p, false);
    if (rc != MCE_OK) return rc;
    if (nparts < 2 || nparts > p.nqblk) return fail(MCE_ERR_INVALID, "part %d of %d", part, nparts);
    const PairsOnceShape sh = pairs_once_shape(p, nr, kmax, nparts);
    if (ws_bytes < sh.total) return fail(MCE_ERR_WORKSPACE, "workspace too small");
    rc = apo_step(ws, nr, d, kmax, part, nparts, sh, 2, 3, 3, "export");
    if (rc != MCE_OK) return rc;
    if (!d_send) return MCE_OK;           // (nothing to ship)
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* wsc = static_cast<char*>(ws);
    mce::KnnF16Args a;
    pairs_once_args(p, wsc, nr, d, kmax - 1, a);
    const int* offs = reinterpret_cast<const int*>(wsc + p.off_sym + p.sl.keys_a);
    hipLaunchKernelGGL(mce::apo_export_kernel, dim3((unsigned)p.nqblk), dim3(256), 0, st, a.sym.bucket, offs, a.sym.cap, p.nqblk, (int)nparts,
                       static_cast<mce::SymEntry*>(d_send));
    MCE_HIP(hipGetLastError());
    return MCE_OK;
}

int mce_pairs_once_finish_dev(const double* dY, int64_t nr, int32_t d, int32_t kmax, int32_t part, int32_t nparts, const double* d_w,
                              const double* d_fs, const void* d_recv, int64_t nrecv, const int32_t* d_flags, double* d_dotp, void* ws,
                              size_t ws_bytes, void* stream)
{
    if (!dY || !d_w || !d_fs || !d_dotp || !d_flags || !ws || (nrecv > 0 && !d_recv)) return fail(MCE_ERR_INVALID, "null pointer argument");
    if (nparts < 2 || part < 0 || part >= nparts || nrecv < 0) return fail(MCE_ERR_INVALID, "part %d of %d, %lld entries", part, nparts, (long long)nrecv);
    Plan p;
    int rc = pairs_once_plan(nr, d, kmax, p, false);
    if (rc != MCE_OK) return rc;
    if (nparts > p.nqblk) return fail(MCE_ERR_INVALID, "pairs-once partition: %d ranks for %d blocks", nparts, p.nqblk);
    const PairsOnceShape sh = pairs_once_shape(p, nr, kmax, nparts);
    if (ws_bytes < sh.total) return fail(MCE_ERR_WORKSPACE, "workspace too small");
    rc = apo_step(ws, nr, d, kmax, part, nparts, sh, 2, 3, -1, "finish");      // (export may be skipped by a rank with nothing to ship)
    if (rc != MCE_OK) return rc;
    pairs_once_apply(p, sh);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* wsc = static_cast<char*>(ws);
    const int K = kmax - 1;
    mce::KnnF16Args a;
    pairs_once_args(p, wsc, nr, d, K, a);
    const int nown = mce::apo_rank_count(p.nqblk, part, nparts);          // this rank's blocks: part, part + nparts, ...
    int* err = reinterpret_cast<int*>(wsc + p.off_sym + p.sl.keys_a) + p.nqblk + 1;
    MCE_HIP(mce::zero_async(err, sizeof(int), st));
    const int qpb = mce::f16_qpb(p.KCAP);
    if (nrecv > 0) {
        const unsigned nb = (unsigned)((nrecv + 255) / 256);
        hipLaunchKernelGGL(pairs_once_check_kernel, dim3(nb), dim3(256), 0, st, static_cast<const mce::SymEntry*>(d_recv), nrecv, qpb, p.nqblk, (int)part, (int)nparts, err);
        MCE_HIP(hipGetLastError());
        hipLaunchKernelGGL(mce::apo_import_kernel, dim3(nb), dim3(256), 0, st, static_cast<const mce::SymEntry*>(d_recv), nrecv, a.sym.bucket, a.sym.bucket_cnt,
                           a.sym.bucket_flag, a.sym.cap, qpb, p.nqblk, (int)part, (int)nparts);
        MCE_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(mce::apo_flags_kernel, dim3((unsigned)((nown + 255) / 256)), dim3(256), 0, st, d_flags, a.sym.bucket_flag, p.nqblk, (int)part, (int)nparts);
    MCE_HIP(hipGetLastError());
    a.seed_cfg = 0;
    MCE_HIP(p.vh->launch_sym_repair(a, st));         // own blocks whose bucket overflowed here or elsewhere, or whose units gave up waiting
    if (sh.nsplit > 1 && nown > 0) {
        hipLaunchKernelGGL(mce::sym_chain_clear_kernel, dim3((unsigned)nown), dim3(512), 0, st, a.sym.bucket_flag, a.part_d, a.part_i, p.nq_pad, p.KCAP, sh.nsplit, (int)part, (int)nparts);
        MCE_HIP(hipGetLastError());
    }
    MCE_HIP(launch_sym_merge(p.KCAP, a.part_d, a.part_i, p.nq_pad, a.sym, part, p.nqblk, st, nparts));      // (into list set 0)
    // the reduction enumerates every nparts-th block of list columns through a block table (reduce_kernels.hpp: border): the identity
    hipLaunchKernelGGL(mce::apo_iota_kernel, dim3((unsigned)((p.nqblk + 255) / 256)), dim3(256), 0, st, a.sym.done, p.nqblk);
    MCE_HIP(hipGetLastError());
    p.part = part; p.nparts = nparts; p.sym_qb_lo = 0; p.sym_qb_hi = p.nqblk; p.sym_active = true; p.L = sh.nsplit;
    double* partial = reinterpret_cast<double*>(wsc + p.total);
    rc = launch_merge(p, false, true, dY, dY, nr, d, K, MCE_SELF_EXCLUDE, 0, nullptr, nullptr, 1, (int)kmax, d_w, d_fs, partial, wsc, st);
    if (rc != MCE_OK) return rc;
    const int64_t ncol = (int64_t)nown * qpb;
    const unsigned blocks = (unsigned)std::max<int64_t>((ncol + mce::kRedThreads - 1) / mce::kRedThreads, 1);
    hipLaunchKernelGGL(mce::dotp_final_kernel, dim3((unsigned)kmax), dim3(mce::kRedThreads), 0, st, partial, (int64_t)blocks, 1, (int)kmax, d_dotp);
    MCE_HIP(hipGetLastError());
    hipLaunchKernelGGL(pairs_once_poison_kernel, dim3(1), dim3(64), 0, st, err, d_dotp, (int)kmax);
    MCE_HIP(hipGetLastError());
    return MCE_OK;
}

}  // extern "C"
