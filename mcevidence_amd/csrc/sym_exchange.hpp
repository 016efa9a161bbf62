// sym_exchange.hpp -- the kernels either side of the exchange of the all-pairs-once partition (sym_types.hpp: PanelGeom.blk_stride;
// capi_apo.hpp): after its sweep a rank holds, in the buckets of blocks it does not own, the row-side candidates of the other
// ranks' rows -- SymEntry {d2, caller row of the query, sorted row} -- which go to the owners; what it receives goes into the
// buckets of its own blocks, and sym_merge_kernel folds them into the lists as on one GPU.  The collective itself is the
// caller's (mcevidence_amd/parallel.py: all_to_all_single over RCCL); the C library has no RCCL dependency.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sym_types.hpp"

namespace mce {

constexpr int kApoScanThreads = 256;

// Rank s of W owns the blocks s, s + W, ... (sym_types.hpp).  The send buffer is ordered by destination rank and, within a
// rank, by block: position v = base(s) + i of block s + W i, base(s) = blocks of the ranks below s.
MCE_HD inline int apo_rank_count(int nqblk, int s, int W) { return s < nqblk ? (nqblk - s + W - 1) / W : 0; }
MCE_HD inline int apo_rank_base(int nqblk, int s, int W) { const int q = nqblk / W, rem = nqblk % W; return s * q + (s < rem ? s : rem); }
MCE_HD inline int apo_block_at(int nqblk, int v, int W)
{
    const int q = nqblk / W, rem = nqblk % W;
    if (v < rem * (q + 1)) return v / (q + 1) + W * (v % (q + 1));
    const int v2 = v - rem * (q + 1);          // (q >= 1: W <= nqblk)
    return rem + v2 / q + W * (v2 % q);
}

// offs[v] = entries of the foreign blocks before position v (own blocks contribute none); counts[s] = entries for rank s;
// flags_out[b] = the overflow flag of foreign block b (its owner searches such a block again), which is then cleared here --
// this rank has nothing to repair there.  One workgroup.
__global__ __launch_bounds__(kApoScanThreads) void apo_offsets_kernel(const int* __restrict__ bucket_cnt, int* __restrict__ bucket_flag, int cap, int nqblk,
                                                                       int part, int nparts, int* __restrict__ offs, long long* __restrict__ counts,
                                                                       int* __restrict__ flags_out)
{
    __shared__ int part_sum[kApoScanThreads];
    const int t = threadIdx.x;
    const int per = (nqblk + kApoScanThreads - 1) / kApoScanThreads;
    const int v0 = t * per, v1 = v0 + per < nqblk ? v0 + per : nqblk;
    auto ship = [&](int b) -> int {
        if (b % nparts == part) return 0;
        const int c = bucket_cnt[b];
        return c < 0 ? 0 : (c > cap ? cap : c);
    };
    int sum = 0;
    for (int v = v0; v < v1; ++v) sum += ship(apo_block_at(nqblk, v, nparts));
    part_sum[t] = sum;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int i = 0; i < kApoScanThreads; ++i) { const int x = part_sum[i]; part_sum[i] = run; run += x; }
        offs[nqblk] = run;
    }
    __syncthreads();
    int run = part_sum[t];
    for (int v = v0; v < v1; ++v) {
        const int b = apo_block_at(nqblk, v, nparts);
        offs[v] = run;
        run += ship(b);
        const bool own = b % nparts == part;
        flags_out[b] = own ? 0 : (bucket_flag[b] != 0 ? 1 : 0);
        if (!own) bucket_flag[b] = 0;
    }
    __syncthreads();
    if (t < nparts) {
        const int lo = apo_rank_base(nqblk, t, nparts);
        counts[t] = (long long)(offs[lo + apo_rank_count(nqblk, t, nparts)] - offs[lo]);          // (this workgroup's own writes: visible after the barrier)
    }
}

// send[offs[v] + i] = bucket[block at v][i]: one workgroup per position (own blocks and empty ones exit)
__global__ __launch_bounds__(256) void apo_export_kernel(const SymEntry* __restrict__ bucket, const int* __restrict__ offs, int cap, int nqblk, int nparts,
                                                         SymEntry* __restrict__ send)
{
    const int v = blockIdx.x;
    const int n = offs[v + 1] - offs[v];
    if (n <= 0) return;
    const SymEntry* const src = bucket + (int64_t)apo_block_at(nqblk, v, nparts) * cap;
    SymEntry* const dst = send + offs[v];
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

// received candidates into the buckets of their rows' blocks (all of them this rank's own).  A sender's entries come block by
// block, so the 64 entries of a wave mostly share a block: one atomic per RUN of equal blocks in the wave, not one per entry
// (1.7 M entries onto 977 counters at C3 over two ranks).
__global__ __launch_bounds__(256) void apo_import_kernel(const SymEntry* __restrict__ recv, int64_t n, SymEntry* __restrict__ bucket, int* __restrict__ bucket_cnt,
                                                         int* __restrict__ bucket_flag, int cap, int qpb, int nqblk, int part, int nparts)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    SymEntry e;
    e.d2 = 0.0; e.src = 0; e.row = -1;
    if (i < n) e = recv[i];
    int jb = e.row >= 0 ? e.row / qpb : -1;
    if (jb >= nqblk || (jb >= 0 && jb % nparts != part)) jb = -1;          // (not this rank's row: flagged by the caller's check; never a wild store)
    const int prev = __shfl_up(jb, 1, 64);
    const bool head = lane == 0 || jb != prev;
    const unsigned long long heads = __ballot(head);
    const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    const int run_start = 63 - __builtin_clzll(below);                       // (bit 0 is always set: lane 0 is a head)
    const unsigned long long above = lane == 63 ? 0ull : (heads & ~((2ull << lane) - 1ull));
    const int run_end = above ? __builtin_ctzll(above) : 64;
    int base = 0;
    if (lane == run_start && jb >= 0) base = atomicAdd(bucket_cnt + jb, run_end - run_start);
    base = __shfl(base, run_start, 64);
    if (jb < 0) return;
    const int slot = base + (lane - run_start);
    if ((unsigned)slot < (unsigned)cap) bucket[(int64_t)jb * cap + slot] = e;
    else bucket_flag[jb] = 1;
}

// flags of this rank's blocks raised on other ranks (their bucket for the block overflowed there): the block is searched again
__global__ __launch_bounds__(256) void apo_flags_kernel(const int* __restrict__ flags_all, int* __restrict__ bucket_flag, int nqblk, int part, int nparts)
{
    const int b = part + nparts * (blockIdx.x * 256 + threadIdx.x);
    if (b < nqblk && flags_all[b] != 0) bucket_flag[b] = 1;
}

// the units of a launch with strided blocks and / or several chains per block, as a table (knn_panel.hpp reads it)
__global__ __launch_bounds__(256) void panel_unit_table_kernel(PanelGeom g, int nunits, PanelUnit* __restrict__ out)
{
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= nunits) return;
    int p, a, lo, hi;
    panel_unit_decode(u, g, p, a);
    panel_unit_tiles(p, a, g, lo, hi);
    const int S = g.nsplit > 1 ? g.nsplit : 1;
    PanelUnit t;
    t.qblk = a; t.t_lo = lo; t.t_hi = hi;
    t.useq = S > 1 ? p / S : panel_unit_seq(p, a, g);
    t.chain = S > 1 ? a * S + p % S : a;
    t.list_set = S > 1 ? p % S : 0;
    t.pad0 = t.pad1 = 0;
    out[u] = t;
}

// chains (PanelGeom.nsplit): every list set of the blocks part, part + nparts, ... starts empty -- a chain without units (a block
// with fewer panels than chains) leaves its set untouched
__global__ __launch_bounds__(512) void sym_chain_init_kernel(double* __restrict__ pd, int* __restrict__ pi, int64_t nq_pad, int KCAP, int S, int part, int nparts)
{
    const int64_t q = (int64_t)(part + nparts * (int)blockIdx.x) * 512 + threadIdx.x;
    for (int s = 0; s < S; ++s)
        for (int k = 0; k < KCAP; ++k) {
            const int64_t o = ((int64_t)s * KCAP + k) * nq_pad + q;
            pd[o] = __builtin_huge_val();
            pi[o] = -1;
        }
}
// ... and the sets 1 .. S - 1 of the blocks that were searched again (their set 0 is complete) are emptied
__global__ __launch_bounds__(512) void sym_chain_clear_kernel(const int* __restrict__ bucket_flag, double* __restrict__ pd, int* __restrict__ pi, int64_t nq_pad, int KCAP,
                                                              int S, int part, int nparts)
{
    const int b = part + nparts * (int)blockIdx.x;
    if (bucket_flag[b] == 0) return;
    const int64_t q = (int64_t)b * 512 + threadIdx.x;
    for (int s = 1; s < S; ++s)
        for (int k = 0; k < KCAP; ++k) {
            const int64_t o = ((int64_t)s * KCAP + k) * nq_pad + q;
            pd[o] = __builtin_huge_val();
            pi[o] = -1;
        }
}

// list column block -> block: identity (the reduction enumerates every nparts-th block through this table: reduce_kernels.hpp, border)
__global__ __launch_bounds__(256) void apo_iota_kernel(int* __restrict__ out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = i;
}

}  // namespace mce
