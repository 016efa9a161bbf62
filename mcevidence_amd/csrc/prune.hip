// prune.hip -- k-d ordering, boxes and sorted chunk lists for the pruned search (see prune.hpp).
#include "prune.hpp"
#include "zero_fill.hpp"

#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>

namespace mce {

namespace {

constexpr int kThreads = 256;

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

constexpr int kCoordBits = 24;    // coordinate bits in a 64-bit k-d sort key
constexpr int kMinCoordBits32 = 18;   // ... at least so many in a 32-bit one (sign + 8 exponent + 9 mantissa bits: medians to 2e-3 of the coordinate)
constexpr int kAlign = 64;        // units (32-row tiles) per aligned group: the list chunk of the pruned walk
constexpr int kAlignLevels = 6;

int top_tree_levels(int64_t n_units)
{
    const int64_t groups = (n_units + kAlign - 1) / kAlign;
    int L = 0;
    while (((int64_t)1 << L) < groups) ++L;
    return L;
}

// key = (node id at `level`) << kCoordBits | the top kCoordBits order-preserving bits of the (float) coordinate; padding
// rows sort behind everything in their node (which is always the last node).  24 of the 32 bits: a radix pass less per
// level, and 15 mantissa bits place a median to 3e-5 of the coordinate -- rows closer than that keep their previous
// order (stable sort), which only matters below ~30 rows per resolution step (d = 1 with 10^7 rows: the leaf boxes then
// overlap by a few rows' spacing; the search stays exact either way).
// Key: unsigned long long with kCoordBits coordinate bits, or -- when node id and coordinate fit -- unsigned with `cbits` of
// them (half the key bytes and a radix pass less: kd_sort decides).
template <class Key>
__global__ __launch_bounds__(kThreads) void kd_key_kernel(const int* __restrict__ perm_in, int64_t n, int64_t n_pad, int unit_rows,
                                                          int n_units, int top_levels, int level, const float* __restrict__ Cf, int d, int dim,
                                                          int cbits, Key* __restrict__ keys, int* __restrict__ vals, int64_t pos0 = 0, int64_t pos1 = -1)
{
    // (pos0, pos1: the positions [pos0, pos1) only -- one rank's subtree of a distributed preparation; default: all of them)
    const int64_t pos = pos0 + (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (pos >= (pos1 < 0 ? n_pad : pos1)) return;
    const int row = perm_in ? perm_in[pos] : (pos < n ? (int)pos : -1);
    // ALIGNED tree: the top levels split whole groups of kAlign units (2048 rows) between the children, the
    // last log2(kAlign) levels halve a group bit by bit -- so every aligned run of 2, 16 or 64 units (a
    // wave's query tiles, a query block, a list chunk) is exactly one subtree: a tight box.  (Splitting at
    // (lo+hi)/2 units instead lets such runs straddle high-level split planes: their boxes span far-apart
    // corners and the walk does up to 2x the work.)
    const int u = (int)(pos / unit_rows);
    const int cu = u / kAlign;
    int clo = 0, chi = (n_units + kAlign - 1) / kAlign;
    unsigned id = 0;
    int l = 0;
    for (; l < level && l < top_levels; ++l) {
        if (chi - clo > 1) {
            const int mid = (clo + chi) >> 1;
            if (cu < mid) { chi = mid; id = 2 * id; }
            else { clo = mid; id = 2 * id + 1; }
        } else {
            id = 2 * id;
        }
    }
    for (int half = kAlign >> 1; l < level; ++l, half >>= 1) id = 2 * id + ((u & half) ? 1u : 0u);
    unsigned b = 0xFFFFFFFFu;
    if (row >= 0) {
        const float c = Cf[(int64_t)dim * n + row];          // (float) of the row's coordinate: coords_f32_kernel
        b = __float_as_uint(c);
        b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
        if (b == 0xFFFFFFFFu) b = 0xFFFFFFFEu;
        b >>= (32 - cbits);
        if (b == (0xFFFFFFFFu >> (32 - cbits))) b -= 1u;      // (the largest pattern is the padding rows')
    } else {
        b >>= (32 - cbits);
    }
    keys[pos] = (Key)(((Key)id << cbits) | (Key)b);
    vals[pos] = row;
}

// The LAST kAlignLevels levels of the tree only reorder rows inside one aligned group of kAlign units (2048 rows): one
// workgroup takes a group through all of them in LDS instead of six device-wide radix sorts.  Level l halves every node of
// the group by position after sorting the node's rows by coordinate (Ltop + l) % d -- exactly kd_key_kernel's bottom levels
// (id = 2 id + bit of the unit index).  Sort: a bitonic network over the group's 2048 slots, run independently on every
// node; keys are unique -- (order-preserving float bits) << 32 | position -- so rows at the same coordinate keep their
// order and the result does not depend on the network; padding rows (-1) and the slots past a partial last group carry the
// largest keys and stay at the end of their node.
constexpr int kGroupRows = kAlign * kPruneTileRows;      // 2048
// Round 5: the network runs in REGISTERS three stages at a time.  A thread holds eight slots spaced by the smallest distance
// j of the group of stages (j, 2j, 4j: all three compare-exchange partners of a slot are then among the thread's own eight),
// so the 251 stages of the six levels take 102 trips through LDS and as many barriers; only the keys travel (their low half
// is the slot they started the level in: the rows follow once per level).  Measured at C5: 1.41 -> 1.10 ms (with the float planes below).  (Staging the
// group's coordinates in LDS instead of re-reading a coordinate per row and level -- 80 KB, two workgroups per CU -- made it
// SLOWER, 2.0 ms: the kernel is bound by the trips through LDS and the barriers, not by those loads.)
constexpr int kBottomSlots = kGroupRows + kGroupRows / 32;          // one slot of padding per 32: eight slots a thread apart stay off one bank
__device__ __forceinline__ int bslot(int i) { return i + (i >> 5); }
__global__ __launch_bounds__(kThreads) void kd_bottom_kernel(int* __restrict__ perm, int64_t n, int64_t n_pad, int top_levels, const float* __restrict__ Cf, int d,
                                                             int64_t group0 = 0)
{
    static_assert(kGroupRows == 8 * kThreads, "eight slots per thread");
    __shared__ unsigned long long key[kBottomSlots];
    __shared__ int val[2][kGroupRows];
    const int64_t base = (group0 + (int64_t)blockIdx.x) * kGroupRows;
    const int m = (int)(n_pad - base < kGroupRows ? n_pad - base : kGroupRows);
    for (int i = threadIdx.x; i < kGroupRows; i += kThreads) val[0][i] = i < m ? perm[base + i] : -2;
    __syncthreads();
    int cur = 0;
    for (int l = 0; l < kAlignLevels; ++l) {
        const int dim = (top_levels + l) % d;
        const int ns = kGroupRows >> l;                  // node size at this level
        for (int i = threadIdx.x; i < kGroupRows; i += kThreads) {
            const int row = val[cur][i];
            unsigned b = 0xFFFFFFFFu;
            if (row >= 0) {
                b = __float_as_uint(Cf[(int64_t)dim * n + row]);
                b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
                if (b >= 0xFFFFFFFEu) b = 0xFFFFFFFDu;
            } else if (row == -1) {
                b = 0xFFFFFFFEu;                         // padding rows: behind every real row, before the slots that do not exist
            }
            key[bslot(i)] = ((unsigned long long)b << 32) | (unsigned)i;
        }
        __syncthreads();
        // bitonic network over the group's slots, run independently on every node (every node ends ascending); keys are
        // unique, so the result does not depend on the network
        for (int k = 2; k <= ns; k <<= 1) {
            int j = k >> 1;
            while (j >= 1) {
                const int g = j >= 4 ? 3 : (j == 2 ? 2 : 1);       // stages of this trip: j, j/2, .. (j >> (g - 1))
                const int q = j >> (g - 1);
                const int b0 = (threadIdx.x / q) * (8 * q) + (threadIdx.x % q);
                unsigned long long e[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) e[u] = key[bslot(b0 + u * q)];
#define MCE_CE(A_, B_)                                                                                     \
                do {                                                                                       \
                    const bool up_ = (((b0 + (A_) * q) & k) == 0) || k == ns;                              \
                    const unsigned long long x_ = e[A_], y_ = e[B_];                                       \
                    const bool sw_ = (x_ > y_) == up_;                                                     \
                    e[A_] = sw_ ? y_ : x_;                                                                 \
                    e[B_] = sw_ ? x_ : y_;                                                                 \
                } while (0)
                if (g == 3) { MCE_CE(0, 4); MCE_CE(1, 5); MCE_CE(2, 6); MCE_CE(3, 7); }
                if (g >= 2) { MCE_CE(0, 2); MCE_CE(1, 3); MCE_CE(4, 6); MCE_CE(5, 7); }
                MCE_CE(0, 1); MCE_CE(2, 3); MCE_CE(4, 5); MCE_CE(6, 7);
#undef MCE_CE
#pragma unroll
                for (int u = 0; u < 8; ++u) key[bslot(b0 + u * q)] = e[u];
                __syncthreads();
                j >>= g;
            }
        }
        for (int i = threadIdx.x; i < kGroupRows; i += kThreads) val[cur ^ 1][i] = val[cur][(int)(unsigned)key[bslot(i)]];
        cur ^= 1;
        __syncthreads();
    }
    for (int i = threadIdx.x; i < m; i += kThreads) perm[base + i] = val[cur][i];
}

// Cf[i][row] = (float)P[row][i]: the coordinates the k-d keys are made of, one plane per dimension (round 5).  Every sort of the
// top levels and every level of the bottom kernel reads ONE coordinate of every row in an order that is random in the caller's
// array: from the row-major fp64 rows that is a 64-byte line per 8 useful bytes (10 M x 6: 0.19 ms per pass, 13 passes); from a
// 40 MB plane of floats the same gather mostly hits the L2 / the Infinity Cache.  Same float, same keys, same order.
__global__ __launch_bounds__(kThreads) void coords_f32_kernel(const double* __restrict__ P, int64_t n, int d, float* __restrict__ Cf)
{
    __shared__ float t[kThreads][kPruneMaxDim + 1];
    const int64_t r0 = (int64_t)blockIdx.x * kThreads;
    const int rows = (int)(n - r0 < kThreads ? n - r0 : kThreads);
    for (int e = threadIdx.x; e < rows * d; e += kThreads) t[e / d][e % d] = (float)P[r0 * d + e];        // coalesced read of the block's rows
    __syncthreads();
    if ((int)threadIdx.x < rows)
        for (int i = 0; i < d; ++i) Cf[(int64_t)i * n + r0 + threadIdx.x] = t[threadIdx.x][i];
}

__global__ __launch_bounds__(kThreads) void identity_perm_kernel(int64_t n, int64_t n_pad, int* __restrict__ perm)
{
    const int64_t pos = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (pos < n_pad) perm[pos] = pos < n ? (int)pos : -1;
}

// out[pos][:] = P[perm[pos]][:] for the n real rows (they occupy the first n sorted positions)
__global__ __launch_bounds__(kThreads) void gather_rows_kernel(const double* __restrict__ P, const int* __restrict__ perm, int64_t n, int d,
                                                               double* __restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (e >= n * d) return;
    const int64_t pos = e / d;
    const int c = (int)(e - pos * d);
    out[e] = P[(int64_t)perm[pos] * d + c];
}

// bounding boxes of the 32-row tiles, as floats rounded OUTWARD (still enclosing):
// tbox[t][0][i] = lo, tbox[t][1][i] = hi; tiles past the last row are empty (lo = +inf, hi = -inf)
__global__ __launch_bounds__(kThreads) void tile_box_kernel(const double* __restrict__ Ps, int64_t n, int d, int64_t ntiles,
                                                            float* __restrict__ tbox)
{
    const int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (e >= ntiles * d) return;
    const int64_t t = e / d;
    const int i = (int)(e - t * d);
    const int64_t r0 = t * kPruneTileRows;
    const int64_t r1 = (r0 + kPruneTileRows < n) ? r0 + kPruneTileRows : n;
    double lo = __builtin_huge_val(), hi = -__builtin_huge_val();
    for (int64_t r = r0; r < r1; ++r) {
        const double v = Ps[r * d + i];
        lo = fmin(lo, v);
        hi = fmax(hi, v);
    }
    tbox[(t * 2 + 0) * d + i] = __double2float_rd(lo);
    tbox[(t * 2 + 1) * d + i] = __double2float_ru(hi);
}

// tbT[c][s][i][t] = tbox[c*group + t][s][i]: lane t of a wave reads tile t of chunk c with unit stride
__global__ __launch_bounds__(kThreads) void transpose_tile_box_kernel(const float* __restrict__ tbox, int64_t ntiles, int d, int group,
                                                                      float* __restrict__ tbT)
{
    const int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (e >= ntiles * 2 * d) return;
    const int64_t t = e / (2 * d);
    const int si = (int)(e - t * 2 * d);                 // s*d + i
    const int64_t c = t / group;
    const int tt = (int)(t - c * group);
    tbT[(c * 2 * d + si) * group + tt] = tbox[e];
}

// box of `group` consecutive tiles (a staging chunk / a query block)
__global__ __launch_bounds__(kThreads) void group_box_kernel(const float* __restrict__ tbox, int64_t ntiles, int d, int group, int ngroups,
                                                             float* __restrict__ gbox)
{
    const int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (e >= (int64_t)ngroups * d) return;
    const int64_t g = e / d;
    const int i = (int)(e - g * d);
    float lo = __builtin_huge_valf(), hi = -__builtin_huge_valf();
    for (int64_t t = g * group; t < (g + 1) * group && t < ntiles; ++t) {
        lo = fminf(lo, tbox[(t * 2 + 0) * d + i]);
        hi = fmaxf(hi, tbox[(t * 2 + 1) * d + i]);
    }
    gbox[(g * 2 + 0) * d + i] = lo;
    gbox[(g * 2 + 1) * d + i] = hi;
}

// Chunk list of one query block per workgroup: the lower bound on the squared distance between any point of block b and any
// point of chunk c (box to box, rounded DOWN to float; an empty box gives +inf), and the chunks ordered by BAND of that bound
// (prune.hpp: prune_band_floor) with a counting sort in LDS -- histogram, scan, scatter; the distances are recomputed for
// the scatter instead of stored, so any number of chunks fits.  Inside a band the order is whatever the atomics give: the
// search result does not depend on the order the chunks are visited in (the lists are exact, ties by row), only the
// stopping rule does, and that one reads band floors.
constexpr int kBandKeys = 1 << (8 + kPruneBandMantissa);      // exponent + mantissa bits of a non-negative float
__device__ __forceinline__ float box_dist2(const float* __restrict__ ql, const float* __restrict__ qh, const float* __restrict__ rl,
                                           const float* __restrict__ rh, int d)
{
    double s = 0.0;
    for (int i = 0; i < d; ++i) {
        const double g = fmax(0.0, fmax((double)ql[i] - (double)rh[i], (double)rl[i] - (double)qh[i]));     // NaN-free: empty boxes give +inf
        s = fma(g, g, s);
    }
    return __double2float_rd(s * (1.0 - 1e-12));
}
// Round 5 (lists of up to kListStage chunks: 16 M reference rows): every box distance is evaluated ONCE and kept in LDS; the
// bands are counted in kListBins bins reaching DOWN from the block's largest finite distance (the 8192 possible bands of a
// float were zeroed, scanned and re-armed per block: most of the kernel's 1.33 ms at C5 -- 19 532 blocks x 4 883 chunks -- was
// that fixed work, not the distances); what lies more than kListBins - 3 bands (2^31 in the squared distance) below the
// largest is listed with the bound 0 -- a valid lower bound that keeps the floors ascending; the scatter goes to LDS and the
// list leaves in coalesced stores.  Longer lists take chunk_list_big_kernel (the round-4 kernel).
constexpr int kListStage = 8192;
constexpr int kListBins = 1024;
__host__ __device__ constexpr size_t chunk_list_lds_bytes(int nchunk) { return (size_t)nchunk * 8; }
__global__ __launch_bounds__(kThreads) void chunk_list_kernel(const float* __restrict__ qbox, const float* __restrict__ rbox, int nchunk, int d,
                                                              float* __restrict__ out_d, int* __restrict__ out_c)
{
    static_assert(kListBins == 4 * kThreads, "scan: four bins per thread");
    __shared__ int cnt[kListBins];
    __shared__ float qb[2 * kPruneMaxDim];
    __shared__ int wsum[kThreads / 64];
    __shared__ int wmax[kThreads / 64];
    extern __shared__ __attribute__((aligned(16))) char cl_raw[];
    float* const sv = reinterpret_cast<float*>(cl_raw);          // [nchunk] box distances
    int* const sc = reinterpret_cast<int*>(cl_raw) + nchunk;     // [nchunk] chunk ids in list order
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < kListBins; i += kThreads) cnt[i] = 0;
    if (threadIdx.x < 2 * d) qb[threadIdx.x] = qbox[(int64_t)b * 2 * d + threadIdx.x];
    __syncthreads();
    // band of a distance: v >= 0 or +inf (a NaN sample makes the box distance NaN: taken as +inf)
    auto band_of = [](float v) { return (int)((__float_as_uint(v) & 0x7fffffffu) >> (23 - kPruneBandMantissa)); };
    constexpr int kInfBand = 0x7f800000 >> (23 - kPruneBandMantissa);
    int kmax = 0;
    for (int c = threadIdx.x; c < nchunk; c += kThreads) {
        float v = box_dist2(qb, qb + d, rbox + (int64_t)c * 2 * d, rbox + (int64_t)c * 2 * d + d, d);
        if (!(v < __builtin_huge_valf())) v = __builtin_huge_valf();
        sv[c] = v;
        const int k = band_of(v);
        kmax = (k < kInfBand && k > kmax) ? k : kmax;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int t = __shfl_xor(kmax, o, 64); kmax = t > kmax ? t : kmax; }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = kmax;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) kmax = wmax[w] > kmax ? wmax[w] : kmax;
    // bins: 0 = bound 0 (exact zeros and everything below the window), 1 .. kListBins - 2 = the bands klo .. kmax, kListBins - 1 = +inf
    const int klo = kmax - (kListBins - 3);
    auto bin_of = [&](float v) { const int k = band_of(v); return k >= kInfBand ? kListBins - 1 : (k < klo || v == 0.0f ? 0 : k - klo + 1); };
    for (int c = threadIdx.x; c < nchunk; c += kThreads) atomicAdd(&cnt[bin_of(sv[c])], 1);
    __syncthreads();
    // exclusive scan of the counters: four per thread, then the thread totals
    int loc[4], tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { loc[i] = tot; tot += cnt[threadIdx.x * 4 + i]; }
    int inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if ((threadIdx.x & 63) >= o) inc += v; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    int base = inc - tot;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
#pragma unroll
    for (int i = 0; i < 4; ++i) cnt[threadIdx.x * 4 + i] = base + loc[i];
    __syncthreads();
    for (int c = threadIdx.x; c < nchunk; c += kThreads) sc[atomicAdd(&cnt[bin_of(sv[c])], 1)] = c;
    __syncthreads();
    float* od = out_d + (int64_t)b * nchunk;
    int* oc = out_c + (int64_t)b * nchunk;
    for (int i = threadIdx.x; i < nchunk; i += kThreads) {
        const int c = sc[i];
        const float v = sv[c];
        oc[i] = c;
        od[i] = bin_of(v) == 0 ? 0.0f : v;
    }
}

__global__ __launch_bounds__(kThreads) void chunk_list_big_kernel(const float* __restrict__ qbox, const float* __restrict__ rbox, int nchunk, int d,
                                                                  float* __restrict__ out_d, int* __restrict__ out_c)
{
    __shared__ int cnt[kBandKeys];
    __shared__ float qb[2 * kPruneMaxDim];
    __shared__ int wsum[kThreads / 64];
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < kBandKeys; i += kThreads) cnt[i] = 0;
    if (threadIdx.x < 2 * d) qb[threadIdx.x] = qbox[(int64_t)b * 2 * d + threadIdx.x];
    __syncthreads();
    // v >= 0 or +inf: < kBandKeys.  A NaN sample makes the box distance NaN (sign bit possibly set): clamped into the last band
    // (the +inf one) instead of indexing past the histogram.
    auto key_of = [](float v) { return min((int)((__float_as_uint(v) & 0x7fffffffu) >> (23 - kPruneBandMantissa)), kBandKeys - 1); };
    for (int c = threadIdx.x; c < nchunk; c += kThreads)
        atomicAdd(&cnt[key_of(box_dist2(qb, qb + d, rbox + (int64_t)c * 2 * d, rbox + (int64_t)c * 2 * d + d, d))], 1);
    __syncthreads();
    // exclusive scan of the kBandKeys counters: each thread its kBandKeys / kThreads consecutive ones, then the thread totals
    constexpr int PER = kBandKeys / kThreads;
    int loc[PER], tot = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) { loc[i] = tot; tot += cnt[threadIdx.x * PER + i]; }
    int inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if ((threadIdx.x & 63) >= o) inc += v; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    int base = inc - tot;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) cnt[threadIdx.x * PER + i] = base + loc[i];
    __syncthreads();
    float* od = out_d + (int64_t)b * nchunk;
    int* oc = out_c + (int64_t)b * nchunk;
    for (int c = threadIdx.x; c < nchunk; c += kThreads) {
        const float v = box_dist2(qb, qb + d, rbox + (int64_t)c * 2 * d, rbox + (int64_t)c * 2 * d + d, d);
        const int pos = atomicAdd(&cnt[key_of(v)], 1);
        od[pos] = v;
        oc[pos] = c;
    }
}

// dispatch order of the WAVES (tpw query tiles each: the unit a workgroup of the walk serves): largest box first.  The sparse
// cells of the k-d order -- and any wave holding a far outlier, whose K-th neighbour distance sets the reach of the whole
// wave -- walk 10-30x farther than the rest: started last they would BE the tail of the launch.  (Round 3 ordered whole
// 512-query blocks by the block's box: one outlier makes that box large but only one of the block's eight waves slow.)
// Waves of padding queries only (they skip the walk) sort last.
__global__ __launch_bounds__(kThreads) void wave_cost_kernel(const float* __restrict__ tbox_q, int nwaves, int tpw, int64_t ntiles, int d,
                                                             float* __restrict__ keys, int* __restrict__ vals)
{
    const int g = blockIdx.x * kThreads + threadIdx.x;
    if (g >= nwaves) return;
    float s = 0.0f;
    bool any = false;
    for (int i = 0; i < d; ++i) {
        float lo = __builtin_huge_valf(), hi = -__builtin_huge_valf();
        for (int t = 0; t < tpw; ++t) {
            const int64_t tt = (int64_t)g * tpw + t;
            if (tt < ntiles) {
                lo = fminf(lo, tbox_q[(tt * 2 + 0) * d + i]);
                hi = fmaxf(hi, tbox_q[(tt * 2 + 1) * d + i]);
            }
        }
        const float w = hi - lo;
        if (w >= 0.0f && w < 3.0e38f) { s += w * w; any = true; }
    }
    keys[g] = any ? s : -1.0f;
    vals[g] = g;
}

// rocPRIM's radix sort of (key, row) pairs.  Above ~1 M keys it takes its "onesweep" path, which clears its histogram and
// look-back states with hipMemsetAsync -- and a memset NODE of a captured graph does not survive replay on this stack
// (zero_fill.hpp).  While the stream is being captured the merge-sort path is forced instead (no memset, same stable
// order: the result is the same permutation, only slower for very large inputs); eager calls keep rocPRIM's choice.
using MergeOnlySort = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, (size_t)-1>;

template <class Key>
size_t sort_pairs_tmp_bytes(int64_t n, unsigned end_bit)
{
    size_t a = 0, b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, a, (const Key*)nullptr, (Key*)nullptr, (const int*)nullptr, (int*)nullptr, (size_t)n, 0u, end_bit);
    (void)rocprim::radix_sort_pairs<MergeOnlySort>(nullptr, b, (const Key*)nullptr, (Key*)nullptr, (const int*)nullptr, (int*)nullptr, (size_t)n, 0u, end_bit);
    return std::max(a, b);
}

template <class Key>
hipError_t sort_pairs(void* tmp, size_t tmp_bytes, const Key* keys_in, Key* keys_out, const int* vals_in, int* vals_out, int64_t n, unsigned end_bit,
                      hipStream_t st)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (st) {                   // (the legacy stream cannot be captured)
        const hipError_t e = hipStreamIsCapturing(st, &cs);
        if (e != hipSuccess) return e;
    }
    size_t tb = tmp_bytes;
    if (cs != hipStreamCaptureStatusNone) return rocprim::radix_sort_pairs<MergeOnlySort>(tmp, tb, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, end_bit, st);
    return rocprim::radix_sort_pairs(tmp, tb, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, end_bit, st);
}

size_t sort_tmp_bytes(int64_t n) { return std::max(sort_pairs_tmp_bytes<unsigned long long>(n, 64u), sort_pairs_tmp_bytes<unsigned>(n, 32u)); }

// k-d order of P[n, d] in units of unit_rows; final permutation in `perm` ([n_pad])
// the node `part` of the 2^lv nodes at level lv of the aligned tree, as a range of groups (kd_key_kernel's bisection)
void kd_node_groups(int64_t n_units, int lv, int part, int64_t& glo, int64_t& ghi)
{
    glo = 0;
    ghi = (n_units + kAlign - 1) / kAlign;
    for (int l = 0; l < lv; ++l) {
        const int bit = (part >> (lv - 1 - l)) & 1;
        if (ghi - glo > 1) {
            const int64_t mid = (glo + ghi) >> 1;
            if (bit) glo = mid; else ghi = mid;
        } else if (bit) {
            glo = ghi;            // (a node that did not split: its second child is empty)
        }
    }
}

// part / nparts (round 6, a DISTRIBUTED preparation): with nparts = 2^lv > 1 ranks, rank `part` finishes only ITS subtree -- node
// `part` of level lv.  The sorts that settle the levels above lv run over all rows on every rank (they decide which rows the
// subtree holds); every later sort, and the bottom levels, only move rows INSIDE a node of level lv, so a rank runs them over its
// own position range [*seg_lo, *seg_hi) alone -- the same keys, the same stable radix sort: the range comes out exactly as the
// full sort would leave it, and the ranks' ranges put together (one all-reduce of the permutation, zeros outside the own range)
// ARE the single-GPU order.  *seg_lo = 0, *seg_hi = n_pad: nothing was left out (one rank, or a tree too shallow to split).
hipError_t kd_sort(const double* P, int64_t n, int64_t n_pad, int d, int unit_rows, int n_units, int* perm, unsigned long long* keys_a,
                   unsigned long long* keys_b, int* vals_b, float* Cf, void* tmp, size_t tmp_bytes, hipStream_t st, int part = 0, int nparts = 1,
                   int64_t* seg_lo = nullptr, int64_t* seg_hi = nullptr)
{
    if (seg_lo) *seg_lo = 0;
    if (seg_hi) *seg_hi = n_pad;
    hipLaunchKernelGGL(coords_f32_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, P, n, d, Cf);
    {
        const hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) return e0;
    }
    const int Ltop = top_tree_levels(n_units);
    const int L = Ltop + (n_units > 1 ? kAlignLevels : 0);
    const unsigned blocks = (unsigned)((n_pad + kThreads - 1) / kThreads);
    if (L == 0) {
        hipLaunchKernelGGL(identity_perm_kernel, dim3(blocks), dim3(kThreads), 0, st, n, n_pad, perm);
        return hipGetLastError();
    }
    // the top levels deal whole groups of kAlign units to the children: device-wide radix sorts on (node, coordinate) ...
    const bool bottom_in_lds = unit_rows == kPruneTileRows;
    const int Lradix = bottom_in_lds ? Ltop : L;
    if (Lradix == 0) {
        hipLaunchKernelGGL(identity_perm_kernel, dim3(blocks), dim3(kThreads), 0, st, n, n_pad, perm);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    // ONE sort per DIMENSION and round, not per level (round 5).  The interleaved tree splits dimension l % d at level l: 13
    // device-wide sorts for 10 M rows.  But a node of the top levels is a position range, and sorting a node's rows by
    // coordinate j places ALL the split planes of that dimension inside the node at once (they are ranks).  So the Ltop levels
    // are taken in two phases: the r = Ltop / d complete rounds of the cycle as ONE sort per dimension (sort j orders the nodes
    // as they stand after the dimensions before it, j r levels deep, by coordinate j, which settles r levels), then the
    // Ltop % d levels of the incomplete round one by one, as before -- d + Ltop % d sorts (7 at C5) instead of Ltop (13).
    // Every branch gets the same number of splits per dimension as in the interleaved tree and every node still holds
    // exactly its share of the rows; the cells differ only in that the r planes a dimension gets in phase one are
    // quantiles of the node it was sorted in rather than medians of the smaller cells the other dimensions cut out of it
    // in between.  (The last, incomplete round must stay last: with 4883 groups most level-12 nodes hold ONE group and do
    // not split again -- folded into the first phase as a third split of dimension 0, dimension 5 was left with one split
    // on those branches, the cells came out elongated and the walk multiplied 4 % more tiles: 65.9 -> 67.6 ms.)  The
    // search result does not depend on the order (the lists are exact).
    // (MCE_KD_TREE=interleaved, tools: the round-4 tree, one sort per level, for same-box comparisons)
    static const bool interleaved = [] { const char* e = getenv("MCE_KD_TREE"); return e && !strcmp(e, "interleaved"); }();
    int sdim[64], level_of[65] = {0}, nsort = 0;
    {
        const int rounds = interleaved ? 0 : Lradix / d;
        if (rounds > 0)
            for (int j = 0; j < d; ++j) { sdim[nsort] = j; level_of[nsort + 1] = level_of[nsort] + rounds; ++nsort; }
        for (int l = rounds * d; l < Lradix; ++l) { sdim[nsort] = l % d; level_of[nsort + 1] = level_of[nsort] + 1; ++nsort; }
    }
    // distributed: which node is this rank's, and from which sort on the rows stay inside it
    int lvW = 0;
    while ((1 << lvW) < nparts) ++lvW;
    const bool ranged = nparts > 1 && (1 << lvW) == nparts && bottom_in_lds && lvW <= Ltop && seg_lo && seg_hi;
    int64_t r_lo = 0, r_hi = n_pad, g_lo = 0, g_hi = (n_pad + kGroupRows - 1) / kGroupRows;
    int j0 = nsort;                 // first sort that runs on the own range only
    if (ranged) {
        kd_node_groups(n_units, lvW, part, g_lo, g_hi);
        r_lo = std::min<int64_t>(g_lo * kGroupRows, n_pad);
        r_hi = std::min<int64_t>(g_hi * kGroupRows, n_pad);
        for (int j = nsort; j >= 0; --j)
            if (level_of[j] >= lvW) j0 = j;
        *seg_lo = r_lo;
        *seg_hi = r_hi;
    }
    const int idbits = nsort > 0 ? level_of[nsort - 1] : 0;        // node id bits of the last sort
    // 32-bit keys where the node id leaves at least kMinCoordBits32 coordinate bits AND no dimension is split more than four
    // times by these levels (a dimension split often needs its quantiles placed finely: d = 1, 2 keep 64 bits)
    const int cb32 = 32 - idbits;
    const bool keys32 = Lradix > 0 && cb32 >= kMinCoordBits32 && (Lradix + d - 1) / d <= 4;
    for (int j = 0; j < nsort; ++j) {
        const int level = level_of[j];
        // (sorts from j0 on: this rank's position range only -- an empty range has nothing to do)
        const bool own = ranged && j >= j0;
        const int64_t p0 = own ? r_lo : 0, p1 = own ? r_hi : n_pad;
        if (p1 <= p0) continue;
        const unsigned kb_blocks = (unsigned)((p1 - p0 + kThreads - 1) / kThreads);
        hipError_t e;
        if (keys32) {
            unsigned* ka = reinterpret_cast<unsigned*>(keys_a);
            unsigned* kb = reinterpret_cast<unsigned*>(keys_b);
            hipLaunchKernelGGL(kd_key_kernel<unsigned>, dim3(kb_blocks), dim3(kThreads), 0, st, j == 0 ? (const int*)nullptr : perm, n, n_pad, unit_rows,
                               n_units, Ltop, level, Cf, d, sdim[j], cb32, ka, vals_b, p0, p1);
            if ((e = hipGetLastError()) != hipSuccess) return e;
            e = sort_pairs(tmp, tmp_bytes, (const unsigned*)ka + p0, kb + p0, (const int*)vals_b + p0, perm + p0, p1 - p0, (unsigned)(cb32 + level), st);
        } else {
            hipLaunchKernelGGL(kd_key_kernel<unsigned long long>, dim3(kb_blocks), dim3(kThreads), 0, st, j == 0 ? (const int*)nullptr : perm, n, n_pad, unit_rows,
                               n_units, Ltop, level, Cf, d, sdim[j], kCoordBits, keys_a, vals_b, p0, p1);
            if ((e = hipGetLastError()) != hipSuccess) return e;
            e = sort_pairs(tmp, tmp_bytes, (const unsigned long long*)keys_a + p0, keys_b + p0, (const int*)vals_b + p0, perm + p0, p1 - p0, (unsigned)(kCoordBits + level), st);
        }
        if (e != hipSuccess) return e;
    }
    // ... the last kAlignLevels stay inside a group: one pass through LDS (10 M rows: 6 x 0.72 ms of sorts -> one kernel)
    if (bottom_in_lds && L > Ltop) {
        const int64_t ng = ranged ? std::min<int64_t>(g_hi, (n_pad + kGroupRows - 1) / kGroupRows) - g_lo : (n_pad + kGroupRows - 1) / kGroupRows;
        if (ng > 0) {
            hipLaunchKernelGGL(kd_bottom_kernel, dim3((unsigned)ng), dim3(kThreads), 0, st, perm, n, n_pad, Ltop, Cf, d, ranged ? g_lo : (int64_t)0);
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

}  // namespace

int prune_layout(int64_t nq, int64_t nq_pad, int nqblk, int64_t nr, int64_t nr_pad, int64_t nchunk, int d, PruneLayout& L)
{
    const int64_t nmax = std::max(nq_pad, nr_pad);
    const int64_t pairs = (int64_t)nqblk * nchunk;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    L.perm_r = take((size_t)nr_pad * 4);
    L.perm_q = take((size_t)nq_pad * 4);
    L.keys_a = take((size_t)nmax * 8);
    L.keys_b = take((size_t)nmax * 8);
    L.vals_b = take((size_t)nmax * 4);
    L.Ys = take((size_t)nr * d * 8);
    L.Xs = take((size_t)nq * d * 8);
    // float planes of the coordinates while a set is being sorted (kd_sort): they live in the Xs region where it is large enough
    // (it is written only after the last sort), else in room of their own
    L.cf32 = (size_t)nq * d * 8 >= (size_t)std::max(nq, nr) * d * 4 ? L.Xs : take((size_t)std::max(nq, nr) * d * 4);
    L.tbox_r = take((size_t)(nr_pad / kPruneTileRows) * 2 * d * 4);
    L.tbox_q = take((size_t)(nq_pad / kPruneTileRows) * 2 * d * 4);
    L.tboxT_r = take((size_t)(nr_pad / kPruneTileRows) * 2 * d * 4);
    L.box_r = take((size_t)nchunk * 2 * d * 4);
    L.box_q = take((size_t)nqblk * 2 * d * 4);
    const size_t nwaves = (size_t)nqblk * kPruneWavesPerBlock;
    L.bkey_a = take(nwaves * 4);
    L.bkey_b = take(nwaves * 4);
    L.bval_a = take(nwaves * 4);
    L.border = take(nwaves * 4);
    L.list_d_b = take((size_t)pairs * 4);
    L.list_c_b = take((size_t)pairs * 4);
    size_t border_tmp = 0;
    (void)rocprim::radix_sort_pairs_desc(nullptr, border_tmp, (const float*)nullptr, (float*)nullptr, (const int*)nullptr, (int*)nullptr,
                                         nwaves, 0u, 32u);
    L.tmp_bytes = std::max(sort_tmp_bytes(nmax), border_tmp);
    L.tmp = take(L.tmp_bytes);
    L.total = off;
    return 0;
}

hipError_t prune_prepare_part(const double* dY, int64_t nr, int d, int64_t nr_pad, char* ws, const PruneLayout& L, int part, int nparts, hipStream_t st,
                              int64_t& seg_lo, int64_t& seg_hi)
{
    int* perm_r = reinterpret_cast<int*>(ws + L.perm_r);
    const int64_t ntile_r = nr_pad / kPruneTileRows;
    hipError_t e = kd_sort(dY, nr, nr_pad, d, kPruneTileRows, (int)ntile_r, perm_r, reinterpret_cast<unsigned long long*>(ws + L.keys_a),
                           reinterpret_cast<unsigned long long*>(ws + L.keys_b), reinterpret_cast<int*>(ws + L.vals_b), reinterpret_cast<float*>(ws + L.cf32),
                           ws + L.tmp, L.tmp_bytes, st, part, nparts, &seg_lo, &seg_hi);
    if (e != hipSuccess) return e;
    // zeros outside the own range: the ranks' permutations ADD up to the whole one (padding rows are -1 inside the range that holds them)
    if (seg_lo > 0 && (e = zero_async(perm_r, (size_t)seg_lo * sizeof(int), st)) != hipSuccess) return e;
    if (seg_hi < nr_pad && (e = zero_async(perm_r + seg_hi, (size_t)(nr_pad - seg_hi) * sizeof(int), st)) != hipSuccess) return e;
    return hipSuccess;
}

hipError_t prune_prepare(const double* dX, int64_t nq, const double* dY, int64_t nr, int d, bool same_set, int qpb, int chunk_rows,
                         int64_t nq_pad, int nqblk, int64_t nr_pad, int64_t nchunk, char* ws, const PruneLayout& L, hipStream_t st,
                         PruneOut& out, bool perm_ready)
{
    int* perm_r = reinterpret_cast<int*>(ws + L.perm_r);
    int* perm_q = reinterpret_cast<int*>(ws + L.perm_q);
    unsigned long long* keys_a = reinterpret_cast<unsigned long long*>(ws + L.keys_a);
    unsigned long long* keys_b = reinterpret_cast<unsigned long long*>(ws + L.keys_b);
    int* vals_b = reinterpret_cast<int*>(ws + L.vals_b);
    double* Ys = reinterpret_cast<double*>(ws + L.Ys);
    double* Xs = reinterpret_cast<double*>(ws + L.Xs);
    float* cf32 = reinterpret_cast<float*>(ws + L.cf32);
    float* tbox_r = reinterpret_cast<float*>(ws + L.tbox_r);
    float* tbox_q = reinterpret_cast<float*>(ws + L.tbox_q);
    float* tboxT_r = reinterpret_cast<float*>(ws + L.tboxT_r);
    float* box_r = reinterpret_cast<float*>(ws + L.box_r);
    float* box_q = reinterpret_cast<float*>(ws + L.box_q);
    float* list_d_b = reinterpret_cast<float*>(ws + L.list_d_b);
    int* list_c_b = reinterpret_cast<int*>(ws + L.list_c_b);
    void* tmp = ws + L.tmp;
    const int64_t ntile_r = nr_pad / kPruneTileRows, ntile_q = nq_pad / kPruneTileRows;
    auto blocks_for = [](int64_t n) { return dim3((unsigned)((n + kThreads - 1) / kThreads)); };

    // references: k-d order down to single 32-row tiles, reordered copy, tile and chunk boxes
    // (perm_ready: the order is in perm_r already -- prune_prepare_part on every rank + the all-reduce of the permutation)
    hipError_t e = perm_ready ? hipSuccess : kd_sort(dY, nr, nr_pad, d, kPruneTileRows, (int)ntile_r, perm_r, keys_a, keys_b, vals_b, cf32, tmp, L.tmp_bytes, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gather_rows_kernel, blocks_for(nr * d), dim3(kThreads), 0, st, dY, perm_r, nr, d, Ys);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(tile_box_kernel, blocks_for(ntile_r * d), dim3(kThreads), 0, st, Ys, nr, d, ntile_r, tbox_r);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(group_box_kernel, blocks_for(nchunk * d), dim3(kThreads), 0, st, tbox_r, ntile_r, d, chunk_rows / kPruneTileRows, (int)nchunk, box_r);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (same_set) {
        out.Xs = Ys;
        out.qperm = perm_r;          // query tile t IS reference tile t
        out.tbox_q = tbox_r;
    } else {
        e = kd_sort(dX, nq, nq_pad, d, kPruneTileRows, (int)ntile_q, perm_q, keys_a, keys_b, vals_b, cf32, tmp, L.tmp_bytes, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(gather_rows_kernel, blocks_for(nq * d), dim3(kThreads), 0, st, dX, perm_q, nq, d, Xs);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        hipLaunchKernelGGL(tile_box_kernel, blocks_for(ntile_q * d), dim3(kThreads), 0, st, Xs, nq, d, ntile_q, tbox_q);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        out.Xs = Xs;
        out.qperm = perm_q;
        out.tbox_q = tbox_q;
    }
    hipLaunchKernelGGL(transpose_tile_box_kernel, blocks_for(ntile_r * 2 * d), dim3(kThreads), 0, st, tbox_r, ntile_r, d,
                       chunk_rows / kPruneTileRows, tboxT_r);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    out.tbox_r = tboxT_r;
    out.cbox_r = box_r;
    hipLaunchKernelGGL(group_box_kernel, blocks_for((int64_t)nqblk * d), dim3(kThreads), 0, st, out.tbox_q, same_set ? ntile_r : ntile_q, d,
                       qpb / kPruneTileRows, nqblk, box_q);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    static_assert(kBandKeys % kThreads == 0 && kPruneMaxDim * 2 <= kThreads, "chunk_list_kernel geometry");
    if (nchunk <= kListStage) {
        static bool attr_set[16] = {};          // per device; benign race: idempotent
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 16 || !attr_set[dev]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(chunk_list_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)chunk_list_lds_bytes(kListStage));
            if (e != hipSuccess) return e;
            if (dev < 16) attr_set[dev] = true;
        }
        hipLaunchKernelGGL(chunk_list_kernel, dim3((unsigned)nqblk), dim3(kThreads), chunk_list_lds_bytes((int)nchunk), st, box_q, box_r, (int)nchunk, d, list_d_b, list_c_b);
    } else {
        hipLaunchKernelGGL(chunk_list_big_kernel, dim3((unsigned)nqblk), dim3(kThreads), 0, st, box_q, box_r, (int)nchunk, d, list_d_b, list_c_b);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    {
        float* bkey_a = reinterpret_cast<float*>(ws + L.bkey_a);
        float* bkey_b = reinterpret_cast<float*>(ws + L.bkey_b);
        int* bval_a = reinterpret_cast<int*>(ws + L.bval_a);
        int* border = reinterpret_cast<int*>(ws + L.border);
        const int nwaves = nqblk * kPruneWavesPerBlock;
        const int tpw = qpb / kPruneTileRows / kPruneWavesPerBlock;
        hipLaunchKernelGGL(wave_cost_kernel, dim3((unsigned)((nwaves + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, out.tbox_q, nwaves, tpw,
                           same_set ? ntile_r : ntile_q, d, bkey_a, bval_a);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        size_t tb2 = L.tmp_bytes;
        // (the keys are non-negative floats or -1: as unsigned bit patterns -1 would sort FIRST in descending order -- the
        //  float comparison of rocPRIM's radix sort handles the sign)
        e = rocprim::radix_sort_pairs_desc(tmp, tb2, (const float*)bkey_a, bkey_b, (const int*)bval_a, border, (size_t)nwaves, 0u, 32u, st);
        if (e != hipSuccess) return e;
        out.border = border;
    }
    out.Ys = Ys;
    out.rperm = perm_r;
    out.clist = list_c_b;
    out.cdist = list_d_b;
    return hipSuccess;
}

// ---------------------------------------------------------------------------
// symmetric sweep: order by distance from the mean
// ---------------------------------------------------------------------------
namespace {

// key = order-preserving bits of (float)|y - c|^2 (non-negative: the bit pattern itself); padding sorts last.
// One row per thread, the block's rows staged through LDS: the global reads are contiguous (a thread reading its own
// row straight from memory strides by D doubles: 0.40 ms at 1M x 27, this 0.1).
constexpr int kNormRows = 256;
__global__ __launch_bounds__(kNormRows) void norm_key_kernel(const double* __restrict__ Y, int64_t n, int64_t n_pad, int d,
                                                             const double* __restrict__ center, unsigned* __restrict__ keys,
                                                             int* __restrict__ vals)
{
    extern __shared__ double tile[];                 // [kNormRows][d + 1]: the odd stride keeps the row-wise reads off one bank
    const int64_t r0 = (int64_t)blockIdx.x * kNormRows;
    const int64_t rows = n - r0 < kNormRows ? (n - r0 > 0 ? n - r0 : 0) : kNormRows;
    const int ld = d + 1;
    for (int64_t e = threadIdx.x; e < rows * d; e += kNormRows) {
        const int r = (int)(e / d), c = (int)(e - (int64_t)r * d);
        tile[r * ld + c] = Y[r0 * d + e] - center[c];
    }
    __syncthreads();
    const int64_t pos = r0 + threadIdx.x;
    if (pos >= n_pad) return;
    unsigned k = 0xFFFFFFFFu;
    if (pos < n) {
        double s = 0.0;
        for (int i = 0; i < d; ++i) {
            const double t = tile[threadIdx.x * ld + i];
            s = fma(t, t, s);
        }
        k = __float_as_uint((float)s);
        if (k > 0xFFFFFFFEu || s != s) k = 0xFFFFFFFEu;      // (NaN rows: anywhere, the search rejects them elsewhere)
    }
    keys[pos] = k;
    vals[pos] = pos < n ? (int)pos : -1;
}

}  // namespace

int sym_layout(int64_t n, int64_t n_pad, int nqblk, int d, int kcap, int qpb, int per_row, SymLayout& L)
{
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    L.perm = take((size_t)n_pad * 4);
    L.keys_a = take((size_t)n_pad * 4);
    L.keys_b = take((size_t)n_pad * 4);
    L.vals_a = take((size_t)n_pad * 4);
    L.Ys = take((size_t)n * d * 8);
    L.thr = take((size_t)n_pad * 8);
    L.rrow = take((size_t)n_pad * 4);
    L.rtile = take((size_t)(n_pad / 32) * 4);
    L.slots = take((size_t)n_pad * kcap * 8);
    L.bucket_cnt = take((size_t)nqblk * 4 * 3);
    L.bucket_flag = L.bucket_cnt + (size_t)nqblk * 4;
    L.done = L.bucket_flag + (size_t)nqblk * 4;
    L.cap = per_row * qpb;
    L.bucket = take((size_t)nqblk * L.cap * 16);
    L.tmp_bytes = sort_pairs_tmp_bytes<unsigned>(n_pad, 32u);
    L.tmp = take(L.tmp_bytes);
    L.total = off;
    return 0;
}

hipError_t sym_prepare(const double* dY, int64_t n, int d, const double* center, int64_t n_pad, char* ws, const SymLayout& L, hipStream_t st)
{
    int* perm = reinterpret_cast<int*>(ws + L.perm);
    unsigned* keys_a = reinterpret_cast<unsigned*>(ws + L.keys_a);
    unsigned* keys_b = reinterpret_cast<unsigned*>(ws + L.keys_b);
    int* vals_a = reinterpret_cast<int*>(ws + L.vals_a);
    double* Ys = reinterpret_cast<double*>(ws + L.Ys);
    const unsigned blocks = (unsigned)((n_pad + kNormRows - 1) / kNormRows);
    hipLaunchKernelGGL(norm_key_kernel, dim3(blocks), dim3(kNormRows), (size_t)kNormRows * (d + 1) * sizeof(double), st, dY, n, n_pad, d, center,
                       keys_a, vals_a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // (stable: rows at the same distance keep the caller's order -- the permutation is deterministic)
    e = sort_pairs(ws + L.tmp, L.tmp_bytes, (const unsigned*)keys_a, keys_b, (const int*)vals_a, perm, n_pad, 32u, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n * d + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, dY, perm, n, d, Ys);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    e = zero_async(ws + L.bucket_cnt, (size_t)3 * (L.bucket_flag - L.bucket_cnt), st);      // counts | flags | done
    return e;
}

}  // namespace mce
