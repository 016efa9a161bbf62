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
#include <type_traits>

#include "sym_types.hpp"
#include "prune.hpp"        // prune_band_floor: the order of the pruned walk's chunk lists

#ifndef MCE_ABLATE
#define MCE_ABLATE 0   // tools/knn_f16_bench.hip only: 1 = the gate never passes (no candidates), 2 = no gate at all (results invalid).  The other
                       // ablation builds of rounds 1-4 (no barriers, no LDS reads, two-level gate folding: profiles/r04_final/kst1_ablation.txt)
                       // are closed and gone from the source.
#endif

namespace mce {

// v_min3_f32 without the NaN-canonicalising v_max the compiler adds around fminf()
// Barrier that publishes a chunk staged by LDS-DMA (global_load_lds).  The landing of a DMA is ordered for OTHER waves only
// by the issuing wave's vmcnt wait FOLLOWED by the workgroup barrier -- and __syncthreads() by itself waits for LDS
// traffic (lgkmcnt) only: the compiler emits vmcnt(0) before a barrier just when other code around it happens to need it.
// The seed loop's barrier had none (ISA: "s_waitcnt lgkmcnt(0); s_barrier"): with several searches sharing the chip -- the
// batched entry point -- a wave could read a buffer another wave's DMA was still filling, and a seed bound computed from
// half-landed rows cut off true neighbours (300 small chains: ~2 wrong sums per batch call; none since).
__device__ __forceinline__ void dma_barrier()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

__device__ __forceinline__ float min3f(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// Workgroup geometry: 8 waves (2 per SIMD) x 2 query tiles.  Measured and dropped (the builds are gone from this file, the
// numbers are in DESIGN.md / profiles/): 4 waves x 4 tiles, one wave per SIMD (83 vs 56 ms at C3, round 1); 16 waves, four per
// SIMD (spills); 12 waves, three per SIMD, at one k-step and K <= 4 (42.7 vs 41.3 ms at C4, round 5).  What did pay at one
// k-step and K <= 4 is 8 waves x FOUR tiles: the kernel's QTT parameter.
constexpr int kHWaves = 8;      // waves per workgroup (two per SIMD)
constexpr int kHQT = 2;         // 32-query tiles per wave (the wide exhaustive sweep: four -- template parameter QTT of the kernel)
constexpr int kHNL = kHQT / 2;                     // top-K lists per owner lane (64 queries per list set)
constexpr int kHThreads = kHWaves * 64;
#ifndef MCE_H_QUEUE
#define MCE_H_QUEUE 448
#endif
#ifndef MCE_H_SEED_ROWS
#define MCE_H_SEED_ROWS 24576  // seed phase: reference rows swept twice (0 = no seed phase; 16 k - 48 k rows: within 1 % at 0.1 - 1 M rows) ...
#endif
#ifndef MCE_H_SEED_SHARE
#define MCE_H_SEED_SHARE 4     // ... at most 1/4 of the split's chunks (binds below ~65 k rows per split; capi_search.hpp: seed_cfg) ...
#endif
#ifndef MCE_H_SEED_TG
#define MCE_H_SEED_TG 8        // ... in groups of 8 tiles (256 rows)
#endif
#ifndef MCE_H_TRIGGER
#define MCE_H_TRIGGER 96
#endif
#ifndef MCE_PRUNE_PROF
#define MCE_PRUNE_PROF 0        // developer builds (tools/build_variant.sh x -DMCE_PRUNE_PROF=n, tools/prune_prof.py): per-wave cycle counts of the
#endif                          // pruned walk's phases (1), of the walk phase's parts (2), of the drains inside gate_exact (3)
#ifndef MCE_H_PRUNE_WAVES
#define MCE_H_PRUNE_WAVES 2     // pruned walk: waves per SIMD the register allocation aims for
#endif
// Pruned walk, LDS per wave: tiles multiplied per batch (slice = BATCH KB), queue entries, drain trigger.  A third wave
// per SIMD needs <= 168 VGPRs AND <= 13.3 KB of LDS per wave: lists of up to 9 entries get there with half the batch and
// half the queue (round 4: 132 / 156 / 162 VGPRs for 4 / 8 / 9 entries, no scratch, since the tiles are staged by LDS-DMA
// and the loop-invariant values the compiler parked in registers are computed where they are used) and run 20-30 % faster
// for it (10 M x 6, K = 9: 96 vs 125 ms); twelve and sixteen entries (182 / 211 VGPRs) stay at two waves.
#ifndef MCE_H_PRUNE_SMALL
#define MCE_H_PRUNE_SMALL 10     // largest list capacity (entries held in registers) on the three-wave configuration (eleven spill inside the walk)
#endif
#ifndef MCE_H_PRUNE_SMALL_WAVES
#define MCE_H_PRUNE_SMALL_WAVES 3
#endif
#ifndef MCE_H_PRUNE_B10
#define MCE_H_PRUNE_B10 4       // tiles per batch with 9 list entries
#endif
#ifndef MCE_H_PRUNE_Q10
#define MCE_H_PRUNE_Q10 128
#endif
#ifndef MCE_H_PRUNE_T10
#define MCE_H_PRUNE_T10 32
#endif
__host__ __device__ constexpr int f16_prune_batch(int KCAP) { return KCAP <= 8 ? 4 : (KCAP <= MCE_H_PRUNE_SMALL ? MCE_H_PRUNE_B10 : 8); }
__host__ __device__ constexpr int f16_prune_queue(int KCAP) { return KCAP <= 8 ? 128 : (KCAP <= MCE_H_PRUNE_SMALL ? MCE_H_PRUNE_Q10 : 256); }   // entries are already exact: only the list insertion is deferred
__host__ __device__ constexpr int f16_prune_trigger(int KCAP) { return KCAP <= 8 ? 32 : (KCAP <= MCE_H_PRUNE_SMALL ? MCE_H_PRUNE_T10 : 48); }
#ifndef MCE_H_STAGE_KB
#define MCE_H_STAGE_KB 48
#endif
constexpr int kHQueue = MCE_H_QUEUE;          // candidate queue entries per wave (16 B each in LDS)
constexpr int kHDrainTrigger = MCE_H_TRIGGER;    // a wave with this many queued candidates asks the workgroup to drain
constexpr int kHRelBits = 26;   // queue entry = query-local (6|7 bits) << kHRelBits | row - first row of the split
constexpr int kHSymRowBits = kHRelBits - 1;            // symmetric sweep: the row field's top bit says "this lane passed the ROW gate"
constexpr double kHTargetRadius = 200.0;
constexpr int kPruneDims = 15;                // pruned walk: largest d (KST = 1)

// device-side scalars shared by the f16 kernels (doubles; maxima kept as bit patterns)
enum { HP_RMAX = 0, HP_SCALE = 1, HP_EY = 2, HP_YHATMAX = 3, HP_RHO = 4, HP_STAT_CHUNKS = 5, HP_STAT_TILES = 6, HP_COUNT = 16 };   // STAT_*: pruned walk, totals over the launch

// |y^|^2 rides in the k dimension as 1..3 fp16 pieces (x' carries matching ones): three pieces (33 bits) when
// they fit the padding of the last 16-wide k-step, fewer when that saves a whole k-step -- d = 14, 15, 30, 31,
// 46, 47, 62, 63.  What the pieces miss (<= 2^-11 |y^|^2 <= 20 with one piece; thresholds there are in the
// thousands) is measured while packing (HP_RHO) and given back to the gate, so the filter stays rigorous.
__host__ __device__ constexpr int f16_ksteps(int D) { return (D + 1 + 15) / 16; }          // 16-wide k-steps
__host__ __device__ constexpr int f16_norm_pieces(int D) { return 16 * f16_ksteps(D) - D < 3 ? 16 * f16_ksteps(D) - D : 3; }
__host__ __device__ constexpr int f16_qt(int) { return kHQT; }
__host__ __device__ constexpr bool f16_supported(int D, int K) { return D >= 1 && f16_ksteps(D) <= 4 && K <= 32; }   // K > 16: two passes
__host__ __device__ constexpr int f16_qpb(int KCAP) { return kHWaves * f16_qt(KCAP) * 32; }
// 32-row reference tiles per LDS chunk (tile = KST KB): MCE_H_STAGE_KB per buffer, even count,
// and a whole number of 16-byte vectors per thread (CT*KST % 8 == 0)
__host__ __device__ constexpr int f16_chunk_tiles(int KST) { return MCE_H_STAGE_KB / KST; }   // 48 KB: 48, 24, 16, 12 tiles
// seed phase of the exhaustive sweep (see the kernel): about `rows` reference rows of the split, at most 1/share of
// its `cps` chunks, in groups of `tg` 32-row tiles -- and only if that makes at least twice the `kneed` = K (+1 with
// self-exclusion) groups the bound needs.  Returns chunks | tg << 16, or 0 for no seed phase.
inline int f16_seed_cfg(int64_t cps, int CT, int kneed, int rows = MCE_H_SEED_ROWS, int share = MCE_H_SEED_SHARE, int tg = MCE_H_SEED_TG)
{
    if (rows <= 0 || share <= 0 || tg <= 0 || kneed <= 0) return 0;
    int64_t chunks = (rows + CT * 32 - 1) / (CT * 32);
    if (chunks > cps / share) chunks = cps / share;
    if (chunks > 0xffff) chunks = 0xffff;
    if (chunks * CT / tg < 2 * (int64_t)kneed) return 0;
    return (int)chunks | (tg << 16);       // (bits 28-29: where the chunks are, set by the caller -- see the seed phase)
}
__host__ __device__ constexpr size_t f16_lds_bytes(int KST, int KCAP, bool sym = false, int qt = kHQT)
{
    return (size_t)2 * f16_chunk_tiles(KST) * KST * 1024             // staging
           + (size_t)kHWaves * kHQueue * 16                            // queues: packed(4) + next(4) + d2(8)
           + (size_t)kHWaves * qt * 32 * 4 + 128                       // chain heads + votes + block thresholds (pruned walk)
           + (sym ? (size_t)kHWaves * qt * 32 * 8 : 0);                // symmetric sweep: K-th bound per query as of the last drain
}

// ---------------------------------------------------------------------------
// Symmetric sweep (auto evidence: the queries ARE the reference rows, one buffer).
// d(i,j) = d(j,i): every 32x32 MFMA tile of the exhaustive sweep is computed twice, once with i as the query and
// once with j.  Here it is computed ONCE and gated for both sides:
//   * the rows are sorted by distance from the mean (sym_prepare), so the 32 rows of a tile have nearly the same
//     K-th neighbour distance; a prepass (SYM = 1: the seed phase of the exhaustive kernel as its own launch) gives
//     EVERY row an upper bound on its K-th distance before any block runs;
//   * query block a (512 rows) sweeps only the rows of the blocks 0..a: the pair of blocks {a, b}, b < a, is handled by
//     a alone.  Blocks are dispatched in index order, so every row a block meets belongs to a block that has started
//     before it -- usually finished -- and has published tight bounds.  (Measured and dropped: a RING, every block
//     handling the n/2 blocks behind it, wrapping around -- equal work per block, but the first blocks to run then
//     meet rows nobody has bounded yet: 362 exact evaluations per query there instead of 45, 88 ms instead of 48.
//     The triangle's price is blocks of unequal length: the last round of workgroups is the longest, ~13 % of tail.
//     Cutting the long blocks into parts with their own lists balances that but every part re-discovers the
//     bounds within its share of the rows: 22 evaluations per query and part, no net gain.)
//   * column side (lane = query i): as before, min of the lane's 16 accumulators <= G_i;
//     row side (the 32 streamed rows j): a pair can be among j's K nearest only if A[i,j] <= R_j + c_i with
//     R_j = (s sqrt(thr_j) + e_j + max e)^2 and c_i = eps_i - |x^_i|^2; the gate tests the lane's minimum against
//     the TILE's largest R_j (rtile, fetched one chunk ahead, wave-uniform): + 2 VALU per tile and query tile;
//   * a pair that passes either gate is evaluated exactly once (phase A).  For the column side it joins query i's
//     chain as before.  For the row side (phase R, one lane per queued pair) it must beat thr[j], then goes through
//     j's K global SLOTS (the K smallest row-side distances so far, lock-free: replace-the-maximum by
//     compare-and-swap) -- whose K-th tightens thr[j] / rrow[j] / rtile for everybody who meets the row later -- and
//     is appended to the BUCKET of j's block.  sym_merge_kernel folds the buckets into the
//     lists afterwards.  A bucket that overflows flags its block; a repair launch (SYM = 3) then searches the flagged
//     blocks exhaustively, so the result never depends on the bucket size.
// Everything published is a valid upper bound at all times and only ever shrinks, so stale reads merely let more
// candidates through.  Lists carry the caller's row numbers and ties break on them: results are bit-identical to
// the exhaustive sweep's.
// ---------------------------------------------------------------------------
// R_j from the bound thr on row j's K-th squared distance and its conversion error ex (same error terms as the
// column gate, see gate_of); inflated by what the gate's fp32 addition R + c can lose
__device__ __forceinline__ float sym_row_gate(double thr, double ex, const double* __restrict__ params, int KST)
{
    if (!(thr < __builtin_huge_val())) return __builtin_huge_valf();
    const double s2 = params[HP_SCALE] * params[HP_SCALE];
    const double ga = (ex + params[HP_EY]) * (1.0 + 1e-9) + 2.0 * sqrt(16.0 * KST) * 0x1p-14;
    const double rr = sqrt(thr * s2) * (1.0 + 1e-12) + ga;
    const double g = rr * rr * (1.0 + 1e-12);
    const double ym = params[HP_YHATMAX];
    return __double2float_ru(g * (1.0 + 0x1p-22) + 0x1p-22 * (ym * ym + 1.0) + 1e-30);
}

__host__ __device__ constexpr int f16_prune_slice_bytes(int KST, int KCAP) { return f16_prune_batch(KCAP) * KST * 1024 + 256; }   // kBatch tiles + pending ids
#ifndef MCE_H_PRUNE_BOOT
#define MCE_H_PRUNE_BOOT 12
#endif
constexpr int kHPruneBoot = MCE_H_PRUNE_BOOT;   // pruned walk: k-d neighbour tiles on either side multiplied before the walk
constexpr int kHPruneChunkTiles = 64; // pruned walk: tiles per list entry ("chunk" = 2048 rows, an aligned k-d subtree)
// [tile slice + pending ids][queue d2 | row | next][heads][the wave's 64 fp64 query rows][one fp64 reference tile][its caller row numbers]
__host__ __device__ constexpr size_t f16_prune_lds_bytes(int KST, int D, int KCAP)
{
    // (the exact evaluation and the per-query reach test may read ONE row past the d rows of the query block / the
    //  reference tile when d is odd: 512 / 256 bytes that the regions after them always cover)
    return (size_t)f16_prune_slice_bytes(KST, KCAP) + (size_t)f16_prune_queue(KCAP) * 16 + (size_t)kHQT * 32 * 4 + 128 +
           (size_t)(kHQT * 32 + 32) * D * 8 + 128 + (size_t)kHQT * 32 * 8 + 64 * 8;
}

// ---------------------------------------------------------------------------
// the filter + exact-refine search kernel
//   grid.x = nqblk * rsplit (as knn_mfma_kernel); lists out: part_d/part_i [rsplit][KCAP][nq_pad]
//   with EXACT squared distances as keys.
// ---------------------------------------------------------------------------
//
// PRUNE = true (prune.hpp): X / Y / Xh / Yh are in k-d order; the workgroup walks ITS list of
// reference chunks (clist/cdist [nqblk][list_len], nearest box first) and stops at the first
// entry whose lower bound exceeds the largest current K-th distance of its 512 queries; inside
// every wave walks the list on its own (no staging, no barriers) and multiplies only the 32-row
// tiles whose box (tbox_r, [chunk][2][D][CT]) is within reach of one of its two query tiles
// (tbox_q [tile][2][D], per-tile thresholds); cbox_r [chunk][2][D] are the chunk boxes, tested 64 list
// entries at a time before any tile box is read.  The lists then carry the CALLER's
// row numbers (rperm), so ties break exactly as without pruning; the own row of query q is
// rperm-row self_offset + qperm[q].  rsplit must be 1.
//   HEAVY waves: the first `nheavy` waves of the launch's dispatch order (largest boxes: sparse cells and waves holding a far
//   outlier walk 10-30x the average list -- alone they are the tail of a multi-GPU part) are served by S workgroups
//   instead of one: sub-wave s does the bootstrap like everybody, then takes the list windows s, s + S, ... and keeps its
//   own lists; sub-wave 0's go to part_d / part_i, the others' to the side arrays (passed in lo_d / lo_i:
//   [S - 1][KCAP][nheavy * 64]), and prune_heavy_fold_kernel (reduce_kernels.hpp) folds them into the wave's columns (what
//   the shared bootstrap found twice is dropped there).  seed_cfg carries nheavy | S << 24 for this instantiation (it has
//   no seed phase).
//
// LOWER = true: second pass of a search for 16 < K <= 32 neighbours.  The lists hold 16 entries, so the
// first pass finds the 16 nearest per (query, reference split) and the second, identical sweep keeps only
// candidates beyond that split's 16th (lo_d / lo_i = the first pass's lists) and finds the next K - 16;
// the merge then sees two sorted lists per split.  Two sweeps at fp16 speed instead of one fp64 sweep.
//   LC (< KCAP, pruned walk only): the lists hold LC entries in registers while the list ARRAYS keep their KCAP rows (the unused
//   ones are written empty).  K = 9 (C5's kmax = 10) with KCAP = 12: nine entries fit the three-wave budget that K <= 8
//   searches run under, twelve do not (see MCE_H_PRUNE_SMALL above).
//   QTT (round 5): 32-query tiles per wave.  2 everywhere but in the WIDE form of the exhaustive one-k-step sweep (QTT = 4, KST = 1,
//   KCAP = 4: a workgroup serves 1024 queries, i.e. two of the plan's 512-query blocks): every A fragment read from LDS then feeds
//   four MFMAs instead of two and the gate of one query tile runs under the MFMAs of the other three -- C4 (1 M x 1 M x 15, K = 4)
//   41.5 -> 37.7 ms on one box (tools/r05_exp2.sh; the gate-never-passes stream 34.8 -> 31.7).  Longer lists or a second k-step
//   do not fit the register file at four tiles (they spill); twelve waves of two tiles -- three per SIMD -- are SLOWER (42.7 ms).
template <int KST, int KCAP, bool PRUNE = false, bool LOWER = false, int SYM = 0, int LC = KCAP, int QTT = kHQT>
__global__ __launch_bounds__(PRUNE ? 64 : kHThreads, PRUNE ? (LC <= MCE_H_PRUNE_SMALL ? MCE_H_PRUNE_SMALL_WAVES : MCE_H_PRUNE_WAVES) : 2) void knn_f16_kernel(
    const _Float16* __restrict__ Yh, int64_t nchunk_total, int rsplit,
    const _Float16* __restrict__ Xh, const double* __restrict__ qinfo, const double* __restrict__ params,
    const double* __restrict__ X, const double* __restrict__ Y, int64_t nq, int64_t nr, int D,
    int64_t nq_pad, int nqblk, int self_exclude, int64_t self_offset, int ksel,
    double* __restrict__ part_d, int* __restrict__ part_i,
    const int* __restrict__ clist, const float* __restrict__ cdist, int list_len,
    const int* __restrict__ rperm, const int* __restrict__ qperm,
    const float* __restrict__ tbox_r, const float* __restrict__ tbox_q, const float* __restrict__ cbox_r, int qblk0, int qblk_stride, const int* __restrict__ border,
    const double* __restrict__ lo_d, const int* __restrict__ lo_i, int seed_cfg, SymParams sym, float* __restrict__ wg_us)
{
    // wg_us (diagnostic, normally null): every workgroup leaves its duration in microseconds (capi.hip: MCE_PRUNE_TIMES)
    const unsigned long long wg_t0 = wg_us ? wall_clock64() : 0ull;
    static_assert(LC == KCAP || (PRUNE && LC < KCAP), "shorter register lists: pruned walk only");
    static_assert(!(PRUNE && LOWER), "second pass: exhaustive sweep only");
    static_assert(SYM == 0 || (!PRUNE && (!LOWER || SYM == 3)), "symmetric sweep: exhaustive; second pass: the repair launch only (the sweep itself: knn_panel.hpp)");
    static_assert(QTT == 2 || (QTT == 4 && !PRUNE && !LOWER && SYM == 0), "four query tiles: the plain exhaustive sweep only");
    constexpr int QT = QTT;
    constexpr int NL = QT / 2;                           // top-K lists per owner lane
    constexpr int RELB = QT == 4 ? 25 : kHRelBits;       // queue entry = query-local (6 | 7 bits) << RELB | row - first row of the split
    static_assert(SYM == 0 || NL == 1, "symmetric sweep: one list per owner lane");
    constexpr int QPW = QT * 32;                         // queries per wave
    constexpr int QPB = kHWaves * QPW;
    constexpr int CT = f16_chunk_tiles(KST);
    static_assert(CT % 2 == 0, "tile loop is unrolled by two");
    constexpr int CHUNK_BYTES = CT * KST * 1024;
    (void)QPB;
    constexpr int CHUNK_VEC = CHUNK_BYTES / 16;
    constexpr int VPT = (CHUNK_VEC + kHThreads - 1) / kHThreads;
    static_assert(CHUNK_VEC % kHThreads == 0, "chunk must be a whole number of 16-byte vectors per thread");
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    // LDS map: [2 staging buffers][queue d2: W*Q f64][queue packed: W*Q i32][queue next: W*Q i32][heads W*64][votes 2]
    // PRUNE: ONE wave per workgroup (the waves share nothing in that mode, and single-wave
    // workgroups let the hardware balance their very uneven walks): [tile slice + pending ids]
    // [queue d2][queue packed][queue next][heads]; workgroup g serves wave g%8 of query block g/8.
    constexpr int LW = PRUNE ? 1 : kHWaves;                       // waves sharing this LDS allocation
    constexpr int STAGE_BYTES = PRUNE ? f16_prune_slice_bytes(KST, LC) : 2 * CT * KST * 1024;
    char* const stage0 = lds_raw;
    constexpr int QN = PRUNE ? f16_prune_queue(LC) : kHQueue;   // queue entries per wave
    double* const qd2_all = reinterpret_cast<double*>(lds_raw + STAGE_BYTES);
    int* const qpk_all = reinterpret_cast<int*>(qd2_all + LW * QN);
    int* const qnx_all = qpk_all + LW * QN;
    int* const head_all = qnx_all + LW * QN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int lwave = PRUNE ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);                  // index into the LDS regions
    // PRUNE: workgroup -> (position k in the launch's dispatch order, sub-wave): the order is over WAVES (border[k] =
    // block * 8 + wave, largest box first: prune.hip); the first hv_n positions hold S sub-waves each (heavy waves, see
    // above), the rest one
    const int hv_n = PRUNE ? (seed_cfg & 0xffffff) : 0;
    const int hv_S = PRUNE ? (((seed_cfg >> 24) & 0x7f) > 1 ? ((seed_cfg >> 24) & 0x7f) : 1) : 1;
    const int hv_wgs = hv_n * hv_S;
    const bool hv_on = PRUNE && (int)blockIdx.x < hv_wgs;
    const int pr_k = !PRUNE ? 0 : (hv_on ? (int)blockIdx.x / hv_S : hv_n + ((int)blockIdx.x - hv_wgs));
    const int pr_sub = hv_on ? (int)blockIdx.x % hv_S : 0;
    const int pr_step = hv_on ? hv_S : 1;                     // this wave takes every pr_step-th window of its block's list
    const int pr_gw = PRUNE ? border[qblk0 + pr_k * qblk_stride] : 0;      // the launch's (or part's) waves: border[qblk0], border[qblk0 + stride], ...
    const int wave = PRUNE ? pr_gw % kHWaves : lwave;                      // position inside the query block
    // SYM == 2: workgroup = UNIT (panel p of the reference rows, query block a), numbered panel by panel
    // (sym_unit_count): the units running at the same time stream the same few MB of packed rows through L2
    int sym_a = 0, sym_p = 0;
    if constexpr (SYM == 2) sym_unit_decode((int)blockIdx.x, nqblk, kHWaves * kHQT, sym.panel * f16_chunk_tiles(KST), sym_p, sym_a);
    // (SYM == 1, the prepass: the blocks qblk0, qblk0 + 1, ... of the launch -- one rank's share of a multi-GPU partition; with a
    //  stride: qblk0, qblk0 + stride, ... -- its share of the all-pairs-once partition)
    const int qblk = PRUNE ? pr_gw / kHWaves : (SYM == 2 ? sym_a : (SYM == 1 ? qblk0 + (int)blockIdx.x * (qblk_stride > 1 ? qblk_stride : 1) : (int)(blockIdx.x % nqblk)));
    const int split = PRUNE ? 0 : (SYM >= 2 ? 0 : (int)(blockIdx.x / nqblk));

    if constexpr (SYM >= 2) {
        if (SYM == 3 && sym.bucket_flag[qblk] == 0) return;      // repair launch: only the blocks whose bucket overflowed
    }
    // balanced splits (sizes differ by at most one chunk: the host sizes the seed phase for the smallest, capi_search.hpp: seed_cfg)
    const int64_t c_begin = (int64_t)split * nchunk_total / rsplit;
    const int64_t c_end = (int64_t)(split + 1) * nchunk_total / rsplit;

    const double INF = __builtin_huge_val();
    double* const wqd = qd2_all + lwave * QN;                   // exact distance of a queued entry (phase A)
    int* const wq = qpk_all + lwave * QN;                       // packed (query-local, relative row)
    int* const wnx = qnx_all + lwave * QN;                      // next entry of the same query
    int* const whead = head_all + lwave * QPW;                  // chain head per wave-local query
    volatile int* const wvote = head_all + LW * QPW;            // [2] drain votes (chunk parity)
    // PRUNE: the wave's fp64 query rows, one fp64 reference tile and its caller row numbers
    double* const xq = reinterpret_cast<double*>(head_all + LW * QPW + 32);
    double* const ytile = xq + QPW * D;
    int* const yorig = reinterpret_cast<int*>(ytile + 32 * D);
    double* const thrq = reinterpret_cast<double*>(yorig + 32);      // current exact K-th squared distance per wave-local query
    double* const sthr = xq + lwave * QPW;                      // SYM: K-th bound of every wave-local query as of the last drain (same LDS region as xq; never both)
    float mythr = __builtin_huge_valf();                        // PRUNE: largest K-th squared distance among this wave's queries (rounded up)
    float Tq[QT];                                             // PRUNE: the same per 32-query tile (wave-uniform)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) Tq[qt] = __builtin_huge_valf();
    const int jsplit0 = SYM ? 0 : (int)(c_begin * (CT * 32));   // first reference row of this split (SYM: queue entries hold absolute rows)
#pragma unroll
    for (int nl = 0; nl < NL; ++nl) whead[nl * 64 + lane] = -1;

    // lane l OWNS wave-local queries nl*64 + l (query ql = qt*32 + column): their sorted top-KCAP lists live here
    double own_d[NL][LC];
    int own_i[NL][LC];
#pragma unroll
    for (int nl = 0; nl < NL; ++nl)
#pragma unroll
        for (int k = 0; k < LC; ++k) { own_d[nl][k] = INF; own_i[nl][k] = -1; }

    if constexpr (SYM == 2) {
        // A block's lists travel from its unit p - 1 to its unit p through the list arrays.  Units are dispatched in
        // number order, panel by panel -- the previous unit of this block started a whole panel's worth of units ago --
        // so the wait below practically never spins; it is there for the ordering guarantee (the earlier unit is
        // resident or done, never waiting for this one).
        if (sym_p > 0) {
            while (__hip_atomic_load(sym.done + qblk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sym_p) __builtin_amdgcn_s_sleep(32);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int64_t q = (int64_t)qblk * QPB + wave * QPW + lane;
#pragma unroll
            for (int k = 0; k < KCAP; ++k) {
                own_d[0][k] = part_d[(int64_t)k * nq_pad + q];
                own_i[0][k] = part_i[(int64_t)k * nq_pad + q];
            }
        }
    }
    // upper bound on the final K-th squared distance of the owned queries, known before the sweep (seed phase below)
    double seed_thr[NL];
#pragma unroll
    for (int nl = 0; nl < NL; ++nl) seed_thr[nl] = INF;

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
        G[qt] = (q < nq && MCE_ABLATE != 1) ? __builtin_huge_valf() : -__builtin_huge_valf();     // padding queries never pass
    }
    // SYM: c_i = eps_i - |x^_i|^2 of the lane's query (rounded up): the row-side gate is  min A <= R_tile + c_i
    float cR[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) cR[qt] = -__builtin_huge_valf();
    if constexpr (SYM >= 2) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int64_t q = qwave0 + qt * 32 + (lane & 31);
            if (q < nq) {
                const double xn = qinfo[2 * q + 1];
                const double r = sqrt(xn) + params[HP_YHATMAX];
                const double eps = (32.0 * KST) * 0x1p-24 * r * r * (1.0 + 0x1p-9) + params[HP_RHO] + 1e-30;
                cR[qt] = __double2float_ru(eps - xn);
            }
        }
    }
    const int k_last = ksel - 1;
    // gate of query (qt, lane&31) from its current K-th best `thr` (exact squared distance, input
    // units).  The per-query constants are re-read from qinfo (L2) here -- this runs once per drain,
    // and keeping them in registers would cost 6 VGPRs per query tile in the sweep.
    auto gate_of = [&](double thr, int qt) __attribute__((always_inline)) -> float {
        const int64_t q = qwave0 + qt * 32 + (lane & 31);
        if (!(q < nq) || MCE_ABLATE == 1) return -__builtin_huge_valf();
        if (!(thr < INF)) return __builtin_huge_valf();
        const double ex = qinfo[2 * q], xn = qinfo[2 * q + 1];
        const double r = sqrt(xn) + params[HP_YHATMAX];
        // + 2*sqrt(16 KST)*2^-14: even if the matrix unit flushed fp16 subnormal inputs (it does not
        // on gfx950) the bound would hold
        const double ga = (ex + params[HP_EY]) * (1.0 + 1e-9) + 2.0 * sqrt(16.0 * KST) * 0x1p-14;
        const double eps = (32.0 * KST) * 0x1p-24 * r * r * (1.0 + 0x1p-9) + params[HP_RHO] + 1e-30;
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
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            v16f z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[qt][0], z, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 1; ks < KST; ++ks)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) acc[qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], b[qt][ks], acc[qt], 0, 0, 0);
    };

    int qcount = 0;   // wave-uniform number of queued candidates

    auto drain = [&]() __attribute__((always_inline)) {
        if constexpr (!PRUNE) {     // (pruned walk: entries arrive evaluated and linked, see gate_exact)
        // ---- phase A: exact distances + chain links.  8 lanes share one queued pair and read
        // the two rows in 64-byte segments (a row is fetched once, not once per element).  The
        // gather is latency-bound, so ALL loads of a group of NPASS*8 pairs are issued before the
        // first use (the compiler otherwise serialises the passes: one round trip each).
        const int sub = lane & 7;
#ifndef MCE_H_NPASS
#define MCE_H_NPASS 3
#endif
#ifndef MCE_H_NPASS_WIDE
#define MCE_H_NPASS_WIDE 1      // D > 31 (16 loads per lane and pass): more in flight spills (3: 160-300 bytes of scratch per lane)
#endif
        constexpr int NPASS = (KST > 2 || QT == 4) ? MCE_H_NPASS_WIDE : MCE_H_NPASS;            // 8*NPASS pairs in flight; 8 (D <= 32) or 16 (D <= 63) loads per lane and pass (four query tiles: one pass -- registers)
        constexpr int EPL = KST > 2 ? 8 : 4;        // elements per lane: 4 covers D <= 32, 8 covers D <= 63
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
                    ql = (int)(ent >> RELB);
                    j = jsplit0 + (int)(ent & ((1u << (SYM >= 2 ? kHSymRowBits : RELB)) - 1u));
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
                bool ok = okp[u];
                if constexpr (LOWER) {
                    // second pass of a K > 16 search: only what lies BEYOND the first pass's 16th neighbour of
                    // this reference split -- lexicographically in (distance, row), like every tie-break here
                    if (ok) {
                        const int64_t o = ((int64_t)split * KCAP + (KCAP - 1)) * nq_pad + qwave0 + qlp[u];
                        const double ld = lo_d[o];
                        const int li = lo_i[o];
                        int j = jsplit0 + (int)((unsigned)wq[ep[u]] & ((1u << (SYM >= 2 ? kHSymRowBits : RELB)) - 1u));
                        if constexpr (SYM >= 2) j = rperm ? rperm[j] : j;       // (the lists of a symmetric search carry the caller's rows)
                        ok = a0 > ld || (a0 == ld && j > li);       // (list not full: ld = +inf, nothing is left)
                    }
                }
                if constexpr (SYM >= 2) {
                    // every entry gets its distance (-1: no pair behind it) for phase R; only what can still enter
                    // the query's list (K-th bound of the last drain) joins its chain
                    if (sub == 0 && ep[u] < qcount) {
                        wqd[ep[u]] = ok ? a0 : -1.0;
                        if (ok && !(a0 > sthr[qlp[u]])) {
                            wnx[ep[u]] = atomicExch(&whead[qlp[u]], ep[u]);
                        }
                    }
                } else if (ok && sub == 0) {
                    wqd[ep[u]] = a0;
                    wnx[ep[u]] = atomicExch(&whead[qlp[u]], ep[u]);      // push onto the query's chain
                }
            }
        }
        if constexpr (SYM >= 2) {
            // ---- phase R: the ROW side of every evaluated pair, one lane per queue entry ----------------------
            // Pair (i, j): j is a row of another block (behind this one on the ring).  If the distance can still be
            // among j's K smallest (thr[j]) it goes through j's slots -- replace the largest of the K smallest
            // row-side distances so far, by compare-and-swap; once they are all finite their maximum is a new bound
            // on j's K-th distance, published for everybody -- and into the bucket of j's block.  Then the entry's
            // packed word is replaced by the caller's row number of j, which is what the lists carry.
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // one candidate distance d2 for sorted row `row`: replace the largest of the row's K slots if d2 is
            // smaller (compare-and-swap; lock-free, any number of writers) and publish the new K-th as the row's
            // bound.  Returns false if K slots hold strictly smaller distances (the candidate cannot be among the K).
            auto slot_insert = [&](int row, double d2) __attribute__((always_inline)) -> bool {
                unsigned long long* const sl = sym.slots + (int64_t)row * KCAP;
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
                            const unsigned long long ob = __hip_atomic_fetch_min(sym.thr + row, nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (nb < ob) {
                                const unsigned rb = __float_as_uint(sym_row_gate(nk, qinfo[2 * (int64_t)row], params, KST));
                                const unsigned orb = __hip_atomic_fetch_min(sym.rrow + row, rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (rb < orb) {
                                    // the tile's largest R_j, from a snapshot (each value >= its current one): safe to store
                                    const unsigned* const rt = sym.rrow + (int64_t)(row >> 5) * 32;
                                    unsigned m = 0;
                                    for (int k = 0; k < 32; ++k) {
                                        const unsigned v = __hip_atomic_load(rt + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        m = v > m ? v : m;
                                    }
                                    __hip_atomic_store(sym.rtile + (row >> 5), __uint_as_float(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                const int ql = (int)(ent >> RELB);
                const int j = jsplit0 + (int)(ent & ((1u << kHSymRowBits) - 1u));
                const bool rowflag = (ent >> kHSymRowBits) & 1u;        // the lane passed the row gate on this tile: only then can the pair matter to row j
                const double d2 = valid ? wqd[e] : -1.0;
                const bool ok = valid && d2 >= 0.0;
                const int oj = ok ? rperm[j] : -1;
                const int jb = j / QPB;
                bool rs = ok && SYM == 2 && jb != qblk && rowflag;
                if (rs) rs = d2 <= __longlong_as_double((long long)__hip_atomic_load(sym.thr + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (rs) rs = slot_insert(j, d2);
                if (rs) {
                    const int slot = atomicAdd(sym.bucket_cnt + jb, 1);
                    if ((unsigned)slot < (unsigned)sym.cap) {      // (unsigned: a count that is not a count ends in the repair pass, not in a wild store)
                        SymEntry en;
                        en.d2 = d2;
                        en.src = rperm[qwave0 + ql];
                        en.row = j;
                        sym.bucket[(int64_t)jb * sym.cap + slot] = en;
                    } else {
                        sym.bucket_flag[jb] = 1;
                    }
                }
                if (valid) wq[e] = oj;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        }
        // ---- phase B: every owner lane folds its chain(s) into its register list(s) ----------
        // (pruned walk: fetching the NEXT chain entry while this one is inserted, and skipping rounds in which no lane's entry
        //  beats its list's last, was measured: C5 95.4 -> 99.4 ms.)
#pragma unroll
        for (int nl = 0; nl < NL; ++nl) {
            int cur = whead[nl * 64 + lane];
            whead[nl * 64 + lane] = -1;
            while (__any(cur >= 0)) {
                const bool on = cur >= 0;
                const int ce = on ? cur : 0;
                const double d2 = on ? wqd[ce] : INF;
                const int j = (PRUNE || SYM >= 2) ? wq[ce] : jsplit0 + (int)((unsigned)wq[ce] & ((1u << RELB) - 1u));
                cur = on ? wnx[ce] : -1;
                // ascending list, ties by row; d2 = +inf (idle lane) changes nothing
                bool c_hi = (d2 < own_d[nl][LC - 1]) || (d2 == own_d[nl][LC - 1] && j < own_i[nl][LC - 1] && d2 < INF);
#pragma unroll
                for (int k = LC - 1; k >= 1; --k) {
                    const bool c_lo = (d2 < own_d[nl][k - 1]) || (d2 == own_d[nl][k - 1] && j < own_i[nl][k - 1] && d2 < INF);
                    own_d[nl][k] = c_lo ? own_d[nl][k - 1] : (c_hi ? d2 : own_d[nl][k]);
                    own_i[nl][k] = c_lo ? own_i[nl][k - 1] : (c_hi ? j : own_i[nl][k]);
                    c_hi = c_lo;
                }
                own_d[nl][0] = c_hi ? d2 : own_d[nl][0];
                own_i[nl][0] = c_hi ? j : own_i[nl][0];
            }
        }
        qcount = 0;
        // ---- refresh the gates: lane l needs the K-th best of queries ql = qt*32 + (l&31),
        // owned by lane ql & 63 in list ql >> 6
        double thr_own[NL];
#pragma unroll
        for (int nl = 0; nl < NL; ++nl) {
            thr_own[nl] = own_d[nl][LC - 1];
#pragma unroll
            for (int k = 0; k < LC - 1; ++k) thr_own[nl] = (k == k_last) ? own_d[nl][k] : thr_own[nl];
            thr_own[nl] = fmin(thr_own[nl], seed_thr[nl]);
        }
        if constexpr (SYM >= 2) {
            // publish: thr[q] takes this list's K-th bound and gives back what the row side knows (the K-th of q's
            // slots); the row-side gate constants follow, and the maximum over each 32-row tile (= half a wave)
            const int64_t q = qwave0 + lane;
            double t = thr_own[0];
            float R = 0.0f;
            if (q < nq) {
                if (t < INF) {
                    const unsigned long long mb = (unsigned long long)__double_as_longlong(t);
                    const unsigned long long ob = __hip_atomic_fetch_min(sym.thr + q, mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    t = fmin(t, __longlong_as_double((long long)ob));
                } else {
                    t = __longlong_as_double((long long)__hip_atomic_load(sym.thr + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                }
                const unsigned rb = __float_as_uint(sym_row_gate(t, qinfo[2 * q], params, KST));
                const unsigned orb = __hip_atomic_fetch_min(sym.rrow + q, rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                R = __uint_as_float(rb < orb ? rb : orb);
            }
            thr_own[0] = t;
            seed_thr[0] = t;
            sthr[lane] = t;
            float m = R;
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            if ((lane & 31) == 0 && SYM == 2) __hip_atomic_store(sym.rtile + ((qwave0 + lane) >> 5), m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) G[qt] = gate_of(__shfl(thr_own[qt >> 1], (qt & 1) * 32 + (lane & 31), 64), qt);
        if constexpr (PRUNE) {
#pragma unroll
            for (int nl = 0; nl < NL; ++nl) thrq[nl * 64 + lane] = thr_own[nl];
            mythr = 0.0f;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const double tl = __shfl(thr_own[qt >> 1], (qt & 1) * 32 + (lane & 31), 64);
                double m = (qwave0 + qt * 32 + (lane & 31) < nq) ? tl : 0.0;     // padding queries do not hold the tile back
#pragma unroll
                for (int o = 16; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
                Tq[qt] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(__double2float_ru(m))));
                mythr = fmaxf(mythr, Tq[qt]);
            }
        }
    };

    // gate + enqueue for one finished tile; jb0 = first reference row of the tile.
    // C layout of 32x32 f32: lane l -> query column l&31, rows (r&3) + 8*(r>>2) + 4*(l>>5)
    // SYM: Rt = the tile's row-side gate constant (wave-uniform; -inf: column side only)
    auto process = [&](const v16f (&acc)[QT], int jb0, float Rt) __attribute__((always_inline)) {
#if MCE_ABLATE == 2
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) asm volatile("" ::"v"(acc[qt]));
#endif
        return;
#endif
        bool passq[QT];
        bool pass = false;
        float mm[QT];                          // the lane's smallest accumulator (SYM: read again in the event path)
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
            mm[qt] = min3f(m0, m3, m3);
            if constexpr (SYM == 2) passq[qt] = mm[qt] <= fmaxf(G[qt], Rt + cR[qt]);      // either side
            else passq[qt] = mm[qt] <= G[qt];
            pass |= passq[qt];
        }
        if (__any(pass)) {
            const int jrel0 = jb0 - jsplit0 + 4 * (lane >> 5);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                if (!__any(passq[qt])) continue;
                // per-lane bit mask of the accumulators under the gate (branch-free) ...
                unsigned pm = 0;
                const float gq = (SYM == 2) ? fmaxf(G[qt], Rt + cR[qt]) : G[qt];
#pragma unroll
                for (int r = 0; r < 16; ++r) pm |= (acc[qt][r] <= gq) ? (1u << r) : 0u;
                // ... then one queue entry per lane and trip (usually a single trip)
                unsigned long long m = __ballot(pm != 0);
                while (m) {
                    if (qcount > kHQueue - 64) drain();
                    if (pm != 0) {
                        const int r = __builtin_ctz(pm);
                        pm &= pm - 1;
                        const int slot = qcount + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                        unsigned rel = (unsigned)(jrel0 + (r & 3) + 8 * (r >> 2));
                        if constexpr (SYM == 2) rel |= (mm[qt] <= Rt + cR[qt]) ? (1u << kHSymRowBits) : 0u;      // (lane-level: most events are column-side only)
                        wq[slot] = (int)(((unsigned)(qt * 32 + (lane & 31)) << RELB) | rel);
                    }
                    qcount += __builtin_popcountll(m);
                    m = __ballot(pm != 0);
                }
            }
        }
    };

    // PRUNE: gate + EXACT evaluation of one finished tile.  The tiles that yield candidates are few
    // (~7 %) but yield them in bursts (~16 each), so instead of queueing (query, row) pairs for a later
    // scattered gather, the tile's 32 fp64 rows are fetched once, contiguously, into LDS and every
    // passing pair is evaluated on the spot against the query rows staged at kernel start -- with the
    // SAME summation tree as the sweep's phase A (8 partial sums over i = s, s+8, ... combined pairwise),
    // so the distances stay bit-identical.  What is queued is (exact d2, caller row), already linked
    // into its query's chain; drain() is then insertion (phase B) only.
#if MCE_PRUNE_PROF
    long long gx_iter = 0, gx_cand = 0, gx_useful = 0, gx_stage_t = 0, gx_drain_t = 0, gx_drain_n = 0;
#endif
    int qorig[QT];
#ifndef MCE_H_PRUNE_GXB
#define MCE_H_PRUNE_GXB 4
#endif
    // gate_exact: fp64 loads in flight per lane (global -> LDS staging of the tile, LDS reads of the exact evaluation).  One at a
    // time -- what a loop over the runtime d with its conditions compiles to -- every load waits for the one before.
    constexpr int GXB = MCE_H_PRUNE_GXB;
    // partial sums of |x - y|^2 (x: xr[i * QPW], y: yr[i * 32], both LDS), dimension i into sp[i & 7] in ascending order of i --
    // the sweep's phase A tree.  DD - 1 may be d itself, one row past the data: read (f16_prune_lds_bytes keeps it inside
    // the allocation) and replaced by zero, fma(0, 0, a) = a.
    auto exact_sums_dd = [&](const double* xr, const double* yr, double (&sp)[8], auto dd) __attribute__((always_inline)) {
        constexpr int DD = decltype(dd)::value;
#pragma unroll
        for (int sb = 0; sb < 8; ++sb) sp[sb] = 0.0;
#pragma unroll
        for (int g0 = 0; g0 < DD; g0 += GXB) {
            double xv[GXB], yv[GXB];
#pragma unroll
            for (int u = 0; u < GXB; ++u)
                if (g0 + u < DD) { xv[u] = xr[(g0 + u) * QPW]; yv[u] = yr[(g0 + u) * 32]; }
#pragma unroll
            for (int u = 0; u < GXB; ++u) {
                const int i = g0 + u;
                if (i < DD) {
                    double t = xv[u] - yv[u];
                    if (i == DD - 1) t = (i < D) ? t : 0.0;
                    sp[i & 7] = fma(t, t, sp[i & 7]);
                }
            }
        }
    };
    auto exact_sums = [&](const double* xr, const double* yr, double (&sp)[8]) __attribute__((always_inline)) {
        static_assert(kPruneDims <= 16, "exact_sums: even dimension counts up to 16");
        switch ((D + 1) >> 1) {
        case 1: exact_sums_dd(xr, yr, sp, std::integral_constant<int, 2>()); break;
        case 2: exact_sums_dd(xr, yr, sp, std::integral_constant<int, 4>()); break;
        case 3: exact_sums_dd(xr, yr, sp, std::integral_constant<int, 6>()); break;
        case 4: exact_sums_dd(xr, yr, sp, std::integral_constant<int, 8>()); break;
        case 5: exact_sums_dd(xr, yr, sp, std::integral_constant<int, 10>()); break;
        case 6: exact_sums_dd(xr, yr, sp, std::integral_constant<int, 12>()); break;
        case 7: exact_sums_dd(xr, yr, sp, std::integral_constant<int, 14>()); break;
        default: exact_sums_dd(xr, yr, sp, std::integral_constant<int, 16>()); break;
        }
    };
    auto gate_exact = [&](const v16f (&acc)[QT], int jb0) {
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
        if (!__any(pass)) return;
        // The tile's rows (contiguous in the k-d ordered copy) and their caller row numbers: the first GXB loads per lane (all
        // of them at d <= 8) go out now, the masks of passing rows are taken while they are in flight, then the LDS writes.
        // Element e = row * d + i, row = floor((e + 0.5) / d) in fp32: e < 512, so the quotient is never within rounding of
        // an integer.
#if MCE_PRUNE_PROF
        const long long gx_ts0 = clock64();
#endif
        const int nrow = (nr - jb0 < 32) ? (int)(nr - jb0) : 32;
        const double* yt = Y + (int64_t)jb0 * D;
        const int ne = nrow * D;
        const float inv_d = 1.0f / (float)D;
        int ln = lane;
        asm volatile("" : "+v"(ln));             // (the eight element numbers and LDS offsets below depend on the lane and d only:
                                                 //  computed once before the walk they would occupy fifteen registers throughout)
        const int oj0 = (lane < 32 && lane < nrow) ? rperm[jb0 + (lane & 31)] : -1;
        double yv0[GXB];
#pragma unroll
        for (int u = 0; u < GXB; ++u) {
            const int e = ln + u * 64;
            yv0[u] = yt[e < ne ? e : 0];
        }
        // the passing rows of the lane's column, per query tile (the accumulators are dead from here: 32 registers free while
        // the rows are evaluated)
        unsigned pmq[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            pmq[qt] = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // (0 / 1 and a shift by an inline constant: as a select of 1 << r the compiler keeps the twelve values that
                //  are no inline constants in registers for the whole walk)
                unsigned bit = (acc[qt][r] <= G[qt]) ? 1u : 0u;
                asm("" : "+v"(bit));
                pmq[qt] |= bit << r;
                asm("" : "+v"(pmq[qt]));               // (folded in here, not where the mask is first used: sixteen registers otherwise)
            }
        }
#pragma unroll
        for (int u = 0; u < GXB; ++u) {
            const int e = ln + u * 64;
            const int row = (int)(((float)e + 0.5f) * inv_d);
            if (e < ne) ytile[(e - row * D) * 32 + row] = yv0[u];      // [i][row]: conflict-free reads
        }
        if (lane < 32) yorig[lane] = oj0;
#pragma unroll
        for (int h0 = GXB; h0 < 8; h0 += GXB) {
            if (h0 * 64 < ne) {
                double yv[GXB];
#pragma unroll
                for (int u = 0; u < GXB; ++u) {
                    const int e = ln + (h0 + u) * 64;
                    yv[u] = yt[e < ne ? e : 0];
                }
#pragma unroll
                for (int u = 0; u < GXB; ++u) {
                    const int e = ln + (h0 + u) * 64;
                    const int row = (int)(((float)e + 0.5f) * inv_d);
                    if (e < ne) ytile[(e - row * D) * 32 + row] = yv[u];
                }
            }
        }
#if MCE_PRUNE_PROF
        gx_stage_t += clock64() - gx_ts0;
#endif
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            if (!__any(passq[qt])) continue;
            unsigned pm = pmq[qt];
            const int ql = qt * 32 + (lane & 31);
            const double* xr = xq + ql;                            // xq[i][ql]
            while (__any(pm != 0)) {
#if MCE_PRUNE_PROF == 3
                if (qcount > QN - 64) { const long long td_ = clock64(); drain(); gx_drain_t += clock64() - td_; gx_drain_n += 1; }
#else
                if (qcount > QN - 64) drain();
#endif
                const bool has = pm != 0;
                const int r = has ? __builtin_ctz(pm) : 0;
                pm &= pm - 1;
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const double* yr = ytile + row;                    // ytile[i][row]
                // D <= 15: at most two elements per partial sum (sp[i & 7]); the code for the even dimension count DD >= d is
                // chosen by a uniform switch (exact_sums): no conditions between the LDS reads, which then go out together.
                double sp[8];
                exact_sums(xr, yr, sp);
                const double d2 = ((sp[0] + sp[1]) + (sp[2] + sp[3])) + ((sp[4] + sp[5]) + (sp[6] + sp[7]));
                const int oj = yorig[row];
                // (beyond the K-th best of the last drain: can never enter -- the thresholds only shrink)
                const bool ok = has && oj >= 0 && qwave0 + ql < nq && !(d2 > thrq[ql]) &&
                                !(self_exclude && (int64_t)oj == self_offset + qorig[qt]);
                const unsigned long long m = __ballot(ok);
                if (ok) {
                    const int slot = qcount + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                    wqd[slot] = d2;
                    wq[slot] = oj;
                    wnx[slot] = atomicExch(&whead[ql], slot);
                }
#if MCE_PRUNE_PROF
                gx_iter += 1; gx_cand += __builtin_popcountll(m); gx_useful += __builtin_popcountll(__ballot(ok));
#endif
                qcount += __builtin_popcountll(m);
            }
        }
    };
    if constexpr (PRUNE) {
        static_assert(KST == 1, "pruned walk: d <= 15");
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int64_t q = qwave0 + qt * 32 + (lane & 31);
            qorig[qt] = q < nq ? qperm[q] : -1;
        }
#pragma unroll
        for (int nl = 0; nl < NL; ++nl) thrq[nl * 64 + lane] = INF;
        for (int e = lane; e < QPW * D; e += 64) {
            const int64_t q = qwave0 + e / D;
            xq[(e % D) * QPW + e / D] = q < nq ? X[q * (int64_t)D + (e % D)] : 0.0;
        }
    } else {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) qorig[qt] = 0;
    }

    v16f accA[QT], accB[QT];
    int jbA = 0, jbB = 0;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[qt][r] = __builtin_nanf("");       // "no pending tile": NaN never passes the gate

    if (!PRUNE && tid < 2) wvote[tid] = 0;

    // one staged chunk (vote / barrier / prefetch are done around it).  A macro, not a lambda: the
    // accumulators must stay in registers across the two call sites.
    // (TLO, THI: the chunk's tiles [TLO, THI), both even -- the whole chunk except at the ends of a symmetric
    //  sweep's ranges; RTL: lane t holds the row-side gate constant of the chunk's tile t, SYM only)
    float rA = -__builtin_huge_valf(), rB = -__builtin_huge_valf();
#define MCE_SWEEP_CHUNK(BUF, JCHUNK, PROC, TLO, THI, RTL)                                                  \
    do {                                                                                                   \
        const char* lbuf = stage0 + (BUF) * CHUNK_BYTES + lane * 16;                                       \
        const int jchunk = (JCHUNK);                                                                       \
        const int thi_ = (THI);                                                                            \
        v8h a0[KST], a1[KST];                                                                              \
        load_a(lbuf + ((TLO) * KST) * 1024, a0);                                                           \
        _Pragma("unroll 1") for (int t = (TLO); t < thi_; t += 2)                                          \
        {                                                                                                  \
            load_a(lbuf + ((t + 1) * KST) * 1024, a1);                                                     \
            mfma_tile(a0, accA);                                                                           \
            jbA = jchunk + t * 32;                                                                         \
            PROC(accB, jbB, rB);                                                                           \
            if constexpr (SYM == 2) rA = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(RTL), t));       \
            load_a(lbuf + ((t + 2 < thi_ ? t + 2 : t) * KST) * 1024, a0); /* last trip: harmless re-read */ \
            mfma_tile(a1, accB);                                                                           \
            jbB = jchunk + (t + 1) * 32;                                                                   \
            PROC(accA, jbA, rA);                                                                           \
            if constexpr (SYM == 2) rB = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(RTL), t + 1));   \
        }                                                                                                  \
    } while (0)

    // Drains are taken by ALL waves of the workgroup at the same chunk boundary (a wave that
    // drained alone would hold the other seven at the next barrier): before the barrier a
    // wave whose queue is filling raises the vote of this chunk's parity; after the barrier
    // everybody reads it.  (process() still drains locally if its queue would overflow.)
    if constexpr (!PRUNE) {
        // ---- seed: an upper bound on every query's final K-th distance BEFORE anything is queued ----------
        // A stream of N references in arbitrary order puts ~K ln(N/K) genuine updates per query through the
        // exact path, half of them within the first ~sqrt(N K) rows while the lists are still loose.  A few
        // chunks of the split are therefore swept twice: once here, only tracking the minimum of A per GROUP of
        // `tg` tiles and, per query, the K' smallest of those group minima (K' = K, + 1 if the query itself is
        // among the references).  They belong to K' different rows, at most one of them the query's own, so the
        // K-th nearest row is no farther than the K'-th smallest group minimum -- turned into a rigorous bound on
        // the true distance with the same error terms as the gate (true <= sqrt(A + |x^|^2 + eps) + e_x + max e_y).
        // With many more groups than K' this is close to the K-th distance within the seed rows.  The sweep
        // proper then starts with that threshold instead of +inf: no flood of early candidates.
        if constexpr (!LOWER && (MCE_ABLATE == 0) && SYM < 2) {
            const int kneed = ksel + (self_exclude ? 1 : 0);
            const int tg = (seed_cfg >> 16) & 0xfff;
            const int smode = (seed_cfg >> 28) & 3;       // where the seed chunks are: 0 spread evenly, 1 the first ones, 2 half and half
            const int nseed = ((int64_t)(seed_cfg & 0xffff) * 2 <= c_end - c_begin) ? (seed_cfg & 0xffff) : 0;   // chunks (host: f16_seed_cfg)
            if (nseed > 0 && tg > 0 && kneed <= KCAP + 1) {
                const float FINF = __builtin_huge_valf();
                static_assert(QT == 2 || QT == 4, "wait-state asm names the accumulator tiles");
                float gm[QT], sm[QT][KCAP + 1];         // running group minimum; the KCAP + 1 smallest group minima, ascending
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    gm[qt] = FINF;
#pragma unroll
                    for (int k = 0; k <= KCAP; ++k) sm[qt][k] = FINF;
                }
                int tcnt = 0;
#define MCE_SEED_GROUP_END()                                                                               \
                do {                                                                                       \
                    _Pragma("unroll") for (int qt = 0; qt < QT; ++qt)                                      \
                    {                                                                                      \
                        float v_ = fminf(gm[qt], __shfl_xor(gm[qt], 32, 64));     /* both row halves */    \
                        gm[qt] = FINF;                                                                     \
                        _Pragma("unroll") for (int k = 0; k <= KCAP; ++k)                                  \
                        {                                                                                  \
                            const float lo_ = fminf(sm[qt][k], v_);                                        \
                            v_ = fmaxf(sm[qt][k], v_);                                                     \
                            sm[qt][k] = lo_;                                                               \
                        }                                                                                  \
                    }                                                                                      \
                    tcnt = 0;                                                                              \
                } while (0)
                // (the minima are inline asm, which the compiler's hazard recogniser does not cover and which -- with
                //  no branch in this loop to hold them in place -- it schedules right behind the MFMAs that write their
                //  operands: the wait states are spelled out, tied to the accumulators)
#define MCE_SEED_WAIT(ACC)                                                                                 \
                do {                                                                                       \
                    if constexpr (QT == 4) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(ACC[0]), "+v"(ACC[1]), "+v"(ACC[QT - 2]), "+v"(ACC[QT - 1])); \
                    else asm volatile("s_nop 15\n\ts_nop 3" : "+v"(ACC[0]), "+v"(ACC[1]));                 \
                } while (0)
#define MCE_SEED_TILE(ACC, JB, RR)                                                                         \
                do {                                                                                       \
                    MCE_SEED_WAIT(ACC);                                                                    \
                    _Pragma("unroll") for (int qt = 0; qt < QT; ++qt)                                      \
                    {                                                                                      \
                        const v16f& c_ = ACC[qt];                                                          \
                        float m0 = min3f(c_[0], c_[1], c_[2]);                                             \
                        float m1 = min3f(c_[3], c_[4], c_[5]);                                             \
                        float m2 = min3f(c_[6], c_[7], c_[8]);                                             \
                        float m3 = min3f(c_[9], c_[10], c_[11]);                                           \
                        float m4 = min3f(c_[12], c_[13], c_[14]);                                          \
                        m0 = min3f(m0, m1, m2);                                                            \
                        m3 = min3f(m3, m4, c_[15]);                                                        \
                        gm[qt] = min3f(gm[qt], m0, m3);                                                    \
                    }                                                                                      \
                    if (++tcnt == tg) MCE_SEED_GROUP_END();                                                \
                } while (0)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accB[qt][r] = FINF;          // "no pending tile" for a minimum
                // the seed chunks are spread evenly over the split: the rows of a Markov chain are correlated in
                // time, and its first stretch alone would say little about queries elsewhere in the posterior
                // Which chunks: spread evenly (rows in the caller's order: a Markov chain's first stretch alone says little
                // about queries elsewhere in the posterior).  Rows sorted by distance from the mean (symmetric sweep): in
                // high dimensions the rows NEAREST the mean are near every query and hold most of every row's neighbours --
                // 1M x 27: the sweep that follows takes 41.4 ms after the first 32 k rows, 45.7 after 32 k spread evenly --
                // in low dimensions they are not (1M x 6: 50 vs 37 ms); half and half serves both (kSeed* in capi.hip).
                const int64_t srange = c_end - c_begin;
                const int nfirst = smode == 1 ? nseed : (smode == 2 ? nseed / 2 : 0);
                auto seed_chunk = [&](int cc) -> int64_t {
                    if (cc < nfirst) return c_begin + cc;
                    const int ns = nseed - nfirst;
                    return c_begin + nfirst + (int64_t)(cc - nfirst) * ((srange - nfirst) / ns);
                };
                stage_async(seed_chunk(0), 0);
                for (int cc = 0; cc < nseed; ++cc) {
                    const int buf = cc & 1;
                    dma_barrier();
                    if (cc + 1 < nseed) stage_async(seed_chunk(cc + 1), buf ^ 1);
                    MCE_SWEEP_CHUNK(buf, 0, MCE_SEED_TILE, 0, CT, 0.0f);
                }
                MCE_SEED_TILE(accB, 0, 0.0f);           // the pending tile
                if (tcnt > 0) MCE_SEED_GROUP_END();     // a last, smaller group
#undef MCE_SEED_TILE
#undef MCE_SEED_GROUP_END
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accB[qt][r] = __builtin_nanf("");
                // owner lane l holds query nl*64 + l = tile 2 nl + (l >> 5), column l & 31 -- its own column
#pragma unroll
                for (int nl = 0; nl < NL; ++nl) {
                    float u0 = sm[2 * nl][KCAP], u1 = sm[2 * nl + 1][KCAP];
#pragma unroll
                    for (int k = 0; k < KCAP; ++k) {
                        u0 = (k == kneed - 1) ? sm[2 * nl][k] : u0;
                        u1 = (k == kneed - 1) ? sm[2 * nl + 1][k] : u1;
                    }
#if defined(__HIP_DEVICE_COMPILE__)
                    asm("" : "+v"(u0), "+v"(u1));       // plain registers: keeps the select from becoming an indexed (scratch) array
#endif
                    const float a_up = (lane >> 5) ? u1 : u0;
                    const int64_t q = qwave0 + nl * 64 + lane;
                    if (q < nq && a_up < FINF) {
                        const double ex = qinfo[2 * q], xn = qinfo[2 * q + 1];
                        const double r = sqrt(xn) + params[HP_YHATMAX];
                        const double ga = (ex + params[HP_EY]) * (1.0 + 1e-9) + 2.0 * sqrt(16.0 * KST) * 0x1p-14;
                        const double eps = (32.0 * KST) * 0x1p-24 * r * r * (1.0 + 0x1p-9) + params[HP_RHO] + 1e-30;
                        const double h2 = fmax((double)a_up + xn + eps, 0.0);
                        const double dd = sqrt(h2) * (1.0 + 1e-12) + ga;
                        seed_thr[nl] = dd * dd * (1.0 + 1e-12) / s2 * (1.0 + 1e-12);
                    }
                }
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) G[qt] = gate_of(__shfl(seed_thr[qt >> 1], (qt & 1) * 32 + (lane & 31), 64), qt);
                __syncthreads();            // everybody is done with the staging buffers
            }
        }
        if constexpr (SYM == 1) {
            // ---- prepass of the symmetric sweep: publish every row's bound and its row-side gate constants,
            // empty its slots; nothing else happens in this launch
            const int64_t q = qwave0 + lane;
            float R = 0.0f;
            if (q < nq) {
                sym.thr[q] = (unsigned long long)__double_as_longlong(seed_thr[0]);
                R = sym_row_gate(seed_thr[0], qinfo[2 * q], params, KST);
                sym.rrow[q] = __float_as_uint(R);
                const int sstride = sym.slot_stride ? sym.slot_stride : KCAP;
                for (int k = 0; k < sstride; ++k) sym.slots[q * sstride + k] = 0x7FF0000000000000ull;
            } else {
                sym.thr[q] = 0x7FF0000000000000ull;
                sym.rrow[q] = 0u;
            }
            float m = R;
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            if ((lane & 31) == 0) sym.rtile[q >> 5] = m;
            return;
        } else if constexpr (SYM >= 2) {
            // ---- symmetric sweep: the blocks 0..a (SYM = 3, repair: everything, column side only)
            {
                const int64_t q = qwave0 + lane;
                const double t0 = __longlong_as_double((long long)__hip_atomic_load(sym.thr + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                seed_thr[0] = t0;
                sthr[lane] = t0;
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) G[qt] = gate_of(__shfl(t0, (qt & 1) * 32 + (lane & 31), 64), qt);
            }
            constexpr int TPB = QPB / 32;                          // tiles per query block
            const int Tr = (int)((nr + 31) / 32);
            const int Tre = Tr + (Tr & 1);                         // (the tile loop takes tiles in pairs; Yh is padded to whole chunks)
            // tiles [0, hi): the blocks 0..a, swept upwards (the block's own rows last).  Upwards, not from the own rows down:
            // the rows met first then belong to the blocks that finished longest ago (tightest bounds: fewest row-side
            // candidates), and blocks running at the same time stream the same chunks through L2 (measured at 1M x 27:
            // 47.6 ms vs 50.0 downwards, 48.5 from a per-block offset).
            //
            // A block's range is worked off PANEL by panel (sym.panel chunks: a few MB of packed rows), one unit each,
            // and the units are numbered panel-major: at any time the chip is on one panel (L2 serves 31 of the 32 CUs
            // of an XCD), and the units are of equal length, so the launch ends with at most one unit of tail instead
            // of the longest block's.  (One workgroup per block, 0..a in one go: blocks of different length drift apart,
            // 79 GB through the fabric per search at 1M x 27 instead of 17, and ~13 % of tail.)
            int lo = 0, hi = Tre;
            if constexpr (SYM == 2) sym_unit_tiles(sym_p, qblk, TPB, sym.panel * CT, Tre, lo, hi);
            const int cfirst = lo / CT;
            const int ntot = hi > lo ? (hi - 1) / CT - cfirst + 1 : 0;
            // k-th chunk of the sequence: its number and its tiles [tlo, thi)
            auto seq_at = [&](int k, int& c, int& tlo, int& thi) {
                c = cfirst + k;
                tlo = 0;
                thi = hi - c * CT < CT ? hi - c * CT : CT;
            };
            // lane t <- the row-side gate constant of tile t of chunk c (own rows and tiles outside the range: none)
            auto rt_load = [&](int c, int tlo, int thi) -> float {
                const int t = c * CT + lane;
                const bool en = SYM == 2 && lane >= tlo && lane < thi && t / TPB != qblk;
                return en ? __hip_atomic_load(sym.rtile + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -__builtin_huge_valf();
            };
            float rt_cur = -__builtin_huge_valf(), rt_next = -__builtin_huge_valf();
            int c0 = 0, tlo0 = 0, thi0 = 0;
            if (ntot > 0) {
                seq_at(0, c0, tlo0, thi0);
                stage_async(c0, 0);
                rt_next = rt_load(c0, tlo0, thi0);
            }
            for (int k = 0; k < ntot; ++k) {
                const int buf = k & 1;
                if (qcount >= kHDrainTrigger && lane == 0) wvote[buf] = 1;
                dma_barrier();
                int c, tlo, thi;
                seq_at(k, c, tlo, thi);
                rt_cur = rt_next;
                if (k + 1 < ntot) {
                    int cn, tlon, thin;
                    seq_at(k + 1, cn, tlon, thin);
                    stage_async(cn, buf ^ 1);
                    rt_next = rt_load(cn, tlon, thin);
                }
                const bool all_drain = wvote[buf] != 0;
                if (tid == 0) wvote[buf ^ 1] = 0;
                if (all_drain) drain();
                MCE_SWEEP_CHUNK(buf, c * (CT * 32), process, tlo, thi, rt_cur);
            }
        } else {
        if (c_begin < c_end) stage_async(c_begin, 0);
        for (int64_t c = c_begin; c < c_end; ++c) {
            const int buf = (int)((c - c_begin) & 1);
            if (qcount >= kHDrainTrigger && lane == 0) wvote[buf] = 1;
            dma_barrier();
            if ((c + 1) < c_end) stage_async(c + 1, buf ^ 1);
            const bool all_drain = wvote[buf] != 0;
            if (tid == 0) wvote[buf ^ 1] = 0;          // re-arm the other parity (read again only after the next barrier)
            if (all_drain) drain();
            MCE_SWEEP_CHUNK(buf, (int)(c * (CT * 32)), process, 0, CT, 0.0f);
        }
        }
    } else {
        // Sparse walk: no chunk staging and no workgroup barriers -- every wave goes down the block's
        // chunk list (nearest box first) on its own, stops at the first entry whose bound exceeds
        // the largest K-th distance among ITS 64 queries, tests the 64 tile boxes of a chunk in one
        // pass (lane t <-> tile t), collects the tiles within reach and multiplies them in batches
        // of kBatch: the batch's A tiles (1 KB each) go through registers into the wave's private
        // slice of the staging area and are swept from there.
        constexpr int kBatch = f16_prune_batch(LC);
        constexpr int PCT = kHPruneChunkTiles;               // tiles per list chunk: one per lane
        constexpr int kPruneDrainTrigger = f16_prune_trigger(LC);
        static_assert(f16_prune_slice_bytes(KST, LC) >= kBatch * KST * 1024 + 256, "tile slice");
        const int* const mylist = clist + (int64_t)qblk * list_len;
        const float* const mydist = cdist + (int64_t)qblk * list_len;
        char* const wbuf = stage0;                                          // [kBatch tiles][pending ids]
        int* const wl = reinterpret_cast<int*>(wbuf + kBatch * KST * 1024);
        // This wave's QT query-tile boxes, all in ONE register: lane qt*32 + i holds the lower edge of dimension i, lane
        // qt*32 + 16 + i the upper.  The box tests read them with v_readlane (constant lanes) where they are needed --
        // as scalar loads the compiler kept 4 * d pointers alive across the walk, spilled, and waited for every
        // dimension's loads one after the other.
        static_assert(QT == 2 && kPruneDims <= 16, "query boxes: one value per lane");
        int qbv;
        {
            const int qi = lane & 15, qh = (lane >> 4) & 1;
            const float* const qb = tbox_q + (qwave0 / 32) * (int64_t)(2 * D);
            qbv = __float_as_int(qi < D ? qb[(lane >> 5) * 2 * D + qh * D + qi] : 0.0f);
        }
        // (largest coordinate magnitude among this wave's queries: a query's float coordinate is within 2^-24 of that of its fp64 one)
        float qabs = fabsf(__int_as_float(qbv));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) qabs = fmaxf(qabs, __shfl_xor(qabs, o, 64));
        const int qabs_bits = __builtin_amdgcn_readfirstlane(__float_as_int(qabs));
        // squared gap between a reference box (lower edges at p[i * stride], upper at p[(D + i) * stride]) and each query
        // tile's box.  The loads of eight dimensions are issued together (indices past d clamped, their terms zeroed:
        // fma(0, 0, acc) = acc, so the sums are those of the one-dimension-at-a-time loop, bit for bit).
        // (keep: the lane's box, widened, goes to LDS for query_reach below -- the tile slice, idle while tiles are collected)
        static_assert(kBatch * KST * 1024 >= 64 * 64, "tile slice: room for 64 boxes of 8 dimensions");
        typedef float v4f_t __attribute__((ext_vector_type(4)));
        typedef float v2f_t __attribute__((ext_vector_type(2)));
        auto box_gap = [&](const float* p, const int stride, float (&acc)[QT], auto keep) __attribute__((always_inline)) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) acc[qt] = 0.0f;
            int qv = qbv, Dq = D;
            // (the lane reads and the clamped offsets stay here: hoisted out of the walk they would only be spilled)
            asm volatile("" : "+v"(qv), "+s"(Dq));
#pragma unroll
            for (int h0 = 0; h0 < kPruneDims; h0 += 8) {
                if (h0 < D) {
                    float rlo[8], rhi[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int ii = (h0 + u < Dq) ? h0 + u : h0;
                        rlo[u] = p[ii * stride];
                        rhi[u] = p[(Dq + ii) * stride];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = h0 + u;
                        if (i < kPruneDims) {
#pragma unroll
                            for (int qt = 0; qt < QT; ++qt) {
                                const float qlo = __int_as_float(__builtin_amdgcn_readlane(qv, qt * 32 + i));
                                const float qhi = __int_as_float(__builtin_amdgcn_readlane(qv, qt * 32 + 16 + i));
                                const float g = fmaxf(0.0f, fmaxf(qlo - rhi[u], rlo[u] - qhi));
                                const float gz = (i < Dq) ? g : 0.0f;
                                acc[qt] = fmaf(gz, gz, acc[qt]);
                            }
                        }
                    }
                    if (decltype(keep)::value && h0 == 0) {      // (both known once the loop is unrolled)
                        // the box, widened by more than the queries' float coordinates can be off (query_reach): 2^-22 of the
                        // larger of the two magnitudes covers 2^-24 |x| and this subtraction's own rounding; -inf / +inf past
                        // d (a gap of zero).  64 bytes per tile: (lo, lo, hi, hi) of dimensions 0-1, 2-3, 4-5, 6-7.
                        float blo[8], bhi[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const float mg = fmaf(fmaxf(__int_as_float(qabs_bits), fmaxf(fabsf(rlo[u]), fabsf(rhi[u]))), 0x1p-22f, 1e-37f);
                            blo[u] = (u < Dq) ? rlo[u] - mg : -__builtin_huge_valf();
                            bhi[u] = (u < Dq) ? rhi[u] + mg : __builtin_huge_valf();
                        }
                        v4f_t* bx = reinterpret_cast<v4f_t*>(wbuf) + lane * 4;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (2 * j < Dq) { v4f_t v = {blo[2 * j], blo[2 * j + 1], bhi[2 * j], bhi[2 * j + 1]}; bx[j] = v; }
                        }
                    }
                }
            }
        };
        // Second, sharper test of the tiles whose box is within reach of a query TILE's box: is it within reach of any single
        // QUERY (its own K-th distance, its own position)?  At d = 6 only one tile in six is (32 queries k-d neighbours of each
        // other still span a box whose corners no query is near, and the tile's bound is its WORST query's).  Lane = query
        // here: the tile's box comes out of LDS, where the lane that tested it left it (box_gap: widened by what a float
        // coordinate can be off), the query's coordinates out of LDS as floats, once per chunk; the gap is then a lower bound
        // like the box test's, and the sum is compared with the same 2^-18 allowance.
        // (compiled for every even number of dimensions DD >= d up to 8, chosen by one uniform switch: with d a run-time bound inside
        //  the loop every dimension costs two branches and the dimensions wait for each other -- 700 cycles per tile measured,
        //  against 200 here.  box_gap leaves -inf / +inf in the one dimension that may lie past d: its gap is zero.)
        auto query_reach_dd = [&](unsigned long long need, auto dd) __attribute__((always_inline)) -> unsigned long long {
            constexpr int DD = decltype(dd)::value;
            // (constant offsets from one address; DD - 1 may be d itself, the row after the last: read, and replaced by zero)
            float xf[DD];
            int ln = lane;
            asm volatile("" : "+v"(ln));                 // (not eight addresses computed before the walk and kept)
#pragma unroll
            for (int i = 0; i < DD; ++i) xf[i] = (float)xq[i * QPW + ln];
            xf[DD - 1] = (DD - 1 < D) ? xf[DD - 1] : 0.0f;
            const float thrf = (qwave0 + lane < nq) ? __double2float_ru(thrq[lane]) : -1.0f;     // (a padding query reaches nothing)
            const v4f_t* const bx = reinterpret_cast<const v4f_t*>(wbuf);
            unsigned long long keep = 0;
#ifndef MCE_H_PRUNE_QR_TILES
#define MCE_H_PRUNE_QR_TILES 4
#endif
            constexpr int NT = DD >= 8 ? 2 : MCE_H_PRUNE_QR_TILES;      // (eight dimensions x four tiles: 64 registers of boxes -- spills at three waves)
            while (need != 0) {
                // NT tiles per round (independent chains; the last round tests its last tile more than once).  Their boxes:
                // the same address in every lane, one 16-byte LDS read per pair of dimensions -- (lo, lo, hi, hi), as the
                // packed fp32 operations below want them
                int t[NT];
                t[0] = (int)__builtin_ctzll(need);
                need &= need - 1;
#pragma unroll
                for (int k = 1; k < NT; ++k) {
                    t[k] = need != 0 ? (int)__builtin_ctzll(need) : t[k - 1];
                    need &= need - 1;                   // (0 & anything = 0)
                }
                v4f_t A[NT][DD / 2];
#pragma unroll
                for (int k = 0; k < NT; ++k)
#pragma unroll
                    for (int j = 0; j < DD / 2; ++j) A[k][j] = bx[t[k] * 4 + j];
                v2f_t sm[NT];
#pragma unroll
                for (int k = 0; k < NT; ++k) { sm[k].x = 0.0f; sm[k].y = 0.0f; }
#pragma unroll
                for (int j = 0; j < DD / 2; ++j) {
                    const v2f_t x2 = {xf[2 * j], xf[2 * j + 1]};
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        const v2f_t u = A[k][j].xy - x2, w = x2 - A[k][j].zw;
                        const v2f_t g = {fmaxf(0.0f, fmaxf(u.x, w.x)), fmaxf(0.0f, fmaxf(u.y, w.y))};
                        sm[k] = g * g + sm[k];
                    }
                }
#pragma unroll
                for (int k = 0; k < NT; ++k)
                    if (__ballot(!((sm[k].x + sm[k].y) * (1.0f - 0x1p-18f) > thrf)) != 0) keep |= 1ull << t[k];
            }
            return keep;
        };
        // (d <= 8 only -- where the pruned walk is chosen at all, capi_common.hpp: kPruneAutoMinRows; sixteen dimensions' worth of
        //  boxes and coordinates do not fit the registers of the larger list capacities)
        auto query_reach = [&](unsigned long long need) __attribute__((always_inline)) -> unsigned long long {
            switch ((D + 1) >> 1) {
            case 1: return query_reach_dd(need, std::integral_constant<int, 2>());
            case 2: return query_reach_dd(need, std::integral_constant<int, 4>());
            case 3: return query_reach_dd(need, std::integral_constant<int, 6>());
            case 4: return query_reach_dd(need, std::integral_constant<int, 8>());
            default: return need;
            }
        };
        int pend = 0;
        int st_tiles = 0, st_chunks = 0;
#if MCE_PRUNE_PROF
        long long pt_pass_n = 0, pt_pass_t = 0, pt_nopass_t = 0;
        long long w2_win = 0, w2_box = 0, w2_qr = 0, w2_nbox = 0, w2_ntile = 0;      // MCE_PRUNE_PROF == 2: the walk phase's parts instead of stage / mul / drain
        long long pt_walk = 0, pt_stage = 0, pt_mul = 0, pt_drain = 0; const long long pt_begin = clock64(); long long pt_t = pt_begin;
#define MCE_PT(acc) do { const long long n_ = clock64(); acc += n_ - pt_t; pt_t = n_; } while (0)
#else
#define MCE_PT(acc) do {} while (0)
#endif
        int e = pr_sub * 64;                        // (heavy blocks: sub-wave s takes the windows s, s + S, ...)
        unsigned long long need = 0, cand = 0;
        int c = 0, win_c = 0;
        // Bootstrap (queries that ARE the references, same k-d order): the tiles next to the wave's own in
        // k-d order are spatial neighbours, so they go first -- the thresholds are then almost final before
        // the walk starts, far fewer pairs beat them later and far fewer tiles stay within reach.  The walk
        // skips these tiles.
        const bool same_order = (qperm == rperm);
        const int own_t = (int)(qwave0 / 32);
        const int boot_lo = same_order ? (own_t - kHPruneBoot > 0 ? own_t - kHPruneBoot : 0) : 0;
        const int boot_hi0 = own_t + QT + kHPruneBoot;
        const int n_rtiles = (int)((nr + 31) / 32);
        const int boot_hi = same_order ? (boot_hi0 < n_rtiles ? boot_hi0 : n_rtiles) : 0;
        // own tiles first, then alternately right and left: k -> own_t + (k odd ? (k+1)/2 : -k/2)
        const int boot_n = same_order ? 2 * kHPruneBoot + QT : 0;
        int boot_k = 0;
        bool boot_flush = false;
        // Bootstrap for a SEPARATE query set (no shared order to start from): of the first window of the list,
        // first only the tiles whose box overlaps a query tile's (xb_state 0), then the thresholds settle and
        // the window is walked again properly, skipping what was done (xb_mask, per window entry, in LDS).
        int xb_state = same_order ? 2 : 0;
        bool xb_win0 = false;
        unsigned long long xb_full = 0;
        int cur_bsel = 0;
        unsigned long long* const xb_mask = reinterpret_cast<unsigned long long*>(thrq + QPW);
        // (a wave of padding queries only would never tighten anything and walk the whole list)
        if (qwave0 >= nq) { boot_k = boot_n; e = list_len; }
        for (;;) {
            // ---- collect: fill the pending list from the current chunk's mask, moving down the list
            while (pend < kBatch) {
                if (boot_k < boot_n) {
                    const int room = kBatch - pend;                 // the next `room` positions of the sequence
                    const int k = boot_k + lane;
                    const int id = own_t + ((k & 1) ? (k + 1) / 2 : -(k / 2));         // own tiles first, then outward (ascending tile number was measured and dropped, round 3)
                    const bool take = lane < room && k < boot_n && id >= boot_lo && id < boot_hi;   // (off the ends: skipped)
                    const unsigned long long tm = __ballot(take);
                    if (take) wl[pend + __builtin_amdgcn_mbcnt_hi((unsigned)(tm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)tm, 0))] = id;
                    pend += __builtin_popcountll(tm);
                    boot_k += room;
                    if (boot_k >= boot_n && pend > 0) { boot_flush = true; break; }    // settle the thresholds before walking
                    continue;
                }
                if (need == 0) {
                    if (xb_state == 0 && xb_win0 && cand == 0) {
                        // the overlapping tiles of the first window are collected: multiply them, settle the
                        // thresholds, then walk that window again with the real test
                        xb_state = 2;
                        cand = xb_full;
                        if (pend > 0) { boot_flush = true; break; }
                    }
                    if (cand == 0) {
                        xb_win0 = false;
                        if (e >= list_len) break;
                        // next window of 64 list entries, lane l <-> entry e + l: still within this wave's
                        // reach?  chunk box within reach of one of the query tiles?
                        const int idx = e + lane;
                        const bool in = idx < list_len;
                        const float cd = in ? mydist[idx] : __builtin_huge_valf();
                        win_c = in ? mylist[idx] : 0;
                        // the list is ordered by BANDS of the bound (prune.hpp): an entry beyond this wave's reach is skipped,
                        // and the walk ends at the first one whose band FLOOR is -- every later entry is at least that far
                        const bool far = cd > mythr;
                        const bool stop = prune_band_floor(cd) > mythr;
                        const float* bb = cbox_r + (int64_t)win_c * (2 * D);
                        float acc[QT];
#if MCE_PRUNE_PROF == 2
                        const long long tw0_ = clock64();
#endif
                        box_gap(bb, 1, acc, std::true_type());
#if MCE_PRUNE_PROF == 2
                        asm volatile("" :: "v"(acc[0]), "v"(acc[1]));
                        w2_win += clock64() - tw0_;
#endif
                        bool reach = false;
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) reach |= !(acc[qt] * (1.0f - 0x1p-18f) > Tq[qt]);
                        cand = __ballot(in && !far && reach);
                        // (the same per-query test on the window's CHUNK boxes: about half of the chunks whose box is within reach
                        //  of a query tile's box are within reach of no single query -- their 64 tile boxes are then never fetched.
                        //  Not in the first window of a separate query set: that one is walked twice, see xb_state.)
                        if (xb_state != 0 && cand != 0) cand = query_reach(cand);
                        e = (__ballot(in && stop) != 0) ? list_len : e + 64 * pr_step;
                        st_chunks += 1;
                        if (xb_state == 0) {                           // first window of a separate query set
                            float amin = acc[0];
#pragma unroll
                            for (int qt = 1; qt < QT; ++qt) amin = fminf(amin, acc[qt]);
                            xb_full = cand;
                            xb_win0 = true;
                            xb_mask[lane] = 0;
                            const unsigned long long ov = __ballot(in && amin == 0.0f) & cand;
                            cand = ov ? ov : (cand & (0ull - cand));   // (no overlapping chunk box: the nearest one)
                        }
                        if (cand == 0) continue;
                    }
                    cur_bsel = (int)__builtin_ctzll(cand);
                    cand &= cand - 1;
                    c = __shfl(win_c, cur_bsel, 64);
                    const float* cb = tbox_r + (int64_t)c * (2 * D * PCT) + lane;
                    // fp32 is enough for a rigorous bound: a gap fl(a - b) of two floats is within 2^-24
                    // of exact, the sum of <= 15 squares within 2^-19; the comparison gives back 2^-18
                    float acc[QT];
#if MCE_PRUNE_PROF == 2
                    const long long tb0_ = clock64();
#endif
                    box_gap(cb, PCT, acc, std::true_type());
#if MCE_PRUNE_PROF == 2
                    asm volatile("" :: "v"(acc[0]), "v"(acc[1]));
                    w2_box += clock64() - tb0_; w2_nbox += 1;
#endif
                    if (xb_state == 0) {
                        float amin = acc[0];
#pragma unroll
                        for (int qt = 1; qt < QT; ++qt) amin = fminf(amin, acc[qt]);
                        need = __ballot(amin == 0.0f);
                        if (need == 0 && xb_full == (1ull << cur_bsel)) need = __ballot(amin < __builtin_huge_valf());   // a lone chunk: all of it
                        if (lane == 0) xb_mask[cur_bsel] = need;
                    } else {
                        const int tile_id = c * PCT + lane;
                        const bool booted = same_order ? (tile_id >= boot_lo && tile_id < boot_hi) : (xb_win0 && ((xb_mask[cur_bsel] >> lane) & 1ull));
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) need |= __ballot(!booted && !(acc[qt] * (1.0f - 0x1p-18f) > Tq[qt]));
#if MCE_PRUNE_PROF == 2
                        const long long tq0_ = clock64(); w2_ntile += __builtin_popcountll(need);
#endif
                        if (need != 0) need = query_reach(need);
#if MCE_PRUNE_PROF == 2
                        w2_qr += clock64() - tq0_;
#endif
                    }
                    st_tiles += __builtin_popcountll(need);
                    if (need == 0) continue;
                }
                // append the lowest (kBatch - pend) set bits
                const bool mine = (need >> lane) & 1ull;
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0));
                const bool take = mine && rank < kBatch - pend;
                if (take) wl[pend + rank] = c * PCT + lane;
                const unsigned long long taken = __ballot(take);
                need &= ~taken;
                pend += __builtin_popcountll(taken);
            }
            MCE_PT(pt_walk);
            if (pend == 0) break;
            // ---- multiply the pending tiles
            // global -> the wave's LDS slice by LDS-DMA (all tiles of the batch in flight together, no registers in between).
            // A DMA's landing is ordered for the ISSUING wave's ds_read by its own vmcnt wait alone (for other waves' reads a
            // workgroup barrier must follow, dma_barrier above -- these workgroups are one wave).
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                if (u < pend) {
                    const int id = __builtin_amdgcn_readfirstlane(wl[u]);
                    const _Float16* src = Yh + (int64_t)id * (KST * 512) + lane * 8;
#pragma unroll
                    for (int ks = 0; ks < KST; ++ks)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ks * 512),
                                                         (__attribute__((address_space(3))) void*)(wbuf + (u * KST + ks) * 1024), 16, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            MCE_PT(pt_stage);
            {
                // (no mfma/gate overlap across tiles: measured, the multiply phase is bound by the gate and
                // enqueue work, and a second accumulator tile costs 50+ spilled registers here.)  The gate
                // reads the accumulators through inline asm (v_min3_f32), which the compiler's hazard
                // recogniser does not cover: the MFMA result latency (16 passes -> 18 wait states) is
                // waited out by hand, tied to the registers so it cannot be scheduled away.
                const char* lbuf = wbuf + lane * 16;
                v8h a0[KST];
#pragma unroll 1
                for (int u = 0; u < pend; ++u) {
                    load_a(lbuf + (u * KST) * 1024, a0);
                    mfma_tile(a0, accA);
                    static_assert(QT == 2 || QT == 4, "wait-state asm names the accumulator tiles");
                    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(accA[0]), "+v"(accA[1]));
#if MCE_PRUNE_PROF
                    const long long q0_ = qcount; const long long tp0 = clock64();
#endif
                    gate_exact(accA, __builtin_amdgcn_readfirstlane(wl[u]) * 32);
#if MCE_PRUNE_PROF
                    { const long long dt = clock64() - tp0; if (qcount != q0_) { pt_pass_n += 1; pt_pass_t += dt; } else pt_nopass_t += dt; }
#endif
                }
            }
            pend = 0;
            MCE_PT(pt_mul);
            if (qcount >= kPruneDrainTrigger || boot_flush) drain();
            boot_flush = false;
            MCE_PT(pt_drain);
        }
        if (lane == 0) {                                   // launch totals: chunks tested and tiles multiplied, per wave
            double* stat = const_cast<double*>(params);
            unsafeAtomicAdd(stat + HP_STAT_CHUNKS, (double)st_chunks);
            unsafeAtomicAdd(stat + HP_STAT_TILES, (double)st_tiles);
#if MCE_PRUNE_PROF
#if MCE_PRUNE_PROF == 2
            pt_stage = w2_box; pt_mul = w2_qr; pt_drain = w2_win; gx_iter = w2_nbox; pt_pass_n = w2_ntile;
#endif
#if MCE_PRUNE_PROF == 3
            pt_stage = gx_drain_t; pt_drain = gx_drain_n;      // (drains inside gate_exact's loop: cycles, count)
#endif
            unsafeAtomicAdd(stat + 8, (double)pt_walk); unsafeAtomicAdd(stat + 9, (double)pt_stage); unsafeAtomicAdd(stat + 10, (double)pt_mul);
            unsafeAtomicAdd(stat + 11, (double)pt_drain); unsafeAtomicAdd(stat + 12, (double)(clock64() - pt_begin)); unsafeAtomicAdd(stat + 13, (double)gx_stage_t); unsafeAtomicAdd(stat + 14, (double)pt_pass_n); unsafeAtomicAdd(stat + 15, (double)pt_pass_t); unsafeAtomicAdd(stat + 7, (double)gx_iter);
#endif
        }
    }
#undef MCE_SWEEP_CHUNK
    if constexpr (!PRUNE && SYM != 1) process(accB, jbB, rB);
    if constexpr (SYM != 1) drain();

    // ---- write the lists: lane l owns wave-local queries nl*64 + l (coalesced over lanes) ----
#pragma unroll
    for (int nl = 0; nl < NL; ++nl) {
        const int64_t q = qwave0 + nl * 64 + lane;
        if (PRUNE && pr_sub > 0) {
            // a heavy block's extra sub-wave: its lists go to the side arrays, column = position among the heavy blocks
            const int64_t hcols = (int64_t)hv_n * QPW;
            const int64_t hc = (int64_t)pr_k * QPW + nl * 64 + lane;
            double* const hd = const_cast<double*>(lo_d);
            int* const hi_ = const_cast<int*>(lo_i);
#pragma unroll
            for (int k = 0; k < KCAP; ++k) {
                hd[((int64_t)(pr_sub - 1) * KCAP + k) * hcols + hc] = k < LC ? own_d[nl][k < LC ? k : 0] : INF;
                hi_[((int64_t)(pr_sub - 1) * KCAP + k) * hcols + hc] = k < LC ? own_i[nl][k < LC ? k : 0] : -1;
            }
            continue;
        }
#pragma unroll
        for (int k = 0; k < KCAP; ++k) {
            const int64_t o = ((int64_t)split * KCAP + k) * nq_pad + q;
            part_d[o] = k < LC ? own_d[nl][k < LC ? k : 0] : INF;           // (rows LC .. KCAP - 1 of the arrays: empty)
            part_i[o] = k < LC ? own_i[nl][k < LC ? k : 0] : -1;
        }
    }
    if constexpr (SYM == 2) {
        // hand the lists to the block's next unit
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(sym.done + qblk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wg_us && tid == 0) wg_us[blockIdx.x] = (float)(wall_clock64() - wg_t0) * 0.01f;      // 100 MHz counter
}

}  // namespace mce
