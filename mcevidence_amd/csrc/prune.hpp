// prune.hpp -- interface of prune.hip: spatial pruning for low-dimensional, large reference sets
// (SURVEY.md section 8f.2(ii)).
//
// The brute-force sweep of knn_f16.hpp visits every (query block, reference chunk) pair.  For
// D <~ 10 and N >~ 10^6 almost all of them are provably irrelevant: if both point sets are laid
// out in k-d order, a block of 512 queries sits in a small box, and a chunk of references whose
// box is farther from it than the block's current worst K-th distance cannot contribute.  This
// module produces, entirely on the device and stream-ordered (no host synchronisation):
//   * a k-d ordering of the references and of the queries down to cells of 32 rows (= one MFMA
//     tile): recursive median splits cycling through the dimensions, implemented as one radix
//     sort per tree level on the key (node id, coordinate);
//   * the reordered fp64 copies Ys / Xs and the permutations back to the caller's row numbers;
//   * the bounding box of every 32-row tile (floats rounded outward);
//   * per query block, the list of ALL reference chunks sorted by box-to-box distance (a rigorous
//     lower bound, rounded down) -- the search kernel walks it, stops at the first entry whose
//     bound exceeds the block's current threshold, and inside a chunk multiplies only the tiles
//     whose box is within reach of one of the wave's two query tiles.
// Results are those of the exhaustive search, bit for bit (same exact distances, same tie-breaks
// on the caller's row numbers).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mce {

constexpr int kPruneTileRows = 32;                        // k-d cells = MFMA reference tiles = query tiles
constexpr int kPruneMaxDim = 15;                          // KST = 1 variants only
constexpr int64_t kPruneMaxPairs = (int64_t)1 << 30;      // nqblk * nchunk list entries
constexpr int kPruneWavesPerBlock = 8;                    // == kHWaves (knn_f16.hpp): the walk's workgroups serve one wave of a query block each

struct PruneLayout {
    size_t perm_r = 0, perm_q = 0;          // int32 [nr_pad], [nq_pad]: sorted position -> caller's row (-1: padding)
    size_t keys_a = 0, keys_b = 0;          // uint64 [max(nr_pad, nq_pad)] sort keys (ping-pong)
    size_t vals_b = 0;                      // int32 [max(nr_pad, nq_pad)]
    size_t Ys = 0, Xs = 0;                  // double [nr * d], [nq * d] reordered rows
    size_t cf32 = 0;                        // float [d][n]: the coordinates of the set being sorted, one plane per dimension (aliases Xs where that is large enough)
    size_t tbox_r = 0, tbox_q = 0;          // float [tiles][2][d] (lo | hi), rounded outward
    size_t tboxT_r = 0;                     // float [nchunk][2][d][tiles per chunk]: the kernel's layout
    size_t box_r = 0, box_q = 0;            // float [nchunk][2][d], [nqblk][2][d]
    size_t bkey_a = 0, bkey_b = 0, bval_a = 0, border = 0;   // the walk's waves (block * 8 + wave) by descending box size: [nqblk * 8]
    size_t list_d_b = 0, list_c_b = 0;      // float / int32 [nqblk * nchunk]: every block's chunks in bands of ascending box distance
    size_t tmp = 0, tmp_bytes = 0;          // rocPRIM scratch
    size_t total = 0;
};

// pure function of its arguments (host-only size queries)
int prune_layout(int64_t nq, int64_t nq_pad, int nqblk, int64_t nr, int64_t nr_pad, int64_t nchunk, int d, PruneLayout& L);

// The chunk lists are ordered by BANDS of the lower bound (the float's exponent and its top kPruneBandMantissa mantissa bits:
// 3 % wide), not entry by entry: prune_band_floor(d) <= d is the same for every entry of a band and never decreases along a
// list, so the walk may stop at the first entry whose band floor exceeds its reach (knn_f16.hpp) -- and a counting sort by
// band in one pass replaces a segmented radix sort of nqblk * nchunk pairs.
constexpr int kPruneBandMantissa = 5;
__host__ __device__ inline float prune_band_floor(float d2)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(__float_as_uint(d2) & ~((1u << (23 - kPruneBandMantissa)) - 1u));
#else
    union { float f; unsigned u; } v; v.f = d2; v.u &= ~((1u << (23 - kPruneBandMantissa)) - 1u); return v.f;
#endif
}

struct PruneOut {
    const double* Xs = nullptr;
    const double* Ys = nullptr;
    const int* qperm = nullptr;
    const int* rperm = nullptr;
    const int* clist = nullptr;             // [nqblk][nchunk] chunk ids, in bands of ascending box distance (prune_band_floor)
    const float* cdist = nullptr;           // [nqblk][nchunk] matching lower bounds on the squared distance
    const float* tbox_r = nullptr;          // reference tile boxes, [chunk][2][d][tiles per chunk]
    const float* tbox_q = nullptr;          // query tile boxes
    const float* cbox_r = nullptr;          // reference chunk boxes [chunk][2][d]
    const int* border = nullptr;            // dispatch order of the walk's waves (block * 8 + wave), largest box first
};

// same_set: the queries ARE the reference rows (same pointer, nq == nr): one ordering serves both
// perm_ready (round 6): the k-d order of the references is in the workspace already (prune_prepare_part + the ranks' all-reduce)
hipError_t prune_prepare(const double* dX, int64_t nq, const double* dY, int64_t nr, int d, bool same_set, int qpb,
                         int chunk_rows, int64_t nq_pad, int nqblk, int64_t nr_pad, int64_t nchunk, char* ws,
                         const PruneLayout& L, hipStream_t st, PruneOut& out, bool perm_ready = false);
// Distributed k-d preparation (round 6; reference: the `fit` of MCEvidence.py:1100-1101, which every rank of a multi-GPU run repeated
// in full -- 4.85 ms of sorts of the 7.8 ms a rank of C5 spends preparing): rank `part` of nparts = 2, 4, 8, ... runs the sorts that
// settle the tree's top log2(nparts) levels over all rows and everything below them over ITS subtree only; perm_r then holds the
// final order in [seg_lo, seg_hi) and zeros elsewhere -- the ranks' arrays add up to the single-GPU permutation, bit for bit.
// seg_lo = 0, seg_hi = nr_pad: the whole order was made here (a count that is not a power of two, a tree too shallow).
hipError_t prune_prepare_part(const double* dY, int64_t nr, int d, int64_t nr_pad, char* ws, const PruneLayout& L, int part, int nparts,
                              hipStream_t st, int64_t& seg_lo, int64_t& seg_hi);

// ---- symmetric sweep (knn_f16.hpp, "Symmetric sweep"): rows sorted by distance from the mean + its scratch ----
struct SymLayout {
    size_t perm = 0;                        // int32 [n_pad]: sorted position -> caller's row (-1: padding)
    size_t keys_a = 0, keys_b = 0;          // uint32 [n_pad] sort keys (ping-pong)
    size_t vals_a = 0;                      // int32 [n_pad]
    size_t Ys = 0;                          // double [n * d] sorted rows (queries and references)
    size_t thr = 0;                         // uint64 [n_pad]
    size_t rrow = 0;                        // uint32 [n_pad]
    size_t rtile = 0;                       // float [n_pad / 32]
    size_t slots = 0;                       // uint64 [n_pad][kcap]
    size_t bucket_cnt = 0, bucket_flag = 0, done = 0; // int32 [nqblk] each (contiguous: one memset)
    size_t bucket = 0;                      // SymEntry [nqblk][cap]
    size_t tmp = 0, tmp_bytes = 0;          // rocPRIM scratch
    size_t total = 0;
    int cap = 0;                            // bucket entries per query block
};
// pure function of its arguments.  per_row: bucket entries per row (cap = per_row * qpb)
int sym_layout(int64_t n, int64_t n_pad, int nqblk, int d, int kcap, int qpb, int per_row, SymLayout& L);
// sorts the rows of Y[n, d] by their distance from `center` (device pointer, d doubles), writes the sorted copy and the
// permutation; stream-ordered, no host synchronisation
hipError_t sym_prepare(const double* dY, int64_t n, int d, const double* center, int64_t n_pad, char* ws, const SymLayout& L,
                       hipStream_t st);

}  // namespace mce
