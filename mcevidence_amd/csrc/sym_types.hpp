// sym_types.hpp -- argument block of the symmetric sweep (knn_f16.hpp, "Symmetric sweep"), shared by the kernel
// header, the dispatch table and the host code.
#pragma once
#include <stdint.h>

namespace mce {

struct SymEntry { double d2; int src; int row; };      // row-side candidate: caller row `src` at squared distance d2 from sorted row `row`
struct SymParams {
    unsigned long long* thr = nullptr;     // [nq_pad] bit pattern of the published bound on the K-th squared distance (input units)
    unsigned* rrow = nullptr;              // [nq_pad] bit pattern of the row-side gate constant R_j (float > 0; padding rows 0)
    float* rtile = nullptr;                // [nq_pad / 32] max of rrow over a tile
    unsigned long long* slots = nullptr;   // [nq_pad][KCAP] bit patterns of the K smallest row-side squared distances
    int* bucket_cnt = nullptr;             // [nqblk]
    int* bucket_flag = nullptr;            // [nqblk] 1: the bucket overflowed, the block is searched again exhaustively
    SymEntry* bucket = nullptr;            // [nqblk][cap]
    int cap = 0;
    int* done = nullptr;                   // [nqblk] units of the block finished so far (its lists are handed from unit to unit)
    int panel = 0;                         // chunks per panel of reference rows (a unit = one block's queries x one panel)
};

#if defined(__HIPCC__)
#define MCE_HD __host__ __device__
#else
#define MCE_HD
#endif

// Units of the symmetric sweep: unit = (panel p, block a) for every block whose range of tiles [0, tpb (a + 1)) reaches into
// panel p = tiles [p tpp, (p + 1) tpp), numbered panel by panel and, within a panel, from the LAST block down.  A block's
// units hand its lists on in panel order, so unit (p + 1, a) waits for unit (p, a): with the blocks descending the two
// are a whole panel's worth of units apart (ascending: 144 fewer at 1M x 27 -- the last panels, which hold fewer units than
// the chip has CUs, then ran as a chain of waits), and a search of a single panel starts its longest blocks first.
// Measured at d = 27, sweep kernel, ascending -> descending: 1 M rows 41.5 -> 40.8 ms, 500 k 12.8 -> 12.1, 200 k 4.14 -> 2.95.
// ntiles: 32-row tiles that hold reference rows (the ranges are clipped there).  tests/native/sym_units_check.cpp checks
// the functions against each other.
MCE_HD inline int sym_unit_count(int nqblk, int tpb, int tpp, int ntiles)
{
    int total = 0;
    for (int p = 0; (int64_t)p * tpp < ntiles; ++p) total += nqblk - (int)(((int64_t)p * tpp) / tpb);
    return total;
}
// unit number -> (panel, block); the block's units before this one are those of the panels below: p of them
MCE_HD inline void sym_unit_decode(int u, int nqblk, int tpb, int tpp, int& p, int& a)
{
    for (p = 0;; ++p) {
        const int amin = (int)(((int64_t)p * tpp) / tpb);      // blocks a >= amin reach into panel p
        const int cnt = nqblk - amin;
        if (u < cnt) { a = nqblk - 1 - u; return; }
        u -= cnt;
    }
}
// tiles [lo, hi) of unit (p, a)
MCE_HD inline void sym_unit_tiles(int p, int a, int tpb, int tpp, int ntiles, int& lo, int& hi)
{
    const int hi_a = tpb * (a + 1) < ntiles ? tpb * (a + 1) : ntiles;
    lo = p * tpp;
    hi = (p + 1) * tpp < hi_a ? (p + 1) * tpp : hi_a;
}

}  // namespace mce
