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
    int slot_stride = 0;                   // slots per row in `slots` when it differs from the kernel's KCAP (0: KCAP) -- the prepass of a second pass
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

// ---------------------------------------------------------------------------
// Units of the PANEL sweep (knn_panel.hpp), which generalises the above to a RANGE of query blocks [qb_lo, qb_hi):
//   * sym_on = 0 (cross evidence, query shards): every block sweeps every tile [0, ntiles), column side only;
//   * sym_on = 1 (auto evidence; one rank's share of it): the tiles of the blocks qb_lo..qb_hi-1 carry the row-side gate
//     and block a takes them only up to its own (the pair of blocks {a, b}, qb_lo <= b < a, is handled by a alone);
//     every other tile -- the blocks below qb_lo and from qb_hi on, i.e. the rows another rank owns -- is swept by every
//     block, column side only.  With [qb_lo, qb_hi) = [0, nqblk) this is the single-GPU symmetric sweep.
// Block a's tiles are therefore [0, hi_a) and [r2, ntiles), hi_a = sym_on ? min(tpb (a + 1), r1) : r1,
// r1 = sym_on ? min(tpb qb_hi, ntiles) : ntiles, r2 = sym_on ? tpb qb_hi : ntiles.  Region 1 = [0, r1) is cut into
// panels of tpp tiles from tile 0; region 2 = [r2, ntiles) into panels whose boundaries lie at r2c + k tpp, r2c = r2
// rounded down to a whole chunk (ct tiles), so that only a region's first chunk can start inside a chunk.  Units are
// numbered panel by panel (region 1, then region 2) and, within a panel, from the LAST block down; a block's units
// hand its lists on in that order (panel_unit_seq = how many came before).
// tests/native/sym_units_check.cpp checks the functions against each other.
// ---------------------------------------------------------------------------
// one unit of a launch whose units come from a table (PanelArgs.units): query block, tiles [t_lo, t_hi), how many units of its
// chain come before it, the chain's hand-over counter, its list set
struct PanelUnit { int qblk, t_lo, t_hi, useq, chain, list_set, pad0, pad1; };
struct PanelGeom {
    int qb_lo = 0, qb_hi = 0;      // query blocks of the launch
    int tpb = 16;                  // tiles per query block
    int tpp = 0;                   // tiles per panel (a whole number of chunks)
    int ct = 0;                    // tiles per chunk
    int ntiles = 0;                // 32-row tiles holding reference rows, rounded up to even
    int sym_on = 0;
    int blk_first = 0, blk_stride = 0;   // blk_stride = W > 1: only the blocks blk_first, blk_first + W, ... of [qb_lo, qb_hi) have units (one rank's share of
                                   // the all-pairs-once partition, below); needs qb_lo = 0 and sym_on = 1
    int nsplit = 1;                // S > 1 (same launches): a block's units form S independent CHAINS -- the panels p = s, s + S, ... hand on list set s
                                   // ([S][KCAP][nq_pad], merged afterwards) -- because a rank with fewer blocks than the chip has workgroup slots is
                                   // as slow as its longest block's chain.  (blk_stride and nsplit are read by panel_unit_table_kernel -- sym_exchange.hpp --
                                   // and the host; the sweep kernel takes such a launch's units from the table.)
};
MCE_HD inline int panel_r1(const PanelGeom& g) { return g.sym_on ? (g.tpb * g.qb_hi < g.ntiles ? g.tpb * g.qb_hi : g.ntiles) : g.ntiles; }
MCE_HD inline int panel_r2(const PanelGeom& g) { return g.sym_on ? g.tpb * g.qb_hi : g.ntiles; }
MCE_HD inline int panel_r2c(const PanelGeom& g) { return panel_r2(g) / g.ct * g.ct; }
MCE_HD inline int panel_n1(const PanelGeom& g) { return (panel_r1(g) + g.tpp - 1) / g.tpp; }
MCE_HD inline int panel_n2(const PanelGeom& g) { return panel_r2(g) < g.ntiles ? (g.ntiles - panel_r2c(g) + g.tpp - 1) / g.tpp : 0; }
MCE_HD inline int panel_hi_a(const PanelGeom& g, int a)
{
    const int r1 = panel_r1(g);
    return g.sym_on ? (g.tpb * (a + 1) < r1 ? g.tpb * (a + 1) : r1) : r1;
}
// first block with a unit in panel p (p < n1: region 1)
MCE_HD inline int panel_amin(const PanelGeom& g, int p)
{
    if (!g.sym_on || p >= panel_n1(g)) return g.qb_lo;
    const int amin = (int)(((int64_t)p * g.tpp) / g.tpb);
    return amin > g.qb_lo ? amin : g.qb_lo;
}
// The all-pairs-once partition of auto evidence over W ranks (round 5; capi_apo.hpp): rank r owns the sorted blocks r, r + W,
// r + 2W, ... and runs exactly the single-GPU units of THOSE blocks -- block a sweeps the tiles [0, tpb (a + 1)) with both
// gates on -- so every pair of blocks {a, b}, b < a, is multiplied once per NODE, by the owner of a, from the side of the
// rows farther from the mean (the orientation the symmetric sweep is built for: the row side then belongs to the denser
// rows, whose bounds are tight).  The row-side candidates it finds for rows of blocks it does not own travel to their
// owners afterwards.  Work per block grows with a; the cyclic ownership balances it to within one block's share.
// number of blocks with units among [amin, qb_hi)
MCE_HD inline int panel_nown(const PanelGeom& g, int amin)
{
    if (g.blk_stride <= 1) return g.qb_hi - amin;
    const int total = g.qb_hi > g.blk_first ? (g.qb_hi - g.blk_first + g.blk_stride - 1) / g.blk_stride : 0;
    const int skip = amin <= g.blk_first ? 0 : (amin - g.blk_first + g.blk_stride - 1) / g.blk_stride;
    return total > skip ? total - skip : 0;
}
MCE_HD inline int panel_unit_count(const PanelGeom& g)
{
    int total = 0;
    const int np = panel_n1(g) + panel_n2(g);
    for (int p = 0; p < np; ++p) total += panel_nown(g, panel_amin(g, p));
    return total;
}
MCE_HD inline void panel_unit_decode(int u, const PanelGeom& g, int& p, int& a)
{
    for (p = 0;; ++p) {
        const int cnt = panel_nown(g, panel_amin(g, p));
        if (u < cnt) {
            // (from the last block with units down)
            a = g.blk_stride <= 1 ? g.qb_hi - 1 - u : g.blk_first + g.blk_stride * (panel_nown(g, 0) - 1 - u);
            return;
        }
        u -= cnt;
    }
}
MCE_HD inline void panel_unit_tiles(int p, int a, const PanelGeom& g, int& lo, int& hi)
{
    const int n1 = panel_n1(g);
    if (p < n1) {
        const int hi_a = panel_hi_a(g, a);
        lo = p * g.tpp;
        hi = (p + 1) * g.tpp < hi_a ? (p + 1) * g.tpp : hi_a;
    } else {
        const int r2 = panel_r2(g), r2c = panel_r2c(g), k = p - n1;
        lo = k == 0 ? r2 : r2c + k * g.tpp;
        hi = r2c + (k + 1) * g.tpp < g.ntiles ? r2c + (k + 1) * g.tpp : g.ntiles;
    }
}
// units of block a before its unit in panel p
MCE_HD inline int panel_unit_seq(int p, int a, const PanelGeom& g)
{
    const int n1 = panel_n1(g);
    if (p < n1) return p;
    return (panel_hi_a(g, a) + g.tpp - 1) / g.tpp + (p - n1);
}

}  // namespace mce
