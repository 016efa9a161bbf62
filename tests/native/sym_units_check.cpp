// sym_units_check.cpp -- CPU check of the symmetric sweep's unit enumeration (mcevidence_amd/csrc/sym_types.hpp):
// every (panel, block) whose ranges intersect appears exactly once, panel by panel, blocks descending; the tile ranges of a
// block's units tile its range [0, min(tpb (a+1), ntiles)) without gaps or overlaps; the count is the grid size.
// Build + run: g++ -std=c++17 -O1 -I mcevidence_amd/csrc tests/native/sym_units_check.cpp -o /tmp/sym_units_check && /tmp/sym_units_check
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sym_types.hpp"

int main()
{
    long checked = 0;
    const int tpbs[] = {16};
    for (int tpb : tpbs)
        for (int nqblk = 1; nqblk <= 70; nqblk += (nqblk < 12 ? 1 : 7))
            for (int tpp : {12, 16, 24, 48, 96, 100, 2304})
                for (int cut = 0; cut < 3; ++cut) {
                    // rows end somewhere inside the last block (even tile count, as the kernel rounds it)
                    int ntiles = tpb * nqblk - (cut == 0 ? 0 : cut == 1 ? 2 : tpb - 2);
                    if (ntiles <= tpb * (nqblk - 1)) ntiles = tpb * (nqblk - 1) + 2;
                    const int n = mce::sym_unit_count(nqblk, tpb, tpp, ntiles);
                    std::vector<int> next_lo(nqblk, 0), units_of(nqblk, 0);
                    int last_p = 0, last_a = nqblk;
                    for (int u = 0; u < n; ++u) {
                        int p, a, lo, hi;
                        mce::sym_unit_decode(u, nqblk, tpb, tpp, p, a);
                        mce::sym_unit_tiles(p, a, tpb, tpp, ntiles, lo, hi);
                        if (a < 0 || a >= nqblk || p < 0) { printf("bad unit %d -> (%d,%d)\n", u, p, a); return 1; }
                        if (p < last_p || (p == last_p && a >= last_a)) { printf("order broken at unit %d\n", u); return 1; }
                        if (p != last_p) last_a = nqblk;
                        if (a >= last_a) { printf("order broken at unit %d\n", u); return 1; }
                        last_p = p; last_a = a;
                        if (hi <= lo) { printf("empty unit %d (%d,%d) [%d,%d) nqblk=%d tpp=%d ntiles=%d\n", u, p, a, lo, hi, nqblk, tpp, ntiles); return 1; }
                        if (lo != next_lo[a]) { printf("gap/overlap: block %d expects lo %d, unit %d has %d\n", a, next_lo[a], u, lo); return 1; }
                        if (units_of[a] != p) { printf("block %d: unit number %d is panel %d\n", a, units_of[a], p); return 1; }   // the hand-over counter
                        next_lo[a] = hi; units_of[a] += 1;
                        ++checked;
                    }
                    for (int a = 0; a < nqblk; ++a) {
                        const int want = tpb * (a + 1) < ntiles ? tpb * (a + 1) : ntiles;
                        if (next_lo[a] != want) { printf("block %d covered to %d, expected %d (nqblk=%d tpp=%d ntiles=%d)\n", a, next_lo[a], want, nqblk, tpp, ntiles); return 1; }
                    }
                }
    printf("ok %ld units\n", checked);
    // ---- panel sweep units (PanelGeom): a range of blocks, symmetric inside the range, column side only outside
    long pchecked = 0;
    for (int nqblk = 1; nqblk <= 60; nqblk += (nqblk < 10 ? 1 : 5))
        for (int ct : {12, 24, 48})
            for (int panel : {1, 2, 5, 96})
                for (int sym_on = 0; sym_on < 2; ++sym_on)
                    for (int parts = 1; parts <= 4; ++parts)
                        for (int part = 0; part < parts; ++part)
                            for (int cut = 0; cut < 2; ++cut) {
                                mce::PanelGeom g;
                                g.tpb = 16; g.ct = ct; g.tpp = panel * ct; g.sym_on = sym_on;
                                g.ntiles = g.tpb * nqblk - (cut ? 6 : 0);
                                if (g.ntiles <= g.tpb * (nqblk - 1)) g.ntiles = g.tpb * (nqblk - 1) + 2;
                                g.qb_lo = (int)((long)nqblk * part / parts);
                                g.qb_hi = (int)((long)nqblk * (part + 1) / parts);
                                if (g.qb_hi <= g.qb_lo) continue;
                                const int n = mce::panel_unit_count(g);
                                std::vector<std::vector<char>> cover(nqblk, std::vector<char>(g.ntiles, 0));
                                std::vector<int> seq(nqblk, 0);
                                int last_p = 0, last_a = g.qb_hi;
                                for (int u = 0; u < n; ++u) {
                                    int p, a, lo, hi;
                                    mce::panel_unit_decode(u, g, p, a);
                                    mce::panel_unit_tiles(p, a, g, lo, hi);
                                    if (a < g.qb_lo || a >= g.qb_hi) { printf("panel: bad block %d\n", a); return 1; }
                                    if (p != last_p) last_a = g.qb_hi;
                                    if (p < last_p || a >= last_a) { printf("panel: order broken at unit %d\n", u); return 1; }
                                    last_p = p; last_a = a;
                                    if (hi <= lo || (lo & 1) || (hi & 1) || hi > g.ntiles) { printf("panel: bad range [%d,%d) unit %d (p %d a %d) nqblk %d ct %d panel %d sym %d part %d/%d\n", lo, hi, u, p, a, nqblk, ct, panel, sym_on, part, parts); return 1; }
                                    // one interval of chunks; only the first chunk may start inside a chunk
                                    if (mce::panel_unit_seq(p, a, g) != seq[a]) { printf("panel: block %d unit %d has seq %d, expected %d\n", a, u, mce::panel_unit_seq(p, a, g), seq[a]); return 1; }
                                    seq[a] += 1;
                                    for (int t = lo; t < hi; ++t) { if (cover[a][t]) { printf("panel: tile %d of block %d covered twice\n", t, a); return 1; } cover[a][t] = 1; }
                                    ++pchecked;
                                }
                                for (int a = g.qb_lo; a < g.qb_hi; ++a)
                                    for (int t = 0; t < g.ntiles; ++t) {
                                        const int tb = t / g.tpb;
                                        const bool want = !sym_on || tb <= a || tb >= g.qb_hi;
                                        if ((bool)cover[a][t] != want) { printf("panel: block %d tile %d covered %d, expected %d (nqblk %d ct %d panel %d part %d/%d ntiles %d)\n", a, t, cover[a][t], (int)want, nqblk, ct, panel, part, parts, g.ntiles); return 1; }
                                    }
                            }
    printf("ok %ld panel units\n", pchecked);
    // ---- the all-pairs-once partition (PanelGeom.blk_stride): over all ranks every block appears on exactly one rank, with the
    // single-GPU units of that block -- tiles [0, min(tpb (a + 1), ntiles)) -- handed over in sequence
    long schecked = 0;
    for (int nqblk = 2; nqblk <= 61; nqblk += (nqblk < 12 ? 1 : 7))
        for (int ct : {12, 48})
            for (int panel : {1, 3, 96})
                for (int W = 2; W <= 9; ++W)
                    for (int cut = 0; cut < 2; ++cut) {
                        const int ntiles = 16 * nqblk - (cut ? 6 : 0);
                        std::vector<int> owner_seen(nqblk, -1), next_lo(nqblk, 0), seq(nqblk, 0);
                        long units_all = 0;
                        for (int r = 0; r < W; ++r) {
                            mce::PanelGeom g;
                            g.tpb = 16; g.ct = ct; g.tpp = panel * ct; g.ntiles = ntiles; g.sym_on = 1;
                            g.qb_lo = 0; g.qb_hi = nqblk; g.blk_first = r; g.blk_stride = W;
                            const int n = mce::panel_unit_count(g);
                            units_all += n;
                            int lp = 0, la = 1 << 30;
                            for (int u = 0; u < n; ++u) {
                                int p, a, lo, hi;
                                mce::panel_unit_decode(u, g, p, a);
                                mce::panel_unit_tiles(p, a, g, lo, hi);
                                if (a < 0 || a >= nqblk || a % W != r) { printf("stride: unit %d of rank %d/%d is block %d\n", u, r, W, a); return 1; }
                                if (p != lp) la = 1 << 30;
                                if (p < lp || a >= la) { printf("stride: order broken at unit %d\n", u); return 1; }
                                lp = p; la = a;
                                if (hi <= lo || lo != next_lo[a] || (lo & 1) || (hi & 1)) { printf("stride: block %d unit %d range [%d,%d), expected lo %d (W %d r %d nqblk %d tpp %d)\n", a, u, lo, hi, next_lo[a], W, r, nqblk, g.tpp); return 1; }
                                if (mce::panel_unit_seq(p, a, g) != seq[a]) { printf("stride: block %d seq\n", a); return 1; }
                                next_lo[a] = hi; seq[a] += 1; owner_seen[a] = r;
                                ++schecked;
                            }
                        }
                        for (int a = 0; a < nqblk; ++a) {
                            const int want = 16 * (a + 1) < ntiles ? 16 * (a + 1) : ntiles;
                            if (owner_seen[a] != a % W || next_lo[a] != want) { printf("stride: block %d owner %d covered to %d, expected %d (W %d nqblk %d)\n", a, owner_seen[a], next_lo[a], want, W, nqblk); return 1; }
                        }
                        // the ranks' units together are the single-GPU launch
                        mce::PanelGeom g1;
                        g1.tpb = 16; g1.ct = ct; g1.tpp = panel * ct; g1.ntiles = ntiles; g1.sym_on = 1; g1.qb_lo = 0; g1.qb_hi = nqblk;
                        if (units_all != mce::panel_unit_count(g1)) { printf("stride: %ld units over the ranks, %d on one GPU\n", units_all, mce::panel_unit_count(g1)); return 1; }
                    }
    printf("ok %ld strided units\n", schecked);
    return 0;
}
