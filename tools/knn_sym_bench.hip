// knn_sym_bench.hip -- developer microbench for the symmetric sweep (knn_f16_kernel<.., SYM>): prepass + sweep with the
// per-wave statistics on; not part of the product.  Rows are sorted by distance from the mean on the host.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/knn_sym_bench.hip -o tools/knn_bench_sym
// -DPANEL=1 (default): the sweep runs on knn_panel_kernel (knn_panel.hpp; per-wave statistics with -DMCE_PANEL_STATS=1; -DMCE_PANEL_ABL=1 / 2:
// gates never pass / no gate); -DPANEL=0: on the SYM = 2 instantiation of knn_f16_kernel (the round-2 kernel; its statistics build is gone).
#ifndef PANEL
#define PANEL 1
#endif
#include "../mcevidence_amd/csrc/f16_prep.hpp"
#include "../mcevidence_amd/csrc/knn_panel.hpp"
#include "../mcevidence_amd/csrc/pack_refs.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>
#ifndef DIM
#define DIM 27
#endif
#ifndef KCAP
#define KCAP 12
#endif
#ifndef KSEL
#define KSEL 9
#endif
using namespace mce;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 200000;
    const int reps = argc > 2 ? atoi(argv[2]) : 2;
    const int seed_rows = argc > 3 ? atoi(argv[3]) : 32768;
    const int per_row = argc > 4 ? atoi(argv[4]) : 6 * KSEL + 24;
    const int sorted = argc > 5 ? atoi(argv[5]) : 1;
    const int panel = argc > 6 ? atoi(argv[6]) : 48;
    constexpr int D = DIM;
    constexpr int KST = f16_ksteps(D);
    constexpr int CT = f16_chunk_tiles(KST);            // the prepass's chunks
    constexpr int PCT = panel_chunk_tiles(KST);         // the panel sweep's
    const int qpb = f16_qpb(KCAP);
    const int nqblk = (int)((n + qpb - 1) / qpb);
    const int64_t nq_pad = (int64_t)nqblk * qpb;
    const int64_t nchunk = (n + CT * 32 - 1) / (CT * 32);
    const int64_t nrow_pad = (nchunk * CT * 32 + (int64_t)PCT * 32 * CT - 1) / ((int64_t)PCT * 32 * CT) * ((int64_t)PCT * 32 * CT);     // whole chunks of either kind
    std::vector<double> h((size_t)n * D), hsorted((size_t)n * D);
    std::mt19937_64 g(1); std::normal_distribution<double> nd;
    for (auto& v : h) v = nd(g);
    std::vector<int> perm(nq_pad, -1);
    {
        std::vector<double> nrm(n);
        for (int64_t i = 0; i < n; ++i) { double s = 0; for (int k = 0; k < D; ++k) s += h[i * D + k] * h[i * D + k]; nrm[i] = s; }
        std::vector<int> idx(n); std::iota(idx.begin(), idx.end(), 0);
        if (sorted) std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return nrm[a] < nrm[b]; });
        for (int64_t i = 0; i < n; ++i) { perm[i] = idx[i]; std::copy(&h[(size_t)idx[i] * D], &h[(size_t)idx[i] * D] + D, &hsorted[(size_t)i * D]); }
    }
    double *X, *pd, *center, *msum, *params, *qinfo; int *pi, *dperm; _Float16 *Yh, *Xh;
    CK(hipMalloc(&X, sizeof(double) * n * D));
    CK(hipMalloc(&Yh, 2 * nrow_pad * 16 * KST)); CK(hipMalloc(&Xh, 2 * nq_pad * 16 * KST));
    CK(hipMalloc(&qinfo, 16 * nq_pad)); const size_t pbytes = 256 + (size_t)nqblk * 40 * 16 * 64; CK(hipMalloc(&params, pbytes)); CK(hipMalloc(&center, 3 * 512)); CK(hipMalloc(&msum, 8 * kStatStride * 256));
    CK(hipMalloc(&dperm, 4 * nq_pad));
    const size_t nl = (size_t)KCAP * nq_pad;
    CK(hipMalloc(&pd, sizeof(double) * nl)); CK(hipMalloc(&pi, sizeof(int) * nl));
    CK(hipMemcpy(X, hsorted.data(), sizeof(double) * n * D, hipMemcpyHostToDevice));
    CK(hipMemcpy(dperm, perm.data(), 4 * nq_pad, hipMemcpyHostToDevice));
    SymParams sp;
    const int cap = per_row * qpb;
    CK(hipMalloc(&sp.thr, 8 * nq_pad)); CK(hipMalloc(&sp.rrow, 4 * nq_pad)); CK(hipMalloc(&sp.rtile, 4 * (nq_pad / 32)));
    CK(hipMalloc(&sp.slots, 8 * nq_pad * KCAP)); CK(hipMalloc(&sp.bucket_cnt, 12 * nqblk)); sp.bucket_flag = sp.bucket_cnt + nqblk; sp.done = sp.bucket_cnt + 2 * nqblk; sp.panel = panel;
    const int nunits = sym_unit_count(nqblk, kHWaves * kHQT, panel * CT, (int)((n + 31) / 32) + (int)(((n + 31) / 32) & 1));
    CK(hipMalloc(&sp.bucket, (size_t)16 * nqblk * cap)); sp.cap = cap;
    col_stats_partial_kernel<<<kMeanBlocks, kMeanThreads>>>(X, n, D, msum);
    col_stats_final_kernel<<<1, 64>>>(msum, n, D, center, center + 64);
    CK(hipMemset(params, 0, pbytes));
    f16_scale_kernel<<<1, 64>>>(center + 64, nullptr, params);
    const int64_t rpb = 4 * (64 / (2 * KST));
    f16_pack_refs_kernel<<<(unsigned)std::min<int64_t>((nrow_pad + rpb - 1) / rpb, 2048), 256>>>(X, n, D, KST, nrow_pad, center, params, Yh);
    f16_pack_queries_kernel<<<(unsigned)((nq_pad + rpb - 1) / rpb), 256>>>(X, n, nq_pad, D, KST, center, params, Xh, qinfo);
    CK(hipDeviceSynchronize());
    constexpr size_t LDS = f16_lds_bytes(KST, KCAP, true);
    auto kpre = knn_f16_kernel<KST, KCAP, false, false, 1>;
    auto kern = knn_f16_kernel<KST, KCAP, false, false, 2>;
    auto kpanel = knn_panel_kernel<KST, KCAP>;
    CK(hipFuncSetAttribute((const void*)kpre, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    CK(hipFuncSetAttribute((const void*)kpanel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)panel_lds_bytes(KST)));
    PanelArgs pa;
    pa.Yh = Yh; pa.Xh = Xh; pa.qinfo = qinfo; pa.params = params; pa.X = X; pa.Y = X; pa.rperm = dperm; pa.part_d = pd; pa.part_i = pi;
    pa.nq = n; pa.nr = n; pa.nq_pad = nq_pad; pa.self_offset = 0; pa.D = D; pa.ksel = KSEL; pa.self_exclude = 1; pa.spin_limit = 1 << 21; pa.debug = 0;
    pa.sym = sp;
    pa.geom.qb_lo = 0; pa.geom.qb_hi = nqblk; pa.geom.tpb = kHWaves * kHQT; pa.geom.ct = PCT; pa.geom.tpp = panel * CT / PCT * PCT; pa.geom.sym_on = 1;
    pa.geom.ntiles = (int)((n + 31) / 32) + (int)(((n + 31) / 32) & 1);
    if (getenv("PARTS")) {      // one rank's share of a multi-GPU symmetric partition: PARTS=<nparts>,<part>
        int np = 1, pt = 0; sscanf(getenv("PARTS"), "%d,%d", &np, &pt);
        pa.geom.qb_lo = (int)((int64_t)nqblk * pt / np); pa.geom.qb_hi = (int)((int64_t)nqblk * (pt + 1) / np);
    }
    const int npanel_units = panel_unit_count(pa.geom);
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const int seed_mode = getenv("SEED_MODE") ? atoi(getenv("SEED_MODE")) : (KST >= 2 ? 1 : 0);      // as capi_common.hpp: kSymSeedMode
    const int seed_cfg0 = f16_seed_cfg(nchunk, CT, KSEL + 1, seed_rows, 8, MCE_H_SEED_TG);
    const int seed_cfg = seed_cfg0 ? (seed_cfg0 | (seed_mode << 28)) : 0;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemset(sp.bucket_cnt, 0, 12 * nqblk));
        CK(hipMemset((char*)params + 128, 0, pbytes - 128));
        CK(hipEventRecord(e0));
        kpre<<<nqblk, kHThreads, LDS>>>(Yh, nchunk, 1, Xh, qinfo, params, X, X, n, n, D, nq_pad, nqblk, 1, 0, KSEL, pd, pi,
                                        (const int*)nullptr, (const float*)nullptr, 0, dperm, (const int*)nullptr, (const float*)nullptr,
                                        (const float*)nullptr, (const float*)nullptr, 0, 1, (const int*)nullptr, (const double*)nullptr, (const int*)nullptr, seed_cfg, sp, (float*)nullptr);
        CK(hipEventRecord(e1));
        if (PANEL && !getenv("SYM_REPAIR_ALL")) kpanel<<<npanel_units, kHThreads, panel_lds_bytes(KST)>>>(pa);
        else if (!getenv("SYM_REPAIR_ALL")) kern<<<nunits, kHThreads, LDS>>>(Yh, nchunk, 1, Xh, qinfo, params, X, X, n, n, D, nq_pad, nqblk, 1, 0, KSEL, pd, pi,
                                        (const int*)nullptr, (const float*)nullptr, 0, dperm, (const int*)nullptr, (const float*)nullptr,
                                        (const float*)nullptr, (const float*)nullptr, 0, 1, (const int*)nullptr, (const double*)nullptr, (const int*)nullptr, 0, sp, (float*)nullptr);
        CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
        if (getenv("SYM_REPAIR_ALL")) {      // experiment: the exhaustive column-only sweep over the sorted rows (every block "repaired")
            std::vector<int> ones(nqblk, 1);
            CK(hipMemcpy(sp.bucket_flag, ones.data(), 4 * nqblk, hipMemcpyHostToDevice));
            auto krep = knn_f16_kernel<KST, KCAP, false, false, 3>;
            CK(hipFuncSetAttribute((const void*)krep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
            hipEvent_t e3; CK(hipEventCreate(&e3));
            CK(hipEventRecord(e2));
            krep<<<nqblk, kHThreads, LDS>>>(Yh, nchunk, 1, Xh, qinfo, params, X, X, n, n, D, nq_pad, nqblk, 1, 0, KSEL, pd, pi,
                                            (const int*)nullptr, (const float*)nullptr, 0, dperm, (const int*)nullptr, (const float*)nullptr,
                                            (const float*)nullptr, (const float*)nullptr, 0, 1, (const int*)nullptr, (const double*)nullptr, (const int*)nullptr, 0, sp, (float*)nullptr);
            CK(hipEventRecord(e3)); CK(hipEventSynchronize(e3));
            float ms3; CK(hipEventElapsedTime(&ms3, e2, e3));
            printf("exhaustive column-only sweep over the sorted rows, thresholds from the prepass: %.2f ms\n", ms3);
        }
        float ms1, ms2; CK(hipEventElapsedTime(&ms1, e0, e1)); CK(hipEventElapsedTime(&ms2, e1, e2));
         printf("%s panel=%d units=%d ", PANEL ? "panel-kernel" : "f16-kernel", panel, PANEL ? npanel_units : nunits); printf("D=%d KST=%d KCAP=%d K=%d n=%lld grid=%d seed=%dx%d sorted=%d: prepass %.2f ms  sweep %.2f ms\n", D, KST, KCAP, KSEL, (long long)n, nqblk,
               seed_cfg & 0xffff, (seed_cfg >> 16) & 0xfff, sorted, ms1, ms2);
    }
    {
        std::vector<int> cnt(2 * nqblk);
        CK(hipMemcpy(cnt.data(), sp.bucket_cnt, 8 * nqblk, hipMemcpyDeviceToHost));
        long long tot = 0, mx = 0, fl = 0;
        for (int b = 0; b < nqblk; ++b) { tot += cnt[b]; mx = std::max<long long>(mx, cnt[b]); fl += cnt[nqblk + b]; }
        printf("buckets: %.1f entries per row on average, fullest %.1f per row (cap %d), %lld overflowed\n", (double)tot / n, (double)mx / qpb, per_row, fl);
    }
#if MCE_PANEL_STATS && PANEL
    {
        const size_t nw = (size_t)npanel_units * 8;
        std::vector<double> hs(nw * 16);
        CK(hipMemcpy(hs.data(), (char*)params + 128, nw * 128, hipMemcpyDeviceToHost));
        double m[16] = {0};
        for (size_t w = 0; w < nw; ++w) for (int k = 0; k < 8; ++k) { m[k] += hs[w * 8 + k] / nw; m[8 + k] += hs[nw * 8 + w * 8 + k] / nw; }
        printf("   drain phases (cycles per drain): A (exact distances) %.0f  R (row side) %.0f  B + publish + gates %.0f\n", m[8] / (m[0] > 0 ? m[0] : 1), m[9] / (m[0] > 0 ? m[0] : 1), m[10] / (m[0] > 0 ? m[0] : 1));
        printf("per wave and unit (mean): drains %.1f  enq %.0f  redo tiles %.2f  event (tile, query tile)s %.0f | cycles: events %.3g (%.1f %%)  drains %.3g (%.1f %%)  prologue %.3g (%.1f %%)  kernel %.3g\n",
               m[0], m[1], m[2], m[3], m[4], 100 * m[4] / m[6], m[5], 100 * m[5] / m[6], m[7], 100 * m[7] / m[6], m[6]);
        printf("   waiting at the chunk barrier (incl. the DMA's landing): %.3g cycles per wave and unit (%.1f %%)\n", m[11], 100 * m[11] / m[6]);
        printf("   per event %.0f cycles, per drain %.0f cycles, per queued pair %.1f drain cycles; enq per query (whole search) %.1f\n", m[4] / (m[3] > 0 ? m[3] : 1), m[5] / (m[0] > 0 ? m[0] : 1), m[5] / (m[1] > 0 ? m[1] : 1), m[1] * nw / 64.0 / n * 64.0 / 64.0);
    }
#endif
    return 0;
}
