// capi_search.hpp -- part of capi_search.hpp: run_search() (pack + the search launches of a plan; leaves the lists in the workspace)
// and launch_merge() (list merge, optional distance output, fused volume / weight reduction).  Reference: MCEvidence.py:1093-1117.
#pragma once
namespace {

double ln_unit_ball(int d) { return 0.5 * d * std::log(M_PI) - std::lgamma(1.0 + 0.5 * d); }

// the row-side candidates of the blocks [qb_lo, qb_hi) -- their buckets -- folded into their lists
// (the blocks qb_lo, qb_lo + stride, ... below qb_hi)
hipError_t launch_sym_merge(int KCAP, double* pd, int* pi, int64_t nq_pad, const mce::SymParams& sym, int qb_lo, int qb_hi, hipStream_t st, int stride = 1)
{
    const int nb = qb_hi > qb_lo ? (qb_hi - qb_lo + stride - 1) / stride : 0;
    if (nb <= 0) return hipSuccess;
    const dim3 g((unsigned)nb), b(mce::kSymMergeThreads);
    static_assert(mce::kSymMergeThreads == mce::f16_qpb(4), "one merge block per query block");
    switch (KCAP) {
        case 4: hipLaunchKernelGGL(mce::sym_merge_kernel<4>, g, b, 0, st, pd, pi, nq_pad, sym.bucket_cnt, sym.bucket_flag, sym.bucket, sym.cap, qb_lo, stride); break;
        case 8: hipLaunchKernelGGL(mce::sym_merge_kernel<8>, g, b, 0, st, pd, pi, nq_pad, sym.bucket_cnt, sym.bucket_flag, sym.bucket, sym.cap, qb_lo, stride); break;
        case 12: hipLaunchKernelGGL(mce::sym_merge_kernel<12>, g, b, 0, st, pd, pi, nq_pad, sym.bucket_cnt, sym.bucket_flag, sym.bucket, sym.cap, qb_lo, stride); break;
        default: hipLaunchKernelGGL(mce::sym_merge_kernel<16>, g, b, 0, st, pd, pi, nq_pad, sym.bucket_cnt, sym.bucket_flag, sym.bucket, sym.cap, qb_lo, stride); break;
    }
    return hipGetLastError();
}

// pack + search; leaves the lane/split lists in the workspace
int run_search(Plan& p, const double* dX, int64_t nq, const double* dY, int64_t nr, int32_t d, int32_t K,
               int32_t self_mode, int64_t self_offset, char* ws, hipStream_t st)
{
    p.sym_active = false;
    // the all-pairs-once partition runs in two calls with a collective between them (capi_apo.hpp): 1 = statistics, sort, packing and
    // the prepass of the rank's own blocks, then return (the bounds are all-reduced); 2 = the sweep on what phase 1 left in the workspace
    const int phase = p.apo ? p.apo_phase : 0;
    double* pd = reinterpret_cast<double*>(ws + p.off_pd);
    int* pi = reinterpret_cast<int*>(ws + p.off_pi);
    if (p.generic) {
        const size_t lds = mce::generic_lds_bytes();        // (static in the kernel; reported in mce_last_kernel())
        hipLaunchKernelGGL(mce::knn_generic_kernel, dim3((unsigned)p.nqblk), dim3(mce::kGenThreads), 0, st, dX, nq, dY, nr, (int)d,
                           (int)K, p.nq_pad, (self_mode == MCE_SELF_EXCLUDE) ? 1 : 0, self_offset, pd, pi);
        MCE_HIP(hipGetLastError());
        snprintf(g_last_kernel, sizeof(g_last_kernel), "knn_generic_kernel grid=%d block=%d lds=%zu", p.nqblk, mce::kGenThreads, lds);
        return MCE_OK;
    }
    double* center = reinterpret_cast<double*>(ws + p.off_center);
    double* msum = reinterpret_cast<double*>(ws + p.off_msum);
    double* box_y = center + mce::kMaxDimPad;
    double* box_x = center + 2 * mce::kMaxDimPad;
    if (p.vl) {                 // (the long-row sweep: its own column means, any d)
        hipLaunchKernelGGL(mce::long_col_mean_partial_kernel, dim3(mce::kLongMeanBlocks), dim3(256), 0, st, dY, nr, (int)d, mce::kLongMeanBlocks, msum);
        MCE_HIP(hipGetLastError());
        hipLaunchKernelGGL(mce::long_col_mean_final_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, st, msum, nr, (int)d, mce::kLongMeanBlocks, center);
        MCE_HIP(hipGetLastError());
    } else if (d > MCE_MAX_DIM) {      // (the fp64 sweep's wide form, 64 <= d <= 127: means only, 128 of the 192 doubles at `center`)
        static_assert(3 * mce::kMaxDimPad >= 128 && mce::kStatStride >= 128, "wide column means");
        hipLaunchKernelGGL(mce::col_mean_wide_partial_kernel, dim3(mce::kMeanBlocks), dim3(256), 0, st, dY, nr, (int)d, msum);
        MCE_HIP(hipGetLastError());
        hipLaunchKernelGGL(mce::col_mean_wide_final_kernel, dim3(1), dim3(128), 0, st, msum, nr, (int)d, center);
        MCE_HIP(hipGetLastError());
    } else if (phase != 2) {
        hipLaunchKernelGGL(mce::col_stats_partial_kernel, dim3(mce::kMeanBlocks), dim3(mce::kMeanThreads), 0, st, dY, nr, (int)d, msum);
        MCE_HIP(hipGetLastError());
        hipLaunchKernelGGL(mce::col_stats_final_kernel, dim3(1), dim3(64), 0, st, msum, nr, (int)d, center, box_y);
        MCE_HIP(hipGetLastError());
    }
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
    if (p.vl) {
        // ---- the long-row fp64 sweep (128 <= d <= 1024; knn_long.hpp): both sets packed in MFMA fragment order, k blocks of 32 dimensions ----
        double* yf = reinterpret_cast<double*>(ws + p.off_yf);
        double* xf = reinterpret_cast<double*>(ws + p.off_xf);
        double* xn = reinterpret_cast<double*>(ws + p.off_xn);
        hipLaunchKernelGGL(mce::pack_refs_kernel, dim3((unsigned)((p.nrow_pad + threads - 1) / threads)), dim3(threads), 0, st, dY, nr, (int)d, p.KS, p.nrow_pad, center, yf);
        MCE_HIP(hipGetLastError());
        const int64_t qe = p.nq_pad * (int64_t)p.KS;
        hipLaunchKernelGGL(mce::long_pack_queries_kernel, dim3((unsigned)((qe + 255) / 256)), dim3(256), 0, st, dX, nq, (int)d, p.KS, p.nq_pad, center, xf);
        MCE_HIP(hipGetLastError());
        hipLaunchKernelGGL(mce::long_query_norms_kernel, dim3((unsigned)((p.nq_pad + 255) / 256)), dim3(256), 0, st, dX, nq, (int)d, p.nq_pad, center, xn);
        MCE_HIP(hipGetLastError());
        mce::LongArgs a;
        a.Yf = yf; a.Xf = xf; a.xn = xn;
        a.nchunk_total = p.nchunk; a.rsplit = p.rsplit; a.KSP = p.KS; a.KSB = mce::long_ksb(d);
        a.nq = nq; a.nq_pad = p.nq_pad; a.nqblk = p.nqblk;
        a.self_exclude = (self_mode == MCE_SELF_EXCLUDE) ? 1 : 0;
        a.self_offset = self_offset;
        a.ksel = p.ksel > K ? p.ksel : K;
        a.part_d = pd; a.part_i = pi;
        int rc = prof_begin();
        if (rc != MCE_OK) return rc;
        MCE_HIP(p.vl->launch(a, st));
        rc = prof_end();
        if (rc != MCE_OK) return rc;
        snprintf(g_last_kernel, sizeof(g_last_kernel), "%s grid=%d block=%d lds=%zu qt=%d ct=%d rsplit=%d ksp=%d", p.vl->name, p.nqblk * p.rsplit, mce::kThreads,
                 p.vl->lds_bytes, p.QT, p.CT, p.rsplit, p.KS);
        g_last_flops_main = g_last_flops_all = (double)p.nq_pad * (double)p.nrow_pad * 2.0 * 4.0 * p.KS;
        return MCE_OK;
    }
    if (p.vd) {
        // ---- the deep fp16 filter (64 <= d <= 127; knn_deep.hpp) + exact fp64 refine -------------------------------------------
        _Float16* yh = reinterpret_cast<_Float16*>(ws + p.off_yh);
        _Float16* xh = reinterpret_cast<_Float16*>(ws + p.off_xh);
        double* qinfo = reinterpret_cast<double*>(ws + p.off_qinfo);
        double* params = reinterpret_cast<double*>(ws + p.off_params);
        MCE_HIP(mce::zero_async(params, mce::HP_COUNT * sizeof(double), st));
        const bool separate_queries = !(dX >= dY && dX + (size_t)nq * d <= dY + (size_t)nr * d);
        // radius about the references' mean (the wide column means above), over both sets; power-of-two scale; packing
        hipLaunchKernelGGL(mce::f16_radius_rows_kernel, dim3((unsigned)std::min<int64_t>((nr + 255) / 256, 2048)), dim3(256), 0, st, dY, nr, (int)d, center, params);
        MCE_HIP(hipGetLastError());
        if (separate_queries && nq > 0) {
            hipLaunchKernelGGL(mce::f16_radius_rows_kernel, dim3((unsigned)std::min<int64_t>((nq + 255) / 256, 2048)), dim3(256), 0, st, dX, nq, (int)d, center, params);
            MCE_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(mce::f16_scale_from_radius_kernel, dim3(1), dim3(1), 0, st, params);
        MCE_HIP(hipGetLastError());
        const int64_t rows_per_block = 4 * (64 / (2 * p.KST));          // 4 waves x R rows
        const int64_t pack_blocks = std::min<int64_t>((p.nrow_pad + rows_per_block - 1) / rows_per_block, 2048);   // grid-stride
        hipLaunchKernelGGL(mce::f16_pack_refs_kernel, dim3((unsigned)pack_blocks), dim3(256), 0, st, dY, nr, (int)d, p.KST, p.nrow_pad, center, params, yh);
        MCE_HIP(hipGetLastError());
        hipLaunchKernelGGL(mce::f16_pack_queries_kernel, dim3((unsigned)((p.nq_pad + rows_per_block - 1) / rows_per_block)), dim3(256), 0, st,
                           dX, nq, p.nq_pad, (int)d, p.KST, center, params, xh, qinfo);
        MCE_HIP(hipGetLastError());
        mce::DeepArgs a;
        a.Yh = yh; a.Xh = xh; a.qinfo = qinfo; a.params = params; a.X = dX; a.Y = dY; a.part_d = pd; a.part_i = pi;
        a.nq = nq; a.nr = nr; a.nq_pad = p.nq_pad; a.self_offset = self_offset; a.nchunk_total = p.nchunk;
        a.D = d; a.ksel = K; a.self_exclude = (self_mode == MCE_SELF_EXCLUDE) ? 1 : 0; a.nqblk = p.nqblk; a.rsplit = p.rsplit;
        a.debug = read_tuning().panel_debug;
        // seed phase: K (+ 1) groups of tg tiles from the start of every split -- about 24 k rows, at most a quarter of the smallest
        // split (a half if a quarter does not hold one tile per group); MCE_F16_SEED_ROWS=0: none (tests)
        {
            const int64_t tiles_split = (p.nchunk / p.rsplit) * p.CT;
            const int G = K + a.self_exclude;
            const Tuning tun = read_tuning();
            const int64_t want = (tun.f16_seed_rows >= 0 ? tun.f16_seed_rows : MCE_H_SEED_ROWS) / 32;
            int64_t tg = std::min<int64_t>(want, tiles_split / 4) / G;
            if (tg < 1) tg = std::min<int64_t>(want, tiles_split / 2) / G;
            a.seed_tg = (int)std::max<int64_t>(tg, 0);
        }
        int rc = prof_begin();
        if (rc != MCE_OK) return rc;
        double seed_rows = (double)a.seed_tg * (K + a.self_exclude) * 32.0 * p.rsplit;
        if (p.twopass) {
            // lists [2 * rsplit][16][nq_pad]: pass 1 fills splits 0 .. rsplit - 1 with each split's 16 nearest (its seed phase bounds the
            // 16th), pass 2 the next K - 16 beyond them into rsplit .. 2 rsplit - 1 (its seed phase bounds the K-th OVERALL: K (+ 1) groups);
            // the merge takes the K best of all
            const int64_t tiles_split = (p.nchunk / p.rsplit) * p.CT;
            const Tuning tun = read_tuning();
            const int64_t want = (tun.f16_seed_rows >= 0 ? tun.f16_seed_rows : MCE_H_SEED_ROWS) / 32;
            auto tg_for = [&](int G) {
                int64_t tg = std::min<int64_t>(want, tiles_split / 4) / G;
                if (tg < 1) tg = std::min<int64_t>(want, tiles_split / 2) / G;
                return (int)std::max<int64_t>(tg, 0);
            };
            const int tg2 = a.seed_tg;                  // (computed above for K (+ 1) groups)
            a.ksel = 16;
            a.seed_groups = 16 + a.self_exclude;
            a.seed_tg = tg_for(a.seed_groups);
            const int tg1 = a.seed_tg;
            MCE_HIP(p.vd->launch(a, st));
            a.lo_d = pd;
            a.lo_i = pi;
            a.part_d = pd + (size_t)p.rsplit * p.KCAP * (size_t)p.nq_pad;
            a.part_i = pi + (size_t)p.rsplit * p.KCAP * (size_t)p.nq_pad;
            a.ksel = K - 16;
            a.seed_groups = K + a.self_exclude;
            a.seed_tg = tg2;
            MCE_HIP(p.vd->launch_lower(a, st));
            seed_rows = ((double)tg1 * (16 + a.self_exclude) + (double)tg2 * (K + a.self_exclude)) * 32.0 * p.rsplit;
        } else {
            MCE_HIP(p.vd->launch(a, st));
        }
        rc = prof_end();
        if (rc != MCE_OK) return rc;
        g_last_flops_main = g_last_flops_all = (double)p.nq_pad * ((p.twopass ? 2.0 : 1.0) * (double)p.nrow_pad + seed_rows) * 2.0 * 16.0 * p.KST;
        snprintf(g_last_kernel, sizeof(g_last_kernel), "%s grid=%d block=%d lds=%zu qt=%d ct=%d rsplit=%d%s seed=%dx%d", p.vd->name, p.nqblk * p.rsplit, mce::kHThreads,
                 p.vd->lds_bytes, p.QT, p.CT, p.rsplit, p.twopass ? " two passes" : "", K + a.self_exclude, a.seed_tg);
        return MCE_OK;
    }
    if (p.vh) {
        // ---- fp16 filter + exact fp64 refine ------------------------------------
        _Float16* yh = reinterpret_cast<_Float16*>(ws + p.off_yh);
        _Float16* xh = reinterpret_cast<_Float16*>(ws + p.off_xh);
        double* qinfo = reinterpret_cast<double*>(ws + p.off_qinfo);
        double* params = reinterpret_cast<double*>(ws + p.off_params);
        if (phase != 2) MCE_HIP(mce::zero_async(params, mce::HP_COUNT * sizeof(double), st));
        // queries that are literally rows of the reference buffer are inside its bounding box already
        bool separate_queries = !(dX >= dY && dX + (size_t)nq * d <= dY + (size_t)nr * d);
        const double* sX = dX;     // the rows the search reads: the caller's, or their k-d ordered copies
        const double* sY = dY;
        mce::PruneOut po;
        const bool use_sym = p.sym && dX == dY && nq == nr && self_offset == 0 && g_split_depth == 0;
        if (use_sym) {
            // rows by distance from the mean: a 32-row tile then holds rows of nearly equal K-th neighbour distance
            if (phase != 2) MCE_HIP(mce::sym_prepare(dY, nr, (int)d, center, p.nq_pad, ws + p.off_sym, p.sl, st));
            sX = sY = reinterpret_cast<const double*>(ws + p.off_sym + p.sl.Ys);
            separate_queries = false;
        }
        if (p.prune) {
            const bool same_set = (dX == dY && nq == nr);
            MCE_HIP(mce::prune_prepare(dX, nq, dY, nr, (int)d, same_set, mce::f16_qpb(p.KCAP), p.CT * 32, p.nq_pad, p.nqblk, p.nrow_pad,
                                       p.nchunk, ws + p.off_prune, p.pl, st, po, p.kd_ready && same_set));
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
        if (phase != 2) {
            hipLaunchKernelGGL(mce::f16_scale_kernel, dim3(1), dim3(64), 0, st, box_y, separate_queries ? box_x : (const double*)nullptr, params);
            MCE_HIP(hipGetLastError());
        }
        if (phase != 2) {
            const int64_t rows_per_block = 4 * (64 / (2 * p.KST));          // 4 waves x R rows
            const int64_t pack_blocks = std::min<int64_t>((p.nrow_pad + rows_per_block - 1) / rows_per_block, 2048);   // grid-stride
            // queries and references are ONE buffer in the same order (auto evidence: the symmetric sweep's sorted rows, the pruned walk's
            // k-d order over one set, or the caller's buffer itself): one packing pass writes both forms (f16_prep.hpp)
            const bool one_pass = sX == sY && nq == nr && p.nq_pad <= p.nrow_pad;
            hipLaunchKernelGGL(mce::f16_pack_refs_kernel, dim3((unsigned)pack_blocks), dim3(256), 0, st,
                               sY, nr, (int)d, p.KST, p.nrow_pad, center, params, yh, one_pass ? xh : (_Float16*)nullptr, one_pass ? qinfo : (double*)nullptr,
                               one_pass ? p.nq_pad : (int64_t)0);
            MCE_HIP(hipGetLastError());
            if (!one_pass) {
                hipLaunchKernelGGL(mce::f16_pack_queries_kernel, dim3((unsigned)((p.nq_pad + rows_per_block - 1) / rows_per_block)), dim3(256), 0, st,
                                   sX, nq, p.nq_pad, (int)d, p.KST, center, params, xh, qinfo);
                MCE_HIP(hipGetLastError());
            }
        }
        mce::KnnF16Args a;
        a.Yh = yh; a.nchunk_total = p.nchunk; a.rsplit = p.rsplit; a.Xh = xh; a.qinfo = qinfo; a.params = params;
        a.X = sX; a.Y = sY; a.nq = nq; a.nr = nr; a.D = d; a.nq_pad = p.nq_pad; a.nqblk = p.nqblk;
        a.self_exclude = (self_mode == MCE_SELF_EXCLUDE) ? 1 : 0;
        a.self_offset = self_offset; a.ksel = K; a.part_d = pd; a.part_i = pi;
        if (p.prune) {
            a.clist = po.clist; a.cdist = po.cdist; a.list_len = (int)p.nchunk; a.rperm = po.rperm; a.qperm = po.qperm;
            a.tbox_r = po.tbox_r; a.tbox_q = po.tbox_q; a.cbox_r = po.cbox_r; a.border = po.border;
            // (a part of a multi-GPU run: every nparts-th WAVE of the dispatch order)
            const int nw_total = p.nqblk * mce::kHWaves;
            if (p.nparts > 1) { a.qblk0 = p.part; a.qblk_stride = p.nparts; a.nqblk_run = (nw_total - p.part + p.nparts - 1) / p.nparts; }
            // heavy waves (the first of this launch's dispatch order): several workgroups each, lists folded afterwards
            const int nblk_run = a.nqblk_run ? a.nqblk_run : nw_total;       // waves of this launch
            int hv_n = 0, hv_S = 1;
            if (p.heavy_max > 0 && dX == dY && nq == nr) {
                // how many: the waves that would run longer than a fraction of the launch.  Measured at C5 (1 wave in 100 takes
                // 3.4x the mean, 1 in 1000 7.5x, 1 in 10 000 13x): ~700 of a 10-round launch, fewer of a longer one (it hides
                // longer waves) -- 1 / 2 / 4 / 8 GPUs: 0 / 184 / 367 / 700 waves per rank
                const double rounds = (double)nblk_run / 2048.0;
                const char* hv_env = getenv("MCE_PRUNE_HEAVY");
                if ((kPruneHeavyDefault || (hv_env && !strcmp(hv_env, "auto"))) && rounds < kPruneHeavyMaxRounds) {
                    const double f = std::min(1.0, kPruneHeavyFullRounds / std::max(rounds, 1.0));
                    hv_n = std::min(std::max((int)(kPruneHeavyCount * f), kPruneHeavyMinCount), std::min(p.heavy_max, nblk_run));
                    hv_S = kPruneHeavySplit;
                }
                if (const char* e = hv_env) {
                    int n_ = hv_n, s_ = kPruneHeavySplit;
                    if (sscanf(e, "%d,%d", &n_, &s_) >= 1) { hv_n = std::min(std::max(n_, 0), std::min(p.heavy_max, nblk_run)); hv_S = std::min(std::max(s_, 1), kPruneHeavyMaxSplit); }
                }
                if (hv_n == 0 || hv_S == 1) { hv_n = 0; hv_S = 1; }
            }
            double* hv_d = reinterpret_cast<double*>(ws + p.off_heavy);
            int* hv_i = reinterpret_cast<int*>(hv_d + (size_t)p.heavy_max * kPruneWaveQueries * (kPruneHeavyMaxSplit - 1) * p.KCAP);
            a.seed_cfg = hv_n | (hv_S << 24);
            a.lo_d = hv_n ? hv_d : nullptr;
            a.lo_i = hv_n ? hv_i : nullptr;
            // MCE_PRUNE_TIMES=<file> (diagnostic): the duration of every workgroup of the walk, by position in the dispatch
            // order -- where the tail of a launch is (tools/heavy_scan.py)
            const char* times_file = getenv("MCE_PRUNE_TIMES");
            const size_t n_wg = (size_t)(nblk_run + hv_n * (hv_S - 1));
            DevBuf wg_times;
            if (times_file && *times_file) { MCE_HIP(wg_times.alloc(n_wg * sizeof(float))); a.wg_us = wg_times.as<float>(); }
            int rc = prof_begin();
            if (rc != MCE_OK) return rc;
            // K <= 9 with 12-row list arrays: the instantiation that keeps nine entries in registers runs three waves per SIMD
            // instead of two (knn_f16.hpp, LC): C5 on one GPU 95.9 vs 124.7 ms.  MCE_PRUNE_LISTS=long / short: comparisons.
            const char* force_lists = getenv("MCE_PRUNE_LISTS");
            // (K = 10 likewise with ten entries: 10 M x 6 105.7 -> 83 ms; eleven spill inside the walk)
            mce::knn_f16_launch_fn short_fn = nullptr;
            int short_lc = 0;
            if (p.vh->launch_prune_short && K <= p.vh->prune_short_lc) { short_fn = p.vh->launch_prune_short; short_lc = p.vh->prune_short_lc; }
            else if (p.vh->launch_prune_short2 && K <= p.vh->prune_short_lc2) { short_fn = p.vh->launch_prune_short2; short_lc = p.vh->prune_short_lc2; }
            const bool short_lists = short_fn && !(force_lists && !strcmp(force_lists, "long"));
            MCE_HIP(short_lists ? short_fn(a, st) : p.vh->launch_prune(a, st));
            rc = prof_end();
            if (rc != MCE_OK) return rc;
            if (a.wg_us) {
                std::vector<float> us(n_wg);
                MCE_HIP(hipStreamSynchronize(st));
                MCE_HIP(hipMemcpy(us.data(), a.wg_us, n_wg * sizeof(float), hipMemcpyDeviceToHost));
                if (FILE* f = fopen(times_file, "wb")) {
                    const int hdr[4] = {(int)n_wg, hv_n, hv_S, 1};
                    fwrite(hdr, sizeof(int), 4, f);
                    fwrite(us.data(), sizeof(float), n_wg, f);
                    fclose(f);
                }
            }
            if (hv_n) {
                const unsigned fb = (unsigned)(((int64_t)hv_n * kPruneWaveQueries + mce::kRedThreads - 1) / mce::kRedThreads);
                hipLaunchKernelGGL(mce::prune_heavy_fold_kernel, dim3(fb), dim3(mce::kRedThreads), 0, st, pd, pi, p.nq_pad, p.KCAP, hv_d, hv_i, hv_n, hv_S,
                                   kPruneWaveQueries, po.border, a.qblk0, a.qblk_stride);
                MCE_HIP(hipGetLastError());
            }
            g_last_params = params;
            g_last_flops_main = g_last_flops_all = -1.0;        // (tiles multiplied: a device counter, mce_last_prune_stats)
            g_last_prune_geom[0] = p.nqblk; g_last_prune_geom[1] = (double)p.nchunk; g_last_prune_geom[2] = p.CT;
            snprintf(g_last_kernel, sizeof(g_last_kernel), "%s pruned grid=%d block=64 lds=%zu qt=%d ct=%d chunks=%lld heavy=%dx%d lists=%d", p.vh->name,
                     nblk_run + hv_n * (hv_S - 1), mce::f16_prune_lds_bytes(p.KST, d, short_lists ? short_lc : p.KCAP), p.QT, p.CT, (long long)p.nchunk, hv_n,
                     hv_S, short_lists ? short_lc : p.KCAP);
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
            // (a rank of the all-pairs-once partition prepasses its own blocks only and every rank's sweep gains from tighter bounds on
            //  everybody's rows: twice the sample from four ranks on -- C3, sample 32 k / 64 k / 128 k rows: step 13.15 / 13.00 / 13.60 ms at
            //  four ranks, 7.76 / 7.50 / 7.59 at eight, candidates shipped 5.9 M / 3.6 M / 2.4 M)
            const int seed_rows = tun.sym_seed_rows > 0 ? tun.sym_seed_rows : (p.KST == 1 ? 65536 : 32768) * (p.apo && p.nparts >= 4 ? 2 : 1);
            const int seed_share = tun.sym_seed_share;
            auto sym_seed_for = [&](int kk) {
                int cfg = mce::f16_seed_cfg(p.nchunk, p.CT, kk + a.self_exclude, seed_rows, seed_share, MCE_H_SEED_TG);
                // tiny sets (forced mode): smaller groups, so that half of the chunks still hold twice the K groups a bound needs --
                // without any bound every pair would go through the row side (20 k x 27: 10.8 ms instead of 0.8)
                for (int tg = MCE_H_SEED_TG / 2; cfg == 0 && tg >= 1; tg /= 2)
                    cfg = mce::f16_seed_cfg(p.nchunk, p.CT, kk + a.self_exclude, seed_rows, seed_share, tg);
                if (cfg) cfg |= ((tun.sym_seed_mode >= 0 ? tun.sym_seed_mode : kSymSeedMode[p.KST]) & 3) << 28;
                return cfg;
            };
            // One rank's share of a multi-GPU partition: the contiguous range of sorted blocks [qb_lo, qb_hi).  Their tiles
            // carry the row-side gate; everybody else's rows are swept column side only (sym_types.hpp, PanelGeom) -- no
            // exchange between the ranks, each ends with complete lists for its own rows.  Only they need a prepass bound.
            // (the all-pairs-once partition, p.apo: every nparts-th block instead -- the geometry's stride, below; the prepass then
            //  covers all blocks: a rank handles the rows of every lower block on the row side)
            const int qb_lo = p.nparts > 1 && !p.apo ? (int)((int64_t)p.nqblk * p.part / p.nparts) : 0;
            const int qb_hi = p.nparts > 1 && !p.apo ? (int)((int64_t)p.nqblk * (p.part + 1) / p.nparts) : p.nqblk;
            p.sym_qb_lo = qb_lo;
            p.sym_qb_hi = qb_hi;
            const bool panel_kernel = !tun.sym_kernel_f16 || p.nparts > 1 || p.twopass;
            mce::PanelGeom geom;
            geom.qb_lo = qb_lo; geom.qb_hi = qb_hi; geom.tpb = mce::kHWaves * mce::kHQT; geom.ct = p.CT; geom.tpp = a.sym.panel * p.CT; geom.sym_on = 1;
            geom.ntiles = (int)((nr + 31) / 32) + (int)(((nr + 31) / 32) & 1);
            // the all-pairs-once partition (capi_apo.hpp; sym_types.hpp): the single-GPU units of the blocks part, part + nparts, ...;
            // this call ends with the sweep -- repair and merge follow the exchange of the row-side candidates
            const bool apo = p.apo && p.nparts > 1 && !p.twopass && panel_kernel;
            // chains of units per block (PanelGeom.nsplit): the partition's own choice, or -- one GPU, one pass -- the plan's for sets with
            // MCE_SYM_CHAINS (capi_plan.hpp)
            const bool chains1 = !apo && p.nparts == 1 && panel_kernel && !p.twopass && p.sym_nsplit > 1;
            if (apo) { geom.blk_first = p.part; geom.blk_stride = p.nparts; }
            if (apo || chains1) {
                const int pn = apo ? p.apo_panel : p.sym_panel;
                if (pn > 0 && (apo || tun.sym_panel <= 0)) { a.sym.panel = pn; geom.tpp = a.sym.panel * p.CT; }      // (MCE_SYM_PANEL, if set, stands)
                geom.nsplit = std::max(1, apo ? p.apo_nsplit : p.sym_nsplit);
                if (geom.nsplit > 1) {
                    // one hand-over counter per chain: nqblk * nsplit words in the sort's second key array (n_pad words, free by now)
                    a.sym.done = reinterpret_cast<int*>(sw + p.sl.keys_b);
                    MCE_HIP(mce::zero_async(a.sym.done, (size_t)p.nqblk * geom.nsplit * sizeof(int), st));
                    if (!apo) {
                        hipLaunchKernelGGL(mce::sym_chain_init_kernel, dim3((unsigned)p.nqblk), dim3(512), 0, st, pd, pi, p.nq_pad, p.KCAP, geom.nsplit, 0, 1);
                        MCE_HIP(hipGetLastError());
                    }
                }
            }
            const bool unit_table = apo || geom.nsplit > 1;
            // 16 < K <= 32 (round 5): TWO symmetric passes over 16-entry lists, as the exhaustive sweep does it (knn_f16.hpp, LOWER) --
            // the first finds every row's 16 nearest (lists A), the second the next K - 16 beyond them (lists B: knn_panel.hpp,
            // LOWER); the merge takes the K best of A and B.  The second pass needs bounds on the K-th distance: a prepass
            // whose seed phase tracks K + 1 <= 33 group minima.
            const int npass = p.twopass ? 2 : 1;
            const size_t list_set = (size_t)p.KCAP * (size_t)p.nq_pad;
            int seed_used = 0;
            for (int pass = 0; pass < npass; ++pass) {
                const bool lower = pass == 1;
                const int Kp = !p.twopass ? K : (lower ? K - 16 : 16);          // neighbours this pass's lists are to hold
                double* const pdp = pd + (lower ? list_set : 0);
                int* const pip = pi + (lower ? list_set : 0);
                if (lower) {
                    MCE_HIP(mce::zero_async(a.sym.bucket_cnt, (size_t)3 * p.nqblk * sizeof(int), st));      // counts | flags | done
                    a.sym.slot_stride = p.KCAP;
                }
                // prepass: every row's bound on the distance of the LAST neighbour wanted (K-th: both passes of a two-pass search
                // publish bounds on what the row will finally hold) before any block runs (the seed phase as its own launch);
                // about 32 k rows (one k-step: 64 k), at most half of the chunks (tools/_tmp-style scans, fused call, share 8 -> 2:
                // 49 k x 27 1.51 -> 1.32 ms, 98 k 2.18 -> 2.03, 131 k 2.58 -> 2.47, from 197 k rows the same; 393 k x 15 7.04 -> 6.88)
                a.ksel = lower ? K : Kp;
                a.seed_cfg = sym_seed_for(a.ksel);
                a.part_d = pdp;
                a.part_i = pip;
                a.qblk0 = qb_lo;
                a.nqblk_run = qb_hi - qb_lo;
                if (apo) {      // only the rank's own blocks: the bounds of everybody's rows are all-reduced after this call (capi_apo.hpp)
                    a.qblk0 = p.part; a.qblk_stride = p.nparts; a.nqblk_run = mce::apo_rank_count(p.nqblk, p.part, p.nparts);
                }
                if (a.nqblk_run > 0 && phase != 2) MCE_HIP((lower ? p.vh->launch_sym_pre32 : p.vh->launch_sym_pre)(a, st));
                a.qblk0 = 0;
                a.qblk_stride = 1;
                a.nqblk_run = 0;
                if (phase == 1) return MCE_OK;
                if (pass == 0) seed_used = a.seed_cfg;
                a.seed_cfg = 0;
                a.ksel = Kp;
                a.lo_d = lower ? pd : nullptr;
                a.lo_i = lower ? pi : nullptr;
                int rc = pass == 0 ? prof_begin() : MCE_OK;             // (the bracket of mce_last_kernel_ms(): the dominant kernel, as for the other searches)
                if (rc != MCE_OK) return rc;
                if (panel_kernel) {
                    mce::PanelArgs pa;
                    pa.Yh = yh; pa.Xh = xh; pa.qinfo = qinfo; pa.params = params; pa.X = sX; pa.Y = sY; pa.rperm = a.rperm;
                    pa.part_d = pdp; pa.part_i = pip; pa.nq = nq; pa.nr = nr; pa.nq_pad = p.nq_pad; pa.self_offset = 0;
                    pa.D = d; pa.ksel = Kp; pa.self_exclude = a.self_exclude; pa.spin_limit = tun.spin_limit;
                    pa.sym = a.sym;
                    pa.debug = tun.panel_debug;
                    pa.geom = geom;
                    pa.lo_d = a.lo_d; pa.lo_i = a.lo_i;
                    if (unit_table) {
                        // strided blocks, several chains per block: the units as a table (in the sort's value array: n_pad words, free by now)
                        const int nun = mce::panel_unit_count(geom);
                        if ((size_t)nun * sizeof(mce::PanelUnit) > (size_t)p.nq_pad * sizeof(int)) return fail(MCE_ERR_INVALID, "symmetric sweep: %d units do not fit the table", nun);
                        mce::PanelUnit* tab = reinterpret_cast<mce::PanelUnit*>(sw + p.sl.vals_a);
                        if (nun > 0) {
                            hipLaunchKernelGGL(mce::panel_unit_table_kernel, dim3((unsigned)((nun + 255) / 256)), dim3(256), 0, st, geom, nun, tab);
                            MCE_HIP(hipGetLastError());
                        }
                        pa.units = tab;
                    }
                    MCE_HIP((lower ? p.vh->launch_panel_lower : p.vh->launch_panel)(pa, st));
                } else {
                    MCE_HIP(p.vh->launch_sym(a, st));
                }
                rc = pass == 0 ? prof_end() : MCE_OK;
                if (rc != MCE_OK) return rc;
                if (apo) break;             // (repair and merge: pairs_once_finish, after the exchange)
                MCE_HIP((lower ? p.vh->launch_sym_repair_lower : p.vh->launch_sym_repair)(a, st));      // blocks whose bucket overflowed (normally none: every workgroup exits at once)
                if (geom.nsplit > 1) {       // (a repaired block's set 0 is complete: its other sets are emptied)
                    hipLaunchKernelGGL(mce::sym_chain_clear_kernel, dim3((unsigned)p.nqblk), dim3(512), 0, st, a.sym.bucket_flag, pdp, pip, p.nq_pad, p.KCAP, geom.nsplit, 0, 1);
                    MCE_HIP(hipGetLastError());
                }
                MCE_HIP(launch_sym_merge(p.KCAP, pdp, pip, p.nq_pad, a.sym, qb_lo, qb_hi, st));
            }
            p.sym_active = true;
            p.L = geom.nsplit > 1 ? geom.nsplit : npass;
            int sym_units = mce::sym_unit_count(p.nqblk, mce::kHWaves * mce::kHQT, a.sym.panel * p.CT, (int)((nr + 31) / 32) + (int)(((nr + 31) / 32) & 1));
            if (panel_kernel) {
                const mce::PanelGeom& g = geom;
                sym_units = mce::panel_unit_count(g);
                // executed MFMA flops: every unit's tiles x 16 query tiles x (32 x 32 x 16 KST) multiply-adds -- a figure for
                // mce_last_search_stats(), counted only while profiling is on (the loop is O(units x panels): seconds of host
                // time per search near the row limit)
                double tiles = 0.0;
                for (int u = 0; g_prof_on && u < sym_units; ++u) {
                    int pp, aa, lo, hi;
                    mce::panel_unit_decode(u, g, pp, aa);
                    mce::panel_unit_tiles(pp, aa, g, lo, hi);
                    tiles += hi - lo;
                }
                g_last_flops_main = tiles * 16.0 * 1024.0 * 32.0 * p.KST;        // (a two-pass search: the first pass's sweep, which the event bracket times)
            } else {
                const double nb = p.nqblk, tpb = mce::kHWaves * mce::kHQT, T = (double)((nr + 31) / 32);
                double tiles = 0.0;
                for (int b = 0; b < p.nqblk; ++b) tiles += std::min(tpb * (b + 1), T);
                (void)nb;
                g_last_flops_main = tiles * 16.0 * 1024.0 * 32.0 * p.KST;
            }
            g_last_flops_all = g_last_flops_main + (double)(seed_used & 0xffff) * p.CT * (double)(qb_hi - qb_lo) * 16.0 * 1024.0 * 32.0 * p.KST;
            snprintf(g_last_kernel, sizeof(g_last_kernel), "%s symmetric%s%s%s grid=%d block=%d lds=%zu qt=%d ct=%d panel=%d seed=%dx%d/%d bucket=%d", p.vh->name, panel_kernel ? " panel-kernel" : "",
                     p.twopass ? " two passes" : "", geom.blk_stride > 1 ? " pairs-once" : (geom.nsplit > 1 ? " chains" : ""), sym_units,
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
            if (p.wide_ok) MCE_HIP(p.vh->launch_wide(a, st));       // four query tiles per wave: a workgroup = two query blocks
            else MCE_HIP(p.vh->launch(a, st));
        }
        rc = prof_end();
        if (rc != MCE_OK) return rc;
        const int seed_first = p.twopass ? seed_cfg(16) : a.seed_cfg;
        g_last_flops_main = (double)p.nqblk * ((double)p.nchunk + (double)(seed_first & 0xffff) * p.rsplit) * p.CT * 16.0 * 1024.0 * 32.0 * p.KST +
                            (p.twopass ? (double)p.nqblk * (double)p.nchunk * p.CT * 16.0 * 1024.0 * 32.0 * p.KST : 0.0);
        g_last_flops_all = g_last_flops_main;
        char seed_txt[48] = "";
        if (seed_first) snprintf(seed_txt, sizeof(seed_txt), " seed=%dx%d", seed_first & 0xffff, seed_first >> 16);   // chunks x tiles per group
        const bool wide = p.wide_ok && !p.twopass;
        snprintf(g_last_kernel, sizeof(g_last_kernel), "%s grid=%d block=%d lds=%zu qt=%d ct=%d rsplit=%d%s%s%s", p.vh->name,
                 (wide ? p.nqblk / 2 : p.nqblk) * p.rsplit, mce::kHThreads, wide ? p.vh->lds_bytes_wide : p.vh->lds_bytes, wide ? 2 * p.QT : p.QT, p.CT, p.rsplit,
                 p.twopass ? " two passes" : "", seed_txt, wide ? " wide" : "");
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
    a.ksel = p.ksel > K ? p.ksel : K;          // (K + kRefineMargin: the merge picks the K on exact distances)
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
    // (a part of a pruned walk takes every nparts-th WAVE of the dispatch order: its list columns come in runs of 64)
    const bool wave_parts = p.prune && p.nparts > 1 && !p.sym_active;
    const int qpb = wave_parts ? kPruneWaveQueries : (p.filter() ? mce::f16_qpb(p.KCAP) : 1);
    const int nunits = wave_parts ? p.nqblk * mce::kHWaves : p.nqblk;
    int64_t ncol = nq;
    int64_t col0 = 0, col1 = INT64_MAX;
    const int* border = p.prune ? reinterpret_cast<const int*>(ws + p.off_prune + p.pl.border) : nullptr;
    if (p.sym_active && p.nparts > 1 && p.apo) {
        // the all-pairs-once partition: every nparts-th block of list columns, enumerated through the identity table the
        // finish call left in the `done` array (the kernel's border path)
        ncol = (int64_t)((nunits - p.part + p.nparts - 1) / p.nparts) * qpb;
        border = reinterpret_cast<const int*>(ws + p.off_sym + p.sl.done);
    } else if (p.sym_active && p.nparts > 1) {         // one rank's blocks of a symmetric partition: a contiguous range of list columns
        col0 = (int64_t)p.sym_qb_lo * qpb;
        col1 = std::min<int64_t>((int64_t)p.sym_qb_hi * qpb, nq);
        ncol = std::max<int64_t>(col1 - col0, 0);
    } else if (p.nparts > 1) ncol = (int64_t)((nunits - p.part + p.nparts - 1) / p.nparts) * qpb;
    const unsigned blocks = (unsigned)std::max<int64_t>((ncol + mce::kRedThreads - 1) / mce::kRedThreads, 1);
    const double* pd = reinterpret_cast<const double*>(ws + p.off_pd);
    const int* pi = reinterpret_cast<const int*>(ws + p.off_pi);
    const bool refine = !p.filter() && !p.generic;   // fp64 sweep keys are GEMM-form: refine; the others are exact
    const double lnc = fuse ? ln_unit_ball(d) : 0.0;
    // pruned search: list column q is the q-th query in k-d order; its caller row is qperm[q]
    const int* qperm = nullptr;
    if (p.prune) qperm = reinterpret_cast<const int*>(ws + p.off_prune + (same_set ? p.pl.perm_r : p.pl.perm_q));
    if (p.sym_active) qperm = reinterpret_cast<const int*>(ws + p.off_sym + p.sl.perm);     // list column = sorted position
#define MCE_MERGE(W, F, R)                                                                                          \
    hipLaunchKernelGGL((mce::merge_lists_kernel<W, F, R>), dim3(blocks), dim3(mce::kRedThreads), 0, st, pd, pi, p.L,  \
                       p.KCAP, nq, p.nq_pad, dX, dY, (int)d, K, (refine && p.ksel > K) ? p.ksel : K, self_mode, self_offset, d_dist, d_idx, K, k0, kmax, \
                       d_w, d_fs, lnc, partial, qperm, p.part, p.nparts, qpb, border, nunits, col0, col1)
    if (write_dist && !fuse) { if (refine) MCE_MERGE(true, false, true); else MCE_MERGE(true, false, false); }
    else if (write_dist && fuse) { if (refine) MCE_MERGE(true, true, true); else MCE_MERGE(true, true, false); }
    else { if (refine) MCE_MERGE(false, true, true); else MCE_MERGE(false, true, false); }
#undef MCE_MERGE
    MCE_HIP(hipGetLastError());
    return MCE_OK;
}

}  // namespace
