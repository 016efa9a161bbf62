// capi_feed.hpp -- part of capi.hip: the host-pointer fused call on one device, the evidence feed (covariance -> eigen-system ->
// whitening -> search -> reduction, MCEvidence.py:842-947 + :1093-1117) as a pipelined batch, and their entry points.
#pragma once
namespace {

// cyclic Jacobi eigen-solver for a symmetric d x d matrix (row-major A, destroyed); eigenvalues in
// lam[d], eigenvectors in the COLUMNS of V (row-major [d][d]).  d <= 1024; converges to ~1e-15.
void jacobi_eig(std::vector<double>& A, int d, std::vector<double>& lam, std::vector<double>& V)
{
    V.assign((size_t)d * d, 0.0);
    for (int i = 0; i < d; ++i) V[(size_t)i * d + i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < d; ++i) {
            diag += A[(size_t)i * d + i] * A[(size_t)i * d + i];
            for (int j = i + 1; j < d; ++j) off += A[(size_t)i * d + j] * A[(size_t)i * d + j];
        }
        if (off <= 1e-32 * diag || off == 0.0) break;
        for (int p = 0; p < d - 1; ++p)
            for (int q = p + 1; q < d; ++q) {
                const double apq = A[(size_t)p * d + q];
                if (apq == 0.0) continue;
                const double app = A[(size_t)p * d + p], aqq = A[(size_t)q * d + q];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < d; ++k) {          // A <- A J   (columns p, q)
                    const double akp = A[(size_t)k * d + p], akq = A[(size_t)k * d + q];
                    A[(size_t)k * d + p] = c * akp - s * akq;
                    A[(size_t)k * d + q] = s * akp + c * akq;
                }
                for (int k = 0; k < d; ++k) {          // A <- J^T A (rows p, q)
                    const double apk = A[(size_t)p * d + k], aqk = A[(size_t)q * d + k];
                    A[(size_t)p * d + k] = c * apk - s * aqk;
                    A[(size_t)q * d + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < d; ++k) {          // V <- V J
                    const double vkp = V[(size_t)k * d + p], vkq = V[(size_t)k * d + q];
                    V[(size_t)k * d + p] = c * vkp - s * vkq;
                    V[(size_t)k * d + q] = s * vkp + c * vkq;
                }
            }
    }
    // canonical form: eigenvalues descending, each eigenvector's largest component positive.  Two
    // sets whitened with their OWN systems (covtype 'single' cross evidence) are then rotated
    // consistently whenever their covariances are close, whatever the sweep order did.
    std::vector<int> order(d);
    for (int i = 0; i < d; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return A[(size_t)a * d + a] > A[(size_t)b * d + b]; });
    lam.resize(d);
    std::vector<double> Vs((size_t)d * d);
    for (int c = 0; c < d; ++c) {
        const int src = order[c];
        lam[c] = A[(size_t)src * d + src];
        int big = 0;
        for (int k = 1; k < d; ++k)
            if (std::fabs(V[(size_t)k * d + src]) > std::fabs(V[(size_t)big * d + src])) big = k;
        const double sgn = V[(size_t)big * d + src] < 0.0 ? -1.0 : 1.0;
        for (int k = 0; k < d; ++k) Vs[(size_t)k * d + c] = sgn * V[(size_t)k * d + src];
    }
    V.swap(Vs);
}

// covariance (two-pass, unweighted, n-1) of the device matrix S[n, d] -> device cov[d*d]; enqueue only
int launch_covariance(const double* dS, int64_t n, int d, double* scratch_partial, double* d_mean3, double* d_cov, hipStream_t st)
{
    if (d > 63) {           // (the wide feeders, 64 <= d <= kFeedMaxDim: the means take 128 of the 192 doubles at d_mean3)
        hipLaunchKernelGGL(mce::col_mean_wide_partial_kernel, dim3(mce::kMeanBlocks), dim3(256), 0, st, dS, n, d, scratch_partial);
        MCE_HIP(hipGetLastError());
        hipLaunchKernelGGL(mce::col_mean_wide_final_kernel, dim3(1), dim3(128), 0, st, scratch_partial, n, d, d_mean3);
        MCE_HIP(hipGetLastError());
    } else {
        hipLaunchKernelGGL(mce::col_stats_partial_kernel, dim3(mce::kMeanBlocks), dim3(mce::kMeanThreads), 0, st, dS, n, d, scratch_partial);
        MCE_HIP(hipGetLastError());
        hipLaunchKernelGGL(mce::col_stats_final_kernel, dim3(1), dim3(64), 0, st, scratch_partial, n, d, d_mean3, (double*)nullptr);
        MCE_HIP(hipGetLastError());
    }
    for (int p0 = 0; p0 < d * (d + 1) / 2; p0 += mce::kCovPairsPerLaunch) {
        hipLaunchKernelGGL(mce::cov_partial_kernel, dim3(mce::kCovBlocks), dim3(mce::kCovThreads), (size_t)mce::kCovTileRows * d * sizeof(double), st,
                           dS, n, d, d_mean3, scratch_partial, p0);
        MCE_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(mce::cov_final_kernel, dim3(1), dim3(mce::kCovThreads), 0, st, scratch_partial, n, d, d_cov);
    MCE_HIP(hipGetLastError());
    return MCE_OK;
}

// one device's share of the fused path: queries [q_lo, q_hi)
int fused_on_device(int device, const double* X, int64_t q_lo, int64_t q_hi, const double* Y, int64_t nr, int32_t d,
                    int32_t kmax, int32_t k0, int64_t self_offset, const double* w, const double* fs,
                    double* dotp_part, double* dist_out)
{
    const int64_t nq = q_hi - q_lo;
    const int K = kmax - k0;
    int rc = select_device(device);
    if (rc != MCE_OK) return rc;
    const double* Xs = X + q_lo * (int64_t)d;
    SameSetHint hint(Xs == Y && nq == nr && self_offset + q_lo == 0);
    Plan p;
    rc = make_plan(nq, nr, d, K, k0 == 1 ? MCE_SELF_EXCLUDE : MCE_SELF_NONE, p);
    if (rc != MCE_OK) return rc;
    const size_t wsb = p.total + dotp_ws_bytes(nq, kmax);
    const bool inside = Xs >= Y && Xs + (size_t)nq * d <= Y + (size_t)nr * d && (Xs - Y) % d == 0;     // as in mce_knn_f64
    DevBuf dX, dY, dW, dF, dO, dD, ws;
    if (!inside) MCE_HIP(dX.alloc((size_t)nq * d * sizeof(double)));
    MCE_HIP(dY.alloc((size_t)nr * d * sizeof(double)));
    MCE_HIP(dW.alloc((size_t)nq * sizeof(double)));
    MCE_HIP(dF.alloc((size_t)nq * sizeof(double)));
    MCE_HIP(dO.alloc((size_t)kmax * sizeof(double)));
    const int nverify = (d <= mce::kVerifyMaxDim && K <= mce::kVerifyMaxK) ? eff_verify(p.filter(), nq) : 0;     // mce_options.verify (default: on behind the fp16 filter)
    if (dist_out || nverify) MCE_HIP(dD.alloc((size_t)nq * K * sizeof(double)));
    MCE_HIP(ws.alloc(wsb));
    if (!inside) MCE_HIP(hipMemcpy(dX.p, Xs, (size_t)nq * d * sizeof(double), hipMemcpyHostToDevice));
    MCE_HIP(hipMemcpy(dY.p, Y, (size_t)nr * d * sizeof(double), hipMemcpyHostToDevice));
    const double* dXp = inside ? dY.as<double>() + (Xs - Y) : dX.as<double>();
    MCE_HIP(hipMemcpy(dW.p, w + q_lo, (size_t)nq * sizeof(double), hipMemcpyHostToDevice));
    MCE_HIP(hipMemcpy(dF.p, fs + q_lo, (size_t)nq * sizeof(double), hipMemcpyHostToDevice));
    rc = mce_knn_dotp_f64_dev(dXp, nq, dY.as<double>(), nr, d, kmax, k0, self_offset + q_lo,
                              dW.as<double>(), dF.as<double>(), dO.as<double>(), (dist_out || nverify) ? dD.as<double>() : nullptr,
                              ws.p, wsb, nullptr);
    if (rc != MCE_OK) return rc;
    MCE_HIP(hipDeviceSynchronize());
    MCE_HIP(hipMemcpy(dotp_part, dO.p, (size_t)kmax * sizeof(double), hipMemcpyDeviceToHost));
    if (dist_out) MCE_HIP(hipMemcpy(dist_out + q_lo * (int64_t)K, dD.p, (size_t)nq * K * sizeof(double), hipMemcpyDeviceToHost));
    g_last_verify_rows.store(0);
    if (nverify)
        return verify_after_search(dXp, nq, dY.as<double>(), nr, d, K, k0 == 1 ? MCE_SELF_EXCLUDE : MCE_SELF_NONE, self_offset + q_lo, dD.as<double>(), K, nverify);
    return MCE_OK;
}

constexpr int kFeedMaxDim = 127;      // device feeders: d <= 63 on the narrow kernels, 64..127 on the wide ones (round 5; the search: knn_mfma.hpp's wide form)
static_assert(kFeedMaxDim <= mce::kWideMaxDim && mce::whiten_wide_lds_bytes(kFeedMaxDim) <= 160 * 1024, "wide feeders");
// ---- evidence feed: covariance -> eigen-system -> whitening -> search -> reduction ---------------
// One problem is four stages; only B runs on the host:
//   A  upload the raw rows, enqueue the covariance kernels, copy cov back (async, pinned)
//   B  d x d Jacobi eigen-solve, whitening scales, Jacobian
//   C  upload eVec/scale, whiten in place, fused search + reduction, copy dotp back (async, pinned)
//   D  hand the results to the caller
// A batch is pipelined two deep in groups of kFeedGroup problems: while the device runs stage C of
// group g, the host performs the (blocking, pageable) uploads of group g+1 and that group's
// covariance kernels run beside the searches on a second stream set.  The searches themselves fill
// the device (make_plan splits the reference set of a small problem over all CUs), so the gain is
// hiding the PCIe upload, the host eigen-solves and the per-problem synchronisations.  Problems are
// processed in waves bounded by kWaveBytes of device memory.
constexpr size_t kWaveBytes = (size_t)8 << 30;
constexpr int kWaveMaxJobs = 1024;
constexpr int kFeedStreams = 4;   // per set (upload+covariance | whiten+search)
constexpr int kFeedGroup = 8;     // problems per pipeline step

struct FeedJob {
    mce_feed_problem* q = nullptr;
    int64_t index = 0;
    Plan plan;
    int k0 = 1, K = 0, rc = MCE_OK;
    int64_t nr = 0, ntot = 0;
    size_t wsb = 0, dev_bytes = 0, host_bytes = 0;
    size_t o_S = 0, o_W = 0, o_F = 0, o_O = 0, o_small = 0, o_part = 0, o_ws = 0, o_vd = 0, o_vw = 0, o_vr = 0;
    int nverify = 0;          // mce_options.verify: rows re-checked after the search (a whole problem only, not a rank's share)
    char* dbase = nullptr;    // this job's slice of the wave's device arena
    double* hbase = nullptr;  // this job's slice of the pinned host arena: cov[2] | evec[2] | scale[2] | dotp
    double jac = 0.0;
    int part = 0, nparts = 1;         // one rank's share of the problem (mce_evidence_feed_part_f64); 1: all of it
    double* out_X = nullptr;          // mce_evidence_feed_whiten_f64: no search -- the whitened rows, the weights and the likelihood terms are left in
    double* out_w = nullptr;          // these DEVICE buffers of the caller's (auto evidence; the all-pairs-once partition searches them in several calls
    double* out_f = nullptr;          // with collectives in between: parallel.py)
    bool src_device = false;          // S1 / S2 / w / fs are DEVICE pointers (mce_evidence_feed_part_dev_f64: the node uploaded the chain once and
                                      // gathered it over xGMI): stage A copies device to device on the job's stream
    bool want_sum = false;            // fingerprint of the uploaded rows / weights / likelihoods (device-side)
    unsigned long long checksum = 0;
    int64_t q_lo = 0, q_hi = 0;       // cross evidence of a part: its rows of s1
    std::vector<double> lam;  // eigenvalues of the system that defines J (s1's in 'single' mode)
    std::string err;
    hipEvent_t upload_ev = nullptr;   // orders the job's stream behind its blocking uploads
    ~FeedJob() { if (upload_ev) (void)hipEventDestroy(upload_ev); }
    FeedJob() = default;
    FeedJob(const FeedJob&) = delete;
    FeedJob& operator=(const FeedJob&) = delete;

    int d() const { return q->d; }
    double* dS1() const { return reinterpret_cast<double*>(dbase + o_S); }
    double* dS2() const { return dS1() + (size_t)q->n1 * q->d; }
    double* dW() const { return reinterpret_cast<double*>(dbase + o_W); }
    double* dF() const { return reinterpret_cast<double*>(dbase + o_F); }
    double* dO() const { return reinterpret_cast<double*>(dbase + o_O); }
    double* d_mean3() const { return reinterpret_cast<double*>(dbase + o_small); }
    double* d_cov() const { return d_mean3() + 3 * 64; }
    double* d_evec() const { return d_cov() + (size_t)q->d * q->d; }
    double* d_scale() const { return d_evec() + (size_t)q->d * q->d; }
    unsigned long long* d_sum() const { return reinterpret_cast<unsigned long long*>(d_scale() + q->d); }
    double* d_part() const { return reinterpret_cast<double*>(dbase + o_part); }
    char* ws() const { return dbase + o_ws; }
    double* h_cov(int i) const { return hbase + (size_t)i * q->d * q->d; }
    double* h_evec(int i) const { return hbase + (size_t)(2 + i) * q->d * q->d; }
    double* h_scale(int i) const { return hbase + (size_t)4 * q->d * q->d + (size_t)i * q->d; }
    double* h_dotp() const { return hbase + (size_t)4 * q->d * q->d + (size_t)2 * q->d; }
    unsigned long long* h_sum() const { return reinterpret_cast<unsigned long long*>(h_dotp() + q->kmax); }
    int* h_verify() const { return reinterpret_cast<int*>(h_sum() + 1); }        // {rows checked, rows failed}
    double* d_vdist() const { return reinterpret_cast<double*>(dbase + o_vd); }
    int32_t* d_vres() const { return reinterpret_cast<int32_t*>(dbase + o_vr); }
    bool two_systems() const { return q->cov_mode == 1 && q->S2 != nullptr; }
    void set_error(int code) { rc = code; err = g_err; }
};

// argument checks + sizes; no device work
int feed_plan(FeedJob& j)
{
    const mce_feed_problem& q = *j.q;
    if (!q.S1 || !q.w || !q.fs || !q.dotp) return fail(MCE_ERR_INVALID, "null pointer argument");
    if (q.n1 < 2 || q.d < 1 || q.ld1 < q.d || (q.S2 && (q.n2 < 1 || q.ld2 < q.d)) || (q.cov_mode != 0 && q.cov_mode != 1))
        return fail(MCE_ERR_INVALID, "invalid sizes n1=%lld ld1=%lld n2=%lld ld2=%lld d=%d cov_mode=%d", (long long)q.n1, (long long)q.ld1,
                    (long long)q.n2, (long long)q.ld2, q.d, q.cov_mode);
    if (q.d > kFeedMaxDim) return fail(MCE_ERR_DIM_RANGE, "device feeders support d <= %d (got %d)", kFeedMaxDim, q.d);
    j.k0 = q.S2 ? 0 : 1;
    j.K = q.kmax - j.k0;
    if (q.kmax <= j.k0) return fail(MCE_ERR_INVALID, "kmax=%d must exceed k0=%d", q.kmax, j.k0);
    j.nr = q.S2 ? q.n2 : q.n1;
    j.ntot = q.n1 + (q.S2 ? q.n2 : 0);
    SameSetHint hint(q.S2 == nullptr);          // auto evidence: one set; cross evidence: never the symmetric sweep
    int rc = make_plan(q.n1, j.nr, q.d, j.K, j.k0 == 1 ? MCE_SELF_EXCLUDE : MCE_SELF_NONE, j.plan);
    if (rc != MCE_OK) return rc;
    j.wsb = j.out_X ? 0 : j.plan.total + dotp_ws_bytes(q.n1, q.kmax);
    j.q_lo = 0;
    j.q_hi = q.n1;
    if (j.nparts > 1 && q.S2) {
        // a part of a cross-evidence problem: contiguous rows of s1 against all of s2
        j.q_lo = q.n1 * j.part / j.nparts;
        j.q_hi = q.n1 * (int64_t)(j.part + 1) / j.nparts;
        if (j.q_hi > j.q_lo) {
            Plan shard;
            rc = make_plan(j.q_hi - j.q_lo, j.nr, q.d, j.K, MCE_SELF_NONE, shard);
            if (rc != MCE_OK) return rc;
            j.wsb = std::max(j.wsb, shard.total + dotp_ws_bytes(j.q_hi - j.q_lo, q.kmax));
        }
    }
    const int d = q.d, npair = d * (d + 1) / 2;
    size_t off = 0;
    j.o_S = off;     off = align_up(off + (size_t)j.ntot * d * sizeof(double), 256);
    j.o_W = off;     off = align_up(off + (size_t)q.n1 * sizeof(double), 256);
    j.o_F = off;     off = align_up(off + (size_t)q.n1 * sizeof(double), 256);
    j.o_O = off;     off = align_up(off + (size_t)q.kmax * sizeof(double), 256);
    j.o_small = off; off = align_up(off + (size_t)(3 * 64 + 2 * d * d + d + 1) * sizeof(double), 256);     // mean3 | cov | evec | scale | checksum
    j.o_part = off;  off = align_up(off + (size_t)std::max<int64_t>((int64_t)mce::kCovBlocks * npair, (int64_t)mce::kMeanBlocks * mce::kStatStride) * sizeof(double), 256);
    j.o_ws = off;    off = align_up(off + j.wsb, 256);
    j.nverify = (j.nparts == 1 && !j.out_X && d <= mce::kVerifyMaxDim && j.K <= mce::kVerifyMaxK) ? (int)std::min<int64_t>(eff_verify(j.plan.filter(), q.n1), q.n1) : 0;
    if (j.nverify > 0) {
        j.o_vd = off; off = align_up(off + (size_t)q.n1 * j.K * sizeof(double), 256);
        j.o_vw = off; off = align_up(off + mce_verify_workspace_bytes(j.nverify, j.K), 256);
        j.o_vr = off; off = align_up(off + 2 * sizeof(int), 256);
    }
    j.dev_bytes = off;
    j.host_bytes = align_up((size_t)(4 * d * d + 2 * d + q.kmax + 2) * sizeof(double), 64);
    return MCE_OK;
}

int feed_stage_a(FeedJob& j, hipStream_t st)
{
    const mce_feed_problem& q = *j.q;
    const int d = q.d;
    const size_t row = (size_t)d * sizeof(double);
    // Uploads: blocking copies from the caller's pageable arrays (measured faster than hipMemcpyAsync on the job's
    // non-blocking stream: 300 Planck-sized chains 0.131 vs 0.153 s), followed by an EXPLICIT dependency -- an event
    // recorded on the stream the copies ran on, waited for by the job's stream -- so the covariance kernels behind them
    // are ordered after the uploads by the API's rules, not by how this runtime happens to implement a pageable copy
    // (a blocking hipMemcpy from pageable memory only promises that the SOURCE has been consumed on return, and
    // hipStreamNonBlocking streams do not synchronise with the legacy default stream).  MCE_FEED_UPLOAD=async: the
    // copies themselves on the job's stream.
    static const bool async_upload = [] { const char* e = getenv("MCE_FEED_UPLOAD"); return e && !strcmp(e, "async"); }();
    if (j.src_device) {
        // the rows are on this device already (the caller has synchronised the stream that produced them)
        MCE_HIP(hipMemcpy2DAsync(j.dS1(), row, q.S1, (size_t)q.ld1 * sizeof(double), row, (size_t)q.n1, hipMemcpyDeviceToDevice, st));
        if (q.S2) MCE_HIP(hipMemcpy2DAsync(j.dS2(), row, q.S2, (size_t)q.ld2 * sizeof(double), row, (size_t)q.n2, hipMemcpyDeviceToDevice, st));
        MCE_HIP(hipMemcpyAsync(j.dW(), q.w, (size_t)q.n1 * sizeof(double), hipMemcpyDeviceToDevice, st));
        MCE_HIP(hipMemcpyAsync(j.dF(), q.fs, (size_t)q.n1 * sizeof(double), hipMemcpyDeviceToDevice, st));
    } else if (async_upload || st == nullptr) {
        MCE_HIP(hipMemcpy2DAsync(j.dS1(), row, q.S1, (size_t)q.ld1 * sizeof(double), row, (size_t)q.n1, hipMemcpyHostToDevice, st));
        if (q.S2) MCE_HIP(hipMemcpy2DAsync(j.dS2(), row, q.S2, (size_t)q.ld2 * sizeof(double), row, (size_t)q.n2, hipMemcpyHostToDevice, st));
        MCE_HIP(hipMemcpyAsync(j.dW(), q.w, (size_t)q.n1 * sizeof(double), hipMemcpyHostToDevice, st));
        MCE_HIP(hipMemcpyAsync(j.dF(), q.fs, (size_t)q.n1 * sizeof(double), hipMemcpyHostToDevice, st));
    } else {
        MCE_HIP(hipMemcpy2D(j.dS1(), row, q.S1, (size_t)q.ld1 * sizeof(double), row, (size_t)q.n1, hipMemcpyHostToDevice));
        if (q.S2) MCE_HIP(hipMemcpy2D(j.dS2(), row, q.S2, (size_t)q.ld2 * sizeof(double), row, (size_t)q.n2, hipMemcpyHostToDevice));
        MCE_HIP(hipMemcpy(j.dW(), q.w, (size_t)q.n1 * sizeof(double), hipMemcpyHostToDevice));
        MCE_HIP(hipMemcpy(j.dF(), q.fs, (size_t)q.n1 * sizeof(double), hipMemcpyHostToDevice));
        if (!j.upload_ev) MCE_HIP(hipEventCreateWithFlags(&j.upload_ev, hipEventDisableTiming));
        MCE_HIP(hipEventRecord(j.upload_ev, nullptr));
        MCE_HIP(hipStreamWaitEvent(st, j.upload_ev, 0));
    }
    if (j.want_sum) {
        // fingerprint of what this rank was handed (RAW rows, weights, fs), before anything is whitened in place
        MCE_HIP(mce::zero_async(j.d_sum(), sizeof(unsigned long long), st));
        const int64_t nw[3] = {j.ntot * (int64_t)d, q.n1, q.n1};
        const double* src[3] = {j.dS1(), j.dW(), j.dF()};
        for (int b = 0; b < 3; ++b) {
            const unsigned blocks = (unsigned)std::min<int64_t>((nw[b] + mce::kSumThreads - 1) / mce::kSumThreads, 2048);
            hipLaunchKernelGGL(mce::checksum_kernel, dim3(blocks), dim3(mce::kSumThreads), 0, st, reinterpret_cast<const unsigned long long*>(src[b]), nw[b],
                               (unsigned long long)(b + 1) << 56 ^ (unsigned long long)q.n1 << 8 ^ (unsigned long long)d, j.d_sum());
            MCE_HIP(hipGetLastError());
        }
        MCE_HIP(hipMemcpyAsync(j.h_sum(), j.d_sum(), sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    }
    // "all": one eigen-system from s1 U s2; "single": s1's own, and s2's own for s2 (J stays s1's)
    int rc = launch_covariance(j.dS1(), q.cov_mode == 0 ? j.ntot : q.n1, d, j.d_part(), j.d_mean3(), j.d_cov(), st);
    if (rc != MCE_OK) return rc;
    MCE_HIP(hipMemcpyAsync(j.h_cov(0), j.d_cov(), (size_t)d * d * sizeof(double), hipMemcpyDeviceToHost, st));
    if (j.two_systems()) {
        rc = launch_covariance(j.dS2(), q.n2, d, j.d_part(), j.d_mean3(), j.d_cov(), st);
        if (rc != MCE_OK) return rc;
        MCE_HIP(hipMemcpyAsync(j.h_cov(1), j.d_cov(), (size_t)d * d * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    return MCE_OK;
}

int feed_stage_b(FeedJob& j)
{
    const int d = j.d();
    const int nsys = j.two_systems() ? 2 : 1;
    for (int sidx = 0; sidx < nsys; ++sidx) {
        std::vector<double> cov(j.h_cov(sidx), j.h_cov(sidx) + (size_t)d * d), lam, V;
        jacobi_eig(cov, d, lam, V);
        for (int i = 0; i < d; ++i) {
            if (lam[i] != lam[i] || std::isinf(lam[i])) return fail(MCE_ERR_INVALID, "samples contain NaN or infinity (non-finite covariance)");
            if (!(lam[i] > 0.0)) return fail(MCE_ERR_INVALID, "math domain error: covariance eigenvalue %d is %g (use fewer parameters, ndim)", i, lam[i]);
        }
        std::copy(V.begin(), V.end(), j.h_evec(sidx));
        for (int i = 0; i < d; ++i) j.h_scale(sidx)[i] = 1.0 / std::sqrt(lam[i]);
        if (sidx == 0) {
            double logdet = 0.0;
            for (int i = 0; i < d; ++i) logdet += std::log(lam[i]);
            j.jac = std::exp(0.5 * logdet);
            j.lam = lam;
        }
    }
    return MCE_OK;
}

int feed_whiten(FeedJob& j, int sidx, double* rows, int64_t n, hipStream_t st)
{
    const int d = j.d();
    MCE_HIP(hipMemcpyAsync(j.d_evec(), j.h_evec(sidx), (size_t)d * d * sizeof(double), hipMemcpyHostToDevice, st));
    MCE_HIP(hipMemcpyAsync(j.d_scale(), j.h_scale(sidx), (size_t)d * sizeof(double), hipMemcpyHostToDevice, st));
    if (d > 63)
        hipLaunchKernelGGL(mce::whiten_wide_kernel, dim3((unsigned)((n + mce::kWhitenRows - 1) / mce::kWhitenRows)), dim3(mce::kWhitenRows),
                           mce::whiten_wide_lds_bytes(d), st, rows, n, d, j.d_evec(), j.d_scale(), rows);
    else
        hipLaunchKernelGGL(mce::whiten_kernel, dim3((unsigned)((n + mce::kWhitenRows - 1) / mce::kWhitenRows)), dim3(mce::kWhitenRows),
                           mce::whiten_lds_bytes(d), st, rows, n, d, j.d_evec(), j.d_scale(), rows);
    MCE_HIP(hipGetLastError());
    return MCE_OK;
}

int feed_stage_c(FeedJob& j, hipStream_t st)
{
    const mce_feed_problem& q = *j.q;
    {
        static std::atomic<bool> attr_set[kMaxDevices];
        int dev = 0;
        MCE_HIP(hipGetDevice(&dev));
        if (dev < kMaxDevices && !attr_set[dev].load()) {
            MCE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mce::whiten_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mce::whiten_lds_bytes(63)));
            MCE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mce::whiten_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)mce::whiten_wide_lds_bytes(kFeedMaxDim)));
            attr_set[dev].store(true);
        }
    }
    int rc;
    if (j.two_systems()) {
        rc = feed_whiten(j, 0, j.dS1(), q.n1, st);
        if (rc != MCE_OK) return rc;
        rc = feed_whiten(j, 1, j.dS2(), q.n2, st);
    } else {
        rc = feed_whiten(j, 0, j.dS1(), q.cov_mode == 0 ? j.ntot : q.n1, st);
    }
    if (rc != MCE_OK) return rc;
    SameSetHint hint(q.S2 == nullptr);          // as in feed_plan: the workspace was sized with it
    if (j.out_X) {                              // whitening only: hand the rows over, no search (the sums stay zero)
        MCE_HIP(hipMemcpyAsync(j.out_X, j.dS1(), (size_t)q.n1 * q.d * sizeof(double), hipMemcpyDeviceToDevice, st));
        MCE_HIP(hipMemcpyAsync(j.out_w, j.dW(), (size_t)q.n1 * sizeof(double), hipMemcpyDeviceToDevice, st));
        MCE_HIP(hipMemcpyAsync(j.out_f, j.dF(), (size_t)q.n1 * sizeof(double), hipMemcpyDeviceToDevice, st));
        rc = (mce::zero_async(j.dO(), (size_t)q.kmax * sizeof(double), st) == hipSuccess) ? MCE_OK : fail(MCE_ERR_HIP, "clearing the sums failed");
    } else if (j.nparts > 1 && !q.S2)           // one rank's share of an auto-evidence search: the library's partition
        rc = mce_knn_dotp_part_f64_dev(j.dS1(), q.n1, q.d, q.kmax, j.part, j.nparts, j.dW(), j.dF(), j.dO(), j.ws(), j.wsb, st);
    else if (j.nparts > 1 && j.q_hi <= j.q_lo)
        rc = (mce::zero_async(j.dO(), (size_t)q.kmax * sizeof(double), st) == hipSuccess) ? MCE_OK : fail(MCE_ERR_HIP, "clearing the sums failed");
    else if (j.nparts > 1)                      // ... of a cross-evidence search: its rows of s1 against all of s2
        rc = mce_knn_dotp_f64_dev(j.dS1() + j.q_lo * (int64_t)q.d, j.q_hi - j.q_lo, j.dS2(), j.nr, q.d, q.kmax, 0, 0, j.dW() + j.q_lo, j.dF() + j.q_lo,
                                  j.dO(), nullptr, j.ws(), j.wsb, st);
    else
        rc = mce_knn_dotp_f64_dev(j.dS1(), q.n1, q.S2 ? j.dS2() : j.dS1(), j.nr, q.d, q.kmax, j.k0, 0, j.dW(), j.dF(), j.dO(),
                                  j.nverify > 0 ? j.d_vdist() : nullptr, j.ws(), j.wsb, st);
    if (rc != MCE_OK) return rc;
    MCE_HIP(hipMemcpyAsync(j.h_dotp(), j.dO(), (size_t)q.kmax * sizeof(double), hipMemcpyDeviceToHost, st));
    if (j.nverify > 0) {
        // mce_options.verify: the whitened rows the search ran on are still here; re-check a sample of them (stream-ordered)
        rc = mce_verify_knn_f64_dev(j.dS1(), q.n1, q.S2 ? j.dS2() : j.dS1(), j.nr, q.d, j.K, j.k0 == 1 ? MCE_SELF_EXCLUDE : MCE_SELF_NONE, 0, j.d_vdist(),
                                    j.K, j.nverify, 0x9E3779B97F4A7C15ull * (unsigned long long)(j.index + 1), j.d_vres(), j.dbase + j.o_vw,
                                    mce_verify_workspace_bytes(j.nverify, j.K), st);
        if (rc != MCE_OK) return rc;
        MCE_HIP(hipMemcpyAsync(j.h_verify(), j.d_vres(), 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    }
    return MCE_OK;
}

void feed_stage_d(FeedJob& j)
{
    mce_feed_problem& q = *j.q;
    std::copy(j.h_dotp(), j.h_dotp() + q.kmax, q.dotp);
    q.jacobian = j.jac;
    if (q.eigenvalues) std::copy(j.lam.begin(), j.lam.end(), q.eigenvalues);
    if (j.want_sum) j.checksum = *j.h_sum();
    g_last_verify_rows.store(j.nverify > 0 ? j.h_verify()[0] : 0);
    if (j.nverify > 0 && j.h_verify()[1] != 0)
        j.set_error(fail(MCE_ERR_VERIFY, "k-NN re-check failed: %d of %d sampled query rows have a neighbour list that an exact fp64 scan of all %lld reference rows "
                         "contradicts", j.h_verify()[1], j.h_verify()[0], (long long)j.nr));
}

// all jobs of one device, in waves; per-job failures are recorded in the job, a failure of the
// machinery itself (allocation, stream) is returned
int feed_run_on_device(int device, std::vector<FeedJob*>& jobs)
{
    int rc = select_device(device);
    if (rc != MCE_OK) return rc;
    // two stream sets so that the covariance of the NEXT group never queues behind the searches of
    // the current one; a single problem runs on the default stream
    std::vector<hipStream_t> sa, sc;
    std::vector<hipEvent_t> events;
    struct Guard {
        std::vector<hipStream_t>&a, &c;
        std::vector<hipEvent_t>& e;
        ~Guard()
        {
            for (hipStream_t x : a) (void)hipStreamDestroy(x);
            for (hipStream_t x : c) (void)hipStreamDestroy(x);
            for (hipEvent_t x : e) (void)hipEventDestroy(x);
        }
    } guard{sa, sc, events};
    const bool piped = jobs.size() > 1;
    if (piped) {
        const int ns = (int)std::min<size_t>(kFeedStreams, jobs.size());
        for (int i = 0; i < ns; ++i) {
            hipStream_t s;
            MCE_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            sa.push_back(s);
            MCE_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            sc.push_back(s);
        }
        for (int i = 0; i < kFeedGroup; ++i) {
            hipEvent_t e;
            MCE_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            events.push_back(e);
        }
    }
    size_t wave_bytes = kWaveBytes;
    if (const size_t wb = read_tuning().feed_wave_bytes) wave_bytes = wb;   // tests: force several waves
    size_t lo = 0;
    while (lo < jobs.size()) {
        size_t hi = lo, dev_bytes = 0, host_bytes = 0;
        while (hi < jobs.size() && (hi == lo || (dev_bytes + jobs[hi]->dev_bytes <= wave_bytes && hi - lo < (size_t)kWaveMaxJobs))) {
            dev_bytes += jobs[hi]->dev_bytes;
            host_bytes += jobs[hi]->host_bytes;
            ++hi;
        }
        DevBuf arena;
        // destroyed BEFORE the arena: an early return (a failing HIP call in the loops below) must not hand the
        // arena back to the pool while kernels of other jobs are still running on the other streams
        struct Quiesce {
            bool armed = true;
            ~Quiesce() { if (armed) (void)hipDeviceSynchronize(); }
        } quiesce;
        PinnedArena& pinned = g_pinned;
        MCE_HIP(arena.alloc(dev_bytes));
        MCE_HIP(pinned.reserve(host_bytes));
        size_t doff = 0, hoff = 0;
        for (size_t i = lo; i < hi; ++i) {
            jobs[i]->dbase = static_cast<char*>(arena.p) + doff;
            jobs[i]->hbase = reinterpret_cast<double*>(static_cast<char*>(pinned.p) + hoff);
            doff += jobs[i]->dev_bytes;
            hoff += jobs[i]->host_bytes;
        }
        if (!piped) {
            FeedJob& j = *jobs[lo];
            if (j.rc == MCE_OK) {
                int r = feed_stage_a(j, nullptr);
                if (r == MCE_OK) { MCE_HIP(hipStreamSynchronize(nullptr)); r = feed_stage_b(j); }
                if (r == MCE_OK) r = feed_stage_c(j, nullptr);
                if (r == MCE_OK) { MCE_HIP(hipStreamSynchronize(nullptr)); feed_stage_d(j); }
                else { j.set_error(r); (void)hipStreamSynchronize(nullptr); }
            }
            quiesce.armed = false;
            lo = hi;
            continue;
        }
        // groups of kFeedGroup problems, two deep: while the device searches group g the host uploads
        // group g+1 (blocking pageable copies) and its covariance kernels run beside the searches
        auto stage_a_group = [&](size_t g0, size_t g1) -> int {
            for (size_t i = g0; i < g1; ++i) {
                FeedJob& j = *jobs[i];
                if (j.rc != MCE_OK) continue;
                hipStream_t st = sa[i % sa.size()];
                const int r = feed_stage_a(j, st);
                if (r != MCE_OK) { j.set_error(r); continue; }
                MCE_HIP(hipEventRecord(events[i - g0], st));
            }
            return MCE_OK;
        };
        rc = stage_a_group(lo, std::min(hi, lo + (size_t)kFeedGroup));
        if (rc != MCE_OK) return rc;
        for (size_t g0 = lo; g0 < hi; g0 += kFeedGroup) {
            const size_t g1 = std::min(hi, g0 + (size_t)kFeedGroup);
            for (size_t i = g0; i < g1; ++i) {
                FeedJob& j = *jobs[i];
                if (j.rc != MCE_OK) continue;
                MCE_HIP(hipEventSynchronize(events[i - g0]));
                int r = feed_stage_b(j);
                if (r == MCE_OK) r = feed_stage_c(j, sc[i % sc.size()]);
                if (r != MCE_OK) j.set_error(r);
            }
            if (g1 < hi) {
                rc = stage_a_group(g1, std::min(hi, g1 + (size_t)kFeedGroup));
                if (rc != MCE_OK) return rc;
            }
        }
        for (hipStream_t st : sc) MCE_HIP(hipStreamSynchronize(st));
        for (hipStream_t st : sa) MCE_HIP(hipStreamSynchronize(st));     // (jobs that failed after stage A left work there)
        quiesce.armed = false;
        for (size_t i = lo; i < hi; ++i)
            if (jobs[i]->rc == MCE_OK) feed_stage_d(*jobs[i]);
        lo = hi;
    }
    return MCE_OK;
}

}  // namespace

extern "C" {

size_t mce_feed_problem_size(void) { return sizeof(mce_feed_problem); }

int mce_evidence_feed_batch_f64(mce_feed_problem* problems, int64_t nprob, const int32_t* devices, int32_t ndev)
{
    if (nprob < 0 || (nprob > 0 && !problems)) return fail(MCE_ERR_INVALID, "invalid problem list");
    if (nprob == 0) return MCE_OK;
    std::vector<FeedJob> jobs((size_t)nprob);
    for (int64_t i = 0; i < nprob; ++i) {
        jobs[i].q = &problems[i];
        jobs[i].index = i;
        problems[i].status = MCE_OK;
        problems[i].jacobian = 0.0;
        const int r = feed_plan(jobs[i]);
        if (r != MCE_OK) jobs[i].set_error(r);
    }
    std::vector<int> devs;
    if (!devices || ndev <= 0) devs.push_back(0);
    else devs.assign(devices, devices + ndev);
    const int n = (int)std::min<int64_t>((int64_t)devs.size(), nprob);
    // greedy balance by pair count (largest first), then restore the caller's order per device
    std::vector<std::vector<FeedJob*>> per_dev(n);
    if (n == 1) {
        for (auto& j : jobs) if (j.rc == MCE_OK) per_dev[0].push_back(&j);
    } else {
        std::vector<FeedJob*> order;
        for (auto& j : jobs) if (j.rc == MCE_OK) order.push_back(&j);
        std::stable_sort(order.begin(), order.end(), [](const FeedJob* a, const FeedJob* b) {
            return (double)a->q->n1 * (double)a->nr > (double)b->q->n1 * (double)b->nr;
        });
        std::vector<double> load(n, 0.0);
        for (FeedJob* j : order) {
            const int t = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            load[t] += (double)j->q->n1 * (double)j->nr + 1e6;
            per_dev[t].push_back(j);
        }
        for (auto& v : per_dev) std::sort(v.begin(), v.end(), [](const FeedJob* a, const FeedJob* b) { return a->index < b->index; });
    }
    std::vector<int> rcs(n, MCE_OK);
    std::vector<std::string> errs(n);
    auto work = [&](int i) {
        if (per_dev[i].empty()) return;
        rcs[i] = feed_run_on_device(devs[i], per_dev[i]);
        if (rcs[i] != MCE_OK) errs[i] = g_err;
    };
    if (n == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        const CallOptions inherited = t_opt;
        for (int i = 0; i < n; ++i) th.emplace_back([&, i]() { t_opt = inherited; work(i); });
        for (auto& t : th) t.join();
    }
    for (int i = 0; i < n; ++i)
        if (rcs[i] != MCE_OK) return fail(rcs[i], "device %d: %s", devs[i], errs[i].c_str());
    int first = MCE_OK;
    for (int64_t i = 0; i < nprob; ++i) {
        problems[i].status = jobs[i].rc;
        if (jobs[i].rc != MCE_OK && first == MCE_OK) {
            first = jobs[i].rc;
            if (nprob == 1) fail(first, "%s", jobs[i].err.c_str());
            else fail(first, "problem %lld: %s", (long long)i, jobs[i].err.c_str());
        }
    }
    return first;
}

int mce_evidence_feed_f64(const double* S1, int64_t n1, int64_t ld1, const double* S2, int64_t n2, int64_t ld2,
                          int32_t d, int32_t cov_mode, int32_t kmax, const double* w, const double* fs,
                          double* dotp, double* jacobian, double* eigenvalues, int32_t device)
{
    if (!jacobian) return fail(MCE_ERR_INVALID, "null pointer argument");
    mce_feed_problem q;
    std::memset(&q, 0, sizeof(q));
    q.S1 = S1; q.n1 = n1; q.ld1 = ld1;
    q.S2 = S2; q.n2 = S2 ? n2 : 0; q.ld2 = S2 ? ld2 : 0;
    q.d = d; q.cov_mode = cov_mode; q.kmax = kmax;
    q.w = w; q.fs = fs; q.dotp = dotp; q.eigenvalues = eigenvalues;
    const int rc = mce_evidence_feed_batch_f64(&q, 1, &device, 1);
    if (rc == MCE_OK) *jacobian = q.jacobian;
    return rc;
}


static int feed_part_impl(bool src_device, const double* S1, int64_t n1, int64_t ld1, const double* S2, int64_t n2, int64_t ld2,
                          int32_t d, int32_t cov_mode, int32_t kmax, const double* w, const double* fs,
                          int32_t part, int32_t nparts, double* dotp_part, double* jacobian, double* eigenvalues,
                          uint64_t* checksum, int32_t device)
{
    if (!jacobian) return fail(MCE_ERR_INVALID, "null pointer argument");
    if (nparts < 1 || part < 0 || part >= nparts) return fail(MCE_ERR_INVALID, "part %d of %d", part, nparts);
    mce_feed_problem q;
    std::memset(&q, 0, sizeof(q));
    q.S1 = S1; q.n1 = n1; q.ld1 = ld1;
    q.S2 = S2; q.n2 = S2 ? n2 : 0; q.ld2 = S2 ? ld2 : 0;
    q.d = d; q.cov_mode = cov_mode; q.kmax = kmax;
    q.w = w; q.fs = fs; q.dotp = dotp_part; q.eigenvalues = eigenvalues;
    FeedJob job;
    job.q = &q;
    job.part = part;
    job.nparts = nparts;
    job.src_device = src_device;
    job.want_sum = checksum != nullptr;
    int rc = feed_plan(job);
    if (rc != MCE_OK) return rc;
    std::vector<FeedJob*> jobs{&job};
    rc = feed_run_on_device(device, jobs);
    if (rc != MCE_OK) return rc;
    if (job.rc != MCE_OK) return fail(job.rc, "%s", job.err.c_str());
    *jacobian = q.jacobian;
    if (checksum) *checksum = job.checksum;
    return MCE_OK;
}

int mce_evidence_feed_part_f64(const double* S1, int64_t n1, int64_t ld1, const double* S2, int64_t n2, int64_t ld2,
                               int32_t d, int32_t cov_mode, int32_t kmax, const double* w, const double* fs,
                               int32_t part, int32_t nparts, double* dotp_part, double* jacobian, double* eigenvalues,
                               uint64_t* checksum, int32_t device)
{
    return feed_part_impl(false, S1, n1, ld1, S2, n2, ld2, d, cov_mode, kmax, w, fs, part, nparts, dotp_part, jacobian, eigenvalues, checksum, device);
}

int mce_evidence_feed_part_dev_f64(const double* dS1, int64_t n1, int64_t ld1, const double* dS2, int64_t n2, int64_t ld2,
                                   int32_t d, int32_t cov_mode, int32_t kmax, const double* d_w, const double* d_fs,
                                   int32_t part, int32_t nparts, double* dotp_part, double* jacobian, double* eigenvalues,
                                   uint64_t* checksum, int32_t device)
{
    return feed_part_impl(true, dS1, n1, ld1, dS2, n2, ld2, d, cov_mode, kmax, d_w, d_fs, part, nparts, dotp_part, jacobian, eigenvalues, checksum, device);
}

static int feed_whiten_impl(bool src_device, const double* S1, int64_t n1, int64_t ld1, int32_t d, int32_t kmax, const double* w, const double* fs,
                            double* d_X_out, double* d_w_out, double* d_fs_out, double* jacobian, double* eigenvalues,
                            uint64_t* checksum, int32_t device)
{
    if (!jacobian || !d_X_out || !d_w_out || !d_fs_out) return fail(MCE_ERR_INVALID, "null pointer argument");
    mce_feed_problem q;
    std::memset(&q, 0, sizeof(q));
    double sums[MCE_MAX_K + 2];
    q.S1 = S1; q.n1 = n1; q.ld1 = ld1;
    q.d = d; q.cov_mode = 0; q.kmax = kmax;
    q.w = w; q.fs = fs; q.dotp = sums; q.eigenvalues = eigenvalues;
    if (kmax < 2 || kmax > MCE_MAX_K + 1) return fail(MCE_ERR_K_RANGE, "kmax=%d", kmax);
    FeedJob job;
    job.q = &q;
    job.out_X = d_X_out; job.out_w = d_w_out; job.out_f = d_fs_out;
    job.src_device = src_device;
    job.want_sum = checksum != nullptr;
    int rc = feed_plan(job);
    if (rc != MCE_OK) return rc;
    std::vector<FeedJob*> jobs{&job};
    rc = feed_run_on_device(device, jobs);
    if (rc != MCE_OK) return rc;
    if (job.rc != MCE_OK) return fail(job.rc, "%s", job.err.c_str());
    *jacobian = q.jacobian;
    if (checksum) *checksum = job.checksum;
    return MCE_OK;
}

int mce_evidence_feed_whiten_f64(const double* S1, int64_t n1, int64_t ld1, int32_t d, int32_t kmax, const double* w, const double* fs,
                                 double* d_X_out, double* d_w_out, double* d_fs_out, double* jacobian, double* eigenvalues,
                                 uint64_t* checksum, int32_t device)
{
    return feed_whiten_impl(false, S1, n1, ld1, d, kmax, w, fs, d_X_out, d_w_out, d_fs_out, jacobian, eigenvalues, checksum, device);
}

int mce_evidence_feed_whiten_dev_f64(const double* dS1, int64_t n1, int64_t ld1, int32_t d, int32_t kmax, const double* d_w, const double* d_fs,
                                     double* d_X_out, double* d_w_out, double* d_fs_out, double* jacobian, double* eigenvalues,
                                     uint64_t* checksum, int32_t device)
{
    return feed_whiten_impl(true, dS1, n1, ld1, d, kmax, d_w, d_fs, d_X_out, d_w_out, d_fs_out, jacobian, eigenvalues, checksum, device);
}

int mce_knn_dotp_f64(const double* X, int64_t nq, const double* Y, int64_t nr, int32_t d, int32_t kmax,
                     int32_t k0, int64_t self_offset, const double* w, const double* fs, double* dotp,
                     double* dist_out, const int32_t* devices, int32_t ndev)
{
    if (!X || !Y || !w || !fs || !dotp) return fail(MCE_ERR_INVALID, "null pointer argument");
    if (k0 != 0 && k0 != 1) return fail(MCE_ERR_INVALID, "k0 must be 0 (cross) or 1 (auto), got %d", k0);
    if (kmax <= k0 || nq < 1) return fail(MCE_ERR_INVALID, "invalid kmax=%d k0=%d nq=%lld", kmax, k0, (long long)nq);
    {   // validate before touching any device
        Plan p;
        int rc = make_plan(nq, nr, d, kmax - k0, k0 == 1 ? MCE_SELF_EXCLUDE : MCE_SELF_NONE, p);
        if (rc != MCE_OK) return rc;
    }
    std::vector<int> devs;
    if (!devices || ndev <= 0) devs.push_back(0);
    else devs.assign(devices, devices + ndev);
    const int n = (int)std::min<int64_t>((int64_t)devs.size(), nq);
    std::vector<std::vector<double>> parts(n, std::vector<double>(kmax, 0.0));
    std::vector<int> rcs(n, MCE_OK);
    std::vector<std::string> errs(n);
    // auto evidence over one set: let each device take a library-chosen part (rows for the sweep, blocks of the
    // shared k-d order for the pruned walk) instead of a row range
    const bool whole_set = n > 1 && X == Y && nq == nr && k0 == 1 && self_offset == 0 && !dist_out;
    auto work = [&](int i) {
        const int64_t lo = nq * i / n, hi = nq * (i + 1) / n;
        if (whole_set) rcs[i] = mce_knn_dotp_part_f64(Y, nr, d, kmax, i, n, w, fs, parts[i].data(), devs[i]);
        else rcs[i] = fused_on_device(devs[i], X, lo, hi, Y, nr, d, kmax, k0, self_offset, w, fs, parts[i].data(), dist_out);
        if (rcs[i] != MCE_OK) errs[i] = g_err;   // g_err is thread-local
    };
    if (n == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        const CallOptions inherited = t_opt;
        for (int i = 0; i < n; ++i) th.emplace_back([&, i]() { t_opt = inherited; work(i); });
        for (auto& t : th) t.join();
    }
    for (int i = 0; i < n; ++i)
        if (rcs[i] != MCE_OK) return fail(rcs[i], "device %d: %s", devs[i], errs[i].c_str());
    for (int k = 0; k < kmax; ++k) {   // fixed device order -> reproducible
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += parts[i][k];
        dotp[k] = s;
    }
    return MCE_OK;
}

int mce_knn_dotp_part_f64(const double* Y, int64_t nr, int32_t d, int32_t kmax, int32_t part, int32_t nparts, const double* w,
                          const double* fs, double* dotp, int32_t device)
{
    if (!Y || !w || !fs || !dotp) return fail(MCE_ERR_INVALID, "null pointer argument");
    if (kmax <= 1 || nr < 1) return fail(MCE_ERR_INVALID, "invalid kmax=%d nr=%lld", kmax, (long long)nr);
    Plan p;
    int rc = make_plan(nr, nr, d, kmax - 1, MCE_SELF_EXCLUDE, p);
    if (rc != MCE_OK) return rc;
    rc = select_device(device);
    if (rc != MCE_OK) return rc;
    const size_t wsb = p.total + dotp_ws_bytes(nr, kmax);
    DevBuf dY, dW, dF, dO, ws;
    MCE_HIP(dY.alloc((size_t)nr * d * sizeof(double)));
    MCE_HIP(dW.alloc((size_t)nr * sizeof(double)));
    MCE_HIP(dF.alloc((size_t)nr * sizeof(double)));
    MCE_HIP(dO.alloc((size_t)kmax * sizeof(double)));
    MCE_HIP(ws.alloc(wsb));
    MCE_HIP(hipMemcpy(dY.p, Y, (size_t)nr * d * sizeof(double), hipMemcpyHostToDevice));
    MCE_HIP(hipMemcpy(dW.p, w, (size_t)nr * sizeof(double), hipMemcpyHostToDevice));
    MCE_HIP(hipMemcpy(dF.p, fs, (size_t)nr * sizeof(double), hipMemcpyHostToDevice));
    rc = mce_knn_dotp_part_f64_dev(dY.as<double>(), nr, d, kmax, part, nparts, dW.as<double>(), dF.as<double>(), dO.as<double>(), ws.p, wsb, nullptr);
    if (rc != MCE_OK) return rc;
    MCE_HIP(hipDeviceSynchronize());
    MCE_HIP(hipMemcpy(dotp, dO.p, (size_t)kmax * sizeof(double), hipMemcpyDeviceToHost));
    return MCE_OK;
}

}  // extern "C"
