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
// Survivors are queued per wave in LDS and drained in batches: every lane takes one
// (query,row) pair and computes the exact fp64 distance from the ORIGINAL rows; accepted
// ones go through the same whole-wave sorted insertion into the per-query LDS list as in
// knn_mfma.hpp, and the gates G_q are refreshed from the lists.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mce {

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kHWaves = 8;
constexpr int kHThreads = kHWaves * 64;
constexpr int kHQueue = 640;          // candidate queue entries (4 B) per wave: 512 (half a tile) + 128
constexpr int kHRelBits = 26;         // queue entry = query-local (6 bits) << 26 | row - first row of the split
constexpr double kHTargetRadius = 200.0;

// device-side scalars shared by the f16 kernels (doubles; maxima kept as bit patterns)
enum { HP_RMAX = 0, HP_SCALE = 1, HP_EY = 2, HP_YHATMAX = 3, HP_RHO = 4, HP_COUNT = 8 };

__host__ __device__ constexpr int f16_ksteps(int D) { return (D + 3 + 15) / 16; }          // 16-wide k-steps
__host__ __device__ constexpr int f16_qt(int KCAP) { return KCAP > 12 ? 1 : 2; }          // 32-query tiles per wave
__host__ __device__ constexpr bool f16_supported(int D, int K) { return D >= 2 && f16_ksteps(D) <= 4 && K <= 16; }
__host__ __device__ constexpr int f16_qpb(int KCAP) { return kHWaves * f16_qt(KCAP) * 32; }
// 32-row reference tiles per LDS chunk (tile = KST KB): <= 32 KB per buffer, even count,
// and a whole number of 16-byte vectors per thread (CT*KST % 8 == 0)
__host__ __device__ constexpr int f16_chunk_tiles(int KST) { return KST == 1 ? 32 : (KST == 2 ? 16 : 8); }
__host__ __device__ constexpr size_t f16_lds_bytes(int KST, int KCAP)
{
    return (size_t)2 * f16_chunk_tiles(KST) * KST * 1024             // staging
           + (size_t)f16_qpb(KCAP) * KCAP * 12                         // lists
           + (size_t)kHWaves * kHQueue * 4 + 1024;                     // queues + slack
}

// ---------------------------------------------------------------------------
// the filter + exact-refine search kernel
//   grid.x = nqblk * rsplit (as knn_mfma_kernel); lists out: part_d/part_i [rsplit][KCAP][nq_pad]
//   with EXACT squared distances as keys.
// ---------------------------------------------------------------------------
template <int KST, int KCAP>
__global__ __launch_bounds__(kHThreads, 2) void knn_f16_kernel(
    const _Float16* __restrict__ Yh, int64_t nchunk_total, int rsplit,
    const _Float16* __restrict__ Xh, const double* __restrict__ qinfo, const double* __restrict__ params,
    const double* __restrict__ X, const double* __restrict__ Y, int64_t nq, int64_t nr, int D,
    int64_t nq_pad, int nqblk, int self_exclude, int64_t self_offset, int ksel,
    double* __restrict__ part_d, int* __restrict__ part_i)
{
    constexpr int QT = f16_qt(KCAP);
    constexpr int QPW = QT * 32;                         // queries per wave
    constexpr int QPB = kHWaves * QPW;
    constexpr int CT = f16_chunk_tiles(KST);
    static_assert(CT % 2 == 0, "tile loop is unrolled by two");
    constexpr int CHUNK_BYTES = CT * KST * 1024;
    constexpr int CHUNK_VEC = CHUNK_BYTES / 16;
    constexpr int VPT = (CHUNK_VEC + kHThreads - 1) / kHThreads;
    static_assert(CHUNK_VEC % kHThreads == 0, "chunk must be a whole number of 16-byte vectors per thread");
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    // LDS map: [2 staging buffers][list keys QPB*KCAP f64][list rows QPB*KCAP i32][queues][slack]
    char* const stage0 = lds_raw;
    double* const list_d = reinterpret_cast<double*>(lds_raw + 2 * CHUNK_BYTES);
    int* const list_i = reinterpret_cast<int*>(list_d + QPB * KCAP);
    int* const queue_all = list_i + QPB * KCAP;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qblk = blockIdx.x % nqblk;
    const int split = blockIdx.x / nqblk;

    const int64_t cps = (nchunk_total + rsplit - 1) / rsplit;
    const int64_t c_begin = (int64_t)split * cps;
    int64_t c_end = c_begin + cps;
    if (c_end > nchunk_total) c_end = nchunk_total;

    const double INF = __builtin_huge_val();
    double* const wl_d = list_d + wave * (QPW * KCAP);
    int* const wl_i = list_i + wave * (QPW * KCAP);
    int* const wq = queue_all + wave * kHQueue;                // packed (query-local, relative row)
    const int jsplit0 = (int)(c_begin * (CT * 32));            // first reference row of this split
    for (int e = lane; e < QPW * KCAP; e += 64) { wl_d[e] = INF; wl_i[e] = -1; }

    const int64_t qwave0 = (int64_t)qblk * QPB + wave * QPW;     // first query of this wave

    // ---- B fragments (fp16 query rows) + per-query gate constants ---------------
    v8h b[QT][KST];
    double gate_a[QT], gate_xn[QT], gate_eps[QT];                // e_x + E_y ; |x^|^2 ; eps_q
    float G[QT];
    bool qlive[QT];
    const double s2 = params[HP_SCALE] * params[HP_SCALE];
    {
        const double Ey = params[HP_EY], Yhm = params[HP_YHATMAX], rho = params[HP_RHO];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int64_t q = qwave0 + qt * 32 + (lane & 31);
#pragma unroll
            for (int ks = 0; ks < KST; ++ks)
                b[qt][ks] = *reinterpret_cast<const v8h*>(Xh + q * (int64_t)(16 * KST) + 16 * ks + 8 * (lane >> 5));
            const double ex = qinfo[2 * q], xn = qinfo[2 * q + 1];
            const double r = sqrt(xn) + Yhm;
            gate_a[qt] = (ex + Ey) * (1.0 + 1e-9) + 1e-300;
            gate_xn[qt] = xn;
            gate_eps[qt] = (32.0 * KST) * 0x1p-24 * r * r * (1.0 + 1e-9) + rho + 1e-30;
            qlive[qt] = q < nq;
            G[qt] = qlive[qt] ? __builtin_huge_valf() : -__builtin_huge_valf();     // padding queries never pass
        }
    }
    const int k_last = ksel - 1;
    auto gate_of = [&](double thr, int qt) -> float {       // thr: exact squared distance, input units
        if (!qlive[qt]) return -__builtin_huge_valf();
        if (!(thr < INF)) return __builtin_huge_valf();
        const double rr = sqrt(thr * s2) * (1.0 + 1e-12) + gate_a[qt];
        const double g = rr * rr * (1.0 + 1e-12) - gate_xn[qt] + gate_eps[qt];
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

    // one 32-row tile: KST A-fragment reads (16 B per lane) + QT chains of KST MFMAs
    auto mfma_tile = [&](const char* lp, v16f (&acc)[QT]) {
        v8h a[KST];
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) a[ks] = *reinterpret_cast<const v8h*>(lp + ks * 1024);
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

    // whole-wave sorted insertion of one accepted candidate (see knn_mfma.hpp)
    auto insert_one = [&](int ql, double vv, int jj) {
        double* ld = wl_d + ql * KCAP;
        int* li = wl_i + ql * KCAP;
        const int i = lane;
        const double e_i = ld[i];
        const double e_p = ld[i - 1];
        const int id_i = li[i];
        const int id_p = li[i - 1];
        const bool c_i = (vv < e_i) || (vv == e_i && jj < id_i);
        const bool c_p = (i > 0) && ((vv < e_p) || (vv == e_p && jj < id_p));
        const double n_e = c_p ? e_p : (c_i ? vv : e_i);
        const int n_id = c_p ? id_p : (c_i ? jj : id_i);
        if (i < KCAP) { ld[i] = n_e; li[i] = n_id; }
    };

    int qcount = 0;   // wave-uniform number of queued candidates

    // exact evaluation + insertion of everything queued, then refresh the gates
    auto drain = [&]() {
        for (int b0 = 0; b0 < qcount; b0 += 64) {
            const int e = b0 + lane;
            const bool valid = e < qcount;
            int ql = 0, j = 0;
            if (valid) {
                const unsigned ent = (unsigned)wq[e];
                ql = (int)(ent >> kHRelBits);
                j = jsplit0 + (int)(ent & ((1u << kHRelBits) - 1u));
            }
            const int64_t q = qwave0 + ql;
            double d2 = INF;
            if (valid && j < nr && q < nq && !(self_exclude && (int64_t)j == self_offset + q)) {
                const double* x = X + q * (int64_t)D;
                const double* y = Y + (int64_t)j * D;
                double acc0 = 0.0, acc1 = 0.0;
                int i = 0;
                for (; i + 1 < D; i += 2) {
                    const double t0 = x[i] - y[i], t1 = x[i + 1] - y[i + 1];
                    acc0 = fma(t0, t0, acc0);
                    acc1 = fma(t1, t1, acc1);
                }
                if (i < D) { const double t0 = x[i] - y[i]; acc0 = fma(t0, t0, acc0); }
                d2 = acc0 + acc1;
            }
            double thrq = INF;
            if (valid) thrq = wl_d[ql * KCAP + k_last];
            unsigned long long m = __ballot(d2 < thrq || (d2 == thrq && d2 < INF));
            while (m) {
                const int src = __builtin_ctzll(m);
                m &= m - 1;
                const int lo = __builtin_amdgcn_readlane(__double2loint(d2), src);
                const int hi = __builtin_amdgcn_readlane(__double2hiint(d2), src);
                const int qq = __builtin_amdgcn_readlane(ql, src);
                const int jj = __builtin_amdgcn_readlane(j, src);
                insert_one(qq, __hiloint2double(hi, lo), jj);
            }
        }
        qcount = 0;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) G[qt] = gate_of(wl_d[(qt * 32 + (lane & 31)) * KCAP + k_last], qt);
    };

    // gate + enqueue for one finished tile; jb0 = first reference row of the tile.
    // C layout of 32x32 f32: lane l -> query column l&31, rows (r&3) + 8*(r>>2) + 4*(l>>5)
    auto process = [&](const v16f (&acc)[QT], int jb0) {
        bool pass = false;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const v16f& c = acc[qt];
            float m0 = fminf(fminf(c[0], c[1]), c[2]);
            float m1 = fminf(fminf(c[3], c[4]), c[5]);
            float m2 = fminf(fminf(c[6], c[7]), c[8]);
            float m3 = fminf(fminf(c[9], c[10]), c[11]);
            float m4 = fminf(fminf(c[12], c[13]), c[14]);
            m0 = fminf(fminf(m0, m1), m2);
            m3 = fminf(fminf(m3, m4), c[15]);
            pass |= fminf(m0, m3) <= G[qt];
        }
        if (__any(pass)) {
            const int jrel0 = jb0 - jsplit0;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if (qcount > kHQueue - 512) drain();          // room for half a tile (8 x 64 entries)
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) {
                        const int r = half * 8 + rr;
                        const bool p = acc[qt][r] <= G[qt];
                        const unsigned long long m = __ballot(p);
                        if (m) {
                            if (p) {
                                const int slot = qcount + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                                const unsigned rel = (unsigned)(jrel0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5));
                                wq[slot] = (int)(((unsigned)(qt * 32 + (lane & 31)) << kHRelBits) | rel);
                            }
                            qcount += __builtin_popcountll(m);
                        }
                    }
                }
            }
            if (qcount >= 64) drain();
        }
    };

    v16f accA[QT], accB[QT];
    int jbA = 0, jbB = 0;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[qt][r] = __builtin_nanf("");       // "no pending tile": NaN never passes the gate

    if (c_begin < c_end) stage_async(c_begin, 0);

    for (int64_t c = c_begin; c < c_end; ++c) {
        const int buf = (int)((c - c_begin) & 1);
        __syncthreads();
        if ((c + 1) < c_end) stage_async(c + 1, buf ^ 1);
        const char* lbuf = stage0 + buf * CHUNK_BYTES + lane * 16;
        const int jchunk = (int)(c * (CT * 32));
#pragma unroll 1
        for (int t = 0; t < CT; t += 2) {
            mfma_tile(lbuf + (t * KST) * 1024, accA);
            jbA = jchunk + t * 32;
            process(accB, jbB);
            mfma_tile(lbuf + ((t + 1) * KST) * 1024, accB);
            jbB = jchunk + (t + 1) * 32;
            process(accA, jbA);
        }
    }
    process(accB, jbB);
    drain();

    // ---- write this wave's lists: lane -> (query lane&31, slots (lane>>5) + 2i) ----
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int ql = qt * 32 + (lane & 31);
        const int64_t q = qwave0 + ql;
        for (int k = lane >> 5; k < KCAP; k += 2) {
            const int64_t o = ((int64_t)split * KCAP + k) * nq_pad + q;
            part_d[o] = wl_d[ql * KCAP + k];
            part_i[o] = wl_i[ql * KCAP + k];
        }
    }
}

}  // namespace mce
