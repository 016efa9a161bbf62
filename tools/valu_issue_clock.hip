// valu_issue_clock.hip -- microbenchmark (round 6): what does ONE wave64 vector instruction cost in
// issue cycles on an MI355X SIMD, by instruction class, at 1 / 2 / 3 / 4 waves per SIMD, alone and beside a stream of
// v_mfma_f32_32x32x16_f16 -- in a partner wave of the same SIMD and interleaved in the same wave?
// (MI355X_MICROARCH.md: CDNA4 SIMDs are SIMD-32, a wave64 fp32 op issues over 2 cycles, fp64 over 4; bench.py priced the
//  pruned walk's roofline at one instruction per 4 cycles whatever the class.  This tool settles the rates bench.py uses.)
// Every class is a block of 32 INDEPENDENT instructions (8 registers x 4) in inline asm -- no dependent-issue stalls, nothing
// for the compiler to fold -- inside a counted loop (3 scalar instructions per 32 vector ones).
// Output: one JSON object per line (tools/valu_issue_report.py collects them into profiles/<round>/valu_issue_clock.json).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_issue_clock.hip -o tools/valu_issue_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <vector>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

enum { C_FMA32 = 0, C_MIN3, C_CMP, C_CMPBR, C_PKFMA, C_FMA64, C_ADD64, C_INT, C_CNDMASK, C_MOV, C_CVT, C_READLANE, C_MAXMIN, C_SELECT, C_NCLS };
static const char* kNames[C_NCLS] = {"v_fma_f32", "v_min3_f32", "v_cmp_lt_f32", "v_cmp_lt_f32+s_cbranch_vccnz", "v_pk_fma_f32", "v_fma_f64", "v_add_f64",
                                     "v_add_u32/v_lshlrev/v_and", "v_cndmask_b32", "v_mov_b32", "v_cvt_f32_f64/v_cvt_f64_f32", "v_readlane_b32",
                                     "v_max_f32/v_min_f32", "v_cmp_lt_f32+v_cndmask_b32"};

// 32 independent instructions of class CLS on the registers r[0..7] (f32) / d[0..7] (f64)
template <int CLS>
__device__ __forceinline__ void block32(float (&r)[8], double (&d)[8], v2f (&p)[8], float k0, float k1, int& sacc)
{
#define REP4(X) X X X X
    if constexpr (CLS == C_FMA32) {
        REP4(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                          : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(k0), "v"(k1));)
    } else if constexpr (CLS == C_MIN3) {
        REP4(asm volatile("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n"
                          "v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9"
                          : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(k0), "v"(k1));)
    } else if constexpr (CLS == C_MAXMIN) {
        REP4(asm volatile("v_max_f32 %0, %0, %8\n v_min_f32 %1, %1, %9\n v_max_f32 %2, %2, %8\n v_min_f32 %3, %3, %9\n"
                          "v_max_f32 %4, %4, %8\n v_min_f32 %5, %5, %9\n v_max_f32 %6, %6, %8\n v_min_f32 %7, %7, %9"
                          : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(k0), "v"(k1));)
    } else if constexpr (CLS == C_CMP) {
        REP4(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                          "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8"
                          : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(k0), "v"(k1) : "vcc");)
    } else if constexpr (CLS == C_CMPBR) {
        // compare + a branch on its result (never taken: k0 is larger than every r) -- the shape of the walk's tests and of an event
        REP4(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n s_cbranch_vccnz 1f\n v_cmp_gt_f32 vcc, %1, %8\n s_cbranch_vccnz 1f\n"
                          "v_cmp_gt_f32 vcc, %2, %8\n s_cbranch_vccnz 1f\n v_cmp_gt_f32 vcc, %3, %8\n s_cbranch_vccnz 1f\n"
                          "v_cmp_gt_f32 vcc, %4, %8\n s_cbranch_vccnz 1f\n v_cmp_gt_f32 vcc, %5, %8\n s_cbranch_vccnz 1f\n"
                          "v_cmp_gt_f32 vcc, %6, %8\n s_cbranch_vccnz 1f\n v_cmp_gt_f32 vcc, %7, %8\n s_cbranch_vccnz 1f\n1:"
                          : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(k0), "v"(k1) : "vcc");)
    } else if constexpr (CLS == C_PKFMA) {
        REP4(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                          "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8"
                          : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(p[0] * 0.f + 1.f));)
    } else if constexpr (CLS == C_FMA64) {
        REP4(asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n"
                          "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8"
                          : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"((double)k0));)
    } else if constexpr (CLS == C_ADD64) {
        REP4(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                          "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8"
                          : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"((double)k0));)
    } else if constexpr (CLS == C_INT) {
        REP4(asm volatile("v_add_u32 %0, %0, %8\n v_lshlrev_b32 %1, 1, %1\n v_and_b32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                          "v_lshlrev_b32 %4, 1, %4\n v_and_b32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_xor_b32 %7, %7, %8"
                          : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(k0));)
    } else if constexpr (CLS == C_CNDMASK) {
        REP4(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                          "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                          : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(k0) : "vcc");)
    } else if constexpr (CLS == C_SELECT) {
        // the select idiom: a compare and the conditional move that reads its mask (16 pairs = 32 instructions)
        REP4(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                          "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc"
                          : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(k0) : "vcc");)
    } else if constexpr (CLS == C_MOV) {
        REP4(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8"
                          : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(k0));)
    } else if constexpr (CLS == C_CVT) {
        REP4(asm volatile("v_cvt_f32_f64 %0, %8\n v_cvt_f64_f32 %9, %1\n v_cvt_f32_f64 %2, %10\n v_cvt_f64_f32 %11, %3\n"
                          "v_cvt_f32_f64 %4, %8\n v_cvt_f64_f32 %9, %5\n v_cvt_f32_f64 %6, %10\n v_cvt_f64_f32 %11, %7"
                          : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));)
    } else if constexpr (CLS == C_READLANE) {
        int s0, s1, s2, s3;
        REP4(asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %5, 5\n v_readlane_b32 %2, %6, 7\n v_readlane_b32 %3, %7, 9\n"
                          "v_readlane_b32 %0, %8, 3\n v_readlane_b32 %1, %9, 5\n v_readlane_b32 %2, %10, 7\n v_readlane_b32 %3, %11, 9"
                          : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7])); sacc += s0 ^ s1 ^ s2 ^ s3;)
    }
#undef REP4
}

// ROLE: 0 = every wave runs the class; 1 = waves 0..3 of a 512-thread workgroup stream MFMAs (one per SIMD), waves 4..7 run the
// class beside them; 2 = every wave interleaves ONE MFMA with NV instructions of the class (NV = 4 x REPS: the block32 above cut
// into quarters is not possible in asm, so NV is 32 x REPS per NMF MFMAs)
template <int CLS, int ROLE, int NMF>
__global__ __launch_bounds__(512) void k(float* out, long long* clk, int iters, const v8h* ab)
{
    float r[8];
    double d[8];
    v2f p[8];
    for (int i = 0; i < 8; ++i) { r[i] = (float)(threadIdx.x + i) * 1e-3f; d[i] = r[i]; p[i] = v2f{r[i], r[i]}; }
    const float k0 = 1e30f, k1 = 2e30f;
    int sacc = 0;
    v16f acc[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const v8h a = ab[threadIdx.x % 64], b = ab[64 + threadIdx.x % 64];
    const int wave = threadIdx.x >> 6;
    const bool mfma_wave = ROLE == 1 && wave < 4;
    __syncthreads();
    const long long w0 = wall_clock64();
    const long long c0 = clock64();
    if (mfma_wave) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            if constexpr (ROLE == 2) {
#pragma unroll
                for (int i = 0; i < NMF; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
            }
            block32<CLS>(r, d, p, k0, k1, sacc);
        }
    }
    const long long c1 = clock64();
    const long long w1 = wall_clock64();
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float s = (float)sacc;
    for (int i = 0; i < 8; ++i) s += r[i] + (float)d[i] + p[i][0] + p[i][1];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        long long* o = clk + ((size_t)blockIdx.x * 8 + wave) * 4;
        o[0] = c1 - c0; o[1] = w0; o[2] = w1; o[3] = (long long)(((unsigned long long)(xcc & 0xf) << 32) | hwid);
    }
}

static float* g_out; static long long* g_clk; static v8h* g_ab;
static int g_cus = 256;

template <int CLS, int ROLE, int NMF>
void run(int waves_per_simd, int iters)
{
    // ROLE 0 / 2: 256-thread workgroups, waves_per_simd of them per CU; ROLE 1: one 512-thread workgroup per CU (4 MFMA + 4 class waves)
    const int threads = ROLE == 1 ? 512 : 256;
    const int blocks = ROLE == 1 ? g_cus : g_cus * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<CLS, ROLE, NMF><<<blocks, threads>>>(g_out, g_clk, iters / 8, g_ab);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<CLS, ROLE, NMF><<<blocks, threads>>>(g_out, g_clk, iters, g_ab);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int wpb = threads / 64;
    std::vector<long long> hc((size_t)blocks * 8 * 4);
    hipMemcpy(hc.data(), g_clk, hc.size() * sizeof(long long), hipMemcpyDeviceToHost);
    const double n_valu = 32.0 * iters;                          // class instructions per class wave
    // per wave: shader cycles (s_memtime), start / end on the constant 100 MHz clock (s_memrealtime), and WHERE it ran (HW_ID: SIMD
    // [5:4], CU [11:8], SH [12], SE [15:13]; XCC_ID).  From these: the waves that shared a SIMD, how far their lifetimes overlapped,
    // the shader clock each wave saw -- so that "waves per SIMD" and "cycles" below are measured, not assumed from the launch.
    std::map<long long, std::vector<int>> by_simd;
    long long wmin = 0x7fffffffffffffffll, wmax = 0;
    double sum_ticks = 0, sum_wall = 0, sum_clock = 0;
    int nclass = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < wpb; ++w) {
            const long long* o = &hc[((size_t)b * 8 + w) * 4];
            const long long id = ((o[3] >> 32) << 16) | (o[3] & 0xfff0 & ~0xc0ll);      // xcc | se, sh, cu, simd (pipe and wave-slot bits dropped)
            by_simd[id].push_back(b * 8 + w);
            wmin = std::min(wmin, o[1]); wmax = std::max(wmax, o[2]);
            if (ROLE == 1 && w < 4) continue;
            sum_ticks += (double)o[0]; sum_wall += (double)(o[2] - o[1]);
            sum_clock += (double)o[0] / ((double)(o[2] - o[1]) * 10.0);                  // cycles per ns x 1000 = MHz / 1000 ... (100 MHz ticks = 10 ns)
            nclass += 1;
        }
    size_t maxw = 0, minw = 1u << 30;
    for (auto& kv : by_simd) { maxw = std::max(maxw, kv.second.size()); minw = std::min(minw, kv.second.size()); }
    const double ticks = sum_ticks / nclass;                      // mean shader cycles of a class wave
    const double span_ns = (double)(wmax - wmin) * 10.0, life_ns = sum_wall / nclass * 10.0;
    const double mhz = sum_clock / nclass * 1000.0;
    const int class_waves_per_simd = ROLE == 1 ? 1 : waves_per_simd;
    const double simds = (double)by_simd.size();
    const double inst_total = n_valu * nclass;
    printf("{\"class\": \"%s\", \"role\": \"%s\", \"waves_per_simd\": %d, \"mfma_per_32\": %d, \"ms\": %.3f, \"clock_mhz\": %.0f, "
           "\"cycles_per_inst_per_wave\": %.3f, \"simd_cycles_per_inst\": %.3f, \"simds_used\": %d, \"waves_on_a_simd_min_max\": [%d, %d], "
           "\"wave_lifetime_over_kernel_span\": %.3f, \"ns_per_inst_per_simd\": %.4f, \"cycles_at_2p4GHz_per_inst_per_simd\": %.3f",
           kNames[CLS], ROLE == 0 ? "alone" : (ROLE == 1 ? "beside an MFMA wave on the same SIMD" : "MFMAs interleaved in the same wave"),
           ROLE == 1 ? 2 : waves_per_simd, ROLE == 2 ? NMF : 0, ms, mhz, ticks / n_valu, ticks / n_valu / class_waves_per_simd, (int)by_simd.size(), (int)minw, (int)maxw,
           life_ns / span_ns, span_ns * simds / inst_total, span_ns * simds / inst_total * 2.4);
    if (ROLE == 1) printf(", \"mfma_wave_cycles_per_mfma\": %.2f", (double)hc[0] / (4.0 * iters));
    if (ROLE == 2) printf(", \"cycles_per_group\": %.2f, \"mfma_floor_cycles_per_group\": %d", ticks / iters, 32 * NMF);
    printf("}\n");
    fflush(stdout);
}

template <int CLS>
void run_class(int iters)
{
    for (int w = 1; w <= 4; ++w) run<CLS, 0, 0>(w, iters);
    run<CLS, 1, 0>(2, iters);
}

int main(int argc, char** argv)
{
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    g_cus = pr.multiProcessorCount;
    fprintf(stderr, "device: %s CUs=%d clockRate=%d MHz\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate / 1000);
    hipMalloc(&g_out, sizeof(float) * (size_t)g_cus * 4 * 512);
    hipMalloc(&g_clk, sizeof(long long) * 8 * 4 * (size_t)g_cus * 4);
    hipMalloc(&g_ab, sizeof(v8h) * 128);
    _Float16 h[128 * 8];
    srand(1);
    for (int i = 0; i < 128 * 8; ++i) h[i] = (_Float16)((rand() % 2001 - 1000) / 500.0f);
    hipMemcpy(g_ab, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = argc > 1 ? atoi(argv[1]) : 40000;
    run_class<C_FMA32>(iters);
    run_class<C_MIN3>(iters);
    run_class<C_MAXMIN>(iters);
    run_class<C_CMP>(iters);
    run_class<C_CMPBR>(iters);
    run_class<C_CNDMASK>(iters);
    run_class<C_SELECT>(iters);
    run_class<C_MOV>(iters);
    run_class<C_INT>(iters);
    run_class<C_PKFMA>(iters);
    run_class<C_FMA64>(iters);
    run_class<C_ADD64>(iters);
    run_class<C_CVT>(iters);
    run_class<C_READLANE>(iters);
    // how many vector instructions hide under an MFMA in the SAME wave: 32 instructions beside 1 / 2 / 4 / 8 MFMAs (32 / 16 / 8 / 4 per MFMA),
    // one and two waves per SIMD
    for (int w = 1; w <= 2; ++w) {
        run<C_MIN3, 2, 1>(w, iters); run<C_MIN3, 2, 2>(w, iters); run<C_MIN3, 2, 4>(w, iters); run<C_MIN3, 2, 8>(w, iters);
        run<C_FMA32, 2, 2>(w, iters); run<C_FMA32, 2, 4>(w, iters); run<C_FMA32, 2, 8>(w, iters);
        run<C_CMP, 2, 4>(w, iters); run<C_FMA64, 2, 4>(w, iters); run<C_FMA64, 2, 8>(w, iters);
    }
    return 0;
}
