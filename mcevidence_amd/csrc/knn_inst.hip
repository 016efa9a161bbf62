// knn_inst.hip -- instantiates knn_mfma_kernel<KS, MCE_KCAP> for KS = 1..16, 20, 24, 28, 32 and knn_f16_kernel<KST, MCE_KCAP, ..> for KST = 1..4.
// Compile with -DMCE_KCAP=<4|8|12|16|24|32>.
#include "knn_mfma.hpp"
#include "knn_f16.hpp"
#include "knn_panel.hpp"
#include "knn_deep.hpp"
#include "knn_dispatch.hpp"

#ifndef MCE_KCAP
#error "compile with -DMCE_KCAP=n"
#endif
// MCE_INST_PART: 1 = the fp16 filter kernels only, 2 = the fp64 sweep kernels only, 3 = the panel sweep kernels only (the
// Makefile builds them as separate objects: they are compiled with different instruction schedulers), 0 = all
#ifndef MCE_INST_PART
#define MCE_INST_PART 0
#endif
#define MCE_INST_F16 (MCE_KCAP <= 16 && (MCE_INST_PART == 0 || MCE_INST_PART == 1))
#define MCE_INST_F64 (MCE_INST_PART == 0 || MCE_INST_PART == 2)
#define MCE_INST_PANEL (MCE_KCAP <= 16 && (MCE_INST_PART == 0 || MCE_INST_PART == 3))
#define MCE_INST_DEEP (MCE_KCAP <= 16 && (MCE_INST_PART == 0 || MCE_INST_PART == 4))     // knn_deep.hpp: 64 <= d <= 127 on the fp16 filter

namespace mce {

#if MCE_INST_F64
template <int KS, int KCAP>
hipError_t launch_variant(const KnnArgs& a, hipStream_t st)
{
    constexpr size_t LDS = lds_bytes(KS, KCAP);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static bool attr_set[kMaxDevices] = {};   // per device; benign race: idempotent
    auto kern = knn_mfma_kernel<KS, KCAP>;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= kMaxDevices || !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) return e;
        if (dev < kMaxDevices) attr_set[dev] = true;
    }
    const dim3 grid((unsigned)(a.nqblk * a.rsplit));
    hipLaunchKernelGGL(kern, grid, dim3(kThreads), LDS, st, a.Yf, a.nchunk_total, a.rsplit, a.X, a.center, a.nq, a.D,
                       a.nq_pad, a.nqblk, a.self_exclude, a.self_offset, a.ksel, a.part_d, a.part_i);
    return hipGetLastError();
}
#endif

#if MCE_INST_F16
template <int KST, int KCAP, bool PRUNE, bool LOWER = false, int SYM = 0, int LC = KCAP, int QTT = kHQT>
hipError_t launch_f16_variant(const KnnF16Args& a, hipStream_t st)
{
    constexpr size_t LDS_MAX = PRUNE ? f16_prune_lds_bytes(KST, 16 * KST - 1, LC) : f16_lds_bytes(KST, KCAP, SYM >= 2, QTT);
    static_assert(LDS_MAX <= 160 * 1024, "LDS budget");
    const size_t LDS = PRUNE ? f16_prune_lds_bytes(KST, a.D, LC) : LDS_MAX;     // pruned walk: sized by the dimension (more waves per CU)
    static bool attr_set[kMaxDevices] = {};
    auto kern = knn_f16_kernel<KST, KCAP, PRUNE, LOWER, SYM, LC, QTT>;
    // (QTT = 4, the wide exhaustive sweep: a workgroup serves QTT / kHQT of the plan's query blocks -- the plan keeps their number a
    //  multiple of that)
    constexpr int WB = QTT / kHQT;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= kMaxDevices || !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
        if (e != hipSuccess) return e;
        if (dev < kMaxDevices) attr_set[dev] = true;
    }
    // pruned walk: one 64-thread workgroup per wave of a query block
    // symmetric sweep: one workgroup per unit (sym_types.hpp); its prepass and repair launches: one per query block
    const int ntiles_even = (int)((a.nr + 31) / 32) + (int)(((a.nr + 31) / 32) & 1);
    const int hv_n = PRUNE ? (a.seed_cfg & 0xffffff) : 0, hv_S = PRUNE && ((a.seed_cfg >> 24) & 0x7f) > 1 ? ((a.seed_cfg >> 24) & 0x7f) : 1;
    // (pruned walk: nqblk_run counts WAVES of the dispatch order -- 0: all nqblk * 8 of them)
    const dim3 grid((unsigned)(PRUNE ? (a.nqblk_run ? a.nqblk_run : a.nqblk * kHWaves) + hv_n * (hv_S - 1)
                               : SYM == 2 ? sym_unit_count(a.nqblk, kHWaves * kHQT, a.sym.panel * f16_chunk_tiles(KST), ntiles_even)
                               : SYM == 1 ? (a.nqblk_run ? a.nqblk_run : a.nqblk) : (a.nqblk / WB) * a.rsplit));
    hipLaunchKernelGGL(kern, grid, dim3(PRUNE ? 64 : kHThreads), LDS, st, static_cast<const _Float16*>(a.Yh), a.nchunk_total, a.rsplit,
                       static_cast<const _Float16*>(a.Xh), a.qinfo, a.params, a.X, a.Y, a.nq, a.nr, a.D, a.nq_pad, a.nqblk / WB,
                       a.self_exclude, a.self_offset, a.ksel, a.part_d, a.part_i, a.clist, a.cdist, a.list_len, a.rperm, a.qperm, a.tbox_r, a.tbox_q, a.cbox_r, a.qblk0, a.qblk_stride, a.border, a.lo_d, a.lo_i, a.seed_cfg, a.sym, a.wg_us);
    return hipGetLastError();
}
#endif

#if MCE_INST_F16 || MCE_INST_PANEL
template <int KST, int KCAP, bool LOWER = false>
hipError_t launch_panel_variant(const PanelArgs& a, hipStream_t st)
#if MCE_INST_PANEL
{
    constexpr size_t LDS = panel_lds_bytes(KST);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static bool attr_set[kMaxDevices] = {};
    auto kern = knn_panel_kernel<KST, KCAP, LOWER>;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= kMaxDevices || !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) return e;
        if (dev < kMaxDevices) attr_set[dev] = true;
    }
    const int units = panel_unit_count(a.geom);
    if (units <= 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3((unsigned)units), dim3(kHThreads), LDS, st, a);
    return hipGetLastError();
}
template hipError_t launch_panel_variant<1, MCE_KCAP>(const PanelArgs&, hipStream_t);
template hipError_t launch_panel_variant<2, MCE_KCAP>(const PanelArgs&, hipStream_t);
template hipError_t launch_panel_variant<3, MCE_KCAP>(const PanelArgs&, hipStream_t);
template hipError_t launch_panel_variant<4, MCE_KCAP>(const PanelArgs&, hipStream_t);
#if MCE_KCAP == 16
template hipError_t launch_panel_variant<1, MCE_KCAP, true>(const PanelArgs&, hipStream_t);
template hipError_t launch_panel_variant<2, MCE_KCAP, true>(const PanelArgs&, hipStream_t);
template hipError_t launch_panel_variant<3, MCE_KCAP, true>(const PanelArgs&, hipStream_t);
template hipError_t launch_panel_variant<4, MCE_KCAP, true>(const PanelArgs&, hipStream_t);
#endif
#else
;     // defined in this list capacity's panel object (MCE_INST_PART = 3)
extern template hipError_t launch_panel_variant<1, MCE_KCAP>(const PanelArgs&, hipStream_t);
extern template hipError_t launch_panel_variant<2, MCE_KCAP>(const PanelArgs&, hipStream_t);
extern template hipError_t launch_panel_variant<3, MCE_KCAP>(const PanelArgs&, hipStream_t);
extern template hipError_t launch_panel_variant<4, MCE_KCAP>(const PanelArgs&, hipStream_t);
#if MCE_KCAP == 16
extern template hipError_t launch_panel_variant<1, MCE_KCAP, true>(const PanelArgs&, hipStream_t);
extern template hipError_t launch_panel_variant<2, MCE_KCAP, true>(const PanelArgs&, hipStream_t);
extern template hipError_t launch_panel_variant<3, MCE_KCAP, true>(const PanelArgs&, hipStream_t);
extern template hipError_t launch_panel_variant<4, MCE_KCAP, true>(const PanelArgs&, hipStream_t);
#endif
#endif
#endif

#if MCE_INST_DEEP
template <int KST, int KCAP, bool LOWER = false>
hipError_t launch_deep_variant(const DeepArgs& a, hipStream_t st)
{
    constexpr size_t LDS = deep_lds_bytes(KST);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static bool attr_set[kMaxDevices] = {};
    auto kern = knn_deep_kernel<KST, KCAP, LOWER>;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= kMaxDevices || !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) return e;
        if (dev < kMaxDevices) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.nqblk * a.rsplit)), dim3(kHThreads), LDS, st, a);
    return hipGetLastError();
}
#endif

#define MCE_STR2(x) #x
#define MCE_STR(x) MCE_STR2(x)
#define MCE_VARIANT(KS)                                                                                  \
    {&launch_variant<KS, MCE_KCAP>, KS, MCE_KCAP, mfma_qt(KS), chunk_tiles(KS, MCE_KCAP), lds_bytes(KS, MCE_KCAP),  \
     "knn_mfma_kernel<KS=" #KS ",KCAP=" MCE_STR(MCE_KCAP) ">"}

#define MCE_CAT2(a, b) a##b
#define MCE_CAT(a, b) MCE_CAT2(a, b)

// host-only table (a namespace-scope const would otherwise be emitted for the device too)
#if !defined(__HIP_DEVICE_COMPILE__)
#if MCE_INST_F64
extern const KnnVariant MCE_CAT(g_knn_kcap, MCE_KCAP)[kMaxKS] = {
    MCE_VARIANT(1),  MCE_VARIANT(2),  MCE_VARIANT(3),  MCE_VARIANT(4),  MCE_VARIANT(5),  MCE_VARIANT(6),
    MCE_VARIANT(7),  MCE_VARIANT(8),  MCE_VARIANT(9),  MCE_VARIANT(10), MCE_VARIANT(11), MCE_VARIANT(12),
    MCE_VARIANT(13), MCE_VARIANT(14), MCE_VARIANT(15), MCE_VARIANT(16),
    MCE_VARIANT(20), MCE_VARIANT(24), MCE_VARIANT(28), MCE_VARIANT(32),        // 64 <= d <= 127: KS rounded up to a multiple of four
};
#endif
#if MCE_INST_F16
#if MCE_KCAP == 16
#define MCE_F16_LOWER(KST) (&launch_f16_variant<KST, MCE_KCAP, false, true>)
// symmetric second pass (16 < K <= 32): the sweep on the LOWER panel kernel, its repair launch, and a prepass that tracks 33 group minima
#define MCE_F16_SYM_LOWER(KST) &launch_panel_variant<KST, MCE_KCAP, true>, &launch_f16_variant<KST, MCE_KCAP, false, true, 3>, &launch_f16_variant<KST, 32, false, false, 1>
#else
#define MCE_F16_LOWER(KST) nullptr
#define MCE_F16_SYM_LOWER(KST) nullptr, nullptr, nullptr
#endif
#if MCE_KCAP == 12
#define MCE_F16_SHORT(KST) ((KST) == 1 ? &launch_f16_variant<1, MCE_KCAP, true, false, 0, 9> : (knn_f16_launch_fn) nullptr), ((KST) == 1 ? 9 : 0), \
                           ((KST) == 1 ? &launch_f16_variant<1, MCE_KCAP, true, false, 0, 10> : (knn_f16_launch_fn) nullptr), ((KST) == 1 ? 10 : 0)
#else
#define MCE_F16_SHORT(KST) nullptr, 0, nullptr, 0
#endif
#if MCE_KCAP == 4
#define MCE_F16_WIDE(KST) ((KST) == 1 ? &launch_f16_variant<1, MCE_KCAP, false, false, 0, MCE_KCAP, 4> : (knn_f16_launch_fn) nullptr), ((KST) == 1 ? f16_lds_bytes(1, MCE_KCAP, false, 4) : 0)
#else
#define MCE_F16_WIDE(KST) nullptr, 0
#endif
#define MCE_F16_VARIANT(KST, PRUNE_FN)                                                                   \
    {&launch_f16_variant<KST, MCE_KCAP, false>, PRUNE_FN, MCE_F16_LOWER(KST), &launch_f16_variant<KST, MCE_KCAP, false, false, 1>, \
     &launch_f16_variant<KST, MCE_KCAP, false, false, 2>, &launch_f16_variant<KST, MCE_KCAP, false, false, 3>, &launch_panel_variant<KST, MCE_KCAP>, panel_lds_bytes(KST), \
     f16_lds_bytes(KST, MCE_KCAP, true), KST, MCE_KCAP, f16_qt(MCE_KCAP), f16_chunk_tiles(KST), \
     f16_lds_bytes(KST, MCE_KCAP), "knn_f16_kernel<KST=" #KST ",KCAP=" MCE_STR(MCE_KCAP) ">", MCE_F16_SHORT(KST), MCE_F16_WIDE(KST), MCE_F16_SYM_LOWER(KST)}
extern const KnnF16Variant MCE_CAT(g_knn_f16_kcap, MCE_KCAP)[kMaxKST] = {
    MCE_F16_VARIANT(1, (&launch_f16_variant<1, MCE_KCAP, true>)), MCE_F16_VARIANT(2, nullptr), MCE_F16_VARIANT(3, nullptr),
    MCE_F16_VARIANT(4, nullptr),
};
#endif
#if MCE_INST_DEEP
#if MCE_KCAP == 16
#define MCE_DEEP_LOWER(KST) (&launch_deep_variant<KST, MCE_KCAP, true>)          // second pass of a search for 16 < K <= 32 neighbours
#else
#define MCE_DEEP_LOWER(KST) nullptr
#endif
#define MCE_DEEP_VARIANT(KST) {&launch_deep_variant<KST, MCE_KCAP>, KST, MCE_KCAP, deep_chunk_tiles(KST), deep_lds_bytes(KST), "knn_deep_kernel<KST=" #KST ",KCAP=" MCE_STR(MCE_KCAP) ">", MCE_DEEP_LOWER(KST)}
extern const KnnDeepVariant MCE_CAT(g_knn_deep_kcap, MCE_KCAP)[kNumDeepKST] = {MCE_DEEP_VARIANT(5), MCE_DEEP_VARIANT(6), MCE_DEEP_VARIANT(8)};
#endif
#else
// device pass: force the kernel instantiations
#if MCE_INST_DEEP
template __global__ void knn_deep_kernel<5, MCE_KCAP>(DeepArgs);
template __global__ void knn_deep_kernel<6, MCE_KCAP>(DeepArgs);
template __global__ void knn_deep_kernel<8, MCE_KCAP>(DeepArgs);
#if MCE_KCAP == 16
template __global__ void knn_deep_kernel<5, MCE_KCAP, true>(DeepArgs);
template __global__ void knn_deep_kernel<6, MCE_KCAP, true>(DeepArgs);
template __global__ void knn_deep_kernel<8, MCE_KCAP, true>(DeepArgs);
#endif
#endif
#if MCE_INST_F16
#define MCE_F16_INST(KST, PR, LW, SY) template __global__ void knn_f16_kernel<KST, MCE_KCAP, PR, LW, SY>(const _Float16*, int64_t, int, const _Float16*, const double*, const double*, const double*, const double*, int64_t, int64_t, int, int64_t, int, int, int64_t, int, double*, int*, const int*, const float*, int, const int*, const int*, const float*, const float*, const float*, int, int, const int*, const double*, const int*, int, SymParams, float*);
MCE_F16_INST(1, false, false, 0) MCE_F16_INST(2, false, false, 0) MCE_F16_INST(3, false, false, 0) MCE_F16_INST(4, false, false, 0) MCE_F16_INST(1, true, false, 0)
MCE_F16_INST(1, false, false, 1) MCE_F16_INST(2, false, false, 1) MCE_F16_INST(3, false, false, 1) MCE_F16_INST(4, false, false, 1)
MCE_F16_INST(1, false, false, 2) MCE_F16_INST(2, false, false, 2) MCE_F16_INST(3, false, false, 2) MCE_F16_INST(4, false, false, 2)
MCE_F16_INST(1, false, false, 3) MCE_F16_INST(2, false, false, 3) MCE_F16_INST(3, false, false, 3) MCE_F16_INST(4, false, false, 3)
#if MCE_KCAP == 12
template __global__ void knn_f16_kernel<1, MCE_KCAP, true, false, 0, 9>(const _Float16*, int64_t, int, const _Float16*, const double*, const double*, const double*, const double*, int64_t, int64_t, int, int64_t, int, int, int64_t, int, double*, int*, const int*, const float*, int, const int*, const int*, const float*, const float*, const float*, int, int, const int*, const double*, const int*, int, SymParams, float*);
template __global__ void knn_f16_kernel<1, MCE_KCAP, true, false, 0, 10>(const _Float16*, int64_t, int, const _Float16*, const double*, const double*, const double*, const double*, int64_t, int64_t, int, int64_t, int, int, int64_t, int, double*, int*, const int*, const float*, int, const int*, const int*, const float*, const float*, const float*, int, int, const int*, const double*, const int*, int, SymParams, float*);
#endif
#if MCE_KCAP == 16
MCE_F16_INST(1, false, true, 0) MCE_F16_INST(2, false, true, 0) MCE_F16_INST(3, false, true, 0) MCE_F16_INST(4, false, true, 0)
MCE_F16_INST(1, false, true, 3) MCE_F16_INST(2, false, true, 3) MCE_F16_INST(3, false, true, 3) MCE_F16_INST(4, false, true, 3)
#define MCE_F16_INST32(KST) template __global__ void knn_f16_kernel<KST, 32, false, false, 1>(const _Float16*, int64_t, int, const _Float16*, const double*, const double*, const double*, const double*, int64_t, int64_t, int, int64_t, int, int, int64_t, int, double*, int*, const int*, const float*, int, const int*, const int*, const float*, const float*, const float*, int, int, const int*, const double*, const int*, int, SymParams, float*);
MCE_F16_INST32(1) MCE_F16_INST32(2) MCE_F16_INST32(3) MCE_F16_INST32(4)
#endif
#if MCE_KCAP == 4
template __global__ void knn_f16_kernel<1, MCE_KCAP, false, false, 0, MCE_KCAP, 4>(const _Float16*, int64_t, int, const _Float16*, const double*, const double*, const double*, const double*, int64_t, int64_t, int, int64_t, int, int, int64_t, int, double*, int*, const int*, const float*, int, const int*, const int*, const float*, const float*, const float*, int, int, const int*, const double*, const int*, int, SymParams, float*);
#endif
#endif
#if MCE_INST_PANEL
template __global__ void knn_panel_kernel<1, MCE_KCAP>(PanelArgs);
template __global__ void knn_panel_kernel<2, MCE_KCAP>(PanelArgs);
template __global__ void knn_panel_kernel<3, MCE_KCAP>(PanelArgs);
template __global__ void knn_panel_kernel<4, MCE_KCAP>(PanelArgs);
#if MCE_KCAP == 16
template __global__ void knn_panel_kernel<1, MCE_KCAP, true>(PanelArgs);
template __global__ void knn_panel_kernel<2, MCE_KCAP, true>(PanelArgs);
template __global__ void knn_panel_kernel<3, MCE_KCAP, true>(PanelArgs);
template __global__ void knn_panel_kernel<4, MCE_KCAP, true>(PanelArgs);
#endif
#endif
#if MCE_INST_F64
#define MCE_INST(KS) template __global__ void knn_mfma_kernel<KS, MCE_KCAP>(const double*, int64_t, int, const double*, const double*, int64_t, int, int64_t, int, int, int64_t, int, double*, int*);
MCE_INST(1) MCE_INST(2) MCE_INST(3) MCE_INST(4) MCE_INST(5) MCE_INST(6) MCE_INST(7) MCE_INST(8)
MCE_INST(9) MCE_INST(10) MCE_INST(11) MCE_INST(12) MCE_INST(13) MCE_INST(14) MCE_INST(15) MCE_INST(16)
MCE_INST(20) MCE_INST(24) MCE_INST(28) MCE_INST(32)
#endif
#endif

}  // namespace mce
