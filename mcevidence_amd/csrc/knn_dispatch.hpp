// knn_dispatch.hpp -- host-side table of the knn_mfma_kernel<KS,KCAP,QT> instantiations.
// One translation unit per list capacity KCAP (knn_inst.hip compiled with -DMCE_KCAP=n)
// so the 96 variants build in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sym_types.hpp"

namespace mce {

constexpr int kMaxDevices = 64;   // per-device one-time kernel attributes

struct KnnArgs {
    const double* Yf;       // packed references [nchunk_total*CT][KS][64]
    int64_t nchunk_total;
    int rsplit;
    const double* X;        // queries [nq, D]
    const double* center;   // [64] reference-set column means
    int64_t nq;
    int D;
    int64_t nq_pad;
    int nqblk;
    int self_exclude;
    int64_t self_offset;
    int ksel;               // neighbours actually wanted (<= KCAP)
    double* part_d;         // [rsplit*4][KCAP][nq_pad]
    int* part_i;
};

typedef hipError_t (*knn_launch_fn)(const KnnArgs&, hipStream_t);

struct KnnVariant {
    knn_launch_fn launch;
    int ks, kcap, qt, ct;
    size_t lds_bytes;
    const char* name;
};

// ---- fp16-filter variants (knn_f16.hpp): KST = 1..4 sixteen-wide k-steps, KCAP <= 16 ----
struct KnnF16Args {
    const void* Yh;         // packed fp16 references
    int64_t nchunk_total;
    int rsplit;
    const void* Xh;         // fp16 query rows [nq_pad][16*KST]
    const double* qinfo;    // [nq_pad][2]
    const double* params;   // HP_* scalars
    const double* X;
    const double* Y;
    int64_t nq, nr;
    int D;
    int64_t nq_pad;
    int nqblk;
    int self_exclude;
    int64_t self_offset;
    int ksel;
    double* part_d;
    int* part_i;
    // pruned walk (prune.hpp); all null / 0 for the exhaustive sweep
    const int* clist = nullptr;
    const float* cdist = nullptr;
    int list_len = 0;
    const int* rperm = nullptr;
    const int* qperm = nullptr;
    const float* tbox_r = nullptr;
    const float* tbox_q = nullptr;
    const float* cbox_r = nullptr;
    const double* lo_d = nullptr;  // second pass (K > 16): the first pass's lists; pruned walk: the heavy blocks' side lists (written)
    const int* lo_i = nullptr;
    int seed_cfg = 0;              // exhaustive sweep: seed phase (f16_seed_cfg: chunks | tiles per group << 16), 0 = none;
                                   // pruned walk: nheavy | S << 24 -- the first nheavy waves of the launch's order are served by S workgroups each
    const int* border = nullptr;   // pruned walk: dispatch order of the WAVES (block * 8 + wave), largest box first
    int qblk0 = 0, qblk_stride = 1, nqblk_run = 0;   // pruned walk: nqblk_run waves border[qblk0], border[qblk0 + stride], ... (0: all);
                                                     // symmetric prepass: nqblk_run blocks from qblk0
    SymParams sym;                 // symmetric sweep (launch_sym_pre / launch_sym); rperm = sorted position -> caller's row
    float* wg_us = nullptr;        // diagnostic: per-workgroup duration in microseconds (MCE_PRUNE_TIMES), normally null
};
typedef hipError_t (*knn_f16_launch_fn)(const KnnF16Args&, hipStream_t);
struct PanelArgs;                  // knn_panel.hpp
typedef hipError_t (*knn_panel_launch_fn)(const PanelArgs&, hipStream_t);
struct KnnF16Variant {
    knn_f16_launch_fn launch;
    knn_f16_launch_fn launch_prune;   // PRUNE = true instantiation (KST = 1 only), else null
    knn_f16_launch_fn launch_lower;   // LOWER = true instantiation (KCAP = 16 only), else null
    knn_f16_launch_fn launch_sym_pre; // SYM = 1: prepass of the symmetric sweep
    knn_f16_launch_fn launch_sym;     // SYM = 2: symmetric sweep
    knn_f16_launch_fn launch_sym_repair;   // SYM = 3: exhaustive search of the blocks whose bucket overflowed
    knn_panel_launch_fn launch_panel; // knn_panel_kernel: the symmetric / exhaustive sweep in units (panel x query block)
    size_t lds_bytes_panel;
    size_t lds_bytes_sym;
    int kst, kcap, qt, ct;
    size_t lds_bytes;
    const char* name;
    knn_f16_launch_fn launch_prune_short;   // PRUNE with prune_short_lc (< kcap) list entries in registers: three waves per SIMD (KST = 1, KCAP = 12), else null
    int prune_short_lc;
    knn_f16_launch_fn launch_prune_short2;  // the same with prune_short_lc2 entries (> prune_short_lc), else null
    int prune_short_lc2;
    knn_f16_launch_fn launch_wide;          // the exhaustive sweep with FOUR query tiles per wave (a workgroup = two query blocks; KST = 1, KCAP = 4), else null
    size_t lds_bytes_wide;
    // second pass of a SYMMETRIC search for 16 < K <= 32 neighbours (KCAP = 16 only, else null): the LOWER panel sweep, its repair
    // launch (SYM = 3, LOWER) and a prepass whose seed phase tracks 33 group minima (a bound on the 32nd neighbour)
    knn_panel_launch_fn launch_panel_lower;
    knn_f16_launch_fn launch_sym_repair_lower;
    knn_f16_launch_fn launch_sym_pre32;
};
// ---- deep fp16-filter variants (knn_deep.hpp): 64 <= d <= 127, KST = 5, 6 or 8 sixteen-wide k-steps, K <= 16 ----
struct DeepArgs;                   // knn_deep.hpp
typedef hipError_t (*knn_deep_launch_fn)(const DeepArgs&, hipStream_t);
struct KnnDeepVariant {
    knn_deep_launch_fn launch;
    int kst, kcap, ct;
    size_t lds_bytes;
    const char* name;
    knn_deep_launch_fn launch_lower;   // LOWER = true instantiation (KCAP = 16 only), else null
};
// ---- long-row fp64 variants (knn_long.hpp): 128 <= d <= 1024, K <= 32; lists of 8, 16 or 32 entries ----
struct LongArgs;                   // knn_long.hpp
typedef hipError_t (*knn_long_launch_fn)(const LongArgs&, hipStream_t);
struct KnnLongVariant {
    knn_long_launch_fn launch;
    int kcap, ct;
    size_t lds_bytes;
    const char* name;
};
constexpr int kNumLongKcap = 3;    // index 0, 1, 2 <-> KCAP = 8, 16, 32
extern const KnnLongVariant g_knn_long[kNumLongKcap];
constexpr int kNumDeepKST = 3;     // index 0, 1, 2 <-> KST = 5, 6, 8
extern const KnnDeepVariant g_knn_deep_kcap4[kNumDeepKST];
extern const KnnDeepVariant g_knn_deep_kcap8[kNumDeepKST];
extern const KnnDeepVariant g_knn_deep_kcap12[kNumDeepKST];
extern const KnnDeepVariant g_knn_deep_kcap16[kNumDeepKST];

constexpr int kMaxKST = 4;
extern const KnnF16Variant g_knn_f16_kcap4[kMaxKST];
extern const KnnF16Variant g_knn_f16_kcap8[kMaxKST];
extern const KnnF16Variant g_knn_f16_kcap12[kMaxKST];
extern const KnnF16Variant g_knn_f16_kcap16[kMaxKST];

constexpr int kMaxKS = 20;         // table entries: KS = 1..16 (index KS - 1), then KS = 20, 24, 28, 32 (index 11 + KS / 4)
constexpr int kWideMaxDim = 127;   // d up to which the fp64 sweep serves what the fp16 filter does not (d > MCE_MAX_DIM: KS = 4 ceil((d + 1) / 16))
__host__ __device__ constexpr int mfma_ks_for(int d) { return d <= 63 ? (d + 1 + 3) / 4 : ((d + 1 + 15) / 16) * 4; }
__host__ __device__ constexpr int mfma_variant_index(int KS) { return KS <= 16 ? KS - 1 : 11 + KS / 4; }
constexpr int kNumKcap = 6;
constexpr int kKcapList[kNumKcap] = {4, 8, 12, 16, 24, 32};

// defined in knn_inst.hip (one per KCAP): variants for KS = 1..16, 20, 24, 28, 32 (mfma_variant_index)
extern const KnnVariant g_knn_kcap4[kMaxKS];
extern const KnnVariant g_knn_kcap8[kMaxKS];
extern const KnnVariant g_knn_kcap12[kMaxKS];
extern const KnnVariant g_knn_kcap16[kMaxKS];
extern const KnnVariant g_knn_kcap24[kMaxKS];
extern const KnnVariant g_knn_kcap32[kMaxKS];

}  // namespace mce
