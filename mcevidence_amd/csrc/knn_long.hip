// knn_long.hip -- instantiations of knn_long_kernel<KCAP> (knn_long.hpp: the fp64 MFMA sweep for 128 <= d <= 1024) and their launcher.
#include "knn_long.hpp"

#include "knn_dispatch.hpp"

namespace mce {

template <int KCAP>
static hipError_t launch_long_variant(const LongArgs& a, hipStream_t st)
{
    constexpr size_t LDS = long_lds_bytes(KCAP);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static bool attr_set[kMaxDevices] = {};
    auto kern = knn_long_kernel<KCAP>;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= kMaxDevices || !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) return e;
        if (dev < kMaxDevices) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.nqblk * a.rsplit)), dim3(kThreads), LDS, st, a);
    return hipGetLastError();
}

// host-only table (a namespace-scope const would otherwise be emitted for the device too)
#if !defined(__HIP_DEVICE_COMPILE__)
extern const KnnLongVariant g_knn_long[kNumLongKcap] = {
    {&launch_long_variant<8>, 8, long_ct(8), long_lds_bytes(8), "knn_long_kernel<KCAP=8>"},
    {&launch_long_variant<16>, 16, long_ct(16), long_lds_bytes(16), "knn_long_kernel<KCAP=16>"},
    {&launch_long_variant<32>, 32, long_ct(32), long_lds_bytes(32), "knn_long_kernel<KCAP=32>"},
};
#else
// device pass: force the kernel instantiations
template __global__ void knn_long_kernel<8>(LongArgs);
template __global__ void knn_long_kernel<16>(LongArgs);
template __global__ void knn_long_kernel<32>(LongArgs);
#endif

}  // namespace mce
