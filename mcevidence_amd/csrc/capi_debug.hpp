// capi_debug.hpp -- part of capi.hip: test hooks that run tiles through the search kernels' MFMA sequence.
#pragma once

// ---------------------------------------------------------------------------
// Test hook: ONE 32x32 tile of the filter's matrix product, exactly as the search kernels issue it
// (v_mfma_f32_32x32x16_f16, KST chained k-steps, C-in = 0), from caller-made fp16 rows.  The rigorous bound of
// knn_f16.hpp rests on a model of this instruction -- products of two fp16 exact in fp32, accumulation error at most
// 32 KST 2^-24 (|x'| |y'|) -- which tests/test_gpu_parity.py::test_mfma_error_model measures directly.
// ---------------------------------------------------------------------------
namespace mce {
template <int KST>
__global__ __launch_bounds__(64) void mfma_tile_probe_kernel(const _Float16* __restrict__ yp, const _Float16* __restrict__ xp, float* __restrict__ out)
{
    const int lane = threadIdx.x;
    yp += (size_t)blockIdx.x * 32 * 16 * KST;      // one tile per workgroup
    xp += (size_t)blockIdx.x * 32 * 16 * KST;
    out += (size_t)blockIdx.x * 1024;
    v16f acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) {
        // A fragment: lane l <-> reference row l & 31, k = 16 ks + 8 (l >> 5) + e; B fragment: query l & 31, the same k
        const v8h a = *reinterpret_cast<const v8h*>(yp + (lane & 31) * (16 * KST) + 16 * ks + 8 * (lane >> 5));
        const v8h b = *reinterpret_cast<const v8h*>(xp + (lane & 31) * (16 * KST) + 16 * ks + 8 * (lane >> 5));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    // C layout: lane l -> query column l & 31, reference rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
}
}  // namespace mce

extern "C" int mce_debug_mfma_tiles_f16(const uint16_t* yprime, const uint16_t* xprime, int32_t kst, int32_t ntiles, float* out, int32_t device)
{
    if (!yprime || !xprime || !out) return fail(MCE_ERR_INVALID, "null pointer argument");
    if (kst < 1 || kst > 8) return fail(MCE_ERR_INVALID, "kst must be 1..8");
    if (ntiles < 1 || ntiles > (1 << 20)) return fail(MCE_ERR_INVALID, "ntiles must be 1..2^20");
    int rc = select_device(device);
    if (rc != MCE_OK) return rc;
    const size_t nb = (size_t)ntiles * 32 * 16 * kst * sizeof(uint16_t);
    void *dy = nullptr, *dx = nullptr, *dout = nullptr;
    hipError_t e = hipMalloc(&dy, nb);
    if (e == hipSuccess) e = hipMalloc(&dx, nb);
    if (e == hipSuccess) e = hipMalloc(&dout, (size_t)ntiles * 1024 * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(dy, yprime, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dx, xprime, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const _Float16* y = static_cast<const _Float16*>(dy);
        const _Float16* x = static_cast<const _Float16*>(dx);
        float* o = static_cast<float*>(dout);
        const dim3 g((unsigned)ntiles);
        switch (kst) {
            case 1: hipLaunchKernelGGL(mce::mfma_tile_probe_kernel<1>, g, dim3(64), 0, nullptr, y, x, o); break;
            case 2: hipLaunchKernelGGL(mce::mfma_tile_probe_kernel<2>, g, dim3(64), 0, nullptr, y, x, o); break;
            case 3: hipLaunchKernelGGL(mce::mfma_tile_probe_kernel<3>, g, dim3(64), 0, nullptr, y, x, o); break;
            case 4: hipLaunchKernelGGL(mce::mfma_tile_probe_kernel<4>, g, dim3(64), 0, nullptr, y, x, o); break;
            case 5: hipLaunchKernelGGL(mce::mfma_tile_probe_kernel<5>, g, dim3(64), 0, nullptr, y, x, o); break;      // (5, 6, 8: knn_deep.hpp)
            case 6: hipLaunchKernelGGL(mce::mfma_tile_probe_kernel<6>, g, dim3(64), 0, nullptr, y, x, o); break;
            case 7: hipLaunchKernelGGL(mce::mfma_tile_probe_kernel<7>, g, dim3(64), 0, nullptr, y, x, o); break;
            default: hipLaunchKernelGGL(mce::mfma_tile_probe_kernel<8>, g, dim3(64), 0, nullptr, y, x, o); break;
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)ntiles * 1024 * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(dy); (void)hipFree(dx); (void)hipFree(dout);
    if (e != hipSuccess) return fail(MCE_ERR_HIP, "mfma tile probe: %s", hipGetErrorString(e));
    return MCE_OK;
}

extern "C" int mce_debug_mfma_tile_f16(const uint16_t* yprime, const uint16_t* xprime, int32_t kst, float* out, int32_t device)
{
    return mce_debug_mfma_tiles_f16(yprime, xprime, kst, 1, out, device);
}
