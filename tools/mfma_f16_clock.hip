// mfma_f16_clock.hip -- microbenchmark: what shader clock does an MI355X sustain under a dense
// v_mfma_f32_32x32x16_f16 stream, with zero and with random operands, and what does s_memtime count?
// (The fp16 filter's sweep sits at ~1.17 PFLOP/s; is that the matrix pipe at the clock the part actually runs?)
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f16_clock.hip -o tools/mfma_f16_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// MODE 0: dependent VALU chain only (light load); 1: MFMA stream; 2: MFMA + 4 VALU per MFMA
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, long long* clk, int iters, const v8h* ab)
{
    v16f acc[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const v8h a = ab[threadIdx.x % 64], b = ab[64 + threadIdx.x % 64];
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) v0 = __builtin_fmaf(v0, 1.0000001f, 1e-9f);      // dependent: 1 issue per op
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
                if (MODE == 2) {
                    v0 = __builtin_fmaf(v0, 1.0000001f, 1e-9f); v1 = __builtin_fmaf(v1, 0.9999999f, 1e-9f);
                    v2 = __builtin_fmaf(v2, 1.0000001f, 1e-9f); v3 = __builtin_fmaf(v3, 0.9999999f, 1e-9f);
                }
            }
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = v0 + v1 + v2 + v3;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE>
void run(const char* name, int blocks, int iters, bool random_data)
{
    float* out; long long* clk; v8h* ab;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&clk, sizeof(long long) * 2 * blocks);
    hipMalloc(&ab, sizeof(v8h) * 128);
    _Float16 h[128 * 8];
    srand(1);
    for (int i = 0; i < 128 * 8; ++i) h[i] = random_data ? (_Float16)((rand() % 2001 - 1000) / 500.0f) : (_Float16)0.f;
    hipMemcpy(ab, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, clk, iters / 10, ab);       // warm the clocks
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, clk, iters, ab);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long hc[2];
    hipMemcpy(hc, clk, sizeof(hc), hipMemcpyDeviceToHost);
    const double nmfma = MODE ? (double)blocks * 4 * iters * 4 : 0.0;
    printf("%-34s %s  %8.3f ms  s_memtime %.4g ticks (%.1f MHz)  s_memrealtime %.4g ticks (%.1f MHz)", name, random_data ? "random" : "zeros ",
           ms, (double)hc[0], hc[0] / (ms * 1e3), (double)hc[1], hc[1] / (ms * 1e3));
    if (MODE) {
        const double per_simd = (double)iters * 4 * (blocks / 256);              // MFMAs issued per SIMD (1 wave each per block)
        printf("  %.1f TFLOP/s  -> %.0f MHz if 8 passes x 4 clk per MFMA", nmfma * 32768.0 / (ms * 1e-3) / 1e12, per_simd * 32.0 / (ms * 1e3));
    } else {
        printf("  dependent fma chain: %.2f ns per op", ms * 1e6 / ((double)iters * 32));
    }
    printf("\n");
    hipFree(out); hipFree(clk); hipFree(ab);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device: %s CUs=%d clockRate=%d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    run<0>("VALU chain, 1 wave/SIMD", 256, 400000, false);
    run<1>("MFMA, 1 wave/SIMD", 256, 400000, false);
    run<1>("MFMA, 1 wave/SIMD", 256, 400000, true);
    run<1>("MFMA, 2 waves/SIMD", 512, 200000, false);
    run<1>("MFMA, 2 waves/SIMD", 512, 200000, true);
    run<2>("MFMA + 4 VALU, 2 waves/SIMD", 512, 200000, true);
    run<0>("VALU chain again", 256, 400000, false);
    return 0;
}
