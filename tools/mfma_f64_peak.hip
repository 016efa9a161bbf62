// mfma_f64_peak.hip -- microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950,
// alone and with f64 VALU work interleaved (does the compare/insert VALU stream steal
// from the DGEMM pipe?).  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o tools/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC, int VALU>
__global__ __launch_bounds__(512, 2) void k(double* out, int iters, double a0, double b0)
{
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    double v0 = a0, v1 = b0, v2 = a0 + 1, v3 = b0 + 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            if (VALU >= 1) { v0 = fma(v0, 1.0000001, 1e-9); v1 = fma(v1, 0.9999999, 1e-9); }
            if (VALU >= 2) { v2 = fma(v2, 1.0000001, 1e-9); v3 = fma(v3, 0.9999999, 1e-9); }
        }
    }
    double s = v0 + v1 + v2 + v3;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int VALU>
void run(const char* name, int blocks, int threads)
{
    double* out;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, VALU><<<blocks, threads>>>(out, 10, 1.0, 2.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, VALU><<<blocks, threads>>>(out, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double nmfma = (double)blocks * (threads / 64) * iters * NACC;
    const double tf = nmfma * 2048.0 / (ms * 1e-3) / 1e12;
    printf("%-28s blocks=%d thr=%d  %.3f ms  %.2f TFLOP/s fp64 (MFMA flops only)\n", name, blocks, threads, ms, tf);
    hipFree(out);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device: %s CUs=%d clock=%d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    run<4, 0>("mfma x4acc, 1 wave/SIMD", 256, 256);
    run<4, 0>("mfma x4acc, 2 waves/SIMD", 256, 512);
    run<2, 0>("mfma x2acc, 2 waves/SIMD", 256, 512);
    run<1, 0>("mfma x1acc, 2 waves/SIMD", 256, 512);
    run<4, 1>("mfma x4acc + 2 fma/mfma", 256, 512);
    run<4, 2>("mfma x4acc + 4 fma/mfma", 256, 512);
    run<4, 0>("mfma x4acc, 2 blocks/CU", 512, 512);
    return 0;
}
