// mfma_bf16_clock.hip -- does a bf16 MFMA stream (8-bit mantissa multipliers) sustain a higher clock than the fp16
// one on this power-limited part?  Same harness as mfma_f16_clock.hip; random operands; with and without the
// filter's gate next to it (8 v_min3_f32 + compare per 2 MFMAs).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_clock.hip -o tools/mfma_bf16_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef __bf16 v8b __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float min3f(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// DT 0: f16, 1: bf16.  GATE 0: MFMA only, 1: + min3 gate per accumulator tile (as the filter's sweep: 2 MFMAs per tile)
template <int DT, int GATE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, const v4u* ab, float g)
{
    v16f acc[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    v4u ar0 = ab[threadIdx.x % 64], ar1 = ab[64 + threadIdx.x % 64], br0 = ab[128 + threadIdx.x % 64], br1 = ab[192 + threadIdx.x % 64];
    int hits = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("" : "+v"(ar0), "+v"(ar1), "+v"(br0), "+v"(br1));      // opaque: the products are not loop-invariant
            v16f z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (DT == 0) {
                v8h a0, a1, b0, b1;
                __builtin_memcpy(&a0, &ar0, 16); __builtin_memcpy(&a1, &ar1, 16); __builtin_memcpy(&b0, &br0, 16); __builtin_memcpy(&b1, &br1, 16);
                z = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, z, 0, 0, 0);
            } else {
                v8b a0, a1, b0, b1;
                __builtin_memcpy(&a0, &ar0, 16); __builtin_memcpy(&a1, &ar1, 16); __builtin_memcpy(&b0, &br0, 16); __builtin_memcpy(&b1, &br1, 16);
                z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, z, 0, 0, 0);
            }
            if (GATE) {
                // gate of the PREVIOUS tile in this slot (software pipelining as in the kernel)
                const v16f& c = acc[i];
                float m0 = min3f(c[0], c[1], c[2]), m1 = min3f(c[3], c[4], c[5]), m2 = min3f(c[6], c[7], c[8]);
                float m3 = min3f(c[9], c[10], c[11]), m4 = min3f(c[12], c[13], c[14]);
                m0 = min3f(m0, m1, m2);
                m3 = min3f(m3, m4, c[15]);
                hits += min3f(m0, m3, m3) <= g ? 1 : 0;
            }
            acc[i] = z;
        }
    }
    float s = (float)hits;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }
static unsigned short f2b(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x8000u) >> 16); }

template <int DT, int GATE>
void run(const char* name, int blocks, int iters)
{
    float* out; v4u* ab;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&ab, 16 * 256);
    unsigned short h[256 * 8];
    srand(1);
    for (int i = 0; i < 256 * 8; ++i) { const float v = (rand() % 2001 - 1000) / 250.0f; h[i] = DT == 0 ? f2h(v) : f2b(v); }
    hipMemcpy(ab, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<DT, GATE><<<blocks, 256>>>(out, iters / 10, ab, -1e30f);
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<DT, GATE><<<blocks, 256>>>(out, iters, ab, -1e30f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    const double nmfma = (double)blocks * 4 * iters * 8;
    printf("%-40s best %8.3f ms  mean %8.3f ms  %.1f TFLOP/s\n", name, best, sum / 3, nmfma * 32768.0 / (best * 1e-3) / 1e12);
    hipFree(out); hipFree(ab);
}

int main()
{
    run<0, 0>("f16  MFMA only, 2 waves/SIMD", 512, 100000);
    run<1, 0>("bf16 MFMA only, 2 waves/SIMD", 512, 100000);
    run<0, 1>("f16  MFMA + min3 gate, 2 waves/SIMD", 512, 100000);
    run<1, 1>("bf16 MFMA + min3 gate, 2 waves/SIMD", 512, 100000);
    run<0, 1>("f16  MFMA + min3 gate again", 512, 100000);
    return 0;
}
