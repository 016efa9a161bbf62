// zero_fill.hpp -- stream-ordered zero fill as a kernel of our own.
//
// Every counter array the search kernels accumulate into is cleared with this, not with hipMemsetAsync: a memset NODE
// of a captured graph was observed (ROCm 7.2, MI355X) to leave its target untouched on a replay that follows an eager
// hipMemsetAsync issued elsewhere by this library -- the symmetric sweep then started from garbage bucket counts and
// wrote gigabytes below its workspace.  A kernel node carries its own arguments and has no such dependence.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mce {

static __global__ __launch_bounds__(256) void zero_words_kernel(uint32_t* __restrict__ p, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0u;
}

// p: 4-byte aligned, bytes: a multiple of 4
inline hipError_t zero_async(void* p, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return hipSuccess;
    const int64_t n = (int64_t)(bytes / 4);
    const unsigned blocks = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(zero_words_kernel, dim3(blocks), dim3(256), 0, st, static_cast<uint32_t*>(p), n);
    return hipGetLastError();
}

}  // namespace mce
