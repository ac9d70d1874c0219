// occupy.hip -- TEST INFRASTRUCTURE: "somebody else's kernel" on the device.  `blocks` workgroups of 64 threads, each
// holding `lds_bytes` of LDS (so that a CU it sits on has no room for a 160 KiB team-kernel workgroup), spin for
// `microseconds` of the 100 MHz wall clock on a stream of their own.  Used by tests/test_gpu_shared_device.py.
#include <hip/hip_runtime.h>

__global__ void occupy_kernel(long long ticks, unsigned* sink) {
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (sink && lds[threadIdx.x] == 0xFFFFFFFFu) *sink = 1;  // keeps the LDS allocation alive
}

static hipStream_t g_stream = nullptr;

extern "C" int occupy_start(int blocks, int lds_bytes, int microseconds) {
    if (!g_stream && hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return -1;
    if (lds_bytes > 48 * 1024 &&
        hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
        return -2;
    hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(64), (size_t)lds_bytes, g_stream, (long long)microseconds * 100ll, (unsigned*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" int occupy_wait(void) { return g_stream && hipStreamSynchronize(g_stream) == hipSuccess ? 0 : -1; }
// 1 while the occupying kernel is still running
extern "C" int occupy_busy(void) { return g_stream && hipStreamQuery(g_stream) == hipErrorNotReady ? 1 : 0; }
