// clock_probe.hip -- in-kernel shader clock: delta s_memtime (shader cycles) / delta s_memrealtime (100 MHz) around a
// dependent v_fma chain, one wavefront per CU-ish block (MI355X_MICROARCH.md "DVFS give-back" item 6).  Diagnostic only.
#include <hip/hip_runtime.h>
extern "C" __global__ void clock_probe_kernel(unsigned long long* out, int iters)
{
    unsigned long long t0, r0, t1, r1;
    float acc = threadIdx.x * 1.0e-3f;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int i = 0; i < iters; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 1.0e-7f);
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) : "v"(acc) : "memory");
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = t1 - t0;
        out[blockIdx.x * 4 + 1] = r1 - r0;
        out[blockIdx.x * 4 + 2] = r0;
        out[blockIdx.x * 4 + 3] = r1;
    }
    if (acc == 12345.0f) out[0] = 0;
}
extern "C" int clock_probe(unsigned long long* out_dev, int blocks, int iters, void* stream)
{
    hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(64), 0, static_cast<hipStream_t>(stream), out_dev, iters);
    return static_cast<int>(hipGetLastError());
}
