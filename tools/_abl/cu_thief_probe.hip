// Lab equipment (tools/_abl/lib/probes.so, not the product library): a "CU thief" - N workgroups that each hold 64 KiB of LDS and
// spin for a given wall-clock time.  Run on a second stream beside the training step it stands in for RCCL's channel workgroups
// (one per channel, resident for the length of a collective): what does the step lose per occupied CU?  (VERDICT r4 item 3,
// DESIGN section 7; driver: tools/cu_thief.py)
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(64) void cu_thief_kernel(unsigned long long ticks, unsigned* out) {
    extern __shared__ char lds[];                                        // 64 KiB: a CU with a 128-160 KiB tenant cannot take it as well
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
    unsigned spins = 0;
    ((volatile char*)lds)[threadIdx.x] = (char)threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        ++spins;
    }
    if (threadIdx.x == 0 && out) out[blockIdx.x] = spins + ((volatile char*)lds)[1];
}

// n workgroups for `microseconds` (bounded: at most 200 ms whatever is asked)
extern "C" int murcl_debug_cu_thief(int n, int lds_bytes, double microseconds, unsigned* out, hipStream_t stream) {
    if (n <= 0) return 0;
    if (microseconds > 2e5) microseconds = 2e5;
    static bool once = false;
    if (!once) { (void)hipFuncSetAttribute((const void*)cu_thief_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
    hipLaunchKernelGGL(cu_thief_kernel, dim3(n), dim3(64), lds_bytes, stream, (unsigned long long)(microseconds * 100.0), out);
    return (int)hipGetLastError();
}


// Round 6: thieves of other shapes and thieves that ARRIVE in the middle of the step.
//  * `threads` per workgroup (64 .. 512) and `busy`: 0 = sleep (holds LDS / a workgroup slot only), 1 = a dependent FMA chain on every
//    lane (takes issue slots of the SIMDs it sits on, like a collective's reduction loop);
//  * `delay_us`: a one-wave kernel that sleeps for that long runs first ON THE SAME STREAM, so the thieves become runnable that long
//    after the side stream was released (the caller makes it wait for the step's start): they then have to find room beside
//    persistent workgroups that are already running - what a collective launched from a backward hook meets.
__global__ __launch_bounds__(64) void cu_thief_delay_kernel(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
__global__ __launch_bounds__(512) void cu_thief2_kernel(unsigned long long ticks, int busy, unsigned* out) {
    extern __shared__ char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = (float)threadIdx.x;
    unsigned spins = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        if (busy) {
#pragma unroll
            for (int i = 0; i < 64; ++i) acc = acc * 1.0000001f + 0.5f;
        } else {
            __builtin_amdgcn_s_sleep(32);
        }
        ++spins;
    }
    if (threadIdx.x == 0 && out) out[blockIdx.x] = spins + (unsigned)acc;
}
extern "C" int murcl_debug_cu_thief2(int n, int lds_bytes, int threads, int busy, double delay_us, double microseconds, unsigned* out,
                                     hipStream_t stream) {
    if (n <= 0) return 0;
    if (microseconds > 2e5) microseconds = 2e5;
    if (delay_us > 2e5) delay_us = 2e5;
    if (threads < 64 || threads > 512 || (threads & 63)) return -1;
    static bool once = false;
    if (!once) { (void)hipFuncSetAttribute((const void*)cu_thief2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
    if (delay_us > 0) hipLaunchKernelGGL(cu_thief_delay_kernel, dim3(1), dim3(64), 0, stream, (unsigned long long)(delay_us * 100.0));
    hipLaunchKernelGGL(cu_thief2_kernel, dim3(n), dim3(threads), lds_bytes, stream, (unsigned long long)(microseconds * 100.0), busy, out);
    return (int)hipGetLastError();
}
