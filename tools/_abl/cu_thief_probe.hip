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
