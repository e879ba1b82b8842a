// Lab equipment (tools/_abl/lib/probes.so): which XCD does workgroup b of a launch land on?  The grouped weight gradient, the panel
// GEMMs and K2 place workgroups that share operand slabs on "b % 8" (speed only).  This probe records HW_REG_XCC_ID, the CU id and
// the start time of every workgroup for a launch of the given shape (threads, dynamic LDS), so that the assumption can be checked
// for one-workgroup-per-CU (128 KiB LDS) launches and for 2-per-CU ones.  (round 6)
#include <hip/hip_runtime.h>

__global__ void xcc_map_kernel(unsigned* out, unsigned long long hold_ticks) {
    extern __shared__ char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    ((volatile char*)lds)[threadIdx.x & 63] = 1;
    while (__builtin_amdgcn_s_memrealtime() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(16);       // hold the CU: placement of a full round
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x] = xcc;
        out[3 * blockIdx.x + 1] = hwid;
        out[3 * blockIdx.x + 2] = (unsigned)t0;
    }
}

extern "C" int murcl_debug_xcc_map(int grid, int threads, int lds_bytes, double hold_us, unsigned* out, hipStream_t stream) {
    static bool once = false;
    if (!once) { (void)hipFuncSetAttribute((const void*)xcc_map_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
    hipLaunchKernelGGL(xcc_map_kernel, dim3(grid), dim3(threads), lds_bytes, stream, out, (unsigned long long)(hold_us * 100.0));
    return (int)hipGetLastError();
}
