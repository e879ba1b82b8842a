// Calibration probe (dev tool, not on the product path): what rate can a CU pull a SMALL table (the encoder's
// weights: 512 KiB per layer, 1.5 MiB for three - resident in every XCD's 4 MiB L2) into LDS by LDS-DMA, with every CU of the chip
// walking the same table?  MI355X_MICROARCH.md (Indexed rows: gather into LDS) measures 66-73 GB/s per CU for 2,048 rows of
// 1,152 B shared by every workgroup with 4 loader waves and <= 72 KiB in flight; the fused-encoder prototype's weight ring ran at
// ~25 GB/s per CU.  This is the prototype's loader alone: NW waves, slots of 32 whole rows of 1 KiB (one DMA per row), NSLOT slots
// with NSLOT-1 in flight, optional workgroup barrier per slot, optional per-workgroup start rotation, no MFMA and no LDS reads.
#include "common.h"

template <int NW, int NSLOT, bool BARRIER>
__global__ __launch_bounds__(64 * NW) void l2_loader_probe_kernel(const char* __restrict__ W, float* __restrict__ out, int table_rows,
                                                                   int stride, int iters, int rotate) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PER = 32 / NW;                    // rows (= DMA instructions) per wave and slot
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = lds_off(smem);
    const int nslots_table = table_rows / 32;
    const int rot = rotate ? (int)((blockIdx.x >> 3) * rotate) % nslots_table : 0;
    auto issue = [&](int g) {
        const int ts = (g + rot) % nslots_table;
        const char* src = W + (size_t)ts * 32 * 1024;
        const unsigned dst = lds0 + (g % NSLOT) * 32 * stride;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int p = u * NW + wave;
            glds16_u(src + (size_t)p * 1024, lane * 16, dst + p * stride);
        }
    };
    for (int g = 0; g < NSLOT - 1 && g < iters; ++g) issue(g);
    for (int g = 0; g < iters; ++g) {
        if (g + NSLOT - 2 < iters) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 2) * PER) : "memory"); }
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (BARRIER) { LDS_BARRIER(); }
        if (g + NSLOT - 1 < iters) issue(g + NSLOT - 1);
    }
    if (lane == 0 && wave == 0 && *(volatile float*)smem == 123.456f) out[0] = 1.f;
}

extern "C" int murcl_debug_l2_loader_probe(const void* W, float* out, int table_rows, int stride, int iters, int rotate, int nw,
                                           int nslot, int barrier, int grid, hipStream_t stream) {
    if (table_rows % 32 || table_rows <= 0 || stride < 1024 || stride % 16) return -1;
#define LP(NW_, NS_, BAR_)                                                                                              \
    if (nw == NW_ && nslot == NS_ && (barrier != 0) == BAR_) {                                                          \
        auto k = l2_loader_probe_kernel<NW_, NS_, BAR_>;                                                                \
        const int lds = NS_ * 32 * stride;                                                                              \
        if (lds > 163840) return -2;                                                                                    \
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                           \
        hipLaunchKernelGGL(k, dim3(grid), dim3(64 * NW_), lds, stream, (const char*)W, out, table_rows, stride, iters, rotate); \
        return MURCL_CHECK_LAUNCH();                                                                                    \
    }
    LP(4, 3, true) LP(4, 4, true) LP(4, 5, true) LP(4, 3, false) LP(4, 4, false) LP(4, 5, false)
    LP(8, 3, true) LP(8, 4, true) LP(8, 5, true) LP(8, 4, false) LP(8, 5, false) LP(2, 4, false) LP(2, 4, true)
    LP(16, 4, true) LP(16, 4, false) LP(16, 5, false)
#undef LP
    return -1;
}
