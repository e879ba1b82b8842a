// Calibration probe (dev tool, not on the product path): the LDS-DMA ring skeleton of the streaming kernels with
// no compute, to measure what HBM read rate a ring configuration can sustain.  Each workgroup streams a
// contiguous slab of `bytes_per_wg` through NSLOT slots of TILE bytes with NSLOT-1 tiles in flight.
#include "common.h"

template <int NW, int TILE, int NSLOT>
__global__ __launch_bounds__(64 * NW) void stream_probe_kernel(const char* __restrict__ src, float* __restrict__ out,
                                                                long bytes_per_wg, int lds_reads, int sleep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int GT = TILE / (NW * 1024);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = lds_off(smem);
    const char* base = src + (size_t)blockIdx.x * bytes_per_wg;
    const int n_tiles = (int)(bytes_per_wg / TILE);
    auto issue = [&](int t) {
#pragma unroll
        for (int j = 0; j < GT; ++j)
            glds16_s(base + (size_t)t * TILE + (j * NW + wave) * 1024, lane * 16, lds0 + (t % NSLOT) * TILE + (j * NW + wave) * 1024);
    };
    for (int t = 0; t < NSLOT - 1 && t < n_tiles; ++t) issue(t);
    float acc = 0.f;
    for (int t = 0; t < n_tiles; ++t) {
        const int after = min(NSLOT - 2, n_tiles - 1 - t);
        // wait until at most after*GT ops are outstanding (literal immediates via a switch)
        switch (after * GT) {
            case 0: WAIT_VMCNT(0); break;
            case 1: WAIT_VMCNT(1); break;
            case 2: WAIT_VMCNT(2); break;
            case 3: WAIT_VMCNT(3); break;
            case 4: WAIT_VMCNT(4); break;
            case 6: WAIT_VMCNT(6); break;
            case 8: WAIT_VMCNT(8); break;
            case 10: WAIT_VMCNT(10); break;
            case 12: WAIT_VMCNT(12); break;
            case 16: WAIT_VMCNT(16); break;
            case 20: WAIT_VMCNT(20); break;
            case 24: WAIT_VMCNT(24); break;
            case 28: WAIT_VMCNT(28); break;
            default: WAIT_VMCNT(0);
        }
        LDS_BARRIER();
        if (t + NSLOT - 1 < n_tiles) issue(t + NSLOT - 1);
        const char* tile = smem + (t % NSLOT) * TILE;
        for (int r = 0; r < lds_reads; ++r) {
            const f32x4 v = *(const f32x4*)(tile + ((r * 64 * NW + threadIdx.x) * 16) % TILE);
            acc += v[0] + v[3];
        }
        if (sleep > 0) for (int k = 0; k < sleep; ++k) __builtin_amdgcn_s_sleep(8);
    }
    if (acc == 123.456f) out[0] = acc;
}

extern "C" int murcl_debug_stream_probe(const void* src, float* out, long total_bytes, int nw, int tile_kb, int nslot,
                                        int wg_per_cu, int lds_reads, int sleep, hipStream_t stream) {
    const int grid = 256 * wg_per_cu;
    const long per = (total_bytes / grid / (tile_kb * 1024)) * (tile_kb * 1024);
#define SP(NW, TK, NS)                                                                                            \
    if (nw == NW && tile_kb == TK && nslot == NS) {                                                               \
        auto k = stream_probe_kernel<NW, TK * 1024, NS>;                                                          \
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, TK * 1024 * NS);          \
        hipLaunchKernelGGL(k, dim3(grid), dim3(64 * NW), TK * 1024 * NS, stream, (const char*)src, out, per,      \
                           lds_reads, sleep);                                                                     \
        return MURCL_CHECK_LAUNCH();                                                                              \
    }
    SP(4, 32, 4) SP(8, 32, 4) SP(4, 16, 4) SP(8, 16, 8) SP(4, 16, 8) SP(8, 32, 3) SP(4, 8, 8) SP(4, 32, 2) SP(8, 16, 4)
    SP(4, 16, 2) SP(4, 16, 3) SP(4, 8, 4) SP(2, 16, 4) SP(2, 8, 4) SP(1, 8, 4) SP(1, 4, 4) SP(1, 4, 8) SP(2, 8, 8)
#undef SP
    return -1;
}
