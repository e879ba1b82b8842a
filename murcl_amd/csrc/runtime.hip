// Process-wide launch policy of the persistent kernels.
//
// The encoder-sized kernels of a step (panel_gemm.hip, gemm.hip's weight gradients, attn_pool*.hip) run ONE round of workgroups with a
// static share of the work each - one workgroup per CU at 128-160 KiB of LDS (K2: two of 65 KiB).  A CU that somebody else holds for
// the length of a launch (RCCL runs one workgroup per channel while a collective is in flight) cannot take its share: that share
// waits for a free CU and runs as a second round, i.e. the launch takes twice as long whether 4 or 32 CUs are missing (measured
// with a co-running "CU thief", tools/cu_thief.py: 1.41 -> 2.15 ms per step for any N in 4..32, profiles/r05_*_cu_thief.txt).
// The CU budget is the number of CUs these launches size their round for: 256 by default; a data-parallel step that overlaps
// collectives with its backward sets 256 - (reserved CUs), so that the channels find free CUs and the round still fits
// (costs budget/256 of the rate of those kernels, always; reference side: nn.DataParallel, train_MuRCL.py:145).
#include "common.h"

static int g_cu_budget = 256;

extern "C" int murcl_cu_budget(void) { return g_cu_budget; }

// -> the budget in force: `cus` rounded down to a multiple of 8 (one step per XCD), clamped to [64, 256]
extern "C" int murcl_set_cu_budget(int cus) {
    if (cus > 256) cus = 256;
    if (cus < 64) cus = 64;
    g_cu_budget = cus & ~7;
    return g_cu_budget;
}

// ---------------------------------------------------------------- box calibration (bench.py's `box` block, round 6)
// Boxes of one MI355X pool differ by a few per cent in the clocks they hold (the same tree read 45.2 k and 42.7 k bags/s on
// two boxes).  bench.py times these two kernels in the same process, before its timed region, so that a reader of the result line
// can tell a slow box from a regression: a plain streaming copy (what the HBM path of THIS box delivers to 16-byte accesses) and a
// register-only bf16 MFMA loop (the matrix clock it sustains under load, on non-trivial operands).  Neither is on the product path.
__global__ __launch_bounds__(256) void calib_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long n16) {
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = (long)blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + u * 256 < n16) v[u] = src[i + u * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + u * 256 < n16) dst[i + u * 256] = v[u];
    }
}
extern "C" int murcl_calib_copy(const void* src, void* dst, long bytes, hipStream_t stream) {
    if (bytes <= 0 || (bytes & 15)) return -1;
    hipLaunchKernelGGL(calib_copy_kernel, dim3(256 * 8), dim3(256), 0, stream, (const u32x4*)src, (u32x4*)dst, bytes / 16);
    return MURCL_CHECK_LAUNCH();
}
// `iters` x 4 independent v_mfma_f32_16x16x32_bf16 per wave, 8 waves per CU: 256 CUs x 8 x iters x 4 x 16384 FLOP per launch
__global__ __launch_bounds__(512) void calib_mfma_kernel(float* __restrict__ out, int iters) {
    // operands that are neither zero nor constant across lanes (the clock a chip holds depends on the data: MI355X_MICROARCH.md, DVFS)
    const unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = (short)(0x3f00 | ((h >> e) & 0xff)) ^ (short)((h >> (8 + e)) << 15);
        b[e] = (short)(0x3e80 | ((h >> (e + 3)) & 0xff)) ^ (short)((h >> (16 + e)) << 15);
    }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
extern "C" int murcl_calib_mfma_bf16(float* out_256x512, int iters, hipStream_t stream) {
    if (iters <= 0) return -1;
    hipLaunchKernelGGL(calib_mfma_kernel, dim3(256), dim3(512), 0, stream, out_256x512, iters);
    return MURCL_CHECK_LAUNCH();
}
