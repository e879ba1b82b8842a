// Dev probe: sustained v_mfma_f32_16x16x32_bf16 rate of the whole chip (the floor under the 137 GFLOP encoder GEMMs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {2, 3, 4, 5, 6, 7, 8, (short)blockIdx.x};
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 4; waves <= 8; waves += 4)
        for (int iters : {2000, 8000, 32000}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                double fl = 256.0 * waves * iters * 4 * 16384.0;
                if (rep) printf("waves/CU %d iters %d: %.3f ms  %.1f TFLOP/s  -> %.0f MHz if 4096 flop/clk/CU\n", waves, iters, ms,
                                fl / ms / 1e9, fl / ms / 1e3 / (256 * 4096.0));
            }
        }
    return 0;
}
