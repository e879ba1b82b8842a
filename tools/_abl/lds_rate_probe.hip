// Calibration probe (dev tool, not on the product path): how many bytes per clock a CU's LDS delivers to NW waves that issue
// nothing but fragment reads - ds_read_b128 (the A / B fragments of an NT product), ds_read_b64 and ds_read_b64_tr_b16 (the
// transposing read both operands of a TN product - the weight gradients - need).  Rows at the K2 stride (1024 + 16 bytes):
// conflict-free for all three.  One workgroup per CU; every wave reads `iters` x 16 fragments; cycles by s_memtime.
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void lds_rate_kernel(int iters, unsigned long long* cycles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32 * 1040 / 4; i += blockDim.x) ((float*)smem)[i] = (float)i;
    __syncthreads();
    const int r16 = lane & 15, q4 = lane >> 4;
    // b128: row r16 (+16 per fragment pair), 16-byte chunk q4 + 4u of the row; b64 / tr: row 4 q4 + (r16 >> 2), 8-byte piece
    const char* p128 = smem + r16 * 1040 + q4 * 16;
    constexpr int ST64 = (MODE >= 4) ? 1056 : 1040;                       // MODE 4 / 5: rows 32 B apart in the bank image (16 rows = two full 256-byte lines)
    const char* p64 = smem + (4 * q4 + (r16 >> 2)) * ST64 + (r16 & 3) * 8;
    float acc = 0.f;
    const unsigned a128 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)(p128);
    const unsigned a64 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)(p64);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // sixteen reads issued back to back (inline asm: the compiler neither merges nor serialises them), one wait for the batch
        f32x4 v[8];
        u32x2 w[16];
        const unsigned o = (it & 1) ? 512u : 0u;
        if (MODE == 0 || MODE == 3) {
        const unsigned ab = MODE == 0 ? a128 : a128 - q4 * 16 + q4 * 256;      // MODE 3: the panel kernels' k assignment (chunk kk + 16 q4)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(ab + (MODE == 0 ? o + u * 64 : (o >> 2) + u * 16) + h * 16 * 1040));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u][0];
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (MODE == 1 || MODE == 4) asm volatile("ds_read_b64 %0, %1" : "=v"(w[u]) : "v"(a64 + (o >> 1) + (u & 7) * 32 + (u >> 3) * 16 * ST64));
                else asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(w[u]) : "v"(a64 + (o >> 1) + (u & 7) * 32 + (u >> 3) * 16 * ST64));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += (float)w[u][0];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
    if (acc == 123.456f) sink[0] = acc;
}

// mode 0: ds_read_b128, 1: ds_read_b64, 2: ds_read_b64_tr_b16; waves = 4 or 8 per workgroup; cycles [grid][8]
extern "C" int murcl_debug_lds_rate(int mode, int waves, int grid, int iters, unsigned long long* cycles, float* sink, hipStream_t s) {
    const dim3 g(grid), b(64 * waves);
    const int lds = 48 * 1056;
    if (mode == 0) hipLaunchKernelGGL(lds_rate_kernel<0>, g, b, lds, s, iters, cycles, sink);
    else if (mode == 3) hipLaunchKernelGGL(lds_rate_kernel<3>, g, b, lds, s, iters, cycles, sink);
    else if (mode == 4) hipLaunchKernelGGL(lds_rate_kernel<4>, g, b, lds, s, iters, cycles, sink);
    else if (mode == 5) hipLaunchKernelGGL(lds_rate_kernel<5>, g, b, lds, s, iters, cycles, sink);
    else if (mode == 1) hipLaunchKernelGGL(lds_rate_kernel<1>, g, b, lds, s, iters, cycles, sink);
    else hipLaunchKernelGGL(lds_rate_kernel<2>, g, b, lds, s, iters, cycles, sink);
    return (int)hipGetLastError();
}
