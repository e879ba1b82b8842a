// Prototype (dev tool, NOT on the product path): the "activations in registers, weights streamed through LDS" shape of a
// fused ABMIL encoder forward, built to MEASURE what that shape sustains
// on MI355X before anything is wired to it.  tools/fused_probe.py runs it against torch and times it.
//
//   H_l = relu(H_{l-1} W_l^T + b_l),  l = 1..layers,  H_0 = X [M,512] bf16, W_l [512,512] bf16, f32 accumulate
//
// Structure.  One 4-wave workgroup per CU (one wave per SIMD, up to 512 VGPRs), 128-row tiles, a wave owns 32 rows for
// all layers.  Computing C^T = W . A^T leaves every lane of a 16x16x32 MFMA with 4 consecutive output FEATURES of one row;
// two such blocks (A-operand rows chosen as W rows 32j+8q+r and 32j+8q+4+r) give the lane 8 consecutive features
// 32j+8q .. +7 of its row - exactly the B-operand fragment of k-step j of the NEXT layer, in natural k order.  So a
// layer's output never leaves the registers: IN (128 VGPRs: 32 rows x 512 k) -> 16 VGPRs of accumulators per 32-feature
// slot -> OUT (128 VGPRs) -> IN of the next layer.  The weights stream L2 -> LDS by LDS-DMA in slots of 32 feature rows
// (1056-byte row stride: conflict-free ds_read_b128 for the natural-k fragment, see fe_frag) through an NSLOT ring that
// runs on across layers and tiles; per slot and wave: 32 ds_read_b128, 64 MFMAs, one barrier.
//
// What it cannot avoid: every CU pulls the whole of W (512 KiB) per layer and 128-row tile = 32 B/clk/CU at the full
// MFMA rate (the measured L2 -> LDS gather rate is ~30 B/clk/CU, MI355X_MICROARCH.md), and there is one wave per SIMD, so
// nothing covers a wave's own stalls.
#include "common.h"

#define FE_STRIDE 1056
#define FE_SLOT (32 * FE_STRIDE)
#define FE_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

__device__ __forceinline__ void fe_store16(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 fe_load16(const void* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int NSLOT>
__global__ __launch_bounds__(256, 1) void fused_encoder_probe_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                                       const float* __restrict__ bias, bf16_t* __restrict__ Hout,
                                                                       int M, int layers, int store_all, int no_mfma, int rotate) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int OFF_BIAS = NSLOT * FE_SLOT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, m16 = lane & 15;
    const unsigned lds0 = lds_off(smem);
    const int n_tiles = M / 128;
    const int my_tiles = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;
    const int total = my_tiles * layers * 16;                 // weight slots this workgroup walks

    // biases -> LDS (compiler-visible loads, retired before the ring starts)
    float* lbias = (float*)(smem + OFF_BIAS);
    for (int i = tid; i < layers * 512; i += 256) lbias[i] = bias[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // The 16 slots of a layer are independent (disjoint output features), so every workgroup may walk them from its own start:
    // in lockstep all 32 CUs of an XCD would pull the SAME 33 KiB of W through the same few L2 channels at the same time.
    const int rot = rotate ? (int)(blockIdx.x >> 3) & 15 : 0;
    // slot g = ((tile * layers) + layer) * 16 + jj: LDS row p = 16b + i  <-  W_layer row 32jj + 8(i/4) + 4b + (i%4)
    auto issue = [&](int g) {
        const int jj = ((g & 15) + rot) & 15, layer = (g >> 4) % layers;
        const char* wl = (const char*)(W + (size_t)layer * 512 * 512);
        const unsigned dst = lds0 + (g % NSLOT) * FE_SLOT;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = u * 4 + wave, b = p >> 4, i = p & 15;
            const int wrow = 32 * jj + 8 * (i >> 2) + 4 * b + (i & 3);
            glds16_u(wl + (size_t)wrow * 1024, lane * 16, dst + p * FE_STRIDE);
        }
    };
    for (int g = 0; g < NSLOT - 1 && g < total; ++g) issue(g);

    bf16x8 in_[2][16], out_[2][16];
    int g = 0;
    for (int t = 0; t < my_tiles; ++t) {
        const int row0 = ((int)blockIdx.x + t * (int)gridDim.x) * 128 + 32 * wave;
        // this wave's 32 rows of X as B-operand fragments: lane (q, m16) <- row 16h + m16, k = 32kk + 8q .. +7
        {
            u32x4 raw[2][16];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kk = 0; kk < 16; ++kk)
                    raw[h][kk] = fe_load16((const char*)X + (size_t)(row0 + 16 * h + m16) * 1024 + 64 * ((kk + rot) & 15) + 16 * q);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    asm volatile("" : "+v"(raw[h][kk]));
                    in_[h][kk] = __builtin_bit_cast(bf16x8, raw[h][kk]);
                }
        }
        for (int layer = 0; layer < layers; ++layer) {
            const bool store = store_all || layer == layers - 1;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj, ++g) {
                // slot g's pieces have landed when at most the (NSLOT-2) younger slots' pieces (8 per wave each) are pending;
                // stores and X loads issued in between only make the count conservative
                if (g + NSLOT - 2 < total) { FE_WAIT((NSLOT - 2) * 8); } else { FE_WAIT(0); }
                LDS_BARRIER();
                if (g + NSLOT - 1 < total) issue(g + NSLOT - 1);
                const char* slot = smem + (g % NSLOT) * FE_SLOT;
                const char* fb = slot + m16 * FE_STRIDE + 16 * q;
                f32x4 acc[2][2];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[h][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    // register position kk holds k block (kk + rot) & 15 (X is loaded, and every layer's output is produced, in
                    // this workgroup's rotated block order)
                    const int kb = rotate ? 64 * ((kk + rot) & 15) : 64 * kk;
                    const bf16x8 a0 = *(const bf16x8*)(fb + kb);
                    const bf16x8 a1 = *(const bf16x8*)(fb + 16 * FE_STRIDE + kb);
                    if (!no_mfma) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            acc[h][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, in_[h][kk], acc[h][0], 0, 0, 0);
                            acc[h][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, in_[h][kk], acc[h][1], 0, 0, 0);
                        }
                    } else {
                        asm volatile("" ::"v"(a0), "v"(a1));
                    }
                }
                // bias + ReLU, 8 consecutive features 32jj + 8q .. +7 of rows 16h + m16 -> next layer's k-step jj
                const int jf = (jj + rot) & 15;                  // the feature block this slot produced
                const f32x4 b0 = *(const f32x4*)(lbias + layer * 512 + 32 * jf + 8 * q);
                const f32x4 b1 = *(const f32x4*)(lbias + layer * 512 + 32 * jf + 8 * q + 4);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 v0 = acc[h][0] + b0, v1 = acc[h][1] + b1;
                    u32x4 w;
                    w[0] = pack_bf2(fmaxf(v0[0], 0.f), fmaxf(v0[1], 0.f));
                    w[1] = pack_bf2(fmaxf(v0[2], 0.f), fmaxf(v0[3], 0.f));
                    w[2] = pack_bf2(fmaxf(v1[0], 0.f), fmaxf(v1[1], 0.f));
                    w[3] = pack_bf2(fmaxf(v1[2], 0.f), fmaxf(v1[3], 0.f));
                    out_[h][jj] = __builtin_bit_cast(bf16x8, w);
                    if (store) {
                        bf16_t* dst = Hout + (size_t)(store_all ? layer : 0) * M * 512;
                        fe_store16(dst + (size_t)(row0 + 16 * h + m16) * 512 + 32 * jf + 8 * q, w);
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) in_[h][kk] = out_[h][kk];
        }
    }
}

extern "C" int murcl_debug_fused_encoder(const void* X, const void* W, const float* bias, void* Hout, int M, int layers,
                                         int store_all, int nslot, int no_mfma, int rotate, hipStream_t stream) {
    if (M <= 0 || M % 128 || layers < 1 || layers > 3) return -1;
    const int grid = M / 128 < 256 ? M / 128 : 256;
#define FE(NS)                                                                                                  \
    if (nslot == NS) {                                                                                          \
        auto k = fused_encoder_probe_kernel<NS>;                                                                \
        const int lds = NS * FE_SLOT + 3 * 512 * 4;                                                             \
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                   \
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, stream, (const bf16_t*)X, (const bf16_t*)W, bias,     \
                           (bf16_t*)Hout, M, layers, store_all, no_mfma, rotate);                               \
        return MURCL_CHECK_LAUNCH();                                                                            \
    }
    FE(3) FE(4)
#undef FE
    return -1;
}
