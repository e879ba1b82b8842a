// Prototype v2 (dev tool, NOT on the product path): the fused ABMIL encoder forward of fused_encoder_probe.hip with the latency
// chain that bound v1 removed.  v1 read each k-step's weight fragments from LDS right before the MFMAs that use them; with ONE
// wave per SIMD nothing covers that round trip, so a 32-feature slot took ~2 900 cycles against 1 024 cycles of MFMAs - and the
// "memory only" ablation (no MFMA, same just-in-time reads) was bound by the same chain, which round 3 read as an L2 -> LDS
// streaming limit.  tools/l2_loader_probe.py: the loader alone pulls W at ~130 GB/s per CU (33 TB/s chip-wide).
//
// v2: the fragments of slot g+1 are requested while the MFMAs of slot g run (a whole slot ahead: 128 VGPRs, there is room at one
// wave per SIMD), and the bias/ReLU/pack/store epilogue of slot g runs inside slot g+1's MFMA stream.
//
//   H_l = relu(H_{l-1} W_l^T + b_l),  l = 1..layers,  H_0 = X [M,512] bf16, W_l [512,512] bf16, f32 accumulate
#include "common.h"

#define F2_STRIDE 1056
#define F2_SLOT (32 * F2_STRIDE)
#define F2_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

__device__ __forceinline__ void f2_store16(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 f2_load16(const void* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// The register plan hipcc does not find by itself (it kept the activations in AGPRs as SPILL space: 64 v_accvgpr_read per slot in
// the MFMA stream): weight fragments F (128) live in the accumulation file and feed the MFMAs from there (ds_read_b128 lands there
// directly); the finished outputs out_ (128, written once per slot, read once per layer) are parked there dword by dword; input
// activations in_ (128) and the accumulators are VGPRs.  Hazards the compiler cannot see: an MFMA's
// result needs >= 12 wait states before a VALU reads it - the epilogue of a slot runs 8 MFMAs into the next one (s_nop before the
// one immediate epilogue per layer).
__device__ __forceinline__ void f2_mfma0(f32x4& acc, bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(a), "v"(b));
}
__device__ __forceinline__ void f2_mfma(f32x4& acc, bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
}

template <int NSLOT, int ABL>      // ABL (timing ablations, wrong results): bit 0 no weight DMA in the loop, bit 1 no fragment reads, bit 2 no barrier / wait, bit 3 no stores
__global__ __launch_bounds__(256, 1) void fused_encoder2_probe_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                                        const float* __restrict__ bias, bf16_t* __restrict__ Hout,
                                                                        int M, int layers, int store_all) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int OFF_BIAS = NSLOT * F2_SLOT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, m16 = lane & 15;
    const unsigned lds0 = lds_off(smem);
    const int n_tiles = M / 128;
    const int my_tiles = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;
    const int total = my_tiles * layers * 16;                 // weight slots this workgroup walks

    float* lbias = (float*)(smem + OFF_BIAS);
    for (int i = tid; i < layers * 512; i += 256) lbias[i] = bias[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // slot g = ((tile * layers) + layer) * 16 + jj covers rows 32 s .. 32 s + 31 of the stacked [layers * 512, 512] weights, s = g mod
    // (16 layers); LDS row p = 16b + i  <-  row 32 s + 8(i/4) + 4b + (i%4).  Piece u of a slot = LDS rows 4u + wave.
    const int period = 16 * layers;
    int s_issue = 0, slot_issue = 0;                   // stacked-weight slot / ring slot of the next slot to request
    auto piece = [&](int u) {
        const int p = u * 4 + wave, b = p >> 4, i = p & 15;
        const int wrow = 32 * s_issue + 8 * (i >> 2) + 4 * b + (i & 3);
        glds16_u((const char*)W + (size_t)wrow * 1024, lane * 16, lds0 + slot_issue * F2_SLOT + p * F2_STRIDE);
    };
    auto advance = [&]() {
        s_issue = s_issue + 1 == period ? 0 : s_issue + 1;
        slot_issue = slot_issue + 1 == NSLOT ? 0 : slot_issue + 1;
    };
    // fragments of slot g: lane (q, m16) reads W-slot rows m16 and 16 + m16, k = 32kk + 8q .. +7
    auto frag = [&](int g, int kk, int b) -> bf16x8 {
        return *(const bf16x8*)(smem + (g % NSLOT) * F2_SLOT + (16 * b + m16) * F2_STRIDE + 16 * q + 64 * kk);
    };
    for (int g = 0; g < NSLOT; ++g) {                                // slots 0 .. NSLOT-1 in flight (requests past the last slot
#pragma unroll
        for (int u = 0; u < 8; ++u) piece(u);                        //  fetch valid weights into ring slots nobody reads)
        advance();
    }

    bf16x8 in_[2][16], F[16][2];
    unsigned out_[2][16][4];                                   // parked in the accumulation file, dword by dword
    // slot 0's fragments: wait for slot 0 (the NSLOT-1 younger slots stay in flight), make it visible, read it
    F2_WAIT((NSLOT - 1) * 8);
    LDS_BARRIER();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) { F[kk][0] = frag(0, kk, 0); F[kk][1] = frag(0, kk, 1); }

    int g = 0;
    for (int t = 0; t < my_tiles; ++t) {
        const int row0 = ((int)blockIdx.x + t * (int)gridDim.x) * 128 + 32 * wave;
        {
            u32x4 raw[2][16];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kk = 0; kk < 16; ++kk)
                    raw[h][kk] = f2_load16((const char*)X + (size_t)(row0 + 16 * h + m16) * 1024 + 64 * kk + 16 * q);
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
            bf16_t* dst = Hout + (size_t)(store_all ? layer : 0) * M * 512;
            f32x4 pacc[2][2];                       // the previous slot's accumulators, finished inside this slot's MFMA stream
            auto epilogue = [&](f32x4 (&a)[2][2], int jf) {
                const f32x4 b0 = *(const f32x4*)(lbias + layer * 512 + 32 * jf + 8 * q);
                const f32x4 b1 = *(const f32x4*)(lbias + layer * 512 + 32 * jf + 8 * q + 4);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 v0 = a[h][0] + b0, v1 = a[h][1] + b1;
                    u32x4 w;
                    w[0] = pack_bf2(fmaxf(v0[0], 0.f), fmaxf(v0[1], 0.f));
                    w[1] = pack_bf2(fmaxf(v0[2], 0.f), fmaxf(v0[3], 0.f));
                    w[2] = pack_bf2(fmaxf(v1[0], 0.f), fmaxf(v1[1], 0.f));
                    w[3] = pack_bf2(fmaxf(v1[2], 0.f), fmaxf(v1[3], 0.f));
#pragma unroll
                    for (int e = 0; e < 4; ++e) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(out_[h][jf][e]) : "v"(w[e]));
                    if (store && !(ABL & 8)) f2_store16(dst + (size_t)(row0 + 16 * h + m16) * 512 + 32 * jf + 8 * q, w);
                }
            };
#pragma unroll
            for (int jj = 0; jj < 16; ++jj, ++g) {
                // slot g+1 must be readable while slot g computes: its pieces were requested NSLOT slots ago; younger than them in this
                // wave's queue are the 8 pieces each of slots g+2 .. g+NSLOT-1 (and stores, which only make the wait conservative)
                if (!(ABL & 4)) {
                    F2_WAIT((NSLOT - 2) * 8);
                    LDS_BARRIER();                    // (also: every wave's reads of slot g have returned - they were issued a slot ago)
                }
                f32x4 acc[2][2];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const bf16x8 a0 = F[kk][0], a1 = F[kk][1];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (kk == 0) { f2_mfma0(acc[h][0], a0, in_[h][kk]); f2_mfma0(acc[h][1], a1, in_[h][kk]); }
                        else { f2_mfma(acc[h][0], a0, in_[h][kk]); f2_mfma(acc[h][1], a1, in_[h][kk]); }
                    }
                    if (!(ABL & 2)) {
                        F[kk][0] = frag(g + 1, kk, 0);              // (past the last slot: stale bytes nobody uses)
                        F[kk][1] = frag(g + 1, kk, 1);
                    }
                    if ((kk & 1) && !(ABL & 1)) piece(kk >> 1);                     // slot g + NSLOT into slot g's place, one piece per two k-steps
                    if (kk == 2 && jj > 0) {
                        // pin: the epilogue's VALU reads of the previous slot's MFMA results may not be scheduled above this point
                        // (12 MFMAs after the last write; the compiler does not know these registers come out of the matrix pipe)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int b = 0; b < 2; ++b) asm volatile("" : "+v"(pacc[h][b]));
                        epilogue(pacc, jj - 1);
                    }
                }
                advance();
                if (jj == 15) {
                    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1])::"memory");   // MFMA results -> VALU: wait states the compiler cannot count
                    epilogue(acc, 15);
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int b = 0; b < 2; ++b) pacc[h][b] = acc[h][b];
                }
            }
            // the parked outputs become the next layer's inputs
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    u32x4 a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a[e]) : "a"(out_[h][kk][e]));
                    in_[h][kk] = __builtin_bit_cast(bf16x8, a);
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the run-ahead requests must land before the LDS is handed on
}

extern "C" int murcl_debug_fused_encoder2(const void* X, const void* W, const float* bias, void* Hout, int M, int layers,
                                          int store_all, int nslot, int abl, hipStream_t stream) {
    if (M <= 0 || M % 128 || layers < 1 || layers > 3) return -1;
    const int grid = M / 128 < 256 ? M / 128 : 256;
#define F2(NS, AB)                                                                                              \
    if (nslot == NS && abl == AB) {                                                                             \
        auto k = fused_encoder2_probe_kernel<NS, AB>;                                                           \
        const int lds = NS * F2_SLOT + 3 * 512 * 4;                                                             \
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                   \
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, stream, (const bf16_t*)X, (const bf16_t*)W, bias,     \
                           (bf16_t*)Hout, M, layers, store_all);                                                \
        return MURCL_CHECK_LAUNCH();                                                                            \
    }
    F2(3, 0) F2(4, 0) F2(4, 1) F2(4, 2) F2(4, 3) F2(4, 4) F2(4, 7) F2(4, 8) F2(4, 15)
#undef F2
    return -1;
}

// ------------------------------------------------------------------------------------------------------------------------------
// v3: the same data flow with EIGHT waves of 16 rows (two waves per SIMD, <= 256 registers each) instead of four waves of 32 rows.
// v2's ablations (tools/fused_probe2.py) put MFMA + epilogue alone at ~1 150 cycles per 32-feature slot (floor 1 024) and the full
// kernel at ~2 230: with one wave per SIMD every LDS-read, LDS-DMA and store instruction that waits for its pipe idles the matrix
// pipe.  Two waves per SIMD cover each other; the price is LDS read volume (every wave reads every weight fragment: 8 x 32 KiB per
// slot = 1 024 LDS cycles, the length of the slot's MFMAs per SIMD).
template <int NSLOT, int D>       // D: k-steps of fragment prefetch
__global__ __launch_bounds__(512, 2) void fused_encoder3_probe_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                                        const float* __restrict__ bias, bf16_t* __restrict__ Hout,
                                                                        int M, int layers, int store_all) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int OFF_BIAS = NSLOT * F2_SLOT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, m16 = lane & 15;
    const unsigned lds0 = lds_off(smem);
    const int n_tiles = M / 128;
    const int my_tiles = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;

    float* lbias = (float*)(smem + OFF_BIAS);
    for (int i = tid; i < layers * 512; i += 512) lbias[i] = bias[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    const int period = 16 * layers;
    int s_issue = 0, slot_issue = 0;
    auto piece = [&](int u) {                       // piece u of a slot = LDS rows 8u + wave (4 pieces per wave and slot)
        const int p = u * 8 + wave, b = p >> 4, i = p & 15;
        const int wrow = 32 * s_issue + 8 * (i >> 2) + 4 * b + (i & 3);
        glds16_u((const char*)W + (size_t)wrow * 1024, lane * 16, lds0 + slot_issue * F2_SLOT + p * F2_STRIDE);
    };
    auto advance = [&]() {
        s_issue = s_issue + 1 == period ? 0 : s_issue + 1;
        slot_issue = slot_issue + 1 == NSLOT ? 0 : slot_issue + 1;
    };
    int slot_read = 0;                               // ring slot of the weight slot whose fragments are being requested
    const char* fbase = smem + m16 * F2_STRIDE + 16 * q;
    auto frag = [&](int rs, int kk, int b) -> bf16x8 {
        return *(const bf16x8*)(fbase + rs * F2_SLOT + 16 * b * F2_STRIDE + 64 * kk);
    };
    for (int g = 0; g < NSLOT; ++g) {
#pragma unroll
        for (int u = 0; u < 4; ++u) piece(u);
        advance();
    }
    bf16x8 in_[16], F[D][2];
    unsigned out_[16][4];
    F2_WAIT((NSLOT - 1) * 4);
    LDS_BARRIER();
#pragma unroll
    for (int kk = 0; kk < D; ++kk) { F[kk][0] = frag(0, kk, 0); F[kk][1] = frag(0, kk, 1); }

    for (int t = 0; t < my_tiles; ++t) {
        const int row0 = ((int)blockIdx.x + t * (int)gridDim.x) * 128 + 16 * wave;
        {
            u32x4 raw[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) raw[kk] = f2_load16((const char*)X + (size_t)(row0 + m16) * 1024 + 64 * kk + 16 * q);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                asm volatile("" : "+v"(raw[kk]));
                in_[kk] = __builtin_bit_cast(bf16x8, raw[kk]);
            }
        }
        for (int layer = 0; layer < layers; ++layer) {
            const bool store = store_all || layer == layers - 1;
            bf16_t* dst = Hout + (size_t)(store_all ? layer : 0) * M * 512;
            f32x4 pacc[2];
            auto epilogue = [&](f32x4 (&a)[2], int jf) {
                const f32x4 b0 = *(const f32x4*)(lbias + layer * 512 + 32 * jf + 8 * q);
                const f32x4 b1 = *(const f32x4*)(lbias + layer * 512 + 32 * jf + 8 * q + 4);
                const f32x4 v0 = a[0] + b0, v1 = a[1] + b1;
                u32x4 w;
                w[0] = pack_bf2(fmaxf(v0[0], 0.f), fmaxf(v0[1], 0.f));
                w[1] = pack_bf2(fmaxf(v0[2], 0.f), fmaxf(v0[3], 0.f));
                w[2] = pack_bf2(fmaxf(v1[0], 0.f), fmaxf(v1[1], 0.f));
                w[3] = pack_bf2(fmaxf(v1[2], 0.f), fmaxf(v1[3], 0.f));
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(out_[jf][e]) : "v"(w[e]));
                if (store) f2_store16(dst + (size_t)(row0 + m16) * 512 + 32 * jf + 8 * q, w);
            };
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                f32x4 acc[2];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    if (kk == 16 - D) {
                        // from here on the fragment requests run into the NEXT weight slot: it must have landed (its pieces are the
                        // oldest in flight: (NSLOT - 2) x 4 younger ones may stay), and once every wave is here the slot being left
                        // can take new weights
                        F2_WAIT((NSLOT - 2) * 4);
                        LDS_BARRIER();
#pragma unroll
                        for (int u = 0; u < 4; ++u) piece(u);
                        advance();
                        slot_read = slot_read + 1 == NSLOT ? 0 : slot_read + 1;
                    }
                    const bf16x8 a0 = F[kk % D][0], a1 = F[kk % D][1];
                    if (kk == 0) { f2_mfma0(acc[0], a0, in_[kk]); f2_mfma0(acc[1], a1, in_[kk]); }
                    else { f2_mfma(acc[0], a0, in_[kk]); f2_mfma(acc[1], a1, in_[kk]); }
                    const int nk = (kk + D) & 15;                   // k-step whose fragments take this ring place (slot_read is already
                    F[kk % D][0] = frag(slot_read, nk, 0);          //  the next slot when kk + D >= 16)
                    F[kk % D][1] = frag(slot_read, nk, 1);
                    if (kk == 4 && jj > 0) {
                        asm volatile("" : "+v"(pacc[0]), "+v"(pacc[1]));
                        epilogue(pacc, jj - 1);
                    }
                }
                if (jj == 15) {
                    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1])::"memory");
                    epilogue(acc, 15);
                } else {
                    pacc[0] = acc[0]; pacc[1] = acc[1];
                }
            }
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                u32x4 a;
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a[e]) : "a"(out_[kk][e]));
                in_[kk] = __builtin_bit_cast(bf16x8, a);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

extern "C" int murcl_debug_fused_encoder3(const void* X, const void* W, const float* bias, void* Hout, int M, int layers,
                                          int store_all, int nslot, int depth, hipStream_t stream) {
    if (M <= 0 || M % 128 || layers < 1 || layers > 3) return -1;
    const int grid = M / 128 < 256 ? M / 128 : 256;
#define F3(NS, DD)                                                                                              \
    if (nslot == NS && depth == DD) {                                                                           \
        auto k = fused_encoder3_probe_kernel<NS, DD>;                                                           \
        const int lds = NS * F2_SLOT + 3 * 512 * 4;                                                             \
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                   \
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, stream, (const bf16_t*)X, (const bf16_t*)W, bias,     \
                           (bf16_t*)Hout, M, layers, store_all);                                                \
        return MURCL_CHECK_LAUNCH();                                                                            \
    }
    F3(4, 4) F3(4, 6) F3(4, 8) F3(3, 4)
#undef F3
    return -1;
}
