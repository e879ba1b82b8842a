// K2 backward: gradient of ABMIL attention pooling w.r.t. the pre-tanh activations
// (autograd of models/abmil.py:38-42).
//
// With p = softmax_N(s), A = p/sqrt(N), M = A.H and upstream dM:
//   g_n  = dM.H_n / sqrt(N)            c = dM.M  ( = sum_n p_n g_n )
//   ds_n = p_n (g_n - c)
//   dT[n,d] = ds_n * wb[d] * (1 - t[n,d]^2),   t = tanh(Wa H_n + ba)   (recomputed on the MFMAs)
//   dba += sum_n dT[n,:]   dwb += sum_n ds_n t[n,:]   dbb += sum_n ds_n
// dH = dT.Wa + A (x) dM and dWa = dT^T.H are then plain GEMMs (gemm.hip epilogue RANK1_MASK / TN).
//
// Same streaming structure as the forward (attn_pool.hip): persistent workgroups, 4-slot LDS-DMA
// ring of 32 KiB H tiles, Wa slice resident in registers.  The saved raw scores of a tile arrive by
// a ninth (4-byte) LDS-DMA per wave so that the main loop contains no compiler-counted loads.
#include "common.h"

#define K2_L 512
#define K2_D 128
#define K2_SLOT 32768
#define K2_NSLOT 4
#define KB_OFF_SC (K2_NSLOT * K2_SLOT)                 // [4 slots][4 waves][64] f32
#define KB_OFF_DS (KB_OFF_SC + 4 * 4 * 64 * 4)         // [32] f32
#define KB_LDS_BYTES (KB_OFF_DS + 32 * 4)

template <typename T> struct KB {
    static constexpr int ROWB = K2_L * (int)sizeof(T);
    static constexpr int TR = K2_SLOT / ROWB;            // 32 / 16
    static constexpr int NI = TR / 16;
    static constexpr int CPR = ROWB / 16;
    static constexpr int NKK = K2_L * (int)sizeof(T) / 64;
    static constexpr int RPW = TR / 4;                   // rows per wave for the g dot: 8 / 4
};
template <typename T> struct BFrag;
template <> struct BFrag<bf16_t> { typedef bf16x8 type; };
template <> struct BFrag<float> { typedef f32x4 type; };
template <typename T>
__device__ __forceinline__ f32x4 kb_mma(typename BFrag<T>::type a, typename BFrag<T>::type b, f32x4 c);
template <> __device__ __forceinline__ f32x4 kb_mma<bf16_t>(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 kb_mma<float>(f32x4 a, f32x4 b, f32x4 c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
    return c;
}

template <typename T, bool EXACT_TANH>
__global__ __launch_bounds__(256, 1) void abmil_pool_bwd_kernel(
    const T* __restrict__ H, const T* __restrict__ Wa, const float* __restrict__ ba, const float* __restrict__ wb,
    const float* __restrict__ scores, const float* __restrict__ ml, const float* __restrict__ Mp,
    const float* __restrict__ dM, T* __restrict__ dT, float* __restrict__ dba, float* __restrict__ dwb,
    float* __restrict__ dbb, int B, int N, int chunk_rows, int S, float inv_sqrt_n) {
    typedef KB<T> C_;
    typedef typename BFrag<T>::type frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q4 = lane >> 4, r16 = lane & 15;
    const unsigned lds0 = lds_off(smem);
    float* scb = (float*)(smem + KB_OFF_SC);
    float* dsbuf = (float*)(smem + KB_OFF_DS);

    const int n_items = B * S;
    const int tiles_per_item = chunk_rows / C_::TR;
    const int my_items = (n_items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int my_tiles = my_items * tiles_per_item;
    if (my_tiles <= 0) return;

    auto issue = [&](int seq) {
        const int item = blockIdx.x + (seq / tiles_per_item) * gridDim.x;
        const int bag = item / S, ch = item - bag * S;
        const int row0 = ch * chunk_rows + (seq % tiles_per_item) * C_::TR;
        const char* base = (const char*)(H + (size_t)bag * N * K2_L);
        const int sl = seq % K2_NSLOT;
        const unsigned slot = lds0 + sl * K2_SLOT;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ci = (j * 4 + wave) * 64 + lane;
            const int row = ci / C_::CPR, pos = ci % C_::CPR;
            const int grow = min(row0 + row, N - 1);
            glds16(base + (size_t)grow * C_::ROWB + ((pos ^ (row & 15)) << 4), slot + (j * 4 + wave) * 1024);
        }
        // ninth op: this wave's private copy of the tile's saved scores (lane r <-> row r)
        glds4(scores + (size_t)bag * N + min(row0 + lane, N - 1), lds0 + KB_OFF_SC + (sl * 4 + wave) * 256);
    };

    const int pre = min(3, my_tiles);
    for (int s = 0; s < pre; ++s) issue(s);

    frag_t wa[2][C_::NKK];
    float ba_r[2][4], wb_r[2][4], dba_r[2][4], dwb_r[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const T* wrow = Wa + (size_t)(32 * wave + 16 * j + r16) * K2_L;
#pragma unroll
        for (int kk = 0; kk < C_::NKK; ++kk) {
            wa[j][kk] = *(const frag_t*)((const char*)wrow + (4 * kk + q4) * 16);
            asm volatile("" : "+v"(wa[j][kk]));      // keep resident: never re-load inside the tile loop
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ba_r[j][r] = ba[32 * wave + 16 * j + 4 * q4 + r];
            wb_r[j][r] = wb[32 * wave + 16 * j + 4 * q4 + r];
            dba_r[j][r] = 0.f;
            dwb_r[j][r] = 0.f;
        }
    }
    float dbb_acc = 0.f;
    float dmr[8], bag_m = 0.f, bag_invl = 0.f, bag_c = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) dmr[e] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    for (int seq = 0; seq < my_tiles; ++seq) {
        // Only the 9 LDS-DMA ops per tile are counted; the dT stores issued in between are also
        // younger than tile `seq`'s loads, which only makes this wait more conservative.
        const int ahead = min(2, my_tiles - 1 - seq);
        if (ahead == 2) { WAIT_VMCNT(18); } else if (ahead == 1) { WAIT_VMCNT(9); } else { WAIT_VMCNT(0); }
        LDS_BARRIER();
        if (seq + 3 < my_tiles) issue(seq + 3);

        const int tin = seq % tiles_per_item;
        const int item = blockIdx.x + (seq / tiles_per_item) * gridDim.x;
        const int bag = item / S, ch = item - bag * S;
        const int row0 = ch * chunk_rows + tin * C_::TR;
        const int sl = seq % K2_NSLOT;
        const char* tile = smem + sl * K2_SLOT;

        if (tin == 0) {                         // new item: per-bag constants
            const float* dmb = dM + (size_t)bag * K2_L;
            const float* mb = Mp + (size_t)bag * K2_L;
            float cpart = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int col = (sizeof(T) == 2) ? 8 * lane + e : (e < 4 ? 4 * lane + e : 256 + 4 * lane + (e - 4));
                dmr[e] = dmb[col];
                cpart += dmr[e] * mb[col];
            }
            bag_c = wave_sum(cpart);
            bag_m = ml[2 * bag];
            bag_invl = 1.0f / ml[2 * bag + 1];
        }

        // ---- g_n and ds_n for this wave's rows
#pragma unroll
        for (int rr = 0; rr < C_::RPW; ++rr) {
            const int r = wave * C_::RPW + rr;
            float part = 0.f;
            if (sizeof(T) == 2) {
                const u32x4 u = *(const u32x4*)(tile + r * C_::ROWB + ((lane ^ (r & 15)) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) part += dmr[2 * e] * bf_lo(u[e]) + dmr[2 * e + 1] * bf_hi(u[e]);
            } else {
                const f32x4 a = *(const f32x4*)(tile + r * C_::ROWB + ((lane ^ (r & 15)) << 4));
                const f32x4 b = *(const f32x4*)(tile + r * C_::ROWB + (((64 + lane) ^ (r & 15)) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) part += dmr[e] * a[e] + dmr[4 + e] * b[e];
            }
            const float g = wave_sum(part) * inv_sqrt_n;
            if (lane == 0) {
                const float s = scb[(sl * 4 + wave) * 64 + r];
                const float p = __expf(s - bag_m) * bag_invl;
                dsbuf[r] = (row0 + r < N) ? p * (g - bag_c) : 0.f;
            }
        }

        // ---- recompute pre-activations for the tile (this wave's 32 columns of D)
        f32x4 acc[C_::NI][2];
#pragma unroll
        for (int i = 0; i < C_::NI; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < C_::NKK; ++kk) {
#pragma unroll
            for (int i = 0; i < C_::NI; ++i) {
                const int row = 16 * i + r16, c = 4 * kk + q4;
                frag_t h = *(const frag_t*)(tile + row * C_::ROWB + ((c ^ (row & 15)) << 4));
                acc[i][0] = kb_mma<T>(wa[0][kk], h, acc[i][0]);
                acc[i][1] = kb_mma<T>(wa[1][kk], h, acc[i][1]);
            }
        }
        LDS_BARRIER();                          // dsbuf complete

        if (wave == 0 && lane < C_::TR) dbb_acc += dsbuf[lane];
#pragma unroll
        for (int i = 0; i < C_::NI; ++i) {
            const int row = 16 * i + r16;
            const float ds = dsbuf[row];
            const int grow = row0 + row;
            // rows past N are redirected to the 32 spare rows after the last bag (never read)
            T* dst = dT + ((grow < N) ? ((size_t)bag * N + grow) : ((size_t)B * N + row)) * K2_D + 32 * wave + 4 * q4;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = acc[i][j][r] + ba_r[j][r];
                    const float t = EXACT_TANH ? tanhf(x) : fast_tanh(x);
                    o[r] = ds * wb_r[j][r] * (1.f - t * t);
                    dba_r[j][r] += o[r];
                    dwb_r[j][r] += ds * t;
                }
                store4<T>(dst + 16 * j, o);
            }
        }
    }

    // ---- flush the parameter-gradient partials
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = dba_r[j][r], w = dwb_r[j][r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); w += __shfl_xor(w, o, 64); }
            if (r16 == 0) {
                atomicAdd(dba + 32 * wave + 16 * j + 4 * q4 + r, a);
                atomicAdd(dwb + 32 * wave + 16 * j + 4 * q4 + r, w);
            }
        }
    if (wave == 0) {
        const float t = wave_sum(dbb_acc);
        if (lane == 0) atomicAdd(dbb, t);
    }
}

extern "C" int murcl_abmil_pool_workspace(int B, int N, int dtype, int* chunk_rows, int* n_chunks);

// C-ABI: see include/murcl_amd.h.  dT must hold (B*N + 32) rows of D elements.
extern "C" int murcl_abmil_pool_bwd(const void* H, const void* Wa, const float* ba, const float* wb, const float* scores,
                                    const float* ml, const float* M, const float* dM, void* dT, float* dba, float* dwb,
                                    float* dbb, int B, int N, int L, int D, int dtype, int exact_tanh,
                                    hipStream_t stream) {
    if (L != K2_L || D != K2_D) return -1;
    if (B <= 0 || N <= 0) return 0;
    int chunk, S;
    murcl_abmil_pool_workspace(B, N, dtype, &chunk, &S);
    const int items = B * S;
    const int grid = items < 256 ? items : 256;
    const float isn = 1.0f / sqrtf((float)N);
#define KB_LAUNCH(T, EX)                                                                                       \
    {                                                                                                          \
        auto k = abmil_pool_bwd_kernel<T, EX>;                                                                 \
        static bool once = false;                                                                              \
        if (!once) {                                                                                           \
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, KB_LDS_BYTES);     \
            once = true;                                                                                       \
        }                                                                                                      \
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), KB_LDS_BYTES, stream, (const T*)H, (const T*)Wa, ba, wb,  \
                           scores, ml, M, dM, (T*)dT, dba, dwb, dbb, B, N, chunk, S, isn);                     \
    }
    if (dtype == MURCL_DTYPE_BF16) {
        if (exact_tanh) KB_LAUNCH(bf16_t, true) else KB_LAUNCH(bf16_t, false)
    } else if (dtype == MURCL_DTYPE_F32) {
        if (exact_tanh) KB_LAUNCH(float, true) else KB_LAUNCH(float, false)
    } else {
        return -1;
    }
#undef KB_LAUNCH
    return MURCL_CHECK_LAUNCH();
}
