// K2 backward: gradient of ABMIL attention pooling w.r.t. the pre-tanh activations
// (autograd of models/abmil.py:38-42).
//
// With p = softmax_N(s), A = p/sqrt(N), M = A.H and upstream dM:
//   g_n  = dM.H_n / sqrt(N)            c = dM.M  ( = sum_n p_n g_n )
//   ds_n = p_n (g_n - c)
//   dT[n,d] = ds_n * wb[d] * (1 - t[n,d]^2),   t = tanh(Wa H_n + ba)   (recomputed on the MFMAs)
//   dba += sum_n dT[n,:]   dwb += sum_n ds_n t[n,:]   dbb += sum_n ds_n
// dH = dT.Wa + A (x) dM and dWa = dT^T.H are then plain GEMMs (panel_gemm.hip RANK1_MASK / gemm.hip TN).
//
// Same streaming structure as the forward (k2_common.h): persistent workgroups (bf16: two 4-wave workgroups per
// CU), 16-row H tiles through a 4-slot LDS-DMA ring with padded rows, Wa slice resident in registers.  The
// per-row dots g_n = dM.H_n also run on the matrix cores: dM enters as rows 0/1 (bf16 hi + lo parts; f32: row 0)
// of an MFMA A operand and every wave covers 1/NW of the k range, partial sums meet in LDS.  The saved raw scores
// of a tile arrive by a fifth (4-byte) LDS-DMA per wave, so the loop contains no compiler-counted loads.
#include "k2_common.h"

#ifndef K2B_REVERSE
#define K2B_REVERSE 0          // 1: walk the items from the last bag down (A/B: tools/ab_build.sh)
#endif

template <typename T> struct KBLds {
    static constexpr int OFF_GPART = K2_NSLOT * K2<T>::SLOT;                        // [NW][16] f32
    static constexpr int OFF_SC = OFF_GPART + K2<T>::NW * 16 * 4;                   // [slot][NW][64] f32
    static constexpr int BYTES = OFF_SC + K2_NSLOT * K2<T>::NW * 256;
};

#ifndef K2B_WPRO
#define K2B_WPRO 1
#endif
template <typename T, bool EXACT_TANH, bool WFRAG = false>
__global__ __launch_bounds__(64 * K2<T>::NW, 2) void abmil_pool_bwd_kernel(
    const T* __restrict__ H, const T* __restrict__ Wa, const float* __restrict__ ba, const float* __restrict__ wb,
    const float* __restrict__ scores, const float* __restrict__ ml, const float* __restrict__ Mp,
    const float* __restrict__ dM, T* __restrict__ dT, float* __restrict__ dba, float* __restrict__ dwb,
    float* __restrict__ dbb, float* __restrict__ part_ws, float* __restrict__ A_out, int B, int N, int chunk_rows, int S,
    float inv_sqrt_n) {
    constexpr bool wfrag = WFRAG && sizeof(T) == 2;      // Wa in fragment order (a template parameter: attn_pool.hip)
    typedef K2<T> C_;
    typedef KBLds<T> L_;
    typedef typename WFrag<T>::type frag_t;
    constexpr int KW = C_::NKK / C_::NW;                 // k-steps of the g dot owned by one wave: 4 / 4
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q4 = lane >> 4, r16 = lane & 15;
    const unsigned lds0 = lds_off(smem);
    float* gpart = (float*)(smem + L_::OFF_GPART);
    const float* scb = (const float*)(smem + L_::OFF_SC);

    const int n_items = B * S;
    const int tiles_per_item = chunk_rows / C_::TR;
    const int my_items = (n_items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int my_tiles = my_items * tiles_per_item;
    if (my_tiles <= 0) return;

    K2Pos ip, cp;
    ip.init(blockIdx.x, S, K2B_REVERSE ? n_items - 1 : -1);
    cp.init(blockIdx.x, S, K2B_REVERSE ? n_items - 1 : -1);
    auto issue = [&](int seq) {
        const int row0 = ip.ch * chunk_rows + ip.tin * C_::TR;
        const int sl = seq & (K2_NSLOT - 1);
        k2_issue_tile<T>(H + (size_t)ip.bag * N * K2_L, row0, N, lds0 + sl * C_::SLOT, wave, lane);
        // fifth op: this wave's private copy of the tile's saved scores (lane r <-> row r, clamped)
        glds4_s(scores + (size_t)ip.bag * N, (unsigned)min(row0 + (lane & 15), N - 1) * 4u,
                lds0 + L_::OFF_SC + (sl * C_::NW + wave) * 256);
        ip.next(tiles_per_item, gridDim.x, S);
    };
    const int pre = min(3, my_tiles);
    // weight prologue as in the forward kernel (attn_pool.hip): whole-row LDS-DMA pieces through the still empty tile ring
    // instead of fragment-shaped global loads that touch 64 cache lines per instruction
    constexpr bool WPRO_C = K2B_WPRO && sizeof(T) == 2 && C_::NW * 16 * C_::PADB <= K2_NSLOT * C_::SLOT;
    constexpr bool WPRO = WPRO_C && !wfrag;             // `wfrag`: Wa in fragment order, loaded straight into registers beside the first tiles (attn_pool.hip)
    if (!WPRO)
        for (int s = 0; s < pre; ++s) issue(s);

    frag_t wa[C_::NJ][C_::NKK];
    float ba_r[C_::NJ][4], wb_r[C_::NJ][4], dba_r[C_::NJ][4], dwb_r[C_::NJ][4];
#pragma unroll
    for (int j = 0; j < C_::NJ; ++j) {
        const char* wrow = (const char*)(Wa + (size_t)(C_::DW * wave + 16 * j + r16) * K2_L);
        if constexpr (wfrag) {
            const char* fblk = (const char*)Wa + ((size_t)((C_::DW * wave) / 16 + j) * C_::NKK) * 1024 + lane * 16;
#pragma unroll
            for (int kk = 0; kk < C_::NKK; ++kk) wa[j][kk] = *(const frag_t*)(fblk + kk * 1024);
        } else if (WPRO) {
            const char* wblk = (const char*)(Wa + (size_t)(C_::DW * wave + 16 * j) * K2_L);
            const unsigned stage = lds0 + wave * 16 * C_::PADB;
#pragma unroll
            for (int u = 0; u < 16; ++u) glds16_u(wblk + (size_t)u * C_::ROWB, lane * 16, stage + u * C_::PADB);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const char* fb = smem + (wave * 16 + r16) * C_::PADB + C_::NKK * q4 * 16;
#pragma unroll
            for (int kk = 0; kk < C_::NKK; ++kk) {
                wa[j][kk] = *(const frag_t*)(fb + kk * 16);
                asm volatile("" : "+v"(wa[j][kk]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the next block's pieces overwrite these rows
        }
#pragma unroll
        for (int kk = 0; kk < C_::NKK && !WPRO && !wfrag; ++kk) {
            wa[j][kk] = *(const frag_t*)(wrow + (kk + C_::NKK * q4) * 16);
            asm volatile("" : "+v"(wa[j][kk]));      // keep resident: never re-load inside the tile loop
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ba_r[j][r] = ba[C_::DW * wave + 16 * j + 4 * q4 + r];
            wb_r[j][r] = wb[C_::DW * wave + 16 * j + 4 * q4 + r];
            dba_r[j][r] = 0.f;
            dwb_r[j][r] = 0.f;
        }
    }
    float dbb_acc = 0.f;
    frag_t dmf[KW];                                   // dM as MFMA A-operand rows for this wave's k-steps
    float bag_m = 0.f, bag_invl = 0.f, bag_c = 0.f;
    int cur_bag = -1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (wfrag) {
#pragma unroll
        for (int j = 0; j < C_::NJ; ++j)
#pragma unroll
            for (int kk = 0; kk < C_::NKK; ++kk) asm volatile("" : "+v"(wa[j][kk]));
    }
    if (WPRO) {
        LDS_BARRIER();                          // every wave has read its fragments back: the ring is free for tiles
        for (int s = 0; s < pre; ++s) issue(s);
    }

    for (int seq = 0; seq < my_tiles; ++seq) {
        // only the 5 LDS-DMA ops per tile are counted; the dT stores issued in between are also younger than
        // tile seq's loads, which only makes this wait more conservative
        const int ahead = min(2, my_tiles - 1 - seq);
        if (ahead == 2) { WAIT_VMCNT(10); } else if (ahead == 1) { WAIT_VMCNT(5); } else { WAIT_VMCNT(0); }
        LDS_BARRIER();
        if (seq + 3 < my_tiles) issue(seq + 3);

        const int bag = cp.bag;
        const int row0 = cp.ch * chunk_rows + cp.tin * C_::TR;
        const int sl = seq & (K2_NSLOT - 1);
        const char* tile = smem + sl * C_::SLOT;

        if (bag != cur_bag) {                   // new bag: per-bag constants (compiler-visible loads, rare)
            cur_bag = bag;
            const float* dmb = dM + (size_t)bag * K2_L;
            const float* mb = Mp + (size_t)bag * K2_L;
            float cpart = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) cpart += dmb[8 * lane + e] * mb[8 * lane + e];
            bag_c = wave_sum(cpart);
            bag_m = ml[2 * bag];
            bag_invl = 1.0f / ml[2 * bag + 1];
            // A fragments: lane (q4, r16) holds A[row r16][k of chunk (kk + NKK*q4)], kk = KW*wave + i
#pragma unroll
            for (int i = 0; i < KW; ++i) {
                const int chunk = (KW * wave + i) + C_::NKK * q4;
                if (sizeof(T) == 2) {
                    bf16x8 f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = dmb[chunk * 8 + e];
                        const bf16_t hi = f2bf(v);
                        const bf16_t lo = f2bf(v - bf2f(hi));
                        f[e] = (short)(r16 == 0 ? hi : (r16 == 1 ? lo : (bf16_t)0));
                    }
                    dmf[i] = __builtin_bit_cast(frag_t, f);
                } else {
                    f32x4 f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] = (r16 == 0) ? dmb[chunk * 4 + e] : 0.f;
                    dmf[i] = __builtin_bit_cast(frag_t, f);
                }
            }
        }

        // ---- phase 1: pre-activations (this wave's DW columns) + this wave's k-quarter of g = H.dM
        f32x4 acc[C_::NJ];
#pragma unroll
        for (int j = 0; j < C_::NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 gacc = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* hbase = tile + r16 * C_::PADB + C_::NKK * q4 * 16;
        {
            // explicit software pipeline (as in the forward): fragments of k-group g+1 are requested before the MFMAs
            // of group g issue.  The g dot of this wave's k-quarter (k-steps KW*wave .. +KW-1) reuses the same
            // fragments: a wave-uniform branch adds its MFMAs to the groups that hold them.
            constexpr int GK = 2, NG = C_::NKK / GK;
            static_assert(KW % GK == 0, "a wave's g-dot k-steps must be whole prefetch groups");
            frag_t hq[2][GK];
#pragma unroll
            for (int k2 = 0; k2 < GK; ++k2) hq[0][k2] = *(const frag_t*)(hbase + k2 * 16);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) {
#pragma unroll
                    for (int k2 = 0; k2 < GK; ++k2) hq[(g + 1) & 1][k2] = *(const frag_t*)(hbase + ((g + 1) * GK + k2) * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k2 = 0; k2 < GK; ++k2)
#pragma unroll
                    for (int j = 0; j < C_::NJ; ++j) acc[j] = k2_mma<T>(wa[j][g * GK + k2], hq[g & 1][k2], acc[j]);
                if ((g * GK) / KW == wave) {
#pragma unroll
                    for (int k2 = 0; k2 < GK; ++k2) gacc = k2_mma<T>(dmf[(g * GK) % KW + k2], hq[g & 1][k2], gacc);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (q4 == 0) gpart[wave * 16 + r16] = gacc[0] + gacc[1];     // rows 0 (+1): hi (+lo) parts
        LDS_BARRIER();

        // ---- phase 2: ds for row r16 (every lane quarter redundantly), then dT
        float g = 0.f;
#pragma unroll
        for (int w = 0; w < C_::NW; ++w) g += gpart[w * 16 + r16];
        g *= inv_sqrt_n;
        const float sc = scb[(sl * C_::NW + wave) * 64 + r16];
        const float p = (EXACT_TANH ? __expf(sc - bag_m) : fast_exp(sc - bag_m)) * bag_invl;
        const int grow = row0 + r16;
        const float ds = (grow < N) ? p * (g - bag_c) : 0.f;
        if (wave == 0 && q4 == 0) {
            dbb_acc += ds;
            // the normalised attention row (abmil.py:40-41) for the rank-1 term of the input gradient: the forward pass no longer
            // forms it (round 6: its per-bag merge lives in the decoder launch), this pass has p in registers
            if (A_out && grow < N) A_out[(size_t)bag * N + grow] = p * inv_sqrt_n;
        }
        // rows past N are redirected to the 32 spare rows after the last bag (never read)
        T* dst = dT + ((grow < N) ? ((size_t)bag * N + grow) : ((size_t)B * N + r16)) * K2_D + C_::DW * wave + 4 * q4;
        f32x4 o[C_::NJ];
#pragma unroll
        for (int j = 0; j < C_::NJ; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = acc[j][r] + ba_r[j][r];
                const float t = EXACT_TANH ? tanhf(x) : fast_tanh(x);
                o[j][r] = ds * wb_r[j][r] * (1.f - t * t);
                dba_r[j][r] += o[j][r];
                dwb_r[j][r] += ds * t;
            }
        }
        if constexpr (sizeof(T) == 2) {
            // a lane holds 4 columns of each of its two 16-column groups (8 + 8 bytes, 32-byte row segments).  Swapping
            // the j = 1 words of the even lane quarters with the j = 0 words of the odd ones (one v_permlane16_swap per
            // word) leaves every lane with 8 consecutive columns: ONE 16-byte store, 64 contiguous bytes per row.
            static_assert(C_::NJ == 2, "bf16 path: two column groups per wave");
            const auto s0 = __builtin_amdgcn_permlane16_swap(pack_bf2(o[0][0], o[0][1]), pack_bf2(o[1][0], o[1][1]), false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(pack_bf2(o[0][2], o[0][3]), pack_bf2(o[1][2], o[1][3]), false, false);
            // (dst already points at column DW*wave + 4*q4) -> column DW*wave + 16*(q4&1) + 4*(q4&~1)
            *(u32x4*)(dst + 16 * (q4 & 1) - 4 * (q4 & 1)) = u32x4{s0[0], s1[0], s0[1], s1[1]};
        } else {
#pragma unroll
            for (int j = 0; j < C_::NJ; ++j) store4<T>(dst + 16 * j, o[j]);
        }
        cp.next(tiles_per_item, gridDim.x, S);
    }

    // ---- publish this workgroup's parameter-gradient partials (sum over the 16 rows = lanes of a quarter) as one row of
    // part_ws.  Adding them atomically to dba/dwb/dbb from all 512 workgroups cost 42 of 121 us: float atomics execute at
    // the memory side and 512 adders per address serialise; abmil_pool_bwd_reduce_kernel sums the rows instead.
    float* prow = part_ws + (size_t)blockIdx.x * (2 * K2_D + 1);
#pragma unroll
    for (int j = 0; j < C_::NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = row16_sum(dba_r[j][r]), w = row16_sum(dwb_r[j][r]);
            if (r16 == 0) {
                prow[C_::DW * wave + 16 * j + 4 * q4 + r] = a;
                prow[K2_D + C_::DW * wave + 16 * j + 4 * q4 + r] = w;
            }
        }
    if (wave == 0) {
        const float t = row16_sum(dbb_acc);
        if (lane == 0) prow[2 * K2_D] = t;
    }
}

// dba[c] += sum_w part[w][c], dwb[c] += sum_w part[w][D + c], dbb += sum_w part[w][2D]   (16 columns x 16 row lanes)
__global__ __launch_bounds__(256) void abmil_pool_bwd_reduce_kernel(const float* __restrict__ part, int n_wg,
                                                                    float* __restrict__ dba, float* __restrict__ dwb,
                                                                    float* __restrict__ dbb) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl, W = 2 * K2_D + 1;
    float s = 0.f;
    if (c < W) {
        float t4[4] = {0.f, 0.f, 0.f, 0.f};             // four independent chains: the loads of a pass are all in flight
        int w = rl;
        for (; w + 48 < n_wg; w += 64)
#pragma unroll
            for (int u = 0; u < 4; ++u) t4[u] += part[(size_t)(w + 16 * u) * W + c];
        for (; w < n_wg; w += 16) t4[0] += part[(size_t)w * W + c];
        s = (t4[0] + t4[1]) + (t4[2] + t4[3]);
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < W) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        float* dst = c < K2_D ? dba + c : (c < 2 * K2_D ? dwb + (c - K2_D) : dbb);
        *dst += t;
    }
}

extern "C" int murcl_abmil_pool_workspace(int B, int N, int dtype, int* chunk_rows, int* n_chunks);

// C-ABI: see include/murcl_amd.h.  dT must hold (B*N + 32) rows of D elements, part_ws 512*(2D+1) floats.
extern "C" int murcl_abmil_pool_bwd(const void* H, const void* Wa, const float* ba, const float* wb, const float* scores,
                                    const float* ml, const float* M, const float* dM, void* dT, float* dba, float* dwb,
                                    float* dbb, float* part_ws, float* A_out, int B, int N, int L, int D, int dtype,
                                    int exact_tanh, hipStream_t stream) {
    if (L != K2_L || D != K2_D || !part_ws) return -1;
    const int wfrag = (exact_tanh >> 1) & 1;         // flags: bit 0 = exact tanh, bit 1 = Wa in fragment order (bf16 only)
    exact_tanh &= 1;
    if (wfrag && dtype != MURCL_DTYPE_BF16) return -1;
    if (B <= 0 || N <= 0) return 0;
    int chunk, S;
    murcl_abmil_pool_workspace(B, N, dtype, &chunk, &S);
    const int items = B * S;
    const int max_grid = murcl_cu_budget() * (dtype == MURCL_DTYPE_BF16 ? 2 : 1);
    const int grid = items < max_grid ? items : max_grid;
    const float isn = 1.0f / sqrtf((float)N);
#define KB_LAUNCH(T, EX, WF)                                                                                      \
    {                                                                                                          \
        auto k = abmil_pool_bwd_kernel<T, EX, WF>;                                                                \
        static MurclOncePerDevice once;                                                                                    \
        if (once.first()) {                                                                                           \
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, KBLds<T>::BYTES);  \
                                                                                                                  \
        }                                                                                                      \
        hipLaunchKernelGGL(k, dim3(grid), dim3(64 * K2<T>::NW), KBLds<T>::BYTES, stream, (const T*)H,         \
                           (const T*)Wa, ba, wb, scores, ml, M, dM, (T*)dT, dba, dwb, dbb, part_ws, A_out, B, N, chunk, S, isn); \
    }
    if (dtype == MURCL_DTYPE_BF16) {
        if (exact_tanh) { if (wfrag) KB_LAUNCH(bf16_t, true, true) else KB_LAUNCH(bf16_t, true, false) }
        else { if (wfrag) KB_LAUNCH(bf16_t, false, true) else KB_LAUNCH(bf16_t, false, false) }
    } else if (dtype == MURCL_DTYPE_F32) {
        if (exact_tanh) KB_LAUNCH(float, true, false) else KB_LAUNCH(float, false, false)
    } else {
        return -1;
    }
#undef KB_LAUNCH
    int rc = MURCL_CHECK_LAUNCH();
    if (rc) return rc;
    hipLaunchKernelGGL(abmil_pool_bwd_reduce_kernel, dim3((2 * K2_D + 1 + 15) / 16), dim3(256), 0, stream, part_ws, grid, dba,
                       dwb, dbb);
    return MURCL_CHECK_LAUNCH();
}
