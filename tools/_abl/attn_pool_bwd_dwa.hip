// K2 backward with the attention weight gradient folded in (bf16 throughput path, L = 512, D = 128):
// autograd of models/abmil.py:23-27,38-42 w.r.t. the pre-tanh activations AND attention.0.weight in ONE pass over H.
//
//   g_n  = dM.H_n / sqrt(N)        c = dM.M        ds_n = p_n (g_n - c)
//   dT[n,d] = ds_n wb[d] (1 - t[n,d]^2),  t = tanh(Wa H_n + ba)          (recomputed on the matrix cores)
//   dWa[d,l] += sum_n dT[n,d] H[n,l]                                      (the product murcl_gemm_tn(dT, H) used to form
//                                                                          in a second pass over H and dT: 335 MB, 67 us at C2)
//   dba += sum_n dT[n,:]   dwb += sum_n ds_n t[n,:]   dbb += sum_n ds_n
//
// Shape of the kernel.  The accumulator of dWa is 128 x 512 f32 = 256 KiB per workgroup: all 256 accumulation registers of
// four waves.  So ONE 4-wave workgroup per CU (one wave per SIMD, 512 registers per lane: 128 of Wa fragments, 256 of dWa),
// 16-row H tiles stream HBM -> LDS by LDS-DMA in PAIRS (32 rows: one k-step of v_mfma_f32_16x16x32_bf16 for the dWa product)
// through a ring of four pair slots (three pairs = 96 KiB in flight per CU).  Wave w owns the 32 columns d = 32w .. 32w+31.
//
// Orientation.  The pre-activation product runs as C[row][d] = H Wa^T (A operand = H rows, B operand = the wave's Wa
// fragments), so a lane holds dT for ONE column d and the rows 4q .. 4q+3 of both tiles of the pair: exactly the B-operand
// fragment of the following product, which sums over the rows (cdna_hip_programming.md, "an accumulator tile as the next MFMA's
// operand").  Its A operand - H with the rows as k - is the LDS tile read k-major by ds_read_b64_tr_b16, as the forward kernel's
// pooling does.  dT never crosses lanes or LDS on its way into dWa; it goes through a wave-private LDS transpose only for its
// row-major store to HBM (the rank-1 dgrad reads it: panel_gemm.hip RANK1_MASK).
//
// Every workgroup leaves its partial dWa [128][512] (and its 2D+1 partial sums of dba / dwb / dbb) in a workspace;
// abmil_pool_bwd_dwa_reduce_kernel adds the rows up in a fixed order (no float atomics: bit-reproducible run to run).
#include "k2_common.h"

namespace kd {
constexpr int NW = 4;                                   // waves per workgroup, DW = 32 columns of D each
constexpr int NJ = 2;                                   // 16-column MFMA tiles per wave
constexpr int NKK = 16;                                 // k-steps (32 elements) of a 512-element row
constexpr int KW = NKK / NW;                            // k-steps of the g dot owned by one wave
constexpr int PADB = K2<bf16_t>::PADB;                  // 1040: LDS row stride (k2_common.h)
constexpr int SLOT = K2<bf16_t>::SLOT;                  // one 16-row tile
constexpr int PAIR = 2 * SLOT;
constexpr int NPAIR = 4;                                // ring slots (pairs): one in use + three in flight
constexpr int NLT = K2_L / 16;                          // 32 column tiles of H / dWa
constexpr int GPAR_F = 2 * NW * 16 + 4;                 // floats per parity: [2 tiles][NW][16] partial g rows + (c, m, 1/l, -) of the pair's bag
constexpr int OFF_GPART = NPAIR * PAIR;                 // [2 parities][GPAR_F] f32
constexpr int OFF_SC = OFF_GPART + 2 * GPAR_F * 4;     // [pair slot][NW][64] f32 (each wave's copy of the pair's 32 saved scores)
constexpr int OFF_STAGE = OFF_SC + NPAIR * NW * 256;    // [NW][32 rows][32 d] bf16: wave-private transpose of the pair's dT
constexpr int STAGE_W = 32 * 64;
constexpr int OFF_DMF = OFF_STAGE + NW * STAGE_W;       // [NW][1 KiB]: the wave's dM fragments (hi / lo columns) + 512 B of zeros
#ifdef KD_STAMPS
constexpr int OFF_STAMPS = OFF_DMF + NW * 1024;
constexpr int BYTES = OFF_STAMPS + 2 * 40 * 12 * 4;
#else
constexpr int BYTES = OFF_DMF + NW * 1024;              // 150,016 B
#endif
constexpr int NOPS = 9;                                 // LDS-DMA instructions per wave per pair (2 x 4 tile pieces + scores)
constexpr int NST = 2;                                  // dT stores per lane per pair
constexpr int PART_W = 2 * K2_D + 1;
}  // namespace kd

// The pre-activation and g products through inline assembly with VGPR destinations.  At one wave per SIMD hipcc selects the
// accumulation-register form for EVERY MFMA of a kernel; dWa alone fills all 256 of them, and the register allocator then parks
// dWa tuples in VGPRs around the third phase (156 v_accvgpr moves per pair in the MFMA stream).  With these two products kept
// in VGPRs the accumulation file holds dWa and nothing else.  hipcc pads no hazards inside asm: kd_mfma_settle covers the one
// that applies here - a VALU read of an MFMA result needs the MFMA's passes + 3 wait states behind its issue (the operands
// come from LDS reads / registers written long before, dependent accumulation on the same tuple is interlocked).
// LDS-DMA pieces without the save / restore of M0 around them (common.h glds16_u): nothing else in this kernel reads M0, and the
// `s_mov m0, <saved>` behind a global_load_lds has to wait until that instruction has left the wave with its M0 (KD_M0=0: the
// common.h forms, for A/B)
#ifndef KD_M0
#define KD_M0 1
#endif
__device__ __forceinline__ void kd_glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    if (KD_M0) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
    else glds16_u(sbase, voff, lds_dst);
}
__device__ __forceinline__ void kd_glds4(const void* sbase, unsigned voff, unsigned lds_dst) {
    if (KD_M0) {
        const unsigned long long b = (unsigned long long)sbase;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        const unsigned long long bs = ((unsigned long long)hi << 32) | lo;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(bs), "s"(dst) : "memory", "m0");
    } else glds4_s(sbase, voff, lds_dst);
}
// one LDS-DMA piece (j of 4) of a 16-row bf16 tile, rows past N clamped (k2_common.h k2_issue_piece with the helper above)
__device__ __forceinline__ void kd_piece(const bf16_t* bag_base, int row0, int N, unsigned slot_lds, int wave, int lane, int j) {
    const int row = j * kd::NW + wave;
    const int grow = min(row0 + row, N - 1);
    kd_glds16((const char*)bag_base + (size_t)grow * 1024, lane * 16, slot_lds + row * kd::PADB);
}
__device__ __forceinline__ void kd_mfma0(f32x4& c, bf16x8 a, bf16x8 b) {         // c = a b
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void kd_mfma(f32x4& c, bf16x8 a, bf16x8 b) {          // c += a b
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void kd_mfma_settle(f32x4& c0, f32x4& c1) {
    asm volatile("s_nop 15" : "+v"(c0), "+v"(c1));
}

#ifndef KD_GK
#define KD_GK 2                   // k-steps per LDS prefetch group of the pre-activation product
#endif
#ifndef KD_NB
#define KD_NB 3                   // fragment buffers of that product: groups g .. g + KD_NB - 2 are in flight behind the one in use
#endif
#ifndef KD_GL
#define KD_GL 2                   // column tiles per LDS prefetch group of the dWa product
#endif
#ifndef KD_NB3
#define KD_NB3 3                  // fragment buffers of the dWa product
#endif
#ifndef KD_ABL
#define KD_ABL 0                  // diagnostic builds (tools/ab_build.py): 1 no dWa product, 2 no pre-activation product, 4 no tanh math, 8 no dT store
#endif

// In-kernel stamps (diagnostic builds only, -DKD_STAMPS; tools/stamps_kd.py): waves 0 and 2 of the first KD_STAMP_WG workgroups
// note s_memtime at the phase boundaries of each pair iteration into a spare LDS block and copy it out at the end.
#ifdef KD_STAMPS
#define KD_STAMP_WG 16
#define KD_STAMP_IT 40
#define KD_STAMP_EV 12
__device__ unsigned g_kd_stamps[KD_STAMP_WG][2][KD_STAMP_IT][KD_STAMP_EV];
extern "C" int murcl_debug_kd_stamps(void* host, long bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_kd_stamps), (size_t)bytes, 0, hipMemcpyDeviceToHost);
}
#define KD_STAMP(ev)                                                                                              \
    do {                                                                                                          \
        if (stamp_w >= 0 && pr < KD_STAMP_IT) {                                                                   \
            const unsigned long long t_ = ((ev) == 11) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
            if (lane == 0) stamps[(stamp_w * KD_STAMP_IT + pr) * KD_STAMP_EV + (ev)] = (unsigned)t_;             \
        }                                                                                                         \
    } while (0)
#else
#define KD_STAMP(ev)
#endif

template <bool EXACT_TANH>
__global__ __launch_bounds__(64 * kd::NW, 1) void abmil_pool_bwd_dwa_kernel(
    const bf16_t* __restrict__ H, const bf16_t* __restrict__ Wa, const float* __restrict__ ba, const float* __restrict__ wb,
    const float* __restrict__ scores, const float* __restrict__ ml, const float* __restrict__ Mp,
    const float* __restrict__ dM, bf16_t* __restrict__ dT, float* __restrict__ part_ws, float* __restrict__ dwa_ws, int B,
    int N, int chunk_rows, int S, float inv_sqrt_n) {
    using namespace kd;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q4 = lane >> 4, c16 = lane & 15;
    const unsigned lds0 = lds_off(smem);
#ifdef KD_STAMPS
    const unsigned long long kt0 = __builtin_amdgcn_s_memtime();
#endif
    float* gpart = (float*)(smem + OFF_GPART);
    const float* scb = (const float*)(smem + OFF_SC);
    char* stage = smem + OFF_STAGE + wave * STAGE_W;

    const int n_items = B * S;
    const int pairs_per_item = chunk_rows / 32;
    // a workgroup walks a CONTIGUOUS run of items: consecutive chunks of a bag share its constants (dM fragments, c, m, l: rebuilt
    // by compiler-visible loads that drain the LDS-DMA queue), and at the headline shape a workgroup stays inside one bag
    const int it_base = n_items / (int)gridDim.x, it_rem = n_items % (int)gridDim.x;
    const int my_items = it_base + ((int)blockIdx.x < it_rem ? 1 : 0);
    const int first_item = (int)blockIdx.x * it_base + min((int)blockIdx.x, it_rem);
    const int my_pairs = my_items * pairs_per_item;      // >= 1: the launcher never starts more workgroups than items

    K2Pos ip, gp, cp;                                    // issue / g-dot / compute positions; `tin` counts PAIRS here
    ip.init(first_item, S);
    gp.init(first_item, S);
    cp.init(first_item, S);
    // One pair = 9 LDS-DMA instructions per wave, always issued back to back at a point where the wave has nothing in flight:
    // in the middle of an MFMA / ds_read stream a single piece stalls the wave for 200-500 cycles, nine in a row for ~600
    // (in-kernel stamps, tools/stamps_kd.py)
    auto issue = [&](int seq) {
        const int row0 = ip.ch * chunk_rows + ip.tin * 32;
        const int ps = seq & (NPAIR - 1);
        const bf16_t* bagb = H + (size_t)ip.bag * N * K2_L;
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) kd_piece(bagb, row0 + 16 * (idx >> 2), N, lds0 + ps * PAIR + (idx >> 2) * SLOT, wave, lane, idx & 3);
        // ninth op: this wave's private copy of the pair's saved scores (lane r <-> row r of the pair, clamped; lanes 32.. repeat)
        kd_glds4(scores + (size_t)ip.bag * N, (unsigned)min(row0 + (lane & 31), N - 1) * 4u, lds0 + OFF_SC + (ps * NW + wave) * 256);
        ip.next(pairs_per_item, 1, S);
    };
    // the first two pairs are requested before anything else: their HBM latency runs under the weight prologue
    for (int s = 0; s < min(2, my_pairs); ++s) issue(s);

    // ---- weight prologue (as attn_pool.hip): the wave's 32 rows of Wa travel as whole 1 KiB LDS-DMA pieces through the still
    // empty half of the ring (pair slots 2, 3) and come back as MFMA fragments: lane (q4, c16) of k-step kk holds
    // Wa[d = 32 wave + 16 j + c16][the 16-byte chunk kk + 16 q4] - the B operand of C[row][d] = H Wa^T (any k assignment works as
    // long as H uses the same)
    bf16x8 wa[NJ][NKK];
    float ba_r[NJ], wb_r[NJ], dba_r[NJ], dwb_r[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const char* wblk = (const char*)(Wa + (size_t)(32 * wave + 16 * j) * K2_L);
        const unsigned stg = lds0 + 2 * PAIR + wave * 16 * PADB;
#pragma unroll
        for (int u = 0; u < 16; ++u) glds16_u(wblk + (size_t)u * 1024, lane * 16, stg + u * PADB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const char* fb = smem + 2 * PAIR + (wave * 16 + c16) * PADB + NKK * q4 * 16;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            wa[j][kk] = *(const bf16x8*)(fb + kk * 16);
            asm volatile("" : "+v"(wa[j][kk]));      // resident for the whole launch
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the next block's pieces overwrite these rows
        ba_r[j] = ba[32 * wave + 16 * j + c16];
        wb_r[j] = wb[32 * wave + 16 * j + c16];
        dba_r[j] = 0.f;
        dwb_r[j] = 0.f;
    }
    float dbb_acc = 0.f;
    f32x4 dwa[NLT][NJ];                                  // dWa^T tile (lt, j): [l = 16 lt + 4 q4 + r][d = 32 wave + 16 j + c16]
#pragma unroll
    for (int lt = 0; lt < NLT; ++lt)
#pragma unroll
        for (int j = 0; j < NJ; ++j) dwa[lt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // dM enters the g dot as columns 0 (hi bf16 part) and 1 (lo part) of a B operand, zeros in columns 2..15.  The fragments of
    // this wave's KW k-steps live in its LDS block (a new bag rewrites them): lanes of columns 0 / 1 read [k-step][column][q4],
    // every other lane the block's zero half
    char* dmblk = smem + OFF_DMF + wave * 1024;
    const char* dmrd = dmblk + (c16 < 2 ? (c16 * 4 + q4) * 16 : 512);
    for (int i = lane; i < 128; i += 64) ((unsigned*)(dmblk + 512))[i] = 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (pairs 0 and 1 landed with the weights: older, same counter)
    LDS_BARRIER();                                       // every wave has read its fragments back: slots 2, 3 are free for tiles
    if (my_pairs > 2) issue(2);

    // ---- the g dot runs ONE PAIR AHEAD of the rest (it needs H and dM only): g = H.dM of pair p+1 is formed at the top of
    // iteration p - every wave its k-quarter, the partial rows parked in gpart[(p+1) & 1] with the bag's constants -
    // and the top barrier of iteration p+1 publishes them.  The pre-activations of a tile then meet their ds without a barrier
    // in between (the 256 accumulation registers hold dWa, 128 of the others Wa: two tiles' accumulators across a barrier do not fit)
    int g_bag = -1;
    float gb_c = 0.f, gb_m = 0.f, gb_invl = 0.f;
    auto g_bag_constants = [&]() {
        const int bag = gp.bag;
        if (bag == g_bag) return;
        g_bag = bag;                                     // new bag: compiler-visible loads (they drain the LDS-DMA queue: rare)
        const float* dmb = dM + (size_t)bag * K2_L;
        const float* mb = Mp + (size_t)bag * K2_L;
        float cpart = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) cpart += dmb[8 * lane + e] * mb[8 * lane + e];
        gb_c = wave_sum(cpart);
        gb_m = ml[2 * bag];
        gb_invl = 1.0f / ml[2 * bag + 1];
        if (c16 < 2) {
#pragma unroll
            for (int i = 0; i < KW; ++i) {
                const int chunk = (KW * wave + i) + NKK * q4;
                bf16x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = dmb[chunk * 8 + e];
                    const bf16_t hi = f2bf(v);
                    f[e] = (short)(c16 == 0 ? hi : f2bf(v - bf2f(hi)));
                }
                *(bf16x8*)(dmblk + i * 128 + (c16 * 4 + q4) * 16) = f;
            }
        }
        asm volatile("" ::: "memory");
    };
    // the g dot of pair `seq`: 3 KW fragment reads, 2 KW MFMAs, the hand-off of the rows - on its own behind the iteration's
    // LDS-DMA burst.  (Riding in the k-groups of the first product its operands + accumulators cost 22 spilled registers; with
    // the reads requested ahead of the burst hipcc parks all of them in scratch around it.)
    f32x4 ga[2];
    bf16x8 gop[KW][3];
    auto g_load = [&](int seq) {
        const char* pb = smem + (seq & (NPAIR - 1)) * PAIR + c16 * PADB + (KW * wave + NKK * q4) * 16;
#pragma unroll
        for (int i = 0; i < KW; ++i) {
            gop[i][0] = *(const bf16x8*)(dmrd + i * 128);
            gop[i][1] = *(const bf16x8*)(pb + i * 16);
            gop[i][2] = *(const bf16x8*)(pb + SLOT + i * 16);
        }
    };
    auto g_finish = [&](int seq) {
#pragma unroll
        for (int i = 0; i < KW; ++i) {
            if (i == 0) { kd_mfma0(ga[0], gop[i][1], gop[i][0]); kd_mfma0(ga[1], gop[i][2], gop[i][0]); }
            else { kd_mfma(ga[0], gop[i][1], gop[i][0]); kd_mfma(ga[1], gop[i][2], gop[i][0]); }
        }
        kd_mfma_settle(ga[0], ga[1]);
        // column 0 of a g tile holds the hi parts, column 1 the lo parts: rows 4 q4 .. +3
        float* gp_out = gpart + (seq & 1) * GPAR_F;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 gs;
#pragma unroll
            for (int r = 0; r < 4; ++r) gs[r] = ga[u][r] + dpp_mov<0xB1>(ga[u][r]);
            if (c16 == 0) *(f32x4*)(gp_out + (u * NW + wave) * 16 + 4 * q4) = gs;
        }
        if (tid == 0) *(f32x4*)(gp_out + 2 * NW * 16) = f32x4{gb_c, gb_m, gb_invl, 0.f};
        gp.next(pairs_per_item, 1, S);
    };
    g_bag_constants();
    g_load(0);
    g_finish(0);

    const unsigned ptoff = (4 * q4 + (c16 >> 2)) * PADB + 4 * (c16 & 3) * 2;      // transposing 4 x 16 read (T10): row, 4-column group

#ifdef KD_STAMPS
    unsigned* stamps = (unsigned*)(smem + OFF_STAMPS);
    const int stamp_w = (blockIdx.x < KD_STAMP_WG) ? (wave == 0 ? 0 : (wave == 2 ? 1 : -1)) : -1;
    for (int i = tid; i < 2 * KD_STAMP_IT * KD_STAMP_EV; i += 64 * NW) stamps[i] = 0u;
    const unsigned long long kt1 = __builtin_amdgcn_s_memtime();
#endif
    for (int pr = 0; pr < my_pairs; ++pr) {
        KD_STAMP(0);
        KD_STAMP(11);
        // Pair pr+1 must have landed: its g dot rides in the first product below.  vmcnt counts LDS-DMA pieces and the dT stores
        // together, in issue order; younger than pair pr+1's pieces are pair pr+2's and the stores of the (at most two) iterations
        // since it was issued (pairs 0..2 go out in the prologue)
        const bool g_next = pr + 1 < my_pairs;
        if (pr + 2 < my_pairs) {
            if (pr == 0) { WAIT_VMCNT(9); } else if (pr == 1) { WAIT_VMCNT(11); } else { WAIT_VMCNT(13); }
        } else {
            WAIT_VMCNT(0);
        }
        static_assert(NOPS == 9 && NST == 2, "the waits above are NOPS + min(pr, 2) NST");
        LDS_BARRIER();                                   // pair pr+1 landed for every wave, gpart[pr & 1] is complete, slot pr-1 is free
        KD_STAMP(1);
        if (pr + NPAIR - 1 < my_pairs) issue(pr + NPAIR - 1);
        KD_STAMP(2);
        if (g_next) { g_bag_constants(); g_load(pr + 1); g_finish(pr + 1); }

        const int bag = cp.bag;
        const int row0 = cp.ch * chunk_rows + cp.tin * 32;
        const int ps = pr & (NPAIR - 1);
        const char* pair = smem + ps * PAIR;
        const float* gin = gpart + (pr & 1) * GPAR_F;

        // ds of this lane's rows (4 q4 .. +3 of a tile): needs the published g rows, the saved scores and the bag's constants only
        float ds[2][4];
        auto ds_tile = [&](int u) {
            f32x4 gq = *(const f32x4*)(gin + (u * NW + 0) * 16 + 4 * q4);
#pragma unroll
            for (int w = 1; w < NW; ++w) gq += *(const f32x4*)(gin + (u * NW + w) * 16 + 4 * q4);
            const f32x4 scq = *(const f32x4*)(scb + (ps * NW + wave) * 64 + 16 * u + 4 * q4);
            const f32x4 bagc = *(const f32x4*)(gin + 2 * NW * 16);   // (c, m, 1 / l) of this pair's bag
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = (EXACT_TANH ? __expf(scq[r] - bagc[1]) : fast_exp(scq[r] - bagc[1])) * bagc[2];
                ds[u][r] = (row0 + 16 * u + 4 * q4 + r < N) ? p * (gq[r] * inv_sqrt_n - bagc[0]) : 0.f;
                dbb_acc += ds[u][r];
            }
        };

        // the pre-activation product of one tile for this wave's 32 columns of D (two accumulators alternate); the fragments of
        // k-group g + NB - 1 are requested before the MFMAs of group g issue (one wave per SIMD: nothing else covers an LDS round
        // trip); hook(g) runs behind the MFMAs of group g: the vector work of the OTHER tile rides in the shadow of the matrix work
        f32x4 acc[2][NJ];
        auto product = [&](int u, auto&& hook) {
            const char* hb = pair + u * SLOT + c16 * PADB + NKK * q4 * 16;
            constexpr int GK = KD_GK, NG = NKK / GK, NB = KD_NB;
            bf16x8 hq[NB][GK];
#pragma unroll
            for (int b = 0; b < NB - 1; ++b)
#pragma unroll
                for (int k2 = 0; k2 < GK; ++k2) hq[b][k2] = *(const bf16x8*)(hb + (b * GK + k2) * 16);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + NB - 1 < NG) {
#pragma unroll
                    for (int k2 = 0; k2 < GK; ++k2) hq[(g + NB - 1) % NB][k2] = *(const bf16x8*)(hb + ((g + NB - 1) * GK + k2) * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k2 = 0; k2 < GK; ++k2)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if ((KD_ABL & 2) && !(g == 0 && k2 == 0)) continue;
                        if (g == 0 && k2 == 0) kd_mfma0(acc[u][j], hq[g % NB][k2], wa[j][g * GK + k2]);
                        else kd_mfma(acc[u][j], hq[g % NB][k2], wa[j][g * GK + k2]);
                    }
                hook(g);
                __builtin_amdgcn_sched_barrier(0);
            }
            kd_mfma_settle(acc[u][0], acc[u][1]);
        };
        // dT of one (tile, column group, row) element; pairs of rows are packed and parked in the wave's transpose block
        u32x4 pk[NJ];
        float o_even = 0.f;
        auto element = [&](int u, int j, int r) {
            const float x = acc[u][j][r] + ba_r[j];
            const float t = (KD_ABL & 4) ? x : (EXACT_TANH ? tanhf(x) : fast_tanh(x));
            const float o = ds[u][r] * wb_r[j] * (1.f - t * t);
            dba_r[j] += o;
            dwb_r[j] += ds[u][r] * t;
            if (!(r & 1)) { o_even = o; return; }
            const unsigned w = pack_bf2(o_even, o);
            pk[j][2 * u + (r >> 1)] = w;
            // wave-private transpose for the row-major store: element (row 16 u + 4 q4 + r, column 16 j + c16)
            char* sp = stage + (16 * u + 4 * q4 + (r & 2)) * 64 + (16 * j + c16) * 2;
            *(uint16_t*)(sp) = (uint16_t)w;
            *(uint16_t*)(sp + 64) = (uint16_t)(w >> 16);
        };
        static_assert(NKK / KD_GK == 8, "the hooks below assume eight k-groups per product");
        product(0, [&](int g) {
            if (g == 7) ds_tile(0);
        });
        KD_STAMP(3);
        product(1, [&](int g) {
            element(0, g >> 2, g & 3);                       // tile 0's eight elements, one per k-group of tile 1
            if (g == 7) ds_tile(1);
        });
        KD_STAMP(4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 8; ++e) element(1, e >> 2, e & 3);
        KD_STAMP(5);
        // k order of the fragments: elements 0..3 = rows 4 q4 .. +3 of tile 0, elements 4..7 = the same rows of tile 1
        bf16x8 dtf[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) dtf[j] = __builtin_bit_cast(bf16x8, pk[j]);
        __builtin_amdgcn_sched_barrier(0);               // (phase boundaries are scheduling fences: overlapped phases do not fit the register file)
        asm volatile("" ::: "memory");                   // the 16-byte reads below alias the 2-byte stores above
        {
            // row (lane >> 2) of tile i, 8 columns: 64 contiguous bytes per row and wave, as the unfused kernel stores them;
            // rows past N go to the 32 spare rows behind the last bag (never read)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32x4 v = *(const u32x4*)(stage + (16 * i + (lane >> 2)) * 64 + (lane & 3) * 16);
                const int prow = 16 * i + (lane >> 2), grow = row0 + prow;
                bf16_t* dst = dT + ((grow < N) ? ((size_t)bag * N + grow) : ((size_t)B * N + prow)) * K2_D + 32 * wave + 8 * (lane & 3);
                if (KD_ABL & 8) dst = dT + ((size_t)B * N + prow) * K2_D + 32 * wave + 8 * (lane & 3);
                *(u32x4*)dst = v;
            }
        }
        KD_STAMP(6);

        KD_STAMP(7);
        KD_STAMP(8);

        // ---- dWa^T[l][d] += sum over the pair's 32 rows of H[row][l] dT[row][d]; H read k-major from the LDS tiles, the
        // fragments of column group g + NB3 - 1 requested before the MFMAs of group g
        __builtin_amdgcn_sched_barrier(0);
        if (!(KD_ABL & 1)) {
            const char* tb0 = pair + ptoff;
            const char* tb1 = tb0 + SLOT;
            constexpr int GL = KD_GL, NGL = NLT / GL, NB3 = KD_NB3;
            s16x4 ha[NB3][GL], hb[NB3][GL];
#define KD_TR(p) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p))
#pragma unroll
            for (int b = 0; b < NB3 - 1; ++b)
#pragma unroll
                for (int t = 0; t < GL; ++t) { ha[b][t] = KD_TR(tb0 + (b * GL + t) * 32); hb[b][t] = KD_TR(tb1 + (b * GL + t) * 32); }
#pragma unroll
            for (int g = 0; g < NGL; ++g) {
                if (g + NB3 - 1 < NGL) {
#pragma unroll
                    for (int t = 0; t < GL; ++t) {
                        ha[(g + NB3 - 1) % NB3][t] = KD_TR(tb0 + ((g + NB3 - 1) * GL + t) * 32);
                        hb[(g + NB3 - 1) % NB3][t] = KD_TR(tb1 + ((g + NB3 - 1) * GL + t) * 32);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < GL; ++t) {
                    const s16x4 a = ha[g % NB3][t], b = hb[g % NB3][t];
                    const bf16x8 hT = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        dwa[g * GL + t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hT, dtf[j], dwa[g * GL + t][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#undef KD_TR
        }
        cp.next(pairs_per_item, 1, S);
        KD_STAMP(9);
        KD_STAMP(10);
    }
#ifdef KD_STAMPS
    const unsigned long long kt2 = __builtin_amdgcn_s_memtime();
#endif

    // ---- publish: the parameter-gradient partial sums (row of part_ws) and this workgroup's partial dWa [128][512]
    float* prow = part_ws + (size_t)blockIdx.x * PART_W;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const float a = quarters_sum(dba_r[j]), w = quarters_sum(dwb_r[j]);
        if (q4 == 0) {
            prow[32 * wave + 16 * j + c16] = a;
            prow[K2_D + 32 * wave + 16 * j + c16] = w;
        }
    }
    if (wave == 0) {
        const float t = quarters_sum(dbb_acc);           // every lane of a quarter summed the same rows
        if (lane == 0) prow[2 * K2_D] = t;
    }
    float* wrow = dwa_ws + (size_t)blockIdx.x * (K2_D * K2_L);
#pragma unroll
    for (int lt = 0; lt < NLT; ++lt)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            *(f32x4*)(wrow + (size_t)(32 * wave + 16 * j + c16) * K2_L + 16 * lt + 4 * q4) = dwa[lt][j];
#ifdef KD_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long kt3 = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
        unsigned* kr = stamps + (KD_STAMP_IT - 1) * KD_STAMP_EV;
        kr[0] = (unsigned)kt0; kr[1] = (unsigned)kt1; kr[2] = (unsigned)kt2; kr[3] = (unsigned)kt3;
    }
    __syncthreads();
    if (blockIdx.x < KD_STAMP_WG)
        for (int i = tid; i < 2 * KD_STAMP_IT * KD_STAMP_EV; i += 64 * NW) (&g_kd_stamps[blockIdx.x][0][0][0])[i] = stamps[i];
#endif
}

// Blocks 0 .. NSMALL-1: dba[c] += sum_w part[w][c], dwb[c] += sum_w part[w][D + c], dbb += sum_w part[w][2D] (as
// abmil_pool_bwd_reduce_kernel).  Blocks NSMALL ..: dWa (+)= sum_w dwa_ws[w]: a block owns 64 float4 columns, its four thread
// groups each walk a quarter of the workgroup rows (four 16-byte loads in flight), and meet through LDS in a fixed order.
#define KD_NSMALL ((kd::PART_W + 15) / 16)
__global__ __launch_bounds__(256) void abmil_pool_bwd_dwa_reduce_kernel(const float* __restrict__ part, const float* __restrict__ dwa_ws,
                                                                        int n_wg, float* __restrict__ dba, float* __restrict__ dwb,
                                                                        float* __restrict__ dbb, float* __restrict__ dWa, int accumulate) {
    __shared__ f32x4 red4[4][64];
    if ((int)blockIdx.x < KD_NSMALL) {
        float (*red)[17] = (float (*)[17])red4;
        const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
        const int c = blockIdx.x * 16 + cl, W = kd::PART_W;
        float s = 0.f;
        if (c < W) {
            float t4[4] = {0.f, 0.f, 0.f, 0.f};
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
        return;
    }
    const int blk = blockIdx.x - KD_NSMALL;              // 256 blocks x 64 float4 = 128 x 512 floats
    const int c4 = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const f32x4* src = (const f32x4*)dwa_ws + (size_t)blk * 64 + c4;
    constexpr size_t STRIDE4 = (size_t)K2_D * K2_L / 4;
    f32x4 t4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t4[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    int w = grp;
    for (; w + 12 < n_wg; w += 16)
#pragma unroll
        for (int u = 0; u < 4; ++u) t4[u] += src[(size_t)(w + 4 * u) * STRIDE4];
    for (; w < n_wg; w += 4) t4[0] += src[(size_t)w * STRIDE4];
    red4[grp][c4] = (t4[0] + t4[1]) + (t4[2] + t4[3]);
    __syncthreads();
    if (grp == 0) {
        const f32x4 t = (red4[0][c4] + red4[1][c4]) + (red4[2][c4] + red4[3][c4]);
        f32x4* dst = (f32x4*)dWa + (size_t)blk * 64 + c4;
        *dst = accumulate ? *dst + t : t;
    }
}

extern "C" int murcl_abmil_pool_workspace(int B, int N, int dtype, int* chunk_rows, int* n_chunks);

static int kd_grid(int B, int N, int* chunk, int* S) {
    murcl_abmil_pool_workspace(B, N, MURCL_DTYPE_BF16, chunk, S);
    const int items = B * *S;
    const int cap = murcl_cu_budget();
    return items < cap ? items : cap;
}

// C-ABI: see include/murcl_amd.h
extern "C" long murcl_abmil_pool_bwd_dwa_ws_floats(int B, int N, int L, int D, int dtype) {
    if (L != K2_L || D != K2_D || dtype != MURCL_DTYPE_BF16 || B <= 0 || N <= 0) return 0;
    int chunk, S;
    const int grid = kd_grid(B, N, &chunk, &S);
    if (chunk % 32) return 0;
    return (long)grid * (kd::PART_W + K2_D * K2_L) + 4;      // + the 16-byte alignment of the partial dWa rows
}

extern "C" int murcl_abmil_pool_bwd_dwa(const void* H, const void* Wa, const float* ba, const float* wb, const float* scores,
                                        const float* ml, const float* M, const float* dM, void* dT, float* dba, float* dwb,
                                        float* dbb, float* dWa, int dwa_accumulate, float* ws, long ws_floats, int B, int N,
                                        int L, int D, int dtype, int exact_tanh, hipStream_t stream) {
    const long need = murcl_abmil_pool_bwd_dwa_ws_floats(B, N, L, D, dtype);
    if (!need || !ws || ws_floats < need || !dWa) return -1;
    int chunk, S;
    const int grid = kd_grid(B, N, &chunk, &S);
    float* part = ws;
    float* dwa_ws = ws + (size_t)grid * kd::PART_W;
    // (the partial dWa rows start 16-byte aligned: the caller's workspace is, and grid * 257 floats is rounded up here)
    dwa_ws = (float*)(((uintptr_t)dwa_ws + 15) & ~(uintptr_t)15);
    if ((dwa_ws - ws) + (long)grid * K2_D * K2_L > ws_floats) return -1;
    const float isn = 1.0f / sqrtf((float)N);
#define KD_LAUNCH(EX)                                                                                              \
    {                                                                                                              \
        auto k = abmil_pool_bwd_dwa_kernel<EX>;                                                                    \
        static MurclOncePerDevice once;                                                                            \
        if (once.first()) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kd::BYTES); \
        hipLaunchKernelGGL(k, dim3(grid), dim3(64 * kd::NW), kd::BYTES, stream, (const bf16_t*)H, (const bf16_t*)Wa, ba, wb, \
                           scores, ml, M, dM, (bf16_t*)dT, part, dwa_ws, B, N, chunk, S, isn);                      \
    }
    if (exact_tanh) KD_LAUNCH(true) else KD_LAUNCH(false)
#undef KD_LAUNCH
    int rc = MURCL_CHECK_LAUNCH();
    if (rc) return rc;
    hipLaunchKernelGGL(abmil_pool_bwd_dwa_reduce_kernel, dim3(KD_NSMALL + 256), dim3(256), 0, stream, part, dwa_ws, grid, dba, dwb,
                       dbb, dWa, dwa_accumulate);
    return MURCL_CHECK_LAUNCH();
}
