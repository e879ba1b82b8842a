// K2: ABMIL attention pooling over bags of encoded patch features (models/abmil.py:38-42).
//
//   s_n = wb . tanh(Wa H_n + ba) + bb ;  A = softmax_N(s) / sqrt(N) ;  M = A . H
//
// One pass over H.  bf16 (throughput path): persistent 4-wave workgroups, TWO per CU - they desynchronise naturally, so one
// workgroup's barriers and LDS latencies are covered by the other's MFMA / VALU work - walk (bag, row-chunk) items from the last
// bag to the first (what the encoder wrote last is still in the Infinity Cache); H row tiles of 16 rows stream HBM -> LDS by
// LDS-DMA into a 4-slot ring and are consumed two at a time (the pair in use + the next pair in flight); each wave keeps its
// 32-column slice of Wa in registers as MFMA operands for the whole launch (fetched once as whole rows through the still empty
// ring, K2_WPRO), so per tile the matrix cores see only LDS reads of H.  Scores are reduced across waves through LDS, the soft-max
// uses a fixed reference (tanh bounds every score) or the online running max, and the weighted sum M = p.H runs on the matrix
// cores too (v_mfma_f32_16x16x16_bf16, p as hi + lo bf16 rows, H read k-major by ds_read_b64_tr_b16).  Per-chunk partials
// (m, l, sum p.H) are merged per bag by abmil_pool_combine_kernel.  f32 (parity path): one 8-wave workgroup per CU, exact tanh.
//
// LDS row image: rows are stored whole at a stride of ROWB+16 bytes (each LDS-DMA instruction writes
// 1 KiB inside one row, so the pad is free): 16 lanes reading the same 16-byte chunk of 16 consecutive
// rows hit 16 different bank slots, every fragment address is base + immediate, and the DMA source is
// plain row-major (wave-uniform SGPR base + lane*16).
#include "common.h"

#include "k2_common.h"

#ifndef K2_FWD_NW
#define K2_FWD_NW 0            // waves per workgroup of the bf16 forward (0 = 4; 8 = four waves per SIMD, A/B)
#endif
#define K2_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define K2F_NWO(T) (sizeof(T) == 2 ? K2_FWD_NW : 0)
#ifndef K2_REVERSE
#define K2_REVERSE 1
#endif
#ifndef K2_GK
#define K2_GK 4                  // k-steps per LDS prefetch group in the score MFMA loop
#endif
#ifndef K2_SPREAD_DMA
#define K2_SPREAD_DMA 0          // 1: the next pair's LDS-DMA pieces go out one per k-group of the score MFMA loop (measured r03: neutral, 62.9 vs 62.3 us)
#endif
#ifndef K2_STAGGER
#define K2_STAGGER 0             // 1: tile 1's score MFMAs run beside tile 0's tanh, pooling B operands requested before the barrier (measured r03: neutral, 61.3-62.3 vs 61.6-62.8 us)
#endif                           // requested before the barrier that precedes the pooling (see the paired loop)
#ifndef K2_WPRO
#define K2_WPRO 1                // bf16 forward: the wave's Wa slice is fetched as whole 1 KiB rows (LDS-DMA into the not yet used ring)
#endif                           // and read back as fragments, instead of fragment-shaped global loads (see the prologue)
#ifndef K2_PAIR
#define K2_PAIR 1                // bf16 forward: two 16-row tiles per iteration (see the paired loop)
#endif

// In-kernel stamps (diagnostic builds only, -DK2_STAMPS; tools/stamps_k2.py): waves 0 and 2 of the first K2_STAMP_WG workgroups
// note s_memtime at fixed points of each pair iteration into the (unused in the paired loop) sbuf region of LDS and copy it
// out at the end.  No output value depends on a stamp.
#ifdef K2_STAMPS
#define K2_STAMP_WG 16
#define K2_STAMP_IT 16
#define K2_STAMP_EV 10
__device__ unsigned g_k2_stamps[K2_STAMP_WG][2][K2_STAMP_IT][K2_STAMP_EV];
extern "C" int murcl_debug_k2_stamps(void* host, long bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_k2_stamps), (size_t)bytes, 0, hipMemcpyDeviceToHost);
}
#define K2_STAMP(ev)                                                                                              \
    do {                                                                                                          \
        if (stamp_w >= 0 && pr < K2_STAMP_IT) {                                                                   \
            const unsigned long long t_ = ((ev) == 8) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
            if (lane == 0) ((unsigned*)sbuf)[(stamp_w * K2_STAMP_IT + pr) * K2_STAMP_EV + (ev)] = (unsigned)t_;   \
        }                                                                                                         \
    } while (0)
#else
#define K2_STAMP(ev)
#endif

template <typename T, bool EXACT_TANH, bool WFRAG = false>
__global__ __launch_bounds__((64 * K2<T, K2F_NWO(T)>::NW), (K2<T, K2F_NWO(T)>::MIN_WAVES)) void abmil_pool_fwd_kernel(
    const T* __restrict__ H, const T* __restrict__ Wa, const float* __restrict__ ba, const float* __restrict__ wb,
    const float* __restrict__ bb_p, float* __restrict__ scores, float* __restrict__ part, int B, int N,
    int chunk_rows, int S) {
    typedef K2<T, K2F_NWO(T)> C_;
    constexpr bool wfrag = WFRAG && sizeof(T) == 2;      // (a template parameter: as a run-time branch the two prologues' fragment registers met in phi nodes and spilled)
    typedef K2Lds<T, K2F_NWO(T)> L_;
    typedef typename WFrag<T>::type frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform -> scalar address math
    const int q4 = lane >> 4, r16 = lane & 15;
    const unsigned lds0 = lds_off(smem);
    float* spart = (float*)(smem + L_::OFF_SPART);
    unsigned* pbuf = (unsigned*)(smem + L_::OFF_PBUF) + wave * 32;   // wave-private (16 words per tile of a pair)
    float* sbuf = (float*)(smem + L_::OFF_SBUF);

    const int n_items = B * S;
    const int tiles_per_item = chunk_rows / C_::TR;
    const int my_items = (n_items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int my_tiles = my_items * tiles_per_item;
    if (my_tiles <= 0) return;

    K2Pos ip, cp;                                                    // issue / compute positions
    // items are walked from the LAST bag to the first: the encoder kernel that has just written H finished with the high
    // rows, so the start of this pass finds them in the Infinity Cache / L2 while the write-back of H is still draining
    ip.init(blockIdx.x, S, K2_REVERSE ? n_items - 1 : -1);
    cp.init(blockIdx.x, S, K2_REVERSE ? n_items - 1 : -1);
    auto issue = [&](int seq) {
        k2_issue_tile<T, K2F_NWO(T)>(H + (size_t)ip.bag * N * K2_L, ip.ch * chunk_rows + ip.tin * C_::TR, N,
                         lds0 + (seq & (K2_NSLOT - 1)) * C_::SLOT, wave, lane);
        ip.next(tiles_per_item, gridDim.x, S);
    };
    constexpr bool PAIR = K2_PAIR && sizeof(T) == 2 && !EXACT_TANH;
    const int pre = min(PAIR ? 4 : 3, my_tiles);
    // Weight prologue.  A fragment-shaped global load (16 rows x 4 pieces of 16 B, 256 B apart) touches 64 cache lines per
    // instruction: 32 such instructions per wave pull each line of the slice through the CU's memory pipe up to eight times -
    // about as many bytes as the workgroup's whole share of H.  WPRO: the wave's 16 rows per column block travel as 16 whole-row
    // LDS-DMA pieces into its private quarter of the (still empty) tile ring and come back as conflict-free ds_read_b128 fragments.
#ifndef K2_ABL_NOWEIGHTS
#define K2_ABL_NOWEIGHTS 0       // dev ablation (tools/ab_build.py): 1 = no weight prologue at all (constant fragments, wrong results): its time is
#endif                           // the upper bound of what any faster prologue can buy
    constexpr bool WPRO_C = !K2_ABL_NOWEIGHTS && K2_WPRO && sizeof(T) == 2 && C_::NW * 16 * C_::PADB <= K2_NSLOT * C_::SLOT;
    // FRAGMENT-ORDER weights (round 6; `wfrag`, bf16): Wa arrives pre-arranged as the fragments themselves (csrc/elementwise.hip
    // frag_index, written by the weight-view launch that follows every optimizer step), so a k-step's fragment is ONE coalesced 1-KiB
    // load straight into registers and the first tiles are requested BEFORE the weights instead of after the staging round trips
    constexpr bool WPRO = WPRO_C && !wfrag;
    if (!WPRO)
        for (int s = 0; s < pre; ++s) issue(s);

    // ---- this wave's DW columns of Wa as MFMA "a" operands: row d = DW*wave + 16j + r16; quarter q4 of k-step
    // kk covers the 16-byte chunk (kk + NKK*q4) of the row (any k assignment works as long as H uses the same)
    frag_t wa[C_::NJ][C_::NKK];
    float ba_r[C_::NJ][4], wb_r[C_::NJ][4];
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
            if (K2_ABL_NOWEIGHTS) {
                wa[j][kk] = frag_t{};
                asm volatile("" : "+v"(wa[j][kk]));
                continue;
            }
            wa[j][kk] = *(const frag_t*)(wrow + (kk + C_::NKK * q4) * 16);
            asm volatile("" : "+v"(wa[j][kk]));      // keep resident: never re-load inside the tile loop
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ba_r[j][r] = ba[C_::DW * wave + 16 * j + 4 * q4 + r];
            wb_r[j][r] = wb[C_::DW * wave + 16 * j + 4 * q4 + r];
        }
    }
    const float bb = bb_p[0];
    // tanh is bounded, so every score satisfies |s| <= sum|wb| + |bb| =: smax.  When smax is moderate the soft-max
    // can use smax as a FIXED reference (p = exp(s - smax) in [e^-2smax, 1], no underflow below smax = 30), which
    // removes the running-max reduction and the accumulator rescale from the per-tile path.  Otherwise: online.
    const float smax = wave_sum(fabsf(wb[lane]) + fabsf(wb[64 + lane])) + fabsf(bb);
    const bool fixed_ref = smax < 30.f;
    // retire the compiler-counted loads above; from here on the only VMEM ops in flight are LDS-DMA tiles
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (wfrag) {
#pragma unroll
        for (int j = 0; j < C_::NJ; ++j)
#pragma unroll
            for (int kk = 0; kk < C_::NKK; ++kk) asm volatile("" : "+v"(wa[j][kk]));      // resident from here on: never re-loaded in the loop
    }
    if (WPRO) {
        LDS_BARRIER();                          // every wave has read its fragments back: the ring is free for tiles
        for (int s = 0; s < pre; ++s) issue(s);
    }

    // Pooling runs on the matrix cores too: M[PC*w + 16j + c] += sum_r p_r H[r][.] as a 16x16 MFMA whose A operand
    // carries p in row 0 (bf16: p = hi + lo split over rows 0 and 1, so p keeps ~16 mantissa bits; f32: exact) and
    // zeros elsewhere, and whose B operand is the LDS tile read k-major (rows of the tile = k).
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 macc[C_::NPJ];
#pragma unroll
    for (int j = 0; j < C_::NPJ; ++j) macc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // byte selector building the A fragment from packed (hi | lo<<16) words: row 0 takes the low halves,
    // row 1 the high halves, every other row zeros (v_perm_b32: 0x0c = constant zero byte)
    const unsigned psel = (r16 == 0) ? 0x05040100u : (r16 == 1 ? 0x07060302u : 0x0c0c0c0cu);

    if constexpr (PAIR) {
        // Paired loop: TWO consecutive 16-row tiles (same item: an item is a multiple of 32 rows) per iteration.  The per-tile
        // chain - fragment reads, 32 MFMAs, tanh, cross-wave sum, barrier, weights, pooling MFMAs - is latency-bound, not
        // throughput-bound (matrix pipe ~30 % busy); two independent chains in one instruction stream give the scheduler
        // twice the work between the same two barriers.  Ring: the pair in use + the next pair in flight (64 KiB per CU with
        // two workgroups; the LDS-DMA skeleton of that shape still streams 6.1 TB/s, tools/stream_probe.py).
        float* spart1 = spart + C_::NW * 16;
        const int npair = my_tiles >> 1;
#ifdef K2_STAMPS
        const int stamp_w = (blockIdx.x < K2_STAMP_WG) ? (wave == 0 ? 0 : (wave == 2 ? 1 : -1)) : -1;
        for (int i = tid; i < 2 * K2_STAMP_IT * K2_STAMP_EV; i += 64 * C_::NW) ((unsigned*)sbuf)[i] = 0u;
#endif
        for (int pr = 0; pr < npair; ++pr) {
            const int seq = 2 * pr;
            K2_STAMP(0);
            K2_STAMP(8);
            if (pr == 0 && my_tiles > 2) { K2_WAIT(2 * C_::GT); } else { K2_WAIT(0); }
            K2_STAMP(1);
            LDS_BARRIER();                     // both tiles of this pair landed; the slots of the previous pair are free
            K2_STAMP(2);
            // the next pair's 2 x GT LDS-DMA pieces: issued in one burst here they queue on the CU's address unit with the other
            // waves' (~0.7 k cycles per pair with this wave's matrix pipe idle, profiles/r03_b_inkernel_stamps_panel_k2_before.txt);
            // K2_SPREAD_DMA hands them out one per k-group of the score MFMA loop below
            const bool more = pr >= 1 && seq + 2 < my_tiles;
            const T* nb_base[2] = {H, H};
            int nb_row0[2] = {0, 0};
            if (K2_SPREAD_DMA) {
                if (more) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        nb_base[u] = H + (size_t)ip.bag * N * K2_L;
                        nb_row0[u] = ip.ch * chunk_rows + ip.tin * C_::TR;
                        ip.next(tiles_per_item, gridDim.x, S);
                    }
                }
            } else if (more) { issue(seq + 2); issue(seq + 3); }
            K2_STAMP(3);
            const int tin = cp.tin;
            const int row0 = cp.ch * chunk_rows + tin * C_::TR;
            const char* tile0 = smem + (seq & (K2_NSLOT - 1)) * C_::SLOT;
            const char* tile1 = smem + ((seq + 1) & (K2_NSLOT - 1)) * C_::SLOT;
            f32x4 acc0[C_::NJ], acc1[C_::NJ];
#pragma unroll
            for (int j = 0; j < C_::NJ; ++j) { acc0[j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[j] = acc0[j]; }
            float ps0 = 0.f, ps1 = 0.f;
            // pooling B operands (k-major 4x16 reads of both tiles): with K2_STAGGER they are requested right after the score
            // MFMAs - the fragment registers of the score loop are free by then - and land while the tanh of tile 1 issues
            s16x4 pb0[C_::NPJ], pb1[C_::NPJ];
            const int pu = lane & 15;
            const unsigned ptoff = (4 * q4 + (pu >> 2)) * C_::PADB + (C_::PC * wave + 4 * (pu & 3)) * 2;
            if constexpr (K2_STAGGER && C_::NJ == 2) {
                // Staggered chains: tile 0's 32 MFMAs first; tile 1's 32 MFMAs then run with tile 0's eight tanh evaluations (one
                // v_exp + one v_rcp each) issued between them - the VALU work of one tile rides in the shadow of the other's
                // matrix work instead of both tiles' tanh following both tiles' MFMAs.
                const char* h0 = tile0 + r16 * C_::PADB + C_::NKK * q4 * 16;
                const char* h1 = tile1 + r16 * C_::PADB + C_::NKK * q4 * 16;
                constexpr int GK = 4, NG = C_::NKK / GK;
                frag_t hq[2][GK];
                auto tile_mfmas = [&](const char* hb, f32x4 (&acc)[C_::NJ], auto&& hook) {
#pragma unroll
                    for (int k2 = 0; k2 < GK; ++k2) hq[0][k2] = *(const frag_t*)(hb + k2 * 16);
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        if (g + 1 < NG) {
#pragma unroll
                            for (int k2 = 0; k2 < GK; ++k2) hq[(g + 1) & 1][k2] = *(const frag_t*)(hb + ((g + 1) * GK + k2) * 16);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k2 = 0; k2 < GK; ++k2)
#pragma unroll
                            for (int j = 0; j < C_::NJ; ++j) acc[j] = k2_mma<T>(wa[j][g * GK + k2], hq[g & 1][k2], acc[j]);
                        hook(g);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                tile_mfmas(h0, acc0, [](int) {});
                tile_mfmas(h1, acc1, [&](int g) {
                    // two of tile 0's eight values per k-group: (j, r) = (g / 2, 2 (g % 2) + {0, 1})
                    const int j = g >> 1, r = 2 * (g & 1);
                    ps0 += wb_r[j][r] * fast_tanh(acc0[j][r] + ba_r[j][r]);
                    ps0 += wb_r[j][r + 1] * fast_tanh(acc0[j][r + 1] + ba_r[j][r + 1]);
                });
                K2_STAMP(4);
#pragma unroll
                for (int j = 0; j < C_::NPJ; ++j) {
                    pb0[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile0 + ptoff + j * 32));
                    pb1[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile1 + ptoff + j * 32));
                }
#pragma unroll
                for (int j = 0; j < C_::NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ps1 += wb_r[j][r] * fast_tanh(acc1[j][r] + ba_r[j][r]);
            } else {
            {
                const char* h0 = tile0 + r16 * C_::PADB + C_::NKK * q4 * 16;
                const char* h1 = tile1 + r16 * C_::PADB + C_::NKK * q4 * 16;
                constexpr int GK = 2, NG = C_::NKK / GK;
                frag_t hq0[2][GK], hq1[2][GK];
#pragma unroll
                for (int k2 = 0; k2 < GK; ++k2) { hq0[0][k2] = *(const frag_t*)(h0 + k2 * 16); hq1[0][k2] = *(const frag_t*)(h1 + k2 * 16); }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g + 1 < NG) {
#pragma unroll
                        for (int k2 = 0; k2 < GK; ++k2) {
                            hq0[(g + 1) & 1][k2] = *(const frag_t*)(h0 + ((g + 1) * GK + k2) * 16);
                            hq1[(g + 1) & 1][k2] = *(const frag_t*)(h1 + ((g + 1) * GK + k2) * 16);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k2 = 0; k2 < GK; ++k2)
#pragma unroll
                        for (int j = 0; j < C_::NJ; ++j) {
                            acc0[j] = k2_mma<T>(wa[j][g * GK + k2], hq0[g & 1][k2], acc0[j]);
                            acc1[j] = k2_mma<T>(wa[j][g * GK + k2], hq1[g & 1][k2], acc1[j]);
                        }
                    if (K2_SPREAD_DMA && more) {
                        static_assert(!K2_SPREAD_DMA || NG == 2 * C_::GT, "one LDS-DMA piece per k-group");
                        const int u = g / C_::GT;
                        k2_issue_piece<T, K2F_NWO(T)>(nb_base[u], nb_row0[u], N, lds0 + ((seq + 2 + u) & (K2_NSLOT - 1)) * C_::SLOT,
                                                      wave, lane, g % C_::GT);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            K2_STAMP(4);
#pragma unroll
            for (int j = 0; j < C_::NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ps0 += wb_r[j][r] * fast_tanh(acc0[j][r] + ba_r[j][r]);
                    ps1 += wb_r[j][r] * fast_tanh(acc1[j][r] + ba_r[j][r]);
                }
            }
            ps0 = quarters_sum(ps0);
            ps1 = quarters_sum(ps1);
            if (q4 == 0) { spart[wave * 16 + r16] = ps0; spart1[wave * 16 + r16] = ps1; }
            K2_STAMP(5);
            LDS_BARRIER();
            K2_STAMP(6);

            // phase B (every wave redundantly): lanes 0..15 <-> rows of tile 0, lanes 16..31 <-> rows of tile 1
            float s = -INFINITY;
            if (lane < 2 * C_::TR) {
                const float* sp = (lane < C_::TR) ? spart : spart1;
                s = bb;
#pragma unroll
                for (int w = 0; w < C_::NW; ++w) s += sp[w * 16 + r16];
                if (row0 + lane >= N) s = -INFINITY;
                // raw scores straight to HBM (128 B per pair): every wait at the top of this loop after the first is
                // vmcnt(0), so a compiler-visible store among the hand-counted LDS-DMA operations cannot be miscounted, and
                // the item end needs no staging buffer, workgroup barrier or drain
                if (wave == 0 && row0 + lane < N) scores[(size_t)cp.bag * N + row0 + lane] = s;
            }
            float p;
            if (fixed_ref) {
                p = (s == -INFINITY) ? 0.f : fast_exp(s - smax);
                l_run += p;                                   // per-lane partial (lanes 0..31), reduced at item end
            } else {
                const float tmax = half_wave_max(s);
                const float m_new = fmaxf(m_run, tmax);
                const float scale = (m_run == -INFINITY) ? 0.f : fast_exp(m_run - m_new);
                p = (s == -INFINITY) ? 0.f : fast_exp(s - m_new);
                l_run = l_run * scale + half_wave_sum(lane < 2 * C_::TR ? p : 0.f);
                m_run = m_new;
#pragma unroll
                for (int j = 0; j < C_::NPJ; ++j) macc[j] *= scale;
            }
            if (lane < 2 * C_::TR) {
                const bf16_t hi = f2bf(p);
                pbuf[lane] = (unsigned)hi | ((unsigned)f2bf(p - bf2f(hi)) << 16);
            }
            {
                const u32x4 pa0 = *(const u32x4*)(pbuf + 4 * q4);
                const u32x4 pa1 = *(const u32x4*)(pbuf + 16 + 4 * q4);
                const u32x2 w0 = u32x2{__builtin_amdgcn_perm(pa0[1], pa0[0], psel), __builtin_amdgcn_perm(pa0[3], pa0[2], psel)};
                const u32x2 w1 = u32x2{__builtin_amdgcn_perm(pa1[1], pa1[0], psel), __builtin_amdgcn_perm(pa1[3], pa1[2], psel)};
                const s16x4 af0 = __builtin_bit_cast(s16x4, w0), af1 = __builtin_bit_cast(s16x4, w1);
#pragma unroll
                for (int j = 0; j < C_::NPJ; ++j) {
                    if (!(K2_STAGGER && C_::NJ == 2)) {
                        pb0[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile0 + ptoff + j * 32));
                        pb1[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile1 + ptoff + j * 32));
                    }
                    macc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af0, pb0[j], macc[j], 0, 0, 0);
                    macc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af1, pb1[j], macc[j], 0, 0, 0);
                }
            }
            if (tin + 1 == tiles_per_item - 1) {    // ---- end of item: publish partial + raw scores
                float* pp = part + ((size_t)cp.bag * S + cp.ch) * K2_PSTRIDE;
                if (q4 == 0) {
#pragma unroll
                    for (int j = 0; j < C_::NPJ; ++j) pp[C_::PC * wave + 16 * j + r16] = macc[j][0] + macc[j][1];
                }
                {
                    const float l_tot = fixed_ref ? half_wave_sum(lane < 2 * C_::TR ? l_run : 0.f) : l_run;
                    if (tid == 0) { pp[K2_L] = fixed_ref ? smax : m_run; pp[K2_L + 1] = l_tot; }
                }
                m_run = -INFINITY; l_run = 0.f;
#pragma unroll
                for (int j = 0; j < C_::NPJ; ++j) macc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            cp.next(tiles_per_item, gridDim.x, S);
            cp.next(tiles_per_item, gridDim.x, S);
            K2_STAMP(7);
        }
#ifdef K2_STAMPS
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (blockIdx.x < K2_STAMP_WG)
            for (int i = tid; i < 2 * K2_STAMP_IT * K2_STAMP_EV; i += 64 * C_::NW) (&g_k2_stamps[blockIdx.x][0][0][0])[i] = ((unsigned*)sbuf)[i];
#endif
        return;
    }
    for (int seq = 0; seq < my_tiles; ++seq) {
        const int ahead = min(2, my_tiles - 1 - seq);
        if (ahead == 2) { K2_WAIT(2 * C_::GT); } else if (ahead == 1) { K2_WAIT(C_::GT); } else { K2_WAIT(0); }
        LDS_BARRIER();                         // all waves' pieces landed; slot (seq+3)%4 is free
        if (seq + 3 < my_tiles) issue(seq + 3);

        const int tin = cp.tin;
        const int row0 = cp.ch * chunk_rows + tin * C_::TR;
        const char* tile = smem + (seq & (K2_NSLOT - 1)) * C_::SLOT;

        // ---- phase A: pre-activations of the tile for this wave's DW columns of D
        f32x4 acc[C_::NJ];
#pragma unroll
        for (int j = 0; j < C_::NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* hbase = tile + r16 * C_::PADB + C_::NKK * q4 * 16;
        {
            // explicit software pipeline: the fragments of k-group g+1 are requested before the MFMAs of group g issue
            // (left alone hipcc keeps two ds_read_b128 in flight and every MFMA pair waits out an LDS round trip)
            constexpr int GK = K2_GK, NG = C_::NKK / GK;
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
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float ps = 0.f;
#pragma unroll
        for (int j = 0; j < C_::NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = acc[j][r] + ba_r[j][r];
                ps += wb_r[j][r] * (EXACT_TANH ? tanhf(x) : fast_tanh(x));
            }
        ps = quarters_sum(ps);                               // over the 4 lane quarters (same r16)
        if (q4 == 0) spart[wave * 16 + r16] = ps;
        LDS_BARRIER();

        // ---- phase B (every wave redundantly): tile soft-max statistics, lane r <-> row r (lanes 0..15)
        float s = -INFINITY;
        if (lane < C_::TR) {
            s = bb;
#pragma unroll
            for (int w = 0; w < C_::NW; ++w) s += spart[w * 16 + lane];
            if (row0 + lane >= N) s = -INFINITY;             // ragged tail rows carry no weight
            if (wave == 0) sbuf[tin * C_::TR + lane] = s;
        }
        float p;
        if (fixed_ref) {
            p = (s == -INFINITY) ? 0.f : (EXACT_TANH ? __expf(s - smax) : fast_exp(s - smax));
            l_run += p;                                       // per-lane partial (lanes 0..15), reduced at item end
        } else {
            const float tmax = rdlane(row16_max(s), 0);
            const float m_new = fmaxf(m_run, tmax);
            const float scale = (m_run == -INFINITY) ? 0.f : (EXACT_TANH ? __expf(m_run - m_new) : fast_exp(m_run - m_new));
            p = (s == -INFINITY) ? 0.f : (EXACT_TANH ? __expf(s - m_new) : fast_exp(s - m_new));
            l_run = l_run * scale + rdlane(row16_sum(p), 0);
            m_run = m_new;
#pragma unroll
            for (int j = 0; j < C_::NPJ; ++j) macc[j] *= scale;
        }
        // ---- pooling on the MFMAs: this wave owns output columns PC*wave .. +PC-1
        if (sizeof(T) == 2) {
            if (lane < 16) {
                const bf16_t hi = f2bf(p);
                pbuf[lane] = (unsigned)hi | ((unsigned)f2bf(p - bf2f(hi)) << 16);
            }
            // A fragment of v_mfma_f32_16x16x16_bf16: lane (q4, r16) = A[row r16][k = 4q4 + e], e = 0..3
            const u32x4 pa = *(const u32x4*)(pbuf + 4 * q4);
            const u32x2 afw = u32x2{__builtin_amdgcn_perm(pa[1], pa[0], psel), __builtin_amdgcn_perm(pa[3], pa[2], psel)};
            const s16x4 af = __builtin_bit_cast(s16x4, afw);
            // B fragment: column 16j + (lane&15) of rows 4q4 .. 4q4+3 = one transposed 4x16 read
            const int u = lane & 15, rq = u >> 2, pp4 = u & 3;
            const char* tb = tile + (4 * q4 + rq) * C_::PADB + (C_::PC * wave + 4 * pp4) * 2;
#pragma unroll
            for (int j = 0; j < C_::NPJ; ++j) {
                const s16x4 bfm = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tb + j * 32));
                macc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af, bfm, macc[j], 0, 0, 0);
            }
        } else {
            // f32: 16 rows = 4 k-steps of v_mfma_f32_16x16x4_f32; A[row r16][k = q4] = p[4s+q4] in row 0 only
            if (lane < 16) pbuf[lane] = __float_as_uint(p);
            const char* tb = tile + q4 * C_::PADB + (C_::PC * wave + r16) * 4;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const float av = (r16 == 0) ? __uint_as_float(pbuf[4 * st + q4]) : 0.f;
#pragma unroll
                for (int j = 0; j < C_::NPJ; ++j) {
                    const float bv = *(const float*)(tb + st * 4 * C_::PADB + j * 64);
                    macc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, macc[j], 0, 0, 0);
                }
            }
        }

        if (tin == tiles_per_item - 1) {        // ---- end of item: publish partial + raw scores
            float* pp = part + ((size_t)cp.bag * S + cp.ch) * K2_PSTRIDE;
            if (q4 == 0) {                      // rows 0 (+1) of the accumulator tiles live in lane quarter 0
#pragma unroll
                for (int j = 0; j < C_::NPJ; ++j) pp[C_::PC * wave + 16 * j + r16] = macc[j][0] + macc[j][1];
            }
            {
                const float l_tot = fixed_ref ? rdlane(row16_sum(lane < 16 ? l_run : 0.f), 0) : l_run;
                if (tid == 0) { pp[K2_L] = fixed_ref ? smax : m_run; pp[K2_L + 1] = l_tot; }
            }
            __syncthreads();                    // sbuf complete
            const int rbeg = cp.ch * chunk_rows;
            for (int r = tid; r < chunk_rows && rbeg + r < N; r += 64 * C_::NW) scores[(size_t)cp.bag * N + rbeg + r] = sbuf[r];
            // compiler-visible stores: retire them so the hand counts of LDS-DMA ops stay exact
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            m_run = -INFINITY; l_run = 0.f;
#pragma unroll
            for (int j = 0; j < C_::NPJ; ++j) macc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        cp.next(tiles_per_item, gridDim.x, S);
    }
}

// Per bag: merge chunk partials, emit A[n] = softmax(s)_n / sqrt(N), pooled M, and (m, l).  K2C_SPLIT workgroups per bag (round 5:
// one per bag left half the chip idle and made this launch 5.2-5.6 us of K2's 58): each takes 1 / K2C_SPLIT of the pooled columns and
// of the rows of A; all of them read the bag's S (m, l) headers (a few hundred bytes, L2-resident).
// Round 6: off the training step (the merge lives in abmil_pool_decoder_kernel below); what is left is the stand-alone entry point and
// `last_attention` on demand.  Any of A / Mout may be NULL; part == NULL: (m, l) are READ from ml (the statistics a decoder launch
// left behind) and only A is formed.
#define K2C_SPLIT 4
__global__ __launch_bounds__(128) void abmil_pool_combine_kernel(const float* __restrict__ scores,
                                                                 const float* __restrict__ part, float* __restrict__ A,
                                                                 float* __restrict__ Mout, float* __restrict__ ml,
                                                                 int N, int S, float inv_sqrt_n) {
    const int bag = blockIdx.x / K2C_SPLIT, q = blockIdx.x % K2C_SPLIT, tid = threadIdx.x;
    constexpr int CPT = K2_L / K2C_SPLIT / 128;          // pooled columns per thread: 1
    static_assert(CPT == 1, "one pooled column per thread");
    const int col = q * (K2_L / K2C_SPLIT) + tid;
    float m = -INFINITY, l = 0.f, a0 = 0.f;
    if (!part) {
        m = ml[2 * bag];
        l = ml[2 * bag + 1];
    } else {
        const float* pp = part + (size_t)bag * S * K2_PSTRIDE;
        // the chunks' (m, l) headers first, one per thread: walked one chunk after the other by every thread they were 2 S dependent
        // round trips (9 us at S = 8)
        __shared__ float hm[128], hl[128];
        if (S <= 128) {
            if (tid < S) { hm[tid] = pp[(size_t)tid * K2_PSTRIDE + K2_L]; hl[tid] = pp[(size_t)tid * K2_PSTRIDE + K2_L + 1]; }
            __syncthreads();
            for (int s = 0; s < S; ++s) m = fmaxf(m, hm[s]);
            int s = 0;
            for (; s + 3 < S; s += 4) {                          // four chunks' partial rows in flight
                float w[4], p0[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    w[u] = (hm[s + u] == -INFINITY) ? 0.f : expf(hm[s + u] - m);
                    p0[u] = pp[(size_t)(s + u) * K2_PSTRIDE + col];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { l += hl[s + u] * w[u]; a0 += p0[u] * w[u]; }
            }
            for (; s < S; ++s) {
                const float w = (hm[s] == -INFINITY) ? 0.f : expf(hm[s] - m);
                l += hl[s] * w;
                a0 += pp[(size_t)s * K2_PSTRIDE + col] * w;
            }
        } else {
            for (int s = 0; s < S; ++s) m = fmaxf(m, pp[(size_t)s * K2_PSTRIDE + K2_L]);
            for (int s = 0; s < S; ++s) {
                const float* p = pp + (size_t)s * K2_PSTRIDE;
                const float w = (p[K2_L] == -INFINITY) ? 0.f : expf(p[K2_L] - m);
                l += p[K2_L + 1] * w;
                a0 += p[col] * w;
            }
        }
    }
    const float inv = inv_sqrt_n / l;
    if (part && Mout) Mout[(size_t)bag * K2_L + col] = a0 * inv;
    if (part && ml && q == 0 && tid == 0) { ml[2 * bag] = m; ml[2 * bag + 1] = l; }
    if (!A) return;
    // this workgroup's quarter of the bag's rows of A
    const int per = (N + K2C_SPLIT - 1) / K2C_SPLIT, n0 = q * per, n1 = min(N, n0 + per);
    const float* sc = scores + (size_t)bag * N;
    float* ab = A + (size_t)bag * N;
    int n = n0 + tid;
    for (; n + 3 * 128 < n1; n += 4 * 128) {                 // four score loads in flight per thread
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = sc[n + u * 128];
#pragma unroll
        for (int u = 0; u < 4; ++u) ab[n + u * 128] = expf(v[u] - m) * inv;
    }
    for (; n < n1; n += 128) ab[n] = expf(sc[n] - m) * inv;
}

#ifndef K2_ITEMS_PER_WG
#define K2_ITEMS_PER_WG 2
#endif
static int pick_chunk(int B, int N, int tr, int n_wg) {
    // rows per item: multiple of the tile height, <= K2_MAX_CHUNK, small enough to give every workgroup >= 2 items
    int chunk = ((N + tr - 1) / tr) * tr;
    if (chunk > K2_MAX_CHUNK) chunk = K2_MAX_CHUNK;
    // (tools/k2_chunk_probe.py, round 4: this choice is within 3 % of the best chunk at every (bags, rows) shape tried)
    while (chunk > 4 * tr && (long)B * ((N + chunk - 1) / chunk) < (long)K2_ITEMS_PER_WG * n_wg) {
        int c2 = ((chunk / 2 + tr - 1) / tr) * tr;
        if (c2 == chunk) break;
        chunk = c2;
    }
    return chunk;
}

extern "C" int murcl_abmil_pool_workspace(int B, int N, int dtype, int* chunk_rows, int* n_chunks) {
    const int tr = 32;                       // multiple of both kernels' tile heights (fwd 16, bwd 32/16)
    const int c = pick_chunk(B, N, tr, dtype == MURCL_DTYPE_BF16 ? 512 : 256);
    *chunk_rows = c;
    *n_chunks = (N + c - 1) / c;
    return 0;
}

// C-ABI: see include/murcl_amd.h
extern "C" int murcl_abmil_pool_fwd(const void* H, const void* Wa, const float* ba, const float* wb, const float* bb,
                                    float* scores, float* A, float* M, float* ml, float* part_ws, int B, int N, int L,
                                    int D, int dtype, int exact_tanh, hipStream_t stream) {
    if (L != K2_L || D != K2_D) return -1;
    const int wfrag = (exact_tanh >> 1) & 1;         // flags: bit 0 = exact tanh, bit 1 = Wa in fragment order (bf16 only)
    exact_tanh &= 1;
    if (wfrag && dtype != MURCL_DTYPE_BF16) return -1;
    if (B <= 0 || N <= 0) return 0;
    int chunk, S;
    murcl_abmil_pool_workspace(B, N, dtype, &chunk, &S);
    const int items = B * S;
    const int max_grid = murcl_cu_budget() * (dtype == MURCL_DTYPE_BF16 ? 2 : 1);
    const int grid = items < max_grid ? items : max_grid;
#define K2_LAUNCH(T, EX, WF)                                                                                      \
    {                                                                                                           \
        typedef K2<T, K2F_NWO(T)> CL_;                                                                          \
        typedef K2Lds<T, K2F_NWO(T)> LL_;                                                                       \
        auto k = abmil_pool_fwd_kernel<T, EX, WF>;                                                              \
        static MurclOncePerDevice once;                                                                                     \
        if (once.first()) {                                                                                            \
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LL_::BYTES);        \
                                                                                                                   \
        }                                                                                                       \
        hipLaunchKernelGGL(k, dim3(grid), dim3(64 * CL_::NW), LL_::BYTES, stream, (const T*)H, (const T*)Wa,    \
                           ba, wb, bb, scores, part_ws, B, N, chunk, S);                                        \
    }
    if (dtype == MURCL_DTYPE_BF16) {
        if (exact_tanh) { if (wfrag) K2_LAUNCH(bf16_t, true, true) else K2_LAUNCH(bf16_t, true, false) }
        else { if (wfrag) K2_LAUNCH(bf16_t, false, true) else K2_LAUNCH(bf16_t, false, false) }
    } else if (dtype == MURCL_DTYPE_F32) {
        if (exact_tanh) K2_LAUNCH(float, true, false) else K2_LAUNCH(float, false, false)
    } else {
        return -1;
    }
#undef K2_LAUNCH
    int rc = MURCL_CHECK_LAUNCH();
    if (rc || (!A && !M && !ml)) return rc;          // partials only: the caller runs murcl_abmil_pool_combine itself
    hipLaunchKernelGGL(abmil_pool_combine_kernel, dim3(B * K2C_SPLIT), dim3(128), 0, stream, scores, part_ws, A, M, ml, N, S,
                       1.0f / sqrtf((float)N));
    return MURCL_CHECK_LAUNCH();
}

// C-ABI: see include/murcl_amd.h
extern "C" int murcl_abmil_pool_combine(const float* scores, const float* part_ws, float* A, float* M, float* ml, int B,
                                        int N, int dtype, hipStream_t stream) {
    if (B <= 0 || N <= 0) return 0;
    int chunk, S;
    murcl_abmil_pool_workspace(B, N, dtype, &chunk, &S);
    hipLaunchKernelGGL(abmil_pool_combine_kernel, dim3(B * K2C_SPLIT), dim3(128), 0, stream, scores, part_ws, A, M, ml, N, S,
                       1.0f / sqrtf((float)N));
    return MURCL_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------- K2's per-bag merge inside K3 (round 6)
// The bag-level consumer of the pooled vector is the decoder Linear + ReLU (abmil.py:29-32,43): out = relu(M Wd^T + bd).  The
// merge of a bag's S chunk partials (m, l, sum p.H) into M = sum_s e^{m_s - m} part_s / (l sqrt N) is a few hundred bytes of
// arithmetic per bag, but as its own launch it cost the K2 row a second dependent launch (5 us of a 58 us row).  Here the
// decoder product forms M while it loads its A operand: a workgroup owns a 16-bag x 16-column output tile (the tile shape of
// gru.hip's gemm_nt_t16_f32_kernel, which ran this product before), its 16 rows of Wd arrive by LDS-DMA while the threads merge
// the 16 bags' partial rows into the A rows of the same LDS image; the workgroups of the first column tile also write M and
// (m, l) for the backward pass.  Exact f32 (v_mfma_f32_16x16x4_f32), single writer per element, no atomics.
// The normalised attention row A = softmax(s) / sqrt(N) is NOT formed in the forward pass any more: nothing in the training
// step reads it before the pooling backward, which computes p from the raw scores and (m, l) anyway and leaves A behind for
// the rank-1 input gradient (attn_pool_bwd.hip); `last_attention` comes from murcl_abmil_pool_combine on demand.
constexpr int PD_T = 16, PD_K = 256, PD_ROW = PD_K * 4 + 16, PD_SLOT = 2 * PD_T * PD_ROW, PD_NCH = K2_L / PD_K;
constexpr int PD_MAXS = 512;                        // chunk headers of 16 bags in LDS: 16 * S * 8 bytes (64 KiB beside the 65 KiB image)
static_assert(PD_NCH == 2, "two 256-k chunks: both in flight at once");
__device__ __forceinline__ const float* pd_uniform(const float* p) {
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const float*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}
__global__ __launch_bounds__(256) void abmil_pool_decoder_kernel(const float* __restrict__ part, const float* __restrict__ Wd,
                                                                 const float* __restrict__ bd, float* __restrict__ Mout,
                                                                 float* __restrict__ ml, float* __restrict__ out, int B, int S,
                                                                 int Lout, float inv_sqrt_n, int relu) {
    extern __shared__ __attribute__((aligned(16))) char pd_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q4 = lane >> 4, r16 = lane & 15;
    const int j0 = blockIdx.x * PD_T, b0 = blockIdx.y * PD_T;
    const unsigned lds0 = lds_off(pd_smem);
    float* hm = (float*)(pd_smem + PD_NCH * PD_SLOT);               // [16][S] chunk maxima -> merge weights
    float* hl = hm + PD_T * S;                                       // [16][S] chunk sums
    // 1. this tile's 16 rows of Wd, both k chunks: 32 row pieces of 1 KiB, eight per wave, straight into image rows 16..31
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = wave * 8 + j, row = i & 15, ch = i >> 4;
        glds16_u(pd_uniform(Wd + (size_t)(j0 + row) * K2_L + ch * PD_K), lane * 16, lds0 + ch * PD_SLOT + (PD_T + row) * PD_ROW);
    }
    // 2. + 3. the merge.  16 threads per bag: thread (b, k) owns the eight 4-column groups 4 (k + 16 i) of bag b, so that a bag's
    // partial row is read as 256-byte segments.  The partial rows do not depend on the headers: the first PD_SB chunks' loads go out
    // together with the header loads - one memory round trip for headers + half the rows at S = 8, one more for the rest (walking
    // them bag after bag behind the weights was eight dependent round trips: 11 us for the launch).
    {
        constexpr int PD_SB = 4;
        const int b = tid >> 4, k = tid & 15;
        const int bag = min(b0 + b, B - 1);                           // bags past B: the last bag again, their rows are never stored
        const float* pp = part + (size_t)bag * S * K2_PSTRIDE;
        f32x4 a[8], p[PD_SB][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto load = [&](int s0) {
#pragma unroll
            for (int v = 0; v < PD_SB; ++v) {
                const float* row = pp + (size_t)min(s0 + v, S - 1) * K2_PSTRIDE + 4 * k;
#pragma unroll
                for (int i = 0; i < 8; ++i) p[v][i] = *(const f32x4*)(row + 64 * i);      // (rows are 16-byte aligned: K2_PSTRIDE)
            }
        };
        load(0);
        // headers: thread k of the bag's 16 takes chunks k, k + 16, ...
        float m = -INFINITY;
        for (int s = k; s < S; s += 16) {
            const float ms = pp[(size_t)s * K2_PSTRIDE + K2_L];
            hm[b * S + s] = ms;
            hl[b * S + s] = pp[(size_t)s * K2_PSTRIDE + K2_L + 1];
            m = fmaxf(m, ms);
        }
        m = row16_max(m);
        float l = 0.f;
        for (int s = k; s < S; s += 16) {
            const float ms = hm[b * S + s];
            const float w = (ms == -INFINITY) ? 0.f : expf(ms - m);
            l += hl[b * S + s] * w;
            hm[b * S + s] = w;                                        // the merge weights replace the maxima
        }
        l = row16_sum(l);
        const float inv = inv_sqrt_n / l;
        if (k == 0 && blockIdx.x == 0 && b0 + b < B) { ml[2 * (b0 + b)] = m; ml[2 * (b0 + b) + 1] = l; }
        __syncthreads();                                              // (a bag's 16 threads share a wave; the barrier is the simple fence)
        for (int s0 = 0; s0 < S; s0 += PD_SB) {
#pragma unroll
            for (int v = 0; v < PD_SB; ++v) {
                const float w = (s0 + v < S) ? hm[b * S + s0 + v] : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] += p[v][i] * w;
            }
            if (s0 + PD_SB < S) load(s0 + PD_SB);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int col = 4 * (k + 16 * i);
            const f32x4 mv = a[i] * inv;
            *(f32x4*)(pd_smem + (col >> 8) * PD_SLOT + b * PD_ROW + (col & (PD_K - 1)) * 4) = mv;
            if (blockIdx.x == 0 && b0 + b < B) *(f32x4*)(Mout + (size_t)(b0 + b) * K2_L + col) = mv;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the Wd rows (and this thread's loads / stores above)
    __syncthreads();
    // 4. the product: wave w takes k units 4w .. 4w + 3 (16 k each) of both chunks, two accumulation chains
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int c = 0; c < PD_NCH; ++c) {
        const char* ab = pd_smem + c * PD_SLOT + r16 * PD_ROW + 16 * q4;
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            const int u = wave * 4 + uu;
            const f32x4 a = *(const f32x4*)(ab + 64 * u);
            const f32x4 b = *(const f32x4*)(ab + PD_T * PD_ROW + 64 * u);
            acc[uu & 1] = k2_mma<float>(a, b, acc[uu & 1]);
        }
    }
    __syncthreads();                                                 // every wave has read the image: park the partial tiles in it
    float* P = (float*)pd_smem;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[((wave * 2 + g) * 16 + 4 * q4 + r) * 16 + r16] = acc[g][r];
    __syncthreads();
    const int b = b0 + (tid >> 4), n = j0 + (tid & 15);
    if (b >= B) return;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += P[(w * 2) * 256 + tid] + P[(w * 2 + 1) * 256 + tid];
    v += bd ? bd[n] : 0.f;
    if (relu) v = fmaxf(v, 0.f);
    out[(size_t)b * Lout + n] = v;
}

// C-ABI: see include/murcl_amd.h
extern "C" int murcl_abmil_pool_decoder(const float* part_ws, const float* Wd, const float* bd, float* M, float* ml, float* out,
                                        int B, int N, int L, int Lout, int dtype, int relu, hipStream_t stream) {
    if (L != K2_L || Lout <= 0 || Lout % PD_T) return -1;
    if (B <= 0 || N <= 0) return 0;
    int chunk, S;
    murcl_abmil_pool_workspace(B, N, dtype, &chunk, &S);
    if (S > PD_MAXS || (B + PD_T - 1) / PD_T > 65535) return -1;     // (callers then run murcl_abmil_pool_combine + murcl_gemm_nt)
    const int lds = PD_NCH * PD_SLOT + (2 * PD_T * S + PD_T) * 4;
    static MurclOncePerDevice once;
    if (once.first())
        hipFuncSetAttribute((const void*)abmil_pool_decoder_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            PD_NCH * PD_SLOT + (2 * PD_T * PD_MAXS + PD_T) * 4);
    hipLaunchKernelGGL(abmil_pool_decoder_kernel, dim3(Lout / PD_T, (B + PD_T - 1) / PD_T), dim3(256), lds, stream, part_ws, Wd, bd,
                       M, ml, out, B, S, Lout, 1.0f / sqrtf((float)N), relu);
    return MURCL_CHECK_LAUNCH();
}
