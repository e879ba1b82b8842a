// Weight-stationary bf16 "panel" GEMM for the patch-level Linear layers (M = B*N rows >> N, K <= 512):
//
//     C[M,N] = epi( A[M,K] . W[N,K]^T )          A, W, C bf16; f32 accumulate
//
// Structure = the K2 streaming kernel: persistent workgroups (one per CU, 8 waves = two per SIMD so that one
// wave's LDS-DMA issue / epilogue overlaps its SIMD partner's MFMAs), each wave keeps a WN-column slice of
// W as MFMA operands in registers for the whole launch, A row tiles (32 rows) stream HBM -> LDS through a
// 4-slot LDS-DMA ring with three tiles in flight.  Every VMEM operation in the main loop is inline asm
// (LDS-DMA loads, output stores, mask stores) and is counted by hand with s_waitcnt vmcnt(N), so hipcc
// never drains the ring.
//
//   K = 512: WN = 32 -> a workgroup owns a 256-column panel; N/256 panels walk the same row tiles on
//                       workgroups b and b+8 (same XCD under round-robin dispatch -> A re-read from L2)
//   K = 128: WN = 64 -> one 512-column panel (dgrad of the attention projection)
//
// Epilogue (per tile, per wave, no workgroup barrier): f32 math in accumulator layout (bias+ReLU, or
// rank-1 term a[m]*v[bag(m)][n]), ReLU' applied / recorded as one bit per element while still in accumulator
// layout (column sums for the bias gradients are taken there too, from the f32 values), round to bf16, transpose
// through a wave-private LDS patch so that each lane owns 8 consecutive columns of a row and store 16 B per lane in
// whole 128/256-byte row segments.  The forward variant emits the bit mask, so the backward reads 1/16 of the bytes of H for ReLU'.
// CLAM-SB's epilogues (round 3): PG_GATE / PG_GATE_U take the two gate Linears interleaved in 16-row blocks, so that the lane that
// holds a_d also holds b_d, and emit per-wave partial attention scores sum_d tanh(a_d) sigmoid(b_d) c_d (PG_GATE_U also stores the
// pre-activations for the backward pass); the DROP template flag applies nn.Dropout from a seed inside the epilogue (behind the ReLU
// of PG_BIAS_RELU, on the two gate branches of PG_GATE_U) - masks are never materialised.
// Mask layout (M*N/8 bytes): blocks of 128 B per (32-row tile, 32-column group), tile-major.  A block is 64
// 16-bit words, one per MFMA lane L = 16*((n&15)>>2) + (m&15); element (m, n) of the block is the lane's
// accumulator value idx = 8*((m>>4)&1) + 4*((n>>4)&1) + (n&3) and sits at bit (7 - idx/2) + 8*(idx&1):
// exactly what "shift left, OR in the packed pair's >0 flags" leaves behind (2 VALU ops per bf16 pair), and
// the consumer applies it with one v_bfe_i32 + v_and per accumulator value.
#include "common.h"
#include <type_traits>

#define PG_TR 32
#define PG_NSLOT 4
#ifndef PG_GK
#define PG_GK 2           // k-steps per LDS prefetch group
#endif
#ifndef PG_PF
#define PG_PF 1           // groups requested ahead of the MFMAs
#endif
#ifndef PG_ROTATE
#define PG_ROTATE 0       // 1: phase rotation of the upper half of the workgroup (see the tile loop); measured 4 % slower
#endif
#ifndef PG_FLIP
#define PG_FLIP 0
#endif
#ifndef PG_FUSE
#define PG_FUSE 1         // bit 0: forward epilogues (bias / bias+ReLU), bit 1: the masked dgrad - every wave runs the epilogue of
#endif                    // tile t-1 INSIDE the MFMA loop of tile t (see the tile loop).  Forward 147 -> 142 us; dgrad neutral, so off there
#ifndef PG_SPREAD_DMA
#define PG_SPREAD_DMA 0   // 1: the LDS-DMA pieces of tile t+3 are issued one per k-group inside the MFMA phase of tile t (measured r03: neutral at K = 512, +10 % time at K = 128)
#endif
#ifndef PG_DMA_SKEW
#define PG_DMA_SKEW 0     // 1: the upper half of the workgroup issues each LDS-DMA piece one k-group earlier than its SIMD partners
#endif
#ifndef PG_PRIO
#define PG_PRIO 0         // 1: s_setprio 1 around the MFMAs of a k-group (T5); 2: the two halves of the workgroup (= the two waves of
#endif                    // a SIMD) take priority 1 in alternate k-groups, so neither is the arbitration loser for a whole tile
#ifndef PG_DMA_HALF
#define PG_DMA_HALF 0     // 1: waves 0..NW/2-1 issue ALL LDS-DMA pieces of a tile (two A pieces each per slot), the upper half none;
#endif                    // 2: the upper half issues them all.  K = 512 variants only.
#ifndef PG_ABL
#define PG_ABL 0          // diagnostic ablations (wrong results): 1 no LDS fragment reads, 2 no LDS-DMA in the loop, 4 no stores, 8 no MFMAs, 16 no epilogue, 32 A loaded by column panel 0 only
#endif
#ifndef PG_REGSTAGE
#define PG_REGSTAGE 0     // 1: forward variants (K = 512, bias epilogues): A tiles travel global -> VGPRs -> LDS (plain 16-byte loads two
#endif                    // tiles ahead into two register sets, ds_write_b128 one tile ahead) instead of LDS-DMA (see the tile loop)
#ifndef PG_WPRO
#define PG_WPRO 1         // K = 512: a wave's W slice arrives as whole 1 KiB rows (LDS-DMA into the still empty tile ring) and is read
#endif                    // back as fragments, instead of fragment-shaped global loads that touch 64 cache lines each (see the prologue)
#ifndef PG_GATE_U_FUSE
#define PG_GATE_U_FUSE 1  // PG_GATE_U rides the fused forward schedule (its epilogue inside the next tile's MFMA loop) like PG_BIAS
#endif
#ifndef PG_EXACT_WAIT
#define PG_EXACT_WAIT 0
#endif
#ifndef PG_WIDE
#define PG_WIDE 0         // K = 512: 1 -> 4 waves x 64 columns (512 registers per wave), 0 -> 8 waves x 32 columns
#endif

enum { PG_BIAS_RELU = 0, PG_MASK = 1, PG_RANK1_MASK = 2, PG_BIAS = 3, PG_GATE = 4, PG_GATE_U = 5 };

// Dropout inside an epilogue (template flag DROP): the keep value of element `flat` of the [M, cols] mask is byte flat & 7 of
// murcl_drop_word(seed, flat >> 3) - the generator of murcl_dropout_mask / murcl_dropout_relu_bitmask, so a mask is the same
// whichever kernel realises it.  A lane's four accumulator values are four consecutive columns: half a word.
struct PgDrop { unsigned long long seed_a, seed_b; unsigned thresh; float scale; };
__device__ __forceinline__ unsigned pg_keep4(unsigned long long seed, long flat4) {
    const unsigned long long rw = murcl_drop_word(seed, flat4 >> 3);
    return (unsigned)(rw >> (8 * (int)(flat4 & 4)));
}

// In-kernel stamps (diagnostic builds only, -DPG_STAMPS; tools/stamps_panel.py): waves 0 and 4 of the first PG_STAMP_WG
// workgroups note s_memtime at fixed points of each tile iteration into a spare LDS region (a global store would join the
// hand-counted vmcnt queue) and copy it out after the last tile.  No output value depends on a stamp.
#ifdef PG_STAMPS
#define PG_STAMP_WG 16
#define PG_STAMP_IT 16            // every 4th tile iteration is recorded (LDS has ~1 KiB to spare)
#define PG_STAMP_EV 8
#define PG_STAMP_BYTES (2 * PG_STAMP_IT * PG_STAMP_EV * 4)
__device__ unsigned g_pg_stamps[PG_STAMP_WG][2][PG_STAMP_IT][PG_STAMP_EV];
extern "C" int murcl_debug_pg_stamps(void* host, long bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pg_stamps), (size_t)bytes, 0, hipMemcpyDeviceToHost);
}
#define PG_STAMP(ev)                                                                                              \
    do {                                                                                                          \
        if (stamp_w >= 0 && (seq & 3) == 0 && (seq >> 2) < PG_STAMP_IT) {                                                                  \
            const unsigned long long t_ = ((ev) == 6) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
            if (lane == 0) stamp_lds[(stamp_w * PG_STAMP_IT + (seq >> 2)) * PG_STAMP_EV + (ev)] = (unsigned)t_;          \
        }                                                                                                         \
    } while (0)
#else
#define PG_STAMP_BYTES 0
#define PG_STAMP(ev)
#endif

#define PG_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

#ifndef PG_STORE_POLICY
#define PG_STORE_POLICY ""        // cache-policy suffix of the output stores: "" | " nt" | " sc0 sc1" (A/B: tools/ab_build.sh)
#endif
__device__ __forceinline__ void pg_store16(void* p, u32x4 v) {
    if (PG_ABL & 4) { asm volatile("" ::"v"(p), "v"(v)); return; }
    asm volatile("global_store_dwordx4 %0, %1, off" PG_STORE_POLICY "\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void pg_store4(void* p, float v) {
    asm volatile("global_store_dword %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void pg_store2(void* p, unsigned v) {
    if (PG_ABL & 4) { asm volatile("" ::"v"(p), "v"(v)); return; }
    asm volatile("global_store_short %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// plain global load of 16 bytes per lane, wave-uniform base + per-lane offset; hipcc does not track it: the caller waits by
// hand (s_waitcnt vmcnt) before the first use of the result
__device__ __forceinline__ u32x4 pg_gload16(const void* sbase, unsigned voff) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
    return v;
}
__device__ __forceinline__ u32x4 pg_gload16_nt(const void* sbase, unsigned voff) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2" GLDS_STREAM_POLICY : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
    return v;
}

// per bf16 half of w: 1 if > 0 (signed 16-bit compare: -0.0 and negatives give 0), else 0.  `ones` arrives opaque (an empty asm
// on the register): with a literal 1 hipcc turns the min/max pair into two compares, two selects and a permute.
// PG_NOASM (default 1, round 3): the packed 16-bit min / max go through __builtin_elementwise_min/max instead of inline asm -
// same instructions (v_pk_min_i16 / v_pk_max_i16), but the scheduler knows them and keeps interleaving the fused epilogue's
// VALU work with the MFMAs of the next tile (an asm statement is a black box with unknown latency in the middle of that region).
#ifndef PG_NOASM
#define PG_NOASM 1
#endif
typedef short pg_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pg_pos_flags(unsigned w, unsigned ones) {
    if (PG_NOASM) {
        const pg_s16x2 z = {0, 0};
        const pg_s16x2 t = __builtin_elementwise_max(__builtin_elementwise_min(__builtin_bit_cast(pg_s16x2, w), __builtin_bit_cast(pg_s16x2, ones)), z);
        return __builtin_bit_cast(unsigned, t);
    }
    unsigned t;
    asm("v_pk_min_i16 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0" : "=&v"(t) : "v"(w), "v"(ones));
    return t;
}
// ReLU on a packed bf16 pair: as signed 16-bit integers every negative bf16 (and -0.0) is < 0, every positive one > 0, so
// max(., 0) per half IS the ReLU - after the rounding to bf16, which commutes with it (rounding is monotonic).  One VALU
// operation per pair instead of two v_max_f32; the > 0 flags of the result are then min(., 1) per half.
__device__ __forceinline__ unsigned pg_relu_pk(unsigned w) {
    if (PG_NOASM) {
        const pg_s16x2 z = {0, 0};
        return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(pg_s16x2, w), z));
    }
    unsigned t;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(t) : "v"(w));
    return t;
}
__device__ __forceinline__ unsigned pg_pos_flags_nonneg(unsigned w, unsigned ones) {
    if (PG_NOASM) return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(pg_s16x2, w), __builtin_bit_cast(pg_s16x2, ones)));
    unsigned t;
    asm("v_pk_min_i16 %0, %1, %2" : "=v"(t) : "v"(w), "v"(ones));
    return t;
}

template <int K, int WN, int PG_NW, int EPI, bool BM_OUT, bool DROP = false, bool WFRAG = false>
__global__ __launch_bounds__(64 * PG_NW) void panel_nt_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, bf16_t* __restrict__ C, int M, int N,
    const float* __restrict__ bias, uint8_t* __restrict__ bm_out, const uint8_t* __restrict__ bm_in,
    const float* __restrict__ rowscale, const float* __restrict__ rank1, int rows_per_bag,
    float* __restrict__ colsum_part, int walk_reverse, PgDrop drop) {
    constexpr int ROWB = K * 2;                 // bytes per A row
    // K = 512: one LDS-DMA instruction writes exactly one row, so rows can be stored at a padded stride (conflict-free
    // 16-row fragment reads with immediate offsets, no swizzle math).  K = 128: four rows per instruction -> XOR swizzle.
    constexpr bool PAD = (K == 512);
    constexpr int PADB = PAD ? ROWB + 16 : ROWB;
    constexpr int SLOT = PG_TR * PADB;          // 32.5 KiB / 8 KiB
    constexpr int CPR = ROWB / 16;              // 16-byte chunks per row
    constexpr int GT = PAD ? PG_TR / PG_NW : SLOT / (PG_NW * 1024);   // tile LDS-DMA ops per wave
    constexpr int NJ = WN / 16, NKK = K / 32;
    constexpr int NP = PG_NW * WN;              // columns per workgroup: 256 / 512
    constexpr int CPW = WN / 8;                 // 16-byte chunks per output row per wave: 4 / 8
    constexpr int RPI = 64 / CPW;               // rows per store instruction: 16 / 8
    constexpr int NS = PG_TR / RPI;             // store instructions per tile: 2 / 4
    constexpr int STG_LD = WN * 2 + 16;         // staging row stride (bytes), 16-B aligned, breaks the pow-2 stride
    constexpr bool MASKED = (EPI == PG_MASK || EPI == PG_RANK1_MASK);
    // PG_GATE (CLAM's gated attention score, clam.py:55-60, forward-only calls): W holds the two gate branches interleaved so that a
    // wave's first 16-column block is attention_a[d .. d+15] and its second block attention_b of the SAME d: the lane that holds
    // a_d also holds b_d, and the epilogue emits sum_d tanh(a_d) sigmoid(b_d) wc_d over the wave's 16 pairs per row - one f32 per
    // row and wave ([N/32][M] partial scores) instead of the [M, 2D] gate pre-activations (no 268 MB written and read back at C3).
    // PG_GATE_U (training forward): the same scores AND the pre-activations U, bias added, in this interleaved column order (the
    // backward pass reads them: murcl_gated_score_bwd with `interleaved`); DROP: the two gate Dropouts (clam.py:47-48) applied to
    // tanh(a_d) / sigmoid(b_d) from the seeds, as murcl_gated_score_fwd does.
    constexpr bool GATE = (EPI == PG_GATE || EPI == PG_GATE_U);
    constexpr bool BIASED = (EPI == PG_BIAS_RELU || EPI == PG_BIAS || GATE);
    constexpr int NB = MASKED ? 1 : 0;                              // mask LDS-DMA op (128 or 256 B per wave)
    constexpr int NR = (EPI == PG_RANK1_MASK) ? 1 : 0;              // rowscale LDS-DMA op
    constexpr bool DMAH = PAD && PG_DMA_HALF != 0;                 // one half of the workgroup issues every A piece
    constexpr int GA = DMAH ? 2 * GT : GT;      // A pieces per tile of an issuing wave
    constexpr int G = GA + NB + NR;             // counted loads per tile per (issuing) wave
    constexpr int NMS = BM_OUT ? WN / 32 : 0;   // mask stores per tile per wave
    constexpr int S = (PG_ABL & 4) ? 0 : (EPI == PG_GATE ? 1 : NS + NMS + (EPI == PG_GATE_U ? 1 : 0));   // counted stores per tile per wave
    // LDS carve
    constexpr int OFF_STG = PG_NSLOT * SLOT;
    constexpr int OFF_BM = OFF_STG + PG_NW * PG_TR * STG_LD;                    // [slot][wave][256 B]
    constexpr int OFF_RS = OFF_BM + (NB ? PG_NSLOT * PG_NW * 256 : 0);          // [slot][64 f32]: every wave copies the same 32 row scales
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef PG_STAMPS
    unsigned* stamp_lds = (unsigned*)(smem + OFF_RS + (NR ? PG_NSLOT * 256 : 0));
    const int stamp_w = (blockIdx.x < PG_STAMP_WG) ? ((threadIdx.x >> 6) == 0 ? 0 : ((threadIdx.x >> 6) == 4 ? 1 : -1)) : -1;
    for (int i = threadIdx.x; i < PG_STAMP_BYTES / 4; i += 64 * PG_NW) stamp_lds[i] = 0u;
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q4 = lane >> 4, r16 = lane & 15;
    const unsigned lds0 = lds_off(smem);
    const int panels = N / NP;
    const int b = blockIdx.x;
    const int panel = (b >> 3) % panels;
    const int streams = gridDim.x / panels;
    const int stream = (b & 7) + 8 * (b / (8 * panels));
    const int n_tiles = M / PG_TR;
    // K = 512: tiles dealt round-robin (tile = stream + seq*streams), so the two panels of a row tile run side by side.
    // RANK1: a workgroup takes a CONTIGUOUS run of tiles - dealt round-robin every tile would land in another bag and
    // reload the per-bag rank-1 vector (compiler-visible loads that drain the LDS-DMA ring) once per tile.
    constexpr bool CONTIG = (EPI == PG_RANK1_MASK);
    const int per = (n_tiles + streams - 1) / streams;
    const int tile0 = CONTIG ? stream * per : stream, tstep = CONTIG ? 1 : streams;
    const int my_tiles = CONTIG ? min(per, n_tiles - tile0) : (n_tiles - stream + streams - 1) / streams;
    if (stream >= streams || my_tiles <= 0) {
        if (!GATE && colsum_part && stream < streams)   // no tiles: this workgroup's row of partial sums is zero
            for (int c = threadIdx.x; c < NP; c += 64 * PG_NW) colsum_part[(size_t)stream * N + panel * NP + c] = 0.f;
        return;
    }
    const int n0 = panel * NP + wave * WN;          // first column of this wave

    // LDS-DMA piece p of tile seq's loads: p < GT a 1 KiB piece of the A tile, then (MASKED) this wave's mask blocks, then
    // (RANK1) the tile's row scales.  One instruction each; the loop spreads them over the k-groups of the MFMA phase.
    auto issue_piece = [&](int seq, int p) {
        const int tix = tile0 + seq * tstep;
        const bool rev = ((walk_reverse & 1) != 0) != ((PG_FLIP >> EPI) & 1);       // PG_FLIP: A/B bit mask per epilogue
        const int row0 = (rev ? n_tiles - 1 - tix : tix) * PG_TR;
        const int sl = seq % PG_NSLOT;
        if (p < GA) {
            const int j = p;
            const char* base = (const char*)(A + (size_t)row0 * K);
            if (PAD) {
                // wave-uniform row: scalar base, lane offset lane*16.  DMAH: the issuing half covers the tile in pieces of NW/2 rows
                const int row = DMAH ? j * (PG_NW / 2) + (wave & (PG_NW / 2 - 1)) : j * PG_NW + wave;
                if (walk_reverse & 2) glds16_u_nt(base + (size_t)row * ROWB, lane * 16, lds0 + sl * SLOT + row * PADB);
                else glds16_u(base + (size_t)row * ROWB, lane * 16, lds0 + sl * SLOT + row * PADB);
            } else {
                const int ci = (j * PG_NW + wave) * 64 + lane;
                const int row = ci / CPR, pos = ci % CPR;
                glds16(base + (size_t)row * ROWB + ((pos ^ (row & 15)) << 4),
                       lds0 + sl * SLOT + (j * PG_NW + wave) * 1024);
            }
        } else if (MASKED && p == GA) {
            // this wave's mask blocks of the tile (WN/32 blocks of 128 B, contiguous): one 4-byte piece per lane
            const int nbytes = (WN / 32) * 128;
            const int off = min(lane * 4, nbytes - 4);
            glds4(bm_in + ((size_t)(row0 / PG_TR) * (N / 32) + (n0 >> 5)) * 128 + off,
                  lds0 + OFF_BM + (sl * PG_NW + wave) * 256);
        } else if (EPI == PG_RANK1_MASK && p == GA + NB) {
            glds4(rowscale + row0 + (lane & 31), lds0 + OFF_RS + sl * 256);
        }
    };
    // does this wave issue A pieces?  (the side pieces - mask words, row scales - are per wave either way)
    const bool dma_wave = (!DMAH || ((wave < PG_NW / 2) == (PG_DMA_HALF == 1))) && !((PG_ABL & 32) && panel != 0);   // ABL 32: only panel 0 loads A
    auto issue = [&](int seq) {
#pragma unroll
        for (int p = 0; p < G; ++p)
            if (p >= GA || dma_wave) issue_piece(seq, p);
    };

    // Register staging (PG_REGSTAGE, forward variants): an LDS-DMA instruction holds the issuing wave for 100-190 cycles (480-640
    // per tile and wave, in-kernel stamps of round 3) wherever it is placed; a plain global_load_dwordx4 issues in a few cycles and
    // a ds_write_b128 in ~13.  Tile t+3's rows are loaded into one of two register sets at the top of iteration t, written to
    // LDS slot (t+1) % NSLOT... one tile ahead at the top of iteration t (tile t+1), so three tiles are on their way as before.
    auto row0_of_early = [&](int seq) {
        const int tix = tile0 + seq * tstep;
        const bool rev = ((walk_reverse & 1) != 0) != ((PG_FLIP >> EPI) & 1);
        return (rev ? n_tiles - 1 - tix : tix) * PG_TR;
    };
    constexpr bool RS = PG_REGSTAGE != 0 && PAD && (EPI == PG_BIAS_RELU || EPI == PG_BIAS);
    u32x4 rs_reg[2][RS ? GT : 1];
    auto rs_load = [&](auto par, int seq) {
        constexpr int P = decltype(par)::value;
        const char* base = (const char*)(A + (size_t)row0_of_early(seq) * K);
#pragma unroll
        for (int j = 0; j < GT; ++j) {
            const int row = j * PG_NW + wave;
            rs_reg[P][j] = (walk_reverse & 2) ? pg_gload16_nt(base + (size_t)row * ROWB, lane * 16) : pg_gload16(base + (size_t)row * ROWB, lane * 16);
        }
    };
    auto rs_write = [&](auto par, int seq) {
        constexpr int P = decltype(par)::value;
        char* slot = smem + (seq % PG_NSLOT) * SLOT;
#pragma unroll
        for (int j = 0; j < GT; ++j) {
            const int row = j * PG_NW + wave;
            asm volatile("" : "+v"(rs_reg[P][j]));
            *(u32x4*)(slot + row * PADB + lane * 16) = rs_reg[P][j];
        }
    };
    // Weight prologue (WPRO, K = 512): a fragment-shaped global load (16 rows x 4 pieces of 16 B, 256 B apart) touches 64 cache
    // lines per instruction, and the NJ * NKK of them a wave needs pull every line of its slice through the CU's memory pipe up to
    // eight times.  Instead each 16-row block of the slice travels as 16 whole-row LDS-DMA pieces into the wave's private 1/NW of
    // the tile ring (not yet in use: the first tiles are requested after the fragments have been read back).  Measured on the K2
    // forward, which has the same prologue: 62.8 -> 55.1 us per launch (profiles/r03_m).
    constexpr bool WPRO_C = PG_WPRO != 0 && PAD && !RS && PG_NW * 16 * PADB <= PG_NSLOT * SLOT;
    // FRAGMENT-ORDER weights (round 6; walk_reverse bit 2, K = 512): W arrives as the fragments themselves (csrc/elementwise.hip
    // frag_index): one coalesced 1-KiB load per k-step straight into registers, the first tiles requested BEFORE them
    constexpr bool wfrag = PAD && WFRAG;                 // (a template parameter: as a run-time branch the two prologues' fragment registers spilled)
    constexpr bool WPRO = WPRO_C && !wfrag;
    const int pre = (RS || WPRO) ? 0 : min(3, my_tiles);
    for (int s = 0; s < pre; ++s) issue(s);
    if (RS) rs_load(std::integral_constant<int, 0>{}, 0);

    // ---- W slice: MFMA "a" operands, rows n = n0 + 16j + r16
    bf16x8 wf[NJ][NKK];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const bf16_t* wrow = W + (size_t)(n0 + 16 * j + r16) * K;
        if constexpr (wfrag) {
            const char* fblk = (const char*)W + ((size_t)((n0 >> 4) + j) * NKK) * 1024 + lane * 16;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) wf[j][kk] = *(const bf16x8*)(fblk + kk * 1024);
            continue;
        }
        if (WPRO) {
            const char* wblk = (const char*)(W + (size_t)(n0 + 16 * j) * K);
            const unsigned stage = lds0 + wave * 16 * PADB;
#pragma unroll
            for (int u = 0; u < 16; ++u) glds16_u(wblk + (size_t)u * ROWB, lane * 16, stage + u * PADB);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const char* fb = smem + (wave * 16 + r16) * PADB + NKK * q4 * 16;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                wf[j][kk] = *(const bf16x8*)(fb + kk * 16);
                asm volatile("" : "+v"(wf[j][kk]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the next block's pieces overwrite these rows
            continue;
        }
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            // k assignment of lane quarter q4 in k-step kk: PAD -> 16-byte chunk (kk + NKK*q4); else chunk (4kk + q4)
            wf[j][kk] = *(const bf16x8*)(wrow + (PAD ? (kk + NKK * q4) * 8 : 32 * kk + 8 * q4));
            // opaque to the optimiser: otherwise hipcc rematerialises the fragments by re-loading them from global
            // memory inside the tile loop (34 loads + vmcnt waits per tile that also drain the LDS-DMA ring)
            asm volatile("" : "+v"(wf[j][kk]));
        }
    }
    // bias for accumulator-layout columns n0 + 16j + 4q4 + r
    float bias_r[BIASED ? NJ : 1][4];
    if (BIASED) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bias_r[j][r] = bias[n0 + 16 * j + 4 * q4 + r];
                // pinned here so that hipcc waits for these loads now: left alone it puts its s_waitcnt vmcnt(0) at
                // the first use, inside the tile loop, where it drains the LDS-DMA ring on every tile
                asm volatile("" : "+v"(bias_r[j][r]));
            }
    }
    float gw[4] = {0.f, 0.f, 0.f, 0.f};              // PG_GATE: wc of this lane's four (a_d, b_d) pairs
    float gbc = 0.f;                                 // attention_c's bias, carried by the first 32-column group's partial score
    if (GATE) {
        if (rowscale && n0 == 0) gbc = rowscale[0];
        asm volatile("" : "+v"(gbc));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gw[r] = rank1[n0 + 4 * q4 + r];
            asm volatile("" : "+v"(gw[r]));
        }
    }
    float gsc[2] = {0.f, 0.f};                       // PG_GATE: the tile's two 16-row partial scores, until they are stored together
    float rk[(EPI == PG_RANK1_MASK) ? NJ : 1][4];
    // PG_RANK1_MASK with `bias` = the [bags][2] soft-max statistics (m, l) of the pooling pass: `rowscale` then holds the RAW scores
    // and the row scale A = exp(s - m) / (l sqrt(rows_per_bag)) is formed here (abmil.py:40-41) - two v_exp per lane and tile - so
    // that no pass has to write and re-read the normalised attention rows (round 6)
    float bag_m = 0.f, bag_inv = 1.f;
    int cur_bag = -1;
    float csum[NJ][4];                       // column sums in accumulator layout: column 16j + 4q4 + r, over this lane's rows
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) csum[j][r] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (wfrag) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) asm volatile("" : "+v"(wf[j][kk]));      // resident from here on: never re-loaded in the loop
    }
    if (WPRO) {
        LDS_BARRIER();                           // every wave has read its fragments back: the ring is free for tiles
        const int pre_w = min(3, my_tiles);
        for (int s = 0; s < pre_w; ++s) issue(s);
    }
    if (RS) {
        rs_write(std::integral_constant<int, 0>{}, 0);
        if (my_tiles > 1) rs_load(std::integral_constant<int, 1>{}, 1);
        if (my_tiles > 2) rs_load(std::integral_constant<int, 0>{}, 2);
    }

    char* stg = smem + OFF_STG + wave * (PG_TR * STG_LD);        // wave-private staging patch
    const int crow = lane / CPW, cchunk = lane % CPW;            // row-wise phase: row RPI*g + crow, chunk cchunk

    // ---- the two phases of a tile, as lambdas so that the two halves of the workgroup can run them in opposite order
    auto row0_of = [&](int seq) {
        const int tix = tile0 + seq * tstep;
        const bool rev = ((walk_reverse & 1) != 0) != ((PG_FLIP >> EPI) & 1);       // PG_FLIP: A/B bit mask per epilogue
        return (rev ? n_tiles - 1 - tix : tix) * PG_TR;
    };
    // side inputs of the epilogue that live in the tile's LDS slot: this lane's mask words (one per 32-column block
    // of the wave) and the two row scales it needs
    auto load_side = [&](int seq, unsigned (&mw)[NJ / 2], float (&am)[2]) {
        const int sl = seq % PG_NSLOT;
#pragma unroll
        for (int bq = 0; bq < NJ / 2; ++bq) mw[bq] = 0u;
        am[0] = am[1] = 0.f;
        if (MASKED) {
            const uint16_t* bml = (const uint16_t*)(smem + OFF_BM + (sl * PG_NW + wave) * 256);
#pragma unroll
            for (int bq = 0; bq < NJ / 2; ++bq) mw[bq] = bml[bq * 64 + lane];
        }
        if (EPI == PG_RANK1_MASK) {
            const float* rs = (const float*)(smem + OFF_RS + sl * 256);
            am[0] = rs[r16];
            am[1] = rs[16 + r16];
        }
    };
    auto mfma_phase = [&](int seq, f32x4 (&acc)[2][NJ], auto&& hook) {
        const char* tile = smem + (seq % PG_NSLOT) * SLOT;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                // the bias is the accumulators' start value: no add in the epilogue (a move either way)
                if (BIASED) acc[i][j] = f32x4{bias_r[j][0], bias_r[j][1], bias_r[j][2], bias_r[j][3]};
                else acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        const char* hb = tile + r16 * PADB + NKK * q4 * 16;               // PAD: base + immediates only
        if (PAD) {
            // explicit software pipeline: the fragments of k-group g+PG_PF are requested before the MFMAs of group g
            // are issued (left alone hipcc keeps only two ds_read_b128 in flight and the MFMAs wait on LDS latency)
            constexpr int GK = PG_GK, NG = NKK / GK, D = PG_PF, NBUF = D + 1;
            bf16x8 hq[NBUF][GK][2];
            auto load_grp = [&](int g, int buf) {
                if ((PG_ABL & 1) && g > 0) {
#pragma unroll
                    for (int k2 = 0; k2 < GK; ++k2)
#pragma unroll
                        for (int i = 0; i < 2; ++i) hq[buf][k2][i] = hq[0][k2][i];
                    return;
                }
#pragma unroll
                for (int k2 = 0; k2 < GK; ++k2)
#pragma unroll
                    for (int i = 0; i < 2; ++i) hq[buf][k2][i] = *(const bf16x8*)(hb + i * 16 * PADB + (g * GK + k2) * 16);
            };
#pragma unroll
            for (int g = 0; g < D; ++g) load_grp(g, g);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + D < NG) load_grp(g + D, (g + D) % NBUF);
                __builtin_amdgcn_sched_barrier(0);
                if (PG_PRIO == 1) __builtin_amdgcn_s_setprio(1);
                if (PG_PRIO == 2) { if (((g & 1) != 0) == (wave >= PG_NW / 2)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#pragma unroll
                for (int k2 = 0; k2 < GK; ++k2)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            if (PG_ABL & 8) asm volatile("" : "+v"(acc[i][j]) : "v"(hq[g % NBUF][k2][i]));
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][g * GK + k2], hq[g % NBUF][k2][i], acc[i][j], 0, 0, 0);
                        }
                if (PG_PRIO == 1) __builtin_amdgcn_s_setprio(0);
                hook(g);                         // same scheduling region as the MFMAs above: VALU / LDS work rides under them
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = 16 * i + r16, c = 4 * kk + q4;
                    const bf16x8 h = *(const bf16x8*)(tile + row * ROWB + ((c ^ (row & 15)) << 4));
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][kk], h, acc[i][j], 0, 0, 0);
                }
                hook(kk);
            }
        }
    };
    // part: -1 = everything; 0 / 1 = accumulator math + staging writes of row half i = part (1 also stores the mask
    // words); 2 + g = row-wise store g.  The fused schedule runs the parts of tile t-1 between the k-groups of tile t.
    auto epilogue = [&](int seq, f32x4 (&acc)[2][NJ], unsigned (&mw)[NJ / 2], float (&am)[2], int part = -1) {
        if (PG_ABL & 16) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(acc[i][j]));
            return;
        }
        const int row0 = row0_of(seq);
        if constexpr (GATE) {
            static_assert(!GATE || NJ == 2, "PG_GATE pairs the wave's two 16-column blocks");
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                // (fused schedule of PG_GATE_U: with the second block of the row half)
                if (EPI == PG_GATE_U && part >= 0 && part != 10 + NJ * i + 1) continue;
                unsigned ka4 = 0u, kb4 = 0u;
                if (DROP) {
                    const long flat4 = (long)(row0 + 16 * i + r16) * (N >> 1) + 16 * (n0 >> 5) + 4 * q4;
                    ka4 = pg_keep4(drop.seed_a, flat4);
                    kb4 = pg_keep4(drop.seed_b, flat4);
                }
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * acc[i][1][r]));
                    float u = fast_tanh(acc[i][0][r]) * sg * gw[r];
                    if (DROP) {
                        const bool keep = ((ka4 >> (8 * r)) & 255u) < drop.thresh && ((kb4 >> (8 * r)) & 255u) < drop.thresh;
                        u = keep ? u * (drop.scale * drop.scale) : 0.f;
                    }
                    t += u;
                }
                gsc[i] = quarters_sum(t) + gbc;
            }
            if (part < 0 || part == 10 + NJ + 1) {
                // ONE full-wave store per tile, no divergent branch inside the MFMA loop's scheduling region: every quarter carries
                // both sums; even quarters write rows 0-15, odd quarters rows 16-31 (the two copies of a row hit the same address
                // with the same value)
                float* sp = colsum_part + (size_t)(n0 >> 5) * M + row0;       // partial scores [N/32][M]
                pg_store4(sp + 16 * (q4 & 1) + r16, (q4 & 1) ? gsc[1] : gsc[0]);
            }
            if (EPI == PG_GATE) return;
        }
        unsigned ones = 0x00010001u;
        asm volatile("" : "+v"(ones));
        // ---- accumulator-layout math: lane holds row 16i+r16, columns 16j+4q4+r
        if (EPI == PG_RANK1_MASK && (part <= 0 || part == 10)) {
            const int bag = row0 / rows_per_bag;
            if (bag != cur_bag) {                                  // rare: compiler-visible loads, drains once per bag
                cur_bag = bag;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) rk[j][r] = rank1[(size_t)bag * N + n0 + 16 * j + 4 * q4 + r];
                if (bias) {
                    bag_m = bias[2 * bag];
                    bag_inv = 1.0f / (sqrtf((float)rows_per_bag) * bias[2 * bag + 1]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (part >= 0 && part != i && (part < 10 || (part - 10) / NJ != i)) continue;
            const int row = 16 * i + r16;
            const float a_m = (EPI == PG_RANK1_MASK && bias) ? fast_exp(am[i] - bag_m) * bag_inv : am[i];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (part >= 10 && (part - 10) % NJ != j) continue;       // 10 + NJ*i + j: one 16x16 block of the tile
                f32x4 v = acc[i][j];
                // (PG_BIAS / PG_BIAS_RELU: the bias is already in the accumulator, the ReLU is applied to the packed pairs below)
                if (EPI == PG_RANK1_MASK) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += a_m * rk[j][r];
                }
                if (MASKED) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = 8 * i + 4 * (j & 1) + r;
                        const int keep = __builtin_amdgcn_sbfe((int)mw[j >> 1], (7 - (idx >> 1)) + 8 * (idx & 1), 1);   // 0 / -1
                        v[r] = __uint_as_float(__float_as_uint(v[r]) & (unsigned)keep);
                    }
                }
                if (DROP && EPI == PG_BIAS_RELU) {   // Dropout behind the ReLU (clam.py:69-72): the flags below record what survives
                    const unsigned k4 = pg_keep4(drop.seed_a, (long)(row0 + row) * N + n0 + 16 * j + 4 * q4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = ((k4 >> (8 * r)) & 255u) < drop.thresh ? v[r] * drop.scale : 0.f;
                }
                if (MASKED && colsum_part) {         // bias gradient from the f32 values, before they are rounded to bf16 (backward variants only)
#pragma unroll
                    for (int r = 0; r < 4; ++r) csum[j][r] += v[r];
                }
                unsigned w0 = pack_bf2(v[0], v[1]), w1 = pack_bf2(v[2], v[3]);
                if (EPI == PG_BIAS_RELU) { w0 = pg_relu_pk(w0); w1 = pg_relu_pk(w1); }
                if (BM_OUT) {
                    const unsigned f0 = (EPI == PG_BIAS_RELU) ? pg_pos_flags_nonneg(w0, ones) : pg_pos_flags(w0, ones);
                    const unsigned f1 = (EPI == PG_BIAS_RELU) ? pg_pos_flags_nonneg(w1, ones) : pg_pos_flags(w1, ones);
                    mw[j >> 1] = (mw[j >> 1] << 1) | f0;
                    mw[j >> 1] = (mw[j >> 1] << 1) | f1;
                }
                *(u32x2*)(stg + row * STG_LD + (16 * j + 4 * q4) * 2) = u32x2{w0, w1};
            }
        }
        if (BM_OUT && (part < 0 || part == 1 || part == 10 + 2 * NJ - 1)) {
            uint8_t* blk = bm_out + ((size_t)(row0 / PG_TR) * (N / 32) + (n0 >> 5)) * 128;
#pragma unroll
            for (int bq = 0; bq < NJ / 2; ++bq) pg_store2(blk + bq * 128 + lane * 2, mw[bq] | (mw[bq] >> 8));
        }
        // ---- row-wise phase: lane owns 8 consecutive columns (one 16-byte chunk) of RPI rows per pass
#pragma unroll
        for (int g = 0; g < NS; ++g) {
            if (part >= 0 && part != 2 + g) continue;
            const int row = RPI * g + crow;
            const u32x4 u = *(const u32x4*)(stg + row * STG_LD + cchunk * 16);
            pg_store16(C + (size_t)(row0 + row) * N + n0 + cchunk * 8, u);
        }
    };

    // Phase rotation (PG_ROTATE, off: the step got 4 % SLOWER with it, 1.70 -> 1.77 ms): the two waves of a SIMD (waves w and w + NW/2) meet at every tile barrier, so left alone they
    // fight over the matrix pipe during the MFMA phase and over the VALU / LDS during the epilogue.  The upper half of
    // the workgroup therefore runs "epilogue of the previous tile, then MFMAs of this tile" between two barriers while
    // the lower half runs "MFMAs, then epilogue" of this tile: at any time one wave of a SIMD feeds the matrix cores
    // and its partner rounds, transposes and stores.
    // Fused schedule (PG_FUSE, K = 512): every wave keeps the accumulators of tile t-1 and runs its epilogue in pieces
    // between the k-groups of tile t, so VALU / LDS-staging / store work issues in the shadow of the MFMAs instead
    // of after them with the matrix pipe idle.  Store counts lag by one tile, exactly like the rotated half.
    constexpr bool FUSE = PAD && (((PG_FUSE & 1) && (EPI == PG_BIAS_RELU || EPI == PG_BIAS || (PG_GATE_U_FUSE && EPI == PG_GATE_U))) || ((PG_FUSE & 2) && EPI == PG_MASK));   // (the K = 512 rank-1 variant would spill)
    constexpr bool ROT = PG_ROTATE != 0 && !FUSE;
    const bool late = FUSE || (ROT && wave >= PG_NW / 2);
    f32x4 acc[2][NJ];
    unsigned mw[NJ / 2];
    float am[2];
    // (measured in round 2: a static s_setprio(1) for the second-dispatched half of the workgroup changes nothing here)
    static_assert(!RS || FUSE, "register staging is counted for the fused forward schedule");
    auto iteration = [&](auto par, int seq) {
        PG_STAMP(0);
        PG_STAMP(6);
        if (RS) {
            // tile seq+1 sits in register set !par (loaded two iterations ago): wait for it, hand it to LDS slot (seq+1) % NSLOT (the
            // tile read there, seq-3, is long done), and send the same registers for tile seq+3.  Operations younger than tile
            // seq+1's loads: the stores of iteration seq-2 (tile seq-3's epilogue), the loads and stores of iteration seq-1.
            constexpr int Q = 1 - decltype(par)::value;
            if (seq + 1 < my_tiles) {
                if (seq + 3 < my_tiles) {
                    if (seq <= 1) { PG_WAIT(GT); } else if (seq == 2) { PG_WAIT(GT + S); } else { PG_WAIT(GT + 2 * S); }
                } else {
                    PG_WAIT(0);
                }
                rs_write(std::integral_constant<int, Q>{}, seq + 1);
                if (seq + 3 < my_tiles) rs_load(std::integral_constant<int, Q>{}, seq + 3);
            }
        } else
        // ops issued after tile seq's loads: 2 more tiles' loads plus the stores of the iterations in between (the late
        // half issues the stores of a tile one iteration later: its counts lag by one)
        if (seq + 2 < my_tiles) {
            const int sq = late ? seq - 1 : seq;
            if (dma_wave) {
                // (from the fourth tile on, the stores issued in the iteration that requested this tile are younger than its loads
                // too: PG_EXACT_WAIT lets them stay in flight - three tiles of stores instead of two)
                if (sq <= 0) { PG_WAIT(2 * G); } else if (sq == 1) { PG_WAIT(2 * G + S); }
                else if (sq == 2 || !PG_EXACT_WAIT) { PG_WAIT(2 * G + 2 * S); } else { PG_WAIT(2 * G + 3 * S); }
            } else {          // (DMAH) this wave's queue holds only its side pieces and stores
                constexpr int G0 = G - GA;
                if (sq <= 0) { PG_WAIT(2 * G0); } else if (sq == 1) { PG_WAIT(2 * G0 + S); } else { PG_WAIT(2 * G0 + 2 * S); }
            }
        } else {
            PG_WAIT(0);
        }
        PG_STAMP(1);
        LDS_BARRIER();
        PG_STAMP(2);
        // The loads of tile seq+3 (its ring slot is free since the barrier above) are NOT issued here in one burst: 8 waves x
        // G pieces of 1 KiB queue on the CU's one address unit (64 B/clk: ~0.5 k cycles per tile with every matrix pipe idle,
        // profiles/r03_b_inkernel_stamps_panel_k2_before.txt).  Each wave issues one piece per k-group of its MFMA phase
        // instead; the LAST A piece stays the last VMEM operation of the iteration's loads, so the hand counts of
        // s_waitcnt vmcnt above are unchanged (stores of the fused epilogue that precede it in program order are older ops).
        const bool more = !RS && !(PG_ABL & 2) && seq + 3 < my_tiles;
#if PG_SPREAD_DMA
        constexpr int NGRP_ = PAD ? NKK / PG_GK : NKK;
        auto dma = [&](int g) {
            if (!more) return;
            // piece q of G goes with k-group NGRP_-1 - (G-1-q)*NGRP_/G: spread over the tile's k-groups, right-aligned; side
            // pieces (mask, row scales) first, then the A pieces, the last A piece in the last k-group (after every store the
            // fused epilogue issues)
            const int gg = (PG_DMA_SKEW && wave >= PG_NW / 2) ? g + 1 : g;
#pragma unroll
            for (int q = 0; q < G; ++q) {
                if (NGRP_ - 1 - ((G - 1 - q) * NGRP_) / G != gg) continue;
                const int pc = q < G - GA ? GA + q : q - (G - GA);
                if (pc >= GA || dma_wave) issue_piece(seq + 3, pc);
            }
        };
#else
        if (more) issue(seq + 3);
        auto dma = [](int) {};
#endif
        PG_STAMP(3);
        if (FUSE) {
            // acc/mw/am still hold tile seq-1; the new tile accumulates into accn and is handed over at the end
            f32x4 accn[2][NJ];
            unsigned mwn[NJ / 2];
            float amn[2];
            load_side(seq, mwn, amn);
            const bool prev = seq > 0;
            mfma_phase(seq, accn, [&](int g) {
                dma(g);
                if (!prev) return;
                // 8 k-groups per tile: math of row half 0, its store, math of row half 1 (+ mask words), its store(s)
                constexpr int NGRP = NKK / PG_GK;
                static_assert(!FUSE || NJ != 2 || NGRP >= 4, "the fused schedule needs four k-groups per tile");
                if (NJ == 2 && NGRP >= 6) {   // one 16x16 block per k-group, the row-wise stores as soon as their rows are staged
                    if (g == 0) epilogue(seq - 1, acc, mw, am, 10);
                    if (g == 1) epilogue(seq - 1, acc, mw, am, 11);
                    if (g == 2) epilogue(seq - 1, acc, mw, am, 2);
                    if (g == 3) epilogue(seq - 1, acc, mw, am, 12);
                    if (g == 4) epilogue(seq - 1, acc, mw, am, 13);
                    if (g == 5) epilogue(seq - 1, acc, mw, am, 3);
                } else if (NJ == 2) {         // four (longer) k-groups: a row half per group, its store in the next
                    if (g == 0) { epilogue(seq - 1, acc, mw, am, 10); epilogue(seq - 1, acc, mw, am, 11); }
                    if (g == 1) epilogue(seq - 1, acc, mw, am, 2);
                    if (g == 2) { epilogue(seq - 1, acc, mw, am, 12); epilogue(seq - 1, acc, mw, am, 13); }
                    if (g == 3) epilogue(seq - 1, acc, mw, am, 3);
                } else {
                    if (g == 0) epilogue(seq - 1, acc, mw, am, 0);
                    if (g == 2) epilogue(seq - 1, acc, mw, am, 2);
                    if (g == 3) epilogue(seq - 1, acc, mw, am, 1);
                    if (g == 5) epilogue(seq - 1, acc, mw, am, 3);
                    if (NS > 2 && g == 6) { epilogue(seq - 1, acc, mw, am, 4); epilogue(seq - 1, acc, mw, am, 5); }
                }
            });
            PG_STAMP(4);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = accn[i][j];
#pragma unroll
            for (int bq = 0; bq < NJ / 2; ++bq) mw[bq] = mwn[bq];
            am[0] = amn[0];
            am[1] = amn[1];
        } else if (!late) {
            load_side(seq, mw, am);          // requested ahead of the MFMA loop
            mfma_phase(seq, acc, dma);
            PG_STAMP(4);
            epilogue(seq, acc, mw, am);
        } else {
            if (seq > 0) epilogue(seq - 1, acc, mw, am);
            PG_STAMP(4);
            load_side(seq, mw, am);
            mfma_phase(seq, acc, dma);
        }
        PG_STAMP(5);
    };
    if (RS) {
        for (int seq = 0; seq < my_tiles; seq += 2) {
            iteration(std::integral_constant<int, 0>{}, seq);
            if (seq + 1 < my_tiles) iteration(std::integral_constant<int, 1>{}, seq + 1);
        }
    } else {
        for (int seq = 0; seq < my_tiles; ++seq) iteration(std::integral_constant<int, 0>{}, seq);
    }
    if (late) epilogue(my_tiles - 1, acc, mw, am);
#ifdef PG_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (blockIdx.x < PG_STAMP_WG)
        for (int i = threadIdx.x; i < PG_STAMP_BYTES / 4; i += 64 * PG_NW) (&g_pg_stamps[blockIdx.x][0][0][0])[i] = stamp_lds[i];
#endif

    if (MASKED && colsum_part) {
        // the 16 lanes of a quarter hold the same columns for different rows: reduce over them and publish this
        // workgroup's partial sums as row `stream` of colsum_part [streams][N] (plain stores; murcl_colsum adds the rows
        // up afterwards - 128 atomic adders per column from here cost 10-15 us per launch)
        float* prow = colsum_part + (size_t)stream * N;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = row16_sum(csum[j][r]);
                if (r16 == 0) prow[n0 + 16 * j + 4 * q4 + r] = t;
            }
    }
}

// workgroups of a launch: one per CU, fewer (a multiple of 8 per column panel) when there are not enough row tiles
static int pg_grid(int M, int panels) {
    int grid = murcl_cu_budget() / (8 * panels) * (8 * panels);      // (the block -> (panel, stream) map wants whole groups of 8 per panel)
    const int n_tiles = M / PG_TR;
    if (n_tiles * panels < grid) grid = ((n_tiles * panels + 8 * panels - 1) / (8 * panels)) * 8 * panels;
    return grid;
}

template <int K, int WN, int PG_NW, int EPI, bool BM_OUT, bool DROP = false, bool WFRAG = false>
static int pg_launch(const bf16_t* A, const bf16_t* W, bf16_t* C, int M, int N, const float* bias, uint8_t* bm_out,
                     const uint8_t* bm_in, const float* rowscale, const float* rank1, int rows_per_bag,
                     float* colsum_part, int* streams_out, int walk_reverse, hipStream_t s, PgDrop drop = PgDrop{0ull, 0ull, 0u, 1.f}) {
    constexpr int SLOT = PG_TR * (K == 512 ? K * 2 + 16 : K * 2);
    constexpr int STG_LD = WN * 2 + 16;
    constexpr int LDS = PG_NSLOT * SLOT + PG_NW * PG_TR * STG_LD +
                        ((EPI == PG_MASK || EPI == PG_RANK1_MASK) ? PG_NSLOT * PG_NW * 256 : 0) +
                        (EPI == PG_RANK1_MASK ? PG_NSLOT * 256 : 0) + PG_STAMP_BYTES;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto k = panel_nt_kernel<K, WN, PG_NW, EPI, BM_OUT, DROP, WFRAG>;
    static MurclOncePerDevice once;      
    if (once.first()) {
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
                       
    }
    const int panels = N / (PG_NW * WN);
    const int grid = pg_grid(M, panels);
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * PG_NW), LDS, s, A, W, C, M, N, bias, bm_out, bm_in, rowscale, rank1,
                       rows_per_bag, colsum_part, walk_reverse, drop);
    *streams_out = grid / panels;
    return MURCL_CHECK_LAUNCH();
}

extern "C" int murcl_colsum(const void* x, float* out, int R, int N, int ld, int dtype, int accumulate, hipStream_t s);

// C-ABI: see include/murcl_amd.h
extern "C" int murcl_panel_gemm_supported(int M, int N, int K, int epilogue, int rows_per_bag) {
    if (M <= 0 || M % PG_TR) return 0;
    if (K == 512)
        return (N % 256 == 0) && (epilogue == PG_BIAS_RELU || epilogue == PG_MASK || epilogue == PG_BIAS || epilogue == PG_GATE ||
                                  epilogue == PG_GATE_U ||
                                  (epilogue == PG_RANK1_MASK && rows_per_bag > 0 && rows_per_bag % PG_TR == 0));
    if (K == 128) return N == 512 && epilogue == PG_RANK1_MASK && rows_per_bag > 0 && rows_per_bag % PG_TR == 0;
    return 0;
}

extern "C" int murcl_panel_gemm_colsum_rows(int M, int N, int K, int epilogue) {
    if (M <= 0 || N <= 0) return 0;
    const int panels = K == 128 ? N / 512 : N / 256;         // columns per workgroup: 8 waves x 64 (K = 128) / 256 (K = 512)
    if (panels <= 0) return 0;
    (void)epilogue;
    return pg_grid(M, panels) / panels;
}

extern "C" int murcl_panel_gemm_drop(const void* A, const void* W, void* C, int M, int N, int K, int epilogue,
                                     const float* bias, void* bitmask_out, const void* bitmask_in, const float* rowscale,
                                     const float* rank1, int rows_per_bag, float* colsum_out, int colsum_accumulate,
                                     float* colsum_ws, int walk_reverse, float keep_p, unsigned long long seed_a,
                                     unsigned long long seed_b, hipStream_t stream);
extern "C" int murcl_panel_gemm(const void* A, const void* W, void* C, int M, int N, int K, int epilogue,
                                const float* bias, void* bitmask_out, const void* bitmask_in, const float* rowscale,
                                const float* rank1, int rows_per_bag, float* colsum_out, int colsum_accumulate,
                                float* colsum_ws, int walk_reverse, hipStream_t stream) {
    return murcl_panel_gemm_drop(A, W, C, M, N, K, epilogue, bias, bitmask_out, bitmask_in, rowscale, rank1, rows_per_bag, colsum_out,
                                 colsum_accumulate, colsum_ws, walk_reverse, 0.f, 0ull, 0ull, stream);
}
// keep_p in (0, 1): Dropout(1 - keep_p) inside the epilogue - PG_BIAS_RELU (K = 512, with the bit mask): after the ReLU, mask of
// seed_a over [M, N]; PG_GATE_U: on tanh(a) and sigmoid(b), masks of seed_a / seed_b over [M, N/2].  Survivors are scaled by
// 256 / round(256 keep_p), as in murcl_dropout_mask.  Any other keep_p: no dropout.
extern "C" int murcl_panel_gemm_drop(const void* A, const void* W, void* C, int M, int N, int K, int epilogue,
                                     const float* bias, void* bitmask_out, const void* bitmask_in, const float* rowscale,
                                     const float* rank1, int rows_per_bag, float* colsum_out, int colsum_accumulate,
                                     float* colsum_ws, int walk_reverse, float keep_p, unsigned long long seed_a,
                                     unsigned long long seed_b, hipStream_t stream) {
    if (!murcl_panel_gemm_supported(M, N, K, epilogue, rows_per_bag)) return -1;
    PgDrop drop{seed_a, seed_b, 0u, 1.f};
    const bool dropping = keep_p > 0.f && keep_p < 1.f;
    if (dropping) {
        drop.thresh = (unsigned)(keep_p * 256.f + 0.5f);
        drop.scale = 256.f / (float)(drop.thresh ? drop.thresh : 1u);
        if (K != 512 || !((epilogue == PG_BIAS_RELU && bitmask_out) || epilogue == PG_GATE_U)) return -1;
    }
    if (colsum_out && !colsum_ws) return -1;
    // walk_reverse bit 2: W is in fragment order (csrc/elementwise.hip frag_index) - the plain K = 512 epilogues without Dropout only
    const bool wfrag = (walk_reverse & 4) != 0;
    if (wfrag && (K != 512 || dropping || !(epilogue == PG_BIAS_RELU || epilogue == PG_MASK))) return -1;
    float* part = colsum_ws;                 // without colsum_out: the partial rows are the result (the caller adds them up)
    int streams = 0, rc = -1;
    const bf16_t* a = (const bf16_t*)A;
    const bf16_t* w = (const bf16_t*)W;
    bf16_t* c = (bf16_t*)C;
    uint8_t* bo = (uint8_t*)bitmask_out;
    const uint8_t* bi = (const uint8_t*)bitmask_in;
    if (K == 512 && epilogue == PG_BIAS_RELU && dropping) {
        if (!bias) return -1;
        rc = pg_launch<512, PG_WIDE ? 64 : 32, PG_WIDE ? 4 : 8, PG_BIAS_RELU, true, true>(a, w, c, M, N, bias, bo, bi, rowscale, rank1, rows_per_bag, part, &streams, walk_reverse, stream, drop);
    } else if (K == 512 && epilogue == PG_GATE_U) {
        // C = the gate pre-activations [M, N] in the interleaved column order of W; rank1 / colsum_ws as for PG_GATE
        if (!bias || !rank1 || !colsum_ws || colsum_out || !c) return -1;
        rc = dropping ? pg_launch<512, 32, 8, PG_GATE_U, false, true>(a, w, c, M, N, bias, bo, bi, rowscale, rank1, rows_per_bag, part, &streams, walk_reverse, stream, drop)
                      : pg_launch<512, 32, 8, PG_GATE_U, false, false>(a, w, c, M, N, bias, bo, bi, rowscale, rank1, rows_per_bag, part, &streams, walk_reverse, stream);
    } else if (K == 512 && epilogue == PG_BIAS_RELU) {
        if (!bias) return -1;
        if (wfrag)
            rc = bo ? pg_launch<512, PG_WIDE ? 64 : 32, PG_WIDE ? 4 : 8, PG_BIAS_RELU, true, false, true>(a, w, c, M, N, bias, bo, bi, rowscale, rank1, rows_per_bag, part, &streams, walk_reverse, stream)
                    : pg_launch<512, PG_WIDE ? 64 : 32, PG_WIDE ? 4 : 8, PG_BIAS_RELU, false, false, true>(a, w, c, M, N, bias, bo, bi, rowscale, rank1, rows_per_bag, part, &streams, walk_reverse, stream);
        else
        rc = bo ? pg_launch<512, PG_WIDE ? 64 : 32, PG_WIDE ? 4 : 8, PG_BIAS_RELU, true>(a, w, c, M, N, bias, bo, bi, rowscale, rank1, rows_per_bag, part, &streams, walk_reverse, stream)
                : pg_launch<512, PG_WIDE ? 64 : 32, PG_WIDE ? 4 : 8, PG_BIAS_RELU, false>(a, w, c, M, N, bias, bo, bi, rowscale, rank1, rows_per_bag, part, &streams, walk_reverse, stream);
    } else if (K == 512 && epilogue == PG_MASK) {
        if (!bi) return -1;
        rc = wfrag ? pg_launch<512, PG_WIDE ? 64 : 32, PG_WIDE ? 4 : 8, PG_MASK, false, false, true>(a, w, c, M, N, bias, bo, bi, rowscale, rank1, rows_per_bag, part, &streams, walk_reverse, stream)
                   : pg_launch<512, PG_WIDE ? 64 : 32, PG_WIDE ? 4 : 8, PG_MASK, false>(a, w, c, M, N, bias, bo, bi, rowscale, rank1, rows_per_bag, part, &streams, walk_reverse, stream);
    } else if (K == 512 && epilogue == PG_BIAS) {
        if (!bias) return -1;
        rc = pg_launch<512, 32, 8, PG_BIAS, false>(a, w, c, M, N, bias, bo, bi, rowscale, rank1, rows_per_bag, part, &streams, walk_reverse, stream);
    } else if (K == 512 && epilogue == PG_GATE) {
        // rank1 = wc per interleaved column [N] f32, colsum_ws = the partial scores [N/32][M] f32 (the only output: C may be NULL);
        // rowscale (may be NULL) = one float, attention_c's bias, added to partial row 0
        if (!bias || !rank1 || !colsum_ws || colsum_out) return -1;
        rc = pg_launch<512, 32, 8, PG_GATE, false>(a, w, c, M, N, bias, bo, bi, rowscale, rank1, rows_per_bag, part, &streams, walk_reverse, stream);
    } else if (K == 512 && epilogue == PG_RANK1_MASK) {
        if (!bi || !rowscale || !rank1) return -1;
        rc = pg_launch<512, 32, 8, PG_RANK1_MASK, false>(a, w, c, M, N, bias, bo, bi, rowscale, rank1, rows_per_bag, part, &streams, walk_reverse, stream);
    } else if (K == 128 && epilogue == PG_RANK1_MASK) {
        if (!bi || !rowscale || !rank1) return -1;
        rc = pg_launch<128, 64, 8, PG_RANK1_MASK, false>(a, w, c, M, N, bias, bo, bi, rowscale, rank1, rows_per_bag, part, &streams, walk_reverse, stream);
    }
    if (rc == 0 && colsum_out)         // sum the per-workgroup rows [streams][N] into the (bias-gradient) output
        rc = murcl_colsum(part, colsum_out, streams, N, N, MURCL_DTYPE_F32, colsum_accumulate, stream);
    return rc;
}
