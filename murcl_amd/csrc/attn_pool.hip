// K2: ABMIL attention pooling over bags of encoded patch features (models/abmil.py:38-42).
//
//   s_n = wb . tanh(Wa H_n + ba) + bb ;  A = softmax_N(s) / sqrt(N) ;  M = A . H
//
// One pass over H.  Persistent workgroups (one per CU, 4 waves, one wave per SIMD) walk
// (bag, row-chunk) items; H row tiles stream HBM -> LDS by LDS-DMA into a 4-slot ring with three
// tiles (96 KiB) in flight per CU; each wave keeps its 32-column slice of Wa in registers as MFMA
// operands for the whole launch, so per tile the matrix cores see only LDS reads of H.  Scores are
// reduced across waves through LDS, the soft-max is kept online (running max / sum per chunk), and
// the weighted sum M is accumulated on the VALU from the same LDS tile.  Per-chunk partials
// (m, l, sum p.H) are merged per bag by abmil_pool_combine_kernel.
//
// LDS row image: a row of L elements is ROWB bytes = ROWB/16 chunks; chunk c of row r is stored at
// position c ^ (r & 15) (conflict-free ds_read_b128 for 16 rows x same k-chunk).  The swizzle is
// applied on the LDS-DMA source address and on every read.
#include "common.h"

#define K2_L 512
#define K2_D 128
#define K2_SLOT 32768
#define K2_NSLOT 4

template <typename T> struct K2 {
    static constexpr int ROWB = K2_L * (int)sizeof(T);       // bytes per row: 1024 / 2048
    static constexpr int TR = K2_SLOT / ROWB;                // rows per tile: 32 / 16
    static constexpr int NI = TR / 16;                       // 16-row MFMA tiles per tile: 2 / 1
    static constexpr int CPR = ROWB / 16;                    // chunks per row: 64 / 128
    static constexpr int GL = K2_SLOT / 4096;                // LDS-DMA instructions per wave per tile = 8
    static constexpr int NKK = K2_L * (int)sizeof(T) / 64;   // 16-byte k groups per quarter: 16 / 32
};

// LDS carve (bytes): ring 4*32 KiB | spart [4 waves][32 rows] f32 | sbuf [<=chunk rows] f32
#define K2_OFF_SPART (K2_NSLOT * K2_SLOT)
#define K2_OFF_SBUF (K2_OFF_SPART + 4 * 32 * 4)
#define K2_MAX_CHUNK 2048
#define K2_LDS_BYTES (K2_OFF_SBUF + K2_MAX_CHUNK * 4)

template <typename T> struct WFrag;
template <> struct WFrag<bf16_t> { typedef bf16x8 type; };
template <> struct WFrag<float> { typedef f32x4 type; };

template <typename T>
__device__ __forceinline__ f32x4 k2_mma(typename WFrag<T>::type a, typename WFrag<T>::type b, f32x4 c);
template <> __device__ __forceinline__ f32x4 k2_mma<bf16_t>(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 k2_mma<float>(f32x4 a, f32x4 b, f32x4 c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
    return c;
}

template <typename T, bool EXACT_TANH>
__global__ __launch_bounds__(256, 1) void abmil_pool_fwd_kernel(
    const T* __restrict__ H, const T* __restrict__ Wa, const float* __restrict__ ba, const float* __restrict__ wb,
    const float* __restrict__ bb_p, float* __restrict__ scores, float* __restrict__ part, int B, int N,
    int chunk_rows, int S) {
    typedef K2<T> C_;
    typedef typename WFrag<T>::type frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q4 = lane >> 4, r16 = lane & 15;
    const unsigned lds0 = lds_off(smem);
    float* spart = (float*)(smem + K2_OFF_SPART);
    float* sbuf = (float*)(smem + K2_OFF_SBUF);

    const int n_items = B * S;
    const int tiles_per_item = chunk_rows / C_::TR;
    const int my_items = (n_items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int my_tiles = my_items * tiles_per_item;
    if (my_tiles <= 0) return;

    // ---- LDS-DMA issue for tile `seq` of this workgroup.  Instruction j of wave w fills rows
    // (4j+w) for bf16 (one row per instruction) / half rows for f32.
    auto issue = [&](int seq) {
        const int item = blockIdx.x + (seq / tiles_per_item) * gridDim.x;
        const int bag = item / S, ch = item - bag * S;
        const int row0 = ch * chunk_rows + (seq % tiles_per_item) * C_::TR;
        const char* base = (const char*)(H + (size_t)bag * N * K2_L);
        const unsigned slot = lds0 + (seq % K2_NSLOT) * K2_SLOT;
#pragma unroll
        for (int j = 0; j < C_::GL; ++j) {
            const int ci = (j * 4 + wave) * 64 + lane;          // 16-byte chunk index within the tile
            const int row = ci / C_::CPR, pos = ci % C_::CPR;
            const int grow = min(row0 + row, N - 1);            // rows past N: clamped, masked below
            const char* src = base + (size_t)grow * C_::ROWB + ((pos ^ (row & 15)) << 4);
            glds16(src, slot + (j * 4 + wave) * 1024);
        }
    };

    const int pre = min(3, my_tiles);
    for (int s = 0; s < pre; ++s) issue(s);

    // ---- this wave's Wa slice as MFMA "a" operands: rows d = 32*wave + 16j + r16
    frag_t wa[2][C_::NKK];
    float ba_r[2][4], wb_r[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const T* wrow = Wa + (size_t)(32 * wave + 16 * j + r16) * K2_L;
#pragma unroll
        for (int kk = 0; kk < C_::NKK; ++kk) wa[j][kk] = *(const frag_t*)((const char*)wrow + (4 * kk + q4) * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ba_r[j][r] = ba[32 * wave + 16 * j + 4 * q4 + r];
            wb_r[j][r] = wb[32 * wave + 16 * j + 4 * q4 + r];
        }
    }
    const float bb = bb_p[0];
    // The loads above are ordinary (compiler-counted) loads issued AFTER the first LDS-DMA tiles; the
    // compiler's own waits for them are conservative w.r.t. the older LDS-DMA ops.  Drain them here
    // so that from now on the only VMEM ops in flight are the LDS-DMA tiles counted by hand.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (that also retired the first `pre` tiles; the steady-state counts below stay valid because
    //  waiting for more than necessary is always safe.)

    float m_run = -INFINITY, l_run = 0.f, macc0 = 0.f, macc1 = 0.f;
    const int col0 = 2 * tid;                 // this thread's two pooled columns
    const int chunk_of_col = (col0 * (int)sizeof(T)) >> 4, inchunk = (col0 * (int)sizeof(T)) & 15;

    for (int seq = 0; seq < my_tiles; ++seq) {
        // tile `seq` landed when at most (tiles issued after it) * GL LDS-DMA ops are outstanding
        const int ahead = min(2, my_tiles - 1 - seq);
        if (ahead == 2) { WAIT_VMCNT(16); } else if (ahead == 1) { WAIT_VMCNT(8); } else { WAIT_VMCNT(0); }
        LDS_BARRIER();                         // all waves' pieces landed; slot (seq+3)%4 is free
        if (seq + 3 < my_tiles) issue(seq + 3);

        const int tin = seq % tiles_per_item;
        const int item = blockIdx.x + (seq / tiles_per_item) * gridDim.x;
        const int bag = item / S, ch = item - bag * S;
        const int row0 = ch * chunk_rows + tin * C_::TR;
        const char* tile = smem + (seq % K2_NSLOT) * K2_SLOT;

        // ---- phase A: scores for the tile, this wave's 32 columns of D
        f32x4 acc[C_::NI][2];
#pragma unroll
        for (int i = 0; i < C_::NI; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < C_::NKK; ++kk) {
#pragma unroll
            for (int i = 0; i < C_::NI; ++i) {
                const int row = 16 * i + r16;
                const int c = 4 * kk + q4;
                frag_t h = *(const frag_t*)(tile + row * C_::ROWB + ((c ^ (row & 15)) << 4));
                acc[i][0] = k2_mma<T>(wa[0][kk], h, acc[i][0]);
                acc[i][1] = k2_mma<T>(wa[1][kk], h, acc[i][1]);
            }
        }
        // lane holds pre-activations for patch row 16i+r16, d = 32w+16j+4q+r
#pragma unroll
        for (int i = 0; i < C_::NI; ++i) {
            float ps = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = acc[i][j][r] + ba_r[j][r];
                    ps += wb_r[j][r] * (EXACT_TANH ? tanhf(x) : fast_tanh(x));
                }
            ps += __shfl_xor(ps, 16, 64);
            ps += __shfl_xor(ps, 32, 64);
            if (q4 == 0) spart[wave * 32 + 16 * i + r16] = ps;
        }
        LDS_BARRIER();

        // ---- phase B (every wave redundantly): tile soft-max statistics, lane r <-> row r
        float s = -INFINITY;
        if (lane < C_::TR) {
            s = spart[lane] + spart[32 + lane] + spart[64 + lane] + spart[96 + lane] + bb;
            if (row0 + lane >= N) s = -INFINITY;             // ragged tail rows carry no weight
            if (wave == 0) sbuf[tin * C_::TR + lane] = s;
        }
        const float tmax = wave_max(s);
        const float m_new = fmaxf(m_run, tmax);
        const float scale = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
        const float p = (s == -INFINITY) ? 0.f : __expf(s - m_new);
        l_run = l_run * scale + wave_sum(p);
        m_run = m_new;
        macc0 *= scale;
        macc1 *= scale;
        // ---- pooling: M[col] += sum_r p_r H[r][col]; p_r broadcast through an SGPR
#pragma unroll
        for (int r = 0; r < C_::TR; ++r) {
            const float pr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), r));
            const char* hp = tile + r * C_::ROWB + ((chunk_of_col ^ (r & 15)) << 4) + inchunk;
            if (sizeof(T) == 2) {
                const uint32_t u = *(const uint32_t*)hp;
                macc0 += pr * bf_lo(u);
                macc1 += pr * bf_hi(u);
            } else {
                const f32x2 v = *(const f32x2*)hp;
                macc0 += pr * v[0];
                macc1 += pr * v[1];
            }
        }

        if (tin == tiles_per_item - 1) {        // ---- end of item: publish partial + raw scores
            float* pp = part + (size_t)item * (K2_L + 2);
            if (tid == 0) { pp[0] = m_run; pp[1] = l_run; }
            pp[2 + col0] = macc0;
            pp[2 + col0 + 1] = macc1;
            __syncthreads();                    // sbuf complete (wave 0 wrote it)
            const int rbeg = ch * chunk_rows;
            for (int r = tid; r < chunk_rows && rbeg + r < N; r += 256) scores[(size_t)bag * N + rbeg + r] = sbuf[r];
            // the stores above are compiler-visible VMEM ops; retire them so the hand counts of
            // LDS-DMA ops stay exact for the next item.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            m_run = -INFINITY; l_run = 0.f; macc0 = 0.f; macc1 = 0.f;
        }
    }
}

// Per bag: merge chunk partials, emit A[n] = softmax(s)_n / sqrt(N), pooled M, and (m, l).
__global__ __launch_bounds__(256) void abmil_pool_combine_kernel(const float* __restrict__ scores,
                                                                 const float* __restrict__ part, float* __restrict__ A,
                                                                 float* __restrict__ Mout, float* __restrict__ ml,
                                                                 int N, int S, float inv_sqrt_n) {
    const int bag = blockIdx.x, tid = threadIdx.x;
    const float* pp = part + (size_t)bag * S * (K2_L + 2);
    float m = -INFINITY;
    for (int s = 0; s < S; ++s) m = fmaxf(m, pp[(size_t)s * (K2_L + 2)]);
    float l = 0.f, a0 = 0.f, a1 = 0.f;
    for (int s = 0; s < S; ++s) {
        const float* p = pp + (size_t)s * (K2_L + 2);
        const float w = (p[0] == -INFINITY) ? 0.f : expf(p[0] - m);
        l += p[1] * w;
        a0 += p[2 + 2 * tid] * w;
        a1 += p[2 + 2 * tid + 1] * w;
    }
    const float inv = inv_sqrt_n / l;
    Mout[(size_t)bag * K2_L + 2 * tid] = a0 * inv;
    Mout[(size_t)bag * K2_L + 2 * tid + 1] = a1 * inv;
    if (tid == 0) { ml[2 * bag] = m; ml[2 * bag + 1] = l; }
    for (int n = tid; n < N; n += 256) A[(size_t)bag * N + n] = expf(scores[(size_t)bag * N + n] - m) * inv;
}

static int pick_chunk(int B, int N, int tr, int n_cu) {
    // rows per item: multiple of the tile height, <= K2_MAX_CHUNK, small enough to give every CU work
    int chunk = ((N + tr - 1) / tr) * tr;
    if (chunk > K2_MAX_CHUNK) chunk = K2_MAX_CHUNK;
    while (chunk > tr && (long)B * ((N + chunk - 1) / chunk) < 2L * n_cu) {
        int c2 = ((chunk / 2 + tr - 1) / tr) * tr;
        if (c2 == chunk) break;
        chunk = c2;
    }
    return chunk;
}

extern "C" int murcl_abmil_pool_workspace(int B, int N, int dtype, int* chunk_rows, int* n_chunks) {
    const int tr = dtype == MURCL_DTYPE_BF16 ? 32 : 16;
    const int c = pick_chunk(B, N, tr, 256);
    *chunk_rows = c;
    *n_chunks = (N + c - 1) / c;
    return 0;
}

// C-ABI: see include/murcl_amd.h
extern "C" int murcl_abmil_pool_fwd(const void* H, const void* Wa, const float* ba, const float* wb, const float* bb,
                                    float* scores, float* A, float* M, float* ml, float* part_ws, int B, int N, int L,
                                    int D, int dtype, int exact_tanh, hipStream_t stream) {
    if (L != K2_L || D != K2_D) return -1;
    if (B <= 0 || N <= 0) return 0;
    int chunk, S;
    murcl_abmil_pool_workspace(B, N, dtype, &chunk, &S);
    const int items = B * S;
    const int grid = items < 256 ? items : 256;
#define K2_LAUNCH(T, EX)                                                                                        \
    {                                                                                                           \
        auto k = abmil_pool_fwd_kernel<T, EX>;                                                                  \
        static bool once = false;                                                                               \
        if (!once) {                                                                                            \
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, K2_LDS_BYTES);      \
            once = true;                                                                                        \
        }                                                                                                       \
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), K2_LDS_BYTES, stream, (const T*)H, (const T*)Wa, ba, wb,   \
                           bb, scores, part_ws, B, N, chunk, S);                                                \
    }
    if (dtype == MURCL_DTYPE_BF16) {
        if (exact_tanh) K2_LAUNCH(bf16_t, true) else K2_LAUNCH(bf16_t, false)
    } else if (dtype == MURCL_DTYPE_F32) {
        if (exact_tanh) K2_LAUNCH(float, true) else K2_LAUNCH(float, false)
    } else {
        return -1;
    }
#undef K2_LAUNCH
    int rc = MURCL_CHECK_LAUNCH();
    if (rc) return rc;
    hipLaunchKernelGGL(abmil_pool_combine_kernel, dim3(B), dim3(256), 0, stream, scores, part_ws, A, M, ml, N, S,
                       1.0f / sqrtf((float)N));
    return MURCL_CHECK_LAUNCH();
}
