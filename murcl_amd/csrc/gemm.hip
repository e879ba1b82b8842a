// Dense projection GEMMs of the MIL aggregators on CDNA4 matrix cores.
//
//   gemm_nt : C[M,N] = epi(A[M,K] . B[N,K]^T)      forward Linear (abmil.py:12-21, clam.py:69,
//                                                    dsmil.py:15,66-67) and dgrad (with W^T as B)
//   gemm_tn : C[N1,N2] += A[M,N1]^T . B[M,N2]       wgrad, reduction over the patch dimension
//
// Tiling (both): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave =
// 4x4 MFMA 16x16 tiles), K slabs of 128 bytes per row staged HBM->LDS by LDS-DMA
// (global_load_lds_dwordx4) into a double buffer.  bf16 inputs use v_mfma_f32_16x16x32_bf16,
// f32 inputs the exact-f32 v_mfma_f32_16x16x4_f32 (parity path).  LDS images are XOR-swizzled on
// the 16-byte chunk index; the swizzle is applied to the per-lane *source* address (the LDS-DMA
// destination is lane-linear) and again on the fragment reads.
#include "common.h"

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_RELU = 2, EPI_MASK = 3, EPI_RANK1_MASK = 4 };

struct GemmEpi {
    const float* bias;      // [N]
    const void* mask;       // [M,ldmask] same dtype as A: output is zeroed where mask <= 0 (ReLU')
    int ldmask;
    const float* rowscale;  // [M]           a[m]
    const float* rank1;     // [M/rows_per_bag, N]   out += a[m] * rank1[bag(m)][n]  (before mask)
    int rows_per_bag;
    float* colsum_ws;       // [ceil(M/128), N] per-tile column sums of the output, or nullptr
    int accumulate;         // C += result (f32 output only)
};

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef f32x4 type; };

template <typename T>
__device__ __forceinline__ f32x4 mma16(typename Frag<T>::type a, typename Frag<T>::type b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mma16<bf16_t>(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mma16<float>(f32x4 a, f32x4 b, f32x4 c) {
    // lane quarter q holds k = 4q..4q+3 of this 16-wide k group; MFMA e pairs element e of both
    // operands, so the k order is permuted identically on both sides (sum is order-independent
    // up to fp32 rounding).
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
    return c;
}

// f32 operands on the bf16 matrix pipe (dtype code MURCL_DTYPE_F32X3): x = hi + mid + lo with three bf16 terms (8 + 8 + 8
// significant bits, residuals exact in f32), the six products of order <= 2 (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid) on
// v_mfma_f32_16x16x32_bf16 with f32 accumulation: each bf16 x bf16 product is exact, the dropped terms are <= 2^-26 of the
// product - f32-level accuracy at 6/16 of the exact-f32 MFMA time (v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate).
// Two f32x4 fragments (the lane's k values of k-groups 0 and 1 of a 32-deep slab) make one 8-deep bf16 fragment per term;
// both operands are built the same way, so the k order is permuted identically on both sides.
struct Split3 { bf16x8 h, m, l; };
__device__ __forceinline__ Split3 split3(f32x4 x0, f32x4 x1) {
    u32x4 h, m, l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = p < 2 ? x0[2 * p] : x1[2 * p - 4], b = p < 2 ? x0[2 * p + 1] : x1[2 * p - 3];
        const uint32_t hp = pack_bf2(a, b);
        const float ra = a - bf_lo(hp), rb = b - bf_hi(hp);
        const uint32_t mp = pack_bf2(ra, rb);
        const float sa = ra - bf_lo(mp), sb = rb - bf_hi(mp);
        h[p] = hp;
        m[p] = mp;
        l[p] = pack_bf2(sa, sb);
    }
    return Split3{__builtin_bit_cast(bf16x8, h), __builtin_bit_cast(bf16x8, m), __builtin_bit_cast(bf16x8, l)};
}
// c += a . b over the slab's 32 k values, a and b as split operands (MFMA rows <- a)
__device__ __forceinline__ f32x4 mma_x3(const Split3& a, const Split3& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b.h, c, 0, 0, 0);       // smallest terms first
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, c, 0, 0, 0);
    return c;
}

// ------------------------------------------------------------------------------------- NT
template <typename T, typename OutT, int EPI, bool X3 = false>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const T* __restrict__ A, const T* __restrict__ B,
                                                      OutT* __restrict__ C, int M, int N, int K, int lda,
                                                      int ldb, int ldc, GemmEpi e) {
    typedef typename Frag<T>::type frag_t;
    constexpr int BKE = 128 / (int)sizeof(T);          // K elements per slab
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (A 16 KiB | B 16 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt_n = (N + 127) >> 7, nt_m = (M + 127) >> 7;
    const int swz = xcd_remap(blockIdx.x, nt_n * nt_m);
    const int tm = swz / nt_n, tn = swz - tm * nt_n;    // the nt_n column tiles of a row panel are adjacent
    const int m0 = tm << 7, n0 = tn << 7;
    const unsigned lds0 = lds_off(smem);

    // per-thread staging sources: chunk ci = t*256+tid -> row ci>>3, LDS pos ci&7, global chunk pos^(row&7)
    const char* asrc[4];
    const char* bsrc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int ci = t * 256 + tid, row = ci >> 3, c = (ci & 7) ^ (row & 7);
        int ar = min(m0 + row, M - 1), br = min(n0 + row, N - 1);
        asrc[t] = (const char*)(A + (size_t)ar * lda) + c * 16;
        bsrc[t] = (const char*)(B + (size_t)br * ldb) + c * 16;
    }
    auto stage = [&](int buf, int ks) {
        const unsigned la = lds0 + buf * 32768, lb = la + 16384;
        const size_t koff = (size_t)ks * 128;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            glds16(asrc[t] + koff, la + (t * 256 + wave * 64) * 16);
            glds16(bsrc[t] + koff, lb + (t * 256 + wave * 64) * 16);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wr = wave >> 1, wc = wave & 1, q4 = lane >> 4, r16 = lane & 15;

    auto compute = [&](int buf) {
        const char* la = smem + buf * 32768;
        const char* lb = la + 16384;
        if constexpr (X3) {
            static_assert(!X3 || sizeof(T) == 4, "the 3-term split is for f32 operands");
            Split3 as[4], bs[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wr * 64 + i * 16 + r16;
                as[i] = split3(*(const f32x4*)(la + row * 128 + (((q4) ^ (row & 7)) << 4)),
                               *(const f32x4*)(la + row * 128 + (((4 + q4) ^ (row & 7)) << 4)));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wc * 64 + j * 16 + r16;
                bs[j] = split3(*(const f32x4*)(lb + row * 128 + (((q4) ^ (row & 7)) << 4)),
                               *(const f32x4*)(lb + row * 128 + (((4 + q4) ^ (row & 7)) << 4)));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mma_x3(bs[j], as[i], acc[i][j]);
            return;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            frag_t af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = wr * 64 + i * 16 + r16;
                af[i] = *(const frag_t*)(la + row * 128 + (((4 * kk + q4) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int row = wc * 64 + j * 16 + r16;
                bfr[j] = *(const frag_t*)(lb + row * 128 + (((4 * kk + q4) ^ (row & 7)) << 4));
            }
            // swapped roles: MFMA rows <- B (n), cols <- A (m): a lane ends with 4 consecutive n
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mma16<T>(bfr[j], af[i], acc[i][j]);
        }
    };

    const int nk = K / BKE;
    stage(0, 0);
    WAIT_VMCNT(0);
    LDS_BARRIER();
    for (int ks = 0; ks < nk - 1; ++ks) {
        stage((ks + 1) & 1, ks + 1);      // that buffer's last reads retired before the previous barrier
        compute(ks & 1);
        WAIT_VMCNT(0);
        LDS_BARRIER();
    }
    compute((nk - 1) & 1);

    // ---- epilogue: lane holds C[m = m0+wr*64+i*16+r16][n = n0+wc*64+j*16+4*q4 + 0..3]
    float* cs = (float*)smem;             // [128] column sums (LDS reuse after the last compute)
    if (e.colsum_ws) {
        __syncthreads();
        if (tid < 128) cs[tid] = 0.f;
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + 4 * q4;
        f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bias4[r] = (n + r < N) ? e.bias[n + r] : 0.f;
        }
        f32x4 csum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wr * 64 + i * 16 + r16;
            f32x4 v = acc[i][j];
            const bool mok = m < M;
            const bool full = mok && (n + 3 < N);
            if (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) v += bias4;
            if (EPI == EPI_BIAS_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (EPI == EPI_RANK1_MASK && mok) {
                const float a = e.rowscale[m];
                const float* rk = e.rank1 + (size_t)(m / e.rows_per_bag) * N + n;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < N) v[r] += a * rk[r];
            }
            if ((EPI == EPI_MASK || EPI == EPI_RANK1_MASK) && mok) {
                const T* mp = (const T*)e.mask + (size_t)m * e.ldmask + n;
                if (full) {
                    f32x4 h = load4<T>(mp);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = h[r] > 0.f ? v[r] : 0.f;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < N) v[r] = to_f<T>(mp[r]) > 0.f ? v[r] : 0.f;
                }
            }
            if (mok) {
                OutT* cp = C + (size_t)m * ldc + n;
                if (full) {
                    if (e.accumulate) {
                        f32x4 old = load4<OutT>(cp);
                        v += old;
                    }
                    store4<OutT>(cp, v);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < N) cp[r] = from_f<OutT>(v[r] + (e.accumulate ? to_f<OutT>(cp[r]) : 0.f));
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) csum[r] += (n + r < N) ? v[r] : 0.f;
            }
        }
        if (e.colsum_ws) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = csum[r];
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                s += __shfl_xor(s, 4, 64);
                s += __shfl_xor(s, 8, 64);
                if (r16 == 0) atomicAdd(&cs[wc * 64 + j * 16 + 4 * q4 + r], s);
            }
        }
    }
    if (e.colsum_ws) {
        __syncthreads();
        if (tid < 128 && n0 + tid < N) e.colsum_ws[(size_t)tm * N + n0 + tid] = cs[tid];
    }
}

template <typename T, typename OutT, bool X3 = false>
static int launch_nt_epi(const T* A, const T* B, OutT* C, int M, int N, int K, int lda, int ldb, int ldc, int epi,
                         const GemmEpi& e, hipStream_t s) {
    const int grid = ((M + 127) / 128) * ((N + 127) / 128);
#define NT_CASE(E)                                                                                             \
    case E: {                                                                                                  \
        auto k = gemm_nt_kernel<T, OutT, E, X3>;                                                               \
        static MurclOncePerDevice once;                                                                                    \
        if (once.first()) {                                                                                           \
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);            \
                                                                                                                  \
        }                                                                                                      \
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 65536, s, A, B, C, M, N, K, lda, ldb, ldc, e);           \
        break;                                                                                                 \
    }
    switch (epi) {
        NT_CASE(EPI_NONE)
        NT_CASE(EPI_BIAS)
        NT_CASE(EPI_BIAS_RELU)
        NT_CASE(EPI_MASK)
        NT_CASE(EPI_RANK1_MASK)
        default: return -2;
    }
#undef NT_CASE
    return MURCL_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------- bag-level NT (f32, M <= 1024)
// Bag-level layers (decoder, GRU projections, heads): M = bags, so a 128 x 128-tile grid has only N/128 workgroups.  The operands
// go through LDS: fragments read straight from global memory (the two register-direct forms of rounds 1-2, removed in round 5)
// are one 16-byte load per lane = 16 ROWS x 64 bytes - sixteen cache lines per instruction, half of each unused - and the
// texture-address path, not the matrix pipe or the memory system, set their time (a 64 x 16 x 512 workgroup took ~9 us however hot
// its operands; LOG.md round 2).  Here a workgroup owns a 32 x 32 output tile; its 32
// rows of A and 32 rows of B arrive by LDS-DMA as whole 1 KiB row pieces (256 k, eight full lines per instruction, shared by
// the four waves) into two chunk buffers with rows at a stride of 1024 + 16 bytes (conflict-free 16-byte fragment reads,
// the K2 layout); K <= 512 is in flight at once, longer reductions ping-pong.  Single writer per element: no memset, no
// atomics, bias / ReLU / accumulate in the epilogue, a result that does not depend on the order of float atomics.
constexpr int SL_K = 256, SL_ROW = 1024 + 16, SL_T = 32, SL_BUF = 2 * SL_T * SL_ROW;
#define SL_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
__global__ __launch_bounds__(256) void gemm_nt_lds32_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                float* __restrict__ C, int M, int N, int K, int lda,
                                                                int ldb, int ldc, const float* __restrict__ bias,
                                                                int relu, int accumulate, int k_per_split,
                                                                const float* __restrict__ mask, int ldmask, int one_slot) {
    extern __shared__ __attribute__((aligned(16))) char sl_smem[];
    // gridDim.z > 1: K split over workgroups (long reductions with few output tiles); the partial tiles are added
    // atomically into a zeroed C (bias from split 0; a ReLU, if any, is the launcher's separate pass)
    if (gridDim.z > 1) {
        const int kb = blockIdx.z * k_per_split;
        A += kb;
        B += kb;
        K = min(K - kb, k_per_split);
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q4 = lane >> 4, r16 = lane & 15;
    const int n0 = blockIdx.x * SL_T, m0 = blockIdx.y * SL_T;
    const unsigned lds0 = lds_off(sl_smem);
    const int nch = (K + SL_K - 1) / SL_K;
    // chunk image: rows 0..31 = A rows m0.., rows 32..63 = B rows n0..; wave w copies image rows 16w..16w+15 (waves 0,1: A;
    // waves 2,3: B), one LDS-DMA instruction per row.  Rows past M / N are clamped (their results are never stored); in a
    // partial last chunk the lanes past K re-read the row's last 16 bytes (never consumed: the k loop stops at K).
    auto issue = [&](int c) {
        const int k0 = c * SL_K;
        const unsigned voff = (unsigned)min(lane * 16, (K - k0) * 4 - 16);
        const unsigned dst = lds0 + (one_slot ? 0 : (c & 1) * SL_BUF) + wave * 16 * SL_ROW;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int r = (wave & 1) * 16 + j;
            const float* src = wave < 2 ? A + (size_t)min(m0 + r, M - 1) * lda + k0 : B + (size_t)min(n0 + r, N - 1) * ldb + k0;
            glds16_u(src, voff, dst + j * SL_ROW);
        }
    };
    // one_slot: a single chunk buffer (half the LDS: two workgroups per CU, whose round trips overlap each other's MFMAs) - for
    // grids of several rounds of workgroups; otherwise two chunks in flight per workgroup
    const int ahead = one_slot ? 1 : 2;
    issue(0);
    if (nch > 1 && !one_slot) issue(1);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;          // two chains: the f32 MFMA's dependent latency exceeds its issue time
    const int wm = wave & 1, wn = wave >> 1;
    for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch && !one_slot) { SL_WAIT(16); } else { SL_WAIT(0); }
        LDS_BARRIER();                                       // every wave's rows of chunk c have landed
        const unsigned sb = one_slot ? 0u : (unsigned)(c & 1) * SL_BUF;
        const char* ab = sl_smem + sb + (16 * wm + r16) * SL_ROW + 16 * q4;
        const char* bb = sl_smem + sb + (SL_T + 16 * wn + r16) * SL_ROW + 16 * q4;
        const int ku = min(SL_K, K - c * SL_K) / 16;
        if (ku == SL_K / 16) {
#pragma unroll
            for (int u = 0; u < SL_K / 16; u += 2) {
                acc0 = mma16<float>(*(const f32x4*)(ab + 64 * u), *(const f32x4*)(bb + 64 * u), acc0);
                acc1 = mma16<float>(*(const f32x4*)(ab + 64 * (u + 1)), *(const f32x4*)(bb + 64 * (u + 1)), acc1);
            }
        } else {
            for (int u = 0; u < ku; ++u) acc0 = mma16<float>(*(const f32x4*)(ab + 64 * u), *(const f32x4*)(bb + 64 * u), acc0);
        }
        if (c + ahead < nch) {
            LDS_BARRIER();                                   // that buffer has been read by every wave
            issue(c + ahead);
        }
    }
    // lane holds C[m0 + 16wm + 4q4 + r][n0 + 16wn + r16]: the 16 lanes of a quarter write 64 contiguous bytes of a row
    const int n = n0 + 16 * wn + r16;
    if (n < N) {
        const float bv = (bias && blockIdx.z == 0) ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * wm + 4 * q4 + r;
            if (m < M) {
                float v = acc0[r] + acc1[r] + bv;
                float* cp = C + (size_t)m * ldc + n;
                if (gridDim.z > 1) {
                    atomicAdd(cp, v);                        // 16 lanes of a quarter: 64 contiguous bytes per request
                } else {
                    if (mask) v = mask[(size_t)m * ldmask + n] > 0.f ? v : 0.f;      // ReLU' of the layer's saved output
                    if (relu) v = fmaxf(v, 0.f);
                    *cp = accumulate ? *cp + v : v;                                  // (accumulation adds the finished epilogue, as gemm_nt_kernel)
                }
            }
        }
    }
}
__global__ void relu_inplace_kernel(float* x, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = fmaxf(x[i], 0.f);
}

// K splits of the LDS form for this shape (1: single writer per element - bias / ReLU / mask / accumulate in the epilogue)
static int skinny_lds_splits(int M, int N, int K) {
    const long sl_tiles = (long)((M + SL_T - 1) / SL_T) * ((N + SL_T - 1) / SL_T);
    const int nch = (K + SL_K - 1) / SL_K;
    int sp = 1;
    if (nch >= 6 && sl_tiles <= 96) {
        sp = (int)((384 + sl_tiles - 1) / sl_tiles);
        if (sp > nch / 2) sp = nch / 2;
    }
    const int kps = ((nch + sp - 1) / sp) * SL_K;
    return (K + kps - 1) / kps;
}
// gru.hip: 16 x 16 tiles, K split over the waves of a workgroup (single writer, every epilogue)
bool murcl_nt_t16_ok(int M, int N, int K);
int murcl_nt_t16_launch(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, const float* bias,
                        int relu, int accumulate, const float* mask, int ldmask, hipStream_t stream);
constexpr int NT_T16_FEW_TILES = 64;       // products of at most this many 32 x 32 tiles (= 256 of the 16 x 16 kind: one round) ...
constexpr int NT_T16_MIN_CHUNKS = 2;       // ... with at least this many 256-k chunks take the 16 x 16 form ([128 x 512 x 512]: 7.2 -> 4.8 us; at 128 tiles it loses: profiles/r04_w_*)
static int launch_skinny(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                         int epi, const float* bias, int accumulate, hipStream_t s, const float* mask = nullptr, int ldmask = 0) {
    // few outputs, long reduction - the shapes the LDS form below would split over K (zero-fill + atomics + a ReLU launch): 16 x 16
    // tiles with the K range split over the waves of one workgroup instead, everything in one launch
    // (also a handful of tiles with three or more chunks each: [128 x 128 x 1024] is 16 workgroups walking 4 chunks, or 64 with all four in flight)
    const long sl_tiles_ = (long)((M + SL_T - 1) / SL_T) * ((N + SL_T - 1) / SL_T);
    const bool few_long = sl_tiles_ <= NT_T16_FEW_TILES && (K + SL_K - 1) / SL_K >= NT_T16_MIN_CHUNKS;
    if ((skinny_lds_splits(M, N, K) > 1 || few_long) && murcl_nt_t16_ok(M, N, K))
        return murcl_nt_t16_launch(A, B, C, M, N, K, lda, ldb, ldc, (epi == EPI_BIAS || epi == EPI_BIAS_RELU) ? bias : nullptr,
                                   (int)(epi == EPI_BIAS_RELU), accumulate, epi == EPI_MASK ? mask : nullptr, ldmask, s);
    // LDS form.  Long reductions with few output tiles that the 16 x 16 form above does not take (N not a multiple of 16) are split
    // over K (per 256-k chunk a workgroup needs ~2 us - one DMA round trip is not covered by one chunk of MFMAs - so
    // [128 x 512 x 3072] on 64 workgroups x 12 chunks took 26 us) so that every workgroup has its whole share (two chunks) in
    // flight at once; those partial tiles meet in a zeroed C by atomics.
    const int sp = skinny_lds_splits(M, N, K);
    const int kps = ((((K + SL_K - 1) / SL_K) + sp - 1) / sp) * SL_K;
    if (sp > 1 && !accumulate) {
        hipError_t e = hipMemset2DAsync(C, (size_t)ldc * 4, 0, (size_t)N * 4, M, s);
        if (e != hipSuccess) return (int)e;
    }
    const bool relu = epi == EPI_BIAS_RELU;
    static MurclOncePerDevice once;
    if (once.first())
        hipFuncSetAttribute((const void*)gemm_nt_lds32_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SL_BUF);
    // more than ~1.5 rounds of workgroups at one per CU: half the LDS, two per CU (profiles/r04_r_nt_oneslot*.txt)
    const long wgs = (long)((N + SL_T - 1) / SL_T) * ((M + SL_T - 1) / SL_T) * sp;
    const int one_slot = (int)(wgs > 384 && K > SL_K);
    hipLaunchKernelGGL(gemm_nt_lds32_f32_kernel, dim3((N + SL_T - 1) / SL_T, (M + SL_T - 1) / SL_T, sp), dim3(256), one_slot ? SL_BUF : 2 * SL_BUF, s, A,
                       B, C, M, N, K, lda, ldb, ldc, (epi == EPI_BIAS || epi == EPI_BIAS_RELU) ? bias : nullptr,
                       (int)(relu && sp == 1), (int)(accumulate && sp == 1), kps, epi == EPI_MASK ? mask : nullptr, ldmask, one_slot);
    int rc = MURCL_CHECK_LAUNCH();
    if (rc) return rc;
    if (relu && sp > 1) {
        if (ldc != N) return -1;
        const long n = (long)M * N;
        hipLaunchKernelGGL(relu_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, C, n);
        rc = MURCL_CHECK_LAUNCH();
    }
    return rc;
}

// C-ABI: see include/murcl_amd.h
extern "C" int murcl_gemm_nt(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                             int dtype_in, int dtype_out, int epilogue, const float* bias, const void* mask,
                             int ldmask, const float* rowscale, const float* rank1, int rows_per_bag,
                             float* colsum_ws, int accumulate, hipStream_t stream) {
    if (M <= 0 || N <= 0) return 0;
    const bool x3 = dtype_in == MURCL_DTYPE_F32X3;       // f32 storage, 3-term bf16 split on the bf16 matrix pipe
    if (x3) dtype_in = MURCL_DTYPE_F32;
    const int bke = dtype_in == MURCL_DTYPE_BF16 ? 64 : 32;
    if (K <= 0 || K % bke) return -1;                    // K must be a whole number of 128-byte slabs
    if ((lda * (dtype_in == MURCL_DTYPE_BF16 ? 2 : 4)) % 16 || (ldb * (dtype_in == MURCL_DTYPE_BF16 ? 2 : 4)) % 16)
        return -1;
    if (accumulate && dtype_out != MURCL_DTYPE_F32) return -1;
    // bag-level / rollout-level f32 layers (a few hundred rows): 32-column slabs x K splits x 128-row chunks fill the chip;
    // the 128 x 128 tile kernel would put e.g. [320 x 2048] x [512 x 2048]^T on 12 workgroups (86 us against ~15)
    if (dtype_in == MURCL_DTYPE_F32 && dtype_out == MURCL_DTYPE_F32 && M <= 1024 && K % 16 == 0 && !colsum_ws &&
        (epilogue == EPI_NONE || epilogue == EPI_BIAS || (epilogue == EPI_BIAS_RELU && !accumulate && ldc == N) ||
         (epilogue == EPI_MASK && mask &&
          (skinny_lds_splits(M, N, K) == 1 || murcl_nt_t16_ok(M, N, K)))))
        return launch_skinny((const float*)A, (const float*)B, (float*)C, M, N, K, lda, ldb, ldc, epilogue, bias,
                             accumulate, stream, (const float*)mask, ldmask);
    GemmEpi e{bias, mask, ldmask, rowscale, rank1, rows_per_bag > 0 ? rows_per_bag : 1, colsum_ws, accumulate};
    if (dtype_in == MURCL_DTYPE_BF16 && dtype_out == MURCL_DTYPE_BF16)
        return launch_nt_epi<bf16_t, bf16_t>((const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, M, N, K, lda, ldb, ldc,
                                             epilogue, e, stream);
    if (dtype_in == MURCL_DTYPE_BF16 && dtype_out == MURCL_DTYPE_F32)
        return launch_nt_epi<bf16_t, float>((const bf16_t*)A, (const bf16_t*)B, (float*)C, M, N, K, lda, ldb, ldc,
                                            epilogue, e, stream);
    if (dtype_in == MURCL_DTYPE_F32 && dtype_out == MURCL_DTYPE_F32 && x3)
        return launch_nt_epi<float, float, true>((const float*)A, (const float*)B, (float*)C, M, N, K, lda, ldb, ldc,
                                                 epilogue, e, stream);
    if (dtype_in == MURCL_DTYPE_F32 && dtype_out == MURCL_DTYPE_F32)
        return launch_nt_epi<float, float>((const float*)A, (const float*)B, (float*)C, M, N, K, lda, ldb, ldc,
                                           epilogue, e, stream);
    return -1;
}

// ------------------------------------------------------------------------------------- TN (wgrad)
// Slab = 16 KiB per operand: bf16 64 rows(m) x 128 cols, f32 32 rows x 128 cols.
template <typename T> struct TnTraits;
template <> struct TnTraits<bf16_t> {
    static constexpr int ROWS = 64, ROWB = 256;
    __device__ static __forceinline__ int swz(int row) { return ((row & 3) << 1) | (((row >> 3) & 1) << 3); }
};
template <> struct TnTraits<float> {
    static constexpr int ROWS = 32, ROWB = 512;
    __device__ static __forceinline__ int swz(int row) { return (row & 1) << 2; }
};

// fragment for output-tile index t (16 columns of the slab), k-group kk:
//  bf16: two transposed 4x16 reads -> k = 32kk + 8q + 0..7 of column 16t + (lane&15)
__device__ __forceinline__ bf16x8 tn_frag(const char* tile, int t, int kk, int lane, bf16_t) {
    const int g = lane >> 4, u = lane & 15, rq = u >> 2, p = u & 3;
    const int row = 32 * kk + 8 * g + rq;
    const int chunk = 2 * t + (p >> 1);
    const char* a0 = tile + row * 256 + ((chunk ^ TnTraits<bf16_t>::swz(row)) << 4) + ((p & 1) << 3);
    const int row1 = row + 4;
    const char* a1 = tile + row1 * 256 + ((chunk ^ TnTraits<bf16_t>::swz(row1)) << 4) + ((p & 1) << 3);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
//  f32: four scalar reads -> k = 16kk + 4e + q (e = 0..3) of column 16t + (lane&15)
__device__ __forceinline__ f32x4 tn_frag(const char* tile, int t, int kk, int lane, float) {
    const int q = lane >> 4, col = 16 * t + (lane & 15);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = 16 * kk + 4 * e + q;
        v[e] = *(const float*)(tile + row * 512 + ((((col >> 2) ^ TnTraits<float>::swz(row)) << 4) | ((col & 3) << 2)));
    }
    return v;
}

template <typename T, int NSLOT, bool X3 = false>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const T* __restrict__ A, const T* __restrict__ B,
                                                         float* __restrict__ C, int M, int N1, int N2, int lda, int ldb,
                                                         int ldc, int m_per_split, int nsplit,
                                                         float* __restrict__ colsum_out) {
    typedef typename Frag<T>::type frag_t;
    typedef TnTraits<T> TT;
    // NSLOT = 2: double buffer per workgroup; TWO workgroups per CU desynchronise and cover each other's barrier /
    // DMA-issue stalls (218 -> 184 us).  NSLOT = 4 (small grids, <= 1 workgroup per CU anyway): three slabs in flight,
    // so a short reduction (bag-level layers: M = 128 rows = 2-4 slabs) costs one memory round trip instead of one per slab.
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (A 16 KiB | B 16 KiB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware work map (speed only): workgroups b and b+8 share an XCD, so all T1*T2 output tiles of one
    // M-split - which read the same A/B slabs - are placed on one XCD and share them through its L2.
    const int T1 = (N1 + 127) >> 7, T2 = (N2 + 127) >> 7, tiles = T1 * T2;
    int tile, sp;
    if ((nsplit & 7) == 0) {
        const int b = blockIdx.x, xcd = b & 7, w = b >> 3;
        sp = xcd + 8 * (w / tiles);
        tile = w % tiles;
    } else {
        tile = blockIdx.x % tiles;
        sp = blockIdx.x / tiles;
    }
    const int t1 = tile % T1, t2 = tile / T1;
    const int n10 = t1 << 7, n20 = t2 << 7;
    const int mbeg = sp * m_per_split, mend = min(M, mbeg + m_per_split);
    if (mbeg >= mend) return;
    // The reduction runs from the last slab to the first: the producer of A (the dgrad / attention-backward kernel that
    // has just finished) wrote the high rows last, so they are still in the Infinity Cache; walking up would evict
    // them with the misses on the low rows before reaching them.
    const int nslab_total = (mend - mbeg + TT::ROWS - 1) / TT::ROWS;
    const unsigned lds0 = lds_off(smem);
    constexpr int CPR = TT::ROWB / 16;            // 16-byte chunks per slab row (16 or 32)
    constexpr int EPC = 16 / (int)sizeof(T);      // elements per chunk

    // chunk ci = t*256+tid -> row ci/CPR, LDS pos ci%CPR, source chunk pos^swz(row)
    int srow[4], acol[4], bcol[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int ci = t * 256 + tid, row = ci / CPR, c = (ci % CPR) ^ TT::swz(row);
        srow[t] = row;
        acol[t] = min(n10 + c * EPC, N1 - EPC);   // clamp: columns past N1/N2 are never stored
        bcol[t] = min(n20 + c * EPC, N2 - EPC);
    }
    auto stage = [&](int s) {
        const int mrow0 = mbeg + (nslab_total - 1 - s) * TT::ROWS;       // slabs are walked from the high rows down
        const unsigned la = lds0 + (s & (NSLOT - 1)) * 32768, lb = la + 16384;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // rows past mend are clamped to a valid row; the ragged tail is zeroed in LDS before use
            int m = min(mrow0 + srow[t], M - 1);
            glds16(A + (size_t)m * lda + acol[t], la + (t * 256 + wave * 64) * 16);
            glds16(B + (size_t)m * ldb + bcol[t], lb + (t * 256 + wave * 64) * 16);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wr = wave >> 1, wc = wave & 1;
    // bias gradient for free: the column sums of A are A^T . 1 - the workgroups of the first N2 tile run one more MFMA
    // per A fragment against a fragment of ones (every column of that accumulator block holds the sums)
    const bool do_cs = colsum_out != nullptr && t2 == 0 && wc == 0;
    f32x4 accs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) accs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    frag_t ones;
    if constexpr (sizeof(T) == 2) {
        ones = bf16x8{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    } else {
        ones = f32x4{1.f, 1.f, 1.f, 1.f};
    }

    // bf16: the transposed-read offsets inside a slab are lane constants (the swizzle depends only on
    // row&3 and (row>>3)&1, both fixed per lane); k-group kk adds 32 rows = 8 KiB.
    unsigned aoff[4][2], boff[4][2];
    if (sizeof(T) == 2) {
        const int g = lane >> 4, u = lane & 15, rq = u >> 2, p4 = u & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = 8 * g + rq + 4 * h;
                const int ca = 2 * (wr * 4 + i) + (p4 >> 1), cb = 2 * (wc * 4 + i) + (p4 >> 1);
                aoff[i][h] = row * 256 + ((ca ^ TnTraits<bf16_t>::swz(row)) << 4) + ((p4 & 1) << 3);
                boff[i][h] = 16384 + row * 256 + ((cb ^ TnTraits<bf16_t>::swz(row)) << 4) + ((p4 & 1) << 3);
            }
    }
    auto tr_frag = [&](const char* slab, unsigned o0, unsigned o1, int kk) -> bf16x8 {
        typedef __attribute__((address_space(3))) s16x4* lp;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(slab + o0 + kk * 8192));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(slab + o1 + kk * 8192));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    auto compute = [&](int s) {
        const char* la = smem + (s & (NSLOT - 1)) * 32768;
        const char* lb = la + 16384;
        if constexpr (X3) {
            static_assert(!X3 || sizeof(T) == 4, "the 3-term split is for f32 operands");
            Split3 as[4], bs[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) as[i] = split3(tn_frag(la, wr * 4 + i, 0, lane, float()), tn_frag(la, wr * 4 + i, 1, lane, float()));
#pragma unroll
            for (int j = 0; j < 4; ++j) bs[j] = split3(tn_frag(lb, wc * 4 + j, 0, lane, float()), tn_frag(lb, wc * 4 + j, 1, lane, float()));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mma_x3(as[i], bs[j], acc[i][j]);
            if (do_cs) {                       // column sums of A: A^T . 1, the ones fragment is exact in bf16
                const bf16x8 one8 = bf16x8{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as[i].l, one8, accs[i], 0, 0, 0);
                    accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as[i].m, one8, accs[i], 0, 0, 0);
                    accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as[i].h, one8, accs[i], 0, 0, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            frag_t af[4], bfr[4];
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = tr_frag(la, aoff[i][0], aoff[i][1], kk);
#pragma unroll
                for (int j = 0; j < 4; ++j) bfr[j] = tr_frag(la, boff[j][0], boff[j][1], kk);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = tn_frag(la, wr * 4 + i, kk, lane, T());
#pragma unroll
                for (int j = 0; j < 4; ++j) bfr[j] = tn_frag(lb, wc * 4 + j, kk, lane, T());
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mma16<T>(af[i], bfr[j], acc[i][j]);
            if (do_cs) {                       // wave-uniform
#pragma unroll
                for (int i = 0; i < 4; ++i) accs[i] = mma16<T>(af[i], ones, accs[i]);
            }
        }
    };

    const int nslab = (mend - mbeg + TT::ROWS - 1) / TT::ROWS;
    constexpr int D = NSLOT - 1;                   // slabs in flight (8 LDS-DMA ops per wave each)
    for (int s = 0; s < D && s < nslab; ++s) stage(s);
    for (int s = 0; s < nslab; ++s) {
        const int ahead = min(D - 1, nslab - 1 - s);   // younger slabs that may stay in flight while slab s is awaited
        if (ahead >= 2) { WAIT_VMCNT(16); } else if (ahead == 1) { WAIT_VMCNT(8); } else { WAIT_VMCNT(0); }
        LDS_BARRIER();                             // slab s visible to all waves; slot of slab s-1 is free
        if (s + D < nslab) stage(s + D);
        const int rows_here = min(TT::ROWS, mend - (mbeg + (nslab - 1 - s) * TT::ROWS));
        if (rows_here < TT::ROWS) {               // ragged tail: zero the invalid rows of both images
            char* la = smem + (s & (NSLOT - 1)) * 32768;
            for (int idx = tid; idx < (TT::ROWS - rows_here) * (TT::ROWB / 16); idx += 256) {
                int row = rows_here + idx / (TT::ROWB / 16), c = idx % (TT::ROWB / 16);
                *(u32x4*)(la + row * TT::ROWB + c * 16) = u32x4{0, 0, 0, 0};
                *(u32x4*)(la + 16384 + row * TT::ROWB + c * 16) = u32x4{0, 0, 0, 0};
            }
            LDS_BARRIER();
        }
        compute(s);
    }
    // lane holds C[n1 = n10+wr*64+i*16+4q+r][n2 = n20+wc*64+j*16+(lane&15)]
    const int q4 = lane >> 4, r16 = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n2 = n20 + wc * 64 + j * 16 + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n1 = n10 + wr * 64 + i * 16 + 4 * q4 + r;
                if (n1 < N1 && n2 < N2) atomicAdd(C + (size_t)n1 * ldc + n2, acc[i][j][r]);
            }
        }
    if (do_cs && r16 == 0) {                   // one adder per column and M-split (<= 64 per address)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n1 = n10 + wr * 64 + i * 16 + 4 * q4 + r;
                if (n1 < N1) atomicAdd(colsum_out + n1, accs[i][r]);
            }
    }
}

// ------------------------------------------------------------------------------------- TN, wide bf16 variant
// Two measured limits shape the big weight gradients (N1 % 256 == 0, N2 % 128 == 0): the slabs are re-read from L2 once
// per output tile that needs them (the 128 x 128 kernel above moves 4x the algorithmic bytes through LDS-DMA), and the
// split-M partial sums leave the chip as f32 atomics, which execute at the memory side at ~1.3 TB/s chip-wide
// (atomic bytes = workgroups x tile bytes).  Here a workgroup owns a 256 x 128 tile (3x re-read) with eight waves:
// two k-groups of four 128 x 64 wave tiles each take half of the 64 patch rows of a slab, so the grid is one
// workgroup per CU (256 workgroups x 128 KiB = 32 MiB of atomics) and the k-groups are summed through LDS at the end.
// Slabs of 48 KiB (A 64 x 256, B 64 x 128) run through a three-slot ring with two slabs in flight.
#ifndef TNW_ABLATE
#define TNW_ABLATE 0       // dev: 1 = no atomics, 2 = no LDS reads / MFMAs, 3 = no LDS-DMA
#endif
__global__ __launch_bounds__(512) void gemm_tn_wide_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                           float* __restrict__ C, int M, int N1, int N2, int lda,
                                                           int ldb, int ldc, int m_per_split, int nsplit) {
    constexpr int ROWS = 64, A_PITCH = 512, B_PITCH = 256, A_BYTES = ROWS * A_PITCH, B_BYTES = ROWS * B_PITCH;
    constexpr int SLOT = A_BYTES + B_BYTES;                                  // 48 KiB
    constexpr int NSLOT = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T1 = N1 >> 8, T2 = N2 >> 7, tiles = T1 * T2;
    int tile, sp;
    if ((nsplit & 7) == 0) {                      // all tiles of one M-split on one XCD (they share the slabs in its L2)
        const int b = blockIdx.x, xcd = b & 7, w = b >> 3;
        sp = xcd + 8 * (w / tiles);
        tile = w % tiles;
    } else {
        tile = blockIdx.x % tiles;
        sp = blockIdx.x / tiles;
    }
    const int t1 = tile % T1, t2 = tile / T1;
    const int n10 = t1 << 8, n20 = t2 << 7;
    const int mbeg = sp * m_per_split, mend = min(M, mbeg + m_per_split);
    if (mbeg >= mend) return;
    const int nslab_total = (mend - mbeg + ROWS - 1) / ROWS;
    const unsigned lds0 = lds_off(smem);
    auto swz = [](int row) { return ((row & 3) << 1) | (((row >> 3) & 1) << 3); };

    // LDS-DMA chunk maps: A slab = 2048 chunks (32 per row), B slab = 1024 chunks (16 per row); the LDS image is
    // linear, the swizzle is applied to the global source chunk
    int arow[4], acol[4], brow[2], bcol[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ci = t * 512 + tid, row = ci >> 5;
        arow[t] = row;
        acol[t] = n10 + (((ci & 31) ^ swz(row)) << 3);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ci = t * 512 + tid, row = ci >> 4;
        brow[t] = row;
        bcol[t] = n20 + (((ci & 15) ^ swz(row)) << 3);
    }
    auto stage = [&](int s) {
        if (TNW_ABLATE == 3) return;
        const int mrow0 = mbeg + (nslab_total - 1 - s) * ROWS;           // slabs are walked from the high rows down (see above)
        const unsigned la = lds0 + (s % NSLOT) * SLOT, lb = la + A_BYTES;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            glds16(A + (size_t)min(mrow0 + arow[t], M - 1) * lda + acol[t], la + (t * 512 + wave * 64) * 16);
#pragma unroll
        for (int t = 0; t < 2; ++t)
            glds16(B + (size_t)min(mrow0 + brow[t], M - 1) * ldb + bcol[t], lb + (t * 512 + wave * 64) * 16);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kg = wave >> 2, w4 = wave & 3, wr = w4 >> 1, wc = w4 & 1;

    // transposed-read offsets (lane constants): rows 32kg + 8g + rq (+4), 16-column group t -> chunks 2t, 2t+1.
    // The swizzle touches chunk bits 1..3 and so does the tile index i (j) -> offset(i) = offset(0) ^ (i << 5); the
    // second row (+4) has the same swizzle -> + 4 pitches as an immediate.
    unsigned abase, bbase;
    {
        const int g = lane >> 4, u = lane & 15, rq = u >> 2, p4 = u & 3;
        const int row = 32 * kg + 8 * g + rq;
        const int ca = 2 * (wr * 8) + (p4 >> 1), cb = 2 * (wc * 4) + (p4 >> 1);
        abase = row * A_PITCH + ((ca ^ swz(row)) << 4) + ((p4 & 1) << 3);
        bbase = A_BYTES + row * B_PITCH + ((cb ^ swz(row)) << 4) + ((p4 & 1) << 3);
    }
    auto tr_frag = [&](const char* slab, unsigned o0, int pitch4) -> bf16x8 {
        typedef __attribute__((address_space(3))) s16x4* lp;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(slab + o0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(slab + o0 + pitch4));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    const int nslab = (mend - mbeg + ROWS - 1) / ROWS;
    // make slab t ready for fragment reads: landed (slab t+1 may stay in flight), visible to all waves, ragged tail zeroed;
    // then refill the slot of slab t-1 (every wave has its fragments in registers) with slab t+2
    auto ready = [&](int t) {
        if (t + 1 < nslab) { WAIT_VMCNT(6); } else { WAIT_VMCNT(0); }
        LDS_BARRIER();
        if (t + 2 < nslab) stage(t + 2);
        char* slab = smem + (t % NSLOT) * SLOT;
        const int rows_here = min(ROWS, mend - (mbeg + (nslab - 1 - t) * ROWS));
        if (rows_here < ROWS) {
            for (int idx = tid; idx < (ROWS - rows_here) * 32; idx += 512)
                *(u32x4*)(slab + (rows_here + (idx >> 5)) * A_PITCH + (idx & 31) * 16) = u32x4{0, 0, 0, 0};
            for (int idx = tid; idx < (ROWS - rows_here) * 16; idx += 512)
                *(u32x4*)(slab + A_BYTES + (rows_here + (idx >> 4)) * B_PITCH + (idx & 15) * 16) = u32x4{0, 0, 0, 0};
            LDS_BARRIER();
        }
    };
    auto load_frags = [&](int t, bf16x8 (&af)[8], bf16x8 (&bfr)[4]) {
        const char* slab = smem + (t % NSLOT) * SLOT;
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = tr_frag(slab, bbase ^ (j << 5), 4 * B_PITCH);
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = tr_frag(slab, abase ^ (i << 5), 4 * A_PITCH);
    };
    stage(0);
    if (nslab > 1) stage(1);
    for (int s = 0; s < nslab; ++s) {
        ready(s);
        if (TNW_ABLATE == 2) continue;
        // one basic block: B fragments and the first two A fragments up front, then A fragment i+2 is requested behind
        // the MFMAs of fragment i (issue order pinned below), so only the head of a slab exposes LDS latency
        const char* slab = smem + (s % NSLOT) * SLOT;
        bf16x8 bfr[4], af[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = tr_frag(slab, bbase ^ (j << 5), 4 * B_PITCH);
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = tr_frag(slab, abase ^ (i << 5), 4 * A_PITCH);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);          // B (8 reads) + A0, A1 (4 reads)
#pragma unroll
        for (int i = 0; i < 6; ++i) {                                // MFMAs of A_i with the reads of A_{i+2} between them
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    }
    // ---- sum the two k-groups through LDS: group 0 keeps rows i < 4 of its wave tile, group 1 rows i >= 4; each wave
    // hands the other half to its partner (wave ^ 4) as [16 tiles][64 lanes] f32x4 = 16 KiB
    __syncthreads();                                // every wave is done with the slab ring
    {
        f32x4* mine = (f32x4*)smem + (size_t)wave * 1024;
        f32x4* theirs = (f32x4*)smem + (size_t)(wave ^ 4) * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mine[(i * 4 + j) * 64 + lane] = acc[kg ? i : i + 4][j];
        __syncthreads();
        const int q4 = lane >> 4, r16 = lane & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 o = theirs[(i * 4 + j) * 64 + lane];
                const f32x4 a = kg ? acc[i + 4][j] : acc[i][j];
                const int ii = kg ? i + 4 : i;
                const int n2 = n20 + wc * 64 + j * 16 + r16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n1 = n10 + wr * 128 + ii * 16 + 4 * q4 + r;
                    const float v = a[r] + o[r];
                    if (TNW_ABLATE == 1) { if (v == 1.2345e-30f) C[0] = 1.f; continue; }
                    atomicAdd(C + (size_t)n1 * ldc + n2, v);
                }
            }
    }
}

// ------------------------------------------------------------------------------------- TN, square-tile bf16 variant, grouped
// 256 x 256 output tiles (N1 % 256 == 0, N2 % 256 == 0), split over the patch rows, partial tiles stored to a workspace
// and summed by a second launch - no float atomics.  Against the 256 x 128 kernel above: every slab byte is moved
// L2 -> LDS twice instead of three times (the big weight gradients are bound by that traffic: 1.6 GB at ~12 TB/s for
// 537 MB of operands), each wave takes all 32 rows of a slab (no k-group halves, no final LDS reduction), and the
// partial tiles leave as plain 64-byte row segments (~6 TB/s) instead of memory-side atomics (~1.3 TB/s).
// Eight waves as 2 (n1) x 4 (n2) wave tiles of 128 x 64; slabs of 32 patch rows x (256 + 256) columns = 32 KiB run
// through a four-slot ring with three slabs in flight.
//
// GROUPED (round 4): one launch takes up to TN_MAXG products (the three encoder weight gradients of a backward pass, deferred
// until the last input gradient exists: abmil.py:12-21).  The launch is ONE workgroup per CU whatever the group holds: the
// (product, tile) pairs share the 256 CUs, each pair's patch rows cut into as many splits as fit (3 products x 4 tiles -> 21
// splits of ~12.5 k rows = 391 slabs per workgroup, where three separate launches ran 64 splits of 128 slabs each).  What that
// buys: ring fill, weight-less prologue, the partial-tile epilogue and the launch ramp are paid once per step instead of three
// times (~24 us of every launch did not shrink with M, DESIGN r3), the partial tiles drop from 3 x 64 x 1 MiB to 21 x 3 MiB
// written and read once, and two reduce launches disappear.  Partials stay f32.
#ifndef TN_SQ_NSLOT
#define TN_SQ_NSLOT 4          // ring slots of 32 KiB (5 = all 160 KiB of LDS, four slabs in flight)
#endif
constexpr int TN_MAXG = 4, TN_MAXWG = 256;
struct TnGroupArgs {
    const bf16_t* A[TN_MAXG];
    const bf16_t* B[TN_MAXG];
    long part_off[TN_MAXG];          // float offset of product g's partial tiles [sp_g][N1_g * N2_g] in the workspace
    int M[TN_MAXG], N1[TN_MAXG], N2[TN_MAXG], lda[TN_MAXG], ldb[TN_MAXG], mps[TN_MAXG];
    unsigned map[TN_MAXWG];          // workgroup -> (g << 28) | (tile << 16) | split; 0xFFFFFFFF: no work
};
// NWV waves per workgroup: 8 = 2 (n1) x 4 (n2) wave tiles of 128 x 64, two waves per SIMD (rounds 3-5); 4 = 2 x 2 wave tiles of
// 128 x 128, ONE wave per SIMD with a 256-register accumulator (round 6 A/B, -DTN_SQ_WAVES=4): a slab's A half is read from LDS by
// two waves instead of four - 64 KiB of fragment reads per slab where the 8-wave form reads 96 - for the same MFMA count per CU.
#ifndef TN_SQ_WAVES
#define TN_SQ_WAVES 8
#endif
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void gemm_tn_sq_kernel(const TnGroupArgs ga, float* __restrict__ ws) {
    constexpr int ROWS = 32, PITCH = 512, A_BYTES = ROWS * PITCH, SLOT = 2 * A_BYTES, NSLOT = TN_SQ_NSLOT;   // 32 KiB slots
    constexpr int NTH = 64 * NWV, NT = 1024 / NTH;          // threads; 16-byte chunks of a slab half per thread: 2 / 4
    constexpr int WCN = NWV / 2, NJ = 16 / WCN;             // wave columns: 4 / 2; 16-column MFMA tiles per wave along n2: 4 / 8
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned code = ga.map[blockIdx.x];
    if (code == 0xFFFFFFFFu) return;
    const int g = code >> 28, tile = (code >> 16) & 0xFFF, sp = code & 0xFFFF;
    const bf16_t* __restrict__ A = ga.A[g];
    const bf16_t* __restrict__ B = ga.B[g];
    const int M = ga.M[g], N1 = ga.N1[g], N2 = ga.N2[g], lda = ga.lda[g], ldb = ga.ldb[g], m_per_split = ga.mps[g];
    float* __restrict__ part = ws + ga.part_off[g];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T1 = N1 >> 8;
    const int t1 = tile % T1, t2 = tile / T1;
    const int n10 = t1 << 8, n20 = t2 << 8;
    const int mbeg = sp * m_per_split, mend = min(M, mbeg + m_per_split);
    const int nslab = mbeg < mend ? (mend - mbeg + ROWS - 1) / ROWS : 0;
    const unsigned lds0 = lds_off(smem);
    auto swz = [](int row) { return ((row & 3) << 1) | (((row >> 3) & 1) << 3); };

    // LDS-DMA chunk maps: a slab half = 32 rows x 32 chunks of 16 B = 1024 chunks, two per thread; linear LDS image,
    // swizzle applied to the global source chunk.  Row pointers advance by a constant per slab (no per-load 64-bit math).
    const bf16_t* ap[NT];
    const bf16_t* bp[NT];
    int srow[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ci = t * NTH + tid, row = ci >> 5;
        srow[t] = row;
        ap[t] = A + (size_t)n10 + (((ci & 31) ^ swz(row)) << 3);
        bp[t] = B + (size_t)n20 + (((ci & 31) ^ swz(row)) << 3);
    }
    auto stage = [&](int s) {
        const int mrow0 = mbeg + (nslab - 1 - s) * ROWS;                 // slabs are walked from the high rows down
        const unsigned la = lds0 + (s % NSLOT) * SLOT, lb = la + A_BYTES;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const size_t r = (size_t)min(mrow0 + srow[t], M - 1);
            glds16(ap[t] + r * lda, la + (t * NTH + wave * 64) * 16);
            glds16(bp[t] + r * ldb, lb + (t * NTH + wave * 64) * 16);
        }
    };

    f32x4 acc[8][NJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wr = wave / WCN, wc = wave % WCN;

    // transposed-read offsets (lane constants): rows 8g + rq (+4), 16-column group t -> chunks 2t, 2t+1; tile index i
    // (j) -> offset ^ (i << 5) as in the kernel above
    unsigned abase, bbase;
    {
        const int gq = lane >> 4, u = lane & 15, rq = u >> 2, p4 = u & 3;
        const int row = 8 * gq + rq;
        const int ca = 2 * (wr * 8) + (p4 >> 1), cb = 2 * (wc * NJ) + (p4 >> 1);
        abase = row * PITCH + ((ca ^ swz(row)) << 4) + ((p4 & 1) << 3);
        bbase = A_BYTES + row * PITCH + ((cb ^ swz(row)) << 4) + ((p4 & 1) << 3);
    }
    auto tr_frag = [&](const char* slab, unsigned o0) -> bf16x8 {
        typedef __attribute__((address_space(3))) s16x4* lp;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(slab + o0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(slab + o0 + 4 * PITCH));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    const int pre = min(NSLOT - 1, nslab);
    for (int s = 0; s < pre; ++s) stage(s);
    for (int s = 0; s < nslab; ++s) {
        // slab s landed (up to NSLOT - 2 younger slabs stay in flight: 2 NT LDS-DMA ops per thread and slab), visible to all waves;
        // then the slot of slab s-1 - every wave has its fragments in registers - takes slab s + NSLOT - 1
        if (NSLOT >= 5 && s + 3 < nslab) { if (NT == 2) { WAIT_VMCNT(12); } else { WAIT_VMCNT(24); } }
        else if (s + 2 < nslab) { if (NT == 2) { WAIT_VMCNT(8); } else { WAIT_VMCNT(16); } }
        else if (s + 1 < nslab) { if (NT == 2) { WAIT_VMCNT(4); } else { WAIT_VMCNT(8); } } else { WAIT_VMCNT(0); }
        LDS_BARRIER();
        if (s + NSLOT - 1 < nslab) stage(s + NSLOT - 1);
        char* slab = smem + (s % NSLOT) * SLOT;
        const int rows_here = min(ROWS, mend - (mbeg + (nslab - 1 - s) * ROWS));
        if (rows_here < ROWS) {                                              // ragged tail of the split: zero the missing rows
            for (int idx = tid; idx < (ROWS - rows_here) * 32; idx += NTH) {
                *(u32x4*)(slab + (rows_here + (idx >> 5)) * PITCH + (idx & 31) * 16) = u32x4{0, 0, 0, 0};
                *(u32x4*)(slab + A_BYTES + (rows_here + (idx >> 5)) * PITCH + (idx & 31) * 16) = u32x4{0, 0, 0, 0};
            }
            LDS_BARRIER();
        }
        bf16x8 bfr[NJ], af[8];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[j] = tr_frag(slab, bbase ^ (j << 5));
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = tr_frag(slab, abase ^ (i << 5));
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NJ + 4, 0);  // B (2 NJ reads) + A0, A1 (4 reads)
#pragma unroll
        for (int i = 0; i < 6; ++i) {                                // MFMAs of A_i with the reads of A_{i+2} between them
            __builtin_amdgcn_sched_group_barrier(0x008, NJ / 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NJ / 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NJ / 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NJ, 0);
    }
    // ---- this split's partial tile -> the workspace in the matrix's own row-major layout: part[sp][n1][n2] (the accumulator-
    // layout alternative - 32 whole-KiB stores per wave instead of 128 four-byte ones - measured neutral in round 3)
    float* pt = part + (size_t)sp * N1 * N2;
    const int q4 = lane >> 4, r16 = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n2 = n20 + wc * (16 * NJ) + j * 16 + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n1 = n10 + wr * 128 + i * 16 + 4 * q4 + r;
                pt[(size_t)n1 * N2 + n2] = acc[i][j][r];
            }
        }
}
// C_g[n] += sum_s part_g[s][n]   (n over N1*N2 elements as float4; four splits of loads in flight per thread), every product of
// the group in ONE launch.  Blocks past the matrix parts add up [cs_rows][N1] partial column-sum rows into cs_out the same
// way: the bias gradient of a layer, whose partial rows the input-gradient kernel left behind, rides along.
struct TnReduceArgs {
    float* C[TN_MAXG];
    const float* cs_part[TN_MAXG];
    float* cs_out[TN_MAXG];
    long part_off[TN_MAXG], n4[TN_MAXG];
    int nsplit[TN_MAXG], N1[TN_MAXG], N2[TN_MAXG], ldc[TN_MAXG], cs_rows[TN_MAXG], flags[TN_MAXG];
    float scale[TN_MAXG];
    int blk0[TN_MAXG + 1];           // first matrix block of product g (blk0[n] = all matrix blocks)
    int cs_blk0[TN_MAXG + 1];        // first column-sum block of product g, counted from blk0[n]
    int n;
};
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ ws, const TnReduceArgs ra) {
    int blk = blockIdx.x;
    if (blk >= ra.blk0[ra.n]) {
        // column-sum blocks: 16 four-column groups x 16 row slices per block, every thread's rows in flight at once, the
        // slices meet in LDS (one serial pass over all rows per thread would take one memory round trip per few rows)
        blk -= ra.blk0[ra.n];
        int g = 0;
        while (g + 1 < ra.n && blk >= ra.cs_blk0[g + 1]) ++g;
        blk -= ra.cs_blk0[g];
        const float* __restrict__ cs_part = ra.cs_part[g];
        if (cs_part == nullptr) return;
        const int cs_rows = ra.cs_rows[g];
        __shared__ f32x4 red[256];
        const int cg = threadIdx.x & 15, rs = threadIdx.x >> 4;
        const long j = (long)blk * 16 + cg;
        const int ng = ra.N1[g] / 4;
        f32x4 b0 = {0, 0, 0, 0}, b1 = b0;
        if (j < ng) {
            const f32x4* q = (const f32x4*)cs_part + j;
            for (int r0 = 0; r0 < cs_rows; r0 += 256) {           // <= 256 rows per pass: 16 independent loads per thread
                f32x4 v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int r = r0 + rs + 16 * k;
                    v[k] = r < cs_rows ? q[(size_t)r * ng] : f32x4{0, 0, 0, 0};
                }
#pragma unroll
                for (int k = 0; k < 16; k += 2) { b0 += v[k]; b1 += v[k + 1]; }
            }
        }
        red[threadIdx.x] = b0 + b1;
        __syncthreads();
        if (rs == 0 && j < ng) {
            f32x4 t = red[cg];
#pragma unroll
            for (int k = 1; k < 16; ++k) t += red[16 * k + cg];
            if (ra.flags[g] & MURCL_TN_SCALE) t *= ra.scale[g];
            f32x4* o = (f32x4*)ra.cs_out[g] + j;
            *o = (ra.flags[g] & MURCL_TN_OVERWRITE) ? t : *o + t;
        }
        return;
    }
    int g = 0;
    while (g + 1 < ra.n && blk >= ra.blk0[g + 1]) ++g;
    const long i = (long)(blk - ra.blk0[g]) * 256 + threadIdx.x;
    const long n4 = ra.n4[g];
    if (i >= n4) return;
    const int nsplit = ra.nsplit[g], N2 = ra.N2[g];
    const f32x4* p = (const f32x4*)(ws + ra.part_off[g]) + i;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    int s = 0;
    for (; s + 3 < nsplit; s += 4) {
        const f32x4 v0 = p[(size_t)s * n4], v1 = p[(size_t)(s + 1) * n4], v2 = p[(size_t)(s + 2) * n4], v3 = p[(size_t)(s + 3) * n4];
        a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; s < nsplit; ++s) a0 += p[(size_t)s * n4];
    f32x4 t = (a0 + a1) + (a2 + a3);
    const int fl = ra.flags[g];
    if (fl & MURCL_TN_SCALE) t *= ra.scale[g];
    const long e = i * 4, col = e % N2;
    long row = e / N2;
    if (fl & MURCL_TN_DEINTERLEAVE) row = ((row >> 4) & 1) * (ra.N1[g] >> 1) + (row >> 5) * 16 + (row & 15);
    f32x4* c = (f32x4*)(ra.C[g] + row * ra.ldc[g] + col);
    *c = (fl & MURCL_TN_OVERWRITE) ? t : *c + t;
}
static bool tn_sq_ok(int M, int N1, int N2, int ldc, int dtype) {
    return dtype == MURCL_DTYPE_BF16 && N1 % 256 == 0 && N2 % 256 == 0 && M >= 16384 && ldc % 4 == 0;
}
// The launch plan of a group: rows per split and split count of every product such that all (product, tile, split) workgroups
// fit one round of the chip (<= 256, one 128 KiB-LDS workgroup per CU) with about equal row counts, and the workgroup -> work
// map: the tiles of one (product, split) read the same slabs, so they sit on ONE XCD (blocks b and b + 8k share an XCD's L2).
struct TnPlan { int sp[TN_MAXG], mps[TN_MAXG], tiles[TN_MAXG]; long part_off[TN_MAXG]; long ws_floats; int wgs; };
static bool tn_group_plan(int n, const int* M, const int* N1, const int* N2, TnPlan* pl, unsigned* map) {
    if (n < 1 || n > TN_MAXG) return false;
    long work = 0;
    int pairs = 0;
    for (int g = 0; g < n; ++g) {
        pl->tiles[g] = (N1[g] / 256) * (N2[g] / 256);
        if (pl->tiles[g] > murcl_cu_budget() / 8) return false;      // a (product, split) group must fit an XCD's workgroup slots
        work += (long)pl->tiles[g] * M[g];
        pairs += pl->tiles[g];
    }
    const int cap = murcl_cu_budget();                                  // workgroups of the one round (<= TN_MAXWG), cap / 8 per XCD
    if (pairs > cap) return false;
    long R = ((work + cap - 1) / cap + 31) / 32 * 32;                  // rows per workgroup, whole slabs
    if (R < 8 * 32) R = 8 * 32;                                        // keep >= 8 slabs per split
    for (;; R += ((R >> 8) + 31) / 32 * 32) {             // (steps of ~0.4 %: the search ends within a few hundred probes for any row counts)
        int total = 0;
        for (int g = 0; g < n; ++g) {
            pl->sp[g] = (int)((M[g] + R - 1) / R);
            total += pl->sp[g] * pl->tiles[g];
        }
        if (total <= cap) { pl->wgs = total; break; }
    }
    long off = 0;
    for (int g = 0; g < n; ++g) {
        long mps = ((M[g] + pl->sp[g] - 1) / pl->sp[g] + 31) / 32 * 32;  // even shares inside a product (R is only the cap)
        pl->mps[g] = (int)mps;
        pl->sp[g] = (int)((M[g] + mps - 1) / mps);
        pl->part_off[g] = off;
        off += (long)pl->sp[g] * N1[g] * N2[g];
    }
    pl->ws_floats = off;
    if (!map) return true;
    for (int b = 0; b < TN_MAXWG; ++b) map[b] = 0xFFFFFFFFu;
    int used[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool fits = true;
    // splits in the outer loop: the products' groups alternate, so every XCD streams a mix of all operands
    int spmax = 0;
    for (int g = 0; g < n; ++g) spmax = pl->sp[g] > spmax ? pl->sp[g] : spmax;
    for (int s = 0; s < spmax && fits; ++s)
        for (int g = 0; g < n; ++g) {
            if (s >= pl->sp[g]) continue;
            int x = 0;
            for (int k = 1; k < 8; ++k) if (used[k] < used[x]) x = k;
            if (used[x] + pl->tiles[g] > cap / 8) { fits = false; break; }
            for (int t = 0; t < pl->tiles[g]; ++t) map[x + 8 * (used[x] + t)] = ((unsigned)g << 28) | ((unsigned)t << 16) | (unsigned)s;
            used[x] += pl->tiles[g];
        }
    if (!fits) {                                         // fragmentation: plain linear fill (correct, tiles may straddle XCDs)
        int b = 0;
        for (int bb = 0; bb < TN_MAXWG; ++bb) map[bb] = 0xFFFFFFFFu;
        for (int g = 0; g < n; ++g)
            for (int s = 0; s < pl->sp[g]; ++s)
                for (int t = 0; t < pl->tiles[g]; ++t) map[b++] = ((unsigned)g << 28) | ((unsigned)t << 16) | (unsigned)s;
    }
    return true;
}
// bytes of workspace murcl_gemm_tn_ws wants for this shape (0: the shape takes the atomics path, no workspace needed)
extern "C" long murcl_gemm_tn_workspace_bytes(int M, int N1, int N2, int dtype) {
    if (!tn_sq_ok(M, N1, N2, N2, dtype)) return 0;
    TnPlan pl;
    if (!tn_group_plan(1, &M, &N1, &N2, &pl, nullptr)) return 0;
    return pl.ws_floats * 4;
}

extern "C" int murcl_colsum(const void* x, float* out, int R, int N, int ld, int dtype, int accumulate, hipStream_t s);

// Bag-level f32 weight gradients (M = a few hundred rows, outputs up to [3072 x 512]): C[N1,N2] += A[M,N1]^T B[M,N2].
// The 128 x 128-tile kernel above puts such a product on N1*N2/16384 workgroups of 512 exact-f32 MFMAs per wave and 128 rows
// (6.8 us of matrix pipe each; [512 x 512] is 16 workgroups on a 256-CU chip).  Here a workgroup owns a 32 x 32 output
// tile - one 16 x 16 MFMA tile per wave, single writer, no atomics - and the whole reduction (<= 512 rows per pass) sits in
// LDS: rows of 32 floats of both operands arrive by LDS-DMA, eight rows per instruction, the 16-byte chunks of a row
// XOR-swizzled at the SOURCE by bit 1 of the row index so that the four lane quarters of a fragment read (rows k..k+3, 16
// consecutive floats each) fall on four disjoint 16-bank groups.  Column sums of A (the bias gradient) come from one more
// MFMA per step against a fragment of ones in the workgroups of the first N2 tile.
constexpr int TS_T = 32, TS_MAXM = 512;
__device__ __forceinline__ void tn_small_tile(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M,
                                              int N1, int N2, int lda, int ldb, int ldc, float* __restrict__ colsum_out, int bx,
                                              int by, char* ts_smem, int pass_rows, int overwrite = 0) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q4 = lane >> 4, r16 = lane & 15;
    const int n10 = bx * TS_T, n20 = by * TS_T;
    const unsigned lds0 = lds_off(ts_smem);
    const int wm = wave & 1, wn = wave >> 1;
    const bool do_cs = colsum_out != nullptr && by == 0 && wn == 0;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, accs = acc0;
    // what the epilogue adds to, requested before the operands so that the read-modify-write does not wait at the end
    f32x4 cold = {0.f, 0.f, 0.f, 0.f};
    {
        const int n2 = n20 + 16 * wn + r16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n1 = n10 + 16 * wm + 4 * q4 + r;
            if (!overwrite && n1 < N1 && n2 < N2) cold[r] = C[(size_t)n1 * ldc + n2];
        }
    }
    // fragment addresses: row 16u + 4e + q4, column 16w + r16 -> chunk (4w + r16/4) ^ swz(row), swz = 4 * bit 1 of the row = 4 * (q4 >> 1)
    const int sw = (q4 >> 1) << 2;
    const unsigned aoff = q4 * 128 + ((((4 * wm + (r16 >> 2)) ^ sw) << 4) | ((r16 & 3) << 2));
    const unsigned boff = q4 * 128 + ((((4 * wn + (r16 >> 2)) ^ sw) << 4) | ((r16 & 3) << 2));
    for (int mb = 0; mb < M; mb += pass_rows) {
        const int mc = min(pass_rows, M - mb), mp = (mc + 15) & ~15, ng = (mc + 7) >> 3;
        const unsigned slab_b = (unsigned)mp * 128;
        if (mb) __syncthreads();                                  // the previous pass has been read by every wave
        for (int i = wave; i < 2 * ng; i += 4) {                  // wave-uniform: one LDS-DMA instruction = 8 rows of one slab
            const bool isb = i >= ng;
            const int gi = isb ? i - ng : i;
            const int lrow = 8 * gi + (lane >> 3);
            const int g = (lane & 7) ^ (((lrow >> 1) & 1) << 2);
            const int row = mb + min(lrow, mc - 1);               // rows past the end: a valid row, zeroed below
            const float* src = isb ? B + (size_t)row * ldb + min(n20 + 4 * g, N2 - 4) : A + (size_t)row * lda + min(n10 + 4 * g, N1 - 4);
            glds16(src, lds0 + (isb ? slab_b : 0u) + gi * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (mc & 15) {                                            // ragged tail of the last 16-row group: zeros
            const int nz = (mp - mc) * 32;
            for (int i = tid; i < 2 * nz; i += 256) {
                const int which = i >= nz, j = which ? i - nz : i;
                *(float*)(ts_smem + (which ? slab_b : 0u) + (mc * 32 + j) * 4) = 0.f;
            }
            __syncthreads();
        }
        const char* la = ts_smem + aoff;
        const char* lb = ts_smem + slab_b + boff;
        for (int u = 0; u < mp / 16; ++u) {
            f32x4 a, b;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = *(const float*)(la + (16 * u + 4 * e) * 128);
                b[e] = *(const float*)(lb + (16 * u + 4 * e) * 128);
            }
            if (u & 1) acc1 = mma16<float>(a, b, acc1); else acc0 = mma16<float>(a, b, acc0);
            if (do_cs) accs = mma16<float>(a, f32x4{1.f, 1.f, 1.f, 1.f}, accs);
        }
    }
    // lane holds C[n10 + 16wm + 4q4 + r][n20 + 16wn + r16]: the 16 lanes of a quarter update 64 contiguous bytes of a row
    const int n2 = n20 + 16 * wn + r16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n1 = n10 + 16 * wm + 4 * q4 + r;
        if (n1 < N1 && n2 < N2) C[(size_t)n1 * ldc + n2] = cold[r] + (acc0[r] + acc1[r]);
        if (do_cs && r16 == 0 && n1 < N1) colsum_out[n1] = overwrite ? accs[r] : colsum_out[n1] + accs[r];
    }
}
__global__ __launch_bounds__(256) void gemm_tn_small_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                float* __restrict__ C, int M, int N1, int N2, int lda,
                                                                int ldb, int ldc, float* __restrict__ colsum_out, int pass_rows) {
    extern __shared__ __attribute__((aligned(16))) char ts_smem[];
    tn_small_tile(A, B, C, M, N1, N2, lda, ldb, ldc, colsum_out, blockIdx.x, blockIdx.y, ts_smem, pass_rows);
}
// Up to four such products as ONE launch (the weight gradients of a PPO epoch, rlmil.py:179: four launches of 16 us whose
// tails and launch gaps add up): workgroup b belongs to the product g with tile0[g] <= b < tile0[g + 1]; its tile index
// runs along N1 first, as the single-product grid does.
struct TnSmallGroup {
    const float* A[4]; const float* B[4]; float* C[4]; float* cs[4];
    int M[4], N1[4], N2[4], lda[4], ldb[4], ldc[4], tile0[5], pass_rows, overwrite[4];
};
__global__ __launch_bounds__(256) void gemm_tn_small_group_kernel(const TnSmallGroup ga) {
    extern __shared__ __attribute__((aligned(16))) char ts_smem[];
    const int b = blockIdx.x;
    int g = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) g += (int)(b >= ga.tile0[i]);
    const int t = b - ga.tile0[g], t1 = (ga.N1[g] + TS_T - 1) / TS_T;
    tn_small_tile(ga.A[g], ga.B[g], ga.C[g], ga.M[g], ga.N1[g], ga.N2[g], ga.lda[g], ga.ldb[g], ga.ldc[g], ga.cs[g], t % t1, t / t1,
                  ts_smem, ga.pass_rows, ga.overwrite[g]);
}
// Rows of the reduction per pass through LDS (2 x 128 bytes per row).  One pass (the whole reduction in flight, one memory round
// trip) when a single round of workgroups covers the tiles; several shorter passes when there are more tiles than that - the
// smaller footprint puts more workgroups on a CU and THEIR round trips overlap (profiles/r04_r_tn_pass*.txt)
static int tn_small_pass_rows(int M, long tiles) {
    int pr = TS_MAXM;
    if (tiles > 512 && M > 192) pr = ((M + 1) / 2 + 15) & ~15;
    pr = (pr + 15) & ~15;
    if (pr > TS_MAXM) pr = TS_MAXM;
    if (pr < 16) pr = 16;
    return pr;
}


extern "C" int murcl_gemm_tn(const void* A, const void* B, float* C, int M, int N1, int N2, int lda, int ldb, int ldc,
                             int dtype, int splits, float* colsum_out, hipStream_t stream) {
    if (M <= 0 || N1 <= 0 || N2 <= 0) return 0;
    const bool x3 = dtype == MURCL_DTYPE_F32X3;          // f32 storage, 3-term bf16 split on the bf16 matrix pipe (gemm_nt above)
    if (x3) dtype = MURCL_DTYPE_F32;
    const int es = dtype == MURCL_DTYPE_BF16 ? 2 : 4, epc = 16 / es;
    if (N1 < epc || N2 < epc || N1 % epc || N2 % epc || (lda * es) % 16 || (ldb * es) % 16) return -1;
    if (dtype == MURCL_DTYPE_BF16 && N1 % 256 == 0 && N2 % 128 == 0 && M >= 4096) {
        const int tiles = (N1 / 256) * (N2 / 128);
        int sp = splits;
        if (sp <= 0) {                           // one 144 KiB-LDS workgroup per CU, splits % 8 == 0
            sp = (256 + tiles - 1) / tiles;
            sp = ((sp + 7) / 8) * 8;
            while (sp > 8 && (long)(sp - 8) * 64 * 4 >= M) sp -= 8;                 // keep >= 4 slabs per split
        }
        int mps = (M + sp - 1) / sp;
        mps = ((mps + 63) / 64) * 64;
        if ((long)mps * (sp - 1) >= M) sp = (M + mps - 1) / mps;
        if (colsum_out) {                        // the wide kernel has no spare MFMA slots for the ones-trick: separate pass
            const int rc = murcl_colsum(A, colsum_out, M, N1, lda, dtype, 1, stream);
            if (rc) return rc;
        }
        auto k = gemm_tn_wide_kernel;
        constexpr int LDS = 3 * 49152;
        static MurclOncePerDevice once;      
        if (once.first()) { hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); }
        hipLaunchKernelGGL(k, dim3(tiles * sp), dim3(512), LDS, stream, (const bf16_t*)A, (const bf16_t*)B, C, M, N1, N2,
                           lda, ldb, ldc, mps, sp);
        return MURCL_CHECK_LAUNCH();
    }
    // one pass only: at M = 768 (the deferred head gradients of a T = 6 step) two serial passes per 32 x 32 tile take 78 us where
    // the 128 x 128 ring kernel below takes 55 us ([768 x 3072 x 512], tools/tn_trace.sh); up to 512 rows the small tiles win
    // (16.6 -> 12.2 us [128 x 3072 x 512], 7.7 -> 5.8 us [128 x 512 x 512], 23.7 -> 18.0 us [320 x 2048 x 512])
    if (dtype == MURCL_DTYPE_F32 && !x3 && splits <= 0 && M <= TS_MAXM) {
        const int pr = tn_small_pass_rows(M, (long)((N1 + TS_T - 1) / TS_T) * ((N2 + TS_T - 1) / TS_T));
        const int mp = ((M < pr ? M : pr) + 15) & ~15;
        static MurclOncePerDevice once;
        if (once.first())
            hipFuncSetAttribute((const void*)gemm_tn_small_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TS_MAXM * 128);
        hipLaunchKernelGGL(gemm_tn_small_f32_kernel, dim3((N1 + TS_T - 1) / TS_T, (N2 + TS_T - 1) / TS_T), dim3(256), 2 * mp * 128,
                           stream, (const float*)A, (const float*)B, C, M, N1, N2, lda, ldb, ldc, colsum_out, pr);
        return MURCL_CHECK_LAUNCH();
    }
    const int rows = dtype == MURCL_DTYPE_BF16 ? 64 : 32;
    const int t1 = (N1 + 127) / 128, t2 = (N2 + 127) / 128;
    if (splits <= 0 && M <= 8 * rows && (long)N1 * N2 >= (1L << 20)) {
        // bag-level layers: the whole reduction fits the four-slot ring (one memory round trip per workgroup), and every
        // extra M-split would add N1*N2*4 bytes of float atomics (dW_ih: 6.3 MB each, ~5 us at the memory side)
        splits = 1;
    } else if (splits <= 0) {
        // ONE workgroup per CU (then the four-slot ring: three slabs in flight), not two: measured per shape with
        // tools/_tn128.py - dWa [262144 x 128]^T [. x 512] bf16 78 -> 67 us (64 splits instead of 128), DSMIL's dWq
        // [131072 x 128]^T [. x 1024] 71 -> 63 us (32 instead of 64), the deferred head gradients [768 x 3072]^T [. x 512]
        // f32 56 -> 40 us (3 instead of 8) and [. x 1024] 99 -> 62 us (2 instead of 8); the exact-f32 dWq [131072 x 128]^T [. x 1024] is MFMA-bound
        // (313 against 289 us with two workgroups per CU: long f32 reductions keep the 512-workgroup target)
        const int target = (dtype == MURCL_DTYPE_F32 && !x3 && M >= 16384) ? 2 * murcl_cu_budget() : murcl_cu_budget();
        splits = (target + t1 * t2 - 1) / (t1 * t2);
        if (splits >= 8) splits = ((splits + 7) / 8) * 8;                       // multiples of 8: the XCD-aware work map
        if (murcl_cu_budget() < 256 && splits >= 8)                             // a reduced CU budget is a cap: round DOWN
            while (splits > 8 && splits * t1 * t2 > target) splits -= 8;
        while (splits > 8 && (long)(splits - 8) * rows * 4 >= M) splits -= 8;   // keep >= 4 slabs per split
        while (splits > 1 && splits < 8 && (long)(splits - 1) * rows * 4 >= M) --splits;
    }
    int mps = (M + splits - 1) / splits;
    mps = ((mps + rows - 1) / rows) * rows;
    if ((long)mps * (splits - 1) >= M) splits = (M + mps - 1) / mps;           // tiny M: drop empty splits
    dim3 grid(t1 * t2 * splits);
    const bool deep = (t1 * t2 * splits <= 256 || splits == 1) && mps > rows;   // small grid, several slabs per workgroup
#define TN_LAUNCH(T, NS, X3)                                                                                        \
    {                                                                                                               \
        auto k = gemm_tn_kernel<T, NS, X3>;                                                                         \
        static MurclOncePerDevice once;                                                                                         \
        if (once.first()) { hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, NS * 32768); } \
        hipLaunchKernelGGL(k, grid, dim3(256), NS * 32768, stream, (const T*)A, (const T*)B, C, M, N1, N2, lda, ldb, ldc, mps, \
                           splits, colsum_out);                                                                     \
    }
    if (dtype == MURCL_DTYPE_BF16) {
        if (deep) TN_LAUNCH(bf16_t, 4, false) else TN_LAUNCH(bf16_t, 2, false)
    } else if (dtype == MURCL_DTYPE_F32 && x3) {
        if (deep) TN_LAUNCH(float, 4, true) else TN_LAUNCH(float, 2, true)
    } else if (dtype == MURCL_DTYPE_F32) {
        if (deep) TN_LAUNCH(float, 4, false) else TN_LAUNCH(float, 2, false)
    } else {
        return -1;
    }
#undef TN_LAUNCH
    return MURCL_CHECK_LAUNCH();
}

// One launch of the grouped square-tile kernel + one reduce launch for n eligible products (tn_sq_ok, 16-byte aligned rows).
typedef murcl_tn_problem TnProblem;     // include/murcl_amd.h
static int tn_sq_launch(const TnProblem* pr, int n, float* ws, hipStream_t stream) {
    int M[TN_MAXG], N1[TN_MAXG], N2[TN_MAXG];
    for (int g = 0; g < n; ++g) { M[g] = pr[g].M; N1[g] = pr[g].N1; N2[g] = pr[g].N2; }
    TnPlan pl;
    TnGroupArgs ga;
    if (!tn_group_plan(n, M, N1, N2, &pl, ga.map)) return -1;
    TnReduceArgs ra;
    ra.n = n;
    int blk = 0, csb = 0;
    for (int g = 0; g < TN_MAXG; ++g) {
        const int h = g < n ? g : 0;                          // unused slots repeat product 0 (never indexed)
        ga.A[g] = (const bf16_t*)pr[h].A; ga.B[g] = (const bf16_t*)pr[h].B;
        ga.M[g] = pr[h].M; ga.N1[g] = pr[h].N1; ga.N2[g] = pr[h].N2; ga.lda[g] = pr[h].lda; ga.ldb[g] = pr[h].ldb;
        ga.mps[g] = pl.mps[h]; ga.part_off[g] = pl.part_off[h];
        ra.C[g] = pr[h].C; ra.cs_part[g] = pr[h].colsum_part; ra.cs_out[g] = pr[h].colsum_out; ra.cs_rows[g] = pr[h].colsum_rows;
        ra.part_off[g] = pl.part_off[h]; ra.n4[g] = (long)pr[h].N1 * pr[h].N2 / 4; ra.nsplit[g] = pl.sp[h];
        ra.N1[g] = pr[h].N1; ra.N2[g] = pr[h].N2; ra.ldc[g] = pr[h].ldc; ra.flags[g] = pr[h].flags; ra.scale[g] = pr[h].scale;
        if (g < n) {
            ra.blk0[g] = blk; blk += (int)((ra.n4[g] + 255) / 256);
            ra.cs_blk0[g] = csb; csb += pr[g].colsum_part ? (pr[g].N1 / 4 + 15) / 16 : 0;
        }
    }
    for (int g = n; g <= TN_MAXG; ++g) { ra.blk0[g] = blk; ra.cs_blk0[g] = csb; }
    auto k = gemm_tn_sq_kernel<TN_SQ_WAVES>;
    constexpr int LDS = TN_SQ_NSLOT * 32768;
    static MurclOncePerDevice once;
    if (once.first()) { hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); }
    hipLaunchKernelGGL(k, dim3(TN_MAXWG), dim3(64 * TN_SQ_WAVES), LDS, stream, ga, ws);
    int rc = MURCL_CHECK_LAUNCH();
    if (rc) return rc;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)(blk + csb)), dim3(256), 0, stream, (const float*)ws, ra);
    return MURCL_CHECK_LAUNCH();
}
static bool tn_sq_problem_ok(const TnProblem& p, int dtype) {
    return tn_sq_ok(p.M, p.N1, p.N2, p.ldc, dtype) && (p.lda * 2) % 16 == 0 && (p.ldb * 2) % 16 == 0 &&
           (!p.colsum_part || (p.colsum_out && p.colsum_rows > 0));
}

// murcl_gemm_tn with a caller-provided workspace: the big bf16 weight gradients (N1, N2 multiples of 256, M >= 16384) run
// on 256 x 256 tiles with the split partial sums stored to `ws` and added to C by a reduce launch (no float atomics);
// every other shape, or a workspace that is too small, falls through to murcl_gemm_tn.
extern "C" int murcl_gemm_tn_ws(const void* A, const void* B, float* C, int M, int N1, int N2, int lda, int ldb, int ldc,
                                int dtype, int splits, float* colsum_out, float* ws, long ws_bytes, const float* colsum_part,
                                int colsum_rows, hipStream_t stream) {
    if (M <= 0 || N1 <= 0 || N2 <= 0) return 0;
    if (colsum_part && (!colsum_out || colsum_rows <= 0 || N1 % 4)) return -1;
    const long need = murcl_gemm_tn_workspace_bytes(M, N1, N2, dtype);
    if (splits > 0 || !ws || !need || !tn_sq_ok(M, N1, N2, ldc, dtype) || (lda * 2) % 16 || (ldb * 2) % 16 || ws_bytes < need) {
        if (colsum_part) {                      // the partial rows are added up by their own small launch
            const int rc = murcl_colsum(colsum_part, colsum_out, colsum_rows, N1, N1, MURCL_DTYPE_F32, 1, stream);
            if (rc) return rc;
            colsum_out = nullptr;
        }
        return murcl_gemm_tn(A, B, C, M, N1, N2, lda, ldb, ldc, dtype, splits, colsum_out, stream);
    }
    if (colsum_out && !colsum_part) {
        const int rc = murcl_colsum(A, colsum_out, M, N1, lda, dtype, 1, stream);
        if (rc) return rc;
    }
    const TnProblem p{A, B, C, colsum_part, colsum_part ? colsum_out : nullptr, M, N1, N2, lda, ldb, ldc, colsum_rows, 0, 1.f};
    return tn_sq_launch(&p, 1, ws, stream);
}

// Several weight gradients C_g[N1_g,N2_g] += A_g[M_g,N1_g]^T B_g[M_g,N2_g] in ONE launch of the square-tile kernel + ONE reduce
// launch (the three encoder layers of a backward pass: abmil.py:12-21; CLAM-SB's fc + gate pair: clam.py:69-72).  colsum_part /
// colsum_rows / colsum_out per product as in murcl_gemm_tn_ws (colsum_out without colsum_part: the column sums of A_g by their
// own launch).  Products the square-tile kernel does not take, or a workspace below murcl_gemm_tn_grouped_workspace_bytes, run one
// by one through murcl_gemm_tn_ws with the same workspace.
extern "C" long murcl_gemm_tn_grouped_workspace_bytes(const TnProblem* pr, int n, int dtype) {
    if (n < 1 || n > TN_MAXG) return 0;
    int M[TN_MAXG], N1[TN_MAXG], N2[TN_MAXG];
    for (int g = 0; g < n; ++g) {
        if (!tn_sq_problem_ok(pr[g], dtype)) return 0;
        M[g] = pr[g].M; N1[g] = pr[g].N1; N2[g] = pr[g].N2;
    }
    TnPlan pl;
    if (!tn_group_plan(n, M, N1, N2, &pl, nullptr)) return 0;
    return pl.ws_floats * 4;
}
// f32 products of a few hundred rows each (the single-product path would take gemm_tn_small_f32_kernel for every one of them)
static bool tn_small_group_ok(const TnProblem* pr, int n, int dtype) {
    if (dtype != MURCL_DTYPE_F32 || n < 1 || n > 4) return false;
    bool flagged = false;
    for (int g = 0; g < n; ++g) flagged |= pr[g].flags != 0;
    if (n == 1 && !flagged) return false;                      // a single plain product: murcl_gemm_tn's own dispatch
    for (int g = 0; g < n; ++g) {
        const TnProblem& p = pr[g];
        if ((p.flags & ~MURCL_TN_OVERWRITE) || p.colsum_part || p.M <= 0 || p.M > TS_MAXM || p.N1 < 4 || p.N2 < 4 || p.N1 % 4 || p.N2 % 4 || p.lda % 4 || p.ldb % 4)
            return false;
    }
    return true;
}
extern "C" int murcl_gemm_tn_grouped(const TnProblem* pr, int n, int dtype, float* ws, long ws_bytes, hipStream_t stream) {
    if (n <= 0) return 0;
    if (tn_small_group_ok(pr, n, dtype)) {
        TnSmallGroup ga;
        int tiles = 0, mmax = 0;
        for (int g = 0; g < 4; ++g) {
            const TnProblem& p = pr[g < n ? g : 0];
            ga.A[g] = (const float*)p.A; ga.B[g] = (const float*)p.B; ga.C[g] = p.C; ga.cs[g] = p.colsum_out;
            ga.M[g] = p.M; ga.N1[g] = p.N1; ga.N2[g] = p.N2; ga.lda[g] = p.lda; ga.ldb[g] = p.ldb; ga.ldc[g] = p.ldc;
            ga.overwrite[g] = (p.flags & MURCL_TN_OVERWRITE) ? 1 : 0;
            ga.tile0[g] = tiles;
            if (g < n) {
                tiles += ((p.N1 + TS_T - 1) / TS_T) * ((p.N2 + TS_T - 1) / TS_T);
                if (p.M > mmax) mmax = p.M;
            }
        }
        for (int g = n; g < 5; ++g) ga.tile0[g] = tiles;                  // unused slots: empty ranges behind the last product
        ga.pass_rows = tn_small_pass_rows(mmax, tiles);
        const int mp = ((mmax < ga.pass_rows ? mmax : ga.pass_rows) + 15) & ~15;
        static MurclOncePerDevice once;
        if (once.first())
            hipFuncSetAttribute((const void*)gemm_tn_small_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TS_MAXM * 128);
        hipLaunchKernelGGL(gemm_tn_small_group_kernel, dim3(tiles), dim3(256), 2 * mp * 128, stream, ga);
        return MURCL_CHECK_LAUNCH();
    }
    const long need = (n <= TN_MAXG) ? murcl_gemm_tn_grouped_workspace_bytes(pr, n, dtype) : 0;
    if (!need || !ws || ws_bytes < need) {
        for (int g = 0; g < n; ++g)
            if (pr[g].flags) return -1;                 // overwrite / de-interleave / scale live in the grouped reduce launch only
        for (int g = 0; g < n; ++g) {
            const TnProblem& p = pr[g];
            const int rc = murcl_gemm_tn_ws(p.A, p.B, p.C, p.M, p.N1, p.N2, p.lda, p.ldb, p.ldc, dtype, 0, p.colsum_out, ws, ws_bytes,
                                            p.colsum_part, p.colsum_rows, stream);
            if (rc) return rc;
        }
        return 0;
    }
    TnProblem q[TN_MAXG];
    for (int g = 0; g < n; ++g) {
        q[g] = pr[g];
        if (q[g].flags && ((q[g].colsum_out && !q[g].colsum_part) || ((q[g].flags & MURCL_TN_DEINTERLEAVE) && q[g].N1 % 32))) return -1;
        if (q[g].colsum_out && !q[g].colsum_part) {
            const int rc = murcl_colsum(q[g].A, q[g].colsum_out, q[g].M, q[g].N1, q[g].lda, dtype, 1, stream);
            if (rc) return rc;
            q[g].colsum_out = nullptr;
        }
    }
    return tn_sq_launch(q, n, ws, stream);
}
