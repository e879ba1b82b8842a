// One GRU time step as ONE launch each way (models/rlmil.py:14-35 Full_layer.rnn, :44-47 ActorCritic.gru; torch.nn.GRU gate
// order r, z, n).
//
// A recurrence of T steps over a few hundred rows is a chain of 2T short launches forward (the [B,3H] = h W_hh^T product,
// then the gate kernel) and 2T back (gate backward, then dh_{t-1} += dgh_t W_hh): microseconds each, ~4.5 us of which is
// the launch itself.  Every gate of hidden unit j needs rows j, H+j, 2H+j of the product and nothing else, so a workgroup
// that owns a 16-row x 16-unit tile of ALL THREE gate blocks can finish the step in its epilogue:
//
//   gru_step_fwd_kernel   (x W_ih^T, optional) + h W_hh^T for the tile's 3 x 16 weight rows, then r, z, n, h' in registers
//   gru_step_bwd_kernel   dh_{t-1}[tile] += dgh_t W_hh (K = 3H), then the gate backward of step t-1 on the finished tile
//
// Operands are f32 and the products run on the exact-f32 MFMA (16x16x4), like the library's small-GEMM path they replace.
// Data path: LDS-DMA, one instruction per 1 KiB row piece (256 k), rows at a pitch of 1024 + 16 bytes (conflict-free
// 16-byte fragment reads); the four waves split the k range of every chunk and meet through LDS before the epilogue.
#include "common.h"
#include "../../include/murcl_amd.h"

constexpr int GS_K = 256;                 // k per chunk
constexpr int GS_ROW = GS_K * 4 + 16;     // LDS row pitch, bytes
constexpr int GS_T = 16;                  // tile: 16 batch rows x 16 hidden units

template <int NB>                         // 16-row groups of the second operand per chunk (3: the gate rows of W; 1: W_hh^T)
struct GruLds {
    static constexpr int ROWS = GS_T * (1 + NB), RPW = ROWS / 4, SLOT = ROWS * GS_ROW, SLOTS = NB == 3 ? 2 : 4, BYTES = SLOT * SLOTS;
};

template <int N> __device__ __forceinline__ void gs_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int RPW> __device__ __forceinline__ void gs_wait_chunks(int ahead) {     // `ahead` younger chunks may stay in flight
    if (ahead <= 0) gs_wait<0>();
    else if (ahead == 1) gs_wait<RPW>();
    else if (ahead == 2) gs_wait<2 * RPW>();
    else gs_wait<3 * RPW>();
}

// a wave-uniform address the compiler cannot prove uniform (selected by the wave index) -> scalar registers
__device__ __forceinline__ const float* gs_uniform(const float* p) {
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const float*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}

// chunk image: rows 0..15 = A rows m0.. (clamped to M-1: their results are never stored), then NB groups of 16 rows of Bm:
// group g holds rows g*gstride + j0 .. +15.  Wave w copies image rows w*RPW .. +RPW-1.
template <int NB>
__device__ __forceinline__ void gs_issue(const float* __restrict__ A, int lda, int M, int m0, const float* __restrict__ Bm, int ldb,
                                         int gstride, int j0, int K, int k0, unsigned dst, int wave, int lane) {
    constexpr int RPW = GruLds<NB>::RPW;
    const unsigned voff = (unsigned)min(lane * 16, (K - k0) * 4 - 16);   // partial last chunk: lanes past K re-read the tail
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int i = wave * RPW + j;
        const float* src;
        if (i < GS_T) {
            src = A + (size_t)min(m0 + i, M - 1) * lda + k0;
        } else {
            const int g = (i - GS_T) >> 4, r = (i - GS_T) & 15;
            src = Bm + ((size_t)g * gstride + j0 + r) * ldb + k0;
        }
        glds16_u(gs_uniform(src), voff, dst + i * GS_ROW);
    }
}

__device__ __forceinline__ f32x4 gs_mma(f32x4 a, f32x4 b, f32x4 c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
    return c;
}

// One operand pair streamed through the slot ring: acc[g] += A[16 x K] . Bm_g[16 x K]^T, this wave's quarter of every chunk.
// Chunk c lands in slot c % SLOTS.  Entered and left with every wave past a barrier that follows its last LDS read, and nothing in
// flight - so two streams (input side, recurrent side) can run one after the other through the same slots.
template <int NB, int NACC>
__device__ __forceinline__ void gs_stream(const float* __restrict__ A, int lda, int M, int m0, const float* __restrict__ Bm, int ldb,
                                          int gstride, int j0, int K, char* smem, int wave, int lane, f32x4 (&acc)[NACC]) {
    typedef GruLds<NB> L;
    const unsigned lds0 = lds_off(smem);
    const int r16 = lane & 15, q4 = lane >> 4;
    const int n = (K + GS_K - 1) / GS_K;
    int issued = 0;
    for (; issued < n && issued < L::SLOTS; ++issued)
        gs_issue<NB>(A, lda, M, m0, Bm, ldb, gstride, j0, K, issued * GS_K, lds0 + issued * L::SLOT, wave, lane);
    for (int c = 0; c < n; ++c) {
        gs_wait_chunks<L::RPW>(issued - c - 1);
        LDS_BARRIER();                                        // every wave's rows of chunk c have landed
        const char* ab = smem + (c % L::SLOTS) * L::SLOT + r16 * GS_ROW + 16 * q4;
        const int ku = min(GS_K, K - c * GS_K) / 16;
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            const int u = wave * 4 + uu;
            if (u < ku) {
                const f32x4 a = *(const f32x4*)(ab + 64 * u);
                if constexpr (NB == 1) {                      // two accumulation chains: the f32 MFMA's latency exceeds its issue time
                    const f32x4 b = *(const f32x4*)(ab + GS_T * GS_ROW + 64 * u);
                    acc[uu & 1] = gs_mma(a, b, acc[uu & 1]);
                } else {
#pragma unroll
                    for (int g = 0; g < NB; ++g) {
                        const f32x4 b = *(const f32x4*)(ab + (GS_T + 16 * g) * GS_ROW + 64 * u);
                        acc[g] = gs_mma(a, b, acc[g]);
                    }
                }
            }
        }
        LDS_BARRIER();                                        // slot c % SLOTS has been read by every wave
        if (issued < n) {
            gs_issue<NB>(A, lda, M, m0, Bm, ldb, gstride, j0, K, issued * GS_K, lds0 + (issued % L::SLOTS) * L::SLOT, wave, lane);
            ++issued;
        }
    }
}

// the four waves' partial tiles -> P[wave][acc][m][n] in LDS (the chunk slots are free: gs_stream ends behind a barrier);
// the caller follows the last part with __syncthreads()
template <int NACC, int N>
__device__ __forceinline__ void gs_park(char* smem, int wave, int lane, int g0, const f32x4 (&acc)[N]) {
    float* P = (float*)smem;
    const int r16 = lane & 15, q4 = lane >> 4;
#pragma unroll
    for (int g = 0; g < N; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[((wave * NACC + g0 + g) * 16 + 4 * q4 + r) * 16 + r16] = acc[g][r];
}
template <int NACC> __device__ __forceinline__ float gs_total(const char* smem, int g, int tid) {
    const float* P = (const float*)smem;
    return (P[(0 * NACC + g) * 256 + tid] + P[(1 * NACC + g) * 256 + tid]) + (P[(2 * NACC + g) * 256 + tid] + P[(3 * NACC + g) * 256 + tid]);
}

// ---------------------------------------------------------------- forward
// gi: [B,3H] pre-activations of the input side INCLUDING b_ih (gi_ld = 3H), or with WITH_X the bias row b_ih itself
// (gi_ld = 0) and the kernel forms x W_ih^T as well.  gh (optional) receives h W_hh^T + b_hh (the backward reads its n block).
template <bool WITH_X>
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(const float* __restrict__ x, const float* __restrict__ Wih, int Kx,
                                                           const float* __restrict__ gi, int gi_ld, const float* __restrict__ hprev,
                                                           const float* __restrict__ Whh, const float* __restrict__ bhh,
                                                           float* __restrict__ hnew, float* __restrict__ gates, float* __restrict__ gh,
                                                           int B, int H) {
    extern __shared__ __attribute__((aligned(16))) char gs_smem[];
    constexpr int NACC = WITH_X ? 6 : 3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j0 = blockIdx.x * GS_T, m0 = blockIdx.y * GS_T;
    // the epilogue's own operands, requested before the streams so that their latency hides behind them
    const int b = m0 + (tid >> 4), j = j0 + (tid & 15);
    const bool live = b < B;
    float g_in[3], b_h[3], hp = 0.f;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        g_in[g] = live ? gi[(size_t)b * gi_ld + g * H + j] : 0.f;
        b_h[g] = bhh[g * H + j];
    }
    if (live && hprev) hp = hprev[(size_t)b * H + j];
    f32x4 ah[3], ax[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) ah[g] = ax[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (WITH_X) gs_stream<3, 3>(x, Kx, B, m0, Wih, Kx, H, j0, Kx, gs_smem, wave, lane, ax);
    if (hprev) gs_stream<3, 3>(hprev, H, B, m0, Whh, H, H, j0, H, gs_smem, wave, lane, ah);    // a zero state: h W_hh^T = 0, gh = b_hh
#pragma unroll
    for (int g = 0; g < 3; ++g) asm volatile("" : "+v"(g_in[g]), "+v"(b_h[g]));       // (early loads: first touched behind the streams)
    asm volatile("" : "+v"(hp));
    gs_park<NACC, 3>(gs_smem, wave, lane, 0, ah);
    if constexpr (WITH_X) gs_park<NACC, 3>(gs_smem, wave, lane, 3, ax);
    __syncthreads();
    if (!live) return;
    float s_h[3], s_i[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        s_h[g] = gs_total<NACC>(gs_smem, g, tid) + b_h[g];
        s_i[g] = WITH_X ? gs_total<NACC>(gs_smem, 3 + g, tid) + g_in[g] : g_in[g];
    }
    const float r = 1.f / (1.f + expf(-(s_i[0] + s_h[0])));
    const float z = 1.f / (1.f + expf(-(s_i[1] + s_h[1])));
    const float nn = tanhf(s_i[2] + r * s_h[2]);
    hnew[(size_t)b * H + j] = (1.f - z) * nn + z * hp;
    if (gates) {
        float* gt = gates + (size_t)b * 3 * H + j;
        gt[0] = r; gt[H] = z; gt[2 * H] = nn;
    }
    if (gh) {
        float* o = gh + (size_t)b * 3 * H + j;
        o[0] = s_h[0]; o[H] = s_h[1]; o[2 * H] = s_h[2];
    }
}

// ---------------------------------------------------------------- backward
// dh [B,H] holds what reached h of THIS step from above (the loss, plus the direct path dh_next * z_next already added by the
// later step); the kernel adds dgh_next . W_hh (whh_t = W_hh^T [H,3H]), writes the total back, and runs the gate backward of
// this step: dgi, dgh [B,3H], and dh * z into dhprev (added when accumulate != 0; skipped when NULL).
__global__ __launch_bounds__(256) void gru_step_bwd_kernel(const float* __restrict__ dgh_next, const float* __restrict__ whh_t,
                                                           float* __restrict__ dh, const float* __restrict__ gates,
                                                           const float* __restrict__ gh, const float* __restrict__ hprev,
                                                           float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ dhprev,
                                                           int B, int H, int gh_bcast, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) char gs_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j0 = blockIdx.x * GS_T, m0 = blockIdx.y * GS_T;
    const int b = m0 + (tid >> 4), j = j0 + (tid & 15);
    const bool live = b < B;
    float d0 = 0.f, r = 0.f, z = 0.f, nn = 0.f, ghn = 0.f, hp = 0.f, dp = 0.f;
    if (live) {
        const size_t ih = (size_t)b * H + j;
        const float* g = gates + (size_t)b * 3 * H + j;
        d0 = dh[ih];
        r = g[0]; z = g[H]; nn = g[2 * H];
        ghn = gh[(gh_bcast ? 0 : (size_t)b * 3 * H) + 2 * H + j];
        if (hprev) hp = hprev[ih];
        if (dhprev && accumulate) dp = dhprev[ih];
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    gs_stream<1, 2>(dgh_next, 3 * H, B, m0, whh_t, 3 * H, 0, j0, 3 * H, gs_smem, wave, lane, acc);
    asm volatile("" : "+v"(d0), "+v"(r), "+v"(z), "+v"(nn), "+v"(ghn), "+v"(hp), "+v"(dp));   // (early loads: first touched behind the stream)
    gs_park<2, 2>(gs_smem, wave, lane, 0, acc);
    __syncthreads();
    if (!live) return;
    const size_t ih = (size_t)b * H + j;
    const float d = d0 + (gs_total<2>(gs_smem, 0, tid) + gs_total<2>(gs_smem, 1, tid));
    dh[ih] = d;
    const float dn = d * (1.f - z) * (1.f - nn * nn);
    const float dz = d * (hp - nn) * z * (1.f - z);
    const float dr = dn * ghn * r * (1.f - r);
    float* a = dgi + (size_t)b * 3 * H + j;
    float* c = dgh + (size_t)b * 3 * H + j;
    a[0] = dr; a[H] = dz; a[2 * H] = dn;
    c[0] = dr; c[H] = dz; c[2 * H] = dn * r;
    if (dhprev) dhprev[ih] = dp + d * z;
}

// ---------------------------------------------------------------- the same stream as a plain product
// C[M,N] = epi(A[M,K] . B[N,K]^T) on 16 x 16 tiles, K split over the four waves: few outputs with a long reduction (the library's
// 32 x 32-tile form would either walk its 256-k chunks one memory round trip after the other or split K over workgroups that
// meet through a zero-fill launch, float atomics and, for a ReLU, a third launch).  Single writer per element: bias, ReLU,
// ReLU' mask and accumulation in the epilogue, the result independent of any atomic order.
__global__ __launch_bounds__(256) void gemm_nt_t16_f32_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                              float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc,
                                                              const float* __restrict__ bias, int relu, int accumulate,
                                                              const float* __restrict__ mask, int ldmask) {
    extern __shared__ __attribute__((aligned(16))) char gs_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j0 = blockIdx.x * GS_T, m0 = blockIdx.y * GS_T;
    const int m = m0 + (tid >> 4), n = j0 + (tid & 15);
    const bool live = m < M;
    float bv = 0.f, c0 = 0.f, mk = 1.f;
    if (live) {
        if (bias) bv = bias[n];
        if (accumulate) c0 = C[(size_t)m * ldc + n];
        if (mask) mk = mask[(size_t)m * ldmask + n];
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    gs_stream<1, 2>(A, lda, M, m0, Bm, ldb, 0, j0, K, gs_smem, wave, lane, acc);
    asm volatile("" : "+v"(bv), "+v"(c0), "+v"(mk));         // nothing derived from the early loads may be computed (= waited for) before here
    gs_park<2, 2>(gs_smem, wave, lane, 0, acc);
    __syncthreads();
    if (!live) return;
    float v = (gs_total<2>(gs_smem, 0, tid) + gs_total<2>(gs_smem, 1, tid)) + bv;
    if (mask) v = mk > 0.f ? v : 0.f;
    if (relu) v = fmaxf(v, 0.f);
    C[(size_t)m * ldc + n] = v + c0;                         // accumulation adds the finished epilogue (c0 = 0 without it)
}

// host side of the above for gemm.hip's small-matrix dispatcher (C++ linkage: not part of the C-ABI)
bool murcl_nt_t16_ok(int M, int N, int K) { return M > 0 && N >= 16 && N % 16 == 0 && K >= 16 && K % 16 == 0 && (M + GS_T - 1) / GS_T <= 65535; }
int murcl_nt_t16_launch(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, const float* bias,
                        int relu, int accumulate, const float* mask, int ldmask, hipStream_t stream) {
    static MurclOncePerDevice once;
    if (once.first())
        hipFuncSetAttribute((const void*)gemm_nt_t16_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GruLds<1>::BYTES);
    hipLaunchKernelGGL(gemm_nt_t16_f32_kernel, dim3(N / GS_T, (M + GS_T - 1) / GS_T), dim3(256), GruLds<1>::BYTES, stream, A, B, C, M, N, K,
                       lda, ldb, ldc, bias, relu, accumulate, mask, ldmask);
    return MURCL_CHECK_LAUNCH();
}

static int gs_shape_ok(int B, int H, int Kx) {
    // 16-unit tiles; whole 16-k groups per wave (k ranges that are multiples of 16); the LDS-DMA's 16-byte pieces
    return B > 0 && H >= 16 && H % 16 == 0 && Kx % 16 == 0 && (long)((B + GS_T - 1) / GS_T) <= 65535;
}

// C-ABI: see include/murcl_amd.h
extern "C" int murcl_gru_step_supported(int B, int H, int Kx) { return gs_shape_ok(B, H, Kx > 0 ? Kx : 16) && (Kx <= 0 || Kx >= 16); }

extern "C" int murcl_gru_step_fwd(const float* x, const float* w_ih, int Kx, const float* gi, const float* hprev,
                                  const float* w_hh, const float* b_hh, float* hnew, float* gates, float* gh, int B, int H,
                                  hipStream_t stream) {
    if (B <= 0) return 0;
    if (!b_hh || !gi || !hnew || (hprev && !w_hh) || (!hprev && !x)) return -1;     // no state and no input product: murcl_gru_gates_fwd
    if (!gs_shape_ok(B, H, x ? Kx : 16) || (x && (!w_ih || Kx < 16))) return -1;
    const dim3 grid(H / GS_T, (B + GS_T - 1) / GS_T);
    static MurclOncePerDevice once;
    if (once.first()) {
        hipFuncSetAttribute((const void*)gru_step_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GruLds<3>::BYTES);
        hipFuncSetAttribute((const void*)gru_step_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, GruLds<3>::BYTES);
    }
    if (x)
        hipLaunchKernelGGL(gru_step_fwd_kernel<true>, grid, dim3(256), GruLds<3>::BYTES, stream, x, w_ih, Kx, gi, 0, hprev, w_hh, b_hh, hnew,
                           gates, gh, B, H);
    else
        hipLaunchKernelGGL(gru_step_fwd_kernel<false>, grid, dim3(256), GruLds<3>::BYTES, stream, nullptr, nullptr, 0, gi, 3 * H, hprev, w_hh,
                           b_hh, hnew, gates, gh, B, H);
    return MURCL_CHECK_LAUNCH();
}

extern "C" int murcl_gru_step_bwd(const float* dgh_next, const float* w_hh_t, float* dh, const float* gates, const float* gh,
                                  const float* hprev, float* dgi, float* dgh, float* dhprev, int B, int H, int gh_bcast,
                                  int accumulate, hipStream_t stream) {
    if (B <= 0) return 0;
    if (!dgh_next || !w_hh_t || !dh || !gates || !gh || !dgi || !dgh) return -1;
    if (!gs_shape_ok(B, H, 16)) return -1;
    const dim3 grid(H / GS_T, (B + GS_T - 1) / GS_T);
    static MurclOncePerDevice once;
    if (once.first())
        hipFuncSetAttribute((const void*)gru_step_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GruLds<1>::BYTES);
    hipLaunchKernelGGL(gru_step_bwd_kernel, grid, dim3(256), GruLds<1>::BYTES, stream, dgh_next, w_hh_t, dh, gates, gh, hprev, dgi, dgh,
                       dhprev, B, H, gh_bcast, accumulate);
    return MURCL_CHECK_LAUNCH();
}
