// K4 / K5: CLAM-SB pieces around the shared GEMM kernels (models/clam.py:55-60,103-132,139-170).
//
//   U = h [Wa;Wb]^T + [ba;bb]                              gemm_nt (one pass over h for both gate branches)
//   s_n = sum_d tanh(U[n,d]) * sigmoid(U[n,D+d]) * wc[d] + bc      gated_score_kernel
//   A = softmax_N(s),  M = A.h                             softmax_rows_kernel + weighted_rowsum (dsmil.hip)
//   instance eval: top-k / bottom-k ids of A (lowest index wins ties), CE on 2-way instance logits
#include "common.h"

// tanh / sigmoid of the gate pre-activations.  f32 (parity path): ocml tanhf / expf.  bf16 storage: U itself carries 8
// significant bits, so the hardware v_exp_f32 / v_rcp_f32 forms (1 ulp each) are exact to storage precision - and with
// ~35 VALU operations per ocml call the streaming kernels below were VALU-bound (115 us for a 268 MB pass), not HBM-bound.
template <typename T> __device__ __forceinline__ float gs_tanh(float x) { return tanhf(x); }
template <> __device__ __forceinline__ float gs_tanh<bf16_t>(float x) { return fast_tanh(x); }
template <typename T> __device__ __forceinline__ float gs_sigmoid(float x) { return sigmoidf_(x); }
template <> __device__ __forceinline__ float gs_sigmoid<bf16_t>(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// ---------------------------------------------------------------- gated attention score
// Streaming kernels over U [rows, 2D]: a thread owns 8 consecutive columns of both gate halves (16-byte loads for
// bf16), G = D/8 column groups, 256/G rows in flight per workgroup pass, a workgroup walks `rows_per_block` rows.
// GATED = false: the plain attention net (clam.py:18-34): U has D columns, s_n = sum_d tanh(U[n,d]) wc[d] + bc.
// Gate dropout without materialised masks: 8 consecutive keep values of row n, columns 8cg.. of a [rows, D] mask = the 8 bytes of
// one counter-based word (murcl_dropout_mask's generator, flat index n*D + 8cg).
#ifndef GS_GROUP_SUM
#define GS_GROUP_SUM group_sum
#endif
struct GsDrop { unsigned long long seed_a, seed_b; unsigned thresh; float scale; };
__device__ __forceinline__ void gs_keep8(unsigned long long seed, long flat8, unsigned thresh, float scale, float* k) {
    const unsigned long long rw = murcl_drop_word(seed, flat8);
#pragma unroll
    for (int e = 0; e < 8; ++e) k[e] = ((unsigned)(rw >> (8 * e)) & 255u) < thresh ? scale : 0.f;
}
#ifndef GS_UR_BF16
#define GS_UR_BF16 4
#endif
template <typename T, bool GATED>
__global__ __launch_bounds__(256) void gated_score_fwd_kernel(const T* __restrict__ U, const float* __restrict__ wc,
                                                              const float* __restrict__ bc,
                                                              const T* __restrict__ keep_a, const T* __restrict__ keep_b,
                                                              float* __restrict__ s, long rows, int D, int rows_per_block, GsDrop drop) {
    __shared__ float red[256];
    const int tid = threadIdx.x, G = D >> 3, RL = 256 / G;
    const int cg = tid % G, rl = tid / G;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = (rl < RL) ? wc[8 * cg + e] : 0.f;
    const float b0 = bc[0];
    const bool pow2 = G <= 64 && (G & (G - 1)) == 0;
    constexpr int UR = sizeof(T) == 2 ? GS_UR_BF16 : 4;   // rows in flight per thread
    for (long base = r0; base < r1; base += UR * RL) {
        float ua[UR][8], ub[UR][8], acc[UR];
        bool live[UR];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const long n = base + u * RL + rl;
            live[u] = rl < RL && n < r1;
            acc[u] = 0.f;
            if (live[u]) {
                load8<T>(U + n * (GATED ? 2 : 1) * D + 8 * cg, ua[u]);
                if (GATED) load8<T>(U + n * 2 * D + D + 8 * cg, ub[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const long n = base + u * RL + rl;
            if (live[u]) {
                if (!GATED) {
                    if (keep_a || drop.thresh) {
                        float ka[8];
                        if (keep_a) load8<T>(keep_a + n * D + 8 * cg, ka);
                        else gs_keep8(drop.seed_a, (n * D + 8 * cg) >> 3, drop.thresh, drop.scale, ka);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[u] += gs_tanh<T>(ua[u][e]) * ka[e] * w[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[u] += gs_tanh<T>(ua[u][e]) * w[e];
                    }
                } else if (keep_a || drop.thresh) {
                    float ka[8], kb[8];
                    if (keep_a) {
                        load8<T>(keep_a + n * D + 8 * cg, ka);
                        load8<T>(keep_b + n * D + 8 * cg, kb);
                    } else {
                        gs_keep8(drop.seed_a, (n * D + 8 * cg) >> 3, drop.thresh, drop.scale, ka);
                        gs_keep8(drop.seed_b, (n * D + 8 * cg) >> 3, drop.thresh, drop.scale, kb);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[u] += (gs_tanh<T>(ua[u][e]) * ka[e]) * (gs_sigmoid<T>(ub[u][e]) * kb[e]) * w[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[u] += gs_tanh<T>(ua[u][e]) * gs_sigmoid<T>(ub[u][e]) * w[e];
                }
            }
        }
        // sum over the G threads of a row: lanes of one wave when G is a power of two <= 64, else through LDS
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const long n = base + u * RL + rl;
            float t = acc[u];
            if (pow2) {
                t = GS_GROUP_SUM(t, G);
            } else {
                red[tid] = t;
                __syncthreads();
                if (cg == 0 && rl < RL) { t = 0.f; for (int k = 0; k < G; ++k) t += red[rl * G + k]; }
                __syncthreads();
            }
            if (cg == 0 && live[u]) s[n] = t + b0;
        }
    }
}
// dU[n,d] = ds_n wc_d g (1-a^2) ka kb ; dU[n,D+d] = ds_n wc_d a g (1-g) ka kb ; dwc_d += ds_n a g ka kb ; dbc += ds_n
// and, from the same pass, the column sums of dU (the bias gradients of the two gate Linears: a separate column-sum
// launch re-read all of dU, 54 us at the C3 shape)
// IL: U / dU in the interleaved column order of the panel GEMM's PG_GATE_U epilogue (16 a-columns, then the 16 b-columns of the
//     same d, per 32-column block); dwc / the column sums stay in natural order.
// HC > 0: ds is not an input - the pooling and soft-max backward (clam.py:144,170) are taken in this pass:
//     ds_n = A_n (h_n . dM_bag - M_bag . dM_bag)     (sum_m A_m (h_m . dM) = M . dM: no reduction over the bag is needed)
//     with h [rows, L], L = 8 HC G: a thread takes HC 16-byte pieces of the row, the G threads of a row add up by lane shuffles
//     (G a power of two <= 64), and a workgroup's rows lie in one bag (rows_per_bag % rows_per_block == 0).
#ifndef GSB_UR
#define GSB_UR (HC > 0 ? 2 : 4)      // rows in flight per thread (measured r03: one-pass form 161 / 171 us at 2 / 4, ds-given form 119 / 111)
#endif
template <typename T, bool GATED, bool IL, int HC>
__global__ __launch_bounds__(256) void gated_score_bwd_kernel(const T* __restrict__ U, const float* __restrict__ wc,
                                                              const T* __restrict__ keep_a, const T* __restrict__ keep_b,
                                                              const float* __restrict__ ds, T* __restrict__ dU,
                                                              float* __restrict__ part,
                                                              long rows, int D, int rows_per_block, GsDrop drop,
                                                              const T* __restrict__ h, const float* __restrict__ dM,
                                                              const float* __restrict__ Mp, const float* __restrict__ Asm,
                                                              int rows_per_bag) {
    __shared__ float red[256][25];
    const int tid = threadIdx.x, G = D >> 3, RL = 256 / G;
    const int cg = tid % G, rl = tid / G;
    const int W = (GATED ? 2 : 1) * D;
    const int off_a = IL ? 32 * (cg >> 1) + 8 * (cg & 1) : 8 * cg;
    const int off_b = IL ? off_a + 16 : D + 8 * cg;
    float w[8], wacc[8], csa[8], csb[8], dbc_acc = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { w[e] = (rl < RL) ? wc[8 * cg + e] : 0.f; wacc[e] = csa[e] = csb[e] = 0.f; }
    const int L = 8 * HC * G;
    // a workgroup walks row segments seg, seg + grid, ...: the grid is capped at the rows of `part`, so calls with more bags
    // than that (stage 1 of the contrastive step: 2 T B = 1536 sub-bags) give a workgroup several segments, each inside one bag
    const long nseg = (rows + rows_per_block - 1) / rows_per_block;
    for (long seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const long r0 = seg * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float dm[HC > 0 ? 8 * HC : 1], mdm = 0.f;
    if (HC > 0 && rl < RL) {
        const long bag = r0 / rows_per_bag;
#pragma unroll
        for (int c = 0; c < HC; ++c) {
            float mp[8];
            load8<float>(dM + bag * L + 8 * (c * G + cg), dm + 8 * c);
            load8<float>(Mp + bag * L + 8 * (c * G + cg), mp);
#pragma unroll
            for (int e = 0; e < 8; ++e) mdm += mp[e] * dm[8 * c + e];
        }
        mdm = GS_GROUP_SUM(mdm, G);
    }
    if (rl < RL) {
        constexpr int UR = GSB_UR;                        // rows in flight per thread
        for (long n0 = r0 + rl; n0 < r1; n0 += UR * RL) {
            float ua[UR][8], ub[UR][8], dsn[UR];
            float hv[HC > 0 ? UR : 1][HC > 0 ? 8 * HC : 1], an[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const long n = n0 + u * RL;
                dsn[u] = an[u] = 0.f;
                if (n < r1) {
                    load8<T>(U + n * W + off_a, ua[u]);
                    if (GATED) load8<T>(U + n * W + off_b, ub[u]);
                    if (HC == 0) dsn[u] = ds[n];
                    if (HC > 0) {
                        an[u] = Asm[n];
#pragma unroll
                        for (int c = 0; c < HC; ++c) load8<T>(h + n * L + 8 * (c * G + cg), hv[u] + 8 * c);
                    }
                }
            }
            if (HC > 0) {
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    const long n = n0 + u * RL;          // (n < r1 is uniform over the G lanes of a row: the shuffles below are safe)
                    float t = 0.f;
                    if (n < r1) {
#pragma unroll
                        for (int e = 0; e < 8 * HC; ++e) t += hv[u][e] * dm[e];
                    }
                    t = GS_GROUP_SUM(t, G);
                    dsn[u] = an[u] * (t - mdm);
                }
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const long n = n0 + u * RL;
                if (n >= r1) continue;
                float da[8], db[8], k[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) k[e] = 1.f;
                if (keep_a || drop.thresh) {
                    float ka[8], kb[8];
                    if (keep_a) {
                        load8<T>(keep_a + n * D + 8 * cg, ka);
                        if (GATED) load8<T>(keep_b + n * D + 8 * cg, kb);
                    } else {
                        gs_keep8(drop.seed_a, (n * D + 8 * cg) >> 3, drop.thresh, drop.scale, ka);
                        if (GATED) gs_keep8(drop.seed_b, (n * D + 8 * cg) >> 3, drop.thresh, drop.scale, kb);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) k[e] = GATED ? ka[e] * kb[e] : ka[e];
                }
                if (cg == 0) dbc_acc += dsn[u];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = gs_tanh<T>(ua[u][e]), g = GATED ? gs_sigmoid<T>(ub[u][e]) : 1.f;
                    da[e] = dsn[u] * w[e] * g * (1.f - a * a) * k[e];
                    db[e] = dsn[u] * w[e] * a * g * (1.f - g) * k[e];
                    wacc[e] += dsn[u] * a * g * k[e];
                    csa[e] += da[e];
                    csb[e] += db[e];
                }
                store8<T>(dU + n * W + off_a, da);
                if (GATED) store8<T>(dU + n * W + off_b, db);
            }
        }
    }
    }   // segments
    // reduce over the row lanes, then this workgroup's row of partial sums (part [grid][3D+1] = dwc | dbc | colsum of
    // dU; summed by gated_score_reduce_kernel - thousands of atomic adders on D addresses would serialise at the memory side)
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid][e] = wacc[e]; red[tid][9 + e] = csa[e]; red[tid][17 + e] = csb[e]; }
    red[tid][8] = dbc_acc;
    __syncthreads();
    if (rl == 0) {
        float t[25];
#pragma unroll
        for (int e = 0; e < 25; ++e) t[e] = 0.f;
        for (int r = 0; r < RL; ++r)
#pragma unroll
            for (int e = 0; e < 25; ++e) t[e] += red[r * G + cg][e];
        float* prow = part + (size_t)blockIdx.x * (3 * D + 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            prow[8 * cg + e] = t[e];
            prow[D + 1 + 8 * cg + e] = t[9 + e];
            prow[2 * D + 1 + 8 * cg + e] = t[17 + e];
        }
        if (cg == 0) prow[D] = t[8];
    }
}
// dwc[c] = sum_w part[w][c], dbc = sum_w part[w][D], dbab[c'] = sum_w part[w][D+1+c']   (16 columns x 16 row lanes per
// workgroup; dbab may be NULL)
__global__ __launch_bounds__(256) void gated_score_reduce_kernel(const float* __restrict__ part, int n_wg, int D,
                                                                 float* __restrict__ dwc, float* __restrict__ dbc,
                                                                 float* __restrict__ dbab) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const int W = 3 * D + 1;
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
        if (c < D) dwc[c] = t;
        else if (c == D) dbc[0] = t;
        else if (dbab) dbab[c - D - 1] = t;
    }
}
#ifndef GS_WGS
#define GS_WGS 4096
#endif
static bool gs_shape_ok(int D) { return D >= 8 && D <= 2048 && D % 8 == 0; }
static int gs_rows_per_block(long rows, int D) {
    const int RL = 256 / (D / 8) > 0 ? 256 / (D / 8) : 1;
    long rpb = (rows + GS_WGS - 1) / GS_WGS;               // ~GS_WGS workgroups
    rpb = ((rpb + RL - 1) / RL) * RL;
    if (rpb < 4 * RL) rpb = 4 * RL;
    return (int)rpb;
}
static GsDrop gs_drop(float keep_p, unsigned long long seed_a, unsigned long long seed_b) {
    GsDrop d{seed_a, seed_b, 0u, 1.f};
    if (keep_p > 0.f && keep_p < 1.f) { d.thresh = (unsigned)(keep_p * 256.f + 0.5f); d.scale = 256.f / (float)(d.thresh ? d.thresh : 1u); }   // 1 / the REALISED keep probability
    return d;
}
extern "C" int murcl_gated_score_fwd(const void* U, const float* wc, const float* bc, const void* keep_a,
                                     const void* keep_b, float* s, long rows, int D, int dtype, int gated, float keep_p,
                                     unsigned long long seed_a, unsigned long long seed_b, hipStream_t st) {
    if (rows <= 0) return 0;
    if (!gs_shape_ok(D)) return -1;
    const GsDrop drop = gs_drop(keep_a ? 0.f : keep_p, seed_a, seed_b);
    const int rpb = gs_rows_per_block(rows, D);
    dim3 grid((unsigned)((rows + rpb - 1) / rpb));
#define GS_FWD(T, G) hipLaunchKernelGGL((gated_score_fwd_kernel<T, G>), grid, dim3(256), 0, st, (const T*)U, wc, bc, (const T*)keep_a, (const T*)keep_b, s, rows, D, rpb, drop)
    if (dtype == MURCL_DTYPE_F32) { if (gated) GS_FWD(float, true); else GS_FWD(float, false); }
    else if (dtype == MURCL_DTYPE_BF16) { if (gated) GS_FWD(bf16_t, true); else GS_FWD(bf16_t, false); }
    else return -1;
#undef GS_FWD
    return MURCL_CHECK_LAUNCH();
}
static int gs_bwd_launch(const void* U, const float* wc, const void* keep_a, const void* keep_b, const float* ds, void* dU,
                         float* dwc, float* dbc, float* dbab, float* part_ws, long rows, int D, int dtype, int gated, float keep_p,
                         unsigned long long seed_a, unsigned long long seed_b, int interleaved, const void* h, const float* dM,
                         const float* Mp, const float* Asm, int L, int rows_per_bag, hipStream_t st) {
    if (rows <= 0) return 0;
    if (!gs_shape_ok(D) || !part_ws) return -1;
    const GsDrop drop = gs_drop(keep_a ? 0.f : keep_p, seed_a, seed_b);
    const int G = D / 8, RL = 256 / G > 0 ? 256 / G : 1;
    long rpb = (rows + 1023) / 1024;                       // <= 1024 workgroups = rows of part_ws [1024][3D+1]
    rpb = ((rpb + RL - 1) / RL) * RL;
    if (rpb < 4 * RL) rpb = 4 * RL;
    int hc = 0;
    if (h) {
        // fused pooling / soft-max backward: bf16, gated, interleaved only (the CLAM-SB training chain), see the kernel
        if (!dM || !Mp || !Asm || ds || dtype != MURCL_DTYPE_BF16 || !gated || !interleaved || G > 64 || (G & (G - 1)) || L % (8 * G) ||
            rows_per_bag <= 0 || rows % rows_per_bag)
            return -1;
        hc = L / (8 * G);
        if (hc != 1 && hc != 2 && hc != 4) return -1;
        // a segment's rows lie in one bag: the smallest divisor of rows_per_bag at or above the target (rows_per_bag itself
        // at the latest, so the search ends for every shape - ADVICE r3: stepping by RL from above rows_per_bag never did)
        if (rpb >= rows_per_bag) rpb = rows_per_bag;
        else while (rows_per_bag % rpb) ++rpb;
    } else if (!ds) {
        return -1;
    }
    if (interleaved && (!gated || D % 16)) return -1;
    const long nseg = (rows + rpb - 1) / rpb;
    const int grid = (int)(nseg < 1024 ? nseg : 1024);     // part_ws has 1024 rows; a workgroup walks segments grid apart
#define GS_BWD(T, G_, IL_, HC_) hipLaunchKernelGGL((gated_score_bwd_kernel<T, G_, IL_, HC_>), dim3(grid), dim3(256), 0, st, (const T*)U, wc, (const T*)keep_a, (const T*)keep_b, ds, (T*)dU, part_ws, rows, D, (int)rpb, drop, (const T*)h, dM, Mp, Asm, rows_per_bag)
    if (hc == 1) GS_BWD(bf16_t, true, true, 1);
    else if (hc == 2) GS_BWD(bf16_t, true, true, 2);
    else if (hc == 4) GS_BWD(bf16_t, true, true, 4);
    else if (dtype == MURCL_DTYPE_F32) { if (interleaved) GS_BWD(float, true, true, 0); else if (gated) GS_BWD(float, true, false, 0); else GS_BWD(float, false, false, 0); }
    else if (dtype == MURCL_DTYPE_BF16) { if (interleaved) GS_BWD(bf16_t, true, true, 0); else if (gated) GS_BWD(bf16_t, true, false, 0); else GS_BWD(bf16_t, false, false, 0); }
    else return -1;
#undef GS_BWD
    int rc = MURCL_CHECK_LAUNCH();
    if (rc) return rc;
    hipLaunchKernelGGL(gated_score_reduce_kernel, dim3((3 * D + 1 + 15) / 16), dim3(256), 0, st, part_ws, grid, D, dwc, dbc, dbab);
    return MURCL_CHECK_LAUNCH();
}
extern "C" int murcl_gated_score_bwd(const void* U, const float* wc, const void* keep_a, const void* keep_b,
                                     const float* ds, void* dU, float* dwc, float* dbc, float* dbab, float* part_ws,
                                     long rows, int D, int dtype, int gated, float keep_p, unsigned long long seed_a,
                                     unsigned long long seed_b, hipStream_t st) {
    return gs_bwd_launch(U, wc, keep_a, keep_b, ds, dU, dwc, dbc, dbab, part_ws, rows, D, dtype, gated, keep_p, seed_a, seed_b, 0,
                         nullptr, nullptr, nullptr, nullptr, 0, 0, st);
}
// The CLAM-SB training chain's form (see gated_score_bwd_kernel): U / dU [rows, 2D] in the interleaved column order of
// murcl_panel_gemm epilogue 5; ds given, or (ds NULL) derived in the same pass from h [rows, L] (dtype of U), the pooled
// vectors Mp [bags, L], their upstream gradient dM [bags, L] and the attention A [rows] (f32).
extern "C" int murcl_gated_score_bwd_il(const void* U, const float* wc, const float* ds, void* dU, float* dwc, float* dbc,
                                        float* dbab, float* part_ws, long rows, int D, int dtype, float keep_p,
                                        unsigned long long seed_a, unsigned long long seed_b, const void* h, const float* dM,
                                        const float* Mp, const float* A, int L, int rows_per_bag, hipStream_t st) {
    return gs_bwd_launch(U, wc, nullptr, nullptr, ds, dU, dwc, dbc, dbab, part_ws, rows, D, dtype, 1, keep_p, seed_a, seed_b, 1,
                         h, dM, Mp, A, L, rows_per_bag, st);
}

// ---------------------------------------------------------------- soft-max over the N patches of each bag
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, float* __restrict__ A, int N) {
    __shared__ float red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* x = s + (size_t)b * N;
    float* y = A + (size_t)b * N;
    float mx = -INFINITY;
    for (int n = tid; n < N; n += 256) mx = fmaxf(mx, x[n]);
    red[tid] = mx; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
    mx = red[0]; __syncthreads();
    float sum = 0.f;
    for (int n = tid; n < N; n += 256) { const float e = expf(x[n] - mx); y[n] = e; sum += e; }
    red[tid] = sum; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float inv = 1.f / red[0];
    for (int n = tid; n < N; n += 256) y[n] *= inv;
}
// The same from P partial score rows part [P][B*N] (murcl_panel_gemm's gate epilogues leave one row per 32-column group; attention_c's
// bias rides in row 0): s = their column sum, A = soft-max_N(s), ONE launch and one read of every partial - the column-sum launch, its
// [B*N] round trip and two of the soft-max kernel's three passes over global memory are gone (6 + 14 us -> 8 at 64 x 4096, P = 16).
// A thread keeps its SR_PT values of a bag in registers (N <= 256 * SR_PT; longer bags re-read s).
#define SR_PT 16
__global__ __launch_bounds__(256) void softmax_rows_parts_kernel(const float* __restrict__ part, int P, long M, float* __restrict__ s,
                                                                 float* __restrict__ A, int N, float* __restrict__ zero, int zero_n) {
    __shared__ float red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < zero_n; i += 256) zero[(size_t)b * zero_n + i] = 0.f;      // the pooled row the next pass adds into
    const float* x = part + (size_t)b * N;
    float* so = s + (size_t)b * N;
    float* y = A + (size_t)b * N;
    const bool in_regs = N <= 256 * SR_PT;
    float v[SR_PT];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < SR_PT; ++t) v[t] = 0.f;
    // partial row by partial row, the thread's SR_PT columns of it in flight together (column by column the P loads of a column were
    // P dependent round trips on 64 workgroups: 115 us at 64 x 4096, P = 16)
    int p = 0;
    for (; p + 3 < P; p += 4) {                              // four partial rows x SR_PT columns = 64 loads in flight per thread
        float w[4][SR_PT];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < SR_PT; ++t) w[q][t] = (tid + 256 * t < N) ? x[(size_t)(p + q) * M + tid + 256 * t] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < SR_PT; ++t) v[t] += w[q][t];
    }
    for (; p < P; ++p) {
        const float* xp = x + (size_t)p * M;
        float w[SR_PT];
#pragma unroll
        for (int t = 0; t < SR_PT; ++t) w[t] = (tid + 256 * t < N) ? xp[tid + 256 * t] : 0.f;
#pragma unroll
        for (int t = 0; t < SR_PT; ++t) v[t] += w[t];
    }
#pragma unroll
    for (int t = 0; t < SR_PT; ++t) {
        const int n = tid + 256 * t;
        if (n < N) { so[n] = v[t]; mx = fmaxf(mx, v[t]); } else v[t] = -INFINITY;
    }
    for (int n = tid + 256 * SR_PT; n < N; n += 256) {
        float a = 0.f;
        for (int p = 0; p < P; ++p) a += x[(size_t)p * M + n];
        so[n] = a;
        mx = fmaxf(mx, a);
    }
    red[tid] = mx; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
    mx = red[0]; __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < SR_PT; ++t) {
        v[t] = (tid + 256 * t < N) ? expf(v[t] - mx) : 0.f;
        sum += v[t];
    }
    for (int n = tid + 256 * SR_PT; n < N; n += 256) { const float e = expf(so[n] - mx); y[n] = e; sum += e; }
    red[tid] = sum; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float inv = 1.f / red[0];
#pragma unroll
    for (int t = 0; t < SR_PT; ++t)
        if (tid + 256 * t < N) y[tid + 256 * t] = v[t] * inv;
    if (!in_regs)
        for (int n = tid + 256 * SR_PT; n < N; n += 256) y[n] *= inv;
}
extern "C" int murcl_softmax_rows_parts(const float* part, int P, float* s, float* A, int B, int N, float* zero, int zero_n,
                                        hipStream_t st) {
    if (B <= 0) return 0;
    if (P <= 0) return -1;
    hipLaunchKernelGGL(softmax_rows_parts_kernel, dim3(B), dim3(256), 0, st, part, P, (long)B * N, s, A, N, zero, zero ? zero_n : 0);
    return MURCL_CHECK_LAUNCH();
}
// ds = A * (dA - sum_n A dA)
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ A, const float* __restrict__ dA,
                                                               float* __restrict__ ds, int N) {
    __shared__ float red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* a = A + (size_t)b * N;
    const float* d = dA + (size_t)b * N;
    float s = 0.f;
    for (int n = tid; n < N; n += 256) s += a[n] * d[n];
    red[tid] = s; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float dot = red[0];
    for (int n = tid; n < N; n += 256) ds[(size_t)b * N + n] = a[n] * (d[n] - dot);
}
extern "C" int murcl_softmax_rows(const float* s, float* A, int B, int N, hipStream_t st) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(B), dim3(256), 0, st, s, A, N);
    return MURCL_CHECK_LAUNCH();
}
extern "C" int murcl_softmax_rows_bwd(const float* A, const float* dA, float* ds, int B, int N, hipStream_t st) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(B), dim3(256), 0, st, A, dA, ds, N);
    return MURCL_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- top-k / bottom-k patch ids per bag (clam.py:107-109,126)
// ids[b, 0:k] = indices of the k largest A[b,:] in descending order, ids[b, k:2k] = the k smallest in ascending
// order; ties go to the lowest index.  k <= 32.
// Bags of up to 256*TK_PT patches: every thread keeps its TK_PT values in registers as 64-bit keys
// (order-preserving value bits | ~index, so ties go to the lowest index) and each of the k rounds is one wave
// arg-max by shuffles + a 4-entry exchange through LDS, for the descending and the ascending selection at once.
// Larger bags fall back to rounds of a full arg-max scan.
#define TK_PT 16
__device__ __forceinline__ unsigned tk_ord(float v) {            // monotone float -> uint32
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned long long tk_wave_max(unsigned long long k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(k, o, 64);
        k = other > k ? other : k;
    }
    return k;
}
__global__ __launch_bounds__(256) void topk_ids_kernel(const float* __restrict__ A, int N, int k, int* __restrict__ ids) {
    __shared__ unsigned long long wk[2][2][4];          // [round parity][desc/asc][wave]
    __shared__ float bv[256];
    __shared__ int bi[256];
    __shared__ int taken[64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* a = A + (size_t)b * N;
    int* out = ids + (size_t)b * 2 * k;
    if (N <= 256 * TK_PT) {
        unsigned long long kd[TK_PT], ka[TK_PT];
        float av[TK_PT];
#pragma unroll
        for (int t = 0; t < TK_PT; ++t) av[t] = a[min(tid + 256 * t, N - 1)];     // all TK_PT loads in flight (a branch per element
#pragma unroll                                                                   // made them TK_PT dependent round trips: 20 us)
        for (int t = 0; t < TK_PT; ++t) {
            const int n = tid + 256 * t;
            const unsigned o = tk_ord(av[t]);
            kd[t] = n < N ? (((unsigned long long)o << 32) | (unsigned)(~n)) : 0ull;
            ka[t] = n < N ? (((unsigned long long)(~o) << 32) | (unsigned)(~n)) : 0ull;
        }
        for (int r = 0; r < k; ++r) {
            unsigned long long md = 0ull, ma = 0ull;
#pragma unroll
            for (int t = 0; t < TK_PT; ++t) { md = kd[t] > md ? kd[t] : md; ma = ka[t] > ma ? ka[t] : ma; }
            md = tk_wave_max(md);
            ma = tk_wave_max(ma);
            if (lane == 0) { wk[r & 1][0][wave] = md; wk[r & 1][1][wave] = ma; }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const unsigned long long d = wk[r & 1][0][w], c = wk[r & 1][1][w];
                md = d > md ? d : md; ma = c > ma ? c : ma;
            }
            const int id = (int)~(unsigned)md, ia = (int)~(unsigned)ma;
            if (tid == 0) { out[r] = id; out[k + r] = ia; }
            // the owners retire the winners (key 0 never wins again)
#pragma unroll
            for (int t = 0; t < TK_PT; ++t) {
                if (tid + 256 * t == id) kd[t] = 0ull;
                if (tid + 256 * t == ia) ka[t] = 0ull;
            }
        }
        return;
    }
    for (int pass = 0; pass < 2; ++pass) {
        const float sgn = pass == 0 ? 1.f : -1.f;
        for (int r = 0; r < k; ++r) {
            float best = -INFINITY;
            int idx = 0x7fffffff;
            for (int n = tid; n < N; n += 256) {
                bool used = false;
                for (int t = 0; t < r; ++t) used |= (taken[t] == n);
                if (used) continue;
                const float v = sgn * a[n];
                if (v > best || (v == best && n < idx)) { best = v; idx = n; }
            }
            bv[tid] = best; bi[tid] = idx;
            __syncthreads();
            for (int o = 128; o > 0; o >>= 1) {
                if (tid < o) {
                    const float v = bv[tid + o]; const int i = bi[tid + o];
                    if (v > bv[tid] || (v == bv[tid] && i < bi[tid])) { bv[tid] = v; bi[tid] = i; }
                }
                __syncthreads();
            }
            if (tid == 0) { taken[r] = bi[0]; out[pass * k + r] = bi[0]; }
            __syncthreads();
        }
    }
}
extern "C" int murcl_topk_ids(const float* A, int B, int N, int k, int* ids, hipStream_t st) {
    if (B <= 0) return 0;
    if (k <= 0 || k > 32 || k > N) return -1;
    hipLaunchKernelGGL(topk_ids_kernel, dim3(B), dim3(256), 0, st, A, N, k, ids);
    return MURCL_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- row gather / masked scatter-add (instance branch)
// out[r,:] = src[base[r] + ids[r], :]
template <typename T>
__global__ void take_rows_kernel(const T* __restrict__ src, const long* __restrict__ rows, float* __restrict__ out, int d) {
    const long r = blockIdx.x;
    const T* s = src + rows[r] * d;
    for (int k = threadIdx.x; k < d; k += blockDim.x) out[r * d + k] = to_f<T>(s[k]);
}
// dst[rows[r],:] += g[r,:] * (h[rows[r],:] > 0)      (rows distinct within a launch)
template <typename T>
__global__ void scatter_add_rows_masked_kernel(T* __restrict__ dst, const T* __restrict__ h, const long* __restrict__ rows,
                                               float* g, int d, int write_back) {
    const long r = blockIdx.x;
    T* o = dst + rows[r] * d;
    const T* m = h + rows[r] * d;
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        const bool on = to_f<T>(m[k]) > 0.f;
        if (on) o[k] = from_f<T>(to_f<T>(o[k]) + g[r * d + k]);
        else if (write_back) g[r * d + k] = 0.f;            // g becomes what was added: its column sums extend dst's
    }
}
extern "C" int murcl_take_rows(const void* src, const long* rows, float* out, int R, int d, int dtype, hipStream_t st) {
    if (R <= 0) return 0;
    if (dtype == MURCL_DTYPE_F32) hipLaunchKernelGGL(take_rows_kernel<float>, dim3(R), dim3(256), 0, st, (const float*)src, rows, out, d);
    else if (dtype == MURCL_DTYPE_BF16) hipLaunchKernelGGL(take_rows_kernel<bf16_t>, dim3(R), dim3(256), 0, st, (const bf16_t*)src, rows, out, d);
    else return -1;
    return MURCL_CHECK_LAUNCH();
}
extern "C" int murcl_scatter_add_rows_masked(void* dst, const void* h, const long* rows, float* g, int R, int d,
                                             int dtype, int write_back, hipStream_t st) {
    if (R <= 0) return 0;
    if (dtype == MURCL_DTYPE_F32) hipLaunchKernelGGL(scatter_add_rows_masked_kernel<float>, dim3(R), dim3(256), 0, st, (float*)dst, (const float*)h, rows, g, d, write_back);
    else if (dtype == MURCL_DTYPE_BF16) hipLaunchKernelGGL(scatter_add_rows_masked_kernel<bf16_t>, dim3(R), dim3(256), 0, st, (bf16_t*)dst, (const bf16_t*)h, rows, g, d, write_back);
    else return -1;
    return MURCL_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- mean cross-entropy per group of G consecutive rows
// (C <= 32 logits per row).  One workgroup per group:  loss[g] = mean over the rows with target >= 0 of
// [ lse(logits_r) - logits_r[target_r] ] (0 for a group without such rows); dlogits = (softmax - onehot) / count for
// those rows, 0 for ignored rows (target < 0); pred = argmax (first max), -1 for ignored rows; conf = the soft-max probability of
// the target class (the confidence the RL-MIL rewards are differences of, train_RLMIL.py:345,537,735), 0 for ignored rows.
__global__ __launch_bounds__(64) void ce_fwd_bwd_kernel(const float* __restrict__ logits, const long* __restrict__ targets,
                                                        int G, int C, float* __restrict__ loss,
                                                        float* __restrict__ dlogits, long* __restrict__ preds,
                                                        float* __restrict__ conf) {
    const int grp = blockIdx.x, tid = threadIdx.x;
    float cnt = 0.f;
    for (int r = tid; r < G; r += 64) cnt += targets[(size_t)grp * G + r] >= 0 ? 1.f : 0.f;
    cnt = wave_sum(cnt);
    const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
    float acc = 0.f;
    for (int r = tid; r < G; r += 64) {
        const size_t row = (size_t)grp * G + r;
        const float* x = logits + row * C;
        const int t = (int)targets[row];
        if (t < 0) {
            if (dlogits)
                for (int c = 0; c < C; ++c) dlogits[row * C + c] = 0.f;
            if (preds) preds[row] = -1;
            if (conf) conf[row] = 0.f;
            continue;
        }
        float mx = x[0];
        int am = 0;
        for (int c = 1; c < C; ++c) if (x[c] > mx) { mx = x[c]; am = c; }
        float sum = 0.f;
        for (int c = 0; c < C; ++c) sum += expf(x[c] - mx);
        const float lse = mx + logf(sum);
        acc += lse - x[t];
        if (dlogits)
            for (int c = 0; c < C; ++c) dlogits[row * C + c] = (expf(x[c] - lse) - (c == t ? 1.f : 0.f)) * inv;
        if (preds) preds[row] = am;
        if (conf) conf[row] = expf(x[t] - lse);
    }
    acc = wave_sum(acc);
    if (tid == 0) loss[grp] = acc * inv;
}
extern "C" int murcl_cross_entropy(const float* logits, const long* targets, int R, int C, float* loss, float* dlogits,
                                   long* preds, float* conf, int group, hipStream_t st) {
    if (R <= 0 || C <= 0 || group <= 0 || R % group) return -1;
    hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3(R / group), dim3(64), 0, st, logits, targets, group, C, loss, dlogits, preds, conf);
    return MURCL_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- the instance branch as two launches (clam.py:103-132,150-168)
// One workgroup per bag.  Forward: the 2k selected rows of h (top-k then bottom-k ids of the attention, murcl_topk_ids) meet ALL
// n_cls two-way instance classifiers (W [n_cls*2, L], b [n_cls*2]) - a wave takes a row, a lane 8 columns per 512-column step, the
// 2 n_cls dot products are wave sums - then one thread per (class, row) pair takes the cross-entropy of clam.py's targets:
//   class == label: rows 0..k-1 -> 1, rows k..2k-1 -> 0 (inst_eval, :105-119);  class != label: rows 0..k-1 -> 0 if subtyping
//   (inst_eval_out, :122-132), otherwise the pair is ignored;  a class's loss = the mean over its live rows.
// Outputs: loss[b] = scale * sum_c loss_c (scale = 1/n_cls with subtyping, :167-168); dl [B*2k, 2 n_cls] = d loss[b] / d logits;
// pt [2][B][n_cls][2k] = (prediction, target), -1 where ignored.
// Backward (up[b] = d L / d loss[b]): g[r,:] = up sum_o dl[r,o] W[o,:] is ADDED to dz[row r] where h[row r] > 0 (the ReLU mask of
// the first layer; the 2k rows of a bag are distinct); part[b] = ( up dl^T feats [2 n_cls][L] | up sum_r dl[r,:] [2 n_cls] |
// sum_r of the masked g [L] ): the caller adds the B rows up (murcl_colsum) - no float atomics.
#define CI_MAXO 16          // 2 * n_cls <= 16
template <typename T>
__global__ __launch_bounds__(256) void clam_inst_fwd_kernel(const T* __restrict__ h, const int* __restrict__ ids,
                                                            const long* __restrict__ labels, const float* __restrict__ W,
                                                            const float* __restrict__ bias, int N, int L, int k, int n_cls,
                                                            int subtyping, float scale, float* __restrict__ loss,
                                                            float* __restrict__ dl, long* __restrict__ pt, int B) {
    __shared__ float lg[64][CI_MAXO];
    __shared__ float lsum[256];
    __shared__ int srow[64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int O = 2 * n_cls, R = 2 * k;
    if (tid < R) srow[tid] = ids[(size_t)b * R + tid];
    __syncthreads();
    // a wave takes the rows wave, wave + 4, ...: FOUR of them in flight at once (one row after the other was 2k / 4 dependent round
    // trips: 18 us at k = 8), the classifier rows read once per four rows
    for (int r0 = wave; r0 < R; r0 += 16) {
        float acc[4][CI_MAXO];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int o = 0; o < CI_MAXO; ++o) acc[i][o] = 0.f;
        for (int c0 = lane * 8; c0 < L; c0 += 512) {
            float x[4][8];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                load8<T>(h + ((size_t)b * N + srow[min(r0 + 4 * i, R - 1)]) * L + c0, x[i]);
#pragma unroll
            for (int o = 0; o < CI_MAXO; ++o)
                if (o < O) {
                    float w[8];
                    load8<float>(W + (size_t)o * L + c0, w);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[i][o] += x[i][e] * w[e];
                }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int o = 0; o < CI_MAXO; ++o)
                if (o < O && r0 + 4 * i < R) {
                    const float t = wave_sum(acc[i][o]);
                    if (lane == 0) lg[r0 + 4 * i][o] = t + bias[o];
                }
    }
    __syncthreads();
    const int lab = (int)labels[b];
    float myloss = 0.f;
    for (int p = tid; p < n_cls * R; p += 256) {
        const int c = p / R, r = p - c * R;
        int t;
        float cnt;
        if (c == lab) { t = r < k ? 1 : 0; cnt = (float)R; }
        else { t = (subtyping && r < k) ? 0 : -1; cnt = subtyping ? (float)k : 0.f; }
        const float x0 = lg[r][2 * c], x1 = lg[r][2 * c + 1];
        float* d = dl + ((size_t)b * R + r) * O + 2 * c;
        long pred = -1;
        if (t < 0) {
            d[0] = 0.f; d[1] = 0.f;
        } else {
            const float inv = 1.f / cnt;
            const float mx = fmaxf(x0, x1);
            const float lse = mx + logf(expf(x0 - mx) + expf(x1 - mx));
            myloss += (lse - (t ? x1 : x0)) * inv;
            d[0] = (expf(x0 - lse) - (t == 0 ? 1.f : 0.f)) * inv * scale;
            d[1] = (expf(x1 - lse) - (t == 1 ? 1.f : 0.f)) * inv * scale;
            pred = x1 > x0 ? 1 : 0;                          // first maximum, as torch.topk(logits, 1) picks it
        }
        pt[((size_t)b * n_cls + c) * R + r] = pred;
        pt[(size_t)B * n_cls * R + ((size_t)b * n_cls + c) * R + r] = t;
    }
    lsum[tid] = myloss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) lsum[tid] += lsum[tid + o]; __syncthreads(); }
    if (tid == 0) loss[b] = lsum[0] * scale;
}
// grid (B, L / 64): a workgroup owns 64 columns of one bag's 2k rows; its four waves take the rows r = wave, wave + 4, ... with the
// loads of FOUR rows (h and dz) in flight at once - the first form (one workgroup per bag, a thread walking the 2k rows with a
// dependent load -> add -> store each) was 2k serial round trips to HBM: 46 us at 64 x 4096 x 512, k = 8.
// A bag whose top and bottom rows coincide somewhere (N < 2k, or ties: a uniform soft-max) is walked by ONE wave in row order, so
// a row taken twice is added twice (as index_select's backward does).
template <typename T>
__global__ __launch_bounds__(256) void clam_inst_bwd_kernel(const T* __restrict__ h, const int* __restrict__ ids,
                                                            const float* __restrict__ W, const float* __restrict__ dl,
                                                            const float* __restrict__ up, int N, int L, int k, int n_cls,
                                                            T* __restrict__ dz, float* __restrict__ part) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int O = 2 * n_cls, R = 2 * k;
    __shared__ float sdl[64][CI_MAXO];
    __shared__ int srow[64];
    __shared__ float red[4][CI_MAXO + 1][64];
    const float u = up[b];
    for (int p = tid; p < R * O; p += 256) sdl[p / O][p % O] = dl[(size_t)b * R * O + p] * u;
    if (tid < R) srow[tid] = ids[(size_t)b * R + tid];
    __syncthreads();
    int dup = 0;
    for (int p = tid; p < R * R; p += 256) {
        const int i = p / R, j = p - i * R;
        dup |= (i < j && srow[i] == srow[j]);
    }
    const int serial = __syncthreads_or(dup);
    const int c = blockIdx.y * 64 + lane;
    const bool col = c < L;
    const int nw = serial ? 1 : 4;
    float wcol[CI_MAXO], dw[CI_MAXO], gsum = 0.f;
#pragma unroll
    for (int o = 0; o < CI_MAXO; ++o) { wcol[o] = (o < O && col) ? W[(size_t)o * L + c] : 0.f; dw[o] = 0.f; }
    if (wave < nw && col) {
        for (int r0 = wave; r0 < R; r0 += 4 * nw) {
            float x[4], z[4];
            size_t at[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = r0 + i * nw;
                at[i] = ((size_t)b * N + srow[min(r, R - 1)]) * L + c;
                x[i] = to_f<T>(h[at[i]]);
                z[i] = to_f<T>(dz[at[i]]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = r0 + i * nw;
                if (r >= R) break;
                if (serial && i > 0) z[i] = to_f<T>(dz[at[i]]);          // (a repeated row sees the earlier addition)
                float g = 0.f;
#pragma unroll
                for (int o = 0; o < CI_MAXO; ++o)
                    if (o < O) { g += sdl[r][o] * wcol[o]; dw[o] += sdl[r][o] * x[i]; }
                if (x[i] > 0.f) {
                    dz[at[i]] = from_f<T>(z[i] + g);
                    gsum += g;
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < CI_MAXO; ++o) red[wave][o][lane] = dw[o];
    red[wave][CI_MAXO][lane] = gsum;
    __syncthreads();
    float* prow = part + (size_t)b * (O * (L + 1) + L);
    if (col)
        for (int o = wave; o <= O; o += 4) {                 // the waves share the O + 1 output rows of these 64 columns
            const int oo = o < O ? o : CI_MAXO;
            const float t = (red[0][oo][lane] + red[1][oo][lane]) + (red[2][oo][lane] + red[3][oo][lane]);
            if (o < O) prow[(size_t)o * L + c] = t;
            else prow[(size_t)O * (L + 1) + c] = t;
        }
    if (blockIdx.y == 0 && tid < O) {
        float t = 0.f;
        for (int r = 0; r < R; ++r) t += sdl[r][tid];
        prow[(size_t)O * L + tid] = t;
    }
}
extern "C" int murcl_clam_inst_fwd(const void* h, const int* ids, const long* labels, const float* W, const float* bias, int B,
                                   int N, int L, int k, int n_cls, int subtyping, float scale, float* loss, float* dl, long* pt,
                                   int dtype, hipStream_t st) {
    if (B <= 0) return 0;
    if (k <= 0 || k > 32 || n_cls <= 0 || 2 * n_cls > CI_MAXO || L % 8) return -1;
    if (dtype == MURCL_DTYPE_F32) hipLaunchKernelGGL(clam_inst_fwd_kernel<float>, dim3(B), dim3(256), 0, st, (const float*)h, ids, labels, W, bias, N, L, k, n_cls, subtyping, scale, loss, dl, pt, B);
    else if (dtype == MURCL_DTYPE_BF16) hipLaunchKernelGGL(clam_inst_fwd_kernel<bf16_t>, dim3(B), dim3(256), 0, st, (const bf16_t*)h, ids, labels, W, bias, N, L, k, n_cls, subtyping, scale, loss, dl, pt, B);
    else return -1;
    return MURCL_CHECK_LAUNCH();
}
extern "C" int murcl_clam_inst_bwd(const void* h, const int* ids, const float* W, const float* dl, const float* up, int B, int N,
                                   int L, int k, int n_cls, void* dz, float* part /* [B][2 n_cls (L+1) + L] */, int dtype,
                                   hipStream_t st) {
    if (B <= 0) return 0;
    if (k <= 0 || k > 32 || n_cls <= 0 || 2 * n_cls > CI_MAXO) return -1;
    const dim3 grid(B, (L + 63) / 64);
    if (dtype == MURCL_DTYPE_F32) hipLaunchKernelGGL(clam_inst_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)h, ids, W, dl, up, N, L, k, n_cls, (float*)dz, part);
    else if (dtype == MURCL_DTYPE_BF16) hipLaunchKernelGGL(clam_inst_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)h, ids, W, dl, up, N, L, k, n_cls, (bf16_t*)dz, part);
    else return -1;
    return MURCL_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- y *= k (dropout keep-mask multiply), y = dy * k
template <typename T>
__global__ void mul_kernel(const T* __restrict__ x, const T* __restrict__ k, T* __restrict__ y, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = from_f<T>(to_f<T>(x[i]) * to_f<T>(k[i]));
}
// Dropout keep mask in the compute dtype: out[i] = scale with probability keep_p (quantised to 1/256), else 0.
// Counter-based: element i takes byte (i & 7) of splitmix64(seed + i / 8), so a mask is a pure function of (seed, i) -
// one write pass instead of torch's uniform draw + compare + cast + scale (four passes over an f32 tensor twice the size).
template <typename T>
__global__ __launch_bounds__(256) void dropout_mask_kernel(T* __restrict__ out, long n8, long n, unsigned thresh, float scale,
                                                           unsigned long long seed) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < n8; g += stride) {
        const unsigned long long r = murcl_drop_word(seed, g);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ((unsigned)(r >> (8 * e)) & 255u) < thresh ? scale : 0.f;
        if (8 * g + 8 <= n) {
            store8<T>(out + 8 * g, v);
        } else {
            for (int e = 0; 8 * g + e < n; ++e) out[8 * g + e] = from_f<T>(v[e]);
        }
    }
}
extern "C" int murcl_dropout_mask(void* out, long n, float keep_p, float scale, unsigned long long seed, int dtype,
                                  hipStream_t st) {
    if (n <= 0) return 0;
    if (!(keep_p >= 0.f && keep_p <= 1.f)) return -1;
    const unsigned thresh = (unsigned)(keep_p * 256.f + 0.5f);
    const long n8 = (n + 7) / 8;
    int grid = (int)((n8 + 255) / 256);
    if (grid > 8192) grid = 8192;
    if (dtype == MURCL_DTYPE_F32) hipLaunchKernelGGL(dropout_mask_kernel<float>, dim3(grid), dim3(256), 0, st, (float*)out, n8, n, thresh, scale, seed);
    else if (dtype == MURCL_DTYPE_BF16) hipLaunchKernelGGL(dropout_mask_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (bf16_t*)out, n8, n, thresh, scale, seed);
    else return -1;
    return MURCL_CHECK_LAUNCH();
}

extern "C" int murcl_mul(const void* x, const void* k, void* y, long n, int dtype, hipStream_t st) {
    if (n <= 0) return 0;
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (dtype == MURCL_DTYPE_F32) hipLaunchKernelGGL(mul_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (const float*)k, (float*)y, n);
    else if (dtype == MURCL_DTYPE_BF16) hipLaunchKernelGGL(mul_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)k, (bf16_t*)y, n);
    else return -1;
    return MURCL_CHECK_LAUNCH();
}
