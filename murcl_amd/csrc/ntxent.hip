// K8/K9: NT-Xent forward + backward and the per-pair cosine reward in ONE launch
// (utils/losses.py:24-41, train_MuRCL.py:253,282-283).
//
//   zh_i = z_i / max(|z_i|, 1e-8);  S = zh zh^T / tau;  loss = mean_i [ lse_{j!=i} S_ij - S_{i,pos(i)} ]
//   d loss / d zh_i = (1/(n tau)) sum_{j!=i} [ P_ij + P_ji - 2 [j == pos(i)] ] zh_j,   P_ij = exp(S_ij - lse_i)
//
// S is symmetric, so the column term P_ji needs only lse_j: the lse of every row comes first, the gradient second.
// n <= 128 (one GPU's batch): ONE launch, every workgroup recomputes all logits in LDS.  Larger n (the global batch of a
// multi-GPU step): two launches, see below.  Nothing of size n x n is ever materialised in memory (the reference builds
// an n x n x 128 broadcast temp, losses.py:29).  Rows [grad_lo, grad_hi) of each view receive a gradient (bag sharding
// across ranks: loss rows are global, gradients local).
#include "common.h"

#define NX_P 128          // projection dim

// Row layout: blocks of `ps` rows alternate between the two views - [view 0 | view 1] with ps = B for one process,
// [rank 0: view 0 | view 1][rank 1: view 0 | view 1]... with ps = bags per rank for an all-gathered global batch (the
// gathered buffer is used as it arrives, no re-ordering copies).  Bag ids are global: rank * ps + b.
__device__ __forceinline__ int nx_pos(int i, int ps) { return ((i / ps) & 1) ? i - ps : i + ps; }
__device__ __forceinline__ int nx_bag(int i, int ps) { return (i / (2 * ps)) * ps + i % ps; }
__device__ __forceinline__ bool nx_view0(int i, int ps) { return ((i / ps) & 1) == 0; }

// ------------------------------------------------------------------------------------------ n > 128: two launches
// (the global batch of a multi-GPU step: n = 2 x 64 x ranks).  The lse of EVERY row must be known before any gradient
// weight P_ji can be formed; the kernel boundary is that dependency (an agent-scope grid barrier inside one kernel cost
// more than a launch here: release/acquire fences write back / invalidate whole L2s on a multi-XCD part, and 64
// workgroups of 16 rows left three quarters of the chip idle: 206 us at n = 1024).
//   launch 1 (ntxent_stats_kernel): workgroup (row block of 16, column tile of 256) - n/16 x n/256 workgroups, 16 waves
//     each.  The tile is loaded raw and normalised on the way into LDS (row block 0 also publishes z-hat and 1/|z| for
//     launch 2), every wave forms one 16 x 16 block of logits on the f32 matrix cores and the workgroup leaves
//     (max, sum-exp, positive logit) per row for its tile.
//   launch 2 (ntxent_grad_kernel): one workgroup per row block that owns rows in [grad_lo, grad_hi) walks all column
//     tiles: lse_j from the tile statistics, the logit block again, W_ij = (P_ij + P_ji - 2[j == pos(i)]) / (n tau)
//     into LDS, G += W . zh on the matrix cores (8 column blocks x 2 k-halves over the 16 waves), then the projection
//     through the normalisation.  Workgroup 0 also reduces the loss in a fixed order.
#define NXT_LD 132        // LDS row stride of z-hat rows (floats): conflict-free 4-byte MFMA operand reads
#define NXT_WLD 260       // row stride of the weight block
#define NXT_TC 256        // rows of a column tile
#define NXT_LDS ((NXT_TC * NXT_LD + 16 * NXT_LD + 16 * NXT_WLD + 16 + NXT_TC + 128) * 4)

__device__ __forceinline__ float nx_inv_norm(float ss) { return __builtin_amdgcn_rsqf(fmaxf(ss, 1e-16f)); }   // 1/max(|z|,1e-8)
__device__ __forceinline__ float nx_sum32(float v) {       // sum over the aligned group of 32 lanes
    v = row16_sum(v);
    return v + __shfl_xor(v, 16, 64);
}
// merge two (max, sum-exp) pairs
__device__ __forceinline__ void nx_merge(float& m, float& l, float m2, float l2) {
    const float mn = fmaxf(m, m2);
    const float x = (m == -INFINITY) ? 0.f : l * __expf(m - mn);
    const float y = (m2 == -INFINITY) ? 0.f : l2 * __expf(m2 - mn);
    l = x + y;
    m = mn;
}

__global__ __launch_bounds__(1024) void ntxent_stats_kernel(const float* __restrict__ z, int n, int ps, float inv_tau,
                                                            float* __restrict__ zh_ws, float* __restrict__ inv_ws,
                                                            float* __restrict__ stats_ws, float* __restrict__ sim) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* zj = sm;                              // [256][NXT_LD]
    float* zi = zj + NXT_TC * NXT_LD;            // [16][NXT_LD]
    float* st = zi + 16 * NXT_LD;                // [16 waves][16 rows][3]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q4 = lane >> 4, r16 = lane & 15;
    const int row0 = blockIdx.x * 16, j0 = blockIdx.y * NXT_TC;
    // column tile: 256 rows x 32 float4, 32 lanes per row
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int idx = tid + 1024 * u, r = idx >> 5, c4 = idx & 31, j = j0 + r;
        const f32x4 v = (j < n) ? *(const f32x4*)(z + (size_t)j * NX_P + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float inv = nx_inv_norm(nx_sum32(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]));
        const f32x4 h = f32x4{v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv};
        *(f32x4*)(zj + r * NXT_LD + 4 * c4) = h;
        if (blockIdx.x == 0 && j < n) {
            *(f32x4*)(zh_ws + (size_t)j * NX_P + 4 * c4) = h;
            if (c4 == 0) inv_ws[j] = inv;
        }
    }
    if (tid < 512) {                             // own rows
        const int i = tid >> 5, c4 = tid & 31, ig = row0 + i;
        const f32x4 v = (ig < n) ? *(const f32x4*)(z + (size_t)ig * NX_P + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float inv = nx_inv_norm(nx_sum32(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]));
        *(f32x4*)(zi + i * NXT_LD + 4 * c4) = f32x4{v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv};
    }
    __syncthreads();
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        const float* ap = zi + r16 * NXT_LD + q4;
        const float* bp = zj + (16 * wave + r16) * NXT_LD + q4;
#pragma unroll
        for (int kk = 0; kk < NX_P / 4; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * kk], bp[4 * kk], acc, 0, 0, 0);
    }
    // lane holds S[4q4 + r][16 wave + r16] of the tile: reduce over the 16 lanes that share a row
    const int j = j0 + 16 * wave + r16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ig = row0 + 4 * q4 + r;
        const int pos = nx_pos(ig, ps);
        const float sv = acc[r] * inv_tau;
        const bool valid = j < n && j != ig;
        const float m = row16_max(valid ? sv : -INFINITY);
        const float l = row16_sum(valid ? __expf(sv - m) : 0.f);
        if (j == pos && ig < n) {
            st[(wave * 16 + 4 * q4 + r) * 3 + 2] = sv;                   // exactly one lane of one wave per row, if any
            if (sim && nx_view0(ig, ps)) sim[nx_bag(ig, ps)] = sv / inv_tau;   // cosine of the positive pair (K9)
        }
        if (r16 == 0) {
            st[(wave * 16 + 4 * q4 + r) * 3 + 0] = m;
            st[(wave * 16 + 4 * q4 + r) * 3 + 1] = l;
        }
    }
    __syncthreads();
    if (tid < 16) {
        const int ig = row0 + tid;
        const int pos = nx_pos(ig, ps);
        float m = -INFINITY, l = 0.f, sp = 0.f;
        for (int w = 0; w < 16; ++w) {
            nx_merge(m, l, st[(w * 16 + tid) * 3], st[(w * 16 + tid) * 3 + 1]);
            if (pos >= j0 + 16 * w && pos < j0 + 16 * w + 16) sp = st[(w * 16 + tid) * 3 + 2];
        }
        if (ig < n) {
            float* o = stats_ws + ((size_t)blockIdx.y * n + ig) * 3;
            o[0] = m; o[1] = l; o[2] = sp;
        }
    }
}

__global__ __launch_bounds__(1024) void ntxent_grad_kernel(int n, int ps, float inv_tau, const float* __restrict__ zh_ws,
                                                           const float* __restrict__ inv_ws,
                                                           const float* __restrict__ stats_ws, int ntile,
                                                           float* __restrict__ dz, float* __restrict__ loss_out,
                                                           int grad_lo, int grad_hi) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* zj = sm;                              // [256][NXT_LD]
    float* zi = zj + NXT_TC * NXT_LD;            // [16][NXT_LD]
    float* wt = zi + 16 * NXT_LD;                // [16][NXT_WLD]
    float* lse_i_s = wt + 16 * NXT_WLD;          // [16]
    float* lsej = lse_i_s + 16;                  // [256]
    float* dotp = lsej + NXT_TC;                 // [8][16]
    float* red = zj;                             // [1024] loss reduction scratch (before the tile walk)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q4 = lane >> 4, r16 = lane & 15;
    const int row0 = blockIdx.x * 16;
    // lse and positive logit of row i from the per-tile statistics
    auto row_stats = [&](int i, float& lse, float& sp) {
        float m = -INFINITY, l = 0.f;
        sp = 0.f;
        for (int t = 0; t < ntile; ++t) {
            const float* o = stats_ws + ((size_t)t * n + i) * 3;
            nx_merge(m, l, o[0], o[1]);
            sp += o[2];
        }
        lse = m + __logf(l);
    };
    if (blockIdx.x == 0) {                       // loss = mean_i (lse_i - s_i,pos(i)), summed in a fixed order
        float t = 0.f;
        for (int i = tid; i < n; i += 1024) {
            float lse, sp;
            row_stats(i, lse, sp);
            t += (lse - sp) / (float)n;
        }
        red[tid] = t;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        if (tid == 0) loss_out[0] = red[0];
        __syncthreads();
    }
    if (!dz) return;
    bool any = false;
    for (int r = 0; r < 16; ++r) {
        const int ig = row0 + r, bg = nx_bag(ig, ps);
        any |= (ig < n && bg >= grad_lo && bg < grad_hi);
    }
    if (!any) {                                  // rows of other ranks: their gradient slice is zero here
        for (int idx = tid; idx < 16 * NX_P; idx += 1024) {
            const int row = row0 + idx / NX_P;
            if (row < n) dz[(size_t)row * NX_P + idx % NX_P] = 0.f;
        }
        return;
    }
    if (tid < 512) {
        const int i = tid >> 5, c4 = tid & 31, ig = row0 + i;
        *(f32x4*)(zi + i * NXT_LD + 4 * c4) = (ig < n) ? *(const f32x4*)(zh_ws + (size_t)ig * NX_P + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    } else if (tid < 528) {
        const int ig = row0 + tid - 512;
        float lse = 0.f, sp;
        if (ig < n) row_stats(ig, lse, sp);
        lse_i_s[tid - 512] = lse;
    }
    __syncthreads();
    float a[NX_P / 4];                           // MFMA a-operands of the own rows: row r16, k = 4kk + q4
#pragma unroll
    for (int kk = 0; kk < NX_P / 4; ++kk) a[kk] = zi[r16 * NXT_LD + 4 * kk + q4];
    int posr[4];
    float lse_i[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ig = row0 + 4 * q4 + r;
        posr[r] = nx_pos(ig, ps);
        lse_i[r] = lse_i_s[4 * q4 + r];
    }
    const float scale = inv_tau / (float)n;
    f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
    const int gc = wave & 7, gh = wave >> 3;     // output column block, k-half of the tile

    f32x4 nxt[8];
    float nlse = 0.f;
    auto fetch = [&](int j0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = tid + 1024 * u, j = j0 + (idx >> 5);
            nxt[u] = (j < n) ? *(const f32x4*)(zh_ws + (size_t)j * NX_P + 4 * (idx & 31)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (tid < NXT_TC) {
            float sp;
            nlse = 0.f;
            if (j0 + tid < n) row_stats(j0 + tid, nlse, sp);
        }
    };
    fetch(0);
    for (int j0 = 0; j0 < n; j0 += NXT_TC) {
        LDS_BARRIER();                           // the previous tile has no readers left
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = tid + 1024 * u;
            *(f32x4*)(zj + (idx >> 5) * NXT_LD + 4 * (idx & 31)) = nxt[u];
        }
        if (tid < NXT_TC) lsej[tid] = nlse;
        LDS_BARRIER();
        if (j0 + NXT_TC < n) fetch(j0 + NXT_TC); // in flight under the MFMAs below
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const float* bp = zj + (16 * wave + r16) * NXT_LD + q4;
#pragma unroll
            for (int kk = 0; kk < NX_P / 4; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], bp[4 * kk], acc, 0, 0, 0);
        }
        const int jl = 16 * wave + r16, j = j0 + jl;
        const int jp = (jl & ~15) | ((jl & 3) << 2) | ((jl >> 2) & 3);       // storage column of W (see the G loop)
        const float lj = lsej[jl];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ig = row0 + 4 * q4 + r;
            const float sv = acc[r] * inv_tau;
            float w = 0.f;
            if (ig < n && j < n && j != ig) {
                w = __expf(sv - lse_i[r]) + __expf(sv - lj);
                if (j == posr[r]) w -= 2.f;
            }
            wt[(4 * q4 + r) * NXT_WLD + jp] = w * scale;
        }
        LDS_BARRIER();
        // G += W . zh over this wave's 128 tile rows.  k-step kk, lane quarter q4 takes tile row 16(kk>>2) + 4q4 + (kk&3):
        // with a row stride of 132 floats the four quarters then read banks 16 apart (a natural k = 4kk + q4 order is a
        // 4-way conflict), and W was stored with those two 2-bit column fields swapped so that its reads stay linear.
        const float* ap = wt + r16 * NXT_WLD + 128 * gh + q4;
        const float* bp = zj + (128 * gh + 4 * q4) * NXT_LD + 16 * gc + r16;
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk)
            g = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * kk], bp[(16 * (kk >> 2) + (kk & 3)) * NXT_LD], g, 0, 0, 0);
    }
    // ---- add the two k-halves, project through the normalisation: dz = (g - (zh.g) zh) / |z|
    __syncthreads();
    if (gh == 1) *(f32x4*)(wt + (gc * 64 + lane) * 4) = g;
    __syncthreads();
    float zv[4];
    if (gh == 0) {
        const f32x4 o = *(const f32x4*)(wt + (gc * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            g[r] += o[r];
            zv[r] = zi[(4 * q4 + r) * NXT_LD + 16 * gc + r16];
            const float t = row16_sum(g[r] * zv[r]);
            if (r16 == 0) dotp[gc * 16 + 4 * q4 + r] = t;
        }
    }
    __syncthreads();
    if (gh == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rl = 4 * q4 + r, row = row0 + rl;
            if (row >= n) continue;
            const int bag = nx_bag(row, ps);
            const bool want = bag >= grad_lo && bag < grad_hi;
            float dot = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) dot += dotp[w * 16 + rl];
            dz[(size_t)row * NX_P + 16 * gc + r16] = want ? (g[r] - dot * zv[r]) * inv_ws[row] : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------ n <= 128: no grid barrier
// The single-GPU batch (64 bags -> n = 128) fits one CU's LDS: z-hat and the n x n logit / weight matrix (2 x 66 KiB,
// row stride 132 floats = conflict-free 4-byte MFMA operand reads).  n/16 workgroups each recompute the logits and the
// row statistics in full (v_mfma_f32_16x16x4_f32; cheaper than exchanging lse through a grid barrier) and then produce
// the gradient of their own 16 rows: no barrier across workgroups, no atomic, no workspace traffic.
#define NXS_LD 132
__global__ __launch_bounds__(1024) void ntxent_small_kernel(const float* __restrict__ z, int n, int ps, float inv_tau,
                                                            float* __restrict__ dz, float* __restrict__ sim,
                                                            float* __restrict__ loss_out, int grad_lo, int grad_hi) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // blockIdx.y: one of several independent batches of the same size (the T patch steps of a training step), back to back
    z += (size_t)blockIdx.y * n * NX_P;
    if (dz) dz += (size_t)blockIdx.y * n * NX_P;
    if (sim) sim += (size_t)blockIdx.y * (n / 2);
    loss_out += blockIdx.y;
    float* zh = sm;                                   // [128][NXS_LD]
    float* S = sm + 128 * NXS_LD;                     // [128][NXS_LD] logits, then gradient weights
    float* lse = S + 128 * NXS_LD;                    // [128]
    float* inorm = lse + 128;                         // [128]
    float* dotp = inorm + 128;                        // [2][128]
    float* red = dotp + 256;                          // [128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q4 = lane >> 4, r16 = lane & 15;

    // ---- phase 0: normalised rows (8 threads per row, 16 columns each); rows >= n are zero
    {
        const int i = tid >> 3, c0 = (tid & 7) * 16;
        float v[16], ss = 0.f;
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
            const f32x4 t = (i < n) ? *(const f32x4*)(z + (size_t)i * NX_P + c0 + e) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[e + k] = t[k]; ss += t[k] * t[k]; }
        }
        ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-8f);
#pragma unroll
        for (int e = 0; e < 16; ++e) zh[i * NXS_LD + c0 + e] = v[e] * inv;
        if ((tid & 7) == 0) inorm[i] = inv;
    }
    __syncthreads();

    // ---- phase 1: S = zh zh^T / tau, 64 tiles of 16 x 16, four per wave (row tile = wave & 7)
    const int ti = wave & 7, tj0 = (wave >> 3) * 4;
    {
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* ar = zh + (16 * ti + r16) * NXS_LD + q4;
#pragma unroll 4
        for (int kk = 0; kk < NX_P / 4; ++kk) {
            const float a = ar[4 * kk];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float b = zh[(16 * (tj0 + t) + r16) * NXS_LD + 4 * kk + q4];
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
            }
        }
        // lane holds S[16ti + 4q4 + r][16tj + r16]
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(16 * ti + 4 * q4 + r) * NXS_LD + 16 * (tj0 + t) + r16] = acc[t][r] * inv_tau;
    }
    __syncthreads();

    // ---- row statistics: 8 threads per row, columns (tid & 7) + 8u
    const int i = tid >> 3, cb = tid & 7;
    const int pos = nx_pos(i, ps);
    float my_lse = 0.f;
    {
        float m = -INFINITY, l = 0.f, sp = 0.f;
#pragma unroll 4
        for (int u = 0; u < 16; ++u) {
            const int j = cb + 8 * u;
            const float sv = S[i * NXS_LD + j];
            if (j < n && j != i) {
                const float mn = fmaxf(m, sv);
                l = l * __expf(m - mn) + __expf(sv - mn);
                m = mn;
            }
            if (j == pos) sp = sv;
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
            const float mn = fmaxf(m, m2);
            const float a = (m == -INFINITY) ? 0.f : l * __expf(m - mn);
            const float b = (m2 == -INFINITY) ? 0.f : l2 * __expf(m2 - mn);
            l = a + b;
            m = mn;
            sp += __shfl_xor(sp, o, 64);
        }
        my_lse = m + __logf(l);
        if (cb == 0) {
            lse[i] = my_lse;
            red[i] = (i < n) ? (my_lse - sp) / (float)n : 0.f;
            if (i < n && sim && blockIdx.x == 0 && nx_view0(i, ps)) sim[nx_bag(i, ps)] = sp / inv_tau;
        }
    }
    __syncthreads();
    if (tid < 64) {
        float t = red[tid] + red[tid + 64];
        t = wave_sum(t);
        if (tid == 0 && blockIdx.x == 0) loss_out[0] = t;
    }
    if (!dz) return;

    // ---- gradient weights in place: W_ij = (P_ij + P_ji - 2 [j == pos(i)]) / (n tau), 0 on the diagonal / padding
    {
        const float scale = inv_tau / (float)n;
#pragma unroll 4
        for (int u = 0; u < 16; ++u) {
            const int j = cb + 8 * u;
            const float sv = S[i * NXS_LD + j];
            float w = 0.f;
            if (i < n && j < n && j != i) {
                w = __expf(sv - my_lse) + __expf(sv - lse[j]);
                if (j == pos) w -= 2.f;
            }
            S[i * NXS_LD + j] = w * scale;
        }
    }
    __syncthreads();

    // ---- G = W zh for THIS workgroup's 16 rows (row tile blockIdx.x; waves 0..7 take one 16-column tile each), then
    // dz = (g - (zh.g) zh) / |z|.  Everything above is recomputed by every workgroup (the n x n logits are needed in
    // full for the column term P_ji), so the workgroups never exchange anything: no grid barrier.
    {
        const int tr = blockIdx.x;
        f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
        float zv[4], pd[4] = {0.f, 0.f, 0.f, 0.f};
        if (wave < 8) {
            const float* ar = S + (16 * tr + r16) * NXS_LD + q4;
#pragma unroll 8
            for (int kk = 0; kk < 128 / 4; ++kk) {
                const float a = ar[4 * kk];
                const float b = zh[(4 * kk + q4) * NXS_LD + 16 * wave + r16];
                g = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, g, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                zv[r] = zh[(16 * tr + 4 * q4 + r) * NXS_LD + 16 * wave + r16];
                pd[r] = g[r] * zv[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = row16_sum(pd[r]);
                if (r16 == 0) dotp[wave * 16 + 4 * q4 + r] = t;
            }
        }
        __syncthreads();
        if (wave < 8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = 4 * q4 + r, row = 16 * tr + rl;
                if (row >= n) continue;
                const int bag = nx_bag(row, ps);
                const bool want = bag >= grad_lo && bag < grad_hi;
                float dot = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) dot += dotp[w * 16 + rl];
                dz[(size_t)row * NX_P + 16 * wave + r16] = want ? (g[r] - dot * zv[r]) * inorm[row] : 0.f;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ n <= 128, round 4: one exchange
// ntxent_small_kernel above makes every one of its n/16 workgroups recompute ALL n x n logits (the column term P_ji of the gradient
// needs lse_j of every row): 64 tiles of exact-f32 MFMAs on one CU = ~8 us of its ~21.  Here a workgroup forms only the logits of
// its OWN 16 rows (8 tiles), publishes their lse (and loss terms) as 8-byte {value, generation} granules - one agent-scope store
// each, MI355X_MICROARCH.md "data-tagged granules" - and collects the other workgroups' granules with agent-scope polls: one
// hand-off (~1-2 us) instead of 7/8 of the matrix work.  The exchange buffer is caller-owned and zeroed ONCE; nothing is reset
// between launches: the generation lives in the buffer itself (two control words behind the granules) - every workgroup reads it
// at its start (the buffer's first 64 bytes), tags its granules with generation + 1, and the workgroup that finishes last advances it, so launches on a stream
// (and replays of a captured launch) never see each other's granules as their own.  All workgroups of a batch must be resident
// together: n/16 <= 8 workgroups per batch on 256 CUs; the spin is bounded (gives up -> NaN loss).
#define NXX_LD 132
__device__ __forceinline__ unsigned long long nxx_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void nxx_store(unsigned long long* p, float v, unsigned gen) {
    __hip_atomic_store(p, ((unsigned long long)gen << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(1024) void ntxent_xchg_kernel(const float* __restrict__ z, int n, int ps, float inv_tau,
                                                           float* __restrict__ dz, float* __restrict__ sim,
                                                           float* __restrict__ loss_out, int grad_lo, int grad_hi,
                                                           unsigned long long* __restrict__ xchg, unsigned* __restrict__ ctrl) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ unsigned gen_s;
    if (threadIdx.x == 0) gen_s = __hip_atomic_load(ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;   // (visible after phase 0's barrier)
    z += (size_t)blockIdx.y * n * NX_P;
    if (dz) dz += (size_t)blockIdx.y * n * NX_P;
    if (sim) sim += (size_t)blockIdx.y * (n / 2);
    loss_out += blockIdx.y;
    xchg += (size_t)blockIdx.y * 256;                 // [128] lse granules, [128] loss-term granules
    float* zh = sm;                                   // [128][NXX_LD]
    float* Ss = sm + 128 * NXX_LD;                    // [16][NXX_LD] logits of the own rows, then their gradient weights
    float* lse_all = Ss + 16 * NXX_LD;                // [128]
    float* inorm = lse_all + 128;                     // [128]
    float* dotp = inorm + 128;                        // [8][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q4 = lane >> 4, r16 = lane & 15;
    const int tr = blockIdx.x;

    // ---- phase 0: normalised rows (8 threads per row, 16 columns each); rows >= n are zero
    {
        const int i = tid >> 3, c0 = (tid & 7) * 16;
        float v[16], ss = 0.f;
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
            const f32x4 t = (i < n) ? *(const f32x4*)(z + (size_t)i * NX_P + c0 + e) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[e + k] = t[k]; ss += t[k] * t[k]; }
        }
        ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-8f);
#pragma unroll
        for (int e = 0; e < 16; ++e) zh[i * NXX_LD + c0 + e] = v[e] * inv;
        if ((tid & 7) == 0) inorm[i] = inv;
    }
    __syncthreads();

    const unsigned gen = gen_s;
    // ---- phase 1: the logits of the own 16 rows against all 128 rows: 8 tiles, one per wave 0..7 (exact-f32 MFMA, same operand
    // order as ntxent_small_kernel: bit-identical logits)
    if (wave < 8) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* ar = zh + (16 * tr + r16) * NXX_LD + q4;
        const float* br = zh + (16 * wave + r16) * NXX_LD + q4;
#pragma unroll 8
        for (int kk = 0; kk < NX_P / 4; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[4 * kk], br[4 * kk], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Ss[(4 * q4 + r) * NXX_LD + 16 * wave + r16] = acc[r] * inv_tau;
    }
    __syncthreads();

    // ---- phase 2: statistics of row i = 16 tr + wave (one wave per row, two columns per lane); publish lse_i and the row's loss term
    const int i = 16 * tr + wave;
    const int pos = nx_pos(i, ps);
    const float s0 = Ss[wave * NXX_LD + lane], s1 = Ss[wave * NXX_LD + lane + 64];
    const bool ok0 = i < n && lane < n && lane != i, ok1 = i < n && lane + 64 < n && lane + 64 != i;
    float my_lse = 0.f;
    {
        const float m = wave_max(fmaxf(ok0 ? s0 : -INFINITY, ok1 ? s1 : -INFINITY));
        const float l = wave_sum((ok0 ? __expf(s0 - m) : 0.f) + (ok1 ? __expf(s1 - m) : 0.f));
        if (i < n) my_lse = m + __logf(l);
        if (lane == 0) {
            const float sp = (i < n) ? Ss[wave * NXX_LD + pos] : 0.f;
            nxx_store(xchg + i, my_lse, gen);
            nxx_store(xchg + 128 + i, (i < n) ? (my_lse - sp) / (float)n : 0.f, gen);
            if (i < n && sim && nx_view0(i, ps)) sim[nx_bag(i, ps)] = sp / inv_tau;
        }
    }
    // ---- phase 3: collect every row's lse (waves 0, 1) and, in workgroup 0, the loss terms (waves 2, 3): agent-scope polls
    if (wave < 4 && (wave < 2 || tr == 0)) {
        const int idx = (wave & 1) * 64 + lane + (wave >> 1) * 128;
        float v = 0.f;
        if ((idx & 127) < (int)gridDim.x * 16) {                            // rows of workgroups that exist (the others: never read)
            unsigned long long g = nxx_load(xchg + idx);
            int tries = 0;
            while ((unsigned)(g >> 32) != gen && tries < (1 << 22)) {
                __builtin_amdgcn_s_sleep(1);
                g = nxx_load(xchg + idx);
                ++tries;
            }
            v = __uint_as_float((unsigned)g);
            if ((unsigned)(g >> 32) != gen) v = __builtin_nanf("");      // gave up: make it visible
        }
        if (wave < 2) {
            lse_all[idx] = v;
        } else {                                                           // loss = sum of the 128 terms, fixed order
            const float t = wave_sum(v);
            if (wave == 2) dotp[0] = t; else dotp[1] = t;
        }
    }
    __syncthreads();
    if (tr == 0 && tid == 0) loss_out[0] = dotp[0] + dotp[1];
    if (tid == 0) {
        // every poll of this workgroup is over: the last workgroup of the launch to get here advances the generation
        const unsigned total = gridDim.x * gridDim.y;
        if (__hip_atomic_fetch_add(ctrl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1u) {
            __hip_atomic_store(ctrl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctrl, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!dz) return;
    __syncthreads();                                                       // (dotp is reused below)

    // ---- phase 4: gradient weights of the own rows in place: W_ij = (P_ij + P_ji - 2 [j == pos(i)]) / (n tau)
    {
        const float scale = inv_tau / (float)n;
        float w0 = 0.f, w1 = 0.f;
        if (ok0) w0 = __expf(s0 - my_lse) + __expf(s0 - lse_all[lane]) - (lane == pos ? 2.f : 0.f);
        if (ok1) w1 = __expf(s1 - my_lse) + __expf(s1 - lse_all[lane + 64]) - (lane + 64 == pos ? 2.f : 0.f);
        Ss[wave * NXX_LD + lane] = w0 * scale;
        Ss[wave * NXX_LD + lane + 64] = w1 * scale;
    }
    __syncthreads();

    // ---- phase 5: G = W zh for the own 16 rows (waves 0..7 take one 16-column tile each), dz = (g - (zh.g) zh) / |z|
    {
        f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
        float zv[4], pd[4] = {0.f, 0.f, 0.f, 0.f};
        if (wave < 8) {
            const float* ar = Ss + r16 * NXX_LD + q4;
#pragma unroll 8
            for (int kk = 0; kk < 128 / 4; ++kk) {
                const float a = ar[4 * kk];
                const float b = zh[(4 * kk + q4) * NXX_LD + 16 * wave + r16];
                g = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, g, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                zv[r] = zh[(16 * tr + 4 * q4 + r) * NXX_LD + 16 * wave + r16];
                pd[r] = g[r] * zv[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = row16_sum(pd[r]);
                if (r16 == 0) dotp[wave * 16 + 4 * q4 + r] = t;
            }
        }
        __syncthreads();
        if (wave < 8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = 4 * q4 + r, row = 16 * tr + rl;
                if (row >= n) continue;
                const int bag = nx_bag(row, ps);
                const bool want = bag >= grad_lo && bag < grad_hi;
                float dot = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) dot += dotp[w * 16 + rl];
                dz[(size_t)row * NX_P + 16 * wave + r16] = want ? (g[r] - dot * zv[r]) * inorm[row] : 0.f;
            }
        }
    }
}
// bytes of the exchange buffer for `batches` problems: allocate once per device, ZERO it once, pass it to every call
extern "C" long murcl_ntxent_xchg_bytes(int batches) { return (long)(batches > 0 ? batches : 1) * 256 * 8 + 64; }   // granules + control words
// `batches` independent NT-Xent problems of n <= 128 rows each (rows / pair_stride / gradient window as murcl_ntxent_fwd_bwd) in ONE
// launch of the exchange kernel.  xchg: murcl_ntxent_xchg_bytes(batches) bytes, zeroed once by the caller, never shared by launches
// that may overlap in time (one per device / stream).
extern "C" int murcl_ntxent_small_xchg(const float* z, int batches, int n, int P, float temperature, float* loss, float* dz, float* sim,
                                       int grad_lo, int grad_hi, int pair_stride, void* xchg, hipStream_t stream) {
    if (P != NX_P || n <= 0 || (n & 1) || n > 128 || batches <= 0 || !xchg) return -1;
    const int ps = pair_stride > 0 ? pair_stride : n / 2;
    if (n % (2 * ps)) return -1;
    // the control words (generation, arrival count) are the first 64 bytes of the buffer, the granules follow: calls with different
    // `batches` share one generation
    unsigned* ctrl = (unsigned*)xchg;
    xchg = (char*)xchg + 64;
    constexpr int LDS = (128 * NXX_LD + 16 * NXX_LD + 128 * 2 + 128) * 4;
    static MurclOncePerDevice once;
    if (once.first()) hipFuncSetAttribute((const void*)ntxent_xchg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipLaunchKernelGGL(ntxent_xchg_kernel, dim3((n + 15) / 16, batches), dim3(1024), LDS, stream, z, n, ps, 1.0f / temperature, dz, sim,
                       loss, grad_lo, grad_hi, (unsigned long long*)xchg, ctrl);
    return MURCL_CHECK_LAUNCH();
}

extern "C" long murcl_ntxent_workspace_bytes(int n) {
    const long ntile = (n + NXT_TC - 1) / NXT_TC;
    return ((long)n * NX_P + n + ntile * n * 3 + 16) * 4;          // z-hat, 1/|z|, per-tile row statistics
}

// C-ABI: see include/murcl_amd.h
static int nx_small_launch(const float* z, int n, int ps, float temperature, float* loss, float* dz, float* sim, int grad_lo,
                           int grad_hi, int batches, hipStream_t stream);
extern "C" int murcl_ntxent_fwd_bwd(const float* z, int n, int P, float temperature, float* loss, float* dz,
                                    float* sim, int grad_lo, int grad_hi, int pair_stride, void* workspace,
                                    hipStream_t stream) {
    if (P != NX_P || n <= 0 || (n & 1)) return -1;
    const int ps = pair_stride > 0 ? pair_stride : n / 2;
    if (n % (2 * ps)) return -1;
    if (n <= 128) return nx_small_launch(z, n, ps, temperature, loss, dz, sim, grad_lo, grad_hi, 1, stream);
    static_assert(NXT_LDS <= 160 * 1024, "LDS budget");
    static MurclOncePerDevice once_t;      
    if (once_t.first()) {
        (void)hipFuncSetAttribute((const void*)ntxent_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NXT_LDS);
        (void)hipFuncSetAttribute((const void*)ntxent_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NXT_LDS);
                       
    }
    const int nblk = (n + 15) / 16, ntile = (n + NXT_TC - 1) / NXT_TC;
    if (ntile > 65535) return -1;
    float* zh = (float*)workspace;
    float* inv = zh + (size_t)n * NX_P;
    float* stats = inv + n;
    hipLaunchKernelGGL(ntxent_stats_kernel, dim3(nblk, ntile), dim3(1024), NXT_LDS, stream, z, n, ps, 1.0f / temperature,
                       zh, inv, stats, sim);
    int rc = MURCL_CHECK_LAUNCH();
    if (rc) return rc;
    hipLaunchKernelGGL(ntxent_grad_kernel, dim3(dz ? nblk : 1), dim3(1024), NXT_LDS, stream, n, ps, 1.0f / temperature,
                       (const float*)zh, (const float*)inv, (const float*)stats, ntile, dz, loss, grad_lo, grad_hi);
    return MURCL_CHECK_LAUNCH();
}

static int nx_small_launch(const float* z, int n, int ps, float temperature, float* loss, float* dz, float* sim, int grad_lo,
                           int grad_hi, int batches, hipStream_t stream) {
    constexpr int LDS = (2 * 128 * NXS_LD + 128 * 5) * 4;
    static MurclOncePerDevice once;
    if (once.first()) hipFuncSetAttribute((const void*)ntxent_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipLaunchKernelGGL(ntxent_small_kernel, dim3((n + 15) / 16, batches), dim3(1024), LDS, stream, z, n, ps, 1.0f / temperature, dz,
                       sim, loss, grad_lo, grad_hi);
    return MURCL_CHECK_LAUNCH();
}
// `batches` independent NT-Xent problems of n <= 128 rows each in ONE launch: z [batches][n][P] -> loss [batches],
// dz [batches][n][P] (may be NULL), sim [batches][n/2].  The T patch steps of a training step (train_MuRCL.py:249,277).
extern "C" int murcl_ntxent_fwd_bwd_batched(const float* z, int batches, int n, int P, float temperature, float* loss, float* dz,
                                            float* sim, hipStream_t stream) {
    if (P != NX_P || n <= 0 || (n & 1) || n > 128 || batches <= 0) return -1;
    return nx_small_launch(z, n, n / 2, temperature, loss, dz, sim, 0, n / 2, batches, stream);
}
