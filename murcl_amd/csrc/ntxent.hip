// K8/K9: NT-Xent forward + backward and the per-pair cosine reward in ONE launch
// (utils/losses.py:24-41, train_MuRCL.py:253,282-283).
//
//   zh_i = z_i / max(|z_i|, 1e-8);  S = zh zh^T / tau;  loss = mean_i [ lse_{j!=i} S_ij - S_{i,pos(i)} ]
//   d loss / d zh_i = (1/(n tau)) sum_{j!=i} [ P_ij + P_ji - 2 [j == pos(i)] ] zh_j,   P_ij = exp(S_ij - lse_i)
//
// S is symmetric, so the column term P_ji needs only lse_j: phase 1 computes lse for every row,
// phase 2 the gradient, separated by an agent-scope grid barrier (<= 64 workgroups of 16 rows, always
// co-resident).  Nothing of size n x n is ever materialised (the reference builds an n x n x 128
// broadcast temp, losses.py:29).  Rows [grad_lo, grad_hi) of each view receive a gradient (bag
// sharding across ranks: loss rows are global, gradients local).
#include "common.h"

#define NX_P 128          // projection dim
#define NX_ROWS 16        // rows per workgroup
#define NX_JT 64          // columns (other embeddings) per LDS tile
#define NX_SPIN_LIMIT (1u << 24)

struct NxCtl {            // zeroed by a memset node before every launch
    unsigned arrive[2];
    unsigned timeout;
    unsigned pad;
    float loss;
    float pad2[3];
};

__device__ __forceinline__ bool nx_grid_barrier(unsigned* counter, unsigned target, unsigned* timeout) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its stores
    __syncthreads();
    __shared__ int ok_s;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > NX_SPIN_LIMIT) { ok = 0; __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}

__global__ __launch_bounds__(256) void ntxent_kernel(const float* __restrict__ z, int n, int Bh, float inv_tau,
                                                     float* __restrict__ zh, float* __restrict__ inorm,
                                                     float* __restrict__ lse, NxCtl* ctl, float* __restrict__ dz,
                                                     float* __restrict__ sim, float* __restrict__ loss_out,
                                                     int grad_lo, int grad_hi) {
    __shared__ __attribute__((aligned(16))) float zi[NX_ROWS][NX_P + 4];    // own rows (normalised)
    __shared__ __attribute__((aligned(16))) float zj[NX_JT][NX_P + 4];      // streamed tile
    __shared__ float st[NX_ROWS][NX_JT + 1];                                // S tile / weights
    __shared__ float lse_j[NX_JT];
    const int tid = threadIdx.x, i_loc = tid >> 4, c16 = tid & 15;
    const int row0 = blockIdx.x * NX_ROWS;
    const int nblk = gridDim.x;

    // ---------------- phase 0: normalise own rows
    {
        const int i = row0 + i_loc;
        float v[8], ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = (i < n) ? z[(size_t)i * NX_P + 8 * c16 + e] : 0.f;
            ss += v[e] * v[e];
        }
        ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64); ss += __shfl_xor(ss, 8, 64);
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-8f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            zi[i_loc][8 * c16 + e] = v[e] * inv;
            if (i < n) zh[(size_t)i * NX_P + 8 * c16 + e] = v[e] * inv;
        }
        if (c16 == 0 && i < n) inorm[i] = inv;
    }
    if (!nx_grid_barrier(&ctl->arrive[0], nblk, &ctl->timeout)) return;

    auto load_tile = [&](int j0) {            // zj <- zh[j0 .. j0+63]
        for (int idx = tid; idx < NX_JT * (NX_P / 4); idx += 256) {
            const int r = idx / (NX_P / 4), c4 = idx % (NX_P / 4);
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j0 + r < n) v = *(const f32x4*)(zh + (size_t)(j0 + r) * NX_P + 4 * c4);
            *(f32x4*)&zj[r][4 * c4] = v;
        }
    };
    // S_ij for this thread's row i_loc and columns j = c16 + 16*u (u = 0..3) of the tile
    auto dots = [&](float* s4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) s4[u] = 0.f;
#pragma unroll 4
        for (int k = 0; k < NX_P; k += 4) {
            const f32x4 a = *(const f32x4*)&zi[i_loc][k];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 b = *(const f32x4*)&zj[c16 + 16 * u][k];
                s4[u] += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) s4[u] *= inv_tau;
    };

    // ---------------- phase 1: row log-sum-exp (j != i) and the positive logit
    const int i_glob = row0 + i_loc;
    const int pos = (i_glob < Bh) ? i_glob + Bh : i_glob - Bh;
    float m_run = -INFINITY, l_run = 0.f, s_pos = 0.f;
    for (int j0 = 0; j0 < n; j0 += NX_JT) {
        __syncthreads();
        load_tile(j0);
        __syncthreads();
        float s4[4];
        dots(s4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + c16 + 16 * u;
            if (j < n && j != i_glob) {
                const float mn = fmaxf(m_run, s4[u]);
                l_run = l_run * __expf(m_run - mn) + __expf(s4[u] - mn);
                m_run = mn;
            }
            if (j == pos) s_pos = s4[u];
        }
    }
    // combine the 16 threads of a row
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        const float m2 = __shfl_xor(m_run, o, 64), l2 = __shfl_xor(l_run, o, 64);
        const float mn = fmaxf(m_run, m2);
        const float a = (m_run == -INFINITY) ? 0.f : l_run * __expf(m_run - mn);
        const float b = (m2 == -INFINITY) ? 0.f : l2 * __expf(m2 - mn);
        l_run = a + b;
        m_run = mn;
        s_pos += __shfl_xor(s_pos, o, 64);
    }
    const float my_lse = m_run + __logf(l_run);
    if (c16 == 0) lse_j[i_loc] = (i_glob < n) ? (my_lse - s_pos) / (float)n : 0.f;     // lse_j is free until phase 2
    if (c16 == 0 && i_glob < n) {
        lse[i_glob] = my_lse;
        if (i_glob < Bh && sim) sim[i_glob] = s_pos / inv_tau;      // cosine of the positive pair (K9)
    }
    __syncthreads();
    if (tid == 0) {                           // one adder per workgroup: same-address float atomics serialise
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < NX_ROWS; ++r) t += lse_j[r];
        atomicAdd(&ctl->loss, t);
    }
    if (!nx_grid_barrier(&ctl->arrive[1], nblk, &ctl->timeout)) return;
    if (blockIdx.x == 0 && tid == 0) loss_out[0] = __hip_atomic_load(&ctl->loss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!dz) return;

    // ---------------- phase 2: gradient for rows whose bag index is in [grad_lo, grad_hi)
    const int bag_i = (i_glob < Bh) ? i_glob : i_glob - Bh;
    const bool want = i_glob < n && bag_i >= grad_lo && bag_i < grad_hi;
    bool any = false;
    for (int r = 0; r < NX_ROWS; ++r) {
        const int ig = row0 + r, bg = ig < Bh ? ig : ig - Bh;
        any |= (ig < n && bg >= grad_lo && bg < grad_hi);
    }
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = 0.f;
    if (any) {
        const float scale = inv_tau / (float)n;
        for (int j0 = 0; j0 < n; j0 += NX_JT) {
            __syncthreads();
            load_tile(j0);
            if (tid < NX_JT) lse_j[tid] = (j0 + tid < n) ? lse[j0 + tid] : 0.f;
            __syncthreads();
            float s4[4];
            dots(s4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jl = c16 + 16 * u, j = j0 + jl;
                float w = 0.f;
                if (j < n && j != i_glob) {
                    w = __expf(s4[u] - my_lse) + __expf(s4[u] - lse_j[jl]);
                    if (j == pos) w -= 2.f;
                }
                st[i_loc][jl] = w * scale;
            }
            __syncthreads();
            // thread (i_loc, c16) accumulates columns 8*c16 .. +7
#pragma unroll 8
            for (int jl = 0; jl < NX_JT; ++jl) {
                const float w = st[i_loc][jl];
                const f32x4 b0 = *(const f32x4*)&zj[jl][8 * c16];
                const f32x4 b1 = *(const f32x4*)&zj[jl][8 * c16 + 4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { g[e] += w * b0[e]; g[4 + e] += w * b1[e]; }
            }
        }
    }
    // project through the normalisation: dz = (g - (zh.g) zh) / |z|
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) dot += g[e] * zi[i_loc][8 * c16 + e];
    dot += __shfl_xor(dot, 1, 64); dot += __shfl_xor(dot, 2, 64); dot += __shfl_xor(dot, 4, 64); dot += __shfl_xor(dot, 8, 64);
    if (i_glob < n) {
        const float inv = inorm[i_glob];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            dz[(size_t)i_glob * NX_P + 8 * c16 + e] = want ? (g[e] - dot * zi[i_loc][8 * c16 + e]) * inv : 0.f;
    }
}

// ------------------------------------------------------------------------------------------ n <= 128: no grid barrier
// The single-GPU batch (64 bags -> n = 128) fits one CU's LDS: z-hat and the n x n logit / weight matrix (2 x 66 KiB,
// row stride 132 floats = conflict-free 4-byte MFMA operand reads).  n/16 workgroups each recompute the logits and the
// row statistics in full (v_mfma_f32_16x16x4_f32; cheaper than exchanging lse through a grid barrier) and then produce
// the gradient of their own 16 rows: no barrier across workgroups, no atomic, no workspace traffic.
#define NXS_LD 132
__global__ __launch_bounds__(1024) void ntxent_small_kernel(const float* __restrict__ z, int n, int Bh, float inv_tau,
                                                            float* __restrict__ dz, float* __restrict__ sim,
                                                            float* __restrict__ loss_out, int grad_lo, int grad_hi) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
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
    const int pos = (i < Bh) ? i + Bh : i - Bh;
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
            if (i < Bh && i < n && sim && blockIdx.x == 0) sim[i] = sp / inv_tau;
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
                const int bag = row < Bh ? row : row - Bh;
                const bool want = bag >= grad_lo && bag < grad_hi;
                float dot = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) dot += dotp[w * 16 + rl];
                dz[(size_t)row * NX_P + 16 * wave + r16] = want ? (g[r] - dot * zv[r]) * inorm[row] : 0.f;
            }
        }
    }
}

extern "C" long murcl_ntxent_workspace_bytes(int n) { return (long)sizeof(NxCtl) + (long)n * (NX_P + 2) * 4; }

// C-ABI: see include/murcl_amd.h
extern "C" int murcl_ntxent_fwd_bwd(const float* z, int n, int P, float temperature, float* loss, float* dz,
                                    float* sim, int grad_lo, int grad_hi, void* workspace, hipStream_t stream) {
    if (P != NX_P || n <= 0 || (n & 1)) return -1;
    if (n <= 128) {
        constexpr int LDS = (2 * 128 * NXS_LD + 128 * 5) * 4;
        static bool once = false;
        if (!once) {
            hipFuncSetAttribute((const void*)ntxent_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            once = true;
        }
        hipLaunchKernelGGL(ntxent_small_kernel, dim3((n + 15) / 16), dim3(1024), LDS, stream, z, n, n / 2, 1.0f / temperature, dz, sim,
                           loss, grad_lo, grad_hi);
        return MURCL_CHECK_LAUNCH();
    }
    const int nblk = (n + NX_ROWS - 1) / NX_ROWS;
    if (nblk > 256) return -1;                               // grid barrier needs co-residency
    NxCtl* ctl = (NxCtl*)workspace;
    float* zh = (float*)((char*)workspace + sizeof(NxCtl));
    float* inorm = zh + (size_t)n * NX_P;
    float* lse = inorm + n;
    hipError_t e = hipMemsetAsync(ctl, 0, sizeof(NxCtl), stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(ntxent_kernel, dim3(nblk), dim3(256), 0, stream, z, n, n / 2, 1.0f / temperature, zh, inorm, lse,
                       ctl, dz, sim, loss, grad_lo, grad_hi);
    return MURCL_CHECK_LAUNCH();
}
