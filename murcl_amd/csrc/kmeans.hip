// K-means (Lloyd) on the patch features of one slide - the clustering pre-step that produces the cluster id lists the
// sub-bag sampler reads (reference: wsi_processing/features_clustering.py:10-16, sklearn KMeans(n_clusters, random_state=985)).
//
//   assign:  label[i] = argmin_k |x_i - c_k|^2 = argmin_k (|c_k|^2 - 2 x_i.c_k)      (first minimum on ties, like argmin)
//   update:  c_k = mean of the rows labelled k  (a cluster that lost all its rows keeps its centre)
//
// One pass over X per iteration (HBM: N*d*4 bytes); N = 20 000, d = 512, K = 10 is 41 MB and ~0.3 GFLOP.
// Layout: one WAVE per workgroup owns a contiguous run of rows.  The K centres and the K partial sums of the wave's rows
// live in LDS (2*K*d*4 bytes, <= 128 KiB at K = 16, d = 1024).  A lane owns the same 4*R floats of every row and centre
// (R float4 pieces, 256 floats apart: coalesced 1 KiB row segments), four rows are in flight per step so every centre
// fragment read from LDS is used four times.  Sums are accumulated by the only wave of the workgroup in row order and
// the per-workgroup partials are added up by `kmeans_update_kernel` in workgroup order: the result does not depend on
// scheduling (no float atomics anywhere).
#include "common.h"

#define KM_KMAX 16
#define KM_ROWS 4

// all-lane sum on the VALU only (DPP row reduction + the gfx950 row swaps): ~8 instructions, where the shuffle-based
// wave_sum is six dependent LDS-crossbar round trips - 44 of these per four rows made the kernel latency-bound
__device__ __forceinline__ float km_sum(float v) { return quarters_sum(row16_sum(v)); }

template <int R>
__global__ __launch_bounds__(64) void kmeans_assign_kernel(const float* __restrict__ X, const float* __restrict__ C, int N,
                                                           int K, int rows_per_wg, int* __restrict__ labels,
                                                           float* __restrict__ part_sums, int* __restrict__ part_counts,
                                                           float* __restrict__ part_inertia, int* __restrict__ part_changed,
                                                           float* __restrict__ mind2) {
    constexpr int D = 256 * R;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* cs = sm;                    // [K][D] centres
    float* acc = sm + K * D;           // [K][D] sums of this workgroup's rows
    __shared__ float cn[KM_KMAX];
    __shared__ int cnt[KM_KMAX];
    const int lane = threadIdx.x;
    for (int k0 = 0; k0 < K; k0 += 4) {               // four centres' fragments in flight per round trip
        f32x4 c[4][R];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r)
                c[u][r] = (k0 + u < K) ? *(const f32x4*)(C + (size_t)(k0 + u) * D + 256 * r + 4 * lane) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (k0 + u < K) {
                const int k = k0 + u;
                float s2 = 0.f;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    *(f32x4*)(cs + k * D + 256 * r + 4 * lane) = c[u][r];
                    *(f32x4*)(acc + k * D + 256 * r + 4 * lane) = f32x4{0.f, 0.f, 0.f, 0.f};
                    s2 += c[u][r][0] * c[u][r][0] + c[u][r][1] * c[u][r][1] + c[u][r][2] * c[u][r][2] + c[u][r][3] * c[u][r][3];
                }
                s2 = km_sum(s2);
                if (lane == 0) { cn[k] = s2; cnt[k] = 0; }
            }
        }
    }
    __syncthreads();
    const int beg = blockIdx.x * rows_per_wg, end = min(N, beg + rows_per_wg);
    float inertia = 0.f;
    int changed = 0;
    f32x4 nx[KM_ROWS][R];
    auto fetch = [&](int i0) {
#pragma unroll
        for (int q = 0; q < KM_ROWS; ++q)
#pragma unroll
            for (int r = 0; r < R; ++r)
                nx[q][r] = (i0 + q < end) ? *(const f32x4*)(X + (size_t)(i0 + q) * D + 256 * r + 4 * lane) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    fetch(beg);
    for (int i0 = beg; i0 < end; i0 += KM_ROWS) {
        f32x4 x[KM_ROWS][R];
#pragma unroll
        for (int q = 0; q < KM_ROWS; ++q)
#pragma unroll
            for (int r = 0; r < R; ++r) x[q][r] = nx[q][r];
        if (i0 + KM_ROWS < end) fetch(i0 + KM_ROWS);        // the next four rows fly under this step's arithmetic
        float best[KM_ROWS], xx[KM_ROWS];
        int arg[KM_ROWS];
#pragma unroll
        for (int q = 0; q < KM_ROWS; ++q) {
            float s2 = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) s2 += x[q][r][0] * x[q][r][0] + x[q][r][1] * x[q][r][1] + x[q][r][2] * x[q][r][2] + x[q][r][3] * x[q][r][3];
            xx[q] = km_sum(s2);
            best[q] = INFINITY;
            arg[q] = 0;
        }
#pragma unroll
        for (int k = 0; k < KM_KMAX; ++k) {
            if (k < K) {                                   // wave-uniform
                float dot[KM_ROWS];
#pragma unroll
                for (int q = 0; q < KM_ROWS; ++q) dot[q] = 0.f;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const f32x4 c = *(const f32x4*)(cs + k * D + 256 * r + 4 * lane);
#pragma unroll
                    for (int q = 0; q < KM_ROWS; ++q)
                        dot[q] += x[q][r][0] * c[0] + x[q][r][1] * c[1] + x[q][r][2] * c[2] + x[q][r][3] * c[3];
                }
                const float ck = cn[k];
#pragma unroll
                for (int q = 0; q < KM_ROWS; ++q) {
                    const float dist = ck - 2.f * km_sum(dot[q]);
                    if (dist < best[q]) { best[q] = dist; arg[q] = k; }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < KM_ROWS; ++q) {
            if (i0 + q < end) {                            // wave-uniform
                const int a = arg[q];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float* p = acc + a * D + 256 * r + 4 * lane;
                    const f32x4 o = *(const f32x4*)p;
                    *(f32x4*)p = f32x4{o[0] + x[q][r][0], o[1] + x[q][r][1], o[2] + x[q][r][2], o[3] + x[q][r][3]};
                }
                if (lane == 0) {
                    cnt[a] += 1;
                    changed += (labels[i0 + q] != a);
                    labels[i0 + q] = a;
                    const float dq = fmaxf(xx[q] + best[q], 0.f);
                    inertia += dq;
                    if (mind2) mind2[i0 + q] = dq;
                }
            }
        }
    }
    __syncthreads();
    float* ps = part_sums + (size_t)blockIdx.x * K * D;
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int r = 0; r < R; ++r) *(f32x4*)(ps + k * D + 256 * r + 4 * lane) = *(const f32x4*)(acc + k * D + 256 * r + 4 * lane);
    if (lane < K) part_counts[blockIdx.x * K + lane] = cnt[lane];
    if (lane == 0) {
        part_inertia[blockIdx.x] = inertia;
        part_changed[blockIdx.x] = changed;
    }
}

// First level of the fixed-order reduction: fold f adds up the partials of workgroups [f*per, (f+1)*per) in order
// (one thread per (cluster, coordinate) and fold: ~160k threads, so the 20 MB of partials stream instead of trickling
// through 5120 threads).
#define KM_FOLDS 32
__global__ void kmeans_fold_kernel(const float* __restrict__ part_sums, const int* __restrict__ part_counts,
                                   const float* __restrict__ part_inertia, const int* __restrict__ part_changed,
                                   int n_parts, int per, int K, int D, float* __restrict__ fold_sums,
                                   int* __restrict__ fold_counts, float* __restrict__ fold_inertia,
                                   int* __restrict__ fold_changed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    const int p0 = f * per, p1 = min(n_parts, p0 + per);
    const size_t stride = (size_t)K * D;
    if (idx < K * D) {
        float s = 0.f;
        for (int p = p0; p < p1; ++p) s += part_sums[p * stride + idx];
        fold_sums[f * stride + idx] = s;
        if (idx % D == 0) {
            const int k = idx / D;
            int n = 0;
            for (int p = p0; p < p1; ++p) n += part_counts[p * K + k];
            fold_counts[f * K + k] = n;
        }
    }
    if (idx == 0) {
        float in = 0.f;
        int ch = 0;
        for (int p = p0; p < p1; ++p) { in += part_inertia[p]; ch += part_changed[p]; }
        fold_inertia[f] = in;
        fold_changed[f] = ch;
    }
}

// one thread per (cluster, coordinate): folds added in order; stats[0] = sum of squared centre shifts,
// stats[1] = inertia of the assignment just made, stats[2] = number of rows that changed label, stats[3 + k] = rows of
// cluster k (all as float, so that one small copy tells the host everything it steers by)
__global__ void kmeans_update_kernel(const float* __restrict__ part_sums, const int* __restrict__ part_counts,
                                     const float* __restrict__ part_inertia, const int* __restrict__ part_changed,
                                     int n_parts, int K, int D, float* __restrict__ C, int* __restrict__ counts,
                                     float* __restrict__ shift2_part, float* __restrict__ stats, int update) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < K * D) {
        const int k = idx / D;
        float s = 0.f;
        int n = 0;
        const size_t stride = (size_t)K * D;
        int p = 0;
        for (; p + 8 <= n_parts; p += 8) {           // eight loads in flight, added in workgroup order
            float v[8];
            int c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { v[u] = part_sums[(p + u) * stride + idx]; c[u] = part_counts[(p + u) * K + k]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += v[u]; n += c[u]; }
        }
        for (; p < n_parts; ++p) {
            s += part_sums[p * stride + idx];
            n += part_counts[p * K + k];
        }
        const float old = C[idx];
        const float c = (n > 0 && update) ? s / (float)n : old;
        if (update) C[idx] = c;
        shift2_part[idx] = (c - old) * (c - old);
        if (idx % D == 0) { counts[k] = n; stats[3 + k] = (float)n; }
    }
    if (idx == 0) {
        float in = 0.f;
        int ch = 0;
        for (int p = 0; p < n_parts; ++p) { in += part_inertia[p]; ch += part_changed[p]; }
        stats[1] = in;
        stats[2] = (float)ch;
    }
}
__global__ void kmeans_shift_kernel(const float* __restrict__ shift2_part, int n, float* __restrict__ stats) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += shift2_part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) stats[0] = red[0];
}

extern "C" int murcl_kmeans_parts(int N) {
    int parts = (N + 4 * KM_ROWS - 1) / (4 * KM_ROWS);          // at least 16 rows per wave
    if (parts > 1024) parts = 1024;                             // four single-wave workgroups per CU
    return parts < 1 ? 1 : parts;
}

// C-ABI: see include/murcl_amd.h
extern "C" int murcl_kmeans_step(const float* X, int N, int d, int K, float* centers, int* labels, int* counts,
                                 float* stats, float* mind2, int update, void* workspace, hipStream_t stream) {
    if (N <= 0 || K < 1 || K > KM_KMAX || (d != 256 && d != 512 && d != 1024)) return -1;
    const int parts = murcl_kmeans_parts(N);
    int rows = (N + parts - 1) / parts;
    rows = ((rows + KM_ROWS - 1) / KM_ROWS) * KM_ROWS;
    float* part_sums = (float*)workspace;
    int* part_counts = (int*)(part_sums + (size_t)parts * K * d);
    float* part_inertia = (float*)(part_counts + parts * K);
    int* part_changed = (int*)(part_inertia + parts);
    float* shift2 = (float*)(part_changed + parts);
    float* fold_sums = shift2 + (size_t)K * d;
    int* fold_counts = (int*)(fold_sums + (size_t)KM_FOLDS * K * d);
    float* fold_inertia = (float*)(fold_counts + KM_FOLDS * K);
    int* fold_changed = (int*)(fold_inertia + KM_FOLDS);
    const int lds = 2 * K * d * 4;
#define KM_LAUNCH(R)                                                                                                 \
    {                                                                                                                \
        auto k = kmeans_assign_kernel<R>;                                                                            \
        static int set_for = 0;                                                                                      \
        if (lds > set_for) {                                                                                         \
            (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KM_KMAX * 256 * R * 4); \
            set_for = 2 * KM_KMAX * 256 * R * 4;                                                                     \
        }                                                                                                            \
        hipLaunchKernelGGL(k, dim3(parts), dim3(64), lds, stream, X, centers, N, K, rows, labels, part_sums,         \
                           part_counts, part_inertia, part_changed, mind2);                                          \
    }
    if (d == 256) KM_LAUNCH(1) else if (d == 512) KM_LAUNCH(2) else KM_LAUNCH(4)
#undef KM_LAUNCH
    int rc = MURCL_CHECK_LAUNCH();
    if (rc) return rc;
    const int per = (parts + KM_FOLDS - 1) / KM_FOLDS;
    hipLaunchKernelGGL(kmeans_fold_kernel, dim3((K * d + 255) / 256, KM_FOLDS), dim3(256), 0, stream, part_sums, part_counts,
                       part_inertia, part_changed, parts, per, K, d, fold_sums, fold_counts, fold_inertia, fold_changed);
    rc = MURCL_CHECK_LAUNCH();
    if (rc) return rc;
    hipLaunchKernelGGL(kmeans_update_kernel, dim3((K * d + 255) / 256), dim3(256), 0, stream, (const float*)fold_sums,
                       (const int*)fold_counts, (const float*)fold_inertia, (const int*)fold_changed, KM_FOLDS, K, d,
                       centers, counts, shift2, stats, update);
    rc = MURCL_CHECK_LAUNCH();
    if (rc) return rc;
    hipLaunchKernelGGL(kmeans_shift_kernel, dim3(1), dim3(256), 0, stream, shift2, K * d, stats);
    return MURCL_CHECK_LAUNCH();
}

extern "C" long murcl_kmeans_workspace_bytes(int N, int d, int K) {
    const long parts = murcl_kmeans_parts(N);
    return (parts * K * d + parts * K + 2 * parts + (long)K * d + (long)KM_FOLDS * (K * d + K + 2) + 16) * 4;
}
