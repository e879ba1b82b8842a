// K4 / K5: CLAM-SB pieces around the shared GEMM kernels (models/clam.py:55-60,103-132,139-170).
//
//   U = h [Wa;Wb]^T + [ba;bb]                              gemm_nt (one pass over h for both gate branches)
//   s_n = sum_d tanh(U[n,d]) * sigmoid(U[n,D+d]) * wc[d] + bc      gated_score_kernel
//   A = softmax_N(s),  M = A.h                             softmax_rows_kernel + weighted_rowsum (dsmil.hip)
//   instance eval: top-k / bottom-k ids of A (lowest index wins ties), CE on 2-way instance logits
#include "common.h"

// ---------------------------------------------------------------- gated attention score (one wave per row)
template <typename T>
__global__ __launch_bounds__(256) void gated_score_fwd_kernel(const T* __restrict__ U, const float* __restrict__ wc,
                                                              const float* __restrict__ bc,
                                                              const T* __restrict__ keep_a, const T* __restrict__ keep_b,
                                                              float* __restrict__ s, long rows, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const T* u = U + row * 2 * D;
    float acc = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
        const f32x4 ua = load4<T>(u + d), ub = load4<T>(u + D + d);
        f32x4 ka = f32x4{1.f, 1.f, 1.f, 1.f}, kb = ka;
        if (keep_a) { ka = load4<T>(keep_a + row * D + d); kb = load4<T>(keep_b + row * D + d); }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += (tanhf(ua[e]) * ka[e]) * (sigmoidf_(ub[e]) * kb[e]) * wc[d + e];
    }
    acc = wave_sum(acc);
    if (lane == 0) s[row] = acc + bc[0];
}
// dU[n,d] = ds_n wc_d g (1-a^2) ka kb ; dU[n,D+d] = ds_n wc_d a g (1-g) ka kb ; dwc_d += ds_n a g ka kb ; dbc += ds_n
template <typename T>
__global__ __launch_bounds__(256) void gated_score_bwd_kernel(const T* __restrict__ U, const float* __restrict__ wc,
                                                              const T* __restrict__ keep_a, const T* __restrict__ keep_b,
                                                              const float* __restrict__ ds, T* __restrict__ dU,
                                                              float* __restrict__ dwc, float* __restrict__ dbc,
                                                              long rows, int D, int rows_per_block) {
    // block: 256 threads = 64 column groups of 4 (D <= 256) ... generic: thread owns columns d0 + 1024*k
    const int tid = threadIdx.x;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float dbc_acc = 0.f;
    for (int d = tid * 4; d < D; d += 1024) {
        f32x4 wacc = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 w = *(const f32x4*)(wc + d);
        for (long n = r0; n < r1; ++n) {
            const T* u = U + n * 2 * D;
            const f32x4 ua = load4<T>(u + d), ub = load4<T>(u + D + d);
            f32x4 ka = f32x4{1.f, 1.f, 1.f, 1.f}, kb = ka;
            if (keep_a) { ka = load4<T>(keep_a + n * D + d); kb = load4<T>(keep_b + n * D + d); }
            const float dsn = ds[n];
            f32x4 da, db;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = tanhf(ua[e]), g = sigmoidf_(ub[e]);
                const float k = ka[e] * kb[e];
                da[e] = dsn * w[e] * g * (1.f - a * a) * k;
                db[e] = dsn * w[e] * a * g * (1.f - g) * k;
                wacc[e] += dsn * a * g * k;
            }
            store4<T>(dU + n * 2 * D + d, da);
            store4<T>(dU + n * 2 * D + D + d, db);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dwc + d + e, wacc[e]);
    }
    if (tid == 0) {
        for (long n = r0; n < r1; ++n) dbc_acc += ds[n];
        atomicAdd(dbc, dbc_acc);
    }
}
extern "C" int murcl_gated_score_fwd(const void* U, const float* wc, const float* bc, const void* keep_a,
                                     const void* keep_b, float* s, long rows, int D, int dtype, hipStream_t st) {
    if (rows <= 0) return 0;
    if (D % 4) return -1;
    dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == MURCL_DTYPE_F32)
        hipLaunchKernelGGL(gated_score_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)U, wc, bc, (const float*)keep_a, (const float*)keep_b, s, rows, D);
    else if (dtype == MURCL_DTYPE_BF16)
        hipLaunchKernelGGL(gated_score_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)U, wc, bc, (const bf16_t*)keep_a, (const bf16_t*)keep_b, s, rows, D);
    else return -1;
    return MURCL_CHECK_LAUNCH();
}
extern "C" int murcl_gated_score_bwd(const void* U, const float* wc, const void* keep_a, const void* keep_b,
                                     const float* ds, void* dU, float* dwc, float* dbc, long rows, int D, int dtype,
                                     hipStream_t st) {
    if (rows <= 0) return 0;
    if (D % 4) return -1;
    hipError_t e = hipMemsetAsync(dwc, 0, (size_t)D * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(dbc, 0, 4, st);
    if (e != hipSuccess) return (int)e;
    const int rpb = 64;
    dim3 grid((unsigned)((rows + rpb - 1) / rpb));
    if (dtype == MURCL_DTYPE_F32)
        hipLaunchKernelGGL(gated_score_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)U, wc, (const float*)keep_a, (const float*)keep_b, ds, (float*)dU, dwc, dbc, rows, D, rpb);
    else if (dtype == MURCL_DTYPE_BF16)
        hipLaunchKernelGGL(gated_score_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)U, wc, (const bf16_t*)keep_a, (const bf16_t*)keep_b, ds, (bf16_t*)dU, dwc, dbc, rows, D, rpb);
    else return -1;
    return MURCL_CHECK_LAUNCH();
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
__global__ __launch_bounds__(256) void topk_ids_kernel(const float* __restrict__ A, int N, int k, int* __restrict__ ids) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    __shared__ int taken[64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* a = A + (size_t)b * N;
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
            if (tid == 0) { taken[r] = bi[0]; ids[(size_t)b * 2 * k + pass * k + r] = bi[0]; }
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
                                               const float* __restrict__ g, int d) {
    const long r = blockIdx.x;
    T* o = dst + rows[r] * d;
    const T* m = h + rows[r] * d;
    for (int k = threadIdx.x; k < d; k += blockDim.x)
        if (to_f<T>(m[k]) > 0.f) o[k] = from_f<T>(to_f<T>(o[k]) + g[r * d + k]);
}
extern "C" int murcl_take_rows(const void* src, const long* rows, float* out, int R, int d, int dtype, hipStream_t st) {
    if (R <= 0) return 0;
    if (dtype == MURCL_DTYPE_F32) hipLaunchKernelGGL(take_rows_kernel<float>, dim3(R), dim3(256), 0, st, (const float*)src, rows, out, d);
    else if (dtype == MURCL_DTYPE_BF16) hipLaunchKernelGGL(take_rows_kernel<bf16_t>, dim3(R), dim3(256), 0, st, (const bf16_t*)src, rows, out, d);
    else return -1;
    return MURCL_CHECK_LAUNCH();
}
extern "C" int murcl_scatter_add_rows_masked(void* dst, const void* h, const long* rows, const float* g, int R, int d,
                                             int dtype, hipStream_t st) {
    if (R <= 0) return 0;
    if (dtype == MURCL_DTYPE_F32) hipLaunchKernelGGL(scatter_add_rows_masked_kernel<float>, dim3(R), dim3(256), 0, st, (float*)dst, (const float*)h, rows, g, d);
    else if (dtype == MURCL_DTYPE_BF16) hipLaunchKernelGGL(scatter_add_rows_masked_kernel<bf16_t>, dim3(R), dim3(256), 0, st, (bf16_t*)dst, (const bf16_t*)h, rows, g, d);
    else return -1;
    return MURCL_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- mean cross-entropy per group of G consecutive rows
// (C <= 32 logits per row).  One workgroup per group:  loss[g] = mean_r [ lse(logits_r) - logits_r[target_r] ];
// dlogits = (softmax - onehot) / G ; pred = argmax (first max)
__global__ __launch_bounds__(64) void ce_fwd_bwd_kernel(const float* __restrict__ logits, const long* __restrict__ targets,
                                                        int G, int C, float* __restrict__ loss,
                                                        float* __restrict__ dlogits, long* __restrict__ preds) {
    const int grp = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    for (int r = tid; r < G; r += 64) {
        const size_t row = (size_t)grp * G + r;
        const float* x = logits + row * C;
        float mx = x[0];
        int am = 0;
        for (int c = 1; c < C; ++c) if (x[c] > mx) { mx = x[c]; am = c; }
        float sum = 0.f;
        for (int c = 0; c < C; ++c) sum += expf(x[c] - mx);
        const float lse = mx + logf(sum);
        const int t = (int)targets[row];
        acc += lse - x[t];
        if (dlogits)
            for (int c = 0; c < C; ++c) dlogits[row * C + c] = (expf(x[c] - lse) - (c == t ? 1.f : 0.f)) / (float)G;
        if (preds) preds[row] = am;
    }
    acc = wave_sum(acc);
    if (tid == 0) loss[grp] = acc / (float)G;
}
extern "C" int murcl_cross_entropy(const float* logits, const long* targets, int R, int C, float* loss, float* dlogits,
                                   long* preds, int group, hipStream_t st) {
    if (R <= 0 || C <= 0 || group <= 0 || R % group) return -1;
    hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3(R / group), dim3(64), 0, st, logits, targets, group, C, loss, dlogits, preds);
    return MURCL_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- y *= k (dropout keep-mask multiply), y = dy * k
template <typename T>
__global__ void mul_kernel(const T* __restrict__ x, const T* __restrict__ k, T* __restrict__ y, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = from_f<T>(to_f<T>(x[i]) * to_f<T>(k[i]));
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
