// K6: DSMIL aggregator pieces (models/dsmil.py:64-81) around the shared GEMM kernels.
//
//   classes = X Wc^T + bc                       gemm_nt (together with Q: one pass over X)
//   m[c]    = argmax_n classes[n,c]             dsmil_argmax_kernel      (lowest index wins ties)
//   q_max   = Q[m]                              dsmil_gather_rows_kernel
//   A       = softmax_n(Q q_max^T / sqrt(128))  dsmil_attn_kernel        (reads only Q: N x 128)
//   Z       = A^T X                             weighted_rowsum_kernel   (one streaming pass over X)
//   bag     = Z Wv^T + bv                       skinny gemm_nt           (= A^T (X Wv^T + bv): columns of A sum to 1,
//                                                                          dropout_v = 0, dsmil.py:53,118)
// Backward adds rows_dot_kernel (dA = X dZ^T) and dsmil_attn_bwd_kernel.
#include "common.h"

#define DS_Q 128

// per (bag, class): first index of the maximum of scores[b, :, c]   (scores [B,N,ld] f32, classes in columns 0..C-1)
__global__ __launch_bounds__(256) void dsmil_argmax_kernel(const float* __restrict__ scores, int N, int ld, int C,
                                                           int* __restrict__ m_out, float* __restrict__ max_out) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    const int b = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
    const float* s = scores + (size_t)b * N * ld + c;
    float best = -INFINITY;
    int idx = tid < N ? tid : 0x7fffffff;                    // a column of -inf: its first row, as torch.max
    int n = tid;
    for (; n + 7 * 256 < N; n += 8 * 256) {                  // eight loads in flight per thread (one after the other they were N/256
        float v[8];                                          // serial round trips: 16 us at N = 8192)
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = s[(size_t)(n + u * 256) * ld];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (v[u] > best) { best = v[u]; idx = n + u * 256; }     // ascending n: a later equal value never replaces an earlier one
    }
    for (; n < N; n += 256) {
        const float v = s[(size_t)n * ld];
        if (v > best) { best = v; idx = n; }
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
    if (tid == 0) {
        m_out[b * C + c] = bi[0];
        if (max_out) max_out[b * C + c] = bv[0];
    }
}
extern "C" int murcl_dsmil_argmax(const float* scores, int B, int N, int ld, int C, int* m_out, hipStream_t s) {
    if (B <= 0 || C <= 0) return 0;
    hipLaunchKernelGGL(dsmil_argmax_kernel, dim3(B, C), dim3(256), 0, s, scores, N, ld, C, m_out, (float*)nullptr);
    return MURCL_CHECK_LAUNCH();
}
// ... and the maxima themselves, max_out [B,C] = scores[b, m[b,c], c]: the max-instance class scores of train_RLMIL.py:516
// (`torch.max(outputs_ins, 0)`), which the kernel has in hand when it has found the critical instances
extern "C" int murcl_dsmil_argmax_max(const float* scores, int B, int N, int ld, int C, int* m_out, float* max_out, hipStream_t s) {
    if (B <= 0 || C <= 0) return 0;
    hipLaunchKernelGGL(dsmil_argmax_kernel, dim3(B, C), dim3(256), 0, s, scores, N, ld, C, m_out, max_out);
    return MURCL_CHECK_LAUNCH();
}

// out[b*C + c, :] = src[b, m[b,c], col0 : col0+width]   (src [B,N,ld])
template <typename T>
__global__ void dsmil_gather_rows_kernel(const T* __restrict__ src, const int* __restrict__ m, int N, int ld, int col0,
                                         int width, T* __restrict__ out, int C) {
    const int r = blockIdx.x, b = r / C;
    const T* s = src + ((size_t)b * N + m[r]) * ld + col0;
    for (int k = threadIdx.x; k < width; k += blockDim.x) out[(size_t)r * width + k] = s[k];
}
extern "C" int murcl_gather_rows(const void* src, const int* m, int B, int C, int N, int ld, int col0, int width,
                                 void* out, int dtype, hipStream_t s) {
    if (B <= 0 || C <= 0) return 0;
    if (dtype == MURCL_DTYPE_F32) {
        hipLaunchKernelGGL(dsmil_gather_rows_kernel<float>, dim3(B * C, 1), dim3(256), 0, s, (const float*)src, m, N, ld, col0, width, (float*)out, C);
    } else if (dtype == MURCL_DTYPE_BF16) {
        hipLaunchKernelGGL(dsmil_gather_rows_kernel<bf16_t>, dim3(B * C, 1), dim3(256), 0, s, (const bf16_t*)src, m, N, ld, col0, width, (bf16_t*)out, C);
    } else {
        return -1;
    }
    return MURCL_CHECK_LAUNCH();
}

// A[b,n,c] = softmax_n( Q[b,n,:] . qmax[b,c,:] * scale ); Q rows at stride ldq, C <= 4.
// Two launches: raw scores with the rows spread over the whole chip (32 threads per row, 16-byte loads), then the
// soft-max over n per (bag, class) on the 4-byte scores (one workgroup per (bag, class) is plenty for N*4 bytes).
#define DS_RPB 64                  // rows per workgroup of the row-parallel kernels
__global__ __launch_bounds__(256) void dsmil_scores_kernel(const float* __restrict__ Q, int ldq, int qcol0,
                                                           const float* __restrict__ qmax, int N, int C, float scale,
                                                           float* __restrict__ A) {
    __shared__ float qm[4 * DS_Q];
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int k = tid; k < C * DS_Q; k += 256) qm[k] = qmax[(size_t)b * C * DS_Q + k];
    __syncthreads();
    const int kq = (tid & 31) * 4, nl = tid >> 5;               // 32 threads per row, 8 rows per pass
    const int r0 = blockIdx.x * DS_RPB, r1 = min(N, r0 + DS_RPB);
    const float* q = Q + (size_t)b * N * ldq + qcol0;
    float* a = A + (size_t)b * N * C;
    for (int n = r0 + nl; n < r1; n += 8) {
        const f32x4 v = *(const f32x4*)(q + (size_t)n * ldq + kq);
        float s[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s[c] = 0.f;
            if (c < C) {
                const f32x4 w = *(const f32x4*)&qm[c * DS_Q + kq];
                s[c] = v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < C) {
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) s[c] += __shfl_xor(s[c], o, 64);
                if ((tid & 31) == 0) a[(size_t)n * C + c] = s[c] * scale;
            }
    }
}
// in-place soft-max over n of A[b, :, c] (stride C)
__global__ __launch_bounds__(256) void dsmil_softmax_kernel(float* __restrict__ A, int N, int C) {
    __shared__ float red[256];
    const int b = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
    float* a = A + (size_t)b * N * C + c;
    float mx = -INFINITY;
    for (int n = tid; n < N; n += 256) mx = fmaxf(mx, a[(size_t)n * C]);
    red[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int n = tid; n < N; n += 256) {
        const float e = expf(a[(size_t)n * C] - mx);
        a[(size_t)n * C] = e;
        sum += e;
    }
    red[tid] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float inv = 1.f / red[0];
    for (int n = tid; n < N; n += 256) a[(size_t)n * C] *= inv;
}
extern "C" int murcl_dsmil_attn(const float* Q, int ldq, int qcol0, const float* qmax, int B, int N, int C, float* A,
                                hipStream_t s) {
    if (B <= 0) return 0;
    if (C > 4) return -1;
    hipLaunchKernelGGL(dsmil_scores_kernel, dim3((N + DS_RPB - 1) / DS_RPB, B), dim3(256), 0, s, Q, ldq, qcol0, qmax, N, C,
                       1.0f / sqrtf((float)DS_Q), A);
    int rc = MURCL_CHECK_LAUNCH();
    if (rc) return rc;
    hipLaunchKernelGGL(dsmil_softmax_kernel, dim3(B, C), dim3(256), 0, s, A, N, C);
    return MURCL_CHECK_LAUNCH();
}

// In-place soft-max over n of raw scores S[b, :, c] that are already scaled (the reassociated K6 path: S = X . v with
// v = qmax Wq / sqrt(128), so that the [B*N, 128] queries are never formed - see functional.DSMILFn).
extern "C" int murcl_dsmil_softmax(float* S, int B, int N, int C, hipStream_t s) {
    if (B <= 0 || N <= 0) return 0;
    if (C <= 0 || C > 4) return -1;
    hipLaunchKernelGGL(dsmil_softmax_kernel, dim3(B, C), dim3(256), 0, s, S, N, C);
    return MURCL_CHECK_LAUNCH();
}

#ifndef WR_UR
#define WR_UR 4
#endif
#ifndef WR_UR_BF16
#define WR_UR_BF16 4
#endif
#ifndef WR_WGS
#define WR_WGS 512                // workgroups per launch, two per CU (each adds its partial sums atomically: WR_WGS / B adders per address).
#endif                            // 16 x 8192 x 1024: 256 -> 102 us, 512 -> 82 us, 1024 -> 116 us, 2048 -> 180 us (bf16); 64 x 4096 x 512: 85 / 54 / 61 / 70 us
// Z[b,c,:] = sum_n A[b,n,c] X[b,n,:]      (one streaming pass over X; C <= 4)
// grid (B, row splits).  A thread owns 8 consecutive columns (16-byte loads for bf16), G = d/8 column groups and
// 256/G row lanes per workgroup; the row lanes meet in LDS and the workgroup adds its partial sums atomically.
template <typename T>
__global__ __launch_bounds__(256) void weighted_rowsum_kernel(const T* __restrict__ X, const float* __restrict__ A,
                                                              int N, int d, int C, int rows_per_block,
                                                              float* __restrict__ Z) {
    __shared__ float red[256][8 + 1];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(N, r0 + rows_per_block);
    const T* x = X + (size_t)b * N * d;
    const float* a = A + (size_t)b * N * C;
    const int G = min(d >> 3, 256), RL = 256 / G, cg = tid % G, rl = tid / G;
    for (int c0 = 8 * cg; c0 < d; c0 += 8 * G) {                // one pass unless d > 2048
        float acc[4][8];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[c][e] = 0.f;
        if (rl < RL) {
            int n = r0 + rl;
            constexpr int UR = sizeof(T) == 2 ? WR_UR_BF16 : WR_UR;   // rows in flight per thread
            for (; n + (UR - 1) * RL < r1; n += UR * RL) {
                float v[UR][8], w[UR][4];
#pragma unroll
                for (int u = 0; u < UR; ++u) load8<T>(x + (size_t)(n + u * RL) * d + c0, v[u]);
#pragma unroll
                for (int u = 0; u < UR; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c) w[u][c] = (c < C) ? a[(size_t)(n + u * RL) * C + c] : 0.f;
#pragma unroll
                for (int u = 0; u < UR; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < C) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[c][e] += w[u][c] * v[u][e];
                        }
            }
            for (; n < r1; n += RL) {
                float v[8];
                load8<T>(x + (size_t)n * d + c0, v);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < C) {
                        const float w = a[(size_t)n * C + c];
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[c][e] += w * v[e];
                    }
            }
        }
        for (int c = 0; c < C; ++c) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) red[tid][e] = acc[c][e];
            __syncthreads();
            if (rl == 0) {
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = 0.f;
                for (int r = 0; r < RL; ++r)
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] += red[r * G + cg][e];
#pragma unroll
                for (int e = 0; e < 8; ++e) atomicAdd(Z + ((size_t)b * C + c) * d + c0 + e, t[e]);
            }
        }
    }
}
static int weighted_rowsum_launch(const void* X, const float* A, float* Z, int B, int N, int d, int C, int dtype, bool zero,
                                  hipStream_t s) {
    if (B <= 0) return 0;
    if (C > 4 || d % 8) return -1;
    if (zero) {
        hipError_t e = hipMemsetAsync(Z, 0, (size_t)B * C * d * 4, s);
        if (e != hipSuccess) return (int)e;
    }
    int splits = (WR_WGS + B - 1) / B;
    if (splits > (N + 63) / 64) splits = (N + 63) / 64;
    if (splits < 1) splits = 1;
    const int rpb = (N + splits - 1) / splits;
    dim3 grid(B, (N + rpb - 1) / rpb);
    if (dtype == MURCL_DTYPE_F32)
        hipLaunchKernelGGL(weighted_rowsum_kernel<float>, grid, dim3(256), 0, s, (const float*)X, A, N, d, C, rpb, Z);
    else if (dtype == MURCL_DTYPE_BF16)
        hipLaunchKernelGGL(weighted_rowsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)X, A, N, d, C, rpb, Z);
    else
        return -1;
    return MURCL_CHECK_LAUNCH();
}
extern "C" int murcl_weighted_rowsum(const void* X, const float* A, float* Z, int B, int N, int d, int C, int dtype,
                                     hipStream_t s) {
    return weighted_rowsum_launch(X, A, Z, B, N, d, C, dtype, true, s);
}
// the same ADDED to Z (the caller - e.g. murcl_softmax_rows_parts - has zeroed it: no fill launch in front of the pass)
extern "C" int murcl_weighted_rowsum_acc(const void* X, const float* A, float* Z, int B, int N, int d, int C, int dtype,
                                         hipStream_t s) {
    return weighted_rowsum_launch(X, A, Z, B, N, d, C, dtype, false, s);
}

// out[b,n,c] = X[b,n,:] . V[b,c,:]      (dA = X dZ^T; C <= 4).  A wave walks `RPW` rows; a lane owns 8 consecutive
// columns per 512-column step (16-byte loads for bf16) and keeps its slice of V in registers when d <= 512.
#ifndef RD_RPW
#define RD_RPW 4                // rows per wave = rows in flight: one round of loads per wave, the hardware scheduler does the rest
#endif                          // (16 rows in four serial rounds: 76 us at 64 x 4096 x 512 bf16; 4 rows: 66 us; 64: 98 us; 256: 175 us)
#ifndef RD_UR_BF16
#define RD_UR_BF16 4             // rows in flight per wave for bf16 rows (half the bytes of an f32 row)
#endif
#ifndef RD_WAVE_SUM
#define RD_WAVE_SUM wave_sum_valu      // (ds_bpermute chains of __shfl_xor: rows_dot 117.5 -> 106.4 us f32, 61.9 -> 49.9 us bf16 at the C5 share)
#endif
template <typename T>
__global__ __launch_bounds__(256) void rows_dot_kernel(const T* __restrict__ X, const float* __restrict__ V, int N, int d,
                                                       int C, float* __restrict__ out, long rows_total,
                                                       const float* __restrict__ bias) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row0 = ((long)blockIdx.x * 4 + wave) * RD_RPW;
    if (row0 >= rows_total) return;
    const long row1 = min(rows_total, row0 + RD_RPW);
    constexpr int UR = sizeof(T) == 2 ? RD_UR_BF16 : 4;         // rows in flight per wave
    for (long rb = row0; rb < row1; rb += UR) {
        float acc[UR][4];
#pragma unroll
        for (int u = 0; u < UR; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[u][c] = 0.f;
        for (int k = lane * 8; k < d; k += 512) {
            float xv[UR][8];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const long row = min(rb + u, row1 - 1);
                load8<T>(X + row * d + k, xv[u]);
            }
            const long bag0 = rb / N, bag3 = min(rb + UR - 1, row1 - 1) / N;
            if (bag0 == bag3) {
                // the four rows belong to one bag (always, when N % 4 == 0): its slice of V is loaded once for all of them
                // (per row it was as many load instructions again as X itself: the kernel was load-issue-bound)
                const float* v = V + (size_t)bag0 * C * d;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < C) {
                        float w[8];
                        load8<float>(v + (size_t)c * d + k, w);
#pragma unroll
                        for (int u = 0; u < UR; ++u)
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[u][c] += xv[u][e] * w[e];
                    }
            } else {
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    const long row = min(rb + u, row1 - 1);
                    const float* v = V + (size_t)(row / N) * C * d;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < C) {
                            float w[8];
                            load8<float>(v + (size_t)c * d + k, w);
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[u][c] += xv[u][e] * w[e];
                        }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            if (rb + u >= row1) break;
            for (int c = 0; c < C; ++c) {
                const float s = RD_WAVE_SUM(acc[u][c]);
                if (lane == 0) out[(rb + u) * C + c] = bias ? s + bias[c] : s;
            }
        }
    }
}
#ifndef RD_STREAM
#define RD_STREAM 1                // shapes the K6 stream passes cover take their load schedule (dsmil_stream_kernel<T, 3>: V in registers per wave,
#endif                             // the next four rows requested before the current four are consumed) - same sums in the same order, bit for bit
static int rows_dot_stream_launch(const void* X, const float* V, const float* bias, float* out, int B, int N, int d, int C, int dtype,
                                  hipStream_t s);
// bias (may be NULL) [C]: out[b,n,c] = X[b,n,:] . V[b,c,:] + bias[c] - the instance classifier's Linear (dsmil.py:9,15) in one launch
extern "C" int murcl_rows_dot_bias(const void* X, const float* V, const float* bias, float* out, int B, int N, int d, int C, int dtype,
                                   hipStream_t s) {
    if (B <= 0) return 0;
    if (C > 4 || d % 8) return -1;
    // (measured r04_m, isolated: d = 1024 f32 108 -> 98 us = 0.69 of 8 TB/s, bf16 53.5 -> 48.1 us = 0.70; d = 512 bf16 55 -> 57 us: not there)
    if (RD_STREAM && d > 512 && (dtype == MURCL_DTYPE_F32 || dtype == MURCL_DTYPE_BF16)) {
        const int rc = rows_dot_stream_launch(X, V, bias, out, B, N, d, C, dtype, s);
        if (rc != -2) return rc;
    }
    const long rows = (long)B * N;
    dim3 grid((unsigned)((rows + 4 * RD_RPW - 1) / (4 * RD_RPW)));
    if (dtype == MURCL_DTYPE_F32)
        hipLaunchKernelGGL(rows_dot_kernel<float>, grid, dim3(256), 0, s, (const float*)X, V, N, d, C, out, rows, bias);
    else if (dtype == MURCL_DTYPE_BF16)
        hipLaunchKernelGGL(rows_dot_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)X, V, N, d, C, out, rows, bias);
    else
        return -1;
    return MURCL_CHECK_LAUNCH();
}
extern "C" int murcl_rows_dot(const void* X, const float* V, float* out, int B, int N, int d, int C, int dtype,
                              hipStream_t s) {
    return murcl_rows_dot_bias(X, V, nullptr, out, B, N, d, C, dtype, s);
}

// rows_dot and a weighted row sum over the SAME pass of X (DSMIL backward: dA = X dZ^T needs every row of X, and so does
// dWc = dcls^T X - two sweeps of 537 MB at the C5 shape):
//   out[b,n,c] = X[b,n,:] . V[b,c,:]           and           part[w][c][:] = sum over wave w's rows of G[b,n,c] X[b,n,:]
// A wave walks `rows_per_wave` consecutive rows (inside one bag: rows_per_wave divides N), four rows in flight, a lane owns
// 8 consecutive columns per 512-column step; its weighted partial sums stay in registers and leave as one row of `part`
// ([n_waves][C][d]; the caller sums the rows - no float atomics).  d <= 1024, C <= 2 (register budget).
template <typename T>
__global__ __launch_bounds__(256) void rows_dot_wsum_kernel(const T* __restrict__ X, const float* __restrict__ V,
                                                            const float* __restrict__ G, int N, int d, int C,
                                                            int rows_per_wave, float* __restrict__ out,
                                                            float* __restrict__ part, long rows_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wid = (long)blockIdx.x * 4 + wave;
    const long row0 = wid * rows_per_wave;
    if (row0 >= rows_total) return;
    const long row1 = min(rows_total, row0 + rows_per_wave);
    const float* v = V + (size_t)(row0 / N) * C * d;
    float wsum[2][2][8];                                    // [class][512-column step][8 columns]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int e = 0; e < 8; ++e) wsum[c][st][e] = 0.f;
    float vr[2][2][8];                                      // this bag's V slice, loaded once for all of the wave's rows
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int k = st * 512 + lane * 8;
            if (c < C && k < d) load8<float>(v + (size_t)c * d + k, vr[c][st]);
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) vr[c][st][e] = 0.f;
        }
    for (long rb = row0; rb < row1; rb += 4) {
        float acc[4][2], xv[4][2][8], gw[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long row = min(rb + u, row1 - 1);
            const bool live = rb + u < row1;
#pragma unroll
            for (int c = 0; c < 2; ++c) { acc[u][c] = 0.f; gw[u][c] = (live && c < C) ? G[row * C + c] : 0.f; }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int k = st * 512 + lane * 8;
                if (k < d) load8<T>(X + row * d + k, xv[u][st]);
                else
#pragma unroll
                    for (int e = 0; e < 8; ++e) xv[u][st][e] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        acc[u][c] += xv[u][st][e] * vr[c][st][e];
                        wsum[c][st][e] += gw[u][c] * xv[u][st][e];
                    }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (rb + u >= row1) break;
            for (int c = 0; c < C; ++c) {
                const float sdot = RD_WAVE_SUM(acc[u][c]);
                if (lane == 0) out[(rb + u) * C + c] = sdot;
            }
        }
    }
    float* pr = part + (size_t)wid * C * d;
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int k = st * 512 + lane * 8;
            if (k < d) {
                *(f32x4*)(pr + (size_t)c * d + k) = f32x4{wsum[c][st][0], wsum[c][st][1], wsum[c][st][2], wsum[c][st][3]};
                *(f32x4*)(pr + (size_t)c * d + k + 4) = f32x4{wsum[c][st][4], wsum[c][st][5], wsum[c][st][6], wsum[c][st][7]};
            }
        }
}
// rows_per_wave (out): how many rows a wave takes = the number of `part` rows the caller must provide is ceil(B*N / it)
#ifndef RDW_MIN_WAVES
#define RDW_MIN_WAVES 2048        // 1024 -> 235 us, 2048 -> 138 us, 4096 -> 149 us, 8192 -> 165 us (16 x 8192 x 1024 bf16; every wave leaves a [C][d] partial row)
#endif
extern "C" int murcl_rows_dot_wsum_plan(int B, int N, int d, int C) {
    if (B <= 0 || N <= 0 || C < 1 || C > 2 || d % 8 || d > 1024) return 0;
    int rpw = 256;                                          // >= 4096 waves (a wave keeps four rows in flight and reduces between
    while (rpw > 4 && (N % rpw || (long)B * N / rpw < RDW_MIN_WAVES)) rpw >>= 1;       // loads: the chip needs them all resident)
    return (N % rpw == 0) ? rpw : 0;
}
extern "C" int murcl_rows_dot_wsum(const void* X, const float* V, const float* G, float* out, float* part, int B, int N,
                                   int d, int C, int dtype, hipStream_t s) {
    if (B <= 0) return 0;
    const int rpw = murcl_rows_dot_wsum_plan(B, N, d, C);
    if (!rpw) return -1;
    const long rows = (long)B * N, waves = rows / rpw;
    dim3 grid((unsigned)((waves + 3) / 4));
    if (dtype == MURCL_DTYPE_F32)
        hipLaunchKernelGGL(rows_dot_wsum_kernel<float>, grid, dim3(256), 0, s, (const float*)X, V, G, N, d, C, rpw, out, part, rows);
    else if (dtype == MURCL_DTYPE_BF16)
        hipLaunchKernelGGL(rows_dot_wsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)X, V, G, N, d, C, rpw, out, part, rows);
    else
        return -1;
    return MURCL_CHECK_LAUNCH();
}

// Soft-max backward over N per (bag, class) + the two small products that follow it:
//   dS = A * (dA - sum_n A dA);  dY[b,n, qcol0:qcol0+128] = sum_c dS[n,c] qmax[c,:] * scale;
//   dqmax[b,c,:] = sum_n dS[n,c] Q[b,n,:] * scale
// dots[b,c] = sum_n A[b,n,c] dA[b,n,c]
__global__ __launch_bounds__(256) void dsmil_attn_dots_kernel(const float* __restrict__ A, const float* __restrict__ dA,
                                                              int N, int C, float* __restrict__ dots) {
    __shared__ float red[256];
    const int b = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
    const float* a = A + (size_t)b * N * C + c;
    const float* da = dA + (size_t)b * N * C + c;
    float s = 0.f;
    for (int n = tid; n < N; n += 256) s += a[(size_t)n * C] * da[(size_t)n * C];
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) dots[b * C + c] = red[0];
}
// rows spread over the chip: dY rows and this workgroup's share of dqmax (N/DS_RPB atomic adders per address)
__global__ __launch_bounds__(256) void dsmil_attn_bwd_kernel(const float* __restrict__ A, const float* __restrict__ dA,
                                                             const float* __restrict__ Q, int ldq, int qcol0,
                                                             const float* __restrict__ qmax, const float* __restrict__ dots_g,
                                                             int N, int C, float scale,
                                                             float* __restrict__ dY, int ldy, float* __restrict__ dqmax) {
    __shared__ float qm[4 * DS_Q];
    __shared__ f32x4 red[8][32][4];
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int k = tid; k < C * DS_Q; k += 256) qm[k] = qmax[(size_t)b * C * DS_Q + k];
    float dots[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) dots[c] = c < C ? dots_g[b * C + c] : 0.f;
    __syncthreads();
    const float* a = A + (size_t)b * N * C;
    const float* da = dA + (size_t)b * N * C;
    const float* q = Q + (size_t)b * N * ldq + qcol0;
    float* dy = dY + (size_t)b * N * ldy + qcol0;
    const int kl = tid & 31, kq = kl * 4, nl = tid >> 5;        // 32 threads per row, 8 rows per pass
    const int r0 = blockIdx.x * DS_RPB, r1 = min(N, r0 + DS_RPB);
    f32x4 dqm[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) dqm[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int n = r0 + nl; n < r1; n += 8) {
        f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 qv = *(const f32x4*)(q + (size_t)n * ldq + kq);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < C) {
                const float ds = a[(size_t)n * C + c] * (da[(size_t)n * C + c] - dots[c]) * scale;
                g += ds * *(const f32x4*)&qm[c * DS_Q + kq];
                dqm[c] += ds * qv;
            }
        *(f32x4*)(dy + (size_t)n * ldy + kq) = g;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) red[nl][kl][c] = dqm[c];
    __syncthreads();
    if (nl == 0) {
        for (int c = 0; c < C; ++c) {
            f32x4 t = red[0][kl][c];
#pragma unroll
            for (int r = 1; r < 8; ++r) t += red[r][kl][c];
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(dqmax + ((size_t)b * C + c) * DS_Q + kq + e, t[e]);
        }
    }
}
extern "C" int murcl_dsmil_attn_bwd(const float* A, const float* dA, const float* Q, int ldq, int qcol0,
                                    const float* qmax, int B, int N, int C, float* dY, int ldy, float* dqmax,
                                    float* dots_ws, hipStream_t s) {
    if (B <= 0) return 0;
    if (C > 4 || !dots_ws) return -1;
    hipError_t e = hipMemsetAsync(dqmax, 0, (size_t)B * C * DS_Q * 4, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(dsmil_attn_dots_kernel, dim3(B, C), dim3(256), 0, s, A, dA, N, C, dots_ws);
    int rc = MURCL_CHECK_LAUNCH();
    if (rc) return rc;
    hipLaunchKernelGGL(dsmil_attn_bwd_kernel, dim3((N + DS_RPB - 1) / DS_RPB, B), dim3(256), 0, s, A, dA, Q, ldq, qcol0, qmax,
                       dots_ws, N, C, 1.0f / sqrtf((float)DS_Q), dY, ldy, dqmax);
    return MURCL_CHECK_LAUNCH();
}

// Soft-max backward alone, dS = A * (dA - sum_n A dA) per (bag, class), for the reassociated path (dS then weights the rows of X:
// sum_n dS[n,c] X[n,:] is all the query projection's backward needs).  dots_ws: B*C floats.
__global__ __launch_bounds__(256) void dsmil_softmax_bwd_kernel(const float* __restrict__ A, const float* __restrict__ dA,
                                                                const float* __restrict__ dots, long total, int N, int C,
                                                                float* __restrict__ dS) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long b = i / ((long)N * C);
        const int c = (int)(i % C);
        dS[i] = A[i] * (dA[i] - dots[b * C + c]);
    }
}
extern "C" int murcl_dsmil_softmax_bwd(const float* A, const float* dA, int B, int N, int C, float* dS, float* dots_ws,
                                       hipStream_t s) {
    if (B <= 0 || N <= 0) return 0;
    if (C <= 0 || C > 4 || !dots_ws) return -1;
    hipLaunchKernelGGL(dsmil_attn_dots_kernel, dim3(B, C), dim3(256), 0, s, A, dA, N, C, dots_ws);
    int rc = MURCL_CHECK_LAUNCH();
    if (rc) return rc;
    const long total = (long)B * N * C;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(dsmil_softmax_bwd_kernel, dim3(grid), dim3(256), 0, s, A, dA, dots_ws, total, N, C, dS);
    return MURCL_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- reassociated K6: attention + pooling in ONE pass over X, and the
// whole backward of (attention, pooling, query projection) in ONE more
//
// Forward (dsmil.py:76-78 with the logits as X . v, see murcl_dsmil_softmax): a wave walks rows_per_wave consecutive rows of one
// bag exactly like rows_dot_wsum_kernel (four rows in flight, a lane owns 8 consecutive columns per 512-column step), takes
// s[n,c] = X[n] . v[c] by a wave reduction, and keeps a running soft-max per class (online: maximum m, sum l, and the weighted row
// sum Z' = sum_n e^{s[n,c] - m} X[n] in registers, rescaled when m moves).  It leaves the raw logits S[n,c], its (m, l) and Z'.
// murcl_dsmil_attn_pool's second launch merges a bag's waves (weights e^{m_w - m}), and the third turns S into A = e^{S-m}/l.
//
// Backward: with dA[n,c] = X[n] . dZ[c] and w = A dA,
//     R[c] = sum_n A[n,c] (dA[n,c] - dot_c) X[n] = (sum_n w[n,c] X[n]) - dot_c Z[c],      dot_c = sum_n w[n,c]
// (Z = the pooled rows the forward saved), so dS never has to exist and X is read once; the same pass takes dWc = dcls^T X.
#ifndef DS_WAVE_SUM
#define DS_WAVE_SUM wave_sum_valu
#endif
// 8 consecutive elements as loaded (no conversion yet): the next batch of rows waits in this form while the current one is consumed
template <typename T> struct DsRaw;
template <> struct DsRaw<float> { f32x4 a, b; };
template <> struct DsRaw<bf16_t> { u32x4 v; };
__device__ __forceinline__ void ds_load_raw(const float* p, DsRaw<float>& r) { r.a = *(const f32x4*)p; r.b = *(const f32x4*)(p + 4); }
__device__ __forceinline__ void ds_load_raw(const bf16_t* p, DsRaw<bf16_t>& r) { r.v = *(const u32x4*)p; }
__device__ __forceinline__ void ds_zero_raw(DsRaw<float>& r) { r.a = f32x4{0.f, 0.f, 0.f, 0.f}; r.b = r.a; }
__device__ __forceinline__ void ds_zero_raw(DsRaw<bf16_t>& r) { r.v = u32x4{0u, 0u, 0u, 0u}; }
__device__ __forceinline__ void ds_to_float(const DsRaw<float>& r, float* v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = r.a[e]; v[4 + e] = r.b[e]; }
}
__device__ __forceinline__ void ds_to_float(const DsRaw<bf16_t>& r, float* v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = bf_lo(r.v[e]); v[2 * e + 1] = bf_hi(r.v[e]); }
}
#ifndef DS_PREFETCH
#define DS_PREFETCH (MODE != 2)   // the next four rows are requested before the current four are consumed.  Measured (C5 share): keeping
#endif                            // the rows in their loaded form until they are consumed is what mattered - bf16 passes 96 -> 52-62 us with
                                  // or without the early request (one batch of converted rows per wave left the loads exposed); the early
                                  // request itself: -3 % in bf16, neutral in f32, +20 % on the given-logits form at d = 512 (so not there)
template <typename T, int MODE>          // MODE 0: forward, 1: backward, 2: forward with the logits given (Ain = S [rows, C]; V unused),
                                         // 3: the logits only, S = X . v + bias (G = bias [C] or NULL): rows_dot on this pass's load schedule
__global__ __launch_bounds__(256) void dsmil_stream_kernel(const T* __restrict__ X, const float* __restrict__ V,
                                                           const float* __restrict__ Ain, const float* __restrict__ G,
                                                           int N, int d, int C, int rows_per_wave, float* __restrict__ S,
                                                           float* __restrict__ part, float* __restrict__ stat,
                                                           float* __restrict__ gpart, long rows_total, float vscale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wid = (long)blockIdx.x * 4 + wave;
    const long row0 = wid * rows_per_wave;
    if (row0 >= rows_total) return;
    const long row1 = min(rows_total, row0 + rows_per_wave);
    const float* v = MODE == 2 ? nullptr : V + (size_t)(row0 / N) * C * d;
    float zacc[2][2][8], gacc[(MODE == 1) ? 2 : 1][2][8], vr[2][2][8];
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};      // MODE 1: l_run = sum_n w[n,c]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int k = st * 512 + lane * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) { zacc[c][st][e] = 0.f; vr[c][st][e] = 0.f; if (MODE == 1) gacc[c][st][e] = 0.f; }
            if (MODE != 2 && c < C && k < d) load8<float>(v + (size_t)c * d + k, vr[c][st]);
        }
    const bool with_g = MODE == 1 && G != nullptr;
    DsRaw<T> nxt[4][2];
    auto request = [&](long rb) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long row = min(rb + u, row1 - 1);
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int k = st * 512 + lane * 8;
                if (k < d) ds_load_raw(X + row * d + k, nxt[u][st]);
                else ds_zero_raw(nxt[u][st]);
            }
        }
    };
    if (DS_PREFETCH) request(row0);
    for (long rb = row0; rb < row1; rb += 4) {
        float acc[4][2], xv[4][2][8], aw[4][2], gw[4][2];
        if (!DS_PREFETCH) request(rb);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long row = min(rb + u, row1 - 1);
            const bool live = rb + u < row1;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                acc[u][c] = 0.f;
                aw[u][c] = (MODE != 0 && live && c < C) ? Ain[row * C + c] : 0.f;      // MODE 2: the row's logit
                gw[u][c] = (with_g && live && c < C) ? G[row * C + c] : 0.f;
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) ds_to_float(nxt[u][st], xv[u][st]);
        }
        if (DS_PREFETCH && rb + 4 < row1) request(rb + 4);
        if (MODE != 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int st = 0; st < 2; ++st)
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[u][c] += xv[u][st][e] * vr[c][st][e];
        }
        float wgt[4][2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float sd[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) sd[u] = MODE == 2 ? aw[u][c] : ((c < C) ? DS_WAVE_SUM(acc[u][c]) * vscale : 0.f);
            if (MODE == 3) {
                const float bv = (G != nullptr && c < C) ? G[c] : 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (lane == 0 && rb + u < row1 && c < C) S[(rb + u) * C + c] = G != nullptr ? sd[u] + bv : sd[u];
                continue;
            }
            if (MODE != 1) {
                float mx = m_run[c];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (rb + u < row1) mx = fmaxf(mx, sd[u]);
                const float alpha = __expf(m_run[c] - mx);          // first batch: e^{-inf} = 0 on zero accumulators
                float ladd = 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    wgt[u][c] = (rb + u < row1 && c < C) ? __expf(sd[u] - mx) : 0.f;
                    ladd += wgt[u][c];
                    if (MODE == 0 && lane == 0 && rb + u < row1 && c < C) S[(rb + u) * C + c] = sd[u];
                }
                l_run[c] = l_run[c] * alpha + ladd;
                m_run[c] = mx;
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int e = 0; e < 8; ++e) zacc[c][st][e] *= alpha;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    wgt[u][c] = aw[u][c] * sd[u];                    // (aw is 0 on dead rows / classes)
                    l_run[c] += wgt[u][c];
                }
            }
        }
        if (MODE == 3) continue;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        zacc[c][st][e] += wgt[u][c] * xv[u][st][e];
                        if (MODE == 1) gacc[c][st][e] += gw[u][c] * xv[u][st][e];
                    }
    }
    if (MODE == 3) return;
    float* pr = part + (size_t)wid * C * d;
    float* gr = (MODE == 1 && with_g) ? gpart + (size_t)wid * C * d : nullptr;
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int k = st * 512 + lane * 8;
            if (k < d) {
                store8<float>(pr + (size_t)c * d + k, zacc[c][st]);
                if (MODE == 1 && gr) store8<float>(gr + (size_t)c * d + k, gacc[c][st]);
            }
        }
        if (lane == 0) { stat[(wid * C + c) * 2] = m_run[c]; stat[(wid * C + c) * 2 + 1] = l_run[c]; }
    }
}
// merge the W = N / rows_per_wave (<= 1024) waves of a bag.  grid (B*C, d/64): a workgroup first derives the bag's (m, l) and the
// waves' weights e^{m_w - m} (in LDS), then 4 x 64 threads add W/4 partial rows each for 64 columns.
// MODE 0: Z = sum_w e^{m_w-m} Z'_w / l and ml[b,c] = (m, l); MODE 1: R = (sum_w P_w - dot Z) * scale with dot = sum_w stat_w[1]
template <int MODE>
__global__ __launch_bounds__(256) void dsmil_merge_kernel(const float* __restrict__ part, const float* __restrict__ stat, int W, int d,
                                                          int C, const float* __restrict__ Zin, float scale,
                                                          float* __restrict__ out, float* __restrict__ ml,
                                                          const float* Sn, float* An, int N) {
    __shared__ float wgt[1024];
    __shared__ float red[256];
    __shared__ float acc[4][64];
    const int bc = blockIdx.x, b = bc / C, c = bc - b * C, tid = threadIdx.x;
    const float* st0 = stat + ((size_t)b * W * C + c) * 2;
    float m = 0.f;
    if (MODE == 0) {
        float mx = -INFINITY;
        for (int w = tid; w < W; w += 256) mx = fmaxf(mx, st0[(size_t)w * C * 2]);
        red[tid] = mx;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
        m = red[0];
        __syncthreads();
    }
    float ls = 0.f;
    for (int w = tid; w < W; w += 256) {
        const float g = MODE == 0 ? __expf(st0[(size_t)w * C * 2] - m) : 1.f;
        wgt[w] = g;
        ls += st0[(size_t)w * C * 2 + 1] * g;
    }
    red[tid] = ls;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float l = red[0];
    const int cl = tid & 63, wl = tid >> 6, k = blockIdx.y * 64 + cl;
    float t = 0.f;
    if (k < d) {
        const float* p0 = part + ((size_t)b * W * C + c) * d + k;
        float t4[4] = {0.f, 0.f, 0.f, 0.f};                     // four loads in flight per thread
        int w = wl;
        for (; w + 12 < W; w += 16)
#pragma unroll
            for (int u = 0; u < 4; ++u) t4[u] += p0[(size_t)(w + 4 * u) * C * d] * wgt[w + 4 * u];
        for (; w < W; w += 4) t4[0] += p0[(size_t)w * C * d] * wgt[w];
        t = (t4[0] + t4[1]) + (t4[2] + t4[3]);
    }
    acc[wl][cl] = t;
    __syncthreads();
    if (wl == 0 && k < d) {
        t = (acc[0][cl] + acc[1][cl]) + (acc[2][cl] + acc[3][cl]);
        out[(size_t)bc * d + k] = MODE == 0 ? t / l : (t - l * Zin[(size_t)bc * d + k]) * scale;
    }
    if (MODE == 0 && blockIdx.y == 0 && tid == 0) { ml[bc * 2] = m; ml[bc * 2 + 1] = l; }
    if (MODE == 0 && An) {
        // A = e^{S - m} / l for this workgroup's slice of the bag's rows (An may alias Sn): the statistics are in hand, so the
        // separate normalise launch over A is folded in here
        const int per = (N + gridDim.y - 1) / gridDim.y;
        const int n1 = min(N, (int)(blockIdx.y + 1) * per);
        const float rl = 1.f / l;
        for (int n = blockIdx.y * per + tid; n < n1; n += 256) {
            const size_t i = ((size_t)b * N + n) * C + c;
            An[i] = __expf(Sn[i] - m) * rl;
        }
    }
}
// rows_dot on the stream passes' load schedule; -2: shape not covered (the caller runs rows_dot_kernel)
static int rows_dot_stream_launch(const void* X, const float* V, const float* bias, float* out, int B, int N, int d, int C, int dtype,
                                  hipStream_t s) {
    const int rpw = murcl_rows_dot_wsum_plan(B, N, d, C);
    if (!rpw) return -2;
    const long rows = (long)B * N, waves = rows / rpw;
    dim3 grid((unsigned)((waves + 3) / 4));
    if (dtype == MURCL_DTYPE_F32)
        hipLaunchKernelGGL((dsmil_stream_kernel<float, 3>), grid, dim3(256), 0, s, (const float*)X, V, nullptr, bias, N, d, C, rpw, out, nullptr, nullptr, nullptr, rows, 1.f);
    else
        hipLaunchKernelGGL((dsmil_stream_kernel<bf16_t, 3>), grid, dim3(256), 0, s, (const bf16_t*)X, V, nullptr, bias, N, d, C, rpw, out, nullptr, nullptr, nullptr, rows, 1.f);
    return MURCL_CHECK_LAUNCH();
}
// plan: rows a wave takes (0: shape not covered -> rows_dot + soft-max + weighted_rowsum); the workspace holds
// (B*N/plan) * C * (d [+ d with dcls] + 2) floats
extern "C" int murcl_dsmil_stream_plan(int B, int N, int d, int C) {
    const int rpw = murcl_rows_dot_wsum_plan(B, N, d, C);
    return (rpw && N / rpw <= 1024) ? rpw : 0;          // (the merge keeps a bag's wave weights in LDS)
}
extern "C" int murcl_dsmil_attn_pool(const void* X, const float* v, float vscale, float* A /* [B,N,C]: logits, then the soft-max */,
                                     float* Z, float* ws, int B, int N, int d, int C, int dtype, hipStream_t s) {
    if (B <= 0) return 0;
    const int rpw = murcl_dsmil_stream_plan(B, N, d, C);
    if (!rpw || !ws) return -1;
    const long rows = (long)B * N, waves = rows / rpw;
    float* part = ws;
    float* stat = part + waves * C * d;
    float* ml = stat + waves * C * 2;
    dim3 grid((unsigned)((waves + 3) / 4));
    if (dtype == MURCL_DTYPE_F32)
        hipLaunchKernelGGL((dsmil_stream_kernel<float, 0>), grid, dim3(256), 0, s, (const float*)X, v, nullptr, nullptr, N, d, C, rpw, A, part, stat, nullptr, rows, vscale);
    else if (dtype == MURCL_DTYPE_BF16)
        hipLaunchKernelGGL((dsmil_stream_kernel<bf16_t, 0>), grid, dim3(256), 0, s, (const bf16_t*)X, v, nullptr, nullptr, N, d, C, rpw, A, part, stat, nullptr, rows, vscale);
    else
        return -1;
    int rc = MURCL_CHECK_LAUNCH();
    if (rc) return rc;
    hipLaunchKernelGGL(dsmil_merge_kernel<0>, dim3(B * C, (d + 63) / 64), dim3(256), 0, s, part, stat, N / rpw, d, C, nullptr, 1.f, Z, ml,
                       (const float*)A, A, N);
    return MURCL_CHECK_LAUNCH();
}
extern "C" int murcl_dsmil_attn_pool_bwd(const void* X, const float* dZ, const float* A, const float* Z, const float* dcls /* may be NULL */,
                                         float scale, float* R, float* gpart /* [B*N/plan][C*d], with dcls */, float* ws, int B, int N,
                                         int d, int C, int dtype, hipStream_t s) {
    if (B <= 0) return 0;
    const int rpw = murcl_dsmil_stream_plan(B, N, d, C);
    if (!rpw || !ws || (dcls && !gpart)) return -1;
    const long rows = (long)B * N, waves = rows / rpw;
    float* part = ws;
    float* stat = part + waves * C * d;
    dim3 grid((unsigned)((waves + 3) / 4));
    if (dtype == MURCL_DTYPE_F32)
        hipLaunchKernelGGL((dsmil_stream_kernel<float, 1>), grid, dim3(256), 0, s, (const float*)X, dZ, A, dcls, N, d, C, rpw, nullptr, part, stat, gpart, rows, 1.f);
    else if (dtype == MURCL_DTYPE_BF16)
        hipLaunchKernelGGL((dsmil_stream_kernel<bf16_t, 1>), grid, dim3(256), 0, s, (const bf16_t*)X, dZ, A, dcls, N, d, C, rpw, nullptr, part, stat, gpart, rows, 1.f);
    else
        return -1;
    int rc = MURCL_CHECK_LAUNCH();
    if (rc) return rc;
    hipLaunchKernelGGL(dsmil_merge_kernel<1>, dim3(B * C, (d + 63) / 64), dim3(256), 0, s, part, stat, N / rpw, d, C, Z, scale, R, nullptr,
                       (const float*)nullptr, (float*)nullptr, N);
    return MURCL_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- the [B*C]-row algebra of the reassociated K6 as 2 + 2 launches
// forward, per critical instance r = (bag, class): x_m = X[bag, m[r]] (kept in f32 for the backward), q_r = Wq x_m + bq
// (dsmil.py:74-75), v_r = Wq^T q_r (the vector the attention pass dots every patch with).  Replaces gather + cast + two GEMMs (+ a
// transpose of Wq) on 32-row tensors.  d <= DQ_MAXD, d % 4 == 0.  (First form: one workgroup per r walking the 128 rows of Wq, 32
// per wave, one wave reduction each: 54-62 us of pure latency - the rows are now spread over grid.y, four per wave, reduced together.)
#define DQ_MAXD 2048
#define DQ_OPB 16                 // outputs per workgroup of the matvec launches (grid.y = DS_Q / DQ_OPB): a wave takes 4, lanes walk k
// out[o] = W[o,:] . xs (+ bias[o]) for this workgroup's DQ_OPB rows of W [DS_Q, d]
__device__ __forceinline__ void dq_matvec(const float* __restrict__ W, const float* xs, const float* __restrict__ bias, int d,
                                          float* __restrict__ gout) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int o0 = blockIdx.y * DQ_OPB + wave * (DQ_OPB / 4);
    float a[DQ_OPB / 4];
#pragma unroll
    for (int i = 0; i < DQ_OPB / 4; ++i) a[i] = 0.f;
    for (int k = lane * 4; k < d; k += 256) {
        const f32x4 x = *(const f32x4*)(xs + k);
#pragma unroll
        for (int i = 0; i < DQ_OPB / 4; ++i) {
            const f32x4 w = *(const f32x4*)(W + (size_t)(o0 + i) * d + k);
            a[i] += w[0] * x[0] + w[1] * x[1] + w[2] * x[2] + w[3] * x[3];
        }
    }
#pragma unroll
    for (int i = 0; i < DQ_OPB / 4; ++i) {
        const float t = wave_sum_valu(a[i]);
        if (lane == 0) gout[o0 + i] = t + (bias ? bias[o0 + i] : 0.f);
    }
}
// forward launch 1, grid (B*C, DS_Q / DQ_OPB): x_m (written by the y = 0 workgroups) and q_r = Wq x_m + bq
template <typename T>
__global__ __launch_bounds__(256) void dsmil_q_kernel(const T* __restrict__ X, const int* __restrict__ m, const float* __restrict__ Wq,
                                                      const float* __restrict__ bq, int N, int d, int C, float* __restrict__ xm,
                                                      float* __restrict__ qmax) {
    __shared__ __attribute__((aligned(16))) float xs[DQ_MAXD];
    const int r = blockIdx.x, b = r / C, tid = threadIdx.x;
    const T* row = X + ((size_t)b * N + m[r]) * d;
    for (int k = tid; k < d; k += 256) {
        const float t = to_f<T>(row[k]);
        xs[k] = t;
        if (blockIdx.y == 0) xm[(size_t)r * d + k] = t;
    }
    __syncthreads();
    dq_matvec(Wq, xs, bq, d, qmax + (size_t)r * DS_Q);
}
// forward launch 2, grid (B*C, ceil(d / 1024)): v_r = Wq^T q_r, a thread owns 4 columns
__global__ __launch_bounds__(256) void dsmil_v_kernel(const float* __restrict__ qmax, const float* __restrict__ Wq, int d,
                                                      float* __restrict__ v) {
    __shared__ float qs[DS_Q];
    const int r = blockIdx.x, tid = threadIdx.x;
    if (tid < DS_Q) qs[tid] = qmax[(size_t)r * DS_Q + tid];
    __syncthreads();
    const int k = blockIdx.y * 1024 + tid * 4;
    if (k < d) {
        f32x4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int o = 0; o < DS_Q; o += 4)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += qs[o + u] * *(const f32x4*)(Wq + (size_t)(o + u) * d + k);
        *(f32x4*)(v + (size_t)r * d + k) = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
}
// backward launch 1, grid (B*C, DS_Q / DQ_OPB): dq_r = Wq R_r     (R = the gradient of v, murcl_dsmil_attn_pool_bwd)
__global__ __launch_bounds__(256) void dsmil_dq_kernel(const float* __restrict__ R, const float* __restrict__ Wq, int d,
                                                       float* __restrict__ dq) {
    __shared__ __attribute__((aligned(16))) float xs[DQ_MAXD];
    const int r = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < d; k += 256) xs[k] = R[(size_t)r * d + k];
    __syncthreads();
    dq_matvec(Wq, xs, nullptr, d, dq + (size_t)r * DS_Q);
}
// backward launch 2, grid (DS_Q rows o of Wq, ceil(d / 1024)): dWq[o,:] = sum_r q_r[o] R_r + dq_r[o] x_m,r ;  dbq[o] = sum_r dq_r[o]
__global__ __launch_bounds__(256) void dsmil_dwq_kernel(const float* __restrict__ R, const float* __restrict__ qmax,
                                                        const float* __restrict__ dq, const float* __restrict__ xm, int BC, int d,
                                                        float* __restrict__ dWq, float* __restrict__ dbq,
                                                        const float* __restrict__ dcmax, int C, float* __restrict__ dWc,
                                                        float* __restrict__ dbc, int accumulate) {
    const int o = blockIdx.x, tid = threadIdx.x;
    const int k = blockIdx.y * 1024 + tid * 4;
    if (o >= DS_Q) {
        // workgroups past the rows of Wq: the instance classifier's gradient from the max-instance term (train_RLMIL.py:516,527-529):
        // class c's score at its critical instance is Wc[c] . x_m[b,c] + bc[c], so dWc[c] += sum_b dcmax[b,c] x_m[b,c], dbc[c] += sum_b dcmax[b,c]
        const int c = o - DS_Q;
        if (k < d) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int r = c; r < BC; r += C) acc += dcmax[r] * *(const f32x4*)(xm + (size_t)r * d + k);
            f32x4* w = (f32x4*)(dWc + (size_t)c * d + k);
            *w = accumulate ? *w + acc : acc;
        }
        if (blockIdx.y == 0 && tid == 0) {
            float t = 0.f;
            for (int r = c; r < BC; r += C) t += dcmax[r];
            dbc[c] = accumulate ? dbc[c] + t : t;
        }
        return;
    }
    if (k < d) {
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        for (int r = 0; r < BC; ++r) {
            acc0 += qmax[(size_t)r * DS_Q + o] * *(const f32x4*)(R + (size_t)r * d + k);
            acc1 += dq[(size_t)r * DS_Q + o] * *(const f32x4*)(xm + (size_t)r * d + k);
        }
        *(f32x4*)(dWq + (size_t)o * d + k) = acc0 + acc1;
    }
    if (blockIdx.y == 0 && tid == 0) {
        float t = 0.f;
        for (int r = 0; r < BC; ++r) t += dq[(size_t)r * DS_Q + o];
        dbq[o] = t;
    }
}
extern "C" int murcl_dsmil_qv(const void* X, const int* m, const float* Wq, const float* bq, int B, int N, int d, int C, float* xm,
                              float* qmax, float* v, int dtype, hipStream_t s) {
    if (B <= 0 || C <= 0) return 0;
    if (d % 4 || d > DQ_MAXD) return -1;
    const dim3 gq(B * C, DS_Q / DQ_OPB);
    if (dtype == MURCL_DTYPE_F32) hipLaunchKernelGGL(dsmil_q_kernel<float>, gq, dim3(256), 0, s, (const float*)X, m, Wq, bq, N, d, C, xm, qmax);
    else if (dtype == MURCL_DTYPE_BF16) hipLaunchKernelGGL(dsmil_q_kernel<bf16_t>, gq, dim3(256), 0, s, (const bf16_t*)X, m, Wq, bq, N, d, C, xm, qmax);
    else return -1;
    int rc = MURCL_CHECK_LAUNCH();
    if (rc) return rc;
    hipLaunchKernelGGL(dsmil_v_kernel, dim3(B * C, (d + 1023) / 1024), dim3(256), 0, s, qmax, Wq, d, v);
    return MURCL_CHECK_LAUNCH();
}
extern "C" int murcl_dsmil_qv_bwd(const float* R, const float* qmax, const float* xm, const float* Wq, int BC, int d, float* dq_ws,
                                  float* dWq, float* dbq, hipStream_t s) {
    if (BC <= 0) return 0;
    if (d % 4 || d > DQ_MAXD || !dq_ws) return -1;
    hipLaunchKernelGGL(dsmil_dq_kernel, dim3(BC, DS_Q / DQ_OPB), dim3(256), 0, s, R, Wq, d, dq_ws);
    int rc = MURCL_CHECK_LAUNCH();
    if (rc) return rc;
    hipLaunchKernelGGL(dsmil_dwq_kernel, dim3(DS_Q, (d + 1023) / 1024), dim3(256), 0, s, R, qmax, dq_ws, xm, BC, d, dWq, dbq,
                       (const float*)nullptr, 1, (float*)nullptr, (float*)nullptr, 0);
    return MURCL_CHECK_LAUNCH();
}
// ... with the gradient of the max-instance class scores dcmax [B*C] (murcl_dsmil_argmax_max): dWc [C,d] and dbc [C] are ADDED to in
// the same second launch (C more workgroup rows) - no dense [B,N,C] gradient of the instance scores is ever formed for that term
extern "C" int murcl_dsmil_qv_bwd_cls(const float* R, const float* qmax, const float* xm, const float* Wq, int BC, int d, float* dq_ws,
                                      float* dWq, float* dbq, const float* dcmax, int C, float* dWc, float* dbc, int accumulate,
                                      hipStream_t s) {
    if (BC <= 0) return 0;
    if (d % 4 || d > DQ_MAXD || !dq_ws || C <= 0 || BC % C || !dcmax || !dWc || !dbc) return -1;
    hipLaunchKernelGGL(dsmil_dq_kernel, dim3(BC, DS_Q / DQ_OPB), dim3(256), 0, s, R, Wq, d, dq_ws);
    int rc = MURCL_CHECK_LAUNCH();
    if (rc) return rc;
    hipLaunchKernelGGL(dsmil_dwq_kernel, dim3(DS_Q + C, (d + 1023) / 1024), dim3(256), 0, s, R, qmax, dq_ws, xm, BC, d, dWq, dbq, dcmax, C,
                       dWc, dbc, accumulate);
    return MURCL_CHECK_LAUNCH();
}
