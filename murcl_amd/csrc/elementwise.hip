// Small bandwidth-bound helpers: dtype casts / weight transposes, column sums (bias grads),
// ReLU backward, GRU gate math (K7/K10, models/rlmil.py:47,78,199,213-217), fused Adam.
#include "common.h"

// ---------------------------------------------------------------- casts
__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) store4<bf16_t>(y + i, *(const f32x4*)(x + i));
    if (i < n && i + 3 >= n)
        for (long k = i; k < n; ++k) y[k] = f2bf(x[k]);
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = bf2f(x[i]);
}
extern "C" int murcl_cast(const void* x, void* y, long n, int dtype_in, int dtype_out, hipStream_t s) {
    if (n <= 0) return 0;
    int grid = (int)((n / 4 + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    if (dtype_in == MURCL_DTYPE_F32 && dtype_out == MURCL_DTYPE_BF16)
        hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, s, (const float*)x, (bf16_t*)y, n);
    else if (dtype_in == MURCL_DTYPE_BF16 && dtype_out == MURCL_DTYPE_F32)
        hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (float*)y, n);
    else
        return -1;
    return MURCL_CHECK_LAUNCH();
}

// y[C,R] (dtype T) = x[R,C]^T (f32): 32x32 LDS tiles
template <typename T>
__global__ void transpose_cast_kernel(const float* __restrict__ x, T* __restrict__ y, int R, int C) {
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < R && c0 + tx < C) t[k][tx] = x[(size_t)(r0 + k) * C + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < C && r0 + tx < R) y[(size_t)(c0 + k) * R + r0 + tx] = from_f<T>(t[tx][k]);
}
extern "C" int murcl_transpose_cast(const float* x, void* y, int R, int C, int dtype_out, hipStream_t s) {
    if (R <= 0 || C <= 0) return 0;
    dim3 grid((C + 31) / 32, (R + 31) / 32);
    if (dtype_out == MURCL_DTYPE_BF16)
        hipLaunchKernelGGL(transpose_cast_kernel<bf16_t>, grid, dim3(256), 0, s, x, (bf16_t*)y, R, C);
    else if (dtype_out == MURCL_DTYPE_F32)
        hipLaunchKernelGGL(transpose_cast_kernel<float>, grid, dim3(256), 0, s, x, (float*)y, R, C);
    else
        return -1;
    return MURCL_CHECK_LAUNCH();
}

// Several weight matrices -> their compute-dtype copies and/or transposes in ONE launch (the encoder needs W1..W3, Wa
// in bf16 for the forward and W2^T, W3^T, Wa^T for the dgrads: seven 4-5 us launches otherwise).  blockIdx.y = job.
struct MurclCastJob {           // 32 bytes, mirrored by murcl_amd/ops.py
    const float* src;           // [rows, cols] f32, contiguous
    void* dst;                  // [rows, cols] or (transpose) [cols, rows] in dtype_out
    int rows, cols;
    int transpose;              // bit 0: transpose; bits 8..: leading dimension of dst in elements (0: cols, or rows when transposed) -
    int dtype_out;              //   a job may fill a row / column block of a larger matrix (CLAM's interleaved gate weights)
                                // bit 1 (round 6): FRAGMENT ORDER - the destination [R, 512] (R % 16 == 0, contiguous) is written as the
                                //   persistent kernels' MFMA weight fragments instead of rows: frag_index() below
};
// Fragment order of a [R, 512] weight matrix (panel_gemm.hip / attn_pool*.hip, K = 512): a 16-row block is 16 k-steps x 64 lanes x 8
// elements - lane (q4, r16) of k-step kk holds elements [(kk + 16 q4) * 8, +8) of row r16 - so that a wave fetches a whole fragment with
// ONE fully coalesced 1-KiB load per k-step, straight into registers, while its first tiles are already in flight.
__device__ __forceinline__ size_t frag_index(int row, int col) {
    const int chunk = col >> 3, q4 = chunk >> 4, kk = chunk & 15;
    return (size_t)(row >> 4) * (16 * 512) + (size_t)((kk * 64 + q4 * 16 + (row & 15)) * 8 + (col & 7));
}
template <typename T>
__device__ __forceinline__ void cast_job_tile(const MurclCastJob& j, int tile, float (*t)[33]) {
    const int tc = (j.cols + 31) / 32;
    const int c0 = (tile % tc) * 32, r0 = (tile / tc) * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    T* y = (T*)j.dst;
    const int ldd = j.transpose >> 8;
    const bool frag = (j.transpose & 2) != 0;                    // (destination width 512: checked by the host side)
    if (j.transpose & 1) {
        const size_t ld = ldd ? ldd : j.rows;
        for (int k = ty; k < 32; k += 8)
            if (r0 + k < j.rows && c0 + tx < j.cols) t[k][tx] = j.src[(size_t)(r0 + k) * j.cols + c0 + tx];
        __syncthreads();
        for (int k = ty; k < 32; k += 8)
            if (c0 + k < j.cols && r0 + tx < j.rows)
                y[frag ? frag_index(c0 + k, r0 + tx) : (size_t)(c0 + k) * ld + r0 + tx] = from_f<T>(t[tx][k]);
    } else {
        const size_t ld = ldd ? ldd : j.cols;
        for (int k = ty; k < 32; k += 8)
            if (r0 + k < j.rows && c0 + tx < j.cols)
                y[frag ? frag_index(r0 + k, c0 + tx) : (size_t)(r0 + k) * ld + c0 + tx] = from_f<T>(j.src[(size_t)(r0 + k) * j.cols + c0 + tx]);
    }
}
__global__ __launch_bounds__(256) void cast_batch_kernel(const MurclCastJob* __restrict__ jobs) {
    __shared__ float t[32][33];
    const MurclCastJob j = jobs[blockIdx.y];
    const int tiles = ((j.rows + 31) / 32) * ((j.cols + 31) / 32);
    if ((int)blockIdx.x >= tiles) return;
    if (j.dtype_out == MURCL_DTYPE_BF16) cast_job_tile<bf16_t>(j, blockIdx.x, t);
    else cast_job_tile<float>(j, blockIdx.x, t);
}
extern "C" int murcl_cast_batch(const void* jobs_dev, int n_jobs, int max_tiles, hipStream_t s) {
    if (n_jobs <= 0 || max_tiles <= 0) return 0;
    hipLaunchKernelGGL(cast_batch_kernel, dim3(max_tiles, n_jobs), dim3(256), 0, s, (const MurclCastJob*)jobs_dev);
    return MURCL_CHECK_LAUNCH();
}

// The same jobs with a FLAT grid: workgroup w takes tile w - first_tile[j] of the job j whose range holds it (first_tile [n_jobs + 1]
// on the device, ascending; a binary search), so tables that mix a 3072 x 1024 matrix with 16-row blocks run as one launch without
// the (largest tile count) x (jobs) grid of mostly empty workgroups.
__global__ __launch_bounds__(256) void cast_flat_kernel(const MurclCastJob* __restrict__ jobs, const int* __restrict__ first_tile,
                                                        int n_jobs, int* __restrict__ tick) {
    __shared__ float t[32][33];
    // (tick: the device-side step counter of a captured optimizer step, murcl_adam_multi_live_deferred - this launch runs behind
    //  the update that read it and before the next one, so it can advance it for free instead of a one-thread launch of its own)
    if (tick && blockIdx.x == 0 && threadIdx.x == 0) tick[0] += 1;
    int lo = 0, hi = n_jobs - 1;
    const int w = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first_tile[mid] <= w) lo = mid; else hi = mid - 1;
    }
    const MurclCastJob j = jobs[lo];
    const int tile = w - first_tile[lo];
    if (j.dtype_out == MURCL_DTYPE_BF16) cast_job_tile<bf16_t>(j, tile, t);
    else cast_job_tile<float>(j, tile, t);
}
extern "C" int murcl_cast_batch_flat_tick(const void* jobs_dev, const int* first_tile_dev, int n_jobs, int total_tiles, int* tick_dev,
                                          hipStream_t s) {
    if (n_jobs <= 0 || total_tiles <= 0) return tick_dev ? -1 : 0;          // (nothing to launch: the caller must advance the counter itself)
    hipLaunchKernelGGL(cast_flat_kernel, dim3(total_tiles), dim3(256), 0, s, (const MurclCastJob*)jobs_dev, first_tile_dev, n_jobs, tick_dev);
    return MURCL_CHECK_LAUNCH();
}
extern "C" int murcl_cast_batch_flat(const void* jobs_dev, const int* first_tile_dev, int n_jobs, int total_tiles, hipStream_t s) {
    return murcl_cast_batch_flat_tick(jobs_dev, first_tile_dev, n_jobs, total_tiles, nullptr, s);
}

// torch.stack of several lists of equally shaped contiguous tensors in ONE launch: job i copies `bytes` (a multiple of 4) from src to
// dst; the table travels in the kernel arguments (PPO.update stacks T-1 states, actions and log-probabilities per rollout,
// rlmil.py:163-165: three ATen concatenations otherwise).  grid (ceil(max bytes / 4096), jobs).
struct StackTable { MurclCopyJob job[MURCL_STACK_MAX_JOBS]; };
__global__ __launch_bounds__(256) void stack_lists_kernel(StackTable t) {
    const MurclCopyJob j = t.job[blockIdx.y];
    const long w0 = (long)blockIdx.x * 1024, n = j.bytes >> 2;
    const unsigned* s = (const unsigned*)j.src;
    unsigned* d = (unsigned*)j.dst;
    for (long w = w0 + threadIdx.x; w < w0 + 1024 && w < n; w += 256) d[w] = s[w];
}
extern "C" int murcl_stack_lists(const MurclCopyJob* jobs_host, int n_jobs, hipStream_t s) {
    if (n_jobs <= 0) return 0;
    if (n_jobs > MURCL_STACK_MAX_JOBS) return -1;
    StackTable t;
    long mx = 0;
    for (int i = 0; i < n_jobs; ++i) {
        if (jobs_host[i].bytes < 0 || (jobs_host[i].bytes & 3)) return -1;
        t.job[i] = jobs_host[i];
        mx = jobs_host[i].bytes > mx ? jobs_host[i].bytes : mx;
    }
    if (mx == 0) return 0;
    hipLaunchKernelGGL(stack_lists_kernel, dim3((unsigned)((mx + 4095) / 4096), n_jobs), dim3(256), 0, s, t);
    return MURCL_CHECK_LAUNCH();
}

// dst += src (f32) for several (src, dst, bytes) jobs in ONE launch: the gradients a backward node produced as its own tensors added
// to the parameters' pre-seated gradient views (what autograd's AccumulateGrad does with one ATen add per parameter - eight for CLAM-SB)
__global__ __launch_bounds__(256) void add_lists_kernel(StackTable t) {
    const MurclCopyJob j = t.job[blockIdx.y];
    const long w0 = (long)blockIdx.x * 1024, n = j.bytes >> 2;
    const float* s = (const float*)j.src;
    float* d = (float*)j.dst;
    for (long w = w0 + threadIdx.x; w < w0 + 1024 && w < n; w += 256) d[w] += s[w];
}
extern "C" int murcl_add_lists(const MurclCopyJob* jobs_host, int n_jobs, hipStream_t s) {
    if (n_jobs <= 0) return 0;
    if (n_jobs > MURCL_STACK_MAX_JOBS) return -1;
    StackTable t;
    long mx = 0;
    for (int i = 0; i < n_jobs; ++i) {
        if (jobs_host[i].bytes < 0 || (jobs_host[i].bytes & 3)) return -1;
        t.job[i] = jobs_host[i];
        mx = jobs_host[i].bytes > mx ? jobs_host[i].bytes : mx;
    }
    if (mx == 0) return 0;
    hipLaunchKernelGGL(add_lists_kernel, dim3((unsigned)((mx + 4095) / 4096), n_jobs), dim3(256), 0, s, t);
    return MURCL_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- column sums: out[n] (+)= sum_r x[r][n]
// grid = (column groups of 16*CPT, row splits); a thread owns CPT = 16/sizeof(T) consecutive columns (16-byte loads)
// for one of 16 row lanes; each block adds its partial sums atomically.  Row splits are capped at 64: float atomics
// execute at the memory side and adders on ONE address serialise (1024 adders per column: 228 us for 134 MB).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int R, int N, int ld,
                                                     int rows_per_block, int overwrite) {
    constexpr int CPT = 16 / (int)sizeof(T);
    __shared__ float red[16][16][CPT + 1];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 * CPT + cl * CPT;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
    float s[CPT];
#pragma unroll
    for (int e = 0; e < CPT; ++e) s[e] = 0.f;
    if (c + CPT - 1 < N && (ld % CPT) == 0) {                   // aligned rows: 16-byte loads
        int r = r0 + rl;
        for (; r + 48 < r1; r += 64) {                          // four rows in flight per thread
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (CPT == 8) {
                    load8<T>(x + (size_t)(r + 16 * u) * ld + c, v[u]);
                } else {
                    const f32x4 t = load4<T>(x + (size_t)(r + 16 * u) * ld + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[u][e] = t[e];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < CPT; ++e) s[e] += v[u][e];
        }
        for (; r < r1; r += 16) {
            if constexpr (CPT == 8) {
                float v[8];
                load8<T>(x + (size_t)r * ld + c, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += v[e];
            } else {
                const f32x4 v = load4<T>(x + (size_t)r * ld + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] += v[e];
            }
        }
    } else if (c < N) {
        for (int r = r0 + rl; r < r1; r += 16)
            for (int e = 0; e < CPT && c + e < N; ++e) s[e] += to_f<T>(x[(size_t)r * ld + c + e]);
    }
#pragma unroll
    for (int e = 0; e < CPT; ++e) red[rl][cl][e] = s[e];
    __syncthreads();
    if (rl == 0 && c < N) {
        for (int e = 0; e < CPT && c + e < N; ++e) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k][cl][e];
            if (overwrite) out[c + e] = t;                     // one row split and nothing to add to: no zero-fill before the launch
            else if (gridDim.y == 1) out[c + e] += t;
            else atomicAdd(out + c + e, t);
        }
    }
}
extern "C" int murcl_colsum(const void* x, float* out, int R, int N, int ld, int dtype, int accumulate, hipStream_t s) {
    if (N <= 0) return 0;
    const int bc = dtype == MURCL_DTYPE_BF16 ? 128 : 64;     // columns per block
    const int cg = (N + bc - 1) / bc;
    int splits = R > 64 ? (1024 + cg - 1) / cg : 1;        // ~1024 blocks, >= 32 rows each, <= 64 adders per column
    if (splits > (R + 31) / 32) splits = (R + 31) / 32;
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
    const int rpb = (R + splits - 1) / splits;
    dim3 grid(cg, (R + rpb - 1) / rpb);
    const int overwrite = !accumulate && grid.y == 1;
    if (!accumulate && !overwrite) {
        hipError_t e = hipMemsetAsync(out, 0, (size_t)N * 4, s);
        if (e != hipSuccess) return (int)e;
    }
    if (dtype == MURCL_DTYPE_F32)
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)x, out, R, N, ld, rpb, overwrite);
    else if (dtype == MURCL_DTYPE_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, out, R, N, ld, rpb, overwrite);
    else
        return -1;
    return MURCL_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- 1-bit ReLU' mask of an existing activation tensor
// bits = (x > 0) in the panel GEMM's mask layout (panel_gemm.hip header): 128-byte blocks per (32-row tile, 32-column
// group); one wave per block, lane L = 16*q4 + r16 owns the 16-bit word of rows {r16, 16+r16} x columns
// {4q4.., 16+4q4..}.  For layers whose forward did not run through the panel kernel (CLAM's 1024 -> 512 layer).
template <typename T>
__global__ __launch_bounds__(256) void relu_bitmask_kernel(const T* __restrict__ x, uint8_t* __restrict__ bits, int M,
                                                           int N, int ld) {
    const int lane = threadIdx.x & 63, q4 = lane >> 4, r16 = lane & 15;
    const long blk = (long)blockIdx.x * 4 + (threadIdx.x >> 6), groups = N >> 5;
    if (blk >= (long)(M >> 5) * groups) return;
    const long tile = blk / groups;
    const int cgp = (int)(blk - tile * groups);
    unsigned word = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const f32x4 v = load4<T>(x + (size_t)(32 * tile + 16 * i + r16) * ld + 32 * cgp + 16 * jj + 4 * q4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int idx = 8 * i + 4 * jj + r;
                word |= (v[r] > 0.f ? 1u : 0u) << ((7 - (idx >> 1)) + 8 * (idx & 1));
            }
        }
    *(uint16_t*)(bits + blk * 128 + lane * 2) = (uint16_t)word;
}
extern "C" int murcl_relu_bitmask(const void* x, void* bits, int M, int N, int ld, int dtype, hipStream_t s) {
    if (M <= 0 || N <= 0) return 0;
    if (M % 32 || N % 32 || ld % 4) return -1;
    const long blocks = (long)(M / 32) * (N / 32);
    dim3 grid((unsigned)((blocks + 3) / 4));
    if (dtype == MURCL_DTYPE_BF16)
        hipLaunchKernelGGL(relu_bitmask_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, (uint8_t*)bits, M, N, ld);
    else if (dtype == MURCL_DTYPE_F32)
        hipLaunchKernelGGL(relu_bitmask_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (uint8_t*)bits, M, N, ld);
    else
        return -1;
    return MURCL_CHECK_LAUNCH();
}

// Dropout applied in place + the 1-bit mask of what survives, one pass: x[i] *= keep(i) with the counter-based keep mask of
// murcl_dropout_mask (same seed -> the same mask, never materialised), bits = (x > 0) afterwards in the panel layout above.
// CLAM's Dropout(0.25) after the first layer's ReLU (clam.py:69-72): replaces mask generation (one 805 MB write at 786 k rows)
// + an elementwise multiply (two reads, one write) + the mask pass (one read) by one read and one write.
template <typename T>
__global__ __launch_bounds__(256) void dropout_relu_bitmask_kernel(T* __restrict__ x, uint8_t* __restrict__ bits, int M, int N,
                                                                   unsigned thresh, float scale, unsigned long long seed) {
    // A wave owns a 32-row x 128-column strip (four 32-column mask blocks).  Rows are read and written in whole 256-byte (bf16)
    // pieces, 16 lanes x 8 consecutive elements - the mask layout's own access pattern (8 bytes per lane over 16 rows) moves
    // 512 bytes per instruction over sixteen cache lines and is bound by the address path (519 us for 1.6 GB at 786 k rows).
    // The 1-bit flags are assembled in LDS: element (row, col) of a block belongs to word 16*((col & 15) >> 2) + (row & 15),
    // bit idx = 8*(row >> 4) + 4*(col >> 4) + (col & 3) -> position (7 - idx/2) + 8*(idx & 1).
    __shared__ unsigned words[4][4][64];                              // [wave][block of the strip][mask lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int strips = N >> 7;
    const long wb = (long)blockIdx.x * 4 + wave;
    const bool live = wb < (long)(M >> 5) * strips;
    for (int k = lane; k < 4 * 64; k += 64) words[wave][k >> 6][k & 63] = 0u;
    __syncthreads();
    if (live) {
        const long tile = wb / strips;
        const int c0 = (int)(wb - tile * strips) * 128;
        const int rsub = lane >> 4, cl = (lane & 15) * 8;              // row within a 4-row set, first of this lane's 8 columns
        const int blk = cl >> 5, cb = cl & 31;                         // mask block of the strip, column inside it (0, 8, 16, 24)
#pragma unroll
        for (int rs = 0; rs < 8; ++rs) {
            const int row = 4 * rs + rsub;                             // 0..31 inside the tile
            const long e0 = (long)(32 * tile + row) * N + c0 + cl;     // flat index, multiple of 8: one random word per lane
            float v[8];
            load8<T>(x + e0, v);
            const unsigned long long rw = murcl_drop_word(seed, e0 >> 3);
            unsigned nib[2] = {0u, 0u};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float kp = ((unsigned)(rw >> (8 * e)) & 255u) < thresh ? scale : 0.f;
                v[e] *= kp;
                const int col = cb + e;                                // column inside the 32-column block
                const int idx = 8 * (row >> 4) + 4 * (col >> 4) + (col & 3);
                nib[e >> 2] |= (v[e] > 0.f ? 1u : 0u) << ((7 - (idx >> 1)) + 8 * (idx & 1));
            }
            store8<T>(x + e0, v);
            if (bits) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int col = cb + 4 * h;
                    atomicOr(&words[wave][blk][16 * ((col & 15) >> 2) + (row & 15)], nib[h]);
                }
            }
        }
    }
    __syncthreads();
    if (live && bits) {
        const long tile = wb / strips;
        const int g0 = (int)(wb - tile * strips) * 4;
#pragma unroll
        for (int b = 0; b < 4; ++b)
            *(uint16_t*)(bits + (tile * (N >> 5) + g0 + b) * 128 + lane * 2) = (uint16_t)words[wave][b][lane];
    }
}
extern "C" int murcl_dropout_relu_bitmask(void* x, void* bits, int M, int N, float keep_p, float scale, unsigned long long seed,
                                          int dtype, hipStream_t s) {
    if (M <= 0 || N <= 0) return 0;
    if (M % 32 || N % 128 || !(keep_p >= 0.f && keep_p <= 1.f)) return -1;
    const unsigned thresh = (unsigned)(keep_p * 256.f + 0.5f);
    const long blocks = (long)(M / 32) * (N / 128);
    dim3 grid((unsigned)((blocks + 3) / 4));
    if (dtype == MURCL_DTYPE_BF16)
        hipLaunchKernelGGL(dropout_relu_bitmask_kernel<bf16_t>, grid, dim3(256), 0, s, (bf16_t*)x, (uint8_t*)bits, M, N, thresh, scale, seed);
    else if (dtype == MURCL_DTYPE_F32)
        hipLaunchKernelGGL(dropout_relu_bitmask_kernel<float>, grid, dim3(256), 0, s, (float*)x, (uint8_t*)bits, M, N, thresh, scale, seed);
    else
        return -1;
    return MURCL_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- ReLU backward: dx = dy * (y > 0)
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
extern "C" int murcl_relu_bwd(const float* dy, const float* y, float* dx, long n, hipStream_t s) {
    if (n <= 0) return 0;
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid), dim3(256), 0, s, dy, y, dx, n);
    return MURCL_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- GRU gates (PyTorch order r, z, n)
// gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh, both [B,3H].  gates out: [B,3H] = (r, z, n) for backward.
// gh_bcast != 0: gh is a single row [3H] shared by every batch row (zero initial state: gh = b_hh).
__global__ void gru_gates_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                     const float* __restrict__ hprev, float* __restrict__ hnew,
                                     float* __restrict__ gates, int B, int H, int gh_bcast) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * H) return;
    const int b = (int)(idx / H), k = (int)(idx % H);
    const float* gib = gi + (size_t)b * 3 * H;
    const float* ghb = gh + (gh_bcast ? 0 : (size_t)b * 3 * H);
    const float r = 1.f / (1.f + expf(-(gib[k] + ghb[k])));
    const float z = 1.f / (1.f + expf(-(gib[H + k] + ghb[H + k])));
    const float nn = tanhf(gib[2 * H + k] + r * ghb[2 * H + k]);
    const float hp = hprev ? hprev[idx] : 0.f;
    hnew[idx] = (1.f - z) * nn + z * hp;
    float* g = gates + (size_t)b * 3 * H;
    g[k] = r; g[H + k] = z; g[2 * H + k] = nn;
}
// dgi, dgh [B,3H]; dhprev_direct [B,H] = dh * z
__global__ void gru_gates_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ gates,
                                     const float* __restrict__ gh, const float* __restrict__ hprev,
                                     float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ dhprev,
                                     int B, int H, int gh_bcast, int accumulate) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * H) return;
    const int b = (int)(idx / H), k = (int)(idx % H);
    const float* g = gates + (size_t)b * 3 * H;
    const float r = g[k], z = g[H + k], nn = g[2 * H + k];
    const float ghn = gh[(gh_bcast ? 0 : (size_t)b * 3 * H) + 2 * H + k];
    const float hp = hprev ? hprev[idx] : 0.f;
    const float d = dh[idx];
    const float dn = d * (1.f - z) * (1.f - nn * nn);
    const float dz = d * (hp - nn) * z * (1.f - z);
    const float dr = dn * ghn * r * (1.f - r);
    float* a = dgi + (size_t)b * 3 * H;
    float* c = dgh + (size_t)b * 3 * H;
    a[k] = dr; a[H + k] = dz; a[2 * H + k] = dn;
    c[k] = dr; c[H + k] = dz; c[2 * H + k] = dn * r;
    if (dhprev) dhprev[idx] = accumulate ? dhprev[idx] + d * z : d * z;
}
extern "C" int murcl_gru_gates_fwd(const float* gi, const float* gh, const float* hprev, float* hnew, float* gates,
                                   int B, int H, int gh_bcast, hipStream_t s) {
    const long n = (long)B * H;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gi, gh, hprev, hnew, gates, B, H, gh_bcast);
    return MURCL_CHECK_LAUNCH();
}
extern "C" int murcl_gru_gates_bwd(const float* dh, const float* gates, const float* gh, const float* hprev, float* dgi,
                                   float* dgh, float* dhprev, int B, int H, int gh_bcast, hipStream_t s) {
    const long n = (long)B * H;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dh, gates, gh, hprev, dgi, dgh, dhprev, B, H, gh_bcast, 0);
    return MURCL_CHECK_LAUNCH();
}
// the same with dhprev += dh * z when accumulate != 0 (backward through time: dhprev already holds the step's own upstream)
extern "C" int murcl_gru_gates_bwd_into(const float* dh, const float* gates, const float* gh, const float* hprev, float* dgi,
                                        float* dgh, float* dhprev, int B, int H, int gh_bcast, int accumulate, hipStream_t s) {
    const long n = (long)B * H;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dh, gates, gh, hprev, dgi, dgh, dhprev, B, H, gh_bcast, accumulate);
    return MURCL_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- Adam (torch.optim.Adam semantics, L2 weight decay)
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd,
                            float bc1, float bc2_sqrt, int zero_grad) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float gi = g[i];
        if (zero_grad) g[i] = 0.f;              // next step's zero_grad folded into this pass
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}
extern "C" int murcl_adam_step(float* p, float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int step, int zero_grad, hipStream_t s) {
    if (n <= 0) return 0;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, zero_grad);
    return MURCL_CHECK_LAUNCH();
}

// The same update for up to MURCL_ADAM_MAX_JOBS flat runs (parameter groups with their own lr / step count) in ONE launch: the
// table travels by value in the kernel arguments, a workgroup takes AM_CHUNK consecutive elements of one run (16-byte accesses
// when the run's four arrays are 16-byte aligned, which flat optimizer buffers are).  Element-wise the arithmetic of adam_kernel.
#define AM_CHUNK 4096
struct AdamTable {
    MurclAdamJob job[MURCL_ADAM_MAX_JOBS];
    float bc1[MURCL_ADAM_MAX_JOBS], bc2s[MURCL_ADAM_MAX_JOBS];
    int first_chunk[MURCL_ADAM_MAX_JOBS + 1];
    int n_jobs;
};
__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, float lr_bc1, float b1, float b2, float eps, float wd,
                                         float bc2_sqrt) {
    float gi = g;
    if (wd != 0.f) gi += wd * p;
    m = b1 * m + (1.f - b1) * gi;
    v = b2 * v + (1.f - b2) * gi * gi;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - lr_bc1 * (m / denom);
}
// `replays` (may be NULL; int[2], [0] = steps taken by replays so far): the step counts of the table are those of the launch that was
// CAPTURED into a hipGraph, and every replay of that graph is one more optimizer step - the bias corrections then use
// step + replays[0], read by every workgroup at its start; a one-thread launch behind this one (adam_replay_tick_kernel, the next
// node of the same graph) advances the counter for the next replay: a replayed step is a real training step, not the captured one
// again (bench.py's graph-timed region).  (An arrival ticket inside this kernel - last workgroup out advances the counter - cost
// 47 us per launch: 28 -> 75 us at 6 M parameters, tools/adam_live_probe.py.)
__global__ __launch_bounds__(256) void adam_multi_kernel(AdamTable t, float b1, float b2, float eps, float wd, int zero_grad,
                                                         int* __restrict__ replays) {
    int j = 0;
    while (j + 1 < t.n_jobs && (int)blockIdx.x >= t.first_chunk[j + 1]) ++j;
    const MurclAdamJob jb = t.job[j];
    const long base = (long)((int)blockIdx.x - t.first_chunk[j]) * AM_CHUNK;
    const long n = jb.n - base < AM_CHUNK ? jb.n - base : AM_CHUNK;
    float* p = jb.p + base; float* g = jb.g + base; float* m = jb.m + base; float* v = jb.v + base;
    float bc1 = t.bc1[j], bc2s = t.bc2s[j];
    if (replays) {
        const int extra = __hip_atomic_load(replays, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (extra) {                                         // (the same expressions as the host side of murcl_adam_multi)
            const float st = (float)(jb.step + extra);
            bc1 = 1.f - powf(b1, st);
            bc2s = sqrtf(1.f - powf(b2, st));
        }
    }
    const float lr_bc1 = jb.lr / bc1;
    const bool vec = ((((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0);
    if (vec && n == AM_CHUNK) {
        f32x4 P[4], G[4], M[4], V[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = (u * 256 + threadIdx.x) * 4;
            P[u] = *(const f32x4*)(p + i); G[u] = *(const f32x4*)(g + i); M[u] = *(const f32x4*)(m + i); V[u] = *(const f32x4*)(v + i);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = (u * 256 + threadIdx.x) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = P[u][e], ge = G[u][e], me = M[u][e], ve = V[u][e];
                adam_one(pe, ge, me, ve, lr_bc1, b1, b2, eps, wd, bc2s);
                P[u][e] = pe; M[u][e] = me; V[u][e] = ve;
            }
            *(f32x4*)(p + i) = P[u]; *(f32x4*)(m + i) = M[u]; *(f32x4*)(v + i) = V[u];
            if (zero_grad) *(f32x4*)(g + i) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    for (long i = threadIdx.x; i < n; i += 256) {
        float pi = p[i], gi = g[i], mi = m[i], vi = v[i];
        adam_one(pi, gi, mi, vi, lr_bc1, b1, b2, eps, wd, bc2s);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (zero_grad) g[i] = 0.f;
    }
}
__global__ void adam_replay_tick_kernel(int* __restrict__ replays) { replays[0] += 1; }
static thread_local bool g_adam_defer_tick = false;
extern "C" int murcl_adam_multi_live(const MurclAdamJob* jobs_host, int n_jobs, float beta1, float beta2, float eps, float weight_decay,
                                     int zero_grad, int* replays_dev, hipStream_t s);
extern "C" int murcl_adam_multi(const MurclAdamJob* jobs_host, int n_jobs, float beta1, float beta2, float eps, float weight_decay,
                                int zero_grad, hipStream_t s) {
    return murcl_adam_multi_live(jobs_host, n_jobs, beta1, beta2, eps, weight_decay, zero_grad, nullptr, s);
}
extern "C" int murcl_adam_multi_live(const MurclAdamJob* jobs_host, int n_jobs, float beta1, float beta2, float eps, float weight_decay,
                                     int zero_grad, int* replays_dev, hipStream_t s) {
    if (n_jobs <= 0) return 0;
    if (n_jobs > MURCL_ADAM_MAX_JOBS) return -1;
    AdamTable t;
    t.n_jobs = 0;
    long chunks = 0;
    for (int j = 0; j < n_jobs; ++j) {
        if (jobs_host[j].n <= 0) continue;
        const int k = t.n_jobs++;
        t.job[k] = jobs_host[j];
        t.bc1[k] = 1.f - powf(beta1, (float)jobs_host[j].step);
        t.bc2s[k] = sqrtf(1.f - powf(beta2, (float)jobs_host[j].step));
        t.first_chunk[k] = (int)chunks;
        chunks += (jobs_host[j].n + AM_CHUNK - 1) / AM_CHUNK;
    }
    if (!t.n_jobs) return 0;
    if (chunks > 0x7fffffffL) return -1;
    t.first_chunk[t.n_jobs] = (int)chunks;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)chunks), dim3(256), 0, s, t, beta1, beta2, eps, weight_decay, zero_grad,
                       replays_dev);
    if (replays_dev && !g_adam_defer_tick) hipLaunchKernelGGL(adam_replay_tick_kernel, dim3(1), dim3(1), 0, s, replays_dev);
    return MURCL_CHECK_LAUNCH();
}
// the same launch WITHOUT the one-thread tick behind it: the caller advances replays_dev[0] with its next launch on the stream
// (murcl_cast_batch_flat_tick, the weight-view refresh that follows every optimizer step) or with murcl_replay_tick
extern "C" int murcl_adam_multi_live_deferred(const MurclAdamJob* jobs_host, int n_jobs, float beta1, float beta2, float eps,
                                              float weight_decay, int zero_grad, int* replays_dev, hipStream_t s) {
    g_adam_defer_tick = true;
    const int rc = murcl_adam_multi_live(jobs_host, n_jobs, beta1, beta2, eps, weight_decay, zero_grad, replays_dev, s);
    g_adam_defer_tick = false;
    return rc;
}
extern "C" int murcl_replay_tick(int* replays_dev, hipStream_t s) {
    if (!replays_dev) return -1;
    hipLaunchKernelGGL(adam_replay_tick_kernel, dim3(1), dim3(1), 0, s, replays_dev);
    return MURCL_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- bag-level odds and ends of the training step
// out = a x + b y over n floats (the rewards of a contrastive step: cosine of step t-1 minus cosine of step t, train_MuRCL.py:282-283)
__global__ void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float a, float b, float* __restrict__ out, long n) {
    if (b == 0.f) {                          // a plain scaling: y is not read (0 * inf would turn an overflow into a NaN)
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = a * x[i];
        return;
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = a * x[i] + b * y[i];
}
extern "C" int murcl_axpby(const float* x, const float* y, float a, float b, float* out, long n, hipStream_t s) {
    if (n <= 0) return 0;
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, s, x, y, a, b, out, n);
    return MURCL_CHECK_LAUNCH();
}
// out[0] = mean of n floats, one workgroup, a fixed summation order (the step loss = mean of the T patch-step losses, train_MuRCL.py:291)
__global__ __launch_bounds__(256) void mean_small_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
    __shared__ float part[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += x[i];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0] / (float)n;
}
// out[g] = mean of x[g * group .. (g + 1) * group): one workgroup per group, a fixed summation order (CLAM-SB's instance loss per bag
// -> per patch step, train_RLMIL.py:336)
__global__ __launch_bounds__(256) void group_mean_kernel(const float* __restrict__ x, int group, float* __restrict__ out) {
    __shared__ float part[256];
    const float* xg = x + (long)blockIdx.x * group;
    float acc = 0.f;
    for (int i = threadIdx.x; i < group; i += 256) acc += xg[i];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = part[0] / (float)group;
}
extern "C" int murcl_group_mean(const float* x, int groups, int group, float* out, hipStream_t s) {
    if (groups <= 0 || group <= 0) return groups == 0 ? 0 : -1;
    hipLaunchKernelGGL(group_mean_kernel, dim3(groups), dim3(256), 0, s, x, group, out);
    return MURCL_CHECK_LAUNCH();
}
extern "C" int murcl_mean_small(const float* x, int n, float* out, hipStream_t s) {
    if (n <= 0 || n > (1 << 20)) return -1;
    hipLaunchKernelGGL(mean_small_kernel, dim3(1), dim3(256), 0, s, x, n, out);
    return MURCL_CHECK_LAUNCH();
}

// dst <- src, `bytes` bytes, both 16-byte aligned (flat parameter buffers: policy -> policy_old after a PPO update, rlmil.py:183): a
// streaming copy of 16-byte pieces, the last 1..15 bytes by one lane - a launch of this library in the step's sequence instead of a
// runtime blit
__global__ __launch_bounds__(256) void copy_bytes_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long n16, int tail) {
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = (long)blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + u * 256 < n16) v[u] = src[i + u * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + u * 256 < n16) dst[i + u * 256] = v[u];
    }
    if (tail && blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned char* s = (const unsigned char*)(src + n16);
        unsigned char* d = (unsigned char*)(dst + n16);
        for (int k = 0; k < tail; ++k) d[k] = s[k];
    }
}
extern "C" int murcl_copy_bytes(const void* src, void* dst, long bytes, hipStream_t stream) {
    if (bytes <= 0) return 0;
    if ((((size_t)src | (size_t)dst) & 15) != 0) return -1;
    const long n16 = bytes / 16;
    const long want = (n16 + 1023) / 1024;
    hipLaunchKernelGGL(copy_bytes_kernel, dim3((unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want))), dim3(256), 0, stream,
                       (const u32x4*)src, (u32x4*)dst, n16, (int)(bytes & 15));
    return MURCL_CHECK_LAUNCH();
}

// C[M,N] (+)= A[M,K] B[N,K]^T for a handful of k (K <= 16, f32): the input gradient of a classifier with two or ten outputs
// (dX = dY W, K = the class count; train_RLMIL.py's heads) as one small launch - the matrix-core GEMMs step k by 32 and would need
// both operands zero-padded to 32 columns first (two fills and two copies per call)
__global__ __launch_bounds__(256) void smallk_nt_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                        int M, int N, int K, int accumulate) {
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / N), n = (int)(i % N);
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc += A[(long)m * K + k] * B[(long)n * K + k];
        C[i] = accumulate ? C[i] + acc : acc;
    }
}
extern "C" int murcl_gemm_nt_smallk(const float* A, const float* B, float* C, int M, int N, int K, int accumulate, hipStream_t s) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || K > 16) return -1;
    const long blocks = ((long)M * N + 255) / 256;
    hipLaunchKernelGGL(smallk_nt_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, A, B, C, M, N, K, accumulate);
    return MURCL_CHECK_LAUNCH();
}
// dst[R, Cp] = [src[R, C] | 0]: zero-padded columns in one launch (f32 / bf16 by element size) for the operands of products whose
// inner or outer extent is not a multiple of what the matrix-core kernels step by
__global__ __launch_bounds__(256) void pad_cols_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, long R, int C,
                                                       int Cp, int es) {
    const long total = R * Cp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / Cp;
        const int c = (int)(i % Cp);
        if (es == 4) ((unsigned*)dst)[i] = c < C ? ((const unsigned*)src)[r * C + c] : 0u;
        else ((unsigned short*)dst)[i] = c < C ? ((const unsigned short*)src)[r * C + c] : (unsigned short)0;
    }
}
extern "C" int murcl_pad_cols(const void* src, void* dst, long R, int C, int Cp, int elem_size, hipStream_t s) {
    if (R <= 0 || Cp <= 0) return 0;
    if (C > Cp || (elem_size != 2 && elem_size != 4)) return -1;
    const long blocks = (R * Cp + 255) / 256;
    hipLaunchKernelGGL(pad_cols_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, (const unsigned char*)src,
                       (unsigned char*)dst, R, C, Cp, elem_size);
    return MURCL_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- SGD (torch.optim.SGD semantics: L2 decay, momentum
// buffer initialised with the first gradient, dampening 0, optional Nesterov; train_MuRCL.py:158-163)
__global__ void sgd_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf, long n, float lr,
                           float momentum, int nesterov, float wd, int first, int zero_grad) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float gi = g[i];
        if (zero_grad) g[i] = 0.f;
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        if (momentum != 0.f) {
            const float b = first ? gi : momentum * buf[i] + gi;
            buf[i] = b;
            gi = nesterov ? gi + momentum * b : b;
        }
        p[i] = pi - lr * gi;
    }
}
extern "C" int murcl_sgd_step(float* p, float* g, float* buf, long n, float lr, float momentum, int nesterov,
                              float weight_decay, int first, int zero_grad, hipStream_t s) {
    if (n <= 0) return 0;
    if (momentum != 0.f && buf == nullptr) return -1;
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid), dim3(256), 0, s, p, g, buf, n, lr, momentum, nesterov, weight_decay, first, zero_grad);
    return MURCL_CHECK_LAUNCH();
}
