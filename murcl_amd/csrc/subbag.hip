// K12 / K13: sub-bag construction (utils/datasets.py:274-308) and mix-up (utils/datasets.py:263-271).
//
// subbag_select_kernel: one workgroup per bag.  Bit-exact restatement of the reference's float32 arithmetic
//     size_j = rint(float(n_j) * ratio)                (torch.round == round-half-even, datasets.py:286-287)
//     l_j    = floor(a_j * float(n_j - size_j))        (datasets.py:290), r_j = l_j + size_j
// followed by Python slice semantics c[l:r] (negative l / r wrap once, then clamp), concatenation over
// clusters, ascending sort and truncation to feat_size.  Cluster id lists partition the bag (k-means labels,
// wsi_processing/features_clustering.py:19-25), so the selected ids are distinct: they are marked in an LDS
// bitmap and emitted in ascending order by a popcount prefix scan - no comparison sort, no host round trip
// (the reference pays two .item() syncs per cluster per bag, datasets.py:294).
//
// subbag_gather_mix_kernel: out[b,r,:] = lam_b * X_b[idx_b[r]] + (1-lam_b) * X_p[idx_p[r]], p = perm[b]; rows past
// the selection count are zero (the reference zero-pads, datasets.py:300-303).  One wave per output row, 16 B per
// lane.  The f32 arithmetic keeps the reference's two products and one sum (no FMA contraction).
#include "common.h"

// The reference rounds every product before the following sum (datasets.py:268-270, 286-290): never fuse.
#pragma clang fp contract(off)

#define SB_MAX_WORDS 15360          // bitmap words in LDS (60 KiB): bags of up to 491,520 patches

__device__ __forceinline__ void py_slice(int n, int l, int r, int& lo, int& hi) {
    // Python: indices < 0 get len added once, then both are clamped to [0, len]
    if (l < 0) { l += n; if (l < 0) l = 0; } else if (l > n) l = n;
    if (r < 0) { r += n; if (r < 0) r = 0; } else if (r > n) r = n;
    lo = l;
    hi = r > l ? r : l;
}

__global__ __launch_bounds__(256) void subbag_select_kernel(const int* __restrict__ cluster_ids,
                                                            const int* __restrict__ cluster_off,   // [B][K+1] into cluster_ids
                                                            const int* __restrict__ n_patches,     // [B]
                                                            const float* __restrict__ ratio,       // [B] float32(feat_size / N_b)
                                                            const float* __restrict__ actions,     // [V][B][K]
                                                            int B, int K, int feat_size, int* __restrict__ idx_out,
                                                            int* __restrict__ count_out) {
    __shared__ unsigned bitmap[SB_MAX_WORDS];
    __shared__ int wsum[256];
    // one workgroup per (view, bag): the bag's tables by b, its actions / outputs by vb = view * B + b
    const int vb = blockIdx.x, b = vb % B, tid = threadIdx.x;
    const int N = n_patches[b];
    const int words = (N + 31) >> 5;
    for (int w = tid; w < words; w += 256) bitmap[w] = 0u;
    __syncthreads();
    const int* off = cluster_off + (size_t)b * (K + 1);
    const float rt = ratio[b];
    // a wave per cluster (wave w: clusters w, w + 4, ...): the clusters' offset -> id-list -> bitmap chains are independent, so
    // the four waves' memory round trips overlap (walked one cluster after the other by the whole workgroup they were K serial
    // round trips: 10 us per call at K = 10)
    const int lane = tid & 63, wave = tid >> 6;
    for (int j = wave; j < K; j += 4) {
        const int beg = off[j], n = off[j + 1] - beg;
        const int size = (int)rintf((float)n * rt);                                      // single f32 products:
        const int l = (int)floorf(actions[(size_t)vb * K + j] * (float)(n - size));      // nothing to contract
        int lo, hi;
        py_slice(n, l, l + size, lo, hi);
        for (int t = lo + lane; t < hi; t += 64) {
            const int id = cluster_ids[beg + t];
            atomicOr(&bitmap[id >> 5], 1u << (id & 31));
        }
    }
    __syncthreads();
    // ascending emission: thread t owns a contiguous run of words; exclusive prefix of popcounts over threads
    const int per = (words + 255) / 256;
    const int w0 = tid * per, w1 = min(words, w0 + per);
    int cnt = 0;
    for (int w = w0; w < w1; ++w) cnt += __popc(bitmap[w]);
    int incl = cnt;                                  // inclusive scan: lane shuffles inside a wave, then the four wave totals
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int t = wsum[w];
        if (w < wave) base += t;
        total += t;
    }
    int pos = base + incl - cnt;
    int* out = idx_out + (size_t)vb * feat_size;
    for (int w = w0; w < w1 && pos < feat_size; ++w) {
        unsigned m = bitmap[w];
        while (m && pos < feat_size) {
            const int bit = __ffs(m) - 1;
            out[pos++] = (w << 5) + bit;
            m &= m - 1;
        }
    }
    const int kept = min(total, feat_size);
    for (int r = kept + tid; r < feat_size; r += 256) out[r] = -1;      // zero-padded rows
    if (tid == 0) count_out[vb] = kept;
}

extern "C" int murcl_subbag_select(const int* cluster_ids, const int* cluster_off, const int* n_patches,
                                   const float* ratio, const float* actions, int views, int B, int K, int feat_size,
                                   int max_patches, int* idx_out, int* count_out, hipStream_t stream) {
    if (B <= 0 || views <= 0) return 0;
    if (K <= 0 || feat_size <= 0 || max_patches > SB_MAX_WORDS * 32) return -1;
    hipLaunchKernelGGL(subbag_select_kernel, dim3(views * B), dim3(256), 0, stream, cluster_ids, cluster_off, n_patches, ratio,
                       actions, B, K, feat_size, idx_out, count_out);
    return MURCL_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------ gather (+ mix-up)
// A wave takes SG_ROWS consecutive output rows; with SG_ROWS > 1 their (wave-uniform) index / partner / lambda look-ups run as
// independent scalar chains and all 2 x SG_ROWS source rows are requested before the first is consumed.  Measured (round 3, 64 x 8192
// raw patches): 1 row per wave - many small waves, latency hidden by occupancy - is the fastest (T = 6 step 5.03-5.05 ms against
// 5.09 with 4 rows and 5.21-5.34 with 8), so that is the default.
#ifndef SG_ROWS
#define SG_ROWS 1
#endif
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void subbag_gather_mix_kernel(const TI* __restrict__ feats,
                                                                const long* __restrict__ bag_row_off,
                                                                const int* __restrict__ idx,
                                                                const float* __restrict__ lam,
                                                                const int* __restrict__ perm, TO* __restrict__ out,
                                                                int VB, int B, int feat_size, int d, int bag_lo, int n_out) {
    // bag_lo / n_out (round 6, batch-global mix-up of a sharded step): only the bags [bag_lo, bag_lo + n_out) of every view are
    // WRITTEN (out [views][n_out][feat_size][d]); idx / lam / perm / bag_row_off cover all B bags of the batch - a partner may be any
    // bag of the global batch, gathered from the replicated store (VB here = views * n_out output bags)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long total = (long)VB * feat_size;
    const long row0 = ((long)blockIdx.x * 4 + wave) * SG_ROWS;    // first output row (view * B + b) * feat_size + r of this wave
    if (row0 >= total) return;
    const bool mix = lam != nullptr;
    const TI* s0[SG_ROWS];
    const TI* s1[SG_ROWS];
    float l0[SG_ROWS];
    bool on0[SG_ROWS], on1[SG_ROWS], live[SG_ROWS];
#pragma unroll
    for (int u = 0; u < SG_ROWS; ++u) {
        const long row = min(row0 + u, total - 1);
        live[u] = row0 + u < total;
        const int ob = (int)(row / feat_size), r = (int)(row - (long)ob * feat_size);       // output bag (view * n_out + local bag)
        const int b = bag_lo + ob % n_out, v0 = (ob / n_out) * B, vb = v0 + b;             // the bag's rows by b; its view's block of idx / lam / perm by v0
        const int i0 = idx[(size_t)vb * feat_size + r];
        const int pb = mix ? perm[vb] : b;                          // mix-up partner: a bag of the same view
        const int i1 = mix ? idx[(size_t)(v0 + pb) * feat_size + r] : -1;
        l0[u] = mix ? lam[vb] : 1.f;
        on0[u] = i0 >= 0;
        on1[u] = i1 >= 0;
        s0[u] = feats + (bag_row_off[b] + (i0 < 0 ? 0 : i0)) * d;
        s1[u] = feats + (bag_row_off[pb] + (i1 < 0 ? 0 : i1)) * d;
    }
    for (int c = lane * 8; c < d; c += 512) {
        float x[SG_ROWS][8], y[SG_ROWS][8];
#pragma unroll
        for (int u = 0; u < SG_ROWS; ++u) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[u][e] = y[u][e] = 0.f;
            if (on0[u]) load8<TI>(s0[u] + c, x[u]);
            if (mix && on1[u]) load8<TI>(s1[u] + c, y[u]);
        }
#pragma unroll
        for (int u = 0; u < SG_ROWS; ++u) {
            if (!live[u]) continue;
            float v[8];
            if (mix) {
                const float l1 = 1.f - l0[u];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = l0[u] * x[u][e];
                    const float cc = l1 * y[u][e];
                    v[e] = a + cc;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = x[u][e];
            }
            store8<TO>(out + (row0 + u) * d + c, v);
        }
    }
}
extern "C" int murcl_subbag_gather_mix_rows(const void* feats, const long* bag_row_off, const int* idx, const float* lam,
                                            const int* perm, void* out, int views, int B, int feat_size, int d, int bag_lo, int n_out,
                                            int dtype_in, int dtype_out, hipStream_t stream);
extern "C" int murcl_subbag_gather_mix(const void* feats, const long* bag_row_off, const int* idx, const float* lam,
                                       const int* perm, void* out, int views, int B, int feat_size, int d, int dtype_in,
                                       int dtype_out, hipStream_t stream) {
    return murcl_subbag_gather_mix_rows(feats, bag_row_off, idx, lam, perm, out, views, B, feat_size, d, 0, B, dtype_in, dtype_out, stream);
}
extern "C" int murcl_subbag_gather_mix_rows(const void* feats, const long* bag_row_off, const int* idx, const float* lam,
                                            const int* perm, void* out, int views, int B, int feat_size, int d, int bag_lo, int n_out,
                                            int dtype_in, int dtype_out, hipStream_t stream) {
    if (B <= 0 || feat_size <= 0 || views <= 0 || n_out <= 0) return 0;
    if (d % 8 || bag_lo < 0 || bag_lo + n_out > B) return -1;
    if ((lam == nullptr) != (perm == nullptr)) return -1;
    const long rows = (long)views * n_out * feat_size;
    dim3 grid((unsigned)((rows + 4 * SG_ROWS - 1) / (4 * SG_ROWS)));
#define GM(TI, TO) hipLaunchKernelGGL((subbag_gather_mix_kernel<TI, TO>), grid, dim3(256), 0, stream, (const TI*)feats, \
                                      bag_row_off, idx, lam, perm, (TO*)out, views * n_out, B, feat_size, d, bag_lo, n_out)
    if (dtype_in == MURCL_DTYPE_F32 && dtype_out == MURCL_DTYPE_F32) GM(float, float);
    else if (dtype_in == MURCL_DTYPE_F32 && dtype_out == MURCL_DTYPE_BF16) GM(float, bf16_t);
    else if (dtype_in == MURCL_DTYPE_BF16 && dtype_out == MURCL_DTYPE_BF16) GM(bf16_t, bf16_t);
    else if (dtype_in == MURCL_DTYPE_BF16 && dtype_out == MURCL_DTYPE_F32) GM(bf16_t, float);
    else return -1;
#undef GM
    return MURCL_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------ stand-alone mix-up
template <typename T>
__global__ void mixup_kernel(const T* __restrict__ x, const float* __restrict__ lam, const int* __restrict__ perm,
                             T* __restrict__ out, int B, long per_bag) {
    const int b = blockIdx.y;
    const float l0 = lam[b], l1 = 1.f - l0;
    const T* x0 = x + (size_t)b * per_bag;
    const T* x1 = x + (size_t)perm[b] * per_bag;
    T* o = out + (size_t)b * per_bag;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_bag; i += (long)gridDim.x * blockDim.x) {
        const float a = l0 * to_f<T>(x0[i]);
        const float c = l1 * to_f<T>(x1[i]);
        o[i] = from_f<T>(a + c);
    }
}
extern "C" int murcl_mixup(const void* x, const float* lam, const int* perm, void* out, int B, long per_bag, int dtype,
                           hipStream_t stream) {
    if (B <= 0 || per_bag <= 0) return 0;
    int gx = (int)((per_bag + 255) / 256);
    if (gx > 1024) gx = 1024;
    dim3 grid(gx, B);
    if (dtype == MURCL_DTYPE_F32)
        hipLaunchKernelGGL(mixup_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, lam, perm, (float*)out, B, per_bag);
    else if (dtype == MURCL_DTYPE_BF16)
        hipLaunchKernelGGL(mixup_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)x, lam, perm, (bf16_t*)out, B, per_bag);
    else
        return -1;
    return MURCL_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------ every random number of a step
// One launch for the draws a MuRCL training step makes before anything is computed (train_MuRCL.py:235,256-258: uniform window
// positions; utils/datasets.py:265-267: per view lambda = alpha + U(0,1)(1 - alpha) per bag and a uniform random permutation of
// the bags; models/rlmil.py:85-86: the sampler's N(0,1) noise) - torch.rand x2, two elementwise ops, an argsort (a radix sort, an
// arange and three copies) and torch.randn otherwise.  Counter-based: every value is a pure function of (seed, stream, index)
// through splitmix64, so a step's draws do not depend on the launch geometry.
//   uni[i]  = 24 random bits / 2^24 in [0,1)                    (stream 0)
//   nrm[i]  = sqrt(-2 ln u1) cos(2 pi u2), u1 in (0,1]          (stream 1: Box-Muller, one value per word)
//   lam[v,b] = alpha + uni * (1 - alpha)                         (stream 2)
//   perm[v,:] = the argsort of B random 32-bit keys (ties by index): uniform over permutations up to key ties (~B^2 / 2^33)
// grid: n_views workgroups for (lam, perm), then ceil((n_uni + n_nrm) / 1024) for the element streams.  B <= MURCL_DRAWS_MAX_B.
#define SD_STREAM(seed, k) ((seed) ^ (0xA0761D6478BD642Full * (unsigned long long)((k) + 1)))
__global__ __launch_bounds__(256) void step_draws_kernel(unsigned long long seed, float* __restrict__ uni, long n_uni,
                                                         float* __restrict__ nrm, long n_nrm, float* __restrict__ lam,
                                                         int* __restrict__ perm, int n_views, int B, float alpha) {
    __shared__ unsigned keys[MURCL_DRAWS_MAX_B];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < n_views) {
        const int v = blockIdx.x;
        for (int b = tid; b < B; b += 256) {
            const unsigned long long r = murcl_drop_word(SD_STREAM(seed, 2), (long)v * B + b);
            lam[(size_t)v * B + b] = alpha + (float)(r >> 40) * (1.f / 16777216.f) * (1.f - alpha);
            keys[b] = (unsigned)r;
        }
        __syncthreads();
        for (int b = tid; b < B; b += 256) {
            const unsigned k = keys[b];
            int rank = 0;
            for (int j = 0; j < B; ++j) rank += (keys[j] < k) || (keys[j] == k && j < b);
            perm[(size_t)v * B + rank] = b;
        }
        return;
    }
    const long i0 = ((long)blockIdx.x - n_views) * 1024 + tid;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long i = i0 + 256 * u;
        if (i < n_uni) {
            uni[i] = (float)(murcl_drop_word(SD_STREAM(seed, 0), i) >> 40) * (1.f / 16777216.f);
        } else if (i < n_uni + n_nrm) {
            const unsigned long long r = murcl_drop_word(SD_STREAM(seed, 1), i - n_uni);
            const float u1 = ((float)(r >> 40) + 1.f) * (1.f / 16777216.f), u2 = (float)((r >> 16) & 0xFFFFFFull) * (1.f / 16777216.f);
            nrm[i - n_uni] = sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
        }
    }
}
extern "C" int murcl_step_draws(unsigned long long seed, float* uni, long n_uni, float* nrm, long n_nrm, float* lam, int* perm,
                                int n_views, int B, float alpha, hipStream_t stream) {
    if (n_uni < 0 || n_nrm < 0 || n_views < 0 || (n_views > 0 && (B <= 0 || B > MURCL_DRAWS_MAX_B))) return -1;
    const long blocks = n_views + (n_uni + n_nrm + 1023) / 1024;
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL(step_draws_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, seed, uni, n_uni, nrm, n_nrm, lam, perm,
                       n_views, B, alpha);
    return MURCL_CHECK_LAUNCH();
}
