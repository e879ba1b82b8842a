// K10 / K11: PPO actor-critic head math (models/rlmil.py:66-127,152-184); the MLP / GRU / heads are the shared
// GEMM + GRU-gate kernels.
//
//   mu = sigmoid(z);  action = clamp(mu + std*eps, 0, 1)                         (act, rlmil.py:82-89)
//   logp = sum_k [ -((a-mu)/std)^2/2 ] - K log std - K/2 log 2pi                 (MultivariateNormal with
//          scale_tril = diag(action_var): `action_var` is used as a std, rlmil.py:84-85,90)
//   returns: discounted (gamma) over the rollout, then (R - mean)/(std_unbiased + 1e-5)      (rlmil.py:153-162)
//   loss = mean[ -min(r A, clip(r,1-e,1+e) A) + 0.5 MSE(v, R) - 0.01 H ],  r = exp(logp - logp_old), A = R - v   (:172-178)
#include "common.h"

#define LOG_2PI 1.8378770664093453f

// one thread per row: R rows x K actions (K <= 64)
__global__ void policy_head_fwd_kernel(const float* __restrict__ z, const float* __restrict__ eps,
                                       const float* __restrict__ act_in, float std_, int R, int K,
                                       float* __restrict__ mu, float* __restrict__ act_out, float* __restrict__ logp) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        const float m = 1.f / (1.f + expf(-z[(size_t)r * K + k]));
        float a;
        if (eps) {
            a = m + std_ * eps[(size_t)r * K + k];
            a = fminf(fmaxf(a, 0.f), 1.f);                     // relu then 1 - relu(1 - a)
            act_out[(size_t)r * K + k] = a;
        } else {
            a = act_in[(size_t)r * K + k];
        }
        mu[(size_t)r * K + k] = m;
        const float t = (a - m) / std_;
        acc += -0.5f * t * t;
    }
    logp[r] = acc - (float)K * logf(std_) - 0.5f * (float)K * LOG_2PI;
}
// dz[r,k] = dlogp[r] * (a - mu)/std^2 * mu (1 - mu)
__global__ void policy_head_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ act,
                                       const float* __restrict__ dlogp, float std_, int R, int K, float* __restrict__ dz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * K) return;
    const int r = i / K;
    const float m = mu[i];
    dz[i] = dlogp[r] * (act[i] - m) / (std_ * std_) * m * (1.f - m);
}
extern "C" int murcl_policy_head_fwd(const float* z, const float* eps, const float* act_in, float std_, int R, int K,
                                     float* mu, float* act_out, float* logp, hipStream_t s) {
    if (R <= 0) return 0;
    if ((eps == nullptr) == (act_in == nullptr)) return -1;      // exactly one of: sample (eps) / evaluate (act_in)
    hipLaunchKernelGGL(policy_head_fwd_kernel, dim3((R + 255) / 256), dim3(256), 0, s, z, eps, act_in, std_, R, K, mu, act_out, logp);
    return MURCL_CHECK_LAUNCH();
}
extern "C" int murcl_policy_head_bwd(const float* mu, const float* act, const float* dlogp, float std_, int R, int K,
                                     float* dz, hipStream_t s) {
    if (R <= 0) return 0;
    hipLaunchKernelGGL(policy_head_bwd_kernel, dim3((R * K + 255) / 256), dim3(256), 0, s, mu, act, dlogp, std_, R, K, dz);
    return MURCL_CHECK_LAUNCH();
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    const float t = red[0];
    __syncthreads();
    return t;
}

// rewards [T,B] -> normalised discounted returns [T,B] (single workgroup: T*B is a few thousand)
__global__ __launch_bounds__(256) void ppo_returns_kernel(const float* __restrict__ rewards, float gamma, int T, int B,
                                                          float* __restrict__ ret) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    float s = 0.f;
    for (int b = tid; b < B; b += 256) {
        float run = 0.f;
        for (int t = T - 1; t >= 0; --t) {
            run = rewards[(size_t)t * B + b] + gamma * run;
            ret[(size_t)t * B + b] = run;
            s += run;
        }
    }
    const int n = T * B;
    const float mean = block_sum_256(s, red) / (float)n;
    float v = 0.f;
    for (int i = tid; i < n; i += 256) { const float d = ret[i] - mean; v += d * d; }
    const float var = block_sum_256(v, red) / (float)(n - 1);            // torch.std: unbiased
    const float inv = 1.f / (sqrtf(var) + 1e-5f);
    for (int i = tid; i < n; i += 256) ret[i] = (ret[i] - mean) * inv;
}
extern "C" int murcl_ppo_returns(const float* rewards, float gamma, int T, int B, float* ret, hipStream_t s) {
    if (T <= 0 || B <= 0) return 0;
    hipLaunchKernelGGL(ppo_returns_kernel, dim3(1), dim3(256), 0, s, rewards, gamma, T, B, ret);
    return MURCL_CHECK_LAUNCH();
}

// Data-parallel form: the rollout rows are sharded by bag over ranks, but the returns are normalised with
// the mean / unbiased std of ALL ranks' returns (rlmil.py:162 sees the whole batch).  `raw` leaves the discounted returns
// un-normalised and the local (sum, sum of squares) in double precision; the caller all-reduces that pair (one 16-byte
// collective) and `finish` normalises with n_total = all ranks' T*B.
__device__ __forceinline__ double block_sum_256d(double v, double* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    const double t = red[0];
    __syncthreads();
    return t;
}
__global__ __launch_bounds__(256) void ppo_returns_raw_kernel(const float* __restrict__ rewards, float gamma, int T, int B,
                                                              float* __restrict__ ret, double* __restrict__ stats) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int b = tid; b < B; b += 256) {
        float run = 0.f;
        for (int t = T - 1; t >= 0; --t) {
            run = rewards[(size_t)t * B + b] + gamma * run;
            ret[(size_t)t * B + b] = run;
            s += (double)run;
            q += (double)run * (double)run;
        }
    }
    s = block_sum_256d(s, red);
    q = block_sum_256d(q, red);
    if (tid == 0) { stats[0] = s; stats[1] = q; }
}
__global__ void ppo_returns_finish_kernel(float* __restrict__ ret, int n, const double* __restrict__ stats, double n_total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double mean = stats[0] / n_total;
    double var = (stats[1] - stats[0] * mean) / (n_total - 1.0);          // torch.std: unbiased
    if (var < 0.0) var = 0.0;
    ret[i] = (float)(((double)ret[i] - mean) / (sqrt(var) + 1e-5));
}
extern "C" int murcl_ppo_returns_raw(const float* rewards, float gamma, int T, int B, float* ret, double* stats, hipStream_t s) {
    if (T <= 0 || B <= 0) return -1;
    hipLaunchKernelGGL(ppo_returns_raw_kernel, dim3(1), dim3(256), 0, s, rewards, gamma, T, B, ret, stats);
    return MURCL_CHECK_LAUNCH();
}
extern "C" int murcl_ppo_returns_finish(float* ret, int n, const double* stats, long n_total, hipStream_t s) {
    if (n <= 0 || n_total < 2) return -1;
    hipLaunchKernelGGL(ppo_returns_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ret, n, stats, (double)n_total);
    return MURCL_CHECK_LAUNCH();
}

// clipped-surrogate loss + gradients (single workgroup).  n_total >= n: the number of rollout rows over ALL ranks - the
// loss is a mean over them, so gradients carry 1/n_total and a SUM all-reduce of the parameter gradients gives the
// global mean's gradient; loss[0] is this rank's share of the global loss (the shares add up to it).
__global__ __launch_bounds__(256) void ppo_loss_kernel(const float* __restrict__ logp, const float* __restrict__ old_logp,
                                                       const float* __restrict__ value, const float* __restrict__ ret,
                                                       float eps_clip, float entropy, int n, float inv_n,
                                                       float* __restrict__ loss, float* __restrict__ dlogp,
                                                       float* __restrict__ dvalue) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    float acc = 0.f;
    for (int i = tid; i < n; i += 256) {
        const float ratio = expf(logp[i] - old_logp[i]);
        const float adv = ret[i] - value[i];                              // value detached here (rlmil.py:174)
        const float s1 = ratio * adv;
        const float rc = fminf(fmaxf(ratio, 1.f - eps_clip), 1.f + eps_clip);
        const float s2 = rc * adv;
        // nn.MSELoss() is a scalar mean broadcast into every element: its mean over the rows is the mean of d^2 itself
        const float dv = value[i] - ret[i];
        acc += -fminf(s1, s2) + 0.5f * dv * dv - 0.01f * entropy;
        // d/dlogp of -min(s1,s2): torch.min sends the gradient to s1 when s1 <= s2 (ties included), else to s2,
        // whose ratio-gradient is zero outside the clip range (clamp passes gradient on the closed interval)
        float g;
        if (s1 <= s2) g = -s1;                                            // d(ratio*adv)/dlogp = ratio*adv
        else g = (ratio >= 1.f - eps_clip && ratio <= 1.f + eps_clip) ? -s2 : 0.f;
        dlogp[i] = g * inv_n;
        dvalue[i] = dv * inv_n;                                           // d(0.5*mse)/dv_i, summed over the copies / their count
    }
    const float tot = block_sum_256(acc, red);
    if (tid == 0) loss[0] = tot * inv_n;
}
extern "C" int murcl_ppo_loss(const float* logp, const float* old_logp, const float* value, const float* ret, float eps_clip,
                              float entropy, int n, long n_total, float* loss, float* dlogp, float* dvalue, hipStream_t s) {
    if (n <= 0 || n_total < n) return -1;
    hipLaunchKernelGGL(ppo_loss_kernel, dim3(1), dim3(256), 0, s, logp, old_logp, value, ret, eps_clip, entropy, n,
                       1.f / (float)n_total, loss, dlogp, dvalue);
    return MURCL_CHECK_LAUNCH();
}
