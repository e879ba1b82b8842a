// PPO sampler as launch SEQUENCES behind one C call each (models/rlmil.py:66-127,152-184).
//
// The actor-critic is tiny (state MLP S -> 2048 -> H, GRU(H,H), actor H -> K, critic H -> 1; 3.7 M parameters) and its
// rollouts are a few hundred rows, so a sampling step or a PPO epoch is a chain of 6-40 short launches: the GPU time is
// microseconds per launch and what the step pays for is the HOST side of each launch.  Driven from Python (one ctypes
// call, one output allocation and one autograd node per launch) stage 2 of the MuRCL step was host-bound: ~950 launches,
// host enqueue time == step time.  Here the whole chain is enqueued natively:
//
//   murcl_ppo_act    one policy step:   2 encoder GEMMs, GRU cell (2 GEMMs + gates), fused actor head
//                    (GEMV + sigmoid + Gaussian sample + clamp + log-prob)                          -> 6 launches, 1 call
//   murcl_ppo_epoch  one K_epoch of PPO.update: evaluate() forward over the rollout, the fused
//                    heads + clipped-surrogate loss + their gradients, full backward (GRU through
//                    time, encoder), every weight / bias gradient ADDED into the caller's gradient
//                    buffers (FlatAdam's flat views)                                                -> 11 + 2T launches, 1 call
//
// All intermediates live in a caller-provided workspace (murcl_ppo_epoch_workspace).  The GEMMs are the library's own
// entry points (murcl_gemm_nt / murcl_gemm_tn); new kernels here are the fused heads.
#include "common.h"
#include "../../include/murcl_amd.h"

#define PS_LOG_2PI 1.8378770664093453f
#define PS_MAXK 16

__device__ __forceinline__ float ps_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- act: one wave per rollout row.  z_k = h . Wa[k] + ba[k]; mu = sigmoid(z); a = clamp(mu + std*eps, 0, 1);
// logp = sum_k -((a-mu)/std)^2/2 - K log std - K/2 log 2pi   (rlmil.py:82-90: action_var is used as scale_tril = a std)
__global__ __launch_bounds__(256) void ps_act_head_kernel(const float* __restrict__ h, const float* __restrict__ Wa,
                                                          const float* __restrict__ ba, const float* __restrict__ eps,
                                                          float std_, int R, int H, int K, float* __restrict__ action,
                                                          float* __restrict__ logp) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* hr = h + (size_t)r * H;
    // the K dots side by side: their loads and their wave reductions are independent chains (one after the other they were
    // K serial memory + shuffle latencies: 13 us for 10 actions)
    float p[PS_MAXK];
#pragma unroll
    for (int k = 0; k < PS_MAXK; ++k) p[k] = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
        const f32x4 a = *(const f32x4*)(hr + c);
        f32x4 b[PS_MAXK];
#pragma unroll
        for (int k = 0; k < PS_MAXK; ++k) b[k] = *(const f32x4*)(Wa + (size_t)min(k, K - 1) * H + c);   // branch-free: rows past K repeat the last
#pragma unroll
        for (int k = 0; k < PS_MAXK; ++k) p[k] += a[0] * b[k][0] + a[1] * b[k][1] + a[2] * b[k][2] + a[3] * b[k][3];
    }
#pragma unroll
    for (int k = 0; k < PS_MAXK; ++k) p[k] = ps_wave_sum(p[k]);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < PS_MAXK; ++k)
        if (k < K) {
            const float z = p[k] + ba[k];
            const float m = 1.f / (1.f + expf(-z));
            float a = m + std_ * eps[(size_t)r * K + k];
            a = fminf(fmaxf(a, 0.f), 1.f);                   // relu, then 1 - relu(1 - a)
            if (lane == 0) action[(size_t)r * K + k] = a;
            const float t = (a - m) / std_;
            acc += -0.5f * t * t;
        }
    if (lane == 0) logp[r] = acc - (float)K * logf(std_) - 0.5f * (float)K * PS_LOG_2PI;
}

// ---- evaluate heads + PPO loss + their gradients, one wave per row (rlmil.py:115-127,169-178):
//   z, v from hs; logp of the STORED action; ratio = exp(logp - logp_old); A = R - v (v detached);
//   row loss = -min(ratio A, clip(ratio) A) + 0.5 (v - R)^2 - 0.01 H       (MSELoss' scalar mean spread over the rows)
//   dlogp, dv carry 1/n_total; dz_k = dlogp (a_k - mu_k)/std^2 mu_k (1 - mu_k);  dhs = sum_k dz_k Wa[k] + dv Wc
// dzv [R, PS_MAXK+1]: dz_0..dz_{K-1}, then dv at column K (consumed by the head weight-gradient kernel).
__global__ __launch_bounds__(256) void ps_eval_head_kernel(const float* __restrict__ hs, const float* __restrict__ Wa,
                                                           const float* __restrict__ ba, const float* __restrict__ Wc,
                                                           const float* __restrict__ bc, const float* __restrict__ act,
                                                           const float* __restrict__ old_logp, const float* __restrict__ ret,
                                                           float std_, float eps_clip, float entropy, float inv_n, int R,
                                                           int H, int K, float* __restrict__ dhs, float* __restrict__ dzv,
                                                           float* __restrict__ lossrow) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* hr = hs + (size_t)r * H;
    float mu[PS_MAXK], da[PS_MAXK];
    // the K actor dots and the critic dot side by side (independent load and reduction chains, see ps_act_head_kernel)
    float p[PS_MAXK], pv = 0.f;
#pragma unroll
    for (int k = 0; k < PS_MAXK; ++k) p[k] = mu[k] = da[k] = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
        const f32x4 a = *(const f32x4*)(hr + c), bcv = *(const f32x4*)(Wc + c);
        pv += a[0] * bcv[0] + a[1] * bcv[1] + a[2] * bcv[2] + a[3] * bcv[3];
        f32x4 b[PS_MAXK];
#pragma unroll
        for (int k = 0; k < PS_MAXK; ++k) b[k] = *(const f32x4*)(Wa + (size_t)min(k, K - 1) * H + c);   // branch-free: rows past K repeat the last
#pragma unroll
        for (int k = 0; k < PS_MAXK; ++k) p[k] += a[0] * b[k][0] + a[1] * b[k][1] + a[2] * b[k][2] + a[3] * b[k][3];
    }
    pv = ps_wave_sum(pv);
#pragma unroll
    for (int k = 0; k < PS_MAXK; ++k) p[k] = ps_wave_sum(p[k]);
    float lp = 0.f;
#pragma unroll
    for (int k = 0; k < PS_MAXK; ++k)
        if (k < K) {
            const float z = p[k] + ba[k];
            const float m = 1.f / (1.f + expf(-z));
            const float d = act[(size_t)r * K + k] - m;
            mu[k] = m;
            da[k] = d;
            const float t = d / std_;
            lp += -0.5f * t * t;
        }
    lp += -(float)K * logf(std_) - 0.5f * (float)K * PS_LOG_2PI;
    const float v = pv + bc[0];
    const float R_ = ret[r];
    const float ratio = expf(lp - old_logp[r]);
    const float adv = R_ - v;
    const float s1 = ratio * adv;
    const float rc = fminf(fmaxf(ratio, 1.f - eps_clip), 1.f + eps_clip);
    const float s2 = rc * adv;
    const float dvr = v - R_;
    // torch.min sends the gradient to s1 when s1 <= s2 (ties included), else to s2, whose ratio-gradient is zero outside
    // the clip range (clamp passes the gradient on the closed interval)
    float g;
    if (s1 <= s2) g = -s1;
    else g = (ratio >= 1.f - eps_clip && ratio <= 1.f + eps_clip) ? -s2 : 0.f;
    const float dlogp = g * inv_n, dv = dvr * inv_n;
    float dz[PS_MAXK];
#pragma unroll
    for (int k = 0; k < PS_MAXK; ++k) dz[k] = (k < K) ? dlogp * da[k] / (std_ * std_) * mu[k] * (1.f - mu[k]) : 0.f;
    if (lane == 0) {
        lossrow[r] = (-fminf(s1, s2) + 0.5f * dvr * dvr - 0.01f * entropy) * inv_n;
#pragma unroll
        for (int k = 0; k < PS_MAXK; ++k)
            if (k < K) dzv[(size_t)r * (PS_MAXK + 1) + k] = dz[k];
        dzv[(size_t)r * (PS_MAXK + 1) + K] = dv;
    }
    float* dr = dhs + (size_t)r * H;
    for (int c = lane * 4; c < H; c += 256) {
        const f32x4 wc = *(const f32x4*)(Wc + c);
        f32x4 o = {dv * wc[0], dv * wc[1], dv * wc[2], dv * wc[3]};
        f32x4 w[PS_MAXK];
#pragma unroll
        for (int k = 0; k < PS_MAXK; ++k) w[k] = *(const f32x4*)(Wa + (size_t)min(k, K - 1) * H + c);   // dz[k] = 0 past K
#pragma unroll
        for (int k = 0; k < PS_MAXK; ++k) { o[0] += dz[k] * w[k][0]; o[1] += dz[k] * w[k][1]; o[2] += dz[k] * w[k][2]; o[3] += dz[k] * w[k][3]; }
        *(f32x4*)(dr + c) = o;
    }
}

// ---- head weight gradients.  Grid (K + 1 heads, H / 64 column chunks) (+ one block for the loss): a block sums
// dz[r,k] * hs[r, 64 columns] over all rows, four waves taking every fourth row with eight rows of loads in flight, and
// adds its slice of dWa[k,:] (k < K) or dWc (k = K); the column-chunk-0 block also adds the bias gradient.
// The extra block (blockIdx.x = K + 1) sums the row losses.
__global__ __launch_bounds__(256) void ps_head_wgrad_kernel(const float* __restrict__ hs, const float* __restrict__ dzv,
                                                            const float* __restrict__ lossrow, int R, int H, int K,
                                                            float* __restrict__ dWa, float* __restrict__ dba,
                                                            float* __restrict__ dWc, float* __restrict__ dbc,
                                                            float* __restrict__ loss) {
    __shared__ float red[256];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (k == K + 1) {
        if (blockIdx.y) return;
        float s = 0.f;
        for (int r = tid; r < R; r += 256) s += lossrow[r];
        red[tid] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
        if (tid == 0 && loss) loss[0] = red[0];
        return;
    }
    const int c = blockIdx.y * 64 + lane;
    float s = 0.f, sb = 0.f;
    if (c < H) {
        int r = wave;
        for (; r + 28 < R; r += 32) {                         // 8 rows of this wave per trip: 8 independent loads in flight
            float d[8], h[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                d[u] = dzv[(size_t)(r + 4 * u) * (PS_MAXK + 1) + k];
                h[u] = hs[(size_t)(r + 4 * u) * H + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += d[u] * h[u]; sb += d[u]; }
        }
        for (; r < R; r += 4) {
            const float d = dzv[(size_t)r * (PS_MAXK + 1) + k];
            s += d * hs[(size_t)r * H + c];
            sb += d;
        }
    }
    red[tid] = s;
    __syncthreads();
    if (wave == 0 && c < H) {
        const float t = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
        float* dW = (k < K) ? dWa + (size_t)k * H : dWc;
        dW[c] += t;
    }
    __syncthreads();
    if (blockIdx.y == 0) {                                   // bias gradient: every lane of a wave holds the same row sum
        red[tid] = sb;
        __syncthreads();
        if (tid == 0) {
            const float t = red[0] + red[64] + red[128] + red[192];
            if (k < K) dba[k] += t; else dbc[0] += t;
        }
    }
}

// parameter order = ActorCritic.parameters(): state_encoder.0.{weight,bias}, state_encoder.2.{weight,bias},
// gru.{weight_ih,weight_hh,bias_ih,bias_hh}_l0, actor.0.{weight,bias}, critic.0.{weight,bias}
enum { P_W1 = 0, P_B1, P_W2, P_B2, P_WIH, P_WHH, P_BIH, P_BHH, P_WA, P_BA, P_WC, P_BC, P_COUNT };
#define PS_E1 2048                       // width of the first encoder layer (rlmil.py:41)
#define PS_CHECK(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

static int ps_nt(const float* A, const float* B, float* C, int M, int N, int K, int epi, const float* bias, int accumulate,
                 hipStream_t s) {
    return murcl_gemm_nt(A, B, C, M, N, K, K, K, N, MURCL_F32, MURCL_F32, epi, bias, nullptr, 0, nullptr, nullptr, 0, nullptr,
                         accumulate, s);
}
static int ps_shape_ok(int S, int H, int K) {
    return S > 0 && H > 0 && K > 0 && K <= PS_MAXK && S % 32 == 0 && H % 32 == 0;
}

extern "C" long murcl_ppo_act_workspace(int B, int S, int H) {
    if (B <= 0) return 0;
    return (long)B * (PS_E1 + H + 3 * H + 3 * H + 3 * H) * 4;       // e1, e2, gi, gh, gates
}

// One sampling step of policy_old (rlmil.py:66-97).  hidden_prev [B,H] (NULL: zeros - restart_batch).
extern "C" int murcl_ppo_act(const float* const* params, int S, int H, int K, const float* state, const float* hidden_prev,
                             const float* eps, float std_, int B, float* hidden_new, float* action, float* logp, float* ws,
                             hipStream_t stream) {
    if (B <= 0) return 0;
    if (!ps_shape_ok(S, H, K)) return -1;
    float* e1 = ws;
    float* e2 = e1 + (size_t)B * PS_E1;
    float* gi = e2 + (size_t)B * H;
    float* gh = gi + (size_t)B * 3 * H;
    float* gates = gh + (size_t)B * 3 * H;
    PS_CHECK(ps_nt(state, params[P_W1], e1, B, PS_E1, S, MURCL_EPI_BIAS_RELU, params[P_B1], 0, stream));
    PS_CHECK(ps_nt(e1, params[P_W2], e2, B, H, PS_E1, MURCL_EPI_BIAS_RELU, params[P_B2], 0, stream));
    if (murcl_gru_step_supported(B, H, H)) {                                    // (hidden_prev NULL: the zero state)
        // the GRU cell as ONE launch: both products of a 16 x 16-unit tile of the three gate blocks, gates in the epilogue
        PS_CHECK(murcl_gru_step_fwd(e2, params[P_WIH], H, params[P_BIH], hidden_prev, params[P_WHH], params[P_BHH], hidden_new, nullptr,
                                    nullptr, B, H, stream));
    } else {
        PS_CHECK(ps_nt(e2, params[P_WIH], gi, B, 3 * H, H, MURCL_EPI_BIAS, params[P_BIH], 0, stream));
        if (hidden_prev) {
            PS_CHECK(ps_nt(hidden_prev, params[P_WHH], gh, B, 3 * H, H, MURCL_EPI_BIAS, params[P_BHH], 0, stream));
            PS_CHECK(murcl_gru_gates_fwd(gi, gh, hidden_prev, hidden_new, gates, B, H, 0, stream));
        } else {                                                      // W_hh . 0 + b_hh: one broadcast row
            PS_CHECK(murcl_gru_gates_fwd(gi, params[P_BHH], nullptr, hidden_new, gates, B, H, 1, stream));
        }
    }
    hipLaunchKernelGGL(ps_act_head_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, hidden_new, params[P_WA], params[P_BA], eps,
                       std_, B, H, K, action, logp);
    return MURCL_CHECK_LAUNCH();
}

extern "C" long murcl_ppo_epoch_workspace(int T, int B, int S, int H) {
    const long R = (long)T * B;
    if (R <= 0) return 0;
    long f = R * (2L * PS_E1 + 2L * H + 5L * 3 * H + 2L * H + (PS_MAXK + 1) + 1);   // e1,de1 | e2,de2 | gi,gh,gates,dgi,dgh | hs,dhs | dzv | lossrow
    f += (long)B * H;                                                                 // dhp (unused tail of the recurrence)
    f += 2L * H * 3 * H + (long)PS_E1 * H;                                            // W_ih^T, W_hh^T, W_2^T
    return f * 4;
}

// One epoch of PPO.update (rlmil.py:169-181) minus the optimizer step: forward of evaluate() over the rollout
// (states [T,B,S], actions [T,B,K], GRU from a zero hidden state), loss, backward; parameter gradients are ADDED to
// grads[] (same order as params[]; the caller zeroes them, all-reduces them across ranks if any, and steps Adam).
// n_total: rollout rows over all ranks (the loss is their mean).  loss_out (may be NULL): this rank's share of the loss.
// wt (may be NULL): {W_ih^T [H,3H], W_hh^T [H,3H], W_2^T [2048,H]} prepared by the caller (the optimizer's one launch of weight
// views per step); NULL: transposed here into the workspace, three launches.
static int ps_epoch(const float* const* params, float* const* grads, const float* const* wt, int S, int H, int K, const float* states,
                    const float* actions, const float* old_logp, const float* returns, int T, int B, long n_total,
                    float std_, float eps_clip, float entropy, float* ws, float* loss_out, hipStream_t stream) {
    const int R = T * B;
    if (R <= 0) return 0;
    if (!ps_shape_ok(S, H, K) || n_total < R) return -1;
    const size_t r = (size_t)R;
    float* e1 = ws;
    float* de1 = e1 + r * PS_E1;
    float* e2 = de1 + r * PS_E1;
    float* de2 = e2 + r * H;
    float* gi = de2 + r * H;
    float* gh = gi + r * 3 * H;
    float* gates = gh + r * 3 * H;
    float* dgi = gates + r * 3 * H;
    float* dgh = dgi + r * 3 * H;
    float* hs = dgh + r * 3 * H;
    float* dhs = hs + r * H;
    float* dzv = dhs + r * H;
    float* lossrow = dzv + r * (PS_MAXK + 1);
    float* dhp = lossrow + r;
    float* wih_ws = dhp + (size_t)B * H;
    float* whh_ws = wih_ws + (size_t)H * 3 * H;
    float* w2_ws = whh_ws + (size_t)H * 3 * H;
    const float* wih_t = wt ? wt[0] : wih_ws;
    const float* whh_t = wt ? wt[1] : whh_ws;
    const float* w2_t = wt ? wt[2] : w2_ws;
    const size_t bh = (size_t)B * H, b3 = (size_t)B * 3 * H;

    const bool fused = murcl_gru_step_supported(B, H, 0);    // one launch per GRU time step and direction
    // ---------------- forward (rlmil.py:103-112)
    PS_CHECK(ps_nt(states, params[P_W1], e1, R, PS_E1, S, MURCL_EPI_BIAS_RELU, params[P_B1], 0, stream));
    PS_CHECK(ps_nt(e1, params[P_W2], e2, R, H, PS_E1, MURCL_EPI_BIAS_RELU, params[P_B2], 0, stream));
    PS_CHECK(ps_nt(e2, params[P_WIH], gi, R, 3 * H, H, MURCL_EPI_BIAS, params[P_BIH], 0, stream));
    for (int t = 0; t < T; ++t) {
        if (t == 0) {
            PS_CHECK(murcl_gru_gates_fwd(gi, params[P_BHH], nullptr, hs, gates, B, H, 1, stream));
        } else if (fused) {
            PS_CHECK(murcl_gru_step_fwd(nullptr, nullptr, 0, gi + t * b3, hs + (t - 1) * bh, params[P_WHH], params[P_BHH], hs + t * bh,
                                        gates + t * b3, gh + t * b3, B, H, stream));
        } else {
            PS_CHECK(ps_nt(hs + (t - 1) * bh, params[P_WHH], gh + t * b3, B, 3 * H, H, MURCL_EPI_BIAS, params[P_BHH], 0, stream));
            PS_CHECK(murcl_gru_gates_fwd(gi + t * b3, gh + t * b3, hs + (t - 1) * bh, hs + t * bh, gates + t * b3, B, H, 0, stream));
        }
    }
    // ---------------- heads, loss, gradient into hs (rlmil.py:114-127,169-178)
    hipLaunchKernelGGL(ps_eval_head_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, hs, params[P_WA], params[P_BA], params[P_WC],
                       params[P_BC], actions, old_logp, returns, std_, eps_clip, entropy, 1.f / (float)n_total, R, H, K, dhs, dzv,
                       lossrow);
    PS_CHECK(MURCL_CHECK_LAUNCH());
    hipLaunchKernelGGL(ps_head_wgrad_kernel, dim3(K + 2, (H + 63) / 64), dim3(256), 0, stream, hs, dzv, lossrow, R, H, K, grads[P_WA], grads[P_BA],
                       grads[P_WC], grads[P_BC], loss_out);
    PS_CHECK(MURCL_CHECK_LAUNCH());
    // ---------------- backward through the GRU (time-reversed), then the encoder
    if (!wt) {
        PS_CHECK(murcl_transpose_cast(params[P_WIH], wih_ws, 3 * H, H, MURCL_F32, stream));
        if (T > 1) PS_CHECK(murcl_transpose_cast(params[P_WHH], whh_ws, 3 * H, H, MURCL_F32, stream));
        PS_CHECK(murcl_transpose_cast(params[P_W2], w2_ws, H, PS_E1, MURCL_F32, stream));
    }
    if (fused && T > 1) {
        // dh_{t-1} += dgh_t . W_hh, then step t-1's gate backward on the finished tile: one launch per step
        const int t1 = T - 1;
        PS_CHECK(murcl_gru_gates_bwd_into(dhs + t1 * bh, gates + t1 * b3, gh + t1 * b3, hs + (t1 - 1) * bh, dgi + t1 * b3, dgh + t1 * b3,
                                          dhs + (t1 - 1) * bh, B, H, 0, 1, stream));
        for (int t = T - 1; t >= 1; --t) {
            const int u = t - 1;                                                     // the step whose gates this launch differentiates
            PS_CHECK(murcl_gru_step_bwd(dgh + t * b3, whh_t, dhs + u * bh, gates + u * b3, u ? gh + u * b3 : params[P_BHH],
                                        u ? hs + (u - 1) * bh : nullptr, dgi + u * b3, dgh + u * b3, u ? dhs + (u - 1) * bh : dhp, B, H,
                                        u ? 0 : 1, u ? 1 : 0, stream));
        }
    } else {
        for (int t = T - 1; t >= 0; --t) {
            if (t == 0) {
                PS_CHECK(murcl_gru_gates_bwd_into(dhs, gates, params[P_BHH], nullptr, dgi, dgh, dhp, B, H, 1, 0, stream));
            } else {
                // dh_{t-1} += dh_t * z_t (direct path, added by the gate kernel) + dgh_t . W_hh (GEMM, accumulated in place)
                PS_CHECK(murcl_gru_gates_bwd_into(dhs + t * bh, gates + t * b3, gh + t * b3, hs + (t - 1) * bh, dgi + t * b3, dgh + t * b3,
                                                  dhs + (t - 1) * bh, B, H, 0, 1, stream));
                PS_CHECK(ps_nt(dgh + t * b3, whh_t, dhs + (t - 1) * bh, B, H, 3 * H, MURCL_EPI_NONE, nullptr, 1, stream));
            }
        }
    }
    PS_CHECK(murcl_colsum(dgh, grads[P_BHH], R, 3 * H, 3 * H, MURCL_F32, 1, stream));
    // the two encoder dgrads with ReLU' (the layer's saved output > 0) in the epilogue
    PS_CHECK(murcl_gemm_nt(dgi, wih_t, de2, R, H, 3 * H, 3 * H, 3 * H, H, MURCL_F32, MURCL_F32, MURCL_EPI_MASK, nullptr, e2, H, nullptr,
                           nullptr, 0, nullptr, 0, stream));
    PS_CHECK(murcl_gemm_nt(de2, w2_t, de1, R, PS_E1, H, H, H, PS_E1, MURCL_F32, MURCL_F32, MURCL_EPI_MASK, nullptr, e1, PS_E1, nullptr,
                           nullptr, 0, nullptr, 0, stream));
    // every weight gradient (+ the bias gradients that are column sums of its left operand) added in ONE launch
    murcl_tn_problem pr[4];
    int np = 0;
    auto add = [&](const float* A_, const float* B_, float* C_, float* cs, int M_, int N1_, int N2_) {
        murcl_tn_problem& q = pr[np++];
        q.A = A_; q.B = B_; q.C = C_; q.colsum_part = nullptr; q.colsum_out = cs;
        q.M = M_; q.N1 = N1_; q.N2 = N2_; q.lda = N1_; q.ldb = N2_; q.ldc = N2_; q.colsum_rows = 0; q.flags = 0; q.scale = 1.f;
    };
    add(de1, states, grads[P_W1], grads[P_B1], R, PS_E1, S);
    add(de2, e1, grads[P_W2], grads[P_B2], R, H, PS_E1);
    add(dgi, e2, grads[P_WIH], grads[P_BIH], R, 3 * H, H);
    if (T > 1) add(dgh + b3, hs, grads[P_WHH], nullptr, R - B, 3 * H, H);
    PS_CHECK(murcl_gemm_tn_grouped(pr, np, MURCL_F32, nullptr, 0, stream));
    return 0;
}

extern "C" int murcl_ppo_epoch(const float* const* params, float* const* grads, int S, int H, int K, const float* states,
                               const float* actions, const float* old_logp, const float* returns, int T, int B, long n_total,
                               float std_, float eps_clip, float entropy, float* ws, float* loss_out, hipStream_t stream) {
    return ps_epoch(params, grads, nullptr, S, H, K, states, actions, old_logp, returns, T, B, n_total, std_, eps_clip, entropy, ws,
                    loss_out, stream);
}
extern "C" int murcl_ppo_epoch_wt(const float* const* params, float* const* grads, const float* const* wt, int S, int H, int K,
                                  const float* states, const float* actions, const float* old_logp, const float* returns, int T, int B,
                                  long n_total, float std_, float eps_clip, float entropy, float* ws, float* loss_out,
                                  hipStream_t stream) {
    return ps_epoch(params, grads, wt, S, H, K, states, actions, old_logp, returns, T, B, n_total, std_, eps_clip, entropy, ws, loss_out,
                    stream);
}
