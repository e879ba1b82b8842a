/* murcl_amd C-ABI: the MI355X (gfx950) drop-in boundary for MuRCL's data-parallel hot path.
 *
 * The reference (wwu98934/MuRCL) is pure Python/PyTorch and has no FFI layer; its boundary for
 * this path is the nn.Module API of models/{abmil,clam,dsmil,cl,rlmil}.py, utils/losses.py and
 * utils/datasets.{get_feats,mixup}.  murcl_amd mirrors those modules in Python (murcl_amd/models,
 * murcl_amd/utils) and every tensor op inside them lands on one of the entry points below, loaded
 * with ctypes from libmurcl_amd.so.  Each entry cites the reference lines it replaces (paths
 * relative to the reference repo).
 *
 * Conventions: plain device pointers and sizes, no torch types; all tensors row-major and
 * contiguous unless a leading dimension is given; `stream` is the hipStream_t to launch on (the
 * caller passes torch.cuda.current_stream()); nothing allocates, frees or synchronises; workspaces
 * are caller-provided.  Return 0 on success, a hipError_t (>0) if a launch failed, <0 for
 * unsupported arguments.  dtype codes: 0 = f32, 1 = bf16 (f32 accumulate everywhere); murcl_gemm_nt
 * (dtype_in) and murcl_gemm_tn / murcl_gemm_tn_ws also take 2 = f32 tensors whose products run as a 3-term bf16 split on the bf16
 * matrix pipe (hi + mid + lo, six MFMAs per product, f32-level accuracy at 6/16 of the exact-f32 MFMA time).
 */
#ifndef MURCL_AMD_H
#define MURCL_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* murcl_stream_t;

#define MURCL_F32 0
#define MURCL_BF16 1

/* epilogues of murcl_gemm_nt */
#define MURCL_EPI_NONE 0
#define MURCL_EPI_BIAS 1        /* + bias[n]                                  */
#define MURCL_EPI_BIAS_RELU 2   /* relu(. + bias[n])                          */
#define MURCL_EPI_MASK 3        /* . * (mask[m][n] > 0)           (ReLU')     */
#define MURCL_EPI_RANK1_MASK 4  /* (. + rowscale[m]*rank1[m/rows_per_bag][n]) * (mask > 0) */

/* C[M,N] = epi(A[M,K] . B[N,K]^T): nn.Linear forward (abmil.py:12-21,23-27,29-32; clam.py:69,40-48;
 * dsmil.py:9,15,55-59; rlmil.py:41-54,199-200) and its input gradient (B = W^T).  K must be a
 * multiple of 128 bytes of dtype_in.  colsum_ws ([ceil(M/128),N] f32, may be NULL) receives per
 * row-tile column sums of the output (bias gradients, reduced by murcl_colsum). */
int murcl_gemm_nt(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                  int dtype_in, int dtype_out, int epilogue, const float* bias, const void* mask, int ldmask,
                  const float* rowscale, const float* rank1, int rows_per_bag, float* colsum_ws, int accumulate,
                  murcl_stream_t stream);

/* C[N1,N2] (f32, pre-zeroed or accumulated into) += A[M,N1]^T . B[M,N2]: weight gradients, i.e. what
 * autograd's mm_backward computes for every Linear above; reduction over the patch dimension M is
 * split over `splits` workgroup groups (<=0: auto).  colsum_out (may be NULL) [N1] += column sums of A: the bias
 * gradient of the same layer, from the same pass over A (one more MFMA per fragment against a fragment of ones). */
int murcl_gemm_tn(const void* A, const void* B, float* C, int M, int N1, int N2, int lda, int ldb, int ldc,
                  int dtype, int splits, float* colsum_out, murcl_stream_t stream);
/* The same with a caller-provided workspace (murcl_gemm_tn_workspace_bytes; 0 = this shape needs none): the big bf16
 * weight gradients (N1, N2 multiples of 256, M >= 16384) run on 256 x 256 tiles whose per-split partial sums go to the
 * workspace as plain stores and are added to C by a second launch - no float atomics; other shapes behave as murcl_gemm_tn. */
long murcl_gemm_tn_workspace_bytes(int M, int N1, int N2, int dtype);
int murcl_gemm_tn_ws(const void* A, const void* B, float* C, int M, int N1, int N2, int lda, int ldb, int ldc, int dtype,
                     int splits, float* colsum_out, float* ws, long ws_bytes, const float* colsum_part, int colsum_rows,
                     murcl_stream_t stream);
/* colsum_part (may be NULL) [colsum_rows][N1] f32: instead of the column sums of A, colsum_out += the sum of these rows -
 * the partial bias-gradient rows murcl_panel_gemm leaves in its colsum_ws when called without colsum_out - folded into the
 * launch that adds up the workspace (a small launch of its own on the other paths). */

/* Several weight gradients C_g[N1_g,N2_g] += A_g[M_g,N1_g]^T . B_g[M_g,N2_g] as ONE launch of the square-tile kernel + ONE
 * reduce launch: the backward of the three encoder nn.Linear layers of abmil.py:12-21 (and of CLAM-SB's fc + gate pair,
 * clam.py:69-72), deferred until the last input gradient of the pass exists.  The (product, tile) pairs share one round of
 * workgroups (one per CU), so the per-launch fixed costs and the partial-tile traffic are paid once.  `probs` is a HOST array of
 * n <= 4 descriptors (f32 products of at most 512 rows each - bag-level / rollout-level layers - run as ONE launch of the 32 x 32
 * single-writer kernel instead, no workspace; of the flags they take MURCL_TN_OVERWRITE); colsum_part / colsum_rows / colsum_out as in murcl_gemm_tn_ws (colsum_out without colsum_part: column
 * sums of A_g by their own launch; not with flags).  flags / scale: applied by the reduce launch (grouped path only: a group with
 * flags that is not eligible returns -1).  Products the square-tile kernel does not take (see murcl_gemm_tn_ws), or a workspace below
 * murcl_gemm_tn_grouped_workspace_bytes (0 = the group is not eligible), run one by one through murcl_gemm_tn_ws. */
typedef struct murcl_tn_problem {
    const void* A; const void* B; float* C;
    const float* colsum_part; float* colsum_out;
    int M, N1, N2, lda, ldb, ldc, colsum_rows;
    int flags;          /* MURCL_TN_* below; 0 = C (and colsum_out) accumulated into */
    float scale;        /* with MURCL_TN_SCALE: the product (and the column sums) times this factor */
} murcl_tn_problem;
#define MURCL_TN_OVERWRITE 1    /* C = product, colsum_out = sums (no read of either: the caller need not zero them) */
#define MURCL_TN_DEINTERLEAVE 2 /* rows of the product arrive as 16-row blocks alternating between two halves (the gate pair of
                                 * murcl_panel_gemm epilogue 5: a-rows, b-rows, a-rows ...): row r of the product is written to row
                                 * ((r >> 4) & 1) * N1/2 + (r >> 5) * 16 + (r & 15) of C, i.e. C = [dWa; dWb] in natural order */
#define MURCL_TN_SCALE 4
long murcl_gemm_tn_grouped_workspace_bytes(const murcl_tn_problem* probs, int n, int dtype);
int murcl_gemm_tn_grouped(const murcl_tn_problem* probs, int n, int dtype, float* ws, long ws_bytes, murcl_stream_t stream);

/* Weight-stationary bf16 variant of murcl_gemm_nt for the patch-level layers (M = bags*patches rows, K in
 * {512,128}): same Linear forward / input-gradient as above (abmil.py:12-21,23-24 and their autograd), with
 * ReLU' taken from a 1-bit-per-element mask ([M,N/8] bytes; bit e of byte c <=> column 8c+e of the forward
 * output was > 0) that the forward epilogue (epilogue 0) can emit through bitmask_out.
 * epilogue: 0 = relu(.+bias) (K=512), 1 = . * mask (K=512), 2 = (. + rowscale[m]*rank1[m/rows_per_bag][n]) * mask
 * (K=128 with N=512, or K=512; with bias != NULL - the [M/rows_per_bag][2] soft-max statistics (m, l) of murcl_abmil_pool_decoder /
 * _combine - rowscale holds the RAW attention scores and the row scale is exp(rowscale[m] - m_bag) / (l_bag sqrt(rows_per_bag)),
 * i.e. abmil.py:40-41's A formed in the epilogue), 3 = . + bias (K=512), 4 = CLAM's gated attention score without the pre-activations
 * (clam.py:55-60; K=512, forward-only calls): W / bias hold attention_a and attention_b interleaved in 16-row blocks (rows
 * 32g..32g+15 = attention_a[16g..], rows 32g+16..32g+31 = attention_b[16g..]), rank1 = attention_c's weight at the a-rows ([N]
 * f32), and the ONLY output is colsum_ws = [N/32][M] f32 partial scores, partial[g][m] = sum over d in 16g..16g+15 of
 * tanh(a_d) sigmoid(b_d) c_d for row m, plus rowscale[0] (attention_c's bias; rowscale may be NULL) in partial[0] (C may be NULL; colsum_out must be NULL).  colsum_out ([N] f32, may be NULL) receives the column sums of the output (bias gradient): overwritten, or added to
 * when colsum_accumulate != 0 (accumulation straight into a gradient buffer); the workgroups' partial sums pass through
 * colsum_ws (256*N floats, required with colsum_out) and a second small launch adds them up.  walk_reverse bit 0: the
 * row tiles are visited from the last to the first (same result; use it when the kernel that has just produced A
 * walked forward, so that this pass starts on the rows that are still in the Infinity Cache); bit 1 (K=512): A is
 * loaded with the non-temporal cache policy (read once: it should not displace the output from the Infinity Cache); bit 2 (K=512,
 * epilogues 0 / 1 without Dropout): W is in FRAGMENT ORDER - a 16-row block stored as 16 k-steps x 64 lanes x 8 elements, lane
 * (q4, r16) of k-step kk holding elements [(kk + 16 q4) * 8, +8) of row r16 (what murcl_cast_batch writes for a job with transpose
 * bit 1) - so that the kernel's weight slice is one coalesced 1-KiB load per k-step beside its first tiles.  murcl_abmil_pool_fwd /
 * _bwd take Wa the same way with bit 1 of their exact_tanh argument (bf16).
 * murcl_panel_gemm_supported tells whether a shape is covered (else use murcl_gemm_nt). */
int murcl_panel_gemm_supported(int M, int N, int K, int epilogue, int rows_per_bag);
/* colsum_ws given WITHOUT colsum_out: the partial rows stay in colsum_ws ([murcl_panel_gemm_colsum_rows][N] f32) and no second
 * launch runs; the caller adds them up (murcl_gemm_tn_ws's colsum_part: the weight gradient of the same layer comes next). */
int murcl_panel_gemm_colsum_rows(int M, int N, int K, int epilogue);
int murcl_panel_gemm(const void* A, const void* W, void* C, int M, int N, int K, int epilogue, const float* bias,
                     void* bitmask_out, const void* bitmask_in, const float* rowscale, const float* rank1,
                     int rows_per_bag, float* colsum_out, int colsum_accumulate, float* colsum_ws /* [256*N] */,
                     int walk_reverse, murcl_stream_t stream);
/* The same with two more things for CLAM-SB's training forward (clam.py:69-72,40-60):
 * epilogue 5 (K=512) = epilogue 4's partial scores AND the gate pre-activations: C [M,N] = A W^T + bias in the interleaved
 * column order of W (what murcl_gated_score_bwd_il reads back), colsum_ws = [N/32][M] partial scores as for epilogue 4;
 * 0 < keep_p < 1 = nn.Dropout(1 - keep_p) inside the epilogue, masks never materialised (the counter-based masks of
 * murcl_dropout_mask: the same seed gives the same mask in every kernel): epilogue 0 with bitmask_out - the mask of seed_a over
 * [M,N] applied behind the ReLU, bitmask_out records what survives (= murcl_dropout_relu_bitmask on the epilogue's output, without
 * that pass); epilogue 5 - the masks of seed_a / seed_b over [M,N/2] applied to tanh(a) and sigmoid(b) inside the partial scores
 * (murcl_gated_score_fwd's seeded form; C stays the un-dropped pre-activations).  Other epilogues: keep_p must be 0 or 1. */
int murcl_panel_gemm_drop(const void* A, const void* W, void* C, int M, int N, int K, int epilogue, const float* bias,
                          void* bitmask_out, const void* bitmask_in, const float* rowscale, const float* rank1,
                          int rows_per_bag, float* colsum_out, int colsum_accumulate, float* colsum_ws, int walk_reverse,
                          float keep_p, unsigned long long seed_a, unsigned long long seed_b, murcl_stream_t stream);

/* Launch policy of the persistent kernels (csrc/runtime.hip): the number of CUs the encoder-sized launches size their ONE round of
 * workgroups for - 256 by default.  A data-parallel step whose RCCL collectives overlap its backward pass (the multi-GPU form of
 * train_MuRCL.py:145's nn.DataParallel) sets 256 - (CUs left to RCCL's channels): with a static share per workgroup a launch that
 * cannot place ALL its workgroups at once runs a second round, i.e. takes twice as long.  murcl_set_cu_budget returns the budget in
 * force (a multiple of 8 in [64, 256]). */
int murcl_cu_budget(void);
int murcl_set_cu_budget(int cus);

/* Box calibration for bench.py (csrc/runtime.hip; not on the product path, no reference counterpart): a streaming copy of `bytes`
 * (multiple of 16) with 16-byte accesses, and a register-only loop of 256 CUs x 8 waves x iters x 4 v_mfma_f32_16x16x32_bf16 (16384 FLOP
 * each) writing 256*512 floats - timed by the caller, they say what HBM rate and matrix clock THIS box sustains. */
int murcl_calib_copy(const void* src, void* dst, long bytes, murcl_stream_t stream);
int murcl_calib_mfma_bf16(float* out_256x512, int iters, murcl_stream_t stream);

/* K2 -- ABMIL attention pooling, abmil.py:38-42:  scores[b,n] = wb.tanh(Wa H[b,n]+ba)+bb,
 * A = softmax_N(scores)/sqrt(N), M[b] = A[b].H[b];  ml[b] = (max, sum exp) of the soft-max.
 * H [B,N,L] and Wa [D,L] in `dtype`; L = 512, D = 128.  part_ws: B*n_chunks*(L+4) floats with
 * n_chunks from murcl_abmil_pool_workspace. */
int murcl_abmil_pool_workspace(int B, int N, int dtype, int* chunk_rows, int* n_chunks);
int murcl_abmil_pool_fwd(const void* H, const void* Wa, const float* ba, const float* wb, const float* bb,
                         float* scores, float* A, float* M, float* ml, float* part_ws, int B, int N, int L, int D,
                         int dtype, int exact_tanh, murcl_stream_t stream);
/* The per-bag merge of murcl_abmil_pool_fwd on its own (chunk partials -> A, M, ml): pass A = M = ml = NULL to
 * murcl_abmil_pool_fwd to get the partials + raw scores only (ONE launch: what a training step runs since round 6 - the merge
 * then happens inside the decoder launch below, and the normalised attention row is formed by the backward pass or, for
 * `last_attention`, by this entry on demand). */
int murcl_abmil_pool_combine(const float* scores, const float* part_ws, float* A, float* M, float* ml, int B, int N,
                             int dtype, murcl_stream_t stream);
/* K3 with K2's merge on load, abmil.py:29-32,43-44: out [B,Lout] = relu?(M Wd^T + bd) with M formed from the chunk partials
 * part_ws of murcl_abmil_pool_fwd while the product loads its A operand; M [B,L] and ml [B,2] are written as by-products (the
 * backward pass reads them).  f32 throughout (exact-f32 MFMA), Wd [Lout,L], L = 512, Lout a multiple of 16.  Returns -1 when the
 * shape is outside the kernel (more than 512 chunks per bag): callers then run murcl_abmil_pool_combine + murcl_gemm_nt. */
int murcl_abmil_pool_decoder(const float* part_ws, const float* Wd, const float* bd, float* M, float* ml, float* out, int B,
                             int N, int L, int Lout, int dtype, int relu, murcl_stream_t stream);
/* backward of the above w.r.t. the pre-tanh activations: dT[b,n,:] = ds_n * wb * (1 - t^2) with
 * ds_n = p_n (dM.H_n / sqrt(N) - dM.M), plus dba += sum dT, dwb += sum ds_n t_n, dbb += sum ds_n
 * (f32, ADDED to the buffers: per-workgroup partial rows in part_ws [512*(2D+1) floats] are summed by a second small
 * launch - 512 atomic adders per address cost a third of the kernel).  A_out (may be NULL) [B,N] receives the normalised
 * attention row softmax(s)/sqrt(N) the pass has in registers.  dH and dWa follow from dT through murcl_panel_gemm / murcl_gemm_nt
 * (MURCL_EPI_RANK1_MASK with rowscale = A, rank1 = dM) and murcl_gemm_tn. */
int murcl_abmil_pool_bwd(const void* H, const void* Wa, const float* ba, const float* wb, const float* scores,
                         const float* ml, const float* M, const float* dM, void* dT, float* dba, float* dwb,
                         float* dbb, float* part_ws, float* A_out, int B, int N, int L, int D, int dtype, int exact_tanh,
                         murcl_stream_t stream);

/* K8/K9 -- NT_Xent.forward + its gradient + torch.cosine_similarity of the positive pairs in one
 * call (one launch for n <= 128, two for the larger global batch of a multi-GPU step; utils/losses.py:24-41;
 * train_MuRCL.py:249,253,277,282).  z [n,P] f32, n = 2B, P = 128.  Row layout: blocks of pair_stride rows alternate
 * between the views - pair_stride = B (or 0) is cat(z_i, z_j) as the reference builds it; pair_stride = bags per rank is
 * an all-gathered [rank][view][bag] buffer used as it arrives; bag ids are global (rank * pair_stride + b).
 * dz (may be NULL) receives d loss / d z for bags in [grad_lo,grad_hi) of both views, zero elsewhere.  sim (may be
 * NULL) [B] by bag id.  workspace: murcl_ntxent_workspace_bytes(n) bytes, contents don't matter. */
long murcl_ntxent_workspace_bytes(int n);
int murcl_ntxent_fwd_bwd(const float* z, int n, int P, float temperature, float* loss, float* dz, float* sim,
                         int grad_lo, int grad_hi, int pair_stride, void* workspace, murcl_stream_t stream);
/* `batches` independent problems of n <= 128 rows (cat(view 0, view 1)) in one launch - the T patch steps of a training
 * step (train_MuRCL.py:249,277): z [batches][n][P] -> loss [batches], dz [batches][n][P] (may be NULL), sim [batches][n/2]. */
int murcl_ntxent_fwd_bwd_batched(const float* z, int batches, int n, int P, float temperature, float* loss, float* dz,
                                 float* sim, murcl_stream_t stream);
/* n <= 128 (one GPU's batch) with ONE cross-workgroup exchange instead of every workgroup recomputing all n x n logits: a workgroup
 * forms the logits of its own 16 rows, publishes their lse as {value, generation} granules and collects the others' by agent-scope
 * polls (losses.py:24-41 unchanged in value: same MFMA operand order, same reductions).  `batches` independent problems per launch;
 * rows / pair_stride / grad window as murcl_ntxent_fwd_bwd.  xchg: murcl_ntxent_xchg_bytes(batches) bytes, zeroed ONCE by the caller
 * and then passed to every call (never to two launches that may overlap in time). */
long murcl_ntxent_xchg_bytes(int batches);
int murcl_ntxent_small_xchg(const float* z, int batches, int n, int P, float temperature, float* loss, float* dz, float* sim,
                            int grad_lo, int grad_hi, int pair_stride, void* xchg, murcl_stream_t stream);

/* K12 -- get_feats (utils/datasets.py:274-308): per bag b and cluster j (ascending id list of length n_j):
 * size_j = rint(float(n_j)*ratio[b]), l_j = floor(actions[b][j]*float(n_j-size_j)), ids cluster_j[l_j : l_j+size_j]
 * (Python slice semantics), all clusters merged, sorted ascending, truncated to feat_size.  ratio[b] must be
 * float32(feat_size / N_b) computed in double on the host, as the reference does.  cluster_ids holds the K id
 * lists of every bag back to back; cluster_off [B][K+1] indexes into it.  idx_out [B][feat_size] (-1 = padding).
 * `views` sub-bags of the SAME B bags in one launch (the T patch steps x 2 views of a step, train_MuRCL.py:237-239,266-269):
 * actions [views][B][K], idx_out [views][B][feat_size], count_out [views][B]. */
int murcl_subbag_select(const int* cluster_ids, const int* cluster_off, const int* n_patches, const float* ratio,
                        const float* actions, int views, int B, int K, int feat_size, int max_patches, int* idx_out,
                        int* count_out, murcl_stream_t stream);
/* K12+K13 -- gather the selected rows (zero padding) and, when lam/perm are given, apply mixup
 * (utils/datasets.py:263-271): out[b] = lam[b]*sub_bag[b] + (1-lam[b])*sub_bag[perm[b]].  feats: all bags' rows
 * back to back ([sum N_b, d]); bag_row_off [B] (int64 row offsets).  `views` as above: idx [views][B][feat_size], lam and
 * perm [views][B] (perm[v][b] = the partner bag's index 0..B-1 inside view v), out [views][B][feat_size][d]. */
int murcl_subbag_gather_mix(const void* feats, const long* bag_row_off, const int* idx, const float* lam,
                            const int* perm, void* out, int views, int B, int feat_size, int d, int dtype_in, int dtype_out,
                            murcl_stream_t stream);
/* The same, writing only the bags [bag_lo, bag_lo + n_out) of every view (out [views][n_out][feat_size][d]) while idx / lam / perm /
 * bag_row_off cover all B bags: the mix-up partner of a rank's bag may be ANY bag of the global batch (utils/datasets.py:267-269
 * permutes over the whole batch) - a data-parallel rank that keeps the whole cohort resident gathers it locally (round 6). */
int murcl_subbag_gather_mix_rows(const void* feats, const long* bag_row_off, const int* idx, const float* lam, const int* perm,
                                 void* out, int views, int B, int feat_size, int d, int bag_lo, int n_out, int dtype_in,
                                 int dtype_out, murcl_stream_t stream);

/* Every random draw of one training step in ONE launch (train_MuRCL.py:235,256-258 window positions ~ U[0,1); utils/datasets.py:265-267
 * mix-up lambda = alpha + U(0,1)(1 - alpha) [n_views, B] and a uniform random permutation of the bags perm [n_views, B] int32 per
 * view; models/rlmil.py:85-86 the sampler's N(0,1) noise).  Counter-based (splitmix64 of seed, stream, index): uni [n_uni] f32,
 * nrm [n_nrm] f32; any of the three parts may be empty.  B <= MURCL_DRAWS_MAX_B. */
#define MURCL_DRAWS_MAX_B 2048
int murcl_step_draws(unsigned long long seed, float* uni, long n_uni, float* nrm, long n_nrm, float* lam, int* perm,
                     int n_views, int B, float alpha, murcl_stream_t stream);
/* K13 -- mixup on an already built batch x [B, per_bag]. */
int murcl_mixup(const void* x, const float* lam, const int* perm, void* out, int B, long per_bag, int dtype,
                murcl_stream_t stream);

/* K6 -- DSMIL aggregator pieces (models/dsmil.py:64-81); the projections themselves are murcl_gemm_nt calls.
 * argmax: m[b,c] = first index of max_n scores[b,n,c] (the reference takes row 0 of a descending sort, :71-73).
 * gather_rows: out[b*C+c,:] = src[b, m[b,c], col0:col0+width].  attn: A[b,n,c] = softmax_n(Q[b,n].qmax[b,c]/sqrt(128))
 * (:76-77).  weighted_rowsum: Z[b,c,:] = sum_n A[b,n,c] X[b,n,:] - with bag = Z Wv^T + bv this equals A^T V (:78)
 * because every column of A sums to one.  rows_dot: out[b,n,c] = X[b,n,:].V[b,c,:].  attn_bwd: soft-max backward,
 * dQ written into dY[:, qcol0:qcol0+128] and dqmax[b,c,:]. */
int murcl_dsmil_argmax(const float* scores, int B, int N, int ld, int C, int* m_out, murcl_stream_t stream);
/* ... and the maxima themselves, max_out [B,C] = scores[b, m[b,c], c]: the max-instance class scores of the DSMIL training body
 * (train_RLMIL.py:516, `torch.max(outputs_ins, 0)`), which the kernel holds once it has found the critical instances. */
int murcl_dsmil_argmax_max(const float* scores, int B, int N, int ld, int C, int* m_out, float* max_out, murcl_stream_t stream);
/* The same attention without the [B*N,128] queries (dsmil.py:64-78 reassociated): Q[n].qmax[c]/sqrt(128) = X[n].v[c] + const with
 * v[c] = Wq^T qmax[c] / sqrt(128), and a soft-max over n ignores the constant - so the scores are a murcl_rows_dot of X against
 * v (a [B*C, d] matrix from two tiny GEMMs), murcl_dsmil_softmax turns them into A in place, and in the backward pass
 * murcl_dsmil_softmax_bwd gives dS = A * (dA - sum_n A dA) (dots_ws: B*C floats), whose weighted row sum over X
 * (murcl_weighted_rowsum) carries everything the query projection's gradient needs:  dWq = (qmax^T R + dqmax^T X[m]) with
 * R[c] = sum_n dS[n,c] X[n] / sqrt(128) and dqmax = R Wq^T.  No GEMM over all patches remains in K6. */
int murcl_dsmil_softmax(float* S, int B, int N, int C, murcl_stream_t stream);
/* The same path with attention AND pooling in one pass over X, and their whole backward in one more (C <= 2, d <= 1024, d % 8 == 0:
 * murcl_dsmil_stream_plan > 0 = the rows a wave takes; 0: use the separate launches above).
 * attn_pool: A [B,N,C] receives the logits X.v and then the soft-max over n (online soft-max per wave, merged per bag); Z [B,C,d]
 * = A^T X.  ws: (B*N/plan)*C*(d+2) + 2*B*C floats.
 * attn_pool_bwd: R [B,C,d] = scale * sum_n A[n,c] (dA[n,c] - sum_m A[m,c] dA[m,c]) X[n] with dA = X dZ^T, taken as
 * (sum_n A dA X[n]) - (sum_n A dA) Z[c] - Z = the forward's pooled rows - so that neither dA nor dS is stored; with dcls [B,N,C]
 * (may be NULL) the pass also leaves the per-wave partial rows of dWc = dcls^T X in gpart [(B*N/plan)][C*d] (sum them with
 * murcl_colsum).  ws: (B*N/plan)*C*(d+2) floats. */
/* The [B*C]-row algebra around those passes, d % 4 == 0, d <= 2048.  qv: x_m = X[b, m[b,c]] (f32 copy xm [B*C,d]), qmax = Wq x_m + bq
 * [B*C,128] (dsmil.py:74-75), v = Wq^T qmax [B*C,d].  qv_bwd: with R [B*C,d] = the gradient of v (murcl_dsmil_attn_pool_bwd):
 * dq = Wq R (dq_ws [B*C,128]), dWq [128,d] = qmax^T R + dq^T xm (OVERWRITTEN), dbq [128] = sum_r dq (overwritten). */
int murcl_dsmil_qv(const void* X, const int* m, const float* Wq, const float* bq, int B, int N, int d, int C, float* xm, float* qmax,
                   float* v, int dtype, murcl_stream_t stream);
int murcl_dsmil_qv_bwd(const float* R, const float* qmax, const float* xm, const float* Wq, int BC, int d, float* dq_ws, float* dWq,
                       float* dbq, murcl_stream_t stream);
/* qv_bwd with the gradient dcmax [B*C] of the max-instance class scores (murcl_dsmil_argmax_max; train_RLMIL.py:527-529): the
 * instance classifier's dWc [C,d] (+)= sum_b dcmax[b,c] xm[b,c] and dbc [C] (+)= sum_b dcmax[b,c] come out of the same second launch -
 * no dense [B,N,C] gradient of the instance scores is formed for that term. */
int murcl_dsmil_qv_bwd_cls(const float* R, const float* qmax, const float* xm, const float* Wq, int BC, int d, float* dq_ws, float* dWq,
                           float* dbq, const float* dcmax, int C, float* dWc, float* dbc, int accumulate /* 0: dWc, dbc overwritten */,
                           murcl_stream_t stream);
int murcl_dsmil_stream_plan(int B, int N, int d, int C);
int murcl_dsmil_attn_pool(const void* X, const float* v, float vscale /* logits = vscale * X.v */, float* A, float* Z, float* ws,
                          int B, int N, int d, int C, int dtype, murcl_stream_t stream);
int murcl_dsmil_attn_pool_bwd(const void* X, const float* dZ, const float* A, const float* Z, const float* dcls, float scale,
                              float* R, float* gpart, float* ws, int B, int N, int d, int C, int dtype, murcl_stream_t stream);
int murcl_dsmil_softmax_bwd(const float* A, const float* dA, int B, int N, int C, float* dS, float* dots_ws,
                            murcl_stream_t stream);
int murcl_gather_rows(const void* src, const int* m, int B, int C, int N, int ld, int col0, int width, void* out,
                      int dtype, murcl_stream_t stream);
int murcl_dsmil_attn(const float* Q, int ldq, int qcol0, const float* qmax, int B, int N, int C, float* A,
                     murcl_stream_t stream);
int murcl_weighted_rowsum(const void* X, const float* A, float* Z, int B, int N, int d, int C, int dtype,
                          murcl_stream_t stream);
/* The same ADDED into Z (no zero fill in front: the caller cleared Z, e.g. through murcl_softmax_rows_parts). */
int murcl_weighted_rowsum_acc(const void* X, const float* A, float* Z, int B, int N, int d, int C, int dtype,
                              murcl_stream_t stream);
int murcl_rows_dot(const void* X, const float* V, float* out, int B, int N, int d, int C, int dtype,
                   murcl_stream_t stream);
/* the same + bias[c] (may be NULL): the instance classifier's Linear(d, C) (dsmil.py:9,15) in one launch */
int murcl_rows_dot_bias(const void* X, const float* V, const float* bias, float* out, int B, int N, int d, int C, int dtype,
                        murcl_stream_t stream);
/* rows_dot and the weighted row sum sum_n G[b,n,c] X[b,n,:] over ONE pass of X (DSMIL backward: dA = X dZ^T and
 * dWc = dcls^T X both need every row of X).  murcl_rows_dot_wsum_plan returns the rows a wave takes (0: shape not covered:
 * C <= 2, d <= 1024, d % 8 == 0); part [B*N / that][C][d] receives per-wave partial sums of the weighted rows, which the
 * caller adds up (murcl_colsum). */
int murcl_rows_dot_wsum_plan(int B, int N, int d, int C);
int murcl_rows_dot_wsum(const void* X, const float* V, const float* G, float* out, float* part, int B, int N, int d, int C,
                        int dtype, murcl_stream_t stream);
int murcl_dsmil_attn_bwd(const float* A, const float* dA, const float* Q, int ldq, int qcol0, const float* qmax, int B,
                         int N, int C, float* dY, int ldy, float* dqmax, float* dots_ws /* [B*C] */, murcl_stream_t stream);

/* K4/K5 -- CLAM-SB pieces (models/clam.py); the fc and gate projections are murcl_gemm_nt / murcl_panel_gemm calls.
 * gated_score: s[n] = sum_d tanh(U[n,d])*sigmoid(U[n,D+d])*wc[d] + bc (clam.py:56-59), optional dropout keep
 * multipliers (0 or 1/0.75) for the two branches (clam.py:47-48); _bwd gives dU, dwc, dbc.  softmax_rows: A =
 * softmax_N(s) per bag (clam.py:144).  topk_ids: ids[b,0:k] top-k of A, ids[b,k:2k] bottom-k (clam.py:107,109,126;
 * lowest index wins ties).  take_rows / scatter_add_rows_masked: gather instance features and add their gradient back
 * under the ReLU mask.  cross_entropy: mean CE over R rows + gradient + arg-max predictions (clam.py:116-118). */
/* gated != 0: U [rows, 2D] = (tanh branch | sigmoid branch) of Attn_Net_Gated (clam.py:37-60); gated == 0: U [rows, D], the
 * plain Attn_Net (clam.py:18-34): s = tanh(U) . wc + bc (keep_b unused; dbab then holds D column sums followed by D zeros) */
/* keep_a / keep_b NULL and 0 < keep_p < 1: the same Dropout with masks that are never materialised - the counter-based masks
 * murcl_dropout_mask would produce for seed_a / seed_b over [rows, D] (scale 1/keep_p), regenerated identically by _bwd. */
int murcl_gated_score_fwd(const void* U, const float* wc, const float* bc, const void* keep_a, const void* keep_b,
                          float* s, long rows, int D, int dtype, int gated, float keep_p, unsigned long long seed_a,
                          unsigned long long seed_b, murcl_stream_t stream);
/* dbab (may be NULL) [2D] receives the column sums of dU - the bias gradients of the two gate Linears - from the same pass */
int murcl_gated_score_bwd(const void* U, const float* wc, const void* keep_a, const void* keep_b, const float* ds,
                          void* dU, float* dwc, float* dbc, float* dbab, float* part_ws /* [1024*(3D+1)] */, long rows, int D,
                          int dtype, int gated, float keep_p, unsigned long long seed_a, unsigned long long seed_b,
                          murcl_stream_t stream);
/* The CLAM-SB training chain's form (gated, U / dU [rows,2D] in the interleaved column order of murcl_panel_gemm_drop's epilogue
 * 5; dwc, dbab in natural order; seeded masks only).  ds given - or ds NULL and the pooling + soft-max backward
 * (clam.py:144,170) taken in the same pass: ds_n = A_n (h_n . dM_bag - Mp_bag . dM_bag) from h [rows,L] (dtype of U, bf16),
 * the pooled vectors Mp [rows/rows_per_bag, L], their upstream gradient dM (same shape) and the attention A [rows], all f32;
 * L = 8 c D/8 with c in {1,2,4}, D/8 a power of two <= 64. */
int murcl_gated_score_bwd_il(const void* U, const float* wc, const float* ds, void* dU, float* dwc, float* dbc, float* dbab,
                             float* part_ws /* [1024*(3D+1)] */, long rows, int D, int dtype, float keep_p,
                             unsigned long long seed_a, unsigned long long seed_b, const void* h, const float* dM,
                             const float* Mp, const float* A, int L, int rows_per_bag, murcl_stream_t stream);
int murcl_softmax_rows(const float* s, float* A, int B, int N, murcl_stream_t stream);
/* The same from P partial score rows part [P][B*N] (the gate epilogues 4 / 5 of murcl_panel_gemm): s [B,N] = their column sum (written
 * too: clam.py:144 returns the raw scores), A = soft-max_N(s) - one launch instead of murcl_colsum + murcl_softmax_rows.
 * `zero` (may be NULL): [B, zero_n] f32 cleared by the same launch - the pooled rows murcl_weighted_rowsum_acc then adds into. */
int murcl_softmax_rows_parts(const float* part, int P, float* s, float* A, int B, int N, float* zero, int zero_n,
                             murcl_stream_t stream);
int murcl_softmax_rows_bwd(const float* A, const float* dA, float* ds, int B, int N, murcl_stream_t stream);
int murcl_topk_ids(const float* A, int B, int N, int k, int* ids, murcl_stream_t stream);
int murcl_take_rows(const void* src, const long* rows, float* out, int R, int d, int dtype, murcl_stream_t stream);
/* CLAM's instance branch (clam.py:103-132,150-168) for ALL (bag, class) pairs as one launch each way.  h [B*N,L] (dtype), ids
 * [B,2k] from murcl_topk_ids, labels [B], W [n_cls*2,L] / bias [n_cls*2] = the stacked instance classifiers (2 n_cls <= 16, k <= 32).
 * fwd: loss[b] = scale * sum_c CE_c (class == label: targets [1]*k + [0]*k over the 2k rows, inst_eval; other classes: [0]*k over
 * the top-k rows if subtyping, inst_eval_out, else nothing), dl [B*2k, 2 n_cls] = d loss[b] / d logits, pt [2][B][n_cls][2k] =
 * (predictions, targets), -1 where a pair has no such row.
 * bwd: with up[b] = dL/dloss[b]: dz[row,:] += (up dl[r,:] W) where h[row,:] > 0 for the bag's 2k rows (the first layer's ReLU
 * mask), and part[b] = (up dl^T h_rows [2 n_cls][L] | up sum_r dl[r,:] [2 n_cls] | the column sums of what was added [L]):
 * murcl_colsum over the B rows gives the classifiers' weight / bias gradients and the extension of the first layer's bias gradient. */
int murcl_clam_inst_fwd(const void* h, const int* ids, const long* labels, const float* W, const float* bias, int B, int N, int L,
                        int k, int n_cls, int subtyping, float scale, float* loss, float* dl, long* pt, int dtype,
                        murcl_stream_t stream);
int murcl_clam_inst_bwd(const void* h, const int* ids, const float* W, const float* dl, const float* up, int B, int N, int L, int k,
                        int n_cls, void* dz, float* part, int dtype, murcl_stream_t stream);
/* write_back != 0: g[r,k] is zeroed where the mask dropped it, so that on return g holds exactly what was added */
int murcl_scatter_add_rows_masked(void* dst, const void* h, const long* rows, float* g, int R, int d, int dtype,
                                  int write_back, murcl_stream_t stream);
/* nn.CrossEntropyLoss() (mean) per group of `group` consecutive rows of [R,C] logits (train_RLMIL.py:316,502,709; clam.py:118,131):
 * loss [R/group], dlogits [R,C] = (soft-max - one-hot) / live rows of the group (NULL: not wanted), preds [R] = arg-max (NULL: not
 * wanted), conf [R] = soft-max probability of the target class - the confidence whose differences are the RL-MIL rewards
 * (train_RLMIL.py:345,369-371) - NULL: not wanted.  Rows with target < 0 are ignored. */
int murcl_cross_entropy(const float* logits, const long* targets, int R, int C, float* loss, float* dlogits,
                        long* preds, float* conf, int group, murcl_stream_t stream);
/* nn.Dropout keep mask (clam.py:47-48,71-72) in the compute dtype: out[i] = scale with probability keep_p (quantised to
 * 1/256), else 0; a pure function of (seed, i) (splitmix64 counter hash), one write pass. */
int murcl_dropout_mask(void* out, long n, float keep_p, float scale, unsigned long long seed, int dtype,
                       murcl_stream_t stream);
/* loss[R/group]: mean per `group` rows */
int murcl_mul(const void* x, const void* k, void* y, long n, int dtype, murcl_stream_t stream);

/* K10/K11 -- PPO head math (models/rlmil.py:66-127,152-184); MLP, GRU and heads are the GEMM / GRU-gate entries.
 * policy_head_fwd: mu = sigmoid(z); with eps: action = clamp(mu + std*eps, 0, 1) (act), else evaluates act_in;
 * logp of the diagonal Gaussian whose scale_tril is diag(std).  ppo_returns: discounted + normalised returns.
 * ppo_loss: clipped surrogate + 0.5 MSE - 0.01 entropy, mean over n_total rows (n local rows; n_total = n on one rank),
 * with d/dlogp and d/dvalue.  ppo_returns_raw / ppo_returns_finish: the data-parallel split of ppo_returns (rlmil.py:162
 * normalises over the whole batch): raw discounted returns + local (sum, sum of squares) as two doubles; after the
 * caller's all-reduce of that pair, finish normalises with the global mean / unbiased std over n_total returns. */
int murcl_policy_head_fwd(const float* z, const float* eps, const float* act_in, float std_, int R, int K, float* mu,
                          float* act_out, float* logp, murcl_stream_t stream);
int murcl_policy_head_bwd(const float* mu, const float* act, const float* dlogp, float std_, int R, int K, float* dz,
                          murcl_stream_t stream);
int murcl_ppo_returns(const float* rewards, float gamma, int T, int B, float* ret, murcl_stream_t stream);
int murcl_ppo_returns_raw(const float* rewards, float gamma, int T, int B, float* ret, double* stats, murcl_stream_t stream);
int murcl_ppo_returns_finish(float* ret, int n, const double* stats, long n_total, murcl_stream_t stream);
int murcl_ppo_loss(const float* logp, const float* old_logp, const float* value, const float* ret, float eps_clip,
                   float entropy, int n, long n_total, float* loss, float* dlogp, float* dvalue, murcl_stream_t stream);

/* helpers */
int murcl_cast(const void* x, void* y, long n, int dtype_in, int dtype_out, murcl_stream_t stream);
int murcl_transpose_cast(const float* x, void* y, int R, int C, int dtype_out, murcl_stream_t stream);
int murcl_colsum(const void* x, float* out, int R, int N, int ld, int dtype, int accumulate, murcl_stream_t stream);
int murcl_relu_bwd(const float* dy, const float* y, float* dx, long n, murcl_stream_t stream);

/* K7/K10 -- one nn.GRU time step's gate math (rlmil.py:47,78,199,213-217); the two projections are
 * murcl_gemm_nt calls.  gates [B,3H] = (r,z,n) saved for backward.  gh_bcast != 0: gh is a single row [3H]
 * shared by all B rows (zero initial state: gh = b_hh, no [B,3H] copy of the bias). */
int murcl_gru_gates_fwd(const float* gi, const float* gh, const float* hprev, float* hnew, float* gates, int B, int H,
                        int gh_bcast, murcl_stream_t stream);
int murcl_gru_gates_bwd(const float* dh, const float* gates, const float* gh, const float* hprev, float* dgi,
                        float* dgh, float* dhprev, int B, int H, int gh_bcast, murcl_stream_t stream);
/* the same; accumulate != 0: dhprev += dh * z (back-propagation through time, where dhprev already holds that step's own
 * upstream gradient) */
int murcl_gru_gates_bwd_into(const float* dh, const float* gates, const float* gh, const float* hprev, float* dgi,
                             float* dgh, float* dhprev, int B, int H, int gh_bcast, int accumulate, murcl_stream_t stream);

/* One nn.GRU time step as ONE launch each way (rlmil.py:47,78 ActorCritic.gru; :14-35 Full_layer.rnn): a workgroup owns a
 * 16-row x 16-unit tile of all three gate blocks, forms its slice of h W_hh^T (forward) or dgh W_hh (backward) on the exact-f32
 * matrix pipe and finishes the gate math in the epilogue - replaces murcl_gemm_nt + murcl_gru_gates_* pairs (same arithmetic;
 * only the summation order inside the products differs).  H % 16 == 0; murcl_gru_step_supported(B, H, Kx) != 0 otherwise.
 * murcl_gru_step_fwd: gi [B,3H] = x W_ih^T + b_ih, or - with x [B,Kx] and w_ih [3H,Kx] given (Kx % 16 == 0) - gi is the bias
 *   row b_ih [3H] and the input product is formed here too.  hprev [B,H]; NULL (with x given) = the zero state of a restart:
 *   only the input product is formed, gh = b_hh.  Writes hnew [B,H], gates [B,3H] = (r,z,n) and gh [B,3H] = h W_hh^T + b_hh (either may
 *   be NULL when no backward follows).
 * murcl_gru_step_bwd: dh [B,H] holds the step's upstream gradient; adds dgh_next [B,3H] . W_hh (w_hh_t = W_hh^T [H,3H]), writes
 *   the total back, then the gate backward exactly as murcl_gru_gates_bwd_into (dgi, dgh, dhprev (+)= dh * z; dhprev may be NULL). */
int murcl_gru_step_supported(int B, int H, int Kx);
int murcl_gru_step_fwd(const float* x, const float* w_ih, int Kx, const float* gi, const float* hprev, const float* w_hh,
                       const float* b_hh, float* hnew, float* gates, float* gh, int B, int H, murcl_stream_t stream);
int murcl_gru_step_bwd(const float* dgh_next, const float* w_hh_t, float* dh, const float* gates, const float* gh,
                       const float* hprev, float* dgi, float* dgh, float* dhprev, int B, int H, int gh_bcast, int accumulate,
                       murcl_stream_t stream);

/* The PPO sampler as native launch sequences (models/rlmil.py:66-97 act, :99-127,169-181 evaluate + loss + backward):
 * one call enqueues the whole chain of GEMM / GRU / head launches, so the host pays one call instead of one per launch
 * (the rollouts are a few hundred rows: launch-bound).  params / grads: HOST arrays of 12 device pointers in
 * ActorCritic.parameters() order (state_encoder.0.{weight,bias}, state_encoder.2.{weight,bias},
 * gru.{weight_ih,weight_hh,bias_ih,bias_hh}_l0, actor.0.{weight,bias}, critic.0.{weight,bias}); S = state_dim, H = hidden
 * (both multiples of 32), K = action_size <= 16.  Workspaces: *_workspace() bytes of f32.
 * murcl_ppo_act: hidden_prev NULL = zeros (restart_batch); eps ~ N(0,1) [B,K]; writes hidden_new [B,H], action [B,K]
 * (clamped to [0,1]), logp [B].
 * murcl_ppo_epoch: states [T,B,S], actions [T,B,K], old_logp / returns [T,B]; parameter gradients are ADDED to grads[];
 * the loss is a mean over n_total >= T*B rows (all ranks' rows: gradients carry 1/n_total); loss_out (may be NULL) [1]. */
long murcl_ppo_act_workspace(int B, int S, int H);
int murcl_ppo_act(const float* const* params, int S, int H, int K, const float* state, const float* hidden_prev,
                  const float* eps, float std_, int B, float* hidden_new, float* action, float* logp, float* ws,
                  murcl_stream_t stream);
/* murcl_ppo_epoch_wt: the same with wt = HOST array {W_ih^T [H,3H], W_hh^T [H,3H], W_2^T [2048,H]} of device pointers prepared by
 * the caller (one murcl_cast_batch launch per optimizer step) instead of three transposes inside every call. */
long murcl_ppo_epoch_workspace(int T, int B, int S, int H);
int murcl_ppo_epoch(const float* const* params, float* const* grads, int S, int H, int K, const float* states,
                    const float* actions, const float* old_logp, const float* returns, int T, int B, long n_total,
                    float std_, float eps_clip, float entropy, float* ws, float* loss_out, murcl_stream_t stream);
int murcl_ppo_epoch_wt(const float* const* params, float* const* grads, const float* const* wt, int S, int H, int K,
                       const float* states, const float* actions, const float* old_logp, const float* returns, int T, int B,
                       long n_total, float std_, float eps_clip, float entropy, float* ws, float* loss_out, murcl_stream_t stream);

/* 1-bit ReLU' mask (x > 0) of an activation tensor x [M,N] in murcl_panel_gemm's bit-mask layout (M*N/8 bytes;
 * M % 32 == 0, N % 32 == 0), for layers whose forward did not emit it (clam.py:69 with a 1024-wide input). */
int murcl_relu_bitmask(const void* x, void* bits, int M, int N, int ld, int dtype, murcl_stream_t stream);
/* nn.Dropout applied in place to x [M,N] (contiguous; M % 32 == 0, N % 128 == 0) with the counter-based keep mask of
 * murcl_dropout_mask for the same seed - the mask is never materialised - and, in the same pass, the 1-bit mask of the
 * surviving positive entries in the layout above (bits may be NULL).  CLAM's Dropout(0.25) after the first layer's ReLU
 * (models/clam.py:69-72). */
int murcl_dropout_relu_bitmask(void* x, void* bits, int M, int N, float keep_p, float scale, unsigned long long seed, int dtype,
                               murcl_stream_t stream);

/* Compute-dtype copies / transposes of several f32 weight matrices in one launch.  jobs_dev: n_jobs records of
 * { const float* src; void* dst; int rows, cols, transpose, dtype_out; } (32 bytes each) in device memory; max_tiles =
 * the largest ceil(rows/32)*ceil(cols/32) among them.  Replaces the per-tensor `.to(dtype)` / `.t()` copies that the
 * reference's nn.Linear calls imply (abmil.py:12-21).  `transpose`: bit 0 = transpose; bits 8.. = the leading dimension of dst in
 * elements (0: the job's own width), so that a job can fill a row or column block of a larger matrix - CLAM's two gate Linears
 * interleaved in 16-row blocks for murcl_panel_gemm epilogues 4 / 5 (clam.py:40-48) are 2 x D/16 such jobs. */
int murcl_cast_batch(const void* jobs_dev, int n_jobs, int max_tiles, murcl_stream_t stream);
/* The same table as a flat grid: first_tile_dev [n_jobs + 1] int32 (device) = ascending first tile index of every job (a job of
 * [rows, cols] has ceil(rows/32) * ceil(cols/32) tiles), total_tiles = its last entry.  For tables whose jobs differ widely in size. */
int murcl_cast_batch_flat(const void* jobs_dev, const int* first_tile_dev, int n_jobs, int total_tiles, murcl_stream_t stream);
/* ... which also advances tick_dev[0] by one (NULL: no counter): the step counter of a captured optimizer step
 * (murcl_adam_multi_live_deferred).  -1 when there is nothing to launch and a counter was given. */
int murcl_cast_batch_flat_tick(const void* jobs_dev, const int* first_tile_dev, int n_jobs, int total_tiles, int* tick_dev,
                               murcl_stream_t stream);

/* torch.stack of several lists of equally shaped tensors in ONE launch (PPO.update, rlmil.py:163-165: the rollout's states, actions
 * and log-probabilities): job i copies `bytes` (a multiple of 4) from src to dst.  `jobs_host` is a HOST array. */
#define MURCL_STACK_MAX_JOBS 96
typedef struct { const void* src; void* dst; long bytes; } MurclCopyJob;
int murcl_stack_lists(const MurclCopyJob* jobs_host, int n_jobs, murcl_stream_t stream);
/* dst += src (f32, `bytes` a multiple of 4) for every job, ONE launch: the gradients a backward node returns as its own tensors added
 * into the parameters' gradient buffers - autograd's AccumulateGrad (`loss.backward()`, train_MuRCL.py:293; one add per parameter). */
int murcl_add_lists(const MurclCopyJob* jobs_host, int n_jobs, murcl_stream_t stream);

/* torch.optim.Adam.step for one flat tensor (train_MuRCL.py:165,295; rlmil.py:141,182).  zero_grad != 0 also clears g
 * (the optimizer.zero_grad() that precedes the next backward pass, train_MuRCL.py:293) in the same pass. */
int murcl_adam_step(float* p, float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, int zero_grad, murcl_stream_t stream);
/* The same step for up to MURCL_ADAM_MAX_JOBS flat runs in ONE launch (the two parameter groups of train_MuRCL.py:165-171 with their
 * own lr; runs of parameters that share a step count).  `jobs_host` is a HOST array (it travels in the kernel arguments). */
#define MURCL_ADAM_MAX_JOBS 8
typedef struct { float *p, *g, *m, *v; long n; float lr; int step; } MurclAdamJob;
int murcl_adam_multi(const MurclAdamJob* jobs_host, int n_jobs, float beta1, float beta2, float eps, float weight_decay,
                     int zero_grad, murcl_stream_t stream);
/* The same launch for a step that is CAPTURED into a hipGraph and replayed: replays_dev (device int[2], zeroed by the caller before
 * the first replay; NULL = murcl_adam_multi) counts the steps the replays have taken - the kernel's bias corrections use
 * job.step + replays_dev[0] and a one-thread launch queued behind it advances the counter, so that every replay is the NEXT optimizer step, not the
 * captured one again (torch.optim.Adam's state['step'], train_MuRCL.py:165-171, kept on the device). */
int murcl_adam_multi_live(const MurclAdamJob* jobs_host, int n_jobs, float beta1, float beta2, float eps, float weight_decay,
                          int zero_grad, int* replays_dev, murcl_stream_t stream);
/* ... without the one-thread launch that advances the counter: the caller's next launch on the stream does it
 * (murcl_cast_batch_flat_tick - the weight-view refresh that follows every optimizer step - or murcl_replay_tick). */
int murcl_adam_multi_live_deferred(const MurclAdamJob* jobs_host, int n_jobs, float beta1, float beta2, float eps, float weight_decay,
                                   int zero_grad, int* replays_dev, murcl_stream_t stream);
int murcl_replay_tick(int* replays_dev, murcl_stream_t stream);
/* out = a x + b y over n floats: the rewards of a contrastive step, cosine of patch step t-1 minus that of step t
 * (train_MuRCL.py:282-283), for all T-1 steps in one launch; b == 0: out = a x, y is not read (the 1/T of the step loss's mean on
 * the stored NT-Xent gradients, :291). */
int murcl_axpby(const float* x, const float* y, float a, float b, float* out, long n, murcl_stream_t stream);
/* out[0] = mean of n floats in a fixed order (one workgroup): the step loss, mean of the T patch-step losses (train_MuRCL.py:291),
 * where nothing differentiates it (frozen-aggregator stage 2). */
int murcl_mean_small(const float* x, int n, float* out, murcl_stream_t stream);
/* out[g] = mean of x[g * group .. (g + 1) * group), g < groups: CLAM-SB's instance loss per bag -> per patch step (train_RLMIL.py:336,
 * `instance_loss` averaged over the batch of every step) when the T x B bags of a step were computed together. */
int murcl_group_mean(const float* x, int groups, int group, float* out, murcl_stream_t stream);
/* dst <- src (both 16-byte aligned, any byte count): `policy_old.load_state_dict(policy.state_dict())` after a PPO update
 * (models/rlmil.py:183) on the two flat parameter buffers, as a launch of this library.  -1: a pointer is not 16-byte aligned. */
int murcl_copy_bytes(const void* src, void* dst, long bytes, murcl_stream_t stream);
/* C[M,N] (+)= A[M,K] B[N,K]^T, f32, K <= 16: the input gradient dX = dY W of a classifier head with a handful of outputs
 * (train_RLMIL.py:316,502,709: nn.Linear(hidden, num_classes)) without zero-padding k to the matrix-core kernels' step.  -1: K > 16. */
int murcl_gemm_nt_smallk(const float* A, const float* B, float* C, int M, int N, int K, int accumulate, murcl_stream_t stream);
/* dst[R,Cp] = [src[R,C] | 0] (elem_size 2 or 4): the zero-padded operand of a product whose extent is not a multiple of the kernels'
 * step, one launch instead of a fill and a strided copy. */
int murcl_pad_cols(const void* src, void* dst, long R, int C, int Cp, int elem_size, murcl_stream_t stream);
/* torch.optim.SGD.step for one flat tensor (train_MuRCL.py:158-163, train_RLMIL.py:258-263): L2 weight decay, momentum
 * buffer (first != 0: the buffer is initialised with the gradient), dampening 0, optional Nesterov. */
int murcl_sgd_step(float* p, float* g, float* buf, long n, float lr, float momentum, int nesterov, float weight_decay,
                   int first, int zero_grad, murcl_stream_t stream);

/* K-means (Lloyd) over the patch features of one slide - the clustering pre-step behind the cluster id lists
 * (wsi_processing/features_clustering.py:10-16; sklearn KMeans).  One call = one iteration: labels[i] = argmin_k
 * |x_i - c_k|^2 against `centers` [K,d] (first minimum on ties), then, if `update`, centers <- means of their rows (a
 * cluster without rows keeps its centre).  X [N,d] f32, d in {256, 512, 1024}, K <= 16.  labels [N] int32 in/out (the
 * previous labels are compared), counts [K], stats [3+K] = {sum of squared centre shifts, inertia of this assignment,
 * rows whose label changed, rows per cluster...}, mind2 (may be NULL) [N] squared distance of every row to its centre
 * (what scikit-learn's empty-cluster relocation ranks by).  Deterministic: no float atomics.
 * workspace: murcl_kmeans_workspace_bytes(N, d, K). */
long murcl_kmeans_workspace_bytes(int N, int d, int K);
int murcl_kmeans_step(const float* X, int N, int d, int K, float* centers, int* labels, int* counts, float* stats,
                      float* mind2, int update, void* workspace, murcl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
