"""autograd.Functions that stitch the HIP kernels into differentiable ops.

Forward and backward are sequences of C-ABI launches (murcl_amd.ops); torch supplies
only tensor storage and the autograd graph.  No CPU / eager-PyTorch fallback exists:
CPU tensors raise.
"""
import torch

from . import ops


def _flat2(x):
    return x.reshape(-1, x.shape[-1])


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) for small (bag-level) f32 matrices; act in {none, relu}."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        x2 = _flat2(x).contiguous()
        y = ops.gemm_nt(x2, w, epi=ops.EPI_BIAS_RELU if relu else ops.EPI_BIAS, bias=b)
        ctx.save_for_backward(x2, w, y if relu else None)
        ctx.relu = relu
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        dy2 = _flat2(dy).contiguous()
        if ctx.relu:
            dy2 = ops.relu_bwd(dy2, y)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(dy2, ops.transpose_cast(w, torch.float32)).view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            dw = ops.gemm_tn(dy2, x2)
        if ctx.needs_input_grad[2]:
            db = ops.colsum(dy2)
        return dx, dw, db, None


class ABMILFn(torch.autograd.Function):
    """Whole ABMIL.bag_forward for a batch of equal-length bags (models/abmil.py:35-45).

    x [B,N,d] in the compute dtype (f32 parity path / bf16 throughput path); parameters f32.
    Patch-level tensors (H1..H3, dZ*, dT) live in the compute dtype with f32 accumulation;
    bag-level tensors are f32.  Returns (out [B,L], A [B,N]) - A is non-differentiable.
    """

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, wa, ba, wb, bb, wd, bd):
        B, N, d = x.shape
        T = x.dtype
        x2 = x.reshape(B * N, d)
        c = (lambda w: w) if T == torch.float32 else (lambda w: ops.cast(w, T))
        L = w3.shape[0]
        # bf16 + panel-friendly shapes: weight-stationary GEMMs that also emit 1-bit ReLU masks
        fast = (T == torch.bfloat16 and d == 512 and ops.panel_supported(B * N, L, 512, ops.PG_BIAS_RELU)
                and ops.panel_supported(B * N, L, 128, ops.PG_RANK1_MASK, N))
        if fast:
            h1, m1, _ = ops.panel_gemm(x2, c(w1), ops.PG_BIAS_RELU, bias=b1, want_bitmask=True)
            h2, m2, _ = ops.panel_gemm(h1, c(w2), ops.PG_BIAS_RELU, bias=b2, want_bitmask=True)
            h3, m3, _ = ops.panel_gemm(h2, c(w3), ops.PG_BIAS_RELU, bias=b3, want_bitmask=True)
        else:
            m1 = m2 = m3 = None
            h1 = ops.gemm_nt(x2, c(w1), epi=ops.EPI_BIAS_RELU, bias=b1)
            h2 = ops.gemm_nt(h1, c(w2), epi=ops.EPI_BIAS_RELU, bias=b2)
            h3 = ops.gemm_nt(h2, c(w3), epi=ops.EPI_BIAS_RELU, bias=b3)
        wac = c(wa)
        scores, A, M, ml = ops.abmil_pool_fwd(h3.view(B, N, L), wac, ba, wb, bb)
        out = ops.gemm_nt(M, wd, epi=ops.EPI_BIAS_RELU, bias=bd)
        ctx.save_for_backward(x2, h1, h2, h3, scores, A, M, ml, out, w1, w2, w3, wa, ba, wb, wd, wac, m1, m2, m3)
        ctx.dims = (B, N, d)
        ctx.mark_non_differentiable(A)
        return out, A

    @staticmethod
    def backward(ctx, dout, _dA):
        x2, h1, h2, h3, scores, A, M, ml, out, w1, w2, w3, wa, ba, wb, wd, wac, m1, m2, m3 = ctx.saved_tensors
        B, N, d = ctx.dims
        T = x2.dtype
        L = h3.shape[1]
        # decoder (bag level, f32)
        dpre = ops.relu_bwd(dout.contiguous(), out)
        dwd = ops.gemm_tn(dpre, M)
        dbd = ops.colsum(dpre)
        dM = ops.gemm_nt(dpre, ops.transpose_cast(wd, torch.float32))
        # attention pooling
        dT, dba, dwb, dbb = ops.abmil_pool_bwd(h3.view(B, N, L), wac, ba, wb, scores, ml, M, dM)
        dwa = ops.gemm_tn(dT, h3)
        # encoder layer 3: dZ3 = (dT Wa + A (x) dM) * relu'(H3)
        if m3 is not None:
            dz3, _, db3 = ops.panel_gemm(dT, ops.transpose_cast(wa, T), ops.PG_RANK1_MASK, bitmask=m3,
                                         rowscale=A.view(-1), rank1=dM, rows_per_bag=N, colsum=True)
            dw3 = ops.gemm_tn(dz3, h2)
            dz2, _, db2 = ops.panel_gemm(dz3, ops.transpose_cast(w3, T), ops.PG_MASK, bitmask=m2, colsum=True)
            dw2 = ops.gemm_tn(dz2, h1)
            dz1, _, db1 = ops.panel_gemm(dz2, ops.transpose_cast(w2, T), ops.PG_MASK, bitmask=m1, colsum=True)
        else:
            dz3, ws = ops.gemm_nt(dT, ops.transpose_cast(wa, T), epi=ops.EPI_RANK1_MASK, mask=h3, rowscale=A.view(-1),
                                  rank1=dM, rows_per_bag=N, colsum=True)
            db3 = ops.colsum(ws)
            dw3 = ops.gemm_tn(dz3, h2)
            dz2, ws = ops.gemm_nt(dz3, ops.transpose_cast(w3, T), epi=ops.EPI_MASK, mask=h2, colsum=True)
            db2 = ops.colsum(ws)
            dw2 = ops.gemm_tn(dz2, h1)
            dz1, ws = ops.gemm_nt(dz2, ops.transpose_cast(w2, T), epi=ops.EPI_MASK, mask=h1, colsum=True)
            db1 = ops.colsum(ws)
        dw1 = ops.gemm_tn(dz1, x2)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(dz1, ops.transpose_cast(w1, T)).view(B, N, d)
        return dx, dw1, db1, dw2, db2, dw3, db3, dwa, dba, dwb.view(1, -1), dbb, dwd, dbd


class GRUStepFn(torch.autograd.Function):
    """One nn.GRU time step (seq_len 1, PyTorch gate order); h_prev None == zeros."""

    @staticmethod
    def forward(ctx, x, h_prev, w_ih, w_hh, b_ih, b_hh):
        x = x.contiguous()
        gi = ops.gemm_nt(x, w_ih, epi=ops.EPI_BIAS, bias=b_ih)
        if h_prev is None:
            gh = b_hh.unsqueeze(0).expand(x.shape[0], -1).contiguous()      # W_hh . 0 + b_hh
        else:
            h_prev = h_prev.contiguous()
            gh = ops.gemm_nt(h_prev, w_hh, epi=ops.EPI_BIAS, bias=b_hh)
        h_new, gates = ops.gru_gates_fwd(gi, gh, h_prev)
        ctx.save_for_backward(x, h_prev, w_ih, w_hh, gates, gh)
        return h_new

    @staticmethod
    def backward(ctx, dh):
        x, h_prev, w_ih, w_hh, gates, gh = ctx.saved_tensors
        dgi, dgh, dhp = ops.gru_gates_bwd(dh.contiguous(), gates, gh, h_prev)
        dx = ops.gemm_nt(dgi, ops.transpose_cast(w_ih, torch.float32)) if ctx.needs_input_grad[0] else None
        dw_ih = ops.gemm_tn(dgi, x)
        db_ih = ops.colsum(dgi)
        db_hh = ops.colsum(dgh)
        if h_prev is None:
            dh_prev, dw_hh = None, torch.zeros_like(w_hh)
        else:
            dw_hh = ops.gemm_tn(dgh, h_prev)
            dh_prev = None
            if ctx.needs_input_grad[1]:
                dh_prev = ops.gemm_nt(dgh, ops.transpose_cast(w_hh, torch.float32), out=dhp, accumulate=True)
        return dx, dh_prev, dw_ih, dw_hh, db_ih, db_hh


class NTXentFn(torch.autograd.Function):
    """NT_Xent.forward (utils/losses.py:24-41); gradient comes out of the same launch."""

    @staticmethod
    def forward(ctx, z_i, z_j, temperature, grad_lo, grad_hi):
        z = torch.cat([z_i, z_j], 0)
        loss, dz, sim = ops.ntxent(z, temperature, want_grad=True, grad_lo=grad_lo, grad_hi=grad_hi)
        ctx.save_for_backward(dz)
        ctx.B = z_i.shape[0]
        ctx.mark_non_differentiable(sim)
        return loss[0], sim

    @staticmethod
    def backward(ctx, dloss, _dsim):
        (dz,) = ctx.saved_tensors
        g = dz * dloss
        return g[:ctx.B], g[ctx.B:], None, None, None


class DSMILFn(torch.autograd.Function):
    """MILNet.forward for a batch of equal-length bags (models/dsmil.py:9-16,64-81,104-113).

    One GEMM over X produces the queries Q (columns 0..127) and the instance scores (columns 128..128+C-1);
    the value projection is applied AFTER pooling: bag = (A^T X) Wv^T + bv, identical to A^T (X Wv^T + bv)
    because every column of the soft-max sums to one (dropout_v = 0).  Returns (classes [B,N,C], bag [B,C,d]).
    """
    QD = 128

    @staticmethod
    def forward(ctx, x, wc, bc, wq, bq, wv, bv):
        B, N, d = x.shape
        T = x.dtype
        C = wc.shape[0]
        QD = DSMILFn.QD
        LD = QD + ((C + 7) // 8) * 8
        w = torch.zeros((LD, d), dtype=torch.float32, device=x.device)
        bias = torch.zeros((LD,), dtype=torch.float32, device=x.device)
        w[:QD], w[QD:QD + C], bias[:QD], bias[QD:QD + C] = wq, wc, bq, bc
        x2 = x.reshape(B * N, d)
        Y = ops.gemm_nt(x2, w if T == torch.float32 else ops.cast(w, T), epi=ops.EPI_BIAS, bias=bias,
                        out_dtype=torch.float32)
        m = ops.dsmil_argmax(Y[:, QD:], B, N, C)
        qmax = ops.gather_rows(Y, m, B, C, N, 0, QD)
        A = ops.dsmil_attn(Y, 0, qmax, B, N, C)
        Z = ops.weighted_rowsum(x, A)
        bag = ops.gemm_nt(Z.view(B * C, d), wv, epi=ops.EPI_BIAS, bias=bv).view(B, C, d)
        classes = Y[:, QD:QD + C].reshape(B, N, C)
        ctx.save_for_backward(x, Y, m, qmax, A, Z, wv)
        ctx.meta = (B, N, d, C, LD)
        ctx.mark_non_differentiable(m)
        return classes, bag, m

    @staticmethod
    def backward(ctx, dclasses, dbag, _dm):
        x, Y, m, qmax, A, Z, wv = ctx.saved_tensors
        B, N, d, C, LD = ctx.meta
        T, QD = x.dtype, DSMILFn.QD
        dev = x.device
        x2 = x.reshape(B * N, d)
        dbag2 = (dbag if dbag is not None else torch.zeros((B, C, d), device=dev)).reshape(B * C, d).contiguous()
        dwv = ops.gemm_tn(dbag2, Z.view(B * C, d))
        dbv = ops.colsum(dbag2)
        dZ = ops.gemm_nt(dbag2, ops.transpose_cast(wv, torch.float32)).view(B, C, d)
        dA = ops.rows_dot(x, dZ)
        dY = torch.zeros((B * N, LD), dtype=torch.float32, device=dev)
        if dclasses is not None:
            dY[:, QD:QD + C] = dclasses.reshape(B * N, C)
        dqmax = ops.dsmil_attn_bwd(A, dA, Y, 0, qmax, dY, B, N, C)
        dW = ops.gemm_tn(dY if T == torch.float32 else ops.cast(dY, T), x2)                 # [LD, d]
        db = ops.colsum(dY)
        xm = ops.gather_rows(x2, m, B, C, N, 0, d)                                          # critical instances
        dwq = ops.gemm_tn(dqmax if T == torch.float32 else ops.cast(dqmax, T), xm, out=dW[:QD].contiguous())
        dbq = ops.colsum(dqmax, out=db[:QD].contiguous(), accumulate=True)
        return None, dW[QD:QD + C].contiguous(), db[QD:QD + C].contiguous(), dwq, dbq, dwv, dbv
