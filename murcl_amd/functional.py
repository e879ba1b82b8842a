"""autograd.Functions that stitch the HIP kernels into differentiable ops.

Forward and backward are sequences of C-ABI launches (murcl_amd.ops); torch supplies
only tensor storage and the autograd graph.  No CPU / eager-PyTorch fallback exists:
CPU tensors raise.
"""
import math

import torch

from . import ops


def _flat2(x):
    return x.reshape(-1, x.shape[-1])


# Direct gradient accumulation.  With FlatAdam every parameter's ``.grad`` is a zeroed view of one flat buffer before
# the backward pass starts, so the weight-gradient GEMM / column-sum kernels can add into it themselves (their split-M
# reduction accumulates into the output anyway) and hand autograd ``None``: no zero-filled temporary and no
# AccumulateGrad add per parameter.  Opt-in because it bypasses per-parameter autograd hooks.
_DIRECT = False


def set_direct_grad(flag):
    """Enable/disable accumulation of parameter gradients straight into pre-seated ``.grad`` buffers."""
    global _DIRECT
    _DIRECT = bool(flag)


_MILESTONE = None
# bit l -> encoder layer l+1 loads its input with the non-temporal policy.  Layer 3's input (H2) is not read again before
# the backward pass, while its output (H3) is what the pooling kernel streams next: kept out of the Infinity Cache, H2
# leaves more of H3 there (K2 forward 67.6 -> 62.4 us inside the step; the other layers measured neutral-to-slower).
_STREAM_A = 4
# Which of two BUILT forms a call takes where both cover its shape.  The right-hand side of each line is the form every default
# configuration runs; the other one is the general chain that shapes outside the fused kernels' reach take anyway (other widths,
# more classes, per-layer gradient milestones ...), so both stay tested: the parity tests flip these through monkeypatch
# (tests/test_gpu_modules.py) - they are test hooks, not environment switches (round 5: the MURCL_* variables are gone).
_FUSED_GATE = True        # CLAM, forward-only calls: the gate score from the gate GEMM's epilogue, no [B*N, 2D] pre-activations
_GATE_U = True            # CLAM, training chain: score + pre-activations from one gate GEMM, one-pass gate backward
_FUSED_FC_DROP = True     # CLAM: the seeded Dropout behind the first layer's ReLU inside that GEMM's epilogue
_FUSED_INST = True        # CLAM: the instance branch as one launch forward, one backward
_DSMIL_REASSOC = True     # DSMIL: attention logits as X . (Wq^T q_max) - no GEMM over all patches (C <= 4)
_DSMIL_ONEPASS = True     # ... with attention + pooling, and their backward, in one pass over X each
_DSMIL_QV = True          # ... and the [B*C]-row algebra around them as three launches
_DSMIL_X3 = True          # DSMIL's long f32 GEMMs (the literal-order chain) as a 3-term bf16 split
_GROUP_WGRAD = True       # the encoder weight gradients of a backward pass as one grouped launch
_FOLD_BIAS = True         # encoder bias gradients folded into the wgrad reduce
_FRAG_WEIGHTS = True      # ABMIL bf16 fast path: the K = 512 weight operands as FRAGMENT-ORDER views (ops.is_frag; round 6)


def _frag_specs(w1, w2, w3, wa, T):
    """The seven weight views of the bf16 ABMIL chain with every K = 512 operand in fragment order: W1..W3, Wa (forward, pooling) and
    W3^T, W2^T (the masked input gradients); Wa^T [512,128] (the rank-1 input gradient, K = 128) stays row-major."""
    return [(w1, False, T, "frag"), (w2, False, T, "frag"), (w3, False, T, "frag"), (wa, False, T, "frag"),
            (wa, True, T), (w3, True, T, "frag"), (w2, True, T, "frag")]


# CUs the launches of the aggregator's backward pass - pooling backward up to the last input gradient - are sized for while a
# collective may be running beside them (None: ops.cu_budget() as it is).  The head group's gradient all-reduce is launched when
# backward reaches the aggregator outputs (dist.OverlappedGradReduce) and is in flight for roughly these launches (19.4 MB at the
# xGMI ring's ~150 GB/s = 0.2-0.3 ms against 65 + 65 + 82 + 2 x 133 us); the forward pass, the grouped weight gradients behind them
# and the optimizer run with no collective beside them and keep the full chip.
_OVERLAP_BUDGET = None


def set_overlap_cu_budget(cus):
    """``cus`` (or None) = the CU budget in force for the backward launches a gradient all-reduce overlaps."""
    global _OVERLAP_BUDGET
    _OVERLAP_BUDGET = None if cus is None else int(cus)


class _overlap_budget:
    """Context: the persistent launches inside are sized for ``_OVERLAP_BUDGET`` CUs (murcl_set_cu_budget), the budget that was
    in force before is restored on exit.  Launchers read the budget when they enqueue, so this is host-side bookkeeping only."""

    def __enter__(self):
        self.prev = None
        if _OVERLAP_BUDGET is not None:
            self.prev = ops.cu_budget()
            ops.set_cu_budget(min(self.prev, _OVERLAP_BUDGET))
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            ops.set_cu_budget(self.prev)
        return False


def set_grad_milestone(callback):
    """``callback(params)`` is called from inside the aggregator's backward as soon as the kernels that complete the
    gradients of ``params`` are enqueued (direct-gradient mode only).  A data-parallel reducer uses it to start their
    all-reduce under the remaining backward kernels; only meaningful when the aggregator runs once per optimizer step."""
    global _MILESTONE
    _MILESTONE = callback


def _final(*params):
    if _MILESTONE is not None and all(_direct(p) for p in params):
        _MILESTONE(params)


def _direct(p):
    # (non-leaf tensors - a row of ``attention.2`` handed to one head of ABMIL(K > 1) - have no pre-seated gradient)
    return _DIRECT and p is not None and p.is_leaf and p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32


# Which parameters received a gradient since their optimizer last stepped.  torch.optim.Adam skips parameters whose
# ``.grad`` is None (never reached by backward: ABMIL.fc, CLAM's classifiers in pre-training, DSMIL's fcc): no weight
# decay, no moment update, no step count.  With pre-seated zeroed gradient views "never reached" is invisible in the
# buffer, so every writer announces itself here: the direct-accumulation helpers below, and an autograd
# post-accumulate hook that FlatAdam registers for gradients that arrive through AccumulateGrad.
_TOUCHED = set()


def _touch(*params):
    for p in params:
        if p is not None:
            _TOUCHED.add(id(p))


def _wgrad(dy, x, w, b=None, bias_parts=None):
    """dW = dy^T x [N1,N2]; returns it, or adds it to w.grad and returns None.  ``bias_parts`` (direct mode only): the
    partial column-sum rows of dy that the kernel which produced dy left behind - the bias gradient, added to b.grad by
    the same launches."""
    if _direct(w):
        if bias_parts is not None:
            ops.gemm_tn(dy, x, out=w.grad, colsum_into=b.grad.view(-1), colsum_parts=bias_parts)
            _touch(w, b)
        else:
            ops.gemm_tn(dy, x, out=w.grad)
            _touch(w)
        return None
    assert bias_parts is None
    return ops.gemm_tn(dy, x)


def _wgrad_group(items):
    """[(dy, x, w, b, bias_parts)] -> [dW or None]: ``_wgrad`` for several layers in one grouped launch (``ops.gemm_tn_grouped``)."""
    probs = []
    for dy, x, w, b, parts in items:
        if _direct(w):
            probs.append((dy, x, w.grad, b.grad.view(-1) if parts is not None else None, parts))
            _touch(w)
            if parts is not None:
                _touch(b)
        else:
            assert parts is None
            probs.append((dy, x, None, None, None))
    Cs = ops.gemm_tn_grouped(probs, fresh=True)           # (products without a pre-seated gradient are written, not accumulated: no fill)
    return [None if _direct(it[2]) else C for it, C in zip(items, Cs)]


# Deferred weight gradients.  A recurrent head that is stepped T times per optimizer step (the MuRCL loop: T patch steps
# through ONE Full_layer) produces T weight gradients per parameter, each a skinny [128 x N]^T [128 x K] product whose
# [N x K] f32 output leaves as float atomics (6.3 MB for W_ih, 12.6 MB for W_hh - 15 us apiece, T times).  Inside a
# ``deferred_wgrads()`` block the (dy, x) pairs are only queued; on exit each parameter gets ONE product over the
# concatenated T*128 rows: the same sum, one launch and one pass of atomics instead of T.
_DEFERRED = None
_DEFER_ON = True           # test hook (see the list at the top)


class deferred_wgrads:
    def __enter__(self):
        global _DEFERRED
        self.prev, _DEFERRED = _DEFERRED, ({} if _DEFER_ON else None)
        return self

    def __exit__(self, *exc):
        global _DEFERRED
        queue, _DEFERRED = _DEFERRED, self.prev
        if exc[0] is None and queue:
            _flush(queue)
        return False


def _flush(queue):
    """The queued (dy, x) pairs of every parameter as one product each, up to four small f32 products per launch."""
    probs = []
    for w, b, dys, xs in queue.values():
        dy = dys[0] if len(dys) == 1 else torch.cat(dys, 0)
        x = xs[0] if len(xs) == 1 else torch.cat(xs, 0)
        probs.append((dy, x, w.grad, b.grad.view(-1) if b is not None else None, None))
    queue.clear()
    for i in range(0, len(probs), 4):
        ops.gemm_tn_grouped(probs[i:i + 4])


def flush_deferred():
    """Run what the active ``deferred_wgrads`` block has queued so far (a gradient all-reduce that starts inside the backward pass
    needs the gradients of the layers behind it complete: dist.OverlappedGradReduce)."""
    if _DEFERRED:
        _flush(_DEFERRED)


def _defer(dy, x, w, b):
    """Queue (dy, x) for w (and b) when a deferral block is active and both accumulate directly; True if queued."""
    if _DEFERRED is None or not _direct(w) or (b is not None and not _direct(b)):
        return False
    ent = _DEFERRED.get(id(w))
    if ent is None:
        ent = _DEFERRED[id(w)] = (w, b, [], [])
    elif (ent[1] is None) != (b is None):
        return False
    ent[2].append(dy)
    ent[3].append(x)
    _touch(w, b)
    return True


def _wbgrad(dy, x, w, b):
    """(dW, db) of one Linear; in direct mode both are added to the flat gradient buffer by ONE launch."""
    if _defer(dy, x, w, b):
        return None, None
    if _direct(w) and _direct(b):
        ops.gemm_tn(dy, x, out=w.grad, colsum_into=b.grad.view(-1))
        _touch(w, b)
        return None, None
    return _wgrad(dy, x, w), _bgrad(dy, b)


def _bgrad(dy, b):
    """db = column sums of dy."""
    if _direct(b):
        ops.colsum(dy, out=b.grad.view(-1), accumulate=True)
        _touch(b)
        return None
    return ops.colsum(dy)


def _pgrads(*pairs):
    """``[(gradient tensor or None, parameter)]`` -> the list of what autograd should still see: parameters that accumulate directly
    get their gradients added to the pre-seated buffers by ONE launch for all of them (and None here), the others get theirs back."""
    direct = [(g, p) for g, p in pairs if g is not None and _direct(p)]
    if len(direct) > 1:
        ops.add_lists([(g.reshape(p.grad.shape) if g.shape != p.grad.shape else g, p.grad) for g, p in direct])
        _touch(*(p for _, p in direct))
        done = {id(p) for _, p in direct}
        return [None if (g is None or id(p) in done) else g for g, p in pairs]
    return [g for g, _ in pairs]


def _pgrad(g, p):
    """A gradient that a kernel already produced as its own tensor."""
    if _direct(p):
        p.grad.add_(g.view_as(p.grad))
        _touch(p)
        return None
    return g


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) for small (bag-level) f32 matrices; act in {none, relu}."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        x2 = _flat2(x).contiguous()
        y = ops.gemm_nt(x2, w, epi=ops.EPI_BIAS_RELU if relu else ops.EPI_BIAS, bias=b)
        ctx.save_for_backward(x2, w, y if relu else None, b)
        ctx.relu = relu
        ctx.xshape = x.shape
        return y if x.dim() == 2 else y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, y, b = ctx.saved_tensors
        dy2 = _flat2(dy).contiguous()
        if ctx.relu:
            dy2 = ops.relu_bwd(dy2, y)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(dy2, ops.transposed(w)).view(ctx.xshape)
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            dw, db = _wbgrad(dy2, x2, w, b)
        else:
            if ctx.needs_input_grad[1]:
                dw = _wgrad(dy2, x2, w)
            if ctx.needs_input_grad[2]:
                db = _bgrad(dy2, b)
        return dx, dw, db, None


class ABMILFn(torch.autograd.Function):
    """Whole ABMIL.bag_forward for a batch of equal-length bags (models/abmil.py:35-45).

    x [B,N,d] in the compute dtype (f32 parity path / bf16 throughput path); parameters f32.
    Patch-level tensors (H1..H3, dZ*, dT) live in the compute dtype with f32 accumulation;
    bag-level tensors are f32.  Returns (out [B,L], att, stats): ``attention_rows(att, stats)`` is A [B,N] (non-differentiable).
    """

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, wa, ba, wb, bb, wd, bd, drops=None, grad_on=True):
        """``drops`` = None or the two Dropout(p) keep masks after encoder layers 1 and 2 (abmil.py:12-19): ``ops.DropSeed``s
        (training: the masks are generated inside the passes that apply them) or materialised keep-multiplier tensors
        (values 0 or 1/keep, parity tests).  ``grad_on``: the caller's ``torch.is_grad_enabled()`` - inside ``forward`` autograd
        is always off and ``ctx.needs_input_grad`` stays True for parameters under ``torch.no_grad()``, so only the caller knows
        that no backward pass can follow (frozen encoder of stage 2, validation): then no ReLU masks are written."""
        B, N, d = x.shape
        T = x.dtype
        x2 = x.reshape(B * N, d)
        L = w3.shape[0]
        pool_fast = (L == 512 and wa.shape[0] == 128)       # the one-pass K2 kernel is built for L = 512, D = 128
        # compute-dtype copies of W1..W3, Wa for this pass and W2^T, W3^T, Wa^T for the dgrads of the backward pass: one
        # launch, and only when a parameter changed since they were last built (ops.weight_views)
        wmats = (w1, w2, w3, wa)
        # bf16 + panel-friendly shapes: weight-stationary GEMMs that also emit 1-bit ReLU masks
        fast = (T == torch.bfloat16 and d == 512 and pool_fast and ops.panel_supported(B * N, L, 512, ops.PG_BIAS_RELU)
                and ops.panel_supported(B * N, L, 128, ops.PG_RANK1_MASK, N))
        if all(w.dtype == torch.float32 and w.dim() == 2 and w.is_contiguous() for w in wmats):
            tr = [(wa, True, T), (w3, True, T), (w2, True, T)]
            if T == torch.float32:
                w1c, w2c, w3c, wac = wmats
                wat, w3t, w2t = ops.weight_views(tr)
            elif fast and _FRAG_WEIGHTS:
                w1c, w2c, w3c, wac, wat, w3t, w2t = ops.weight_views(_frag_specs(w1, w2, w3, wa, T))
            else:
                w1c, w2c, w3c, wac, wat, w3t, w2t = ops.weight_views([(w, False, T) for w in wmats] + tr)
        else:
            w1c, w2c, w3c, wac = (ops.cast(w.contiguous(), T) for w in wmats)
            wat, w3t, w2t = (ops.transpose_cast(w, T) for w in (wa, w3, w2))
        seeded = drops is not None and isinstance(drops[0], ops.DropSeed)

        def drop(h, k, bits):
            """Dropout after a ReLU, in place: h *= keep.  -> the 1-bit mask of the surviving positive entries (fast path)."""
            if seeded and h.shape[0] % 32 == 0 and h.shape[1] % 128 == 0:
                return ops.dropout_relu_bitmask(h, k, want_bits=bits)
            ops.mul(h, ops.dropout_mask(h.shape, T, k.keep_p, h.device, seed=k.seed) if seeded else k.to(T).reshape(h.shape))
            return ops.relu_bitmask(h) if bits else None

        if fast:
            nt = _STREAM_A
            keep = bool(grad_on) and any(ctx.needs_input_grad)     # forward-only passes: no ReLU masks to write
            h1, m1, _ = ops.panel_gemm(x2, w1c, ops.PG_BIAS_RELU, bias=b1, want_bitmask=keep and drops is None, stream_a=bool(nt & 1))
            if drops is not None:
                m1 = drop(h1, drops[0], keep)
            # layer 2 walks the rows backwards (layer 1 has just written the high rows of h1), layer 3 forwards again,
            # and the pooling kernel backwards: every pass starts on what its producer left in the Infinity Cache
            h2, m2, _ = ops.panel_gemm(h1, w2c, ops.PG_BIAS_RELU, bias=b2, want_bitmask=keep and drops is None, reverse=True,
                                       stream_a=bool(nt & 2))
            if drops is not None:
                m2 = drop(h2, drops[1], keep)
            h3, m3, _ = ops.panel_gemm(h2, w3c, ops.PG_BIAS_RELU, bias=b3, want_bitmask=keep, stream_a=bool(nt & 4))
        else:
            m1 = m2 = m3 = None
            h1 = ops.gemm_nt(x2, w1c, epi=ops.EPI_BIAS_RELU, bias=b1)
            if drops is not None:
                drop(h1, drops[0], False)
            h2 = ops.gemm_nt(h1, w2c, epi=ops.EPI_BIAS_RELU, bias=b2)
            if drops is not None:
                drop(h2, drops[1], False)
            h3 = ops.gemm_nt(h2, w3c, epi=ops.EPI_BIAS_RELU, bias=b3)
        if pool_fast:
            # ONE launch for the K2 row (scores + chunk partials); the per-bag merge is part of the decoder launch below and the
            # normalised attention row is formed by the backward pass (or on demand: ``attention_rows``)
            scores, part = ops.abmil_pool_partials(h3.view(B, N, L), wac, ba, wb, bb)
            out, M, ml = ops.abmil_pool_decoder(part, B, N, T, wd, bd)
            A = None
        else:
            # any L / D (abmil.py:8-30 takes them as arguments): the same attention pooling as a chain of the generic kernels -
            # projection GEMM, tanh score, row soft-max, /sqrt(N) (abmil.py:40-41), weighted row sum
            U = ops.gemm_nt(h3, wac, epi=ops.EPI_BIAS, bias=ba)                                     # [B*N, D]
            s = ops.gated_score_fwd(U, wb.reshape(-1).contiguous(), bb, gated=False).view(B, N)
            Asm = ops.softmax_rows(s)
            A = ops.mul(Asm, torch.full_like(Asm, 1.0 / (N ** 0.5)), out=torch.empty_like(Asm))
            M = ops.weighted_rowsum(h3.view(B, N, L), A.view(B, N, 1)).view(B, L)
            scores, ml = U, Asm                                                                    # what the generic backward needs
            out = ops.gemm_nt(M, wd, epi=ops.EPI_BIAS_RELU, bias=bd)
        ctx.save_for_backward(x2, h1, h2, h3, scores, A, M, ml, out, w1, w2, w3, wa, ba, wb, wd, wac, m1, m2, m3,
                              b1, b2, b3, bb, bd, wat, w3t, w2t)
        ctx.dims = (B, N, d)
        ctx.pool_fast = pool_fast
        # the surviving entries of a keep mask all equal 1/keep: the masked dgrads below run unscaled and the (linear) factors
        # are applied to the few gradients behind them
        ctx.drop_scale = None
        if drops is not None:
            ctx.drop_scale = tuple(1.0 / k.keep_q if isinstance(k, ops.DropSeed) else float(k.max().item()) for k in drops)
        # second / third result: what ``attention_rows`` turns into A [B,N] - (raw scores, (m, l)) from the one-pass pooling kernel,
        # (A, None) from the generic chain
        att, stats = (scores, ml) if pool_fast else (A, None)
        ctx.mark_non_differentiable(*([att, stats] if pool_fast else [att]))
        ctx.set_materialize_grads(False)         # no zero-filled dA (a launch) for the attention output nobody differentiates
        return out, att, stats

    @staticmethod
    def backward(ctx, dout, _datt, _dstats):
        if dout is None:
            return (None,) * 15
        if ctx.drop_scale is not None or not ctx.pool_fast:
            return ABMILFn._backward_general(ctx, dout) + (None,)
        return ABMILFn._backward_default(ctx.saved_tensors, ctx.dims, dout, ctx.needs_input_grad[0]) + (None, None)

    @staticmethod
    def _backward_default(saved, dims, dout, need_dx):
        """The backward pass of the default configuration on explicit tensors (``ABMILFn.backward`` hands it one call's saved
        tensors, ``EncoderSession`` the activations of all patch steps of a training step as one batch)."""
        (x2, h1, h2, h3, scores, A, M, ml, out, w1, w2, w3, wa, ba, wb, wd, wac, m1, m2, m3,
         b1, b2, b3, bb, bd, wat, w3t, w2t) = saved
        B, N, d = dims
        T = x2.dtype
        L = h3.shape[1]
        # decoder (bag level, f32)
        dpre = ops.relu_bwd(dout.contiguous(), out)
        dwd, dbd = _wbgrad(dpre, M, wd, bd)
        dM = ops.gemm_nt(dpre, ops.transposed(wd))
        # attention pooling.  From here to the last input gradient a data-parallel step has the head group's all-reduce in flight:
        # these launches leave its channel workgroups their CUs (_overlap_budget: a no-op on one GPU)
        budget_scope = _overlap_budget()
        budget_scope.__enter__()
        direct_k2 = _direct(ba) and _direct(wb) and _direct(bb)      # the kernel's atomics add straight into the grads
        into_k2 = (ba.grad, wb.grad.view(-1), bb.grad) if direct_k2 else None
        # The row scale of the rank-1 term below is A = softmax(s)/sqrt(N), which the forward pass no longer forms: the bf16 panel
        # kernel makes it from the raw scores and (m, l) in its epilogue; the f32 GEMM takes the rows this pass leaves behind
        if m3 is not None:
            dT, dba, dwb, dbb = ops.abmil_pool_bwd(h3.view(B, N, L), wac, ba, wb, scores, ml, M, dM, into=into_k2)
        else:
            dT, dba, dwb, dbb, A = ops.abmil_pool_bwd(h3.view(B, N, L), wac, ba, wb, scores, ml, M, dM, into=into_k2, want_A=True)
        dwa = _wgrad(dT, h3, wa)
        if direct_k2:
            _touch(ba, wb, bb)
            _final(wa, ba, wb, bb, wd, bd)
        # encoder layer 3: dZ3 = (dT Wa + A (x) dM) * relu'(H3)
        if m3 is not None:
            def into(b):                                                  # bias gradients straight from the epilogue
                if _direct(b):
                    _touch(b)
                    return b.grad.view(-1)
                return None
            # bias gradients: the dgrad kernels leave per-workgroup partial column sums; when weight AND bias accumulate
            # directly into the flat gradient buffer the rows are added up inside the weight gradient's reduce launch
            fold = lambda w, b: _FOLD_BIAS and _direct(w) and _direct(b)      # noqa: E731
            f3, f2, f1 = fold(w3, b3), fold(w2, b2), fold(w1, b1)
            # the three weight gradients wait until the last input gradient exists and run as ONE grouped launch (one round of
            # workgroups, one reduce launch: ops.gemm_tn_grouped) unless a data-parallel reducer asked for per-layer milestones
            grouped = _GROUP_WGRAD and _MILESTONE is None
            dz3, _, db3 = ops.panel_gemm(dT, wat, ops.PG_RANK1_MASK, bitmask=m3, rowscale=scores.view(-1), bias=ml, rank1=dM,
                                         rows_per_bag=N, colsum=True, colsum_into=None if f3 else into(b3), colsum_defer=f3)
            if not grouped:
                dw3 = _wgrad(dz3, h2, w3, b3, db3 if f3 else None)
                _final(w3, b3)
            dz2, _, db2 = ops.panel_gemm(dz3, w3t, ops.PG_MASK, bitmask=m2, colsum=True,
                                         colsum_into=None if f2 else into(b2), colsum_defer=f2)
            if not grouped:
                dw2 = _wgrad(dz2, h1, w2, b2, db2 if f2 else None)
                _final(w2, b2)
            dz1, _, db1 = ops.panel_gemm(dz2, w2t, ops.PG_MASK, bitmask=m1, colsum=True,
                                         colsum_into=None if f1 else into(b1), colsum_defer=f1)
            budget_scope.__exit__()                                # the grouped weight gradients run behind the collective: full chip
            if grouped:
                dw3, dw2, dw1 = _wgrad_group([(dz3, h2, w3, b3, db3 if f3 else None), (dz2, h1, w2, b2, db2 if f2 else None),
                                              (dz1, x2, w1, b1, db1 if f1 else None)])
                _final(w3, b3, w2, b2, w1, b1)
            db3, db2 = None if f3 else db3, None if f2 else db2
            if grouped:
                db1 = None if f1 else db1
        else:
            dz3, ws = ops.gemm_nt(dT, wat, epi=ops.EPI_RANK1_MASK, mask=h3, rowscale=A.view(-1),
                                  rank1=dM, rows_per_bag=N, colsum=True)
            db3 = _bgrad(ws, b3)
            dw3 = _wgrad(dz3, h2, w3)
            _final(w3, b3)
            dz2, ws = ops.gemm_nt(dz3, w3t, epi=ops.EPI_MASK, mask=h2, colsum=True)
            db2 = _bgrad(ws, b2)
            dw2 = _wgrad(dz2, h1, w2)
            _final(w2, b2)
            dz1, ws = ops.gemm_nt(dz2, w2t, epi=ops.EPI_MASK, mask=h1, colsum=True)
            db1 = _bgrad(ws, b1)
            budget_scope.__exit__()
        if m3 is not None and grouped:
            pass
        elif m3 is not None and f1:
            dw1, db1 = _wgrad(dz1, x2, w1, b1, db1), None
        else:
            dw1 = _wgrad(dz1, x2, w1)
        dx = None
        if need_dx:
            dx = ops.gemm_nt(dz1, ops.transpose_cast(w1, T)).view(B, N, d)
        if direct_k2:
            dba = dwb = dbb = None
        else:
            dba, dwb, dbb = _pgrad(dba, ba), _pgrad(dwb.reshape(1, -1), wb), _pgrad(dbb, bb)
        return dx, dw1, db1, dw2, db2, dw3, db3, dwa, dba, dwb, dbb, dwd, dbd

    @staticmethod
    def _backward_general(ctx, dout):
        """The configurations outside the tuned default (``--dropout`` > 0 while training, ``--L`` / ``--D`` other than 512 /
        128): the same backward pass with every gradient returned to autograd as its own tensor, so that the dropout
        factors can be applied to them (no direct accumulation into the flat gradient buffer on this path)."""
        (x2, h1, h2, h3, scores, A, M, ml, out, w1, w2, w3, wa, ba, wb, wd, wac, m1, m2, m3,
         b1, b2, b3, bb, bd, wat, w3t, w2t) = ctx.saved_tensors
        B, N, d = ctx.dims
        T = x2.dtype
        L = h3.shape[1]
        dpre = ops.relu_bwd(dout.contiguous(), out)
        dwd, dbd = ops.gemm_tn(dpre, M), ops.colsum(dpre)
        dM = ops.gemm_nt(dpre, ops.transposed(wd))
        if ctx.pool_fast and m3 is not None:
            dT, dba, dwb, dbb = ops.abmil_pool_bwd(h3.view(B, N, L), wac, ba, wb, scores, ml, M, dM)
            dwb = dwb.reshape(1, -1)
        elif ctx.pool_fast:
            dT, dba, dwb, dbb, A = ops.abmil_pool_bwd(h3.view(B, N, L), wac, ba, wb, scores, ml, M, dM, want_A=True)
            dwb = dwb.reshape(1, -1)
        else:
            U, Asm = scores, ml
            dA = ops.rows_dot(h3.view(B, N, L), dM.view(B, 1, L)).view(B, N)
            dAs = ops.mul(dA, torch.full_like(dA, 1.0 / (N ** 0.5)))                               # A = softmax / sqrt(N)
            ds = ops.softmax_rows_bwd(Asm, dAs).view(-1)
            dT, dwb, dbb, dba = ops.gated_score_bwd(U, wb.reshape(-1).contiguous(), ds, gated=False)
            dwb, dba = dwb.view(1, -1), dba.contiguous()
        dwa = ops.gemm_tn(dT, h3)
        if m3 is not None:
            if ctx.pool_fast:
                dz3, _, db3 = ops.panel_gemm(dT, wat, ops.PG_RANK1_MASK, bitmask=m3, rowscale=scores.view(-1), bias=ml, rank1=dM,
                                             rows_per_bag=N, colsum=True)
            else:
                dz3, _, db3 = ops.panel_gemm(dT, wat, ops.PG_RANK1_MASK, bitmask=m3, rowscale=A.view(-1), rank1=dM, rows_per_bag=N, colsum=True)
            dz2, _, db2 = ops.panel_gemm(dz3, w3t, ops.PG_MASK, bitmask=m2, colsum=True)
            dz1, _, db1 = ops.panel_gemm(dz2, w2t, ops.PG_MASK, bitmask=m1, colsum=True)
        else:
            dz3, ws = ops.gemm_nt(dT, wat, epi=ops.EPI_RANK1_MASK, mask=h3, rowscale=A.view(-1), rank1=dM, rows_per_bag=N, colsum=True)
            db3 = ops.colsum(ws)
            dz2, ws = ops.gemm_nt(dz3, w3t, epi=ops.EPI_MASK, mask=h2, colsum=True)
            db2 = ops.colsum(ws)
            dz1, ws = ops.gemm_nt(dz2, w2t, epi=ops.EPI_MASK, mask=h1, colsum=True)
            db1 = ops.colsum(ws)
        dw3, dw2, dw1 = ops.gemm_tn(dz3, h2), ops.gemm_tn(dz2, h1), ops.gemm_tn(dz1, x2)
        dx = ops.gemm_nt(dz1, ops.transpose_cast(w1, T)).view(B, N, d) if ctx.needs_input_grad[0] else None
        if ctx.drop_scale is not None:
            # dZ2 = (dZ3 W3) * relu'(H2) * keep2 and dZ1 = (dZ2 W2) * relu'(H1) * keep1, the masks above hold relu' AND kept
            s2 = ctx.drop_scale[1]
            s1 = ctx.drop_scale[0] * s2
            dw2, db2, dw1, db1 = dw2 * s2, db2 * s2, dw1 * s1, db1 * s1
            dx = dx * s1 if dx is not None else None
        return dx, dw1, db1, dw2, db2, dw3, db3, dwa, dba, dwb, dbb, dwd, dbd, None


def attention_rows(att, stats):
    """The attention rows A [B,N] = softmax(s)/sqrt(N) of an ``ABMILFn`` / ``ABMILStepFn`` call from its second and third result:
    the one-pass pooling kernel hands back raw scores + (m, l) and A costs one small launch HERE, when somebody asks for it
    (``ABMIL.last_attention``); the generic chain already has A."""
    return att if stats is None else ops.abmil_attention(att, stats)


def abmil_fast_path(rows, N, d, L, D, dtype):
    """Does an ABMIL call of this shape take the bf16 weight-stationary encoder + one-pass pooling kernels?"""
    return (dtype == torch.bfloat16 and d == 512 and L == 512 and D == 128 and ops.panel_supported(rows, L, 512, ops.PG_BIAS_RELU)
            and ops.panel_supported(rows, L, 128, ops.PG_RANK1_MASK, N))


class EncoderSession:
    """ONE aggregator backward for the T patch steps of a sequential training step (train_MuRCL.py:233-304 at train_stage 3).

    The PPO sampler picks step t+1's windows from step t's aggregator states, so the T forward passes cannot be batched -
    but their backward passes can: every step's sub-bags and activations are written into row blocks of ONE set of buffers
    (``x``, ``h1..h3``, ReLU bit masks, scores, pooled vectors), the per-step autograd nodes (``ABMILStepFn``) only collect their
    upstream gradients, and the node autograd reaches last runs the backward kernels once over all T * bags bags: each dgrad /
    wgrad / pooling-backward kernel once at full size instead of T times at 1/T of it (a launch costs ~15 us before its first tile
    and the weight gradients re-reduce their partial tiles per launch)."""

    def __init__(self, steps, bags, N, d, L, dtype, device):
        self.steps, self.bags, self.N, self.d, self.L = steps, bags, N, d, L
        R, Bt = steps * bags * N, steps * bags
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=device)      # noqa: E731
        self.x = e((steps, bags, N, d), dtype)
        self.h1, self.h2, self.h3 = e((R, L), dtype), e((R, L), dtype), e((R, L), dtype)
        self.m1, self.m2, self.m3 = (e((R, L // 8), torch.uint8) for _ in range(3))
        self.scores = e((Bt, N), torch.float32)
        self.M, self.ml, self.out = e((Bt, L), torch.float32), e((Bt, 2), torch.float32), e((Bt, L), torch.float32)
        self.t, self.pending, self.dout, self.weights = 0, 0, [None] * steps, None

    def views(self, t):
        """The [bags, N, d] block step t's sub-bags are gathered into (``subbag_views(out=...)``)."""
        return self.x[t]

    def rows(self, buf, t, per=None):
        per = self.bags * self.N if per is None else per
        return buf[t * per:(t + 1) * per]


class ABMILStepFn(torch.autograd.Function):
    """One patch step's ABMIL forward inside an ``EncoderSession`` (default shape, bf16): same kernels as ``ABMILFn``, results in
    the session's row blocks; the backward only files its upstream gradient until the session's last node runs them all."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, wa, ba, wb, bb, wd, bd, session):
        s, t = session, session.t
        B, N, d = x.shape
        T = x.dtype
        if not (t < s.steps and (B, N, d) == (s.bags, s.N, s.d)):
            # (a session left behind by a step that raised: drop it rather than trip every later call)
            raise RuntimeError("EncoderSession: shape / step count differ from what it was built for - a previous step may have "
                               "raised before clearing model.encoder.session; set it to None")
        x2 = s.x[t].view(B * N, d)
        if x.data_ptr() != x2.data_ptr():
            x2.copy_(x.reshape(B * N, d))
        w1c, w2c, w3c, wac, wat, w3t, w2t = ops.weight_views(_frag_specs(w1, w2, w3, wa, T) if _FRAG_WEIGHTS else
                                                             [(w, False, T) for w in (w1, w2, w3, wa)] +
                                                             [(wa, True, T), (w3, True, T), (w2, True, T)])
        nt = _STREAM_A
        h1, _, _ = ops.panel_gemm(x2, w1c, ops.PG_BIAS_RELU, bias=b1, want_bitmask=True, stream_a=bool(nt & 1),
                                  out=s.rows(s.h1, t), bitmask_out=s.rows(s.m1, t))
        h2, _, _ = ops.panel_gemm(h1, w2c, ops.PG_BIAS_RELU, bias=b2, want_bitmask=True, reverse=True, stream_a=bool(nt & 2),
                                  out=s.rows(s.h2, t), bitmask_out=s.rows(s.m2, t))
        h3, _, _ = ops.panel_gemm(h2, w3c, ops.PG_BIAS_RELU, bias=b3, want_bitmask=True, stream_a=bool(nt & 4),
                                  out=s.rows(s.h3, t), bitmask_out=s.rows(s.m3, t))
        blk = lambda buf: s.rows(buf, t, B)                                          # noqa: E731
        scores, part = ops.abmil_pool_partials(h3.view(B, N, s.L), wac, ba, wb, bb, scores=blk(s.scores))
        out, _, ml = ops.abmil_pool_decoder(part, B, N, T, wd, bd, out=(blk(s.out), blk(s.M), blk(s.ml)))
        s.weights = (w1, w2, w3, wa, ba, wb, wd, wac, b1, b2, b3, bb, bd, wat, w3t, w2t)
        s.t, s.pending = t + 1, s.pending + 1
        ctx.session, ctx.t = s, t
        ctx.mark_non_differentiable(scores, ml)
        ctx.set_materialize_grads(False)
        return out, scores, ml

    @staticmethod
    def backward(ctx, dout, _datt, _dstats):
        s = ctx.session
        s.dout[ctx.t] = dout
        s.pending -= 1
        if s.pending > 0:
            return (None,) * 14
        n = s.t                                                                      # steps that ran
        if all(g is None for g in s.dout[:n]):
            return (None,) * 14
        like = next(g for g in s.dout[:n] if g is not None)
        from .utils.views import adjacent as _adjacent, as_one as _as_one       # (module level would be a circular import)
        douts = [g if g is not None else torch.zeros_like(like) for g in s.dout[:n]]
        if all(g is not None for g in s.dout[:n]) and _adjacent(douts):
            dout_all = _as_one(douts)                                                # row blocks of one gradient (SessionOutFn): a re-view
        else:
            dout_all = torch.cat([g.contiguous() for g in douts], 0)
        (w1, w2, w3, wa, ba, wb, wd, wac, b1, b2, b3, bb, bd, wat, w3t, w2t) = s.weights
        R, Bt = n * s.bags * s.N, n * s.bags
        saved = (s.x.view(-1, s.d)[:R], s.h1[:R], s.h2[:R], s.h3[:R], s.scores[:Bt], None, s.M[:Bt], s.ml[:Bt], s.out[:Bt],
                 w1, w2, w3, wa, ba, wb, wd, wac, s.m1[:R], s.m2[:R], s.m3[:R], b1, b2, b3, bb, bd, wat, w3t, w2t)
        return ABMILFn._backward_default(saved, (Bt, s.N, s.d), dout_all, False) + (None,)


class SessionOutFn(torch.autograd.Function):
    """The aggregator outputs of all patch steps of an ``EncoderSession`` as ONE tensor [steps * bags, L] without a copy: the steps
    wrote them into consecutive row blocks of ``session.out``, so the forward is a view of that buffer and the backward hands every
    step the row block of the ONE upstream gradient that belongs to it (views again: ``ABMILStepFn.backward`` then finds its T
    upstream gradients adjacent in one buffer).  Replaces a concatenation of 2T tensors in the forward pass and, in the backward pass,
    T concatenations of the two views' gradients plus one of the T steps' (stage 3 of train_MuRCL.py: 8 ATen launches)."""

    @staticmethod
    def forward(ctx, session, *hs):
        ctx.n, ctx.rows = len(hs), session.bags
        ctx.set_materialize_grads(False)
        return session.out[:len(hs) * session.bags].view(len(hs) * session.bags, -1)

    @staticmethod
    def backward(ctx, d):
        if d is None:
            return (None,) * (1 + ctx.n)
        d = d.contiguous()
        return (None,) + tuple(d[t * ctx.rows:(t + 1) * ctx.rows] for t in range(ctx.n))


def session_whole(session, hs):
    """``hs`` = the per-step aggregator outputs [bags, L] of ``session`` in step order -> the [steps * bags, L] tensor that equals
    ``torch.cat(hs)`` (a view of the session's buffer, differentiable), or None when they are not the session's row blocks."""
    if session is None or not hs or len(hs) > session.steps:
        return None
    row = session.out.shape[1] * session.out.element_size()
    for t, h in enumerate(hs):
        if not (torch.is_tensor(h) and h.dtype == session.out.dtype and tuple(h.shape) == (session.bags, session.out.shape[1])
                and h.is_contiguous() and h.data_ptr() == session.out.data_ptr() + t * session.bags * row):
            return None
    return SessionOutFn.apply(session, *hs)


class GRUStepFn(torch.autograd.Function):
    """One nn.GRU time step (seq_len 1, PyTorch gate order); h_prev None == zeros."""

    @staticmethod
    def forward(ctx, x, h_prev, w_ih, w_hh, b_ih, b_hh):
        x = x.contiguous()
        if h_prev is None and x.dtype == torch.float32 and ops.gru_step_ok(x.shape[0], w_hh.shape[1], x.shape[1]):
            # restart: the input product and the gates in one launch (h W_hh^T = 0, gh = b_hh)
            h_new, gates, _ = ops.gru_step_fwd(b_ih.detach(), None, w_hh.detach(), b_hh.detach(), x=x, w_ih=w_ih.detach(), want_gh=False)
            ctx.save_for_backward(x, None, w_ih, w_hh, gates, b_hh.detach().view(1, -1), b_ih, b_hh)
            return h_new
        gi = ops.gemm_nt(x, w_ih, epi=ops.EPI_BIAS, bias=b_ih)
        if h_prev is None:
            gh = b_hh.detach().view(1, -1)                                   # W_hh . 0 + b_hh: one row, read by every batch row
        else:
            h_prev = h_prev.contiguous()
            if ops.gru_step_ok(h_prev.shape[0], h_prev.shape[1]):            # product + gates in one launch
                h_new, gates, gh = ops.gru_step_fwd(gi, h_prev, w_hh.detach(), b_hh.detach())
                ctx.save_for_backward(x, h_prev, w_ih, w_hh, gates, gh, b_ih, b_hh)
                return h_new
            gh = ops.gemm_nt(h_prev, w_hh, epi=ops.EPI_BIAS, bias=b_hh)
        h_new, gates = ops.gru_gates_fwd(gi, gh, h_prev)
        ctx.save_for_backward(x, h_prev, w_ih, w_hh, gates, gh, b_ih, b_hh)
        return h_new

    @staticmethod
    def backward(ctx, dh):
        x, h_prev, w_ih, w_hh, gates, gh, b_ih, b_hh = ctx.saved_tensors
        dgi, dgh, dhp = ops.gru_gates_bwd(dh.contiguous(), gates, gh, h_prev)
        dx = ops.gemm_nt(dgi, ops.transposed(w_ih)) if ctx.needs_input_grad[0] else None
        dw_ih, db_ih = _wbgrad(dgi, x, w_ih, b_ih)
        if h_prev is None:
            db_hh = _bgrad(dgh, b_hh)
            _touch(w_hh)                       # nn.GRU from a zero state: a zero gradient, but a gradient (Adam applies decay)
            dh_prev, dw_hh = None, (None if _direct(w_hh) else torch.zeros_like(w_hh))
        else:
            dw_hh, db_hh = _wbgrad(dgh, h_prev, w_hh, b_hh)
            dh_prev = None
            if ctx.needs_input_grad[1]:
                dh_prev = ops.gemm_nt(dgh, ops.transposed(w_hh), out=dhp, accumulate=True)
        return dx, dh_prev, dw_ih, dw_hh, db_ih, db_hh


class GRUSeqFn(torch.autograd.Function):
    """nn.GRU over a whole rollout from a zero hidden state (ActorCritic.evaluate, models/rlmil.py:99-112).

    x [T,B,I] -> all hidden states [T,B,H].  The input projection and every weight/bias gradient are one GEMM /
    one column sum over the T*B rows; only h W_hh^T and the gate kernels stay inside the time loop.
    """

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        T, B, _ = x.shape
        H = w_hh.shape[1]
        x2 = x.reshape(T * B, -1).contiguous()
        gi = ops.gemm_nt(x2, w_ih, epi=ops.EPI_BIAS, bias=b_ih).view(T, B, 3 * H)
        gh = torch.empty((T, B, 3 * H), dtype=torch.float32, device=x.device)
        gates = torch.empty_like(gh)
        hs = torch.empty((T, B, H), dtype=torch.float32, device=x.device)
        fused = ops.gru_step_ok(B, H)                                    # h W_hh^T and the gate math of a step in one launch
        whh, bhh = w_hh.detach(), b_hh.detach()
        gh0 = bhh.view(1, -1)                                            # W_hh . 0 + b_hh: ONE row that every batch row of step 0 reads (gh[0] stays unwritten)
        for t in range(T):
            if t and fused:
                ops.gru_step_fwd(gi[t], hs[t - 1], whh, bhh, hnew=hs[t], gates=gates[t], gh=gh[t])
                continue
            if t:
                ops.gemm_nt(hs[t - 1], w_hh, epi=ops.EPI_BIAS, bias=b_hh, out=gh[t])
            ops.gru_gates_fwd(gi[t], gh[t] if t else gh0, hs[t - 1] if t else None, hnew=hs[t], gates=gates[t])
        ctx.save_for_backward(x2, w_ih, w_hh, gates, gh, hs, b_ih, b_hh)
        return hs

    @staticmethod
    def backward(ctx, dhs):
        x2, w_ih, w_hh, gates, gh, hs, b_ih, b_hh = ctx.saved_tensors
        T, B, H = hs.shape
        dhs = dhs.contiguous()
        dgi, dgh = torch.empty_like(gh), torch.empty_like(gh)
        w_hh_t = ops.transposed(w_hh)
        gh0 = b_hh.detach().view(1, -1)                                  # step 0's gh (see forward)
        if T > 1 and ops.gru_step_ok(B, H):
            # back through time, one launch per step: dh_{t-1} += dgh_t W_hh, then step t-1's gate backward on the finished rows;
            # the working copy of the upstream gradients also collects the direct path dh_t * z_t of every step
            work = ops.copy_flat(torch.empty_like(dhs), dhs)
            ops.gru_gates_bwd_into(work[T - 1], gates[T - 1], gh[T - 1], hs[T - 2], dgi[T - 1], dgh[T - 1], work[T - 2], accumulate=True)
            for t in range(T - 1, 0, -1):
                ops.gru_step_bwd(dgh[t], w_hh_t, work[t - 1], gates[t - 1], gh[t - 1] if t > 1 else gh0, hs[t - 2] if t > 1 else None, dgi[t - 1],
                                 dgh[t - 1], work[t - 2] if t > 1 else None, accumulate=True)
        else:
            carry = None
            for t in range(T - 1, -1, -1):
                dh = dhs[t] if carry is None else dhs[t] + carry
                _, _, dhp = ops.gru_gates_bwd(dh, gates[t], gh[t] if t else gh0, hs[t - 1] if t else None, dgi=dgi[t], dgh=dgh[t])
                if t:
                    carry = ops.gemm_nt(dgh[t], w_hh_t, out=dhp, accumulate=True)
        dgi2, dgh2 = dgi.view(T * B, 3 * H), dgh.view(T * B, 3 * H)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(dgi2, ops.transposed(w_ih)).view(T, B, -1)
        dw_ih = _wgrad(dgi2, x2, w_ih)
        if T > 1:
            dw_hh = _wgrad(dgh2[B:], hs.view(T * B, H)[:-B], w_hh)
        else:
            _touch(w_hh)
            dw_hh = None if _direct(w_hh) else torch.zeros_like(w_hh)
        return dx, dw_ih, dw_hh, _bgrad(dgi2, b_ih), _bgrad(dgh2, b_hh)


class GRUViewSeqFn(torch.autograd.Function):
    """The recurrent head over the n = 2T aggregator outputs of a contrastive step as ONE node (Full_layer.forward_view_sequence;
    train_MuRCL.py:243,272 with the reference's one shared hidden state): x [n*B, F] = the blocks x_00, x_01, x_10, x_11, ... of B rows;
    blocks 0 and 1 start from the zero state (``restart`` at patch step 0), block k >= 2 continues from block k-1 -> every hidden
    state [n*B, H] in one buffer.  The input projection, the input gradient and each weight / bias gradient are ONE launch over all
    n*B rows; no slice of x enters the graph (slicing x into a single step and a sequence cost two zero-fills, two copies and an
    add in the backward pass, and a concatenation of the two results in the forward pass).  Needs n >= 3 and the one-launch step
    kernels (``ops.gru_step_ok``)."""

    @staticmethod
    def forward(ctx, x, B, w_ih, w_hh, b_ih, b_hh):
        x2 = x.contiguous()
        R, H = x2.shape[0], w_hh.shape[1]
        n = R // B
        assert R == n * B and n >= 3 and ops.gru_step_ok(B, H)
        gi = ops.gemm_nt(x2, w_ih, epi=ops.EPI_BIAS, bias=b_ih)                      # [R, 3H]
        gh = torch.empty((R, 3 * H), dtype=torch.float32, device=x.device)          # (blocks 0 and 1 stay unwritten: they read the bias row)
        gates = torch.empty_like(gh)
        hs = torch.empty((R, H), dtype=torch.float32, device=x.device)
        whh, bhh = w_hh.detach(), b_hh.detach()
        ops.gru_gates_fwd(gi[:2 * B], bhh.view(1, -1), None, hnew=hs[:2 * B], gates=gates[:2 * B])     # both zero-state blocks: one launch
        for k in range(2, n):
            lo, hi = k * B, (k + 1) * B
            ops.gru_step_fwd(gi[lo:hi], hs[lo - B:lo], whh, bhh, hnew=hs[lo:hi], gates=gates[lo:hi], gh=gh[lo:hi])
        ctx.save_for_backward(x2, w_ih, w_hh, gates, gh, hs, b_ih, b_hh)
        ctx.B = B
        return hs

    @staticmethod
    def backward(ctx, dhs):
        x2, w_ih, w_hh, gates, gh, hs, b_ih, b_hh = ctx.saved_tensors
        B = ctx.B
        R, H = hs.shape
        n = R // B
        blk = lambda t, k: t[k * B:(k + 1) * B]                                       # noqa: E731
        dgi, dgh = torch.empty_like(gh), torch.empty_like(gh)
        w_hh_t = ops.transposed(w_hh)
        gh0 = b_hh.detach().view(1, -1)
        # back through time on a working copy of the upstream gradients (it collects dh_t * z_t and dgh_{t+1} W_hh of every step)
        work = ops.copy_flat(torch.empty((R, H), dtype=torch.float32, device=hs.device), dhs.contiguous())
        ops.gru_gates_bwd_into(blk(work, n - 1), blk(gates, n - 1), blk(gh, n - 1), blk(hs, n - 2), blk(dgi, n - 1), blk(dgh, n - 1),
                               blk(work, n - 2), accumulate=True)
        for k in range(n - 1, 1, -1):             # block k's dgh is complete: dh_{k-1} += dgh_k W_hh, then block k-1's gate backward
            first = k - 1 == 1                    # block 1 started from the zero state: bias row, no previous state to pass a gradient to
            ops.gru_step_bwd(blk(dgh, k), w_hh_t, blk(work, k - 1), blk(gates, k - 1), gh0 if first else blk(gh, k - 1),
                             None if first else blk(hs, k - 2), blk(dgi, k - 1), blk(dgh, k - 1), None if first else blk(work, k - 2),
                             accumulate=True)
        ops.gru_gates_bwd_into(blk(work, 0), blk(gates, 0), gh0, None, blk(dgi, 0), blk(dgh, 0))      # block 0: a single step beside the chain
        dx = ops.gemm_nt(dgi, ops.transposed(w_ih)) if ctx.needs_input_grad[0] else None
        dw_ih, db_ih = _wbgrad(dgi, x2, w_ih, b_ih)
        dw_hh = _wgrad(dgh[2 * B:], hs[B:R - B], w_hh)                                 # blocks k >= 2 against the state of block k-1
        return dx, None, dw_ih, dw_hh, db_ih, _bgrad(dgh, b_hh)


class NTXentSeqFn(torch.autograd.Function):
    """NT_Xent of T independent (view 0, view 1) batches at once: z [T,2B,P] -> (loss [T], cos(z_i, z_j) [T,B]); one launch
    computes all losses, gradients and cosines (the T patch steps of a pre-training step, train_MuRCL.py:249,277)."""

    @staticmethod
    def forward(ctx, z, temperature):
        """-> (loss [T], sim [T,B], mean of the T losses): the step loss of train_MuRCL.py:291 comes out of this node as its own
        launch (``ops.mean_small``), so that its gradient - 1/T on every step - is ONE scaling of the stored dz instead of a mean
        node's reduce, expand and multiply."""
        loss, dz, sim = ops.ntxent_batched(z, temperature, want_grad=True)
        ctx.save_for_backward(dz)
        ctx.mark_non_differentiable(sim)
        ctx.set_materialize_grads(False)
        return loss, sim, ops.mean_small(loss)

    @staticmethod
    def backward(ctx, dloss, _dsim, dmean):
        (dz,) = ctx.saved_tensors
        if dloss is None and dmean is None:
            return None, None
        T_ = dz.shape[0]
        if dloss is None and ops.is_unit_grad(dmean):
            return ops.axpby(dz, dz, 1.0 / T_, 0.0), None                                # (b = 0: y is not read)
        w = dloss if dmean is None else (dmean / T_).expand(T_) if dloss is None else dloss + dmean / T_
        return dz * w.reshape(-1, 1, 1), None


class NTXentFn(torch.autograd.Function):
    """NT_Xent.forward (utils/losses.py:24-41); gradient comes out of the same launch."""

    @staticmethod
    def forward(ctx, z_i, z_j, temperature, grad_lo, grad_hi):
        # z_j None: z_i already is the stacked [2B,P] batch (both views came out of one GEMM: no concatenation)
        z = z_i if z_j is None else torch.cat([z_i, z_j], 0)
        loss, dz, sim = ops.ntxent(z, temperature, want_grad=True, grad_lo=grad_lo, grad_hi=grad_hi)
        ctx.save_for_backward(dz)
        ctx.B = z.shape[0] // 2
        ctx.joint = z_j is None
        ctx.mark_non_differentiable(sim)
        ctx.set_materialize_grads(False)
        return loss[0], sim

    @staticmethod
    def backward(ctx, dloss, _dsim):
        (dz,) = ctx.saved_tensors
        if dloss is None:
            return None, None, None, None, None
        g = dz if ops.is_unit_grad(dloss) else dz * dloss
        if ctx.joint:
            return g, None, None, None, None
        return g[:ctx.B], g[ctx.B:], None, None, None


class DSMILFn(torch.autograd.Function):
    """MILNet.forward for a batch of equal-length bags (models/dsmil.py:9-16,64-81,104-113).

    The value projection is applied AFTER pooling: bag = (A^T X) Wv^T + bv, identical to A^T (X Wv^T + bv) because every column
    of the soft-max sums to one (dropout_v = 0).  The query projection is reassociated the same way (round 3): the attention
    logits Q[n] . q_c / sqrt(128) with Q = X Wq^T + bq equal X[n] . v_c + const for v_c = Wq^T q_c / sqrt(128), and the soft-max
    over n ignores the constant - so K6 is four streaming passes over X (instance scores; attention logits; pooling; and, going
    back, dA = X dZ^T with dWc, then R = dS^T X) and a handful of [B*C]-row GEMMs; the [B*N, 128] queries and the two GEMMs over
    all patches (forward and dWq) are never formed.  ``_DSMIL_REASSOC = False`` keeps the literal order of the reference (queries
    by one GEMM, 3-term bf16 split for f32).  Returns (classes [B,N,C], bag [B,C,d]).
    """
    QD = 128

    @staticmethod
    def forward(ctx, x, wc, bc, wq, bq, wv, bv, want_max=False, keep_v=None):
        """``keep_v`` (BClassifier(dropout_v > 0) in training mode, dsmil.py:53-59: ``v = Linear(Dropout(feats))``): None, or a keep
        multiplier for the VALUE branch's input - an ``ops.DropSeed`` (its mask is materialised here: a path no script takes) or a
        [B,N,d] tensor of 0 / 1/keep (parity tests).  The attention logits see the un-dropped features (``q = Linear(feats)``), so the
        pooled operand is X * keep while scores and logits use X: the explicit chain (soft-max, then a weighted row sum over the
        dropped copy) instead of the one-pass kernels.
        ``want_max``: also return cmax [B,C] = the max-instance class scores (train_RLMIL.py:516, ``torch.max(outputs_ins, 0)``) as a
        differentiable output - the arg-max launch has them in hand, and their gradient reaches the instance classifier through the
        B*C critical rows only (no dense [B,N,C] gradient, no ATen max / scatter / fill launches)."""
        B, N, d = x.shape
        T = x.dtype
        C = wc.shape[0]
        QD = DSMILFn.QD
        LD = QD + ((C + 7) // 8) * 8
        x2 = x.reshape(B * N, d)
        xv = x                                                                          # the value branch's input (dsmil.py:66)
        if keep_v is not None:
            km = ops.dropout_mask((B, N, d), T, keep_v.keep_p, x.device, seed=keep_v.seed) if isinstance(keep_v, ops.DropSeed) else \
                keep_v.to(T).reshape(B, N, d).contiguous()
            xv = ops.mul(x, km, out=torch.empty_like(x))
        cls = ops.rows_dot(x2.view(1, B * N, d), wc.view(1, C, d), bias=bc).view(B * N, C)    # instance scores (dsmil.py:9-16)
        m, cmax = ops.dsmil_argmax(cls, B, N, C, want_max=True)                         # critical instances (:71-73) and their scores
        reassoc = _DSMIL_REASSOC and C <= 4
        qv = reassoc and _DSMIL_QV and wq.shape[0] == QD and d % 4 == 0 and d <= 2048
        xm = None
        if qv:
            xm, qmax, v = ops.dsmil_qv(x2, m, wq, bq, B, N, C)                          # x_m, q_c = Wq x_m + bq, Wq^T q_c: one launch
        elif reassoc:
            xm = ops.cast(ops.gather_rows(x2, m, B, C, N, 0, d), torch.float32)         # [B*C, d]
            qmax = ops.gemm_nt(xm, wq, epi=ops.EPI_BIAS, bias=bq)                       # q_c = Wq x_m + bq     [B*C, 128]
            v = ops.gemm_nt(qmax, ops.transposed(wq))                                   # Wq^T q_c              [B*C, d]
        if reassoc:
            Y = v
            # attention + pooling from one pass over X: A = soft-max_n(X v_c / sqrt(128)) (:76-77), Z = A^T X (:78)
            one = ops.dsmil_attn_pool(x, v.view(B, C, d), 1.0 / math.sqrt(QD)) if (_DSMIL_ONEPASS and keep_v is None) else None
            if one is None:
                v *= 1.0 / math.sqrt(QD)
                A = ops.dsmil_softmax_(ops.rows_dot(x, v.view(B, C, d)))
        else:
            # queries: one 128-column GEMM; f32: as a 3-term bf16 split on the bf16 matrix pipe (ops.gemm_nt x3)
            Y = ops.gemm_nt(x2, wq if T == torch.float32 else ops.cast(wq, T), epi=ops.EPI_BIAS, bias=bq,
                            out_dtype=torch.float32, x3=_DSMIL_X3)                      # Q [B*N, 128]
            qmax = ops.gather_rows(Y, m, B, C, N, 0, QD)
            A = ops.dsmil_attn(Y, 0, qmax, B, N, C)
            one = None
        A, Z = one if one is not None else (A, ops.weighted_rowsum(xv, A))              # Z = A^T X  (:78; X * keep with dropout_v)
        bag = ops.gemm_nt(Z.view(B * C, d), wv, epi=ops.EPI_BIAS, bias=bv).view(B, C, d)
        classes = cls.view(B, N, C)
        ctx.save_for_backward(x, Y, m, qmax, A, Z, wv, wq, xm if qv else _placeholder(x), xv if keep_v is not None else _placeholder(x))
        ctx.meta = (B, N, d, C, LD, reassoc, qv, keep_v is not None)
        ctx.params = (wc, bc, wq, bq, wv, bv)             # (the parameters themselves: the backward pass adds into their gradient buffers)
        ctx.mark_non_differentiable(m)
        ctx.set_materialize_grads(False)
        if not want_max:
            ctx.mark_non_differentiable(cmax)
        return classes, bag, m, cmax

    @staticmethod
    def backward(ctx, dclasses, dbag, _dm, dcmax=None):
        x, Y, m, qmax, A, Z, wv, wq, xm_saved, xv = ctx.saved_tensors
        B, N, d, C, LD, reassoc, qv, dropped = ctx.meta
        if not dropped:
            xv = x                                                                          # the pooled operand (X * keep under dropout_v)
        T, QD = x.dtype, DSMILFn.QD
        dev = x.device
        x2 = x.reshape(B * N, d)
        dbag2 = (dbag if dbag is not None else torch.zeros((B, C, d), device=dev)).reshape(B * C, d).contiguous()
        dwv, dbv = ops.gemm_tn_with_colsum(dbag2, Z.view(B * C, d))           # (dWv, dbv) of the value projection: one launch
        dZ = ops.gemm_nt(dbag2, ops.transposed(wv)).view(B, C, d)
        xm = None if qv else ops.gather_rows(x2, m, B, C, N, 0, d)                          # critical instances
        dcls = dclasses.reshape(B, N, C).float().contiguous() if dclasses is not None else None
        # reassociated: ONE pass over X gives R (below) and dWc - neither dA nor dS is stored
        one = ops.dsmil_attn_pool_bwd(x, dZ, A, Z, dcls, 1.0 / math.sqrt(QD)) if (reassoc and _DSMIL_ONEPASS and not dropped) else None
        # otherwise dA = X dZ^T and, when the instance scores carry a gradient, dWc = dcls^T X from the SAME pass over X
        fused = None
        if one is None:
            if dclasses is not None and not dropped:
                fused = ops.rows_dot_wsum(x, dZ, dcls)
            dA = fused[0] if fused is not None else ops.rows_dot(xv, dZ)                      # dA = (X * keep) dZ^T
        dwc = dbc = None
        cmax_done = False
        if one is not None and qv and dcmax is not None:
            # ... and the max-instance term's share of the instance classifier's gradient from the same second launch
            if dclasses is not None:
                dwc, dbc = one[1], dcls.view(B * N, C).sum(0)
            else:
                dwc, dbc = torch.empty((C, d), dtype=torch.float32, device=dev), torch.empty((C,), dtype=torch.float32, device=dev)
            dwq, dbq = ops.dsmil_qv_bwd(one[0].view(B * C, d), qmax, xm_saved, wq, dcmax=dcmax.float(), dwc=dwc, dbc=dbc,
                                        accumulate=dclasses is not None)
            cmax_done = True
        elif one is not None and qv:
            dwq, dbq = ops.dsmil_qv_bwd(one[0].view(B * C, d), qmax, xm_saved, wq)         # dq = R Wq^T, dWq = q^T R + dq^T x_m, dbq
        elif one is not None:
            R = one[0].view(B * C, d)
            dqmax = ops.gemm_nt(R, wq)                                                      # [B*C, 128]
            dwq = ops.gemm_tn(qmax, R)                                                      # [128, d]
            dbq = None
        elif reassoc:
            # dS weights the rows of X once more: R_c = sum_n dS[n,c] X[n] / sqrt(128) is the gradient of v_c, and
            #   sum_n dQ[n]^T X[n] = qmax^T R,   dqmax = sum_n dS[n,c] Q[n] / sqrt(128) = R Wq^T  (+ bq sum_n dS[n,c], and a soft-max
            #   gradient sums to nothing)
            R = ops.weighted_rowsum(x, ops.dsmil_softmax_bwd(A, dA)).view(B * C, d)
            R *= 1.0 / math.sqrt(QD)
            dqmax = ops.gemm_nt(R, wq)                                                      # [B*C, 128]
            dwq = ops.gemm_tn(qmax, R)                                                      # [128, d]
            dbq = None
        else:
            dQ = torch.empty((B * N, QD), dtype=torch.float32, device=dev)                  # written in full below
            dqmax = ops.dsmil_attn_bwd(A, dA, Y, 0, qmax, dQ, B, N, C)
            dwq = ops.gemm_tn(dQ if T == torch.float32 else ops.cast(dQ, T), x2, x3=_DSMIL_X3)  # [128, d]: one tile row
            dbq = ops.colsum(dQ)
        if one is not None and qv:
            pass
        elif reassoc:
            ops.gemm_tn(dqmax, xm_saved if qv else ops.cast(xm, torch.float32), out=dwq)
            dbq = ops.colsum(dqmax) if dbq is None else ops.colsum(dqmax, out=dbq, accumulate=True)
        else:
            ops.gemm_tn(dqmax if T == torch.float32 else ops.cast(dqmax, T), xm, out=dwq)
            dbq = ops.colsum(dqmax) if dbq is None else ops.colsum(dqmax, out=dbq, accumulate=True)
        if dclasses is not None and not cmax_done:
            # the C instance-score columns: dWc = dcls^T X as a weighted row sum over all patches (a 128-wide wgrad tile
            # for 2 columns would read X a second time through the GEMM path)
            if one is not None:
                dwc = one[1]
            else:
                dwc = fused[1] if fused is not None else ops.weighted_rowsum(x2.view(1, B * N, d), dcls.view(1, B * N, C)).view(C, d)
            dbc = dcls.view(B * N, C).sum(0)
        if dcmax is not None and not cmax_done:
            # paths without the two-launch [B*C]-row algebra: the same sums in plain tensor ops on the B*C critical rows
            xm_f = (xm_saved if qv else ops.cast(ops.gather_rows(x2, m, B, C, N, 0, d), torch.float32)).view(B, C, d)
            g = dcmax.float().view(B, C, 1)
            dwc_m, dbc_m = (g * xm_f).sum(0), g.view(B, C).sum(0)
            dwc = dwc_m if dwc is None else dwc + dwc_m
            dbc = dbc_m if dbc is None else dbc + dbc_m
        # six parameter gradients: ONE launch adds them to the optimizer's pre-seated buffers (no AccumulateGrad add per parameter)
        dwc, dbc, dwq, dbq, dwv, dbv = _pgrads(*zip((dwc, dbc, dwq, dbq, dwv, dbv), ctx.params))
        return None, dwc, dbc, dwq, dbq, dwv, dbv, None, None


_INST_CONST = {}
_ZERO_CONST = {}


def _zeros_const(dev, n):
    """A shared read-only zero vector (the instance loss of a call without instance evaluation): no fill launch per call."""
    z = _ZERO_CONST.get((dev, n))
    if z is None:
        z = _ZERO_CONST[(dev, n)] = torch.zeros((n,), dtype=torch.float32, device=dev)
    return z


def _placeholder(like):
    """A shared one-element tensor that stands in for an absent saved tensor (``save_for_backward`` takes tensors): no fill launch per call."""
    key = (like.device, like.dtype, "placeholder")
    z = _ZERO_CONST.get(key)
    if z is None:
        z = _ZERO_CONST[key] = torch.zeros((1,), dtype=like.dtype, device=like.device)
    return z


def _inst_constants(dev, B, N, k, n_cls, subtyping):
    """Index / target constants of CLAM's instance branch, built once per shape (they were five tiny launches and three
    host->device copies per call): bag row offsets [B,1], class ids [1,n_cls], in-class targets [1,1,2k] = [1]*k + [0]*k
    (clam.py:105-119), out-of-class targets [0]*k (subtyping, clam.py:122-132) or none, the rest ignored (-1)."""
    key = (dev, B, N, k, n_cls, bool(subtyping))
    c = _INST_CONST.get(key)
    if c is None:
        if len(_INST_CONST) > 32:
            _INST_CONST.clear()
        base = (torch.arange(B, device=dev, dtype=torch.int64) * N).unsqueeze(1)
        cls_ids = torch.arange(n_cls, device=dev).view(1, n_cls)
        t_in = torch.tensor([1] * k + [0] * k, dtype=torch.int64, device=dev).view(1, 1, -1)
        t_out = torch.tensor(([0] * k if subtyping else [-1] * k) + [-1] * k, dtype=torch.int64, device=dev).view(1, 1, -1)
        c = _INST_CONST[key] = (base, cls_ids, t_in, t_out)
    return c


class StackParamsFn(torch.autograd.Function):
    """``torch.stack`` of the n instance classifiers' weights and of their biases (clam.py:103-132 reads them per class) -> ([n,R,L],
    [n,R]) as one launch - none between optimizer steps (``ops.stacked_views``) - instead of two ATen concatenations per forward;
    the backward hands every parameter its slice of the stacked gradient (no launch)."""

    @staticmethod
    def forward(ctx, n, *params):
        ws, bs = params[:n], params[n:]
        W, b = ops.stacked_views(ws, bs)
        ctx.n = n
        return W.view(n, *ws[0].shape), b.view(n, *bs[0].shape)

    @staticmethod
    def backward(ctx, dW, db):
        n = ctx.n
        gw = [None] * n if dW is None else [dW[i] for i in range(n)]
        gb = [None] * n if db is None else [db[i] for i in range(n)]
        return (None, *gw, *gb)


class CLAMFn(torch.autograd.Function):
    """CLAM_SB.bag_forward for a batch of equal-length bags, optionally with the instance-level loss
    (models/clam.py:134-181,103-132).

    x [B,N,d] in the compute dtype; parameters f32.  ``keeps`` = None (eval) or the three dropout keep-multiplier
    tensors (values 0 or 1/0.75) for h, the tanh branch and the sigmoid branch (clam.py:71-72,47-48).
    ``inst`` = None or (W [n_cls,2,512], b [n_cls,2], labels: list[int], k_sample, subtyping).
    Returns (M [B,512], A [B,N], raw scores [B,N], inst_loss [B], ids [B,2k], preds/targets [2,B,n_cls,2k]); only M and
    inst_loss carry grad.
    """

    @staticmethod
    def forward(ctx, x, w1, b1, wa, ba, wb, bb, wc, bc, inst_w, inst_b, keeps, inst_cfg, grad_on=True):
        B, N, d = x.shape
        T = x.dtype
        f32 = T == torch.float32
        c = (lambda w: w) if f32 else (lambda w: ops.cast(w, T))
        x2 = x.reshape(B * N, d)
        L, D = w1.shape[0], wa.shape[0]
        # bf16 gated chain: the compute-dtype copy of fc, the two gate Linears interleaved for the panel kernel and their transpose
        # come out of ONE launch (none between optimizer steps) instead of a dozen cat / gather / cast launches
        views = ops.clam_views(w1, wa, ba, wb, bb, wc, T) if (T == torch.bfloat16 and wb is not None and D % 16 == 0) else None
        if views is not None:
            c = lambda w: views[0] if w is w1 else ops.cast(w, T)          # noqa: E731
        m1 = None
        k1 = ka = kb = None
        if keeps is not None:
            k1, ka, kb = keeps
        seeded = isinstance(k1, ops.DropSeed)
        fc_drop = False
        if T == torch.bfloat16 and d == 512 and ops.panel_supported(B * N, L, 512, ops.PG_BIAS_RELU):
            # weight-stationary panel kernel; its 1-bit ReLU mask also serves the backward pass.  Seeded Dropout(0.25) behind the
            # ReLU (clam.py:69-72) happens in the same epilogue: the mask is never materialised and the bits record what survives
            fc_drop = _FUSED_FC_DROP and seeded
            h, m1, _ = ops.panel_gemm(x2, c(w1), ops.PG_BIAS_RELU, bias=b1, want_bitmask=fc_drop or (keeps is None and bool(grad_on)),
                                      drop=k1 if fc_drop else None)
        else:
            h = ops.gemm_nt(x2, c(w1), epi=ops.EPI_BIAS_RELU, bias=b1)                # clam.py:69
        if keeps is not None and not fc_drop:
            if seeded:                                                                 # Dropout(0.25) after ReLU (clam.py:69-72)
                if (B * N) % 32 == 0 and L % 128 == 0:
                    # the mask is generated inside the pass that applies it, which also leaves the 1-bit mask of the
                    # surviving positive entries for the backward pass (bf16 panel dgrad)
                    m1 = ops.dropout_relu_bitmask(h, k1, want_bits=(T == torch.bfloat16))
                else:
                    ops.mul(h, ops.dropout_mask(h.shape, T, k1.keep_p, h.device, seed=k1.seed))
            else:
                ops.mul(h, k1)                                                         # an injected keep mask (parity tests)
        gated = wb is not None                   # False: the plain Attn_Net (CLAM_SB(gate=False), clam.py:18-34,80-81)
        GW = 2 * D if gated else D               # gate columns
        # bf16 with 512-wide h and gates: the weight-stationary panel kernel (same GEMM, half the time of the tile kernel)
        panel = (T == torch.bfloat16 and L == 512 and ops.panel_supported(B * N, GW, 512, ops.PG_BIAS))
        # (``grad_on`` = the caller's torch.is_grad_enabled(): see ABMILFn.forward)
        fused_gate = (_FUSED_GATE and panel and gated and keeps is None and GW == 512 and not (grad_on and any(ctx.needs_input_grad))
                      and ops.panel_supported(B * N, GW, 512, ops.PG_GATE))
        # calls a backward pass may follow: the same epilogue also leaves the pre-activations (interleaved column order) for it,
        # and the separate score pass over U (97-120 us at C3) disappears; the gate Dropouts are applied inside from their seeds
        gate_u = (_GATE_U and not fused_gate and panel and gated and GW == 512 and (keeps is None or seeded)
                  and ops.panel_supported(B * N, GW, 512, ops.PG_GATE_U) and ops.gated_bwd_il_supported(B * N, D, L, N)
                  and ops.panel_supported(B * N, L, 512, ops.PG_RANK1_MASK, N))
        if fused_gate:
            # forward-only calls (validation, heat-map scoring, the frozen aggregator of stage 2): the score comes out of the gate
            # GEMM's epilogue - tanh(a_d) sigmoid(b_d) c_d summed per wave - and the [B*N, 2D] pre-activations are never written
            U = None
            s_parts = ops.panel_gate_score(h, views[1], views[3], views[4], bc, parts=True)
        elif gate_u:
            U, s_parts = ops.panel_gate_u(h, views[1], views[3], views[4], bc, ka, kb, parts=True)
        else:
            wab = torch.cat([wa, wb], 0) if gated else wa
            bab = torch.cat([ba, bb], 0) if gated else ba
            if panel:
                U, _, _ = ops.panel_gemm(h, c(wab), ops.PG_BIAS, bias=bab)              # both gate branches, one pass
            else:
                U = ops.gemm_nt(h, c(wab), epi=ops.EPI_BIAS, bias=bab)
        if not fused_gate and not gate_u:
            s = ops.gated_score_fwd(U, wc.reshape(-1).contiguous(), bc, ka, kb, gated=gated).view(B, N)
        if fused_gate or gate_u:
            # the epilogue's partial rows summed on the way, and the pooled rows cleared for the pass below: one launch
            M = torch.empty((B, L), dtype=torch.float32, device=x.device)
            s, A = ops.softmax_rows_parts(s_parts, B, N, zero=M)
            ops.weighted_rowsum(h.view(B, N, L), A.view(B, N, 1), into=M)              # clam.py:170
        else:
            A = ops.softmax_rows(s)                                                    # clam.py:144
            M = ops.weighted_rowsum(h.view(B, N, L), A.view(B, N, 1)).view(B, L)       # clam.py:170
        dev = x.device
        inst_loss = _zeros_const(dev, B) if inst_cfg is None else None          # (both instance branches below write their own)
        saved_inst = None
        ids = None
        inst_pt = None
        if inst_cfg is not None:
            # instance-level evaluation for ALL (bag, class) pairs at once (clam.py:103-132,150-168): the k top and k
            # bottom patches of a bag are the same rows for every class, so one gather, one stacked classifier GEMM and
            # one grouped cross-entropy launch replace the per-class / per-bag loops; pairs differ only in their targets:
            #   class == label      -> 2k rows, targets [1]*k + [0]*k            (inst_eval,     clam.py:105-119)
            #   class != label      -> k top rows with target 0 if subtyping    (inst_eval_out, clam.py:122-132), else none
            labels, k, subtyping = inst_cfg[:3]
            custom_loss = inst_cfg[3] if len(inst_cfg) > 3 else None          # a caller-supplied instance_loss_fn (clam.py:64-65,118,131)
            n_cls = inst_w.shape[0]
            ids = ops.topk_ids(A, k)                                                   # [B, 2k]
            lab = labels.to(device=dev, dtype=torch.int64) if isinstance(labels, torch.Tensor) else \
                torch.as_tensor([int(v) for v in labels], dtype=torch.int64).to(dev)
            w_st = inst_w.reshape(n_cls * 2, -1).contiguous()
            if custom_loss is None and _FUSED_INST and 2 * n_cls <= 16 and k <= 32 and L % 8 == 0:
                # one launch: gather the 2k rows, all 2 n_cls instance logits, the cross-entropies and their gradients
                inst_loss, dl, inst_pt = ops.clam_inst_fwd(h, ids, lab, w_st, inst_b.reshape(-1), B, N, k, n_cls, subtyping)
                saved_inst = ("fused", ids, dl, w_st, k, n_cls)
            else:
                base, cls_ids, t_in, t_out = _inst_constants(dev, B, N, k, n_cls, subtyping)
                rows_all = (base + ids.to(torch.int64)).reshape(-1)                        # [B*2k] rows of h
                feats = ops.take_rows(h, rows_all)                                         # [B*2k, L] f32
                logits = ops.gemm_nt(feats, w_st, epi=ops.EPI_BIAS, bias=inst_b.reshape(-1).contiguous())   # [B*2k, 2 n_cls]
                logits_g = logits.view(B, 2 * k, n_cls, 2).permute(0, 2, 1, 3).contiguous()                 # [B, n_cls, 2k, 2]
                same = lab.view(B, 1) == cls_ids                                                            # [B, n_cls]
                targets = torch.where(same.unsqueeze(2), t_in, t_out).contiguous()                          # [B, n_cls, 2k]
                loss_g, dl_g, preds_g = ops.cross_entropy(logits_g.view(-1, 2), targets.view(-1), 2 * k)
                if custom_loss is not None:
                    # The reference hands (logits [rows,2], targets [rows]) of every evaluated (bag, class) pair to whatever loss it was
                    # constructed with (clam.py:118,131).  The gather, the classifier product and the predictions above are the HIP
                    # kernels; the caller's loss runs on the pair's few logits as given (rows with target -1 do not exist for that pair)
                    # and its gradient w.r.t. them - taken here with autograd on that [rows,2] leaf - takes the place of the
                    # cross-entropy gradient in the backward pass below.  A loss with parameters of its own gets no gradient for them.
                    with torch.enable_grad():
                        leaf = logits_g.detach().requires_grad_()
                        tg = targets.view(B, n_cls, 2 * k)
                        pair = []
                        for b_ in range(B):
                            for c_ in range(n_cls):
                                keep_rows = tg[b_, c_] >= 0
                                if bool(keep_rows.any()):
                                    pair.append(custom_loss(leaf[b_, c_][keep_rows], tg[b_, c_][keep_rows]))
                                else:
                                    pair.append(leaf.new_zeros(()))
                        loss_pairs = torch.stack(pair).view(B, n_cls)
                        dl_g, = torch.autograd.grad(loss_pairs.sum(), leaf, allow_unused=True)
                    loss_g = loss_pairs.detach().reshape(-1)
                    dl_g = (torch.zeros_like(logits_g) if dl_g is None else dl_g).reshape(-1, 2).contiguous()
                scale = 1.0 / n_cls if subtyping else 1.0                                  # clam.py:167-168
                inst_loss = loss_g.view(B, n_cls).sum(1) * scale
                inst_pt = torch.stack([preds_g.view(B, n_cls, 2 * k), targets], 0)         # -1 where a pair has no such row
                saved_inst = (rows_all, feats, dl_g, scale, k, n_cls)
        ctx.save_for_backward(x2, h, U if U is not None else _placeholder(x2), A, M, w1, wa, wb, wc,
                              inst_w if inst_w is not None else _placeholder(x2), m1)
        ctx.gated, ctx.gate_u = gated, gate_u
        ctx.bias_params = (b1, ba, bb, bc)               # (the parameters themselves: the backward pass adds into their gradient buffers)
        ctx.wab_t = views[2] if gate_u else None         # (a cached view: parameters do not change between a forward and its backward)
        ctx.keeps, ctx.saved_inst, ctx.dims = keeps, saved_inst, (B, N, d, L, D)
        if ids is None:
            ids = torch.zeros((B, 0), dtype=torch.int32, device=dev)
        if inst_pt is None:
            inst_pt = torch.zeros((2, B, 0, 0), dtype=torch.int64, device=dev)
        if inst_cfg is None:
            # the cached zero is shared by every call without an instance branch: as a differentiable output autograd would
            # re-point its grad_fn at this node (keeping the node's saved activations alive through the cache) - hand out a view,
            # marked non-differentiable
            inst_loss = inst_loss.detach()[:]
            ctx.mark_non_differentiable(A, s, ids, inst_pt, inst_loss)
        else:
            ctx.mark_non_differentiable(A, s, ids, inst_pt)
        ctx.set_materialize_grads(False)
        return M, A, s, inst_loss, ids, inst_pt

    @staticmethod
    def backward(ctx, dM, _dA, _ds, dinst, _dids, _dpt):
        x2, h, U, A, M, w1, wa, wb, wc, inst_w, m1 = ctx.saved_tensors
        b1, ba, bb, bc = ctx.bias_params
        B, N, d, L, D = ctx.dims
        T = x2.dtype
        f32 = T == torch.float32
        c = (lambda t: t) if f32 else (lambda t: ops.cast(t, T))
        k1 = ka = kb = None
        if ctx.keeps is not None:
            k1, ka, kb = ctx.keeps
        dM = dM.contiguous() if dM is not None else torch.zeros_like(M)
        gated = ctx.gated
        if ctx.gate_u:
            # pooling + soft-max + gate backward in ONE pass over (h, U): sum_m A_m (h_m . dM) = M . dM, so
            # ds_n = A_n (h_n . dM - M . dM) needs no reduction over the bag; U / dU in the interleaved column order of the forward
            dU, dwc, dbc, dbab = ops.gated_score_bwd_il(U, wc.reshape(-1).contiguous(), ka, kb, h=h, dM=dM, Mp=M, A=A.view(-1),
                                                        rows_per_bag=N)
            # gate + first-layer weight gradients (clam.py:69-72) wait for dz1 and share one round of workgroups; the reduce launch
            # also undoes the 16-row interleave, applies the Dropout factor and sums the bias-gradient rows (no ATen launches)
            grouped = _GROUP_WGRAD and ops.gemm_tn_grouped_ok([(dU, h, None, None, None), (h, x2, None, None, None)])
            dwab = None
            if not grouped:
                dwab = ops.gemm_tn(dU, h).view(D // 16, 2, 16, L).permute(1, 0, 2, 3).reshape(2 * D, L)   # rows back in [Wa; Wb] order
            wab_t = ctx.wab_t                                                         # [L, 2D] interleaved columns, as dU's
        else:
            # pooling: dA[n] = h[n].dM ; soft-max backward ; gate backward
            dA = ops.rows_dot(h.view(B, N, L), dM.view(B, 1, L)).view(B, N)
            ds = ops.softmax_rows_bwd(A, dA).view(-1)
            dU, dwc, dbc, dbab = ops.gated_score_bwd(U, wc.reshape(-1).contiguous(), ds, ka, kb, gated=gated)   # dbab: column sums, same pass
            dwab = ops.gemm_tn(dU, h)                                                     # [2D, L] (gated) / [D, L]
            wab_t = None
            grouped = False
        wab = None if wab_t is not None else (torch.cat([wa, wb], 0) if gated else wa)
        # dZ1 = (dU [Wa;Wb] + A (x) dM) * relu'(h)   (h here is already the dropped h: zero where dropped)
        if (T == torch.bfloat16 and dU.shape[1] == 512 and ops.panel_supported(B * N, L, 512, ops.PG_RANK1_MASK, N)):
            # column sums (the bias gradient) come out of the same launch; the instance branch below extends them by the
            # few rows it adds
            dz1, _, db1 = ops.panel_gemm(dU, wab_t if wab_t is not None else ops.transpose_cast(wab, T), ops.PG_RANK1_MASK,
                                         bitmask=m1 if m1 is not None else ops.relu_bitmask(h),
                                         rowscale=A.view(-1), rank1=dM, rows_per_bag=N, colsum=True, colsum_defer=grouped)
            db1_parts = db1 if grouped else None
        else:
            db1 = db1_parts = None
            assert not grouped
            dz1 = ops.gemm_nt(dU, ops.transpose_cast(wab, T), epi=ops.EPI_RANK1_MASK, mask=h, rowscale=A.view(-1),
                              rank1=dM, rows_per_bag=N)
        # instance branch: classifier grads + sparse feature grads added under the same ReLU mask
        dinst_w = dinst_b = db1_extra = None
        if ctx.saved_inst is not None and dinst is not None and ctx.saved_inst[0] == "fused":
            _, ids, dl, w_st, k, n_cls = ctx.saved_inst
            dwi, dbi, gsum = ops.clam_inst_bwd(h, ids, w_st, dl, dinst.float(), B, N, k, n_cls, dz1)
            dinst_w, dinst_b = dwi.view(n_cls, 2, -1), dbi.view(n_cls, 2)
            if grouped:
                db1_extra = gsum
            elif db1 is not None:
                db1 = db1 + gsum
        elif ctx.saved_inst is not None and dinst is not None:
            rows_all, feats, dl_g, scale, k, n_cls = ctx.saved_inst
            up = (dinst * scale).view(B, 1, 1, 1)                                      # upstream weight per (bag, ...)
            dlog = (dl_g.view(B, n_cls, 2 * k, 2) * up).permute(0, 2, 1, 3).reshape(B * 2 * k, 2 * n_cls).contiguous()
            dinst_w = ops.gemm_tn(dlog, feats).view(n_cls, 2, -1)                      # (gemm_tn pads narrow N1 itself)
            dinst_b = ops.colsum(dlog).view(n_cls, 2)
            g = ops.gemm_nt(dlog, inst_w.reshape(n_cls * 2, -1).t().contiguous())      # [B*2k, L]; K = 2 n_cls is padded
            ops.scatter_add_rows_masked(dz1, h, rows_all, g, write_back=db1 is not None)
            if grouped:
                db1_extra = ops.colsum(g)
            elif db1 is not None:
                ops.colsum(g, out=db1, accumulate=True)                                  # [B*2k, L] f32: the rows just added
        kp = None
        if k1 is not None:                       # the surviving entries of the keep mask all equal 1/0.75
            kp = k1.keep_q if isinstance(k1, ops.DropSeed) else 0.75
        if grouped:
            db1 = torch.empty((L,), dtype=torch.float32, device=dz1.device)
            sc = None if kp is None else {"scale": 1.0 / kp}
            dwab, dw1 = ops.gemm_tn_grouped([(dU, h, None, None, None, {"deinterleave": True}),
                                             (dz1, x2, None, db1, db1_parts, dict(sc or {}, overwrite=True))], fresh=True)
            if db1_extra is not None:            # the rows the instance branch added to dz1 after its column sums were taken
                db1 = db1 + (db1_extra if kp is None else db1_extra / kp)
        else:
            dw1 = ops.gemm_tn(dz1, x2)
            if db1 is None:
                db1 = ops.colsum(dz1)
            if kp is not None:
                dw1, db1 = dw1 / kp, db1 / kp
        # the eight (six without the gate) parameter gradients: added to the optimizer's pre-seated buffers by ONE launch (autograd's
        # AccumulateGrad would run one ATen add per parameter); parameters without such a buffer get their tensors back
        if not gated:
            dw1, db1, dwab, dbab_, dwc_, dbc = _pgrads((dw1, w1), (db1, b1), (dwab, wa), (dbab.contiguous(), ba), (dwc.view(1, -1), wc), (dbc, bc))
            return (None, dw1, db1, dwab, dbab_, None, None, dwc_, dbc, dinst_w, dinst_b, None, None, None)
        dw1, db1, dwa_, dba_, dwb_, dbb_, dwc_, dbc = _pgrads((dw1, w1), (db1, b1), (dwab[:D].contiguous(), wa), (dbab[:D].contiguous(), ba),
                                                              (dwab[D:].contiguous(), wb), (dbab[D:].contiguous(), bb), (dwc.view(1, -1), wc), (dbc, bc))
        return (None, dw1, db1, dwa_, dba_, dwb_, dbb_, dwc_, dbc, dinst_w, dinst_b, None, None, None)


class PolicyHeadFn(torch.autograd.Function):
    """log-prob of given actions under N(sigmoid(z), diag(std)) (ActorCritic.evaluate, rlmil.py:115-121)."""

    @staticmethod
    def forward(ctx, z, actions, std):
        mu, act, logp = ops.policy_head_fwd(z, std, actions=actions)
        ctx.save_for_backward(mu, act)
        ctx.std = std
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        mu, act = ctx.saved_tensors
        return ops.policy_head_bwd(mu, act, dlogp, ctx.std), None, None


class PPOLossFn(torch.autograd.Function):
    """mean(-min(surr1, surr2) + 0.5*MSE - 0.01*entropy) (PPO.update, rlmil.py:172-181)."""

    @staticmethod
    def forward(ctx, logp, old_logp, value, ret, eps_clip, entropy, n_total=None):
        loss, dlogp, dvalue = ops.ppo_loss(logp, old_logp, value, ret, eps_clip, entropy, n_total)
        ctx.save_for_backward(dlogp, dvalue)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dlogp, dvalue = ctx.saved_tensors
        return dlogp * g, None, dvalue * g, None, None, None, None


class GroupedCrossEntropyFn(torch.autograd.Function):
    """Mean CE of every group of ``group`` consecutive rows of [R,C] logits -> [R/group] losses, one launch: the T per-patch-step
    ``nn.CrossEntropyLoss()`` values of a step whose T x B head rows were computed together (train_RLMIL.py:316,502,709).  With
    ``want_conf`` -> (losses, conf [R]): the soft-max confidence of each row's true class from the same launch (the RL-MIL rewards
    are its differences between patch steps, train_RLMIL.py:345,369-371); conf is not differentiable."""

    @staticmethod
    def forward(ctx, logits, targets, group, want_conf=False):
        res = ops.cross_entropy(logits.float().contiguous(), targets.to(torch.int64).contiguous(), int(group), want_conf=bool(want_conf))
        ctx.save_for_backward(res[1])
        ctx.group = int(group)
        if want_conf:
            ctx.mark_non_differentiable(res[3])
            return res[0], res[3]
        return res[0]

    @staticmethod
    def backward(ctx, g, _gconf=None):
        (dl,) = ctx.saved_tensors
        return dl * g.reshape(-1, 1).repeat_interleave(ctx.group, 0), None, None, None


class StepCEMeanFn(torch.autograd.Function):
    """The loss of a supervised RL-MIL step whose T x B head rows were computed together (train_RLMIL.py:316,502,709 per patch step,
    their mean over the T steps as the step loss): logits [T*B, C], targets [T*B] -> (mean_t loss_t, loss_t [T], conf [T*B]) with
    loss_t = nn.CrossEntropyLoss() of step t's B rows and conf the soft-max confidence of each row's true class (the rewards are its
    differences between steps, :345,369-371).  One cross-entropy launch + one mean launch forward, ONE scaling of the stored
    d(loss_t)/d(logits) backward; loss_t and conf are not differentiable (a sum / division node over loss_t, a repeat_interleave and a
    multiply in the backward pass were ~10 ATen launches per step)."""

    @staticmethod
    def forward(ctx, logits, targets, group):
        loss_t, dl, _, conf = ops.cross_entropy(logits.float().contiguous(), targets.to(torch.int64).contiguous(), int(group), want_conf=True)
        ctx.save_for_backward(dl)
        ctx.T = loss_t.numel()
        ctx.mark_non_differentiable(loss_t, conf)
        ctx.set_materialize_grads(False)
        return ops.mean_small(loss_t), loss_t, conf

    @staticmethod
    def backward(ctx, g, _gl=None, _gc=None):
        (dl,) = ctx.saved_tensors
        if g is None:
            return None, None, None
        if ops.is_unit_grad(g):
            return ops.axpby(dl, dl, 1.0 / ctx.T, 0.0), None, None
        return dl * (g / ctx.T), None, None


class StepLossFn(torch.autograd.Function):
    """``StepCEMeanFn`` for the two architectures whose step loss mixes a second term into the head's cross-entropy
    (train_RLMIL.py:336 CLAM-SB: ``bag_weight * ce + (1 - bag_weight) * instance_loss``, the instance loss averaged over the B bags of
    the step; :527-529 DSMIL: ``0.5 * ce + 0.5 * ce(max-instance scores)``): logits [T*B, C], targets [T*B], ``extra`` = the per-bag
    instance losses [T*B] (``extra_is_logits`` False) or the second logits [T*B, C'] -> (mean_t loss_t, loss_t [T], conf [T*B]).
    Forward: the cross-entropy launch(es), a grouped mean, one mixing launch, one mean launch; backward: one scaling per input (the
    instance-loss gradient is a constant).  ``loss_t`` and ``conf`` are not differentiable."""

    @staticmethod
    def forward(ctx, logits, targets, group, w_ce, extra, w_x, extra_is_logits):
        tg = targets.to(torch.int64).contiguous()
        ce_t, dl, _, conf = ops.cross_entropy(logits.float().contiguous(), tg, int(group), want_conf=True)
        T_ = ce_t.numel()
        if extra_is_logits:
            x_t, dlx, _ = ops.cross_entropy(extra.float().contiguous(), tg, int(group))
            ctx.save_for_backward(dl, dlx)
        else:
            x_t = ops.group_mean(extra.float().contiguous().view(-1), T_, int(group))
            ctx.save_for_backward(dl)
        loss_t = ops.axpby(ce_t, x_t, float(w_ce), float(w_x))
        ctx.meta = (T_, int(group), float(w_ce), float(w_x), bool(extra_is_logits), tuple(extra.shape))
        ctx.mark_non_differentiable(loss_t, conf)
        ctx.set_materialize_grads(False)
        return ops.mean_small(loss_t), loss_t, conf

    @staticmethod
    def backward(ctx, g, _gl=None, _gc=None):
        if g is None:
            return (None,) * 7
        T_, group, w_ce, w_x, is_logits, xshape = ctx.meta
        dl = ctx.saved_tensors[0]
        if ops.is_unit_grad(g):
            dlogits = ops.axpby(dl, dl, w_ce / T_, 0.0)
            if is_logits:
                dlx = ctx.saved_tensors[1]
                dextra = ops.axpby(dlx, dlx, w_x / T_, 0.0)
            else:
                dextra = ops.filled(xshape, w_x / (T_ * group), dl.device)
        else:
            dlogits = dl * (g * (w_ce / T_))
            dextra = ctx.saved_tensors[1] * (g * (w_x / T_)) if is_logits else (g * (w_x / (T_ * group))).expand(xshape).contiguous()
        return dlogits, None, None, None, dextra, None, None


class CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss() (mean) over [R,C] logits (train_RLMIL.py:316,502,709)."""

    @staticmethod
    def forward(ctx, logits, targets):
        loss, dl, preds = ops.cross_entropy(logits.float().contiguous(), targets.to(torch.int64).contiguous(), logits.shape[0])
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None
