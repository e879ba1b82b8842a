"""Tensor-level wrappers over the C-ABI (include/murcl_amd.h).  No autograd here.

Every function launches hand-written HIP kernels on ``torch.cuda.current_stream()``;
torch is used only to allocate outputs/workspaces.  Inputs must be CUDA (HIP) tensors:
there is no CPU path.
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import BF16, F32, EPI_BIAS, EPI_BIAS_RELU, EPI_MASK, EPI_NONE, EPI_RANK1_MASK, check, dt, ptr, stream

_TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16}
_EPI_NAME = {EPI_NONE: "NONE", EPI_BIAS: "BIAS", EPI_BIAS_RELU: "BIAS_RELU", EPI_MASK: "MASK", EPI_RANK1_MASK: "RANK1_MASK"}
_DT_NAME = {torch.float32: "f32", torch.bfloat16: "bf16"}


class KernelTimers:
    """Optional HIP-event timing of individual launches on the launch stream (bench.py).

    ``only`` restricts recording to some kernel keys and ``every`` to every k-th launch of a key, so that the timed
    region carries a few event records per step instead of one pair per launch (each pair costs ~2-3 us of stream
    time)."""

    def __init__(self, only=None, every=1, pool=0):
        self.only, self.every, self.records, self._seen = only, max(1, int(every)), {}, {}
        # ``pool``: events created up front (hipEventCreate costs tens of microseconds of host time: not inside a timed region)
        self._pool = [torch.cuda.Event(enable_timing=True) for _ in range(int(pool))]

    def event(self):
        return self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)

    def span(self, key, work=None):
        if self.only is not None and key not in self.only:
            return _NULL
        if self.only is None and key.startswith("row:"):       # a span AROUND several launches that have spans of their own:
            return _NULL                                       # only on request, or the breakdown would count them twice
        n = self._seen.get(key, 0)
        self._seen[key] = n + 1
        return _Span(self, key, work) if n % self.every == 0 else _NULL

    def reset(self):
        """Forget what has been recorded so far (bench.py: the settle steps before the timed region)."""
        self.records, self._seen = {}, {}

    def summary(self):
        """key -> dict(calls, ms_total, ms_avg, flops, bytes)   (call after a device synchronize)"""
        out = {}
        for key, recs in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _ in recs]
            out[key] = dict(calls=len(ms), ms_total=sum(ms), ms_avg=sum(ms) / len(ms),
                            flops=sum(w.get("flops", 0) for _, _, w in recs),
                            bytes=sum(w.get("bytes", 0) for _, _, w in recs))
        return out


class _Span:
    def __init__(self, owner, key, work):
        self.o, self.k, self.w = owner, key, work or {}

    def __enter__(self):
        self.a = self.o.event()
        self.b = self.o.event()
        self.a.record()

    def __exit__(self, *exc):
        self.b.record()
        self.o.records.setdefault(self.k, []).append((self.a, self.b, self.w))


class _Null:
    def __enter__(self):
        pass

    def __exit__(self, *exc):
        pass


_NULL = _Null()
TIMERS = None            # set to a KernelTimers to record


def _span(meta):
    """meta() -> (timer key, dict(flops=..., bytes=...)); only evaluated while a KernelTimers is installed."""
    return TIMERS.span(*meta()) if TIMERS is not None else _NULL


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("murcl_amd kernels run on the GPU only (got a CPU tensor); there is no CPU fallback")


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


F32X3 = 2          # GEMM dtype code: f32 tensors, products as a 3-term bf16 split on the bf16 matrix pipe (include/murcl_amd.h)


def gemm_nt(A, B, *, epi=EPI_NONE, bias=None, mask=None, rowscale=None, rank1=None, rows_per_bag=0,
            out_dtype=None, colsum=False, out=None, accumulate=False, x3=False):
    """C[M,N] = epi(A[M,K] @ B[N,K]^T).  Returns C or (C, colsum_ws[ceil(M/128),N]).
    ``x3`` (f32 operands, more than 1024 rows): the products run as a 3-term bf16 split on the bf16 matrix pipe - f32-level
    accuracy (relative error ~1e-7 against the exact-f32 MFMA path) at a fraction of its time."""
    _need_cuda(A, B)
    assert not is_frag(B), "gemm_nt takes row-major operands (a fragment-order weight view belongs to panel_gemm / the K2 passes)"
    A, B = _c(A), _c(B)
    M, K = A.shape
    N = B.shape[0]
    assert B.shape[1] == K and A.dtype == B.dtype
    kq = 32 if A.dtype == torch.float32 else 64
    if K % kq:                      # head layers with a handful of outputs (dgrad K = 1, 2, 10)
        if (K <= 16 and A.dtype == torch.float32 and epi == EPI_NONE and mask is None and not colsum and not x3
                and (out_dtype or A.dtype) == torch.float32):
            C = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=A.device)      # one small launch, no padding
            assert C.is_contiguous() and C.dtype == torch.float32 and tuple(C.shape) == (M, N)
            check(_lib.lib().murcl_gemm_nt_smallk(ptr(A), ptr(B), ptr(C), M, N, K, int(accumulate), stream()), "gemm_nt_smallk")
            return C
        Kp = ((K + kq - 1) // kq) * kq                                                   # zero-pad K: one launch per operand
        A, B, K = pad_cols(A, Kp), pad_cols(B, Kp), Kp
    odt = out_dtype or A.dtype
    C = out if out is not None else torch.empty((M, N), dtype=odt, device=A.device)
    ws = torch.empty(((M + 127) // 128, N), dtype=torch.float32, device=A.device) if colsum else None
    if mask is not None:
        mask = _c(mask)
        assert mask.dtype == A.dtype and mask.shape == (M, N)
    es = A.element_size()
    x3 = bool(x3) and A.dtype == torch.float32 and C.dtype == torch.float32 and M > 1024
    with _span(lambda: (f"gemm_nt<{'f32x3' if x3 else _DT_NAME[A.dtype]},{_DT_NAME[C.dtype]},{_EPI_NAME[epi]}>",
               dict(flops=2.0 * M * N * K, bytes=(M * K + N * K) * es + M * N * C.element_size()
                    + (M * N * es if mask is not None else 0)))):
        check(_lib.lib().murcl_gemm_nt(ptr(A), ptr(B), ptr(C), M, N, K, K, K, N, F32X3 if x3 else dt(A), dt(C), epi, ptr(bias),
                                       ptr(mask), N, ptr(rowscale), ptr(rank1), rows_per_bag, ptr(ws), int(accumulate),
                                       stream()), "gemm_nt")
    return (C, ws) if colsum else C


PG_BIAS_RELU, PG_MASK, PG_RANK1_MASK, PG_BIAS, PG_GATE, PG_GATE_U = 0, 1, 2, 3, 4, 5
_PG_NAME = {0: "BIAS_RELU", 1: "MASK", 2: "RANK1_MASK", 3: "BIAS", 4: "GATE", 5: "GATE_U"}


_GATE_IDX = {}


def gate_interleave(wa, ba, wb, bb, wc, dtype):
    """CLAM's gate weights in the layout of ``panel_gate_score``: rows 32g .. 32g+15 = attention_a[16g ..], rows 32g+16 .. 32g+31 =
    attention_b[16g ..] (so that one wave of the panel GEMM holds matching (a_d, b_d) pairs) -> (W [2D,L] in ``dtype``, bias [2D]
    f32, wc per interleaved row [2D] f32: attention_c's weight at the a-rows, 0 at the b-rows)."""
    D = wa.shape[0]
    key = (D, wa.device)
    idx = _GATE_IDX.get(key)
    if idx is None:
        n = torch.arange(2 * D)
        g, j, i = n // 32, (n % 32) // 16, n % 16
        idx = _GATE_IDX[key] = ((16 * g + i) + D * j).to(wa.device)
    W = torch.cat([wa, wb], 0).index_select(0, idx)
    b = torch.cat([ba, bb], 0).index_select(0, idx).contiguous()
    c = torch.cat([wc.reshape(-1), torch.zeros_like(wc.reshape(-1))], 0).index_select(0, idx).contiguous()
    return (W.contiguous() if dtype == torch.float32 else cast(W.contiguous(), dtype)), b, c


def panel_gate_score(h, W_il, b_il, c_il, bc, parts=False):
    """CLAM's gated attention score straight from the gate GEMM's epilogue (``murcl_panel_gemm`` epilogue 4): h [M,512] bf16 ->
    raw scores s [M] f32 = sum_d tanh(a_d) sigmoid(b_d) wc_d + bc, without materialising the [M, 2D] gate pre-activations.
    ``parts``: the [N/32, M] partial score rows instead of their sum (``softmax_rows_parts`` sums them on its way to the soft-max)."""
    _need_cuda(h, W_il)
    h = _c(h)
    M, K = h.shape
    N = W_il.shape[0]
    part = torch.empty((N // 32, M), dtype=torch.float32, device=h.device)
    with _span(lambda: (f"panel_gemm<K{K},GATE>", dict(flops=2.0 * M * N * K, bytes=M * K * 2 + N * K * 2 + (N // 32) * M * 4))):
        check(_lib.lib().murcl_panel_gemm(ptr(h), ptr(W_il), None, M, N, K, PG_GATE, ptr(b_il), None, None, ptr(_c(bc)), ptr(c_il), 0,
                                          None, 0, ptr(part), 0, stream()), "panel_gemm(gate)")
    return part if parts else colsum(part)                                # (attention_c's bias rides in partial row 0)


def panel_gate_u(h, W_il, b_il, c_il, bc, keep_a=None, keep_b=None, parts=False):
    """CLAM's gate GEMM for a call that a backward pass may follow (``murcl_panel_gemm_drop`` epilogue 5): the raw scores as in
    ``panel_gate_score`` AND the gate pre-activations U [M, 2D] bf16 in the interleaved column order of ``gate_interleave`` (read
    back by ``gated_score_bwd_il``).  ``keep_a`` / ``keep_b``: DropSeed specs of the two gate Dropouts (clam.py:47-48) or None.
    -> (U, s [M] f32)."""
    _need_cuda(h, W_il)
    h = _c(h)
    M, K = h.shape
    N = W_il.shape[0]
    U = torch.empty((M, N), dtype=torch.bfloat16, device=h.device)
    part = torch.empty((N // 32, M), dtype=torch.float32, device=h.device)
    kp, sa, sb = (keep_a.keep_p, keep_a.seed, keep_b.seed) if keep_a is not None else (0.0, 0, 0)
    with _span(lambda: (f"panel_gemm<K{K},GATE_U>", dict(flops=2.0 * M * N * K, bytes=M * K * 2 + N * K * 2 + M * N * 2 + (N // 32) * M * 4))):
        check(_lib.lib().murcl_panel_gemm_drop(ptr(h), ptr(W_il), ptr(U), M, N, K, PG_GATE_U, ptr(b_il), None, None, ptr(_c(bc)), ptr(c_il), 0,
                                               None, 0, ptr(part), 0, kp, sa, sb, stream()), "panel_gemm(gate_u)")
    return U, (part if parts else colsum(part))                           # (attention_c's bias rides in partial row 0; ``parts``: as above)


def gated_score_bwd_il(U, wc, keep_a=None, keep_b=None, *, ds=None, h=None, dM=None, Mp=None, A=None, rows_per_bag=0):
    """``gated_score_bwd`` for U in the interleaved layout of ``panel_gate_u`` -> (dU in the same layout, dwc [D], dbc [1], column
    sums [2D] in NATURAL order).  ``ds`` [M] given, or None with (h [M,L], dM [B,L], Mp [B,L], A [M], rows_per_bag): the pooling and
    soft-max backward are then taken in the same pass (ds_n = A_n (h_n . dM - Mp . dM))."""
    U = _c(U)
    M, W = U.shape
    D = W // 2
    dU = torch.empty_like(U)
    dwc = torch.empty((D,), dtype=torch.float32, device=U.device)
    dbc = torch.empty((1,), dtype=torch.float32, device=U.device)
    dbab = torch.empty((2 * D,), dtype=torch.float32, device=U.device)
    part = torch.empty((1024 * (3 * D + 1),), dtype=torch.float32, device=U.device)
    kp, sa, sb = (keep_a.keep_p, keep_a.seed, keep_b.seed) if keep_a is not None else (0.0, 0, 0)
    L = 0
    if ds is None:
        h, dM, Mp, A = _c(h), _c(dM), _c(Mp), _c(A)
        assert h.dtype == U.dtype and dM.dtype == Mp.dtype == A.dtype == torch.float32
        L = h.shape[1]
    else:
        ds = _c(ds)
    with _span(lambda: ("gated_score_bwd_il" + ("<one pass>" if L else ""), dict(bytes=2 * U.numel() * 2 + (M * L * 2 if L else 0)))):
        check(_lib.lib().murcl_gated_score_bwd_il(ptr(U), ptr(wc), ptr(ds), ptr(dU), ptr(dwc), ptr(dbc), ptr(dbab), ptr(part), M, D, dt(U),
                                                  kp, sa, sb, ptr(h), ptr(dM), ptr(Mp), ptr(A), L, rows_per_bag, stream()), "gated_score_bwd_il")
    return dU, dwc, dbc, dbab


def gated_bwd_il_supported(M, D, L, rows_per_bag):
    G = D // 8
    return (D % 16 == 0 and G <= 64 and (G & (G - 1)) == 0 and L % (8 * G) == 0 and L // (8 * G) in (1, 2, 4) and rows_per_bag > 0
            and M % rows_per_bag == 0 and rows_per_bag % (256 // G) == 0)


def panel_supported(M, N, K, epi, rows_per_bag=0):
    return bool(_lib.lib().murcl_panel_gemm_supported(M, N, K, epi, rows_per_bag))


def panel_gemm(A, W, epi, *, bias=None, want_bitmask=False, bitmask=None, rowscale=None, rank1=None, rows_per_bag=0,
               colsum=False, colsum_into=None, colsum_defer=False, reverse=False, stream_a=False, out=None, bitmask_out=None,
               drop=None):
    """bf16 weight-stationary C = epi(A @ W^T).  Returns (C, bitmask_out or None, colsum or None).
    ``colsum_into`` ([N] f32): the column sums are ADDED to it (gradient accumulation) and returned as None.
    ``colsum_defer``: no second launch - the third result is (partial rows [R,N] f32, R) for ``gemm_tn(colsum_parts=...)``,
    the weight gradient of the same layer, whose reduce launch adds them up on the way.
    ``reverse``: visit the row tiles last-to-first (cache reuse after a producer that walked forward; same result).
    ``stream_a``: load A with the non-temporal policy (K = 512): it is read once and should not displace the output, which
    the next kernel reads, from the Infinity Cache.
    ``drop`` (a DropSeed; PG_BIAS_RELU with ``want_bitmask``, K = 512): Dropout behind the ReLU inside the epilogue, the mask
    never materialised; the bit mask records what survives (= ``dropout_relu_bitmask`` on the output, without that pass)."""
    _need_cuda(A, W)
    wfrag = 4 if is_frag(W) else 0                          # (W in fragment order: walk_reverse bit 2 of the C-ABI)
    A, W = _c(A), (W if wfrag else _c(W))
    M, K = A.shape
    N = W.shape[0]
    assert A.dtype == torch.bfloat16 and W.dtype == torch.bfloat16 and W.shape[1] == K
    assert not wfrag or (K == 512 and epi in (PG_BIAS_RELU, PG_MASK) and drop is None), "fragment-order weights: K = 512, BIAS_RELU / MASK"
    # ``out`` / ``bitmask_out``: caller-owned result buffers (row blocks of a larger tensor: functional.EncoderSession)
    C = torch.empty((M, N), dtype=torch.bfloat16, device=A.device) if out is None else out
    assert C.is_contiguous() and C.dtype == torch.bfloat16 and tuple(C.shape) == (M, N)
    bm = None
    if want_bitmask:
        bm = torch.empty((M, N // 8), dtype=torch.uint8, device=A.device) if bitmask_out is None else bitmask_out
        assert bm.is_contiguous() and bm.dtype == torch.uint8 and bm.numel() == M * N // 8
    cs = torch.empty((N,), dtype=torch.float32, device=A.device) if (colsum and colsum_into is None and not colsum_defer) else None
    if colsum_into is not None:
        assert not colsum_defer
        assert colsum_into.is_contiguous() and colsum_into.dtype == torch.float32 and colsum_into.numel() == N
    ws = torch.empty((256 * N,), dtype=torch.float32, device=A.device) if (colsum or colsum_into is not None or colsum_defer) else None
    with _span(lambda: (f"panel_gemm<K{K},{_PG_NAME[epi]}>",
               dict(flops=2.0 * M * N * K, bytes=M * K * 2 + N * K * 2 + M * N * 2 + (M * N // 8 if (want_bitmask or bitmask is not None) else 0)))):
        if drop is not None:
            assert epi == PG_BIAS_RELU and want_bitmask and K == 512
            check(_lib.lib().murcl_panel_gemm_drop(ptr(A), ptr(W), ptr(C), M, N, K, epi, ptr(bias), ptr(bm), ptr(bitmask),
                                                   ptr(rowscale), ptr(rank1), rows_per_bag,
                                                   ptr(colsum_into if colsum_into is not None else cs),
                                                   int(colsum_into is not None), ptr(ws), int(reverse) | (2 if stream_a else 0) | wfrag,
                                                   drop.keep_p, drop.seed, 0, stream()), "panel_gemm(drop)")
        else:
            check(_lib.lib().murcl_panel_gemm(ptr(A), ptr(W), ptr(C), M, N, K, epi, ptr(bias), ptr(bm), ptr(bitmask),
                                              ptr(rowscale), ptr(rank1), rows_per_bag,
                                              ptr(colsum_into if colsum_into is not None else cs),
                                              int(colsum_into is not None), ptr(ws), int(reverse) | (2 if stream_a else 0) | wfrag, stream()),
                  "panel_gemm")
    if colsum_defer:
        return C, bm, (ws, _lib.lib().murcl_panel_gemm_colsum_rows(M, N, K, epi))
    return C, bm, cs


def relu_bitmask(x):
    """x [M,N] (M % 32 == 0, N % 32 == 0) -> the panel GEMM's 1-bit mask of x > 0, [M, N/8] uint8."""
    _need_cuda(x)
    x = _c(x)
    M, N = x.shape
    bits = torch.empty((M, N // 8), dtype=torch.uint8, device=x.device)
    check(_lib.lib().murcl_relu_bitmask(ptr(x), ptr(bits), M, N, N, dt(x), stream()), "relu_bitmask")
    return bits


import os as _os
_TN_SQ = True             # test hook: False keeps the 256 x 128 atomics kernel (the form shapes without a workspace take)


def gemm_tn_with_colsum(A, B):
    """(A^T B, column sums of A) for a fresh bag-level f32 gradient pair - a Linear's (dW, db) from (dy, x) - in ONE launch where the
    single-writer 32 x 32 kernel takes the shape (its workgroups of the first column tile form the sums with one more MFMA per
    step), else as ``gemm_tn`` + ``colsum``."""
    _need_cuda(A, B)
    A, B = _c(A), _c(B)
    M, N1 = A.shape
    N2 = B.shape[1]
    if A.dtype == torch.float32 and B.dtype == torch.float32 and M <= 512 and _TN_SMALL_GROUP and N1 % 4 == 0 and N2 % 4 == 0:
        C = torch.empty((N1, N2), dtype=torch.float32, device=A.device)
        cs = torch.empty((N1,), dtype=torch.float32, device=A.device)
        arr = (_lib.TnProblem * 1)(_lib.TnProblem(ptr(A), ptr(B), ptr(C), None, ptr(cs), M, N1, N2, N1, N2, N2, 0, _lib.TN_OVERWRITE, 1.0))
        if _lib.lib().murcl_gemm_tn_grouped(arr, 1, F32, None, 0, stream()) == 0:
            return C, cs
    return gemm_tn(A, B), colsum(A)


def gemm_tn(A, B, *, splits=0, out=None, colsum_into=None, colsum_parts=None, x3=False):
    """C[N1,N2] (f32) = A[M,N1]^T @ B[M,N2]  (adds into ``out`` when given).  ``colsum_into`` [N1] f32: the column sums
    of A are ADDED to it in the same launch (the bias gradient that goes with this weight gradient); with
    ``colsum_parts`` = (rows [R,N1] f32, R) from ``panel_gemm(colsum_defer=True)`` those rows are summed instead."""
    _need_cuda(A, B)
    A, B = _c(A), _c(B)
    M, N1 = A.shape
    N2 = B.shape[1]
    assert B.shape[0] == M and A.dtype == B.dtype
    epc = 4 if A.dtype == torch.float32 else 8
    if colsum_parts is not None and (colsum_into is None or N1 % 4):
        raise ValueError("colsum_parts needs colsum_into and N1 % 4 == 0")
    if N1 % epc:                    # tiny head gradients (N1 = 1, 2, 10): zero-pad the columns of A, slice the result
        res = gemm_tn(pad_cols(A, ((N1 + epc - 1) // epc) * epc), B, splits=splits)[:N1]      # (the first N1 rows: contiguous)
        if colsum_into is not None:
            colsum(A, out=colsum_into, accumulate=True)
        if out is None:
            return res.contiguous()
        if out.is_contiguous() and out.dtype == torch.float32:
            return axpby(out, res, 1.0, 1.0, out=out)
        return out.add_(res)
    if colsum_into is not None:
        assert colsum_into.dtype == torch.float32 and colsum_into.is_contiguous() and colsum_into.numel() == N1
    if (out is None and colsum_into is None and A.dtype == torch.float32 and M <= 512 and splits <= 0 and not x3 and _TN_SMALL_GROUP
            and N2 % 4 == 0):
        # a fresh bag-level gradient: the single-writer 32 x 32 kernel WRITES it (no zero-fill launch in front, no read of C)
        C = torch.empty((N1, N2), dtype=torch.float32, device=A.device)
        arr = (_lib.TnProblem * 1)(_lib.TnProblem(ptr(A), ptr(B), ptr(C), None, None, M, N1, N2, N1, N2, N2, 0, _lib.TN_OVERWRITE, 1.0))
        if _lib.lib().murcl_gemm_tn_grouped(arr, 1, F32, None, 0, stream()) == 0:
            return C
        # (the library declines flagged products when its small-tile kernel is switched off, MURCL_TN_SMALL=0: zero-fill + the
        #  accumulating entry point below)
    C = out if out is not None else torch.zeros((N1, N2), dtype=torch.float32, device=A.device)
    wide = A.dtype == torch.bfloat16 and N1 % 256 == 0 and N2 % 128 == 0 and M >= 4096     # murcl_gemm_tn's dispatch
    wsb = _lib.lib().murcl_gemm_tn_workspace_bytes(M, N1, N2, dt(A)) if (splits <= 0 and _TN_SQ) else 0
    if wsb:         # 256 x 256 tiles, partial sums through a workspace + reduce launch (no float atomics)
        ws = torch.empty((wsb // 4,), dtype=torch.float32, device=A.device)
        with _span(lambda: (f"gemm_tn_sq<{_DT_NAME[A.dtype]}>",
                   dict(flops=2.0 * M * N1 * N2, bytes=M * (N1 + N2) * A.element_size() + N1 * N2 * 4))):
            check(_lib.lib().murcl_gemm_tn_ws(ptr(A), ptr(B), ptr(C), M, N1, N2, N1, N2, N2, dt(A), splits, ptr(colsum_into),
                                              ptr(ws), wsb, ptr(colsum_parts[0]) if colsum_parts else None,
                                              colsum_parts[1] if colsum_parts else 0, stream()), "gemm_tn_ws")
        return C
    if colsum_parts is not None:                      # other paths: the partial rows get their own small launch
        colsum(colsum_parts[0].view(-1, N1)[:colsum_parts[1]], out=colsum_into, accumulate=True)
        colsum_into = None
    x3 = bool(x3) and A.dtype == torch.float32 and M >= 4096       # (``x3``: the 3-term bf16 split of gemm_nt, long f32 reductions)
    with _span(lambda: (f"gemm_tn{'_wide' if wide else ''}<{'f32x3' if x3 else _DT_NAME[A.dtype]}>",
               dict(flops=2.0 * M * N1 * N2, bytes=M * (N1 + N2) * A.element_size() + N1 * N2 * 4))):
        check(_lib.lib().murcl_gemm_tn(ptr(A), ptr(B), ptr(C), M, N1, N2, N1, N2, N2, F32X3 if x3 else dt(A), splits, ptr(colsum_into),
                                       stream()), "gemm_tn")
    return C


def gemm_tn_grouped(problems, fresh=False):
    """Several weight gradients in ONE launch (+ one reduce launch): ``problems`` = list of (A [M,N1], B [M,N2], out or None,
    colsum_into or None, colsum_parts or None[, opts]) with the meanings of ``gemm_tn``; returns the list of C tensors.  bf16 products
    with N1, N2 multiples of 256 and M >= 16384 share one round of workgroups (murcl_gemm_tn_grouped); anything else, or more
    than four products, runs through ``gemm_tn`` one by one.  ``opts`` (dict, grouped path only - check ``gemm_tn_grouped_ok``):
    ``scale`` (product and column sums times a factor), ``deinterleave`` (rows arrive as alternating 16-row blocks of two halves: C
    = [first half; second half] in natural order).  ``fresh``: products without ``out`` / column sums into a fresh tensor are WRITTEN
    by the reduce launch (no zero-fill launch before it)."""
    problems = [tuple(p) + (None,) * (6 - len(p)) for p in problems]
    if gemm_tn_small_grouped_ok(problems):
        n = len(problems)
        arr = (_lib.TnProblem * n)()
        keep, Cs = [], []
        for g, (A, B, out, ci, _, _) in enumerate(problems):
            _need_cuda(A, B)
            A, B = _c(A), _c(B)
            (M, N1), N2 = A.shape, B.shape[1]
            write = out is None and ci is None                 # a fresh product is written, not added to zeros
            C = out if out is not None else (torch.empty if write else torch.zeros)((N1, N2), dtype=torch.float32, device=A.device)
            assert C.dtype == torch.float32 and C.is_contiguous() and tuple(C.shape) == (N1, N2)
            assert ci is None or (ci.dtype == torch.float32 and ci.is_contiguous() and ci.numel() == N1)
            keep.append((A, B))
            Cs.append(C)
            arr[g] = _lib.TnProblem(ptr(A), ptr(B), ptr(C), None, ptr(ci), M, N1, N2, N1, N2, N2, 0, _lib.TN_OVERWRITE if write else 0, 1.0)
        if _lib.lib().murcl_gemm_tn_grouped(arr, n, F32, None, 0, stream()) == 0:
            return Cs
        # (declined: the library's small-tile kernel is switched off, MURCL_TN_SMALL=0 - one product at a time below)
        return [gemm_tn(A, B, out=out, colsum_into=ci) for A, B, out, ci, _, _ in problems]
    if not gemm_tn_grouped_ok(problems):
        assert all(p[5] is None for p in problems), "scale / deinterleave need the grouped launch (gemm_tn_grouped_ok)"
        return [gemm_tn(A, B, out=out, colsum_into=ci, colsum_parts=cp) for A, B, out, ci, cp, _ in problems]
    n = len(problems)
    arr = (_lib.TnProblem * n)()
    keep, Cs = [], []
    for g, (A, B, out, ci, cp, opts) in enumerate(problems):
        _need_cuda(A, B)
        A, B = _c(A), _c(B)
        M, N1 = A.shape
        N2 = B.shape[1]
        flags, scale = 0, 1.0
        if out is None and fresh:
            C = torch.empty((N1, N2), dtype=torch.float32, device=A.device)
            flags |= _lib.TN_OVERWRITE
        else:
            C = out if out is not None else torch.zeros((N1, N2), dtype=torch.float32, device=A.device)
        assert C.dtype == torch.float32 and C.is_contiguous() and tuple(C.shape) == (N1, N2)
        if ci is not None:
            assert ci.dtype == torch.float32 and ci.is_contiguous() and ci.numel() == N1
        if opts:
            if opts.get("scale") is not None:
                flags, scale = flags | _lib.TN_SCALE, float(opts["scale"])
            if opts.get("deinterleave"):
                flags |= _lib.TN_DEINTERLEAVE
            if opts.get("overwrite"):
                flags |= _lib.TN_OVERWRITE
        keep.append((A, B, cp))
        Cs.append(C)
        arr[g] = _lib.TnProblem(ptr(A), ptr(B), ptr(C), ptr(cp[0]) if cp else None, ptr(ci), M, N1, N2, N1, N2, N2, cp[1] if cp else 0,
                                flags, scale)
    wsb = _lib.lib().murcl_gemm_tn_grouped_workspace_bytes(arr, n, BF16)
    assert wsb, "gemm_tn_grouped_ok and the library disagree"
    ws = torch.empty((wsb // 4,), dtype=torch.float32, device=Cs[0].device)
    with _span(lambda: (f"gemm_tn_sq_grouped{n}<bf16>",
               dict(flops=sum(2.0 * A.shape[0] * A.shape[1] * B.shape[1] for A, B, _ in keep),
                    bytes=sum(A.shape[0] * (A.shape[1] + B.shape[1]) * 2 + A.shape[1] * B.shape[1] * 4 for A, B, _ in keep)))):
        check(_lib.lib().murcl_gemm_tn_grouped(arr, n, BF16, ptr(ws), wsb, stream()), "gemm_tn_grouped")
    return Cs


def gemm_tn_small_grouped_ok(problems):
    """2-4 f32 products of at most 512 rows each (bag-level / rollout-level weight gradients), accumulated into their outputs:
    ONE launch of the 32 x 32-tile single-writer kernel over all their tiles."""
    if not (_TN_SMALL_GROUP and 1 < len(problems) <= 4):
        return False
    for p in problems:
        A, B = p[0], p[1]
        if not (A.dtype == torch.float32 and B.dtype == torch.float32 and A.dim() == 2 and B.dim() == 2 and 0 < A.shape[0] <= 512
                and A.shape[0] == B.shape[0] and A.shape[1] % 4 == 0 and B.shape[1] % 4 == 0 and p[4] is None and not p[5]):
            return False
    return True


def gemm_tn_grouped_ok(problems):
    """Does this list run as ONE grouped launch?  (bf16, 2-4 products, N1 and N2 multiples of 256, M >= 16384, at most 32 tiles each,
    all (product, tile) pairs within one round of workgroups.)"""
    if not (_TN_SQ and 1 < len(problems) <= 4):
        return False
    pairs = 0
    for p in problems:
        A, B, ci, cp = p[0], p[1], p[3], p[4]
        if not (A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and A.shape[1] % 256 == 0 and B.shape[1] % 256 == 0
                and A.shape[0] >= 16384 and A.shape[0] == B.shape[0] and (cp is None or ci is not None)):
            return False
        tiles = (A.shape[1] // 256) * (B.shape[1] // 256)
        if tiles > cu_budget() // 8:
            return False
        pairs += tiles
    return pairs <= cu_budget()


def cu_budget():
    """CUs the persistent kernels size their one round of workgroups for (murcl_cu_budget; 256 unless ``set_cu_budget`` lowered it)."""
    return _lib.lib().murcl_cu_budget()


def set_cu_budget(cus):
    """Leave 256 - ``cus`` CUs to somebody else (RCCL's channel workgroups while a collective overlaps the backward pass): the
    encoder-sized launches then place all their workgroups at once instead of running a second round.  -> the budget in force."""
    return _lib.lib().murcl_set_cu_budget(int(cus))


def pool_chunks(B, N, dtype_code):
    cr, nc = ctypes.c_int(), ctypes.c_int()
    _lib.lib().murcl_abmil_pool_workspace(B, N, dtype_code, ctypes.byref(cr), ctypes.byref(nc))
    return cr.value, nc.value


def _pool_work(B, N, L, D, es):
    # algorithmic bytes per bag: H once + scores out + pooled M out; Wa amortised over the launch
    return dict(flops=B * (2.0 * N * L * D + 2.0 * N * D + 2.0 * N * L), bytes=B * (N * L * es + N * 4 + L * 4) + L * D * es)


def abmil_pool_partials(H, Wa, ba, wb, bb, exact_tanh=None, scores=None):
    """The K2 streaming pass on its own - ONE launch: H [B,N,512], Wa [128,512] (same dtype) -> raw scores [B,N] f32 and the chunk
    partials ``part`` [B*S*(L+4)] f32 ((sum p.H, m, l) per (bag, row chunk)).  The per-bag merge belongs to the consumer:
    ``abmil_pool_decoder`` (the training / inference path), or ``abmil_pool_combine`` for A, M, ml as tensors."""
    _need_cuda(H, Wa)
    wfrag = 2 if is_frag(Wa) else 0                         # (Wa in fragment order: bit 1 of the C-ABI's exact_tanh / flags argument)
    H, Wa = _c(H), (Wa if wfrag else _c(Wa))
    B, N, L = H.shape
    D = Wa.shape[0]
    if exact_tanh is None:
        exact_tanh = H.dtype == torch.float32
    dev = H.device
    _, S = pool_chunks(B, N, dt(H))
    if scores is None:
        scores = torch.empty((B, N), dtype=torch.float32, device=dev)
    else:
        assert scores.is_contiguous() and scores.dtype == torch.float32 and tuple(scores.shape) == (B, N)
    part = torch.empty((B * S * (L + 4),), dtype=torch.float32, device=dev)
    es = H.element_size()
    # "row:k2_fwd" = bench.py's roofline_k2: the whole K2 row of the step, which is this one launch since round 6
    with _span(lambda: (f"row:k2_fwd<{_DT_NAME[H.dtype]}>", _pool_work(B, N, L, D, es))):
        with _span(lambda: (f"abmil_pool_fwd<{_DT_NAME[H.dtype]}>", _pool_work(B, N, L, D, es))):
            check(_lib.lib().murcl_abmil_pool_fwd(ptr(H), ptr(Wa), ptr(ba), ptr(wb), ptr(bb), ptr(scores), None, None,
                                                  None, ptr(part), B, N, L, D, dt(H), int(exact_tanh) | wfrag, stream()),
                  "abmil_pool_fwd")
    return scores, part


def abmil_pool_combine(scores, part, dtype, out=None):
    """scores [B,N], part (``abmil_pool_partials``) -> A [B,N] = softmax(s)/sqrt(N), M [B,512], ml [B,2] - the per-bag merge as its own
    launch (``dtype``: the dtype H had - it fixes the chunking)."""
    B, N = scores.shape
    L = 512
    dev = scores.device
    code = F32 if dtype == torch.float32 else BF16
    _, S = pool_chunks(B, N, code)
    if out is None:
        A = torch.empty((B, N), dtype=torch.float32, device=dev)
        M = torch.empty((B, L), dtype=torch.float32, device=dev)
        ml = torch.empty((B, 2), dtype=torch.float32, device=dev)
    else:
        A, M, ml = out
        assert all(t.is_contiguous() and t.dtype == torch.float32 for t in out)
        assert tuple(A.shape) == (B, N) and tuple(M.shape) == (B, L) and tuple(ml.shape) == (B, 2)
    with _span(lambda: ("abmil_pool_combine", dict(flops=0.0, bytes=B * (2 * N * 4 + S * (L + 4) * 4 + L * 4)))):
        check(_lib.lib().murcl_abmil_pool_combine(ptr(scores), ptr(part), ptr(A), ptr(M), ptr(ml), B, N, code, stream()),
              "abmil_pool_combine")
    return A, M, ml


def abmil_pool_fwd(H, Wa, ba, wb, bb, exact_tanh=None, out=None):
    """H [B,N,512], Wa [128,512] (same dtype) -> scores [B,N], A [B,N], M [B,512], ml [B,2] (all f32): the streaming pass and the
    per-bag merge as two launches (stand-alone use; the modules run ``abmil_pool_partials`` + ``abmil_pool_decoder``).
    ``out`` = (scores, A, M, ml) caller-owned contiguous buffers of those shapes."""
    scores, part = abmil_pool_partials(H, Wa, ba, wb, bb, exact_tanh, scores=None if out is None else out[0])
    A, M, ml = abmil_pool_combine(scores, part, H.dtype, out=None if out is None else out[1:])
    return scores, A, M, ml


def abmil_pool_decoder(part, B, N, dtype, wd, bd, relu=True, out=None):
    """The decoder layer with K2's per-bag merge on load (abmil.py:29-32,43): part (``abmil_pool_partials`` of a [B,N,512] batch in
    ``dtype``) -> out [B,Lout] = relu(M wd^T + bd), and the by-products M [B,512], ml [B,2] the backward pass reads - ONE launch where
    ``abmil_pool_combine`` + ``gemm_nt`` are two.  ``out`` = (out, M, ml) caller-owned buffers.  Shapes outside the kernel (more than
    512 row chunks per bag) take those two launches."""
    L, Lout = 512, wd.shape[0]
    dev = part.device
    code = F32 if dtype == torch.float32 else BF16
    if out is None:
        o = torch.empty((B, Lout), dtype=torch.float32, device=dev)
        M = torch.empty((B, L), dtype=torch.float32, device=dev)
        ml = torch.empty((B, 2), dtype=torch.float32, device=dev)
    else:
        o, M, ml = out
        assert all(t.is_contiguous() and t.dtype == torch.float32 for t in out)
        assert tuple(o.shape) == (B, Lout) and tuple(M.shape) == (B, L) and tuple(ml.shape) == (B, 2)
    wd = _c(wd)
    assert wd.dtype == torch.float32 and wd.shape[1] == L
    _, S = pool_chunks(B, N, code)
    with _span(lambda: ("abmil_pool_decoder", dict(flops=2.0 * B * L * Lout, bytes=(B * S * (L + 4) + Lout * L + B * (L + Lout)) * 4))):
        rc = _lib.lib().murcl_abmil_pool_decoder(ptr(part), ptr(wd), ptr(bd), ptr(M), ptr(ml), ptr(o), B, N, L, Lout, code,
                                                 int(relu), stream())
    if rc == -1:                                  # (a HIP kernel path either way: the merge launch + the library's f32 GEMM)
        scratch = torch.empty((1,), dtype=torch.float32, device=dev)
        check(_lib.lib().murcl_abmil_pool_combine(ptr(scratch), ptr(part), None, ptr(M), ptr(ml), B, N, code, stream()),
              "abmil_pool_combine")
        gemm_nt(M, wd, epi=EPI_BIAS_RELU if relu else EPI_BIAS, bias=bd, out=o)
    else:
        check(rc, "abmil_pool_decoder")
    return o, M, ml


def abmil_attention(scores, ml):
    """A [B,N] = exp(s - m) / (l sqrt N) from the raw scores and the soft-max statistics of a pooling pass (abmil.py:40-41): the
    attention row ``last_attention`` shows, formed on demand (no training-step launch needs it in the forward pass)."""
    B, N = scores.shape
    A = torch.empty_like(scores)
    check(_lib.lib().murcl_abmil_pool_combine(ptr(_c(scores)), None, ptr(A), None, ptr(_c(ml)), B, N, F32, stream()),
          "abmil_pool_combine")
    return A


def abmil_pool_bwd(H, Wa, ba, wb, scores, ml, M, dM, exact_tanh=None, into=None, want_A=False):
    """-> dT [B*N,128] (dtype of H; 32 spare rows allocated behind it), dba[128], dwb[128], dbb[1] (, A [B,N] with ``want_A``: the
    normalised attention row softmax(s)/sqrt(N), which the pass has in registers - the rank-1 input gradient's row scale).
    ``into`` = (dba, dwb, dbb) f32 buffers: the kernel ADDS to them (gradient accumulation) instead of fresh zeros."""
    _need_cuda(H, Wa, dM)
    wfrag = 2 if is_frag(Wa) else 0
    H, Wa, dM = _c(H), (Wa if wfrag else _c(Wa)), _c(dM)
    B, N, L = H.shape
    D = Wa.shape[0]
    if exact_tanh is None:
        exact_tanh = H.dtype == torch.float32
    dev = H.device
    dT_full = torch.empty((B * N + 32, D), dtype=H.dtype, device=dev)
    if into is not None:
        dba, dwb, dbb = into
        assert all(t.is_contiguous() and t.dtype == torch.float32 for t in into) and dba.numel() == D and dwb.numel() == D
    else:
        z = torch.zeros((2 * D + 1,), dtype=torch.float32, device=dev)          # one fill for the three accumulators
        dba, dwb, dbb = z[:D], z[D:2 * D], z[2 * D:]
    es = H.element_size()
    A = torch.empty((B, N), dtype=torch.float32, device=dev) if want_A else None
    part = torch.empty((512 * (2 * D + 1),), dtype=torch.float32, device=dev)      # per-workgroup parameter-gradient rows
    with _span(lambda: (f"abmil_pool_bwd<{_DT_NAME[H.dtype]}>",
               dict(flops=B * (2.0 * N * L * D + 2.0 * N * L), bytes=B * (N * L * es + N * D * es + N * 4) + L * D * es))):
        check(_lib.lib().murcl_abmil_pool_bwd(ptr(H), ptr(Wa), ptr(ba), ptr(wb), ptr(scores), ptr(ml), ptr(M), ptr(dM),
                                              ptr(dT_full), ptr(dba), ptr(dwb), ptr(dbb), ptr(part), ptr(A), B, N, L, D, dt(H),
                                              int(exact_tanh) | wfrag, stream()), "abmil_pool_bwd")
    if want_A:
        return dT_full[:B * N], dba, dwb, dbb, A
    return dT_full[:B * N], dba, dwb, dbb


_NTX_XCHG = True          # test hook: n <= 128, P = 128 through the one-exchange kernel (False: the recompute kernel other P take)
_NTX_BUF = {}


def _ntx_xchg(dev, batches):
    """The exchange buffer of murcl_ntxent_small_xchg for this device AND stream: allocated (and zeroed) once, grown when a call needs
    more.  Launches that may overlap in time must not share one (generation word, arrival counter, granules: ntxent.hip), and launches
    of one stream never overlap - so the cache is keyed by (device, stream)."""
    need = _lib.lib().murcl_ntxent_xchg_bytes(batches)
    key = (dev, stream())
    buf = _NTX_BUF.get(key)
    if buf is None or buf.numel() < need:
        buf = _NTX_BUF[key] = torch.zeros((need,), dtype=torch.uint8, device=dev)
    return buf


def ntxent(z, temperature, want_grad=True, grad_lo=0, grad_hi=None, pair_stride=None):
    """z [2B,128] f32 -> (loss [1], dz [2B,128] or None, sim [B]).  Rows: cat(view 0, view 1) by default; with
    ``pair_stride`` = bags per rank, an all-gathered [rank][view][bag] batch (global bag id = rank * pair_stride + b)."""
    _need_cuda(z)
    z = _c(z.float())
    n, P = z.shape
    Bh = n // 2
    if grad_hi is None:
        grad_hi = Bh
    ps = Bh if pair_stride is None else int(pair_stride)
    dev = z.device
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    dz = torch.empty_like(z) if want_grad else None
    sim = torch.empty((Bh,), dtype=torch.float32, device=dev)
    if n <= 128 and P == 128 and _NTX_XCHG:
        with _span(lambda: ("ntxent", dict(flops=6.0 * n * n * P, bytes=2 * n * P * 4))):
            check(_lib.lib().murcl_ntxent_small_xchg(ptr(z), 1, n, P, float(temperature), ptr(loss), ptr(dz), ptr(sim), grad_lo, grad_hi,
                                                     ps, ptr(_ntx_xchg(dev, 1)), stream()), "ntxent_small_xchg")
        return loss, dz, sim
    ws = torch.empty((_lib.lib().murcl_ntxent_workspace_bytes(n) + 3) // 4, dtype=torch.float32, device=dev)
    with _span(lambda: ("ntxent", dict(flops=6.0 * n * n * P, bytes=2 * n * P * 4))):
        check(_lib.lib().murcl_ntxent_fwd_bwd(ptr(z), n, P, float(temperature), ptr(loss), ptr(dz), ptr(sim), grad_lo,
                                              grad_hi, ps, ptr(ws), stream()), "ntxent_fwd_bwd")
    return loss, dz, sim


def ntxent_batched(z, temperature, want_grad=True):
    """z [T,2B,128] f32 (2B <= 128): T independent NT-Xent problems in one launch -> (loss [T], dz [T,2B,128] or None, sim [T,B])."""
    _need_cuda(z)
    z = _c(z.float())
    T_, n, P = z.shape
    dev = z.device
    loss = torch.empty((T_,), dtype=torch.float32, device=dev)
    dz = torch.empty_like(z) if want_grad else None
    sim = torch.empty((T_, n // 2), dtype=torch.float32, device=dev)
    if _NTX_XCHG and P == 128 and n <= 128:
        with _span(lambda: ("ntxent", dict(flops=6.0 * T_ * n * n * P, bytes=2 * T_ * n * P * 4))):
            check(_lib.lib().murcl_ntxent_small_xchg(ptr(z), T_, n, P, float(temperature), ptr(loss), ptr(dz), ptr(sim), 0, n // 2, n // 2,
                                                     ptr(_ntx_xchg(dev, T_)), stream()), "ntxent_small_xchg")
        return loss, dz, sim
    with _span(lambda: ("ntxent", dict(flops=6.0 * T_ * n * n * P, bytes=2 * T_ * n * P * 4))):
        check(_lib.lib().murcl_ntxent_fwd_bwd_batched(ptr(z), T_, n, P, float(temperature), ptr(loss), ptr(dz), ptr(sim), stream()),
              "ntxent_fwd_bwd_batched")
    return loss, dz, sim


def cast(x, dtype):
    _need_cuda(x)
    x = _c(x)
    if x.dtype == dtype:
        return x
    y = torch.empty_like(x, dtype=dtype)
    check(_lib.lib().murcl_cast(ptr(x), ptr(y), x.numel(), dt(x), dt(y), stream()), "cast")
    return y


# ---- weight views: compute-dtype copies / transposes of parameter matrices, rebuilt in ONE launch when a parameter
# changed.  The views are reused across calls only for parameters a FlatAdam owns: its step() advances PARAM_EPOCH (the
# kernel writes through raw pointers, which torch's version counters do not see) and in-place torch ops on such a
# parameter (load_state_dict, copy_) bump its ``_version``.  Anything else (``p.data`` edits have their own version
# counter) cannot be tracked, so unmanaged parameters are re-converted on every call - still one launch, not seven.
PARAM_EPOCH = 0
MANAGED_PARAMS = {}           # data_ptr() -> weakref of a parameter whose every raw update is announced through PARAM_EPOCH
_VIEWS = {}


def manage_param(p, on=True):
    import weakref
    if on:
        MANAGED_PARAMS[p.data_ptr()] = weakref.ref(p)
    else:
        MANAGED_PARAMS.pop(p.data_ptr(), None)


def is_managed(p):
    r = MANAGED_PARAMS.get(p.data_ptr())
    return r is not None and r() is p


def is_frag(w):
    """Is ``w`` a FRAGMENT-ORDER weight view (``weight_views`` spec with a fourth element "frag")?  Such a [R,512] bf16 tensor holds the
    persistent kernels' MFMA weight fragments, not rows (csrc/elementwise.hip frag_index): only ``panel_gemm`` (K = 512) and the K2
    pooling passes take it - they then fetch their weight slice with coalesced 1-KiB loads beside their first tiles instead of staging
    it through the still empty tile ring first."""
    return getattr(w, "_murcl_frag", False)


def weight_views(specs):
    """specs: sequence of (param [R,C] f32 contiguous, transpose: bool, dtype[, "frag"]).  Returns the prepared tensors (read
    only; they persist and are refreshed lazily).  "frag": the view ([.., 512] after the optional transpose, rows % 16 == 0, bf16) in
    fragment order (``is_frag``)."""
    import numpy as np
    specs = [(sp[0], sp[1], sp[2], len(sp) > 3 and sp[3] == "frag") for sp in specs]
    key = tuple((p.data_ptr(), p.shape[0], p.shape[1], bool(tr), d, fr) for p, tr, d, fr in specs)
    ver = (PARAM_EPOCH, tuple(p._version for p, _, _, _ in specs))
    st = _VIEWS.get(key)
    if st is None:
        if len(_VIEWS) >= 64:
            _VIEWS.clear()
            _MERGED.clear()
        dev = specs[0][0].device
        outs, rec, max_tiles = [], [], 0
        for p, tr, d, fr in specs:
            _need_cuda(p)
            assert p.dtype == torch.float32 and p.dim() == 2 and p.is_contiguous()
            R, C = p.shape
            o = torch.empty((C, R) if tr else (R, C), dtype=d, device=dev)
            if fr:
                assert d == torch.bfloat16 and o.shape[1] == 512 and o.shape[0] % 16 == 0, "fragment-order views: [16k, 512] bf16"
                o._murcl_frag = True
            outs.append(o)
            rec.append((p.data_ptr(), o.data_ptr(), R, C, int(tr) | (2 if fr else 0), _lib.BF16 if d == torch.bfloat16 else _lib.F32))
            max_tiles = max(max_tiles, ((R + 31) // 32) * ((C + 31) // 32))
        jobs = np.array(rec, dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("rows", "<i4"), ("cols", "<i4"),
                                             ("tr", "<i4"), ("dt", "<i4")]))
        table = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
        st = _VIEWS[key] = dict(outs=outs, table=table, n=len(rec), max_tiles=max_tiles, ver=None,
                                keep=[p for p, _, _, _ in specs], managed=False,
                                tiles=[((r[2] + 31) // 32) * ((r[3] + 31) // 32) for r in rec])
    st["managed"] = all(is_managed(p) for p, _, _, _ in specs)
    if st["ver"] != ver or not st["managed"]:
        check(_lib.lib().murcl_cast_batch(ptr(st["table"]), st["n"], st["max_tiles"], stream()), "cast_batch")
        st["ver"] = ver
    return st["outs"]


def _job_views(key, params, build):
    """Shared cache of ``weight_views`` / ``clam_views``: ``build()`` -> (outs, job records, max_tiles); refreshed like weight_views."""
    import numpy as np
    ver = (PARAM_EPOCH, tuple(p._version for p in params))
    st = _VIEWS.get(key)
    if st is None:
        if len(_VIEWS) >= 64:
            _VIEWS.clear()
            _MERGED.clear()
        outs, rec, max_tiles = build()
        jobs = np.array(rec, dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("rows", "<i4"), ("cols", "<i4"),
                                             ("tr", "<i4"), ("dt", "<i4")]))
        table = torch.from_numpy(jobs.view(np.uint8).copy()).to(params[0].device)
        st = _VIEWS[key] = dict(outs=outs, table=table, n=len(rec), max_tiles=max_tiles, ver=None, keep=list(params), managed=False,
                                tiles=[((r[2] + 31) // 32) * ((r[3] + 31) // 32) for r in rec])
    st["managed"] = all(is_managed(p) for p in params)
    if st["ver"] != ver or not st["managed"]:
        check(_lib.lib().murcl_cast_batch(ptr(st["table"]), st["n"], st["max_tiles"], stream()), "cast_batch")
        st["ver"] = ver
    return st["outs"]


def clam_views(w1, wa, ba, wb, bb, wc, dtype):
    """Everything CLAM-SB's bf16 chain needs from its parameters, prepared by ONE launch (none while an optimizer that announces its
    steps owns them and nothing changed): -> (w1 [L,d] in ``dtype``; W_il [2D,L] ``dtype``: attention_a / attention_b interleaved in
    16-row blocks (``gate_interleave``); W_il^T [L,2D] ``dtype`` (the dgrad operand); b_il [2D] f32; c_il [2D] f32 = attention_c's
    weight at the a-rows, 0 at the b-rows)."""
    params = (w1, wa, ba, wb, bb, wc)
    D, L = wa.shape
    key = tuple((p.data_ptr(), "clam", i, dtype) for i, p in enumerate(params))

    def build():
        dev = w1.device
        code = _lib.BF16 if dtype == torch.bfloat16 else _lib.F32
        es = 2 if dtype == torch.bfloat16 else 4
        for p in params:
            _need_cuda(p)
            assert p.dtype == torch.float32 and p.is_contiguous()
        w1c = torch.empty(w1.shape, dtype=dtype, device=dev)
        W_il = torch.empty((2 * D, L), dtype=dtype, device=dev)
        W_ilT = torch.empty((L, 2 * D), dtype=dtype, device=dev)
        b_il = torch.empty((2 * D,), dtype=torch.float32, device=dev)
        c_il = torch.zeros((2 * D,), dtype=torch.float32, device=dev)
        rec = []
        R1, C1 = w1.shape
        for r0 in range(0, R1, 32):                  # 32-row strips: every job of the table has about the same number of tiles
            rows = min(32, R1 - r0)
            rec.append((w1.data_ptr() + r0 * C1 * 4, w1c.data_ptr() + r0 * C1 * es, rows, C1, 0, code))
        for g in range(D // 16):
            for j, (w, b) in enumerate(((wa, ba), (wb, bb))):
                r = 32 * g + 16 * j                  # destination row block / column block
                rec.append((w.data_ptr() + 16 * g * L * 4, W_il.data_ptr() + r * L * es, 16, L, 0, code))
                rec.append((w.data_ptr() + 16 * g * L * 4, W_ilT.data_ptr() + r * es, 16, L, 1 | ((2 * D) << 8), code))
                rec.append((b.data_ptr() + 16 * g * 4, b_il.data_ptr() + r * 4, 1, 16, 0, _lib.F32))
            rec.append((wc.data_ptr() + 16 * g * 4, c_il.data_ptr() + 32 * g * 4, 1, 16, 0, _lib.F32))
        max_tiles = max(((rw + 31) // 32) * ((cl + 31) // 32) for _, _, rw, cl, _, _ in rec)
        return [w1c, W_il, W_ilT, b_il, c_il], rec, max_tiles
    return _job_views(key, params, build)


def stacked_views(ws, bs):
    """``torch.stack`` of n same-shaped f32 weights [R,L] and of their biases [R] -> ([n*R, L], [n*R]) f32 by ONE launch (none while an
    optimizer that announces its steps owns them and nothing changed): CLAM's per-class instance classifiers (clam.py:103-132) as the
    one table the instance launch reads."""
    params = tuple(ws) + tuple(bs)
    n = len(ws)
    R, L = ws[0].shape
    key = tuple((p.data_ptr(), "stack", i, n) for i, p in enumerate(params))

    def build():
        dev = ws[0].device
        for p in params:
            _need_cuda(p)
            assert p.dtype == torch.float32 and p.is_contiguous()
        W = torch.empty((n * R, L), dtype=torch.float32, device=dev)
        b = torch.empty((n * R,), dtype=torch.float32, device=dev)
        rec = []
        for i in range(n):
            assert tuple(ws[i].shape) == (R, L) and tuple(bs[i].shape) == (R,)
            rec.append((ws[i].data_ptr(), W.data_ptr() + i * R * L * 4, R, L, 0, _lib.F32))
            rec.append((bs[i].data_ptr(), b.data_ptr() + i * R * 4, 1, R, 0, _lib.F32))
        max_tiles = max(((rw + 31) // 32) * ((cl + 31) // 32) for _, _, rw, cl, _, _ in rec)
        return [W, b], rec, max_tiles
    return _job_views(key, params, build)


_UNIT = {}


def unit_grad(loss):
    """A persistent ones tensor shaped like the scalar ``loss``: ``loss.backward(unit_grad(loss))`` spares autograd the
    fill launch that seeds every backward pass."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    t = _UNIT.get(key)
    if t is None:
        t = _UNIT[key] = torch.ones(loss.shape, dtype=loss.dtype, device=loss.device)
    return t


def is_unit_grad(g):
    """True when ``g`` is (storage-identical to) a tensor handed out by ``unit_grad``: a multiplication by it is a no-op."""
    t = _UNIT.get((g.device, g.dtype, tuple(g.shape)))
    return t is not None and t.data_ptr() == g.data_ptr()


def refresh_views(owned, tick=None):
    """Called by an optimizer right after it rewrote the parameters whose ``data_ptr()`` are in ``owned``: advance
    PARAM_EPOCH and rebuild, in ONE launch, every cached view built from those parameters; views of other optimizers'
    parameters stay valid.  ``tick`` (a device int32 tensor; a captured optimizer step): that launch also advances ``tick[0]``
    (the live step counter of ``adam_multi(..., tick=False)``); returns True when it did - False means nothing was launched and the
    caller has to advance the counter itself (``replay_tick``)."""
    global PARAM_EPOCH
    old = PARAM_EPOCH
    PARAM_EPOCH += 1
    todo = []
    for key, st in _VIEWS.items():
        if not st["managed"] or st["ver"] is None:
            continue
        if any(k[0] in owned for k in key):
            if all(is_managed(p) for p in st["keep"]):
                todo.append((key, st))
        elif st["ver"][0] == old:
            st["ver"] = (PARAM_EPOCH, st["ver"][1])
    if not todo:
        return False
    # ONE launch for all of them: a flat grid over the tiles of every job (the (largest tile count) x (jobs) grid of murcl_cast_batch
    # spent 78 us on 400 k almost all empty workgroups when CLAM's 16-row blocks met the GRU's 3072 x 1024)
    mkey = tuple(k for k, _ in todo)
    merged = _MERGED.get(mkey)
    if merged is None:
        if len(_MERGED) >= 16:
            _MERGED.clear()
        table = torch.cat([st["table"] for _, st in todo]) if len(todo) > 1 else todo[0][1]["table"]
        first, acc = [0], 0
        for _, st in todo:
            for t_ in st["tiles"]:
                acc += t_
                first.append(acc)
        merged = _MERGED[mkey] = (table, torch.tensor(first, dtype=torch.int32).to(table.device), len(first) - 1, acc)
    check(_lib.lib().murcl_cast_batch_flat_tick(ptr(merged[0]), ptr(merged[1]), merged[2], merged[3], ptr(tick), stream()), "cast_batch_flat")
    for _, st in todo:
        st["ver"] = (PARAM_EPOCH, tuple(p._version for p in st["keep"]))
    return tick is not None


_MERGED = {}


def transposed(w, dtype=torch.float32):
    """w [R,C] f32 parameter -> [C,R] in ``dtype``: a cached view for optimizer-managed parameters (rebuilt with all other
    views in the optimizer's one launch per step), a fresh transpose otherwise."""
    if w.dim() == 2 and w.dtype == torch.float32 and w.is_contiguous() and is_managed(w):
        return weight_views(((w, True, dtype),))[0]
    return transpose_cast(w, dtype)


def transpose_cast(w, dtype):
    """w [R,C] f32 -> [C,R] in ``dtype``."""
    _need_cuda(w)
    w = _c(w)
    R, C = w.shape
    y = torch.empty((C, R), dtype=dtype, device=w.device)
    check(_lib.lib().murcl_transpose_cast(ptr(w), ptr(y), R, C, dt(y), stream()), "transpose_cast")
    return y


def colsum(x, out=None, accumulate=False):
    """sum over rows of x [R,N] -> [N] f32."""
    _need_cuda(x)
    x = _c(x)
    R, N = x.shape
    if out is None:
        out = torch.empty((N,), dtype=torch.float32, device=x.device)
    check(_lib.lib().murcl_colsum(ptr(x), ptr(out), R, N, N, dt(x), int(accumulate), stream()), "colsum")
    return out


def relu_bwd(dy, y):
    _need_cuda(dy, y)
    dy, y = _c(dy), _c(y)
    dx = torch.empty_like(dy)
    check(_lib.lib().murcl_relu_bwd(ptr(dy), ptr(y), ptr(dx), dy.numel(), stream()), "relu_bwd")
    return dx


def gru_gates_fwd(gi, gh, hprev, hnew=None, gates=None):
    _need_cuda(gi, gh)
    B, H3 = gi.shape
    H = H3 // 3
    assert gi.is_contiguous() and gh.is_contiguous()
    if hnew is None:
        hnew = torch.empty((B, H), dtype=torch.float32, device=gi.device)
    if gates is None:
        gates = torch.empty((B, H3), dtype=torch.float32, device=gi.device)
    assert hnew.is_contiguous() and gates.is_contiguous()
    check(_lib.lib().murcl_gru_gates_fwd(ptr(gi), ptr(gh), ptr(hprev), ptr(hnew), ptr(gates), B, H,
                                         int(gh.shape[0] == 1 and B != 1), stream()),
          "gru_gates_fwd")
    return hnew, gates


def gru_gates_bwd(dh, gates, gh, hprev, dgi=None, dgh=None):
    _need_cuda(dh, gates, gh)
    dh = _c(dh)
    B, H = dh.shape
    if dgi is None:
        dgi = torch.empty((B, 3 * H), dtype=torch.float32, device=dh.device)
    if dgh is None:
        dgh = torch.empty((B, 3 * H), dtype=torch.float32, device=dh.device)
    assert dgi.is_contiguous() and dgh.is_contiguous() and gates.is_contiguous() and gh.is_contiguous()
    dhp = torch.empty((B, H), dtype=torch.float32, device=dh.device)
    check(_lib.lib().murcl_gru_gates_bwd(ptr(dh), ptr(gates), ptr(gh), ptr(hprev), ptr(dgi), ptr(dgh), ptr(dhp), B, H,
                                         int(gh.shape[0] == 1 and B != 1), stream()), "gru_gates_bwd")
    return dgi, dgh, dhp


def gru_gates_bwd_into(dh, gates, gh, hprev, dgi, dgh, dhprev=None, accumulate=False):
    """Gate backward of one step into caller buffers; dhprev (+)= dh * z (``accumulate``: back-propagation through time, where
    dhprev already holds that step's own upstream gradient)."""
    _need_cuda(dh, gates, gh)
    B, H = dh.shape
    assert all(t.is_contiguous() for t in (dh, gates, gh, dgi, dgh)) and (dhprev is None or dhprev.is_contiguous())
    check(_lib.lib().murcl_gru_gates_bwd_into(ptr(dh), ptr(gates), ptr(gh), ptr(hprev), ptr(dgi), ptr(dgh), ptr(dhprev), B, H,
                                              int(gh.shape[0] == 1 and B != 1), int(accumulate), stream()), "gru_gates_bwd_into")


_TN_SMALL_GROUP = True    # test hook: bag-level f32 weight gradients through the single-writer 32 x 32 kernel


def gru_step_ok(B, H, Kx=0):
    """Whether the one-launch GRU step kernels (``gru_step_fwd`` / ``gru_step_bwd``) take this shape."""
    return bool(_lib.lib().murcl_gru_step_supported(int(B), int(H), int(Kx)))


def gru_step_fwd(gi, hprev, w_hh, b_hh, hnew=None, gates=None, gh=None, x=None, w_ih=None, want_backward=True, want_gh=True):
    """One GRU time step as ONE launch (murcl_gru_step_fwd; ``hprev`` None = zero state, needs ``x``): h W_hh^T for a 16 x 16-unit tile of all three
    gate blocks, the gate math in the epilogue.  gi [B,3H] = x W_ih^T + b_ih - or, with ``x`` [B,Kx] and ``w_ih`` given, gi = b_ih
    [3H] and the input product is formed by the same launch.  -> (hnew [B,H], gates [B,3H], gh [B,3H]) (the last two None
    with ``want_backward=False``, gh None with ``want_gh=False``)."""
    _need_cuda(gi, w_hh)
    (B, H), dev = ((hprev.shape, hprev.device) if hprev is not None else ((x.shape[0], w_hh.shape[1]), x.device))
    if hnew is None:
        hnew = torch.empty((B, H), dtype=torch.float32, device=dev)
    if want_backward:
        gates = torch.empty((B, 3 * H), dtype=torch.float32, device=dev) if gates is None else gates
        if want_gh and gh is None:                   # (from the zero state gh = b_hh for every row: callers keep the bias row instead)
            gh = torch.empty((B, 3 * H), dtype=torch.float32, device=dev)
    assert all(t is None or (t.is_contiguous() and t.dtype == torch.float32) for t in (gi, hprev, w_hh, b_hh, hnew, gates, gh, x, w_ih))
    assert gi.numel() == (3 * H if x is not None else B * 3 * H) and w_hh.shape == (3 * H, H)
    Kx = 0
    if x is not None:
        Kx = x.shape[1]
        assert x.shape[0] == B and w_ih.shape == (3 * H, Kx)
    check(_lib.lib().murcl_gru_step_fwd(ptr(x), ptr(w_ih), Kx, ptr(gi), ptr(hprev), ptr(w_hh), ptr(b_hh), ptr(hnew), ptr(gates),
                                        ptr(gh), B, H, stream()), "gru_step_fwd")
    return hnew, gates, gh


def gru_step_bwd(dgh_next, w_hh_t, dh, gates, gh, hprev, dgi, dgh, dhprev=None, accumulate=False):
    """dh [B,H] += dgh_next [B,3H] . W_hh (``w_hh_t`` = W_hh^T [H,3H]) IN PLACE, then this step's gate backward on the finished
    rows (dgi, dgh; dhprev (+)= dh * z) - one launch (murcl_gru_step_bwd) for ``gemm_nt(accumulate)`` + ``gru_gates_bwd_into``."""
    _need_cuda(dgh_next, dh, gates, gh)
    B, H = dh.shape
    assert all(t is None or (t.is_contiguous() and t.dtype == torch.float32) for t in (dgh_next, w_hh_t, dh, gates, gh, hprev, dgi, dgh, dhprev))
    assert w_hh_t.shape == (H, 3 * H) and dgh_next.shape == (B, 3 * H)
    check(_lib.lib().murcl_gru_step_bwd(ptr(dgh_next), ptr(w_hh_t), ptr(dh), ptr(gates), ptr(gh), ptr(hprev), ptr(dgi), ptr(dgh),
                                        ptr(dhprev), B, H, int(gh.shape[0] == 1 and B != 1), int(accumulate), stream()), "gru_step_bwd")


class _AdamJob(_lib.ctypes.Structure):                 # MurclAdamJob (include/murcl_amd.h)
    _fields_ = [("p", _lib.ctypes.c_void_p), ("g", _lib.ctypes.c_void_p), ("m", _lib.ctypes.c_void_p), ("v", _lib.ctypes.c_void_p),
                ("n", _lib.ctypes.c_long), ("lr", _lib.ctypes.c_float), ("step", _lib.ctypes.c_int)]


ADAM_MAX_JOBS = 8


def replay_tick(replays):
    """Advance the live step counter of a captured optimizer step by one (a one-thread launch; ``adam_multi(..., tick=False)``)."""
    check(_lib.lib().murcl_replay_tick(ptr(replays), stream()), "replay_tick")


def adam_multi(jobs, betas, eps, weight_decay, zero_grad=False, replays=None, tick=True):
    """torch.optim.Adam.step over several flat runs in ONE launch.  ``jobs``: up to ``ADAM_MAX_JOBS`` tuples (p, g, m, v, lr, step) of
    equally long contiguous f32 tensors (each run with its own learning rate and step count).  ``replays``: None, or a device int32[2]
    (zeros) when the launch is being captured into a hipGraph - the step counts then advance on the device with every replay
    (murcl_adam_multi_live)."""
    assert 0 < len(jobs) <= ADAM_MAX_JOBS
    arr = (_AdamJob * len(jobs))()
    for a, (p, g, m, v, lr, step) in zip(arr, jobs):
        _need_cuda(p)
        assert all(t.is_contiguous() and t.dtype == torch.float32 and t.numel() == p.numel() for t in (p, g, m, v))
        a.p, a.g, a.m, a.v, a.n, a.lr, a.step = ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), int(step)
    if replays is not None:
        assert replays.dtype == torch.int32 and replays.numel() >= 2 and replays.is_contiguous()
        # ``tick=False``: the caller advances the counter with its next launch (``refresh_views(tick=)`` / ``replay_tick``)
        fn = _lib.lib().murcl_adam_multi_live if tick else _lib.lib().murcl_adam_multi_live_deferred
        check(fn(_lib.ctypes.addressof(arr), len(jobs), float(betas[0]), float(betas[1]), float(eps),
                 float(weight_decay), int(bool(zero_grad)), ptr(replays), stream()), "adam_multi_live")
        return
    check(_lib.lib().murcl_adam_multi(_lib.ctypes.addressof(arr), len(jobs), float(betas[0]), float(betas[1]), float(eps),
                                      float(weight_decay), int(bool(zero_grad)), stream()), "adam_multi")


def adam_step(p, g, m, v, lr, betas, eps, weight_decay, step, zero_grad=False):
    _need_cuda(p, g, m, v)
    assert p.is_contiguous() and g.is_contiguous() and p.dtype == torch.float32 and g.dtype == torch.float32
    check(_lib.lib().murcl_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), float(betas[0]),
                                     float(betas[1]), float(eps), float(weight_decay), int(step), int(zero_grad),
                                     stream()), "adam_step")


def sgd_step(p, g, buf, lr, momentum, nesterov, weight_decay, first, zero_grad=False):
    _need_cuda(p, g)
    assert p.is_contiguous() and g.is_contiguous() and p.dtype == torch.float32 and g.dtype == torch.float32
    check(_lib.lib().murcl_sgd_step(ptr(p), ptr(g), ptr(buf), p.numel(), float(lr), float(momentum), int(bool(nesterov)),
                                    float(weight_decay), int(bool(first)), int(zero_grad), stream()), "sgd_step")


# ------------------------------------------------------------------------------------------ DSMIL (K6)
def dsmil_argmax(scores_view, B, N, C, want_max=False):
    """scores_view: [B*N, >=C] f32 view (row stride = its stride(0)); -> m [B,C] int32 (, the maxima [B,C] f32 with ``want_max``)."""
    m = torch.empty((B, C), dtype=torch.int32, device=scores_view.device)
    if not want_max:
        check(_lib.lib().murcl_dsmil_argmax(ptr(scores_view), B, N, scores_view.stride(0), C, ptr(m), stream()), "dsmil_argmax")
        return m
    mx = torch.empty((B, C), dtype=torch.float32, device=scores_view.device)
    check(_lib.lib().murcl_dsmil_argmax_max(ptr(scores_view), B, N, scores_view.stride(0), C, ptr(m), ptr(mx), stream()), "dsmil_argmax_max")
    return m, mx


def gather_rows(src, m, B, C, N, col0, width):
    """out[b*C+c,:] = src[b*N + m[b,c], col0:col0+width]  (src [B*N, ld] contiguous)."""
    out = torch.empty((B * C, width), dtype=src.dtype, device=src.device)
    check(_lib.lib().murcl_gather_rows(ptr(src), ptr(m), B, C, N, src.stride(0), col0, width, ptr(out), dt(src), stream()),
          "gather_rows")
    return out


def dsmil_attn(Y, qcol0, qmax, B, N, C):
    A = torch.empty((B, N, C), dtype=torch.float32, device=Y.device)
    check(_lib.lib().murcl_dsmil_attn(ptr(Y), Y.stride(0), qcol0, ptr(qmax), B, N, C, ptr(A), stream()), "dsmil_attn")
    return A


def dsmil_softmax_(S):
    """In-place soft-max over n of S [B,N,C] f32 (already scaled raw scores) -> the same tensor, now A."""
    assert S.is_contiguous() and S.dtype == torch.float32
    B, N, C = S.shape
    check(_lib.lib().murcl_dsmil_softmax(ptr(S), B, N, C, stream()), "dsmil_softmax")
    return S


def dsmil_qv(X, m, wq, bq, B, N, C):
    """One launch: the critical instances' rows x_m [B*C,d] (f32), their queries q = Wq x_m + bq [B*C,128] and v = Wq^T q [B*C,d]."""
    X, wq = _c(X), _c(wq)
    d = X.shape[-1]
    dev = X.device
    xm = torch.empty((B * C, d), dtype=torch.float32, device=dev)
    qmax = torch.empty((B * C, wq.shape[0]), dtype=torch.float32, device=dev)
    v = torch.empty((B * C, d), dtype=torch.float32, device=dev)
    check(_lib.lib().murcl_dsmil_qv(ptr(X), ptr(m), ptr(wq), ptr(_c(bq)), B, N, d, C, ptr(xm), ptr(qmax), ptr(v), dt(X), stream()), "dsmil_qv")
    return xm, qmax, v


def dsmil_qv_bwd(R, qmax, xm, wq, dcmax=None, dwc=None, dbc=None, accumulate=True):
    """-> (dWq [128,d] = qmax^T R + (R Wq^T)^T xm, dbq [128]) in two launches.  With ``dcmax`` [B,C] (the gradient of the
    max-instance class scores) the second launch also adds sum_b dcmax[b,c] xm[b,c] to ``dwc`` [C,d] and sum_b dcmax[b,c] to ``dbc`` [C]
    (``accumulate=False``: overwrites them)."""
    R, qmax, xm, wq = _c(R), _c(qmax), _c(xm), _c(wq)
    BC, d = R.shape
    dq = torch.empty_like(qmax)
    dwq = torch.empty_like(wq)
    dbq = torch.empty((wq.shape[0],), dtype=torch.float32, device=R.device)
    if dcmax is None:
        check(_lib.lib().murcl_dsmil_qv_bwd(ptr(R), ptr(qmax), ptr(xm), ptr(wq), BC, d, ptr(dq), ptr(dwq), ptr(dbq), stream()), "dsmil_qv_bwd")
        return dwq, dbq
    dcmax = _c(dcmax)
    C = dcmax.shape[-1]
    assert dcmax.numel() == BC and dwc.is_contiguous() and dbc.is_contiguous() and tuple(dwc.shape) == (C, d) and dbc.numel() == C
    check(_lib.lib().murcl_dsmil_qv_bwd_cls(ptr(R), ptr(qmax), ptr(xm), ptr(wq), BC, d, ptr(dq), ptr(dwq), ptr(dbq), ptr(dcmax), C,
                                            ptr(dwc), ptr(dbc), int(accumulate), stream()), "dsmil_qv_bwd_cls")
    return dwq, dbq


def dsmil_attn_pool(X, v, scale=1.0):
    """DSMIL's attention and pooling from ONE pass over X: A [B,N,C] = soft-max_n(scale * X . v), Z [B,C,d] = A^T X - or None when
    the shape is not covered (then rows_dot + dsmil_softmax_ + weighted_rowsum).  X [B,N,d] f32/bf16, v [B,C,d] f32."""
    X, v = _c(X), _c(v)
    B, N, d = X.shape
    C = v.shape[1]
    rpw = _lib.lib().murcl_dsmil_stream_plan(B, N, d, C)
    if not rpw:
        return None
    A = torch.empty((B, N, C), dtype=torch.float32, device=X.device)
    Z = torch.empty((B, C, d), dtype=torch.float32, device=X.device)
    ws = torch.empty(((B * N // rpw) * C * (d + 2) + 2 * B * C,), dtype=torch.float32, device=X.device)
    with _span(lambda: (f"dsmil_attn_pool<{_DT_NAME[X.dtype]}>", dict(bytes=X.numel() * X.element_size(), flops=4.0 * B * N * d * C))):
        check(_lib.lib().murcl_dsmil_attn_pool(ptr(X), ptr(v), float(scale), ptr(A), ptr(Z), ptr(ws), B, N, d, C, dt(X), stream()),
              "dsmil_attn_pool")
    return A, Z


def dsmil_attn_pool_bwd(X, dZ, A, Z, dcls=None, scale=1.0):
    """One pass over X for the backward of (attention, pooling): R [B,C,d] = scale * sum_n dS[n,c] X[n] with dS = A (dA - sum A dA),
    dA = X dZ^T, neither stored; with ``dcls`` [B,N,C] also dWc [C,d] = dcls^T X.  None when the shape is not covered."""
    X, dZ, A, Z = _c(X), _c(dZ), _c(A), _c(Z)
    B, N, d = X.shape
    C = dZ.shape[1]
    rpw = _lib.lib().murcl_dsmil_stream_plan(B, N, d, C)
    if not rpw:
        return None
    W = B * N // rpw
    R = torch.empty((B, C, d), dtype=torch.float32, device=X.device)
    ws = torch.empty((W * C * (d + 2),), dtype=torch.float32, device=X.device)
    gpart = torch.empty((W, C * d), dtype=torch.float32, device=X.device) if dcls is not None else None
    if dcls is not None:
        dcls = _c(dcls)
    with _span(lambda: (f"dsmil_attn_pool_bwd<{_DT_NAME[X.dtype]}>", dict(bytes=X.numel() * X.element_size(), flops=(4.0 if dcls is None else 6.0) * B * N * d * C))):
        check(_lib.lib().murcl_dsmil_attn_pool_bwd(ptr(X), ptr(dZ), ptr(A), ptr(Z), ptr(dcls), float(scale), ptr(R), ptr(gpart), ptr(ws),
                                                   B, N, d, C, dt(X), stream()), "dsmil_attn_pool_bwd")
    return R, (colsum(gpart).view(C, d) if dcls is not None else None)


def dsmil_softmax_bwd(A, dA):
    """dS = A * (dA - sum_n A dA) per (bag, class); A, dA [B,N,C] f32."""
    A, dA = _c(A), _c(dA)
    B, N, C = A.shape
    dS = torch.empty_like(A)
    dots = torch.empty((B * C,), dtype=torch.float32, device=A.device)
    check(_lib.lib().murcl_dsmil_softmax_bwd(ptr(A), ptr(dA), B, N, C, ptr(dS), ptr(dots), stream()), "dsmil_softmax_bwd")
    return dS


def weighted_rowsum(X, A, into=None):
    """Z[b,c,:] = sum_n A[b,n,c] X[b,n,:]   X [B,N,d] (f32/bf16), A [B,N,C] f32 -> Z [B,C,d] f32.  ``into``: a [B,C,d] f32 tensor the
    sums are ADDED to (cleared by the caller: ``murcl_weighted_rowsum_acc``, no fill launch)."""
    X, A = _c(X), _c(A)
    B, N, d = X.shape
    C = A.shape[2]
    if into is not None:
        assert into.is_contiguous() and into.dtype == torch.float32 and into.numel() == B * C * d
        with _span(lambda: (f"weighted_rowsum<{_DT_NAME[X.dtype]}>", dict(bytes=X.numel() * X.element_size(), flops=2.0 * B * N * d * C))):
            check(_lib.lib().murcl_weighted_rowsum_acc(ptr(X), ptr(A), ptr(into), B, N, d, C, dt(X), stream()), "weighted_rowsum_acc")
        return into.view(B, C, d)
    if B == 1 and N >= 1 << 16 and N % 64 == 0:
        # one long "bag" (a weight gradient over all patches): the kernel's row splits all add atomically into the same
        # C*d addresses - 1024 adders per address serialise at the memory side (250 us for a 537 MB pass).  Cut the rows
        # into 64 pseudo-bags (16 adders per address) and sum their rows afterwards.
        Zp = weighted_rowsum(X.view(64, N // 64, d), A.view(64, N // 64, C))
        return colsum(Zp.view(64, C * d)).view(1, C, d)
    Z = torch.empty((B, C, d), dtype=torch.float32, device=X.device)
    with _span(lambda: (f"weighted_rowsum<{_DT_NAME[X.dtype]}>", dict(bytes=X.numel() * X.element_size(), flops=2.0 * B * N * d * C))):
        check(_lib.lib().murcl_weighted_rowsum(ptr(X), ptr(A), ptr(Z), B, N, d, C, dt(X), stream()), "weighted_rowsum")
    return Z


def rows_dot(X, V, bias=None):
    """out[b,n,c] = X[b,n,:] . V[b,c,:] (+ bias[c])."""
    X, V = _c(X), _c(V)
    B, N, d = X.shape
    C = V.shape[1]
    out = torch.empty((B, N, C), dtype=torch.float32, device=X.device)
    if bias is not None:
        bias = _c(bias)
        assert bias.dtype == torch.float32 and bias.numel() == C
    with _span(lambda: (f"rows_dot<{_DT_NAME[X.dtype]}>", dict(bytes=X.numel() * X.element_size(), flops=2.0 * B * N * d * C))):
        check(_lib.lib().murcl_rows_dot_bias(ptr(X), ptr(V), ptr(bias), ptr(out), B, N, d, C, dt(X), stream()), "rows_dot")
    return out


def rows_dot_wsum(X, V, G):
    """(out[b,n,c] = X[b,n,:] . V[b,c,:],  W[c,:] = sum_{b,n} G[b,n,c] X[b,n,:]) from ONE pass over X, or None when the
    shape is not covered (then: rows_dot + weighted_rowsum).  X [B,N,d] (f32/bf16), V [B,C,d] f32, G [B,N,C] f32."""
    X, V, G = _c(X), _c(V), _c(G)
    B, N, d = X.shape
    C = V.shape[1]
    rpw = _lib.lib().murcl_rows_dot_wsum_plan(B, N, d, C)
    if not rpw:
        return None
    out = torch.empty((B, N, C), dtype=torch.float32, device=X.device)
    part = torch.empty((B * N // rpw, C * d), dtype=torch.float32, device=X.device)
    with _span(lambda: (f"rows_dot_wsum<{_DT_NAME[X.dtype]}>", dict(bytes=X.numel() * X.element_size(), flops=4.0 * B * N * d * C))):
        check(_lib.lib().murcl_rows_dot_wsum(ptr(X), ptr(V), ptr(G), ptr(out), ptr(part), B, N, d, C, dt(X), stream()),
              "rows_dot_wsum")
    return out, colsum(part).view(C, d)


def dsmil_attn_bwd(A, dA, Y, qcol0, qmax, dY, B, N, C):
    dqmax = torch.empty((B * C, qmax.shape[1]), dtype=torch.float32, device=Y.device)
    dots = torch.empty((B * C,), dtype=torch.float32, device=Y.device)
    check(_lib.lib().murcl_dsmil_attn_bwd(ptr(A), ptr(_c(dA)), ptr(Y), Y.stride(0), qcol0, ptr(qmax), B, N, C, ptr(dY),
                                          dY.stride(0), ptr(dqmax), ptr(dots), stream()), "dsmil_attn_bwd")
    return dqmax


# ------------------------------------------------------------------------------------------ CLAM-SB (K4/K5)
def gated_score_fwd(U, wc, bc, keep_a=None, keep_b=None, gated=True):
    """gated: U [M,2D] -> s [M] f32: sum_d tanh(U[:, :D]) * sigmoid(U[:, D:]) * wc + bc (Attn_Net_Gated);
    not gated: U [M,D] -> sum_d tanh(U[:, d]) * wc + bc (Attn_Net, clam.py:18-34)."""
    U = _c(U)
    M, W = U.shape
    D = W // 2 if gated else W
    s = torch.empty((M,), dtype=torch.float32, device=U.device)
    ka, kb, kp, sa, sb = _gate_drops(keep_a, keep_b)
    check(_lib.lib().murcl_gated_score_fwd(ptr(U), ptr(wc), ptr(bc), ptr(ka), ptr(kb), ptr(s), M, D, dt(U), int(gated),
                                           kp, sa, sb, stream()), "gated_score_fwd")
    return s


def _gate_drops(keep_a, keep_b):
    """keep_a / keep_b: materialised masks (tensors) or DropSeed specs (masks generated inside the kernels) -> kernel arguments."""
    if isinstance(keep_a, DropSeed):
        assert keep_b is None or isinstance(keep_b, DropSeed)
        return None, None, keep_a.keep_p, keep_a.seed, keep_b.seed if keep_b is not None else 0
    return keep_a, keep_b, 0.0, 0, 0


def gated_score_bwd(U, wc, ds, keep_a=None, keep_b=None, gated=True):
    """-> dU (shape of U), dwc [D], dbc [1], column sums of dU [U.shape[1]] (the bias gradients of the gate Linears)."""
    U, ds = _c(U), _c(ds)
    M, W = U.shape
    D = W // 2 if gated else W
    dU = torch.empty_like(U)
    dwc = torch.empty((D,), dtype=torch.float32, device=U.device)
    dbc = torch.empty((1,), dtype=torch.float32, device=U.device)
    dbab = torch.empty((2 * D,), dtype=torch.float32, device=U.device)                    # column sums of dU, same pass
    part = torch.empty((1024 * (3 * D + 1),), dtype=torch.float32, device=U.device)       # per-workgroup partial rows
    ka, kb, kp, sa, sb = _gate_drops(keep_a, keep_b)
    check(_lib.lib().murcl_gated_score_bwd(ptr(U), ptr(wc), ptr(ka), ptr(kb), ptr(ds), ptr(dU), ptr(dwc), ptr(dbc),
                                           ptr(dbab), ptr(part), M, D, dt(U), int(gated), kp, sa, sb, stream()), "gated_score_bwd")
    return dU, dwc, dbc, dbab[:W]


def softmax_rows(s):
    s = _c(s)
    A = torch.empty_like(s)
    check(_lib.lib().murcl_softmax_rows(ptr(s), ptr(A), s.shape[0], s.shape[1], stream()), "softmax_rows")
    return A


def softmax_rows_parts(part, B, N, zero=None):
    """part [P, B*N] partial score rows -> (s [B,N] = their column sum, A [B,N] = soft-max over N): one launch.  ``zero``: a [B, n] f32
    tensor cleared by the same launch (the pooled rows ``weighted_rowsum(..., into=zero)`` then adds into: no fill launch)."""
    part = _c(part)
    assert part.dtype == torch.float32 and part.shape[1] == B * N
    assert zero is None or (zero.is_contiguous() and zero.dtype == torch.float32 and zero.shape[0] == B)
    s = torch.empty((B, N), dtype=torch.float32, device=part.device)
    A = torch.empty_like(s)
    check(_lib.lib().murcl_softmax_rows_parts(ptr(part), part.shape[0], ptr(s), ptr(A), B, N, ptr(zero), 0 if zero is None else zero.numel() // B,
                                              stream()), "softmax_rows_parts")
    return s, A


def softmax_rows_bwd(A, dA):
    A, dA = _c(A), _c(dA)
    ds = torch.empty_like(A)
    check(_lib.lib().murcl_softmax_rows_bwd(ptr(A), ptr(dA), ptr(ds), A.shape[0], A.shape[1], stream()), "softmax_rows_bwd")
    return ds


def topk_ids(A, k):
    """A [B,N] -> ids [B,2k] int32: top-k (descending) then bottom-k (ascending); lowest index wins ties."""
    A = _c(A)
    ids = torch.empty((A.shape[0], 2 * k), dtype=torch.int32, device=A.device)
    check(_lib.lib().murcl_topk_ids(ptr(A), A.shape[0], A.shape[1], k, ptr(ids), stream()), "topk_ids")
    return ids


def clam_inst_fwd(h, ids, labels, W, bias, B, N, k, n_cls, subtyping):
    """CLAM's instance branch for all (bag, class) pairs, one launch (``murcl_clam_inst_fwd``) -> (loss [B] f32, dl [B*2k, 2 n_cls],
    pt [2,B,n_cls,2k] int64 = predictions / targets).  h [B*N,L], ids [B,2k] int32, labels [B] int64, W [2 n_cls, L], bias [2 n_cls]."""
    h, W, bias = _c(h), _c(W), _c(bias)
    L = h.shape[1]
    dev = h.device
    loss = torch.empty((B,), dtype=torch.float32, device=dev)
    dl = torch.empty((B * 2 * k, 2 * n_cls), dtype=torch.float32, device=dev)
    pt = torch.empty((2, B, n_cls, 2 * k), dtype=torch.int64, device=dev)
    scale = 1.0 / n_cls if subtyping else 1.0
    check(_lib.lib().murcl_clam_inst_fwd(ptr(h), ptr(ids), ptr(labels), ptr(W), ptr(bias), B, N, L, k, n_cls, int(bool(subtyping)), scale,
                                         ptr(loss), ptr(dl), ptr(pt), dt(h), stream()), "clam_inst_fwd")
    return loss, dl, pt


def clam_inst_bwd(h, ids, W, dl, up, B, N, k, n_cls, dz):
    """Backward of ``clam_inst_fwd``: the feature gradients are ADDED into dz [B*N,L] (in place, under h > 0) and the sums over bags of
    ``part`` are returned: (dW [2 n_cls, L], db [2 n_cls], column sums of what was added to dz [L])."""
    h, W = _c(h), _c(W)
    L = h.shape[1]
    O = 2 * n_cls
    assert dz.is_contiguous() and dz.dtype == h.dtype
    part = torch.empty((B, O * (L + 1) + L), dtype=torch.float32, device=h.device)
    check(_lib.lib().murcl_clam_inst_bwd(ptr(h), ptr(ids), ptr(W), ptr(_c(dl)), ptr(_c(up)), B, N, L, k, n_cls, ptr(dz), ptr(part), dt(h),
                                         stream()), "clam_inst_bwd")
    sums = colsum(part)
    return sums[:O * L].view(O, L), sums[O * L:O * L + O], sums[O * (L + 1):]


def take_rows(src, rows):
    """src [R0,d] (f32/bf16), rows int64 [R] -> f32 [R,d]."""
    src, rows = _c(src), _c(rows)
    out = torch.empty((rows.numel(), src.shape[1]), dtype=torch.float32, device=src.device)
    check(_lib.lib().murcl_take_rows(ptr(src), ptr(rows), ptr(out), rows.numel(), src.shape[1], dt(src), stream()), "take_rows")
    return out


def scatter_add_rows_masked(dst, h, rows, g, write_back=False):
    """dst[rows[r]] += g[r] * (h[rows[r]] > 0).  ``write_back``: g (f32, contiguous) is masked in place to what was added."""
    assert g.is_contiguous() and g.dtype == torch.float32
    check(_lib.lib().murcl_scatter_add_rows_masked(ptr(dst), ptr(h), ptr(_c(rows)), ptr(g), rows.numel(), dst.shape[1],
                                                   dt(dst), int(write_back), stream()), "scatter_add_rows_masked")


def cross_entropy(logits, targets, group, want_conf=False):
    """Mean CE per group of `group` consecutive rows -> (loss [R/group], dlogits [R,C] (already / group), preds [R]); with
    ``want_conf`` a fourth result conf [R]: the soft-max probability of each row's target class."""
    logits, targets = _c(logits), _c(targets)
    R, C = logits.shape
    G = R // group
    loss = torch.empty((G,), dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits)
    preds = torch.empty((R,), dtype=torch.int64, device=logits.device)
    conf = torch.empty((R,), dtype=torch.float32, device=logits.device) if want_conf else None
    check(_lib.lib().murcl_cross_entropy(ptr(logits), ptr(targets), R, C, ptr(loss), ptr(dl), ptr(preds), ptr(conf), group, stream()),
          "cross_entropy")
    return (loss, dl, preds, conf) if want_conf else (loss, dl, preds)


_DROP_COUNTER = 0


class DropSeed:
    """A dropout keep mask that is never materialised: the (seed, keep probability) its elements are a pure function of."""
    __slots__ = ("seed", "keep_p")

    def __init__(self, keep_p, seed=None):
        self.keep_p, self.seed = float(keep_p), dropout_seed() if seed is None else int(seed)

    @property
    def keep_q(self):
        """The keep probability the kernels REALISE: one byte per element, so ``keep_p`` rounded to 1/256.  Survivors are scaled
        by 1/keep_q (exactly 1/keep_p for CLAM's 0.75; for e.g. 0.9 -> 230/256 scaling by 1/0.9 would bias the mean by 0.2 %)."""
        return max(1, int(self.keep_p * 256.0 + 0.5)) / 256.0


def dropout_seed():
    """Seed of the next counter-based dropout mask: torch's global seed and a call counter (no device synchronisation;
    ``torch.manual_seed`` makes a run reproducible)."""
    global _DROP_COUNTER
    _DROP_COUNTER += 1
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _DROP_COUNTER * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF


DRAWS_MAX_B = 2048


def step_draws(device, n_uni=0, n_nrm=0, n_views=0, B=0, alpha=0.0, seed=None):
    """Every random draw of a training step in ONE launch (``murcl_step_draws``): -> (uni [n_uni] f32 ~ U[0,1), nrm [n_nrm] f32 ~ N(0,1),
    lam [n_views,B] f32 = alpha + U(0,1)(1 - alpha), perm [n_views,B] int32: a uniform random permutation per view).  Counter-based; the
    seed is drawn from torch's CPU generator (no device work; ``torch.manual_seed`` makes a run reproducible)."""
    assert device.type == "cuda" and (n_views == 0 or 0 < B <= DRAWS_MAX_B)
    uni = torch.empty((n_uni,), dtype=torch.float32, device=device)
    nrm = torch.empty((n_nrm,), dtype=torch.float32, device=device)
    lam = torch.empty((n_views, B), dtype=torch.float32, device=device)
    perm = torch.empty((n_views, B), dtype=torch.int32, device=device)
    if seed is None:                                 # 63 bits off torch's CPU generator: host-only, reproducible under torch.manual_seed
        seed = int(torch.empty((), dtype=torch.int64).random_())
    check(_lib.lib().murcl_step_draws(int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(uni), n_uni, ptr(nrm), n_nrm, ptr(lam), ptr(perm),
                                      n_views, B, float(alpha), stream()), "step_draws")
    return uni, nrm, lam, perm


def dropout_relu_bitmask(x, drop, want_bits=True):
    """x [M,N] (contiguous, M % 32 == 0, N % 128 == 0): ``x *= keep`` in place for the mask of ``drop`` (a DropSeed) and, in the
    same pass, the panel GEMM's 1-bit mask of x > 0 afterwards -> bits [M, N/8] uint8 (None without ``want_bits``)."""
    _need_cuda(x)
    assert x.is_contiguous() and x.dim() == 2
    M, N = x.shape
    bits = torch.empty((M, N // 8), dtype=torch.uint8, device=x.device) if want_bits else None
    check(_lib.lib().murcl_dropout_relu_bitmask(ptr(x), ptr(bits), M, N, drop.keep_p, 1.0 / drop.keep_q, drop.seed, dt(x), stream()),
          "dropout_relu_bitmask")
    return bits


def dropout_mask(shape, dtype, keep_p, device, seed=None):
    """Keep mask of ``nn.Dropout(1 - keep_p)``: ``1/keep_p`` where kept, 0 where dropped.  Seeded from torch's global
    seed and a call counter (no device synchronisation; ``torch.manual_seed`` makes a run reproducible)."""
    seed = dropout_seed() if seed is None else int(seed)
    out = torch.empty(shape, dtype=dtype, device=device)
    check(_lib.lib().murcl_dropout_mask(ptr(out), out.numel(), float(keep_p), 1.0 / DropSeed(keep_p, seed).keep_q, seed, dt(out), stream()),
          "dropout_mask")
    return out


def mul(x, k, out=None):
    x, k = _c(x), _c(k)
    out = x if out is None else out
    check(_lib.lib().murcl_mul(ptr(x), ptr(k), ptr(out), x.numel(), dt(x), stream()), "mul")
    return out


def pad_cols(x, Cp):
    """[R,C] -> [R,Cp] = [x | 0] (contiguous f32 / bf16), one launch (murcl_pad_cols)."""
    _need_cuda(x)
    x = _c(x)
    R, C = x.shape
    assert Cp >= C and x.element_size() in (2, 4)
    out = torch.empty((R, Cp), dtype=x.dtype, device=x.device)
    check(_lib.lib().murcl_pad_cols(ptr(x), ptr(out), R, C, Cp, x.element_size(), stream()), "pad_cols")
    return out


def axpby(x, y, a, b, out=None):
    """a x + b y for equally shaped contiguous f32 tensors, one launch (murcl_axpby); ``out`` may be x or y (elementwise)."""
    _need_cuda(x, y)
    assert x.shape == y.shape and x.dtype == torch.float32 and y.dtype == torch.float32 and x.is_contiguous() and y.is_contiguous()
    out = torch.empty_like(x) if out is None else out
    assert out.shape == x.shape and out.dtype == torch.float32 and out.is_contiguous()
    check(_lib.lib().murcl_axpby(ptr(x), ptr(y), float(a), float(b), ptr(out), x.numel(), stream()), "axpby")
    return out


def mean_small(x):
    """Mean of a small contiguous f32 tensor -> a 0-dim tensor, one launch in a fixed order (murcl_mean_small); not differentiable."""
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() > 0
    out = torch.empty((), dtype=torch.float32, device=x.device)
    check(_lib.lib().murcl_mean_small(ptr(x), x.numel(), ptr(out), stream()), "mean_small")
    return out


def group_mean(x, groups, group):
    """x [groups * group] contiguous f32 -> [groups]: the mean of every run of ``group`` values, one launch (murcl_group_mean)."""
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() == groups * group
    out = torch.empty((groups,), dtype=torch.float32, device=x.device)
    check(_lib.lib().murcl_group_mean(ptr(x), int(groups), int(group), ptr(out), stream()), "group_mean")
    return out


_ONES = {}


def filled(shape, value, device):
    """A FRESH f32 tensor of ``shape`` filled with ``value`` by an own launch (a scaling of a cached, never handed-out ones tensor)."""
    import math as _m
    n = _m.prod(shape)
    key = (torch.device(device), n)
    one = _ONES.get(key)
    if one is None:
        one = _ONES[key] = torch.ones((n,), dtype=torch.float32, device=device)
    return axpby(one, one, float(value), 0.0).view(shape)


def copy_flat(dst, src):
    """dst <- src for two equally long contiguous buffers with 16-byte aligned bases (flat parameter buffers) as ONE launch of this
    library (murcl_copy_bytes; no runtime blit in the step's launch sequence); anything else goes through ``copy_``."""
    if (src.is_cuda and dst.is_cuda and src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype and src.numel() == dst.numel()
            and src.data_ptr() % 16 == 0 and dst.data_ptr() % 16 == 0):
        check(_lib.lib().murcl_copy_bytes(ptr(src), ptr(dst), src.numel() * src.element_size(), stream()), "copy_bytes")
        return dst
    return dst.copy_(src)


# ------------------------------------------------------------------------------------------ PPO (K10/K11)
def policy_head_fwd(z, std, eps=None, actions=None):
    """z [R,K] -> (mu [R,K], action [R,K], logp [R]); sample with eps or evaluate given actions."""
    z = _c(z)
    R, K = z.shape
    mu = torch.empty_like(z)
    logp = torch.empty((R,), dtype=torch.float32, device=z.device)
    act = torch.empty_like(z) if eps is not None else _c(actions)
    check(_lib.lib().murcl_policy_head_fwd(ptr(z), ptr(_c(eps)) if eps is not None else None,
                                           None if eps is not None else ptr(act), float(std), R, K, ptr(mu),
                                           ptr(act) if eps is not None else None, ptr(logp), stream()), "policy_head_fwd")
    return mu, act, logp


def policy_head_bwd(mu, act, dlogp, std):
    dz = torch.empty_like(mu)
    check(_lib.lib().murcl_policy_head_bwd(ptr(mu), ptr(_c(act)), ptr(_c(dlogp)), float(std), mu.shape[0], mu.shape[1],
                                           ptr(dz), stream()), "policy_head_bwd")
    return dz


def stack_rows(ts):
    """[n_i, ...] tensors -> their rows stacked; free (a view) when they already are consecutive blocks of one buffer (halves
    of a batched output, slices of one noise draw), one ``torch.cat`` otherwise."""
    t0 = ts[0]
    if all(t.is_contiguous() and t.dtype == t0.dtype and t.shape[1:] == t0.shape[1:]
           and t.untyped_storage().data_ptr() == t0.untyped_storage().data_ptr() for t in ts):
        off, ok = t0.storage_offset(), True
        for t in ts:
            ok, off = ok and t.storage_offset() == off, off + t.numel()
        if ok:
            return t0.as_strided((sum(t.shape[0] for t in ts),) + tuple(t0.shape[1:]), t0.stride(), t0.storage_offset())
    return torch.cat(ts, 0)


class _CopyJob(ctypes.Structure):                      # MurclCopyJob (include/murcl_amd.h)
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("bytes", ctypes.c_long)]


STACK_MAX_JOBS = 96


def stack_lists(lists):
    """``[torch.stack(l, 0) for l in lists]`` in ONE launch (``murcl_stack_lists``): every list holds equally shaped contiguous CUDA
    tensors of a 4- or 8-byte dtype; anything else (or more than ``STACK_MAX_JOBS`` tensors in all) goes through ``torch.stack``."""
    flat = [t for l in lists for t in l]
    if (not flat or len(flat) > STACK_MAX_JOBS or not all(t.is_cuda and t.is_contiguous() and t.element_size() in (4, 8) for t in flat)
            or any(t.shape != l[0].shape or t.dtype != l[0].dtype for l in lists for t in l)):
        return [torch.stack(l, 0) for l in lists]
    outs = [torch.empty((len(l),) + tuple(l[0].shape), dtype=l[0].dtype, device=l[0].device) for l in lists]
    arr, i = (_CopyJob * len(flat))(), 0
    for l, o in zip(lists, outs):
        nb = l[0].numel() * l[0].element_size()
        for k, t in enumerate(l):
            arr[i].src, arr[i].dst, arr[i].bytes = t.data_ptr(), o.data_ptr() + k * nb, nb
            i += 1
    check(_lib.lib().murcl_stack_lists(ctypes.addressof(arr), len(flat), stream()), "stack_lists")
    return outs


def add_lists(pairs):
    """``dst += src`` for every (src, dst) pair of equally sized contiguous f32 CUDA tensors in ONE launch (murcl_add_lists)."""
    pairs = [(s_, d_) for s_, d_ in pairs if s_ is not None]
    if not pairs:
        return
    assert len(pairs) <= STACK_MAX_JOBS
    arr = (_CopyJob * len(pairs))()
    keep = []
    for i, (s_, d_) in enumerate(pairs):
        s_ = _c(s_)
        assert (s_.is_cuda and d_.is_cuda and s_.dtype == torch.float32 and d_.dtype == torch.float32 and d_.is_contiguous()
                and s_.numel() == d_.numel())
        keep.append(s_)
        arr[i].src, arr[i].dst, arr[i].bytes = s_.data_ptr(), d_.data_ptr(), s_.numel() * 4
    check(_lib.lib().murcl_add_lists(ctypes.addressof(arr), len(pairs), stream()), "add_lists")


def pointer_table(tensors):
    """Host array of device pointers (the ``const float* const*`` arguments of the sequence entry points)."""
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def ppo_act(ptable, S, H, K, state, hidden_prev, eps, std):
    """One sampling step of the policy behind ONE native call (murcl_ppo_act: 6 launches).
    state [B,S] f32, hidden_prev [B,H] or None (zeros), eps [B,K] -> (hidden_new [B,H], action [B,K], logp [B])."""
    _need_cuda(state, eps)
    state, eps = _c(state), _c(eps)
    B, dev = state.shape[0], state.device
    hnew = torch.empty((B, H), dtype=torch.float32, device=dev)
    action = torch.empty((B, K), dtype=torch.float32, device=dev)
    logp = torch.empty((B,), dtype=torch.float32, device=dev)
    ws = torch.empty((_lib.lib().murcl_ppo_act_workspace(B, S, H) // 4,), dtype=torch.float32, device=dev)
    check(_lib.lib().murcl_ppo_act(ptable, S, H, K, ptr(state), ptr(_c(hidden_prev)) if hidden_prev is not None else None,
                                   ptr(eps), float(std), B, ptr(hnew), ptr(action), ptr(logp), ptr(ws), stream()), "ppo_act")
    return hnew, action, logp


def ppo_epoch(ptable, gtable, S, H, K, states, actions, old_logp, returns, n_total, std, eps_clip, entropy, want_loss=False, wt=None):
    """One K_epoch of PPO.update minus the optimizer step behind ONE native call (murcl_ppo_epoch): evaluate() forward,
    loss, backward; parameter gradients are ADDED into the tensors behind ``gtable``.  ``wt``: (W_ih^T, W_hh^T, W_2^T) f32,
    e.g. ``weight_views`` of the three parameters - without it the call transposes them itself (three launches)."""
    _need_cuda(states, actions)
    states, actions, old_logp, returns = _c(states.float()), _c(actions.float()), _c(old_logp), _c(returns)
    T_, B = states.shape[0], states.shape[1]
    dev = states.device
    ws = torch.empty((_lib.lib().murcl_ppo_epoch_workspace(T_, B, S, H) // 4,), dtype=torch.float32, device=dev)
    loss = torch.empty((1,), dtype=torch.float32, device=dev) if want_loss else None
    if wt is not None:
        assert len(wt) == 3 and all(t.is_contiguous() and t.dtype == torch.float32 for t in wt)
        assert wt[0].shape == (H, 3 * H) and wt[1].shape == (H, 3 * H) and wt[2].shape[1] == H
        check(_lib.lib().murcl_ppo_epoch_wt(ptable, gtable, pointer_table(wt), S, H, K, ptr(states), ptr(actions), ptr(old_logp),
                                            ptr(returns), T_, B, int(n_total), float(std), float(eps_clip), float(entropy), ptr(ws),
                                            ptr(loss), stream()), "ppo_epoch_wt")
        return loss
    check(_lib.lib().murcl_ppo_epoch(ptable, gtable, S, H, K, ptr(states), ptr(actions), ptr(old_logp), ptr(returns), T_, B,
                                     int(n_total), float(std), float(eps_clip), float(entropy), ptr(ws), ptr(loss), stream()),
          "ppo_epoch")
    return loss


def ppo_returns(rewards, gamma):
    """rewards [T,B] f32 -> normalised discounted returns [T,B]."""
    rewards = _c(rewards.float())
    ret = torch.empty_like(rewards)
    check(_lib.lib().murcl_ppo_returns(ptr(rewards), float(gamma), rewards.shape[0], rewards.shape[1], ptr(ret), stream()),
          "ppo_returns")
    return ret


def ppo_returns_raw(rewards, gamma):
    """rewards [T,B] f32 -> (raw discounted returns [T,B], stats f64 [2] = local (sum, sum of squares))."""
    rewards = _c(rewards.float())
    ret = torch.empty_like(rewards)
    stats = torch.empty((2,), dtype=torch.float64, device=rewards.device)
    check(_lib.lib().murcl_ppo_returns_raw(ptr(rewards), float(gamma), rewards.shape[0], rewards.shape[1], ptr(ret),
                                           ptr(stats), stream()), "ppo_returns_raw")
    return ret, stats


def ppo_returns_finish(ret, stats, n_total):
    """Normalise raw returns in place with the (all-reduced) sum / sum of squares over ``n_total`` returns."""
    check(_lib.lib().murcl_ppo_returns_finish(ptr(ret), ret.numel(), ptr(stats), int(n_total), stream()), "ppo_returns_finish")
    return ret


def ppo_loss(logp, old_logp, value, ret, eps_clip, entropy, n_total=None):
    n = logp.numel()
    loss = torch.empty((1,), dtype=torch.float32, device=logp.device)
    dlogp, dvalue = torch.empty_like(logp), torch.empty_like(value)
    check(_lib.lib().murcl_ppo_loss(ptr(_c(logp)), ptr(_c(old_logp)), ptr(_c(value)), ptr(_c(ret)), float(eps_clip),
                                    float(entropy), n, int(n if n_total is None else n_total), ptr(loss), ptr(dlogp),
                                    ptr(dvalue), stream()), "ppo_loss")
    return loss, dlogp, dvalue
