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


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("murcl_amd kernels run on the GPU only (got a CPU tensor); there is no CPU fallback")


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def gemm_nt(A, B, *, epi=EPI_NONE, bias=None, mask=None, rowscale=None, rank1=None, rows_per_bag=0,
            out_dtype=None, colsum=False, out=None, accumulate=False):
    """C[M,N] = epi(A[M,K] @ B[N,K]^T).  Returns C or (C, colsum_ws[ceil(M/128),N])."""
    _need_cuda(A, B)
    A, B = _c(A), _c(B)
    M, K = A.shape
    N = B.shape[0]
    assert B.shape[1] == K and A.dtype == B.dtype
    odt = out_dtype or A.dtype
    C = out if out is not None else torch.empty((M, N), dtype=odt, device=A.device)
    ws = torch.empty(((M + 127) // 128, N), dtype=torch.float32, device=A.device) if colsum else None
    if mask is not None:
        mask = _c(mask)
        assert mask.dtype == A.dtype and mask.shape == (M, N)
    check(_lib.lib().murcl_gemm_nt(ptr(A), ptr(B), ptr(C), M, N, K, K, K, N, dt(A), dt(C), epi, ptr(bias), ptr(mask),
                                   N, ptr(rowscale), ptr(rank1), rows_per_bag, ptr(ws), int(accumulate), stream()),
          "gemm_nt")
    return (C, ws) if colsum else C


def gemm_tn(A, B, *, splits=0, out=None):
    """C[N1,N2] (f32) = A[M,N1]^T @ B[M,N2]  (adds into ``out`` when given)."""
    _need_cuda(A, B)
    A, B = _c(A), _c(B)
    M, N1 = A.shape
    N2 = B.shape[1]
    assert B.shape[0] == M and A.dtype == B.dtype
    C = out if out is not None else torch.zeros((N1, N2), dtype=torch.float32, device=A.device)
    check(_lib.lib().murcl_gemm_tn(ptr(A), ptr(B), ptr(C), M, N1, N2, N1, N2, N2, dt(A), splits, stream()), "gemm_tn")
    return C


def pool_chunks(B, N, dtype_code):
    cr, nc = ctypes.c_int(), ctypes.c_int()
    _lib.lib().murcl_abmil_pool_workspace(B, N, dtype_code, ctypes.byref(cr), ctypes.byref(nc))
    return cr.value, nc.value


def abmil_pool_fwd(H, Wa, ba, wb, bb, exact_tanh=None):
    """H [B,N,512], Wa [128,512] (same dtype) -> scores [B,N], A [B,N], M [B,512], ml [B,2] (all f32)."""
    _need_cuda(H, Wa)
    H, Wa = _c(H), _c(Wa)
    B, N, L = H.shape
    D = Wa.shape[0]
    if exact_tanh is None:
        exact_tanh = H.dtype == torch.float32
    dev = H.device
    _, S = pool_chunks(B, N, dt(H))
    scores = torch.empty((B, N), dtype=torch.float32, device=dev)
    A = torch.empty((B, N), dtype=torch.float32, device=dev)
    M = torch.empty((B, L), dtype=torch.float32, device=dev)
    ml = torch.empty((B, 2), dtype=torch.float32, device=dev)
    part = torch.empty((B * S * (L + 2),), dtype=torch.float32, device=dev)
    check(_lib.lib().murcl_abmil_pool_fwd(ptr(H), ptr(Wa), ptr(ba), ptr(wb), ptr(bb), ptr(scores), ptr(A), ptr(M),
                                          ptr(ml), ptr(part), B, N, L, D, dt(H), int(exact_tanh), stream()),
          "abmil_pool_fwd")
    return scores, A, M, ml


def abmil_pool_bwd(H, Wa, ba, wb, scores, ml, M, dM, exact_tanh=None):
    """-> dT [B*N,128] (dtype of H; 32 spare rows allocated behind it), dba[128], dwb[128], dbb[1]."""
    _need_cuda(H, Wa, dM)
    H, Wa, dM = _c(H), _c(Wa), _c(dM)
    B, N, L = H.shape
    D = Wa.shape[0]
    if exact_tanh is None:
        exact_tanh = H.dtype == torch.float32
    dev = H.device
    dT_full = torch.empty((B * N + 32, D), dtype=H.dtype, device=dev)
    dba = torch.zeros((D,), dtype=torch.float32, device=dev)
    dwb = torch.zeros((D,), dtype=torch.float32, device=dev)
    dbb = torch.zeros((1,), dtype=torch.float32, device=dev)
    check(_lib.lib().murcl_abmil_pool_bwd(ptr(H), ptr(Wa), ptr(ba), ptr(wb), ptr(scores), ptr(ml), ptr(M), ptr(dM),
                                          ptr(dT_full), ptr(dba), ptr(dwb), ptr(dbb), B, N, L, D, dt(H),
                                          int(exact_tanh), stream()), "abmil_pool_bwd")
    return dT_full[:B * N], dba, dwb, dbb


def ntxent(z, temperature, want_grad=True, grad_lo=0, grad_hi=None):
    """z [2B,128] f32 -> (loss [1], dz [2B,128] or None, sim [B])."""
    _need_cuda(z)
    z = _c(z.float())
    n, P = z.shape
    Bh = n // 2
    if grad_hi is None:
        grad_hi = Bh
    dev = z.device
    ws = torch.empty((_lib.lib().murcl_ntxent_workspace_bytes(n) + 3) // 4, dtype=torch.float32, device=dev)
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    dz = torch.empty_like(z) if want_grad else None
    sim = torch.empty((Bh,), dtype=torch.float32, device=dev)
    check(_lib.lib().murcl_ntxent_fwd_bwd(ptr(z), n, P, float(temperature), ptr(loss), ptr(dz), ptr(sim), grad_lo,
                                          grad_hi, ptr(ws), stream()), "ntxent_fwd_bwd")
    return loss, dz, sim


def cast(x, dtype):
    _need_cuda(x)
    x = _c(x)
    if x.dtype == dtype:
        return x
    y = torch.empty_like(x, dtype=dtype)
    check(_lib.lib().murcl_cast(ptr(x), ptr(y), x.numel(), dt(x), dt(y), stream()), "cast")
    return y


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


def gru_gates_fwd(gi, gh, hprev):
    _need_cuda(gi, gh)
    B, H3 = gi.shape
    H = H3 // 3
    hnew = torch.empty((B, H), dtype=torch.float32, device=gi.device)
    gates = torch.empty((B, H3), dtype=torch.float32, device=gi.device)
    check(_lib.lib().murcl_gru_gates_fwd(ptr(gi), ptr(gh), ptr(hprev), ptr(hnew), ptr(gates), B, H, stream()),
          "gru_gates_fwd")
    return hnew, gates


def gru_gates_bwd(dh, gates, gh, hprev):
    _need_cuda(dh, gates, gh)
    dh = _c(dh)
    B, H = dh.shape
    dgi = torch.empty((B, 3 * H), dtype=torch.float32, device=dh.device)
    dgh = torch.empty((B, 3 * H), dtype=torch.float32, device=dh.device)
    dhp = torch.empty((B, H), dtype=torch.float32, device=dh.device)
    check(_lib.lib().murcl_gru_gates_bwd(ptr(dh), ptr(gates), ptr(gh), ptr(hprev), ptr(dgi), ptr(dgh), ptr(dhp), B, H,
                                         stream()), "gru_gates_bwd")
    return dgi, dgh, dhp


def adam_step(p, g, m, v, lr, betas, eps, weight_decay, step):
    _need_cuda(p, g, m, v)
    assert p.is_contiguous() and g.is_contiguous() and p.dtype == torch.float32 and g.dtype == torch.float32
    check(_lib.lib().murcl_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), float(betas[0]),
                                     float(betas[1]), float(eps), float(weight_decay), int(step), stream()), "adam_step")
