"""Python side of the parked one-pass pooling backward (tools/_abl/attn_pool_bwd_dwa.hip -> tools/_abl/lib/kd.so, build_kd.py).
Lab equipment: not imported by the product; tests/test_gpu_kernels.py keeps its parity test through this wrapper."""
import ctypes
import os

import torch

from murcl_amd import _lib, ops

LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "kd.so")
_P, _I, _L = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
_kd = None


def lib():
    global _kd
    if _kd is None:
        _lib.lib()                                                # the product library first (kd.so resolves two symbols from it)
        L = ctypes.CDLL(LIB)
        L.murcl_abmil_pool_bwd_dwa_ws_floats.argtypes = [_I, _I, _I, _I, _I]
        L.murcl_abmil_pool_bwd_dwa_ws_floats.restype = _L
        L.murcl_abmil_pool_bwd_dwa.argtypes = [_P] * 13 + [_I, _P, _L, _I, _I, _I, _I, _I, _I, _P]
        L.murcl_abmil_pool_bwd_dwa.restype = _I
        _kd = L
    return _kd


def available():
    return os.path.exists(LIB)


def pool_bwd_dwa(H, Wa, ba, wb, scores, ml, M, dM, dwa="new"):
    """-> dT [B*N,128], dba, dwb, dbb, dWa [128,512] f32 = dT^T H from the same pass over H (bf16, L = 512, D = 128).
    ``dwa``: "new" or a [128,512] f32 tensor that is ADDED to."""
    H, Wa, dM = H.contiguous(), Wa.contiguous(), dM.contiguous()
    B, N, L = H.shape
    D = Wa.shape[0]
    dev = H.device
    wsf = lib().murcl_abmil_pool_bwd_dwa_ws_floats(B, N, L, D, _lib.dt(H))
    if not wsf:
        raise ValueError("pool_bwd_dwa: bf16, L = 512, D = 128 only")
    dT_full = torch.empty((B * N + 32, D), dtype=H.dtype, device=dev)
    z = torch.zeros((2 * D + 1,), dtype=torch.float32, device=dev)
    dba, dwb, dbb = z[:D], z[D:2 * D], z[2 * D:]
    acc = not isinstance(dwa, str)
    dWa = dwa if acc else torch.empty((D, L), dtype=torch.float32, device=dev)
    ws = torch.empty((wsf,), dtype=torch.float32, device=dev)
    p = _lib.ptr
    _lib.check(lib().murcl_abmil_pool_bwd_dwa(p(H), p(Wa), p(ba), p(wb), p(scores), p(ml), p(M), p(dM), p(dT_full), p(dba), p(dwb),
                                              p(dbb), p(dWa), int(acc), p(ws), wsf, B, N, L, D, _lib.dt(H), 0, _lib.stream()),
               "abmil_pool_bwd_dwa")
    return dT_full[:B * N], dba, dwb, dbb, dWa


__all__ = ["available", "pool_bwd_dwa", "ops"]
