"""ctypes binding of libmurcl_amd.so (the C-ABI declared in include/murcl_amd.h).

There is deliberately NO fallback: if the HIP library is missing or a launch fails, the call
raises.  torch must be imported first so the library binds to the HIP runtime torch loaded.
"""
import ctypes
import os

import torch  # noqa: F401  (loads libamdhip64.so.7 before our library resolves it)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MURCL_AMD_LIB") or os.path.join(_HERE, "libmurcl_amd.so")      # override: A/B builds (tools/)

F32, BF16 = 0, 1
EPI_NONE, EPI_BIAS, EPI_BIAS_RELU, EPI_MASK, EPI_RANK1_MASK = 0, 1, 2, 3, 4

_c = ctypes
_P, _I, _L, _F = _c.c_void_p, _c.c_int, _c.c_long, _c.c_float

SIGNATURES = {
    "murcl_gemm_nt": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _I, _P, _I, _P],
    "murcl_gemm_tn": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "murcl_gemm_tn_workspace_bytes": [_I, _I, _I, _I],
    "murcl_gemm_tn_ws": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _L, _P, _I, _P],
    "murcl_gemm_tn_grouped_workspace_bytes": [_P, _I, _I],
    "murcl_gemm_tn_grouped": [_P, _I, _I, _P, _L, _P],
    "murcl_panel_gemm_colsum_rows": [_I, _I, _I, _I],
    "murcl_panel_gemm_supported": [_I, _I, _I, _I, _I],
    "murcl_panel_gemm": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _P],
    "murcl_panel_gemm_drop": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _F, ctypes.c_ulonglong, ctypes.c_ulonglong, _P],
    "murcl_cu_budget": [],
    "murcl_set_cu_budget": [_I],
    "murcl_calib_copy": [_P, _P, _L, _P],
    "murcl_calib_mfma_bf16": [_P, _I, _P],
    "murcl_abmil_pool_workspace": [_I, _I, _I, _c.POINTER(_I), _c.POINTER(_I)],
    "murcl_abmil_pool_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "murcl_dropout_mask": [_P, _L, _F, _F, ctypes.c_ulonglong, _I, _P],
    "murcl_kmeans_workspace_bytes": [_I, _I, _I],
    "murcl_kmeans_step": [_P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P],
    "murcl_abmil_pool_combine": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "murcl_abmil_pool_decoder": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "murcl_abmil_pool_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "murcl_ntxent_workspace_bytes": [_I],
    "murcl_ntxent_fwd_bwd": [_P, _I, _I, _F, _P, _P, _P, _I, _I, _I, _P, _P],
    "murcl_ntxent_fwd_bwd_batched": [_P, _I, _I, _I, _F, _P, _P, _P, _P],
    "murcl_ntxent_xchg_bytes": [_I],
    "murcl_ntxent_small_xchg": [_P, _I, _I, _I, _F, _P, _P, _P, _I, _I, _I, _P, _P],
    "murcl_subbag_select": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "murcl_subbag_gather_mix": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "murcl_subbag_gather_mix_rows": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "murcl_mixup": [_P, _P, _P, _P, _I, _L, _I, _P],
    "murcl_dsmil_argmax": [_P, _I, _I, _I, _I, _P, _P],
    "murcl_dsmil_argmax_max": [_P, _I, _I, _I, _I, _P, _P, _P],
    "murcl_gather_rows": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    "murcl_dsmil_attn": [_P, _I, _I, _P, _I, _I, _I, _P, _P],
    "murcl_weighted_rowsum": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "murcl_weighted_rowsum_acc": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "murcl_rows_dot": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "murcl_rows_dot_bias": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "murcl_rows_dot_wsum_plan": [_I, _I, _I, _I],
    "murcl_rows_dot_wsum": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "murcl_dsmil_attn_bwd": [_P, _P, _P, _I, _I, _P, _I, _I, _I, _P, _I, _P, _P, _P],
    "murcl_gated_score_fwd": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _F, ctypes.c_ulonglong, ctypes.c_ulonglong, _P],
    "murcl_gated_score_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _F, ctypes.c_ulonglong, ctypes.c_ulonglong, _P],
    "murcl_dsmil_softmax": [_P, _I, _I, _I, _P],
    "murcl_dsmil_stream_plan": [_I, _I, _I, _I],
    "murcl_dsmil_qv": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P],
    "murcl_dsmil_qv_bwd": [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "murcl_dsmil_qv_bwd_cls": [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _P, _P, _I, _P],
    "murcl_dsmil_attn_pool": [_P, _P, _F, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "murcl_dsmil_attn_pool_bwd": [_P, _P, _P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "murcl_dsmil_softmax_bwd": [_P, _P, _I, _I, _I, _P, _P, _P],
    "murcl_clam_inst_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _I, _P],
    "murcl_clam_inst_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "murcl_gated_score_bwd_il": [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, ctypes.c_ulonglong, ctypes.c_ulonglong, _P, _P, _P, _P, _I, _I, _P],
    "murcl_softmax_rows": [_P, _P, _I, _I, _P],
    "murcl_softmax_rows_parts": [_P, _I, _P, _P, _I, _I, _P, _I, _P],
    "murcl_softmax_rows_bwd": [_P, _P, _P, _I, _I, _P],
    "murcl_topk_ids": [_P, _I, _I, _I, _P, _P],
    "murcl_take_rows": [_P, _P, _P, _I, _I, _I, _P],
    "murcl_scatter_add_rows_masked": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "murcl_cross_entropy": [_P, _P, _I, _I, _P, _P, _P, _P, _I, _P],
    "murcl_mul": [_P, _P, _P, _L, _I, _P],
    "murcl_axpby": [_P, _P, _F, _F, _P, _L, _P],
    "murcl_mean_small": [_P, _I, _P, _P],
    "murcl_group_mean": [_P, _I, _I, _P, _P],
    "murcl_copy_bytes": [_P, _P, _L, _P],
    "murcl_gemm_nt_smallk": [_P, _P, _P, _I, _I, _I, _I, _P],
    "murcl_pad_cols": [_P, _P, _L, _I, _I, _I, _P],
    "murcl_policy_head_fwd": [_P, _P, _P, _F, _I, _I, _P, _P, _P, _P],
    "murcl_policy_head_bwd": [_P, _P, _P, _F, _I, _I, _P, _P],
    "murcl_ppo_returns": [_P, _F, _I, _I, _P, _P],
    "murcl_ppo_returns_raw": [_P, _F, _I, _I, _P, _P, _P],
    "murcl_ppo_returns_finish": [_P, _I, _P, _L, _P],
    "murcl_ppo_loss": [_P, _P, _P, _P, _F, _F, _I, _L, _P, _P, _P, _P],
    "murcl_cast": [_P, _P, _L, _I, _I, _P],
    "murcl_transpose_cast": [_P, _P, _I, _I, _I, _P],
    "murcl_colsum": [_P, _P, _I, _I, _I, _I, _I, _P],
    "murcl_relu_bwd": [_P, _P, _P, _L, _P],
    "murcl_gru_gates_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "murcl_gru_gates_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "murcl_gru_gates_bwd_into": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "murcl_gru_step_supported": [_I, _I, _I],
    "murcl_gru_step_fwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "murcl_gru_step_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "murcl_ppo_act_workspace": [_I, _I, _I],
    "murcl_ppo_act": [_P, _I, _I, _I, _P, _P, _P, _F, _I, _P, _P, _P, _P, _P],
    "murcl_ppo_epoch_workspace": [_I, _I, _I, _I],
    "murcl_ppo_epoch": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _L, _F, _F, _F, _P, _P, _P],
    "murcl_ppo_epoch_wt": [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _L, _F, _F, _F, _P, _P, _P],
    "murcl_step_draws": [ctypes.c_ulonglong, _P, _L, _P, _L, _P, _P, _I, _I, _F, _P],
    "murcl_stack_lists": [_P, _I, _P],
    "murcl_add_lists": [_P, _I, _P],
    "murcl_cast_batch": [_P, _I, _I, _P],
    "murcl_cast_batch_flat": [_P, _P, _I, _I, _P],
    "murcl_cast_batch_flat_tick": [_P, _P, _I, _I, _P, _P],
    "murcl_relu_bitmask": [_P, _P, _I, _I, _I, _I, _P],
    "murcl_dropout_relu_bitmask": [_P, _P, _I, _I, _F, _F, ctypes.c_ulonglong, _I, _P],
    "murcl_adam_step": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _I, _P],
    "murcl_adam_multi": [_P, _I, _F, _F, _F, _F, _I, _P],
    "murcl_adam_multi_live": [_P, _I, _F, _F, _F, _F, _I, _P, _P],
    "murcl_adam_multi_live_deferred": [_P, _I, _F, _F, _F, _F, _I, _P, _P],
    "murcl_replay_tick": [_P, _P],
    "murcl_sgd_step": [_P, _P, _P, _L, _F, _F, _I, _F, _I, _I, _P],
}
_RESTYPE = {"murcl_ntxent_workspace_bytes": _L, "murcl_ntxent_xchg_bytes": _L, "murcl_kmeans_workspace_bytes": _L, "murcl_ppo_act_workspace": _L, "murcl_gemm_tn_workspace_bytes": _L, "murcl_gemm_tn_grouped_workspace_bytes": _L,
            "murcl_ppo_epoch_workspace": _L}



class TnProblem(ctypes.Structure):
    """murcl_tn_problem of include/murcl_amd.h (one product of murcl_gemm_tn_grouped)."""
    _fields_ = [("A", _P), ("B", _P), ("C", _P), ("colsum_part", _P), ("colsum_out", _P),
                ("M", _I), ("N1", _I), ("N2", _I), ("lda", _I), ("ldb", _I), ("ldc", _I), ("colsum_rows", _I), ("flags", _I), ("scale", _F)]


TN_OVERWRITE, TN_DEINTERLEAVE, TN_SCALE = 1, 2, 4


_lib = None


def lib():
    """The loaded library; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m murcl_amd.build` "
                "(murcl_amd has no CPU/PyTorch fallback for its kernels)")
        L = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)           # AttributeError if the .so lacks a declared symbol
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, _I)
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"murcl_amd: {what} failed with code {rc}")


def ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream on the current device (the raw getter: ~1 us instead of ~10 us per launch
    for torch.cuda.current_stream(), which matters with ~50 launches per 1.7 ms step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"murcl_amd kernels take float32 or bfloat16, got {t.dtype}")
