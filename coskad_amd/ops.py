"""Thin, checked wrappers: torch CUDA tensors -> C-ABI calls (include/coskad_hip.h).

Every wrapper validates device / dtype / contiguity / shape on the host before a kernel
sees a pointer, enqueues on torch's current stream and never synchronises.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from ._lib import call, i32, ptr

Tensor = torch.Tensor


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: Optional[Tensor], name: str, shape=None, dtype=torch.float32, optional=False) -> None:
    if t is None:
        if optional:
            return
        raise ValueError(f"{name}: tensor required")
    if not t.is_cuda:
        raise _lib.CoskadHipError(f"{name}: expected a CUDA (ROCm) tensor, got device {t.device}; "
                                  "the COSKAD hot path runs only on the HIP extension (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")


def cop(co: int) -> int:
    return (co + 15) // 16 * 16


def gcn(x: Tensor, A: Tensor, Tm: Tensor, adjoint: bool = False) -> Tensor:
    """ConvTemporalGraphical.forward (reference stsgcn.py:143-156) or its adjoint."""
    N, C, T, V = x.shape
    _chk(x, "x"); _chk(A, "A", (T, V, V)); _chk(Tm, "T", (V, T, T))
    out = torch.empty_like(x)
    call("coskad_gcn_f32", ptr(x), ptr(out), ptr(A), ptr(Tm), i32(N * C), i32(T), i32(V),
         i32(1 if adjoint else 0), _stream())
    return out


def bn_fold(Wt, bt, gt, bet, mean_t, var_t, Wr, br, gr, ber, mean_r, var_r):
    """BN stats -> (wfold [2Ci, CoP], bias [CoP]) for layer_apply."""
    Co, Ci = Wt.shape[0], Wt.shape[1]
    for n, t in (("Wt", Wt), ("gamma_t", gt), ("beta_t", bet), ("mean_t", mean_t), ("var_t", var_t)):
        _chk(t, n)
    for n, t in (("bt", bt), ("Wr", Wr), ("br", br), ("gamma_r", gr), ("beta_r", ber), ("mean_r", mean_r), ("var_r", var_r)):
        _chk(t, n, optional=True)
    wfold = torch.empty(2 * Ci, cop(Co), device=Wt.device, dtype=torch.float32)
    bias = torch.empty(cop(Co), device=Wt.device, dtype=torch.float32)
    call("coskad_bn_fold_f32", ptr(Wt), ptr(bt), ptr(gt), ptr(bet), ptr(mean_t), ptr(var_t),
         ptr(Wr), ptr(br), ptr(gr), ptr(ber), ptr(mean_r), ptr(var_r), ptr(wfold), ptr(bias),
         i32(Ci), i32(Co), _stream())
    return wfold, bias


def layer_apply(x: Tensor, A: Tensor, Tm: Tensor, wfold: Tensor, bias: Tensor, Co: int,
                in_slope: Optional[Tensor] = None, out_slope: Optional[Tensor] = None,
                out: Optional[Tensor] = None) -> Tensor:
    """ST_GCNN_layer.forward with folded BN (reference stsgcn.py:94-116)."""
    B, Ci, T, V = x.shape
    _chk(x, "x"); _chk(A, "A", (T, V, V)); _chk(Tm, "T", (V, T, T))
    _chk(wfold, "wfold", (2 * Ci, cop(Co))); _chk(bias, "bias", (cop(Co),))
    _chk(in_slope, "in_slope", (1,), optional=True); _chk(out_slope, "out_slope", (1,), optional=True)
    if out is None:
        out = torch.empty(B, Co, T, V, device=x.device, dtype=torch.float32)
    else:
        _chk(out, "out", (B, Co, T, V))
    call("coskad_layer_apply_f32", ptr(x), ptr(out), ptr(A), ptr(Tm), ptr(wfold), ptr(bias),
         ptr(in_slope), ptr(out_slope), i32(B), i32(Ci), i32(Co), i32(T), i32(V), _stream())
    return out


def stat_floats(Ci: int, Co: int) -> int:
    fn = _lib.lib().coskad_stat_floats
    fn.restype = ctypes.c_int
    return fn(i32(Ci), i32(Co))


def train_stats_ws_bytes(Ci: int) -> int:
    fn = _lib.lib().coskad_train_stats_ws_bytes
    fn.restype = ctypes.c_size_t
    return fn(i32(Ci))


def layer_train_stats(x, A, Tm, in_slope, Wt, bt, gt, bet, rm_t, rv_t, nbt_t,
                      Wr, br, gr, ber, rm_r, rv_r, nbt_r, ws, momentum: float = 0.1):
    """Train-mode BN statistics of one layer -> (wfold, bias, stat); updates running stats in place."""
    B, Ci, T, V = x.shape
    Co = Wt.shape[0]
    _chk(x, "x"); _chk(A, "A", (T, V, V)); _chk(Tm, "T", (V, T, T))
    _chk(Wt, "Wt", (Co, Ci)); _chk(gt, "gamma_t", (Co,)); _chk(bet, "beta_t", (Co,))
    for n, t in (("bt", bt), ("rm_t", rm_t), ("rv_t", rv_t), ("br", br), ("gamma_r", gr), ("beta_r", ber),
                 ("rm_r", rm_r), ("rv_r", rv_r)):
        _chk(t, n, (Co,), optional=True)
    _chk(Wr, "Wr", (Co, Ci), optional=True)
    _chk(in_slope, "in_slope", (1,), optional=True)
    _chk(nbt_t, "nbt_t", (), dtype=torch.int64, optional=True)
    _chk(nbt_r, "nbt_r", (), dtype=torch.int64, optional=True)
    need = train_stats_ws_bytes(Ci)
    if ws is None or ws.numel() * ws.element_size() < need:
        raise ValueError(f"workspace too small: need {need} bytes")
    wfold = torch.empty(2 * Ci, cop(Co), device=x.device, dtype=torch.float32)
    bias = torch.empty(cop(Co), device=x.device, dtype=torch.float32)
    stat = torch.empty(stat_floats(Ci, Co), device=x.device, dtype=torch.float32)
    call("coskad_layer_train_stats_f32", ptr(x), ptr(A), ptr(Tm), ptr(in_slope), ptr(Wt), ptr(bt), ptr(gt),
         ptr(bet), ptr(rm_t), ptr(rv_t), ptr(nbt_t), ptr(Wr), ptr(br), ptr(gr), ptr(ber), ptr(rm_r),
         ptr(rv_r), ptr(nbt_r), ctypes.c_float(momentum), ptr(wfold), ptr(bias), ptr(stat), ptr(ws),
         ctypes.c_size_t(ws.numel() * ws.element_size()), i32(B), i32(Ci), i32(Co), i32(T), i32(V), _stream())
    return wfold, bias, stat
