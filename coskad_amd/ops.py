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


def _bytes(t: Tensor) -> int:
    return t.numel() * t.element_size()


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
         ptr(in_slope), ptr(out_slope), i32(B), i32(Ci), i32(Co), i32(T), i32(V), _stream(), tag=(Ci, Co))
    return out


def layer_first_pair_ok(Ci: int, Cm: int, Co: int, T: int, V: int) -> bool:
    """the first layer (2 -> Cm) and the layer behind it (Cm -> Co), folded, in one pass (csrc/eval_layer_bpc.hip, FIRST form)"""
    return bool(_lib.lib().coskad_layer_first_pair_ok(i32(T), i32(V), i32(Ci), i32(Cm), i32(Co)))


def layer_first_pair_apply(x: Tensor, A1: Tensor, T1: Tensor, wfold1: Tensor, bias1: Tensor, A2: Tensor, T2: Tensor, wfold2: Tensor,
                           bias2: Tensor, Cm: int, Co: int, mid_slope: Tensor, out_slope: Optional[Tensor] = None) -> Tensor:
    """layer_apply twice (folded BatchNorm: eval mode) for the first two layers of the encoder without the activation between them in
    HBM: x [B, 2, T, V] (the network input) -> [B, Co, T, V]; `mid_slope`: the first layer's PReLU weight."""
    B, Ci, T, V = x.shape
    _chk(x, "x", (B, 2, T, V)); _chk(A1, "A1", (T, V, V)); _chk(T1, "T1", (V, T, T)); _chk(A2, "A2", (T, V, V)); _chk(T2, "T2", (V, T, T))
    _chk(wfold1, "wfold1", (4, Cm)); _chk(bias1, "bias1", (Cm,)); _chk(wfold2, "wfold2", (2 * Cm, Co)); _chk(bias2, "bias2", (Co,))
    _chk(mid_slope, "mid_slope", (1,)); _chk(out_slope, "out_slope", (1,), optional=True)
    out = torch.empty(B, Co, T, V, device=x.device, dtype=torch.float32)
    call("coskad_layer_first_pair_apply_f32", ptr(x), ptr(out), ptr(A1), ptr(T1), ptr(wfold1), ptr(bias1), ptr(A2), ptr(T2), ptr(wfold2),
         ptr(bias2), ptr(mid_slope), ptr(out_slope), i32(B), i32(Cm), i32(Co), i32(T), i32(V), _stream())
    return out


def layer_fits(Ci: int, Co: int, T: int, V: int) -> bool:
    """Do the LDS-resident tile kernels take a (Ci -> Co) layer at this geometry?  (pure host arithmetic)"""
    fn = _lib.lib().coskad_layer_fits
    fn.restype = ctypes.c_int
    return bool(fn(i32(Ci), i32(Co), i32(T), i32(V)))


def stat_floats(Ci: int, Co: int) -> int:
    fn = _lib.lib().coskad_stat_floats
    fn.restype = ctypes.c_int
    return fn(i32(Ci), i32(Co))


def train_stats_ws_bytes(Ci: int) -> int:
    fn = _lib.lib().coskad_train_stats_ws_bytes
    fn.restype = ctypes.c_size_t
    return fn(i32(Ci))


def layer_train_stats(x, A, Tm, in_slope, Wt, bt, gt, bet, rm_t, rv_t, nbt_t,
                      Wr, br, gr, ber, rm_r, rv_r, nbt_r, ws, momentum: float = 0.1, Z: Optional[Tensor] = None):
    """Train-mode BN statistics of one layer -> (wfold, bias, stat); updates running stats in place.
    Z (optional, same shape as x): receives gcn(PReLU(x)) for layer_apply_z and the stored-Z backward."""
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
    _chk(Z, "Z", tuple(x.shape), optional=True)
    args = (ptr(x), ptr(A), ptr(Tm), ptr(in_slope), ptr(Wt), ptr(bt), ptr(gt),
            ptr(bet), ptr(rm_t), ptr(rv_t), ptr(nbt_t), ptr(Wr), ptr(br), ptr(gr), ptr(ber), ptr(rm_r),
            ptr(rv_r), ptr(nbt_r), ctypes.c_float(momentum), ptr(wfold), ptr(bias), ptr(stat), ptr(ws),
            ctypes.c_size_t(ws.numel() * ws.element_size()), i32(B), i32(Ci), i32(Co), i32(T), i32(V), _stream())
    if Z is None:
        call("coskad_layer_train_stats_f32", *args)
    else:
        call("coskad_layer_train_stats_z_f32", *args, ptr(Z))
    return wfold, bias, stat


# ---- SyncBN building blocks (csrc: "SyncBN" section of include/coskad_hip.h): batch reductions and folds as separate calls ---------
def layer_train_moments(x, A, Tm, in_slope, ws, Z: Optional[Tensor] = None) -> Tensor:
    """This rank's fp64 moment sums [sum xx^T Ci^2][sum x Ci][sum zz^T Ci^2][sum z Ci] of one layer's input (and Z = gcn(PReLU(x)))."""
    B, Ci, T, V = x.shape
    _chk(x, "x"); _chk(A, "A", (T, V, V)); _chk(Tm, "T", (V, T, T)); _chk(in_slope, "in_slope", (1,), optional=True)
    _chk(Z, "Z", tuple(x.shape), optional=True)
    need = train_stats_ws_bytes(Ci)
    if ws is None or _bytes(ws) < need:
        raise ValueError(f"workspace too small: need {need} bytes")
    sums = torch.empty(2 * (Ci * Ci + Ci), device=x.device, dtype=torch.float64)
    call("coskad_layer_train_moments_f32", ptr(x), ptr(A), ptr(Tm), ptr(in_slope), ptr(Z), ptr(sums), ptr(ws),
         ctypes.c_size_t(_bytes(ws)), i32(B), i32(Ci), i32(T), i32(V), _stream())
    return sums


def layer_moment_sums(partials: Tensor, rows: int, Ci: int) -> Tensor:
    """fp64 sums of the moment partial rows a layer_apply_next wrote for the next layer."""
    _chk(partials, "partials")
    if partials.numel() < rows * 2 * (Ci * Ci + Ci):
        raise ValueError("layer_moment_sums: partials smaller than rows x 2 (Ci^2 + Ci)")
    sums = torch.empty(2 * (Ci * Ci + Ci), device=partials.device, dtype=torch.float64)
    call("coskad_layer_moment_sums_f32", ptr(partials), i32(rows), i32(Ci), ptr(sums), _stream())
    return sums


def layer_train_fold_sums(sums, count, Wt, bt, gt, bet, rm_t, rv_t, nbt_t, Wr, br, gr, ber, rm_r, rv_r, nbt_r, momentum: float = 0.1):
    """(wfold, bias, stat) from moment sums over `count` positions (global clips x T x V once the ranks' sums are added)."""
    Co, Ci = Wt.shape
    _chk(sums, "sums", (2 * (Ci * Ci + Ci),), dtype=torch.float64); _chk(Wt, "Wt", (Co, Ci)); _chk(gt, "gamma_t", (Co,)); _chk(bet, "beta_t", (Co,))
    for n, t in (("bt", bt), ("rm_t", rm_t), ("rv_t", rv_t), ("br", br), ("gamma_r", gr), ("beta_r", ber), ("rm_r", rm_r), ("rv_r", rv_r)):
        _chk(t, n, (Co,), optional=True)
    _chk(Wr, "Wr", (Co, Ci), optional=True)
    _chk(nbt_t, "nbt_t", (), dtype=torch.int64, optional=True); _chk(nbt_r, "nbt_r", (), dtype=torch.int64, optional=True)
    wfold = torch.empty(2 * Ci, cop(Co), device=Wt.device, dtype=torch.float32)
    bias = torch.empty(cop(Co), device=Wt.device, dtype=torch.float32)
    stat = torch.empty(stat_floats(Ci, Co), device=Wt.device, dtype=torch.float32)
    call("coskad_layer_train_fold_sums_f32", ptr(sums), ctypes.c_double(float(count)), ptr(Wt), ptr(bt), ptr(gt), ptr(bet), ptr(rm_t),
         ptr(rv_t), ptr(nbt_t), ptr(Wr), ptr(br), ptr(gr), ptr(ber), ptr(rm_r), ptr(rv_r), ptr(nbt_r), ctypes.c_float(momentum),
         ptr(wfold), ptr(bias), ptr(stat), i32(Ci), i32(Co), _stream())
    return wfold, bias, stat


def layer_bwd_stats(x_in, dU, A, Tm, in_slope, has_residual: bool, ws, Z=None):
    """Stage 1 of layer_bwd alone -> (chain buffer, rows): the partial rows, then their fp64 sums (chain_sums(buf, rows, Ci, Co))."""
    B, Ci, T, V = x_in.shape
    Co = dU.shape[1]
    _chk(x_in, "x_in"); _chk(dU, "dU", (B, Co, T, V)); _chk(A, "A", (T, V, V)); _chk(Tm, "T", (V, T, T))
    _chk(in_slope, "in_slope", (1,), optional=True); _chk(Z, "Z", (B, Ci, T, V), optional=True)
    need = layer_bwd_ws_bytes(B, Ci, Co, T, V)
    if ws is None or _bytes(ws) < need:
        raise ValueError(f"workspace too small: need {need} bytes")
    fn = _lib.lib().coskad_layer_bwd_stats_floats
    fn.restype = ctypes.c_size_t
    buf = torch.empty(fn(i32(B), i32(Ci), i32(Co), i32(T), i32(V)), device=x_in.device, dtype=torch.float32)
    rows = ctypes.c_int(0)
    call("coskad_layer_bwd_stats_f32", ptr(x_in), ptr(dU), ptr(A), ptr(Tm), ptr(in_slope), i32(1 if has_residual else 0), ptr(buf),
         ctypes.c_size_t(_bytes(buf)), ctypes.byref(rows), ptr(ws), ctypes.c_size_t(_bytes(ws)), i32(B), i32(Ci), i32(Co), i32(T), i32(V),
         _stream(), ptr(Z))
    return buf, rows.value


def chain_sums(buf: Tensor, rows: int, Ci: int, Co: int) -> Tensor:
    """The fp64 sums [P Co*Ci][Q Co*Ci][sdU Co] inside a backward chain buffer, as a view (all-reduce it in place for SyncBN)."""
    fn = _lib.lib().coskad_layer_bwd_sums_offset
    fn.restype = ctypes.c_size_t
    off = fn(i32(rows), i32(Ci), i32(Co))
    E = 2 * Co * Ci + Co
    return buf[off:off + 2 * E].view(torch.float64)


def layer_apply_z(Z, x, A, Tm, wfold, bias, Co, in_slope=None, out_slope=None, out=None):
    """U = Wz.Z + Wx.PReLU(x) + b from the stored Z = gcn(PReLU(x)) (training forward; streaming, no recompute)."""
    B, Ci, T, V = x.shape
    _chk(x, "x"); _chk(Z, "Z", (B, Ci, T, V)); _chk(wfold, "wfold", (2 * Ci, cop(Co))); _chk(bias, "bias", (cop(Co),))
    _chk(in_slope, "in_slope", (1,), optional=True); _chk(out_slope, "out_slope", (1,), optional=True)
    if out is None:
        out = torch.empty(B, Co, T, V, device=x.device, dtype=torch.float32)
    _chk(A, "A", (T, V, V)); _chk(Tm, "T", (V, T, T))
    call("coskad_layer_apply_z_f32", ptr(Z), ptr(x), ptr(out), ptr(A), ptr(Tm), ptr(wfold), ptr(bias), ptr(in_slope), ptr(out_slope),
         i32(B), i32(Ci), i32(Co), i32(T), i32(V), _stream())
    return out


def layer_apply_next_ok(Ci: int, Co: int, T: int, V: int) -> bool:
    """Does csrc/fused_apply_next.hip take a (Ci -> Co) layer (apply + the next layer's statistics in one kernel)?"""
    fn = _lib.lib().coskad_layer_apply_next_ok
    fn.restype = ctypes.c_int
    return bool(fn(i32(Ci), i32(Co), i32(T), i32(V)))


def layer_apply_next_rows(B: int, Ci: int, Co: int) -> int:
    """Partial rows layer_apply_next writes for a batch of B clips (each 2 (Co^2 + Co) floats)."""
    fn = _lib.lib().coskad_layer_apply_next_rows
    fn.restype = ctypes.c_int
    return fn(i32(B), i32(Ci), i32(Co))


def ftab_floats() -> int:
    fn = _lib.lib().coskad_ftab_floats
    fn.restype = ctypes.c_int
    return fn()


def build_ftabs(As, Ts, tabs) -> None:
    """Forward mixing tables of up to 4 layers (lists of A [T,V,V], T [V,T,T], tab [ftab_floats()]) in one launch."""
    n = len(As)
    if not (1 <= n <= 4) or len(Ts) != n or len(tabs) != n:
        raise ValueError("build_ftabs: 1..4 layers, equal list lengths")
    T, V = As[0].shape[0], As[0].shape[1]
    nf = ftab_floats()
    for A, Tm, tab in zip(As, Ts, tabs):
        _chk(A, "A", (T, V, V)); _chk(Tm, "T", (V, T, T)); _chk(tab, "tab", (nf,))
    arr = ctypes.c_void_p * n
    call("coskad_build_ftab_f32", arr(*[t.data_ptr() for t in As]), arr(*[t.data_ptr() for t in Ts]),
         arr(*[t.data_ptr() for t in tabs]), i32(n), i32(T), i32(V), _stream())


def layer_apply_next(Z, x, wfold, bias, Co, in_slope, out_slope, ftab_next, partials, T, V, out=None, Z_next=None):
    """U = Wz.Z + Wx.PReLU(x) + b  AND  the next layer's Z_next = gcn_next(PReLU_out(U)) + its moment partials
    (csrc/fused_apply_next.hip).  -> (U, Z_next, rows): `rows` partial rows of 2 (Co^2 + Co) floats were written."""
    B, Ci = x.shape[0], x.shape[1]
    _chk(x, "x", (B, Ci, T, V)); _chk(Z, "Z", (B, Ci, T, V)); _chk(wfold, "wfold", (2 * Ci, cop(Co))); _chk(bias, "bias", (cop(Co),))
    _chk(in_slope, "in_slope", (1,), optional=True); _chk(out_slope, "out_slope", (1,)); _chk(ftab_next, "ftab_next", (ftab_floats(),))
    _chk(partials, "partials")
    if out is None:
        out = torch.empty(B, Co, T, V, device=x.device, dtype=torch.float32)
    if Z_next is None:
        Z_next = torch.empty(B, Co, T, V, device=x.device, dtype=torch.float32)
    _chk(out, "out", (B, Co, T, V)); _chk(Z_next, "Z_next", (B, Co, T, V))
    rows = layer_apply_next_rows(B, Ci, Co)
    call("coskad_layer_apply_next_f32", ptr(Z), ptr(x), ptr(out), ptr(wfold), ptr(bias), ptr(in_slope), ptr(out_slope),
         ptr(ftab_next), ptr(Z_next), ptr(partials), ctypes.c_size_t(_bytes(partials)), i32(B), i32(Ci), i32(Co), i32(T), i32(V),
         _stream(), tag=(Ci, Co))
    return out, Z_next, rows


def layer_apply_next_flat_ok(Ci: int, Co: int, T: int, V: int) -> bool:
    """apply + the next layer's statistics in one kernel on the 25-joint layout (csrc/fused_apply_flat.hip, NX form)"""
    return bool(_lib.lib().coskad_layer_apply_next_flat_ok(i32(Ci), i32(Co), i32(T), i32(V)))


def layer_apply_next_flat(Z, x, wfold, bias, Co, in_slope, out_slope, A_next, T_next):
    """U = Wz.Z + Wx.PReLU(x) + b  AND  the next layer's Z_next = gcn_next(PReLU_out(U)) + its moment partials.
    -> (U, Z_next, partials, rows)"""
    B, Ci, T, V = x.shape
    _chk(x, "x"); _chk(Z, "Z", (B, Ci, T, V)); _chk(wfold, "wfold", (2 * Ci, cop(Co))); _chk(bias, "bias", (cop(Co),))
    _chk(in_slope, "in_slope", (1,), optional=True); _chk(out_slope, "out_slope", (1,))
    _chk(A_next, "A_next", (T, V, V)); _chk(T_next, "T_next", (V, T, T))
    rows = int(_lib.lib().coskad_layer_apply_next_flat_rows(i32(B)))
    out = torch.empty(B, Co, T, V, device=x.device, dtype=torch.float32)
    Zn = torch.empty(B, Co, T, V, device=x.device, dtype=torch.float32)
    partials = torch.empty(rows * 2 * (Co * Co + Co), device=x.device, dtype=torch.float32)
    call("coskad_layer_apply_next_flat_f32", ptr(Z), ptr(x), ptr(out), ptr(wfold), ptr(bias), ptr(in_slope), ptr(out_slope), ptr(A_next),
         ptr(T_next), ptr(Zn), ptr(partials), ctypes.c_size_t(_bytes(partials)), i32(B), i32(Ci), i32(Co), i32(T), i32(V), _stream(),
         tag=(Ci, Co))
    return out, Zn, partials, rows


def layer_train_fold(partials, rows, B, T, V, Wt, bt, gt, bet, rm_t, rv_t, nbt_t, Wr, br, gr, ber, rm_r, rv_r, nbt_r, ws,
                     momentum: float = 0.1):
    """The statistics of a layer from moment partials a previous layer_apply_next wrote -> (wfold, bias, stat)."""
    Co, Ci = Wt.shape
    _chk(partials, "partials"); _chk(Wt, "Wt", (Co, Ci)); _chk(gt, "gamma_t", (Co,)); _chk(bet, "beta_t", (Co,))
    for n, t in (("bt", bt), ("rm_t", rm_t), ("rv_t", rv_t), ("br", br), ("gamma_r", gr), ("beta_r", ber),
                 ("rm_r", rm_r), ("rv_r", rv_r)):
        _chk(t, n, (Co,), optional=True)
    _chk(Wr, "Wr", (Co, Ci), optional=True)
    _chk(nbt_t, "nbt_t", (), dtype=torch.int64, optional=True)
    _chk(nbt_r, "nbt_r", (), dtype=torch.int64, optional=True)
    if partials.numel() < rows * 2 * (Ci * Ci + Ci):
        raise ValueError("layer_train_fold: partials smaller than rows x 2 (Ci^2 + Ci)")
    need = train_stats_ws_bytes(Ci)
    if ws is None or _bytes(ws) < need:
        raise ValueError(f"workspace too small: need {need} bytes")
    wfold = torch.empty(2 * Ci, cop(Co), device=Wt.device, dtype=torch.float32)
    bias = torch.empty(cop(Co), device=Wt.device, dtype=torch.float32)
    stat = torch.empty(stat_floats(Ci, Co), device=Wt.device, dtype=torch.float32)
    call("coskad_layer_train_fold_f32", ptr(partials), i32(rows), ptr(Wt), ptr(bt), ptr(gt), ptr(bet), ptr(rm_t), ptr(rv_t),
         ptr(nbt_t), ptr(Wr), ptr(br), ptr(gr), ptr(ber), ptr(rm_r), ptr(rv_r), ptr(nbt_r), ctypes.c_float(momentum),
         ptr(wfold), ptr(bias), ptr(stat), ptr(ws), ctypes.c_size_t(_bytes(ws)), i32(B), i32(Ci), i32(Co), i32(T), i32(V),
         _stream())
    return wfold, bias, stat


def gather(src: Tensor, idx: Tensor) -> Tensor:
    """out[i] = src[idx[i]] (0 where idx < 0): operand streams of the fused encoder from the concatenated parameters."""
    _chk(src, "src"); _chk(idx, "idx", dtype=torch.int32)
    out = torch.empty(idx.shape, device=src.device, dtype=torch.float32)
    call("coskad_gather_f32", ptr(src), ptr(idx), ptr(out), ctypes.c_size_t(idx.numel()), _stream())
    return out


def fused_encoder_out_floats() -> int:
    fn = _lib.lib().coskad_fused_encoder_out_floats
    fn.restype = ctypes.c_int
    return fn()


def fused_encoder(x: Tensor, tab: Tensor, wreg: Tensor, slopes: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """Eval-mode Encoder.forward (components.py:94-105) of the default stack in one kernel -> [B, KP]: the activated last
    layer in tile-major order (coskad_amd/fused_plan.py), zero in the padding columns."""
    B, C, T, V = x.shape
    _chk(x, "x"); _chk(tab, "tab"); _chk(wreg, "wreg"); _chk(slopes, "slopes", (4,))
    if C != 2:
        raise ValueError(f"fused_encoder: 2 input channels expected, got {C}")
    kp = fused_encoder_out_floats()
    if out is None:
        out = torch.empty(B, kp, device=x.device, dtype=torch.float32)
    else:
        _chk(out, "out", (B, kp))
    call("coskad_fused_encoder_f32", ptr(x), ptr(out), ptr(tab), ptr(wreg), ptr(slopes), i32(B), i32(T), i32(V), _stream())
    return out


def layer_bwd_ws_bytes(B, Ci, Co, T, V) -> int:
    fn = _lib.lib().coskad_layer_bwd_ws_bytes
    fn.restype = ctypes.c_size_t
    return fn(i32(B), i32(Ci), i32(Co), i32(T), i32(V))


def layer_bwd_below_rows(B: int, Ci: int, Co: int, below_Ci: int, T: int, V: int) -> int:
    """Partial rows the (Ci -> Co) backward data kernel writes for the layer below it (0: that kernel cannot form them)."""
    fn = _lib.lib().coskad_layer_bwd_below_rows
    fn.restype = ctypes.c_int
    return fn(i32(B), i32(Ci), i32(Co), i32(below_Ci), i32(T), i32(V))


def layer_bwd_below_floats(B: int, Ci: int, Co: int, below_Ci: int, T: int, V: int) -> int:
    """Floats of the chain buffer (partial rows + their fp64 sums) layer_bwd(..., below=...) fills for the layer below."""
    fn = _lib.lib().coskad_layer_bwd_below_floats
    fn.restype = ctypes.c_size_t
    return fn(i32(B), i32(Ci), i32(Co), i32(below_Ci), i32(T), i32(V))


def layer_bwd(x_in, dU, A, Tm, in_slope, stat, Wt, gt, Wr, gr, grads: dict, ws, need_dx=True,
              dIn=None, accumulate=False, Z=None, stats_in=None, below=None, stats_count=0.0):
    """Backward of one layer.  `grads` maps names -> preallocated gradient tensors:
    A, T, Wt, bt (opt), gt, bet, Wr, br (opt), gr, ber, slope_in (opt).  Returns dIn (or None).
    Chain mode (csrc: coskad_layer_bwd_chain_f32, needs Z): `stats_in` = (partial rows tensor, rows) the call for the layer above
    wrote for this layer; `below` = (x_below, Z_below, in_slope_below, below_stats) makes this call write the layer below's partial rows;
    `stats_count`: positions the sums in `stats_in` cover (0: this batch; SyncBN: global clips x T x V)."""
    B, Ci, T, V = x_in.shape
    Co = Wt.shape[0]
    _chk(x_in, "x_in"); _chk(dU, "dU", (B, Co, T, V)); _chk(A, "A", (T, V, V)); _chk(Tm, "T", (V, T, T))
    _chk(Wt, "Wt", (Co, Ci)); _chk(gt, "gamma_t", (Co,)); _chk(stat, "stat", (stat_floats(Ci, Co),))
    _chk(Wr, "Wr", (Co, Ci), optional=True); _chk(gr, "gamma_r", (Co,), optional=True)
    _chk(in_slope, "in_slope", (1,), optional=True)
    for k, shp in (("A", (T, V, V)), ("T", (V, T, T)), ("Wt", (Co, Ci)), ("gt", (Co,)), ("bet", (Co,))):
        _chk(grads[k], "grad " + k, shp)
    for k, shp in (("bt", (Co,)), ("Wr", (Co, Ci)), ("br", (Co,)), ("gr", (Co,)), ("ber", (Co,)), ("slope_in", (1,))):
        _chk(grads.get(k), "grad " + k, shp, optional=True)
    need = layer_bwd_ws_bytes(B, Ci, Co, T, V)
    if ws is None or _bytes(ws) < need:
        raise ValueError(f"workspace too small: need {need} bytes")
    if need_dx and dIn is None:
        dIn = torch.empty_like(x_in)
    _chk(Z, "Z", (B, Ci, T, V), optional=True)
    args = (ptr(x_in), ptr(dU), ptr(A), ptr(Tm), ptr(in_slope), ptr(stat), ptr(Wt), ptr(gt),
            ptr(Wr), ptr(gr), ptr(dIn if need_dx else None), ptr(grads["A"]), ptr(grads["T"]), ptr(grads["Wt"]),
            ptr(grads.get("bt")), ptr(grads["gt"]), ptr(grads["bet"]), ptr(grads.get("Wr")), ptr(grads.get("br")),
            ptr(grads.get("gr")), ptr(grads.get("ber")), ptr(grads.get("slope_in")), ptr(ws),
            ctypes.c_size_t(_bytes(ws)), i32(1 if accumulate else 0), i32(B), i32(Ci), i32(Co), i32(T), i32(V), _stream())
    if stats_in is not None or below is not None:
        if Z is None:
            raise ValueError("layer_bwd: chain mode needs the stored Z")
        sp, srows = stats_in if stats_in is not None else (None, 0)
        _chk(sp, "stats_in", optional=True)
        if sp is not None and sp.numel() < (srows * (2 * Co * Ci + Co) + 1) // 2 * 2 + 2 * (2 * Co * Ci + Co):
            raise ValueError("layer_bwd: stats_in smaller than a chain buffer of that many rows")
        xb, zb, sb, bs = below if below is not None else (None, None, None, None)
        cb = xb.shape[1] if xb is not None else 0
        _chk(xb, "below x", (B, cb, T, V), optional=True); _chk(zb, "below Z", (B, cb, T, V), optional=True); _chk(bs, "below_stats", optional=True)
        _chk(sb, "below in_slope", (1,), optional=True)
        call("coskad_layer_bwd_chain_f32", *args, ptr(Z), ptr(sp), i32(srows), ctypes.c_size_t(_bytes(sp) if sp is not None else 0),
             ptr(xb), ptr(zb), ptr(sb), i32(cb), ptr(bs),
             ctypes.c_size_t(_bytes(bs) if bs is not None else 0), ctypes.c_double(float(stats_count)))
    elif Z is None:
        call("coskad_layer_bwd_f32", *args)
    else:       # stored gcn(PReLU(x_in)) from layer_train_stats(..., Z=...): no mixing recompute in the backward kernels
        call("coskad_layer_bwd_z_f32", *args, ptr(Z))
    return dIn if need_dx else None


def layer_bwd_data(x_in, dU, A, Tm, in_slope, stat, Wt, gt, Wr, gr, grads: dict, ws, dZ, need_dx=True, dIn=None,
                   accumulate=False, Z=None):
    """Stages 1-3 of layer_bwd (everything but dA, dT); dZ [B,Ci,T,V] receives the mixing-output gradient that
    layer_gcn_params consumes (possibly on another stream).  Returns dIn (or None)."""
    B, Ci, T, V = x_in.shape
    Co = Wt.shape[0]
    _chk(x_in, "x_in"); _chk(dU, "dU", (B, Co, T, V)); _chk(A, "A", (T, V, V)); _chk(Tm, "T", (V, T, T))
    _chk(Wt, "Wt", (Co, Ci)); _chk(gt, "gamma_t", (Co,)); _chk(stat, "stat", (stat_floats(Ci, Co),))
    _chk(Wr, "Wr", (Co, Ci), optional=True); _chk(gr, "gamma_r", (Co,), optional=True)
    _chk(in_slope, "in_slope", (1,), optional=True); _chk(dZ, "dZ", (B, Ci, T, V)); _chk(Z, "Z", (B, Ci, T, V), optional=True)
    for k, shp in (("Wt", (Co, Ci)), ("gt", (Co,)), ("bet", (Co,))):
        _chk(grads[k], "grad " + k, shp)
    for k, shp in (("bt", (Co,)), ("Wr", (Co, Ci)), ("br", (Co,)), ("gr", (Co,)), ("ber", (Co,)), ("slope_in", (1,))):
        _chk(grads.get(k), "grad " + k, shp, optional=True)
    need = layer_bwd_ws_bytes(B, Ci, Co, T, V)
    if ws is None or _bytes(ws) < need:
        raise ValueError(f"workspace too small: need {need} bytes")
    if need_dx and dIn is None:
        dIn = torch.empty_like(x_in)
    call("coskad_layer_bwd_data_f32", ptr(x_in), ptr(dU), ptr(A), ptr(Tm), ptr(in_slope), ptr(stat), ptr(Wt), ptr(gt),
         ptr(Wr), ptr(gr), ptr(dIn if need_dx else None), ptr(dZ), ptr(grads["Wt"]), ptr(grads.get("bt")),
         ptr(grads["gt"]), ptr(grads["bet"]), ptr(grads.get("Wr")), ptr(grads.get("br")), ptr(grads.get("gr")),
         ptr(grads.get("ber")), ptr(grads.get("slope_in")), ptr(ws), ctypes.c_size_t(_bytes(ws)),
         i32(1 if accumulate else 0), i32(B), i32(Ci), i32(Co), i32(T), i32(V), _stream(), ptr(Z))
    return dIn if need_dx else None


def layer_gcn_params_ws_bytes(T, V) -> int:
    fn = _lib.lib().coskad_layer_gcn_params_ws_bytes
    fn.restype = ctypes.c_size_t
    return fn(i32(T), i32(V))


def layer_gcn_params(x_in, in_slope, dZ, A, Tm, dA, dT, ws, accumulate=False):
    """dA, dT of one layer from its stored input and dZ (enqueued on the CURRENT torch stream)."""
    B, Ci, T, V = x_in.shape
    _chk(x_in, "x_in"); _chk(dZ, "dZ", (B, Ci, T, V)); _chk(A, "A", (T, V, V)); _chk(Tm, "T", (V, T, T))
    _chk(dA, "dA", (T, V, V)); _chk(dT, "dT", (V, T, T)); _chk(in_slope, "in_slope", (1,), optional=True)
    if ws is None or _bytes(ws) < layer_gcn_params_ws_bytes(T, V):
        raise ValueError("workspace too small")
    call("coskad_layer_gcn_params_f32", ptr(x_in), ptr(in_slope), ptr(dZ), ptr(A), ptr(Tm), ptr(dA), ptr(dT), ptr(ws),
         ctypes.c_size_t(_bytes(ws)), i32(1 if accumulate else 0), i32(B), i32(Ci), i32(T), i32(V), _stream())


def btlnk_fwd(U: Tensor, W: Tensor, bias: Optional[Tensor], slope: Optional[Tensor], ws=None) -> Tensor:
    """z = Linear(flatten(PReLU(U)))  (reference ae.py:97-101).  ws: an engine.Workspace to take the split-K scratch from
    (stream-ordered reuse across the kernels of one module); None allocates it per call."""
    B = U.shape[0]
    K = U.numel() // B
    L = W.shape[0]
    _chk(U, "U"); _chk(W, "W", (L, K)); _chk(bias, "bias", (L,), optional=True); _chk(slope, "slope", (1,), optional=True)
    z = torch.empty(B, L, device=U.device, dtype=torch.float32)
    if B >= BTLNK_SPLITK_MIN_B and K % 16 == 0:
        # blocks of 64 clips x 4 K slices (W operands shared by four clip tiles), fixed-order partial sums
        fn = _lib.lib().coskad_btlnk_fwd_ws_bytes
        fn.restype = ctypes.c_size_t
        nbytes = fn(i32(B))
        buf = ws.get(nbytes, U.device) if ws is not None else torch.empty(nbytes, dtype=torch.uint8, device=U.device)
        call("coskad_btlnk_fwd_ws_f32", ptr(U), ptr(W), ptr(bias), ptr(slope), ptr(z), ptr(buf), ctypes.c_size_t(nbytes),
             i32(B), i32(K), i32(L), _stream())
        return z
    call("coskad_btlnk_fwd_f32", ptr(U), ptr(W), ptr(bias), ptr(slope), ptr(z), i32(B), i32(K), i32(L), _stream())
    return z


BTLNK_SPLITK_MIN_B = 1      # always (K % 16 == 0): a clip's latent must not depend on the batch it arrives in (bit-exact chunking)


def btlnk_bwd_ws_bytes(B, K, L) -> int:
    fn = _lib.lib().coskad_btlnk_bwd_ws_bytes
    fn.restype = ctypes.c_size_t
    return fn(i32(B), i32(K), i32(L))


def btlnk_bwd(U, W, dz, slope, dW, db, dslope, ws, dU=None, accumulate=False):
    B = U.shape[0]
    K = U.numel() // B
    L = W.shape[0]
    _chk(U, "U"); _chk(W, "W", (L, K)); _chk(dz, "dz", (B, L)); _chk(dW, "dW", (L, K))
    _chk(db, "db", (L,), optional=True); _chk(dslope, "dslope", (1,), optional=True); _chk(slope, "slope", (1,), optional=True)
    need = btlnk_bwd_ws_bytes(B, K, L)
    if ws is None or _bytes(ws) < need:
        raise ValueError(f"workspace too small: need {need} bytes")
    if dU is None:
        dU = torch.empty_like(U)
    call("coskad_btlnk_bwd_f32", ptr(U), ptr(W), ptr(dz), ptr(slope), ptr(dU), ptr(dW), ptr(db), ptr(dslope), ptr(ws),
         ctypes.c_size_t(_bytes(ws)), i32(1 if accumulate else 0), i32(B), i32(K), i32(L), _stream())
    return dU


def btlnk_bwd_chain_ok(K: int, TV: int, below_Ci: int) -> bool:
    """coskad_btlnk_bwd_chain_f32 takes the shape: a 64-channel last layer over 16 / 32 input channels."""
    fn = _lib.lib().coskad_btlnk_bwd_chain_ok
    fn.restype = ctypes.c_int
    return bool(fn(i32(K), i32(TV), i32(below_Ci)))


def btlnk_bwd_chain(U, W, dz, slope, dW, db, dslope, ws, below_in, below_Z, below_in_slope, dU=None, accumulate=False):
    """btlnk_bwd AND the top layer's backward batch reductions from one pass (csrc/btlnk_chain.hip).
    U [B, 64, T, V]: the top layer's pre-activation; below_in / below_Z [B, Ci, T, V]: that layer's input (pre-activation of the
    layer below + its slope) and stored gcn output.  -> (dU, (chain buffer, rows)): the second item is layer_bwd's `stats_in`.
    `ws`: an engine.Workspace."""
    B, Ch, T, V = U.shape
    TV = T * V
    K = Ch * TV
    L = W.shape[0]
    Ci = below_in.shape[1]
    _chk(U, "U"); _chk(W, "W", (L, K)); _chk(dz, "dz", (B, L)); _chk(dW, "dW", (L, K))
    _chk(db, "db", (L,), optional=True); _chk(dslope, "dslope", (1,), optional=True); _chk(slope, "slope", (1,), optional=True)
    _chk(below_in, "below_in", (B, Ci, T, V)); _chk(below_Z, "below_Z", (B, Ci, T, V)); _chk(below_in_slope, "below_in_slope", (1,), optional=True)
    if not btlnk_bwd_chain_ok(K, TV, Ci):
        raise ValueError(f"btlnk_bwd_chain: shape K={K} TV={TV} Ci={Ci} not supported")
    lib = _lib.lib()
    lib.coskad_btlnk_bwd_chain_ws_bytes.restype = ctypes.c_size_t
    lib.coskad_btlnk_bwd_chain_floats.restype = ctypes.c_size_t
    need = lib.coskad_btlnk_bwd_chain_ws_bytes(i32(B), i32(K), i32(L), i32(TV))
    buf = ws.get(need, U.device)
    stats = torch.empty(lib.coskad_btlnk_bwd_chain_floats(i32(B), i32(TV), i32(Ci)), device=U.device, dtype=torch.float32)
    if dU is None:
        dU = torch.empty_like(U)
    rows = ctypes.c_int(0)
    call("coskad_btlnk_bwd_chain_f32", ptr(U), ptr(W), ptr(dz), ptr(slope), ptr(dU), ptr(dW), ptr(db), ptr(dslope), ptr(buf),
         ctypes.c_size_t(_bytes(buf)), i32(1 if accumulate else 0), i32(B), i32(K), i32(L), ptr(below_in), ptr(below_Z),
         ptr(below_in_slope), i32(Ci), i32(TV), ptr(stats), ctypes.c_size_t(_bytes(stats)), ctypes.byref(rows), _stream())
    return dU, (stats, rows.value)


def _cuda_f32(t: Tensor, name: str) -> None:
    if not t.is_cuda:
        raise _lib.CoskadHipError(f"{name}: expected a CUDA (ROCm) tensor, got device {t.device}; no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")


def _b3(t: Tensor):
    """(tensor viewed as [batch, rows, cols], element strides) -- 2-D operands broadcast over the batch (stride 0)."""
    if t.dim() == 2:
        return 1, t.shape[0], t.shape[1], 0, t.stride(0), t.stride(1)
    if t.dim() == 3:
        return t.shape[0], t.shape[1], t.shape[2], t.stride(0), t.stride(1), t.stride(2)
    raise ValueError("gemm operands are 2-D or 3-D")


def gemm(A: Tensor, B: Tensor, out: Optional[Tensor] = None, bias: Optional[Tensor] = None, bias_mode: int = 0,
         bias_mod: int = 1, relu: bool = False, accumulate: bool = False) -> Tensor:
    """C[b] = act(A[b] @ B[b] + bias) on the fp32 MFMA GEMM kernel; A [.., M, K], B [.., K, N] may be ANY strided views
    (transposes, broadcast batch): nothing is copied."""
    _cuda_f32(A, "A"); _cuda_f32(B, "B")
    ba, M, K, sab, sam, sak = _b3(A)
    bb, K2, N, sbb, sbk, sbn = _b3(B)
    if K != K2:
        raise ValueError(f"gemm: inner sizes differ ({K} vs {K2})")
    batch = max(ba, bb)
    if ba not in (1, batch) or bb not in (1, batch):
        raise ValueError("gemm: batch sizes do not broadcast")
    if ba == 1:
        sab = 0
    if bb == 1:
        sbb = 0
    if out is None:
        out = torch.empty((batch, M, N) if (A.dim() == 3 or B.dim() == 3) else (M, N), device=A.device, dtype=torch.float32)
    _cuda_f32(out, "out")
    bc, Mc, Nc, scb, scm, scn = _b3(out)
    if (Mc, Nc) != (M, N) or bc != batch:
        raise ValueError(f"gemm: out has shape {tuple(out.shape)}, expected batch {batch} x {M} x {N}")
    _chk(bias, "bias", optional=True)
    ll = ctypes.c_longlong
    call("coskad_gemm_f32", ptr(A), ptr(B), ptr(out), ptr(bias), ll(sab), ll(sam), ll(sak), ll(sbb), ll(sbk), ll(sbn),
         ll(scb), ll(scm), ll(scn), i32(M), i32(N), i32(K), i32(batch), i32(bias_mode), i32(bias_mod), i32(1 if relu else 0),
         i32(0), i32(0), ll(0), i32(1 if accumulate else 0), _stream())
    return out


def gemm_reduce(A: Tensor, B: Tensor, out: Tensor, target_chunks: int = 64, ktotal: int = 0, accumulate: bool = False) -> Tensor:
    """out (+)= sum_b A[b] @ B[b] (weight gradients): partial sums per batch chunk on the GEMM kernel, then a fixed-order
    fp64 sum.  ktotal > 0: the reduction axis is one long axis cut into `batch` pieces of K (the last one partial)."""
    _cuda_f32(A, "A"); _cuda_f32(B, "B"); _chk(out, "out")
    ba, M, K, sab, sam, sak = _b3(A)
    bb, K2, N, sbb, sbk, sbn = _b3(B)
    if K != K2 or ba != bb or tuple(out.shape) != (M, N):
        raise ValueError("gemm_reduce: shape mismatch")
    chunk = max(1, (ba + target_chunks - 1) // target_chunks)
    chunks = (ba + chunk - 1) // chunk
    part = torch.empty(chunks, M, N, device=A.device, dtype=torch.float32)
    ll = ctypes.c_longlong
    call("coskad_gemm_f32", ptr(A), ptr(B), ptr(part), ptr(None), ll(sab), ll(sam), ll(sak), ll(sbb), ll(sbk), ll(sbn),
         ll(0), ll(0), ll(0), i32(M), i32(N), i32(K), i32(ba), i32(0), i32(1), i32(0), i32(1), i32(chunk), ll(ktotal), i32(0), _stream())
    call("coskad_gemm_sum_f32", ptr(part), i32(chunks), ctypes.c_size_t(M * N), ptr(out), i32(1 if accumulate else 0), _stream())
    return out


def gemm_rows_outer(G: Tensor, S: Tensor, out: Tensor, rows_per_piece: int = 256, target_chunks: int = 64,
                    accumulate: bool = False) -> Tensor:
    """out[m, n] (+)= sum_r G[r, m] * S[r, n] for row-major G [R, M], S [R, N]: the long reduction axis r is cut into pieces
    of `rows_per_piece` (the kernel's batch axis, the last piece guarded by ktotal = R), pieces are summed per chunk."""
    R, M = G.shape
    R2, N = S.shape
    _chk(G, "G"); _chk(S, "S"); _chk(out, "out", (M, N))
    if R != R2:
        raise ValueError("gemm_rows_outer: row counts differ")
    nb = (R + rows_per_piece - 1) // rows_per_piece
    chunk = max(1, (nb + target_chunks - 1) // target_chunks)
    chunks = (nb + chunk - 1) // chunk
    part = torch.empty(chunks, M, N, device=G.device, dtype=torch.float32)
    ll = ctypes.c_longlong
    call("coskad_gemm_f32", ptr(G), ptr(S), ptr(part), ptr(None), ll(rows_per_piece * M), ll(1), ll(M),
         ll(rows_per_piece * N), ll(N), ll(1), ll(0), ll(0), ll(0), i32(M), i32(N), i32(rows_per_piece), i32(nb), i32(0), i32(1),
         i32(0), i32(1), i32(chunk), ll(R), i32(0), _stream())
    call("coskad_gemm_sum_f32", ptr(part), i32(chunks), ctypes.c_size_t(M * N), ptr(out), i32(1 if accumulate else 0), _stream())
    return out


def conv1x1_ok(M: int, K: int, P: int) -> bool:
    fn = _lib.lib().coskad_conv1x1_ok
    fn.restype = ctypes.c_int
    return bool(fn(i32(M), i32(K), i32(P)))


def conv1x1(W: Tensor, x: Tensor, bias: Optional[Tensor] = None, out: Optional[Tensor] = None, accumulate: bool = False,
            want_stats: bool = False):
    """out[b] (+)= W @ x[b] (+ bias) for W [M, K] (any 2-D view contiguous along one axis: pass `w.t()` for the data gradient),
    x [B, K, P] contiguous -> (out [B, M, P], stats partials or None).  Shapes outside csrc/conv1x1.hip go to the strided GEMM."""
    M, K = W.shape
    B, K2, P = x.shape
    _cuda_f32(W, "W"); _chk(x, "x"); _chk(bias, "bias", (M,), optional=True)
    if K != K2:
        raise ValueError(f"conv1x1: weight has {K} input channels, x has {K2}")
    sm, sk = W.stride(0), W.stride(1)
    fast = conv1x1_ok(M, K, P) and (sk == 1 or sm == 1) and (sm % 4 == 0 if sk == 1 else sk % 4 == 0) and W.data_ptr() % 16 == 0 \
        and x.data_ptr() % 16 == 0
    if out is None:
        out = torch.empty(B, M, P, device=x.device, dtype=torch.float32)
    _chk(out, "out", (B, M, P))
    if not fast or out.data_ptr() % 16 != 0:
        gemm(W, x, out=out, bias=bias, bias_mode=1 if bias is not None else 0, bias_mod=M, accumulate=accumulate)
        return out, None
    parts = None
    if want_stats:
        fn = _lib.lib().coskad_conv1x1_stat_rows
        fn.restype = ctypes.c_int
        rows = fn(i32(M), i32(K), i32(P), i32(B))
        parts = torch.empty(rows, M, 2, device=x.device, dtype=torch.float64)
    ll = ctypes.c_longlong
    call("coskad_conv1x1_f32", ptr(W), ll(sm), ll(sk), ptr(x), ptr(out), ptr(bias), ptr(parts), i32(M), i32(K), i32(P), i32(B),
         i32(1 if accumulate else 0), _stream())
    return out, parts


WGRAD_BLOCKS = 256   # workgroups a weight-gradient launch aims for (one per CU; sweep 64..2048: wide step 19.7 / 18.0 / 18.2 / 18.7 / 19.5 ms at 64 / 256 / 512 / 1024 / 2048)


def conv1x1_wgrad(G: Tensor, x: Tensor, out: Tensor, target_chunks: int = 64, accumulate: bool = False) -> Tensor:
    """out[m, k] (+)= sum_b sum_p G[b, m, p] x[b, k, p]: the weight gradient of a 1x1 convolution (G [B, M, P], x [B, K, P]).
    csrc/conv1x1.hip where the shape allows, the strided GEMM's chunked reduction otherwise; deterministic either way."""
    B, M, P = G.shape
    K = x.shape[1]
    _chk(G, "G"); _chk(x, "x", (B, K, P)); _chk(out, "out", (M, K))
    fn = _lib.lib().coskad_conv1x1_wgrad_ok
    fn.restype = ctypes.c_int
    if not fn(i32(M), i32(K), i32(P)) or G.data_ptr() % 16 or x.data_ptr() % 16:
        return gemm_reduce(G, x.transpose(1, 2), out, target_chunks=target_chunks, accumulate=accumulate)
    # the kernel's grid is (K tiles, M tiles, clip chunks): enough chunks for ~WGRAD_BLOCKS workgroups (narrow layers have ONE tile)
    tm, tk = (128, 128) if (M % 128 == 0 and K % 128 == 0) else ((64, 64) if M % 64 == 0 else (32, 64))
    tiles = (M // tm) * (K // tk)
    target_chunks = max(target_chunks, -(-WGRAD_BLOCKS // tiles))
    chunk = max(1, (B + target_chunks - 1) // target_chunks)
    chunks = (B + chunk - 1) // chunk
    part = torch.empty(chunks, M, K, device=G.device, dtype=torch.float32)
    call("coskad_conv1x1_wgrad_f32", ptr(G), ptr(x), ptr(part), i32(M), i32(K), i32(P), i32(B), i32(chunk), _stream())
    call("coskad_gemm_sum_f32", ptr(part), i32(chunks), ctypes.c_size_t(M * K), ptr(out), i32(1 if accumulate else 0), _stream())
    return out


def bn_momentum(bn) -> float:
    """nn.BatchNorm's `exponential_average_factor` for ONE training-mode forward (torch/nn/modules/batchnorm.py, _BatchNorm.forward):
    the fixed `momentum`, or with momentum=None the cumulative moving average 1 / num_batches_tracked, counted after this batch.  The
    kernels increment the device counter through its raw pointer; the host keeps a mirror of it so that no forward waits for the
    device -- re-read (one .item()) whenever torch itself wrote the tensor (load_state_dict, fill_: its version counter moves).
    Call exactly once per BatchNorm and training forward."""
    if bn.momentum is not None:
        return float(bn.momentum)
    nbt = bn.num_batches_tracked
    if nbt is None:                       # track_running_stats=False: nothing to average
        return 0.0
    st = bn.__dict__.get("_coskad_nbt")
    n = st[2] if (st is not None and st[0] is nbt and st[1] == nbt._version) else int(nbt.item())
    bn.__dict__["_coskad_nbt"] = (nbt, nbt._version, n + 1)
    return 1.0 / (n + 1)


def bn_batch_stats(bn, training: bool) -> bool:
    """does this BatchNorm normalise with batch statistics in this mode? (track_running_stats=False: always, batchnorm.py's `bn_training`)"""
    return bool(training) or bn.running_mean is None or bn.running_var is None


def bn2_stats_parts(parts: Tensor, bn, count: int) -> Tensor:
    """Train-mode statistics of nn.BatchNorm2d `bn` from the partial sums a conv1x1 epilogue wrote (+ running update)."""
    rows, C, _ = parts.shape
    _chk(parts, "parts", (rows, C, 2), dtype=torch.float64)
    stat = torch.empty(2 * C, device=parts.device, dtype=torch.float32)
    call("coskad_bn2_stats_parts_f32", ptr(parts), i32(rows), ptr(stat), ptr(bn.running_mean), ptr(bn.running_var),
         ptr(bn.num_batches_tracked), ctypes.c_float(bn_momentum(bn)), ctypes.c_float(bn.eps),
         ctypes.c_double(float(count)), i32(C), _stream())
    return stat


def relu_bwd(out: Tensor, dout: Tensor, dbias: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """g = dout * (out > 0) on [N, C, P]; dbias[c] (+)= sum of g over (n, p)."""
    Nb, C, P = out.shape
    _chk(out, "out"); _chk(dout, "dout", (Nb, C, P)); _chk(dbias, "dbias", (C,), optional=True)
    g = torch.empty_like(out)
    slices = min(Nb, 64)
    part = torch.empty(slices, C, device=out.device, dtype=torch.float32)
    call("coskad_relu_bwd_f32", ptr(out), ptr(dout), ptr(g), ptr(part), i32(Nb), i32(C), i32(P), i32(slices), _stream())
    if dbias is not None:
        call("coskad_gemm_sum_f32", ptr(part), i32(slices), ctypes.c_size_t(C), ptr(dbias), i32(1 if accumulate else 0), _stream())
    return g


def softmax_rows(x: Tensor) -> Tensor:
    n = x.shape[0]
    _chk(x, "x", (n, n))
    y = torch.empty_like(x)
    call("coskad_softmax_rows_f32", ptr(x), ptr(y), i32(n), _stream())
    return y


def softmax_rows_bwd(y: Tensor, dy: Tensor) -> Tensor:
    n = y.shape[0]
    _chk(y, "y", (n, n)); _chk(dy, "dy", (n, n))
    dx = torch.empty_like(y)
    call("coskad_softmax_rows_bwd_f32", ptr(y), ptr(dy), ptr(dx), i32(n), _stream())
    return dx


def _bn2_ws(Nb: int, C: int, device, bwd: bool = False) -> Tensor:
    fn = getattr(_lib.lib(), "coskad_bn2_bwd_ws_bytes" if bwd else "coskad_bn2_ws_bytes")
    fn.restype = ctypes.c_size_t
    return torch.empty(fn(i32(Nb), i32(C)), dtype=torch.uint8, device=device)


def bn2_stats(x: Tensor, bn, training: bool) -> Tensor:
    """Per-channel (mean, invstd) of x [N, C, P] for nn.BatchNorm2d `bn`: batch statistics (+ running update) in training
    mode, the running statistics otherwise."""
    Nb, C, P = x.shape
    _chk(x, "x")
    stat = torch.empty(2 * C, device=x.device, dtype=torch.float32)
    ws = _bn2_ws(Nb, C, x.device)
    call("coskad_bn2_stats_f32", ptr(x), ptr(stat), ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked),
         ctypes.c_float(bn_momentum(bn) if training else 0.0), ctypes.c_float(bn.eps), i32(1 if training else 0),
         ptr(ws), ctypes.c_size_t(ws.numel()), i32(Nb), i32(C), i32(P), _stream())
    return stat


def bn2_apply_prelu(Ct, Cr, stat_t, gt, bt, stat_r, gr, br, slope, drop_p: float = 0.0, drop_seed: int = 0) -> Tensor:
    """out = PReLU(Dropout_p(BN_t(Ct)) + BN_r(Cr)); the dropout mask is a counter-based function of (drop_seed, element)."""
    Nb, C, P = Ct.shape
    _chk(Ct, "Ct"); _chk(Cr, "Cr", (Nb, C, P)); _chk(stat_t, "stat_t", (2 * C,)); _chk(stat_r, "stat_r", (2 * C,), optional=True)
    out = torch.empty_like(Ct)
    call("coskad_bn2_apply_prelu_f32", ptr(Ct), ptr(Cr), ptr(stat_t), ptr(gt), ptr(bt), ptr(stat_r), ptr(gr), ptr(br), ptr(slope),
         ptr(out), i32(Nb), i32(C), i32(P), _stream(), ctypes.c_float(drop_p), ctypes.c_ulonglong(drop_seed))
    return out


def dropout_mask(shape, drop_p: float, drop_seed: int, device) -> Tensor:
    """The mask bn2_apply_prelu / bn2_bwd apply for (drop_p, drop_seed): values 0 or 1 / (1 - p), in element order."""
    out = torch.empty(shape, device=device, dtype=torch.float32)
    call("coskad_dropout_mask_f32", ptr(out), ctypes.c_size_t(out.numel()), ctypes.c_float(drop_p), ctypes.c_ulonglong(drop_seed),
         _stream())
    return out


def bn2_bwd(Ct, Cr, dOut, stat_t, gt, bt, stat_r, gr, br, slope, training: bool, drop_p: float = 0.0, drop_seed: int = 0,
            into: Optional[dict] = None):
    """-> (dCt, dCr, dgt, dbt, dgr, dbr, dslope) (residual gradients None for an identity residual).  `into`: destinations for the
    parameter gradients ('gt', 'bt', 'gr', 'br', 'slope': e.g. views of a flat gradient buffer) instead of fresh tensors."""
    Nb, C, P = Ct.shape
    _chk(dOut, "dOut", (Nb, C, P))
    into = into or {}
    dCt, dCr = torch.empty_like(Ct), torch.empty_like(Ct)
    dgt = into.get("gt") if into.get("gt") is not None else torch.empty_like(gt)
    dbt = into.get("bt") if into.get("bt") is not None else torch.empty_like(bt)
    dgr = (into.get("gr") if into.get("gr") is not None else torch.empty_like(gr)) if stat_r is not None else None
    dbr = (into.get("br") if into.get("br") is not None else torch.empty_like(br)) if stat_r is not None else None
    dslope = into.get("slope") if into.get("slope") is not None else torch.empty(1, device=Ct.device, dtype=torch.float32)
    _chk(dgt, "dgt", (C,)); _chk(dbt, "dbt", (C,)); _chk(dgr, "dgr", (C,), optional=True); _chk(dbr, "dbr", (C,), optional=True)
    _chk(dslope, "dslope", (1,))
    ws = _bn2_ws(Nb, C, Ct.device, bwd=True)
    call("coskad_bn2_bwd_f32", ptr(Ct), ptr(Cr), ptr(dOut), ptr(stat_t), ptr(gt), ptr(bt), ptr(stat_r), ptr(gr), ptr(br), ptr(slope),
         ptr(dCt), ptr(dCr), ptr(dgt), ptr(dbt), ptr(dgr), ptr(dbr), ptr(dslope), i32(1 if training else 0), ptr(ws),
         ctypes.c_size_t(ws.numel()), i32(Nb), i32(C), i32(P), _stream(), ctypes.c_float(drop_p), ctypes.c_ulonglong(drop_seed))
    return dCt, dCr, dgt, dbt, dgr, dbr, dslope


def mlp_head_ws_floats(B: int, H: int, L: int) -> int:
    fn = _lib.lib().coskad_mlp_head_ws_floats
    fn.restype = ctypes.c_size_t
    return fn(i32(B), i32(H), i32(L))


def mlp_head_fwd(y1, gamma, beta, running_mean, running_var, nbt, W2, b2, training: bool, momentum: float = 0.1,
                 eps: float = 1e-5):
    """z = W2 . relu(BatchNorm1d(y1)) + b2 (the `mlp` projector behind its first Linear, components.py:209-226).
    -> (z [B,L], stat: [0:2H] = mean, invstd (the kernels' partial sums behind them)).  Train mode updates the running statistics."""
    B, H = y1.shape
    L = W2.shape[0]
    _chk(y1, "y1"); _chk(gamma, "gamma", (H,)); _chk(beta, "beta", (H,)); _chk(W2, "W2", (L, H)); _chk(b2, "b2", (L,), optional=True)
    _chk(running_mean, "running_mean", (H,), optional=True); _chk(running_var, "running_var", (H,), optional=True)
    _chk(nbt, "num_batches_tracked", (), dtype=torch.int64, optional=True)
    z = torch.empty(B, L, device=y1.device, dtype=torch.float32)
    stat = torch.empty(mlp_head_ws_floats(B, H, L), device=y1.device, dtype=torch.float32)
    call("coskad_mlp_head_fwd_f32", ptr(y1), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), ptr(nbt),
         ctypes.c_float(momentum), ctypes.c_float(eps), i32(1 if training else 0), ptr(W2), ptr(b2), ptr(z), ptr(stat),
         i32(B), i32(H), i32(L), _stream())
    return z, stat


def mlp_head_bwd(y1, stat, gamma, beta, W2, dz, grads: dict, training: bool, accumulate: bool = False):
    """-> dy1 [B,H]; fills grads['gamma'], grads['beta'], grads['W2'] and (optional) grads['b2']."""
    B, H = y1.shape
    L = W2.shape[0]
    _chk(y1, "y1"); _chk(stat, "stat"); _chk(gamma, "gamma", (H,)); _chk(beta, "beta", (H,)); _chk(W2, "W2", (L, H))
    if stat.numel() < 2 * H:
        raise ValueError("mlp_head_bwd: stat holds fewer than 2 H values")
    _chk(dz, "dz", (B, L)); _chk(grads["gamma"], "dgamma", (H,)); _chk(grads["beta"], "dbeta", (H,)); _chk(grads["W2"], "dW2", (L, H))
    _chk(grads.get("b2"), "db2", (L,), optional=True)
    dy1 = torch.empty_like(y1)
    red = torch.empty(mlp_head_ws_floats(B, H, L), device=y1.device, dtype=torch.float32)
    call("coskad_mlp_head_bwd_f32", ptr(y1), ptr(stat), ptr(gamma), ptr(beta), ptr(W2), ptr(dz), ptr(dy1), ptr(grads["gamma"]),
         ptr(grads["beta"]), ptr(grads["W2"]), ptr(grads.get("b2")), ptr(red), i32(1 if training else 0),
         i32(1 if accumulate else 0), i32(B), i32(H), i32(L), _stream())
    return dy1


HEAD_SLOTS = 19


def head_ws(B: int, device) -> Tensor:
    fn = _lib.lib().coskad_head_ws_floats
    fn.restype = ctypes.c_size_t
    return torch.empty(fn(i32(B)), device=device, dtype=torch.float32)


def mse_head(z, c, need_grad=True, need_score=False, acc=None, upstream=1.0, ws=None):
    """-> (stats[19], dz or None, score or None).  stats[0] = F.mse_loss(z, c)."""
    B, L = z.shape
    _chk(z, "z"); _chk(c, "c", (L,)); _chk(acc, "acc", (HEAD_SLOTS,), optional=True)
    ws = head_ws(B, z.device) if ws is None else ws
    dz = torch.empty_like(z) if need_grad else None
    score = torch.empty(B, device=z.device, dtype=torch.float32) if need_score else None
    stats = torch.empty(HEAD_SLOTS, device=z.device, dtype=torch.float32)
    call("coskad_mse_head_f32", ptr(z), ptr(c), ptr(dz), ptr(score), ptr(stats), ptr(acc), ctypes.c_float(upstream),
         ptr(ws), i32(B), i32(L), _stream())
    return stats, dz, score


def mahalanobis_head(z, c, VI, need_grad=True, need_score=False, acc=None, gram=None, gram_accumulate=True,
                     upstream=1.0, ws=None):
    """-> (stats[19], dz or None, score or None).  stats[0] = mahalanobis(z, c, VI) (mean); gram (+)= sum z z^T."""
    B, L = z.shape
    _chk(z, "z"); _chk(c, "c", (L,)); _chk(VI, "VI", (L, L)); _chk(acc, "acc", (HEAD_SLOTS,), optional=True)
    _chk(gram, "gram", (L, L), optional=True)
    ws = head_ws(B, z.device) if ws is None else ws
    dz = torch.empty_like(z) if need_grad else None
    score = torch.empty(B, device=z.device, dtype=torch.float32) if need_score else None
    stats = torch.empty(HEAD_SLOTS, device=z.device, dtype=torch.float32)
    call("coskad_mahalanobis_head_f32", ptr(z), ptr(c), ptr(VI), ptr(dz), ptr(score), ptr(stats), ptr(acc), ptr(gram),
         i32(int(gram_accumulate)), ctypes.c_float(upstream), ptr(ws), i32(B), i32(L), _stream())
    return stats, dz, score


def poincare_head(z, c, need_grad=True, need_zh=False, need_score=False, acc=None, upstream=1.0, ws=None):
    """-> (stats[19], dz, zh, score).  stats[0] = dist(c, project(expmap0(z))).mean()."""
    B, L = z.shape
    _chk(z, "z"); _chk(c, "c", (L,), optional=True); _chk(acc, "acc", (HEAD_SLOTS,), optional=True)
    ws = head_ws(B, z.device) if ws is None else ws
    dz = torch.empty_like(z) if (need_grad and c is not None) else None
    zh = torch.empty_like(z) if need_zh else None
    score = torch.empty(B, device=z.device, dtype=torch.float32) if (need_score and c is not None) else None
    stats = torch.empty(HEAD_SLOTS, device=z.device, dtype=torch.float32)
    call("coskad_poincare_head_f32", ptr(z), ptr(c), ptr(dz), ptr(zh), ptr(score), ptr(stats), ptr(acc),
         ctypes.c_float(upstream), ptr(ws), i32(B), i32(L), _stream())
    return stats, dz, zh, score


def poincare_dist(zh, c):
    B, L = zh.shape
    _chk(zh, "zh"); _chk(c, "c", (L,))
    score = torch.empty(B, device=zh.device, dtype=torch.float32)
    call("coskad_poincare_dist_f32", ptr(zh), ptr(c), ptr(score), i32(B), i32(L), _stream())
    return score


def poincare_logmap0(y):
    """logmap0 on the unit Poincare ball (reference utils/hyper_math.py:367-370, c = 1): tangent vector at the origin."""
    B, L = y.shape
    _chk(y, "y")
    out = torch.empty_like(y)
    call("coskad_poincare_logmap0_f32", ptr(y), ptr(out), i32(B), i32(L), _stream())
    return out


def lowrank_fold_ok(latent: int, TV: int) -> bool:
    return bool(_lib.lib().coskad_lowrank_fold_ok(i32(latent), i32(TV)))


def lowrank_fold_fwd(X: Tensor, G: Tensor, bn_t, bn_r, bias_t, bias_r, n_pos: float):
    """csrc/lowrank_fold.hip: X [2, 9, Co, TV] fp32, G [9, 9] fp64 -> (Mw [Co TV, 8], Mb [Co TV], ctx); updates the running statistics of
    the two BatchNorm modules (call once per training forward)."""
    R, K, Co, TV = X.shape
    _chk(X, "X", (2, 9, Co, TV)); _chk(G, "G", (9, 9), dtype=torch.float64)
    dev = X.device
    Mw = torch.empty(Co * TV, K - 1, device=dev, dtype=torch.float32)
    Mb = torch.empty(Co * TV, device=dev, dtype=torch.float32)
    saved = torch.empty(3, 2, Co, device=dev, dtype=torch.float64)
    xbar = torch.empty(2, K, Co, device=dev, dtype=torch.float64)
    xx = torch.empty(2, Co, K, K, device=dev, dtype=torch.float64)
    args = []
    for bn, cb in ((bn_t, bias_t), (bn_r, bias_r)):
        args += [ptr(bn.weight), ptr(bn.bias), ptr(cb), ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked),
                 ctypes.c_float(bn_momentum(bn)), ctypes.c_float(bn.eps)]
    call("coskad_lowrank_fold_fwd_f32", ptr(X), ptr(G), *args, ctypes.c_double(n_pos), ptr(Mw), ptr(Mb), ptr(saved), ptr(xbar), ptr(xx),
         i32(Co), i32(TV), _stream())
    return Mw, Mb, (saved, xbar, xx)


def lowrank_fold_bwd(X: Tensor, G: Tensor, dMw: Tensor, dMb: Tensor, ctx, gamma_t: Tensor, gamma_r: Tensor, n_pos: float):
    """-> (dX [2, 9, Co, TV], dgamma [2, Co], dbeta [Co], dG [9, 9] fp64)"""
    R, K, Co, TV = X.shape
    saved, xbar, xx = ctx
    _chk(dMw, "dMw", (Co * TV, K - 1)); _chk(dMb, "dMb", (Co * TV,))
    dev = X.device
    dX = torch.empty_like(X)
    dgamma = torch.empty(2, Co, device=dev, dtype=torch.float32)
    dbeta = torch.empty(Co, device=dev, dtype=torch.float32)
    dGc = torch.empty(Co, K, K, device=dev, dtype=torch.float64)
    call("coskad_lowrank_fold_bwd_f32", ptr(X), ptr(G), ptr(dMw), ptr(dMb), ptr(saved), ptr(xbar), ptr(xx), ptr(gamma_t), ptr(gamma_r),
         ctypes.c_double(n_pos), ptr(dX), ptr(dgamma), ptr(dbeta), ptr(dGc), i32(Co), i32(TV), _stream())
    return dX, dgamma, dbeta, dGc.sum(0)


def narrow_conv_ok(Ci: int, J: int, TV: int) -> bool:
    return Ci in (16, 32, 64) and J in (2, 4, 6, 8) and TV % 4 == 0


def narrow_conv_fwd(U: Tensor, in_slope: Optional[Tensor], W: Tensor) -> Tensor:
    """[Y; R] = W . PReLU(U): U [B, Ci, T, V] (pre-activation when in_slope is given), W [J, Ci] -> [B, J, T, V] (csrc/last_layer.hip)"""
    B, Ci, T, V = U.shape
    J = W.shape[0]
    _chk(U, "U"); _chk(W, "W", (J, Ci)); _chk(in_slope, "in_slope", (1,), optional=True)
    out = torch.empty(B, J, T, V, device=U.device, dtype=torch.float32)
    call("coskad_narrow_conv_fwd_f32", ptr(U), ptr(in_slope), ptr(W), ptr(out), i32(B), i32(Ci), i32(J), i32(T * V), _stream())
    return out


def narrow_conv_bwd(U: Tensor, in_slope: Optional[Tensor], W: Tensor, dOut: Tensor):
    """-> (dU [B, Ci, T, V], sums [J Ci + 1]): dU = (W^T dOut) PReLU'(U); sums[:J Ci] = dW (row-major [J, Ci]), sums[-1] = the producer's
    slope gradient (0 without in_slope)."""
    B, Ci, T, V = U.shape
    J = W.shape[0]
    _chk(U, "U"); _chk(W, "W", (J, Ci)); _chk(dOut, "dOut", (B, J, T, V)); _chk(in_slope, "in_slope", (1,), optional=True)
    rows = _lib.lib().coskad_narrow_conv_rows(i32(B), i32(T * V))
    E = J * Ci + 1
    part = torch.empty(rows, E, device=U.device, dtype=torch.float32)
    dU = torch.empty_like(U)
    call("coskad_narrow_conv_bwd_f32", ptr(U), ptr(in_slope), ptr(W), ptr(dOut), ptr(dU), ptr(part), ctypes.c_size_t(part.numel()),
         i32(B), i32(Ci), i32(J), i32(T * V), _stream())
    sums = torch.empty(E, device=U.device, dtype=torch.float32)
    call("coskad_gemm_sum_f32", ptr(part), i32(rows), ctypes.c_size_t(E), ptr(sums), i32(0), _stream())
    return dU, sums


def commute_ok(T: int, V: int, Ci: int, Co: int) -> bool:
    """the (C_in -> C_out) layers csrc/commute_layer.hip runs by commutation (convolutions first, mixing on C_out channels)"""
    return bool(_lib.lib().coskad_commute_ok(i32(T), i32(V), i32(Ci), i32(Co)))


_commute_ws: dict = {}


def _commute_scratch(B: int, T: int, V: int, dev) -> Tensor:
    lib = _lib.lib()
    lib.coskad_commute_ws_floats.restype = ctypes.c_size_t
    n = int(lib.coskad_commute_ws_floats(i32(B), i32(T), i32(V)))
    key = (str(dev), T, V)
    ws = _commute_ws.get(key)
    if ws is None or ws.numel() < n:
        ws = _commute_ws[key] = torch.empty(n, device=dev, dtype=torch.float32)
    return ws


def commute_fwd(U_prev: Tensor, in_slope: Optional[Tensor], Wt: Tensor, Wr: Tensor, A: Tensor, Tm: Tensor, gamma_t: Tensor, beta_t: Tensor,
                gamma_r: Tensor, beta_r: Tensor, bias_t: Optional[Tensor], bias_r: Optional[Tensor], rm_t: Optional[Tensor],
                rv_t: Optional[Tensor], rm_r: Optional[Tensor], rv_r: Optional[Tensor], nbt_t: Optional[Tensor],
                nbt_r: Optional[Tensor], momentum: float, eps: float, next_layer=None, slope_out: Optional[Tensor] = None):
    """Training-mode forward of a 32 -> 16 ST_GCNN layer by commutation (csrc/commute_layer.hip; reference
    models/graph_layers/stsgcn.py:94-116): U_prev [B, 32, T, V] (pre-activation when in_slope is given), Wt / Wr [16, 32(, 1, 1)] the
    two convolutions' weights -> (U [B, 16, T, V] pre-activation output, saved, pending).  next_layer = (A, T) of the layer behind
    (16 input channels) with slope_out = this layer's PReLU weight: its statistics pass rides on the last kernel and pending =
    (Z_next, partials, rows) is what engine.chain_forward takes as `pending0`; else pending is None."""
    B, Ci, T, V = U_prev.shape
    _chk(U_prev, "U_prev"); _chk(in_slope, "in_slope", (1,), optional=True)
    for n, t in (("Wt", Wt), ("Wr", Wr)):
        _cuda_f32(t, n)
        if t.numel() != 512 or not t.is_contiguous():
            raise ValueError(f"{n}: expected a contiguous [16, 32] weight")
    _chk(A, "A", (T, V, V)); _chk(Tm, "T", (V, T, T))
    for n, t in (("gamma_t", gamma_t), ("beta_t", beta_t), ("gamma_r", gamma_r), ("beta_r", beta_r)):
        _chk(t, n, (16,))
    for n, t in (("bias_t", bias_t), ("bias_r", bias_r), ("rm_t", rm_t), ("rv_t", rv_t), ("rm_r", rm_r), ("rv_r", rv_r)):
        _chk(t, n, (16,), optional=True)
    dev = U_prev.device
    YR = torch.empty(B, 32, T, V, device=dev, dtype=torch.float32)
    Zy = torch.empty(B, 16, T, V, device=dev, dtype=torch.float32)
    U = torch.empty(B, 16, T, V, device=dev, dtype=torch.float32)
    stat = torch.empty(128, device=dev, dtype=torch.float32)
    ws = _commute_scratch(B, T, V, dev)
    An = Tn = Zn = pn = None
    rows = ctypes.c_int(0)
    if next_layer is not None:
        An, Tn = next_layer
        _chk(An, "A_next", (T, V, V)); _chk(Tn, "T_next", (V, T, T)); _chk(slope_out, "slope_out", (1,))
        Zn = torch.empty(B, 16, T, V, device=dev, dtype=torch.float32)
        pn = torch.empty(768 * 2 * (16 * 16 + 16), device=dev, dtype=torch.float32)
    call("coskad_commute_fwd_f32", ptr(U_prev), ptr(in_slope), ptr(Wt), ptr(Wr), ptr(A), ptr(Tm), ptr(gamma_t), ptr(beta_t),
         ptr(gamma_r), ptr(beta_r), ptr(bias_t), ptr(bias_r), ptr(rm_t), ptr(rv_t), ptr(rm_r), ptr(rv_r), ptr(nbt_t), ptr(nbt_r),
         ctypes.c_float(momentum), ctypes.c_float(eps), ptr(YR), ptr(Zy), ptr(U), ptr(stat), ptr(ws), ctypes.c_size_t(ws.numel()),
         ptr(An), ptr(Tn), ptr(slope_out if next_layer is not None else None), ptr(Zn), ptr(pn), ctypes.byref(rows),
         i32(B), i32(T), i32(V), _stream())
    pending = (Zn, pn, int(rows.value)) if next_layer is not None else None
    return U, (U_prev, in_slope, Wt, Wr, A, Tm, YR, Zy, stat), pending


def commute_bwd(saved, dU: Tensor, into: dict, below=None):
    """-> (gradient of U_prev [B, 32, T, V] (PReLU mask applied), chain).  below = (x_below [B, 2, T, V], Z_below) of a 2-channel layer
    in front (the first layer, fed by the network input): its batch reductions are formed by the same kernel and chain = (buffer, rows) is
    what engine.chain_backward takes as `stats_in`; else chain is None.  `into`: destinations (all overwritten) 'A' [T, V, V], 'T' [V, T, T],
    'Wt' / 'Wr' [16, 32(, 1, 1)], 'gt' 'bet' 'gr' 'ber' [16], 'in_slope' [1] (the producer's PReLU weight gradient; with in_slope)."""
    U_prev, in_slope, Wt, Wr, A, Tm, YR, Zy, stat = saved
    B, Ci, T, V = U_prev.shape
    _chk(dU, "dU", (B, 16, T, V))
    for n, shp in (("A", (T, V, V)), ("T", (V, T, T)), ("gt", (16,)), ("bet", (16,)), ("gr", (16,)), ("ber", (16,))):
        _chk(into[n], "into." + n, shp)
    for n in ("Wt", "Wr"):
        _cuda_f32(into[n], "into." + n)
        if into[n].numel() != 512 or not into[n].is_contiguous():
            raise ValueError(f"into.{n}: expected a contiguous [16, 32] gradient view")
    dsl = into.get("in_slope") if in_slope is not None else None
    if in_slope is not None:
        _chk(dsl, "into.in_slope", (1,))
    d_in = torch.empty_like(U_prev)
    ws = _commute_scratch(B, T, V, U_prev.device)
    bx = bz = buf = None
    rows = 0
    if below is not None:
        bx, bz = below
        _chk(bx, "below.x", (B, 2, T, V)); _chk(bz, "below.Z", (B, 2, T, V))
        lib = _lib.lib()
        lib.coskad_commute_below_floats.restype = ctypes.c_size_t
        rows = int(lib.coskad_commute_below_rows(i32(B)))
        buf = torch.empty(int(lib.coskad_commute_below_floats(i32(B))), device=U_prev.device, dtype=torch.float32)
    call("coskad_commute_bwd_f32", ptr(U_prev), ptr(in_slope), ptr(Wt), ptr(Wr), ptr(A), ptr(Tm), ptr(YR), ptr(Zy), ptr(stat), ptr(dU), ptr(d_in),
         ptr(into["A"]), ptr(into["T"]), ptr(into["Wt"]), ptr(into["Wr"]), ptr(into["gt"]), ptr(into["bet"]), ptr(into["gr"]),
         ptr(into["ber"]), ptr(dsl), ptr(ws), ctypes.c_size_t(ws.numel()), ptr(bx), ptr(bz), ptr(buf),
         ctypes.c_size_t(buf.numel() if buf is not None else 0), i32(B), i32(T), i32(V), _stream())
    return d_in, ((buf, rows) if below is not None else None)


def _rows(t: Tensor, name: str, cols: int):
    """a [B, cols] fp32 device view with unit column stride (a column block of a wider row-major tensor is fine) -> its row stride"""
    _cuda_f32(t, name)
    if t.dim() != 2 or t.shape[1] != cols or (cols > 1 and t.stride(1) != 1):
        raise ValueError(f"{name}: expected a [B, {cols}] view with contiguous columns, got shape {tuple(t.shape)} strides {t.stride()}")
    return t.stride(0)


def ps_head_forward(mean_raw: Tensor, var_raw: Tensor, generator=None):
    """The spherical VAE's latent head on csrc/vae_head.hip (reference models/sts/vae.py:79-91,104-118; PowerSpherical restated in
    coskad_amd/models/sts/vae.py): mean_raw [B, L], var_raw [B, 1] (views of one tensor are fine) -> (z [B, L] sampled latent,
    kl [B], inv_kappa [B], saved) with saved = what ps_head_backward needs.  The Beta draw and the Gaussian direction are torch's
    (torch._sample_dirichlet, torch.randn): the noise streams are the module path's."""
    B, L = mean_raw.shape
    ldm, ldv = _rows(mean_raw, "mean_raw", L), _rows(var_raw, "var_raw", 1)
    dev = mean_raw.device
    mu = torch.empty(B, L, device=dev, dtype=torch.float32)
    kappa = torch.empty(B, device=dev, dtype=torch.float32)
    conc = torch.empty(B, 2, device=dev, dtype=torch.float32)
    total = torch.empty(B, device=dev, dtype=torch.float32)
    call("coskad_ps_head_prep_f32", ptr(mean_raw), i32(ldm), ptr(var_raw), i32(ldv), ptr(mu), ptr(kappa), ptr(conc), ptr(total),
         i32(B), i32(L), _stream())
    x = torch._sample_dirichlet(conc, generator) if generator is not None else torch._sample_dirichlet(conc)
    eps = torch.randn(B, L - 1, device=dev, dtype=torch.float32, generator=generator)
    z = torch.empty(B, L, device=dev, dtype=torch.float32)
    kl = torch.empty(B, device=dev, dtype=torch.float32)
    ik = torch.empty(B, device=dev, dtype=torch.float32)
    call("coskad_ps_head_sample_f32", ptr(x), ptr(eps), ptr(mu), ptr(kappa), ptr(z), ptr(kl), ptr(ik), i32(B), i32(L), _stream())
    return z, kl, ik, (mean_raw, var_raw, mu, kappa, conc, total, x, eps)


def ps_head_backward(saved, dz: Tensor, w_kl: float, w_exp: float, d_mean_raw: Optional[Tensor] = None,
                     d_var_raw: Optional[Tensor] = None):
    """-> (d_mean_raw [B, L], d_var_raw [B, 1]) for loss = <dz, z> + w_kl * sum kl + w_exp * sum inv_kappa; the destinations may be
    column views of one [B, L + 1] tensor."""
    mean_raw, var_raw, mu, kappa, conc, total, x, eps = saved
    B, L = mean_raw.shape
    _chk(dz, "dz", (B, L))
    g = torch._dirichlet_grad(x, conc, total.unsqueeze(-1).expand(B, 2).contiguous())
    if d_mean_raw is None:
        d_mean_raw = torch.empty(B, L, device=dz.device, dtype=torch.float32)
    if d_var_raw is None:
        d_var_raw = torch.empty(B, 1, device=dz.device, dtype=torch.float32)
    call("coskad_ps_head_bwd_f32", ptr(dz), ptr(x), ptr(g), ptr(eps), ptr(mu), ptr(kappa), ptr(mean_raw), i32(_rows(mean_raw, "mean_raw", L)),
         ptr(var_raw), i32(_rows(var_raw, "var_raw", 1)), ctypes.c_float(w_kl), ctypes.c_float(w_exp), ptr(d_mean_raw),
         i32(_rows(d_mean_raw, "d_mean_raw", L)), ptr(d_var_raw), i32(_rows(d_var_raw, "d_var_raw", 1)), i32(B), i32(L), _stream())
    return d_mean_raw, d_var_raw


def center_finalize(acc, eps: float, L: int):
    _chk(acc, "acc", (HEAD_SLOTS,))
    c = torch.empty(L, device=acc.device, dtype=torch.float32)
    call("coskad_center_finalize_f32", ptr(acc), ptr(c), ctypes.c_float(eps), i32(L), _stream())
    return c


def midpoint_finalize(acc, L: int):
    _chk(acc, "acc", (HEAD_SLOTS,))
    c = torch.empty(L, device=acc.device, dtype=torch.float32)
    call("coskad_midpoint_finalize_f32", ptr(acc), ptr(c), i32(L), _stream())
    return c


def sqnorm(p, mask, scale: float, ws=None):
    _chk(p, "p"); _chk(mask, "mask", p.shape, optional=True)
    ws = torch.empty(256, device=p.device, dtype=torch.float32) if ws is None else ws
    out = torch.empty(1, device=p.device, dtype=torch.float32)
    call("coskad_sqnorm_f32", ptr(p), ptr(mask), ctypes.c_size_t(p.numel()), ctypes.c_float(scale), ptr(out), ptr(ws), _stream())
    return out


def adam(p, g, m, v, mask, lr, beta1, beta2, eps, step, gscale=1.0, reg_coef=0.0):
    for n, t in (("p", p), ("g", g), ("m", m), ("v", v)):
        _chk(t, n, p.shape)
    _chk(mask, "mask", p.shape, optional=True)
    call("coskad_adam_f32", ptr(p), ptr(g), ptr(m), ptr(v), ptr(mask), ctypes.c_size_t(p.numel()), ctypes.c_float(lr),
         ctypes.c_float(beta1), ctypes.c_float(beta2), ctypes.c_float(eps), i32(step), ctypes.c_float(gscale),
         ctypes.c_float(reg_coef), _stream())


def adam_pow(p, g, m, v, mask, lr, beta1, beta2, eps, b1pow, b2pow, gscale=1.0, reg_coef=0.0):
    """Adam with the caller's fp32 running products beta^t (the arithmetic of adam_dev's device-side tick)."""
    for n, t in (("p", p), ("g", g), ("m", m), ("v", v)):
        _chk(t, n, p.shape)
    _chk(mask, "mask", p.shape, optional=True)
    call("coskad_adam_pow_f32", ptr(p), ptr(g), ptr(m), ptr(v), ptr(mask), ctypes.c_size_t(p.numel()), ctypes.c_float(lr),
         ctypes.c_float(beta1), ctypes.c_float(beta2), ctypes.c_float(eps), ctypes.c_float(b1pow), ctypes.c_float(b2pow),
         ctypes.c_float(gscale), ctypes.c_float(reg_coef), _stream())


def prelu_fwd(u: Tensor, slope: Tensor) -> Tensor:
    _chk(u, "u"); _chk(slope, "slope", (1,))
    out = torch.empty_like(u)
    call("coskad_prelu_fwd_f32", ptr(u), ptr(slope), ptr(out), ctypes.c_size_t(u.numel()), _stream())
    return out


def prelu_bwd(u: Tensor, dout: Tensor, slope: Tensor, dslope: Optional[Tensor] = None, accumulate=False) -> Tensor:
    _chk(u, "u"); _chk(dout, "dout", u.shape); _chk(slope, "slope", (1,)); _chk(dslope, "dslope", (1,), optional=True)
    du = torch.empty_like(u)
    ws = torch.empty(1024, device=u.device, dtype=torch.float32)
    call("coskad_prelu_bwd_f32", ptr(u), ptr(dout), ptr(slope), ptr(du), ptr(dslope), ptr(ws),
         i32(1 if accumulate else 0), ctypes.c_size_t(u.numel()), _stream())
    return du


def rec_head(U: Tensor, x: Tensor, slope: Tensor, need_grad: bool = True, need_xrec: bool = False, dslope: Optional[Tensor] = None,
             upstream: float = 1.0, accumulate: bool = False):
    """-> (loss [1] = F.mse_loss(PReLU(U), x), dU or None, x_rec or None); dslope (+)= the slope gradient."""
    _chk(U, "U"); _chk(x, "x", U.shape); _chk(slope, "slope", (1,)); _chk(dslope, "dslope", (1,), optional=True)
    dU = torch.empty_like(U) if need_grad else None
    xrec = torch.empty_like(U) if need_xrec else None
    loss = torch.empty(1, device=U.device, dtype=torch.float32)
    ws = torch.empty(2048, device=U.device, dtype=torch.float32)
    call("coskad_rec_head_f32", ptr(U), ptr(x), ptr(slope), ptr(xrec), ptr(dU), ptr(loss), ptr(dslope if need_grad else None),
         ctypes.c_float(upstream), ptr(ws), i32(1 if accumulate else 0), ctypes.c_size_t(U.numel()), _stream())
    return loss, dU, xrec


def rev_btlnk_ok(N: int, L: int) -> bool:
    return L in (8, 16) and N % 4 == 0


def rev_btlnk_fwd(z: Tensor, W: Tensor, bias: Optional[Tensor]) -> Tensor:
    """H = z W^T + bias (rev_btlnk, ae.py:223-227) as one streaming pass (csrc/rev_btlnk.hip); other latent sizes: the strided GEMM."""
    B, L = z.shape
    N = W.shape[0]
    _chk(z, "z"); _chk(W, "W", (N, L)); _chk(bias, "bias", (N,), optional=True)
    if not rev_btlnk_ok(N, L):
        return gemm(z, W.t(), bias=bias, bias_mode=2 if bias is not None else 0)
    H = torch.empty(B, N, device=z.device, dtype=torch.float32)
    call("coskad_rev_btlnk_fwd_f32", ptr(z), ptr(W), ptr(bias), ptr(H), i32(B), i32(N), i32(L), _stream())
    return H


def rev_btlnk_bwd(dH: Tensor, z: Tensor, W: Tensor, dW: Tensor, db: Optional[Tensor], dz: Optional[Tensor] = None,
                  accumulate: bool = False) -> Tensor:
    """-> dz [B, L] (added to the given `dz` when one is passed); fills dW [N, L] and db [N] ((+)= with `accumulate`)."""
    B, N = dH.shape
    L = z.shape[1]
    _chk(dH, "dH"); _chk(z, "z", (B, L)); _chk(W, "W", (N, L)); _chk(dW, "dW", (N, L)); _chk(db, "db", (N,), optional=True)
    _chk(dz, "dz", (B, L), optional=True)
    if not rev_btlnk_ok(N, L) or dH.data_ptr() % 16:
        zs1 = torch.cat([z, torch.ones(B, 1, device=z.device)], 1)
        gw = torch.empty(N, L + 1, device=dH.device, dtype=torch.float32)
        gemm_rows_outer(dH, zs1, gw)
        if accumulate:
            dW.add_(gw[:, :-1])
            if db is not None:
                db.add_(gw[:, -1])
        else:
            dW.copy_(gw[:, :-1])
            if db is not None:
                db.copy_(gw[:, -1])
        if dz is None:
            return gemm(dH, W)
        return gemm(dH, W, out=dz, accumulate=True)
    fn = _lib.lib().coskad_rev_btlnk_ws_floats
    fn.restype = ctypes.c_size_t
    ws = torch.empty(fn(i32(B), i32(N), i32(L)), device=dH.device, dtype=torch.float32)
    acc_dz = dz is not None
    if dz is None:
        dz = torch.empty(B, L, device=dH.device, dtype=torch.float32)
    call("coskad_rev_btlnk_bwd_f32", ptr(dH), ptr(z), ptr(W), ptr(dz), i32(1 if acc_dz else 0), ptr(dW), ptr(db),
         i32(1 if accumulate else 0), ptr(ws), i32(B), i32(N), i32(L), _stream())
    return dz


def gcn_bwd_params(x: Tensor, dZ: Tensor, A: Tensor, Tm: Tensor):
    """(dA, dT) of ConvTemporalGraphical given its input and output gradient."""
    N, C, T, V = x.shape
    _chk(x, "x"); _chk(dZ, "dZ", x.shape); _chk(A, "A", (T, V, V)); _chk(Tm, "T", (V, T, T))
    fn = _lib.lib().coskad_gcn_bwd_params_ws_bytes
    fn.restype = ctypes.c_size_t
    nbytes = fn(i32(T), i32(V))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    dA, dT = torch.empty_like(A), torch.empty_like(Tm)
    call("coskad_gcn_bwd_params_f32", ptr(x), ptr(dZ), ptr(A), ptr(Tm), ptr(dA), ptr(dT), ptr(ws),
         ctypes.c_size_t(nbytes), i32(0), i32(N * C), i32(T), i32(V), _stream())
    return dA, dT


def gcn_bwd_params_dx(x: Tensor, dZ: Tensor, A: Tensor, Tm: Tensor, add: Optional[Tensor] = None, dA: Optional[Tensor] = None,
                      dT: Optional[Tensor] = None):
    """(dA, dT, dX) of ConvTemporalGraphical in one pass over dZ: dX = gcn^T(dZ) (+ add, e.g. an identity residual's gradient);
    dA / dT: destinations (e.g. views of a flat gradient buffer) instead of fresh tensors."""
    N, C, T, V = x.shape
    _chk(x, "x"); _chk(dZ, "dZ", x.shape); _chk(A, "A", (T, V, V)); _chk(Tm, "T", (V, T, T)); _chk(add, "add", x.shape, optional=True)
    fn = _lib.lib().coskad_gcn_bwd_params_ws_bytes
    fn.restype = ctypes.c_size_t
    nbytes = fn(i32(T), i32(V))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    dA = torch.empty_like(A) if dA is None else dA
    dT = torch.empty_like(Tm) if dT is None else dT
    _chk(dA, "dA", (T, V, V)); _chk(dT, "dT", (V, T, T))
    dX = torch.empty_like(x)
    call("coskad_gcn_bwd_params_dx_f32", ptr(x), ptr(dZ), ptr(A), ptr(Tm), ptr(dA), ptr(dT), ptr(dX), ptr(add), ptr(ws),
         ctypes.c_size_t(nbytes), i32(0), i32(N * C), i32(T), i32(V), _stream())
    return dA, dT, dX


def adam_dev(p, g, m, v, mask, hyper, beta1, beta2, eps, gscale=1.0, reg_coef=0.0):
    """Adam with {lr, beta1^t, beta2^t} in the device tensor `hyper` (replayable in a hipGraph)."""
    for n, t in (("p", p), ("g", g), ("m", m), ("v", v)):
        _chk(t, n, p.shape)
    _chk(mask, "mask", p.shape, optional=True); _chk(hyper, "hyper", (4,))
    call("coskad_adam_dev_f32", ptr(p), ptr(g), ptr(m), ptr(v), ptr(mask), ctypes.c_size_t(p.numel()), ptr(hyper),
         ctypes.c_float(beta1), ctypes.c_float(beta2), ctypes.c_float(eps), ctypes.c_float(gscale),
         ctypes.c_float(reg_coef), _stream())


def gather_transform(xy, mats, index, T, V):
    """Batch [B, 2, T, V] from the device-resident window table xy [N, 2, T*V]: item i = transform i // N of window
    i % N (utils/dataset.py:65-77; PoseTransform of utils/dataset_utils.py:272-310)."""
    N = xy.shape[0]
    _chk(xy, "xy", (N, 2, T * V)); _chk(mats, "mats", (mats.shape[0], 3, 3)); _chk(index, "index", dtype=torch.int64)
    B = index.numel()
    out = torch.empty(B, 2, T, V, device=xy.device, dtype=torch.float32)
    call("coskad_gather_transform_f32", ptr(xy), ptr(index), ptr(mats), ptr(out), i32(B), i32(N), i32(mats.shape[0]),
         i32(T * V), _stream())
    return out
