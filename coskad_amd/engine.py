"""Host-side orchestration of the HIP kernels for a stack of ST_GCNN layers.

A *chain* is the ``nn.Sequential`` of ST_GCNN layers of the reference's Encoder / Decoder
(models/common/components.py:70-105,143-179).  Layers exchange PRE-activations: layer i
writes U_i, layer i+1 applies PReLU_i while staging U_i (see csrc/stsgcn_fwd.hip).

  chain_forward  : eval  -> bn_fold (running stats) + layer_apply per layer
                   train -> layer_train_stats (batch stats, running-stat update) + layer_apply
  chain_backward : layer_bwd per layer, last to first

Everything here is plumbing around C-ABI calls (coskad_amd.ops); there is no CPU path.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from . import ops

Tensor = torch.Tensor


@dataclass
class LayerTensors:
    """Views of one ST_GCNN_layer's parameters / buffers (reference stsgcn.py:47-91)."""
    A: Tensor
    T: Tensor
    Wt: Tensor              # tcn.0.weight  [Co,Ci,1,1]
    bt: Optional[Tensor]    # tcn.0.bias
    gt: Tensor              # tcn.1.weight
    bet: Tensor             # tcn.1.bias
    rm_t: Tensor
    rv_t: Tensor
    nbt_t: Tensor
    Wr: Optional[Tensor]    # residual.0.weight (None: identity residual)
    br: Optional[Tensor]
    gr: Optional[Tensor]
    ber: Optional[Tensor]
    rm_r: Optional[Tensor]
    rv_r: Optional[Tensor]
    nbt_r: Optional[Tensor]
    slope: Tensor           # prelu.weight [1]
    momentum: float = 0.1
    bn: object = None       # the tcn BatchNorm module when its momentum is None (cumulative moving average: the factor changes per step)
    cache: Optional[dict] = None   # owned by the layer module: eval-mode folded weights, keyed by tensor versions

    def fold_key(self):
        """Identity + in-place version of everything the eval-mode fold reads (torch optimizers and load_state_dict bump
        the versions; a training forward clears the cache because this library's kernels write through raw pointers)."""
        ts = (self.Wt, self.bt, self.gt, self.bet, self.rm_t, self.rv_t, self.Wr, self.br, self.gr, self.ber, self.rm_r, self.rv_r)
        return tuple((t.data_ptr(), t._version) if t is not None else None for t in ts)

    def step_momentum(self) -> float:
        """exponential_average_factor of this training forward (both BatchNorms of a layer count their batches in lockstep)"""
        return ops.bn_momentum(self.bn) if self.bn is not None else self.momentum

    @property
    def Co(self) -> int:
        return self.Wt.shape[0]

    @property
    def Ci(self) -> int:
        return self.Wt.shape[1]

    def w2(self, w: Optional[Tensor]) -> Optional[Tensor]:
        return None if w is None else w.view(w.shape[0], w.shape[1])

    def param_list(self) -> List[Tensor]:
        ps = [self.A, self.T, self.Wt]
        if self.bt is not None:
            ps.append(self.bt)
        ps += [self.gt, self.bet]
        if self.Wr is not None:
            ps.append(self.Wr)
            if self.br is not None:
                ps.append(self.br)
            ps += [self.gr, self.ber]
        ps.append(self.slope)
        return ps

    def grad_names(self) -> List[str]:
        ns = ["A", "T", "Wt"]
        if self.bt is not None:
            ns.append("bt")
        ns += ["gt", "bet"]
        if self.Wr is not None:
            ns.append("Wr")
            if self.br is not None:
                ns.append("br")
            ns += ["gr", "ber"]
        ns.append("slope")
        return ns


class Workspace:
    """Grow-only scratch shared by all kernels of one module (byte tensor on the device)."""

    def __init__(self) -> None:
        self.buf: Optional[Tensor] = None

    def get(self, nbytes: int, device) -> Tensor:
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self.buf


@dataclass
class ChainCtx:
    """What chain_backward needs from chain_forward (train mode)."""
    inputs: List[Tensor] = field(default_factory=list)   # input of layer i (x, U_0, U_1, ...)
    stats: List[Tensor] = field(default_factory=list)    # stat block of layer i
    zs: List[Optional[Tensor]] = field(default_factory=list)   # stored gcn(input) of layer i (None: recompute)
    in_slope: Optional[Tensor] = None                    # activation applied to inputs[0] (None: raw)
    sync: object = None                                  # SyncBN: the process group of the forward ...
    sync_count: float = 0.0                              # ... and the positions of the global batch


# eval-mode forwards run the first two layers in one kernel where it is built (csrc/eval_layer_bpc.hip, FIRST form); tests flip it
EVAL_FIRST_PAIR = True


def chain_forward(x: Tensor, layers: List[LayerTensors], training: bool, ws: Workspace,
                  in_slope: Optional[Tensor] = None, want_ctx: bool = False, sync=None, pending0=None):
    """-> (U_last, ctx).  U_last is the last layer's PRE-activation; apply layers[-1].slope to it.
    sync: a torch.distributed process group -> SyncBN (optional; the reference trains with per-rank statistics,
    train_COSKAD.py:75-78): every BatchNorm boundary adds the other ranks' fp64 moment sums before the fold.
    pending0: (Z, partials, rows) of layers[0] when the producer of x ran its statistics pass (ops.commute_fwd)."""
    B, C, T, V = x.shape
    ctx = ChainCtx(in_slope=in_slope) if want_ctx else None
    sync_count = None
    # BatchNorm with track_running_stats=False normalises with batch statistics in eval mode too (nothing to update: NULL buffers)
    batch_stats = [training or L.rm_t is None for L in layers]
    if sync is not None and training:
        import torch.distributed as dist
        if not STORE_Z:
            raise ValueError("SyncBN runs on the stored-Z training path")
        cnt = torch.tensor([float(B * T * V)], device=x.device, dtype=torch.float64)
        dist.all_reduce(cnt, group=sync)
        sync_count = float(cnt.item())               # positions of the global batch (ranks may hold different batch sizes)
        if ctx is not None:
            ctx.sync, ctx.sync_count = sync, sync_count
    h, slope = x, in_slope
    n = len(layers)
    # fuse[i]: layer i's apply kernel also forms layer i+1's Z and moment partials (csrc/fused_apply_next.hip), so
    # layer i+1 needs no statistics pass of its own
    fuse = [False] * n
    ftab = None
    if all(batch_stats) and STORE_Z and FUSE_NEXT:
        for i in range(n - 1):
            fuse[i] = layers[i + 1].Ci == layers[i].Co and ops.layer_apply_next_ok(layers[i].Ci, layers[i].Co, T, V)
        nxt = [i + 1 for i in range(n - 1) if fuse[i]]
        if nxt:
            ftab = torch.empty(n, ops.ftab_floats(), device=x.device, dtype=torch.float32)
            for k in range(0, len(nxt), 4):
                grp = nxt[k:k + 4]
                ops.build_ftabs([layers[i].A for i in grp], [layers[i].T for i in grp], [ftab[i] for i in grp])
    # the same fusion on the 25-joint layout (csrc/fused_apply_flat.hip, NX form): no operand tables, the next layer's A / T directly
    fuse_flat = [False] * n
    if all(batch_stats) and STORE_Z and FUSE_NEXT and sync is None:
        for i in range(n - 1):
            fuse_flat[i] = (not fuse[i] and layers[i + 1].Ci == layers[i].Co and layers[i].Wr is not None
                            and ops.layer_apply_next_flat_ok(layers[i].Ci, layers[i].Co, T, V))
    def eval_fold(L):
        key = L.fold_key() if L.cache is not None else None
        if key is not None and L.cache.get("key") == key:
            return L.cache["fold"]                         # weights unchanged since the last eval forward
        wb = ops.bn_fold(L.w2(L.Wt), L.bt, L.gt, L.bet, L.rm_t, L.rv_t, L.w2(L.Wr), L.br, L.gr, L.ber, L.rm_r, L.rv_r)
        if key is not None:
            L.cache["key"], L.cache["fold"] = key, wb
        return wb

    skip = -1
    pending = pending0          # (Z, partials, rows) of THIS layer, written by the previous layer's apply (or by x's producer)
    if pending0 is not None and not (batch_stats[0] and STORE_Z and sync is None):
        raise ValueError("chain_forward: pending0 is for the stored-Z training path without SyncBN")
    for i, L in enumerate(layers):
        if i == skip:
            continue
        if h.shape[1] != L.Ci:
            raise ValueError(f"layer expects {L.Ci} input channels, got {h.shape[1]}")
        Z = None
        if batch_stats[i]:
            if L.cache:
                L.cache.clear()   # running stats (and, after the optimiser, the weights) change through raw-pointer kernels
                                  # that do not bump torch's version counters: drop the eval-mode fold
            buf = ws.get(ops.train_stats_ws_bytes(L.Ci), x.device)
            if sync_count is not None:
                if pending is not None:
                    Z, partials, rows = pending
                    sums = ops.layer_moment_sums(partials, rows, L.Ci)
                else:
                    Z = torch.empty_like(h)
                    sums = ops.layer_train_moments(h, L.A, L.T, slope, buf, Z=Z)
                dist.all_reduce(sums, group=sync)
                wfold, bias, stat = ops.layer_train_fold_sums(
                    sums, sync_count, L.w2(L.Wt), L.bt, L.gt, L.bet, L.rm_t, L.rv_t, L.nbt_t,
                    L.w2(L.Wr), L.br, L.gr, L.ber, L.rm_r, L.rv_r, L.nbt_r, momentum=L.step_momentum())
            elif pending is not None:
                Z, partials, rows = pending
                wfold, bias, stat = ops.layer_train_fold(
                    partials, rows, B, T, V, L.w2(L.Wt), L.bt, L.gt, L.bet, L.rm_t, L.rv_t, L.nbt_t,
                    L.w2(L.Wr), L.br, L.gr, L.ber, L.rm_r, L.rv_r, L.nbt_r, buf, momentum=L.step_momentum())
            else:
                if STORE_Z:
                    Z = torch.empty_like(h)     # gcn(PReLU(h)): written by the statistics pass, read by everything after
                wfold, bias, stat = ops.layer_train_stats(
                    h, L.A, L.T, slope, L.w2(L.Wt), L.bt, L.gt, L.bet, L.rm_t, L.rv_t, L.nbt_t,
                    L.w2(L.Wr), L.br, L.gr, L.ber, L.rm_r, L.rv_r, L.nbt_r, buf, momentum=L.step_momentum(), Z=Z)
        else:
            wfold, bias = eval_fold(L)
            stat = None
            # the first layer of a stack fed by the network input runs inside the second layer's kernel where nobody asks for its output
            if (EVAL_FIRST_PAIR and i == 0 and ctx is None and slope is None and n > 1 and not batch_stats[1] and layers[1].Ci == L.Co
                    and L.Wr is not None and wfold.shape[1] == L.Co
                    and ops.layer_first_pair_ok(L.Ci, L.Co, layers[1].Co, T, V)):
                N = layers[1]
                wf2, b2 = eval_fold(N)
                if wf2.shape[1] == N.Co:
                    h = ops.layer_first_pair_apply(h, L.A, L.T, wfold, bias, N.A, N.T, wf2, b2, L.Co, N.Co, L.slope)
                    slope, skip = N.slope, 1
                    continue
        pending = None
        if fuse[i]:
            rows_max = ops.layer_apply_next_rows(B, h.shape[1], L.Co)
            partials = torch.empty(rows_max * 2 * (L.Co * L.Co + L.Co), device=x.device, dtype=torch.float32)
            u, Zn, rows = ops.layer_apply_next(Z, h, wfold, bias, L.Co, slope, L.slope, ftab[i + 1], partials, T, V)
            pending = (Zn, partials, rows)
        elif fuse_flat[i] and Z is not None:
            u, Zn, partials, rows = ops.layer_apply_next_flat(Z, h, wfold, bias, L.Co, slope, L.slope, layers[i + 1].A, layers[i + 1].T)
            pending = (Zn, partials, rows)
        elif Z is not None:
            u = ops.layer_apply_z(Z, h, L.A, L.T, wfold, bias, L.Co, in_slope=slope)
        else:
            u = ops.layer_apply(h, L.A, L.T, wfold, bias, L.Co, in_slope=slope)
        if ctx is not None:
            ctx.inputs.append(h)
            ctx.stats.append(stat)
            ctx.zs.append(Z)
        h, slope = u, L.slope
    return h, ctx


# training forward keeps Z = gcn(X) of every layer (82 channels x 816 B per clip) instead of recomputing the mixing in the
# apply and backward kernels.  A module constant, not an environment switch: tests flip it to cover the recompute path.
STORE_Z = True
# backward: the data kernel of layer 2 also forms the batch reductions of layer 1 (csrc/fused_bwd.hip, NS = 1) where the shapes allow
FUSE_BELOW = True
# (A/B hook, tools/ab_chain.py) layer indices whose data kernel does NOT form the reductions of the layer below although it could
CHAIN_SKIP = frozenset()
# backward: the bottleneck's backward also forms the batch reductions of the top layer (csrc/btlnk_chain.hip) where the shapes allow
FUSE_TOP = True
# with the stored-Z path: layer i's apply kernel also produces layer i+1's Z and BatchNorm moment partials where
# csrc/fused_apply_next.hip takes the shape (tests flip it to cover the separate statistics pass)
FUSE_NEXT = True


class SideStream:
    """Second HIP stream for the mixing-parameter gradients (dA, dT): they depend only on a layer's input and dZ, so
    they run beside the next layer's reductions and fill the GPU during its single-block fold kernels.  Owns the
    buffers that must outlive the main stream's reuse: two dZ images (alternating layers) and the partial sums."""

    def __init__(self) -> None:
        self.stream: Optional[torch.cuda.Stream] = None
        self.dz = [None, None]
        self.ws = None
        self.done = [None, None]      # events: side work that last read dz[k] has finished

    def ensure(self, nfloats: int, ws_bytes: int, device) -> None:
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=device)
        for k in range(2):
            if self.dz[k] is None or self.dz[k].numel() < nfloats:
                self.dz[k] = torch.empty(nfloats, device=device, dtype=torch.float32)
        if self.ws is None or self.ws.numel() < ws_bytes:
            self.ws = torch.empty(ws_bytes, device=device, dtype=torch.uint8)


def btlnk_backward(ctx: Optional[ChainCtx], layers: List[LayerTensors], U: Tensor, W: Tensor, dz: Tensor, slope: Optional[Tensor],
                   dW: Tensor, db: Optional[Tensor], dslope: Optional[Tensor], ws: Workspace):
    """Backward of the bottleneck Linear on the chain's output U (ae.py:97-101) -> (dU, top_stats).  Where the shapes allow
    (csrc/btlnk_chain.hip: a 64-channel top layer over 16 / 32 channels, stored Z) the same pass forms the TOP layer's batch
    reductions; `top_stats` is then chain_backward's `stats_in`, else None."""
    B = U.shape[0]
    K, L = U.numel() // B, W.shape[0]
    i = len(layers) - 1 if layers else -1
    Z = ctx.zs[i] if (ctx is not None and ctx.zs and i >= 0) else None
    if (FUSE_TOP and Z is not None and U.dim() == 4 and layers[i].Wr is not None and L <= 16
            and ops.btlnk_bwd_chain_ok(K, U.shape[2] * U.shape[3], layers[i].Ci)):
        in_slope = layers[i - 1].slope if i > 0 else ctx.in_slope
        return ops.btlnk_bwd_chain(U, W, dz, slope, dW, db, dslope, ws, ctx.inputs[i], Z, in_slope)
    buf = ws.get(ops.btlnk_bwd_ws_bytes(B, K, L), U.device)
    return ops.btlnk_bwd(U, W, dz, slope, dW, db, dslope, buf), None


def chain_backward(ctx: ChainCtx, layers: List[LayerTensors], dU: Tensor, ws: Workspace,
                   grads: List[Dict[str, Tensor]], need_dx: bool, accumulate: bool = False,
                   side: Optional[SideStream] = None, stats_in=None, in_slope_grad: Optional[Tensor] = None) -> Optional[Tensor]:
    """Backward through the chain.  `grads[i]` maps A,T,Wt,bt,gt,bet,Wr,br,gr,ber,slope -> tensors to
    fill for layer i.  The slope gradient of layer i is produced while back-propagating through
    layer i+1 (its consumer); the caller owns the last layer's slope gradient.
    With `side`, dA / dT are computed on its stream (joined into the current stream before returning).
    `stats_in`: (chain buffer, rows) of the TOP layer's batch reductions when the producer of dU formed them (btlnk_backward).
    `in_slope_grad`: where the gradient of ctx.in_slope goes (the PReLU weight of a layer in front of this chain that handed over
    its pre-activation: coskad_amd/lowrank.py); None: nobody owns it.
    Returns d(inputs[0]) if need_dx."""
    n = len(layers)
    if side is not None and stats_in is not None:
        raise ValueError("chain_backward: the side-stream path runs its own batch reductions (stats_in must be None)")
    if ctx.sync is not None and side is not None:
        raise ValueError("SyncBN runs on the main stream (side=None)")
    main = torch.cuda.current_stream() if side is not None else None
    if side is not None:
        side.done = [None, None]     # the previous call joined the side stream: nothing of it is still in flight
    # stats_in: (partial rows, rows) of layer i's batch reductions, formed by layer i + 1's data kernel (or by the bottleneck's)
    for i in range(n - 1, -1, -1):
        L = layers[i]
        x_in = ctx.inputs[i]
        B, Ci, T, V = x_in.shape
        in_slope = layers[i - 1].slope if i > 0 else ctx.in_slope
        g = dict(grads[i])
        g.pop("slope", None)
        if i > 0:
            g["slope_in"] = grads[i - 1]["slope"]
        elif in_slope_grad is not None and in_slope is not None:
            g["slope_in"] = in_slope_grad
        want_dx = need_dx or i > 0
        buf = ws.get(ops.layer_bwd_ws_bytes(B, Ci, L.Co, T, V), x_in.device)
        args = (x_in, dU, L.A, L.T, in_slope, ctx.stats[i], L.w2(L.Wt), L.gt, L.w2(L.Wr), L.gr, _as2d(g), buf)
        if side is None:
            sync = ctx.sync
            if sync is not None:
                import torch.distributed as dist
                if stats_in is None:                      # the top layer (or a pair without a chain kernel): stage 1 by itself
                    stats_in = ops.layer_bwd_stats(x_in, dU, L.A, L.T, in_slope, L.Wr is not None, buf, Z=ctx.zs[i])
                dist.all_reduce(ops.chain_sums(stats_in[0], stats_in[1], Ci, L.Co), group=sync)
            below = None
            Zi = ctx.zs[i] if ctx.zs else None
            if (FUSE_BELOW and i not in CHAIN_SKIP and Zi is not None and i > 0 and ctx.zs[i - 1] is not None
                    and in_slope is not None and layers[i - 1].Wr is not None):
                cb = ctx.inputs[i - 1].shape[1]
                rows = ops.layer_bwd_below_rows(B, Ci, L.Co, cb, T, V)
                if rows:
                    below = (ctx.inputs[i - 1], ctx.zs[i - 1], layers[i - 2].slope if i > 1 else ctx.in_slope,
                             torch.empty(ops.layer_bwd_below_floats(B, Ci, L.Co, cb, T, V), device=x_in.device, dtype=torch.float32))
            dIn = ops.layer_bwd(*args, need_dx=want_dx, accumulate=accumulate, Z=Zi,
                                stats_in=stats_in, below=below, stats_count=ctx.sync_count if sync is not None else 0.0)
            stats_in = (below[3], rows) if below is not None else None
        else:
            k = i & 1
            side.ensure(max(x.numel() for x in ctx.inputs), ops.layer_gcn_params_ws_bytes(T, V), x_in.device)
            if side.done[k] is not None:
                main.wait_event(side.done[k])          # the side kernels that read this dZ image are finished
            dz = side.dz[k][:x_in.numel()].view(B, Ci, T, V)
            dIn = ops.layer_bwd_data(*args, dZ=dz, need_dx=want_dx, accumulate=accumulate, Z=ctx.zs[i] if ctx.zs else None)
            side.stream.wait_stream(main)
            with torch.cuda.stream(side.stream):
                ops.layer_gcn_params(x_in, in_slope, dz, L.A, L.T, g["A"], g["T"], side.ws, accumulate=accumulate)
                side.done[k] = side.stream.record_event()
        dU = dIn
    if side is not None:
        main.wait_stream(side.stream)
    return dU if need_dx else None


def _as2d(g: Dict[str, Tensor]) -> Dict[str, Tensor]:
    out = dict(g)
    for k in ("Wt", "Wr"):
        if out.get(k) is not None and out[k].dim() == 4:
            out[k] = out[k].view(out[k].shape[0], out[k].shape[1])
    return out


# ---- fused eval-mode encoder (csrc/fused_fwd.hip) -------------------------------------------------------------------
class FusedEncoderPlan:
    """Operand streams of the fused eval-mode encoder for one model: built on the device from the BatchNorm-folded
    weights (coskad_bn_fold_f32), the mixing matrices and the bottleneck weight by ONE gather launch per stream
    (index maps: coskad_amd/fused_plan.py), cached until a parameter changes.

    Validity: torch version counters of every tensor the streams read, plus a token left in each layer's fold cache --
    training forwards clear those caches (their kernels and the fused Adam write through raw pointers)."""

    def __init__(self) -> None:
        self.key = None
        self.token = None
        self.tab = self.wreg = self.wb = self.slopes = None
        self._idx = {}

    def _indices(self, latent: int, device):
        k = (latent, str(device))
        if k not in self._idx:
            from . import fused_plan as FP
            self._idx[k] = tuple(torch.from_numpy(a.reshape(-1)).to(device) for a in
                                 (FP.tab_index(latent), FP.wreg_index(latent), FP.wb_index(latent)))
        return self._idx[k]

    def get(self, layers: List[LayerTensors], W, ver_extra=()):
        """W: the weight [L, hid*T*V] the tile-major output is multiplied with, or a tuple of such weights stacked along
        their rows (the VAE's mean / concentration heads share one pass)."""
        Ws = tuple(W) if isinstance(W, (tuple, list)) else (W,)
        key = tuple(L.fold_key() for L in layers) + tuple((t.data_ptr(), t._version) for L in layers for t in (L.A, L.T, L.slope)) \
            + tuple((w.data_ptr(), w._version) for w in Ws) + tuple(ver_extra)
        if self.key == key and self.token is not None and all(L.cache is not None and L.cache.get("fused") is self.token for L in layers):
            return self
        from . import fused_plan as FP
        latent = sum(w.shape[0] for w in Ws)
        parts = []
        for L in layers:
            wfold, bias = ops.bn_fold(L.w2(L.Wt), L.bt, L.gt, L.bet, L.rm_t, L.rv_t, L.w2(L.Wr), L.br, L.gr, L.ber, L.rm_r, L.rv_r)
            parts += [L.A.detach().reshape(-1), L.T.detach().reshape(-1), wfold.reshape(-1), bias.reshape(-1)]
        parts += [w.detach().reshape(-1) for w in Ws]
        src = torch.cat(parts)
        assert src.numel() == FP.src_layout(latent).total
        ti, wi, bi = self._indices(latent, src.device)
        self.tab, self.wreg = ops.gather(src, ti), ops.gather(src, wi)
        self.wb = ops.gather(src, bi).view(latent, FP.KP)
        self.slopes = torch.cat([L.slope.detach().reshape(1) for L in layers]).contiguous()
        self.key, self.token = key, object()
        for L in layers:
            if L.cache is not None:
                L.cache["fused"] = self.token
        return self


def fused_encoder_supported(layers: List[LayerTensors], n_frames: int, n_joints: int) -> bool:
    from . import fused_plan as FP
    if any(L.Wr is None for L in layers):
        return False
    if any(L.rm_t is None or L.rm_r is None for L in layers):   # no running statistics: batch statistics even in eval mode
        return False
    return FP.supports((layers[0].Ci,) + tuple(L.Co for L in layers), n_frames, n_joints)
