"""Mirror of the reference's models/graph_layers/stsgcn.py on the HIP path.

Same class names, constructor arguments, attribute / state_dict names
(``gcn.A``, ``gcn.T``, ``tcn.0.weight``, ``tcn.1.running_mean``, ``residual.0.weight``,
``prelu.weight`` ...), so a reference checkpoint loads unchanged.  The torch sub-modules
(`nn.Conv2d`, `nn.BatchNorm2d`, `nn.PReLU`) are PARAMETER CONTAINERS only: forward never calls
them; it calls the gfx950 kernels through coskad_amd.engine / coskad_amd.ops.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from ... import engine, ops

Tensor = torch.Tensor
_FITS: dict = {}      # (Ci, Co, T, V) -> the tile kernels take the layer (coskad_layer_fits)
WIDE_CHANNELS = 64   # widest layer the fused tile kernels take (csrc: `channels > 64 not supported`)


def layer_fits(Ci: int, Co: int, T: int, V: int) -> bool:
    """Host arithmetic of `coskad_layer_fits` (csrc/stsgcn_bwd.hip): does one clip of a (Ci -> Co) layer fit the
    LDS-resident tile kernels?  Restated here so that building or inspecting a model needs no native library;
    tests/test_lib_abi.py holds the two in agreement."""
    if Ci <= 0 or Co <= 0 or Ci > 64 or Co > 64:
        return False
    up = lambda a, b: (a + b - 1) // b * b
    TV = T * V
    LD = TV + 1 if TV % 2 == 0 else TV
    NB = 1 if Ci >= 32 else 32 // Ci
    CiP, KZ, K1 = up(Ci, 16), up(Ci, 4), up(Co, 4)
    tables = T * V * V + V * T * T
    data = (NB * Ci * LD + tables + 2 * (KZ + K1) * CiP + 2 * CiP) * 4
    CH = ((TV + 2) // 3 + 3) // 4 * 4
    red = (Ci * LD + Co * (CH + 1) + tables) * 4
    fwd = (NB * Ci * LD + tables + 2 * KZ * up(Co, 16) + up(Co, 16)) * 4
    cap = 160 * 1024
    return data <= cap and red <= cap and fwd <= cap


class _GcnFn(torch.autograd.Function):
    """ConvTemporalGraphical.forward (reference stsgcn.py:143-156) and its gradients."""

    @staticmethod
    def forward(ctx, X, A, T):
        X = X.contiguous()
        ctx.save_for_backward(X, A, T)
        return ops.gcn(X, A.contiguous(), T.contiguous(), adjoint=False)

    @staticmethod
    def backward(ctx, dZ):
        X, A, T = ctx.saved_tensors
        dZ = dZ.contiguous()
        dX = ops.gcn(dZ, A, T, adjoint=True) if ctx.needs_input_grad[0] else None
        dA = dT = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dA, dT = ops.gcn_bwd_params(X, dZ, A, T)
        return dX, dA, dT


class ConvTemporalGraphical(nn.Module):
    """Learned space-time-separable mixing: T[V,T,T] per joint, A[T,V,V] per frame
    (reference stsgcn.py:120-156; init stsgcn.py:134-140)."""

    def __init__(self, time_dim: int, joints_dim: int) -> None:
        super().__init__()
        self.A = nn.Parameter(torch.empty(time_dim, joints_dim, joints_dim))
        stdv = 1.0 / math.sqrt(self.A.size(1))
        self.A.data.uniform_(-stdv, stdv)
        self.T = nn.Parameter(torch.empty(joints_dim, time_dim, time_dim))
        stdv = 1.0 / math.sqrt(self.T.size(1))
        self.T.data.uniform_(-stdv, stdv)

    def forward(self, X: Tensor) -> Tensor:
        return _GcnFn.apply(X, self.A, self.T)


def check_bn(tb, rb=None) -> None:
    """The two BatchNorms of a layer share one statistics / fold kernel: they must agree on HOW they average (fixed momentum
    or momentum=None, the cumulative moving average) and on whether they track running statistics at all.  nn.BatchNorm's
    non-default configurations themselves are on the HIP path (ops.bn_momentum / ops.bn_batch_stats); the reference builds the
    default one (stsgcn.py:65,76)."""
    if rb is None:
        return
    if (tb.momentum is None) != (rb.momentum is None) or (tb.momentum is not None and tb.momentum != rb.momentum):
        raise NotImplementedError("coskad_amd: the tcn and residual BatchNorm of a layer must use the same momentum")
    if (tb.running_mean is None) != (rb.running_mean is None):
        raise NotImplementedError("coskad_amd: the tcn and residual BatchNorm of a layer must both (or neither) track running statistics")
    if tb.momentum is None and tb.num_batches_tracked is not None and rb.num_batches_tracked is not None:
        a, b = tb.__dict__.get("_coskad_nbt"), rb.__dict__.get("_coskad_nbt")
        if a is None and b is None and int(tb.num_batches_tracked) != int(rb.num_batches_tracked):
            raise NotImplementedError("coskad_amd: momentum=None with different num_batches_tracked on the two BatchNorms of a layer")


def layer_tensors(layer: "ST_GCNN_layer") -> engine.LayerTensors:
    tc, tb = layer.tcn[0], layer.tcn[1]
    has_res = not isinstance(layer.residual, nn.Identity)
    rc, rb = (layer.residual[0], layer.residual[1]) if has_res else (None, None)
    check_bn(tb, rb)
    return engine.LayerTensors(
        A=layer.gcn.A, T=layer.gcn.T, Wt=tc.weight, bt=tc.bias, gt=tb.weight, bet=tb.bias,
        rm_t=tb.running_mean, rv_t=tb.running_var, nbt_t=tb.num_batches_tracked,
        Wr=rc.weight if has_res else None, br=rc.bias if has_res else None,
        gr=rb.weight if has_res else None, ber=rb.bias if has_res else None,
        rm_r=rb.running_mean if has_res else None, rv_r=rb.running_var if has_res else None,
        nbt_r=rb.num_batches_tracked if has_res else None, slope=layer.prelu.weight,
        momentum=tb.momentum if tb.momentum is not None else 0.1, bn=tb if tb.momentum is None else None,
        cache=layer.__dict__.setdefault("_fold_cache", {}))


class _ChainFn(torch.autograd.Function):
    """A stack of ST_GCNN layers as ONE autograd node.

    forward(x, in_slope, meta, *params) -> U_last (pre-activation of the last layer).
    meta = (layers, training, workspace).  Parameter order per layer = LayerTensors.param_list().
    """

    @staticmethod
    def forward(ctx, x, in_slope, meta, *params):
        layers, training, ws = meta
        x = x.contiguous()
        with torch.no_grad():
            # (an eval-mode chain has no backward: no context, so layers may run fused without their intermediates in HBM)
            u, cctx = engine.chain_forward(x, layers, training, ws, in_slope=in_slope, want_ctx=training)
        ctx.layers, ctx.cctx, ctx.ws, ctx.training = layers, cctx, ws, training
        return u

    @staticmethod
    def backward(ctx, dU):
        if not ctx.training:
            raise RuntimeError("coskad_amd: backward through an eval-mode (running-stats) ST-GCN chain is not "
                               "implemented; call .train() for training (reference trains in train mode)")
        layers = ctx.layers
        grads = []
        for L in layers:
            g = {n: torch.empty_like(p) for n, p in zip(L.grad_names(), L.param_list())}
            grads.append(g)
        # the last layer's slope gradient belongs to the consumer of U_last: zero here
        grads[-1]["slope"].zero_()
        need_dx = ctx.needs_input_grad[0]
        dx = engine.chain_backward(ctx.cctx, layers, dU.contiguous(), ctx.ws, grads, need_dx=need_dx)
        flat = []
        for L, g in zip(layers, grads):
            flat += [g[n] for n in L.grad_names()]
        d_in_slope = None  # an external in_slope is not a parameter of this chain
        return (dx, d_in_slope, None, *flat)


def run_chain(x: Tensor, layer_modules: List["ST_GCNN_layer"], ws: engine.Workspace,
              in_slope: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """-> (U_last, slope_last): pre-activation of the last layer and the PReLU weight to apply to it."""
    layers = [layer_tensors(m) for m in layer_modules]
    training = layer_modules[0].training
    params: List[Tensor] = []
    for L in layers:
        params += L.param_list()
    u = _ChainFn.apply(x, in_slope, (layers, training, ws), *params)
    return u, layers[-1].slope


class _PReLUFn(torch.autograd.Function):
    """PReLU with one shared weight on the HIP kernels; slope None = identity (input already activated)."""

    @staticmethod
    def forward(ctx, u, slope):
        u = u.contiguous()
        ctx.save_for_backward(u, slope)
        return ops.prelu_fwd(u, slope)

    @staticmethod
    def backward(ctx, dout):
        u, slope = ctx.saved_tensors
        dslope = torch.empty_like(slope)
        du = ops.prelu_bwd(u, dout.contiguous(), slope, dslope)
        return du, dslope


import os as _os
WIDE_RES_ADD = _os.environ.get("COSKAD_WIDE_RES_ADD", "1") != "0"   # (A/B hook: 0 = accumulate Wr^T dCr into dX behind the adjoint mix)


def wide_forward(X, A, T, Wt, bt, gt, bet, Wr, br, gr, ber, slope, bn_t, bn_r, training, drop_p=0.0, drop_seed=0):
    """out = PReLU(BN_t(Wt . gcn(X) + bt) + BN_r(Wr . X + br))  (identity residual when Wr is None) for one wide layer
    (reference stsgcn.py:106-110) -> (out [B, Co, T, V], saved, meta) for wide_backward; no autograd involved."""
    X = X.contiguous()
    B, Ci, Tn, V = X.shape
    Co, P = Wt.shape[0], Tn * V
    Z = ops.gcn(X, A.contiguous(), T.contiguous(), adjoint=False)
    Wt2 = Wt.view(Co, Ci)
    training = ops.bn_batch_stats(bn_t, training)   # batch statistics: training mode, or BatchNorms without running statistics
    # 1x1 convolutions: the layout-specialised MFMA kernel (csrc/conv1x1.hip) where the shape allows, with the train-mode
    # BatchNorm sums formed in its epilogue (no statistics pass over the conv output); the strided GEMM otherwise
    Ct, pt = ops.conv1x1(Wt2, Z.view(B, Ci, P), bias=bt, want_stats=training)
    st_t = ops.bn2_stats_parts(pt, bn_t, B * P) if pt is not None else ops.bn2_stats(Ct, bn_t, training)
    if Wr is not None:
        Cr, pr = ops.conv1x1(Wr.view(Co, Ci), X.view(B, Ci, P), bias=br, want_stats=training)
        st_r = ops.bn2_stats_parts(pr, bn_r, B * P) if pr is not None else ops.bn2_stats(Cr, bn_r, training)
    else:
        Cr, st_r = X.view(B, Ci, P), None
    out = ops.bn2_apply_prelu(Ct, Cr, st_t, gt, bet, st_r, gr, ber, slope, drop_p, drop_seed)
    saved = (X, Z, A, T, Wt, gt, bet, Wr, gr, ber, slope, Ct, Cr if Wr is not None else None, st_t, st_r)
    meta = (training, bt is not None, br is not None, (drop_p, drop_seed))
    return out.view(B, Co, Tn, V), saved, meta


def wide_backward(saved, meta, dOut, need_dx: bool = True, into: Optional[dict] = None):
    """-> (dX, dA, dT, dWt, dbt, dgt, dbet, dWr, dbr, dgr, dber, dslope) of wide_forward (None where the layer has no such tensor).
    `into`: destinations keyed by the layer's state_dict names ('gcn.A', 'tcn.0.weight', ...: views of a flat gradient buffer) --
    the kernels then write there (conv biases in front of a train-mode BatchNorm are not written: their gradient is exactly 0)."""
    into = into or {}
    X, Z, A, T, Wt, gt, bet, Wr, gr, ber, slope, Ct, Cr, st_t, st_r = saved
    training, has_bt, has_br, drop = meta
    B, Ci, Tn, V = X.shape
    Co, P = Wt.shape[0], Tn * V
    Xv = X.view(B, Ci, P)
    Crv = Cr if Cr is not None else Xv
    dst = lambda n, shape=None: None if into.get(n) is None else (into[n] if shape is None else into[n].view(shape))
    dCt, dCr, dgt, dbet, dgr, dber, dslope = ops.bn2_bwd(
        Ct, Crv, dOut.contiguous().view(B, Co, P), st_t, gt, bet, st_r, gr, ber, slope, training, *drop,
        into={"gt": dst("tcn.1.weight"), "bt": dst("tcn.1.bias"), "gr": dst("residual.1.weight"), "br": dst("residual.1.bias"),
              "slope": dst("prelu.weight", (1,))})
    Wt2 = Wt.view(Co, Ci)
    dWt_out = dst("tcn.0.weight", (Co, Ci))
    dWt = ops.conv1x1_wgrad(dCt, Z.view(B, Ci, P), dWt_out if dWt_out is not None else torch.empty(Co, Ci, device=X.device, dtype=torch.float32))
    dZ = ops.conv1x1(Wt2.t(), dCt)[0].view(B, Ci, Tn, V)
    # dA, dT and dX = gcn^T(dZ) from ONE pass over dZ (csrc/stsgcn_bwd.hip: k_bwd_gcn_params writes the adjoint mix too); an
    # identity residual's gradient joins it there instead of in an add of its own
    # the residual branch's data gradient joins the adjoint mix as its `add` input (an identity residual: dCr itself; a conv residual:
    # Wr^T dCr formed first) instead of a read-modify-write pass over dX behind it
    if Wr is None:
        add = dCr.view(B, Ci, Tn, V)
    elif need_dx and WIDE_RES_ADD:
        add = ops.conv1x1(Wr.view(Co, Ci).t(), dCr)[0].view(B, Ci, Tn, V)
    else:
        add = None
    dA, dT, dX = ops.gcn_bwd_params_dx(X, dZ, A, T, add=add, dA=dst("gcn.A"), dT=dst("gcn.T"))
    dWr = dbr = None
    if Wr is not None:
        dWr_out = dst("residual.0.weight", (Co, Ci))
        dWr = ops.conv1x1_wgrad(dCr, Xv, dWr_out if dWr_out is not None else torch.empty(Co, Ci, device=X.device, dtype=torch.float32)).view_as(Wr)
        if need_dx and add is None:
            ops.conv1x1(Wr.view(Co, Ci).t(), dCr, out=dX.view(B, Ci, P), accumulate=True)
        if has_br:      # a bias in front of a BatchNorm: its gradient is the sum of a mean-free tensor (exactly 0 in training)
            dbr = dCr.sum((0, 2)) if not training else (None if into else torch.zeros(Co, device=X.device, dtype=torch.float32))
    dbt = None
    if has_bt:
        dbt = dCt.sum((0, 2)) if not training else (None if into else torch.zeros(Co, device=X.device, dtype=torch.float32))
    return dX, dA, dT, dWt.view_as(Wt), dbt, dgt, dbet, dWr, dbr, dgr, dber, dslope.view_as(slope)


class _WideLayerFn(torch.autograd.Function):
    """wide_forward / wide_backward as one autograd node (the module surface; the flat train steps call the two functions directly)."""

    @staticmethod
    def forward(ctx, X, A, T, Wt, bt, gt, bet, Wr, br, gr, ber, slope, bn_t, bn_r, training, drop_p=0.0, drop_seed=0):
        out, saved, meta = wide_forward(X, A, T, Wt, bt, gt, bet, Wr, br, gr, ber, slope, bn_t, bn_r, training, drop_p, drop_seed)
        ctx.save_for_backward(*saved)
        ctx.meta = meta
        return out

    @staticmethod
    def backward(ctx, dOut):
        return wide_backward(ctx.saved_tensors, ctx.meta, dOut) + (None, None, None, None, None)


class ST_GCNN_layer(nn.Module):
    """Space-Time-Separable graph-conv block (reference stsgcn.py:9-116):
    out = PReLU( BN(Conv1x1(gcn(X))) + residual(X) ),  residual = BN(Conv1x1(X)) or identity."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: Union[Tuple[int], List[int]],
                 stride: int, time_dim: int, joints_dim: int, dropout: float, bias: bool = True,
                 emb_dim: int = None) -> None:
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.stride, self.time_dim, self.joints_dim = stride, time_dim, joints_dim
        self.dropout, self.bias, self.emb_dim, self.kernel_size = dropout, bias, emb_dim, kernel_size
        assert self.kernel_size[0] % 2 == 1
        assert self.kernel_size[1] % 2 == 1
        if tuple(kernel_size) != (1, 1) or stride != 1:
            raise NotImplementedError("coskad_amd ST_GCNN_layer: only kernel_size (1,1), stride 1 (what every "
                                      "reference Encoder/Decoder builds, components.py:77-78,150-151)")
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError(f"dropout probability has to be in [0, 1), but got {dropout}")
        if emb_dim is not None:
            raise NotImplementedError("coskad_amd ST_GCNN_layer: emb_dim is unused by the reference's models")
        self.build_model()
        self._ws = engine.Workspace()

    def build_model(self) -> None:
        self.gcn = ConvTemporalGraphical(self.time_dim, self.joints_dim)
        self.tcn = nn.Sequential(
            nn.Conv2d(self.in_channels, self.out_channels, (1, 1), (1, 1), (0, 0), bias=self.bias),
            nn.BatchNorm2d(self.out_channels),
            nn.Dropout(self.dropout, inplace=True))
        if self.stride != 1 or self.in_channels != self.out_channels:
            self.residual = nn.Sequential(
                nn.Conv2d(self.in_channels, self.out_channels, kernel_size=1, stride=(1, 1), bias=self.bias),
                nn.BatchNorm2d(self.out_channels))
        else:
            self.residual = nn.Identity()
        self.prelu = nn.PReLU()

    def train(self, mode: bool = True):
        if mode:
            # eval-mode folds (the BatchNorm-folded weights of engine.chain_forward, the folded images of models/sts/ae.py): training is
            # about to change what they fold, through raw-pointer kernels that move no torch version counter -- and not every training
            # path runs through this layer's own LayerTensors (trainer._FlatStack `narrow` segments, coskad_amd/lowrank.py)
            self.__dict__.pop("_lowrank_eval", None)
            self.__dict__.get("_fold_cache", {}).clear()
        return super().train(mode)

    @property
    def is_wide(self) -> bool:
        """Beyond the LDS-resident tile kernels: more than 64 channels on either side, or a clip whose images do not fit
        the 160 KB of LDS (64 input channels on the 25-joint layout: the default-width decoder of BASELINE config 4)."""
        if max(self.in_channels, self.out_channels) > WIDE_CHANNELS:
            return True
        if self.dropout > 0:
            # train-mode Dropout (stsgcn.py:66) sits between the tcn BatchNorm and the residual add, so the two branches cannot be
            # folded into one GEMM: such layers take the composed path, whose BatchNorm / add / PReLU kernels apply the mask
            # (every reference config sets dropout: 0)
            return True
        key = (self.in_channels, self.out_channels, self.time_dim, self.joints_dim)
        if key not in _FITS:
            _FITS[key] = layer_fits(*key)
        return not _FITS[key]

    def forward_wide(self, X: Tensor) -> Tensor:
        """Layers beyond the LDS-resident tile kernels (the C = 2 -> 256 stack of BASELINE.json's north_star; 64 input
        channels on the 25-joint layout): the reference's composition (stsgcn.py:106-110) on this repo's kernels in the
        native NCHW layout -- mixing (coskad_gcn_f32 + adjoint + dA/dT), both 1x1 convolutions and their gradients on the
        strided MFMA GEMM (csrc/gemm.hip), BatchNorm statistics / normalise + residual add + PReLU and their backward on
        csrc/wide.hip.  X is the post-activation input; returns the activated output."""
        return _WideLayerFn.apply(X, *self.wide_args())

    def wide_args(self):
        """the arguments of wide_forward behind X, for this layer in its current mode"""
        has_res = not isinstance(self.residual, nn.Identity)
        tc, tb = self.tcn[0], self.tcn[1]
        rc, rb = (self.residual[0], self.residual[1]) if has_res else (None, None)
        check_bn(tb, rb)
        return (self.gcn.A, self.gcn.T, tc.weight, tc.bias, tb.weight, tb.bias,
                rc.weight if has_res else None, rc.bias if has_res else None,
                rb.weight if has_res else None, rb.bias if has_res else None, self.prelu.weight,
                tb, rb, self.training, *self._dropout_args())

    def _dropout_args(self):
        """(p, seed) of this forward's train-mode Dropout mask: the seed is drawn from torch's CPU generator, so
        torch.manual_seed makes runs repeatable (no device synchronisation); p = 0 in eval mode or without dropout."""
        if not self.training or self.dropout <= 0:
            return 0.0, 0
        return float(self.dropout), int(torch.randint(0, 2 ** 62, (1,)).item())

    def forward(self, X: Tensor, t: Tensor = None) -> Tensor:
        if self.is_wide:
            return self.forward_wide(X)
        u, slope = run_chain(X, [self], self._ws)
        return _PReLUFn.apply(u, slope)


def run_stack(x: Tensor, layer_modules: List["ST_GCNN_layer"], ws: engine.Workspace,
              in_slope: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """A stack that may mix fused (<= 64 channels) and wide layers.  x: activated input, or (in_slope given) a pre-activation
    whose PReLU weight is in_slope (no gradient flows to it: eval-mode hand-over of coskad_amd/lowrank.py).  -> (h, slope): apply
    PReLU(slope) to h to get the stack's output (slope None: h is already activated)."""
    h, slope = x, in_slope
    i, n = 0, len(layer_modules)
    while i < n:
        if layer_modules[i].is_wide:
            if slope is not None:
                h, slope = _PReLUFn.apply(h, slope), None
            h = layer_modules[i].forward_wide(h)
            i += 1
        else:
            j = i
            while j < n and not layer_modules[j].is_wide:
                j += 1
            h, slope = run_chain(h, layer_modules[i:j], ws, in_slope=slope)
            i = j
    return h, slope
