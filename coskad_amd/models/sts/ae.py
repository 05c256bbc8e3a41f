"""Mirror of the reference's models/sts/ae.py (STSE, STSAE) on the HIP path.

Accepts the shipped keyword set (`input_dim, layer_channels, hidden_dimension, ...`, ae.py:16-18) AND
the legacy one the Lightning wrappers still use (`c_in, h_dim, channels`,
euclidean_encoder_staticCenter.py:77-80) -- SURVEY 8b.
"""
from __future__ import annotations

from typing import List, Optional, Tuple, Union

import torch
import torch.nn as nn

from ... import engine, ops
from ..common.alternative_components import EncoderLearnablePlainGCN, EncoderStaticPlainGCN
from ..common.components import MLP, Decoder, Encoder
from ..graph_layers.stsgcn import _PReLUFn

Tensor = torch.Tensor


class _BottleneckFn(torch.autograd.Function):
    """z = Linear(flatten(PReLU_slope(U)))  (reference ae.py:97-101): PReLU fused into the load."""

    @staticmethod
    def forward(ctx, U, slope, W, b, ws):
        U = U.contiguous()
        ctx.save_for_backward(U, slope, W)
        ctx.ws, ctx.has_bias = ws, b is not None
        return ops.btlnk_fwd(U, W.contiguous(), b, slope, ws=ws)

    @staticmethod
    def backward(ctx, dz):
        U, slope, W = ctx.saved_tensors
        B, L = dz.shape
        K = W.shape[1]
        dW = torch.empty_like(W)
        db = torch.empty(L, device=W.device, dtype=W.dtype) if ctx.has_bias else None
        dslope = torch.empty_like(slope) if slope is not None else None   # no activation in front: plain-GCN encoders
        buf = ctx.ws.get(ops.btlnk_bwd_ws_bytes(B, K, L), U.device)
        dU = ops.btlnk_bwd(U, W, dz.contiguous(), slope, dW, db, dslope, buf)
        return dU, dslope, dW, db, None


class _RevBtlnkFn(torch.autograd.Function):
    """H = Linear(latent -> hidden*T*V)(Z)  (reference ae.py:223-227) and its autograd on csrc/rev_btlnk.hip."""

    @staticmethod
    def forward(ctx, Z, W, b):
        Z = Z.contiguous()
        ctx.save_for_backward(Z, W)
        ctx.has_bias = b is not None
        return ops.rev_btlnk_fwd(Z, W.contiguous(), b)

    @staticmethod
    def backward(ctx, dH):
        Z, W = ctx.saved_tensors
        dW = torch.empty_like(W)
        db = torch.empty(W.shape[0], device=W.device, dtype=W.dtype) if ctx.has_bias else None
        dz = ops.rev_btlnk_bwd(dH.contiguous(), Z, W.contiguous(), dW, db)
        return dz, dW, db


def _legacy(kw: dict, new: str, old: str, default=None):
    if new in kw and kw[new] is not None:
        return kw[new]
    if old in kw and kw[old] is not None:
        return kw[old]
    return default


class STSE(nn.Module):
    """STS-GCN encoder + bottleneck to a latent pulled towards a centre `c` (reference ae.py:12-165)."""

    encoder_classes = {'sts_gcn': Encoder, 'learnable_gcn': EncoderLearnablePlainGCN, 'static_gcn': EncoderStaticPlainGCN}

    def __init__(self, input_dim: int = None, layer_channels: List[int] = None, hidden_dimension: int = None,
                 latent_dim: int = None, n_frames: int = None, n_joints: int = None, encoder_type: str = 'sts_gcn',
                 projector: str = 'linear', distance: str = 'euclidean', dropout: float = 0.0, bias: bool = True,
                 device: Union[str, torch.device] = 'cpu', *, projector_hidden_layers: List[int] = None,
                 c_in: int = None, h_dim: int = None, channels: List[int] = None, **unused) -> None:
        super().__init__()
        self.input_dim = _legacy(dict(a=input_dim, b=c_in), 'a', 'b')
        self.layer_channels = list(_legacy(dict(a=layer_channels, b=channels), 'a', 'b'))
        self.hidden_dimension = _legacy(dict(a=hidden_dimension, b=h_dim), 'a', 'b')
        self.latent_dim, self.n_frames, self.n_joints = latent_dim, n_frames, n_joints
        self.encoder_type = encoder_type.lower()
        self.projector = projector.lower()
        # the reference's yamls select 'mlp' without giving sizes: default [latent_dim]
        self.projector_hidden_layers = projector_hidden_layers if projector_hidden_layers is not None else [latent_dim]
        self.distance = distance.lower()
        self.dropout, self.bias, self.device = dropout, bias, device
        self._ws = engine.Workspace()
        self.build_model()

    def build_model(self) -> None:
        self._set_encoder_type()
        self._set_projector_type()
        self.register_buffer('c', torch.zeros(self.latent_dim))
        if self.distance == 'mahalanobis':
            self.register_buffer('inv_cov_matrix', torch.zeros((self.latent_dim, self.latent_dim)))

    def _set_encoder_type(self) -> None:
        if self.encoder_type in self.encoder_classes:
            self.encoder = self.encoder_classes[self.encoder_type](
                input_dim=self.input_dim, layer_channels=self.layer_channels, hidden_dimension=self.hidden_dimension,
                n_frames=self.n_frames, n_joints=self.n_joints, dropout=self.dropout, bias=self.bias, device=self.device)
        elif self.encoder_type == 'st_gcn':
            raise NotImplementedError("coskad_amd: encoder type st_gcn is not mirrored (its constructor raises a "
                                      "TypeError in the reference snapshot, SURVEY 8c)")
        else:
            raise ValueError(f'Encoder type {self.encoder_type} not supported.')

    def _set_projector_type(self) -> None:
        input_size = self.hidden_dimension * self.n_frames * self.n_joints
        if self.projector == 'linear':
            self.btlnk = nn.Linear(in_features=input_size, out_features=self.latent_dim, bias=self.bias)
        elif self.projector == 'mlp':
            self.btlnk = MLP(input_size=input_size, output_size=self.latent_dim,
                             hidden_size=self.projector_hidden_layers, bias=self.bias, device=self.device)
        else:
            raise ValueError(f'Projector type {self.projector} not supported.')

    def encode(self, X: Tensor, return_shape: bool = False):
        assert len(X.shape) == 4, f'Input tensor must have shape [batch_size, input_dim, n_frames, n_joints]. Got {X.shape}'
        B = X.shape[0]
        Zf = self._encode_fused(X)
        if Zf is not None:
            return (Zf, (B, self.hidden_dimension, self.n_frames, self.n_joints, 1)) if return_shape else Zf
        if hasattr(self.encoder, 'forward_preact'):
            U, slope = self.encoder.forward_preact(X)   # [B, hid, T, V] pre-activation of the last layer
        else:
            U, slope = self.encoder(X), None            # plain-GCN encoders: already activated (ReLU)
        X_shape = (B, self.hidden_dimension, self.n_frames, self.n_joints, 1)
        if isinstance(self.btlnk, nn.Linear) and self.latent_dim <= 16:
            Z = _BottleneckFn.apply(U, slope, self.btlnk.weight, self.btlnk.bias, self._ws)
        elif isinstance(self.btlnk, MLP) and self.btlnk.hip_ok:
            Z = self.btlnk.forward_preact(U, slope, self._ws, _BottleneckFn.apply)
        else:
            Z = self.btlnk((U if slope is None else _PReLUFn.apply(U, slope)).reshape(B, -1))
        if return_shape:
            return Z, X_shape
        return Z

    def _fused_hidden(self, X: Tensor, Ws):
        """Eval-mode fast path, first half: the whole encoder in ONE kernel (csrc/fused_fwd.hip: every activation stays in
        LDS / registers).  -> (H [B, KP]: the activated last layer in tile-major order, plan with `wb` = the weights `Ws`
        stacked and permuted the same way), or None when the model is outside the kernel's geometry (T = 12, V = 17,
        channels 2-32-16-32-64, at most 16 rows of weights) or a gradient is needed."""
        if self.training or torch.is_grad_enabled() or not X.is_cuda or not isinstance(self.encoder, Encoder):
            return None
        if sum(w.shape[0] for w in Ws) > 16:
            return None
        from ..graph_layers.stsgcn import layer_tensors
        mods = list(self.encoder.model)
        if len(mods) != 4 or any(m.is_wide for m in mods):
            return None
        layers = [layer_tensors(m) for m in mods]
        if not engine.fused_encoder_supported(layers, self.n_frames, self.n_joints):
            return None
        plan = self.__dict__.setdefault("_fused_plan", engine.FusedEncoderPlan()).get(layers, tuple(Ws))
        return ops.fused_encoder(X.contiguous(), plan.tab, plan.wreg, plan.slopes), plan

    def _project_fused(self, X: Tensor) -> Optional[Tensor]:
        """fused encoder + this model's projector (components.py:209-226 for `mlp`): the first (wide) Linear is the
        bottleneck kernel on the tile-major output, every [BatchNorm1d, ReLU, Linear] block behind it csrc/mlp_head.hip
        with the running statistics."""
        if isinstance(self.btlnk, nn.Linear):
            first, blocks = self.btlnk, []
        elif isinstance(self.btlnk, MLP) and self.btlnk.hip_ok:
            first, blocks = self.btlnk.net[0], self.btlnk.blocks()
            if any(bn.running_mean is None or bn.running_var is None for bn, _ in blocks):
                return None          # track_running_stats=False: batch statistics even in eval mode -> module path
        else:
            return None
        r = self._fused_hidden(X, (first.weight,))
        if r is None:
            return None
        H, plan = r
        y = ops.btlnk_fwd(H, plan.wb, first.bias, None, ws=self._ws)
        for bn, lin in blocks:
            y, _ = ops.mlp_head_fwd(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                    lin.weight, lin.bias, False, eps=bn.eps)
        return y

    def _encode_fused(self, X: Tensor) -> Optional[Tensor]:
        return self._project_fused(X)

    def forward(self, X: Tensor) -> Tensor:
        return self.encode(X)


class STSAE(STSE):
    """STSE + rev_btlnk Linear + mirrored STS-GCN decoder (reference ae.py:168-265).  forward -> (Z, X_rec)."""

    def build_model(self) -> None:
        super().build_model()
        self.rev_btlnk = nn.Linear(in_features=self.latent_dim,
                                   out_features=self.hidden_dimension * self.n_frames * self.n_joints)
        self._set_decoder_type()

    def _set_decoder_type(self) -> None:
        if self.encoder_type == 'sts_gcn':
            self.decoder = Decoder(self.input_dim, self.layer_channels, self.hidden_dimension, self.n_frames,
                                   self.n_joints, self.dropout, self.bias)
        else:
            raise ValueError(f'No decoder available for encoder type {self.encoder_type}.')

    def decode(self, Z: Tensor, input_shape: Tuple[int]) -> Tensor:
        B, C, T, V, M = input_shape
        folded = self._decode_folded(Z, B * M, T, V)
        if folded is not None:
            return folded
        if Z.is_cuda and Z.dtype == torch.float32 and ops.rev_btlnk_ok(self.rev_btlnk.out_features, self.latent_dim):
            H = _RevBtlnkFn.apply(Z, self.rev_btlnk.weight, self.rev_btlnk.bias)   # streaming kernels of csrc/rev_btlnk.hip
        else:
            H = self.rev_btlnk(Z)        # other latent sizes: torch
        H = H.view(B * M, C, T, V)
        return self.decoder(H)

    def _decode_folded(self, Z: Tensor, N: int, T: int, V: int) -> Optional[Tensor]:
        """Eval-mode fast path (no gradient): rev_btlnk + the decoder's first layer as ONE streaming pass where that layer would
        take the composed wide path (coskad_amd/lowrank.py: the latent makes its input rank latent + 1), the remaining layers
        as usual.  None: not applicable."""
        from ... import lowrank
        from ..graph_layers.stsgcn import run_stack
        mods = list(self.decoder.model) if isinstance(self.decoder, Decoder) else []
        if (self.training or torch.is_grad_enabled() or not Z.is_cuda or Z.dtype != torch.float32 or len(mods) < 2
                or not lowrank.eval_supported(self.rev_btlnk, mods[0])):
            return None
        # the folded images depend on parameters and running statistics only: kept until one of them changes (torch-side writes
        # move the version counters; this library's training kernels write through raw pointers, but training needs .train() first,
        # which drops the cache: ST_GCNN_layer.train)
        l0, rev = mods[0], self.rev_btlnk
        ts = [rev.weight, rev.bias, l0.gcn.A, l0.gcn.T] + [t for seq in (l0.tcn, l0.residual) if not isinstance(seq, nn.Identity)
                                                          for t in (seq[0].weight, seq[0].bias, seq[1].weight, seq[1].bias,
                                                                    seq[1].running_mean, seq[1].running_var) if t is not None]
        key = tuple((t.data_ptr(), t._version) for t in ts)
        cached = l0.__dict__.get("_lowrank_eval")
        if cached is None or cached[0] != key:
            cached = (key, lowrank.fold_eval(rev, l0))
            l0.__dict__["_lowrank_eval"] = cached
        Mw, Mb = cached[1]
        U1 = ops.rev_btlnk_fwd(Z.contiguous(), Mw, Mb).view(N, mods[0].out_channels, T, V)
        u, slope = run_stack(U1, mods[1:], self.decoder._ws, in_slope=mods[0].prelu.weight)
        return u if slope is None else _PReLUFn.apply(u, slope)

    def forward(self, X: Tensor) -> Tuple[Tensor]:
        Z, X_shape = self.encode(X, return_shape=True)
        return Z, self.decode(Z, X_shape)
