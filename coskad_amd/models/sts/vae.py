"""Mirror of the reference's models/sts/vae.py (STSVAE) on the HIP path.

Encoder / decoder run on the gfx950 kernels (STSAE); the two small heads (`fc_mean`, `fc_var`) and the
sampler are torch ops on [B, latent] tensors.  The reference imports `power_spherical` (nicola-decao), which is
neither vendored nor pinned: `PowerSpherical` / `HypersphericalUniform` below restate the published
algorithm (De Cao & Aziz, "The Power Spherical distribution", 2020) -- parity unpinned, checked by moments.
"""
from __future__ import annotations

import math
from typing import List, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ..common.components import MLP
from ..graph_layers.stsgcn import _PReLUFn
from .ae import STSAE

Tensor = torch.Tensor


class HypersphericalUniform:
    """Uniform distribution on S^{dim} (dim = latent_dim - 1)."""

    def __init__(self, dim: int, device="cpu") -> None:
        self.dim, self.device = dim, device

    def entropy(self) -> Tensor:
        d = self.dim + 1
        return torch.tensor(math.log(2) + (d / 2) * math.log(math.pi) - math.lgamma(d / 2), device=self.device)


class PowerSpherical:
    """p(x; mu, kappa) ∝ (1 + mu^T x)^kappa on S^{d-1}; rsample via t ~ 2 Beta(a,b) - 1 and a Householder
    reflection of e1 onto mu (reparameterised through the Beta sample)."""

    def __init__(self, loc: Tensor, scale: Tensor) -> None:
        self.loc, self.scale = loc, scale
        d = loc.shape[-1]
        self.alpha = (d - 1) / 2 + scale
        self.beta = torch.full_like(scale, (d - 1) / 2)

    def rsample(self) -> Tensor:
        d = self.loc.shape[-1]
        z = torch.distributions.Beta(self.alpha, self.beta).rsample()
        t = (2 * z - 1).unsqueeze(-1)
        v = F.normalize(torch.randn(*self.loc.shape[:-1], d - 1, device=self.loc.device, dtype=self.loc.dtype), dim=-1)
        y = torch.cat([t, torch.sqrt(torch.clamp(1 - t * t, min=0)) * v], -1)
        e1 = torch.zeros_like(self.loc)
        e1[..., 0] = 1
        u = F.normalize(e1 - self.loc, dim=-1)
        return y - 2 * (y * u).sum(-1, keepdim=True) * u

    def log_normalizer(self) -> Tensor:
        return -((self.alpha + self.beta) * math.log(2) + torch.lgamma(self.alpha) - torch.lgamma(self.alpha + self.beta)
                 + self.beta * math.log(math.pi))

    def entropy(self) -> Tensor:
        return -(self.log_normalizer() + self.scale * (math.log(2) + torch.digamma(self.alpha)
                                                       - torch.digamma(self.alpha + self.beta)))


def kl_ps_uniform(q: PowerSpherical, p: HypersphericalUniform) -> Tensor:
    """KL(PowerSpherical || HypersphericalUniform) = -H(q) + H(p)  (the term at spherical_vae.py:92)."""
    return -q.entropy() + p.entropy().to(q.scale.device)


class STSVAE(STSAE):
    """STSAE + mean / concentration heads + reparameterised sample (reference vae.py:13-170).
    forward -> (Z, X_rec, (q_Z, p_Z, Z_var))."""

    def __init__(self, *args, distribution: str = 'ps', **kwargs) -> None:
        self.distribution = distribution.lower()
        super().__init__(*args, **kwargs)

    def build_model(self) -> None:
        super().build_model()
        if self.distribution == 'normal':
            self.register_buffer('mean_vector', torch.zeros((1, self.latent_dim)))
        self.register_buffer('threshold_dist', torch.tensor(0, dtype=torch.float32))

    def _set_projector_type(self) -> None:
        input_size = self.hidden_dimension * self.n_frames * self.n_joints
        if self.projector == 'mlp':
            self.btlnk = MLP(input_size=input_size, output_size=self.latent_dim, hidden_size=[self.latent_dim])
            input_size = self.latent_dim
        else:
            self.btlnk = nn.Identity()
            assert self.projector == 'linear', f'Projector type {self.projector} not supported.'
        self.fc_mean = nn.Linear(in_features=input_size, out_features=self.latent_dim)
        if self.distribution == 'normal':
            var_out_features = self.latent_dim
        elif self.distribution == 'ps':
            var_out_features = 1
        else:
            raise ValueError(f'Distribution {self.distribution} not supported.')
        self.fc_var = nn.Linear(in_features=input_size, out_features=var_out_features)

    def encode(self, X: Tensor, return_shape: bool = False):
        assert len(X.shape) == 4, f'Input tensor must have shape [batch_size, input_dim, n_frames, n_joints]. Got {X.shape}'
        X_shape = (X.shape[0], self.hidden_dimension, self.n_frames, self.n_joints, 1)
        Z_mean, var_raw = self._raw_heads(X)
        return self._finish_heads(Z_mean, var_raw, X_shape, return_shape)

    def _raw_heads(self, X: Tensor):
        """-> (fc_mean's output before the normalisation, fc_var's output before softplus + 1)  (vae.py:79-85)"""
        B = X.shape[0]
        n_var = self.fc_var.out_features
        heads = self._heads_fused(X)
        if heads is not None:
            return heads
        U, slope = self.encoder.forward_preact(X)
        if isinstance(self.btlnk, nn.Identity) and self.latent_dim + n_var <= 16:
            # `linear` projector (vae.py:147-150): both heads read the flattened encoder output -- ONE pass of the
            # bottleneck kernel over U (PReLU fused into the load) with the two weights stacked
            from .ae import _BottleneckFn
            Wc = torch.cat([self.fc_mean.weight, self.fc_var.weight], 0)
            bc = torch.cat([self.fc_mean.bias, self.fc_var.bias], 0)
            H = _BottleneckFn.apply(U, slope, Wc, bc, self._ws)
            Z_mean, var_raw = H[:, :self.latent_dim], H[:, self.latent_dim:]
        else:
            if isinstance(self.btlnk, MLP) and self.btlnk.hip_ok:      # wide Linear + BN/ReLU/Linear tail on the HIP kernels
                from .ae import _BottleneckFn
                Z = self.btlnk.forward_preact(U, slope, self._ws, _BottleneckFn.apply)
            else:
                Z = self.btlnk((U if slope is None else _PReLUFn.apply(U, slope)).reshape(B, -1))
            Z_mean, var_raw = self.fc_mean(Z), self.fc_var(Z)
        return Z_mean, var_raw

    def _finish_heads(self, Z_mean: Tensor, var_raw: Tensor, X_shape, return_shape: bool):
        if self.distribution == 'ps':
            Z_mean = Z_mean / torch.norm(Z_mean, dim=-1, keepdim=True)
        Z_var = F.softplus(var_raw) + 1          # the `+ 1` prevents collapse (vae.py:85)
        if return_shape:
            return Z_mean, Z_var, X_shape
        return Z_mean, Z_var

    def _heads_fused(self, X: Tensor):
        """Eval-mode fast path (no gradient): the fused encoder kernel, then
          `linear` projector (vae.py:147-150): ONE bottleneck pass over its tile-major output with fc_mean | fc_var stacked;
          `mlp` projector (vae.py:141-146): the projector as in STSE, then both small heads as one strided MFMA GEMM.
        -> (Z_mean pre-normalisation, var_raw) or None."""
        L, n_var = self.latent_dim, self.fc_var.out_features
        if isinstance(self.btlnk, nn.Identity):
            r = self._fused_hidden(X, (self.fc_mean.weight, self.fc_var.weight))
            if r is None:
                return None
            Hh, plan = r
            Hd = ops.btlnk_fwd(Hh, plan.wb, torch.cat([self.fc_mean.bias, self.fc_var.bias]), None, ws=self._ws)
            return Hd[:, :L], Hd[:, L:]
        Z = self._project_fused(X)
        if Z is None:
            return None
        Wc = torch.cat([self.fc_mean.weight, self.fc_var.weight], 0)          # [L + n_var, latent]
        Hd = ops.gemm(Z, Wc.t(), bias=torch.cat([self.fc_mean.bias, self.fc_var.bias]), bias_mode=2)
        return Hd[:, :L], Hd[:, L:]

    def reparameterize(self, Z_mean: Tensor, Z_var: Tensor):
        if self.distribution == 'normal':
            q_Z = torch.distributions.normal.Normal(Z_mean, Z_var)
            p_Z = torch.distributions.normal.Normal(torch.zeros_like(Z_mean), torch.ones_like(Z_var))
        else:
            q_Z = PowerSpherical(loc=Z_mean, scale=torch.squeeze(Z_var, dim=-1))
            p_Z = HypersphericalUniform(self.latent_dim - 1, device=Z_mean.device)
        return q_Z, p_Z

    def forward(self, X: Tensor):
        if (not self.training and not torch.is_grad_enabled() and X.is_cuda and self.distribution == 'ps' and 2 <= self.latent_dim <= 16):
            # scoring forward (spherical_vae.py:76-78): normalise, softplus + 1 and the PowerSpherical sample on csrc/vae_head.hip
            # (two launches around torch's Beta draw and Gaussian direction: the module path's noise streams) instead of ~40
            # element-wise launches
            assert len(X.shape) == 4, f'Input tensor must have shape [batch_size, input_dim, n_frames, n_joints]. Got {X.shape}'
            input_shape = (X.shape[0], self.hidden_dimension, self.n_frames, self.n_joints, 1)
            m_raw, v_raw = self._raw_heads(X)
            Z, _, _, saved = ops.ps_head_forward(m_raw, v_raw)
            Z_var = saved[3].unsqueeze(-1)
            q_Z, p_Z = self.reparameterize(saved[2], Z_var)
            return Z, self.decode(Z, input_shape=input_shape), (q_Z, p_Z, Z_var)
        Z_mean, Z_var, input_shape = self.encode(X, return_shape=True)
        q_Z, p_Z = self.reparameterize(Z_mean, Z_var)
        Z = q_Z.rsample()
        X_rec = self.decode(Z, input_shape=input_shape)
        return Z, X_rec, (q_Z, p_Z, Z_var)
