"""CPU oracle for the COSKAD STS-GCN encoder hot path -- TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (plain PyTorch fp32 ops, no custom kernels) of
the reference algorithm for the hot path named in BASELINE.json.  It is the
*checker*: only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
leg may import it.  Nothing under coskad_amd/ imports it; the product path is
the HIP extension and fails loudly without it.

Parity pinning: every function here is checked against golden vectors that
oracle/make_golden.py produced by importing the reference's own modules
(models/sts/ae.py, utils/hyper_math.py, utils/model_utils.py) in the build
container; the vectors live in tests/golden/*.npz and
tests/test_oracle_golden.py replays them.  Paths whose reference code is not
importable (Lightning wrappers, geoopt, power_spherical) are restated from the
cited lines and are "parity unpinned" -- see DESIGN.md.

Style: functional.  The model state is a flat ``dict[str, Tensor]`` with the
reference's state_dict key names, so a reference checkpoint
drops straight in.  Autograd of these plain ops provides reference gradients.

Citations are file:line into the reference tree (aleflabo/COSKAD @ 2024-08-07).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor
State = Dict[str, Tensor]

BN_EPS = 1e-5        # nn.BatchNorm2d default, models/graph_layers/stsgcn.py:65,76
BN_MOMENTUM = 0.1    # nn.BatchNorm2d default


# --------------------------------------------------------------------------
# STS-GCN layer pieces            models/graph_layers/stsgcn.py
# --------------------------------------------------------------------------
def temporal_mix(x: Tensor, Tw: Tensor) -> Tensor:
    """Y[n,c,q,v] = sum_t X[n,c,t,v] * T[v,t,q]          (stsgcn.py:154)."""
    # written as a broadcast-multiply-sum, not einsum, to be an independent statement
    # x: [N,C,T,V] -> [N,C,T,1,V] ; Tw: [V,T,Q] -> [1,1,T,Q,V]
    w = Tw.permute(1, 2, 0)[None, None]
    return (x[:, :, :, None, :] * w).sum(dim=2)


def spatial_mix(y: Tensor, Aw: Tensor) -> Tensor:
    """Z[n,c,t,w] = sum_v Y[n,c,t,v] * A[t,v,w]          (stsgcn.py:155)."""
    return torch.matmul(y[:, :, :, None, :], Aw[None, None])[:, :, :, 0, :]


def gcn(x: Tensor, Aw: Tensor, Tw: Tensor) -> Tensor:
    """ConvTemporalGraphical.forward                      (stsgcn.py:143-156)."""
    return spatial_mix(temporal_mix(x, Tw), Aw)


def conv1x1(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """nn.Conv2d with kernel (1,1), stride 1              (stsgcn.py:57-64,71-75)."""
    co, ci = w.shape[0], w.shape[1]
    n, _, t, v = x.shape
    y = torch.matmul(w.reshape(co, ci), x.reshape(n, ci, t * v))
    if b is not None:
        y = y + b[None, :, None]
    return y.reshape(n, co, t, v)


def batchnorm(x: Tensor, st: State, prefix: str, training: bool, update: bool = True) -> Tensor:
    """nn.BatchNorm2d / BatchNorm1d semantics (stsgcn.py:65,76; components.py:212).

    training: biased batch variance for normalisation, unbiased for the running
    estimate, momentum 0.1, eps 1e-5.  eval: running stats.
    Reduction axes: all but the channel axis (dim 1).
    """
    gamma, beta = st[prefix + ".weight"], st[prefix + ".bias"]
    dims = [d for d in range(x.dim()) if d != 1]
    shape = [1, -1] + [1] * (x.dim() - 2)
    if training:
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=dims)
        var = ((x - mean.reshape(shape)) ** 2).mean(dim=dims)
        if update:
            with torch.no_grad():
                rm, rv = st[prefix + ".running_mean"], st[prefix + ".running_var"]
                rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
                rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var.detach() * (n / max(n - 1, 1)))
                st[prefix + ".num_batches_tracked"] += 1
    else:
        mean, var = st[prefix + ".running_mean"], st[prefix + ".running_var"]
    xhat = (x - mean.reshape(shape)) / torch.sqrt(var.reshape(shape) + BN_EPS)
    return xhat * gamma.reshape(shape) + beta.reshape(shape)


def prelu(x: Tensor, a: Tensor) -> Tensor:
    """nn.PReLU() with one shared slope                   (stsgcn.py:82,110)."""
    return torch.where(x > 0, x, a * x)


def st_gcnn_layer(x: Tensor, st: State, prefix: str, training: bool,
                  return_preact: bool = False, update: bool = True, drop_mask: Optional[Tensor] = None) -> Tensor:
    """ST_GCNN_layer.forward, kernel (1,1), stride 1 (stsgcn.py:94-116).  `drop_mask` (values 0 or 1 / (1 - p), shape of
    the output) stands for the train-mode nn.Dropout at the end of `tcn` (stsgcn.py:66); None = dropout 0 / eval mode.

    residual is Conv1x1+BN when the key exists (C_in != C_out, stsgcn.py:69-77),
    identity otherwise (stsgcn.py:79-80).
    """
    if prefix + ".residual.0.weight" in st:
        res = conv1x1(x, st[prefix + ".residual.0.weight"], st.get(prefix + ".residual.0.bias"))
        res = batchnorm(res, st, prefix + ".residual.1", training, update)
    else:
        res = x
    z = gcn(x, st[prefix + ".gcn.A"], st[prefix + ".gcn.T"])
    s = conv1x1(z, st[prefix + ".tcn.0.weight"], st.get(prefix + ".tcn.0.bias"))
    s = batchnorm(s, st, prefix + ".tcn.1", training, update)
    if drop_mask is not None:
        s = s * drop_mask
    u = s + res
    if return_preact:
        return u
    return prelu(u, st[prefix + ".prelu.weight"])


def n_layers(st: State, prefix: str) -> int:
    i = 0
    while f"{prefix}.{i}.gcn.A" in st:
        i += 1
    return i


def encoder(x: Tensor, st: State, training: bool, prefix: str = "encoder.model",
            collect: Optional[List[Tensor]] = None, update: bool = True) -> Tensor:
    """Encoder.forward: nn.Sequential of ST_GCNN layers   (components.py:70-105)."""
    for i in range(n_layers(st, prefix)):
        x = st_gcnn_layer(x, st, f"{prefix}.{i}", training, update=update)
        if collect is not None:
            collect.append(x)
    return x


def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    y = torch.matmul(x, w.t())
    return y if b is None else y + b


def mlp(x: Tensor, st: State, prefix: str, training: bool, update: bool = True) -> Tensor:
    """MLP: [Linear -> BatchNorm1d -> ReLU]* + Linear     (components.py:209-226, evident intent;
    the reference constructor is broken, SURVEY 8a row a8 -> parity unpinned)."""
    i = 0
    while f"{prefix}.net.{i}.weight" in st:
        w = st[f"{prefix}.net.{i}.weight"]
        if w.dim() == 2:  # Linear
            x = linear(x, w, st.get(f"{prefix}.net.{i}.bias"))
            if f"{prefix}.net.{i + 1}.running_mean" in st:
                x = batchnorm(x, st, f"{prefix}.net.{i + 1}", training, update)
                x = torch.relu(x)
                i += 3
                continue
        i += 1
    return x


def stse_encode(x: Tensor, st: State, training: bool, collect: Optional[List[Tensor]] = None,
                update: bool = True) -> Tensor:
    """STSE.encode: encoder, flatten in (c,t,v) order, bottleneck (ae.py:76-105).

    The unsqueeze/permute round trip at ae.py:89-93 is the identity for M=1.
    """
    assert x.dim() == 4, "Input tensor must have shape [batch_size, input_dim, n_frames, n_joints]"
    h = encoder(x, st, training, collect=collect, update=update)
    flat = h.reshape(h.shape[0], -1)
    if "btlnk.weight" in st:
        return linear(flat, st["btlnk.weight"], st.get("btlnk.bias"))
    if "btlnk.net.0.weight" in st:
        return mlp(flat, st, "btlnk", training, update)
    return flat  # nn.Identity (VAE with linear projector, vae.py:150)


def stsae_decode(z: Tensor, st: State, hidden: int, T: int, V: int, training: bool,
                 update: bool = True) -> Tensor:
    """STSAE.decode: rev_btlnk Linear -> view [B,hid,T,V] -> Decoder (ae.py:200-221,
    components.py:143-179)."""
    h = linear(z, st["rev_btlnk.weight"], st.get("rev_btlnk.bias"))
    h = h.reshape(z.shape[0], hidden, T, V)
    return encoder(h, st, training, prefix="decoder.model", update=update)


def stsvae_heads(z: Tensor, st: State, distribution: str = "ps") -> Tuple[Tensor, Tensor]:
    """Deterministic part of STSVAE.encode               (vae.py:79-85)."""
    zm = linear(z, st["fc_mean.weight"], st.get("fc_mean.bias"))
    if distribution == "ps":
        zm = zm / torch.norm(zm, dim=-1, keepdim=True)
    zv = torch.nn.functional.softplus(linear(z, st["fc_var.weight"], st.get("fc_var.bias"))) + 1
    return zm, zv


# --------------------------------------------------------------------------
# Regularisation                  utils/model_utils.py:90-105
# --------------------------------------------------------------------------
def calc_reg_loss(named_params: Sequence[Tuple[str, Tensor]]) -> Tensor:
    """0.5 * sum ||p||^2 over params whose NAME lacks 'bias', / #such tensors."""
    ps = [p for n, p in named_params if "bias" not in n]
    tot = None
    for p in ps:
        term = 0.5 * (p * p).sum()
        tot = term if tot is None else tot + term
    return tot / len(ps)


PARAM_SUFFIXES = (".gcn.A", ".gcn.T", ".weight", ".bias")


def is_param_key(k: str) -> bool:
    """state_dict keys that are nn.Parameters (everything but BN buffers, c, ...)."""
    if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
        return False
    if k in ("c", "inv_cov_matrix", "mean_vector", "threshold_dist"):
        return False
    return True


# --------------------------------------------------------------------------
# Poincare ball                   utils/hyper_math.py   (c = 1 throughout the hot path)
# --------------------------------------------------------------------------
MIN_NORM = 1e-5          # hyper_math.py:101,303,368
BALL_EPS = 1e-3          # hyper_math.py:102
ARTANH_EPS = 1e-5        # hyper_math.py:21
MOBIUS_DEN_EPS = 1e-5    # hyper_math.py:179
TANH_CLAMP = 15.0        # hyper_math.py:13


class _Artanh(torch.autograd.Function):
    """hyper_math.py:18-29: clamp to +-(1-1e-5); backward uses the CLAMPED input."""

    @staticmethod
    def forward(ctx, x):
        xc = x.clamp(-1 + ARTANH_EPS, 1 - ARTANH_EPS)
        ctx.save_for_backward(xc)
        return 0.5 * (torch.log(1 + xc) - torch.log(1 - xc))

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        return g / (1 - xc * xc)


def artanh(x: Tensor) -> Tensor:
    return _Artanh.apply(x)


def _norm(x: Tensor) -> Tensor:
    # torch's 2-norm: sub-gradient 0 at x == 0 (what hyper_math's x.norm(...) gives), not sqrt's NaN
    return torch.linalg.vector_norm(x, ord=2, dim=-1, keepdim=True)


def expmap0(u: Tensor, c: float = 1.0) -> Tensor:
    """hyper_math.py:302-306."""
    sc = math.sqrt(c)
    un = _norm(u).clamp_min(MIN_NORM)
    return torch.tanh((sc * un).clamp(-TANH_CLAMP, TANH_CLAMP)) * u / (sc * un)


def project(x: Tensor, c: float = 1.0) -> Tensor:
    """hyper_math.py:100-105."""
    n = _norm(x).clamp_min(MIN_NORM)
    maxnorm = (1 - BALL_EPS) / math.sqrt(c)
    return torch.where(n > maxnorm, x / n * maxnorm, x)


def mobius_add(x: Tensor, y: Tensor, c: float = 1.0) -> Tensor:
    """hyper_math.py:173-179."""
    x2 = (x * x).sum(-1, keepdim=True)
    y2 = (y * y).sum(-1, keepdim=True)
    xy = (x * y).sum(-1, keepdim=True)
    num = (1 + 2 * c * xy + c * y2) * x + (1 - c * x2) * y
    den = 1 + 2 * c * xy + c * c * x2 * y2
    return num / (den + MOBIUS_DEN_EPS)


def dist(x: Tensor, y: Tensor, c: float = 1.0) -> Tensor:
    """hyper_math.py:207-210."""
    sc = math.sqrt(c)
    m = mobius_add(-x, y, c)
    return artanh(sc * _norm(m)[..., 0]) * 2 / sc


def dist0(x: Tensor, c: float = 1.0) -> Tensor:
    """hyper_math.py:233-236."""
    sc = math.sqrt(c)
    return artanh(sc * _norm(x)[..., 0]) * 2 / sc


def logmap0(y: Tensor, c: float = 1.0) -> Tensor:
    """hyper_math.py:367-370."""
    sc = math.sqrt(c)
    yn = _norm(y).clamp_min(MIN_NORM)
    return y / yn / sc * artanh(sc * yn)


def poincare_mean(x: Tensor, c: float = 1.0) -> Tensor:
    """hyper_math.py:438-477 (Poincare -> Klein, Lorentz-weighted mean, back)."""
    k = 2 * x / (1 + c * (x * x).sum(-1, keepdim=True))
    lam = 1 / torch.sqrt(1 - c * (k * k).sum(-1, keepdim=True))
    mean = (lam * k).sum(0, keepdim=True) / lam.sum(0, keepdim=True)
    mean = mean / (1 + torch.sqrt(1 - c * (mean * mean).sum(-1, keepdim=True)))
    return mean[0]


def weighted_midpoint(x: Tensor, c: float = 1.0) -> Tensor:
    """Gyromidpoint with unit weights as geoopt 0.5.0 states it (called at
    models/hyperbolic_encoder.py:122,179): with gamma_i = lambda_x = 2/(1-c|x_i|^2),
    m = sum(gamma_i x_i) / sum(gamma_i - 1), midpoint = (1/2) (x) m  (Mobius scalar mul).
    geoopt is absent from the container -> parity unpinned against geoopt itself;
    mathematically equal to poincare_mean above, which is pinned (hyper_math.py).
    """
    gamma = 2 / (1 - c * (x * x).sum(-1, keepdim=True))
    m = (gamma * x).sum(0) / (gamma - 1).sum(0).clamp_min(1e-10)
    sc = math.sqrt(c)
    mn = torch.sqrt((m * m).sum()).clamp_min(1e-15)
    # mobius scalar mul by 1/2: tanh(0.5 * artanh(sc*|m|)) * m / (sc*|m|)
    at = 0.5 * torch.log((1 + (sc * mn).clamp(max=1 - 1e-7)) / (1 - (sc * mn).clamp(max=1 - 1e-7)))
    return torch.tanh(0.5 * at) * m / (sc * mn)


# --------------------------------------------------------------------------
# Loss heads                      models/*.py training_step
# --------------------------------------------------------------------------
def mse_to_center(z: Tensor, cvec: Tensor) -> Tensor:
    """F.mse_loss(hidden_out, c): broadcast [B,L] vs [L], mean over B*L
    (euclidean_encoder_staticCenter.py:187, euclidean_encoder_dynamicCenter.py:116)."""
    return ((z - cvec[None, :]) ** 2).mean()


def euclid_window_score(z: Tensor, cvec: Tensor) -> Tensor:
    """MSELoss(reduction='none')(c, z).mean(-1)           (eval_utils.py:63-64)."""
    return ((cvec[None, :] - z) ** 2).mean(-1)


def poincare_loss(z: Tensor, cvec: Tensor) -> Tuple[Tensor, Tensor]:
    """z_h = project(expmap0(z)); loss = dist(c, z_h).mean()   (hyperbolic_encoder.py:147,157).
    Returns (loss, z_h)."""
    zh = project(expmap0(z))
    return dist(cvec[None, :], zh).mean(), zh


def mahalanobis(u: Tensor, v: Tensor, VI: Tensor) -> Tensor:
    """sqrt((u-v)^T VI (u-v)) per row                     (eval_utils.py:28-38)."""
    d = u - v
    return torch.sqrt(torch.einsum("bi,ij,bj->b", d, VI, d))


def clamp_center(cvec: Tensor, eps: float) -> Tensor:
    """c[|c|<eps & c<0] = -eps ; c[|c|<eps & c>0] = +eps  (staticCenter.py:120-121)."""
    cvec = cvec.clone()
    cvec[(cvec.abs() < eps) & (cvec < 0)] = -eps
    cvec[(cvec.abs() < eps) & (cvec > 0)] = eps
    return cvec


# --------------------------------------------------------------------------
# Model construction with the reference's initialisers (for synthetic benches)
# --------------------------------------------------------------------------
def init_stse_state(input_dim: int = 2, layer_channels: Sequence[int] = (32, 16, 32), hidden: int = 64,
                    latent: int = 16, T: int = 12, V: int = 17, seed: int = 0,
                    decoder: bool = False) -> State:
    """Random-init state with the reference's distributions: A,T ~ U(+-1/sqrt(size(1)))
    (stsgcn.py:134-140); Conv2d/Linear: PyTorch default kaiming_uniform(a=sqrt(5)) =>
    U(+-1/sqrt(fan_in)) for weight and bias; BN weight 1, bias 0; PReLU 0.25.
    Not bit-identical to the reference's RNG stream (different draw order); the
    golden fixtures carry the reference's own draws where bit parity matters."""
    g = torch.Generator().manual_seed(seed)
    st: State = {}

    def U(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def stack(prefix, chans):
        for i in range(len(chans) - 1):
            ci, co = chans[i], chans[i + 1]
            p = f"{prefix}.{i}"
            st[p + ".gcn.A"] = U((T, V, V), 1 / math.sqrt(V))
            st[p + ".gcn.T"] = U((V, T, T), 1 / math.sqrt(T))
            for br, bn in (("tcn.0", "tcn.1"), ("residual.0", "residual.1")):
                if br.startswith("residual") and ci == co:
                    continue
                st[f"{p}.{br}.weight"] = U((co, ci, 1, 1), 1 / math.sqrt(ci))
                st[f"{p}.{br}.bias"] = U((co,), 1 / math.sqrt(ci))
                st[f"{p}.{bn}.weight"] = torch.ones(co)
                st[f"{p}.{bn}.bias"] = torch.zeros(co)
                st[f"{p}.{bn}.running_mean"] = torch.zeros(co)
                st[f"{p}.{bn}.running_var"] = torch.ones(co)
                st[f"{p}.{bn}.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
            st[p + ".prelu.weight"] = torch.full((1,), 0.25)

    stack("encoder.model", [input_dim] + list(layer_channels) + [hidden])
    K = hidden * T * V
    st["btlnk.weight"] = U((latent, K), 1 / math.sqrt(K))
    st["btlnk.bias"] = U((latent,), 1 / math.sqrt(K))
    st["c"] = torch.zeros(latent)
    if decoder:
        st["rev_btlnk.weight"] = U((K, latent), 1 / math.sqrt(latent))
        st["rev_btlnk.bias"] = U((K,), 1 / math.sqrt(latent))
        stack("decoder.model", [hidden] + list(layer_channels)[::-1] + [input_dim])
    return st


def synthetic_clips(B: int, C: int = 2, T: int = 12, V: int = 17, seed: int = 0) -> Tensor:
    """SURVEY 8d synthetic input: 0.5*N(0,1) clipped to +-3 with 2% exact zeros."""
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(B, C, T, V, generator=g) * 0.5).clamp(-3, 3)
    mask = torch.rand(B, 1, T, V, generator=g) < 0.02
    return torch.where(mask, torch.zeros(()), x).contiguous()
