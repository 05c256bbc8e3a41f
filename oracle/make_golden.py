"""Generate golden fixtures by IMPORTING the reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
Needs /root/reference (read-only); writes tests/golden/*.npz.  The fixtures are
data (inputs, weights drawn by the reference's own initialisers, and the
reference's outputs / gradients); no reference source travels.

What is pinned:
  stse_default.npz  models/sts/ae.py::STSE (sts_gcn, linear, euclidean), T=12 V=17,
                    eval-mode per-layer activations + latent; train-mode latent,
                    MSE-to-centre loss, utils/model_utils.py::calc_reg_loss, all
                    parameter gradients, BN running stats after one step.
  stse_v25.npz      same on a small V=25 stack.
  stsae_small.npz   models/sts/ae.py::STSAE (decoder path), eval + train + grads.
  stsae_v25.npz     STSAE at the DEFAULT widths (channels 32-16-32, hidden 64, latent 8) on the 25-joint layout
                    (BASELINE config 4's model shape): eval + train + grads.
  hyper_math.npz    utils/hyper_math.py expmap0/project/mobius_add/dist/dist0/
                    logmap0/poincare_mean + autograd of the Poincare loss.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from models.sts.ae import STSE, STSAE  # noqa: E402  (reference)
import utils.hyper_math as hm          # noqa: E402  (reference)
from utils.model_utils import calc_reg_loss  # noqa: E402  (reference)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_cpu import synthetic_clips  # noqa: E402  (input generator only)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def sd_np(model, prefix="sd.", buffers_only=False):
    keep = {n for n, _ in model.named_buffers()} if buffers_only else None
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()
            if keep is None or k in keep}


def perturb_affine(model, seed):
    """BN affine/bias start at 1/0 and PReLU at 0.25: perturb so parity sees them."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if ".tcn.1." in n or ".residual.1." in n:
                p.add_(0.2 * torch.randn(p.shape, generator=g))
            if n.endswith("prelu.weight"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
        for n, b in model.named_buffers():
            if n.endswith("running_mean"):
                b.add_(0.1 * torch.randn(b.shape, generator=g))
            if n.endswith("running_var"):
                b.mul_(1 + 0.3 * torch.rand(b.shape, generator=g))


def layer_hooks(seq, store):
    hs = []
    for i, layer in enumerate(seq):
        hs.append(layer.register_forward_hook(lambda m, a, o, i=i: store.__setitem__(i, o.detach().numpy().copy())))
    return hs


def stse_case(name, B, cfg, seed, alpha=1e-6):
    torch.manual_seed(seed)
    model = STSE(cfg["input_dim"], cfg["channels"], cfg["hidden"], cfg["latent"], 12, cfg["V"],
                 "sts_gcn", "linear", "euclidean", 0.0)
    perturb_affine(model, seed + 1)
    x = synthetic_clips(B, cfg["input_dim"], 12, cfg["V"], seed=seed + 2)
    out = {"x": x.numpy(), "alpha": np.float32(alpha)}
    out.update(sd_np(model, "sd0."))

    # eval mode
    model.eval()
    acts = {}
    hs = layer_hooks(model.encoder.model, acts)
    with torch.no_grad():
        z_eval = model(x)
    for h in hs:
        h.remove()
    for i, a in acts.items():
        out[f"eval.act{i}"] = a
    out["eval.z"] = z_eval.numpy()

    # centre: mean of eval latents (as setup() does), then one train step
    c = z_eval.mean(0)
    model.c = c.clone()
    out["c"] = c.numpy()
    model.train()
    z = model(x)
    loss_h = torch.nn.functional.mse_loss(z, model.c)
    loss_r = calc_reg_loss(model)
    loss = loss_h + alpha * loss_r
    loss.backward()
    out["train.z"] = z.detach().numpy()
    out["train.loss_hypersphere"] = loss_h.detach().numpy()
    out["train.loss_reg"] = loss_r.detach().numpy()
    for n, p in model.named_parameters():
        out["grad." + n] = p.grad.numpy().copy()
    out.update(sd_np(model, "sd1.", buffers_only=True))  # BN running stats after the step

    # Poincare head on the same train-mode latents: loss and dL/dz (hyperbolic_encoder.py:147,157)
    zz = z.detach().clone().requires_grad_(True)
    ch = hm.project(hm.expmap0(c[None] * 0.5))[0]
    zh = hm.project(hm.expmap0(zz))
    lp = hm.dist(ch, zh).mean()
    lp.backward()
    out["hyp.c"] = ch.numpy()
    out["hyp.zh"] = zh.detach().numpy()
    out["hyp.loss"] = lp.detach().numpy()
    out["hyp.dz"] = zz.grad.numpy()
    out["hyp.center"] = hm.poincare_mean(zh.detach(), dim=0, c=1.0).numpy()

    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, {k: v.shape for k, v in out.items() if not k.startswith(("sd", "grad"))})


def stsae_case(name, B, cfg, seed):
    torch.manual_seed(seed)
    model = STSAE(cfg["input_dim"], cfg["channels"], cfg["hidden"], cfg["latent"], 12, cfg["V"],
                  "sts_gcn", "linear", "euclidean", 0.0)
    perturb_affine(model, seed + 1)
    x = synthetic_clips(B, cfg["input_dim"], 12, cfg["V"], seed=seed + 2)
    out = {"x": x.numpy()}
    out.update(sd_np(model, "sd0."))
    model.eval()
    with torch.no_grad():
        z, xr = model(x)
    out["eval.z"], out["eval.xrec"] = z.numpy(), xr.numpy()
    model.c = z.mean(0).clone()
    out["c"] = model.c.numpy()
    model.train()
    z, xr = model(x)
    loss = torch.nn.functional.mse_loss(xr, x) + torch.nn.functional.mse_loss(z, model.c)
    loss.backward()
    out["train.z"], out["train.xrec"] = z.detach().numpy(), xr.detach().numpy()
    out["train.loss"] = loss.detach().numpy()
    for n, p in model.named_parameters():
        out["grad." + n] = p.grad.numpy().copy()
    out.update(sd_np(model, "sd1.", buffers_only=True))
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "ok")


def hyper_case(name, seed=11):
    g = torch.Generator().manual_seed(seed)
    L = 16
    # rows spanning tiny, ordinary, large (tanh saturation / project clamp) norms
    scales = torch.tensor([1e-7, 1e-3, 0.05, 0.3, 1.0, 2.5, 6.0, 20.0]).repeat_interleave(8)
    u = torch.randn(64, L, generator=g)
    u = u / u.norm(dim=-1, keepdim=True) * scales[:, None]
    u[0] = 0.0
    out = {"u": u.numpy()}
    e = hm.expmap0(u)
    p = hm.project(e)
    out["expmap0"], out["project_expmap0"] = e.numpy(), p.numpy()
    raw = torch.randn(64, L, generator=g) * 0.4
    out["raw"], out["project_raw"] = raw.numpy(), hm.project(raw).numpy()
    a = hm.project(hm.expmap0(torch.randn(64, L, generator=g) * 0.3))
    out["a"] = a.numpy()
    out["mobius_add"] = hm.mobius_add(a, p).numpy()
    out["dist"] = hm.dist(a, p).numpy()
    out["dist_bcast"] = hm.dist(a[3], p).numpy()
    out["dist0"] = hm.dist0(p).numpy()
    out["logmap0"] = hm.logmap0(p).numpy()
    out["poincare_mean"] = hm.poincare_mean(p, dim=0, c=1.0).numpy()
    uu = u.clone().requires_grad_(True)
    loss = hm.dist(a[3], hm.project(hm.expmap0(uu))).mean()
    loss.backward()
    out["loss"], out["dloss_du"] = loss.detach().numpy(), uu.grad.numpy()
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "ok")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "v25":       # only the round-2 fixture (the others are unchanged)
        stsae_case("stsae_v25.npz", 3, dict(input_dim=2, channels=[32, 16, 32], hidden=64, latent=8, V=25), seed=50)
        sys.exit(0)
    stse_case("stse_default.npz", 8, dict(input_dim=2, channels=[32, 16, 32], hidden=64, latent=16, V=17), seed=0)
    stse_case("stse_v25.npz", 4, dict(input_dim=2, channels=[8, 4, 8], hidden=8, latent=8, V=25), seed=20)
    stse_case("stse_b1.npz", 1, dict(input_dim=2, channels=[4, 4, 8], hidden=4, latent=4, V=17), seed=30)
    stsae_case("stsae_small.npz", 4, dict(input_dim=2, channels=[16, 8, 16], hidden=16, latent=8, V=17), seed=40)
    stsae_case("stsae_v25.npz", 3, dict(input_dim=2, channels=[32, 16, 32], hidden=64, latent=8, V=25), seed=50)
    hyper_case("hyper_math.npz")
