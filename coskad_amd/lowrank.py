"""The decoder's first ST_GCNN layer behind `rev_btlnk`, folded into ONE streaming pass (training mode).

`rev_btlnk` (reference models/sts/ae.py:223-227) maps the latent z [B, Lz] LINEARLY to the decoder's input H = Wrev z + brev, and
everything the first layer does in front of its BatchNorms is linear too (mixing and 1x1 convolutions: models/graph_layers/
stsgcn.py:106-110,154-155).  With zt = [z, 1] in R^K, K = Lz + 1, and the K "basis images" Hb = (columns of Wrev, brev):

    t[n] = Wt gcn(H[n]) + bt = sum_l zt_l[n] P_l,   P_l = Wt gcn(Hb_l) (+ bt on the last)
    r[n] = Wr H[n] + br      = sum_l zt_l[n] Q_l,   Q_l = Wr Hb_l      (+ br on the last;  Q = Hb for an identity residual)

The training-mode batch statistics of t and r over (clips, positions) follow from the K x K Gram matrix G = sum_n zt zt^T alone:

    mean_c = 1/N sum_l G[l, K-1] sum_p X_l[c, p],      E[x^2]_c = 1/N sum_{l,m} G[l, m] sum_p X_l[c, p] X_m[c, p]       (X = P, Q)

so the layer's pre-activation output is  U1[n] = sum_l zt_l[n] M_l  with the folded images  M_l = a_t P_l + a_r Q_l  (+ the BatchNorm
shifts on the last): the rev_btlnk forward kernel (csrc/rev_btlnk.hip) with M as its weight -- one pass that writes [B, C_out, T, V] --
instead of rev_btlnk, the mixing, two convolutions and a BatchNorm / add pass over [B, 64, T, V] tensors (the layer is `wide` on the
25-joint layout: 64 input channels do not fit the LDS tile kernels).  Backward: the rev_btlnk backward kernel with M as its weight
returns dM_l = sum_n zt_l[n] dU1[n] and the direct part of dz; how M depends on the layer's parameters, on Wrev / brev (K images: tiny
tensors) and on G is differentiated by torch autograd on [K, C, T V] tensors; dz += zt (dG + dG^T).  Same function, same gradients
(tests/test_gpu_ae_step.py holds it against the REFERENCE's gradients, tests/golden/stsae_*.npz); V = 25 spherical-VAE step
7.0 -> 5.4 ms, default-width autoencoder step at 17 joints 3.83 -> 3.08 ms.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops

Tensor = torch.Tensor

# which decoders take the folded first layer: 'always' (wherever rev_btlnk's latent size is one the streaming kernels take), 'wide' =
# only where the layer would run the composed wide path (64 input channels on the 25-joint layout), 'never'.  Measured at B = 4096,
# default-width autoencoder step (tools/bench_ae.py): 17 joints 3.83 -> 3.08 ms (64 input channels are beyond the fused 17-joint
# kernels too: that layer ran on the round-1 tile kernels), 25 joints 6.41 -> 4.87 ms.  Tests flip it.
MODE = 'always'
# True: the fold's ~60 small torch launches and the ~170 of its autograd backward are replayed as two hipGraphs per position count
# (their shapes do not depend on the batch).  Measured on the V = 25 VAE step (B = 4096): host enqueue time 4.4 -> 2.0 ms per step, but
# the step itself 5.53 -> 6.0 ms -- a graph node costs more device time than an eager back-to-back launch on this stack -- so the
# default is eager; a host that cannot keep 4.4 ms of launches ahead of 5.5 ms of kernels would flip it.  (A BatchNorm with
# momentum=None -- its averaging factor changes per step -- and a failed capture are always eager.)
GRAPH_FOLD = False
# the fold's backward written out (no autograd; the K-image mixing and its gradient on this library's gcn kernels): ~100 launches per
# step instead of ~215, 1.3 ms less host time -- the 17-joint VAE step was host-bound on the autograd fold (4.10 ms for 3.4 ms of
# kernels).  False: torch autograd (the form the tests hold the explicit one against); identity-residual layers always take it.
EXPLICIT = True
# with the explicit fold: its statistics algebra on csrc/lowrank_fold.hip (one launch each way; latent 8) / as ~35 torch launches
FOLD_KERNEL = True


class LowRankFirstLayer:
    def __init__(self, rev: nn.Linear, layer, gviews: dict, rev_prefix: str = "rev_btlnk.", layer_prefix: str = "decoder.model.0.") -> None:
        self.rev, self.layer, self.gv = rev, layer, gviews
        self.rev_prefix, self.layer_prefix = rev_prefix, layer_prefix
        self._saved = None

    @staticmethod
    def supports(rev, layer) -> bool:
        if not isinstance(rev, nn.Linear) or rev.bias is None or float(getattr(layer, "dropout", 0.0)) > 0:
            return False
        n_out = layer.out_channels * layer.time_dim * layer.joints_dim
        if rev.out_features != layer.in_channels * layer.time_dim * layer.joints_dim:
            return False
        return ops.rev_btlnk_ok(n_out, rev.in_features)

    # ---- the K folded images as a differentiable function of the parameters and of G ------------------------------------------------
    def _fold(self, G: Tensor, n_pos: float, update_running: bool):
        rev, lay = self.rev, self.layer
        Lz = rev.in_features
        K = Lz + 1
        Ci, Co, T, V = lay.in_channels, lay.out_channels, lay.time_dim, lay.joints_dim
        TV = T * V
        Hb = torch.cat([rev.weight.t(), rev.bias[None]], 0).view(K, Ci, T, V)
        Y = torch.einsum('kctv,vtq->kcqv', Hb, lay.gcn.T)                 # stsgcn.py:154
        Zb = torch.einsum('kctv,tvw->kctw', Y, lay.gcn.A)                 # stsgcn.py:155
        last = torch.zeros(K, 1, 1, device=G.device, dtype=torch.float32)
        last[K - 1] = 1.0                                                 # conv biases and BatchNorm shifts live on the constant image

        def conv(w, b, X):
            out = torch.matmul(w.view(Co, Ci), X.reshape(K, Ci, TV))
            return out if b is None else out + last * b.view(1, Co, 1)

        def fold(bns, X):
            """X [R, K, Co, TV]: the rank-K tensors of R BatchNorm branches -> (scale a [R, Co] fp32, shift [R, Co] fp32), batch
            statistics from G in fp64 (both branches in one set of launches)"""
            Xd = X.double()
            mean = torch.einsum('l,rlc->rc', G[:, K - 1], Xd.sum(-1)) / n_pos
            e2 = (torch.matmul(Xd.permute(0, 2, 1, 3), Xd.permute(0, 2, 3, 1)) * G).sum((2, 3)) / n_pos       # [R, Co, K, K] . G
            var = (e2 - mean * mean).clamp_min(0.0)
            if update_running:
                with torch.no_grad():                                      # nn.BatchNorm2d: momentum average, unbiased variance
                    unb = n_pos / (n_pos - 1.0) if n_pos > 1 else 1.0
                    mf, vf = mean.float(), (var * unb).float()
                    for r, bn in enumerate(bns):
                        if bn.running_mean is not None:
                            mom = ops.bn_momentum(bn)
                            bn.running_mean.mul_(1 - mom).add_(mf[r], alpha=mom)
                            bn.running_var.mul_(1 - mom).add_(vf[r], alpha=mom)
                            bn.num_batches_tracked.add_(1)
            gamma = torch.stack([bn.weight for bn in bns]).double()
            beta = torch.stack([bn.bias for bn in bns]).double()
            if all(bn.eps == bns[0].eps for bn in bns):
                eps = bns[0].eps
            else:                                                          # (built once: no host-to-device copy per step / under capture)
                key = tuple(bn.eps for bn in bns)
                cache = self.__dict__.setdefault("_eps", {})
                if key not in cache:
                    cache[key] = torch.tensor(key, device=G.device, dtype=torch.float64).view(-1, 1)
                eps = cache[key]
            a = gamma / torch.sqrt(var + eps)
            return a.float(), (beta - a * mean).float()

        tc, tb = lay.tcn[0], lay.tcn[1]
        P = conv(tc.weight, tc.bias, Zb)
        if isinstance(lay.residual, nn.Identity):
            a, shift = fold([tb], P[None])
            M = a[0].view(1, Co, 1) * P + Hb.reshape(K, Ci, TV)
            shift = shift[0]
        else:
            rc, rb = lay.residual[0], lay.residual[1]
            X = torch.stack([P, conv(rc.weight, rc.bias, Hb)])
            a, shift = fold([tb, rb], X)
            M = (a.view(2, 1, Co, 1) * X).sum(0)
            shift = shift.sum(0)
        M = M + last * shift.view(1, Co, 1)
        Mw = M[:K - 1].reshape(K - 1, Co * TV).t().contiguous()            # the streaming kernels' weight layout [features, Lz]
        Mb = M[K - 1].reshape(-1).contiguous()
        return Mw, Mb

    # ---- the same fold with its backward written out (conv-residual layers) ---------------------------------------------------
    def _fold_explicit(self, G: Tensor, n_pos: float):
        """-> (Mw, Mb, ctx); same arithmetic as _fold (conv biases drop out of a train-mode BatchNorm's output: they only shift the
        running means), intermediates kept for _fold_explicit_bwd"""
        rev, lay = self.rev, self.layer
        K = rev.in_features + 1
        Ci, Co, T, V = lay.in_channels, lay.out_channels, lay.time_dim, lay.joints_dim
        TV = T * V
        tc, tb, rc, rb = lay.tcn[0], lay.tcn[1], lay.residual[0], lay.residual[1]
        with torch.no_grad():
            Hb = torch.cat([rev.weight.t(), rev.bias[None]], 0).view(K, Ci, T, V)
            Zb = ops.gcn(Hb, lay.gcn.A, lay.gcn.T)
            X = torch.empty(2, K, Co, TV, device=G.device, dtype=torch.float32)
            torch.matmul(tc.weight.view(Co, Ci), Zb.view(K, Ci, TV), out=X[0])
            torch.matmul(rc.weight.view(Co, Ci), Hb.view(K, Ci, TV), out=X[1])
            if FOLD_KERNEL and ops.lowrank_fold_ok(rev.in_features, TV):      # the statistics algebra as ONE kernel (csrc/lowrank_fold.hip)
                Mw, Mb, kctx = ops.lowrank_fold_fwd(X, G, tb, rb, tc.bias, rc.bias, n_pos)
                return Mw, Mb, ("kernel", Hb, Zb, X, kctx, n_pos)
            Xd = X.double()
            xbar = Xd.sum(-1)                                                # [2, K, Co]
            s = G[:, K - 1]
            mean = (s.view(1, K, 1) * xbar).sum(1) / n_pos                    # [2, Co] (without the conv biases)
            XX = torch.matmul(Xd.permute(0, 2, 1, 3), Xd.permute(0, 2, 3, 1))  # [2, Co, K, K]
            var = ((XX * G).sum((2, 3)) / n_pos - mean * mean).clamp_min(0.0)
            unb = n_pos / (n_pos - 1.0) if n_pos > 1 else 1.0
            mf, vf = mean.float(), (var * unb).float()
            for r, (bn, cv) in enumerate(((tb, tc), (rb, rc))):
                if bn.running_mean is not None:
                    mom = ops.bn_momentum(bn)
                    bn.running_mean.mul_(1 - mom).add_(mf[r] + cv.bias if cv.bias is not None else mf[r], alpha=mom)
                    bn.running_var.mul_(1 - mom).add_(vf[r], alpha=mom)
                    bn.num_batches_tracked.add_(1)
            gamma = torch.stack([tb.weight, rb.weight]).double()
            beta = torch.stack([tb.bias, rb.bias]).double()
            eps = tb.eps if tb.eps == rb.eps else torch.tensor([tb.eps, rb.eps], device=G.device, dtype=torch.float64).view(2, 1)
            istd = torch.rsqrt(var + eps)
            a = gamma * istd
            M = (a.float().view(2, 1, Co, 1) * X).sum(0)
            M[K - 1] += (beta - a * mean).sum(0).float().view(Co, 1)
            Mw = M[:K - 1].reshape(K - 1, Co * TV).t().contiguous()
            Mb = M[K - 1].reshape(-1).contiguous()
        return Mw, Mb, ("torch", Hb, Zb, X, xbar, XX, mean, istd, a, gamma, n_pos)

    def _fold_explicit_bwd(self, G: Tensor, ctx, dMw: Tensor, dMb: Tensor):
        """-> gradients in _params() order, then dG"""
        rev, lay = self.rev, self.layer
        K = rev.in_features + 1
        Ci, Co, T, V = lay.in_channels, lay.out_channels, lay.time_dim, lay.joints_dim
        TV = T * V
        tc, rc = lay.tcn[0], lay.residual[0]
        if ctx[0] == "kernel":
            _, Hb, Zb, X, kctx, n_pos = ctx
            with torch.no_grad():
                dX, dgam, dbeta, dG = ops.lowrank_fold_bwd(X, G, dMw, dMb, kctx, lay.tcn[1].weight, lay.residual[1].weight, n_pos)
                return self._fold_tail(Hb, Zb, dX, dgam[0], dgam[1], dbeta) + [dG]
        _, Hb, Zb, X, xbar, XX, mean, istd, a, gamma, n_pos = ctx
        with torch.no_grad():
            dM = torch.empty(K, Co, TV, device=G.device, dtype=torch.float32)
            dM[:K - 1] = dMw.t().view(K - 1, Co, TV)
            dM[K - 1] = dMb.view(Co, TV)
            S1 = dM[K - 1].sum(-1).double()                                   # d shift: both BatchNorms' d beta
            da = (dM[None] * X).sum((1, 3)).double() - mean * S1              # M = a X (+ shift = beta - a mean on the constant image)
            dgamma = da * istd
            dvar = -0.5 * (da * gamma) * istd ** 3                            # a = gamma (var + eps)^(-1/2)
            dmean = -a * S1 - 2.0 * mean * dvar                               # var = E[x^2] - mean^2
            s = G[:, K - 1]
            GX = torch.matmul(G.float(), X.view(2, K, Co * TV)).view(2, K, Co, TV)
            dX = (a.float().view(2, 1, Co, 1) * dM[None] + ((dmean / n_pos).float().view(2, 1, Co, 1) * s.float().view(1, K, 1, 1))
                  + (2.0 * dvar / n_pos).float().view(2, 1, Co, 1) * GX)
            dG = (dvar.view(2, Co, 1, 1) * XX).sum((0, 1)) / n_pos
            dG[:, K - 1] += (dmean.view(2, 1, Co) * xbar).sum((0, 2)) / n_pos
            S1f = S1.float()
            return self._fold_tail(Hb, Zb, dX, dgamma[0].float(), dgamma[1].float(), S1f) + [dG]

    def _fold_tail(self, Hb, Zb, dX, dgamma_t, dgamma_r, dbeta):
        """back through P = Wt gcn(Hb), Q = Wr Hb and Hb = (columns of Wrev, brev): gradients in _params() order"""
        rev, lay = self.rev, self.layer
        K = rev.in_features + 1
        Ci, Co, T, V = lay.in_channels, lay.out_channels, lay.time_dim, lay.joints_dim
        TV = T * V
        tc, rc = lay.tcn[0], lay.residual[0]
        dP, dQ = dX[0], dX[1]
        dWt = torch.einsum('kop,kcp->oc', dP, Zb.view(K, Ci, TV))
        dWr = torch.einsum('kop,kcp->oc', dQ, Hb.view(K, Ci, TV))
        dZb = torch.matmul(tc.weight.view(Co, Ci).t(), dP).view(K, Ci, T, V)
        dHr = torch.matmul(rc.weight.view(Co, Ci).t(), dQ).view(K, Ci, T, V)
        dA, dT, dHb = ops.gcn_bwd_params_dx(Hb, dZb, lay.gcn.A, lay.gcn.T, add=dHr)
        dHb = dHb.view(K, Ci * TV)
        return [dHb[:K - 1].t(), dHb[K - 1], dA, dT, dWt.view_as(tc.weight), dgamma_t, dbeta, dWr.view_as(rc.weight), dgamma_r, dbeta]

    def _bns(self):
        lay = self.layer
        return [lay.tcn[1]] + ([] if isinstance(lay.residual, nn.Identity) else [lay.residual[1]])

    def _graph(self, n_pos: float, device):
        """the fold's forward / backward captured for this position count, or None (eager)"""
        if not GRAPH_FOLD or any(bn.momentum is None for bn in self._bns()):
            return None
        graphs = self.__dict__.setdefault("_graphs", {})
        if n_pos in graphs:
            return graphs[n_pos]
        bufs = [b for bn in self._bns() for b in (bn.running_mean, bn.running_var, bn.num_batches_tracked) if b is not None]
        keep = [b.clone() for b in bufs]                    # warm-up really runs the running-statistics update
        fg = None
        try:
            fg = _FoldGraph(self, n_pos, device)
        except Exception:                                   # capture not possible on this stack: eager from now on
            fg = None
        for b, k in zip(bufs, keep):
            b.copy_(k)
        graphs[n_pos] = fg
        return fg

    def forward(self, z: Tensor) -> Tensor:
        """z [B, Lz] (no autograd) -> U1 [B, C_out, T, V]: the first decoder layer's PRE-activation (apply its PReLU on load)"""
        lay = self.layer
        B, Lz = z.shape
        lay.__dict__.pop("_lowrank_eval", None)               # eval-mode folds of this layer go stale with this step
        lay.__dict__.get("_fold_cache", {}).clear()
        zt = torch.cat([z, torch.ones(B, 1, device=z.device, dtype=z.dtype)], 1)
        # G = sum_n zt zt^T: 256-row pieces on the fp32 MFMA GEMM, pieces summed in fp64 in a fixed order (ONE [K, B] x [B, K] fp64
        # library product is a single-workgroup kernel: 224 us at B = 4096)
        G32 = ops.gemm_rows_outer(zt, zt, torch.empty(Lz + 1, Lz + 1, device=z.device, dtype=torch.float32))
        n_pos = float(B * lay.time_dim * lay.joints_dim)
        fg = self._graph(n_pos, z.device)
        if fg is not None:
            with torch.no_grad():
                fg.G.copy_(G32)
            fg.fwd.replay()
            G, Mw, Mb = fg.G, fg.Mw, fg.Mb
        elif EXPLICIT and not isinstance(lay.residual, nn.Identity):
            G = G32.double()
            Mw, Mb, ectx = self._fold_explicit(G, n_pos)
            fg = ("explicit", ectx)
        else:
            G = G32.double().requires_grad_(True)
            with torch.enable_grad():
                Mw, Mb = self._fold(G, n_pos, update_running=True)
        Mwd, Mbd = Mw.detach(), Mb.detach()
        U1 = ops.rev_btlnk_fwd(z, Mwd, Mbd)
        self._saved = (z, zt, G, Mw, Mb, Mwd, fg)
        return U1.view(B, lay.out_channels, lay.time_dim, lay.joints_dim)

    def _params(self):
        lay, rp, lp = self.layer, self.rev_prefix, self.layer_prefix
        named = [(rp + "weight", self.rev.weight), (rp + "bias", self.rev.bias), (lp + "gcn.A", lay.gcn.A), (lp + "gcn.T", lay.gcn.T),
                 (lp + "tcn.0.weight", lay.tcn[0].weight), (lp + "tcn.1.weight", lay.tcn[1].weight), (lp + "tcn.1.bias", lay.tcn[1].bias)]
        if not isinstance(lay.residual, nn.Identity):
            named += [(lp + "residual.0.weight", lay.residual[0].weight), (lp + "residual.1.weight", lay.residual[1].weight),
                      (lp + "residual.1.bias", lay.residual[1].bias)]
        # (conv biases in front of a train-mode BatchNorm: gradient exactly 0, never written -- as everywhere on the flat path)
        return named

    def backward(self, dU1: Tensor, dz: Optional[Tensor] = None) -> Tensor:
        """dU1: gradient w.r.t. U1 (the next layer's backward has already applied this layer's PReLU'); fills the flat gradient views
        of rev_btlnk and of the layer, returns dz [B, Lz] (added to `dz` when one is given)."""
        z, zt, G, Mw, Mb, Mwd, fg = self._saved
        self._saved = None
        B = z.shape[0]
        named = self._params()
        explicit = isinstance(fg, tuple)
        if fg is not None and not explicit:
            dMw, dMb = fg.dMw, fg.dMb
        else:
            dMw, dMb = torch.empty_like(Mwd), torch.empty(Mwd.shape[0], device=z.device, dtype=torch.float32)
        dz = ops.rev_btlnk_bwd(dU1.reshape(B, -1), z, Mwd, dMw, dMb, dz=dz)
        if explicit:
            grads = self._fold_explicit_bwd(G, fg[1], dMw, dMb)
        elif fg is not None:
            fg.bwd.replay()
            grads = fg.grads
        else:
            grads = torch.autograd.grad([Mw, Mb], [p for _, p in named] + [G], [dMw, dMb], allow_unused=True)
        for (n, _), g in zip(named, grads[:-1]):
            if g is None:
                self.gv[n].zero_()
            else:
                self.gv[n].copy_(g.view_as(self.gv[n]))
        dG = grads[-1]
        if dG is not None:                                                 # the statistics' dependence on the latents: G = sum zt zt^T
            dz.add_((zt @ (dG + dG.t()).float())[:, :z.shape[1]])
        return dz


def fold_eval(rev: nn.Linear, layer):
    """Eval mode (BatchNorm from the running statistics): the K folded images of rev_btlnk + `layer` as the rev_btlnk kernel's
    (weight [C_out T V, Lz], bias [C_out T V]) -- no statistics, no autograd; ~30 small launches on [K, C, T V] tensors."""
    with torch.no_grad():
        Lz = rev.in_features
        K = Lz + 1
        Ci, Co, T, V = layer.in_channels, layer.out_channels, layer.time_dim, layer.joints_dim
        TV = T * V
        Hb = torch.cat([rev.weight.t(), rev.bias[None]], 0).view(K, Ci, T, V)
        Zb = torch.einsum('kctv,tvw->kctw', torch.einsum('kctv,vtq->kcqv', Hb, layer.gcn.T), layer.gcn.A)

        def branch(conv, bn, X):
            a = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            out = a.view(1, Co, 1) * torch.matmul(conv.weight.view(Co, Ci), X.reshape(K, Ci, TV))
            shift = bn.bias - a * bn.running_mean + (a * conv.bias if conv.bias is not None else 0.0)
            return out, shift

        M, shift = branch(layer.tcn[0], layer.tcn[1], Zb)
        if isinstance(layer.residual, nn.Identity):
            M = M + Hb.reshape(K, Ci, TV)
        else:
            Mr, shift_r = branch(layer.residual[0], layer.residual[1], Hb)
            M, shift = M + Mr, shift + shift_r
        Mb = (M[K - 1] + shift.view(Co, 1)).reshape(-1).contiguous()
        Mw = M[:K - 1].reshape(K - 1, Co * TV).t().contiguous()
    return Mw, Mb


def eval_supported(rev, layer) -> bool:
    if MODE == 'never' or not LowRankFirstLayer.supports(rev, layer) or not (MODE == 'always' or layer.is_wide):
        return False
    bns = [layer.tcn[1]] + ([] if isinstance(layer.residual, nn.Identity) else [layer.residual[1]])
    return all(bn.running_mean is not None and bn.running_var is not None for bn in bns)


class _FoldGraph:
    """LowRankFirstLayer._fold and its autograd backward for ONE position count as two hipGraphs over static buffers: G in, the
    folded images out; their gradients in, the parameters' and G's gradients out."""

    def __init__(self, owner: LowRankFirstLayer, n_pos: float, device) -> None:
        K = owner.rev.in_features + 1
        params = [p for _, p in owner._params()]
        self.G = torch.eye(K, device=device, dtype=torch.float64).mul_(n_pos).requires_grad_(True)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm-up outside capture: allocator, library GEMM handles
            for _ in range(2):
                with torch.enable_grad():
                    Mw, Mb = owner._fold(self.G, n_pos, True)
                torch.autograd.grad([Mw, Mb], params + [self.G], [torch.zeros_like(Mw), torch.zeros_like(Mb)], allow_unused=True)
        torch.cuda.current_stream().wait_stream(side)
        self.fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd):
            with torch.enable_grad():
                self.Mw, self.Mb = owner._fold(self.G, n_pos, True)
        self.dMw, self.dMb = torch.zeros_like(self.Mw), torch.zeros_like(self.Mb)
        self.bwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.bwd, pool=self.fwd.pool()):
            self.grads = torch.autograd.grad([self.Mw, self.Mb], params + [self.G], [self.dMw, self.dMb], allow_unused=True)
