"""Explicit (autograd-free) train step of STSE on the HIP path: forward, one-class head, backward,
regulariser, Adam -- the reference's `training_step` + optimizer step
(models/euclidean_encoder_dynamicCenter.py:105-122, models/hyperbolic_encoder.py:137-172, Adam at :196)
as a fixed sequence of C-ABI calls on flat parameter / gradient buffers.

Data parallelism (reference: Lightning DDPStrategy, train_COSKAD.py:78): one process per GPU, clips
sharded over ranks, ONE all-reduce of the flat fp32 gradient buffer per step over RCCL, centre statistics
all-reduced when the centre is refreshed.  BatchNorm statistics stay per rank (reference semantics: no
sync_batchnorm).
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import engine, ops, parallel
from .models.graph_layers.stsgcn import layer_tensors

Tensor = torch.Tensor


class FlatParams:
    """Re-home a module's parameters as views of one flat fp32 buffer (+ a flat gradient buffer),
    in named_parameters() order.  state_dict keys and shapes are untouched."""

    def __init__(self, module: torch.nn.Module) -> None:
        named = [(n, p) for n, p in module.named_parameters()]
        self.names = [n for n, _ in named]
        # every tensor starts on a 16-byte boundary (the layout-specialised kernels load weights as float4); the padding floats
        # stay 0 in the parameter, gradient and Adam buffers
        al = lambda k: (k + 3) // 4 * 4
        total = sum(al(p.numel()) for _, p in named)
        dev = named[0][1].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(total, device=dev, dtype=torch.float32)
        # regulariser mask: calc_reg_loss skips tensors whose NAME contains 'bias' (model_utils.py:92)
        self.reg_mask = torch.zeros(total, device=dev, dtype=torch.float32)
        self.views: Dict[str, Tensor] = {}
        self.gviews: Dict[str, Tensor] = {}
        self.n_reg_tensors = 0
        self.offsets: Dict[str, int] = {}
        off = 0
        for n, p in named:
            k = p.numel()
            self.offsets[n] = off
            self.flat[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.flat[off:off + k].view(p.shape)
            p.grad = self.grad[off:off + k].view(p.shape)
            self.views[n], self.gviews[n] = p.data, p.grad
            if 'bias' not in n:
                self.reg_mask[off:off + k] = 1.0
                self.n_reg_tensors += 1
            off += al(k)


def inv_cov_from_moments(gram: Tensor, acc: Tensor, mu: Tensor, L: int) -> Tensor:
    """inverse of sum (z-mu)(z-mu)^T / (n-1), written with gram = sum z z^T, s = acc[1..L] = sum z, n = acc[17]."""
    s, n = acc[1:1 + L].double(), acc[17].double()
    m = mu.double()
    S = gram.double() - torch.outer(m, s) - torch.outer(s, m) + n * torch.outer(m, m)
    return torch.inverse((S / (n - 1)).float())


class STSETrainStep:
    """One-class training of an STSE (`linear` projector, or `mlp` within the HIP kernels' widths) without autograd.

    head: 'euclidean' -> F.mse_loss(z, c);  'poincare' -> dist(c, project(expmap0(z))).mean().
    """

    def __init__(self, model, lr: float = 1e-4, alpha: float = 1e-6, head: str = 'euclidean',
                 betas=(0.9, 0.999), eps: float = 1e-8, process_group=None, use_graph: bool = False,
                 side_stream: bool = False, sync_bn: bool = False) -> None:
        from .models.sts.ae import STSE
        from .models.common.components import MLP
        self.mlp = isinstance(model.btlnk, MLP)
        if not isinstance(model, STSE) or not (isinstance(model.btlnk, torch.nn.Linear) or (self.mlp and model.btlnk.hip_ok)):
            raise TypeError("STSETrainStep drives an STSE with projector='linear' or an 'mlp' within the kernels' widths")
        self.model, self.head, self.alpha = model, head, float(alpha)
        self.beta1, self.beta2, self.eps = float(betas[0]), float(betas[1]), float(eps)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.fp = FlatParams(model)
        dev = self.fp.flat.device
        self.m = torch.zeros_like(self.fp.flat)
        self.v = torch.zeros_like(self.fp.flat)
        self.hyper = torch.tensor([lr, 1.0, 1.0, 0.0], device=dev, dtype=torch.float32)
        self.lr = float(lr)
        # an encoder with layers beyond the LDS tile kernels (the wide C = 2 -> 256 stack, dropout) runs as a _FlatStack: tile runs
        # through engine.chain_*, wide layers through their explicit forward / backward -- same flat buffers, same fused Adam
        # (so does an encoder with a layer the commuted kernels take -- 32 -> 16 on the 25-joint layout -- unless the step is asked for
        # something only the plain chain does: hipGraph capture, the side stream, SyncBN)
        self.wide = any(l.is_wide for l in model.encoder.model)
        if not self.wide and not (use_graph or side_stream or sync_bn):
            self.wide = any(_is_commute(l) for l in model.encoder.model)
        self.layers = [] if self.wide else [layer_tensors(l) for l in model.encoder.model]
        self.stack = _FlatStack(list(model.encoder.model), self.fp, "encoder.model.") if self.wide else None
        self.ws = engine.Workspace()
        self.center_acc = torch.zeros(ops.HEAD_SLOTS, device=dev, dtype=torch.float32)
        L = model.latent_dim
        self.gram_acc = torch.zeros(L, L, device=dev, dtype=torch.float32) if head == 'mahalanobis' else None
        self.reg_scale = 0.5 / self.fp.n_reg_tensors          # calc_reg_loss value = reg_scale * sum p^2
        self.reg_coef = self.alpha * 2.0 * self.reg_scale      # its gradient coefficient, times alpha
        # gradient views per layer, in the kernels' vocabulary
        self.grads: List[Dict[str, Tensor]] = []
        for i, L in enumerate(self.layers):
            pre = f"encoder.model.{i}."
            g = {"A": pre + "gcn.A", "T": pre + "gcn.T", "Wt": pre + "tcn.0.weight", "bt": pre + "tcn.0.bias",
                 "gt": pre + "tcn.1.weight", "bet": pre + "tcn.1.bias", "Wr": pre + "residual.0.weight",
                 "br": pre + "residual.0.bias", "gr": pre + "residual.1.weight", "ber": pre + "residual.1.bias",
                 "slope": pre + "prelu.weight"}
            self.grads.append({k: self.fp.gviews[n] for k, n in g.items() if n in self.fp.gviews})
        # optional: dA / dT on a second stream beside the next layer's reductions.  Measured SLOWER on MI355X (2.43 vs
        # 2.28 ms/step: the two LDS-heavy persistent kernels halve each other's occupancy), so it is off by default.
        self.side = engine.SideStream() if side_stream else None
        # optional SyncBN of the encoder's BatchNorm2d layers (the reference's DDP keeps per-rank statistics): every
        # BatchNorm boundary of the forward and the backward adds the other ranks' fp64 sums (engine.chain_forward / _backward)
        self.sync_group = None
        if self.wide and (side_stream or use_graph):
            raise ValueError("an encoder with wide layers runs on the main stream, outside hipGraph capture")
        if sync_bn and self.world > 1:
            if self.mlp or side_stream or use_graph or self.wide:
                raise ValueError("sync_bn: encoder BatchNorm only (STS-GCN encoder within the tile kernels, linear projector), on the "
                                 "main stream, outside hipGraph capture")
            self.sync_group = process_group if process_group is not None else dist.group.WORLD
        # gradient buckets for the data-parallel all-reduce: [encoder | bottleneck]; the bottleneck parameters are the
        # tail of the flat buffer (named_parameters order) and their gradients are final before the encoder backward
        names = self.fp.names
        first_tail = next((i for i, n in enumerate(names) if n.startswith("btlnk.")), None)
        self.tail_off = None
        if first_tail is not None and all(n.startswith("btlnk.") for n in names[first_tail:]):
            self.tail_off = self.fp.offsets[names[first_tail]]
        if use_graph and any(isinstance(b, torch.nn.modules.batchnorm._BatchNorm) and b.momentum is None for b in model.modules()):
            raise ValueError("use_graph: BatchNorm with momentum=None changes its averaging factor every step (a launch argument here); "
                             "capture needs a fixed momentum")
        self.use_graph = use_graph
        self._graph = None
        self._x_static: Optional[Tensor] = None
        self._stats_static: Optional[Tensor] = None
        self.steps = 0

    def set_lr(self, lr: float) -> None:
        self.hyper[0] = lr
        self.lr = float(lr)

    def _adam(self) -> None:
        """torch.optim.Adam step on the flat buffers with alpha * calc_reg_loss' gradient and the 1 / world of the gradient
        all-reduce folded in.  Outside hipGraph capture lr and the running products beta^t come from the host (one launch);
        a captured step keeps them in device memory (`hyper`: a one-thread launch advances beta^t in front of the update)."""
        if self.use_graph:
            ops.adam_dev(self.fp.flat, self.fp.grad, self.m, self.v, self.fp.reg_mask, self.hyper, self.beta1,
                         self.beta2, self.eps, gscale=1.0 / self.world, reg_coef=self.reg_coef)
        else:
            import numpy as np
            b1p, b2p = getattr(self, "_bpow", (np.float32(1.0), np.float32(1.0)))
            self._bpow = (np.float32(b1p * np.float32(self.beta1)), np.float32(b2p * np.float32(self.beta2)))   # fp32, as the device tick
            ops.adam_pow(self.fp.flat, self.fp.grad, self.m, self.v, self.fp.reg_mask, self.lr, self.beta1, self.beta2, self.eps,
                         float(self._bpow[0]), float(self._bpow[1]), gscale=1.0 / self.world, reg_coef=self.reg_coef)

    # -- the step ---------------------------------------------------------------------------
    def _body(self, x: Tensor) -> Tensor:
        m = self.model
        B = x.shape[0]
        if self.stack is None:
            U, ctx = engine.chain_forward(x, self.layers, True, self.ws, want_ctx=True, sync=self.sync_group)
            slope, top_layers, last_slope_grad = self.layers[-1].slope, self.layers, self.grads[-1]["slope"]
        else:
            U, slope, saved_stack = self.stack.forward(x, self.ws)      # slope None: the stack ended in a wide layer (activated output)
            ctx, top_layers = self.stack.top(saved_stack)
            top_layers = top_layers or []
            last_slope_grad = self.stack.last_slope_grad
        gv = self.fp.gviews
        if self.mlp:
            # mlp projector (components.py:209-226): wide Linear on the bottleneck kernel (PReLU on load), then every
            # [BatchNorm1d, ReLU, Linear] block on csrc/mlp_head.hip; parameters and gradients stay in the flat buffers
            first, wname = m.btlnk.net[0], "btlnk.net.0."
            W, b = first.weight, first.bias
            y = ops.btlnk_fwd(U, W, b, slope, ws=self.ws)
            saved = []
            if B == 1:                     # nn.BatchNorm1d's own check in training mode
                raise ValueError("Expected more than 1 value per channel when training (BatchNorm1d of the mlp projector)")
            for i, (bn, lin) in enumerate(m.btlnk.blocks()):
                z, stat = ops.mlp_head_fwd(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                           lin.weight, lin.bias, True, momentum=ops.bn_momentum(bn), eps=bn.eps)
                saved.append((y, stat, bn, lin, f"btlnk.net.{3 * i + 1}.", f"btlnk.net.{3 * i + 3}."))
                y = z
            z = y
        else:
            wname = "btlnk."
            W, b = m.btlnk.weight, m.btlnk.bias
            z = ops.btlnk_fwd(U, W, b, slope, ws=self.ws)
        if self.head == 'euclidean':
            stats, dz, _ = ops.mse_head(z, m.c, acc=self.center_acc)
        elif self.head == 'poincare':
            stats, dz, _, _ = ops.poincare_head(z, m.c, acc=self.center_acc)
        elif self.head == 'mahalanobis':
            stats, dz, _ = ops.mahalanobis_head(z, m.c, m.inv_cov_matrix, acc=self.center_acc, gram=self.gram_acc)
        else:
            raise ValueError(f"unknown head {self.head}")
        if self.mlp:
            for y_in, stat, bn, lin, bname, lname in reversed(saved):
                g = {"gamma": gv[bname + "weight"], "beta": gv[bname + "bias"], "W2": gv[lname + "weight"], "b2": gv.get(lname + "bias")}
                dz = ops.mlp_head_bwd(y_in, stat, bn.weight, bn.bias, lin.weight, dz, g, True)
        dU, top_stats = engine.btlnk_backward(ctx if self.side is None else None, top_layers, U, W, dz, slope, gv[wname + "weight"],
                                              gv.get(wname + "bias"), last_slope_grad, self.ws)
        work = None
        if self.world > 1 and self.tail_off is not None:
            # bucket 1 (87 % of the bytes: the bottleneck weight) is complete now: its all-reduce runs on the collective
            # stream while the encoder backward proceeds (SUM; the 1/W is folded into Adam)
            work = dist.all_reduce(self.fp.grad[self.tail_off:], group=self.pg, async_op=True)
        if self.stack is None:
            engine.chain_backward(ctx, self.layers, dU, self.ws, self.grads, need_dx=False, side=self.side, stats_in=top_stats)
        else:
            self.stack.backward(saved_stack, dU, self.ws, need_dx=False, top_stats=top_stats)
        if self.world > 1:
            head = self.fp.grad if work is None else self.fp.grad[:self.tail_off]
            dist.all_reduce(head, group=self.pg)           # bucket 2: the encoder's gradients (0.12 MB)
            if work is not None:
                work.wait()
        self._adam()
        return stats

    def step(self, x: Tensor) -> Tensor:
        """One optimisation step on clips x [B,C,T,V]; returns the head's stats block (stats[0] = loss)."""
        self.steps += 1
        x = x.contiguous()
        if not self.use_graph:
            return self._body(x)
        if self._graph is None or self._x_static.shape != x.shape:
            self._x_static = x.clone()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):          # warm-up outside capture (allocator, lazy module load)
                self._body(self._x_static)
            torch.cuda.current_stream().wait_stream(s)
            self.steps += 1
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._stats_static = self._body(self._x_static)
        self._x_static.copy_(x)
        self._graph.replay()
        return self._stats_static

    def refresh_inv_cov(self, mu: Tensor, reset: bool = True) -> Tensor:
        """inv_cov_matrix <- inverse(sum_n (z_n - mu)(z_n - mu)^T / (n - 1)) over the latents seen since the last
        reset (staticCenter.py:40-46,133-142), from the accumulated second moments (all-reduced over ranks)."""
        parallel.allreduce_sum_(self.gram_acc, self.pg)
        acc = self.center_acc.clone()
        parallel.allreduce_sum_(acc, self.pg)
        self.model.inv_cov_matrix.copy_(inv_cov_from_moments(self.gram_acc, acc, mu, self.model.latent_dim))
        if reset:
            self.gram_acc.zero_()
        return self.model.inv_cov_matrix

    def reg_loss(self) -> Tensor:
        """utils/model_utils.py::calc_reg_loss value of the current parameters (1-element tensor)."""
        return ops.sqnorm(self.fp.flat, self.fp.reg_mask, self.reg_scale)

    # -- centre bookkeeping (staticCenter.py:145-155; hyperbolic_encoder.py:175-183) ----------
    def refresh_center(self, eps: float = 1e-3) -> Tensor:
        """c <- statistics accumulated since the last refresh (all-reduced over ranks), then reset them."""
        parallel.allreduce_sum_(self.center_acc, self.pg)
        L = self.model.latent_dim
        # Euclidean and Mahalanobis heads accumulate plain sums (mean centre); only the Poincare head's sums are the
        # gyromidpoint's
        c = ops.center_finalize(self.center_acc, eps, L) if self.head != 'poincare' else ops.midpoint_finalize(self.center_acc, L)
        self.model.c.copy_(c)
        self.center_acc.zero_()
        return self.model.c


def _layer_grad_views(fp: "FlatParams", prefix: str) -> Dict[str, Tensor]:
    """gradient views of one ST_GCNN layer (state_dict prefix `encoder.model.3.`) in the kernels' vocabulary"""
    g = {"A": "gcn.A", "T": "gcn.T", "Wt": "tcn.0.weight", "bt": "tcn.0.bias", "gt": "tcn.1.weight", "bet": "tcn.1.bias",
         "Wr": "residual.0.weight", "br": "residual.0.bias", "gr": "residual.1.weight", "ber": "residual.1.bias",
         "slope": "prelu.weight"}
    return {k: fp.gviews[prefix + n] for k, n in g.items() if prefix + n in fp.gviews}


# layers with <= 4 output channels behind 16 / 32 / 64 input channels (the decoder's last layer) run by commutation: both 1x1
# convolutions first, as one streaming pass over the wide input, then mixing / BatchNorm / PReLU on 2 C_out-channel tensors
# (csrc/last_layer.hip).  Tests flip it to hold the two paths against each other.
NARROW_OUT = True


def _is_narrow(m) -> bool:
    return (NARROW_OUT and m.out_channels <= 4 and not m.is_wide and not isinstance(m.residual, torch.nn.Identity)
            and ops.narrow_conv_ok(m.in_channels, 2 * m.out_channels, m.time_dim * m.joints_dim))


# Layers with fewer output than input channels (32 -> 16 on the 25-joint layout) by commutation as well, on their own kernels
# (csrc/commute_layer.hip): convolutions first, mixing / BatchNorm statistics / both adjoints / dA, dT on 16 channels.
COMMUTE = True
COMMUTE_NEXT = True      # ... and the statistics pass of the layer behind them rides on their last kernel


def _is_commute(m, decoder: bool = False) -> bool:
    """decoder: the layer sits in a decoder stack.  At 17 joints the encoder's chained kernels (apply + next statistics, backward +
    reductions of the layer below) already hold a 32 -> 16 layer on chip and the commuted form costs the chain more than it saves; the
    decoder's 32 -> 16 layer stands alone between the folded first layer and the narrow last one (a statistics pass, an apply, a
    reduction pass and a backward kernel of its own), and there the commuted kernels win."""
    tb, rb = m.tcn[1], (m.residual[1] if not isinstance(m.residual, torch.nn.Identity) else None)
    if m.joints_dim == 17 and not decoder:
        return False
    return (COMMUTE and not m.is_wide and rb is not None and ops.commute_ok(m.time_dim, m.joints_dim, m.in_channels, m.out_channels)
            and all(b.momentum is not None and b.affine and b.track_running_stats for b in (tb, rb)) and tb.eps == rb.eps
            and tb.momentum == rb.momentum)


def _virtual_narrow_layer(mod, fp: "FlatParams", prefix: str):
    """The (2 C_out -> C_out) layer that remains behind the commuted convolutions: input [Y; R] = [Wt X; Wr X], `tcn` convolution =
    selector of the Y channels (+ the real bias), `residual` convolution = selector of the R channels (+ the real bias); mixing
    parameters, BatchNorms and PReLU are the real layer's.  -> (LayerTensors, gradient views: the selectors' go to scratch)."""
    from .models.graph_layers.stsgcn import check_bn
    Co = mod.out_channels
    J = 2 * Co
    tc, tb, rc, rb = mod.tcn[0], mod.tcn[1], mod.residual[0], mod.residual[1]
    check_bn(tb, rb)
    dev = mod.gcn.A.device
    sel_t = torch.zeros(Co, J, 1, 1, device=dev)
    sel_r = torch.zeros(Co, J, 1, 1, device=dev)
    for o in range(Co):
        sel_t[o, o] = 1.0
        sel_r[o, Co + o] = 1.0
    lt = engine.LayerTensors(A=mod.gcn.A, T=mod.gcn.T, Wt=sel_t, bt=tc.bias, gt=tb.weight, bet=tb.bias, rm_t=tb.running_mean,
                             rv_t=tb.running_var, nbt_t=tb.num_batches_tracked, Wr=sel_r, br=rc.bias, gr=rb.weight, ber=rb.bias,
                             rm_r=rb.running_mean, rv_r=rb.running_var, nbt_r=rb.num_batches_tracked, slope=mod.prelu.weight,
                             momentum=tb.momentum if tb.momentum is not None else 0.1, bn=tb if tb.momentum is None else None, cache={})
    gv = fp.gviews
    g = {"A": gv[prefix + "gcn.A"], "T": gv[prefix + "gcn.T"], "Wt": torch.empty_like(sel_t), "gt": gv[prefix + "tcn.1.weight"],
         "bet": gv[prefix + "tcn.1.bias"], "Wr": torch.empty_like(sel_r), "gr": gv[prefix + "residual.1.weight"],
         "ber": gv[prefix + "residual.1.bias"], "slope": gv[prefix + "prelu.weight"]}
    if tc.bias is not None:
        g["bt"] = gv[prefix + "tcn.0.bias"]
    if rc.bias is not None:
        g["br"] = gv[prefix + "residual.0.bias"]
    return lt, g


class _FlatStack:
    """A stack of ST_GCNN layers (Encoder / Decoder `model`, components.py:70-105,143-179) on flat parameter / gradient
    buffers: runs of layers the LDS tile kernels take go through engine.chain_forward / chain_backward (no autograd);
    a layer beyond them (`is_wide`: 64 input channels on the 25-joint layout, > 64 channels, dropout) runs its composed
    HIP path (stsgcn.wide_forward / wide_backward: explicit forward and backward, no autograd), gradients written to the flat
    buffer's views."""

    def __init__(self, modules, fp: "FlatParams", prefix: str, first: int = 0) -> None:
        """modules: the layers first, first + 1, .. of the nn.Sequential whose parameters are named `{prefix}{index}.`"""
        self.segs = []                 # ('tile', [LayerTensors], [grad dicts]) | ('wide', module, names)
        dec = prefix.startswith("decoder")
        i, n = 0, len(modules)
        while i < n:
            if modules[i].is_wide:
                self.segs.append(('wide', modules[i], f"{prefix}{first + i}."))
                i += 1
            elif _is_narrow(modules[i]):
                pre = f"{prefix}{first + i}."
                self.segs.append(('narrow', modules[i], pre) + _virtual_narrow_layer(modules[i], fp, pre))
                i += 1
            elif _is_commute(modules[i], dec):
                self.segs.append(('commute', modules[i], f"{prefix}{first + i}."))
                i += 1
            else:
                j = i
                while j < n and not modules[j].is_wide and not _is_narrow(modules[j]) and not _is_commute(modules[j], dec):
                    j += 1
                self.segs.append(('tile', [layer_tensors(m) for m in modules[i:j]],
                                  [_layer_grad_views(fp, f"{prefix}{first + k}.") for k in range(i, j)]))
                i = j
        self.fp = fp
        self.last_slope_grad = (fp.gviews[f"{prefix}{first + n - 1}.prelu.weight"]
                                if self.segs[-1][0] in ('tile', 'narrow', 'commute') else None)

    def forward(self, x: Tensor, ws: engine.Workspace, in_slope: Optional[Tensor] = None):
        """x: the stack's input, activated (in_slope None) or a pre-activation whose PReLU weight is `in_slope`
        -> (h, slope, saved): apply PReLU(slope) to h for the stack's output (slope None: done)."""
        h, slope, saved = x, in_slope, []
        pend = None                    # the next tile run's first-layer statistics, when the commuted layer in front of it formed them
        for k, seg in enumerate(self.segs):
            if seg[0] == 'tile':
                u, ctx = engine.chain_forward(h, seg[1], True, ws, in_slope=slope, want_ctx=True, pending0=pend)
                pend = None
                saved.append(ctx)
                h, slope = u, seg[1][-1].slope
            elif seg[0] == 'narrow':
                mod, virt = seg[1], seg[3]
                Co, Ci = mod.out_channels, mod.in_channels
                W4 = torch.cat([mod.tcn[0].weight.view(Co, Ci), mod.residual[0].weight.view(Co, Ci)], 0)
                mod.__dict__.get("_fold_cache", {}).clear()                # (the real layer's eval-mode fold goes stale with this step)
                YR = ops.narrow_conv_fwd(h, slope, W4)                     # [Wt X; Wr X] with X = PReLU(h)
                u, ctx = engine.chain_forward(YR, [virt], True, ws, want_ctx=True)
                saved.append((ctx, h, slope, W4))
                h, slope = u, virt.slope
            elif seg[0] == 'commute':
                mod = seg[1]
                Co, Ci = mod.out_channels, mod.in_channels
                tc, tb, rc, rb = mod.tcn[0], mod.tcn[1], mod.residual[0], mod.residual[1]
                mod.__dict__.get("_fold_cache", {}).clear()
                # a tile run behind this layer: its first layer's statistics pass rides on this layer's last kernel
                nxt = self.segs[k + 1][1][0] if (COMMUTE_NEXT and k + 1 < len(self.segs) and self.segs[k + 1][0] == 'tile') else None
                # (25 joints: at 17 the tile kernels' own 16-channel statistics pass is the faster one -- 2.746 vs 2.754 ms on the VAE step)
                if nxt is not None and (nxt.Ci != Co or nxt.rm_t is None or not engine.STORE_Z or mod.joints_dim != 25):
                    nxt = None
                u, sv, pend = ops.commute_fwd(h, slope, tc.weight, rc.weight, mod.gcn.A, mod.gcn.T, tb.weight, tb.bias, rb.weight, rb.bias,
                                              tc.bias, rc.bias, tb.running_mean, tb.running_var, rb.running_mean, rb.running_var,
                                              tb.num_batches_tracked, rb.num_batches_tracked, tb.momentum, tb.eps,
                                              next_layer=(nxt.A, nxt.T) if nxt is not None else None, slope_out=mod.prelu.weight)
                saved.append(sv)
                h, slope = u, mod.prelu.weight
            else:
                from .models.graph_layers.stsgcn import wide_forward
                pre_u, pre_slope = (h, slope) if slope is not None else (None, None)
                xin = ops.prelu_fwd(h, slope) if slope is not None else h
                out, wsaved, wmeta = wide_forward(xin, *seg[1].wide_args())
                saved.append((wsaved, wmeta, pre_u, pre_slope))
                h, slope = out, None
        return h, slope, saved

    def top(self, saved):
        """(ChainCtx, layers) of the last segment when it is a tile run (engine.btlnk_backward), else (None, None)"""
        return (saved[-1], self.segs[-1][1]) if self.segs[-1][0] == 'tile' else (None, None)

    def backward(self, saved, d_last: Tensor, ws: engine.Workspace, need_dx: bool, top_stats=None,
                 in_slope_grad: Optional[Tensor] = None) -> Optional[Tensor]:
        """d_last: gradient w.r.t. the last segment's output (pre-activation U of a tile run -- the caller owns its slope
        gradient -- or the activated output of a wide layer); top_stats: the last tile run's top-layer batch reductions when
        the producer of d_last formed them (engine.btlnk_backward); in_slope_grad: where the gradient of forward's `in_slope`
        goes (the returned gradient is then w.r.t. the PRE-activation input)."""
        d = d_last

        def below_slope_grad(k):
            """where the PReLU-weight gradient of segment k's INPUT goes: the segment below's last slope, or the caller's"""
            if k == 0:
                return in_slope_grad
            below = self.segs[k - 1]
            if below[0] == 'tile':
                return below[2][-1]["slope"]
            if below[0] == 'narrow':
                return below[4]["slope"]
            if below[0] == 'commute':
                return self.fp.gviews[below[2] + "prelu.weight"]
            return None

        chained = None                 # batch reductions of the tile run below, formed by the commuted layer's backward kernel
        for k in range(len(self.segs) - 1, -1, -1):
            seg, sv = self.segs[k], saved[k]
            first = k == 0
            if seg[0] == 'tile':
                d = engine.chain_backward(sv, seg[1], d, ws, seg[2], need_dx=need_dx or not first,
                                          stats_in=top_stats if k == len(self.segs) - 1 else chained,
                                          in_slope_grad=below_slope_grad(k) if (first or self.segs[k - 1][0] == 'commute') else None)
            elif seg[0] == 'commute':
                prefix, gv = seg[2], self.fp.gviews
                into = {"A": gv[prefix + "gcn.A"], "T": gv[prefix + "gcn.T"], "Wt": gv[prefix + "tcn.0.weight"],
                        "Wr": gv[prefix + "residual.0.weight"], "gt": gv[prefix + "tcn.1.weight"], "bet": gv[prefix + "tcn.1.bias"],
                        "gr": gv[prefix + "residual.1.weight"], "ber": gv[prefix + "residual.1.bias"]}
                if sv[1] is not None:
                    into["in_slope"] = below_slope_grad(k)
                    if into["in_slope"] is None:
                        raise RuntimeError("commuted layer: nowhere to put the gradient of its input's PReLU weight")
                # a tile run in front that ends in a 2-channel layer fed by the network input: its batch reductions ride on this kernel
                below = None
                if COMMUTE_NEXT and k > 0 and self.segs[k - 1][0] == 'tile':
                    bl, bctx = self.segs[k - 1][1][-1], saved[k - 1]
                    if (bl.Ci == 2 and bl.Co == 32 and bl.Wr is not None and len(self.segs[k - 1][1]) == 1 and bctx.in_slope is None
                            and bctx.zs and bctx.zs[-1] is not None and bctx.sync is None):
                        below = (bctx.inputs[-1], bctx.zs[-1])
                d, chained = ops.commute_bwd(sv, d.contiguous(), into, below=below)
                if not (need_dx or not first):
                    d = None
            elif seg[0] == 'narrow':
                mod, prefix, virt, vg = seg[1], seg[2], seg[3], seg[4]
                ctx, pre_u, pre_slope, W4 = sv
                Co, Ci = mod.out_channels, mod.in_channels
                dYR = engine.chain_backward(ctx, [virt], d, ws, [vg], need_dx=True)
                d, sums = ops.narrow_conv_bwd(pre_u, pre_slope, W4, dYR)
                gv = self.fp.gviews
                gv[prefix + "tcn.0.weight"].copy_(sums[:Co * Ci].view(Co, Ci, 1, 1))
                gv[prefix + "residual.0.weight"].copy_(sums[Co * Ci:2 * Co * Ci].view(Co, Ci, 1, 1))
                if pre_slope is not None:
                    dslope = (self.segs[k - 1][2][-1]["slope"] if self.segs[k - 1][0] == 'tile' else self.segs[k - 1][4]["slope"]) \
                        if k > 0 else in_slope_grad
                    if dslope is not None:
                        dslope.copy_(sums[2 * Co * Ci:])
                if not (need_dx or not first):
                    d = None
            else:
                from .models.graph_layers.stsgcn import wide_backward
                wsaved, wmeta, pre_u, pre_slope = sv
                prefix = seg[2]
                want_x = need_dx or not first
                names = ("gcn.A", "gcn.T", "tcn.0.weight", "tcn.0.bias", "tcn.1.weight", "tcn.1.bias", "residual.0.weight",
                         "residual.0.bias", "residual.1.weight", "residual.1.bias", "prelu.weight")
                # the kernels write straight into the flat gradient buffer's views; conv biases in front of a train-mode BatchNorm keep
                # the exact 0 the buffer was created with (nothing ever writes them)
                into = {n: self.fp.gviews[prefix + n] for n in names if prefix + n in self.fp.gviews}
                res = wide_backward(wsaved, wmeta, d, need_dx=want_x, into=into)
                for n, g in zip(names, res[1:]):
                    if g is not None and g.data_ptr() != into[n].data_ptr():      # (eval-statistics layers: bias sums)
                        into[n].copy_(g.view_as(into[n]))
                d = res[0] if want_x else None
                if d is not None and pre_u is not None:
                    # the wide layer consumed PReLU(pre_u): back through it, into the producing tile run's last slope
                    dslope = self.segs[k - 1][2][-1]["slope"] if k > 0 else in_slope_grad
                    d = ops.prelu_bwd(pre_u, d.contiguous(), pre_slope, dslope)
        return d


class STSAETrainStep:
    """One optimisation step of the decoder models on flat parameter / gradient buffers with the fused Adam -- the
    reference's training_step + optimizer step of

      mode 'ae'  (models/euclidean_autoencoder.py:106-118): lambda_ * MSE(x_rec, x) + MSE(z, c) + alpha * reg
      mode 'vae' (models/spherical_vae.py:81-107):          phi * MSE(x_rec, x) + alpha * reg + beta * KL(q || p) + gamma * mean(1 / kappa)

    encoder -> bottleneck (one kernel pass; the VAE's mean | concentration heads stacked) -> [VAE: normalise, softplus + 1,
    PowerSpherical rsample, KL: torch ops on [B, latent] tensors under a local autograd graph] -> rev_btlnk (MFMA GEMM) ->
    decoder -> reconstruction head (PReLU of the last layer + MSE + its gradient in one kernel) -> everything backwards
    through the same kernels.  Gradients land in the flat buffer; data-parallel all-reduce and Adam as in STSETrainStep."""

    def __init__(self, model, mode: str = 'ae', lr: float = 1e-4, alpha: float = 0.0, lambda_: float = 0.01, phi: float = 1.0,
                 beta: float = 1.0, gamma: float = 1.0, betas=(0.9, 0.999), eps: float = 1e-8, process_group=None) -> None:
        from .models.sts.ae import STSAE
        from .models.sts.vae import STSVAE
        from .models.common.components import Encoder
        if not isinstance(model, STSAE) or not isinstance(model.encoder, Encoder):
            raise TypeError("STSAETrainStep drives an STSAE / STSVAE with the STS-GCN encoder")
        if mode not in ('ae', 'vae') or (mode == 'vae') != isinstance(model, STSVAE):
            raise ValueError(f"mode {mode!r} does not fit {type(model).__name__}")
        self.model, self.mode = model, mode
        self.alpha, self.lambda_, self.phi, self.beta, self.gamma = float(alpha), float(lambda_), float(phi), float(beta), float(gamma)
        self.beta1, self.beta2, self.eps = float(betas[0]), float(betas[1]), float(eps)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        if not self.supports(model):
            raise TypeError("STSAETrainStep: projector / latent size outside the bottleneck kernels (latent rows <= 16, "
                            "'linear' projector)")
        self.fp = FlatParams(model)
        dev = self.fp.flat.device
        self.m = torch.zeros_like(self.fp.flat)
        self.v = torch.zeros_like(self.fp.flat)
        self.lr = float(lr)
        self.use_graph = False
        self.ws = engine.Workspace()
        self.enc = _FlatStack(list(model.encoder.model), self.fp, "encoder.model.")
        # the decoder's first layer sees a rank-(latent + 1) input (rev_btlnk of the latent): folded into one streaming pass where that
        # layer would otherwise take the composed wide path (coskad_amd/lowrank.py)
        from . import lowrank
        dec_layers = list(model.decoder.model)
        self.lowrank = None
        if (len(dec_layers) > 1 and lowrank.LowRankFirstLayer.supports(model.rev_btlnk, dec_layers[0])
                and (lowrank.MODE == 'always' or (lowrank.MODE == 'wide' and dec_layers[0].is_wide))):
            self.lowrank = lowrank.LowRankFirstLayer(model.rev_btlnk, dec_layers[0], self.fp.gviews)
            self.dec = _FlatStack(dec_layers[1:], self.fp, "decoder.model.", first=1)
        else:
            self.dec = _FlatStack(dec_layers, self.fp, "decoder.model.")
        self.center_acc = torch.zeros(ops.HEAD_SLOTS, device=dev, dtype=torch.float32)
        self.reg_scale = 0.5 / self.fp.n_reg_tensors
        self.reg_coef = self.alpha * 2.0 * self.reg_scale
        self.steps = 0
        self.last = {}

    @staticmethod
    def supports(model) -> bool:
        from .models.sts.vae import STSVAE
        from .models.common.components import MLP
        if isinstance(model, STSVAE):
            if isinstance(model.btlnk, MLP):              # `projector: 'mlp'` (spherical_vae.yaml:37): MLP, then the heads on its output
                return model.btlnk.hip_ok and model.latent_dim <= 16
            return isinstance(model.btlnk, torch.nn.Identity) and model.latent_dim + model.fc_var.out_features <= 16
        return isinstance(model.btlnk, torch.nn.Linear) and model.latent_dim <= 16

    def set_lr(self, lr: float) -> None:
        self.lr = float(lr)

    _adam = STSETrainStep._adam

    def reg_loss(self) -> Tensor:
        return ops.sqnorm(self.fp.flat, self.fp.reg_mask, self.reg_scale)

    def step(self, x: Tensor) -> Dict[str, Tensor]:
        """-> {'rec': F.mse_loss(x_rec, x), 'head': MSE(z, c) | KL, ('exp': mean(1 / kappa)), 'z': the latents}"""
        self.steps += 1
        m, gv = self.model, self.fp.gviews
        x = x.contiguous()
        B = x.shape[0]
        T, V, hid = m.n_frames, m.n_joints, m.hidden_dimension
        U, slope, enc_saved = self.enc.forward(x, self.ws)
        if slope is None:              # the encoder ended in a wide layer: already activated
            raise NotImplementedError("STSAETrainStep: an encoder ending in a wide layer")
        out: Dict[str, Tensor] = {}
        if self.mode == 'ae':
            W, b, wname = m.btlnk.weight, m.btlnk.bias, "btlnk."
            z = ops.btlnk_fwd(U, W, b, slope, ws=self.ws)
            stats, dz, _ = ops.mse_head(z, m.c, acc=self.center_acc)              # MSE(z, c) and its gradient
            out['head'] = stats[0:1]
            z_dec, graph = z, None
        else:
            from .models.sts.vae import kl_ps_uniform
            from .models.common.components import MLP
            L = m.latent_dim
            mlp = isinstance(m.btlnk, MLP)
            mlp_saved, head_params = None, []
            if mlp:
                # `mlp` projector (vae.py:141-146): wide Linear on the bottleneck kernel, [BatchNorm1d, ReLU, Linear] blocks on
                # csrc/mlp_head.hip; the two small heads act on its [B, latent] output inside the local autograd graph
                if B == 1:
                    raise ValueError("Expected more than 1 value per channel when training (BatchNorm1d of the mlp projector)")
                first = m.btlnk.net[0]
                W, b = first.weight, first.bias
                y = ops.btlnk_fwd(U, W, b, slope, ws=self.ws)
                mlp_saved = []
                for i, (bn, lin) in enumerate(m.btlnk.blocks()):
                    zz, stat = ops.mlp_head_fwd(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                                lin.weight, lin.bias, True, momentum=ops.bn_momentum(bn), eps=bn.eps)
                    mlp_saved.append((y, stat, bn, lin, f"btlnk.net.{3 * i + 1}.", f"btlnk.net.{3 * i + 3}."))
                    y = zz
                Hd = y
                head_params = [m.fc_mean.weight, m.fc_mean.bias, m.fc_var.weight, m.fc_var.bias]
            else:
                W = torch.cat([m.fc_mean.weight, m.fc_var.weight], 0)             # heads stacked: one pass over U
                b = torch.cat([m.fc_mean.bias, m.fc_var.bias], 0)
                Hd = ops.btlnk_fwd(U, W, b, slope, ws=self.ws)
            if m.distribution == 'ps':
                # the PowerSpherical head on csrc/vae_head.hip (normalise, softplus + 1, Householder sample, KL, 1 / kappa: three
                # launches + torch's Beta draw instead of ~80 element-wise launches under autograd); the two small Linears of the
                # `mlp` projector's heads ride on the strided GEMM with the weights stacked
                if mlp:
                    Wc = torch.cat([m.fc_mean.weight, m.fc_var.weight], 0)            # [L + 1, L]
                    bc = torch.cat([m.fc_mean.bias, m.fc_var.bias], 0)
                    H2 = ops.gemm(Hd, Wc.t(), bias=bc, bias_mode=2)
                else:
                    Wc, H2 = None, Hd
                zs, kl_rows, ik_rows, ps_saved = ops.ps_head_forward(H2[:, :L], H2[:, L:L + 1])
                out['head'], out['exp'] = kl_rows.mean().reshape(1), ik_rows.mean().reshape(1)
                z_dec, graph = zs, ('ps', Hd, H2, Wc, ps_saved, mlp_saved)
            else:
                Hd.requires_grad_(True)
                with torch.enable_grad():                                         # [B, latent] tensors: vae.py:79-91,104-118
                    if mlp:
                        Z_mean, Z_var = m._finish_heads(m.fc_mean(Hd), m.fc_var(Hd), None, False)
                    else:
                        Z_mean, Z_var = m._finish_heads(Hd[:, :L], Hd[:, L:], None, False)
                    q, p = m.reparameterize(Z_mean, Z_var)
                    zs = q.rsample()
                    loss_kl = torch.distributions.kl.kl_divergence(q, p).sum(-1).mean()
                    loss_exp = (1 / Z_var).mean()
                    small = self.beta * loss_kl + self.gamma * loss_exp
                out['head'], out['exp'] = loss_kl.detach().reshape(1), loss_exp.detach().reshape(1)
                z_dec, graph = zs.detach().contiguous(), ('autograd', Hd, zs, small, head_params, mlp_saved)
            dz = None
        out['z'] = z_dec
        # rev_btlnk (ae.py:223-227): H = z Wr^T + br on the strided MFMA GEMM, straight into the decoder's [B, hid, T, V] view
        Wr, br = m.rev_btlnk.weight, m.rev_btlnk.bias
        if self.lowrank is not None:
            l0 = m.decoder.model[0]
            Ud, dslope_d, dec_saved = self.dec.forward(self.lowrank.forward(z_dec), self.ws, in_slope=l0.prelu.weight)
        else:
            H = ops.rev_btlnk_fwd(z_dec, Wr, br)
            Ud, dslope_d, dec_saved = self.dec.forward(H.view(B, hid, T, V), self.ws)
        w_rec = self.lambda_ if self.mode == 'ae' else self.phi
        if dslope_d is None:
            raise NotImplementedError("STSAETrainStep: a decoder ending in a wide layer")
        loss_rec, dUd, _ = ops.rec_head(Ud, x, dslope_d, dslope=self.dec.last_slope_grad, upstream=w_rec)
        out['rec'] = loss_rec
        # ---- backward ----------------------------------------------------------------------------------------------------
        if self.lowrank is not None:
            dU1 = self.dec.backward(dec_saved, dUd, self.ws, need_dx=True, in_slope_grad=gv["decoder.model.0.prelu.weight"])
            rev_bwd = lambda dz_: self.lowrank.backward(dU1, dz=dz_)
        else:
            dH = self.dec.backward(dec_saved, dUd, self.ws, need_dx=True).reshape(B, -1)
            # rev_btlnk: dWr = dH^T z, dbr = sum dH, dz (+)= dH Wr: streaming kernels over dH (csrc/rev_btlnk.hip)
            rev_bwd = lambda dz_: ops.rev_btlnk_bwd(dH, z_dec, Wr, gv["rev_btlnk.weight"], gv["rev_btlnk.bias"], dz=dz_)
        if self.mode == 'ae':
            dHd = rev_bwd(dz)                                                     # dz = d MSE(z, c) + dH Wr
        else:
            dz_dec = rev_bwd(None)
            L = m.latent_dim
            if graph[0] == 'ps':
                _, Hd, H2, Wc, ps_saved, mlp_saved = graph
                dH2 = torch.empty_like(H2)
                ops.ps_head_backward(ps_saved, dz_dec, self.beta / B, self.gamma / B, dH2[:, :L], dH2[:, L:L + 1])
                if mlp_saved is not None:
                    # the two heads' Linears (weights stacked): dWc = dH2^T Hd, db = column sums, dHd = dH2 Wc
                    dWc = ops.gemm_rows_outer(dH2, Hd.contiguous(), torch.empty(L + 1, L, device=dH2.device, dtype=torch.float32))
                    dbc = dH2.sum(0)
                    gv["fc_mean.weight"].copy_(dWc[:L]); gv["fc_var.weight"].copy_(dWc[L:])
                    gv["fc_mean.bias"].copy_(dbc[:L]); gv["fc_var.bias"].copy_(dbc[L:])
                    dHd = ops.gemm(dH2, Wc)
                else:
                    dHd = dH2
            else:
                _, Hd, zs, small, head_params, mlp_saved = graph
                res = torch.autograd.grad([zs, small], [Hd] + head_params, [dz_dec, torch.ones_like(small)])
                dHd = res[0].contiguous()
                if mlp_saved is not None:
                    for n, g_ in zip(("fc_mean.weight", "fc_mean.bias", "fc_var.weight", "fc_var.bias"), res[1:]):
                        gv[n].copy_(g_)
            if mlp_saved is not None:
                for y_in, stat, bn, lin, bname, lname in reversed(mlp_saved):
                    g = {"gamma": gv[bname + "weight"], "beta": gv[bname + "bias"], "W2": gv[lname + "weight"], "b2": gv.get(lname + "bias")}
                    dHd = ops.mlp_head_bwd(y_in, stat, bn.weight, bn.bias, lin.weight, dHd, g, True)
        tctx, tlayers = self.enc.top(enc_saved)
        bb = lambda gW_, gb_: engine.btlnk_backward(tctx, tlayers, U, W, dHd, slope, gW_, gb_, self.enc.last_slope_grad, self.ws)
        if self.mode == 'ae':
            dU, top_stats = bb(gv["btlnk.weight"], gv.get("btlnk.bias"))
        elif mlp_saved is not None:          # mlp projector: the wide first Linear of the MLP
            dU, top_stats = bb(gv["btlnk.net.0.weight"], gv.get("btlnk.net.0.bias"))
        else:
            gW = torch.empty_like(W)
            gb = torch.empty_like(b)
            dU, top_stats = bb(gW, gb)
            L = m.latent_dim
            gv["fc_mean.weight"].copy_(gW[:L]); gv["fc_var.weight"].copy_(gW[L:])
            gv["fc_mean.bias"].copy_(gb[:L]); gv["fc_var.bias"].copy_(gb[L:])
        self.enc.backward(enc_saved, dU, self.ws, need_dx=False, top_stats=top_stats)
        if self.world > 1:
            dist.all_reduce(self.fp.grad, group=self.pg)                          # SUM; the 1 / W is folded into Adam
        self._adam()
        self.last = out
        return out


class AutogradTrainStep:
    """Same interface as STSETrainStep for models the flat-buffer path does not take: the plain-GCN encoders, `mlp` projectors and
    latents beyond the bottleneck kernels' widths.  Forward / backward go through the module
    surface (autograd nodes around the HIP kernels, library GEMMs where the module uses them); the one-class head and
    its gradient are the HIP head kernels (`z.backward(dz)`), the regulariser gradient is added to `.grad`, the
    optimiser is torch's Adam (calc_reg_loss / configure_optimizers of the reference wrappers)."""

    def __init__(self, model, lr: float = 1e-4, alpha: float = 1e-6, head: str = 'euclidean', betas=(0.9, 0.999),
                 eps: float = 1e-8, process_group=None) -> None:
        self.model, self.head, self.alpha, self.pg = model, head, float(alpha), process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.params = [(n, p) for n, p in model.named_parameters()]
        self.reg_params = [p for n, p in self.params if 'bias' not in n]          # model_utils.py:92
        self.reg_scale = 0.5 / max(1, len(self.reg_params))
        self.opt = torch.optim.Adam([p for _, p in self.params], lr=lr, betas=betas, eps=eps)
        dev = self.params[0][1].device
        self.center_acc = torch.zeros(ops.HEAD_SLOTS, device=dev, dtype=torch.float32)
        L = model.latent_dim
        self.gram_acc = torch.zeros(L, L, device=dev, dtype=torch.float32) if head == 'mahalanobis' else None
        self.steps = 0

    def set_lr(self, lr: float) -> None:
        for g in self.opt.param_groups:
            g['lr'] = lr

    def step(self, x: Tensor) -> Tensor:
        self.steps += 1
        m = self.model
        self.opt.zero_grad(set_to_none=True)
        z = m(x)
        zd = z.detach().contiguous()
        if self.head == 'euclidean':
            stats, dz, _ = ops.mse_head(zd, m.c, acc=self.center_acc)
        elif self.head == 'poincare':
            stats, dz, _, _ = ops.poincare_head(zd, m.c, acc=self.center_acc)
        elif self.head == 'mahalanobis':
            stats, dz, _ = ops.mahalanobis_head(zd, m.c, m.inv_cov_matrix, acc=self.center_acc, gram=self.gram_acc)
        else:
            raise ValueError(f"unknown head {self.head}")
        z.backward(dz)
        with torch.no_grad():
            coef = self.alpha * 2.0 * self.reg_scale          # d/dp of alpha * reg_scale * sum p^2
            for p in self.reg_params:
                if p.grad is not None:
                    p.grad.add_(p, alpha=coef)
            parallel.allreduce_grads_mean_([p for _, p in self.params], self.pg)   # one flat bucket
        self.opt.step()
        return stats

    def reg_loss(self) -> Tensor:
        with torch.no_grad():
            return self.reg_scale * sum((p.float() ** 2).sum() for p in self.reg_params).reshape(1)

    def refresh_center(self, eps: float = 1e-3) -> Tensor:
        parallel.allreduce_sum_(self.center_acc, self.pg)
        L = self.model.latent_dim
        c = ops.center_finalize(self.center_acc, eps, L) if self.head != 'poincare' else ops.midpoint_finalize(self.center_acc, L)
        self.model.c.copy_(c)
        self.center_acc.zero_()
        return self.model.c

    refresh_inv_cov = STSETrainStep.refresh_inv_cov


def make_train_step(model, **kw):
    """STSETrainStep (flat buffers, fused Adam, no autograd) for every STS-GCN encoder -- tile kernels and wide layers alike -- with a
    linear or in-width mlp projector; AutogradTrainStep for what is left: the plain-GCN encoders, projectors / latents beyond the
    bottleneck kernels."""
    from .models.common.components import Encoder
    from .models.common.components import MLP
    btl = getattr(model, 'btlnk', None)
    proj_ok = isinstance(btl, torch.nn.Linear) or (isinstance(btl, MLP) and btl.hip_ok)
    fast = proj_ok and isinstance(getattr(model, 'encoder', None), Encoder) and model.latent_dim <= 16
    if fast:
        if any(l.is_wide for l in model.encoder.model):     # wide layers: main stream, eager launches
            kw.pop('use_graph', None); kw.pop('side_stream', None)
            if kw.get('sync_bn') and not (dist.is_available() and dist.is_initialized() and dist.get_world_size(kw.get('process_group')) > 1):
                kw.pop('sync_bn')
        return STSETrainStep(model, **kw)
    kw.pop('use_graph', None); kw.pop('side_stream', None)
    if kw.pop('sync_bn', False):
        # the optional key `sync_batchnorm` (absent in the reference): with one rank there is nothing to synchronise
        world = dist.get_world_size(kw.get('process_group')) if (dist.is_available() and dist.is_initialized()) else 1
        if world > 1:
            raise ValueError("sync_bn: encoder BatchNorm only (STS-GCN encoder within the tile kernels, linear projector)")
    return AutogradTrainStep(model, **kw)
