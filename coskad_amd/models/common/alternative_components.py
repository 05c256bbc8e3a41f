"""Plain-GCN encoders of the reference (models/common/alternative_components.py:122-297 with
models/graph_layers/learnable_gcn.py:9-113 and gcn.py:8-99).

A layer is `ReLU(A' . (X W) + b)` on X [B, T*V, C] with one dense (T*V x T*V) adjacency: learnable `softmax(Adj)` or the
fixed row-normalised skeleton-in-time graph.  Here the activations stay in the [B, C, T*V] layout of the rest of the
library (the reference's permutes at alternative_components.py:173-175 become strides), and both products, their
gradients, the bias + ReLU epilogue and the adjacency softmax run on this repo's fp32 MFMA GEMM / elementwise kernels
(csrc/gemm.hip) -- `_PlainGCNLayerFn` below.  The (T V x T V) mixing is applied on the narrower side of the layer
(A'.(X W) = (A'.X) W).  state_dict keys equal the reference's (`gcns.{i}.gcn.{weight,bias,Adj}`, buffer `Adj`)."""
from __future__ import annotations

import math
from typing import List, Union

import numpy as np
import torch
import torch.nn as nn

from ... import ops

Tensor = torch.Tensor


class _SoftmaxRowsFn(torch.autograd.Function):
    """softmax over dim 1 of the learnable adjacency (learnable_gcn.py:36,66) on the HIP kernels."""

    @staticmethod
    def forward(ctx, adj):
        y = ops.softmax_rows(adj.contiguous())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.softmax_rows_bwd(y, dy.contiguous())


class _PlainGCNLayerFn(torch.autograd.Function):
    """O[b] = relu(W^T . X[b] . A'^T + bias) on X [B, Ci, P] -> [B, Co, P]   (learnable_gcn.py:65-72 / gcn.py:48-54 + ReLU).

    Forward and backward are strided GEMMs on csrc/gemm.hip: the channel product per clip (K = C), the position mixing as
    ONE GEMM over all (clip, channel) rows (K = N = P), weight gradients as chunked reductions with fixed-order sums."""

    @staticmethod
    def forward(ctx, X, W, Ap, bias):
        X = X.contiguous()
        B, Ci, P = X.shape
        Co = W.shape[1]
        mix_first = Ci <= Co
        if mix_first:       # Y = X . A'^T on Ci channels, then the channel product with bias + ReLU in its epilogue
            Y = ops.gemm(X.view(B * Ci, P), Ap.t()).view(B, Ci, P)
            O = ops.gemm(W.t(), Y, bias=bias, bias_mode=1 if bias is not None else 0, bias_mod=Co, relu=True)
            ctx.save_for_backward(X, W, Ap, Y, O)
        else:               # H = W^T . X on Co channels, then the mixing with bias (row % Co) + ReLU in its epilogue
            H = ops.gemm(W.t(), X)
            O = ops.gemm(H.view(B * Co, P), Ap.t(), bias=bias, bias_mode=1 if bias is not None else 0, bias_mod=Co,
                         relu=True).view(B, Co, P)
            ctx.save_for_backward(X, W, Ap, H, O)
        ctx.mix_first, ctx.has_bias = mix_first, bias is not None
        return O

    @staticmethod
    def backward(ctx, dO):
        X, W, Ap, S, O = ctx.saved_tensors
        B, Ci, P = X.shape
        Co = W.shape[1]
        need_x, need_w, need_a, need_b = ctx.needs_input_grad
        db = torch.empty(Co, device=X.device, dtype=torch.float32) if (ctx.has_bias and need_b) else None
        G = ops.relu_bwd(O, dO.contiguous(), db)                       # dO * (O > 0), bias gradient
        dX = dW = dA = None

        def adj_grad(Grows, Srows):      # dA'[p', p] = sum_r Grows[r, p'] * Srows[r, p] over all (clip, channel) rows
            return ops.gemm_rows_outer(Grows, Srows, torch.empty(P, P, device=X.device, dtype=torch.float32))

        if ctx.mix_first:
            Y = S
            if need_w:       # dW[c, o] = sum_b Y[b] . G[b]^T
                dW = ops.gemm_reduce(Y, G.transpose(1, 2), torch.empty_like(W))
            dY = ops.gemm(W, G)                                           # [B, Ci, P]
            if need_a:
                dA = adj_grad(dY.view(B * Ci, P), X.view(B * Ci, P))
            if need_x:
                dX = ops.gemm(dY.view(B * Ci, P), Ap).view(B, Ci, P)
        else:
            H = S
            if need_a:
                dA = adj_grad(G.view(B * Co, P), H.view(B * Co, P))
            dH = ops.gemm(G.view(B * Co, P), Ap).view(B, Co, P)
            if need_w:
                dW = ops.gemm_reduce(X, dH.transpose(1, 2), torch.empty_like(W))
            if need_x:
                dX = ops.gemm(W, dH)
        return dX, dW, dA, db


class LearnableGraphConvBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, n_frames: int, n_joints: int, bias: bool = True) -> None:
        super().__init__()
        self.in_channels, self.out_channels, self.n_frames, self.n_joints = in_channels, out_channels, n_frames, n_joints
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.Adj = nn.Parameter(torch.empty(n_frames * n_joints, n_frames * n_joints))
        self.reset_parameters()

    def reset_parameters(self) -> None:
        stdv = 1. / math.sqrt(self.weight.size(1))                   # learnable_gcn.py:46-50
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)
        self.Adj.data.uniform_(0.0, 1.0)

    def adjacency(self) -> Tensor:
        return _SoftmaxRowsFn.apply(self.Adj)                        # nn.Softmax() on a 2-D tensor: implicit dim=1 (:36,66)


class LearnablePlain_GCNN_Layer(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, time_dim: int, joints_dim: int, bias: bool = True) -> None:
        super().__init__()
        self.in_channels, self.out_channels, self.time_dim, self.joints_dim = in_channels, out_channels, time_dim, joints_dim
        self.gcn = LearnableGraphConvBlock(in_channels, out_channels, time_dim, joints_dim, bias)
        self.act = nn.ReLU()

    def forward(self, X: Tensor) -> Tensor:
        """X [B, Ci, T*V] -> [B, Co, T*V] (bias + ReLU fused into the second GEMM's epilogue)."""
        return _PlainGCNLayerFn.apply(X, self.gcn.weight, self.gcn.adjacency(), self.gcn.bias)


class GraphConvBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, bias: bool = True) -> None:
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels))
        stdv = 1. / math.sqrt(self.weight.size(1))                   # gcn.py:26-33
        self.weight.data.uniform_(-stdv, stdv)
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
            self.bias.data.uniform_(-stdv, stdv)
        else:
            self.register_parameter('bias', None)



class StaticPlain_GCNN_Layer(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, time_dim: int, joints_dim: int, bias: bool = True) -> None:
        super().__init__()
        self.in_channels, self.out_channels, self.time_dim, self.joints_dim = in_channels, out_channels, time_dim, joints_dim
        self.gcn = GraphConvBlock(in_channels, out_channels, bias)
        self.act = nn.ReLU()

    def forward(self, X: Tensor, Adj: Tensor) -> Tensor:
        return _PlainGCNLayerFn.apply(X, self.gcn.weight, Adj, self.gcn.bias)


class _PlainGCNEncoder(nn.Module):
    def __init__(self, input_dim: int, layer_channels: List[int], hidden_dimension: int, n_frames: int, n_joints: int,
                 dropout: float, bias: bool = True, device: Union[str, torch.device] = 'cpu') -> None:
        super().__init__()
        self.input_dim, self.layer_channels, self.hidden_dimension = input_dim, list(layer_channels), hidden_dimension
        self.n_frames, self.n_joints, self.dropout, self.bias, self.device = n_frames, n_joints, dropout, bias, device
        self.build_model()

    def _layers(self, cls):
        chans = [self.input_dim] + self.layer_channels + [self.hidden_dimension]
        return nn.ModuleList(cls(chans[i], chans[i + 1], self.n_frames, self.n_joints, bias=self.bias)
                             for i in range(len(chans) - 1))

    def _run(self, X: Tensor, *extra) -> Tensor:
        B, C, T, V = X.size()
        X = X.reshape(B, C, T * V)              # the reference's [B, T*V, C] (alternative_components.py:173-175) as strides
        for gcn in self.gcns:
            X = gcn(X, *extra)
        return X.view(B, X.size(1), T, V)


class EncoderLearnablePlainGCN(_PlainGCNEncoder):
    """Learnable dense adjacency per layer (alternative_components.py:122-181)."""

    def build_model(self) -> None:
        self.gcns = self._layers(LearnablePlain_GCNN_Layer)

    def forward(self, X: Tensor) -> Tensor:
        return self._run(X)


class EncoderStaticPlainGCN(_PlainGCNEncoder):
    """Fixed adjacency: skeleton links + self loops in every frame pair block, temporal self-links between consecutive
    frames, row-normalised D^-1 (A) (alternative_components.py:185-297)."""

    links = [(0, 1), (0, 2), (0, 5), (0, 6), (1, 2), (1, 3), (2, 4), (5, 6), (5, 7), (7, 9), (6, 8), (8, 10), (5, 11),
             (6, 12), (11, 12), (11, 13), (12, 14), (13, 15), (14, 16)]

    def build_model(self) -> None:
        T, V = self.n_frames, self.n_joints
        A = np.zeros((V, V), dtype=np.float32)
        for i, j in self.links:
            A[i, j] = A[j, i] = 1.0
        A = A + np.eye(V, V)
        A = np.repeat(np.repeat(A[np.newaxis, :, np.newaxis, :], T, axis=2), T, axis=0)   # the skeleton in EVERY (t, t') block
        t = np.arange(T - 1)[:, None]
        j = np.arange(V)[None, :]
        A[t, j, t + 1, j] = 1.0
        A[t + 1, j, t, j] = 1.0
        A = A.reshape(T * V, T * V)
        rowsum = A.sum(1)
        with np.errstate(divide='ignore'):
            r_inv = np.power(rowsum, -1).flatten()
        r_inv[np.isinf(r_inv)] = 0.
        A = np.diag(r_inv).dot(A)
        self.register_buffer('Adj', torch.tensor(A, dtype=torch.float32))
        self.gcns = self._layers(StaticPlain_GCNN_Layer)

    def forward(self, X: Tensor) -> Tensor:
        return self._run(X, self.Adj)
