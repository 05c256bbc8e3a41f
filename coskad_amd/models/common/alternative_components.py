"""Plain-GCN encoders of the reference (models/common/alternative_components.py:122-297 with
models/graph_layers/learnable_gcn.py:9-113 and gcn.py:8-99).

A layer is `ReLU(A' . (X W) + b)` on X [B, T*V, C] with one dense (T*V x T*V) adjacency: learnable `softmax(Adj)` or the
fixed row-normalised skeleton-in-time graph.  These are two plain GEMMs per layer (K = C and K = T*V = 204), so on
MI355X they go to the GEMM library through torch.matmul (rocBLAS / hipBLASLt); the hand-written kernels of this repo
are reserved for the fused STS-GCN path.  state_dict keys equal the reference's (`gcns.{i}.gcn.{weight,bias,Adj}`,
buffer `Adj`)."""
from __future__ import annotations

import math
from typing import List, Union

import numpy as np
import torch
import torch.nn as nn

Tensor = torch.Tensor


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

    def forward(self, X: Tensor) -> Tensor:
        X = torch.matmul(X, self.weight)                             # 'bij,jk->bik'
        adj = torch.softmax(self.Adj, dim=1)                         # nn.Softmax() on a 2-D tensor: implicit dim=1 (:36,66)
        X = torch.matmul(adj, X)                                     # 'ij,bjk->bik'
        return X if self.bias is None else X + self.bias


class LearnablePlain_GCNN_Layer(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, time_dim: int, joints_dim: int, bias: bool = True) -> None:
        super().__init__()
        self.in_channels, self.out_channels, self.time_dim, self.joints_dim = in_channels, out_channels, time_dim, joints_dim
        self.gcn = LearnableGraphConvBlock(in_channels, out_channels, time_dim, joints_dim, bias)
        self.act = nn.ReLU()

    def forward(self, X: Tensor) -> Tensor:
        return self.act(self.gcn(X))


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

    def forward(self, X: Tensor, Adj: Tensor) -> Tensor:
        X = torch.matmul(Adj, torch.matmul(X, self.weight))
        return X if self.bias is None else X + self.bias


class StaticPlain_GCNN_Layer(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, time_dim: int, joints_dim: int, bias: bool = True) -> None:
        super().__init__()
        self.in_channels, self.out_channels, self.time_dim, self.joints_dim = in_channels, out_channels, time_dim, joints_dim
        self.gcn = GraphConvBlock(in_channels, out_channels, bias)
        self.act = nn.ReLU()

    def forward(self, X: Tensor, Adj: Tensor) -> Tensor:
        return self.act(self.gcn(X, Adj))


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
        X = X.permute(0, 2, 3, 1).reshape(B, T * V, C)               # alternative_components.py:173-175
        for gcn in self.gcns:
            X = gcn(X, *extra)
        return X.view(B, T, V, X.size(-1)).permute(0, 3, 1, 2).contiguous()


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
