"""Operand streams of the fused eval-mode encoder kernel (csrc/fused_fwd.hip).

The kernel runs the reference's whole eval-mode `Encoder` (models/common/components.py:94-105: four
ST_GCNN layers, stsgcn.py:94-116) for one clip per wavefront with every activation resident in LDS or
registers.  It consumes its parameters as *operand streams* -- each MFMA operand already in the lane order
the instruction wants -- so that nothing is indexed, transposed or padded inside the kernel:

  tab   : the mixing matrices T[V,T,T], A[T,V,V] of the four layers as B operands of
          v_mfma_f32_16x16x4_f32, in consumption order (one 16-byte record per lane and chunk);
  wreg  : the BatchNorm-folded 1x1-conv weights as A operands + the folded biases in accumulator
          layout, one row of 64 floats per register the wave keeps for the whole launch;
  wb    : the bottleneck weight (models/sts/ae.py:157) permuted to the kernel's tile-major output.

This module only builds the int32 GATHER INDICES of those streams (pure numpy, cached per geometry);
`coskad_gather_f32` applies them on the device to the concatenated parameter vector.  The index maps are
the contract between this file and the kernel; tests/test_fused_plan.py replays the kernel's tile
algebra in numpy from these very streams against the CPU oracle.

Lane geometry of v_mfma_f32_16x16x4_f32 (cdna_hip_programming.md, section 3): lane l, j = l & 15, q = l >> 4:
  A operand: A[i = j][k = q]     B operand: B[k = q][col = j]     D: reg r <-> D[row = 4q + r][col = j]
An accumulator tile D[channel][position] is the B operand of the next 1x1 conv as it stands: register r
carries channel 4q + r on the k = q slot, so the conv weights are stored in that (permuted) k order.
"""
from __future__ import annotations

from dataclasses import dataclass
from functools import lru_cache
from typing import Dict, List, Tuple

import numpy as np

T, V = 12, 17
TV = T * V
CH = (2, 32, 16, 32, 64)          # the reference's default stack (config/*/*.yaml: channels [32,16,32], h_dim 64)
LANES = 64
NTILE = T + 1                     # position tiles: one per frame (joints 0..15) + the 17th-joint column of all frames
KP = NTILE * 4 * LANES * 4        # padded bottleneck K of the tile-major output (13 312)

# ---- tab stream: float offsets (per layer) -------------------------------------------------------
TEMP_REC = 4                      # floats per lane and temporal item (3 k-steps + pad)
SPAT_REC = 12                     # floats per lane and spatial item: b[5], bw[5], pad[2]  (3 chunks of 4)
LAYER_TAB = V * LANES * TEMP_REC + T * 3 * LANES * 4


def tab_temporal_off(layer: int, v: int) -> int:
    return layer * LAYER_TAB + v * LANES * TEMP_REC


def tab_spatial_off(layer: int, t: int, chunk: int) -> int:
    return layer * LAYER_TAB + V * LANES * TEMP_REC + (t * 3 + chunk) * LANES * 4


TAB_FLOATS = 4 * LAYER_TAB

# ---- wreg rows ------------------------------------------------------------------------------------
# conv1 (2 -> 32):  W1A[ot], W1B[ot]                      ot in 0..1
# conv2 (32 -> 16): WP[ot][r], WR[ot][r]                  (commuted layer 2: P = Wz.X2 goes through the mixing,
#                                                           R = Wx.X2 + b joins after it)
# conv3 (16 -> 32): WX3[ot][r], WZ3[ot][r]
# conv4 (32 -> 64): WZ4[ot][rt][r], WX4[ot][rt][r]
# biases in accumulator layout: B1[ot][r], B2[r], B3[ot][r], B4[ot][r]
W1A, W1B = 0, 2
WP, WR = 4, 12
WX3, WZ3 = 20, 28
WZ4, WX4 = 36, 68
B1, B2, B3, B4 = 100, 108, 112, 120
NWREG = 136


@dataclass(frozen=True)
class SrcLayout:
    """Offsets of the parameter tensors inside the concatenated source vector."""
    A: Tuple[int, ...]
    Tm: Tuple[int, ...]
    wfold: Tuple[int, ...]      # [2*Ci][CoP]
    bias: Tuple[int, ...]       # [CoP]
    wb: int                     # [L][hid*T*V]
    total: int


def src_layout(latent: int) -> SrcLayout:
    off = 0
    A, Tm, wf, bs = [], [], [], []
    for l in range(4):
        ci, co = CH[l], CH[l + 1]
        cop = (co + 15) // 16 * 16
        A.append(off); off += T * V * V
        Tm.append(off); off += V * T * T
        wf.append(off); off += 2 * ci * cop
        bs.append(off); off += cop
    wb = off
    off += latent * CH[4] * TV
    return SrcLayout(tuple(A), tuple(Tm), tuple(wf), tuple(bs), wb, off)


def _lanes():
    l = np.arange(LANES)
    return l & 15, l >> 4


@lru_cache(maxsize=None)
def tab_index(latent: int = 16) -> np.ndarray:
    """int32 [TAB_FLOATS]: source index of every float of the tab stream (-1 = 0.0)."""
    S = src_layout(latent)
    j, q = _lanes()
    idx = np.full(TAB_FLOATS, -1, dtype=np.int64)
    for l in range(4):
        for v in range(V):               # temporal item: B[k = t][col = q'] = T[v][t][q'],  t = 4s + q
            base = tab_temporal_off(l, v)
            for s in range(3):
                t = 4 * s + q
                val = np.where(j < T, S.Tm[l] + v * T * T + t * T + np.minimum(j, T - 1), -1)
                idx[base + np.arange(LANES) * TEMP_REC + s] = val
        for t in range(T):               # spatial item: B[k = v][col = w] = A[t][v][w], v = 4s + q; bw = column 16
            rec = np.full((LANES, SPAT_REC), -1, dtype=np.int64)
            for s in range(5):
                vv = 4 * s + q
                ok = vv < V
                vc = np.minimum(vv, V - 1)
                rec[:, s] = np.where(ok, S.A[l] + t * V * V + vc * V + j, -1)
                rec[:, 5 + s] = np.where(ok, S.A[l] + t * V * V + vc * V + 16, -1)
            for c in range(3):
                base = tab_spatial_off(l, t, c)
                idx[base:base + LANES * 4] = rec[:, 4 * c:4 * c + 4].reshape(-1)
    return idx.astype(np.int32)


@lru_cache(maxsize=None)
def wreg_index(latent: int = 16) -> np.ndarray:
    """int32 [NWREG, 64]: A operands (lane: output row i = j of the tile, k = q) and bias quads."""
    S = src_layout(latent)
    j, q = _lanes()
    idx = np.full((NWREG, LANES), -1, dtype=np.int64)

    def w(layer, part, c, o):            # folded weight of `layer` (0-based): part 0 = Z (tcn), 1 = X (residual)
        ci, co = CH[layer], CH[layer + 1]
        cop = (co + 15) // 16 * 16
        return S.wfold[layer] + (part * ci + c) * cop + o

    for ot in range(2):                  # conv1: k slots of step A = (Z c0, X c0, X c1, -), of step B = (Z c1, -, -, -)
        o = 16 * ot + j
        idx[W1A + ot] = np.select([q == 0, q == 1, q == 2], [w(0, 0, 0, o), w(0, 1, 0, o), w(0, 1, 1, o)], -1)
        idx[W1B + ot] = np.where(q == 0, w(0, 0, 1, o), -1)
    for ot in range(2):                  # conv2: input channel 16*ot + 4q + r (accumulator order), 16 outputs
        for r in range(4):
            c = 16 * ot + 4 * q + r
            idx[WP + 4 * ot + r] = w(1, 0, c, j)
            idx[WR + 4 * ot + r] = w(1, 1, c, j)
    for ot in range(2):                  # conv3: 16 inputs (4q + r), outputs 16*ot + j
        for r in range(4):
            c = 4 * q + r
            idx[WX3 + 4 * ot + r] = w(2, 1, c, 16 * ot + j)
            idx[WZ3 + 4 * ot + r] = w(2, 0, c, 16 * ot + j)
    for ot in range(4):                  # conv4: 32 inputs (16*rt + 4q + r), outputs 16*ot + j
        for rt in range(2):
            for r in range(4):
                c = 16 * rt + 4 * q + r
                idx[WZ4 + 8 * ot + 4 * rt + r] = w(3, 0, c, 16 * ot + j)
                idx[WX4 + 8 * ot + 4 * rt + r] = w(3, 1, c, 16 * ot + j)
    for layer, base, nt in ((0, B1, 2), (1, B2, 1), (2, B3, 2), (3, B4, 4)):
        for ot in range(nt):
            for r in range(4):
                idx[base + 4 * ot + r] = S.bias[layer] + 16 * ot + 4 * q + r
    return idx.astype(np.int32)


def out_position(tile: int, j: np.ndarray) -> np.ndarray:
    """Position p = t*V + v of column j of a tile (-1: padding column of the 17th-joint tile)."""
    if tile < T:
        return tile * V + j
    return np.where(j < T, np.minimum(j, T - 1) * V + 16, -1)


@lru_cache(maxsize=None)
def wb_index(latent: int = 16) -> np.ndarray:
    """int32 [latent, KP]: bottleneck weight in the kernel's output order
    k' = ((tile*4 + ot)*64 + lane)*4 + r  <->  channel 16*ot + 4q + r, position out_position(tile, j)."""
    S = src_layout(latent)
    j, q = _lanes()
    idx = np.full((latent, KP), -1, dtype=np.int64)
    for tile in range(NTILE):
        p = out_position(tile, j)
        for ot in range(4):
            for r in range(4):
                o = 16 * ot + 4 * q + r
                kp = ((tile * 4 + ot) * LANES + np.arange(LANES)) * 4 + r
                src = np.where(p >= 0, o * TV + np.maximum(p, 0), -1)
                for lat in range(latent):
                    idx[lat, kp] = np.where(src >= 0, S.wb + lat * CH[4] * TV + src, -1)
    return idx.astype(np.int32)


def supports(chans, n_frames: int, n_joints: int) -> bool:
    """The fused kernel is built for the reference's default geometry only; everything else takes the per-layer path."""
    return tuple(chans) == CH and n_frames == T and n_joints == V
