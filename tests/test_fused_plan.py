"""Host-side contract of the fused eval-mode encoder kernel (coskad_amd/fused_plan.py <-> csrc/fused_fwd.hip).

A lane-level numpy replay of the kernel's schedule -- the same phases, LDS image, accumulator-as-operand chaining and
tile-major output -- driven by the operand streams that fused_plan builds, checked against the CPU oracle's eval-mode
encoder + bottleneck on reference-initialised weights.  Runs without a GPU: it pins the index maps the kernel relies on."""
import numpy as np
import torch

from coskad_amd import fused_plan as FP
from oracle import ref_cpu as R

T, V, TV, LD = FP.T, FP.V, FP.TV, 206
J = np.arange(64) & 15
Q = np.arange(64) >> 4


def mfma(a, b, c):
    """v_mfma_f32_16x16x4_f32 on lane vectors: a[l] = A[i = l&15][k = l>>4], b[l] = B[k = l>>4][col = l&15],
    c/d[l][r] = D[4*(l>>4) + r][l & 15]."""
    A = np.zeros((16, 4), np.float32)
    B = np.zeros((4, 16), np.float32)
    A[J, Q] = a
    B[Q, J] = b
    D = A @ B
    out = c.copy()
    for r in range(4):
        out[:, r] += D[4 * Q + r, J]
    return out


def gather(src, idx):
    out = np.zeros(idx.shape, np.float32)
    m = idx >= 0
    out[m] = src[idx[m]]
    return out


def fold_layer(st, i):
    """Eval-mode BatchNorm folded into the two 1x1 convs (what coskad_bn_fold_f32 produces): wfold [2Ci, CoP], bias [CoP]."""
    p = f"encoder.model.{i}."
    Wt, bt = st[p + "tcn.0.weight"][:, :, 0, 0].double(), st[p + "tcn.0.bias"].double()
    Wr, br = st[p + "residual.0.weight"][:, :, 0, 0].double(), st[p + "residual.0.bias"].double()
    out = []
    bias = 0
    for W, b, pre in ((Wt, bt, p + "tcn.1."), (Wr, br, p + "residual.1.")):
        s = st[pre + "weight"].double() / torch.sqrt(st[pre + "running_var"].double() + 1e-5)
        out.append((W * s[:, None]).T)                                   # [Ci, Co]
        bias = bias + (b - st[pre + "running_mean"].double()) * s + st[pre + "bias"].double()
    return torch.cat(out, 0).float().numpy(), bias.float().numpy()


def build_src(st, latent):
    S = FP.src_layout(latent)
    src = np.zeros(S.total, np.float32)
    for i in range(4):
        wf, b = fold_layer(st, i)
        src[S.A[i]:S.A[i] + T * V * V] = st[f"encoder.model.{i}.gcn.A"].numpy().reshape(-1)
        src[S.Tm[i]:S.Tm[i] + V * T * T] = st[f"encoder.model.{i}.gcn.T"].numpy().reshape(-1)
        src[S.wfold[i]:S.wfold[i] + wf.size] = wf.reshape(-1)
        src[S.bias[i]:S.bias[i] + b.size] = b
    src[S.wb:] = st["btlnk.weight"].numpy().reshape(-1)
    return src


def prelu(x, a):
    return np.where(x > 0, x, a * x).astype(np.float32)


class Wave:
    """One wavefront of csrc/fused_fwd.hip for one clip."""

    def __init__(self, tab, wreg, slopes):
        self.tab, self.w, self.sl = tab, wreg, slopes
        self.lds = np.full(48 * LD, np.nan, np.float32)      # R1 = rows 0..31, R2 = rows 32..47 (garbage until written)
        self.R1, self.R2 = 0, 32 * LD

    # -- mixing phases --------------------------------------------------------------------------
    def temporal(self, layer, base, rows):
        for rt in range((rows + 15) // 16):
            for v in range(V):
                rec = self.tab[FP.tab_temporal_off(layer, v):][:64 * FP.TEMP_REC].reshape(64, FP.TEMP_REC)
                acc = np.zeros((64, 4), np.float32)
                for s in range(3):
                    a = np.where(16 * rt + J < rows, self.lds[base + (16 * rt + J) * LD + (4 * s + Q) * V + v], 0).astype(np.float32)
                    a = np.nan_to_num(a) if rows < 16 else a
                    acc = mfma(a, rec[:, s], acc)
                for r in range(4):
                    row = 16 * rt + 4 * Q + r
                    ok = (J < T) & (row < rows)
                    self.lds[(base + row * LD + J * V + v)[ok]] = acc[ok, r]

    def spatial(self, layer, base, rows, rt, t):
        """-> accumulator tile D[channel 16rt + 4q + r][joint j]; joint 16 goes back to the image in place."""
        chunks = [self.tab[FP.tab_spatial_off(layer, t, c):][:256].reshape(64, 4) for c in range(3)]
        rec = np.concatenate(chunks, 1)
        acc = np.zeros((64, 4), np.float32)
        ex = np.zeros(64, np.float32)
        rowok = 16 * rt + J < rows
        for s in range(5):
            vv = 4 * s + Q
            ok = (vv < V) & rowok
            a = np.zeros(64, np.float32)
            a[ok] = self.lds[(base + (16 * rt + J) * LD + t * V + vv)[ok]]
            acc = mfma(a, rec[:, s], acc)
            ex = ex + a * rec[:, 5 + s]
        ex = ex + ex[np.arange(64) ^ 16]
        ex = ex + ex[np.arange(64) ^ 32]
        w = (Q == 0) & rowok
        self.lds[(base + (16 * rt + J) * LD + t * V + 16)[w]] = ex[w]
        return acc

    # -- D-layout tile <-> LDS -----------------------------------------------------------------------
    def pos(self, tile):
        return FP.out_position(tile, J)

    def tile_write(self, base, row0, tile, acc):
        p = self.pos(tile)
        ok = p >= 0
        for r in range(4):
            self.lds[(base + (row0 + 4 * Q + r) * LD + p)[ok]] = acc[ok, r]

    def tile_read(self, base, row0, tile):
        p = np.maximum(self.pos(tile), 0)
        return np.stack([self.lds[base + (row0 + 4 * Q + r) * LD + p] for r in range(4)], 1)

    # -- the clip ------------------------------------------------------------------------------------
    def run(self, x):
        w, sl = self.w, self.sl
        R1, R2 = self.R1, self.R2
        bias = lambda row: np.stack([w[row + r] for r in range(4)], 1)
        # stage the clip: rows 0,1 = mixing copy, rows 2,3 = conv copy
        for c in range(2):
            self.lds[R2 + c * LD:R2 + c * LD + TV] = x[c].reshape(-1)
            self.lds[R2 + (2 + c) * LD:R2 + (2 + c) * LD + TV] = x[c].reshape(-1)
        # ---- layer 1 mixing, conv1, conv2 (P -> R1 rows 0..15, R -> R1 rows 16..31)
        self.temporal(0, R2, 2)
        for tile in range(FP.NTILE):
            p = np.maximum(self.pos(tile), 0)
            xc = np.select([Q == 1, Q == 2], [self.lds[R2 + 2 * LD + p], self.lds[R2 + 3 * LD + p]], 0).astype(np.float32)
            if tile < T:
                z = self.spatial(0, R2, 2, 0, tile)
                z0, z1 = z[:, 0], z[:, 1]
            else:
                z0 = np.where(Q == 0, self.lds[R2 + 0 * LD + p], 0).astype(np.float32)
                z1 = np.where(Q == 0, self.lds[R2 + 1 * LD + p], 0).astype(np.float32)
            bA = np.where(Q == 0, z0, xc).astype(np.float32)
            bB = np.where(Q == 0, z1, 0).astype(np.float32)
            x2 = []
            for ot in range(2):
                u = mfma(w[FP.W1A + ot], bA, bias(FP.B1 + 4 * ot))
                u = mfma(w[FP.W1B + ot], bB, u)
                x2.append(prelu(u, sl[0]))
            P = np.zeros((64, 4), np.float32)
            Rr = bias(FP.B2)
            for ot in range(2):
                for r in range(4):
                    P = mfma(w[FP.WP + 4 * ot + r], x2[ot][:, r], P)
                    Rr = mfma(w[FP.WR + 4 * ot + r], x2[ot][:, r], Rr)
            self.tile_write(R1, 0, tile, P)
            self.tile_write(R1, 16, tile, Rr)
        # ---- layer 2 mixing on P (commuted), U2 = gcn(P) + R, X3 -> R2, conv3 X part -> R1 (32 rows, in place)
        self.temporal(1, R1, 16)
        for tile in range(FP.NTILE):
            z = self.spatial(1, R1, 16, 0, tile) if tile < T else self.tile_read(R1, 0, tile)
            x3 = prelu(z + self.tile_read(R1, 16, tile), sl[1])
            self.tile_write(R2, 0, tile, x3)
            for ot in range(2):
                a = bias(FP.B3 + 4 * ot)
                for r in range(4):
                    a = mfma(w[FP.WX3 + 4 * ot + r], x3[:, r], a)
                self.tile_write(R1, 16 * ot, tile, a)
        # ---- layer 3 mixing on X3, conv3 Z part on top of the stored X part, X4 -> R1 + registers
        self.temporal(2, R2, 16)
        x4 = []
        for tile in range(FP.NTILE):
            z = self.spatial(2, R2, 16, 0, tile) if tile < T else self.tile_read(R2, 0, tile)
            xt = []
            for ot in range(2):
                a = self.tile_read(R1, 16 * ot, tile)
                for r in range(4):
                    a = mfma(w[FP.WZ3 + 4 * ot + r], z[:, r], a)
                a = prelu(a, sl[2])
                self.tile_write(R1, 16 * ot, tile, a)
                xt.append(a)
            x4.append(xt)
        # ---- layer 4 mixing on X4 (two row tiles), conv4 from the mixing accumulators + the X4 registers, tile-major output
        self.temporal(3, R1, 32)
        out = np.zeros(FP.KP, np.float32)
        for tile in range(FP.NTILE):
            z = [self.spatial(3, R1, 32, rt, tile) if tile < T else self.tile_read(R1, 16 * rt, tile) for rt in range(2)]
            valid = self.pos(tile) >= 0
            for ot in range(4):
                a = bias(FP.B4 + 4 * ot)
                for rt in range(2):
                    for r in range(4):
                        a = mfma(w[FP.WZ4 + 8 * ot + 4 * rt + r], z[rt][:, r], a)
                for rt in range(2):
                    for r in range(4):
                        a = mfma(w[FP.WX4 + 8 * ot + 4 * rt + r], x4[tile][rt][:, r], a)
                a = np.where(valid[:, None], prelu(a, sl[3]), 0).astype(np.float32)
                out[((tile * 4 + ot) * 64) * 4:((tile * 4 + ot) * 64 + 64) * 4] = a.reshape(-1)
        return out


def test_streams_reproduce_the_oracle_encoder():
    latent = 16
    st = R.init_stse_state(2, (32, 16, 32), 64, latent, T, V, seed=3)
    g = torch.Generator().manual_seed(4)
    for k, v in st.items():                       # non-trivial BN statistics / affine / slopes so the fold is exercised
        if k.endswith("running_mean"):
            v.add_(0.1 * torch.randn(v.shape, generator=g))
        if k.endswith("running_var"):
            v.mul_(1 + 0.3 * torch.rand(v.shape, generator=g))
        if ".tcn.1." in k or ".residual.1." in k:
            if k.endswith(("weight", "bias")):
                v.add_(0.2 * torch.randn(v.shape, generator=g))
        if k.endswith("prelu.weight"):
            v.add_(0.1 * torch.randn(v.shape, generator=g))
    x = R.synthetic_clips(3, seed=5)
    acts = []
    with torch.no_grad():
        z_ref = R.stse_encode(x, st, training=False, collect=acts)
    src = build_src(st, latent)
    tab = gather(src, FP.tab_index(latent))
    wreg = gather(src, FP.wreg_index(latent))
    wb = gather(src, FP.wb_index(latent))
    slopes = np.array([float(st[f"encoder.model.{i}.prelu.weight"]) for i in range(4)], np.float32)
    for n in range(x.shape[0]):
        out = Wave(tab, wreg, slopes).run(x[n].numpy())
        # tile-major output == PReLU(U4) of the oracle, element for element
        h = acts[-1][n].numpy().reshape(64, TV)             # activated output of the last layer
        for tile in (0, 5, 11, 12):
            p = FP.out_position(tile, J)
            for ot in range(4):
                blk = out[((tile * 4 + ot) * 64) * 4:((tile * 4 + ot) * 64 + 64) * 4].reshape(64, 4)
                for r in range(4):
                    ok = p >= 0
                    np.testing.assert_allclose(blk[ok, r], h[(16 * ot + 4 * Q + r)[ok], p[ok]], rtol=2e-4, atol=2e-5)
                    assert np.all(blk[~ok, r] == 0)
        z = wb @ out + st["btlnk.bias"].numpy()
        np.testing.assert_allclose(z, z_ref[n].numpy(), rtol=1e-4, atol=1e-4)


def test_index_shapes_and_padding():
    assert FP.tab_index().shape == (FP.TAB_FLOATS,) and FP.wreg_index().shape == (FP.NWREG, 64)
    wb = FP.wb_index(8)
    assert wb.shape == (8, FP.KP)
    S = FP.src_layout(8)
    used = wb[0][wb[0] >= 0] - S.wb
    assert sorted(used.tolist()) == list(range(64 * TV))          # every bottleneck column exactly once
    assert (wb[0] < 0).sum() == FP.KP - 64 * TV
    assert FP.supports((2, 32, 16, 32, 64), 12, 17) and not FP.supports((2, 32, 16, 32, 64), 12, 25)
