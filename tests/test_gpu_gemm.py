"""The strided fp32 MFMA GEMM (csrc/gemm.hip) and its companions against float64 torch on the CPU: ragged sizes, transposed
/ broadcast / non-contiguous operand views, bias modes, ReLU, chunked reductions (weight gradients) with a partial last
piece, ReLU backward with channel sums, row softmax forward / backward."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(A, B):
    return torch.matmul(A.double().cpu(), B.double().cpu())


@pytest.mark.parametrize("M,N,K,batch", [(64, 64, 16, 1), (37, 204, 204, 1), (16, 204, 2, 5), (130, 70, 33, 3), (1, 1, 1, 2)])
def test_gemm_shapes_and_views(M, N, K, batch):
    from coskad_amd import ops
    g = torch.Generator().manual_seed(M * 1000 + N)
    A = torch.randn(batch, M, K, generator=g).cuda()
    B = torch.randn(batch, K, N, generator=g).cuda()
    tol = dict(rtol=1e-5, atol=1e-5 * K ** 0.5)
    np.testing.assert_allclose(ops.gemm(A, B).cpu().numpy(), _ref(A, B).numpy(), **tol)
    # transposed views (k not contiguous in A, n not contiguous in B), broadcast 2-D operands
    At = torch.randn(batch, K, M, generator=g).cuda()
    Bt = torch.randn(batch, N, K, generator=g).cuda()
    np.testing.assert_allclose(ops.gemm(At.transpose(1, 2), Bt.transpose(1, 2)).cpu().numpy(),
                               _ref(At.transpose(1, 2), Bt.transpose(1, 2)).numpy(), **tol)
    W = torch.randn(M, K, generator=g).cuda()
    np.testing.assert_allclose(ops.gemm(W, B).cpu().numpy(), _ref(W, B).numpy(), **tol)
    W2 = torch.randn(K, N, generator=g).cuda()
    np.testing.assert_allclose(ops.gemm(A, W2).cpu().numpy(), _ref(A, W2).numpy(), **tol)
    # epilogues: bias per row (modulo), per column, ReLU
    bm = torch.randn(max(1, M // 2 if M % 2 == 0 else M), generator=g).cuda()
    mod = bm.numel()
    got = ops.gemm(A, B, bias=bm, bias_mode=1, bias_mod=mod, relu=True).cpu()
    want = torch.relu(_ref(A, B) + bm.double().cpu()[torch.arange(M) % mod][:, None])
    np.testing.assert_allclose(got.numpy(), want.numpy(), **tol)
    bn = torch.randn(N, generator=g).cuda()
    np.testing.assert_allclose(ops.gemm(A, B, bias=bn, bias_mode=2).cpu().numpy(), (_ref(A, B) + bn.double().cpu()).numpy(), **tol)
    # strided output view
    out = torch.zeros(batch, M, 2 * N, device="cuda")
    ops.gemm(A, B, out=out[:, :, ::2])
    np.testing.assert_allclose(out[:, :, ::2].cpu().numpy(), _ref(A, B).numpy(), **tol)
    assert float(out[:, :, 1::2].abs().max()) == 0.0


def test_gemm_reductions():
    from coskad_amd import ops
    g = torch.Generator().manual_seed(7)
    Bn, Ci, Co, P = 150, 8, 20, 204
    Y = torch.randn(Bn, Ci, P, generator=g).cuda()
    G = torch.randn(Bn, Co, P, generator=g).cuda()
    dW = ops.gemm_reduce(Y, G.transpose(1, 2), torch.empty(Ci, Co, device="cuda"))
    want = torch.matmul(Y.double().cpu(), G.double().cpu().transpose(1, 2)).sum(0)
    np.testing.assert_allclose(dW.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-3)
    acc = ops.gemm_reduce(Y, G.transpose(1, 2), dW.clone(), accumulate=True)
    np.testing.assert_allclose(acc.cpu().numpy(), 2 * want.numpy(), rtol=1e-5, atol=2e-3)
    # one long reduction axis cut into pieces of 256 rows, the last one partial (1200 = 4 * 256 + 176)
    R = Bn * Ci
    Gr, Sr = torch.randn(R, P, generator=g).cuda(), torch.randn(R, P, generator=g).cuda()
    dA = ops.gemm_rows_outer(Gr, Sr, torch.empty(P, P, device="cuda"))
    np.testing.assert_allclose(dA.cpu().numpy(), (Gr.double().cpu().t() @ Sr.double().cpu()).numpy(), rtol=1e-5, atol=2e-3)


def test_relu_bwd_and_softmax():
    from coskad_amd import ops
    g = torch.Generator().manual_seed(3)
    O = torch.randn(70, 12, 204, generator=g).cuda()
    dO = torch.randn(70, 12, 204, generator=g).cuda()
    db = torch.empty(12, device="cuda")
    G = ops.relu_bwd(O, dO, db)
    want = dO.cpu() * (O.cpu() > 0)
    assert torch.equal(G.cpu(), want)
    np.testing.assert_allclose(db.cpu().numpy(), want.double().sum((0, 2)).numpy(), rtol=1e-5, atol=1e-4)
    x = torch.rand(204, 204, generator=g).cuda()
    y = ops.softmax_rows(x)
    np.testing.assert_allclose(y.cpu().numpy(), torch.softmax(x.cpu().double(), 1).numpy(), rtol=1e-5, atol=1e-8)
    dy = torch.randn(204, 204, generator=g).cuda()
    xr = x.cpu().double().requires_grad_(True)
    (torch.softmax(xr, 1) * dy.cpu().double()).sum().backward()
    np.testing.assert_allclose(ops.softmax_rows_bwd(y, dy).cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("M,K,P,B", [(256, 256, 204, 5), (128, 64, 204, 7), (64, 128, 204, 3), (32, 64, 204, 9), (32, 64, 300, 6),
                                     (64, 32, 300, 5), (128, 256, 204, 1)])
def test_conv1x1_vs_torch(M, K, P, B):
    """csrc/conv1x1.hip (layout-specialised 1x1 convolution of the wide layers, stsgcn.py:57-63,71-75): forward with bias, the
    data-gradient form (transposed weight view, accumulate into the output), ragged clip groups, guarded output, and the
    BatchNorm sums of its epilogue against a float64 reference."""
    from coskad_amd import ops
    g = torch.Generator().manual_seed(M + K + P + B)
    W = (torch.randn(M, K, generator=g) / K ** 0.5).cuda()
    x = torch.randn(B, K, P, generator=g).cuda()
    bias = torch.randn(M, generator=g).cuda()
    assert ops.conv1x1_ok(M, K, P)
    n, guard = B * M * P, 4096
    buf = torch.full((n + 2 * guard,), 777.0, device="cuda")
    out = buf[guard:guard + n].view(B, M, P)
    _, parts = ops.conv1x1(W, x, bias=bias, out=out, want_stats=True)
    torch.cuda.synchronize()
    assert bool((buf[:guard] == 777.0).all()) and bool((buf[guard + n:] == 777.0).all()), "wrote outside the output"
    ref = torch.einsum("mk,bkp->bmp", W.double(), x.double()) + bias.double()[None, :, None]
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-5, atol=1e-5)
    s = parts.sum(0)
    np.testing.assert_allclose(s[:, 0].cpu().numpy(), ref.sum((0, 2)).cpu().numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(s[:, 1].cpu().numpy(), (ref ** 2).sum((0, 2)).cpu().numpy(), rtol=1e-5, atol=1e-3)
    # data gradient: dX[b] += W^T dC[b] through a transposed view of W [M, K] (M of the forward is the contraction axis here)
    if ops.conv1x1_ok(K, M, P):
        dC = torch.randn(B, M, P, generator=g).cuda()
        dX0 = torch.randn(B, K, P, generator=g).cuda()
        dX = dX0.clone()
        ops.conv1x1(W.t(), dC, out=dX, accumulate=True)
        refd = dX0.double() + torch.einsum("mk,bmp->bkp", W.double(), dC.double())
        np.testing.assert_allclose(dX.cpu().numpy(), refd.float().cpu().numpy(), rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("M,K,P,B,chunks", [(256, 256, 204, 9, 4), (128, 256, 204, 5, 64), (64, 64, 204, 7, 3), (64, 128, 300, 6, 2),
                                            (128, 128, 300, 3, 64), (32, 64, 300, 4, 2), (32, 48, 204, 4, 2)])
def test_conv1x1_weight_gradient_vs_torch(M, K, P, B, chunks):
    """dW[m][k] = sum_b sum_p G[b][m][p] X[b][k][p] (autograd of the 1x1 convolution, csrc/conv1x1.hip: contraction over positions
    from LDS row images, chunked deterministic reduction) against float64; the last shape falls to the strided GEMM; accumulate."""
    from coskad_amd import ops
    g = torch.Generator().manual_seed(M * 3 + K + P + B)
    G = torch.randn(B, M, P, generator=g).cuda()
    X = torch.randn(B, K, P, generator=g).cuda()
    ref = torch.einsum("bmp,bkp->mk", G.double(), X.double())
    out = torch.empty(M, K, device="cuda")
    ops.conv1x1_wgrad(G, X, out, target_chunks=chunks)
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().cpu().numpy(), rtol=2e-5, atol=2e-4)
    again = torch.empty(M, K, device="cuda")
    ops.conv1x1_wgrad(G, X, again, target_chunks=chunks)
    assert torch.equal(out, again), "the chunked reduction is deterministic"
    ops.conv1x1_wgrad(G, X, out, target_chunks=chunks, accumulate=True)
    np.testing.assert_allclose(out.cpu().numpy(), (2 * ref).float().cpu().numpy(), rtol=2e-5, atol=4e-4)


@pytest.mark.parametrize("B,N,L", [(37, 13056, 16), (5, 19200, 8), (1030, 13056, 8), (3, 13056, 12)])
def test_rev_btlnk_kernels_vs_torch(B, N, L):
    """rev_btlnk = nn.Linear(latent -> hidden*T*V) of the decoder models (ae.py:223-227) and its autograd on csrc/rev_btlnk.hip
    (L = 12 falls to the strided GEMM): forward, dz (plain and added to an existing gradient), dW, db against float64."""
    from coskad_amd import ops
    g = torch.Generator().manual_seed(B + N + L)
    z = torch.randn(B, L, generator=g).cuda()
    W = (torch.randn(N, L, generator=g) / L ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    H = ops.rev_btlnk_fwd(z, W, b)
    ref = z.double() @ W.double().t() + b.double()
    np.testing.assert_allclose(H.cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-5, atol=1e-5)
    dH = torch.randn(B, N, generator=g).cuda()
    dW, db = torch.empty_like(W), torch.empty_like(b)
    dz = ops.rev_btlnk_bwd(dH, z, W, dW, db)
    np.testing.assert_allclose(dz.cpu().numpy(), (dH.double() @ W.double()).float().cpu().numpy(), rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(dW.cpu().numpy(), (dH.double().t() @ z.double()).float().cpu().numpy(), rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(db.cpu().numpy(), dH.double().sum(0).float().cpu().numpy(), rtol=2e-5, atol=2e-4)
    dz0 = torch.randn(B, L, generator=g).cuda()
    dz1 = ops.rev_btlnk_bwd(dH, z, W, dW, db, dz=dz0.clone())
    np.testing.assert_allclose(dz1.cpu().numpy(), (dz0.double() + dH.double() @ W.double()).float().cpu().numpy(), rtol=2e-5, atol=2e-4)
