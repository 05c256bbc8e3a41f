import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from coskad_amd import ops
from oracle import ref_cpu as R
g = torch.Generator().manual_seed(0)
for cn, zs in [(0.3, 1.0), (0.8, 1.0), (0.95, 1.0), (0.5, 3.0), (0.9, 3.0)]:
    z = torch.randn(512, 8, generator=g) * zs
    c = torch.randn(8, generator=g); c = c / c.norm() * cn
    _, _, zh, s = ops.poincare_head(z.cuda(), c.cuda(), need_grad=False, need_zh=True, need_score=True)
    zr = R.project(R.expmap0(z))
    sr = R.dist(c[None], zr)
    s64 = R.dist(c[None].double(), R.project(R.expmap0(z.double())))
    print(f"|c|={cn} zscale={zs}: zh err {float((zh.cpu()-zr).abs().max()):.2e}  hip-vs-f32oracle {float(((s.cpu()-sr)/sr).abs().max()):.2e}  hip-vs-f64 {float(((s.cpu().double()-s64)/s64).abs().max()):.2e}  f32oracle-vs-f64 {float(((sr.double()-s64)/s64).abs().max()):.2e}")
