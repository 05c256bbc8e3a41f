"""Times the fused eval-mode forward at B = 4096 (and the two kernels separately) on the current GPU."""
import sys

import torch

sys.path.insert(0, ".")
from coskad_amd import engine, ops
from coskad_amd.models.graph_layers.stsgcn import layer_tensors
from coskad_amd.models.sts.ae import STSE
from coskad_amd.utils.synthetic import synthetic_clips

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
m = STSE(2, [32, 16, 32], 64, 16, 12, 17, 'sts_gcn', 'linear', 'euclidean', 0.0).cuda().eval()
x = synthetic_clips(B, seed=1).cuda()
layers = [layer_tensors(l) for l in m.encoder.model]
plan = engine.FusedEncoderPlan().get(layers, m.btlnk.weight)
H = ops.fused_encoder(x, plan.tab, plan.wreg, plan.slopes)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


t_f = timeit(lambda: ops.fused_encoder(x, plan.tab, plan.wreg, plan.slopes, out=H))
t_b = timeit(lambda: ops.btlnk_fwd(H, plan.wb, m.btlnk.bias, None))
with torch.no_grad():
    t_all = timeit(lambda: m(x))
print(f"B={B}: fused encoder {t_f:.1f} us, bottleneck {t_b:.1f} us, model(x) {t_all:.1f} us -> {B / t_all:.2f} M clips/s; "
      f"fwd roofline frac {B / (t_all * 1e-6) * 236704 / 8e12:.3f}")
