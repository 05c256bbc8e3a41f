"""Same-box A/B of the spherical-VAE train step with the commuted 32 -> 16 layers on / off (trainer.COMMUTE), three alternating rounds:
python tools/ab_commute.py [V]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coskad_amd import trainer
from coskad_amd.models.sts.vae import STSVAE
from coskad_amd.utils.synthetic import synthetic_clips

V = int(sys.argv[1]) if len(sys.argv) > 1 else 17
x = synthetic_clips(4096, 2, 12, V, seed=1).cuda()
engs = {}
for on in (True, False):
    trainer.COMMUTE = on
    torch.manual_seed(0)
    m = STSVAE(2, [32, 16, 32], 64, 8, 12, V, 'sts_gcn', 'mlp', 'euclidean', 0.0, distribution='ps')
    engs[on] = trainer.STSAETrainStep(m.cuda().train(), mode='vae', lr=1e-4, alpha=1e-6, phi=1.0, beta=1e-3, gamma=1e-2)
    for _ in range(5):
        engs[on].step(x)
torch.cuda.synchronize()
for rnd in range(3):
    for on in (True, False):
        trainer.COMMUTE = on
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            engs[on].step(x)
        torch.cuda.synchronize()
        print(f"V={V} round {rnd} commute={on}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step", flush=True)
