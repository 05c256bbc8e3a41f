"""BASELINE config 5 through the reference's own entry points: `python train_COSKAD.py --config <hyperbolic yaml>`
then `python eval_COSKAD.py --config <same yaml + load_ckpt>` as child processes (train_COSKAD.py:15-85,
eval_COSKAD.py:40-253 contract: flat yaml, <exp_dir>/<dataset_choice>/<dir_name>/ checkpoints + config.yaml copy,
`final AUC score:` line), single process and 2 ranks under torch.distributed.run.  The AUC the evaluation script prints
must equal the AUC of the same checkpoint scored in-process through the Trainer."""
import ast
import glob
import os
import re
import socket
import subprocess
import sys
from argparse import Namespace

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cfg(tmp_path, name, **over):
    cfg = yaml.load(open(os.path.join(ROOT, "config", "synthetic", name)), Loader=yaml.FullLoader)
    cfg.update(dict(dict(exp_dir=str(tmp_path / "ckpt"), ae_epochs=2), **over))
    path = str(tmp_path / name)
    yaml.safe_dump(cfg, open(path, "w"))
    return cfg, path


def _run(cmd, extra_env=None):
    env = dict(os.environ, PYTHONPATH=ROOT, **(extra_env or {}))
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, f"{' '.join(cmd)}\n--- stdout\n{r.stdout[-3000:]}\n--- stderr\n{r.stderr[-3000:]}"
    return r.stdout


def _history(stdout):
    return [ast.literal_eval(l) for l in stdout.splitlines() if l.startswith("{") and "epoch" in l]


def _eval_cli(tmp_path, cfg, ckpt):
    cfg2 = dict(cfg, load_ckpt=os.path.basename(ckpt))
    p2 = str(tmp_path / "eval.yaml")
    yaml.safe_dump(cfg2, open(p2, "w"))
    out = _run([sys.executable, "eval_COSKAD.py", "--config", p2])
    m = re.search(r"final AUC score: ([0-9.eE+-]+)", out)
    assert m, out[-2000:]
    return float(m.group(1))


def _in_process_auc(cfg, ckpt):
    from coskad_amd.lit import LitEncoder, Trainer
    from coskad_amd.utils.argparser import init_sub_args
    from coskad_amd.utils.synthetic import batches, make_dataset
    args, *_ = init_sub_args(Namespace(**dict(cfg, create_experiment_dir=False)))
    lit = LitEncoder(args).cuda()
    test, gts = make_dataset(n_scenes=2, n_clips=3, n_persons=3, clip_len=200, num_transform=args.dataset_num_transform,
                             anomaly=True, seed=args.seed + 1)
    lit.gts = gts
    outs = Trainer().predict(lit, lambda: batches(test, args.dataset_batch_size), ckpt_path=ckpt)
    return float(lit.validation_epoch_end(outs))


def _check_run(tmp_path, cfg, stdout):
    hist = _history(stdout)
    assert len(hist) == cfg["ae_epochs"] and all("validation_auc" in h for h in hist), stdout[-2000:]
    ckdir = os.path.join(cfg["exp_dir"], cfg["dataset_choice"], cfg["dir_name"])
    assert os.path.exists(os.path.join(ckdir, "config.yaml"))                      # train_COSKAD.py:33
    ckpts = sorted(glob.glob(os.path.join(ckdir, "epoch=*-validation_auc=*.ckpt")))
    assert 1 <= len(ckpts) <= 2                                                    # save_top_k=2
    ck = torch.load(ckpts[-1], map_location="cpu", weights_only=False)
    assert all(k.startswith("model.") for k in ck["state_dict"]) and "args" in ck["hyper_parameters"]
    ep = int(re.search(r"epoch=(\d+)", ckpts[-1]).group(1))
    auc_cli = _eval_cli(tmp_path, cfg, ckpts[-1])
    auc_here = _in_process_auc(cfg, ckpts[-1])
    assert abs(auc_cli - auc_here) < 1e-6, (auc_cli, auc_here)
    # the synthetic validation split is the evaluation script's test split: the AUC logged while training at that
    # epoch is the AUC of that checkpoint
    assert abs(auc_cli - hist[ep]["validation_auc"]) < 1e-6, (auc_cli, hist[ep])
    assert 0.0 <= auc_cli <= 1.0
    return auc_cli


def test_train_eval_cli_hyperbolic(tmp_path):
    cfg, path = _cfg(tmp_path, "hyperbolic_encoder.yaml")
    out = _run([sys.executable, "train_COSKAD.py", "--config", path])
    _check_run(tmp_path, cfg, out)


def test_train_eval_cli_hyperbolic_two_ranks(tmp_path):
    """Same run under `python -m torch.distributed.run --nproc-per-node 2` (gloo carries the collectives when both ranks
    share one GPU; on a multi-GPU box the default backend nccl = RCCL is used)."""
    cfg, path = _cfg(tmp_path, "hyperbolic_encoder.yaml")
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "train_COSKAD.py", "--config", path],
               {"COSKAD_DIST_BACKEND": backend})
    _check_run(tmp_path, cfg, out)


def test_train_eval_cli_autoencoder_and_vae(tmp_path):
    """The decoder wrappers through the same entry points (BASELINE config 4's wrapper is the VAE one)."""
    for name in ("euclidean_autoencoder.yaml", "spherical_vae.yaml"):
        cfg, path = _cfg(tmp_path, name, ae_epochs=1)
        out = _run([sys.executable, "train_COSKAD.py", "--config", path])
        hist = _history(out)
        assert len(hist) == 1 and 0.0 <= hist[0]["validation_auc"] <= 1.0
        ckdir = os.path.join(cfg["exp_dir"], cfg["dataset_choice"], cfg["dir_name"])
        ckpts = sorted(glob.glob(os.path.join(ckdir, "*.ckpt")))
        assert ckpts
        auc = _eval_cli(tmp_path, cfg, ckpts[-1])
        assert 0.0 <= auc <= 1.0


def test_bench_dry_collectives_two_ranks_on_one_gpu():
    """bench.py --dry-collectives: two ranks (gloo, one GPU) run the data-parallel step on different clips; every step the ranks
    compare the gradient buckets' element counts, the 1 / world folded into Adam and the step count, at the end a parameter
    checksum (train_COSKAD.py:75-78: DDP keeps the replicas equal) -- the rehearsal of the first N-GPU RCCL run."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--dry-collectives", "--steps", "3", "--warmup", "1",
                        "--batch", "96"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["dry_collectives"] == "ok" and out["world"] == 2 and len(out["devices"]) == 2
    assert out["bucket_bottleneck_elems"] == 16 * 64 * 12 * 17 + 16 and out["gscale"] == 0.5 and out["steps_checked"] == 4
    assert out["bucket_bottleneck_elems"] + out["bucket_encoder_elems"] >= 239716        # every parameter is in one of the two buckets
