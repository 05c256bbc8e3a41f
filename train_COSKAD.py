#!/usr/bin/env python
"""train_COSKAD.py -- same CLI / yaml contract as the reference (train_COSKAD.py:15-85) on the MI355X path.

  python train_COSKAD.py --config config/synthetic/euclidean_encoder.yaml
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_COSKAD.py --config ...

`data_dir: synthetic` uses coskad_amd.utils.synthetic (no datasets are reachable from this environment); any other
`data_dir` goes through the Morais-CSV pipeline of coskad_amd.utils.dataset (reference utils/dataset.py).
Multi-rank runs: every rank seeds torch with `seed`, rank 0's parameters and buffers are broadcast before the first
step (Trainer.fit), shards are wrap-padded to equal length (DistributedSampler semantics)."""
import argparse
import os
import shutil

import torch
import torch.distributed as dist
import yaml

from coskad_amd.lit import LitEncoder, Trainer
from coskad_amd.utils.argparser import init_sub_args
from coskad_amd.utils.synthetic import batches, make_dataset


def main():
    parser = argparse.ArgumentParser(description='Pose_AD_Experiment')
    parser.add_argument('-c', '--config', type=str, required=True)
    config_path = parser.parse_args().config
    args = argparse.Namespace(**yaml.load(open(config_path), Loader=yaml.FullLoader))
    args, dataset_args, ae_args, res_args, opt_args = init_sub_args(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
    if world > 1:
        # "nccl" = RCCL on ROCm; COSKAD_DIST_BACKEND=gloo rehearses the multi-rank control flow on one GPU
        dist.init_process_group(os.environ.get("COSKAD_DIST_BACKEND", "nccl"))
    if rank == 0:
        os.makedirs(args.ckpt_dir, exist_ok=True)
        shutil.copy(config_path, os.path.join(args.ckpt_dir, "config.yaml"))      # train_COSKAD.py:33
    torch.manual_seed(int(args.seed))                # same initial weights on every rank (and reproducible runs)
    if args.use_decoder:                             # wrapper selection order: train_COSKAD.py:36-55
        from coskad_amd.lit import LitAutoEncoder
        model = LitAutoEncoder(args).cuda()
    elif args.use_vae:
        from coskad_amd.lit import LitVAE
        model = LitVAE(args).cuda()
    else:
        model = LitEncoder(args).cuda()              # hyperbolic / static_center switches inside
    bs = args.dataset_batch_size
    trainer = Trainer(max_epochs=args.ae_epochs, ckpt_dir=args.ckpt_dir, save_top_k=2)
    if args.data_dir == 'synthetic':
        train, _ = make_dataset(n_scenes=4, n_clips=4, n_persons=3, clip_len=200, num_transform=args.dataset_num_transform,
                                anomaly=False, seed=args.seed)
        val, gts = make_dataset(n_scenes=2, n_clips=3, n_persons=3, clip_len=200, num_transform=args.dataset_num_transform,
                                anomaly=True, seed=args.seed + 1)
        model.gts = gts
        epoch = [0]

        def train_batches():                         # DataLoader(shuffle=True) reshuffles every epoch (the centre
            epoch[0] += 1                            # initialisation pass of setup() consumes the first permutation)
            return batches(train, bs, shuffle=True, seed=args.seed, rank=rank, world=world, epoch=epoch[0] - 1)

        trainer.fit(model, train_batches,
                    (lambda: batches(val, bs, rank=rank, world=world)) if args.validation else None)
    else:
        # Morais-format trajectories (`dataset_path_to_robust`, train_COSKAD.py:80-85): window table resident in HBM,
        # batches formed on the device; frame masks are read from args.gt_path by LitEncoder.post_processing
        from coskad_amd.utils.dataset import get_dataset_and_loader
        args.exp_dir_scaler = args.ckpt_dir
        dataset_args.exp_dir = args.ckpt_dir
        if args.validation:
            _, train_loader, _, val_loader = get_dataset_and_loader(dataset_args, split=args.split, validation=True,
                                                                    rank=rank, world=world)
        else:
            _, train_loader = get_dataset_and_loader(dataset_args, split=args.split, rank=rank, world=world)
            val_loader = None
        trainer.fit(model, lambda: train_loader, (lambda: val_loader) if val_loader is not None else None)
    if rank == 0:
        for rec in trainer.history:
            print(rec)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
