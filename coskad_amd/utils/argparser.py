"""Config contract of the reference (utils/argparser.py:10-45,154-166): flat yaml -> Namespace ->
derived paths + prefix-stripped sub-namespaces."""
from __future__ import annotations

import argparse
import os


def args_rm_prefix(args, prefix):
    """Namespace of the keys that start with `prefix`, with the prefix stripped (argparser.py:154-166)."""
    wp = argparse.Namespace(**vars(args))
    for key, value in vars(args).items():
        if key.startswith(prefix):
            setattr(wp, key[len(prefix):], value)
    return wp


def init_sub_args(args):
    if getattr(args, "debug", False):
        args.ae_epochs = 10                                                    # argparser.py:11-12
    args.gt_path = getattr(args, "test_path", None)
    if args.dataset_choice in ['STC', 'HR-STC', 'HR-Avenue', 'UBnormal']:
        args.pose_path = {
            'train': os.path.join(args.data_dir, 'pose', 'training/tracked_person/'),
            'test': os.path.join(args.data_dir, 'pose', 'testing/tracked_person/'),
            'validation': os.path.join(args.data_dir, 'pose', 'validating/tracked_person/')}
        if getattr(args, "validation", False):
            sub = 'validating' if args.dataset_choice == 'UBnormal' else 'testing'
            args.pose_path['validation'] = os.path.join(args.data_dir, 'pose', f'{sub}/tracked_person/')
            args.gt_path = os.path.join(args.data_dir, sub, 'test_frame_mask')
    args.ckpt_dir = os.path.join(args.exp_dir, args.dataset_choice, args.dir_name)   # create_experiment_dirs
    if getattr(args, "create_experiment_dir", False):
        os.makedirs(args.ckpt_dir, exist_ok=True)
    if not getattr(args, "dataset_sub_mean", False):
        args.dataset_return_mean = False
    return (args, args_rm_prefix(args, 'dataset_'), args_rm_prefix(args, 'ae_'), args_rm_prefix(args, 'res_'),
            args_rm_prefix(args, 'opt_'))
