"""Runs ONE of bench.py's legs by itself (for rocprofv3): python tools/bench_leg.py <v25_spherical_vae|v25_encoder|wide_c256|mlp_projector|poincare_head> [steps]"""
import sys
import torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.cuda.set_device(0)
res = bench.run_legs(4096, 0, 1, torch.cuda.synchronize, steps, 3, only=name)
print(res)
