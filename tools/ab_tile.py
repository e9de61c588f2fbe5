#!/usr/bin/env python3
"""Tile kernel check + timing at one size: checksum of the block-mode tree == checksum of the asynchronous one-launch tree, per-kernel
times.  python tools/ab_tile.py [N=10000000] [uniform|sponza|bunny] [REPS=20]"""
import os, sys
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
tris = pkg.meshgen.uniform(n, 1) if kind == "uniform" else pkg.meshgen.sponza_like(n, 3) if kind == "sponza" else pkg.meshgen.bunny_like(n, 2)
n = len(tris)
d_tris = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
cks = {}
for mode in ("async", "block"):
    ctx.set_option("hploc", mode)
    b = pkg.HPLOC().build(ctx, d_tris, on_device=True, n=n)
    cks[mode] = b.checksum()
print(f"{kind} n={n}: checksum async {cks['async']:016x} block {cks['block']:016x} {'EQUAL' if cks['async'] == cks['block'] else 'DIFFERENT'}", flush=True)
ctx.set_option("hploc", "block")
b = pkg.HPLOC()
for _ in range(3): b.build(ctx, d_tris, on_device=True, n=n)
ctx.set_profiling(2)
for _ in range(reps): b.build(ctx, d_tris, on_device=True, n=n)
kt = ctx.kernel_times()
print("  " + "  ".join(f"{k} {v[0] / reps:.4f}" for k, v in kt.items()) + f"  | total {sum(v[0] for v in kt.values()) / reps:.4f} ms", flush=True)
if os.environ.get("BVH_HPB_OLD") is None and os.path.exists(os.environ.get("BVH_MI355X_LIB", "")):
    pass
