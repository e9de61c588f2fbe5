#!/usr/bin/env python3
"""per-kernel times of a PLOC++ build at one size + checksum.  python tools/time_ploc.py [N=10000000] [uniform|sponza] [REPS=20]"""
import os, sys
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
tris = pkg.meshgen.uniform(n, 1) if kind == "uniform" else pkg.meshgen.sponza_like(n, 3)
n = len(tris)
d_tris = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
b = pkg.PLOCNew()
for _ in range(3): b.build(ctx, d_tris, on_device=True, n=n)
ck = b.checksum()
ctx.set_profiling(2)
for _ in range(reps): b.build(ctx, d_tris, on_device=True, n=n)
kt = ctx.kernel_times(); ctx.set_profiling(0)
print(f"PLOC++ {kind} n={n} checksum {ck:016x}: " + "  ".join(f"{k} {v[0] / reps:.4f}" for k, v in kt.items()) + f"  | total {sum(v[0] for v in kt.values()) / reps:.4f} ms", flush=True)
