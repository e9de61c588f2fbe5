#!/usr/bin/env python3
"""emit-stage time of one HPLOC mode at one size (no validation; for ablations).  python tools/time_hploc.py MODE N [REPS [uniform|sponza|bunny]]"""
import os, sys, time
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
mode, n = sys.argv[1], int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
kind = sys.argv[4] if len(sys.argv) > 4 else "uniform"
ctx.set_option("hploc", mode)
tris = pkg.meshgen.uniform(n, 1) if kind == "uniform" else pkg.meshgen.sponza_like(n, 3) if kind == "sponza" else pkg.meshgen.bunny_like(n, 2)
n = len(tris)
d_tris = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
b = pkg.HPLOC()
for _ in range(3): b.build(ctx, d_tris, on_device=True, n=n)
ctx.set_profiling(1)
ms = []
for _ in range(reps):
    b.build(ctx, d_tris, on_device=True, n=n); ms.append(b.timings.ms_build)
print(f"{kind} mode={mode} n={n} dbg={os.environ.get('BVH_HPLOC_DEBUG','0')}: emit min {min(ms):.3f} median {sorted(ms)[len(ms)//2]:.3f} ms", flush=True)
