#!/usr/bin/env python3
"""per-kernel times of builds with 60-bit Morton codes (u64 keys).  python tools/time_60bit.py [N=10000000] [REPS=20]"""
import os, sys
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tris = pkg.meshgen.uniform(n, 1)
d_tris = ctx.upload(tris)
for algo in (3, 1):
    b = pkg.BUILDERS[algo]()
    for _ in range(3): b.build_ex(ctx, n, tris=d_tris, morton_bits=60)
    ck = b.checksum()
    ctx.set_profiling(2)
    for _ in range(reps): b.build_ex(ctx, n, tris=d_tris, morton_bits=60)
    kt = ctx.kernel_times(); ctx.set_profiling(0)
    print(f"{pkg.ALGO_NAMES[algo]} 60-bit n={n} checksum {ck:016x}: " + "  ".join(f"{k} {v[0] / reps:.4f}" for k, v in kt.items()) + f"  | total {sum(v[0] for v in kt.values()) / reps:.4f} ms", flush=True)
