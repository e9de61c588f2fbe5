#!/usr/bin/env python3
"""single-pass LBVH: one-launch kernel vs tile scheduler — byte equality and emit time.  python tools/ab_lbvh.py [N ...]"""
import os, sys
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
for n in [int(x) for x in sys.argv[1:]] or [1000, 5000, 262144, 2000000, 10000000]:
    for kind in ("uniform", "sponza"):
        tris = pkg.meshgen.uniform(n, 1) if kind == "uniform" else pkg.meshgen.sponza_like(n, 3)
        d_tris = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
        res = {}
        for mode in ("single", "block"):
            ctx.set_option("lbvh", mode)
            b = pkg.SinglePassLbvh()
            for _ in range(3): b.build(ctx, d_tris, on_device=True, n=n)
            ctx.set_profiling(1); em = []
            for _ in range(10):
                b.build(ctx, d_tris, on_device=True, n=n); em.append(b.timings.ms_build)
            ctx.set_profiling(0)
            got = b.download()
            res[mode] = (got["nodes"].tobytes(), got["root"], sorted(em)[5], b.timings.ms_total)
        same = res["single"][0] == res["block"][0] and res["single"][1] == res["block"][1]
        print(f"n={n} {kind}: identical={same}  emit single {res['single'][2]:.3f} ms  block {res['block'][2]:.3f} ms  (block build total {res['block'][3]:.3f} ms = {n/res['block'][3]/1e3:.0f} Mtris/s)", flush=True)
