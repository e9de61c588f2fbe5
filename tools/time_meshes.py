#!/usr/bin/env python3
"""whole-build time of one builder on several mesh kinds/sizes.  python tools/time_meshes.py ALGO(hploc|ploc|lbvh_single|lbvh_two) [N ...]"""
import os, sys, time
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
algo = {"hploc": pkg.ALGO_HPLOC, "ploc": pkg.ALGO_PLOCPP, "lbvh_single": pkg.ALGO_SINGLEPASS, "lbvh_two": pkg.ALGO_TWOPASS}[sys.argv[1]]
sizes = [int(x) for x in sys.argv[2:]] or [262144, 2000000, 10000000]
for n in sizes:
    for kind in ("uniform", "sponza", "bunny"):
        tris = pkg.meshgen.uniform(n, 1) if kind == "uniform" else pkg.meshgen.sponza_like(n, 3) if kind == "sponza" else pkg.meshgen.bunny_like(n, 2)
        n_eff = len(tris)
        d_tris = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
        b = pkg.BUILDERS[algo]()
        for _ in range(3): b.build(ctx, d_tris, on_device=True, n=n_eff)
        ctx.set_profiling(1); tot = []; em = []
        for _ in range(10):
            b.build(ctx, d_tris, on_device=True, n=n_eff); tot.append(b.timings.ms_total); em.append(b.timings.ms_build)
        ctx.set_profiling(0)
        print(f"{sys.argv[1]} {kind} n={n_eff} hpb={os.environ.get('BVH_HPB','-')}: total {sorted(tot)[5]:.3f} ms ({n_eff/sorted(tot)[5]/1e3:.0f} Mtris/s) emit {sorted(em)[5]:.3f} ms", flush=True)
