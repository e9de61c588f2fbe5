import os, sys
import numpy as np, torch
torch.cuda.init()
sys.path.insert(0, "/root/repo")
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
for n, kind, algo in ((262144, "sponza", 1), (262144, "sponza", 3), (150000, "bunny", 0), (900000, "uniform", 3), (50000, "uniform", 1)):
    tris = pkg.meshgen.uniform(n, 1) if kind == "uniform" else pkg.meshgen.sponza_like(n, 3) if kind == "sponza" else pkg.meshgen.bunny_like(n, 2)
    n = len(tris)
    d = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
    b = pkg.BUILDERS[algo]()
    for _ in range(5): b.build(ctx, d, on_device=True, n=n)
    ctx.set_profiling(2)
    for _ in range(30): b.build(ctx, d, on_device=True, n=n)
    kt = ctx.kernel_times(); ctx.set_profiling(0)
    print(kind, n, pkg.ALGO_NAMES[algo], "  ".join(f"{k} {v[0]/30:.4f}" for k, v in kt.items()), f"| total {sum(v[0] for v in kt.values())/30:.4f}", flush=True)
