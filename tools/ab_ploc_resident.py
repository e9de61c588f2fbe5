"""A/B of the PLOC++ resident first launch (BVH_OPT_PLOC_SCHEDULER 1 = per-iteration launches, 2 = resident): whole-build wall ms per size, alternating."""
import os, sys, time
import numpy as np, torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
for kind, n in (("sponza", 262_144), ("uniform", 262_144), ("bunny", 150_000), ("uniform", 65_536), ("uniform", 16_384)):
    tris = {"sponza": lambda: pkg.meshgen.sponza_like(n, 3), "uniform": lambda: pkg.meshgen.uniform(n, 1), "bunny": lambda: pkg.meshgen.bunny_like(n, 2)}[kind]()
    d = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
    res = {}
    for rep in range(2):
        for mode in ("iter", "resident"):
            ctx.set_option("ploc", mode)
            b = pkg.PLOCNew()
            for _ in range(5): b.build(ctx, d, on_device=True, n=n)
            ctx.synchronize(); t0 = time.perf_counter()
            for _ in range(50): b.build(ctx, d, on_device=True, n=n)
            ctx.synchronize(); ms = (time.perf_counter() - t0) / 50 * 1e3
            ctx.set_profiling(1); b.build(ctx, d, on_device=True, n=n); emit = b.timings.ms_build; it = b.timings.ploc_iterations; ctx.set_profiling(0)
            res.setdefault(mode, []).append((ms, emit, it))
    print(kind, n, "  ".join(f"{m}: build {min(x[0] for x in v):.4f} ms (emit {min(x[1] for x in v):.4f}, {v[0][2]} iterations)" for m, v in res.items()), flush=True)
