#!/usr/bin/env python3
"""time of bvh_collapse4 (BVH2 -> BVH4) after a PLOC++ / LBVH build.  python tools/time_collapse.py [N]"""
import os, sys, time, ctypes as C
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0); L = pkg.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
for kind in ("sponza", "uniform"):
    tris = pkg.meshgen.sponza_like(n, 3) if kind == "sponza" else pkg.meshgen.uniform(n, 1)
    for algo in (pkg.ALGO_PLOCPP, pkg.ALGO_SINGLEPASS):
        b = pkg.BUILDERS[algo]().build(ctx, tris)
        wide = ctx.alloc(n * 128); prims = ctx.alloc(n * 8); nw = C.c_uint32()
        ts = []
        for _ in range(6):
            ctx.synchronize(); t0 = time.perf_counter()
            rc = L.bvh_collapse4(ctx.handle, C.byref(b.result), wide.ptr, prims.ptr, C.byref(nw)); ctx.synchronize()
            ts.append(time.perf_counter() - t0); assert rc == 0
        print(f"{kind} n={n} {pkg.ALGO_NAMES[algo]}: collapse4 {min(ts)*1e3:.3f} ms, {nw.value} wide nodes", flush=True)
        wide.free(); prims.free()
