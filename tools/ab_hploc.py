#!/usr/bin/env python3
"""A/B of the HPLOC schedulers on the GPU box: correctness of each mode against the CPU oracle (topology hash) at small sizes,
agreement between modes at large sizes, and per-mode build time.  Usage: python tools/ab_hploc.py [--modes block,levels,async]"""
import argparse, os, sys, time
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bvh_pkg
import oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--modes", default="block,async")
ap.add_argument("--check", default="2100,3000,5000,20000,100000,262144")
ap.add_argument("--time", default="262144,2000000,10000000")
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
modes = a.modes.split(",")

def build(mode, tris):
    ctx.set_option("hploc", mode)
    b = pkg.HPLOC().build(ctx, tris)
    return b

for n in [int(x) for x in a.check.split(",") if x]:
    for mesh in ("uniform", "sponza"):
        tris = pkg.meshgen.uniform(n, 7 + n) if mesh == "uniform" else pkg.meshgen.sponza_like(n, 3)
        ref = orc.build_tree(3, tris); h_ref = orc.topology_hash(ref["nodes"], ref["leaves"], 0, n, 1)
        for m in modes:
            got = build(m, tris).download()
            ok = orc.validate_bvh2(got["nodes"], got["leaves"], got["root"], n, got["layout"]) == 0
            same = orc.topology_hash(got["nodes"], got["leaves"], 0, n, 1) == h_ref and got["leaves"].tobytes() == ref["leaves"].tobytes()
            print(f"check n={n} {mesh} mode={m}: valid={ok} topology_equal={same}", flush=True)

for n in [int(x) for x in a.time.split(",") if x]:
    tris = pkg.meshgen.uniform(n, 1)
    d_tris = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
    hashes = {}
    for m in modes:
        ctx.set_option("hploc", m)
        b = pkg.HPLOC()
        for _ in range(3): b.build(ctx, d_tris, on_device=True, n=n)
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(a.reps): b.build(ctx, d_tris, on_device=True, n=n)
        ctx.synchronize(); dt = (time.perf_counter() - t0) / a.reps
        ctx.set_profiling(1); b.build(ctx, d_tris, on_device=True, n=n); tm = b.timings; ctx.set_profiling(0)
        got = b.download()
        ok = orc.validate_bvh2(got["nodes"], got["leaves"], got["root"], n, got["layout"]) == 0
        hashes[m] = orc.topology_hash(got["nodes"], got["leaves"], 0, n, 1)
        print(f"time n={n} mode={m}: {dt*1e3:.3f} ms/build = {n/dt/1e6:.0f} Mtris/s  (emit {tm.ms_build:.3f} ms) valid={ok}", flush=True)
    print(f"  modes agree at n={n}: {len(set(hashes.values())) == 1}", flush=True)
