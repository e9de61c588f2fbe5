#!/usr/bin/env python3
"""Large-input companion of tools/soak.py: 4 M .. 12 M leaves (the tile schedulers, the ticket-dealt external climb), random mesh kind and key
width; every tree validated, the two schedulers of each builder compared with each other (LBVH: node arrays byte for byte, HPLOC: topology
hash + leaves), PLOC++ validated.  Usage: python tools/soak_large.py [SECONDS=240] [SEED=1]"""
import os, sys, time
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bvh_pkg
import oracle as orc
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
builds = fails = 0
while time.time() < t_end:
    n = int(rng.integers(4_000_000, 12_000_000))
    kind = rng.choice(["uniform", "sponza", "bunny"])
    tris = pkg.meshgen.uniform(n, int(rng.integers(1, 1 << 30))) if kind == "uniform" else pkg.meshgen.sponza_like(n, int(rng.integers(1, 99))) if kind == "sponza" else pkg.meshgen.bunny_like(n, int(rng.integers(1, 99)))
    n = len(tris); bits = 60 if rng.random() < 0.3 else 30
    d_tris = ctx.upload(tris); del tris
    for algo in (1, 0, 3, 2):
        res = []
        for mode in (("async", "single"), ("block", "block")) if algo != 2 else (("", ""),):
            ctx.set_option("hploc", mode[0] or "auto"); ctx.set_option("lbvh", mode[1] or "auto")
            got = pkg.BUILDERS[algo]().build_ex(ctx, n, tris=d_tris, morton_bits=bits).download()
            k = got["sorted_keys"]
            ok = orc.validate_bvh2(got["nodes"], got["leaves"], got["root"], n, got["layout"]) == 0 and bool(np.all(k[1:] >= k[:-1]))
            builds += 1
            if not ok: fails += 1; print(f"FAIL n={n} {kind} bits={bits} {pkg.ALGO_NAMES[algo]} mode={mode}: invalid", flush=True)
            res.append(got["nodes"].tobytes() if algo != 3 else (orc.topology_hash(got["nodes"], got["leaves"], 0, n, 1), got["leaves"].tobytes()))
            del got
        if len(res) == 2 and res[0] != res[1]: fails += 1; print(f"FAIL n={n} {kind} bits={bits} {pkg.ALGO_NAMES[algo]}: schedulers disagree", flush=True)
    print(f"ok n={n} {kind} bits={bits}", flush=True)
    del d_tris
print(f"soak_large: {builds} builds, {fails} failures", flush=True)
sys.exit(1 if fails else 0)
