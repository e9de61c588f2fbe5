#!/usr/bin/env python3
"""Randomised soak of the four builders on the GPU box: random size / mesh kind / key width / scheduler, every tree validated
(structure, boxes, sort order) and — where it is cheap — compared with the oracle (LBVH node arrays byte for byte, HPLOC topology
hash, repeated builds byte-identical).  Hunts for rare orderings in the inter-workgroup hand-offs.
Usage: python tools/soak.py [SECONDS=240] [SEED=1]"""
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
sizes = [2, 3, 17, 33, 511, 513, 1025, 2100, 40_000, 262_144, 500_001, 1_000_003, 2_345_678, 8_100_000]
while time.time() < t_end:
    n = int(rng.choice(sizes)) if rng.random() < 0.5 else int(np.exp(rng.uniform(np.log(2), np.log(3_000_000))))
    kind = rng.choice(["uniform", "sponza", "bunny"])
    tris = pkg.meshgen.uniform(n, int(rng.integers(1, 1 << 30))) if kind == "uniform" else pkg.meshgen.sponza_like(n, int(rng.integers(1, 99))) if kind == "sponza" else pkg.meshgen.bunny_like(n, int(rng.integers(1, 99)))
    n = len(tris)
    if n < 2: continue
    bits = 60 if rng.random() < 0.3 else 30
    d_tris = ctx.upload(tris)
    fe = orc.front_end(tris, morton_bits=bits) if n <= 1_200_000 else None
    for algo in (0, 1, 2, 3):
        for mode in (("async", "single"), ("block", "block")) if algo != 2 else (("", ""),):
            os.environ["BVH_HPLOC_MODE"] = mode[0]; os.environ["BVH_LBVH_MODE"] = mode[1]
            if not mode[0]: os.environ.pop("BVH_HPLOC_MODE"); os.environ.pop("BVH_LBVH_MODE")
            b = pkg.BUILDERS[algo]().build_ex(ctx, n, tris=d_tris, morton_bits=bits); got = b.download()
            ok = orc.validate_bvh2(got["nodes"], got["leaves"], got["root"], n, got["layout"]) == 0
            k = got["sorted_keys"]; ok = ok and bool(np.all(k[1:] >= k[:-1]))
            why = "" if ok else "invalid tree / unsorted keys"
            if ok and fe is not None:
                if not (np.array_equal(k, fe["skeys"]) and np.array_equal(got["sorted_vals"], fe["svals"])): ok = False; why = "sort differs from oracle"
                elif algo == 1:
                    ref, root = orc.lbvh_single(tris, fe["skeys"], fe["svals"])
                    if not (root == got["root"] and got["nodes"].tobytes() == ref.tobytes()): ok = False; why = "single-pass LBVH differs"
                elif algo == 0:
                    ref, _ = orc.lbvh_two(tris, fe["skeys"], fe["svals"])
                    if got["nodes"].tobytes() != ref.tobytes(): ok = False; why = "two-pass LBVH differs"
                elif algo == 3 and n <= 300_000:
                    hn, hl, _ = orc.hploc(fe["boxes"], fe["skeys"], fe["svals"])
                    if orc.topology_hash(got["nodes"], got["leaves"], 0, n, 1) != orc.topology_hash(hn, hl, 0, n, 1): ok = False; why = "HPLOC topology differs"
            if ok:                                   # a second build on the same ctx: byte-identical (scratch left clean, deterministic numbering)
                again = pkg.BUILDERS[algo]().build_ex(ctx, n, tris=d_tris, morton_bits=bits).download()
                if again["nodes"].tobytes() != got["nodes"].tobytes(): ok = False; why = "rebuild differs"
            builds += 2
            if not ok:
                fails += 1
                print(f"FAIL n={n} kind={kind} bits={bits} algo={pkg.ALGO_NAMES[algo]} mode={mode}: {why}", flush=True)
    del d_tris
for v in ("BVH_HPLOC_MODE", "BVH_LBVH_MODE"): os.environ.pop(v, None)
print(f"soak: {builds} builds, {fails} failures", flush=True)
sys.exit(1 if fails else 0)
