#!/usr/bin/env python3
"""CPU baseline on the configs' own meshes ("exactly the meshes of the GPU run"): the reference's single-threaded binned-SAH
builder SahBvh::build (src/BinnedSahBvh.cpp:13-203; oracle port, bucket index clamped) timed on this box's host cores for Bunny-class 150 k
(config 1), Sponza-class 262 144 (configs 2 and 4), the 2 M meshes of config 5 and, with --full, the 10 M mesh of config 3 (bench.py times that
one itself).  Prints one JSON object; python tools/cpu_baselines.py [--full] > profiles/rNN_cpu_baselines.json"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bvh_pkg
import oracle as orc
pkg = bvh_pkg.load(); mg = pkg.meshgen
cases = [("bunny_like(150000, seed 2)", lambda: mg.bunny_like(150_000, 2)), ("sponza_like(262144, seed 3)", lambda: mg.sponza_like(262_144, 3)),
         ("uniform(2000000, seed 100, offset (0,0,0))", lambda: mg.uniform(2_000_000, 100))]
if "--full" in sys.argv:
    cases.append(("uniform(10000000, seed 1)", lambda: mg.uniform(10_000_000, 1)))
out = {"builder": "oracle port of SahBvh::build (src/BinnedSahBvh.cpp:13-203), 1 thread", "host": os.uname().nodename, "cpus": os.cpu_count(), "meshes": {}}
for name, gen in cases:
    tris = np.ascontiguousarray(gen()); n = len(tris)
    best = None
    for _ in range(3 if n <= 300_000 else 1):
        t0 = time.perf_counter(); nodes, total = orc.binned_sah_build(tris); dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    sah, ref_formula = orc.sah_binned(nodes, total, n)
    out["meshes"][name] = {"n": n, "build_ms": round(best * 1e3, 2), "Mtris_per_s": round(n / best / 1e6, 4), "sah": round(sah, 4), "reference_formula_cost": round(float(ref_formula), 4)}
print(json.dumps(out, indent=1))
