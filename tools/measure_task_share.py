#!/usr/bin/env python3
"""Share of the HPLOC merge tasks that the tile kernel (k_hploc_block) runs, measured on the device: the ablation build (tools/build_variant.sh abl "")
counts the local big nodes of every tile; the total is the oracle's merge_calls for the same mesh (tests/golden/reference_outputs.json "_fullsize",
or a live oracle run for other sizes).  Replaces round 2's hand-set 85 / 15 split of the emit stage's algorithmic bytes between the two kernels.
Usage (GPU box):  BVH_MI355X_LIB=build/variants/libbvh_abl.so python tools/measure_task_share.py [N=10000000] [out.json]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bvh_pkg

pkg = bvh_pkg.load(); ctx = pkg.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
tris = pkg.meshgen.uniform(n, 1)
ctx.set_option("hploc", "block")
b = pkg.HPLOC().build(ctx, tris)
v = C.c_int64()
rc = pkg.lib().bvh_ctx_get_option(ctx.handle, 1000, C.byref(v))
if rc != 0:
    raise SystemExit("this needs the ablation build of the library (BVH_MI355X_LIB=build/variants/libbvh_abl.so)")
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_outputs.json"))).get("_fullsize", {}).get(f"uniform{n}_s1")
if gold:
    total = gold["hploc"]["stats"]["merge_calls"]
else:
    import oracle as orc
    total = int(orc.build_tree(3, tris)["stats"]["merge_calls"])
out = {"mesh": f"uniform({n}, 1)", "n": n, "tile_leaves": 512, "merge_tasks_total": int(total), "merge_tasks_tile_kernel": int(v.value),
       "tile_kernel_share": round(v.value / total, 5)}
print(json.dumps(out))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
