#!/usr/bin/env python3
"""Exact algorithmic bytes per primitive of the PLOC-family emit stages ("oracle must report exact L, S per mesh"), from the
pinned CPU oracle's cluster-load / store counts.  TEST / MEASUREMENT INFRASTRUCTURE: writes profiles/algorithmic_bytes.json, which bench.py reads as data
(the product and the bench's timed path never call the oracle).  The headline mesh uses the committed full-size golden (tools/make_golden.py fullsize);
other (mesh, n) pairs are evaluated on a twin of at most 1 M triangles of the same generator, as §8(d) prescribes.
  HPLOC  emit: keys 4 + parentIdx exchange 16 + cluster-id loads / stores 4 (L + S) / N + AABB loads 28 L / N + internal node 32
  PLOC++ emit: [sum_i C_i (id 4 + AABB 28) + sum_i C_(i+1) 4] / N + 32 per merge          (L = sum C_i, S = sum C_(i+1))
  pipeline   : E 88 + M 32 + S 68 + SetupClusters (64 HPLOC / 60 PLOC++) + emit
Usage: python tools/algorithmic_bytes.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bvh_pkg
import oracle as orc

pkg = bvh_pkg.load()
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_outputs.json"))).get("_fullsize", {})


def emit_bytes(algo, st, n):
    L, S = st["cluster_loads"], st["cluster_stores"]
    if algo == "hploc":
        return 4.0 + 16.0 + 4.0 * (L + S) / n + 28.0 * L / n + 32.0
    return (32.0 * L + 4.0 * S) / n + 32.0


def entry(algo, st, n, source):
    e = emit_bytes(algo, st, n)
    setup = 64.0 if algo == "hploc" else 60.0
    return {"n_evaluated": n, "source": source, "L_over_N": round(st["cluster_loads"] / n, 5), "S_over_N": round(st["cluster_stores"] / n, 5),
            "merge_calls": st.get("merge_calls"), "nn_rounds": st.get("nn_rounds"), "iterations": st.get("iterations"),
            "emit_bytes_per_prim": round(e, 3), "setup_bytes_per_prim": setup, "pipeline_bytes_per_prim": round(88.0 + 32.0 + 68.0 + setup + e, 3)}


out = {"_doc": __doc__.split("Usage")[0].strip()}
for name, g in gold.items():                      # the configs' own meshes at full size
    n = g["n"]
    for algo in ("hploc", "ploc"):
        if algo in g and "stats" in g[algo]:
            out[f"{g['generator'].split('.')[1].split('(')[0]}_{n}_tris_{algo}"] = entry(algo, g[algo]["stats"], n, f"pinned oracle at full size (tests/golden/reference_outputs.json _fullsize/{name})")
for mesh, n_full, gen in (("sponza", 262_144, lambda m: pkg.meshgen.sponza_like(m, 3)), ("bunny", 150_000, lambda m: pkg.meshgen.bunny_like(m, 2)),
                          ("sponza", 10_000_000, lambda m: pkg.meshgen.sponza_like(m, 3)), ("bunny", 10_000_000, lambda m: pkg.meshgen.bunny_like(m, 2))):
    m = min(n_full, 1_000_000)
    tris = gen(m)
    for algo, a in (("hploc", 3), ("ploc", 2)):
        st = {k: int(v) for k, v in orc.build_tree(a, tris)["stats"].items()}
        out[f"{mesh}_{n_full}_tris_{algo}"] = entry(algo, st, len(tris), "pinned oracle on the mesh itself" if m == n_full else f"pinned oracle on a {len(tris)}-triangle twin of the same generator")
json.dump(out, open(os.path.join(ROOT, "profiles", "algorithmic_bytes.json"), "w"), indent=1, sort_keys=True)
for k, v in out.items():
    if k != "_doc":
        print(k, v["emit_bytes_per_prim"], v["pipeline_bytes_per_prim"], v["source"])
