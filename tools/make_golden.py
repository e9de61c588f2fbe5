#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.

  step 1 (dev container, needs /root/reference):  python tools/make_golden.py meshes
      Cornell-box triangle lists exactly as the reference's MeshLoader::loadScene (src/Utility.cpp:614-760, run from
      oracle/_ref/libref_utility.so = the reference's unmodified Utility.cpp) emits them, as raw little-endian f32 (n x 9).
  step 2 (GPU box, through gpurun):  python tools/make_golden.py reference gpurun_out/golden_reference.json
      Outputs of the REFERENCE's own kernels (oracle/_ref/*.co via oracle/ref_driver.cpp) on the golden meshes: FNV-1a of the
      Morton keys and of the LBVH node arrays, single-pass root index, HPLOC canonical topology hash / SAH (kernel built
      with FP contraction off), plus the reference's Utility::calculateLbvhCost of its own LBVH tree.
      Copy the JSON to tests/golden/reference_outputs.json.
  step 3 (dev container, needs /root/reference):  python tools/make_golden.py ploc
      Outputs of the REFERENCE's PLOC++ kernels (SetupClusters, Ploc, SinglePassPloc) executed under the CPU SIMT emulator
      (tools/oracle/ref_emulator.cpp, built by oracle/Makefile into oracle/_ref/libref_ploc_emu.so) on the golden meshes and three larger
      ones: FNV-1a of the node array, canonical topology hash, SAH, host-loop iterations — merged into reference_outputs.json under
      "_ploc_emulated"; and of its two CollapseToWide4Bvh kernels (whole grid resident; on the GPU they hang unless every workgroup is) under
      "_collapse_emulated".  (Rounds 2-4 believed this to be the only executable form of `Ploc` here; step 5 runs the reference's own wave64 flavour on the MI355X.)
  step 4 (anywhere; CPU only, ~1 min):  python tools/make_golden.py fullsize
      The pinned CPU oracle on the configs' own sizes: uniform(10 000 000, seed 1) (config 3: HPLOC; also PLOC++) and uniform(2 000 000, seed 100)
      (config 5's first mesh: HPLOC, PLOC++, single-pass LBVH): canonical topology hash, f64 SAH, FNV-1a of leaves / node array, PLOC++ iterations,
      and the oracle's exact cluster loads / stores / merge calls / NN rounds (the L, S of SURVEY.md §8(d)'s byte formulas) -> "_fullsize".
  step 5 (GPU box, through gpurun):  python tools/make_golden.py ploc_hw gpurun_out/golden_ploc_hw.json
      Outputs of the REFERENCE's CalculateSceneExtents / SetupClusters / Ploc / SinglePassPloc / CollapseToWide4Bvh kernels ON THE MI355X — the reference's own
      wave64 flavour of the unmodified headers (oracle/_ref/*.w64.nofma.co; src/Common.h:100-106 under -D__gfx90a__=1): FNV-1a of boxes / scene extent / leaves,
      host-loop iterations, canonical BVH2 topology, SAH, number of wide nodes and canonical wide topology of its collapse of its own tree.  Merge the JSON into
      tests/golden/reference_outputs.json under "_ploc_hw" (python tools/make_golden.py merge_ploc_hw gpurun_out/golden_ploc_hw.json).
Fixtures are data (inputs and expected outputs); no reference source text is stored.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CORNELL = {"cornell32": "cornellBox", "cornell82": "untitled", "cornell382": "xyz"}


def golden_meshes(pkg):
    mg = pkg.meshgen
    out = {name: mg.load_tri(os.path.join(GOLDEN, name + ".tri")) for name in CORNELL}
    out["uniform5000_s77"] = mg.uniform(5000, 77)
    out["sponza8000_s3"] = mg.sponza_like(8000, 3)
    out["bunny6000_s2"] = mg.bunny_like(6000, 2)
    out["probe5000"] = mg.np_mt_mesh(5000)
    out["a4_probe5000"] = mg.probe_mesh(5000)        # SURVEY.md Appendix A.4 (std::mt19937(1234))
    return out


def ploc_meshes(pkg):
    mg = pkg.meshgen
    out = dict(golden_meshes(pkg))
    out["a4_probe50000"] = mg.probe_mesh(50000)            # 105.958 / 18 iterations
    out["sponza40000_s3"] = mg.sponza_like(40000, 3)
    out["bunny30000_s2"] = mg.bunny_like(30000, 2)
    return out


def ploc_hw_meshes(pkg):
    mg = pkg.meshgen
    out = dict(ploc_meshes(pkg))
    out["uniform1023_s31"] = mg.uniform(1023, 31); out["uniform1024_s32"] = mg.uniform(1024, 32); out["uniform1025_s33"] = mg.uniform(1025, 33)
    out["uniform2049_s34"] = mg.uniform(2049, 34)
    out["sponza262144_s3"] = mg.sponza_like(262_144, 3)        # BASELINE.json config 4's own size
    return out


def ploc_hw_entry(orc, tris):
    """what the reference's own kernels produce on the MI355X for one mesh (schedule-independent quantities only: node numbering is not)"""
    n = len(tris)
    boxes, scene = orc.ref_extents(tris, nofma=True)
    keys, _ = orc.morton_codes(boxes, scene)
    order = np.argsort(keys, kind="stable").astype(np.uint32)
    nodes, leaves, iters = orc.ref_ploc(boxes, order, nofma=True)
    assert orc.validate_bvh2(nodes, leaves, 0, n, 1) == 0
    wide, prims, total, placed = orc.ref_collapse(nodes, leaves, 0, n, 1, nofma=True)
    assert placed == n
    return {"n": n, "boxes_fnv": "%016x" % orc.fnv1a(boxes), "scene_fnv": "%016x" % orc.fnv1a(scene), "ploc_iterations": iters,
            "ploc_leaves_fnv": "%016x" % orc.fnv1a(leaves), "ploc_topology": "%016x" % orc.topology_hash(nodes, leaves, 0, n, 1),
            "ploc_sah_f64": round(orc.sah_bvh2(nodes, leaves, 0, n, 1)[0], 9),
            "collapse_n_wide": total, "collapse_topology4": "%016x" % orc.topology_hash4(wide, prims, total, n)}


def make_ploc_hw(out_path):
    import bvh_pkg
    import oracle as orc
    pkg = bvh_pkg.load()
    out = {}
    for name, tris in ploc_hw_meshes(pkg).items():
        out[name] = ploc_hw_entry(orc, tris)
        print(name, out[name], flush=True)
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)


def merge_ploc_hw(in_path):
    path = os.path.join(GOLDEN, "reference_outputs.json")
    res = json.load(open(path)); res["_ploc_hw"] = json.load(open(in_path))
    json.dump(res, open(path, "w"), indent=1, sort_keys=True)


def make_ploc():
    import bvh_pkg
    import oracle as orc
    pkg = bvh_pkg.load()
    path = os.path.join(GOLDEN, "reference_outputs.json")
    res = json.load(open(path))
    emu = {}
    for name, tris in ploc_meshes(pkg).items():
        n = len(tris)
        boxes, scene = orc.prim_bounds(tris)
        keys, _ = orc.morton_codes(boxes, scene)
        order = np.argsort(keys, kind="stable").astype(np.uint32)
        r = orc.ref_emu_ploc(boxes, order)
        if r is None:
            raise SystemExit("oracle/_ref/libref_ploc_emu.so missing: run make -C oracle ref in the dev container")
        nodes, leaves, iters = r
        assert orc.validate_bvh2(nodes, leaves, 0, n, 1) == 0
        emu[name] = {"n": n, "ploc_iterations": iters, "ploc_nodes_fnv": "%016x" % orc.fnv1a(nodes), "ploc_leaves_fnv": "%016x" % orc.fnv1a(leaves),
                     "ploc_topology": "%016x" % orc.topology_hash(nodes, leaves, 0, n, 1), "ploc_sah_f64": orc.sah_bvh2(nodes, leaves, 0, n, 1)[0]}
        print(name, emu[name])
    res["_ploc_emulated"] = emu
    # the reference's CollapseToWide4Bvh kernels (LBVH flavour on the single-pass tree, PLOC flavour on the PLOC++ tree) under the same emulator,
    # whole grid resident: number of wide nodes, canonical wide topology, BVH4 cost (f64 sum of the reference formula's terms)
    col = {}
    for name, tris in golden_meshes(pkg).items():
        n = len(tris); boxes, _ = orc.prim_bounds(tris)
        entry = {"n": n}
        for algo, tag in ((1, "lbvh_single"), (2, "ploc")):
            t = orc.build_tree(algo, tris)
            w, p, nw = orc.ref_emu_collapse(t["nodes"], t["leaves"], t["root"], n, t["layout"])
            entry[tag] = {"n_wide": nw, "topology4": "%016x" % orc.topology_hash4(w, p, nw, n), "bvh4_cost_f64": orc.sah_bvh4(w, p, boxes, nw, n)[0]}
        col[name] = entry
        print("collapse", name, entry)
    res["_collapse_emulated"] = col
    json.dump(res, open(path, "w"), indent=1, sort_keys=True)


def make_meshes():
    import oracle as orc
    R = orc.ref_utility()
    if R is None:
        raise SystemExit("oracle/_ref/libref_utility.so missing: run make -C oracle ref in the dev container")
    base = "/root/reference/src/Meshes/cornellbox/"
    for out_name, obj in CORNELL.items():
        path = (base + obj + ".obj").encode()
        n = R.ref_loadScene(path, base.encode(), None, 0)
        t = np.zeros(n, dtype=orc.TRIANGLE)
        R.ref_loadScene(path, base.encode(), t.ctypes.data, n)
        raw = np.stack([t["v1"], t["v2"], t["v3"]], axis=1).astype("<f4")
        raw.tofile(os.path.join(GOLDEN, out_name + ".tri"))
        print(out_name, n, "triangles")


def make_reference(out_path):
    import bvh_pkg
    import oracle as orc
    pkg = bvh_pkg.load()
    R = orc.ref_utility()
    res = {}
    for name, tris in golden_meshes(pkg).items():
        n = len(tris)
        boxes, scene = orc.prim_bounds(tris)            # min/max only: order independent, identical on any correct implementation
        keys, _ = orc.ref_morton(boxes, scene)
        order = np.argsort(keys, kind="stable")
        skeys, svals = keys[order], order.astype(np.uint32)
        n1, root1 = orc.ref_lbvh_single(tris, skeys, svals)
        n2 = orc.ref_lbvh_two(tris, skeys, svals)
        cover = (n - 1) % 32 == 0
        hn, hl, merged = orc.ref_hploc(boxes, skeys, svals, nofma=True, cover_all=cover)
        entry = {
            "n": n, "keys_fnv": "%016x" % orc.fnv1a(keys), "sorted_key_first": int(skeys[0]), "sorted_key_last": int(skeys[-1]),
            "duplicate_keys": int(n - len(np.unique(keys))),
            "lbvh_single_fnv": "%016x" % orc.fnv1a(n1), "lbvh_single_root": root1,
            "lbvh_two_fnv": "%016x" % orc.fnv1a(n2),
            "lbvh_sah_f64": orc.sah_bvh2(n1, None, root1, n, 0)[0],
            "hploc_topology": "%016x" % orc.topology_hash(hn, hl, 0, n, 1), "hploc_sah_f64": orc.sah_bvh2(hn, hl, 0, n, 1)[0],
            "hploc_merged": merged,
        }
        if R is not None:
            entry["ref_utility_lbvh_cost_f32"] = float(R.ref_calculateLbvhCost(n1.ctypes.data, root1, n, n - 1))
        res[name] = entry
        print(name, entry)
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)


def make_fullsize():
    import time
    import bvh_pkg
    import oracle as orc
    pkg = bvh_pkg.load()
    path = os.path.join(GOLDEN, "reference_outputs.json")
    res = json.load(open(path))
    out = {}
    for name, n, seed, algos in (("uniform10000000_s1", 10_000_000, 1, (3, 2)), ("uniform2000000_s100", 2_000_000, 100, (3, 2, 1))):
        tris = pkg.meshgen.uniform(n, seed)
        e = {"n": n, "generator": f"meshgen.uniform({n}, {seed})"}
        for algo in algos:
            t0 = time.time(); t = orc.build_tree(algo, tris); dt = time.time() - t0
            tag = {1: "lbvh_single", 2: "ploc", 3: "hploc"}[algo]
            lay = t["layout"]
            assert orc.validate_bvh2(t["nodes"], t.get("leaves"), t["root"], n, lay) == 0
            e[tag] = {"topology": "%016x" % orc.topology_hash(t["nodes"], t.get("leaves"), t["root"], n, lay), "sah_f64": orc.sah_bvh2(t["nodes"], t.get("leaves"), t["root"], n, lay)[0],
                      "nodes_fnv": "%016x" % orc.fnv1a(t["nodes"]), "root": int(t["root"])}
            if t.get("leaves") is not None:
                e[tag]["leaves_fnv"] = "%016x" % orc.fnv1a(t["leaves"])
            if "stats" in t:
                e[tag]["stats"] = {k: int(v) for k, v in t["stats"].items()}
            print(name, tag, "%.1f s" % dt, e[tag], flush=True)
        e["sorted_keys_fnv"] = "%016x" % orc.fnv1a(t["skeys"]); e["sorted_vals_fnv"] = "%016x" % orc.fnv1a(t["svals"])
        out[name] = e
    res["_fullsize"] = out
    json.dump(res, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "fullsize":
        make_fullsize()
    elif len(sys.argv) >= 2 and sys.argv[1] == "meshes":
        make_meshes()
    elif len(sys.argv) >= 2 and sys.argv[1] == "ploc":
        make_ploc()
    elif len(sys.argv) >= 3 and sys.argv[1] == "ploc_hw":
        make_ploc_hw(sys.argv[2])
    elif len(sys.argv) >= 3 and sys.argv[1] == "merge_ploc_hw":
        merge_ploc_hw(sys.argv[2])
    elif len(sys.argv) >= 3 and sys.argv[1] == "reference":
        make_reference(sys.argv[2])
    else:
        raise SystemExit(__doc__)
