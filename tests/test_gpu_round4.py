"""Round-4 GPU tests: the log2f boundary of the Morton bit plan, collapse hints across different same-size trees,
the batched builder's per-mesh outputs and its pipelined form."""
import ctypes as C
import itertools
import os

import numpy as np
import pytest

from conftest import require_ref

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


def _boundary_scene(pkg, perm, e, n=96, seed=0):
    """n triangles whose union is EXACTLY the box [0, e[0]] x [0, e[1]] x [0, e[2]] on the axes perm (two corner triangles pin the extent, the rest lie inside)"""
    rng = np.random.default_rng(seed)
    ext = np.zeros(3, np.float32); ext[list(perm)] = np.asarray(e, np.float32)
    tris = np.zeros(n, dtype=pkg.meshgen.TRIANGLE)
    c = rng.random((n, 3), dtype=np.float32) * 0.9 + 0.05
    for k, f in enumerate(("v1", "v2", "v3")):
        tris[f] = (c + (rng.random((n, 3), dtype=np.float32) - 0.5) * 0.05) * ext
        tris[f] = np.clip(tris[f], 0.0, ext)
    tris["v1"][0] = 0.0; tris["v2"][0] = 0.0; tris["v3"][0] = ext * np.float32(0.25)           # touches the minimum corner
    tris["v1"][1] = ext; tris["v2"][1] = ext; tris["v3"][1] = ext * np.float32(0.75)           # touches the maximum corner
    return tris


def _ulps(x, d):
    x = np.float32(x)
    for _ in range(abs(d)):
        x = np.nextafter(x, np.float32(np.inf if d > 0 else 0), dtype=np.float32)
    return x


def test_morton_plan_at_log2_boundaries(pkg, orc, ctx):
    """`numPrebits = (int)log2f(extent ratio)` (src/CommonBlocksKernel.h:175-248) is evaluated by the DEVICE's log2f in the reference and in the product, by the
    host's libm in the CPU oracle (SURVEY.md section 7 flagged it).  Scenes whose extent ratios are 2^k * (1 - 2 ulp .. 1 + 2 ulp), k = 1..12, on every axis order:
      * product == the reference's own CalculateMortonCodes kernel on the MI355X, key for key, on EVERY scene (that is the parity that counts);
      * the device's plan (bvh_stage_morton_plan) == the oracle's plan wherever host and device log2f truncate alike, and where they do not, the oracle's encoder
        driven by the device's plan reproduces the reference's keys: the log2f truncation is the ONLY difference."""
    require_ref(os.path.exists(orc.REF_DRIVER), "oracle/_ref/libref_driver.so (the reference's kernels)")
    L = pkg.lib()
    scenes = 0; plan_diff = []
    for perm in itertools.permutations(range(3)):
        for k in range(1, 13):
            for d in (-2, -1, 0, 1, 2):
                r = _ulps(np.float32(2.0) ** k, d)
                # family A: (r, 1, 1): the a0 / a1 and a0 / a2 ratios sit on the boundary; family B: (2 r, r, 1): a1 / a2 does
                for e in ((r, np.float32(1), np.float32(1)), (np.float32(2) * r, r, np.float32(1))):
                    tris = _boundary_scene(pkg, perm, e, seed=scenes); n = len(tris)
                    boxes, scene = orc.prim_bounds(tris)
                    got_e = scene.view(np.float32)[3:6] - scene.view(np.float32)[0:3]
                    assert np.array_equal(np.sort(got_e), np.sort(np.asarray(e, np.float32))), "scene construction"
                    keys_ref, _ = orc.ref_morton(boxes, scene)
                    d_box = ctx.upload(boxes); d_scene = ctx.upload(scene); d_keys = ctx.alloc(n * 4)
                    assert L.bvh_stage_morton(ctx.handle, d_box.ptr, n, d_scene.ptr, d_keys.ptr, None) == 0
                    keys_dev = d_keys.download(np.uint32, n)
                    assert np.array_equal(keys_dev, keys_ref), f"product != reference CalculateMortonCodes at extents {e} on axes {perm}"
                    plan = (C.c_int32 * 10)()
                    assert L.bvh_stage_morton_plan(ctx.handle, d_scene.ptr, 30, plan) == 0
                    plan_dev = list(plan)
                    po = orc.morton_plan(scene)
                    plan_orc = po["axis"] + po["bits"] + po["pre"] + [po["pre_sum"], po["swap"]]
                    if plan_dev == plan_orc:
                        assert np.array_equal(orc.morton_codes(boxes, scene)[0], keys_ref), f"oracle != reference at extents {e} on axes {perm} although the plans agree"
                    else:
                        plan_diff.append((perm, k, d, plan_dev, plan_orc))
                        assert plan_dev[0:3] == plan_orc[0:3], "axis order does not depend on log2f"
                        assert np.array_equal(orc.morton_codes_with_plan(boxes, scene, plan_dev), keys_ref), "oracle encoder with the device's plan != reference"
                    scenes += 1
    print(f"\n{scenes} boundary scenes: product == reference kernel on all; host libm and device log2f truncate differently on {len(plan_diff)}")
    for x in plan_diff[:8]:
        print("   axes %s  2^%d %+d ulp: device plan %s, host plan %s" % x)
    # host and device may disagree only ON the boundary (|d| <= 2 ulp was all that was generated); a disagreement elsewhere would have failed above


def test_collapse_hints_do_not_leak_between_different_trees_of_one_size(pkg, orc, ctx):
    """the collapse sizes its launches from the previous collapse of a tree of the same size.  Alternating different trees of one size (LBVH vs PLOC++ vs
    HPLOC layouts, a staircase vs a uniform scene) must leave every wide tree equal to the one a fresh context builds."""
    n = 60_000
    uni = pkg.meshgen.uniform(n, 11)
    stair = pkg.meshgen.uniform(n, 12)
    for f in ("v1", "v2", "v3"):
        stair[f] *= np.float32(0.001); stair[f][:, 0] += (np.float32(1.7) ** (np.arange(n) % 40)).astype(np.float32)     # clusters at exponentially spaced x: a deep, lopsided tree
    seq = [(pkg.SinglePassLbvh, uni), (pkg.PLOCNew, uni), (pkg.SinglePassLbvh, stair), (pkg.HPLOC, uni), (pkg.PLOCNew, stair), (pkg.SinglePassLbvh, uni), (pkg.HPLOC, stair)]
    fresh = {}
    for cls, mesh in seq:
        key = (cls.__name__, mesh is uni)
        if key not in fresh:
            c2 = pkg.Context(0)
            try:
                wide, prims, nw = cls().build(c2, mesh).collapse4()
                fresh[key] = (nw, orc.topology_hash4(wide, prims, nw, n))
            finally:
                c2.close()
    for rep in range(2):
        for cls, mesh in seq:
            wide, prims, nw = cls().build(ctx, mesh).collapse4()
            assert (nw, orc.topology_hash4(wide, prims, nw, n)) == fresh[(cls.__name__, mesh is uni)]
