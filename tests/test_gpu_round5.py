"""Round-5 GPU tests: Morton codes beyond the 30-bit budget (the build's narrow top sort pass), non-finite and extreme inputs through the product AND
the reference's kernels."""
import os

import numpy as np
import pytest

from conftest import require_ref

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


def _stretched(pkg, n, ext, seed):
    """uniform(n) mapped into the box [0, ext): planar / needle scenes whose axis ratios exceed 2^32"""
    t = pkg.meshgen.uniform(n, seed)
    for v in ("v1", "v2", "v3"):
        for a in range(3):
            t[v][:, a] = (t[v][:, a] % np.float32(1.0)).astype(np.float32) * np.float32(ext[a])
    return t


WIDE_EXTENTS = [(0.0, 9.1e-13, 1.0), (7e-43, 4.9e-4, 1.07e9), (1.0, 1e-10, 0.0)]


@pytest.mark.parametrize("n", [5000, 1_200_000])            # both sort tile shapes (narrow below 1 M keys, wide above)
@pytest.mark.parametrize("ext", WIDE_EXTENTS)
def test_codes_beyond_30_bits_sort_by_all_32(pkg, orc, ctx, ext, n):
    """The extended Morton code is the reference's unsigned wrap-around arithmetic (src/CommonBlocksKernel.h:252-356): on a planar scene with an axis ratio >= 2^32
    it leaves bits 30 / 31 set, and the reference sorts all 32 bits (src/Hploc.cpp:63-81).  Round 4's build sorted bits [0, 30) only: colliding scatter destinations,
    unwritten output slots.  Now the narrow top pass is gated on a device word the Morton kernel raises, and the full-width pass behind it takes over."""
    tris = _stretched(pkg, n, ext, 3)
    boxes, scene = orc.prim_bounds(tris)
    keys, _ = orc.morton_codes(boxes, scene)
    assert np.count_nonzero(keys >> np.uint32(30)) > 0, "scene construction: no code beyond 30 bits"
    if n <= 100_000:
        require_ref(os.path.exists(orc.REF_DRIVER), "oracle/_ref/libref_driver.so (the reference's kernels)")
        kref, _ = orc.ref_morton(boxes, scene)
        assert np.array_equal(kref, keys), "CPU oracle != reference CalculateMortonCodes on a degenerate extent"
    order = np.argsort(keys, kind="stable").astype(np.uint32)
    for algo in (0, 1, 2, 3):
        if algo == 2 and n > 100_000:
            continue                                                      # (PLOC++ on a planar needle scene takes thousands of iterations)
        b = pkg.BUILDERS[algo]().build(ctx, tris)
        got = b.download()
        assert np.array_equal(got["sorted_keys"], keys[order]), f"algo {algo}: keys not sorted by all 32 bits"
        assert np.array_equal(got["sorted_vals"], order), f"algo {algo}: not the stable order"
        assert orc.validate_bvh2(got["nodes"], got["leaves"], got["root"], n, got["layout"]) == 0
        if algo in (0, 1) and n <= 100_000:
            ref = orc.build_tree(algo, tris)
            assert got["root"] == ref["root"] and got["nodes"].tobytes() == ref["nodes"].tobytes()
        if algo == 3 and n <= 100_000:
            ref = orc.build_tree(3, tris)
            assert orc.topology_hash(got["nodes"], got["leaves"], 0, n, 1) == orc.topology_hash(ref["nodes"], ref["leaves"], 0, n, 1)
    # a regular scene on the same context afterwards: the gate word is per build
    tris = pkg.meshgen.uniform(n, 5)
    ref = orc.front_end(tris)
    got = pkg.SinglePassLbvh().build(ctx, tris).download()
    assert np.array_equal(got["sorted_keys"], ref["skeys"]) and np.array_equal(got["sorted_vals"], ref["svals"])


@pytest.mark.parametrize("ext", WIDE_EXTENTS)
def test_codes_beyond_60_bits_sort_by_all_64(pkg, orc, ctx, ext):
    n = 5000
    tris = _stretched(pkg, n, ext, 3)
    d_tris = ctx.upload(tris)
    got = pkg.SinglePassLbvh().build_ex(ctx, n, tris=d_tris, morton_bits=60).download()
    ref = orc.build_tree(1, tris, morton_bits=60)
    assert np.array_equal(got["sorted_keys"], ref["skeys"]) and np.array_equal(got["sorted_vals"], ref["svals"])
    assert got["nodes"].tobytes() == ref["nodes"].tobytes()


# ---- non-finite and extreme inputs ---------------------------------------------------------------------------------------------
def _special(pkg, kind):
    t = pkg.meshgen.uniform(4000, 77)
    if kind == "inf_vertex":
        t["v2"][1234, 0] = np.float32(np.inf)
    elif kind == "neg_inf_vertex":
        t["v1"][17, 1] = np.float32(-np.inf)
    elif kind == "huge_triangle":
        t["v1"][100] = (-3e38, -3e38, -3e38); t["v2"][100] = (3e38, 3e38, 3e38); t["v3"][100] = (0, 3e38, -3e38)
    elif kind == "huge_offset":
        for v in ("v1", "v2", "v3"):
            t[v][:, 0] += np.float32(3e38)                          # extent 0 in x after rounding
    elif kind == "nan_coordinate":
        t["v3"][2000, 2] = np.float32(np.nan)                        # (positive sign bit: numpy's default NaN)
    elif kind == "nan_vertex":
        t["v1"][5] = (np.nan, np.nan, np.nan)
    elif kind == "neg_nan_coordinate":
        t["v2"].view(np.uint32)[3000, 1] = 0xFFC00000                 # NaN with the sign bit set
    elif kind == "ff_filled_triangle":
        for v in ("v1", "v2", "v3"):
            t[v].view(np.uint32)[777] = 0xFFFFFFFF                    # a triangle out of a 0xFF-filled buffer (negative NaNs with a full payload)
    elif kind == "denormals":
        for v in ("v1", "v2", "v3"):
            t[v][:] = (t[v] * np.float32(1e-41)).astype(np.float32)
    else:
        raise KeyError(kind)
    return t


SPECIALS = ["inf_vertex", "neg_inf_vertex", "huge_triangle", "huge_offset", "nan_coordinate", "nan_vertex", "neg_nan_coordinate", "ff_filled_triangle", "denormals"]


@pytest.mark.parametrize("kind", SPECIALS)
def test_non_finite_inputs_against_the_reference_kernels(pkg, orc, ctx, kind):
    """±inf / ±3e38 / NaN / denormal coordinates through the product and through the reference's own kernels on the MI355X (contraction off): boxes and scene extent,
    Morton keys, both LBVH node arrays byte for byte, HPLOC and PLOC++ leaves + canonical topology.  The emit kernels are built -fno-honor-nans -mno-amdgpu-ieee
    (csrc/Makefile): this is the test that the flags change nothing observable.  Divergences found are listed in DESIGN.md section 4."""
    require_ref(os.path.exists(orc.REF_DRIVER), "oracle/_ref/libref_driver.so (the reference's kernels)")
    tris = _special(pkg, kind); n = len(tris)
    L = pkg.lib()
    # E
    boxes_ref, scene_ref = orc.ref_extents(tris, nofma=True)
    d_tris = ctx.upload(np.ascontiguousarray(tris)); d_box = ctx.alloc(n * 24); d_scene = ctx.alloc(32); d_keys = ctx.alloc(n * 4)
    assert L.bvh_stage_extents(ctx.handle, d_tris.ptr, n, d_box.ptr, d_scene.ptr) == 0
    assert d_box.download(pkg.AABB, n).tobytes() == boxes_ref.tobytes(), "boxes != reference CalculateSceneExtents"
    assert d_scene.download(pkg.AABB, 1).tobytes() == scene_ref.tobytes(), "scene extent != reference CalculateSceneExtents"
    # M
    keys_ref, _ = orc.ref_morton(boxes_ref, scene_ref, nofma=True)
    assert L.bvh_stage_morton(ctx.handle, d_box.ptr, n, d_scene.ptr, d_keys.ptr, None) == 0
    keys = d_keys.download(np.uint32, n)
    assert np.array_equal(keys, keys_ref), f"{np.count_nonzero(keys != keys_ref)} Morton keys != reference CalculateMortonCodes"
    order = np.argsort(keys_ref, kind="stable").astype(np.uint32)
    skeys = keys_ref[order]
    # B, LBVH
    g1 = pkg.SinglePassLbvh().build(ctx, tris).download()
    assert np.array_equal(g1["sorted_keys"], skeys) and np.array_equal(g1["sorted_vals"], order)
    nodes1, root1 = orc.ref_lbvh_single(tris, skeys, order, nofma=True)
    assert g1["root"] == root1 and g1["nodes"].tobytes() == nodes1.tobytes(), "single-pass LBVH != reference kernels"
    g0 = pkg.TwoPassLbvh().build(ctx, tris).download()
    nodes0 = orc.ref_lbvh_two(tris, skeys, order, nofma=True)
    assert g0["nodes"].tobytes() == nodes0.tobytes(), "two-pass LBVH != reference kernels"
    # B, HPLOC and PLOC++
    cover_all = (n - 1) % 32 == 0
    h_nodes, h_leaves, merged = orc.ref_hploc(boxes_ref, skeys, order, nofma=True, cover_all=cover_all)
    assert merged == n - 1
    g3 = pkg.HPLOC().build(ctx, tris).download()
    assert g3["leaves"].tobytes() == h_leaves.tobytes()
    assert orc.validate_bvh2(g3["nodes"], g3["leaves"], 0, n, 1) == 0
    assert orc.topology_hash(g3["nodes"], g3["leaves"], 0, n, 1) == orc.topology_hash(h_nodes, h_leaves, 0, n, 1), "HPLOC topology != reference HPloc kernel"
    p_nodes, p_leaves, p_iters = orc.ref_ploc(boxes_ref, order, nofma=True)
    b2 = pkg.PLOCNew().build(ctx, tris); g2 = b2.download()
    assert g2["leaves"].tobytes() == p_leaves.tobytes()
    assert orc.validate_bvh2(g2["nodes"], g2["leaves"], 0, n, 1) == 0
    assert orc.topology_hash(g2["nodes"], g2["leaves"], 0, n, 1) == orc.topology_hash(p_nodes, p_leaves, 0, n, 1), "PLOC++ topology != reference Ploc kernels"
    assert b2.timings.ploc_iterations == p_iters
    # and the tile schedulers (the bench's kernels) on the same input
    with ctx.options(hploc="block", lbvh="block"):
        g3b = pkg.HPLOC().build(ctx, tris).download()
        assert orc.topology_hash(g3b["nodes"], g3b["leaves"], 0, n, 1) == orc.topology_hash(h_nodes, h_leaves, 0, n, 1), "HPLOC (tile scheduler) != reference HPloc kernel"
        g1b = pkg.SinglePassLbvh().build(ctx, tris).download()
        assert g1b["root"] == root1 and g1b["nodes"].tobytes() == nodes1.tobytes(), "single-pass LBVH (tile scheduler) != reference kernels"
