"""The reference's CalculateSceneExtents, Ploc / SinglePassPloc and both CollapseToWide4Bvh kernels executed ON THE MI355X (round 5).

The reference's `WarpSize` is a compile-time constant selected by its own gfx9 list (src/Common.h:100-106); `-D__gfx90a__=1` picks its wave64
flavour, and the UNMODIFIED src/CommonBlocksKernel.h / src/Ploc++Kernel.h compile for gfx950 that way (oracle/Makefile `_ref/%.w64.co`).  Driven by
oracle/ref_driver.cpp in the reference's host order (src/PLOC++Bvh.cpp:19-37, :82-152, :154-184; src/TwoPassLbvh.cpp:154-183), they pin on silicon
what rounds 2-4 pinned only through the CPU SIMT emulator: PLOC++ (a8), CalculateSceneExtents (a2) and the collapse (a11).  With this file every
(a) row except the sort (Orochi is not in the reference tree) is pinned by the reference's own kernels run on the target."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, require_ref

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


@pytest.fixture(scope="module")
def drv(orc):
    require_ref(os.path.exists(orc.REF_DRIVER), "oracle/_ref/libref_driver.so (the reference's kernels)")
    require_ref(os.path.exists(os.path.join(os.path.dirname(orc.REF_DRIVER), "Ploc++Kernel.w64.nofma.co")), "oracle/_ref/Ploc++Kernel.w64.nofma.co (the reference's wave64 build)")
    return orc


def mesh(pkg, kind, n, seed):
    mg = pkg.meshgen
    if kind == "cornell":
        return mg.load_tri(os.path.join(GOLDEN, f"cornell{n}.tri"))
    if kind == "probe":
        return mg.probe_mesh(n)                                  # SURVEY.md Appendix A.4 (std::mt19937(1234)): 5 k -> 42.2733 / 8, 50 k -> 105.958 / 18
    if kind == "dups":
        t = mg.uniform(n, seed); t[::3] = t[1]; return t          # heavy duplicate keys: area ties everywhere
    if kind == "flat":
        t = mg.uniform(n, seed)
        for v in ("v1", "v2", "v3"):
            t[v][:, 2] = 0.25                                     # zero extent in z
        return t
    return {"uniform": lambda: mg.uniform(n, seed), "sponza": lambda: mg.sponza_like(n, seed), "bunny": lambda: mg.bunny_like(n, seed)}[kind]()


EXT_MESHES = [("cornell", 32, 0), ("cornell", 382, 0), ("uniform", 2, 1), ("uniform", 255, 2), ("uniform", 257, 3), ("uniform", 50_000, 22), ("flat", 2000, 8),
              ("sponza", 262_144, 3), ("bunny", 150_000, 2), ("uniform", 1_000_000, 1)]


@pytest.mark.parametrize("kind,n,seed", EXT_MESHES)
def test_reference_extents_kernel(pkg, orc, drv, ctx, kind, n, seed):
    """CalculateSceneExtents (src/CommonBlocksKernel.h:92-114) on the MI355X == oracle == product: per-primitive boxes and the scene box, byte for byte
    (a min / max reduction is order independent, so the float atomics of src/Common.h:291-307 must land on the same 24 bytes)."""
    tris = mesh(pkg, kind, n, seed); n = len(tris)
    boxes_ref, scene_ref = orc.ref_extents(tris)
    boxes, scene = orc.prim_bounds(tris)
    assert boxes_ref.tobytes() == boxes.tobytes(), "CPU oracle boxes != reference CalculateSceneExtents"
    assert scene_ref.tobytes() == scene.tobytes(), "CPU oracle scene extent != reference CalculateSceneExtents"
    L = pkg.lib()
    d_tris = ctx.upload(np.ascontiguousarray(tris)); d_box = ctx.alloc(n * 24); d_scene = ctx.alloc(32)
    assert L.bvh_stage_extents(ctx.handle, d_tris.ptr, n, d_box.ptr, d_scene.ptr) == 0
    assert d_box.download(pkg.AABB, n).tobytes() == boxes_ref.tobytes()
    assert d_scene.download(pkg.AABB, 1).tobytes() == scene_ref.tobytes()


@pytest.mark.parametrize("kind,n,seed", [("cornell", 382, 0), ("uniform", 257, 3), ("flat", 2000, 8), ("sponza", 262_144, 3), ("bunny", 150_000, 2)])
def test_reference_primref_frontend(pkg, orc, drv, ctx, kind, n, seed):
    """the PrimRef flavours of the LBVH front end — CalculatePrimRefExtents (src/CommonBlocksKernel.h:116-137) and CalculateMortonCodesPrimRef
    (:387-398), what src/TwoPassLbvh.cpp:48,66 and src/SinglePassLbvh.cpp:49,67 launch — on the MI355X, fed with the PrimRef array Utility::doEarlySplitClipping leaves
    when nothing is split (src/Utility.cpp:456-477): scene extent and Morton keys == the product's front end (stage E + stage M), byte for byte, at config 2's and
    config 1's sizes.  (This implementation keeps the PrimRefs on the device: SURVEY.md §8(d).)"""
    tris = mesh(pkg, kind, n, seed); n = len(tris)
    boxes, scene_o = orc.prim_bounds(tris)
    refs = np.zeros(n, dtype=orc.PRIMREF); refs["prim"] = np.arange(n, dtype=np.uint32); refs["min"] = boxes["min"]; refs["max"] = boxes["max"]
    scene_ref, keys_ref, vals_ref = orc.ref_primref_frontend(refs)
    assert scene_ref.tobytes() == scene_o.tobytes(), "CPU oracle scene extent != reference CalculatePrimRefExtents"
    assert np.array_equal(vals_ref, np.arange(n, dtype=np.uint32))
    L = pkg.lib()
    d_tris = ctx.upload(np.ascontiguousarray(tris)); d_box = ctx.alloc(n * 24); d_scene = ctx.alloc(32); d_keys = ctx.alloc(n * 4); d_vals = ctx.alloc(n * 4)
    assert L.bvh_stage_extents(ctx.handle, d_tris.ptr, n, d_box.ptr, d_scene.ptr) == 0
    assert L.bvh_stage_morton(ctx.handle, d_box.ptr, n, d_scene.ptr, d_keys.ptr, d_vals.ptr) == 0
    assert d_scene.download(pkg.AABB, 1).tobytes() == scene_ref.tobytes()
    assert np.array_equal(d_keys.download(np.uint32, n), keys_ref), "product Morton keys != reference CalculateMortonCodesPrimRef"
    assert np.array_equal(d_vals.download(np.uint32, n), vals_ref)
    # ... and the build path's own keys (the Morton kernel fused with the sort's histograms)
    b = pkg.SinglePassLbvh().build(ctx, tris)
    got = b.download()
    order = np.argsort(keys_ref, kind="stable")
    assert np.array_equal(got["sorted_keys"], keys_ref[order]) and np.array_equal(got["sorted_vals"], order.astype(np.uint32))


PLOC_MESHES = [("cornell", 32, 0), ("cornell", 82, 0), ("cornell", 382, 0), ("probe", 5000, 0), ("probe", 50_000, 0),
               ("uniform", 2, 1), ("uniform", 3, 2), ("uniform", 1023, 31), ("uniform", 1024, 32), ("uniform", 1025, 33), ("uniform", 2049, 34), ("uniform", 4097, 5),
               ("dups", 3000, 9), ("flat", 2000, 8), ("sponza", 40_000, 3), ("bunny", 30_000, 2),
               ("sponza", 262_144, 3)]                            # BASELINE.json config 4's own size


@pytest.mark.parametrize("kind,n,seed", PLOC_MESHES)
@pytest.mark.parametrize("nofma", [True, False])
def test_reference_ploc_kernels(pkg, orc, drv, ctx, kind, n, seed, nofma):
    """SetupClusters + Ploc + SinglePassPloc (src/Ploc++Kernel.h:39-362) on the MI355X under the reference's host loop.
    nofma=True (contraction off: the oracle's area bit patterns): leaves byte-identical, same canonical topology, same number of host-loop iterations, SAH equal
    up to the f64 summation order — reference == oracle == product.  Node NUMBERING is schedule dependent in the reference (global atomicAdd per wave,
    :57-68) and is not compared.  nofma=False (as hiprtc builds it): ties may break differently; SAH within 1e-4 (north_star's bar)."""
    tris = mesh(pkg, kind, n, seed); n = len(tris)
    fe = orc.front_end(tris)
    nodes, leaves, iters = orc.ref_ploc(fe["boxes"], fe["svals"], nofma=nofma)
    assert orc.validate_bvh2(nodes, leaves, 0, n, 1) == 0, "the reference's own tree is invalid"
    onodes, oleaves, ostats = orc.ploc(fe["boxes"], fe["svals"])
    s_ref = orc.sah_bvh2(nodes, leaves, 0, n, 1)[0]
    s_orc = orc.sah_bvh2(onodes, oleaves, 0, n, 1)[0]
    b = pkg.PLOCNew().build(ctx, tris)
    assert abs(s_ref - s_orc) <= 1e-4 * s_orc
    assert abs(b.sah_cost() - s_ref) <= 1e-4 * s_ref, "product SAH vs the reference's Ploc kernels"
    if nofma:
        assert leaves.tobytes() == oleaves.tobytes()
        assert iters == ostats["iterations"] == b.timings.ploc_iterations
        t_ref = orc.topology_hash(nodes, leaves, 0, n, 1)
        assert t_ref == orc.topology_hash(onodes, oleaves, 0, n, 1), "CPU oracle topology != reference Ploc on the MI355X"
        g = b.download()
        assert g["leaves"].tobytes() == leaves.tobytes()
        assert orc.topology_hash(g["nodes"], g["leaves"], 0, n, 1) == t_ref, "product topology != reference Ploc on the MI355X"
        assert s_ref == pytest.approx(s_orc, rel=1e-12)


@pytest.mark.parametrize("kind,n,seed", [m for m in PLOC_MESHES if m[1] <= 50_000])
def test_emulator_matches_hardware_on_ploc(pkg, orc, drv, kind, n, seed):
    """the CPU SIMT emulator (wave32 flavour of the same header, one inserted barrier) against the silicon (wave64 flavour, unmodified): the wave size and the
    arrival order change the numbering only — same leaves, same iteration count, same canonical topology."""
    require_ref(os.path.exists(orc.REF_PLOC_EMU), "oracle/_ref/libref_ploc_emu.so")
    tris = mesh(pkg, kind, n, seed); n = len(tris)
    fe = orc.front_end(tris)
    e_nodes, e_leaves, e_iters = orc.ref_emu_ploc(fe["boxes"], fe["svals"])
    h_nodes, h_leaves, h_iters = orc.ref_ploc(fe["boxes"], fe["svals"], nofma=True)
    assert e_iters == h_iters and e_leaves.tobytes() == h_leaves.tobytes()
    assert orc.topology_hash(e_nodes, e_leaves, 0, n, 1) == orc.topology_hash(h_nodes, h_leaves, 0, n, 1), "emulated Ploc != Ploc on the MI355X"


def test_ploc_hardware_goldens(pkg, orc, drv):
    """tests/golden/reference_outputs.json `_ploc_hw` (written on the MI355X by tools/make_golden.py ploc_hw): the live kernels reproduce the committed outputs"""
    gold = json.load(open(os.path.join(GOLDEN, "reference_outputs.json"))).get("_ploc_hw")
    if not gold:
        pytest.skip("no _ploc_hw goldens committed yet")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(GOLDEN), "..", "tools", "make_golden.py"))
    mgold = importlib.util.module_from_spec(spec); spec.loader.exec_module(mgold)
    meshes = mgold.ploc_hw_meshes(pkg)
    for name, want in gold.items():
        assert mgold.ploc_hw_entry(orc, meshes[name]) == want, name


COLLAPSE_MESHES = [("cornell", 32, 0), ("cornell", 382, 0), ("uniform", 2, 1), ("uniform", 3, 2), ("uniform", 33, 23), ("uniform", 1000, 21), ("dups", 3000, 9),
                   ("uniform", 50_000, 22), ("bunny", 150_000, 2), ("sponza", 262_144, 3)]


@pytest.mark.parametrize("kind,n,seed", COLLAPSE_MESHES)
@pytest.mark.parametrize("algo", [1, 2, 3])
def test_reference_collapse_kernels(pkg, orc, drv, ctx, kind, n, seed, algo):
    """CollapseToWide4Bvh on the MI355X (LBVH flavour src/TwoPassLbvhKernel.h:237-336 on the single-pass tree; PLOC flavour src/Ploc++Kernel.h:364-465 on the
    PLOC++ and HPLOC trees; ceil(2n/3) threads, all resident) applied to the PRODUCT's BVH2 == the oracle's restatement == the product's level-synchronous
    collapse: same number of wide nodes, every leaf placed, same canonical wide topology, same BVH4 cost (the reference's own calculatebvh4Cost on each)."""
    tris = mesh(pkg, kind, n, seed); n = len(tris)
    b = pkg.BUILDERS[algo]().build(ctx, tris)
    got = b.download()
    r_wide, r_prims, r_total, r_placed = orc.ref_collapse(got["nodes"], got["leaves"], got["root"], n, got["layout"])
    assert r_placed == n, "the reference's kernel did not place every leaf"
    o_wide, o_prims, o_total = orc.collapse4(got["nodes"], got["leaves"], got["root"], n, got["layout"])
    p_wide, p_prims, p_total = b.collapse4()
    assert r_total == o_total == p_total
    h_ref = orc.topology_hash4(r_wide, r_prims, r_total, n)
    assert h_ref != 0
    assert h_ref == orc.topology_hash4(o_wide, o_prims, o_total, n), "CPU oracle wide topology != reference CollapseToWide4Bvh on the MI355X"
    assert h_ref == orc.topology_hash4(p_wide, p_prims, p_total, n), "product wide topology != reference CollapseToWide4Bvh on the MI355X"
    boxes, _ = orc.prim_bounds(tris)
    c_ref = orc.sah_bvh4(np.ascontiguousarray(r_wide), r_prims, boxes, r_total, n)[0]
    assert c_ref == pytest.approx(orc.sah_bvh4(o_wide, o_prims, boxes, o_total, n)[0], rel=1e-9)
    assert c_ref == pytest.approx(orc.sah_bvh4(p_wide, p_prims, boxes, p_total, n)[0], rel=1e-9)
    assert b.collapse4_cost()[0] == pytest.approx(c_ref, rel=1e-6)          # bvh_bvh4_cost: the reference's m_cost (device reduction)
    R = orc.ref_utility()
    require_ref(R is not None, "oracle/_ref/libref_utility.so (the reference's Utility.cpp)")
    rw = np.ascontiguousarray(r_wide); rp = np.ascontiguousarray(r_prims); pw = np.ascontiguousarray(p_wide); pp = np.ascontiguousarray(p_prims)
    # the reference's host check of its own kernel's output and of the product's (f32 accumulation in node-index order: equal up to that loop's rounding)
    c1 = R.ref_calculatebvh4Cost(rw.ctypes.data, rp.ctypes.data, boxes.ctypes.data, 0, r_total, n - 1)
    c2 = R.ref_calculatebvh4Cost(pw.ctypes.data, pp.ctypes.data, boxes.ctypes.data, 0, p_total, n - 1)
    assert c1 == pytest.approx(c2, rel=5e-3) and c1 == pytest.approx(c_ref, rel=5e-3)
