"""GPU parity of the widened rows, through the C ABI:

* triangle input formats (packed 36-byte, indexed): stage E and the whole tree are byte-identical to the 64-byte padded path;
* 60-bit Morton codes in u64 keys: the 64-bit encoder with a 30-bit budget reproduces the 30-bit codes (the pin: the reference
  has no 60-bit code), with a 60-bit budget it equals the oracle's restatement; the u64 one-sweep sort is stable; every builder on
  60-bit keys equals the oracle built on the same keys (LBVH / PLOC++ byte-exact, HPLOC canonical topology + SAH within 1e-4)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def packed36(tris):
    return np.ascontiguousarray(np.concatenate([tris["v1"], tris["v2"], tris["v3"]], axis=1).astype(np.float32))


def indexed(tris):
    """(vertices float32 (m,3), indices uint32 (n,3)) with shared vertices merged"""
    allv = np.concatenate([tris["v1"], tris["v2"], tris["v3"]], axis=0)
    verts, inv = np.unique(allv, axis=0, return_inverse=True)
    n = len(tris)
    idx = np.stack([inv[:n], inv[n:2 * n], inv[2 * n:]], axis=1).astype(np.uint32)
    return np.ascontiguousarray(verts.astype(np.float32)), np.ascontiguousarray(idx)


def _meshes(pkg):
    mg = pkg.meshgen
    return {"uniform_3001": mg.uniform(3001, 21), "bunny_40k": mg.bunny_like(40_000, 2), "sponza_70k": mg.sponza_like(70_001, 3),
            "uniform_2": mg.uniform(2, 22), "uniform_257": mg.uniform(257, 23)}


@pytest.mark.parametrize("name", ["uniform_2", "uniform_257", "uniform_3001", "bunny_40k", "sponza_70k"])
def test_triangle_formats_stage_e(pkg, orc, ctx, name):
    tris = _meshes(pkg)[name]; n = len(tris)
    boxes, scene = orc.prim_bounds(tris)
    L = pkg.lib()
    d_box = ctx.alloc(n * 24); d_scene = ctx.alloc(32)
    pk = packed36(tris); verts, idx = indexed(tris)
    d_pk = ctx.upload(pk); d_v = ctx.upload(verts); d_i = ctx.upload(idx)
    for inp in (pkg.BuildInput(pkg.TRI_PACKED36, 30, d_pk.ptr, None, None, 0, 0),
                pkg.BuildInput(pkg.TRI_INDEXED, 30, None, d_v.ptr, d_i.ptr, len(verts), 0)):
        assert L.bvh_stage_extents_ex(ctx.handle, C.byref(inp), n, d_box.ptr, d_scene.ptr) == 0
        assert d_box.download(pkg.AABB, n).tobytes() == boxes.tobytes()
        assert d_scene.download(pkg.AABB, 1).tobytes() == scene.tobytes()
    # argument validation: misaligned packed pointer, missing index buffer, unknown format
    bad = pkg.BuildInput(pkg.TRI_PACKED36, 30, d_pk.ptr + 4, None, None, 0, 0)
    assert L.bvh_stage_extents_ex(ctx.handle, C.byref(bad), n, d_box.ptr, d_scene.ptr) == -10001
    bad = pkg.BuildInput(pkg.TRI_INDEXED, 30, None, d_v.ptr, None, len(verts), 0)
    assert L.bvh_stage_extents_ex(ctx.handle, C.byref(bad), n, d_box.ptr, d_scene.ptr) == -10001
    bad = pkg.BuildInput(7, 30, d_pk.ptr, None, None, 0, 0)
    assert L.bvh_stage_extents_ex(ctx.handle, C.byref(bad), n, d_box.ptr, d_scene.ptr) == -10001


@pytest.mark.parametrize("algo", [0, 1, 2, 3])
def test_triangle_formats_same_tree(pkg, ctx, algo):
    tris = _meshes(pkg)["bunny_40k"]; n = len(tris)
    ref = pkg.BUILDERS[algo]().build(ctx, tris).download()
    pk = packed36(tris); verts, idx = indexed(tris)
    assert len(verts) < 3 * n, "the bunny-like mesh shares vertices"
    d_pk = ctx.upload(pk); d_v = ctx.upload(verts); d_i = ctx.upload(idx)
    for kw in (dict(tris=d_pk, tri_format=pkg.TRI_PACKED36), dict(vertices=d_v, indices=d_i, n_vertices=len(verts), tri_format=pkg.TRI_INDEXED)):
        got = pkg.BUILDERS[algo]().build_ex(ctx, n, **kw).download()
        assert got["root"] == ref["root"] and np.array_equal(got["sorted_keys"], ref["sorted_keys"]) and np.array_equal(got["sorted_vals"], ref["sorted_vals"])
        if algo != 3:
            assert got["nodes"].tobytes() == ref["nodes"].tobytes()
        else:       # HPLOC numbering is topology-derived here, so even HPLOC is byte-identical across input formats
            assert got["nodes"].tobytes() == ref["nodes"].tobytes() and got["leaves"].tobytes() == ref["leaves"].tobytes()


def _dup_heavy(pkg):
    """many primitives per 30-bit cell: 30-bit keys collide heavily, 60-bit keys separate them"""
    t = pkg.meshgen.uniform(30_000, 31)
    for v in ("v1", "v2", "v3"):
        t[v][1:] = t[v][1:] * np.float32(1e-4) + np.float32(0.5)     # everything but triangle 0 inside a 1e-4 cube of the unit scene
    return np.ascontiguousarray(t)


@pytest.mark.parametrize("name", ["uniform_257", "uniform_3001", "sponza_70k", "dups"])
def test_morton64_and_sort64(pkg, orc, ctx, name):
    tris = _dup_heavy(pkg) if name == "dups" else _meshes(pkg)[name]; n = len(tris)
    boxes, scene = orc.prim_bounds(tris)
    L = pkg.lib()
    d_box = ctx.upload(boxes); d_scene = ctx.upload(scene); d_k = ctx.alloc(n * 8)
    # 30-bit budget through the 64-bit encoder == the reference's 30-bit codes
    assert L.bvh_stage_morton64(ctx.handle, d_box.ptr, n, d_scene.ptr, d_k.ptr, 30) == 0
    k30, _ = orc.morton_codes(boxes, scene)
    assert np.array_equal(d_k.download(np.uint64, n), k30.astype(np.uint64))
    # 60-bit budget == oracle restatement
    assert L.bvh_stage_morton64(ctx.handle, d_box.ptr, n, d_scene.ptr, d_k.ptr, 60) == 0
    k60 = orc.morton_codes64(boxes, scene, 60)
    got = d_k.download(np.uint64, n)
    assert np.array_equal(got, k60), f"{np.count_nonzero(got != k60)} 60-bit keys differ"
    assert int(k60.max()) < 2**60
    if name == "dups":
        assert len(np.unique(k60)) > 4 * len(np.unique(k30)), "60-bit keys must separate what 30-bit keys merge"
    # u64 one-sweep sort: stable ascending, all 64 bits and a sub-range
    d_sk = ctx.alloc(n * 8); d_sv = ctx.alloc(n * 4)
    for lo, hi in ((0, 64), (0, 60), (8, 37)):
        assert L.bvh_sort_pairs64(ctx.handle, d_k.ptr, None, n, d_sk.ptr, d_sv.ptr, lo, hi) == 0
        ctx.synchronize()
        field = (k60 >> np.uint64(lo)) & np.uint64((1 << (hi - lo)) - 1) if hi - lo < 64 else k60
        order = np.argsort(field, kind="stable").astype(np.uint32)
        assert np.array_equal(d_sv.download(np.uint32, n), order)
        assert np.array_equal(d_sk.download(np.uint64, n), k60[order])
    assert L.bvh_sort_pairs64(ctx.handle, d_k.ptr, None, n, d_sk.ptr, d_sv.ptr, 3, 70) == -10001


@pytest.mark.parametrize("mode", ["async", "block"])
@pytest.mark.parametrize("algo", [0, 1, 2, 3])
@pytest.mark.parametrize("name", ["uniform_3001", "sponza_70k", "dups"])
def test_build_60bit_keys(pkg, orc, ctx, name, algo, mode, sched_opts):
    if mode == "block" and algo == 2:
        pytest.skip("scheduler choice only concerns HPLOC and the LBVH builders")
    sched_opts(hploc=mode, lbvh="block" if mode == "block" else "single")
    tris = _dup_heavy(pkg) if name == "dups" else _meshes(pkg)[name]; n = len(tris)
    d_tris = ctx.upload(tris)
    b = pkg.BUILDERS[algo]().build_ex(ctx, n, tris=d_tris, morton_bits=60)
    got = b.download()
    ref = orc.build_tree(algo, tris, morton_bits=60)
    assert got["sorted_keys"].dtype == np.uint64 and np.array_equal(got["sorted_keys"], ref["skeys"]) and np.array_equal(got["sorted_vals"], ref["svals"])
    assert orc.validate_bvh2(got["nodes"], got["leaves"], got["root"], n, got["layout"]) == 0
    if algo in (0, 1, 2):
        assert got["root"] == ref["root"] and got["nodes"].tobytes() == ref["nodes"].tobytes()
        if algo == 2:
            assert got["leaves"].tobytes() == ref["leaves"].tobytes()
    else:
        assert got["leaves"].tobytes() == ref["leaves"].tobytes()
        assert orc.topology_hash(got["nodes"], got["leaves"], 0, n, 1) == orc.topology_hash(ref["nodes"], ref["leaves"], 0, n, 1)
    s_ref = orc.sah_bvh2(ref["nodes"], ref["leaves"], ref["root"], n, ref["layout"])[0]
    s60 = b.sah_cost()                       # (a result aliases the ctx arena until the next build on the ctx)
    assert abs(s60 - s_ref) <= 1e-4 * s_ref
    if name == "dups" and algo in (0, 1):
        # what the longer keys buy: the LBVH over 60-bit keys is better than over colliding 30-bit keys
        s30 = pkg.BUILDERS[algo]().build(ctx, tris).sah_cost()
        assert s60 < s30
