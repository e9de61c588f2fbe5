"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar: bit-exact for keys / sort / LBVH node arrays / PLOC++ node arrays; HPLOC (schedule-dependent node numbering, as in the
reference) identical canonical topology + SAH within 1e-4 relative (north_star tolerance)."""
import numpy as np
import pytest

from conftest import require_ref

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]

SAH_RTOL = 1e-4   # BASELINE.json north_star: "match SAH cost within 1e-4 for PLOC variants"


def meshes(pkg):
    mg = pkg.meshgen
    return {
        "uniform_2": mg.uniform(2, 5), "uniform_3": mg.uniform(3, 6), "uniform_33": mg.uniform(33, 7), "uniform_65": mg.uniform(65, 8),
        "uniform_1000": mg.uniform(1000, 9), "uniform_4097": mg.uniform(4097, 10), "uniform_50k": mg.uniform(50_000, 11),
        "probe_5k": mg.probe_mesh(5000), "bunny_150k": mg.bunny_like(150_000, 2), "sponza_262k": mg.sponza_like(262_144, 3),
        "dups_3000": np.repeat(mg.uniform(300, 12), 10), "flat_2000": _flat(mg.uniform(2000, 13)),
    }


def _flat(t):
    t = t.copy()
    for v in ("v1", "v2", "v3"):
        t[v][:, 2] = 0.25
    return t


@pytest.fixture(scope="module")
def cases(pkg):
    return meshes(pkg)


def _dev(ctx, a):
    return ctx.upload(np.ascontiguousarray(a))


@pytest.mark.parametrize("name", ["uniform_2", "uniform_33", "uniform_1000", "uniform_50k", "bunny_150k", "sponza_262k", "dups_3000", "flat_2000"])
def test_stage_extents_and_morton(pkg, orc, ctx, cases, name):
    tris = cases[name]; n = len(tris)
    L = pkg.lib()
    d_tris = _dev(ctx, tris); d_box = ctx.alloc(n * 24); d_scene = ctx.alloc(32); d_keys = ctx.alloc(n * 4); d_vals = ctx.alloc(n * 4)
    assert L.bvh_stage_extents(ctx.handle, d_tris.ptr, n, d_box.ptr, d_scene.ptr) == 0
    assert L.bvh_stage_morton(ctx.handle, d_box.ptr, n, d_scene.ptr, d_keys.ptr, d_vals.ptr) == 0
    boxes, scene = orc.prim_bounds(tris)
    keys, vals = orc.morton_codes(boxes, scene)
    assert d_box.download(pkg.AABB, n).tobytes() == boxes.tobytes()
    assert d_scene.download(pkg.AABB, 1).tobytes() == scene.tobytes()
    got = d_keys.download(np.uint32, n)
    assert np.array_equal(got, keys), f"{np.count_nonzero(got != keys)} Morton keys differ"
    assert np.array_equal(d_vals.download(np.uint32, n), vals)


@pytest.mark.parametrize("n,bits", [(1, 32), (2, 32), (255, 32), (4096, 32), (4097, 32), (100_003, 32), (1_000_000, 32), (999_999, 32), (2_500_001, 21), (100_003, 30), (100_003, 13), (5000, 0),
                                    # round 4: narrow top digits run with their own digit width (6 / 7 bits: 64 / 128 digit threads) in both tile shapes
                                    (1_500_000, 30), (1_500_000, 31), (100_003, 31), (1_500_003, 26), (300_000, 17)])
def test_sort_pairs(pkg, orc, ctx, n, bits):
    rng = np.random.default_rng(n * 31 + bits)
    keys = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    if n > 1000:
        keys[::7] = keys[3]            # heavy duplicates: stability matters
    vals = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    L = pkg.lib()
    dk, dv, ok, ov = _dev(ctx, keys), _dev(ctx, vals), ctx.alloc(n * 4), ctx.alloc(n * 4)
    assert L.bvh_sort_pairs(ctx.handle, dk.ptr, dv.ptr, n, ok.ptr, ov.ptr, 0, bits) == 0
    mask = np.uint32((1 << bits) - 1) if bits < 32 else np.uint32(0xFFFFFFFF)
    order = np.argsort(keys & mask, kind="stable")
    assert np.array_equal(ok.download(np.uint32, n), keys[order])
    assert np.array_equal(ov.download(np.uint32, n), vals[order])
    assert np.array_equal(dk.download(np.uint32, n), keys)      # src untouched
    # (key, index) flavour
    assert L.bvh_sort_pairs(ctx.handle, dk.ptr, None, n, ok.ptr, ov.ptr, 0, bits) == 0
    assert np.array_equal(ov.download(np.uint32, n), order.astype(np.uint32))


ALL = ["uniform_2", "uniform_3", "uniform_33", "uniform_65", "uniform_1000", "uniform_4097", "uniform_50k", "probe_5k", "bunny_150k",
       "sponza_262k", "dups_3000", "flat_2000"]


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("algo", [0, 1])
def test_lbvh_bit_exact(pkg, orc, ctx, cases, name, algo):
    tris = cases[name]; n = len(tris)
    b = pkg.BUILDERS[algo]().build(ctx, tris)
    got = b.download()
    ref = orc.build_tree(algo, tris)
    assert np.array_equal(got["sorted_keys"], ref["skeys"])
    assert np.array_equal(got["sorted_vals"], ref["svals"])
    assert got["root"] == ref["root"]
    assert got["nodes"].tobytes() == ref["nodes"].tobytes(), "Bvh2Node[2n-1] differs from the oracle"
    assert orc.validate_bvh2(got["nodes"], None, got["root"], n, 0) == 0
    assert abs(b.sah_cost() - orc.sah_bvh2(ref["nodes"], None, ref["root"], n, 0)[0]) <= 1e-9 * max(1.0, b.m_cost)


@pytest.mark.parametrize("name", ALL)
def test_ploc_bit_exact(pkg, orc, ctx, cases, name):
    tris = cases[name]; n = len(tris)
    b = pkg.PLOCNew().build(ctx, tris)
    got = b.download()
    ref = orc.build_tree(2, tris)
    assert got["leaves"].tobytes() == ref["leaves"].tobytes()
    assert orc.validate_bvh2(got["nodes"], got["leaves"], 0, n, 1) == 0
    assert orc.topology_hash(got["nodes"], got["leaves"], 0, n, 1) == orc.topology_hash(ref["nodes"], ref["leaves"], 0, n, 1)
    assert got["nodes"].tobytes() == ref["nodes"].tobytes(), "deterministic numbering: node arrays must be identical"
    assert b.timings.ploc_iterations == ref["stats"]["iterations"]
    s_ref = orc.sah_bvh2(ref["nodes"], ref["leaves"], 0, n, 1)[0]
    assert abs(b.sah_cost() - s_ref) <= SAH_RTOL * s_ref


@pytest.mark.parametrize("name", ALL)
def test_hploc_topology_and_sah(pkg, orc, ctx, cases, name):
    tris = cases[name]; n = len(tris)
    b = pkg.HPLOC().build(ctx, tris)
    got = b.download()
    ref = orc.build_tree(3, tris)
    assert got["leaves"].tobytes() == ref["leaves"].tobytes()
    assert orc.validate_bvh2(got["nodes"], got["leaves"], 0, n, 1) == 0
    s_ref = orc.sah_bvh2(ref["nodes"], ref["leaves"], 0, n, 1)[0]
    s_got = orc.sah_bvh2(got["nodes"], got["leaves"], 0, n, 1)[0]
    assert abs(s_got - s_ref) <= SAH_RTOL * s_ref
    assert abs(b.sah_cost() - s_ref) <= SAH_RTOL * s_ref
    assert orc.topology_hash(got["nodes"], got["leaves"], 0, n, 1) == orc.topology_hash(ref["nodes"], ref["leaves"], 0, n, 1)


def test_layout_adapter(pkg, orc, ctx, cases):
    tris = cases["uniform_50k"]; n = len(tris)
    b = pkg.HPLOC().build(ctx, tris)
    got = b.download()
    conv = b.to_lbvh_layout()
    assert conv.tobytes() == orc.ploc_to_lbvh_layout(got["nodes"], got["leaves"]).tobytes()
    assert orc.validate_bvh2(conv, None, 0, n, 0) == 0


def test_repeat_builds_same_ctx(pkg, orc, ctx, cases):
    """arena reuse: builds of different sizes / algorithms on one ctx do not leak state into each other"""
    for name, algo in (("uniform_50k", 1), ("uniform_1000", 3), ("bunny_150k", 2), ("uniform_33", 0), ("uniform_50k", 3)):
        tris = cases[name]; n = len(tris)
        b = pkg.BUILDERS[algo]().build(ctx, tris); got = b.download(); ref = orc.build_tree(algo, tris)
        assert orc.topology_hash(got["nodes"], got["leaves"], got["root"], n, got["layout"]) == \
               orc.topology_hash(ref["nodes"], ref["leaves"], ref["root"], n, ref["layout"])


@pytest.mark.parametrize("name", ["uniform_2", "uniform_3", "uniform_33", "uniform_1000", "uniform_50k", "sponza_262k", "dups_3000"])
@pytest.mark.parametrize("algo", [1, 2, 3])
def test_collapse4(pkg, orc, ctx, cases, name, algo):
    """BVH2 -> BVH4 collapse (config 4's n-wide collapse): same wide topology and BVH4 SAH as the oracle's restatement of
    CollapseToWide4Bvh applied to the same BVH2; the reference's own validator / cost function accept it."""
    tris = cases[name]; n = len(tris)
    b = pkg.BUILDERS[algo]().build(ctx, tris)
    got = b.download()
    wide, prims, total = b.collapse4()
    ow, opn, ototal = orc.collapse4(got["nodes"], got["leaves"], got["root"], n, got["layout"])
    assert total == ototal
    hsh = orc.topology_hash4(wide, prims, total, n)
    assert hsh != 0 and hsh == orc.topology_hash4(ow, opn, ototal, n)
    assert np.array_equal(np.sort(prims["prim"]), np.arange(n, dtype=np.uint32))
    boxes, _ = orc.prim_bounds(tris)
    c_got = orc.sah_bvh4(wide, prims, boxes, total, n)[0]; c_orc = orc.sah_bvh4(ow, opn, boxes, ototal, n)[0]
    assert abs(c_got - c_orc) <= 1e-9 * c_orc
    R = orc.ref_utility()
    require_ref(R is not None, "oracle/_ref/libref_utility.so (the reference's Utility.cpp)")
    if R is not None:
        w = np.ascontiguousarray(wide); p = np.ascontiguousarray(prims)
        depth_ok = n <= 60_000                      # the reference validator's DFS stack is 64 entries
        if depth_ok:
            assert R.ref_checkLBvh4Correctness(w.ctypes.data, p.ctypes.data, 0, n - 1) == 1
        # f32 accumulation in node-index order, exactly as the reference does it (the f64 value differs by accumulated rounding)
        assert R.ref_calculatebvh4Cost(w.ctypes.data, p.ctypes.data, boxes.ctypes.data, 0, total, n - 1) == pytest.approx(orc.sah_bvh4(wide, prims, boxes, total, n)[1], rel=1e-6)


def test_batched_build_single_process(pkg, orc):
    """bvh_batched_build (C ABI): meshes sharded over the visible devices, roots all-gathered with RCCL"""
    import torch
    devs = tuple(range(torch.cuda.device_count()))
    meshes = [pkg.meshgen.uniform(3000 + 500 * m, 40 + m, offset=(float(m), 0.0, 0.0)) for m in range(5)]
    roots, ms = pkg.batched_build(meshes, pkg.ALGO_HPLOC, devs)
    for m, t in enumerate(meshes):
        _, scene = orc.prim_bounds(t)
        assert np.array_equal(roots[m], np.concatenate([scene["min"][0], scene["max"][0]]))
    assert (ms > 0).all()


@pytest.mark.parametrize("name", ["uniform_1000", "uniform_50k", "dups_3000"])
def test_stage_level_emitters(pkg, orc, ctx, cases, name):
    """the per-kernel C-ABI entry points (bvh_emit_*) fed with the oracle's sorted arrays, outputs into caller buffers"""
    import ctypes as C
    tris = cases[name]; n = len(tris)
    fe = orc.front_end(tris)
    L = pkg.lib()
    d_box, d_k, d_v = _dev(ctx, fe["boxes"]), _dev(ctx, fe["skeys"]), _dev(ctx, fe["svals"])
    d_nodes = ctx.alloc((2 * n - 1) * 32); d_leaves = ctx.alloc(n * 28)
    root = C.c_uint32()
    assert L.bvh_emit_lbvh_single(ctx.handle, d_box.ptr, d_k.ptr, d_v.ptr, n, d_nodes.ptr, C.byref(root)) == 0
    ref, oroot = orc.lbvh_single(tris, fe["skeys"], fe["svals"])
    assert root.value == oroot and d_nodes.download(pkg.BVH2_NODE, 2 * n - 1).tobytes() == ref.tobytes()
    assert L.bvh_emit_lbvh_two(ctx.handle, d_box.ptr, d_k.ptr, d_v.ptr, n, d_nodes.ptr) == 0
    ctx.synchronize()
    assert d_nodes.download(pkg.BVH2_NODE, 2 * n - 1).tobytes() == orc.lbvh_two(tris, fe["skeys"], fe["svals"])[0].tobytes()
    it = C.c_uint32()
    assert L.bvh_emit_ploc(ctx.handle, d_box.ptr, d_v.ptr, n, d_nodes.ptr, d_leaves.ptr, C.byref(it)) == 0
    pn, pl, st = orc.ploc(fe["boxes"], fe["svals"])
    assert d_nodes.download(pkg.BVH2_NODE, n - 1).tobytes() == pn.tobytes() and d_leaves.download(pkg.PRIMREF, n).tobytes() == pl.tobytes()
    assert it.value == st["iterations"]
    assert L.bvh_emit_hploc(ctx.handle, d_box.ptr, d_k.ptr, d_v.ptr, n, d_nodes.ptr, d_leaves.ptr) == 0
    ctx.synchronize()
    hn, hl, _ = orc.hploc(fe["boxes"], fe["skeys"], fe["svals"])
    gn, gl = d_nodes.download(pkg.BVH2_NODE, n - 1), d_leaves.download(pkg.PRIMREF, n)
    assert gl.tobytes() == hl.tobytes() and orc.topology_hash(gn, gl, 0, n, 1) == orc.topology_hash(hn, hl, 0, n, 1)
    # argument validation (no GPU work): the reference's kernels have no checks; the ABI returns BVH_E_INVALID_ARG
    assert L.bvh_emit_hploc(ctx.handle, None, d_k.ptr, d_v.ptr, n, d_nodes.ptr, d_leaves.ptr) == -10001
    assert L.bvh_sort_pairs(ctx.handle, d_k.ptr, None, n, d_k.ptr, d_v.ptr, 5, 40) == -10001


@pytest.mark.parametrize("mode", ["async", "block"])
def test_every_small_size(pkg, orc, ctx, mode, sched_opts):
    """n = 2..70 and a few sizes around the 16/32/64 thresholds and the block scheduler's 512-leaf tile, all four builders, both
    HPLOC schedulers (the block scheduler needs more than two tiles and hands smaller inputs to the asynchronous one)"""
    sched_opts(hploc=mode, lbvh="block" if mode == "block" else "single")     # single-pass LBVH: tile scheduler / one-launch kernel
    for n in list(range(2, 71)) + [127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 1026, 1535, 1536, 1537, 2047, 2048, 2049, 4097]:
        tris = pkg.meshgen.uniform(n, 1000 + n)
        for algo in ((0, 1, 2, 3) if mode == "async" else (0, 1, 3)):
            got = pkg.BUILDERS[algo]().build(ctx, tris).download(); ref = orc.build_tree(algo, tris)
            assert orc.validate_bvh2(got["nodes"], got["leaves"], got["root"], n, got["layout"]) == 0, (n, algo)
            if algo in (0, 1, 2):
                assert got["nodes"].tobytes() == ref["nodes"].tobytes() and got["root"] == ref["root"], (n, algo)
            else:
                assert orc.topology_hash(got["nodes"], got["leaves"], 0, n, 1) == orc.topology_hash(ref["nodes"], ref["leaves"], 0, n, 1), (n, algo)


@pytest.mark.parametrize("sched", ["default", "block"])
@pytest.mark.parametrize("kind", ["identical", "two_far_clusters", "line", "point_cloud_dups", "huge_and_tiny", "staircase", "denormal_areas"])
@pytest.mark.parametrize("algo", [0, 1, 2, 3])
def test_degenerate_distributions(pkg, orc, ctx, kind, algo, sched, sched_opts):
    if sched == "block":
        if algo == 2:
            pytest.skip("tile schedulers exist for the LBVH builders and HPLOC")
        sched_opts(hploc="block", lbvh="block")
    mg = pkg.meshgen
    if kind == "identical":            # every Morton key equal: the hierarchy comes from the position bits only
        tris = np.repeat(mg.uniform(1, 3), 3000)
    elif kind == "two_far_clusters":
        a = mg.uniform(2000, 4); b = mg.uniform(2000, 5, offset=(1.0e6, 0.0, 0.0)); tris = np.concatenate([a, b])
    elif kind == "line":               # two zero extents: 30-bit 1-D code
        tris = mg.uniform(3000, 6)
        for v in ("v1", "v2", "v3"):
            tris[v][:, 1] = 0.5; tris[v][:, 2] = -2.0
    elif kind == "point_cloud_dups":   # 64 distinct positions, each 50 times
        tris = np.tile(mg.uniform(64, 7), 50)
    elif kind == "staircase":          # geometric spacing along a line: the LBVH degenerates into long chains (deep, skewed hierarchy;
        tris = mg.uniform(6000, 9)     # many nodes cross every tile boundary of the block schedulers)
        x = (np.float32(2.0) ** (-(np.arange(6000) % 120).astype(np.float32) / 4)) + (np.arange(6000) // 120).astype(np.float32) * np.float32(1e-6)
        for v in ("v1", "v2", "v3"):
            tris[v][:, 0] = x; tris[v][:, 1] = 0.0; tris[v][:, 2] = 0.0
    elif kind == "denormal_areas":     # coordinates ~1e-20: every union's area is an f32 denormal (~1e-40) — as the high word of the 64-bit selection key that is an f64
        tris = mg.uniform(4000, 10)    # denormal too, which the register half of the neighbour selection (v_min_f64, HP_NN_LDS = 3) must compare, not flush
        for v in ("v1", "v2", "v3"):
            tris[v] *= np.float32(2.0 ** -66)
    else:                              # one scene-sized triangle among tiny ones (Sponza-like size variance)
        tris = mg.uniform(4000, 8); tris["v1"][0] = (-50, -50, -50); tris["v2"][0] = (60, 0, 0); tris["v3"][0] = (0, 70, 55)
    tris = np.ascontiguousarray(tris); n = len(tris)
    got = pkg.BUILDERS[algo]().build(ctx, tris).download(); ref = orc.build_tree(algo, tris)
    assert np.array_equal(got["sorted_keys"], ref["skeys"]) and np.array_equal(got["sorted_vals"], ref["svals"])
    assert orc.validate_bvh2(got["nodes"], got["leaves"], got["root"], n, got["layout"]) == 0
    if algo in (0, 1, 2):
        assert got["nodes"].tobytes() == ref["nodes"].tobytes() and got["root"] == ref["root"]
    else:
        assert orc.topology_hash(got["nodes"], got["leaves"], 0, n, 1) == orc.topology_hash(ref["nodes"], ref["leaves"], 0, n, 1)
