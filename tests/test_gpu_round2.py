"""GPU: BASELINE.json config 5 at its stated shape, the product-side BVH4 cost, caller-supplied 32-bit keys through bvh_emit_hploc,
the device checksum, config 4's image at Sponza size, and a slice of the randomised soak (tools/soak.py)."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

from conftest import ROOT, require_ref

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500)]


def test_config5_eight_2m_meshes_batched(pkg, orc, ctx):
    """BASELINE.json configs[4] / SURVEY.md §8(e): 8 x uniform(2 000 000, seed 100 + m, offset (m, 0, 0)) through the batched builder on
    the visible device list (mesh m -> device m mod G), RCCL all-gather of the root boxes; per-mesh trees identical to single builds."""
    import torch
    devs = tuple(range(torch.cuda.device_count()))
    meshes = [pkg.meshgen.uniform(2_000_000, 100 + m, offset=(float(m), 0.0, 0.0)) for m in range(8)]
    batch = pkg.Batch(devs)
    try:
        rep = batch.build(meshes, pkg.ALGO_HPLOC, checksums=True, sah=True)
        rep2 = batch.build(meshes, pkg.ALGO_HPLOC, checksums=True)          # contexts, arenas and communicator are reused
    finally:
        batch.close()
    print("\nconfig 5 on %d device(s): per-mesh build ms %s, all-gather %.1f us (second call %.1f us), wall %.1f ms" %
          (len(devs), np.round(rep["build_ms"], 3).tolist(), rep["allgather_us"], rep2["allgather_us"], rep2["wall_ms"]))
    assert (rep["build_ms"] > 0).all() and rep["allgather_us"] > 0
    assert np.array_equal(rep["checksums"], rep2["checksums"]) and np.array_equal(rep["root_aabbs"], rep2["root_aabbs"])
    for m, t in enumerate(meshes):
        n = len(t)
        _, scene = orc.prim_bounds(t)
        assert np.array_equal(rep["root_aabbs"][m], np.concatenate([scene["min"][0], scene["max"][0]])), m      # root box = scene extent, all-gathered
        b = pkg.HPLOC().build(ctx, t)                                                                            # the single-GPU build of the same mesh
        assert b.checksum() == int(rep["checksums"][m]), f"mesh {m}: batched tree differs from the single build"
        assert b.sah_cost() == pytest.approx(rep["sah"][m], rel=1e-12)
        if m in (0, 7):                                                                                          # and that single build is a valid tree
            got = b.download()
            assert pkg.checksum_host(got["nodes"], got["leaves"], got["root"]) == b.checksum()
            assert orc.validate_bvh2(got["nodes"], got["leaves"], 0, n, 1) == 0
            root = got["nodes"][0]
            assert np.array_equal(root["min"], scene["min"][0]) and np.array_equal(root["max"], scene["max"][0])


@pytest.mark.parametrize("algo", [0, 1, 2, 3])
def test_device_checksum_equals_host_mirror(pkg, ctx, algo):
    tris = pkg.meshgen.sponza_like(70_001, 4)
    b = pkg.BUILDERS[algo]().build(ctx, tris); got = b.download()
    assert b.checksum() == pkg.checksum_host(got["nodes"], got["leaves"], got["root"])
    nodes = got["nodes"].copy(); nodes["max"][123, 1] = np.nextafter(nodes["max"][123, 1], np.float32(np.inf))
    assert pkg.checksum_host(nodes, got["leaves"], got["root"]) != b.checksum()                                  # one flipped bit is seen


@pytest.mark.parametrize("name,n", [("cornell", 0), ("uniform", 50_000), ("sponza", 262_144), ("dups", 3000)])
@pytest.mark.parametrize("algo", [0, 1, 2, 3])
def test_bvh4_cost_on_device(pkg, orc, ctx, algo, name, n):
    """bvh_bvh4_cost = Utility::calculatebvh4Cost (src/Utility.cpp:351-396), the reference's m_cost: equal to the oracle's restatement on
    the same wide tree (f64 accumulation) and to the reference's own function (f32 accumulation in node order) within f32 rounding"""
    mg = pkg.meshgen
    tris = {"cornell": lambda: mg.load_tri(os.path.join(ROOT, "tests", "golden", "cornell382.tri")), "uniform": lambda: mg.uniform(n, 17),
            "sponza": lambda: mg.sponza_like(n, 3), "dups": lambda: np.repeat(mg.uniform(300, 12), 10)}[name]()
    n = len(tris)
    b = pkg.BUILDERS[algo]().build(ctx, tris)
    cost, n_wide, ms = b.collapse4_cost()
    wide, prims, total = b.collapse4()
    assert total == n_wide and ms >= 0
    boxes, _ = orc.prim_bounds(tris)
    c64, c32 = orc.sah_bvh4(wide, prims, boxes, total, n)
    assert cost == pytest.approx(c64, rel=1e-6)
    R = orc.ref_utility()
    require_ref(R is not None, "oracle/_ref/libref_utility.so (the reference's Utility.cpp)")
    if R is not None:
        w = np.ascontiguousarray(wide); p = np.ascontiguousarray(prims)
        # the reference accumulates in f32 in node-index order (0.2 % off at 262 k): the oracle's f32 emulation of exactly that loop reproduces it,
        # the device value is the same sum accumulated in f64
        assert R.ref_calculatebvh4Cost(w.ctypes.data, p.ctypes.data, boxes.ctypes.data, 0, total, n - 1) == pytest.approx(c32, rel=1e-6)
        assert cost == pytest.approx(c32, rel=5e-3)


@pytest.mark.parametrize("mode", ["async", "block"])
@pytest.mark.parametrize("n", [5000, 40_000])
def test_emit_hploc_keys_spanning_bit_31(pkg, orc, ctx, n, mode, sched_opts):
    """bvh_emit_hploc takes ANY sorted u32 keys (the reference compares the full 64-bit {key, index} words): keys on both sides of bit 31
    make the root gap's common prefix empty (length 0) — the case a shift by 64 - c mishandles — and must not poison the ctx's scratch"""
    sched_opts(hploc=mode)
    rng = np.random.default_rng(n)
    tris = pkg.meshgen.uniform(n, 3)
    boxes, _ = orc.prim_bounds(tris)
    keys = np.sort(rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32))
    assert keys[0] < 2**31 <= keys[-1]
    vals = rng.permutation(n).astype(np.uint32)
    L = pkg.lib()
    d_box, d_k, d_v = ctx.upload(boxes), ctx.upload(keys), ctx.upload(vals)
    d_nodes = ctx.alloc((n - 1) * 32); d_leaves = ctx.alloc(n * 28)
    for _ in range(2):
        assert L.bvh_emit_hploc(ctx.handle, d_box.ptr, d_k.ptr, d_v.ptr, n, d_nodes.ptr, d_leaves.ptr) == 0
        ctx.synchronize()
        gn, gl = d_nodes.download(pkg.BVH2_NODE, n - 1), d_leaves.download(pkg.PRIMREF, n)
        assert orc.validate_bvh2(gn, gl, 0, n, 1) == 0
        hn, hl, _ = orc.hploc(boxes, keys, vals)
        assert gl.tobytes() == hl.tobytes() and orc.topology_hash(gn, gl, 0, n, 1) == orc.topology_hash(hn, hl, 0, n, 1)
    # the same keys through the LBVH emitters (plen-based: bit 31 was never a problem there) and an ordinary build afterwards
    root = C.c_uint32(); d_n2 = ctx.alloc((2 * n - 1) * 32)
    assert L.bvh_emit_lbvh_single(ctx.handle, d_box.ptr, d_k.ptr, d_v.ptr, n, d_n2.ptr, C.byref(root)) == 0
    got = pkg.HPLOC().build(ctx, tris).download(); ref = orc.build_tree(3, tris)
    assert orc.topology_hash(got["nodes"], got["leaves"], 0, n, 1) == orc.topology_hash(ref["nodes"], ref["leaves"], 0, n, 1)


def test_small_arena_block_schedulers(pkg, orc):
    """a FRESH ctx whose arena is sized by a small build, tile schedulers forced: the tile kernels must never run on a root queue smaller
    than a tile can fill (round 1's session-wide, large-arena ctx hid that)"""
    c = pkg.Context(0)
    c.set_option("hploc", "block"); c.set_option("lbvh", "block")
    try:
        for n in (1500, 3000, 20_000):
            tris = pkg.meshgen.sponza_like(n, 9); n = len(tris)
            for algo in (0, 1, 3):
                got = pkg.BUILDERS[algo]().build(c, tris).download(); ref = orc.build_tree(algo, tris)
                if algo == 3:
                    assert orc.topology_hash(got["nodes"], got["leaves"], 0, n, 1) == orc.topology_hash(ref["nodes"], ref["leaves"], 0, n, 1)
                else:
                    assert got["nodes"].tobytes() == ref["nodes"].tobytes() and got["root"] == ref["root"]
    finally:
        c.close()


@pytest.mark.parametrize("algo", [1, 2, 3])
def test_collapse_has_no_depth_limit(pkg, orc, ctx, algo):
    """round 1's collapse gave up (BVH_E_INTERNAL) after 192 wide levels; the level loop now runs until a level creates nothing.  A 1 M
    "staircase" (geometric spacing along a line: long skewed chains) and nested shells (every triangle encloses the previous one: the
    agglomerative builders produce a chain as deep as the mesh is long) through every builder, collapse checked against the oracle's."""
    mg = pkg.meshgen
    stair = mg.uniform(1_000_000, 9)
    x = (np.float32(2.0) ** (-(np.arange(len(stair)) % 120).astype(np.float32) / 4)) + (np.arange(len(stair)) // 120).astype(np.float32) * np.float32(1e-6)
    for v in ("v1", "v2", "v3"):
        stair[v][:, 0] = x; stair[v][:, 1] = 0.0; stair[v][:, 2] = 0.0
    m = 3000
    shells = mg.uniform(m, 4)
    r = (1.0 + np.arange(m, dtype=np.float32))
    shells["v1"] = np.stack([-r, -r, -r], 1); shells["v2"] = np.stack([r, -r, r], 1); shells["v3"] = np.stack([-r, r, r], 1)
    for tris in (np.ascontiguousarray(stair), np.ascontiguousarray(shells)):
        if algo == 2:
            tris = tris[:60_000]        # PLOC++ on a collinear zero-area scene merges ~one pair per iteration (the reference too): keep it short
        n = len(tris)
        b = pkg.BUILDERS[algo]().build(ctx, tris); got = b.download()
        wide, prims, total = b.collapse4()
        ow, opn, ototal = orc.collapse4(got["nodes"], got["leaves"], got["root"], n, got["layout"])
        assert total == ototal and orc.topology_hash4(wide, prims, total, n) == orc.topology_hash4(ow, opn, ototal, n)
        assert np.array_equal(np.sort(prims["prim"]), np.arange(n, dtype=np.uint32))
        depth = orc.depth_bvh4(wide, total) if hasattr(orc, "depth_bvh4") else None
        print(f"\n{pkg.ALGO_NAMES[algo]} n={n}: {total} wide nodes" + (f", depth {depth}" if depth else ""))


def test_config4_image_at_sponza_262k(pkg, orc, ctx):
    """BASELINE.json configs[3] at its stated size: PLOC++ on the 262 144-triangle Sponza-class mesh -> LBVH-layout adapter -> while-while
    traversal, 512 x 512 image pixel-exact against the oracle's traversal of the oracle's tree.  Both use a 64-entry stack (the reference
    guards with `top < 64` but reserves 32 entries; the number of rays that go deeper than 32 is printed)."""
    tris = pkg.meshgen.sponza_like(262_144, 3); n = len(tris)
    cam, xf = pkg.cornell_view()
    cam["eye"][0] = (15.0, 6.0, 17.0, 0.0); cam["quat"][0] = pkg.qt_rotation((0.0, 1.0, 0.0, 0.0)); xf["translation"][0] = (0.0, 0.0, 0.0)
    b = pkg.PLOCNew().build(ctx, tris)
    rgba, rays = b.render(tris, cam, xf, 512)
    assert rgba[3::4].sum() > 255 * 100_000, "the view must see the room"
    ref = orc.build_tree(2, tris)
    assert b.download()["nodes"].tobytes() == ref["nodes"].tobytes()
    onodes = orc.ploc_to_lbvh_layout(ref["nodes"], ref["leaves"])
    img, deep = orc.trace_while(rays, tris, onodes, xf, 0, 512, n - 1)
    print(f"\nrays deeper than the reference's 32-entry stack: {deep} of {512 * 512}")
    assert np.array_equal(rgba, img), f"{np.count_nonzero(rgba != img)} of {rgba.size} bytes differ"
    # SAH <= the LBVH tree's on the same mesh (config 4: "SAH cost <= reference")
    assert b.sah_cost() < pkg.SinglePassLbvh().build(ctx, tris).sah_cost()


@pytest.mark.parametrize("knobs", [8, 32, 40])
def test_sort_makes_progress_under_any_dispatch_order(pkg, ctx, knobs, sched_opts):
    """The one-sweep sort uses workgroup ids as tile ids (no ticket atomic) and stays deadlock-free because a thread that polls an unpublished
    predecessor long enough computes that tile's digit total itself.  BVH_OPT_SORT_TEST_KNOBS = 8 hands the tiles out in REVERSE order (every resident
    tile's predecessors are not running), 32 makes threads help at the first empty poll: results must be the stable sort either way."""
    sched_opts(sort_knobs=knobs)
    L = pkg.lib()
    for n, bits in ((1, 32), (4097, 32), (300_001, 32), (1_200_003, 30)):
        rng = np.random.default_rng(n)
        keys = rng.integers(0, 2**bits, n, dtype=np.uint64).astype(np.uint32); keys[::5] = keys[1 % n]
        d_k = ctx.upload(keys); d_sk = ctx.alloc(n * 4); d_sv = ctx.alloc(n * 4)
        assert L.bvh_sort_pairs(ctx.handle, d_k.ptr, None, n, d_sk.ptr, d_sv.ptr, 0, bits) == 0
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(d_sk.download(np.uint32, n), keys[order]) and np.array_equal(d_sv.download(np.uint32, n), order.astype(np.uint32))
    tris = pkg.meshgen.sponza_like(1_100_000, 3)                   # and a whole build (u64 keys: the 16-byte record path, large-input tiles)
    sched_opts(sort_knobs=0)
    ref = pkg.HPLOC().build_ex(ctx, len(tris), tris=ctx.upload(tris), morton_bits=60).checksum()
    sched_opts(sort_knobs=knobs)
    assert pkg.HPLOC().build_ex(ctx, len(tris), tris=ctx.upload(tris), morton_bits=60).checksum() == ref


def test_soak_slice(pkg, orc, ctx):
    """30 seconds of tools/soak.py (random sizes incl. the scheduler thresholds at 0.3 M / 1 M / 8 M, both schedulers, 30- and 60-bit keys):
    re-validates the inline-asm / relaxed-atomic hand-off protocol (csrc/common.hpp) on every driver run"""
    spec = importlib.util.spec_from_file_location("soak", os.path.join(ROOT, "tools", "soak.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    sizes = mod.SIZES[:-1] + mod.THRESHOLD_SIZES[:5]
    builds, fails = mod.soak(pkg, orc, ctx, 30.0, 20260929, sizes=sizes, max_random=1_500_000)
    print(f"\nsoak slice: {builds} builds")
    assert builds >= 16 and not fails, fails
    # the 8 M threshold (ticketed external climb) once, deterministic
    for n in (7_999_999, 8_000_001):
        tris = pkg.meshgen.uniform(n, 5)
        d = ctx.upload(tris)
        cks = []
        for mode in ("block", "async"):
            with ctx.options(hploc=mode):
                cks.append(pkg.HPLOC().build_ex(ctx, n, tris=d).checksum())
        d.free()
        assert cks[0] == cks[1], n
