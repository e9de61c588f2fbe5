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
    """`numPrebits = (int)log2f(extent ratio)` (src/CommonBlocksKernel.h:175-248) is evaluated by the DEVICE's log2f in the reference and in the product; SURVEY.md
    section 7 flagged that a host libm may truncate differently.  Scenes whose extent ratios are 2^k * (1 - 2 ulp .. 1 + 2 ulp), k = 1..30, on every axis order:
      * product == the reference's own CalculateMortonCodes kernel on the MI355X, key for key, on EVERY scene (the parity that counts);
      * the device's plan (bvh_stage_morton_plan) == the oracle's plan and the oracle's keys == the reference's on every scene.  Round 4 measured this test with the
        oracle calling the host's log2f: 228 of 720 scenes (every ratio 1-2 ulp below 2^k, k >= 2) got a different plan — the device's log2f never rounds up to k, a
        correctly rounded one does — and the oracle's encoder driven by the device's plan reproduced the reference's keys on all of them: the truncation was the only
        difference.  The oracle now takes the exact floor (the ratio's binary exponent; oracle/bvh_oracle.cpp morton_plan), which is what the device computes."""
    require_ref(os.path.exists(orc.REF_DRIVER), "oracle/_ref/libref_driver.so (the reference's kernels)")
    L = pkg.lib()
    scenes = 0; plan_diff = []
    for perm in itertools.permutations(range(3)):
        for k in range(1, 31):
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
                    if plan_dev != plan_orc:
                        plan_diff.append((perm, k, d, plan_dev, plan_orc))
                        assert np.array_equal(orc.morton_codes_with_plan(boxes, scene, plan_dev), keys_ref), "oracle encoder with the device's plan != reference"
                    else:
                        assert np.array_equal(orc.morton_codes(boxes, scene)[0], keys_ref), f"oracle != reference at extents {e} on axes {perm} although the plans agree"
                    for b in (d_box, d_scene, d_keys):
                        b.free()
                    scenes += 1
    print(f"\n{scenes} boundary scenes: product == reference kernel on all; oracle plan != device plan on {len(plan_diff)}")
    for x in plan_diff[:8]:
        print("   axes %s  2^%d %+d ulp: device plan %s, oracle plan %s" % x)
    assert not plan_diff, "the oracle's (int)log2f differs from the device's"


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


def test_batch_keeps_every_mesh_and_pipelines_a_device(pkg, orc, ctx):
    """VERDICT r03 items 6 and 8.  (a) keep=True: every mesh's tree stays on its device (bvh_batch_mesh; the reference's d_bvhNodes / d_primRefs / d_rootNodes) and
    equals the tree of a single build; (b) a device that holds several meshes pipelines them on up to three contexts — H2D of one mesh, build of another, checksum /
    copy of a third overlap — and the result is the same as with the meshes built one after another.  Prints the aggregate rate from wall_ms for both."""
    import torch
    devs = tuple(range(torch.cuda.device_count()))
    meshes = [pkg.meshgen.uniform(2_000_000, 100 + m, offset=(float(m), 0.0, 0.0)) for m in range(8)]       # config 5's shape
    total = sum(len(t) for t in meshes)
    batch = pkg.Batch(devs)
    try:
        batch.build(meshes, pkg.ALGO_HPLOC, checksums=True)                                                    # (arenas, staging buffers and lanes get created)
        rep = batch.build(meshes, pkg.ALGO_HPLOC, checksums=True, sah=True, keep=True)
        assert rep["lanes_per_device"] == min(3, -(-len(meshes) // len(devs)))
        for m in (0, 3, 7):
            nodes, leaves = batch.download(m)
            bm = rep["meshes"][m]
            assert bm["n_leaves"] == len(meshes[m]) and bm["n_nodes"] == len(meshes[m]) - 1 and bm["layout"] == 1 and bm["root"] == 0
            assert pkg.checksum_host(nodes, leaves, 0) == int(rep["checksums"][m])
            single = pkg.HPLOC().build(ctx, meshes[m])
            assert single.checksum() == int(rep["checksums"][m])
            _, scene = orc.prim_bounds(meshes[m])
            assert np.array_equal(nodes[0]["min"], scene["min"][0]) and np.array_equal(nodes[0]["max"], scene["max"][0])
        walls = [batch.build(meshes, pkg.ALGO_HPLOC, checksums=True)["wall_ms"] for _ in range(3)]
        # the strictly serial form for comparison: one mesh per call
        t_serial = []
        for _ in range(2):
            w = 0.0
            for t in meshes:
                w += batch.build([t], pkg.ALGO_HPLOC, checksums=True)["wall_ms"]
            t_serial.append(w)
        print("\nconfig-5 batch on %d device(s), %d lanes per device: wall %.1f ms = %.0f Mtris/s (host buffers, H2D included); one mesh per call: %.1f ms = %.0f Mtris/s"
              % (len(devs), rep["lanes_per_device"], min(walls), total / min(walls) / 1e3, min(t_serial), total / min(t_serial) / 1e3))
        if len(devs) == 1:
            assert min(walls) < min(t_serial), "pipelining a device's meshes should not be slower than building them one call at a time"
    finally:
        batch.close()
    # LBVH algos keep 2n-1 nodes and no leaf array
    small = [pkg.meshgen.uniform(5000 + 11 * m, 7 + m) for m in range(3)]
    b2 = pkg.Batch(devs[:1])
    try:
        r2 = b2.build(small, pkg.ALGO_SINGLEPASS, checksums=True, keep=True)
        for m, t in enumerate(small):
            nodes, leaves = b2.download(m)
            assert leaves is None and len(nodes) == 2 * len(t) - 1
            fe = orc.front_end(t)
            o_nodes, o_root = orc.lbvh_single(t, fe["skeys"], fe["svals"])
            assert r2["meshes"][m]["root"] == o_root and nodes.tobytes() == o_nodes.tobytes()
    finally:
        b2.close()


@pytest.mark.parametrize("name,n", [("uniform", 4096), ("uniform", 5000), ("uniform", 33_333), ("uniform", 100_000), ("bunny", 150_000), ("sponza", 262_144), ("uniform", 262_144),
                                    ("uniform", 261_130), ("uniform", 4096 + 7), ("dups", 40_000), ("flat", 20_000), ("line", 6000), ("identical", 5000)])
def test_ploc_chunk_boundaries_and_degenerate_scenes_bit_exact(pkg, orc, ctx, name, n):
    """PLOC++ node array, leaves and iteration count equal the pinned oracle's byte for byte (test_ploc_bit_exact's bar) at sizes whose last chunk is shorter than the halo
    (4096 + 7, 261 130 = 255 x 1024 + 10) and on degenerate scenes (duplicates, flat, collinear, identical: thousands of iterations, restarts of the per-iteration bookkeeping);
    a second build of the same size (its launch batch is sized from the first) gives the same tree.  (Round 4 ran this list through a cooperative resident-launch variant too —
    measured slower, removed in round 5: tools/probes/r05_pruned_switches.patch.)"""
    mg = pkg.meshgen
    if name == "uniform":
        tris = mg.uniform(n, 31)
    elif name == "bunny":
        tris = mg.bunny_like(n, 2)
    elif name == "sponza":
        tris = mg.sponza_like(n, 3)
    elif name == "dups":
        tris = mg.uniform(n, 9); tris[::3] = tris[1]
    elif name == "flat":
        tris = mg.uniform(n, 8); tris["v1"][:, 2] = 0.25; tris["v2"][:, 2] = 0.25; tris["v3"][:, 2] = 0.25
    elif name == "line":
        tris = mg.uniform(n, 5)
        for f in ("v1", "v2", "v3"):
            tris[f][:, 1] = 0.0; tris[f][:, 2] = 0.0
    else:
        tris = np.repeat(mg.uniform(1, 3), n)
    ref = orc.build_tree(pkg.ALGO_PLOCPP, tris)
    b = pkg.PLOCNew().build(ctx, tris)
    got = b.download()
    assert got["nodes"].tobytes() == ref["nodes"].tobytes() and got["leaves"].tobytes() == ref["leaves"].tobytes(), (name, n)
    assert b.timings.ploc_iterations == ref["stats"]["iterations"], (b.timings.ploc_iterations, ref["stats"]["iterations"])     # (also across restarts of the bookkeeping: flat / collinear scenes)
    b2 = pkg.PLOCNew().build(ctx, tris)
    assert b2.checksum() == b.checksum()


class _RawDevice:
    """a device address as a torch tensor (no copy)"""
    def __init__(self, ptr, n, typestr): self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def test_seventy_million_triangles_cross_the_4_gib_offset(pkg, orc):
    """Maximum sizes (the boundary accepts n < 2^30; tools/large_n.py ran 50 M .. 400 M): 70 M triangles = 4.2 GiB of 64-byte records, so every byte offset into the
    input that was computed in 32 bits would wrap.  The mesh is generated on the device; both schedulers of HPLOC and of the single-pass LBVH must agree (bvh_checksum),
    the sorted keys are checked on the device, the values are a permutation (sum), and the LBVH tree is downloaded and run through the oracle's validator."""
    import torch
    n = 70_000_000
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    d = torch.zeros((n, 16), device="cuda", dtype=torch.float32)
    for lo in range(0, n, 10_000_000):
        hi = min(n, lo + 10_000_000)
        c = torch.rand((hi - lo, 3), device="cuda", generator=g)
        for v in range(3):
            d[lo:hi, 3 * v:3 * v + 3] = c + 0.001 * (torch.rand((hi - lo, 3), device="cuda", generator=g) - 0.5)
    torch.cuda.synchronize()
    big = pkg.Context(0)
    try:
        for algo, opt, modes in ((pkg.ALGO_HPLOC, "hploc", ("block", "async")), (pkg.ALGO_SINGLEPASS, "lbvh", ("block", "single"))):
            sums = []
            for mode in modes:
                big.set_option(opt, mode)
                b = pkg.BUILDERS[algo]().build(big, d, on_device=True, n=n)
                r = b.result
                assert r.n_leaves == n
                keys = torch.as_tensor(_RawDevice(r.d_sorted_keys, n, "<u4"), device="cuda").to(torch.int64)
                assert bool((keys[1:] >= keys[:-1]).all()), "sorted keys"
                del keys
                vals = torch.as_tensor(_RawDevice(r.d_sorted_vals, n, "<u4"), device="cuda").to(torch.int64)
                assert int(vals.sum()) == n * (n - 1) // 2 and int(vals.max()) == n - 1, "values are a permutation of 0..n-1"
                del vals
                sums.append(b.checksum())
                if algo == pkg.ALGO_SINGLEPASS and mode == "block":
                    got = b.download()
                    assert orc.validate_bvh2(got["nodes"], got["leaves"], got["root"], n, got["layout"]) == 0
                    root = got["nodes"][got["root"]]; sc = got["scene"][0]
                    assert np.array_equal(root["min"], sc["min"]) and np.array_equal(root["max"], sc["max"])
                    # the last triangle's box (byte offset 4 479 999 936 in the input) is the leaf that carries primitive n - 1
                    last = d[n - 1, :9].cpu().numpy().reshape(3, 3)
                    leaf = got["nodes"][n - 1 + int(np.nonzero(got["sorted_vals"] == n - 1)[0][0])]
                    assert np.array_equal(leaf["min"], last.min(axis=0)) and np.array_equal(leaf["max"], last.max(axis=0))
                    del got
            assert sums[0] == sums[1], f"{opt}: the two schedulers built different trees"
    finally:
        big.close()


@pytest.mark.parametrize("pattern", ["sorted", "constant", "two_values_per_wave", "shared_top_digit", "runs_of_17"])
@pytest.mark.parametrize("n", [1_000_003, 5_003])
def test_sort_histograms_on_coherent_keys(pkg, ctx, pattern, n):
    """The digit histograms count a wave's keys in groups when they share digits (csrc/common.hpp hist_add_passes: k_hist here, k_morton / k_morton64 in the builds — a
    Morton-ordered mesh is exactly this input).  Key patterns that exercise the grouped paths — every lane in the first lane's group, two groups, a shared top digit with random
    low digits, groups that straddle waves — with n not a multiple of 64 (partial last wave): stable-argsort parity for 30- and 32-bit sorts."""
    rng = np.random.default_rng(len(pattern) * 7 + n)
    if pattern == "sorted": keys = np.sort(rng.integers(0, 2**30, n, dtype=np.uint64)).astype(np.uint32)
    elif pattern == "constant": keys = np.full(n, 0x2AAAAAAA, np.uint32)
    elif pattern == "two_values_per_wave": keys = np.where(np.arange(n) % 3 == 0, 0x00FF00FF, 0x3F00FF00).astype(np.uint32)
    elif pattern == "shared_top_digit": keys = (np.uint32(0x2A000000) | rng.integers(0, 2**24, n, dtype=np.uint64).astype(np.uint32))
    else: keys = np.repeat(rng.integers(0, 2**30, n // 17 + 1, dtype=np.uint64).astype(np.uint32), 17)[:n]
    L = pkg.lib()
    dk, ok, ov = ctx.upload(keys), ctx.alloc(n * 4), ctx.alloc(n * 4)
    for bits in (30, 32):
        assert L.bvh_sort_pairs(ctx.handle, dk.ptr, None, n, ok.ptr, ov.ptr, 0, bits) == 0
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(ok.download(np.uint32, n), keys[order])
        assert np.array_equal(ov.download(np.uint32, n), order.astype(np.uint32))


@pytest.mark.parametrize("bits", [30, 60])
def test_builds_on_a_morton_ordered_mesh(pkg, orc, ctx, bits):
    """the same grouped counting inside the builds (k_morton, k_morton64): a mesh re-ordered by its own Morton order — single-pass LBVH byte-exact against the oracle,
    HPLOC valid with the same SAH as the oracle's, the sorted keys ascending and the values a permutation, for both key widths"""
    tris = pkg.meshgen.uniform(300_007, 5)
    order = pkg.BUILDERS[pkg.ALGO_SINGLEPASS]().build(ctx, tris).download()["sorted_vals"]
    tris = np.ascontiguousarray(tris[order]); n = len(tris)
    d = ctx.upload(tris)
    got = pkg.BUILDERS[pkg.ALGO_SINGLEPASS]().build_ex(ctx, n, tris=d, morton_bits=bits).download()
    k = got["sorted_keys"]
    assert bool(np.all(k[1:] >= k[:-1])) and np.array_equal(np.sort(got["sorted_vals"]), np.arange(n, dtype=np.uint32))
    assert orc.validate_bvh2(got["nodes"], None, got["root"], n, 0) == 0
    ref = orc.build_tree(1, tris, bits)
    assert np.array_equal(k, ref["skeys"]) and np.array_equal(got["sorted_vals"], ref["svals"])
    assert got["nodes"].tobytes() == ref["nodes"].tobytes()
    h = pkg.HPLOC().build_ex(ctx, n, tris=d, morton_bits=bits)
    hg = h.download()
    assert orc.validate_bvh2(hg["nodes"], hg["leaves"], hg["root"], n, 1) == 0
    if True:
        ref = orc.build_tree(3, tris, bits)
        assert abs(h.sah_cost() - orc.sah_bvh2(ref["nodes"], ref["leaves"], ref["root"], n, 1)[0]) <= 1e-9 * max(1.0, h.m_cost)
