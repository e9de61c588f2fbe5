"""GPU, BASELINE.json's full sizes (10 M-triangle uniform mesh; 262 144-triangle Sponza-class mesh): size-independent properties —
sortedness and permutation of the sort output, structural validity of the tree (every primitive reachable exactly once, every
internal box = union of its children, bit exact), root box = scene extent, the two HPLOC schedulers agree on the topology, and the
device-side SAH equals the CPU evaluation of the downloaded tree — and, since round 3, the ORACLE's own tree at the headline size: the
committed goldens of tools/make_golden.py fullsize (tests/golden/reference_outputs.json "_fullsize": topology hash, f64 SAH, leaf / node
FNV of the pinned CPU oracle on uniform(10 M, 1) and uniform(2 M, 100)) plus a live oracle run (11 s at 10 M)."""
import json
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500)]

N_FULL = 10_000_000


@pytest.fixture(scope="module")
def fullsize_golden():
    from conftest import GOLDEN
    return json.load(open(os.path.join(GOLDEN, "reference_outputs.json")))["_fullsize"]


@pytest.fixture(scope="module")
def big(pkg):
    return pkg.meshgen.uniform(N_FULL, 1)


def _check_common(pkg, orc, got, tris, scene):
    n = len(tris)
    k = got["sorted_keys"]
    assert np.all(k[1:] >= k[:-1]), "sorted keys not ascending"
    assert np.array_equal(np.sort(got["sorted_vals"]), np.arange(n, dtype=np.uint32)), "sorted values are not a permutation"
    assert got["scene"].tobytes() == scene.tobytes()
    assert orc.validate_bvh2(got["nodes"], got["leaves"], got["root"], n, got["layout"]) == 0
    root = got["nodes"][got["root"]]
    assert np.array_equal(root["min"], scene["min"][0]) and np.array_equal(root["max"], scene["max"][0])


def test_hploc_10m_properties_and_scheduler_agreement(pkg, orc, ctx, big, fullsize_golden):
    n = len(big)
    _, scene = orc.prim_bounds(big)
    gold = fullsize_golden["uniform10000000_s1"]["hploc"]
    hashes, sahs = [], []
    for mode in ("block", "async"):
        with ctx.options(hploc=mode):
            b = pkg.HPLOC().build(ctx, big)
            got = b.download()
        _check_common(pkg, orc, got, big, scene)
        assert np.array_equal(got["leaves"]["prim"], got["sorted_vals"])
        hashes.append(orc.topology_hash(got["nodes"], got["leaves"], 0, n, 1))
        s_cpu = orc.sah_bvh2(got["nodes"], got["leaves"], 0, n, 1)[0]
        assert abs(b.sah_cost() - s_cpu) <= 1e-9 * s_cpu
        sahs.append(s_cpu)
        # BASELINE.json config 3 against the pinned oracle (committed golden): same leaves, same canonical topology, SAH within 1e-4
        assert "%016x" % orc.fnv1a(got["leaves"]) == gold["leaves_fnv"], mode
        assert "%016x" % hashes[-1] == gold["topology"], f"{mode}: HPLOC topology at 10 M differs from the oracle's"
        assert abs(s_cpu - gold["sah_f64"]) <= 1e-4 * gold["sah_f64"]
    assert hashes[0] == hashes[1], "block-local and asynchronous HPLOC schedulers must build the same tree"
    # and the oracle run live on this box reproduces its own golden (the fixture is not stale)
    ref = orc.build_tree(3, big)
    assert "%016x" % orc.topology_hash(ref["nodes"], ref["leaves"], 0, n, 1) == gold["topology"] and ref["stats"]["merge_calls"] == gold["stats"]["merge_calls"]


def test_hploc_ticket_climb_on_clustered_60bit_keys(pkg, orc, ctx):
    """above 8 M leaves k_hploc_ext deals its queue by tickets; a clustered mesh with u64 keys, both schedulers, same tree"""
    tris = pkg.meshgen.sponza_like(8_000_123, 5); n = len(tris)
    d_tris = ctx.upload(tris)
    hashes = []
    for mode in ("block", "async"):
        with ctx.options(hploc=mode):
            got = pkg.HPLOC().build_ex(ctx, n, tris=d_tris, morton_bits=60).download()
        k = got["sorted_keys"]
        assert k.dtype == np.uint64 and np.all(k[1:] >= k[:-1])
        assert orc.validate_bvh2(got["nodes"], got["leaves"], got["root"], n, got["layout"]) == 0
        hashes.append(orc.topology_hash(got["nodes"], got["leaves"], 0, n, 1))
    assert hashes[0] == hashes[1]


@pytest.mark.parametrize("algo", [1, 0])
def test_lbvh_10m_bit_exact(pkg, orc, ctx, big, algo):
    """the LBVH oracle is linear-time: bit-exact comparison at full size"""
    n = len(big)
    b = pkg.BUILDERS[algo]().build(ctx, big)
    got = b.download()
    boxes, scene = orc.prim_bounds(big)
    _check_common(pkg, orc, got, big, scene)
    keys, vals = orc.morton_codes(boxes, scene)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(got["sorted_keys"], keys[order]) and np.array_equal(got["sorted_vals"], order.astype(np.uint32))
    if algo == 1:
        ref, root = orc.lbvh_single(big, got["sorted_keys"], got["sorted_vals"])
        assert root == got["root"]
    else:
        ref, _ = orc.lbvh_two(big, got["sorted_keys"], got["sorted_vals"])
    assert got["nodes"].tobytes() == ref.tobytes()


def test_ploc_10m_properties(pkg, orc, ctx, big, fullsize_golden):
    n = len(big)
    _, scene = orc.prim_bounds(big)
    b = pkg.PLOCNew().build(ctx, big)
    got = b.download()
    _check_common(pkg, orc, got, big, scene)
    s_cpu = orc.sah_bvh2(got["nodes"], got["leaves"], 0, n, 1)[0]
    assert abs(b.sah_cost() - s_cpu) <= 1e-9 * s_cpu
    # against the pinned oracle at 10 M (committed golden): PLOC++ node arrays are byte-identical (deterministic numbering), same iteration count
    gold = fullsize_golden["uniform10000000_s1"]["ploc"]
    assert "%016x" % orc.fnv1a(got["leaves"]) == gold["leaves_fnv"] and "%016x" % orc.fnv1a(got["nodes"]) == gold["nodes_fnv"]
    assert b.timings.ploc_iterations == gold["stats"]["iterations"]
    assert abs(s_cpu - gold["sah_f64"]) <= 1e-4 * gold["sah_f64"]


def test_config5_mesh_2m_vs_oracle_golden(pkg, orc, ctx, fullsize_golden):
    """config 5's first mesh, uniform(2 000 000, seed 100): HPLOC / PLOC++ / single-pass LBVH against the committed oracle outputs"""
    tris = pkg.meshgen.uniform(2_000_000, 100); n = len(tris)
    g = fullsize_golden["uniform2000000_s100"]
    d = ctx.upload(tris)
    for algo, tag in ((3, "hploc"), (2, "ploc"), (1, "lbvh_single")):
        got = pkg.BUILDERS[algo]().build_ex(ctx, n, tris=d).download()
        assert "%016x" % orc.fnv1a(got["sorted_keys"]) == g["sorted_keys_fnv"] and "%016x" % orc.fnv1a(got["sorted_vals"]) == g["sorted_vals_fnv"]
        assert "%016x" % orc.topology_hash(got["nodes"], got["leaves"], got["root"], n, got["layout"]) == g[tag]["topology"], tag
        if tag != "hploc":
            assert "%016x" % orc.fnv1a(got["nodes"]) == g[tag]["nodes_fnv"] and got["root"] == g[tag]["root"], tag
        s = orc.sah_bvh2(got["nodes"], got["leaves"], got["root"], n, got["layout"])[0]
        assert abs(s - g[tag]["sah_f64"]) <= 1e-4 * g[tag]["sah_f64"]
    d.free()


def test_sponza_262k_all_builders_vs_oracle(pkg, orc, ctx):
    """BASELINE.json configs[1] / configs[3] size: full oracle comparison (the oracle takes ~1 s here)"""
    tris = pkg.meshgen.sponza_like(262_144, 3); n = len(tris)
    for algo in (0, 1, 2, 3):
        b = pkg.BUILDERS[algo]().build(ctx, tris); got = b.download(); ref = orc.build_tree(algo, tris)
        if algo in (0, 1, 2):
            assert got["nodes"].tobytes() == ref["nodes"].tobytes() and got["root"] == ref["root"]
        else:
            assert orc.topology_hash(got["nodes"], got["leaves"], 0, n, 1) == orc.topology_hash(ref["nodes"], ref["leaves"], 0, n, 1)
        s_ref = orc.sah_bvh2(ref["nodes"], ref["leaves"], ref["root"], n, ref["layout"])[0]
        assert abs(b.sah_cost() - s_ref) <= 1e-4 * s_ref
    # PLOC variants must not be worse than LBVH on their own metric (config 4: "SAH <= reference")
    sah = {a: pkg.BUILDERS[a]().build(ctx, tris).sah_cost() for a in (1, 2, 3)}
    assert sah[2] < sah[1] and sah[3] < sah[1]


@pytest.mark.parametrize("n,kind", [(1_234_567, "sponza"), (600_001, "bunny")])
def test_tile_schedulers_at_odd_sizes(pkg, orc, ctx, n, kind):
    """sizes that are no multiple of anything, above the tile schedulers' thresholds: LBVH arrays byte-exact, HPLOC topology, PLOC++ valid"""
    tris = pkg.meshgen.sponza_like(n, 3) if kind == "sponza" else pkg.meshgen.bunny_like(n, 2)
    n = len(tris)
    fe = orc.front_end(tris)
    for algo in (0, 1, 3, 2):
        got = pkg.BUILDERS[algo]().build(ctx, tris).download()
        assert np.array_equal(got["sorted_keys"], fe["skeys"]) and np.array_equal(got["sorted_vals"], fe["svals"])
        assert orc.validate_bvh2(got["nodes"], got["leaves"], got["root"], n, got["layout"]) == 0
        if algo == 1:
            ref, root = orc.lbvh_single(tris, fe["skeys"], fe["svals"])
            assert root == got["root"] and got["nodes"].tobytes() == ref.tobytes()
        elif algo == 0:
            ref, _ = orc.lbvh_two(tris, fe["skeys"], fe["svals"])
            assert got["nodes"].tobytes() == ref.tobytes()
        elif algo == 3:
            hn, hl, _ = orc.hploc(fe["boxes"], fe["skeys"], fe["svals"])
            assert got["leaves"].tobytes() == hl.tobytes() and orc.topology_hash(got["nodes"], got["leaves"], 0, n, 1) == orc.topology_hash(hn, hl, 0, n, 1)


def test_reference_kernels_at_the_headline_size(pkg, orc, ctx, big):
    """Round 5: the reference's OWN kernels at BASELINE.json's headline size on the MI355X, not only the oracle's goldens — `CalculateSceneExtents` (wave64 build),
    `CalculateMortonCodes`, `InitBvhNodes` + `BvhBuildAndFit`, and `SetupClusters` + `HPloc` (contraction off) on uniform(10 M, 1): boxes / scene / keys / single-pass
    LBVH node array byte for byte, HPLOC leaves byte for byte and canonical topology, against the PRODUCT's build of the same mesh."""
    from conftest import require_ref
    require_ref(os.path.exists(orc.REF_DRIVER), "oracle/_ref/libref_driver.so (the reference's kernels)")
    n = len(big)
    boxes_ref, scene_ref = orc.ref_extents(big, nofma=True)
    keys_ref, _ = orc.ref_morton(boxes_ref, scene_ref, nofma=True)
    order = np.argsort(keys_ref, kind="stable").astype(np.uint32); skeys = keys_ref[order]
    b1 = pkg.SinglePassLbvh().build(ctx, big); g1 = b1.download()
    assert g1["scene"].tobytes() == scene_ref.tobytes()
    assert np.array_equal(g1["sorted_keys"], skeys) and np.array_equal(g1["sorted_vals"], order)
    nodes1, root1 = orc.ref_lbvh_single(big, skeys, order, nofma=True)
    assert g1["root"] == root1 and g1["nodes"].tobytes() == nodes1.tobytes(), "single-pass LBVH at 10 M != the reference's kernels"
    del nodes1, g1
    h_nodes, h_leaves, merged = orc.ref_hploc(boxes_ref, skeys, order, nofma=True, cover_all=(n - 1) % 32 == 0)
    assert merged == n - 1
    t_ref = orc.topology_hash(h_nodes, h_leaves, 0, n, 1)
    for mode in ("block", "async"):
        with ctx.options(hploc=mode):
            g3 = pkg.HPLOC().build(ctx, big).download()
        assert g3["leaves"].tobytes() == h_leaves.tobytes()
        assert orc.topology_hash(g3["nodes"], g3["leaves"], 0, n, 1) == t_ref, f"HPLOC ({mode}) at 10 M != the reference's HPloc kernel on the MI355X"


def test_reference_ploc_kernels_at_config5_size(pkg, orc, ctx):
    """the reference's Ploc / SinglePassPloc kernels (wave64 build, host loop with its per-iteration read-back) on uniform(2 M, 100) — config 5's first mesh — against the product"""
    from conftest import require_ref
    require_ref(os.path.exists(orc.REF_DRIVER), "oracle/_ref/libref_driver.so (the reference's kernels)")
    tris = pkg.meshgen.uniform(2_000_000, 100); n = len(tris)
    fe = orc.front_end(tris)
    p_nodes, p_leaves, p_iters = orc.ref_ploc(fe["boxes"], fe["svals"], nofma=True)
    b = pkg.PLOCNew().build(ctx, tris); g = b.download()
    assert g["leaves"].tobytes() == p_leaves.tobytes() and b.timings.ploc_iterations == p_iters
    assert orc.topology_hash(g["nodes"], g["leaves"], 0, n, 1) == orc.topology_hash(p_nodes, p_leaves, 0, n, 1), "PLOC++ at 2 M != the reference's Ploc kernels on the MI355X"
