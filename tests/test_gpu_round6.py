"""Round-6 GPU tests: the overlapped HPLOC schedule (BVH_OPT_HPLOC_SCHEDULER = 3: k_hploc_live on the context's side stream beside the tile kernel — measured slower,
LEADS.md row 87, kept selectable) builds the same trees as the other two schedulers; PLOC++ with chunk tickets only (BVH_OPT_PLOC_SCHEDULER = 1, what the batched
builder's lanes use) builds the same trees as the default; several PLOC++ builds at once on one device."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


def _mesh(pkg, kind, n, seed):
    return pkg.meshgen.uniform(n, seed) if kind == "uniform" else pkg.meshgen.sponza_like(n, seed) if kind == "sponza" else pkg.meshgen.bunny_like(n, seed)


@pytest.mark.parametrize("kind,n", [("uniform", 2_100_000), ("sponza", 1_000_000), ("bunny", 900_000), ("uniform", 1500), ("uniform", 10_000_000)])
def test_overlapped_schedule_builds_the_same_tree(pkg, ctx, kind, n):
    """the three HPLOC schedulers agree byte for byte (node numbering follows from the topology alone: equal checksums = equal arrays), also when one context switches
    between the classic tile schedule (which leaves its queue items behind) and the overlapped one (which needs the slots all-zero and leaves them so), repeatedly"""
    tris = _mesh(pkg, kind, n, 7)
    with ctx.options(hploc="async"):
        ref = pkg.HPLOC().build(ctx, tris).checksum()
    got = []
    for mode in ("live", "block", "live", "live", "block", "live"):
        with ctx.options(hploc=mode):
            got.append(pkg.HPLOC().build(ctx, tris).checksum())
    assert got == [ref] * len(got), [f"{g:016x}" for g in got]
    # ... and the overlapped schedule is what ran (a context whose side stream could not be created falls back to the classic schedule)
    with ctx.options(hploc="live"):
        try:
            ctx.set_profiling(2)
            pkg.HPLOC().build(ctx, tris)
            kt = ctx.kernel_times()
        finally:
            ctx.set_profiling(0)
    assert "k_hploc_live(tail)" in kt and "k_hploc_ext" not in kt, kt


def test_overlapped_schedule_60_bit_keys(pkg, ctx):
    import torch
    tris = pkg.meshgen.uniform(1_300_000, 9)
    d = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
    cks = {}
    for mode in ("block", "live", "async"):
        with ctx.options(hploc=mode):
            cks[mode] = pkg.HPLOC().build_ex(ctx, len(tris), tris=d, morton_bits=60).checksum()
    assert cks["block"] == cks["live"] == cks["async"]


def test_overlapped_schedule_soak(pkg, ctx):
    """200 back-to-back overlapped builds without a synchronisation in between (the join event orders build k + 1's clearing kernel behind build k's consumers), then
    the last tree is checked"""
    tris = pkg.meshgen.uniform(2_000_000, 100)
    with ctx.options(hploc="block"):
        ref = pkg.HPLOC().build(ctx, tris).checksum()
    import torch
    d = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
    with ctx.options(hploc="live"):
        b = pkg.HPLOC()
        for _ in range(200):
            b.build(ctx, d, on_device=True, n=len(tris))
        assert b.checksum() == ref


@pytest.mark.parametrize("kind,n", [("sponza", 262_144), ("uniform", 1_100_000), ("uniform", 700)])
def test_ploc_tickets_only_builds_the_same_tree(pkg, ctx, kind, n):
    tris = _mesh(pkg, kind, n, 3)
    cks = []
    for v in (0, 1, 2):                 # 2: ABI 4's removed cooperative launch — still accepted, means 0
        ctx.set_option("ploc", v)
        cks.append(pkg.PLOCNew().build(ctx, tris).checksum())
    ctx.set_option("ploc", 0)
    assert ctx.get_option("ploc") == 0 and cks[0] == cks[1] == cks[2]


def test_batched_ploc_lanes_soak(pkg, ctx):
    """the batched builder runs up to three lanes (contexts, streams) per device at once; its lanes take chunk tickets in every PLOC++ iteration (no reliance
    on a grid being co-resident).  Nine meshes of three sizes, five rounds: every tree equals the single build's."""
    meshes = [pkg.meshgen.sponza_like(262_144, 3 + m) if m % 3 == 0 else pkg.meshgen.uniform(150_000 + 1000 * m, 40 + m) if m % 3 == 1 else pkg.meshgen.uniform(1_050_000, 60 + m)
              for m in range(9)]
    ref = [pkg.PLOCNew().build(ctx, t).checksum() for t in meshes]
    batch = pkg.Batch((0,))
    try:
        for _ in range(5):
            rep = batch.build(meshes, pkg.ALGO_PLOCPP, checksums=True)
            assert [int(c) for c in rep["checksums"]] == ref
        assert rep["lanes_per_device"] >= 2
    finally:
        batch.close()
