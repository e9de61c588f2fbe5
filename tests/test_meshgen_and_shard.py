"""CPU: synthetic mesh generators are deterministic and sliceable; the multi-GPU shard (one mesh per rank + one all-gather of
root AABBs) is exercised with two gloo processes, the CPU oracle standing in for the per-rank builder."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def test_generators_deterministic_and_sliceable(pkg, orc):
    mg = pkg.meshgen
    a, b = mg.uniform(10_000, 7), mg.uniform(10_000, 7)
    assert a.tobytes() == b.tobytes() and a.dtype.itemsize == 64
    part = mg.uniform_slice(10_000, 1000, 7, start=4000)
    assert part.tobytes() == a[4000:5000].tobytes()
    assert mg.uniform(1000, 8).tobytes() != mg.uniform(1000, 9).tobytes()
    for t in (mg.bunny_like(5000, 2), mg.sponza_like(5000, 3)):
        assert len(t) == 5000 and np.isfinite(np.stack([t["v1"], t["v2"], t["v3"]])).all()
    s = mg.sponza_like(20_000, 3)
    boxes, _ = orc.prim_bounds(s)
    ext = boxes["max"] - boxes["min"]
    assert (ext.min(axis=1) == 0).sum() > 1000                # axis-aligned walls: zero-extent AABBs, as in Sponza
    assert mg.load_tri(os.path.join(ROOT, "tests", "golden", "cornell32.tri")).shape[0] == 32


def test_shard_assignment(pkg):
    assert pkg.shard(8, 8, 3) == [3]
    assert pkg.shard(8, 2, 1) == [1, 3, 5, 7]
    assert sorted(sum((pkg.shard(5, 3, r) for r in range(3)), [])) == list(range(5))


class _OracleBuilder:
    """CPU stand-in with the builder interface (tests only)"""

    def build(self, context, prims):
        import oracle as orc
        r = orc.build_tree(3, prims)
        self.root = np.concatenate([r["nodes"]["min"][0], r["nodes"]["max"][0]])
        return self


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bvh_pkg
    pkg = bvh_pkg.load()
    batch = [pkg.meshgen.uniform(500 + 37 * m, 100 + m, offset=(float(m), 0.0, 0.0)) for m in range(5)]
    bb = pkg.BatchedBvhBuilder(_OracleBuilder, root_aabb_fn=lambda b: b.root).build(None, batch)
    q.put((rank, bb.root_aabbs.copy(), bb.owner.copy(), sorted(bb.builders)))
    dist.barrier()
    dist.destroy_process_group()


def test_batched_builder_two_ranks_gloo(pkg, orc):
    world, port = 2, 29611
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60); assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])      # every rank holds every root AABB
    assert res[0][3] == [0, 2, 4] and res[1][3] == [1, 3]
    assert res[0][2].tolist() == [0, 1, 0, 1, 0]
    for m in range(5):                                                                         # == single-process builds
        tris = pkg.meshgen.uniform(500 + 37 * m, 100 + m, offset=(float(m), 0.0, 0.0))
        _, scene = orc.prim_bounds(tris)
        assert np.array_equal(res[0][1][m], np.concatenate([scene["min"][0], scene["max"][0]]))
