#!/usr/bin/env python3
"""whole-build / stage times of the extended build path: triangle input formats and 60-bit keys.  python tools/time_variants.py [N]"""
import os, sys
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
tris = pkg.meshgen.uniform(n, 1)
pk = np.ascontiguousarray(np.concatenate([tris["v1"], tris["v2"], tris["v3"]], axis=1))
bun = pkg.meshgen.bunny_like(n, 2); nb = len(bun)
allv = np.concatenate([bun["v1"], bun["v2"], bun["v3"]], axis=0)
verts, first, inv = np.unique(allv, axis=0, return_index=True, return_inverse=True)
# vertices in first-use order (np.unique sorts by coordinate, which would make every gather a cold miss; real index buffers are local)
face_major = np.stack([np.arange(nb), nb + np.arange(nb), 2 * nb + np.arange(nb)], axis=1).reshape(-1)      # v1,v2,v3 of face 0, of face 1, ...
rank_of_slot = np.empty(3 * nb, dtype=np.int64); rank_of_slot[face_major] = np.arange(3 * nb)
order = np.argsort(rank_of_slot[first], kind="stable"); remap = np.empty(len(verts), dtype=np.int64); remap[order] = np.arange(len(verts))
verts = verts[order]; inv = remap[inv]
idx = np.ascontiguousarray(np.stack([inv[:nb], inv[nb:2 * nb], inv[2 * nb:]], axis=1).astype(np.uint32)); verts = np.ascontiguousarray(verts.astype(np.float32))
d_tris = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda(); d_pk = torch.from_numpy(pk).cuda()
d_bun = torch.from_numpy(bun.view(np.uint8).reshape(-1)).cuda(); d_v = torch.from_numpy(verts).cuda(); d_i = torch.from_numpy(idx.view(np.int32)).cuda()

def run(label, algo, nn, **kw):
    b = pkg.BUILDERS[algo]()
    for _ in range(3): b.build_ex(ctx, nn, **kw)
    ctx.set_profiling(1); rows = []
    for _ in range(10):
        b.build_ex(ctx, nn, **kw); t = b.timings; rows.append((t.ms_total, t.ms_extents, t.ms_morton, t.ms_sort, t.ms_build))
    ctx.set_profiling(0)
    r = sorted(rows)[5]
    print(f"{label}: total {r[0]:.3f} ms ({nn / r[0] / 1e3:.0f} Mtris/s)  E {r[1]:.3f}  M {r[2]:.3f}  S {r[3]:.3f}  B {r[4]:.3f}", flush=True)

for algo, name in ((pkg.ALGO_HPLOC, "hploc"), (pkg.ALGO_SINGLEPASS, "lbvh_single"), (pkg.ALGO_TWOPASS, "lbvh_two"), (pkg.ALGO_PLOCPP, "ploc")):
    run(f"{name} uniform n={n} padded64 30-bit", algo, n, tris=d_tris)
    if algo in (pkg.ALGO_HPLOC, pkg.ALGO_SINGLEPASS):
        run(f"{name} uniform n={n} packed36 30-bit", algo, n, tris=d_pk, tri_format=pkg.TRI_PACKED36)
        run(f"{name} uniform n={n} padded64 60-bit", algo, n, tris=d_tris, morton_bits=60)
run(f"hploc bunny n={nb} padded64", pkg.ALGO_HPLOC, nb, tris=d_bun)
run(f"hploc bunny n={nb} indexed ({len(verts)} vertices)", pkg.ALGO_HPLOC, nb, vertices=d_v, indices=d_i, n_vertices=len(verts), tri_format=pkg.TRI_INDEXED)
