import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch; torch.cuda.init()
import bvh_pkg, oracle as orc
pkg = bvh_pkg.load(); ctx = pkg.Context(0); mg = pkg.meshgen
def stair(n):
    t = mg.uniform(n, 9)
    x = (np.float32(2.0) ** (-(np.arange(n) % 120).astype(np.float32) / 4)) + (np.arange(n) // 120).astype(np.float32) * np.float32(1e-6)
    for v in ("v1", "v2", "v3"):
        t[v][:, 0] = x; t[v][:, 1] = 0.0; t[v][:, 2] = 0.0
    return np.ascontiguousarray(t)
def shells(m):
    t = mg.uniform(m, 4); r = (1.0 + np.arange(m, dtype=np.float32))
    t["v1"] = np.stack([-r, -r, -r], 1); t["v2"] = np.stack([r, -r, r], 1); t["v3"] = np.stack([-r, r, r], 1)
    return np.ascontiguousarray(t)
for name, t in (("shells3000", shells(3000)), ("stair100k", stair(100_000)), ("stair1M", stair(1_000_000))):
    t0 = time.time()
    try:
        b = pkg.PLOCNew().build(ctx, t)
        print(name, "ok iterations", b.timings.ploc_iterations, f"{time.time()-t0:.2f}s", flush=True)
    except Exception as e:
        print(name, "FAILED", e, f"{time.time()-t0:.2f}s", flush=True)
    if len(t) <= 100_000:
        r = orc.build_tree(2, t); print("  oracle iterations", r["stats"]["iterations"], flush=True)
