import os, sys, time
import numpy as np, torch
torch.cuda.init()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bvh_pkg
pkg = bvh_pkg.load()
tris = pkg.meshgen.sponza_like(262_144, 3); n = len(tris)
d = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
torch.cuda.synchronize()
t0 = time.perf_counter(); ctx = pkg.Context(0); t1 = time.perf_counter(); ctx.reserve(n); ctx.synchronize(); t2 = time.perf_counter()
print(f"ctx create {1e3*(t1-t0):.3f} ms, reserve {1e3*(t2-t1):.3f} ms")
for algo, name in ((pkg.ALGO_SINGLEPASS, "lbvh1"), (pkg.ALGO_HPLOC, "hploc"), (pkg.ALGO_PLOCPP, "ploc"), (pkg.ALGO_TWOPASS, "lbvh2")):
    b = pkg.BUILDERS[algo]()
    ts = []
    for i in range(4):
        t0 = time.perf_counter(); b.build(ctx, d, on_device=True, n=n); ctx.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print(name, " ".join(f"{t:.3f}" for t in ts))
