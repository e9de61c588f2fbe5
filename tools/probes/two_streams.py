"""throughput of back-to-back builds on one ctx vs two ctxs (two HIP streams, builds alternating): the VALU-bound emit of one build can
overlap the HBM-bound E/M/S stages of the next.  python tools/probes/two_streams.py [N] [STEPS]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch; torch.cuda.init()
import bvh_pkg
pkg = bvh_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
tris = pkg.meshgen.uniform(n, 1)
d = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
for nctx in (1, 2, 3):
    ctxs = [pkg.Context(0) for _ in range(nctx)]
    bs = [pkg.HPLOC() for _ in range(nctx)]
    for c in ctxs: c.reserve(n); c.set_profiling(0)
    for i in range(6): bs[i % nctx].build(ctxs[i % nctx], d, on_device=True, n=n)
    for c in ctxs: c.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): bs[i % nctx].build(ctxs[i % nctx], d, on_device=True, n=n)
    for c in ctxs: c.synchronize()
    dt = time.perf_counter() - t0
    print(f"{nctx} stream(s): {dt / steps * 1e3:.4f} ms per build, {n / (dt / steps) / 1e6:.0f} Mtris/s", flush=True)
    for c in ctxs: c.close()
