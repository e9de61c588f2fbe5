#!/bin/bash
# per-kernel times of small / mid-size HPLOC builds for library variants (build/variants/libbvh_<name>.so)
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  echo "== $v"
  BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so timeout 200 python - <<'PY'
import os, sys
import numpy as np, torch
torch.cuda.init()
sys.path.insert(0, "/root/repo")
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
for n, kind in ((150000, "bunny"), (262144, "sponza"), (900000, "uniform"), (2000000, "uniform"), (10000000, "uniform")):
    tris = pkg.meshgen.uniform(n, 1) if kind == "uniform" else pkg.meshgen.sponza_like(n, 3) if kind == "sponza" else pkg.meshgen.bunny_like(n, 2)
    n = len(tris)
    d = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
    b = pkg.HPLOC()
    for _ in range(5): b.build(ctx, d, on_device=True, n=n)
    ctx.set_profiling(2)
    for _ in range(30): b.build(ctx, d, on_device=True, n=n)
    kt = ctx.kernel_times(); ctx.set_profiling(0)
    print(kind, n, "  ".join(f"{k} {v[0]/30:.4f}" for k, v in kt.items() if "hploc" in k), flush=True)
PY
done
