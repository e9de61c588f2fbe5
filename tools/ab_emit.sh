#!/bin/bash
# A/B of HPLOC emit variants (build/variants/libbvh_<name>.so from tools/build_variant.sh): per-kernel HIP-event times of 20 builds at 10 M and 2 M uniform
# triangles + the tree checksum (node numbering depends on the topology only: every correct variant prints the same checksum).  Usage: tools/ab_emit.sh v1 v2 ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
echo "== $v"
BVH_MI355X_LIB=$R/build/variants/libbvh_$v.so timeout 150 python - <<PY
import os, sys
import numpy as np, torch
torch.cuda.init()
sys.path.insert(0, "$R")
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
for n in (10_000_000, 2_000_000):
    tris = pkg.meshgen.uniform(n, 1)
    d = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
    ctx.set_option("hploc", "block")
    b = pkg.HPLOC()
    for _ in range(3): b.build(ctx, d, on_device=True, n=n)
    chk = b.checksum()
    ctx.set_profiling(2)
    for _ in range(20): b.build(ctx, d, on_device=True, n=n)
    kt = ctx.kernel_times(); ctx.set_profiling(0)
    print(n, "  ".join(f"{k} {v[0]/20:.4f}" for k, v in kt.items() if "hploc" in k), " checksum %016x" % chk, flush=True)
PY
done
