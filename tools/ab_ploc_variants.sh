#!/bin/bash
# Same-box alternating A/B of library variants on whole PLOC++ builds: VARIANTS="base x" [N="10000000 262144"] [REPS=2] TAG=name bash tools/ab_ploc_variants.sh (GPU box)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; N=${N:-"10000000 262144"}; REPS=${REPS:-2}; TAG=${TAG:-ploc}
for rep in $(seq $REPS); do for v in $VARIANTS; do echo "== $v"; for n in $N; do BVH_MI355X_LIB=$R/build/variants/libbvh_$v.so timeout 200 python - $n <<'PY'
import sys, time, os
import numpy as np, torch
torch.cuda.init()
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
n = int(sys.argv[1])
tris = pkg.meshgen.uniform(n, 1) if n > 1_000_000 else pkg.meshgen.sponza_like(n, 3)
d = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
b = pkg.PLOCNew()
for _ in range(5): b.build(ctx, d, on_device=True, n=len(tris))
ck = b.checksum()
ctx.synchronize(); t0 = time.perf_counter()
for _ in range(30): b.build(ctx, d, on_device=True, n=len(tris))
ctx.synchronize(); print(f"  n={len(tris)} PLOC++ {(time.perf_counter() - t0) / 30 * 1e3:.4f} ms/build  checksum {ck:016x}", flush=True)
PY
done; done; done 2>&1 | grep -v amdgpu | tee $R/gpurun_out/ab_$TAG.log
