#!/usr/bin/env python3
"""Randomised soak of the four builders on the GPU box: random size / mesh kind / key width / scheduler, every tree validated
(structure, boxes, sort order) and — where it is cheap — compared with the oracle (LBVH node arrays byte for byte, HPLOC topology
hash, repeated builds byte-identical).  Hunts for rare orderings in the inter-workgroup hand-offs (relaxed agent-scope atomics +
sc1 accesses, csrc/common.hpp), which live outside what the compiler's memory model checks.
Usage: python tools/soak.py [SECONDS=240] [SEED=1]        (tests/test_gpu_round2.py::test_soak_slice runs a 30-second slice of it)"""
import os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = [2, 3, 17, 33, 511, 513, 1025, 2100, 40_000, 262_144, 500_001, 1_000_003, 2_345_678, 8_100_000]
# sizes straddling the scheduler thresholds: LBVH tiles from 240 k, HPLOC tiles from 800 k, wide sort tiles from 1 M, one-shot PLOC++ tickets below 2^20, ticketed external climb from 8 M
THRESHOLD_SIZES = [239_999, 240_000, 240_001, 799_999, 800_000, 800_001, 999_999, 1_000_000, 1_000_001, 1_048_575, 1_048_576, 1_048_577, 7_999_999, 8_000_000]


def soak(pkg, orc, ctx, budget: float, seed: int, sizes=SIZES, max_random: int = 3_000_000, log=print):
    """-> (builds, list of failure descriptions)"""
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    builds = 0; fails = []
    saved = {k: ctx.get_option(k) for k in ("hploc", "lbvh", "ploc")}
    try:
        while time.time() < t_end:
            n = int(rng.choice(sizes)) if rng.random() < 0.5 else int(np.exp(rng.uniform(np.log(2), np.log(max_random))))
            kind = rng.choice(["uniform", "sponza", "bunny"])
            tris = pkg.meshgen.uniform(n, int(rng.integers(1, 1 << 30))) if kind == "uniform" else pkg.meshgen.sponza_like(n, int(rng.integers(1, 99))) if kind == "sponza" else pkg.meshgen.bunny_like(n, int(rng.integers(1, 99)))
            n = len(tris)
            if n < 2: continue
            bits = 60 if rng.random() < 0.3 else 30
            d_tris = ctx.upload(tris)
            fe = orc.front_end(tris, morton_bits=bits) if n <= 1_200_000 else None
            for algo in (0, 1, 2, 3):
                for mode in ((("async", "single"), ("block", "block"), ("live", "block")) if algo == 3 else (("async", "single"), ("block", "block"))) if algo != 2 else (("", ""),):
                    if time.time() > t_end + 30: break
                    ctx.set_option("hploc", mode[0] or "auto"); ctx.set_option("lbvh", mode[1] or "auto")
                    if algo == 2: ctx.set_option("ploc", int(rng.integers(0, 2)))          # static chunk ids / tickets only
                    b = pkg.BUILDERS[algo]().build_ex(ctx, n, tris=d_tris, morton_bits=bits); got = b.download()
                    ok = orc.validate_bvh2(got["nodes"], got["leaves"], got["root"], n, got["layout"]) == 0
                    k = got["sorted_keys"]; ok = ok and bool(np.all(k[1:] >= k[:-1]))
                    why = "" if ok else "invalid tree / unsorted keys"
                    if ok and fe is not None:
                        if not (np.array_equal(k, fe["skeys"]) and np.array_equal(got["sorted_vals"], fe["svals"])): ok = False; why = "sort differs from oracle"
                        elif algo == 1:
                            ref, root = orc.lbvh_single(tris, fe["skeys"], fe["svals"])
                            if not (root == got["root"] and got["nodes"].tobytes() == ref.tobytes()): ok = False; why = "single-pass LBVH differs"
                        elif algo == 0:
                            ref, _ = orc.lbvh_two(tris, fe["skeys"], fe["svals"])
                            if got["nodes"].tobytes() != ref.tobytes(): ok = False; why = "two-pass LBVH differs"
                        elif algo == 3 and n <= 300_000:
                            hn, hl, _ = orc.hploc(fe["boxes"], fe["skeys"], fe["svals"])
                            if orc.topology_hash(got["nodes"], got["leaves"], 0, n, 1) != orc.topology_hash(hn, hl, 0, n, 1): ok = False; why = "HPLOC topology differs"
                    if ok:                                   # a second build on the same ctx: same checksum (scratch left clean, deterministic numbering)
                        ck = b.checksum()
                        again = pkg.BUILDERS[algo]().build_ex(ctx, n, tris=d_tris, morton_bits=bits)
                        if again.checksum() != ck: ok = False; why = "rebuild differs"
                    builds += 2
                    if not ok:
                        fails.append(f"n={n} kind={kind} bits={bits} algo={pkg.ALGO_NAMES[algo]} mode={mode}: {why}")
                        log("FAIL " + fails[-1])
            d_tris.free()
    finally:
        for k, old in saved.items():
            ctx.set_option(k, old)
    return builds, fails


if __name__ == "__main__":
    import torch
    torch.cuda.init()
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bvh_pkg
    import oracle as orc
    pkg = bvh_pkg.load(); ctx = pkg.Context(0)
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    builds, fails = soak(pkg, orc, ctx, budget, int(sys.argv[2]) if len(sys.argv) > 2 else 1, log=lambda s: print(s, flush=True))
    print(f"soak: {builds} builds, {len(fails)} failures", flush=True)
    sys.exit(1 if fails else 0)
