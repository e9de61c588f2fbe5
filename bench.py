#!/usr/bin/env python3
"""bench.py — BVH build throughput on MI355X (metric of BASELINE.json: build Mtris/s over extents+Morton+sort+emit).

One step = one complete build (stage E, M, S, B) of a synthetic mesh whose triangles are already resident in HBM.
N = 1: the configuration the metric is quoted on — 10 M-triangle uniform-random mesh, HPLOC (BASELINE.json configs[2]).
N > 1: one process per GPU (launched by torch.distributed.run), every rank builds its own mesh of the same size (different
seed) — the path shards at scene granularity, no data-path collective; the only exchange is an all-gather of the 24-byte
root AABBs (RCCL) per step.  scaling = weak.  value = triangles built by all ranks / time of K steps (max over ranks).

Also reported on the same JSON line:
  roofline     — dominant kernel's algorithmic bytes / its HIP-event time (events recorded on the launch stream inside the
                 timed region) against the 8 TB/s HBM peak; roofline.issue = how busy the VALUs and the LDS pipe are in that kernel
                 (from the committed counter file profiles/issue_counters.json: the second bound of a kernel that is not HBM-bound);
  cpu_baseline — the reference's CPU binned-SAH builder (oracle port, 1 thread) timed on a bounded sample of the same mesh;
  secondary    — (N = 1) the other BASELINE.json configs at their own sizes, 50-build loops after the timed region: Sponza-class 262 144 single-pass LBVH (config 2),
                 Sponza-class 262 144 PLOC++ + the BVH4 collapse (config 4), uniform 2 M HPLOC (config 5's per-GPU mesh), each with the FIRST build of that size on a
                 fresh context beside the warm loop (the library sizes some launch batches from the previous same-size build);
  config5      — (N > 1) the same exchange with config 5's own shape: 2 M triangles per GPU, seed 100 + rank, offset (rank, 0, 0).
Algorithmic bytes of the PLOC-family emit stages are exact per mesh: profiles/algorithmic_bytes.json, written by
tools/algorithmic_bytes.py from the pinned oracle's cluster-load / store counts; the split of the HPLOC emit between its two kernels is the
measured task share of profiles/hploc_task_share.json (tools/measure_task_share.py).  Both are data files: nothing here calls the oracle outside
the cpu_baseline leg.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

# algorithmic bytes per primitive of each kernel (restated in DESIGN.md §5)
KERNEL_BYTES_PER_PRIM = {
    "k_extents": 88.0,            # R Triangle 64 + W Aabb 24
    "k_morton": 32.0,             # R Aabb 24 + W key 4 + W val 4   (this build: 28, value is implicit)
    "k_onesweep": 17.0,           # per pass: R 8 + W 8 (+ hist R 4 amortised over 4 passes)
    "k_hploc": 198.0,             # one-launch HPLOC (n < 800 k): SetupClusters 64 + HPloc 134 (keys 4 + parent xchg 16 + cluster id L/S 18.3 + AABB loads 63.9 + W node 32)
    "k_hploc_block": 177.9,       # block-local kernel: SetupClusters 64 + 85 % of HPloc's 134 (the merge tasks whose range lies inside a 512-leaf tile)
    "k_hploc_ext": 20.1,          # the other 15 % of the merge tasks (ranges crossing tiles)
    "k_lbvh_single": 224.0,
    "k_lbvh_block": 218.2,        # tile scheduler: the 224 of single-pass LBVH by node share — every leaf (R val 4 + gather 64 + W leaf 32 = 100)
                                  # and 95.3 % of the internal nodes' 124 (keys 4 + R 2 children 64 + W 32 + spans 16 + counter 8)
    "k_lbvh_ext": 5.8,            # the other 4.7 % of the internal nodes (ranges crossing tiles)
    "k_karras": 100.0, "k_refit": 88.0,
    # two-pass LBVH on the tile scheduler (same kernels, Karras numbering): its 188 by node share
    "lbvh_two:k_lbvh_block": 183.9, "lbvh_two:k_lbvh_ext": 4.1,
    "k_ploc_iter": 250.0,         # summed over all iterations: SetupClusters 60 (fused into the first iteration) + the iterations' 190
}
# bytes per primitive a kernel of THIS implementation must move through HBM itself (its compulsory traffic, not the reference algorithm's):
# the tile kernel keeps the work lists of its merge tasks in LDS, so its own traffic is R sorted value 4 + key 4 + box gather 24, W PrimRef 28 +
# one 32-byte node per merge it performs (~0.92 per primitive) + the hand-over records (~11).  Reported next to the SURVEY §8(d) figure.
KERNEL_OWN_BYTES_PER_PRIM = {"k_hploc_block": 100.0, "k_hploc_ext": 40.0}
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def _load_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def kernel_source_hash() -> str:
    """sha256[:16] over the gfx950 machine code (.text of every code object) inside the library this process measures: the committed counter files (profiles/*.json
    written by tools/prof_round.sh) carry the hash of the kernels they were measured on, so that a line can say whether its static inputs (PMC traffic, SQ counters,
    rocprof averages) belong to the kernels it timed.  Round 5 hashed the source text, and a default-off #ifdef line added after the profile run marked three unchanged
    profiles stale; the machine code does not change for text the preprocessor drops.  (Falls back to the source text if the library cannot be parsed.)"""
    import hashlib
    import struct
    h = hashlib.sha256()
    try:
        import bvh_pkg
        d = open(bvh_pkg.load().LIB_PATH, "rb").read()

        def sections(elf):      # ELF64 little endian: name -> (offset, size)
            shoff = struct.unpack_from("<Q", elf, 0x28)[0]; entsize, num, strndx = struct.unpack_from("<HHH", elf, 0x3A)
            raw = [struct.unpack_from("<IIQQQQ", elf, shoff + i * entsize) for i in range(num)]
            stro = raw[strndx][4]
            return {elf[stro + r[0]: elf.index(b"\0", stro + r[0])].decode(): (r[4], r[5]) for r in raw}
        off, size = sections(d)[".hip_fatbin"]
        fat = d[off: off + size]
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        found = 0
        at = fat.find(magic)
        while at >= 0:
            n_entries = struct.unpack_from("<Q", fat, at + len(magic))[0]
            q = at + len(magic) + 8
            for _ in range(n_entries):
                eoff, esize, tsize = struct.unpack_from("<QQQ", fat, q)
                triple = fat[q + 24: q + 24 + tsize].decode(); q += 24 + tsize
                if "gfx950" in triple and esize:
                    co = fat[at + eoff: at + eoff + esize]
                    toff, tlen = sections(co)[".text"]
                    h.update(co[toff: toff + tlen]); found += 1
            at = fat.find(magic, at + len(magic))
        if not found:
            raise ValueError("no gfx950 code object")
        return h.hexdigest()[:16]
    except Exception:
        h = hashlib.sha256()
        d = os.path.join(ROOT, "hip-bvh-construction_amd", "csrc")
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".hpp")) or f == "Makefile":
                h.update(open(os.path.join(d, f), "rb").read())
        return "src-" + h.hexdigest()[:12]


def run_secondary(pkg, torch, local):
    """The other BASELINE.json configs at their own sizes (N = 1 only; after the timed region, before the CPU baseline): warm 50-build loops + the cold first build."""
    import ctypes as C
    out = []
    specs = [("sponza_262144_tris_lbvh_single", pkg.ALGO_SINGLEPASS, lambda: pkg.meshgen.sponza_like(262_144, 3), 420.0, False),
             ("sponza_262144_tris_ploc", pkg.ALGO_PLOCPP, lambda: pkg.meshgen.sponza_like(262_144, 3), None, True),
             ("uniform_2000000_tris_hploc", pkg.ALGO_HPLOC, lambda: pkg.meshgen.uniform(2_000_000, 100), None, False)]
    for workload, algo, gen, const_bytes, with_collapse in specs:
        tris = gen(); n = len(tris)
        d_tris = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
        ctx = pkg.Context(local)                                   # a FRESH context: nothing is known about this size
        try:
            ctx.reserve(n)                                          # (allocation is outside every timer, as in the headline loop)
            b = pkg.BUILDERS[algo]()
            torch.cuda.synchronize()
            t0 = time.perf_counter(); b.build(ctx, d_tris, on_device=True, n=n); ctx.synchronize(); cold = (time.perf_counter() - t0) * 1e3
            cold_collapse = None
            if with_collapse:
                ctx.set_profiling(1); b.build(ctx, d_tris, on_device=True, n=n); _, _, cold_collapse = b.collapse4_cost(); ctx.set_profiling(0)
            for _ in range(5):
                b.build(ctx, d_tris, on_device=True, n=n)
            ctx.synchronize()
            steps = 50
            t0 = time.perf_counter()
            for _ in range(steps):
                b.build(ctx, d_tris, on_device=True, n=n)
            ctx.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            ex = exact_bytes(workload)
            per_prim = ex[2] if ex else const_bytes
            e = {"workload": workload, "builder": pkg.ALGO_NAMES[algo], "tris": n, "steps": steps, "ms_per_step": round(ms, 4), "Mtris/s": round(n / ms / 1e3, 1),
                 "pipeline_bytes_per_prim": per_prim, "pipeline_frac": round(per_prim * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if per_prim else None,
                 "cold_first_build_ms": round(cold, 4)}
            if with_collapse:
                ctx.set_profiling(1)
                cms = []
                for _ in range(10):
                    b.build(ctx, d_tris, on_device=True, n=n); cms.append(b.collapse4_cost()[2])
                ctx.set_profiling(0)
                e["bvh_collapse4_ms"] = round(float(np.median(cms)), 4); e["cold_collapse4_ms"] = round(float(cold_collapse), 4)
            ctx.set_profiling(1); b.build(ctx, d_tris, on_device=True, n=n); e["stage_ms"] = {k: round(v, 4) for k, v in dict(b.m_timer).items()}; ctx.set_profiling(0)
            out.append(e)
        finally:
            ctx.close()
    return out


def exact_bytes(workload: str):
    """(emit bytes / prim, SetupClusters bytes / prim, pipeline bytes / prim, source) of a PLOC-family workload from profiles/algorithmic_bytes.json, or None"""
    tab = _load_json("algorithmic_bytes.json") or {}
    e = tab.get(workload)
    return None if not e else (e["emit_bytes_per_prim"], e["setup_bytes_per_prim"], e["pipeline_bytes_per_prim"], e["source"])


def kernel_bytes_per_prim(name: str, algo: str, workload: str):
    """algorithmic bytes per primitive of one kernel of this workload: exact per-mesh figures where the oracle supplies them, SURVEY.md §8(d)'s constants otherwise"""
    ex = exact_bytes(workload)
    if ex and name in ("k_hploc_block", "k_hploc_ext", "k_hploc", "k_ploc_iter"):
        emit, setup = ex[0], ex[1]
        if name in ("k_hploc", "k_ploc_iter"):
            return setup + emit, "exact: " + ex[3]
        share = (_load_json("hploc_task_share.json") or {}).get("tile_kernel_share", 0.85)        # merge tasks run by the tile kernel (measured; round 2 assumed 0.85)
        return (setup + share * emit, f"exact, tile-kernel task share {share}") if name == "k_hploc_block" else ((1.0 - share) * emit, f"exact, tile-kernel task share {share}")
    return KERNEL_BYTES_PER_PRIM.get(f"{algo}:{name}", KERNEL_BYTES_PER_PRIM.get(name, 0.0)), "SURVEY.md 8(d) constant"


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed builds (200 x ~1.5 ms: a 0.3 s timed region, so that one slow launch does not move the number)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tris", type=int, default=10_000_000)
    ap.add_argument("--algo", default="hploc", choices=["hploc", "ploc", "lbvh_single", "lbvh_two"])
    ap.add_argument("--mesh", default="uniform", choices=["uniform", "bunny", "sponza"])
    ap.add_argument("--cpu-sample", type=int, default=10_000_000, help="triangles of the mesh the CPU baseline is timed on (0 = skip); the whole 10 M mesh is ~25 s of single-thread work")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket kernels with HIP events in the timed region")
    ap.add_argument("--no-secondary", dest="secondary", action="store_false", help="skip the other configs' loops (N = 1) / the config-5-shaped exchange (N > 1)")
    args = ap.parse_args()

    import torch
    import bvh_pkg
    pkg = bvh_pkg.load()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    local = local % torch.cuda.device_count()                 # (several ranks may share a GPU in the gloo smoke test)
    torch.cuda.set_device(local)
    dist = None
    backend = os.environ.get("BVH_BENCH_BACKEND", "nccl")     # "nccl" = RCCL over xGMI; "gloo" only to exercise the N>1 path on one GPU
    # BVH_BENCH_FORCE_GATHER=1: run the multi-GPU exchange (process group, staging copy, all-gather, max-over-ranks reduction) even at world size 1,
    # so that the exact code path of an N-GPU run executes on a one-GPU box (tests/test_gpu_round3.py)
    gather = world > 1 or os.environ.get("BVH_BENCH_FORCE_GATHER") == "1"
    if gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)

    algo = {"hploc": pkg.ALGO_HPLOC, "ploc": pkg.ALGO_PLOCPP, "lbvh_single": pkg.ALGO_SINGLEPASS, "lbvh_two": pkg.ALGO_TWOPASS}[args.algo]
    n = args.tris
    seed = 1 + rank
    t0 = time.time()
    if args.mesh == "uniform":
        tris = pkg.meshgen.uniform(n, seed, offset=(float(rank), 0.0, 0.0))
    elif args.mesh == "bunny":
        tris = pkg.meshgen.bunny_like(n, seed + 1)
    else:
        tris = pkg.meshgen.sponza_like(n, seed + 2)
    gen_s = time.time() - t0

    # builds and the RCCL all-gather share ONE torch side stream (the null stream would not order against the ctx's own stream)
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    ctx = pkg.Context(local, side.cuda_stream)
    d_tris = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()      # input resident in HBM before the timed region
    ctx.reserve(n)
    builder = pkg.BUILDERS[algo]()
    root_box = torch.zeros(6, dtype=torch.float32, device="cuda")
    gathered = torch.zeros(6 * world, dtype=torch.float32, device="cuda") if gather else None
    lib = pkg.lib()
    import ctypes as C

    gather_events = []
    step_no = [0]

    def step():
        builder.build(ctx, d_tris, on_device=True, n=n)
        if gather:
            # the exchange is bracketed by HIP events on every 8th step only (like the kernels': an event between two launches costs a few microseconds of launch gap,
            # and these two sat in EVERY step of the N > 1 lines — the N = 1 line has no exchange to bracket)
            timed = step_no[0] % 8 == 0; step_no[0] += 1
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if timed else None
            if timed:
                gather_events.append(ev); ev[0].record(side)
            # root AABB = nodes[root].aabb (24 bytes at offset 8 of the 32-byte node)
            src = builder.result.d_nodes + 32 * builder.result.root + 8
            rc = lib.bvh_dev_copy(ctx.handle, root_box.data_ptr(), src, 24)
            assert rc == 0
            if backend == "nccl":
                dist.all_gather_into_tensor(gathered, root_box)
            else:
                host = root_box.cpu(); out = [torch.zeros(6) for _ in range(world)]
                dist.all_gather(out, host)
            if timed:
                ev[1].record(side)

    def barrier():
        if gather:
            dist.barrier()
        torch.cuda.synchronize()

    # Per-kernel HIP events are recorded on the launch stream INSIDE the timed region, for every 8th build (an event between two launches
    # costs a few microseconds of launch gap — one per launch of every build stretched a 1.42 ms build to 1.50): the kernels' average launch
    # durations and the roofline come from those sampled builds of the timed region — builds 0, 8, 16, ... of it, the first timed build included
    # (round 3 shifted the phase so that the slightly slower first build after the barrier was never sampled; VERDICT / ADVICE r03: not any more).
    ctx.set_profiling(0)
    sample_every = 1 if args.steps < 16 else 8
    for _ in range(args.warmup):
        step()
    barrier()
    if not args.no_kernel_events:
        ctx.set_kernel_sampling(sample_every); ctx.set_profiling(2)
    n_sampled = sum(1 for i in range(args.steps) if i % sample_every == 0)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if gather:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ktimes = {} if args.no_kernel_events else ctx.kernel_times()
    # all-gather of the root boxes: mean / max over the timed steps on this rank, device time incl. the 24-byte staging copy
    gather_us = [a.elapsed_time(b) * 1e3 for a, b in gather_events[-max(1, args.steps // 8):]] if gather_events else []      # (the sampled steps of the timed region)
    if gather and backend == "nccl":      # every rank holds every root box, and this rank's slot is its own tree's root
        torch.cuda.synchronize()
        assert torch.equal(gathered[6 * rank: 6 * rank + 6], root_box), "all-gather of root AABBs is inconsistent"
    ctx.set_kernel_sampling(1); ctx.set_profiling(1)
    builder.build(ctx, d_tris, on_device=True, n=n)          # one extra build with stage events (reference Timer tokens)
    stage = dict(builder.m_timer)
    sah = builder.sah_cost()

    # ---- BASELINE.json config 5 at its own shape (N > 1 lines carry it next to the 10 M weak-scaling value): 2 M triangles per GPU, seed 100 + rank, offset (rank, 0, 0)
    config5 = None
    if gather and args.secondary:
        n5 = 2_000_000
        tris5 = pkg.meshgen.uniform(n5, 100 + rank, offset=(float(rank), 0.0, 0.0))
        d5 = torch.from_numpy(tris5.view(np.uint8).reshape(-1)).cuda()
        ctx.set_profiling(0)
        d_keep, n_keep = d_tris, n
        d_tris, n = d5, n5                                     # (step() builds whatever d_tris / n name)
        for _ in range(5):
            step()
        barrier()
        steps5 = 100
        t0 = time.perf_counter()
        for _ in range(steps5):
            step()
        barrier()
        e5 = time.perf_counter() - t0
        t5 = torch.tensor([e5], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t5, op=dist.ReduceOp.MAX)
        e5 = float(t5.item())
        config5 = {"workload": f"uniform_{n5}_tris_hploc x {world} (seed 100 + rank, offset (rank, 0, 0))", "tris_per_gpu": n5, "steps": steps5,
                   "ms_per_step": round(e5 / steps5 * 1e3, 4), "Mtris/s": round(n5 * world / (e5 / steps5) / 1e6, 1)}
        d_tris, n = d_keep, n_keep
    if rank != 0:
        if gather:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = (n * world) / (elapsed / args.steps) / 1e6
    workload = f"{args.mesh}_{n}_tris_{args.algo}"
    ex = exact_bytes(workload)
    pipeline_bytes = int(round(ex[2] * n)) if ex else int(builder.timings.bytes_algorithmic)
    # ---- roofline of the dominant kernel (largest summed event time)
    roof = None
    if ktimes:
        dom = max(ktimes.items(), key=lambda kv: kv[1][0])
        name, (ms_sum, launches) = dom
        per_build_ms = ms_sum / n_sampled                       # all launches of that kernel in one (sampled) build
        launches_per_build = launches / n_sampled
        per_prim, bytes_source = kernel_bytes_per_prim(name, args.algo, workload)
        alg_bytes = per_prim * n * (launches_per_build if name == "k_onesweep" else 1.0)
        achieved = alg_bytes / (per_build_ms * 1e-3) / 1e9 if per_build_ms > 0 else 0.0
        src_hash = kernel_source_hash()
        pmc = _load_json("pmc_traffic.json") or {}
        traffic = pmc.get(f"{name}@{n}")
        rp = _load_json("rocprof_kernel_avg.json") or {}             # rocprofv3 --kernel-trace --stats averages of this command (tools/prof_round.sh)
        roof = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "avg_launch_ms": round(ms_sum / launches, 4), "launches_per_step": launches_per_build,
                "algorithmic_bytes_per_launch": alg_bytes / max(launches_per_build, 1.0) if name == "k_onesweep" else alg_bytes,
                "algorithmic_bytes_per_prim": round(per_prim, 3), "algorithmic_bytes_source": bytes_source,
                # static inputs of this line (PMC traffic, SQ counters, rocprof average) were measured on the kernel sources with this hash; false = they predate HEAD's kernels
                "profiles_match_kernel_sources": {"pmc_traffic": pmc.get("_kernel_source_hash") == src_hash, "kernel_source_hash": src_hash}}
        rk = rp.get(f"{name}@{n}")
        if rk:                                                      # the same fraction from the rocprofv3 kernel-trace average (all launches of the profiled run, warm-up included)
            roof["frac_rocprof"] = round(alg_bytes / max(launches_per_build, 1.0) / (rk["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if name == "k_onesweep" else round(alg_bytes / (rk["avg_us"] * 1e-6 * max(launches_per_build, 1.0)) / 1e9 / HBM_PEAK_GBS, 4)
            roof["rocprof_avg_us"] = rk["avg_us"]; roof["profiles_match_kernel_sources"]["rocprof"] = rp.get("_kernel_source_hash") == src_hash
        # what the waves of this kernel do with their cycles, straight from the committed SQ counters (profiles/issue_counters.json, a rocprofv3 --pmc pass of this
        # command): no priced estimates (round 4 printed a "VALU busy" that multiplied instruction counts with micro-benchmarked cycle costs; the in-situ probes and the
        # occupancy sweep of profiles/r05_att_hploc_block.md bound that figure to 0.62-0.72 and show that no pipe is what the launch waits for)
        ic = (_load_json("issue_counters.json") or {}).get(f"{name}@{n}")
        if ic:
            cyc = ic["SQ_BUSY_CYCLES"] / ic["shader_engines"]                                    # launch length in shader cycles
            wc = ic["SQ_WAVE_CYCLES"]
            roof["issue"] = {"waves_parked_frac": round(ic["SQ_WAIT_ANY"] / wc, 4),                    # at s_waitcnt / barriers: the dependent chain
                             "waves_ready_not_issued_frac": round(ic["SQ_WAIT_INST_ANY"] / wc, 4),    # lost arbitration / pipe busy
                             "waves_issuing_frac": round(ic["SQ_ACTIVE_INST_ANY"] / wc, 4),
                             # VALU wave-instructions per SIMD per four cycles: a SIMD retires at most 2.0 full-rate (add / mul / mov: 2 cycles per wave64) or 1.0 half-rate
                             # (min / max / compare / select / DPP / fma: 4 cycles) instructions in that time, one wave alone issues at most ~0.95 (profiles/r03_ubench_issue.md)
                             "valu_insts_per_simd_quad_cycle": round(ic["SQ_INSTS_VALU"] * 4.0 / (ic["simds"] * cyc), 4),
                             "resident_waves_per_simd": round(wc * 4.0 / (ic["simds"] * cyc), 2),
                             "source": "profiles/issue_counters.json (direct SQ counter ratios)"}
            if name == "k_hploc_block":       # (the account of what bounds THIS kernel; other dominant kernels carry the counters only)
                roof["issue"]["bound"] = "dependent chain of the PLOC rounds at the residency cap of 8 waves per SIMD (profiles/r05_att_hploc_block.md)"
            roof["profiles_match_kernel_sources"]["issue_counters"] = (_load_json("issue_counters.json") or {}).get("_kernel_source_hash") == src_hash
        if name in KERNEL_OWN_BYTES_PER_PRIM:     # the same kernel against the bytes it really has to move (work lists stay in LDS)
            own = KERNEL_OWN_BYTES_PER_PRIM[name] * n
            roof["own_bytes_per_launch"] = own; roof["own_frac"] = round(own / (per_build_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    # ---- the other configs on the record (N = 1): after the timed region, before the CPU baseline
    secondary = None
    if world == 1 and args.secondary:
        del d_tris; torch.cuda.empty_cache()
        secondary = run_secondary(pkg, torch, local)
    # ---- CPU baseline: reference's binned-SAH builder (oracle port), single thread, bounded sample of the same mesh
    cpu = None
    if args.cpu_sample > 0 and world == 1:                 # (rank 0 at N = 1 only: an N-GPU run does not hold its ranks for 23 s of CPU work)
        import oracle as orc
        m = min(args.cpu_sample, n)
        sample = np.ascontiguousarray(tris[:m])
        t0 = time.perf_counter()
        nodes, total = orc.binned_sah_build(sample)
        dt = time.perf_counter() - t0
        cpu = {"value": round(m / dt / 1e6, 4), "unit": "Mtris/s", "cores": 1, "kind": "port",
               "sample": (f"the whole {m}-triangle benchmark mesh" if m == n else f"first {m} triangles of the benchmark mesh") + f", oracle port of SahBvh::build (src/BinnedSahBvh.cpp:13-203), {dt:.1f} s",
               "sah": round(orc.sah_binned(nodes, total, m)[0], 4)}
    out = {
        "metric": "bvh_build_throughput", "value": round(value, 2), "unit": "Mtris/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32+u32",
        "data": "synthetic",
        "config": {"workload": workload, "builder": pkg.ALGO_NAMES[algo], "tris_per_gpu": n, "mesh": args.mesh,
                   "seed": "1+rank", "parallelism": f"scene-shard x{world}" + (" + allgather(root aabb)" if world > 1 else "")},
        "stage_ms": {k: round(v, 4) for k, v in stage.items()},
        "kernel_ms_per_step": {k: round(v[0] / n_sampled, 4) for k, v in ktimes.items()},     # from the sampled builds of the timed region
        "kernel_event_sampling": f"every {sample_every}th of the {args.steps} timed builds, the first timed build included" + (f" ({n_sampled} builds)" if sample_every > 1 else "")
                                 + "; a sampled build carries one event per launch and is stretched by them: the sum of kernel_ms_per_step exceeds ms_per_step",
        "sah_bvh2": round(sah, 4),
        "pipeline_roofline": {"algorithmic_bytes": pipeline_bytes, "source": ("exact: " + ex[3]) if ex else "SURVEY.md 8(d) constants (bvh_timings.bytes_algorithmic)",
                              "achieved_GBs": round(pipeline_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                              "frac": round(pipeline_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        "roofline": roof, "cpu_baseline": cpu, "secondary": secondary, "config5": config5, "mesh_gen_s": round(gen_s, 2),
        "allgather_us": ({"mean": round(float(np.mean(gather_us)), 2), "max": round(float(np.max(gather_us)), 2), "bytes_per_rank": 24} if gather_us else None),
    }
    print(json.dumps(out))
    if gather:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
