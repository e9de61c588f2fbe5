#!/usr/bin/env python3
"""model_hploc.py — discrete-event model of k_hploc_block's level loop.

MEASUREMENT / DECISION INFRASTRUCTURE (imports the CPU oracle to obtain the merge tasks; never part of the product or of bench.py's timed path).

What it does.  The oracle's merge tasks of a mesh ({L, R, split, rounds} per plocMerge call: oracle.hploc_tasks) are grouped into the 512-leaf tiles of
k_hploc_block; tasks whose range crosses a tile boundary belong to k_hploc_ext and are dropped.  A tile's tasks are replayed on a model of one MI355X:
256 CUs x 4 SIMDs, a tile (workgroup) takes a CU slot when one is free (7 per CU), spends a fixed pre-phase there (leaf staging + range searches: memory
latency, no issue load), then runs its level loop, then a fixed post-phase (hand-over).  The level loop is simulated round by round:

  round duration = LAT + VALU x (1 + C x (a - 1) x duty)          a = waves executing a round on the same SIMD when the round starts, duty = VALU / (LAT + VALU)

LAT + VALU is the measured lone-wave round (profiles/r03_ext_chain.md: 2000-2500 cycles; 245 VALU instructions at the lone-wave issue rate of 4.5 cycles plus three
dependent LDS round trips), VALU the SIMD time of a round's instructions at the saturated-pipe prices of profiles/r03_ubench_issue.md (3.3 cycles average).  C is the
ONE free parameter of today's scheduler, fitted so that the 10 M level loop matches its measurement; 2 M and 40 M are then predictions.

Schedulers:
  levels    today: four waves per tile, tasks of one hierarchy level dealt in pairs (size-class sorted), __syncthreads between levels (waiting costs nothing);
  deps      the same four waves and static deal, but a task starts when its own children are done (no barrier; a waiting wave polls: POLL issue load on its SIMD).
            Built and measured in round 4 (HPB_DEPS=1: +0.02 ... +0.03 ms): POLL is fitted to that.
  pool(k)   the VERDICT's proposal: a workgroup of 4 k waves owns k tiles' lists and deals ready tasks of ANY of its tiles to idle waves (dependency driven, polling).

Usage: python tools/model_hploc.py [--n 2000000 10000000 40000000] [--fit]      -> profiles/r04_model_hploc.md (printed; redirect)"""
from __future__ import annotations

import argparse
import heapq
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))

TILE = 512
CUS, SIMDS, SLOTS = 256, 4, 7
GHZ = 2.4
LAT, VALU = 1200.0, 808.0          # cycles of a lone round: latency part (three LDS round trips, 16 LDS atomics) / issue part (245 VALU instructions x 3.3 cycles)
PASS_LAT, PASS_VALU = 350.0, 120.0  # per wave pass: work-list assembly from LDS (ballot, left-pack, one LDS round trip) and the tail (invalidate, loop)
BARRIER = 60.0


def tile_tasks(n: int, seed: int = 1):
    """per tile: list of (level, size, rounds, gap, left child gap or -1, right child gap or -1) of the tile-local merge tasks of uniform(n, seed)"""
    import bvh_pkg
    import oracle as orc
    pkg = bvh_pkg.load()
    tris = pkg.meshgen.uniform(n, seed)
    fe = orc.front_end(tris)
    tk = orc.hploc_tasks(fe["boxes"], fe["skeys"], fe["svals"])
    L, R, split, rounds = (tk[:, i].astype(np.int64) for i in range(4))
    local = (L // TILE) == (R // TILE)
    keys = fe["skeys"].astype(np.uint64)
    p = split - 1                                                                        # the task's LBVH gap
    aug = (keys << np.uint64(32)) | np.arange(len(keys), dtype=np.uint64)
    x = aug[p] ^ aug[p + 1]
    plen = 64 - np.floor(np.log2(x.astype(np.float64))).astype(np.int64) - 1            # common prefix length (exact for these magnitudes: the xor's top bit)
    tiles = {}
    for i in np.nonzero(local)[0]:
        tiles.setdefault(int(L[i] // TILE), []).append((int(plen[i]), int(R[i] - L[i] + 1), int(rounds[i]), int(p[i]), int(L[i]), int(R[i])))
    out = []
    n_tiles = (n + TILE - 1) // TILE
    for t in range(n_tiles):
        ts = tiles.get(t, [])
        # children: the tile-local tasks whose range is [L, p] / [p + 1, R]
        by_range = {(a[4], a[5]): k for k, a in enumerate(ts)}
        recs = []
        for (pl, size, rd, gap, l, r) in ts:
            recs.append((pl, size, rd, by_range.get((l, gap), -1), by_range.get((gap + 1, r), -1)))
        out.append(recs)
    return out, int(local.sum()), len(tk)


class Chip:
    """CU slots + per-SIMD count of waves inside a round + per-SIMD polling load"""

    def __init__(self, c, poll):
        self.c, self.poll = c, poll
        self.active = np.zeros((CUS, SIMDS), dtype=np.int32)
        self.polling = np.zeros((CUS, SIMDS), dtype=np.float64)
        self.next_simd = np.zeros(CUS, dtype=np.int32)

    def round_time(self, cu, simd, rounds):
        a = self.active[cu, simd] + 1
        duty = VALU / (LAT + VALU)
        load = self.c * (a - 1) * duty + self.poll * self.polling[cu, simd]
        return rounds * (LAT + VALU * (1.0 + load)) + PASS_LAT + PASS_VALU * (1.0 + load)


def simulate(tiles, sched: str, c: float, poll: float, pre: float, post: float, pool: int = 1):
    """returns the launch length in cycles.  sched: 'levels' | 'deps' | 'pool' (pool = tiles per workgroup)"""
    chip = Chip(c, poll)
    k = pool if sched == "pool" else 1
    groups = [tiles[i:i + k] for i in range(0, len(tiles), k)]
    slots_per_cu = SLOTS // k
    free = [(0.0, cu) for cu in range(CUS) for _ in range(slots_per_cu)]
    heapq.heapify(free)
    ev = []                     # (time, seq, kind, payload)
    seq = 0
    end_time = 0.0
    gi = 0
    # workgroups are dispatched in order to the earliest free slot
    while gi < len(groups) or ev:
        # start as many groups as slots are free at the time of the next event
        if gi < len(groups) and (not ev or free and free[0][0] <= ev[0][0]):
            t0, cu = heapq.heappop(free)
            g = groups[gi]; gi += 1
            simd0 = int(chip.next_simd[cu]); chip.next_simd[cu] = (simd0 + 1) % SIMDS
            st = GroupState(g, cu, simd0, sched, 4 * k)
            heapq.heappush(ev, (t0 + pre, seq, "start", st)); seq += 1
            continue
        t, _, kind, st = heapq.heappop(ev)
        if kind == "start":
            for w in range(st.nw):
                seq = st.wave_next(chip, t, w, ev, seq)
            if st.done_waves == st.nw:
                heapq.heappush(ev, (t + post, seq, "exit", st)); seq += 1
        elif isinstance(kind, tuple):      # ("end", wave)
            w = kind[1]
            chip.active[st.cu, st.simd(w)] -= 1
            st.finish_pass(w)
            seq = st.wake(chip, t, ev, seq)
            seq = st.wave_next(chip, t, w, ev, seq)
            if st.done_waves == st.nw and not st.exited:
                st.exited = True
                heapq.heappush(ev, (t + post, seq, "exit", st)); seq += 1
        elif kind == "exit":
            end_time = max(end_time, t)
            heapq.heappush(free, (t, st.cu))
    return end_time


class GroupState:
    """one workgroup: its tiles' tasks and its waves"""

    def __init__(self, group, cu, simd0, sched, nw):
        self.cu, self.simd0, self.sched, self.nw = cu, simd0, sched, nw
        self.exited = False
        self.done_waves = 0
        self.wave_done = [False] * nw
        self.waiting = [False] * nw            # polling for a dependency
        self.cur = [None] * nw                 # tasks of the wave's running pass
        # flatten tasks: (tile, level, size, rounds, childL, childR)
        self.tasks = []
        base = []
        for ti, ts in enumerate(group):
            base.append(len(self.tasks))
            for (pl, size, rd, cl, cr) in ts:
                self.tasks.append([ti, pl, size, rd, cl, cr])
        for tk in self.tasks:                  # child indices -> global task indices of this group
            b = base[tk[0]]
            tk[4] = tk[4] + b if tk[4] >= 0 else -1
            tk[5] = tk[5] + b if tk[5] >= 0 else -1
        self.finished = [False] * len(self.tasks)
        nt = len(group)
        if sched in ("levels", "deps"):
            # per tile (= this group): levels deepest first, size-class order inside a level, pairs dealt to waves 0..3 with stride 4 pairs
            order = sorted(range(len(self.tasks)), key=lambda i: (-self.tasks[i][1], cls(self.tasks[i][2]), i))
            self.levels = []
            for i in order:
                if not self.levels or self.tasks[self.levels[-1][0]][1] != self.tasks[i][1]:
                    self.levels.append([])
                self.levels[-1].append(i)
            # each wave's sequence of passes: list of (level index, [task ids])
            self.seq = [[] for _ in range(nw)]
            for li, lv in enumerate(self.levels):
                for pi in range(0, len(lv), 2):
                    self.seq[(pi // 2) % nw].append((li, lv[pi:pi + 2]))
            self.pos = [0] * nw
            self.level_left = [len(lv) for lv in self.levels]
            self.at_barrier = {}
            self.cur_level = 0
        else:
            # pool: ready tasks in (deepest level first, size class) order
            self.started = [False] * len(self.tasks)
            self.order = sorted(range(len(self.tasks)), key=lambda i: (-self.tasks[i][1], cls(self.tasks[i][2]), i))
            self.left = len(self.tasks)

    def simd(self, w):
        return (self.simd0 + w) % SIMDS

    def ready(self, i):
        t = self.tasks[i]
        return (t[4] < 0 or self.finished[t[4]]) and (t[5] < 0 or self.finished[t[5]])

    def finish_pass(self, w):
        for i in self.cur[w]:
            self.finished[i] = True
            if self.sched in ("levels", "deps"):
                pass
            else:
                self.left -= 1
        self.cur[w] = None

    def set_wait(self, chip, w, on):
        if self.waiting[w] != on:
            self.waiting[w] = on
            chip.polling[self.cu, self.simd(w)] += 1.0 if on else -1.0

    def wake(self, chip, t, ev, seq):
        """a pass finished: idle waves (at a barrier, or polling for a dependency) look again"""
        for w in range(self.nw):
            if self.cur[w] is None and not self.wave_done[w]:
                seq = self.wave_next(chip, t, w, ev, seq, woken=True)
        return seq

    def wave_next(self, chip, t, w, ev, seq, woken=False):
        if self.wave_done[w] or self.cur[w] is not None:
            return seq
        if self.sched == "levels":
            # a wave may start the passes of level li only when every pass of the levels before is finished (the barrier)
            if self.pos[w] >= len(self.seq[w]):
                self.wave_done[w] = True; self.done_waves += 1; self.at_barrier.pop(w, None)
                return seq
            li, ids = self.seq[w][self.pos[w]]
            if any(not all(self.finished[i] for i in self.levels[l]) for l in range(li)):
                self.at_barrier[w] = li                   # waits at the barrier: no issue load
                return seq
            self.at_barrier.pop(w, None)
            return self.start_pass(chip, t + (BARRIER if woken else 0.0), w, ids, ev, seq, adv=True)
        if self.sched == "deps":
            if self.pos[w] >= len(self.seq[w]):
                self.set_wait(chip, w, False)
                self.wave_done[w] = True; self.done_waves += 1
                return seq
            li, ids = self.seq[w][self.pos[w]]
            if not all(self.ready(i) for i in ids):
                self.set_wait(chip, w, True)              # polls
                return seq
            self.set_wait(chip, w, False)
            return self.start_pass(chip, t, w, ids, ev, seq, adv=True)
        # pool
        if self.left == 0 or all(self.started):
            self.set_wait(chip, w, False)
            self.wave_done[w] = True; self.done_waves += 1
            return seq
        ids = []
        for i in self.order:
            if not self.started[i] and self.ready(i):
                ids.append(i)
                if len(ids) == 2:
                    break
        if not ids:
            self.set_wait(chip, w, True)
            return seq
        self.set_wait(chip, w, False)
        for i in ids:
            self.started[i] = True
        return self.start_pass(chip, t, w, ids, ev, seq, adv=False)

    def start_pass(self, chip, t, w, ids, ev, seq, adv):
        if adv:
            self.pos[w] += 1
        self.cur[w] = ids
        rounds = max(self.tasks[i][3] for i in ids)
        dur = chip.round_time(self.cu, self.simd(w), rounds)
        chip.active[self.cu, self.simd(w)] += 1
        heapq.heappush(ev, (t + dur, seq, ("end", w), self)); seq += 1
        return seq


def cls(size):
    return 4 if size > 32 else (size - 17) >> 2


def run(tiles, sched, c, poll, pre, post, pool=1):
    t = simulate(tiles, sched, c, poll, pre, post, pool)
    return t / (GHZ * 1e6)          # ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[2_000_000, 10_000_000])
    ap.add_argument("--c", type=float, default=None, help="contention factor (default: fitted on the 10 M measurement)")
    ap.add_argument("--poll", type=float, default=None)
    ap.add_argument("--measured", type=float, nargs="+", default=None, help="measured level-loop ms for --n (same order)")
    args = ap.parse_args()
    # measured on the MI355X, round 4 (tools/phases_sizes.sh, ablation build, min of 20 launches, ms): the kernel stopped before its level loop (BVH_HPLOC_DEBUG=2),
    # after it (=3) and after the hand-over (=4).  level loop = (3) - (2)
    phases = {2_000_000: (0.057, 0.153, 0.165), 10_000_000: (0.268, 0.594, 0.633), 40_000_000: (1.256, 2.315, 2.455)}
    measured = {n: round(p[1] - p[0], 4) for n, p in phases.items()}
    if args.measured:
        measured.update(dict(zip(args.n, args.measured)))
    data = {}
    for n in sorted(set(args.n) | {10_000_000}):
        t0 = time.time()
        tiles, n_local, n_all = tile_tasks(n)
        data[n] = tiles
        rounds = sum(t[2] for ts in tiles for t in ts)
        print(f"# uniform({n}, 1): {len(tiles)} tiles, {n_local} tile-local merge tasks of {n_all} ({n_local / n_all:.4f}), {rounds / max(n_local, 1):.3f} rounds per task  [{time.time() - t0:.1f} s]", flush=True)
    # pre / post phases: a slot's residence time outside the level loop, from the launches that stop before / after it (a slot holds len(tiles) / 1792 tiles in a row)
    n10 = 10_000_000
    pre_of, post_of = {}, {}
    for n in data:
        ph = phases.get(n, phases[n10])
        gens = len(data[n]) / (CUS * SLOTS) if n in phases else len(data[n10]) / (CUS * SLOTS)
        pre_of[n] = ph[0] * 1e-3 * GHZ * 1e9 / gens
        post_of[n] = (ph[2] - ph[1]) * 1e-3 * GHZ * 1e9 / gens
    pre, post = pre_of[n10], post_of[n10]
    base_only = {n: run([[] for _ in data[n]], "levels", 0.0, 0.0, pre_of[n], post_of[n]) for n in data}
    c = args.c
    if c is None:                                  # fit C on the 10 M level loop by bisection
        lo, hi = 0.0, 6.0
        for _ in range(18):
            mid = 0.5 * (lo + hi)
            v = run(data[n10], "levels", mid, 0.0, pre, post) - base_only[n10]
            if v < measured[n10]:
                lo = mid
            else:
                hi = mid
        c = 0.5 * (lo + hi)
    poll = args.poll
    if poll is None:                               # fit POLL on the measured cost of the barrier-free tile scheduler (HPB_DEPS=1: +0.025 ms at 10 M)
        target = run(data[n10], "levels", c, 0.0, pre, post) + 0.025
        lo, hi = 0.0, 3.0
        for _ in range(14):
            mid = 0.5 * (lo + hi)
            if run(data[n10], "deps", c, mid, pre, post) < target:
                lo = mid
            else:
                hi = mid
        poll = 0.5 * (lo + hi)
    print(f"# parameters: LAT {LAT:.0f} + VALU {VALU:.0f} cycles per lone round, pass overhead {PASS_LAT:.0f} + {PASS_VALU:.0f}, pre-phase {pre:.0f} cycles per tile, C = {c:.3f} (fitted at 10 M), POLL = {poll:.3f} (fitted on HPB_DEPS=1)")
    print("| n | scheduler | launch ms | level loop ms (launch - launch without level loop) | measured level loop ms | model / measured |")
    print("|---|---|---|---|---|---|")
    for n in sorted(data):
        tiles = data[n]; pre, post = pre_of[n], post_of[n]
        rows = [("levels (today)", run(tiles, "levels", c, 0.0, pre, post)),
                ("deps (HPB_DEPS=1, polling)", run(tiles, "deps", c, poll, pre, post)),
                ("deps, polling free", run(tiles, "deps", c, 0.0, pre, post)),
                ("pool of 2 tiles / 8 waves, polling", run(tiles, "pool", c, poll, pre, post, 2)),
                ("pool of 2 tiles / 8 waves, polling free", run(tiles, "pool", c, 0.0, pre, post, 2)),
                ("pool of 3 tiles / 12 waves, polling", run(tiles, "pool", c, poll, pre, post, 3)),
                ("pool of 3 tiles / 12 waves, polling free", run(tiles, "pool", c, 0.0, pre, post, 3))]
        for name, ms in rows:
            lvl = ms - base_only[n]
            m = measured.get(n) if name.startswith("levels") else None
            print(f"| {n} | {name} | {ms:.4f} | {lvl:.4f} | {'' if m is None else f'{m:.3f}'} | {'' if m is None else f'{lvl / m:.3f}'} |", flush=True)


if __name__ == "__main__":
    main()
