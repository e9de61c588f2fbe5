#!/usr/bin/env python3
"""Aggregates a rocprofv3 PC-sampling CSV (stochastic or host-trap) per kernel and per instruction.  python tools/pcs_summary.py SAMPLES.csv [KERNEL_TRACE.csv]
Columns differ between rocprofv3 versions: everything is looked up by (case-insensitive) header name and missing columns are tolerated."""
import collections
import csv
import re
import sys

csv.field_size_limit(1 << 30)
path = sys.argv[1]
ktrace = sys.argv[2] if len(sys.argv) > 2 else None
disp2kernel = {}
if ktrace:
    with open(ktrace, newline="") as f:
        for row in csv.DictReader(f):
            low = {k.lower(): v for k, v in row.items()}
            disp2kernel[low.get("dispatch_id")] = low.get("kernel_name", "?")


def short(k):
    m = re.search(r"(k_[a-z0-9_]+)", k or "")
    return m.group(1) if m else (k or "?")[:40]


rows = 0
per_kernel = collections.Counter()
issued = collections.defaultdict(collections.Counter)          # kernel -> {issued / not}
reason = collections.defaultdict(collections.Counter)          # kernel -> stall reason
itype = collections.defaultdict(collections.Counter)           # kernel -> instruction type (of issued samples)
inst = collections.defaultdict(lambda: collections.defaultdict(collections.Counter))   # kernel -> (instruction, comment) -> {reason or 'issued'}
lanes = collections.defaultdict(collections.Counter)           # kernel -> active-lane bucket
arb = collections.defaultdict(collections.Counter)
with open(path, newline="") as f:
    rd = csv.DictReader(f)
    hdr = [h.lower() for h in rd.fieldnames]
    print("columns:", rd.fieldnames)
    for row in rd:
        low = {k.lower(): v for k, v in row.items()}
        rows += 1
        k = short(disp2kernel.get(low.get("dispatch_id"), low.get("kernel_name", "?")))
        per_kernel[k] += 1
        ins = low.get("instruction", "?"); cm = low.get("instruction_comment", "")
        wi = low.get("wave_issued_instruction", low.get("wave_issued", ""))
        rs = low.get("stall_reason", low.get("reason_not_issued", ""))
        tag = "issued" if str(wi).strip() in ("1", "True", "true") else (rs or "not issued")
        issued[k]["issued" if tag == "issued" else "not issued"] += 1
        if tag != "issued":
            reason[k][rs or "?"] += 1
        else:
            itype[k][low.get("instruction_type", "?")] += 1
        inst[k][(ins, cm)][tag] += 1
        em = low.get("exec_mask")
        if em:
            try:
                c = bin(int(em, 0) if em.lower().startswith("0x") else int(em)).count("1")
                lanes[k]["%2d-%2d" % (c // 8 * 8, c // 8 * 8 + 7) if c < 64 else "64"] += 1
            except ValueError:
                pass
        for col in hdr:
            if col.startswith("arb_state") or col.startswith("arb"):
                arb[k][col + "=" + low[col]] += 1
print(f"{rows} samples")
for k, n in per_kernel.most_common(8):
    print(f"\n## {k}: {n} samples")
    tot = sum(issued[k].values())
    print("  issued:", {a: f"{b} ({100 * b / tot:.1f} %)" for a, b in issued[k].items()})
    print("  not issued, by reason:", {a: f"{100 * b / tot:.1f} %" for a, b in reason[k].most_common()})
    print("  issued, by instruction type:", {a: f"{100 * b / tot:.1f} %" for a, b in itype[k].most_common()})
    print("  active lanes of the sampled wave:", {a: f"{100 * b / tot:.1f} %" for a, b in sorted(lanes[k].items())})
    if arb[k]:
        print("  arbiter columns:", {a: f"{100 * b / tot:.1f} %" for a, b in arb[k].most_common(40)})
    print("  | % of samples | issued % | top reasons when not issued | instruction | source |")
    print("  |---|---|---|---|---|")
    for (ins, cm), c in sorted(inst[k].items(), key=lambda kv: -sum(kv[1].values()))[:60]:
        t = sum(c.values())
        rs = ", ".join(f"{a} {100 * b / t:.0f}%" for a, b in c.most_common(4) if a != "issued")
        print(f"  | {100 * t / n:.2f} | {100 * c['issued'] / t:.0f} | {rs} | `{ins}` | {cm[-70:]} |")
