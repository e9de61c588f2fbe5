#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (results.db): per-kernel time stats and, if present, PMC counter sums per kernel.
Usage: tools/rocpd_summary.py <results.db> [--filter substring]   -> markdown on stdout (kept under profiles/)."""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else None
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), max(lds_size), "
                       "max(workgroup_x), avg(grid_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total us | avg us | min us | max us | % | vgpr | sgpr | lds B | wg | grid |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        if flt and flt not in r[0]:
            continue
        name = r[0].split("(")[0][-60:]
        print(f"| {name} | {r[1]} | {r[2]/1e3:.1f} | {r[3]/1e3:.2f} | {r[4]/1e3:.2f} | {r[5]/1e3:.2f} | {100*r[2]/total:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {int(r[10])} |")
    try:
        pm = cur.execute("select k.name, p.name, count(*), sum(e.value), avg(e.value) from pmc_events e join pmc_info p on e.pmc_id = p.id "
                         "join kernels k on e.event_id = k.id group by k.name, p.name").fetchall()
    except Exception as ex:  # schema differences between rocprofv3 versions
        pm = []
        try:
            cols = [d[1] for d in cur.execute("pragma table_info('counters_collection')")]
            name_col = "kernel_name" if "kernel_name" in cols else "name"
            pm = cur.execute(f"select {name_col}, counter_name, count(*), sum(value), avg(value) from counters_collection group by {name_col}, counter_name").fetchall()
        except Exception as ex2:
            print(f"\n(no PMC data: {ex} / {ex2})")
    if pm:
        print("\n| kernel | counter | dispatches | sum | avg per dispatch |")
        print("|---|---|---|---|---|")
        for r in pm:
            if flt and flt not in r[0]:
                continue
            print(f"| {r[0].split('(')[0][-60:]} | {r[1]} | {r[2]} | {r[3]:.6g} | {r[4]:.6g} |")


if __name__ == "__main__":
    main()
