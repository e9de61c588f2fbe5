#!/usr/bin/env python3
"""Copy the judged files of the last tools/prof_round.sh run (gpurun_out/prof_round/) into profiles/ and rewrite the measured sections of profiles/rNN_summary.md
(bench line, kernel trace, HBM traffic, SQ counters) from them.  python tools/refresh_profiles.py [r04]"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
O = os.path.join(ROOT, "gpurun_out", "prof_round"); P = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(O, "bench.json"), os.path.join(P, f"{rnd}_bench.json"))
for f in ("pmc_traffic.json", "issue_counters.json", "rocprof_kernel_avg.json"): shutil.copy(os.path.join(O, f), os.path.join(P, f))
b = json.load(open(os.path.join(O, "bench.json")))
path = os.path.join(P, f"{rnd}_summary.md"); s = open(path).read()
i0 = s.index("## bench line"); i1 = s.index("## Reading")
rf = b["roofline"]
sec = ["## bench line (`profiles/%s_bench.json`)\n\n" % rnd,
       f"`value` {b['value']} Mtris/s, `ms_per_step` {b['ms_per_step']}, kernels {b['kernel_ms_per_step']}\n\n",
       "`roofline` (`profiles_match_kernel_sources` compares the counter files that were on the box DURING this run with the sources; the files committed beside this one come "
       "from this run and carry the hash `%s` of the sources): %s\n\n" % (rf["profiles_match_kernel_sources"]["kernel_source_hash"], json.dumps(rf)),
       "`pipeline_roofline`: %s\n\n" % json.dumps(b["pipeline_roofline"]), "`cpu_baseline`: %s\n\n" % json.dumps(b["cpu_baseline"]),
       "`secondary`:\n" + "".join("* %s\n" % json.dumps(x) for x in b["secondary"]) + "\n",
       "## rocprofv3 --kernel-trace --stats (the bench command itself without its CPU leg and secondary loops: 206 builds = 5 warm-up + 200 timed + 1 stage-timed)\n\n" + open(os.path.join(O, "stats.md")).read().strip() + "\n\n",
       "## HBM traffic per build (PMC: 2 x FETCH_SIZE + WRITE_SIZE, KB units; MI355X_MICROARCH.md's gfx950 correction)\n\n" + open(os.path.join(O, "traffic.md")).read().strip() + "\n\n",
       "## issue (SQ counters of the bench command: direct ratios)\n\n" + open(os.path.join(O, "issue.md")).read().strip() + "\n\n"]
open(path, "w").write(s[:i0] + "".join(sec) + s[i1:])
print(b["value"], b["ms_per_step"], b["kernel_ms_per_step"], rf["frac"], rf.get("frac_rocprof"))
