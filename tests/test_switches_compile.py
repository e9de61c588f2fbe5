"""CPU: the A/B switches of the kernel sources still compile (off-by-default paths live in production sources and nothing built them).
Device-only compiles of gfx950 code objects with sets of non-default switches — a few seconds each; nothing is run (the measured verdict of every
switch is in LEADS.md).  Round 5 pruned every switch whose verdict was "dropped / no effect / measured slower" out of the sources (tools/probes/r05_pruned_switches.patch
keeps them); what is left: the alt variant, the tunables and the measurement builds, each compiled here.  test_no_unlisted_switch keeps the list honest."""
import os
import subprocess
import tempfile

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "hip-bvh-construction_amd", "csrc")
BASE = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
        "--cuda-device-only", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-Wno-unused-value", "-Wno-unused-result", "-Wno-pass-failed"]
EMIT = ["-fno-honor-nans", "-mno-amdgpu-ieee"]

SETS = [
    # the alt variant that tests/test_gpu_variants.py also RUNS (same trees), and the measurement builds behind profiles/ (round clock, tile phases, sensitivity probes)
    ("hploc.hip", EMIT + ["-fno-slp-vectorize"], ["-DHP_NN_LDS=1", "-DHPB_WIDE=1", "-DHPB_IL=0"]),
    ("hploc.hip", EMIT + ["-fno-slp-vectorize"], ["-DBVH_ABLATION", "-DABL_ROUND_CLOCK", "-DABL_LDS_PAD=16384", "-DABL_TILE_PHASES=2"]),
    ("hploc.hip", EMIT + ["-fno-slp-vectorize"], ["-DBVH_ABLATION", "-DABL_EXTRA_VALU", "-DABL_EXTRA_BPERM", "-DABL_EXTRA_TRIP", "-DABL_EXT_TRACE", "-DHPB_OCC=7", "-DHPX_OCC=5", "-DHPB_CUT=128"]),
    ("hploc.hip", EMIT + ["-fno-slp-vectorize"], ["-DBVH_ABLATION", "-DABL_EXT_TIMING"]),
    # the overlapped schedule's shapes of profiles/r06_live_timeline.md (LEADS.md row 87)
    ("hploc.hip", EMIT + ["-fno-slp-vectorize"], ["-DBVH_ABLATION", "-DHPB_LIVE_PAD=2560", "-DHPL_OCC=8", "-DHPL_GRID=1024u", "-DHPL_SLEEP=120"]),
    ("ploc.hip", EMIT + ["-fno-slp-vectorize"], ["-DPLOC_NN_OWN_F64=0", "-DPLOC_TAIL_PAIRS=0", "-DPLOC_ABL=1", "-DPLOC_OCC=5", "-DPLOC_STATIC_G=0"]),
    ("sort.hip", [], ["-DBVH_ABLATION", "-DSORT_WIDE_IPT=12", "-DSORT_HELP_AFTER=64u"]),
    ("lbvh.hip", EMIT, ["-DLBVH_EXT_MAX_SHIFT=6", "-DBVH_ABLATION"]),
    ("api.hip", [], ["-DSORT_GATE_TOP=0", "-DBVH_ABLATION"]),
]


@pytest.mark.parametrize("src,flags,switches", SETS, ids=[f"{s[0]}:{' '.join(s[2])}" for s in SETS])
def test_switch_set_compiles(src, flags, switches):
    if not os.path.exists(BASE[0]):
        pytest.skip("hipcc not installed")
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run(BASE + flags + switches + ["-c", os.path.join(CSRC, src), "-o", os.path.join(tmp, "x.o")], capture_output=True, text=True, cwd=CSRC)
        assert r.returncode == 0, r.stderr[-3000:]


def test_no_unlisted_switch():
    """every #if / #ifdef / #ifndef macro of csrc/ is a tunable with a default, a measurement-build flag exercised above, or platform plumbing — a new A/B switch has to be
    added to one of these lists (and thereby to the compiled sets), or be removed once measured."""
    import re
    known = {  # tunables (#ifndef X / #define X default) and the measurement flags
        "BVH_ABLATION", "HP_NN_LDS", "HPB_WIDE", "HPB_IL", "HPA_OCC", "HPB_OCC", "HPB_OCC_1024", "HPB_OCC64", "HPX_OCC", "HPB_T", "HPB_NT", "HPX_GRID",
        "HPB_LIVE_PAD", "HPL_OCC", "HPL_GRID", "HPL_SLEEP",
        "ABL_ROUND_CLOCK", "ABL_LDS_PAD", "HPB_CUT", "ABL_TILE_PHASES", "ABL_EXTRA_VALU", "ABL_EXTRA_BPERM", "ABL_EXTRA_TRIP", "ABL_EXT_TRACE", "ABL_EXT_TIMING",
        "PLOC_NARROW", "PLOC_NN_OWN_F64", "PLOC_TAIL_PAIRS", "PLOC_ONE_SHOT_MAX_N", "PLOC_ABL", "PLOC_OCC", "PLOC_STATIC_G",
        "SORT_HELP_AFTER", "SORT_WIDE_IPT", "BVH_SORT_IPT", "BVH_SORT_WIDE_MIN_N", "SORT_GATE_TOP", "LEAF_FROM_TRIS",
        "EM_PPT", "EX_PPT", "MORTON_GROUP", "MORTON64_GROUP", "SORT_HIST_GROUP", "LBVH_TILE_SIZE", "LBVH_EXT_MAX_SHIFT", "__x86_64__"}
    found = set()
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".hpp")):
            for line in open(os.path.join(CSRC, f)):
                m = re.match(r"\s*#\s*(?:if|ifdef|ifndef|elif)\b(.*)", line)
                if m:
                    found |= set(re.findall(r"\b[A-Za-z_][A-Za-z0-9_]*\b", re.sub(r"//.*", "", m.group(1)))) - {"defined"}
    assert found <= known, sorted(found - known)
    sites = sum(1 for line in open(os.path.join(CSRC, "hploc.hip")) if re.match(r"\s*#\s*(if|ifdef|ifndef)\b", line))
    assert sites <= 48, sites
