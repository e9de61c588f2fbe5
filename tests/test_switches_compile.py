"""CPU: the A/B switches of the kernel sources still compile (off-by-default paths live in production sources and nothing built them).
Device-only compiles of gfx950 code objects with sets of non-default switches — a few seconds each; nothing is run (the measured verdict of every
switch is in DESIGN.md section 9).  The sets avoid the combinations the sources reject by static_assert (HPB_DEPS needs HPB_LEAN = 0)."""
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
    ("hploc.hip", EMIT + ["-fno-slp-vectorize"], ["-DHP_NN_LDS=1", "-DHPB_IL=1", "-DHPX_NO_HOIST=1", "-DHPB_NO_HOIST=1", "-DHPB_SEARCH_BOTH=1"]),
    ("hploc.hip", EMIT + ["-fno-slp-vectorize"], ["-DHPB_LEAN=0", "-DHPB_OCC=7", "-DHPB_DEPS=1", "-DHP_NN_BF=1", "-DHPB_WIDE=1", "-DHPX_LDS_LIST=1", "-DHPB_PRIO=1", "-DHPB_ROT=1"]),
    ("hploc.hip", EMIT + ["-fno-slp-vectorize"], ["-DHP_NN_LDS=2", "-DHP_NN_SCALAR=0", "-DHPB_PREPROBE=0", "-DHPB_PAIR_SORT=0", "-DHPB_STAGE_SERIAL_GATHER=0", "-DBVH_ABLATION"]),
    ("hploc.hip", EMIT + ["-fno-slp-vectorize"], ["-DHP_NN_LDS=0", "-DHPB_LEAN=0", "-DHPB_OCC=7"]),
    ("ploc.hip", EMIT + ["-fno-slp-vectorize"], ["-DPLOC_NN_OWN_F64=0", "-DPLOC_TAIL_PAIRS=0", "-DPLOC_ONE_SHOT=0", "-DPLOC_DEFER=0", "-DPLOC_LATE=1"]),
    ("sort.hip", [], ["-DSORT_NT=3", "-DSORT_PRIO=1", "-DSORT_EARLY_PUBLISH=1", "-DSORT_EXCHANGE_FIRST=1", "-DBVH_ABLATION"]),
    ("lbvh.hip", EMIT, ["-DLBVH_EXT_MAX_SHIFT=6", "-DBVH_ABLATION"]),
]


@pytest.mark.parametrize("src,flags,switches", SETS, ids=[f"{s[0]}:{' '.join(s[2])}" for s in SETS])
def test_switch_set_compiles(src, flags, switches):
    if not os.path.exists(BASE[0]):
        pytest.skip("hipcc not installed")
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run(BASE + flags + switches + ["-c", os.path.join(CSRC, src), "-o", os.path.join(tmp, "x.o")], capture_output=True, text=True, cwd=CSRC)
        assert r.returncode == 0, r.stderr[-3000:]
