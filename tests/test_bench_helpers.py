"""CPU: the data files bench.py prices its roofline with are consistent with their sources — the exact algorithmic bytes follow SURVEY.md §8(d)'s
formulas from the committed oracle statistics, the task share and the issue counters are well formed, and bench.py's helpers pick them up."""
import importlib.util
import json
import os

import pytest

from conftest import GOLDEN, ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod


def test_exact_bytes_follow_the_survey_formulas():
    tab = json.load(open(os.path.join(ROOT, "profiles", "algorithmic_bytes.json")))
    gold = json.load(open(os.path.join(GOLDEN, "reference_outputs.json")))["_fullsize"]["uniform10000000_s1"]
    n = gold["n"]
    st = gold["hploc"]["stats"]
    emit = 4 + 16 + 4 * (st["cluster_loads"] + st["cluster_stores"]) / n + 28 * st["cluster_loads"] / n + 32      # keys + parentIdx exchange + ids + AABB loads + node
    e = tab["uniform_10000000_tris_hploc"]
    assert e["emit_bytes_per_prim"] == pytest.approx(emit, abs=2e-3) and e["pipeline_bytes_per_prim"] == pytest.approx(88 + 32 + 68 + 64 + emit, abs=2e-3)
    assert 380 < e["pipeline_bytes_per_prim"] < 395                      # SURVEY.md's estimate: ~386
    sp = gold["ploc"]["stats"]
    emit_p = (32 * sp["cluster_loads"] + 4 * sp["cluster_stores"]) / n + 32
    assert tab["uniform_10000000_tris_ploc"]["emit_bytes_per_prim"] == pytest.approx(emit_p, abs=2e-3)


def test_bench_helpers_read_the_data_files():
    b = _bench()
    ex = b.exact_bytes("uniform_10000000_tris_hploc")
    assert ex is not None and ex[1] == 64.0
    share = json.load(open(os.path.join(ROOT, "profiles", "hploc_task_share.json")))
    assert share["merge_tasks_total"] == 918014 and 0.8 < share["tile_kernel_share"] < 0.95
    blk, src = b.kernel_bytes_per_prim("k_hploc_block", "hploc", "uniform_10000000_tris_hploc")
    ext, _ = b.kernel_bytes_per_prim("k_hploc_ext", "hploc", "uniform_10000000_tris_hploc")
    assert blk + ext == pytest.approx(64.0 + ex[0], abs=1e-6) and "share" in src          # the two kernels split SetupClusters + emit, nothing lost
    assert b.kernel_bytes_per_prim("k_extents", "hploc", "uniform_10000000_tris_hploc")[0] == 88.0
    assert b.kernel_bytes_per_prim("k_hploc_block", "hploc", "no_such_workload")[0] == 177.9   # falls back to the survey constants


def test_issue_counter_file_is_recomputable():
    ic = json.load(open(os.path.join(ROOT, "profiles", "issue_counters.json")))
    e = ic["k_hploc_block@10000000"]
    cyc = e["SQ_BUSY_CYCLES"] / e["shader_engines"]
    valu = e["SQ_INSTS_VALU"] * e["cycles_per_valu_inst"] / (e["simds"] * cyc)
    lds = e["SQ_INSTS_LDS"] * e["cycles_per_lds_inst"] / (e["cus"] * cyc)
    assert 0.3 < valu <= 1.0 and 0.3 < lds < 1.0 and 0.3 < e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"] < 0.8
