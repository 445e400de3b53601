"""bench.py's host-side pieces that need no GPU: the fail-fast of a launch wider than the node, the leg selection, the
compact summary of the secondary legs, and the roofline rule (longest average launch, no byte-weighted tie-break)."""
import importlib.util
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_main", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_launch_wider_than_the_node_is_refused_before_any_rank_starts():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "64"], capture_output=True, text=True, timeout=300,
                       cwd=str(ROOT))
    assert r.returncode == 3 and r.stdout == "" and "needs 64 visible GPUs" in r.stderr


def test_leg_selection_follows_the_size_flags():
    b = _bench()
    sys.argv = ["bench.py"]
    a = b.parse_args()
    assert b.legs_to_run(a) == "estimated,latency,streaming,c2k,c3,c5"
    sys.argv = ["bench.py", "--latency-frames", "0", "--config-frames", "0", "--no-estimated"]
    assert b.legs_to_run(b.parse_args()) == "streaming"
    sys.argv = ["bench.py", "--legs", "none"]
    assert b.legs_to_run(b.parse_args()) == ""
    sys.argv = ["bench.py", "--legs", "c5b256t,latency", "--latency-frames", "0"]
    assert b.legs_to_run(b.parse_args()) == "c5b256t"


def test_brief_of_the_legs_is_small_and_tolerates_missing_legs():
    b = _bench()
    assert b.brief(None) is None and b.brief({"configs": {}}) == {}
    detail = {"plane_estimated": {"ms_per_step": 0.9, "x": "y" * 5000},
              "latency": {"ms_per_frame_median": 0.13, "estimated": {"ransac": {"ms_per_frame_median": 0.15}}},
              "configs": {"5": {"ms_per_frame": 0.14, "batched": {"256": {"associations_per_s": 2e9, "ms_per_step": 1.4,
                                                                           "roofline": {"kernel": "k_feature_fused", "frac": 0.05}}}}}}
    br = b.brief(detail)
    assert br["plane_estimated_ms_per_step"] == 0.9 and br["frame_call_ms"]["ransac"] == 0.15
    assert br["frame_call_ms"]["semantic"] is None and br["config5"]["batched_value"] == {"256": 2e9}
    assert br["config5"]["kernel"] == "k_feature_fused" and len(json.dumps(br)) < 1024


def test_roofline_rule_is_the_longest_average_launch():
    from bench_support.rooflines import HBM_PEAK_GBS, headline_roofline, pmc_traffic
    S = B = 1024
    for fused_ms, want in ((0.66, "k_feature_fused"), (0.60, "k_project_scatter")):
        kt = {"k_project_scatter": {"avg_ms": 0.65, "launches": 10}, "k_classify": {"avg_ms": 0.05, "launches": 10},
              "k_feature_fused": {"avg_ms": fused_ms, "launches": 10}, "k_feature_wave": {"avg_ms": 0.03, "launches": 10}}
        compact, detail = headline_roofline(kt, {}, S, B, 131072, 10, 10 * 0.72e-3, 2, 2.2e9, 5e9, 1.7e9)
        assert compact["kernel"] == want and compact["second"]["kernel"] != want
        assert compact["kernel_ms"] >= compact["second"]["kernel_ms"]
        nbytes = pmc_traffic(want, S)[0]   # the committed counter profile, scaled to this launch size
        assert compact["traffic"] == nbytes
        assert abs(compact["frac"] - nbytes / (compact["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) < 1e-12
        assert compact["frac"] <= 1.0 and "formula_frac" not in compact and len(json.dumps(compact)) < 2048
        assert set(detail["kernels"]) == set(kt) and detail["kernels"]["k_project_scatter"]["formula_frac"] > 0
