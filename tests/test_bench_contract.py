"""bench.py prints ONE JSON line, last on stdout, with the fields the driver reads, and its
frame 0 is the cfg3 fixture case (so the line carries its own parity figure)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                          "--time-batch", "8", "--streams", "2", "--no-cpu-baseline", "--no-uint8", "--no-secondary"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1
    assert line["vs_baseline"] is None and line["dtype"] == "f32" and line["higher_is_better"] is True
    assert line["config"]["frames_per_step"] == 16 and line["config"]["streams"] == 2
    r = line["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0 < r["frac"] <= 1.0, "roofline.frac is a utilisation of the executed work, never above 1"
    assert line["config"]["valid_frames_per_step"] == line["config"]["frames_per_step"]
    assert len(line["kernels"]) >= 5 and all(0 < k["frac"] <= 1.0 for k in line["kernels"])
    assert {"median", "p10", "p90"} <= set(line["step_time_ms"])
    assert line["value"] > 0 and abs(line["value"] - 16 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"] + 1e-3
    # an empty frame set inside a batch leaves its neighbours untouched
    assert line["invalid_frame_check"]["other_frames_bit_equal"] is True
    # the separately labelled reduced-precision line: its own dtype, parity against the same fixture
    rp = line["reduced_precision"]
    assert rp["mode"] == "bf16x3" and rp["value"] > 0 and rp["dtype"].startswith("bf16x3")
    assert rp["parity_max_abs_mm_vs_reference_fixture"] < 1e-3 and 0 < rp["kernels"][0]["frac"] <= 1.0
    # frame 0 of the bench workload is the reference's own output for that input; with --time-batch 8 the timed
    # path is the time_batch >= 8 class (row-streaming BiFPN nodes), i.e. the form the default bench runs
    assert line["parity_max_abs_mm_vs_reference_fixture"] < 1e-3
    assert line["config"]["time_batch"] >= 8
    # every convolution row carries both floors, and `frac` is the binding (higher) one
    convs = [k for k in line["kernels"] if k["kernel"].startswith(("conv2d_", "conv3d_"))]
    assert convs and all(abs(k["frac"] - max(k["frac_mfma"], k["frac_hbm"])) < 1e-4 for k in convs)
    assert all(k["bound"] == ("hbm" if k["frac_hbm"] > k["frac_mfma"] else "mfma") for k in convs)


def test_bench_line_with_cpu_baseline_is_strict_on_any_host():
    """With the CPU baseline leg on: the host-oracle comparison is made with the HIP gather indices substituted
    (oracle.host_parity) and held to the north-star bar; every index flip of this host is a truncation tie."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                          "--time-batch", "8", "--streams", "1", "--cpu-seconds", "1", "--no-uint8", "--no-secondary",
                          "--no-reduced-precision", "--profile-passes", "1"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port"
    assert line["parity_max_abs_mm_vs_host_oracle_same_indices"] < 1e-3
    flips = line["host_oracle_index_flips"]
    assert flips["of"] == 12 * 64 ** 3 and flips["flips"] <= 64
    assert flips["max_dist_to_truncation_boundary"] <= 2.5e-4
