"""bench.py's contract line: built from a full result dict, it must stay parseable by the driver (round 4's 26 KB line was not).
The canned input is round 4's own full output (profiles/r4j_bench_steps20.json) — the dict that broke the parser."""
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import bench  # noqa: E402  (imports numpy only at module level; torch and the library are imported inside main())


def _canned():
    with open(os.path.join(REPO, "profiles", "r4j_bench_steps20.json")) as f:
        return json.load(f)


def test_line_is_short_and_round_trips():
    full = _canned()
    assert len(json.dumps(full)) > 20000  # the input really is the oversized one
    full["full_results_file"] = "bench_full.json"
    text = bench.compact_line(full)
    assert "\n" not in text
    assert len(text) < bench.LINE_LIMIT == 4096
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["value"] == pytest.approx(full["value"], rel=1e-5)
    assert line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert line["config"]["workload"].startswith("independent batch")
    rf = line["roofline"]
    for key in ("kernel", "kernel_ms", "bound", "achieved", "peak", "unit", "frac", "traffic", "fractions", "wait_fraction"):
        assert key in rf, key
    assert {"hbm", "l2", "issue"} <= set(rf["fractions"]) <= {"hbm", "l2", "issue", "valu"}
    assert rf["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-4)
    assert rf["fifo_chain"]["filter_p50_ms"] == pytest.approx(full["roofline"]["fifo_chain"]["filter_p50_ms"], rel=1e-5)
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["sample"]
    assert cb["fifo_chain"]["literal_p50_ms"] > 0


def test_line_without_optional_legs():
    """N > 1 ranks and --no-extras runs have no cpu_baseline / fifo_chain: the line is still complete and parseable."""
    full = _canned()
    for k in ("cpu_baseline", "extras", "fifo_filter", "end_to_end", "node_sharded", "cpu_baseline_variants"):
        full.pop(k, None)
    full["roofline"].pop("fifo_chain", None)
    full["n_gpus"] = 8
    line = json.loads(bench.compact_line(full))
    assert line["n_gpus"] == 8 and "cpu_baseline" not in line and "fifo_chain" not in line["roofline"]
    assert line["roofline"]["frac"] is not None


def test_emit_writes_the_full_file_and_prints_one_line(tmp_path, capsys, monkeypatch):
    full = _canned()
    path = tmp_path / "bench_full.json"
    monkeypatch.setenv("GANGFIT_BENCH_FULL", str(path))
    bench.emit(full)
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and len(out[0]) < bench.LINE_LIMIT
    assert json.loads(out[0])["full"] == str(path)
    assert json.load(open(path))["extras"]  # every leg of the run is in the file


def test_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` without a rank environment starts torch.distributed.run with N ranks on 127.0.0.1."""
    import subprocess

    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20"])
    assert bench.self_launch(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "20"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
