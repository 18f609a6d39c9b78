"""The ONE stdout line of bench.py must stay parseable by the driver (round 3 printed 31 KB and the driver's 8 KB tail
held no complete object): compact_line() of a canned full result -- round 3's own committed run -- is < 4 KB,
round-trips through json and keeps every key the contract names."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import bench  # noqa: E402

CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"]


def _canned():
    return json.loads((ROOT / "profiles" / "r03_bench.json").read_text())


def test_compact_line_is_small_and_complete():
    full = _canned()
    assert len(json.dumps(full)) > 20000  # the canned run really is the oversized one
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    assert "\n" not in line and len(line) < bench.LINE_LIMIT == 4096
    out = json.loads(line)
    for k in CONTRACT:
        assert k in out, k
    assert out["value"] == float(f"{full['value']:.6g}") and out["unit"] == "stereo frames/s"
    assert "workload" in out["config"] and "model" not in out["config"]
    rf = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "stage", "pipeline", "stages_exclusive_ms"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert rf["valu_issue"]["pipeline_frac"] > 0
    cb = out["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert "variants" not in cb and cb["kind"] in ("port", "reference")
    assert out["parity_check"]["ok"] is True
    assert [s_["key"] for s_ in out["secondary"]] == ["tum", "euroc", "euroc_stereo"]
    for s_ in out["secondary"]:
        assert set(s_) <= {"key", "value", "unit", "ms_per_step", "vs_cpu_baseline", "pipeline_frac", "parity_ok", "scaling", "plans"}
    assert out["detail"] == "gpurun_out/bench_detail.json"
    assert "latency" not in out and "e2e" not in out and "matching_work" not in out


def test_compact_line_sheds_before_it_overflows():
    full = _canned()
    full["secondary"] = full["secondary"] * 12  # a pathological run: the line must still fit
    line = bench.compact_line(full, None)
    assert len(line) < bench.LINE_LIMIT
    out = json.loads(line)
    for k in CONTRACT:
        assert k in out, k


def test_emit_writes_the_detail_file_and_prints_the_line_last(tmp_path, capsys):
    import argparse
    full = _canned()
    args = argparse.Namespace(no_detail=False, detail_out=str(tmp_path / "d.json"), full_line=False)
    bench.emit(full, args)
    last = capsys.readouterr().out.strip().splitlines()[-1]
    assert len(last) < 4096 and json.loads(last)["detail"].endswith("d.json")
    kept = json.loads((tmp_path / "d.json").read_text())
    assert kept["latency"]["rows"] and kept["secondary"][0]["roofline"]["stages"]
