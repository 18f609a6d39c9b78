"""configs[4] on the hardware a builder has: TWO real ranks (not the stub) through bench.py's own launcher on the ONE GPU
of the box -- gloo for the barrier / MAX / SUM (RCCL refuses two ranks on one device), the ranks sharing device 0.  The
rate of such a run means nothing (the line says ranks_per_gpu = 2); what is checked is the N > 1 code path end to end:
every rank renders, extracts and matches its shard, the headline's in-run parity check against the CPU oracle holds,
and the KITTI 00-07 `sequence` / `round_robin` plans ride along with the shares shard.py computes."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

from orb_slam2_annotate_amd import shard

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_two_ranks_run_the_sharded_plans_on_one_gpu(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    scale = 0.02
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--batch", "16", "--steps", "2",
           "--warmup", "1", "--min-seconds", "0", "--max-repeats", "1", "--seq-scale", str(scale), "--no-cpu-baseline",
           "--render-procs", "1", "--detail-out", str(tmp_path / "detail.json")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    line = p.stdout.strip().splitlines()[-1]
    assert len(line) < 4096
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["ranks_per_gpu"] == 2
    assert out["parity_check"]["ok"] is True
    assert out["config"]["units_per_gpu_per_step"] == 16
    assert out["value"] > 0 and out["roofline"]["bound"] == "hbm"
    (seq,) = [s_ for s_ in out["secondary"] if s_["key"] == "kitti_seq"]
    lengths = [max(1, int(round(n * scale))) for n in shard.KITTI_00_07]
    for mode in ("sequence", "round_robin"):
        plan = shard.shard_sequences(lengths, 2, mode)
        got = seq["plans"][mode]
        assert got["stereo_frames_per_step"] == sum(lengths)
        assert got["frames_of_rank0"] == shard.frames_of(plan[0])
        assert got["max_frames_of_a_rank"] == max(shard.frames_of(q) for q in plan)
        assert got["value"] > 0
