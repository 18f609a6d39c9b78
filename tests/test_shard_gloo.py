"""Host logic of the multi-GPU path on CPU: sharding plans, and the barrier/reduction protocol of
bench.py with torch.distributed (gloo, world_size 2)."""
import os
import socket

import pytest

from orb_slam2_annotate_amd import shard


def test_sequence_sharding_covers_everything_once():
    for world in (1, 2, 4, 8):
        for mode in ("sequence", "round_robin"):
            plan = shard.shard_sequences(shard.KITTI_00_07, world, mode)
            assert len(plan) == world
            seen = {}
            for r in range(world):
                for s, b, e in plan[r]:
                    assert 0 <= b < e <= shard.KITTI_00_07[s]
                    for f in (b, e - 1):
                        assert (s, f) not in seen
                    seen.setdefault(s, []).append((b, e))
            for s, n in enumerate(shard.KITTI_00_07):
                iv = sorted(seen[s])
                assert iv[0][0] == 0 and iv[-1][1] == n
                assert all(iv[i][1] == iv[i + 1][0] for i in range(len(iv) - 1))
    # one-per-GPU is bounded by the longest sequence; round robin is balanced
    p8 = shard.shard_sequences(shard.KITTI_00_07, 8, "sequence")
    assert max(shard.frames_of(p) for p in p8) == 4661
    rr = shard.shard_sequences(shard.KITTI_00_07, 8, "round_robin")
    counts = [shard.frames_of(p) for p in rr]
    assert max(counts) - min(counts) <= 1 and sum(counts) == sum(shard.KITTI_00_07)
    assert shard.shard_sequences([], 3) == [[], [], []]
    with pytest.raises(ValueError):
        shard.shard_sequences([1], 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = shard.shard_sequences(shard.KITTI_00_07, world, "sequence")[rank]
    frames = shard.frames_of(plan)
    dist.barrier()
    elapsed = 1.0 + rank  # pretend the ranks took different times
    dist.barrier()
    t, n = shard.aggregate(elapsed, frames, dist)
    q.put((rank, t, n))
    dist.destroy_process_group()


def test_gloo_world2_barrier_and_reductions():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, t, n in res:
        assert t == 2.0  # max over ranks
        assert n == float(sum(shard.KITTI_00_07))  # every frame counted exactly once


def test_bench_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must spawn two ranks (fresh child
    processes), run the barrier / MAX / SUM protocol (gloo here, a stub step: no GPU) and relay rank 0's single
    JSON line with n_gpus = 2."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--stub", "--steps", "3"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3
    assert out["total_units"] == 3 * (100 + 101)  # SUM over both ranks
    assert out["ms_per_step"] >= 4.0  # MAX over ranks: rank 1 sleeps 4 ms per step
    # with more than one rank the line also carries the configs[4] plans (strong scaling) as a secondary
    (seq,) = [s_ for s_ in out["secondary"] if s_["key"] == "kitti_seq"]
    assert seq["scaling"] == "strong" and set(seq["plans"]) == {"sequence", "round_robin"}
    total = sum(shard.KITTI_00_07)
    for mode in ("sequence", "round_robin"):
        assert seq["plans"][mode]["stereo_frames_per_step"] == total  # SUM over ranks: every frame exactly once
    assert seq["plans"]["sequence"]["frames_of_rank0"] == sum(shard.KITTI_00_07[0::2])
    assert abs(seq["plans"]["round_robin"]["frames_of_rank0"] - total / 2) <= 1


def test_bench_launcher_reports_a_failing_rank():
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    # an unknown workload makes every rank exit with an argparse error: the launcher must not report success
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--stub", "--workload", "nope"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0


def test_bench_launcher_world8_keeps_the_compact_line():
    """configs[4] rehearsed without hardware: `--gpus 8 --stub` through the real launcher (8 gloo ranks): ONE compact
    line (< 4 KB, the driver parses the last stdout line) that still carries the kitti_seq plans; the `sequence` plan is
    bounded by the longest sequence (4661 frames of KITTI 02), `round_robin` is balanced."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "8", "--stub", "--steps", "2", "--no-detail"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.strip().splitlines()
    assert len(lines[-1]) < 4096
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 8 and out["total_units"] == 2 * sum(100 + r for r in range(8))
    (seq,) = [s_ for s_ in out["secondary"] if s_["key"] == "kitti_seq"]
    total = sum(shard.KITTI_00_07)
    assert seq["plans"]["sequence"]["max_frames_of_a_rank"] == 4661
    assert seq["plans"]["sequence"]["frames_of_rank0"] == shard.KITTI_00_07[0]
    assert seq["plans"]["round_robin"]["max_frames_of_a_rank"] - total // 8 <= 1
    for mode in ("sequence", "round_robin"):
        assert seq["plans"][mode]["stereo_frames_per_step"] == total
