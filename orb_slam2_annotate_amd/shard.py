"""Multi-GPU sharding of the ORB front-end: frames and whole sequences are independent
(SURVEY.md 8(e)), so ranks exchange NO data -- one process per GPU, an RCCL barrier around the
timed region and two scalar reductions for the report.  Pure host logic (torch.distributed
handles are passed in), tested on CPU with the gloo backend."""
from __future__ import annotations

from typing import List, Sequence, Tuple

# KITTI odometry 00-07 sequence lengths (frames); not in the reference repo, parameters only.
KITTI_00_07 = (4541, 1101, 4661, 801, 271, 2761, 1101, 1101)


def shard_sequences(lengths: Sequence[int], world: int, mode: str = "sequence") -> List[List[Tuple[int, int, int]]]:
    """Returns, per rank, a list of (sequence id, first frame, end frame) work items.

    mode "sequence":    sequence s -> rank s % world (BASELINE.json configs[4]); bounded by the
                        longest sequence.
    mode "round_robin": contiguous frame blocks dealt so that every rank gets the same number of
                        frames +-1 (the balanced variant SURVEY.md 8(e) asks to report as well).
    """
    if world < 1:
        raise ValueError("world must be >= 1")
    plan: List[List[Tuple[int, int, int]]] = [[] for _ in range(world)]
    if mode == "sequence":
        for s, n in enumerate(lengths):
            if n > 0:
                plan[s % world].append((s, 0, int(n)))
        return plan
    if mode == "round_robin":
        total = int(sum(lengths))
        base, extra = divmod(total, world)
        quota = [base + (1 if r < extra else 0) for r in range(world)]
        r = 0
        for s, n in enumerate(lengths):
            f = 0
            while f < n:
                while r < world and quota[r] == 0:
                    r += 1
                take = min(quota[r], n - f)
                plan[r].append((s, f, f + take))
                quota[r] -= take
                f += take
        return plan
    raise ValueError(f"unknown mode {mode!r}")


def frames_of(plan_for_rank) -> int:
    return sum(e - b for _, b, e in plan_for_rank)


def aggregate(elapsed_s: float, frames: float, dist=None, device=None):
    """(max elapsed over ranks, total frames over ranks).  `dist` = torch.distributed or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), float(frames)
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    n = torch.tensor([frames], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), float(n.item())
