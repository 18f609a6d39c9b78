#!/usr/bin/env python3
"""bench.py -- ORB extract(+match) frames/s on MI355X, one process per GPU.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N>1 it
is launched under torch.distributed.run with one rank per GPU.  A "step" is one pass of the hot
path (ORBextractor::operator() for every frame of the rank's resident batch).  Frames are
independent, so ranks share nothing: the only collectives are the barriers around the timed
region and the max/sum reductions of the report (RCCL) -- weak scaling.

Workloads (BASELINE.json configs):
  tum   (default) configs[1]: 640x480 mono stream, nFeatures=1000, extract only
  kitti           configs[2]: 1241x376 stereo, nFeatures=2000, extract L+R + ComputeStereoMatches
  euroc           configs[3]: 752x480, nFeatures=1200, extract + SearchByBoW(t-1, t)
Inputs are synthetic (no datasets offline), resident in HBM before the timed region starts.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    "tum": dict(name="TUM fr1_xyz mono 640x480 nFeatures=1000 extract-only (synthetic frames)",
                w=640, h=480, nfeatures=1000, ini=20, mn=7),
    "kitti": dict(name="KITTI 00 stereo 1241x376 nFeatures=2000 extract L+R + ComputeStereoMatches "
                       "(synthetic stereo pairs; frames/s counts stereo frames)",
                  w=1241, h=376, nfeatures=2000, ini=20, mn=7, stereo=True, bf=386.1448, fx=718.856),
    "euroc": dict(name="EuRoC MH_01 752x480 nFeatures=1200 extract + ComputeBoW + SearchByBoW(t-1,t) "
                       "(synthetic frames, synthetic k=10 L=2 vocabulary)",
                  w=752, h=480, nfeatures=1200, ini=20, mn=7, bow=True),
}


def level_pixels(ext, w, h):
    return [ext.level_size(w, h, l) for l in range(ext.GetLevels())]


def algorithmic_bytes(sizes, n_kp):
    """per-frame algorithmic bytes, split by the kernel that moves them."""
    P0 = sizes[0][0] * sizes[0][1]
    P = sum(a * b for a, b in sizes)
    last = sizes[-1][0] * sizes[-1][1]
    parts = {
        "pyramid": (P - P0) + (P - last),      # write levels 1..7 + read levels 0..6
        "fast": P,                             # read every level once for FAST
        "blur": 2 * P,                         # read + write every level
        "orient_desc": n_kp * (749 + 512 + 28 + 32),
    }
    parts["extract_total"] = P0 + sum(parts.values())  # + read of the input frame
    return parts


STAGE_KERNELS = {  # kernels (and launches per step) behind each timed stage
    "pyramid": [("k_resize_flat", 7)], "fast": [("k_fast_cells", 1)],
    "octree": [("k_gather_candidates", 1), ("k_octree", 1)], "blur": [("k_blur7", 1)],
    "orient_desc": [("k_orient_desc", 1)],
}


def pmc_traffic(stage, workload, batch, frames_per_launch):
    """HBM bytes per launch of the stage's kernels from the committed rocprofv3 --pmc summary
    (profiles/*_traffic.json, produced by tools/collect_traffic.py, collected at `batch` frames per
    launch and scaled to the frames one timed launch processes); None when no summary matches."""
    best = None
    for f in sorted((ROOT / "profiles").glob("*_traffic.json")):
        try:
            t = json.loads(f.read_text())
        except Exception:
            continue
        if t.get("workload") == workload:
            best = t
    if best is None:
        return None
    tot = 0.0
    for k, n in STAGE_KERNELS[stage]:
        if k not in best["kernels"]:
            return None
        tot += best["kernels"][k]["traffic_bytes_per_launch"] * n
    return tot * frames_per_launch / best["batch"]


VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4  # wave64 VALU instructions/s: 256 CUs x 4 SIMDs, 4 cycles per wave64 op


def pmc_valu(stage, workload):
    """VALU wave-instructions per frame of the stage's kernels from the committed SQ_INSTS_VALU summary
    (profiles/*_valu.json, tools/collect_valu.py); None when no summary matches."""
    best = None
    for f in sorted((ROOT / "profiles").glob("*_valu.json")):
        try:
            t = json.loads(f.read_text())
        except Exception:
            continue
        if t.get("workload") == workload:
            best = t
    if best is None:
        return None, None
    tot = 0.0
    for k, _ in STAGE_KERNELS[stage]:
        if k not in best["kernels"]:
            return None, None
        tot += best["kernels"][k]["valu_wave_instr_per_frame"]
    return tot, best.get("total_valu_wave_instr_per_frame")


def cpu_baseline(frames, wl, seconds=12.0):
    """The CPU oracle (a scalar C port of the reference path, oracle/orb_oracle.c) on 1 host core."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as orc
    o = orc.Oracle(wl["nfeatures"], 1.2, 8, wl["ini"], wl["mn"])
    o.extract(frames[0])  # warm
    t0 = time.perf_counter()
    n = 0
    per = []
    while True:
        t1 = time.perf_counter()
        o.extract(frames[n % len(frames)])
        per.append(time.perf_counter() - t1)
        n += 1
        if time.perf_counter() - t0 > seconds:
            break
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="frames/s", cores=1, kind="port",
                sample=f"{n} synthetic {wl['w']}x{wl['h']} frames of the same workload, oracle/orb_oracle.c "
                       f"(scalar C, gcc -O3), median {1e3 * float(np.median(per)):.2f} ms/frame",
                stages_s=o.stage_times())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="frames resident per GPU and processed per step")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="tum")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true", help="verify frame 0 of the batch against the oracle")
    ap.add_argument("--streams", type=int, default=4,
                    help="sub-batch HIP streams per call in the timed region (1..8): the L2-bound descriptor "
                         "kernel of one sub-batch overlaps the VALU-bound FAST/blur kernels of the others")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even for 1 rank")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = args.gpus > 1 or world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import orb_slam2_annotate_amd as amd
    from orb_slam2_annotate_amd import synth

    wl = WORKLOADS[args.workload]
    W, H, B = wl["w"], wl["h"], args.batch
    # each rank renders its own shard of the synthetic stream (sequence id = rank)
    stereo = bool(wl.get("stereo"))
    if stereo:  # B stereo frames = 2B images, ordered L0,R0,L1,R1,... through one extractor handle
        pairs = [synth.render_stereo(5000 + 1000 * rank + i, W, H) for i in range(B)]
        frames = [im for pr in pairs for im in pr]
    else:
        frames = synth.render_sequence(1000 + rank, B, W, H, step=1.5)
    NI = len(frames)  # images resident per GPU
    ext = amd.ORBextractor(wl["nfeatures"], 1.2, 8, wl["ini"], wl["mn"], device=local_rank)
    cap = ext.max_keypoints()
    d_img = torch.from_numpy(np.stack(frames)).to(dev)
    d_kp = torch.zeros((NI, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((NI, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((NI,), dtype=torch.int32, device=dev)
    if stereo:
        mbf = np.float32(wl["bf"])
        mb = np.float32(mbf / np.float32(wl["fx"]))  # src/Frame.cc:114
        d_u = torch.zeros((B, cap), dtype=torch.float32, device=dev)
        d_dep = torch.zeros((B, cap), dtype=torch.float32, device=dev)
        d_ns = torch.zeros((B,), dtype=torch.int32, device=dev)
    bow = bool(wl.get("bow"))
    if bow:
        import tempfile
        from orb_slam2_annotate_amd.vocabulary import write_synthetic_vocabulary
        vpath = os.path.join(tempfile.gettempdir(), f"orbfe_voc_{os.getpid()}.txt")
        write_synthetic_vocabulary(vpath, k=10, L=2, seed=1)
        voc = amd.ORBVocabulary(device=local_rank)
        assert voc.loadFromTextFile(vpath)
        os.unlink(vpath)
        d_match = torch.zeros((NI - 1, cap), dtype=torch.int32, device=dev)
        d_nm = torch.zeros((NI - 1,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def step(wait=False):
        ext.extract_batch_device(d_img.data_ptr(), NI, W, H, W, W * H, d_kp.data_ptr(), d_desc.data_ptr(), cap,
                                 d_n.data_ptr(), wait=wait and not stereo and not bow)
        if stereo:
            ext.stereo_match_batch_device(B, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, float(mbf),
                                          float(mb), d_u.data_ptr(), d_dep.data_ptr(), d_ns.data_ptr())
            if wait:
                ext.synchronize()
        if bow:
            ext.synchronize()  # the vocabulary handle runs on its own stream
            voc.bow_match_consecutive_batch_device(NI, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap,
                                                   d_match.data_ptr(), d_nm.data_ptr(), nnratio=0.7,
                                                   check_orientation=True, levelsup=0)

    def barrier():
        ext.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    # warmup on ONE stream, every stage timed with HIP events on the extractor's stream: finds the
    # dominant kernel and gives its exclusive (un-shared) duration
    gpu_stages = ["pyramid", "fast", "octree", "blur", "orient_desc"]
    ext.set_streams(1)
    ext.profile(True)
    for _ in range(max(args.warmup, 1)):
        step(wait=True)
    warm = ext.profile_get()
    n_warm = max(args.warmup, 1)
    dom = max(gpu_stages, key=lambda s_: warm[s_][0])
    # timed region: K steps enqueued back to back (each step = one pass of ORBextractor::operator()
    # over the resident batch, split over `streams` sub-batch streams); only the dominant stage keeps
    # its events: one pair per sub-batch launch, recorded on that sub-batch's own stream
    ext.set_streams(max(1, min(8, args.streams)))
    ext.profile(False)
    step(wait=True)  # one untimed pass on the new stream split
    ext.profile([dom])
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(wait=False)
    barrier()
    dt = time.perf_counter() - t0
    prof = ext.profile_get()
    ext.profile(False)

    n_kp = float(d_n.float().mean().item())
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    frames_done = torch.tensor([float(B * args.steps)], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(frames_done, op=dist.ReduceOp.SUM)
    dt_max = float(t.item())
    total_frames = float(frames_done.item())

    if args.check and rank == 0:
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_lib as orc
        o = orc.Oracle(wl["nfeatures"], 1.2, 8, wl["ini"], wl["mn"])
        for fi in sorted({0, NI // 2, NI - 1}):  # first, middle (second sub-batch) and last image
            kr, dr = o.extract(frames[fi])
            n0 = int(d_n[fi].item())
            kg = d_kp[fi, :n0].cpu().numpy().view(np.uint8).reshape(n0, 28)
            assert n0 == len(kr) and np.array_equal(kg, kr.view(np.uint8).reshape(-1, 28)), "keypoints differ from oracle"
            assert np.array_equal(d_desc[fi, :n0].cpu().numpy(), dr), "descriptors differ from oracle"
        if stereo:
            kL, dL, pL = o.extract(frames[0], want_pyramid=True)
            kR, dR, pR = o.extract(frames[1], want_pyramid=True)
            u_ref, dep_ref = o.stereo(W, H, kL, dL, kR, dR, pL, pR, float(mbf), float(mb))
            assert np.array_equal(d_u[0, :len(kL)].cpu().numpy(), u_ref), "mvuRight differs from oracle"
            assert np.array_equal(d_dep[0, :len(kL)].cpu().numpy(), dep_ref), "mvDepth differs from oracle"
        if bow:
            vpath = os.path.join(tempfile.gettempdir(), f"orbfe_voc_chk_{os.getpid()}.txt")
            write_synthetic_vocabulary(vpath, k=10, L=2, seed=1)
            vo = orc.Vocabulary(vpath)
            os.unlink(vpath)
            k0, de0 = o.extract(frames[0])
            k1, de1 = o.extract(frames[1])
            _, _, _, nd0 = vo.transform(de0, 0)
            _, _, _, nd1 = vo.transform(de1, 0)
            rn, rm = orc.search_by_bow(de0, np.ones(len(k0), np.uint8), k0["angle"], orc.FeatVec(nd0), de1,
                                       k1["angle"], orc.FeatVec(nd1), 0.7, True)
            assert int(d_nm[0].item()) == rn and np.array_equal(d_match[0, :len(k1)].cpu().numpy(), rm), \
                "SearchByBoW differs from oracle"

    if rank == 0:
        sizes = level_pixels(ext, W, H)
        alg = algorithmic_bytes(sizes, n_kp)
        alg["octree"] = 0
        # dominant kernel = the stage with the largest event time; timed live in the timed region
        # every sub-batch launch of the dominant stage is timed: prof[dom] = (ms, launches, frames) in total
        n_groups = max(prof[dom][1] / sum(n for _, n in STAGE_KERNELS[dom]), 1)  # timed launches of the stage
        dom_ms_per_group = prof[dom][0] / n_groups
        dom_frames_per_group = prof[dom][2] / n_groups
        ach = alg[dom] * prof[dom][2] / (prof[dom][0] * 1e-3) / 1e9 if prof[dom][0] > 0 else 0.0
        imgs_per_frame = NI / B
        value = total_frames / dt_max
        out = {
            "metric": "ORB extract frames/sec (bit-exact kp/desc vs CPU oracle)",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_max / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": wl["name"], "frames_per_gpu_per_step": B, "images_per_gpu_per_step": NI,
                       "keypoints_per_frame": n_kp,
                       "sharding": f"frames sharded one batch per GPU x{world}, no data-path collective"},
            "roofline": {
                "bound": "hbm", "kernel": "+".join(k for k, _ in STAGE_KERNELS[dom]), "stage": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                "traffic": pmc_traffic(dom, args.workload, B, dom_frames_per_group),
                "algorithmic_bytes_per_launch_group": alg[dom] * dom_frames_per_group,
                "algorithmic_bytes_per_frame": alg[dom], "ms_per_launch_group": dom_ms_per_group,
                "frames_per_launch": dom_frames_per_group, "launch_groups_timed": n_groups,
                "pipeline": {"algorithmic_bytes_per_image": alg["extract_total"],
                             "achieved": alg["extract_total"] * imgs_per_frame * value / world / 1e9,
                             "frac": alg["extract_total"] * imgs_per_frame * value / world / 1e9 / HBM_PEAK_GBS},
                "streams": max(1, min(8, args.streams)),
                "exclusive": {  # the same stage alone on the GPU (single-stream warm-up passes)
                    "ms_per_launch_group": warm[dom][0] / n_warm, "frames_per_launch": NI,
                    "achieved": alg[dom] * NI / (warm[dom][0] / n_warm * 1e-3) / 1e9,
                    "frac": alg[dom] * NI / (warm[dom][0] / n_warm * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "stage_ms_per_step_exclusive": {s_: warm[s_][0] / n_warm for s_ in gpu_stages},
            },
        }
        # what actually limits the dominant kernel: VALU issue, from the committed PMC summary
        v_stage, v_total = pmc_valu(dom, args.workload)
        if v_stage is not None:
            excl_s = warm[dom][0] / n_warm * 1e-3
            out["roofline"]["valu_issue"] = {
                "wave_instr_per_frame": v_stage, "peak": VALU_ISSUE_PEAK, "unit": "wave64 VALU instr/s",
                "achieved_exclusive": v_stage * NI / excl_s, "frac_exclusive": v_stage * NI / excl_s / VALU_ISSUE_PEAK,
                "note": "peak = nominal rate of full wave64 instructions (4 cycles each); an instruction whose upper or "
                        "lower half-wave is inactive issues in 2, so a divergent kernel can exceed it",
                "pipeline_wave_instr_per_frame": v_total,
                "pipeline_frac": (v_total * imgs_per_frame * value / world / VALU_ISSUE_PEAK) if v_total else None}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames, wl)
            if stereo:
                out["cpu_baseline"]["note"] = "extraction only (per image); stereo matching not included"
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
