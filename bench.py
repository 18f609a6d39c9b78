#!/usr/bin/env python3
"""bench.py -- ORB extract+match frames/s on MI355X, one process per GPU.

Contract (task statement): `python bench.py --gpus N --steps K --warmup W`.  For N > 1 either the
driver starts the ranks (torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE in the environment) or
this script does it itself: with `--gpus N` and no WORLD_SIZE in the environment the parent spawns N
fresh `python bench.py` children (one per GPU, RCCL rendezvous on 127.0.0.1) BEFORE anything touches
the GPU, relays rank 0's JSON line and exits non-zero if any rank failed.

torch.distributed is initialised at the FIRST collective (LazyDist), after the extractor's HIP streams exist: a RCCL
communicator created before them shifts their hardware-queue binding and costs every rank 11-15 %.

A "step" is one pass of the hot path over the rank's resident batch.  Frames / stereo pairs are
independent, so ranks share nothing: the only collectives are the barriers around the timed region
and the MAX(elapsed) / SUM(frames) reductions of the report (RCCL) -- weak scaling.

Workloads (BASELINE.json configs), `--workload all` (default) runs the three of them:
  kitti  (HEADLINE `value`) configs[2]: 1241x376 stereo, nFeatures=2000, ORBextractor::operator() on L and R
         (src/ORBextractor.cc:1119-1197) + Frame::ComputeStereoMatches (src/Frame.cc:512-686)
  tum    (secondary)        configs[1]: 640x480 mono stream, nFeatures=1000, extract only
  euroc  (secondary)        configs[3]: 752x480, nFeatures=1200, extract + ComputeBoW + SearchByBoW(t-1, t)
         (src/ORBmatcher.cc:185-325) with a synthetic vocabulary of ORBvoc's shape (k=10, L=6: 1 111 111 nodes) descended
         with levelsup=4 as src/Frame.cc:438 does (ORBvoc.txt itself is not in the reference); --voc-shape 10,2,0 = rounds 1-2
  euroc_stereo (secondary)  configs[3] as the reference runs a EuRoC frame (Examples/Stereo/stereo_euroc.cc:136-137 -> src/Frame.cc:61-117
         -> src/Tracking.cc:836-843): cv::remap of both raw images, extract L+R, ComputeStereoMatches, ComputeBoW, SearchByBoW(t-1, t)
  kitti_seq                 configs[4]: the KITTI 00-07 sequence lengths sharded over the ranks by shard.py's
         `sequence` (one sequence per GPU) and `round_robin` (balanced) plans -- strong scaling, both reported
Inputs are synthetic (no datasets offline), rendered before the GPU is touched and resident in HBM
before the timed region starts.  Every workload is checked against the CPU oracle on >= 3 frames
outside the timed region (keypoints, descriptors and match outputs bit-identical) on every run.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E (MI355X_MICROARCH.md)
N_SIMD = 256 * 4
# VALU issue ceiling, wave64 instructions/s chip-wide.  Guide: "v_fma_f32 (wave64): 2 cyc (SIMD-32)" at 2.4 GHz.
# tools/ubench/valu_rate.hip (profiles/r02_valu_rate.txt) measures two classes on this part:
#   v_add_u32 / v_xor_b32 / v_fma_f32           0.93-1.02 T/s with >= 2 waves per SIMD (2.4-2.7 cycles)
#   v_perm_b32 / v_min3_i32 / v_bcnt / v_pk_*   0.58 T/s whatever the occupancy (4.2 cycles); a half EXEC mask
#                                               changes neither class
# No instruction issues faster than the first figure, so a fraction of VALU_PEAK is <= 1 by construction.
VALU_PEAK = N_SIMD * 2.4e9 / 2
VALU_MEASURED = {"fast_class(v_add_u32,v_xor_b32,v_fma_f32)": 1.017e12,
                 "full_rate_class(v_perm_b32,v_min3_i32,v_bcnt_u32_b32,v_pk_min_i16)": 0.585e12}
BCNT_PEAK = 0.585e12 * 64  # popcount-32 lane-ops/s: v_bcnt_u32_b32 measured at 0.585 T wave-instr/s x 64 lanes

WORKLOADS = {
    "tum": dict(name="TUM fr1_xyz mono 640x480 nFeatures=1000 extract-only (synthetic frames)",
                w=640, h=480, nfeatures=1000, ini=20, mn=7, batch=4096, streams=8),
    "kitti": dict(name="KITTI 00 stereo 1241x376 nFeatures=2000: extract L+R + ComputeStereoMatches "
                       "(synthetic stereo pairs; frames/s counts STEREO frames = 2 images each)",
                  w=1241, h=376, nfeatures=2000, ini=20, mn=7, stereo=True, bf=386.1448, fx=718.856, batch=512, streams=8),
    "euroc": dict(name="EuRoC MH_01 752x480 nFeatures=1200: extract + ComputeBoW + SearchByBoW(t-1,t) "
                       "(synthetic frames, synthetic vocabulary of ORBvoc's shape k=10 L=6, levelsup=4)",
                  w=752, h=480, nfeatures=1200, ini=20, mn=7, bow=True, batch=2048, streams=8),
    "euroc_stereo": dict(name="EuRoC MH_01 FULL stereo frame 752x480 nFeatures=1200: cv::remap L+R + extract L+R + ComputeStereoMatches + "
                              "ComputeBoW + SearchByBoW(t-1,t) (synthetic RAW pairs, vocabulary of ORBvoc's shape k=10 L=6, levelsup=4; "
                              "frames/s counts STEREO frames = 2 images each)",
                         w=752, h=480, nfeatures=1200, ini=20, mn=7, stereo=True, bow=True, raw=True, bf=47.90639384423901,
                         fx=435.2046959714599, batch=512, streams=8),
}
STEREO_SCENE = "textured"  # --stereo-scene: "textured" (one scene in both eyes, piecewise-planar sub-pixel disparity: >= 50 % of the
#   left keypoints obtain a stereo match, as on a rectified KITTI pair) | "shapes" (rounds 1-2: per-object integer shifts, 14 % match)
SINGLE_SCENE = False  # --single-scene: the round-1 synthetic input (sparser; for continuity with profiles/history/r01_bench.json)
VOC_SHAPE = (10, 6, 4)  # (k, L, levelsup) of the euroc workload's vocabulary: ORBvoc.txt's shape and src/Frame.cc:438's levelsup
GPU_STAGES = ["h2d", "pyramid", "fast", "octree", "blur", "orient_desc", "match"]  # "h2d" = the ingest stage: k_remap x2 (euroc_stereo)
STAGE_KERNELS = {  # kernels (and launches per step) behind each timed stage
    "h2d": [("k_remap", 2)], "pyramid": [("k_copy2d", 1), ("k_resize_flat", 7)], "fast": [("k_fast_cells", 1)],
    "octree": [("k_gather_candidates", 1), ("k_octree", 1)], "blur": [("k_blur7", 1)],
    "orient_desc": [("k_orient_desc", 1)],
    "match": [("k_stereo_bucket", 1), ("k_stereo_match_batch", 1), ("k_stereo_median_cut", 1),
              ("k_vocab_transform16", 1), ("k_vocab_featvec", 1), ("k_search_by_bow_batch", 1)],
}


# ------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` starts the N ranks itself (the parent never touches the GPU)
# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n: int) -> int:
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = b""
    rc = 0
    deadline = None
    try:
        out0 = procs[0].communicate()[0] or b""
        deadline = time.time() + 120
        for p in procs:
            p.wait(timeout=max(1.0, deadline - time.time()))
    except subprocess.TimeoutExpired:
        rc = 1
    for p in procs:
        if p.poll() is None:  # a rank that is still alive after rank 0 ended and the grace period: stop exactly it
            p.kill()
            rc = 1
        elif p.returncode != 0:
            rc = rc or p.returncode or 1
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return rc


# ------------------------------------------------------------------------------------------------
# synthetic inputs (rendered in worker processes forked before any GPU use)
# ------------------------------------------------------------------------------------------------
def _render_task(t):
    from orb_slam2_annotate_amd import synth
    kind, seed, n, w, h = t
    if kind == "stereo_raw":
        l, r = synth.render_stereo_raw(seed, w, h)
        return [l, r]
    if kind.startswith("stereo"):
        l, r = synth.STEREO_SCENES[kind.partition(":")[2] or "shapes"](seed, w, h)
        return [l, r]
    return synth.render_sequence(seed, n, w, h, step=1.5)


def render_inputs(names, batches, rank, procs=0, cache=None):
    """{workload: list of u8 frames}.  Mono streams: chunks of 64 consecutive frames of one scene (sequence id =
    seed); stereo: one scene per pair, ordered L0,R0,L1,R1,...  `cache`: directory holding the rendered batches
    as .npy (written on first use) -- the profiler passes of tools/profile_round.sh reuse what the plain run rendered."""
    out, todo = {}, []

    def cache_file(nm):
        tag = f"_{STEREO_SCENE}" if WORKLOADS[nm].get("stereo") and not WORKLOADS[nm].get("raw") else ""
        return Path(cache) / f"{nm}{tag}_{batches[nm]}_{rank}.npy"

    for nm in names:
        f = cache_file(nm) if cache else None
        if f is not None and f.exists():
            out[nm] = list(np.load(f))
        else:
            todo.append(nm)
    if todo:
        rendered = _render(todo, batches, rank, procs)
        for nm in todo:
            out[nm] = rendered[nm]
            if cache:
                Path(cache).mkdir(parents=True, exist_ok=True)
                np.save(cache_file(nm), np.stack(rendered[nm]))
    return out


def _render(names, batches, rank, procs=0):
    tasks, owner = [], []
    for nm in names:
        wl, B = WORKLOADS[nm], batches[nm]
        if SINGLE_SCENE and not wl.get("stereo"):  # round-1 style input: ONE scene for the whole mono batch
            tasks.append(("seq", 1000 + rank, B, wl["w"], wl["h"]))
            owner.append(nm)
        elif wl.get("raw"):
            for i in range(B):
                tasks.append(("stereo_raw", 9000 + 100000 * rank + i, 1, wl["w"], wl["h"]))
                owner.append(nm)
        elif wl.get("stereo"):
            for i in range(B):
                tasks.append(("stereo:" + STEREO_SCENE, 5000 + 100000 * rank + i, 1, wl["w"], wl["h"]))
                owner.append(nm)
        else:
            for c in range(0, B, 64):
                tasks.append(("seq", 1000 + 100000 * rank + c // 64 + (7000 if nm == "euroc" else 0), min(64, B - c),
                              wl["w"], wl["h"]))
                owner.append(nm)
    # every rank renders its own inputs at the same time: share the host's cores between the ranks of this node
    world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    nproc = max(1, min(16, host_cores() // world, len(tasks)))
    # a profiler that preloads itself has initialised the GPU runtime before main(): never fork from such a process
    # (rocprofv3 --pmc does; plain --kernel-trace --stats does not, and tools/profile_round.sh asks for the pool there)
    if procs > 0:
        nproc = min(procs, max(1, len(tasks)))
    elif "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith("ROCPROF") for k in os.environ):
        nproc = 1
    if nproc > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(nproc) as pool:
            res = pool.map(_render_task, tasks, chunksize=1)
    else:
        res = [_render_task(t) for t in tasks]
    out = {nm: [] for nm in names}
    for nm, fr in zip(owner, res):
        out[nm].extend(fr)
    return out


# ------------------------------------------------------------------------------------------------
# algorithmic bytes and committed PMC summaries
# ------------------------------------------------------------------------------------------------
def algorithmic_bytes(sizes, n_kp, wl, n_stereo=0.0):
    """Per IMAGE, split by the stage that moves them; 'match' per FRAME (stereo pair / consecutive pair)."""
    P0 = sizes[0][0] * sizes[0][1]
    P = sum(a * b for a, b in sizes)
    last = sizes[-1][0] * sizes[-1][1]
    parts = {
        "pyramid": (P - P0) + (P - last),      # write levels 1..7 + read levels 0..6
        "fast": P,                             # read every level once for FAST
        "octree": 0,                           # candidates only (<< P): latency-bound, no algorithmic bytes
        "blur": 2 * P,                         # read + write every level
        "orient_desc": n_kp * (749 + 512 + 28 + 32),
    }
    parts["extract_total"] = P0 + sum(parts.values())  # + read of the input frame
    # cv::remap with two CV_32F maps: raw pixel + 8 map bytes in, rectified pixel out (the product's maps are 6 B/px)
    parts["h2d"] = 10 * P0 if wl.get("raw") else 0
    parts["match"] = 0
    if wl.get("stereo"):  # 32(N+Nr) + 28(N+Nr) + 8N + matched*(121 + 11*121)
        parts["match"] += 60 * 2 * n_kp + 8 * n_kp + n_stereo * (121 + 11 * 121)
    if wl.get("bow"):   # 32(na+nb) + 8 na (+ 4 bytes of node id per feature written and read back); the descent's node
        parts["match"] += 64 * n_kp + 8 * n_kp  # records are not counted: L2 / MALL traffic of a shared read-only tree
    return parts


def _latest_profile(suffix, workload):
    best = None
    for f in sorted((ROOT / "profiles").glob(f"*_{suffix}.json")):
        try:
            t = json.loads(f.read_text())
        except Exception:
            continue
        if t.get("workload") == workload:
            best = t
            best["_file"] = f.name
    return best


def pmc_traffic(stage, workload, images_per_launch, stage_kernels=None):
    """HBM bytes per launch group of the stage's kernels from the committed rocprofv3 --pmc summary
    (profiles/*_traffic.json, tools/collect_traffic.py), scaled to the images one timed launch processes."""
    t = _latest_profile("traffic", workload)
    if t is None:
        return None
    tot, found = 0.0, False
    for k, n in (stage_kernels or STAGE_KERNELS)[stage]:
        if k in t["kernels"]:
            tot += t["kernels"][k]["traffic_bytes_per_launch"] * n
            found = True
    return tot * images_per_launch / t["batch"] if found else None


def pmc_valu(workload):
    return _latest_profile("valu", workload)


# ------------------------------------------------------------------------------------------------
# CPU baseline: the oracle on the host cores (1 core; 2 threads L/R; all cores frame-parallel)
# ------------------------------------------------------------------------------------------------
class _CpuUnit:
    """One unit of the workload's CPU path = one frame (tum, euroc) or one stereo frame (kitti)."""

    def __init__(self, wlname, frames, voc_arrays=None):
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_lib as orc
        self.orc, self.wl, self.name, self.frames = orc, WORKLOADS[wlname], wlname, frames
        self.voc = orc.Vocabulary.from_arrays(voc_arrays) if voc_arrays is not None else None
        self.units = len(frames) // 2 if self.wl.get("stereo") else len(frames)
        wl = self.wl
        self.mbf = float(np.float32(wl.get("bf", 0)))
        self.mb = float(np.float32(np.float32(wl.get("bf", 0)) / np.float32(wl.get("fx", 1))))
        self.maps = rectify_maps_of(wl) if wl.get("raw") else None

    def new_oracle(self):
        wl = self.wl
        return self.orc.Oracle(wl["nfeatures"], 1.2, 8, wl["ini"], wl["mn"])

    def run(self, o, i, state, pool=None):
        wl, fr, orc = self.wl, self.frames, self.orc
        i %= self.units
        if wl.get("stereo"):
            imL, imR = fr[2 * i], fr[2 * i + 1]
            if self.maps:  # cv::remap of both images in the caller's thread (Examples/Stereo/stereo_euroc.cc:136-137)
                imL, imR = orc.remap_linear(imL, *self.maps[0]), orc.remap_linear(imR, *self.maps[1])
            if pool is not None:  # left and right image in two threads, src/Frame.cc:78-81
                fl = pool.submit(o[0].extract, imL, None, True)
                fr_ = pool.submit(o[1].extract, imR, None, True)
                (kL, dL, pL), (kR, dR, pR) = fl.result(), fr_.result()
                o0 = o[0]
            else:
                kL, dL, pL = o.extract(imL, want_pyramid=True)
                kR, dR, pR = o.extract(imR, want_pyramid=True)
                o0 = o
            o0.stereo(wl["w"], wl["h"], kL, dL, kR, dR, pL, pR, self.mbf, self.mb)
            if wl.get("bow"):  # Frame::ComputeBoW + SearchByBoW(previous frame as key frame, this frame), left keypoints
                fv = orc.FeatVec(self.voc.transform(dL, VOC_SHAPE[2])[3])
                prev = state.get("prev")
                if prev is not None and prev[0] == i - 1:
                    _, k0, d0, fv0 = prev
                    orc.search_by_bow(d0, np.ones(len(k0), np.uint8), k0["angle"], fv0, dL, kL["angle"], fv, 0.7, True)
                state["prev"] = (i, kL, dL, fv)
        elif wl.get("bow"):
            k, d = o.extract(fr[i])
            fv = orc.FeatVec(self.voc.transform(d, VOC_SHAPE[2])[3])
            prev = state.get("prev")
            if prev is not None and prev[0] == i - 1:
                _, k0, d0, fv0 = prev
                orc.search_by_bow(d0, np.ones(len(k0), np.uint8), k0["angle"], fv0, d, k["angle"], fv, 0.7, True)
            state["prev"] = (i, k, d, fv)
        else:
            o.extract(fr[i])


def rectify_maps_of(wl):
    """the two CV_32F map pairs cv::initUndistortRectifyMap hands stereo_euroc.cc (:97-98): an INPUT of the path, synthetic here"""
    from orb_slam2_annotate_amd import synth
    return (synth.rectify_maps(wl["w"], wl["h"], **synth.EUROC_CAM0), synth.rectify_maps(wl["w"], wl["h"], **synth.EUROC_CAM1))


def host_cores():
    """Cores this process may use: the affinity mask, capped by the cgroup CPU quota (a 1-GPU box share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return n


def _timed_loop(fn, seconds, min_units=3):
    per = []
    t0 = time.perf_counter()
    while True:
        t1 = time.perf_counter()
        fn(len(per))
        per.append(time.perf_counter() - t1)
        if time.perf_counter() - t0 > seconds and len(per) >= min_units:
            break
    dt = time.perf_counter() - t0
    return dict(value=len(per) / dt, units=len(per), mean_ms=1e3 * float(np.mean(per)), median_ms=1e3 * float(np.median(per)))


def cpu_baseline(wlname, frames, voc_arrays, seconds):
    """The CPU oracle (oracle/orb_oracle.c, scalar C, gcc -O3) timed like the reference examples time Track
    (steady wall clock around each unit, mean and median as Examples/Stereo/stereo_kitti.cc:113-122), in the three
    threadings BASELINE.md 2 names.  `value` is the reference's own threading for the workload."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    u = _CpuUnit(wlname, frames, voc_arrays)
    wl = u.wl
    unit = "stereo frames/s" if wl.get("stereo") else "frames/s"
    variants = {}
    o = u.new_oracle()
    st = {}
    u.run(o, 0, st)  # warm
    one = _timed_loop(lambda i: u.run(o, i, st), seconds)
    one["cores"] = 1
    one["stages_s"] = o.stage_times()
    variants["one_core"] = one
    if wl.get("stereo"):
        o2 = (u.new_oracle(), u.new_oracle())
        st2 = {}
        with ThreadPoolExecutor(2) as pool:
            u.run(o2, 0, st2, pool)
            two = _timed_loop(lambda i: u.run(o2, i, st2, pool), seconds * 0.6)
        two["cores"] = 2
        variants["two_threads_left_right"] = two
    ncpu = host_cores()
    T = max(1, min(ncpu, u.units, 64))
    counts, pers = [0] * T, [[] for _ in range(T)]
    stop = threading.Event()
    span = max(1, u.units // T)

    def work(t):
        ot, stt = u.new_oracle(), {}
        i = 0
        while not stop.is_set():
            t1 = time.perf_counter()
            u.run(ot, t * span + (i % span), stt)
            pers[t].append(time.perf_counter() - t1)
            i += 1
        counts[t] = i

    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    time.sleep(seconds * 0.8)
    stop.set()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    allp = [p for ps in pers for p in ps]
    variants["all_cores_frame_parallel"] = dict(value=sum(counts) / dt, units=sum(counts), cores=T, nproc=ncpu,
                                                mean_ms=1e3 * float(np.mean(allp)), median_ms=1e3 * float(np.median(allp)))
    ref = "two_threads_left_right" if wl.get("stereo") else "one_core"
    what = {"tum": "ORBextractor::operator()", "kitti": "operator() on L and R + ComputeStereoMatches",
            "euroc": "operator() + vocabulary transform + SearchByBoW(t-1,t)",
            "euroc_stereo": "cv::remap L and R + operator() on L and R + ComputeStereoMatches + vocabulary transform + SearchByBoW(t-1,t)"}[wlname]
    return dict(value=variants[ref]["value"], unit=unit, cores=variants[ref]["cores"], kind="port",
                sample=f"{variants[ref]['units']} units of the same synthetic {wl['w']}x{wl['h']} workload ({what}), "
                       f"oracle/orb_oracle.c (scalar C, gcc -O3), the reference's threading: {ref}",
                mean_ms=variants[ref]["mean_ms"], median_ms=variants[ref]["median_ms"], variants=variants)


# ------------------------------------------------------------------------------------------------
# one GPU workload
# ------------------------------------------------------------------------------------------------
class GpuWorkload:
    def __init__(self, name, frames, B, local_rank, torch, voc_arrays=None):
        import orb_slam2_annotate_amd as amd
        self.amd, self.torch, self.name = amd, torch, name
        wl = self.wl = WORKLOADS[name]
        self.W, self.H, self.B = wl["w"], wl["h"], B
        self.stereo, self.bow, self.raw = bool(wl.get("stereo")), bool(wl.get("bow")), bool(wl.get("raw"))
        self.frames = frames
        self.NI = NI = len(frames)
        assert NI == (2 * B if self.stereo else B)
        dev = self.dev = torch.device("cuda", local_rank)
        self.ext = ext = amd.ORBextractor(wl["nfeatures"], 1.2, 8, wl["ini"], wl["mn"], device=local_rank)
        self.cap = cap = ext.max_keypoints(self.W, self.H)
        self.d_img = torch.from_numpy(np.stack(frames)).to(dev)
        self.d_kp = torch.zeros((NI, cap, 7), dtype=torch.float32, device=dev)
        self.d_desc = torch.zeros((NI, cap, 32), dtype=torch.uint8, device=dev)
        self.d_n = torch.zeros((NI,), dtype=torch.int32, device=dev)
        if self.stereo:
            self.mbf = np.float32(wl["bf"])
            self.mb = np.float32(self.mbf / np.float32(wl["fx"]))  # src/Frame.cc:114
            self.d_u = torch.zeros((B, cap), dtype=torch.float32, device=dev)
            self.d_dep = torch.zeros((B, cap), dtype=torch.float32, device=dev)
            self.d_ns = torch.zeros((B,), dtype=torch.int32, device=dev)
        if self.raw:  # d_img holds the RAW pairs; the rectified ones are written here by the step
            self.maps = rectify_maps_of(wl)
            self.rect = (amd.Rectifier(*self.maps[0], device=local_rank), amd.Rectifier(*self.maps[1], device=local_rank))
            self.d_rect = torch.zeros((NI, self.H, self.W), dtype=torch.uint8, device=dev)
        if self.bow:
            self.voc = amd.ORBVocabulary(device=local_rank)
            assert self.voc.createFromArrays(voc_arrays)
            self.d_match = torch.zeros((B - 1, cap), dtype=torch.int32, device=dev)  # B = frames (mono) or pairs (stereo: left frames)
            self.d_nm = torch.zeros((B - 1,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()

    def step(self, n_units=None):
        """Enqueue one pass over the first n_units frames / stereo pairs (default: the whole batch); no host wait."""
        n_units = self.B if n_units is None else n_units
        ni = 2 * n_units if self.stereo else n_units
        W, H, cap, ext = self.W, self.H, self.cap, self.ext
        if self.raw:  # raw L at even / raw R at odd slots of d_img: two views with a frame stride of 2 images
            ext.extract_stereo_rectified_batch_device(self.rect[0], self.rect[1], self.d_img.data_ptr(), self.d_img.data_ptr() + W * H,
                                                      n_units, W, H, W, 2 * W * H, self.d_rect.data_ptr(), self.d_kp.data_ptr(),
                                                      self.d_desc.data_ptr(), cap, self.d_n.data_ptr())
        else:
            ext.extract_batch_device(self.d_img.data_ptr(), ni, W, H, W, W * H, self.d_kp.data_ptr(), self.d_desc.data_ptr(),
                                     cap, self.d_n.data_ptr(), wait=False)
        if self.stereo:
            ext.stereo_match_batch_device(n_units, self.d_kp.data_ptr(), self.d_desc.data_ptr(), self.d_n.data_ptr(), cap,
                                          float(self.mbf), float(self.mb), self.d_u.data_ptr(), self.d_dep.data_ptr(),
                                          self.d_ns.data_ptr())
        if self.bow:
            self.voc.bow_match_consecutive_batch_device(n_units, self.d_kp.data_ptr(), self.d_desc.data_ptr(),
                                                        self.d_n.data_ptr(), cap, self.d_match.data_ptr(),
                                                        self.d_nm.data_ptr(), nnratio=0.7, check_orientation=True,
                                                        levelsup=VOC_SHAPE[2], extractor=ext, stereo=self.stereo)

    def sync(self):
        self.ext.synchronize()
        self.torch.cuda.synchronize()

    def check(self, voc_arrays):
        """>= 3 units of the resident batch against the CPU oracle (first, middle = another sub-batch, last): every image
        of the unit (and of the previous one where the unit's SearchByBoW needs it) re-extracted by the oracle -- after the
        oracle's own cv::remap for raw pairs --, then the match outputs.  Returns the report and the oracle's work
        counters per unit of the matching stage."""
        if os.environ.get("ORBFE_BENCH_NO_CHECK"):  # kernel ablation runs (tools/ablate_desc_tiles.sh): results are NOT valid
            return {"ok": None, "skipped": "ORBFE_BENCH_NO_CHECK set: timing experiment, outputs unchecked"}, {"distance_pairs_per_unit": 0.0}
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_lib as orc
        wl, fr, B = self.wl, self.frames, self.B
        o = orc.Oracle(wl["nfeatures"], 1.2, 8, wl["ini"], wl["mn"])
        n = self.d_n.cpu().numpy()
        units = sorted({0, B // 2, B - 1})
        dist_st, dist_bow, matched, bow_matched, scanned, sads, cpu_ms = [], [], [], [], [], [], []
        vo = orc.Vocabulary.from_arrays(voc_arrays) if self.bow else None
        ipu = 2 if self.stereo else 1
        cache = {}

        def image(fi):  # oracle extraction of image slot fi, compared with the device outputs
            if fi in cache:
                return cache[fi]
            im = fr[fi]
            if self.raw:
                im = orc.remap_linear(im, *self.maps[fi & 1])
                if not np.array_equal(self.d_rect[fi].cpu().numpy(), im):
                    raise SystemExit(f"PARITY FAILURE ({self.name}): rectified image {fi} differs from the oracle's cv::remap")
            kr, dr, pr = o.extract(im, want_pyramid=True)
            n0 = int(n[fi])
            kg = self.d_kp[fi, :n0].cpu().numpy().view(np.uint8).reshape(n0, 28)
            if n0 != len(kr) or not np.array_equal(kg, kr.view(np.uint8).reshape(-1, 28)):
                raise SystemExit(f"PARITY FAILURE ({self.name}): keypoints of image {fi} differ from the oracle")
            if not np.array_equal(self.d_desc[fi, :n0].cpu().numpy(), dr):
                raise SystemExit(f"PARITY FAILURE ({self.name}): descriptors of image {fi} differ from the oracle")
            cache[fi] = (kr, dr, pr)
            return cache[fi]

        for ui in units:
            kL, dL, pL = image(ipu * ui)
            if self.stereo:
                kR, dR, pR = image(2 * ui + 1)
                orc.distance_calls_reset()
                t1 = time.perf_counter()
                u_ref, dep_ref = o.stereo(self.W, self.H, kL, dL, kR, dR, pL, pR, float(self.mbf), float(self.mb))
                cpu_ms.append(1e3 * (time.perf_counter() - t1))
                dist_st.append(orc.distance_calls())
                sc, sd = orc.stereo_counters()
                scanned.append(sc)
                sads.append(sd)
                matched.append(int((u_ref >= 0).sum()))
                if not (np.array_equal(self.d_u[ui, :len(kL)].cpu().numpy(), u_ref) and
                        np.array_equal(self.d_dep[ui, :len(kL)].cpu().numpy(), dep_ref)):
                    raise SystemExit(f"PARITY FAILURE ({self.name}): mvuRight/mvDepth of pair {ui} differ from the oracle")
            if self.bow and ui > 0:  # (left) keypoints of unit ui-1 as the key frame, of unit ui as the frame
                k0, de0, _ = image(ipu * (ui - 1))
                fv0, fv1 = orc.FeatVec(vo.transform(de0, VOC_SHAPE[2])[3]), orc.FeatVec(vo.transform(dL, VOC_SHAPE[2])[3])
                orc.distance_calls_reset()
                rn, rm = orc.search_by_bow(de0, np.ones(len(k0), np.uint8), k0["angle"], fv0, dL, kL["angle"], fv1, 0.7, True)
                dist_bow.append(orc.distance_calls() + (len(de0) + len(dL)) // 2 * VOC_SHAPE[0] * VOC_SHAPE[1])
                bow_matched.append(int(rn))
                if int(self.d_nm[ui - 1].item()) != rn or not np.array_equal(self.d_match[ui - 1, :len(kL)].cpu().numpy(), rm):
                    raise SystemExit(f"PARITY FAILURE ({self.name}): SearchByBoW({ui - 1},{ui}) differs from the oracle")
        rep = {"ok": True, "units_checked": units,
               "compared": ("rectified images (cv::remap) identical, " if self.raw else "")
                           + "keypoints (28-byte records) and descriptors bit-identical"
                           + (", mvuRight / mvDepth identical" if self.stereo else "")
                           + (", SearchByBoW match arrays identical" if self.bow else "")}
        mean = lambda v: float(np.mean(v)) if v else 0.0  # noqa: E731
        work = {"distance_pairs_per_unit": mean(dist_st) + mean(dist_bow)}
        if self.stereo:
            work.update({"stereo_matches_per_unit_oracle": mean(matched), "stereo_distance_pairs_per_unit": mean(dist_st),
                         "bucket_entries_scanned_per_unit": mean(scanned), "sad_refinements_per_unit": mean(sads),
                         "cpu_oracle_stereo_ms": mean(cpu_ms),
                         "note_stereo": "oracle counts on the checked units: row-bucket entries a left keypoint walks (src/Frame.cc:573-595), "
                                        "of which the octave (:579) and disparity-range (:584) gates let stereo_distance_pairs through to "
                                        "DescriptorDistance; SAD refinements = 11 windows of 11x11 each (:598-652)"})
        if self.bow:
            work.update({"bow_matches_per_unit_oracle": mean(bow_matched), "bow_distance_pairs_per_unit": mean(dist_bow),
                         "note_bow": "DescriptorDistance calls of SearchByBoW plus k*L node distances per feature of the frame's "
                                     "vocabulary descent (TemplatedVocabulary.h:1218-1259: %d x %d)" % (VOC_SHAPE[0], VOC_SHAPE[1])})
        return rep, work


def run_gpu_workload(name, frames, args, rank, world, local_rank, torch, dist, use_dist, voc_arrays, batches):
    wl = WORKLOADS[name]
    B = batches[name]
    g = GpuWorkload(name, frames, B, local_rank, torch, voc_arrays)
    ext, NI = g.ext, g.NI
    S = max(1, min(32, args.streams if args.streams > 0 else wl.get("streams", 8)))

    def barrier():
        g.sync()
        if use_dist:
            dist.barrier()

    # (1) warm-up on ONE stream, every stage bracketed by HIP events on its own stream: the exclusive
    #     (un-shared) duration of every stage
    ext.set_streams(1)
    ext.profile(False)
    g.step()   # the process's FIRST launch of every kernel (code objects, cold clocks and caches) stays out of the figures:
    g.sync()   # rounds 1-3 averaged it in, and the stage times read 10-20 % above rocprofv3's kernel durations of the same run
    ext.profile(True)
    n_warm = max(args.warmup, 2)
    for _ in range(n_warm):
        g.step()
        g.sync()
    warm = ext.profile_get()
    excl = {s_: warm[s_][0] / n_warm for s_ in GPU_STAGES}
    # (2) the timed configuration (S sub-batch streams), all stages still timed: the LIVE split -- which kernel
    #     the GPU spends its time in when the sub-batches overlap; the dominant stage is the largest one here
    ext.set_streams(S)
    ext.set_schedule(args.schedule == "lanes")
    ext.profile(False)
    g.step()
    g.sync()
    ext.profile(True)
    n_live = 2
    for _ in range(n_live):
        g.step()
    g.sync()
    livep = ext.profile_get()
    live = {s_: livep[s_][0] / n_live for s_ in GPU_STAGES}
    dom = max(GPU_STAGES, key=lambda s_: live[s_])
    # (3) timed region: K steps enqueued back to back; only the dominant stage keeps its events (one pair per
    #     sub-batch launch, on that sub-batch's own stream)
    #     The K-step block is repeated until >= --min-seconds have been timed (each block barrier-bracketed, max over
    #     ranks): `value` is the MEDIAN block, min / max / repeats are reported beside it.
    ext.profile([dom])
    blocks, enq = [], []
    while True:
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            g.step()
        enq.append(time.perf_counter() - t0)  # the host is done ENQUEUEING here; the device still runs
        barrier()
        dt = time.perf_counter() - t0
        dt_blk, total_units = reduce_report(dt, float(B * args.steps), torch, dist, use_dist, g.dev)  # same on every rank
        blocks.append(dt_blk)
        if sum(blocks) >= args.min_seconds or len(blocks) >= args.max_repeats:
            break
    prof = ext.profile_get()
    ext.profile(False)
    dt_max = float(np.median(blocks))
    # the same enqueue with EMPTY device queues (4 steps from a synchronised state): what the launching thread needs on its
    # own, without any back-pressure from full queues
    g.sync()
    t0 = time.perf_counter()
    for _ in range(4):
        g.step()
    enq_idle = (time.perf_counter() - t0) / 4
    g.sync()

    n_kp = float(g.d_n.float().mean().item())
    n_st = float(g.d_ns.float().mean().item()) if g.stereo else 0.0
    res = None
    if rank == 0:
        check, work = g.check(voc_arrays)
        dist_per_unit = work["distance_pairs_per_unit"]
        sizes = [ext.level_size(g.W, g.H, l) for l in range(ext.GetLevels())]
        alg = algorithmic_bytes(sizes, n_kp, wl, n_st)
        stage_kernels = dict(STAGE_KERNELS)
        kernel_label = {"k_fast_cells": "k_fast_cells<kLowFirst, FUSE_BLUR=false> (one wavefront per FAST grid cell)"}
        fused_pyramid_blur = excl["blur"] <= 0 < excl["pyramid"]
        if fused_pyramid_blur:
            # k_blur7<.., RESIZE>: one kernel per level blurs level l and writes level l+1 from the same staged tiles
            # (csrc/k_blur.hip); the "pyramid" stage then carries both stages' algorithmic bytes (the
            # fraction is computed from the algorithmic figure whatever a fused kernel really moves)
            alg["pyramid"] += alg["blur"]
            alg["blur"] = 0
            stage_kernels["pyramid"] = [("k_copy2d", 1), ("k_blur7", len(sizes))]
            kernel_label["k_blur7"] = "k_blur7<SPEC=0, 64, 64, RESIZE=true> (per level: 7x7 blur of level l + bilinear level l+1 from the same LDS tile)"
            stage_kernels["blur"] = []
        ipu = NI / B  # images per unit (2 for stereo)
        per_launch_units = {s_: (B if s_ == "match" else NI) for s_ in GPU_STAGES}  # exclusive launches: whole batch

        def stage_bytes(s_, images):  # algorithmic bytes of a launch group of stage s_ over `images` images
            return alg[s_] * (images / ipu if s_ == "match" else images)

        value = total_units / dt_max
        n_groups = max(prof[dom][1] / (1 if dom == "match" else sum(n for k, n in stage_kernels[dom] if k != "k_copy2d")), 1)
        dom_ms = prof[dom][0] / n_groups
        dom_imgs = prof[dom][2] / n_groups
        ach = stage_bytes(dom, prof[dom][2]) / (prof[dom][0] * 1e-3) / 1e9 if prof[dom][0] > 0 else 0.0
        stages = {}
        for s_ in GPU_STAGES:
            if excl[s_] <= 0:
                continue
            b = stage_bytes(s_, NI)
            stages[s_] = {"ms_per_step_exclusive": excl[s_], "ms_per_step_live": live[s_],
                          "algorithmic_bytes_per_step": b,
                          "hbm_frac_exclusive": b / (excl[s_] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "hbm_frac_live": (b / (live[s_] * 1e-3) / 1e9 / HBM_PEAK_GBS) if live[s_] > 0 else None}
        pipe_bytes = alg["extract_total"] * ipu + alg["match"]
        roof = {
            "bound": "hbm", "kernel": " + ".join(kernel_label.get(k, k) for k, _ in stage_kernels[dom] if k != "k_copy2d"),  # (k_copy2d only runs under $ORBFE_COPY_UNALIGNED)
            "stage": dom, "dominant_by": "largest live (multi-stream) HIP-event time of a step",
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": pmc_traffic(dom, name, dom_imgs, stage_kernels),
            "algorithmic_bytes_per_launch_group": stage_bytes(dom, dom_imgs), "ms_per_launch_group": dom_ms,
            "images_per_launch": dom_imgs, "launch_groups_timed": n_groups, "streams": S, "schedule": args.schedule,
            # the same kernel ALONE on the chip (1-stream warm-up pass, HIP events): the co-resident `frac` above shares the GPU
            # with the other sub-batches' kernels
            "exclusive": {"ms_per_step": excl[dom], "achieved": stage_bytes(dom, per_launch_units[dom] if dom != "match" else NI) / (excl[dom] * 1e-3) / 1e9 if excl[dom] > 0 else None,
                          "frac": stages.get(dom, {}).get("hbm_frac_exclusive")},
            "stages": stages, "fused_pyramid_blur": fused_pyramid_blur,
            "pipeline": {"algorithmic_bytes_per_unit": pipe_bytes, "achieved": pipe_bytes * value / world / 1e9,
                         "frac": pipe_bytes * value / world / 1e9 / HBM_PEAK_GBS,
                         "sum_exclusive_ms": sum(excl.values()), "ms_per_step": 1e3 * dt_max / args.steps},
        }
        if (g.stereo or g.bow) and excl["match"] > 0:
            pairs = dist_per_unit * B  # distance evaluations of one step (oracle count on the checked units x B)
            roof["matching"] = {
                "what": "256-bit Hamming distances of the matcher stage (the reference's DescriptorDistance calls, counted "
                        "by the oracle on the checked units), alone on the GPU",
                **{k: v for k, v in work.items() if not k.startswith("note")}, "distance_pairs_per_s": pairs / (excl["match"] * 1e-3),
                "popcount32_per_s": 8 * pairs / (excl["match"] * 1e-3), "peak": BCNT_PEAK, "unit": "popcount-32 lane-ops/s",
                "frac": 8 * pairs / (excl["match"] * 1e-3) / BCNT_PEAK,
                "note": "v_bcnt_u32_b32 peak measured by tools/ubench/valu_rate.hip (0.585 T wave-instr/s x 64 lanes); the "
                        "stage is latency-bound (row buckets / node lists of a few dozen candidates per query), not popcount-bound"}
        v = pmc_valu(name)
        if v is not None:
            key = "valu_wave_instr_per_image" if "total_valu_wave_instr_per_image" in v else "valu_wave_instr_per_frame"
            tot = v.get("total_" + key)
            per = {}
            for s_ in GPU_STAGES:
                ks = [k for k, _ in stage_kernels[s_] if k in v["kernels"]]
                if not ks or excl[s_] <= 0:
                    continue
                wi = sum(v["kernels"][k][key] for k in ks) * NI
                busy = [v["kernels"][k].get("valu_busy") for k in ks if v["kernels"][k].get("valu_busy") is not None]
                per[s_] = {"wave_instr_per_step": wi, "frac_exclusive": wi / (excl[s_] * 1e-3) / VALU_PEAK,
                           "valu_busy_pmc": (max(busy) if busy else None)}
            roof["valu_issue"] = {
                "peak": VALU_PEAK, "unit": "wave64 VALU instr/s", "measured_class_ceilings": VALU_MEASURED,
                "source": v.get("_file"), "stages": per,
                "pipeline_frac": (tot * NI * (value / world / B) / VALU_PEAK) if tot else None,
                "note": "peak = 2 cycles per wave64 instruction per SIMD (guide); SQ_INSTS_VALU from the committed PMC "
                        "pass; valu_busy_pmc = 4*SQ_ACTIVE_INST_VALU/(SIMDs*GRBM_GUI_ACTIVE/8 XCDs) of the stage's busiest kernel: the hardware's own VALU utilisation"}
            best_valu = max([p["frac_exclusive"] for p in per.values()] + [0.0])
            if dom in per and per[dom]["frac_exclusive"] > stages.get(dom, {}).get("hbm_frac_exclusive", 0):
                roof["bound_closest"] = "valu_issue"
            else:
                roof["bound_closest"] = "hbm"
            roof["valu_issue"]["max_stage_frac"] = best_valu
        res = {"workload": wl["name"], "key": name, "value": value,
               "unit": "stereo frames/s" if g.stereo else "frames/s", "ms_per_step": 1e3 * dt_max / args.steps,
               "repeats": {"blocks_of_K_steps": len(blocks), "timed_seconds": float(sum(blocks)), "value_is": "median block",
                           "value_min": total_units / max(blocks), "value_max": total_units / min(blocks),
                           "spread_pct": 100.0 * (max(blocks) - min(blocks)) / dt_max,
                           # host time to ENQUEUE a step (all HIP calls of the step issued, nothing waited for): well below
                           # ms_per_step = the device, not the launching thread, sets the rate
                           "host_enqueue_ms_per_step": 1e3 * float(np.median(enq)) / args.steps,
                           "host_enqueue_ms_per_step_idle_queues": 1e3 * enq_idle},
               "units_per_gpu_per_step": B, "images_per_gpu_per_step": NI, "keypoints_per_image": n_kp,
               "stereo_matches_per_frame": n_st if g.stereo else None, "matching_work": work, "roofline": roof,
               "parity_check": check}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(name, frames[: min(len(frames), 256)], voc_arrays, args.cpu_seconds)
            res["vs_cpu_baseline"] = value / res["cpu_baseline"]["value"]
        if not args.no_e2e and not g.raw and world == 1:
            res["e2e"] = e2e_rate(g, args)
    del g
    torch.cuda.empty_cache()
    return res


def e2e_rate(g, args):
    """Host images in -> host keypoints/descriptors out (what ORBextractor::operator() is, :1119-1197) through
    orbfe_extract_batch_pipelined: pinned host buffers, chunked H2D / kernels / D2H overlapped on separate streams.
    PCIe-inclusive, reported beside `value`, never as `value`; extraction only (the matchers take device arrays)."""
    n = min(g.NI, 2048)
    ext = g.ext
    ext.set_streams(2)
    img, kps, desc, cnt = ext.pinned_buffers(n, g.H, g.W, g.cap)
    np.copyto(img, np.stack(g.frames[:n]))
    ext.extract_pinned(args.e2e_chunk)  # warm: device slabs, copy streams
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        ext.extract_pinned(args.e2e_chunk)
    dt = (time.perf_counter() - t0) / reps
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as orc
    o = orc.Oracle(g.wl["nfeatures"], 1.2, 8, g.wl["ini"], g.wl["mn"])
    for fi in sorted({0, n // 2 + 1, n - 1}):  # parity of the pipelined path too (first / middle / last chunk)
        kr, dr = o.extract(g.frames[fi])
        if int(cnt[fi]) != len(kr) or not np.array_equal(kps[fi, :len(kr)].view(np.uint8), kr.view(np.uint8)) or \
                not np.array_equal(desc[fi, :len(kr)], dr):
            raise SystemExit(f"PARITY FAILURE ({g.name}): pipelined host path, image {fi}")
    bytes_in = n * g.W * g.H
    bytes_out = n * g.cap * (28 + 32) + 4 * n
    return {"images_per_s": n / dt, "images": n, "chunk_frames": args.e2e_chunk or 256, "ms": 1e3 * dt,
            "pcie_GBps": {"h2d": bytes_in / dt / 1e9, "d2h": bytes_out / dt / 1e9},
            "what": "orbfe_extract_batch_pipelined: pinned host images -> H2D / kernels / D2H overlapped on separate streams -> "
                    "host keypoints + descriptors (whole capacity-sized output blocks come back); oracle-checked on 3 images"}


class LazyDist:
    """torch.distributed, initialised at the FIRST collective instead of at start-up.  Round 4 measured that a live RCCL
    communicator in the process costs the single-GPU pipeline 11 % when it is created BEFORE the extractor's HIP streams
    (104.0 k -> 92.5 k stereo frames/s; gloo: no effect; tools/dist_ab.sh): the first collective of a workload is the
    barrier in front of its timed region, i.e. after the warm-up passes have created every sub-batch stream."""

    def __init__(self, dist, backend, rank, world, device, eager=False):
        self.dist, self.backend, self.rank, self.world, self.device, self.up = dist, backend, rank, world, device, False
        if eager:
            self._ensure()

    def _ensure(self):
        if self.up:
            return
        if self.backend == "nccl":
            self.dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.device)
        else:
            self.dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
        self.up = True

    @property
    def ReduceOp(self):
        return self.dist.ReduceOp

    def barrier(self):
        self._ensure()
        self.dist.barrier()

    def all_reduce(self, t, op=None):
        self._ensure()
        self.dist.all_reduce(t, op=op)

    def shutdown(self):
        if self.up:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.up = False


def reduce_report(dt, units, torch, dist, use_dist, dev):
    if not use_dist:
        return dt, units
    if getattr(dist, "backend", "nccl") != "nccl":
        dev = "cpu"  # gloo (ranks sharing one GPU in the rehearsal test): host tensors
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    n = torch.tensor([units], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), float(n.item())


def kitti_seq_entry(plans, lengths, pool):
    """configs[4] figures as one JSON object (its own line for --workload kitti_seq, a `secondary` of the default line
    when --gpus N > 1: one driver run then yields the weak-scaling headline AND the strong-scaling sequence plans)"""
    wl = WORKLOADS["kitti"]
    return {"key": "kitti_seq", "value": plans["sequence"]["value"], "unit": "stereo frames/s", "scaling": "strong",
            "ms_per_step": plans["sequence"]["ms_per_step"],
            "workload": "KITTI 00-07 (sequence lengths %s; synthetic 1241x376 stereo pairs cycled from a resident pool of %d), "
                        "nFeatures=2000, extract L+R + ComputeStereoMatches" % (lengths, pool),
            "sharding": "plan 'sequence': sequence s -> rank s mod N (configs[4]); plan 'round_robin': frames dealt evenly; no "
                        "data-path collective", "per_image": wl["name"], "plans": plans}


def run_kitti_seq(frames, args, rank, world, local_rank, torch, dist, use_dist, batches):
    """configs[4]: the KITTI 00-07 sequences (lengths only -- the images are synthetic pairs cycled from the
    rank's resident pool) dealt to the ranks by shard.py; one step = every rank runs its share once."""
    from orb_slam2_annotate_amd import shard
    B = batches["kitti"]
    g = GpuWorkload("kitti", frames, B, local_rank, torch)
    g.ext.set_streams(max(1, min(32, args.streams if args.streams > 0 else WORKLOADS["kitti"]["streams"])))
    g.ext.set_schedule(args.schedule == "lanes")
    lengths = [max(1, int(round(n * args.seq_scale))) for n in shard.KITTI_00_07]
    out = {}
    for mode in ("sequence", "round_robin"):
        plan = shard.shard_sequences(lengths, world, mode)
        mine, longest = shard.frames_of(plan[rank]), max(shard.frames_of(p) for p in plan)

        def one_pass():
            left = mine
            while left > 0:
                g.step(min(B, left))
                left -= min(B, left)

        for _ in range(max(args.warmup, 1)):
            one_pass()
        g.sync()
        if use_dist:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_pass()
        g.sync()
        if use_dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        dt_max, total = reduce_report(dt, float(mine * args.steps), torch, dist, use_dist, g.dev)
        out[mode] = {"value": total / dt_max, "ms_per_step": 1e3 * dt_max / args.steps, "stereo_frames_per_step": total / args.steps,
                     "frames_of_rank0": mine, "max_frames_of_a_rank": longest}
    check = g.check(None)[0] if rank == 0 else None
    del g
    return out, lengths, check



# ------------------------------------------------------------------------------------------------
# the ONE stdout line: compact (< 4 KB); everything else goes to the detail file
# ------------------------------------------------------------------------------------------------
LINE_LIMIT = 4096  # bytes; the driver keeps an 8 KB tail of stdout and parses the LAST line (round 3 printed 31 KB: unparsed)


def _r(x, sig=6):
    """floats to `sig` significant digits (the detail file keeps full precision)"""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}")
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and d.get(k) is not None}


def compact_line(full, detail_path=None):
    """The contract line from the full result dict: metric / value / config / roofline (dominant kernel + pipeline +
    per-stage exclusive ms) / cpu_baseline / parity_check / repeats, one short object per secondary workload.  Stage
    tables, CPU variants, e2e, matching work and the latency rows stay in the detail file named by `detail`."""
    out = _pick(full, ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"])
    out["vs_baseline"] = full.get("vs_baseline")
    out.update(_pick(full, ["dtype", "data", "vs_cpu_baseline", "stereo_matches_per_frame"]))
    cfg = dict(full.get("config", {}))
    if len(str(cfg.get("workload", ""))) > 200:
        cfg["workload"] = cfg["workload"][:197] + "..."
    out["config"] = cfg
    rf = full.get("roofline")
    if rf:
        c = _pick(rf, ["bound", "bound_closest", "kernel", "stage", "achieved", "peak", "unit", "frac", "traffic",
                       "algorithmic_bytes_per_launch_group", "ms_per_launch_group", "images_per_launch", "streams"])
        if len(str(c.get("kernel", ""))) > 120:
            c["kernel"] = c["kernel"][:117] + "..."
        c.setdefault("traffic", None)
        if rf.get("exclusive"):
            c["exclusive"] = rf["exclusive"]
        if rf.get("pipeline"):
            c["pipeline"] = rf["pipeline"]
        if rf.get("valu_issue"):
            c["valu_issue"] = _pick(rf["valu_issue"], ["pipeline_frac", "max_stage_frac"])
        if rf.get("stages"):
            c["stages_exclusive_ms"] = {k: v["ms_per_step_exclusive"] for k, v in rf["stages"].items()}
        out["roofline"] = c
    cb = full.get("cpu_baseline")
    if cb:
        c = _pick(cb, ["value", "unit", "cores", "kind", "sample", "mean_ms", "median_ms"])
        if len(str(c.get("sample", ""))) > 300:
            c["sample"] = c["sample"][:297] + "..."
        out["cpu_baseline"] = c
    if full.get("parity_check"):
        out["parity_check"] = _pick(full["parity_check"], ["ok", "units_checked", "skipped"])
    if full.get("repeats"):
        out["repeats"] = _pick(full["repeats"], ["blocks_of_K_steps", "timed_seconds", "value_is", "value_min", "value_max", "spread_pct"])
    sec = []
    for s_ in full.get("secondary") or []:
        e = _pick(s_, ["key", "value", "unit", "ms_per_step", "vs_cpu_baseline", "scaling"])
        pf = (s_.get("roofline") or {}).get("pipeline") or {}
        if pf.get("frac") is not None:
            e["pipeline_frac"] = pf["frac"]
        if (s_.get("parity_check") or {}).get("ok") is not None:
            e["parity_ok"] = s_["parity_check"]["ok"]
        if s_.get("plans"):  # configs[4]: both sharding plans, value + the longest share
            e["plans"] = {m: _pick(p, ["value", "ms_per_step", "stereo_frames_per_step", "frames_of_rank0", "max_frames_of_a_rank"])
                          for m, p in s_["plans"].items()}
        sec.append(e)
    if sec:
        out["secondary"] = sec
    for k in ("stub", "total_units", "plans"):
        if k in full:
            out[k] = full[k]
    if full.get("latency"):
        out["latency_rows"] = len(full["latency"].get("rows", []))
    if detail_path:
        out["detail"] = str(detail_path)
    out = _r(out)
    line = json.dumps(out, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:  # never again an unparseable line: shed the optional parts, largest first
        for k in ("secondary", "repeats", "stereo_matches_per_frame"):
            out.pop(k, None)
            line = json.dumps(out, separators=(",", ":"))
            if len(line) < LINE_LIMIT:
                break
    return line


def emit(full, args):
    """Write the full result to the detail file and print the compact line LAST on stdout (`--full-line`: the full dict
    on stdout instead, what the tools/ scripts parse)."""
    path = None
    if not args.no_detail:
        path = Path(args.detail_out) if args.detail_out else ROOT / "gpurun_out" / "bench_detail.json"
        try:
            path.parent.mkdir(parents=True, exist_ok=True)
            path.write_text(json.dumps(full, indent=1))
        except OSError as e:
            print(f"bench.py: could not write the detail file {path}: {e}", file=sys.stderr)
            path = None
    if args.full_line:
        print(json.dumps(full), flush=True)
        return
    rel = path
    if path is not None:
        try:
            rel = path.resolve().relative_to(ROOT)
        except ValueError:
            pass
    print(compact_line(full, rel), flush=True)


# ------------------------------------------------------------------------------------------------
def stub_worker(args, rank, world):
    """The launcher / barrier / reduction protocol with a CPU stub step (tests/test_shard_gloo.py): no GPU."""
    import torch
    import torch.distributed as dist
    use_dist = world > 1
    if use_dist:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    units = 100 + rank
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (1 + rank))
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    dt_max, total = reduce_report(dt, float(units * args.steps), torch, dist, use_dist, torch.device("cpu"))
    out = {"metric": "stub", "value": total / dt_max, "unit": "units/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * dt_max / args.steps, "stub": True, "total_units": total}
    if world > 1:  # the shape of the real line for --gpus N > 1: the configs[4] plans as a `secondary`
        from orb_slam2_annotate_amd import shard
        lengths = [max(1, int(round(n * args.seq_scale))) for n in shard.KITTI_00_07]
        plans = {}
        for mode in ("sequence", "round_robin"):
            plan = shard.shard_sequences(lengths, world, mode)
            mine, longest = shard.frames_of(plan[rank]), max(shard.frames_of(p) for p in plan)
            if use_dist:
                dist.barrier()
            t0 = time.perf_counter()
            time.sleep(1e-6 * mine)  # a stub "pass": 1 us per stereo frame of this rank's share
            if use_dist:
                dist.barrier()
            dtm, tot = reduce_report(time.perf_counter() - t0, float(mine), torch, dist, use_dist, torch.device("cpu"))
            plans[mode] = {"value": tot / dtm, "ms_per_step": 1e3 * dtm, "stereo_frames_per_step": tot, "frames_of_rank0": mine,
                           "max_frames_of_a_rank": longest}
        out["secondary"] = [kitti_seq_entry(plans, lengths, 0)]
    if rank == 0:
        emit(out, args)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--min-seconds", type=float, default=1.5, help="repeat the timed block of K steps until this much has been "
                    "timed per workload; value = median block (min / max / repeats reported)")
    ap.add_argument("--max-repeats", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="units (frames / stereo pairs) resident per GPU and processed "
                    "per step; 0 = per-workload default (tum 4096, kitti 512, euroc 2048)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["all", "auto", "kitti_seq"], default="auto",
                    help="auto (default) = all four workloads on ONE GPU; with --gpus N > 1 the headline only (kitti, the fast path: "
                         "a rank renders and runs one workload) plus the configs[4] kitti_seq plans; 'all' forces the secondaries on N > 1 too")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="CPU-baseline budget per variant and workload")
    ap.add_argument("--streams", type=int, default=0,
                    help="sub-batch HIP streams per call in the timed region (1..32; 0 = per-workload default from "
                         "profiles/r02_stream_batch_sweep.txt: kitti 8, tum 8, euroc 4): the latency-bound kernels "
                         "of one sub-batch overlap the VALU-bound ones of the others")
    ap.add_argument("--schedule", choices=["streams", "lanes"], default="streams",
                    help="sub-batches on independent streams, or as the three-lane software pipeline (pyramid | FAST+blur | tail)")
    ap.add_argument("--render-procs", type=int, default=0, help="processes rendering the synthetic inputs (0 = auto; 1 = in "
                    "this process, which a run under a GPU-initialising profiler needs)")
    ap.add_argument("--single-scene", action="store_true", help="mono workloads: render ONE scene for the whole batch, the "
                    "round-1 input (sparser than the default 64-scene batches); for continuity with the round-1 line")
    ap.add_argument("--stereo-scene", choices=["textured", "shapes"], default="textured",
                    help="synthetic stereo pairs: 'textured' = one scene in both eyes with a piecewise-planar sub-pixel disparity field "
                         "(>= 50 %% of the left keypoints obtain a stereo match; default and headline), 'shapes' = the round-1/2 "
                         "generator (per-object integer shifts, ~14 %% match)")
    ap.add_argument("--voc-shape", default="10,6,4", help="k,L,levelsup of the euroc workload's synthetic vocabulary (default: ORBvoc.txt's "
                    "shape with the levelsup of src/Frame.cc:438; 10,2,0 = the cache-resident tree of rounds 1-2)")
    ap.add_argument("--dataset", default=os.environ.get("ORBFE_DATASET"), help="real frames for one workload: kitti:<sequence dir> | "
                    "euroc:<cam0 dir>,<cam1 dir>,<timestamps file> | tum:<sequence dir> (image lists as the reference's example mains "
                    "read them, orb_slam2_annotate_amd/datasets.py); default: $ORBFE_DATASET, else synthetic")
    ap.add_argument("--input-cache", default=None, help="directory for the rendered synthetic batches (.npy), reused when present")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-call latency block (tools/matcher_latency.py)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-in/host-out (PCIe-inclusive) rate of each workload")
    ap.add_argument("--e2e-chunk", type=int, default=0, help="frames per chunk of the pipelined host path (0 = 256)")
    ap.add_argument("--seq-scale", type=float, default=1.0, help="kitti_seq: scale factor on the sequence lengths")
    ap.add_argument("--dist-backend", default="nccl")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even for 1 rank")
    ap.add_argument("--eager-dist", action="store_true", help="create the process group at start-up (rounds 1-3) instead of at the first "
                    "collective, i.e. after the extractor's streams exist (LazyDist: RCCL created first costs the pipeline 11 %%)")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--full-line", action="store_true", help="print the FULL result dict as the stdout line (tens of KB: stage tables, "
                    "CPU variants, e2e, latency rows) instead of the compact contract line; what the tools/ scripts parse")
    ap.add_argument("--detail-out", default=None, help="file for the full result dict (default: gpurun_out/bench_detail.json)")
    ap.add_argument("--no-detail", action="store_true", help="do not write the detail file")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))  # the parent only spawns, waits and relays: no torch, no HIP

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if args.stub:
        return stub_worker(args, rank, world)

    if args.workload == "auto":
        args.workload = "all" if world == 1 else "kitti"
    names = ["kitti", "tum", "euroc", "euroc_stereo"] if args.workload == "all" else (["kitti"] if args.workload == "kitti_seq" else [args.workload])
    batches = {nm: (args.batch if args.batch > 0 else WORKLOADS[nm]["batch"]) for nm in names}
    global SINGLE_SCENE, STEREO_SCENE, VOC_SHAPE
    VOC_SHAPE = tuple(int(x) for x in args.voc_shape.split(","))
    if len(VOC_SHAPE) != 3 or VOC_SHAPE != (10, 6, 4):
        WORKLOADS["euroc"]["name"] = WORKLOADS["euroc"]["name"].replace("ORBvoc's shape k=10 L=6, levelsup=4", "k=%d L=%d, levelsup=%d" % VOC_SHAPE)
    SINGLE_SCENE = args.single_scene
    STEREO_SCENE = args.stereo_scene
    inputs = render_inputs(names, batches, rank, args.render_procs, None if args.single_scene else args.input_cache)
    data_tag = "synthetic"
    if args.dataset:  # real frames for the one workload whose layout the spec names (kitti | euroc | tum)
        from orb_slam2_annotate_amd import datasets
        kind = args.dataset.partition(":")[0]
        if kind not in names:
            raise SystemExit(f"--dataset {kind}:... needs --workload {kind} (or all)")
        B = batches[kind]
        wl = WORKLOADS[kind]
        _, fr = datasets.load_frames(args.dataset, B, device=local_rank)
        if kind == "euroc":
            fr = fr[0::2]  # the euroc workload (configs[3]) is the left stream: extract + SearchByBoW(t-1, t)
        bad = [f.shape for f in fr if f.shape != (wl["h"], wl["w"])]
        if bad:
            raise SystemExit(f"--dataset: frames of {bad[0][1]}x{bad[0][0]}, the {kind} workload is {wl['w']}x{wl['h']}")
        inputs[kind] = fr
        data_tag = f"synthetic, except {kind}: {args.dataset}"  # may fork workers: before torch / HIP

    import torch
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # one rank per GPU; with fewer devices than ranks (the 2-rank rehearsal of configs[4] on a one-GPU box, gloo backend:
    # RCCL refuses two ranks on one device) the ranks share the devices round-robin and the line says so
    n_dev = max(1, torch.cuda.device_count())
    ranks_per_gpu = (world + n_dev - 1) // n_dev
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)  # before the process group: RCCL binds the communicator to the current device
    if use_dist:
        dist = LazyDist(dist, args.dist_backend, rank, world, torch.device("cuda", local_rank), eager=args.eager_dist)

    voc_arrays = None
    if "euroc" in names or "euroc_stereo" in names:
        from orb_slam2_annotate_amd.vocabulary import synthetic_vocabulary_arrays
        voc_arrays = synthetic_vocabulary_arrays(VOC_SHAPE[0], VOC_SHAPE[1], seed=1)

    if args.workload == "kitti_seq":
        plans, lengths, check = run_kitti_seq(inputs["kitti"], args, rank, world, local_rank, torch, dist, use_dist, batches)
        if rank == 0:
            ent = kitti_seq_entry(plans, lengths, batches["kitti"])
            emit({
                "metric": "ORB extract+match stereo frames/sec, KITTI 00-07 sharded over the GPUs (bit-exact vs CPU oracle on checked frames)",
                "value": ent["value"], "unit": "stereo frames/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": ent["ms_per_step"], "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                "config": {"workload": ent["workload"], "sharding": ent["sharding"], "per_image": ent["per_image"]},
                "plans": plans, "parity_check": check}, args)
    else:
        results = []
        kitti_frames = inputs.get("kitti") if world > 1 else None  # (kept for the configs[4] plans below)
        for nm in names:
            r = run_gpu_workload(nm, inputs.pop(nm), args, rank, world, local_rank, torch, dist, use_dist, voc_arrays, batches)
            results.append(r)
        if kitti_frames is not None:
            # more than one GPU: the KITTI 00-07 `sequence` / `round_robin` plans (configs[4], strong scaling) ride along
            plans, lengths, _ = run_kitti_seq(kitti_frames, args, rank, world, local_rank, torch, dist, use_dist, batches)
            if rank == 0:
                results.append(kitti_seq_entry(plans, lengths, batches["kitti"]))
        if rank == 0:
            head = results[0]
            out = {
                "metric": "ORB extract+match frames/sec (kp/desc/matches bit-exact vs CPU oracle on the checked frames of every run)",
                "value": head["value"], "unit": head["unit"], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                # BASELINE.md publishes no number (the reference has none) -> null; the CPU oracle timed in this run, in the
                # reference's own threading for the workload, is `cpu_baseline` and the ratio to it `vs_cpu_baseline`
                "vs_baseline": None,
                "dtype": "u8", "data": data_tag,
                "config": {"workload": head["workload"] + (" [single-scene round-1 input]" if args.single_scene else ""), "units_per_gpu_per_step": head["units_per_gpu_per_step"],
                           "images_per_gpu_per_step": head["images_per_gpu_per_step"],
                           "keypoints_per_image": head["keypoints_per_image"],
                           "sharding": f"independent frames, one resident batch per GPU x{world}, no data-path collective"},
                "roofline": head["roofline"], "parity_check": head["parity_check"],
            }
            if ranks_per_gpu > 1:
                out["config"]["ranks_per_gpu"] = ranks_per_gpu  # oversubscribed rehearsal: the value is not a rate of N GPUs
            for k in ("repeats", "cpu_baseline", "vs_cpu_baseline", "e2e", "stereo_matches_per_frame", "matching_work"):
                if head.get(k) is not None:
                    out[k] = head[k]
            if len(results) > 1:
                out["secondary"] = results[1:]
            if world == 1 and not args.no_latency and not args.no_cpu_baseline:
                # the live system's call pattern (one call at a time): tools/matcher_latency.py, GPU next to the CPU oracle
                sys.path.insert(0, str(ROOT / "tools"))
                import matcher_latency
                rows = matcher_latency.measure(reps=12, verbose=False)
                out["latency"] = {"what": "median ms of ONE call through the class-level API on a 1241x376 / 2000-feature frame (1000 projected "
                                          "map points), GPU next to the CPU oracle on one core; 'resident' = frames uploaded once with "
                                          "orbfe_frame_upload; multi-neighbour rows give the per-neighbour cost of one call",
                                  "rows": rows}
            emit(out, args)
    if use_dist:
        dist.shutdown()


if __name__ == "__main__":
    main()
