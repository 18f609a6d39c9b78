#!/usr/bin/env python3
"""Frame::ComputeBoW (src/Frame.cc:433-440) on the device: a cache-resident k=10 L=2 tree against ORBvoc's shape
(k=10, L=6, 1 111 111 nodes, levelsup=4).  Times (a) orbfe_vocabulary_transform on host descriptors of ONE frame
(upload, descent, download), (b) orbfe_vocabulary_featvec_batch_device on 1 / 256 / 2048 device-resident frames, next to
the CPU oracle's transform of one frame.  Output: profiles/rNN_vocab_latency.txt"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as orc  # noqa: E402
import orb_slam2_annotate_amd as amd  # noqa: E402
from orb_slam2_annotate_amd import synth  # noqa: E402
from orb_slam2_annotate_amd.vocabulary import synthetic_vocabulary_arrays  # noqa: E402


def med(fn, reps=30):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(t))


def main():
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    e = amd.ORBextractor(nf, 1.2, 8, 20, 7)
    frames = [synth.render_frame(100 + i, 1241, 376) for i in range(8)]
    descs = [e(f)[1] for f in frames]
    n0 = len(descs[0])
    print(f"# {n0} descriptors per frame (1241x376, nFeatures={nf}); median of 30 calls, ms")
    dev = torch.device("cuda", 0)
    for (k, L, ls) in ((10, 2, 0), (10, 6, 4)):
        arrays = synthetic_vocabulary_arrays(k, L, 1)
        voc = amd.ORBVocabulary()
        assert voc.createFromArrays(arrays)
        vo = orc.Vocabulary.from_arrays(arrays)
        w, wt, nd = voc.transform_features(descs[0], ls)
        u, w0, wt0, nd0 = vo.transform(descs[0], ls)
        assert np.array_equal(w, w0) and np.array_equal(wt, wt0) and np.array_equal(nd, nd0)
        row = {"host_1_frame": med(lambda: voc.transform_features(descs[0], ls)),
               "cpu_oracle_1_frame": med(lambda: vo.transform(descs[0], ls), 10)}
        cap = max(len(d) for d in descs)
        for B in (1, 256, 2048):
            dd = np.zeros((B, cap, 32), np.uint8)
            dn = np.zeros(B, np.int32)
            for i in range(B):
                d = descs[i % len(descs)]
                dd[i, :len(d)] = d
                dn[i] = len(d)
            d_desc = torch.from_numpy(dd).to(dev)
            d_n = torch.from_numpy(dn).to(dev)
            d_nodes = torch.zeros((B, cap), dtype=torch.int32, device=dev)
            d_off = torch.zeros((B, cap + 1), dtype=torch.int32, device=dev)
            d_idx = torch.zeros((B, cap), dtype=torch.int32, device=dev)
            d_cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            row[f"device_{B}_frames"] = med(lambda: voc.featvec_batch_device(
                d_desc.data_ptr(), d_n.data_ptr(), B, cap, d_nodes.data_ptr(), d_off.data_ptr(), d_idx.data_ptr(),
                d_cnt.data_ptr(), levelsup=ls), 20)
        print(f"k={k} L={L} levelsup={ls} nodes={voc.info()['nodes']}: " + "  ".join(f"{a}={b:.3f}" for a, b in row.items()))


if __name__ == "__main__":
    main()
