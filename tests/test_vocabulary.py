"""DBoW2 vocabulary: text loader + transform + FeatureVector, and the
device-resident ComputeBoW + SearchByBoW batch, against the oracle."""
import numpy as np
import pytest

import oracle_lib as orc
from orb_slam2_annotate_amd import synth
from orb_slam2_annotate_amd.vocabulary import (synthetic_vocabulary_arrays, write_synthetic_vocabulary,
                                               write_vocabulary_text)


def _brute_transform(path, desc, levelsup):
    """independent numpy restatement of the k-ary descent, used to pin the oracle"""
    lines = open(path).read().split("\n")
    k, L = [int(x) for x in lines[0].split()[:2]]
    parent, leaf, d, w = [0], [0], [np.zeros(32, np.uint8)], [0.0]
    for ln in lines[1:]:
        if not ln.strip():
            continue
        t = ln.split()
        parent.append(int(t[0])); leaf.append(int(t[1])); d.append(np.array(t[2:34], dtype=np.int64).astype(np.uint8)); w.append(float(t[34]))
    children = [[] for _ in parent]
    for i in range(1, len(parent)):
        children[parent[i]].append(i)
    words = {}
    for i in range(1, len(parent)):
        if leaf[i] > 0:
            words[i] = len(words)
    out = []
    for f in desc:
        node, level, nid = 0, 0, 0
        while True:
            level += 1
            ch = children[node]
            dist = [int(np.unpackbits(f ^ d[c]).sum()) for c in ch]
            node = ch[int(np.argmin(dist))]  # argmin = first minimum
            if level == L - levelsup:
                nid = node
            if not children[node]:
                break
        out.append((words.get(node, 0), w[node], nid))
    return out


def test_oracle_vocabulary_load_and_transform(tmp_path):
    path = tmp_path / "voc.txt"
    n = write_synthetic_vocabulary(path, k=5, L=3, seed=1)
    v = orc.Vocabulary(path)
    assert v.info() == dict(k=5, L=3, nodes=n, words=125)
    rng = np.random.default_rng(3)
    desc = rng.integers(0, 256, size=(200, 32), dtype=np.uint8)
    for levelsup in (1, 2, 4):
        used, word, weight, node = v.transform(desc, levelsup)
        ref = _brute_transform(path, desc, levelsup)
        assert used == 200
        assert [int(x) for x in word] == [r[0] for r in ref]
        assert np.allclose(weight, [r[1] for r in ref], rtol=0, atol=0)
        assert [int(x) for x in node] == [r[2] for r in ref]
    # a file without a trailing newline loads identically
    open(tmp_path / "voc2.txt", "w").write(open(path).read().rstrip("\n"))
    assert orc.Vocabulary(tmp_path / "voc2.txt").info() == v.info()


@pytest.mark.gpu
@pytest.mark.parametrize("k,L,levelsup", [(10, 2, 0), (5, 3, 1), (10, 3, 2), (3, 5, 4)])
def test_gpu_transform_matches_oracle(tmp_path, k, L, levelsup):
    import orb_slam2_annotate_amd as amd
    path = tmp_path / "voc.txt"
    write_synthetic_vocabulary(path, k=k, L=L, seed=k * 10 + L)
    vo = orc.Vocabulary(path)
    voc = amd.ORBVocabulary()
    assert voc.loadFromTextFile(path)
    assert voc.info() == vo.info()
    e = amd.ORBextractor(800, 1.2, 8, 20, 7)
    kps, desc = e(synth.render_frame(5))
    word, weight, node = voc.transform_features(desc, levelsup)
    used, w_ref, wt_ref, n_ref = vo.transform(desc, levelsup)
    assert np.array_equal(word, w_ref) and np.array_equal(weight, wt_ref) and np.array_equal(node, n_ref)
    bow, fv = voc.transform(desc, levelsup)
    assert abs(sum(bow.values()) - 1.0) < 1e-12 and len(fv.indices) == used
    assert not voc.loadFromTextFile(tmp_path / "missing.txt")


def _irregular_tree(seed, max_children=20, depth=5, n_max=4000):
    """A tree no k-means would build but the text format allows: 1..max_children children per node (17..20 take the
    second trip of the 16-lane descent), leaves at every depth, breadth-first ids, near-duplicate siblings (ties)."""
    rng = np.random.default_rng(seed)
    parent, leaf, desc, weight = [], [], [], []
    level = [(0, rng.integers(0, 256, 32, dtype=np.uint8))]
    nid = 1
    for d in range(1, depth + 1):
        nxt = []
        for pid, pdesc in level:
            nc = int(rng.integers(1, max_children + 1)) if d > 1 else max_children
            for c in range(nc):
                dd = pdesc ^ (rng.integers(0, 256, 32, dtype=np.uint8) & rng.integers(0, 256, 32, dtype=np.uint8))
                if c and rng.random() < 0.15:
                    dd = desc[-1].copy()  # an exact duplicate of the previous sibling: the first one must win
                is_leaf = d == depth or nid > n_max or (d > 1 and rng.random() < 0.2)
                parent.append(pid); leaf.append(1 if is_leaf else 0); desc.append(dd)
                weight.append(float(np.round(rng.uniform(0.0, 3.0), 6)) if is_leaf and rng.random() < 0.9 else 0.0)
                if not is_leaf:
                    nxt.append((nid, dd))
                nid += 1
        level = nxt
    return max_children, depth, np.array(parent, np.int32), np.array(leaf, np.uint8), np.stack(desc), np.array(weight)


def test_oracle_vocabulary_arrays_equal_text(tmp_path):
    """The oracle built from arrays == the oracle that parsed the same tree as a DBoW2 text file, at ORBvoc's depth
    (L = 6) and for an irregular tree; both == the independent numpy descent."""
    rng = np.random.default_rng(5)
    desc = rng.integers(0, 256, size=(150, 32), dtype=np.uint8)
    for arrays, levelsup in ((synthetic_vocabulary_arrays(3, 6, 2), 4), (_irregular_tree(9, 20, 4, 600), 2)):
        path = tmp_path / "v.txt"
        n = write_vocabulary_text(path, arrays)
        v1, v2 = orc.Vocabulary(path), orc.Vocabulary.from_arrays(arrays)
        assert v1.info() == v2.info() and v1.info()["nodes"] == n
        r1, r2 = v1.transform(desc, levelsup), v2.transform(desc, levelsup)
        assert r1[0] == r2[0] and all(np.array_equal(a, b) for a, b in zip(r1[1:], r2[1:]))
        ref = _brute_transform(path, desc[:40], levelsup)
        assert [int(x) for x in r1[1][:40]] == [r[0] for r in ref]
        assert [float(x) for x in r1[2][:40]] == [r[1] for r in ref]
        assert [int(x) for x in r1[3][:40]] == [r[2] for r in ref]


@pytest.mark.gpu
def test_gpu_transform_orbvoc_shape_from_arrays():
    """k = 10, L = 6, levelsup = 4 -- the tree and the call of Frame::ComputeBoW (src/Frame.cc:438): 1 111 111 nodes,
    53 MB of records, built from arrays on both sides."""
    import orb_slam2_annotate_amd as amd
    arrays = synthetic_vocabulary_arrays(10, 6, 3)
    vo = orc.Vocabulary.from_arrays(arrays)
    voc = amd.ORBVocabulary()
    assert voc.createFromArrays(arrays)
    assert voc.info() == vo.info() == dict(k=10, L=6, nodes=1111111, words=1000000)
    e = amd.ORBextractor(2000, 1.2, 8, 20, 7)
    kps, desc = e(synth.render_frame(6, 1241, 376))
    rng = np.random.default_rng(1)
    leafs = arrays[4][-1000000:]  # queries that ARE leaf centroids (distance 0 somewhere) and noisy copies of them
    near = leafs[rng.integers(0, len(leafs), 300)] ^ (rng.integers(0, 256, (300, 32), dtype=np.uint8) & rng.integers(0, 256, (300, 32), dtype=np.uint8) & rng.integers(0, 256, (300, 32), dtype=np.uint8))
    desc = np.concatenate([desc, leafs[::5003], near, np.zeros((1, 32), np.uint8), np.full((1, 32), 255, np.uint8)])
    for levelsup in (4, 0, 6, 7):
        word, weight, node = voc.transform_features(desc, levelsup)
        used, w_ref, wt_ref, n_ref = vo.transform(desc, levelsup)
        assert np.array_equal(word, w_ref) and np.array_equal(weight, wt_ref) and np.array_equal(node, n_ref), levelsup
    assert len(np.unique(voc.transform_features(desc, 4)[2])) <= 100


@pytest.mark.gpu
@pytest.mark.parametrize("seed,maxc,depth,levelsup", [(1, 20, 4, 2), (2, 17, 5, 3), (3, 5, 7, 4), (4, 16, 3, 1), (5, 20, 2, 0)])
def test_gpu_transform_irregular_trees(tmp_path, seed, maxc, depth, levelsup):
    """Text-file trees with 1..20 children per node, leaves at every depth, zero-weight leaves, duplicate siblings."""
    import orb_slam2_annotate_amd as amd
    arrays = _irregular_tree(seed, maxc, depth)
    path = tmp_path / "v.txt"
    write_vocabulary_text(path, arrays)
    vo = orc.Vocabulary(path)
    voc = amd.ORBVocabulary()
    assert voc.loadFromTextFile(path)
    assert voc.info() == vo.info()
    rng = np.random.default_rng(seed)
    desc = np.concatenate([rng.integers(0, 256, size=(777, 32), dtype=np.uint8), arrays[4][::7]])
    word, weight, node = voc.transform_features(desc, levelsup)
    used, w_ref, wt_ref, n_ref = vo.transform(desc, levelsup)
    assert np.array_equal(word, w_ref) and np.array_equal(weight, wt_ref) and np.array_equal(node, n_ref)
    assert 0 < used < len(desc)  # zero-weight leaves are "stopped" words (:1161)
    voc2 = amd.ORBVocabulary()
    assert voc2.createFromArrays(arrays)
    w2, wt2, n2 = voc2.transform_features(desc, levelsup)
    assert np.array_equal(w2, w_ref) and np.array_equal(wt2, wt_ref) and np.array_equal(n2, n_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["k10_L2", "orbvoc_k10_L6_stopped_words"])
def test_gpu_bow_batch_device_matches_oracle(tmp_path, shape):
    """ComputeBoW + SearchByBoW(t-1, t) for a device-resident batch."""
    torch = pytest.importorskip("torch")
    import orb_slam2_annotate_amd as amd
    voc = amd.ORBVocabulary()
    if shape == "k10_L2":
        path = tmp_path / "voc.txt"
        write_synthetic_vocabulary(path, k=10, L=2, seed=4)
        vo = orc.Vocabulary(path)
        assert voc.loadFromTextFile(path)
        LS = 1
    else:  # ORBvoc's shape with levelsup = 4 (src/Frame.cc:438); a third of the words "stopped" (weight 0, :1161)
        arrays = list(synthetic_vocabulary_arrays(10, 6, 8))
        arrays[5] = arrays[5] * (np.random.default_rng(2).random(len(arrays[5])) > 0.33)
        vo = orc.Vocabulary.from_arrays(arrays)
        assert voc.createFromArrays(arrays)
        LS = 4
    frames = np.stack(synth.render_sequence(77, 5, 480, 360, step=2.0))
    e = amd.ORBextractor(700, 1.2, 8, 20, 7)
    cap = e.max_keypoints()
    dev = torch.device("cuda", 0)
    B = len(frames)
    d_img = torch.from_numpy(frames).to(dev)
    d_kp = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    d_match = torch.zeros((B - 1, cap), dtype=torch.int32, device=dev)
    d_nm = torch.zeros((B - 1,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    e.extract_batch_device(d_img.data_ptr(), B, 480, 360, 480, 480 * 360, d_kp.data_ptr(), d_desc.data_ptr(), cap,
                           d_n.data_ptr())
    # FeatureVectors on the device vs oracle transform
    d_nodes = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    d_off = torch.zeros((B, cap + 1), dtype=torch.int32, device=dev)
    d_idx = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    d_cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    voc.featvec_batch_device(d_desc.data_ptr(), d_n.data_ptr(), B, cap, d_nodes.data_ptr(), d_off.data_ptr(),
                             d_idx.data_ptr(), d_cnt.data_ptr(), levelsup=LS)
    n = d_n.cpu().numpy()
    kp = d_kp.cpu().numpy()
    desc = d_desc.cpu().numpy()
    fvs = []
    for f in range(B):
        used, word, weight, node = vo.transform(desc[f, :n[f]], LS)
        fv = orc.FeatVec(node, weight > 0)
        assert used == n[f] if shape == "k10_L2" else 0.5 * n[f] < used < 0.8 * n[f]
        c = int(d_cnt[f].item())
        assert c == len(fv.node_ids)
        assert np.array_equal(d_nodes[f, :c].cpu().numpy().astype(np.uint32), fv.node_ids)
        assert np.array_equal(d_off[f, :c + 1].cpu().numpy(), fv.offsets)
        assert np.array_equal(d_idx[f, :used].cpu().numpy().astype(np.uint32), fv.indices)
        fvs.append(fv)
    # consecutive-frame SearchByBoW on the device vs oracle
    voc.bow_match_consecutive_batch_device(B, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap,
                                           d_match.data_ptr(), d_nm.data_ptr(), nnratio=0.7, check_orientation=True,
                                           levelsup=LS)
    total = 0
    for t in range(1, B):
        n1, n2 = n[t - 1], n[t]
        ref_n, ref = orc.search_by_bow(desc[t - 1, :n1], np.ones(n1, np.uint8), kp[t - 1, :n1, 3], fvs[t - 1],
                                       desc[t, :n2], kp[t, :n2, 3], fvs[t], 0.7, True)
        got = d_match[t - 1, :n2].cpu().numpy()
        assert int(d_nm[t - 1].item()) == ref_n
        assert np.array_equal(got, ref)
        total += ref_n
    assert total > (100 if shape == "k10_L2" else 30)


@pytest.mark.gpu
def test_gpu_euroc_composite_extract_then_search_by_bow(tmp_path):
    """BASELINE.json configs[3] as one composite: 752x480 / 1200-feature extraction output of consecutive
    frames -> ComputeBoW -> SearchByBoW(t-1, t), device-resident and asynchronous behind the extractor's
    4 sub-batch streams, twice back to back (no host wait in between), against the oracle end to end."""
    torch = pytest.importorskip("torch")
    import orb_slam2_annotate_amd as amd
    arrays = synthetic_vocabulary_arrays(10, 6, 1)  # ORBvoc's shape; levelsup = 4 as src/Frame.cc:438 (bench.py's euroc step)
    vo = orc.Vocabulary.from_arrays(arrays)
    voc = amd.ORBVocabulary()
    assert voc.createFromArrays(arrays)
    W, H, NF, B = 752, 480, 1200, 6
    seqs = [np.stack(synth.render_sequence(s, B, W, H, step=1.5)) for s in (4100, 4200)]
    e = amd.ORBextractor(NF, 1.2, 8, 20, 7)
    e.set_streams(4)
    cap = e.max_keypoints()
    dev = torch.device("cuda", 0)
    d_imgs = [torch.from_numpy(s).to(dev) for s in seqs]
    d_kp = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    d_match = torch.zeros((2, B - 1, cap), dtype=torch.int32, device=dev)
    d_nm = torch.zeros((2, B - 1), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for k in range(2):
        e.extract_batch_device(d_imgs[k].data_ptr(), B, W, H, W, W * H, d_kp.data_ptr(), d_desc.data_ptr(), cap,
                               d_n.data_ptr(), wait=False)
        voc.bow_match_consecutive_batch_device(B, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap,
                                               d_match[k].data_ptr(), d_nm[k].data_ptr(), nnratio=0.7,
                                               check_orientation=True, levelsup=4, extractor=e)
    e.synchronize()
    o = orc.Oracle(NF, 1.2, 8, 20, 7)
    for k in range(2):
        ref = [o.extract(f) for f in seqs[k]]
        fvs = [orc.FeatVec(vo.transform(d, 4)[3]) for _, d in ref]
        total = 0
        for t in range(1, B):
            (k1, de1), (k2, de2) = ref[t - 1], ref[t]
            rn, rm = orc.search_by_bow(de1, np.ones(len(k1), np.uint8), k1["angle"], fvs[t - 1], de2, k2["angle"],
                                       fvs[t], 0.7, True)
            assert int(d_nm[k, t - 1].item()) == rn, (k, t)
            assert np.array_equal(d_match[k, t - 1, :len(k2)].cpu().numpy(), rm), (k, t)
            total += rn
        assert total > 200
    # the last extract call left sequence 1 in the output buffers: bit-exact too
    n = d_n.cpu().numpy()
    for f in range(B):
        kr, dr = o.extract(seqs[1][f])
        assert n[f] == len(kr)
        assert np.array_equal(d_desc[f, :n[f]].cpu().numpy(), dr)


def _short_line_vocabulary(tmp_path):
    """A text vocabulary with two TRUNCATED leaf lines (no weight token; no weight and two descriptor bytes missing) and
    the arrays the reference's per-line `getline` + `>>` parsing turns them into: the missing fields read as 0 and the
    FOLLOWING nodes keep their own tokens (TemplatedVocabulary.h:1374-1417)."""
    k, L, parent, leaf, desc, weight = synthetic_vocabulary_arrays(4, 3, seed=9)
    desc, weight = desc.copy(), np.round(weight, 6)
    path = tmp_path / "short.txt"
    write_vocabulary_text(path, (k, L, parent, leaf, desc, weight))
    lines = open(path).read().split("\n")
    leaves = np.flatnonzero(leaf)
    a, b = int(leaves[3]), int(leaves[17])
    lines[1 + a] = " ".join(lines[1 + a].split()[:-1])   # weight token gone
    lines[1 + b] = " ".join(lines[1 + b].split()[:-3])   # weight and the last two descriptor bytes gone
    open(path, "w").write("\n".join(lines))
    weight[a] = 0.0
    weight[b] = 0.0
    desc[b, 30:] = 0
    return path, (k, L, parent, leaf, desc, weight)


def test_oracle_vocabulary_short_lines_do_not_shift_later_nodes(tmp_path):
    path, arrays = _short_line_vocabulary(tmp_path)
    vt, va = orc.Vocabulary(path), orc.Vocabulary.from_arrays(arrays)
    assert vt.info() == va.info()
    desc = np.random.default_rng(4).integers(0, 256, size=(500, 32), dtype=np.uint8)
    desc[:64] = arrays[4][np.flatnonzero(arrays[3])[:64]]  # the leaves' own descriptors: every early word is hit
    for levelsup in (0, 1):
        rt, ra = vt.transform(desc, levelsup), va.transform(desc, levelsup)
        assert rt[0] == ra[0] and all(np.array_equal(x, y) for x, y in zip(rt[1:], ra[1:]))


@pytest.mark.gpu
def test_gpu_vocabulary_short_lines_do_not_shift_later_nodes(tmp_path):
    """round-3 ADVICE: strtol / strtod skip '\\n', so a short node line used to swallow the next node's tokens and shift
    every later node.  Each line is now parsed on its own, like the reference's stringstream per getline."""
    import orb_slam2_annotate_amd as amd
    path, arrays = _short_line_vocabulary(tmp_path)
    va = orc.Vocabulary.from_arrays(arrays)
    voc = amd.ORBVocabulary()
    assert voc.loadFromTextFile(path)
    assert voc.info() == va.info()
    desc = np.random.default_rng(4).integers(0, 256, size=(500, 32), dtype=np.uint8)
    desc[:64] = arrays[4][np.flatnonzero(arrays[3])[:64]]
    for levelsup in (0, 1):
        word, weight, node = voc.transform_features(desc, levelsup)
        _, w_ref, wt_ref, n_ref = va.transform(desc, levelsup)
        assert np.array_equal(word, w_ref) and np.array_equal(weight, wt_ref) and np.array_equal(node, n_ref)
