"""The REAL drop-in classes executed: include/ORBextractor.h, include/ORBmatcher.h + src/ORBmatcher_orbfe.cc are
compiled (g++ -std=c++11, like the reference's CMakeLists.txt:13-25) against the functional test doubles of
tests/cpp/doubles/ (cv::Mat / Frame / KeyFrame / MapPoint members the wrappers touch -- authored here, not a reference
build, no claim about OpenCV) into tests/cpp/test_dropin.cpp, which drives ALL 12 ORBmatcher methods and
ORBextractor::operator() on the GPU with synthetic poses.  The un-flattened MapPoint* results (as map-point ids) are
compared with the CPU oracle fed by an INDEPENDENT numpy-float32 restatement of each method's prologue (pose
arithmetic, gates, PredictScale: src/ORBmatcher.cc:1494-1541, 1645-1700, 342-400, 761-769, 960-1020, 1120-1180,
1260-1340): the flatten / prologue / un-flatten code of every method runs and matches."""
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as orc
from orb_slam2_annotate_amd import synth

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
f32, f64 = np.float32, np.float64
W, H = 640.0, 480.0
FX, FY, CX, CY, BF = f32(520.0), f32(518.0), f32(320.5), f32(240.25), f32(40.0)
MB = f32(BF / FX)
SF = np.array(orc.Oracle(600, 1.2, 8, 20, 7).scale_factors(), f32)
SIG2 = np.array(orc.Oracle(600, 1.2, 8, 20, 7).level_sigma2(), f32)
ISIG2 = np.array(orc.Oracle(600, 1.2, 8, 20, 7).inv_level_sigma2(), f32)
LOGSF = f32(np.log(1.2))  # mfLogScaleFactor = log(mfScaleFactor) (src/Frame.cc:71)


# ---------------- named-array files ----------------
def _write(path, arrays):
    with open(path, "wb") as f:
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            kind = {np.dtype(np.uint8): 0, np.dtype(np.int32): 1, np.dtype(np.float32): 2}[a.dtype]
            f.write(struct.pack("<I", len(name)) + name.encode() + struct.pack("<BI", kind, a.size) + a.tobytes())


def _read(path):
    out, b, p = {}, open(path, "rb").read(), 0
    while p < len(b):
        nl, = struct.unpack_from("<I", b, p); p += 4
        name = b[p:p + nl].decode(); p += nl
        kind, cnt = struct.unpack_from("<BI", b, p); p += 5
        dt = [np.uint8, np.int32, np.float32][kind]
        n = cnt * np.dtype(dt).itemsize
        out[name] = np.frombuffer(b[p:p + n], dtype=dt).copy(); p += n
    return out


# ---------------- the doubles' arithmetic, restated in numpy ----------------
def mm(A, B):
    """cv::Mat product of the test double: accumulate in double, round once to float"""
    return (A.astype(f64) @ B.astype(f64)).astype(f32)


def camera_center(T):  # -(R^T) * t
    R, t = T[:3, :3], T[:3, 3:4]
    return mm((R.T.astype(f64) * -1.0).astype(f32), t)


def predict_scale(max_raw, dist):
    ratio = (max_raw / dist).astype(f32)  # float division
    n = np.ceil(np.log(ratio.astype(f64)) / f64(LOGSF)).astype(np.int64)
    return np.clip(n, 0, len(SF) - 1).astype(np.int32)


def rot(ax, ay, az):
    cx_, sx = np.cos(ax), np.sin(ax); cy_, sy = np.cos(ay), np.sin(ay); cz, sz = np.cos(az), np.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx_, -sx], [0, sx, cx_]]); Ry = np.array([[cy_, 0, sy], [0, 1, 0], [-sy, 0, cy_]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def pose(rng, scale=1.0):
    T = np.eye(4)
    T[:3, :3] = rot(*(rng.normal(0, 0.02 * scale, 3)))
    T[:3, 3] = rng.normal(0, 0.15 * scale, 3)
    return T.astype(f32)


class World:
    def __init__(self, seed):
        rng = self.rng = np.random.default_rng(seed)
        M = self.M = 1600
        z = rng.uniform(4, 12, M)
        self.X = np.stack([(rng.uniform(20, W - 20, M) - CX) / FX * z, (rng.uniform(20, H - 20, M) - CY) / FY * z, z], 1).astype(f32)
        self.desc = rng.integers(0, 256, (M, 32), dtype=np.uint8)
        self.oct = rng.integers(0, 8, M)
        # duplicated structure (what Fuse exists for): the last 150 points are near-copies of the first 150
        self.X[-150:] = self.X[:150] + rng.normal(0, 2e-3, (150, 3)).astype(f32)
        self.desc[-150:] = self.desc[:150] ^ (rng.integers(0, 256, (150, 32), dtype=np.uint8) & rng.integers(0, 256, (150, 32), dtype=np.uint8) &
                                               rng.integers(0, 256, (150, 32), dtype=np.uint8) & rng.integers(0, 256, (150, 32), dtype=np.uint8) &
                                               rng.integers(0, 256, (150, 32), dtype=np.uint8))
        self.oct[-150:] = self.oct[:150]
        d0 = np.linalg.norm(self.X, axis=1)
        self.max_raw = (d0 * 1.2 ** self.oct * rng.uniform(0.93, 0.99, M)).astype(f32)
        self.min_inv = (0.8 * self.max_raw / 1.2 ** 7).astype(f32)
        self.max_inv = (1.2 * self.max_raw).astype(f32)
        n = self.X / d0[:, None] + rng.normal(0, 0.15, (M, 3))  # mean viewing direction: camera -> point (src/MapPoint.cc:380-396)
        self.normal = (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(f32)
        self.nobs = rng.integers(0, 6, M).astype(np.int32)
        self.bad = (rng.random(M) < 0.04).astype(np.uint8)
        self.cams = {}
        for c, sc in (("A", 1.0), ("B", 1.0), ("C", 0.6), ("D", 0.6)):
            self.cams[c] = self.make_cam(pose(rng, sc))

    def project(self, T, idx):
        Xc = (T[:3, :3].astype(f64) @ self.X[idx].T.astype(f64)).T + T[:3, 3].astype(f64)
        return FX * Xc[:, 0] / Xc[:, 2] + CX, FY * Xc[:, 1] / Xc[:, 2] + CY, Xc[:, 2]

    def make_cam(self, T):
        rng = self.rng
        idx = rng.permutation(self.M)[:900]
        u, v, z = self.project(T, idx)
        ok = (u > 8) & (u < W - 8) & (v > 8) & (v < H - 8)
        idx, u, v, z = idx[ok], u[ok], v[ok], z[ok]
        n = len(idx)
        kp = np.zeros((n, 7), f32)
        kp[:, 0] = (u + rng.normal(0, 0.6, n)).astype(f32)
        kp[:, 1] = (v + rng.normal(0, 0.6, n)).astype(f32)
        octv = np.clip(self.oct[idx] + rng.choice([-1, 0, 0, 0, 1], n), 0, 7).astype(np.int32)
        kp[:, 2] = 31.0 * SF[octv]
        kp[:, 3] = rng.uniform(0, 360, n).astype(f32)
        kp[:, 4] = rng.uniform(20, 90, n).astype(f32)
        kp.view(np.int32)[:, 5] = octv
        kp.view(np.int32)[:, 6] = -1
        noise = rng.integers(0, 256, (n, 32), dtype=np.uint8) & rng.integers(0, 256, (n, 32), dtype=np.uint8) & \
            rng.integers(0, 256, (n, 32), dtype=np.uint8) & rng.integers(0, 256, (n, 32), dtype=np.uint8)
        desc = self.desc[idx] ^ noise
        ur = np.where(rng.random(n) < 0.6, kp[:, 0] - BF / z.astype(f32), -1.0).astype(f32)
        mp = np.where(rng.random(n) < 0.6, idx, -1).astype(np.int32)
        node = ((desc[:, 0].astype(np.int32) >> 3) * 3 + 1).astype(np.int32)  # 32 "vocabulary nodes"
        outlier = (rng.random(n) < 0.05).astype(np.uint8)
        return dict(T=T, kp=kp, ur=ur, desc=desc, mp=mp, node=node, outlier=outlier, src=idx)

    def arrays(self):
        a = {"K": np.array([FX, FY, CX, CY, BF, MB], f32), "bounds": np.array([0, W, 0, H], f32), "sf": SF, "sigma2": SIG2,
             "invsigma2": ISIG2, "logsf": np.array([LOGSF], f32), "mp_pos": self.X, "mp_normal": self.normal, "mp_desc": self.desc,
             "mp_maxraw": self.max_raw, "mp_min": self.min_inv, "mp_max": self.max_inv, "mp_nobs": self.nobs, "mp_bad": self.bad}
        for c, d in self.cams.items():
            a.update({f"{c}_kp": d["kp"], f"{c}_ur": d["ur"], f"{c}_desc": d["desc"], f"{c}_mp": d["mp"], f"{c}_node": d["node"],
                      f"{c}_Tcw": d["T"], f"{c}_outlier": d["outlier"]})
        return a

    # oracle views
    def oframe(self, c, stereo=True):
        d = self.cams[c]
        k = d["kp"]
        return orc.Frame(k[:, 0], k[:, 1], k.view(np.int32)[:, 5], d["desc"], (0.0, float(W), 0.0, float(H)), angle=k[:, 3],
                         u_right=d["ur"] if stereo else None)

    def octave(self, c):
        return self.cams[c]["kp"].view(np.int32)[:, 5].copy()


def cam_project(R, t, X):
    """x3Dc = Rcw * x3Dw + tcw per point; iz = (float)(1.0 / z)"""
    Xc = (mm(R, X.T) + t).T  # [n,3] float32: double-accumulated product rounded, then a float add
    iz = (1.0 / Xc[:, 2].astype(f64)).astype(f32)
    return Xc, iz


def in_image(u, v):  # KeyFrame::IsInImage with integer bounds 0, W, 0, H
    return (u >= 0) & (u < W) & (v >= 0) & (v < H)


def project_into_keyframe(w, T_R, T_t, Ow, sel, bf):
    """the gates shared by SearchByProjection(KF,Scw) / Fuse / Fuse(Scw): src/ORBmatcher.cc:366-400, 975-1008"""
    X = w.X[sel]
    Xc = (mm(T_R, X.T) + T_t).T
    zpos = ~(Xc[:, 2] < 0)
    invz = (1.0 / Xc[:, 2].astype(f64)).astype(f32)
    x, y = Xc[:, 0] * invz, Xc[:, 1] * invz
    u, v = FX * x + CX, FY * y + CY
    ur = u - f32(bf) * invz
    PO = X - Ow.T
    dist = np.sqrt((PO.astype(f64) ** 2).sum(1)).astype(f32)
    rng_ok = ~((dist < w.min_inv[sel]) | (dist > w.max_inv[sel]))
    cone = ~((PO.astype(f64) * w.normal[sel].astype(f64)).sum(1) < 0.5 * dist.astype(f64))
    ok = zpos & in_image(u, v) & rng_ok & cone
    return ok, u.astype(f32), v.astype(f32), ur.astype(f32), predict_scale(w.max_raw[sel], dist)


def split_sim3(S):
    sR = S[:3, :3]
    scw = f32(np.sqrt((sR[0].astype(f64) ** 2).sum()))
    R = (sR.astype(f64) / f64(scw)).astype(f32)
    t = (S[:3, 3:4].astype(f64) / f64(scw)).astype(f32)
    Ow = mm((R.T.astype(f64) * -1.0).astype(f32), t)
    return R, t, Ow


def _ids(mp, idx):
    """MapPoint ids behind keypoint indices (-1 stays -1)"""
    return np.where(idx >= 0, mp[np.clip(idx, 0, len(mp) - 1)], -1).astype(np.int32)


def test_dropin_classes_run_and_match_the_oracle(tmp_path):
    w = World(2024)
    rng = np.random.default_rng(7)
    A, B, Cc, D = (w.cams[c] for c in "ABCD")
    arrays = w.arrays()
    # --- extractor input
    img = synth.render_frame(12, 320, 240)
    arrays["image"] = img.reshape(-1)
    arrays["image_wh"] = np.array([320, 240], np.int32)
    # --- (1) local map points projected into frame C (fields Frame::isInFrustum would have stored)
    sel1 = rng.permutation(w.M)[:1100].astype(np.int32)
    u1, v1, z1 = w.project(Cc["T"], sel1)
    inview1 = ((u1 > 0) & (u1 < W) & (v1 > 0) & (v1 < H) & (rng.random(len(sel1)) < 0.9)).astype(np.uint8)
    arrays.update({"m1_points": sel1, "m1_inview": inview1, "m1_level": np.clip(w.oct[sel1] + rng.integers(0, 2, len(sel1)), 0, 7).astype(np.int32),
                   "m1_viewcos": rng.uniform(0.99, 1.0, len(sel1)).astype(f32), "m1_px": u1.astype(f32), "m1_py": v1.astype(f32),
                   "m1_pxr": (u1 - BF / z1).astype(f32), "m1_th": np.array([3.0], f32), "m2_th": np.array([7.0], f32),
                   "m3_th": np.array([10.0], f32)})
    arrays["m3_found"] = rng.permutation(w.M)[:100].astype(np.int32)
    S = np.eye(4)
    S[:3, :3] = 1.07 * rot(0.01, -0.015, 0.02) @ B["T"][:3, :3].astype(f64)
    S[:3, 3] = 1.07 * B["T"][:3, 3] + [0.05, -0.02, 0.03]
    arrays["m4_Scw"] = S.astype(f32)
    sel4 = rng.permutation(w.M)[:1000].astype(np.int32)
    arrays["m4_points"] = sel4
    pre4 = np.where(rng.random(len(B["mp"])) < 0.2, rng.integers(0, w.M, len(B["mp"])), -1).astype(np.int32)
    arrays["m4_matched"] = pre4
    TA, TB = A["T"].astype(f64), B["T"].astype(f64)
    T12 = TA @ np.linalg.inv(TB)  # maps camera-2 coordinates to camera-1 coordinates
    tx = np.array([[0, -T12[2, 3], T12[1, 3]], [T12[2, 3], 0, -T12[0, 3]], [-T12[1, 3], T12[0, 3], 0]])
    Kinv = np.linalg.inv(np.array([[FX, 0, CX], [0, FY, CY], [0, 0, 1]], f64))
    arrays["m8_F12"] = (Kinv.T @ tx @ T12[:3, :3] @ Kinv).astype(f32)
    s12 = f32(1.03)
    arrays.update({"m9_s12": np.array([s12], f32), "m9_R12": (T12[:3, :3] @ rot(0.004, -0.003, 0.002)).astype(f32),
                   "m9_t12": (T12[:3, 3] / 1.03 + [0.01, 0.0, -0.01]).astype(f32)})
    pre9 = np.where(rng.random(len(A["mp"])) < 0.15, rng.integers(0, w.M, len(A["mp"])), -1).astype(np.int32)
    arrays["m9_pre"] = pre9
    sel10 = rng.permutation(w.M)[:1200].astype(np.int32)
    sel10[::97] = -1  # NULL entries (:955)
    arrays["m10_points"] = sel10
    arrays["m11_points"] = rng.permutation(w.M)[:1000].astype(np.int32)
    _write(tmp_path / "in.bin", arrays)

    exe = tmp_path / "test_dropin"
    lib = ROOT / "orb_slam2_annotate_amd"
    subprocess.run(["g++", "-O1", "-std=c++11", "-ffp-contract=off", "-pthread", f"-I{ROOT / 'tests/cpp/doubles'}", f"-I{ROOT / 'include'}",
                    str(ROOT / "tests/cpp/test_dropin.cpp"), str(ROOT / "src/ORBmatcher_orbfe.cc"), "-o", str(exe), f"-L{lib}", "-lorbfe",
                    f"-Wl,-rpath,{lib}"], check=True)
    subprocess.run([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], check=True, timeout=120)
    out = _read(tmp_path / "out.bin")

    # ---- ORBextractor::operator() ----
    o = orc.Oracle(600, 1.2, 8, 20, 7)
    kr, dr, pr = o.extract(img, want_pyramid=True)
    assert out["ex_kp"].tobytes() == kr.tobytes() and out["ex_desc"].tobytes() == dr.tobytes() and len(kr) > 100
    assert out["ex_pyr"].tobytes() == pr.tobytes() and out["ex_levels"][0] == 8 and out["ex_empty_untouched"][0] == 3
    assert out["dd"][0] == orc.descriptor_distance(w.desc[0], w.desc[1])

    bad, nobs = w.bad.astype(bool), w.nobs
    Co, Do, Ao, Bo = w.oframe("C"), w.oframe("D"), w.oframe("A"), w.oframe("B")

    def has_good(mp):  # "pMP && !pMP->isBad()"
        return ((mp >= 0) & ~bad[np.clip(mp, 0, w.M - 1)]).astype(np.uint8)

    # ---- (1) SearchByProjection(Frame, MapPoints) ----
    inv = inview1 & ~w.bad[sel1]
    blocked = ((Cc["mp"] >= 0) & (nobs[np.clip(Cc["mp"], 0, w.M - 1)] > 0)).astype(np.uint8)
    n, m = orc.search_by_projection_mappoints(Co, SF, blocked, inv, arrays["m1_level"], arrays["m1_viewcos"], arrays["m1_px"], arrays["m1_py"],
                                              arrays["m1_pxr"], w.desc[sel1], (nobs[sel1] > 0).astype(np.uint8), 3.0, 0.8)
    exp = np.where(m >= 0, sel1[np.clip(m, 0, len(sel1) - 1)], Cc["mp"]).astype(np.int32)
    assert out["m1_n"][0] == n > 50 and np.array_equal(out["m1_out"], exp)

    # ---- (2) SearchByProjection(CurrentFrame, LastFrame) ----
    Rcw, tcw = Cc["T"][:3, :3], Cc["T"][:3, 3:4]
    twc = camera_center(Cc["T"])
    tlc = mm(D["T"][:3, :3], twc) + D["T"][:3, 3:4]
    for mono, tag in ((False, "m2"), (True, "m2m")):
        fwd = bool(tlc[2, 0] > MB) and not mono
        bwd = bool(-tlc[2, 0] > MB) and not mono
        mpD = D["mp"]
        have = (mpD >= 0) & (D["outlier"] == 0)
        Xc, iz = cam_project(Rcw, tcw, w.X[np.clip(mpD, 0, w.M - 1)])
        uu = FX * Xc[:, 0] * iz + CX
        vv = FY * Xc[:, 1] * iz + CY
        valid = have & ~(iz < 0) & ~((uu < 0) | (uu > W)) & ~((vv < 0) | (vv > H))
        obs = (nobs[np.clip(mpD, 0, w.M - 1)] > 0).astype(np.uint8)
        n, m = orc.search_by_projection_lastframe(Co, SF, float(BF), valid.astype(np.uint8), uu, vv, iz, w.octave("D"), D["kp"][:, 3],
                                                  w.desc[np.clip(mpD, 0, w.M - 1)], obs, 1 if fwd else (2 if bwd else 0), 7.0, True, blocked)
        exp = np.where(m >= 0, mpD[np.clip(m, 0, len(mpD) - 1)], Cc["mp"]).astype(np.int32)
        assert out[tag + "_n"][0] == n > 30 and np.array_equal(out[tag + "_out"], exp), tag

    # ---- (3) SearchByProjection(CurrentFrame, KeyFrame, sAlreadyFound) ----
    mpA = A["mp"]
    found = np.zeros(w.M, bool); found[arrays["m3_found"]] = True
    ia = np.clip(mpA, 0, w.M - 1)
    cand = (mpA >= 0) & ~bad[ia] & ~found[ia]
    Xc, iz = cam_project(Rcw, tcw, w.X[ia])
    uu, vv = FX * Xc[:, 0] * iz + CX, FY * Xc[:, 1] * iz + CY
    PO = w.X[ia] - twc.T
    dist = np.sqrt((PO.astype(f64) ** 2).sum(1)).astype(f32)
    valid = cand & ~((uu < 0) | (uu > W)) & ~((vv < 0) | (vv > H)) & ~((dist < w.min_inv[ia]) | (dist > w.max_inv[ia]))
    n, m = orc.search_by_projection_reloc(Co, SF, valid.astype(np.uint8), uu, vv, predict_scale(w.max_raw[ia], dist), A["kp"][:, 3], w.desc[ia],
                                          (Cc["mp"] >= 0).astype(np.uint8), 10.0, 100, True)
    exp = np.where(m >= 0, mpA[np.clip(m, 0, len(mpA) - 1)], Cc["mp"]).astype(np.int32)
    assert out["m3_n"][0] == n > 20 and np.array_equal(out["m3_out"], exp)

    # ---- (4) SearchByProjection(KeyFrame, Scw, vpPoints, vpMatched) ----
    R4, t4, Ow4 = split_sim3(arrays["m4_Scw"])
    already = np.zeros(w.M, bool); already[pre4[pre4 >= 0]] = True
    ok, u, v, ur, lev = project_into_keyframe(w, R4, t4, Ow4, sel4, 0.0)
    valid = ok & ~bad[sel4] & ~already[sel4]
    n, m = orc.search_by_projection_sim3(w.oframe("B"), SF, valid.astype(np.uint8), u, v, lev, w.desc[sel4], (pre4 >= 0).astype(np.uint8), 10.0)
    exp = np.where(m >= 0, sel4[np.clip(m, 0, len(sel4) - 1)], pre4).astype(np.int32)
    assert out["m4_n"][0] == n > 20 and np.array_equal(out["m4_out"], exp)

    # ---- (5), (6) SearchByBoW ----
    fvA, fvB, fvC = orc.FeatVec(A["node"]), orc.FeatVec(B["node"]), orc.FeatVec(Cc["node"])
    n, m = orc.search_by_bow(A["desc"], has_good(mpA), A["kp"][:, 3], fvA, Cc["desc"], Cc["kp"][:, 3], fvC, 0.7, True)
    assert out["m5_n"][0] == n > 10 and np.array_equal(out["m5_out"], _ids(mpA, m))
    n, m = orc.search_by_bow_kf(A["desc"], has_good(mpA), A["kp"][:, 3], fvA, B["desc"], has_good(B["mp"]), B["kp"][:, 3], fvB, 0.75, True)
    assert out["m6_n"][0] == n > 5 and np.array_equal(out["m6_out"], _ids(B["mp"], m))

    # ---- (7) SearchForInitialization ----
    n, m, prev = orc.search_for_initialization(w.oframe("C"), w.oframe("D"), Cc["kp"][:, :2].copy(), 100, 0.9, True)
    assert out["m7_n"][0] == n > 10 and np.array_equal(out["m7_out"], m) and np.array_equal(out["m7_prev"].reshape(-1, 2), prev)

    # ---- (8) SearchForTriangulation ----
    Cw = camera_center(A["T"])
    C2 = mm(B["T"][:3, :3], Cw) + B["T"][:3, 3:4]
    invz = f32(1.0) / C2[2, 0]
    ex, ey = FX * C2[0, 0] * invz + CX, FY * C2[1, 0] * invz + CY
    for only, tag in ((False, "m8"), (True, "m8s")):
        n, m = orc.search_for_triangulation(A["desc"], (mpA >= 0).astype(np.uint8), A["kp"][:, 0], A["kp"][:, 1], A["kp"][:, 3], A["ur"] >= 0, fvA,
                                            B["desc"], (B["mp"] >= 0).astype(np.uint8), B["kp"][:, 0], B["kp"][:, 1], B["kp"][:, 3], w.octave("B"),
                                            B["ur"] >= 0, fvB, arrays["m8_F12"].reshape(-1), float(ex), float(ey), SF, SIG2, only, False)
        idx = np.nonzero(m >= 0)[0]
        assert out[tag + "_n"][0] == n and np.array_equal(out[tag + "_out"], np.stack([idx, m[idx]], 1).astype(np.int32).reshape(-1)), tag
    assert out["m8_n"][0] > 5

    # ---- (9) SearchBySim3 ----
    R12, t12 = arrays["m9_R12"].reshape(3, 3), arrays["m9_t12"].reshape(3, 1)
    sR12 = (R12.astype(f64) * f64(s12)).astype(f32)
    sR21 = (R12.T.astype(f64) * (1.0 / f64(s12))).astype(f32)
    t21 = mm((sR21.astype(f64) * -1.0).astype(f32), t12)
    mpB = B["mp"]
    done1 = pre9 >= 0
    done2 = np.zeros(len(mpB), bool)
    kfB_index = {int(p): i for i, p in enumerate(mpB) if p >= 0}  # GetIndexInKeyFrame(pKF2)
    for p in pre9[done1]:
        if int(p) in kfB_index:
            done2[kfB_index[int(p)]] = True

    def direction(mp, done, Tw, sR, t):
        i = np.clip(mp, 0, w.M - 1)
        pa = (mm(Tw[:3, :3], w.X[i].T) + Tw[:3, 3:4])
        pb = (mm(sR, pa) + t).T
        ok = (mp >= 0) & ~done & ~bad[i] & ~(pb[:, 2] < 0)
        invz_ = (1.0 / pb[:, 2].astype(f64)).astype(f32)
        u_, v_ = FX * (pb[:, 0] * invz_) + CX, FY * (pb[:, 1] * invz_) + CY
        d = np.sqrt((pb.astype(f64) ** 2).sum(1)).astype(f32)
        ok &= in_image(u_, v_) & ~((d < w.min_inv[i]) | (d > w.max_inv[i]))
        return ok.astype(np.uint8), u_.astype(f32), v_.astype(f32), predict_scale(w.max_raw[i], d), w.desc[i]

    v1_, u1_, vv1_, l1_, d1_ = direction(mpA, done1, A["T"], sR21, t21)
    v2_, u2_, vv2_, l2_, d2_ = direction(mpB, done2, B["T"], sR12, t12)
    n, m = orc.search_by_sim3(w.oframe("A"), w.oframe("B"), SF, SF, v1_, u1_, vv1_, l1_, d1_, v2_, u2_, vv2_, l2_, d2_, 7.5)
    exp = np.where(m >= 0, mpB[np.clip(m, 0, len(mpB) - 1)], pre9).astype(np.int32)
    assert out["m9_n"][0] == n > 10 and np.array_equal(out["m9_out"], exp)

    # ---- (10) Fuse(KeyFrame, vpMapPoints): search + the replayed Replace / AddObservation bookkeeping ----
    RB, tB = B["T"][:3, :3], B["T"][:3, 3:4]
    OwB = camera_center(B["T"])
    s10 = np.clip(sel10, 0, w.M - 1)
    in_kf = np.zeros(w.M, bool); in_kf[mpB[mpB >= 0]] = True
    ok, u, v, ur, lev = project_into_keyframe(w, RB, tB, OwB, s10, float(BF))
    valid = ok & (sel10 >= 0) & ~bad[s10] & ~in_kf[s10]
    best = orc.fuse_search(Bo, SF, ISIG2, valid.astype(np.uint8), u, v, ur, lev, w.desc[s10], 3.0, True)
    kf = mpB.copy(); b_ = bad.copy(); rep = np.full(w.M, -1, np.int32); no = nobs.copy(); inkf = in_kf.copy()
    fused = 0
    for i in np.nonzero(best >= 0)[0]:
        p = int(sel10[i])
        if b_[p] or inkf[p]:
            continue
        q = int(kf[best[i]])
        if q >= 0:
            if not b_[q]:
                if no[q] > no[p]:
                    b_[p] = True; rep[p] = q
                else:
                    b_[q] = True; rep[q] = p
        else:
            no[p] += 1; inkf[p] = True; kf[best[i]] = p
        fused += 1
    assert out["m10_n"][0] == fused > 20 and np.array_equal(out["m10_kf"], kf) and np.array_equal(out["m10_replaced"], rep)
    assert np.array_equal(out["m10_bad"].astype(bool), b_) and np.array_equal(out["m10_nobs"], no) and (rep >= 0).sum() > 3

    # ---- (11) Fuse(KeyFrame, Scw, vpPoints, th, vpReplacePoint) ----
    sel11 = arrays["m11_points"]
    in_set = np.zeros(w.M, bool); in_set[mpB[(mpB >= 0) & ~bad[np.clip(mpB, 0, w.M - 1)]]] = True  # pKF->GetMapPoints()
    ok, u, v, ur, lev = project_into_keyframe(w, R4, t4, Ow4, sel11, 0.0)
    valid = ok & ~bad[sel11] & ~in_set[sel11]
    best = orc.fuse_search(Bo, SF, ISIG2, valid.astype(np.uint8), u, v, ur, lev, w.desc[sel11], 4.0, False)
    kf = mpB.copy(); repl = np.full(len(sel11), -1, np.int32); fused = 0
    for i in np.nonzero(best >= 0)[0]:
        q = int(kf[best[i]])
        if q >= 0:
            if not bad[q]:
                repl[i] = q
        else:
            kf[best[i]] = sel11[i]
        fused += 1
    assert out["m11_n"][0] == fused > 20 and np.array_equal(out["m11_rep"], repl) and np.array_equal(out["m11_kf"], kf)
