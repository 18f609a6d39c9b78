"""cv::initUndistortRectifyMap (Examples/Stereo/stereo_euroc.cc:97-98): known answers of the oracle's restatement that follow
from the published algorithm without OpenCV, and the GPU kernel against the oracle bit for bit.  Calibration values are the
reference's own (Examples/Stereo/EuRoC.yaml: LEFT / RIGHT .K .D .R .P)."""
import numpy as np
import pytest

import oracle_lib as O

EUROC = {
    "LEFT": dict(K=[458.654, 0.0, 367.215, 0.0, 457.296, 248.375, 0.0, 0.0, 1.0],
                 D=[-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0],
                 R=[0.999966347530033, -0.001422739138722922, 0.008079580483432283, 0.001365741834644127, 0.9999741760894847,
                    0.007055629199258132, -0.008089410156878961, -0.007044357138835809, 0.9999424675829176],
                 P=[435.2046959714599, 0, 367.4517211914062, 0, 0, 435.2046959714599, 252.2008514404297, 0, 0, 0, 1, 0]),
    "RIGHT": dict(K=[457.587, 0.0, 379.999, 0.0, 456.134, 255.238, 0.0, 0.0, 1],
                  D=[-0.28368365, 0.07451284, -0.00010473, -3.555907e-05, 0.0],
                  R=[0.9999633526194376, -0.003625811871560086, 0.007755443660172947, 0.003680398547259526, 0.9999684752771629,
                     -0.007035845251224894, -0.007729688520722713, 0.007064130529506649, 0.999945173484644],
                  P=[435.2046959714599, 0, 367.4517211914062, -47.90639384423901, 0, 435.2046959714599, 252.2008514404297, 0, 0, 0, 1, 0]),
}
SIZE = (752, 480)


def _numpy_model(K, D, R, P, size):
    """the same camera model evaluated directly (x = iR [j i 1]^T, no running sums): agrees with the oracle to float rounding"""
    K = np.asarray(K, np.float64).reshape(3, 3)
    P = np.asarray(P, np.float64).reshape(3, -1)[:, :3]
    R = np.asarray(R, np.float64).reshape(3, 3)
    k1, k2, p1, p2, k3 = (list(D) + [0] * 5)[:5]
    iR = np.linalg.inv(P @ R)
    j, i = np.meshgrid(np.arange(size[0], dtype=np.float64), np.arange(size[1], dtype=np.float64))
    v = np.stack([j, i, np.ones_like(j)], -1) @ iR.T
    x, y = v[..., 0] / v[..., 2], v[..., 1] / v[..., 2]
    r2 = x * x + y * y
    kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * kr + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * kr + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]


def test_no_distortion_identity_rotation_same_camera_is_the_identity_map():
    K = EUROC["LEFT"]["K"]
    mx, my = O.init_undistort_rectify_map(K, None, None, None, (64, 48))
    j, i = np.meshgrid(np.arange(64, dtype=np.float32), np.arange(48, dtype=np.float32))
    assert np.abs(mx - j).max() < 2e-4 and np.abs(my - i).max() < 2e-4
    # a new camera matrix with twice the focal length samples half the field of view around the principal point
    P = [2 * K[0], 0, K[2], 0, 2 * K[4], K[5], 0, 0, 1]
    mx, my = O.init_undistort_rectify_map(K, [0, 0, 0, 0], np.eye(3), P, (64, 48))
    assert np.abs(mx - (K[2] + (j - K[2]) / 2)).max() < 2e-4 and np.abs(my - (K[5] + (i - K[5]) / 2)).max() < 2e-4


@pytest.mark.parametrize("side", ["LEFT", "RIGHT"])
def test_euroc_maps_follow_the_camera_model(side):
    c = EUROC[side]
    mx, my = O.init_undistort_rectify_map(c["K"], c["D"], c["R"], c["P"], SIZE)
    ux, uy = _numpy_model(c["K"], c["D"], c["R"], c["P"], SIZE)
    assert mx.dtype == np.float32 and mx.shape == (480, 752)
    assert np.abs(mx - ux).max() < 1e-3 and np.abs(my - uy).max() < 1e-3   # (float32 maps of values up to ~800: ulp 6e-5)
    # the rectified principal point (P: 367.45, 252.20) looks at the raw one up to the small rectifying rotation; the map is
    # monotonic along rows and columns (no fold-over inside the image)
    assert abs(mx[252, 367] - c["K"][2]) < 15 and abs(my[252, 367] - c["K"][5]) < 15
    assert (np.diff(mx, axis=1) > 0).all() and (np.diff(my, axis=0) > 0).all()


def test_argument_checks():
    K = EUROC["LEFT"]["K"]
    with pytest.raises(ValueError):
        O.init_undistort_rectify_map(K, [0.1, 0.2, 0.3], None, None, (8, 8))       # 3 coefficients: not a model
    with pytest.raises(ValueError):
        O.init_undistort_rectify_map(K, None, None, [0, 0, 0, 0, 0, 0, 0, 0, 0], (8, 8))  # singular P * R


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["LEFT", "RIGHT", "eight_coefficients", "defaults"])
def test_gpu_init_undistort_rectify_map_matches_the_oracle(case):
    import orb_slam2_annotate_amd as amd
    if case in EUROC:
        c = EUROC[case]
        args = (c["K"], c["D"], c["R"], c["P"], SIZE)
    elif case == "eight_coefficients":
        c = EUROC["LEFT"]
        args = (c["K"], [-0.28, 0.07, 2e-4, 2e-5, 0.01, 0.02, -0.01, 0.003], c["R"], c["P"], (640, 480))
    else:
        args = (EUROC["RIGHT"]["K"], EUROC["RIGHT"]["D"][:4], None, None, (333, 77))
    mx_ref, my_ref = O.init_undistort_rectify_map(*args)
    mx, my = amd.initUndistortRectifyMap(*args)
    assert np.array_equal(mx.view(np.uint32), mx_ref.view(np.uint32)) and np.array_equal(my.view(np.uint32), my_ref.view(np.uint32))
    # ... and the maps drive cv::remap through the product's Rectifier exactly as the oracle's do
    if case == "LEFT":
        from orb_slam2_annotate_amd import synth
        raw = synth.render_frame(3, 752, 480)
        assert np.array_equal(amd.Rectifier(mx, my)(raw), O.remap_linear(raw, mx_ref, my_ref))
    with pytest.raises(amd.OrbfeError):
        amd.initUndistortRectifyMap(args[0], [0.1, 0.2, 0.3], None, None, (8, 8))
