"""CPU known-answer tests of the oracle's cv::undistortPoints restatement (orc_undistort_points):
properties of the published algorithm that need no OpenCV."""
import numpy as np

import oracle_lib as O

TUM1 = (np.array([517.306408, 516.469215, 318.643040, 255.313989], np.float32),       # Examples/Monocular/TUM1.yaml
        np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32))


def _distort(xy, K4, d):
    """forward radial-tangential model in float64 (the model undistortPoints inverts)"""
    fx, fy, cx, cy = [float(v) for v in K4]
    k1, k2, p1, p2, k3 = [float(v) for v in d]
    x, y = (xy[:, 0] - cx) / fx, (xy[:, 1] - cy) / fy
    r2 = x * x + y * y
    kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * kr + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * kr + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([xd * fx + cx, yd * fy + cy], axis=1)


def test_zero_distortion_is_identity():
    rng = np.random.default_rng(0)
    pts = rng.uniform(0, 640, (500, 2)).astype(np.float32)
    out = O.undistort_points(pts, TUM1[0], np.zeros(5, np.float32))
    assert np.abs(out - pts).max() <= 6.2e-5  # one float ulp at 640
    assert np.array_equal(O.undistort_points(pts, TUM1[0], np.zeros(0, np.float32)), out)  # 1 iteration, same result


def test_inverts_the_forward_model_near_the_centre():
    rng = np.random.default_rng(1)
    ideal = np.stack([rng.uniform(200, 440, 400), rng.uniform(160, 350, 400)], axis=1)
    seen = _distort(ideal, *TUM1).astype(np.float32)
    back = O.undistort_points(seen, *TUM1)
    assert np.abs(back - ideal).max() < 0.02  # 5 fixed-point iterations, not a converged solve
    # principal point is a fixed point
    pp = np.array([[TUM1[0][2], TUM1[0][3]]], np.float32)
    assert np.array_equal(O.undistort_points(pp, *TUM1), pp)


def test_image_bounds_and_rgbd():
    b = O.image_bounds(640, 480, TUM1[0], TUM1[1])
    c = O.undistort_points(np.array([[0, 0], [640, 0], [0, 480], [640, 480]], np.float32), *TUM1)
    assert b == (min(c[0, 0], c[2, 0]), max(c[1, 0], c[3, 0]), min(c[0, 1], c[1, 1]), max(c[2, 1], c[3, 1]))
    assert O.image_bounds(640, 480, TUM1[0], np.zeros(5, np.float32)) == (0.0, 640.0, 0.0, 480.0)
    depth = np.zeros((480, 640), np.float32)
    depth[100, 200] = 2.5
    ur, dp = O.stereo_from_rgbd([200.9, 10.0], [100.9, 10.0], [198.0, 9.0], depth, 40.0)
    assert dp.tolist() == [2.5, -1.0] and ur.tolist() == [float(np.float32(198.0) - np.float32(40.0) / np.float32(2.5)), -1.0]
