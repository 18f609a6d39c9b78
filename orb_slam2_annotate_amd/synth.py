"""Seeded synthetic grayscale frames / stereo pairs / sequences.

The datasets named in BASELINE.json (TUM fr1_xyz, KITTI 00-07, EuRoC MH_01) are
not available offline, so tests and bench.py render
scenes with the statistics the ORB front-end cares about: corners at many
scales (random rectangles / rotated rectangles / triangles of uniform
intensity over low-frequency value noise) plus sensor noise.  Everything is a
pure function of the integer seed, so the GPU box regenerates the same bytes.
"""
from __future__ import annotations

import numpy as np

__all__ = ["render_frame", "render_stereo", "render_stereo_textured", "STEREO_SCENES", "render_stereo_raw", "rectify_maps", "EUROC_CAM0", "EUROC_CAM1",
           "render_sequence", "adversarial"]


def _background(rng: np.random.Generator, w: int, h: int) -> np.ndarray:
    cell = 48
    gw, gh = w // cell + 3, h // cell + 3
    g = rng.uniform(50.0, 200.0, size=(gh, gw))
    xs = np.arange(w) / cell
    ys = np.arange(h) / cell
    x0 = xs.astype(int)
    y0 = ys.astype(int)
    fx = (xs - x0)[None, :]
    fy = (ys - y0)[:, None]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def _make_shapes(rng: np.random.Generator, w: int, h: int, n: int, max_disp: int):
    """Returns list of (poly[k,2] float, intensity, disparity)."""
    shapes = []
    kinds = rng.integers(0, 3, size=n)
    cx = rng.uniform(-20, w + 20, size=n)
    cy = rng.uniform(-20, h + 20, size=n)
    size = np.exp(rng.uniform(np.log(5.0), np.log(90.0), size=n))
    aspect = rng.uniform(0.4, 1.0, size=n)
    theta = rng.uniform(0, np.pi, size=n)
    inten = rng.integers(0, 256, size=n)
    disp = rng.integers(0, max_disp + 1, size=n) if max_disp > 0 else np.zeros(n, dtype=int)
    tri = rng.uniform(-1.0, 1.0, size=(n, 3, 2))
    for i in range(n):
        if kinds[i] == 0:  # axis-aligned rectangle
            hw, hh = size[i] / 2, size[i] * aspect[i] / 2
            poly = np.array([[-hw, -hh], [hw, -hh], [hw, hh], [-hw, hh]])
        elif kinds[i] == 1:  # rotated rectangle
            hw, hh = size[i] / 2, size[i] * aspect[i] / 2
            base = np.array([[-hw, -hh], [hw, -hh], [hw, hh], [-hw, hh]])
            c, s = np.cos(theta[i]), np.sin(theta[i])
            poly = base @ np.array([[c, s], [-s, c]])
        else:  # triangle
            poly = tri[i] * size[i] / 2
            # make it counter-clockwise consistent for the half-plane test
            e1, e2 = poly[1] - poly[0], poly[2] - poly[0]
            if e1[0] * e2[1] - e1[1] * e2[0] < 0:
                poly = poly[::-1].copy()
        poly = poly + np.array([cx[i], cy[i]])
        shapes.append((poly, int(inten[i]), int(disp[i])))
    return shapes


def _paint(canvas: np.ndarray, shapes, shift_sign: float, ox: float = 0.0, oy: float = 0.0):
    h, w = canvas.shape
    for poly, inten, disp in shapes:
        p = poly.copy()
        p[:, 0] -= shift_sign * disp + ox
        p[:, 1] -= oy
        x0 = max(int(np.floor(p[:, 0].min())), 0)
        x1 = min(int(np.ceil(p[:, 0].max())) + 1, w)
        y0 = max(int(np.floor(p[:, 1].min())), 0)
        y1 = min(int(np.ceil(p[:, 1].max())) + 1, h)
        if x0 >= x1 or y0 >= y1:
            continue
        xs = np.arange(x0, x1)[None, :] + 0.5
        ys = np.arange(y0, y1)[:, None] + 0.5
        k = len(p)
        # orientation sign of polygon
        area = 0.0
        for j in range(k):
            a, b = p[j], p[(j + 1) % k]
            area += a[0] * b[1] - a[1] * b[0]
        sgn = 1.0 if area >= 0 else -1.0
        mask = np.ones((y1 - y0, x1 - x0), dtype=bool)
        for j in range(k):
            a, b = p[j], p[(j + 1) % k]
            mask &= sgn * ((b[0] - a[0]) * (ys - a[1]) - (b[1] - a[1]) * (xs - a[0])) >= 0
        canvas[y0:y1, x0:x1][mask] = inten


def _finish(canvas: np.ndarray, rng: np.random.Generator, sigma: float) -> np.ndarray:
    noisy = canvas + rng.normal(0.0, sigma, size=canvas.shape)
    return np.clip(np.rint(noisy), 0, 255).astype(np.uint8)


def render_frame(seed: int, w: int = 640, h: int = 480, n_shapes: int | None = None,
                 sigma: float = 3.0) -> np.ndarray:
    rng = np.random.default_rng([0x0B5EED, int(seed)])
    if n_shapes is None:
        n_shapes = int(rng.integers(400, 1500) * (w * h) / (640 * 480))
    canvas = _background(rng, w, h)
    _paint(canvas, _make_shapes(rng, w, h, n_shapes, 0), 0.0)
    return _finish(canvas, rng, sigma)


def render_stereo(seed: int, w: int = 1241, h: int = 376, n_shapes: int | None = None,
                  max_disp: int = 96, sigma: float = 3.0):
    """Left/right pair: every object is re-rendered in the right view shifted left
    by its own integer disparity (0..max_disp); the background has disparity 0."""
    rng = np.random.default_rng([0x57E8E0, int(seed)])
    if n_shapes is None:
        n_shapes = int(rng.integers(400, 1500) * (w * h) / (640 * 480))
    bg = _background(rng, w, h)
    shapes = _make_shapes(rng, w, h, n_shapes, max_disp)
    # paint far objects first so near ones (large disparity) occlude them
    shapes.sort(key=lambda s: s[2])
    left, right = bg.copy(), bg.copy()
    _paint(left, shapes, 0.0)
    _paint(right, shapes, 1.0)
    return _finish(left, rng, sigma), _finish(right, rng, sigma)


def _disparity_planes(rng: np.random.Generator, w: int, h: int, max_disp: float) -> np.ndarray:
    """Piecewise-planar, sub-pixel disparity field of a driving scene in RIGHT-view coordinates: a far backdrop above the
    horizon, a ground plane (disparity grows linearly with the image row, as bf*(y - cy)/(fy*h_cam) does), two slanted
    side walls and a few fronto-parallel / slanted boxes in front of them.  Nearer surfaces overwrite farther ones."""
    ys = np.arange(h, dtype=np.float64)[:, None]
    xs = np.arange(w, dtype=np.float64)[None, :]
    horizon = h * rng.uniform(0.40, 0.52)
    far = rng.uniform(0.6, 3.0)
    ground_gain = rng.uniform(0.55, 0.95) * max_disp / max(h - horizon, 1.0)
    d = np.full((h, w), far) + 0.0 * xs
    d = np.maximum(d, far + ground_gain * (ys - horizon))
    # side walls: disparity falls off linearly from the image border towards a vanishing column
    for side in (0, 1):
        reach = w * rng.uniform(0.18, 0.36)
        near = rng.uniform(0.35, 0.8) * max_disp
        top = horizon - h * rng.uniform(0.25, 0.45)
        t = (reach - xs) / reach if side == 0 else (xs - (w - reach)) / reach
        wall = far + (near - far) * np.clip(t, 0.0, 1.0)
        m = (t > 0) & (ys > top + (horizon - top) * (1 - np.clip(t, 0, 1))) & (wall > d)
        d = np.where(m, wall, d)
    for _ in range(int(rng.integers(3, 8))):  # boxes ("cars", "signs"): planes with a small horizontal / vertical slant
        bw, bh = w * rng.uniform(0.05, 0.2), h * rng.uniform(0.12, 0.4)
        cx, cy = rng.uniform(0, w), horizon + rng.uniform(-0.15, 0.35) * h
        d0 = rng.uniform(0.1, 0.9) * max_disp
        gx, gy = rng.uniform(-0.03, 0.03), rng.uniform(-0.02, 0.02)
        plane = d0 + gx * (xs - cx) + gy * (ys - cy)
        m = (np.abs(xs - cx) < bw / 2) & (np.abs(ys - cy) < bh / 2) & (plane > d)
        d = np.where(m, plane, d)
    return np.clip(d, 0.25, max_disp)


def render_stereo_textured(seed: int, w: int = 1241, h: int = 376, n_shapes: int | None = None,
                           max_disp: float = 72.0, sigma: float = 3.0):
    """Left/right pair of ONE textured scene: the same scene content in both eyes, related by a piecewise-planar
    SUB-PIXEL disparity field, plus independent sensor noise per eye.  The left view is a window of a wide canvas
    (value-noise background + shapes at all scales, as render_frame); the right view samples that canvas bilinearly at
    x + D_R(x, y) (x_R = x_L - d), so almost every left corner has its counterpart on the same row -- what a rectified
    KITTI pair looks like to Frame::ComputeStereoMatches (src/Frame.cc:512-686), unlike render_stereo's per-object
    integer shifts whose corners are mostly occlusion junctions."""
    rng = np.random.default_rng([0x7E87ED, int(seed)])
    if n_shapes is None:
        n_shapes = int(rng.integers(400, 1500) * (w * h) / (640 * 480))
    pad = int(np.ceil(max_disp)) + 2
    W = w + pad
    canvas = _background(rng, W, h)
    _paint(canvas, _make_shapes(rng, W, h, int(n_shapes * W / w), 0), 0.0)
    d = _disparity_planes(rng, w, h, float(max_disp))
    sx = np.arange(w, dtype=np.float64)[None, :] + d
    x0 = np.floor(sx).astype(np.int64)
    fx = sx - x0
    x0 = np.clip(x0, 0, W - 2)
    rows = np.arange(h)[:, None]
    right = canvas[rows, x0] * (1.0 - fx) + canvas[rows, x0 + 1] * fx
    return _finish(canvas[:, :w].copy(), rng, sigma), _finish(right, rng, sigma)


STEREO_SCENES = {"shapes": render_stereo, "textured": render_stereo_textured}

# shape of the EuRoC cam0 / cam1 calibrations (Examples/Stereo/EuRoC.yaml: LEFT.K / LEFT.D, RIGHT.K / RIGHT.D)
EUROC_CAM0 = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375, k1=-0.28340811, k2=0.07395907, p1=0.00019359,
                  p2=1.76187114e-05, rot_deg=0.4)
EUROC_CAM1 = dict(fx=457.587, fy=456.134, cx=379.999, cy=255.238, k1=-0.28368365, k2=0.07451284, p1=-0.00010473,
                  p2=-3.55590700e-05, rot_deg=-0.3)


def _rectify_forward(x, y, fx, fy, cx, cy, k1, k2, p1, p2, k3=0.0, rot_deg=0.4, new_f_scale=1.0):
    """rectified pixel (x, y) -> raw pixel (u, v): a radial-tangential camera behind a small rectifying rotation."""
    a = np.deg2rad(rot_deg)
    R = np.array([[np.cos(a), -np.sin(a), 0.002], [np.sin(a), np.cos(a), -0.003], [-0.002, 0.003, 1.0]])
    nfx, nfy = fx * new_f_scale, fy * new_f_scale
    pts = np.stack([(x - cx) / nfx, (y - cy) / nfy, np.ones_like(x)], axis=-1) @ np.linalg.inv(R).T
    xn, yn = pts[..., 0] / pts[..., 2], pts[..., 1] / pts[..., 2]
    r2 = xn * xn + yn * yn
    kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    u = fx * (xn * kr + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)) + cx
    v = fy * (yn * kr + p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn) + cy
    return u, v


def rectify_maps(w, h, fx, fy, cx, cy, k1, k2, p1, p2, k3=0.0, rot_deg=0.4, new_f_scale=1.0):
    """Test / bench data generator (NOT a restatement of cv::initUndistortRectifyMap, which is outside the path): the
    two CV_32F maps of a radial-tangential camera with a small rectifying rotation -- same model and output type as
    the EuRoC setup of Examples/Stereo/stereo_euroc.cc:97-98."""
    x, y = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    u, v = _rectify_forward(x, y, fx, fy, cx, cy, k1, k2, p1, p2, k3, rot_deg, new_f_scale)
    return u.astype(np.float32), v.astype(np.float32)


def _unrectify(img: np.ndarray, cam: dict) -> np.ndarray:
    """The raw (distorted) image whose rectification with rectify_maps(**cam) gives `img` back (up to the two bilinear
    resamplings): raw(u, v) = img(M^-1(u, v)), M^-1 by fixed-point iteration on the forward model; float result."""
    h, w = img.shape
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    x, y = u.copy(), v.copy()
    for _ in range(12):
        fu, fv = _rectify_forward(x, y, **cam)
        x += u - fu
        y += v - fv
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    ax, ay = x - x0, y - y0
    inside = (x0 >= 0) & (x0 < w - 1) & (y0 >= 0) & (y0 < h - 1)
    x0 = np.clip(x0, 0, w - 2)
    y0 = np.clip(y0, 0, h - 2)
    f = img.astype(np.float64)
    out = (f[y0, x0] * (1 - ax) + f[y0, x0 + 1] * ax) * (1 - ay) + (f[y0 + 1, x0] * (1 - ax) + f[y0 + 1, x0 + 1] * ax) * ay
    return np.where(inside, out, 0.0)


def render_stereo_raw(seed: int, w: int = 752, h: int = 480, max_disp: float = 40.0, sigma: float = 3.0,
                      cam_left: dict = EUROC_CAM0, cam_right: dict = EUROC_CAM1):
    """A RAW (unrectified) EuRoC-like stereo pair: the textured scene of render_stereo_textured seen through two
    radial-tangential cameras, sensor noise added on the raw images.  cv::remap with rectify_maps(w, h, **cam_left) /
    (**cam_right) -- Examples/Stereo/stereo_euroc.cc:136-137 -- brings the pair back to a row-aligned, matchable one."""
    left, right = render_stereo_textured(seed, w, h, max_disp=max_disp, sigma=0.0)
    rng = np.random.default_rng([0xE0C0, int(seed)])
    return (_finish(_unrectify(left, cam_left), rng, sigma), _finish(_unrectify(right, cam_right), rng, sigma))


def render_sequence(seed: int, n_frames: int, w: int = 640, h: int = 480,
                    n_shapes: int | None = None, sigma: float = 3.0, step: float = 1.5):
    """Same scene under a small per-frame camera translation (fresh noise per frame)."""
    rng = np.random.default_rng([0x5E0CE, int(seed)])
    if n_shapes is None:
        n_shapes = int(rng.integers(400, 1500) * (w * h) / (640 * 480))
    margin = int(step * n_frames) + 8
    bg = _background(rng, w + margin, h + margin)
    shapes = _make_shapes(rng, w + margin, h + margin, int(n_shapes * (1 + margin / w)), 0)
    big = bg.copy()
    _paint(big, shapes, 0.0)
    frames = []
    for f in range(n_frames):
        ox = int(round(step * f))
        oy = int(round(0.5 * step * f))
        frames.append(_finish(big[oy:oy + h, ox:ox + w], rng, sigma))
    return frames


def adversarial(kind: str, w: int = 640, h: int = 480, seed: int = 0) -> np.ndarray:
    """Edge-case images: 'constant' (0 keypoints), 'noise', 'checker' (ties everywhere)."""
    if kind == "constant":
        return np.full((h, w), 127, dtype=np.uint8)
    if kind == "noise":
        return np.random.default_rng([0xA0153, int(seed)]).integers(0, 256, size=(h, w), dtype=np.uint8)
    if kind == "checker":
        yy, xx = np.mgrid[0:h, 0:w]
        return np.where(((xx // 8) + (yy // 8)) % 2 == 0, 30, 220).astype(np.uint8)
    raise ValueError(kind)
