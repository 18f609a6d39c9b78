"""Image lists of the three dataset layouts the reference's example mains read, so that `bench.py --dataset`
(or $ORBFE_DATASET) can feed real frames instead of the synthetic ones where a dataset is available:

  kitti:<sequence dir>   Examples/Stereo/stereo_kitti.cc:130-164  times.txt + image_0/%06d.png, image_1/%06d.png
  euroc:<cam0 data dir>,<cam1 data dir>,<timestamps file>
                         Examples/Stereo/stereo_euroc.cc:193-222  one timestamp per line, <dir>/<t>.png, t in ns
  tum:<sequence dir>     Examples/Monocular/mono_tum.cc:129-160   rgb.txt: 3 header lines, then "<t> <file>"

Decoding uses Pillow (the reference uses cv::imread(..., CV_LOAD_IMAGE_UNCHANGED)); colour images are converted with
the library's own cvtColor restatement on the GPU (src/Tracking.cc:176-262), so no OpenCV is involved anywhere.
EuRoC frames are returned as stored, NOT rectified: the example main rectifies with maps from
cv::initUndistortRectifyMap, which is outside the path; pass rectified images or use ingest.Rectifier
with your own maps.  No dataset ships with the reference or this repository.
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Tuple

import numpy as np


def load_kitti(sequence_dir) -> Tuple[List[str], List[str], List[float]]:
    """stereo_kitti.cc:130-164 -> (left files, right files, timestamps)."""
    seq = Path(sequence_dir)
    times = [float(s.split()[0]) for s in (seq / "times.txt").read_text().splitlines() if s.strip()]
    left = [str(seq / "image_0" / f"{i:06d}.png") for i in range(len(times))]
    right = [str(seq / "image_1" / f"{i:06d}.png") for i in range(len(times))]
    return left, right, times


def load_euroc(left_dir, right_dir, times_file) -> Tuple[List[str], List[str], List[float]]:
    """stereo_euroc.cc:193-222: every non-empty line is a timestamp in ns and the file name stem."""
    stamps = [s.strip() for s in Path(times_file).read_text().splitlines() if s.strip()]
    left = [str(Path(left_dir) / f"{s}.png") for s in stamps]
    right = [str(Path(right_dir) / f"{s}.png") for s in stamps]
    return left, right, [float(s) / 1e9 for s in stamps]


def load_tum(sequence_dir) -> Tuple[List[str], List[float]]:
    """mono_tum.cc:129-160: rgb.txt, three comment lines skipped, then "<timestamp> <relative file>"."""
    seq = Path(sequence_dir)
    lines = (seq / "rgb.txt").read_text().splitlines()[3:]
    files, times = [], []
    for s in lines:
        if not s.strip():
            continue
        t, f = s.split()[:2]
        times.append(float(t))
        files.append(str(seq / f))
    return files, times


def read_image(path) -> np.ndarray:
    """The stored pixels (H x W uint8 for 8-bit gray, H x W x 3|4 for colour), like imread(..., UNCHANGED)."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode in ("L", "P", "1"):
            return np.asarray(im.convert("L"), dtype=np.uint8)
        if im.mode in ("I;16", "I"):
            raise ValueError(f"{path}: 16-bit image; ORBextractor asserts CV_8UC1 (src/ORBextractor.cc:1126)")
        return np.asarray(im.convert("RGBA" if "A" in im.mode else "RGB"), dtype=np.uint8)


# Camera.RGB of EVERY example configuration of the reference is 1 (Examples/*/*.yaml: TUM1-3, KITTI00-12, EuRoC)
MB_RGB = {"tum": True, "kitti": True, "euroc": True}


def read_gray(path, mbRGB: bool = True, device: int = 0) -> np.ndarray:
    """An 8-bit gray frame as Tracking::GrabImage* hands it to the extractor.  Gray files as stored.  Colour files the way
    the reference processes them: the example mains call cv::imread(..., CV_LOAD_IMAGE_UNCHANGED)
    (Examples/Monocular/mono_tum.cc:66), which delivers B, G, R(, A) byte order, and Tracking applies
    cvtColor(CV_RGB2GRAY / CV_RGBA2GRAY) when mbRGB (= Camera.RGB) is set, else CV_BGR2GRAY (src/Tracking.cc:250-262).
    With Camera.RGB: 1 on imread's BGR data that is gray = (4899*B + 9617*G + 1868*R + 2^13) >> 14 -- NOT the
    luminance of the picture, but what the reference feeds its extractor (round-2 ADVICE: this function used to convert
    Pillow's true RGB order, i.e. it extracted from different pixels than the reference)."""
    a = read_image(path)  # Pillow: R, G, B(, A)
    if a.ndim == 2:
        return a
    from .ingest import cvtColorToGray
    bgr = np.ascontiguousarray(a[..., ::-1] if a.shape[2] == 3 else a[..., [2, 1, 0, 3]])  # what imread delivers
    return cvtColorToGray(bgr, rgb=bool(mbRGB), device=device)


def load_frames(spec: str, n_units: int, device: int = 0):
    """`kitti:DIR` / `euroc:L,R,TIMES` / `tum:DIR` -> (kind, list of gray frames): the first n_units stereo pairs
    (ordered L0,R0,L1,R1,...) or mono frames; raises if the dataset holds fewer."""
    kind, _, arg = spec.partition(":")
    if kind == "kitti":
        left, right, _ = load_kitti(arg)
    elif kind == "euroc":
        left, right, _ = load_euroc(*arg.split(","))
    elif kind == "tum":
        files, _ = load_tum(arg)
        if len(files) < n_units:
            raise ValueError(f"{spec}: {len(files)} frames, {n_units} asked")
        return kind, [read_gray(f, MB_RGB[kind], device) for f in files[:n_units]]
    else:
        raise ValueError(f"unknown dataset kind {kind!r} (kitti | euroc | tum)")
    if len(left) < n_units:
        raise ValueError(f"{spec}: {len(left)} stereo frames, {n_units} asked")
    out = []
    for a, b in zip(left[:n_units], right[:n_units]):
        out.append(read_gray(a, MB_RGB[kind], device))
        out.append(read_gray(b, MB_RGB[kind], device))
    return kind, out
