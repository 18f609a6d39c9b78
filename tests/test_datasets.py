"""Image-list formats of the reference's example mains (orb_slam2_annotate_amd/datasets.py): KITTI
(Examples/Stereo/stereo_kitti.cc:130-164), EuRoC (stereo_euroc.cc:193-222), TUM (mono_tum.cc:129-160), on tiny
datasets written here (none ships with the reference)."""
import numpy as np
import pytest

from orb_slam2_annotate_amd import datasets, synth

PIL = pytest.importorskip("PIL.Image")


def _png(path, a):
    PIL.fromarray(a).save(path)


def test_kitti_layout(tmp_path):
    seq = tmp_path / "00"
    (seq / "image_0").mkdir(parents=True)
    (seq / "image_1").mkdir()
    (seq / "times.txt").write_text("0.000000e+00\n1.037224e-01\n\n2.074559e-01\n")  # an empty line is skipped (:140-149)
    pairs = [synth.render_stereo(3 + i, 64, 48) for i in range(3)]
    for i, (l, r) in enumerate(pairs):
        _png(seq / "image_0" / f"{i:06d}.png", l)
        _png(seq / "image_1" / f"{i:06d}.png", r)
    left, right, times = datasets.load_kitti(seq)
    assert times == [0.0, 0.1037224, 0.2074559] and left[2].endswith("image_0/000002.png") and right[0].endswith("image_1/000000.png")
    kind, frames = datasets.load_frames(f"kitti:{seq}", 2)
    assert kind == "kitti" and len(frames) == 4
    assert np.array_equal(frames[0], pairs[0][0]) and np.array_equal(frames[3], pairs[1][1])
    with pytest.raises(ValueError):
        datasets.load_frames(f"kitti:{seq}", 5)


def test_euroc_layout(tmp_path):
    l, r = tmp_path / "cam0", tmp_path / "cam1"
    l.mkdir()
    r.mkdir()
    stamps = ["1403636579763555584", "1403636579813555456"]
    (tmp_path / "MH01.txt").write_text("\n".join(stamps) + "\n")
    for s in stamps:
        _png(l / f"{s}.png", synth.render_frame(1, 40, 30))
        _png(r / f"{s}.png", synth.render_frame(2, 40, 30))
    left, right, times = datasets.load_euroc(l, r, tmp_path / "MH01.txt")
    assert left[1].endswith(stamps[1] + ".png") and abs(times[0] - 1403636579.763555584) < 1e-6
    kind, frames = datasets.load_frames(f"euroc:{l},{r},{tmp_path / 'MH01.txt'}", 2)
    assert len(frames) == 4 and frames[0].shape == (30, 40)


def test_tum_list(tmp_path):
    (tmp_path / "rgb").mkdir()
    (tmp_path / "rgb.txt").write_text("# color images\n# file: 'x.bag'\n# timestamp filename\n"
                                      "1305031102.175304 rgb/1305031102.175304.png\n1305031102.211214 rgb/1305031102.211214.png\n")
    files, times = datasets.load_tum(tmp_path)
    assert len(files) == 2 and files[0].endswith("rgb/1305031102.175304.png") and times[1] == 1305031102.211214


@pytest.mark.gpu
def test_tum_colour_frames_are_converted_like_cvtcolor(tmp_path):
    import oracle_lib as orc
    (tmp_path / "rgb").mkdir()
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (30, 40, 3), dtype=np.uint8)
    _png(tmp_path / "rgb" / "a.png", rgb)
    (tmp_path / "rgb.txt").write_text("#\n#\n#\n1.0 rgb/a.png\n")
    _, frames = datasets.load_frames(f"tum:{tmp_path}", 1)
    # imread delivers B,G,R; TUM1.yaml has Camera.RGB: 1, so Tracking applies CV_RGB2GRAY to that BGR data
    # (src/Tracking.cc:250-262): the first channel the conversion sees -- weight 4899 -- is BLUE
    R_, G_, B_ = (rgb[..., k].astype(np.int64) for k in range(3))
    want = (B_ * 4899 + G_ * 9617 + R_ * 1868 + 8192) >> 14
    assert np.array_equal(frames[0], want.astype(np.uint8))
    # Camera.RGB: 0 (CV_BGR2GRAY on the same BGR data) is the true luminance
    lum = (R_ * 4899 + G_ * 9617 + B_ * 1868 + 8192) >> 14
    assert np.array_equal(datasets.read_gray(tmp_path / "rgb" / "a.png", mbRGB=False), lum.astype(np.uint8))
    assert not np.array_equal(want, lum)
