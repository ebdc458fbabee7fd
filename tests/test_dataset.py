"""orb_slam3_amd/dataset.py -- the real-dataset input of the bench (ORBX_EUROC_DIR / ORBX_KITTI_DIR / ORBX_TUMVI_DIR) on generated PNG
folders in the three layouts the reference's examples read (Examples/Monocular/mono_euroc.cc:99 `mav0/cam0/data`, Examples/Stereo/
stereo_kitti.cc `image_0` / `image_1`, Examples/Monocular/mono_tum_vi.cc): 8-bit and 16-bit PNGs, centre crop, cycling, rank offsets."""
import numpy as np
import pytest


def _write_png(path, arr):
    from PIL import Image
    path.parent.mkdir(parents=True, exist_ok=True)
    if arr.dtype == np.uint16:
        Image.fromarray(arr).save(path)   # uint16 -> mode I;16
    else:
        Image.fromarray(arr).save(path)


def test_euroc_layout_8bit_crop_cycle_and_rank_offset(tmp_path, monkeypatch):
    from orb_slam3_amd import dataset
    rng = np.random.default_rng(1)
    frames = [rng.integers(0, 256, (484, 760), dtype=np.uint8) for _ in range(5)]   # larger than 752x480: centre-cropped
    for t, f in enumerate(frames):
        _write_png(tmp_path / "mav0" / "cam0" / "data" / f"{1403636579763555584 + t * 50000000}.png", f)
    monkeypatch.setenv("ORBX_EUROC_DIR", str(tmp_path))
    assert dataset.dataset_dir("euroc") == tmp_path
    got = dataset.load_mono("euroc", 7, 752, 480)
    assert got.shape == (7, 480, 752) and got.dtype == np.uint8 and got.flags["C_CONTIGUOUS"]
    for t in range(7):
        assert np.array_equal(got[t], frames[t % 5][2:482, 4:756]), t   # sorted by name = by timestamp; cycled past the end
    got = dataset.load_mono("euroc", 2, 752, 480, start=3)                 # rank r of the bench starts at r * B
    assert np.array_equal(got[0], frames[3][2:482, 4:756]) and np.array_equal(got[1], frames[4][2:482, 4:756])
    with pytest.raises(ValueError):
        dataset.load_mono("euroc", 1, 1024, 1024)


def test_tumvi_layout_16bit_high_byte(tmp_path, monkeypatch):
    from orb_slam3_amd import dataset
    rng = np.random.default_rng(2)
    f16 = rng.integers(0, 65536, (1024, 1024), dtype=np.uint16)
    f16[0, 0] = 65535
    _write_png(tmp_path / "mav0" / "cam0" / "data" / "1520530308199447626.png", f16)
    monkeypatch.setenv("ORBX_TUMVI_DIR", str(tmp_path))
    got = dataset.load_mono("tumvi", 1, 1024, 1024)
    assert got.shape == (1, 1024, 1024) and np.array_equal(got[0], (f16 >> 8).astype(np.uint8))   # imread(IMREAD_GRAYSCALE) strips to the high byte


def test_kitti_layout_stereo(tmp_path, monkeypatch):
    from orb_slam3_amd import dataset
    rng = np.random.default_rng(3)
    L = [rng.integers(0, 256, (376, 1241), dtype=np.uint8) for _ in range(3)]
    R = [rng.integers(0, 256, (376, 1241), dtype=np.uint8) for _ in range(3)]
    for t in range(3):
        _write_png(tmp_path / "image_0" / f"{t:06d}.png", L[t])
        _write_png(tmp_path / "image_1" / f"{t:06d}.png", R[t])
    monkeypatch.setenv("ORBX_KITTI_DIR", str(tmp_path))
    pairs = dataset.load_stereo("kitti", 4, 1241, 376)
    assert len(pairs) == 4
    for t, (l, r) in enumerate(pairs):
        assert np.array_equal(l, L[t % 3]) and np.array_equal(r, R[t % 3])
    monkeypatch.delenv("ORBX_KITTI_DIR")
    assert dataset.dataset_dir("kitti") is None
    with pytest.raises(FileNotFoundError):
        monkeypatch.setenv("ORBX_EUROC_DIR", str(tmp_path / "nothing"))
        dataset.load_mono("euroc", 1, 752, 480)
