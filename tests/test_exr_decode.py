"""CPU: OpenEXR decoding of the C++ loader (include/trayracer_image.hpp, namespace imgdec::exr) -- the format of the
reference's skyboxes (res/skyboxes/*.exr, skybox.rs:56-63: image::open(path)?.into_rgb8()).  Parity unpinned: those files
are git-LFS stubs and no EXR library is installed, so the files come from tests/exr_writer.py, an encoder written separately
from the decoder (forward transforms, Python).  Expected pixels: the image crate's Rgb32F -> Rgb8 conversion,
round(clamp(v, 0, 1) * 255), on the values the file holds (HALF rounds on writing; PXR24 keeps 24 bits of a FLOAT)."""
import os
import subprocess

import numpy as np
import pytest

from tests import exr_writer as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "examples", "trr_cli")
pytestmark = pytest.mark.skipif(not os.path.exists(CLI), reason="examples/trr_cli not built (make -C raytracing_amd/csrc)")


def decode(path, tmp_path):
    out = str(tmp_path / "out.ppm")
    r = subprocess.run([CLI, "--decode-image", path, out], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr.strip())
    d = open(out, "rb").read()
    head = d.split(b"\n", 3)
    w, h = (int(x) for x in head[1].split())
    return np.frombuffer(head[3], np.uint8, w * h * 3).reshape(h, w, 3)


def to_u8(v):
    c = np.clip(np.asarray(v, np.float32), np.float32(0), np.float32(1)) * np.float32(255)
    c = np.where(np.isnan(c), 0, c)
    return np.floor(c.astype(np.float64) + 0.5).astype(np.uint8)


def stored(arr, typ, comp):
    if typ == W.HALF:
        return arr.astype(np.float16).astype(np.float32)
    if typ == W.FLOAT:
        return W.pxr24_expected(arr.astype(np.float32), typ) if comp == W.PXR24 else arr.astype(np.float32)
    return arr.astype(np.float32)


def picture(w, h, seed):
    """A smooth HDR-like gradient with noise, values from below 0 to above 1, exact zeros and repeated runs."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    r = 1.4 * x / max(1, w - 1) - 0.2 + 0.02 * rng.standard_normal((h, w)).astype(np.float32)
    g = 0.5 + 0.5 * np.sin(x / 5.0) * np.cos(y / 7.0)
    b = (y / max(1, h - 1)) ** 2 * 3.0
    b[:, : w // 3] = 0.0
    g[h // 2:, w // 2:] = 0.25
    return r.astype(np.float32), g.astype(np.float32), b.astype(np.float32)


COMPS = {"none": W.NONE, "rle": W.RLE, "zips": W.ZIPS, "zip": W.ZIP, "piz": W.PIZ, "pxr24": W.PXR24}


@pytest.mark.parametrize("comp", list(COMPS))
@pytest.mark.parametrize("typ", [W.HALF, W.FLOAT])
@pytest.mark.parametrize("size", [(67, 45), (32, 32), (5, 70)])
def test_scanline_files(tmp_path, comp, typ, size):
    w, h = size
    r, g, b = picture(w, h, 1)
    path = str(tmp_path / "a.exr")
    for k in W.STATS:
        W.STATS[k] = 0
    W.write_exr(path, {"R": (typ, r), "G": (typ, g), "B": (typ, b)}, COMPS[comp])
    if comp in ("zip", "piz", "pxr24") and typ == W.HALF and size == (67, 45):
        assert W.STATS["compressed"] > 0   # the path under test was taken (a noisy last block may still be stored raw)
    got = decode(path, tmp_path)
    want = np.stack([to_u8(stored(c, typ, COMPS[comp])) for c in (r, g, b)], axis=-1)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("comp", ["none", "zip", "piz"])
def test_extra_channels_mixed_types_and_a_data_window_off_the_origin(tmp_path, comp):
    """Alpha and depth channels are skipped (into_rgb8 drops them); channel types may differ; the data window's origin does not
    move the pixels; decreasing line order stores the blocks bottom-up."""
    w, h = 41, 37
    r, g, b = picture(w, h, 2)
    a = np.full((h, w), 0.5, np.float32)
    z = np.arange(w * h, dtype=np.uint32).reshape(h, w)
    path = str(tmp_path / "b.exr")
    W.write_exr(path, {"R": (W.HALF, r), "G": (W.FLOAT, g), "B": (W.HALF, b), "A": (W.HALF, a), "Z": (W.UINT, z)}, COMPS[comp],
                data_window_origin=(-7, 13), line_order=1)
    got = decode(path, tmp_path)
    want = np.stack([to_u8(stored(r, W.HALF, 0)), to_u8(g), to_u8(stored(b, W.HALF, 0))], axis=-1)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("comp", ["none", "zips", "zip", "piz", "pxr24"])
def test_tiled_files(tmp_path, comp):
    w, h = 75, 50
    r, g, b = picture(w, h, 3)
    path = str(tmp_path / "t.exr")
    W.write_exr(path, {"R": (W.HALF, r), "G": (W.HALF, g), "B": (W.HALF, b)}, COMPS[comp], tile=(32, 16))
    got = decode(path, tmp_path)
    want = np.stack([to_u8(stored(c, W.HALF, 0)) for c in (r, g, b)], axis=-1)
    assert np.array_equal(got, want)


def test_piz_with_many_distinct_values_uses_the_16_bit_wavelet(tmp_path):
    """More than 2^14 distinct 16-bit values in a block switch PIZ to its modulo-2^16 wavelet: three HALF ramps whose bit
    patterns cover 0 .. 24575 inside every 32-line block (smooth, so the block does shrink and is not stored raw)."""
    w, h = 256, 64
    idx = (np.arange(w * h, dtype=np.uint32) % (w * 32)).reshape(h, w)
    chans = [(idx + k * w * 32).astype(np.uint16).view(np.float16).astype(np.float32) for k in range(3)]
    path = str(tmp_path / "p.exr")
    for k in W.STATS:
        W.STATS[k] = 0
    W.write_exr(path, {"R": (W.HALF, chans[0]), "G": (W.HALF, chans[1]), "B": (W.HALF, chans[2])}, W.PIZ)
    assert W.STATS["compressed"] == 2 and W.STATS["raw"] == 0 and W.STATS["wavelet16"] == 6
    got = decode(path, tmp_path)
    assert np.array_equal(got, np.stack([to_u8(c) for c in chans], axis=-1))


def test_piz_float_channels_and_noisy_blocks_stored_raw(tmp_path):
    w, h = 96, 40
    rng = np.random.default_rng(4)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    smooth = [np.round(x / 8) / 16, np.round(y / 4) / 8, np.round((x + y) / 16) / 4]   # few distinct values: FLOAT words compress
    noisy = [rng.random((h, w), dtype=np.float32) * 2.0 for _ in range(3)]
    for chans, want_raw in ((smooth, False), (noisy, True)):
        path = str(tmp_path / "f.exr")
        for k in W.STATS:
            W.STATS[k] = 0
        W.write_exr(path, {"R": (W.FLOAT, chans[0]), "G": (W.FLOAT, chans[1]), "B": (W.FLOAT, chans[2])}, W.PIZ)
        assert (W.STATS["raw"] > 0) == want_raw and (W.STATS["compressed"] > 0) == (not want_raw)
        got = decode(path, tmp_path)
        assert np.array_equal(got, np.stack([to_u8(c) for c in chans], axis=-1))


def test_special_values(tmp_path):
    """NaN -> 0 (the image crate would panic on it), infinities clamp, negative zero, subnormal halves."""
    r = np.array([[np.nan, np.inf, -np.inf, -0.0, 6e-8, 0.5, 1.0, 0.49803925]], np.float32)
    path = str(tmp_path / "s.exr")
    W.write_exr(path, {"R": (W.HALF, r), "G": (W.FLOAT, r), "B": (W.HALF, r)}, W.ZIP)
    got = decode(path, tmp_path)
    assert got[0, :, 1].tolist() == [0, 255, 0, 0, 0, 128, 255, 127]
    assert got[0, :, 0].tolist() == [0, 255, 0, 0, 0, 128, 255, 127]


def test_unsupported_and_broken_files_fail_like_a_failed_image_open(tmp_path):
    w, h = 16, 16
    r, g, b = picture(w, h, 5)
    path = str(tmp_path / "y.exr")
    W.write_exr(path, {"Y": (W.HALF, r)}, W.ZIP)   # luminance only: no R, G, B
    with pytest.raises(RuntimeError, match="R, G and B"):
        decode(path, tmp_path)
    W.write_exr(path, {"R": (W.HALF, r), "G": (W.HALF, g), "B": (W.HALF, b)}, W.ZIP)
    d = bytearray(open(path, "rb").read())
    open(path, "wb").write(d[: len(d) - 40])   # truncated
    with pytest.raises(RuntimeError, match="EXR"):
        decode(path, tmp_path)
    d[len(d) // 2] ^= 0x55
    d[-20] ^= 0xFF   # corrupt the compressed data
    open(path, "wb").write(d)
    with pytest.raises(RuntimeError, match="EXR"):
        decode(path, tmp_path)
    # the reference's own skyboxes are git-LFS pointer files: not EXR at all
    stub = "/root/reference/res/skyboxes/studio_small_08_4k.exr"
    if os.path.exists(stub):
        with pytest.raises(RuntimeError):
            decode(stub, tmp_path)


def test_piz_with_an_over_subscribed_code_table_is_refused(tmp_path, monkeypatch):
    """A PIZ code-length table whose Kraft sum exceeds 1 (here: every symbol claims a 1-bit code) gives canonical codes that
    do not fit their length; the decoder must refuse the table, not index its look-up table with them."""
    w, h = 64, 32   # smooth ramps: the PIZ block is smaller than the raw one, so the writer keeps it (a noisy block is stored raw)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    r, g, b = x / w, y / h, np.full((h, w), 0.25, np.float32)
    path = str(tmp_path / "bad_table.exr")
    for claimed in (1, 2):   # every symbol claims a 1-bit code; every symbol claims a 2-bit code
        monkeypatch.setattr(W, "LENGTH_HOOK", lambda length, c=claimed: {s: (c if l else 0) for s, l in length.items()})
        W.STATS.update(compressed=0, raw=0)
        W.write_exr(path, {"R": (W.HALF, r), "G": (W.HALF, g), "B": (W.HALF, b)}, W.PIZ)
        assert W.STATS["compressed"] == 1 and W.STATS["raw"] == 0
        with pytest.raises(RuntimeError, match="over-subscribed"):
            decode(path, tmp_path)
    monkeypatch.setattr(W, "LENGTH_HOOK", None)   # the same picture with an honest table decodes
    W.write_exr(path, {"R": (W.HALF, r), "G": (W.HALF, g), "B": (W.HALF, b)}, W.PIZ)
    assert decode(path, tmp_path).shape == (h, w, 3)


def test_a_header_that_claims_an_absurd_size_is_refused_before_anything_is_allocated(tmp_path):
    w, h = 8, 8
    r, g, b = picture(w, h, 3)
    path = str(tmp_path / "huge.exr")
    W.write_exr(path, {"R": (W.HALF, r), "G": (W.HALF, g), "B": (W.HALF, b)}, W.NONE)
    d = bytearray(open(path, "rb").read())
    i = d.index(b"dataWindow\0box2i\0") + len(b"dataWindow\0box2i\0") + 4
    d[i:i + 16] = np.array([0, 0, 60000, 60000], "<i4").tobytes()   # 60001 x 60001 pixels
    open(path, "wb").write(d)
    with pytest.raises(RuntimeError, match="EXR|dimensions"):
        decode(path, tmp_path)


def test_the_python_loader_reads_exr_through_the_same_decoder(tmp_path):
    from raytracing_amd import loader
    w, h = 48, 24
    r, g, b = picture(w, h, 6)
    path = str(tmp_path / "sky.exr")
    W.write_exr(path, {"R": (W.HALF, r), "G": (W.HALF, g), "B": (W.HALF, b)}, W.PIZ)
    img = loader.load_image_rgb8(path)
    assert img.shape == (h, w, 3) and np.array_equal(img, decode(path, tmp_path))
