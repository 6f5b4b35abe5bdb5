"""CPU tests of causal_vae_amd/volume_io.py (3D `*.vessel.tiff` stacks in front of the hot path).  The known-answer files are
assembled byte by byte here, independently of the module's own writer; the preprocessing is checked against the reference's 2D recipe
(causal_cascade/dataset.py:119-135) on stacks where 3D and 2D coincide."""
import struct
import zlib

import numpy as np
import pytest
import torch

from causal_vae_amd.volume_io import TiffError, load_volume, mip, read_tiff_stack, write_tiff_stack


def _classic_tiff(pages, bo=">", rows_per_strip=1, compression=1):
    """uint16 pages -> classic TIFF bytes with `rows_per_strip` rows per strip, IFDs after the data, strip tables out of line."""
    e = bo
    out = bytearray((b"MM" if bo == ">" else b"II") + struct.pack(e + "HI", 42, 0))
    nxt_at = 4
    for img in pages:
        h, w = img.shape
        strips = []
        for y in range(0, h, rows_per_strip):
            raw = img[y:y + rows_per_strip].astype(np.dtype("u2").newbyteorder(bo)).tobytes()
            raw = zlib.compress(raw) if compression == 8 else raw
            strips.append((len(out), len(raw)))
            out += raw
            out += b"\0" * (len(out) % 2)
        n = len(strips)
        tab_off = len(out)
        out += struct.pack(e + "I" * n, *[o for o, _ in strips])
        tab_cnt = len(out)
        out += struct.pack(e + "I" * n, *[c for _, c in strips])
        ifd = len(out)
        ents = [(256, 3, 1, w), (257, 3, 1, h), (258, 3, 1, 16), (259, 3, 1, compression), (262, 3, 1, 1), (273, 4, n, tab_off if n > 1 else strips[0][0]),
                (277, 3, 1, 1), (278, 3, 1, rows_per_strip), (279, 4, n, tab_cnt if n > 1 else strips[0][1]), (339, 3, 1, 1)]
        out += struct.pack(e + "H", len(ents))
        for tag, typ, cnt, val in ents:
            out += struct.pack(e + "HHI", tag, typ, cnt) + (struct.pack(e + "HH", val, 0) if typ == 3 else struct.pack(e + "I", val))
        struct.pack_into(e + "I", out, nxt_at, ifd)
        nxt_at = len(out)
        out += struct.pack(e + "I", 0)
    return bytes(out)


@pytest.mark.parametrize("bo", [">", "<"], ids=["big-endian", "little-endian"])
@pytest.mark.parametrize("rps,comp", [(1, 1), (2, 1), (3, 8)])
def test_known_answer_classic_tiff(tmp_path, bo, rps, comp):
    pages = [np.array([[1, 2, 3], [400, 500, 65535]], dtype=np.uint16) + k for k in (0, 7, 20)]
    pages[2][1, 2] = 9
    p = tmp_path / "plate-25250-01-504002.vessel.tiff"
    p.write_bytes(_classic_tiff(pages, bo, rps, comp))
    vol = read_tiff_stack(str(p))
    assert vol.shape == (3, 2, 3) and vol.dtype == np.uint16
    assert np.array_equal(vol, np.stack(pages))
    assert np.array_equal(read_tiff_stack(str(p), max_pages=2), np.stack(pages[:2]))
    assert np.array_equal(mip(vol), np.maximum(np.maximum(pages[0], pages[1]), pages[2]))       # dataset.py:103-109


def test_known_answer_bigtiff_float32(tmp_path):
    img = np.arange(12, dtype=np.float32).reshape(3, 4) * 0.5 - 1.0
    raw = img.astype("<f4").tobytes()
    out = bytearray(b"II" + struct.pack("<HHHQ", 43, 8, 0, 0))
    doff = len(out)
    out += raw
    ifd = len(out)
    ents = [(256, 4, 1, 4), (257, 4, 1, 3), (258, 3, 1, 32), (259, 3, 1, 1), (273, 16, 1, doff), (277, 3, 1, 1), (278, 4, 1, 3), (279, 16, 1, len(raw)), (339, 3, 1, 3)]
    out += struct.pack("<Q", len(ents))
    for tag, typ, cnt, val in ents:
        out += struct.pack("<HHQ", tag, typ, cnt) + struct.pack({3: "<H6x", 4: "<I4x", 16: "<Q"}[typ], val)
    out += struct.pack("<Q", 0)
    struct.pack_into("<Q", out, 8, ifd)
    p = tmp_path / "a.tiff"
    p.write_bytes(bytes(out))
    vol = read_tiff_stack(str(p))
    assert vol.shape == (1, 3, 4) and vol.dtype == np.float32 and np.array_equal(vol[0], img)


def test_tiled_page(tmp_path):
    g = np.random.default_rng(0)
    img = g.integers(0, 4000, size=(20, 37), dtype=np.uint16)
    tw = tl = 16
    across, down = 3, 2
    out = bytearray(b"II" + struct.pack("<HI", 42, 0))
    offs, cnts = [], []
    for ty in range(down):
        for tx in range(across):
            t = np.zeros((tl, tw), np.uint16)
            blk = img[ty * tl:(ty + 1) * tl, tx * tw:(tx + 1) * tw]
            t[:blk.shape[0], :blk.shape[1]] = blk
            offs.append(len(out)); cnts.append(tl * tw * 2)
            out += t.astype("<u2").tobytes()
    to, tc = len(out), len(out) + 4 * len(offs)
    out += struct.pack("<" + "I" * len(offs), *offs) + struct.pack("<" + "I" * len(cnts), *cnts)
    ifd = len(out)
    ents = [(256, 3, 1, 37), (257, 3, 1, 20), (258, 3, 1, 16), (259, 3, 1, 1), (277, 3, 1, 1), (322, 3, 1, tw), (323, 3, 1, tl), (324, 4, 6, to), (325, 4, 6, tc)]
    out += struct.pack("<H", len(ents))
    for tag, typ, cnt, val in ents:
        out += struct.pack("<HHI", tag, typ, cnt) + (struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val))
    out += struct.pack("<I", 0)
    struct.pack_into("<I", out, 4, ifd)
    p = tmp_path / "t.tiff"
    p.write_bytes(bytes(out))
    assert np.array_equal(read_tiff_stack(str(p))[0], img)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16, np.float32])
@pytest.mark.parametrize("compress", [False, True])
def test_write_read_round_trip(tmp_path, dtype, compress):
    g = np.random.default_rng(1)
    vol = (g.random((5, 9, 14)) * 200).astype(dtype)
    p = tmp_path / "rt.vessel.tiff"
    write_tiff_stack(str(p), vol, compress=compress)
    back = read_tiff_stack(str(p))
    assert back.dtype == vol.dtype and np.array_equal(back, vol)


def test_unsupported_files_fail_loudly(tmp_path):
    p = tmp_path / "x.tiff"
    p.write_bytes(b"not a tiff at all")
    with pytest.raises(TiffError):
        read_tiff_stack(str(p))
    pages = [np.zeros((2, 3), np.uint16)]
    p.write_bytes(_classic_tiff(pages, "<", 2, compression=5))                # LZW
    with pytest.raises(TiffError, match="compression 5"):
        read_tiff_stack(str(p))
    raw = bytearray(_classic_tiff(pages, "<", 2))
    i = raw.index(struct.pack("<HHI", 277, 3, 1))                             # SamplesPerPixel 1 -> 3
    struct.pack_into("<H", raw, i + 8, 3)
    p.write_bytes(bytes(raw))
    with pytest.raises(TiffError, match="samples per pixel"):
        read_tiff_stack(str(p))
    with pytest.raises(TiffError):
        load_volume(np.zeros((4, 4)))


def test_load_volume_follows_the_reference_recipe(tmp_path):
    """A stack of identical pages at the target size: resize is the identity, so every slice must be the reference's 2D pipeline applied to the page
    (dataset.py:119-135: clip at 3000, drop 100 rows top and bottom when H > 200, float32, z-score with std + 1e-5)."""
    g = np.random.default_rng(2)
    page = g.integers(0, 6000, size=(232, 24), dtype=np.uint16)
    vol = np.repeat(page[None], 6, axis=0)
    p = tmp_path / "s.vessel.tiff"
    write_tiff_stack(str(p), vol)
    x = load_volume(str(p), size=(6, 32, 24))
    assert x.shape == (1, 6, 32, 24) and x.dtype == torch.float32
    img = np.clip(page, page.min(), 3000)[100:-100, :].astype("float32")      # the reference lines, verbatim arithmetic
    ref = (img - img.mean()) / (img.std() + 1e-5)
    for d in range(6):
        np.testing.assert_allclose(x[0, d].numpy(), ref, rtol=1e-5, atol=1e-5)
    # a real resize: shape, finite, z-scored; shrinking by >= 2x goes through the box filter (a constant volume stays constant before the z-score)
    big = (g.random((40, 300, 64)) * 5000).astype(np.float32)
    y = load_volume(big, size=(16, 32, 32))
    assert y.shape == (1, 16, 32, 32) and bool(torch.isfinite(y).all())
    assert abs(float(y.mean())) < 1e-4 and abs(float(y.std(unbiased=False)) - 1.0) < 1e-3
    z = load_volume(np.full((8, 16, 16), 7.0, np.float32), size=(4, 8, 8))
    assert float(z.abs().max()) < 1e-3                                         # (v - mean) / (0 + 1e-5) with v == mean
