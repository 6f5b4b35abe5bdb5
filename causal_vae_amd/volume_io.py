"""3D `*.vessel.tiff` stacks -> network input volumes: the step in front of the hot path for the 3D lift.

The reference never feeds the stack itself: `causal_cascade/dataset.py:103-109` (`load_mip_safe`) reads the file page by page and keeps the
maximum-intensity projection, which `__getitem__` (:111-135) clips at 3000, crops by 100 rows top and bottom, resizes and z-scores.  The 3D model
takes the stack, so this module provides (a) a dependency-free reader for the TIFF flavours such stacks come in — `tifffile` is not part of this
image, and nothing else of it is needed: classic TIFF and BigTIFF, either byte order, one IFD per page, one sample per pixel of 8 / 16 / 32 / 64
bits (unsigned, signed or IEEE float), strips or tiles stored raw or deflate-compressed; anything else raises — (b) `mip()`, the reference's
projection, so that a consumer can check `mip(read_tiff_stack(p))` against the 2D pipeline, and (c) `load_volume()`, the 3D restatement of the
reference's per-image preprocessing: clip at `clip_max`, the same 100-row crop, resize to the network's grid, z-score.  Host-side numpy / torch-CPU
code: I/O, not part of the timed path (the boundary takes device tensors).
"""
import struct
import zlib

import numpy as np
import torch

_TYPES = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("II", 8), 6: ("b", 1), 7: ("B", 1), 8: ("h", 2), 9: ("i", 4), 10: ("ii", 8), 11: ("f", 4),
          12: ("d", 8), 16: ("Q", 8), 17: ("q", 8), 18: ("Q", 8)}
_COMPRESSION_NONE, _COMPRESSION_DEFLATE = (1,), (8, 32946)


class TiffError(ValueError):
    pass


def _ifd_entries(buf, off, bo, big):
    """Tags of the IFD at `off` -> ({tag: tuple of values}, offset of the next IFD)."""
    if big:
        (n,) = struct.unpack_from(bo + "Q", buf, off)
        pos, esz, cfmt, csz, inline = off + 8, 20, "Q", 8, 8
    else:
        (n,) = struct.unpack_from(bo + "H", buf, off)
        pos, esz, cfmt, csz, inline = off + 2, 12, "I", 4, 4
    tags = {}
    for i in range(n):
        e = pos + i * esz
        tag, typ = struct.unpack_from(bo + "HH", buf, e)
        (count,) = struct.unpack_from(bo + cfmt, buf, e + 4)
        if typ not in _TYPES:
            continue
        fmt, size = _TYPES[typ]
        nbytes = size * count
        voff = e + 4 + csz
        if nbytes > inline:
            (voff,) = struct.unpack_from(bo + cfmt, buf, voff)
        if voff + nbytes > len(buf):
            raise TiffError(f"tag {tag}: value runs past the end of the file")
        if typ == 2:
            tags[tag] = (bytes(buf[voff:voff + nbytes]),)
        else:
            tags[tag] = struct.unpack_from(bo + fmt * count, buf, voff)
    (nxt,) = struct.unpack_from(bo + cfmt, buf, pos + n * esz)
    return tags, nxt


def _page(buf, tags, bo):
    g = lambda t, d=None: tags.get(t, (d,))[0]
    w, h = g(256), g(257)
    if w is None or h is None:
        raise TiffError("page without ImageWidth / ImageLength")
    spp, bits, fmt, comp = g(277, 1), g(258, 1), g(339, 1), g(259, 1)
    if spp != 1 or g(284, 1) != 1:
        raise TiffError(f"{spp} samples per pixel: a vessel stack has one")
    if comp not in _COMPRESSION_NONE + _COMPRESSION_DEFLATE:
        raise TiffError(f"compression {comp} is not supported (raw and deflate are)")
    if g(317, 1) != 1:
        raise TiffError("a predictor is set: not supported")
    kind = {1: "u", 2: "i", 3: "f"}.get(fmt)
    if kind is None or bits not in (8, 16, 32, 64) or (kind == "f" and bits < 32):
        raise TiffError(f"sample format {fmt} with {bits} bits is not supported")
    dt = np.dtype(f"{'<' if bo == '<' else '>'}{kind}{bits // 8}")
    inflate = (lambda b: zlib.decompress(b)) if comp in _COMPRESSION_DEFLATE else (lambda b: b)
    out = np.empty((h, w), dtype=dt.newbyteorder("="))

    def chunk(off, n):
        if off + n > len(buf):
            raise TiffError("strip / tile runs past the end of the file")
        return inflate(bytes(buf[off:off + n]))
    if 322 in tags:                                          # tiled
        tw, tl = g(322), g(323)
        offs, cnts = tags[324], tags[325]
        across = (w + tw - 1) // tw
        for i, (o, n) in enumerate(zip(offs, cnts)):
            t = np.frombuffer(chunk(o, n), dtype=dt, count=tw * tl).reshape(tl, tw)
            y0, x0 = (i // across) * tl, (i % across) * tw
            out[y0:y0 + tl, x0:x0 + tw] = t[:min(tl, h - y0), :min(tw, w - x0)]
    else:
        rps = min(g(278, h), h)
        offs, cnts = tags.get(273), tags.get(279)
        if offs is None or cnts is None:
            raise TiffError("page without StripOffsets / StripByteCounts")
        for i, (o, n) in enumerate(zip(offs, cnts)):
            y0 = i * rps
            rows = min(rps, h - y0)
            out[y0:y0 + rows] = np.frombuffer(chunk(o, n), dtype=dt, count=rows * w).reshape(rows, w)
    return out


def read_tiff_stack(path, max_pages=None):
    """All pages of a TIFF as one array [D, H, W] in the file's sample type (native byte order).  Pages must agree in size and type."""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    if len(buf) < 8 or bytes(buf[:2]) not in (b"II", b"MM"):
        raise TiffError(f"{path}: not a TIFF file")
    bo = "<" if bytes(buf[:2]) == b"II" else ">"
    (magic,) = struct.unpack_from(bo + "H", buf, 2)
    if magic == 42:
        big = False
        (off,) = struct.unpack_from(bo + "I", buf, 4)
    elif magic == 43:
        big = True
        osz, zero = struct.unpack_from(bo + "HH", buf, 4)
        if osz != 8 or zero != 0:
            raise TiffError(f"{path}: malformed BigTIFF header")
        (off,) = struct.unpack_from(bo + "Q", buf, 8)
    else:
        raise TiffError(f"{path}: not a TIFF file (magic {magic})")
    pages, seen = [], set()
    while off and (max_pages is None or len(pages) < max_pages):
        if off in seen or off >= len(buf):
            raise TiffError(f"{path}: broken IFD chain")
        seen.add(off)
        tags, off = _ifd_entries(buf, off, bo, big)
        if tags.get(254, (0,))[0] & 1:                       # reduced-resolution copy of another page (pyramids): not a slice
            continue
        pages.append(_page(buf, tags, bo))
    if not pages:
        raise TiffError(f"{path}: no image pages")
    if any(p.shape != pages[0].shape or p.dtype != pages[0].dtype for p in pages):
        raise TiffError(f"{path}: pages differ in size or sample type")
    return np.stack(pages)


def write_tiff_stack(path, vol, compress=False):
    """[D, H, W] array -> multi-page little-endian classic TIFF, one strip per page (raw or deflate).  For fixtures and exports."""
    vol = np.ascontiguousarray(vol)
    if vol.ndim != 3:
        raise TiffError("write_tiff_stack expects [D, H, W]")
    kind = {"u": 1, "i": 2, "f": 3}.get(vol.dtype.kind)
    if kind is None or vol.dtype.itemsize not in (1, 2, 4, 8):
        raise TiffError(f"dtype {vol.dtype} cannot be stored")
    le = vol.astype(vol.dtype.newbyteorder("<"), copy=False)
    out = bytearray(b"II" + struct.pack("<HI", 42, 0))
    prev_next = 4                                            # where the offset of the next IFD gets patched in
    D, H, W = vol.shape
    for d in range(D):
        data = le[d].tobytes()
        if compress:
            data = zlib.compress(data)
        if len(out) % 2:
            out += b"\0"
        doff = len(out)
        out += data
        if len(out) % 2:
            out += b"\0"
        ifd = len(out)
        ents = [(256, 4, W), (257, 4, H), (258, 3, vol.dtype.itemsize * 8), (259, 3, 8 if compress else 1), (262, 3, 1), (273, 4, doff), (277, 3, 1), (278, 4, H),
                (279, 4, len(data)), (339, 3, kind)]
        out += struct.pack("<H", len(ents))
        for tag, typ, val in ents:
            out += struct.pack("<HHI", tag, typ, 1) + (struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val))
        struct.pack_into("<I", out, prev_next, ifd)
        prev_next = len(out)
        out += struct.pack("<I", 0)
    with open(path, "wb") as f:
        f.write(out)


def mip(vol):
    """The reference's projection (causal_cascade/dataset.py:103-109): element-wise maximum over the pages."""
    return np.asarray(vol).max(axis=0)


def load_volume(path_or_array, size=(128, 128, 128), clip_max=3000.0, crop_rows=100):
    """3D restatement of causal_cascade/dataset.py:119-135 for the stack itself: clip to [min, clip_max] (:120), drop `crop_rows` rows at the top
    and bottom of every slice when it has more than 2 * crop_rows (:121-122), resize to `size` (trilinear, after an integer box-filter where an axis
    shrinks by 2x or more — the 3D counterpart of skimage's anti_aliasing=True at :124), float32, z-score with std + 1e-5 (:131-135).
    Returns a CPU tensor [1, D, H, W] (the batch contract of SURVEY.md §2 row 4 with a depth axis)."""
    vol = read_tiff_stack(path_or_array) if isinstance(path_or_array, (str, bytes)) or hasattr(path_or_array, "__fspath__") else np.asarray(path_or_array)
    if vol.ndim != 3:
        raise TiffError("load_volume expects a [D, H, W] stack")
    v = torch.from_numpy(np.ascontiguousarray(vol).astype(np.float32))
    v = v.clamp(max=float(clip_max))                         # np.clip(image, image.min(), 3000): the lower bound is the minimum itself
    if v.shape[1] > 2 * crop_rows:
        v = v[:, crop_rows:-crop_rows, :]
    v = v[None, None]
    pool = [max(1, int(s // t)) for s, t in zip(v.shape[2:], size)]
    if any(p > 1 for p in pool):
        v = torch.nn.functional.avg_pool3d(v, kernel_size=pool, stride=pool, ceil_mode=True, count_include_pad=False)
    if tuple(v.shape[2:]) != tuple(size):
        v = torch.nn.functional.interpolate(v, size=tuple(size), mode="trilinear", align_corners=False)
    v = v[0]
    return (v - v.mean()) / (v.std(unbiased=False) + 1e-5)
