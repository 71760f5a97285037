"""Minimal PNG codec (8-bit, non-interlaced, colour types 0/2/4/6) on zlib + numpy.

Replaces the `image` crate the reference uses for `ggx_lut.png` (src/main.rs:300-316 via
`load_image_from_bytes`) and is the frame-out writer.  Colour-management chunks (iCCP, gAMA) are
ignored exactly as the reference ignores them: the bytes go to the GPU as R8G8B8A8_UNORM.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"
_CHANNELS = {0: 1, 2: 3, 4: 2, 6: 4}


def _unfilter(raw: np.ndarray, height: int, stride: int, bpp: int) -> np.ndarray:
    out = np.zeros((height, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.uint8)
    pos = 0
    for y in range(height):
        ftype = int(raw[pos])
        line = raw[pos + 1 : pos + 1 + stride].astype(np.uint8)
        pos += 1 + stride
        if ftype == 0:
            cur = line
        elif ftype == 1:  # Sub: prefix sums per byte lane, mod 256
            cur = line.reshape(-1, bpp).astype(np.uint32).cumsum(axis=0).astype(np.uint8).reshape(-1)
        elif ftype == 2:  # Up
            cur = (line.astype(np.uint16) + prev).astype(np.uint8)
        else:  # Average / Paeth are sequential along the row
            cur = np.zeros(stride, dtype=np.uint8)
            ln = line.tolist()
            pv = prev.tolist()
            c = [0] * stride
            for i in range(stride):
                a = c[i - bpp] if i >= bpp else 0
                b = pv[i]
                if ftype == 3:
                    pred = (a + b) >> 1
                else:
                    cc = pv[i - bpp] if i >= bpp else 0
                    p = a + b - cc
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - cc)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else cc)
                c[i] = (ln[i] + pred) & 0xFF
            cur = np.array(c, dtype=np.uint8)
        out[y] = cur
        prev = cur
    return out


def decode_png(data: bytes, what: str = "<bytes>") -> np.ndarray:
    """Returns an (H, W, C) uint8 array (C = 1, 2, 3 or 4) from the bytes of an 8-bit non-interlaced PNG."""
    if data[:8] != _SIG:
        raise ValueError(f"{what}: not a PNG")
    pos = 8
    idat = []
    width = height = ctype = None
    while pos < len(data):
        (length,) = struct.unpack(">I", data[pos : pos + 4])
        kind = data[pos + 4 : pos + 8]
        body = data[pos + 8 : pos + 8 + length]
        pos += 12 + length
        if kind == b"IHDR":
            width, height, depth, ctype, _comp, _filt, interlace = struct.unpack(">IIBBBBB", body)
            if depth != 8 or interlace != 0 or ctype not in _CHANNELS:
                raise ValueError(f"{what}: unsupported PNG (depth {depth}, colour type {ctype}, interlace {interlace})")
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    ch = _CHANNELS[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8)
    img = _unfilter(raw, height, width * ch, ch)
    return img.reshape(height, width, ch)


def read_png(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        return decode_png(f.read(), path)


def _to_rgba8(img: np.ndarray) -> np.ndarray:
    h, w, c = img.shape
    if c == 4:
        return np.ascontiguousarray(img)
    out = np.full((h, w, 4), 255, dtype=np.uint8)
    if c == 3:
        out[..., :3] = img
    elif c == 1:
        out[..., :3] = img
    else:  # grey + alpha
        out[..., :3] = img[..., :1]
        out[..., 3] = img[..., 1]
    return out


def read_png_rgba8(path: str) -> np.ndarray:
    """(H, W, 4) uint8; RGB gets alpha 255 (src/model_loading.rs:36-52 does the same widening)."""
    return _to_rgba8(read_png(path))


def read_png_rgba8_bytes(data: bytes) -> np.ndarray:
    return _to_rgba8(decode_png(data))


def write_png(path: str, img: np.ndarray) -> None:
    """Writes an (H, W, 3|4) uint8 array as an 8-bit PNG (filter 0)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, c = img.shape
    ctype = {3: 2, 4: 6}[c]
    raw = np.zeros((h, 1 + w * c), dtype=np.uint8)
    raw[:, 1:] = img.reshape(h, w * c)

    def chunk(kind: bytes, body: bytes) -> bytes:
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(_SIG)
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)))
        f.write(chunk(b"IEND", b""))
