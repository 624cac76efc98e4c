"""The host loader's PNG decoder (texture maps, scene.zig:279-293 goes through zigimg there): images written
here with zlib in every colour type / bit depth / filter the decoder claims, read back through the scene loader."""
import json
import os
import struct
import zlib

import numpy as np
import pytest


def _png(width, height, color_type, depth, rows, palette=None, filters=None):
    """rows: height byte strings of unfiltered scanlines.  filters: per-row filter type to APPLY."""
    bpp = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type] * (depth // 8)
    raw = b""
    prev = bytes(len(rows[0]))
    for y, row in enumerate(rows):
        f = filters[y] if filters else 0
        out = bytearray()
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if f == 0:
                pred = 0
            elif f == 1:
                pred = a
            elif f == 2:
                pred = b
            elif f == 3:
                pred = (a + b) // 2
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out.append((v - pred) & 255)
        raw += bytes([f]) + bytes(out)
        prev = row

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)

    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, color_type, 0, 0, 0))
    if palette is not None:
        png += chunk(b"PLTE", bytes(palette))
    comp = zlib.compress(raw)
    half = len(comp) // 2   # two IDAT chunks: the decoder must concatenate them
    png += chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b"")
    return png


def _load(rtc, tmp_path, png_bytes):
    (tmp_path / "t.png").write_bytes(png_bytes)
    scene = {"camera": {"width": 4, "height": 4, "field-of-view": 1, "from": [0, 0, -5], "to": [0, 0, 0], "up": [0, 1, 0]},
             "lights": [],
             "objects": [{"type": {"sphere": {}},
                          "material": {"pattern": {"type": {"texture-map": {"spherical": {"uv-pattern": {"image": {"file": "t.png"}}}}}}}}]}
    hs = rtc.HostScene(json.dumps(scene), str(tmp_path) + os.sep)
    d = hs.desc
    assert d.n_images == 1 and d.n_texmaps == 1 and d.n_uvs == 1
    w, h = int(hs.array("img_width", 1)[0]), int(hs.array("img_height", 1)[0])
    return np.ctypeslib.as_array(d.img_rgb, shape=(h * w * 3,)).reshape(h, w, 3).copy()


def test_rgb8_all_filters(rtc, tmp_path):
    rng = np.random.default_rng(5)
    px = rng.integers(0, 256, size=(5, 7, 3), dtype=np.uint8)
    rows = [px[y].tobytes() for y in range(5)]
    got = _load(rtc, tmp_path, _png(7, 5, 2, 8, rows, filters=[0, 1, 2, 3, 4]))
    want = px.astype(np.float32) / np.float32(255.0)       # zigimg toF32Color: f32 division
    assert got.dtype == np.float32 and np.array_equal(got, want)


def test_other_colour_types(rtc, tmp_path):
    rng = np.random.default_rng(6)
    grey = rng.integers(0, 256, size=(3, 4), dtype=np.uint8)
    got = _load(rtc, tmp_path, _png(4, 3, 0, 8, [grey[y].tobytes() for y in range(3)], filters=[4, 3, 1]))
    assert np.array_equal(got, np.repeat((grey.astype(np.float32) / np.float32(255))[:, :, None], 3, axis=2))
    rgba = rng.integers(0, 256, size=(3, 4, 4), dtype=np.uint8)
    got = _load(rtc, tmp_path, _png(4, 3, 6, 8, [rgba[y].tobytes() for y in range(3)], filters=[2, 4, 0]))
    assert np.array_equal(got, rgba[:, :, :3].astype(np.float32) / np.float32(255))      # alpha is dropped (canvas.zig:41)
    palette = rng.integers(0, 256, size=(16, 3), dtype=np.uint8)
    idx = rng.integers(0, 16, size=(3, 4), dtype=np.uint8)
    got = _load(rtc, tmp_path, _png(4, 3, 3, 8, [idx[y].tobytes() for y in range(3)], palette=palette.tobytes()))
    assert np.array_equal(got, palette[idx].astype(np.float32) / np.float32(255))
    deep = rng.integers(0, 65536, size=(2, 3, 3), dtype=np.uint16)
    got = _load(rtc, tmp_path, _png(3, 2, 2, 16, [deep[y].astype(">u2").tobytes() for y in range(2)], filters=[1, 4]))
    assert np.array_equal(got, deep.astype(np.float32) / np.float32(65535))


def test_rejections(rtc, tmp_path):
    with pytest.raises(rtc.RtcError) as e:
        _load(rtc, tmp_path, b"GIF89a" + bytes(64))
    assert e.value.name == "Unsupported"
    good = _png(2, 2, 2, 8, [bytes(6), bytes(6)])
    with pytest.raises(rtc.RtcError):
        _load(rtc, tmp_path, good[:40])                     # truncated
    interlaced = bytearray(good)
    interlaced[28] = 1                                      # IHDR interlace byte (CRC not checked by the decoder)
    with pytest.raises(rtc.RtcError) as e:
        _load(rtc, tmp_path, bytes(interlaced))
    assert e.value.name == "Unsupported"
