"""Minimal baseline-TIFF reader / writer (uncompressed, strips) for the z-stack splitter.

The reference uses ``tifffile`` (``split_zstack.py:50-51, 64-65``), which is not installed on the system python.
This covers what that path touches: multi-page little/big-endian classic TIFF, 8/16/32-bit unsigned, signed or float
samples, chunky or planar multi-sample pages, and the series shape from tifffile's ``{"shape": [...]}`` or ImageJ's
``images= / channels= / slices=`` ImageDescription.  Compressed or tiled files raise ``ValueError``.
"""
import json
import struct

import numpy as np

_TYPES = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("II", 8), 6: ("b", 1), 7: ("B", 1), 8: ("h", 2),
          9: ("i", 4), 10: ("ii", 8), 11: ("f", 4), 12: ("d", 8), 16: ("Q", 8)}


def _read_ifd(buf, off, bo):
    n = struct.unpack_from(bo + "H", buf, off)[0]
    tags = {}
    for i in range(n):
        tag, typ, cnt = struct.unpack_from(bo + "HHI", buf, off + 2 + 12 * i)
        fmt, size = _TYPES.get(typ, ("B", 1))
        total = size * cnt
        voff = off + 2 + 12 * i + 8
        if total > 4:
            voff = struct.unpack_from(bo + "I", buf, voff)[0]
        if typ == 2:
            val = bytes(buf[voff:voff + cnt]).split(b"\x00")[0].decode("latin-1")
        elif typ in (5, 10):
            raw = struct.unpack_from(bo + fmt[0] * (2 * cnt), buf, voff)
            val = [raw[2 * k] / raw[2 * k + 1] if raw[2 * k + 1] else 0.0 for k in range(cnt)]
        else:
            val = list(struct.unpack_from(bo + fmt * cnt, buf, voff))
        tags[tag] = val
    nxt = struct.unpack_from(bo + "I", buf, off + 2 + 12 * n)[0]
    return tags, nxt


def _page_array(buf, tags, bo):
    if tags.get(259, [1])[0] != 1:
        raise ValueError("compressed TIFF pages are not supported")
    if 322 in tags or 324 in tags:
        raise ValueError("tiled TIFF pages are not supported")
    w, h = tags[256][0], tags[257][0]
    spp = tags.get(277, [1])[0]
    bits = tags.get(258, [1])[0]
    fmtc = tags.get(339, [1])[0]
    kind = {1: "u", 2: "i", 3: "f"}.get(fmtc, "u")
    dt = np.dtype("%s%s%d" % ("<" if bo == "<" else ">", kind, bits // 8))
    data = b"".join(bytes(buf[o:o + c]) for o, c in zip(tags[273], tags[279]))
    arr = np.frombuffer(data, dtype=dt)
    planar = tags.get(284, [1])[0]
    if spp > 1 and planar == 2:
        arr = arr[: spp * h * w].reshape(spp, h, w)
    elif spp > 1:
        arr = arr[: spp * h * w].reshape(h, w, spp)
    else:
        arr = arr[: h * w].reshape(h, w)
    return arr.astype(dt.newbyteorder("="))


def imread(path):
    """Whole series as one array, like ``tifffile.TiffReader(path).asarray()`` (split_zstack.py:50-51)."""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    head = bytes(buf[:4])
    if head[:2] == b"II":
        bo = "<"
    elif head[:2] == b"MM":
        bo = ">"
    else:
        raise ValueError("not a TIFF file: %s" % path)
    if struct.unpack_from(bo + "H", buf, 2)[0] != 42:
        raise ValueError("BigTIFF / unknown TIFF version is not supported")
    off = struct.unpack_from(bo + "I", buf, 4)[0]
    pages, first = [], None
    while off:
        tags, off = _read_ifd(buf, off, bo)
        if first is None:
            first = tags
        pages.append(_page_array(buf, tags, bo))
    out = pages[0] if len(pages) == 1 else np.stack(pages)
    desc = first.get(270, "") if first else ""
    if isinstance(desc, str) and desc.startswith("{"):
        try:
            shape = tuple(json.loads(desc)["shape"])
            if int(np.prod(shape)) == out.size:
                out = out.reshape(shape)
        except (ValueError, KeyError):
            pass
    elif isinstance(desc, str) and desc.startswith("ImageJ="):
        kv = dict(line.split("=", 1) for line in desc.splitlines() if "=" in line)
        c, z, t = int(kv.get("channels", 1)), int(kv.get("slices", 1)), int(kv.get("frames", 1))
        lead = tuple(d for d in (t, z, c) if d > 1)
        if lead and int(np.prod(lead)) == len(pages):
            out = out.reshape(lead + pages[0].shape)
    return out


def imwrite(path, arr):
    """One uncompressed single-strip page per leading index (2-D array: one page), tifffile-style shape JSON in the
    ImageDescription -- what ``TiffWriter(path, bigtiff=False).write(channel)`` leaves (split_zstack.py:64-65)."""
    arr = np.ascontiguousarray(arr)
    if arr.ndim < 2:
        raise ValueError("need at least a 2-D array")
    h, w = arr.shape[-2:]
    pages = arr.reshape((-1, h, w))
    kind = {"u": 1, "i": 2, "f": 3}[arr.dtype.kind]
    bits = arr.dtype.itemsize * 8
    desc = (json.dumps({"shape": list(arr.shape)}) + "\x00").encode()
    le = arr.dtype.newbyteorder("<")
    out = bytearray(b"II*\x00" + struct.pack("<I", 8))
    for pi in range(pages.shape[0]):
        data = pages[pi].astype(le, copy=False).tobytes()
        entries = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, bits), (259, 3, 1, 1), (262, 3, 1, 1)]
        if pi == 0:
            entries.append((270, 2, len(desc), None))
        entries += [(273, 4, 1, None), (277, 3, 1, 1), (278, 4, 1, h), (279, 4, 1, len(data)), (339, 3, 1, kind)]
        ifd_off = len(out)
        ifd_size = 2 + 12 * len(entries) + 4
        desc_off = ifd_off + ifd_size
        data_off = desc_off + (len(desc) if pi == 0 else 0)
        data_off += data_off % 2
        nxt = data_off + len(data)
        nxt += nxt % 2
        blob = bytearray(struct.pack("<H", len(entries)))
        for tag, typ, cnt, val in entries:
            if tag == 270:
                val = desc_off
            elif tag == 273:
                val = data_off
            blob += struct.pack("<HHI", tag, typ, cnt)
            blob += struct.pack("<HH", val, 0) if (typ == 3 and cnt == 1) else struct.pack("<I", val)
        blob += struct.pack("<I", nxt if pi + 1 < pages.shape[0] else 0)
        out += blob
        if pi == 0:
            out += desc
        out += b"\x00" * (data_off - len(out))
        out += data
        out += b"\x00" * (nxt - len(out))
    with open(path, "wb") as f:
        f.write(bytes(out))
