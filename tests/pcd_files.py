"""PCD writers for the tests (test data only): ASCII, binary and binary_compressed `x y z label` clouds the way PCL
lays them out, with a small LZF compressor (liblzf's format: literal runs and back references) so that the
reader's decompressor sees real back references, overlapping copies and long matches."""
from __future__ import annotations

import struct

import numpy as np


def lzf_compress(data: bytes, literal_only: bool = False) -> bytes:
    """Greedy LZF: 3-byte hash -> last position; matches of 3..264 bytes within 8192 bytes."""
    n = len(data)
    out = bytearray()
    lit = bytearray()

    def flush():
        i = 0
        while i < len(lit):
            run = lit[i:i + 32]
            out.append(len(run) - 1)
            out.extend(run)
            i += 32
        lit.clear()

    table = {}
    i = 0
    while i < n:
        m_len = 0
        if not literal_only and i + 2 < n:
            key = data[i:i + 3]
            j = table.get(key)
            table[key] = i
            if j is not None and 0 < i - j <= 8192:
                m_len = 3
                while i + m_len < n and m_len < 264 and data[j + m_len] == data[i + m_len]:
                    m_len += 1
        if m_len >= 3:
            flush()
            dist = i - j - 1
            ln = m_len - 2
            if ln < 7:
                out.append((ln << 5) | (dist >> 8))
            else:
                out.append((7 << 5) | (dist >> 8))
                out.append(ln - 7)
            out.append(dist & 0xFF)
            i += m_len
        else:
            lit.append(data[i])
            i += 1
    flush()
    return bytes(out)


def lzf_decompress(data: bytes, out_len: int) -> bytes:
    out = bytearray()
    i = 0
    while i < len(data):
        ctrl = data[i]
        i += 1
        if ctrl < 32:
            out += data[i:i + ctrl + 1]
            i += ctrl + 1
        else:
            ln = ctrl >> 5
            if ln == 7:
                ln += data[i]
                i += 1
            dist = ((ctrl & 0x1F) << 8 | data[i]) + 1
            i += 1
            for _ in range(ln + 2):
                out.append(out[-dist])
    assert len(out) == out_len
    return bytes(out)


def header(n, data_kind, fields="x y z label", sizes="4 4 4 4", types="F F F U", counts="1 1 1 1", width=None, height=1):
    width = n if width is None else width
    return (f"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS {fields}\nSIZE {sizes}\nTYPE {types}\n"
            f"COUNT {counts}\nWIDTH {width}\nHEIGHT {height}\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA {data_kind}\n").encode()


def write_pcd(path, xyz, labels, kind="ascii", literal_only=False, width=None, height=1):
    """kind: ascii | binary | binary_compressed"""
    xyz = np.asarray(xyz, dtype=np.float32)
    labels = np.asarray(labels, dtype=np.uint32)
    n = len(xyz)
    with open(path, "wb") as f:
        f.write(header(n, kind, width=width, height=height))
        if kind == "binary":
            rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("l", "<u4")])
            rec["x"], rec["y"], rec["z"], rec["l"] = xyz[:, 0], xyz[:, 1], xyz[:, 2], labels
            f.write(rec.tobytes())
        elif kind == "binary_compressed":
            # PCL: the fields one after the other (all x, all y, all z, all labels), LZF-compressed as one block
            planes = xyz[:, 0].tobytes() + xyz[:, 1].tobytes() + xyz[:, 2].tobytes() + labels.tobytes()
            comp = lzf_compress(planes, literal_only)
            f.write(struct.pack("<II", len(comp), len(planes)))
            f.write(comp)
        else:
            for p, l in zip(xyz, labels):
                f.write(f"{p[0]:.9g} {p[1]:.9g} {p[2]:.9g} {int(l)}\n".encode())
