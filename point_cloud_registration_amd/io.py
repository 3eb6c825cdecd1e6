"""Minimal PCD (Point Cloud Data v0.7) reader / writer.

The reference's benchmarks load ``data/B-01.pcd`` through ``q3dviewer.utils.cloud_io.load_pcd``
(``benchmark/test_data.py:11,24``), a GUI dependency that is not part of this path.  This module
reads the same files (``DATA ascii``, ``binary`` and ``binary_compressed``) into NumPy so the real
B-01 cloud can be used wherever it is available; ``bench.py`` falls back to the synthetic stand-in
otherwise.
"""

import numpy as np

_NP = {("F", 4): "<f4", ("F", 8): "<f8", ("U", 1): "<u1", ("U", 2): "<u2", ("U", 4): "<u4", ("U", 8): "<u8",
       ("I", 1): "<i1", ("I", 2): "<i2", ("I", 4): "<i4", ("I", 8): "<i8"}


def _lzf_decompress(data, out_len):
    """LZF (the compression of ``DATA binary_compressed``): ``pcr_lzf_decompress`` inside libpcr_hip.so (host C; the pure-Python
    byte loop of rounds 1-5 took minutes per million points)."""
    from . import _capi
    return _capi.lzf_decompress(bytes(data), int(out_len))


def read_pcd(path):
    """Read a PCD file -> structured array with one field per PCD field (multi-count fields as sub-arrays)."""
    with open(path, "rb") as f:
        raw = f.read()
    header, pos = {}, 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        key, _, val = line.partition(" ")
        header[key.upper()] = val.split()
        if key.upper() == "DATA":
            break
    fields = header["FIELDS"]
    sizes = [int(v) for v in header["SIZE"]]
    types = header["TYPE"]
    counts = [int(v) for v in header.get("COUNT", ["1"] * len(fields))]
    npts = int(header["POINTS"][0]) if "POINTS" in header else int(header["WIDTH"][0]) * int(header["HEIGHT"][0])
    dt = np.dtype([(n if n != "_" else f"_pad{i}", _NP[(t, s)], (c,)) if c != 1 else (n if n != "_" else f"_pad{i}", _NP[(t, s)])
                   for i, (n, s, t, c) in enumerate(zip(fields, sizes, types, counts))])
    mode = header["DATA"][0].lower()
    if mode == "ascii":
        cols = np.loadtxt(raw[pos:].decode("ascii").splitlines(), dtype=np.float64, ndmin=2)
        out = np.zeros(npts, dtype=dt)
        c = 0
        for name, cnt in zip(dt.names, counts):
            out[name] = cols[:npts, c] if cnt == 1 else cols[:npts, c:c + cnt]
            c += cnt
        return out
    if mode == "binary":
        return np.frombuffer(raw, dtype=dt, count=npts, offset=pos).copy()
    if mode == "binary_compressed":
        csize, usize = np.frombuffer(raw, dtype="<u4", count=2, offset=pos)
        buf = _lzf_decompress(raw[pos + 8:pos + 8 + int(csize)], int(usize))
        out = np.zeros(npts, dtype=dt)                  # stored field by field (SoA)
        off = 0
        for name, s, t, c in zip(dt.names, sizes, types, counts):
            arr = np.frombuffer(buf, dtype=_NP[(t, s)], count=npts * c, offset=off)
            out[name] = arr if c == 1 else arr.reshape(c, npts).T
            off += npts * c * s
        return out
    raise ValueError(f"unsupported PCD DATA mode {mode!r}")


def load_pcd(path):
    """Drop-in for the loader the reference's harness uses: returns a record array whose ``['xyz']``
    field is the (N, 3) float32 cloud (``benchmark/test_data.py:24-31``: ``map = load_pcd(f); map['xyz']``)."""
    rec = read_pcd(path)
    xyz = np.stack([np.asarray(rec["x"], np.float32), np.asarray(rec["y"], np.float32),
                    np.asarray(rec["z"], np.float32)], axis=1)
    out = np.zeros(xyz.shape[0], dtype=[("xyz", "<f4", (3,))])
    out["xyz"] = xyz
    return out


def save_pcd(path, xyz, binary=True, compressed=False):
    """Write an (N, 3) cloud as PCD v0.7 (x y z float32): ``DATA binary`` / ``ascii`` / (``compressed=True``)
    ``binary_compressed`` -- field by field, LZF."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    n = xyz.shape[0]
    mode = "binary_compressed" if compressed else ("binary" if binary else "ascii")
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\n"
            f"COUNT 1 1 1\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\n"
            f"DATA {mode}\n")
    with open(path, "wb") as f:
        f.write(head.encode("ascii"))
        if compressed:
            from . import _capi
            soa = np.ascontiguousarray(xyz.T).tobytes()
            lz = _capi.lzf_compress(soa)
            f.write(np.array([len(lz), len(soa)], "<u4").tobytes() + lz)
        elif binary:
            f.write(xyz.tobytes())
        else:
            np.savetxt(f, xyz, fmt="%.9g")
