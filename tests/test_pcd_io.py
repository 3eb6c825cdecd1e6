"""PCD reader (the I/O either side of the path; the reference uses q3dviewer's load_pcd)."""

import numpy as np

from point_cloud_registration_amd.io import load_pcd, read_pcd, save_pcd, _lzf_decompress


def test_roundtrip_ascii_and_binary(tmp_path):
    pts = np.random.default_rng(0).normal(size=(257, 3)).astype(np.float32) * 30
    for binary in (True, False):
        p = tmp_path / f"c_{binary}.pcd"
        save_pcd(str(p), pts, binary=binary)
        got = load_pcd(str(p))["xyz"]
        assert got.dtype == np.float32 and np.array_equal(got, pts)


def test_binary_with_extra_fields_and_compressed(tmp_path):
    n = 100
    rng = np.random.default_rng(1)
    x, y, z = (rng.normal(size=n).astype("<f4") for _ in range(3))
    inten = rng.integers(0, 255, n).astype("<u1")
    head = (f"VERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 1\nTYPE F F F U\nCOUNT 1 1 1 1\nWIDTH {n}\nHEIGHT 1\n"
            f"POINTS {n}\n")
    rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<u1")])
    rec["x"], rec["y"], rec["z"], rec["intensity"] = x, y, z, inten
    p = tmp_path / "b.pcd"
    p.write_bytes(head.encode() + b"DATA binary\n" + rec.tobytes())
    r = read_pcd(str(p))
    assert np.array_equal(r["x"], x) and np.array_equal(r["intensity"], inten)
    # binary_compressed: SoA payload, LZF with literal runs only (a valid LZF stream)
    soa = x.tobytes() + y.tobytes() + z.tobytes() + inten.tobytes()
    lzf = b"".join(bytes([len(soa[i:i + 32]) - 1]) + soa[i:i + 32] for i in range(0, len(soa), 32))
    p2 = tmp_path / "c.pcd"
    p2.write_bytes(head.encode() + b"DATA binary_compressed\n" + np.array([len(lzf), len(soa)], "<u4").tobytes() + lzf)
    r2 = load_pcd(str(p2))["xyz"]
    assert np.array_equal(r2, np.stack([x, y, z], 1))


def test_lzf_back_references():
    # "abcabcabcabc": literal 'abc' + back reference (offset 3, length 9, overlapping)
    stream = bytes([2]) + b"abc" + bytes([(7 << 5) | 0, 0, 2])
    assert _lzf_decompress(stream, 12) == b"abcabcabcabc"


def test_binary_compressed_million_points_roundtrip(tmp_path):
    """a B-01-sized ``binary_compressed`` file (1.06 M points, LZF inside libpcr_hip.so) loads in well under
    a second -- the pure-Python decoder of rounds 1-5 needed minutes -- and comes back bit for bit."""
    import time
    from point_cloud_registration_amd.synthetic import street
    pts = street(1_060_000, seed=0)
    p = tmp_path / "big.pcd"
    save_pcd(str(p), pts, compressed=True)
    t0 = time.perf_counter()
    got = load_pcd(str(p))["xyz"]
    dt = time.perf_counter() - t0
    assert got.dtype == np.float32 and np.array_equal(got, pts)
    assert dt < 1.0, dt
    # a compressible payload (repeated coordinates: long, overlapping back references) through the same codec
    rep = np.tile(pts[:1000], (300, 1))
    save_pcd(str(p), rep, compressed=True)
    assert p.stat().st_size < rep.nbytes // 4
    assert np.array_equal(load_pcd(str(p))["xyz"], rep)


def test_lzf_rejects_corrupt_streams():
    import pytest
    with pytest.raises(ValueError):
        _lzf_decompress(bytes([5]) + b"ab", 6)                      # literal run longer than the input
    with pytest.raises(ValueError):
        _lzf_decompress(bytes([(1 << 5) | 0, 9]), 3)                 # back reference before the start of the output
    with pytest.raises(ValueError):
        _lzf_decompress(bytes([2]) + b"abc", 5)                      # shorter than the header promised
