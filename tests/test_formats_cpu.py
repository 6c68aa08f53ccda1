"""On-disk formats either side of the path: KITTI .bin loader and the matched-pair PCD
cache.  Host code of libquatro_hip.so; checked against independent numpy / pure-Python restatements of the formats
(the reference's loader, examples/run_global_registration.cpp:377-402; PCL's PCD v0.7 as FPFHManager::saveFeaturePair
writes it, include/fpfh_manager.hpp:179-232)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from quatro_amd import lib as ql


def test_kitti_bin_loader(tmp_path):
    rng = np.random.default_rng(0)
    a = rng.standard_normal((1000, 4)).astype(np.float32)
    p = tmp_path / "000000.bin"
    a.tofile(p)
    assert np.array_equal(ql.read_kitti_bin(str(p)), a)
    assert np.array_equal(ql.read_kitti_bin(str(p), max_points=10), a[:10])  # the demo's cap (1e6 floats = 250k points)
    # a trailing partial record is dropped (integer division in the demo)
    with open(p, "ab") as f:
        f.write(b"\x00" * 9)
    assert ql.read_kitti_bin(str(p)).shape == (1000, 4)
    (tmp_path / "empty.bin").write_bytes(b"")
    assert ql.read_kitti_bin(str(tmp_path / "empty.bin")).shape == (0, 4)
    with pytest.raises(OSError):
        ql.read_kitti_bin(str(tmp_path / "missing.bin"))


def _expected_ascii_pcd(xyz):
    n = xyz.shape[0]
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n"
            f"WIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA ascii\n")
    fmt = lambda v: "nan" if np.isnan(v) else "%.8g" % float(v)
    return head + "".join(" ".join(fmt(v) for v in row) + "\n" for row in xyz)


def test_pcd_ascii_writer_is_pcl_default_layout(tmp_path):
    rng = np.random.default_rng(1)
    xyz = (rng.standard_normal((257, 3)) * 30).astype(np.float32)
    xyz[3] = [0.0, -0.0, 1e-20]
    xyz[4] = [np.nan, 1.0, 123456792.0]
    p = tmp_path / "a.pcd"
    ql.write_pcd_xyz(str(p), xyz)
    assert p.read_text() == _expected_ascii_pcd(xyz)
    back = ql.read_pcd_xyz(str(p))
    assert back.shape == (257, 4) and np.all(back[:, 3] == 0)
    # 8 significant digits: the cache is lossy at the last binary32 digit, exactly like the reference's
    want = np.array([[np.float32(float("%.8g" % float(v))) if not np.isnan(v) else np.nan for v in row] for row in xyz],
                    dtype=np.float32)
    assert np.array_equal(back[:, :3], want, equal_nan=True)
    assert np.allclose(back[:, :3], xyz, rtol=1e-7, atol=0, equal_nan=True)


def _lzf_literals(data: bytes) -> bytes:
    out = bytearray()
    for i in range(0, len(data), 32):
        chunk = data[i:i + 32]
        out.append(len(chunk) - 1)
        out += chunk
    return bytes(out)


def _lzf_greedy(data: bytes) -> bytes:
    """A small LZF encoder (literal runs + back references, including long and overlapping ones)."""
    out, lit, i, n = bytearray(), bytearray(), 0, len(data)

    def flush():
        nonlocal lit
        for k in range(0, len(lit), 32):
            c = lit[k:k + 32]
            out.append(len(c) - 1)
            out.extend(c)
        lit = bytearray()
    while i < n:
        best_len, best_dist = 0, 0
        for dist in range(1, min(i, 8191) + 1):
            if data[i - dist] != data[i]:
                continue
            ln = 0
            while i + ln < n and ln < 264 and data[i + ln - dist] == data[i + ln]:
                ln += 1
            if ln > best_len:
                best_len, best_dist = ln, dist
            if dist > 64 and best_len >= 8:
                break
        if best_len >= 3:
            flush()
            ln, d = best_len - 2, best_dist - 1
            if ln < 7:
                out.append((ln << 5) | (d >> 8))
            else:
                out.append((7 << 5) | (d >> 8))
                out.append(ln - 7)
            out.append(d & 0xff)
            i += best_len
        else:
            lit.append(data[i])
            i += 1
    flush()
    return bytes(out)


@pytest.mark.parametrize("kind", ["binary", "binary_compressed_literal", "binary_compressed_refs"])
def test_pcd_reader_binary_kinds_and_field_order(tmp_path, kind):
    """Files as other PCL tools write them: extra fields, x/y/z not first, binary and LZF-compressed payloads."""
    rng = np.random.default_rng(2)
    n = 300
    xyz = (rng.standard_normal((n, 3)) * 10).astype(np.float32)
    xyz[:100] = xyz[0]  # repetition for the back-reference path
    inten = rng.uniform(0, 1, n).astype(np.float32)
    ring = rng.integers(0, 64, n).astype(np.uint16)
    head = (f"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS intensity x y z ring\nSIZE 4 4 4 4 2\n"
            f"TYPE F F F F U\nCOUNT 1 1 1 1 1\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\n")
    p = tmp_path / "b.pcd"
    if kind == "binary":
        rec = b"".join(struct.pack("<ffffH", inten[i], *xyz[i], ring[i]) for i in range(n))
        p.write_bytes((head + "DATA binary\n").encode() + rec)
    else:
        soa = inten.tobytes() + xyz[:, 0].tobytes() + xyz[:, 1].tobytes() + xyz[:, 2].tobytes() + ring.tobytes()
        comp = _lzf_literals(soa) if kind.endswith("literal") else _lzf_greedy(soa)
        if kind.endswith("refs"):
            assert len(comp) < len(soa)
        p.write_bytes((head + "DATA binary_compressed\n").encode() + struct.pack("<II", len(comp), len(soa)) + comp)
    back = ql.read_pcd_xyz(str(p))
    assert np.array_equal(back[:, :3], xyz)


def test_pcd_binary_round_trip_and_errors(tmp_path):
    rng = np.random.default_rng(3)
    xyz = rng.standard_normal((50, 3)).astype(np.float32)
    p = tmp_path / "c.pcd"
    ql.write_pcd_xyz(str(p), xyz, binary=True)
    assert np.array_equal(ql.read_pcd_xyz(str(p))[:, :3], xyz)  # lossless
    ql.write_pcd_xyz(str(p), np.zeros((0, 3), dtype=np.float32))
    assert ql.read_pcd_xyz(str(p)).shape == (0, 4)
    with pytest.raises(OSError):
        ql.read_pcd_xyz(str(tmp_path / "missing.pcd"))
    (tmp_path / "bad.pcd").write_text("VERSION 0.7\nFIELDS a b c\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\n"
                                      "POINTS 1\nDATA ascii\n1 2 3\n")
    with pytest.raises(OSError):  # no x y z fields
        ql.read_pcd_xyz(str(tmp_path / "bad.pcd"))
    (tmp_path / "short.pcd").write_text("VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 2\n"
                                        "HEIGHT 1\nPOINTS 2\nDATA ascii\n1 2 3\n")
    with pytest.raises(OSError):  # truncated payload
        ql.read_pcd_xyz(str(tmp_path / "short.pcd"))


def test_feature_pair_cache_python_and_cpp_agree(tmp_path):
    """FPFHManager::saveFeaturePair / loadFeaturePair: "%06d_to_%06d.pcd", source half then target half; the C++
    drop-in header and the Python mirror write byte-identical files and read each other's."""
    from quatro_amd import api
    rng = np.random.default_rng(4)
    src = (rng.standard_normal((40, 3)) * 20).astype(np.float32)
    tgt = (rng.standard_normal((40, 3)) * 20).astype(np.float32)
    fm = api.FPFHManager(0.5, 0.75)
    with pytest.raises(ValueError):
        fm.saveFeaturePair(1, 2)
    with pytest.raises(ValueError):
        fm.loadFeaturePair(1, 2)
    fm._src_kps, fm._tgt_kps = api._as_cloud(src), api._as_cloud(tgt)
    fm.setSaveDir(tmp_path)
    fm.saveFeaturePair(7, 123)
    name = tmp_path / "000007_to_000123.pcd"
    assert name.read_text() == _expected_ascii_pcd(np.concatenate([src, tgt]))
    fm2 = api.FPFHManager(0.5, 0.75)
    fm2.setLoadDir(tmp_path)
    fm2.loadFeaturePair(7, 123)
    assert fm2.getSrcKps().shape == (40, 4) and fm2.getTgtKps().shape == (40, 4)
    assert np.allclose(fm2.getSrcKps()[:, :3], src, rtol=1e-7) and np.allclose(fm2.getTgtKps()[:, :3], tgt, rtol=1e-7)
    assert fm2.getSrcMatched().shape == (3, 40)
    with pytest.raises(ValueError):
        fm2.loadFeaturePair(1, 2)
    # the C++ header: load the Python-written file, save it again under another index
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    exe = str(tmp_path / "cache_demo")
    libdir = os.path.join(root, "quatro_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "cache_demo.cpp"), "-o", exe, "-L", libdir,
                           "-lquatro_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    import torch
    out = None
    for extra in ("", os.path.join(os.path.dirname(torch.__file__), "lib")):
        env = dict(os.environ)
        if extra:
            env["LD_LIBRARY_PATH"] = extra + ":" + env.get("LD_LIBRARY_PATH", "")
        pr = subprocess.run([exe, str(tmp_path), "7", "123", "8", "124"], env=env, capture_output=True, text=True,
                            timeout=120)
        if pr.returncode == 0:
            out = pr.stdout
            break
    assert out is not None, pr.stderr[-500:]
    assert out.split() == ["40", "40"]
    assert (tmp_path / "000008_to_000124.pcd").read_text() == _expected_ascii_pcd(fm2_merge(fm2))


def fm2_merge(fm):
    return np.concatenate([fm.getSrcKps()[:, :3], fm.getTgtKps()[:, :3]])


def test_pcd_reader_survives_hostile_headers(tmp_path):
    """Sizes come from the file: absurd SIZE / COUNT / POINTS / compressed-size words must end in an I/O error, not in
    arithmetic overflow or a giant allocation (found by fuzzing the reader under ASan/UBSan)."""
    body = "1 2 3\n"
    for bad in ("SIZE 4 4 2147483647", "COUNT 1 1 2147483647", "SIZE 4 4 0", "COUNT 1 1 -3"):
        head = "VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA ascii\n"
        key = bad.split()[0]
        head = "\n".join(bad if ln.startswith(key) else ln for ln in head.split("\n"))
        (tmp_path / "h.pcd").write_text(head + body)
        with pytest.raises(OSError):
            ql.read_pcd_xyz(str(tmp_path / "h.pcd"))
    head = "VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 2000000000\nHEIGHT 1\nPOINTS 2000000000\n"
    (tmp_path / "big.pcd").write_bytes((head + "DATA binary\n").encode() + b"\x00" * 24)
    with pytest.raises(OSError):
        ql.read_pcd_xyz(str(tmp_path / "big.pcd"))
    (tmp_path / "lzf.pcd").write_bytes((head.replace("2000000000", "10") + "DATA binary_compressed\n").encode() +
                                       struct.pack("<II", 0xFFFFFFFF, 120) + b"\x00" * 16)
    with pytest.raises(OSError):
        ql.read_pcd_xyz(str(tmp_path / "lzf.pcd"))
