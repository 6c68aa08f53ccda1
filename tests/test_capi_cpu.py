"""No-GPU checks of the boundary: the C-ABI library builds, loads and exports every symbol that
include/quatro_hip.h declares; the host-side mirror validates arguments like the reference; the product
has no CPU fallback."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def libpath():
    from quatro_amd import build as qbuild
    return qbuild.build(force=False, verbose=False)


def test_library_exports_every_declared_symbol(libpath):
    hdr = open(os.path.join(ROOT, "include", "quatro_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(qtr_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 17
    lib = ctypes.CDLL(libpath)
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    from quatro_amd import lib as ql
    assert set(ql.EXPORTS) == set(declared)


def test_every_entry_point_with_pointer_arguments_declares_its_ctypes_signature(libpath):
    """ctypes passes a bare Python int as a 32-bit C int: an entry point called with data pointers but without argtypes
    truncates them (a crash on the GPU box, nothing on a CPU run).  Only the four one-struct-by-reference helpers may go
    without."""
    from quatro_amd import lib as ql
    lib = ql.load()
    bare = {n for n in ql.EXPORTS if getattr(lib, n).argtypes is None}
    assert bare <= {"qtr_default_limits", "qtr_default_params", "qtr_demo_params", "qtr_default_frontend_params"}, bare


def test_dynamic_symbol_table_is_exactly_the_declared_c_abi(libpath):
    """-fvisibility=hidden + the linker version script (csrc/exports.map): the library's dynamic symbol table holds the
    entry points of include/quatro_hip.h and nothing else — no mangled internals, kernel handles or device stubs."""
    hdr = open(os.path.join(ROOT, "include", "quatro_hip.h")).read()
    declared = set(re.findall(r"^QTR_API [^\n(]*?\b(qtr_[a-z_0-9]+)\s*\(", hdr, flags=re.M))
    assert declared == set(re.findall(r"\b(qtr_[a-z_0-9]+)\s*\(", hdr)), "an entry point without QTR_API"
    out = subprocess.check_output(["nm", "-D", "--defined-only", libpath], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == declared, (sorted(exported - declared)[:10], sorted(declared - exported)[:10])


def test_struct_layouts_match_header(libpath):
    """ctypes mirrors must have the C sizes (checked by compiling a tiny C program against the header)."""
    src = r'''
#include <stdio.h>
#include "quatro_hip.h"
int main(void){printf("%zu %zu %zu %zu %zu\n", sizeof(qtr_limits), sizeof(qtr_params), sizeof(qtr_frontend_params),
 sizeof(qtr_result), sizeof(qtr_stage_times)); return 0;}
'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = list(map(int, subprocess.check_output([exe]).split()))
    from quatro_amd import lib as ql
    assert sizes == [ctypes.sizeof(ql.Limits), ctypes.sizeof(ql.Params), ctypes.sizeof(ql.FrontendParams),
                     ctypes.sizeof(ql.Result), ctypes.sizeof(ql.StageTimes)]


def test_defaults_match_reference_params(libpath):
    from quatro_amd import lib as ql
    p = ql.default_params()  # Quatro::Params defaults (reference include/quatro.hpp:202-268)
    assert (p.noise_bound, p.cbar2, p.rotation_gnc_factor, p.rotation_max_iterations) == (0.3, 1.0, 1.4, 100)
    assert p.rotation_cost_threshold == 1e-6 and p.inlier_selection_mode == ql.INLIER_PMC_HEU and p.cote_median == 1
    d = ql.demo_params()  # config/params.yaml:22-44
    assert (d.rotation_max_iterations, d.rotation_cost_threshold) == (50, 1.1e-4)
    f = ql.default_frontend_params()
    assert (round(f.voxel_size, 3), round(f.normal_radius, 3), round(f.fpfh_radius, 3), round(f.tuple_scale, 3)) == \
        (0.3, 0.5, 0.75, 0.95)


def test_create_refuses_limits_the_kernels_cannot_address(libpath):
    """qtr_create checks its limits before it touches a device: k_recheck_filter's per-wave lists pack a base row into 20
    bits (quatro_amd/csrc/match.hip, QTR_NN_MAX_ROWS), so a cloud of 2^20 rows would alias rows silently — refused."""
    import ctypes
    from quatro_amd import lib as ql
    lib = ql.load()
    for mv in (1 << 20, (1 << 20) - 31, 1 << 21):
        lim = ql.Limits(1 << 22, mv, 8192, 1, 0)
        h = ctypes.c_void_p()
        assert lib.qtr_create(0, ctypes.byref(lim), ctypes.byref(h)) == ql.QTR_ERR_BAD_ARG and not h


def test_no_cpu_fallback_when_library_missing(tmp_path, monkeypatch):
    from quatro_amd import lib as ql
    monkeypatch.setattr(ql, "_lib", None)
    monkeypatch.setattr(ql, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ql.load()


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "quatro_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                for pat in (r"^\s*(from|import)\s+oracle", r"libquatro_oracle", r"\bqo_[a-z]", r"oracle[/\\]",
                            r"#include\s+\".*oracle"):
                    assert not re.search(pat, txt, flags=re.M), f"{f} reaches into oracle/: {pat}"


def test_host_mirror_argument_checks():
    from quatro_amd import api
    fm = api.FPFHManager(0.9, 0.5)
    with pytest.raises(ValueError, match="Normal should be lower than fpfh_radius"):
        fm.setFeaturePair(np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32))
    q = api.Quatro()
    p = api.Params()
    assert p.inlier_selection_mode == api.INLIER_SELECTION_MODE.PMC_HEU and p.cote_mode == "median"
    q.reset(p)
    with pytest.raises(ValueError):
        q.computeTransformation()
    q.setInputSource(np.zeros((5, 3), np.float32))
    q.setInputTarget(np.zeros((0, 3), np.float32))  # prints the PCL error, keeps target unset
    assert q.target_ is None
    q.params_.cote_mode = "bogus"
    with pytest.raises(ValueError, match="COTE"):
        q._c_params()


def test_cpp_dropin_header_compiles_against_c_abi(libpath, tmp_path):
    """include/quatro.hpp + include/fpfh_manager.hpp (the reference's class surface over the C ABI) build with
    plain g++ and link against libquatro_hip.so; no GPU is touched."""
    src = os.path.join(ROOT, "tests", "cpp", "dropin_demo.cpp")
    if not os.path.exists(src):
        pytest.skip("C++ drop-in demo not present")
    exe = tmp_path / "dropin_demo"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), src, "-o", str(exe),
                           "-L", os.path.dirname(libpath), "-lquatro_hip", "-Wl,-rpath," + os.path.dirname(libpath),
                           "-Wl,-rpath,/opt/rocm/lib"])
    assert exe.exists()


def test_patchwork_parameter_mirror():
    """api.PatchWork keeps the reference's parameter names and its consistency checks (patchwork.hpp:590-614); the
    defaults of the C ABI equal config/patchwork_params.yaml as restated by the tests' oracle wrapper."""
    from oracle import oracle as qo
    from quatro_amd import api
    from quatro_amd import lib as ql
    a, b = ql.pw_params(), qo.pw_params()
    for name, _ in ql.PwParams._fields_:
        va, vb = getattr(a, name), getattr(b, name)
        if hasattr(va, "__len__"):
            assert list(va) == list(vb), name
        else:
            assert va == vb, name
    pw = api.PatchWork(sensor_height=1.9, czm={"num_zones": 2, "num_sectors_each_zone": [8, 16],
                                               "num_rings_each_zone": [2, 3], "min_ranges_each_zone": [2.7, 10.0],
                                               "elevation_thresholds": [-1.0, -0.8], "flatness_thresholds": [1e-4, 2e-4]})
    assert pw.params.num_zones == 2 and pw.params.num_thr == 2 and pw.params.sensor_height == 1.9
    with pytest.raises(ValueError):
        api.PatchWork(min_r=3.0)
    with pytest.raises(ValueError):
        api.PatchWork(czm={"num_zones": 3, "num_sectors_each_zone": [8, 16]})
    with pytest.raises(ValueError):
        api.PatchWork(czm={"elevation_thresholds": [-1.0], "flatness_thresholds": []})
    with pytest.raises(TypeError):
        api.PatchWork(bogus=1)


def test_drop_in_headers_type_check_against_pcl_shaped_headers():
    """The QUATRO_HAVE_PCL branch of the drop-in headers (real pcl:: / Eigen:: / boost:: types instead of the built-in
    stand-ins) is compiled against tests/cpp/pcl_stub: declaration-level headers shaped like PCL 1.10 / Eigen 3.3
    (boost::shared_ptr cloud pointers, aligned-allocator point storage, column-major six-parameter Eigen::Matrix,
    pcl::Registration<Source, Target, Scalar> with its pure virtual computeTransformation).  Every demo must build in
    that configuration too, warning-free."""
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cpp = os.path.join(root, "tests", "cpp")
    stub = os.path.join(cpp, "pcl_stub")
    macros = subprocess.run(["g++", "-std=c++17", "-dM", "-E", "-I", stub, "-I", os.path.join(root, "include"),
                             os.path.join(cpp, "dropin_demo.cpp")], capture_output=True, text=True, check=True).stdout
    assert "#define QUATRO_HAVE_PCL 1" in macros  # the branch under test is really the one selected
    for name in sorted(os.listdir(cpp)):
        if name.endswith(".cpp"):
            subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", stub,
                                   "-I", os.path.join(root, "include"), os.path.join(cpp, name)])


def test_cpp_drop_in_headers_compile_and_read_ros_style_parameters(tmp_path):
    """Every C++ demo compiles against the drop-in headers (syntax check, no GPU needed), and PatchWork's NodeHandle-style
    constructor reads "/patchwork/..." parameters from any object with ros::NodeHandle's param()/getParam()."""
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cpp = os.path.join(root, "tests", "cpp")
    for name in sorted(os.listdir(cpp)):
        if name.endswith(".cpp"):
            subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                                   os.path.join(cpp, name)])
    from quatro_amd import build as qbuild
    qbuild.build(force=False)
    libdir = os.path.join(root, "quatro_amd")
    exe = str(tmp_path / "nodehandle_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"), os.path.join(cpp, "nodehandle_demo.cpp"),
                           "-o", exe, "-L", libdir, "-lquatro_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    import torch
    ok = False
    for extra in ("", os.path.join(os.path.dirname(torch.__file__), "lib")):
        env = dict(os.environ)
        if extra:
            env["LD_LIBRARY_PATH"] = extra + ":" + env.get("LD_LIBRARY_PATH", "")
        if subprocess.run([exe], env=env, timeout=120).returncode == 0:
            ok = True
            break
    assert ok


def test_teaser_utils_header_against_numpy_svd(tmp_path):
    """include/teaser/utils.h (svdRot, svdRot2d, findNonzero, maskVector, calculateDiameter) against numpy's SVD
    construction of the reference's helpers (reference include/teaser/utils.h:109-200); host code, no GPU."""
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    exe = str(tmp_path / "utils_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "utils_demo.cpp"), "-o", exe])
    rng = np.random.default_rng(3)
    n = 40
    X = rng.standard_normal((3, n)) * 5
    a = rng.standard_normal(4)
    a /= np.linalg.norm(a)
    from scipy.spatial.transform import Rotation as Rt
    Rm = Rt.from_quat(a).as_matrix()
    Y = Rm @ X + 0.05 * rng.standard_normal((3, n))
    w = rng.random(n)
    text = f"{n}\n" + "\n".join(" ".join(repr(float(v)) for v in (*X[:, j], *Y[:, j], w[j])) for j in range(n)) + "\n"
    out = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=60, check=True).stdout.strip().splitlines()

    def svd_rot(Xm, Ym):
        U, _, Vt = np.linalg.svd((Xm * w) @ Ym.T)
        V = Vt.T
        if np.linalg.det(U) * np.linalg.det(V) < 0:
            V[:, -1] *= -1
        return V @ U.T
    R3 = np.array([float(v) for v in out[0].split()[1:]]).reshape(3, 3)
    R2 = np.array([float(v) for v in out[1].split()[1:]]).reshape(2, 2)
    assert np.abs(R3 - svd_rot(X, Y)).max() < 1e-12 and np.abs(R2 - svd_rot(X[:2], Y[:2])).max() < 1e-12
    diam = 2 * np.sqrt(((X - X.mean(1, keepdims=True)) ** 2).sum(0).max())
    assert abs(float(out[2].split()[1]) - diam) < 1e-5 * diam
    keep = np.nonzero(w >= 0.5)[0]
    assert [int(v) for v in out[3].split()[1:]] == keep.tolist()
    assert [int(v) for v in out[4].split()[1:]] == (100 + keep).tolist()


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """Every struct of include/quatro_hip.h has the same size and the same field offsets in quatro_amd/lib.py's ctypes
    mirror (a C program compiled against the header prints sizeof / offsetof; field names must match too)."""
    from quatro_amd import lib as ql
    pairs = {"qtr_limits": ql.Limits, "qtr_params": ql.Params, "qtr_frontend_params": ql.FrontendParams,
             "qtr_result": ql.Result, "qtr_stage_times": ql.StageTimes, "qtr_pw_params": ql.PwParams,
             "qtr_ip_params": ql.IpParams}
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    hdr = open(os.path.join(root, "include", "quatro_hip.h")).read()
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "quatro_hip.h"', "int main(void) {"]
    for cname, cls in pairs.items():
        body = re.search(r"typedef struct " + cname + r" \{(.*?)\} " + cname + ";", hdr, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(",") if "[" not in decl or decl.count(",") else [decl]:
                names.append(re.sub(r"\[.*?\]", "", part.strip().split()[-1]).lstrip("*"))
        assert names == [n for n, _ in cls._fields_], (cname, names)
        lines.append(f'  printf("{cname} %zu", sizeof({cname}));')
        for n in names:
            lines.append(f'  printf(" %zu", offsetof({cname}, {n}));')
        lines.append('  printf("\\n");')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(root, "include"), str(src), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    for line in out:
        w = line.split()
        cls = pairs[w[0]]
        assert int(w[1]) == ctypes.sizeof(cls), w[0]
        assert [int(v) for v in w[2:]] == [getattr(cls, n).offset for n, _ in cls._fields_], w[0]


def test_teaser_graph_host_class_under_sanitizers(tmp_path):
    """teaser::Graph of include/teaser/graph.h (addEdge refuses duplicates / unknown vertices, removeEdge, adjacency
    lists, the bit matrix handed to qtr_max_clique) against a Python model, compiled with ASan + UBSan; no GPU."""
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    exe = str(tmp_path / "graph_host_demo")
    libdir = os.path.join(root, "quatro_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "graph_host_demo.cpp"),
                           "-o", exe, "-L", libdir, "-lquatro_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    rng = np.random.default_rng(5)
    n = 150
    ops, adj, nedges = [], [[] for _ in range(n)], 0
    for _ in range(3000):
        a, b = int(rng.integers(-2, n + 2)), int(rng.integers(-2, n + 2))
        if rng.random() < 0.8:
            ops.append(f"+ {a} {b}")
            if 0 <= a < n and 0 <= b < n and b not in adj[a]:
                adj[a].append(b)
                if a != b:
                    adj[b].append(a)
                else:
                    adj[a].append(a)
                nedges += 1
        else:
            ops.append(f"- {a} {b}")
            if 0 <= a < n and 0 <= b < n and b in adj[a]:
                adj[a] = [x for x in adj[a] if x != b]
                adj[b] = [x for x in adj[b] if x != a]
                nedges -= 1
    import torch
    out = None
    for extra in ("", os.path.join(os.path.dirname(torch.__file__), "lib")):
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
        if extra:
            env["LD_LIBRARY_PATH"] = extra + ":" + env.get("LD_LIBRARY_PATH", "")
        p = subprocess.run([exe], input=f"{n}\n" + "\n".join(ops) + "\n", capture_output=True, text=True, env=env, timeout=120)
        if p.returncode == 0:
            out = p.stdout.strip().splitlines()
            break
    assert out is not None, p.stderr[-800:]
    assert out[0].split() == [str(n), str(nedges)]
    for v in range(n):
        assert out[1 + v] == f"{v}:" + "".join(f" {u}" for u in adj[v]), v
    W = (n + 63) // 64
    words = [int(x, 16) for x in out[1 + n:1 + n + n * W]]
    for v in range(n):
        bits = 0
        for u in adj[v]:
            if u != v:
                bits |= 1 << u
        got = sum(words[v * W + w] << (64 * w) for w in range(W))
        assert got == bits, v
    assert out[-1].split() == ["has", "1" if 1 in adj[0] else "0", "0", "0"]
