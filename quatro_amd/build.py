"""Builds quatro_amd/libquatro_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

Flags that matter for parity with the CPU oracle:
  -ffp-contract=off                          no FMA contraction: float/double expressions round exactly as written
  -fhip-fp32-correctly-rounded-divide-sqrt   IEEE fp32 division / sqrt (the default, stated explicitly)
  no -ffast-math, denormals preserved (hipcc default)
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libquatro_hip.so")
# the product library above has ONE path; the comparison engines of the test-suite (all-exact / f32-MFMA nearest-neighbour
# search, one-workgroup matcher tails, peeling / sweep core numbers, quadratic ranking) and the knobs that select them
# live in a second build with -DQTR_TEST_ENGINES that only tests load (quatro_amd.lib.Handle(lib_path=...))
TEST_LIB = os.path.join(HERE, "libquatro_hip_testengines.so")
SOURCES = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc", ".map"))) + [
    os.path.join("..", "..", "include", "qtr_math.h"), os.path.join("..", "..", "include", "quatro_hip.h")]


def is_stale(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES)


def build_test_engines(force: bool = False, verbose: bool = True) -> str:
    return build(force, verbose, lib=TEST_LIB, defines=("-DQTR_TEST_ENGINES",))


def build(force: bool = False, verbose: bool = True, lib: str = LIB, defines=()) -> str:
    if not force and not is_stale(lib):
        return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, *defines, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wno-unused-value", "-fvisibility=hidden",
           "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"),
           os.path.join(CSRC, "unity.hip"), "-ldl", "-o", lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_test_engines(force="--force" in sys.argv)
