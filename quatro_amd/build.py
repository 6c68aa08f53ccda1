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
SOURCES = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc", ".map"))) + [
    os.path.join("..", "..", "include", "qtr_math.h"), os.path.join("..", "..", "include", "quatro_hip.h")]


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not is_stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wno-unused-value", "-fvisibility=hidden",
           "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"),
           os.path.join(CSRC, "unity.hip"), "-ldl", "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
