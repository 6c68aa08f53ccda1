#!/usr/bin/env python3
"""ref_extract.py — build step of `make -C oracle ref_solver` (test infrastructure).

Cuts the definitions of a few member functions of class Quatro out of the reference header where it lies
(/root/reference/include/quatro.hpp) and writes them, unchanged, to the file given as the second argument — a temporary
under oracle/_ref/ that oracle/ref_solver_wrap.cpp includes inside a small host struct and that the Makefile deletes
after compiling.  The header as a whole cannot be compiled here (PCL, FLANN, ROS, PMC are absent); these functions only
need Eigen, for which oracle/ref_shim_solver/ holds a stand-in.  Nothing of the reference enters the repository.

usage: ref_extract.py /root/reference/include/quatro.hpp out.inc
"""
import re
import sys

WANTED = [  # (regular expression matching the start of the definition, what it is)
    (r"Eigen::Matrix<double, 3, Eigen::Dynamic> computeTIMs\(", "computeTIMs :307-344"),
    (r"void solveForScale\(", "solveForScale (4-argument form) :355-386"),
    (r"void solveForRotation2D\(", "solveForRotation2D :430-572"),
    (r"void solveForTranslation\(", "solveForTranslation (5-argument form) :585-616"),
    (r"void estimate\(", "estimate :618-747"),
]


def skip_noncode(s, i):
    """index after a comment / string / char literal starting at i, or i if none starts there"""
    if s.startswith("//", i):
        j = s.find("\n", i)
        return len(s) if j < 0 else j
    if s.startswith("/*", i):
        return s.index("*/", i) + 2
    if s[i] in "\"'":
        q = s[i]
        j = i + 1
        while s[j] != q:
            j += 2 if s[j] == "\\" else 1
        return j + 1
    return i


def definition(s, start):
    i = s.index("(", start)
    depth = 0
    while True:  # the parameter list
        j = skip_noncode(s, i)
        if j != i:
            i = j
            continue
        depth += s[i] == "("
        depth -= s[i] == ")"
        i += 1
        if depth == 0:
            break
    while s[i] != "{":
        assert s[i] in " \t\n", "unexpected text between the parameter list and the body"
        i += 1
    depth = 0
    while True:  # the body
        j = skip_noncode(s, i)
        if j != i:
            i = j
            continue
        depth += s[i] == "{"
        depth -= s[i] == "}"
        i += 1
        if depth == 0:
            return s[start:i]


src = open(sys.argv[1]).read()
out = []
for pat, what in WANTED:
    m = [x for x in re.finditer(pat, src)]
    assert m, pat
    # several overloads share a name: take the one whose text up to the first ')' holds pointer parameters (the worker form)
    pick = None
    for x in m:
        head = src[x.start():src.index("{", x.start())]
        if "computeTIMs" in pat or "*" in head:
            pick = x
            break
    assert pick is not None, pat
    ls = src.rfind("\n", 0, pick.start()) + 1
    out.append("// ---- %s\n%s\n" % (what, definition(src, ls)))
open(sys.argv[2], "w").write("\n".join(out))
