#!/usr/bin/env python3
"""ref_extract.py — build step of `make -C oracle ref_solver` (test infrastructure).

Cuts the definitions of the Eigen-only members of class Quatro (enums, Params, the solver functions and
computeTransformation) out of the reference header where it lies (/root/reference/include/quatro.hpp), and pcl2eigen out
of conversion.hpp, and writes them, unchanged, to the two files given — temporaries under oracle/_ref/ that
oracle/ref_solver_wrap.cpp includes (the members inside a small host struct) and that the Makefile deletes after
compiling.  The header as a whole cannot be compiled here (PCL, FLANN, ROS, PMC are absent); these functions only
need Eigen, for which oracle/ref_shim_solver/ holds a stand-in.  Nothing of the reference enters the repository.

usage: ref_extract.py /root/reference/include members.inc free.inc
"""
import re
import sys

# (regular expression matching the start of the piece, what it is, kind): "fn" = a member function definition, "decl" = a
# struct / enum definition (ends at the ';' after its closing brace)
WANTED = [
    (r"struct RegistrationSolution \{", "RegistrationSolution :161-168", "decl"),
    (r"enum class ROTATION_ESTIMATION_ALGORITHM \{", "ROTATION_ESTIMATION_ALGORITHM :172-175", "decl"),
    (r"enum class INLIER_SELECTION_MODE \{", "INLIER_SELECTION_MODE :184-189", "decl"),
    (r"enum class INLIER_GRAPH_FORMULATION \{", "INLIER_GRAPH_FORMULATION :197-200", "decl"),
    (r"struct Params \{", "Params :202-268", "decl"),
    (r"Params getParams\(\)", "getParams :271", "fn"),
    (r"void setParams\(Params params\)", "setParams :273", "fn"),
    (r"Eigen::Matrix<double, 3, Eigen::Dynamic> computeTIMs\(", "computeTIMs :307-344", "fn"),
    (r"double solveForScale\(", "solveForScale (2-argument form) :346-353", "fn"),
    (r"void solveForScale\(", "solveForScale (4-argument form) :355-386", "fn"),
    (r"Eigen::Matrix3d solveForRotation\(", "solveForRotation :388-428", "fn"),
    (r"void solveForRotation2D\(", "solveForRotation2D :430-572", "fn"),
    (r"Eigen::Vector3d solveForTranslation\(", "solveForTranslation (3-argument form) :574-583", "fn"),
    (r"void solveForTranslation\(", "solveForTranslation (5-argument form) :585-616", "fn"),
    (r"void estimate\(", "estimate :618-747", "fn"),
    (r"void computeTransformation\(Eigen::Matrix4d &output\)", "computeTransformation :769-936", "fn"),
]
CONVERSION = [(r"template<typename T>\s*void pcl2eigen\(", "pcl2eigen (include/conversion.hpp:37-44)", "fn")]


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


def cut(src, wanted):
    out = []
    for pat, what, kind in wanted:
        m = list(re.finditer(pat, src))
        assert len(m) == 1, (pat, len(m))
        ls = src.rfind("\n", 0, m[0].start()) + 1
        if kind == "fn":
            text = definition(src, ls)
        else:
            i = src.index("{", m[0].start())
            depth = 0
            while True:
                j = skip_noncode(src, i)
                if j != i:
                    i = j
                    continue
                depth += src[i] == "{"
                depth -= src[i] == "}"
                i += 1
                if depth == 0:
                    break
            text = src[ls:src.index(";", i) + 1]
        out.append("// ---- %s\n%s\n" % (what, text))
    return out


# usage: ref_extract.py <reference include dir> members.inc free.inc
inc = sys.argv[1]
open(sys.argv[2], "w").write("\n".join(cut(open(inc + "/quatro.hpp").read(), WANTED)))
open(sys.argv[3], "w").write("\n".join(cut(open(inc + "/conversion.hpp").read(), CONVERSION)))
