#!/usr/bin/env python3
"""The one check of go/hering/*.go that is possible without a Go toolchain: every `C.he_*(...)` call names a function
declared in include/hering.h (or hering_debug.h) and passes as many arguments as the declaration has parameters; every
`C.HE_*` constant exists in the header's enums.  Also lists which of rlwe.EvaluatorProvider's seven methods and of
schemes.Evaluator's methods the Go package defines.

    python tools/check_go_abi.py            # exit status 0 when everything matches
"""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_decls():
    src = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("hering.h", "hering_debug.h"))
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|const char \*)\s*(he_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        params = m.group(2).strip()
        decls[m.group(1)] = 0 if params in ("", "void") else len(split_args(params))
    consts = set(re.findall(r"\b(HE_[A-Z0-9_]+)\b", src))
    return decls, consts


def split_args(s):
    """top-level comma split (parentheses / brackets / braces nest)"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def go_calls(path):
    src = open(path).read()
    src = re.sub(r"//[^\n]*", "", src)
    for m in re.finditer(r"\bC\.(he_[a-z0-9_]+)\s*\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        yield m.group(1), len(split_args(src[m.end():i - 1])), src.count("\n", 0, m.start()) + 1
    for m in re.finditer(r"\bC\.(HE_[A-Z0-9_]+)\b", src):
        yield m.group(1), None, src.count("\n", 0, m.start()) + 1


PROVIDER = ["DecomposeNTT", "CheckAndGetGaloisKey", "GadgetProductLazy", "GadgetProductHoistedLazy", "AutomorphismHoistedLazy",
            "ModDownQPtoQNTT", "AutomorphismIndex"]                                   # core/rlwe/rlwe.go:10-18
SCHEMES = ["Add", "AddNew", "Sub", "SubNew", "Mul", "MulNew", "MulRelin", "MulRelinNew", "MulThenAdd", "Relinearize",
           "Rescale", "GetRLWEParameters"]                                            # schemes/schemes.go:14-28


def main():
    decls, consts = header_decls()
    errors, ncalls, used = [], 0, set()
    files = sorted(glob.glob(os.path.join(ROOT, "go", "hering", "*.go")))
    for f in files:
        for name, nargs, line in go_calls(f):
            where = f"{os.path.relpath(f, ROOT)}:{line}"
            if nargs is None:
                if name not in consts:
                    errors.append(f"{where}: constant {name} is not in the header")
                continue
            ncalls += 1
            used.add(name)
            if name not in decls:
                errors.append(f"{where}: {name} is not declared in include/*.h")
            elif decls[name] != nargs:
                errors.append(f"{where}: {name} called with {nargs} arguments, declared with {decls[name]}")
    src = "".join(open(f).read() for f in files)
    methods = set(re.findall(r"func \(\w+ \*(?:Evaluator|SchemeEvaluator)\) (\w+)\(", src))
    for m in PROVIDER + SCHEMES:
        if m not in methods:
            errors.append(f"go/hering: method {m} (rlwe.EvaluatorProvider / schemes.Evaluator) is not defined")
    print(f"{len(files)} Go files, {ncalls} C.he_* calls to {len(used)} of {len(decls)} declared entry points; "
          f"{len(PROVIDER)} EvaluatorProvider + {len(SCHEMES)} schemes.Evaluator methods present" if not errors else "\n".join(errors))
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
