#!/usr/bin/env python3
"""The check of go/hering/*.go that is possible without a Go toolchain: every `C.he_*(...)` call names a function
declared in include/hering.h (or hering_debug.h), passes as many arguments as the declaration has parameters, and -- round 4 --
every argument has the KIND the parameter wants (int / uint64_t / he_handle / he_handle* / uint64_t* / int*), judged from the
shape of the Go expression (`C.int(..)`, `C.uint64_t(..)`, `x.h`, `&x.h`, `(*C.uint64_t)(unsafe.Pointer(..))`, identifiers whose
declaration in the file names a Handle or a C pointer type); every `C.HE_*` constant exists in the header's enums.  Also
lists which of rlwe.EvaluatorProvider's seven methods and of schemes.Evaluator's methods the Go package defines.  It is not a type
checker: the package has still never met the Go compiler.

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
    decls, kinds = {}, {}
    for m in re.finditer(r"\b(?:int|const char \*)\s*(he_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        params = m.group(2).strip()
        plist = [] if params in ("", "void") else split_args(params)
        decls[m.group(1)] = len(plist)
        kinds[m.group(1)] = [param_kind(p) for p in plist]
    consts = set(re.findall(r"\b(HE_[A-Z0-9_]+)\b", src))
    return decls, consts, kinds


def param_kind(p):
    """kind of a C parameter declaration"""
    p = " ".join(p.replace("const", " ").split())
    ptr = "*" in p or "[" in p
    if "he_handle" in p:
        return "handle*" if ptr else "handle"
    if "uint64_t" in p:
        return "u64*" if ptr else "u64"
    if "uint8_t" in p:
        return "u8*" if ptr else "u8"
    if re.match(r"^(int|size_t)\b", p) and not ptr:
        return "int"
    if re.match(r"^int\b", p) and ptr:
        return "int*"
    return "other"


def arg_kind(a, src):
    """kind of a Go argument expression, or None when its shape says nothing"""
    a = a.strip()
    if re.match(r"^C\.int\(", a) or re.match(r"^C\.HE_[A-Z0-9_]+$", a) or re.match(r"^C\.size_t\(", a):
        return "int"
    if re.match(r"^C\.uint64_t\(", a):
        return "u64"
    if re.match(r"^\(\*C\.uint64_t\)\(unsafe\.Pointer\(", a):
        return "u64*"
    if re.match(r"^\(\*C\.uint8_t\)\(unsafe\.Pointer\(", a):
        return "u8*"
    if re.match(r"^&[\w.\[\]]+\.h$", a):
        return "handle*"
    if re.match(r"^[\w.\[\]]+\.h$", a) or re.match(r"^h\(\w+\)$", a):
        return "handle"
    m = re.match(r"^&(\w+)\[0\]$", a)
    if m and re.search(r"\b%s\s*:?=\s*make\(\[\]C\.int\b" % m.group(1), src):
        return "int*"
    m = re.match(r"^(\w+)(\[\d+\])?$", a)
    if m:
        name = m.group(1)
        if re.search(r"\bvar\s+%s\s+(\[\d+\])?Handle\b" % name, src) or re.search(r"\b%s\s+(\[\d+\])?Handle\b" % name, src):
            return "handle"
        if re.search(r"\bvar\s+%s\s+\*C\.uint64_t\b" % name, src):
            return "u64*"
    return None


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
        args = split_args(src[m.end():i - 1])
        yield m.group(1), args, src.count("\n", 0, m.start()) + 1, src
    for m in re.finditer(r"\bC\.(HE_[A-Z0-9_]+)\b", src):
        yield m.group(1), None, src.count("\n", 0, m.start()) + 1, src


PROVIDER = ["DecomposeNTT", "CheckAndGetGaloisKey", "GadgetProductLazy", "GadgetProductHoistedLazy", "AutomorphismHoistedLazy",
            "ModDownQPtoQNTT", "AutomorphismIndex"]                                   # core/rlwe/rlwe.go:10-18
SCHEMES = ["Add", "AddNew", "Sub", "SubNew", "Mul", "MulNew", "MulRelin", "MulRelinNew", "MulThenAdd", "Relinearize",
           "Rescale", "GetRLWEParameters"]                                            # schemes/schemes.go:14-28


def main():
    decls, consts, kinds = header_decls()
    errors, ncalls, used, nchecked, nargs_total = [], 0, set(), 0, 0
    files = sorted(glob.glob(os.path.join(ROOT, "go", "hering", "*.go")))
    for f in files:
        for name, args, line, gosrc in go_calls(f):
            where = f"{os.path.relpath(f, ROOT)}:{line}"
            if args is None:
                if name not in consts:
                    errors.append(f"{where}: constant {name} is not in the header")
                continue
            ncalls += 1
            used.add(name)
            if name not in decls:
                errors.append(f"{where}: {name} is not declared in include/*.h")
            elif decls[name] != len(args):
                errors.append(f"{where}: {name} called with {len(args)} arguments, declared with {decls[name]}")
            else:
                for i, (a, want) in enumerate(zip(args, kinds[name])):
                    nargs_total += 1
                    got = arg_kind(a, gosrc)
                    if got is None or want == "other":
                        continue
                    nchecked += 1
                    if got != want:
                        errors.append(f"{where}: {name} argument {i + 1} `{a.strip()}` looks like {got}, the header wants {want}")
    for f in files:  # a structural sanity check of each file: brackets balance outside strings, runes and comments
        txt = re.sub(r"//[^\n]*|/\*.*?\*/", "", open(f).read(), flags=re.S)
        txt = re.sub(r'"(?:\\.|[^"\\\n])*"|`[^`]*`|\'(?:\\.|[^\'\\])\'', '""', txt)
        for o, c in ("{}", "()", "[]"):
            if txt.count(o) != txt.count(c):
                errors.append(f"{os.path.relpath(f, ROOT)}: {txt.count(o)} '{o}' against {txt.count(c)} '{c}'")
    src = "".join(open(f).read() for f in files)
    methods = set(re.findall(r"func \(\w+ \*(?:Evaluator|SchemeEvaluator)\) (\w+)\(", src))
    for m in PROVIDER + SCHEMES:
        if m not in methods:
            errors.append(f"go/hering: method {m} (rlwe.EvaluatorProvider / schemes.Evaluator) is not defined")
    print(f"{len(files)} Go files, {ncalls} C.he_* calls to {len(used)} of {len(decls)} declared entry points, {nchecked} of {nargs_total} "
          f"arguments kind-checked; {len(PROVIDER)} EvaluatorProvider + {len(SCHEMES)} schemes.Evaluator methods present"
          if not errors else "\n".join(errors))
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
