"""The Julia shim julia/BNRHip.jl cannot run here (no `julia` binary in the image), so its `ccall`s are checked statically against
the prototypes of include/bnr_hip.h: every symbol exists, the argument count agrees, and every argument and the return value have
the same width class (32/64-bit integer, double, pointer).  A parameter added to a header prototype makes this test fail until
the shim follows.  Also: a `ccall` target must be a literal `(:symbol, LIB)` (a symbol chosen at run time does not lower)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def split_top(s):
    """split on commas that are not nested in (), {} or []"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def c_class(t):
    t = re.sub(r"/\*.*?\*/", "", t).strip()
    if "*" in t or "[" in t or re.search(r"\b(bnr_progress_cb|bnr_allgather_fn)\b", t):
        return "ptr"
    t = re.sub(r"\bconst\b", "", t).split()
    base = t[0] if t else ""
    return {"int": "i32", "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "double": "f64", "void": "void"}[base]


def julia_class(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t in ("Cstring", "Ptr"):
        return "ptr"
    return {"Cint": "i32", "Int32": "i32", "UInt32": "u32", "Int64": "i64", "UInt64": "u64", "Cdouble": "f64", "Float64": "f64", "Cvoid": "void"}[t]


def header_prototypes():
    src = open(os.path.join(ROOT, "include", "bnr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"^\s*((?:const\s+)?(?:int|int32_t|double|void|char)\s*\*?)\s*(bnr_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.M | re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("void", "") else split_top(args)
        protos[name] = ("ptr" if "*" in ret else c_class(ret), [c_class(re.sub(r"\b\w+\s*(\[\d*\])?$", lambda k: k.group(1) or "", p).strip() or p) for p in params])
    return protos


def shim_ccalls():
    src = open(os.path.join(ROOT, "julia", "BNRHip.jl")).read()
    src = "\n".join(line.split("#")[0] if not line.lstrip().startswith("#") else "" for line in src.split("\n"))
    calls = []
    for m in re.finditer(r"ccall\(\(", src):
        i, depth = m.start() + len("ccall"), 0
        j = i
        while True:
            if src[j] == "(":
                depth += 1
            elif src[j] == ")":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        parts = split_top(src[i + 1:j])
        calls.append(parts)
    return src, calls


def test_every_ccall_matches_the_header():
    protos = header_prototypes()
    assert len(protos) >= 50 and "bnr_group_run" in protos and "bnr_rhat" in protos
    src, calls = shim_ccalls()
    assert len(calls) >= 19
    assert src.count("ccall(") == len(calls), "a ccall whose target is not a literal (:symbol, LIB) tuple does not lower in Julia"
    for parts in calls:
        target = parts[0]
        m = re.fullmatch(r"\(\s*:(bnr_\w+)\s*,\s*LIB\s*\)", target)
        assert m, "ccall target must be a literal (:symbol, LIB): %r" % target
        name = m.group(1)
        assert name in protos, "%s is not declared in include/bnr_hip.h" % name
        ret, argt = parts[1], parts[2]
        assert argt.startswith("(") and argt.endswith(")"), (name, argt)
        jargs = [julia_class(a) for a in split_top(argt[1:-1])]
        cret, cargs = protos[name]
        assert julia_class(ret) == cret, (name, "return", ret, cret)
        assert len(jargs) == len(cargs), (name, "argument count", len(jargs), len(cargs))
        assert jargs == cargs, (name, jargs, cargs)
        assert len(parts) - 3 == len(cargs), (name, "values passed", len(parts) - 3, len(cargs))


def test_the_checker_notices_a_changed_prototype():
    """non-vacuity: the same comparison fails for a prototype with one more parameter, a narrower one, or a missing symbol"""
    protos = header_prototypes()
    _, calls = shim_ccalls()
    name_of = lambda parts: re.fullmatch(r"\(\s*:(bnr_\w+)\s*,\s*LIB\s*\)", parts[0]).group(1)
    run = next(p for p in calls if name_of(p) == "bnr_group_run")
    jargs = [julia_class(a) for a in split_top(run[2][1:-1])]
    assert jargs == protos["bnr_group_run"][1]
    assert jargs != protos["bnr_group_run"][1] + ["i32"]
    widened = list(protos["bnr_group_run"][1]); widened[1] = "i64"
    assert jargs != widened
    assert "bnr_group_run_v2" not in protos
