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


# ---- the public entry points keep the reference's keyword names and defaults
def _julia_keywords(src, fname):
    """keyword name -> default text of `function fname(...; kw=default, ...)` in a Julia source"""
    m = re.search(r"^function " + re.escape(fname) + r"\(", src, flags=re.M)
    assert m, "function %s is not defined" % fname
    i, depth = m.end() - 1, 0
    j = i
    while True:
        if src[j] == "(":
            depth += 1
        elif src[j] == ")":
            depth -= 1
            if depth == 0:
                break
        j += 1
    sig = " ".join(src[i + 1:j].split())
    assert ";" in sig, (fname, "no keyword section")
    kws = {}
    for part in split_top(sig.split(";", 1)[1]):
        name, _, default = part.partition("=")
        name = name.split("::")[0].strip()
        kws[name] = default.strip()
    return kws


def _fixture():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "reference_api_keywords.json"), encoding="utf-8"))


def _check_keywords(src, fx):
    for fname in ("Fit!", "generate_samples!", "generate_samples_dbl!"):
        got = _julia_keywords(src, fname)
        for name, default in fx[fname].items():
            assert name in got, "%s lost the reference's keyword %s" % (fname, name)
            assert got[name] == default, "%s: keyword %s defaults to %s, the reference's to %s" % (fname, name, got[name], default)
        extra = set(got) - set(fx[fname])
        assert extra <= set(fx["shim_extensions"]), "%s has keywords the reference does not know: %s" % (fname, sorted(extra))


def test_julia_entry_points_keep_the_reference_keywords():
    """`Fit!`, `generate_samples!`, `generate_samples_dbl!` of julia/BNRHip.jl exist with every keyword of the reference (same names, same
    defaults: src/gibbs.jl:725-727, 897-899, 1051-1053, transcribed into tests/golden/reference_api_keywords.json) plus only the placement
    extensions (device, comm, tick); `Fit!` writes parameters.log with the reference's lines and dispatches on mingen / maxgen; the doubling
    scheme moves the retained tail and grows the table through the C ABI (bnr_chain_move_rows, bnr_chain_resize)."""
    src = open(os.path.join(ROOT, "julia", "BNRHip.jl"), encoding="utf-8").read()
    fx = _fixture()
    _check_keywords(src, fx)
    fit = src[src.index("function Fit!("):]
    for needle in ('"BayesianNetworkRegression.jl Fit! function\\n"', 'citation(returnstring=true)', '"\\n\\nParameters:\\n"', 'nsamples=$nsamples', 'purge_burn=$purge_burn',
                   '"seed=$seed"', "(mingen > 0) && (maxgen > 0)", "generate_samples_dbl!(X, y, R;", "maxburn = nburn + nsamples"):
        assert needle in fit, needle
    dbl = src[src.index("function generate_samples_dbl!("):src.index("function Fit!(")]
    for needle in ("round(mingen / 2)", "num2move = tot_samples", "move_rows!(ch, 1, tot_sze - num2move + 1, num2move)", "resize_table!(ch, tot_save)",
                   "run!(runner, num2move + 1, 0, tot_save, purge_burn", "tot_generated < maxgen", "tot_generated + mingen"):
        assert needle in dbl, needle


def test_python_mirror_keeps_the_reference_keywords():
    """the ctypes mirror the GPU tests drive (api.py: Fit, generate_samples, generate_samples_dbl) carries the same keywords under their
    ASCII names with the same defaults"""
    import inspect
    import bnr_amd
    fx = _fixture()
    conv = {"true": True, "false": False, "nothing": None}
    for jname, pyfn in (("Fit!", bnr_amd.Fit), ("generate_samples!", bnr_amd.generate_samples), ("generate_samples_dbl!", bnr_amd.generate_samples_dbl)):
        params = inspect.signature(pyfn).parameters
        for name, default in fx[jname].items():
            pn = fx["python_names"].get(name, name)
            assert pn in params, (jname, pn)
            want = conv[default] if default in conv else (default.strip('"') if default.startswith('"') else float(default))
            got = params[pn].default
            assert (got == want) if not isinstance(want, float) else (float(got) == want), (jname, pn, got, want)


def test_the_keyword_checker_notices_a_dropped_keyword():
    """non-vacuity: the same check fails when a keyword disappears from the shim, changes its default, or a foreign one appears"""
    src = open(os.path.join(ROOT, "julia", "BNRHip.jl"), encoding="utf-8").read()
    fx = _fixture()
    import pytest
    for broken in (src.replace("psrf_cutoff=1.01, x_transform=true, suppress_timer=false, num_chains=2, seed=nothing, purge_burn=nothing,\n              filename", "psrf_cutoff=1.01, x_transform=true, suppress_timer=false, num_chains=2, purge_burn=nothing,\n              filename", 1),
                   src.replace("mingen=10000, maxgen=100000", "mingen=10000, maxgen=50000", 1),
                   src.replace("function Fit!(X, y, R; η=1.01,", "function Fit!(X, y, R; chains_on_gpu=8, η=1.01,", 1)):
        assert broken != src
        with pytest.raises(AssertionError):
            _check_keywords(broken, fx)
