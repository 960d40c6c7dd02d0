"""Layout guard for the two bindings of include/mci.h.  The Julia `ccall` binding (mcintegration.jl_amd/julia/MCIntegrationHIP.jl)
cannot be executed in the build image (no julia), so what CAN be checked without running it is checked here, field by field:

    C header (offsetof/sizeof from a tiny gcc program)  ==  ctypes Structures (_lib.py)  ==  Julia struct declarations

plus every symbol the Julia file `ccall`s: it must be declared in mci.h, exported by the library, and called with as many
arguments as the ctypes signature has.  A field added to mci_integrate_args without touching the bindings fails here instead
of silently corrupting a call."""
import ctypes as C
import os
import re
import subprocess

import pytest

import mcintegration_jl_amd as mci
from mcintegration_jl_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
JL = os.path.join(ROOT, "mcintegration.jl_amd", "julia", "MCIntegrationHIP.jl")
HDR = os.path.join(ROOT, "include", "mci.h")

PAIRS = [  # C struct, ctypes Structure, Julia struct
    ("mci_leaf_desc", _lib.LeafDesc, "LeafDesc"),
    ("mci_problem_desc", _lib.ProblemDesc, "ProblemDesc"),
    ("mci_integrate_args", _lib.IntegrateArgs, "IntegrateArgs"),
    ("mci_result", _lib.ResultC, "ResultC"),
]
JL_TYPES = {"Int32": (4, 4), "UInt32": (4, 4), "Cint": (4, 4), "Int64": (8, 8), "UInt64": (8, 8), "Float64": (8, 8), "Cdouble": (8, 8)}


def julia_struct(name):
    """[(field, julia type)] of `struct name ... end` / `mutable struct name ... end`"""
    src = open(JL).read()
    m = re.search(r"^(?:mutable )?struct %s\b(.*?)^end" % re.escape(name), src, flags=re.S | re.M)
    assert m, "struct %s not found in the Julia binding" % name
    body = re.sub(r"#.*", "", m.group(1))
    return re.findall(r"(\w+)::((?:Ptr\{[^}]*\})|\w+)", body)


def julia_layout(fields):
    """C-compatible layout of an isbits Julia struct: natural alignment, like the C compiler's"""
    off, out, maxal = 0, [], 1
    for name, ty in fields:
        size, al = (8, 8) if ty.startswith("Ptr{") else JL_TYPES[ty]
        off = (off + al - 1) // al * al
        out.append((name, off, size))
        off += size
        maxal = max(maxal, al)
    return out, (off + maxal - 1) // maxal * maxal


@pytest.fixture(scope="module")
def c_layout(tmp_path_factory):
    """{struct: ([(field, offset, size)], sizeof)} as gcc sees include/mci.h"""
    d = tmp_path_factory.mktemp("layout")
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "mci.h"', "int main(void) {"]
    for cname, cty, _ in PAIRS:
        prog.append('  printf("S %s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f, _t in cty._fields_:
            prog.append('  printf("F %s %s %%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s*)0)->%s));' % (cname, f, cname, f, cname, f))
    prog += ["  return 0;", "}"]
    src = d / "layout.c"
    src.write_text("\n".join(prog))
    exe = d / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True)
    res = {}
    for line in out.splitlines():
        p = line.split()
        if p[0] == "S":
            res.setdefault(p[1], [[], 0])[1] = int(p[2])
        else:
            res.setdefault(p[1], [[], 0])[0].append((p[2], int(p[3]), int(p[4])))
    return res


@pytest.mark.parametrize("cname,cty,jname", PAIRS, ids=[p[0] for p in PAIRS])
def test_struct_layouts_agree(c_layout, cname, cty, jname):
    cfields, csize = c_layout[cname]
    # the header declares exactly the fields the ctypes Structure lists (a field missing from ctypes would not compile above;
    # a field missing from the header is caught by counting the members of the typedef)
    hdr = open(HDR).read()
    end = hdr.index("} %s;" % cname)
    body = hdr[hdr.rindex("typedef struct {", 0, end) + len("typedef struct {"):end]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names_in_header = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names_in_header.append(re.findall(r"\*?\s*(\w+)\s*$", part.strip())[0])
    assert names_in_header == [f for f, _ in cty._fields_], (names_in_header, [f for f, _ in cty._fields_])
    # ctypes == C
    assert C.sizeof(cty) == csize
    for (f, off, size), (pf, pt) in zip(cfields, cty._fields_):
        assert f == pf and getattr(cty, pf).offset == off and C.sizeof(pt) == size, (cname, f)
    # Julia == C
    jfields = julia_struct(jname)
    assert [f for f, _ in jfields] == [f for f, _, _ in cfields], (jname, jfields)
    jl, jsize = julia_layout(jfields)
    assert jsize == csize
    for (f, off, size), (jf, joff, jsz) in zip(cfields, jl):
        assert (f, off, size) == (jf, joff, jsz), (cname, f, off, size, joff, jsz)
    # pointer-ness and signedness, too
    for (jf, jt), (pf, pt) in zip(jfields, cty._fields_):
        is_ptr = jt.startswith("Ptr{")
        assert is_ptr == (hasattr(pt, "contents") or pt in (C.c_void_p, C.c_char_p)), (jf, jt, pt)
        if jt in ("Float64", "Cdouble"):
            assert pt is C.c_double
        if jt == "UInt64":
            assert pt is C.c_uint64


def test_every_ccall_of_the_julia_binding_exists_and_has_the_right_arity():
    src = open(JL).read()
    sigs = {name: args for name, _res, args in _lib.SIGNATURES}
    hdr = open(HDR).read()
    L = mci.lib()
    calls = re.findall(r"ccall\(\(:(\w+), libmci\), (\w+), \((.*?)\)(?:,|\))", src, flags=re.S)
    assert len(calls) >= 20
    seen = set()
    for name, ret, argt in calls:
        seen.add(name)
        assert name in sigs, "%s is not in the ctypes signature table" % name
        assert re.search(r"\b%s\(" % name, hdr), "%s is not declared in include/mci.h" % name
        assert hasattr(L, name), "%s is not exported by libmci_hip.so" % name
        argt = argt.strip()
        # split top-level commas of the Julia argument-type tuple
        depth, n, cur = 0, 0, ""
        for ch in argt:
            if ch in "{(":
                depth += 1
            elif ch in "})":
                depth -= 1
            if ch == "," and depth == 0:
                n += 1 if cur.strip() else 0
                cur = ""
            else:
                cur += ch
        n += 1 if cur.strip() else 0
        assert n == len(sigs[name]), "%s: the Julia ccall passes %d arguments, the C function takes %d" % (name, n, len(sigs[name]))
        res = dict((nm, r) for nm, r, _ in _lib.SIGNATURES)[name]
        assert (ret == "Cvoid") == (res is None), (name, ret, res)
    # the wrapper covers what a drop-in user of the reference needs from the library
    for needed in ("mci_integrate", "mci_problem_create", "mci_set_integrand_source", "mci_set_measure_source", "mci_set_integrand_host",
                   "mci_save_state", "mci_load_state", "mci_get_acceptance", "mci_get_reweight", "mci_average", "mci_comm_unique_id",
                   "mci_comm_init", "mci_comm_rank", "mci_get_grid"):
        assert needed in seen, needed


def test_header_library_and_ctypes_table_list_the_same_symbols():
    hdr = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    declared = set(re.findall(r"\b(mci_\w+)\s*\(", hdr)) - {"mci_host_integrand_fn"}
    table = {name for name, _r, _a in _lib.SIGNATURES}
    assert declared == table, (sorted(declared - table), sorted(table - declared))
    L = mci.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_the_julia_tracer_writes_the_python_tracers_grammar():
    """The Julia binding traces closures into device source with the same body grammar as mcintegration_jl_amd/trace.py (`Sym <: Real`
    + operator overloads instead of numpy object arrays): the two format tables, the operand-kind table and the math functions written
    out by name must be the same, so that the same closure gives the same text -- and the same kernel-cache entry -- from either side."""
    from mcintegration_jl_amd import trace
    src = open(JL).read()

    def jdict(name):
        m = re.search(r"const %s = Dict\((.*?)\)\n" % name, src, flags=re.S)
        assert m, name
        return dict(re.findall(r'"([^"]+)"\s*=>\s*"([^"]*)"', m.group(1)))
    assert jdict("C_FORMAT") == trace.C_FORMAT
    assert jdict("C_OPERANDS") == trace.C_OPERANDS
    m = re.search(r"const C_FUNCS = \((.*?)\)\n", src, flags=re.S)
    jfuncs = set(re.findall(r":(\w+)", m.group(1)))
    # (erf lives in SpecialFunctions.jl, not in Base; rint / trunc / fmod -- Julia's round, trunc, mod, fld, rem -- came to trace.py in round 6,
    # after the Julia file was frozen: INTEGRATION.md section 2, item 8 of the first things to try)
    pyfuncs = {trace._CNAME.get(f, f) for f in trace._FUNCS} - {"erf", "erfc", "rint", "trunc"}
    assert jfuncs == pyfuncs, (sorted(jfuncs ^ pyfuncs))
    for needle in ("struct Sym <: Real", "function trace_integrand(", "function hoist(", "function emit(", "Base.literal_pow(::typeof(^), a::Sym",
                   "Base.ifelse(c::Sym", "integrand = trace_integrand(integrand, config"):
        assert needle in src, needle
    # temporaries are named after 0-based node ids on both sides, comparisons are `const int`, and a comparison used as a number is cast
    assert '"const int t" : "const double t"' in src and '"(double)" * name[n]' in src and '" != 0.0)"' in src


def test_julia_binding_calls_closures_in_the_form_the_solver_calls_them():
    """the same three edits as integrate.py (tests/test_callback_forms.py), checked statically -- no Julia in the image: the form follows
    solver + `inplace` (src/main.jl:26-28), the in-place form has its trampoline over mci_set_integrand_host and its tracer branch, and
    nothing picks a form by counting a closure's methods' parameters any more"""
    src = open(JL).read()
    for needle in ("function callback_form(f::Function, solver::Symbol, inplace::Bool=false; what::Symbol=:integrand)",
                   "form = solver == :mcmc ? :indexed : (inplace && what == :integrand) ? :inplace : :plain",
                   "inplace::Bool=false, reweight_goal", "callback_form(integrand, solver, inplace)", "callback_form(measure, solver; what=:measure)",
                   "function _host_inplace_trampoline(", "struct WeightRows <: AbstractVector{Any}", "elseif form == :inplace",
                   "trace_integrand(integrand, config; indexed=form == :indexed, inplace=form == :inplace)", "throw(ArgumentError("):
        assert needle in src, needle
    assert "nargs - 1 >= 3" not in src and "nargs - 1 >= 5" not in src      # (round 5: the form was read off the parameter count)
    # every @cfunction signature of the in-place trampoline is the host-integrand callback type of include/mci.h
    assert src.count("@cfunction(_host_inplace_trampoline, Cint, (Ptr{Float64}, Ptr{Float64}, Int64, Int32, Int32, Ptr{Cvoid}))") == 1


def test_debug_hooks_are_not_part_of_the_public_header():
    """test and development hooks live in csrc/mci_debug.h: the drop-in boundary (include/mci.h) declares none of them, the bindings call
    none of them"""
    hdr = open(HDR).read()
    dbg = open(os.path.join(ROOT, "mcintegration.jl_amd", "csrc", "mci_debug.h")).read()
    names = re.findall(r"\b(mci_debug_\w+)\s*\(", dbg)
    assert len(names) >= 4 and "mci_debug" not in hdr and "mci_debug" not in open(JL).read()
    L = mci.lib()
    for n in names:
        assert hasattr(L, n) and n in {s[0] for s in _lib.DEBUG_SIGNATURES}, n
    assert "test hook" not in hdr


def test_julia_file_is_balanced():
    """cheap syntax sanity without a Julia parser: block openers and `end`s balance, brackets balance outside strings"""
    src = re.sub(r'"""(.*?)"""', '""', open(JL).read(), flags=re.S)
    code = []
    for line in src.splitlines():
        line = re.sub(r'"(?:\\.|[^"\\])*"', '""', line)
        line = re.sub(r"#.*", "", line)
        code.append(line)
    text = "\n".join(code)
    for a, b in ("()", "[]", "{}"):
        assert text.count(a) == text.count(b), (a, text.count(a), text.count(b))
    openers = 0
    for line in code:
        st = line.strip()
        if re.match(r"(?:function|if|for|while|struct|mutable struct|module|try|let|quote|begin)\b", st):
            openers += 1                                           # a block statement (comprehension `for`s and `&&` one-liners are not)
        elif re.search(r"\b(?:do|begin)\s*(?:\|[^|]*\|)?\s*$", st) or re.search(r"\bdo\s+\w+\s*$", st):
            openers += 1                                           # `... do x` / `GC.@preserve a b begin`
    ends = len(re.findall(r"(?<![\w.:\[])end\b", text))
    assert openers == ends, (openers, ends)
