"""The code the documents show must at least name things that exist: the README's quick-start block and the examples are parsed, and
every `mci.<name>` / keyword of `mci.integrate(...)` in them is checked against the package (they RUN on the GPU box:
tools/readme_snippet.py, tests/test_hip_reference_examples.py)."""
import ast
import inspect
import os
import re

import mcintegration_jl_amd as mci

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sources():
    readme = open(os.path.join(ROOT, "README.md")).read()
    yield "README.md", re.search(r"```python\n(.*?)```", readme, re.S).group(1)
    for name in sorted(os.listdir(os.path.join(ROOT, "examples"))):
        if name.endswith(".py"):
            yield "examples/" + name, open(os.path.join(ROOT, "examples", name)).read()
    yield "tools/readme_snippet.py", open(os.path.join(ROOT, "tools", "readme_snippet.py")).read()


def test_documented_code_names_what_exists():
    integrate_kw = set(inspect.signature(mci.integrate).parameters) | set(inspect.signature(mci.Configuration.__init__).parameters)
    seen = 0
    for where, src in _sources():
        tree = ast.parse(src, where)
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id == "mci":
                assert hasattr(mci, node.attr), "%s names mci.%s" % (where, node.attr)
                seen += 1
            if (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and isinstance(node.func.value, ast.Name)
                    and node.func.value.id == "mci" and node.func.attr == "integrate"):
                for kw in node.keywords:
                    assert kw.arg is None or kw.arg in integrate_kw, "%s passes integrate(%s=...)" % (where, kw.arg)
    assert seen > 20


def test_documents_cite_files_that_exist():
    """`tests/...py`, `tools/...`, `profiles/...`, `examples/...`, `csrc/...` paths quoted in the top-level documents"""
    missing = []
    for doc in ("README.md", "DESIGN.md", "INTEGRATION.md", "CHANGELOG.md", "profiles/README.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for m in re.finditer(r"`((?:tests|tools|profiles|examples|oracle|include)/[A-Za-z0-9_./-]+\.(?:py|txt|json|sh|h|c|hip|md))`", text):
            path = m.group(1)
            if doc == "profiles/README.md" and path.startswith("tools/"):
                continue                       # (the per-round index names the script that made a file THEN; one-shot scripts of rounds 1-4 are in git history)
            if not os.path.exists(os.path.join(ROOT, path)) and "*" not in path:
                missing.append((doc, path))
    assert not missing, missing


def test_documents_cite_tests_that_exist():
    import glob
    defs = set()
    for f in glob.glob(os.path.join(ROOT, "tests", "*.py")):
        defs |= set(re.findall(r"def (test_[A-Za-z0-9_]+)", open(f).read()))
    missing = []
    for doc in ("README.md", "DESIGN.md", "INTEGRATION.md", "CHANGELOG.md", "profiles/README.md"):
        for m in re.finditer(r"`(?:tests/[a-z_]+\.py::)?(test_[A-Za-z0-9_]+?)(?:\[[^\]]*\])?`", open(os.path.join(ROOT, doc)).read()):
            name = m.group(1)
            if name not in defs and not any(d.startswith(name.rstrip("_")) for d in defs):      # (a name cut short with `...` is a prefix)
                missing.append((doc, name))
    assert not missing, missing


def test_reference_citations_point_inside_the_cited_files():
    """`src/main.jl:253-264`-style citations all over the repository: the cited file exists in the reference and is at least that long
    (checked where the reference is at hand; the GPU box has no /root/reference)."""
    import glob
    import pytest
    ref = "/root/reference"
    if not os.path.isdir(ref):
        pytest.skip("no reference tree here")
    by_name = {}
    for root, _, names in os.walk(ref):
        for n in names:
            by_name.setdefault(n, []).append(os.path.relpath(os.path.join(root, n), ref))
    length = {}

    def nlines(rel):
        if rel not in length:
            with open(os.path.join(ref, rel), errors="replace") as fh:
                length[rel] = sum(1 for _ in fh)
        return length[rel]
    files = [os.path.join(ROOT, f) for f in ("bench.py", "__graft_entry__.py")]
    for pat in ("include/*.h", "mcintegration.jl_amd/*.py", "mcintegration.jl_amd/csrc/*", "mcintegration.jl_amd/julia/*.jl", "oracle/*.c", "oracle/*.py",
                "oracle/*.h", "tests/*.py", "tools/*.py", "*.md", "profiles/README.md"):
        files += glob.glob(os.path.join(ROOT, pat))
    cite = re.compile(r"((?:[\w.-]+/)*[\w.-]+\.(?:jl|md|c|f)):(\d+)(?:-(\d+))?")
    checked, bad = 0, []
    for path in files:
        if not os.path.isfile(path):
            continue
        with open(path, errors="replace") as fh:
            for ln, line in enumerate(fh, 1):
                for m in cite.finditer(line):
                    name, last = m.group(1), int(m.group(3) or m.group(2))
                    rels = by_name.get(os.path.basename(name))
                    if not rels:
                        continue                     # (one of our own files)
                    rels = [r for r in rels if r.endswith(name)] or rels
                    checked += 1
                    if not any(nlines(r) >= last for r in rels):
                        bad.append((os.path.relpath(path, ROOT), ln, m.group(0)))
    assert checked > 1000 and not bad, bad[:20]
