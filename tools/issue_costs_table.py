#!/usr/bin/env python3
"""Development tool (GPU box): runs mcintegration.jl_amd/lib/issue_microbench (all rows, 1/2/4/8 waves per SIMD) and writes
profiles/<tag>_issue_costs.json (the fallback table of bench.py's roofline) and profiles/<tag>_issue_costs.txt.
usage: tools/issue_costs_table.py [tag=r02] [iters=2000]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
iters = sys.argv[2] if len(sys.argv) > 2 else "2000"
exe = os.path.join(ROOT, "mcintegration.jl_amd", "lib", "issue_microbench")
out = subprocess.run([exe, iters], check=True, capture_output=True, text=True).stdout
rows = [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]
dev, rows = rows[0], rows[1:]
doc = {"tool": "tools/issue_microbench.hip (mcintegration.jl_amd/lib/issue_microbench %s)" % iters, "device": dev,
       "how": "one instruction form in inline asm, 8 independent destinations, 64 instructions per loop trip, W waves on every SIMD (grid = CUs x W "
              "workgroups of 256 threads); wall_ns_per_wave_inst_per_simd = HIP-event launch time / (instructions per wave x W), an upper bound "
              "(launch + tail included); slope_ns_per_wave_inst_per_simd = (time of 2 x iters - time of iters) / the extra instructions: the fixed "
              "part cancels -- the issue cost bench.py prices the sample loop with (mean over W = 4, 8); cycles_per_wave_inst = the waves' own "
              "s_memtime ticks / (instructions x W).  LDS rows: cost seen from one SIMD with the CU's four SIMDs competing",
       "rows": rows}
with open(os.path.join(ROOT, "profiles", "%s_issue_costs.json" % tag), "w") as fh:
    json.dump(doc, fh, indent=0)
ops = []
for r in rows:
    if r["op"] not in ops:
        ops.append(r["op"])
with open(os.path.join(ROOT, "profiles", "%s_issue_costs.txt" % tag), "w") as fh:
    fh.write("== tools/issue_microbench.hip on MI355X (gfx950): ns a SIMD is occupied per wave64 instruction, W waves per SIMD competing\n"
             "   (slope between two launch lengths; in brackets the single-launch wall figure at W = 8) ==\n")
    fh.write("%-100s %8s %8s %8s %8s\n" % ("instruction form", "W=1", "W=2", "W=4", "W=8"))
    for op in ops:
        v = {r["waves_per_simd"]: r for r in rows if r["op"] == op}
        fh.write("%-100s " % op[:100] + " ".join("%8.3f" % v[w].get("slope_ns_per_wave_inst_per_simd", v[w]["wall_ns_per_wave_inst_per_simd"]) if w in v else "%8s" % "-"
                                               for w in (1, 2, 4, 8)))
        fh.write("   [%.3f]\n" % v[8]["wall_ns_per_wave_inst_per_simd"] if 8 in v else "\n")
print(open(os.path.join(ROOT, "profiles", "%s_issue_costs.txt" % tag)).read())
