#!/usr/bin/env python3
"""Summarises the rocprofv3 outputs of profiles/collect.sh (rocpd sqlite databases under gpurun_out/)
into small text/JSON files that are committed under profiles/ (gpurun_out/ itself is scratch).

    python profiles/summarize.py r02 mci_vegas_batch "python bench.py ..."
        reads gpurun_out/prof_*_r02/, prints the summary, writes profiles/r02_kernel_stats.txt + r02_pmc_traffic.json
"""
import glob
import json
import os
import re
import sqlite3
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
KERNELS = (sys.argv[2] if len(sys.argv) > 2 else "mci_vegas_batch").split(",")
CMD = sys.argv[3] if len(sys.argv) > 3 else "python bench.py --steps 10 --warmup 5 --no-cpu-baseline"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out")


def dbs(kind):
    return sorted(glob.glob(os.path.join(out, "prof_%s_%s" % (kind, tag), "**", "*.db"), recursive=True))


def query(kind, sql):
    rows = []
    for f in dbs(kind):
        c = sqlite3.connect(f)
        rows += c.execute(sql).fetchall()
        c.close()
    return rows


lines = []


def emit(s=""):
    print(s)
    lines.append(s)


emit("== rocprofv3 --kernel-trace --stats, tag %s: `%s` ==" % (tag, CMD))
emit("%-64s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
stats = query("stats", "select name, total_calls, total_duration, average, percentage from top_kernels")
for name, calls, tot, avg, pct in stats:
    emit("%-64s %8d %14.1f %12.2f %8.2f" % (name[:64], calls, tot, avg, pct))

# per-call durations in launch order: the first ~16 launches of a process run up to 20 % slower (the GPU coming out of idle:
# tools/ramp_probe.py shows the same ramp on a grid that never adapts, and again after 2 s of idle), so the figure to hold against
# bench.py's HIP-event average -- taken over the timed iterations, after the warm-up -- is the steady state (the median of the
# second half), not the average over warm-up and all
percall = {}
try:
    for name, dur in query("stats", "select name, (end - start) from kernels order by start"):
        percall.setdefault(name, []).append(dur / 1e3)
except Exception as e:  # older rocprofv3 schema
    emit("(no per-call view: %s)" % e)
for K in KERNELS:
    for name, ds in percall.items():
        if name.startswith(K) and len(ds) >= 4:
            half = sorted(ds[len(ds) // 2:])
            emit("%-28s per call, us, launch order: %s%s" % (K, " ".join("%.0f" % d for d in ds[:40]), " ... (%d launches)" % len(ds) if len(ds) > 40 else ""))
            if len(ds) > 60:   # a long run: the average after the idle ramp is the figure bench.py's HIP events see
                tail = ds[32:]
                emit("%-28s average over launches 33..%d = %.1f us" % (K, len(ds), sum(tail) / len(tail)))
            emit("%-28s steady state (median of the second half of the launches) = %.1f us" % (K, half[len(half) // 2]))

summary = {"tag": tag, "kernels": {}, "command": CMD}
# the code object the profiled run loaded (bench.py prints it in config.code_object): PMC numbers are only valid for THAT kernel binary
try:
    log = open(os.path.join(out, "prof_stats_%s.log" % tag)).read()
    m = re.search(r'"code_object": "([^"]+)"', log)
    if m:
        summary["code_object"] = m.group(1)
except Exception:
    pass

for K in KERNELS:
    ks = summary["kernels"].setdefault(K, {})
    for name, calls, tot, avg, pct in stats:
        if name.startswith(K):
            ks["kernel_avg_us"] = avg
            ks["kernel_calls"] = calls
            ds = percall.get(name, [])
            if len(ds) >= 4:
                half = sorted(ds[len(ds) // 2:])
                ks["kernel_steady_us"] = half[len(half) // 2]
    emit()
    emit("== HBM traffic per launch of %s (PMC; FETCH_SIZE and WRITE_SIZE in separate passes) ==" % K)
    sql = "select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%s%%' group by counter_name" % K
    fetch = {r[0]: r for r in query("fetch", sql)}
    write = {r[0]: r for r in query("write", sql)}
    if "FETCH_SIZE" in fetch and "WRITE_SIZE" in write:
        f_kib, w_kib = fetch["FETCH_SIZE"][2], write["WRITE_SIZE"][2]
        # rocprofv3 reports KiB.  gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streams
        # (MI355X_MICROARCH.md "HBM"): double it -- an upper bound for kernels whose few loads are 8-byte gathers.
        f_b, w_b = f_kib * 1024 * 2, w_kib * 1024
        emit("FETCH_SIZE  avg %.1f KiB/launch raw  -> %.4g B after the gfx950 x2 correction" % (f_kib, f_b))
        emit("WRITE_SIZE  avg %.1f KiB/launch      -> %.4g B" % (w_kib, w_b))
        emit("HBM bytes per launch = %.4g" % (f_b + w_b))
        ks.update(hbm_bytes_per_launch=f_b + w_b, fetch_kib_raw_per_launch=f_kib, write_kib_per_launch=w_kib, launches_profiled=fetch["FETCH_SIZE"][1])
    else:
        emit("no counter rows found")
    emit()
    emit("== SQ counters of %s, average per launch ==" % K)
    sq = {r[0]: r[2] for r in query("sq", sql)}
    for k in sorted(sq):
        emit("%-24s %.5g" % (k, sq[k]))
    ks["sq_avg_per_launch"] = sq
    if sq.get("SQ_INSTS_VALU") and sq.get("SQ_WAVES"):
        emit("VALU instructions per wave = %.0f ; LDS instructions per wave = %.0f" % (sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"], sq.get("SQ_INSTS_LDS", 0) / sq["SQ_WAVES"]))

# flat keys of the first kernel: what bench.py's recorded_traffic() reads
first = summary["kernels"].get(KERNELS[0], {})
for k in ("kernel_avg_us", "kernel_steady_us", "kernel_calls", "hbm_bytes_per_launch", "fetch_kib_raw_per_launch", "write_kib_per_launch", "launches_profiled", "sq_avg_per_launch"):
    if k in first:
        summary[k] = first[k]
summary["kernel"] = KERNELS[0]

os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", "%s_kernel_stats.txt" % tag), "w") as fh:
    fh.write("\n".join(lines) + "\n")
with open(os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % tag), "w") as fh:
    json.dump(summary, fh, indent=1)
