#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database (first *.db under the given directory)."""
import glob, os, sqlite3, sys
dbs = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True))
if not dbs:
    print("no .db under", sys.argv[1]); sys.exit(0)
c = sqlite3.connect(dbs[0])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print("tables:", tabs); sys.exit(0)
cols = [r[1] for r in c.execute("pragma table_info(%s)" % view)]
kn = "kernel_name" if "kernel_name" in cols else "name"
rows = c.execute("select %s, counter_name, sum(value), count(distinct dispatch_id) from %s group by %s, counter_name" % (kn, view, kn)).fetchall()
for name, cn, v, n in rows:
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    print("%-40s %-24s per-dispatch %.4g  (%d dispatches)" % (name[:40], cn, v / max(n, 1), n))
