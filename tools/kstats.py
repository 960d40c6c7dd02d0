#!/usr/bin/env python3
"""Print the top_kernels table of a rocprofv3 rocpd database (first *.db under the given directory)."""
import glob, os, sqlite3, sys
dbs = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True))
if not dbs:
    print("no .db under", sys.argv[1]); sys.exit(0)
c = sqlite3.connect(dbs[0])
for name, calls, tot, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    print("%-70s %6d %12.1f us %10.2f us %6.2f%%" % (name[:70], calls, tot / 1e3 if tot > 1e6 else tot, avg / 1e3 if avg > 1e5 else avg, pct))
