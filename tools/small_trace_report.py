#!/usr/bin/env python3
"""Prints the kernel timeline (start offsets, durations, gaps) of the last 12 kernels in a rocprofv3 rocpd database."""
import glob, os, sqlite3, sys
db = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True))[0]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
rows = rows[-15:]
t0 = rows[0][1]
prev_end = None
for name, s, e in rows:
    print("%-40s start %9.1f us  dur %7.1f us  gap %6.1f us" % (name[:40], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = e
