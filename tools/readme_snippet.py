"""Runs the quick-start block of README.md as it stands there (needs an MI355X), then examples/histogram_measure.py."""
import os
import re
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
block = re.search(r"```python\n(.*?)```", open(os.path.join(ROOT, "README.md")).read(), re.S).group(1)
ns = {}
exec(compile(block, "README.md", "exec"), ns)
print(ns["res"])
ns["mci"].report(ns["res"].config)
runpy.run_path(os.path.join(ROOT, "examples", "histogram_measure.py"), run_name="__main__")
