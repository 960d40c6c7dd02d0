"""Import shim: the package directory is named `mcintegration.jl_amd` (not a valid Python identifier),
so `import mcintegration_jl_amd` loads it from there."""
import importlib.util
import os
import sys

_pkg = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mcintegration.jl_amd")
_spec = importlib.util.spec_from_file_location("mcintegration_jl_amd", os.path.join(_pkg, "__init__.py"),
                                               submodule_search_locations=[_pkg])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mcintegration_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
