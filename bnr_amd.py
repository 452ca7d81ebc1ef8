"""Import shim: the package directory is named `bayesiannetworkregression.jl_amd` (after the reference repo), which is
not a legal Python identifier.  `import bnr_amd` loads that directory as the package `bnr_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bayesiannetworkregression.jl_amd")
_spec = importlib.util.spec_from_file_location("bnr_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["bnr_amd"] = _mod
_spec.loader.exec_module(_mod)
