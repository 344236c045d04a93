"""Import shim: exposes the directory `awr-adaptive-weighting-regression_amd/` (not a valid Python
identifier) as the package `awr_amd`."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "awr-adaptive-weighting-regression_amd")
_spec = importlib.util.spec_from_file_location("awr_amd", os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["awr_amd"] = _mod
_spec.loader.exec_module(_mod)
