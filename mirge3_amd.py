"""Import shim: ``import mirge3_amd`` loads the package kept in ``mirge3.0_amd/``.

The package directory carries the reference's name with its dot (``mirge3.0_amd``),
which is not a valid Python identifier, so it is registered under the importable
alias ``mirge3_amd`` with the directory as its submodule search path.
"""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "mirge3.0_amd")
_spec = importlib.util.spec_from_file_location(
    "mirge3_amd", os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mirge3_amd"] = _mod
_spec.loader.exec_module(_mod)
