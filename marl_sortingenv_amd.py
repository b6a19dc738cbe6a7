"""Import alias: the package directory is named `marl-sortingenv_amd/` (not a valid Python
identifier), so `import marl_sortingenv_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "marl-sortingenv_amd")
_spec = importlib.util.spec_from_file_location(
    "marl_sortingenv_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["marl_sortingenv_amd"] = _pkg
_spec.loader.exec_module(_pkg)
