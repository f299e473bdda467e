"""Import alias for the package directory `g-phocs_amd/` (a hyphen is not importable)."""
import importlib.util
import os
import sys

_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g-phocs_amd", "__init__.py")
_spec = importlib.util.spec_from_file_location("gphocs_amd_pkg", _path,
                                               submodule_search_locations=[os.path.dirname(_path)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gphocs_amd_pkg"] = _mod
_spec.loader.exec_module(_mod)
globals().update({k: v for k, v in vars(_mod).items() if not k.startswith("__")})
