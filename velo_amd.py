"""Import shim: exposes the package directory `vision-enhanced-lidar-odometry_amd/` as module `velo_amd`."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "vision-enhanced-lidar-odometry_amd")
_spec = _u.spec_from_file_location("velo_amd", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["velo_amd"] = _mod
_spec.loader.exec_module(_mod)
