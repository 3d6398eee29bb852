"""Import shim: loads the package directory `systemlevelcontrol.jl_amd/` as module `slc_amd`.

The product directory keeps the reference's name (with a dot, so `import` cannot spell it);
`import slc_amd` from the repo root gives the package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "systemlevelcontrol.jl_amd")
_spec = importlib.util.spec_from_file_location("slc_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["slc_amd"] = _mod
_spec.loader.exec_module(_mod)
