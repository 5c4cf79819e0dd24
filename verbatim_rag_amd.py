"""Import shim: `import verbatim_rag_amd` loads the package that lives in the
directory `verbatim-rag_amd/` (a hyphen is not importable as a module name)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "verbatim-rag_amd")
_spec = _ilu.spec_from_file_location(
    "verbatim_rag_amd", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = _ilu.module_from_spec(_spec)
_sys.modules["verbatim_rag_amd"] = _mod
_spec.loader.exec_module(_mod)
