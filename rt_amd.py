"""Import shim: the package directory is named `ray-tracer_amd` (hyphen), which is not a
valid identifier.  `import rt_amd as rt` gives the same module object."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_m = importlib.import_module("ray-tracer_amd")
sys.modules[__name__] = _m
