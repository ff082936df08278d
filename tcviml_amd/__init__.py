"""Importable name for the package directory `tc-viml_amd/` (a hyphen is not importable): `import tcviml_amd` puts that directory on
`sys.path` and exposes its modules -- `tcviml_amd.tcv` (ctypes mirror of include/tcv.h), `.synth`, `.replay`, `.ate`, `.build`.
Plumbing only: everything numerical is `tc-viml_amd/libtcv_hip.so`."""
import importlib
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tc-viml_amd")
if _DIR not in sys.path:
    sys.path.insert(0, _DIR)


def __getattr__(name):
    if name in ("tcv", "synth", "replay", "ate", "build"):
        return importlib.import_module(name)
    raise AttributeError(name)
