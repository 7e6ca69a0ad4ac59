"""Import shim: the package directory is named `go-pocket-tts_amd` (not a valid Python
identifier), so it is loaded here under the module name `go_pocket_tts_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_ROOT, "go-pocket-tts_amd")


def load():
    name = "go_pocket_tts_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod
