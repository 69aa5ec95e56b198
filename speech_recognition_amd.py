"""Import shim: the package directory is ``speech-recognition_amd/`` (a hyphen is
not legal in a module name), so ``import speech_recognition_amd`` is served
from that directory by this file."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "speech-recognition_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
