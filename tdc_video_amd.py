"""Import alias: the package directory is `tdc-video_amd/` (not a Python identifier); `import tdc_video_amd` loads it."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("tdc-video_amd")
sys.modules[__name__] = _pkg
