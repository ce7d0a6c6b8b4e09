"""A/B builds only (never imported by toad_amd): point THIS process at a tagged library variant before the first call.

    import tools.ab.select_lib            # honours TOAD_HIP_LIB=<path to libtoad_hip_<tag>.so>

The product's loader (toad_amd/_lib.py) reads no environment variable; the override lives here, in the measurement tools' process."""
import os

from toad_amd import _lib

_path = os.environ.get("TOAD_HIP_LIB")
if _path:
    if _lib._lib is not None:
        raise RuntimeError("tools.ab.select_lib must be imported before the first toad_amd call")
    _lib.LIB_PATH = os.path.abspath(_path)
TAG = (os.path.basename(_lib.LIB_PATH).replace("libtoad_hip", "").replace(".so", "") or "(shipped)")
