"""Import shim: the package directory is ``image-stitcher_amd/`` (a hyphen is not a
legal identifier), so ``import image_stitcher_amd`` loads that directory under the
importable name.  The package is executed exactly once, under this name."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "image-stitcher_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
