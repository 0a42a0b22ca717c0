"""squid-stitch-mi355x: MI355X-native registration-and-fusion core behind the
``Stitcher`` class surface of sohamazing/image-stitcher.

Only the hot path lives here: phase-cross-correlation of
tile-overlap crops, integer placements, and fusion of tiles (optional flatfield
divide) into the TCZYX canvas.  Device work is hand-written HIP for gfx950 in
``csrc/`` behind the C-ABI declared in ``include/squidstitch.h``.
"""
__version__ = "0.1.0"

from .stitcher_parameters import StitchingParameters  # noqa: F401
