"""dungeon_maps_amd -- MI355X-native depth -> top-down map projector.

Drop-in for the hot path of Ending2015a/dungeon_maps: ``import dungeon_maps_amd
as dmap`` gives ``dmap.MapProjector``, ``dmap.MapBuilder``, ``dmap.TopdownMap``,
``dmap.orth_project`` ... with the reference's signatures, computed by
hand-written HIP kernels for gfx950 behind a C ABI
(include/dungeon_maps_amd.h).  There is no CPU fallback.
"""
from . import utils
from . import functional
from . import maps
from . import parallel

from .maps import *  # noqa: F401,F403
from .maps import MapProjector, TopdownMap, MapBuilder, crop_topdown_map, fuse_topdown_maps
from .functional import CenterMode, get, mask_from_map
from .utils import NINF, Reduction, CameraIntrinsics

__version__ = "0.1.0"
