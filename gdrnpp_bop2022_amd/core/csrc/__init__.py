"""Import-compatible mirror of the reference's ``core`` package for the hot-path ops.
Overlaying this directory on the reference's ``core/csrc`` makes
``from core.csrc.fps.fps_utils import farthest_point_sampling`` etc. resolve to the HIP library."""
