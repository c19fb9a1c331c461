"""``vipformer.preproc`` aliases for the point-cloud preproc ops.

BASELINE.json's north_star places FPS / kNN grouping under ``vipformer/preproc``; in the
reference they live in ``vipformer/model/pointcloud/utils.py:6-141`` (its ``preproc``
package only holds an unused ImagePreprocessor).  Both import paths resolve to the same
HIP-backed functions here.
"""
from ..model.pointcloud.utils import (divide_patches, farthest_point_sample, fps, index_points,  # noqa: F401
                                      knn_point, square_distance)

__all__ = ["divide_patches", "farthest_point_sample", "fps", "index_points", "knn_point", "square_distance"]
