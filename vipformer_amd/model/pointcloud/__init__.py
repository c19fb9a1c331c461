"""Mirror of ``vipformer.model.pointcloud`` (reference __init__.py:1-3 re-exports)."""
from . import utils  # noqa: F401
from .classifier import PointCloudInputAdapter  # noqa: F401
from .partseg import CrossFormer_img_mp, CrossFormer_partseg, CrossFormer_pc_mp, CrossFormer_pc_mp_ft  # noqa: F401
