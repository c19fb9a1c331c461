"""vipformer_amd -- MI355X-native (gfx950) ViPFormer --mp pre-training hot path.

Hand-written HIP kernels behind the reference's own Python interfaces
(``vipformer.model.pointcloud.{utils,partseg,classifier}``), reached through the C ABI in
include/vipformer_hip.h.  There is no CPU / eager fallback: every op raises
``vipformer_amd._lib.VpfError`` if libvipformer_hip.so is missing or given CPU tensors.
"""
__version__ = "0.1.0"


def install_as_vipformer() -> None:
    """Register this package under the reference's import names so that
    ``from vipformer.model.pointcloud import CrossFormer_pc_mp`` (utils.py:13-14, pretrain.py:27)
    and ``vipformer.model.pointcloud.utils.divide_patches`` resolve to the MI355X implementation."""
    import sys
    import types

    from . import model, preproc
    from .model import pointcloud
    from .model.pointcloud import classifier, partseg, utils

    root = types.ModuleType("vipformer")
    root.model, root.preproc = model, preproc
    sys.modules["vipformer"] = root
    sys.modules["vipformer.model"] = model
    sys.modules["vipformer.preproc"] = preproc
    sys.modules["vipformer.model.pointcloud"] = pointcloud
    sys.modules["vipformer.model.pointcloud.utils"] = utils
    sys.modules["vipformer.model.pointcloud.partseg"] = partseg
    sys.modules["vipformer.model.pointcloud.classifier"] = classifier
