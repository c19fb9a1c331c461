"""vipformer_amd -- MI355X-native (gfx950) ViPFormer --mp pre-training hot path.

Hand-written HIP kernels behind the reference's own Python interfaces
(``vipformer.model.pointcloud.{utils,partseg,classifier}``), reached through the C ABI in
include/vipformer_hip.h.  There is no CPU / eager fallback: every op raises
``vipformer_amd._lib.VpfError`` if libvipformer_hip.so is missing or given CPU tensors.
"""
__version__ = "0.3.0"

# names of the reference that this package implements, per reference module
_OVERLAY = {
    "vipformer.model.pointcloud.utils": ("divide_patches", "fps", "farthest_point_sample", "index_points", "knn_point",
                                         "square_distance", "Group2Emb", "PointNetFeaturePropagation", "Sequential"),
    "vipformer.model.pointcloud.partseg": ("MultiHeadAttention", "CrossAttention", "SelfAttention", "CrossAttentionLayer",
                                           "SelfAttentionLayer", "MLP", "Residual", "Encoder", "CrossFormer_pc_mp",
                                           "CrossFormer_img_mp", "CrossFormer_pc_mp_ft", "CrossFormer_partseg"),
    "vipformer.model.pointcloud.classifier": ("PointCloudInputAdapter",),
    "vipformer.model.pointcloud": ("PointCloudInputAdapter", "CrossFormer_pc_mp", "CrossFormer_img_mp", "CrossFormer_pc_mp_ft",
                                   "CrossFormer_partseg"),
}
# names root utils.py:12-16 imports that are OUTSIDE the hot path (Perceiver-IO non-mp variant, per-pixel image
# adapter, S3DIS model): importable placeholders when no reference checkout is present, failing loudly when used
_OUTSIDE = {
    "vipformer.model.core": ("PerceiverEncoder", "PerceiverEncoder_feats_head", "PerceiverDecoder", "PerceiverIO",
                             "ClassificationOutputAdapter", "InputAdapter", "OutputAdapter"),
    "vipformer.model.image": ("ImageInputAdapter", "ImageClassifier"),
    "vipformer.model.pointcloud.semseg": ("CrossFormer_semseg",),
}


def _outside(modname, name):
    from ._lib import VpfError

    class _Outside:
        def __init__(self, *a, **k):
            raise VpfError(f"{modname}.{name} is outside the MI355X hot path (the --mp pre-training step and its fine-tuning "
                           "heads); put the reference checkout on sys.path before install_as_vipformer() and it is used as is")

    _Outside.__name__ = _Outside.__qualname__ = name
    return _Outside


def install_as_vipformer() -> str:
    """Make ``from vipformer.model.pointcloud import CrossFormer_pc_mp ...`` (utils.py:12-16, pretrain.py:27) and
    ``vipformer.model.pointcloud.utils.divide_patches`` resolve to the MI355X implementation.

    * a reference checkout is importable (``import vipformer`` works): it stays in place and only the names this package
      implements are overlaid on its modules -- everything else (``vipformer.model.core``, ``.image``, ``semseg``) falls through
      to the reference.  Returns "overlay".
    * otherwise the package tree is synthesised: the implemented modules are this package's, the rest are placeholders that
      import fine and raise VpfError when constructed.  Returns "standalone".
    ``vipformer.preproc`` additionally carries the FPS / kNN entry points (``north_star`` words the path that way)."""
    import importlib
    import sys
    import types

    from . import model, preproc
    from .model import pointcloud
    from .model.pointcloud import classifier, partseg, utils
    ours = {"vipformer.model.pointcloud.utils": utils, "vipformer.model.pointcloud.partseg": partseg,
            "vipformer.model.pointcloud.classifier": classifier, "vipformer.model.pointcloud": pointcloud}
    real = None
    try:
        real = importlib.import_module("vipformer.model.pointcloud")
        if getattr(real, "__name__", "").startswith("vipformer_amd") or real is pointcloud:
            real = None                                    # an earlier standalone install
    except Exception:
        for k in [k for k in sys.modules if k == "vipformer" or k.startswith("vipformer.")]:
            del sys.modules[k]                             # a half-imported reference must not linger
    if real is not None:
        for modname, names in _OVERLAY.items():
            target = importlib.import_module(modname)
            for n in names:
                if hasattr(ours[modname], n):
                    setattr(target, n, getattr(ours[modname], n))
        pre = importlib.import_module("vipformer.preproc")
        for n in preproc.__all__:
            setattr(pre, n, getattr(preproc, n))
        return "overlay"
    root = types.ModuleType("vipformer")
    root.__path__ = []
    root.model, root.preproc = model, preproc
    sys.modules["vipformer"] = root
    sys.modules["vipformer.model"] = model
    sys.modules["vipformer.preproc"] = preproc
    for modname, mod in ours.items():
        sys.modules[modname] = mod
    for modname, names in _OUTSIDE.items():
        m = types.ModuleType(modname)
        for n in names:
            setattr(m, n, _outside(modname, n))
        sys.modules[modname] = m
        parent, _, leaf = modname.rpartition(".")
        setattr(sys.modules[parent], leaf, m)
    if not hasattr(pointcloud, "CrossFormer_semseg"):
        pointcloud.CrossFormer_semseg = sys.modules["vipformer.model.pointcloud.semseg"].CrossFormer_semseg
    return "standalone"
