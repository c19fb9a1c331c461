"""ctypes binding of libvipformer_hip.so (the C ABI in include/vipformer_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails,
this raises.  (The CPU oracle under oracle/ is test infrastructure and is never
imported from here.)
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VPF_LIB", os.path.join(_HERE, "libvipformer_hip.so"))
_lib = None

VP = ctypes.c_void_p
I = ctypes.c_int
F = ctypes.c_float
U64 = ctypes.c_uint64


class VpfError(RuntimeError):
    pass


# The storage dtype of every 16-bit tensor that crosses the C ABI (weights' shadow, activations, gradient operands): IEEE fp16 -- the
# reference's autocast dtype (pretrain.py:154,176); csrc/vpf_common.h VPF_OPERAND_FP16.  VPF_OPERAND=bf16 + a library built with
# `python -m vipformer_amd.build --bf16` is the A/B switch back to rounds 1-3's bf16 (lib() refuses a mismatch).
H16 = torch.bfloat16 if os.environ.get("VPF_OPERAND", "f16") == "bf16" else torch.float16


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VpfError(
                f"{LIB_PATH} not found: build it with `python -m vipformer_amd.build` "
                "(vipformer_amd has no CPU/eager fallback)")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.vpf_strerror.restype = ctypes.c_char_p
        _lib.vpf_strerror.argtypes = [I]
        _lib.vpf_operand_dtype.restype = I
        built = torch.float16 if _lib.vpf_operand_dtype() == 1 else torch.bfloat16
        if built != H16:
            raise VpfError(f"{LIB_PATH} computes with {built} operands but this process expects {H16} (VPF_OPERAND / build --bf16 mismatch)")
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise VpfError(f"{what} failed: {lib().vpf_strerror(rc).decode()} (rc={rc})")


def stream_ptr() -> VP:
    return VP(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> VP:
    if t is None:
        return VP(0)
    return VP(t.data_ptr())


def need_cuda(*ts) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise VpfError("vipformer_amd ops run on an MI355X only (got a CPU tensor); there is no CPU fallback")


# --------------------------------------------------------------------------- signatures
L_ = ctypes.c_long
U32 = ctypes.c_uint32
SIGS = {
    "vpf_fps_f32": [VP, I, I, I, VP, I, VP, VP],
    "vpf_index_points_f32": [VP, I, I, I, VP, I, VP, VP],
    "vpf_square_distance_f32": [VP, I, VP, I, I, I, I, VP, VP],
    "vpf_knn_group_f32": [VP, I, I, I, VP, I, I, I, I, VP, VP, VP, VP],
    "vpf_gemm_h16": [VP, I, L_, VP, I, L_, I, I, I, I, L_, L_, L_, VP, L_, I, I, VP, VP, L_, VP, L_, VP, L_, VP, I, VP, U32, F, I, VP, VP],
    "vpf_gemm_h16_fused": [VP, I, L_, I, VP, VP, VP, VP, I, L_, VP, I, L_, I, VP, VP, I, I, I, VP, L_, I, I, VP, VP, L_, I, I, VP, VP],
    "vpf_cast_f32_h16": [VP, VP, L_, VP],
    "vpf_cast_h16_f32": [VP, VP, L_, VP],
    "vpf_layernorm_fwd": [VP, I, VP, I, VP, VP, VP, VP, VP, VP, L_, I, F, VP],
    "vpf_layernorm_bwd": [VP, VP, I, VP, VP, VP, VP, VP, I, VP, VP, VP, L_, L_, I, VP],
    "vpf_dropout_add_fwd": [VP, VP, VP, L_, VP, U32, F, VP],
    "vpf_dropout_bwd": [VP, VP, L_, VP, U32, F, VP],
    "vpf_dropout_mask": [VP, L_, VP, U32, F, VP],
    "vpf_rng_advance": [VP, VP],
    "vpf_stamp": [VP, I, VP],
    "vpf_wall_clock_khz": [],
    "vpf_colsum": [VP, I, L_, I, VP, VP, VP],
    "vpf_bn_finalize": [VP, VP, L_, I, F, F, I, VP, VP, VP, VP, VP],
    "vpf_bn_affine": [VP, VP, VP, I, VP, VP],
    "vpf_bn_act_fwd": [VP, I, VP, VP, VP, VP, I, L_, I, I, VP],
    "vpf_bn_bwd": [VP, I, VP, I, VP, VP, VP, L_, I, I, I, VP, VP, I, VP, VP, VP],
    "vpf_g2e_fold_bn1": [VP, VP, VP, I, VP, VP, VP],
    "vpf_g2e_fwd_a": [VP, L_, I, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP],
    "vpf_g2e_fwd_b": [VP, L_, VP, VP, VP, I, VP, VP, VP],
    "vpf_g2e_wgrad4": [VP, L_, VP, VP, VP, I, VP, VP, VP],
    "vpf_g2e_bwd": [VP, VP, I, L_, VP, VP, VP, VP, VP, VP, I, VP, VP, VP, VP, VP, VP, VP, VP],
    "vpf_grad_check": [VP, L_, VP, VP],
    "vpf_transpose_h16": [VP, L_, I, I, VP, VP],
    "vpf_group_max_fwd": [VP, L_, I, I, VP, I, VP, VP],
    "vpf_group_max_bwd": [VP, I, VP, L_, I, I, VP, VP],
    "vpf_g2e_concat_fwd": [VP, VP, L_, I, I, VP, VP],
    "vpf_g2e_concat_bwd": [VP, VP, L_, I, I, VP, VP],
    "vpf_pool_fwd": [VP, I, I, I, VP, VP, VP],
    "vpf_pool_bwd": [VP, VP, I, I, I, VP, VP],
    "vpf_sum_rows_f32": [VP, I, I, VP, VP],
    "vpf_axpy_f32": [VP, VP, L_, F, VP],
    "vpf_rowsum_mod_f32": [VP, L_, I, I, VP, VP],
    "vpf_attention_fwd": [VP, L_, VP, L_, VP, L_, I, I, I, I, I, F, F, VP, U32, VP, L_, VP, VP],
    "vpf_attention_bwd": [VP, L_, VP, L_, VP, L_, VP, L_, VP, L_, VP, I, I, I, I, I, F, F, VP, U32, VP, L_, VP, L_, VP, L_, VP, VP],
    "vpf_attention_fwd_pad": [VP, L_, VP, L_, VP, L_, I, I, I, I, I, F, F, VP, U32, VP, L_, VP, VP, VP],
    "vpf_attention_bwd_pad": [VP, L_, VP, L_, VP, L_, VP, L_, VP, L_, VP, I, I, I, I, I, F, F, VP, U32, VP, L_, VP, L_, VP, L_, VP, VP, VP],
    "vpf_adapter_front_fwd": [VP, L_, I, VP, VP, VP, VP, VP, VP],
    "vpf_adapter_front_bwd": [VP, VP, L_, I, VP, VP, VP, VP, VP, VP, VP, VP, VP, L_, VP],
    "vpf_smallk_fwd": [VP, L_, I, VP, VP, I, I, VP, VP],
    "vpf_smallk_bwd": [VP, VP, L_, I, VP, VP, I, I, VP, VP, VP],
    "vpf_g2e_conv1_stats": [VP, L_, I, VP, VP, VP, VP, VP],
    "vpf_g2e_conv1_stats_moments": [VP, L_, I, VP, VP, VP, VP, VP, VP],
    "vpf_group_sum": [VP, L_, I, I, VP, VP],
    "vpf_group_max_scatter_add": [VP, VP, L_, I, I, VP, VP],
    "vpf_g2e_conv1_apply": [VP, L_, I, VP, VP, VP, VP, VP, VP, VP],
    "vpf_g2e_conv1_bwd": [VP, VP, L_, I, VP, VP, VP, VP, VP, I, VP, VP, VP, VP, VP, VP, L_, VP],
    "vpf_g2e_conv1_bwd_fused": [VP, VP, L_, I, VP, VP, VP, VP, VP, I, VP, VP, VP, VP, VP, VP, VP, L_, VP],
    "vpf_patchify": [VP, L_, L_, L_, L_, I, I, I, I, I, VP, VP],
    "vpf_ntxent_fwd": [VP, VP, I, I, F, VP, VP, VP, VP, VP, VP],
    "vpf_ntxent_bwd": [VP, VP, VP, I, I, F, VP, VP, VP, VP],
    "vpf_adamw_step": [VP, VP, VP, VP, VP, L_, VP, I, VP],
    "vpf_g2e_bn1_prepare": [VP, L_, I, VP, VP, VP, VP, VP, F, F, VP, VP, VP, VP, VP, VP, VP, VP],
    "vpf_bn_partials_finalize": [VP, I, I, L_, VP, VP, F, F, VP, VP, VP, VP, VP, VP],
    "vpf_bn_small_fwd": [VP, I, I, VP, VP, F, F, VP, VP, VP, VP, VP, I, VP],
    "vpf_bn_small_bwd": [VP, VP, VP, VP, VP, I, I, I, VP, I, VP, VP, VP],
    "vpf_pretrain_loss_fwd": [VP, VP, I, I, F, F, VP, VP, VP, VP, VP, VP, VP],
    "vpf_pretrain_loss_bwd": [VP, VP, VP, I, I, F, F, VP, VP, VP, VP, VP],
    "vpf_pack_wfrag": [VP, I, VP],
    "vpf_sa_layer_fwd": [VP, VP],
    "vpf_abi_sizeof": [I],
    "vpf_ca_front_fwd": [VP, VP],
    "vpf_wgrad_group": [VP, I, VP, L_, VP],
    "vpf_sa_layer_bwd_mlp": [VP, VP],
    "vpf_sa_layer_bwd_qkv": [VP, VP],
    "vpf_sa_layer_bwd_qkv_mlp": [VP, VP, VP],
    "vpf_ca_front_bwd": [VP, VP],
    "vpf_ca_kv_bwd": [VP, VP],
    "vpf_ln_pgrad_reduce": [VP, I, VP],
    "vpf_adapter_kv_fwd": [VP, VP],
    "vpf_adapter_kv_bwd": [VP, VP],
    "vpf_three_nn_f32": [VP, I, I, I, VP, I, I, VP, VP, VP],
    "vpf_ln_taps_fwd": [VP, VP, VP, VP, I, L_, I, VP, VP, F, VP, VP, VP, VP],
    "vpf_ln_taps_bwd": [VP, VP, VP, VP, VP, I, L_, I, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP],
    "vpf_interp_rows_fwd": [VP, VP, I, I, I, I, I, VP, VP, I, VP, VP],
    "vpf_interp_rows_bwd": [VP, I, I, I, I, I, VP, VP, I, VP, VP],
    "vpf_pad_h16": [VP, I, L_, I, L_, L_, I, VP, VP],
    "vpf_ce_smooth": [VP, L_, VP, L_, I, F, VP, VP, VP, L_, VP],
    "vpf_augment_points": [VP, I, I, I, VP, U32, VP, VP, VP],
    "vpf_image_u8_normalize": [VP, I, I, I, VP, VP, VP, U32, F, VP, VP, VP],
}


class PackJob(ctypes.Structure):
    """struct VpfPackJob (include/vipformer_hip.h)."""
    _fields_ = [("src", VP), ("dst", VP), ("N", I), ("K", I), ("transposed", I), ("pad_", I)]


class WgradJob(ctypes.Structure):
    """struct VpfWgradJob (include/vipformer_hip.h)."""
    _fields_ = [("dy", VP), ("x", VP), ("M", I), ("N", I), ("K", I), ("dW", VP), ("dbias", VP)]


class SaLayerFwd(ctypes.Structure):
    """struct VpfSaLayerFwd (include/vipformer_hip.h) -- same field order."""
    _fields_ = [("B", I), ("L", I), ("chunk_rows", I), ("D", I), ("H", I), ("hidden", I),
                ("qkv", VP), ("base", VP), ("rng", VP),
                ("scale", F), ("p_att", F), ("site_att", U32),
                ("Wo", VP), ("bo", VP),
                ("p_res1", F), ("site_res1", U32),
                ("ln2_g", VP), ("ln2_b", VP),
                ("W1", VP), ("b1", VP), ("W2", VP), ("b2", VP),
                ("p_res2", F), ("site_res2", U32),
                ("o", VP), ("lse", VP),
                ("x1", VP), ("mean2", VP), ("rstd2", VP), ("n2", VP),
                ("u", VP), ("h", VP),
                ("out", VP),
                ("pos", VP), ("pos_rows", I),
                ("ln1n_g", VP), ("ln1n_b", VP), ("Wqkv_next", VP),
                ("mean1n", VP), ("rstd1n", VP), ("n1n", VP), ("qkv_next", VP), ("attention_done", I), ("dbg", VP)]


class PgradJob(ctypes.Structure):
    """struct VpfPgradJob (include/vipformer_hip.h)."""
    _fields_ = [("partials", VP), ("rows", I), ("D", I), ("dgamma", VP), ("dbeta", VP)]


class CaFront(ctypes.Structure):
    """struct VpfCaFront (include/vipformer_hip.h)."""
    _fields_ = [("M", L_), ("D", I), ("hidden", I), ("C", I),
                ("centers", VP), ("W0", VP), ("b0", VP), ("W1", VP), ("b1", VP),
                ("x", VP), ("lnq_g", VP), ("lnq_b", VP), ("Wq", VP),
                ("hpos", VP), ("pos", VP), ("base", VP), ("mean", VP), ("rstd", VP), ("nq", VP), ("q", VP)]


class AdapterKv(ctypes.Structure):
    """struct VpfAdapterKv (include/vipformer_hip.h)."""
    _fields_ = [("M", L_), ("C", I), ("D", I), ("x", VP), ("W1", VP), ("b1", VP), ("ln_g", VP), ("ln_b", VP),
                ("W2", VP), ("b2", VP), ("lnkv_g", VP), ("lnkv_b", VP), ("Wkv", VP),
                ("a1", VP), ("xkv", VP), ("mean", VP), ("rstd", VP), ("nk", VP), ("kv", VP)]


class AdapterKvBwd(ctypes.Structure):
    """struct VpfAdapterKvBwd (include/vipformer_hip.h)."""
    _fields_ = [("M", L_), ("C", I), ("D", I), ("dkv", VP), ("WkvT", VP), ("xkv", VP), ("mean", VP), ("rstd", VP), ("lnkv_g", VP),
                ("W2T", VP), ("dxkv", VP), ("da1", VP), ("pgrad_kv", VP)]


class SaLayerBwd(ctypes.Structure):
    """struct VpfSaLayerBwd (include/vipformer_hip.h) -- same field order."""
    _fields_ = [("M", I), ("D", I), ("hidden", I), ("rng", VP),
                ("p_res1", F), ("site_res1", U32), ("p_res2", F), ("site_res2", U32),
                ("d", VP), ("u", VP), ("x1", VP), ("mean2", VP), ("rstd2", VP), ("ln2_g", VP),
                ("W2T", VP), ("W1T", VP), ("WoT", VP),
                ("dz2", VP), ("du", VP), ("dx1", VP), ("dz1", VP), ("dout_attn", VP),
                ("pgrad2", VP),
                ("dqkv", VP), ("WqkvT", VP), ("base", VP), ("mean1", VP), ("rstd1", VP), ("ln1_g", VP),
                ("dbase", VP), ("dsum", VP),
                ("pgrad1", VP), ("dsum_init", I)]


def call_struct(name: str, struct, *extra) -> None:
    """Entry points that take a pointer to a host-side argument struct (copied into the kernel arguments)."""
    fn = _bound.get(name)
    if fn is None:
        fn = getattr(lib(), name)
        fn.argtypes = SIGS[name]
        fn.restype = I
        _bound[name] = fn
    rc = fn(ctypes.addressof(struct), *extra, torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise VpfError(f"{name} failed: {lib().vpf_strerror(rc).decode()} (rc={rc})")
_bound = {}


def call(name: str, *args) -> None:
    """Invoke a C-ABI entry point on torch's current stream (appended as the last argument);
    tensors are passed by data_ptr; raises VpfError on a non-zero return."""
    fn = _bound.get(name)
    if fn is None:
        fn = getattr(lib(), name)
        fn.argtypes = SIGS[name]
        fn.restype = I
        _bound[name] = fn
    conv = []
    for a in args:
        if a is None:
            conv.append(None)
        elif isinstance(a, torch.Tensor):
            if not a.is_cuda:
                raise VpfError(f"{name}: got a CPU tensor; vipformer_amd runs on an MI355X only (no CPU fallback)")
            conv.append(a.data_ptr())
        else:
            conv.append(a)
    conv.append(torch.cuda.current_stream().cuda_stream)
    rc = fn(*conv)
    if rc != 0:
        raise VpfError(f"{name} failed: {lib().vpf_strerror(rc).decode()} (rc={rc})")


def debug_set(key: str, value: int) -> None:
    """Set one launch-time experiment knob of the library (csrc/vpf_common.h VpfDebug; tests and tools only)."""
    fn = lib().vpf_debug_set
    fn.argtypes = [ctypes.c_char_p, I]
    fn.restype = I
    check(fn(key.encode(), int(value)), f"vpf_debug_set({key})")


def debug_get(key: str) -> int:
    fn = lib().vpf_debug_get
    fn.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    fn.restype = I
    v = ctypes.c_int(0)
    check(fn(key.encode(), ctypes.byref(v)), f"vpf_debug_get({key})")
    return v.value
