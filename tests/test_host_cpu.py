"""CPU (no GPU): the C-ABI library loads and exports every declared symbol, the ctypes table matches
the header, the mirrored modules reproduce the reference's state-dict keys / shapes / parameter counts /
default initialisation, error behaviour, and the data-parallel gradient exchange over gloo (world size 2)."""
import json
import os
import re

import pytest
import torch

from tests import helpers as Hh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    hdr = open(os.path.join(ROOT, "include", "vipformer_hip.h")).read()
    return {m.group(1): m.group(2) for m in re.finditer(r"\b(?:int|const char\*) (vpf_[a-z0-9_]+)\((.*?)\);", hdr, re.S)}


def test_library_exports_every_declared_symbol():
    from vipformer_amd import _lib, build
    build.build(verbose=False)
    lib = _lib.lib()
    fns = _header_functions()
    assert len(fns) >= 40
    for name in fns:
        assert hasattr(lib, name), f"{name} declared in include/vipformer_hip.h but not exported"
    assert lib.vpf_version() >= 200
    import ctypes
    lib.vpf_build_id.restype = ctypes.c_char_p
    assert lib.vpf_build_id().decode() == "VPF_BUILD_ID=" + build.source_hash()      # the .so was compiled from THIS tree
    assert _lib.lib().vpf_strerror(-3).decode().startswith("unsupported")
    # no packed-fp32 VALU instruction anywhere in the device code (the co-residency fault's trigger)
    # ... and no scalar load whose address is split over a base pair and an offset register (NOTES.md round 5: gfx950 truncates each part)
    assert build.check_no_packed_f32(build.LIB) >= 8
    hit = "\ts_load_dwordx2 s[24:25], s[4:5], s21 offset:0x44\n"
    assert build.SPLIT_SLOAD.findall(hit) and not build.SPLIT_SLOAD.findall("\ts_load_dwordx2 s[24:25], s[4:5], 0x44\n\ts_load_dword s6, s[0:1], 0xce0\n")
    for reg in ("s7", "m0", "vcc_lo", "vcc_hi", "ttmp4"):                      # every register an offset can sit in, with and without an immediate
        assert build.SPLIT_SLOAD.findall(f"\ts_load_dword s6, s[0:1], {reg}\n"), reg
        assert build.SPLIT_SLOAD.findall(f"\ts_buffer_load_dwordx4 s[8:11], s[0:3], {reg} offset:0x10\n"), reg


def test_ctypes_table_matches_header():
    from vipformer_amd import _lib
    fns = _header_functions()
    kind = {_lib.VP: "p", _lib.I: "i", _lib.L_: "l", _lib.F: "f", _lib.U32: "u"}
    for name, sig in _lib.SIGS.items():
        params = [p.strip() for p in fns[name].replace("\n", " ").split(",") if p.strip() != "void"]
        want = []
        for p in params:
            if "*" in p:
                want.append("p")
            else:
                t = p.split()[1] if p.startswith("const") else p.split()[0]
                want.append({"int": "i", "long": "l", "float": "f", "uint32_t": "u"}[t])
        assert [kind[t] for t in sig] == want, name
    assert set(fns) - set(_lib.SIGS) <= {"vpf_version", "vpf_strerror", "vpf_build_id", "vpf_operand_dtype", "vpf_debug_set", "vpf_debug_get", "vpf_sa_layer_pgrad_rows", "vpf_sa_layer_pgrad_rows_h", "vpf_adapter_kv_pgrad_rows"}
    # the 16-bit operand type the library was built for is the one the Python side allocates (fp16: the reference's autocast dtype)
    import torch as _t
    from tests import helpers as _Hh
    assert _lib.lib().vpf_operand_dtype() == 1 and _lib.H16 == _t.float16 == _Hh.H16
    # the launch-time knobs: one struct, read by name, environment consulted once (csrc/api.hip)
    assert _lib.debug_get("knn_select") == 1 and _lib.debug_get("wgroup_cfg") == 2
    _lib.debug_set("knn_select", 0)
    assert _lib.debug_get("knn_select") == 0
    _lib.debug_set("knn_select", 1)
    import pytest as _pt
    with _pt.raises(_lib.VpfError):
        _lib.debug_get("no_such_knob")


def test_ctypes_structs_match_the_library_layout():
    """The argument structs of the fused entry points: the ctypes mirrors must have the size the library was compiled
    with (vpf_abi_sizeof), and _lib.SIGS entries that take a struct pass it as one pointer."""
    import ctypes
    from vipformer_amd import _lib, build
    build.build(verbose=False)
    lib = _lib.lib()
    for which, cls in enumerate((_lib.PackJob, _lib.SaLayerFwd, _lib.WgradJob, _lib.SaLayerBwd, _lib.PgradJob, _lib.AdapterKv, _lib.AdapterKvBwd,
                                 _lib.CaFront)):
        assert lib.vpf_abi_sizeof(which) == ctypes.sizeof(cls), cls.__name__
    assert lib.vpf_abi_sizeof(99) == -1


def _build(name):
    from vipformer_amd.train import build_models
    a = Hh.ARCHS[name]
    return build_models(D=a["D"], H=a["H"], G=a["G"], K=a["K"], S=a["S"], MR=a["MR"], N=a["N"], img=a["img"], patch=a["patch"],
                        device="cpu")


@pytest.mark.parametrize("name", ["tiny", "tiny2", "c1", "c3", "c4", "ref144", "ref144m4"])
def test_state_dict_keys_shapes_and_counts_match_reference(name):
    pc, im = _build(name)
    assert [(k, tuple(v.shape)) for k, v in pc.state_dict().items()] == Hh.load_keyshapes(f"keys_pc_{name}.json")
    assert [(k, tuple(v.shape)) for k, v in im.state_dict().items()] == Hh.load_keyshapes(f"keys_img_{name}.json")
    c = json.load(open(os.path.join(Hh.GOLDEN_DIR, "param_counts.json")))[name]
    assert sum(p.numel() for p in pc.parameters()) == c["pc_params"]
    assert sum(p.numel() for p in im.parameters()) == c["img_params"]
    assert [k for k, _ in pc.named_parameters()] == c["pc_named"] and [k for k, _ in im.named_parameters()] == c["img_named"]
    # cross_attn_1 IS cross_attn_n (partseg.py:297-298): one parameter set under two prefixes
    assert pc.encoder.cross_attn_1 is pc.encoder.cross_attn_n
    sd = pc.state_dict()
    assert sd["encoder.cross_attn_1.0.module.q_norm.weight"].data_ptr() == sd["encoder.cross_attn_n.0.module.q_norm.weight"].data_ptr()
    # reference checkpoints load strictly
    pc.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pc_{name}.json"), 1), strict=True)
    im.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes(f"keys_img_{name}.json"), 2), strict=True)


def test_paper_parameter_counts():
    """assets/tab1.png / tab2.png: 5.1 M (E1CL8SL-H4D256-L128-MR2) and 16.7 M (E1CL8SL-H6D384-L128-MR4)."""
    assert sum(p.numel() for p in _build("c3")[0].parameters()) == 5127040
    assert sum(p.numel() for p in _build("c4")[0].parameters()) == 16654336
    assert sum(p.numel() for p in _build("c1")[0].parameters()) == 4074368


@pytest.mark.parametrize("name", ["tiny", "c1"])
def test_default_initialisation_matches_reference_construction_order(name):
    """Same seed -> same parameters as the reference's build_model (modules are created in the same order
    with the same initialisers), so a run is reproducible against the reference from torch.manual_seed alone."""
    ref = json.load(open(os.path.join(Hh.GOLDEN_DIR, "init_checksums.json")))[name]
    torch.manual_seed(1)
    pc, im = _build(name)
    for model, want in ((pc, ref["pc"]), (im, ref["img"])):
        got = {k: float(v.double().sum()) for k, v in model.state_dict().items()}
        assert got.keys() == want.keys()
        for k in want:
            assert abs(got[k] - want[k]) <= 1e-9 * max(1.0, abs(want[k])), k


def test_error_behaviour_matches_reference_and_no_cpu_fallback():
    from vipformer_amd import _lib
    from vipformer_amd.model.pointcloud import partseg as P
    from vipformer_amd.model.pointcloud import utils as U
    with pytest.raises(ValueError):
        P.MultiHeadAttention(3, 64, 64, 64)                       # partseg.py:39-40
    with pytest.raises(ValueError):
        P.Encoder(64, num_cross_attention_layers=0, dpr_list=[0.0] * 6)   # partseg.py:279-280
    mha = P.MultiHeadAttention(1, 64, 64, 64)
    x = torch.zeros(1, 4, 64)
    with pytest.raises(NotImplementedError):
        mha(x, x, attn_mask=torch.zeros(1))                       # partseg.py:64-65
    with pytest.raises(_lib.VpfError, match="pad_mask"):
        mha(x, x, pad_mask=torch.zeros(1, 5, dtype=torch.bool))   # partseg.py:73-76: [B, Lkv]; a mask of another shape is refused before any launch
    with pytest.raises(_lib.VpfError):
        mha(x, x)                                                 # CPU tensors: there is no eager fallback
    with pytest.raises(_lib.VpfError):
        U.divide_patches(torch.zeros(1, 64, 3), 4, 4)
    with pytest.raises(_lib.VpfError):
        U.Group2Emb(64)(torch.zeros(1, 2, 4, 3))


def _vipformer_modules_saved():
    import sys
    return {k: v for k, v in sys.modules.items() if k == "vipformer" or k.startswith("vipformer.")}


def _vipformer_modules_restore(saved):
    import sys
    for k in [k for k in sys.modules if k == "vipformer" or k.startswith("vipformer.")]:
        del sys.modules[k]
    sys.modules.update(saved)


def test_install_as_vipformer_standalone():
    """No reference checkout on sys.path (the GPU box): every import root utils.py:12-16 and pretrain.py:27 make must resolve --
    the implemented names to this package, the names outside the hot path to placeholders that fail loudly when USED."""
    import sys
    import vipformer_amd
    from vipformer_amd import _lib
    saved = _vipformer_modules_saved()
    path = list(sys.path)
    try:
        _vipformer_modules_restore({})
        sys.path[:] = [p for p in sys.path if not os.path.isdir(os.path.join(p, "vipformer"))]
        assert vipformer_amd.install_as_vipformer() == "standalone"
        # the import lines of the reference's utils.py:12-16
        from vipformer.model.core import PerceiverEncoder, PerceiverEncoder_feats_head  # noqa: F401
        from vipformer.model.core import PerceiverDecoder, PerceiverIO, ClassificationOutputAdapter  # noqa: F401
        from vipformer.model.image import ImageInputAdapter  # noqa: F401
        from vipformer.model.pointcloud import PointCloudInputAdapter, CrossFormer_partseg, CrossFormer_semseg  # noqa: F401
        from vipformer.model.pointcloud import CrossFormer_pc_mp, CrossFormer_img_mp, CrossFormer_pc_mp_ft  # noqa: F401
        from vipformer.model.pointcloud.utils import Group2Emb, divide_patches, farthest_point_sample, knn_point  # noqa: F401
        from vipformer.preproc import fps  # noqa: F401
        assert CrossFormer_pc_mp.__module__.startswith("vipformer_amd") and CrossFormer_partseg.__module__.startswith("vipformer_amd")
        with pytest.raises(_lib.VpfError):
            PerceiverEncoder()
        with pytest.raises(_lib.VpfError):
            CrossFormer_semseg()
    finally:
        sys.path[:] = path
        _vipformer_modules_restore(saved)


@pytest.mark.skipif(not os.path.isdir("/root/reference/vipformer"), reason="needs the reference checkout (build container only)")
def test_install_as_vipformer_overlays_a_reference_checkout():
    """With the reference importable, it stays in place: only the implemented names are replaced, everything else
    (vipformer.model.core, .image, semseg) is the reference's own (the drop-in must not hide those)."""
    import sys
    import types
    import torch.nn as nn
    import vipformer_amd
    saved = _vipformer_modules_saved()
    path = list(sys.path)
    shims = {}
    try:
        _vipformer_modules_restore({})
        fs = types.ModuleType("fairscale"); fsnn = types.ModuleType("fairscale.nn"); fsnn.checkpoint_wrapper = lambda m, *a, **k: m; fs.nn = fsnn
        timm = types.ModuleType("timm"); tm = types.ModuleType("timm.models"); tl = types.ModuleType("timm.models.layers")
        tl.DropPath = type("DropPath", (nn.Identity,), {}); timm.models = tm; tm.layers = tl
        shims = {"fairscale": fs, "fairscale.nn": fsnn, "timm": timm, "timm.models": tm, "timm.models.layers": tl}
        shims = {k: v for k, v in shims.items() if k not in sys.modules}
        sys.modules.update(shims)
        sys.path.insert(0, "/root/reference")
        assert vipformer_amd.install_as_vipformer() == "overlay"
        import vipformer.model.core as core
        import vipformer.model.pointcloud as pcd
        from vipformer.model.pointcloud import utils as RU
        assert core.PerceiverEncoder.__module__ == "vipformer.model.core.modules"          # the reference's own
        assert pcd.CrossFormer_semseg.__module__ == "vipformer.model.pointcloud.semseg"
        assert pcd.CrossFormer_pc_mp.__module__.startswith("vipformer_amd")
        assert RU.divide_patches.__module__.startswith("vipformer_amd") and RU.Group2Emb.__module__.startswith("vipformer_amd")
    finally:
        sys.path[:] = path
        for k in shims:
            sys.modules.pop(k, None)
        _vipformer_modules_restore(saved)


# ------------------------------------------------------------------------------------------ data parallel over gloo
def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    from oracle import torch_oracle as O
    from vipformer_amd.train import GradientExchange
    dist.init_process_group("gloo", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
    # the REAL exchange object of Pretrainer (train.GradientExchange), on CPU tensors over gloo: two regions launched
    # asynchronously in backward-completion order (image model first), finished one by one
    n, cut = 1000, 616
    g = torch.Generator().manual_seed(10 + rank)
    flat_g = torch.randn(n, generator=g)
    mine = flat_g.clone()
    ex = GradientExchange(flat_g, [("img", cut, n), ("pc", 0, cut)], world)
    ex.start("img")
    after_first_launch = flat_g[:cut].clone()                    # the pc region is still local while img travels
    ex.start("pc")
    ex.finish("img"); ex.finish("pc")
    gathered = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(gathered, mine)
    ok_sum = torch.allclose(flat_g, sum(gathered), atol=1e-6) and torch.equal(after_first_launch, mine[:cut])
    try:
        GradientExchange(flat_g, [("a", 0, 10), ("b", 12, n)], world)
        ok_sum = False                                           # regions that do not tile the buffer must be rejected
    except ValueError:
        pass
    # AdamW with grad_scale = 1/world on the SUM == AdamW on the mean of the per-rank gradients
    p0 = torch.linspace(-1, 1, n)
    pa, pb = {"w": p0.clone()}, {"w": p0.clone()}
    O.adamw_step(pa, {"w": flat_g / world}, {}, 1)
    O.adamw_step(pb, {"w": sum(gathered) / world}, {}, 1)
    ok_step = torch.equal(pa["w"], pb["w"])
    # every rank ends with identical parameters
    allp = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(allp, pa["w"])
    ok_same = all(torch.equal(allp[0], t) for t in allp)
    # DDP's per-forward buffer broadcast (train.sync_module_buffers, called by Pretrainer.eval() / probe.extract_features(trainer=...)):
    # every rank ends with rank 0's BatchNorm running statistics and step counter
    from vipformer_amd.train import sync_module_buffers
    bn = torch.nn.BatchNorm1d(16)
    bn.running_mean.fill_(float(rank) + 0.5); bn.running_var.fill_(2.0 + rank); bn.num_batches_tracked.fill_(7 + rank)
    n_b = sync_module_buffers([bn], 0)
    ok_same = ok_same and n_b == 3 and bool((bn.running_mean == 0.5).all()) and bool((bn.running_var == 2.0).all()) and int(bn.num_batches_tracked) == 7
    q.put((rank, ok_sum, ok_step, ok_same))
    dist.destroy_process_group()


def test_gradient_allreduce_world_size_2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] and r[3] for r in res), res


def _dp_split_worker(rank, world, port, wire_bf16, q):
    """The split-capture exchange order of Pretrainer.exchange_and_step (GradientExchange.exchange): early regions travel while
    `between` (the second backward graph) writes the late regions, which follow; AdamW per region on arrival."""
    import torch.distributed as dist
    from vipformer_amd.train import GradientExchange
    dist.init_process_group("gloo", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
    n = 4096
    regions = [("img", 2048, n), ("pc.early0", 0, 256), ("pc.early1", 1024, 2048), ("pc.late0", 256, 1024)]   # Pretrainer's layout
    g = torch.Generator().manual_seed(50 + rank)
    full = torch.randn(n, generator=g)                     # what both backward graphs produce on this rank
    if wire_bf16:
        full = full.bfloat16().float()                     # exactly representable on the wire: the check below stays tight
    flat_g = full.clone()
    flat_g[256:1024] = 0.0                                 # the late region is not written until `between` runs
    ex = GradientExchange(flat_g, regions, world, wire_bf16=wire_bf16)
    arrived, seen_between = [], {}

    def between():
        seen_between["pending"] = sorted(ex._pending)      # the early collectives are in flight, the late one is not launched
        flat_g[256:1024] = full[256:1024]

    def on_arrival(i, name, a, b):
        arrived.append(name)
        assert name not in ex._pending

    ex.exchange(["pc.late0"], between, on_arrival)
    gathered = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(gathered, full)
    want = sum(gathered)
    tol = 2e-2 * world if wire_bf16 else 1e-5              # bf16 wire: the SUM is rounded to 8 bits per hop
    ok_sum = torch.allclose(flat_g, want, atol=tol, rtol=tol)
    ok_order = arrived == [r[0] for r in regions] and seen_between.get("pending") == ["img", "pc.early0", "pc.early1"]
    # every rank holds the same reduced gradient (bitwise: all-reduce delivers one result)
    allg = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(allg, flat_g)
    ok_same = all(torch.equal(allg[0], t) for t in allg)
    # without `between` nothing is late: one launch wave
    flat2 = full.clone()
    GradientExchange(flat2, regions, world, wire_bf16=wire_bf16).exchange(["pc.late0"], None, None)
    ok_plain = torch.allclose(flat2, want, atol=tol, rtol=tol)
    q.put((rank, ok_sum, ok_order, ok_same, ok_plain))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,wire_bf16", [(4, False), (8, False), (4, True)])
def test_split_exchange_order_gloo(world, wire_bf16):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world * 3 + int(wire_bf16)
    procs = [ctx.Process(target=_dp_split_worker, args=(r, world, port, wire_bf16, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(all(r[1:]) for r in res), res


# ------------------------------------------------------------------------------------------ CrossFormer_partseg (config 5)
def _build_partseg(name):
    from vipformer_amd.model.pointcloud import CrossFormer_partseg, PointCloudInputAdapter
    a = Hh.ARCHS[name]
    ad = PointCloudInputAdapter((a["N"], 3), a["D"])
    return CrossFormer_partseg(ad, a["G"], a["D"], a["K"], 1, a["H"], a["S"], a["H"], a["MR"], 0.0, 0.0, 0.0, Hh.PARTSEG_LAYERS[name], 50)


@pytest.mark.parametrize("name", ["tinyseg", "c3"])
def test_partseg_state_dict_matches_reference_and_loads_a_pretraining_checkpoint(name):
    m = _build_partseg(name)
    want = Hh.load_keyshapes(f"keys_partseg_{name}.json")
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == want
    m.load_state_dict(Hh.synth_state_dict(want, 3), strict=True)
    if name == "c3":
        # ft_partseg.py:80-83: a hot-path pc_model_best.pth goes in with strict=False -- 162 shared keys, the 12 latent_head.* are
        # unexpected, the part-segmentation head's keys are missing
        pre = Hh.synth_state_dict(Hh.load_keyshapes("keys_pc_c3.json"), 100)
        res = m.load_state_dict(pre, strict=False)
        assert len(res.unexpected_keys) == 12 and all(k.startswith("latent_head.") for k in res.unexpected_keys)
        assert len(pre) - 12 == 162
        assert len(res.missing_keys) == 33
        assert torch.equal(m.state_dict()["encoder.sa_layers.7.1.module.3.weight"], pre["encoder.sa_layers.7.1.module.3.weight"])
    with pytest.raises(ValueError):
        from vipformer_amd.model.pointcloud import CrossFormer_partseg, PointCloudInputAdapter
        bad = CrossFormer_partseg(PointCloudInputAdapter((64, 3), 64), 8, 64, 8, 1, 1, 3, 1, 2, 0.0, 0.0, 0.0, [1, 2], 50)
        bad(torch.zeros(1, 64, 3), torch.zeros(1, 16))            # the reference only defines 3 or 4 taps (partseg.py:430-435)


# ------------------------------------------------------------------------------------------ checkpoint files
def test_checkpoint_files_interchange_with_the_reference(tmp_path):
    """tests/golden/ckpt_pc_tiny.pth was WRITTEN BY THE REFERENCE (torch.save(module.state_dict()), pretrain.py:283-285; made by
    make_golden.py make_ckpt).  It must load strictly into the mirrored model; a file written by probe.save_best must hold the same
    keys in the same order with the same bytes; and the fine-tuning convention ("module." prefix, strict=False) must work on it."""
    from vipformer_amd import probe
    from vipformer_amd.model.pointcloud import CrossFormer_pc_mp_ft, PointCloudInputAdapter
    ref_file = os.path.join(Hh.GOLDEN_DIR, "ckpt_pc_tiny.pth")
    ref_sd = torch.load(ref_file, map_location="cpu")
    pc, im = _build("tiny")
    res = pc.load_state_dict(ref_sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    want = Hh.synth_state_dict(Hh.load_keyshapes("keys_pc_tiny.json"), 100)
    assert all(torch.equal(ref_sd[k], want[k]) for k in want)
    p1, p2 = probe.save_best(pc, im, str(tmp_path))
    mine = torch.load(p1, map_location="cpu")
    assert list(mine.keys()) == list(ref_sd.keys())
    assert all(torch.equal(mine[k], ref_sd[k]) and mine[k].dtype == ref_sd[k].dtype for k in ref_sd)
    assert os.path.exists(p2)
    # ft_cls.py:92-98: a DDP-style wrapper receives "module."-prefixed keys, strict=False
    a = Hh.ARCHS["tiny"]
    ft = CrossFormer_pc_mp_ft(PointCloudInputAdapter((a["N"], 3), a["D"]), a["G"], a["D"], a["K"], 1, a["H"], a["S"], a["H"], a["MR"],
                              0.0, 0.1, 0.5, True, 40)

    class Wrapper(torch.nn.Module):          # what DistributedDataParallel looks like to load_state_dict
        def __init__(self, m):
            super().__init__()
            self.module = m
    res = probe.load_pretrained(Wrapper(ft), ref_file)
    assert res.unexpected_keys == [] and all(k.startswith("module.finetune_head.") for k in res.missing_keys)
    assert torch.equal(ft.state_dict()["encoder.sa_layers.1.1.module.3.weight"], ref_sd["encoder.sa_layers.1.1.module.3.weight"])


def test_optimizer_step_hook_counts_every_torch_optimizer_step():
    """ops.shadow keys the h16 weight copies on (p._version, data_ptr) AND on a process-wide count of Optimizer.step calls: torch's fused
    optimizers do not bump `_version` (GPU side: test_fused_torch_optimizer_reaches_the_mfma_operands).  The hook must be installed by
    importing the package and fire for any optimizer class."""
    import torch
    from vipformer_amd import ops
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    for opt in (torch.optim.SGD([p], lr=0.1), torch.optim.AdamW([p], lr=0.1), torch.optim.Adam([p], lr=0.1, foreach=True)):
        e = ops._OPT_EPOCH[0]
        opt.step()
        assert ops._OPT_EPOCH[0] == e + 1, type(opt).__name__


def _run_launcher(mode, nproc=2, timeout_s=None):
    """vipformer_amd.launch.launch_ranks in a child python (it prints to ITS stdout / stderr: captured here)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); from vipformer_amd.launch import launch_ranks; "
            "sys.exit(launch_ranks(%r, [%r], %d, timeout_s=%r))" % (root, os.path.join(root, "tests", "_launch_stub.py"), mode, nproc, timeout_s))
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)


def test_launch_ranks_relays_the_json_line_as_the_last_stdout_line():
    """`python bench.py --gpus N` starts its own ranks (the reference: pretrain.py:332-341 mp.spawn).  The launcher with
    a CPU stand-in for bench.py: two ranks over gloo, rank 0 prints the JSON line, then every rank prints more to stdout -- the launcher's
    stdout must hold exactly the JSON line (last), everything else goes to stderr, exit code 0."""
    import json
    r = _run_launcher("ok")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[-1])
    assert d == {"metric": "stub", "ranks_seen": 2, "n_gpus": 2}
    assert "banner of rank 0" in r.stderr and "banner of rank 1" in r.stderr


def test_launch_ranks_propagates_a_failed_rank_and_ends_a_hung_launch():
    """A rank that exits non-zero makes the launch exit non-zero with NO result line on stdout; a rank that never comes back is ended
    after the timeout by killing the session the launcher started (exit code 124), not waited for."""
    import time
    r = _run_launcher("fail")
    assert r.returncode != 0 and r.stdout.strip() == "", (r.returncode, r.stdout)
    t0 = time.time()
    r = _run_launcher("hang", timeout_s=20.0)
    assert r.returncode == 124 and r.stdout.strip() == "", (r.returncode, r.stdout, r.stderr[-2000:])
    assert time.time() - t0 < 120
