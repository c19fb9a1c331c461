"""The drop-in boundary as pretrain.py uses it WITHOUT vipformer_amd.train.Pretrainer (VERDICT r01 row b):

  * DistributedDataParallel around the mirrored models (pretrain.py:104-105): weight gradients come back through autograd, so
    AccumulateGrad runs, ``p.grad`` is populated and the reducer's hooks fire;
  * a torch optimizer + the loop body of pretrain.py:174-211: parameters move, dropout masks are fresh every forward pass and the
    backward pass regenerates the masks of ITS forward pass even if other forward passes ran in between;
  * writes to parameters that bump ``p._version`` (load_state_dict, optimizers) reach the h16 MFMA operands, also for parameters
    a Pretrainer owns.
"""
import os

import pytest
import torch

from tests import helpers as Hh
from tests.test_modules_gpu import build, cosine, forced_start, rel

pytestmark = pytest.mark.gpu


def _batch(a, B, seed=0):
    t1 = Hh.synth_points(seed + 1, B, a["N"]).cuda(); t2 = Hh.synth_points(seed + 2, B, a["N"]).cuda()
    imgs = Hh.synth_images(seed + 3, B, a["img"], a["img"]).cuda()
    start = Hh.synth_start(seed + 4, 2 * B, a["N"]).cuda()
    return t1, t2, imgs, start


def _loop_body(pc, im, t1, t2, imgs, start):
    """pretrain.py:183-207 with the models as the caller holds them (possibly DDP-wrapped)."""
    from vipformer_amd import ops
    b = t1.shape[0]
    with forced_start(start):
        feats = pc(torch.cat([t1, t2]))[0]
    f1, f2 = feats[:b], feats[b:]
    loss_imid = ops.ntxent_loss(f1, f2, 0.1)
    img_feats = im(imgs)[0]
    loss_cmid = ops.ntxent_loss((f1 + f2) / 2, img_feats, 0.1)
    return loss_imid + 1.0 * loss_cmid


def test_torch_optimizer_loop_trains_with_fresh_dropout_masks():
    """pretrain.py:174-211 with torch.optim.AdamW and no Pretrainer: two consecutive training-mode forward passes draw different
    masks, the parameters move, and the h16 operands follow the optimizer's in-place updates."""
    pc, im, a = build("tiny", (0.1, 0.5))
    pc.train(); im.train()
    opt = torch.optim.AdamW(list(pc.parameters()) + list(im.parameters()), lr=1e-3)
    t1, t2, imgs, start = _batch(a, 8)
    with forced_start(start), torch.no_grad():
        b1 = pc(torch.cat([t1, t2]))[1].clone()
        b2 = pc(torch.cat([t1, t2]))[1].clone()
    assert not torch.equal(b1, b2), "two training-mode forward passes drew the same dropout masks"
    losses = []
    before = [p.detach().clone() for p in pc.parameters()]
    for it in range(3):
        opt.zero_grad(set_to_none=True)                                       # pretrain.py:174
        loss = _loop_body(pc, im, t1, t2, imgs, start)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(l == l for l in losses)
    moved = sum(int(not torch.equal(p.detach(), q)) for p, q in zip(pc.parameters(), before))
    assert moved > 0.9 * len(before)
    # eval forward with the updated weights == a fresh model loaded with them (the shadow followed the optimizer)
    pc.eval()
    sd = {k: v.clone() for k, v in pc.state_dict().items()}
    pcf, _, _ = build("tiny", (0.1, 0.5))
    pcf.load_state_dict(sd); pcf.eval()
    with torch.no_grad(), forced_start(start):
        x = pc(torch.cat([t1, t2]))[1]
        y = pcf(torch.cat([t1, t2]))[1]
    assert torch.equal(x, y)


def test_backward_regenerates_the_masks_of_its_own_forward():
    """Residual dropout: out = dropout(y) + res.  d out / d y is the keep mask / (1 - p): it must be the mask of the forward pass
    the gradient belongs to, even when another forward pass of the same site ran in between."""
    from vipformer_amd import ops
    site = ops.new_site()
    y = torch.ones(4, 64, 256, device="cuda", requires_grad=True)
    res = torch.zeros(4, 64, 256, device="cuda")
    o1 = ops.DropoutAddFn.apply(y, res, 0.5, site)
    o2 = ops.DropoutAddFn.apply(y, res, 0.5, site)
    assert not torch.equal(o1, o2)
    (g1,) = torch.autograd.grad(o1.sum(), y, retain_graph=True)
    (g2,) = torch.autograd.grad(o2.sum(), y)
    assert torch.equal(g1 != 0, o1 != 0) and torch.equal(g2 != 0, o2 != 0)
    with ops.rng.pinned():          # a trainer's regime: the owner advances the state, same state -> same mask
        o3 = ops.DropoutAddFn.apply(y, res, 0.5, site)
        o4 = ops.DropoutAddFn.apply(y, res, 0.5, site)
    assert torch.equal(o3, o4)


def test_load_state_dict_after_pretrainer_reaches_the_mfma_operands():
    """the trainer-owned h16 shadow must follow load_state_dict (--resume, best checkpoint before eval)."""
    from vipformer_amd.train import Pretrainer
    pc, im, a = build("tiny")
    tr = Pretrainer(pc, im)                                                  # noqa: F841  (owns the parameters from here on)
    t1, t2, imgs, start = _batch(a, 4)
    other = Hh.synth_state_dict(Hh.load_keyshapes("keys_pc_tiny.json"), 4242)
    pc.load_state_dict(other)
    pc.eval()
    with torch.no_grad(), forced_start(start):
        got = pc(torch.cat([t1, t2]))[1]
    from vipformer_amd.model.pointcloud import CrossFormer_pc_mp, PointCloudInputAdapter
    ref = CrossFormer_pc_mp(PointCloudInputAdapter((a["N"], 3), a["D"]), a["G"], a["D"], a["K"], 1, a["H"], a["S"], a["H"], a["MR"],
                            0.0, 0.0, 0.0, True)
    ref.load_state_dict(other)
    ref = ref.cuda().eval()
    with torch.no_grad(), forced_start(start):
        want = ref(torch.cat([t1, t2]))[1]
    assert torch.equal(got, want)


def test_two_trainers_in_one_process_do_not_alias():
    """ownership lives on the parameters (weak references), dropout sites on module names."""
    from vipformer_amd.train import Pretrainer
    a = Hh.ARCHS["tiny"]
    t1, t2, imgs, start = _batch(a, 4)
    outs = []
    trainers = []
    for _ in range(2):
        pc, im, _ = build("tiny", (0.1, 0.5))
        pc.train(); im.train()
        trainers.append(Pretrainer(pc, im))
    for tr in trainers:                     # interleaved use: each trainer reads ITS OWN shadow and writes ITS OWN gradients
        with forced_start(start):
            losses = tr.forward_backward(t1, t2, imgs.permute(0, 3, 1, 2))
        outs.append((float(losses[0]), tr.flat.g.clone()))
    assert abs(outs[0][0] - outs[1][0]) < 1e-6 * abs(outs[0][0])            # same weights, same sites, same state -> same loss
    assert cosine(outs[0][1], outs[1][1]) > 0.999999
    assert trainers[0].flat.g.data_ptr() != trainers[1].flat.g.data_ptr()
    # the gradient regions of the data-parallel exchange: image model, then the point-cloud parameters whose gradients the first
    # backward graph completes, then the input stages' (one contiguous run each: three collectives per step)
    regs = trainers[0].regions
    assert [n for n, _, _ in regs] == ["img", "pc.early1", "pc.late0"]
    assert sorted((a, b) for _, a, b in regs)[0][0] == 0 and sum(b - a for _, a, b in regs) == trainers[0].flat.numel


def test_host_feeder_pipelines_batches_into_the_captured_step():
    """train.HostFeeder: the batch submitted BEFORE a step is the one that step consumes (losses equal the resident replay's on the same
    data), a new batch can be submitted while the step runs, and unpinned tensors are rejected."""
    from vipformer_amd import _lib
    from vipformer_amd.train import HostFeeder, Pretrainer
    pc, im, a = build("tiny")                                     # dropout 0: the loss is a function of the batch and the weights only
    pc.train(); im.train()
    tr = Pretrainer(pc, im)
    tr.hyper[0] = 0.0; tr.hyper[4] = 0.0                          # lr = wd = 0: the weights stay put, so losses are comparable across steps
    B = 8
    batches = [(Hh.synth_points(10 + i, B, a["N"]), Hh.synth_points(20 + i, B, a["N"]),
                Hh.synth_images(30 + i, B, a["img"], a["img"]).permute(0, 3, 1, 2).contiguous()) for i in range(3)]
    start = Hh.synth_start(4, 2 * B, a["N"]).cuda()
    with forced_start(start):
        static = tr.capture(*(t.cuda() for t in batches[0]), warmup=1)
        want = []
        for t1, t2, im_ in batches:
            for dst, src in zip(static, (t1, t2, im_)):
                dst.copy_(src.cuda())
            want.append(float(tr.replay()[0]))
        feeder = HostFeeder(tr)
        with pytest.raises(_lib.VpfError):
            feeder.submit(*batches[0])                             # not pinned
        pinned = [tuple(t.pin_memory() for t in b) for b in batches]
        got = []
        feeder.submit(*pinned[0])
        for i in range(3):
            loss = feeder.step()[0]
            if i + 1 < 3:
                feeder.submit(*pinned[i + 1])                      # travels while step i runs
            got.append(float(loss))
        with pytest.raises(_lib.VpfError):
            feeder.step()                                          # nothing submitted
        # raw clouds + uint8 images: the augmentation runs on the device in front of the replay
        raw = (Hh.synth_points(50, B, a["N"]) * 3.0 + 1.0).pin_memory()
        u8 = (torch.rand(B, a["img"], a["img"], 3) * 255).to(torch.uint8).pin_memory()
        feeder.submit_raw(raw, u8)
        l_raw = float(feeder.step()[0])
        assert l_raw == l_raw and abs(l_raw) < 1e3
        assert float(static[0].norm(dim=2).max()) < 2.0 * 1.8 + 1.0   # the graph's inputs are augmented clouds: normalised, scaled <= 2, translated, jittered
    assert len({round(w, 4) for w in want}) == 3                   # three different batches give three different losses
    assert all(abs(g - w) < 1e-6 * abs(w) for g, w in zip(got, want)), (got, want)


def test_zero_grad_set_to_none_does_not_detach_a_trainer_from_its_gradients():
    """pretrain.py:174 calls optimizer.zero_grad(set_to_none=True).  On models a Pretrainer owns that drops the flat-buffer views
    installed as ``p.grad``; the next backward pass must put them back (not write into temporaries AdamW never sees)."""
    from vipformer_amd.train import Pretrainer
    pc, im, a = build("tiny")
    pc.train(); im.train()
    tr = Pretrainer(pc, im)
    t1, t2, imgs, start = _batch(a, 4)
    pc.zero_grad(set_to_none=True); im.zero_grad(set_to_none=True)
    assert all(p.grad is None for p in pc.parameters())
    with forced_start(start):
        tr.forward_backward(t1, t2, imgs.permute(0, 3, 1, 2))
    torch.cuda.synchronize()
    assert float(tr.flat.g.abs().max()) > 0.0
    base = tr.flat.g.data_ptr()
    for p, off in zip(tr.flat.params, tr.flat.offsets):
        assert p.grad is not None and p.grad.data_ptr() == base + 4 * off


def test_replayed_step_is_bitwise_reproducible():
    """Race detector: with lr = wd = 0, the dropout step pinned and the FPS start indices a constant of the graph, every replay runs the
    same forward pass on the same weights -- the three losses must be bitwise equal replay after replay (tools/diag_replay_determinism.py;
    a kernel reading memory it did not write, or a missing edge between the two branches, shows up as a second value)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("diag_replay_determinism", os.path.join(root, "tools", "diag_replay_determinism.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main("c2", 8, 150) == 1


def test_pretrainer_loss_scale_backs_off_on_overflow_inside_the_captured_graph():
    """GradScaler semantics on the device (pretrain.py:154,209-211), followed by a REPLAYED hipGraph: started at a loss scale that
    overflows fp16 (2 ** 30), the first replays find inf / NaN in the flat gradient, skip AdamW (parameters, moments and the
    bias-correction counter stay put) and halve the scale; once the gradients fit, the steps train (the loss falls) and the scale stays."""
    from vipformer_amd.train import Pretrainer
    pc, im, a = build("tiny", (0.1, 0.5))
    pc.train(); im.train()
    tr = Pretrainer(pc, im, loss_scale=2.0 ** 30, growth_interval=1000)
    t1, t2, imgs, start = _batch(a, 8)
    tr.capture(t1, t2, imgs.permute(0, 3, 1, 2).contiguous(), warmup=2)
    assert tr.loss_scale == 2.0 ** 30 and tr.skipped_steps == 0 and float(tr.hyper[6]) == 0.0      # the warm-ups changed nothing
    p0 = tr.flat.p.clone()
    scales, losses = [], []
    for it in range(40):
        l = tr.replay()
        scales.append(tr.loss_scale); losses.append(float(l[0]))
        if it == 0:
            assert tr.skipped_steps == 1 and torch.equal(tr.flat.p, p0) and float(tr.hyper[6]) == 0.0, "an overflowed step must change nothing"
    skipped = tr.skipped_steps
    assert 1 <= skipped < 30, skipped
    assert scales[-1] == 2.0 ** 30 * 0.5 ** skipped and scales[-1] == scales[-5]                     # halved once per skipped step, then stable
    assert float(tr.hyper[6]) == 40 - skipped                                                       # AdamW's step counter counts the good steps only
    assert all(l == l for l in losses) and losses[-1] < losses[0] - 0.5, (losses[0], losses[-1])     # and those steps train
    assert torch.isfinite(tr.flat.p).all().item()


def test_pretrainer_at_the_default_loss_scale_backs_off_and_trains():
    """every other test pins VPF_LOSS_SCALE to 256 (tests/conftest.py).  Here the trainer starts at GradScaler's DEFAULT
    65 536 on a 2-pair batch, whose per-sample gradients overflow fp16 at that scale (measured: 4 - 7 skipped steps, scale 512 - 4 096;
    4 pairs: 1; 8 pairs: none): the first replays of the captured graph are skipped (parameters untouched), the scale halves once per
    skipped step inside the graph, then the steps train -- parameters move, the loss falls, `loss_scale` == 65 536 / 2 ** skipped_steps.
    "The loss falls" on TWO pairs under dropout 0.5 is a statement about means: single replays of the trained model scatter between
    0.4 and 5.6 (tools/diag_train_margin.py, 12 dropout seeds: the old `losses[-1] < losses[0] - 0.5` held for 9 of them and failed on
    the round-6 driver rehearsal once in three suites) -- compared are the mean over the SKIPPED replays (all at the initial weights,
    different masks: 3.3 - 4.9) and the mean of the last eight (1.8 - 3.3; smallest difference over the 12 seeds 0.67).  The dropout
    state is seeded per test (tests/conftest.py), so the masks do not depend on which tests ran before."""
    from vipformer_amd.train import Pretrainer
    pc, im, a = build("tiny", (0.1, 0.5))                   # (tests/conftest.py seeds the dropout state per test)
    pc.train(); im.train()
    tr = Pretrainer(pc, im, loss_scale=65536.0, growth_interval=1000)
    t1, t2, imgs, start = _batch(a, 2)
    tr.capture(t1, t2, imgs.permute(0, 3, 1, 2).contiguous(), warmup=2)
    assert tr.loss_scale == 65536.0 and tr.skipped_steps == 0
    p0 = tr.flat.p.clone()
    losses = []
    for _ in range(40):
        losses.append(float(tr.replay()[0]))
    sk = tr.skipped_steps
    assert 1 <= sk < 20, sk
    assert tr.loss_scale == 65536.0 * 0.5 ** sk
    assert float(tr.hyper[6]) == 40 - sk                                   # AdamW counted the good steps only
    assert not torch.equal(tr.flat.p, p0) and torch.isfinite(tr.flat.p).all().item()
    at_start, at_end = sum(losses[:sk]) / sk, sum(losses[-8:]) / 8
    assert all(l == l for l in losses) and at_end < at_start - 0.3, (at_start, at_end, sk, losses)


def test_reference_loop_with_torch_gradscaler_and_autocast_trains():
    """pretrain.py:154,173-211 as it stands on the mirrored modules -- torch's own GradScaler (default init_scale 2 ** 16) around autocast,
    `scaler.scale(loss).backward(); scaler.step(opt); scaler.update()`, one torch.optim.AdamW over both models, no Pretrainer: the
    kernels take the scaled gradients as fp16 operands and turn what does not fit into inf / NaN, so the scaler skips those steps and
    backs its scale off exactly as it does for the reference's autocast modules; the good steps train."""
    from vipformer_amd import ops
    pc, im, a = build("tiny", (0.1, 0.5))
    pc.train(); im.train()
    opt = torch.optim.AdamW(list(pc.parameters()) + list(im.parameters()), lr=1e-3)          # pretrain.py:106,121-124
    scaler = torch.amp.GradScaler("cuda")                                                    # pretrain.py:154
    t1, t2, imgs, start = _batch(a, 8)
    p0 = [p.detach().clone() for p in pc.parameters()]
    losses, scales = [], []
    for it in range(30):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            loss = _loop_body(pc, im, t1, t2, imgs, start)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(float(loss)); scales.append(scaler.get_scale())
    assert all(l == l for l in losses), losses
    assert scales[-1] <= 65536.0 and scales[-1] == scales[-3], scales                       # backed off (or never had to), then stable
    assert all(torch.isfinite(p).all().item() for p in pc.parameters())
    assert any(not torch.equal(a_, b_) for a_, b_ in zip(p0, pc.parameters()))
    assert losses[-1] < losses[0] - 0.5, (losses[0], losses[-1], scales)


def test_reference_loop_body_captured_by_graphed_step_trains():
    """train.GraphedStep on the same loop body: torch's GradScaler + autocast + ONE torch AdamW (fused, capturable: GradScaler.step
    then hands it found_inf on the device) captured into one hipGraph.  Dropout off, so the loss is a function of the weights alone:
    replays train (the loss falls steadily), the scale backs off inside the graph if the fp16 gradients overflow, and the h16 operands
    follow the optimizer -- torch's fused optimizers do not bump `p._version`, the kernels' weight copies are keyed on the count of
    optimizer steps as well (ops._OPT_EPOCH)."""
    from vipformer_amd.train import GraphedStep
    pc, im, a = build("tiny")
    pc.train(); im.train()
    opt = torch.optim.AdamW(list(pc.parameters()) + list(im.parameters()), lr=1e-3, fused=True, capturable=True)
    scaler = torch.amp.GradScaler("cuda")
    scaler.scale(torch.zeros(1, device="cuda"))
    t1, t2, imgs, start = _batch(a, 8)
    out = {}

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            loss = _loop_body(pc, im, t1, t2, imgs, start)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        out["loss"] = loss.detach()

    run = GraphedStep(step, warmup=2)
    losses, scales = [], []
    for _ in range(30):
        run()
        torch.cuda.synchronize()
        losses.append(float(out["loss"])); scales.append(scaler.get_scale())
    assert all(l == l for l in losses), losses
    assert scales[-1] <= 65536.0 and scales[-1] == scales[-3], scales
    assert all(torch.isfinite(p).all().item() for p in pc.parameters())
    assert losses[-1] < losses[0] - 0.5 and losses[-1] < losses[10] < losses[0], (losses[0], losses[10], losses[-1], scales)
    # the replayed steps computed with the weights they updated: an eval forward of the trained model equals a fresh model loaded with them
    pc.eval()
    sd = {k: v.clone() for k, v in pc.state_dict().items()}
    pcf, _, _ = build("tiny")
    pcf.load_state_dict(sd); pcf.eval()
    with torch.no_grad(), forced_start(start):
        x = pc(torch.cat([t1, t2]))[1]
        y = pcf(torch.cat([t1, t2]))[1]
    assert torch.equal(x, y)
    # ... and again after MORE replays (train by replay, evaluate, train, evaluate -- ADVICE r04): the replay steps the optimizer on the
    # device, Python's optimizer hooks do not run, and the second evaluation used to hit the h16 copies the first one had cached
    pc.train()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    pc.eval()
    pcf.load_state_dict({k: v.clone() for k, v in pc.state_dict().items()})
    with torch.no_grad(), forced_start(start):
        x2 = pc(torch.cat([t1, t2]))[1]
        y2 = pcf(torch.cat([t1, t2]))[1]
    assert torch.equal(x2, y2) and not torch.equal(x2, x)


def test_graphed_step_without_warmup_on_a_model_that_has_run_captures_the_weight_casts():
    """warmup = 0 on a model whose h16 weight copies are already cached (it ran an eager forward): the capture must still contain
    the casts, or every replay would compute with the capture-time weights.  Dropout off: the loss of a fixed batch
    falls replay after replay only if the replays see their own updates."""
    from vipformer_amd.train import GraphedStep
    pc, im, a = build("tiny")
    pc.train(); im.train()
    opt = torch.optim.AdamW(list(pc.parameters()) + list(im.parameters()), lr=1e-3, fused=True, capturable=True)
    t1, t2, imgs, start = _batch(a, 8)
    out = {}

    def step():
        opt.zero_grad(set_to_none=True)
        loss = _loop_body(pc, im, t1, t2, imgs, start)
        (loss * 256.0).backward()
        for p in opt.param_groups[0]["params"]:
            if p.grad is not None:
                p.grad.mul_(1.0 / 256.0)
        opt.step()
        out["loss"] = loss.detach()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()                                                  # an eager step: the optimizer's state exists, the h16 caches are filled
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    run = GraphedStep(step, warmup=0)
    losses = []
    for _ in range(12):
        run()
        torch.cuda.synchronize()
        losses.append(float(out["loss"]))
    assert losses[-1] < losses[0] - 0.3 and losses[-1] < losses[5] < losses[0], losses


def test_fused_torch_optimizer_reaches_the_mfma_operands():
    """torch.optim.AdamW(fused=True) updates the parameters without bumping `p._version`; the h16 weight copies must follow it all the
    same (eager loop, no scaler): the loss falls on a fixed batch without dropout, and the trained model equals a fresh one loaded
    with its weights."""
    pc, im, a = build("tiny")
    pc.train(); im.train()
    opt = torch.optim.AdamW(list(pc.parameters()) + list(im.parameters()), lr=1e-3, fused=True)
    t1, t2, imgs, start = _batch(a, 8)
    losses = []
    for _ in range(12):
        opt.zero_grad(set_to_none=True)
        loss = _loop_body(pc, im, t1, t2, imgs, start)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0] - 0.3 and losses[-1] < losses[5] < losses[0], losses
    pc.eval()
    sd = {k: v.clone() for k, v in pc.state_dict().items()}
    pcf, _, _ = build("tiny")
    pcf.load_state_dict(sd); pcf.eval()
    with torch.no_grad(), forced_start(start):
        assert torch.equal(pc(torch.cat([t1, t2]))[1], pcf(torch.cat([t1, t2]))[1])
