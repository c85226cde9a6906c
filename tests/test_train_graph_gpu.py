"""GPU: the training step recorded as a hipGraph (mdfnet_hip/graphstep.py) against the same step issued launch by launch.

A recording freezes kernel ARGUMENTS; everything that changes from step to step has to reach the kernels through memory.  The
tests therefore change everything between steps -- images, cameras, depth range, ground truth, Adam's step count -- and compare
with the eager step (train.py:36-45 on the hand-written kernels, itself pinned to the reference's training golden by
tests/test_train_gpu.py).  Eager and replayed steps run the same kernels; what differs is the order of their fp32 / fp64
atomics, so the bars are summation-order level."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from mdfnet_hip import ddp, synth
from mdfnet_hip.graphstep import GraphedTrainStep
from mdfnet_hip.optim import FlatAdam
from modelutil import build_model
from net.loss import Loss

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
W, H, V = 192, 128, 3


def _scene(k):
    imgs, extr, intr, dr = synth.make_scene(W, H, V, batch=1, rot_deg=2.0 + k, seed=11 + 5 * k)
    dr = dr * (1.0 + 0.03 * k)                                    # another depth range per step as well
    rng = np.random.RandomState(100 + k)
    lo, hi = float(dr[0, 0]), float(dr[0, 1])
    gt = {str(s): torch.from_numpy((lo - 20 + (hi - lo + 20) * rng.rand(1, H >> s, W >> s)).astype(np.float32)) for s in (3, 2, 1, 0)}
    return imgs, extr, intr, dr, gt


def _model():
    m = build_model()
    m.load_state_dict(synth.seeded_state_dict(m.state_dict(), seed=1))
    return m.train().to(DEV)


def _eager_step(m, crit, bucket, opt, scene):
    imgs, extr, intr, dr, gt = scene
    out = m(imgs.to(DEV), extr.to(DEV), intr.to(DEV), dr.to(DEV))
    loss = crit(out, {k: v.to(DEV) for k, v in gt.items()}, dr.to(DEV))
    bucket.zero_grad()
    loss.backward()
    bucket.allreduce_gradients()
    opt.step()
    return float(loss)


def _l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def test_adam_launch_with_scalars_in_memory_is_bit_identical():
    """mdf_adam_step_hyper (lr and bias corrections read from a device buffer) against mdf_adam_step (the same scalars as
    arguments), three steps on the same gradients."""
    torch.manual_seed(3)
    nets = [nn.Sequential(nn.Conv2d(3, 8, 3), nn.Conv2d(8, 5, 1), nn.BatchNorm2d(5)).to(DEV) for _ in range(2)]
    nets[1].load_state_dict(nets[0].state_dict())
    buckets = [ddp.FlatBucket(n) for n in nets]
    opts = [FlatAdam(b, lr=3e-3, weight_decay=1e-2) for b in buckets]
    hyper = torch.zeros(3, device=DEV)
    for it in range(3):
        g = torch.randn(buckets[0].flat.numel(), device=DEV) * (10.0 ** -it)
        for b in buckets:
            b.flat.copy_(g)
        opts[0].step()
        hyper.copy_(torch.tensor(opts[1].hyper_values(opts[1].steps + 1)))
        opts[1].step(hyper=hyper)
        opts[1].steps += 1
        for p, q in zip(nets[0].parameters(), nets[1].parameters()):
            assert torch.equal(p, q)
        assert torch.equal(opts[0].exp_avg, opts[1].exp_avg) and torch.equal(opts[0].exp_avg_sq, opts[1].exp_avg_sq)
    assert opts[0].steps == opts[1].steps == 3


def test_replayed_step_follows_its_inputs():
    """Learning rate 0, so the weights stay put and a step's loss and gradients are a function of its inputs alone: every replay,
    on inputs the recording has never seen, must give the eager step's loss and gradients."""
    crit = Loss().to(DEV)
    me, mg = _model(), _model()
    be, bg = ddp.FlatBucket(me), ddp.FlatBucket(mg)
    oe, og = FlatAdam(be, lr=0.0), FlatAdam(bg, lr=0.0)
    s0 = _scene(0)
    step = GraphedTrainStep(mg, crit, bg, og, tuple(t.to(DEV) for t in s0[:4]) + ({k: v.to(DEV) for k, v in s0[4].items()},), warmup=2)
    assert og.steps == 2
    before = [p.detach().clone() for p in mg.parameters()]
    for k in (1, 2, 0, 3):
        sc = _scene(k)
        le = _eager_step(me, crit, be, oe, sc)
        lg = float(step(sc[0], sc[1], sc[2], sc[3], sc[4]))          # host tensors, as a loader hands them over
        ge, gg = be.flat, bg.flat
        print(f"\nscene {k}: loss eager {le:.6f} graph {lg:.6f}; gradient L2 rel {_l2(gg, ge):.2e}, |g| {float(ge.norm()):.3e}")
        assert abs(lg - le) <= 2e-5 * abs(le)
        assert torch.isfinite(gg).all() and _l2(gg, ge) < 1e-3
    for p, q in zip(mg.parameters(), before):
        assert torch.equal(p, q)
    # a recording is bound to its shapes and to its control-plane layout: other shapes are refused, not replayed on stale addresses
    sc = _scene(1)
    with pytest.raises(RuntimeError, match="shapes differ"):
        step(sc[0][:, :, :, : H // 2], sc[1], sc[2], sc[3], sc[4])
    with pytest.raises(RuntimeError, match="layout changed"):
        step.staging.upload([("cam0", torch.zeros(5))])
    # per-step losses differ from scene to scene by far more than the bar above, i.e. the comparison can tell a frozen input
    assert abs(_eager_step(me, crit, be, oe, _scene(1)) - _eager_step(me, crit, be, oe, _scene(2))) > 1e-2


def test_replayed_training_tracks_eager_training():
    """Five steps each way from the same weights (two of them are the recording's warm-up steps on the example).  Adam turns the
    summation-order noise of a near-zero gradient into a full +-lr step, so two runs of the SAME eager loop already drift apart by up to
    2 lr per step and weight, and the loss follows (lr 1e-3: 1e-5 / 5e-4 / 5e-3 relative after 3 / 4 / 5 steps, measured); the test runs
    at lr 1e-4, where the drift is ten times smaller, with bars that leave it a factor of ten."""
    crit = Loss().to(DEV)
    me, mg = _model(), _model()
    be, bg = ddp.FlatBucket(me), ddp.FlatBucket(mg)
    lr = 1e-4
    oe, og = FlatAdam(be, lr=lr), FlatAdam(bg, lr=lr)
    s0 = _scene(0)
    losses_e = [_eager_step(me, crit, be, oe, s0) for _ in range(2)]
    step = GraphedTrainStep(mg, crit, bg, og, tuple(t.to(DEV) for t in s0[:4]) + ({k: v.to(DEV) for k, v in s0[4].items()},), warmup=2)
    v0 = next(mg.parameters())._version
    losses_g = []
    for k in (1, 2, 3):
        sc = _scene(k)
        losses_e.append(_eager_step(me, crit, be, oe, sc))
        losses_g.append(float(step(*sc)))
    print("\neager", losses_e[2:], "\ngraph", losses_g)
    assert oe.steps == og.steps == 5
    assert next(mg.parameters())._version > v0                       # caches keyed on the version see the in-graph update
    for a, b in zip(losses_e[2:], losses_g):
        assert abs(a - b) <= 5e-3 * abs(a)
    for (name, p), q in zip(mg.named_parameters(), me.parameters()):
        assert float((p - q).abs().max()) <= 2 * lr * 5 * 1.01, name          # Adam moves a weight by <= lr per step, either way
    for (name, b1), b2 in zip(mg.named_buffers(), me.buffers()):
        if name.endswith("num_batches_tracked"):
            assert int(b1) == int(b2), name
        else:
            assert _l2(b1, b2) < 2e-2, name
    # the trained weights are what an eval forward of the same module now uses (folded-weight caches keyed on the versions)
    imgs, extr, intr, dr, _ = _scene(2)
    fresh = build_model()
    fresh.load_state_dict(mg.state_dict())
    fresh.eval().to(DEV)
    mg.eval()
    with torch.no_grad():
        a = mg(imgs.to(DEV), extr.to(DEV), intr.to(DEV), dr.to(DEV))["depth"]
        b = fresh(imgs.to(DEV), extr.to(DEV), intr.to(DEV), dr.to(DEV))["depth"]
    assert torch.equal(a, b)


def test_split_recording_equals_the_one_graph_recording():
    """The step recorded as eight chain-shaped graphs (forward | the three stages' backward chains on their own streams, the refinement
    net's beside them | pyramid + trunk backward beside the stages' weight gradients | bucket, Adam: layers.StageCuts cuts the autograd
    graph at every stage's features and depth, train_ops.hold_wgrad_flush keeps the stages' weight gradients back)
    against the same step recorded as ONE graph: no sum crosses a cut, so losses and gradients agree to the summation order of the
    atomics both use, on inputs neither recording has seen."""
    from mdfnet_hip import graphstep
    crit = Loss().to(DEV)
    s0 = _scene(0)
    example = tuple(t.to(DEV) for t in s0[:4]) + ({k: v.to(DEV) for k, v in s0[4].items()},)
    steps, buckets = [], []
    for split in (True, False):
        m = _model()
        b = ddp.FlatBucket(m)
        o = FlatAdam(b, lr=0.0)
        graphstep.SPLIT_BACKWARD = split
        try:
            st = GraphedTrainStep(m, crit, b, o, example, warmup=2)
        finally:
            graphstep.SPLIT_BACKWARD = True
        steps.append(st)
        buckets.append(b)
    sp, one = steps
    assert sp.split and len(sp.graph_s) == 3 and len(set(sp.side)) == 3 and None not in (sp.graph_r, sp.graph_c, sp.graph_w, sp.graph_d)
    assert not one.split and one.graph_c is None and not one.graph_s
    for k in (2, 1, 3):
        sc = _scene(k)
        la, lb = float(sp(*sc)), float(one(*sc))
        ga, gb = buckets[0].flat, buckets[1].flat
        print(f"\nscene {k}: loss split {la:.6f} one graph {lb:.6f}; gradient L2 rel {_l2(ga, gb):.2e}")
        assert abs(la - lb) <= 2e-5 * abs(lb)
        assert torch.isfinite(ga).all() and _l2(ga, gb) < 1e-3
        # every parameter received a gradient through the cuts (a cut that lost a branch would leave zeros behind)
        off = 0
        for p in buckets[0].params:
            n = p.numel()
            assert float(ga[off:off + n].abs().max()) > 0 or float(gb[off:off + n].abs().max()) == 0, "a parameter lost its gradient"
            off += n
