"""A whole training step -- forward, loss, backward, gradient bucket, Adam (train.py:36-45) -- recorded ONCE as a hipGraph and
replayed per step.

Why: at BASELINE config 3 the step is ~330 launches of 5-300 us.  Issued one by one from Python they keep the host busy for the
whole step (~9 ms of issue time for ~9 ms of wall); a replay is one host call (0.45 ms per step including the uploads below), so
the host is free for the loader, for the collective of a multi-rank step and for the next batch's control plane.  It does NOT
shorten the step: its wall clock is the sum of its kernels' durations either way (cfg3: 8.96 ms replayed, 9.0-9.1 ms eager; the rocprof
trace of the eager step shows the kernels back to back).  Opt-in: train.py takes it with MDF_TRAIN_HIPGRAPH=1.

What has to hold for a recording to stay valid, and how each point is met:
  * every address the kernels touch is the same at every replay: activations, gradients and workspaces come from the graph's
    private pool (torch.cuda.graph); the step's INPUTS live in fixed device tensors that `__call__` overwrites, stream-ordered,
    before the replay (images, ground truth, depth range, and the packed control plane -- cameras, relative projections,
    hypotheses row, fit row -- which is host arithmetic: controlplane.host_pieces -> controlplane.Staging.upload);
  * nothing the host computes per step is frozen into a kernel ARGUMENT: Adam's learning rate and bias corrections are read from
    a 3-float device buffer uploaded per replay (mdf_adam_step_hyper), the control plane as above; the step is built from
    hand-written kernels only, none of which synchronises or allocates (the C library never calls hipMalloc / hipMemcpy);
  * the per-step zeroed reduction pool (train_ops.ZeroPool) hands out the same offsets in the same order every step, and its one
    fill per step is part of the recording; a pool that overflowed DURING the recording would have been replaced by a pool
    tensor from the graph's memory, which is equally stable;
  * the parameters change through raw pointers inside the graph, so after every replay the version counters of parameters and
    buffers are bumped on the host: any cache keyed on them (eval-mode folded weights) sees the update.

Stage-parallel backward (default; MDF_TRAIN_GRAPH_SPLIT=0 records the step as ONE graph): the backward chains of the three stages
(regulariser + aggregation, ~60 % of the step's launches, many of them small-volume layers that fill a fraction of the chip) share
nothing until they meet at the feature pyramid.  A hipGraph with parallel branches does not help -- the runtime replays it node
by node from the host (6.5 ms of host time per cfg3 step, slower than the chain) -- so the step is recorded as EIGHT chain-shaped
graphs: F (forward, loss, the loss's own backward) | S0 / S1 / S2 (one stage's input-gradient chain each, recorded on the stage's own
stream into its own memory pool) and R (the refinement net's backward: parameters only) side by side | C (feature pyramid + trunk
backward: a chain of small launches) and W (the three stages' weight gradients, kept back from the stage chains by
train_ops.hold_wgrad_flush: a few chip-filling launches) side by side | D (bucket gather, Adam).  The cuts are made by CoreNet's forward
(layers.StageCuts); the gradients are those of the one-piece backward pass (no sum crosses a cut).

Data parallelism: with more than one rank the gradient all-reduce stays OUTSIDE the graphs -- recording A ends with the bucket
gather, the collective runs eagerly on the same stream (RCCL), recording B is the Adam launch.

The recorded step equals the eager step on the same inputs up to the summation order of the floating-point atomics that both
use (tests/test_train_graph_gpu.py: three steps with different cameras and images per step, both ways).
"""
import os

import torch
from torch.autograd import graph as _graph

from . import controlplane, hostmirror, layers

SPLIT_BACKWARD = bool(int(os.environ.get("MDF_TRAIN_GRAPH_SPLIT", "1")))      # dev A/B: 0 = the whole step as one graph


class GraphedTrainStep:
    def __init__(self, model, criterion, bucket, optimizer, example, warmup=2):
        """example = (imgs, extrinsics, intrinsics, depth_range, gt dict): shapes and device of every later step.  Runs `warmup`
        eager steps on a side stream (lazy initialisation, autograd's buffers, the weight-pack plan), then records.  The model,
        the optimizer state and the BatchNorm buffers ARE advanced by the warm-up steps and by the recording step (a recording
        executes nothing, but the warm-up does): restore a checkpoint afterwards if the example is not a real step."""
        imgs, extr, intr, dr, gt = example
        if not imgs.is_cuda:
            raise RuntimeError("GraphedTrainStep records hand-written MI355X kernels: the example tensors must be on the GPU")
        if not controlplane.builtin_slots(model):
            raise RuntimeError("GraphedTrainStep needs the builtin slot set (scale_cam, HyposByFit): other slots compute their "
                               "control plane where the recording cannot follow")
        if warmup < 1:
            # the weight re-pack launch (train_ops.PackPlan.run) and the per-layer pack caches are keyed on parameter versions: only
            # after one eager optimizer step does a step START with "the weights have moved", which is what every replay must do
            raise ValueError("GraphedTrainStep needs warmup >= 1: a step recorded before any optimizer step would never re-pack weights")
        self.model, self.crit, self.bucket, self.opt = model, criterion, bucket, optimizer
        self.split = SPLIT_BACKWARD
        dev = imgs.device
        self.device = dev
        self.world = bucket.world
        # fixed inputs
        self.imgs = imgs.detach().clone()
        self.dr = dr.detach().clone()
        self.gt = {k: v.detach().clone() for k, v in gt.items()}
        # cameras and range reach the kernels through the packed control plane only; the slot API still wants tensors to key on
        self.extr, self.intr = extr.detach().clone(), intr.detach().clone()
        self.staging = controlplane.Staging(dev)
        self.hyper = torch.zeros(3, device=dev, dtype=torch.float32)
        self.stream = torch.cuda.Stream(dev)
        self.loss = None
        self.graph_a = self.graph_b = None
        self._versioned = list(bucket.params) + [b for b in model.buffers() if b.is_cuda]
        host = [hostmirror.get(t).clone() for t in (extr, intr, dr)]
        cur = torch.cuda.current_stream(dev)
        self.stream.wait_stream(cur)
        self.warmup_loss = None                             # loss of the last warm-up step (a real training step on the example)
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                self._upload(*host)
                self.warmup_loss = self._eager_step().clone()
        cur.wait_stream(self.stream)
        torch.cuda.synchronize(dev)
        self._record(host)

    # ---- pieces of a step ------------------------------------------------------------------------------------------------
    def _upload(self, extr_h, intr_h, dr_h):
        """This step's host-computed values -> the fixed device buffers (stream-ordered copies on the current stream)."""
        for fixed, h in ((self.extr, extr_h), (self.intr, intr_h), (self.dr, dr_h)):
            hostmirror.put(fixed, h)                        # the slots find the host values without a device->host hop
        pieces, _, _ = controlplane.host_pieces(self.model, intr_h, extr_h, dr_h)
        self.staging.upload(pieces)
        hv = torch.tensor(self.opt.hyper_values(self.opt.steps + 1), dtype=torch.float32).pin_memory()
        self.hyper.copy_(hv, non_blocking=True)

    def _forward_backward(self):
        with controlplane.staged(self.staging):
            out = self.model(self.imgs, self.extr, self.intr, self.dr)
        loss = self.crit(out, self.gt, self.dr)
        self.bucket.zero_grad()
        loss.backward()
        self.bucket.gather()
        return loss.detach()

    # the same step in pieces (module docstring): F, one chain per stage, C
    def _piece_f(self):
        with controlplane.staged(self.staging), layers.stage_cuts() as cuts:
            out = self.model(self.imgs, self.extr, self.intr, self.dr)
        if len(cuts.depth) == 0 or cuts.streams is None:
            raise RuntimeError("GraphedTrainStep: the model's forward made no stage cuts (not the HIP training path?)")
        loss = self.crit(out, self.gt, self.dr)
        self.bucket.zero_grad()
        loss.backward()                                     # loss -> d depth of every stage; the refinement net's parameters
        return loss.detach(), cuts

    @staticmethod
    def _piece_stage(cuts, s):
        depth, cut = cuts.depth[s]
        if cut.grad is not None:
            from . import train_ops
            with train_ops.hold_wgrad_flush():              # the stage's weight gradients are launched by piece W, beside piece C
                torch.autograd.backward([depth], [cut.grad])    # regulariser + aggregation of stage s -> d features, parameters

    def _piece_w(self):
        from . import train_ops
        train_ops.flush_held(self.device)                   # the three stages' weight gradients: a few chip-filling launches

    @staticmethod
    def _piece_r(cuts):
        if cuts.refine is not None and cuts.refine[1].grad is not None:
            torch.autograd.backward([cuts.refine[0]], [cuts.refine[1].grad])      # the refinement net's parameter gradients

    @staticmethod
    def _piece_c(cuts):
        roots = [(y, yc.grad) for pairs in cuts.feat for (y, yc) in pairs if yc.grad is not None]
        torch.autograd.backward([r[0] for r in roots], [r[1] for r in roots])      # feature pyramid + trunk (a chain of small launches)

    def _eager_step(self):
        if self.split:
            # the warm-up takes the step in the pieces the recording will (same streams: autograd runs a node's backward on
            # the stream of its forward, and the recording of stage s must find its whole chain on stream s)
            cur = torch.cuda.current_stream(self.device)
            loss, cuts = self._piece_f()
            for s, st in enumerate(cuts.streams):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    self._piece_stage(cuts, s)
            self._piece_r(cuts)
            for st in cuts.streams:
                cur.wait_stream(st)
            self._piece_c(cuts)
            self._piece_w()
            self.bucket.gather()
        else:
            loss = self._forward_backward()
        self.bucket.allreduce_gradients()
        self.opt.step(hyper=self.hyper)
        self.opt.steps += 1
        return loss

    def _record(self, host):
        dev = self.device
        self._upload(*host)                                 # (values for the recording pass: it executes nothing, but the host-side
        torch.cuda.synchronize(dev)                         #  slot code reads the mirrors)
        from . import train_ops
        self._pool = train_ops.step_pool(dev)               # its buffer's address is in the recording: not replaced while this
        self._pool.held_by_recording += 1                   # object lives (ZeroPool.take raises instead)
        self.graph_a = torch.cuda.CUDAGraph()
        self.graph_s, self.graph_c, self.graph_r, self.graph_w, self.graph_d, self.side = [], None, None, None, None, None
        if self.split:
            with torch.cuda.graph(self.graph_a, stream=self.stream):
                self.loss, cuts = self._piece_f()
            self.side = list(cuts.streams)
            for s, st in enumerate(self.side):              # a pool of its own each: the three replay at the same time
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    self._piece_stage(cuts, s)
                self.graph_s.append(g)
            if cuts.refine is not None and cuts.refine[1].grad is not None:
                self.graph_r = torch.cuda.CUDAGraph()       # on the caller's stream, while the stage chains run on theirs
                with torch.cuda.graph(self.graph_r, stream=self.stream):
                    self._piece_r(cuts)
            # C (pyramid + trunk backward: a chain of small launches) and W (the stages' weight gradients: a few chip-filling
            # launches) replay side by side.  C is recorded FIRST, while W's operands are still referenced by the queued launches:
            # nothing C allocates from F's pool can alias an activation W reads.  D (bucket gather, Adam) closes the step.
            self.graph_c = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_c, stream=self.stream, pool=self.graph_a.pool()):
                self._piece_c(cuts)
            from . import train_ops
            if train_ops.has_held(self.device):
                self.graph_w = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_w, stream=self.side[0]):
                    self._piece_w()
            self.graph_d = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_d, stream=self.stream, pool=self.graph_a.pool()):
                self.bucket.gather()
                if self.world <= 1:
                    self.opt.step(hyper=self.hyper)
            del cuts
        else:
            with torch.cuda.graph(self.graph_a, stream=self.stream):
                self.loss = self._forward_backward()
                if self.world <= 1:
                    self.opt.step(hyper=self.hyper)
        if self.world > 1:
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, stream=self.stream, pool=self.graph_a.pool()):
                self.opt.step(hyper=self.hyper)
        torch.cuda.synchronize(dev)

    def __del__(self):
        pool = self.__dict__.get("_pool")
        if pool is not None:
            pool.held_by_recording -= 1

    # ---- one step --------------------------------------------------------------------------------------------------------
    def __call__(self, imgs, extrinsics, intrinsics, depth_range, gt):
        """Same arguments as one iteration of train.py's loop body; tensors may live on the host (the loader's) or on the GPU.
        Cameras and range are needed on the HOST (their arithmetic is LAPACK's, scale.py / base.py): hand them over as CPU
        tensors, or registered with hostmirror.put, to avoid a device->host hop.  -> the loss (a fixed device tensor,
        overwritten by the next step)."""
        extr_h, intr_h, dr_h = (hostmirror.get(t) for t in (extrinsics, intrinsics, depth_range))
        if imgs.shape != self.imgs.shape or any(gt[k].shape != v.shape for k, v in self.gt.items()):
            raise RuntimeError("GraphedTrainStep: tensor shapes differ from the recorded step's")
        if imgs is not self.imgs:
            self.imgs.copy_(imgs, non_blocking=True)
        for k, v in self.gt.items():
            if gt[k] is not v:
                v.copy_(gt[k], non_blocking=True)
        if depth_range is not self.dr:
            self.dr.copy_(depth_range if depth_range.is_cuda else dr_h, non_blocking=True)
        self._upload(extr_h, intr_h, dr_h)
        self.graph_a.replay()
        if self.graph_c is not None:
            cur = torch.cuda.current_stream(self.device)
            for st, g in zip(self.side, self.graph_s):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    g.replay()
            if self.graph_r is not None:
                self.graph_r.replay()
            for st in self.side:
                cur.wait_stream(st)
            w_st = self.side[0]
            if self.graph_w is not None:
                w_st.wait_stream(cur)                       # W starts when EVERY stage chain is done (it reads all their operands)
                with torch.cuda.stream(w_st):
                    self.graph_w.replay()
            # (C on a high-priority stream beside W: the replay takes 13.8 ms instead of 6.9 -- measured, dropped)
            self.graph_c.replay()
            cur.wait_stream(w_st)
            self.graph_d.replay()
        if self.world > 1:
            self.bucket.allreduce_gradients()
            self.graph_b.replay()
        self.opt.steps += 1
        _graph.increment_version(self._versioned)           # written through raw pointers inside the graph
        return self.loss
