"""Stage orchestrator (reference: net/core.py:4-78): same constructor slots, forward signature, output
dicts and state_dict keys; the stage loop runs on one HIP stream with the hot operators as fused kernels."""
import torch

from mdfnet_hip import controlplane, hostmirror, layers, ops


import contextlib
import os as _os

_STAGE_STREAMS = bool(int(_os.environ.get("MDF_TRAIN_STAGE_STREAMS", "1")))      # dev A/B: 0 = the whole training step on one stream
_SIDE = {}


def _stage_streams(device, n):
    key = (device.index if device.index is not None else torch.cuda.current_device(), n)
    if key not in _SIDE:
        _SIDE[key] = [torch.cuda.Stream(device) for _ in range(n)]
    return _SIDE[key]


@contextlib.contextmanager
def _on_stage_stream(side, stage, main, feats):
    """Issue the body on side[stage]: forked from `main` before, joined into it after (inside a hipGraph recording both edges are
    recorded).  The feature maps were allocated on `main` and are read on the stage's stream."""
    if not side:
        yield
        return
    st = side[stage]
    st.wait_stream(main)
    for f in feats:
        base = getattr(f, "_mdf_parent", None)
        (base[0] if base is not None else f).record_stream(st)
    with torch.cuda.stream(st):
        yield
    main.wait_stream(st)


def _cut_features(feats, cuts):
    """One stage's per-view feature maps -> detached copies that require grad (leaves of the stage's backward chain), noted in
    `cuts`.  The views of one view-major tensor (FPN_4Scales.forward_views) stay slices of ONE cut tensor, so the aggregation slot
    still takes it -- and returns its gradient -- whole."""
    parents = [getattr(f, "_mdf_parent", None) for f in feats]
    pairs = []
    if all(p is not None and p[0] is parents[0][0] and p[1] == i and p[2] == len(feats) for i, p in enumerate(parents)):
        y = parents[0][0]
        yc = y.detach().requires_grad_(True)
        b = y.shape[0] // len(feats)
        out = [yc[g * b:(g + 1) * b] for g in range(len(feats))]
        for g, t in enumerate(out):
            t._mdf_parent = (yc, g, len(feats))
        pairs.append((y, yc))
    else:
        out = [f.detach().requires_grad_(True) for f in feats]
        pairs.extend(zip(feats, out))
    cuts.feat.append(pairs)
    return out


class CoreNet(torch.nn.Module):
    def __init__(self, Backbone, Depth_hypos, scale, Homoaggre, Regular, Regress, Refine):
        """Backbone: img -> 3 feature maps; Depth_hypos/Homoaggre/Regular: ModuleLists (one per stage);
        scale: callable(K, E, stage); Regress: [depth_regression, confidence_regress]; Refine: module."""
        super().__init__()
        self.Backbone, self.Depth_hypos, self.scale = Backbone, Depth_hypos, scale
        self.Homoaggre, self.Regular, self.Refine = Homoaggre, Regular, Refine
        self.Depth_regress, self.Confidence_regress = Regress
        print("{} parameters: {}".format(self._get_name(), sum(p.data.nelement() for p in self.parameters())))

    def forward(self, origin_imgs, extrinsics, intrinsics, depth_range, feature_cache=None, view_keys=None):
        """Same call as the reference (net/core.py:30).  Optional, for whole-scan evaluation: a dict
        `feature_cache` and per-view hashable `view_keys` [V] (batch 1) -- feature pyramids are then computed once per
        image and reused while it serves as a source view of other items; results are identical."""
        with layers.model_mode(self.training):
            plan = None
            if origin_imgs.is_cuda:
                # one device->host hop for the control-plane tensors (cameras, range) when the caller has not registered host
                # mirrors; then every slot's few dozen host-computed floats go up in ONE packed copy (mdfnet_hip/controlplane.py)
                hostmirror.ensure((extrinsics, intrinsics, depth_range))
                plan = controlplane.prepare(self, intrinsics, extrinsics, depth_range)
            with controlplane.active(plan):
                return self._forward(origin_imgs, extrinsics, intrinsics, depth_range, feature_cache, view_keys)

    def _pyramids(self, imgs, feature_cache, view_keys):
        nb, nv = imgs.shape[:2]
        if feature_cache is not None and view_keys is not None and nb == 1 and not self.training:
            if hasattr(feature_cache, "pin"):
                feature_cache.pin(view_keys)                             # a bounded cache must keep this item's views
            missing = [v for v in range(nv) if view_keys[v] not in feature_cache]
            if missing:
                f = self.Backbone(imgs[0, missing])                      # only the images not seen yet, batched
                ev = torch.cuda.Event()
                ev.record()                                              # on the stream that produced them
                for j, v in enumerate(missing):
                    feature_cache[view_keys[v]] = (tuple(lv[j:j + 1] for lv in f), ev)
            cur = torch.cuda.current_stream(imgs.device)
            out = []
            for v in range(nv):
                pyr, ev = feature_cache[view_keys[v]]
                cur.wait_event(ev)                                       # another item in flight may have produced it
                for t in pyr:
                    t.record_stream(cur)
                out.append(pyr)
            return out
        if self.training and imgs.is_cuda and hasattr(self.Backbone, "forward_views") and layers.hip_train(self.Backbone, imgs):
            return self.Backbone.forward_views(imgs)          # all views in one pass, BatchNorm statistics per view
        if getattr(self.Backbone, "batch_views", False) and not self.training:
            # eval BatchNorm is per-sample: one batched pass over the B*V images == V separate calls (core.py:42)
            f = self.Backbone(imgs.reshape(nb * nv, *imgs.shape[2:]))
            return [tuple(lv.reshape(nb, nv, *lv.shape[1:])[:, v] for lv in f) for v in range(nv)]
        return [self.Backbone(v) for v in torch.unbind(imgs, 1)]

    def _forward(self, origin_imgs, extrinsics, intrinsics, depth_range, feature_cache=None, view_keys=None):
        """imgs [B,V,3,H,W] (view 0 = reference), E [B,V,4,4], K [B,V,3,3], range [B,2]
        -> train: {"depth": [1/8, 1/4, 1/2, 1/1]};  eval: {"depth": [B,H,W], "confidence": [B,H,W]}."""
        if not self.training and not origin_imgs.is_cuda:
            raise RuntimeError("CoreNet inference runs on hand-written MI355X kernels only: move the model and inputs to a "
                               "GPU (there is no CPU fallback; model.train() on a GPU runs the hand-written training kernels)")
        if origin_imgs.is_cuda:
            # one device->host hop for the control-plane tensors (cameras, range); the slots then find host
            # mirrors and never synchronise again
            hostmirror.ensure((extrinsics, intrinsics, depth_range))
        if self.training and origin_imgs.is_cuda and layers.hip_train(self, origin_imgs):
            # this step's weight packs (the optimizer has just changed every weight): one batched launch
            from mdfnet_hip import train_ops
            with torch.no_grad():
                train_ops.prepack(self)
        pyramids = self._pyramids(origin_imgs.float(), feature_cache, view_keys)
        depth = hypos = prob = None
        depths = []
        # Training on the HIP kernels: every stage's aggregation + regulariser is issued on a stream of its own.  The forward pass
        # gains nothing (stage s+1 needs stage s's depth: the streams are chained by events), but autograd runs a node's backward
        # on the stream of its forward, and the three stages' backward chains do not depend on each other -- the hypotheses are
        # built under no_grad (depthhypos.py:40,188) and the refinement net detaches its input (refine.py:29) -- so they run side
        # by side until they meet at the feature pyramid: the small-volume launches of one stage fill the CUs another leaves idle.
        side = cuts = None
        if self.training and origin_imgs.is_cuda and torch.is_grad_enabled() and layers.hip_train(self, origin_imgs):
            cuts = layers.active_cuts()      # a recorded step that takes the backward pass in pieces (layers.StageCuts)
            # (a hipGraph with parallel branches is replayed node by node by the host -- 6.5 ms per cfg3 step --, so a recording
            #  keeps one stream unless it cuts the stage chains out into graphs of their own)
            if cuts is not None or (_STAGE_STREAMS and not torch.cuda.is_current_stream_capturing()):
                side = _stage_streams(origin_imgs.device, len(self.Regular))
                main = torch.cuda.current_stream(origin_imgs.device)
                if cuts is not None:
                    cuts.streams = side
        for stage, (make_hypos, aggregate, regular) in enumerate(zip(self.Depth_hypos, self.Homoaggre, self.Regular)):
            feats = [p[stage] for p in pyramids]
            if cuts is not None:
                feats = _cut_features(feats, cuts)
            ref_proj, src_projs = self.scale(intrinsics, extrinsics, stage)
            with _on_stage_stream(side, stage, main if side else None, feats):
                hypos = make_hypos(depth, depth_range, prob, hypos, upsample=True)
                cost = aggregate(feats, ref_proj, src_projs, hypos)
                if ((not self.training or cost.is_cuda) and getattr(regular, "fused_regress", False)
                        and getattr(self.Depth_regress, "mdf_builtin", False)):
                    # both slots are the built-in ones: the soft-argmin (core.py:64) rides in the regulariser's softmax kernel
                    # (same arithmetic as the standalone slot, asserted equal in tests/test_regular_gpu.py)
                    prob, depth = regular(cost, hypos)
                else:
                    prob = regular(cost)
                    depth = self.Depth_regress(prob, hypos)
            if side:
                # produced on the stage's stream; read by the next stage's stream and by the caller's (loss, refinement net)
                for t in (hypos, prob, depth):
                    t.record_stream(main)
                    if stage + 1 < len(side):
                        t.record_stream(side[stage + 1])
            if cuts is not None:
                cut = depth.detach().requires_grad_(True)
                cuts.depth.append((depth, cut))
                depth = cut
            depths.append(depth)
        depth = self.Refine(depth, depth_range)
        if cuts is not None and depth.requires_grad:
            cut = depth.detach().requires_grad_(True)       # the refinement net's backward (parameters only: refine.py:29 detaches
            cuts.refine = (depth, cut)                      # its input) is a piece of its own, replayed beside the stage chains
            depth = cut
        depths.append(depth)
        if self.training:
            return {"depth": depths}
        if getattr(self.Confidence_regress, "mdf_builtin", False) and prob.is_cuda:
            conf = ops.confidence_up2(prob.detach())          # regress.py:9-25 + the nearest x2 of core.py:76 in one launch
        else:
            conf = self.Confidence_regress(prob)
            conf = torch.nn.functional.interpolate(conf.unsqueeze(1), scale_factor=2, mode="nearest").squeeze(1)
        return {"depth": depth, "confidence": conf}
