"""CPU: every product module imports and the GPU-only helpers exist (guards against a broken edit that only the GPU
suite would notice)."""


def test_product_modules_import_and_expose_their_entry_points():
    from mdfnet_hip import ddp, dropin, hostmirror, layers, ops, shard, synth  # noqa: F401
    import rehearsal  # noqa: F401
    from rehearsal import stockops  # noqa: F401
    import net.core, net.loss  # noqa: F401,E401
    from net.unit import backbone, base, depthhypos, homoaggregate, refine, regress, regular, scale  # noqa: F401
    for name in ("conv2d_layer", "use_hip", "hip_eval", "cache_of", "model_mode"):
        assert hasattr(layers, name), name
    for name in ("conv2d_nhwc", "conv3d_ndhwc", "prob_head", "warp_aggregate_vec", "warp_aggregate_var", "homo_warp",
                 "hypos_fit", "hypos_from_fit", "depth_regress", "confidence", "gauss1_fit_row", "relative_projections"):
        assert hasattr(ops, name), name
