"""tl_forward (the C-side forward executor, csrc/tl_exec.hip) against the Python-driven engine (model/engine.py): the same launches with
the same arguments, so every output must be BIT-identical -- and, through the engine's own tests, equal to the oracle.  The reference
call this replaces: `model(batch, return_loss=False)` in the tile loop (tree_learn/util/pipeline.py:86)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(dtype, channels=32, num_blocks=7, spatial_shape=(500, 500, 1000), voxel=0.1, seed=7):
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import random_state_dict
    m = TreeLearn(channels=channels, num_blocks=num_blocks, use_feats=False, use_coords=False, spatial_shape=list(spatial_shape) if spatial_shape else None,
                  voxel_size=voxel, compute_dtype=dtype)
    m.load_state_dict(random_state_dict(seed, channels=channels, num_blocks=num_blocks), strict=True)
    return m.cuda().eval()


def _batch(tiles):
    from treelearn_amd.synth import make_batch
    b = make_batch(tiles)
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}


def _both(model, gb):
    """(executor outputs, Python-engine outputs, executor) of the same forward."""
    with torch.no_grad():
        ex = model._executor(model.active_dtype(False))
        assert ex is not None, "this configuration must be served by tl_forward"
        out_c = model(gb, return_loss=False)
        os.environ["TL_EXEC"] = "0"
        try:
            assert model._executor(model.active_dtype(False)) is None
            out_p = model(gb, return_loss=False)
        finally:
            del os.environ["TL_EXEC"]
    torch.cuda.synchronize()
    return out_c, out_p, ex


def _assert_equal(out_c, out_p):
    for k in ("backbone_feats", "semantic_prediction_logits", "offset_predictions"):
        assert out_c[k].shape == out_p[k].shape and out_c[k].dtype == out_p[k].dtype, k
        a, b = out_c[k], out_p[k]                       # (float16 on random-init weights overflows in places: inf / nan must agree too)
        same = (a == b) | (torch.isnan(a) & torch.isnan(b))
        assert bool(same.all()), f"{k}: executor and Python-driven engine differ in {int((~same).sum())} elements"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_exec_equals_engine_on_a_blocked_tile(dtype):
    """A 16 m tile (0.29 M voxels: level 1 in the block-local order for the 16-bit types, canonical in fp32)."""
    from treelearn_amd.synth import make_tile
    gb = _batch([make_tile(extent=16.0, voxel=0.1, n_trees=10, fill=0.1, seed=3)])
    m = _model(dtype)
    out_c, out_p, ex = _both(m, gb)
    _assert_equal(out_c, out_p)
    assert ex.last["blocked"] == (dtype != torch.float32)
    assert ex.last["launches"] == 72 if ex.last["blocked"] else ex.last["launches"] == 71
    assert dtype == torch.float16 or all(bool(torch.isfinite(out_c[k]).all()) for k in out_c)


def test_exec_equals_engine_small_tile_and_batch_of_two():
    """Below BLK_MIN_ROWS the 16-bit forward stays canonical; a batch of two tiles; spatial_shape = None (the tile's own extent)."""
    from treelearn_amd.synth import make_tile
    tiles = [make_tile(extent=6.0, voxel=0.1, n_trees=2, fill=0.1, seed=s) for s in (1, 2)]
    for shape in ((500, 500, 1000), None):
        m = _model(torch.bfloat16, num_blocks=5, spatial_shape=shape)
        for gb in (_batch(tiles[:1]), _batch(tiles)):
            out_c, out_p, ex = _both(m, gb)
            _assert_equal(out_c, out_p)
            assert ex.last["blocked"] == (ex.last["level_n"][0] >= 16384)


def test_exec_other_widths_and_depths():
    from treelearn_amd.synth import make_tile
    gb = _batch([make_tile(extent=10.0, voxel=0.1, n_trees=4, fill=0.1, seed=5)])
    for ch, nb in ((16, 3), (64, 2), (32, 4)):
        m = _model(torch.bfloat16, channels=ch, num_blocks=nb)
        out_c, out_p, _ = _both(m, gb)
        _assert_equal(out_c, out_p)


def test_exec_without_backbone_and_repeated_calls_reuse_the_arena():
    from treelearn_amd.synth import make_tile
    gb = _batch([make_tile(extent=12.0, voxel=0.1, n_trees=6, fill=0.1, seed=9)])
    m = _model(torch.bfloat16)
    out_c, out_p, ex = _both(m, gb)
    _assert_equal(out_c, out_p)
    arena = next(iter(ex._ctx.values()))[1]
    ptr0, size0 = arena.data_ptr(), arena.numel()
    assert ex.last["arena_bytes"] <= size0
    m.return_backbone_feats = False
    with torch.no_grad():
        o2 = m(gb, return_loss=False)
    assert o2["backbone_feats"] is None and torch.equal(o2["offset_predictions"], out_c["offset_predictions"])
    arena = next(iter(ex._ctx.values()))[1]
    assert (arena.data_ptr(), arena.numel()) == (ptr0, size0)
    # a larger tile grows the arena, results stay right
    gb2 = _batch([make_tile(extent=20.0, voxel=0.1, n_trees=14, fill=0.1, seed=11)])
    m.return_backbone_feats = True
    out_c2, out_p2, _ = _both(m, gb2)
    _assert_equal(out_c2, out_p2)
    out_c3, _, _ = _both(m, gb)                      # ... and the small tile still gives what it gave
    _assert_equal(out_c3, out_c)


def test_exec_on_side_streams_in_flight():
    """Four forwards in flight on four streams (what the tile loop and bench.py do): every stream has its own context and arena."""
    from treelearn_amd.synth import make_tile
    gbs = [_batch([make_tile(extent=12.0, voxel=0.1, n_trees=5, fill=0.1, seed=20 + i)]) for i in range(4)]
    m = _model(torch.bfloat16)
    with torch.no_grad():
        ref = [m(g, return_loss=False) for g in gbs]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream() for _ in range(4)]
        outs = [None] * 8
        for i in range(8):
            with torch.cuda.stream(streams[i % 4]):
                outs[i] = m(gbs[i % 4], return_loss=False)
        torch.cuda.synchronize()
    for i in range(8):
        _assert_equal(outs[i], ref[i % 4])
    assert len(m._plan._exec._ctx) == 5


def test_exec_raises_reach_zero_like_the_engine():
    from treelearn_amd.synth import make_tile
    gb = _batch([make_tile(extent=3.0, voxel=0.1, n_trees=1, fill=0.1, seed=0)])
    m = _model(torch.bfloat16, spatial_shape=None)
    with pytest.raises(ValueError, match="reach zero!!!"), torch.no_grad():
        m(gb, return_loss=False)
    m2 = _model(torch.bfloat16, spatial_shape=(16, 16, 16))
    with pytest.raises(ValueError, match="exceeds spatial_shape"), torch.no_grad():
        m2(gb, return_loss=False)


def test_exec_profile_records_every_launch():
    from treelearn_amd.synth import make_tile
    gb = _batch([make_tile(extent=16.0, voxel=0.1, n_trees=10, fill=0.1, seed=3)])
    m = _model(torch.bfloat16)
    with torch.no_grad():
        m(gb, return_loss=False)
        ex = m._plan._exec
        ex.profile(True)
        m(gb, return_loss=False)
        recs = ex.profile_read()
        ex.profile(False)
    assert len(recs) == ex.last["launches"] == 72
    assert all(r["ms"] > 0 for r in recs)
    assert sum(r["kind"] == "subm" for r in recs) == 52 + 1 and sum(r["kind"] == "input" for r in recs) == 1     # 53 SubM convs: the input conv + 52, the level-1 64 -> 32 one as two launches
    assert sum(r["kind"] == "down" for r in recs) == 6 and sum(r["kind"] == "inverse" for r in recs) == 6 and sum(r["kind"] == "1x1" for r in recs) == 6
    # the same launches the Python-driven engine reports to bench.py
    from treelearn_amd import ops
    ops.PROFILE = []
    try:
        with torch.no_grad():
            m(gb, return_loss=False)
        torch.cuda.synchronize()
        py = [meta for _, _, meta in ops.PROFILE]
    finally:
        ops.PROFILE = None
    assert [(r["K"], r["Cin"], r["Cout"], r["n_out"], r["n_in"], r["residual"], r["split"]) for r in recs] == \
           [(p["K"], p["Cin"], p["Cout"], p["n_out"], p["n_in"], p["residual"], p["split"]) for p in py]


def test_exec_config2_full_tile(tile2_batch):
    """The headline tile (1.89 M points): bit-identical to the Python-driven engine, which tests/test_gpu_configs.py holds against the oracle."""
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in tile2_batch.items()}
    m = _model(torch.bfloat16)
    out_c, out_p, ex = _both(m, gb)
    _assert_equal(out_c, out_p)
    assert ex.last["blocked"] and ex.last["level_n"][0] > 1_800_000


def test_model_with_a_live_executor_can_be_copied_and_pickled():
    """copy.deepcopy / pickle / torch.save of a model that has run (eval plan, C-side executor handles, arenas): the copy carries the
    parameters only and rebuilds its own plan -- same outputs, and neither object's destruction touches the other's handles."""
    import copy, gc, io, pickle
    from treelearn_amd.synth import make_tile
    gb = _batch([make_tile(extent=12.0, voxel=0.1, n_trees=6, fill=0.1, seed=2)])
    m = _model(torch.bfloat16)
    with torch.no_grad():
        ref = m(gb, return_loss=False)
    assert m._plan is not None and m._plan._exec is not None and m._plan._exec._ctx
    c = copy.deepcopy(m)
    assert c._plan is None and c.training == m.training
    r = pickle.loads(pickle.dumps(m))
    buf = io.BytesIO(); torch.save(m, buf); buf.seek(0)
    t = torch.load(buf, weights_only=False)
    with torch.no_grad():
        for other in (c, r.cuda(), t.cuda()):
            out = other(gb, return_loss=False)
            _assert_equal(out, ref)
    assert c._plan is not m._plan and c._plan._exec is not m._plan._exec
    del m, r, t
    gc.collect(); torch.cuda.synchronize()
    with torch.no_grad():
        _assert_equal(c(gb, return_loss=False), ref)             # the copy lives on after the original (and its executor) are gone
    with torch.enable_grad():                                    # a training model with a pack plan copies too
        c.train()
        loss, _ = c(gb, return_loss=True)
        loss.backward()
        d = copy.deepcopy(c)
        assert getattr(d, "_pack_plan", None) is None
        loss2, _ = d(gb, return_loss=True)
    assert float(loss2.detach()) == float(loss.detach())


def test_exec_called_from_several_python_threads():
    """Host threads: ctypes releases the GIL inside tl_forward, so two Python threads may be inside it at once.  Forwards on the SAME stream share
    one context (read-back buffer, arena) and are serialised by its lock; forwards on different streams have their own.  Every result equals
    the single-threaded one."""
    import threading
    from treelearn_amd.synth import make_tile
    tiles = [_batch([make_tile(extent=12.0, voxel=0.1, n_trees=6, fill=0.1, seed=s)]) for s in (11, 12, 13)]
    m = _model(torch.bfloat16)
    with torch.no_grad():
        refs = [m(t, return_loss=False) for t in tiles]
    torch.cuda.synchronize()
    errors = []

    def worker(k, stream):
        try:
            with torch.no_grad():
                for it in range(6):
                    i = (k + it) % len(tiles)
                    with torch.cuda.stream(stream):
                        out = m(tiles[i], return_loss=False)
                    stream.synchronize()
                    for key in ("semantic_prediction_logits", "offset_predictions", "backbone_feats"):
                        if not torch.equal(out[key], refs[i][key]):
                            errors.append((k, it, key))
        except Exception as e:                                   # noqa: BLE001
            errors.append((k, repr(e)))

    shared = torch.cuda.Stream()
    for streams in ([shared, shared, shared], [torch.cuda.Stream() for _ in range(3)]):
        for st in set(streams):
            st.wait_stream(torch.cuda.current_stream())
        th = [threading.Thread(target=worker, args=(k, streams[k])) for k in range(3)]
        for t in th: t.start()
        for t in th: t.join()
        assert not errors, errors[:5]


def test_eval_forward_under_inference_mode(monkeypatch):
    """`torch.inference_mode()` instead of `torch.no_grad()` around the evaluation call (tensors without version counters): both launch paths,
    same outputs; and a plan built inside inference mode keeps serving calls outside it."""
    from treelearn_amd.synth import make_tile
    gb = _batch([make_tile(extent=12.0, voxel=0.1, n_trees=6, fill=0.1, seed=5)])
    m = _model(torch.bfloat16)
    with torch.inference_mode():
        a = m(gb, return_loss=False)                      # the plan (packed weights, folded BatchNorms) is built in here
    with torch.no_grad():
        b = m(gb, return_loss=False)
    _assert_equal(a, b)
    monkeypatch.setenv("TL_EXEC", "0")
    m2 = _model(torch.bfloat16)
    with torch.inference_mode():
        c = m2(gb, return_loss=False)
    _assert_equal(a, c)
    m3 = _model(torch.float32)
    with torch.inference_mode():
        d = m3(gb, return_loss=False)
    assert torch.isfinite(d["semantic_prediction_logits"]).all()


def test_models_come_and_go_without_leaking_device_memory():
    """Ten models in a row, each built, run (plan, executor, arenas) and dropped: the caching allocator's live bytes return to where they were
    (the plan <-> executor link is a weak reference, `tl_exec_destroy` runs from the executor's finaliser)."""
    import gc
    from treelearn_amd.synth import make_tile
    gb = _batch([make_tile(extent=12.0, voxel=0.1, n_trees=6, fill=0.1, seed=6)])
    gc.collect(); torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    seen = []
    for i in range(10):
        m = _model(torch.bfloat16, seed=20 + i)
        with torch.no_grad():
            m(gb, return_loss=False)
        torch.cuda.synchronize()
        del m
        gc.collect()
        seen.append(torch.cuda.memory_allocated() - base)
    assert max(seen[2:]) <= seen[1] + (1 << 20), seen       # no growth from model to model
    assert seen[-1] < (8 << 20), seen                        # and (almost) nothing left behind


# ------------------------------------------------------------------------------------------------ round 6: features, verdicts, bounded memory
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, "bf16x3"])
@pytest.mark.parametrize("flags", [(True, False), (False, True), (True, True)], ids=["use_feats", "use_coords", "both"])
def test_exec_serves_voxel_mean_features(dtype, flags):
    """The reference constructor's own defaults are use_feats=True (tree_learn/model/tree_learn.py:18; the yaml sets False): voxel-mean
    features (:149-155) built INSIDE tl_forward (tl_voxel_feats) -- bit-identical to the Python-driven engine (tl_voxel_mean_feats + torch
    column edits + the input conv on the canonical table + a row permutation into the block-local order).  Two clouds in the batch, several
    points per voxel (the mean is over the first three), exact zeros among the features."""
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import make_tile, random_state_dict
    use_feats, use_coords = flags
    tiles = [make_tile(extent=16.0, voxel=0.05, n_trees=8, fill=0.2, seed=5), make_tile(extent=9.0, voxel=0.05, n_trees=3, fill=0.2, seed=6)]
    gb = _batch(tiles)
    gb["input_feats"][::17] = 0.0
    m = TreeLearn(use_feats=use_feats, use_coords=use_coords, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=dtype)
    m.load_state_dict(random_state_dict(7, channels=32, num_blocks=7), strict=True)
    m = m.cuda().eval()
    out_c, out_p, ex = _both(m, gb)
    assert ex.needs_feats and ex.last["level_n"][0] < 0.6 * gb["coords"].shape[0]            # several points per voxel
    assert ex.last["blocked"] == (dtype != torch.float32)      # (16-bit and the parity-fast mode run level 1 in the block-local order)
    _assert_equal(out_c, out_p)
    # ... and the features matter: the all-ones net on the same weights gives something else
    ones = _model(dtype if dtype != "bf16x3" else torch.float32)
    with torch.no_grad():
        o1 = ones(gb, return_loss=False)
    assert not torch.equal(o1["semantic_prediction_logits"], out_c["semantic_prediction_logits"])


def test_exec_default_constructor_takes_the_one_call_path():
    """`TreeLearn()` exactly as the reference's constructor defaults build it (use_feats=True, use_coords=False, spatial_shape=None) is served
    by tl_forward, and its fp32 forward is within 1e-3 of the CPU oracle's."""
    from oracle import model as om
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import make_batch, make_tile, random_state_dict
    batch = make_batch([make_tile(extent=8.0, voxel=0.1, n_trees=3, fill=0.1, seed=9)])
    sd = random_state_dict(4, channels=32, num_blocks=7)
    m = TreeLearn()
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        out = m(batch, return_loss=False)
    assert m._plan._exec is not None and m._plan._exec.needs_feats and m._plan._exec.last["launches"] > 60
    ref = om.forward(sd, batch["coords"].numpy(), batch["input_feats"].numpy(), batch["batch_ids"].numpy(), 1, voxel_size=0.1, num_blocks=7,
                     use_feats=True, use_coords=False)
    for k in ("semantic_prediction_logits", "offset_predictions"):
        a, b = out[k].cpu().numpy().astype(np.float64), ref[k].numpy().astype(np.float64)
        assert np.abs(a - b).max() / np.abs(b).max() < 1e-3, k


def test_exec_contexts_are_bounded_and_forty_streams_do_not_grow_memory():
    """One context + arena per (device, stream) -- but at most MAX_CONTEXTS of them (least recently used first out): a caller that runs every
    request on a fresh stream holds eight arenas, not one per stream handle it ever used."""
    from treelearn_amd.model import executor as E
    from treelearn_amd.synth import make_tile
    gb = _batch([make_tile(extent=12.0, voxel=0.1, n_trees=6, fill=0.1, seed=2)])
    m = _model(torch.bfloat16)
    with torch.no_grad():
        ref = m(gb, return_loss=False)
    ex = m._plan._exec
    streams = [torch.cuda.Stream() for _ in range(40)]
    assert len({s.cuda_stream for s in streams}) > E.MAX_CONTEXTS + 4        # torch hands out a pool of distinct handles
    peak_ctx, mem = 0, []
    with torch.no_grad():
        for i, s in enumerate(streams):
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                out = m(gb, return_loss=False)
            s.synchronize()
            _assert_equal(out, ref)
            peak_ctx = max(peak_ctx, len(ex._ctx))
            mem.append(torch.cuda.memory_allocated())
    assert peak_ctx <= E.MAX_CONTEXTS and len(ex._ctx) == E.MAX_CONTEXTS
    arena = next(iter(ex._ctx.values()))[1].numel()
    assert max(mem[E.MAX_CONTEXTS:]) - mem[E.MAX_CONTEXTS - 1] < arena, "memory kept growing after the context cap was reached"
    ex.release_memory()
    assert all(c[1] is None for c in ex._ctx.values())
    with torch.no_grad():
        _assert_equal(m(gb, return_loss=False), ref)                        # arenas come back on demand


def test_exec_check_reports_the_unit_builder_flag_for_the_right_forward():
    """tl_exec_check: TL_OK after clean forwards (and it may be called repeatedly / with nothing in flight); with the flag word of the context
    forced to 1 -- the state a builder that skipped units leaves behind -- Executor.check() raises, once; get_pointwise_preds asks after its
    last tile; TL_EXEC_CHECK=1 asks after every forward."""
    import ctypes
    from treelearn_amd import _hip
    from treelearn_amd.synth import make_tile
    from treelearn_amd.util.pipeline import get_pointwise_preds
    gb = _batch([make_tile(extent=12.0, voxel=0.1, n_trees=6, fill=0.1, seed=2)])
    m = _model(torch.bfloat16)
    with torch.no_grad():
        m(gb, return_loss=False)
    ex = m._plan._exec
    assert ex.last["blocked"]
    ex.check(); ex.check()
    L = _hip.lib()
    fresh = L.tl_exec_create()
    assert L.tl_exec_check(fresh) == _hip.TL_OK
    L.tl_exec_destroy(fresh)
    assert L.tl_exec_check(None) == _hip.TL_ERR_ARG
    os.environ["TL_EXEC_CHECK"] = "1"
    try:
        with torch.no_grad():
            m(gb, return_loss=False)
    finally:
        del os.environ["TL_EXEC_CHECK"]
    res = get_pointwise_preds(m, [gb, gb, gb], dict(voxel_size=0.1), keep_on_device=True)
    assert len(res[0]) > 0


def test_tl_forward_rejects_inconsistent_descriptors():
    """tl_forward is a public entry point: a descriptor whose recursion flags, widths or pointers do not fit together is TL_ERR_ARG before
    anything is enqueued (the recursion of the U-Net trusts `deeper`; a wrong flag would read past the levels)."""
    import copy, ctypes
    from treelearn_amd import _hip
    from treelearn_amd.synth import make_tile
    gb = _batch([make_tile(extent=10.0, voxel=0.1, n_trees=4, fill=0.1, seed=2)])
    m = _model(torch.bfloat16)
    with torch.no_grad():
        ref = m(gb, return_loss=False)
    ex = m._plan._exec
    L = _hip.lib()
    N = gb["coords"].shape[0]
    arena = torch.empty(int(ex.last["arena_bytes"] * 1.2) + (1 << 20), dtype=torch.uint8, device="cuda")
    logits = torch.empty((N, 2), device="cuda"); offsets = torch.empty((N, 3), device="cuda")
    ctx = L.tl_exec_create()

    def call(desc):
        a = _hip.ForwardArgs()
        a.xyz = gb["coords"].data_ptr(); a.batch_ids = gb["batch_ids"].data_ptr(); a.N = N; a.B = 1
        a.logits = logits.data_ptr(); a.offsets = offsets.data_ptr(); a.arena = arena.data_ptr(); a.arena_bytes = arena.numel()
        return L.tl_forward(ctx, ctypes.byref(desc), ctypes.byref(a), torch.cuda.current_stream().cuda_stream)

    def mutated(f):
        d = _hip.NetDesc()
        ctypes.memmove(ctypes.byref(d), ctypes.byref(ex.desc), ctypes.sizeof(d))
        f(d)
        return d

    assert call(mutated(lambda d: None)) == _hip.TL_OK
    torch.cuda.synchronize()
    assert torch.equal(logits, ref["semantic_prediction_logits"])
    bad = [lambda d: setattr(d.u[6], "deeper", 1),                  # the last level claims a deeper one
           lambda d: setattr(d.u[3], "deeper", 0),                  # a truncated net
           lambda d: setattr(d, "num_levels", 5),                   # fewer levels than the flags say
           lambda d: setattr(d.u[2], "C", 64),                      # a width that does not match its weights
           lambda d: setattr(d.u[1].blocks[0].w2, "w", None),       # a missing weight
           lambda d: setattr(d.u[0].tail[0].w1x1, "Cin", 32),       # the 1x1 i_branch of a 2C -> C block with the wrong width
           lambda d: setattr(d.w_in, "Cout", 64),
           lambda d: setattr(d, "in_channels", 7),
           lambda d: setattr(d, "use_feats", 1),                    # features wanted, none passed
           lambda d: setattr(d.out_bn, "scale", None)]
    for i, f in enumerate(bad):
        assert call(mutated(f)) == _hip.TL_ERR_ARG, i
    L.tl_exec_destroy(ctx)
