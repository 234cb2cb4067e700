"""CPU-side tests: C-ABI export check, host logic (tile loop, label bookkeeping, sharding), oracle self-consistency."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """Build (hipcc cross-compiles without a GPU), load, and check every function include/*.h declares."""
    from treelearn_amd import build as b
    lib = b.build(verbose=False)
    L = ctypes.CDLL(lib)
    hdr = open(os.path.join(REPO, "include", "treelearn_hip.h")).read()
    names = set(re.findall(r"\b(tl_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 20
    for n in sorted(names):
        assert hasattr(L, n), f"{n} declared in include/treelearn_hip.h but not exported"
    from treelearn_amd import _hip
    assert set(_hip.PROTOTYPES) == names, (set(_hip.PROTOTYPES) ^ names)
    assert _hip.lib().tl_version() >= 1
    assert _hip.lib().tl_error_string(-1) == b"invalid argument"


def test_every_entry_point_rejects_null_arguments():
    """The C ABI at its bluntest: every function include/treelearn_hip.h declares, called with null pointers, zero sizes and zeroed arrays.
    Compute entries must come back with a negative code before touching a pointer or launching anything (no GPU is needed to find that
    out); workspace-size queries may answer with a size.  Runs in a child process so that a fault names the function instead of ending pytest."""
    import subprocess
    import sys
    child = r"""
import ctypes, re, sys
sys.path.insert(0, %r)
from treelearn_amd import _hip
L = _hip.lib()
names = sorted(set(re.findall(r"\b(tl_[a-z0-9_]+)\s*\(", open(%r).read())))
for n in names:
    f = getattr(L, n)
    args = []
    for t in f.argtypes:
        if issubclass(t, ctypes.Array): args.append(t())
        elif t in (ctypes.c_float, ctypes.c_double): args.append(0.0)
        elif issubclass(t, ctypes._SimpleCData) and t not in (ctypes.c_void_p, ctypes.c_char_p): args.append(0)
        else: args.append(None)
    print("CALL", n, flush=True)
    r = f(*args)
    print("RET", n, r if isinstance(r, int) else ("null" if r is None else "obj"), flush=True)
print("DONE", len(names))
""" % (REPO, os.path.join(REPO, "include", "treelearn_hip.h"))
    p = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=600)
    lines = p.stdout.strip().splitlines()
    assert p.returncode == 0 and lines and lines[-1].startswith("DONE"), (p.returncode, lines[-2:], p.stderr[-500:])
    sizes = re.compile(r"_(ws_(bytes|words|doubles|floats)(_k)?|red_parts|bytes|slots|parts)$")
    seen = 0
    for ln in lines:
        if not ln.startswith("RET"):
            continue
        _, n, r = ln.split()
        seen += 1
        if n in ("tl_version", "tl_error_string", "tl_exec_create", "tl_exec_destroy") or sizes.search(n):
            continue                                                   # (tl_exec_create answers NULL without a GPU; a destroy of NULL is a no-op)
        assert r not in ("null", "obj") and int(r) < 0, f"{n}(nulls) returned {r}"
    assert seen >= 60


def test_product_path_fails_loudly_without_gpu():
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import make_batch, make_tile
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = TreeLearn(channels=8, num_blocks=2, use_feats=False).eval()
    with pytest.raises(Exception):
        with torch.no_grad():
            m(make_batch([make_tile(extent=3, voxel=0.5, n_trees=1, seed=0)]), return_loss=False)


def test_state_dict_matches_reference_manifest(golden_dir):
    import json
    from treelearn_amd.model import TreeLearn
    man = json.load(open(os.path.join(golden_dir, "g8_manifest.json")))
    sd = TreeLearn(use_feats=False).state_dict()
    assert list(sd.keys()) == [k for k, _, _ in man["keys"]]
    assert all(tuple(sd[k].shape) == tuple(s) for k, s, _ in man["keys"])
    assert sum(p.numel() for p in TreeLearn(use_feats=False).parameters()) == man["n_params"]


def test_loss_golden(golden_dir):
    from treelearn_amd.util.train import point_wise_loss_impl
    g = np.load(os.path.join(golden_dir, "g1_g2_loss.npz"))
    T = torch.from_numpy
    for c in "abcd":
        sem, off = point_wise_loss_impl(*[T(g[f"{c}_{k}"]) for k in ("logits", "offsets", "masks_sem", "masks_off", "semantic_labels", "offset_labels")])
        assert float(sem) == pytest.approx(float(g[f"{c}_pw_sem"]), rel=1e-6, abs=1e-7)
        assert float(off) == pytest.approx(float(g[f"{c}_pw_off"]), rel=1e-6, abs=1e-7)


def test_make_labels_consecutive_golden(golden_dir):
    from treelearn_amd.util.pipeline import make_labels_consecutive
    g = np.load(os.path.join(golden_dir, "g5_clustering.npz"))
    new, mapping = make_labels_consecutive(g["mlc_in"], 1)
    np.testing.assert_array_equal(new, g["mlc_out"])
    assert list(mapping.keys()) == g["mlc_map_keys"].tolist() and list(mapping.values()) == g["mlc_map_vals"].tolist()


def test_oracle_sparse_equals_dense():
    """rulebook (gather-mm) form == dense conv3d / conv_transpose3d form of every sparse op."""
    from oracle import sparse_ops as so, voxel as ov
    rng = np.random.default_rng(3)
    pts = rng.uniform(0, 2.3, size=(900, 3)).astype(np.float32)
    bids = (rng.uniform(size=900) < 0.4).astype(np.int64)
    _, vc, _, ss = ov.voxelize(pts, np.zeros((900, 1), np.float32), bids, 2, 0.2)
    shape = ss + np.array([0, 1, 0])                       # an odd dim to exercise the out-of-range drop
    x = torch.from_numpy(rng.normal(size=(len(vc), 5)).astype(np.float32))
    w3 = torch.from_numpy(rng.normal(size=(7, 3, 3, 3, 5)).astype(np.float32))
    a = so.conv_table(x, w3, ov.rulebook_subm(vc))
    b = so.subm_conv_dense(x, vc, w3, shape, 2)
    assert torch.allclose(a, b, atol=1e-4)
    cc, parent, child, oshape = ov.rulebook_down(vc, shape)
    w2 = torch.from_numpy(rng.normal(size=(6, 2, 2, 2, 5)).astype(np.float32))
    d1 = so.conv_table(x, w2, child)
    d2 = so.down_conv_dense(x, vc, w2, shape, 2, cc)
    assert torch.allclose(d1, d2, atol=1e-4)
    wi = torch.from_numpy(rng.normal(size=(5, 2, 2, 2, 6)).astype(np.float32))
    u1 = so.inverse_conv(d1, wi, parent, vc)
    u2 = so.inverse_conv_dense(d1, cc, wi, oshape, 2, vc, shape)
    assert torch.allclose(u1, u2, atol=1e-4)


def test_assign_tiles_lpt():
    from treelearn_amd.util.sharding import assign_tiles
    a = assign_tiles([10, 9, 8, 7, 6, 5, 4, 3], 2)
    assert sorted(a[0] + a[1]) == list(range(8))
    loads = [sum([10, 9, 8, 7, 6, 5, 4, 3][i] for i in r) for r in a]
    assert abs(loads[0] - loads[1]) <= 2
    assert assign_tiles([5, 5, 5], 4)[3] == []


_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["TL_REPO"])
import numpy as np, torch, torch.distributed as dist
from treelearn_amd.util.sharding import get_pointwise_preds_sharded
from treelearn_amd.util.pipeline import get_pointwise_preds
dist.init_process_group("gloo")
g = np.load(os.path.join(os.environ["TL_REPO"], "tests", "golden", "g9_tile_loop.npz"))
class Fake(torch.nn.Module):
    def forward(self, batch, return_loss):
        c = batch["coords"]
        if float(c[:, 0].mean()) > 900: raise RuntimeError("your out spatial shape reach zero!!! (fake)")
        return dict(offset_predictions=c * 0.5 + 1, semantic_prediction_logits=torch.stack([c[:, 0], -c[:, 1]], 1), backbone_feats=c.repeat(1, 11)[:, :32])
keys = ["coords", "input_feats", "batch_ids", "semantic_labels", "instance_labels", "masks_inner", "masks_off", "masks_sem", "offset_labels", "centers"]
def tiles():
    out = []
    for i in range(3):
        b = {k: torch.from_numpy(g[f"t{i}_{k}"]) for k in keys}; b["batch_size"] = 1; out.append(b)
    return out
# count the collectives the sharded loop issues (SURVEY 8e: one count gather + one payload gather)
calls = []
for name in ("all_gather_into_tensor", "all_gather", "broadcast", "all_reduce"):
    orig = getattr(dist, name)
    setattr(dist, name, (lambda o, n: (lambda *a, **k: (calls.append(n), o(*a, **k))[1]))(orig, name))
res = get_pointwise_preds_sharded(Fake(), tiles(), dict(voxel_size=0.2), device=torch.device("cpu"), return_backbone_feats=True)
assert calls == ["all_gather_into_tensor"] * 2, calls
assert get_pointwise_preds_sharded.last_record_width == 1 + 2 + 1 + 3 + 3 + 3 + 1 + 32 + 1          # with the backbone columns: 188 bytes
lean = get_pointwise_preds_sharded(Fake(), tiles(), dict(voxel_size=0.2), device=torch.device("cpu"))
assert get_pointwise_preds_sharded.last_record_width == 15 and lean[6].shape == (len(lean[0]), 0)    # default: the 60-byte record
for i in (0, 1, 2, 3, 4, 5, 7):
    np.testing.assert_array_equal(lean[i], res[i])
calls.clear()
for i, r in enumerate(res):
    np.testing.assert_allclose(r, g[f"out{i}"], rtol=1e-6, atol=1e-6)
    assert r.dtype == g[f"out{i}"].dtype, (i, r.dtype, g[f"out{i}"].dtype)
# lazy source: a rank materialises only its own tiles
from treelearn_amd.util.sharding import TileList, assign_tiles, segment_plot_sharded
made = []
tl = tiles()
src = TileList([t["coords"].shape[0] for t in tl], lambda i: (made.append(i), tl[i])[1])
res2 = get_pointwise_preds_sharded(Fake(), src, dict(voxel_size=0.2), device=torch.device("cpu"), return_backbone_feats=True)
assert sorted(made) == assign_tiles(src.n_points, dist.get_world_size())[dist.get_rank()], made
for a, b in zip(res, res2):
    np.testing.assert_array_equal(a, b)
# whole-plot order of operations (tile loop -> gather -> ensemble -> grouping on rank 0 -> id broadcast) with CPU stand-ins
def ens(coords, sem, seml, off, offl, instl, bb, infeat):
    return tuple(np.asarray(x) for x in (coords, sem, seml, off, offl, instl, bb, infeat))
def inst_fn(coords, off, sem, cfg, vert, tree_class, non_trees, not_assigned, start):
    assert dist.get_rank() == 0
    return (np.arange(len(coords)) % 5).astype(np.int64)
calls.clear()
c, ids = segment_plot_sharded(Fake(), tiles(), dict(voxel_size=0.2), {}, device=torch.device("cpu"), ensemble_fn=ens, instances_fn=inst_fn,
                              fill_fn=lambda c, p, na: p)
assert calls == ["all_gather_into_tensor", "all_gather_into_tensor", "broadcast"], calls
np.testing.assert_array_equal(ids, np.arange(len(res[0])) % 5)
np.testing.assert_allclose(c, g["out4"], rtol=1e-6, atol=1e-6)
print("rank", dist.get_rank(), "ok")
dist.destroy_process_group()
'''


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_tile_loop_gloo(tmp_path, world):
    """world_size 2 and 3 gloo runs of the sharded tile loop reproduce the single-process golden: three tiles of unequal size, one of
    them skipped by the "reach zero" error -- at world 3 the rank that owns it has NO rows (an empty rank in both collectives); two
    collectives (+ one broadcast for the plot), the 60-byte record by default."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, TL_REPO=REPO, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("ok") == world


def test_tile_grid_and_offset_labels_host_logic(golden_dir):
    """Host side of the device tiler: the float32 lay-out arithmetic of the tile grid equals the oracle's
    for random plots, and the host offset-label derivation reproduces the reference's labels on the golden tiles."""
    from oracle import tiles as ot
    from treelearn_amd.util.tiles import _offset_labels_host, tile_grid
    rng = np.random.default_rng(0)
    for _ in range(200):
        x0 = np.float32(rng.uniform(-500, 500)); x1 = np.float32(x0 + rng.uniform(5, 120))
        y0 = np.float32(rng.uniform(-500, 500)); y1 = np.float32(y0 + rng.uniform(5, 120))
        ie = float(rng.choice([8, 3, 5.5, 10])); oe = float(rng.choice([13.5, 2.5, 6.0])); st = float(rng.choice([0.5, 1.0, 0.25]))
        a = ot.tile_grid((x0, x1), (y0, y1), ie, oe, st); b = tile_grid((x0, x1), (y0, y1), ie, oe, st)
        assert a[0].shape == b[0].shape and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    g = np.load(os.path.join(golden_dir, "g11_tiles.npz"))
    for i in g["full_tiles"]:
        off, valid = _offset_labels_host(g[f"tile{i}_coords"], g[f"tile{i}_instance_labels"], g[f"tile{i}_semantic_labels"])
        assert np.array_equal(off.astype(np.float32), g[f"tile{i}_offset_labels"])
        m_off = g[f"tile{i}_masks_sem"] & (g[f"tile{i}_semantic_labels"] != 1) & valid
        assert np.array_equal(m_off, g[f"tile{i}_masks_off"])


def test_column_form_rulebook_property_on_oracle_tables():
    """The column form used by tl_rulebook_compact relies on one property of the canonical voxel order: the present dz
    neighbours of a (dx, dy) column are CONSECUTIVE rows.  Checked on the oracle's rulebook of a ragged two-element batch."""
    from oracle import voxel as ov
    from treelearn_amd.synth import make_tile
    tiles = [make_tile(extent=5.0, voxel=0.2, n_trees=2, fill=0.1, seed=s) for s in (3, 4)]
    pts = np.concatenate([t["points"] for t in tiles]); bid = np.concatenate([np.full(len(t["points"]), i) for i, t in enumerate(tiles)])
    vc = np.asarray(ov.voxelize(torch.from_numpy(pts), torch.zeros(len(pts), 1), torch.from_numpy(bid), 2, 0.2)[1]).astype(np.int64)
    nbr = np.asarray(ov.rulebook_subm(vc))
    nbr = nbr.T if nbr.shape[0] == 27 else nbr                      # [N, 27]
    for c in range(9):
        col = nbr[:, 3 * c:3 * c + 3].astype(np.int64)
        present = col >= 0
        base = np.where(present.any(1), np.where(present, col, np.iinfo(np.int64).max).min(1), -1)
        rank = np.cumsum(present, 1) - present                      # present neighbours below this dz
        np.testing.assert_array_equal(np.where(present, base[:, None] + rank, -1), col)


def test_packed_weight_cache_follows_the_parameter_object():
    """autograd._packed: one packed copy per live parameter, refreshed after an in-place update, dropped when the parameter dies
    (no aliasing through a recycled id())."""
    import gc
    from unittest import mock
    import treelearn_amd.autograd as ag
    with mock.patch.object(ag.ops, "pack_weight", lambda w, dt: w.detach().clone()):
        n0 = len(ag._packed_cache)
        w = torch.nn.Parameter(torch.zeros(4, 1, 1, 1, 4))
        a = ag._packed(w, torch.float32)
        assert ag._packed(w, torch.float32) is a
        with torch.no_grad():
            w.add_(1)
        assert ag._packed(w, torch.float32) is not a
        assert len(ag._packed_cache) == n0 + 1
        del w
        gc.collect()
        assert len(ag._packed_cache) == n0


def test_hdbscan_host_stage_prim_order_of_a_shuffled_tree():
    """tl_hdbscan_prim_order_host (host code of the library, no GPU): a spanning tree handed over in any order / orientation comes back
    in the order Prim's algorithm from point 0 walks it (numpy restatement below), and tl_hdbscan_labels_host then labels it the same."""
    from treelearn_amd import _hip
    L = _hip.lib()
    rng = np.random.default_rng(3)
    n = 400
    X = np.concatenate([rng.normal(0, 0.3, (200, 2)), rng.normal(4, 0.3, (200, 2))])
    D = np.sqrt(((X[:, None] - X[None]) ** 2).sum(-1))
    core = np.sort(D, 1)[:, 9]
    M = np.maximum(np.maximum(core[:, None], core[None]), D)
    reach = np.full(n, np.inf); src = np.zeros(n, np.int64); intree = np.zeros(n, bool); intree[0] = True; cur = 0
    ps, pd, pw = [], [], []
    for _ in range(n - 1):                                   # Prim on the complete graph, smallest index among equal reachabilities
        upd = (M[cur] < reach) & ~intree
        reach[upd] = M[cur][upd]; src[upd] = cur
        j = int(np.argmin(np.where(intree, np.inf, reach)))
        ps.append(src[j]); pd.append(j); pw.append(reach[j]); intree[j] = True; cur = j
    ps, pd, pw = np.array(ps, np.int32), np.array(pd, np.int32), np.array(pw, np.float64)
    perm = rng.permutation(n - 1); flip = rng.random(n - 1) < 0.5
    s = np.ascontiguousarray(np.where(flip, pd, ps)[perm]); d = np.ascontiguousarray(np.where(flip, ps, pd)[perm]); w = np.ascontiguousarray(pw[perm])
    os_, od, ow = np.empty_like(s), np.empty_like(d), np.empty_like(w)
    assert L.tl_hdbscan_prim_order_host(s.ctypes.data, d.ctypes.data, w.ctypes.data, n, os_.ctypes.data, od.ctypes.data, ow.ctypes.data) == 0
    np.testing.assert_array_equal(od, pd); np.testing.assert_array_equal(os_, ps); np.testing.assert_array_equal(ow, pw)
    la, lb = np.empty(n, np.int32), np.empty(n, np.int32)
    assert L.tl_hdbscan_labels_host(ps.ctypes.data, pd.ctypes.data, pw.ctypes.data, n, 10, la.ctypes.data) == 0
    assert L.tl_hdbscan_labels_host(os_.ctypes.data, od.ctypes.data, ow.ctypes.data, n, 10, lb.ctypes.data) == 0
    np.testing.assert_array_equal(la, lb)
    assert set(la.tolist()) >= {0, 1}
    s[0] = s[1]; d[0] = d[1]                                  # a repeated edge: no longer a spanning tree
    assert L.tl_hdbscan_prim_order_host(s.ctypes.data, d.ctypes.data, w.ctypes.data, n, os_.ctypes.data, od.ctypes.data, ow.ctypes.data) != 0


def test_conv_wrapper_rejects_a_rulebook_of_the_wrong_shape():
    """The C ABI takes the rulebook as a bare pointer; ops.conv_fwd / conv_wgrad check its shape first, so handing over the table of
    another level raises instead of reading out of bounds on the device.  (Argument checks only: nothing is launched.)"""
    from treelearn_amd import ops
    x = torch.zeros(10, 32, dtype=torch.bfloat16); w = torch.zeros(27, 32, 32, dtype=torch.bfloat16)
    for bad in (torch.zeros(27, 9, dtype=torch.int32), torch.zeros(8, 10, dtype=torch.int32), torch.zeros(27, 10, dtype=torch.int64),
                torch.zeros(10, 27, dtype=torch.int32).t()):
        with pytest.raises(ValueError, match="rulebook"):
            ops.conv_fwd(x, w, bad, 10)
        with pytest.raises(ValueError, match="rulebook"):
            ops._check_table(bad, 27, 10, x.device)          # what conv_wgrad calls before its launch
    with pytest.raises(ValueError, match="needs a rulebook"):
        ops.conv_fwd(x, w, None, 10)


def test_blk_oracle_decodes_to_the_canonical_rulebook():
    """oracle/blk.py (block-local order + staged rulebook form of include/treelearn_hip.h `tl_blk`): both builders agree, the staged
    form decodes to the canonical table carried into the new order, every unit respects the halo bound, also under forced splitting."""
    from oracle import blk as ob
    from oracle import voxel as ov
    from treelearn_amd.synth import make_tile
    tile = make_tile(extent=7.0, voxel=0.1, n_trees=3, fill=0.1, seed=11)
    pts = tile["points"].astype(np.float32)
    _, vc, _, _ = ov.voxelize(pts, tile["feat"], np.zeros(len(pts), np.int64), 1, 0.1)
    vc = np.asarray(vc, np.int32)
    nbr = np.ascontiguousarray(ov.rulebook_subm(vc).T)
    n = len(vc)
    for hm in (126, 30):
        a, b = ob.build(vc, nbr, hm), ob.build_fast(vc, nbr, hm)
        assert len(a["units"]) == len(b["units"]) and np.array_equal(a["lrb"], b["lrb"])
        for u, w in zip(a["units"], b["units"]):
            assert u[:2] == w[:2] and np.array_equal(u[2], w[2])
        assert np.array_equal(ob.decode(a["units"], a["lrb"], n), a["nn"])
        exp = np.where(nbr[:, a["perm"]] >= 0, a["o2n"][np.clip(nbr[:, a["perm"]], 0, None)], -1)
        assert np.array_equal(a["nn"], exp)
        assert max(len(u[2]) for u in a["units"]) <= hm
        cover = np.zeros(n, np.int64)
        for lo, cnt, _ in a["units"]:
            cover[lo:lo + cnt] += 1
        assert (cover == 1).all()
        assert np.array_equal(a["nn"][13], np.arange(n))                       # the centre tap is the row itself


def test_conv_epilogue_statistics_are_dropped_when_the_features_change_in_place():
    """autograd.set_stats / get_stats: the partial sums a conv epilogue attached to its result are only handed to the BatchNorm while the
    tensor is what the kernel summed -- same version counter, same storage (advisor, round 3)."""
    import torch
    from treelearn_amd.autograd import get_stats, set_stats
    y = torch.randn(16, 8)
    set_stats(y, [("parts", 3, 8)])
    assert get_stats(y) == [("parts", 3, 8)]
    v = y.view_as(y)
    set_stats(v, get_stats(y))                       # the skip hand-back: a view of the same storage
    assert get_stats(v) is not None
    y.mul_(2.0)                                      # an in-place op between the conv and its BatchNorm
    assert get_stats(y) is None and get_stats(v) is None
    assert get_stats(torch.randn(4, 8)) is None


def test_forward_arena_allocator_under_sanitizers(tmp_path):
    """tl_forward's device-memory arena (csrc/tl_arena.h: best fit, coalescing, dry-run planning) is host-only code: random take / give
    sequences against a brute-force model, compiled with AddressSanitizer + UBSan (the CPU build is where sanitizers can run)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "arena_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(root, "treelearn_amd", "csrc"), os.path.join(root, "tests", "tools", "arena_test.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "arena_test OK" in r.stdout, r.stdout + r.stderr


def test_public_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/treelearn_hip.h must compile as C99 (what cgo / JNI / a ctypes generator would feed it to) and as
    C++11, pedantic, without warnings, and a C program must be able to link every declared entry point against the built library."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(root, "include")
    hdr = open(os.path.join(inc, "treelearn_hip.h")).read()
    names = sorted(set(re.findall(r"\b(tl_[a-z0-9_]+)\s*\(", hdr)))
    src = tmp_path / "use.c"
    src.write_text('#include "treelearn_hip.h"\n#include <stdio.h>\nint main(void) {\n  tl_forward_args a; tl_net_desc d; tl_conv_args c; (void)a; (void)d; (void)c;\n'
                   + "".join(f"  printf(\"%p\\n\", (void*)(size_t)&{n});\n" for n in names) + "  return 0;\n}\n")
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++11")):
        r = subprocess.run([cc, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc] + (["-x", "c++"] if cc == "g++" else []) +
                           ["-fsyntax-only", str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    import __graft_entry__
    __graft_entry__.build()
    lib = os.path.join(root, "treelearn_amd", "lib")
    r = subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-L", lib, "-ltreelearn_hip", "-Wl,--unresolved-symbols=ignore-in-shared-libs", "-o", str(tmp_path / "use")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr                      # every declared function resolves against libtreelearn_hip.so


def test_load_checkpoint_handles_mismatched_missing_and_unexpected_keys(tmp_path):
    """reference tree_learn/util/train.py:65-102: a `.pth` of {'net', 'optimizer', 'epoch'}; keys whose size differs from the model's are
    dropped (pretraining with another input width), missing / unexpected keys are reported, the optimizer state is restored, the return
    value is epoch + 1 (1 when the file has no epoch); a DataParallel-style wrapper (`.module`) is looked through."""
    import logging
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import random_state_dict
    from treelearn_amd.util.train import load_checkpoint
    cfg = dict(channels=16, num_blocks=3)
    sd = random_state_dict(3, **cfg)
    src = {k: v.clone() for k, v in sd.items()}
    src["input_conv.0.weight"] = torch.zeros(16, 3, 3, 3, 7)                   # another input width: must be skipped, not raise
    del src["semantic_linear.3.bias"]                                           # missing in the file
    src["some.other.key"] = torch.zeros(1)                                      # unexpected in the file
    model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[64, 64, 64], voxel_size=0.1, **cfg)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    opt.step()
    keep = model.state_dict()["input_conv.0.weight"].clone(); keep_bias = model.state_dict()["semantic_linear.3.bias"].clone()
    f = str(tmp_path / "epoch_7.pth")
    torch.save(dict(net=src, optimizer=opt.state_dict(), epoch=7), f)
    msgs = []
    log = logging.getLogger("tl_test_ckpt"); log.setLevel(logging.INFO)
    h = logging.Handler(); h.emit = lambda r: msgs.append(r.getMessage()); log.addHandler(h)
    opt2 = torch.optim.AdamW(model.parameters(), lr=1e-3)
    class Wrapped:                                                              # DataParallel-like
        module = model
    assert load_checkpoint(f, log, Wrapped(), optimizer=opt2, strict=False) == 8
    new = model.state_dict()
    assert torch.equal(new["input_conv.0.weight"], keep) and torch.equal(new["semantic_linear.3.bias"], keep_bias)
    assert torch.equal(new["unet.blocks.block0.conv_branch.2.weight"], sd["unet.blocks.block0.conv_branch.2.weight"])
    text = "\n".join(msgs)
    assert "size mismatch: input_conv.0.weight" in text and "missing keys in source state_dict: input_conv.0.weight, semantic_linear.3.bias" in text and "unexpected key in source state_dict: some.other.key" in text
    assert len(opt2.state_dict()["state"]) == len(opt.state_dict()["state"]) > 0
    torch.save(dict(net=sd), f)
    assert load_checkpoint(f, None, model) == 1 and model._plan is None


def test_result_sink_files_tiles_in_order_grows_and_reports_errors():
    """util/pipeline._ResultSink (the worker thread that assembles the tile loop's numpy results): tiles are filed in the order they were
    pushed whatever their sizes, the arrays grow past the first estimate, mixed integer types promote as torch.cat / np.concatenate would,
    an empty loop gives None, and an exception on the worker thread is re-raised by finish()."""
    from treelearn_amd.util.pipeline import _ResultSink
    rng = np.random.default_rng(3)
    for hint in (None, 5, 40):
        sink = _ResultSink(hint)
        want = [[] for _ in range(3)]
        for t in range(40):
            n = int(rng.integers(0, 700)) if t else 3
            a = torch.from_numpy(rng.normal(size=(n, 7)).astype(np.float32))
            lab = torch.from_numpy(rng.integers(0, 9, n).astype(np.int32 if t < 20 else np.int64))
            z = torch.zeros((n, 0))
            for w, x in zip(want, (a, lab, z)):
                w.append(x.numpy())
            sink.push(None, (lambda a=a, lab=lab, z=z: [a[:, :7], lab, z]))
        res = sink.finish()
        for r, w in zip(res, want):
            ref = np.concatenate(w, 0)
            assert r.dtype == ref.dtype and r.shape == ref.shape
            np.testing.assert_array_equal(r, ref)
    assert _ResultSink(3).finish() is None
    sink = _ResultSink()
    sink.push(None, lambda: [torch.zeros(4, 2)])
    sink.push(None, lambda: 1 / 0)
    sink.push(None, lambda: [torch.zeros(4, 2)])
    with pytest.raises(ZeroDivisionError):
        sink.finish()
    sink = _ResultSink()
    sink.push(None, lambda: [torch.zeros(4, 2)])
    sink.push(None, lambda: [torch.zeros(4, 3)])
    with pytest.raises(RuntimeError, match="different widths"):
        sink.finish()
