"""Pin the CPU oracle to the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import model as om
from oracle import voxel as ov

T = torch.from_numpy


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_g1_g2_loss(golden_dir):
    g = _load(golden_dir, "g1_g2_loss.npz")
    for c in "abcd":
        args = [T(g[f"{c}_{k}"]) for k in ("logits", "offsets", "masks_sem", "masks_off", "semantic_labels", "offset_labels")]
        sem, off = om.point_wise_loss(*args)
        assert float(sem) == pytest.approx(float(g[f"{c}_pw_sem"]), rel=1e-6, abs=1e-7)
        assert float(off) == pytest.approx(float(g[f"{c}_pw_off"]), rel=1e-6, abs=1e-7)
        loss, ld = om.get_loss(dict(semantic_prediction_logits=args[0], offset_predictions=args[1]),
                               args[4], args[5], args[3], args[2])
        assert float(loss) == pytest.approx(float(g[f"{c}_loss"]), rel=1e-6, abs=1e-7)
        assert float(ld["semantic_loss"]) == pytest.approx(float(g[f"{c}_semantic_loss"]), rel=1e-6, abs=1e-7)
        assert float(ld["offset_loss"]) == pytest.approx(float(g[f"{c}_offset_loss"]), rel=1e-6, abs=1e-7)


@pytest.mark.parametrize("uc,uf", [(False, False), (False, True), (True, False), (True, True)])
def test_g3_voxelize(golden_dir, uc, uf):
    g = _load(golden_dir, "g3_voxelize.npz")
    tag = f"c{int(uc)}f{int(uf)}"
    vf, vc, v2p, ss = ov.voxelize(g["coords"], g["input_feats"], g["batch_ids"], 2, 0.2, uc, uf, 3)
    # reference returns float coords (b,x,y,z) in per-element ascending (x,y,z) order == ascending key
    np.testing.assert_array_equal(vc.astype(np.float32), g[f"{tag}_voxel_coords"])
    np.testing.assert_array_equal(v2p, g[f"{tag}_v2p"])
    np.testing.assert_array_equal(ss.astype(np.float32), g[f"{tag}_spatial_shape"])
    np.testing.assert_allclose(vf, g[f"{tag}_voxel_feats"], rtol=1e-6, atol=1e-6)


def test_g8_manifest(golden_dir):
    with open(os.path.join(golden_dir, "g8_manifest.json")) as f:
        man = json.load(f)
    ours = om.state_dict_manifest(channels=32, num_blocks=7)
    assert man["n_keys"] == 414 == len(ours)
    assert [k for k, _, _ in man["keys"]] == [k for k, _ in ours]
    assert [tuple(s) for _, s, _ in man["keys"]] == [tuple(s) for _, s in ours]
    n_params = sum(int(np.prod(s)) for k, s in ours if "running_" not in k and "num_batches" not in k)
    assert n_params == man["n_params"] == 30106981


def _batch(g, name):
    keys = ["coords", "input_feats", "batch_ids", "semantic_labels", "instance_labels", "masks_inner", "masks_off",
            "masks_sem", "offset_labels", "centers"]
    b = {k: T(g[f"{name}_in_{k}"]) for k in keys}
    b["batch_size"] = int(g[f"{name}_in_batch_size"])
    return b


@pytest.mark.parametrize("name", ["m3", "m7", "m2b"])
def test_g10_forward_eval(golden_dir, name):
    g = _load(golden_dir, "g10_forward.npz")
    cfg = json.loads(str(g[f"{name}_cfg"]))
    sd = om.random_state_dict(cfg["seed"], **cfg["cfg"])
    b = _batch(g, name)
    out = om.forward(sd, b["coords"].numpy(), b["input_feats"].numpy(), b["batch_ids"].numpy(), b["batch_size"],
                     voxel_size=cfg["voxel_size"], num_blocks=cfg["cfg"]["num_blocks"], spatial_shape=cfg["spatial_shape"])
    for k in ("backbone_feats", "semantic_prediction_logits", "offset_predictions"):
        ref = g[f"{name}_eval_{k}"]
        scale = np.abs(ref).max()
        np.testing.assert_allclose(out[k].numpy(), ref, rtol=1e-4, atol=1e-5 * max(scale, 1.0))
    loss, ld = om.get_loss(out, b["semantic_labels"], b["offset_labels"], b["masks_off"], b["masks_sem"])
    assert float(loss) == pytest.approx(float(g[f"{name}_eval_loss"]), rel=1e-4)


@pytest.mark.parametrize("name", ["m3", "m2b"])
def test_g10_forward_train_loss(golden_dir, name):
    g = _load(golden_dir, "g10_forward.npz")
    cfg = json.loads(str(g[f"{name}_cfg"]))
    sd = om.random_state_dict(cfg["seed"], **cfg["cfg"])
    b = _batch(g, name)
    out = om.forward(sd, b["coords"].numpy(), b["input_feats"].numpy(), b["batch_ids"].numpy(), b["batch_size"],
                     voxel_size=cfg["voxel_size"], num_blocks=cfg["cfg"]["num_blocks"], spatial_shape=cfg["spatial_shape"],
                     training=True)
    loss, ld = om.get_loss(out, b["semantic_labels"], b["offset_labels"], b["masks_off"], b["masks_sem"])
    assert float(loss) == pytest.approx(float(g[f"{name}_train_loss"]), rel=2e-4)
    assert float(ld["offset_loss"]) == pytest.approx(float(g[f"{name}_train_offset_loss"]), rel=2e-4)


def test_g10_reach_zero(golden_dir):
    g = _load(golden_dir, "g10_forward.npz")
    assert bool(g["reach_zero_raised"])
    sd = om.random_state_dict(1, channels=8, num_blocks=4)
    with pytest.raises(ValueError, match="reach zero!!!"):
        om.forward(sd, g["rz_in_coords"], g["rz_in_input_feats"], g["rz_in_batch_ids"], 1, voxel_size=0.5, num_blocks=4)


def test_g11_tiles_oracle_matches_reference(golden_dir):
    """the numpy restatement of tile_generate_and_save + TreeDataset/collate (oracle/tiles.py) reproduces every
    array of every tile the reference produced (SHA-1 over the raw bytes: dtype, order and values)."""
    import hashlib
    from oracle import tiles as ot
    g = np.load(os.path.join(golden_dir, "g11_tiles.npz"))
    ie, oe, st, isel = (float(v) for v in g["params"])
    res = ot.plot_tiles(g["points"], g["labels"], g["feats"], ie, oe, st, isel)
    assert len(res) == int(g["n_tiles"])
    keys = [str(k) for k in g["keys"]]
    for i, b in enumerate(res):
        assert len(b["coords"]) == int(g["counts"][i])
        for j, k in enumerate(keys):
            assert hashlib.sha1(np.ascontiguousarray(b[k]).tobytes()).hexdigest() == str(g["digests"][i][j]), (i, k)
    for i in g["full_tiles"]:
        for k in keys:
            ref = g[f"tile{i}_{k}"]
            assert res[i][k].dtype == ref.dtype and np.array_equal(res[i][k], ref), (i, k)


def test_oracle_vs_handwritten_known_answers():
    """The oracle's sparse-conv conventions (tap orientation, [Cout,kx,ky,kz,Cin] layout, odd-extent drop, un-flipped inverse)
    against vectors written by hand from spconv's documented behaviour -- not produced by any code in this repo."""
    import torch
    from oracle import sparse_ops as osp
    from oracle import voxel as ov
    from kat_cases import build_weight, case_points, load_cases
    for case in load_cases():
        pts, bids = case_points(case)
        B = int(bids.max()) + 1
        _, vc, _, _ = ov.voxelize(pts, np.zeros((len(pts), 1), np.float32), bids, B, 1.0)
        vc = np.asarray(vc)
        mins = np.stack([np.asarray(case["coords"])[np.asarray(case["coords"])[:, 0] == b][:, 1:].min(0) for b in range(B)])
        want = np.asarray(case["coords"]).copy(); want[:, 1:] -= mins[want[:, 0]]           # voxelize shifts every batch element to its minimum
        np.testing.assert_array_equal(vc, want, err_msg=case["name"])
        w = torch.from_numpy(build_weight(case))
        if case["kind"] == "subm":
            out = osp.conv_table(torch.tensor(case["feats"]), w, ov.rulebook_subm(vc))
        elif case["kind"] == "down":
            cc, parent, child, _ = ov.rulebook_down(vc, case["shape"])
            np.testing.assert_array_equal(cc, np.asarray(case["expect_coords"]), err_msg=case["name"])
            out = osp.conv_table(torch.tensor(case["feats"]), w, child)
        else:
            cc, parent, child, _ = ov.rulebook_down(vc, case["shape"])
            out = osp.inverse_conv(torch.tensor(case["coarse_feats"]), w, parent, vc)
        np.testing.assert_array_equal(out.numpy(), np.asarray(case["expect"], np.float32), err_msg=case["name"])


def test_g12_default_architecture_eval_and_train_loss(golden_dir):
    """Golden g12 (the reference's default 7-level / 32-channel module tree on a batch of two crops): the oracle reproduces its
    eval outputs and both losses."""
    g = _load(golden_dir, "g12_train7.npz")
    cfg = json.loads(str(g["cfg"]))
    sd = om.random_state_dict(cfg["seed"], **cfg["cfg"])
    keys = ["coords", "input_feats", "batch_ids", "semantic_labels", "offset_labels", "masks_off", "masks_sem"]
    b = {k: T(g[f"in_{k}"]) for k in keys}
    B = int(g["in_batch_size"])
    kw = dict(voxel_size=cfg["voxel_size"], num_blocks=cfg["cfg"]["num_blocks"], spatial_shape=cfg["spatial_shape"])
    out = om.forward(sd, b["coords"].numpy(), b["input_feats"].numpy(), b["batch_ids"].numpy(), B, **kw)
    for k in ("semantic_prediction_logits", "offset_predictions"):
        ref = g[f"eval_{k}"]
        np.testing.assert_allclose(out[k].numpy(), ref, rtol=1e-4, atol=2e-5 * np.abs(ref).max())
    loss, _ = om.get_loss(out, b["semantic_labels"], b["offset_labels"], b["masks_off"], b["masks_sem"])
    assert float(loss) == pytest.approx(float(g["eval_loss"]), rel=2e-4)
    out = om.forward(sd, b["coords"].numpy(), b["input_feats"].numpy(), b["batch_ids"].numpy(), B, training=True, **kw)
    loss, ld = om.get_loss(out, b["semantic_labels"], b["offset_labels"], b["masks_off"], b["masks_sem"])
    assert float(loss) == pytest.approx(float(g["train_loss"]), rel=2e-4)
    assert float(ld["offset_loss"]) == pytest.approx(float(g["train_offset_loss"]), rel=2e-4)


def test_g12_gradients_float64_second_opinion(golden_dir):
    """Golden g12's gradients (reference module tree through the dense stand-in, fp32 on the CPU) against float64 autograd through
    the oracle's rulebook formulation: same loss to 1e-7, gradient norms agree to 4e-4 in the median, but the deep levels differ by
    a systematic 1-2 % (max-norm) that is identical in float32 and float64 autograd of the oracle -- i.e. it belongs to the dense
    stand-in's backward (conv3d / conv_transpose3d gradients on the CPU), not to round-off here.  The GPU test therefore pins the
    HIP gradients to the float64 oracle at 2e-3 and to the golden at 2.5e-2."""
    g = _load(golden_dir, "g12_train7.npz")
    cfg = json.loads(str(g["cfg"]))
    sd = om.random_state_dict(cfg["seed"], **cfg["cfg"])
    batch = {k: g[f"in_{k}"] for k in ("coords", "input_feats", "batch_ids", "semantic_labels", "offset_labels", "masks_off", "masks_sem")}
    batch["batch_size"] = int(g["in_batch_size"])
    loss, grads = om.train_step_grads(sd, batch, cfg["voxel_size"], cfg["cfg"]["num_blocks"], cfg["spatial_shape"])
    assert loss == pytest.approx(float(g["train_loss"]), rel=1e-6)
    names = [str(s) for s in g["grad_names"]]
    n64 = np.array([float(grads[n].norm()) for n in names]); ref = g["grad_norms"]
    big = ref > 1e-6 * ref.max()
    assert np.median(np.abs(ref[big] / n64[big] - 1)) < 1e-3 and np.abs(ref[big] / n64[big] - 1).max() < 2.5e-2
    deep = "unet.u.u.u.blocks_tail.block0"
    for key, val in (("grad_input_conv", grads["input_conv.0.weight"]), ("grad_l4_cat_conv_centre", grads[deep + ".conv_branch.2.weight"][:, 1, 1, 1, :]),
                     ("grad_l6_deconv", grads["unet.u.u.u.u.u.deconv.2.weight"][:, 1, 0, 1, :]), ("grad_sem3", grads["semantic_linear.3.weight"])):
        a = val.numpy(); b = g[key].astype(np.float64)
        assert np.abs(a - b).max() / np.abs(a).max() < 2.5e-2, key


def test_oracle_verticality_vs_hand_derived_planes():
    """oracle/prepare.verticality against answers written down from the definition (tests/kat_cases.verticality_planes): a plane whose
    normal is theta off the vertical has verticality 1 - cos(theta) at every point."""
    from oracle import prepare as op
    from kat_cases import verticality_planes
    for pts, expect in verticality_planes():
        v, _ = op.verticality(pts, 0.6)
        assert not np.isnan(v).any()
        assert np.abs(v - expect).max() < 1e-6, expect


def test_oracle_voxel_downsample_vs_hand_worked_example():
    from oracle import prepare as op
    from kat_cases import voxel_downsample_by_hand
    data, voxel, expect, first, p2v = voxel_downsample_by_hand()
    out, f, m = op.voxelize(data, voxel)
    np.testing.assert_allclose(out, expect, atol=1e-6)
    np.testing.assert_array_equal(f, first); np.testing.assert_array_equal(m, p2v)
