"""End-to-end parity at workload scale: the HIP forward against oracle.model.forward (the CPU restatement of tree_learn/model/tree_learn.py:75-103
+ blocks.py + SURVEY.md Appendix B) on whole tiles.  The oracle's forwards take minutes of host time; tests/conftest.py starts them as child
processes when the session begins and this file sorts LAST, so they run beside the other GPU tests instead of in front of them.

  config2_28m   a 28x28 m tile of the config-2 generator (0.9 M points; the kernel mix of the 40 m tile)
  config2_full  THE config-2 tile (40x40 m, 1.89 M points: BASELINE.json configs[1], the workload the headline is quoted on)
  config5_like  14x14 m at 0.05 m voxels (~1.2 M points), spatial_shape=None (the derived shape, as config 5 needs: 800 > 500)
each in the exact fp32 mode and -- same gate, same oracle run -- in the parity-fast mode bf16x3."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import ORACLE_JOBS
from treelearn_amd.synth import make_batch, make_tile, random_state_dict

REL_TOL = 1e-3          # BASELINE.json north_star: semantic / offset tensors within 1e-3 relative fp32


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def _model(dtype, voxel, sshape, seed):
    from treelearn_amd.model import TreeLearn
    m = TreeLearn(use_feats=False, use_coords=False, spatial_shape=list(sshape) if sshape is not None else None, voxel_size=voxel, compute_dtype=dtype)
    m.load_state_dict(random_state_dict(seed, channels=32, num_blocks=7), strict=True)
    return m.cuda().eval()


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("job", ["config2_28m", "config2_full", "config5_like"])
def test_forward_fp32_and_bf16x3_vs_oracle_end_to_end(job, oracle_runs, tile2_batch):
    tile_kw, vs, sshape, seed = ORACLE_JOBS[job]
    batch = tile2_batch if job == "config2_full" else make_batch([make_tile(**tile_kw)])
    model = _model(torch.float32, vs, sshape, seed)
    with torch.no_grad():
        out = model(batch, return_loss=False)
        # the parity-fast mode (fp32 storage, split-bf16 contraction on the bf16 matrix cores) is held to the SAME gate by the same oracle run
        model_x3 = _model("bf16x3", vs, sshape, seed)
        out_x3 = model_x3(batch, return_loss=False)
    assert model_x3._plan.x3 and not model._plan.x3
    assert not torch.equal(out_x3["backbone_feats"], out["backbone_feats"]), "the bf16x3 plan ran the exact kernels"
    ref = oracle_runs.result(job)
    assert int(ref["n_points"]) == batch["coords"].shape[0]
    if job == "config2_full":
        assert batch["coords"].shape[0] > 1_800_000
    errs = {}
    for k in ("backbone_feats", "semantic_prediction_logits", "offset_predictions"):
        errs[k] = (rel_err(out[k].cpu().numpy(), ref[k]), rel_err(out_x3[k].cpu().numpy(), ref[k]))
    print(f"{job}: rel err vs oracle (exact fp32, bf16x3):", errs)
    for k, (e32, e3) in errs.items():
        assert e32 < REL_TOL, (k, e32)
        assert e3 < REL_TOL, ("bf16x3", k, e3)
