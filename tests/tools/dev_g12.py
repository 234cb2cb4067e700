"""Golden g12 (7-level training step): per-slice gradient errors vs the reference, and the level-4 256->128 wgrad against a float64
gather + matmul of the SAME saved activations / output gradients (separates kernel error from upstream numerics)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import random_state_dict
g = np.load(os.path.join(os.path.dirname(__file__), "..", "golden", "g12_train7.npz"))
cfg = json.loads(str(g["cfg"]))
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=cfg["spatial_shape"], voxel_size=cfg["voxel_size"], **cfg["cfg"])
model.load_state_dict(random_state_dict(cfg["seed"], **cfg["cfg"]), strict=True); model = model.cuda().train()
keys = ["coords", "input_feats", "batch_ids", "semantic_labels", "instance_labels", "masks_inner", "masks_off", "masks_sem", "offset_labels", "centers"]
batch = {k: torch.from_numpy(g["in_" + k]) for k in keys}; batch["batch_size"] = int(g["in_batch_size"])
deep = "unet.u.u.u.blocks_tail.block0"
conv = dict(model.named_modules())[deep + ".conv_branch.2"]
saved = {}
def fwd_hook(mod, inp, out):
    saved["x"] = inp[0].features.detach(); saved["geom"] = inp[0].geometry; saved["level"] = inp[0].level
    out.features.register_hook(lambda gr: saved.__setitem__("gout", gr.detach()))
conv.register_forward_hook(fwd_hook)
model.zero_grad(); loss, ld = model(batch, return_loss=True); loss.backward()
P = dict(model.named_parameters())
def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64); return np.abs(a - b).max() / np.abs(b).max()
checks = {"grad_input_conv": P["input_conv.0.weight"].grad, "grad_sem3": P["semantic_linear.3.weight"].grad,
          "grad_l4_cat_conv_centre": P[deep + ".conv_branch.2.weight"].grad[:, 1, 1, 1, :], "grad_l4_cat_conv_corner": P[deep + ".conv_branch.2.weight"].grad[:, 0, 2, 1, :],
          "grad_l4_1x1": P[deep + ".i_branch.0.weight"].grad, "grad_l6_deconv": P["unet.u.u.u.u.u.deconv.2.weight"].grad[:, 1, 0, 1, :],
          "grad_l7_conv_centre": P["unet.u.u.u.u.u.u.blocks.block0.conv_branch.2.weight"].grad[:, 1, 1, 1, :], "grad_l2_down": P["unet.u.conv.2.weight"].grad[:, 1, 1, 0, :]}
print("loss", float(loss), "golden", float(g["train_loss"]))
for k, v in checks.items():
    a = v.cpu().numpy().astype(np.float64); b = g[k].astype(np.float64)
    cos = (a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum())
    print(f"{k:28s} max-rel {rel(a, b):.3e}  rms-rel {np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()):.3e}  cos {cos:.8f}")
x, gout = saved["x"].double(), saved["gout"].double()
nbr = saved["geom"].levels[saved["level"]].nbr
print("level", saved["level"], "rows", x.shape, gout.shape)
ours = P[deep + ".conv_branch.2.weight"].grad
for (a, b, c) in ((1, 1, 1), (0, 2, 1)):
    k = (a * 3 + b) * 3 + c
    idx = nbr[k].long(); ok = idx >= 0
    ref = gout[ok].T @ x[idx[ok]]                      # [Cout, Cin]
    o = ours[:, a, b, c, :].double()
    print(f"tap {a}{b}{c}: ours vs float64 of our own activations: {float((o - ref).abs().max() / ref.abs().max()):.3e};  golden vs that: "
          f"{rel(g['grad_l4_cat_conv_centre' if k == 13 else 'grad_l4_cat_conv_corner'], ref.cpu().numpy()):.3e}")

# every gradient against float64 autograd through the oracle
from oracle import model as om
_, g64 = om.train_step_grads(random_state_dict(cfg["seed"], **cfg["cfg"]), batch, cfg["voxel_size"], cfg["cfg"]["num_blocks"], cfg["spatial_shape"])
errs = sorted(((rel(P[n].grad.cpu().numpy(), g64[n].numpy()), n) for n in g64 if n in P and P[n].grad is not None), reverse=True)
for e, n in errs[:12]:
    print(f"{e:.3e}  {n}  shape {tuple(P[n].shape)}  |g|max {float(g64[n].abs().max()):.3e}")
e, n = next((e, n) for e, n in errs if P[n].dim() == 5)
print('per tap:', n)
a = P[n].grad.cpu().numpy().astype(np.float64); b = g64[n].numpy()
if a.ndim == 5:
    for t in range(27):
        i, j, kk = t // 9, (t // 3) % 3, t % 3
        aa, bb = a[:, i, j, kk, :], b[:, i, j, kk, :]
        print(f"  tap {i}{j}{kk}: max-rel {np.abs(aa - bb).max() / max(np.abs(bb).max(), 1e-30):.3e}  |ref|max {np.abs(bb).max():.3e}  |ours|max {np.abs(aa).max():.3e}")
print("levels n:", [lv.n for lv in saved["geom"].levels])
d = np.abs(a - b); w = np.unravel_index(np.argsort(d.ravel())[-8:], d.shape)
for q in range(8):
    ix = tuple(int(x[q]) for x in w); print("  worst element", ix, "ours", a[ix], "f64", b[ix])
names = [str(s_) for s_ in g["grad_names"]]; j = names.index(n)
print("norms: ours", float(P[n].grad.norm()), "f64", float(g64[n].norm()), "golden", float(g["grad_norms"][j]))
