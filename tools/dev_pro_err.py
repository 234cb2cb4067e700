"""Dev: error of the bf16 forward against the fp32 forward for the three level-1 forms (canonical order, block-local with two views,
block-local with the prologue at staging)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_blk as T

batch = T._batch(16.0, [5])
for settle in (False, True):
    m = T._model(torch.bfloat16, False, settle_on=batch if settle else None)
    m.compute_dtype = torch.float32
    os.environ["TL_BLK"] = "0"
    with torch.no_grad():
        ref = {k: v.float().cpu() for k, v in m(batch, return_loss=False).items()}
    m.compute_dtype = torch.bfloat16
    for name, blk, pro in (("canonical", "0", "0"), ("blk 2-view", "1", "0"), ("blk staged", "1", "1")):
        os.environ["TL_BLK_PRO"] = pro
        o = T._fwd(m, batch, blk == "1")
        print("settled=%d %-11s" % (settle, name), "  ".join("%s max %.4f mean %.5f" % (k[:12], float((o[k] - ref[k]).abs().max() / ref[k].abs().max()),
              float((o[k] - ref[k]).abs().mean() / ref[k].abs().mean())) for k in ref))
