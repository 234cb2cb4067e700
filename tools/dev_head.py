"""Dev: time the per-point head kernel (tl_head_mlp) on the config-2 tile."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import ops, _hip
N, M, C = 1890607, 1772986, 32
x = torch.randn(M, C, device="cuda").to(torch.bfloat16)
v2p = torch.randint(0, M, (N,), device="cuda")
so = torch.rand(C, device="cuda") + 0.5; ho = torch.randn(C, device="cuda") * 0.1
w1 = torch.randn(2, C, C, device="cuda") * 0.2; b1 = torch.randn(2, C, device="cuda") * 0.1
w2 = torch.randn(5, C, device="cuda") * 0.2; b2 = torch.randn(5, device="cuda") * 0.1
v2p_sorted = torch.sort(v2p).values
ref = None
for mode in (1, 0):
    _hip.check(_hip.lib().tl_set_tuning(b"head_mode", mode), "head_mode")
    bbf, lg, of = ops.head_mlp(x, v2p, so, ho, w1, b1, w2, b2, True)
    if ref is None: ref = (bbf.clone(), lg.clone(), of.clone())
    else:
        print("MFMA head vs scalar-weight head: backbone identical =", bool(torch.equal(bbf, ref[0])),
              " logits max |diff| = %.3g (max |ref| %.3g)" % (float((lg - ref[1]).abs().max()), float(ref[1].abs().max())),
              " offsets max |diff| = %.3g (max |ref| %.3g)" % (float((of - ref[2]).abs().max()), float(ref[2].abs().max())))
  
for hm, nm, vp, bb in [(m,) + t for m in (1, 0) for t in (("random v2p, backbone written", v2p, True), ("random v2p, no backbone", v2p, False), ("sorted v2p, backbone written", v2p_sorted, True), ("sorted v2p, no backbone", v2p_sorted, False))]:
    _hip.check(_hip.lib().tl_set_tuning(b"head_mode", hm), "head_mode")
    run = lambda: ops.head_mlp(x, vp, so, ho, w1, b1, w2, b2, bb)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): run()
    e1.record(); torch.cuda.synchronize()
    print(f"head_mode {hm}, {nm:30s}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us")
