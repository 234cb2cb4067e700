import sys; sys.path.insert(0, "/root/repo")
import torch, numpy as np
from treelearn_amd import ops
torch.manual_seed(0)
def run(cin, cout, n_in, n_out, with_out2):
    x = torch.randn(n_in, cin, device="cuda").bfloat16()
    w = ops.pack_weight(torch.randn(cout, 1, 1, 1, cin, device="cuda") * 0.05, torch.bfloat16)
    tab = torch.randint(0, n_in, (1, n_out), device="cuda", dtype=torch.int32)
    out = torch.empty(n_out, cout, device="cuda", dtype=torch.bfloat16)
    kw = {}
    if with_out2:
        kw["out2"] = (torch.empty_like(out), torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda"), True)
    y = ops.conv_fwd(x, w, tab, n_out, out=out, **kw)
    torch.cuda.synchronize()
    ref = (x.float()[tab[0].long()] @ w[0].float().T)
    print(cin, cout, n_in, n_out, with_out2, "max err", float((y.float() - ref).abs().max()), flush=True)
for args in ((64, 32, 30000, 20000, False), (64, 32, 30000, 20000, True), (32, 64, 30000, 20000, False), (32, 64, 30000, 20000, True),
             (32, 64, 1849940, 1105126, True), (96, 64, 300000, 250000, True)):
    run(*args)
