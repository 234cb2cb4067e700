import sys; sys.path.insert(0,"/root/repo")
import torch
from treelearn_amd.autograd import bias_add
for n, C, dt in ((100001, 3, torch.float32), (100002, 2, torch.bfloat16), (70, 3, torch.bfloat16), (5000, 32, torch.float32), (33, 3, torch.float32)):
    x = torch.randn(n, C, device="cuda", dtype=dt, requires_grad=True); b = torch.randn(C, device="cuda", requires_grad=True)
    g = torch.randn(n, C, device="cuda", dtype=dt)
    y = bias_add(x, b); y.backward(g)
    ref = g.double().sum(0)
    print(n, C, dt, float((b.grad.double() - ref).abs().max() / ref.abs().max()), torch.equal(x.grad, g))
