"""PROTOTYPE driver (DESIGN.md R3.6 "what comes next"): builds the block-local row order of the level-1 rulebook of the config-2 tile with torch ops,
compiles tools/proto_blk/conv_blk.hip on the box, runs it against the production direct kernel.      python tools/proto_blk/run_blk.py"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch
from treelearn_amd import geometry as G, ops
from treelearn_amd.synth import CONFIGS, make_batch, make_tile

so = "/tmp/libblk.so"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", os.path.join(HERE, "conv_blk.hip"), "-o", so], check=True)
L = ctypes.CDLL(so)
L.conv_blk.restype = ctypes.c_int
L.conv_blk.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

tile = make_tile(**CONFIGS["config2"], seed=0)
b = make_batch([tile])
geom = G.build_geometry(b["coords"].cuda().float(), b["batch_ids"].cuda().long(), 1, 0.1, 7, [500, 500, 1000])
lv = geom.levels[0]
n = lv.n
dev = lv.coords.device
B, R = 8, 64
c = lv.coords.long()
key = ((c[:, 0] * 4096 + (c[:, 1] // B)) * 4096 + (c[:, 2] // B)) * 4096 + (c[:, 3] // B)
perm = torch.argsort(key, stable=True)
o2n = torch.empty_like(perm); o2n[perm] = torch.arange(n, device=dev)
ks = key[perm]
first = torch.ones(n, dtype=torch.bool, device=dev); first[1:] = ks[1:] != ks[:-1]
bstart = torch.nonzero(first).flatten()
bid = torch.cumsum(first.long(), 0) - 1
pib = torch.arange(n, device=dev) - bstart[bid]
ufirst = first | (pib % R == 0)
uid = torch.cumsum(ufirst.long(), 0) - 1
nu = int(uid[-1]) + 1
row0 = torch.nonzero(ufirst).flatten()
nown = torch.bincount(uid, minlength=nu)
nbr = lv.nbr.long()[:, perm]
pres = nbr >= 0
nn = torch.where(pres, o2n[nbr.clamp(min=0)], torch.full_like(nbr, -1))               # [27, n] neighbours in new rows
u_row = uid[None, :].expand_as(nn)
inside = pres & (uid[nn.clamp(min=0)] == u_row)
outside = pres & ~inside
pk = (u_row[outside] * n + nn[outside])
up = torch.unique(pk)                                                                    # sorted distinct (unit, row) pairs = the halo lists, unit by unit
hu = up // n
nh = torch.bincount(hu, minlength=nu)
hstart = torch.cumsum(nh, 0) - nh
# local staged position of every (tap, row)
loc = torch.full_like(nn, 255)
loc[inside] = (nn - row0[uid][None, :])[inside]
posu = torch.searchsorted(up, pk)                                                        # index into `up`
loc[outside] = (nown[u_row[outside]] + posu - hstart[u_row[outside]])
assert int((nown + nh).max()) <= 224, int((nown + nh).max())
# padded halo array (multiples of 16 per unit)
nh16 = (nh + 15) // 16 * 16
h0 = torch.cumsum(nh16, 0) - nh16
halo = torch.full((int(nh16.sum()) + 16,), -1, dtype=torch.int32, device=dev)
halo[(h0[hu] + (torch.arange(up.numel(), device=dev) - hstart[hu]))] = (up % n).int()
unit = torch.stack([row0, nown, h0, nh], 1).int().contiguous()
# local rulebook [nu][4][16][32] u8
lrb = torch.full((nu, 64, 32), 255, dtype=torch.uint8, device=dev)
lr = torch.arange(n, device=dev) - row0[uid]
lrb[uid, lr, :27] = loc.t().to(torch.uint8)
lrb = lrb.contiguous()
print(f"rows {n}, units {nu}, own/unit {float(nown.float().mean()):.1f}, halo/unit {float(nh.float().mean()):.1f}, staged max {int((nown + nh).max())}, "
      f"staged rows per output row {float((nown + nh).sum()) / n:.2f}", flush=True)

torch.manual_seed(0)
x = torch.randn(n, 32, device=dev).bfloat16()
w = torch.randn(32, 3, 3, 3, 32, device=dev) * 0.1
wp = ops.pack_weight(w.reshape(32, 27, 32), torch.bfloat16)
ref = ops.conv_fwd(x, wp, lv.nbr, n)
xn = x[perm].contiguous()
out = torch.zeros(n, 32, device=dev, dtype=torch.bfloat16)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(waves=6, dbg=0):
    rc = L.conv_blk(xn.data_ptr(), wp.data_ptr(), out.data_ptr(), unit.data_ptr(), halo.data_ptr(), lrb.data_ptr(), n, nu, waves, dbg, st)
    assert rc == 0, rc


def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


run(); torch.cuda.synchronize()
got = torch.empty_like(out); got[perm] = out
err = float((got.float() - ref.float()).abs().max() / ref.float().abs().max())
print(f"max |blocked - direct| / max |direct| = {err:.2e}   equal rows {float((got == ref).all(dim=1).float().mean()) * 100:.2f} %", flush=True)
t_ref = timeit(lambda: ops.conv_fwd(x, wp, lv.nbr, n))
for waves in (6, 5, 4):
    print(f"direct kernel {t_ref:.3f} ms | block-local prototype, {waves} waves/CU: {timeit(lambda: run(waves)):.3f} ms", flush=True)
for waves in (104, 102, 101):
    out.zero_(); run(waves); torch.cuda.synchronize()
    got = torch.empty_like(out); got[perm] = out
    print(f"variant R (weights in registers, double-buffered stage), {waves - 100} waves per workgroup, one wave per SIMD: {timeit(lambda: run(waves)):.3f} ms, "
          f"equal rows {float((got == ref).all(dim=1).float().mean()) * 100:.2f} %", flush=True)
for dbg in (1, 2, 4, 3, 7):
    print(f"  variant R ablation {dbg}: {timeit(lambda: run(104, dbg)):.3f} ms", flush=True)
for dbg in (1, 2, 4, 3, 7):
    print(f"  ablation {dbg} (1 no staging DMA, 2 no taps, 4 no stores): {timeit(lambda: run(6, dbg)):.3f} ms", flush=True)
