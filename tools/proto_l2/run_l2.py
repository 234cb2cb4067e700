"""PROTOTYPE driver (developer tool): the staged-unit form of the level-2 64 -> 64 conv (conv_l2.hip) against the production gather kernel
on the level-2 rulebook of the config-2 tile.  The block order, the 384-row units, their halo lists and the ten-bit local rulebook are
built here with numpy from the product's canonical geometry (a product version would build them on the device, as tl_blk.hip does for
level 1).

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/proto_l2/conv_l2.hip -o tools/proto_l2/libproto_l2.so
    python tools/proto_l2/run_l2.py
"""
import ctypes, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
import torch
from treelearn_amd import _hip, geometry as G, ops
from treelearn_amd.synth import CONFIGS, make_batch, make_tile

U, POS = 384, 1024
HMAX = POS - 1 - U


def timeit(f, reps=15, warm=2):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def build_units(coords, nbr):
    """coords i32[n,4] (canonical order), nbr i32[27,n] -> perm (new -> canonical), halo i32[units,HMAX], nhalo i32[units], lrb u32[n,9]"""
    n = coords.shape[0]
    c = coords.astype(np.int64)
    bx, by, bz = c[:, 1] >> 3, c[:, 2] >> 3, c[:, 3] >> 3
    key = (((((c[:, 0] * 4096 + (bx >> 2)) * 4096 + (by >> 2)) * 4 + (bx & 3)) * 4 + (by & 3)) * 8192) + bz
    perm = np.argsort(key, kind="stable")
    o2n = np.empty_like(perm); o2n[perm] = np.arange(n)
    t = nbr[:, perm].astype(np.int64)
    nn = np.where(t >= 0, o2n[np.clip(t, 0, None)], -1)                         # [27, n] in new rows
    units = (n + U - 1) // U
    chunk = np.arange(n) // U
    lo = chunk * U
    inside = (nn >= lo[None, :]) & (nn < (lo + U)[None, :])
    outside = (nn >= 0) & ~inside
    ch27 = np.broadcast_to(chunk[None, :], nn.shape)
    pk = np.unique(ch27[outside] * n + nn[outside])                              # (unit, outside row), sorted
    cnt = np.bincount(pk // n, minlength=units)
    start = np.cumsum(cnt) - cnt
    assert U + cnt.max() <= POS - 1, (U, int(cnt.max()))
    pos = np.full(nn.shape, POS - 1, dtype=np.int64)
    pos[inside] = (nn - lo[None, :])[inside]
    pos[outside] = U + np.searchsorted(pk, ch27[outside] * n + nn[outside]) - start[ch27[outside]]
    if os.environ.get("PROTO_FAKE") == "1":                                       # timing experiment: every tap reads the row's own position (no bank conflicts)
        pos = np.broadcast_to((np.arange(n) - lo)[None, :], pos.shape).copy()
    e = pos.T                                                                     # [n, 27]
    lrb = np.zeros((n, 9), np.int64)
    for k in range(27):
        lrb[:, k // 3] |= e[:, k] << (10 * (k % 3))
    halo = np.full((units, HMAX), -1, np.int32)
    rows = (pk % n).astype(np.int32)
    idx_in_unit = np.arange(len(pk)) - start[pk // n]
    halo[pk // n, idx_in_unit] = rows
    return perm, halo, cnt.astype(np.int32), lrb.astype(np.uint32), float((cnt.sum() + n) / n)


def main():
    lib = ctypes.CDLL(os.path.join(HERE, "libproto_l2.so"))
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    lib.proto_l2_conv.restype = i32
    lib.proto_l2_conv.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, i32]
    b = make_batch([make_tile(**CONFIGS["config2"], seed=0)])
    g = G.build_geometry(b["coords"].cuda().float(), b["batch_ids"].cuda().long(), 1, 0.1, 7, [500, 500, 1000])
    lv = g.levels[1]; n = lv.n
    t0 = time.time()
    perm, halo, nhalo, lrb, ratio = build_units(lv.coords.cpu().numpy(), lv.nbr.cpu().numpy())
    units = len(nhalo)
    print("level 2: %d rows, %d units of %d, halo mean %.0f max %d, %.2f staged rows per output row (numpy build %.1f s)" % (n, units, U, nhalo.mean(), nhalo.max(), ratio, time.time() - t0), flush=True)
    dev = "cuda"
    perm_t = torch.from_numpy(perm).to(dev)
    halo_t, nhalo_t = torch.from_numpy(halo).to(dev), torch.from_numpy(nhalo).to(dev)
    lrb_t = torch.from_numpy(lrb.view(np.int32)).to(dev)
    torch.manual_seed(0)
    x = (torch.randn(n, 64, device=dev) * 0.7).bfloat16()
    w_ref = torch.randn(64, 3, 3, 3, 64, device=dev) * 0.05
    w = ops.pack_weight(w_ref, torch.bfloat16)
    sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.3
    x_new = x.index_select(0, perm_t).contiguous()
    out = torch.empty_like(x_new)

    def proto(pro, mode=0):
        rc = lib.proto_l2_conv(x_new.data_ptr(), w._tl_frag.data_ptr(), out.data_ptr(), halo_t.data_ptr(), nhalo_t.data_ptr(), lrb_t.data_ptr(),
                               sc.data_ptr() if pro else None, sh.data_ptr() if pro else None, n, units, _hip.stream(), mode)
        assert rc == 0, rc
        return out

    lib2 = None
    if os.path.exists(os.path.join(HERE, "libproto_l2_v2.so")):          # conv_l2_v2.hip: staging / descriptors / epilogue overlapped with the taps
        lib2 = ctypes.CDLL(os.path.join(HERE, "libproto_l2_v2.so"))
        lib2.proto_l2v2_conv.restype = i32
        lib2.proto_l2v2_conv.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, vp]
    lib3 = None
    if os.path.exists(os.path.join(HERE, "libproto_l2_v3.so")):          # conv_l2_v3.hip: loader waves fill the other half-row stage
        lib3 = ctypes.CDLL(os.path.join(HERE, "libproto_l2_v3.so"))
        lib3.proto_l2v3_conv.restype = i32
        lib3.proto_l2v3_conv.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, vp, i32]
    out3 = torch.empty_like(x_new)

    def proto3(mode=0):
        rc = lib3.proto_l2v3_conv(x_new.data_ptr(), w._tl_frag.data_ptr(), out3.data_ptr(), halo_t.data_ptr(), nhalo_t.data_ptr(), lrb_t.data_ptr(), n, units, _hip.stream(), mode)
        assert rc == 0, rc
        return out3

    out2 = torch.empty_like(x_new)

    def proto2():
        rc = lib2.proto_l2v2_conv(x_new.data_ptr(), w._tl_frag.data_ptr(), out2.data_ptr(), halo_t.data_ptr(), nhalo_t.data_ptr(), lrb_t.data_ptr(), n, units, _hip.stream())
        assert rc == 0, rc
        return out2

    y = ops.conv_fwd(x, w, lv.nbr, n)
    o = proto(False).clone()
    if lib2 is not None:
        o2v = proto2().clone(); torch.cuda.synchronize()
        print("v2 (overlapped): equal to v1: %s (max |diff| to the gather kernel %.3g, rows differing from v1: %d)" % (
            torch.equal(o2v, o), float((o2v.float() - y.index_select(0, perm_t).float()).abs().max()), int((o2v != o).any(1).sum())), flush=True)
    d = (o.float() - y.index_select(0, perm_t).float()).abs().max()
    print("plain: equal to the gather kernel: %s (max |diff| %.3g, max |y| %.3g)" % (torch.equal(o, y.index_select(0, perm_t)), float(d), float(y.float().abs().max())), flush=True)
    if lib3 is not None:
        o3v = proto3().clone(); torch.cuda.synchronize()
        print("v3 (loader waves): max |diff| to the gather kernel %.3g, rows differing from v2: %s" % (
            float((o3v.float() - y.index_select(0, perm_t).float()).abs().max()), int((o3v != o2v).any(1).sum()) if lib2 is not None else "-"), flush=True)
    act = ops.affine_relu(x, sc, sh, True)
    ya = ops.conv_fwd(act, w, lv.nbr, n)
    oa = proto(True).clone()
    da = (oa.float() - ya.index_select(0, perm_t).float()).abs().max()
    print("with BatchNorm + ReLU at staging: equal to affine_relu + gather kernel: %s (max |diff| %.3g)" % (torch.equal(oa, ya.index_select(0, perm_t)), float(da)), flush=True)
    o2 = torch.empty_like(x)
    res = {"gather kernel, one view": [], "gather kernel, two views": [], "staged prototype": [], "staged prototype + prologue": []}
    for rnd in range(5):
        res["gather kernel, one view"].append(timeit(lambda: ops.conv_fwd(x, w, lv.nbr, n)))
        res["gather kernel, two views"].append(timeit(lambda: ops.conv_fwd(x, w, lv.nbr, n, out2=(o2, sc, sh, True))))
        res["staged prototype"].append(timeit(lambda: proto(False)))
        res["staged prototype + prologue"].append(timeit(lambda: proto(True)))
        if lib2 is not None:
            res.setdefault("staged prototype v2 (overlapped)", []).append(timeit(proto2))
        if lib3 is not None:
            res.setdefault("staged prototype v3 (loader waves)", []).append(timeit(proto3))
    for mode, name in ((1, "no staging"), (2, "one tap instead of 27"), (4, "no output stores"), (3, "no staging, one tap"), (7, "barriers and LDS transposition only")):
        print("ablation %-40s %.3f ms" % (name, timeit(lambda: proto(False, mode))), flush=True)
    if lib3 is not None:
        for mode, name in ((1, "v3, loaders issue no DMA"), (4, "v3, no output stores"), (5, "v3, neither")):
            print("ablation %-40s %.3f ms" % (name, timeit(lambda: proto3(mode))), flush=True)
    for k, v in res.items():
        v = sorted(v[1:])
        print("%-32s median %.3f ms (min %.3f max %.3f)" % (k, v[len(v) // 2], v[0], v[-1]), flush=True)


if __name__ == "__main__":
    main()
