"""What the per-tile geometry costs the THROUGHPUT (tiles in flight on several streams): the config-2 forward as the tile loop runs it,
against the same loop with the geometry built once and only network + heads issued per step.  python tools/dev_geom_value.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import treelearn_amd
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict

torch.set_num_threads(treelearn_amd.host_cores() if hasattr(treelearn_amd, "host_cores") else 8)
b = make_batch([make_tile(**CONFIGS["config2"], seed=0)])
gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
m = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
m.load_state_dict(random_state_dict(7, channels=32, num_blocks=7), strict=True)
m = m.cuda().eval()


def loop(f, nf, steps=60, warm=8):
    streams = [torch.cuda.Stream() for _ in range(nf)]
    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % nf]):
                f()
    run(warm); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(steps); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


with torch.no_grad():
    h = m.prepare(gb); torch.cuda.synchronize()
    for nf in (1, 2, 4):
        full = loop(lambda: m(gb, return_loss=False), nf)
        os.environ["TL_EXEC"] = "0"
        full_py = loop(lambda: m(gb, return_loss=False), nf)
        del os.environ["TL_EXEC"]
        net = loop(lambda: m.infer(h), nf)
        print(f"{nf} in flight: full forward (tl_forward) {full:.3f} ms, (Python engine) {full_py:.3f} ms, network + heads only on a prebuilt geometry {net:.3f} ms "
              f"-> geometry costs {full_py - net:.3f} ms of throughput", flush=True)
