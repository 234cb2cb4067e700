#!/usr/bin/env python3
"""Headline benchmark: Mpoints/s through the sparse U-Net forward on a synthetic 40x40 m / 0.1 m tile
(BASELINE.json configs[1]).  One step = model(batch, return_loss=False) on one tile whose points are
already resident in HBM: voxel hashing, all 13 rulebooks, 65 sparse convs, heads.

    python bench.py --gpus N --steps K --warmup W [--dtype bf16|fp32] [--workload config2|config4|config5]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)

With --gpus N > 1 and no launcher environment (WORLD_SIZE unset) this process only SPAWNS the N ranks
(`python -m torch.distributed.run`, a child process -- the parent never touches the GPU), relays rank 0's
JSON line and exits with the children's status.

Multi-GPU: tiles are independent -> every rank runs its own tile, no data-path
collective; weak scaling; value = points processed by all ranks / max-over-ranks time.
Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = the gather-GEMM conv, timed live with
HIP events) and `cpu_baseline` (the oracle port on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime starts (treelearn_amd/__init__.py: streams of the tile loop + side streams)

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_MFMA_F32_TFLOPS = 157.3       # MI355X_MICROARCH.md: fp32-input MFMA = vector rate
PEAK_MFMA_BF16_TFLOPS = 2500.0     # dense
PEAK_HBM_GBS = 8000.0


def conv_work(meta):
    """Algorithmic work of one conv launch: flops = 2*pairs*Cin*Cout;
    compulsory bytes = (N_in*Cin + N_out*Cout [+ N_out*Cout residual]) * e + 8*pairs."""
    t = meta.get("table")
    if "pairs" in meta:                                      # launch records of the C-side executor: pairs counted per (level, kind) beforehand
        pairs = meta["pairs"]
    elif t is None:
        pairs = meta["n_out"]
    elif hasattr(t, "count_pairs"):                          # block-local form of a rulebook (geometry.BlockedRulebook)
        pairs = getattr(t, "_bench_pairs", None)              # cached ON the object (an id()-keyed cache would outlive the rulebook)
        if pairs is None:
            pairs = t._bench_pairs = t.count_pairs()
    else:
        pairs = int((t >= 0).sum())
    e = meta["esize"]
    flops = 2.0 * pairs * meta["Cin"] * meta["Cout"]
    sp = meta.get("split")
    if sp is not None:
        # one input-channel half of a wider conv (the 64 -> 32 decoder conv of level 1 runs as two 32 -> 32 launches, the second taking the
        # first one's result as residual): the LOGICAL conv's compulsory bytes are charged once, to part 0; the hand-over is not algorithmic
        part, cin = sp
        byts = ((meta["n_in"] * cin + meta["n_out"] * meta["Cout"]) * e + 8.0 * pairs) if part == 0 else 0.0
        return flops, byts, pairs
    byts = (meta["n_in"] * meta["Cin"] + meta["n_out"] * meta["Cout"] * (2 if meta["residual"] else 1)) * e + 8.0 * pairs
    return flops, byts, pairs


def conv_event_pass(model, gbatch, reps):
    """Live per-launch timing of the conv launches of `reps` lone forwards of `model` on `gbatch`, HIP events on the launch stream around
    every launch.  The product path is measured: when the forward runs through the C-side executor (tl_forward) its own event recorder
    is switched on (tl_exec_profile) and the rulebook pair counts of the same tile come from one separately built geometry; a
    configuration that stays on the Python-driven engine is timed through ops.PROFILE.  Returns [[(ms, meta), ...] per forward]."""
    from treelearn_amd import ops

    def step():
        with torch.no_grad():
            return model(gbatch, return_loss=False)
    step(); torch.cuda.synchronize()
    ex = model._executor(model.active_dtype(False)) if hasattr(model, "_executor") else None
    out = []
    if ex is not None:
        with torch.no_grad():
            _, geom = model._voxelize(gbatch["coords"].float(), gbatch["input_feats"].float(), gbatch["batch_ids"].long(), gbatch["batch_size"])
        pairs = {}
        for li, lv in enumerate(geom.levels):
            pairs[(li, "subm")] = int((lv.nbr >= 0).sum())
            if lv.child is not None:
                pairs[(li, "down")] = pairs[(li, "inverse")] = int((lv.child >= 0).sum())
        pairs[(0, "input")] = pairs[(0, "subm")]
        del geom
        ex.profile(True)
        try:
            step(); torch.cuda.synchronize()
            for _ in range(reps):
                step()
                recs = ex.profile_read()
                for r in recs:
                    r["pairs"] = r["n_out"] if r["kind"] == "1x1" else pairs[(r["level"], r["kind"])]
                out.append([(r["ms"], r) for r in recs])
        finally:
            ex.profile(False)
        return out
    for _ in range(reps):
        ops.PROFILE = []
        try:
            step(); torch.cuda.synchronize()
            out.append([(e0.elapsed_time(e1), m) for e0, e1, m in ops.PROFILE])
        finally:
            ops.PROFILE = None
    return out


def host_cores():
    """CPU cores this process may actually use: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:                                    # noqa: BLE001
        pass
    return n


def cpu_baseline_worker():
    """Oracle (CPU port of the same path) on a bounded sample: a 26x26 m tile of the config-2
    generator (about 40 % of the 40x40 m workload, 10-20 s of CPU work), same voxel size / model; all usable host cores.
    Runs in a child process."""
    from oracle import model as om
    from treelearn_amd.synth import make_tile
    cores = host_cores()
    torch.set_num_threads(cores)
    t = make_tile(extent=26.0, voxel=0.1, n_trees=27, fill=0.10, seed=0)
    pts = t["points"]; n = len(pts)
    sd = om.random_state_dict(7, channels=32, num_blocks=7)
    t0 = time.time()
    om.forward(sd, pts, t["feat"], np.zeros(n, np.int64), 1, voxel_size=0.1, num_blocks=7, spatial_shape=[500, 500, 1000])
    dt = time.time() - t0
    return dict(value=n / dt / 1e6, unit="Mpoints/s", cores=cores, kind="port",
                sample=f"26x26 m tile of the config-2 generator (voxel 0.1 m, 7-level 32-ch model), {n} points, 1 forward, "
                       f"fp32 torch-CPU oracle, {dt:.1f} s")


def cpu_baseline(timeout_s=240):
    """Timed in a child process (never touches the GPU; bounded by a timeout so the bench cannot hang)."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(host_cores()), MKL_NUM_THREADS=str(host_cores()), HIP_VISIBLE_DEVICES="")
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker"], capture_output=True, text=True,
                           timeout=timeout_s, env=env)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:                               # noqa: BLE001
        return dict(value=None, unit="Mpoints/s", cores=host_cores(), kind="port", sample=f"failed: {type(e).__name__}")


def emit(res, used_rccl=False):
    """The ONE JSON line, and the last thing on stdout.  RCCL writes a version banner that reaches stdout only when the process
    exits (after anything Python prints; flushing C stdio does not bring it forward), so a rank that created a process group ends
    with os._exit right after its line: the group is already destroyed, nothing else is pending, and the banner never follows the result."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:                                        # noqa: BLE001
        pass
    sys.stdout.flush()
    print(json.dumps(res), flush=True)
    if used_rccl:
        sys.stderr.flush()
        os._exit(0)


def power_probe(step_fn, seconds=2.5):
    """Board power and shader clock while the forward loops (rocm-smi sampled from a thread; not part of the timed region).
    The mid-level conv kernels run at the 1400 W cap, so the clock is part of the story."""
    import re, subprocess, threading
    out = {}

    def sample():
        time.sleep(seconds * 0.55)
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
            m = re.search(r"sclk clock level[^(]*\((\d+)Mhz\)", txt)
            w = re.search(r"Power \(W\):\s*([0-9.]+)", txt)
            if m: out["sclk_mhz"] = int(m.group(1))
            if w: out["board_w"] = float(w.group(1))
        except Exception:                                    # noqa: BLE001
            pass

    th = threading.Thread(target=sample); th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            step_fn()
        torch.cuda.synchronize()
    th.join()
    return out or None


def pmc_traffic(dtype, workload):
    """HBM bytes per step of the conv launches from the committed rocprofv3 PMC passes (profiles/*/traffic.json:
    2 x FETCH_SIZE + WRITE_SIZE, KB units, gfx950 half-count correction) -- collected offline, not in this run."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "*", "traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:                                    # noqa: BLE001
            continue
        if d.get("dtype") == dtype and d.get("workload") == workload and (best is None or d.get("sequence", 0) >= best[0].get("sequence", 0)):
            best = (d, f)                                    # the most recently collected set (`sequence` = collection time), else the last path
    return best


def sharded_plot(model, dist, rank, world, n_tiles, steps, warmup):
    """BASELINE config 4: `n_tiles` overlapping-tile forwards of one plot sharded over the ranks (round-robin = LPT on equal
    tiles), every rank materialising only its own tiles, inputs resident in HBM; one step = the whole plot through
    util.sharding.get_pointwise_preds_sharded: per-rank tile loop (inner-square filter on the device) + the two collectives of
    the record gather INSIDE the timed region.  Returns (seconds per plot (max over ranks), total points, gathered rows)."""
    from treelearn_amd.synth import PLOT4, make_plot, plot_squares
    from treelearn_amd.util.sharding import TileList, assign_tiles, get_pointwise_preds_sharded
    from treelearn_amd.util.tiles import PlotTiler
    # ONE synthetic plot (68 x 68 m for 64 tiles), resident in HBM on every rank; the tiles are its overlapping 40 x 40 m crops (inner squares of
    # 8 m every 4 m: neighbouring tiles share 36 m, the ensemble has duplicates to average), cut on the device by tl_tile_crop (PlotTiler)
    side = int(np.ceil(n_tiles ** 0.5))                                  # (a count that is not a square takes the first n_tiles squares of the next grid)
    geo = dict(PLOT4, tiles_per_side=side)
    plot = make_plot(**geo, seed=0)
    inner, outer = plot_squares(**geo)
    tiler = PlotTiler(plot["points"], plot["instance_label"].astype(np.float32), plot["feat"])
    mine = assign_tiles([1] * n_tiles, world)[rank]
    batches = {}
    for i in mine:
        b = tiler.tile_batch(inner[i], outer[i], geo["inner_edge"], offset_labels="none", tile_index=i)
        assert b is not None, "every inner square of the synthetic plot holds points"
        torch.cuda.current_stream().wait_event(b.pop("_ready_event"))
        batches[i] = b
    torch.cuda.synchronize()
    del tiler
    npts = torch.zeros(n_tiles, dtype=torch.int64, device="cuda")
    for i in mine:
        npts[i] = batches[i]["coords"].shape[0]
    dist.all_reduce(npts)                                                # setup only (point counts for the metric), not timed
    src = TileList([1] * n_tiles, lambda i: batches[i])
    run = lambda: get_pointwise_preds_sharded(model, src, dict(voxel_size=0.1), return_device=True)     # noqa: E731
    for _ in range(max(warmup, 1)):
        out = run()
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = run()
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], device="cuda", dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    return float(dt) / steps, int(npts.sum()), int(out[0].shape[0])


def host_resident_tile_loop(model, n_tiles=16, reps=3):
    """What a drop-in user of the reference's tile loop gets: the tiles arrive as HOST tensors (DataLoader batches, pin_memory=True:
    tree_learn/util/train.py:132-141; the loop of tree_learn/util/pipeline.py:83-103) -> util.pipeline.get_pointwise_preds: H2D of the next
    tile on a copy stream into a ring of persistent staging buffers, four tiles in flight, inner-square filter on the device, ONE packed D2H
    of the surviving rows per tile, results as numpy arrays -- PCIe both ways inside the timed region.  `n_tiles` distinct config-2 tiles
    (four generator seeds x four symmetries).  Returns the block for the bench line (best and median of `reps` passes over all tiles)."""
    from treelearn_amd.synth import CONFIGS, make_batch, make_tile, tile_variant
    from treelearn_amd.util import get_pointwise_preds
    base = [make_tile(**CONFIGS["config2"], seed=100 + s) for s in range(4)]
    tiles = []
    for i in range(n_tiles):
        b = make_batch([tile_variant(base[i % 4], (i // 4) % 8)], inner_square_edge_length=8.0)
        tiles.append({k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in b.items()})
    npts = sum(b["coords"].shape[0] for b in tiles)
    res = get_pointwise_preds(model, tiles, dict(voxel_size=0.1))                   # warm-up: staging ring, per-stream arenas
    times = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = get_pointwise_preds(model, tiles, dict(voxel_size=0.1))
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return dict(value=npts / med / 1e6, unit="Mpoints/s", ms_per_tile=med * 1e3 / n_tiles, ms_per_tile_best=times[0] * 1e3 / n_tiles, tiles=n_tiles, passes=reps,
                total_points=npts, rows_returned=int(len(res[0])), h2d_mb_per_tile=round(npts / n_tiles * 20 / 1e6, 1),
                path="pinned host tiles -> H2D ring on a copy stream -> 4 tiles in flight -> device-side inner filter -> one packed D2H per tile -> numpy",
                note="PCIe-inclusive (both directions inside the timed region); `value` of the headline is device-resident by the metric's definition")


def training_step_bench(args, rank, world, dist):
    """BASELINE config 3: the step body of reference tools/training/train.py:30-44 on a batch of two 40x40 m crops, random-init
    default model: zero_grad, forward with loss (mixed precision = the reference's autocast regime, bf16 here: no GradScaler
    needed), the two loss .item() reads, backward, clip_grad_norm_(1.0), AdamW step (lr 3e-3, wd 1e-3: configs/training/train.yaml)."""
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict
    from treelearn_amd import ops
    cfg = CONFIGS["config2"]
    batch = make_batch([make_tile(**cfg, seed=2 * rank + s) for s in (0, 1)])
    n_pts = batch["coords"].shape[0]
    # bf16: compute_dtype switch, no scaler needed.  fp16: the REFERENCE's regime verbatim -- torch.autocast(float16) around the forward and a
    # GradScaler around backward / step (tools/training/train.py:32,40-44) on a model left at its fp32 default; the float16 training kernels run
    fp16 = args.dtype == "fp16"
    dtype = torch.float32 if (args.dtype == "fp32" or fp16) else torch.bfloat16
    model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=cfg["voxel"], compute_dtype=dtype)
    model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7), strict=True)
    model = model.cuda().train()
    # torch's own AdamW in its single-kernel form (`fused=True`; the reference builds the optimizer from the config's kwargs, util/train.py:105-110,
    # so `fused: true` in the yaml selects it there too): the for-each form is ~36 launches and 2.3 ms of a host-bound tail per step
    fused = os.environ.get("TL_BENCH_ADAMW", "fused") == "fused"
    opt = torch.optim.AdamW(model.parameters(), lr=3e-3, weight_decay=1e-3, fused=fused)
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}

    scaler = torch.amp.GradScaler("cuda", enabled=fp16)
    skipped = [0]

    def step():
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16, enabled=fp16):
            loss, ld = model(gb, return_loss=True)
        vals = [v.detach().cpu().item() for v in ld.values()]
        scaler.scale(loss).backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0, norm_type=2)
        s0 = scaler.get_scale() if fp16 else 1.0
        scaler.step(opt)
        scaler.update()
        if fp16:
            skipped[0] += int(scaler.get_scale() < s0)
        return vals

    for _ in range(max(args.warmup, 1)):
        vals = step()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        vals = step()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device="cuda", dtype=torch.float64); npts = torch.tensor([float(n_pts)], device="cuda", dtype=torch.float64)
    if dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX); dist.all_reduce(npts, op=dist.ReduceOp.SUM)
    dt = float(tmax)
    if rank != 0:
        return None
    # algorithmic conv flops of one step: forward + dgrad + wgrad = 3 x (2 * pairs * Cin * Cout) over the 71 conv launches
    model.eval()
    fl = sum(conv_work(m)[0] for _, m in conv_event_pass(model, gb, 1)[0])
    model.train()
    sec = dt / args.steps
    peak = PEAK_MFMA_F32_TFLOPS if args.dtype == "fp32" else PEAK_MFMA_BF16_TFLOPS
    ach = 3.0 * fl / sec / 1e12
    return dict(metric="Mpoints/sec through sparse U-Net training step (fwd + bwd + AdamW; 0.1 m voxel, 2 x 40x40 m crops)",
                value=float(npts) / sec / 1e6, unit="Mpoints/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=sec * 1e3,
                higher_is_better=True, scaling="weak", vs_baseline=None, dtype=args.dtype, data="synthetic",
                config=dict(workload=f"config3: training step, batch of 2 crops of 40x40 m at 0.1 m ({n_pts} points), default 7-level 32-ch model "
                                     f"(30.1 M params, random init), torch.optim.AdamW({'fused=True' if fused else 'for-each'}), grad-norm clip 1.0, " + {"bf16": "bf16 mixed precision (autocast-like)", "fp32": "fp32", "fp16":
                            "float16 autocast + GradScaler (the reference's regime, tools/training/train.py:32,40-44)"}.get(args.dtype, args.dtype),
                            points_per_step=n_pts, last_losses=vals, **(dict(gradscaler_skipped_steps=skipped[0], gradscaler_final_scale=scaler.get_scale()) if fp16 else {})),
                roofline=dict(bound="mfma", achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak, traffic=None,
                              note="algorithmic conv flops of the step (forward + dgrad + wgrad = 3 x 2 * pairs * Cin * Cout) / whole step time"),
                cpu_baseline=None)


DTYPES = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32, "bf16x3": "bf16x3"}     # bf16x3: fp32 storage, split-bf16 contraction


def forward_block(workload, dtype_name, steps, warmup, nfl, trained_like=False):
    """A compact forward measurement of another BASELINE config inside the default run (config 5: the 0.05 m stress tile), so that the
    driver's one line carries it: same step definition as the headline (model(batch, return_loss=False), inputs resident in HBM, `nfl`
    independent tiles in flight), its own conv-family roofline from a live HIP-event pass."""
    from treelearn_amd import ops
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict
    cfg = CONFIGS[workload]
    batch = make_batch([make_tile(**cfg, seed=0)])
    n_pts = batch["coords"].shape[0]
    dtype = DTYPES[dtype_name]
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats(); mem0 = torch.cuda.memory_allocated()
    model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000] if cfg["voxel"] >= 0.1 else None, voxel_size=cfg["voxel"], compute_dtype=dtype)
    model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7), strict=True)
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    if trained_like:
        # "trained-like" weights: the synthetic ones with every BatchNorm's running statistics set to this tile's statistics (one
        # training-mode forward at momentum 1, untimed).  A random-init net with arbitrary running statistics reaches 1e5 > fp16's 65504 in
        # places; a trained net's activations are O(1), which is what fp16 inference presumes (the reference relies on it under autocast)
        model.compute_dtype = torch.bfloat16
        model = model.cuda().train()
        bns = [m_ for m_ in model.modules() if isinstance(m_, torch.nn.BatchNorm1d)]
        for m_ in bns:
            m_.momentum = 1.0
        with torch.no_grad():
            model(gb, return_loss=False)
        for m_ in bns:
            m_.momentum = 0.1
        model.compute_dtype = dtype
    model = model.cuda().eval()

    def step():
        with torch.no_grad():
            return model(gb, return_loss=False)
    streams = [torch.cuda.Stream() for _ in range(nfl)] if nfl > 1 else []

    def run(k):
        if not streams:
            for _ in range(k): step()
            return
        cur = torch.cuda.current_stream()
        for st in streams: st.wait_stream(cur)
        for i in range(k):
            with torch.cuda.stream(streams[i % nfl]): step()
        for st in streams: cur.wait_stream(st)
    step(); torch.cuda.synchronize()
    run(max(nfl, 1)); torch.cuda.synchronize()
    run(max(warmup, 1)); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); sec = (time.perf_counter() - t0) / steps
    recs = conv_event_pass(model, gb, 1)[0]
    last = step(); torch.cuda.synchronize()
    nonfinite = sum(int((~torch.isfinite(last[k])).sum()) for k in ("semantic_prediction_logits", "offset_predictions"))
    ms = sum(t for t, _ in recs)
    fl = by = 0.0
    for _, m in recs:
        f, b, _ = conv_work(m); fl += f; by += b
    ach = by / (ms * 1e-3) / 1e9
    out = dict(value=n_pts / sec / 1e6, unit="Mpoints/s", ms_per_step=sec * 1e3, steps=steps, warmup=warmup, dtype=dtype_name, tiles_in_flight=nfl,
               weights="synthetic, BatchNorm running statistics = the tile's batch statistics (trained-like)" if trained_like else "synthetic random init",
               nonfinite_outputs=nonfinite, active_voxels=(model._plan._exec.last["level_n"][0] if getattr(model._plan, "_exec", None) else None),
               peak_hbm_gb=(torch.cuda.max_memory_allocated() - mem0) / 1e9,
               workload=f"{workload}: single {cfg['extent']:.0f}x{cfg['extent']:.0f} m tile, voxel {cfg['voxel']} m, {n_pts} points, {model.num_blocks}-level 32-ch sparse U-Net fwd",
               roofline=dict(bound="hbm", achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS, traffic=None, launches_per_step=len(recs),
                             conv_ms_per_step=ms, algorithmic_gb_per_step=by / 1e9, mfma_tflops=fl / (ms * 1e-3) / 1e12))
    del model, gb
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child `torch.distributed.run` (fresh processes;
    this parent never initialises the GPU and never re-execs), relay their output, print rank 0's JSON line last."""
    import socket
    import subprocess
    have = torch.cuda.device_count()                                     # device_count() does not create a HIP context
    if have < n and os.environ.get("TL_BENCH_SHARE_GPU") != "1":
        print(f"bench.py: --gpus {n} requested but only {have} GPU(s) are visible on this node; one GPU per rank is required "
              f"(ranks never share a device).  Run with --gpus {max(have, 1)} or on a node with {n} GPUs.", file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    js = [ln for ln in lines if ln.startswith("{") and '"metric"' in ln]
    for ln in lines:
        if not js or ln is not js[-1]:
            print(ln)
    if js:
        print(js[-1], flush=True)
    return r.returncode if (r.returncode or js) else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32", "bf16x3"],
                    help="fp16: the float16 inference kernels (forward workloads; the training step runs its 16-bit kernels in bf16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-mode", action="store_true", help="skip the fp32 parity-mode reference timing (profiling runs)")
    ap.add_argument("--no-extra-workloads", action="store_true", help="skip the config-3 training step and the config-5 tile that ride along in the default line")
    ap.add_argument("--no-power-probe", action="store_true", help="skip the 2.5 s rocm-smi power/clock sample (profiling runs)")
    ap.add_argument("--tiles-in-flight", type=int, default=4, help="independent steps overlapped on this many streams (1 = strictly one after the other)")
    ap.add_argument("--layer-table", default=None, help="write the per-launch table of the conv event pass (time, work, both roofs) to this file")
    ap.add_argument("--workload", default="config2", choices=["config2", "config3", "config4", "config5"])
    ap.add_argument("--plot-tiles", type=int, default=64, help="config4: tiles of the plot")
    ap.add_argument("--no-sharded-plot", action="store_true", help="N > 1: skip the config-4 sharded tile loop + gather measurement")
    ap.add_argument("--cpu-baseline-worker", action="store_true")
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        print(json.dumps(cpu_baseline_worker()), flush=True)
        return

    if (args.gpus > 1 or os.environ.get("TL_BENCH_FORCE_SPAWN") == "1") and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))                                 # parent: no GPU call before or after this line
                                                                         # (TL_BENCH_FORCE_SPAWN=1: take this path with --gpus 1 too, for testing)
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    torch.set_num_threads(max(1, host_cores() // int(os.environ.get("LOCAL_WORLD_SIZE", world))))     # ranks share the host-core quota
    local = int(os.environ.get("LOCAL_RANK", 0))
    # TEST switches (tests/test_gpu_configs.py): TL_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and TL_BENCH_BACKEND=gloo replaces RCCL (which
    # refuses two ranks on one device), so that a 1-GPU box can run the whole N > 1 flow of this file -- barriers, max-over-ranks timing, the
    # sharded plot's collectives, rank 0's line.  Numbers from such a run mean nothing and say so (`config.test_mode`).
    share = os.environ.get("TL_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("TL_BENCH_BACKEND", "nccl")
    if share:
        local = 0
    if torch.cuda.device_count() <= local:                               # counting devices does not initialise the GPU
        sys.exit(f"bench.py: rank {rank} needs GPU {local} but only {torch.cuda.device_count()} are visible (one GPU per rank, no sharing)")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get("TL_BENCH_FORCE_DIST") == "1":          # the env switch lets a 1-GPU box exercise the RCCL path
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from treelearn_amd import ops
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict

    if args.workload == "config3":
        res = training_step_bench(args, rank, world, dist)
        if dist:
            dist.destroy_process_group()
        if rank == 0:
            emit(res, used_rccl=bool(dist))
        return
    if args.workload == "config4":
        if dist is None:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(_free_port()))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
        model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1,
                          compute_dtype=DTYPES[args.dtype])
        model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7), strict=True)
        model = model.cuda().eval()
        sec, total_pts, rows = sharded_plot(model, dist, rank, world, args.plot_tiles, args.steps, args.warmup)
        if rank == 0:
            res = dict(metric="Mpoints/sec through sparse U-Net fwd (0.1 m voxel, 40x40 m tile)", value=total_pts / sec / 1e6, unit="Mpoints/s",
                       n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=sec * 1e3, higher_is_better=True, scaling="strong",
                       vs_baseline=None, dtype=args.dtype, data="synthetic",
                       config=dict(workload=f"config4: whole-plot inference, {args.plot_tiles} 40x40 m tiles (voxel 0.1 m, 8 m inner squares every 4 m) of ONE synthetic plot, cut on the device (PlotTiler), sharded "
                                            f"round-robin over {world} GPU(s), device-resident record gather (2 collectives) inside the timed region",
                                   tiles=args.plot_tiles, total_points=total_pts, gathered_rows=rows, ms_per_tile=sec * 1e3 / args.plot_tiles * world),
                       roofline=None, cpu_baseline=None)
        dist.destroy_process_group()
        if rank == 0:
            emit(res, used_rccl=True)
        return

    used_rccl_extra = False
    cfg = CONFIGS["config5_20m" if args.workload == "config5" else args.workload]     # (config 5 at BASELINE's ~20 M active voxels)
    tile = make_tile(**cfg, seed=rank)                                 # every rank its own tile
    batch = make_batch([tile])
    n_pts = batch["coords"].shape[0]
    dtype = DTYPES[args.dtype]
    model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000] if cfg["voxel"] >= 0.1 else None,
                      voxel_size=cfg["voxel"], compute_dtype=dtype)
    model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7), strict=True)
    model = model.cuda().eval()
    model.return_backbone_feats = True
    gbatch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}     # inputs resident in HBM

    def step():
        with torch.no_grad():
            return model(gbatch, return_loss=False)

    # Tiles in flight: consecutive steps (independent tiles) go to NF streams round-robin, so the stretches of one forward that
    # leave most CUs idle (36 small launches of the deep levels, geometry kernels + their two host syncs) are filled by the big
    # convs of the neighbouring steps -- what the production tile loop (util/pipeline.get_pointwise_preds) does.  Every step still
    # does all of its work inside the timed region; `value` is throughput.  --tiles-in-flight 1 gives the one-after-the-other time.
    nfl = max(1, args.tiles_in_flight)
    streams = [torch.cuda.Stream() for _ in range(nfl)] if nfl > 1 else []

    def run_steps(k):
        if not streams:
            for _ in range(k):
                out = step()
            return out
        cur = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(cur)
        for i in range(k):
            with torch.cuda.stream(streams[i % nfl]):
                out = step()
        for st in streams:
            cur.wait_stream(st)
        return out

    step(); torch.cuda.synchronize()                                   # builds the inference plan (packed weights) before steps spread over streams
    if streams:                                                        # setup, like the plan: one forward per stream, so that every stream's pool of the
        run_steps(nfl); torch.cuda.synchronize()                       # caching allocator exists however small --warmup is (a cold pool = hipMalloc in the timed region)
    run_steps(max(args.warmup, 1))
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
    npts = torch.tensor([float(n_pts)], device="cuda", dtype=torch.float64)
    if dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(npts, op=dist.ReduceOp.SUM)
    dt = float(tmax); total_pts = float(npts)

    # live per-kernel timing of the conv launches (separate pass: event pairs perturb the pipeline)
    roof = None
    if rank == 0:
        reps = 3
        passes = conv_event_pass(model, gbatch, reps)
        per = len(passes[0])
        tot_ms = sum(t for ps in passes for t, _ in ps) / reps
        flops = byts = 0.0
        for _, m in passes[0]:
            f, b, _ = conv_work(m); flops += f; byts += b
        avg_ms = tot_ms / per
        if args.layer_table:
            # per launch: measured time against its own two roofs -- matrix work if every tap of every 32-row block is contracted
            # (what an output-stationary MFMA kernel must do: "dense-equivalent") and compulsory bytes
            with open(args.layer_table, "w") as f:
                f.write(f"# {args.workload} {args.dtype}, one tile at a time inside the event pass; ms = mean of {reps} runs (HIP events on the launch stream)\n")
                f.write("#  i   K  Cin->Cout     rows  pairs/row      ms  TFLOP/s(present)  dense-equiv %of 2.5PF  compulsory GB  GB/s  t_mfma@2.5PF  t_hbm@8TB/s (ms)\n")
                lo = 0.0
                for i in range(per):
                    m = passes[0][i][1]
                    ms = sum(passes[r][i][0] for r in range(reps)) / reps
                    fl, by, pairs = conv_work(m)
                    dense = 2.0 * ((m["n_out"] + 31) // 32 * 32) * m["K"] * m["Cin"] * m["Cout"]
                    t_m, t_h = dense / (PEAK_MFMA_BF16_TFLOPS * 1e12) * 1e3, by / (PEAK_HBM_GBS * 1e9) * 1e3
                    lo += max(t_m, t_h)
                    f.write(f"{i:4d} {m['K']:3d} {m['Cin']:4d}->{m['Cout']:<4d} {m['n_out']:8d} {pairs / max(m['n_out'], 1):10.2f} {ms:7.3f} {fl / ms / 1e9:17.0f} "
                            f"{100 * dense / (ms * 1e-3) / (PEAK_MFMA_BF16_TFLOPS * 1e12):22.1f} {by / 1e9:14.3f} {by / ms / 1e6:5.0f} {t_m:13.3f} {t_h:12.3f}\n")
                f.write(f"# sum of measured times {tot_ms:.3f} ms; sum over launches of max(t_mfma at 2.5 PFLOP/s on dense-equivalent work, t_hbm at 8 TB/s on compulsory bytes) = {lo:.3f} ms\n")
        if args.dtype == "fp32":
            ach = flops / (tot_ms * 1e-3) / 1e12
            roof = dict(bound="mfma", achieved=ach, peak=PEAK_MFMA_F32_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_MFMA_F32_TFLOPS, traffic=None)
        else:
            ach = byts / (tot_ms * 1e-3) / 1e9
            roof = dict(bound="hbm", achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS, traffic=None,
                        mfma_tflops=flops / (tot_ms * 1e-3) / 1e12)
        tr = pmc_traffic(args.dtype, args.workload)
        if tr is not None:
            roof["traffic"] = tr[0]["hbm_gb_per_step"]
            roof["traffic_unit"] = "GB per step (sum over the conv launches; PMC 2*FETCH_SIZE+WRITE_SIZE)"
            roof["traffic_source"] = os.path.relpath(tr[1], REPO)
            roof["traffic_measured_in_run"] = False          # PMC passes are collected offline (rocprofv3 --pmc, profiles/*/traffic.json), never in this run
            # the north-star's "HBM bandwidth on the rulebook gather": PMC bytes of the conv kernels / their measured time
            roof["traffic_gbs"] = roof["traffic"] / (tot_ms * 1e-3)
            roof["traffic_frac_of_peak"] = roof["traffic_gbs"] / PEAK_HBM_GBS
        roof.update(kernel="tl_conv_fwd family (k_conv_streamq / k_conv_stream / k_conv_blk / k_conv_direct / k_conv_up / k_conv_small / k_conv_ones27)", launches_per_step=per, conv_ms_per_step=tot_ms, avg_launch_ms=avg_ms,
                    algorithmic_gflop_per_step=flops / 1e9, algorithmic_gb_per_step=byts / 1e9)

    if rank == 0:
        res = dict(metric=f"Mpoints/sec through sparse U-Net fwd ({cfg['voxel']:g} m voxel, 40x40 m tile)",
                   value=total_pts * args.steps / dt / 1e6, unit="Mpoints/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype=args.dtype, data="synthetic",
                   config=dict(workload=f"{args.workload}: single {cfg['extent']:.0f}x{cfg['extent']:.0f} m tile, voxel {cfg['voxel']} m, "
                                        f"{n_pts} points/tile, 7-level 32-ch sparse U-Net fwd (30.1 M params, random init), 1 tile per GPU",
                               points_per_tile=n_pts, tiles_per_step=world, tiles_in_flight=nfl),
                   roofline=roof)
        if share or backend != "nccl":
            res["config"]["test_mode"] = f"ranks share GPU 0 / backend {backend}: a plumbing test of the N > 1 flow, not a measurement"

        if nfl > 1:                                                    # the same forward strictly one tile after the other, for reference
            with torch.no_grad():
                for _ in range(2): step()
                torch.cuda.synchronize(); t1 = time.perf_counter()
                for _ in range(8): step()
                torch.cuda.synchronize(); d1 = (time.perf_counter() - t1) / 8
            res["one_tile_at_a_time"] = dict(value=n_pts / d1 / 1e6, unit="Mpoints/s", ms_per_step=d1 * 1e3)
        # SURVEY.md 8d's latency definition next to the throughput `value`: wall time of ONE model(batch, return_loss=False), device-resident
        # input to device-resident outputs, strictly sequential (a device synchronisation after every forward), median of 24
        lat, enq = [], []
        with torch.no_grad():
            for _ in range(3): step()
            torch.cuda.synchronize()
            for _ in range(24):
                t1 = time.perf_counter(); step(); t2 = time.perf_counter(); torch.cuda.synchronize()
                lat.append((time.perf_counter() - t1) * 1e3); enq.append((t2 - t1) * 1e3)
        lat.sort(); enq.sort()
        res["latency_ms_median"] = 0.5 * (lat[11] + lat[12])
        ex = model._executor(model.active_dtype(False))
        res["latency"] = dict(median_ms=res["latency_ms_median"], min_ms=lat[0], max_ms=lat[-1], forwards=24,
                              host_enqueue_ms_median=0.5 * (enq[11] + enq[12]),
                              launch_path="tl_forward: one C-ABI call per tile enqueues geometry + convs + heads" if ex is not None else "Python-driven engine: one ctypes call per launch",
                              definition="one model(batch, return_loss=False) on device-resident input, torch.cuda.synchronize() after each (SURVEY.md 8d); "
                                         "host_enqueue = time until the call returns (includes the two geometry read-backs)")
        if world == 1 and args.dtype in ("bf16", "fp16") and not args.no_fp32_mode and args.workload == "config2":
            # the fp32 parity mode (the precision the 1e-3 parity gate is checked in), same tile, for reference
            m32 = TreeLearn(use_feats=False, use_coords=False, spatial_shape=model.spatial_shape, voxel_size=cfg["voxel"], compute_dtype=torch.float32)
            m32.load_state_dict(model.state_dict(), strict=True); m32 = m32.cuda().eval()
            with torch.no_grad():
                for _ in range(2): m32(gbatch, return_loss=False)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                for _ in range(5): m32(gbatch, return_loss=False)
                torch.cuda.synchronize(); d32 = (time.perf_counter() - t1) / 5
            res["fp32_parity_mode"] = dict(value=n_pts / d32 / 1e6, unit="Mpoints/s", ms_per_step=d32 * 1e3)
            out32 = m32(gbatch, return_loss=False)
            del m32
            # the parity-FAST mode: fp32 storage / BatchNorm / residuals / heads, the convs of the large levels contracted as split-bf16
            # products on the bf16 matrix cores (compute_dtype="bf16x3"); held to the same 1e-3 gate by the tests, reported beside the exact mode
            m3 = TreeLearn(use_feats=False, use_coords=False, spatial_shape=model.spatial_shape, voxel_size=cfg["voxel"], compute_dtype="bf16x3")
            m3.load_state_dict(model.state_dict(), strict=True); m3 = m3.cuda().eval()
            with torch.no_grad():
                for _ in range(2): out3 = m3(gbatch, return_loss=False)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                for _ in range(8): m3(gbatch, return_loss=False)
                torch.cuda.synchronize(); d3 = (time.perf_counter() - t1) / 8
            dev3 = {k: float((out3[k] - out32[k]).abs().max() / out32[k].abs().max()) for k in ("semantic_prediction_logits", "offset_predictions")}
            res["parity_fast_mode"] = dict(value=n_pts / d3 / 1e6, unit="Mpoints/s", ms_per_step=d3 * 1e3, compute_dtype="bf16x3",
                                           max_rel_dev_from_exact_fp32=dev3,
                                           note="fp32 storage; conv contraction = 3 bf16 MFMAs on hi/lo splits (tl_conv_args.weight_x3); one tile at a time like fp32_parity_mode")
            del m3, out3, out32
        if world == 1 and args.workload == "config2" and not args.no_extra_workloads:
            # BASELINE configs 3 and 5 ride along in the default line (each with its own ms_per_step / value / roofline), so the driver's
            # one run carries them: the training step (config 3) and the 0.05 m stress tile (config 5)
            import types
            try:
                ts = training_step_bench(types.SimpleNamespace(steps=8, warmup=3, dtype=args.dtype), 0, 1, None)
                res["training_step"] = {k: ts[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "roofline")}
                if args.dtype == "bf16":
                    # the reference's own regime beside it: float16 autocast + GradScaler on the float16 training kernels
                    t16 = training_step_bench(types.SimpleNamespace(steps=6, warmup=3, dtype="fp16"), 0, 1, None)
                    res["training_step"]["fp16_autocast_gradscaler"] = dict(ms_per_step=t16["ms_per_step"], value=t16["value"], unit=t16["unit"],
                                                                            gradscaler_skipped_steps=t16["config"]["gradscaler_skipped_steps"],
                                                                            gradscaler_final_scale=t16["config"]["gradscaler_final_scale"], last_losses=t16["config"]["last_losses"])
            except Exception as e:                                      # noqa: BLE001
                res["training_step"] = dict(error=f"{type(e).__name__}: {e}")
            torch.cuda.empty_cache()
            try:
                # BASELINE words config 5 as fp16: float16 kernels on trained-like weights, overflow count in the block
                ex0 = model._executor(model.active_dtype(False))
                if ex0 is not None:
                    ex0.release_memory()                                # the headline model's arenas (re-allocated by its next forward)
                torch.cuda.empty_cache()
                # BASELINE's size: ~20 M active voxels (fill 0.16); two tiles in flight (a tile of this size fills the chip by itself and
                # every stream in flight holds an arena of ~35 GB)
                res["config5"] = forward_block("config5_20m", "fp16" if args.dtype == "bf16" else args.dtype, 5, 2, min(nfl, 2), trained_like=True)
            except Exception as e:                                      # noqa: BLE001
                res["config5"] = dict(error=f"{type(e).__name__}: {e}")
        if world == 1 and args.workload == "config2" and not args.no_extra_workloads and dist is None:
            # BASELINE config 4 in the default line: the 64-tile plot through the sharded tile loop at world 1 (reference
            # tools/pipeline/pipeline.py:66-94 -> util/pipeline.py:79-109), inner-square filter and the device-resident record gather
            # (2 collectives on a world-1 RCCL group) inside the timed region
            try:
                import torch.distributed as d4
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(_free_port()))
                d4.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
                used_rccl_extra = True
                sec4, tp4, rows4 = sharded_plot(model, d4, 0, 1, args.plot_tiles, 1, 1)
                res["config4"] = dict(value=tp4 / sec4 / 1e6, unit="Mpoints/s", ms_per_plot=sec4 * 1e3, ms_per_tile=sec4 * 1e3 / args.plot_tiles, tiles=args.plot_tiles,
                                      total_points=tp4, gathered_rows=rows4, collectives_per_plot=2, n_gpus=1, dtype=args.dtype,
                                      workload=f"config4: whole-plot inference, {args.plot_tiles} overlapping 40x40 m tiles (voxel 0.1 m, 8 m inner squares every 4 m) of ONE synthetic plot through "
                                               "get_pointwise_preds_sharded on 1 GPU, record gather timed")
                d4.destroy_process_group()
            except Exception as e:                                      # noqa: BLE001
                res["config4"] = dict(error=f"{type(e).__name__}: {e}")
        if world == 1 and args.workload == "config2" and not args.no_extra_workloads:
            # the drop-in user's number: the reference's loop hands `forward` host tensors and takes numpy arrays back
            try:
                res["host_resident_tile_loop"] = host_resident_tile_loop(model)
            except Exception as e:                                      # noqa: BLE001
                res["host_resident_tile_loop"] = dict(error=f"{type(e).__name__}: {e}")
        if world == 1 and not args.no_power_probe:
            pw = power_probe(step)
            if pw:
                res["power"] = dict(pw, note="rocm-smi sample while the same forward loops for 2.5 s (untimed)")
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
            res["config"]["cpu_baseline_sample"] = "26x26 m tile of the same generator / voxel size / model (about 40 % of the points of the timed tile), 1 forward"
    if dist and (world > 1 or os.environ.get("TL_BENCH_FORCE_DIST") == "1") and args.workload == "config2" and not args.no_sharded_plot:
        # BASELINE config 4 alongside: 8 tiles per rank through the sharded tile loop WITH the record gather timed (weak form of the
        # 64-tiles-on-8-GPUs plot; `--workload config4` times the fixed 64-tile plot instead)
        # (a ride-along block: the headline above is already measured, so a failure here -- this is the one flow no 1-GPU box can rehearse with RCCL
        # across devices -- is reported in the line instead of costing the line)
        try:
            sec, tp, rows = sharded_plot(model, dist, rank, world, 8 * world, 2, 1)
            if rank == 0:
                res["sharded_plot"] = dict(value=tp / sec / 1e6, unit="Mpoints/s", tiles=8 * world, ms_per_plot=sec * 1e3, gathered_rows=rows,
                                           collectives_per_plot=2, note="tile loop + inner-square filter + device-resident record gather, timed together")
        except Exception as e:                                         # noqa: BLE001
            if rank == 0:
                res["sharded_plot"] = dict(error=f"{type(e).__name__}: {e}"[:400])
    if dist:
        # proof that the collective library saw `world` ranks: an all-reduce of ones over the group the timing barriers used
        try:
            one = torch.ones(1, device="cuda", dtype=torch.float32)
            dist.all_reduce(one)
            if rank == 0:
                res["rccl_world"] = int(round(float(one)))
            dist.destroy_process_group()
        except Exception as e:                                         # noqa: BLE001
            if rank == 0:
                res["rccl_world"] = f"{type(e).__name__}: {e}"[:200]
    if rank == 0:
        emit(res, used_rccl=bool(dist) or used_rccl_extra)


if __name__ == "__main__":
    main()
