import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """CPU tests take seconds each (the whole `-m "not gpu"` suite ~40 s): a 300 s ceiling per test (pytest-timeout, where installed) turns a
    hang into a failure with every thread's stack instead of a run that never returns."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for it in items:
        if it.get_closest_marker("gpu") is None and it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(300))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# Full-size tiles are built ONCE per session and shared by the test files (a config-2 tile takes seconds, the config-5 tile half a minute of numpy).
@pytest.fixture(scope="session")
def tile2_batch():
    """THE config-2 tile of bench.py (seed 0: 1.89 M points) as a CPU batch dict."""
    from treelearn_amd.synth import CONFIGS, make_batch, make_tile
    return make_batch([make_tile(**CONFIGS["config2"], seed=0)])


@pytest.fixture(scope="session")
def tile5_batch():
    """BASELINE config 5 at its stated size -- what bench.py's `config5` block runs: CONFIGS["config5_20m"] (fill 0.16: 19.0 M points, 17.9 M
    voxels at 0.05 m), seed 0, as a CPU batch dict."""
    from treelearn_amd.synth import CONFIGS, make_batch, make_tile
    return make_batch([make_tile(**CONFIGS["config5_20m"], seed=0)])


# ------------------------------------------------------------------------------------------------ CPU-oracle forwards beside the GPU tests
# The end-to-end parity tests compare the HIP forward of workload-size tiles with oracle.model.forward, which takes minutes on the host
# cores.  Those runs do not need the GPU: they are started as child processes when the session begins (only if a selected test asks for them)
# and run BESIDE the GPU tests; the tests that consume them sort last (tests/test_gpu_zz_oracle_end_to_end.py) and only wait for what is
# still running.  A child regenerates its tile from the generator's seed (deterministic numpy), so nothing but the result crosses.
ORACLE_JOBS = {
    # name: (make_tile kwargs, voxel size, spatial_shape, model seed)
    "config2_28m": (dict(extent=28.0, voxel=0.1, n_trees=31, fill=0.10, seed=0), 0.1, [500, 500, 1000], 7),
    "config2_full": (dict(extent=40.0, voxel=0.1, n_trees=64, fill=0.10, seed=0), 0.1, [500, 500, 1000], 7),
    "config5_like": (dict(extent=14.0, voxel=0.05, n_trees=8, fill=0.12, seed=4), 0.05, None, 7),
}
_ORACLE_CHAINS = (("config2_full", "config2_28m", "config5_like"),)         # ONE niced child on a third of the host cores: it only has to be done when the
                                                                             # last test file starts (two chains on 3/4 of the cores slowed every other test 1.5-5 x)
_oracle_state = {}


def _granted_cores():
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:                                        # noqa: BLE001
        pass
    return max(1, n)


_CHILD = """
import os, sys, json
os.nice(15)
import numpy as np, torch
sys.path.insert(0, {repo!r})
torch.set_num_threads({threads})
from oracle import model as om
from treelearn_amd.synth import make_batch, make_tile, random_state_dict
for name, (tile_kw, vs, sshape, seed) in {jobs!r}:
    b = make_batch([make_tile(**tile_kw)])
    o = om.forward(random_state_dict(seed, channels=32, num_blocks=7), b["coords"].numpy(), b["input_feats"].numpy(), b["batch_ids"].numpy(), 1,
                   voxel_size=vs, num_blocks=7, spatial_shape=sshape)
    tmp = os.path.join({out!r}, name + ".tmp.npz")
    np.savez(tmp, n_points=b["coords"].shape[0], **{{k: o[k].numpy() for k in ("backbone_feats", "semantic_prediction_logits", "offset_predictions")}})
    os.replace(tmp, os.path.join({out!r}, name + ".npz"))
"""


def pytest_collection_finish(session):
    if _oracle_state or session.config.option.collectonly or not any("oracle_runs" in getattr(it, "fixturenames", ()) for it in session.items):
        return
    import subprocess, tempfile
    out = tempfile.mkdtemp(prefix="tl_oracle_")
    threads = max(2, _granted_cores() // 3)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    procs = []
    for chain in _ORACLE_CHAINS:
        code = _CHILD.format(repo=REPO, threads=threads, jobs=[(n, ORACLE_JOBS[n]) for n in chain], out=out)
        procs.append((chain, subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
    _oracle_state.update(dir=out, procs=procs)


def pytest_sessionfinish(session, exitstatus):
    import shutil
    for _, pr in _oracle_state.get("procs", ()):
        if pr.poll() is None:
            pr.kill()                                        # (the exact children this session started)
    if _oracle_state.get("dir"):
        shutil.rmtree(_oracle_state["dir"], ignore_errors=True)


class _OracleRuns:
    def result(self, name, timeout=1500):
        """dict of the oracle's outputs for ORACLE_JOBS[name] (waits for the child that computes it)."""
        import time
        import numpy as np
        path = os.path.join(_oracle_state["dir"], name + ".npz")
        proc = next(pr for chain, pr in _oracle_state["procs"] if name in chain)
        t0 = time.time()
        while not os.path.exists(path):
            if proc.poll() is not None and not os.path.exists(path):
                raise RuntimeError(f"oracle child for {name} exited with {proc.returncode}: {proc.stderr.read().decode()[-2000:]}")
            if time.time() - t0 > timeout:
                raise TimeoutError(f"oracle child for {name} still running after {timeout} s")
            time.sleep(0.5)
        return dict(np.load(path))


@pytest.fixture(scope="session")
def oracle_runs():
    assert _oracle_state, "the oracle children are started at collection time (pytest_collection_finish)"
    return _OracleRuns()
