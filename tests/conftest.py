import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
