import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # measured kernel choices of a test session never come from (or go to) state an earlier run left in $HOME: the tuning
    # cache of the session is a fresh file (LH_TUNE_CACHE set by the caller is respected; the shipped database still applies)
    if "LH_TUNE_CACHE" not in os.environ:
        import tempfile
        os.environ["LH_TUNE_CACHE"] = os.path.join(tempfile.mkdtemp(prefix="lh_tune_"), "tune_gfx950.txt")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def resnet_cfg(num_layers=50, style="pytorch", deconv_bias=False, final_kernel=1):
    import types
    ns = types.SimpleNamespace
    extra = ns(NUM_LAYERS=num_layers, DECONV_WITH_BIAS=deconv_bias, NUM_DECONV_LAYERS=3,
               NUM_DECONV_FILTERS=[256, 256, 256], NUM_DECONV_KERNELS=[4, 4, 4], FINAL_CONV_KERNEL=final_kernel)
    return ns(MODEL=ns(EXTRA=extra, STYLE=style))
