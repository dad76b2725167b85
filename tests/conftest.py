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


@pytest.fixture(autouse=True)
def poisoned_allocator(request):
    """GPU tests: before every test, the blocks the caching allocator hands out next are filled with NaN bit patterns (a few
    hundred MB of NaN tensors of many sizes, allocated and released).  A kernel that reads memory it should not -- past the end of
    an operand, a lane that should have been masked -- then fails HERE, in the test that exercises it, instead of only behind some
    other test that happened to leave NaNs in a recycled block (round 5: the buffer form of the tiled kernel read past a short K run,
    and only a particular test order showed it).  LH_TEST_POISON=0 turns it off."""
    if request.node.get_closest_marker("gpu") is None or os.environ.get("LH_TEST_POISON", "1") == "0":
        yield
        return
    import torch
    if torch.cuda.is_available():
        junk = []
        for e in range(9, 27):
            for mul in (1.0, 1.5):
                junk.append(torch.full((int((1 << e) * mul) // 4,), float("nan"), device="cuda"))
        torch.cuda.synchronize()
        del junk
    yield
