import os
import sys

import pytest

try:
    # torch's wheel carries its own copy of the HIP runtime.  In one process it has to come up BEFORE
    # the system runtime that libmvo_hip.so links (the other order leaves torch with "No HIP GPUs are
    # available"), so it is imported here, ahead of any fixture that touches the library -- a full
    # run imports it during collection anyway, a single test file would not.
    import torch  # noqa: F401
except ImportError:  # the CPU-only checks do not need it
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "monocular-visual-odometry_amd"), os.path.join(ROOT, "oracle"), os.path.dirname(__file__)):
    if p not in sys.path:
        sys.path.insert(0, p)


# Every device allocation of libmvo_hip.so starts as 0xFF bytes under the tests (csrc/mvo_alloc.h): a
# kernel must not depend on what hipMalloc happens to hand out.  Inherited by the adapter subprocesses.
os.environ.setdefault("MVO_HIP_POISON_ALLOC", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The shared libraries are build products (git-ignored).  On a fresh checkout compile them
    # once (hipcc cross-compiles gfx950 without a GPU); this builds the product and the checker,
    # it does not substitute one for the other.
    import subprocess

    need = [os.path.join(ROOT, "monocular-visual-odometry_amd", "libmvo_hip.so"),
            os.path.join(ROOT, "monocular-visual-odometry_amd", "host", "adapter_check"),
            os.path.join(ROOT, "oracle", "libmvo_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "monocular-visual-odometry_amd"), "all"])
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"])


@pytest.fixture(scope="session")
def oracle():
    import mvo_oracle_py as orc

    orc.lib()  # builds oracle/libmvo_oracle.so on first use if it is missing
    return orc


@pytest.fixture(scope="session")
def solver():
    """One mvo_hip handle for the whole GPU session; fails loudly when the library or the GPU
    is missing (there is no CPU fallback to hide behind)."""
    from mvo_amd import capi

    s = capi.Solver(0)
    yield s
    s.close()


@pytest.fixture(autouse=True)
def _poisoned_lds(request):
    """Every GPU test starts with NaN patterns in the LDS of all CUs: a kernel that reads LDS it never
    wrote must not get away with the zeros a fresh box happens to hold (LDS survives kernels AND
    processes; this is how an unwritten column of k_lm's second system buffer was found), and ends
    with a check of the guards behind the library's device allocations (csrc/mvo_alloc.h)."""
    if request.node.get_closest_marker("gpu") is not None:
        from mvo_amd import capi

        capi.poison_lds(0)
        yield
        # ... and no kernel of the test wrote past the end of a device allocation (256-byte guards)
        assert capi.guard_violations() == 0
        return
    yield
