import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def hip():
    """The product library through its C-ABI; GPU tests only."""
    from vacancy_amd import capi
    return capi.load()


@pytest.fixture(autouse=True)
def _brick_rows_on_small_grids(request):
    """The library takes the brick-row pass of marching cubes (mc_bits_bricks: bricks whose kept minimum lies above the
    iso level are not read) only for voxel rows of 1024 and more, where it is the faster one; the parity tests run on
    small grids, so every context they create through the Python mirror asks for it on any size ("mcskip" 2) -- the
    tests that compare it with the dense pass switch between 0 and 2 themselves."""
    if "gpu" not in request.keywords:
        yield
        return
    from vacancy_amd import carver as vc
    orig = vc.VoxelCarver.Init

    def init(self):
        ok = orig(self)
        if ok:
            self.set_param("mcskip", 2)
        return ok

    vc.VoxelCarver.Init = init
    try:
        yield
    finally:
        vc.VoxelCarver.Init = orig
