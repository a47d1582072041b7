"""csrc/comm.hip over RCCL on real devices: the same worker as tests/test_sharded_native.py, one GPU per rank.
Skipped on a box with fewer devices (the build box has one GPU; the driver's 8-GPU node runs these)."""
import pytest

from sharded_check import run_and_check

pytestmark = pytest.mark.gpu


def _devices():
    from edrgp_amd import _lib
    return _lib.device_count()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_row_sharded_rccl(tmp_path, world):
    if _devices() < world:
        pytest.skip("needs %d GPUs, %d visible" % (world, _devices()))
    run_and_check(tmp_path, world=world, transport="rccl")
