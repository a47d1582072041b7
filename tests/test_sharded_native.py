"""The row-sharded control flow of the NATIVE library with two processes on ONE GPU: the all-reduce is supplied by
the launcher (edrgp_comm_init_host over a gloo group), everything else -- chunked and single-chunk ranks side by
side, the automatic choice of the stable form on an ill-conditioned Kmm, the sync-free and the careful driver, the
packed lower triangle of [Phi | b], the gradient partials -- is csrc/sparse_eval.hip + csrc/comm.hip as they run
over RCCL.  Compared with the whole problem on one context and with the oracle (tests/sharded_check.py)."""
import pytest

from sharded_check import run_and_check

pytestmark = pytest.mark.gpu


def test_two_processes_one_gpu_host_allreduce(tmp_path):
    run_and_check(tmp_path, world=2, transport="gloo")


def test_three_processes_one_gpu_host_allreduce(tmp_path):
    run_and_check(tmp_path, world=3, transport="gloo")
