"""Multi-process GPU checks of the data-parallel path on the one-GPU box: two ranks share the card over gloo (the exchange
semantics are the backend's SUM all-reduce either way), and a one-rank RCCL group exercises RCCL together with HIP-graph replay and the
asynchronous bucket exchange.  The body of each check lives in tests/dist_worker.py; the checker is the CPU oracle.  What these cannot
show is RCCL timing or a multi-GPU RCCL ring: no scaling curve has been measured on hardware."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(cmd, env, mode):
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and f"DIST_WORKER_OK {mode}" in r.stdout, r.stdout[-6000:]


@pytest.mark.parametrize("mode", ["dp_step", "sync_bn", "pos_weight", "eps"])
def test_two_ranks_on_one_card_gloo(mode):
    env = dict(os.environ, CVAE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           WORKER, mode]
    _run(cmd, env, mode)


def test_one_rank_rccl_group_with_graph_replay_and_async_exchange():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    _run([sys.executable, WORKER, "nccl1"], env, "nccl1")


def test_one_rank_exchange_through_the_rccl_c_abi():
    """include/cvae_dp.h: cvae_dp_init / cvae_dp_allreduce_sum behind GradAllReducer(comm=RcclComm()) — eager step, whole-step graph and split-backward capture
    with the asynchronous exchange give the same training (one-rank communicator: bootstrap, handle life cycle and stream ordering, not wire traffic)."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    _run([sys.executable, WORKER, "dp_abi1"], env, "dp_abi1")
