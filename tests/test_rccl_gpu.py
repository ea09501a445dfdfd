"""GPU: the RCCL code paths of bench.py / tools/train.py executed once on the one GPU a test box has (world_size = 1):
process-group initialisation with device_id, all_gather of the poses after a hipGraph replay with side streams, all_reduce of
the 85.8 MB flat gradient buffer.  Runs in a child process with a time limit (a collective that hangs must not take the
test session with it)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_world_size_one_next_to_graph_replay():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py")], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, out.stdout[-3000:]
