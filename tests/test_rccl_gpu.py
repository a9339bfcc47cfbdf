"""GPU: the one exchange step of the multi-GPU sweep (all_gather of the controls) on RCCL itself.
Only one GPU is available to the tests, so the process group has world size 1 -- the collective still goes through
RCCL's all_gather_into_tensor on the user's HIP streams, which is what bench.py issues per in-flight step; the
world-size-2 logic is covered on gloo by tests/test_shard_gloo.py.  Runs in its own interpreter (a process group is
process-global state)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from tests import _shard_torch as shard
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % PORT, rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
S = 256
streams = [torch.cuda.Stream() for _ in range(4)]
us = [torch.randn((S, 4), dtype=torch.float64, device="cuda") for _ in streams]
outs = [torch.zeros((S, 4), dtype=torch.float64, device="cuda") for _ in streams]
torch.cuda.synchronize()
for rep in range(8):                      # the pattern of bench.py: one gather per in-flight step, each on its own stream
    for st, u, o in zip(streams, us, outs):
        with torch.cuda.stream(st):
            u.mul_(1.0 + 1e-3 * rep)
            r = shard.gather_controls(u, out=o)
            assert r is o
torch.cuda.synchronize()
for u, o in zip(us, outs):
    assert torch.equal(u, o)
# uneven-shard path (list all_gather) and the max-over-ranks reduction
r = shard.gather_controls(us[0], counts=[S])
assert torch.equal(r, us[0])
assert shard.max_over_ranks(1.25, torch.device("cuda", 0)) == 1.25
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_OK")
'''


def test_gather_controls_on_rccl_from_several_streams():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    code = "ROOT = %r\nPORT = %d\n" % (ROOT, port) + CHILD
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
