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


CHILD_INFO = r'''
import os, sys, json
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
from avoid_mpc_amd import capi
from avoid_mpc_amd.host import Shard
torch.cuda.set_device(0)
info = Shard.rccl_info()
mapped = sorted({l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l})
sh = Shard(0, 1, Shard.unique_id())
a = torch.arange(32, dtype=torch.float64, device="cuda"); b = torch.zeros_like(a)
sh.gather(a, b)
rc = sh.wait(timeout_s=60.0)
mapped_after = sorted({l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l})
print("INFO " + json.dumps(dict(info=info, mapped=mapped, mapped_after=mapped_after, wait=rc, equal=bool(torch.equal(a, b)),
                                 bad_wait=sh.lib.amk_shard_wait(sh.h, None, -1.0))))
'''


def test_the_library_binds_the_rccl_that_is_already_in_the_process():
    """bench.py's ranks hold a torch.distributed "nccl" group (PyTorch's bundled librccl) AND amk_shard's own
    communicator.  The library must take the copy that is already mapped (RTLD_NOLOAD) instead of loading a second one, say
    which file it bound (dladdr of ncclAllGather), and its watchdog must return AMK_OK for a healthy gather."""
    import json
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + CHILD_INFO], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("INFO ")][0][5:])
    assert d["equal"] and d["wait"] == 0 and d["bad_wait"] == 1, d        # AMK_OK; a non-positive timeout is AMK_ERR_INVALID_ARG
    info = d["info"]
    assert info and os.path.basename(info["path"]).startswith("librccl"), d
    real = lambda xs: {os.path.realpath(x) for x in xs}
    if d["mapped"]:   # import torch had mapped its RCCL before the library asked: that one must be the one bound, and no other appeared
        assert info["loaded_before_amk"] and os.path.realpath(info["path"]) in real(d["mapped"]), d
        assert real(d["mapped_after"]) == real(d["mapped"]), d
    else:
        assert len(real(d["mapped_after"])) == 1, d
    print("bound RCCL:", info)
