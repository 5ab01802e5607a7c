"""One-rank smoke of the RCCL branch of the multi-GPU concatenation on the GPU box (round 6: the pool has 1-GPU boxes and NCCL refuses two
ranks on one device, so this is as much of `bench.py --gpus N`'s exchange step as can run here: process-group set-up on the "nccl" backend,
the all_gather of the block sizes and the max-reduction of the ranks' times on device tensors, the staging copy on rank 0).  The N > 1
logic — ragged sizes, posted receives, block order — is covered by tests/test_multigpu_gloo.py with two ranks on gloo."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_world1_gather_and_reduce():
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but torch.cuda.is_available() is False")
    from libbsc_amd.multigpu import gather_blocks_to_rank0, Concatenator
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        payload = np.random.default_rng(3).integers(0, 256, 15_277_890, dtype=np.uint8)       # the size of a compressed bench block
        got = gather_blocks_to_rank0(payload, 0, 1, dev, always_collective=True)
        assert got is not None and len(got) == 1 and np.array_equal(got[0], payload)
        t = torch.tensor([1.25], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.25
        dist.barrier()
        cat = Concatenator(0, 1, dev, keep=True)
        cat.put(payload[:1000]); cat.put(payload[:2000])
        cat.close()
        assert [int(b[0].size) for b in cat.blocks] == [1000, 2000]
    finally:
        dist.destroy_process_group()
