"""Block-parallel multi-GPU driver pieces (one process per GPU, torch.distributed; "nccl" = RCCL on ROCm).

Blocks are independent (own header, checksums and model state), so the path shards with NO data-path
collective: block b belongs to rank b % world (the reference's CLI hands blocks to OpenMP threads the same way,
bsc.cpp:197-423).  The only exchange is the final concatenation of the variable-size compressed blocks on
rank 0: an all_gather of the sizes, then point-to-point send/recv of the payloads over xGMI.
The functions take the process group's device so the same code runs on gloo/CPU in the unit tests.
"""
from typing import List, Optional

import numpy as np


def assign_blocks(num_blocks: int, world: int) -> List[List[int]]:
    """block ids owned by each rank (round-robin, like a static OpenMP schedule over blocks)."""
    return [list(range(r, num_blocks, world)) for r in range(world)]


def gather_blocks_to_rank0(payload: np.ndarray, rank: int, world: int, device, staging=None) -> Optional[List[np.ndarray]]:
    """Every rank contributes one compressed block (np.uint8).  Rank 0 returns them (np.uint8 arrays) in rank order
    (= block order for one block per rank), other ranks return None."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return [payload]
    n = int(payload.size)
    size_t = torch.tensor([n], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, size_t)
    sizes = [int(s.item()) for s in sizes]
    if staging is not None and staging.numel() >= n:
        buf = staging[:n]
        buf.copy_(torch.from_numpy(payload))
    else:
        buf = torch.from_numpy(payload).to(device)
    if rank != 0:
        dist.send(buf, dst=0)
        return None
    parts = [payload]
    reqs = []
    recv = []
    for r in range(1, world):
        t = torch.empty(sizes[r], dtype=torch.uint8, device=device)
        reqs.append(dist.irecv(t, src=r))
        recv.append(t)
    for q in reqs:
        q.wait()
    return parts + [t.cpu().numpy() for t in recv]


def bsc_file_image(blocks, block_offsets: List[int]) -> bytes:
    """The reference CLI's container around independent blocks (bsc.cpp:46-59, 163-178, 397-418):
    'bsc1', int32 nBlocks, then per block {int64 offset, int8 recordSize=1, int8 sortingContexts=1} + block."""
    import struct
    out = bytearray(b"bsc1" + struct.pack("<i", len(blocks)))
    for blk, off in zip(blocks, block_offsets):
        out += struct.pack("<qbb", off, 1, 1) + bytes(blk)
    return bytes(out)
