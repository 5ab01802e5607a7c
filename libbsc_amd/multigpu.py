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


def gather_blocks_to_rank0(payload: np.ndarray, rank: int, world: int, device, staging=None, always_collective: bool = False) -> Optional[List[np.ndarray]]:
    """Every rank contributes one compressed block (np.uint8).  Rank 0 returns them (np.uint8 arrays) in rank order
    (= block order for one block per rank), other ranks return None.
    One all_gather of the sizes, then the peers' payloads arrive concurrently (all receives are posted before the first
    wait; xGMI is point to point, so the seven links of rank 0 work in parallel) into one staging tensor that is copied to
    the host once."""
    import torch
    import torch.distributed as dist
    if world == 1 and not always_collective:        # (always_collective: the one-rank smoke test of the RCCL branch on a 1-GPU box)
        return [payload]
    n = int(payload.size)
    mine = torch.tensor([n], dtype=torch.int64, device=device)
    allsz = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allsz, mine)
    sizes = [int(x) for x in allsz.tolist()]
    if rank != 0:
        if staging is not None and staging.numel() >= n:
            buf = staging[:n]
            buf.copy_(torch.from_numpy(payload))
        else:
            buf = torch.from_numpy(payload).to(device)
        dist.send(buf, dst=0)
        return None
    total = sum(sizes[1:])
    if staging is not None and staging.numel() >= total:
        big = staging[:total]
    else:
        big = torch.empty(total, dtype=torch.uint8, device=device)
    reqs, off = [], 0
    for r in range(1, world):
        reqs.append(dist.irecv(big[off:off + sizes[r]], src=r))
        off += sizes[r]
    for q in reqs:
        q.wait()
    host = big.cpu().numpy()
    parts, off = [payload], 0
    for r in range(1, world):
        parts.append(host[off:off + sizes[r]])
        off += sizes[r]
    return parts


class Concatenator:
    """The final concatenation as a background stage: every rank hands over its finished blocks in order, a thread per rank
    runs gather_blocks_to_rank0 for them, so the collective and the D2H on rank 0 overlap the next blocks' GPU and host work
    instead of stalling the submitting thread.  All ranks must put() the same number of blocks."""

    def __init__(self, rank: int, world: int, device, staging=None, keep: bool = False):
        import queue
        import threading
        self.rank, self.world, self.device, self.staging, self.keep = rank, world, device, staging, keep
        self.q = queue.Queue()
        self.blocks: List[List[np.ndarray]] = []        # rank 0 with keep=True: the gathered blocks of every round
        self.bytes = 0
        self.err = None
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _run(self):
        if getattr(self.device, "type", "cpu") == "cuda":       # the current device is per thread
            import torch
            torch.cuda.set_device(self.device)
        while True:
            item = self.q.get()
            if item is None:
                return
            try:
                got = gather_blocks_to_rank0(item, self.rank, self.world, self.device, staging=self.staging)
                if got is not None:
                    self.bytes += sum(int(g.size) for g in got)
                    if self.keep:
                        self.blocks.append([np.array(g, copy=True) for g in got])
            except Exception as e:  # surfaced by close()
                self.err = e
                return

    def put(self, payload: np.ndarray):
        self.q.put(np.array(payload, copy=True))        # the caller may recycle its buffer

    def close(self):
        self.q.put(None)
        self.t.join()
        if self.err is not None:
            raise self.err


def bsc_file_image(blocks, block_offsets: List[int]) -> bytes:
    """The reference CLI's container around independent blocks (bsc.cpp:46-59, 163-178, 397-418):
    'bsc1', int32 nBlocks, then per block {int64 offset, int8 recordSize=1, int8 sortingContexts=1} + block."""
    import struct
    out = bytearray(b"bsc1" + struct.pack("<i", len(blocks)))
    for blk, off in zip(blocks, block_offsets):
        out += struct.pack("<qbb", off, 1, 1) + bytes(blk)
    return bytes(out)
