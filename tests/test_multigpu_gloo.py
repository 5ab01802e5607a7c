"""N > 1 path on CPU: world_size-2 gloo process group exercising the block assignment and the
variable-size concatenation used by bench.py --gpus N (RCCL on the GPU box)."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from libbsc_amd.multigpu import assign_blocks, bsc_file_image


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from libbsc_amd.multigpu import gather_blocks_to_rank0
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    payload = rng.integers(0, 256, 1000 + 777 * rank, dtype=np.uint8)      # ragged sizes
    got = gather_blocks_to_rank0(payload, rank, world, torch.device("cpu"))
    # the background concatenation stage used by bench.py: three rounds of ragged blocks
    from libbsc_amd.multigpu import Concatenator
    cat = Concatenator(rank, world, torch.device("cpu"), keep=True)
    for rnd in range(3):
        cat.put(np.random.default_rng(1000 * rnd + rank).integers(0, 256, 500 + 300 * rank + rnd, dtype=np.uint8))
    cat.close()
    if rank == 0:
        for rnd in range(3):
            for r in range(world):
                want = np.random.default_rng(1000 * rnd + r).integers(0, 256, 500 + 300 * r + rnd, dtype=np.uint8)
                assert np.array_equal(cat.blocks[rnd][r], want), (rnd, r)
    # timing reduction used by bench.py: max over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        import hashlib
        q.put(([hashlib.md5(bytes(b)).hexdigest() for b in got], [len(b) for b in got], float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    hashes, lens, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [np.random.default_rng(100 + r).integers(0, 256, 1000 + 777 * r, dtype=np.uint8).tobytes() for r in range(world)]
    assert lens == [len(w) for w in want]
    import hashlib
    assert hashes == [hashlib.md5(w).hexdigest() for w in want]
    assert tmax == 2.0


def test_assign_blocks_and_file_image():
    assert assign_blocks(8, 8) == [[i] for i in range(8)]
    assert assign_blocks(5, 2) == [[0, 2, 4], [1, 3]]
    assert assign_blocks(0, 4) == [[], [], [], []]
    img = bsc_file_image([b"AAAA", b"BB"], [0, 64 << 20])
    assert img[:4] == b"bsc1" and int.from_bytes(img[4:8], "little") == 2
    assert int.from_bytes(img[8:16], "little") == 0 and img[16:18] == b"\x01\x01" and img[18:22] == b"AAAA"
    assert int.from_bytes(img[22:30], "little") == 64 << 20
