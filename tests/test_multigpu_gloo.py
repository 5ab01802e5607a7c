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


# ---- real compressed blocks through the concatenation stage -----------------------------------------------------------
# What bench.py --gpus N and the file driver do with N ranks, minus the GPU: every rank compresses ITS blocks (block b belongs
# to rank b % world; here with the CPU oracle, ragged sizes), hands them to the Concatenator round by round, and rank 0's image
# of the gathered blocks must be the reference container of those blocks (bsc.cpp:163-178, 397-418) — byte for byte what one
# process would have written, and unpackable by the reference's own `bsc d`.
_BLOCK_SIZES = [20011, 31000, 8190, 26001, 17000]      # five blocks, two ranks: rank 0 gets 3, rank 1 gets 2 (padded with a skip round)


def _block_data(b):
    from libbsc_amd.synth import synth_text_v1
    return synth_text_v1(70 + b, _BLOCK_SIZES[b])


def _real_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from libbsc_amd.multigpu import Concatenator
    from oracle.refbind import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle()
    mine = assign_blocks(len(_BLOCK_SIZES), world)[rank]
    rounds = max(len(x) for x in assign_blocks(len(_BLOCK_SIZES), world))
    cat = Concatenator(rank, world, torch.device("cpu"), keep=True)
    for rnd in range(rounds):
        # a rank that has run out of blocks still takes part in the round (zero-length payload): all ranks put() equally often
        blk = np.frombuffer(orc.compress(_block_data(mine[rnd]), 1, 1), dtype=np.uint8) if rnd < len(mine) else np.zeros(0, np.uint8)
        cat.put(blk)
    cat.close()
    if rank == 0:
        blocks = []
        for rnd in range(rounds):
            for r in range(world):
                if len(cat.blocks[rnd][r]):
                    blocks.append(cat.blocks[rnd][r].tobytes())        # round-major, rank-minor = block order b = rnd * world + r
        q.put(blocks)
    dist.barrier()
    dist.destroy_process_group()


def test_compressed_blocks_through_concatenator_match_reference_container(tmp_path):
    import subprocess
    from oracle.refbind import Oracle
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_real_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    blocks = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    orc = Oracle()
    want_blocks = [orc.compress(_block_data(b), 1, 1) for b in range(len(_BLOCK_SIZES))]
    assert [len(b) for b in blocks] == [len(b) for b in want_blocks]
    assert blocks == want_blocks                                           # every rank's blocks arrived whole, in block order
    offsets = list(np.cumsum([0] + _BLOCK_SIZES[:-1]))
    image = bsc_file_image(blocks, [int(o) for o in offsets])
    assert image == bsc_file_image(want_blocks, [int(o) for o in offsets])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bsc = os.path.join(root, "oracle", "_ref", "bsc")
    if os.path.exists(bsc):                                                # the reference's own decoder takes the file
        f = tmp_path / "two_ranks.bsc"
        f.write_bytes(image)
        out = tmp_path / "back.bin"
        r = subprocess.run([bsc, "d", str(f), str(out)], capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS="2"))
        assert r.returncode == 0, r.stdout + r.stderr
        assert out.read_bytes() == b"".join(_block_data(b).tobytes() for b in range(len(_BLOCK_SIZES)))


def test_assign_blocks_and_file_image():
    assert assign_blocks(8, 8) == [[i] for i in range(8)]
    assert assign_blocks(5, 2) == [[0, 2, 4], [1, 3]]
    assert assign_blocks(0, 4) == [[], [], [], []]
    img = bsc_file_image([b"AAAA", b"BB"], [0, 64 << 20])
    assert img[:4] == b"bsc1" and int.from_bytes(img[4:8], "little") == 2
    assert int.from_bytes(img[8:16], "little") == 0 and img[16:18] == b"\x01\x01" and img[18:22] == b"AAAA"
    assert int.from_bytes(img[22:30], "little") == 64 << 20
