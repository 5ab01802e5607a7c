"""`.bsc` file driver: independent blocks in the reference CLI's container (bsc.cpp:46-59, 163-178,
397-418): "bsc1", int32 nBlocks, then per block {int64 blockOffset, int8 recordSize = 1, int8 sortingContexts = 1}
followed by the bsc_compress block.  Files written here unpack with the reference `bsc d` and the other way round;
with the same flags the two files are byte-identical (LZP on by default like the reference, -H15 -M128; -p turns it
off).  Segmentation / record reordering (`-s`, `-r`, off by default in the reference too) are not offered.

    python -m libbsc_amd.cli e <in> <out> [-b<MiB>] [-m0|-m3..8] [-e0|-e1|-e2] [-H<bits>] [-M<len>] [-p]
    python -m libbsc_amd.cli d <in> <out>
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 -m libbsc_amd.cli e <in> <out> ...

Block-parallel across GPUs: block b goes to rank b % world; rank 0 collects over RCCL and writes the file.
"""
import os
import struct
import sys

import numpy as np


def parse_container(buf):
    if buf[:4] != b"bsc1":
        raise ValueError("not a bsc1 file")
    (nblocks,) = struct.unpack_from("<i", buf, 4)
    pos = 8
    blocks = []
    from . import api
    for _ in range(nblocks):
        off, rec, ctx = struct.unpack_from("<qbb", buf, pos)
        pos += 10
        rc, bsize, dsize = api.bsc_block_info(buf[pos:pos + 28])
        if rc != 0:
            raise ValueError(f"bad block header ({rc})")
        blocks.append((off, rec, ctx, buf[pos:pos + bsize], dsize))
        pos += bsize
    return blocks


def decompress_file(in_path, out_path):
    from . import api
    buf = open(in_path, "rb").read()
    blocks = parse_container(buf)
    total = max((off + dsize for off, _, _, _, dsize in blocks), default=0)
    out = bytearray(total)
    for off, rec, ctx, blk, dsize in blocks:
        if rec != 1 or ctx != 1:
            raise ValueError("record reordering / preceding contexts are CLI filters outside the hot-path scope")
        data = api.bsc_decompress(blk)
        if isinstance(data, int):
            raise ValueError(f"bsc_decompress failed: {data}")
        out[off:off + dsize] = data
    with open(out_path, "wb") as f:
        f.write(out)
    return total


def compress_file(in_path, out_path, block_size=64 << 20, sorter=1, coder=1, depth=3, lzp_hash=0, lzp_min=0):
    import torch
    from . import GpuContext
    from .multigpu import assign_blocks, bsc_file_image, gather_blocks_to_rank0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if not dist.is_initialized():
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    data = np.fromfile(in_path, dtype=np.uint8)
    n = data.size
    nblocks = (n + block_size - 1) // block_size if n else 0
    mine = assign_blocks(nblocks, world)[rank]
    done = {}
    if lzp_hash or lzp_min:
        # LZP is host work in front of the GPU stage: compress the blocks from a few concurrent callers of the drop-in
        # bsc_compress (what the reference CLI's OpenMP team does, bsc.cpp:197) — one caller's LZP and entropy coding overlap
        # another's GPU stage
        from concurrent.futures import ThreadPoolExecutor
        from . import api
        os.environ.setdefault("BSC_GPU_DEVICE", str(local))

        def one(b):
            lo = b * block_size
            blk = api.bsc_compress(data[lo:lo + block_size], sorter, coder, lzp_hash=lzp_hash, lzp_min=lzp_min, features=3)
            if isinstance(blk, int):
                raise RuntimeError(f"bsc_compress failed on block {b}: {blk}")
            return b, np.frombuffer(blk, np.uint8)

        with ThreadPoolExecutor(max_workers=4) as pool:
            for b, blk in pool.map(one, mine):
                done[b] = blk
        ctx = None
    else:
        ctx = GpuContext(local, max_n=min(block_size, max(n, 1)) + 4096)
        pipe = ctx.pipe(depth)
        inflight = []
        for b in mine:
            lo = b * block_size
            d = torch.from_numpy(data[lo:lo + block_size]).to(dev)
            inflight.append((b, pipe.submit(d, d.numel(), sorter, coder, 3)))
            if len(inflight) >= depth:
                bb, t = inflight.pop(0)
                done[bb] = pipe.wait(t)
        for bb, t in inflight:
            done[bb] = pipe.wait(t)
        pipe.close()
    # rounds of one block per rank: gather to rank 0 in block order
    ordered = []
    rounds = (nblocks + world - 1) // world
    for r in range(rounds):
        b = r * world + rank
        payload = done.get(b, np.zeros(0, np.uint8))
        got = gather_blocks_to_rank0(np.ascontiguousarray(payload), rank, world, dev)
        if rank == 0:
            ordered += [bytes(g) for g in got if len(g)]
    if ctx is not None:
        ctx.close()
    if rank == 0:
        img = bsc_file_image(ordered, [b * block_size for b in range(nblocks)])
        with open(out_path, "wb") as f:
            f.write(img)
        return len(img)
    return 0


def main(argv):
    if len(argv) < 4 or argv[1] not in ("e", "d"):
        print(__doc__)
        return 2
    if argv[1] == "d":
        print(decompress_file(argv[2], argv[3]), "bytes")
        return 0
    block, sorter, coder = 64 << 20, 1, 1
    lzp, lzp_hash, lzp_min = True, 15, 128           # the reference CLI's defaults (bsc.cpp:73-78): LZP on, -H15 -M128
    for a in argv[4:]:
        if a.startswith("-b"): block = int(a[2:]) << 20
        elif a.startswith("-m"): sorter = 1 if a[2:] == "0" else int(a[2:])
        elif a.startswith("-e"): coder = {0: 3, 1: 1, 2: 2}[int(a[2:])]
        elif a.startswith("-H"): lzp_hash = int(a[2:])
        elif a.startswith("-M"): lzp_min = int(a[2:])
        elif a == "-p": lzp = False
        elif a == "-l": lzp = True
    size = compress_file(argv[2], argv[3], block, sorter, coder, lzp_hash=lzp_hash if lzp else 0, lzp_min=lzp_min if lzp else 0)
    if int(os.environ.get("RANK", "0")) == 0:
        print(size, "bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
