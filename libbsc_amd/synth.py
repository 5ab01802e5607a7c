"""`synth-text v1` — the synthetic block generator every benchmark and parity test uses.

Integer-only and deterministic: xorshift64* PRNG, a 4096-word vocabulary of 2..9 lowercase letters,
words drawn with a product-of-uniforms skew, separated by ' ' (every 16th word by '\\n').
The word loop is sequential; this vectorises the PRNG stream in chunks with numpy so 64 MiB takes seconds.
"""
import numpy as np

_M64 = (1 << 64) - 1
_MUL = 2685821657736338717


class _XorShift64Star:
    def __init__(self, seed):
        assert seed != 0
        self.s = seed & _M64

    def next(self):
        s = self.s
        s ^= s >> 12
        s ^= (s << 25) & _M64
        s ^= s >> 27
        self.s = s
        return (s * _MUL) & _M64


def synth_text_v1(seed: int, n: int) -> np.ndarray:
    """n bytes of synthetic text as np.uint8."""
    g = _XorShift64Star(seed)
    vocab = []
    for _ in range(4096):
        ln = 2 + g.next() % 8
        vocab.append(np.array([97 + g.next() % 26 for _ in range(ln)], np.uint8))
    lens = np.array([v.size for v in vocab], np.int64)
    flat = np.concatenate(vocab)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])

    out = np.empty(n + 16, np.uint8)
    pos = 0
    words = 0
    CH = 1 << 16
    while pos < n:
        # draw a chunk of word ids sequentially (the PRNG is a serial recurrence)
        rs = np.empty(CH, np.uint64)
        s = g.s
        for i in range(CH):
            s ^= s >> 12
            s ^= (s << 25) & _M64
            s ^= s >> 27
            rs[i] = (s * _MUL) & _M64
        # how many of these words fit?
        idx = ((rs & np.uint64(0xfff)) * ((rs >> np.uint64(12)) & np.uint64(0xfff))) >> np.uint64(12)
        idx = idx.astype(np.int64)
        wl = lens[idx] + 1
        ends = pos + np.cumsum(wl)
        # words whose START is < n are emitted (the last one truncated)
        wstart = ends - wl
        k = int(np.searchsorted(wstart, n, side="left"))
        k = min(k, CH)
        if k == 0:
            break
        # advance the generator by exactly k draws
        if k == CH:
            g.s = s
        else:
            for _ in range(k):
                g.next()
        tot = int(ends[k - 1] - pos)
        buf = np.empty(tot, np.uint8)
        # gather word bytes
        wl_k = wl[:k]
        offs = np.concatenate([[0], np.cumsum(wl_k)[:-1]])
        src = np.repeat(starts[idx[:k]] - offs, wl_k) + np.arange(tot)
        sep_pos = offs + wl_k - 1
        src[sep_pos] = 0
        buf[:] = flat[src]
        wn = words + 1 + np.arange(k)
        buf[sep_pos] = np.where(wn % 16 == 0, 10, 32).astype(np.uint8)
        take = min(tot, n - pos)
        out[pos:pos + take] = buf[:take]
        pos += take
        words += k
    return out[:n].copy()


def _splitmix64(i: np.ndarray, seed: int) -> np.ndarray:
    """Counter-based 64-bit mix (splitmix64 finaliser) of the uint64 counters i; wraps mod 2^64."""
    with np.errstate(over="ignore"):
        z = i.astype(np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15) & _M64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synth_repeat_v1(seed: int, n: int, period: int, noise_every: int = 512) -> np.ndarray:
    """`synth-repeat v1`: a `period`-byte slice of synth-text v1 tiled to n bytes, then about n / noise_every bytes
    overwritten at hashed positions, every other one with 0xF2 (the LZP escape byte).  Long repeats with sparse
    damage: the kind of input the LZP preprocessor exists for.  Deterministic, integer-only."""
    base = synth_text_v1(seed, period)
    out = np.tile(base, n // period + 1)[:n].copy()
    k = n // noise_every
    if k:
        h = _splitmix64(np.arange(k, dtype=np.uint64), seed)
        pos = (h % np.uint64(n)).astype(np.int64)
        val = ((h >> np.uint64(40)) & np.uint64(0xff)).astype(np.uint8)
        val[::2] = 0xF2
        out[pos] = val          # duplicate positions: numpy assigns in index order, the last write wins
    return out


IMAGE_CORPORA = {
    # classes built from what the container image holds (there is no network for real corpora); the GPU boxes run the same image
    "python-source": ["/usr/lib/python3*/**/*.py", "/usr/local/lib/python3*/**/*.py"],
    "binary": ["/opt/rocm/lib/*.so*", "/usr/lib/x86_64-linux-gnu/*.so*"],
}


def image_corpus(kind: str, n: int):
    """n bytes of the image's own files of one kind, concatenated in sorted path order ("python-source": the *.py of the Python
    installations — identifiers, English comments, many duplicated files, LCPs up to 2^18; "binary": shared objects — zero runs,
    tables, 256 symbols).  Repeated when the image holds less than n bytes.  None when it holds none."""
    import glob
    out = bytearray()
    for pat in IMAGE_CORPORA[kind]:
        for f in sorted(glob.glob(pat, recursive=True)):
            try:
                out += open(f, "rb").read()
            except OSError:
                continue
            if len(out) >= n:
                return np.frombuffer(bytes(out[:n]), dtype=np.uint8).copy()
    if not out:
        return None
    reps = (n + len(out) - 1) // len(out)
    return np.frombuffer((bytes(out) * reps)[:n], dtype=np.uint8).copy()
