"""Numpy model of the device pipeline in libbsc_amd/csrc/device/bwt.hip and st.hip.

Each function mirrors ONE kernel (same inputs, same outputs, same index arithmetic) with vectorised
numpy, so the algorithm (tail-suffix ordering, head/rank semantics, doubling keys, emit indexing,
aux indexes, ST key layout) can be checked against the reference on a machine without a GPU.
It is test infrastructure; nothing in the product imports it.
"""
import numpy as np


def bwt_pack(T, n):
    tc = min(n, 7)
    Tp = np.concatenate([T, np.zeros(16, np.uint8)]).astype(np.uint64)
    i = np.arange(n, dtype=np.int64)
    key = np.zeros(n, np.uint64)
    for b in range(8):
        key |= Tp[i + b] << np.uint64(8 * (7 - b))
    tail = i + 8 > n
    slot = np.where(tail, n - 1 - i, i + tc)
    keys = np.empty(n, np.uint64); vals = np.empty(n, np.uint32)
    keys[slot] = key; vals[slot] = i
    return keys, vals


def radix_sort(keys, vals, passes):
    """stable LSD passes [(shift, bits)]"""
    for shift, bits in passes:
        d = (keys >> np.uint64(shift)) & np.uint64((1 << bits) - 1)
        o = np.argsort(d, kind="stable")
        keys = keys[o]
        if vals is not None:
            vals = vals[o]
    return keys, vals


def seg(keys, sa, cpos_in, m, n, initial):
    """returns flags-derived (rank per element, unsorted mask)"""
    head = np.ones(m + 1, bool)
    if m > 1:
        head[1:m] = keys[1:] != keys[:-1]
        if initial:
            tail_lo = n - 7 if n >= 8 else 0
            t = sa >= tail_lo
            head[1:m] |= t[1:] | t[:-1]
    uns = ~(head[:m] & head[1:m + 1])
    pos = np.arange(m, dtype=np.int64) if initial else cpos_in.astype(np.int64)
    hp = np.where(head[:m], pos + 1, 0)
    rank = np.maximum.accumulate(hp) - 1
    return pos, rank.astype(np.uint32), uns


def seg_long_counts(pos, rank, uns, limit=1024):
    """What seg_apply counts for free (bwt.hip, DS_NLONG / DS_EXCESS): an unsorted record knows its SA slot (pos) and the slot of its
    group's head (rank), and members of a group are contiguous in SA — the record `limit` places behind its head proves a group of more
    than `limit` records (one such record per long group), every record at or beyond that distance counts as excess."""
    d = pos[uns].astype(np.int64) - rank[uns].astype(np.int64)
    return int(np.count_nonzero(d == limit)), int(np.count_nonzero(d >= limit))


def bit_length(x):
    return int(x).bit_length()


def bwt_model(T, r=None):
    """-> (L, primary (1-based), I array like libsais_bwt_aux), rounds"""
    T = np.asarray(T, np.uint8)
    n = T.size
    keys, vals = bwt_pack(T, n)
    keys, vals = radix_sort(keys, vals, [(8 * p, 8) for p in range(8)])
    SA = np.zeros(n, np.uint32); ISA = np.zeros(n, np.uint32)
    pos, rank, uns = seg(keys, vals, None, n, n, True)
    SA[pos] = vals; ISA[vals] = rank
    cpos, csa, cgrp = pos[uns].astype(np.uint32), vals[uns], rank[uns]
    h = 8; rounds = 0
    lo_bits, hi_bits = bit_length(n), bit_length(n - 1)
    while cpos.size:
        rounds += 1
        assert rounds <= 40
        p = csa.astype(np.int64) + h
        nxt = np.where(p < n, ISA[np.minimum(p, n - 1)].astype(np.uint64) + 1, 0).astype(np.uint64)
        k = (cgrp.astype(np.uint64) << np.uint64(32)) | nxt
        passes = [(s, min(8, lo_bits - s)) for s in range(0, lo_bits, 8)] + [(32 + s, min(8, hi_bits - s)) for s in range(0, hi_bits, 8)]
        ks, vs = radix_sort(k, csa, passes)
        pos, rank, uns = seg(ks, vs, cpos, cpos.size, n, False)
        SA[pos] = vs; ISA[vs] = rank
        cpos, csa, cgrp = pos[uns].astype(np.uint32), vs[uns], rank[uns]
        h *= 2
    # emit
    pidx = int(ISA[0])
    L = np.empty(n, np.uint8)
    o = np.arange(n, dtype=np.int64)
    j = np.where(o <= pidx, o - 1, o)
    j[0] = 0
    src = SA[j].astype(np.int64) - 1
    L[:] = T[np.maximum(src, 0)]
    L[0] = T[n - 1]
    I = None
    if r:
        cnt = (n - 1) // r + 1
        I = ISA[np.arange(cnt, dtype=np.int64) * r].astype(np.int64) + 1
    return L, pidx + 1, I, rounds, SA


def st_model(T, k):
    T = np.asarray(T, np.uint8)
    n = T.size
    i = np.arange(n, dtype=np.int64)
    Tc = T.astype(np.uint64)
    if k < 8:
        key = Tc[(i - 1) % n] << np.uint64(56)
        for b in range(7):
            key |= Tc[(i + b) % n] << np.uint64(8 * (6 - b))
        ks, _ = radix_sort(key, None, [((7 - k) * 8 + 8 * p, 8) for p in range(k)])
        out = (ks >> np.uint64(56)).astype(np.uint8)
        index = int(np.flatnonzero(ks == key[0])[0])
    else:
        key = np.zeros(n, np.uint64)
        for b in range(8):
            key |= Tc[(i + b) % n] << np.uint64(8 * (7 - b))
        val = T[(i - 1) % n].astype(np.uint32) | np.where(i == 0, 0x100, 0).astype(np.uint32)
        ks, vs = radix_sort(key, val, [(8 * p, 8) for p in range(8)])
        out = (vs & 0xff).astype(np.uint8)
        index = int(np.flatnonzero(vs & 0x100)[0])
    return out, index
