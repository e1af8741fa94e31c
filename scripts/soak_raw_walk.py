#!/usr/bin/env python3
"""Soak of the raw-record route (k_raw_starts / _count / _scan / _index, then the depth kernels on the records in place) on
DAMAGED record streams: tests/golden/bam/rand.bam inflated, 1 - 3 random bytes changed (mostly in some record's fixed part:
block_size, refID, pos, l_read_name, n_cigar_op, l_seq ...), packed into BGZF blocks of a random size again.  Every stream
is walked on the host by the kernel's own rules (kernels/bam_raw.hip: head_check + the name's NUL = bam_read1's reads plus the
checks that keep the in-place kernels inside the record); the device must flag what that walk refuses, and where it flags
nothing its record count and tail must be the walk's.  Accepted streams then go through the depth kernels (no reference
behaviour for a record whose pos lies beyond its target -- the reference writes out of bounds there --: only that they return).

    python3 scripts/soak_raw_walk.py [N=400]  -> one JSON line"""
import json
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
import highperformancengs_amd as hp  # noqa: E402
import test_bam_raw_gpu as T  # noqa: E402
from conftest import golden_path  # noqa: E402
from highperformancengs_amd import bamio  # noqa: E402


def host_walk(d, pos):
    """-> (broken, n_records, tail_bytes) by the rules of head_check / walk_block_ahead"""
    n, L = 0, len(d)
    while pos < L:
        room = L - pos
        if room < 36:
            return False, n, room
        bs, tid, p, w3, w4, l_seq, mtid, mpos = struct.unpack_from("<IiiIIIii", d, pos)
        l_name, n_cigar = w3 & 255, w4 & 0xffff
        if bs < 32 or bs > (1 << 28):
            return True, n, 0
        if tid < -1 or p < -1 or mtid < -1 or mpos < -1 or l_name == 0 or l_seq > 0x7fffffff:
            return True, n, 0
        if 32 + l_name + 4 * n_cigar + ((l_seq + 1) >> 1) + l_seq > bs:
            return True, n, 0
        if room < 36 + l_name:
            return False, n, room
        if d[pos + 35 + l_name] != 0:
            return True, n, 0
        if pos + 4 + bs > L:
            return False, n, room
        n += 1
        pos += 4 + bs
    return False, n, 0


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    raw = open(golden_path("bam", "rand.bam"), "rb").read()
    soa = bamio.read_bam_records(golden_path("bam", "rand.bam"))
    data = b"".join(zlib.decompress(raw[a:a + n], -15) for a, n, _ in T._blocks(raw))
    hl = T._header_len(data)
    rec_at = [hl]
    while rec_at[-1] < len(data):
        rec_at.append(rec_at[-1] + 4 + struct.unpack_from("<i", data, rec_at[-1])[0])
    rec_at.pop()
    ctx = hp.Context(0)
    flagged = same = tails = wrongly_flagged = depth_runs = depth_refused = 0
    for i in range(N):
        rng = np.random.default_rng(424_000 + i)
        d = bytearray(data)
        for _ in range(int(rng.integers(1, 4)) if i % 10 else 0):       # (every tenth stream undamaged)
            at = rec_at[int(rng.integers(0, len(rec_at)))]
            bs = struct.unpack_from("<i", data, at)[0]
            off = int(rng.integers(0, 36)) if rng.random() < 0.7 else int(rng.integers(0, 4 + bs))
            d[at + off] = int(rng.integers(0, 256)) if rng.random() < 0.5 else d[at + off] ^ (1 << int(rng.integers(0, 8)))
        d = bytes(d)
        broken, n, tail = host_walk(d, hl)
        block = int(rng.choice([700, 1000, 4096, 20000, 65280]))
        d_raw, info, keep = T._to_device(ctx, T._bgzf_pack(d, block))
        assert info.flags & 2 == 0, "a well-formed BGZF block reported as damaged"
        if broken:
            assert info.flags & 1, (i, "the host walk refuses this stream, the device took it", n, info.n_records)
        if info.flags & 1:
            flagged += 1
            wrongly_flagged += not broken      # (allowed: a chain that lands on its feet but not on the found starts)
            continue
        assert (info.n_records, info.tail_bytes) == (n, tail), (i, info.n_records, n, info.tail_bytes, tail)
        same += 1
        tails += tail != 0
        for tid, (name, tlen) in enumerate(soa.refs):
            try:
                ctx.depth_target_raw(d_raw, tid, tlen, 100, 0x704)
                depth_runs += 1
            except hp.HpnError as e:               # a changed pos / CIGAR may leave the reference's domain: said so, not computed
                assert e.status == -4, e
                depth_refused += 1
        del d_raw, keep
    ctx.close()
    print(json.dumps({"streams": N, "flagged": flagged, "flagged_though_the_walk_arrives": wrongly_flagged, "equal_to_the_host_walk": same,
                      "of_those_with_an_unfinished_last_record": tails, "depth_calls_on_accepted_streams": depth_runs, "depth_calls_refused_as_outside_the_domain": depth_refused}))


if __name__ == "__main__":
    main()
