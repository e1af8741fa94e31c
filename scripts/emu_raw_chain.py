"""CPU emulation of k_raw_starts / k_raw_count / k_raw_scan (kernels/bam_raw.hip: the algorithm, not the code) over every cut of a
packed BAM: the golden rand.bam cut into blocks of <block> bytes, a first call over blocks [first, first + cut), a second over the
rest with the first call's unfinished record in front.   python3 scripts/emu_raw_chain.py <block> <lowest cut> <highest cut>"""
import struct, sys, gzip
NONE = 0xffffffff

def record_at(d, p, end):
    room = end - p
    if room < 36: return 2, 0
    bs, tid, pos = struct.unpack_from('<Iii', d, p)
    if bs < 32 or bs > (1 << 28): return 0, 0
    l_name = d[p + 12]; n_cigar = struct.unpack_from('<H', d, p + 16)[0]; l_seq = struct.unpack_from('<I', d, p + 20)[0]
    mtid, mpos = struct.unpack_from('<ii', d, p + 24)
    if tid < -1 or pos < -1 or mtid < -1 or mpos < -1 or l_name == 0 or l_seq > 0x7fffffff: return 0, 0
    if 32 + l_name + 4 * n_cigar + ((l_seq + 1) >> 1) + l_seq > bs: return 0, 0
    if room < 36 + l_name: return 2, 0
    if d[p + 35 + l_name] != 0: return 0, 0
    return 1, 4 + bs

def likely_record_at(d, p, end):
    r, step = record_at(d, p, end)
    if r != 1: return r, step
    if step > (1 << 24): return 0, 0
    l_name = d[p + 12]
    for k in range(l_name - 1):
        if not 33 <= d[p + 36 + k] <= 126: return 0, 0
    return 1, step

def index(d, blocks, first_abs):
    """d: stream bytes; blocks: list of (out_off, out_len); -> flags, n, tail_bytes, rec offsets"""
    L = blocks[-1][0] + blocks[-1][1]
    nb = len(blocks)
    starts = [NONE] * nb
    for b, (o, n) in enumerate(blocks):
        if b == 0:
            starts[0] = 0; continue
        weak = NONE
        for s0 in range(n):
            at = o + s0; ok = True; whole = True
            for hop in range(4):
                r, step = likely_record_at(d, at, L) if at < L else (2, 0)
                if r == 0 or (r == 2 and hop == 0): ok = False
                if r != 1:
                    whole = False; break
                at += step
            if ok and whole:
                starts[b] = s0; break
            if ok and weak == NONE: weak = s0
        if starts[b] == NONE: starts[b] = weak
    flags = 0; tail = None
    counts = [0] * nb; exits = [0] * nb; recs = [[] for _ in range(nb)]; broken = [False] * nb; tails = [None] * nb

    def walk(b, at):
        o, n = blocks[b]; end = o + n
        r_ = []; t = None; br = False
        while at < end:
            r, step = record_at(d, at, L)
            if r == 0: br = True; break
            if r == 2 or at + step > L: t = at; break
            r_.append(at); at += step
        return r_, at, t, br
    for b, (o, n) in enumerate(blocks):
        end = o + n
        none = b != 0 and starts[b] == NONE
        at = first_abs if b == 0 else end if none else o + starts[b]
        recs[b], exits[b], tails[b], broken[b] = walk(b, at)
    frm = first_abs; bad = False; fixed = 0
    for b, (o, n) in enumerate(blocks):
        end = o + n
        own = first_abs if b == 0 else None if starts[b] == NONE else o + starts[b]
        if frm >= end:
            recs[b] = []
            continue
        if own != frm:
            fixed += 1
            recs[b], exits[b], tails[b], broken[b] = walk(b, frm)
        if broken[b]: bad = True
        if tails[b] is not None:
            tail = tails[b]; frm = L
        else:
            frm = exits[b]
        if frm > end and b + 1 < nb: flags |= 4
    if bad or frm != L: flags |= 1
    out = [r for b in range(nb) for r in recs[b]]
    index.fixed = fixed
    return flags, len(out), (L - tail) if tail is not None else 0, out

if __name__ == '__main__':
    data = gzip.open(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..', 'tests', 'golden', 'bam', 'rand.bam')).read()
    l_text = struct.unpack_from('<i', data, 4)[0]; p = 8 + l_text
    n_ref = struct.unpack_from('<i', data, p)[0]; p += 4
    for _ in range(n_ref): p += 8 + struct.unpack_from('<i', data, p)[0]
    hl = p
    rec = [hl]
    while rec[-1] < len(data): rec.append(rec[-1] + 4 + struct.unpack_from('<i', data, rec[-1])[0])
    print(len(rec) - 1, 'records', len(data), 'bytes')
    block = int(sys.argv[1]); lo_cut = int(sys.argv[2]); hi_cut = int(sys.argv[3])
    first = hl // block
    nblk = -(-len(data) // block)
    fails = 0
    for cut in range(lo_cut, hi_cut):
        mid = first + cut
        if mid >= nblk: break
        a0 = first * block; a1 = min(mid * block, len(data))
        blocks = [(i * block - a0, min(block, len(data) - i * block)) for i in range(first, mid)]
        f, n, tail, offs = index(data[a0:a1], blocks, hl - a0)
        n_a = sum(1 for k in range(len(rec) - 1) if rec[k + 1] <= a1)
        want_tail = a1 - rec[n_a]
        ok = (f & 3) == 0 and n == n_a and tail == want_tail and offs == [r - a0 for r in rec[:n_a]]
        if ok and tail:   # second call with the carry
            front = data[a1 - tail:a1]
            blocks2 = [(tail + i * block - a1, min(block, len(data) - i * block)) for i in range(mid, nblk)]
            f2, n2, t2, offs2 = index(front + data[a1:], blocks2, 0)
            ok = (f2 & 3) == 0 and n2 == len(rec) - 1 - n_a and t2 == 0
        if not ok:
            fails += 1
            print('cut', cut, 'FAIL', f, n, n_a, tail, want_tail)
    print('block', block, 'fails', fails, 'last call fixed', index.fixed)
