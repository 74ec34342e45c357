"""ndgpu_write_records (the output loop of lib/nextcorrect.py:236-260 in the library) on fabricated records, without a GPU: the bytes
Python's own formatting gives -- header with '%f' of the float32 identity, bases, index lines with the offset of the bases -- for
accepted records, rejected ones, out-of-memory seeds (len 3: no index line) and identities at the rounding edges; -1 on a descriptor
that cannot be written to."""
import ctypes as C
import os

import numpy as np


def test_records_as_the_reference_loop_prints_them(tmp_path):
    from nextdenovo_amd import api
    lib = api.load()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    rng = np.random.default_rng(2)
    idents = [0.8, 0.79999995, 0.99999994, 1.0, 0.9876543, 0.5, 0.8000001, 0.999, 0.0, 0.85]
    lens = [600, 700, 500, 499, 12000, 900, 501, 3, 2, 4]
    n = len(lens)
    names = np.asarray([7, 1234567, 42, 9, 4000000000, 5, 6, 77, 88, 99], dtype=np.uint32)

    def make():
        recs = (C.POINTER(api.ConsensusTrimed) * n)()
        seqs = []
        for i in range(n):
            seq = bytes(rng.choice(np.frombuffer(b"ACGTacgt", dtype=np.uint8), lens[i]))
            seqs.append(seq)
            p = libc.malloc(lens[i] + 1)
            C.memmove(p, seq + b"\0", lens[i] + 1)
            c = C.cast(libc.malloc(C.sizeof(api.ConsensusTrimed)), C.POINTER(api.ConsensusTrimed))
            c.contents.len, c.contents.identity = lens[i], idents[i]
            C.cast(C.byref(c.contents, api.ConsensusTrimed.seq.offset), C.POINTER(C.c_void_p))[0] = p
            recs[i] = c
        return recs, seqs
    recs, seqs = make()
    order = np.asarray([3, 0, 9, 1, 4, 2, 8, 5, 7, 6], dtype=np.uint32)
    fa, ix = tmp_path / "o.fa", tmp_path / "o.idx"
    head = b">0 5 1.000000\nACGTA\n"
    with open(fa, "wb") as OUT, open(ix, "wb") as IDX:
        OUT.write(head)
        OUT.flush()
        pos = C.c_uint64(len(head))
        out_len, out_ide = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.float32)
        rc = lib.ndgpu_write_records(recs, order.ctypes.data_as(C.POINTER(C.c_uint32)), n, names.ctypes.data, 500, 0.8, OUT.fileno(), IDX.fileno(),
                                     C.byref(pos), out_len.ctypes.data, out_ide.ctypes.data)
    assert rc == 0 and all(not recs[i] for i in range(n))          # handed over: freed, slots cleared
    want, want_idx, at = head, b"", len(head)
    for i in order.tolist():
        ide = float(np.float32(idents[i]))
        if lens[i] >= 500 and lens[i] > 4 and ide >= 0.8:
            h = b">%d %d %f\n" % (int(names[i]), lens[i], ide)
            want += h + seqs[i] + b"\n"
            at += len(h) + lens[i] + 1
            want_idx += b"%d\t%d\t%d\n" % (int(names[i]), at - lens[i] - 1, lens[i])
        elif lens[i] != 3:
            want_idx += b"%d\t0\t0\n" % int(names[i])
    assert open(fa, "rb").read() == want and open(ix, "rb").read() == want_idx and pos.value == at == os.path.getsize(fa)
    assert out_len.tolist() == lens and np.array_equal(out_ide, np.asarray(idents, dtype=np.float32))
    assert want.count(b">") == 1 + 4                                 # (0.8 and 0.8000001 pass, 0.79999995 does not; 499 is too short)
    # a descriptor that cannot be written to
    recs, _ = make()
    rd = os.open(fa, os.O_RDONLY)
    try:
        pos = C.c_uint64(0)
        assert lib.ndgpu_write_records(recs, order.ctypes.data_as(C.POINTER(C.c_uint32)), n, names.ctypes.data, 500, 0.8, rd, -1, C.byref(pos), None, None) == -1
        assert pos.value == 0 and all(not recs[i] for i in range(n))
    finally:
        os.close(rd)
