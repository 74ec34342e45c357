"""The gzip reader under the FASTA / FASTQ parser (csrc/pinflate.cpp: one gzip file inflated by several host threads) against zlib's
gzread -- what the reference's kseq reads through (lib/bseq.h:3) -- through the same entry point of the product library
(ndgpu_gzin_open with 1 thread IS gzread): the same bytes for whole files, concatenated members, every block type, flush points,
header fields, trailing garbage and truncated files; an error where zlib reports one (a bad code, a CRC-32 or length that does not
match).  Host code: runs without a GPU."""
import ctypes as C
import gzip
import os
import struct
import zlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib():
    from nextdenovo_amd import overlap
    lb = overlap.load()
    lb.ndgpu_gzin_open.restype = C.c_void_p
    lb.ndgpu_gzin_open.argtypes = [C.c_char_p, C.c_int]
    lb.ndgpu_gzin_read.restype = C.c_int64
    lb.ndgpu_gzin_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lb.ndgpu_gzin_stats.argtypes = [C.c_void_p, C.c_void_p]
    lb.ndgpu_gzin_close.argtypes = [C.c_void_p]
    return lb


def read_all(lib, path, threads, piece=1 << 20):
    h = lib.ndgpu_gzin_open(str(path).encode(), threads)
    assert h
    out, buf, err = bytearray(), C.create_string_buffer(piece), False
    while True:
        n = lib.ndgpu_gzin_read(h, buf, piece)
        if n < 0:
            err = True
            break
        if n == 0:
            break
        out += buf.raw[:n]
    st = (C.c_uint64 * 3)()
    threaded = lib.ndgpu_gzin_stats(h, st)
    lib.ndgpu_gzin_close(h)
    return bytes(out), err, threaded, list(st)


def fastq(n, rng):
    recs = []
    for i in range(n):
        ln = int(rng.integers(50, 3000))
        recs.append("@r%d\n%s\n+\n%s\n" % (i, "".join("ACGT"[x] for x in rng.integers(0, 4, ln)), "".join(chr(33 + x) for x in rng.integers(0, 40, ln))))
    return "".join(recs).encode()


def _cases():
    rng = np.random.default_rng(1)
    data = fastq(1500, rng)
    yield "level 1", gzip.compress(data, 1)
    yield "level 6", gzip.compress(data, 6)
    yield "level 9", gzip.compress(data, 9)
    yield "two members", gzip.compress(data[:1000000], 6) + gzip.compress(data[1000000:], 3)
    yield "many small members", b"".join(gzip.compress(data[i:i + 5000], 6) for i in range(0, 600000, 5000))
    yield "empty member first", gzip.compress(b"") + gzip.compress(data[:70000])
    yield "trailing garbage", gzip.compress(data[:300000]) + b"garbage here" * 10
    yield "stored blocks", gzip.compress(rng.integers(0, 256, 700000, dtype=np.uint8).tobytes(), 6)
    yield "zeros", gzip.compress(bytes(5000000), 9)
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)
    yield "fixed codes", co.compress(data[:400000]) + co.flush()
    co, b = zlib.compressobj(6, zlib.DEFLATED, 31), b""
    for i in range(0, 900000, 30000):
        b += co.compress(data[i:i + 30000]) + co.flush(zlib.Z_FULL_FLUSH if i % 60000 else zlib.Z_SYNC_FLUSH)
    yield "flush points", b + co.flush()
    yield "truncated", gzip.compress(data, 6)[:700001]
    yield "truncated in the trailer", gzip.compress(data[:200000], 6)[:-3]
    c = bytearray(gzip.compress(data, 6))
    c[500000] ^= 0x55
    yield "corrupt", bytes(c)
    c = bytearray(gzip.compress(data[:100000], 6))
    c[-5] ^= 1
    yield "bad crc", bytes(c)
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = raw.compress(data[:50000]) + raw.flush()
    yield "header fields", (b"\x1f\x8b\x08\x1c" + b"\0" * 6 + b"\x05\x00hello" + b"name.fq\0" + b"a comment\0" + body
                            + struct.pack("<II", zlib.crc32(data[:50000]), 50000))


@pytest.mark.parametrize("chunk", [20000, 65536, 2 << 20])
def test_same_bytes_as_gzread(lib, tmp_path, monkeypatch, chunk):
    monkeypatch.setenv("NDGPU_INFLATE_CHUNK", str(chunk))
    p = tmp_path / "x.gz"
    speculative = 0
    for name, blob in _cases():
        p.write_bytes(blob)
        want, werr, threaded, _ = read_all(lib, p, 1)
        assert not threaded                                     # one thread: zlib itself
        for th in (2, 5):
            got, gerr, threaded, st = read_all(lib, p, th, piece=(1 << 20) if th == 2 else 40961)
            assert threaded and gerr == werr, (name, th)
            if not werr:
                assert got == want, (name, th, len(got), len(want))
            else:
                # in front of the error gzread hands out what it decoded; so does the threaded reader (one of the two may see the error a
                # block earlier than the other: what both hand out is the same stream)
                k = min(len(got), len(want))
                assert got[:k] == want[:k], (name, th)
                assert len(got) * 2 >= len(want), (name, th, len(got), len(want))
            speculative += st[2] - st[0]                        # accepted chunks beyond every round's first
    if chunk < (2 << 20):
        assert speculative > 20                                  # (the guessed chunks do carry the decoding)
    p.write_bytes(b">r1\nACGT\n")
    assert read_all(lib, p, 4)[:3] == (b">r1\nACGT\n", False, 0)  # not gzip: gzread's transparent mode


def test_fuzzed_streams(lib, tmp_path, monkeypatch):
    """Random members (levels 0-9, every strategy, flushes in between, text / random / run-length / periodic data), cut short, with a
    flipped bit or followed by garbage, in small chunks (many guessed block starts per file)."""
    rng = np.random.default_rng(7)

    def blob():
        kind, n = int(rng.integers(0, 5)), int(rng.integers(1, 250000))
        if kind == 0:
            return fastq(max(1, n // 3000), rng)
        if kind == 1:
            return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        if kind == 2:
            return bytes(rng.integers(0, 4, n, dtype=np.uint8) + 65)
        if kind == 3:
            u = rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8).tobytes()
            return (u * (n // len(u) + 1))[:n]
        return b"".join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 600)) for _ in range(n // 300 + 1))

    p = tmp_path / "f.gz"
    for it in range(60):
        members = []
        for _ in range(int(rng.integers(1, 4))):
            co = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, 31, int(rng.integers(1, 10)),
                                  int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])))
            b = b""
            for _ in range(int(rng.integers(1, 6))):
                b += co.compress(blob())
                if rng.random() < 0.4:
                    b += co.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH])))
            members.append(b + co.flush())
        data = b"".join(members)
        r = rng.random()
        if r < 0.15:
            data = data[:int(rng.integers(1, len(data)))]
        elif r < 0.25:
            d = bytearray(data)
            d[int(rng.integers(10, len(d)))] ^= 1 << int(rng.integers(0, 8))
            data = bytes(d)
        elif r < 0.3:
            data += rng.integers(0, 256, 50, dtype=np.uint8).tobytes()
        p.write_bytes(data)
        monkeypatch.setenv("NDGPU_INFLATE_CHUNK", str(int(rng.choice([1024, 3000, 8192, 40000, 200000]))))
        want, werr, _, _ = read_all(lib, p, 1)
        got, gerr, _, _ = read_all(lib, p, int(rng.integers(2, 7)), piece=int(rng.choice([1 << 20, 4097, 65536])))
        assert gerr == werr, it
        if not werr:
            assert got == want, (it, len(got), len(want))


def test_parser_reads_through_the_threaded_reader(lib, tmp_path, monkeypatch):
    """ndgpu_fastx over a multi-megabyte FASTQ.gz: the records with 1 and with 4 inflate threads."""
    rng = np.random.default_rng(3)
    recs = []
    for i in range(1200):
        ln = int(rng.integers(500, 9000))
        recs.append("@%d\n%s\n+\n%s\n" % (i + 1, "".join("ACGT"[x] for x in rng.integers(0, 4, ln)), "I" * ln))
    p = tmp_path / "r.fq.gz"
    p.write_bytes(gzip.compress("".join(recs).encode(), 6))
    monkeypatch.setenv("NDGPU_INFLATE_CHUNK", "300000")
    from nextdenovo_amd import seq_dump
    sets = []
    for th in ("1", "4"):
        monkeypatch.setenv("NDGPU_INFLATE_THREADS", th)
        chunks = list(seq_dump.iter_chunks(str(p), names=True))
        sets.append((np.concatenate([c[0] for c in chunks]), np.concatenate([c[2] for c in chunks]), np.concatenate([c[3] for c in chunks])))
    a, b = sets
    assert a[1].size == 1200 and int(a[1].sum()) == a[0].size
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
