#!/usr/bin/env python
"""The threaded gzip reader (csrc/pinflate.cpp) against zlib's gzread through the same entry point of the product library
(ndgpu_gzin_open with 1 thread IS gzread): random members -- levels 0-9, every strategy, flushes in between; text, random, run-length
and periodic data -- whole, cut short, with a flipped bit or followed by garbage, in chunks of 1 KB to 200 KB and 2-6 threads.
Host code: no GPU needed.  usage: fuzz_inflate.py <seed> <cases>"""
import ctypes as C
import os
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    seed, cases = int(sys.argv[1]), int(sys.argv[2])
    from nextdenovo_amd import overlap
    lib = overlap.load()
    lib.ndgpu_gzin_open.restype = C.c_void_p
    lib.ndgpu_gzin_open.argtypes = [C.c_char_p, C.c_int]
    lib.ndgpu_gzin_read.restype = C.c_int64
    lib.ndgpu_gzin_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.ndgpu_gzin_close.argtypes = [C.c_void_p]

    def read_all(path, threads, piece):
        h = lib.ndgpu_gzin_open(path.encode(), threads)
        out, buf, err = bytearray(), C.create_string_buffer(piece), False
        while True:
            n = lib.ndgpu_gzin_read(h, buf, piece)
            if n < 0:
                err = True
                break
            if n == 0:
                break
            out += buf.raw[:n]
        lib.ndgpu_gzin_close(h)
        return bytes(out), err
    rng = np.random.default_rng(seed)

    def blob():
        kind, n = int(rng.integers(0, 5)), int(rng.integers(1, 400000))
        if kind == 0:
            return b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 200 + i % 900)), b"I" * (200 + i % 900))
                            for i in range(max(1, n // 1500)))
        if kind == 1:
            return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        if kind == 2:
            return bytes(rng.integers(0, 4, n, dtype=np.uint8) + 65)
        if kind == 3:
            u = rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8).tobytes()
            return (u * (n // len(u) + 1))[:n]
        return b"".join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 600)) for _ in range(n // 300 + 1))
    bad = 0
    path = os.path.join(tempfile.mkdtemp(prefix="ndgz"), "f.gz")
    for it in range(cases):
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
        with open(path, "wb") as f:
            f.write(data)
        os.environ["NDGPU_INFLATE_CHUNK"] = str(int(rng.choice([1024, 3000, 8192, 40000, 200000])))
        want, werr = read_all(path, 1, 1 << 20)
        th = int(rng.integers(2, 7))
        got, gerr = read_all(path, th, int(rng.choice([1 << 20, 4097, 65536])))
        if gerr != werr or (not werr and got != want):
            bad += 1
            print("MISMATCH case %d: %d bytes in, zlib %d bytes (error %s), threaded %d bytes (error %s), chunk %s, threads %d"
                  % (it, len(data), len(want), werr, len(got), gerr, os.environ["NDGPU_INFLATE_CHUNK"], th), flush=True)
    print("seed %d: %d cases, %d bad" % (seed, cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
