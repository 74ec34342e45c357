// pinflate.cpp -- one gzip stream inflated by several host threads (the front of `seq_dump`, util/seq_dump.c:36-118: the reference reads
// its FASTA / FASTQ.gz input through zlib's gzread, one thread per file).  Host code of the overlap library; a drop-in for the
// gzread calls of fastx_reader.cpp: the same bytes in the same order, members concatenated, trailing garbage ignored, a truncated
// file delivers what it holds.
//
// A deflate stream has no index: a block can start at any bit, and a match can copy from the 32 KB before it.  The compressed file is
// cut into chunks; the first chunk of a round starts where the previous round ended (bit position and window known), every other
// chunk
//   1. looks for a block start at or after its first byte: a non-final dynamic-Huffman header whose three codes are complete (a random
//      bit position passes that test about once in 10^8), and decodes from there;
//   2. decodes into 16-bit symbols: 0..255 a byte, 256 + i "the byte i of the unknown 32 KB window before my first byte" -- a match
//      that reaches back into the unknown copies those symbols on;
//   3. stops at the first block start at or after the next chunk's first byte.
// Then, in order: chunk j is accepted when chunk j - 1 stopped exactly where chunk j started (so chunk j decoded the true
// continuation, whatever the heuristic of step 1 thought); its symbols become bytes through the window chunk j - 1 left.  The first
// chunk that does not fit ends the round -- the next round starts at the exact position and window the accepted chunks end with,
// so a wrong guess costs time, never bytes.  Every member's CRC-32 and length are checked as gzread checks them.
#include "pinflate.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace ndovl {
namespace {

constexpr int kWin = 32768;
constexpr int kLitBits = 10, kDistBits = 9;
constexpr uint64_t kMaxScanBytes = 512u << 10;
constexpr uint64_t kMaxChunkOut = 512ull << 20;   // a chunk that inflates beyond this stops at its next block (the round ends there)

struct BitIn {
    const uint8_t *base, *p, *end;
    uint64_t buf = 0;
    int cnt = 0;       // valid bits in buf
    bool over = false; // asked for bits beyond the end of the input
    void seek(uint64_t bitpos) {
        p = base + (bitpos >> 3), buf = 0, cnt = 0, over = false;
        const int sk = (int)(bitpos & 7);
        if (sk) { need(sk); drop(sk); }
    }
    uint64_t bitpos() const { return (uint64_t)(p - base) * 8 - (uint64_t)cnt; }
    void refill() {
        if (p + 8 <= end) {
            uint64_t w;
            memcpy(&w, p, 8);
            buf |= w << cnt;
            p += (63 - cnt) >> 3;
            cnt |= 56;
        } else {
            while (cnt <= 56 && p < end) buf |= (uint64_t)*p++ << cnt, cnt += 8;
        }
    }
    // at least n (<= 32) bits in buf, zeros past the end of the input (`over` is set when they are consumed)
    void need(int n) { if (cnt < n) refill(); }
    uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    void drop(int n) {
        if (n > cnt) { over = true; buf = 0; cnt = 0; return; }
        buf >>= n, cnt -= n;
    }
    uint32_t take(int n) { need(n); const uint32_t v = peek(n); drop(n); return v; }
    void align() { drop(cnt & 7); }
};

// one canonical Huffman code: a table over the first `bits` bits, the canonical walk (by code length) for longer codes
struct Huff {
    uint16_t fast[1 << kLitBits];  // (symbol << 4) | length; 0 = longer than the table / unused
    uint16_t count[16], symbol[288];
    int bits = 0, max_len = 0;
    // 0 = complete; 1 = incomplete; -1 = over-subscribed.  n <= 288 code lengths (0 = unused)
    int build(const uint8_t *len, int n, int table_bits) {
        bits = table_bits;
        memset(count, 0, sizeof(count));
        for (int i = 0; i < n; ++i) count[len[i]]++;
        max_len = 15;
        while (max_len > 0 && !count[max_len]) --max_len;
        int left = 1;
        for (int l = 1; l <= 15; ++l) {
            left <<= 1;
            left -= count[l];
            if (left < 0) return -1;
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int i = 0; i < n; ++i)
            if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
        memset(fast, 0, sizeof(uint16_t) << bits);
        uint32_t code = 0;
        int idx = 0;
        for (int l = 1; l <= 15; ++l) {
            for (int k = 0; k < count[l]; ++k, ++code, ++idx) {
                if (l > bits) continue;
                uint32_t rev = 0;
                for (int b = 0; b < l; ++b) rev |= (code >> b & 1u) << (l - 1 - b);
                const uint16_t e = (uint16_t)(symbol[idx] << 4 | l);
                for (uint32_t j = rev; j < (1u << bits); j += 1u << l) fast[j] = e;
            }
            code <<= 1;
        }
        return left > 0 ? 1 : 0;
    }
    // the next symbol, or -1 (no such code)
    int decode(BitIn &in) const {
        in.need(15);
        const uint16_t e = fast[in.peek(bits)];
        if (e) { in.drop(e & 15); return e >> 4; }
        int code = 0, first = 0, index = 0;
        uint64_t b = in.buf;
        for (int l = 1; l <= max_len; ++l) {
            code |= (int)(b & 1);
            b >>= 1;
            const int c = count[l];
            if (code - c < first) { in.drop(l); return symbol[index + (code - first)]; }
            index += c, first += c;
            first <<= 1, code <<= 1;
        }
        return -1;
    }
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Codes { Huff lit, dist; bool dist_none = false; };

// The dynamic block header after the three type bits (RFC 1951 3.2.7), with zlib's verdicts (inflate.c, inftrees.c): an incomplete
// code is refused unless it is a single code of one bit; `strict` (looking for a block start) wants all three codes complete (the
// distance code may be that single code) -- what every block of a compressor's output has.
bool read_dynamic(BitIn &in, Codes &C, bool strict) {
    const int hlit = (int)in.take(5) + 257, hdist = (int)in.take(5) + 1, hclen = (int)in.take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; ++i) cl[kClOrder[i]] = (uint8_t)in.take(3);
    Huff h;
    const int r = h.build(cl, 19, 7);
    if (r != 0) return false;   // (zlib: an incomplete code-length code is an error)
    uint8_t len[320];
    int n = 0;
    while (n < hlit + hdist) {
        const int s = h.decode(in);
        if (s < 0 || in.over) return false;
        if (s < 16) { len[n++] = (uint8_t)s; continue; }
        int rep, v = 0;
        if (s == 16) {
            if (n == 0) return false;
            v = len[n - 1], rep = 3 + (int)in.take(2);
        } else if (s == 17) rep = 3 + (int)in.take(3);
        else rep = 11 + (int)in.take(7);
        if (n + rep > hlit + hdist) return false;
        while (rep--) len[n++] = (uint8_t)v;
    }
    if (in.over || len[256] == 0) return false;
    const int rl = C.lit.build(len, hlit, kLitBits);
    if (rl < 0 || (rl > 0 && (strict || C.lit.max_len != 1))) return false;
    const int rd = C.dist.build(len + hlit, hdist, kDistBits);
    C.dist_none = C.dist.max_len == 0;
    if (rd < 0 || (rd > 0 && C.dist.max_len > 1)) return false;
    if (strict && C.dist_none) return false;
    return true;
}

void fixed_codes(Codes &C) {
    uint8_t len[288];
    int i = 0;
    for (; i < 144; ++i) len[i] = 8;
    for (; i < 256; ++i) len[i] = 9;
    for (; i < 280; ++i) len[i] = 7;
    for (; i < 288; ++i) len[i] = 8;
    C.lit.build(len, 288, kLitBits);
    uint8_t d[30];
    for (i = 0; i < 30; ++i) d[i] = 5;
    C.dist.build(d, 30, kDistBits);
    C.dist_none = false;
}

struct Segment {          // a run of a chunk's output that belongs to one gzip member
    uint64_t begin, end;  // offsets in the chunk's output
    bool ends_member = false;
    uint32_t crc = 0, isize = 0;
};

enum Stop { kStopBoundary, kStopInputEnd, kStopNoMember, kStopError };

// What a chunk decoded.  T = uint8_t: window known, bytes; T = uint16_t: symbols (see the head of the file).
template <class T>
struct ChunkOut {
    std::vector<T> out;   // [kWin of window][output]
    uint64_t n = 0;       // output symbols
    std::vector<Segment> segs;
    uint64_t start_bit = 0, end_bit = 0;
    bool end_in_member = false;     // stopped at a block start inside a member (else: in front of a member header / at the end)
    Stop stop = kStopError;
    bool first_block_done = false;  // (a failure before that sends the search for a block start on)
    uint64_t member_out0 = 0;       // output offset where the current member began; UINT64_MAX = before this chunk
};

// Decodes from `bit` (inside a member when in_member, else in front of a member header) until the first block start at or after
// stop_bit, the end of the data, or something that is not a gzip member.
template <class T>
void decode_run(const uint8_t *data, uint64_t size, uint64_t bit, bool in_member, uint64_t stop_bit, ChunkOut<T> &R) {
    BitIn in;
    in.base = data, in.end = data + size;
    in.seek(bit);
    R.start_bit = bit, R.n = 0, R.segs.clear(), R.first_block_done = false, R.stop = kStopError;
    R.member_out0 = in_member ? UINT64_MAX : 0;
    if (R.out.size() < (size_t)kWin + (1u << 20)) R.out.resize((size_t)kWin + (1u << 20));
    T *o = R.out.data() + kWin;
    uint64_t n = 0, cap = R.out.size() - kWin, seg0 = 0;
    auto grow = [&](uint64_t want) {
        if (n + want <= cap) return;
        R.out.resize((size_t)kWin + (size_t)std::max<uint64_t>(cap * 2, n + want + (1u << 20)));
        o = R.out.data() + kWin, cap = R.out.size() - kWin;
    };
    Codes C;
    auto finish = [&](Stop s, bool inm) {
        if (n > seg0) R.segs.push_back(Segment{seg0, n, false, 0, 0});
        R.n = n, R.stop = s, R.end_in_member = inm, R.end_bit = in.bitpos();
    };
    for (;;) {
        if (!in_member) {
            // member header (RFC 1952); anything else after a member is trailing garbage (zlib: ignored)
            in.align();
            const uint64_t at = in.bitpos() >> 3;
            if (at >= size) return finish(kStopInputEnd, false);
            if (size - at < 2 || data[at] != 0x1f || data[at + 1] != 0x8b) return finish(kStopNoMember, false);
            if (size - at < 10) return finish(kStopInputEnd, false);
            if (data[at + 2] != 8 || (data[at + 3] & 0xe0)) return finish(kStopError, false);
            const int flg = data[at + 3];
            uint64_t q = at + 10;
            if (flg & 4) {
                if (q + 2 > size) return finish(kStopInputEnd, false);
                q += 2 + (uint64_t)(data[q] | data[q + 1] << 8);
            }
            for (int t = 0; t < 2; ++t)
                if (flg & (t ? 16 : 8)) {
                    while (q < size && data[q]) ++q;
                    ++q;
                }
            if (flg & 2) q += 2;
            if (q > size) return finish(kStopInputEnd, false);
            in.seek(q * 8);
            in_member = true;
            R.member_out0 = n;
        }
        if (in.bitpos() >= stop_bit || n >= kMaxChunkOut) return finish(kStopBoundary, true);
        const uint32_t hdr = in.take(3);
        if (in.over) return finish(kStopInputEnd, true);
        const int final_block = hdr & 1, type = hdr >> 1;
        if (type == 3) return finish(kStopError, true);
        if (type == 0) {
            in.align();
            const uint32_t len = in.take(16), nlen = in.take(16);
            if (in.over) return finish(kStopInputEnd, true);
            if ((len ^ 0xffff) != nlen) return finish(kStopError, true);
            const uint64_t at = in.bitpos() >> 3;
            const uint64_t have = std::min<uint64_t>(len, size - at);
            grow(have);
            for (uint64_t i = 0; i < have; ++i) o[n + i] = (T)data[at + i];
            n += have;
            if (have < len) { in.seek(size * 8); return finish(kStopInputEnd, true); }
            in.seek((at + len) * 8);
        } else {
            if (type == 1) fixed_codes(C);
            else if (!read_dynamic(in, C, false)) return finish(in.over ? kStopInputEnd : kStopError, true);
            const uint64_t back0 = R.member_out0 == UINT64_MAX ? (uint64_t)kWin : 0, base0 = R.member_out0 == UINT64_MAX ? 0 : R.member_out0;
            for (;;) {
                grow(300);
                // The common case, without the bookkeeping of the careful path below: with 48 bits in the buffer a whole symbol --
                // a literal / length code (15), its extra bits (5), a distance code (15), its extra bits (13) -- needs no refill and
                // cannot run past the input.  The bit state lives in locals for the run.
                if (in.cnt < 48) in.refill();
                if (in.cnt >= 48) {
                    uint64_t buf = in.buf;
                    int cnt = in.cnt;
                    const uint8_t *p = in.p;
                    bool leave = false;   // a symbol for the careful path: a long code, the end of the block, an error
                    while (n + 300 <= cap) {
                        if (cnt < 48) {
                            if (p + 8 > in.end) break;
                            uint64_t w;
                            memcpy(&w, p, 8);
                            buf |= w << cnt;
                            p += (63 - cnt) >> 3;
                            cnt |= 56;
                        }
                        const uint16_t e = C.lit.fast[buf & ((1u << kLitBits) - 1)];
                        if (!e) { leave = true; break; }
                        const int sym = e >> 4;
                        if (sym < 256) {
                            buf >>= (e & 15), cnt -= (e & 15);
                            o[n++] = (T)sym;
                            continue;
                        }
                        if (sym == 256 || sym >= 286 || C.dist_none) { leave = true; break; }
                        uint64_t b2 = buf >> (e & 15);
                        int used = e & 15;
                        const int ls = sym - 257;
                        const uint32_t len = kLenBase[ls] + (uint32_t)(b2 & ((1u << kLenExtra[ls]) - 1));
                        b2 >>= kLenExtra[ls], used += kLenExtra[ls];
                        const uint16_t de = C.dist.fast[b2 & ((1u << kDistBits) - 1)];
                        if (!de || (de >> 4) >= 30) { leave = true; break; }
                        b2 >>= (de & 15), used += (de & 15);
                        const int ds = de >> 4;
                        const uint32_t dist = kDistBase[ds] + (uint32_t)(b2 & ((1u << kDistExtra[ds]) - 1));
                        used += kDistExtra[ds];
                        if (dist > n - base0 + back0) { leave = true; break; }   // (the careful path reports it)
                        buf >>= used, cnt -= used;
                        const T *src = o + n - dist;
                        T *dst = o + n;
                        if (dist >= len) memcpy(dst, src, len * sizeof(T));
                        else for (uint32_t i2 = 0; i2 < len; ++i2) dst[i2] = src[i2];
                        n += len;
                    }
                    in.buf = buf, in.cnt = cnt, in.p = p;
                    if (!leave) continue;   // (room or input to fetch: round again)
                }
                int s = C.lit.decode(in);
                if (s < 0) return finish(in.over ? kStopInputEnd : kStopError, true);
                if (in.over) return finish(kStopInputEnd, true);
                if (s < 256) { o[n++] = (T)s; continue; }
                if (s == 256) break;
                s -= 257;
                if (s >= 29) return finish(kStopError, true);
                const uint32_t len = kLenBase[s] + in.take(kLenExtra[s]);
                if (C.dist_none) return finish(kStopError, true);
                const int ds = C.dist.decode(in);
                if (ds < 0 || ds >= 30) return finish(in.over ? kStopInputEnd : kStopError, true);
                const uint32_t dist = kDistBase[ds] + in.take(kDistExtra[ds]);
                if (in.over) return finish(kStopInputEnd, true);
                // how far back a match may reach: into this member's output, and (member begun before this chunk) the window
                const uint64_t avail = R.member_out0 == UINT64_MAX ? n + kWin : n - R.member_out0;
                if (dist > avail) return finish(kStopError, true);
                const T *src = o + n - dist;   // (n - dist may be negative: the window lies in front of o)
                T *dst = o + n;
                if (dist >= len) memcpy(dst, src, len * sizeof(T));
                else for (uint32_t i = 0; i < len; ++i) dst[i] = src[i];
                n += len;
            }
        }
        R.first_block_done = true;
        if (final_block) {
            in.align();
            const uint64_t at = in.bitpos() >> 3;
            if (at + 8 > size) { in.seek(size * 8); return finish(kStopInputEnd, true); }
            Segment sg{seg0, n, true, 0, 0};
            memcpy(&sg.crc, data + at, 4), memcpy(&sg.isize, data + at + 4, 4);
            R.segs.push_back(sg);
            seg0 = n;
            in.seek((at + 8) * 8);
            in_member = false;
        }
    }
}

// the first bit position in [from, to) that reads as the start of a non-final dynamic block with complete codes
bool find_block_start(const uint8_t *data, uint64_t size, uint64_t from, uint64_t to, uint64_t &found) {
    BitIn in;
    in.base = data, in.end = data + size;
    Codes C;
    for (uint64_t b = from; b < to; ++b) {
        const uint64_t byte = b >> 3;
        if (byte + 8 > size) return false;
        const uint32_t three = (uint32_t)((data[byte] | (uint32_t)data[byte + 1] << 8) >> (b & 7)) & 7u;
        if (three != 4u) continue;   // BFINAL = 0, BTYPE = 2 (bits, LSB first: 0, then 10b)
        in.seek(b + 3);
        if (read_dynamic(in, C, true) && !in.over) { found = b; return true; }
    }
    return false;
}

}  // namespace

struct PInflate {
    int fd = -1;
    const uint8_t *data = nullptr;
    uint64_t size = 0;
    int threads = 1;
    uint64_t chunk = 2u << 20;
    // where the next round starts
    uint64_t bit = 0;
    bool in_member = false;
    uint8_t window[kWin];
    uint32_t run_crc = 0;       // CRC-32 / length of the current member so far
    uint64_t run_len = 0;
    bool done = false, failed = false;
    int lone_rounds = 0;        // consecutive rounds in which no guessed chunk was accepted
    unsigned lone_waited = 0;   // rounds since
    std::vector<uint8_t> out;   // the bytes of the round being decoded (the producer's)
    std::atomic<uint64_t> stat_rounds{0}, stat_chunks{0}, stat_accepted{0};
    // The rounds run ahead of the reader on a thread of their own (two finished rounds may wait): the parser that calls
    // pinflate_read works on one round's bytes while the next ones are decoded.
    std::thread producer;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::vector<uint8_t>> ready;
    bool prod_over = false, prod_failed = false, stop = false;
    std::vector<uint8_t> cur;   // the round the reader is in
    uint64_t cur_pos = 0;
    void produce();
    ChunkOut<uint8_t> first;
    std::vector<ChunkOut<uint16_t>> guess;

    bool round();
};

void PInflate::produce() {
    for (;;) {
        const bool more = round();
        std::unique_lock<std::mutex> lk(mu);
        if (!out.empty()) {   // (also of a round that failed: what it decoded in front of the error is the reader's, as gzread's is)
            cv.wait(lk, [&] { return stop || ready.size() < 2; });
            if (stop) return;
            ready.emplace_back(std::move(out));
            out = std::vector<uint8_t>();
        }
        if (failed) { prod_failed = prod_over = true; cv.notify_all(); return; }
        if (!more || done) { prod_over = true; cv.notify_all(); return; }
        cv.notify_all();
        if (stop) return;
    }
}

bool PInflate::round() {
    out.clear();
    if (done || failed) return false;
    ++stat_rounds;
    const uint64_t byte0 = bit >> 3;
    // two rounds in a row in which no guessed chunk fitted: one chunk per round from here on, with another try every 32 rounds
    // (a stretch of stored blocks inside a file that is compressible again after it)
    int n_chunks = (lone_rounds >= 2 && (++lone_waited & 31)) ? 1 : threads;
    n_chunks = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)n_chunks, (size - byte0 + chunk - 1) / chunk));
    if ((int)guess.size() < n_chunks - 1) guess.resize((size_t)n_chunks - 1);
    std::vector<uint8_t> tried((size_t)n_chunks, 0);
    auto bound = [&](int j) { return (byte0 + (uint64_t)j * chunk) * 8; };   // (the last chunk stops at its end too: a round's output is bounded)
    auto work = [&](int j) {
        if (j == 0) {
            // the window: what a match of the first block may reach (a member begun in an earlier round)
            if (first.out.size() < (size_t)kWin + (1u << 20)) first.out.resize((size_t)kWin + (1u << 20));
            memcpy(first.out.data(), window, kWin);
            decode_run<uint8_t>(data, size, bit, in_member, bound(1), first);
            return;
        }
        ChunkOut<uint16_t> &G = guess[(size_t)j - 1];
        if (G.out.size() < (size_t)kWin + (1u << 20)) G.out.resize((size_t)kWin + (1u << 20));
        for (int i = 0; i < kWin; ++i) G.out[(size_t)i] = (uint16_t)(256 + i);
        G.stop = kStopError;
        uint64_t from = bound(j);
        // (a compressor starts a block every few tens of kilobytes; a chunk without a dynamic block in its first 512 KB -- stored
        // or fixed-code data -- is not searched to its end: that would cost seconds for nothing)
        const uint64_t to = std::min<uint64_t>(std::min<uint64_t>(bound(j + 1), from + 8 * kMaxScanBytes), size * 8);
        for (int attempt = 0; attempt < 8; ++attempt) {
            uint64_t s;
            if (!find_block_start(data, size, from, to, s)) return;
            decode_run<uint16_t>(data, size, s, true, bound(j + 1), G);
            if (G.stop != kStopError || G.first_block_done) { tried[(size_t)j] = 1; return; }
            from = s + 1;   // (not a block after all)
        }
    };
    if (n_chunks == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int j = 1; j < n_chunks; ++j) th.emplace_back(work, j);
        work(0);
        for (auto &t : th) t.join();
    }
    stat_chunks += (uint64_t)n_chunks;
    // in order: which guessed chunks continue the one before
    if (first.stop == kStopError) {
        // what was decoded in front of the bad code / the failed CRC-32 is handed out first, as zlib's gzread hands out everything it
        // decoded before it reports the error; the NEXT call then fails
        failed = true;
        if (!first.n) return false;
        out.assign(first.out.data() + kWin, first.out.data() + kWin + first.n);
        return true;
    }
    int accepted = 1;
    uint64_t end_bit = first.end_bit;
    bool end_inm = first.end_in_member;
    Stop last_stop = first.stop;
    for (int j = 1; j < n_chunks; ++j) {
        const ChunkOut<uint16_t> &G = guess[(size_t)j - 1];
        if (last_stop != kStopBoundary || !end_inm || !tried[(size_t)j] || G.stop == kStopError || G.start_bit != end_bit) break;
        ++accepted, end_bit = G.end_bit, end_inm = G.end_in_member, last_stop = G.stop;
    }
    stat_accepted += (uint64_t)accepted;
    if (n_chunks > 1) lone_rounds = accepted == 1 ? lone_rounds + 1 : 0;   // (a round of one chunk says nothing about the guesses)
    // sizes, windows, bytes
    std::vector<uint64_t> off((size_t)accepted + 1, 0);
    off[1] = first.n;
    for (int j = 1; j < accepted; ++j) off[(size_t)j + 1] = off[(size_t)j] + guess[(size_t)j - 1].n;
    out.resize((size_t)off[(size_t)accepted]);
    std::vector<std::vector<uint8_t>> wins((size_t)accepted);   // wins[j] = the kWin bytes in front of chunk j + 1
    auto tail_window = [&](const uint8_t *prev, const uint8_t *bytes, uint64_t n, std::vector<uint8_t> &w) {
        w.resize(kWin);
        if (n >= (uint64_t)kWin) memcpy(w.data(), bytes + n - kWin, kWin);
        else {
            memcpy(w.data(), prev + n, (size_t)(kWin - n));
            if (n) memcpy(w.data() + (kWin - n), bytes, (size_t)n);
        }
    };
    tail_window(window, first.out.data() + kWin, first.n, wins[0]);
    for (int j = 1; j < accepted; ++j) {
        // the last kWin symbols of chunk j through the window in front of it
        const ChunkOut<uint16_t> &G = guess[(size_t)j - 1];
        const uint64_t n = G.n, t0 = n > (uint64_t)kWin ? n - kWin : 0;
        std::vector<uint8_t> tail((size_t)(n - t0));
        const uint16_t *sy = G.out.data() + kWin;
        const uint8_t *w = wins[(size_t)j - 1].data();
        for (uint64_t i = t0; i < n; ++i) tail[(size_t)(i - t0)] = sy[i] < 256 ? (uint8_t)sy[i] : w[sy[i] - 256];
        tail_window(w, tail.data(), n - t0, wins[(size_t)j]);
    }
    std::vector<std::vector<uint32_t>> crcs((size_t)accepted);
    auto emit = [&](int j) {
        uint8_t *dst = out.data() + off[(size_t)j];
        const std::vector<Segment> *segs;
        if (j == 0) {
            if (first.n) memcpy(dst, first.out.data() + kWin, (size_t)first.n);
            segs = &first.segs;
        } else {
            const ChunkOut<uint16_t> &G = guess[(size_t)j - 1];
            const uint16_t *sy = G.out.data() + kWin;
            const uint8_t *w = wins[(size_t)j - 1].data();
            for (uint64_t i = 0; i < G.n; ++i) dst[i] = sy[i] < 256 ? (uint8_t)sy[i] : w[sy[i] - 256];
            segs = &G.segs;
        }
        for (const Segment &s : *segs) {
            uint32_t c = (uint32_t)crc32(0L, Z_NULL, 0);
            for (uint64_t p = s.begin; p < s.end; p += 1u << 30)
                c = (uint32_t)crc32(c, dst + p, (uInt)std::min<uint64_t>(1u << 30, s.end - p));
            crcs[(size_t)j].push_back(c);
        }
    };
    if (accepted == 1) emit(0);
    else {
        std::vector<std::thread> th;
        for (int j = 1; j < accepted; ++j) th.emplace_back(emit, j);
        emit(0);
        for (auto &t : th) t.join();
    }
    // the members' check values, in order
    for (int j = 0; j < accepted; ++j) {
        const std::vector<Segment> &segs = j == 0 ? first.segs : guess[(size_t)j - 1].segs;
        for (size_t k = 0; k < segs.size(); ++k) {
            const uint64_t len = segs[k].end - segs[k].begin;
            run_crc = run_len ? (uint32_t)crc32_combine(run_crc, crcs[(size_t)j][k], (z_off_t)len) : crcs[(size_t)j][k];
            if (!run_len && !len) run_crc = (uint32_t)crc32(0L, Z_NULL, 0);
            run_len += len;
            if (segs[k].ends_member) {
                if (run_crc != segs[k].crc || (uint32_t)run_len != segs[k].isize) {
                    // a member whose check values do not hold: its bytes are handed out (zlib checks at the member's end too), nothing behind it
                    failed = true;
                    out.resize((size_t)(off[(size_t)j] + segs[k].end));
                    return false;
                }
                run_crc = 0, run_len = 0;
            }
        }
    }
    memcpy(window, wins[(size_t)accepted - 1].data(), kWin);
    bit = end_bit, in_member = end_inm;
    if (last_stop != kStopBoundary) done = true;   // the end of the data, a truncated member, or trailing garbage
    return !out.empty() || !done;
}

PInflate *pinflate_open(const char *path, int threads) {
    if (threads < 2) return nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return nullptr;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 18) { close(fd); return nullptr; }
    void *m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { close(fd); return nullptr; }
    const uint8_t *d = (const uint8_t *)m;
    if (d[0] != 0x1f || d[1] != 0x8b) { munmap(m, (size_t)sb.st_size); close(fd); return nullptr; }   // (gzread copies such a file as it is)
    (void)madvise(m, (size_t)sb.st_size, MADV_SEQUENTIAL);
    PInflate *h = new PInflate;
    h->fd = fd, h->data = d, h->size = (uint64_t)sb.st_size, h->threads = std::min(threads, 64);
    if (const char *e = getenv("NDGPU_INFLATE_CHUNK")) h->chunk = std::max<uint64_t>(1024, strtoull(e, nullptr, 10));
    memset(h->window, 0, kWin);
    h->producer = std::thread([h] { h->produce(); });
    return h;
}

int64_t pinflate_read(PInflate *h, void *buf, uint64_t len) {
    uint8_t *dst = (uint8_t *)buf;
    uint64_t got = 0;
    while (got < len) {
        if (h->cur_pos >= h->cur.size()) {
            std::unique_lock<std::mutex> lk(h->mu);
            h->cv.wait(lk, [&] { return !h->ready.empty() || h->prod_over; });
            if (h->ready.empty()) {   // the data is over, or the round after the last one handed out failed
                if (h->prod_failed && !got) return -1;
                break;
            }
            h->cur = std::move(h->ready.front());
            h->ready.pop_front();
            h->cur_pos = 0;
            h->cv.notify_all();
            continue;
        }
        const uint64_t n = std::min<uint64_t>(len - got, h->cur.size() - h->cur_pos);
        memcpy(dst + got, h->cur.data() + h->cur_pos, (size_t)n);
        got += n, h->cur_pos += n;
    }
    return (int64_t)got;
}

void pinflate_stats(PInflate *h, uint64_t out[3]) { out[0] = h->stat_rounds, out[1] = h->stat_chunks, out[2] = h->stat_accepted; }

void pinflate_close(PInflate *h) {
    if (!h) return;
    {
        std::lock_guard<std::mutex> g(h->mu);
        h->stop = true;
    }
    h->cv.notify_all();
    if (h->producer.joinable()) h->producer.join();
    if (h->data) munmap((void *)h->data, (size_t)h->size);
    if (h->fd >= 0) close(h->fd);
    delete h;
}

}  // namespace ndovl
