// Resident read database (see nd_host.h).  Host copy + the pool that is uploaded
// once to HBM.  Input is the reference's .2bit payload (lib/bseq.c:114-139).
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "nd_host.h"

namespace ndgpu {

namespace {
// reverse the order of the sixteen 2-bit groups of a word: MSB-first -> LSB-first
inline uint32_t flip_groups(uint32_t w) {
    w = __builtin_bswap32(w);
    w = ((w & 0xf0f0f0f0u) >> 4) | ((w & 0x0f0f0f0fu) << 4);
    w = ((w & 0xccccccccu) >> 2) | ((w & 0x33333333u) << 2);
    return w;
}
inline uint32_t code_at(const uint32_t *lsb_words, uint64_t base) {
    return (lsb_words[base >> 4] >> ((base & 15u) * 2u)) & 3u;
}
}  // namespace

ReadDb::ReadDb(uint32_t n_reads, const uint32_t *words, const uint64_t *word_off, const uint32_t *len) {
    len_.assign(len, len + n_reads);
    fwd_off_.resize(n_reads);
    rc_off_.resize(n_reads);
    uint64_t nw = 0;
    for (uint32_t r = 0; r < n_reads; r++) {
        const uint64_t w = ((uint64_t)len[r] + 15) / 16 + 1;  // +1 pad word between reads
        fwd_off_[r] = nw * 16;
        nw += w;
        rc_off_[r] = nw * 16;
        nw += w;
        total_ += len[r];
    }
    pool_.assign(nw + 2, 0);
    // reads are independent: blocks of 256 reads dealt to the host's cores (a 5.6 Gb DB is ~10^10 base moves)
    auto one = [&](uint32_t r) {
        const uint32_t L = len[r];
        const uint64_t w = ((uint64_t)L + 15) / 16;
        uint32_t *f = pool_.data() + fwd_off_[r] / 16;
        const uint32_t *src = words + word_off[r];
        for (uint64_t i = 0; i < w; i++) f[i] = flip_groups(src[i]);
        if (L & 15u) f[w - 1] &= (1u << ((L & 15u) * 2u)) - 1u;
        // reverse complement (.2bit code: A0 C1 G2 T3 -> complement = 3 - code)
        uint32_t *rc = pool_.data() + rc_off_[r] / 16;
        for (uint32_t i = 0; i < L; i++) {
            const uint32_t c = 3u - code_at(f, L - 1 - i);
            rc[i >> 4] |= c << ((i & 15u) * 2u);
        }
    };
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nth = (unsigned)std::min<uint64_t>(hw, total_ / 2000000 + 1);
    if (nth <= 1) {
        for (uint32_t r = 0; r < n_reads; r++) one(r);
    } else {
        std::atomic<uint32_t> next(0);
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nth; t++)
            th.emplace_back([&] {
                for (;;) {
                    const uint32_t a = next.fetch_add(256);
                    if (a >= n_reads) break;
                    const uint32_t b = std::min<uint32_t>(n_reads, a + 256);
                    for (uint32_t r = a; r < b; r++) one(r);
                }
            });
        for (auto &x : th) x.join();
    }
}

std::string ReadDb::window(uint32_t r, uint32_t start, uint32_t end, int rev) const {
    static const char kAsc[4] = {'A', 'C', 'G', 'T'};
    const uint64_t off = (uint64_t)window_offset(r, start, end, rev);
    const uint32_t n = end - start + 1;
    std::string s(n, 'A');
    const uint32_t *p = pool_.data();
    for (uint32_t i = 0; i < n; i++) s[i] = kAsc[code_at(p, off + i)];
    return s;
}

}  // namespace ndgpu
