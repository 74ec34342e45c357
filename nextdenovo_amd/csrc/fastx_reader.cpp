// FASTA / FASTQ[.gz] reader of the read-ingestion step (`seq_dump`, util/seq_dump.c:60-118), host code of the overlap library.
// Restates kseq_read() as the reference instantiates it (lib/bseq.h:3 KSEQ_INIT(gzFile, gzread); util/kseq.h:178-222) -- only what
// seq_dump uses of it: the sequences, in order, and where the stream stops:
//   * a record starts at the next '>' or '@' (anything before the first one is skipped); the rest of the header line is ignored;
//   * sequence lines run until a line that starts with '>', '+' or '@'; empty lines are skipped; the line terminator is '\n',
//     and a '\r' before it is dropped when the sequence so far is longer than one character (util/kseq.h:131);
//   * after '+': the rest of that line is skipped, quality lines are read until they cover the sequence; a quality string of another
//     length ends the file (kseq_read returns -2 and seq_dump's loop stops); after a FASTQ record the next header is searched for.
// The reader streams: the file is inflated through a 1 MB window, records are handed out in chunks the caller sizes, so a
// multi-GB .fastq.gz never sits in memory (the Python parser this replaces read the whole file).  A gzip file is inflated by
// several threads (pinflate.cpp: NDGPU_INFLATE_THREADS, default = the CPUs the process may use, at most 16; 1 = zlib's gzread,
// which also reads what is not a regular gzip file).
#include <sched.h>
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <string>

#include "../../include/ndgpu_overlap.h"
#include "pinflate.h"

namespace {

// the CPUs this process may run on at once: its affinity mask and, in a container, the cgroup's quota (cpu.max / cfs_quota_us)
int usable_cpus() {
    int n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    long long quota = -1, period = 100000;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32];
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(g, "%lld", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(h, "%lld", &period) != 1) period = 100000;
            fclose(h);
        }
    }
    if (quota > 0 && period > 0) n = (int)std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
    return n < 1 ? 1 : n;
}

// gzread over one of two readers
struct GzIn {
    gzFile f = nullptr;
    ndovl::PInflate *par = nullptr;
    bool open(const char *path, int threads) {
        par = ndovl::pinflate_open(path, threads);
        if (par) return true;
        f = gzopen(path, "r");
        if (!f) return false;
        (void)gzbuffer(f, 1 << 20);
        return true;
    }
    int read(void *buf, unsigned len) { return par ? (int)ndovl::pinflate_read(par, buf, len) : gzread(f, buf, len); }
    void close() {
        if (par) ndovl::pinflate_close(par);
        if (f) gzclose(f);
        par = nullptr, f = nullptr;
    }
};

struct Stream {
    GzIn f;
    unsigned char buf[1 << 20];
    int begin = 0, end = 0;
    bool eof = false, err = false;

    int getc() {
        if (err) return -3;
        if (begin >= end) {
            if (eof) return -1;
            begin = 0;
            end = f.read(buf, sizeof(buf));
            if (end == 0) { eof = true; return -1; }
            if (end < 0) { eof = true; err = true; end = 0; return -3; }
        }
        return buf[begin++];
    }
    // append the rest of the current line to s (the '\n' is consumed, not stored); -1: nothing read and the stream is at its end
    int rest_of_line(std::string &s) {
        bool got = false;
        for (;;) {
            if (err) return -3;
            if (begin >= end) {
                if (eof) break;
                begin = 0;
                end = f.read(buf, sizeof(buf));
                if (end == 0) { eof = true; break; }
                if (end < 0) { eof = true; err = true; end = 0; return -3; }
            }
            const unsigned char *nl = (const unsigned char *)memchr(buf + begin, '\n', (size_t)(end - begin));
            const int i = nl ? (int)(nl - buf) : end;
            got = true;
            s.append((const char *)buf + begin, (size_t)(i - begin));
            begin = i + 1;
            if (nl) break;
        }
        if (!got && eof && begin >= end) return -1;
        if (s.size() > 1 && s.back() == '\r') s.pop_back();
        return 0;
    }
    // the header's name: up to the first white-space character; *delim = that character (0 at the end of the stream)
    int name(int *delim, std::string *keep) {
        bool got = false;
        *delim = 0;
        keep->clear();
        for (;;) {
            if (err) return -3;
            if (begin >= end) {
                if (eof) break;
                begin = 0;
                end = f.read(buf, sizeof(buf));
                if (end == 0) { eof = true; break; }
                if (end < 0) { eof = true; err = true; end = 0; return -3; }
            }
            int i = begin;
            while (i < end && !(buf[i] == ' ' || (buf[i] >= '\t' && buf[i] <= '\r'))) i++;
            got = true;
            keep->append((const char *)buf + begin, (size_t)(i - begin));
            begin = i + 1;
            if (i < end) {
                *delim = buf[i];
                break;
            }
        }
        return (!got && eof && begin >= end) ? -1 : 0;
    }
};

}  // namespace

struct ndgpu_fastx {
    Stream st;
    int last_char = 0;
    bool done = false, have = false;   // have: `seq` holds a record that did not fit the caller's last buffer
    std::string seq, qual, skip, name;

    // kseq_read: 1 = a record is in `seq`, 0 = the stream is over for the caller (end, truncated quality), -3 = read error
    int next() {
        int c;
        if (last_char == 0) {
            while ((c = st.getc()) >= 0 && c != '>' && c != '@') {}
            if (c < 0) return c == -3 ? -3 : 0;
            last_char = c;
        }
        seq.clear();
        qual.clear();
        int delim;
        const int r = st.name(&delim, &name);
        if (r < 0) return r == -3 ? -3 : 0;
        if (delim != '\n') {
            skip.clear();
            (void)st.rest_of_line(skip);
        }
        while ((c = st.getc()) >= 0 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            seq.push_back((char)c);
            (void)st.rest_of_line(seq);
        }
        if (c == '>' || c == '@') last_char = c;
        if (c != '+') return st.err ? -3 : 1;  // FASTA record (or the last record of the stream)
        while ((c = st.getc()) >= 0 && c != '\n') {}
        if (c == -1) return 0;  // no quality string: kseq_read -2
        if (c == -3) return -3;
        while (st.rest_of_line(qual) >= 0 && qual.size() < seq.size()) {}
        if (st.err) return -3;
        last_char = 0;
        if (seq.size() != qual.size()) return 0;  // kseq_read -2: seq_dump stops reading this file
        return 1;
    }
};

extern "C" {

static int inflate_threads() {
    if (const char *e = getenv("NDGPU_INFLATE_THREADS")) return std::max(1, atoi(e));
    return std::min(16, usable_cpus());
}

ndgpu_fastx *ndgpu_fastx_open(const char *path) {
    ndgpu_fastx *h = new ndgpu_fastx;
    if (!h->st.f.open(path, inflate_threads())) {
        delete h;
        return nullptr;
    }
    return h;
}

/* gzread by several threads, by itself (what ndgpu_fastx_open reads through) */
struct ndgpu_gzin { GzIn in; };

ndgpu_gzin *ndgpu_gzin_open(const char *path, int threads) {
    ndgpu_gzin *h = new ndgpu_gzin;
    if (!h->in.open(path, threads > 0 ? threads : inflate_threads())) {
        delete h;
        return nullptr;
    }
    return h;
}

int64_t ndgpu_gzin_read(ndgpu_gzin *h, void *buf, uint32_t len) { return h->in.read(buf, len); }

int ndgpu_gzin_stats(const ndgpu_gzin *h, uint64_t out[3]) {
    out[0] = out[1] = out[2] = 0;
    if (!h->in.par) return 0;
    ndovl::pinflate_stats(h->in.par, out);
    return 1;
}

void ndgpu_gzin_close(ndgpu_gzin *h) {
    if (!h) return;
    h->in.close();
    delete h;
}

int64_t ndgpu_fastx_read(ndgpu_fastx *h, uint8_t *buf, uint64_t cap, uint64_t *off, uint32_t *len, int64_t max_recs) {
    return ndgpu_fastx_read_named(h, buf, cap, off, len, nullptr, max_recs);
}

int64_t ndgpu_fastx_read_named(ndgpu_fastx *h, uint8_t *buf, uint64_t cap, uint64_t *off, uint32_t *len, uint32_t *ids, int64_t max_recs) {
    int64_t n = 0;
    uint64_t used = 0;
    while (n < max_recs) {
        if (!h->have) {
            if (h->done) break;
            const int r = h->next();
            if (r == -3) return -3;
            if (r == 0) {
                h->done = true;
                break;
            }
            h->have = true;
        }
        const uint64_t l = h->seq.size();
        if (used + l > cap) {
            if (n == 0) return -4;  // the record alone does not fit: ndgpu_fastx_pending() tells its length
            break;
        }
        memcpy(buf + used, h->seq.data(), (size_t)l);
        off[n] = used;
        if (ids) ids[n] = (uint32_t)strtoul(h->name.c_str(), nullptr, 10);  // minimap2-nd names reads by number (map.c:1298-1300)
        len[n] = (uint32_t)(l > 0xffffffffull ? 0xffffffffull : l);
        used += l;
        n++;
        h->have = false;
    }
    return n;
}

uint64_t ndgpu_fastx_pending(const ndgpu_fastx *h) { return h->have ? (uint64_t)h->seq.size() : 0; }

void ndgpu_fastx_close(ndgpu_fastx *h) {
    if (!h) return;
    h->st.f.close();
    delete h;
}

}  // extern "C"
