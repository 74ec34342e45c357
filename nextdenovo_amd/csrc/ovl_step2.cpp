// ovl_step2.cpp -- the host half of `minimap2-nd --step 2`: the dovetail / contained filter that follows the mapper's own
// record filter, its `.bl` table, and the 10-field record encoder.
//
//   filter   reference lib/ovl.c:449-563 (filter_ovl): one state record per read name, kept for the whole run -- how many
//            overlaps reach its 5' / 3' end, the best identity / longest span of the dovetails at either end, how often it
//            looked contained, the longest internal overlap and a list of covered intervals.  An overlap is kept when it is a
//            dovetail within maxhan1 (or covers one of the reads end to end within maxhan1) and neither read has looked
//            contained twice.  The verdict of a record depends on every record before it, whichever query it came from, so this
//            is a sequential pass over the device's records (a few hundred ns per record), not a kernel.
//   .bl      lib/ovl.c:339-362 (out_bl): one line per read at the end of the run, in the order the reference's hash table
//            (util/khash.h, identity hash, triangular probing, in-place rehash at 77 % load) iterates; ReadTable below lays its
//            keys out the same way, so the lines come out in the same order.
//   encoder  lib/ovl.c:205-253 (encode_ovl_i): ten varints per record, read lengths only when the name changes.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ndgpu_overlap.h"

namespace {

constexpr uint32_t kMaxCon = 2;      // MAX_CON
constexpr uint32_t kEdgeBack = 10;   // EDGEBACKLEN
constexpr uint16_t kListStep = 5;    // INIT_ALNM

struct Span {
	uint32_t s = 0, e = 0;
};

struct ReadState {
	uint16_t lc = 0, rc = 0;           // overlaps reaching the 5' / 3' end (within maxhan2)
	uint16_t cur = 0;                  // slot of `spans` written last
	uint32_t con = 0;                  // times the read looked contained
	uint32_t lim = 0, rim = 0;         // best identity of a dovetail at the 5' / 3' end (15-bit fields in the reference)
	uint32_t llm = 0, rlm = 0;         // longest dovetail at either end
	uint32_t len = 0;
	Span longest;                      // longest overlap that was neither a dovetail nor a containment
	std::vector<Span> spans;           // covered intervals, pulled in by kEdgeBack at both ends; e == 0 marks a free slot

	void merge_spans()
	{
		const size_t n = spans.size();
		for (size_t i = 1; i < n; ++i) { // insertion sort by start: equal starts keep their order
			const Span t = spans[i];
			size_t j = i;
			for (; j > 0 && spans[j - 1].s > t.s; --j) spans[j] = spans[j - 1];
			spans[j] = t;
		}
		size_t i = 0;
		while (n && i < n - 1) {
			if (!spans[i].e) { ++i; continue; }
			size_t j = i + 1;
			for (; j < n; ++j) {
				if (spans[j].e <= spans[i].e) spans[j].e = 0;
				else if (spans[j].s <= spans[i].e && spans[j].e >= spans[i].e) spans[i].e = spans[j].e, spans[j].e = 0;
				else break;
			}
			i = j;
		}
	}

	uint16_t free_slot()
	{
		const uint16_t n = (uint16_t)spans.size();
		if (cur != n - 1)
			for (uint16_t i = 0; i < n; ++i) if (!spans[i].e) return i;
		merge_spans();
		for (uint16_t i = 0; i < n; ++i) if (!spans[i].e) return i;
		spans.resize((size_t)n + kListStep);
		return n;
	}

	void cover(uint32_t a, uint32_t b)
	{
		if (con >= kMaxCon) return;
		cur = free_slot();
		spans[cur].s = a + kEdgeBack, spans[cur].e = b - kEdgeBack;
	}

	void note_internal(uint32_t a, uint32_t b)
	{
		if (con < kMaxCon && b - a > longest.e - longest.s) longest.s = a, longest.e = b;
	}

	void contained_once_more()
	{
		if (++con >= kMaxCon) std::vector<Span>().swap(spans);
	}
};

// open addressing with khash's probe sequence and growth rule, so that iteration visits the names in the reference's order
class ReadTable {
public:
	static constexpr uint32_t npos = 0xffffffffu;

	uint32_t find(uint32_t key) const
	{
		if (slot_.empty()) return npos;
		const uint32_t mask = (uint32_t)slot_.size() - 1;
		uint32_t i = key & mask, step = 0;
		const uint32_t start = i;
		while (slot_[i].used && slot_[i].key != key) {
			i = (i + (++step)) & mask;
			if (i == start) return npos;
		}
		return slot_[i].used ? i : npos;
	}

	uint32_t insert(uint32_t key)
	{
		if (occupied_ >= upper_) {
			const uint32_t nb = (uint32_t)slot_.size();
			grow(nb > (size_ << 1) ? nb - 1 : nb + 1);
		}
		const uint32_t mask = (uint32_t)slot_.size() - 1;
		uint32_t i = key & mask, step = 0;
		while (slot_[i].used && slot_[i].key != key) i = (i + (++step)) & mask;
		if (!slot_[i].used) {
			slot_[i].used = 1, slot_[i].key = key, slot_[i].val = ReadState();
			++size_, ++occupied_;
		}
		return i;
	}

	ReadState &at(uint32_t i) { return slot_[i].val; }
	size_t buckets() const { return slot_.size(); }
	bool used(size_t i) const { return slot_[i].used != 0; }
	uint32_t key(size_t i) const { return slot_[i].key; }

private:
	struct Slot {
		uint8_t used = 0; // during a rehash: 1 = still where the old table had it, 2 = placed
		uint32_t key = 0;
		ReadState val;
	};
	std::vector<Slot> slot_;
	uint32_t size_ = 0, occupied_ = 0, upper_ = 0;

	void grow(uint32_t want)
	{
		uint32_t nb = want;
		--nb, nb |= nb >> 1, nb |= nb >> 2, nb |= nb >> 4, nb |= nb >> 8, nb |= nb >> 16, ++nb;
		if (nb < 4) nb = 4;
		if (size_ >= (uint32_t)(nb * 0.77 + 0.5)) return;
		const uint32_t old = (uint32_t)slot_.size();
		if (nb > old) slot_.resize(nb);
		std::vector<uint8_t> taken(nb, 0);
		for (uint32_t j = 0; j < old; ++j) {
			if (slot_[j].used != 1) continue;
			uint32_t key = slot_[j].key;
			ReadState val = std::move(slot_[j].val);
			slot_[j].used = 2; // moved out
			for (;;) {
				uint32_t i = key & (nb - 1), step = 0;
				while (taken[i]) i = (i + (++step)) & (nb - 1);
				taken[i] = 1;
				if (i < old && slot_[i].used == 1) { // an element that has not moved yet lives here: it is displaced and placed next
					std::swap(key, slot_[i].key);
					std::swap(val, slot_[i].val);
					slot_[i].used = 2;
				} else {
					slot_[i].key = key, slot_[i].val = std::move(val);
					break;
				}
			}
		}
		for (uint32_t i = 0; i < nb; ++i) slot_[i].used = taken[i];
		occupied_ = size_;
		upper_ = (uint32_t)(nb * 0.77 + 0.5);
	}
};

} // namespace

struct ndgpu_s2_state {
	ReadTable table;

	// first sight of a read, or one more overlap at its ends
	uint32_t touch(uint32_t name, uint32_t len, uint32_t lo_gap, uint32_t hi_gap, uint32_t han2, bool target_side)
	{
		uint32_t k = table.find(name);
		if (k != ReadTable::npos) {
			ReadState &r = table.at(k);
			if (r.con < kMaxCon) {
				// (the target side looks at the 3' counter before it bumps the 5' one, lib/ovl.c:480)
				if (lo_gap <= han2 && (target_side ? r.rc : r.lc) < UINT16_MAX) r.lc++;
				if (hi_gap <= han2 && r.rc < UINT16_MAX) r.rc++;
			}
			return k;
		}
		k = table.insert(name);
		ReadState &r = table.at(k);
		r.len = len;
		r.spans.assign(kListStep, Span());
		if (lo_gap <= han2) r.lc++;
		if (hi_gap <= han2) r.rc++;
		return k;
	}

	bool keep(const ndgpu_ovl_rec10 &o, int32_t maxhan1, int32_t maxhan2)
	{
		const uint32_t h1 = (uint32_t)maxhan1, h2 = (uint32_t)maxhan2; // the reference compares uint32 with int32: unsigned
		touch(o.qname, o.qlen, o.qs, o.qlen - o.qe, h2, false);
		const uint32_t tk = touch(o.tname, o.tlen, o.ts, o.tlen - o.te, h2, true);
		ReadState &t = table.at(tk);
		ReadState &q = table.at(table.find(o.qname)); // the insertion of the target may have moved the table
		q.cover(o.qs, o.qe);
		t.cover(o.ts, o.te);
		if (q.con < kMaxCon && o.qs <= h2 && o.qe + h2 >= o.qlen) { q.contained_once_more(); return false; }
		if (t.con < kMaxCon && o.ts <= h2 && o.te + h2 >= o.tlen) { t.contained_once_more(); return false; }
		if (q.con >= kMaxCon || t.con >= kMaxCon) return false;
		const uint32_t span = o.qe - o.qs > o.te - o.ts ? o.qe - o.qs : o.te - o.ts;
		const uint32_t q_lo = o.qs, q_hi = o.qlen - o.qe, t_lo = o.ts, t_hi = o.tlen - o.te;
		int q_end = -1, t_end = -1; // which end of either read the overlap reaches: 0 = 5', 1 = 3'
		uint32_t gq = 0, gt = 0;
		if (o.rev) {
			if (q_lo <= h1 && t_lo <= h1) q_end = 0, t_end = 0, gq = q_lo, gt = t_lo;
			else if (q_hi <= h1 && t_hi <= h1) q_end = 1, t_end = 1, gq = q_hi, gt = t_hi;
		} else {
			if (q_hi <= h1 && t_lo <= h1) q_end = 1, t_end = 0, gq = q_hi, gt = t_lo;
			else if (q_lo <= h1 && t_hi <= h1) q_end = 0, t_end = 1, gq = q_lo, gt = t_hi;
		}
		if (q_end >= 0) {
			if (gq <= h2 && gt <= h2) {
				uint32_t &ql = q_end ? q.rlm : q.llm, &tl = t_end ? t.rlm : t.llm;
				uint32_t &qi = q_end ? q.rim : q.lim, &ti = t_end ? t.rim : t.lim;
				if (span > ql) ql = span;
				if (span > tl) tl = span;
				if (o.identity > qi) qi = o.identity & 0x7fff;
				if (o.identity > ti) ti = o.identity & 0x7fff;
			}
			return true;
		}
		if (o.qs <= h1 && o.qe + h1 >= o.qlen) return true; // contained once the read ends are clipped: kept
		if (o.ts <= h1 && o.te + h1 >= o.tlen) return true;
		q.note_internal(o.qs, o.qe);
		t.note_internal(o.ts, o.te);
		return false;
	}
};

namespace {

int put_varint(uint8_t *out, uint32_t v) // lib/ovl.c:10-29,129-145: 7 bits per byte, most significant group first
{
	if (v <= 127) { out[0] = (uint8_t)v; return 1; }
	int m = 0;
	for (int sh = 28; sh >= 0; sh -= 7) {
		const uint32_t g = v >> sh & 127;
		if (g > 0 || m > 0) out[m++] = (uint8_t)(g | 128);
	}
	out[m - 1] &= 127;
	return m;
}

int put_record10(uint8_t *out, const ndgpu_ovl_rec10 &o, uint32_t prev[2])
{
	uint32_t f[10], flags = o.rev;
	const uint32_t qspan = o.qe - o.qs, tspan = o.te - o.ts;
	if (o.qname >= prev[0]) f[0] = o.qname - prev[0]; else flags |= 2, f[0] = prev[0] - o.qname;
	if (o.tname >= prev[1]) f[4] = o.tname - prev[1]; else flags |= 4, f[4] = prev[1] - o.tname;
	f[7] = o.qname == prev[0] ? 0 : o.qlen;
	f[8] = o.tname == prev[1] ? 0 : o.tlen;
	if (qspan >= tspan) f[6] = qspan - tspan; else flags |= 8, f[6] = tspan - qspan;
	prev[0] = o.qname, prev[1] = o.tname;
	f[1] = flags & 0xff, f[2] = o.qs, f[3] = qspan, f[5] = o.ts, f[9] = o.identity;
	int n = 0;
	for (int i = 0; i < 10; ++i) n += put_varint(out + n, f[i]);
	return n;
}

} // namespace

extern "C" {

ndgpu_s2_state *ndgpu_s2_new(void) { return new (std::nothrow) ndgpu_s2_state(); }

void ndgpu_s2_free(ndgpu_s2_state *st) { delete st; }

int64_t ndgpu_s2_filter_encode(ndgpu_s2_state *st, const ndgpu_ovl_rec10 *recs, int64_t n, int32_t maxhan1, int32_t maxhan2,
                               uint32_t prev[2], uint8_t **out, uint8_t *kept)
{
	*out = nullptr;
	if (!st || n < 0) return -1;
	uint8_t *buf = (uint8_t*)malloc((size_t)(n > 0 ? n : 1) * 50);
	if (!buf) return -1;
	int64_t bytes = 0;
	try {
		for (int64_t i = 0; i < n; ++i) {
			const bool k = st->keep(recs[i], maxhan1, maxhan2);
			if (kept) kept[i] = k ? 1 : 0;
			if (k) bytes += put_record10(buf + bytes, recs[i], prev);
		}
	} catch (...) {
		free(buf);
		return -1;
	}
	*out = buf;
	return bytes;
}

int64_t ndgpu_s2_bl(ndgpu_s2_state *st, char **text)
{
	*text = nullptr;
	if (!st) return -1;
	std::string s;
	char line[160];
	try {
		for (size_t k = 0; k < st->table.buckets(); ++k) {
			if (!st->table.used(k)) continue;
			ReadState &r = st->table.at((uint32_t)k);
			if (r.con < kMaxCon) {
				snprintf(line, sizeof(line), "%u\t%u\t%hu\t%hu\t%u\t%u\t%u\t%u\t%u\t%u\t%u", st->table.key(k), r.con, r.lc, r.rc, r.lim, r.rim,
				         r.llm, r.rlm, r.len, r.longest.s, r.longest.e);
				s += line;
				r.merge_spans();
				for (const Span &v : r.spans)
					if (v.e) {
						snprintf(line, sizeof(line), "\t%u\t%u", v.s - kEdgeBack, v.e + kEdgeBack);
						s += line;
					}
				s += "\n";
			} else {
				snprintf(line, sizeof(line), "%u\t%u\n", st->table.key(k), r.con);
				s += line;
			}
		}
	} catch (...) {
		return -1;
	}
	char *p = (char*)malloc(s.size() + 1);
	if (!p) return -1;
	memcpy(p, s.c_str(), s.size() + 1);
	*text = p;
	return (int64_t)s.size();
}

} // extern "C"
