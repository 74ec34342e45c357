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
#include <algorithm>
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

// check_realign_nextdenovo (minimap2/map.c:803-821)
int realign_class(int rev, uint32_t qs, uint32_t qe, uint32_t qlen, uint32_t ts, uint32_t te, uint32_t tlen, int32_t h1, int32_t h2)
{
	const uint32_t a = (uint32_t)h1, b = (uint32_t)h2;
	if (rev) {
		if (qs <= a && ts <= a) return 1;
		else if (qlen - qe <= a && tlen - te <= a) return 2;
	} else {
		if (qlen - qe <= a && ts <= a) return 4;
		else if (qs <= a && tlen - te <= a) return 7;
	}
	if (h2 > 0) {
		if (qs <= b && qe + b >= qlen) return 8;
		if (ts <= b && te + b >= tlen) return 9;
	}
	return 0;
}

struct Reg {  // the fields of mm_reg1_t the --step 2 path looks at
	int32_t rev, rid, qs, qe, rs, re, mlen, blen;
};

inline Reg reg_of(const ndgpu_ovl_rec &o) { return Reg{(int32_t)o.rev, (int32_t)o.qname, (int32_t)o.qs, (int32_t)o.qe, (int32_t)o.ts, (int32_t)o.te,
	                                                    (int32_t)o.match, (int32_t)o.tname}; }

// update_reg_nextdenovo (minimap2/map.c:823-877)
int update_regs(const Reg *rn, int n_new, Reg *reg, int s, int e, int t_l, const uint32_t *batch_len, int32_t maxhan1, int32_t maxhan2)
{
	int i, c, t, l, pi;
	uint32_t alnlen;
	for (i = c = 0; s < e; s++) {
		Reg &r = reg[s];
		if (r.mlen != 2) continue;
		for (l = -1, pi = i, alnlen = 0, t = 0; i < n_new && t < 10; i++) {
			const Reg &q = rn[i];
			if (q.rid == r.blen) {
				if (l == -1) l = i, alnlen = (uint32_t)(rn[i].blen * 0.8);
				if ((uint32_t)q.blen >= alnlen && realign_class(q.rev, (uint32_t)q.qs, (uint32_t)q.qe, (uint32_t)t_l, (uint32_t)q.rs, (uint32_t)q.re,
				                                                batch_len[q.rid], maxhan1, maxhan2)) {
					l = i;
					break;
				}
				t++;
				if ((uint32_t)q.qs <= (uint32_t)maxhan2 && (uint32_t)q.qe + (uint32_t)maxhan2 >= (uint32_t)t_l) {
					l = i;
					c++;
					break;
				}
			} else if (l >= 0) {
				i--;
				break;
			}
		}
		if (l >= 0) {
			const Reg &q = rn[l];
			r.rev = q.rev, r.qs = q.qs, r.qe = q.qe, r.rs = q.rs, r.re = q.re, r.mlen = q.mlen, r.blen = q.blen;
		} else i = pi;
	}
	return c;
}

} // namespace

extern "C" {

// `minimap2-nd --step 2` as nextDenovo runs it (no --mode: --mode 2, minimap2/options.c:56; worker_for, map.c:988-1126) for the
// reads of one query file against one index part.  The device maps (ndgpu_ovl_map_regs: the hits themselves); this function does
// what the reference's worker does with them, read by read -- the per-target marking (:997-1030), then the re-alignment of the
// marked candidates with the short k-mer sketch (--kn 17 --wn 10, main.c:197): a read with fewer than 200 candidates becomes a
// one-read index and every candidate is mapped against it (:1033-1075: units of the second device call, target = the query's
// number in q_mini), a read with more is mapped against its candidates cn at a time (:1076-1123: units of the third call, targets
// = the batch in t_mini), then the writer's record filter (:1305-1309).  The re-alignments of a read are independent of one another
// except for the count of "the query is contained" verdicts that ends them: all of them are computed, the sequential rule is
// applied afterwards.  Returns the records in output order (malloc'd), < 0 on error.
int64_t ndgpu_ovl_map2_realign(ndgpu_ovl_index *idx, ndgpu_ovl_index *q_mini, ndgpu_ovl_index *t_mini, const ndgpu_ovl_opt *opt,
                               int32_t mid_occ, int32_t cn, uint32_t n_t, const uint32_t *t_words, uint64_t t_n_words,
                               const uint64_t *t_word_off, const uint32_t *t_lens, const uint32_t *t_ids, uint32_t n_q,
                               const uint32_t *q_words, uint64_t q_n_words, const uint64_t *q_word_off, const uint32_t *q_lens,
                               const uint32_t *q_ids, ndgpu_ovl_rec10 **recs)
{
	*recs = nullptr;
	if (opt->step != 2 || (opt->mode != 2 && opt->mode != 1) || cn < 1) return -1;
	const int one_read_below = opt->mode == 2 ? 200 : 20;  // map.c:1032
	ndgpu_ovl_rec *raw = nullptr;
	uint32_t *cnt = nullptr;
	const int64_t n_raw = ndgpu_ovl_map_regs(idx, opt, mid_occ, n_q, q_words, q_n_words, q_word_off, q_lens, q_ids, nullptr, nullptr, 0, &raw, &cnt, nullptr);
	if (n_raw < 0) return n_raw;
	try {
		std::vector<Reg> R((size_t)n_raw);
		for (int64_t i = 0; i < n_raw; ++i) R[(size_t)i] = reg_of(raw[i]);
		std::vector<uint64_t> off((size_t)n_q + 1, 0);
		for (uint32_t i = 0; i < n_q; ++i) off[i + 1] = off[i] + cnt[i];
		ndgpu_ovl_free(raw), ndgpu_ovl_free(cnt);
		raw = nullptr, cnt = nullptr;

		// ---- marking (map.c:997-1030), and what is to be re-aligned
		struct Plan { int c = 0, seq_index = 0; };
		std::vector<Plan> plan(n_q);
		std::vector<int32_t> first(n_t, -1);
		struct UnitA { uint32_t q; uint64_t k; };            // candidate R[k] of query q against the query's one-read index
		struct UnitB { uint32_t q; uint64_t w0, w1, k0, k1; }; // query q against candidates want_b[w0, w1); updates R[k0, k1)
		std::vector<UnitA> ua;
		std::vector<UnitB> ub;
		std::vector<uint32_t> want_b, len_b;
		for (uint32_t i = 0; i < n_q; ++i) {
			Reg *r0 = R.data() + off[i];
			const int n = (int)(off[i + 1] - off[i]);
			const uint32_t ql = q_lens[i];
			int c = 0, seq_index = 0;
			for (int k = 0; k < n; ++k) {
				Reg &r = r0[k];
				const uint32_t tl = t_lens[r.rid], tp = (uint32_t)r.mlen, longer = tl > ql ? tl : ql;
				if (first[r.rid] < 0) first[r.rid] = k; else r.mlen = 1;
				const int l = first[r.rid];
				Reg &head = r0[l];
				if (l != k && (head.mlen == 2 || r.blen < head.blen * 0.8 || (uint32_t)r.blen < longer / 3)) continue;
				if (r.qe - r.qs >= opt->minlen && tp >= r.blen * opt->minide && tp >= (uint32_t)opt->minmatch) {
					if (realign_class(r.rev, (uint32_t)r.qs, (uint32_t)r.qe, ql, (uint32_t)r.rs, (uint32_t)r.re, tl, opt->maxhan1, 0)) {
						if (head.mlen == 3) c--;
						head.mlen = 2;
						seq_index++;
					} else if ((uint32_t)r.qs <= (uint32_t)opt->maxhan2 && (uint32_t)r.qe + (uint32_t)opt->maxhan2 >= ql) {
						head.mlen = 3;
						if (++c >= (int)kMaxCon) break;
					}
				}
			}
			for (int k = 0; k < n; ++k) first[r0[k].rid] = -1;
			plan[i].c = c, plan[i].seq_index = seq_index;
			if (c >= (int)kMaxCon) continue;
			if (seq_index < one_read_below) {
				for (int k = 0; k < n; ++k)
					if (r0[k].mlen == 2) ua.push_back(UnitA{i, off[i] + (uint64_t)k});
			} else {
				const int per = (int)((float)seq_index / ((seq_index + cn - 1) / cn) + 0.999);
				int si = 0, tp = 0;
				uint64_t w0 = want_b.size();
				for (int k = 0; k < n; ++k) {
					if (r0[k].mlen != 2) continue;
					want_b.push_back((uint32_t)r0[k].rid), len_b.push_back(t_lens[r0[k].rid]);
					if (++si >= per) {
						ub.push_back(UnitB{i, w0, (uint64_t)want_b.size(), off[i] + (uint64_t)tp, off[i] + (uint64_t)k + 1});
						si = 0, tp = k + 1, w0 = want_b.size();
					}
				}
				if (si) ub.push_back(UnitB{i, w0, (uint64_t)want_b.size(), off[i] + (uint64_t)tp, off[i] + (uint64_t)n});
			}
		}

		// ---- the device calls, a bounded number of bases at a time (every unit is sketched whole)
		const uint64_t kBases = 1500000000ull;
		ndgpu_ovl_opt ro = *opt;
		std::vector<std::vector<Reg>> res_a(ua.size()), res_b(ub.size());
		uint64_t most_anchors = 0;
		auto run_units = [&](ndgpu_ovl_index *ix, size_t n_units, auto seq_of, auto want_of, auto store, int how = 1) -> int {
			size_t u0 = 0;
			while (u0 < n_units) {
				size_t u1 = u0;
				uint64_t bases = 0;
				while (u1 < n_units && (u1 == u0 || bases + seq_of(u1).second <= kBases)) bases += seq_of(u1).second, ++u1;
				const uint32_t m = (uint32_t)(u1 - u0);
				std::vector<uint64_t> woff(m), want_off(m + 1, 0);
				std::vector<uint32_t> lens(m), ids(m, 0u), want;
				for (uint32_t j = 0; j < m; ++j) {
					const auto sq = seq_of(u0 + j);
					woff[j] = sq.first, lens[j] = (uint32_t)sq.second;
					want_of(u0 + j, want);
					want_off[j + 1] = want.size();
				}
				ndgpu_ovl_rec *rr = nullptr;
				uint32_t *rc = nullptr;
				uint64_t ma = 0;
				const int64_t nn = ndgpu_ovl_map_regs(ix, &ro, mid_occ, m, seq_of.words, seq_of.n_words, woff.data(), lens.data(), ids.data(), want_off.data(),
				                                      want.data(), how, &rr, &rc, &ma);
				if (nn < 0) return (int)nn;
				most_anchors = std::max(most_anchors, ma);
				uint64_t at = 0;
				for (uint32_t j = 0; j < m; ++j) {
					std::vector<Reg> v(rc[j]);
					for (uint32_t x = 0; x < rc[j]; ++x) v[x] = reg_of(rr[at + x]);
					at += rc[j];
					store(u0 + j, std::move(v));
				}
				ndgpu_ovl_free(rr), ndgpu_ovl_free(rc);
				u0 = u1;
			}
			return 0;
		};
		struct SeqT {  // a unit of the first kind is a TARGET read mapped against its query's one-read index
			const uint32_t *words; uint64_t n_words; const uint64_t *word_off; const uint32_t *lens; const std::vector<UnitA> *ua; const std::vector<Reg> *R;
			std::pair<uint64_t, uint64_t> operator()(size_t u) const { const int32_t rid = (*R)[(*ua)[u].k].rid; return {word_off[rid], lens[rid]}; }
		} seq_t{t_words, t_n_words, t_word_off, t_lens, &ua, &R};
		struct SeqQ {  // a unit of the second kind is the QUERY read mapped against a batch of its candidates
			const uint32_t *words; uint64_t n_words; const uint64_t *word_off; const uint32_t *lens; const std::vector<UnitB> *ub;
			std::pair<uint64_t, uint64_t> operator()(size_t u) const { const uint32_t q = (*ub)[u].q; return {word_off[q], lens[q]}; }
		} seq_q{q_words, q_n_words, q_word_off, q_lens, &ub};
		int rc0 = run_units(q_mini, ua.size(), seq_t, [&](size_t u, std::vector<uint32_t> &w) { w.push_back(ua[u].q); },
		                    [&](size_t u, std::vector<Reg> &&v) { res_a[u] = std::move(v); },
		                    // --mode 1 maps these through mm_map_nextdenovo1 (map.c:1047): mm_chain_dp_nextdenovo, which thins the anchors of
		                    // a mapping that has more than 100,000 of them before it chains (bit 1 of `nameless`)
		                    opt->mode == 1 ? 3 : 1);
		if (rc0 < 0) return rc0;
		rc0 = run_units(t_mini, ub.size(), seq_q, [&](size_t u, std::vector<uint32_t> &w) { w.insert(w.end(), want_b.begin() + ub[u].w0, want_b.begin() + ub[u].w1); },
		                [&](size_t u, std::vector<Reg> &&v) { res_b[u] = std::move(v); });
		if (rc0 < 0) return rc0;

		// ---- apply, read by read, in the reference's order; then the writer's filter (map.c:1305-1309)
		std::vector<ndgpu_ovl_rec10> out;
		size_t ia = 0, ib = 0;
		for (uint32_t i = 0; i < n_q; ++i) {
			Reg *r0 = R.data() + off[i];
			const int n = (int)(off[i + 1] - off[i]);
			const uint32_t ql = q_lens[i];
			int c = plan[i].c;
			if (c < (int)kMaxCon && plan[i].seq_index < one_read_below) {
				bool stop = false;
				for (; ia < ua.size() && ua[ia].q == i; ++ia) {
					if (stop) continue;
					Reg &r = R[ua[ia].k];
					const std::vector<Reg> &rn = res_a[ia];
					if (rn.empty()) continue;  // mm_map found nothing: the marked hit stays
					const uint32_t tl = t_lens[r.rid], alnlen = (uint32_t)(rn[0].blen * 0.8);
					size_t l = 0;
					for (; l < rn.size() && l < 10; ++l)
						if ((uint32_t)rn[l].blen >= alnlen && realign_class(rn[l].rev, (uint32_t)rn[l].qs, (uint32_t)rn[l].qe, tl, (uint32_t)rn[l].rs,
						                                                    (uint32_t)rn[l].re, ql, opt->maxhan1, opt->maxhan2)) break;
					if (l == 10 || l == rn.size()) l = 0;
					r.rev = rn[l].rev, r.qs = rn[l].rs, r.qe = rn[l].re, r.rs = rn[l].qs, r.re = rn[l].qe, r.mlen = rn[l].mlen, r.blen = rn[l].blen;
					if ((uint32_t)r.qs <= (uint32_t)opt->maxhan2 && (uint32_t)r.qe + (uint32_t)opt->maxhan2 >= ql)
						if (++c >= (int)kMaxCon) stop = true;
				}
			} else if (c < (int)kMaxCon) {
				bool stop = false;
				int si = 0;
				size_t b = ib;
				while (b < ub.size() && ub[b].q == i) ++b;
				// (the reference numbers the candidates of a batch as it walks the hits, batch by batch, and stops numbering when it stops)
				for (; ib < b; ++ib) {
					if (stop) continue;
					const UnitB &U = ub[ib];
					si = 0;
					for (uint64_t k = U.k0; k < U.k1; ++k)
						if (R[k].mlen == 2) R[k].blen = si++;
					std::vector<Reg> &rn = res_b[ib];
					std::stable_sort(rn.begin(), rn.end(), [](const Reg &x, const Reg &y) { return x.rid < y.rid; });  // qsort + cmpfunc_nextdenovo
					c += update_regs(rn.data(), (int)rn.size(), r0, (int)(U.k0 - off[i]), (int)(U.k1 - off[i]), (int)ql, len_b.data() + U.w0, opt->maxhan1,
					                 opt->maxhan2);
					if (c >= (int)kMaxCon) stop = true;
				}
			}
			while (ia < ua.size() && ua[ia].q == i) ++ia;
			while (ib < ub.size() && ub[ib].q == i) ++ib;
			for (int k = 0; k < n; ++k) {
				const Reg &r = r0[k];
				const uint32_t tl = t_lens[r.rid];
				if (!((r.qe - r.qs >= opt->minlen || r.mlen == r.blen) && (r.mlen == 3 || (r.mlen >= r.blen * opt->minide && r.mlen >= opt->minmatch)) &&
				      r.blen >= (int32_t)ql / 50 && r.blen >= (int32_t)tl / 50)) continue;
				ndgpu_ovl_rec10 o;
				o.rev = (uint32_t)r.rev, o.qname = q_ids[i], o.qs = (uint32_t)r.qs, o.qe = (uint32_t)r.qe, o.qlen = ql, o.tname = t_ids[r.rid];
				o.ts = (uint32_t)r.rs, o.te = (uint32_t)r.re, o.tlen = tl;
				o.identity = (uint32_t)((uint64_t)(uint32_t)r.mlen * 10000ull / (uint64_t)(uint32_t)r.blen);
				out.push_back(o);
			}
		}
		*recs = (ndgpu_ovl_rec10*)malloc(sizeof(ndgpu_ovl_rec10) * (out.empty() ? 1 : out.size()));
		if (!out.empty()) memcpy(*recs, out.data(), sizeof(ndgpu_ovl_rec10) * out.size());
		return (int64_t)out.size();
	} catch (...) {
		if (raw) ndgpu_ovl_free(raw);
		if (cnt) ndgpu_ovl_free(cnt);
		return -2;
	}
}

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
