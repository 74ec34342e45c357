// ovlsort_engine.hip -- host orchestration + C ABI of the overlap sort / filter stage (util/ovl_sort.c path).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/ndgpu_overlap.h"
#include "ovl_device.h"
#include "ovl_pool.h"

namespace ndovl {

#define HIP_OK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "[ndgpu_overlap] HIP error at %s:%d\n", __FILE__, __LINE__); ndovl::device_check((int)_e, "hip call"); } } while (0)

namespace {
template <class T> struct Buf {
	T *p = nullptr;
	size_t n = 0;
	Buf() = default;
	explicit Buf(size_t c) { n = c; if (c) p = (T*)pool_alloc(c * sizeof(T)); }
	Buf(const Buf&) = delete;
	Buf &operator=(const Buf&) = delete;
	~Buf() { if (p) pool_free(p); }
};
}

void launch_expand_flags(const OvlRec *raw, uint64_t n, const uint32_t *seed_len, uint32_t n_ids, uint32_t *hq, uint32_t *ht, uint32_t *mq,
                         uint32_t *mt, hipStream_t s);
void launch_expand_count(uint64_t n, const uint32_t *file_of, const uint64_t *file_start, const uint32_t *hq, const uint32_t *ht,
                         const uint64_t *mqs, const uint64_t *mts, uint32_t *sel, hipStream_t s);
void launch_sel_count(const uint32_t *sel, uint64_t n, uint32_t *cnt, hipStream_t s);
void launch_expand_count_piece(uint64_t n, const uint32_t *hq, const uint32_t *ht, const uint64_t *mqs, const uint64_t *mts, uint64_t carry_q,
                               uint64_t carry_t, uint32_t *sel, hipStream_t s);
void launch_cand_hist(const OvlRec *raw, const uint32_t *sel, uint64_t n, uint32_t *hist, hipStream_t s);
void launch_range_sel(const OvlRec *raw, const uint8_t *sel, uint64_t n, uint32_t lo, uint32_t hi, uint32_t *out, hipStream_t s);
void launch_narrow_u32_u8(const uint32_t *in, uint64_t n, uint8_t *out, hipStream_t s);
void launch_expand_write(const OvlRec *raw, uint64_t n, const uint32_t *sel, const uint64_t *pos, OvlRec *cand, uint32_t *k_span,
                         uint32_t *k_match, uint32_t *k_seed, hipStream_t s);
void launch_iota(uint32_t *a, uint64_t n, hipStream_t s);
void launch_gather_u32(const uint32_t *src, const uint32_t *idx, uint64_t n, uint32_t *dst, hipStream_t s);
void launch_seed_flag(const OvlRec *cand, const uint32_t *perm, uint64_t n, uint32_t *flag, hipStream_t s);
void launch_seed_start(const uint32_t *flag, const uint64_t *rank, uint64_t n, uint64_t *start, hipStream_t s);
int sort_pairs_u32(void *tmp, size_t &tmp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n,
                   hipStream_t s);
void launch_seed_filter(const OvlRec *cand, const uint32_t *perm, const uint64_t *seed_start, uint32_t n_seeds, uint64_t n_cand,
                        const uint32_t *seed_len, int max_bin_cov, int flank, int min_seed_len, uint32_t max_bins, uint32_t *kept, OvlRec *out,
                        uint32_t *n_out, uint32_t *bl_id, uint8_t *bl_kind, bool hq, hipStream_t s);
void launch_compact_seed_recs(const uint64_t *seed_start, uint32_t n_seeds, const OvlRec *out, const uint32_t *n_out, const uint64_t *off,
                              OvlRec *dense, hipStream_t s);

struct SortRun {
	hipStream_t st = nullptr;
	Buf<uint8_t> *tmp = nullptr;
	size_t tmp_bytes = 0;
	~SortRun() { delete tmp; }
	void *temp(size_t b)
	{
		if (!tmp || tmp->n < b) { delete tmp; tmp = new Buf<uint8_t>(b + b / 4 + 256); }
		return tmp->p;
	}
	void exscan(const uint32_t *in, uint64_t *out, size_t n)
	{
		size_t tb = 0;
		exscan_u32_to_u64(nullptr, tb, in, out, n, st);
		exscan_u32_to_u64(temp(tb), tb, in, out, n, st);
	}
	// stable LSD pass: reorder perm by key[perm]
	void pass(const uint32_t *key, uint32_t *perm, uint32_t *perm2, uint32_t *k1, uint32_t *k2, size_t n)
	{
		launch_gather_u32(key, perm, n, k1, st);
		size_t tb = 0;
		sort_pairs_u32(nullptr, tb, k1, k2, perm, perm2, n, st);
		sort_pairs_u32(temp(tb), tb, k1, k2, perm, perm2, n, st);
		HIP_OK(hipMemcpyAsync(perm, perm2, n * 4, hipMemcpyDeviceToDevice, st));
	}
};

} // namespace ndovl

using namespace ndovl;

// the records of a call, in the malloc'd block the caller gets (grown by realloc when a call sorts in seed ranges): the download
// goes straight into it -- through a std::vector the records were zero-filled, downloaded and copied once more, 5 ms per 18 MB
struct RecOut {
	OvlRec *p = nullptr;
	size_t n = 0, cap = 0;
	~RecOut() { free(p); }
	RecOut() = default;
	RecOut(const RecOut &) = delete;
	RecOut &operator=(const RecOut &) = delete;
	size_t size() const { return n; }
	OvlRec *grow(size_t more) {   // room for `more` records at the end; returns where they go
		if (n + more + 1 > cap) {
			const size_t want = std::max(n + more + 1, cap + cap / 2);
			OvlRec *q = (OvlRec*)realloc(p, want * sizeof(OvlRec));
			if (!q) throw std::runtime_error("malloc");
			p = q, cap = want;
		}
		OvlRec *at = p + n;
		n += more;
		return at;
	}
	OvlRec *release() { OvlRec *q = p ? p : (OvlRec*)malloc(sizeof(OvlRec)); p = nullptr, n = cap = 0; return q; }
};

struct SortOut {
	RecOut recs;
	std::vector<uint32_t> bl_id;
	std::vector<uint8_t> bl_kind;
	uint64_t seeds = 0;
};

// S2 + S3 over one set of candidates (all of a call, or those of one seed range): the sorted, filtered records and the `.bl`
// verdicts are appended to `o`
static void sort_and_filter(SortRun &R, const OvlRec *cand_p, const uint32_t *k_span_p, const uint32_t *k_match_p, const uint32_t *k_seed_p,
                            uint64_t nc, const uint32_t *d_seed_p, const uint32_t *seed_len, uint32_t n_ids, int32_t min_seed_len,
                            int32_t max_bin_cov, int32_t max_flank_len, bool hq_mode, SortOut &o)
{
	Buf<uint32_t> perm(nc), perm2(nc), k1(nc), k2(nc);
	// S2: (seed asc, match desc, span asc), stable
	launch_iota(perm.p, nc, R.st);
	R.pass(k_span_p, perm.p, perm2.p, k1.p, k2.p, nc);
	R.pass(k_match_p, perm.p, perm2.p, k1.p, k2.p, nc);
	R.pass(k_seed_p, perm.p, perm2.p, k1.p, k2.p, nc);

	// seeds
	Buf<uint32_t> flag(nc + 1);
	Buf<uint64_t> rank(nc + 1);
	HIP_OK(hipMemsetAsync(flag.p, 0, (nc + 1) * 4, R.st));
	launch_seed_flag(cand_p, perm.p, nc, flag.p, R.st);
	R.exscan(flag.p, rank.p, nc + 1);
	uint64_t n_seeds = 0;
	HIP_OK(hipMemcpyAsync(&n_seeds, rank.p + nc, 8, hipMemcpyDeviceToHost, R.st));
	HIP_OK(hipStreamSynchronize(R.st));
	Buf<uint64_t> sstart(n_seeds + 1);
	launch_seed_start(flag.p, rank.p, nc, sstart.p, R.st);

	// S3
	uint32_t max_len = 0;
	for (uint32_t i = 0; i < n_ids; ++i) max_len = std::max(max_len, seed_len[i]);
	const uint32_t max_bins = (max_len >> 6) + 2;
	Buf<uint32_t> kept(nc + n_seeds + 1), n_out(n_seeds + 1), d_bl_id(n_seeds + 1);
	Buf<uint8_t> d_bl_kind(n_seeds + 1);
	Buf<OvlRec> outrec(nc + n_seeds + 1);
	HIP_OK(hipMemsetAsync(n_out.p, 0, (n_seeds + 1) * 4, R.st));
	launch_seed_filter(cand_p, perm.p, sstart.p, (uint32_t)n_seeds, nc, d_seed_p, max_bin_cov, max_flank_len, min_seed_len, max_bins, kept.p,
	                   outrec.p, n_out.p, d_bl_id.p, d_bl_kind.p, hq_mode, R.st);
	Buf<uint64_t> off(n_seeds + 1);
	R.exscan(n_out.p, off.p, n_seeds + 1);
	uint64_t total = 0;
	HIP_OK(hipMemcpyAsync(&total, off.p + n_seeds, 8, hipMemcpyDeviceToHost, R.st));
	HIP_OK(hipStreamSynchronize(R.st));
	HIP_OK(hipGetLastError());
	Buf<OvlRec> dense(total + 1);
	launch_compact_seed_recs(sstart.p, (uint32_t)n_seeds, outrec.p, n_out.p, off.p, dense.p, R.st);
	std::vector<uint32_t> h_id(n_seeds);
	std::vector<uint8_t> h_kind(n_seeds);
	OvlRec *const dst = o.recs.grow(total);
	if (total) HIP_OK(hipMemcpyAsync(dst, dense.p, total * sizeof(OvlRec), hipMemcpyDeviceToHost, R.st));
	HIP_OK(hipMemcpyAsync(h_id.data(), d_bl_id.p, n_seeds * 4, hipMemcpyDeviceToHost, R.st));
	HIP_OK(hipMemcpyAsync(h_kind.data(), d_bl_kind.p, n_seeds, hipMemcpyDeviceToHost, R.st));
	HIP_OK(hipStreamSynchronize(R.st));
	HIP_OK(hipGetLastError());
	for (uint64_t i = 0; i < n_seeds; ++i)
		if (h_kind[i]) o.bl_id.push_back(h_id[i]), o.bl_kind.push_back(h_kind[i]);
	o.seeds += n_seeds;
}

static void hand_out(SortOut &so, ndgpu_ovl_rec **out, uint32_t **bl_id, uint8_t **bl_kind, int64_t *n_bl)
{
	static_assert(sizeof(OvlRec) == sizeof(ndgpu_ovl_rec), "record layout");
	*bl_id = (uint32_t*)malloc(4 * (so.bl_id.size() + 1));
	*bl_kind = (uint8_t*)malloc(so.bl_kind.size() + 1);
	if (!*bl_id || !*bl_kind) throw std::runtime_error("malloc");
	*out = (ndgpu_ovl_rec*)so.recs.release();
	if (!*out) throw std::runtime_error("malloc");
	if (!so.bl_id.empty()) memcpy(*bl_id, so.bl_id.data(), so.bl_id.size() * 4), memcpy(*bl_kind, so.bl_kind.data(), so.bl_kind.size());
	*n_bl = (int64_t)so.bl_id.size();
}

// The sort when the candidates of a seed file do not fit the device at once (`ovl_sort -m` with less memory than data: the
// reference spills sorted runs to temporary files and merges them, util/ovl_sort.c:1079-1110; the result does not depend on -m).
// Here the raw records stay in host memory (the caller's arrays) and pass the device twice, in pieces:
//   pass A  per piece of a file: which sides of which records become candidates (the "5 misses then stop" rule is per file, so
//           a piece carries the miss counts of the file's earlier pieces), one byte per record kept on the host, and a histogram of
//           candidates per seed;
//   the seeds are cut into consecutive id ranges whose candidates fit;
//   pass B  per range: every piece again, only the sides whose seed lies in the range are expanded; S2 + S3 on them.
// A seed's records depend on its own candidates only and the output is in seed order, so the ranges' outputs, one after the other,
// are the output of the whole sort; equal (seed, match, span) keys keep input order in either form.
static void sort_out_of_core(SortRun &R, const ndgpu_ovl_rec *const *files, const int64_t *n_per_file, int32_t n_files, const uint32_t *seed_len,
                             uint32_t n_ids, int32_t min_seed_len, int32_t max_bin_cov, int32_t max_flank_len, bool hq_mode, uint64_t piece_cap,
                             uint64_t range_cap, SortOut &so, uint64_t *nc_total, uint64_t *n_ranges)
{
	Buf<uint32_t> d_seed(n_ids + 1), hist(n_ids + 1);
	HIP_OK(hipMemcpyAsync(d_seed.p, seed_len, (size_t)n_ids * 4, hipMemcpyHostToDevice, R.st));
	HIP_OK(hipMemsetAsync(hist.p, 0, ((size_t)n_ids + 1) * 4, R.st));
	std::vector<std::vector<uint8_t>> sel_of((size_t)n_files);
	Buf<OvlRec> raw(piece_cap);
	Buf<uint32_t> hq(piece_cap + 1), ht(piece_cap + 1), mq(piece_cap + 1), mt(piece_cap + 1), sel(piece_cap + 1), cnt(piece_cap + 1);
	Buf<uint64_t> mqs(piece_cap + 1), mts(piece_cap + 1), pos(piece_cap + 1);
	Buf<uint8_t> sel8(piece_cap + 1);
	for (int f = 0; f < n_files; ++f) {  // pass A
		const uint64_t nf = (uint64_t)n_per_file[f];
		sel_of[(size_t)f].resize(nf);
		uint64_t carry_q = 0, carry_t = 0;
		for (uint64_t a = 0; a < nf; a += piece_cap) {
			const uint64_t m = std::min<uint64_t>(piece_cap, nf - a);
			HIP_OK(hipMemcpyAsync(raw.p, files[f] + a, m * sizeof(OvlRec), hipMemcpyHostToDevice, R.st));
			HIP_OK(hipMemsetAsync(mq.p, 0, (m + 1) * 4, R.st));
			HIP_OK(hipMemsetAsync(mt.p, 0, (m + 1) * 4, R.st));
			launch_expand_flags(raw.p, m, d_seed.p, n_ids, hq.p, ht.p, mq.p, mt.p, R.st);
			R.exscan(mq.p, mqs.p, m + 1);
			R.exscan(mt.p, mts.p, m + 1);
			launch_expand_count_piece(m, hq.p, ht.p, mqs.p, mts.p, carry_q, carry_t, sel.p, R.st);
			launch_cand_hist(raw.p, sel.p, m, hist.p, R.st);
			launch_narrow_u32_u8(sel.p, m, sel8.p, R.st);
			uint64_t dq = 0, dt = 0;
			HIP_OK(hipMemcpyAsync(sel_of[(size_t)f].data() + a, sel8.p, m, hipMemcpyDeviceToHost, R.st));
			HIP_OK(hipMemcpyAsync(&dq, mqs.p + m, 8, hipMemcpyDeviceToHost, R.st));
			HIP_OK(hipMemcpyAsync(&dt, mts.p + m, 8, hipMemcpyDeviceToHost, R.st));
			HIP_OK(hipStreamSynchronize(R.st));
			carry_q += dq, carry_t += dt;
		}
	}
	std::vector<uint32_t> h_hist((size_t)n_ids + 1);
	HIP_OK(hipMemcpyAsync(h_hist.data(), hist.p, ((size_t)n_ids + 1) * 4, hipMemcpyDeviceToHost, R.st));
	HIP_OK(hipStreamSynchronize(R.st));
	*nc_total = 0, *n_ranges = 0;
	for (uint32_t lo = 0; lo < n_ids;) {  // pass B, range by range
		uint64_t nc = h_hist[lo];
		uint32_t hi = lo + 1;
		while (hi < n_ids && nc + h_hist[hi] <= range_cap) nc += h_hist[hi++];
		if (nc == 0) { lo = hi; continue; }
		if (nc >= 0x7fffffffull) throw std::runtime_error("one seed has more than 2^31 candidates");
		Buf<OvlRec> cand(nc);
		Buf<uint32_t> k_span(nc), k_match(nc), k_seed(nc);
		uint64_t base = 0;
		for (int f = 0; f < n_files; ++f) {
			const uint64_t nf = (uint64_t)n_per_file[f];
			for (uint64_t a = 0; a < nf; a += piece_cap) {
				const uint64_t m = std::min<uint64_t>(piece_cap, nf - a);
				HIP_OK(hipMemcpyAsync(raw.p, files[f] + a, m * sizeof(OvlRec), hipMemcpyHostToDevice, R.st));
				HIP_OK(hipMemcpyAsync(sel8.p, sel_of[(size_t)f].data() + a, m, hipMemcpyHostToDevice, R.st));
				HIP_OK(hipMemsetAsync(cnt.p, 0, (m + 1) * 4, R.st));
				launch_range_sel(raw.p, sel8.p, m, lo, hi, sel.p, R.st);
				launch_sel_count(sel.p, m, cnt.p, R.st);
				R.exscan(cnt.p, pos.p, m + 1);
				uint64_t got = 0;
				HIP_OK(hipMemcpyAsync(&got, pos.p + m, 8, hipMemcpyDeviceToHost, R.st));
				HIP_OK(hipStreamSynchronize(R.st));
				if (base + got > nc) throw std::runtime_error("candidate count changed between the passes");
				launch_expand_write(raw.p, m, sel.p, pos.p, cand.p + base, k_span.p + base, k_match.p + base, k_seed.p + base, R.st);
				HIP_OK(hipStreamSynchronize(R.st));  // (raw / sel are reused by the next piece)
				base += got;
			}
		}
		if (base != nc) throw std::runtime_error("candidate count changed between the passes");
		sort_and_filter(R, cand.p, k_span.p, k_match.p, k_seed.p, nc, d_seed.p, seed_len, n_ids, min_seed_len, max_bin_cov, max_flank_len, hq_mode, so);
		*nc_total += nc, ++*n_ranges;
		lo = hi;
	}
}

static int64_t sort_impl(const ndgpu_ovl_rec *const *files, const int64_t *n_per_file, int32_t n_files, const uint32_t *seed_len,
                         uint32_t n_ids, int32_t min_seed_len, int32_t max_bin_cov, int32_t max_flank_len, ndgpu_ovl_rec **out,
                         uint32_t **bl_id, uint8_t **bl_kind, int64_t *n_bl, ndgpu_ovl_sort_stats *stats, bool hq_mode)
{
	*out = nullptr, *bl_id = nullptr, *bl_kind = nullptr, *n_bl = 0;
	if (stats) memset(stats, 0, sizeof(*stats));
	const bool prof = getenv("NDGPU_PROF") != nullptr;
	const auto tp0 = std::chrono::steady_clock::now();
	auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count(); };
	try {
		int n_dev = 0;
		if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
			fprintf(stderr, "[ndgpu_overlap] no HIP device: the overlap sort has no CPU path\n");
			return -1;
		}
		int dev = 0;
		if (const char *e = getenv("NDGPU_DEVICE")) dev = atoi(e);
		HIP_OK(hipSetDevice(dev % n_dev));
		SortRun R;
		HIP_OK(ndovl::create_stage_stream(&R.st));
		struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } } guard{R.st};
		hipEvent_t ev0, ev1;
		HIP_OK(hipEventCreate(&ev0)); HIP_OK(hipEventCreate(&ev1));
		HIP_OK(hipEventRecord(ev0, R.st));
		const double t_setup = since();

		uint64_t n = 0;
		std::vector<uint64_t> h_fstart((size_t)n_files + 1);
		for (int f = 0; f < n_files; ++f) { h_fstart[f] = n; n += (uint64_t)n_per_file[f]; }
		h_fstart[n_files] = n;
		if (n == 0) { *out = (ndgpu_ovl_rec*)malloc(sizeof(ndgpu_ovl_rec)); *bl_id = (uint32_t*)malloc(4); *bl_kind = (uint8_t*)malloc(1); return 0; }
		// in one piece when the device holds the raw records, their flags and up to two candidates per record (~360 bytes per record
		// with the sort's scratch); otherwise -- or when told to -- in seed ranges (sort_out_of_core)
		uint64_t piece_cap = 0, range_cap = 0;
		if (const char *e = getenv("NDGPU_OVLSORT_PIECE_RECORDS")) piece_cap = strtoull(e, nullptr, 10);
		if (const char *e = getenv("NDGPU_OVLSORT_RANGE_CANDIDATES")) range_cap = strtoull(e, nullptr, 10);
		size_t mem_free = 0, mem_total = 0;
		HIP_OK(hipMemGetInfo(&mem_free, &mem_total));
		const uint64_t avail = (uint64_t)mem_free + (uint64_t)pool_cached_bytes();
		if (piece_cap || range_cap || n * 360ull > avail || 2 * n >= 0x7fffffffull) {
			if (!piece_cap) piece_cap = std::max<uint64_t>(1u << 20, std::min<uint64_t>(n, avail / 8 / 96));
			if (!range_cap) range_cap = std::max<uint64_t>(1u << 20, std::min<uint64_t>(0x7ffffff0ull, avail / 2 / 180));
			SortOut so;
			uint64_t nc_total = 0, n_ranges = 0;
			sort_out_of_core(R, files, n_per_file, n_files, seed_len, n_ids, min_seed_len, max_bin_cov, max_flank_len, hq_mode, piece_cap, range_cap, so,
			                 &nc_total, &n_ranges);
			HIP_OK(hipEventRecord(ev1, R.st));
			HIP_OK(hipStreamSynchronize(R.st));
			const uint64_t kept = so.recs.size();
			hand_out(so, out, bl_id, bl_kind, n_bl);
			if (stats) {
				float ms = 0;
				(void)hipEventElapsedTime(&ms, ev0, ev1);
				stats->gpu_ms = ms, stats->raw_records = n, stats->candidates = nc_total, stats->seeds = so.seeds, stats->kept = kept;
				stats->ranges = n_ranges;
			}
			(void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
			return (int64_t)kept;
		}
		std::vector<uint32_t> h_file_of(n);
		Buf<OvlRec> raw(n);
		for (int f = 0; f < n_files; ++f) {
			std::fill(h_file_of.begin() + h_fstart[f], h_file_of.begin() + h_fstart[f + 1], (uint32_t)f);
			if (n_per_file[f]) HIP_OK(hipMemcpyAsync(raw.p + h_fstart[f], files[f], (size_t)n_per_file[f] * sizeof(OvlRec), hipMemcpyHostToDevice, R.st));
		}
		Buf<uint32_t> file_of(n), d_seed(n_ids + 1);
		Buf<uint64_t> fstart((size_t)n_files + 1);
		HIP_OK(hipMemcpyAsync(file_of.p, h_file_of.data(), n * 4, hipMemcpyHostToDevice, R.st));
		HIP_OK(hipMemcpyAsync(fstart.p, h_fstart.data(), ((size_t)n_files + 1) * 8, hipMemcpyHostToDevice, R.st));
		HIP_OK(hipMemcpyAsync(d_seed.p, seed_len, (size_t)n_ids * 4, hipMemcpyHostToDevice, R.st));

		// S1: candidates
		Buf<uint32_t> hq(n + 1), ht(n + 1), mq(n + 1), mt(n + 1), sel(n + 1), cnt(n + 1);
		Buf<uint64_t> mqs(n + 1), mts(n + 1), pos(n + 1);
		HIP_OK(hipMemsetAsync(mq.p, 0, (n + 1) * 4, R.st));
		HIP_OK(hipMemsetAsync(mt.p, 0, (n + 1) * 4, R.st));
		HIP_OK(hipMemsetAsync(cnt.p, 0, (n + 1) * 4, R.st));
		launch_expand_flags(raw.p, n, d_seed.p, n_ids, hq.p, ht.p, mq.p, mt.p, R.st);
		R.exscan(mq.p, mqs.p, n + 1);
		R.exscan(mt.p, mts.p, n + 1);
		launch_expand_count(n, file_of.p, fstart.p, hq.p, ht.p, mqs.p, mts.p, sel.p, R.st);
		launch_sel_count(sel.p, n, cnt.p, R.st);
		R.exscan(cnt.p, pos.p, n + 1);
		uint64_t nc = 0;
		HIP_OK(hipMemcpyAsync(&nc, pos.p + n, 8, hipMemcpyDeviceToHost, R.st));
		HIP_OK(hipStreamSynchronize(R.st));
		if (nc == 0) { *out = (ndgpu_ovl_rec*)malloc(sizeof(ndgpu_ovl_rec)); *bl_id = (uint32_t*)malloc(4); *bl_kind = (uint8_t*)malloc(1); return 0; }
		if (nc >= 0x7fffffffull) { fprintf(stderr, "[ndgpu_overlap] too many candidates for one sort call\n"); return -3; }
		Buf<OvlRec> cand(nc);
		Buf<uint32_t> k_span(nc), k_match(nc), k_seed(nc);
		launch_expand_write(raw.p, n, sel.p, pos.p, cand.p, k_span.p, k_match.p, k_seed.p, R.st);

		SortOut so;
		sort_and_filter(R, cand.p, k_span.p, k_match.p, k_seed.p, nc, d_seed.p, seed_len, n_ids, min_seed_len, max_bin_cov, max_flank_len, hq_mode, so);
		HIP_OK(hipEventRecord(ev1, R.st));
		HIP_OK(hipStreamSynchronize(R.st));
		const uint64_t total = so.recs.size(), n_seeds = so.seeds;
		const double t_dev = since();
		hand_out(so, out, bl_id, bl_kind, n_bl);
		if (prof) fprintf(stderr, "[ndgpu_ovl_sort] %llu records: set-up %.2f ms, uploads + kernels + downloads %.2f ms, hand-out %.2f ms\n",
		                  (unsigned long long)n, t_setup, t_dev - t_setup, since() - t_dev);
		if (stats) {
			float ms = 0;
			(void)hipEventElapsedTime(&ms, ev0, ev1);
			stats->gpu_ms = ms, stats->raw_records = n, stats->candidates = nc, stats->seeds = n_seeds, stats->kept = total, stats->ranges = 1;
		}
		(void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
		return (int64_t)total;
	} catch (...) {
		return -2;
	}
}

extern "C" int64_t ndgpu_ovl_sort(const ndgpu_ovl_rec *const *files, const int64_t *n_per_file, int32_t n_files, const uint32_t *seed_len,
                                  uint32_t n_ids, int32_t min_seed_len, int32_t max_bin_cov, int32_t max_flank_len, ndgpu_ovl_rec **out,
                                  uint32_t **bl_id, uint8_t **bl_kind, int64_t *n_bl, ndgpu_ovl_sort_stats *stats)
{
	return sort_impl(files, n_per_file, n_files, seed_len, n_ids, min_seed_len, max_bin_cov, max_flank_len, out, bl_id, bl_kind, n_bl, stats, false);
}

extern "C" int64_t ndgpu_ovl_sort_hq(const ndgpu_ovl_rec *const *files, const int64_t *n_per_file, int32_t n_files, const uint32_t *seed_len,
                                     uint32_t n_ids, int32_t min_seed_len, int32_t max_bin_cov, int32_t max_flank_len, ndgpu_ovl_rec **out,
                                     uint32_t **bl_id, uint8_t **bl_kind, int64_t *n_bl, ndgpu_ovl_sort_stats *stats)
{
	return sort_impl(files, n_per_file, n_files, seed_len, n_ids, min_seed_len, max_bin_cov, max_flank_len, out, bl_id, bl_kind, n_bl, stats, true);
}


// Pile admission of lib/nextcorrect.py:92-143 (read_seq_data) over the records of a sorted.ovl, in file order -- host
// logic, one pass, no device work: what the stage CLI and the fused stage feed to ndgpu_correct_piles.
extern "C" int64_t ndgpu_assemble_piles(const ndgpu_ovl_rec *sorted, int64_t n, uint32_t n_ids, uint32_t min_len_seed, uint32_t min_len_aln,
                                        uint32_t max_cov_aln, uint32_t min_cov_seed, const uint32_t *skip_ids, int64_t n_skip, uint32_t **recs8,
                                        uint64_t **pile_off, uint32_t **seeds, int64_t *n_piles)
{
	*recs8 = nullptr, *pile_off = nullptr, *seeds = nullptr, *n_piles = 0;
	std::vector<uint8_t> skip(n_ids ? n_ids : 1, 0);
	for (int64_t i = 0; i < n_skip; ++i) if (skip_ids[i] < n_ids) skip[skip_ids[i]] = 1;
	std::vector<uint32_t> used(n_ids ? n_ids : 1, 0); // stamp = group counter of the last pile that admitted the read
	std::vector<uint32_t> out;
	std::vector<uint64_t> off{0};
	std::vector<uint32_t> names;
	out.reserve((size_t)n * 8);
	const double lim = (double)max_cov_aln * 1.5;
	uint32_t stamp = 1;
	// state of the reference's loop: seed_state 0 = '' (no seed yet), 1 = valid seed, 2 = '+' (rejected seed)
	int seed_state = 0;
	uint32_t seed_name = 0;
	uint64_t total_length = 0, seed_length = 0;
	int64_t last_seed = -1;
	size_t pile_start = 0;
	auto close_pile = [&](bool keep) {
		if (keep) { off.push_back(out.size() / 8); names.push_back(seed_name); }
		else out.resize(pile_start * 8);
		pile_start = out.size() / 8;
		++stamp;
	};
	for (int64_t k = 0; k < n; ++k) {
		const ndgpu_ovl_rec &r = sorted[k]; // qname = the seed (field 0 of a sorted.ovl record), tname = the other read
		const uint32_t t_name = r.qname, t_s = r.qs, t_e = r.qe, q_name = r.tname;
		if (seed_state == 2 || (last_seed != -1 && (int64_t)t_name != last_seed)) {
			close_pile(seed_length && (double)total_length / (double)seed_length >= (double)min_cov_seed && seed_state == 1);
			seed_state = 0, total_length = seed_length = 0;
		}
		if (seed_state == 0) {
			seed_length = (uint64_t)t_e + 1;
			total_length = 0;
			seed_name = t_name;
			seed_state = (seed_length >= min_len_seed && !(t_name < n_ids && skip[t_name])) ? 1 : 2;
		}
		if (t_e - t_s < min_len_aln || (double)total_length / (double)seed_length > lim || (q_name < n_ids && used[q_name] == stamp) || seed_state == 2)
			continue;
		const uint32_t row[8] = {r.qname, r.rev, r.qs, r.qe, r.tname, r.ts, r.te, r.match};
		out.insert(out.end(), row, row + 8);
		if (q_name < n_ids) used[q_name] = stamp;
		total_length += (uint64_t)(t_e - t_s) + 1;
		last_seed = (int64_t)t_name;
	}
	close_pile(seed_length && (double)total_length / (double)seed_length >= (double)min_cov_seed && seed_state == 1);
	const size_t np = names.size(), nr = off.back();
	*recs8 = (uint32_t*)malloc(sizeof(uint32_t) * 8 * (nr ? nr : 1));
	*pile_off = (uint64_t*)malloc(sizeof(uint64_t) * (np + 1));
	*seeds = (uint32_t*)malloc(sizeof(uint32_t) * (np ? np : 1));
	if (nr) memcpy(*recs8, out.data(), sizeof(uint32_t) * 8 * nr);
	memcpy(*pile_off, off.data(), sizeof(uint64_t) * (np + 1));
	if (np) memcpy(*seeds, names.data(), sizeof(uint32_t) * np);
	*n_piles = (int64_t)np;
	return (int64_t)nr;
}
