// ovl_engine.hip -- host orchestration + C ABI of the overlap engine (include/ndgpu_overlap.h).
//
// Data flow of one `minimap2-nd --step 1 target query` run (reference: minimap2/main.c:474-507):
//   index_create : .2bit words -> HBM, K1 sketch (count, scan, fill), K2 two LSD sorts (position, then
//                  minimizer) + run-length encode -> (ukey, ustart, pos) resident in HBM
//   map          : K1 on the query set, K3a lookup/count over all query minimizers, then per batch of
//                  query reads (bounded by an anchor budget): K3b fill, K3s sort (+ exact replay for reads
//                  with equal keys), K4 chain DP, K5 hits, compaction, D2H of the overlap records
//   encode       : serial delta + varint coding of the records on the host (lib/ovl.c:109-150)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include <atomic>

#include "../../include/ndgpu_overlap.h"
#include "ovl_device.h"
#include "ovl_pool.h"

#include <map>
#include <chrono>
#include <mutex>
#include <thread>
#include <exception>
#include <unordered_map>

namespace ndovl {

// ---- device block pool (ovl_pool.h) ----
// Slabs taken from the driver (hipMalloc) and carved up here: best-fit free ranges, split on allocation, merged with their
// neighbours on release.  Until round 4 the pool cached whole blocks by size class and gave back to the driver what exceeded 1.25 x
// its working set: every query batch asks for slightly different sizes, so a genome-scale job went to the driver 2,205 times a
// step -- 20 s of a 27 s overlap stage on a device whose memory the consensus contexts mostly hold (config 3, profiles/r04), and
// each hipFree waits for every stream of the device.  A range allocator serves any size from what it holds: in steady state no
// step goes to the driver at all.
namespace {
std::atomic<int> g_last_error{0};  // why the last failed entry point failed: 1 = out of device memory, 2 = anything else
std::mutex g_pool_mu;
struct Range { size_t size; int slab; };
std::map<uintptr_t, Range> g_free_at;                  // free ranges by address
std::multimap<size_t, uintptr_t> g_free_by_size;       // and by size (best fit)
std::unordered_map<uintptr_t, Range> g_live_at;        // handed-out ranges
struct Slab { void *base; size_t size; size_t live; };
std::vector<Slab> g_slabs;                             // (released slabs keep their index: base == nullptr)
size_t g_pool_slab_bytes = 0, g_pool_live = 0, g_pool_peak = 0;
std::atomic<unsigned long long> g_pool_calls{0}, g_pool_ns{0};  // hipMalloc / hipFree calls the pool made and their wall time
struct PoolTimer {
	std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
	~PoolTimer() { g_pool_calls++; g_pool_ns += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
constexpr size_t kPoolAlign = 512;
void erase_by_size(size_t size, uintptr_t at)
{
	auto r = g_free_by_size.equal_range(size);
	for (auto it = r.first; it != r.second; ++it)
		if (it->second == at) { g_free_by_size.erase(it); return; }
}
void add_free(uintptr_t at, size_t size, int slab) // merges with the free neighbours of the same slab
{
	auto nx = g_free_at.lower_bound(at);
	if (nx != g_free_at.end() && nx->second.slab == slab && at + size == nx->first) {
		size += nx->second.size;
		erase_by_size(nx->second.size, nx->first);
		nx = g_free_at.erase(nx);
	}
	if (nx != g_free_at.begin()) {
		auto pv = std::prev(nx);
		if (pv->second.slab == slab && pv->first + pv->second.size == at) {
			at = pv->first;
			size += pv->second.size;
			erase_by_size(pv->second.size, pv->first);
			g_free_at.erase(pv);
		}
	}
	g_free_at[at] = Range{size, slab};
	g_free_by_size.emplace(size, at);
}
size_t release_idle_slabs() // hipFree of every slab nothing lives in; returns the bytes released
{
	size_t freed = 0;
	for (size_t k = 0; k < g_slabs.size(); ++k) {
		Slab &sl = g_slabs[k];
		if (!sl.base || sl.live) continue;
		const uintptr_t at = (uintptr_t)sl.base;
		auto it = g_free_at.find(at);
		if (it == g_free_at.end() || it->second.size != sl.size) continue; // (cannot happen: an idle slab is one free range)
		erase_by_size(it->second.size, at);
		g_free_at.erase(it);
		const PoolTimer timer;
		(void)hipFree(sl.base);
		g_pool_slab_bytes -= sl.size;
		freed += sl.size;
		sl.base = nullptr;
	}
	return freed;
}
}

void *pool_alloc(size_t bytes)
{
	const size_t c = (std::max<size_t>(bytes, 1) + kPoolAlign - 1) / kPoolAlign * kPoolAlign;
	if (fault_injected()) device_check((int)hipErrorOutOfMemory, "pool_alloc (injected)");
	std::lock_guard<std::mutex> g(g_pool_mu);
	auto it = g_free_by_size.lower_bound(c);
	if (it == g_free_by_size.end()) {
		// a new slab: large enough that the requests to come are carved out of it, not bigger than the device can give
		// (under the kernel interpreter of tests/simt a slab is the request itself unless a slab size is given: the interpreter's
		// allocations are exact-size host blocks, which is what lets AddressSanitizer see a kernel read past a buffer's end)
#ifdef SIMT_EMULATION
		const size_t unit = getenv("NDGPU_OVL_SLAB_MB") ? (size_t)atol(getenv("NDGPU_OVL_SLAB_MB")) << 20 : 0;
#else
		// (8 GB on a 288 GB MI355X; a sixteenth of the memory on a smaller part, so that one long-lived block -- the resident read
		// words -- does not pin an eighth of the device in its slab)
		static const size_t dev_unit = [] {
			size_t free_b = 0, total_b = 0;
			if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || !total_b) return (size_t)8 << 30;
			return std::max<size_t>((size_t)256 << 20, std::min<size_t>((size_t)8 << 30, total_b / 16));
		}();
		const size_t unit = getenv("NDGPU_OVL_SLAB_MB") ? (size_t)atol(getenv("NDGPU_OVL_SLAB_MB")) << 20 : dev_unit;
#endif
		size_t want = std::max(c, unit);
		void *p = nullptr;
		hipError_t e;
		{
			const PoolTimer timer;
			e = hipMalloc(&p, want);
		}
		if (e != hipSuccess && want > c) { // the device is short: what this request needs and no more
			(void)hipGetLastError();
			want = c;
			const PoolTimer timer;
			e = hipMalloc(&p, want);
		}
		if (e != hipSuccess) { // give idle slabs back and try once more
			(void)hipGetLastError();
			if (release_idle_slabs()) {
				const PoolTimer timer;
				e = hipMalloc(&p, want);
			}
		}
		if (e != hipSuccess) {
			fprintf(stderr, "[ndgpu_overlap] hipMalloc of %zu bytes failed: %s (pool: %zu bytes in slabs, %zu live)\n", want, hipGetErrorString(e),
			        g_pool_slab_bytes, g_pool_live);
			g_last_error = e == hipErrorOutOfMemory ? 1 : 2;
			(void)hipGetLastError();
			throw std::runtime_error("hipMalloc");
		}
		g_slabs.push_back(Slab{p, want, 0});
		g_pool_slab_bytes += want;
		add_free((uintptr_t)p, want, (int)g_slabs.size() - 1);
		it = g_free_by_size.lower_bound(c);
	}
	const uintptr_t at = it->second;
	const Range r = g_free_at[at];
	g_free_by_size.erase(it);
	g_free_at.erase(at);
	if (r.size > c) add_free(at + c, r.size - c, r.slab);
	g_live_at[at] = Range{c, r.slab};
	g_slabs[(size_t)r.slab].live += c;
	g_pool_live += c;
	if (g_pool_live > g_pool_peak) g_pool_peak = g_pool_live;
	return (void*)at;
}

int last_error_take() { return g_last_error.exchange(0); }
void note_oom() { g_last_error = 1; }

void device_check(int hip_error, const char *what)
{
	if (hip_error == (int)hipSuccess) return;
	fprintf(stderr, "[ndgpu_overlap] %s failed: %s\n", what, hipGetErrorString((hipError_t)hip_error));
	g_last_error = hip_error == (int)hipErrorOutOfMemory ? 1 : 2;
	(void)hipGetLastError();
	throw std::runtime_error(what);
}

bool fault_injected()
{
	static std::mutex mu;
	static unsigned long long n_ops = 0, target = 0;
	const char *e = getenv("NDGPU_OVL_FAIL_AT");
	if (!e) { if (target) { std::lock_guard<std::mutex> g(mu); target = 0; } return false; }
	const unsigned long long t = strtoull(e, nullptr, 10);
	std::lock_guard<std::mutex> g(mu);
	if (t != target) target = t, n_ops = 0;  // a new value: the count starts again
	return ++n_ops == target;
}

void pool_free(void *p)
{
	if (!p) return;
	std::lock_guard<std::mutex> g(g_pool_mu);
	auto it = g_live_at.find((uintptr_t)p);
	if (it == g_live_at.end()) { (void)hipFree(p); return; }
	const Range r = it->second;
	g_live_at.erase(it);
	g_pool_live -= r.size;
	g_slabs[(size_t)r.slab].live -= r.size;
	add_free((uintptr_t)p, r.size, r.slab);
	// (NDGPU_OVL_POOL_GB: a bound on what the pool keeps from the driver -- idle slabs beyond it go back)
	static const size_t cap = getenv("NDGPU_OVL_POOL_GB") ? (size_t)(atof(getenv("NDGPU_OVL_POOL_GB")) * (double)(1ull << 30)) : ~(size_t)0;
	if (g_pool_slab_bytes > cap) (void)release_idle_slabs();
#ifdef SIMT_EMULATION
	if (!getenv("NDGPU_OVL_SLAB_MB")) (void)release_idle_slabs();  // (exact-size blocks go back when they are free: use-after-free is seen)
#endif
}

size_t pool_trim() // the bytes given back to the driver (idle slabs only: a slab that holds a live block stays)
{
	std::lock_guard<std::mutex> g(g_pool_mu);
	return release_idle_slabs();
}

// what the pool holds beyond what is in use (an upper bound of what pool_trim() can give back: idle slabs only)
size_t pool_cached_bytes() { std::lock_guard<std::mutex> g(g_pool_mu); return g_pool_slab_bytes - g_pool_live; }
void pool_bytes(uint64_t out[3]) { std::lock_guard<std::mutex> g(g_pool_mu); out[0] = g_pool_live, out[1] = g_pool_slab_bytes - g_pool_live, out[2] = g_pool_peak; }
void pool_calls(uint64_t out[2], int reset) { out[0] = g_pool_calls.load(), out[1] = g_pool_ns.load(); if (reset) g_pool_calls = 0, g_pool_ns = 0; }

// (a device filled to the brim makes the runtime's own allocations fail too -- launch arguments, staging: "out of memory" may
// surface at any call; it is reported as what it is, so that the caller can release memory and try again)
#define HIP_OK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "[ndgpu_overlap] HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); if (_e == hipErrorOutOfMemory) { ndovl::note_oom(); (void)hipGetLastError(); } throw std::runtime_error("hip"); } } while (0)

template <class T> struct DevBuf {
	T *p = nullptr;
	size_t n = 0;
	DevBuf() = default;
	explicit DevBuf(size_t count) { alloc(count); }
	DevBuf(const DevBuf&) = delete;
	DevBuf &operator=(const DevBuf&) = delete;
	DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr, o.n = 0; }
	DevBuf &operator=(DevBuf &&o) noexcept { if (this != &o) { release(); p = o.p, n = o.n; o.p = nullptr, o.n = 0; } return *this; }
	~DevBuf() { release(); }
	void alloc(size_t count) { release(); n = count; if (count) p = (T*)pool_alloc(count * sizeof(T)); }
	void release() { if (p) pool_free(p); p = nullptr, n = 0; }
	void upload(const T *src, size_t count, hipStream_t s) { if (count) HIP_OK(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s)); }
	void download(T *dst, size_t count, hipStream_t s, size_t from = 0) const { if (count) HIP_OK(hipMemcpyAsync(dst, p + from, count * sizeof(T), hipMemcpyDeviceToHost, s)); }
	void zero(hipStream_t s) { if (n) HIP_OK(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
};

// strcmp() order of "%u" names as an integer key: digits left-aligned to 10 places, length as tie-break
static uint64_t name_key(uint32_t id)
{
	static const uint64_t p10[11] = {1ULL, 10ULL, 100ULL, 1000ULL, 10000ULL, 100000ULL, 1000000ULL, 10000000ULL, 100000000ULL, 1000000000ULL,
	                                 10000000000ULL};
	int nd = 1;
	while (nd < 10 && (uint64_t)id >= p10[nd]) ++nd;
	return ((uint64_t)id * p10[10 - nd]) << 4 | (uint64_t)nd;
}

static uint32_t wang32(uint32_t key)
{
	key += ~(key << 15); key ^= key >> 10; key += key << 3;
	key ^= key >> 6; key += ~(key << 11); key ^= key >> 16;
	return key;
}

// per-read hash that orders a read's hits (minimap2/map.c:519-521): X31 over the decimal name
static uint32_t read_hash(uint32_t id, uint32_t qlen, int seed)
{
	char name[16];
	snprintf(name, sizeof(name), "%u", id);
	uint32_t h = (uint32_t)name[0];
	for (const char *s = name + 1; *s; ++s) h = (h << 5) - h + (uint32_t)*s;
	h ^= wang32(qlen) + wang32((uint32_t)seed);
	return wang32(h);
}

static unsigned bits_for(uint64_t max_value) // bits needed to hold values 0..max_value
{
	unsigned b = 1;
	while (b < 64 && (max_value >> b)) ++b;
	return b;
}

struct EvTimer {
	hipEvent_t a, b;
	hipStream_t s;
	EvTimer(hipStream_t st) : s(st) { HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b)); }
	~EvTimer() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
	void start() { HIP_OK(hipEventRecord(a, s)); }
	double stop() { HIP_OK(hipEventRecord(b, s)); HIP_OK(hipEventSynchronize(b)); float ms = 0; HIP_OK(hipEventElapsedTime(&ms, a, b)); return ms; }
};

// Resident read words (ndgpu_ovl_words_resident): a host buffer of 2-bit words the caller has declared resident is uploaded once;
// every read set that names words inside it afterwards works on the device copy.  The stage maps the same reads step after step and job
// after job -- index build and query side of every job upload their words, 2 x 57.5 MB per config-2 step from pageable memory -- and
// the metric is defined with the inputs resident in HBM.
struct ResidentWords { const uint32_t *host; uint64_t n_words; uint32_t *dev; };
static std::mutex g_res_mu;
static std::vector<ResidentWords> g_resident;
static const uint32_t *resident_view(const uint32_t *w, uint64_t n_words)
{
	std::lock_guard<std::mutex> g(g_res_mu);
	for (const ResidentWords &r : g_resident)
		if (w >= r.host && w + n_words <= r.host + r.n_words) return r.dev + (w - r.host);
	return nullptr;
}

struct ReadSetDev {
	uint32_t n = 0;
	uint64_t bases = 0;
	uint32_t max_len = 0;
	DevBuf<uint32_t> words, len, id, order;
	const uint32_t *wp = nullptr;   // the read words on the device: `words`, or a view into a resident copy
	DevBuf<uint64_t> woff, namekey;
	std::vector<uint32_t> h_len;
	void upload(uint32_t n_reads, const uint32_t *w, uint64_t n_words, const uint64_t *off, const uint32_t *lens, const uint32_t *ids,
	            hipStream_t s)
	{
		n = n_reads;
		h_len.assign(lens, lens + n_reads);
		wp = resident_view(w, n_words);
		if (!wp) { words.alloc(n_words + 4); words.upload(w, n_words, s); wp = words.p; }
		woff.alloc(n); woff.upload(off, n, s);
		len.alloc(n); len.upload(lens, n, s);
		id.alloc(n); id.upload(ids, n, s);
		std::vector<uint64_t> nk(n);
		std::vector<uint32_t> ord(n);
		bases = 0, max_len = 0;
		for (uint32_t i = 0; i < n; ++i) { nk[i] = name_key(ids[i]); ord[i] = i; bases += lens[i]; max_len = std::max(max_len, lens[i]); }
		std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return lens[a] > lens[b]; });
		namekey.alloc(n); namekey.upload(nk.data(), n, s);
		order.alloc(n); order.upload(ord.data(), n, s);
		HIP_OK(hipStreamSynchronize(s));
	}
};

struct Sketch { // minimizers of a read set
	uint64_t n = 0;
	DevBuf<uint64_t> x, y, off; // off: n_reads + 1
	DevBuf<uint32_t> read;
};

// the passes of --step 2's re-alignment (ndgpu_ovl_map_regs): the hits themselves, per query read, nothing judged
struct Regs {
	const uint64_t *want_off;       // nullptr: every read of the index is a target
	const uint32_t *want;
	bool nameless;                  // mm_map(..., qname = 0)
	std::vector<uint32_t> *counts;  // hits of every query read
	uint64_t *max_anchors;          // (may be null) the most anchors any one read had
	// -c (ndgpu_ovl_map_chains): the hits are chains -- OvlRec = (strand, target, offset into the read's anchors, anchor count, chain
	// score, hash, 0, 0) -- and the chained anchors of every read come with them, in the reference's a[] order; null otherwise
	std::vector<uint64_t> *ca_x = nullptr, *ca_y = nullptr, *ca_off = nullptr;
	bool thin = false;              // chain with mm_chain_dp_nextdenovo (anchor thinning beyond 100,000 anchors)
};

struct Engine {
	int device = 0;
	hipStream_t stream = nullptr;
	OvlParams P{};
	ReadSetDev T;
	uint64_t n_min = 0, n_keys = 0;
	DevBuf<uint64_t> ukey, ustart, pos;
	DevBuf<uint32_t> ucnt, bucket;
	uint32_t bucket_shift = 0;
	DevBuf<HashSlot> htab;   // the lookup table (build_hash: from the index's second map call on)
	uint64_t hsize = 0;
	uint32_t maps_served = 0;
	double query_minimizers_seen = 0;   // estimate (2 / (w + 1) per base) over the map calls of this index: what decides the lookup table
	ndgpu_ovl_stats st{};
	// debug view of the last map batch
	std::vector<uint64_t> dbg_aoff;
	DevBuf<uint64_t> dbg_ax, dbg_ay;
	DevBuf<int32_t> dbg_f, dbg_p;
	uint32_t dbg_r0 = 0, dbg_n = 0;

	DevBuf<uint8_t> tmp;
	std::vector<hipStream_t> lane_streams;   // the streams of map()'s concurrent batches (created on first use)
	void *temp(size_t bytes) { if (tmp.n < bytes) tmp.alloc(bytes + bytes / 4); return tmp.p; }

	IndexDev index_dev() const { return IndexDev{n_keys, ukey.p, ustart.p, pos.p, T.len.p, T.id.p, T.namekey.p, bucket.p, bucket_shift, htab.p, hsize}; }

	// tiles of `tile` symbols over reads whose symbol counts are n_sym[]; first[r] = first tile of read r
	// (of the next non-empty read for an empty one), first[n] = number of tiles
	static void make_tiles(const uint32_t *n_sym, uint32_t n, uint32_t tile, std::vector<SketchTile> &tiles, std::vector<uint32_t> &first)
	{
		tiles.clear();
		first.assign((size_t)n + 1, 0);
		for (uint32_t r = 0; r < n; ++r) {
			first[r] = (uint32_t)tiles.size();
			for (uint32_t s = 0; s < n_sym[r]; s += tile) tiles.push_back(SketchTile{r, s});
		}
		first[n] = (uint32_t)tiles.size();
	}

	void exscan(const uint32_t *in, uint64_t *out, size_t n)
	{
		size_t tb = 0;
		exscan_u32_to_u64(nullptr, tb, in, out, n, stream);
		exscan_u32_to_u64(temp(tb), tb, in, out, n, stream);
	}

	// K1, position-parallel form (odd k)
	void sketch_tiled(const ReadSetDev &R, int rid_is_index, bool want_read, Sketch &out)
	{
		const uint32_t TS = (uint32_t)sketch_tile_symbols();
		EvTimer tm(stream);
		tm.start();
		std::vector<SketchTile> tiles;
		std::vector<uint32_t> first;
		DevBuf<uint8_t> sym;
		DevBuf<uint32_t> rstart, n_sym_d;
		DevBuf<uint64_t> roff;
		std::vector<uint32_t> h_nsym;
		if (P.hpc) {
			make_tiles(R.h_len.data(), R.n, TS, tiles, first);
			const uint32_t nt = (uint32_t)tiles.size();
			DevBuf<SketchTile> d_tiles(nt + 1);
			d_tiles.upload(tiles.data(), nt, stream);
			DevBuf<uint32_t> d_first(R.n + 1), cnt(nt + 1);
			d_first.upload(first.data(), R.n + 1, stream);
			cnt.zero(stream);
			std::vector<uint64_t> h_roff(R.n + 1);
			uint64_t acc = 0;
			for (uint32_t r = 0; r < R.n; ++r) { h_roff[r] = acc + r; acc += R.h_len[r]; }
			h_roff[R.n] = acc + R.n;
			roff.alloc(R.n + 1); roff.upload(h_roff.data(), R.n + 1, stream);
			sym.alloc(acc + R.n + 1); rstart.alloc(acc + R.n + 1); n_sym_d.alloc(R.n + 1);
			n_sym_d.zero(stream);
			launch_run_compact(false, R.wp, R.woff.p, R.len.p, d_tiles.p, nt, nullptr, d_first.p, roff.p, cnt.p, sym.p, rstart.p,
			                   n_sym_d.p, stream);
			DevBuf<uint64_t> prefix(nt + 1);
			exscan(cnt.p, prefix.p, nt + 1);
			launch_run_compact(true, R.wp, R.woff.p, R.len.p, d_tiles.p, nt, prefix.p, d_first.p, roff.p, cnt.p, sym.p, rstart.p,
			                   n_sym_d.p, stream);
			h_nsym.resize(R.n);
			n_sym_d.download(h_nsym.data(), R.n, stream);
			HIP_OK(hipStreamSynchronize(stream));
			HIP_OK(hipGetLastError());
			make_tiles(h_nsym.data(), R.n, TS, tiles, first);
		} else make_tiles(R.h_len.data(), R.n, TS, tiles, first);
		const uint32_t nt = (uint32_t)tiles.size();
		DevBuf<SketchTile> d_tiles(nt + 1);
		d_tiles.upload(tiles.data(), nt, stream);
		DevBuf<uint32_t> d_first(R.n + 1), cnt(nt + 1);
		d_first.upload(first.data(), R.n + 1, stream);
		cnt.zero(stream);
		launch_sketch_tiles(false, P.hpc != 0, R.wp, R.woff.p, R.len.p, sym.p, rstart.p, roff.p, n_sym_d.p, d_tiles.p, nt, P, rid_is_index,
		                    nullptr, cnt.p, nullptr, nullptr, nullptr, stream);
		DevBuf<uint64_t> tile_off(nt + 1);
		exscan(cnt.p, tile_off.p, nt + 1);
		uint64_t total = 0;
		tile_off.download(&total, 1, stream, nt);
		HIP_OK(hipStreamSynchronize(stream));
		out.n = total;
		out.x.alloc(total + 1); out.y.alloc(total + 1);
		if (want_read) out.read.alloc(total + 1);
		launch_sketch_tiles(true, P.hpc != 0, R.wp, R.woff.p, R.len.p, sym.p, rstart.p, roff.p, n_sym_d.p, d_tiles.p, nt, P, rid_is_index,
		                    tile_off.p, cnt.p, out.x.p, out.y.p, want_read ? out.read.p : nullptr, stream);
		out.off.alloc(R.n + 1);
		launch_gather_u64(tile_off.p, d_first.p, R.n + 1, out.off.p, stream);
		HIP_OK(hipGetLastError());
		st.sketch_ms += tm.stop();
		st.bases_sketched += 2 * R.bases;
		st.minimizers += total;
	}

	void sketch(const ReadSetDev &R, int rid_is_index, bool want_read, Sketch &out)
	{
		// (the position-parallel kernels take an odd k of one word up to 28, or of two words: 33..63)
		if ((P.k & 1) && (P.k <= 28 || (P.k >= 33 && P.k <= 63)) && !getenv("NDGPU_OVL_SEQ_SKETCH")) { sketch_tiled(R, rid_is_index, want_read, out); return; }
		DevBuf<uint32_t> cnt(R.n + 1);
		cnt.zero(stream);
		EvTimer tm(stream);
		tm.start();
		launch_sketch(false, R.wp, R.woff.p, R.len.p, R.order.p, R.n, P, rid_is_index, nullptr, nullptr, nullptr, nullptr, cnt.p, stream);
		out.off.alloc(R.n + 1);
		size_t tb = 0;
		exscan_u32_to_u64(nullptr, tb, cnt.p, out.off.p, R.n + 1, stream);
		exscan_u32_to_u64(temp(tb), tb, cnt.p, out.off.p, R.n + 1, stream);
		uint64_t total = 0;
		out.off.download(&total, 1, stream, R.n);
		HIP_OK(hipStreamSynchronize(stream));
		out.n = total;
		out.x.alloc(total + 1); out.y.alloc(total + 1);
		if (want_read) out.read.alloc(total + 1);
		launch_sketch(true, R.wp, R.woff.p, R.len.p, R.order.p, R.n, P, rid_is_index, out.off.p, out.x.p, out.y.p,
		              want_read ? out.read.p : nullptr, nullptr, stream);
		HIP_OK(hipGetLastError());
		st.sketch_ms += tm.stop();
		st.bases_sketched += 2 * R.bases; // count pass + fill pass
		st.minimizers += total;
	}

	void build_index()
	{
		Sketch S;
		sketch(T, 1, false, S);
		n_min = S.n;
		EvTimer tm(stream);
		tm.start();
		DevBuf<uint64_t> key(n_min + 1), key2(n_min + 1), y2(n_min + 1);
		launch_shift_keys(S.x.p, key.p, n_min, stream);
		// by position first (read<<32 | pos<<1 | strand), then stably by minimizer
		const unsigned ybits = 32 + bits_for(T.n ? T.n - 1 : 0);
		size_t tb = 0;
		sort_pairs_u64(nullptr, tb, S.y.p, y2.p, key.p, key2.p, n_min, 0, ybits, stream);
		if (n_min) sort_pairs_u64(temp(tb), tb, S.y.p, y2.p, key.p, key2.p, n_min, 0, ybits, stream);
		pos.alloc(n_min + 1);
		const unsigned kbits = std::min(56u, 2u * (unsigned)P.k); // x >> 8: the long k-mer hash of ava-hifi fills all 56 bits
		tb = 0;
		sort_pairs_u64(nullptr, tb, key2.p, key.p, y2.p, pos.p, n_min, 0, kbits, stream);
		if (n_min) sort_pairs_u64(temp(tb), tb, key2.p, key.p, y2.p, pos.p, n_min, 0, kbits, stream);
		// distinct minimizers and their occurrence counts
		DevBuf<uint64_t> uniq(n_min + 1), n_runs(1);
		ucnt.alloc(n_min + 2);
		ucnt.zero(stream);
		n_keys = 0;
		if (n_min) {
			tb = 0;
			rle_u64(nullptr, tb, key.p, n_min, uniq.p, ucnt.p, n_runs.p, stream);
			rle_u64(temp(tb), tb, key.p, n_min, uniq.p, ucnt.p, n_runs.p, stream);
			n_runs.download(&n_keys, 1, stream);
			HIP_OK(hipStreamSynchronize(stream));
		}
		ukey.alloc(n_keys + 1);
		if (n_keys) HIP_OK(hipMemcpyAsync(ukey.p, uniq.p, n_keys * 8, hipMemcpyDeviceToDevice, stream));
		ustart.alloc(n_keys + 2);
		tb = 0;
		exscan_u32_to_u64(nullptr, tb, ucnt.p, ustart.p, n_keys + 1, stream);
		exscan_u32_to_u64(temp(tb), tb, ucnt.p, ustart.p, n_keys + 1, stream);
		// top-bits table over the distinct keys (hash values have 2k bits)
		const unsigned key_bits = std::min(56u, 2u * (unsigned)P.k);
		bucket_shift = key_bits > (unsigned)kBucketBits ? key_bits - (unsigned)kBucketBits : 0;
		bucket.alloc(((size_t)1 << kBucketBits) + 2);
		launch_build_buckets(ukey.p, n_keys, bucket_shift, bucket.p, stream);
		HIP_OK(hipGetLastError());
		htab.release();
		hsize = 0, maps_served = 0, hash_skipped = false, query_minimizers_seen = 0;
		st.index_sort_ms += tm.stop();
	}

	// The lookup table over the distinct keys: two thirds full, 16 bytes a slot.  Measured on config 2 (r5_10): the seed pass of a map
	// call 13.5 -> 9.6 ms, the table's build 6.7 ms (a power-of-two table at most half full; smaller since) -- built when the query
	// minimizers it serves outweigh that (map_once: by cost, not by call count).  Skipped when it would take
	// more than a sixteenth of the device -- a decision taken once per index (the memory query goes through the runtime's device
	// enumeration: milliseconds) -- and when its block cannot be had: the table is a short cut, the bucket search answers the same
	// lookups, so a map call that would succeed without it must not fail for it.
	bool hash_skipped = false;
	void build_hash()
	{
		if (htab.p || hash_skipped || !n_keys || getenv("NDGPU_OVL_NO_HASH")) return;
		const uint64_t slots = n_keys + n_keys / 2 + 64;
		size_t free_b = 0, total_b = 0;
		if (slots >= (1ull << 32) || (hipMemGetInfo(&free_b, &total_b) == hipSuccess && slots * sizeof(HashSlot) > total_b / 16)) {
			hash_skipped = true;
			return;
		}
		EvTimer tm(stream);
		tm.start();
		try {
			htab.alloc(slots);
		} catch (const std::runtime_error &) {   // out of device memory (or an injected failure): no table
			(void)ndovl::last_error_take();
			hash_skipped = true;
			return;
		}
		htab.zero(stream);
		hsize = slots;
		launch_build_hash(ukey.p, ustart.p, n_keys, htab.p, hsize, stream);
		HIP_OK(hipGetLastError());
		st.index_sort_ms += tm.stop();
	}

	int32_t mid_occ(float f)
	{
		if (f <= 0.) return INT32_MAX;
		if (!n_keys) return 1;
		DevBuf<uint32_t> sorted(n_keys);
		size_t tb = 0;
		sort_keys_u32(nullptr, tb, ucnt.p, sorted.p, n_keys, stream);
		sort_keys_u32(temp(tb), tb, ucnt.p, sorted.p, n_keys, stream);
		const uint32_t kth = (uint32_t)((1. - f) * n_keys);
		uint32_t v = 0;
		sorted.download(&v, 1, stream, kth);
		HIP_OK(hipStreamSynchronize(stream));
		return (int32_t)(v + 1);
	}

	// One pass over the query set.  read_mid (one occurrence threshold per query read) and chains_per_read (what the pass found)
	// belong to map_rechain() below.
	const int32_t *read_mid = nullptr;
	std::vector<uint32_t> *chains_per_read = nullptr;
	std::vector<uint32_t> *rep_per_read = nullptr;   // (with chains_per_read) != 0: a minimizer of the read was dropped for its occurrences
	int64_t map_once(const ndgpu_ovl_opt &o, int32_t mid, uint32_t n_q, const uint32_t *words, uint64_t n_words, const uint64_t *woff,
	                 const uint32_t *lens, const uint32_t *ids, std::vector<OvlRec> &out, std::vector<OvlRec10> *out10 = nullptr,
	                 const Regs *regs = nullptr);
	// -f FLOAT,INT (mm_mapopt_t::max_occ > mid_occ; minimap2/map.c:553-575 and :678-700): a query read that ends its chaining without a
	// chain is seeded again with every minimizer below max_occ occurrences and chained again (with one segment per query that is the
	// test together with `rep_len > 0`: a read none of whose minimizers was dropped would collect the same seeds again -- and in the
	// one-read-index mappings of --step 2 --mode 1, whose second chaining skips the anchor thinning, must not be chained again at all:
	// ADVICE round 5).  Two passes over the query set: the
	// first tells which reads found no chain, the second maps every read with ITS threshold -- the reads of a batch are independent,
	// so the second pass's records are the reference's, in its order, whatever the mode of the call (records, hits, chains).
	int64_t map(const ndgpu_ovl_opt &o, int32_t mid, uint32_t n_q, const uint32_t *words, uint64_t n_words, const uint64_t *woff,
	            const uint32_t *lens, const uint32_t *ids, std::vector<OvlRec> &out, std::vector<OvlRec10> *out10 = nullptr,
	            const Regs *regs = nullptr)
	{
		if (o.max_occ <= mid) return map_once(o, mid, n_q, words, n_words, woff, lens, ids, out, out10, regs);
		std::vector<uint32_t> chains, rep;
		struct Unset { Engine *e; ~Unset() { e->read_mid = nullptr, e->chains_per_read = nullptr, e->rep_per_read = nullptr; } } unset{this};
		chains_per_read = &chains;
		rep_per_read = &rep;
		int64_t n = map_once(o, mid, n_q, words, n_words, woff, lens, ids, out, out10, regs);
		chains_per_read = nullptr;
		rep_per_read = nullptr;
		if (n < 0) return n;
		chains.resize(n_q, 0u);
		rep.resize(n_q, 0u);
		std::vector<int32_t> thr(n_q, mid);
		uint64_t again = 0;
		for (uint32_t i = 0; i < n_q; ++i)
			if (!chains[i] && rep[i]) thr[i] = o.max_occ, ++again;   // map.c:553 / :678: no chain AND rep_len > 0
		if (!again) return n;
		st.rechained += again;
		read_mid = thr.data();
		return map_once(o, mid, n_q, words, n_words, woff, lens, ids, out, out10, regs);
	}
};

static OvlParams to_params(const ndgpu_ovl_opt &o)
{
	OvlParams P{};
	P.k = o.k, P.w = o.w, P.hpc = o.hpc, P.no_diag = o.no_diag, P.no_dual = o.no_dual, P.min_cnt = o.min_cnt, P.min_sc = o.min_chain_score;
	P.bw = o.bw, P.max_gap = o.max_gap, P.max_skip = o.max_chain_skip, P.max_iter = o.max_chain_iter, P.minlen = o.minlen, P.dvt = o.dvt;
	P.maxhan1 = o.maxhan1, P.maxhan2 = o.maxhan2;
	P.mode3 = o.mode == 3, P.ide_ml = 6000 /* mm_mapopt_t::ide_ml, options.c:60: no command-line switch */, P.d_factor = o.d_factor;
	P.step2 = o.step == 2, P.minmatch = o.minmatch, P.minide = o.minide;
	return P;
}

static const char *check_opt(const ndgpu_ovl_opt &o)
{
	if (o.k < 1 || o.k > 127 || (o.k > 28 && !(o.k & 31)))
		return "k must be in 1..127 and not 32, 64 or 96 (the reference's long k-mer mask is undefined there: sketch.c:286-287)";
	if (o.w < 1 || o.w > 64) return "w must be in 1..64";
	if (o.max_chain_iter < 1 || o.max_chain_iter >= 8192) return "max_chain_iter must be in 1..8191";
	if (o.max_gap < 0 || o.bw < 0) return "negative max_gap / bw";
	if (o.step == 2 && (o.mode < 0 || o.mode > 2)) return "--step 2 is built for --mode 0 (no re-alignment), 1 and 2 (the default)";
	return nullptr;
}

int64_t Engine::map_once(const ndgpu_ovl_opt &o, int32_t mid, uint32_t n_q, const uint32_t *words, uint64_t n_words, const uint64_t *woff,
                    const uint32_t *lens, const uint32_t *ids, std::vector<OvlRec> &out, std::vector<OvlRec10> *out10, const Regs *regs)
{
	OvlParams Pm = to_params(o);
	Pm.k = P.k, Pm.w = P.w, Pm.hpc = P.hpc; // the sketch parameters belong to the index
	if (regs) {
		Pm.provisional = 1, Pm.step2 = 0, Pm.mode3 = 0, Pm.dvt = 0, Pm.nameless = regs->nameless, Pm.thin = regs->thin, Pm.chains = regs->ca_x != nullptr;
		if (regs->ca_x) regs->ca_x->clear(), regs->ca_y->clear(), regs->ca_off->assign(1, 0);
		if (regs->nameless) Pm.no_diag = Pm.no_dual = 0; // skip_seed looks at names only when there is one (minimap2/map.c:129)
		regs->counts->clear();
	}
	const OvlParams Pi = P;
	P = Pm;
	struct Restore { Engine *e; OvlParams p; ~Restore() { e->P = p; } } restore{this, Pi};
	++st.map_calls;
	// The lookup table is built when it pays, by cost and not by the number of calls (one fused call may carry the query files of eight
	// jobs): config 2 measured 25 ps saved per query minimizer looked up (seed pass 13.5 -> 9.6 ms for 156 M) against 86 ps per distinct
	// key to build (6.7 ms for 78 M) -- so once the minimizers this index has been, and is about to be, asked for exceed ~3.4 x its keys.
	// NDGPU_OVL_HASH=1: with the first call; NDGPU_OVL_NO_HASH: never.
	{
		++maps_served;
		uint64_t bases = 0;
		for (uint32_t i = 0; i < n_q; ++i) bases += lens[i];
		query_minimizers_seen += 2.0 * (double)bases / (double)(P.w + 1);
		if (getenv("NDGPU_OVL_HASH") || query_minimizers_seen * 25.0 > (double)n_keys * 86.0) build_hash();
	}
	out.clear();
	if (out10) out10->clear();
	if (chains_per_read) chains_per_read->clear();
	if ((P.step2 != 0) != (out10 != nullptr)) return -1; // the two record types have an entry point each
	if (!n_q) return 0;

	ReadSetDev Q;
	Q.upload(n_q, words, n_words, woff, lens, ids, stream);
	std::vector<uint32_t> qh(n_q);
	for (uint32_t i = 0; i < n_q; ++i)
		qh[i] = regs && regs->nameless ? wang32(wang32(lens[i]) + wang32((uint32_t)o.seed)) : read_hash(ids[i], lens[i], o.seed);
	DevBuf<uint32_t> qhash(n_q);
	qhash.upload(qh.data(), n_q, stream);

	Sketch S;
	sketch(Q, 0, true, S);
	const uint64_t n_m = S.n;
	const IndexDev ix = index_dev();
	DevBuf<uint64_t> d_want_off;
	DevBuf<uint32_t> d_want;
	if (regs && regs->want_off) {
		const uint64_t nw = regs->want_off[n_q];
		d_want_off.alloc(n_q + 1), d_want.alloc(nw + 1);
		d_want_off.upload(regs->want_off, n_q + 1, stream);
		if (nw) d_want.upload(regs->want, nw, stream);
	}
	DevBuf<int32_t> d_read_mid;
	if (read_mid) {
		d_read_mid.alloc(n_q);
		d_read_mid.upload(read_mid, n_q, stream);
	}
	DevBuf<uint32_t> d_rep;
	if (rep_per_read) {
		d_rep.alloc(n_q);
		d_rep.zero(stream);
	}
	const QueryDev qd{Q.len.p, Q.id.p, qhash.p, Q.namekey.p, S.off.p, d_want_off.p, regs && regs->want_off ? d_want.p : nullptr, d_read_mid.p, d_rep.p};

	// K3a over every query minimizer
	EvTimer tm(stream);
	tm.start();
	DevBuf<uint32_t> m_start(n_m + 1), m_cnt(n_m + 1), m_surv(n_m + 2);
	m_surv.zero(stream);
	launch_seed_count(S.x.p, S.y.p, S.read.p, n_m, ix, qd, P, mid, m_start.p, m_cnt.p, m_surv.p, stream);
	DevBuf<uint64_t> a_off(n_m + 2);
	size_t tb = 0;
	exscan_u32_to_u64(nullptr, tb, m_surv.p, a_off.p, n_m + 1, stream);
	exscan_u32_to_u64(temp(tb), tb, m_surv.p, a_off.p, n_m + 1, stream);
	uint64_t total_a = 0;
	a_off.download(&total_a, 1, stream, n_m);
	HIP_OK(hipStreamSynchronize(stream));
	DevBuf<uint64_t> r_aoff_all(n_q + 1);
	launch_gather_read_off(a_off.p, S.off.p, n_q, n_m, total_a, r_aoff_all.p, stream);
	std::vector<uint64_t> h_raoff(n_q + 1), h_moff(n_q + 1);
	r_aoff_all.download(h_raoff.data(), n_q + 1, stream);
	S.off.download(h_moff.data(), n_q + 1, stream);
	HIP_OK(hipGetLastError());
	if (regs && regs->max_anchors) {
		HIP_OK(hipStreamSynchronize(stream));
		uint64_t m = 0;
		for (uint32_t i = 0; i < n_q; ++i) m = std::max<uint64_t>(m, h_raoff[i + 1] - h_raoff[i]);
		*regs->max_anchors = m;
	}
	st.seed_ms += tm.stop();
	st.anchors += total_a;

	// batches of query reads bounded by an anchor budget and by the sort-key width
	const unsigned pos_bits = bits_for(T.max_len ? T.max_len - 1 : 0);
	const unsigned rid_bits = bits_for(T.n ? T.n - 1 : 0);
	if (pos_bits + rid_bits + 1 >= 63) { fprintf(stderr, "[ndgpu_overlap] target set too large for the anchor sort key\n"); return -1; }
	const unsigned read_bits_max = 64 - (pos_bits + rid_bits + 1);
	const uint64_t max_batch_reads = read_bits_max >= 32 ? 0xffffffffULL : (1ULL << read_bits_max);
	uint64_t budget = 192ULL << 20; // anchors per batch (~100 B of HBM each)
	if (const char *e = getenv("NDGPU_OVL_BATCH_ANCHORS")) budget = std::max<uint64_t>(1024, strtoull(e, nullptr, 10));

	// The batches can run side by side (NDGPU_OVL_LANES of them at a time, a host thread and a stream each; the reads of a batch know
	// nothing of the other batches).  Measured in round 5 on the config-2 job (one batch cut into `lanes`): 1 lane 91.5 ms, 2 lanes
	// 97.5, 3 lanes 96.8, 4 lanes 100.4, 6 lanes 133.9 -- the kernels of a batch are not the idle chains they look like in a trace
	// (K4 and K5 fill the device while their heaviest read pair finishes), and every further stream costs what more device contexts
	// cost the consensus stage.  So one lane is the default; the lanes stay for sets whose batches are many and small.
	int lanes = 1;
	if (const char *e = getenv("NDGPU_OVL_LANES")) lanes = std::max(1, atoi(e));
	if (lanes > 1 && total_a >= (uint64_t)lanes * (2ULL << 20)) budget = std::min<uint64_t>(budget, total_a / (uint64_t)lanes + 1);
	std::vector<std::pair<uint32_t, uint32_t>> ranges;
	for (uint32_t r0 = 0; r0 < n_q;) {
		uint32_t r1 = r0;
		while (r1 < n_q && (r1 - r0) < max_batch_reads && (r1 == r0 || h_raoff[r1 + 1] - h_raoff[r0] <= budget)) ++r1;
		ranges.push_back({r0, r1});
		r0 = r1;
	}
	struct BatchOut {
		std::vector<OvlRec> recs;
		std::vector<OvlRec10> recs10;
		std::vector<uint32_t> counts, chains;
		std::vector<uint64_t> ca_x, ca_y, ca_off{0};   // (ca_off: relative to the batch)
		ndgpu_ovl_stats st{};
	};
	std::vector<BatchOut> outs(ranges.size());
	const Regs *const regs_all = regs;
	const bool want10 = out10 != nullptr;
	std::mutex dbg_mu;
	const size_t n_lanes = std::min<size_t>((size_t)lanes, ranges.size());
	auto run_batch = [&](size_t bi, hipStream_t stream, DevBuf<uint8_t> &lane_tmp) {
		// (everything the batch touches by these names is its own: its stream, its scratch, its counters, its output)
		BatchOut &BO = outs[bi];
		ndgpu_ovl_stats &st = BO.st;
		std::vector<OvlRec> &out = BO.recs;
		std::vector<OvlRec10> *const out10 = want10 ? &BO.recs10 : nullptr;
		Regs lregs{};
		if (regs_all) {
			lregs = *regs_all;
			lregs.counts = &BO.counts, lregs.max_anchors = nullptr;
			if (regs_all->ca_x) lregs.ca_x = &BO.ca_x, lregs.ca_y = &BO.ca_y, lregs.ca_off = &BO.ca_off;
		}
		const Regs *const regs = regs_all ? &lregs : nullptr;
		auto temp = [&](size_t bytes) -> void * { if (lane_tmp.n < bytes) lane_tmp.alloc(bytes + bytes / 4); return lane_tmp.p; };
		auto exscan = [&](const uint32_t *in, uint64_t *o, size_t n) {
			size_t tb2 = 0;
			exscan_u32_to_u64(nullptr, tb2, in, o, n, stream);
			exscan_u32_to_u64(temp(tb2), tb2, in, o, n, stream);
		};
		EvTimer tm(stream);
		size_t tb = 0;
		const uint32_t r0 = ranges[bi].first, r1 = ranges[bi].second;
		if (chains_per_read) BO.chains.assign(r1 - r0, 0u);
		const uint32_t nb = r1 - r0;
		const uint64_t a_base = h_raoff[r0], na = h_raoff[r1] - a_base;
		++st.batches;
		if (na == 0) {
			if (regs) regs->counts->insert(regs->counts->end(), nb, 0u);
			if (regs && regs->ca_off) regs->ca_off->insert(regs->ca_off->end(), nb, regs->ca_off->back());
			return;
		}
		KeyLayout L;
		L.pos_bits = pos_bits, L.rev_shift = pos_bits + rid_bits, L.read_shift = L.rev_shift + 1, L.read_base = r0;
		L.total_bits = L.read_shift + bits_for(nb - 1);

		DevBuf<uint64_t> r_aoff(nb + 1);
		launch_local_off(r_aoff_all.p, r0, nb, r_aoff.p, stream);
		DevBuf<uint64_t> ckey(na), uy(na), skey(na), ay(na);
		tm.start();
		launch_seed_fill(S.x.p, S.y.p, S.read.p, h_moff[r0], h_moff[r1], ix, qd, P, m_start.p, m_cnt.p, a_off.p, a_base, L, ckey.p, uy.p,
		                 stream);
		HIP_OK(hipGetLastError());
		st.seed_ms += tm.stop();

		tm.start();
		tb = 0;
		sort_pairs_u64(nullptr, tb, ckey.p, skey.p, uy.p, ay.p, na, 0, L.total_bits, stream);
		sort_pairs_u64(temp(tb), tb, ckey.p, skey.p, uy.p, ay.p, na, 0, L.total_bits, stream);
		DevBuf<uint64_t> ax(na);
		DevBuf<uint32_t> tie(nb);
		tie.zero(stream);
		DevBuf<uint64_t> segval(na), segstart1(na);
		launch_anchor_decode(skey.p, na, L, ax.p, tie.p, segval.p, stream);
		// K4 work units: runs of whole (strand, target) segments of a read, ~1024 anchors each
		tb = 0;
		incl_max_scan_u64(nullptr, tb, segval.p, segstart1.p, na, stream);
		incl_max_scan_u64(temp(tb), tb, segval.p, segstart1.p, na, stream);
		DevBuf<uint32_t> slab_flag(na + 1);
		slab_flag.zero(stream);
		launch_slab_flag(skey.p, segstart1.p, na, L, r_aoff.p, slab_flag.p, stream);
		DevBuf<uint64_t> slab_rank(na + 1);
		exscan(slab_flag.p, slab_rank.p, na + 1);
		uint64_t n_slabs = 0;
		slab_rank.download(&n_slabs, 1, stream, na);
		std::vector<uint32_t> h_tie(nb);
		tie.download(h_tie.data(), nb, stream);
		HIP_OK(hipGetLastError());
		st.sort_ms += tm.stop();
		std::vector<uint32_t> tie_reads;
		for (uint32_t i = 0; i < nb; ++i) if (h_tie[i]) tie_reads.push_back(i);
		st.tie_reads += tie_reads.size();
		DevBuf<uint8_t> stacks((na / 64 + 2 * (size_t)nb + 4) * sort_job_bytes());
		DevBuf<uint64_t> bx(na), by(na);   // K5 chain buffers; scratch of the replay passes before that
		DevBuf<int32_t> t(na);             // K4/K5 marks; scratch of the replay passes before that
		if (!tie_reads.empty()) {
			tm.start();
			DevBuf<uint32_t> d_tie(tie_reads.size());
			d_tie.upload(tie_reads.data(), tie_reads.size(), stream);
			const size_t job_cap = na / 64 + tie_reads.size() + 4;
			DevBuf<uint8_t> jobs_a(job_cap * sort_job_bytes()), jobs_b(job_cap * sort_job_bytes());
			DevBuf<uint32_t> n_jobs(2);
			n_jobs.zero(stream);
			launch_sort_init(d_tie.p, (uint32_t)tie_reads.size(), r_aoff.p, ckey.p, uy.p, L, ax.p, ay.p, jobs_a.p, n_jobs.p, stream);
			uint32_t cur = 0;
			n_jobs.download(&cur, 1, stream);
			HIP_OK(hipStreamSynchronize(stream));
			DevBuf<uint8_t> *ja = &jobs_a, *jb = &jobs_b;
			int which = 0;
			while (cur) { // at most 8 rounds (digit positions 56, 48, ..., 0)
				HIP_OK(hipMemsetAsync(n_jobs.p + (1 - which), 0, 4, stream));
				launch_sort_pass(ja->p, cur, ax.p, ay.p, bx.p, by.p, (uint32_t*)t.p, jb->p, n_jobs.p + (1 - which), stream);
				n_jobs.download(&cur, 1, stream, 1 - which);
				HIP_OK(hipStreamSynchronize(stream));
				std::swap(ja, jb);
				which = 1 - which;
			}
			HIP_OK(hipGetLastError());
			st.exact_sort_ms += tm.stop();
		}
		DevBuf<uint64_t> slab_i0(n_slabs + 1);
		DevBuf<uint32_t> slab_read(n_slabs + 1);
		launch_slab_write(skey.p, slab_flag.p, slab_rank.p, na, L, slab_i0.p, slab_read.p, stream);
		DevBuf<float> avg_span(nb + 1);
		launch_read_span(r_aoff.p, nb, ay.p, avg_span.p, stream);
		HIP_OK(hipStreamSynchronize(stream));
		ckey.release(); uy.release(); skey.release(); segval.release(); segstart1.release(); slab_flag.release(); slab_rank.release();

		// K4
		DevBuf<int32_t> f(na), p(na), v(na);
		DevBuf<uint64_t> u(na);
		DevBuf<uint32_t> n_end(nb + 1);
		DevBuf<unsigned long long> cells(1);
		cells.zero(stream);
		t.zero(stream);
		tm.start();
		if (P.thin) launch_thin_anchors(r_aoff.p, nb, ax.p, t.p, v.p, qd.read_mid, r0, mid, stream);
		launch_chain(slab_i0.p, slab_read.p, (uint32_t)n_slabs, na, r_aoff.p, avg_span.p, ax.p, ay.p, P, f.p, p.p, v.p, cells.p, stream);
		launch_chain_ends(r_aoff.p, nb, P, f.p, p.p, v.p, t.p, u.p, n_end.p, stream);
		HIP_OK(hipGetLastError());
		st.chain_ms += tm.stop();
		unsigned long long h_cells = 0;
		cells.download(&h_cells, 1, stream);

		// K5 (keeps copies of f/p for the debug view first: K5 reuses v and t only)
		const uint64_t rec_cap = na / (uint64_t)std::max(1, P.min_cnt) + nb + 1;
		const uint64_t n_w = P.min_cnt < 2 ? 2 * na : na;   // (K5's chain tables: see hits_kernel)
		DevBuf<uint64_t> wx(n_w), wy(n_w);
		DevBuf<uint32_t> tables((size_t)nb * 512), n_rec(nb + 1), n_chain(nb);
		DevBuf<OvlRec> recs(rec_cap);
		DevBuf<OvlRec10> recs10(P.step2 ? rec_cap : 0);
		DevBuf<uint64_t> cx(P.chains ? na : 0), cy(P.chains ? na : 0);
		DevBuf<uint32_t> n_ca(P.chains ? nb + 1 : 0);
		n_rec.zero(stream);
		if (P.chains) n_ca.zero(stream);
		tm.start();
		launch_hits(r_aoff.p, nb, r0, ax.p, ay.p, ix, qd, P, f.p, p.p, v.p, t.p, u.p, bx.p, by.p, wx.p, wy.p, tables.p, stacks.p, n_end.p,
		            recs.p, n_rec.p, n_chain.p, recs10.p, cx.p, cy.p, n_ca.p, stream);
		DevBuf<uint64_t> rec_off(nb + 1);
		tb = 0;
		exscan_u32_to_u64(nullptr, tb, n_rec.p, rec_off.p, nb + 1, stream);
		exscan_u32_to_u64(temp(tb), tb, n_rec.p, rec_off.p, nb + 1, stream);
		uint64_t n_out = 0;
		rec_off.download(&n_out, 1, stream, nb);
		std::vector<uint32_t> h_chain(nb);
		n_chain.download(h_chain.data(), nb, stream);
		if (regs) {
			const size_t at = regs->counts->size();
			regs->counts->resize(at + nb);
			n_rec.download(regs->counts->data() + at, nb, stream);
		}
		HIP_OK(hipStreamSynchronize(stream));
		if (chains_per_read) BO.chains = h_chain;
		if (P.chains) {
			DevBuf<uint64_t> ca_off(nb + 1);
			tb = 0;
			exscan_u32_to_u64(nullptr, tb, n_ca.p, ca_off.p, nb + 1, stream);
			exscan_u32_to_u64(temp(tb), tb, n_ca.p, ca_off.p, nb + 1, stream);
			std::vector<uint64_t> h_off(nb + 1);
			ca_off.download(h_off.data(), nb + 1, stream);
			HIP_OK(hipStreamSynchronize(stream));
			const uint64_t n_c = h_off[nb], base = regs->ca_off->back();
			DevBuf<uint64_t> dx(n_c + 1), dy(n_c + 1);
			launch_compact_anchors(r_aoff.p, nb, cx.p, cy.p, n_ca.p, ca_off.p, dx.p, dy.p, stream);
			HIP_OK(hipGetLastError());
			const size_t at = regs->ca_x->size();
			regs->ca_x->resize(at + n_c), regs->ca_y->resize(at + n_c);
			if (n_c) {
				HIP_OK(hipMemcpyAsync(regs->ca_x->data() + at, dx.p, n_c * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
				HIP_OK(hipMemcpyAsync(regs->ca_y->data() + at, dy.p, n_c * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
			}
			HIP_OK(hipStreamSynchronize(stream));
			for (uint32_t i = 1; i <= nb; ++i) regs->ca_off->push_back(base + h_off[i]);
		}
		if (P.step2) {
			DevBuf<OvlRec10> dense10(n_out + 1);
			launch_compact_recs10(r_aoff.p, nb, P.min_cnt, recs10.p, n_rec.p, rec_off.p, dense10.p, stream);
			HIP_OK(hipGetLastError());
			st.hits_ms += tm.stop();
			const size_t old10 = out10->size();
			out10->resize(old10 + n_out);
			if (n_out) HIP_OK(hipMemcpyAsync(out10->data() + old10, dense10.p, n_out * sizeof(OvlRec10), hipMemcpyDeviceToHost, stream));
			HIP_OK(hipStreamSynchronize(stream));
			st.chain_cells += h_cells;
			for (uint32_t c : h_chain) st.chains += c;
			st.overlaps += n_out;
			return;
		}
		DevBuf<OvlRec> dense(n_out + 1);
		launch_compact_recs(r_aoff.p, nb, P.min_cnt, recs.p, n_rec.p, rec_off.p, dense.p, stream);
		HIP_OK(hipGetLastError());
		st.hits_ms += tm.stop();
		if (P.mode3 && n_out) {
			// --mode 3: the records are provisional (see K5); extend both ends of every hit, then filter and name them
			tm.start();
			const uint64_t n_task = 2 * n_out;
			DevBuf<uint32_t> need(n_task + 1);
			need.zero(stream);
			launch_ext_size(dense.p, n_out, Q.len.p, T.len.p, P, need.p, stream);
			DevBuf<uint64_t> fr_off(n_task + 1);
			exscan(need.p, fr_off.p, n_task + 1);
			std::vector<uint64_t> h_off(n_task + 1);
			fr_off.download(h_off.data(), n_task + 1, stream);
			HIP_OK(hipStreamSynchronize(stream));
			DevBuf<int32_t> ext_x(n_task), ext_y(n_task), fr;
			uint64_t scratch = 256ULL << 20; // ints of furthest-reaching scratch per launch (1 GB)
			if (const char *e = getenv("NDGPU_OVL_EXT_SCRATCH")) scratch = std::max<uint64_t>(1024, strtoull(e, nullptr, 10));
			for (uint64_t t0 = 0; t0 < n_task;) {
				uint64_t t1 = t0;
				while (t1 < n_task && (t1 == t0 || h_off[t1 + 1] - h_off[t0] <= scratch)) ++t1;
				const uint64_t ints = h_off[t1] - h_off[t0];
				if (fr.n < ints + 1) fr.alloc(ints + 1);
				if (ints) HIP_OK(hipMemsetAsync(fr.p, 0, ints * sizeof(int32_t), stream));
				launch_ext_ends(dense.p, t0, t1, Q.wp, Q.woff.p, Q.len.p, T.wp, T.woff.p, T.len.p, P, fr_off.p, h_off[t0], fr.p,
				                ext_x.p, ext_y.p, stream);
				++st.ext_launches;
				t0 = t1;
			}
			for (uint64_t t = 0; t < n_task; ++t) st.ext_problems += h_off[t + 1] != h_off[t];
			DevBuf<uint32_t> keep(n_out + 1);
			keep.zero(stream);
			launch_ext_apply(dense.p, n_out, ext_x.p, ext_y.p, Q.id.p, Q.len.p, T.id.p, T.len.p, P, keep.p, stream);
			DevBuf<uint64_t> pos(n_out + 1);
			exscan(keep.p, pos.p, n_out + 1);
			uint64_t n_keep = 0;
			pos.download(&n_keep, 1, stream, n_out);
			HIP_OK(hipStreamSynchronize(stream));
			DevBuf<OvlRec> fin(n_keep + 1);
			launch_scatter_recs(dense.p, n_out, keep.p, pos.p, fin.p, stream);
			HIP_OK(hipGetLastError());
			dense = std::move(fin);
			n_out = n_keep;
			st.ext_ms += tm.stop();
		}
		const size_t old = out.size();
		out.resize(old + n_out);
		if (n_out) HIP_OK(hipMemcpyAsync(out.data() + old, dense.p, n_out * sizeof(OvlRec), hipMemcpyDeviceToHost, stream));
		HIP_OK(hipStreamSynchronize(stream));
		st.chain_cells += h_cells;
		for (uint32_t c : h_chain) st.chains += c;
		st.overlaps += n_out;

		// debug view (last batch)
		if (bi + 1 == ranges.size()) {
			std::lock_guard<std::mutex> g(dbg_mu);
			dbg_r0 = r0, dbg_n = nb;
			dbg_aoff.assign(h_raoff.begin() + r0, h_raoff.begin() + r1 + 1);
			for (auto &x : dbg_aoff) x -= a_base;
			dbg_ax = std::move(ax); dbg_ay = std::move(ay); dbg_f = std::move(f); dbg_p = std::move(p);
		}
	};
	if (n_lanes <= 1) {
		for (size_t bi = 0; bi < ranges.size(); ++bi) run_batch(bi, stream, tmp);
	} else {
		HIP_OK(hipStreamSynchronize(stream));   // (the minimizers, their seed counts and offsets are final)
		while (lane_streams.size() < n_lanes) {
			hipStream_t ls = nullptr;
			HIP_OK(hipStreamCreate(&ls));
			lane_streams.push_back(ls);
		}
		std::atomic<size_t> next{0};
		std::vector<std::exception_ptr> errs(n_lanes);
		std::vector<std::thread> th;
		for (size_t w = 0; w < n_lanes; ++w)
			th.emplace_back([&, w] {
				try {
					HIP_OK(hipSetDevice(device));
					DevBuf<uint8_t> lane_tmp;
					for (;;) {
						const size_t bi = next.fetch_add(1);
						if (bi >= ranges.size()) break;
						run_batch(bi, lane_streams[w], lane_tmp);
					}
					HIP_OK(hipStreamSynchronize(lane_streams[w]));
				} catch (...) {
					errs[w] = std::current_exception();
					next = ranges.size();   // (the other lanes stop at their next batch)
					(void)hipStreamSynchronize(lane_streams[w]);   // nothing of this lane is in flight when its blocks go back to the pool
				}
			});
		for (auto &t : th) t.join();
		for (auto &e : errs) if (e) std::rethrow_exception(e);
	}
	// the batches' outputs, in read order
	for (BatchOut &BO : outs) {
		if (out.empty()) out = std::move(BO.recs);   // (the usual call is one batch: its vector is the result)
		else out.insert(out.end(), BO.recs.begin(), BO.recs.end());
		if (out10) {
			if (out10->empty()) *out10 = std::move(BO.recs10);
			else out10->insert(out10->end(), BO.recs10.begin(), BO.recs10.end());
		}
		if (chains_per_read) chains_per_read->insert(chains_per_read->end(), BO.chains.begin(), BO.chains.end());
		if (regs) {
			regs->counts->insert(regs->counts->end(), BO.counts.begin(), BO.counts.end());
			if (regs->ca_x) {
				const uint64_t base = regs->ca_off->back();
				regs->ca_x->insert(regs->ca_x->end(), BO.ca_x.begin(), BO.ca_x.end());
				regs->ca_y->insert(regs->ca_y->end(), BO.ca_y.begin(), BO.ca_y.end());
				for (size_t i = 1; i < BO.ca_off.size(); ++i) regs->ca_off->push_back(base + BO.ca_off[i]);
			}
		}
		const ndgpu_ovl_stats &b = BO.st;
		st.seed_ms += b.seed_ms, st.sort_ms += b.sort_ms, st.exact_sort_ms += b.exact_sort_ms, st.chain_ms += b.chain_ms, st.hits_ms += b.hits_ms;
		st.ext_ms += b.ext_ms, st.tie_reads += b.tie_reads, st.chain_cells += b.chain_cells, st.chains += b.chains, st.overlaps += b.overlaps;
		st.batches += b.batches, st.ext_problems += b.ext_problems, st.ext_launches += b.ext_launches;
	}
	if (rep_per_read) {   // (K3a ran over every query minimizer before the first batch: the flags are whole)
		rep_per_read->assign(n_q, 0u);
		d_rep.download(rep_per_read->data(), n_q, stream);
		HIP_OK(hipStreamSynchronize(stream));
	}
	return out10 ? (int64_t)out10->size() : (int64_t)out.size();
}

} // namespace ndovl

using namespace ndovl;

struct ndgpu_ovl_index { Engine e; };

extern "C" {

int ndgpu_ovl_opt_preset(const char *preset, ndgpu_ovl_opt *o)
{
	memset(o, 0, sizeof(*o));
	// mm_idxopt_init / mm_mapopt_init (minimap2/options.c:4-62) + --step 1 (main.c:192-193)
	o->k = 15, o->w = 10, o->hpc = 0;
	o->seed = 11, o->mid_occ_frac = 2e-4f, o->min_cnt = 3, o->min_chain_score = 40, o->bw = 500, o->max_gap = 5000;
	o->max_chain_skip = 25, o->max_chain_iter = 5000, o->minlen = 500, o->maxhan1 = 5000, o->maxhan2 = 500, o->dvt = 0;
	o->mode = 2, o->d_factor = 0.1f; // options.c:56,62 (--step 1 looks at the mode only to see whether it is 3)
	o->step = 1, o->minide = 0.05f, o->minmatch = 100; // --step 2 (main.c:194-197) also sets minlen = 2000: the caller's to do
	if (!preset) return 0;
	if (strcmp(preset, "ava-ont") == 0) { // options.c:84-88
		o->k = 15, o->w = 5, o->hpc = 0, o->no_diag = 1, o->no_dual = 1;
		o->min_chain_score = 100, o->max_gap = 10000, o->max_chain_skip = 25, o->bw = 2000;
	} else if (strcmp(preset, "ava-pb") == 0) { // options.c:89-92
		o->k = 19, o->w = 5, o->hpc = 1, o->no_diag = 1, o->no_dual = 1;
		o->min_chain_score = 100, o->max_gap = 10000, o->max_chain_skip = 25;
	} else if (strcmp(preset, "ava-hifi") == 0) { // options.c:99-111 (the alignment scores it also sets are not used by --step 1)
		o->k = 51, o->w = 51, o->hpc = 1, o->no_diag = 1, o->no_dual = 1, o->mid_occ_frac = 1e-4f;
		o->min_chain_score = 100, o->max_gap = 10000, o->max_chain_skip = 25;
	} else {
		fprintf(stderr, "[ndgpu_overlap] preset '%s' is not supported (ava-ont, ava-pb, ava-hifi)\n", preset);
		return -1;
	}
	return 0;
}

ndgpu_ovl_index *ndgpu_ovl_index_create(const ndgpu_ovl_opt *opt, uint32_t n_reads, const uint32_t *words, uint64_t n_words,
                                        const uint64_t *word_off, const uint32_t *lens, const uint32_t *ids)
{
	if (const char *msg = check_opt(*opt)) { fprintf(stderr, "[ndgpu_overlap] %s\n", msg); return nullptr; }
	ndgpu_ovl_index *h = nullptr;
	try {
		setenv("GPU_MAX_HW_QUEUES", "16", 0); // see csrc/device_runtime.hip: the process may go on to drive 8 consensus streams
		int n_dev = 0;
		if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
			fprintf(stderr, "[ndgpu_overlap] no HIP device: the overlap engine has no CPU path\n");
			return nullptr;
		}
		int dev = 0;
		if (const char *e = getenv("NDGPU_DEVICE")) dev = atoi(e);
		HIP_OK(hipSetDevice(dev));
		if (!getenv("NDGPU_SPIN_SYNC")) (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);  // (a waiting host thread sleeps: see device_runtime.hip)
		h = new ndgpu_ovl_index();
		h->e.device = dev;
		HIP_OK(ndovl::create_stage_stream(&h->e.stream));
		h->e.P = to_params(*opt);
		h->e.T.upload(n_reads, words, n_words, word_off, lens, ids, h->e.stream);
		h->e.build_index();
		return h;
	} catch (...) {
		delete h;
		return nullptr;
	}
}

void ndgpu_ovl_index_destroy(ndgpu_ovl_index *h)
{
	if (!h) return;
	hipStream_t s = h->e.stream;
	if (s) (void)hipStreamSynchronize(s);
	const std::vector<hipStream_t> lanes = h->e.lane_streams;
	for (hipStream_t ls : lanes) (void)hipStreamSynchronize(ls);
	delete h;
	if (s) (void)hipStreamDestroy(s);
	for (hipStream_t ls : lanes) (void)hipStreamDestroy(ls);
}

int32_t ndgpu_ovl_index_mid_occ(ndgpu_ovl_index *h, float frac)
{
	try { return h->e.mid_occ(frac); } catch (...) { return -1; }
}

void ndgpu_ovl_index_stat(const ndgpu_ovl_index *h, uint64_t n[3]) { n[0] = h->e.n_min, n[1] = h->e.n_keys, n[2] = h->e.T.n; }

int64_t ndgpu_ovl_map2(ndgpu_ovl_index *h, const ndgpu_ovl_opt *opt, int32_t mid_occ, uint32_t n_reads, const uint32_t *words,
                       uint64_t n_words, const uint64_t *word_off, const uint32_t *lens, const uint32_t *ids, ndgpu_ovl_rec10 **recs)
{
	*recs = nullptr;
	if (const char *msg = check_opt(*opt)) { fprintf(stderr, "[ndgpu_overlap] %s\n", msg); return -1; }
	if (opt->step != 2) { fprintf(stderr, "[ndgpu_overlap] ndgpu_ovl_map2 is the --step 2 entry (opt->step = 2)\n"); return -1; }
	try {
		HIP_OK(hipSetDevice(h->e.device));
		std::vector<OvlRec> none;
		std::vector<OvlRec10> out;
		int64_t n = h->e.map(*opt, mid_occ, n_reads, words, n_words, word_off, lens, ids, none, &out);
		if (n < 0) return n;
		static_assert(sizeof(OvlRec10) == sizeof(ndgpu_ovl_rec10), "record layout");
		*recs = (ndgpu_ovl_rec10*)malloc(sizeof(ndgpu_ovl_rec10) * (size_t)(n ? n : 1));
		if (n) memcpy(*recs, out.data(), sizeof(ndgpu_ovl_rec10) * (size_t)n);
		return n;
	} catch (...) {
		return -2;
	}
}

int64_t ndgpu_ovl_map(ndgpu_ovl_index *h, const ndgpu_ovl_opt *opt, int32_t mid_occ, uint32_t n_reads, const uint32_t *words,
                      uint64_t n_words, const uint64_t *word_off, const uint32_t *lens, const uint32_t *ids, ndgpu_ovl_rec **recs)
{
	*recs = nullptr;
	if (const char *msg = check_opt(*opt)) { fprintf(stderr, "[ndgpu_overlap] %s\n", msg); return -1; }
	if (opt->step == 2) { fprintf(stderr, "[ndgpu_overlap] --step 2 records come from ndgpu_ovl_map2\n"); return -1; }
	try {
		HIP_OK(hipSetDevice(h->e.device));
		std::vector<OvlRec> out;
		int64_t n = h->e.map(*opt, mid_occ, n_reads, words, n_words, word_off, lens, ids, out);
		if (n < 0) return n;
		static_assert(sizeof(OvlRec) == sizeof(ndgpu_ovl_rec), "record layout");
		*recs = (ndgpu_ovl_rec*)malloc(sizeof(ndgpu_ovl_rec) * (size_t)(n ? n : 1));
		if (n) memcpy(*recs, out.data(), sizeof(ndgpu_ovl_rec) * (size_t)n);
		return n;
	} catch (...) {
		return -2;
	}
}

// The hits of every query read, nothing judged or filtered: hit order, target = position in the read's wanted list (or the
// index-local read number when want_off is NULL), block length in `tname`, match count in `match` -- the raw material of
// --step 2's marking and re-alignment (minimap2/map.c:997-1126), which run on the host (csrc/ovl_step2.cpp).
int64_t ndgpu_ovl_map_regs(ndgpu_ovl_index *h, const ndgpu_ovl_opt *opt, int32_t mid_occ, uint32_t n_reads, const uint32_t *words,
                           uint64_t n_words, const uint64_t *word_off, const uint32_t *lens, const uint32_t *ids, const uint64_t *want_off,
                           const uint32_t *want, int nameless, ndgpu_ovl_rec **recs, uint32_t **counts, uint64_t *max_anchors)
{
	*recs = nullptr, *counts = nullptr;
	if (max_anchors) *max_anchors = 0;
	if (const char *msg = check_opt(*opt)) { fprintf(stderr, "[ndgpu_overlap] %s\n", msg); return -1; }
	try {
		HIP_OK(hipSetDevice(h->e.device));
		std::vector<OvlRec> out;
		std::vector<uint32_t> cnt;
		Regs rg{want_off, want, (nameless & 1) != 0, &cnt, max_anchors};
		rg.thin = (nameless & 2) != 0;
		int64_t n = h->e.map(*opt, mid_occ, n_reads, words, n_words, word_off, lens, ids, out, nullptr, &rg);
		if (n < 0) return n;
		if (cnt.size() != n_reads) cnt.resize(n_reads, 0u);
		*recs = (ndgpu_ovl_rec*)malloc(sizeof(ndgpu_ovl_rec) * (size_t)(n ? n : 1));
		*counts = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n_reads ? n_reads : 1));
		if (n) memcpy(*recs, out.data(), sizeof(ndgpu_ovl_rec) * (size_t)n);
		if (n_reads) memcpy(*counts, cnt.data(), sizeof(uint32_t) * n_reads);
		return n;
	} catch (...) {
		return -2;
	}
}

// The chains of every query read as the base-level alignment of -c takes them (minimap2/align.c:857-905 after mm_gen_regs,
// minimap2/hit.c:52-85): hits in hit order, self hits included, chains[i] = (strand, index-local target, first anchor's offset in
// the read's slice of ax / ay, anchor count, chain score, hash, 0, 0); the chained anchors of read i = ax / ay[a_off[i] .. a_off[i + 1])
// in the order of the reference's a[] (chains by the x of their first anchor).
int64_t ndgpu_ovl_map_chains(ndgpu_ovl_index *h, const ndgpu_ovl_opt *opt, int32_t mid_occ, uint32_t n_reads, const uint32_t *words,
                             uint64_t n_words, const uint64_t *word_off, const uint32_t *lens, const uint32_t *ids, ndgpu_ovl_rec **chains,
                             uint32_t **counts, uint64_t **ax, uint64_t **ay, uint64_t **a_off)
{
	*chains = nullptr, *counts = nullptr, *ax = *ay = *a_off = nullptr;
	if (const char *msg = check_opt(*opt)) { fprintf(stderr, "[ndgpu_overlap] %s\n", msg); return -1; }
	if (opt->step != 1 || opt->mode == 3) { fprintf(stderr, "[ndgpu_overlap] chains are handed out for --step 1 without --mode 3 only\n"); return -1; }
	try {
		HIP_OK(hipSetDevice(h->e.device));
		std::vector<OvlRec> out;
		std::vector<uint32_t> cnt;
		std::vector<uint64_t> x, y, off;
		Regs rg{nullptr, nullptr, false, &cnt, nullptr};
		rg.ca_x = &x, rg.ca_y = &y, rg.ca_off = &off;
		int64_t n = h->e.map(*opt, mid_occ, n_reads, words, n_words, word_off, lens, ids, out, nullptr, &rg);
		if (n < 0) return n;
		if (cnt.size() != n_reads) cnt.resize(n_reads, 0u);
		if (off.size() != (size_t)n_reads + 1) off.resize((size_t)n_reads + 1, off.empty() ? 0 : off.back());
		*chains = (ndgpu_ovl_rec*)malloc(sizeof(ndgpu_ovl_rec) * (size_t)(n ? n : 1));
		*counts = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n_reads ? n_reads : 1));
		*ax = (uint64_t*)malloc(sizeof(uint64_t) * (x.size() ? x.size() : 1));
		*ay = (uint64_t*)malloc(sizeof(uint64_t) * (y.size() ? y.size() : 1));
		*a_off = (uint64_t*)malloc(sizeof(uint64_t) * ((size_t)n_reads + 1));
		if (!*chains || !*counts || !*ax || !*ay || !*a_off) return -2;
		if (n) memcpy(*chains, out.data(), sizeof(ndgpu_ovl_rec) * (size_t)n);
		if (n_reads) memcpy(*counts, cnt.data(), sizeof(uint32_t) * n_reads);
		if (!x.empty()) memcpy(*ax, x.data(), sizeof(uint64_t) * x.size()), memcpy(*ay, y.data(), sizeof(uint64_t) * y.size());
		memcpy(*a_off, off.data(), sizeof(uint64_t) * ((size_t)n_reads + 1));
		return n;
	} catch (...) {
		return -2;
	}
}

static int put_varint(uint8_t *out, uint32_t v)
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

int64_t ndgpu_ovl_encode(const ndgpu_ovl_rec *recs, int64_t n, uint32_t prev[2], uint8_t *out)
{
	int64_t nb = 0;
	for (int64_t i = 0; i < n; ++i) {
		const ndgpu_ovl_rec &r = recs[i];
		uint32_t fld[8], flags = r.rev;
		const uint32_t qspan = r.qe - r.qs, tspan = r.te - r.ts;
		fld[3] = qspan;
		if (r.qname >= prev[0]) fld[0] = r.qname - prev[0]; else flags |= 2, fld[0] = prev[0] - r.qname;
		prev[0] = r.qname;
		if (r.tname >= prev[1]) fld[4] = r.tname - prev[1]; else flags |= 4, fld[4] = prev[1] - r.tname;
		prev[1] = r.tname;
		if (qspan >= tspan) fld[6] = qspan - tspan; else flags |= 8, fld[6] = tspan - qspan;
		fld[1] = flags & 0xff, fld[2] = r.qs, fld[5] = r.ts, fld[7] = r.match;
		for (int k = 0; k < 8; ++k) nb += put_varint(out + nb, fld[k]);
	}
	return nb;
}

// decode_ovl() over a whole buffer (lib/ovl.c:152-203): `n_bytes` of 8-varint records -> out[8 * k] in decode_ovl's field order
// (qname, rev, qs, qe, tname, ts, te, match), prev[2] = the running (qname, tname) state.  A trailing partial record is left
// alone (*consumed tells where it starts).  Returns the number of records (at most cap).
int64_t ndgpu_ovl_decode(const uint8_t *buf, uint64_t n_bytes, uint32_t prev[2], uint32_t *out, int64_t cap, uint64_t *consumed)
{
	int64_t n = 0;
	uint64_t at = 0, rec_start = 0;
	while (n < cap) {
		uint32_t f[8];
		int k = 0;
		rec_start = at;
		for (; k < 8; ++k) {
			uint32_t v = 0;
			bool done = false;
			while (at < n_bytes) {
				const uint8_t c = buf[at++];
				v = (v << 7) | (c & 127u);
				if (c < 128) { done = true; break; }
			}
			if (!done) break;
			f[k] = v;
		}
		if (k < 8) { at = rec_start; break; }
		const uint32_t fl = f[1];
		uint32_t *o = out + 8 * n;
		prev[0] = (fl & 2) ? prev[0] - f[0] : prev[0] + f[0];
		prev[1] = (fl & 4) ? prev[1] - f[4] : prev[1] + f[4];
		o[0] = prev[0], o[1] = fl & 1, o[2] = f[2], o[3] = f[2] + f[3], o[4] = prev[1], o[5] = f[5];
		o[6] = (fl & 8) ? f[5] + f[3] + f[6] : f[5] + f[3] - f[6];
		o[7] = f[7];
		++n;
	}
	if (consumed) *consumed = at;
	return n;
}

// kbit_read()'s walk over a `.2bit` payload (lib/bseq.c:257-299): per read u32 id, u32 len, ceil(len / 16) words.
// Returns the number of reads (call with cap = 0 to count); word_off[i] = index of read i's first sequence word.
int64_t ndgpu_2bit_index(const uint32_t *w, uint64_t n_words, uint32_t *ids, uint32_t *lens, uint64_t *word_off, int64_t cap)
{
	int64_t n = 0;
	uint64_t p = 0;
	while (p + 2 <= n_words) {
		const uint32_t ln = w[p + 1];
		if (p + 2 + (((uint64_t)ln + 15) >> 4) > n_words) return -1;  // the record's words run past the buffer: truncated / corrupt
		if (n < cap) ids[n] = w[p], lens[n] = ln, word_off[n] = p + 2;
		++n;
		p += 2 + (((uint64_t)ln + 15) >> 4);
	}
	return n;
}

void ndgpu_ovl_free(void *p) { free(p); }

// 1 if an entry point has failed for lack of device memory since the last call of this function (the caller may free
// memory and try again), 2 for another allocation error, 0 otherwise; reading it clears it
int ndgpu_ovl_last_error(void) { return ndovl::last_error_take(); }

// device bytes of the library's block pool: in use now, cached for reuse, the most that ever were in use at once
void ndgpu_ovl_pool_bytes(uint64_t out[3]) { ndovl::pool_bytes(out); }
void ndgpu_ovl_pool_calls(uint64_t out[2], int reset) { ndovl::pool_calls(out, reset); }

// release the device blocks the library keeps cached between calls (returns the bytes released)
int ndgpu_ovl_words_resident(const uint32_t *words, uint64_t n_words)
{
	if (!words || !n_words) return -1;
	try {
		int device = 0;
		if (const char *d = getenv("NDGPU_DEVICE")) device = atoi(d);
		HIP_OK(hipSetDevice(device));
		{
			std::lock_guard<std::mutex> g(ndovl::g_res_mu);
			for (const ndovl::ResidentWords &r : ndovl::g_resident)
				if (r.host == words && r.n_words == n_words) return 0;
		}
		uint32_t *dev = (uint32_t*)ndovl::pool_alloc((n_words + 8) * sizeof(uint32_t));
		HIP_OK(hipMemcpy(dev, words, n_words * sizeof(uint32_t), hipMemcpyHostToDevice));
		HIP_OK(hipMemset(dev + n_words, 0, 8 * sizeof(uint32_t)));
		std::lock_guard<std::mutex> g(ndovl::g_res_mu);
		ndovl::g_resident.push_back(ndovl::ResidentWords{words, n_words, dev});
		return 0;
	} catch (...) {
		return -1;
	}
}

void ndgpu_ovl_words_release(const uint32_t *words)
{
	std::lock_guard<std::mutex> g(ndovl::g_res_mu);
	for (size_t i = 0; i < ndovl::g_resident.size(); ++i)
		if (ndovl::g_resident[i].host == words) {
			(void)hipDeviceSynchronize();
			ndovl::pool_free(ndovl::g_resident[i].dev);
			ndovl::g_resident.erase(ndovl::g_resident.begin() + (long)i);
			return;
		}
}

uint64_t ndgpu_ovl_trim(void)
{
	return (uint64_t)ndovl::pool_trim();   // (what went back to the driver: idle slabs, not everything that was cached)
}

int64_t ndgpu_pack_2bit(uint32_t n_reads, const uint8_t *ascii, uint64_t n_bytes, const uint64_t *ascii_off, const uint32_t *lens,
                        const uint64_t *word_off, uint32_t *words)
{
	if (!n_reads) return 0;
	try {
		int n_dev = 0;
		if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { fprintf(stderr, "[ndgpu_overlap] no HIP device\n"); return -1; }
		int device = 0;
		if (const char *d = getenv("NDGPU_DEVICE")) device = atoi(d);
		HIP_OK(hipSetDevice(device));
		hipStream_t st;
		HIP_OK(hipStreamCreate(&st));
		const uint64_t n_words = word_off[n_reads - 1] + ((uint64_t)lens[n_reads - 1] + 15) / 16;
		{
			DevBuf<uint8_t> d_a(n_bytes + 16);
			DevBuf<uint64_t> d_ao(n_reads), d_wo(n_reads);
			DevBuf<uint32_t> d_len(n_reads), d_w(n_words + 1);
			d_a.upload(ascii, n_bytes, st); d_ao.upload(ascii_off, n_reads, st); d_wo.upload(word_off, n_reads, st); d_len.upload(lens, n_reads, st);
			launch_pack_2bit(d_a.p, d_ao.p, d_len.p, d_wo.p, n_reads, n_words, d_w.p, st);
			HIP_OK(hipGetLastError());
			d_w.download(words, n_words, st);
			HIP_OK(hipStreamSynchronize(st));
		}
		(void)hipStreamDestroy(st);
		return (int64_t)n_words;
	} catch (...) {
		return -2;
	}
}

int64_t ndgpu_ovl_sketch(const ndgpu_ovl_opt *opt, uint32_t n_reads, const uint32_t *words, uint64_t n_words, const uint64_t *word_off,
                         const uint32_t *lens, int rid_is_index, uint64_t **x, uint64_t **y, uint64_t *off)
{
	*x = *y = nullptr;
	if (const char *msg = check_opt(*opt)) { fprintf(stderr, "[ndgpu_overlap] %s\n", msg); return -1; }
	try {
		int n_dev = 0;
		if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { fprintf(stderr, "[ndgpu_overlap] no HIP device\n"); return -1; }
		Engine e;
		if (const char *d = getenv("NDGPU_DEVICE")) e.device = atoi(d);
		HIP_OK(hipSetDevice(e.device));
		HIP_OK(ndovl::create_stage_stream(&e.stream));
		e.P = to_params(*opt);
		std::vector<uint32_t> ids(n_reads, 0);
		ReadSetDev R;
		R.upload(n_reads, words, n_words, word_off, lens, ids.data(), e.stream);
		Sketch S;
		e.sketch(R, rid_is_index, false, S);
		*x = (uint64_t*)malloc(8 * (S.n + 1)), *y = (uint64_t*)malloc(8 * (S.n + 1));
		S.x.download(*x, S.n, e.stream); S.y.download(*y, S.n, e.stream); S.off.download(off, n_reads + 1, e.stream);
		HIP_OK(hipStreamSynchronize(e.stream));
		const int64_t n = (int64_t)S.n;
		S = Sketch(); R = ReadSetDev(); e.tmp.release();
		(void)hipStreamDestroy(e.stream);
		return n;
	} catch (...) {
		return -2;
	}
}

void ndgpu_ovl_index_dump(const ndgpu_ovl_index *h, uint64_t *key, uint64_t *start, uint64_t *pos)
{
	const Engine &e = h->e;
	e.ukey.download(key, e.n_keys, e.stream);
	e.ustart.download(start, e.n_keys + 1, e.stream);
	e.pos.download(pos, e.n_min, e.stream);
	(void)hipStreamSynchronize(e.stream);
}

int64_t ndgpu_ovl_debug_anchors(ndgpu_ovl_index *h, uint32_t q, uint64_t **ax, uint64_t **ay, int32_t **f, int32_t **p)
{
	Engine &e = h->e;
	*ax = *ay = nullptr, *f = *p = nullptr;
	if (q < e.dbg_r0 || q >= e.dbg_r0 + e.dbg_n) return -1;
	const uint64_t a0 = e.dbg_aoff[q - e.dbg_r0], n = e.dbg_aoff[q - e.dbg_r0 + 1] - a0;
	*ax = (uint64_t*)malloc(8 * (n + 1)), *ay = (uint64_t*)malloc(8 * (n + 1));
	*f = (int32_t*)malloc(4 * (n + 1)), *p = (int32_t*)malloc(4 * (n + 1));
	e.dbg_ax.download(*ax, n, e.stream, a0); e.dbg_ay.download(*ay, n, e.stream, a0);
	e.dbg_f.download(*f, n, e.stream, a0); e.dbg_p.download(*p, n, e.stream, a0);
	(void)hipStreamSynchronize(e.stream);
	return (int64_t)n;
}

void ndgpu_ovl_get_stats(const ndgpu_ovl_index *h, ndgpu_ovl_stats *st) { *st = h->e.st; }
void ndgpu_ovl_reset_stats(ndgpu_ovl_index *h) { h->e.st = ndgpu_ovl_stats{}; }

} // extern "C"
