// ovl_kernels.hip -- gfx950 kernels of the overlap engine (`minimap2-nd --step 1` path).
//
//   K1  sketch_kernel        (w,k)-minimizers of every read            (minimap2/sketch.c:75-143)
//   K2  rocPRIM radix sort + run-length encode -> index arrays          (minimap2/index.c:197-250)
//   K3a seed_count_kernel    index lookup + surviving-hit count         (minimap2/map.c:91-152)
//   K3b seed_fill_kernel     anchors in generation order + sort key     (minimap2/map.c:214-246)
//   K3s rocPRIM radix sort of (read | strand | target | position) keys; anchor_decode_kernel flags
//       reads that hold equal keys; exact_sort_kernel replays the reference's unstable in-place
//       radix sort (minimap2/ksort.h:100-151) for exactly those reads
//   K4  chain_kernel         chaining DP, one wavefront per query read  (minimap2/chain.c:44-85)
//   K5  hits_kernel          chain ends, backtrack, ordering, hit coordinates, step-1 filter
//                            (minimap2/chain.c:87-162, hit.c:8-95, map.c:1296-1304)
//
// Integer / byte work, HBM- and latency-bound: nothing here is shaped like a GEMM.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "ovl_device.h"

namespace ndovl {

// ------------------------------------------------------------------------------------------------
// helpers

__device__ __forceinline__ uint64_t hash_masked(uint64_t key, uint64_t mask)
{
	key = (~key + (key << 21)) & mask;
	key ^= key >> 24;
	key = (key + (key << 3) + (key << 8)) & mask;
	key ^= key >> 14;
	key = (key + (key << 2) + (key << 4)) & mask;
	key ^= key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

__device__ __forceinline__ uint64_t hash_full(uint64_t key)
{
	key = ~key + (key << 21);
	key ^= key >> 24;
	key = key + (key << 3) + (key << 8);
	key ^= key >> 14;
	key = key + (key << 2) + (key << 4);
	key ^= key >> 28;
	key = key + (key << 31);
	return key;
}

// ------------------------------------------------------------------------------------------------
// K1: minimizer sketch.  One lane walks one read (the window automaton is sequential); reads are
// handed out longest-first so that the lanes of a wavefront finish together.  The w-slot ring lives
// in LDS, transposed so that lane l touches bank (l mod 32) only.

struct BaseReader {
	const uint32_t *w;
	uint32_t cur;
	int cur_idx;
	__device__ BaseReader(const uint32_t *p) : w(p), cur(0), cur_idx(-1) {}
	__device__ __forceinline__ int at(int i)
	{
		int wi = i >> 4;
		if (wi != cur_idx) cur = w[wi], cur_idx = wi;
		return (int)(cur >> (30 - 2 * (i & 15)) & 3u);
	}
};

// LONG: k > 28 -- mm_sketch_nextdenovo_longkmer (sketch.c:283-356) with the k-mer in up to four words (b2kmer, b2kmer_rc, kmer_cmp,
// hash256to64: sketch.c:219-281); the launcher sends here what the position-parallel kernels below do not take (even k, where a k-mer
// can equal its reverse complement and the window stands still; k = 29..31 and k > 63).
template <bool FILL, bool LONG>
__global__ void __launch_bounds__(64) sketch_kernel(const uint32_t *__restrict__ words, const uint64_t *__restrict__ woff,
                                                     const uint32_t *__restrict__ len, const uint32_t *__restrict__ order,
                                                     uint32_t n_reads, int w, int k, int hpc, int rid_is_index,
                                                     const uint64_t *__restrict__ out_off, uint64_t *__restrict__ out_x,
                                                     uint64_t *__restrict__ out_y, uint32_t *__restrict__ out_read,
                                                     uint32_t *__restrict__ out_cnt)
{
	extern __shared__ uint8_t smem[];
	uint64_t *ring_x = reinterpret_cast<uint64_t*>(smem);                      // [w][64]
	uint32_t *ring_y = reinterpret_cast<uint32_t*>(smem + (size_t)w * 64 * 8); // [w][64]
	uint16_t *runq = reinterpret_cast<uint16_t*>(smem + (size_t)w * 64 * 12);  // [32][64]
	const int lane = threadIdx.x;
	const uint32_t slot_r = blockIdx.x * 64u + lane;
	if (slot_r >= n_reads) return;
	const uint32_t r = order[slot_r];
	const int n = (int)len[r];
	const uint64_t rid_hi = rid_is_index ? (uint64_t)r << 32 : 0;
	uint64_t o = FILL ? out_off[r] : 0;
	uint32_t cnt = 0;
	if (n > 0) {
		BaseReader rd(words + woff[r]);
		const int k_idx = LONG ? (k - 1) / 32 : 0;
		const uint64_t mask = LONG ? (1ULL << 2 * (((k - 1) & 31) + 1)) - 1 : (1ULL << 2 * k) - 1;
		const uint64_t top = LONG ? (uint64_t)(((k - 1) & 31) << 1) : 2ULL * (k - 1);
		uint64_t fw = 0, rv = 0, best_x = ~0ULL;
		uint64_t F[4] = {0, 0, 0, 0}, R[4] = {0, 0, 0, 0};
		uint32_t best_y = ~0u;
		int good = 0, slot = 0, best_slot = 0, span = 0, rq_front = 0, rq_count = 0;
		for (int j = 0; j < w; ++j) ring_x[j * 64 + lane] = ~0ULL, ring_y[j * 64 + lane] = ~0u;
#define EMIT(X, Y) do { if (FILL) { out_x[o + cnt] = (X); out_y[o + cnt] = rid_hi | (uint64_t)(Y); if (out_read) out_read[o + cnt] = r; } ++cnt; } while (0)
		for (int i = 0; i < n; ++i) {
			const int c = rd.at(i);
			uint64_t cur_x = ~0ULL;
			uint32_t cur_y = ~0u;
			if (hpc) {
				int run = 1;
				if (i + 1 < n && rd.at(i + 1) == c) {
					for (run = 2; i + run < n; ++run)
						if (rd.at(i + run) != c) break;
					i += run - 1;
				}
				const int rc = run > 256 ? 256 : run; // only "span < 256" and spans below it are observable
				runq[((rq_count++ + rq_front) & 31) * 64 + lane] = (uint16_t)rc;
				span += rc;
				if (rq_count > k) { span -= runq[rq_front * 64 + lane]; rq_front = (rq_front + 1) & 31; --rq_count; }
			} else span = good + 1 < k ? good + 1 : k;
			int strand;
			if (LONG) {
				F[3] = F[3] << 2 | F[2] >> 62, F[2] = F[2] << 2 | F[1] >> 62, F[1] = F[1] << 2 | F[0] >> 62, F[0] = F[0] << 2 | (uint64_t)c;
				R[0] = R[0] >> 2 | R[1] << 62, R[1] = R[1] >> 2 | R[2] << 62, R[2] = R[2] >> 2 | R[3] << 62, R[3] >>= 2;
				int cmp = 0;
#pragma unroll
				for (int j = 3; j >= 0; --j) {
					if (j == k_idx) F[j] &= mask, R[j] |= (uint64_t)(3 ^ c) << top;
					if (!cmp) cmp = F[j] < R[j] ? -1 : F[j] > R[j] ? 1 : 0;
				}
				if (cmp == 0) continue;
				strand = cmp < 0 ? 0 : 1;
			} else {
				fw = (fw << 2 | (uint64_t)c) & mask;
				rv = rv >> 2 | (uint64_t)(3 ^ c) << top;
				if (fw == rv) continue;
				strand = fw < rv ? 0 : 1;
			}
			++good;
			if (good >= k && span < 256) {
				uint64_t h;
				if (LONG) {
					h = 0;
#pragma unroll
					for (int j = 3; j >= 0; --j) {
						const uint64_t kj = strand ? R[j] : F[j];
						if (j == k_idx) h = hash_masked(kj, mask);
						else if (j < k_idx && kj) h += hash_full(kj);
					}
				} else h = hash_masked(strand ? rv : fw, mask);
				cur_x = h << 8 | (uint64_t)span;
				cur_y = (uint32_t)i << 1 | (uint32_t)strand;
			}
			ring_x[slot * 64 + lane] = cur_x, ring_y[slot * 64 + lane] = cur_y;
			if (good == w + k - 1 && best_x != ~0ULL) {
				for (int j = slot + 1; j < w; ++j)
					if (ring_x[j * 64 + lane] == best_x && ring_y[j * 64 + lane] != best_y) EMIT(best_x, ring_y[j * 64 + lane]);
				for (int j = 0; j < slot; ++j)
					if (ring_x[j * 64 + lane] == best_x && ring_y[j * 64 + lane] != best_y) EMIT(best_x, ring_y[j * 64 + lane]);
			}
			if (cur_x <= best_x) {
				if (good >= w + k && best_x != ~0ULL) EMIT(best_x, best_y);
				best_x = cur_x, best_y = cur_y, best_slot = slot;
			} else if (slot == best_slot) {
				if (good >= w + k - 1 && best_x != ~0ULL) EMIT(best_x, best_y);
				best_x = ~0ULL;
				for (int j = slot + 1; j < w; ++j)
					if (ring_x[j * 64 + lane] <= best_x) best_x = ring_x[j * 64 + lane], best_y = ring_y[j * 64 + lane], best_slot = j;
				for (int j = 0; j <= slot; ++j)
					if (ring_x[j * 64 + lane] <= best_x) best_x = ring_x[j * 64 + lane], best_y = ring_y[j * 64 + lane], best_slot = j;
				if (good >= w + k - 1 && best_x != ~0ULL) {
					for (int j = slot + 1; j < w; ++j)
						if (ring_x[j * 64 + lane] == best_x && ring_y[j * 64 + lane] != best_y) EMIT(best_x, ring_y[j * 64 + lane]);
					for (int j = 0; j <= slot; ++j)
						if (ring_x[j * 64 + lane] == best_x && ring_y[j * 64 + lane] != best_y) EMIT(best_x, ring_y[j * 64 + lane]);
				}
			}
			if (++slot == w) slot = 0;
		}
		if (best_x != ~0ULL) EMIT(best_x, best_y);
#undef EMIT
	}
	if (!FILL) out_cnt[r] = cnt;
}

size_t sketch_smem(int w) { return (size_t)w * 64 * 12 + 32 * 64 * 2; }

void launch_sketch(bool fill, const uint32_t *words, const uint64_t *woff, const uint32_t *len, const uint32_t *order,
                   uint32_t n_reads, const OvlParams &P, int rid_is_index, const uint64_t *out_off, uint64_t *out_x,
                   uint64_t *out_y, uint32_t *out_read, uint32_t *out_cnt, hipStream_t s)
{
	if (!n_reads) return;
	dim3 grid((n_reads + 63) / 64), block(64);
	size_t sm = sketch_smem(P.w);
#define SK(F_, L_) ND_LAUNCH((sketch_kernel<F_, L_>), grid, block, sm, s, words, woff, len, order, n_reads, P.w, P.k, P.hpc, rid_is_index, \
                             out_off, out_x, out_y, out_read, out_cnt)
	if (P.k > 28) { if (fill) SK(true, true); else SK(false, true); }
	else { if (fill) SK(true, false); else SK(false, false); }
#undef SK
}

// ------------------------------------------------------------------------------------------------
// K1, position-parallel form (odd k: no k-mer equals its reverse complement, so every symbol advances
// the window).  What the window automaton of sketch.c emits is, in closed form:
//   * symbol s is emitted iff its value x_s is a minimum (ties included) of at least one full window of
//     w consecutive k-mers containing it, i.e. iff a + b >= w - 1 where a / b count the neighbours to
//     the left / right (at most w-1, inside the read) whose value is >= x_s;
//   * output order = position order;
//   * the first window is special (sketch.c:122-127 runs before the w-th k-mer is compared and the
//     replaced minimum is not written while l < w+k): with R' = rightmost minimum of the first w-1
//     k-mers and D' its equal-valued copies, D' is emitted too when x[w-1] < x[R'], and R' is NOT
//     emitted when x[w-1] == x[R'];
//   * a read with fewer than w k-mers emits only the rightmost minimum of all its k-mers.
// (tests/test_gpu_overlap.py checks this against the sequential restatement on every read, and
// tools/ + the oracle tests fuzz the rule on low-complexity reads.)
// With HPC the symbols are homopolymer runs: a first pass compacts each read into (run base, run start).

constexpr int kTS = 1024;          // symbols per tile
constexpr int kSkThreads = 256;
constexpr int kCH = kTS / kSkThreads;
constexpr int kHalo = 63;          // w - 1 <= 63
constexpr int kExt = kTS + 2 * kHalo;
constexpr int kECH = (kExt + kSkThreads - 1) / kSkThreads;

__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *wave_tot /*LDS[4]*/, uint32_t &total)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	uint32_t inc = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
	if (lane == 63) wave_tot[wv] = inc;
	__syncthreads();
	uint32_t base = 0;
	total = 0;
	for (int i = 0; i < kSkThreads / 64; ++i) { if (i < wv) base += wave_tot[i]; total += wave_tot[i]; }
	__syncthreads();
	return base + inc - v;
}

// HPC pre-pass over base tiles: run starts -> (run base, run start position) per read
template <bool FILL>
__global__ void __launch_bounds__(kSkThreads) run_compact_kernel(const uint32_t *__restrict__ words, const uint64_t *__restrict__ woff,
                                                                 const uint32_t *__restrict__ len, const SketchTile *__restrict__ tiles,
                                                                 uint32_t n_tiles, const uint64_t *__restrict__ tile_prefix,
                                                                 const uint32_t *__restrict__ first_tile, const uint64_t *__restrict__ roff,
                                                                 uint32_t *__restrict__ tile_cnt, uint8_t *__restrict__ sym,
                                                                 uint32_t *__restrict__ rstart, uint32_t *__restrict__ n_sym)
{
	__shared__ uint32_t wave_tot[kSkThreads / 64];
	const uint32_t tile = blockIdx.x;
	if (tile >= n_tiles) return;
	const uint32_t r = tiles[tile].read, t0 = tiles[tile].start, n = len[r];
	const uint32_t *w = words + woff[r];
	const uint32_t s0 = t0 + threadIdx.x * kCH;
	uint32_t flags = 0, cnt = 0;
	int codes[kCH];
	int prev = s0 > 0 && s0 - 1 < n ? (int)(w[(s0 - 1) >> 4] >> (30 - 2 * ((s0 - 1) & 15)) & 3u) : -1;
#pragma unroll
	for (int c = 0; c < kCH; ++c) {
		const uint32_t i = s0 + c;
		codes[c] = -1;
		if (i < n) {
			const int b = (int)(w[i >> 4] >> (30 - 2 * (i & 15)) & 3u);
			codes[c] = b;
			if (b != prev) flags |= 1u << c, ++cnt;
			prev = b;
		}
	}
	uint32_t total;
	const uint32_t excl = block_excl_scan(cnt, wave_tot, total);
	if (!FILL) {
		if (threadIdx.x == 0) tile_cnt[tile] = total;
		return;
	}
	const uint64_t base = roff[r] + (tile_prefix[tile] - tile_prefix[first_tile[r]]) + excl;
	uint32_t k = 0;
#pragma unroll
	for (int c = 0; c < kCH; ++c)
		if (flags >> c & 1) { sym[base + k] = (uint8_t)codes[c]; rstart[base + k] = s0 + c; ++k; }
	if (threadIdx.x == 0 && t0 + kTS >= n) { // last tile of the read: sentinel + run count
		const uint32_t ns = (uint32_t)(tile_prefix[tile + 1] - tile_prefix[first_tile[r]]);
		rstart[roff[r] + ns] = n;
		n_sym[r] = ns;
	}
}

void launch_run_compact(bool fill, const uint32_t *words, const uint64_t *woff, const uint32_t *len, const SketchTile *tiles,
                        uint32_t n_tiles, const uint64_t *tile_prefix, const uint32_t *first_tile, const uint64_t *roff, uint32_t *tile_cnt,
                        uint8_t *sym, uint32_t *rstart, uint32_t *n_sym, hipStream_t s)
{
	if (!n_tiles) return;
	if (fill) ND_LAUNCH(run_compact_kernel<true>, dim3(n_tiles), dim3(kSkThreads), 0, s, words, woff, len, tiles, n_tiles, tile_prefix,
	                             first_tile, roff, tile_cnt, sym, rstart, n_sym);
	else ND_LAUNCH(run_compact_kernel<false>, dim3(n_tiles), dim3(kSkThreads), 0, s, words, woff, len, tiles, n_tiles, tile_prefix,
	                        first_tile, roff, tile_cnt, sym, rstart, n_sym);
}

// LONGK: 33 <= k <= 63 (the ava-hifi preset's k = 51, sketch.c:283-356): the k-mer takes two words and its value is
// hash64(top word) + hash64_no_mask(low word) (hash256to64, sketch.c:274-281); with HPC the span is what the reference's
// 32-slot run-length ring (sketch.c:40-58) leaves once more than 32 runs are queued: the first 32 runs of the read plus
// the last k mod 32.
template <bool FILL, bool HPC, bool LONGK>
__global__ void __launch_bounds__(kSkThreads) sketch_tile_kernel(const uint32_t *__restrict__ words, const uint64_t *__restrict__ woff,
                                                                 const uint32_t *__restrict__ len, const uint8_t *__restrict__ sym,
                                                                 const uint32_t *__restrict__ rstart, const uint64_t *__restrict__ roff,
                                                                 const uint32_t *__restrict__ n_sym, const SketchTile *__restrict__ tiles,
                                                                 uint32_t n_tiles, int w, int k, int rid_is_index,
                                                                 const uint64_t *__restrict__ tile_off, uint32_t *__restrict__ tile_cnt,
                                                                 uint64_t *__restrict__ out_x, uint64_t *__restrict__ out_y,
                                                                 uint32_t *__restrict__ out_read)
{
	__shared__ uint64_t xs[kExt];
	__shared__ uint32_t ys[kExt];
	__shared__ uint32_t wave_tot[kSkThreads / 64];
	__shared__ unsigned long long force_mask;
	__shared__ int clear_m, short_emit; // short_emit: -2 = normal read, -1 = short read without output, >= 0 symbol to emit
	const uint32_t tile = blockIdx.x;
	if (tile >= n_tiles) return;
	const uint32_t r = tiles[tile].read;
	const int t0 = (int)tiles[tile].start;
	const int N = HPC ? (int)n_sym[r] : (int)len[r];
	const int hw = w - 1, lo = k - 1;
	const int e0 = t0 - hw > 0 ? t0 - hw : 0;
	const int e1 = t0 + kTS + hw < N ? t0 + kTS + hw : N;
	const uint64_t mask = (1ULL << (LONGK ? 2 * (k - 32) : 2 * k)) - 1, top = LONGK ? 2ULL * (k - 1) - 64 : 2ULL * (k - 1);
	const uint32_t *wp = words + woff[r];
	const uint64_t ro = HPC ? roff[r] : 0;

	// values of the extended range, a contiguous chunk per thread with a rolling k-mer
	{
		const int first = e0 + (int)threadIdx.x * kECH;
		int last = first + kECH; // exclusive
		if (last > e1) last = e1;
		if (first < last) {
			int j = first - (k - 1);
			if (j < 0) j = 0;
			uint64_t fw = 0, rv = 0, fw_lo = 0, rv_lo = 0;
			for (; j < last; ++j) {
				int c;
				if (HPC) c = (int)sym[ro + j];
				else c = (int)(wp[j >> 4] >> (30 - 2 * (j & 15)) & 3u);
				if (LONGK) {
					fw = (fw << 2 | fw_lo >> 62) & mask;
					fw_lo = fw_lo << 2 | (uint64_t)c;
					rv_lo = rv_lo >> 2 | rv << 62;
					rv = rv >> 2 | (uint64_t)(3 ^ c) << top;
				} else {
					fw = (fw << 2 | (uint64_t)c) & mask;
					rv = rv >> 2 | (uint64_t)(3 ^ c) << top;
				}
				if (j >= first) {
					uint64_t x = ~0ULL;
					uint32_t y = ~0u;
					if (j >= lo) {
						uint32_t pos = (uint32_t)j, span = (uint32_t)k;
						if (HPC) {
							const uint32_t nxt = rstart[ro + j + 1];
							pos = nxt - 1;
							if (LONGK && j >= k) span = rstart[ro + 32] - rstart[ro] + nxt - rstart[ro + j + 1 - (k & 31)];
							else span = nxt - rstart[ro + j + 1 - k];
						}
						if (span < 256) {
							int strand;
							uint64_t h;
							if (LONGK) {
								strand = (fw < rv || (fw == rv && fw_lo < rv_lo)) ? 0 : 1;
								const uint64_t l_ = strand ? rv_lo : fw_lo;
								h = hash_masked(strand ? rv : fw, mask);
								if (l_) h += hash_full(l_);
							} else {
								strand = fw < rv ? 0 : 1;
								h = hash_masked(strand ? rv : fw, mask);
							}
							x = h << 8 | (uint64_t)span;
							y = pos << 1 | (uint32_t)strand;
						}
					}
					xs[j - e0] = x, ys[j - e0] = y;
				}
			}
		}
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		force_mask = 0, clear_m = -1, short_emit = -2;
		if (t0 == 0) {
			const int nk = N - lo;
			if (nk < w) { // fewer k-mers than one window: only the final minimum is written (sketch.c:141-142)
				short_emit = -1;
				uint64_t best = ~0ULL;
				for (int s = lo; s < N; ++s) if (xs[s - e0] <= best && xs[s - e0] != ~0ULL) best = xs[s - e0], short_emit = s;
			} else if (w >= 2) {
				uint64_t best = ~0ULL;
				int rp = -1;
				for (int m = 0; m < w - 1; ++m) if (xs[lo + m - e0] <= best) best = xs[lo + m - e0], rp = m;
				if (best != ~0ULL) {
					const uint64_t xw = xs[lo + w - 1 - e0];
					if (xw < best) {
						for (int m = 0; m < w - 1; ++m) if (m != rp && xs[lo + m - e0] == best) force_mask |= 1ULL << m;
					} else if (xw == best) clear_m = rp;
				}
			}
		}
	}
	__syncthreads();
	uint32_t flags = 0, cnt = 0;
	const int s0 = t0 + (int)threadIdx.x * kCH;
#pragma unroll
	for (int c = 0; c < kCH; ++c) {
		const int s = s0 + c;
		if (s >= N || s < lo) continue;
		bool emit;
		if (short_emit != -2) emit = s == short_emit;
		else {
			const uint64_t x = xs[s - e0];
			emit = false;
			if (x != ~0ULL) {
				int a = 0, b = 0;
				for (int t = s - 1; t >= lo && a < hw && xs[t - e0] >= x; --t) ++a;
				for (int t = s + 1; t < N && b < hw && xs[t - e0] >= x; ++t) ++b;
				emit = a + b >= hw;
			}
			const int m = s - lo;
			if (t0 == 0 && m < hw) {
				if (force_mask >> m & 1) emit = true;
				if (m == clear_m) emit = false;
			}
		}
		if (emit) flags |= 1u << c, ++cnt;
	}
	uint32_t total;
	const uint32_t excl = block_excl_scan(cnt, wave_tot, total);
	if (!FILL) {
		if (threadIdx.x == 0) tile_cnt[tile] = total;
		return;
	}
	uint64_t o = tile_off[tile] + excl;
	const uint64_t rid_hi = rid_is_index ? (uint64_t)r << 32 : 0;
#pragma unroll
	for (int c = 0; c < kCH; ++c)
		if (flags >> c & 1) {
			const int s = s0 + c;
			out_x[o] = xs[s - e0];
			out_y[o] = rid_hi | (uint64_t)ys[s - e0];
			if (out_read) out_read[o] = r;
			++o;
		}
}

void launch_sketch_tiles(bool fill, bool hpc, const uint32_t *words, const uint64_t *woff, const uint32_t *len, const uint8_t *sym,
                         const uint32_t *rstart, const uint64_t *roff, const uint32_t *n_sym, const SketchTile *tiles, uint32_t n_tiles,
                         const OvlParams &P, int rid_is_index, const uint64_t *tile_off, uint32_t *tile_cnt, uint64_t *out_x, uint64_t *out_y,
                         uint32_t *out_read, hipStream_t s)
{
	if (!n_tiles) return;
	dim3 g(n_tiles), b(kSkThreads);
#define SK_ARGS words, woff, len, sym, rstart, roff, n_sym, tiles, n_tiles, P.w, P.k, rid_is_index, tile_off, tile_cnt, out_x, out_y, out_read
	if (P.k > 32) {
		if (fill && hpc) ND_LAUNCH((sketch_tile_kernel<true, true, true>), g, b, 0, s, SK_ARGS);
		else if (fill) ND_LAUNCH((sketch_tile_kernel<true, false, true>), g, b, 0, s, SK_ARGS);
		else if (hpc) ND_LAUNCH((sketch_tile_kernel<false, true, true>), g, b, 0, s, SK_ARGS);
		else ND_LAUNCH((sketch_tile_kernel<false, false, true>), g, b, 0, s, SK_ARGS);
	} else if (fill && hpc) ND_LAUNCH((sketch_tile_kernel<true, true, false>), g, b, 0, s, SK_ARGS);
	else if (fill) ND_LAUNCH((sketch_tile_kernel<true, false, false>), g, b, 0, s, SK_ARGS);
	else if (hpc) ND_LAUNCH((sketch_tile_kernel<false, true, false>), g, b, 0, s, SK_ARGS);
	else ND_LAUNCH((sketch_tile_kernel<false, false, false>), g, b, 0, s, SK_ARGS);
#undef SK_ARGS
}

int sketch_tile_symbols() { return kTS; }

// off[r] = tile_off[first_tile[r]] for r in 0..n_reads
__global__ void gather_u64_kernel(const uint64_t *__restrict__ src, const uint32_t *__restrict__ idx, uint32_t n, uint64_t *__restrict__ dst)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) dst[i] = src[idx[i]];
}

// ------------------------------------------------------------------------------------------------
// seq_dump's 2-bit packing (seq2bit, lib/bseq.c:114-139) for a batch of reads: one thread per output word, 16 input
// bytes each.  A byte that is not ACGTU (either case) has code 4 and is OR-ed in like the others, so its bit 2 lands on
// the low bit of the base before it (and is lost at the start of a word), exactly as `buffer << 2 | nt_table[c]` does.
__global__ void pack_2bit_kernel(const uint8_t *__restrict__ ascii, const uint64_t *__restrict__ a_off, const uint32_t *__restrict__ len,
                                 const uint64_t *__restrict__ w_off, uint32_t n_reads, uint64_t n_words, uint32_t *__restrict__ words)
{
	const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= n_words) return;
	uint32_t lo = 0, hi = n_reads; // last read whose first word is <= w
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (w_off[mid] <= w) lo = mid; else hi = mid; }
	const uint32_t r = lo, n = len[r];
	const uint32_t b0 = (uint32_t)(w - w_off[r]) * 16u;
	const uint8_t *s = ascii + a_off[r] + b0;
	const uint32_t cnt = n - b0 < 16u ? n - b0 : 16u;
	uint64_t acc = 0;
	for (uint32_t j = 0; j < cnt; ++j) {
		const uint8_t c = s[j];
		uint32_t code = 4;
		switch (c) {
		case 'A': case 'a': code = 0; break;
		case 'C': case 'c': code = 1; break;
		case 'G': case 'g': code = 2; break;
		case 'T': case 't': case 'U': case 'u': code = 3; break;
		default: break;
		}
		acc |= (uint64_t)code << (30 - 2 * (int)j);
	}
	words[w] = (uint32_t)acc;
}

void launch_pack_2bit(const uint8_t *ascii, const uint64_t *a_off, const uint32_t *len, const uint64_t *w_off, uint32_t n_reads, uint64_t n_words,
                      uint32_t *words, hipStream_t s)
{
	if (!n_words) return;
	ND_LAUNCH(pack_2bit_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, s, ascii, a_off, len, w_off, n_reads, n_words,
	                   words);
}

void launch_gather_u64(const uint64_t *src, const uint32_t *idx, uint32_t n, uint64_t *dst, hipStream_t s)
{
	if (n) ND_LAUNCH(gather_u64_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, idx, n, dst);
}

// ------------------------------------------------------------------------------------------------
// small utility kernels

__global__ void shift_keys_kernel(const uint64_t *__restrict__ x, uint64_t *__restrict__ key, uint64_t n)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) key[i] = x[i] >> 8;
}

void launch_shift_keys(const uint64_t *x, uint64_t *key, uint64_t n, hipStream_t s)
{
	if (n) ND_LAUNCH(shift_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, key, n);
}

// ------------------------------------------------------------------------------------------------
// rocPRIM wrappers (plain library primitives: LSD radix sort, run-length encode, scans)

// (the size query -- tmp == nullptr -- touches nothing; the call that works is a checked device operation)
#define RP_CHECK(e) do { if (tmp && ndovl::fault_injected()) ndovl::device_check((int)hipErrorOutOfMemory, __func__); ndovl::device_check((int)(e), __func__); } while (0)

int sort_pairs_u64(void *tmp, size_t &tmp_bytes, const uint64_t *kin, uint64_t *kout, const uint64_t *vin, uint64_t *vout,
                   size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t s)
{
	RP_CHECK(rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, begin_bit, end_bit, s));
	return 0;
}

int sort_keys_u32(void *tmp, size_t &tmp_bytes, const uint32_t *kin, uint32_t *kout, size_t n, hipStream_t s)
{
	RP_CHECK(rocprim::radix_sort_keys(tmp, tmp_bytes, kin, kout, n, 0, 32, s));
	return 0;
}

int rle_u64(void *tmp, size_t &tmp_bytes, const uint64_t *kin, size_t n, uint64_t *uniq, uint32_t *cnt, uint64_t *n_runs,
            hipStream_t s)
{
	RP_CHECK(rocprim::run_length_encode(tmp, tmp_bytes, kin, (unsigned int)n, uniq, cnt, n_runs, s));
	return 0;
}

int exscan_u32_to_u64(void *tmp, size_t &tmp_bytes, const uint32_t *in, uint64_t *out, size_t n, hipStream_t s)
{
	auto it = rocprim::make_transform_iterator(in, [] __device__(uint32_t v) { return (uint64_t)v; });
	RP_CHECK(rocprim::exclusive_scan(tmp, tmp_bytes, it, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), s));
	return 0;
}

// ------------------------------------------------------------------------------------------------
// K3: seeds

// bucket[h] = number of keys below h << shift (one binary search per table entry, at index build)
__global__ void build_buckets_kernel(const uint64_t *__restrict__ ukey, uint64_t n_keys, uint32_t shift, uint32_t *__restrict__ bucket)
{
	const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
	if (h > (1u << kBucketBits)) return;
	const uint64_t want = (uint64_t)h << shift;
	uint64_t lo = 0, hi = n_keys;
	while (lo < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		if (ukey[mid] < want) lo = mid + 1; else hi = mid;
	}
	bucket[h] = (uint32_t)lo;
}

void launch_build_buckets(const uint64_t *ukey, uint64_t n_keys, uint32_t shift, uint32_t *bucket, hipStream_t s)
{
	ND_LAUNCH(build_buckets_kernel, dim3(((1u << kBucketBits) + 256) / 256), dim3(256), 0, s, ukey, n_keys, shift, bucket);
}

// the distinct keys into the open-addressing table (index build): a slot is claimed by compare-and-swap on its key word
__global__ void build_hash_kernel(const uint64_t *__restrict__ ukey, const uint64_t *__restrict__ ustart, uint64_t n_keys, HashSlot *__restrict__ tab,
                                  uint64_t size)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_keys) return;
	const unsigned long long k1 = ukey[i] + 1ull;
	uint64_t s = hash_slot_of(ukey[i], size);
	for (;;) {
		const unsigned long long found = atomicCAS(&tab[s].key1, 0ull, k1);
		if (found == 0ull) break;
		s = s + 1 == size ? 0 : s + 1;
	}
	tab[s].start = (uint32_t)ustart[i];
	tab[s].cnt = (uint32_t)(ustart[i + 1] - ustart[i]);
}

void launch_build_hash(const uint64_t *ukey, const uint64_t *ustart, uint64_t n_keys, HashSlot *tab, uint64_t size, hipStream_t s)
{
	if (!n_keys) return;
	ND_LAUNCH(build_hash_kernel, dim3((unsigned)((n_keys + 255) / 256)), dim3(256), 0, s, ukey, ustart, n_keys, tab, size);
}

__device__ __forceinline__ bool index_lookup(const IndexDev &ix, uint64_t minier, uint32_t &start, uint32_t &cnt)
{
	if (ix.htab) {
		const unsigned long long want = minier + 1ull;
		for (uint64_t s = hash_slot_of(minier, ix.hsize);; s = s + 1 == ix.hsize ? 0 : s + 1) {
			const HashSlot e = ix.htab[s];
			if (e.key1 == want) { start = e.start, cnt = e.cnt; return true; }
			if (e.key1 == 0ull) { start = 0, cnt = 0; return false; }
		}
	}
	const uint32_t h = (uint32_t)(minier >> ix.bucket_shift);
	uint64_t lo = ix.bucket[h], hi = ix.bucket[h + 1];
	while (lo < hi) {
		uint64_t mid = (lo + hi) >> 1;
		if (ix.ukey[mid] < minier) lo = mid + 1; else hi = mid;
	}
	if (lo == ix.n_keys || ix.ukey[lo] != minier) { start = 0, cnt = 0; return false; }
	start = (uint32_t)ix.ustart[lo];
	cnt = (uint32_t)(ix.ustart[lo + 1] - ix.ustart[lo]);
	return true;
}

// skip_seed(): 0 = keep, 1 = drop; *self as the reference's is_self
__device__ __forceinline__ int seed_skipped(const OvlParams &P, const IndexDev &ix, uint64_t r, uint32_t q_pos, uint64_t q_namekey,
                                            uint32_t q_len, int *self)
{
	*self = 0;
	if (P.no_diag || P.no_dual) {
		const uint32_t rid = (uint32_t)(r >> 32);
		const uint64_t tk = ix.namekey[rid];
		if (P.no_diag && q_namekey == tk && ix.len[rid] == q_len) {
			if ((uint32_t)r >> 1 == q_pos >> 1) return 1;
			if ((r & 1) == (q_pos & 1)) *self = 1;
		}
		if (P.no_dual && q_namekey > tk) return 1;
	}
	return 0;
}

// occurrences [lo, hi) of read `rid` inside one key's position list (ascending read << 32 | position << 1 | strand)
__device__ __forceinline__ void occ_run(const uint64_t *__restrict__ pos, uint32_t start, uint32_t cnt, uint32_t rid, uint32_t &lo, uint32_t &hi)
{
	uint32_t a = 0, b = cnt;
	while (a < b) { const uint32_t m = (a + b) >> 1; if ((uint32_t)(pos[start + m] >> 32) < rid) a = m + 1; else b = m; }
	lo = a, b = cnt;
	while (a < b) { const uint32_t m = (a + b) >> 1; if ((uint32_t)(pos[start + m] >> 32) <= rid) a = m + 1; else b = m; }
	hi = a;
}

__global__ void seed_count_kernel(const uint64_t *__restrict__ mx, const uint64_t *__restrict__ my, const uint32_t *__restrict__ m_read,
                                  uint64_t n_m, IndexDev ix, QueryDev q, OvlParams P, int mid_occ, uint32_t *__restrict__ m_start,
                                  uint32_t *__restrict__ m_cnt, uint32_t *__restrict__ m_surv)
{
	uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (m >= n_m) return;
	uint32_t start, cnt, surv = 0;
	if (q.read_mid) mid_occ = q.read_mid[m_read[m]];
	index_lookup(ix, mx[m] >> 8, start, cnt);
	if (q.want) { // re-alignment: the index the reference looks this minimizer up in holds the wanted reads only
		const uint32_t rd = m_read[m];
		uint32_t n_in = 0;
		if (cnt)
			for (uint64_t i = q.want_off[rd]; i < q.want_off[rd + 1]; ++i) {
				uint32_t lo, hi;
				occ_run(ix.pos, start, cnt, q.want[i], lo, hi);
				n_in += hi - lo;
			}
		if ((int64_t)n_in >= (int64_t)mid_occ) {
			n_in = 0;
			if (q.rep) q.rep[rd] = 1u;
		}
		m_start[m] = start, m_cnt[m] = n_in ? cnt : 0, m_surv[m] = n_in;
		return;
	}
	if ((int64_t)cnt >= (int64_t)mid_occ) { // repetitive minimizer: contributes nothing
		cnt = 0;
		if (q.rep) q.rep[m_read[m]] = 1u;
	}
	if (cnt) {
		const uint32_t rd = m_read[m], q_pos = (uint32_t)my[m];
		const uint64_t qk = q.namekey[rd];
		const uint32_t ql = q.len[rd];
		if (P.no_diag || P.no_dual) {
			for (uint32_t j = 0; j < cnt; ++j) {
				int self;
				surv += !seed_skipped(P, ix, ix.pos[start + j], q_pos, qk, ql, &self);
			}
		} else surv = cnt;
	}
	m_start[m] = start, m_cnt[m] = cnt, m_surv[m] = surv;
}

__global__ void seed_fill_kernel(const uint64_t *__restrict__ mx, const uint64_t *__restrict__ my, const uint32_t *__restrict__ m_read,
                                 uint64_t m0, uint64_t m1, IndexDev ix, QueryDev q, OvlParams P, const uint32_t *__restrict__ m_start,
                                 const uint32_t *__restrict__ m_cnt, const uint64_t *__restrict__ a_off, uint64_t a_base, KeyLayout L,
                                 uint64_t *__restrict__ ckey, uint64_t *__restrict__ ay)
{
	uint64_t m = m0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (m >= m1) return;
	const uint32_t cnt = m_cnt[m];
	if (!cnt) return;
	const uint32_t start = m_start[m], rd = m_read[m];
	const uint64_t minier = mx[m] >> 8;
	const uint32_t q_pos = (uint32_t)my[m], q_span = (uint32_t)(mx[m] & 0xff);
	const uint64_t qk = q.namekey[rd];
	const uint32_t ql = q.len[rd];
	// neighbours inside the same read carrying the same minimizer -> MM_SEED_TANDEM
	bool tandem = false;
	if (m > q.m_off[rd] && mx[m - 1] >> 8 == minier) tandem = true;
	if (m + 1 < q.m_off[rd + 1] && mx[m + 1] >> 8 == minier) tandem = true;
	uint64_t o = a_off[m] - a_base;
	const uint64_t rd_bits = (uint64_t)(rd - L.read_base) << L.read_shift;
	if (q.want) { // re-alignment: occurrences in the order of the mini-index (wanted read by wanted read), numbered by list position
		for (uint64_t i = q.want_off[rd]; i < q.want_off[rd + 1]; ++i) {
			uint32_t lo, hi;
			occ_run(ix.pos, start, cnt, q.want[i], lo, hi);
			for (uint32_t j = lo; j < hi; ++j) {
				const uint64_t r = ix.pos[start + j];
				const uint64_t rid = i - q.want_off[rd], rpos = (uint32_t)r >> 1;
				uint64_t y, rev;
				if ((r & 1) == (q_pos & 1)) {
					rev = 0;
					y = (uint64_t)q_span << 32 | (uint64_t)(q_pos >> 1);
				} else {
					rev = 1;
					y = (uint64_t)q_span << 32 | (uint64_t)(uint32_t)((int32_t)ql - (int32_t)((q_pos >> 1) + 1 - q_span) - 1);
				}
				if (tandem) y |= kSeedTandem;
				ckey[o] = rd_bits | rev << L.rev_shift | rid << L.pos_bits | rpos;
				ay[o] = y;
				++o;
			}
		}
		return;
	}
	for (uint32_t j = 0; j < cnt; ++j) {
		const uint64_t r = ix.pos[start + j];
		int self;
		if (seed_skipped(P, ix, r, q_pos, qk, ql, &self)) continue;
		const uint64_t rid = r >> 32, rpos = (uint32_t)r >> 1;
		uint64_t y, rev;
		if ((r & 1) == (q_pos & 1)) {
			rev = 0;
			y = (uint64_t)q_span << 32 | (uint64_t)(q_pos >> 1);
		} else {
			rev = 1;
			y = (uint64_t)q_span << 32 | (uint64_t)(uint32_t)((int32_t)ql - (int32_t)((q_pos >> 1) + 1 - q_span) - 1);
		}
		if (tandem) y |= kSeedTandem;
		if (self) y |= kSeedSelf;
		ckey[o] = rd_bits | rev << L.rev_shift | rid << L.pos_bits | rpos;
		ay[o] = y;
		++o;
	}
}

void launch_seed_count(const uint64_t *mx, const uint64_t *my, const uint32_t *m_read, uint64_t n_m, const IndexDev &ix,
                       const QueryDev &q, const OvlParams &P, int mid_occ, uint32_t *m_start, uint32_t *m_cnt, uint32_t *m_surv,
                       hipStream_t s)
{
	if (n_m) ND_LAUNCH(seed_count_kernel, dim3((unsigned)((n_m + 255) / 256)), dim3(256), 0, s, mx, my, m_read, n_m, ix, q, P,
	                            mid_occ, m_start, m_cnt, m_surv);
}

void launch_seed_fill(const uint64_t *mx, const uint64_t *my, const uint32_t *m_read, uint64_t m0, uint64_t m1, const IndexDev &ix,
                      const QueryDev &q, const OvlParams &P, const uint32_t *m_start, const uint32_t *m_cnt, const uint64_t *a_off,
                      uint64_t a_base, const KeyLayout &L, uint64_t *ckey, uint64_t *ay, hipStream_t s)
{
	if (m1 > m0) ND_LAUNCH(seed_fill_kernel, dim3((unsigned)((m1 - m0 + 255) / 256)), dim3(256), 0, s, mx, my, m_read, m0, m1, ix, q,
	                                P, m_start, m_cnt, a_off, a_base, L, ckey, ay);
}

// anchor offset of every read (= a_off at the read's first minimizer), n_reads + 1 entries
__global__ void gather_read_off_kernel(const uint64_t *__restrict__ a_off, const uint64_t *__restrict__ m_off, uint32_t n_reads, uint64_t n_m,
                                       uint64_t total, uint64_t *__restrict__ r_aoff)
{
	uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n_reads) return;
	const uint64_t m = m_off[r];
	r_aoff[r] = m < n_m ? a_off[m] : total;
}

void launch_gather_read_off(const uint64_t *a_off, const uint64_t *m_off, uint32_t n_reads, uint64_t n_m, uint64_t total,
                            uint64_t *r_aoff, hipStream_t s)
{
	ND_LAUNCH(gather_read_off_kernel, dim3((n_reads + 256) / 256), dim3(256), 0, s, a_off, m_off, n_reads, n_m, total, r_aoff);
}

__global__ void local_off_kernel(const uint64_t *__restrict__ all, uint32_t r0, uint32_t n, uint64_t *__restrict__ out)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i <= n) out[i] = all[r0 + i] - all[r0];
}

void launch_local_off(const uint64_t *r_aoff_all, uint32_t r0, uint32_t n, uint64_t *r_aoff, hipStream_t s)
{
	ND_LAUNCH(local_off_kernel, dim3((n + 256) / 256), dim3(256), 0, s, r_aoff_all, r0, n, r_aoff);
}

// sorted keys -> anchor x; a read is flagged when two neighbouring anchors of it carry the same key
__device__ __forceinline__ uint64_t key_to_x(uint64_t key, const KeyLayout &L)
{
	const uint64_t rpos = key & ((1ULL << L.pos_bits) - 1);
	const uint64_t rid = key >> L.pos_bits & ((1ULL << (L.rev_shift - L.pos_bits)) - 1);
	const uint64_t rev = key >> L.rev_shift & 1;
	return rev << 63 | rid << 32 | rpos;
}

// segval[i] = i + 1 where a (query read, strand, target read) segment starts, else 0: an inclusive max-scan of it
// gives every anchor the start of its segment.  Chains never leave a segment (chain.c:51: ri > a[st].x + max_dist).
__global__ void anchor_decode_kernel(const uint64_t *__restrict__ skey, uint64_t n, KeyLayout L, uint64_t *__restrict__ ax,
                                     uint32_t *__restrict__ tie_flag, uint64_t *__restrict__ segval)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t key = skey[i];
	ax[i] = key_to_x(key, L);
	const uint64_t prev = i > 0 ? skey[i - 1] : ~key;
	if (i > 0 && prev == key) tie_flag[(uint32_t)(key >> L.read_shift)] = 1; // benign race: all writers store 1
	segval[i] = (i == 0 || prev >> L.pos_bits != key >> L.pos_bits) ? i + 1 : 0;
}

void launch_anchor_decode(const uint64_t *skey, uint64_t n, const KeyLayout &L, uint64_t *ax, uint32_t *tie_flag, uint64_t *segval,
                          hipStream_t s)
{
	if (n) ND_LAUNCH(anchor_decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, skey, n, L, ax, tie_flag, segval);
}

int incl_max_scan_u64(void *tmp, size_t &tmp_bytes, const uint64_t *in, uint64_t *out, size_t n, hipStream_t s)
{
	RP_CHECK(rocprim::inclusive_scan(tmp, tmp_bytes, in, out, n, rocprim::maximum<uint64_t>(), s));
	return 0;
}

// K4 work units ("slabs"): runs of whole segments of one read, cut where the running anchor count since the read's
// start passes a multiple of kSlab.  flag[i] = 1 on the first anchor of a slab.
constexpr uint32_t kSlabShift = 10;

__global__ void slab_flag_kernel(const uint64_t *__restrict__ skey, const uint64_t *__restrict__ segstart1, uint64_t n, KeyLayout L,
                                 const uint64_t *__restrict__ r_aoff, uint32_t *__restrict__ flag)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t rl = (uint32_t)(skey[i] >> L.read_shift);
	const uint64_t a0 = r_aoff[rl];
	uint32_t f = 0;
	if (i == a0) f = 1;
	else {
		const uint64_t sid = (segstart1[i] - 1 - a0) >> kSlabShift, pid = (segstart1[i - 1] - 1 - a0) >> kSlabShift;
		f = sid != pid;
	}
	flag[i] = f;
}

__global__ void slab_write_kernel(const uint64_t *__restrict__ skey, const uint32_t *__restrict__ flag, const uint64_t *__restrict__ rank,
                                  uint64_t n, KeyLayout L, uint64_t *__restrict__ slab_i0, uint32_t *__restrict__ slab_read)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n || !flag[i]) return;
	slab_i0[rank[i]] = i;
	slab_read[rank[i]] = (uint32_t)(skey[i] >> L.read_shift);
}

void launch_slab_flag(const uint64_t *skey, const uint64_t *segstart1, uint64_t n, const KeyLayout &L, const uint64_t *r_aoff,
                      uint32_t *flag, hipStream_t s)
{
	if (n) ND_LAUNCH(slab_flag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, skey, segstart1, n, L, r_aoff, flag);
}

void launch_slab_write(const uint64_t *skey, const uint32_t *flag, const uint64_t *rank, uint64_t n, const KeyLayout &L,
                       uint64_t *slab_i0, uint32_t *slab_read, hipStream_t s)
{
	if (n) ND_LAUNCH(slab_write_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, skey, flag, rank, n, L, slab_i0,
	                          slab_read);
}

// average seed span of every read, as the reference's float quotient (chain.c:41-42)
__global__ void __launch_bounds__(64) read_span_kernel(const uint64_t *__restrict__ r_aoff, uint32_t n_reads, const uint64_t *__restrict__ ay,
                                                        float *__restrict__ avg_span)
{
	const uint32_t rd = blockIdx.x;
	if (rd >= n_reads) return;
	const uint64_t a0 = r_aoff[rd];
	const int64_t n = (int64_t)(r_aoff[rd + 1] - a0);
	unsigned long long sum = 0;
	for (int64_t i = threadIdx.x; i < n; i += 64) sum += ay[a0 + i] >> 32 & 0xff;
	for (int d = 32; d; d >>= 1) sum += __shfl_xor(sum, d, 64);
	if (threadIdx.x == 0) avg_span[rd] = n ? (float)((double)(float)sum / (double)(float)n) : 0.f;
}

void launch_read_span(const uint64_t *r_aoff, uint32_t n_reads, const uint64_t *ay, float *avg_span, hipStream_t s)
{
	if (n_reads) ND_LAUNCH(read_span_kernel, dim3(n_reads), dim3(64), 0, s, r_aoff, n_reads, ay, avg_span);
}

// ------------------------------------------------------------------------------------------------
// The reference's radix_sort_128x, replayed step by step (ksort.h:100-151): in-place MSD "American
// flag" passes of 8 bits starting at bit 56, buckets of <= 64 elements finished by insertion sort.
// It is not stable, and the order it leaves equal keys in reaches the chaining DP, so reads that own
// equal keys are sorted by this code instead of the LSD sort.  Sequential by nature (each step
// depends on the element just displaced): one lane runs it, tables in LDS or scratch.

struct SortJob { uint32_t beg, end; int32_t shift; };

__device__ void insertion_sort_xy(uint64_t *x, uint64_t *y, uint32_t beg, uint32_t end)
{
	for (uint32_t i = beg + 1; i < end; ++i) {
		if (x[i] < x[i - 1]) {
			const uint64_t tx = x[i], ty = y[i];
			uint32_t j = i;
			for (; j > beg && tx < x[j - 1]; --j) x[j] = x[j - 1], y[j] = y[j - 1];
			x[j] = tx, y[j] = ty;
		}
	}
}

// head/tail: 256 entries each; stack: room for n/64 + 2 jobs
__device__ void reference_sort_xy(uint64_t *x, uint64_t *y, uint32_t n, uint32_t *head, uint32_t *tail, SortJob *stack)
{
	if (n <= 64) { insertion_sort_xy(x, y, 0, n); return; }
	int sp = 0;
	stack[sp++] = SortJob{0, n, 56};
	while (sp) {
		const SortJob job = stack[--sp];
		const int sh = job.shift;
		for (int d = 0; d < 256; ++d) tail[d] = 0;
		for (uint32_t i = job.beg; i < job.end; ++i) ++tail[x[i] >> sh & 255];
		uint32_t run = job.beg;
		for (int d = 0; d < 256; ++d) { head[d] = run; run += tail[d]; tail[d] = run; }
		for (int d = 0; d < 256;) {
			if (head[d] == tail[d]) { ++d; continue; }
			int to = (int)(x[head[d]] >> sh & 255);
			if (to == d) { ++head[d]; continue; }
			uint64_t hx = x[head[d]], hy = y[head[d]];
			do {
				const uint32_t at = head[to]++;
				const uint64_t px = hx, py = hy;
				hx = x[at], hy = y[at];
				x[at] = px, y[at] = py;
				to = (int)(hx >> sh & 255);
			} while (to != d);
			const uint32_t at = head[d]++;
			x[at] = hx, y[at] = hy;
		}
		if (sh) {
			const int next = sh > 8 ? sh - 8 : 0;
			uint32_t lo = job.beg;
			for (int d = 0; d < 256; ++d) {
				const uint32_t hi = tail[d], sz = hi - lo;
				if (sz > 64) stack[sp++] = SortJob{lo, hi, next};
				else if (sz > 1) insertion_sort_xy(x, y, lo, hi);
				lo = hi;
			}
		}
	}
}

// Replay driver.  The recursion of the reference sort is unrolled into rounds: a round handles every open
// bucket ("job": range + digit position) with one wavefront each and appends the sub-buckets that still
// hold more than 64 elements to the next round's list.  Inside a job:
//   histogram                      all lanes (LDS atomics)
//   one non-empty digit            nothing moves
//   two non-empty digits (A < B)   closed form of the cycle-leader pass, all lanes:  bucket A keeps its
//                                  elements in place and the j-th misplaced slot receives the j-th A-element
//                                  found in bucket B's region; bucket B becomes  e_1, natives before f_1,
//                                  e_2, natives between f_1 and f_2, ...  (e_j = j-th B-element found in A's
//                                  region, f_j = slot of the j-th A-element in B's region)
//   otherwise                      the pass itself, lane 0 (each step depends on the element just displaced)
//   buckets of <= 64 elements      insertion sort, one lane per bucket
__global__ void __launch_bounds__(64) sort_init_kernel(const uint32_t *__restrict__ tie_reads, uint32_t n_tie,
                                                        const uint64_t *__restrict__ r_aoff, const uint64_t *__restrict__ ukey,
                                                        const uint64_t *__restrict__ uy, KeyLayout L, uint64_t *__restrict__ ax,
                                                        uint64_t *__restrict__ ay, SortJob *__restrict__ jobs, uint32_t *__restrict__ n_jobs)
{
	if (blockIdx.x >= n_tie) return;
	const uint32_t rl = tie_reads[blockIdx.x]; // batch-local read index
	const uint64_t a0 = r_aoff[rl], a1 = r_aoff[rl + 1];
	const uint32_t n = (uint32_t)(a1 - a0);
	for (uint32_t i = threadIdx.x; i < n; i += 64) {
		ax[a0 + i] = key_to_x(ukey[a0 + i], L);
		ay[a0 + i] = uy[a0 + i];
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		if (n <= 64) insertion_sort_xy(ax + a0, ay + a0, 0, n);
		else jobs[atomicAdd(n_jobs, 1u)] = SortJob{(uint32_t)a0, (uint32_t)a1, 56};
	}
}

__global__ void __launch_bounds__(64) sort_pass_kernel(const SortJob *__restrict__ jobs, uint32_t n_jobs, uint64_t *__restrict__ x,
                                                        uint64_t *__restrict__ y, uint64_t *__restrict__ tx, uint64_t *__restrict__ ty,
                                                        uint32_t *__restrict__ gs, SortJob *__restrict__ next, uint32_t *__restrict__ n_next)
{
	__shared__ uint32_t cnt[256], head[256], tail[256];
	__shared__ int nz, dA, dB;
	if (blockIdx.x >= n_jobs) return;
	__builtin_amdgcn_s_setprio(3);   // (see chain_kernel)
	const SortJob job = jobs[blockIdx.x];
	const int lane = threadIdx.x, sh = job.shift;
	const uint32_t beg = job.beg, end = job.end;
	for (int d = lane; d < 256; d += 64) cnt[d] = 0;
	__syncthreads();
	for (uint32_t i = beg + lane; i < end; i += 64) atomicAdd(&cnt[x[i] >> sh & 255], 1u);
	__syncthreads();
	if (lane == 0) {
		uint32_t run = beg;
		int k = 0, a = -1, b = -1;
		for (int d = 0; d < 256; ++d) {
			if (cnt[d]) { if (k == 0) a = d; else if (k == 1) b = d; ++k; }
			head[d] = run; run += cnt[d]; tail[d] = run;
		}
		nz = k, dA = a, dB = b;
	}
	__syncthreads();
	if (nz == 2) {
		const uint32_t mid = tail[dA], nB = end - mid, nA = mid - beg;
		// slots of the A-elements inside B's region, in order
		uint32_t m = 0;
		for (uint32_t base = 0; base < nB; base += 64) {
			const uint32_t pos = base + lane;
			const bool g = pos < nB && (int)(x[mid + pos] >> sh & 255) == dA;
			const unsigned long long bm = __ballot(g);
			if (g) gs[beg + m + __popcll(bm & ((1ULL << lane) - 1))] = pos;
			m += (uint32_t)__popcll(bm);
		}
		__threadfence_block();
		__builtin_amdgcn_wave_barrier();
		// A's region: natives stay; the j-th misplaced element e_j moves behind f_(j-1), its slot takes g_j
		uint32_t j = 0;
		for (uint32_t base = 0; base < nA; base += 64) {
			const uint32_t pos = base + lane;
			const bool in = pos < nA;
			const uint64_t vx = in ? x[beg + pos] : 0, vy = in ? y[beg + pos] : 0;
			const bool e = in && (int)(vx >> sh & 255) == dB;
			const unsigned long long bm = __ballot(e);
			if (e) {
				const uint32_t jj = j + (uint32_t)__popcll(bm & ((1ULL << lane) - 1)); // 0-based rank
				const uint32_t tgt = jj == 0 ? 0 : gs[beg + jj - 1] + 1;
				tx[mid + tgt] = vx, ty[mid + tgt] = vy;
				const uint32_t src = mid + gs[beg + jj];
				tx[beg + pos] = x[src], ty[beg + pos] = y[src];
			} else if (in) tx[beg + pos] = vx, ty[beg + pos] = vy;
			j += (uint32_t)__popcll(bm);
		}
		// B's region: natives in front of a remaining A-slot shift right by one
		uint32_t fcnt = 0;
		for (uint32_t base = 0; base < nB; base += 64) {
			const uint32_t pos = base + lane;
			const bool in = pos < nB;
			const uint64_t vx = in ? x[mid + pos] : 0, vy = in ? y[mid + pos] : 0;
			const bool g = in && (int)(vx >> sh & 255) == dA;
			const unsigned long long bm = __ballot(g);
			if (in && !g) {
				const uint32_t before = fcnt + (uint32_t)__popcll(bm & ((1ULL << lane) - 1));
				const uint32_t np = before < m ? pos + 1 : pos;
				tx[mid + np] = vx, ty[mid + np] = vy;
			}
			fcnt += (uint32_t)__popcll(bm);
		}
		__threadfence_block();
		__builtin_amdgcn_wave_barrier();
		for (uint32_t i = beg + lane; i < end; i += 64) x[i] = tx[i], y[i] = ty[i];
		__threadfence_block();
	} else if (nz > 2) {
		if (lane == 0) {
			for (int d = 0; d < 256;) {
				if (head[d] == tail[d]) { ++d; continue; }
				int to = (int)(x[head[d]] >> sh & 255);
				if (to == d) { ++head[d]; continue; }
				uint64_t hx = x[head[d]], hy = y[head[d]];
				do {
					const uint32_t at = head[to]++;
					const uint64_t px = hx, py = hy;
					hx = x[at], hy = y[at];
					x[at] = px, y[at] = py;
					to = (int)(hx >> sh & 255);
				} while (to != d);
				const uint32_t at = head[d]++;
				x[at] = hx, y[at] = hy;
			}
		}
		__threadfence_block();
	}
	__syncthreads();
	if (sh) {
		const int nsh = sh > 8 ? sh - 8 : 0;
		for (int d = lane; d < 256; d += 64) {
			const uint32_t hi = tail[d], lo = hi - cnt[d], sz = cnt[d];
			if (sz > 64) next[atomicAdd(n_next, 1u)] = SortJob{lo, hi, nsh};
			else if (sz > 1) insertion_sort_xy(x, y, lo, hi);
		}
	}
}

void launch_sort_init(const uint32_t *tie_reads, uint32_t n_tie, const uint64_t *r_aoff, const uint64_t *ukey, const uint64_t *uy,
                      const KeyLayout &L, uint64_t *ax, uint64_t *ay, void *jobs, uint32_t *n_jobs, hipStream_t s)
{
	if (n_tie) ND_LAUNCH(sort_init_kernel, dim3(n_tie), dim3(64), 0, s, tie_reads, n_tie, r_aoff, ukey, uy, L, ax, ay, (SortJob*)jobs,
	                              n_jobs);
}

void launch_sort_pass(const void *jobs, uint32_t n_jobs, uint64_t *x, uint64_t *y, uint64_t *tx, uint64_t *ty, uint32_t *gs, void *next,
                      uint32_t *n_next, hipStream_t s)
{
	if (n_jobs) ND_LAUNCH(sort_pass_kernel, dim3(n_jobs), dim3(64), 0, s, (const SortJob*)jobs, n_jobs, x, y, tx, ty, gs,
	                               (SortJob*)next, n_next);
}

size_t sort_job_bytes() { return sizeof(SortJob); }

// ------------------------------------------------------------------------------------------------
// K4: chaining DP.  One wavefront per query read walks its anchors in order (f[i] depends on every
// earlier f[j]); the predecessor window of anchor i is scored 64 candidates at a time, newest first,
// and the reference's sequential inner loop -- running maximum, skip counter with its early exit,
// "already on a better path" marks -- is reproduced with wavefront prefix operations:
//   running max        exclusive prefix max over lanes
//   skip counter       prefix composition of x -> max(x + a, b) maps (+1 / -1 floored at 0)
//   exit               first lane whose counter exceeds max_skip
// Marks (t[] in the reference) live in an LDS ring indexed by anchor number; scores/predecessors in HBM.

constexpr int kRing = 8192; // > max_chain_iter (5000)

// Wave-wide scans with data-parallel-primitive moves (row_shr 1 / 2 / 3 of the input, row_shr 4 and 8 of the partial results, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3): seven dependent steps of a few cycles each.  K4's inner loop ran
// them as 19 ds_bpermute shuffles until round 4 -- ~100 cycles apiece on a chain that is walked once per anchor: more than half of the
// ~1.5 us an anchor of the longest slab cost.  A lane whose source lies outside its row (or is masked off) receives the identity.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int dpp_move(int identity, int src)
{
	return __builtin_amdgcn_update_dpp(identity, src, CTRL, ROW_MASK, BANK_MASK, false);
}

__device__ __forceinline__ int wave_shr1_i32(int first, int v) // lane l gets v of lane l - 1, lane 0 gets `first`
{
	return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ uint64_t wave_shr1_u64(uint64_t v) // (lane 0's result is the caller's to set)
{
	const uint32_t lo = (uint32_t)wave_shr1_i32(0, (int)(uint32_t)v), hi = (uint32_t)wave_shr1_i32(0, (int)(uint32_t)(v >> 32));
	return (uint64_t)hi << 32 | lo;
}

__device__ __forceinline__ int wave_excl_max(int v, int lane)
{
	// inclusive max-scan, then shift by one lane
	const int id = INT32_MIN;
	int r = v, t;
	t = dpp_move<0x111, 0xf, 0xf>(id, v); r = r > t ? r : t;
	t = dpp_move<0x112, 0xf, 0xf>(id, v); r = r > t ? r : t;
	t = dpp_move<0x113, 0xf, 0xf>(id, v); r = r > t ? r : t;
	t = dpp_move<0x114, 0xf, 0xe>(id, r); r = r > t ? r : t;
	t = dpp_move<0x118, 0xf, 0xc>(id, r); r = r > t ? r : t;
	t = dpp_move<0x142, 0xa, 0xf>(id, r); r = r > t ? r : t;
	t = dpp_move<0x143, 0xc, 0xf>(id, r); r = r > t ? r : t;
	(void)lane;
	return wave_shr1_i32(INT32_MIN, r);
}

// inclusive scan of the maps x -> max(x + a, b) under composition (lane l's result = the maps of lanes 0..l applied in lane order)
__device__ __forceinline__ void wave_incl_compose(int &a, int &b)
{
	const int ida = 0, idb = INT32_MIN / 2;
	const int a0 = a, b0 = b;
	auto after = [&](int oa, int ob) { // (a, b) after (oa, ob)
		const int nb = ob + a > b ? ob + a : b;
		a = oa + a, b = nb;
	};
	after(dpp_move<0x111, 0xf, 0xf>(ida, a0), dpp_move<0x111, 0xf, 0xf>(idb, b0));
	after(dpp_move<0x112, 0xf, 0xf>(ida, a0), dpp_move<0x112, 0xf, 0xf>(idb, b0));
	after(dpp_move<0x113, 0xf, 0xf>(ida, a0), dpp_move<0x113, 0xf, 0xf>(idb, b0));
	{ const int oa = dpp_move<0x114, 0xf, 0xe>(ida, a), ob = dpp_move<0x114, 0xf, 0xe>(idb, b); after(oa, ob); }
	{ const int oa = dpp_move<0x118, 0xf, 0xc>(ida, a), ob = dpp_move<0x118, 0xf, 0xc>(idb, b); after(oa, ob); }
	{ const int oa = dpp_move<0x142, 0xa, 0xf>(ida, a), ob = dpp_move<0x142, 0xa, 0xf>(idb, b); after(oa, ob); }
	{ const int oa = dpp_move<0x143, 0xc, 0xf>(ida, a), ob = dpp_move<0x143, 0xc, 0xf>(idb, b); after(oa, ob); }
}

// The 64 most recent anchors (x, query position, f, p, v) ride in registers, lane l = anchor i-1-l, and are
// shifted by one lane per iteration: the usual predecessor window never touches memory.  Older chunks of
// a long window come from HBM (F/P/V are written there by lane 0 every iteration).
__global__ void __launch_bounds__(64) chain_kernel(const uint64_t *__restrict__ slab_i0, const uint32_t *__restrict__ slab_read,
                                                    uint32_t n_slabs, uint64_t n_anchors, const uint64_t *__restrict__ r_aoff,
                                                    const float *__restrict__ read_avg_span, const uint64_t *__restrict__ ax,
                                                    const uint64_t *__restrict__ ay, OvlParams P, int32_t *__restrict__ f,
                                                    int32_t *__restrict__ p, int32_t *__restrict__ v,
                                                    unsigned long long *__restrict__ cells)
{
	__shared__ uint16_t ring[kRing];
	const uint32_t sb = blockIdx.x;
	if (sb >= n_slabs) return;
	// (the stage's chain kernels ask the SIMD's arbiter for priority: when the stage runs beside the consensus of another seed file
	// -- stage.StagePipeline -- a wavefront of theirs shares its SIMD with seven of K7's, and a chain that gets an eighth of the issue
	// slots is eight times as long; they are few wavefronts, what they take nobody misses)
	__builtin_amdgcn_s_setprio(3);
	const int lane = threadIdx.x;
	// one slab = whole segments [i0, n) of one read (indices relative to the read: p[] holds read-relative indices)
	const uint32_t rd = slab_read[sb];
	const uint64_t a0 = r_aoff[rd];
	const int32_t i0 = (int32_t)(slab_i0[sb] - a0);
	const int32_t n = (int32_t)((sb + 1 < n_slabs && slab_read[sb + 1] == rd ? slab_i0[sb + 1] : r_aoff[rd + 1]) - a0);
	(void)n_anchors;
	const uint64_t *X = ax + a0, *Y = ay + a0;
	int32_t *F = f + a0, *Pp = p + a0, *V = v + a0;
	const float avg_span = read_avg_span[rd];
	const double lin = .01;
	const uint64_t max_dist = (uint64_t)P.max_gap;

	uint64_t wx = 0;                       // register window
	int32_t wq = 0, wf = 0, wp = -1, wv = 0;
	uint64_t cx = i0 + lane < n ? X[i0 + lane] : 0, cy = i0 + lane < n ? Y[i0 + lane] : 0;  // anchors of the current block of 64
	uint64_t nx = i0 + 64 + lane < n ? X[i0 + 64 + lane] : 0, ny = i0 + 64 + lane < n ? Y[i0 + 64 + lane] : 0; // next block
	unsigned long long my_cells = 0;

	for (int32_t i = i0; i < n; ++i) {
		const int bl = (i - i0) & 63;
		if (bl == 0 && i != i0) {
			cx = nx, cy = ny;
			const int32_t q = i + 64 + lane;
			nx = q < n ? X[q] : 0, ny = q < n ? Y[q] : 0;
		}
		const uint64_t ri = __shfl(cx, bl, 64), yi = __shfl(cy, bl, 64);
		const int32_t qi = (int32_t)yi, span = (int32_t)(yi >> 32 & 0xff);
		if (ri == ~0ull) {  // an anchor the thinning dropped (chain.c:231-234): f = p = v = -1, and nobody chains through it
			if (lane == 0) F[i] = -1, Pp[i] = -1, V[i] = -1, ring[i & (kRing - 1)] = (uint16_t)i;
			wx = wave_shr1_u64(wx), wq = wave_shr1_i32(0, wq), wf = wave_shr1_i32(0, wf), wp = wave_shr1_i32(0, wp);
			wv = wave_shr1_i32(0, wv);
			if (lane == 0) wx = ri, wq = qi, wf = -1, wp = -1, wv = -1;
			__builtin_amdgcn_wave_barrier();
			continue;
		}
		int32_t best = span, skipped = 0, best_j = -1;
		bool stop = false, fenced = false;
		for (int32_t base = i - 1; base >= i0 && !stop; base -= 64) {
			const int32_t j = base - lane;
			uint64_t xj;
			int32_t qj, fj, pj;
			if (base == i - 1) xj = wx, qj = wq, fj = wf, pj = wp;
			else {
				if (!fenced) { __threadfence_block(); fenced = true; }
				xj = 0, qj = 0, fj = 0, pj = -1;
				if (j >= i0) xj = X[j], qj = (int32_t)Y[j], fj = F[j], pj = Pp[j];
			}
			// (a dropped anchor -- x all ones -- lies inside the window as far as the index range goes and is stepped over:
			// chain.c:241,244; the window's far end is the first anchor that is neither dropped nor in reach)
			const bool dropped = xj == ~0ull;
			const bool in_win = j >= i0 && i - j <= P.max_iter && (dropped || ri <= xj + max_dist);
			bool act = false;
			int32_t sc = INT32_MIN;
			if (in_win && !dropped) {
				const int64_t dr = (int64_t)(ri - xj);
				const int32_t dq = qi - qj;
				if (!(dr == 0 || dq <= 0 || dq > P.max_gap)) {
					const int32_t dd = (int32_t)(dr > dq ? dr - dq : dq - dr);
					if (dd <= P.bw) {
						int32_t s0 = dq < dr ? dq : (int32_t)dr;
						if (s0 > span) s0 = span;
						const int32_t lg = dd ? 31 - __clz(dd) : 0;
						s0 -= (int)(dd * lin * avg_span) + (lg >> 1);
						sc = s0 + fj;
						act = true;
					}
				}
			}
			my_cells += in_win;
			const unsigned long long win_mask = __ballot(in_win);
			if (win_mask == 0) break;
			// marks made by this chunk must be visible to its own later lanes: write all, then read.  A slot is
			// shared by anchors kRing apart, so marks for anchors more than kRing behind i are dropped (never read).
			if (act && pj >= 0 && pj > i - kRing) ring[pj & (kRing - 1)] = (uint16_t)i;
			__builtin_amdgcn_wave_barrier();
			const bool marked = act && ring[j & (kRing - 1)] == (uint16_t)i;
			int run = wave_excl_max(act ? sc : INT32_MIN, lane);
			if (run < best) run = best;
			const bool newmax = act && sc > run;
			// skip counter: compose x -> max(x + a, b)
			int a = 0, b = INT32_MIN / 2;
			if (newmax) a = -1, b = 0;
			else if (marked) a = 1;
			wave_incl_compose(a, b);
			const int after = skipped + a > b ? skipped + a : b;
			const bool exits = marked && !newmax && after > P.max_skip;
			const unsigned long long exit_mask = __ballot(exits);
			unsigned long long live = ~0ULL;
			if (exit_mask) {
				const int el = __ffsll((long long)exit_mask) - 1;
				live = el ? (~0ULL >> (64 - el)) : 0ULL; // lanes strictly before the exit lane
				stop = true;
			}
			const unsigned long long nm = __ballot(newmax) & live;
			if (nm) {
				const int last = 63 - __clzll((long long)nm);
				best = __shfl(sc, last, 64);
				best_j = base - last;
			}
			if (win_mask != ~0ULL) break; // the window ended inside this chunk
			if (!stop) skipped = __shfl(after, 63, 64);
		}
		// peak score on the path ending here
		int32_t peak = best;
		if (best_j >= 0) {
			int32_t vb;
			const int back = i - 1 - best_j;
			if (back < 64) vb = __shfl(wv, back, 64);
			else { if (!fenced) { __threadfence_block(); fenced = true; } vb = V[best_j]; }
			if (vb > best) peak = vb;
		}
		if (lane == 0) F[i] = best, Pp[i] = best_j, V[i] = peak, ring[i & (kRing - 1)] = (uint16_t)i;
		// shift the register window by one anchor
		wx = wave_shr1_u64(wx), wq = wave_shr1_i32(0, wq), wf = wave_shr1_i32(0, wf), wp = wave_shr1_i32(0, wp);
		wv = wave_shr1_i32(0, wv);
		if (lane == 0) wx = ri, wq = qi, wf = best, wp = best_j, wv = peak;
		__builtin_amdgcn_wave_barrier();
	}
	for (int d = 32; d; d >>= 1) my_cells += __shfl_xor(my_cells, d, 64);
	if (lane == 0 && cells) atomicAdd(cells, my_cells);

}

// chain ends (chain.c:87-104), one wavefront per read after all its slabs: anchors nobody points to whose peak
// reaches min_sc, with the peak walk; t[] arrives zeroed and leaves zeroed for the backtrack of K5
__global__ void __launch_bounds__(64) chain_ends_kernel(const uint64_t *__restrict__ r_aoff, uint32_t n_reads, OvlParams P,
                                                         const int32_t *__restrict__ f, const int32_t *__restrict__ p,
                                                         const int32_t *__restrict__ v, int32_t *__restrict__ t, uint64_t *__restrict__ u,
                                                         uint32_t *__restrict__ n_end)
{
	const uint32_t rd = blockIdx.x;
	if (rd >= n_reads) return;
	__builtin_amdgcn_s_setprio(3);   // (see chain_kernel)
	const int lane = threadIdx.x;
	const uint64_t a0 = r_aoff[rd];
	const int32_t n = (int32_t)(r_aoff[rd + 1] - a0);
	if (n == 0) { if (lane == 0) n_end[rd] = 0; return; }
	const int32_t *F = f + a0, *Pp = p + a0, *V = v + a0;
	int32_t *T = t + a0;
	uint64_t *U = u + a0;
	for (int32_t i = lane; i < n; i += 64) { const int32_t pi = Pp[i]; if (pi >= 0) T[pi] = 1; }
	__threadfence_block();
	__builtin_amdgcn_wave_barrier();
	uint32_t n_u = 0;
	for (int32_t base = 0; base < n; base += 64) {
		const int32_t i = base + lane;
		bool e = false;
		uint64_t uv = 0;
		if (i < n && T[i] == 0 && V[i] >= P.min_sc) {
			int32_t j = i;
			while (j >= 0 && F[j] < V[j]) j = Pp[j];
			if (j < 0) j = i;
			uv = (uint64_t)(uint32_t)F[j] << 32 | (uint32_t)j;
			e = true;
		}
		const unsigned long long m = __ballot(e);
		if (e) U[n_u + __popcll(m & ((1ULL << lane) - 1))] = uv;
		n_u += (uint32_t)__popcll(m);
	}
	__builtin_amdgcn_wave_barrier();
	for (int32_t i = lane; i < n; i += 64) T[i] = 0;
	if (lane == 0) n_end[rd] = n_u;
}

// mm_chain_dp_nextdenovo's anchor thinning (minimap2/chain.c:185-226), for the mappings of --step 2 --mode 1 that chain through it: a
// read with more than 100,000 anchors loses the anchors of crowded target positions before the DP.  Groups = runs of anchors with the
// same 32-bit target position (strand and read number are not looked at); t[] counts them from slot 1, v[g - 1] holds group g's
// position, v[last] the last position + 20; when the largest group has more than 200 anchors, an anchor of a group larger than 0.8 x
// the largest is dropped if it lies within 10 of the last position kept and the next group starts within 10 of it.  A dropped anchor's
// x becomes all ones (the reference's a[i].x = -1).  Two sequential passes per read, one lane each: rare reads, ~10 ms.
// (The reference's counters can step one slot past its arrays when every anchor is a group of its own; that slot is not written here.)
__global__ void thin_anchors_kernel(const uint64_t *__restrict__ r_aoff, uint32_t n_reads, uint64_t *__restrict__ ax, int32_t *__restrict__ t,
                                    int32_t *__restrict__ v, const int32_t *__restrict__ read_mid, uint32_t read_base, int32_t mid)
{
	const uint32_t rd = blockIdx.x * blockDim.x + threadIdx.x;
	if (rd >= n_reads) return;
	// a read that is chained a second time (-f FLOAT,INT: its threshold was raised) is chained by mm_chain_dp there, without the
	// thinning: the re-chaining branch of mm_map_frag_nextdenovo1 does not call the _nextdenovo form (minimap2/map.c:696-698)
	if (read_mid && read_mid[read_base + rd] != mid) return;
	const uint64_t a0 = r_aoff[rd];
	const int64_t n = (int64_t)(r_aoff[rd + 1] - a0);
	if (n <= 100000) return;
	uint64_t *X = ax + a0;
	int32_t *T = t + a0, *V = v + a0;   // T arrives zeroed and leaves zeroed; V is K4's to overwrite
	const int32_t maxc = 200, maxw = 10;
	int64_t i, j = 0;
	int32_t px = 0, k = 0, pm, pi;
	for (i = 0; i < n; ++i) {
		pi = (int32_t)X[i];
		if (pi != px) {
			if (j < n && T[j] > k) k = T[j];
			j++;
			V[j - 1] = px = pi;
		}
		if (j < n) T[j]++;
	}
	if (j < n && T[j] > k) k = T[j];
	if (j < n) V[j] = (int32_t)X[n - 1] + maxw * 2;
	const int64_t groups = j;
	if (k > maxc) {
		k = (int32_t)((double)k * (double)0.8f + .499);
		for (i = j = 0, px = 0, pm = (int32_t)X[0]; i < n; ++i) {
			pi = (int32_t)X[i];
			if (pi != px) px = pi, j++;
			if (j < n && T[j] > k && pi > pm && pi < pm + maxw && V[j] < pi + maxw) X[i] = ~0ull;
			else pm = pi;
		}
	}
	for (i = 0; i <= groups && i < n; ++i) T[i] = 0;
}

void launch_thin_anchors(const uint64_t *r_aoff, uint32_t n_reads, uint64_t *ax, int32_t *t, int32_t *v, const int32_t *read_mid, uint32_t read_base,
                         int32_t mid, hipStream_t s)
{
	if (n_reads) ND_LAUNCH(thin_anchors_kernel, dim3((n_reads + 63) / 64), dim3(64), 0, s, r_aoff, n_reads, ax, t, v, read_mid, read_base, mid);
}

void launch_chain(const uint64_t *slab_i0, const uint32_t *slab_read, uint32_t n_slabs, uint64_t n_anchors, const uint64_t *r_aoff,
                  const float *read_avg_span, const uint64_t *ax, const uint64_t *ay, const OvlParams &P, int32_t *f, int32_t *p, int32_t *v,
                  unsigned long long *cells, hipStream_t s)
{
	if (n_slabs) ND_LAUNCH(chain_kernel, dim3(n_slabs), dim3(64), 0, s, slab_i0, slab_read, n_slabs, n_anchors, r_aoff, read_avg_span,
	                                ax, ay, P, f, p, v, cells);
}

void launch_chain_ends(const uint64_t *r_aoff, uint32_t n_reads, const OvlParams &P, const int32_t *f, const int32_t *p, const int32_t *v,
                       int32_t *t, uint64_t *u, uint32_t *n_end, hipStream_t s)
{
	if (n_reads) ND_LAUNCH(chain_ends_kernel, dim3(n_reads), dim3(64), 0, s, r_aoff, n_reads, P, f, p, v, t, u, n_end);
}

// ------------------------------------------------------------------------------------------------
// K5: chains -> hits.  Per read the work is a few hundred pointer-chasing steps (peak search,
// backtrack) and two tiny sorts: one lane per read, 64 reads per wavefront, all reads in flight.

__device__ void heap_sort_desc(uint64_t *a, int32_t n)
{
	// min-heap based in-place sort -> descending order
	for (int32_t start = n / 2 - 1; start >= 0; --start) {
		int32_t root = start;
		for (;;) {
			int32_t c = 2 * root + 1;
			if (c >= n) break;
			if (c + 1 < n && a[c + 1] < a[c]) ++c;
			if (a[root] <= a[c]) break;
			uint64_t t = a[root]; a[root] = a[c], a[c] = t;
			root = c;
		}
	}
	for (int32_t end = n - 1; end > 0; --end) {
		uint64_t t = a[0]; a[0] = a[end], a[end] = t;
		int32_t root = 0;
		for (;;) {
			int32_t c = 2 * root + 1;
			if (c >= end) break;
			if (c + 1 < end && a[c + 1] < a[c]) ++c;
			if (a[root] <= a[c]) break;
			uint64_t t2 = a[root]; a[root] = a[c], a[c] = t2;
			root = c;
		}
	}
}

__device__ __forceinline__ int dovetail_class(int rev, uint32_t qs, uint32_t qe, uint32_t qlen, uint32_t ts, uint32_t te, uint32_t tlen,
                                              int32_t h1, int32_t h2)
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

__global__ void __launch_bounds__(64) hits_kernel(const uint64_t *__restrict__ r_aoff, uint32_t n_reads, uint32_t read_base, const uint64_t *__restrict__ ax,
                            const uint64_t *__restrict__ ay, IndexDev ix, QueryDev q, OvlParams P, const int32_t *__restrict__ f,
                            const int32_t *__restrict__ p, int32_t *__restrict__ v, int32_t *__restrict__ t, uint64_t *__restrict__ u,
                            uint64_t *__restrict__ bx, uint64_t *__restrict__ by, uint64_t *__restrict__ wx, uint64_t *__restrict__ wy,
                            uint32_t *__restrict__ tables, SortJob *__restrict__ stacks, const uint32_t *__restrict__ n_end,
                            OvlRec *__restrict__ recs, uint32_t *__restrict__ n_rec, uint32_t *__restrict__ n_chain,
                            OvlRec10 *__restrict__ recs10, uint64_t *__restrict__ cx, uint64_t *__restrict__ cy, uint32_t *__restrict__ n_ca,
                            unsigned long long *__restrict__ prof)
{
	__builtin_amdgcn_s_setprio(3);   // (see chain_kernel)
	// (NDGPU_K5_PROF: per phase the sum over reads and the longest single read, in 10 ns ticks of the constant clock)
#ifdef SIMT_EMULATION
#define K5_TICK(ph) ((void)0)
#else
	unsigned long long tk = prof ? wall_clock64() : 0ull;
#define K5_TICK(ph) do { if (prof && lane == 0) { const unsigned long long now = wall_clock64(); atomicAdd(&prof[2 * (ph)], now - tk); atomicMax(&prof[2 * (ph) + 1], now - tk); tk = now; } } while (0)
#endif
	// One WAVEFRONT per read.  Until round 4 a lane took a read and walked it alone: ~1.4 us per anchor of dependent global loads,
	// 5 ms for an average read and 40 ms for the heaviest one -- the length of the launch (the per-phase clock of NDGPU_K5_PROF:
	// backtrack 19 ms, chain copy 9 ms, records 5 ms on that read).  The walks of a read's chains are independent but for one
	// rule -- a chain stops at the first anchor a better chain has taken -- and that rule is order-free once stated as "an anchor
	// belongs to the best chain end of its subtree": every lane walks one chain, best ranks first, and claims with atomicMax.
	const int lane = (int)threadIdx.x;
	const uint32_t rl = blockIdx.x;
	if (rl >= n_reads) return;
	const uint32_t rd = read_base + rl;
	const uint64_t a0 = r_aoff[rl];
	const int32_t n = (int32_t)(r_aoff[rl + 1] - a0);
	if (lane == 0) {
		n_rec[rl] = 0, n_chain[rl] = 0;
		if (P.chains) n_ca[rl] = 0;
	}
	if (n == 0) return;
	const uint64_t *X = ax + a0, *Y = ay + a0;
	const int32_t *F = f + a0, *Pp = p + a0;
	int32_t *T = t + a0;  // (v[] is K4's; the walks are not listed any more)
	// (wx / wy: one slot per anchor, or two when chains of ONE anchor pass -- min_cnt < 2 -- and a read can have as many chains as anchors)
	const uint64_t w0 = P.min_cnt < 2 ? 2 * a0 : a0;
	uint64_t *U = u + a0, *BX = bx + a0, *BY = by + a0, *WX = wx + w0, *WY = wy + w0;
	uint32_t *head = tables + (size_t)rl * 512, *tail = head + 256;
	SortJob *stack = stacks + a0 / 64 + 2 * (size_t)rl;

	// chain ends were collected by K4 (u[], n_end[]); t[] is zero
	int32_t n_u = (int32_t)n_end[rl];
	if (n_u == 0) return;
	auto sync_wave = [] {  // what lanes wrote to global memory before is visible to the lanes after
		__threadfence_block();
		__builtin_amdgcn_wave_barrier();
	};
	auto excl_sum = [&](int32_t val, int32_t &total) {  // exclusive prefix sum over the 64 lanes
		int32_t inc = val;
		for (int dd = 1; dd < 64; dd <<= 1) {
			const int32_t o = __shfl_up(inc, dd, 64);
			if (lane >= dd) inc += o;
		}
		total = __shfl(inc, 63, 64);
		return inc - val;
	};
	if (lane == 0) heap_sort_desc(U, n_u);
	sync_wave();
	K5_TICK(0);

	// backtrack, best chain first (chain.c:106-125).  Sequentially: chain i takes its end anchor, then follows p[] while the
	// anchors are free.  An anchor is taken by the best-ranked chain end among those whose walk leads through it, and nothing below
	// it on that chain's walk can belong to a better one (it would lead through the anchor too): so the claims are the maximum of
	// (n_u - i) over the chains that walk through, whatever the order of the walks -- a lane stops where it meets a better claim,
	// and a better chain that arrives later walks on over the worse claims above.
	for (int32_t base = 0; base < n_u; base += 64) {
		const int32_t i = base + lane;
		if (i < n_u) {
			const int32_t mine = n_u - i;
			int32_t j = (int32_t)U[i];
			do {
				if (atomicMax(&T[j], mine) > mine) break;
				j = Pp[j];
			} while (j >= 0);
		}
	}
	sync_wave();
	// every chain's length and verdict; the chains that pass, in rank order: U[k] = score << 32 | count, WX[k] = end anchor << 32 | k0
	int32_t k = 0, n_v = 0;
	for (int32_t base = 0; base < n_u; base += 64) {
		const int32_t i = base + lane;
		bool keep = false;
		int32_t cnt = 0, end = 0;
		uint64_t nu = 0;
		if (i < n_u) {
			const int32_t mine = n_u - i;
			const uint64_t ui = U[i];
			end = (int32_t)ui;
			cnt = 1;  // (the end anchor is taken whoever holds it: do { ... } while)
			int32_t j = Pp[end];
			while (j >= 0 && T[j] == mine) cnt++, j = Pp[j];
			if (j < 0) keep = cnt >= P.min_cnt, nu = ui >> 32 << 32 | (uint64_t)(uint32_t)cnt;
			else if ((int32_t)(ui >> 32) - F[j] >= P.min_sc) keep = cnt >= P.min_cnt, nu = ((ui >> 32) - (uint64_t)F[j]) << 32 | (uint64_t)(uint32_t)cnt;
		}
		ND_LOCKSTEP();  // (every lane has read its U[i]: slots up to base + lane are overwritten now)
		const unsigned long long km = __ballot(keep);
		int32_t tot = 0;
		const int32_t off = excl_sum(keep ? cnt : 0, tot);
		if (keep) {
			const int32_t kk = k + __popcll(km & ((1ULL << lane) - 1));
			U[kk] = nu;
			WX[kk] = (uint64_t)(uint32_t)end << 32 | (uint32_t)(n_v + off);
		}
		k += __popcll(km);
		n_v += tot;
	}
	n_u = k;
	sync_wave();
	K5_TICK(1);
	if (n_u == 0) return;
	// chained anchors, each chain in increasing order; then the chains' first anchors for the ordering below
	for (int32_t base = 0; base < n_u; base += 64) {
		const int32_t i = base + lane;
		if (i < n_u) {
			const int32_t cnt = (int32_t)U[i], k0 = (int32_t)(uint32_t)WX[i];
			int32_t j = (int32_t)(WX[i] >> 32);
			for (int32_t m = cnt - 1; m >= 0; --m) { BX[k0 + m] = X[j], BY[k0 + m] = Y[j]; j = Pp[j]; }
			WY[i] = (uint64_t)(uint32_t)k0 << 32 | (uint32_t)i;
			WX[i] = BX[k0];
		}
	}
	sync_wave();
	K5_TICK(2);
	// order chains by the x of their first anchor (the reference sort, ties included: a sequential replay, lane 0)
	if (lane == 0) reference_sort_xy(WX, WY, (uint32_t)n_u, head, tail, stack);
	sync_wave();
	K5_TICK(3);
	if (P.chains) {
		// -c: the base-level alignment walks a[] itself (the chain before / after a hit's own: minimap2/align.c:629-664), so the
		// chains are copied out in that order and addressed there from here on
		uint64_t *CX = cx + a0, *CY = cy + a0;
		int32_t kk = 0;
		for (int32_t base = 0; base < n_u; base += 64) {
			const int32_t i = base + lane;
			int32_t src = 0, first = 0, cnt = 0;
			if (i < n_u) src = (int32_t)WY[i], first = (int32_t)(WY[i] >> 32), cnt = (int32_t)U[src];
			int32_t tot = 0;
			const int32_t at = kk + excl_sum(cnt, tot);
			if (i < n_u) {
				for (int32_t j = 0; j < cnt; ++j) CX[at + j] = BX[first + j], CY[at + j] = BY[first + j];
				WY[i] = (uint64_t)(uint32_t)at << 32 | (uint32_t)src;
			}
			kk += tot;
		}
		BX = CX, BY = CY;
		if (lane == 0) n_ca[rl] = (uint32_t)kk;
		sync_wave();
	}
	// hash-ordered hits: z.x = score<<32 | (cnt ^ h), z.y = first<<32 | cnt, over the re-ordered chains
	const uint32_t qhash = q.hash[rd];
	// (a chain has min_cnt >= 2 anchors, so n_u <= n / 2 and the upper halves of WX/WY are free; with min_cnt < 2 the arrays are twice as long)
	uint64_t *ZX = WX + n_u, *ZY = WY + n_u;
	// The reference now copies the chains back in this order; the anchors are the same, so they are addressed
	// in place through `first` (their offset in B).
	for (int32_t i = lane; i < n_u; i += 64) {
		const int32_t src = (int32_t)WY[i];
		const int32_t first = (int32_t)(WY[i] >> 32), cnt = (int32_t)U[src];
		const uint32_t h = (uint32_t)hash_full((hash_full(BX[first]) + hash_full(BY[first])) ^ (uint64_t)qhash);
		ZX[i] = U[src] ^ (uint64_t)h;
		ZY[i] = (uint64_t)(uint32_t)first << 32 | (uint32_t)cnt;
	}
	sync_wave();
	K5_TICK(4);
	if (lane == 0) reference_sort_xy(ZX, ZY, (uint32_t)n_u, head, tail, stack);
	sync_wave();
	K5_TICK(5);
	// reversed: larger first
	const uint32_t qid = q.id[rd], qlen = q.len[rd];
	OvlRec *out = recs + a0 / (uint64_t)(P.min_cnt > 1 ? P.min_cnt : 1) + rl;
	uint32_t n_out = 0;
	// one lane per hit, in rounds of 64 from the last sorted hit down; the hits that pass keep their order
	auto hit = [&](int32_t i, OvlRec &r) -> bool {
		const int32_t first = (int32_t)(ZY[i] >> 32), cnt = (int32_t)ZY[i], last = first + cnt - 1;
		const int32_t span0 = (int32_t)(BY[first] >> 32 & 0xff);
		const uint32_t rev = (uint32_t)(BX[first] >> 63), rid = (uint32_t)(BX[first] << 1 >> 33);
		const int32_t rs = (int32_t)BX[first] + 1 > span0 ? (int32_t)BX[first] + 1 - span0 : 0;
		const int32_t re = (int32_t)BX[last] + 1;
		int32_t qs, qe;
		if (!rev) qs = (int32_t)BY[first] + 1 - span0, qe = (int32_t)BY[last] + 1;
		else qs = (int32_t)qlen - ((int32_t)BY[last] + 1), qe = (int32_t)qlen - ((int32_t)BY[first] + 1 - span0);
		if (P.chains) {
			OvlRec c;
			c.rev = rev, c.qname = rid, c.qs = (uint32_t)first, c.qe = (uint32_t)cnt, c.tname = (uint32_t)(ZX[i] >> 32), c.ts = (uint32_t)ZX[i];
			c.te = 0, c.match = 0;
			r = c;
			return true;
		}
		const uint32_t tid = P.nameless ? 0u : ix.id[rid]; // (re-alignment: `rid` numbers the wanted list, nobody has a name)
		if (!P.nameless && tid == qid) return false;
		int32_t mlen = span0, blen = span0; // mm_cal_fuzzy_len (minimap2/hit.c): matching bases, block length
		for (int32_t m = first + 1; m <= last; ++m) {
			const int sp = (int)(BY[m] >> 32 & 0xff);
			const int tl = (int32_t)BX[m] - (int32_t)BX[m - 1];
			const int ql = (int32_t)BY[m] - (int32_t)BY[m - 1];
			blen += tl > ql ? tl : ql;
			mlen += tl > sp && ql > sp ? sp : tl < ql ? tl : ql;
		}
		if (P.step2 || P.provisional) { // provisional: local target index and block length in the name fields, judged below / on the host
			r.rev = rev, r.qname = rid, r.qs = (uint32_t)qs, r.qe = (uint32_t)qe, r.tname = (uint32_t)blen, r.ts = (uint32_t)rs, r.te = (uint32_t)re,
			r.match = (uint32_t)mlen;
			return true;
		}
		if (P.mode3) {
			// nd_fix_bad_ends + nd_update_coors (minimap2/map.c:313-373): anchors at either end of the chain that sit off its
			// diagonal by more than half the length walked so far are dropped; the match length keeps the whole chain's value.
			// The record is provisional: local read indices instead of names, no length / dovetail filter yet -- both follow
			// the end extension (ext_apply_kernel).
			int32_t as = first, ce = first + cnt; // anchors [as, ce) survive
			if (cnt >= 3) {
				const int32_t bw = P.bw, min_match = P.min_sc * 2;
				int32_t l = span0, mm = span0;
				for (int32_t i = first + 1; i < first + cnt - 1; ++i) {
					if (BY[i] & kSeedLongJoin) break;
					const int32_t sp = (int32_t)(BY[i] >> 32 & 0xff);
					const int32_t lr = (int32_t)BX[i] - (int32_t)BX[i - 1], lq = (int32_t)BY[i] - (int32_t)BY[i - 1];
					const int32_t lo = lr < lq ? lr : lq, hi = lr > lq ? lr : lq;
					if (hi - lo > l >> 1) as = i;
					l += lo;
					mm += lo < sp ? lo : sp;
					if (l >= bw << 1 || (mm >= min_match && mm >= bw) || mm >= mlen >> 1) break;
				}
				l = mm = (int32_t)(BY[last] >> 32 & 0xff);
				for (int32_t i = last - 1; i > as; --i) {
					if (BY[i + 1] & kSeedLongJoin) break;
					const int32_t sp = (int32_t)(BY[i + 1] >> 32 & 0xff);
					const int32_t lr = (int32_t)BX[i + 1] - (int32_t)BX[i], lq = (int32_t)BY[i + 1] - (int32_t)BY[i];
					const int32_t lo = lr < lq ? lr : lq, hi = lr > lq ? lr : lq;
					if (hi - lo > l >> 1) ce = i + 1;
					l += lo;
					mm += lo < sp ? lo : sp;
					if (l >= bw << 1 || (mm >= min_match && mm >= bw) || mm >= mlen >> 1) break;
				}
			}
			int32_t ts = rs, te = re, q0 = qs, q1 = qe;
			if (as != first || ce != first + cnt) {
				const int32_t sp = (int32_t)(BY[as] >> 32 & 0xff), lz = ce - 1;
				ts = (int32_t)BX[as] + 1 > sp ? (int32_t)BX[as] + 1 - sp : 0;
				te = (int32_t)BX[lz] + 1;
				if (!rev) q0 = (int32_t)BY[as] + 1 - sp, q1 = (int32_t)BY[lz] + 1;
				else q0 = (int32_t)qlen - ((int32_t)BY[lz] + 1), q1 = (int32_t)qlen - ((int32_t)BY[as] + 1 - sp);
			}
			r.rev = rev, r.qname = rd, r.qs = (uint32_t)q0, r.qe = (uint32_t)q1, r.tname = rid, r.ts = (uint32_t)ts, r.te = (uint32_t)te,
			r.match = (uint32_t)mlen;
			return true;
		}
		if (qe - qs < P.minlen) return false;
		if (P.dvt && !dovetail_class((int)rev, (uint32_t)qs, (uint32_t)qe, qlen, (uint32_t)rs, (uint32_t)re, ix.len[rid], P.maxhan1,
		                             P.maxhan2)) return false;
		r.rev = rev, r.qname = qid, r.qs = (uint32_t)qs, r.qe = (uint32_t)qe, r.tname = tid, r.ts = (uint32_t)rs, r.te = (uint32_t)re,
		r.match = (uint32_t)mlen;
		return true;
	};
	for (int32_t base = 0; base < n_u; base += 64) {
		const int32_t i = n_u - 1 - (base + lane);
		OvlRec r;
		const bool keep = i >= 0 && hit(i, r);
		const unsigned long long km = __ballot(keep);
		if (keep) out[n_out + (uint32_t)__popcll(km & ((1ULL << lane) - 1))] = r;
		n_out += (uint32_t)__popcll(km);
	}
	sync_wave();
	K5_TICK(6);
	if (P.step2 && !P.provisional && lane == 0) {
		// worker_for with the re-alignment switched off (--mode 0, minimap2/map.c:988-1031): the first hit of a target carries
		// the verdict of that target -- the match count of a dovetail hit, 3 = the query looks contained -- later hits of the
		// same target are marked 1 and count only when they are nearly as long; two contained verdicts end the marking.
		int32_t c = 0;
		for (uint32_t k2 = 0; k2 < n_out; ++k2) {
			OvlRec &r = out[k2];
			const uint32_t rid = r.qname, tl = ix.len[rid], tp = r.match, longer = tl > qlen ? tl : qlen;
			uint32_t l = k2;
			for (uint32_t j = 0; j < k2; ++j)
				if (out[j].qname == rid) { l = j; break; }
			if (l != k2) r.match = 1;
			OvlRec &head = out[l];
			if (l != k2 && (head.match == 2u || (double)(int32_t)r.tname < (double)(int32_t)head.tname * 0.8 || r.tname < longer / 3)) continue;
			if ((int32_t)(r.qe - r.qs) >= P.minlen && (float)tp >= (float)(int32_t)r.tname * P.minide && tp >= (uint32_t)P.minmatch) {
				if (dovetail_class((int)r.rev, r.qs, r.qe, qlen, r.ts, r.te, tl, P.maxhan1, 0)) {
					if (head.match == 3u) c--;
					head.match = tp;
				} else if (r.qs <= (uint32_t)P.maxhan2 && r.qe + (uint32_t)P.maxhan2 >= qlen) {
					head.match = 3u;
					if (++c >= 2) break; // MAX_CON
				}
			}
		}
		// the writer's record filter (map.c:1305-1309); the dovetail / contained filter that follows it keeps state over the
		// whole run and stays on the host (csrc/ovl_step2.cpp)
		OvlRec10 *out10 = recs10 + (out - recs);
		uint32_t n10 = 0;
		for (uint32_t k2 = 0; k2 < n_out; ++k2) {
			const OvlRec r = out[k2];
			const uint32_t tl = ix.len[r.qname];
			const int32_t ml = (int32_t)r.match, bl = (int32_t)r.tname;
			if (!(((int32_t)(r.qe - r.qs) >= P.minlen || ml == bl) && (ml == 3 || ((float)ml >= (float)bl * P.minide && ml >= P.minmatch)) &&
			      bl >= (int32_t)qlen / 50 && bl >= (int32_t)tl / 50)) continue;
			OvlRec10 o;
			o.rev = r.rev, o.qname = qid, o.qs = r.qs, o.qe = r.qe, o.qlen = qlen, o.tname = ix.id[r.qname], o.ts = r.ts, o.te = r.te, o.tlen = tl;
			o.identity = (uint32_t)((uint64_t)(uint32_t)ml * 10000ull / (uint64_t)(uint32_t)bl);
			out10[n10++] = o;
		}
		n_out = n10;
	}
	if (lane == 0) n_rec[rl] = n_out, n_chain[rl] = (uint32_t)n_u;
}

void launch_hits(const uint64_t *r_aoff, uint32_t n_reads, uint32_t read_base, const uint64_t *ax, const uint64_t *ay, const IndexDev &ix,
                 const QueryDev &q, const OvlParams &P, const int32_t *f, const int32_t *p, int32_t *v, int32_t *t, uint64_t *u,
                 uint64_t *bx, uint64_t *by, uint64_t *wx, uint64_t *wy, uint32_t *tables, void *stacks, const uint32_t *n_end,
                 OvlRec *recs, uint32_t *n_rec, uint32_t *n_chain, OvlRec10 *recs10, uint64_t *cx, uint64_t *cy, uint32_t *n_ca, hipStream_t s)
{
	static const bool want_prof = getenv("NDGPU_K5_PROF") != nullptr;
	unsigned long long *prof = nullptr;
	if (want_prof && n_reads && hipMalloc((void**)&prof, 16 * sizeof(unsigned long long)) == hipSuccess) (void)hipMemsetAsync(prof, 0, 16 * sizeof(unsigned long long), s);
	if (n_reads) ND_LAUNCH(hits_kernel, dim3(n_reads), dim3(64), 0, s, r_aoff, n_reads, read_base, ax, ay, ix, q, P, f, p,
	                                v, t, u, bx, by, wx, wy, tables, (SortJob*)stacks, n_end, recs, n_rec, n_chain, recs10, cx, cy, n_ca, prof);
	if (prof) {
		unsigned long long h[16];
		(void)hipStreamSynchronize(s);
		(void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
		(void)hipFree(prof);
		static const char *name[7] = {"heap sort of chain ends", "backtrack", "copy chains", "sort chains by x", "hash keys", "sort by hash", "records"};
		for (int i = 0; i < 7; ++i) fprintf(stderr, "[k5 prof] %-26s sum %9.3f ms  longest read %8.3f ms\n", name[i], h[2 * i] * 1e-5, h[2 * i + 1] * 1e-5);
	}
}

__global__ void compact_anchors_kernel(const uint64_t *__restrict__ r_aoff, uint32_t n_reads, const uint64_t *__restrict__ cx,
                                       const uint64_t *__restrict__ cy, const uint32_t *__restrict__ n_ca, const uint64_t *__restrict__ ca_off,
                                       uint64_t *__restrict__ dx, uint64_t *__restrict__ dy)
{
	const uint32_t rl = blockIdx.x;
	if (rl >= n_reads) return;
	const uint64_t src = r_aoff[rl], dst = ca_off[rl];
	for (uint32_t i = threadIdx.x; i < n_ca[rl]; i += blockDim.x) dx[dst + i] = cx[src + i], dy[dst + i] = cy[src + i];
}

void launch_compact_anchors(const uint64_t *r_aoff, uint32_t n_reads, const uint64_t *cx, const uint64_t *cy, const uint32_t *n_ca,
                            const uint64_t *ca_off, uint64_t *dx, uint64_t *dy, hipStream_t s)
{
	if (n_reads) ND_LAUNCH(compact_anchors_kernel, dim3(n_reads), dim3(256), 0, s, r_aoff, n_reads, cx, cy, n_ca, ca_off, dx, dy);
}

// gather per-read record runs into one dense array
__global__ void compact_recs_kernel(const uint64_t *__restrict__ r_aoff, uint32_t n_reads, int min_cnt, const OvlRec *__restrict__ recs,
                                    const uint32_t *__restrict__ n_rec, const uint64_t *__restrict__ rec_off, OvlRec *__restrict__ dense)
{
	const uint32_t rl = blockIdx.x;
	if (rl >= n_reads) return;
	const OvlRec *src = recs + r_aoff[rl] / (uint64_t)(min_cnt > 1 ? min_cnt : 1) + rl;
	OvlRec *dst = dense + rec_off[rl];
	for (uint32_t i = threadIdx.x; i < n_rec[rl]; i += blockDim.x) dst[i] = src[i];
}

__global__ void compact_recs10_kernel(const uint64_t *__restrict__ r_aoff, uint32_t n_reads, int min_cnt, const OvlRec10 *__restrict__ recs,
                                      const uint32_t *__restrict__ n_rec, const uint64_t *__restrict__ rec_off, OvlRec10 *__restrict__ dense)
{
	const uint32_t rl = blockIdx.x;
	if (rl >= n_reads) return;
	const OvlRec10 *src = recs + r_aoff[rl] / (uint64_t)(min_cnt > 1 ? min_cnt : 1) + rl;
	OvlRec10 *dst = dense + rec_off[rl];
	for (uint32_t i = threadIdx.x; i < n_rec[rl]; i += blockDim.x) dst[i] = src[i];
}

void launch_compact_recs10(const uint64_t *r_aoff, uint32_t n_reads, int min_cnt, const OvlRec10 *recs, const uint32_t *n_rec,
                           const uint64_t *rec_off, OvlRec10 *dense, hipStream_t s)
{
	if (n_reads) ND_LAUNCH(compact_recs10_kernel, dim3(n_reads), dim3(64), 0, s, r_aoff, n_reads, min_cnt, recs, n_rec, rec_off, dense);
}

void launch_compact_recs(const uint64_t *r_aoff, uint32_t n_reads, int min_cnt, const OvlRec *recs, const uint32_t *n_rec,
                         const uint64_t *rec_off, OvlRec *dense, hipStream_t s)
{
	if (n_reads) ND_LAUNCH(compact_recs_kernel, dim3(n_reads), dim3(64), 0, s, r_aoff, n_reads, min_cnt, recs, n_rec, rec_off, dense);
}

// ------------------------------------------------------------------------------------------------
// --mode 3 (HiFi): nd_extend_ends (minimap2/map.c:385-482, called from map.c:919-928).  Every hit is extended into the
// unaligned read ends with the greedy O(ND) extension (extend_rev on the query's 5' side, extend_fwd on its 3' side,
// lib/align.c:256-426): over at most 2 x the shorter overhang of the target, edit budget overhang / 4 capped at ide_ml,
// band 500.  Two independent problems per hit; one lane owns one problem (they are short: the overhang left by a chain's
// last minimizer), its furthest-reaching array in an HBM slice that the host zeroes (clean_V).  Bases come straight from
// the resident 2-bit reads; on a reverse hit the target slice is read backwards and complemented.

struct ExtGeom {
	int32_t subq, subt, max_d; // problem sizes (subt already cut to 2 x the shorter overhang)
	int32_t t_st, t_en;        // target slice [t_st, t_en)
	int32_t q_st;              // first query base of the query slice
	bool t_low;                // the target overhang below ts (else above te)
};

__device__ __forceinline__ bool ext_geometry(const OvlRec &r, int side, uint32_t qlen, uint32_t tlen, const OvlParams &P, ExtGeom &g)
{
	if (P.dvt && !dovetail_class((int)r.rev, r.qs, r.qe, qlen, r.ts, r.te, tlen, P.maxhan1 * 3, P.maxhan2 * 3)) return false;
	const bool left_q = side == 0;
	g.t_low = r.rev ? !left_q : left_q;
	g.subq = left_q ? (int32_t)r.qs : (int32_t)(qlen - r.qe);
	g.subt = g.t_low ? (int32_t)r.ts : (int32_t)(tlen - r.te);
	const int32_t minlen = g.subt > g.subq ? g.subq : g.subt;
	if (minlen < 10) return false;
	g.max_d = minlen / 4 > P.ide_ml ? P.ide_ml : (minlen > 20 ? minlen / 4 : minlen);
	if (g.subt > (minlen << 1)) {
		g.subt = minlen << 1;
		if (g.t_low) g.t_st = (int32_t)r.ts - g.subt, g.t_en = (int32_t)r.ts;
		else g.t_st = (int32_t)r.te, g.t_en = (int32_t)r.te + g.subt;
	} else {
		if (g.t_low) g.t_st = 0, g.t_en = (int32_t)r.ts;
		else g.t_st = (int32_t)r.te, g.t_en = (int32_t)tlen;
	}
	g.q_st = left_q ? 0 : (int32_t)r.qe;
	return true;
}

__global__ void ext_size_kernel(const OvlRec *__restrict__ recs, uint64_t n, const uint32_t *__restrict__ qlen, const uint32_t *__restrict__ tlen,
                                OvlParams P, uint32_t *__restrict__ need)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= 2 * n) return;
	const OvlRec r = recs[t >> 1];
	ExtGeom g;
	need[t] = ext_geometry(r, (int)(t & 1), qlen[r.qname], tlen[r.tname], P, g) ? 2u * (uint32_t)(g.max_d + 2) + 2u : 0u;
}

// (x + y) * d_factor - d with the product rounded to float before the subtraction, as the host code computes it (a fused
// multiply-add would keep the exact product and flip near-ties of the peak test)
__device__ __forceinline__ float ext_score(int xy, float f, int d)
{
#pragma clang fp contract(off)
	float p = (float)xy * f;
	asm volatile("" : "+v"(p));
	return p - (float)d;
}

__device__ __forceinline__ int base2(const uint32_t *__restrict__ w, int i) { return (int)(w[i >> 4] >> (30 - 2 * (i & 15)) & 3u); }

__global__ void __launch_bounds__(64) ext_ends_kernel(const OvlRec *__restrict__ recs, uint64_t t0, uint64_t t1, const uint32_t *__restrict__ qwords,
                                                       const uint64_t *__restrict__ qwoff, const uint32_t *__restrict__ qlen,
                                                       const uint32_t *__restrict__ twords, const uint64_t *__restrict__ twoff,
                                                       const uint32_t *__restrict__ tlen, OvlParams P, const uint64_t *__restrict__ fr_off,
                                                       uint64_t fr_base, int32_t *__restrict__ fr_pool, int32_t *__restrict__ ext_x,
                                                       int32_t *__restrict__ ext_y)
{
	const uint64_t t = t0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= t1) return;
	const OvlRec r = recs[t >> 1];
	ExtGeom g;
	const int side = (int)(t & 1);
	if (!ext_geometry(r, side, qlen[r.qname], tlen[r.tname], P, g)) { ext_x[t] = 0, ext_y[t] = 0; return; }
	const uint32_t *__restrict__ Q = qwords + qwoff[r.qname];
	const uint32_t *__restrict__ T = twords + twoff[r.tname];
	int32_t *fr = fr_pool + (fr_off[t] - fr_base);
	const bool back = side == 0; // extend_rev: both strings are consumed from their 3' ends
	const int ql = g.subq, tl = g.subt, off = g.max_d + 2;
	// string positions -> read positions: query char j = Q[q_st + j]; target char j = T[t_st + j], or on a reverse hit the
	// complement of T[t_en - 1 - j]
	const int comp = r.rev ? 3 : 0;
	int lo = 0, hi = 0, reach = -1, done = 0, o1 = 0, o2 = 0;
	float peak = 0;
	for (int d = 0; d < g.max_d && hi - lo <= 500 && !done; ++d) {
		for (int k = lo; k <= hi; k += 2) {
			int x;
			if (k == lo || (k != hi && fr[k - 1 + off] < fr[k + 1 + off])) x = fr[k + 1 + off];
			else x = fr[k - 1 + off] + 1;
			int y = x - k;
			while (x < ql && y < tl) {
				const int jq = back ? ql - x - 1 : x, jt = back ? tl - y - 1 : y;
				const int cq = base2(Q, g.q_st + jq);
				const int ct = r.rev ? base2(T, g.t_en - 1 - jt) ^ comp : base2(T, g.t_st + jt);
				if (cq != ct) break;
				++x, ++y;
			}
			fr[k + off] = x;
			if (x + y > reach) {
				reach = x + y;
				const float score = ext_score(x + y, P.d_factor, d);
				if (score > peak) peak = score, o1 = x, o2 = y;
				else if (score < peak - 30) { done = 2; break; }
			}
			if (x >= ql || y >= tl) {
				if (ext_score(x + y, P.d_factor, d) > 0) o1 = x, o2 = y;
				done = 1;
				break;
			}
		}
		if (done) break;
		int nlo = hi, nhi = lo; // band re-centring (lib/align.c:473-489)
		for (int k2 = lo; k2 < nlo; k2 += 2)
			if (fr[k2 + off] * 2 - k2 >= reach - 150) nlo = k2;
		for (int k2 = hi; k2 > nhi; k2 -= 2)
			if (fr[k2 + off] * 2 - k2 >= reach - 150) nhi = k2;
		hi = nhi + 1, lo = nlo - 1;
	}
	ext_x[t] = o1, ext_y[t] = o2;
}

// coordinates moved by the two extensions, names restored, then the step-1 output filter (minimap2/map.c:1297-1304)
__global__ void ext_apply_kernel(OvlRec *__restrict__ recs, uint64_t n, const int32_t *__restrict__ ext_x, const int32_t *__restrict__ ext_y,
                                 const uint32_t *__restrict__ qid, const uint32_t *__restrict__ qlen, const uint32_t *__restrict__ tid,
                                 const uint32_t *__restrict__ tlen, OvlParams P, uint32_t *__restrict__ keep)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	OvlRec r = recs[i];
	const uint32_t ql = qlen[r.qname], tl = tlen[r.tname];
	// 5' side of the query: qs moves down; the target moves below ts (forward hit) or above te (reverse hit)
	r.qs -= (uint32_t)ext_x[2 * i];
	if (r.rev) r.te += (uint32_t)ext_y[2 * i]; else r.ts -= (uint32_t)ext_y[2 * i];
	r.qe += (uint32_t)ext_x[2 * i + 1];
	if (r.rev) r.ts -= (uint32_t)ext_y[2 * i + 1]; else r.te += (uint32_t)ext_y[2 * i + 1];
	bool ok = (int32_t)(r.qe - r.qs) >= P.minlen;
	if (ok && P.dvt && !dovetail_class((int)r.rev, r.qs, r.qe, ql, r.ts, r.te, tl, P.maxhan1, P.maxhan2)) ok = false;
	r.qname = qid[r.qname], r.tname = tid[r.tname];
	recs[i] = r;
	keep[i] = ok ? 1u : 0u;
}

__global__ void scatter_recs_kernel(const OvlRec *__restrict__ recs, uint64_t n, const uint32_t *__restrict__ keep, const uint64_t *__restrict__ pos,
                                    OvlRec *__restrict__ out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && keep[i]) out[pos[i]] = recs[i];
}

void launch_ext_size(const OvlRec *recs, uint64_t n, const uint32_t *qlen, const uint32_t *tlen, const OvlParams &P, uint32_t *need,
                     hipStream_t s)
{
	if (n) ND_LAUNCH(ext_size_kernel, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, s, recs, n, qlen, tlen, P, need);
}

void launch_ext_ends(const OvlRec *recs, uint64_t t0, uint64_t t1, const uint32_t *qwords, const uint64_t *qwoff, const uint32_t *qlen,
                     const uint32_t *twords, const uint64_t *twoff, const uint32_t *tlen, const OvlParams &P, const uint64_t *fr_off,
                     uint64_t fr_base, int32_t *fr_pool, int32_t *ext_x, int32_t *ext_y, hipStream_t s)
{
	if (t1 > t0) ND_LAUNCH(ext_ends_kernel, dim3((unsigned)((t1 - t0 + 63) / 64)), dim3(64), 0, s, recs, t0, t1, qwords, qwoff, qlen,
	                                twords, twoff, tlen, P, fr_off, fr_base, fr_pool, ext_x, ext_y);
}

void launch_ext_apply(OvlRec *recs, uint64_t n, const int32_t *ext_x, const int32_t *ext_y, const uint32_t *qid, const uint32_t *qlen,
                      const uint32_t *tid, const uint32_t *tlen, const OvlParams &P, uint32_t *keep, hipStream_t s)
{
	if (n) ND_LAUNCH(ext_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, recs, n, ext_x, ext_y, qid, qlen, tid, tlen,
	                          P, keep);
}

void launch_scatter_recs(const OvlRec *recs, uint64_t n, const uint32_t *keep, const uint64_t *pos, OvlRec *out, hipStream_t s)
{
	if (n) ND_LAUNCH(scatter_recs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, recs, n, keep, pos, out);
}


} // namespace ndovl
