// ovlsort_kernels.hip -- gfx950 kernels of the overlap sort / filter stage (util/ovl_sort.c path, raw reads).
//
//   S1 expand_flags / expand_write   step-1 records -> per-seed candidates in both directions
//                                    (ovl_sort.c:980-1037: pre-filters, seed lookup, "5 misses then stop" per file)
//   S2 rocPRIM LSD radix sorts       order (seed asc, match desc, span asc), stable (ovl_sort.c:246-261, 876-925)
//   S3 seed_filter_kernel            one wavefront per seed: 64-base coverage bins in LDS, candidates admitted
//                                    one after the other (ovl_sort.c:675-741), then the chimera / low-coverage
//                                    trimming and the .bl verdict (ovl_sort.c:316-383, 433-571)
// HBM/LDS integer work; the admission chain is sequential per seed, seeds are independent.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "ovl_device.h"

namespace ndovl {

constexpr int kBinShift = 6;

__global__ void expand_flags_kernel(const OvlRec *__restrict__ raw, uint64_t n, const uint32_t *__restrict__ seed_len, uint32_t n_ids,
                                    uint32_t *__restrict__ hit_q, uint32_t *__restrict__ hit_t, uint32_t *__restrict__ miss_q,
                                    uint32_t *__restrict__ miss_t)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const OvlRec r = raw[i];
	uint32_t hq = 0, ht = 0, mq = 0, mt = 0;
	if (!(r.qname == r.tname || r.qe - r.qs < 500 || r.te - r.ts < 500)) {
		hq = r.qname < n_ids && seed_len[r.qname] && seed_len[r.qname] >= r.qe;
		ht = r.tname < n_ids && seed_len[r.tname] && seed_len[r.tname] >= r.te;
		mq = !hq, mt = !ht;
	}
	hit_q[i] = hq, hit_t[i] = ht, miss_q[i] = mq, miss_t[i] = mt;
}

// file_of[i] = input file of record i; miss_*_scan = exclusive scans over all records; file_start[f] = first record of file f
__global__ void expand_count_kernel(uint64_t n, const uint32_t *__restrict__ file_of, const uint64_t *__restrict__ file_start,
                                    const uint32_t *__restrict__ hit_q, const uint32_t *__restrict__ hit_t,
                                    const uint64_t *__restrict__ miss_q_scan, const uint64_t *__restrict__ miss_t_scan,
                                    uint32_t *__restrict__ n_out)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t f0 = file_start[file_of[i]];
	// a side stops being collected once 5 of its records in this file missed the seed table
	const bool q = hit_q[i] && miss_q_scan[i] - miss_q_scan[f0] < 5;
	const bool t = hit_t[i] && miss_t_scan[i] - miss_t_scan[f0] < 5;
	n_out[i] = (uint32_t)q | (uint32_t)t << 1;
}

__global__ void expand_write_kernel(const OvlRec *__restrict__ raw, uint64_t n, const uint32_t *__restrict__ sel, const uint64_t *__restrict__ pos,
                                    OvlRec *__restrict__ cand, uint32_t *__restrict__ k_span, uint32_t *__restrict__ k_match,
                                    uint32_t *__restrict__ k_seed)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t s = sel[i];
	if (!s) return;
	const OvlRec r = raw[i];
	uint64_t o = pos[i];
	if (s & 1) {
		OvlRec c;
		c.rev = r.rev, c.qname = r.qname, c.qs = r.qs, c.qe = r.qe - 1, c.tname = r.tname, c.ts = r.ts, c.te = r.te - 1, c.match = r.match;
		cand[o] = c;
		k_span[o] = c.qe - c.qs, k_match[o] = ~c.match, k_seed[o] = c.qname;
		++o;
	}
	if (s & 2) {
		OvlRec c;
		c.rev = r.rev, c.qname = r.tname, c.qs = r.ts, c.qe = r.te - 1, c.tname = r.qname, c.ts = r.qs, c.te = r.qe - 1, c.match = r.match;
		cand[o] = c;
		k_span[o] = c.qe - c.qs, k_match[o] = ~c.match, k_seed[o] = c.qname;
	}
}

__global__ void sel_count_kernel(const uint32_t *__restrict__ sel, uint64_t n, uint32_t *__restrict__ cnt)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) cnt[i] = (sel[i] & 1) + (sel[i] >> 1 & 1);
}

__global__ void iota_kernel(uint32_t *__restrict__ a, uint64_t n)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) a[i] = (uint32_t)i;
}

__global__ void gather_u32_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ idx, uint64_t n, uint32_t *__restrict__ dst)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) dst[i] = src[idx[i]];
}

// seed boundaries in sorted order: flag[i] = 1 where a new seed starts
__global__ void seed_flag_kernel(const OvlRec *__restrict__ cand, const uint32_t *__restrict__ perm, uint64_t n, uint32_t *__restrict__ flag)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) flag[i] = i == 0 || cand[perm[i]].qname != cand[perm[i - 1]].qname;
}

__global__ void seed_start_kernel(const uint32_t *__restrict__ flag, const uint64_t *__restrict__ rank, uint64_t n, uint64_t *__restrict__ start)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && flag[i]) start[rank[i]] = i;
}

#define GRID1(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256)
// ---- the out-of-core form (csrc/ovlsort_engine.hip: sort_out_of_core): the records of ONE file arrive in pieces ----
// like expand_count_kernel for a piece of one file: carry_* = records of this file before the piece that missed the seed table
__global__ void expand_count_piece_kernel(uint64_t n, const uint32_t *__restrict__ hit_q, const uint32_t *__restrict__ hit_t,
                                          const uint64_t *__restrict__ miss_q_scan, const uint64_t *__restrict__ miss_t_scan, uint64_t carry_q,
                                          uint64_t carry_t, uint32_t *__restrict__ sel)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const bool q = hit_q[i] && carry_q + miss_q_scan[i] < 5;
	const bool t = hit_t[i] && carry_t + miss_t_scan[i] < 5;
	sel[i] = (uint32_t)q | (uint32_t)t << 1;
}

// candidates per seed (which seed ranges fit the device at a time)
__global__ void cand_hist_kernel(const OvlRec *__restrict__ raw, const uint32_t *__restrict__ sel, uint64_t n, uint32_t *__restrict__ hist)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t s = sel[i];
	if (s & 1) atomicAdd(hist + raw[i].qname, 1u);
	if (s & 2) atomicAdd(hist + raw[i].tname, 1u);
}

// the sides of a record whose seed lies in [lo, hi)
__global__ void range_sel_kernel(const OvlRec *__restrict__ raw, const uint8_t *__restrict__ sel, uint64_t n, uint32_t lo, uint32_t hi,
                                 uint32_t *__restrict__ out)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t s = sel[i];
	const OvlRec r = raw[i];
	out[i] = (uint32_t)((s & 1) && r.qname >= lo && r.qname < hi) | (uint32_t)((s & 2) && r.tname >= lo && r.tname < hi) << 1;
}

__global__ void narrow_u32_u8_kernel(const uint32_t *__restrict__ in, uint64_t n, uint8_t *__restrict__ out)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = (uint8_t)in[i];
}


void launch_expand_flags(const OvlRec *raw, uint64_t n, const uint32_t *seed_len, uint32_t n_ids, uint32_t *hq, uint32_t *ht, uint32_t *mq,
                         uint32_t *mt, hipStream_t s)
{
	if (n) ND_LAUNCH(expand_flags_kernel, GRID1(n), 0, s, raw, n, seed_len, n_ids, hq, ht, mq, mt);
}
void launch_expand_count(uint64_t n, const uint32_t *file_of, const uint64_t *file_start, const uint32_t *hq, const uint32_t *ht,
                         const uint64_t *mqs, const uint64_t *mts, uint32_t *sel, hipStream_t s)
{
	if (n) ND_LAUNCH(expand_count_kernel, GRID1(n), 0, s, n, file_of, file_start, hq, ht, mqs, mts, sel);
}
void launch_expand_count_piece(uint64_t n, const uint32_t *hq, const uint32_t *ht, const uint64_t *mqs, const uint64_t *mts, uint64_t carry_q,
                               uint64_t carry_t, uint32_t *sel, hipStream_t s)
{
	if (n) ND_LAUNCH(expand_count_piece_kernel, GRID1(n), 0, s, n, hq, ht, mqs, mts, carry_q, carry_t, sel);
}
void launch_cand_hist(const OvlRec *raw, const uint32_t *sel, uint64_t n, uint32_t *hist, hipStream_t s)
{
	if (n) ND_LAUNCH(cand_hist_kernel, GRID1(n), 0, s, raw, sel, n, hist);
}
void launch_range_sel(const OvlRec *raw, const uint8_t *sel, uint64_t n, uint32_t lo, uint32_t hi, uint32_t *out, hipStream_t s)
{
	if (n) ND_LAUNCH(range_sel_kernel, GRID1(n), 0, s, raw, sel, n, lo, hi, out);
}
void launch_narrow_u32_u8(const uint32_t *in, uint64_t n, uint8_t *out, hipStream_t s)
{
	if (n) ND_LAUNCH(narrow_u32_u8_kernel, GRID1(n), 0, s, in, n, out);
}
void launch_sel_count(const uint32_t *sel, uint64_t n, uint32_t *cnt, hipStream_t s)
{
	if (n) ND_LAUNCH(sel_count_kernel, GRID1(n), 0, s, sel, n, cnt);
}
void launch_expand_write(const OvlRec *raw, uint64_t n, const uint32_t *sel, const uint64_t *pos, OvlRec *cand, uint32_t *k_span,
                         uint32_t *k_match, uint32_t *k_seed, hipStream_t s)
{
	if (n) ND_LAUNCH(expand_write_kernel, GRID1(n), 0, s, raw, n, sel, pos, cand, k_span, k_match, k_seed);
}
void launch_iota(uint32_t *a, uint64_t n, hipStream_t s) { if (n) ND_LAUNCH(iota_kernel, GRID1(n), 0, s, a, n); }
void launch_gather_u32(const uint32_t *src, const uint32_t *idx, uint64_t n, uint32_t *dst, hipStream_t s)
{
	if (n) ND_LAUNCH(gather_u32_kernel, GRID1(n), 0, s, src, idx, n, dst);
}
void launch_seed_flag(const OvlRec *cand, const uint32_t *perm, uint64_t n, uint32_t *flag, hipStream_t s)
{
	if (n) ND_LAUNCH(seed_flag_kernel, GRID1(n), 0, s, cand, perm, n, flag);
}
void launch_seed_start(const uint32_t *flag, const uint64_t *rank, uint64_t n, uint64_t *start, hipStream_t s)
{
	if (n) ND_LAUNCH(seed_start_kernel, GRID1(n), 0, s, flag, rank, n, start);
}

int sort_pairs_u32(void *tmp, size_t &tmp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n,
                   hipStream_t s)
{
	if (tmp && fault_injected()) device_check((int)hipErrorOutOfMemory, __func__);
	device_check((int)rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0, 32, s), __func__);
	return 0;
}

// ------------------------------------------------------------------------------------------------
// S3: one wavefront per seed

__device__ __forceinline__ int wave_sum(int v) { for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d, 64); return v; }
__device__ __forceinline__ int wave_min(int v) { for (int d = 32; d; d >>= 1) { int o = __shfl_xor(v, d, 64); v = o < v ? o : v; } return v; }

constexpr uint32_t kSelf = 0x7fffffffu;  // kept[] entry of the synthetic self record
constexpr uint32_t kDropped = 0x80000000u;

__global__ void __launch_bounds__(64) seed_filter_kernel(const OvlRec *__restrict__ cand, const uint32_t *__restrict__ perm,
                                                          const uint64_t *__restrict__ seed_start, uint32_t n_seeds, uint64_t n_cand,
                                                          const uint32_t *__restrict__ seed_len, int max_bin_cov, int flank, int min_seed_len,
                                                          uint32_t max_bins, uint32_t *__restrict__ kept, OvlRec *__restrict__ out,
                                                          uint32_t *__restrict__ n_out, uint32_t *__restrict__ bl_id, uint8_t *__restrict__ bl_kind)
{
	extern __shared__ uint16_t bins[];
	const uint32_t sd = blockIdx.x;
	if (sd >= n_seeds) return;
	const int lane = threadIdx.x;
	const uint64_t c0 = seed_start[sd], c1 = sd + 1 < n_seeds ? seed_start[sd + 1] : n_cand;
	const uint32_t seed = cand[perm[c0]].qname;
	const uint32_t qlen = seed_len[seed];
	const int nb = (int)(qlen >> kBinShift) + 1;
	uint32_t *K = kept + c0 + sd;       // room for the self record + every candidate of the seed
	OvlRec *O = out + c0 + sd;
	for (int i = lane; i < nb && i < (int)max_bins; i += 64) bins[i] = 0;
	__syncthreads();

	// state of the admission chain (identical in every lane)
	uint8_t repeat_run = 1;
	uint32_t n_kept = 0, last_qs = 0, last_qe = qlen - 1, qcov = qlen, bins_touched = 0, bins_sum = 0, contained = 0;
	const uint32_t qcap = qlen * 150u;
	if (lane == 0) K[0] = kSelf;
	n_kept = 1;
	for (uint64_t c = c0; c < c1; ++c) {
		const uint32_t ci = perm[c];
		const OvlRec o = cand[ci];
		if (qcov > qcap || n_kept > 65535u - 1000u) continue;
		const int j = (int)((o.qs + 10) >> kBinShift), k = (int)((o.qe - 10) >> kBinShift);
		int label = 1;
		const int dqs = (int)(o.qs - last_qs), dqe = (int)(o.qe - last_qe);
		if ((j > 15 || k < nb - 16) && (dqs < 0 ? -dqs : dqs) < 50 && (dqe < 0 ? -dqe : dqe) < 50) label = repeat_run++ < 5 ? 2 : 0;
		if (!label) continue;
		int fresh = 0, lowest = 200, sum = 0;
		for (int i = j + 1 + lane; i <= k; i += 64) {
			int v = bins[i];
			if (!v) ++fresh;
			++v;
			if (v < lowest) lowest = v;
			if (v > 65535 - 1000) --v;
			bins[i] = (uint16_t)v;
			sum += v;
		}
		fresh = wave_sum(fresh), sum = wave_sum(sum), lowest = wave_min(lowest);
		// float quotients of the reference, formed in double and rounded once (exact for 24-bit operands)
		const float lhs = (float)((double)(float)sum / (double)(float)(k - j));
		const float dens = (float)((double)(float)bins_sum / (double)(float)bins_touched);
		const float clampd = dens > 10 ? dens : 10;                      // max(dens, 10) of the reference's macro
		const float lim = clampd > max_bin_cov ? (float)max_bin_cov : clampd; // min(.., max_bin_cov)
		const bool reject = (lowest > max_bin_cov || (double)lhs > 1.3 * (double)lim) && ((double)(o.qe - o.qs) <= qlen * 0.8);
		if (reject) {
			for (int i = j + 1 + lane; i <= k; i += 64) bins[i]--;
			ND_LOCKSTEP();  // the next record's lanes own other bins
			continue;
		}
		if (label != 2) repeat_run = 1;
		bins_touched += (uint32_t)fresh;
		bins_sum += (uint32_t)(k - j);
		last_qs = o.qs, last_qe = o.qe;
		qcov += o.qe - o.qs + 1;
		if (o.qname != o.tname && o.qs <= (uint32_t)flank && o.qe + (uint32_t)flank >= qlen) ++contained;
		if (lane == 0) K[n_kept] = ci;
		++n_kept;
	}
	__syncthreads();

	// end of the seed (lane 0 walks the bins; the few list scans are short)
	uint32_t chimera = 0;
	int lo = 0, hi = 0;
	if (lane == 0) {
		// coverage-shape chimera test
		{
			int label = 0, ll = 0, rl = 0;
			for (int i = 1; i < nb - 1; ++i) {
				if (bins[i] > 20 && ++ll) {
					if (label && ++rl >= 5) break;
				} else {
					const int l = i - 5 > 0 ? i - 5 : 0, r = i + 5 > nb - 1 ? nb - 1 : i + 5;
					const int mn = bins[l] > bins[r] ? bins[r] : bins[l];
					if (ll > 5 && (bins[l] > 20 || bins[r] > 20) && bins[i] <= (3 > mn / 5 ? 3 : mn / 5)) label = i;
				}
			}
			if (rl < 5) label = 0;
			chimera = (uint32_t)label;
		}
		if (chimera || !contained) {
			int j = 0;
			uint16_t *b = bins; // (first, last) pairs of low-coverage bin runs overwrite the front of bins[]
			if (qcov > qlen * 10u) {
				const int low = 4 > max_bin_cov / 10 ? max_bin_cov / 10 : 4;
				for (int i = 1; i < nb - 1; ++i) {
					if (b[i] < low) {
						if (lo == 0) lo = i;
						hi = i;
					} else if (lo) {
						if (chimera && chimera < (uint32_t)lo && ((!j) || chimera > b[j - 1])) b[j++] = (uint16_t)chimera, b[j++] = (uint16_t)chimera;
						b[j++] = (uint16_t)lo, b[j++] = (uint16_t)hi;
						lo = hi = 0;
					}
				}
				if (lo) {
					if (chimera && chimera < (uint32_t)lo && ((!j) || chimera > b[j - 1])) b[j++] = (uint16_t)chimera, b[j++] = (uint16_t)chimera;
					b[j++] = (uint16_t)lo, b[j++] = (uint16_t)hi;
				}
				if (chimera && (j == 0 || chimera > b[j - 1])) b[j++] = (uint16_t)chimera, b[j++] = (uint16_t)chimera;
			} else if (chimera) b[j++] = (uint16_t)chimera, b[j++] = (uint16_t)chimera;
			if (j) {
				int m = j, k = 0, i;
				if (b[0] < 5) m -= 2;
				if (b[j - 1] > nb - 5) m -= 2;
				if (m > 0) {
					m = b[0];
					for (i = 2; i < j; i += 2)
						if (b[i] - b[i - 1] > m) m = b[i] - b[i - 1], k = i;
					if (nb - b[i - 1] > m) {
						m = nb - b[i - 1];
						lo = b[i - 1], hi = nb;
					} else if (b[k + 1] > nb - 5) {
						lo = b[k - 1], hi = nb;
					} else if (k == 0 || b[k - 2] < 5) {
						lo = 0, hi = b[k];
					} else {
						lo = b[k - 1], hi = b[k];
					}
					lo = lo > 5 ? (lo - 5) << kBinShift : 0;
					hi = (hi + 5) << kBinShift;
					if (m > (min_seed_len >> kBinShift) * 2 / 3) {
						chimera = 0;
						for (uint32_t q = 1; q < n_kept; ++q) {
							const OvlRec o = cand[K[q]];
							if (o.qs < (uint32_t)lo || o.qe > (uint32_t)hi) K[q] |= kDropped;
						}
					} else chimera = 1;
				} else lo = hi = 0;
			}
		}
		if (qcov > qlen * 20u && !chimera && contained < 2) {
			// hot break ends: alignment ends piling up well inside the read (128-base bins)
			const int sh = kBinShift + 1;
			for (int i = 0; i < nb / 2 + 1; ++i) bins[i] = 0;
			int s0 = nb, e0 = 0, c = 0, t = 0;
			for (uint32_t q = 1; q < n_kept; ++q) {
				if (K[q] & kDropped) continue;
				const OvlRec o = cand[K[q]];
				++c;
				int x = (int)((o.qs + 10) >> sh);
				if (x < s0) s0 = x;
				bins[x]++;
				x = (int)((o.qe - 10) >> sh);
				if (x > e0) e0 = x;
				bins[x]++;
			}
			if (c > 20) {
				while (s0 < e0 && bins[s0] < 4) ++s0;
				while (e0 > s0 && bins[e0] < 4) --e0;
				int m = 0, ms = bins[s0], me = bins[e0];
				for (int i = s0; i < e0 + 1; ++i) {
					if (i < s0 + 5 && bins[i] > ms) ms = bins[i];
					if (i > e0 - 5 && bins[i] > me) me = bins[i];
					if (bins[i] > bins[m]) m = i;
				}
				if (m > s0 + 5 && m < e0 - 5 && (float)bins[m] > 1.f * (float)(ms > me ? ms : me) && ((c > 75 && m > c / 5) || (c < 75 && m > c / 2)))
					t = m << sh;
			}
			chimera = (uint32_t)t;
			if (!hi) hi = (int)qlen;
			if (chimera <= (uint32_t)(lo + (15 << kBinShift)) || chimera + (15u << kBinShift) >= (uint32_t)hi) chimera = 0;
		}
		// survivors, in order; the self record first
		uint32_t n = 0, cont = 0;
		for (uint32_t q = 0; q < n_kept; ++q) {
			const uint32_t e = K[q];
			OvlRec o;
			if (e == kSelf) {
				o.rev = 0, o.qname = o.tname = seed, o.qs = o.ts = 0, o.qe = o.te = qlen - 1, o.match = 0;
				if (!o.qe) continue;
			} else {
				if (e & kDropped) continue;
				o = cand[e];
			}
			O[n++] = o;
			if (o.qname != o.tname && o.qs <= (uint32_t)flank && o.qe + (uint32_t)flank >= qlen) ++cont;
		}
		n_out[sd] = n;
		bl_id[sd] = seed;
		bl_kind[sd] = cont >= 2 ? (uint8_t)'c' : chimera ? (uint8_t)'k' : (uint8_t)0;
	}
}

// S3 for `ovl_sort -H` (high-quality reads): encode_ovl_filter_hq (ovl_sort.c:616-655), del_repeat_alns (:389-431),
// check_chimer_hq (:287-314) and the -H branches of ovl_filter (:433-571).  Every candidate is collected (up to the caps)
// while the two halves of bins[] count alignment starts / ends per 128 bases; overlaps that start and end at hot break
// points are repeat-induced and dropped, the 64-base coverage is rebuilt over the rest (spans that would sit above
// 2 x max_bin_cov everywhere are dropped), a bin covered at most once that no overlap spans with 15 bins to spare marks a
// chimera, and the low-coverage trimming of the raw-read path follows.  The chain is sequential per seed (lane 0 walks it;
// seeds are independent wavefronts).
__global__ void __launch_bounds__(64) seed_filter_hq_kernel(const OvlRec *__restrict__ cand, const uint32_t *__restrict__ perm,
                                                             const uint64_t *__restrict__ seed_start, uint32_t n_seeds, uint64_t n_cand,
                                                             const uint32_t *__restrict__ seed_len, int max_bin_cov, int flank, int min_seed_len,
                                                             uint32_t max_bins, uint32_t *__restrict__ kept, OvlRec *__restrict__ out,
                                                             uint32_t *__restrict__ n_out, uint32_t *__restrict__ bl_id, uint8_t *__restrict__ bl_kind)
{
	extern __shared__ uint16_t bins[];
	const uint32_t sd = blockIdx.x;
	if (sd >= n_seeds) return;
	const int lane = threadIdx.x;
	const uint64_t c0 = seed_start[sd], c1 = sd + 1 < n_seeds ? seed_start[sd + 1] : n_cand;
	const uint32_t seed = cand[perm[c0]].qname;
	const uint32_t qlen = seed_len[seed];
	const int nb = (int)(qlen >> kBinShift) + 2;
	uint32_t *K = kept + c0 + sd;
	OvlRec *O = out + c0 + sd;
	for (int i = lane; i < nb && i < (int)max_bins; i += 64) bins[i] = 0;
	__syncthreads();
	if (lane != 0) return;

	// collection + break-point histogram
	const int sh2 = kBinShift + 1;
	const int off = 1 + (int)(qlen >> sh2);
	const uint32_t qcap = qlen * 150u * 6u;
	uint32_t n_kept = 1, qcov = qlen;
	K[0] = kSelf;
	for (uint64_t c = c0; c < c1; ++c) {
		const uint32_t ci = perm[c];
		const OvlRec o = cand[ci];
		if (qcov > qcap || n_kept > 65535u - 1000u) continue;
		bins[(o.qs + 10) >> sh2]++;
		bins[((o.qe - 10) >> sh2) + off]++;
		qcov += o.qe - o.qs + 1;
		K[n_kept++] = ci;
	}
	// repeat-induced overlaps
	{
		const uint32_t fl = flank > 100 ? (uint32_t)flank * 3u : 300u;
		for (uint32_t q = 1; q < n_kept; ++q) {
			const OvlRec o = cand[K[q]];
			if (o.qs <= fl && o.qe + fl >= qlen) continue;
			if (bins[(o.qs + 10) >> sh2] >= 5 && bins[((o.qe - 10) >> sh2) + off] >= 5) K[q] |= kDropped;
		}
		for (int i = 0; i < nb; ++i) bins[i] = 0;
		for (uint32_t q = 1; q < n_kept; ++q) {
			if (K[q] & kDropped) continue;
			const OvlRec o = cand[K[q]];
			const int j = (int)((o.qs + 10) >> kBinShift), k = (int)((o.qe - 10) >> kBinShift);
			int lowest = 65535;
			for (int t = j + 1; t <= k; ++t) {
				int v = (int)bins[t] + 1;
				if (v > 65535 - 1000) --v;
				bins[t] = (uint16_t)v;
				if (v < lowest) lowest = v;
			}
			if (lowest > 2 * max_bin_cov) {
				for (int t = j + 1; t <= k; ++t) bins[t]--;
				K[q] |= kDropped;
			}
		}
	}
	// chimera: an uncovered bin inside the covered part that no overlap spans
	uint32_t chimera = 0;
	{
		int l = 0, r = nb;
		while (l < nb && bins[l] < 2) ++l;
		while (r > 0 && bins[r - 1] < 2) --r;
		for (int i = l + 1; i < r - 1 && !chimera; ++i) {
			if (bins[i] > 1) continue;
			const uint32_t lo2 = (uint32_t)(i > l + 15 ? (i - 15) << kBinShift : l << kBinShift);
			const uint32_t hi2 = (uint32_t)(i + 15 < r ? (i + 15) << kBinShift : r << kBinShift);
			bool spanned = false;
			for (uint32_t q = 1; q < n_kept && !spanned; ++q) {
				if (K[q] & kDropped) continue;
				const OvlRec o = cand[K[q]];
				spanned = o.qs < lo2 && o.qe > hi2;
			}
			if (!spanned) chimera = (uint32_t)i;
		}
	}
	int lo = 0, hi = 0;
	if (chimera) {
		int j = 0;
		uint16_t *b = bins; // (first, last) pairs of low-coverage bin runs overwrite the front of bins[]
		if (qcov > qlen * 10u) {
			const int low = 4 > max_bin_cov / 10 ? max_bin_cov / 10 : 4;
			for (int i = 1; i < nb - 1; ++i) {
				if (b[i] < low) {
					if (lo == 0) lo = i;
					hi = i;
				} else if (lo) {
					if (chimera && chimera < (uint32_t)lo && ((!j) || chimera > b[j - 1])) b[j++] = (uint16_t)chimera, b[j++] = (uint16_t)chimera;
					b[j++] = (uint16_t)lo, b[j++] = (uint16_t)hi;
					lo = hi = 0;
				}
			}
			if (lo) {
				if (chimera && chimera < (uint32_t)lo && ((!j) || chimera > b[j - 1])) b[j++] = (uint16_t)chimera, b[j++] = (uint16_t)chimera;
				b[j++] = (uint16_t)lo, b[j++] = (uint16_t)hi;
			}
			if (chimera && (j == 0 || chimera > b[j - 1])) b[j++] = (uint16_t)chimera, b[j++] = (uint16_t)chimera;
		} else b[j++] = (uint16_t)chimera, b[j++] = (uint16_t)chimera;
		if (j) {
			int m = j, k = 0, i;
			if (b[0] < 5) m -= 2;
			if (b[j - 1] > nb - 5) m -= 2;
			if (m > 0) {
				m = b[0];
				for (i = 2; i < j; i += 2)
					if (b[i] - b[i - 1] > m) m = b[i] - b[i - 1], k = i;
				if (nb - b[i - 1] > m) {
					m = nb - b[i - 1];
					lo = b[i - 1], hi = nb;
				} else if (b[k + 1] > nb - 5) {
					lo = b[k - 1], hi = nb;
				} else if (k == 0 || b[k - 2] < 5) {
					lo = 0, hi = b[k];
				} else {
					lo = b[k - 1], hi = b[k];
				}
				lo = lo > 5 ? (lo - 5) << kBinShift : 0;
				hi = (hi + 5) << kBinShift;
				if (m > (min_seed_len >> kBinShift) * 2 / 3) {
					chimera = 0;
					for (uint32_t q = 1; q < n_kept; ++q) {
						if (K[q] & kDropped) continue;
						const OvlRec o = cand[K[q]];
						if (o.qs < (uint32_t)lo || o.qe > (uint32_t)hi) K[q] |= kDropped;
					}
				} else chimera = 1;
			}
		}
	}
	// survivors, in order; the self record first.  A containing overlap of high-quality reads must be >= 90 % matches.
	uint32_t n = 0, cont = 0;
	for (uint32_t q = 0; q < n_kept; ++q) {
		const uint32_t e = K[q];
		OvlRec o;
		if (e == kSelf) {
			o.rev = 0, o.qname = o.tname = seed, o.qs = o.ts = 0, o.qe = o.te = qlen - 1, o.match = 0;
			if (!o.qe) continue;
		} else {
			if (e & kDropped) continue;
			o = cand[e];
		}
		O[n++] = o;
		if (o.qname != o.tname && o.qs <= (uint32_t)flank && o.qe + (uint32_t)flank >= qlen && (double)o.match >= (double)(o.qe - o.qs + 1) * 0.9) ++cont;
	}
	n_out[sd] = n;
	bl_id[sd] = seed;
	bl_kind[sd] = cont >= 2 ? (uint8_t)'c' : chimera ? (uint8_t)'k' : (uint8_t)0;
}

void launch_seed_filter(const OvlRec *cand, const uint32_t *perm, const uint64_t *seed_start, uint32_t n_seeds, uint64_t n_cand,
                        const uint32_t *seed_len, int max_bin_cov, int flank, int min_seed_len, uint32_t max_bins, uint32_t *kept, OvlRec *out,
                        uint32_t *n_out, uint32_t *bl_id, uint8_t *bl_kind, bool hq, hipStream_t s)
{
	if (n_seeds && hq) ND_LAUNCH(seed_filter_hq_kernel, dim3(n_seeds), dim3(64), (size_t)max_bins * 2, s, cand, perm, seed_start, n_seeds,
	                                      n_cand, seed_len, max_bin_cov, flank, min_seed_len, max_bins, kept, out, n_out, bl_id, bl_kind);
	else if (n_seeds) ND_LAUNCH(seed_filter_kernel, dim3(n_seeds), dim3(64), (size_t)max_bins * 2, s, cand, perm, seed_start, n_seeds, n_cand,
	                                seed_len, max_bin_cov, flank, min_seed_len, max_bins, kept, out, n_out, bl_id, bl_kind);
}

// dense output: seed sd's records start at out + seed_start[sd] + sd
__global__ void compact_seed_recs_kernel(const uint64_t *__restrict__ seed_start, uint32_t n_seeds, const OvlRec *__restrict__ out,
                                         const uint32_t *__restrict__ n_out, const uint64_t *__restrict__ off, OvlRec *__restrict__ dense)
{
	const uint32_t sd = blockIdx.x;
	if (sd >= n_seeds) return;
	const OvlRec *src = out + seed_start[sd] + sd;
	OvlRec *dst = dense + off[sd];
	for (uint32_t i = threadIdx.x; i < n_out[sd]; i += blockDim.x) dst[i] = src[i];
}

void launch_compact_seed_recs(const uint64_t *seed_start, uint32_t n_seeds, const OvlRec *out, const uint32_t *n_out, const uint64_t *off,
                              OvlRec *dense, hipStream_t s)
{
	if (n_seeds) ND_LAUNCH(compact_seed_recs_kernel, dim3(n_seeds), dim3(64), 0, s, seed_start, n_seeds, out, n_out, off, dense);
}

} // namespace ndovl
