// ovl_device.h -- device-side types of the overlap engine and the launcher prototypes
// (kernels: ovl_kernels.hip; host orchestration + C ABI: ovl_engine.hip).
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <hip/hip_runtime.h>

#include "nd_lockstep.h"

namespace ndovl {

// The stage's streams run at the device's highest priority: when the stage of the next seed file shares the device with the consensus of
// this one (stage.StagePipeline), its kernels are the short ones -- 65 ms of device time against 580 -- and the consensus of the seed
// file after next waits for their result.  NDGPU_OVL_PRIO=0: the runtime's default priority (the A/B knob).
inline hipError_t create_stage_stream(hipStream_t *s)
{
	static const bool plain = getenv("NDGPU_OVL_PRIO") && atoi(getenv("NDGPU_OVL_PRIO")) == 0;
	int least = 0, greatest = 0;
	if (plain || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return hipStreamCreate(s);
	return hipStreamCreateWithPriority(s, hipStreamDefault, greatest);
}


// Every device operation of the overlap library reports its failure by throwing (defined in ovl_engine.hip): a rocPRIM primitive
// that returns an error, a kernel launch the runtime refuses.  Until round 4 the primitives' return values were dropped -- and a
// primitive that fails also clears the runtime's sticky error, so the stage-end hipGetLastError() saw nothing: on a device short
// of memory a sort or a scan that never ran left its output buffer as it was and the call returned fewer records, silently.
// Out of memory is noted for ndgpu_ovl_last_error() (1), so that the caller can release memory and try again.
void device_check(int hip_error, const char *what);
// Test hook: NDGPU_OVL_FAIL_AT=k makes the k-th checked device operation of the process (block-pool allocation, rocPRIM primitive,
// kernel launch) fail as if the device were out of memory; read at every operation, so a test can move it between calls.
bool fault_injected();
#define ND_LAUNCH(...)                                                                     \
	do {                                                                                   \
		if (ndovl::fault_injected()) ndovl::device_check((int)hipErrorOutOfMemory, __func__); \
		hipLaunchKernelGGL(__VA_ARGS__);                                                   \
		ndovl::device_check((int)hipGetLastError(), __func__);                             \
	} while (0)

constexpr uint64_t kSeedTandem = 1ULL << 42; // MM_SEED_TANDEM (minimap2/mmpriv.h:20)
constexpr uint64_t kSeedSelf = 1ULL << 43;   // MM_SEED_SELF   (minimap2/mmpriv.h:21)
constexpr uint64_t kSeedLongJoin = 1ULL << 40; // MM_SEED_LONG_JOIN (minimap2/mmpriv.h:18)

struct OvlParams {
	int32_t k, w, hpc;
	int32_t no_diag, no_dual;
	int32_t min_cnt, min_sc, bw, max_gap, max_skip, max_iter;
	int32_t minlen, dvt, maxhan1, maxhan2;
	int32_t mode3;     // --mode 3: chain ends trimmed, hits extended into the unaligned read ends before the output filter
	int32_t ide_ml;    // mm_mapopt_t::ide_ml (6000): cap of the extension's edit budget
	float d_factor;    // --df (0.1)
	int32_t step2;     // --step 2 (corrected reads): per-target marking + length / identity / block filters, 10-field records
	int32_t minmatch;  // --step 2: 100
	float minide;      // --step 2: 0.05
	int32_t provisional; // K5 stops at the hits themselves (target number, block length and match count in the name / match
	                     // fields, hit order): the passes of --step 2's re-alignment, whose marking and filters follow on the host
	int32_t nameless;    // the query has no name (mm_map(..., qname = 0), minimap2/map.c:1052,1088): no self test in K5
	int32_t thin;        // the chaining is mm_chain_dp_nextdenovo (--step 2 --mode 1's one-read-index mappings): beyond 100,000 anchors the
	                     // anchors of crowded target positions are dropped first (thin_anchors_kernel, minimap2/chain.c:185-226)
	int32_t chains;      // -c: K5 hands out the chains themselves -- every hit (self hits too, mm_align_skeleton aligns them) as
	                     // (strand, target, a[] offset, anchor count, chain score, hash) in hit order, and the chained anchors in the
	                     // order of the reference's a[] (chains by the x of their first anchor, minimap2/chain.c:150-160)
};

// minimizer index of the target reads, resident in HBM
struct HashSlot { unsigned long long key1; uint32_t start, cnt; };
// slot of a key in a table of `size` slots (any size: the mixed key's top 32 bits scaled into [0, size); size < 2^32)
__host__ __device__ inline uint64_t hash_slot_of(uint64_t key, uint64_t size) { return ((key * 0x9E3779B97F4A7C15ull >> 32) * size) >> 32; }

struct IndexDev {
	uint64_t n_keys;
	const uint64_t *ukey;    // distinct minimizers (hash value), ascending
	const uint64_t *ustart;  // n_keys + 1: first occurrence of each key in pos[]
	const uint64_t *pos;     // read<<32 | last base<<1 | strand, ascending inside a key
	const uint32_t *len;     // target read lengths
	const uint32_t *id;      // numeric read names
	const uint64_t *namekey; // order-preserving key of the decimal name string (strcmp order)
	const uint32_t *bucket;  // 2^kBucketBits + 1 entries: first key index of every top-bits bucket (narrows the binary search)
	uint32_t bucket_shift;   // key >> bucket_shift = bucket
	// open-addressing table over the distinct keys (nullptr: none -- the lookup is the bucket + binary search): one 16-byte slot
	// {key + 1 (0 = empty), first occurrence, occurrences}, linear probing, two thirds full -- ONE cache line per lookup where the
	// bucket table, the binary search's probes and the start offsets were five or six (minimap2/index.c:81-98 probes a hash table too)
	const HashSlot *htab;
	uint64_t hsize;          // slots (1.5 x the keys)
};

constexpr int kBucketBits = 22;

// query side (whole query set of one ndgpu_ovl_map call)
struct QueryDev {
	const uint32_t *len, *id, *hash;
	const uint64_t *namekey;
	const uint64_t *m_off;   // n_q + 1 offsets into the minimizer arrays
	// re-alignment (--step 2 --mode 2): query read i is mapped against want[want_off[i] .. want_off[i + 1]) only -- index-local
	// read numbers in the order the reference puts them into its per-thread mini-index (minimap2/index.c:434-575); a hit's
	// target number is then the POSITION in that list.  nullptr: the whole index.
	const uint64_t *want_off;
	const uint32_t *want;
	// -f FLOAT,INT (re-chaining, minimap2/map.c:553-575): the occurrence threshold of every query read; nullptr: the launch's one
	const int32_t *read_mid;
	// ... and, in the pass that tells which reads are seeded again: rep[i] != 0 once a minimizer of query read i was dropped for its
	// occurrences (the reference's `rep_len > 0`, collect_matches, map.c:101-113); nullptr: nobody asks
	uint32_t *rep;
};

// sort key of an anchor: | read (batch local) | strand | target read | target position |
struct KeyLayout {
	uint32_t pos_bits;   // bits of the target position field
	uint32_t rev_shift;  // = pos_bits + bits of the target read field
	uint32_t read_shift; // = rev_shift + 1
	uint32_t read_base;  // first query read of the batch
	uint32_t total_bits;
};

struct OvlRec { uint32_t rev, qname, qs, qe, tname, ts, te, match; };
struct OvlRec10 { uint32_t rev, qname, qs, qe, qlen, tname, ts, te, tlen, identity; }; // `overlap_i`, lib/ovl.h (the --step 2 record)
struct SketchTile { uint32_t read, start; }; // one block of K1: symbols [start, start + tile) of a read

size_t sketch_smem(int w);
void launch_sketch(bool fill, const uint32_t *words, const uint64_t *woff, const uint32_t *len, const uint32_t *order,
                   uint32_t n_reads, const OvlParams &P, int rid_is_index, const uint64_t *out_off, uint64_t *out_x,
                   uint64_t *out_y, uint32_t *out_read, uint32_t *out_cnt, hipStream_t s);
void launch_run_compact(bool fill, const uint32_t *words, const uint64_t *woff, const uint32_t *len, const SketchTile *tiles,
                        uint32_t n_tiles, const uint64_t *tile_prefix, const uint32_t *first_tile, const uint64_t *roff, uint32_t *tile_cnt,
                        uint8_t *sym, uint32_t *rstart, uint32_t *n_sym, hipStream_t s);
void launch_sketch_tiles(bool fill, bool hpc, const uint32_t *words, const uint64_t *woff, const uint32_t *len, const uint8_t *sym,
                         const uint32_t *rstart, const uint64_t *roff, const uint32_t *n_sym, const SketchTile *tiles, uint32_t n_tiles,
                         const OvlParams &P, int rid_is_index, const uint64_t *tile_off, uint32_t *tile_cnt, uint64_t *out_x, uint64_t *out_y,
                         uint32_t *out_read, hipStream_t s);
int sketch_tile_symbols();
void launch_pack_2bit(const uint8_t *ascii, const uint64_t *a_off, const uint32_t *len, const uint64_t *w_off, uint32_t n_reads, uint64_t n_words,
                      uint32_t *words, hipStream_t s);
void launch_gather_u64(const uint64_t *src, const uint32_t *idx, uint32_t n, uint64_t *dst, hipStream_t s);
void launch_shift_keys(const uint64_t *x, uint64_t *key, uint64_t n, hipStream_t s);
void launch_build_buckets(const uint64_t *ukey, uint64_t n_keys, uint32_t shift, uint32_t *bucket, hipStream_t s);
void launch_build_hash(const uint64_t *ukey, const uint64_t *ustart, uint64_t n_keys, HashSlot *tab, uint64_t size, hipStream_t s);  // tab zeroed by the caller

int sort_pairs_u64(void *tmp, size_t &tmp_bytes, const uint64_t *kin, uint64_t *kout, const uint64_t *vin, uint64_t *vout,
                   size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t s);
int sort_keys_u32(void *tmp, size_t &tmp_bytes, const uint32_t *kin, uint32_t *kout, size_t n, hipStream_t s);
int rle_u64(void *tmp, size_t &tmp_bytes, const uint64_t *kin, size_t n, uint64_t *uniq, uint32_t *cnt, uint64_t *n_runs,
            hipStream_t s);
int exscan_u32_to_u64(void *tmp, size_t &tmp_bytes, const uint32_t *in, uint64_t *out, size_t n, hipStream_t s);

void launch_seed_count(const uint64_t *mx, const uint64_t *my, const uint32_t *m_read, uint64_t n_m, const IndexDev &ix,
                       const QueryDev &q, const OvlParams &P, int mid_occ, uint32_t *m_start, uint32_t *m_cnt, uint32_t *m_surv,
                       hipStream_t s);
void launch_seed_fill(const uint64_t *mx, const uint64_t *my, const uint32_t *m_read, uint64_t m0, uint64_t m1, const IndexDev &ix,
                      const QueryDev &q, const OvlParams &P, const uint32_t *m_start, const uint32_t *m_cnt, const uint64_t *a_off,
                      uint64_t a_base, const KeyLayout &L, uint64_t *ckey, uint64_t *ay, hipStream_t s);
void launch_gather_read_off(const uint64_t *a_off, const uint64_t *m_off, uint32_t n_reads, uint64_t n_m, uint64_t total,
                            uint64_t *r_aoff, hipStream_t s);
void launch_local_off(const uint64_t *r_aoff_all, uint32_t r0, uint32_t n, uint64_t *r_aoff, hipStream_t s);
void launch_anchor_decode(const uint64_t *skey, uint64_t n, const KeyLayout &L, uint64_t *ax, uint32_t *tie_flag, uint64_t *segval,
                          hipStream_t s);
int incl_max_scan_u64(void *tmp, size_t &tmp_bytes, const uint64_t *in, uint64_t *out, size_t n, hipStream_t s);
void launch_slab_flag(const uint64_t *skey, const uint64_t *segstart1, uint64_t n, const KeyLayout &L, const uint64_t *r_aoff,
                      uint32_t *flag, hipStream_t s);
void launch_slab_write(const uint64_t *skey, const uint32_t *flag, const uint64_t *rank, uint64_t n, const KeyLayout &L,
                       uint64_t *slab_i0, uint32_t *slab_read, hipStream_t s);
void launch_read_span(const uint64_t *r_aoff, uint32_t n_reads, const uint64_t *ay, float *avg_span, hipStream_t s);
void launch_sort_init(const uint32_t *tie_reads, uint32_t n_tie, const uint64_t *r_aoff, const uint64_t *ukey, const uint64_t *uy,
                      const KeyLayout &L, uint64_t *ax, uint64_t *ay, void *jobs, uint32_t *n_jobs, hipStream_t s);
void launch_sort_pass(const void *jobs, uint32_t n_jobs, uint64_t *x, uint64_t *y, uint64_t *tx, uint64_t *ty, uint32_t *gs, void *next,
                      uint32_t *n_next, hipStream_t s);
void launch_thin_anchors(const uint64_t *r_aoff, uint32_t n_reads, uint64_t *ax, int32_t *t, int32_t *v, const int32_t *read_mid, uint32_t read_base,
                         int32_t mid, hipStream_t s);
void launch_chain(const uint64_t *slab_i0, const uint32_t *slab_read, uint32_t n_slabs, uint64_t n_anchors, const uint64_t *r_aoff,
                  const float *read_avg_span, const uint64_t *ax, const uint64_t *ay, const OvlParams &P, int32_t *f, int32_t *p, int32_t *v,
                  unsigned long long *cells, hipStream_t s);
void launch_chain_ends(const uint64_t *r_aoff, uint32_t n_reads, const OvlParams &P, const int32_t *f, const int32_t *p, const int32_t *v,
                       int32_t *t, uint64_t *u, uint32_t *n_end, hipStream_t s);
void launch_hits(const uint64_t *r_aoff, uint32_t n_reads, uint32_t read_base, const uint64_t *ax, const uint64_t *ay, const IndexDev &ix,
                 const QueryDev &q, const OvlParams &P, const int32_t *f, const int32_t *p, int32_t *v, int32_t *t, uint64_t *u,
                 uint64_t *bx, uint64_t *by, uint64_t *wx, uint64_t *wy, uint32_t *tables, void *stacks, const uint32_t *n_end,
                 OvlRec *recs, uint32_t *n_rec, uint32_t *n_chain, OvlRec10 *recs10, uint64_t *cx, uint64_t *cy, uint32_t *n_ca, hipStream_t s);
// -c: the chained anchors of every read (K5 left n_ca[read] of them at the front of the read's slice of cx / cy), back to back
void launch_compact_anchors(const uint64_t *r_aoff, uint32_t n_reads, const uint64_t *cx, const uint64_t *cy, const uint32_t *n_ca,
                            const uint64_t *ca_off, uint64_t *dx, uint64_t *dy, hipStream_t s);
void launch_compact_recs10(const uint64_t *r_aoff, uint32_t n_reads, int min_cnt, const OvlRec10 *recs, const uint32_t *n_rec,
                           const uint64_t *rec_off, OvlRec10 *dense, hipStream_t s);
void launch_compact_recs(const uint64_t *r_aoff, uint32_t n_reads, int min_cnt, const OvlRec *recs, const uint32_t *n_rec,
                         const uint64_t *rec_off, OvlRec *dense, hipStream_t s);
size_t sort_job_bytes();

// --mode 3 (minimap2/map.c:385-482): two extension problems per hit (the query's 5' side, its 3' side)
void launch_ext_size(const OvlRec *recs, uint64_t n, const uint32_t *qlen, const uint32_t *tlen, const OvlParams &P, uint32_t *need,
                     hipStream_t s);
void launch_ext_ends(const OvlRec *recs, uint64_t t0, uint64_t t1, const uint32_t *qwords, const uint64_t *qwoff, const uint32_t *qlen,
                     const uint32_t *twords, const uint64_t *twoff, const uint32_t *tlen, const OvlParams &P, const uint64_t *fr_off,
                     uint64_t fr_base, int32_t *fr_pool, int32_t *ext_x, int32_t *ext_y, hipStream_t s);
void launch_ext_apply(OvlRec *recs, uint64_t n, const int32_t *ext_x, const int32_t *ext_y, const uint32_t *qid, const uint32_t *qlen,
                      const uint32_t *tid, const uint32_t *tlen, const OvlParams &P, uint32_t *keep, hipStream_t s);
void launch_scatter_recs(const OvlRec *recs, uint64_t n, const uint32_t *keep, const uint64_t *pos, OvlRec *out, hipStream_t s);

} // namespace ndovl
