/* ndgpu_overlap.h -- C ABI of the MI355X overlap engine (libndgpu_overlap.so).
 *
 * Drop-in scope: the `minimap2-nd --step 1` path of NextDenovo v2.5.2 (all-vs-all raw-read overlap that feeds
 * ovl_sort -> nextcorrect).  Citations are relative to the reference tree.
 *
 *   reference interface                                   this library
 *   ----------------------------------------------------  -------------------------------------------
 *   mm_set_opt()            minimap2/options.c:84-97       ndgpu_ovl_opt_preset()
 *   main.c:190-193 (--step 1), :241,:328 (--dual), :337-343 (-f)   fields of ndgpu_ovl_opt
 *   mm_idx_gen()            minimap2/index.c:351-370       ndgpu_ovl_index_create()   (K1 sketch + K2 sort/group, HBM resident)
 *   mm_idx_cal_max_occ()    minimap2/index.c:170-191       ndgpu_ovl_index_mid_occ()
 *   mm_idx_destroy()        minimap2/index.c:55-79         ndgpu_ovl_index_destroy()
 *   util/ovl_sort.c (whole program, raw reads)               ndgpu_ovl_sort()  (S1 expand, S2 sort, S3 per-seed filter)
 *   mm_map_file() + the step-1 writer  minimap2/map.c:1376-1403, :1296-1304 + encode_ovl() lib/ovl.c:109-150
 *                                                           ndgpu_ovl_map()  (K1, K3 seeds, sort, K4 chain DP, K5 hits; .ovl bytes)
 *
 * Reads are passed exactly as they are stored in NextDenovo's `.2bit` files (lib/bseq.c:114-139): 16 bases per
 * uint32, first base in the two top bits, A=0 C=1 G=2 T=3; word_off[i] = index of read i's first word in `words`.
 * ids[i] is the numeric read name (the reference prints it with "%u" and compares names as strings,
 * map.c:136,1296; this library reproduces that ordering).
 *
 * All functions fail loudly (return NULL / negative and print to stderr) when no HIP device is usable; there
 * is no CPU fallback.
 */
#ifndef NDGPU_OVERLAP_H
#define NDGPU_OVERLAP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: these are its only exports */

typedef struct ndgpu_ovl_opt {
	int32_t k, w, hpc;                  /* mm_idxopt_t::k, ::w, ::flag & MM_I_HPC          (minimap.h) */
	int32_t no_diag, no_dual;           /* MM_F_NO_DIAG, MM_F_NO_DUAL                      (minimap.h:8-9) */
	int32_t min_cnt, min_chain_score;   /* mm_mapopt_t::min_cnt, ::min_chain_score */
	int32_t bw, max_gap;                /* ::bw, ::max_gap */
	int32_t max_chain_skip, max_chain_iter;
	int32_t minlen;                     /* --minlen (500 for --step 1) */
	int32_t seed;                       /* --seed (11) */
	int32_t dvt, maxhan1, maxhan2;      /* --dvt, --maxhan1, --maxhan2 */
	float   mid_occ_frac;               /* -f FLOAT (< 1) */
	int32_t mid_occ;                    /* -f INT (>= 1); 0 = derive from mid_occ_frac */
	int32_t mode;                       /* --mode (2); 3 = HiFi: chain ends trimmed (nd_fix_bad_ends, map.c:340-373) and every hit
	                                       extended into the unaligned read ends (nd_extend_ends, map.c:385-482) before the filter */
	float   d_factor;                   /* --df (0.1): weight of the extension's running score (x + y) * d_factor - d */
	int32_t step;                       /* --step: 1 (raw reads, 8-field records) or 2 (corrected reads, `cns_align`, 10-field records) */
	float   minide;                     /* --step 2: --minide (0.05) */
	int32_t minmatch;                   /* --step 2: --minmatch (100) */
	int32_t max_occ;                    /* -f FLOAT,INT (mm_mapopt_t::max_occ, main.c:343): when > the mid_occ of a map call, a query read
	                                       that chained nothing is seeded again with this threshold and chained again (map.c:553-575,
	                                       :678-700); 0 (every preset) = never */
} ndgpu_ovl_opt;

typedef struct ndgpu_ovl_index ndgpu_ovl_index;

/* one step-1 overlap before varint coding: the fields of `overlap` (lib/ovl.h:20-25) */
typedef struct ndgpu_ovl_rec {
	uint32_t rev, qname, qs, qe, tname, ts, te, match;
} ndgpu_ovl_rec;

/* mm_mapopt_init + mm_idxopt_init + mm_set_opt(preset) + `--step 1`; preset = "ava-ont" | "ava-pb" | "ava-hifi"
 * (options.c:84-111; ava-hifi = k 51, w 51, HPC: the two-word k-mer sketch mm_sketch_nextdenovo_longkmer, sketch.c:283-356).
 * Returns 0, or -1 for an unknown preset.  k may be 1..127 except 32, 64 and 96 (mm_sketch's own range: sketch.c:84, :286-293). */
int ndgpu_ovl_opt_preset(const char *preset, ndgpu_ovl_opt *opt);

/* Sketch the target reads on the device and build the minimizer index in HBM. */
ndgpu_ovl_index *ndgpu_ovl_index_create(const ndgpu_ovl_opt *opt, uint32_t n_reads, const uint32_t *words, uint64_t n_words,
                                        const uint64_t *word_off, const uint32_t *lens, const uint32_t *ids);
void ndgpu_ovl_index_destroy(ndgpu_ovl_index *idx);
/* occurrence threshold of the top `frac` repetitive minimizers (+1), as mm_idx_cal_max_occ */
int32_t ndgpu_ovl_index_mid_occ(ndgpu_ovl_index *idx, float frac);
/* n[0] = minimizers, n[1] = distinct minimizers, n[2] = target reads */
void ndgpu_ovl_index_stat(const ndgpu_ovl_index *idx, uint64_t n[3]);

/* Map the query reads against the index; returns the number of overlaps kept by the step-1 filter
 * (self hits dropped, query span >= minlen, optional --dvt) in reference output order, or < 0 on error.
 * *recs is malloc'd (release with ndgpu_ovl_free). */
int64_t ndgpu_ovl_map(ndgpu_ovl_index *idx, const ndgpu_ovl_opt *opt, int32_t mid_occ, uint32_t n_reads, const uint32_t *words,
                      uint64_t n_words, const uint64_t *word_off, const uint32_t *lens, const uint32_t *ids,
                      ndgpu_ovl_rec **recs);

/* encode_ovl(): append `n` records to a byte buffer as 8 big-endian base-128 varints each;
 * prev[2] = running (qname, tname) state (`prev_t pid`, minimap2/main.c:29).  out needs 40 bytes per record.
 * Returns bytes written. */
int64_t ndgpu_ovl_encode(const ndgpu_ovl_rec *recs, int64_t n, uint32_t prev[2], uint8_t *out);

/* decode_ovl() over a whole buffer (lib/ovl.c:152-203, the routine lib/nextcorrect.py:105 calls per record through ovlseq.so):
 * out[8 * k ..] = qname, rev, qs, qe, tname, ts, te, match of record k; prev[2] = running (qname, tname) state; a trailing
 * partial record is not consumed (*consumed = bytes used, may be NULL).  Returns the records decoded (<= cap). */
int64_t ndgpu_ovl_decode(const uint8_t *buf, uint64_t n_bytes, uint32_t prev[2], uint32_t *out, int64_t cap, uint64_t *consumed);
/* the walk of kbit_read() (lib/bseq.c:257-299) over a .2bit payload without its 2 magic bytes: ids, lengths and the word
 * index of every read's sequence.  Returns the number of reads in the payload (cap = 0: count only), -1 when a record's words
 * run past the payload (truncated / corrupt file). */
int64_t ndgpu_2bit_index(const uint32_t *words, uint64_t n_words, uint32_t *ids, uint32_t *lens, uint64_t *word_off, int64_t cap);

/* ---- read ingestion: the FASTA / FASTQ[.gz] reader of `seq_dump` (util/seq_dump.c:60-118 over kseq_read, util/kseq.h:178-222) ----
 * Streams the file (1 MB inflate window).  ndgpu_fastx_read appends the next records' sequences back to back into buf (cap bytes),
 * off[k] / len[k] = where record k starts / its length, at most max_recs records and never a partial one.  Returns the number of
 * records; 0 = the stream is over (end of file, or the point where kseq_read gives up: a quality string of another length than its
 * sequence); -3 = read error; -4 = the next record alone exceeds cap (ndgpu_fastx_pending = its length; call again with room). */
typedef struct ndgpu_fastx ndgpu_fastx;
ndgpu_fastx *ndgpu_fastx_open(const char *path);
int64_t ndgpu_fastx_read(ndgpu_fastx *h, uint8_t *buf, uint64_t cap, uint64_t *off, uint32_t *len, int64_t max_recs);
/* the same, also handing out strtoul(name) of every record (the numeric read names of the corrected-read files that
 * minimap2-nd --step 2 maps, minimap2/map.c:1298-1300); ids may be NULL */
int64_t ndgpu_fastx_read_named(ndgpu_fastx *h, uint8_t *buf, uint64_t cap, uint64_t *off, uint32_t *len, uint32_t *ids, int64_t max_recs);
uint64_t ndgpu_fastx_pending(const ndgpu_fastx *h);
void ndgpu_fastx_close(ndgpu_fastx *h);
/* The byte stream under ndgpu_fastx, by itself: gzread() (zlib.h; what KSEQ_INIT(gzFile, gzread) reads through, lib/bseq.h:3) with the
 * gzip file inflated by `threads` host threads (0 = NDGPU_INFLATE_THREADS, or the CPUs the process may use, at most 16) -- the same
 * bytes: members concatenated, trailing garbage ignored, a truncated file hands out what it holds, -1 for corrupt data.  With 1
 * thread, or for what is not a regular file starting with a gzip member, the reader IS zlib's.  ndgpu_gzin_stats: 1 and
 * {rounds, chunks decoded, chunks accepted} when the threaded reader is in use, else 0. */
typedef struct ndgpu_gzin ndgpu_gzin;
ndgpu_gzin *ndgpu_gzin_open(const char *path, int threads);
int64_t ndgpu_gzin_read(ndgpu_gzin *h, void *buf, uint32_t len);
int ndgpu_gzin_stats(const ndgpu_gzin *h, uint64_t out[3]);
void ndgpu_gzin_close(ndgpu_gzin *h);

void ndgpu_ovl_free(void *p);
/* Why an entry point of this library failed: 1 = out of device memory (release memory, e.g. ndgpu_ovl_trim() and the consensus
 * library's ndgpu_release_memory(), and call again), 2 = another allocation error, 0 = not an allocation failure.  Cleared by the call. */
int ndgpu_ovl_last_error(void);
/* The library keeps freed device blocks cached between calls (at most 1.25 x the most it ever had in use at once, and
 * NDGPU_OVL_POOL_GB if set); this releases them.
 * Returns the bytes released. */
uint64_t ndgpu_ovl_trim(void);
/* Declare a host buffer of 2-bit read words (the `words` array of the calls below, as stored in a .2bit file) resident: it is uploaded once,
 * and every later call whose `words` lie inside it works on the device copy instead of uploading them again (index build and query side of
 * every raw_align job name the same reads).  The buffer must stay valid and unchanged until ndgpu_ovl_words_release().  Additive: the
 * reference has no counterpart (its reads live in host memory, lib/bseq.c).  Returns 0, -1 on failure (out of device memory). */
int ndgpu_ovl_words_resident(const uint32_t *words, uint64_t n_words);
void ndgpu_ovl_words_release(const uint32_t *words);
/* out[0] = device bytes the library has in use now, out[1] = cached for reuse, out[2] = the most it ever had in use at once -- what a
 * caller that runs another memory-hungry stage on the same device between two calls should leave free (out[2] - out[1]). */
void ndgpu_ovl_pool_bytes(uint64_t out[3]);
/* out[0] = hipMalloc / hipFree calls the library's block pool has made, out[1] = nanoseconds they took (such a call stalls every stream
 * of the process, the consensus contexts' included); reset != 0 clears the counters */
void ndgpu_ovl_pool_calls(uint64_t out[2], int reset);

/* ---- overlap sort / filter: the `ovl_sort` program between the two stages (util/ovl_sort.c, raw reads, no -H) ----
 *
 *   files[f][0..n_per_file[f])  step-1 records of input file f in file order (what ndgpu_ovl_map returned, or a decoded .ovl:
 *                               half-open ends), files in input.fofn order
 *   seed_len[id]                length of read `id` if it is a seed of this seed file (its .idx, util/ovl_sort.c:106-131), else 0
 *   min_seed_len                shortest seed of the .idx;  max_bin_cov = -k (40);  max_flank_len = -l (300)
 *
 * Returns the records of `sorted.ovl` in file order (per seed: the self record, then the admitted overlaps, inclusive
 * ends; util/ovl_sort.c:675-741, 433-571, 876-925) and the `.bl` verdicts ('c' contained, 'k' chimeric) in seed order.
 * Equal (seed, match, span) keys keep input order.  When the candidates do not fit the device at once (~360 bytes per raw record all
 * told) the raw records -- which stay in the caller's arrays -- pass the device twice in pieces and the seeds are sorted and filtered
 * in consecutive id ranges (the counterpart of the reference's temporary files and merge under a small -m, util/ovl_sort.c:1079-1110:
 * like there, the result is the same); NDGPU_OVLSORT_PIECE_RECORDS / NDGPU_OVLSORT_RANGE_CANDIDATES force that form.
 * All three outputs are malloc'd (ndgpu_ovl_free). */
typedef struct ndgpu_ovl_sort_stats { double gpu_ms; uint64_t raw_records, candidates, seeds, kept, ranges; } ndgpu_ovl_sort_stats;  /* ranges: seed-id ranges the candidates were sorted in (1 = all at once) */
int64_t ndgpu_ovl_sort(const ndgpu_ovl_rec *const *files, const int64_t *n_per_file, int32_t n_files, const uint32_t *seed_len,
                       uint32_t n_ids, int32_t min_seed_len, int32_t max_bin_cov, int32_t max_flank_len, ndgpu_ovl_rec **out,
                       uint32_t **bl_id, uint8_t **bl_kind, int64_t *n_bl, ndgpu_ovl_sort_stats *stats);
/* the same with `-H` (high-quality reads, util/ovl_sort.c:27,1045): every candidate is collected, overlaps that start and end at hot
 * break points (del_repeat_alns, :389-431) are dropped, chimeras come from uncovered bins no overlap spans (check_chimer_hq,
 * :287-314), a containing overlap counts only with >= 90 % matches (:555) */
int64_t ndgpu_ovl_sort_hq(const ndgpu_ovl_rec *const *files, const int64_t *n_per_file, int32_t n_files, const uint32_t *seed_len,
                          uint32_t n_ids, int32_t min_seed_len, int32_t max_bin_cov, int32_t max_flank_len, ndgpu_ovl_rec **out,
                          uint32_t **bl_id, uint8_t **bl_kind, int64_t *n_bl, ndgpu_ovl_sort_stats *stats);

/* ---- pile admission between the sort and the consensus: read_seq_data of lib/nextcorrect.py:92-143 on the records of a sorted.ovl
 * (what ndgpu_ovl_sort returned).  One pile per seed whose length is >= min_len_seed and that is not in skip_ids (the `.bl` list /
 * seeds already corrected): overlaps shorter than min_len_aln, repeated query reads and overlaps beyond 1.5 x max_cov_aln of
 * cumulative depth are dropped, piles below min_cov_seed of depth are dropped.  recs8[i] = the admitted record i in the field order
 * nextCorrect's callers use (seed, rev, seed start, seed end (inclusive), read, read start, read end, match); pile p =
 * recs8[pile_off[p] .. pile_off[p+1]); seeds[p] = its seed id.  Host logic (no device work); outputs malloc'd (ndgpu_ovl_free).
 * Returns the number of admitted records. */
int64_t ndgpu_assemble_piles(const ndgpu_ovl_rec *sorted, int64_t n, uint32_t n_ids, uint32_t min_len_seed, uint32_t min_len_aln,
                             uint32_t max_cov_aln, uint32_t min_cov_seed, const uint32_t *skip_ids, int64_t n_skip, uint32_t **recs8,
                             uint64_t **pile_off, uint32_t **seeds, int64_t *n_piles);

/* ---- read ingestion: the 2-bit packing of `seq_dump` (seq2bit, lib/bseq.c:114-139; called from util/seq_dump.c:36-41) ----
 * Read i is the lens[i] ASCII bytes at ascii + ascii_off[i]; its ceil(lens[i]/16) words go to words + word_off[i] (word_off ascending,
 * reads back to back).  Bytes other than ACGTU (either case) are coded 4 and OR-ed in as the reference does, so they disturb the low
 * bit of the preceding base.  Returns the number of words written, < 0 on error. */
int64_t ndgpu_pack_2bit(uint32_t n_reads, const uint8_t *ascii, uint64_t n_bytes, const uint64_t *ascii_off, const uint32_t *lens,
                        const uint64_t *word_off, uint32_t *words);

/* ---- array-level views used by the parity tests (same library, same kernels) ---- */

/* K1 alone: minimizers of every read; rid_is_index != 0 puts the read index in the high word of y (index
 * side), 0 leaves it 0 (query side, map.c:71).  off[n_reads+1], x/y malloc'd. */
int64_t ndgpu_ovl_sketch(const ndgpu_ovl_opt *opt, uint32_t n_reads, const uint32_t *words, uint64_t n_words,
                         const uint64_t *word_off, const uint32_t *lens, int rid_is_index, uint64_t **x, uint64_t **y,
                         uint64_t *off);
/* index arrays: key[n_keys] ascending, start[n_keys+1], pos[n_min] */
void ndgpu_ovl_index_dump(const ndgpu_ovl_index *idx, uint64_t *key, uint64_t *start, uint64_t *pos);
/* after ndgpu_ovl_map(..): the last batch's sorted anchors and chain-DP arrays of query `q` (index into that
 * call's reads).  Returns the anchor count; arrays malloc'd. */
int64_t ndgpu_ovl_debug_anchors(ndgpu_ovl_index *idx, uint32_t q, uint64_t **ax, uint64_t **ay, int32_t **f, int32_t **p);

/* kernel timing of the calls so far on this index (ms, HIP events) and work counters */
typedef struct ndgpu_ovl_stats {
	double sketch_ms, index_sort_ms, seed_ms, sort_ms, exact_sort_ms, chain_ms, hits_ms;
	uint64_t bases_sketched, minimizers, anchors, tie_reads, chain_cells, chains, overlaps, map_calls, batches;
	uint64_t ext_problems, ext_launches; /* --mode 3: end extensions run (two candidates per hit), launches they took */
	double ext_ms;
	uint64_t rechained;                  /* -f FLOAT,INT: query reads seeded and chained a second time */
} ndgpu_ovl_stats;
/* ---- `minimap2-nd --step 2 --mode 0` (the cns_align command of nextDenovo:356-366 with the re-alignment switched off) ----
 * one --step 2 overlap before varint coding: the fields of `overlap_i` (lib/ovl.h); identity = matches * 10000 / block length */
typedef struct ndgpu_ovl_rec10 {
	uint32_t rev, qname, qs, qe, qlen, tname, ts, te, tlen, identity;
} ndgpu_ovl_rec10;
/* replaces: worker_for without re-alignment + the writer's record filter (minimap2/map.c:988-1031, 1305-1309) for the reads of one
 * query file against one index part: hits marked per target, filtered by length / identity / minimum block length.  opt->step must
 * be 2 and opt->mode 0.  *recs is malloc'd (ndgpu_ovl_free).  Returns the record count, < 0 on error. */
int64_t ndgpu_ovl_map2(ndgpu_ovl_index *idx, const ndgpu_ovl_opt *opt, int32_t mid_occ, uint32_t n_reads, const uint32_t *words,
                       uint64_t n_words, const uint64_t *word_off, const uint32_t *lens, const uint32_t *ids, ndgpu_ovl_rec10 **recs);
/* The hits of every query read with nothing judged or filtered, in hit order (mm_map_frag's, minimap2/map.c:506-623): rev, qs, qe,
 * ts, te as usual; `qname` = the hit's target -- its index-local read number, or, with a wanted list, its POSITION in the read's list
 * want[want_off[i] .. want_off[i + 1]) (the read is then mapped against those reads only, which are seen in list order: the
 * per-thread mini-index of the re-alignment, minimap2/index.c:434-575); `tname` = block length, `match` = match count.
 * nameless bit 0: the reads have no names (mm_map(..., qname = 0)): no name-based seed skipping, no self test; bit 1: the chaining
 * is mm_chain_dp_nextdenovo (mm_map_nextdenovo1, minimap2/map.c:1047: a mapping with more than 100,000 anchors is thinned first).  counts[i] = hits
 * of read i; *max_anchors (may be NULL) = the most anchors any one read had.  Both arrays malloc'd (ndgpu_ovl_free).  Returns the
 * number of hits, < 0 on error. */
int64_t ndgpu_ovl_map_regs(ndgpu_ovl_index *idx, const ndgpu_ovl_opt *opt, int32_t mid_occ, uint32_t n_reads, const uint32_t *words,
                           uint64_t n_words, const uint64_t *word_off, const uint32_t *lens, const uint32_t *ids, const uint64_t *want_off,
                           const uint32_t *want, int nameless, ndgpu_ovl_rec **recs, uint32_t **counts, uint64_t *max_anchors);
/* replaces: worker_for WITH the re-alignment (minimap2/map.c:988-1126) + the writer's record filter (:1305-1309): `--step 2` as
 * nextDenovo runs it (no --mode, i.e. --mode 2, options.c:56), or --mode 1 (the two forms of the re-alignment switch at 20 candidates
 * instead of 200, and the one-read-index mappings chain through mm_chain_dp_nextdenovo: a mapping with more than 100,000 anchors
 * loses the anchors of crowded target positions first, chain.c:185-226).  idx = the index part (the preset's k, w); q_mini / t_mini = indexes
 * with the short sketch (--kn 17 --wn 10, main.c:197) over the query reads / over the part's target reads; cn = --cn (20).
 * opt->step must be 2 and opt->mode 1 or 2.  *recs is malloc'd (ndgpu_ovl_free).  Returns the record count, < 0 on error. */
int64_t ndgpu_ovl_map2_realign(ndgpu_ovl_index *idx, ndgpu_ovl_index *q_mini, ndgpu_ovl_index *t_mini, const ndgpu_ovl_opt *opt,
                               int32_t mid_occ, int32_t cn, uint32_t n_t, const uint32_t *t_words, uint64_t t_n_words,
                               const uint64_t *t_word_off, const uint32_t *t_lens, const uint32_t *t_ids, uint32_t n_q,
                               const uint32_t *q_words, uint64_t q_n_words, const uint64_t *q_word_off, const uint32_t *q_lens,
                               const uint32_t *q_ids, ndgpu_ovl_rec10 **recs);
/* replaces: filter_ovl (lib/ovl.c:449-563), whose per-read state lives for the whole run (opt.os, main.c:272), encode_ovl_i
 * (lib/ovl.c:205-253) and out_bl (lib/ovl.c:339-362).  Host code: a record's verdict depends on every record before it. */
typedef struct ndgpu_s2_state ndgpu_s2_state;
ndgpu_s2_state *ndgpu_s2_new(void);
void ndgpu_s2_free(ndgpu_s2_state *st);
/* the records of one ndgpu_ovl_map2 call, in order: verdicts (kept[i], may be NULL) and the bytes of the kept ones (*out, malloc'd,
 * ndgpu_ovl_free); prev[2] is the encoder's delta state, carried from call to call (start: 0, 0).  The file starts with the two
 * bytes 00 FF (init_ovl_mode, lib/ovl.c:70-75), the caller's to write.  Returns the byte count, < 0 on error. */
int64_t ndgpu_s2_filter_encode(ndgpu_s2_state *st, const ndgpu_ovl_rec10 *recs, int64_t n, int32_t maxhan1, int32_t maxhan2,
                               uint32_t prev[2], uint8_t **out, uint8_t *kept);
/* the `.bl` text written when the run ends (*text malloc'd, ndgpu_ovl_free); the state is spent afterwards */
int64_t ndgpu_s2_bl(ndgpu_s2_state *st, char **text);

/* ---- the base-level extension kernel of minimap2's -c / -a path (off the default correction path: `minimap2-nd --step 1` runs
 *      without -c) ----
 * replaces: ksw_extz_t (minimap2/ksw2.h:23-32), same layout */
typedef struct ndgpu_ksw_extz {
	uint32_t max:31, zdropped:1;
	int max_q, max_t;      /* max extension coordinate */
	int mqe, mqe_t;        /* max score when reaching the end of query */
	int mte, mte_q;        /* max score when reaching the end of target */
	int score;             /* max score reaching both ends; may be KSW_NEG_INF (-0x40000000) */
	int m_cigar, n_cigar;
	int reach_end;
	uint32_t *cigar;
} ndgpu_ksw_extz;
/* replaces: ksw_extd2_sse (minimap2/ksw2.h:60-61, minimap2/ksw2_extd2_sse.c:26-399; caller mm_align_pair, minimap2/align.c:331):
 * global / extension alignment with the two-piece affine gap cost min(gapo + k * gape, gapo2 + k * gape2), band w, z-drop, the
 * KSW_EZ_* flags (score only, right-aligned gaps, generic matrix, approximate maximum / z-drop, extension only, reversed CIGAR).
 * Sequences are codes < m; mat is m x m.  Every field of *ez and the CIGAR are what the reference computes.  km is ignored;
 * ez->cigar is malloc'd (as the reference does with km == NULL) and replaces a buffer the caller passed in. */
void ksw_extd2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat, int8_t gapo,
                   int8_t gape, int8_t gapo2, int8_t gape2, int w, int zdrop, int end_bonus, int flag, ndgpu_ksw_extz *ez);
/* additive: a batch of such problems in one launch, one wavefront per problem (the gaps between the anchors of the chains of a
 * batch of reads, their end extensions).  res[i].cigar is malloc'd (free) when n_cigar > 0.  Returns 0, < 0 on error. */
typedef struct ndgpu_ksw_job {
	const uint8_t *query, *target;
	const int8_t *mat;
	int32_t qlen, tlen, w, zdrop, end_bonus, flag;
	int8_t m, gapo, gape, gapo2, gape2;
} ndgpu_ksw_job;
typedef struct ndgpu_ksw_result {
	int32_t max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, n_cigar, reach_end;
	uint32_t *cigar;
} ndgpu_ksw_result;
int ndgpu_ksw_extd2_batch(const ndgpu_ksw_job *jobs, int n, ndgpu_ksw_result *res);

/* replaces: ksw_ll_qinit(km, 2, qlen, query, 5, mat) + ksw_ll_i16 (minimap2/ksw2_ll_sse.c:32-156; callers mm_test_zdrop and
 * mm_align1_inv, minimap2/align.c:80-82,813-815): the striped local-alignment SCORE of a batch of problems, one wavefront each, with
 * the values of the SSE schedule (score, query end, target end -- also for equal maxima and the padding columns).  mat is 5 x 5,
 * codes <= 4.  Returns 0, < 0 on error. */
typedef struct ndgpu_ll_job { const uint8_t *query, *target; const int8_t *mat; int32_t qlen, tlen, gapo, gape; } ndgpu_ll_job;
typedef struct ndgpu_ll_result { int32_t score, qe, te; } ndgpu_ll_result;
int ndgpu_ksw_ll_batch(const ndgpu_ll_job *jobs, int n, ndgpu_ll_result *res);

/* ---- `minimap2-nd --step 1 -c`: base-level alignment through the chains (off the default correction path: nextDenovo runs
 *      --step 1 without -c) ----
 * the scoring side of mm_mapopt_t (minimap.h; defaults mm_mapopt_init, minimap2/options.c:36-43; -A -B -O -E -z -s of main.c) */
typedef struct ndgpu_ovl_aln_opt {
	int32_t a, b, q, e, q2, e2, sc_ambi;  /* match, mismatch, gap open / extend of the two pieces, score of a base against N */
	int32_t zdrop, zdrop_inv, end_bonus;
	int32_t min_dp_max;                   /* -s: 80 (min_chain_score * a when mm_mapopt_init runs; the ava presets do not touch it) */
	int32_t min_ksw_len;                  /* 200: anchors closer than this on either read are bridged by the next gap's alignment */
	int64_t max_sw_mat;                   /* --cap-sw-mem; 0 = none (the default: mm_mapopt_init leaves it 0): a problem of more cells counts as
	                                         z-dropped without being aligned (mm_align_pair, minimap2/align.c:323; the compiled reference segfaults on the first such gap of a
	                                         chain that has no CIGAR yet, so there is nothing to compare a cap with) */
	int32_t host_threads;                 /* threads of the bookkeeping between the batches; 0 = all */
} ndgpu_ovl_aln_opt;
void ndgpu_ovl_aln_opt_default(ndgpu_ovl_aln_opt *o, int32_t min_chain_score);
typedef struct ndgpu_ovl_cigar_stats {
	uint64_t chains, first_pass, second_pass, inversion_tests, inversions, cells, overlaps, inversions_aligned, splits;
	uint64_t chains_ns, ksw_ns, ksw_ll_ns, total_ns;
	/* wall time of the call: sketch / seeds / chaining on the device (ndgpu_ovl_map_chains); the ksw_extd2 batches (upload, kernel,
	 * backtrack, download); the ksw_ll batches; everything (the rest is the host's chain walking, CIGAR joining and filters) */
	/* chains aligned (pieces of split chains included); extension / gap problems of the first pass; gaps aligned again after a
	 * z-drop; local alignments of the inversion test; inversions looked at (mm_align1_inv calls); DP cells (query x target) of all
	 * problems; records out; inversions that were aligned; chains a z-drop split */
} ndgpu_ovl_cigar_stats;
/* array-level view (and the first half of ndgpu_ovl_map_cigar): the chains of every query read as mm_align_skeleton takes them
 * (minimap2/hit.c:52-85): chains[] in hit order, self hits included, = (strand, index-local target, offset of the first anchor in
 * the read's anchors, anchor count, chain score, hash, 0, 0); counts[i] = chains of read i; the chained anchors of read i =
 * ax / ay[a_off[i] .. a_off[i + 1]) in the order of the reference's a[].  Everything malloc'd (ndgpu_ovl_free). */
int64_t ndgpu_ovl_map_chains(ndgpu_ovl_index *idx, const ndgpu_ovl_opt *opt, int32_t mid_occ, uint32_t n_reads, const uint32_t *words,
                             uint64_t n_words, const uint64_t *word_off, const uint32_t *lens, const uint32_t *ids, ndgpu_ovl_rec **chains,
                             uint32_t **counts, uint64_t **ax, uint64_t **ay, uint64_t **a_off);
/* replaces: mm_map_file() + the step-1 writer WITH MM_F_CIGAR: mm_align_skeleton (minimap2/align.c:857-913: mm_align1 per chain --
 * left extension, gap filling with the approximate-then-exact z-drop passes, right extension --, z-drop splits, mm_align1_inv,
 * mm_filter_regs, mm_hit_sort) between chaining and the writer's filter (minimap2/map.c:484-503, 1297-1304).  The records carry the
 * alignments' coordinates and exact match counts.  t_words / t_word_off / t_lens / t_ids = the reads the index was built from (the
 * caller's arrays of ndgpu_ovl_index_create).  opt->step must be 1, opt->mode not 3, k <= 28 (the compiled reference aborts on
 * ava-hifi -c);
 * with q == q2 and e == e2 the reference takes ksw_extz2_sse, which equals the two-piece kernel with equal pieces under the flags this path passes).
 * Every alignment runs on the device (one wavefront per problem, csrc/ksw2_kernels.hip); the joining of CIGARs, the z-drop
 * bookkeeping and the seed filters are host logic.  *recs is malloc'd (ndgpu_ovl_free).  Returns the record count, < 0 on error. */
int64_t ndgpu_ovl_map_cigar(ndgpu_ovl_index *idx, const ndgpu_ovl_opt *opt, const ndgpu_ovl_aln_opt *aopt, int32_t mid_occ,
                            uint32_t n_reads, const uint32_t *words, uint64_t n_words, const uint64_t *word_off, const uint32_t *lens,
                            const uint32_t *ids, const uint32_t *t_words, const uint64_t *t_word_off, const uint32_t *t_lens,
                            const uint32_t *t_ids, ndgpu_ovl_rec **recs, ndgpu_ovl_cigar_stats *stats);

void ndgpu_ovl_get_stats(const ndgpu_ovl_index *idx, ndgpu_ovl_stats *st);
void ndgpu_ovl_reset_stats(ndgpu_ovl_index *idx);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
