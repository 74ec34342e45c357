/* ndgpu_nextcorrect.h -- C ABI of the MI355X-native read-correction engine.
 *
 * Drop-in boundary: the symbols below are exactly what NextDenovo's correction
 * stage binds through ctypes (reference lib/nextcorrect.py:56-59) and what
 * minimap2-nd / ctg_cns link against (reference lib/align.h:45-62).  Build
 * product: nextdenovo_amd/libndgpu_nextcorrect.so (install it as lib/nextcorrect.so
 * next to lib/nextcorrect.py).
 *
 * All compute behind these entry points runs in hand-written HIP kernels on
 * gfx950; there is no CPU fallback: the first call aborts with a message if no
 * HIP device is visible.
 */
#ifndef NDGPU_NEXTCORRECT_H
#define NDGPU_NEXTCORRECT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: these are its only exports */

/* replaces: reference lib/nextcorrect.h:70-74 (`consensus_trimed`), mirrored by
 * lib/nextcorrect.py:19-24.  `seq` is malloc'd, NUL-terminated, mixed case
 * (lower case = low confidence).  len 2 = uncorrectable, 3 = out of memory,
 * 4 = everything clipped (lib/nextcorrect.c:2254-2261, 1999, 2126). */
typedef struct {
    unsigned int len;
    float identity;
    char *seq;
} consensus_trimed;

/* replaces: reference lib/nextcorrect.h:161-162 / lib/nextcorrect.c:2219-2305.
 * seqs[0] = full seed, seqs[i>0] = strand-corrected overlapping substrings (ASCII,
 * upper-case ACGT, NUL-terminated); aln_start/aln_end = inclusive seed coordinates.
 * read_type: 1 ont, 2 clr, 3 hifi. */
consensus_trimed *nextCorrect(char **seqs, unsigned int *aln_start, unsigned int *aln_end, unsigned int seq_count,
                              unsigned int max_mem_len, unsigned int min_len_aln, unsigned int max_cov_aln,
                              unsigned int min_cov, unsigned int lqseq_max_length, float min_error_corrected_ratio,
                              unsigned int split, unsigned int fast, int read_type);

/* replaces: reference lib/nextcorrect.h:164 / lib/nextcorrect.c:2307-2310 */
void free_consensus_trimed(consensus_trimed *c);

/* ---- lib/align.h family (short `alignment` layout, i.e. without -DLGS_CORRECT,
 *      reference lib/align.h:21-32 / lib/nextcorrect.h:120-129) ---- */
typedef struct {
    unsigned int shift;
    unsigned int aln_len;
    unsigned int aln_t_s;
    unsigned int aln_t_e;
    unsigned int aln_t_len;
    unsigned int aln_q_len;
    char *q_aln_str;
    char *t_aln_str;
} alignment;

/* replaces: reference lib/align.h:61-62 / lib/align.c:572-578.  V and D are accepted
 * for signature compatibility and ignored (the device keeps its own state). */
void align(char *query_seq, int q_len, char *target_seq, int t_len, alignment *align_rtn, int *V, uint8_t **D);
/* replaces: reference lib/align.h:59-60 / lib/align.c:563-570 */
void align_hq(char *query_seq, int q_len, char *target_seq, int t_len, alignment *align_rtn, int *V, uint8_t **D);
/* replaces: reference lib/nextcorrect.h:157 / lib/align.c:580-679: full-matrix global alignment (match 2, mismatch -4, gap
 * open -4, gap extend -2).  s1 indexes the columns and is written to q_aln_str, s2 to t_aln_str; only aln_len and the two
 * strings (caller-allocated, s1_l + s2_l + 1 bytes each) are set.  Host routine: the reference calls it from nowhere. */
void align_nd(const char *s1, const uint32_t s1_l, const char *s2, const uint32_t s2_l, alignment *aln);
/* replaces: reference lib/align.h:45-50 / lib/align.c:22-78 (host-only helpers) */
void malloc_vd(int **V, uint8_t ***D, uint64_t max_mem_d);
void clean_V(int *V, int max_mem_d);
void destory_vd(int *V, uint8_t **D);
void revcomp_bseq(char *str, int len);
void reverse_str(char *str, int len);
void str_tolower(char *p);
void str_toupper(char *p);
/* ---- the prefix / extension members of the O(ND) family (reference lib/align.h:36-58, lib/align.c:80-426; callers
 *      minimap2/map.c:385-482, 941-956 and lib/ctg_cns.c).  GPU-backed (csrc/ext_kernels.hip); V and D are ignored. ---- */
typedef struct {
    unsigned int aln_len;  /* alignment columns */
    unsigned int aln_mlen; /* matching columns */
    unsigned int aln_t_s;
    unsigned int aln_t_e;  /* exclusive */
    unsigned int aln_q_s;
    unsigned int aln_q_e;  /* exclusive */
} alignpos;
/* replaces lib/align.c:80-141: *mlen / *blen are written when either sequence is exhausted within max_d steps and the band cap */
void ide(const char *query_seq, int q_len, const char *target_seq, int t_len, int *V, uint8_t **D, int max_d, int band_size,
         int *mlen, int *blen);
/* replaces lib/align.c:146-253: *aln is written under the same condition */
void alnpos(const char *query_seq, int q_len, const char *target_seq, int t_len, int *V, uint8_t **D, int max_d, int band_size,
            alignpos *aln);
/* replace lib/align.c:256-340 / :343-426: (*bstx, *bsty) = query / target bases covered at the peak of (x + y) * d_factor - d */
void extend_fwd(const char *query_seq, int q_len, const char *target_seq, int t_len, int *V, uint8_t **D, int max_d, int band_size,
                float d_factor, int *bstx, int *bsty);
void extend_rev(const char *query_seq, int q_len, const char *target_seq, int t_len, int *V, uint8_t **D, int max_d, int band_size,
                float d_factor, int *bstx, int *bsty);
/* additive: a batch of such problems in one launch (one lane per problem) -- what nd_extend_ends' loop over the overlaps of a read
 * (minimap2/map.c:385-482) would hand over.  kind: 0 ide, 1 alnpos, 2 extend_fwd, 3 extend_rev.  result: done = an end condition
 * fired; (a, b) = (mlen, blen) or (bstx, bsty); pos[6] = the fields of `alignpos` (alnpos only).  Returns 0, < 0 on error. */
typedef struct ndgpu_ext_job {
    const char *q;
    int32_t q_len;
    const char *t;
    int32_t t_len;
    int32_t max_d, band_size;
    float d_factor;
    int32_t kind;
} ndgpu_ext_job;
typedef struct ndgpu_ext_result {
    int32_t done, a, b;
    uint32_t pos[6];
} ndgpu_ext_result;
int ndgpu_ext_batch(const ndgpu_ext_job *jobs, int n, ndgpu_ext_result *res);

/* replaces: reference lib/nextcorrect.h:166 / lib/dag.c:658-694.  `seqs` points at
 * `seq_count` records laid out as the reference's `struct seq_`
 * (lib/nextcorrect.h:63-68: u16 order, u16 kscore, u16 len, char seq[10000]). */
char *poa_to_consensus(const void *seqs, const int seq_count);

/* ---- additive entry points (not in the reference) ---- */

/* Correct `n_piles` seeds in one call so that thousands of alignments share each
 * kernel launch.  Per-pile arguments are arrays of length n_piles whose elements have
 * the meaning of the corresponding nextCorrect() argument; scalar options apply to all
 * piles.  out[i] receives a malloc'd consensus_trimed (free with
 * free_consensus_trimed).  host_threads <= 0 selects hardware concurrency.
 * Returns 0. */
int ndgpu_correct_batch(int n_piles, char ***seqs, unsigned int **aln_start, unsigned int **aln_end,
                        const unsigned int *seq_count, const unsigned int *max_mem_len,
                        const unsigned int *lqseq_max_length, unsigned int min_len_aln, unsigned int max_cov_aln,
                        unsigned int min_cov, float min_error_corrected_ratio, unsigned int split, unsigned int fast,
                        int read_type, int host_threads, consensus_trimed **out);

/* The output loop of lib/nextcorrect.py:236-260 (without -s) over finished records: for every ids[k] in order, a record with
 * len >= min_len_seed, len > 4 and identity >= min_ratio is written to fd_out as ">NAME LEN IDENTITY\nBASES\n" (IDENTITY as Python's
 * '%f') and, if fd_idx >= 0, "NAME\tOFFSET\tLEN\n" to fd_idx (OFFSET = where the bases start in the output file); any other record
 * but an out-of-memory seed (len 3) gets "NAME\t0\t0\n" in the index.  names[i] = the number the reference prints for pile i;
 * *pos = the output file's size before the call, updated.  lens / identities (may be NULL) receive every handed-over record's
 * values at [ids[k]]; the records are freed (free_consensus_trimed) and their slots set to NULL.  What a caller's completion
 * callback (ndgpu_piles_done_fn) does with its sub-batch.  Returns 0, or -1 when a write fails. */
int ndgpu_write_records(consensus_trimed **recs, const uint32_t *ids, int n, const uint32_t *names, uint32_t min_len_seed,
                        double min_ratio, int fd_out, int fd_idx, uint64_t *pos, uint32_t *lens, float *identities);

/* Resident read database: the additive replacement for ovlseq.so's init_ovls + getseq
 * (reference lib/ovlseq.c:39-138, lib/nextcorrect.py:62-69,193).  `words` is the .2bit
 * payload layout of lib/bseq.c:114-139 (16 bases per uint32, first base in the top two
 * bits); read i starts at words[word_off[i]] and has len[i] bases.  The DB is uploaded to
 * HBM once (forward + reverse complement) and stays there. */
typedef struct ndgpu_db ndgpu_db;
/* NULL when the DB does not fit the device memory */
ndgpu_db *ndgpu_db_create(uint32_t n_reads, const uint32_t *words, const uint64_t *word_off, const uint32_t *len);
/* The same DB from the files the stage already has: `idx_fofn` as given to `nextcorrect.py -f` (one `.NAME.idx` path per line;
 * the `.2bit` next to each is read, as init_ovls() does: lib/ovlseq.c:24-37,50-138).  NULL on I/O / format errors. */
ndgpu_db *ndgpu_db_open(const char *idx_fofn);
void ndgpu_db_destroy(ndgpu_db *db);

/* Correct piles given as overlap records against a resident DB: the batched
 * equivalent of lib/nextcorrect.py:183-199 (worker) + nextCorrect().  `recs` holds
 * 8 uint32 per record in decode_ovl order (lib/ovl.c:189-200, lib/nextcorrect.py:106:
 * seed, rev, t_s, t_e, query read, q_s, q_e, match; coordinates inclusive); pile i owns
 * records [pile_off[i], pile_off[i+1]) and its first record is the seed self record.
 * max_aln_length and the per-pile max_lq_length = min(seed_len/2, max_lq_length) are
 * derived exactly as lib/nextcorrect.py:117,135-137,188 does.  Returns 0; -1 for a dead handle; -2 (nothing computed) when
 * a record names a read that is not in `db` or a window outside its read.  Sub-batches that do not fit the device memory
 * are halved; a single pile that still does not fit comes back as the reference's out-of-memory seed (len 3,
 * lib/nextcorrect.c:2254-2261).  Several handles may be alive; every call works against the one it is given. */
int ndgpu_correct_piles(ndgpu_db *db, int n_piles, const uint32_t *recs, const uint64_t *pile_off,
                        unsigned int min_len_aln, unsigned int max_cov_aln, unsigned int min_cov,
                        unsigned int max_lq_length, float min_error_corrected_ratio, unsigned int split,
                        unsigned int fast, int read_type, int host_threads, consensus_trimed **out);

/* The same call, handing the records over as they become ready: `done(user, pile_ids, n)` is called once per finished sub-batch
 * (n piles whose out[pile_ids[k]] are final), from the host thread that drove that sub-batch, one call at a time (the library
 * serialises them).  The reference streams its records the same way -- the parent of lib/nextcorrect.py:232-260 prints each seed
 * as its worker returns it, in no particular order -- so a caller can write cns.fasta while the later sub-batches are still on the
 * device.  Every pile is reported exactly once before the call returns; done == NULL makes this ndgpu_correct_piles. */
typedef void (*ndgpu_piles_done_fn)(void *user, const uint32_t *pile_ids, int n);
int ndgpu_correct_piles_stream(ndgpu_db *db, int n_piles, const uint32_t *recs, const uint64_t *pile_off,
                               unsigned int min_len_aln, unsigned int max_cov_aln, unsigned int min_cov,
                               unsigned int max_lq_length, float min_error_corrected_ratio, unsigned int split,
                               unsigned int fast, int read_type, int host_threads, consensus_trimed **out,
                               ndgpu_piles_done_fn done, void *user);

/* Counters accumulated by this process's device runtime since the last reset. */
typedef struct {
    uint64_t tasks, wide_tasks, cells, d_steps, trace_bits, columns, pool_bases, seq_bases;
    uint32_t max_band, forward_launches;
    double forward_ms;      /* HIP-event time of K7 (O(ND) forward) launches */
    double traceback_ms, tags_ms, links_ms, score_ms, extract_ms;
    uint64_t piles, tags, cells_msa, path_items, links, score_launches;
    double backtrack_ms;
    uint64_t score_segments;   /* scoring-DP segments (K10) */
    uint64_t score_repairs;    /* of which scored again after a failed boundary check */
    uint64_t score_slow_piles; /* piles scored by the int64 HBM-resident kernel */
    uint64_t trace_words;      /* 64-bit words of move-bit records K7 wrote (8 bytes per edit step up to 56 cells, 16 beyond) */
    uint64_t lq_rounds;        /* low-quality-region rounds (pile x round) handed to K12 */
    uint64_t lq_declined;      /* of which the kernel declined: the host path took them */
    double lq_ms;              /* HIP-event time of K12 */
    uint64_t allocs;           /* device / pinned buffers (re)allocated while batches were running, since the last reset ... */
    double alloc_ms;           /* ... and the wall time those calls took: an allocation in the middle of a step stalls every context,
                                  so steady-state steps should show none */
    uint64_t level_allocs;     /* the same between batches (buffers brought up to the largest sub-batch seen, nothing in flight) */
    double level_ms;
    uint64_t traceback_launches; /* K8a launches */
    uint64_t lq_launches;      /* K12 launches, and what they moved algorithmically: */
    uint64_t lq_columns;       /* columns of the linked pseudo-seeds walked */
    uint64_t lq_aln_columns;   /* alignment columns (2-bit kinds) read */
    uint64_t lq_bases;         /* candidate bases (2-bit) read */
    uint64_t lq_out;           /* consensus characters written */
    uint64_t lq_jobs;          /* K12 jobs (runs of regions scored from a speculative start) */
    uint64_t lq_repairs;       /* of which scored again after a failed boundary check */
    uint64_t tb_tasks;         /* alignments whose traceback ran in segments (one lane per 2^k edit steps) */
    uint64_t tb_walkers;       /* segments they were cut into */
    uint64_t tb_fallbacks;     /* of the alignments: walked again in one piece (a segment boundary did not agree) */
} ndgpu_stats;
void ndgpu_get_stats(ndgpu_stats *out);
void ndgpu_reset_stats(void);
/* The device contexts keep their (grow-only) buffers between calls.  This hands them back -- for a caller that alternates
 * the consensus with another memory-hungry stage on the same device (the overlap stage of a genome-scale read set) and
 * finds that stage short of memory.  Safe to call at any time from any thread: a context that is in the middle of a batch
 * (another thread inside nextCorrect / ndgpu_correct_*) keeps its buffers and is skipped.  Returns the bytes released; buffers
 * are re-created on demand. */
uint64_t ndgpu_release_memory(void);
/* Device bytes the consensus contexts must leave free when they size their buffers (the memory plan of every batch call subtracts
 * them from what is free): what another stage on the same device -- the overlap stage of the next seed file -- will need again. */
void ndgpu_reserve_device_memory(uint64_t bytes);
/* Number of HIP devices visible (0 if none); does not create a context. */
int ndgpu_device_count(void);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
