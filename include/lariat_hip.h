/* lariat_hip.h — C-ABI of liblariat_hip.so, the MI355X-native replacement for the
 * per-barcode align loop of 10XGenomics/lariat.
 *
 * What it replaces (reference file:line):
 *   go/src/gobwa/bwa_bridge.h:35-39      the cgo prototypes into BWA (mem_align1_core, mem_chain,
 *                                        mem_reg2aln, mem_matesw, bns_fetch_seq)
 *   go/src/gobwa/gobwa.go:128-153        GoBwaLoadReference / GoBwaAllocSettings
 *   go/src/gobwa/gobwa.go:226-337        GoBwaMemMateSW   (SE align both mates + mate rescue)
 *   go/src/gobwa/gobwa.go:400-415        GoBwaSmithWaterman (reg -> CIGAR/NM)
 *   go/src/gobwa/gobwa.go:50-80          GoBwaReference.GetSeq
 *   go/src/inference/lariat.go:461-547   DoRFAForOneBarcode (GetChains, GetAlignments, tagBestAlignments,
 *                                        inferMolecules, RFA optimizer, estimateMapQualities, markDuplicates,
 *                                        CheckSplitReads) — everything up to, not including, DumpToBams
 *
 * Design differences from the reference boundary:
 *   - batched: ONE call per batch of barcodes instead of thousands of cgo calls per barcode;
 *   - plain fixed-width arrays (SoA), no struct aliasing across the boundary, no pointers retained;
 *   - int status codes + lh_last_error(); the library never abort()s;
 *   - results live in a library-owned arena released by lh_result_free.
 *
 * Threading: an lh_index is immutable after load; one lh_context per host thread / HIP stream.
 */
#ifndef LARIAT_HIP_H
#define LARIAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LH_ABI_VERSION 5

/* status codes */
#define LH_OK 0
#define LH_E_ARG 1       /* bad argument */
#define LH_E_IO 2        /* index files unreadable / inconsistent (reference: gobwa.go:132-135 only logs) */
#define LH_E_HIP 3       /* HIP runtime error (message in lh_last_error) */
#define LH_E_CAPACITY 4  /* a per-BATCH workspace pool overflowed: split the batch and retry */
#define LH_E_NODEVICE 5  /* no HIP device / extension not usable: there is NO CPU fallback */
#define LH_E_LIMIT 6     /* input outside documented limits (read length > LH_MAX_READ_LEN, a per-READ / per-candidate slot limit ...):
                          * splitting the batch does not help; lh_last_error names the first offending read */

#define LH_MAX_READ_LEN 250 /* bases per read after trimming (u8 mate-rescue path needs l*a < 250, as upstream) */

typedef struct lh_index lh_index;     /* FM-index + 2-bit reference resident in HBM */
typedef struct lh_context lh_context; /* per-stream workspace */

/* mem_opt_t defaults (mem_opt_init, gobwa.go:149-153) + the lariat knobs of the hot path
 * (main.go:10 improper_pair_penalty; lariat.go:475-476 score_delta 25 / aln delta 17; lariat.go:885 genome length). */
typedef struct lh_opts {
    int32_t abi_version;
    int32_t a, b, o_del, e_del, o_ins, e_ins;
    int32_t pen_unpaired, pen_clip5, pen_clip3;
    int32_t w, zdrop, T;
    int32_t min_seed_len, min_chain_weight, max_chain_extend;
    float split_factor;
    int32_t split_width, max_occ, max_chain_gap, max_ins;
    float mask_level, drop_ratio, XA_drop_ratio, mask_level_redun;
    float mapQ_coef_len;
    int32_t max_mem_intv, max_matesw;
    /* mate-rescue window: gobwa.go:229-237 (orientation FR only) */
    int32_t pes_low, pes_high;
    /* lariat */
    int32_t rescue_score_delta; /* 25, lariat.go:475 */
    int32_t rescue_max_hits;    /* 50, gobwa.go:287,311 */
    int32_t aln_score_delta;    /* 17, lariat.go:476 */
    double improper_pair_penalty; /* -4.0, main.go:10 */
    double genome_length;         /* 3.2e9, lariat.go:885 */
    int32_t run_inference;        /* 0: stop after candidate generation (GetChains+GetAlignments) */
    uint32_t flags;               /* LH_F_*: A/B switches of the kernel sequence; results are identical under every combination */
} lh_opts;

/* lh_opts.flags (development / measurement; the defaults are the measured best) */
#define LH_F_NO_SWEEP_FILTER 1u  /* K1 sweeps every interval like bwt_smem1a does (n_ext then counts every bwt_extend of the reference) */
#define LH_F_EXT_WAVE 16u        /* K4 wave-per-read only */
#define LH_F_EXT_SERIAL 32u      /* K4's rounds and wave-kernel launches one after the other on one stream (per-round timings) */
#define LH_F_P2_TASKS 128u       /* K1 pass 2: a read's re-seeding calls shared by up to four lanes whatever the previous batch looked like (default: only after a repeat-rich batch) */
#define LH_F_RESCUE_FULL 256u     /* K6: every mate-rescue Smith-Waterman runs all rows of its window (default: the rows k_resc_cert proves sufficient, k_rescue3.h) */
#define LH_F_CHAIN_WAVE 64u      /* K3: the reads a lane does not chain all go to the wave-per-seed kernel (k_chain), none to the cluster kernel (k_chain_cl) */

/* how an index is made resident (lh_index_load / lh_index_from_arrays / lh_index_build_device); NULL = defaults */
typedef struct lh_index_opts {
    int32_t abi_version;
    int32_t sa_intv;          /* resident suffix-array sampling interval (power of two); 0 = the densest that fits half the free HBM (<= 128 GiB) */
    int32_t sb_shift;         /* occurrence-table super-block = 2^sb_shift symbols, 7..31; 0 = 31 */
    int32_t no_kmer_table;    /* leave out K1's 12-mer table */
    int32_t no_unique_runs;   /* leave out the inverse suffix array and the 4-bit text (and with them the sweep filters) */
    int32_t no_sweep_filter;  /* leave out the Bloom filters of K1's sweep filter */
    int32_t build_chunk_log2; /* lh_index_build_device: at most 2^this suffixes per sort chunk; 0 = 29 */
    int32_t ktree_levels;     /* K1's table of the bi-intervals of all strings of up to this many bases (<= 14; needs the dense SA); 0 = by the text's size, -1 = none */
} lh_index_opts;
void lh_index_opts_init(lh_index_opts* io);

/* per-context launch geometry (NULL = defaults; 0 in a field = its default) */
typedef struct lh_context_opts {
    int32_t abi_version;
    int32_t smem_grid;    /* K1 persistent waves (6144) */
    int32_t aln_grid;     /* K7 waves (10240) */
    int32_t rfa_grid;     /* K8 waves (4096) */
    int32_t rfa_slab_kb;  /* K8 first-pass slab per wave in KiB (3072); tests force the second pass with a small value */
    int32_t lanes;        /* 1 (default) .. 4: with L > 1 the context owns L - 1 further, smaller pipelines; every batch is cut at barcode
                           * boundaries and the L parts (barcodes are independent) are aligned side by side from L host threads, so
                           * that one part's kernels fill the idle tails of the others'; results are merged.  No stage dumps then. */
    int32_t big_slots;    /* K1: slots of the slab for reads with more than 64 SMEM intervals (max_pairs / 128, at least 64); the slab grows when a
                           * batch needs more, tests force that with a small value */
    int32_t rfa_tier_kb[2];   /* (ABI 5) K8: slab sizes in KiB of the two tiers between the regular slabs and the few large ones (16384, 131072); -1 = no such tier.
                               * The slabs are allocated when a batch first lists a barcode for the tier, as many as the list is long (at most rfa_tier_grid;
                               * together at most 32 KiB per read of the context's capacity); tests force the tiers with small values */
    int32_t rfa_tier_grid[2]; /* at most this many slabs (= waves) per tier (1024, 64) */
    int32_t reserved;
} lh_context_opts;
void lh_context_opts_init(lh_context_opts* co);

/* one batch of barcodes.  Reads are post-trim (reader.go trims read1) nt4 bytes: A0 C1 G2 T3 other 4
 * (SequenceConvert, gobwa.go:159).  read index r = 2*pair + mate (lariat.go:1720-1721,1758-1759). */
typedef struct lh_batch {
    int32_t n_barcodes;
    int32_t n_pairs;
    const int32_t* bc_pair_off; /* [n_barcodes+1] pair range of each barcode (one WorkUnit, lariat.go:211) */
    const uint8_t* bc_do_rfa;   /* [n_barcodes] worthRunningRFA (lariat.go:1088): >=5 pairs, barcode has '-', complete barcode */
    const int64_t* seq_off;     /* [2*n_pairs+1] byte offsets into seq */
    const uint8_t* seq;         /* nt4 bases */
    const uint64_t* name_seed;  /* [n_pairs] LE u64 of md5(read name)[0:8] (lariat.go:1483-1484) */
    /* centromeres (lariat.go:392-420), per contig, -1/-1 if none */
    const int64_t* cen_start;   /* [n_contigs] or NULL */
    const int64_t* cen_end;
} lh_batch;

/* Result arena.  Candidate order per read = BWA reg order after rescue (gobwa.go:330-336) = lariat's `full`
 * list (lariat.go:1697); a read without hits has ONE placeholder candidate (rid -1, pos -1; lariat.go:1737,1773).
 * All per-candidate arrays have n_cand entries, per-read arrays n_reads entries. */
typedef struct lh_result {
    int32_t abi_version;
    int32_t n_reads;
    int64_t n_cand;
    const int64_t* cand_off;  /* [n_reads+1] */
    /* --- candidate generation (Alignment fields set at lariat.go:1655-1696) --- */
    const int32_t* rid;       /* contig id, -1 for placeholder */
    const int64_t* pos;       /* Alignment.pos  (lariat.go:1645-1650) */
    const int64_t* aend;      /* Alignment.aend */
    const int64_t* rb;        /* mem_alnreg_t.rb/re in fwd||rev coordinates (-1 placeholder) */
    const int64_t* re;
    const uint8_t* reversed;
    const int32_t* score;     /* mem_alnreg_t.score */
    const int32_t* qb;        /* readmap_s */
    const int32_t* qe;        /* readmap_e */
    const int32_t* nm;        /* EditDistance (gobwa.go:482) */
    const int32_t* matches;
    const int32_t* mismatches;
    const int32_t* indels;
    const int32_t* soft_clipped;
    const int32_t* soft_clipped_length;
    const uint8_t* in_filtered;   /* member of `alignments` (score >= best-17, lariat.go:1698) */
    const int64_t* cigar_off;     /* [n_cand+1] */
    const uint32_t* cigar;        /* BAM encoding len<<4|op, MIDSH=01234 (forward-strand order, as mem_reg2aln) */
    const int64_t* mm_off;        /* [n_cand+1] */
    const int32_t* mm_ref_loc;    /* mismatchLocs (lariat.go:1609-1611) */
    const int32_t* mm_read_loc;   /* mismatchReadLocs */
    const double* log_alignment_probability;
    /* --- inference (per candidate) --- */
    const uint8_t* active;
    const uint8_t* is_proper;
    const uint8_t* bwa_pick;
    const uint8_t* active_molecule;
    const uint8_t* duplicate;
    const int32_t* molecule_id;
    const int32_t* mapq;
    const double* molecule_difference;
    const double* molecule_confidence;
    const double* sum_move_probability_change;
    const int64_t* mate_idx;      /* Alignment.mate_alignment as candidate index, -1 if nil */
    /* --- per read --- */
    const int64_t* active_idx;        /* the active candidate */
    const int64_t* second_best_idx;   /* mapq_data.second_best, -1 if nil */
    const double* second_best_score;  /* XS */
    const double* as_score;           /* AS: mapq_data.score */
    const int64_t* split_idx;         /* Alignment.secondary (split.go:142-158), -1 if none */
    const int32_t* split_mapq;
    const double* split_second_best;
    const double* split_score;
    /* --- telemetry: device-side counters of the algorithmic work (roofline accounting) --- */
    uint64_t n_ext, n_lf, n_sa, win_bases, n_chain_ext, ext_cells, glob_cells, n_rescue, rescue_cells;
    /* bwt_extend calls K1 really executed on the occurrence table in passes 1, 2, 3 of mem_collect_intv (n_ext counts the reference's
     * calls it performed OR accounted for: unique runs, the 12-mer jump; with the sweep filter on, intervals it left out are in neither) */
    uint64_t n_ext_exec_p1, n_ext_exec_p2, n_ext_exec_p3;
    /* bwt_extend results K1 read from its k-mer tree table instead (one 16-B entry each), per pass */
    uint64_t n_ktree_p1, n_ktree_p2, n_ktree_p3;
    /* bwt_smem1a calls of pass 1 that K1 decided from the text at the read's known locus (two PLCP bytes + the comparison; their bwt_extend calls are in neither count) */
    uint64_t n_calls_by_text;
    /* (ABI 5) the ksw_u8 cells K6 really executed: rescue_cells counts the cells the REFERENCE evaluates for the same attempts (tlen x striped width per pass) */
    uint64_t rescue_cells_exec;
    /* (ABI 5) K7, region -> CIGAR: the candidates whose alignment is not settled by bounds on the first look (k_aln_flat: equal spans, no DP needed by BWA's own rule
     * or at most four mismatches and no shifted diagonal in reach), and those still unsettled after the second (k_aln_flat2: five or six mismatches) — the ones a
     * banded global DP ran for.  The reference runs ksw_global2 for every candidate but bwa_gen_cigar2's "no gap" case: the oracle reports that count in both */
    uint64_t n_glob_listed, n_glob_exec;
    void* arena_; /* private */
} lh_result;

/* stage dumps for parity tests (one array set per stage; freed with lh_result_free-like lh_dump_free) */
typedef struct lh_stage_dump {
    int32_t n_reads;
    /* SMEM intervals after the three passes + sort (mem_collect_intv) */
    const int64_t* intv_off; /* [n_reads+1] */
    const uint64_t* intv;    /* 4 u64 per interval: x0 x1 x2 info */
    /* seeds in mem_chain order: rbeg, qbeg, len, rid(-1/-2 dropped) */
    const int64_t* seed_off;
    const int64_t* seed_rbeg;
    const int32_t* seed_qbeg;
    const int32_t* seed_len;
    const int32_t* seed_rid;
    /* chains after mem_chain_flt: per chain n_seeds, rid, w, kept, first seed rbeg */
    const int64_t* chain_off;
    const int32_t* chain_nseeds;
    const int32_t* chain_rid;
    const int32_t* chain_w;
    const int32_t* chain_kept;
    const int64_t* chain_pos;
    /* regs after mem_align1_core (before rescue): 12 ints per reg */
    const int64_t* reg_off;
    const int64_t* reg_rb;
    const int64_t* reg_re;
    const int32_t* reg_qb;
    const int32_t* reg_qe;
    const int32_t* reg_rid;
    const int32_t* reg_score;
    const int32_t* reg_truesc;
    const int32_t* reg_w;
    const int32_t* reg_seedcov;
    const int32_t* reg_seedlen0;
    const int32_t* reg_csub;
    const int32_t* reg_secondary;
    void* arena_;
} lh_stage_dump;

const char* lh_last_error(void);
int lh_device_count(void);

void lh_opts_init(lh_opts* o); /* replaces mem_opt_init + lariat flag defaults */

/* replaces bwa_idx_load(path, BWA_IDX_ALL) (gobwa.go:130): reads <prefix>.bwt/.sa/.pac/.ann/.amb and uploads to `device` */
int lh_index_load(const char* prefix, int device, const lh_index_opts* io, lh_index** out);
/* same, from in-memory images laid out exactly like the BWA files (used by the index builder / synthetic genomes) */
int lh_index_from_arrays(int device, uint64_t primary, const uint64_t L2[5], const uint32_t* bwt, uint64_t bwt_words,
                         int32_t sa_intv, const uint64_t* sa, uint64_t n_sa, const uint8_t* pac, int64_t l_pac,
                         int32_t n_contigs, const int64_t* contig_off, const int32_t* contig_len, const char* const* contig_name,
                         const lh_index_opts* io, lh_index** out);
/* GetReferenceContigsInfo (gobwa.go:26) */
/* ALT contigs — bntann1_t.is_alt, which bwa_idx_load(path, BWA_IDX_ALL) (gobwa/gobwa.go:130) restores from <prefix>.alt (bntseq.c
 * bns_restore: per line the first token up to a tab / newline is a contig name, lines starting with '@' are skipped, a name that
 * matches a contig of .ann marks it; a last line without a newline is not seen).  mem_chain_flt does not let an ALT chain shadow a
 * primary one, and every region on an ALT contig carries is_alt.  lh_index_load reads the file when it exists; lh_index_set_alt does
 * the same from an array of n_contigs flags (indexes built from arrays or on the device); lh_index_alt returns the resident flags
 * (NULL: none); lh_index_save writes <prefix>.alt when any contig is ALT. */
int lh_index_set_alt(lh_index* idx, const uint8_t* is_alt);
const uint8_t* lh_index_alt(const lh_index* idx);
int lh_index_contigs(const lh_index* idx, int32_t* n, const char* const** names, const int64_t** lens, const int64_t** offsets);
int64_t lh_index_l_pac(const lh_index* idx);
/* The .sa file holds every 32nd suffix-array row (bwa index default); bwt_sa (reached from mem_chain via gobwa.go:244,253)
 * walks the BWT to the next sampled row.  On load the samples are re-derived ON THE DEVICE at the densest power-of-two
 * interval whose table fits half of the free HBM (<= 128 GiB; lh_index_opts.sa_intv forces one), so a
 * lookup costs (interval-1)/2 occurrence-block reads instead of 15.5.  The values are exact at any interval; results do not
 * change.  lh_index_resample_sa switches the resident table to another interval (denser: walk; sparser: sub-sample). */
int lh_index_resample_sa(lh_index* idx, int32_t sa_intv);
int32_t lh_index_sa_interval(const lh_index* idx);
void lh_index_free(lh_index* idx);

/* FM-index construction: text = fwd || revcomp of the 2-bit contigs, BWA-byte-compatible output.
 * Host-side (multi-threaded suffix sorting); writes <prefix>.bwt/.sa/.pac/.ann/.amb.  Ambiguous bases as in lh_reference_pack. */
int lh_index_build(const char* prefix, int32_t n_contigs, const char* const* names, const uint8_t* const* nt4, const int64_t* lens, int32_t threads);
/* The same construction ON THE DEVICE, sized for a human genome in 288 GB of HBM (hg38: 6.2 G suffixes): suffixes of
 * fwd || revcomp are gathered by 8-mer prefix into chunks, radix-sorted by their first 32 bases, ties finished by direct
 * comparison through the packed text; BWT, occurrence table, dense suffix array and K1's side tables are built from the
 * result without leaving HBM.  `pac` is the .pac file's 2-bit forward reference (l_pac/4+1 bytes).  The index equals the
 * one lh_index_load makes from `bwa index` files of the same reference (lh_index_export gives those files' contents). */
int lh_index_build_device(int device, const uint8_t* pac, int64_t l_pac, int32_t n_contigs, const int64_t* contig_off, const int32_t* contig_len,
                          const char* const* contig_name, const lh_index_opts* io, lh_index** out);
/* the resident index in the byte layout of <prefix>.bwt / .sa: call with bwt == NULL to get the sizes, then with buffers.
 * sa_intv: 32 for `bwa index` compatibility (a multiple of the resident interval); sa[0] = (uint64_t)-1 as in memory. */
int lh_index_export(const lh_index* idx, uint64_t* primary, uint64_t L2[5], uint32_t* bwt, uint64_t* bwt_words, int32_t sa_intv, uint64_t* sa, uint64_t* n_sa);
/* writes <prefix>.bwt .sa .pac .ann .amb (what lh_index_build writes) from the resident index */
int lh_index_save(const lh_index* idx, const char* prefix);
/* bns_fasta2bntseq (bntseq.c) restated, host-only: the .pac image (l_pac/4+1 bytes) of the contigs, ambiguous bases replaced by
 * lrand48() & 3 under srand48(11) in file order as `bwa index` does, and the holes they leave (.amb: offset, length, letter; a run
 * of the same letter is one hole; n_ambs[c] = holes starting in contig c, for .ann).  Bases are nt4 codes (0..3, 4 = N) or raw
 * FASTA letters (values > 4).  Call with max_holes 0 and NULL arrays to count.  lh_index_build does this itself; for
 * lh_index_build_device pack first, build, then lh_index_set_holes so that lh_index_save writes the same .ann / .amb. */
int lh_reference_pack(int32_t n_contigs, const uint8_t* const* seqs, const int64_t* lens, uint8_t* pac, int32_t* n_ambs, int32_t max_holes,
                      int64_t* hole_off, int32_t* hole_len, char* hole_char, int32_t* n_holes);
int lh_index_set_holes(lh_index* idx, int32_t n_holes, const int64_t* hole_off, const int32_t* hole_len, const char* hole_char);

int lh_context_create(lh_index* idx, int64_t max_pairs_per_batch, const lh_context_opts* co, lh_context** out);
void lh_context_free(lh_context* ctx);

/* THE hot path: DoRFAForOneBarcode for every barcode of the batch (lariat.go:461-547), minus DumpToBams.
 * Inputs are host buffers; lh_align_barcodes = upload + lh_align_resident + download. */
int lh_align_barcodes(lh_context* ctx, const lh_opts* opts, const lh_batch* batch, lh_result** out);
/* split form used by bench.py: inputs resident in HBM when the timed region starts */
int lh_batch_upload(lh_context* ctx, const lh_batch* batch);
int lh_align_resident(lh_context* ctx, const lh_opts* opts);            /* enqueue all kernels + sync */
/* further input batches kept resident next to lh_batch_upload's (which is slot 0): a driver that streams many batches through
 * one context uploads ahead (slot k+1 while slot k is aligned) and switches with lh_batch_select; bench.py has every batch of
 * its timed region in HBM this way.  A slot's batch must fit the context's capacity like any other. */
int lh_batch_upload_slot(lh_context* ctx, int32_t slot, const lh_batch* batch);
int lh_batch_select(lh_context* ctx, int32_t slot);
/* The download in two steps, for a host that keeps ONE context busy: lh_result_download_begin checks the batch (the errors of
 * lh_result_download are reported here), packs its variable-length arrays and enqueues the device-to-host copies on the context's copy
 * stream; lh_result_download_end waits for them and hands out the result.  Between the two the host may upload / select / align the NEXT
 * batch: its kernels run while the copies do (lh_align_resident waits for them only before it first writes a result array), which is the
 * double buffering of the reference's work-unit channels (inference/lariat.go:333, bamwriter.go:188).  One download in flight per context. */
int lh_result_download_begin(lh_context* ctx);
int lh_result_download_end(lh_context* ctx, lh_result** out);
/* lh_batch_upload_slot without selecting the slot, on the context's upload stream: may be called from a second host thread while
 * lh_align_resident runs on another (selected) slot.  One staging call at a time; contexts with one lane. */
int lh_batch_stage_slot(lh_context* ctx, int32_t slot, const lh_batch* batch);
/* page-locked host memory for the arrays of a batch: uploads from it are asynchronous DMA transfers that run beside
 * the kernels of another batch; from ordinary memory they are staged by the calling thread and slow down under a busy device.  A host fills
 * such buffers straight from its FASTQ reader. */
void* lh_host_alloc(size_t bytes);
void lh_host_free(void* p);
int lh_result_download(lh_context* ctx, lh_result** out);
void lh_result_free(lh_result* r);

/* per-kernel timing of the last lh_align_resident (HIP events on the context's stream), ms; names are static strings */
int lh_last_timings(lh_context* ctx, int32_t* n, const char* const** names, const float** ms);

/* stage dump of the candidate-generation front end (mem_align1_core) for the resident batch: parity tests only */
int lh_stage_dump_resident(lh_context* ctx, const lh_opts* opts, lh_stage_dump** out);
void lh_stage_dump_free(lh_stage_dump* d);

/* GetSeq (gobwa.go:50-80): forward-coordinate slice [start,end) of contig rid as ASCII, reverse-complemented if reversed */
int lh_get_seq(const lh_index* idx, int32_t rid, int64_t start, int64_t end, int32_t reversed, char* out /* end-start bytes */);

/* diagnostics: rate (GB/s of requested bytes) of independent random reads of `granule_bytes` blocks (multiple of 16)
 * from a `table_bytes` table in HBM — the practical ceiling for the FM-index walks */
/* free / total bytes of device memory right now (hipMemGetInfo): hosts sizing batches and lanes leave room for the kernels' scratch */
int lh_device_memory(int device, int64_t* free_bytes, int64_t* total_bytes);
int lh_diag_random_read(int device, int64_t table_bytes, int32_t granule_bytes, int64_t n_access, double* gbps, double* ms);

/* diagnostics: the rate at which the SIMDs issue vector-ALU instructions (csrc/k_valu_rate.h) — the ceiling of K6's Smith-Waterman
 * (gobwa.go:286-325 -> mem_matesw -> ksw_align2 in packed 16-bit cells).  op: 0..49, 53..55 one opcode in eight independent chains, 100 + op the same in
 * one dependent chain, 50 the instruction mix of k_resc_sw's column, 51 v_pk_fma_f32, 52 s_nop, 56 and 57 half- and full-rate opcodes interleaved.  waves_per_simd 1..8 over the whole chip.
 * out[0..9]: ms, wave-instructions, G wave-instructions/s, median shader MHz, median cycles per instruction per WAVE, cycles per instruction per
 * SIMD chip-wide, SIMDs seen, fewest / most waves on one SIMD, lowest MHz.  LH_E_NODEVICE without a GPU (and under the emulator). */
int lh_diag_valu_rate(int device, int32_t op, int32_t waves_per_simd, int32_t iters, double* out, int32_t n_out);

/* diagnostics: K6's Smith-Waterman as the pipeline runs it (k_resc_cert -> k_resc_sw forward -> reverse; csrc/k_rescue3.h) on caller-supplied cases: queries as
 * ksw_align2 sees them and windows, nt4 bytes 0..3, q_off / t_off [n_cases + 1].  full != 0: every window whole (LH_F_RESCUE_FULL).  weaken: bits that switch terms
 * of the certificate off (tests of the crafted cases each term guards against; 0 = the pipeline's).  out: 8 per case — score, te, qe, tb, qb (ksw_align2's, mem_matesw's
 * call: gobwa.go:286-325; tb = qb = -1 below min_seed_len), then rlo (-1: settled without DP), rn (rows the forward pass ran), rows2 (rows of the reverse pass). */
int lh_diag_rescue_sw(int device, int32_t n_cases, const int32_t* q_off, const uint8_t* q, const int32_t* t_off, const uint8_t* t, int32_t full, int32_t weaken, int32_t* out);

/* diagnostics: self-check of a resident index with a dense suffix array, on every stride-th row: adjacent suffixes are in
 * order (direct text comparison); the stored BWT symbol is the base before the suffix and the LF-mapping through the
 * occurrence table reaches that suffix's row.  A size-independent property for indexes the oracle cannot hold. */
int lh_diag_index_check(const lh_index* idx, uint64_t stride, uint64_t* n_checked, uint64_t* n_bad_order, uint64_t* n_bad_lf);

/* diagnostics: order-sensitive checksums of the resident LCP array and k-mer tree table (0 when absent) — the tables
 * lh_index_build_device derives from its sort keys must equal the ones derived from the text for a loaded index */
int lh_diag_index_digest(const lh_index* idx, uint64_t* lcp_digest, uint64_t* ktree_digest, int32_t* ktree_levels);

/* Workload generators for bench.py and the tests (no genome but PhiX exists offline).  Host-only, threaded.
 * lh_synth_genome: iid ACGT with GC fraction `gc` as a .pac image (l_pac/4+1 bytes, MSB-first 2-bit), reproducible from `seed`
 * whatever the thread count.  lh_synth_reads: barcode-sorted FR read pairs under the linked-read model — per barcode
 * mol_min..mol_max molecules (log-normal length, median 50 kb, clipped to [10 kb, 200 kb]), pairs_per_barcode pairs spread
 * over them in proportion to length, insert ~ N(350,50) clipped to [max(200,len),700], substitution error ramping
 * sub_lo -> sub_hi along each read, indels at indel_rate per base (1-3 bases), junk_frac of the reads replaced by random
 * bases.  Output arrays as an lh_batch wants them; truth_pos (forward-strand start of read 1 / read 2's fragment end) may be NULL. */
int lh_synth_genome(uint64_t seed, double gc, int64_t l_pac, uint8_t* pac, int32_t threads);
/* the reads of lh_synth_reads as the reference's input file: 9-line barcode-sorted FASTQ (README.md:34-48), gzip'ed when gz_level > 0; read 1 gets `trim`
 * random bases in front; barcode b is the 16-mer of first_barcode + b in base 4 + "-1" (ascending = sorted); names mol:<bc>:<rid>:0:0:<pos1>:<pos2> */
int lh_synth_write_fastq9(const char* path, const uint8_t* seq, const int64_t* seq_off, const int32_t* bc_pair_off, int32_t n_barcodes, int64_t first_barcode,
                          int32_t trim, int32_t gz_level, uint64_t seed, const int32_t* truth_rid, const int64_t* truth_pos1, const int64_t* truth_pos2);
int lh_synth_reads(const uint8_t* pac, int64_t l_pac, int32_t n_contigs, const int64_t* contig_off, const int32_t* contig_len, uint64_t seed,
                   int32_t n_barcodes, int32_t pairs_per_barcode, int32_t len1, int32_t len2, double sub_lo, double sub_hi, double indel_rate,
                   double junk_frac, int32_t mol_min, int32_t mol_max, int32_t threads,
                   uint8_t* seq /* cap 2*n_pairs*(max(len1,len2)+3) */, int64_t* seq_off /* 2*n_pairs+1 */, int32_t* bc_pair_off /* n_barcodes+1 */,
                   uint64_t* name_seed /* n_pairs */, int32_t* truth_rid /* n_pairs or NULL */, int64_t* truth_pos1, int64_t* truth_pos2);

/* diagnostics: the device's restatement of Go's math/rand source (the jitter stream of tagBestAlignments, lariat.go:1486,
 * 1499,1510): n draws of rand.New(rand.NewSource(seed)) as Uint64 — out_fast from K8's state-free path (first min(n,273)
 * draws, the rest 0), out_ring from its state ring — and as Float64 (out_f64). */
/* diagnostics: Go 1.9's sort.Sort as K8 restates it (lariat.go:1546 ByPosition; its order of equal keys is part of the result): n_sorts
 * index spaces [first[k], first[k+1]) of keys[] (first[0] = 0) sorted by the serial restatement (one sort per lane) and by the wave-wide one;
 * each returns the permutation of every index space (position -> original index inside its space) */
int lh_diag_gosort(int device, int32_t n_sorts, const int32_t* first, const int64_t* keys, int32_t* perm_serial, int32_t* perm_wave);
/* the same sort of ONE index space [0, n) the way K8 sorts a contig list too long for LDS: ranges longer than `limit` (> 12) partitioned by the whole wave, the
 * ranges left sorted one by one with the depth the long sort has left them.  perm: where Go leaves the elements (equal keys included). */
int lh_diag_gosort_split(int device, int32_t n, const int64_t* keys, int32_t* perm, int32_t limit);
/* K8's sorting network for lists of distinct keys (the position sort of a contig's candidates, lariat.go:1546 ByPosition, when no two positions are equal): n words
 * ascending.  block = 64 or 1024: the places that run in LDS at a time (longer lists: block by block with passes in memory between them); 0: every step in memory. */
int lh_diag_bitonic(int device, int32_t n, const uint64_t* keys, uint64_t* sorted, int32_t block);
/* diagnostics: klib's ks_introsort as K5 / K6 restate it for the region sorts of mem_sort_dedup_patch (bwamem.c via gobwa.go:244,253,291,315; its order of equal keys is part
 * of the result): n_sorts index spaces of keys[] (at most 1024 elements each, keys below 2^50) sorted by the one-lane restatement and by the wave-wide one; each returns the
 * permutation of every index space (position -> original index inside its space) */
int lh_diag_introsort(int device, int32_t n_sorts, const int32_t* first, const int64_t* keys, int32_t* perm_serial, int32_t* perm_wave);
int lh_diag_go_rand(int device, int64_t seed, int32_t n, uint64_t* out_fast, uint64_t* out_ring, double* out_f64);
/* diagnostics: K6's exact shortcut for the mem_sort_dedup_patch(opt, 0, 0, 0, ...) that follows every mem_matesw attempt (gobwa.go:291,315 -> bwamem
 * mem_matesw), against the call as written.  n_cases lists of regions [first[c], first[c+1]) (6 int64 each: rb, re, qb, qe, score, rid) and one region
 * added[c] per case; verdict[c]: 0 = the incremental form and the full call agree on the resulting list, 1 = the incremental form declined (equal keys:
 * the pipeline runs the full call), 2 = THEY DIFFER, 3 = the list itself has equal end positions (not eligible); n_out[c] = the full call's result length | the cleaned list's length << 16 */
int lh_diag_rescue_dedup(int device, int32_t n_cases, const int32_t* first, const int64_t* regions, const int64_t* added, int32_t max_chain_gap, int32_t* verdict,
                         int32_t* n_out);

/* ------------------------------------------------------------------------------------------------------------------
 * N2 — 9-line barcode-sorted FASTQ ingest: go/src/fastqreader/reader.go (ReadOneLine :91-147,
 * ReadBarcodeSet :173-260), zipread.go:62-85 (gunzip pipe), and the driver's use of a set (lariat.go:353-375 the read loop,
 * :1088-1100 worthRunningRFA, :1483-1484 the tie-break seed).  Host-only code: no GPU is touched.
 *
 * A "set" is what one call of ReadBarcodeSet returns = one WorkUnit = one barcode entry of an lh_batch.  The reader's
 * quirks are part of the contract and are reproduced: a set holds at most `cap` (30,000) pairs; a barcode without '-'
 * (not whitelisted) is cut every `chunk` (200) pairs; the continuation of a barcode that was cut at the cap breaks at 200
 * ("abnormal break") and is flagged incomplete; text before a '@' line is skipped as "Bad line"; a truncated final
 * record is dropped; read 1 loses its first min(len, trim) bases (kept as trim_bases / trim_quals for the TR/TQ tags);
 * the barcode line may be "corrected,raw". */
typedef struct lh_ingest lh_ingest;

typedef struct lh_ingest_batch {
    lh_batch batch;              /* ready for lh_align_barcodes / lh_batch_upload (cen_* left NULL) */
    int32_t n_sets;              /* == batch.n_barcodes */
    const uint8_t* set_complete; /* [n_sets] ReadBarcodeSet's third result (WorkUnit.unique_barcode, lariat.go:214,374) */
    /* per-pair text kept for the BAM stage (bamwriter.go): offsets [n_pairs+1] into byte arenas, no terminators */
    const int64_t *name_off, *rgid_off, *qual1_off, *qual2_off, *trim_off, *bc_off, *rawbc_off, *bcqual_off, *si_off, *siqual_off;
    const char *name, *rgid, *qual1, *qual2, *trim_bases, *trim_quals, *bc, *rawbc, *bcqual, *si, *siqual;
    int64_t first_set_index;     /* index of batch set 0 among all sets read so far (lariat's barcode_num - 1) */
    int32_t at_eof;              /* 1: the input is exhausted after this batch */
    void* arena_;
} lh_ingest_batch;

/* gz input (magic 1f 8b) is read through a `gunzip -c` pipe like the reference, anything else directly.
 * cap / chunk <= 0 select the reference's 30000 / 200. */
int lh_ingest_open(const char* path, int32_t trim, int32_t cap, int32_t chunk, lh_ingest** out);
/* Appends whole sets until the next one would exceed max_pairs (always at least one set; a set never straddles batches).
 * Returns LH_OK with out->batch.n_pairs == 0 and at_eof == 1 when nothing is left. */
int lh_ingest_next(lh_ingest* in, int64_t max_pairs, lh_ingest_batch** out);
void lh_ingest_batch_free(lh_ingest_batch* b);
void lh_ingest_close(lh_ingest* in);
/* md5 of `n` bytes -> the tie-break seed lariat derives from a read name (LE u64 of digest[0:8]) */
uint64_t lh_name_seed(const char* name, int64_t n);

/* ------------------------------------------------------------------------------------------------------------------
 * N1, first half — BAM record CONTENT: what go/src/inference/bamwriter.go puts into each record
 * (DoDumpToBam :634-657 which alignments, in which order; AppendBam :286-568 flags incl. the `!is_proper && score-17 < 19`
 * unmapping rule and its side effect on the mate's record, mate fields, TLEN, reverse-complemented SEQ/QUAL, HardClip
 * :663-688 for the split record, and the tags RX QX TR TQ BC QT RG XS XC AC AS XM AM XT SA BX DM in the reference's
 * order; the -debugTags set is not produced).  Host-only.  One text line per record, tab separated, BAM-native values:
 *   QNAME FLAG RNAME|* POS0 MAPQ CIGAR|* RNEXT|* PNEXT0 TLEN SEQ|* QUAL|* TAG:TYPE:VALUE...
 * (positions 0-based, -1 = none, as bam.Record holds them).  The binary encoding / BGZF / bucketing into files is the
 * second half of N1.  `res` and `in` must describe the same batch; contig_names[rid] as from lh_index_contigs. */
int lh_records_text(const lh_result* res, const lh_ingest_batch* in, int32_t n_contigs, const char* const* contig_names, char** text, int64_t* text_len);
/* the same with options: LH_REC_DEBUG_TAGS adds the tags of -debugBamTags (bamwriter.go:498-558: XM XZ XX XL XP XR XC for the second-best
 * alignment, AA CP CM CU CS RD MS MC PP PS PL AC PC), all derived from the result's per-candidate fields; AA is empty, as it is in the
 * reference without -debug */
#define LH_REC_DEBUG_TAGS 1
int lh_records_text_ex(const lh_result* res, const lh_ingest_batch* in, int32_t n_contigs, const char* const* contig_names, int32_t flags, char** text, int64_t* text_len);
void lh_records_free(char* text);

/* N1, second half — the BAM container (bamwriter.go:46-191 CreateBAM / CreateBAMs, :281-284 AppendBams): BGZF-compressed
 * BAM files in `dir`: bc_sorted_bam.bam (every record, input order) and the position-bucketed files
 * %06d-<contig>_%010d_pos_bucketed.bam (contigs longer than position_chunk_size split every position_chunk_size bases,
 * shorter contigs packed together up to that size) plus ZZZ_unmapped_pos_bucketed.bam for IsUnmapped records
 * (lariat.go:143-148).  Header: @HD, @SQ per contig, one @RG per fully specified read group id
 * (sample:library:gem_group:flowcell:lane), @PG lariat, and the three 10x_bam_to_fastq @CO lines when first_chunk != 0
 * (only on the first file of each kind, as upstream).  Blocks are compressed by `threads` host threads (<= 0: all cores). */
typedef struct lh_bam_writer lh_bam_writer;
int lh_bam_open(const char* dir, int32_t n_contigs, const char* const* contig_names, const int64_t* contig_lens, const char* read_groups,
                int32_t position_chunk_size, int32_t first_chunk, const char* command_line, int32_t threads, lh_bam_writer** out);
/* appends the records of one batch (lh_records_text order) to bc_sorted_bam.bam and to their position bucket */
int lh_bam_append(lh_bam_writer* w, const lh_result* res, const lh_ingest_batch* in);
int lh_bam_set_level(lh_bam_writer* w, int32_t level);  /* zlib level of the BGZF blocks written from now on: -1 (default) .. 9 */
int lh_bam_timings(const lh_bam_writer* w, double* records_s, double* join_s, double* write_s);  /* the last lh_bam_append's phases */
int lh_bam_set_flags(lh_bam_writer* w, int32_t flags); /* LH_REC_* for the records appended from now on (CreateBAMs' debugTags, bamwriter.go:133) */
int lh_bam_close(lh_bam_writer* w); /* flushes, writes the BGZF end-of-file blocks, frees w */
/* The host-side step of the multi-GPU path: every rank aligns a contiguous barcode range and writes its own file set
 * (lh_bam_open with first_chunk only on rank 0); the job's files are the rank-ordered concatenation, file by file — for
 * bc_sorted_bam.bam that is the single-process file (input order), for a position bucket the same multiset of records.
 * BGZF blocks are copied as they are; later shards lose their header, every end-of-file block but the last is dropped. */
int lh_bam_concat(int32_t n_shards, const char* const* shard_dirs, const char* out_dir);

#ifdef __cplusplus
}
#endif
#endif /* LARIAT_HIP_H */
