/*
 * circminer_hot.h — C-ABI of the MI355X-native CircMiner mapping hot path.
 *
 * The reference (CircMiner 0.4.5) has no plugin/FFI boundary: the seam this
 * library replaces is
 *
 *     int FilterRead::process_read(int thid, Record* r1, Record* r2, int kmer_size,
 *                                  GIMatchedKmer* fl, GIMatchedKmer* bl,
 *                                  chain_list& fbc_r1, chain_list& bbc_r1,
 *                                  chain_list& fbc_r2, chain_list& bbc_r2)
 *                                                       (src/filter.h:49-52, src/filter.cpp:124-241)
 *
 * called once per read pair from map_reads (src/circminer.cpp:384), together with the
 * process-wide state it reads: the loaded mrsfast k-mer table (src/mrsfast/HashTable.c:971-1098),
 * the decoded contig string (src/match_read.cpp:288-299,334), the GTF model
 * (src/gene_annotation.{h,cpp}) and the threshold globals (src/common.h:92-103).
 *
 * Everything here is plain C: pointers, sizes, PODs.  No C++ / torch types.
 * All functions return 0 on success or a negative CM_E* code; none of them calls exit().
 */
#ifndef CIRCMINER_HOT_H
#define CIRCMINER_HOT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- category codes: src/common.h:59-72 (the order matters) ---- */
enum {
    CM_CONCRD = 0, CM_DISCRD = 1, CM_CHIORF = 2, CM_CHIBSJ = 3, CM_CHI2BSJ = 4, CM_CONGEN = 5,
    CM_CHIFUS = 6, CM_CONGNM = 7, CM_OEA2 = 8, CM_CANDID = 9, CM_OEANCH = 10, CM_ORPHAN = 11,
    CM_NOPROC_MANYHIT = 12, CM_NOPROC_NOMATCH = 13
};

/* ---- error codes ---- */
enum {
    CM_OK = 0,
    CM_EINVAL = -1,     /* bad argument / unsupported parameter combination            */
    CM_ENODEV = -2,     /* no HIP device (the product path never falls back to the CPU) */
    CM_ENOMEM = -3,     /* host or device allocation failed                             */
    CM_EHIP = -4,       /* a HIP runtime call or kernel failed                          */
    CM_ESTATE = -5,     /* contig / annotation not loaded                               */
    CM_ELIMIT = -6,     /* a documented capacity limit of the device path was exceeded  */
    CM_EIO = -7         /* a read from an input file or a write to an output file failed */
};

#define CM_WINDOW_SIZE 14          /* WINDOW_SIZE, src/common.cpp:7                         */
#define CM_BESTCHAINLIM 30         /* BESTCHAINLIM, src/common.h:51; chain.h:14-17          */
#ifndef CM_MAX_CHAIN_FRAGS         /* (the library's second build of the kernels sets 24 for itself: reads of more than 16 seeds) */
#define CM_MAX_CHAIN_FRAGS 16      /* fragments a cm_chain of THIS ABI holds (cm_chain_batch, a test hook, is limited to reads of <= 16 seeds) */
#endif
#define CM_MAX_SEEDS_PER_READ 24   /* floor(read length / k) the mapping entry points accept: 300 bp at k = 14 is 21 (commandline_parser.cpp:14,242-247) */
#define CM_CONTIG_SIZE 1100000000u /* DEF_CONTIG_SIZE, src/common.h:81 (gspos stride)       */

/* ---- thresholds: the reference keeps these as globals (src/common.h:92-103,
 *      defaults src/commandline_parser.cpp:7-33) ---- */
typedef struct cm_params {
    int32_t kmer;            /* WINDOW_SIZE + checkSumLength, 14..22 (src/commandline_parser.cpp:242-247) */
    int32_t seed_lim;        /* seedLim       default 500     */
    int32_t max_read_len;    /* maxReadLength default 300     */
    int32_t scan_level;      /* scanLevel     default 0       */
    int32_t max_ed;          /* maxEd         default 4       */
    int32_t max_sc;          /* maxSc         default 7       */
    int32_t band;            /* bandWidth     default 3       */
    int32_t max_tlen;        /* maxTlen       default 500     */
    int32_t max_intron;      /* maxIntronLen  default 2000000 */
    int32_t max_chain_len;   /* maxChainLen   default 30 (must be <= CM_BESTCHAINLIM, see chain.h:14-17) */
    int32_t device;          /* HIP device ordinal            */
    int32_t reserved;
} cm_params;

/* ---- one packed contig of the k-mer index, as loadHashTable leaves it in memory
 *      (src/mrsfast/HashTable.c:971-1057) but flattened: bucket hv owns entries
 *      [bucket_off[hv], bucket_off[hv+1]) sorted by (checksum, pos)
 *      (src/mrsfast/Sort.c:116-117); pos is the 1-based k-mer start
 *      (HashTable.c:785-806).  genome is what pac2char serves
 *      (src/match_read.cpp:288-299): ASCII, genome[p-1] = base at 1-based p. ---- */
typedef struct cm_index_view {
    int32_t contig_num;          /* contigNum = atoi(contigName)-1, src/circminer.cpp:266-268 */
    uint32_t ref_len;            /* getRefGenLength()                                         */
    const uint8_t *genome;       /* ref_len bytes, 'A','C','G','T','N'                        */
    const uint32_t *bucket_off;  /* 4^14 + 1 offsets                                          */
    const uint16_t *checksum;    /* n_entries                                                 */
    const uint32_t *pos;         /* n_entries                                                 */
    uint64_t n_entries;
} cm_index_view;

/* ---- flattened query side of GTFParser for one packed contig
 *      (src/gene_annotation.{h,cpp}, src/interval_tree_impl.h) ---- */
typedef struct cm_annot_view {
    /* disjoint intervals, ascending (FlatIntervalTree::disjoint_intervals) */
    uint32_t n_iv;
    const uint32_t *iv_spos, *iv_epos;
    const uint32_t *iv_max_end, *iv_min_end, *iv_max_next_exon; /* interval_tree_impl.h:198-211 */
    const uint32_t *iv_seg_off;  /* n_iv+1 : CSR into iv_seg                                   */
    const uint32_t *iv_seg;      /* indices into the unique-segment tables, seg_list order     */
    /* unique segments (UniqSeg, src/common.h:227-251) */
    uint32_t n_seg;
    const uint32_t *seg_start, *seg_end, *seg_next_exon_beg, *seg_gene_id;
    const uint32_t *seg_tid_off; /* n_seg+1 : CSR into seg_tid (trans_id vector order)         */
    const uint32_t *seg_tid;
    /* transcripts */
    uint32_t n_trans;
    const int32_t *trans_start_ind;  /* get_trans_start_ind, gene_annotation.h:143-145         */
    const uint32_t *t2s_off;         /* n_trans+1 : CSR into t2s                               */
    const uint8_t *t2s;              /* trans2seg states 0/1/2/3, interval_tree_impl.h:232-241 */
    /* genes (gid2ginfo, src/gene_annotation.cpp:243-245) */
    uint32_t n_gene;
    const uint32_t *gene_start, *gene_end;
    /* position bitsets (src/common.h:117-118); bit p = contig position p */
    uint64_t n_bits;
    const uint64_t *near_border_bits;
    const uint64_t *intronic_bits;
    /* chromosomes packed into this contig, ascending by shift (con2chr, gene_annotation.cpp:424-457) */
    uint32_t n_chr;
    const uint32_t *chr_shift;   /* ContigLen.start_pos                                        */
    const int32_t *chr_id;       /* global chromosome ordinal (row of .index.info)             */
    /* optional accelerator for FlatIntervalTree::search (interval_tree_impl.h:136-150): iv_bucket[b] =
     * number of intervals with spos < (b << iv_bucket_shift), b = 0 .. n_iv_bucket-1 (covers n_bits + one
     * extra bucket).  NULL = plain binary search.  Results are identical either way. */
    const uint32_t *iv_bucket;
    uint32_t iv_bucket_shift, n_iv_bucket;
    /* genes_int_map (stage 2 only, never uploaded: GTFParser::get_gene_overlap, src/gene_annotation.cpp:572-585): disjoint
     * intervals over the gene spans; interval i is covered by the genes giv_gene[giv_gene_off[i] .. giv_gene_off[i+1]) in the
     * reference's seg_list order (indices into gene_start / gene_end; genes with identical spans are represented by the
     * first of them, as in the reference's map keyed by (start, end)) */
    uint32_t n_giv;
    const uint32_t *giv_spos, *giv_epos, *giv_gene_off, *giv_gene;
} cm_annot_view;

/* ---- POD mirror of MatchedRead (src/common.h:311-352); carried between rounds through the
 *      23-token FASTQ header in the reference (src/filter.cpp:413-455, fastq_parser.cpp:203-269).
 *      chr_r1 == chr_r2 always (src/common.cpp:295-296) so one chr_id is kept; -1 is "-". ---- */
typedef struct cm_mapped_read {
    uint32_t spos_r1, spos_r2, epos_r1, epos_r2;
    uint32_t qspos_r1, qspos_r2, qepos_r1, qepos_r2;
    uint32_t mlen_r1, mlen_r2;
    int32_t ed_r1, ed_r2;
    int32_t type;
    int32_t tlen;
    int32_t contig_num;
    int32_t chr_id;
    uint16_t junc_num;
    uint8_t r1_forward, r2_forward, gm_compatible, pad[3];
} cm_mapped_read;

/* ---- one chain (chain_t, src/common.h:150-154) as produced by chain_seeds_sorted_kbest ---- */
typedef struct cm_chain {
    float score;
    uint32_t chain_len;
    uint32_t rpos[CM_MAX_CHAIN_FRAGS];
    int32_t qpos[CM_MAX_CHAIN_FRAGS];   /* fragment_t.len is always kmer (src/chain.cpp:265,293) */
} cm_chain;

typedef struct cm_ctx cm_ctx;

/* Reads are passed as concatenated upper/lower-case ASCII with offsets:
 * pair i = R1 bytes [off1[i], off1[i+1]) of seq1, R2 bytes [off2[i], off2[i+1]) of seq2. */
typedef struct cm_reads {
    uint64_t n_pairs;
    const uint8_t *seq1;
    const uint64_t *off1;   /* n_pairs+1 */
    const uint8_t *seq2;
    const uint64_t *off2;   /* n_pairs+1 */
} cm_reads;

/* ---------------- context ---------------- */
int cm_create(const cm_params *p, cm_ctx **out);
void cm_destroy(cm_ctx *ctx);
const char *cm_last_error(const cm_ctx *ctx);      /* never NULL */

/* Stage one packed contig (index + decoded genome) into HBM.  Replaces loadHashTable +
 * pac2char_whole_contig (src/circminer.cpp:232,244).  slot is the caller's handle (0..15). */
int cm_load_contig(cm_ctx *ctx, int slot, const cm_index_view *iv);
/* Stage the flattened annotation of the same contig.  Replaces gtf_parser.load_gtf's
 * query-side state for contigNum (src/circminer.cpp:205-206). */
struct cm_index_raw;
/* cm_load_contig from a record as cm_host_next_contig_raw leaves it: the table goes over PCIe as it is in the file and is
 * flattened into the cm_index_view layout BY THE DEVICE (bucket offsets by two scans, entries scattered at HBM bandwidth)
 * instead of by host threads (loadHashTable's pointer table + arena, src/mrsfast/HashTable.c:1013-1034, never exists).  The
 * resident contig is the same as after cm_host_next_contig + cm_load_contig.  A malformed table (a header slot that claims
 * more entries than its bucket has slots) is CM_EINVAL. */
int cm_load_contig_raw(cm_ctx *ctx, int slot, const struct cm_index_raw *raw);
int cm_load_annotation(cm_ctx *ctx, int slot, const cm_annot_view *av);
int cm_unload_contig(cm_ctx *ctx, int slot);

/* Upload a batch of read pairs once; they stay resident for all rounds.
 * prior may be NULL (= first round, fill_map_info's "cnt != 23" state). */
int cm_reads_upload(cm_ctx *ctx, const cm_reads *reads, const cm_mapped_read *prior);

/* Double buffering (the reference overlaps nothing: map_reads parses a block, maps it, prints it, src/circminer.cpp:354-400):
 * cm_reads_stage copies the NEXT batch into a second set of read buffers on a copy stream and returns at once when the host
 * arrays are page-locked (cm_host_alloc; pageable memory is accepted and copied synchronously by the runtime); the rounds of
 * the resident batch run meanwhile.  cm_reads_swap makes the staged batch the resident one (first-round state, or `prior`):
 * it is ordered on the device behind the staged copies and behind the work already queued for the old batch, so the host
 * does not wait.  The host arrays passed to cm_reads_stage must stay untouched until cm_reads_swap has been followed by a
 * cm_sync / download / collect, or until the next cm_reads_stage returns. */
int cm_reads_stage(cm_ctx *ctx, const cm_reads *reads, const cm_mapped_read *prior);
int cm_reads_swap(cm_ctx *ctx);

/* One mapping round of the resident batch against contig `slot`:
 * FilterRead::process_read for every pair still active + the skip rule of
 * map_reads (src/circminer.cpp:386-397).  Pairs whose carried state says they were
 * retired in an earlier round (active[i]==0) are left untouched.
 * Asynchronous on the ctx stream; cm_sync() or cm_reads_download() waits. */
int cm_map_round(cm_ctx *ctx, int slot, int is_last_round);

/* Several rounds of the resident batch in one call: rounds slots[0 .. n_rounds) in that order, the last one with
 * is_last_round = last_is_final -- the rounds loop of mapping() (src/circminer.cpp:229-308) for contigs that are all resident.
 * Results are those of n_rounds cm_map_round calls.  What differs is the schedule: seeds and chains of a round depend on the
 * reads and the contig only (the carried MatchedRead enters in the pair stage), so round r + 1 is seeded and chained on the
 * main streams while the pair stage of round r still runs on a second pair of streams; chain buffers and active flags are
 * double-buffered.  Asynchronous like cm_map_round.  A batch of more than 2^20 pairs is mapped in tiles (two of up to 2^21 pairs,
 * more for batches beyond 2^22; ~46 KB of HBM workspace per pair of a tile), walked round by round
 * (every tile through round r, then every tile through round r + 1), so a tile's seeding sees the flags its previous pair stage
 * wrote and only pairs still active are seeded and chained.
 * Across batches: when last_is_final is set and a batch is staged (cm_reads_stage) that fits the workspace of the resident
 * one, its first tile's first round against slots[0] is seeded and chained under this batch's last pair stage; after cm_reads_swap the next
 * cm_map_rounds takes that work over if its slots[0] is the same slot, still holding the same contig and annotation
 * (otherwise it is discarded and redone: results never depend on it). */
int cm_map_rounds(cm_ctx *ctx, const int *slots, int n_rounds, int last_is_final);

/* Copy back the carried state, the per-pair return value of process_read in the last
 * cm_map_round (state[i], -1 if the pair was inactive) and the active flags after it. */
int cm_reads_download(cm_ctx *ctx, cm_mapped_read *out_state, int32_t *out_category, uint8_t *out_active);

/* Convenience wrapper = upload + one round + download (the literal batched process_read). */
int cm_map_batch(cm_ctx *ctx, int slot, int is_last_round, const cm_reads *reads,
                 const cm_mapped_read *prior, cm_mapped_read *out_state, int32_t *out_category);

/* Waits for everything the context has in flight -- the mapping streams and the staging copy of cm_reads_stage -- and reports
 * device-side capacity flags (CM_ELIMIT).  After it returns no host array handed to an earlier call is read any more. */
int cm_sync(cm_ctx *ctx);

/* Put the resident batch back into its first-round state (fill_map_info's "cnt != 23" state,
 * every pair active) without touching the read bytes: lets a caller re-run all rounds on reads
 * that are already in HBM. */
int cm_reads_reset(cm_ctx *ctx);

/* Compact the pairs that are still active into host buffers: after the last round these are the
 * CHIBSJ / CHI2BSJ pairs that write_read_category hands to stage 2 (src/circminer.cpp:395-397);
 * after an earlier round, the pairs re-queued for the next contig.  Ascending pair index.
 * Returns CM_ELIMIT (and the needed count in *out_n) if cap is too small. */
int cm_collect_active(cm_ctx *ctx, uint64_t cap, uint64_t *out_idx, cm_mapped_read *out_state, uint64_t *out_n);
/* The same pairs as self-contained records (global pair index = index_base + index in the batch, then the state):
 * what a rank hands to the BSJ gather, assembled on the device so the host does no packing. */
typedef struct cm_record {
    uint64_t pair;
    cm_mapped_read state;
} cm_record;
int cm_collect_records(cm_ctx *ctx, uint64_t index_base, uint64_t cap, cm_record *out, uint64_t *out_n);
/* Same records, left in device memory the caller owns (`d_out`: cap * sizeof(cm_record) bytes on ctx's device; complete
 * when the call returns): the multi-GPU hand-off gathers them to rank 0 over RCCL straight from HBM
 * (circminer_amd/dist.py) and only rank 0 copies anything to the host. */
int cm_collect_records_device(cm_ctx *ctx, uint64_t index_base, uint64_t cap, void *d_out, uint64_t *out_n);

/* Page-locked host memory for the buffers that cross PCIe (read batches, downloaded states, collected
 * records): the copies in cm_reads_upload / cm_reads_download / cm_collect_active are direct DMA for such
 * buffers instead of staged pageable copies.  Optional: every entry point also accepts ordinary memory. */
int cm_host_alloc(cm_ctx *ctx, uint64_t bytes, void **out);
/* The same for memory the caller already owns (e.g. the batch arrays cm_fastq_next returns): page-locks [p, p + bytes) until
 * cm_host_unregister(p); the range must stay allocated meanwhile. */
int cm_host_register(cm_ctx *ctx, void *p, uint64_t bytes);
int cm_host_unregister(cm_ctx *ctx, void *p);
/* MatchedRead::type histogram of the resident batch (CM_CONCRD .. CM_NOPROC_NOMATCH), counted on the device. */
int cm_type_histogram(cm_ctx *ctx, uint64_t out[14]);
int cm_host_free(cm_ctx *ctx, void *p);

/* ---------------- finer-grained entry points used by the parity tests ---------------- */
/* Seeds (GenomeSeeder::split_match_hash, src/match_read.cpp:270-286) of the resident batch:
 * for probe q = ((pair*2 + mate)*2 + orient)*n_slots + s (orient 0 = forward, 1 = reverse
 * complement; s = seed ordinal, qpos = s*kmer):
 *   out_start[q] = index of first hit in the contig's entry arrays (valid if cnt>0 or high)
 *   out_cnt[q]   = frag_count after the seedLim rule (0 if > seedLim)
 *   out_raw[q]   = occurrence count before the rule (0 = frags==NULL)
 * n_slots = max over the batch of floor(len/kmer); returned through out_n_slots. */
int cm_seed_batch(cm_ctx *ctx, int slot, uint32_t *out_start, uint32_t *out_cnt, uint32_t *out_raw,
                  uint32_t cap_probes, uint32_t *out_n_slots);
/* Chains (chain_seeds_sorted_kbest, src/chain.cpp:73-301) of the resident batch: problem
 * r = (pair*2 + mate)*2 + orient; out_nchain[r] chains stored at out_chains[r*CM_BESTCHAINLIM ...],
 * out_high[r] = high_hits of get_best_chains (src/filter.cpp:478-481). */
int cm_chain_batch(cm_ctx *ctx, int slot, cm_chain *out_chains, int32_t *out_nchain, int32_t *out_high);

/* ---------------- timing hooks for bench.py (HIP events on the ctx stream) ---------------- */
/* Milliseconds spent in each kernel class since the last cm_prof_reset(), and launch counts:
 * [0]=k_seed [1]=k_chain (light problems) [2]=k_pair (light pairs) [3]=k_scan_* [4]=k_pair_heavy
 * [5]=class / counting-sort kernels [6]=k_chain_heavy [7]=no time; launches[7] = first rounds taken over from the
 * cross-batch prefetch of cm_map_rounds.  [1] and [2] are timed up to the join with the second
 * stream, on which the heavy kernels [6] / [4] run concurrently. */
int cm_prof_enable(cm_ctx *ctx, int on);
int cm_prof_reset(cm_ctx *ctx);
int cm_prof_get(cm_ctx *ctx, double ms[8], uint64_t launches[8]);
/* Algorithmic byte counters of SURVEY §8(d) accumulated by the kernels since cm_prof_reset():
 * [0]=probes [1]=binary-search touches [2]=hits consumed (cnt<=seedLim) [3]=pair-rounds; [4]=pair-rounds that needed the re-run
 * launch (a device capacity the reference does not have, e.g. more than 8 memoised exon pieces in one extension: mapped again
 * with room for 2048, results identical); [5..7] reserved (0). */
int cm_prof_counters(cm_ctx *ctx, uint64_t c[8]);

/* ---------------- host-side builders (stay on host; north_star "index build ... on host") ---- */
/* In-memory equivalent of generateHashTableOnDisk for one contig
 * (src/mrsfast/HashTable.c:257-380,769-839): caller frees with cm_host_free_index. */
int cm_host_build_index(const uint8_t *genome, uint32_t ref_len, int32_t kmer, int32_t contig_num,
                        int n_threads, cm_index_view *out);
void cm_host_free_index(cm_index_view *iv);
/* Hit multiplicity of an index: over all indexed k-mer positions of the contig, how many share their k-mer with at least one
 * other position (out[1]) and with more than seed_lim - 1 others, i.e. a probe of that k-mer returns more than seed_lim hits and
 * the seed is dropped (src/match_read.cpp:231,259-263) (out[2]); out[0] = indexed positions, out[3] = distinct k-mers.  The two
 * fractions out[1] / out[0] and out[2] / out[0] characterise a workload's repeat content. */
int cm_host_index_stats(const cm_index_view *iv, int32_t seed_lim, int n_threads, uint64_t out[4]);

/* GTF -> flattened annotation for every packed contig (GTFParser::load_gtf,
 * src/gene_annotation.cpp:191-399).  chromosome table = rows of .index.info
 * (src/genome.cpp:147-167): name, packed contig id (1-based), start_pos.
 * out must point to n_contigs views; free with cm_host_free_annotation. */
typedef struct cm_chr_info {
    const char *name;
    uint32_t contig_id;   /* 1-based */
    uint32_t start_pos;   /* shift   */
    uint32_t len;
} cm_chr_info;
int cm_host_build_annotation(const char *gtf_path, const cm_chr_info *chrs, uint32_t n_chr,
                             const uint32_t *contig_len, uint32_t n_contigs, int32_t max_read_len,
                             cm_annot_view *out);
void cm_host_free_annotation(cm_annot_view *av, uint32_t n_contigs);
/* get_gene_overlap(pos, use_mask = false): the genes whose span covers contig position pos (n_genes = 0: none) */
int cm_host_gene_overlap(const cm_annot_view *av, uint32_t pos, const uint32_t **genes, uint32_t *n_genes);

/* ---------------- on-disk genome / index formats of stock CircMiner ------ */
/* FASTA -> <ref>.packed.fa + <ref>.packed.fa.index.info (GenomePacker::pack_genome,
 * src/genome.cpp:96-146): chromosomes concatenated with a 50-N spacer while they fit contig_size
 * (DEF_CONTIG_SIZE = CM_CONTIG_SIZE in the reference). */
int cm_host_pack_genome(const char *fasta_path, const char *packed_fa_path, const char *index_info_path,
                        uint32_t contig_size);
/* rows of .index.info (GenomePacker::load_index_info, src/genome.cpp:147-167); names are owned by the
 * array: release with cm_host_free_index_info. */
int cm_host_read_index_info(const char *path, cm_chr_info **out, uint32_t *n);
void cm_host_free_index_info(cm_chr_info *chrs, uint32_t n);
/* packed FASTA -> mrsfast index file: full table (generateHashTableOnDisk, src/mrsfast/HashTable.c:257-380,
 * magic 3) or compact (generateHashTable, :383-474, magic 2: the table is rebuilt at load time). */
int cm_host_write_index(const char *packed_fa_path, const char *index_path, int32_t kmer, int compact,
                        int n_threads);
/* Reader (checkHashTable / initLoadingHashTable / loadHashTable, HashTable.c:485-509, 618-700, 971-1098):
 * open parses the header; every cm_host_next_contig call loads the next packed contig -- genome decoded to
 * ASCII and the table flattened into a cm_index_view ready for cm_load_contig -- and sets *loaded = 0
 * after the last one.  Views are released with cm_host_free_loaded_contig. */
typedef struct cm_index_file cm_index_file;
int cm_host_open_index(const char *index_path, cm_index_file **out, int32_t *kmer, int32_t *is_full,
                       uint32_t *n_records);
int cm_host_next_contig(cm_index_file *f, int n_threads, cm_index_view *out, int *loaded);
/* The same record with its k-mer table stepped over: only genome / ref_len / contig_num of *out are set (ProcessCirc::load_genome,
 * src/process_circ.cpp:1659-1680: loadCompressedRefGenome, the sequence alone); stage 2 uses this. */
/* The same record left as it is in the file, for cm_load_contig_raw: decoded genome, the (hv, count14) pair of every non-empty
 * bucket in file order, and the table itself (GeneralIndex slots of 8 bytes: per bucket a header slot whose info = number of
 * valid entries, then count14 slots).  Full-format index files only (CM_EINVAL for the compact format: take
 * cm_host_next_contig).  The arrays belong to the file handle and stay valid until the call after next (two sets take turns, so
 * the next record can be read while this one uploads) or cm_host_close_index. */
typedef struct cm_index_raw {
    int32_t contig_num;
    uint32_t ref_len;
    const uint8_t *genome;       /* ref_len bytes                                   */
    uint32_t n_buckets;          /* non-empty 14-mer buckets                        */
    const uint32_t *hv;          /* ascending                                       */
    const uint32_t *count14;     /* slots of bucket i = count14[i] + 1              */
    const void *table;           /* table_slots x 8 bytes, as in the file           */
    uint64_t table_slots;
} cm_index_raw;
int cm_host_next_contig_raw(cm_index_file *f, int n_threads, cm_index_raw *out, int *loaded);
int cm_host_next_contig_genome(cm_index_file *f, cm_index_view *out, int *loaded);
void cm_host_free_loaded_contig(cm_index_view *iv);
void cm_host_close_index(cm_index_file *f);

/* ---------------- FASTQ ingest, carry-over header, PAM / remain writers ---------- */
/* One batch of parsed pairs in the layout cm_reads_upload takes.  All pointers belong to the parser and stay
 * valid over the next THREE cm_fastq_next calls (four generations of storage take turns: batch k-1 can be with a writer
 * thread, batch k on the GPU and batch k+1 staged for it while batch k+2 is parsed) or until cm_fastq_close.  names*: NUL-terminated read names (first header token,
 * trailing "/x" cut: FASTQParser::extract_map_info, src/fastq_parser.cpp:178-198), name i at names + name_off[i].
 * prior: the MatchedRead each pair carried in its 23-token header (fill_map_info, :200-269) or NULL when no
 * pair of the batch carried one (fresh reads: cm_reads_upload's default state). */
typedef struct cm_fastq_batch {
    cm_reads reads;
    const uint8_t *qual1, *qual2;        /* same offsets as reads.seq1 / reads.seq2 */
    const char *names1, *names2;
    const uint64_t *name_off1, *name_off2;
    const cm_mapped_read *prior;
} cm_fastq_batch;
typedef struct cm_fastq cm_fastq;
/* plain or gzip FASTQ (gzread, as src/fastq_parser.cpp:84-98); chrs = rows of .index.info (chromosome
 * names of the carried headers -> chr_id); max_ed = maxEd of the run (state of unmapped carried reads). */
int cm_fastq_open(const char *r1_path, const char *r2_path, const cm_chr_info *chrs, uint32_t n_chr, int32_t max_ed,
                  cm_fastq **out);
/* One rank's contiguous block of pairs of a paired FASTQ ("rank r gets pairs [r*N/W, (r+1)*N/W)"): both files
 * are cut at the same record, found by counting the newlines of the files in blocks on n_threads threads (no parsing).
 * gzip files (the reference reads .gz everywhere, src/fastq_parser.cpp:84-98) cannot be entered in the middle: their records are
 * counted with one inflate pass over R1 and every rank inflates from the start, stepping over the blocks of the ranks before it --
 * the same blocks and output bytes as plain text, at zlib's speed.  A pipe cannot be read twice: CM_EINVAL when world > 1.
 * first_pair / n_pairs (nullable): the block's position in the whole input.  Every rank of a node can open its share at the same time. */
int cm_fastq_open_shard(const char *r1_path, const char *r2_path, const cm_chr_info *chrs, uint32_t n_chr, int32_t max_ed,
                        int32_t rank, int32_t world, int n_threads, cm_fastq **out, uint64_t *first_pair, uint64_t *n_pairs);
int cm_fastq_next(cm_fastq *f, uint64_t max_pairs, cm_fastq_batch *out);   /* out->reads.n_pairs == 0 at the end; CM_ENOMEM when a batch array cannot grow */
/* A caller that page-locks the arrays of a batch (cm_host_register on reads.seq1/seq2/off1/off2, prior) must know when one of
 * them goes away: `fn(user, ptr, bytes)` is called -- on whichever thread runs cm_fastq_next / cm_fastq_close -- right BEFORE the
 * block starting at `ptr` is freed (a generation's array that has to grow is allocated anew, never moved in place), so that the
 * caller can unregister it first.  No copy out of that block may still be in flight then: the block belongs to the generation
 * handed out four cm_fastq_next calls ago. */
void cm_fastq_set_release_hook(cm_fastq *f, void (*fn)(void *user, const void *ptr, uint64_t bytes), void *user);
void cm_fastq_close(cm_fastq *f);

/* Writers.  cm_write_remain = FilterRead::write_read_category PE (src/filter.cpp:413-455) into
 * <out>_<round>_remain_R1.fastq / _R2.fastq; cm_write_pam = SAMOutput::write_pam_rec_pe
 * (src/output.cpp:279-299) into <out>.mapping.pam (path2 = NULL).  sel = indices of the pairs to write
 * (NULL: all): map_reads writes PAM rows for (skip || last round) and remain records for the pairs
 * cm_collect_active returns (src/circminer.cpp:386-397). */
typedef struct cm_writer cm_writer;
int cm_writer_open(const char *path1, const char *path2, const cm_chr_info *chrs, uint32_t n_chr, cm_writer **out);
int cm_write_remain(cm_writer *w, const cm_fastq_batch *b, const cm_mapped_read *states, const uint64_t *sel, uint64_t n_sel);
/* cm_write_remain for (pair index, state) records as cm_collect_records delivers them (index_base 0): only the re-queued
 * pairs' states have to leave the device. */
int cm_write_remain_records(cm_writer *w, const cm_fastq_batch *batch, const cm_record *recs, uint64_t n);
int cm_write_pam(cm_writer *w, const cm_fastq_batch *b, const cm_mapped_read *states, const uint64_t *sel, uint64_t n_sel);
/* <out>.mapping.sam (--sam): header = SAMOutput::print_header (src/output.cpp:301-311, one @SQ per row of the chromosome
 * table), records = write_sam_rec_pe with set_flag_pe / set_output_pe (src/output.cpp:118-277): two lines per pair,
 * CIGAR "*", MAPQ 255, tags AT / NM / JC / TC; TLEN is printed with %u like the reference does. */
int cm_write_sam_header(cm_writer *w);
int cm_write_sam(cm_writer *w, const cm_fastq_batch *b, const cm_mapped_read *states, const uint64_t *sel, uint64_t n_sel);
/* Pushes buffered rows to the file(s); CM_EIO if any write since the writer was opened came up short (the cm_write_* calls
 * report the same as soon as they notice).  cm_writer_close flushes too but cannot report. */
int cm_writer_flush(cm_writer *w);
void cm_writer_close(cm_writer *w);

/* ---------------- stage 1 end to end: the caller of the hot path ---------------- */
/* mapping() + map_reads() of the reference (src/circminer.cpp:98-352, :354-400) over the entry points above: reads
 * <ref>.packed.fa.index(.info) and the GTF, keeps every packed contig resident, takes the paired FASTQ through all
 * rounds batch by batch, and writes what the reference leaves behind for the user and for stage 2:
 *   <out>.mapping.pam (report 1) or <out>.mapping.sam (report 2): the rows map_reads prints (skip || last round);
 *   <out>_<R>_remain_R1.fastq / _R2.fastq, R = number of packed contigs: the CHIBSJ / CHI2BSJ pairs (:395-397).
 * The per-round remain files of the reference (its way of carrying pairs from one contig to the next) are not
 * written: the carried state stays in HBM.  params.kmer == 0 takes the index file's k. */
typedef struct cm_mapping_args {
    const char *index_path;        /* <ref>.packed.fa.index        */
    const char *index_info_path;   /* <ref>.packed.fa.index.info   */
    const char *gtf_path;
    const char *fastq1, *fastq2;   /* plain or gzip                */
    const char *out_prefix;        /* outputFilename               */
    cm_params params;
    int32_t report;                /* reportMapping: 0 none, 1 PAM, 2 SAM */
    int32_t n_threads;             /* host threads for the index loader   */
    uint64_t batch_pairs;          /* pairs per resident batch, 0 = 2^18  */
    /* One process per GPU: rank `rank` of `world` maps the rank-th contiguous block of pairs
     * (cm_fastq_open_shard) on device params.device and writes <file>.part<rank> instead of <file> for every output;
     * cm_merge_parts on rank 0 then makes the files of an unsharded run (same bytes: blocks are contiguous, rows are in
     * input order).  world 0 or 1: one process, final file names. */
    int32_t rank, world;
} cm_mapping_args;
typedef struct cm_mapping_stats {
    uint64_t pairs, bsj_pairs;
    uint64_t by_type[14];          /* final MatchedRead::type histogram (CM_CONCRD ... ) */
    int32_t rounds, reserved;
    double seconds_load, seconds_map;
    /* where seconds_map went, summed over the batches (the three overlap, so they add up to more than seconds_map):
     * parsing FASTQ (parser thread), driving the device (stage + rounds + results + swap, calling thread), writing rows
     * (writer thread) */
    double seconds_parse, seconds_device, seconds_write;
} cm_mapping_stats;
int cm_mapping_run(const cm_mapping_args *args, cm_mapping_stats *stats, char *err, uint64_t err_cap);
/* After every rank's cm_mapping_run(rank, world) has returned: <out>_<rounds>_remain_R{1,2}.fastq and the mapping file
 * (report 1 / 2) are concatenated from their .part<r> files in rank order, the parts are removed.  The result is what one
 * process writes for the whole input; cm_circ_run takes it from there (stage 2 runs once, on the host of rank 0). */
int cm_merge_parts(const char *out_prefix, int32_t rounds, int32_t world, int32_t report);

/* ---------------- stage 2: ProcessCirc, src/process_circ.cpp -- host code, no GPU involved -------- */
/* ProcessCirc::sort_fq (src/process_circ.cpp:179-193): the remain FASTQ of the last round ordered like
 * `paste - - - - | sort -k2,2n | tr "\t" "\n"` does in the C locale (key = genome_spos, ties by the whole pasted line). */
int cm_sort_remain(const char *in_path, const char *out_path);
/* RegionalHashTable::create_table (src/hash_table.cpp:58-78) flattened: the window_size-mers of a gene region, bucket hv =
 * loc[off[hv] .. off[hv+1]) ascending, location = start + offset in seq; buckets above MAXHIT = 1000 are emptied. */
int cm_regional_table_build(const uint8_t *seq, uint32_t start, int32_t len, int32_t window_size, uint32_t **off, uint32_t **loc);
void cm_regional_table_free(uint32_t *off, uint32_t *loc);
/* ProcessCirc::report_events (src/process_circ.cpp:1570-1631): the BSJ calls of stage 2 (CircRes, src/common.h:406-423;
 * type 20 = CR, 21 = NCR, 22 = MCR, src/process_circ.h:16-18) -> <out>.circ_report rows
 * chr, start, end, read count, "STC", consensus start-end signal, reference start-end signal, Pass|Fail, read names. */
typedef struct cm_circ_res {
    const char *chr, *rname;
    uint32_t spos, epos;
    int32_t type, reserved;
    const char *start_signal, *end_signal, *start_bp_ref, *end_bp_ref;
} cm_circ_res;
int cm_circ_report(const cm_circ_res *res, uint64_t n, const char *report_path);
/* The back-splice-junction calling between the two (ProcessCirc::do_process, src/process_circ.cpp:195-331: call_circ_single_split /
 * call_circ_double_split :360-645, chaining of 8-mer seeds inside the overlapping genes :677-737 + src/chain.cpp:310-539,
 * find_exact_coord :739-789, check_split_map :892-1134, final_check :1136-1341, split_realignment :1343-1486,
 * rescue_overlapping_bsj :1488-1552) over records that are already in the order of the sorted remain files: `sorted` is one
 * batch from cm_fastq_next on <out>_<R>_remain_R{1,2}.fastq.srt (its `prior` = the MatchedRead each pair carries in its header).
 * Writes <out>.candidates.pam rows (print_split_mapping :1670-1708 + type) to candidates_path and the <out>.circ_report rows
 * (report_events) to report_path.  window_size 0 = 8 (circ_detect, src/circminer.cpp:347-352).  Pairs are called on all host
 * cores (CM_CIRC_THREADS overrides; cm_circ_run uses its n_threads); the files do not depend on the number of threads. */
typedef struct cm_circ_stats {
    uint64_t pairs, candidate_rows, calls;
    double seconds;
} cm_circ_stats;
int cm_circ_call(const cm_params *p, int32_t window_size, uint32_t n_contigs, const cm_index_view *contigs, const cm_annot_view *annots,
                 const cm_chr_info *chrs, uint32_t n_chr, const cm_fastq_batch *sorted, const char *candidates_path,
                 const char *report_path, cm_circ_stats *stats);
/* circ_detect() of the reference (src/circminer.cpp:347-352) from files to files: sorts <out>_<last_round>_remain_R{1,2}.fastq
 * (-> .srt), loads the packed genome from the index file and the GTF, calls the junctions, writes <out>.candidates.pam and
 * <out>.circ_report.  params.kmer == 0 takes the index file's k. */
typedef struct cm_circ_args {
    const char *index_path, *index_info_path, *gtf_path, *out_prefix;
    cm_params params;
    int32_t last_round;            /* number of packed contigs = suffix of the remain files */
    int32_t window_size;           /* 0 = 8 */
    int32_t n_threads, reserved;
} cm_circ_args;
int cm_circ_run(const cm_circ_args *args, cm_circ_stats *stats, char *err, uint64_t err_cap);

/* sizeof of the structs above, in this order: cm_params, cm_index_view, cm_annot_view, cm_mapped_read, cm_reads, cm_record,
 * cm_chr_info, cm_fastq_batch, cm_mapping_args, cm_mapping_stats, cm_circ_res, cm_circ_args, cm_circ_stats, cm_index_raw -- for a binding to
 * check its mirrors of them against the library it loaded.  Returns the number of entries written (cap must hold them). */
int cm_abi_sizes(uint32_t *out, uint32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* CIRCMINER_HOT_H */
