/*
 * floria_hip.h — C ABI of libfloria_hip.so: the MI355X (gfx950) replacement for
 * floria's per-SNP-block read->haplotype clustering path.
 *
 * The reference (bluenote-1577/floria, Rust) has no FFI/plugin interface; this header cuts the
 * seam at the two Rust call sites a drop-in must satisfy:
 *
 *   S1  graph_processing.rs:345-362  the rayon par-for over blocks calling
 *       get_local_hap_blocks(all_frags, snp_to_genome_pos, dir, j, snp_range_vec, options)
 *       (graph_processing.rs:103-110)            -> floria_hip_phase_blocks()
 *   S2  floria.rs:359-366 calling part_block_manip::process_reads_for_final_parts(parts,
 *       short_frags, ranges, options, snp_to_gn) (part_block_manip.rs:174-180)
 *                                                -> floria_hip_reassign()
 *   plus utils_frags::get_range_with_lengths (utils_frags.rs:405-463), the host-side function
 *   that defines the work units (blocks)        -> floria_hip_block_ranges()
 *
 * Conventions
 *   - plain pointers and sizes only; every input is caller-owned and only read during the call;
 *   - outputs are library-owned and released with the matching *_free();
 *   - every entry point returns 0 on success, <0 on error (FLORIA_E_*); the message of the last
 *     error on the calling thread is floria_hip_last_error(); nothing throws across the ABI;
 *   - a context is bound to one device; one host thread at a time per context;
 *   - there is NO CPU fallback: if no gfx950 device / HIP runtime is usable, create() fails.
 *
 * SNP positions are floria's 1-based SNP indices (types_structs.rs:12 `SnpPosition = u32`),
 * reads are `Frag`s (types_structs.rs:68-85) flattened to CSR and sorted by `Frag::cmp`
 * (types_structs.rs:87-93: first_position asc, last_position desc, counter_id asc) with
 * counter_id == index (floria.rs:289-293).
 */
#ifndef FLORIA_HIP_H
#define FLORIA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLORIA_OK            0
#define FLORIA_E_INVALID    -1   /* malformed argument / pileup violates an invariant            */
#define FLORIA_E_DEVICE     -2   /* HIP runtime / device error (message has the hipError string) */
#define FLORIA_E_NOMEM      -3   /* host or device allocation failed                             */
#define FLORIA_E_UNSUPPORTED -4  /* input outside the supported envelope (e.g. allele index > 3) */

#define FLORIA_MAX_ALLELES   4   /* 2-bit allele codes: 0 = ref, 1..3 = alt (file_reader.rs:702-710) */
#define FLORIA_MAX_PLOIDY    16

/* One contig's reads as a sparse reads x SNPs pileup (CSR).  Replaces `&Vec<Frag>`
 * (types_structs.rs:68-85): seq_dict -> (snp, allele), qual_dict -> qual, first/last_position.
 * Cells of a read are sorted by ascending snp; snp[read_off[r]] == first[r] and
 * snp[read_off[r+1]-1] == last[r]; every read has >= 1 cell. */
typedef struct {
    const uint32_t* read_off;  /* [n_reads+1] cell offsets                      */
    const uint32_t* snp;       /* [n_cells]   1-based SNP index of the cell     */
    const uint8_t*  allele;    /* [n_cells]   allele index 0..3                 */
    const uint8_t*  qual;      /* [n_cells]   base quality (phred, 0..255)      */
    const uint32_t* first;     /* [n_reads]   first_position                    */
    const uint32_t* last;      /* [n_reads]   last_position                     */
    uint32_t        n_reads;
    /* OPTIONAL (NULL = not given): the iteration order of every read's `positions` set — `for pos in r.positions.iter()`, utils_frags.rs:35 — for the
     * reference-arithmetic mode (floria_hip_set_option "arith" = 1), which adds a read's terms in that order.  set_order[read_off[r] + j] = index within
     * read r (0-based, in the ascending-SNP order of its cells) of the cell whose position the set yields j-th: a permutation of 0 .. L_r - 1 per read
     * (checked on the device; FLORIA_E_INVALID otherwise).  A Rust host writes it down while it marshals a Frag (it HAS the FxHashSet: whatever built it —
     * one CIGAR walk, `positions.extend(mate.positions)` of combine_frags, file_reader.rs:539-541 and :636-639, the removals of --ignore-monomorphic,
     * utils_frags.rs:745-755 — is in that order by construction).  With NULL the library emulates the set of ONE CIGAR walk (seq_dict.keys().collect(),
     * file_reader.rs:729-733; csrc/arith_kernel.h), which is what every fragment built from a single alignment has.  Ignored in arithmetic mode 0. */
    const uint32_t* set_order; /* [n_cells] or NULL */
} floria_pileup;

/* The same pileup in the compact wire form SURVEY.md §8(d) counts (per read: first, last; a presence bit per SNP of its span; a 2-bit
 * allele and one quality byte per observed SNP): 1.375 B per cell + 16 B per read instead of 6 B per cell + 12 B per read — what a
 * host marshals its `Vec<Frag>` into for the PCIe link (floria.rs:255-293); the device expands it to the CSR form above and then
 * validates + flattens that exactly as if it had been uploaded (upload_kernel.h: expand_kernel, flatten_kernel).
 *   present : read r owns bits [bit_off[r], bit_off[r+1]) with bit_off[r+1] - bit_off[r] == last[r] - first[r] + 1; bit j of the read
 *             (bit (bit_off[r]+j) & 7 of byte (bit_off[r]+j) >> 3, LSB first) is set iff SNP first[r] + j carries a call; the first and
 *             the last bit of a read are set and the number of set bits equals read_off[r+1] - read_off[r];
 *   allele2 : cell c (global cell index of the contig, as in read_off) at bits 2 (c & 3) of byte c >> 2; allele indices 0..3;
 *   qual    : one byte per cell. */
typedef struct {
    const uint32_t* read_off;  /* [n_reads+1] cell offsets                              */
    const uint32_t* first;     /* [n_reads]   first_position                            */
    const uint32_t* last;      /* [n_reads]   last_position                             */
    const uint32_t* bit_off;   /* [n_reads+1] offsets into `present`, in bits (< 2^32)  */
    const uint8_t*  present;   /* [ceil(bit_off[n_reads] / 8)]                          */
    const uint8_t*  allele2;   /* [ceil(n_cells / 4)]                                   */
    const uint8_t*  qual;      /* [n_cells]                                             */
    uint32_t        n_reads;
    const uint32_t* set_order; /* [n_cells] or NULL: as in floria_pileup (travels as it is, 4 B per cell, only when given) */
} floria_pileup_packed;

/* The fields of `Options` (types_structs.rs:20-51) the hot path reads
 * (graph_processing.rs:111-113,198-226,234). */
typedef struct {
    double   epsilon;             /* -e  */
    uint32_t max_ploidy;          /* -p  (default 5)  */
    uint32_t beam;                /* -n  max_number_solns (default 10) */
    uint32_t ploidy_sensitivity;  /* -s  1|2|3 (default 2) */
    int32_t  stopping_heuristic;  /* !--no-stop-heuristic (default 1) */
} floria_params;

/* Result of S1 for a batch of blocks (what get_local_hap_blocks returns, minus the HapNode
 * wrappers the host rebuilds with HapNode::new, types_structs.rs:169).  Block b has
 * n_b = read_off[b+1]-read_off[b] reads (0 <=> the reference returns None); read_id is ascending;
 * part[i] in [0,best_ploidy) is the partition (haplotype) of read_id[i] at the chosen ploidy;
 * mec[b*max_ploidy + p-1] is mec_vector[p-1] (graph_processing.rs:156-162), 0 for untried p. */
typedef struct {
    uint32_t  n_blocks;
    uint32_t  max_ploidy;
    uint32_t* best_ploidy;     /* [n_blocks]  0 if the block holds no reads */
    uint32_t* ploidies_tried;  /* [n_blocks]  last ploidy evaluated before the stop rule fired */
    uint64_t* read_off;        /* [n_blocks+1] */
    uint32_t* read_id;         /* [read_off[n_blocks]] */
    uint8_t*  part;            /* [read_off[n_blocks]] */
    double*   mec;             /* [n_blocks*max_ploidy] */
    double    min_prune_margin;/* min |p_k - lse - ln(PROB_CUTOFF)| over all pruning decisions
                                  (global_clustering.rs:98); parity certificate, see DESIGN.md */
    uint64_t  batch_token;     /* identifies the device-resident copy of this batch (floria_hip_hap_graph) */
} floria_block_result;

/* Hap-graph columns of the batch: HapNode::new's coverage statistic (types_structs.rs:179-193) for
 * every node (block b has best_ploidy[b] nodes) and update_hap_graph's out_weights (graph_processing.rs:28-47) for every
 * consecutive pair of non-empty blocks of a contig, BEFORE the >= MIN_SHARED_READS_UNAMBIG (2.0) filter of :51. */
typedef struct {
    uint32_t  n_blocks;
    uint64_t* node_off;        /* [n_blocks+1] */
    double*   node_cov;        /* [node_off[n_blocks]] */
    int32_t*  pred;            /* [n_blocks] previous non-empty block of the same contig, -1 if none */
    uint64_t* edge_off;        /* [n_blocks+1]; block b holds best_ploidy[pred[b]] x best_ploidy[b] counts, row-major */
    uint32_t* edge_w;          /* [edge_off[n_blocks]] */
} floria_hap_graph;

/* Haplogroups (S2 in/out): group g holds reads grp_read[grp_off[g]..grp_off[g+1]) and spans the
 * inclusive SNP range (range[2g], range[2g+1]).  Output groups are sorted by range
 * (part_block_manip.rs:276-288) and read ids ascend within a group. */
typedef struct {
    uint32_t  n_groups;
    uint64_t* grp_off;         /* [n_groups+1] */
    uint32_t* grp_read;        /* [grp_off[n_groups]] */
    uint32_t* range;           /* [2*n_groups] */
} floria_groups;

typedef struct {
    uint32_t  n;
    uint32_t* start;           /* [n] 1-based inclusive */
    uint32_t* end;             /* [n] 1-based inclusive */
} floria_ranges;

/* Per-kernel timing of the last phase_blocks / reassign call, measured with hipEvents on the
 * context's stream (bench.py's roofline leg reads this). */
typedef struct {
    double   beam_ms;          /* sum over launches of the beam-search kernel     */
    double   optimize_ms;      /* sum over launches of the optimise/MEC kernel    */
    double   select_ms;        /* stop-rule + gather kernels                      */
    double   reassign_ms;      /* S2 kernel                                       */
    double   h2d_ms, d2h_ms;   /* copies issued by the call                       */
    double   total_ms;         /* first launch -> last completion on the stream   */
    uint32_t beam_launches, optimize_launches;
    uint64_t algorithmic_bytes;/* SURVEY.md §8(d) bytes(block) summed over the call's blocks */
    uint64_t beam_steps;       /* reads consumed by beam search, summed over (block, ploidy) jobs */
    uint64_t beam_launch_bytes;/* sum over beam launches of bytes(block) of the blocks that launch phased */
    uint64_t jobs;             /* (block, ploidy) jobs actually run */
    uint32_t streams;          /* job groups of the last S1 call: their launch triples overlap on separate streams, so the
                                * per-kernel sums above can exceed phase_ms */
    uint32_t stage_width;      /* ploidies run concurrently per stage (1 = the reference's sequential ploidy loop, no speculative jobs) */
    double   phase_ms;         /* wall time of the per-ploidy launch loop (fork of the first group -> join of the last) */
    uint64_t upload_pinned_bytes; /* last upload: bytes that went by DMA straight from the caller's pinned memory */
    uint64_t upload_staged_bytes; /* last upload: bytes staged through the library's pinned ring (pageable sources)  */
    uint32_t upload_chunks;    /* floria_hip_phase_pileups_batch: chunks whose transfer overlapped the kernels (1 = not pipelined) */
    uint32_t reserved;
    double   beam_union_ms;    /* wall time with at least one beam-search launch in flight (<= phase_ms; beam_ms, a sum over overlapping launches, can exceed it) */
    double   optimize_union_ms;/* ... with at least one optimise launch in flight */
} floria_timing;

typedef struct floria_hip_ctx floria_hip_ctx;
typedef struct floria_hip_contig floria_hip_contig;   /* a pileup resident in HBM */

int  floria_hip_create(int device, floria_hip_ctx** out);
void floria_hip_destroy(floria_hip_ctx* ctx);
const char* floria_hip_last_error(void);
const char* floria_hip_version(void);
/* Step 0 for a host that links this library: call ONCE, from the main thread, BEFORE the process's first HIP call (floria_hip_create included).  It sets
 * GPU_MAX_HW_QUEUES=12 unless the variable is already set: HIP reads it once, when the runtime initialises, and maps streams onto that many hardware queues
 * (default 4); the job groups of S1, the upload pipeline and the speculative ploidy stages want their streams on separate queues.  Exporting the variable in the
 * process environment does the same.  Without it everything still works and returns the same results: floria_hip_create measures how many streams really run
 * side by side, keeps its launch plans within that and says so once on stderr (about 25 % less throughput from host memory).  The library itself never touches
 * the environment (setenv is not safe beside threads that read it). */
int  floria_hip_init_env(void);

/* utils_frags::get_range_with_lengths (utils_frags.rs:405-463), host side, exact restatement.
 * snp_to_genome_pos has n_snps entries (0-based SNP index -> bp).  Fails (FLORIA_E_INVALID) where
 * the reference exits (positions not increasing, utils_frags.rs:424-427). */
int  floria_hip_block_ranges(const uint64_t* snp_to_genome_pos, uint32_t n_snps,
                             uint64_t block_length, uint64_t overlap_len, double minimal_density,
                             floria_ranges** out);
void floria_hip_ranges_free(floria_ranges* r);

/* Upload one contig's pileup to HBM.  The invariants above are validated ON THE DEVICE while the pileup is flattened into
 * its resident form (allele | Q24 weight per cell, per-read hash constants and metadata records): the host touches no cell.
 * The handle can be phased any number of times. */
int  floria_hip_contig_upload(floria_hip_ctx* ctx, const floria_pileup* pileup, floria_hip_contig** out);
void floria_hip_contig_free(floria_hip_contig* c);

/* Upload n contigs in one go (what a host does once per batch of contigs it marshals from `Vec<Frag>`, floria.rs:255-293):
 * one device allocation, the raw arrays by DMA (arrays of consecutive contigs that are back to back in host memory travel as
 * ONE transfer; sources in floria_hip_host_alloc memory are not staged), one validate + flatten launch, one synchronisation.
 * out[0..n) receive the handles (each freed with floria_hip_contig_free); on error no handle is returned. */
int  floria_hip_contig_upload_batch(floria_hip_ctx* ctx, const floria_pileup* pileups, uint32_t n, floria_hip_contig** out);

/* Diagnostic: copy one resident array of a contig back to the host.  CELL_AW = allele << 28 | Q24 weight per cell
 * (phred_scale, utils_frags.rs:702-711), TW = the two per-read constants of the linear state hash, META = the packed per-read
 * record {cell offset, cell count, first, last, tw1 lo/hi, tw2 lo/hi}. */
#define FLORIA_FIELD_READ_OFF 0
#define FLORIA_FIELD_FIRST    1
#define FLORIA_FIELD_LAST     2
#define FLORIA_FIELD_SNP      3
#define FLORIA_FIELD_CELL_AW  4
#define FLORIA_FIELD_TW       5
#define FLORIA_FIELD_META     6
int  floria_hip_contig_download(const floria_hip_contig* c, int field, void* dst, size_t bytes);

/* Pinned host memory for pileup arrays (NULL on failure). */
void* floria_hip_host_alloc(size_t bytes);
void  floria_hip_host_free(void* p);

/* S1: phase n_blocks SNP ranges of one resident contig. */
int  floria_hip_phase_blocks_resident(floria_hip_ctx* ctx, const floria_hip_contig* contig,
                                      const uint32_t* blk_start, const uint32_t* blk_end,
                                      uint32_t n_blocks, const floria_params* params,
                                      floria_block_result** out);
/* S1 convenience: upload + phase + free. */
int  floria_hip_phase_blocks(floria_hip_ctx* ctx, const floria_pileup* pileup,
                             const uint32_t* blk_start, const uint32_t* blk_end, uint32_t n_blocks,
                             const floria_params* params, floria_block_result** out);
void floria_hip_block_result_free(floria_block_result* r);

/* S1 straight from HOST pileups for many contigs: upload_batch + phase_blocks_batch as one pipelined call — the timed region of
 * the reference's "Phasing time taken" span (floria.rs:330-337) with the pileups in host memory.  The cell arrays travel in
 * chunks of consecutive contigs and the blocks of a chunk start phasing when the chunk has landed, so most of the PCIe time
 * hides behind the kernels (needs sources in floria_hip_host_alloc memory; pageable sources are uploaded first, then phased).
 * keep == NULL: nothing stays resident.  keep != NULL: keep[0..n_contigs) receive the resident handles (floria_hip_contig_free
 * each), e.g. for floria_hip_hap_graph / S2 on the same batch.  Results are identical to upload + phase_blocks_batch. */
int  floria_hip_phase_pileups_batch(floria_hip_ctx* ctx, const floria_pileup* pileups, uint32_t n_contigs,
                                    const uint32_t* blk_contig, const uint32_t* blk_start, const uint32_t* blk_end,
                                    uint32_t n_blocks, const floria_params* params, floria_block_result** out,
                                    floria_hip_contig** keep);

/* The compact wire form: bytes a packed copy of `in` needs (0 if `in` is malformed on its face: null field, first > last), and the
 * packing itself into caller memory (floria_hip_host_alloc memory for a DMA upload): *out receives views into buf.  Host-side, exact. */
size_t floria_hip_pack_bytes(const floria_pileup* in);
int    floria_hip_pack_pileup(const floria_pileup* in, void* buf, size_t buf_bytes, floria_pileup_packed* out);
/* The same for a batch of contigs, laid out field by field (all read_off arrays back to back, then all bit_off, ...): an upload of the batch
 * then is one transfer per field and chunk.  out[0..n) receive the views. */
size_t floria_hip_pack_bytes_batch(const floria_pileup* in, uint32_t n);
int    floria_hip_pack_pileups_batch(const floria_pileup* in, uint32_t n, void* buf, size_t buf_bytes, floria_pileup_packed* out);
/* floria_hip_contig_upload_batch / floria_hip_phase_pileups_batch from the compact form: same handles, same results, a fifth of the bytes
 * on the PCIe link. */
int  floria_hip_contig_upload_batch_packed(floria_hip_ctx* ctx, const floria_pileup_packed* pileups, uint32_t n, floria_hip_contig** out);
int  floria_hip_phase_pileups_batch_packed(floria_hip_ctx* ctx, const floria_pileup_packed* pileups, uint32_t n_contigs,
                                           const uint32_t* blk_contig, const uint32_t* blk_start, const uint32_t* blk_end,
                                           uint32_t n_blocks, const floria_params* params, floria_block_result** out,
                                           floria_hip_contig** keep);

/* S1 over MANY contigs in one launch sequence (the unit bench.py times: all blocks of all
 * contigs a rank owns are phased together so the device sees >> 256 concurrent jobs).
 * blk_contig[b] indexes `contigs`; results are in block order. */
int  floria_hip_phase_blocks_batch(floria_hip_ctx* ctx, const floria_hip_contig* const* contigs,
                                   uint32_t n_contigs, const uint32_t* blk_contig,
                                   const uint32_t* blk_start, const uint32_t* blk_end,
                                   uint32_t n_blocks, const floria_params* params,
                                   floria_block_result** out);

/* S2: process_reads_for_final_parts (part_block_manip.rs:174-274) with reassign_short = false
 * (the CLI default; the short-read branch :235-270 sits behind a hidden flag). */
int  floria_hip_reassign(floria_hip_ctx* ctx, const floria_hip_contig* contig,
                         const uint64_t* grp_off, const uint32_t* grp_read,
                         const uint32_t* grp_range, uint32_t n_groups, double epsilon,
                         floria_groups** out);
void floria_hip_groups_free(floria_groups* g);

/* S2 for many contigs in one launch (one wavefront per contig; the chain is sequential inside a contig and independent
 * across contigs).  grp_contig[g] indexes `contigs`; *out is an array of n_contigs floria_groups* in contig order. */
int  floria_hip_reassign_batch(floria_hip_ctx* ctx, const floria_hip_contig* const* contigs, uint32_t n_contigs,
                               const uint32_t* grp_contig, const uint64_t* grp_off, const uint32_t* grp_read,
                               const uint32_t* grp_range, uint32_t n_groups,
                               const uint32_t* read_order, const uint64_t* order_off,   /* both NULL, or see below */
                               double epsilon, floria_groups*** out);
/* The greedy re-insertion is strongly order-dependent and the reference visits reads in the iteration order of its
 * `read_to_parts_map: FxHashMap<&Frag, _>` (part_block_manip.rs:203).  A host that wants the reference's exact result passes
 * that order: read_order[order_off[c] .. order_off[c+1]) = counter_ids of contig c's reads in visiting order (every read that
 * sits in a group exactly once).  With NULL the reads are visited in ascending counter_id. */
int  floria_hip_reassign_ordered(floria_hip_ctx* ctx, const floria_hip_contig* contig,
                                 const uint64_t* grp_off, const uint32_t* grp_read, const uint32_t* grp_range, uint32_t n_groups,
                                 const uint32_t* read_order, uint32_t n_order, double epsilon, floria_groups** out);
void floria_hip_groups_array_free(floria_groups** arr, uint32_t n_contigs);

/* Nodes and edges of the hap graph for the batch `res` came from.  Must be called on the same context directly after the
 * floria_hip_phase_blocks* call that produced `res` (the block lists and partitions are still resident in HBM);
 * returns FLORIA_E_INVALID if the resident copy has been overwritten by a later call. */
int  floria_hip_hap_graph(floria_hip_ctx* ctx, const floria_block_result* res, floria_hap_graph** out);
void floria_hip_hap_graph_free(floria_hap_graph* g);

/* Coverage / error statistics of haplosets (utils_frags::get_errors_cov_from_frags, utils_frags.rs:596-655 — the COV and ERR
 * fields of the vartig / haploset headers): out4[4g..4g+3] = (cov, err, total_err, total_cov) of group g over its inclusive SNP
 * range.  err of an empty haploset is NaN, as in the reference (0/0). */
int  floria_hip_haploset_stats(floria_hip_ctx* ctx, const floria_hip_contig* const* contigs, uint32_t n_contigs,
                               const uint32_t* grp_contig, const uint64_t* grp_off, const uint32_t* grp_read,
                               const uint32_t* grp_range, uint32_t n_groups, double* out4);

/* part_block_manip::get_hapq (part_block_manip.rs:517-616 — the HAPQ and REL_ERR fields of the vartig / haploset headers and the
 * avg_err column of the contig ploidy table) for the haplosets of one contig: groups are read-id lists (counter_ids) with their
 * inclusive 1-based SNP ranges; snp_to_genome_pos[s-1] is the base position of SNP s; block_length is the run's -l.
 * hapq[g] in 0..60, rel_err[g] = err_g / avg_err (NaN / inf exactly where the reference's f64 division gives them). */
int  floria_hip_hapq(floria_hip_ctx* ctx, const floria_hip_contig* contig,
                     const uint64_t* grp_off, const uint32_t* grp_read, const uint32_t* grp_range, uint32_t n_groups,
                     const uint64_t* snp_to_genome_pos, uint32_t n_snps, uint64_t block_length,
                     uint8_t* hapq, double* rel_err, double* avg_err);

/* The same for the haplosets of many contigs in one call (grp_contig[g] indexes `contigs`, pairs are searched inside a contig,
 * snp_to_genome_pos[c] / n_snps[c] / avg_err[c] are per contig): three launches for the whole run instead of three per contig. */
int  floria_hip_hapq_batch(floria_hip_ctx* ctx, const floria_hip_contig* const* contigs, uint32_t n_contigs, const uint32_t* grp_contig,
                           const uint64_t* grp_off, const uint32_t* grp_read, const uint32_t* grp_range, uint32_t n_groups,
                           const uint64_t* const* snp_to_genome_pos, const uint32_t* n_snps, uint64_t block_length,
                           uint8_t* hapq, double* rel_err, double* avg_err);

/* alignment::realign (alignment.rs:7-64) for many SNP calls at once: call i has the read's 32 bases around the SNP (read_windows[32 i ..],
 * A C G T, the SNP in column 16), the reference's 32 bases (ref_windows, upper case) and n_alleles[i] candidate bases (alleles[4 i ..], the
 * record's REF and ALTs in order).  best[i] = index of the FIRST allele whose window, with the allele in column 16, has the maximal global
 * alignment score against the read window (match +1, mismatch -1, gap open -2, extend -1).  score (may be NULL) receives that score.
 * The reference scores with block-aligner (an adaptive-band approximation); this is the exact affine-gap DP. */
int  floria_hip_realign(floria_hip_ctx* ctx, const uint8_t* read_windows, const uint8_t* ref_windows, const uint8_t* alleles,
                        const uint8_t* n_alleles, uint64_t n, uint8_t* best, int32_t* score);

int  floria_hip_last_timing(const floria_hip_ctx* ctx, floria_timing* out);

/* Self-test of a hardware assumption: the beam kernel screens the pruning test (global_clustering.rs:98) with an f32 evaluation of stable_binom_cdf_p_rev
 * (utils_frags.rs:211-248) built on the hardware reciprocal and log2, and falls back to the exact host-libm table wherever the screen's error bound leaves a
 * decision open.  *max_err_per_n = max over n <= n_max, k <= n of |screen(n, k) - table(n, k)| / n on THIS device; the kernel assumes <= 2e-5. */
int  floria_hip_selftest(floria_hip_ctx* ctx, double epsilon, uint32_t n_max, double* max_err_per_n);

/* Tuning knob (0 = default): how many (block, ploidy) jobs may be resident at once. */
int  floria_hip_set_slots(floria_hip_ctx* ctx, uint32_t beam_slots);

/* The one option that selects WHICH function is computed: "arith".
 *   0 (default)  every weighted sum is carried as an exact (Q24 integer, number of epsilon terms) pair and rounded once;
 *   1            the reference's own arithmetic: running f64 sums, `diff += epsilon` between `diff += w` over a read's cells in the iteration order of its
 *                FxHashSet of positions (utils_frags.rs:33-72), one running sum per partition in a search node (global_clustering.rs:196-202), `errors +=`
 *                over a haplotype's positions in the bucket order of its FxHashMap (local_clustering.rs:226-256), those orders emulated on the device
 *                (csrc/arith_kernel.h; a pileup may instead CARRY its reads' set orders: floria_pileup::set_order, for fragments merged from several alignments).
 *                Applies to floria_hip_phase_* (S1) and floria_hip_reassign* (S2).  Every pileup - biallelic or not, with or without q = 0 cells - runs the
 *                shared-slab beam kernels with the running sums (beam_slab_kernel<.., ARITH>: terms folded per live slab; wide beams, ploidy * beam > 63:
 *                beam_wide_kernel<.., ARITH>, one lane per live slab, round 6); the optimise kernel lists a partition's position map in bucket order straight
 *                from the histogram where its keys span fewer positions than the map has buckets (every key then sits in its home bucket) and replays the
 *                insertions otherwise: about 1.5 x the time of mode 0 on BASELINE config 4 (97 against 66 ms at -e 0.04), 1.8 x on config 5; the host-pileup
 *                entry points pipeline in this mode too (every chunk's cell orders behind its flatten launch; two chunks by default).
 * For an epsilon that is a multiple of 2^-10 both modes return the same bits (every sum is exact in f64 in any order); for any other epsilon they are
 * different functions (about 60 % of the blocks of the BASELINE configs come out differently at 0.04) and mode 1 is the one a Rust host's CPU path computes,
 * as far as the emulated std hash-table orders are right.  Checked bit for bit against the oracle's arithmetic mode 1 (tests/test_gpu_arith.py).
 *
 * "s2_assign_only" = 1: floria_hip_reassign* stop behind the greedy re-insertion (part_block_manip.rs:203-222) and return the haplogroups as re-inserted —
 * input group order, input ranges, reads ascending — WITHOUT separate_broken_haplogroups (:27-98) and sort_parts (:276-288).  The read a split drops is the
 * first one behind a coverage gap in the iteration order of the reference's FxHashSet among reads that share a first_position; with the default (0) that
 * order is ascending counter_id, and which read goes changes the output of every short-read contig measured (scripts/a14_sensitivity.py; no long-read
 * one).  A Rust host that wants its own sets' order inserts the returned reads into its sets and keeps calling its own two functions: they are integer
 * bookkeeping, a few microseconds per contig.
 *
 * Tuning / test knobs; none changes results.  Keys: "groups" (job groups on separate streams, 0 = auto), "speculate" (ploidy stages: -1 auto,
 * 0 one ploidy at a time, 1 all ploidies of a block at once, 2 {1,2,3} then {4..P}), "spec_gate_div" (grid divisor of the gated ploidies of a
 * speculative stage, default 2), "tail_overlap" (one ploidy per stage: the beam launch of the LAST ploidy runs beside the optimise launch of the
 * ploidy below, every job waiting for its block's stop rule; 0 off = default (measured level with it on BASELINE config 4), 1 on), "tail_waves" (waves per CU of that launch,
 * default 2), "beam_path" (0 auto, 1 generic, 2 slab, 3 wide), "no_specialized", "no_p1_shortcut", "opt_threads"
 * (0|128|512|1024), "opt_global", "opt_block_order" (tests / A/B: the optimise kernel's passes visit a block's reads in block order instead of longest first), "slots", "stage_threads" (host threads that fill the pinned staging ring of a pageable upload),
 * "upload_chunks" (chunks of floria_hip_phase_pileups_batch, 0 = auto), "trace" (host-side timestamps of an S1 call on stderr), "reassign_path", "arith_hbm" (tests: the reference-arithmetic mode's tables in HBM scratch even where they fit into LDS), "arith_replay" (tests: that mode replays every position map insertion by insertion instead of listing it by the home-bucket rule where that applies), "fx_tags" / "arith_ow6" (A/B: size of the replay's claim table, the six-wave optimise instance at every ploidy), "no_bulk" (tests: every beam step through the general
 * insert path with its duplicate test), "hw_queues" (tests: override the number of
 * concurrently running streams floria_hip_create measured — 6 on an MI355X whose host set GPU_MAX_HW_QUEUES=12 before HIP initialised, 4 or fewer
 * with the runtime's default; below 5 the launch plans stay within two job groups and do not speculate). */
int  floria_hip_set_option(floria_hip_ctx* ctx, const char* key, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* FLORIA_HIP_H */
