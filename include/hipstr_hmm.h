/*
 * hipstr_hmm.h — C-ABI of the MI355X-native read-to-haplotype HMM alignment and
 * diplotype-posterior core (drop-in boundary for tfwillems/HipSTR's hot path).
 *
 * Every entry point below names the reference interface it replaces
 * (paths relative to the HipSTR v0.7 tree).  Plain pointers and sizes only; no
 * C++ or torch types cross this boundary.  The library never falls back to a
 * CPU path: if no gfx950 device can be opened every call returns non-zero and
 * hipstr_last_error() says why.
 *
 * Conventions
 *   - all log-likelihoods are natural logs, IEEE double (the reference computes
 *     in double; only its log-sum-exp approximations drop to float, and those
 *     are bit-replicated on the device)
 *   - a "locus" is one STR region = one Haplotype of exactly three HapBlocks
 *     [left flank, STR block, right flank] (the reference asserts this shape,
 *     src/SeqAlignment/Haplotype.cpp:12)
 *   - a "read" on this boundary is a pooled read (ReadPooler pool,
 *     src/read_pooler.h:13-53) — the unit HapAligner::process_reads sees
 *   - candidate haplotypes ("alleles") of a locus are indexed in the visit
 *     order of Haplotype::next() (reflected mixed-radix Gray code, block 0
 *     fastest; src/SeqAlignment/Haplotype.cpp:157-196)
 *   - return value 0 = ok; non-zero = error (the C++ adapter turns it into
 *     printErrorAndDie, src/error.cpp:5-9, to keep the reference convention)
 */
#ifndef HIPSTR_HMM_H_
#define HIPSTR_HMM_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The functions declared here — and nothing else — are the library's ABI: libhipstr_hmm.so is built with -fvisibility=hidden. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define HIPSTR_NUM_BLOCKS        3   /* Haplotype.cpp:12 */
#define HIPSTR_MAX_STUTTER_REPS  6   /* RepeatStutterInfo.h:10-11 (MAX_STUTTER_REPEAT_INS / _DEL) */
#define HIPSTR_NUM_ARTIFACTS     (2 * HIPSTR_MAX_STUTTER_REPS + 1)
#define HIPSTR_MAX_HOMOP_LEN     15  /* AlignmentModel.h:6 */

/*
 * A batch of independent loci, flattened structure-of-arrays.  Host memory,
 * read-only, owned by the caller; the library copies what it needs.
 *
 * It carries exactly what the reference hands to
 *   HapAligner::HapAligner(Haplotype*, std::vector<bool>& realign_to_haplotype)   (HapAligner.h:56)
 *   HapAligner::process_reads(alignments, init_read_index, base_quality,
 *                             realign_read, aln_probs, seed_positions)           (HapAligner.h:86)
 * i.e. the HapBlock/RepeatBlock contents of the Haplotype (HapBlock.h:18-148,
 * RepeatBlock.h:26-43), the StutterModel parameters of the STR block
 * (stutter_model.h:31-60) and, per pooled read, the fields of `Alignment` the
 * path touches: sequence, base qualities, start and CIGAR (AlignmentData.h:28-137).
 *
 * Every entry point that takes a batch first checks that its tables agree with each other — counts in range, offsets non-negative
 * and never decreasing, no option longer than 65536 bases, no read longer than 1 Mi bases, CIGAR runs of positive length — and fails
 * the call with a message otherwise (where the reference would assert or index out of its vectors); what the pointers point AT cannot
 * be checked: the arrays must be as long as their offset tables say.
 */
typedef struct hipstr_batch {
  int32_t        n_loci;

  /* ---- haplotype structure ---- */
  const int32_t* blk_start;    /* [3*n_loci] HapBlock::start(), inclusive reference coordinate        */
  const int32_t* blk_end;      /* [3*n_loci] HapBlock::end(), exclusive                                */
  const int32_t* blk_nopts;    /* [3*n_loci] HapBlock::num_options() (option 0 = reference sequence)   */
  const int32_t* period;       /* [n_loci]   RepeatStutterInfo::get_period() of block 1                */
  const double*  stutter;      /* [6*n_loci] StutterModel ctor order: inframe_geom, inframe_up,
                                  inframe_down, outframe_geom, outframe_up, outframe_down
                                  (stutter_model.h:31-32); motif_len = period                         */
  const int32_t* opt_off;      /* [n_opts+1] byte offsets into seq; options are enumerated
                                  locus-major, block-major, option-minor                               */
  const char*    seq;          /* concatenated option sequences, ACGTN                                 */
  const int32_t* hap_off;      /* [n_loci+1] prefix sums of Haplotype::num_combs()                     */
  const uint8_t* realign_hap;  /* [hap_off[n_loci]] realign_to_haplotype flags, or NULL = all true     */

  /* ---- pooled reads ---- */
  const int32_t* read_off;     /* [n_loci+1] prefix sums of reads per locus                            */
  const int32_t* base_off;     /* [n_reads+1] offsets into bases/quals                                 */
  const char*    bases;        /* Alignment::get_sequence()                                            */
  const char*    quals;        /* Alignment::get_base_qualities(), Phred+33                            */
  const int32_t* read_start;   /* [n_reads] Alignment::get_start()                                     */
  const int32_t* cigar_off;    /* [n_reads+1] offsets into cigar_op/cigar_len                          */
  const char*    cigar_op;     /* CigarElement::get_type(): '=', 'X', 'I', 'D' only (HapAligner.cpp:309) */
  const int32_t* cigar_len;    /* CigarElement::get_num()                                              */
  const uint8_t* realign_read; /* [n_reads] realign_read flags, or NULL = all true                     */
} hipstr_batch_t;

/* Output layout shared by every align entry point:
 *   aln_probs[out_off[l] + i*A_l + k], i = read within locus l, k = allele, A_l = num_combs,
 *   out_off[l] = sum_{l'<l} P_l' * A_l'   — the row-major layout process_reads writes
 *   (HapAligner.cpp:324); seeds[read] = HapAligner::calc_seed_base (HapAligner.cpp:270-318). */
int hipstr_batch_out_offsets(const hipstr_batch_t* batch, int64_t* out_off /* [n_loci+1] */);

/*
 * Flat on-disk / wire form of a batch (host only): what a CPU worker that decoded and filtered the reads of a region shard
 * (the reference's read_and_filter_reads, bam_processor.cpp:173-473) hands to the process that owns a GPU.  One header, a
 * section table and the arrays of hipstr_batch_t back to back, checksummed; the reader points a hipstr_batch_t into one
 * allocation.  hipstr_batch_serialized_size returns -1 on an inconsistent batch.
 */
typedef struct hipstr_batch_file hipstr_batch_file_t;
int64_t hipstr_batch_serialized_size(const hipstr_batch_t* batch);
int  hipstr_batch_serialize(const hipstr_batch_t* batch, void* out, int64_t cap);
hipstr_batch_file_t* hipstr_batch_deserialize(const void* data, int64_t size);      /* NULL + hipstr_last_error() on a bad image */
int  hipstr_batch_write(const char* path, const hipstr_batch_t* batch);
hipstr_batch_file_t* hipstr_batch_read(const char* path);
const hipstr_batch_t* hipstr_batch_file_batch(const hipstr_batch_file_t* f);
void hipstr_batch_file_free(hipstr_batch_file_t* f);

/* Opaque device-resident batch (prepared tables + reads + output buffers in HBM). */
typedef struct hipstr_dev_batch hipstr_dev_batch_t;

/* Selects the device and builds the constant tables the reference keeps in
 * globals: INT_LOGS (mathops.cpp:13-21, precompute_integer_logs), the
 * LOG_MATCH_TO_* transition tables (AlignmentModel.cpp:20-32,
 * init_alignment_model) and BaseQuality's log tables (base_quality.h:29-38). */
int hipstr_hmm_init(int device_ordinal);
void hipstr_hmm_shutdown(void);
/* Device and pinned memory the library holds in its block caches but no object of the caller uses goes back to the driver (whole
 * chunks only: a chunk with one block still out stays).  The caches exist because a hipMalloc / hipFree next to running kernels stalls
 * for 0.1-0.9 s; a process that is done with a large stream and stays alive next to other users of the device calls this.  Returns the
 * bytes released.  (No counterpart in the reference: its matrices are new[]/delete[] per read, HapAligner.cpp:593-602.) */
int64_t hipstr_hmm_trim(void);

/* Flattens and uploads a batch.  Replaces HapAligner's constructor work
 * (reversed haplotype, StutterAlignerClass tables; HapAligner.h:56-69,
 * RepeatBlock.h:29-43) and calc_seed_base for every read. */
hipstr_dev_batch_t* hipstr_hmm_upload(const hipstr_batch_t* batch);
void hipstr_hmm_free(hipstr_dev_batch_t* dev);

/* Runs the forward HMM for every (realign_read, realign_hap) pair of the batch:
 * HapAligner::process_reads → process_read → align_seq_to_hap +
 * compute_aln_logprob (HapAligner.cpp:320-343, 573-709, 26-161, 163-231) with
 * StutterAlignerClass (StutterAlignerClass.cpp:12-162).  Asynchronous on
 * `hip_stream` (a hipStream_t, or NULL for the library's own stream). */
int hipstr_hmm_align(hipstr_dev_batch_t* dev, void* hip_stream);

/* Same, repeated `reps` times between two HIP events recorded on the stream the
 * kernels are launched on; *ms_total = elapsed milliseconds for all reps.
 * *ms_kernel (may be NULL) = time of the dominant DP kernel only. */
int hipstr_hmm_align_timed(hipstr_dev_batch_t* dev, int reps, float* ms_total, float* ms_kernel);

/* Per-pass, per-phase timing with HIP events recorded on the launch stream: after hipstr_hmm_profile(dev, 1)
 * every hipstr_hmm_align call records events at its phase boundaries; hipstr_hmm_profile_read synchronises and
 * writes, for up to `cap` passes (oldest first), four durations in ms: ms[4*i+0] leading-flank kernels,
 * [4*i+1] STR-block kernel, [4*i+2] trailing-flank kernels, [4*i+3] combine kernel; then clears the log.
 * Returns the number of passes written, or -1. */
int hipstr_hmm_profile(hipstr_dev_batch_t* dev, int enable);
int hipstr_hmm_profile_read(hipstr_dev_batch_t* dev, float* ms, int cap);

/* Number of (read x allele) HMM alignments one hipstr_hmm_align pass computes (realigned reads with a
 * seed x realigned alleles) and the ALGORITHMIC bytes of that pass (read bases+quals+len+seed,
 * per-allele haplotype bytes + homopolymer bytes + stutter pmf + run tables + block table, 8 B per output
 * log-likelihood, 4 B per seed). */
int hipstr_hmm_workload(hipstr_dev_batch_t* dev, int64_t* n_alignments, int64_t* algorithmic_bytes, int64_t* dp_cells);

/* Copies results back.  Entries whose read or allele was not realigned are left
 * untouched in the caller's buffers, as process_reads does (HapAligner.cpp:326-329,
 * 615-619); a realigned read without a seed (-1) gets 0 for EVERY allele of its row, realigned or
 * not, as in the reference (HapAligner.cpp:333-337). */
int hipstr_hmm_fetch(hipstr_dev_batch_t* dev, double* aln_probs, int32_t* seeds);

/* Device pointer to the batch's aln_probs buffer (same layout), for chaining
 * into hipstr_post_run without a host round trip.  The device buffer holds 0 where the host contract
 * says "untouched" (reads or alleles that were not realigned): chain it only for batches that realign
 * everything, or merge on the host (hipstr_hmm_fetch) when earlier values must survive. */
double* hipstr_hmm_dev_aln_probs(hipstr_dev_batch_t* dev);

/* One-shot convenience = upload + align + fetch + free: the drop-in for
 * HapAligner::process_reads (HapAligner.h:86-87). */
int hipstr_hmm_process_reads(const hipstr_batch_t* batch, double* aln_probs, int32_t* seeds);
/* The same, locus by locus as far as failures go: a locus the library cannot take (an input HapAligner::process_reads would die on —
 * an invalid seed or CIGAR, HapAligner.cpp:309,316 — or one beyond the library's limits) is left out, its part of aln_probs / seeds
 * stays untouched and locus_status[l] = 1; every other locus is processed as if it had been submitted alone (status 0).  Returns 0
 * unless the call itself failed (device, memory); hipstr_last_error() holds the message of the first refused locus. */
int hipstr_hmm_process_reads_each(const hipstr_batch_t* batch, double* aln_probs, int32_t* seeds, int32_t* locus_status /* [n_loci] */);

/* The same with seeds chosen by the caller: HapAligner::process_read takes the seed base as an argument (HapAligner.h:83,
 * HapAligner.cpp:573-575) and so does trace_optimal_aln (HapAligner.h:93); process_reads is the one caller that derives it
 * with calc_seed_base.  seed_base[r] >= 1 is used as given (it must leave a base on either side, HapAligner.cpp:316),
 * -1 marks a read without a seed (row of zeros), HIPSTR_SEED_AUTO asks for calc_seed_base; seed_base == NULL = all auto. */
#define HIPSTR_SEED_AUTO (-2)
hipstr_dev_batch_t* hipstr_hmm_upload_seeded(const hipstr_batch_t* batch, const int32_t* seed_base /* [n_reads] or NULL */);
int hipstr_hmm_process_reads_seeded(const hipstr_batch_t* batch, const int32_t* seed_base, double* aln_probs, int32_t* seeds);

/*
 * Streaming form of hipstr_hmm_process_reads: the host pipeline between a caller that produces loci one region at a time — as
 * the reference's does (BamProcessor::process_regions, bam_processor.cpp:550-617 -> GenotyperBamProcessor::analyze_reads_and_phasing,
 * genotyper_bam_processor.cpp:229-243) — and kernels that want ~10^6-10^7 alignments per launch.  Submitted loci are copied, collected
 * into batches of about `batch_alignments` (reads x haplotypes), prepared on the host threads and sent while the previous batch
 * still runs on the device (`slots` batches may be in flight), and results are handed back STRICTLY IN SUBMISSION ORDER: with
 * loci submitted in region order that is the order the VCF writer needs (vcf_writer.cpp:7-36), i.e. the per-GPU ordered gather.
 * One stream per device; a stream may be fed from one thread and drained from another.
 */
typedef struct hipstr_stream hipstr_stream_t;
typedef struct hipstr_stream_opts {
  int32_t device;             /* ordinal; hipstr_stream_open initialises it like hipstr_hmm_init                              */
  int32_t slots;              /* batches in flight (prepared / running / waiting to be collected); 0 = 6                      */
  int64_t batch_alignments;   /* a pending batch is sent once it holds this many (read x haplotype) pairs; 0 = 2 Mi           */
                              /* (the pairs in flight stay within slots x min(batch_alignments, 2 Mi), two batches at least:  */
                              /*  a larger batch size does not multiply the device memory the stream holds)                   */
} hipstr_stream_opts_t;
typedef struct hipstr_stream_stats {
  int64_t batches, tickets, alignment_slots;  /* batches launched, tickets delivered, (read x haplotype) pairs submitted       */
  double  host_seconds;       /* worker thread: prepare + staging + launches, summed over batches                              */
  double  wait_seconds;       /* hipstr_stream_next: time spent waiting for a batch to land                                    */
  double  open_seconds;       /* since hipstr_stream_open                                                                      */
  /* CPU seconds (thread CPU clocks, not wall time) by role: what the stream costs the host */
  double  cpu_submit_seconds;   /* inside hipstr_stream_submit / _submit_each on the callers' threads (checks, seeds, the copy in)  */
  double  cpu_prepare_seconds;  /* workers: prepare_batch (a worker with a budget of one host thread does all of it itself)         */
  double  cpu_upload_seconds;   /* workers: packing the staging block, copies and launches queued                                   */
  double  cpu_collect_seconds;  /* inside hipstr_stream_take / _next / _collect on the callers' threads (wait + copy out)           */
} hipstr_stream_stats_t;
hipstr_stream_t* hipstr_stream_open(const hipstr_stream_opts_t* opts /* NULL = defaults on device 0 */);
/* Queues the loci of `loci` (1..n loci; arrays are copied).  Returns the submission's ticket (0, 1, 2, ...) or -1. */
int64_t hipstr_stream_submit(hipstr_stream_t* s, const hipstr_batch_t* loci);
/* Every locus of `loci` as its own submission, in order; *first_ticket = the ticket of locus 0 (the others follow consecutively). */
int hipstr_stream_submit_each(hipstr_stream_t* s, const hipstr_batch_t* loci, int64_t* first_ticket);
/* The next n_tickets submissions, in order, into back-to-back buffers; *n_out / *n_reads = doubles / seeds written. */
int hipstr_stream_collect(hipstr_stream_t* s, int64_t n_tickets, double* aln_probs, int64_t cap_probs, int32_t* seeds, int64_t cap_seeds,
                          int64_t* n_out, int64_t* n_reads);
/* Sends the pending batch now, whatever its size. */
int hipstr_stream_flush(hipstr_stream_t* s);
/* Sizes of the next submission to be delivered: n_out doubles of aln_probs, n_reads seeds.  Returns 2 when nothing is outstanding. */
int hipstr_stream_next_size(hipstr_stream_t* s, int64_t* ticket, int64_t* n_out, int64_t* n_reads);
/* Blocks until the next submission (in submission order) is done and writes its results exactly as hipstr_hmm_process_reads
 * would have for that submission alone: aln_probs / seeds laid out as hipstr_batch_out_offsets of the submitted batch, entries of
 * reads / haplotypes that were not realigned left untouched.  Returns 0, 1 on error (the submission is consumed: its batch failed),
 * 2 when nothing is outstanding, 3 when the buffers are too small for it (hipstr_stream_next_size) — the submission stays
 * outstanding and the call can be repeated with larger ones. */
int hipstr_stream_next(hipstr_stream_t* s, int64_t* ticket, double* aln_probs, int64_t cap_probs, int32_t* seeds, int64_t cap_seeds);
/* Collects ONE submission by its ticket, in any order (each ticket once): blocks until its batch has run.  For callers that keep
 * many loci in flight, one host thread per locus: SeqStutterGenotyper::genotype is a per-locus state machine (align, posteriors,
 * tracebacks, new alleles, align only those ... seq_stutter_genotyper.cpp:603-671), and the rounds of different loci share batches.
 * hipstr_stream_next(s, ...) == hipstr_stream_take(s, lowest ticket not collected yet, ...), same return codes.  A ticket whose batch
 * has not been launched yet is sent out even when all `slots` are held by batches with uncollected earlier tickets; two collectors
 * asking for the same ticket: the second is refused; hipstr_stream_close makes blocked collectors return with an error. */
int hipstr_stream_take(hipstr_stream_t* s, int64_t ticket, double* aln_probs, int64_t cap_probs, int32_t* seeds, int64_t cap_seeds);
int hipstr_stream_stats(hipstr_stream_t* s, hipstr_stream_stats_t* out);
/* Drops whatever has not been delivered and releases the stream. */
int hipstr_stream_close(hipstr_stream_t* s);

/*
 * Several GPUs from one process (STR loci are independent, the region list shards across the GPUs of a node with no
 * collective on the data path).  One hipstr_stream_t per device; submissions fill contiguous blocks of about `block_alignments`
 * (read x haplotype) pairs; a new block goes to the device that has been dealt the least estimated WORK so far (hipstr_locus_costs:
 * loci with interrupted repeats cost several times a periodic locus' pairs), ties to the lowest slot; hipstr_multi_next hands results
 * back in GLOBAL submission order.
 * devices == NULL: ordinals 0..n_devices-1.  (One process per GPU — torch.distributed / MPI launchers — needs nothing of this:
 * every process opens its own stream on its own device.)
 */
typedef struct hipstr_multi hipstr_multi_t;
hipstr_multi_t* hipstr_multi_open(int32_t n_devices, const int32_t* devices, int64_t block_alignments /* 0 = 16 Mi */, const hipstr_stream_opts_t* per_stream);
int64_t hipstr_multi_submit(hipstr_multi_t* m, const hipstr_batch_t* loci);
int hipstr_multi_flush(hipstr_multi_t* m);
int hipstr_multi_next_size(hipstr_multi_t* m, int64_t* ticket, int64_t* n_out, int64_t* n_reads);
int hipstr_multi_next(hipstr_multi_t* m, int64_t* ticket, double* aln_probs, int64_t cap_probs, int32_t* seeds, int64_t cap_seeds);
int hipstr_multi_close(hipstr_multi_t* m);
/* Estimated work dealt to every device so far (units of hipstr_locus_costs); returns the number of devices, fills at most `cap` entries. */
int hipstr_multi_dealt(hipstr_multi_t* m, double* cost_per_device, int32_t cap);
/* Work estimate per locus, in units of one (150-base read, allele) pair of a periodic repeat with 60 flank bases: reads x realigned alleles
 * x mean read length / 150 x [1.4 x flank bases / 60 + 1 + 2.4 x interruptions of the repeat per allele] — SURVEY 8(e)'s "sum of P A L H"
 * with the STR block priced by what it costs on the device.  What the region list is split by across ranks (hipstr_amd/shard.py) and
 * what hipstr_multi_submit deals blocks by.  Host-only (no device needed). */
int hipstr_locus_costs(const hipstr_batch_t* batch, double* costs /* [n_loci] */);

/*
 * Host-side ordered gather of per-worker record streams: the reference's VCF writer accepts out-of-order positions
 * only within MAX_RECORD_PAD = 50 bp (vcf_writer.h:53, vcf_writer.cpp:7-36), so the records produced by the workers of a sharded
 * region list are merged back by (chromosome index, position).  Every stream pushes its records in order and ends; a record is
 * released once it is the smallest pending one and no unfinished stream is empty.  Host only.
 * hipstr_gather_pop: 0 = record released; 2 = not yet (an unfinished stream has nothing pending); 3 = all streams ended and
 * drained; 1 = error (*bytes holds the size needed when the buffer was too small; the record stays queued).
 */
typedef struct hipstr_gather hipstr_gather_t;
hipstr_gather_t* hipstr_gather_open(int32_t n_streams);
int hipstr_gather_push(hipstr_gather_t* g, int32_t stream, int32_t chrom_index, int32_t pos, const void* record, int64_t bytes);
int hipstr_gather_end(hipstr_gather_t* g, int32_t stream);
int hipstr_gather_pop(hipstr_gather_t* g, int32_t* stream, int32_t* chrom_index, int32_t* pos, void* out, int64_t cap, int64_t* bytes);
void hipstr_gather_close(hipstr_gather_t* g);

/* HapAligner::calc_seed_base (HapAligner.cpp:270-318) for every read of a batch,
 * host only (no device needed): seeds[r] = read offset of the seed base or -1.
 * Returns non-zero on the inputs the reference dies on ("Invalid alignment
 * seed", "Unrecognized CIGAR char"). */
int hipstr_calc_seed_bases(const hipstr_batch_t* batch, int32_t* seeds);

/*
 * Diplotype posteriors: Genotyper::calc_log_sample_posteriors (genotyper.cpp:44-80)
 * with the default priors of Genotyper::init_log_sample_priors (genotyper.cpp:20-42)
 * and the MAP scan of Genotyper::get_optimal_haplotypes (genotyper.cpp:82-97),
 * batched over loci.  All arrays are host pointers unless stated.
 */
typedef struct hipstr_post_batch {
  int32_t        n_loci;
  const int32_t* n_alleles;     /* [n_loci] num_alleles_                                               */
  const int32_t* n_samples;     /* [n_loci] num_samples_                                               */
  const int32_t* read_off;      /* [n_loci+1] prefix sums of num_reads_ (un-pooled reads)              */
  const int32_t* sample_label;  /* [n_reads] sample index within the locus (genotyper.h:26)            */
  const double*  log_p1;        /* [n_reads] SNP phasing log-likelihoods (genotyper.h:25)              */
  const double*  log_p2;        /* [n_reads]                                                           */
  const int32_t* read_weight;   /* [n_reads] read_weights_ (0 for second mates; genotyper.h:44-46)     */
  const double*  log_aln_probs; /* [sum R_l*A_l] log_aln_probs_, read-major (genotyper.h:33); may be
                                   NULL when a device buffer is given to hipstr_post_run              */
  const uint8_t* haploid;       /* [n_loci] haploid_ flag, or NULL = diploid                           */
  const double*  log_prior;     /* [sum S_l*A_l^2] optional: the array a derived class's virtual
                                   init_log_sample_priors fills (genotyper.h:69; EMStutterGenotyper
                                   overrides it, em_stutter_genotyper.cpp:129-144); NULL = the default
                                   hom/het priors of genotyper.cpp:20-42                               */
} hipstr_post_batch_t;

/* log_post[post_off[l] + (s*A + a1)*A + a2]; post_off[l] = sum_{l'<l} S_l'*A_l'^2
 * sample_total_ll[samp_off[l] + s];           samp_off[l] = sum_{l'<l} S_l'
 * map_gt[2*(samp_off[l]+s) + {0,1}]                                                                     */
int hipstr_post_offsets(const hipstr_post_batch_t* pb, int64_t* post_off, int64_t* samp_off);

/* dev_log_aln_probs: optional device pointer overriding pb->log_aln_probs.
 * locus_total_ll[l] = the return value of calc_log_sample_posteriors (sum of sample totals). */
int hipstr_post_run(const hipstr_post_batch_t* pb, const double* dev_log_aln_probs,
                    double* log_post, double* sample_total_ll, int32_t* map_gt, double* locus_total_ll);

/* Resident form of the same computation (inputs uploaded once, kernel launched asynchronously on the
 * library stream or `hip_stream`, results fetched on demand) — what bench.py times. */
typedef struct hipstr_post_dev hipstr_post_dev_t;
hipstr_post_dev_t* hipstr_post_upload(const hipstr_post_batch_t* pb, const double* dev_log_aln_probs);
int  hipstr_post_launch(hipstr_post_dev_t* pd, void* hip_stream);
int  hipstr_post_fetch(hipstr_post_dev_t* pd, double* log_post, double* sample_total_ll, int32_t* map_gt, double* locus_total_ll);
void hipstr_post_free(hipstr_post_dev_t* pd);

/*
 * Genotype calls: Genotyper::extract_genotypes_and_likelihoods (genotyper.cpp:129-251) with calc_PLs (99-104) and calc_gl_diff
 * (106-127) on the resident posteriors of a hipstr_post_dev_t (after hipstr_post_launch): MAP haplotype pair -> genotype of
 * the STR block, haplotype posteriors marginalised to genotype posteriors (streaming log-sum-exp in the reference's order),
 * Q / PQ values, GL (log10, priors removed), GLDIFF, PL, PHASEDGL.  hap_to_allele maps every haplotype of a locus to its
 * variant (option of the STR block, SeqStutterGenotyper::haps_to_alleles); every variant must be hit by a haplotype.
 * Per sample the GL / PL arrays hold V(V+1)/2 values (diploid, VCF order) or V (haploid); PHASEDGL V*V or V.
 */
typedef struct hipstr_gt_request {
  const int32_t* n_variants;      /* [n_loci] V                                                                      */
  const int32_t* hap_to_allele;   /* [sum A_l]                                                                       */
  int32_t calc_gls, calc_pls, calc_phased_gls;
} hipstr_gt_request_t;
typedef struct hipstr_gt_out {
  int32_t* best_hap;              /* [2*n_samp] best_haplotypes                                                      */
  int32_t* best_gt;               /* [2*n_samp] best_gts                                                             */
  double*  log_phased_post;       /* [n_samp]                                                                        */
  double*  log_unphased_post;     /* [n_samp]                                                                        */
  double*  hap_log_phased_post;   /* [n_samp]                                                                        */
  double*  hap_log_unphased_post; /* [n_samp]                                                                        */
  double*  gl_diff;               /* [n_samp]; written when any calc_* flag is set                                    */
  double*  gls;                   /* [gl_off[n_samp]]  when calc_gls                                                  */
  int32_t* pls;                   /* [gl_off[n_samp]]  when calc_pls                                                  */
  double*  phased_gls;            /* [pgl_off[n_samp]] when calc_phased_gls                                           */
} hipstr_gt_out_t;
/* gl_off / pgl_off: [n_samp+1] starts of every sample's piece (samples in locus order, as sample_total_ll). */
int hipstr_gt_offsets(const hipstr_post_batch_t* pb, const hipstr_gt_request_t* rq, int64_t* gl_off, int64_t* pgl_off);
int hipstr_post_extract(hipstr_post_dev_t* pd, const hipstr_gt_request_t* rq, hipstr_gt_out_t* out);

/*
 * De novo stutter model: EMStutterGenotyper::train (em_stutter_genotyper.cpp:146-226) with its E-step
 * (calc_hap_aln_probs :146-150, Genotyper::calc_log_sample_posteriors under the allele-frequency priors of :129-144,
 * recalc_log_read_phase_posteriors :152-169) and M-step (recalc_log_gt_priors :22-57, recalc_stutter_model :64-127), batched
 * over loci.  A read is its observed STR size (bp difference from the reference); the alleles of a locus are the distinct
 * sizes, the reference size first and the rest ascending (em_stutter_genotyper.h:55-78).  Reads of a locus are grouped by
 * ascending sample, as in hipstr_post_batch_t.
 */
typedef struct hipstr_em_batch {
  int32_t        n_loci;
  const int32_t* period;        /* [n_loci] motif length                                                          */
  const uint8_t* haploid;       /* [n_loci] or NULL = diploid                                                     */
  const int32_t* n_samples;     /* [n_loci]                                                                       */
  const int32_t* read_off;      /* [n_loci+1]                                                                     */
  const int32_t* sample_label;  /* [n_reads]                                                                      */
  const int32_t* num_bps;       /* [n_reads] observed STR size of the read                                        */
  const double*  log_p1;        /* [n_reads]                                                                      */
  const double*  log_p2;        /* [n_reads]                                                                      */
  int32_t        ref_allele;    /* size of the reference allele (0 at both call sites of the reference)           */
  int32_t        max_iter;      /* MAX_EM_ITER = 100 (genotyper_bam_processor.h:106)                              */
  double         min_ll_abs_change;   /* ABS_LL_CONVERGE = 0.01                                                   */
  double         min_ll_frac_change;  /* FRAC_LL_CONVERGE = 0.001                                                 */
} hipstr_em_batch_t;
/* trained[l]   = the return value of train();
 * stutter[6*l] = inframe geom, up, down, outframe geom, up, down of get_stutter_model() after train();
 * n_iter[l]    = E-steps performed;  final_ll[l] = the last E-step's total log-likelihood. */
int hipstr_em_train(const hipstr_em_batch_t* eb, uint8_t* trained, double* stutter, int32_t* n_iter, double* final_ll);

/*
 * Needleman-Wunsch with affine gaps: NeedlemanWunsch::Align (NeedlemanWunsch.cpp:370-420: initMatrices 326-367, nw_helper 195-245,
 * findOptimalStop 142-171 / findOptimalStopEndPenalty 173-193, traceAlignment 247-324) for a batch of (reference, read) pairs —
 * the step before the HMM: realign() aligns every unique read to its reference window (AlignmentOps.cpp:14-26,
 * genotyper_bam_processor.cpp:68) and Haplotype::aln_haps_to_ref aligns every haplotype to the reference haplotype with the
 * end penalty (Haplotype.cpp:58-86).  Scores are float sums of 2, -2, -5 and -0.125, exact in any order; ties are broken as
 * bestIndex does (NeedlemanWunsch.cpp:120-140).  Limits: second sequence <= 1536 bases, reference <= 4095.
 */
typedef struct hipstr_nw_batch {
  int32_t        n_pairs;
  const int32_t* ref_off;        /* [n_pairs+1] into ref_seqs                                                        */
  const char*    ref_seqs;
  const int32_t* read_off;       /* [n_pairs+1] into read_seqs                                                       */
  const char*    read_seqs;
  int32_t        use_ref_end_penalty;
} hipstr_nw_batch_t;
typedef struct hipstr_nw_out {
  float*   score;                /* [n_pairs]                                                                         */
  uint8_t* ok;                   /* [n_pairs] the return value of Align                                               */
  int64_t* aln_off;              /* [n_pairs+1]: ref_seq_al / read_seq_al of pair i occupy [aln_off[i], aln_off[i+1])  */
  char*    ref_al;
  char*    read_al;
  int64_t* cigar_off;            /* [n_pairs+1]                                                                       */
  char*    cigar_op;             /* '=', 'X', 'I', 'D'                                                                */
  int32_t* cigar_len;
  int64_t  cap_aln, cap_cigar;   /* capacities of the pools                                                           */
} hipstr_nw_out_t;
int hipstr_nw_align(const hipstr_nw_batch_t* nb, hipstr_nw_out_t* out);

/* Haplotype::get_aln_info() of every haplotype of every locus (Haplotype::aln_haps_to_ref, Haplotype.cpp:58-86): Needleman-Wunsch
 * of the haplotype against the reference haplotype (all-first-options) with the end penalty, indels in the leading flank pushed
 * into the repeat (adjust_indels, Haplotype.cpp:8-56), then one of 'M','I','D' per alignment column.  The NUL-terminated string
 * of haplotype k of locus l starts at out + offs[hap_off[l] + k]; offs has hap_off[n_loci] + 1 entries.  This is the hap_to_ref
 * input of hipstr_hmm_trace.  Limits: hipstr_nw_align's — a haplotype of at most 1536 bases, a reference haplotype of at most 4095 (a locus beyond
 * them makes the call fail with that message: 1 + hipstr_last_error()). */
int hipstr_hap_aln_info(const hipstr_batch_t* batch, char* out, int64_t out_cap, int64_t* offs);

/*
 * Viterbi traceback: HapAligner::trace_optimal_aln (HapAligner.cpp:711-722) = process_read(..., retrace_aln=true) on one
 * fixed haplotype: full M/I/D matrices of both sides, arg-max seed position (compute_aln_logprob's max_index,
 * HapAligner.cpp:184-222), HapAligner::retrace (HapAligner.cpp:363-571) with its 0.001-nat tie tolerances, and — when the
 * caller supplies the haplotype-to-reference alignment strings (Haplotype::get_aln_info) — stitch_alignment_trace
 * (AlignmentTraceback.cpp:55-144).  One request = (read, allele): req_read indexes the reads of the whole batch (which fixes
 * the locus), req_allele the haplotypes of that locus; the read must have a seed.  Requests of many loci go in one call —
 * that is what fills the device (one locus alone offers only ~2 wavefronts per request).
 * Everything an AlignmentTrace holds (AlignmentTraceback.h:10-108) comes back flattened; all arrays are caller-allocated,
 * string pools are filled back to back with *_off[] giving the start of each request's piece ([n_req+1] entries).
 */
#define HIPSTR_NO_STR_DATA (-100000)     /* str_data_[block] == NULL: the alignment never entered the STR block */
typedef struct hipstr_trace_out {
  double*  ll;             /* [n_req] log-likelihood of the traced alignment (== the forward score)                  */
  int32_t* max_index;      /* [n_req] haplotype position aligned with the seed base                                   */
  int32_t* hap_aln_off;    /* [n_req+1] */
  char*    hap_aln;        /* AlignmentTrace::hap_aln(): read-vs-haplotype operations 'M','I','D','S'                 */
  int32_t* stutter_size;   /* [n_req] AlignmentTrace::stutter_size(1) or HIPSTR_NO_STR_DATA                            */
  int32_t* str_seq_off;    /* [n_req+1] */
  char*    str_seq;        /* AlignmentTrace::str_seq(1)                                                              */
  int32_t* flank_seq_off;  /* [2*n_req+1]: left flank (block 0) then right flank (block 2) of each request            */
  char*    flank_seq;      /* AlignmentTrace::flank_seq(block)                                                        */
  int32_t* flank_ins;      /* [n_req] flank_ins_size()                                                                */
  int32_t* flank_del;      /* [n_req] flank_del_size()                                                                */
  int32_t* indel_off;      /* [n_req+1] */
  int32_t* indel_pos;      /* flank_indel_data(): (position, size) pairs in the order retrace records them             */
  int32_t* indel_size;
  int32_t* snp_off;        /* [n_req+1] */
  int32_t* snp_pos;        /* flank_snp_data(): (position, base)                                                       */
  char*    snp_base;
  /* stitched alignment against the reference (only when hap_to_ref is given) */
  int32_t* aln_start;      /* [n_req] traced_aln().get_start()                                                        */
  int32_t* aln_stop;       /* [n_req] traced_aln().get_stop()                                                         */
  int32_t* cigar_off;      /* [n_req+1] */
  char*    cigar_op;       /* traced_aln().get_cigar_list()                                                           */
  int32_t* cigar_len;
  int32_t* aln_str_off;    /* [n_req+1] */
  char*    aln_str;        /* traced_aln().get_alignment()                                                            */
  int32_t  cap_chars;      /* capacity of every char pool / pair pool above (per pool)                                */
} hipstr_trace_out_t;

/* hap_to_ref: NULL, or for every haplotype of every locus the NUL-terminated Haplotype::get_aln_info() string ('M','I','D' of
 * the haplotype against the reference haplotype): hap_to_ref[hap_off[locus] + k], [hap_off[n_loci]] pointers. */
int hipstr_hmm_trace(const hipstr_batch_t* batch, int32_t n_req, const int32_t* req_read, const int32_t* req_allele,
                     const char* const* hap_to_ref, hipstr_trace_out_t* out);
/* req_seed: the seed_base argument of trace_optimal_aln per request (HapAligner.h:93), or NULL / HIPSTR_SEED_AUTO entries to
 * have it computed with calc_seed_base as process_reads does. */
int hipstr_hmm_trace_seeded(const hipstr_batch_t* batch, int32_t n_req, const int32_t* req_read, const int32_t* req_allele,
                            const int32_t* req_seed, const char* const* hap_to_ref, hipstr_trace_out_t* out);

/* The diagnostics entry points (hipstr_debug_*: what the tests, the fuzzers and bench.py look inside the library with) are declared in
 * hipstr_hmm_debug.h — not part of the drop-in ABI; a build with -DHIPSTR_NO_DEBUG_ABI leaves them out of the library. */

const char* hipstr_last_error(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* HIPSTR_HMM_H_ */
