// HapAlignerMI355X.h — the reference-side binding a HipSTR maintainer adds to src/SeqAlignment/.
//
// A class with HapAligner's public interface (HapAligner.h:56-93) written against the REFERENCE's
// own types (Haplotype, HapBlock, RepeatBlock, Alignment, BaseQuality, AlignmentTrace).
//
// Two ways in:
//   * ONE include switch, no source edit: compile seq_stutter_genotyper.cpp with
//         -include SeqAlignment/HapAlignerMI355X.h -DHIPSTR_MI355X_AS_HAPALIGNER
//     This header pulls in the reference's HapAligner.h first (so the CPU class keeps its name and
//     its include guard is spent) and then renames every later mention of `HapAligner` to this class:
//     the three construction sites (seq_stutter_genotyper.cpp:522, :814, :1076) compile unedited.
//     oracle/Makefile target `flow` does exactly that and runs the reference's own
//     SeqStutterGenotyper::genotype() on the MI355X (integration/genotype_flow.cpp).
//   * or name the class explicitly at those three sites.
//
//   process_reads / calc_seed_base  ->  libhipstr_hmm.so  (include/hipstr_hmm.h, MI355X kernels)
//   trace_optimal_aln               ->  hipstr_hmm_trace (one request), same AlignmentTrace the reference builds
//   trace_optimal_alns              ->  hipstr_hmm_trace (all requests of a locus in one launch): what
//                                       SeqStutterGenotyper::retrace_alignments (seq_stutter_genotyper.cpp:805-841) and
//                                       the loop at :1111-1122 should call once instead of trace_optimal_aln per read
//
// This file is NOT part of the product library and is only compiled where the HipSTR tree is
// available (oracle/Makefile target `dropin`, which also builds the drop-in equivalence check).
#ifndef HAP_ALIGNER_MI355X_H_
#define HAP_ALIGNER_MI355X_H_

#include <string>
#include <vector>

#include "AlignmentData.h"
#include "AlignmentTraceback.h"
#include "HapAligner.h"
#include "Haplotype.h"
#include "../base_quality.h"

class HapAlignerMI355X {
 private:
  Haplotype* fw_haplotype_;                 // borrowed, as in HapAligner (HapAligner.h:58)
  std::vector<bool> realign_to_hap_;

  // flattened haplotype (built once per aligner, like the reference builds its reversed haplotype once)
  std::vector<int32_t> blk_start_, blk_end_, blk_nopts_, opt_off_, hap_off_;
  std::vector<double> stutter_;
  std::vector<uint8_t> realign_hap_;
  std::string seq_;
  int32_t period_;

  HapAlignerMI355X(const HapAlignerMI355X& other);
  HapAlignerMI355X& operator=(const HapAlignerMI355X& other);

 public:
  HapAlignerMI355X(Haplotype* haplotype, std::vector<bool>& realign_to_haplotype);

  // Optional: route process_reads through a shared hipstr_stream_t (include/hipstr_hmm.h) instead of a one-shot call.  With several
  // loci in flight — one host thread per SeqStutterGenotyper — their alignment rounds then share device batches: each call submits its
  // locus and waits for its own ticket (hipstr_stream_take).  NULL (the default) = one-shot calls.
  static void use_stream(struct hipstr_stream* stream);
  // genotype_flow --profile: seconds[0..2] = flatten, fill AlignmentTrace objects, haplotype strings, summed over threads; then reset and switch on / off
  static void profile(bool enable, double seconds[3]);
  // trace_optimal_aln's prefetch (see HapAlignerMI355X.cpp): requests served from it, requests that had to go to the device, device calls
  // made for them, tracebacks computed ahead — summed over threads since the process started
  static void trace_cache_stats(long long counts[4]);

  int calc_seed_base(const Alignment& alignment);

  // HapAligner::process_read (HapAligner.h:83, HapAligner.cpp:573-709): one read against every haplotype from the
  // haplotype's CURRENT position to the last one (a fixed haplotype: just that one), with the seed the caller names.
  // *prob_ptr advances one entry per haplotype visited; entries of haplotypes that are not realigned are left untouched.
  // With retrace_aln, every haplotype that improves on the best likelihood so far is traced into traced_aln, as the
  // reference does (its only caller with retrace_aln = true, trace_optimal_aln, fixes the haplotype first).
  void process_read(const Alignment& aln, int seed_base, const BaseQuality* base_quality, bool retrace_aln,
		    double* prob_ptr, AlignmentTrace& traced_aln);

  void process_reads(const std::vector<Alignment>& alignments, int init_read_index, const BaseQuality* base_quality,
		     const std::vector<bool>& realign_read, double* aln_probs, int* seed_positions);

  // HapAligner::trace_optimal_aln (HapAligner.h:88-92).  The caller owns the returned object, as with the reference.
  AlignmentTrace* trace_optimal_aln(const Alignment& orig_aln, int seed_base, int best_haplotype, const BaseQuality* base_quality);

  // Batched form: request i traces alignments[i], split at seed_bases[i], against haplotype best_haplotypes[i];
  // traces[i] is a new AlignmentTrace.  seed_bases may be empty: calc_seed_base's value is used then.
  void trace_optimal_alns(const std::vector<Alignment>& alignments, const std::vector<int>& seed_bases, const std::vector<int>& best_haplotypes,
			  const BaseQuality* base_quality, std::vector<AlignmentTrace*>& traces);
  void trace_optimal_alns(const std::vector<Alignment>& alignments, const std::vector<int>& best_haplotypes,
			  const BaseQuality* base_quality, std::vector<AlignmentTrace*>& traces){
    trace_optimal_alns(alignments, std::vector<int>(), best_haplotypes, base_quality, traces);
  }

 private:
  void run_traces(const std::vector<Alignment>& alignments, const std::vector<int>& seed_bases, const std::vector<int>& best_haplotypes,
		  const std::vector<AlignmentTrace*>& targets);
  // the general form: request i = reads[req_read[i]] split at req_seed[i] against haplotype req_hap[i]
  void run_trace_requests(const std::vector<Alignment>& reads, const std::vector<int32_t>& req_read, const std::vector<int32_t>& req_seed,
			  const std::vector<int32_t>& req_hap, const std::vector<AlignmentTrace*>& targets);
  void run_trace_requests(const struct HipstrFlatReads& reads, const std::vector<int32_t>& req_read, const std::vector<int32_t>& req_seed,
			  const std::vector<int32_t>& req_hap, const std::vector<AlignmentTrace*>& targets);
  unsigned long long haplotype_hash() const;
  AlignmentTrace* prefetched_trace(const Alignment& orig_aln, int seed_base, int best_haplotype);
  void fill_trace(int i, const struct hipstr_trace_out* o, const std::string& orig_quals, const std::string& orig_seq, AlignmentTrace& t) const;
};

#ifdef HIPSTR_MI355X_AS_HAPALIGNER
#define HapAligner HapAlignerMI355X
#endif

#endif
