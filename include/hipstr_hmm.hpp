// hipstr_hmm.hpp — C++ host API above the C-ABI (include/hipstr_hmm.h), header-only.
//
// Mirrors, for the hot path only, the classes a HipSTR caller touches — same names, same
// constructor/argument meaning, same ownership and error behaviour — so code written against
// the reference's SeqAlignment/HapAligner + Genotyper interfaces reads the same here:
//
//   reference (HipSTR v0.7)                               here (namespace hipstr_amd)
//   ----------------------------------------------------- -----------------------------------------
//   StutterModel(ig, iu, id, og, ou, od, motif)           StutterModel                (stutter_model.h:31)
//   BaseQuality::median_base_qualities                    BaseQuality                 (base_quality.cpp:11-28)
//   CigarElement / Alignment                              CigarElement / Alignment    (AlignmentData.h:12-137)
//   HapBlock(start,end,ref) / add_alternate               HapBlock                    (HapBlock.h:43-70)
//   RepeatBlock(start,end,ref,period,stutter_model)       RepeatBlock                 (RepeatBlock.h:28-43)
//   Haplotype(std::vector<HapBlock*>&) / num_combs        Haplotype                   (Haplotype.h:33-52)
//   ReadPooler::add_alignment / pool / get_alignments     ReadPooler                  (read_pooler.h:13-53)
//   HapAligner(Haplotype*, std::vector<bool>&)            HapAligner                  (HapAligner.h:56)
//     ::calc_seed_base / ::process_reads                                              (HapAligner.h:81-87)
//   Genotyper::calc_log_sample_posteriors /               Genotyper                   (genotyper.h:72-79)
//     ::get_optimal_haplotypes
//   SeqStutterGenotyper::calc_hap_aln_probs               calc_hap_aln_probs()        (seq_stutter_genotyper.cpp:519-568)
//
// All arithmetic of the path happens on the MI355X behind hipstr_hmm_process_reads /
// hipstr_post_run; these classes only hold and flatten data.  Fatal conditions follow the
// reference's convention: message on stderr, exit(1) (error.cpp:5-9).
#ifndef HIPSTR_HMM_HPP_
#define HIPSTR_HMM_HPP_

#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <map>
#include <unordered_map>
#include <string>
#include <utility>
#include <vector>

#include "hipstr_hmm.h"

namespace hipstr_amd {

inline void printErrorAndDie(const std::string& message){
  std::cerr << "ERROR: " << message << "\n" << "Exiting..." << std::endl;
  exit(1);
}

class StutterModel {
  double p_[6]; int motif_len_;
 public:
  StutterModel(double inframe_geom, double inframe_up, double inframe_down,
               double outframe_geom, double outframe_up, double outframe_down, int motif_len){
    assert(inframe_geom > 0.0 && inframe_geom < 1.0 && outframe_geom > 0.0 && outframe_geom < 1.0);
    assert(inframe_up + inframe_down + outframe_up + outframe_down < 1.0 && motif_len > 0 && motif_len < 10);
    p_[0] = inframe_geom; p_[1] = inframe_up; p_[2] = inframe_down; p_[3] = outframe_geom; p_[4] = outframe_up; p_[5] = outframe_down;
    motif_len_ = motif_len;
  }
  StutterModel* copy() const { return new StutterModel(*this); }
  int period() const { return motif_len_; }
  const double* parameters() const { return p_; }
};

class BaseQuality {
 public:
  static const char MIN_BASE_QUALITY = '!';
  static const char MAX_BASE_QUALITY = 'J';
  // upper median per position over the reads of a pool (base_quality.cpp:11-28)
  std::string median_base_qualities(const std::vector<const std::string*>& qualities) const {
    assert(!qualities.empty());
    for (size_t i = 0; i < qualities.size(); i++)
      if (qualities[i]->size() != qualities[0]->size())
        printErrorAndDie("All base quality strings must be of the same length when averaging probabilities");
    std::string med(qualities[0]->size(), 'N');
    std::vector<char> col(qualities.size());
    for (size_t i = 0; i < med.size(); i++){
      for (size_t j = 0; j < qualities.size(); j++) col[j] = (*qualities[j])[i];
      std::sort(col.begin(), col.end());
      med[i] = col[col.size()/2];
    }
    return med;
  }
};

class CigarElement {
  char type_; int num_;
 public:
  CigarElement(char type, int num) : type_(type), num_(num) {}
  char get_type() const { return type_; }
  int  get_num()  const { return num_;  }
};

class Alignment {
  int32_t start_, stop_;
  std::vector<CigarElement> cigar_list_;
  std::string name_, base_qualities_, sequence_, alignment_;
  bool rev_strand_;
 public:
  Alignment(int32_t start, int32_t stop, bool rev_strand, const std::string& name, const std::string& base_qualities,
            const std::string& sequence, const std::string& alignment)
    : start_(start), stop_(stop), name_(name), base_qualities_(base_qualities), sequence_(sequence), alignment_(alignment), rev_strand_(rev_strand) {}
  explicit Alignment(const std::string& name) : start_(0), stop_(-1), name_(name), rev_strand_(false) {}    // AlignmentData.h:53-59
  const std::string& get_name() const { return name_; }
  std::string getCigarString() const {                           // AlignmentData.h:131-137
    std::string s;
    for (size_t i = 0; i < cigar_list_.size(); i++){ s += std::to_string(cigar_list_[i].get_num()); s += cigar_list_[i].get_type(); }
    return s;
  }
  int32_t get_start() const { return start_; }
  int32_t get_stop()  const { return stop_;  }
  bool is_from_reverse_strand() const { return rev_strand_; }
  void add_cigar_element(CigarElement e){ cigar_list_.push_back(e); }
  void set_cigar_list(const std::vector<CigarElement>& l){ cigar_list_ = l; }
  void set_base_qualities(const std::string& q){ base_qualities_ = q; }
  const std::string& get_base_qualities() const { return base_qualities_; }
  const std::string& get_sequence()       const { return sequence_; }
  const std::string& get_alignment()      const { return alignment_; }
  const std::vector<CigarElement>& get_cigar_list() const { return cigar_list_; }
};

class HapBlock {
 protected:
  std::string ref_seq_;
  std::vector<std::string> alt_seqs_;
  int32_t start_, end_;
 public:
  HapBlock(int32_t start, int32_t end, const std::string& ref_seq) : ref_seq_(ref_seq), start_(start), end_(end) {}
  virtual ~HapBlock() {}
  virtual const StutterModel* get_stutter_model() const { return NULL; }   // non-NULL <=> the reference's get_repeat_info() != NULL
  virtual int get_period() const { return 0; }
  int32_t start() const { return start_; }
  int32_t end()   const { return end_;   }
  int num_options() const { return 1 + (int)alt_seqs_.size(); }
  virtual void add_alternate(const std::string& alt){ alt_seqs_.push_back(alt); }
  const std::string& get_seq(unsigned int index) const { return index == 0 ? ref_seq_ : alt_seqs_.at(index-1); }
};

class RepeatBlock : public HapBlock {
  int period_; StutterModel* model_;
 public:
  RepeatBlock(int32_t start, int32_t end, const std::string& ref_seq, int period, const StutterModel* stutter_model)
    : HapBlock(start, end, ref_seq), period_(period), model_(stutter_model->copy()) { assert(period > 0); }
  ~RepeatBlock(){ delete model_; }
  const StutterModel* get_stutter_model() const { return model_; }
  int get_period() const { return period_; }
};

// Candidate-haplotype container.  Borrowed blocks, as in the reference (Haplotype.h:33-40); allele k means the k-th
// haplotype in the visit order of the reference's Haplotype::next() (the device uses the same order).
class Haplotype {
  std::vector<HapBlock*> blocks_;
  int ncombs_;
 public:
  explicit Haplotype(std::vector<HapBlock*>& blocks) : blocks_(blocks), ncombs_(1) {
    if (blocks_.size() != HIPSTR_NUM_BLOCKS) printErrorAndDie("Haplotype must consist of [flank, repeat, flank] blocks");   // Haplotype.cpp:12
    if (blocks_[0]->get_stutter_model() != NULL || blocks_[1]->get_stutter_model() == NULL || blocks_[2]->get_stutter_model() != NULL)
      printErrorAndDie("Haplotype must consist of [flank, repeat, flank] blocks");
    for (size_t i = 0; i < blocks_.size(); i++) ncombs_ *= blocks_[i]->num_options();
  }
  int num_blocks() const { return (int)blocks_.size(); }
  int num_combs()  const { return ncombs_; }
  HapBlock* get_block(int i)   const { return blocks_[i]; }
  HapBlock* get_first_block()  const { return blocks_.front(); }
  HapBlock* get_last_block()   const { return blocks_.back();  }
  // Haplotype::get_aln_info() (Haplotype.h:65) per haplotype index.  The reference derives these strings in its constructor
  // (Needleman-Wunsch of every haplotype against the reference haplotype + adjust_indels, Haplotype.cpp:8-86); here
  // HapAligner::trace_optimal_alns asks the library for them on first use (hipstr_hap_aln_info).  set_aln_info overrides.
  void set_aln_info(const std::vector<std::string>& hap_aln_info){
    if ((int)hap_aln_info.size() != ncombs_) printErrorAndDie("set_aln_info needs one string per haplotype");
    hap_aln_info_ = hap_aln_info;
  }
  bool has_aln_info() const { return !hap_aln_info_.empty(); }
  const std::string& get_aln_info(int hap_index) const { return hap_aln_info_[hap_index]; }
  // The cursor of the reference's Haplotype (Haplotype.h:93-117): haplotypes are visited in index order (the index IS the position in
  // the reflected Gray code of Haplotype::next); a fixed haplotype has no next one.  HapAligner::process_read runs from the
  // current haplotype to the last (HapAligner.cpp:613-692).
  void reset(){ cur_ = 0; }
  bool next(){ if (fixed_ || cur_ + 1 >= ncombs_) return false; cur_++; return true; }
  void go_to(int hap_index){ if (hap_index < 0 || hap_index >= ncombs_) printErrorAndDie("Invalid haplotype index"); cur_ = hap_index; }
  void fix(){ fixed_ = true; }
  void unfix(){ fixed_ = false; }
  int cur_index() const { return cur_; }
 private:
  std::vector<std::string> hap_aln_info_;
  int cur_ = 0; bool fixed_ = false;
};

class ReadPooler {
  // one record per distinct read sequence, in order of first appearance: the representative alignment and every member's qualities
  struct Pool { Alignment rep; std::vector<std::string> member_quals; };
  std::vector<Pool> pools_;
  std::vector<Alignment> reps_;                                  // filled by pool(): what get_alignments() hands out
  std::unordered_map<std::string, int32_t> index_of_seq_;
  bool closed_ = false;
 public:
  int32_t num_pools() const { return (int32_t)pools_.size(); }
  // the pool index of the read's sequence; a sequence seen for the first time opens a pool around a copy of the alignment
  // (read_pooler.cpp:3-20: same position, cigar and alignment string, name "READPOOL", no qualities until pool())
  int32_t add_alignment(Alignment& aln){
    if (closed_) printErrorAndDie("Cannot call add_alignment function once pool() function has been invoked");
    const std::string& seq = aln.get_sequence();
    const auto ins = index_of_seq_.emplace(seq, (int32_t)pools_.size());
    if (ins.second){
      Pool p{Alignment(aln.get_start(), aln.get_stop(), false, "READPOOL", "", seq, aln.get_alignment()), {}};
      p.rep.set_cigar_list(aln.get_cigar_list());
      pools_.push_back(std::move(p));
    }
    pools_[ins.first->second].member_quals.push_back(aln.get_base_qualities());
    return ins.first->second;
  }
  // every pool's representative gets the per-position median of its members' qualities (read_pooler.h:42-48)
  void pool(const BaseQuality& base_quality){
    reps_.clear(); reps_.reserve(pools_.size());
    for (Pool& p : pools_){
      std::vector<const std::string*> members; members.reserve(p.member_quals.size());
      for (const std::string& q : p.member_quals) members.push_back(&q);
      p.rep.set_base_qualities(base_quality.median_base_qualities(members));
      reps_.push_back(p.rep);
    }
    closed_ = true;
  }
  std::vector<Alignment>& get_alignments(){
    if (!closed_){ reps_.clear(); for (Pool& p : pools_) reps_.push_back(p.rep); }
    return reps_;
  }
};

// AlignmentTrace (AlignmentTraceback.h:10-108): what a traceback reports about one read.
class AlignmentTrace {
  std::string hap_aln_;
  Alignment trace_vs_ref_;
  int flank_ins_size_, flank_del_size_;
  std::vector<int> stutter_size_; std::vector<std::string> str_seq_; std::vector<bool> str_set_;
  std::vector<std::string> flank_seqs_;
  std::vector< std::pair<int32_t,int32_t> > flank_indel_data_;
  std::vector< std::pair<int32_t,char> > flank_snp_data_;
  friend class HapAligner;
 public:
  explicit AlignmentTrace(int num_haplotype_blocks)
    : trace_vs_ref_("TRACE"), flank_ins_size_(0), flank_del_size_(0), stutter_size_(num_haplotype_blocks, 0), str_seq_(num_haplotype_blocks),
      str_set_(num_haplotype_blocks, false), flank_seqs_(num_haplotype_blocks) {}
  int flank_ins_size()         const { return flank_ins_size_; }
  int flank_del_size()         const { return flank_del_size_; }
  const std::string& hap_aln() const { return hap_aln_; }
  Alignment& traced_aln()            { return trace_vs_ref_; }
  const std::vector< std::pair<int32_t,int32_t> >& flank_indel_data() const { return flank_indel_data_; }
  const std::vector< std::pair<int32_t,char> >& flank_snp_data()      const { return flank_snp_data_; }
  bool has_stutter() const {                                    // AlignmentTraceback.h:79-85
    for (size_t i = 0; i < str_set_.size(); i++) if (str_set_[i] && stutter_size_[i] != 0) return true;
    return false;
  }
  bool has_str_data(int block_index) const { return str_set_[block_index]; }     // str_data_[block_index] != NULL
  int stutter_size(int block_index)  const { assert(str_set_[block_index]); return stutter_size_[block_index]; }
  const std::string& str_seq(int block_index)   const { assert(str_set_[block_index]); return str_seq_[block_index]; }
  const std::string& flank_seq(int block_index) const { return flank_seqs_[block_index]; }
};

// Flat one-locus batch built from the classes above (what the C-ABI consumes).
struct FlatLocus {
  std::vector<int32_t> blk_start, blk_end, blk_nopts, opt_off, hap_off, read_off, base_off, read_start, cigar_off, cigar_len;
  std::vector<double> stutter;
  std::vector<uint8_t> realign_hap, realign_read;
  std::string seq, bases, quals, cigar_op;
  int32_t period;
  hipstr_batch_t batch;
  void set_haplotype(const Haplotype* h, const std::vector<bool>& realign_to_hap){
    opt_off.assign(1, 0); hap_off.assign(1, 0);
    for (int i = 0; i < h->num_blocks(); i++){
      const HapBlock* b = h->get_block(i);
      blk_start.push_back(b->start()); blk_end.push_back(b->end()); blk_nopts.push_back(b->num_options());
      for (int o = 0; o < b->num_options(); o++){ seq += b->get_seq(o); opt_off.push_back((int32_t)seq.size()); }
    }
    const HapBlock* rb = h->get_block(1);
    period = rb->get_period();
    stutter.assign(rb->get_stutter_model()->parameters(), rb->get_stutter_model()->parameters() + 6);
    hap_off.push_back(h->num_combs());
    realign_hap.resize(realign_to_hap.size());
    for (size_t k = 0; k < realign_to_hap.size(); k++) realign_hap[k] = realign_to_hap[k] ? 1 : 0;
  }
  void set_reads(const std::vector<Alignment>& alns, const std::vector<bool>& realign){
    read_off.assign(1, 0); base_off.assign(1, 0); cigar_off.assign(1, 0);
    for (size_t i = 0; i < alns.size(); i++){
      const Alignment& a = alns[i];
      if (a.get_sequence().size() != a.get_base_qualities().size()) printErrorAndDie("Read sequence and base qualities differ in length");
      bases += a.get_sequence(); quals += a.get_base_qualities();
      base_off.push_back((int32_t)bases.size());
      read_start.push_back(a.get_start());
      for (size_t c = 0; c < a.get_cigar_list().size(); c++){ cigar_op += a.get_cigar_list()[c].get_type(); cigar_len.push_back(a.get_cigar_list()[c].get_num()); }
      cigar_off.push_back((int32_t)cigar_op.size());
      realign_read.push_back(realign[i] ? 1 : 0);
    }
    read_off.push_back((int32_t)alns.size());
  }
  const hipstr_batch_t* finish(){
    if (cigar_len.empty()) cigar_len.push_back(0);
    batch.n_loci = 1;
    batch.blk_start = blk_start.data(); batch.blk_end = blk_end.data(); batch.blk_nopts = blk_nopts.data(); batch.period = &period;
    batch.stutter = stutter.data(); batch.opt_off = opt_off.data(); batch.seq = seq.data(); batch.hap_off = hap_off.data();
    batch.realign_hap = realign_hap.data(); batch.read_off = read_off.data(); batch.base_off = base_off.data();
    batch.bases = bases.data(); batch.quals = quals.data(); batch.read_start = read_start.data(); batch.cigar_off = cigar_off.data();
    batch.cigar_op = cigar_op.data(); batch.cigar_len = cigar_len.data(); batch.realign_read = realign_read.data();
    return &batch;
  }
};

class HapAligner {
  Haplotype* fw_haplotype_;                // borrowed (HapAligner.h:58)
  std::vector<bool> realign_to_hap_;
 public:
  HapAligner(Haplotype* haplotype, std::vector<bool>& realign_to_haplotype) : fw_haplotype_(haplotype), realign_to_hap_(realign_to_haplotype) {
    assert((int)realign_to_haplotype.size() == haplotype->num_combs());
  }
  // 0-based index of the seed base, or -1 if none (HapAligner.h:78-81)
  int calc_seed_base(const Alignment& alignment){
    FlatLocus f; f.set_haplotype(fw_haplotype_, realign_to_hap_);
    f.set_reads(std::vector<Alignment>(1, alignment), std::vector<bool>(1, true));
    int32_t seed = -1;
    if (hipstr_calc_seed_bases(f.finish(), &seed) != 0) printErrorAndDie(hipstr_last_error());
    return seed;
  }
  // Same contract as HapAligner::process_reads (HapAligner.cpp:320-343): row init_read_index+i of aln_probs / entry
  // init_read_index+i of seed_positions is written for every realign_read[i]; other entries are left untouched.
  // The BaseQuality argument is kept for signature compatibility: its tables are constants of the model.
  void process_reads(const std::vector<Alignment>& alignments, int init_read_index, const BaseQuality* /*base_quality*/,
                     const std::vector<bool>& realign_read, double* aln_probs, int* seed_positions){
    assert(alignments.size() == realign_read.size());
    FlatLocus f; f.set_haplotype(fw_haplotype_, realign_to_hap_); f.set_reads(alignments, realign_read);
    if (hipstr_hmm_process_reads(f.finish(), aln_probs + (size_t)init_read_index*fw_haplotype_->num_combs(), seed_positions + init_read_index) != 0)
      printErrorAndDie(hipstr_last_error());
  }
  // HapAligner::trace_optimal_aln (HapAligner.h:88-92, HapAligner.cpp:711-722); the caller owns the result.
  // The read is split at seed_base, the CALLER's choice (HapAligner.h:93), as in the reference — not at a recomputed seed.
  AlignmentTrace* trace_optimal_aln(const Alignment& orig_aln, int seed_base, int best_haplotype, const BaseQuality* base_quality){
    std::vector<AlignmentTrace*> traces;
    trace_optimal_alns(std::vector<Alignment>(1, orig_aln), std::vector<int>(1, seed_base), std::vector<int>(1, best_haplotype), base_quality, traces);
    return traces[0];
  }
  // HapAligner::process_read (HapAligner.h:83, HapAligner.cpp:573-709): one read, split at seed_base, against every haplotype from the
  // haplotype's current position to the last one (a fixed haplotype: just that one); *prob_ptr advances one entry per haplotype
  // visited, entries of haplotypes that are not realigned are left untouched.  With retrace_aln the likeliest haplotype visited (the
  // first one among equals, as the reference's strict > keeps it) is traced into traced_aln.
  void process_read(const Alignment& aln, int seed_base, const BaseQuality* base_quality, bool retrace_aln, double* prob_ptr, AlignmentTrace& traced_aln){
    assert(seed_base != -1);
    std::vector<int> visited;
    do { visited.push_back(fw_haplotype_->cur_index()); } while (fw_haplotype_->next());
    fw_haplotype_->reset();
    std::vector<bool> mask(realign_to_hap_.size(), false);
    for (size_t i = 0; i < visited.size(); i++) mask[visited[i]] = realign_to_hap_[visited[i]];
    FlatLocus f; f.set_haplotype(fw_haplotype_, mask); f.set_reads(std::vector<Alignment>(1, aln), std::vector<bool>(1, true));
    std::vector<double> row(realign_to_hap_.size(), 0.0);
    int32_t seed_in = seed_base, seed_out = -1;
    if (hipstr_hmm_process_reads_seeded(f.finish(), &seed_in, row.data(), &seed_out) != 0) printErrorAndDie(hipstr_last_error());
    double max_LL = -100000000; int best = -1;
    for (size_t i = 0; i < visited.size(); i++, prob_ptr++){
      const int k = visited[i];
      if (!mask[k]) continue;
      *prob_ptr = row[k];
      if (row[k] > max_LL){ max_LL = row[k]; best = k; }
    }
    if (retrace_aln && best >= 0){
      AlignmentTrace* t = trace_optimal_aln(aln, seed_base, best, base_quality);
      traced_aln = *t;
      delete t;
    }
  }
  // All tracebacks of a locus in one launch: request i = alignments[i] against haplotype best_haplotypes[i] (what
  // SeqStutterGenotyper::retrace_alignments, seq_stutter_genotyper.cpp:805-841, asks for one read at a time).  The stitched
  // alignment against the reference (traced_aln()) is produced when the haplotype carries its get_aln_info() strings.
  void trace_optimal_alns(const std::vector<Alignment>& alignments, const std::vector<int>& best_haplotypes, const BaseQuality* base_quality,
                          std::vector<AlignmentTrace*>& traces){
    trace_optimal_alns(alignments, std::vector<int>(), best_haplotypes, base_quality, traces);
  }
  // ... with the seed base of every request (empty: calc_seed_base's value, what process_reads uses)
  void trace_optimal_alns(const std::vector<Alignment>& alignments, const std::vector<int>& seed_bases, const std::vector<int>& best_haplotypes,
                          const BaseQuality* /*base_quality*/, std::vector<AlignmentTrace*>& traces){
    assert(alignments.size() == best_haplotypes.size() && (seed_bases.empty() || seed_bases.size() == alignments.size()));
    traces.clear();
    std::vector<Alignment> alns; std::vector<int32_t> req_read, req_allele, req_seed;
    for (size_t i = 0; i < alignments.size(); i++)          // a masked haplotype is skipped by process_read: empty trace (HapAligner.cpp:614-618)
      if (realign_to_hap_[best_haplotypes[i]]){
        req_read.push_back((int32_t)alns.size()); req_allele.push_back(best_haplotypes[i]); alns.push_back(alignments[i]);
        req_seed.push_back(seed_bases.empty() ? HIPSTR_SEED_AUTO : seed_bases[i]);
      }
    const int n = (int)alns.size();
    FlatLocus f; f.set_haplotype(fw_haplotype_, realign_to_hap_); f.set_reads(alns, std::vector<bool>(alns.size(), true));
    std::vector<const char*> hap_to_ref;
    size_t chars = 64;
    if (!fw_haplotype_->has_aln_info()){                     // Haplotype::aln_haps_to_ref (Haplotype.cpp:58-86), once per haplotype set
      FlatLocus g; g.set_haplotype(fw_haplotype_, realign_to_hap_); g.set_reads(std::vector<Alignment>(), std::vector<bool>());
      size_t cap = 64;
      for (int i = 0; i < fw_haplotype_->num_blocks(); i++){
        size_t longest = 0;
        for (int o2 = 0; o2 < fw_haplotype_->get_block(i)->num_options(); o2++) longest = std::max(longest, fw_haplotype_->get_block(i)->get_seq(o2).size());
        cap += 2*longest;
      }
      cap = (cap + 1)*(size_t)fw_haplotype_->num_combs();
      std::vector<char> pool(cap); std::vector<int64_t> offs(fw_haplotype_->num_combs() + 1);
      if (hipstr_hap_aln_info(g.finish(), pool.data(), (int64_t)cap, offs.data()) != 0) printErrorAndDie(hipstr_last_error());
      std::vector<std::string> info;
      for (int k = 0; k < fw_haplotype_->num_combs(); k++) info.push_back(std::string(pool.data() + offs[k]));
      fw_haplotype_->set_aln_info(info);
    }
    if (fw_haplotype_->has_aln_info())
      for (int k = 0; k < fw_haplotype_->num_combs(); k++) hap_to_ref.push_back(fw_haplotype_->get_aln_info(k).c_str());
    for (int i = 0; i < n; i++)
      chars += 2*alns[i].get_sequence().size() + 64 + (hap_to_ref.empty() ? 0 : 2*fw_haplotype_->get_aln_info(req_allele[i]).size());
    const int32_t cap = (int32_t)chars;
    std::vector<double> ll(n + 1);
    std::vector<int32_t> max_index(n + 1), hap_aln_off(n + 1), stutter_size(n + 1), str_seq_off(n + 1), flank_seq_off(2*n + 1), flank_ins(n + 1), flank_del(n + 1),
      indel_off(n + 1), indel_pos(cap), indel_size(cap), snp_off(n + 1), snp_pos(cap), aln_start(n + 1), aln_stop(n + 1), cigar_off(n + 1), cigar_len(cap), aln_str_off(n + 1);
    std::vector<char> hap_aln(cap), str_seq(cap), flank_seq(cap), snp_base(cap), cigar_op(cap), aln_str(cap);
    hipstr_trace_out_t o;
    o.ll = ll.data(); o.max_index = max_index.data(); o.hap_aln_off = hap_aln_off.data(); o.hap_aln = hap_aln.data();
    o.stutter_size = stutter_size.data(); o.str_seq_off = str_seq_off.data(); o.str_seq = str_seq.data();
    o.flank_seq_off = flank_seq_off.data(); o.flank_seq = flank_seq.data(); o.flank_ins = flank_ins.data(); o.flank_del = flank_del.data();
    o.indel_off = indel_off.data(); o.indel_pos = indel_pos.data(); o.indel_size = indel_size.data();
    o.snp_off = snp_off.data(); o.snp_pos = snp_pos.data(); o.snp_base = snp_base.data();
    o.aln_start = aln_start.data(); o.aln_stop = aln_stop.data(); o.cigar_off = cigar_off.data(); o.cigar_op = cigar_op.data(); o.cigar_len = cigar_len.data();
    o.aln_str_off = aln_str_off.data(); o.aln_str = aln_str.data(); o.cap_chars = cap;
    if (n > 0 && hipstr_hmm_trace_seeded(f.finish(), n, req_read.data(), req_allele.data(), req_seed.data(), hap_to_ref.empty() ? NULL : hap_to_ref.data(), &o) != 0)
      printErrorAndDie(hipstr_last_error());
    for (size_t i = 0, q = 0; i < alignments.size(); i++){
      AlignmentTrace* t = new AlignmentTrace(fw_haplotype_->num_blocks());
      traces.push_back(t);
      if (!realign_to_hap_[best_haplotypes[i]]) continue;
      t->hap_aln_.assign(hap_aln.data() + hap_aln_off[q], hap_aln_off[q+1] - hap_aln_off[q]);
      if (stutter_size[q] != HIPSTR_NO_STR_DATA){
        t->str_set_[1] = true; t->stutter_size_[1] = stutter_size[q];
        t->str_seq_[1].assign(str_seq.data() + str_seq_off[q], str_seq_off[q+1] - str_seq_off[q]);
      }
      t->flank_seqs_[0].assign(flank_seq.data() + flank_seq_off[2*q], flank_seq_off[2*q+1] - flank_seq_off[2*q]);
      t->flank_seqs_[2].assign(flank_seq.data() + flank_seq_off[2*q+1], flank_seq_off[2*q+2] - flank_seq_off[2*q+1]);
      t->flank_ins_size_ = flank_ins[q]; t->flank_del_size_ = flank_del[q];
      for (int k = indel_off[q]; k < indel_off[q+1]; k++) t->flank_indel_data_.push_back(std::pair<int32_t,int32_t>(indel_pos[k], indel_size[k]));
      for (int k = snp_off[q]; k < snp_off[q+1]; k++) t->flank_snp_data_.push_back(std::pair<int32_t,char>(snp_pos[k], snp_base[k]));
      if (!hap_to_ref.empty()){
        t->trace_vs_ref_ = Alignment(aln_start[q], aln_stop[q], false, "TRACE", alignments[i].get_base_qualities(), alignments[i].get_sequence(),
                                     std::string(aln_str.data() + aln_str_off[q], aln_str_off[q+1] - aln_str_off[q]));
        std::vector<CigarElement> cigar_list;
        for (int k = cigar_off[q]; k < cigar_off[q+1]; k++) cigar_list.push_back(CigarElement(cigar_op[k], cigar_len[k]));
        t->trace_vs_ref_.set_cigar_list(cigar_list);
      }
      q++;
    }
  }
};

// SeqStutterGenotyper::calc_hap_aln_probs (seq_stutter_genotyper.cpp:519-568): align the pooled reads, copy each pool's row
// to its reads for the realigned haplotypes, and give both mates of a pair the sum of their rows (only for newly aligned
// haplotypes, "or we'll effectively keep doubling those values with each iteration").
inline void calc_hap_aln_probs(Haplotype* haplotype, ReadPooler& pooler, const BaseQuality& base_quality,
                               const int* pool_index, const bool* second_mate, unsigned int num_reads,
                               std::vector<bool>& realign_to_haplotype, std::vector<bool>& realign_pool, std::vector<bool>& copy_read,
                               double* log_aln_probs, int* seed_positions){
  const int num_alleles = haplotype->num_combs();
  assert((int)realign_to_haplotype.size() == num_alleles);
  HapAligner hap_aligner(haplotype, realign_to_haplotype);
  std::vector<Alignment>& pooled_alns = pooler.get_alignments();
  std::vector<double> pool_probs(pooled_alns.size()*(size_t)num_alleles);
  std::vector<int> pool_seeds(pooled_alns.size());
  hap_aligner.process_reads(pooled_alns, 0, &base_quality, realign_pool, pool_probs.data(), pool_seeds.data());
  for (unsigned int i = 0; i < num_reads; i++){
    if (!copy_read[i]) continue;
    seed_positions[i] = pool_seeds[pool_index[i]];
    for (int j = 0; j < num_alleles; j++)
      if (realign_to_haplotype[j]) log_aln_probs[(size_t)i*num_alleles + j] = pool_probs[(size_t)pool_index[i]*num_alleles + j];
  }
  for (unsigned int i = 0; i < num_reads; i++){
    if (!second_mate[i] || !copy_read[i]) continue;
    for (int j = 0; j < num_alleles; j++)
      if (realign_to_haplotype[j]){
        const double total = log_aln_probs[(size_t)(i-1)*num_alleles + j] + log_aln_probs[(size_t)i*num_alleles + j];
        log_aln_probs[(size_t)(i-1)*num_alleles + j] = total;
        log_aln_probs[(size_t)i*num_alleles + j]     = total;
      }
  }
}

// Posterior half of the reference's Genotyper base class (genotyper.h:14-130): flat per-read arrays owned by the object,
// calc_log_sample_posteriors() fills log_sample_posteriors_ / sample_total_LLs_ and returns the total log-likelihood.
class Genotyper {
 protected:
  unsigned int num_reads_; int num_samples_, num_alleles_;
  double *log_p1_, *log_p2_; int* sample_label_; bool haploid_;
  double* log_sample_posteriors_;   // samples, then allele_1, then allele_2
  double* log_aln_probs_;           // reads, then alleles
  double* sample_total_LLs_;
  std::vector<int> read_weights_;
  std::vector< std::pair<int,int> > map_gts_;
  bool custom_priors_;              // set by a derived class that overrides init_log_sample_priors

  // Default: leave the array untouched and let the device apply the hom/het priors of genotyper.cpp:20-42.
  // A derived class (the reference's EMStutterGenotyper does, em_stutter_genotyper.cpp:129-144) overrides this,
  // fills log_sample_ptr[samples][allele_1][allele_2] and sets custom_priors_ = true.
  virtual void init_log_sample_priors(double* /*log_sample_ptr*/){}

  double calc_log_sample_posteriors(std::vector<int>& read_weights){
    assert(read_weights.size() == num_reads_ && log_sample_posteriors_ != NULL && log_aln_probs_ != NULL);
    int32_t A = num_alleles_, S = num_samples_, read_off[2] = {0, (int32_t)num_reads_};
    uint8_t hap = haploid_ ? 1 : 0;
    std::vector<int32_t> w(read_weights.begin(), read_weights.end()), gt(2*(size_t)S);
    hipstr_post_batch_t pb;
    pb.n_loci = 1; pb.n_alleles = &A; pb.n_samples = &S; pb.read_off = read_off; pb.sample_label = sample_label_;
    pb.log_p1 = log_p1_; pb.log_p2 = log_p2_; pb.read_weight = w.data(); pb.log_aln_probs = log_aln_probs_; pb.haploid = &hap;
    std::vector<double> prior((size_t)S*A*A);
    init_log_sample_priors(prior.data());        // virtual, as in the reference (genotyper.h:69)
    pb.log_prior = custom_priors_ ? prior.data() : NULL;
    double total = 0;
    if (hipstr_post_run(&pb, NULL, log_sample_posteriors_, sample_total_LLs_, gt.data(), &total) != 0) printErrorAndDie(hipstr_last_error());
    map_gts_.resize(S);
    for (int s = 0; s < S; s++) map_gts_[s] = std::pair<int,int>(gt[2*s], gt[2*s+1]);
    return total;
  }
  double calc_log_sample_posteriors(){ return calc_log_sample_posteriors(read_weights_); }

  // MAP phased diplotype per sample: first maximum in allele_1-major order (genotyper.cpp:82-97); computed by the same kernel
  void get_optimal_haplotypes(std::vector< std::pair<int,int> >& gts) const { assert(gts.empty()); gts = map_gts_; }

  // Genotyper::extract_genotypes_and_likelihoods (genotyper.h:98-106, genotyper.cpp:129-251): same arguments, same outputs.
  // The posteriors are those of the current log_aln_probs_ / read_weights_ (recomputed on the device, where they stay).
  void extract_genotypes_and_likelihoods(int num_variants, std::vector<int>& hap_to_allele,
                                         std::vector< std::pair<int,int> >& best_haplotypes, std::vector< std::pair<int,int> >& best_gts,
                                         std::vector<double>& log_phased_posteriors, std::vector<double>& log_unphased_posteriors,
                                         std::vector<double>& hap_log_phased_posteriors, std::vector<double>& hap_log_unphased_posteriors,
                                         bool calc_gls, std::vector< std::vector<double> >& gls, std::vector<double>& gl_diffs,
                                         bool calc_pls, std::vector< std::vector<int> >& pls,
                                         bool calc_phased_gls, std::vector< std::vector<double> >& phased_gls){
    assert(log_phased_posteriors.empty() && log_unphased_posteriors.empty() && gl_diffs.empty());
    assert(best_haplotypes.empty() && best_gts.empty() && gls.empty() && pls.empty() && phased_gls.empty());
    assert((int)hap_to_allele.size() == num_alleles_);
    int32_t A = num_alleles_, S = num_samples_, V = num_variants, read_off[2] = {0, (int32_t)num_reads_};
    uint8_t hap = haploid_ ? 1 : 0;
    std::vector<int32_t> w(read_weights_.begin(), read_weights_.end()), h2a(hap_to_allele.begin(), hap_to_allele.end());
    hipstr_post_batch_t pb;
    pb.n_loci = 1; pb.n_alleles = &A; pb.n_samples = &S; pb.read_off = read_off; pb.sample_label = sample_label_;
    pb.log_p1 = log_p1_; pb.log_p2 = log_p2_; pb.read_weight = w.data(); pb.log_aln_probs = log_aln_probs_; pb.haploid = &hap;
    std::vector<double> prior((size_t)S*A*A);
    init_log_sample_priors(prior.data());
    pb.log_prior = custom_priors_ ? prior.data() : NULL;
    hipstr_gt_request_t rq; rq.n_variants = &V; rq.hap_to_allele = h2a.data();
    rq.calc_gls = calc_gls; rq.calc_pls = calc_pls; rq.calc_phased_gls = calc_phased_gls;
    const bool any = calc_gls || calc_pls || calc_phased_gls;
    const size_t ngl = haploid_ ? V : (size_t)V*(V+1)/2, npgl = haploid_ ? V : (size_t)V*V;
    std::vector<int32_t> bh(2*(size_t)S), bg(2*(size_t)S), pl(any ? S*ngl : 1);
    std::vector<double> lp(S), lu(S), hlp(S), hlu(S), gd(S), gl(any ? S*ngl : 1), pgl(calc_phased_gls ? S*npgl : 1);
    hipstr_gt_out_t o;
    o.best_hap = bh.data(); o.best_gt = bg.data(); o.log_phased_post = lp.data(); o.log_unphased_post = lu.data();
    o.hap_log_phased_post = hlp.data(); o.hap_log_unphased_post = hlu.data(); o.gl_diff = gd.data();
    o.gls = gl.data(); o.pls = pl.data(); o.phased_gls = pgl.data();
    rq.calc_gls = any;                       // GLDIFF needs the GLs; they are dropped below when not asked for (genotyper.cpp:247-248)
    hipstr_post_dev_t* pd = hipstr_post_upload(&pb, NULL);
    if (pd == NULL || hipstr_post_launch(pd, NULL) != 0 || hipstr_post_extract(pd, &rq, &o) != 0) printErrorAndDie(hipstr_last_error());
    hipstr_post_free(pd);
    for (int s = 0; s < S; s++){
      best_haplotypes.push_back(std::pair<int,int>(bh[2*s], bh[2*s+1]));
      best_gts.push_back(std::pair<int,int>(bg[2*s], bg[2*s+1]));
    }
    log_phased_posteriors = lp; log_unphased_posteriors = lu; hap_log_phased_posteriors = hlp; hap_log_unphased_posteriors = hlu;
    if (any){
      gl_diffs = gd;
      if (calc_gls) for (int s = 0; s < S; s++) gls.push_back(std::vector<double>(gl.begin() + s*ngl, gl.begin() + (s+1)*ngl));
      if (calc_pls) for (int s = 0; s < S; s++) pls.push_back(std::vector<int>(pl.begin() + s*ngl, pl.begin() + (s+1)*ngl));
      if (calc_phased_gls) for (int s = 0; s < S; s++) phased_gls.push_back(std::vector<double>(pgl.begin() + s*npgl, pgl.begin() + (s+1)*npgl));
    }
  }

 public:
  Genotyper(bool haploid, const std::vector<std::string>& sample_names,
            const std::vector< std::vector<double> >& log_p1, const std::vector< std::vector<double> >& log_p2){
    assert(log_p1.size() == log_p2.size() && log_p1.size() == sample_names.size());
    num_reads_ = 0;
    for (size_t i = 0; i < log_p1.size(); i++) num_reads_ += log_p1[i].size();
    num_alleles_ = -1; haploid_ = haploid; num_samples_ = (int)log_p1.size();
    log_p1_ = new double[num_reads_]; log_p2_ = new double[num_reads_]; sample_label_ = new int[num_reads_];
    sample_total_LLs_ = new double[num_samples_];
    read_weights_ = std::vector<int>(num_reads_, 1);
    unsigned int r = 0;
    for (size_t i = 0; i < log_p1.size(); i++)
      for (size_t j = 0; j < log_p1[i].size(); j++, r++){
        assert(log_p1[i][j] <= 0.0 && log_p2[i][j] <= 0.0);
        log_p1_[r] = log_p1[i][j]; log_p2_[r] = log_p2[i][j]; sample_label_[r] = (int)i;
      }
    log_sample_posteriors_ = NULL; log_aln_probs_ = NULL; custom_priors_ = false;
  }
  virtual ~Genotyper(){
    delete [] log_p1_; delete [] log_p2_; delete [] sample_label_; delete [] sample_total_LLs_;
    delete [] log_sample_posteriors_; delete [] log_aln_probs_;
  }
};

// EMStutterGenotyper (em_stutter_genotyper.h:15-127): length-based EM that learns a locus' stutter model.  Same constructor and
// train()/get_stutter_model() as the reference; the whole EM loop runs in the library (hipstr_em_train).
class EMStutterGenotyper {
  bool haploid_; int motif_len_, ref_allele_;
  std::vector<int32_t> n_bps_, label_; std::vector<double> p1_, p2_; int32_t num_samples_;
  StutterModel* stutter_model_;
  int num_iter_; double final_LL_;
  EMStutterGenotyper(const EMStutterGenotyper&); EMStutterGenotyper& operator=(const EMStutterGenotyper&);
 public:
  EMStutterGenotyper(bool haploid, int motif_length, const std::vector< std::vector<int> >& num_bps,
                     const std::vector< std::vector<double> >& log_p1, const std::vector< std::vector<double> >& log_p2,
                     const std::vector<std::string>& sample_names, int ref_allele)
    : haploid_(haploid), motif_len_(motif_length), ref_allele_(ref_allele), num_samples_((int32_t)num_bps.size()), stutter_model_(NULL), num_iter_(0), final_LL_(0) {
    assert(num_bps.size() == log_p1.size() && num_bps.size() == log_p2.size() && num_bps.size() == sample_names.size());
    for (size_t i = 0; i < num_bps.size(); i++){
      assert(num_bps[i].size() == log_p1[i].size() && num_bps[i].size() == log_p2[i].size());
      for (size_t j = 0; j < num_bps[i].size(); j++){
        assert(log_p1[i][j] <= 0.0 && log_p2[i][j] <= 0.0);
        n_bps_.push_back(num_bps[i][j]); label_.push_back((int32_t)i); p1_.push_back(log_p1[i][j]); p2_.push_back(log_p2[i][j]);
      }
    }
  }
  ~EMStutterGenotyper(){ delete stutter_model_; }
  // EMStutterGenotyper::train (em_stutter_genotyper.cpp:171-226); disp_stats/logger are accepted for signature compatibility
  bool train(int max_iter, double min_LL_abs_change, double min_LL_frac_change, bool /*disp_stats*/, std::ostream& /*logger*/){
    int32_t period = motif_len_, read_off[2] = {0, (int32_t)n_bps_.size()}; uint8_t hap = haploid_ ? 1 : 0, ok = 0;
    hipstr_em_batch_t eb;
    eb.n_loci = 1; eb.period = &period; eb.haploid = &hap; eb.n_samples = &num_samples_; eb.read_off = read_off;
    eb.sample_label = label_.data(); eb.num_bps = n_bps_.data(); eb.log_p1 = p1_.data(); eb.log_p2 = p2_.data();
    eb.ref_allele = ref_allele_; eb.max_iter = max_iter; eb.min_ll_abs_change = min_LL_abs_change; eb.min_ll_frac_change = min_LL_frac_change;
    double sp[6];
    if (hipstr_em_train(&eb, &ok, sp, &num_iter_, &final_LL_) != 0) printErrorAndDie(hipstr_last_error());
    delete stutter_model_;
    stutter_model_ = new StutterModel(sp[0], sp[1], sp[2], sp[3], sp[4], sp[5], motif_len_);
    return ok != 0;
  }
  StutterModel* get_stutter_model() const {
    if (stutter_model_ == NULL) printErrorAndDie("No stutter model has been specified or learned");
    return stutter_model_;
  }
  int num_iterations() const { return num_iter_; }      // extras: E-steps performed and the last total log-likelihood
  double final_LL()    const { return final_LL_; }
};

}  // namespace hipstr_amd
#endif  // HIPSTR_HMM_HPP_
