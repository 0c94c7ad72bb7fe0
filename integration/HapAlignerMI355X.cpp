// HapAlignerMI355X.cpp — see HapAlignerMI355X.h.  Flattens the reference's objects into the C-ABI batch.
#include <assert.h>

#include "HapAlignerMI355X.h"
#include "RepeatBlock.h"
#include "../error.h"
#include "../stutter_model.h"

#include "hipstr_hmm.h"

HapAlignerMI355X::HapAlignerMI355X(Haplotype* haplotype, std::vector<bool>& realign_to_haplotype)
  : fw_haplotype_(haplotype), realign_to_hap_(realign_to_haplotype){
  assert(realign_to_haplotype.size() == (size_t)haplotype->num_combs());
  if (haplotype->num_blocks() != HIPSTR_NUM_BLOCKS)
    printErrorAndDie("HapAlignerMI355X requires a [flank, repeat, flank] haplotype");
  opt_off_.push_back(0);
  period_ = 0;
  for (int i = 0; i < haplotype->num_blocks(); i++){
    HapBlock* block = haplotype->get_block(i);
    blk_start_.push_back(block->start());
    blk_end_.push_back(block->end());
    blk_nopts_.push_back(block->num_options());
    for (int o = 0; o < block->num_options(); o++){
      seq_ += block->get_seq(o);
      opt_off_.push_back((int32_t)seq_.size());
    }
    RepeatStutterInfo* info = block->get_repeat_info();
    if ((i == 1) != (info != NULL))
      printErrorAndDie("HapAlignerMI355X requires a [flank, repeat, flank] haplotype");
    if (info != NULL){
      period_ = info->get_period();
      StutterModel* m = info->get_stutter_model();
      stutter_.push_back(m->get_parameter(true,  'P')); stutter_.push_back(m->get_parameter(true,  'U')); stutter_.push_back(m->get_parameter(true,  'D'));
      stutter_.push_back(m->get_parameter(false, 'P')); stutter_.push_back(m->get_parameter(false, 'U')); stutter_.push_back(m->get_parameter(false, 'D'));
    }
  }
  hap_off_.push_back(0);
  hap_off_.push_back(haplotype->num_combs());
  for (size_t k = 0; k < realign_to_hap_.size(); k++)
    realign_hap_.push_back(realign_to_hap_[k] ? 1 : 0);
}

namespace {
struct FlatReads {
  std::vector<int32_t> read_off, base_off, read_start, cigar_off, cigar_len;
  std::vector<uint8_t> realign;
  std::string bases, quals, cigar_op;
  FlatReads(const std::vector<Alignment>& alns, const std::vector<bool>& realign_read){
    read_off.push_back(0); base_off.push_back(0); cigar_off.push_back(0);
    for (size_t i = 0; i < alns.size(); i++){
      bases += alns[i].get_sequence();
      quals += alns[i].get_base_qualities();
      base_off.push_back((int32_t)bases.size());
      read_start.push_back(alns[i].get_start());
      const std::vector<CigarElement>& cig = alns[i].get_cigar_list();
      for (size_t c = 0; c < cig.size(); c++){ cigar_op += cig[c].get_type(); cigar_len.push_back(cig[c].get_num()); }
      cigar_off.push_back((int32_t)cigar_op.size());
      realign.push_back(realign_read[i] ? 1 : 0);
    }
    read_off.push_back((int32_t)alns.size());
    if (cigar_len.empty()) cigar_len.push_back(0);
  }
};
}

#define FILL_BATCH(b, r)                                                                                          \
  hipstr_batch_t b;                                                                                              \
  b.n_loci = 1; b.blk_start = blk_start_.data(); b.blk_end = blk_end_.data(); b.blk_nopts = blk_nopts_.data();   \
  b.period = &period_; b.stutter = stutter_.data(); b.opt_off = opt_off_.data(); b.seq = seq_.data();            \
  b.hap_off = hap_off_.data(); b.realign_hap = realign_hap_.data(); b.read_off = r.read_off.data();              \
  b.base_off = r.base_off.data(); b.bases = r.bases.data(); b.quals = r.quals.data();                            \
  b.read_start = r.read_start.data(); b.cigar_off = r.cigar_off.data(); b.cigar_op = r.cigar_op.data();          \
  b.cigar_len = r.cigar_len.data(); b.realign_read = r.realign.data();

int HapAlignerMI355X::calc_seed_base(const Alignment& alignment){
  FlatReads r(std::vector<Alignment>(1, alignment), std::vector<bool>(1, true));
  FILL_BATCH(b, r)
  int32_t seed = -1;
  if (hipstr_calc_seed_bases(&b, &seed) != 0)
    printErrorAndDie(hipstr_last_error());
  return seed;
}

// ---- where the adapter's own host time goes (genotype_flow --profile): seconds summed over threads
#include <atomic>
#include <chrono>
static std::atomic<long long> g_adapter_ns[3];       // 0 flatten (FlatReads + batch struct)  1 fill AlignmentTrace objects  2 haplotype strings for the traces
static std::atomic<bool> g_adapter_prof(false);
namespace {
struct AdapterTimer {
  int k; std::chrono::steady_clock::time_point t0; bool on;
  explicit AdapterTimer(int k_) : k(k_), on(g_adapter_prof.load()) { if (on) t0 = std::chrono::steady_clock::now(); }
  void stop(){ if (on){ g_adapter_ns[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); on = false; } }
  ~AdapterTimer(){ stop(); }
};
}
void HapAlignerMI355X::profile(bool enable, double seconds[3]){
  if (seconds) for (int i = 0; i < 3; i++) seconds[i] = 1e-9*(double)g_adapter_ns[i].load();
  if (enable) for (int i = 0; i < 3; i++) g_adapter_ns[i] = 0;
  g_adapter_prof = enable;
}

static hipstr_stream_t* g_shared_stream = NULL;
void HapAlignerMI355X::use_stream(hipstr_stream* stream){ g_shared_stream = stream; }

void HapAlignerMI355X::process_reads(const std::vector<Alignment>& alignments, int init_read_index, const BaseQuality* base_quality,
				     const std::vector<bool>& realign_read, double* aln_probs, int* seed_positions){
  assert(alignments.size() == realign_read.size());
  (void)base_quality;     // BaseQuality's tables are constants of the model; the device holds the same values
  AdapterTimer t_flat(0);
  FlatReads r(alignments, realign_read);
  FILL_BATCH(b, r)
  t_flat.stop();
  if (g_shared_stream != NULL){      // this locus' round joins whatever the other loci in flight submitted
    const int64_t ticket = hipstr_stream_submit(g_shared_stream, &b);
    if (ticket < 0 || hipstr_stream_take(g_shared_stream, ticket, aln_probs + (size_t)init_read_index*fw_haplotype_->num_combs(),
					 (int64_t)alignments.size()*fw_haplotype_->num_combs(), seed_positions + init_read_index, (int64_t)alignments.size()) != 0)
      printErrorAndDie(hipstr_last_error());
    return;
  }
  if (hipstr_hmm_process_reads(&b, aln_probs + (size_t)init_read_index*fw_haplotype_->num_combs(), seed_positions + init_read_index) != 0)
    printErrorAndDie(hipstr_last_error());
}

void HapAlignerMI355X::process_read(const Alignment& aln, int seed_base, const BaseQuality* base_quality, bool retrace_aln,
				    double* prob_ptr, AlignmentTrace& traced_aln){
  assert(seed_base != -1);
  assert(aln.get_sequence().size() == aln.get_base_qualities().size());
  (void)base_quality;
  // the haplotypes the reference's do/while visits (HapAligner.cpp:613-692): from the current one to the last, one if fixed
  std::vector<int> visited;
  do { visited.push_back(fw_haplotype_->cur_index()); } while (fw_haplotype_->next());
  fw_haplotype_->reset();
  FlatReads r(std::vector<Alignment>(1, aln), std::vector<bool>(1, true));
  FILL_BATCH(b, r)
  std::vector<uint8_t> mask(realign_hap_.size(), 0);
  for (size_t i = 0; i < visited.size(); i++) mask[visited[i]] = realign_hap_[visited[i]];
  b.realign_hap = mask.data();
  std::vector<double> row(realign_hap_.size(), 0.0);
  int32_t seed_in = seed_base, seed_out = -1;
  if (hipstr_hmm_process_reads_seeded(&b, &seed_in, row.data(), &seed_out) != 0)
    printErrorAndDie(hipstr_last_error());
  double max_LL = -100000000;
  for (size_t i = 0; i < visited.size(); i++, prob_ptr++){
    const int k = visited[i];
    if (!mask[k]) continue;
    *prob_ptr = row[k];
    if (row[k] > max_LL){
      max_LL = row[k];
      if (retrace_aln)      // filled through the same public mutators the reference's retrace uses (HapAligner.cpp:642-684)
	run_traces(std::vector<Alignment>(1, aln), std::vector<int>(1, seed_base), std::vector<int>(1, k), std::vector<AlignmentTrace*>(1, &traced_aln));
    }
  }
}

AlignmentTrace* HapAlignerMI355X::trace_optimal_aln(const Alignment& orig_aln, int seed_base, int best_haplotype, const BaseQuality* base_quality){
  std::vector<AlignmentTrace*> traces;
  trace_optimal_alns(std::vector<Alignment>(1, orig_aln), std::vector<int>(1, seed_base), std::vector<int>(1, best_haplotype), base_quality, traces);
  return traces[0];
}

void HapAlignerMI355X::trace_optimal_alns(const std::vector<Alignment>& alignments, const std::vector<int>& seed_bases, const std::vector<int>& best_haplotypes,
					  const BaseQuality* base_quality, std::vector<AlignmentTrace*>& traces){
  assert(alignments.size() == best_haplotypes.size() && (seed_bases.empty() || seed_bases.size() == alignments.size()));
  (void)base_quality;
  traces.clear();
  // A haplotype the aligner was told not to realign to is skipped by process_read (HapAligner.cpp:614-618), which leaves the
  // AlignmentTrace empty; the reference's own callers always trace with an all-true mask (seq_stutter_genotyper.cpp:813).
  {
    bool any_masked = false;
    for (size_t i = 0; i < best_haplotypes.size(); i++) any_masked |= !realign_to_hap_[best_haplotypes[i]];
    if (any_masked){
      std::vector<Alignment> sub_alns; std::vector<int> sub_haps, sub_seeds;
      for (size_t i = 0; i < alignments.size(); i++)
        if (realign_to_hap_[best_haplotypes[i]]){
	  sub_alns.push_back(alignments[i]); sub_haps.push_back(best_haplotypes[i]);
	  if (!seed_bases.empty()) sub_seeds.push_back(seed_bases[i]);
	}
      std::vector<AlignmentTrace*> sub;
      trace_optimal_alns(sub_alns, sub_seeds, sub_haps, base_quality, sub);
      for (size_t i = 0, k = 0; i < alignments.size(); i++)
        traces.push_back(realign_to_hap_[best_haplotypes[i]] ? sub[k++] : new AlignmentTrace(fw_haplotype_->num_blocks()));
      return;
    }
  }
  for (size_t i = 0; i < alignments.size(); i++) traces.push_back(new AlignmentTrace(fw_haplotype_->num_blocks()));
  run_traces(alignments, seed_bases, best_haplotypes, traces);
}

// One hipstr_hmm_trace_seeded call for all requests; targets[i] receives request i.
void HapAlignerMI355X::run_traces(const std::vector<Alignment>& alignments, const std::vector<int>& seed_bases, const std::vector<int>& best_haplotypes,
				  const std::vector<AlignmentTrace*>& targets){
  const int n = (int)alignments.size();
  if (n == 0) return;
  AdapterTimer t_flat(0);
  FlatReads r(alignments, std::vector<bool>(alignments.size(), true));
  FILL_BATCH(b, r)
  t_flat.stop();
  AdapterTimer t_info(2);

  // Haplotype::get_aln_info() of every haplotype, in the order the reference visits them (Haplotype::next)
  std::vector<std::string> aln_info;
  fw_haplotype_->reset();
  do { aln_info.push_back(fw_haplotype_->get_aln_info()); } while (fw_haplotype_->next());
  fw_haplotype_->reset();
  std::vector<const char*> hap_to_ref;
  for (size_t k = 0; k < aln_info.size(); k++) hap_to_ref.push_back(aln_info[k].c_str());

  t_info.stop();
  std::vector<int32_t> req_read(n), req_allele(best_haplotypes.begin(), best_haplotypes.end()), req_seed(n, HIPSTR_SEED_AUTO);
  size_t chars = 64;
  for (int i = 0; i < n; i++){
    req_read[i] = i; chars += 2*alignments[i].get_sequence().size() + 2*aln_info[best_haplotypes[i]].size() + 64;
    if (!seed_bases.empty()) req_seed[i] = seed_bases[i];           // the caller's seed_base (HapAligner.h:93), not a recomputed one
  }
  const int32_t cap = (int32_t)chars;
  std::vector<double> ll(n);
  std::vector<int32_t> max_index(n), hap_aln_off(n+1), stutter_size(n), str_seq_off(n+1), flank_seq_off(2*n+1), flank_ins(n), flank_del(n),
    indel_off(n+1), indel_pos(cap), indel_size(cap), snp_off(n+1), snp_pos(cap), aln_start(n), aln_stop(n), cigar_off(n+1), cigar_len(cap), aln_str_off(n+1);
  std::vector<char> hap_aln(cap), str_seq(cap), flank_seq(cap), snp_base(cap), cigar_op(cap), aln_str(cap);
  hipstr_trace_out_t o;
  o.ll = ll.data(); o.max_index = max_index.data(); o.hap_aln_off = hap_aln_off.data(); o.hap_aln = hap_aln.data();
  o.stutter_size = stutter_size.data(); o.str_seq_off = str_seq_off.data(); o.str_seq = str_seq.data();
  o.flank_seq_off = flank_seq_off.data(); o.flank_seq = flank_seq.data(); o.flank_ins = flank_ins.data(); o.flank_del = flank_del.data();
  o.indel_off = indel_off.data(); o.indel_pos = indel_pos.data(); o.indel_size = indel_size.data();
  o.snp_off = snp_off.data(); o.snp_pos = snp_pos.data(); o.snp_base = snp_base.data();
  o.aln_start = aln_start.data(); o.aln_stop = aln_stop.data(); o.cigar_off = cigar_off.data(); o.cigar_op = cigar_op.data(); o.cigar_len = cigar_len.data();
  o.aln_str_off = aln_str_off.data(); o.aln_str = aln_str.data(); o.cap_chars = cap;
  if (hipstr_hmm_trace_seeded(&b, n, req_read.data(), req_allele.data(), req_seed.data(), hap_to_ref.data(), &o) != 0)
    printErrorAndDie(hipstr_last_error());
  AdapterTimer t_fill(1);
  for (int i = 0; i < n; i++) fill_trace(i, &o, alignments[i], *targets[i]);
}

void HapAlignerMI355X::fill_trace(int i, const hipstr_trace_out_t* o, const Alignment& orig, AlignmentTrace& t) const {
  std::string s(o->hap_aln + o->hap_aln_off[i], o->hap_aln_off[i+1] - o->hap_aln_off[i]);
  t.set_hap_aln(s);
  if (o->stutter_size[i] != HIPSTR_NO_STR_DATA){
    std::string ss(o->str_seq + o->str_seq_off[i], o->str_seq_off[i+1] - o->str_seq_off[i]);
    t.add_str_data(1, o->stutter_size[i], ss);
  }
  for (int side = 0; side < 2; side++){
    std::string fs(o->flank_seq + o->flank_seq_off[2*i+side], o->flank_seq_off[2*i+side+1] - o->flank_seq_off[2*i+side]);
    t.add_flank_data(side == 0 ? 0 : 2, fs);
  }
  for (int k = 0; k < o->flank_ins[i]; k++) t.inc_flank_ins();
  for (int k = 0; k < o->flank_del[i]; k++) t.inc_flank_del();
  for (int k = o->indel_off[i]; k < o->indel_off[i+1]; k++) t.add_flank_indel(std::pair<int32_t,int32_t>(o->indel_pos[k], o->indel_size[k]));
  for (int k = o->snp_off[i]; k < o->snp_off[i+1]; k++) t.add_flank_snp(o->snp_pos[k], o->snp_base[k]);
  t.traced_aln() = Alignment(o->aln_start[i], o->aln_stop[i], false, "TRACE", orig.get_base_qualities(), orig.get_sequence(),
			      std::string(o->aln_str + o->aln_str_off[i], o->aln_str_off[i+1] - o->aln_str_off[i]));
  std::vector<CigarElement> cigar_list;
  for (int k = o->cigar_off[i]; k < o->cigar_off[i+1]; k++) cigar_list.push_back(CigarElement(o->cigar_op[k], o->cigar_len[k]));
  t.traced_aln().set_cigar_list(cigar_list);
}
