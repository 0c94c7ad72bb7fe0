// HapAlignerMI355X.cpp — see HapAlignerMI355X.h.  Flattens the reference's objects into the C-ABI batch.
#include <assert.h>

#include "HapAlignerMI355X.h"
#include "RepeatBlock.h"
#include "../error.h"
#include "../stutter_model.h"

#include "hipstr_hmm.h"

HapAlignerMI355X::HapAlignerMI355X(Haplotype* haplotype, std::vector<bool>& realign_to_haplotype)
  : fw_haplotype_(haplotype), realign_to_hap_(realign_to_haplotype), cpu_aligner_(haplotype, realign_to_haplotype){
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

void HapAlignerMI355X::process_reads(const std::vector<Alignment>& alignments, int init_read_index, const BaseQuality* base_quality,
				     const std::vector<bool>& realign_read, double* aln_probs, int* seed_positions){
  assert(alignments.size() == realign_read.size());
  (void)base_quality;     // BaseQuality's tables are constants of the model; the device holds the same values
  FlatReads r(alignments, realign_read);
  FILL_BATCH(b, r)
  if (hipstr_hmm_process_reads(&b, aln_probs + (size_t)init_read_index*fw_haplotype_->num_combs(), seed_positions + init_read_index) != 0)
    printErrorAndDie(hipstr_last_error());
}
