// post_layout.h — device-side description of a posterior batch (post_kernels.hip, api.hip).
#pragma once
#include <stdint.h>

// One (locus, sample) pair: the unit a workgroup processes.
struct hs_post_unit_t {
  int64_t post_off;        // offset of this sample's [A x A] block in log_post
  int64_t ll_off;          // offset in log_aln_probs of the row of this sample's FIRST read
  int32_t n_alleles;
  int32_t read_begin;      // global read index of the sample's first read (reads of a sample are contiguous)
  int32_t n_reads;
  int32_t samp_index;      // global sample slot
  double  log_hom_prior;   // Genotyper::log_homozygous_prior   (genotyper.cpp:20-25)
  double  log_het_prior;   // Genotyper::log_heterozygous_prior (genotyper.cpp:27-32)
};

struct hs_post_dev_t {
  const hs_post_unit_t* units;
  const double*  log_aln_probs;
  const double*  log_p1;
  const double*  log_p2;
  const int32_t* read_weight;
  const double*  log_prior;      // optional [n_post] prior array (NULL = hom/het defaults of the unit)
  double*        log_post;
  double*        sample_total;
  int32_t*       map_gt;
  double         log_thresh, log_half;
};
