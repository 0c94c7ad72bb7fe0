// post_layout.h — device-side description of a posterior batch (post_kernels.hip, api.hip).
#pragma once
#include <stdint.h>

// One (locus, sample) pair: the unit a workgroup processes.
struct hs_post_unit_t {
  int64_t post_off;        // offset of this sample's [A x A] block in log_post
  int64_t prior_off;       // offset of this sample's [A x A] block in log_prior (the EM shares one block per locus)
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
  const double*  log_prior;      // optional prior array (NULL = hom/het defaults of the unit)
  const int32_t* unit_active;    // optional per-unit flag: 0 = skip (loci whose EM has converged); NULL = all
  double*        log_post;
  double*        sample_total;
  int32_t*       map_gt;
  double         log_thresh, log_half;
  int32_t        raw;            // HIPSTR_DEBUG_HOST_LIBM: leave the accumulated log P(reads, diplotype) unnormalised — the host takes the log-sum-exp with its libm
  // optional indirection (the device-resident EM loop, em.hip): workgroup b takes unit unit_list[b] if b < *n_list and leaves otherwise —
  // the launch is sized by a bound the host knows, the live units by a count only the device knows.  NULL = workgroup b takes unit b.
  const int32_t* unit_list;
  const int32_t* n_list;
  int32_t        sym_prior;      // the prior array is symmetric in the two alleles (the EM's: log f(a1) + log f(a2)): see the symmetric accumulation in post_kernels.hip
};

// One (locus, sample) pair of the genotype extraction (Genotyper::extract_genotypes_and_likelihoods, genotyper.cpp:129-251).
struct hs_gt_unit_t {
  int64_t post_off;        // this sample's [A x A] posteriors
  int64_t tot_off;         // scratch: this sample's [V x V] total_log_phased_posteriors
  int64_t gl_off, pgl_off; // this sample's pieces of the GL/PL and PHASEDGL outputs
  int32_t n_alleles, n_variants;
  int32_t samp_index;
  int32_t haploid;
  int32_t map_off;         // into h2a / gmem (per locus: A entries)
  int32_t goff_off;        // into goff (per locus: V+1 entries, relative to map_off)
  double  hom_corr, het_corr;     // priors to take out again (genotyper.cpp:197-198)
  double  gl_ncfg, pgl_ncfg;      // corrections for the number of averaged haplotype configurations (:201-209)
};

struct hs_gt_dev_t {
  const hs_gt_unit_t* units;
  const double*  log_post;
  const double*  sample_total;
  const int32_t* map_gt;          // MAP haplotype pair per sample (hs_posterior_kernel)
  const int32_t* h2a;             // hap_to_allele
  const int32_t* gmem;            // haplotypes of a locus sorted by (variant, haplotype index)
  const int32_t* goff;            // start of every variant's haplotypes in gmem
  double*  tot;                   // scratch
  int32_t* best_gt;               // [2*n_samp]
  double*  log_phased, *log_unphased, *hap_log_phased, *hap_log_unphased, *gl_diff;   // [n_samp]
  double*  gls;  int32_t* pls;  double* pgls;
  int32_t  calc_any, calc_gls, calc_pls, calc_pgls;
  double   log_thresh;
  int32_t  tot_given;            // HIPSTR_DEBUG_HOST_LIBM: `tot` (and log_unphased) come from the host
};
