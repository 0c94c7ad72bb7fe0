#!/bin/bash
# A/B of the allele-frequency scans of the stutter EM (em_gt_priors): the library as built against variants under hipstr_amd/csrc/ablate/
# (em_old: the scans with exponentials inside the chain; *_t: -DHS_EM_TIME, a workgroup's cycles per stage).  3000 loci x 100 samples x 6 reads.
A=hipstr_amd/csrc/ablate
for lib in "" $A/libhipstr_hmm_em_old.so; do
  for ser in 0 1; do
    echo "== lib=${lib:-product} HIPSTR_EM_SERIAL=$ser"
    HIPSTR_HMM_LIB=$lib HIPSTR_EM_SERIAL=$ser python tools/r05_em_small.py 2>&1 | grep "^em"
  done
done
for lib in $A/libhipstr_hmm_em_old_t.so $A/libhipstr_hmm_em_new_t.so; do
  echo "== $lib"
  HIPSTR_HMM_LIB=$lib python tools/r05_em_small.py 2>&1 | grep "gt_priors locus" | sort | uniq -c | sort -rn | head -6
done
