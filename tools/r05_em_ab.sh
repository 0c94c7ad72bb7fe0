#!/bin/bash
# A/B of the allele-frequency scans of the stutter EM (em_gt_priors): the library as built against variants under hipstr_amd/csrc/ablate/
# (em_old: the scans of commit 9d9995a, exponentials inside the chain; em_mid: prefix maxima first, exponentials by all threads; the product:
# + the exponentials that are formed listed and evaluated densely; *_t: -DHS_EM_TIME, a workgroup's cycles per stage).  3000 loci x 100 samples x 6 reads.
A=hipstr_amd/csrc/ablate
for rep in 1 2; do
for lib in "" $A/libhipstr_hmm_em_mid.so $A/libhipstr_hmm_em_old.so; do
  [ -n "$lib" ] && [ ! -f "$lib" ] && continue
  for ser in 0 1; do
    echo "== lib=${lib:-product} HIPSTR_EM_SERIAL=$ser"
    HIPSTR_HMM_LIB=$lib HIPSTR_EM_SERIAL=$ser python tools/r05_em_small.py 2>&1 | grep "^em"
  done
done
done
for lib in $A/libhipstr_hmm_em_new_t.so; do
  [ -f "$lib" ] || continue
  echo "== $lib"
  HIPSTR_HMM_LIB=$lib python tools/r05_em_small.py 2>&1 | grep "gt_priors locus" | sort | uniq -c | sort -rn | head -6
done
