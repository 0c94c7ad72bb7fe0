#!/bin/bash
# Diagnostic builds of hs_em_gt_priors' rows phase (-DHS_EM_TIME; results invalid): abl0 as is, abl1 no exponentials, abl2 no tile loads from
# memory, abl3 no per-row walks (maximum, sum) — a workgroup's cycles per stage (tools/r05_em_small.py: 3000 loci x 100 samples x 6 reads)
for v in 0 1 2 3; do
  lib=hipstr_amd/csrc/ablate/libhipstr_hmm_em_abl$v.so; [ -f $lib ] || continue
  echo "== abl$v"
  HIPSTR_HMM_LIB=$lib timeout 300 python tools/r05_em_small.py 2>&1 | grep "gt_priors locus\|^em" | head -5
done
