#!/bin/bash
rm -f gpurun_out/ab_pwv.txt
export HIPSTR_SYNTH_IMPERFECT=1.0
tools/gpu_ab.sh pwv X=base HIPSTR_HMM_LIB=hipstr_amd/csrc/ablate/libhipstr_hmm_br.so HIPSTR_HMM_LIB=hipstr_amd/csrc/ablate/libhipstr_hmm_nh.so HIPSTR_HMM_LIB=hipstr_amd/csrc/ablate/libhipstr_hmm_nt.so HIPSTR_HMM_LIB=hipstr_amd/csrc/ablate/libhipstr_hmm_all3.so
