A=$PWD/hipstr_amd/csrc/ablate
rm -f gpurun_out/ab_r3i.txt
export HIPSTR_BENCH_NOCHECK=1
tools/gpu_ab.sh r3i HIPSTR_STR_GROUP_P=1 HIPSTR_HMM_LIB=$A/libhipstr_hmm_noconst.so HIPSTR_HMM_LIB=$A/libhipstr_hmm_noeval.so HIPSTR_HMM_LIB=$A/libhipstr_hmm_notable.so HIPSTR_HMM_LIB=$A/libhipstr_hmm_nochain.so HIPSTR_HMM_LIB=$A/libhipstr_hmm_noconst.so
