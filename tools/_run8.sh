python tools/fuzz_align.py 120 783 > gpurun_out/r3j_fuzz.txt 2>&1; tail -1 gpurun_out/r3j_fuzz.txt
HIPSTR_DEBUG_REDO=3 python tools/fuzz_align.py 30 784 > gpurun_out/r3j_fuzz_redo.txt 2>&1; tail -1 gpurun_out/r3j_fuzz_redo.txt
rm -f gpurun_out/ab_r3j.txt
tools/gpu_ab.sh r3j HIPSTR_STR_GROUP_P=1 HIPSTR_STR_GROUP_P=1 HIPSTR_STR_GROUP_P=0
