python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3e_smoke.txt 2>&1; tail -1 gpurun_out/r3e_smoke.txt
python tools/fuzz_align.py 250 779 > gpurun_out/r3e_fuzz.txt 2>&1; tail -2 gpurun_out/r3e_fuzz.txt
HIPSTR_DEBUG_REDO=3 python tools/fuzz_align.py 40 778 > gpurun_out/r3e_fuzz_redo.txt 2>&1; tail -1 gpurun_out/r3e_fuzz_redo.txt
rm -f gpurun_out/ab_r3e.txt
tools/gpu_ab.sh r3e HIPSTR_STR_GROUP_P=0 HIPSTR_STR_GROUP_P=1 HIPSTR_STR_GROUP_P=1 HIPSTR_STR_GROUP_P=1 HIPSTR_STR_GROUP_P=0 HIPSTR_STR_GROUP_P=1
grep -i fault gpurun_out/ab_r3e_err.txt
