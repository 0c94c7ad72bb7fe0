A=$PWD/hipstr_amd/csrc/ablate
mkdir -p gpurun_out/r3l
python -m pytest tests/test_stream_gpu.py tests/test_genotypes_gpu.py tests/test_config4_gpu.py -m gpu -x -q -rP 2>&1 | tail -40 > gpurun_out/r3l/pytest.txt; tail -15 gpurun_out/r3l/pytest.txt
rm -f gpurun_out/ab_r3l.txt
tools/gpu_ab.sh r3l HIPSTR_STR_GROUP_P=1 HIPSTR_HMM_LIB=$A/libhipstr_hmm_occ5.so HIPSTR_STR_GROUP_P=1 HIPSTR_HMM_LIB=$A/libhipstr_hmm_occ5.so
python bench.py --workload c4 --loci 20 --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > gpurun_out/r3l/c4.json 2> gpurun_out/r3l/c4.err; python -c "import json; d=json.loads(open('gpurun_out/r3l/c4.json').read()); print('c4', round(d['value']/1e6,2), d.get('c4_step'))"
