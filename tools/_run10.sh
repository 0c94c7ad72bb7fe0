mkdir -p gpurun_out/r3k
python -m pytest tests -m gpu -x -q > gpurun_out/r3k/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r3k/pytest_gpu.txt
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pipeline > gpurun_out/r3k/ns.json 2> gpurun_out/r3k/ns.err
HIPSTR_SYNTH_IMPERFECT=1.0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline > gpurun_out/r3k/ns_imp.json 2> gpurun_out/r3k/ns_imp.err
python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline > gpurun_out/r3k/c5.json 2> gpurun_out/r3k/c5.err
python bench.py --workload p30 --steps 5 --warmup 1 --no-cpu-baseline --no-pipeline > gpurun_out/r3k/p30.json 2> gpurun_out/r3k/p30.err
for f in ns ns_imp c5 p30; do python -c "import json,sys; d=json.loads(open('gpurun_out/r3k/$f.json').read()); print('$f', round(d['value']/1e6,2), 'M/s', {k:round(v,2) for k,v in d['roofline']['phase_ms'].items()})"; done
