mkdir -p gpurun_out/r3o
HIPSTR_TIMING=1 python bench.py --workload p30 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3o/p30.json 2> gpurun_out/r3o/p30.err
grep "^stream:" gpurun_out/r3o/p30.err | tail -40
python -c "
import json; d=json.loads(open('gpurun_out/r3o/p30.json').read()); e=d['end_to_end']; print('p30 2 workers', 'resident', round(d['value']/1e6,1), 'e2e', round(e['alignments_per_s']/1e6,1), 'frac', round(e['fraction_of_resident_rate'],3))"
for w in 1 3; do HIPSTR_STREAM_WORKERS=$w python bench.py --workload p30 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3o/p30_w$w.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r3o/p30_w$w.json').read()); e=d['end_to_end']; print('p30 workers $w', 'resident', round(d['value']/1e6,1), 'e2e', round(e['alignments_per_s']/1e6,1), 'frac', round(e['fraction_of_resident_rate'],3))"; done
python -m pytest tests/test_stream_gpu.py -m gpu -x -q 2>&1 | tail -2
