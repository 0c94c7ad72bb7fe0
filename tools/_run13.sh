mkdir -p gpurun_out/r3n
run(){ tag=$1; shift; env "$@" python bench.py --workload ${WL:-ns} --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3n/$tag.json 2> gpurun_out/r3n/$tag.err; python -c "
import json; d=json.loads(open('gpurun_out/r3n/$tag.json').read()); e=d['end_to_end']; print('$tag', 'resident', round(d['value']/1e6,1), 'e2e', round(e['alignments_per_s']/1e6,1), 'frac', round(e['fraction_of_resident_rate'],3), 'worker_s', round(e['worker_host_seconds'],3), 'wait_s', round(e['collector_wait_seconds'],3), 'sec', round(e['seconds'],3))"; }
run ns_default A=1
run ns_nopool HIPSTR_HOST_POOL=0
run ns_nocache HIPSTR_STROPT_CACHE=0
run ns_neither HIPSTR_HOST_POOL=0 HIPSTR_STROPT_CACHE=0
WL=p30 run p30_default A=1
WL=p30 run p30_nopool HIPSTR_HOST_POOL=0
WL=p30 run p30_neither HIPSTR_HOST_POOL=0 HIPSTR_STROPT_CACHE=0
