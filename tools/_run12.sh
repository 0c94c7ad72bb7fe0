mkdir -p gpurun_out/r3m
python -m pytest tests/test_stream_gpu.py tests/test_hmm_gpu.py tests/test_host_api.py -m gpu -x -q 2>&1 | tail -3
HIPSTR_TIMING=1 python bench.py --workload p30 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3m/p30.json 2> gpurun_out/r3m/p30.err
python - <<'PY'
import json,re,collections
d=json.loads(open('gpurun_out/r3m/p30.json').read())
print('resident', round(d['value']/1e6,1), 'e2e', round(d['end_to_end']['alignments_per_s']/1e6,1), 'frac', round(d['end_to_end']['fraction_of_resident_rate'],3))
print({k:d['end_to_end'][k] for k in ('seconds','passes','batches','worker_host_seconds','collector_wait_seconds')})
print(d['end_to_end']['one_locus_process_reads_latency'])
big=[]
for line in open('gpurun_out/r3m/p30.err'):
    m=re.search(r'hipstr_hmm_upload: total ([\d.]+) ms \(prepare ([\d.]+), blocks \+ staging ([\d.]+)\), (\d+) B of tables, (\d+) alignments', line)
    if m and int(m.group(5))>100000: big.append(tuple(float(x) for x in m.groups()))
for b in big[-8:]: print('upload total %.2f ms prepare %.2f staging %.2f tables %.1f MB alignments %d'%(b[0],b[1],b[2],b[3]/1e6,b[4]))
acc=collections.defaultdict(list)
for line in open('gpurun_out/r3m/p30.err'):
    m=re.match(r'prepare_batch: (\w+)\s+([\d.]+) ms', line)
    if m: acc[m.group(1)].append(float(m.group(2)))
for k,v in acc.items():
    v=sorted(v)[-8:]; print('prepare_batch', k, 'largest:', [round(x,2) for x in v])
PY
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3m/ns.json 2> gpurun_out/r3m/ns.err; python -c "
import json; d=json.loads(open('gpurun_out/r3m/ns.json').read()); print('ns resident', round(d['value']/1e6,1), 'e2e', round(d['end_to_end']['alignments_per_s']/1e6,1), d['end_to_end']['one_locus_process_reads_latency'])"
