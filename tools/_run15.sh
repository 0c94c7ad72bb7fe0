mkdir -p gpurun_out/r3p
python -m pytest tests -m gpu -x -q > gpurun_out/r3p/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r3p/pytest_gpu.txt
for w in 2 3; do HIPSTR_TIMING=1 HIPSTR_STREAM_WORKERS=$w python bench.py --workload p30 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3p/p30_w$w.json 2> gpurun_out/r3p/p30_w$w.err; python -c "
import json; d=json.loads(open('gpurun_out/r3p/p30_w$w.json').read()); e=d['end_to_end']; print('p30 workers $w', 'resident', round(d['value']/1e6,1), 'e2e', round(e['alignments_per_s']/1e6,1), 'frac', round(e['fraction_of_resident_rate'],3), e['one_locus_process_reads_latency']['40x32'])"; done
python - <<'PY'
import re
lines=open('gpurun_out/r3p/p30_w2.err').read().split('\n')
for i,l in enumerate(lines):
    m=re.search(r'hipstr_hmm_upload: total ([\d.]+) ms \(prepare ([\d.]+), blocks \+ staging ([\d.]+)\), (\d+) B of tables, (\d+) alignments', l)
    if m and int(m.group(5))>500000:
        ctx=[x for x in lines[max(0,i-8):i] if x.startswith('prepare_batch')]
        print(l[19:]); print('   ', ' | '.join(c[15:] for c in ctx[-4:]))
PY
