#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
python bench.py > $O/r03_bench_ns.json 2> $O/bench_ns.err
python bench.py --workload c4 --no-cpu-baseline > $O/r03_bench_c4.json 2> $O/bench_c4.err; tail -3 $O/bench_c4.err
tools/fuzz_big.sh 12 60 $O/r03_fuzz.txt 2>&1 | tail -3
