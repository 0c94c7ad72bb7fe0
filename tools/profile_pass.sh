#!/bin/bash
# profile_pass.sh — every rocprofv3 pass the round's profile summaries come from, on the GPU box (run through gpurun from the repo root):
#   tools/profile_pass.sh <tag> [bench.py arguments...]        e.g.  tools/profile_pass.sh r02 --loci 200
# Kernel trace and every counter group are SEPARATE runs (--pmc never together with --stats / trace domains other than the kernel
# trace: MI355X_MICROARCH.md, HBM/rocprofv3 section).  Raw databases stay under gpurun_out/prof_<tag>/; the summaries are written to
# gpurun_out/prof_<tag>/*.{txt,json}, to be copied into profiles/ and committed.
set -u
TAG=$1; shift
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-pipeline $*"
# the generator overrides of the caller's environment are part of what was profiled: they go in front of the recorded command (bench.py
# does not take a summary with such an override for the plain workload's)
SYNTH_ENV=$(env | grep '^HIPSTR_SYNTH' | sort | tr '\n' ' ')
cd /tmp && export TMPDIR=/tmp
python $R/bench.py $ARGS > $OUT/bench.log 2>&1
N_ALN=$(python -c "import json;print(json.loads([l for l in open('$OUT/bench.log') if l.startswith('{')][-1])['config']['alignments_per_step_per_gpu'])")
rocprofv3 --kernel-trace --stats -d $OUT/trace -o v -- python $R/bench.py $ARGS > $OUT/trace.log 2>&1
python $R/tools/rocprof_summary.py $(find $OUT/trace -name '*results.db' | head -1) $OUT/bench.log > $OUT/kernel_stats.txt
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32"
rocprofv3 --kernel-trace --pmc $SQ1 -d $OUT/sq1 -o v -- python $R/bench.py $ARGS > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ2 -d $OUT/sq2 -o v -- python $R/bench.py $ARGS > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o v -- python $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o v -- python $R/bench.py $ARGS > $OUT/write.log 2>&1
db(){ find $OUT/$1 -name '*results.db' | head -1; }
python $R/tools/sq_counters.py $(db sq1) $(db sq2) $N_ALN "${SYNTH_ENV}bench.py $ARGS" $R/profiles > $OUT/sq_counters.json
python $R/tools/pmc_traffic.py $(db fetch) $(db write) $N_ALN "${SYNTH_ENV}bench.py $ARGS" > $OUT/pmc_traffic.json
ls -la $OUT | head -30
