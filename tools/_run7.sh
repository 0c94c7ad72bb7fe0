R=$PWD
G=$R/hipstr_amd/csrc/ablate/libhipstr_hmm_gt.so
OUT=$R/gpurun_out/r3g; mkdir -p $OUT
HIPSTR_HMM_LIB=$G python bench.py --loci 100 --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline > $OUT/gt_p1.txt 2>&1
grep "^grp" $OUT/gt_p1.txt | head -12
ARGS="--loci 200 --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline"
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQ3="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_ANY"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $SQ1 -d $OUT/p1a -o v -- python $R/bench.py $ARGS > $OUT/p1a.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ3 -d $OUT/p1b -o v -- python $R/bench.py $ARGS > $OUT/p1b.log 2>&1
cd $R
for t in p1a p1b; do echo "=== $t"; python tools/pmc_quick.py $(find $OUT/$t -name '*results.db' | head -1) hs_; done > $OUT/summary.txt 2>&1
rm -rf $OUT/p1a $OUT/p1b
cat $OUT/summary.txt
tools/gpu_ab.sh r3g HIPSTR_STR_GROUP_P=1
