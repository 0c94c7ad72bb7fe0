#!/bin/bash
# Incremental developer build of libhipstr_hmm.so into $OUT (default /tmp/hs_dev): one object per source, relinked when any changed.
# The product build stays hipstr_amd/build.py (one hipcc call); this only shortens the edit-compile loop.  Use with HIPSTR_HMM_LIB=$OUT/libhipstr_hmm.so.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/hipstr_amd/csrc
OUT=${OUT:-/tmp/hs_dev}
mkdir -p $OUT/obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -mno-amdgpu-ieee -fPIC -pthread -fvisibility=hidden -Wno-unused-result -Wno-unused-value $EXTRA"
SRCS="api.hip hmm_kernels.hip expand_kernels.hip post_kernels.hip prep.cpp trace.hip em.hip nw.hip batch_io.cpp stream.hip gather.cpp"
pids=""
for s in $SRCS; do
  o=$OUT/obj/${s%.*}.o
  if [ ! -f $o ] || [ $CSRC/$s -nt $o ] || [ -n "$(find $CSRC/*.h $ROOT/include/hipstr_hmm.h -newer $o 2>/dev/null | head -1)" ]; then
    ( /opt/rocm/bin/hipcc $FLAGS -c -o $o.tmp $CSRC/$s && mv $o.tmp $o && echo "built $s" ) &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -Wl,--version-script=$CSRC/exports.map -o $OUT/libhipstr_hmm.so $(for s in $SRCS; do echo $OUT/obj/${s%.*}.o; done)
echo "linked $OUT/libhipstr_hmm.so"
