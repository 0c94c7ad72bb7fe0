#!/bin/bash
# A variant of the library that differs in hmm_kernels.hip's -D flags only: that one object is compiled, the other objects are the
# product build's (build/hmm_obj, python -m hipstr_amd.build first).  -> hipstr_amd/csrc/ablate/libhipstr_hmm_<name>.so (git-ignored;
# run it with HIPSTR_HMM_LIB=<path>).   usage: tools/build_kernel_variant.sh <name> [-DMACRO=value ...]
R="$(cd "$(dirname "$0")/.." && pwd)"; NAME=$1; shift
mkdir -p $R/hipstr_amd/csrc/ablate $R/build/var_obj
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -mno-amdgpu-ieee -fPIC -pthread -fvisibility=hidden -Wno-unused-result -Wno-unused-value"
/opt/rocm/bin/hipcc $FL "$@" -c -o $R/build/var_obj/hmm_kernels_$NAME.o $R/hipstr_amd/csrc/hmm_kernels.hip || exit 1
OBJS=$(ls $R/build/hmm_obj/*.o | grep -v hmm_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -Wl,--version-script=$R/hipstr_amd/csrc/exports.map -o $R/hipstr_amd/csrc/ablate/libhipstr_hmm_$NAME.so $OBJS $R/build/var_obj/hmm_kernels_$NAME.o
