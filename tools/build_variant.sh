#!/bin/bash
# Builds a variant of the library with extra -D flags into hipstr_amd/csrc/ablate/libhipstr_hmm_<name>.so (git-ignored); run it with
# HIPSTR_HMM_LIB=<path> python bench.py ...      usage: tools/build_variant.sh <name> [-DMACRO=value ...]
cd "$(dirname "$0")/../hipstr_amd/csrc"
NAME=$1; shift
mkdir -p ablate
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -mno-amdgpu-ieee -fPIC -shared -pthread -fvisibility=hidden -Wno-unused-result -Wno-unused-value"
SRC="api.hip hmm_kernels.hip expand_kernels.hip post_kernels.hip prep.cpp trace.hip em.hip nw.hip batch_io.cpp stream.hip gather.cpp"
/opt/rocm/bin/hipcc $FL -Wl,--version-script=exports.map "$@" -o ablate/libhipstr_hmm_$NAME.so $SRC
