#!/bin/bash
# Builds timing-only variants of the library with parts of hs_str_kernel switched off (HS_ABLATE=k, results INVALID) into
# hipstr_amd/csrc/ablate/libhipstr_hmm_ab<k>.so; run them with HIPSTR_HMM_LIB=<path> HIPSTR_BENCH_NOCHECK=1 python bench.py ...
cd "$(dirname "$0")/../hipstr_amd/csrc"
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -mno-amdgpu-ieee -fPIC -shared -pthread -Wno-unused-result -Wno-unused-value"
SRC="api.hip hmm_kernels.hip post_kernels.hip prep.cpp trace.hip em.hip nw.hip batch_io.cpp stream.hip gather.cpp"
for k in "$@"; do /opt/rocm/bin/hipcc $FL -DHS_ABLATE=$k -o ablate/libhipstr_hmm_ab$k.so $SRC & done
wait
ls -la ablate/
