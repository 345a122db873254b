#!/bin/bash
# GPU side of the sliced-product variant experiments (profiles/r04_experiments.md): same box, same graph, one process per
# build; the product's own build first and last.   scripts/run_sliced_variants.sh <out-tag> name ...
mkdir -p gpurun_out
out=gpurun_out/r04_sliced_$1.txt; shift
: > $out
for v in "" "$@" ""; do
    if [ -z "$v" ]; then unset DIFFORMER_HIP_LIB; else export DIFFORMER_HIP_LIB=$PWD/scripts/bin/libdifformer_hip_$v.so; fi
    timeout 300 python scripts/exp_sliced_variant.py 2>&1 | grep -v amdgpu.ids >> $out
done
unset DIFFORMER_HIP_LIB
cat $out
