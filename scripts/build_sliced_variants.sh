#!/bin/bash
# Measurement builds (profiles/r04_experiments.md): one shared library per variant under scripts/bin/ (git-ignored, shipped to
# the GPU box), identical to the product's except for ONE object, compiled with the given flags -- from the fork
# scripts/variants/<OBJ>.hip when there is one (the probe / trace / alternative-layout branches live there, not in csrc/).
#   scripts/build_sliced_variants.sh name="flags" ...   then   scripts/run_sliced_variants.sh name ...   on the GPU box
#   (OBJ=<source stem> rebuilds another object with the flags, e.g. OBJ=simple_layer_xwide)
set -e
cd "$(dirname "$0")/.."
make -C difformer_amd/csrc -j8 >/dev/null
mkdir -p scripts/bin
for spec in "$@"; do
    name=${spec%%=*}; flags=${spec#*=}
    obj=/tmp/dif_obj_$name
    rm -rf $obj && cp -r difformer_amd/lib/obj $obj && rm -f $obj/${OBJ:-gcn_sliced}.o
    fork=scripts/variants/${OBJ:-gcn_sliced}.hip
    if [ -f $fork ]; then
        extra=""; [ "${OBJ:-gcn_sliced}" = gcn_sliced ] && extra="-fno-slp-vectorize"
        ${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra $flags \
            -I difformer_amd/csrc -c $fork -o $obj/${OBJ:-gcn_sliced}.o
    fi
    make -C difformer_amd/csrc OBJDIR=$obj OUT=../../scripts/bin/libdifformer_hip_$name.so EXTRA="$flags" >/dev/null
    echo "built scripts/bin/libdifformer_hip_$name.so ($flags)"
done
