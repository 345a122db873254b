# scripts/exp_layer_probes.py on the product build and on each measurement build named on the command line
for v in "" "$@"; do
  if [ -z "$v" ]; then unset DIFFORMER_HIP_LIB; else export DIFFORMER_HIP_LIB=$PWD/scripts/bin/libdifformer_hip_$v.so; fi
  python scripts/exp_layer_probes.py 2>&1 | grep "layer +"
done
