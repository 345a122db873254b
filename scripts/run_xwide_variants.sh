# scripts/exp_xwide.py + scripts/exp_wide_linear.py on the product build and on each measurement build named on the command line
for v in "" "$@"; do
  if [ -z "$v" ]; then unset DIFFORMER_HIP_LIB; else export DIFFORMER_HIP_LIB=$PWD/scripts/bin/libdifformer_hip_$v.so; fi
  echo "== ${v:-product build}"
  python scripts/exp_xwide.py 2>&1 | grep "x 256\|x 300\|x 400"
  python scripts/exp_wide_linear.py 2>&1 | grep "50000 x 512"
done
