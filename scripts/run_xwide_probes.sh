for v in "" xp1 xp2 xp3; do
  if [ -z "$v" ]; then unset DIFFORMER_HIP_LIB; else export DIFFORMER_HIP_LIB=$PWD/scripts/bin/libdifformer_hip_$v.so; fi
  echo "== ${v:-product build}"; python scripts/exp_xwide.py 2>&1 | grep -v amdgpu | grep "x 300\|x 192"
done
