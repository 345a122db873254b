set -x
./scripts/profile_c4.sh r02c > gpurun_out/r02c_c4.log 2>&1
./scripts/profile_configs.sh r02c_cfg > gpurun_out/r02c_cfg.log 2>&1
python bench.py --workload ogbn-proteins-zipf-s --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r02c/bench_zipf.json 2> gpurun_out/r02c/bench_zipf.err
python scripts/exp_pokec_epoch.py > gpurun_out/r02c/pokec_epoch.log 2>&1
tail -5 gpurun_out/r02c_c4.log; tail -14 gpurun_out/r02c_cfg.log; cat gpurun_out/r02c/bench_zipf.json | cut -c1-300; tail -8 gpurun_out/r02c/pokec_epoch.log
