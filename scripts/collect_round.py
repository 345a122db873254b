"""Copy what scripts/profile_round.sh <tag> left under gpurun_out/ into profiles/ under a round prefix:
    python scripts/collect_round.py r04e r04_e
-> profiles/<prefix>_bench_c4.json, _bench_<workload>_kernel_stats.csv (headline), _<workload>_kernel_stats.csv (other configs),
   _bench_small_configs.json ({workload_mode: bench line}), _bench_zipf_c4.json, _pokec_epoch.txt, _train_step.txt,
   _sigmoid_bwd.txt, _row_shard_per_rank.txt, and <round>_pmc_traffic_c4.json (through scripts/make_traffic_json.py)."""
import glob
import json
import os
import shutil
import sys

tag, prefix = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, cfg, dst = (os.path.join(root, "gpurun_out", t) for t in (tag, tag + "_cfg", "")) 
dst = os.path.join(root, "profiles")


def first_json_line(path):
    for line in open(path):
        if line.startswith("{"):
            return json.loads(line)
    raise ValueError(path)


shutil.copy(os.path.join(src, "bench_c4.json"), os.path.join(dst, f"{prefix}_bench_c4.json"))
shutil.copy(os.path.join(src, "bench_c4_kernel_stats.csv"), os.path.join(dst, f"{prefix}_bench_ogbn-proteins-s_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_zipf.json"), os.path.join(dst, f"{prefix}_bench_zipf_c4.json"))
if os.path.exists(os.path.join(src, "zipf_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "zipf_kernel_stats.csv"), os.path.join(dst, f"{prefix}_ogbn-proteins-zipf-s_kernel_stats.csv"))
for name, out in (("pokec_epoch.log", "pokec_epoch.txt"), ("train_step.log", "train_step.txt"), ("sigmoid_bwd.log", "sigmoid_bwd.txt"),
                  ("sliced_shard.log", "row_shard_per_rank.txt"), ("st_epoch_tiny.log", "st_epoch_tiny.txt"),
                  ("st_epoch_layers.log", "st_epoch_layers.txt"), ("c5_bf16.log", "c5_bf16.txt"), ("bf16_scaling.log", "bf16_scaling.txt"),
                  ("regional_order.log", "regional_order.txt")):
    p = os.path.join(src, name)
    if os.path.exists(p):
        lines = [l for l in open(p) if "amdgpu.ids" not in l and "UserWarning" not in l and "Consider using tensor.detach" not in l and "return float(cost_tr)" not in l]
        open(os.path.join(dst, f"{prefix}_{out}"), "w").writelines(lines)
small = {}
for p in sorted(glob.glob(os.path.join(cfg, "bench_*.json"))):
    key = os.path.basename(p)[len("bench_"):-len(".json")]
    try:
        small[key] = first_json_line(p)
    except Exception as e:          # a config that did not run: say so instead of dropping it
        small[key] = {"error": str(e)}
json.dump(small, open(os.path.join(dst, f"{prefix}_bench_small_configs.json"), "w"), indent=1)
for p in sorted(glob.glob(os.path.join(cfg, "*_kernel_stats.csv"))):
    shutil.copy(p, os.path.join(dst, f"{prefix}_{os.path.basename(p)}"))
pmc = os.path.join(src, "pmc_summary.json")
if os.path.exists(pmc):       # per-kernel HBM bytes with the calibrated corrections (keeps the calibration block of the tracked file)
    import subprocess
    subprocess.run([sys.executable, os.path.join(root, "scripts", "make_traffic_json.py"), pmc,
                    os.path.join(dst, f"{prefix[:3]}_pmc_traffic_c4.json"), "ogbn-proteins-s"], check=True, stdout=subprocess.DEVNULL)
print("collected", len(small), "bench lines")
for k, v in small.items():
    if "ms_per_step" in v:
        r = v.get("roofline") or {}
        print(f"  {k:36s} {v['ms_per_step']:.4f} ms  {r.get('entry_point')} [{r.get('bound')}] frac {r.get('frac')}")
    else:
        print(f"  {k:36s} {v}")
