"""profiles/<tag>_pmc_traffic_c4.json from the per-kernel counter means of scripts/pmc_passes.sh (summary.json).
HBM bytes per launch = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB (calibration block: a known 2 GiB stream reads exactly half
in FETCH_SIZE on gfx950 for 4- and 16-byte-per-lane loads, WRITE_SIZE is exact; MI355X_MICROARCH.md, HBM section).
    python scripts/make_traffic_json.py gpurun_out/r02c/pmc_summary.json profiles/r02_pmc_traffic_c4.json ogbn-proteins-s
"""
import json, sys

src, dst, workload = sys.argv[1:4]
raw = json.load(open(src))
old = json.load(open(dst)) if len(sys.argv) < 5 else json.load(open(sys.argv[4]))
out = {"workload": workload, "collected_with": old["collected_with"], "calibration": old["calibration"], "kernels": {},
       "raw_counters": raw}
fc, wc = old["calibration"]["fetch_correction"], old["calibration"]["write_correction"]
for k, v in raw.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        out["kernels"][k] = {"FETCH_SIZE_KiB_raw": v["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": v["WRITE_SIZE"],
                             "launches_averaged": v.get("dispatches"),
                             "hbm_bytes_per_launch": (fc * v["FETCH_SIZE"] + wc * v["WRITE_SIZE"]) * 1024.0}
        for a, b, name in (("TCC_HIT_sum", "TCC_REQ_sum", "l2_hit_rate"),):
            if a in v and b in v and v[b]:
                out["kernels"][k][name] = v[a] / v[b]
json.dump(out, open(dst, "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k:32s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB / launch   L2 hit {v.get('l2_hit_rate', float('nan')):.3f}")
