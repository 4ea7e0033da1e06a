#!/usr/bin/env python3
"""Copy the judged summaries of one scripts/gpu_round_report.sh run from gpurun_out/<tag>/ into
profiles/ under <prefix>_*: bench lines, rocprofv3 kernel stats, per-launch PMC values of the
dominant kernel, the GPU test log and the microbenchmark outputs.

    python3 scripts/collect_profiles.py gpurun_out/r01_v7 r01_v7
"""
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

src, prefix = Path(sys.argv[1]), sys.argv[2]
dst = Path(__file__).resolve().parent.parent / "profiles"
dst.mkdir(exist_ok=True)

for c in (2, 3, 4, 5):
    b = src / f"bench_cfg{c}.json"
    if b.exists() and b.stat().st_size:
        line = b.read_text().strip().splitlines()[-1]
        json.loads(line)   # must be the one JSON line of the contract
        (dst / f"{prefix}_cfg{c}_bench.json").write_text(line + "\n")
    stats = glob.glob(str(src / f"prof_cfg{c}" / "**" / "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], dst / f"{prefix}_cfg{c}_kernel_stats.csv")

# PMC: value per launch of the kernel with the largest total (the dominant kernel of the bench line)
rows = []
for d in sorted(glob.glob(str(src / "pmc_*"))):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    per = {}
    for r in csv.DictReader(open(files[0])):
        if "bgsa::" not in r["Kernel_Name"] or "pack" in r["Kernel_Name"] or "preprocess" in r["Kernel_Name"] \
                or "map_queries" in r["Kernel_Name"]:
            continue
        key = (r["Counter_Name"], r["Kernel_Name"].split("(")[0])
        tot, n = per.get(key, (0.0, set()))
        n.add(r["Dispatch_Id"])
        per[key] = (tot + float(r["Counter_Value"]), n)
    for (ctr, kern), (tot, disp) in sorted(per.items()):
        note = ""
        if ctr == "FETCH_SIZE":
            note = "KB; gfx950 counts 64 B per 128-B request on coalesced reads (MI355X_MICROARCH.md): x2"
        if ctr == "WRITE_SIZE":
            note = "KB"
        rows.append((Path(d).name, ctr, f"{tot / len(disp):.0f}", kern, note))
if rows:
    with open(dst / f"{prefix}_cfg2_pmc.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["pass", "counter", "value_per_launch", "kernel", "note"])
        w.writerows(rows)

for name, out in (("pytest_gpu.log", f"{prefix}_pytest_gpu.log"), ("ubench_valu_rate.txt", f"{prefix}_ubench_valu_rate.txt"),
                  ("ubench_body_rate.txt", f"{prefix}_ubench_body_rate.txt"), ("ubench_operand_cost.txt", f"{prefix}_ubench_operand_cost.txt"),
                  ("bitpal_sets.jsonl", f"{prefix}_bitpal_sets.jsonl"), ("summary.txt", f"{prefix}_summary.txt")):
    if (src / name).exists() and (src / name).stat().st_size:
        shutil.copy(src / name, dst / out)
print("collected into", dst)
