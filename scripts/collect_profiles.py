#!/usr/bin/env python3
"""Copy the judged summaries of one scripts/gpu_round_report.sh run from gpurun_out/<tag>/ into
profiles/ under <prefix>_*: bench lines, rocprofv3 kernel stats, per-launch PMC values of the
dominant kernel of every config (and, for config 3, of every subject mix), the GPU test log and the
microbenchmark outputs.

    python3 scripts/collect_profiles.py gpurun_out/r02 r02
"""
import csv
import glob
import json
import re
import shutil
import sys
from pathlib import Path

src, prefix = Path(sys.argv[1]), sys.argv[2]
dst = Path(__file__).resolve().parent.parent / "profiles"
dst.mkdir(exist_ok=True)
HELPERS = ("pack", "preprocess", "map_queries", "scale_scores", "corrupt_stream")

extra = src / "bench_cfg3_k31.json"
if extra.exists() and extra.stat().st_size:
    line = [x for x in extra.read_text().strip().splitlines() if x.startswith("{")][-1]
    json.loads(line)
    (dst / f"{prefix}_cfg3_k31_bench.json").write_text(line + "\n")
for c in (2, 3, 4, 5):
    b = src / f"bench_cfg{c}.json"
    if b.exists() and b.stat().st_size:
        line = [x for x in b.read_text().strip().splitlines() if x.startswith("{")][-1]
        json.loads(line)   # must be the one JSON line of the contract
        (dst / f"{prefix}_cfg{c}_bench.json").write_text(line + "\n")
    stats = sorted(glob.glob(str(src / f"prof_cfg{c}" / "**" / "*kernel_stats.csv"), recursive=True), key=lambda f: Path(f).stat().st_mtime)
    if stats:
        shutil.copy(stats[-1], dst / f"{prefix}_cfg{c}_kernel_stats.csv")

# PMC: value per launch of the scoring kernel, one file per config (and banded mix)
groups = {}
for d in sorted(glob.glob(str(src / "pmc_cfg*"))):
    m = re.match(r"pmc_(cfg\d(?:_(?:planted|random|dense1pct|survivors|k\d+))?)_(FETCH_SIZE|WRITE_SIZE|SQ)$", Path(d).name)
    if m and Path(d).is_dir():
        groups.setdefault(m.group(1), []).append(d)
for key, dirs in groups.items():
    rows = []
    for d in dirs:
        files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=lambda f: Path(f).stat().st_mtime)
        if not files:
            continue
        per = {}
        for r in csv.DictReader(open(files[-1])):      # the latest run into this directory
            name = r["Kernel_Name"]
            if "bgsa::" not in name or any(h in name for h in HELPERS):
                continue
            k = (r["Counter_Name"], name.split("(")[0])
            tot, n = per.get(k, (0.0, set()))
            n.add(r["Dispatch_Id"])
            per[k] = (tot + float(r["Counter_Value"]), n)
        for (ctr, kern), (tot, disp) in sorted(per.items()):
            note = ""
            if ctr == "FETCH_SIZE":
                note = "KB; gfx950 counts 64 B per 128-B request on coalesced reads (MI355X_MICROARCH.md): x2"
            if ctr == "WRITE_SIZE":
                note = "KB"
            rows.append((Path(d).name, ctr, f"{tot / len(disp):.0f}", kern, note))
    if rows:
        # provenance: the kernel source id the PROFILED bench.py printed in its line (config.kernel_source_id); bench.py's
        # pmc_values() refuses the file when the library it runs was built from other kernel sources
        ids = set()
        for d in dirs:
            log = Path(str(d) + ".log")
            if log.exists():
                for ln in log.read_text(errors="replace").splitlines():
                    if ln.startswith("{") and "kernel_source_id" in ln:
                        try:
                            ids.add(json.loads(ln)["config"]["kernel_source_id"])
                        except Exception:
                            pass
        if len(ids) == 1:
            rows.insert(0, ("_meta", "kernel_source_id", ids.pop(), rows[0][3], "sha256[:16] of the kernel's sources (bench.py: kernel_source_id)"))
        elif ids:
            print(f"{key}: the passes disagree on the kernel source id {sorted(ids)} — not stamped", file=sys.stderr)
        with open(dst / f"{prefix}_{key}_pmc.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["pass", "counter", "value_per_launch", "kernel", "note"])
            w.writerows(rows)

for name, out in (("pytest_gpu.log", f"{prefix}_pytest_gpu.log"), ("ubench_valu_rate.txt", f"{prefix}_ubench_valu_rate.txt"),
                  ("ubench_body_rate.txt", f"{prefix}_ubench_body_rate.txt"), ("ubench_operand_cost.txt", f"{prefix}_ubench_operand_cost.txt"),
                  ("bitpal_sets.jsonl", f"{prefix}_bitpal_sets.jsonl"), ("summary.txt", f"{prefix}_summary.txt")):
    if (src / name).exists() and (src / name).stat().st_size:
        shutil.copy(src / name, dst / out)
print("collected into", dst)
