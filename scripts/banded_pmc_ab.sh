export TMPDIR=/tmp
out=gpurun_out/r2r; mkdir -p $out
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU"
SQ2="SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_IFETCH SQ_WAIT_ANY"
for impl in p d; do
  if [ $impl = d ]; then export BGSA_BANDED_IMPL=d; else unset BGSA_BANDED_IMPL; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $out/pmc_${impl}_SQ -- python3 bench.py --config 3 --banded-mix survivors --banded-variants '' --steps 1 --warmup 1 --no-cpu-baseline --no-total > $out/pmc_${impl}_SQ.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $out/pmc_${impl}_SQ2 -- python3 bench.py --config 3 --banded-mix survivors --banded-variants '' --steps 1 --warmup 1 --no-cpu-baseline --no-total > $out/pmc_${impl}_SQ2.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections
for impl in "pd":
    for grp in ("SQ","SQ2"):
        f = sorted(glob.glob(f"gpurun_out/r2r/pmc_{impl}_{grp}/**/*counter_collection.csv", recursive=True))[-1]
        acc = collections.defaultdict(float); n = collections.Counter()
        for row in csv.DictReader(open(f)):
            if "banded_asm_kernel" in row["Kernel_Name"]:
                acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
        disp = max(n.values()) if n else 1
        print(impl, grp, {k: f"{v / max(1, n[k]) * 1:.4g}" for k, v in acc.items()}, "dispatches", disp)
PY
