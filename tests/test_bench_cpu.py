"""bench.py's provenance logic, on the CPU: a PMC pass is only used for the build it was collected from."""
import csv
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bench_under_test"] = mod
    spec.loader.exec_module(mod)
    return mod


def test_kernel_source_id_names_the_kernel_sources():
    bench = _bench()
    ids = {algo: bench.kernel_source_id(algo) for algo in bench.KERNEL_SOURCES}
    assert all(len(i) == 16 and int(i, 16) >= 0 for i in ids.values())
    assert len(set(ids.values())) == len(ids)                      # Myers, banded and BitPAl are different kernels
    assert ids == {algo: bench.kernel_source_id(algo) for algo in bench.KERNEL_SOURCES}   # and the id is a function of the files


def test_pmc_pass_of_another_build_is_refused(tmp_path, monkeypatch):
    bench = _bench()
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", tmp_path)

    def write(name, source_id):
        with open(tmp_path / "profiles" / name, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["pass", "counter", "value_per_launch", "kernel", "note"])
            if source_id:
                w.writerow(["_meta", "kernel_source_id", source_id, "k", ""])
            w.writerow(["pmc_cfg2_SQ", "SQ_INSTS_VALU", "1177813002817", "k", ""])
            w.writerow(["pmc_cfg2_FETCH_SIZE", "FETCH_SIZE", "15297673", "k", "KB"])

    write("r02_cfg2_pmc.csv", None)                                # round 2's files carry no stamp
    vals, why = bench.pmc_values(2, "", "aaaaaaaaaaaaaaaa")
    assert vals is None and "refused" in why and "unstamped" in why
    write("r03_cfg2_pmc.csv", "bbbbbbbbbbbbbbbb")                  # the newest file wins; stamped by another build
    vals, why = bench.pmc_values(2, "", "aaaaaaaaaaaaaaaa")
    assert vals is None and "bbbbbbbbbbbbbbbb" in why and "re-collect" in why
    vals, src = bench.pmc_values(2, "", "bbbbbbbbbbbbbbbb")        # the same build: used, without the meta row
    assert src == "r03_cfg2_pmc.csv" and vals == {"SQ_INSTS_VALU": 1177813002817.0, "FETCH_SIZE": 15297673.0}
    assert bench.pmc_values(4, "", "bbbbbbbbbbbbbbbb") == (None, None)   # no pass at all
