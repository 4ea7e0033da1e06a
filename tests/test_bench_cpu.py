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


def _run_watchdog_script(body: str, timeout=120):
    """A child process that arms bench.py's RunWatchdog and then runs `body` (the watchdog ends it with os._exit)."""
    import subprocess
    import textwrap
    script = textwrap.dedent("""
        import argparse, importlib.util, sys, time
        spec = importlib.util.spec_from_file_location("bench_under_test", r"{bench}")
        bench = importlib.util.module_from_spec(spec); sys.modules["bench_under_test"] = bench; spec.loader.exec_module(bench)
        args = argparse.Namespace(config=2, steps=3, warmup=1)
    """).format(bench=ROOT / "bench.py") + textwrap.dedent(body)
    return subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=timeout, cwd=str(ROOT))


def test_watchdog_prints_a_stub_line_and_exits_3_when_nothing_was_measured():
    """A hang before the timed region (RCCL init, the first barrier, the query broadcast): rank 0's line is a stub that
    names the stage, rccl_ok is false, exit status 3."""
    import json
    p = _run_watchdog_script("""
        wd = bench.RunWatchdog(0, 8, args, limit=0.3)
        wd.stage = "init_process_group(nccl)"
        time.sleep(30)
    """)
    assert p.returncode == 3, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["value"] is None and r["n_gpus"] == 8 and r["rccl_ok"] is False and r["unit"] == "GCUPS"
    assert r["watchdog"]["fired"] and r["watchdog"]["stage"] == "init_process_group(nccl)"


def test_watchdog_keeps_the_measured_line_and_a_leg_limit_marks_its_leg():
    import json
    p = _run_watchdog_script("""
        wd = bench.RunWatchdog(0, 2, args, limit=60)
        wd.result = {"metric": "m", "value": 123.0, "gather_ok": None, "rccl_ok": None}
        wd.stage = "gather leg"
        wd.leg(0.3, "gather leg did not finish", lambda line: line.update(gather_ok=False))
        time.sleep(30)
    """)
    assert p.returncode == 3, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert r["value"] == 123.0 and r["gather_ok"] is False and r["rccl_ok"] is False and r["watchdog"]["stage"] == "gather leg"


def test_watchdog_of_a_peer_rank_prints_nothing_and_a_cancelled_one_does_not_fire():
    p = _run_watchdog_script("""
        wd = bench.RunWatchdog(1, 2, args, limit=0.3)
        time.sleep(30)
    """)
    assert p.returncode == 3 and not [l for l in p.stdout.splitlines() if l.startswith("{")]
    p = _run_watchdog_script("""
        wd = bench.RunWatchdog(0, 1, args, limit=0.5)
        wd.cancel()
        time.sleep(1.0)
        print("clean")
    """)
    assert p.returncode == 0 and "clean" in p.stdout


def test_checksum_needs_no_int64_copy_of_the_matrix():
    import torch
    bench = _bench()
    out = torch.randint(-150, 1, (1000, 130), dtype=torch.int16)
    assert bench.checksum_int64(out, 100, rows_per_block=64) == int(out[:, :100].to(torch.int64).sum())


def test_utilisation_scalars_have_the_same_names_in_every_entry():
    """The <= 1 roofline figure must survive a parser that keeps only the scalars of `roofline`: three flat keys, the same names in
    the headline's roofline object and in each entry of other_configs (bench.flat_issued is what both call)."""
    bench = _bench()
    assert bench.ISSUED_SCALARS == ("issued_frac", "issued_frac_sustained", "valu_per_wave_row")
    pmc_form = {"source": "SQ_INSTS_VALU, r05_cfg2_pmc.csv", "frac": 0.906, "frac_at_sustained_clock": 0.98, "valu_per_nominal_wave_row": 40.26}
    flat = bench.flat_issued(pmc_form)
    assert flat == {"issued_frac": 0.906, "issued_frac_sustained": 0.98, "valu_per_wave_row": 40.26, "issued_source": "SQ_INSTS_VALU, r05_cfg2_pmc.csv"}
    gen_form = {"source": "generator instruction lists (rows_ir.py) x rows x waves", "frac": 0.9, "valu_per_row": 40}
    assert bench.flat_issued(gen_form)["valu_per_wave_row"] == 40 and bench.flat_issued(gen_form)["issued_frac_sustained"] is None
    assert set(bench.flat_issued(None)) == set(bench.ISSUED_SCALARS) | {"issued_source"} and not any(bench.flat_issued(None).values())
    src = (ROOT / "bench.py").read_text()
    assert src.count("**flat_issued(issued)") == 1 and "entry.update(flat_issued(entry[\"issued\"]))" in src


def test_every_config_is_one_job_at_every_n():
    """At N > 1 the headline is the SAME bucket cut by plan_shards, with the streamed gather inside the timed region: total work is
    fixed as N grows, so every config says "strong" — and the watchdog's stub says the same as the measured line."""
    import argparse
    bench = _bench()
    assert {cfg[6] for cfg in bench.CONFIGS.values()} == {"strong"}
    assert bench.CONFIGS[2][2:5] == (10_000, 1_000_000, 150) and bench.CONFIGS[5][2:5] == (1_000, 1_000_000, 1000)
    wd = bench.RunWatchdog(1, 4, argparse.Namespace(config=2, steps=3, warmup=1), limit=3600)
    try:
        stub = wd.stub()
    finally:
        wd.cancel()
    assert stub["scaling"] == "strong" and stub["config"]["workload"] == bench.CONFIGS[2][1] and stub["n_gpus"] == 4
    # rows per block of the sharded run's timed region: 1,000 for the 10k-query configs, ten blocks of 100 for config 5's 1,000 queries,
    # never below the reference's block
    assert bench.gather_block_rows(10_000) == 1000 and bench.gather_block_rows(1_000) == 100 and bench.gather_block_rows(300) == 100
    assert bench.gather_block_rows(50_000) == 1000 and bench.REF_BUCKET_COUNT == 100
