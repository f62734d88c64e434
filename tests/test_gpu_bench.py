"""bench.py on the GPU box at small sizes: the N = 1 line keeps its contract keys, the dr12q-shard
workload (BASELINE configs[2] as a strong-scaling run) shards by pixel count and reports every rank's
timings, and the N = 2 form runs as two processes sharing cuda:0 over gloo (GPDLA_BENCH_REHEARSAL=1;
RCCL refuses two ranks on one device).  bench.py is started from a clean child of conftest.py's fork
server: it starts its own ranks, and a process that holds a HIP context must not do that."""
import json
import multiprocessing as mp
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env, out_path):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=e, capture_output=True,
                       text=True, timeout=900)
    with open(out_path, "w") as f:
        json.dump({"rc": r.returncode, "stdout": r.stdout, "stderr": r.stderr[-4000:]}, f)


def bench(argv, tmp_path, env=None):
    out = tmp_path / "bench.json"
    pr = mp.get_context("forkserver").Process(target=_run, args=(argv, env or {}, str(out)))
    pr.start()
    pr.join(1000)
    if pr.is_alive():  # our own child, by handle
        pr.kill()
        pr.join()
    res = json.load(open(out))
    assert res["rc"] == 0, res["stderr"]
    lines = [ln for ln in res["stdout"].splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_headline_line_keeps_its_contract(tmp_path):
    d = bench(["--spectra", "64", "--samples", "512", "--steps", "2", "--warmup", "1"], tmp_path)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["roofline"]["kernel"] == "k_sweep_slim<3>" and 0 < d["roofline"]["frac"] < 1  # (the name in the rocprofv3 trace)
    assert d["parity"]["max_abs_delta_vs_oracle"] < 1e-8
    mix = d["config"]["dr12q_mix"]  # the production shape rides along, kernel-timed, never `value`
    assert mix["evals_per_s"] > 0 and 0 < mix["frac"] < 1 and 200 < mix["kept_pixels_mean"] < 1250
    pc = d["config"]["pcie_c"]  # ... and the PCIe-inclusive rate through the one-shot C entry
    assert "error" not in pc and pc["evals_per_s"] > 0 and 0 < pc["over_resident"] < 1.5 and "gpdla_process_batch" in pc["what"]
    k40 = d["config"]["k40"]  # ... and the 20 < k <= 40 class
    assert k40["evals_per_s"] > 0 and 0 < k40["frac"] < 1 and "k_sweep_split_slim" in k40["what"]


@pytest.mark.parametrize("gpus", [1, 2])
def test_dr12q_shard_workload(tmp_path, gpus):
    d = bench(["--gpus", str(gpus), "--workload", "dr12q-shard", "--total-spectra", "700", "--samples", "512",
               "--steps", "2", "--warmup", "1"], tmp_path, env={"GPDLA_BENCH_REHEARSAL": "1"} if gpus > 1 else {})
    assert d["scaling"] == "strong" and d["n_gpus"] == gpus and d["config"]["total_spectra"] == 700
    assert abs(d["value"] - 700 * 512 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    if gpus == 1:
        assert "per_rank" not in d and d["config"]["spectra_rank0"] == 700 and "rehearsal" not in d
    else:
        pr = d["per_rank"]
        assert sum(pr["quasars"]) == 700 and len(pr["kernel_ms"]) == len(pr["gather_ms"]) == len(pr["setup_s"]) == 2
        assert abs(pr["kept_pixels"][0] - pr["kept_pixels"][1]) < 0.02 * sum(pr["kept_pixels"])  # balanced by pixels
        assert d["rehearsal"] is True and d["backend"] == "gloo"
        assert abs(max(pr["step_ms"]) - d["ms_per_step"]) < 1e-9
