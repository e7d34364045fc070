"""Two processes on the HIP path (-m gpu): the multi-GPU mode of SURVEY.md section 8(e) - one process per GPU, no
collective on the data path - rehearsed on the one GPU a test box has: both ranks share device 0 and meet over gloo.
What the driver's 1/2/4/8-GPU run does with RCCL and one device per rank is this code path."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_two_rank_sharded_batch_on_the_device(tmp_path):
    """world_size 2: deal (longest first), solve every shard with the device kernels, gather, MAX - the gathered
    solutions equal the unsharded device solve and the oracle's, bit for bit."""
    script = os.path.join(ROOT, "tests", "gloo_shard_worker.py")
    port = 33500 + (os.getpid() % 2000)
    out = tmp_path / "result.npz"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script, str(out), "device"]
    r = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = np.load(out)
    assert "pantr_solve" in str(d["kernel"])                              # the shards ran on the HIP path
    assert np.array_equal(d["gathered"], d["unsharded_device"])           # sharded == unsharded
    assert np.array_equal(d["gathered"], d["serial"])                     # == the oracle
    assert d["world"] == 2 and sorted(np.concatenate([d["idx0"], d["idx1"]]).tolist()) == list(range(len(d["serial"])))


@pytest.mark.parametrize("mode", [["--headline-only"], ["--workload", "sweep", "--scaling", "strong"], []])
def test_bench_launches_its_own_ranks(mode):
    """`python bench.py --gpus 2` started bare (no torch.distributed.run around it, no WORLD_SIZE): it starts its two
    ranks itself, before any HIP call of its own, and rank 0 prints the line (PANTR_BENCH_REHEARSAL: both ranks on
    the one GPU, rendezvous over gloo).  Weak mode: two replicas of the batch; strong mode: one batch dealt to the
    ranks by its recorded costs, gathered == unsharded."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PANTR_BENCH_REHEARSAL="1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"] + mode
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0
    if "strong" in mode:
        assert line["scaling"] == "strong" and line["config"]["gathered_equals_unsharded"] is True
        assert "recorded n_hess_psi_prod" in line["config"]["partition"]
        assert line["bound"]["slowest_instance_in_kernel_ms"] <= line["bound"]["ms_per_pass"] * 1.05
    else:
        assert line["scaling"] == "weak" and line["config"]["converged"] == 600
        assert line["config"]["max_abs_dev_from_recorded_closed_loop"] == 0.0
    if mode == []:
        # default flags, several ranks (what the driver's scaling run starts): the weak headline carries the two batches
        # that do shard, dealt to the ranks - BASELINE.json configs[3] and configs[4] - as `strong`
        assert set(line["strong"]) == {"sweep", "stress"}
        for wl, instances in (("sweep", 720), ("stress", 1024)):
            rec = line["strong"][wl]
            assert rec["scaling"] == "strong" and rec["instances"] == instances and rec["gathered_equals_unsharded"] is True
            assert rec["value"] > 0 and 0 < rec["bound"]["slowest_instance_in_kernel_ms"] <= rec["bound"]["ms_per_pass"] * 1.05
    else:
        assert "strong" not in line or mode[0] == "--workload"


def test_bench_line_contract():
    """The one JSON line of `bench.py` at N = 1 (--headline-only: seconds): the keys and types the driver reads, the
    roofline object for the dominant kernel - live HIP-event timing, algorithmic bytes from the instances' own
    counters against the 8 TB/s peak - and a traffic figure only together with the source hash it was measured at."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--headline-only"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"] == "MPC-steps solved/sec, hanging_chain N=60 cold+warm, 1/2/4/8 MI355X" and d["unit"] == "steps/s"
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 3, 1) and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert abs(d["value"] - 600 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "workload" in d["config"] and d["config"]["instances_per_gpu"] == 600 and d["config"]["converged"] == 600
    assert d["config"]["max_abs_dev_from_recorded_closed_loop"] == 0.0          # measured, not a literal
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.01 < rf["frac"] < 1.0
    assert rf["kernel"] == "pantr_solve_spec_kernel<ChainModel, 3>" and 0 < rf["kernel_ms"] <= d["ms_per_step"] * 1.01
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    src = rf["traffic_source"]
    assert (rf["traffic"] is None) == ("refused" in src)                        # a figure only with its provenance
    if rf["traffic"] is not None:
        assert src["profiled_csrc_sha256"] == src["csrc_sha256"] and rf["traffic"] > rf["algorithmic_bytes_per_launch"]
