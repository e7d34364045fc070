"""CPU tests of the host logic: C-ABI surface, option parser, error behaviour, sharding.

No compute call is made here (there is no GPU): the library must load, export every symbol
include/pantr_mi355x.h declares, and refuse to solve without a device.
"""
import os
import re
import subprocess
import sys
import time

import numpy as np
import pytest

import pantr_amd as pa
from pantr_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pantr_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pantr_[a-z_A-Z0-9]+)\s*\(", hdr))
    assert declared == set(pa.EXPORTS)
    L = pa.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert "gfx950" in pa.version()


def test_stats_struct_layout():
    hdr = open(os.path.join(ROOT, "include", "pantr_mi355x.h")).read()
    body = re.search(r"typedef struct pantr_stats \{(.*?)\} pantr_stats_t;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            names += [x.strip() for x in decl.split(None, 1)[1].split(",")]
    names = [x.split('[')[0] for x in names]   # phase_s[4]
    assert tuple(names) == pa.STATS_DTYPE.names
    assert pa.STATS_DTYPE.itemsize == 6 * 8 + 16 * 4 + 2 * 8 + 6 * 4 + 8 + 4 * 8


def test_driver_defaults_and_dodo_options():
    """mpc-driver.cpp:51-53,109-111 defaults, then dodo.py:12-33 overrides."""
    p = pa.PantrParams()
    assert p.get("solver.max_iter") == 50000
    assert p.get("solver.stop_crit") == 6          # FPRNorm
    assert p.get("alm.max_iter") == 200 and p.get("alm.tolerance") == 1e-8
    assert p.get("solver.radius_factor_good") == 2.5 and p.get("dir.hessian_vec_factor") == 0.5
    assert np.isnan(p.get("solver.initial_radius"))
    for o in pa.DODO_PANTR_OPTS:
        p.set(o)
    assert p.get("solver.stop_crit") == 4          # ProjGradUnitNorm
    assert p.get("alm.max_time") == 30.0 and p.get("alm.max_iter") == 300
    assert p.get("solver.radius_factor_good") == 10 and p.get("solver.ratio_threshold_good") == 0.5
    assert p.get("accel.tol_scale") == 1e-2 and p.get("dir.hessian_vec_factor") == 1
    assert p.get("alm.initial_penalty") == 1e4 and p.get("alm.penalty_update_factor") == 5


def test_option_grammar():
    p = pa.PantrParams()
    p.set("solver.max_iter=250")                   # dodo.py:98-100
    assert p.get("solver.max_iter") == 250
    p.set("alm.max_time=500ms")
    assert p.get("alm.max_time") == 0.5
    p.set("solver.max_time=2min")
    assert p.get("solver.max_time") == 120
    p.set("solver.disable_acceleration=true")
    assert p.get("solver.disable_acceleration") == 1
    p.set("solver.Lipschitz.L_0=3.5")
    assert p.get("solver.Lipschitz.L_0") == 3.5
    # options addressed to other objects are ignored, as set_params(obj, prefix, opts) does
    for foreign in ("num_dist=5", "u_dist=-1,1,1", "warm=true", "method=pantr",
                    "results_name=mpc-x", "ipopt.tol=1e-8"):
        p.set(foreign)
    for bad in ("solver.no_such_key=1", "solver.max_iter=abc", "solver.stop_crit=Nope",
                "alm.max_time=30parsecs", "accel.tol_scale", "dir.finite_diff=maybe"):
        with pytest.raises(pa.PantrError) as e:
            p.set(bad)
        assert e.value.code == -1
    with pytest.raises(pa.PantrError):
        p.get("solver.no_such_key")
    # speculation=auto|on|off: an option of this implementation, spelled like method=
    assert p.get("speculation") == -1
    for text, value in (("speculation=on", 1), ("speculation=off", 0), ("speculation=auto", -1)):
        p.set(text)
        assert p.get("speculation") == value
    with pytest.raises(pa.PantrError) as e:
        p.set("speculation=maybe")
    assert e.value.code == -1 and "speculation" in str(e.value)
    # residency=auto|<k>, spec_patience=<duration>: options of this implementation as well
    assert p.get("residency") == 0 and p.get("spec_patience") == 1.0
    for text, key, value in (("residency=3", "residency", 3), ("residency=auto", "residency", 0),
                             ("spec_patience=250ms", "spec_patience", 0.25), ("spec_patience=0", "spec_patience", 0.0)):
        p.set(text)
        assert p.get(key) == value
    for bad in ("residency=-1", "residency=many", "spec_patience=soon", "spec_patience=-1s"):
        with pytest.raises(pa.PantrError) as e:
            p.set(bad)
        assert e.value.code == -1 and bad.split("=")[0] in str(e.value)
    # launch_order=natural|previous|auto (default: natural)
    assert p.get("launch_order") == 0
    for text, value in (("launch_order=previous", 1), ("launch_order=natural", 0), ("launch_order=auto", -1)):
        p.set(text)
        assert p.get("launch_order") == value
    with pytest.raises(pa.PantrError) as e:
        p.set("launch_order=random")
    assert e.value.code == -1 and "launch_order" in str(e.value)


def test_no_cpu_fallback():
    """Without a HIP device the product refuses to create a problem: there is no CPU path."""
    if pa.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(pa.PantrError) as e:
        pa.PantrBatch("hanging_chain", 60, 1)
    assert e.value.code == -3 and "no CPU path" in str(e.value)


def test_create_argument_errors():
    with pytest.raises(pa.PantrError) as e:
        pa.PantrBatch("no_such_problem", 10, 1)
    assert e.value.code == -2
    with pytest.raises(pa.PantrError) as e:
        pa.PantrBatch("hanging_chain", 0, 1)
    assert e.value.code == -1


def test_product_does_not_import_the_oracle():
    """Nothing under the package may reference oracle/ (SURVEY.md section 8c rule)."""
    pkg = os.path.join(ROOT, "pantr-cdc2023-experiments_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_binding" not in text and "liboracle" not in text, f
                assert not re.search(r'#include\s+"[^"]*oracle', text), f


def test_algorithmic_bytes_formula():
    """SURVEY.md section 8(d) worked numbers for hanging_chain N=60."""
    st = np.zeros(1, dtype=pa.STATS_DTYPE)
    st["n_psi"] = 1
    assert pa.algorithmic_bytes(st, 60, 57, 3) == 8 * (180 + 57 + 60 * 57)          # 29.3 KB
    st["n_psi"], st["n_grad_L"] = 0, 1
    assert pa.algorithmic_bytes(st, 60, 57, 3) == 8 * (360 + 57 + 2 * 60 * 57)      # 58.1 KB
    st["n_grad_L"], st["n_hess_psi_prod"] = 0, 1
    assert pa.algorithmic_bytes(st, 60, 57, 3) == 8 * (540 + 57 + 4 * 60 * 57)      # 114.2 KB
    st["n_hess_psi_prod"], st["cg_iterations"], st["iterations"] = 0, 1, 1
    assert pa.algorithmic_bytes(st, 60, 57, 3) == 8 * (17 * 180 + 15 * 180)


def test_lpt_partition_and_launch_order():
    costs = np.array([5, 1, 1, 1, 4, 4, 1, 3.0])
    shards = sharding.lpt_partition(costs, 2)
    assert sorted(np.concatenate(shards).tolist()) == list(range(8))
    loads = [costs[s].sum() for s in shards]
    assert max(loads) - min(loads) <= 1.0
    assert sharding.launch_order(costs).tolist() == [0, 4, 5, 7, 1, 2, 3, 6]
    assert sharding.shard_indices(7, 1, 3).tolist() == [1, 4]
    # empty shards are legal (more ranks than instances)
    assert [len(s) for s in sharding.lpt_partition([1.0], 3)] == [1, 0, 0]


def test_two_rank_sharded_batch_gloo(tmp_path):
    """world_size 2 over gloo: shard, solve (oracle stands in for the device), gather, MAX."""
    script = os.path.join(ROOT, "tests", "gloo_shard_worker.py")
    port = 29500 + (os.getpid() % 2000)
    out = tmp_path / "result.npz"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script, str(out)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = np.load(out)
    assert np.array_equal(d["gathered"], d["serial"])      # sharded == unsharded, bit for bit
    assert d["tmax"] >= max(d["t0"], d["t1"]) - 1e-12 and d["world"] == 2
    assert sorted(np.concatenate([d["idx0"], d["idx1"]]).tolist()) == list(range(len(d["serial"])))


def test_two_rank_failure_ends_both_ranks_gloo(tmp_path):
    """A check failing on one rank ends every rank at the same point (no peer left in a collective)."""
    script = os.path.join(ROOT, "tests", "gloo_shard_worker.py")
    port = 31500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script, str(tmp_path / "unused.npz"), "fail"]
    t = time.time()
    r = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and time.time() - t < 120
    assert "rank 1 failed its check" in r.stderr and "another rank reported a failure" in r.stderr


def test_mt19937_64_and_config5_disturbances():
    from pantr_amd import workloads
    g = workloads.MT19937_64()
    for _ in range(9999):
        g()
    assert g() == 9981545732273789042          # C++ standard: 10000th output of a default mt19937_64
    d = workloads.config5_disturbances(8)
    assert d.shape == (8, 3) and np.all((d >= -1) & (d < 1))
    assert np.array_equal(d, workloads.config5_disturbances(8))
    # first draw of mt19937_64(12345), scaled: x / 2^64 * 2 - 1
    x = workloads.MT19937_64(12345)()
    assert abs(d[0, 0] - (x / 2.0 ** 64 * 2 - 1)) < 1e-15


def test_dodo_enumeration():
    """The PANTR rows of dodo.py:69-111 for hanging_chain."""
    from pantr_amd import workloads
    tasks = workloads.dodo_pantr_tasks("hanging_chain")
    assert len(tasks) == 12
    assert tasks[0]["name"] == "mpc-hanging_chain-10,60-warm-pantr"
    assert tasks[-1]["name"] == "mpc-hanging_chain-60,60-cold-pantr"
    t = tasks[5]
    assert t["argv"][:4] == ["hanging_chain", "60", "ss2p", "method=pantr"]
    assert "num_dist=5" in t["argv"] and "u_dist=-1,1,1" in t["argv"] and "warm=true" in t["argv"]
    assert "solver.radius_factor_good=10" in t["argv"] and "alm.max_time=30s" in t["argv"]
    q = workloads.dodo_pantr_tasks("realistic_quadcopter")[0]["argv"]
    assert q.index("accel.tol_scale=1") > q.index("accel.tol_scale=1e-2")  # later option overrides


# ---- the data half of the problem-side boundary (sidecar files) -----------------------------------
def test_problem_defaults_match_the_reference_models():
    """Compiled-in data against problems/problems/hanging_chain.cpp:107-135 and quadcopter.py:55-111."""
    import oracle_binding as ob
    assert pa.problem_dims("hanging_chain", 60) == (180, 0, 57, 3, 0, 0)
    assert pa.problem_dims("dl:realistic_quadcopter", 60) == (240, 244, 9, 4, 4, 4)
    d = pa.problem_defaults("hanging_chain", 7)
    assert np.all(d["C_lb"] == -1) and np.all(d["C_ub"] == 1) and np.all(d["guess"] == 0)
    assert np.array_equal(d["param"], ob.initial_state())
    q = pa.problem_defaults("quadcopter", 5)
    assert np.array_equal(q["C_lb"][:4], [0, -0.1, -0.1, -0.1]) and np.array_equal(q["C_ub"][:4], [9.81 * 5, 0.1, 0.1, 0.1])
    assert q["D_lb"][3] == -np.inf and q["D_ub"][2] == np.inf and q["D_lb"].shape == (24,)
    assert np.array_equal(q["param"], ob.quad_initial_state()) and np.array_equal(q["guess"], ob.quad_initial_guess(5))
    with pytest.raises(pa.PantrError):
        pa.problem_dims("lasso", 5)


@pytest.mark.parametrize("problem,N", [("hanging_chain", 12), ("quadcopter", 7)])
def test_sidecar_files_round_trip(tmp_path, problem, N):
    """Files in the layout of generate_problem.py:62-73 (incl. inf bounds and the empty rows of a
    problem without constraints) are read back exactly; modified data are honoured."""
    n, m, nx, nu = pa.problem_dims(problem, N)[:4]
    stem = pa.write_problem_files(str(tmp_path), problem, "ss2p", N)
    assert os.path.exists(stem + ".csv") and os.path.exists(stem + ".guess.csv")
    assert len(open(stem + ".csv").read().split("\n")) == 8          # seven rows + final newline
    d, ref = pa.load_problem_files(os.path.join(str(tmp_path), problem), "ss2p", N), pa.problem_defaults(problem, N)
    assert d["has_csv"] and d["penalty_alm_split"] == m               # generate_problem.py:32
    for k in ref:
        assert np.array_equal(d[k], ref[k]), k
    rng = np.random.default_rng(5)
    mod = dict(C_lb=np.tile(-rng.uniform(0.2, 0.9, nu), N), C_ub=np.tile(rng.uniform(0.2, 0.9, nu), N),
               param=rng.standard_normal(nx), guess=rng.standard_normal(n), penalty_alm_split=0)
    pa.write_problem_files(str(tmp_path), problem, "ss2p", N, mod)
    d = pa.load_problem_files(os.path.join(str(tmp_path), problem), "ss2p", N)
    for k in ("C_lb", "C_ub", "param", "guess"):
        assert np.array_equal(d[k], mod[k]), k
    assert d["penalty_alm_split"] == 0 and np.array_equal(d["D_lb"], ref["D_lb"])
    # only the guess file (alpaqa loads the .csv only when it exists): compiled-in bounds and state
    os.remove(stem + ".csv")
    d = pa.load_problem_files(os.path.join(str(tmp_path), problem), "ss2p", N)
    assert not d["has_csv"] and np.array_equal(d["guess"], mod["guess"]) and np.array_equal(d["C_lb"], ref["C_lb"])


def test_sidecar_files_fail_loudly(tmp_path):
    problem, N, d = "quadcopter", 6, str(tmp_path)
    path = os.path.join(d, problem)
    with pytest.raises(pa.PantrError, match="Failed to open guess file"):   # casadi-dll-wrapper.cpp:58-59
        pa.load_problem_files(path, "ss2p", N)
    pa.write_problem_files(d, problem, "ss2p", N)
    pa.load_problem_files(path, "ss2p", N)
    with pytest.raises(pa.PantrError, match="Failed to open guess file"):   # another horizon: other files
        pa.load_problem_files(path, "ss2p", N + 1)
    with pytest.raises(pa.PantrError, match="not available"):
        pa.load_problem_files(path, "ms2p", N)
    ref = pa.problem_defaults(problem, N)
    bad = {
        "needs 24": dict(guess=np.zeros(20)),                                  # n = 4 N
        "box C \\(lower\\) has": dict(C_lb=ref["C_lb"][:-1]),
        "param \\(the initial state\\) has": dict(param=np.zeros(8)),
        "not the stage box tiled": dict(C_ub=np.r_[ref["C_ub"][:-1], 0.05]),
        "box D .* not the stage box tiled": dict(D_lb=np.r_[ref["D_lb"][:-1], -1.0]),
        "box C is empty": dict(C_lb=np.tile([0, 0.2, -0.1, -0.1], N)),
        "l1 regularisation": dict(l1_reg=[0.5]),
        "penalty_alm_split exceeds": dict(penalty_alm_split=1000),
    }
    for msg, mod in bad.items():
        pa.write_problem_files(d, problem, "ss2p", N, mod)
        with pytest.raises(pa.PantrError, match=msg):
            pa.load_problem_files(path, "ss2p", N)
    pa.write_problem_files(d, problem, "ss2p", N)
    with open(os.path.join(d, f"{problem}_ss2p_{N}.guess.csv"), "w") as f:
        f.write("1.0,abc\n28\n")
    with pytest.raises(pa.PantrError, match="cannot parse 'abc'"):
        pa.load_problem_files(path, "ss2p", N)
    # a problem this library has no device code for
    for ext in (".csv", ".guess.csv"):
        os.replace(os.path.join(d, f"{problem}_ss2p_{N}{ext}"), os.path.join(d, f"lasso_ss2p_{N}{ext}"))
    with open(os.path.join(d, f"lasso_ss2p_{N}.guess.csv"), "w") as f:
        f.write("0\n0\n")
    with pytest.raises(pa.PantrError, match="unknown problem 'lasso'"):
        pa.load_problem_files(os.path.join(d, "lasso"), "ss2p", N)


def test_eval_wrappers_validate_shapes():
    """A 1-D x0 must not be read as a batch of nx instances (no device call is reached)."""
    class Dummy(pa.PantrBatch):
        def __init__(self):
            self._h, self.n, self.m, self.nx, self.nu = None, 9, 0, 57, 3
    b = Dummy()
    with pytest.raises(ValueError):
        b.eval_psi(np.zeros(57), np.zeros(9))
    with pytest.raises(ValueError):
        b.eval_psi_grad_psi(np.zeros((2, 57)), np.zeros((2, 8)))
    with pytest.raises(ValueError):
        b.simulate_dynamics(np.zeros((2, 57)), np.zeros((2, 2)))
    with pytest.raises(ValueError):
        b.eval_prox_grad_step(1.0, np.zeros((1, 9)), np.zeros((1, 57)))


def test_explicit_vmcnt_waits_are_covered_in_the_compiled_kernel():
    """scripts/check_vmem_counts.py (the Makefile runs it on every library it links): every marked `s_waitcnt vmcnt(n)
    expcnt(6)` of the two-stages-ahead product sweeps is preceded, on EVERY path of the compiled control flow, by at least
    n vector-memory instructions since the synchronisation point before it; and no stage loop of a product, in any of the
    hanging chain's PANTR kernels, waits for tape rows it has only just requested."""
    if not os.path.exists(pa.LIB_PATH) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("needs the built library and llvm-objdump")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_vmem_counts.py"), pa.LIB_PATH],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("marked waits checked on every path, 0 not covered") >= 4   # three solve kernels + the evaluation kernel


def test_the_vmcnt_checker_finds_what_it_should():
    """The checker on hand-written instruction streams: a wait that a path reaches with too few operations is reported,
    the compiler's two lowerings of `if (c) A else B` (two branches on one condition register; a flag register set to -1
    before and 0 inside A) are not, and a path from a full drain with nothing issued passes."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    argv, sys.argv = sys.argv, [sys.argv[0]]
    try:
        import check_vmem_counts as cv
    finally:
        sys.argv = argv

    def run(lines):
        ins = [l.strip() for l in lines.strip().splitlines()]
        addr = [4 * i for i in range(len(ins))]            # (every instruction four bytes: branch offsets count instructions)
        return cv.check_marked_waits("t", ins, addr)

    LD, ST = "global_load_lds_dwordx4 v[0:1], off", "global_store_dwordx2 v[0:1], v[2:3], off"
    W = lambda n: f"s_waitcnt vmcnt({n}) expcnt(6)"
    # straight line: 2 + 3 operations behind the wait before -> vmcnt(5) is covered, vmcnt(6) is not
    assert run("\n".join(["s_nop 0", W(2), LD, LD, ST, ST, ST, W(5), "s_endpgm"]))[:2] == (2, 0)
    assert run("\n".join(["s_nop 0", W(2), LD, LD, ST, ST, ST, W(6), "s_endpgm"]))[:2] == (2, 1)
    # a request under a condition: the path around it has only the three stores
    skip = ["s_nop 0", W(2), "s_cbranch_vccz 2", LD, LD, ST, ST, ST, W(5), "s_endpgm"]
    assert run("\n".join(skip))[:2] == (2, 1)
    # if / else on one unchanged condition register: one of the two requests is always made
    both = ["s_nop 0", W(2), "s_cbranch_vccz 2", LD, LD, "s_cbranch_vccnz 2", LD, LD, ST, ST, ST, W(5), "s_endpgm"]
    assert run("\n".join(both))[:2] == (2, 0)
    # ... but not if the register is written in between
    clob = both[:5] + ["v_cmp_lt_f64_e32 vcc, v[0:1], v[2:3]"] + both[5:]
    clob[2] = "s_cbranch_vccz 2"
    assert run("\n".join(clob))[:2] == (2, 1)
    # if / else through a flag register, the flag set BEFORE the synchronisation point
    flag = ["s_nop 0", "s_mov_b64 s[48:49], -1", W(2), "s_and_b64 vcc, exec, s[44:45]", "s_cbranch_vccz 3", LD, LD,
            "s_mov_b64 s[48:49], 0", "s_andn2_b64 vcc, exec, s[48:49]", "s_cbranch_vccnz 2", LD, LD, ST, ST, ST, W(5), "s_endpgm"]
    assert run("\n".join(flag))[:2] == (2, 0)
    flag[1] = "s_mov_b64 s[48:49], 0"                       # the flag says "A ran" on the path that skipped A: both may be skipped
    assert run("\n".join(flag))[:2] == (2, 1)
    # from a full drain: no LDS-DMA request since -> the rows were requested before it and have landed, passes (whatever
    # stores were issued); a request but too few operations behind it -> reported
    assert run("\n".join(["s_nop 0", "s_waitcnt vmcnt(0)", W(5), "s_endpgm"]))[:2] == (1, 0)
    assert run("\n".join(["s_nop 0", "s_waitcnt vmcnt(0)", ST, W(5), "s_endpgm"]))[:2] == (1, 0)
    assert run("\n".join(["s_nop 0", "s_waitcnt vmcnt(0)", LD, W(5), "s_endpgm"]))[:2] == (1, 1)
    assert run("\n".join(["s_nop 0", "s_waitcnt vmcnt(0)", LD, ST, ST, ST, ST, ST, W(5), "s_endpgm"]))[:2] == (1, 0)
    # the conservative build
    assert run("\n".join(["s_nop 0", "s_waitcnt vmcnt(0) expcnt(6)", "s_endpgm"])) == (0, 0, 1)


def test_bench_started_bare_launches_ranks_and_reports_their_failure():
    """`python bench.py --gpus 2` without WORLD_SIZE becomes the launcher (before any HIP call): it starts two rank
    processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set and leaves with a failure when they fail - which
    they do here, loudly, because there is no GPU (the product has no CPU path)."""
    if pa.device_count() > 0:
        pytest.skip("a HIP device is present (tests/test_gpu_multirank.py runs the real thing)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    t = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and time.time() - t < 120
    assert r.stderr.count("bench.py needs an MI355X") == 2          # both ranks were started, both said why they left
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]  # and no result line was made up


def test_concurrent_batches_take_four_streams_longest_first():
    """bench.plan_streams: batches launched together go on at most four streams (the runtime's hardware queues per
    process: streams that share one run in turn), the longest first, a fifth and sixth behind the least loaded."""
    sys.path.insert(0, ROOT)
    import bench

    class FakeStream:
        pass

    class FakeTorch:
        class cuda:
            Stream = FakeStream

    class R:
        def __init__(self, B, N):
            self.B, self.w = B, {"horizon": N}

    rs = [R(120, N) for N in (10, 20, 30, 40, 50, 60)]
    plan = bench.plan_streams(rs, FakeTorch)
    assert [r.w["horizon"] for r, _ in plan] == [60, 50, 40, 30, 20, 10]           # longest first
    streams = [s for _, s in plan]
    assert len({id(s) for s in streams}) == 4                                       # four streams for six batches
    assert len({id(s) for s in streams[:4]}) == 4                                   # the four longest each on their own
    assert streams[4] is streams[3] and streams[5] is streams[3]                    # 20 and 10 behind 30: 900 + 400 + 100 < 1600
    assert [r for r, _ in bench.plan_streams(rs[:2], FakeTorch)] == [rs[1], rs[0]]  # fewer batches: a stream each
    assert len({id(s) for _, s in bench.plan_streams(rs[:2], FakeTorch)}) == 2
    empty = [R(0, 60), R(5, 10)]                                                    # a rank that got no rows of a horizon
    assert [r for r, _ in bench.plan_streams(empty, FakeTorch)] == [empty[1], empty[0]]
