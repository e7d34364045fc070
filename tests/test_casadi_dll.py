"""The reference's CasADi-generated problem library as a cross-checker (host/casadi_dll.hpp,
pantr_casadi_dll_*, pantr_batch_check_casadi_dll; reference: drivers/casadi-dll-wrapper.cpp:25-67,
problems/problems/formulation.hpp:79-101,145-168, problems/generate_problem.py:55-61).

No generated library exists in this pipeline (CasADi and alpaqa are absent), so these tests build
tests/casadi_stub.cpp - a library written to the generated code's C convention whose bodies wrap THIS
PROJECT'S ORACLE.  A STUB PINS NOTHING about parity with the reference: device == stub only repeats
device == oracle.  What is tested here is the loader and the checker themselves - symbol lookup, work-size
queries, both sparsity forms, marshalling, and the argument order (the stub refuses a scale other than 1 and
a zl / zu that is not box D; a stub built with -DSTUB_SWAP reads two arguments in the wrong order and must be
reported as different).  With a real `<name>_ss2p_<N>.so` the same entry point gives the L1 parity figure
(INTEGRATION.md section 6).
"""
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
import pantr_amd as pa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBLEM_ID = {"hanging_chain": 0, "quadcopter": 1, "realistic_quadcopter": 2}


def build_stub(tmp_path, problem, N, *, scale=True, full_patterns=False, swap=False, tag=""):
    """g++ the stub as `<tmp>/<tag><problem>_ss2p_<N>.so` (the reference's file naming, casadi-dll-wrapper.cpp:29-31)."""
    out = os.path.join(tmp_path, f"{tag}{problem}_ss2p_{N}.so")
    cmd = ["g++", "-O1", "-ffp-contract=off", "-std=c++20", "-shared", "-fPIC", "-I", os.path.join(ROOT, "oracle"),
           f"-DSTUB_PROBLEM={PROBLEM_ID[problem]}", f"-DSTUB_N={N}", f"-DSTUB_WITH_SCALE={int(scale)}"]
    cmd += ["-DSTUB_FULL_PATTERNS"] if full_patterns else []
    cmd += ["-DSTUB_SWAP"] if swap else []
    subprocess.check_call(cmd + ["-o", out, os.path.join(ROOT, "tests", "casadi_stub.cpp")])
    return out


@pytest.fixture(scope="module")
def stubs(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("casadi_stub"))
    return {
        "chain": build_stub(d, "hanging_chain", 5),
        "chain_noscale_full": build_stub(d, "hanging_chain", 5, scale=False, full_patterns=True, tag="b_"),
        "chain_swap": build_stub(d, "hanging_chain", 5, swap=True, tag="s_"),
        "quad": build_stub(d, "quadcopter", 3),
        "quad_noscale": build_stub(d, "quadcopter", 3, scale=False, tag="b_"),
        "quad_swap": build_stub(d, "quadcopter", 3, swap=True, tag="s_"),
        "rquad_full": build_stub(d, "realistic_quadcopter", 3, full_patterns=True),
    }


# ---- CPU: the loader resolves, sizes and calls the generated functions ---------------------------------
def test_loader_reads_counts_patterns_and_work_sizes(stubs):
    for key, n, m, nx, n_psi in (("chain", 15, 0, 57, 7), ("chain_noscale_full", 15, 0, 57, 6),
                                 ("quad", 12, 16, 9, 7), ("quad_noscale", 12, 16, 9, 6), ("rquad_full", 12, 16, 9, 7)):
        d = pa.CasadiDll(stubs[key])
        s = [1] if n_psi == 7 else []
        info = d.function_info("psi_grad_psi")
        assert (info["n_in"], info["n_out"]) == (n_psi, 2)
        assert info["nnz_in"] == [n, nx, m, m] + s + [m, m] and info["nnz_out"] == [1, n]
        assert info["work"] == (n_psi + 2, 3, 3, 7)            # what the stub's psi_grad_psi_work reports
        info = d.function_info("hess_psi_prod")
        assert info["nnz_in"] == [n, nx, m, m, 1, m, m, n] and info["nnz_out"] == [n]
        assert d.function_info("g")["nnz_out"] == [m] and d.function_info("psi")["nnz_out"] == [1, m]
        d.close()


def test_loader_calls_match_the_functions_the_stub_wraps(stubs):
    rng = np.random.default_rng(3)
    # hanging chain (m = 0): empty y, Sigma, zl, zu
    N, n = 5, 15
    x0 = ob.initial_state() + 0.01 * rng.standard_normal(57)
    u, v = rng.uniform(-1, 1, n), rng.standard_normal(n)
    for key, s in (("chain", [1.0]), ("chain_noscale_full", [])):
        d = pa.CasadiDll(stubs[key])
        psi, grad = d.call("psi_grad_psi", [u, x0, None, None] + s + [None, None], [1, n])
        psi_o, grad_o = ob.psi_grad(N, x0, u)
        assert psi[0] == psi_o and np.array_equal(grad, grad_o)
        (Hv,) = d.call("hess_psi_prod", [u, x0, None, None, 1.0, None, None, v], [n])
        assert np.array_equal(Hv, ob.hess_prod(N, x0, u, v))
        assert d.call("psi", [u, x0, None, None] + s + [None, None], [1, 0])[0][0] == ob.psi(N, x0, u)
    # quadcopter (m = 4 N + 4): multipliers, penalties, box D
    N, n, m = 3, 12, 16
    x0 = ob.quad_initial_state() + 0.01 * rng.standard_normal(9)
    u = ob.quad_initial_guess(N) + rng.uniform(-0.05, 0.05, n)
    y, Sig, v = rng.standard_normal(m), rng.uniform(1, 100, m), rng.standard_normal(n)
    dflt = pa.problem_defaults("quadcopter", N)                # box D as generate_problem.py:62-69 would write it
    dlb, dub = dflt["D_lb"], dflt["D_ub"]
    d = pa.CasadiDll(stubs["quad"])
    psi, grad = d.call("psi_grad_psi", [u, x0, y, Sig, 1.0, dlb, dub], [1, n])
    psi_o, grad_o, yhat_o = ob.quad_psi_grad("quadcopter", N, x0, y, Sig, u)
    assert psi[0] == psi_o and np.array_equal(grad, grad_o)
    psi2, yhat = d.call("psi", [u, x0, y, Sig, 1.0, dlb, dub], [1, m])
    assert psi2[0] == psi_o and np.array_equal(yhat, yhat_o)
    (Hv,) = d.call("hess_psi_prod", [u, x0, y, Sig, 1.0, dlb, dub, v], [n])
    assert np.array_equal(Hv, ob.quad_hess_prod("quadcopter", N, x0, y, Sig, u, v))
    (g,) = d.call("g", [u, x0], [m])
    assert np.array_equal(g, ob.quad_f_g("quadcopter", N, x0, u)[1])


def test_loader_errors_follow_the_reference(stubs, tmp_path):
    with pytest.raises(pa.PantrError) as e:                    # alpaqa::CasADiProblem: the file cannot be opened
        pa.CasadiDll(str(tmp_path / "nowhere_ss2p_5.so"))
    assert e.value.code == -4 and "cannot open" in str(e.value)
    d = pa.CasadiDll(stubs["chain"])
    with pytest.raises(pa.PantrError) as e:                    # std::out_of_range (casadi-dll-wrapper.cpp:104-106)
        d.function_info("hess_L")
    assert e.value.code == -2 and "hess_L" in str(e.value)
    with pytest.raises(pa.PantrError) as e:                    # wrong argument count
        d.call("psi_grad_psi", [np.zeros(15), np.zeros(57)], [1, 15])
    assert e.value.code == -1
    with pytest.raises(pa.PantrError) as e:                    # the generated function itself fails (scale != 1 in the stub)
        d.call("psi_grad_psi", [np.zeros(15), ob.initial_state(), None, None, 2.0, None, None], [1, 15])
    assert e.value.code == -4 and "returned 1" in str(e.value)
    # incref on resolve, decref on close (the stub counts)
    import ctypes
    raw = ctypes.CDLL(stubs["quad_noscale"])
    d2 = pa.CasadiDll(stubs["quad_noscale"])
    d2.function_info("psi"), d2.function_info("g"), d2.function_info("psi")
    assert raw.stub_refcount() == 2
    d2.close()
    assert raw.stub_refcount() == 0


# ---- GPU: the checker ------------------------------------------------------------------------------------
BAR = 1e-10  # BASELINE.json north_star: "<= 1e-10 relative on identical inputs"


@pytest.mark.gpu
def test_checker_reports_no_difference_against_the_stub_and_a_difference_against_a_swapped_one(stubs):
    b = pa.PantrBatch("hanging_chain", 5, 4)
    for key in ("chain", "chain_noscale_full"):
        err = b.check_casadi_dll(stubs[key], samples=6)          # more samples than the handle's batch: two chunks
        assert err["g"] is None and err["yhat"] is None          # m = 0
        assert max(err[k] for k in ("psi", "grad_psi", "hess_psi_prod")) <= BAR, err
    err = b.check_casadi_dll(stubs["chain_swap"], samples=4)     # reads (v, x) for (x, v) in hess_psi_prod
    assert err["psi"] <= BAR and err["grad_psi"] <= BAR and err["hess_psi_prod"] > 1e-3, err
    with pytest.raises(pa.PantrError) as e:                      # a library of another problem is refused, not mis-fed
        b.check_casadi_dll(stubs["quad"])
    assert e.value.code == -1 and "sizes" in str(e.value)
    b.close()
    for problem, key in (("quadcopter", "quad"), ("quadcopter", "quad_noscale"), ("realistic_quadcopter", "rquad_full")):
        b = pa.PantrBatch(problem, 3, 8)
        err = b.check_casadi_dll(stubs[key], samples=8)
        assert all(err[k] is not None and err[k] <= BAR for k in pa.DLL_CHECK_FIELDS), err
        b.close()
    b = pa.PantrBatch("quadcopter", 3, 8)
    err = b.check_casadi_dll(stubs["quad_swap"], samples=8)      # reads (Sigma, y) for (y, Sigma)
    assert err["g"] <= BAR and min(err[k] for k in ("psi", "grad_psi", "hess_psi_prod", "yhat")) > 1e-3, err
    b.close()


@pytest.mark.gpu
def test_driver_check_dll_option(stubs, tmp_path):
    """`benchmark-mpc-driver <dir>/<name> <N> ss2p check_dll=true`: finds <dir>/<name>_ss2p_<N>.so next to the
    sidecar files (casadi-dll-wrapper.cpp:29-31), prints the comparison and fails on a library that differs."""
    drv = os.path.join(ROOT, "pantr-cdc2023-experiments_amd", "bin", "benchmark-mpc-driver")
    d = os.path.dirname(stubs["quad"])
    pa.write_problem_files(d, "quadcopter", "ss2p", 3)
    r = subprocess.run([drv, os.path.join(d, "quadcopter"), "3", "ss2p", "method=pantr", "check_dll=true", "num_sim=1",
                        "num_dist=0", f"results_name={tmp_path}/r"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "check_dll: " in r.stdout and "hess_psi_prod" in r.stdout and "agree" in r.stdout
    # the swapped library under the same name in another directory
    d2 = str(tmp_path / "swapped")
    os.makedirs(d2)
    os.replace(build_stub(d2, "quadcopter", 3, swap=True), os.path.join(d2, "quadcopter_ss2p_3.so"))
    pa.write_problem_files(d2, "quadcopter", "ss2p", 3)
    r = subprocess.run([drv, os.path.join(d2, "quadcopter"), "3", "ss2p", "method=pantr", "check_dll=true", "num_sim=1",
                        "num_dist=0", f"results_name={tmp_path}/r2"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "DIFFER" in r.stdout + r.stderr
