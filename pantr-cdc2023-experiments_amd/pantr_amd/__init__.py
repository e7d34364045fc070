"""Python host binding (ctypes) of the MI355X-native batched PANTR solver.

Thin mirror of include/pantr_mi355x.h; the compute lives in
lib/libpantr_mi355x.so (hand-written HIP for gfx950).  There is no CPU fallback:
if the library is missing, or no HIP device is present when a compute call is
made, this module raises.

The Python-level names follow the reference's driver surface:
  PantrBatch.solve        <->  stats = solver(problem, u, y)       drivers/mpc-driver.cpp:217
  PantrBatch.set_params   <->  alpaqa::params::set_params          drivers/mpc-driver.cpp:54,61,62,113
  DODO_PANTR_OPTS         <->  SOLVER_OPT["pantr"]                 new-benchmarks-paper/dodo.py:12-33
  DODO_OPTS[method]       <->  SOLVER_OPT[method] + method=        new-benchmarks-paper/dodo.py:24-40
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# PANTR_LIB_PATH: load another build of the library (e.g. the profiling build `make phase` writes to
# lib/libpantr_mi355x_phase.so) without ever overwriting the product library.
LIB_PATH = os.environ.get("PANTR_LIB_PATH") or os.path.join(PKG_DIR, "lib", "libpantr_mi355x.so")

_dbl, _i32 = C.c_double, C.c_int32

STATUS_NAMES = ("Busy", "Converged", "MaxTime", "MaxIter", "NotFinite", "NoProgress",
                "Interrupted", "Exception")
TRACE_COLS = ("eps", "psi", "pnorm", "gamma", "radius", "rho", "q_model", "fbe", "nJ",
              "cg_its", "accepted")

# new-benchmarks-paper/dodo.py:12-22 (OPT_ALM) + :25-33 (SOLVER_OPT["pantr"])
DODO_PANTR_OPTS = (
    "alm.max_iter=300", "alm.max_time=30s", "alm.initial_penalty=1e4",
    "alm.penalty_update_factor=5", "alm.initial_tolerance=1e2",
    "alm.tolerance_update_factor=0.1", "solver.stop_crit=ProjGradUnitNorm",
    "alm.tolerance=1e-08", "alm.dual_tolerance=1e-08",
    "solver.radius_factor_rejected=0.35", "solver.radius_factor_acceptable=0.99",
    "solver.radius_factor_good=10", "solver.ratio_threshold_acceptable=0.3",
    "solver.ratio_threshold_good=0.5", "dir.hessian_vec_factor=1", "accel.tol_scale=1e-2",
)
# new-benchmarks-paper/dodo.py:12-22 (OPT_ALM); :34-36 SOLVER_OPT["panoc"]; :37-40 SOLVER_OPT["strucpanoc"]
DODO_ALM_OPTS = DODO_PANTR_OPTS[:9]
DODO_PANOC_OPTS = ("method=panoc",) + DODO_ALM_OPTS + ("accel.memory=50",)
DODO_STRUCPANOC_OPTS = ("method=strucpanoc",) + DODO_ALM_OPTS + ("accel.memory=50", "dir.hessian_vec=0")
DODO_OPTS = {"pantr": ("method=pantr",) + DODO_PANTR_OPTS, "panoc": DODO_PANOC_OPTS,
             "strucpanoc": DODO_STRUCPANOC_OPTS}
PANOC_TRACE_COLS = ("eps", "psi", "pnorm", "gamma", "tau", "fbe", "lbfgs_pairs")
# new-benchmarks-paper/dodo.py:94-95
DODO_HANGING_CHAIN_OPTS = ("num_dist=5", "u_dist=-1,1,1")


class PantrStats(C.Structure):
    """pantr_stats_t"""
    _fields_ = [(k, _dbl) for k in ("eps", "final_gamma", "final_psi", "final_fbe",
                                    "final_radius", "elapsed_s")] + \
               [(k, _i32) for k in ("status", "iterations", "cg_iterations", "accepted",
                                    "rejected", "qub_backtracks", "n_psi", "n_grad_psi",
                                    "n_psi_grad_psi", "n_grad_L", "n_hess_psi_prod",
                                    "n_prox_grad_step", "outer_iterations",
                                    "inner_convergence_failures", "n_hess_psi_prod_calls",
                                    "spec_timeouts")] + \
               [(k, _dbl) for k in ("delta", "penalty_norm")] + \
               [(k, _i32) for k in ("linesearch_failures", "linesearch_backtracks", "lbfgs_failures",
                                    "lbfgs_rejected", "tau_1_accepted", "count_tau")] + \
               [("sum_tau", _dbl), ("phase_s", _dbl * 4)]


STATS_DTYPE = np.dtype([(k, np.float64) for k in ("eps", "final_gamma", "final_psi", "final_fbe",
                                                  "final_radius", "elapsed_s")] +
                       [(k, np.int32) for k in ("status", "iterations", "cg_iterations",
                                                "accepted", "rejected", "qub_backtracks", "n_psi",
                                                "n_grad_psi", "n_psi_grad_psi", "n_grad_L",
                                                "n_hess_psi_prod", "n_prox_grad_step",
                                                "outer_iterations", "inner_convergence_failures",
                                                "n_hess_psi_prod_calls", "spec_timeouts")] +
                       [(k, np.float64) for k in ("delta", "penalty_norm")] +
                       [(k, np.int32) for k in ("linesearch_failures", "linesearch_backtracks",
                                                "lbfgs_failures", "lbfgs_rejected", "tau_1_accepted",
                                                "count_tau")] +
                       [("sum_tau", np.float64), ("phase_s", np.float64, (4,))])
assert STATS_DTYPE.itemsize == C.sizeof(PantrStats)

# every symbol include/pantr_mi355x.h declares
EXPORTS = (
    "pantr_params_create", "pantr_params_destroy", "pantr_params_set", "pantr_params_get",
    "pantr_batch_create", "pantr_batch_destroy", "pantr_batch_dims", "pantr_batch_get_box_C",
    "pantr_batch_get_box_D", "pantr_batch_eval_psi_grad_psi_al", "pantr_batch_eval_hess_psi_prod_al",
    "pantr_batch_get_initial_guess", "pantr_batch_get_initial_state", "pantr_batch_set_param",
    "pantr_batch_get_param", "pantr_batch_solve", "pantr_batch_solve_device",
    "pantr_batch_last_kernel_ms", "pantr_batch_set_launch_order", "pantr_batch_last_launch_order", "pantr_batch_set_trace",
    "pantr_batch_get_trace", "pantr_batch_eval_psi", "pantr_batch_eval_psi_grad_psi",
    "pantr_batch_eval_hess_psi_prod", "pantr_batch_eval_prox_grad_step",
    "pantr_batch_simulate_dynamics", "pantr_batch_mpc", "pantr_last_error", "pantr_version", "pantr_device_count",
    "pantr_batch_footprint", "pantr_batch_footprint_for", "pantr_batch_forget_buffers", "pantr_batch_eval_grad_L",
    "pantr_batch_create_from_files",
    "pantr_batch_get_penalty_alm_split", "pantr_problem_data_load", "pantr_problem_data_destroy",
    "pantr_problem_data_get", "pantr_problem_dims", "pantr_problem_defaults", "pantr_batch_eval_g",
    "pantr_batch_mpc_record", "pantr_device_copy_bandwidth", "pantr_source_hash", "pantr_stats_size",
    "pantr_batch_kernel_name",
    "pantr_casadi_dll_open", "pantr_casadi_dll_close", "pantr_casadi_dll_function_info", "pantr_casadi_dll_call",
    "pantr_batch_check_casadi_dll",
)


class PantrError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[{code}] {message}")
        self.code = code


def build(force=False):
    """Compile lib/libpantr_mi355x.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", PKG_DIR] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    """Load the HIP extension.  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `make -C {PKG_DIR}` (or "
                              "__graft_entry__.build()); the PANTR solver has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        vp, cp = C.c_void_p, C.c_char_p
        pd = C.POINTER(_dbl)
        L.pantr_last_error.restype = cp
        L.pantr_version.restype = cp
        L.pantr_source_hash.restype = cp
        built, tree = L.pantr_source_hash().decode(), source_hash()
        if built != tree and not os.environ.get("PANTR_ALLOW_STALE_LIB"):
            raise ImportError(f"{LIB_PATH} was built from other sources than this tree's csrc/, host/ and "
                              f"include/ (library {built[:12]}..., tree {tree[:12]}...): run `make -C {PKG_DIR}`")
        L.pantr_stats_size.restype = C.c_size_t
        if L.pantr_stats_size() != STATS_DTYPE.itemsize:
            raise ImportError(f"{LIB_PATH}: pantr_stats_t is {L.pantr_stats_size()} bytes, this binding reads "
                              f"{STATS_DTYPE.itemsize}")
        L.pantr_params_create.argtypes = [C.POINTER(vp)]
        L.pantr_params_destroy.argtypes = [vp]
        L.pantr_params_destroy.restype = None
        L.pantr_params_set.argtypes = [vp, cp]
        L.pantr_params_get.argtypes = [vp, cp, pd]
        L.pantr_batch_create.argtypes = [cp, _i32, _i32, _i32, C.POINTER(vp)]
        L.pantr_batch_destroy.argtypes = [vp]
        L.pantr_batch_destroy.restype = None
        L.pantr_batch_dims.argtypes = [vp] + [C.POINTER(_i32)] * 6
        L.pantr_batch_get_box_C.argtypes = [vp, vp, vp]
        L.pantr_batch_get_box_D.argtypes = [vp, vp, vp]
        L.pantr_batch_eval_psi_grad_psi_al.argtypes = [vp, _i32, vp, vp, vp, vp, vp, vp, vp]
        L.pantr_batch_eval_hess_psi_prod_al.argtypes = [vp, _i32, vp, vp, vp, vp, vp, vp]
        L.pantr_batch_get_initial_guess.argtypes = [vp, vp]
        L.pantr_batch_get_initial_state.argtypes = [vp, vp]
        L.pantr_batch_set_param.argtypes = [vp, cp]
        L.pantr_batch_get_param.argtypes = [vp, cp, pd]
        L.pantr_batch_solve.argtypes = [vp, _i32, vp, vp, vp, vp]
        L.pantr_batch_solve_device.argtypes = [vp, _i32, vp, vp, vp, vp, vp]
        L.pantr_batch_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.pantr_batch_set_launch_order.argtypes = [vp, _i32, vp]
        L.pantr_batch_last_launch_order.argtypes = [vp, _i32, C.POINTER(_i32), vp]
        L.pantr_batch_set_trace.argtypes = [vp, _i32, _i32]
        L.pantr_batch_get_trace.argtypes = [vp, vp]
        L.pantr_batch_eval_psi.argtypes = [vp, _i32, vp, vp, vp]
        L.pantr_batch_eval_psi_grad_psi.argtypes = [vp, _i32, vp, vp, vp, vp]
        L.pantr_batch_eval_hess_psi_prod.argtypes = [vp, _i32, vp, vp, vp, vp]
        L.pantr_batch_eval_prox_grad_step.argtypes = [vp, _i32, _dbl, vp, vp, vp, vp, vp]
        L.pantr_batch_simulate_dynamics.argtypes = [vp, _i32, vp, vp, vp]
        L.pantr_batch_mpc.argtypes = [vp, _i32, _i32, _i32, _i32, vp, vp, vp, vp, vp]
        L.pantr_batch_footprint.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.pantr_batch_footprint_for.argtypes = [vp, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.pantr_batch_forget_buffers.argtypes = [vp]
        L.pantr_batch_eval_grad_L.argtypes = [vp, _i32, vp, vp, vp, vp]
        L.pantr_batch_eval_g.argtypes = [vp, _i32, vp, vp, vp]
        L.pantr_batch_mpc_record.argtypes = [vp, _i32, _i32, _i32, _i32, vp, vp, vp, vp, vp, vp, vp, vp]
        L.pantr_device_copy_bandwidth.argtypes = [_i32, C.c_int64, _i32, pd]
        L.pantr_batch_create_from_files.argtypes = [cp, cp, _i32, _i32, _i32, C.POINTER(vp)]
        L.pantr_batch_get_penalty_alm_split.argtypes = [vp, C.POINTER(_i32)]
        L.pantr_problem_data_load.argtypes = [cp, cp, _i32, C.POINTER(vp)]
        L.pantr_problem_data_destroy.argtypes = [vp]
        L.pantr_problem_data_destroy.restype = None
        L.pantr_problem_data_get.argtypes = [vp, C.POINTER(_i32), vp, vp, vp, vp, vp, vp, C.POINTER(_i32)]
        L.pantr_problem_dims.argtypes = [cp, _i32] + [C.POINTER(_i32)] * 6
        L.pantr_problem_defaults.argtypes = [cp, _i32, vp, vp, vp, vp, vp, vp]
        L.pantr_casadi_dll_open.argtypes = [cp, C.POINTER(vp)]
        L.pantr_casadi_dll_close.argtypes = [vp]
        L.pantr_casadi_dll_close.restype = None
        L.pantr_casadi_dll_function_info.argtypes = [vp, cp, C.POINTER(_i32), C.POINTER(_i32), vp, vp, _i32, vp]
        L.pantr_casadi_dll_call.argtypes = [vp, cp, _i32, vp, _i32, vp]
        L.pantr_batch_check_casadi_dll.argtypes = [vp, cp, _i32, vp]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise PantrError(rc, lib().pantr_last_error().decode())


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


class PantrParams:
    """Solver option set without a device (pantr_params_t)."""

    def __init__(self, opts=()):
        self._h = C.c_void_p()
        _check(lib().pantr_params_create(C.byref(self._h)))
        for o in opts:
            self.set(o)

    def set(self, kv):
        _check(lib().pantr_params_set(self._h, kv.encode()))

    def get(self, key):
        v = _dbl()
        _check(lib().pantr_params_get(self._h, key.encode(), C.byref(v)))
        return v.value

    def __del__(self):
        if getattr(self, "_h", None):
            lib().pantr_params_destroy(self._h)
            self._h = None


def problem_dims(problem, horizon):
    """(n, m, nx, nu, nc, nc_N) of a compiled-in problem; no device needed."""
    d = [_i32() for _ in range(6)]
    _check(lib().pantr_problem_dims(problem.encode(), horizon, *[C.byref(x) for x in d]))
    return tuple(x.value for x in d)


def problem_defaults(problem, horizon):
    """Compiled-in numeric data of a problem (what problems/generate_problem.py:62-73 would write into
    its sidecar files): dict with C_lb, C_ub (n), D_lb, D_ub (m), param (nx), guess (n)."""
    n, m, nx = problem_dims(problem, horizon)[:3]
    out = dict(C_lb=np.zeros(n), C_ub=np.zeros(n), D_lb=np.zeros(m), D_ub=np.zeros(m),
               param=np.zeros(nx), guess=np.zeros(n))
    _check(lib().pantr_problem_defaults(problem.encode(), horizon, *[_ptr(out[k]) for k in
                                        ("C_lb", "C_ub", "D_lb", "D_ub", "param", "guess")]))
    return out


def write_problem_files(directory, problem, formulation, horizon, data=None):
    """Write "<dir>/<problem>_<formulation>_<N>.csv" and ".guess.csv" in the reference's format:
    the seven rows of alpaqa.casadi_generator.write_casadi_problem_data (called at
    problems/generate_problem.py:62-69) and the two rows of generate_problem.py:71-73, each row one
    numpy.savetxt call with delimiter "," as there.  `data` defaults to the compiled-in values."""
    d = dict(problem_defaults(problem, horizon))
    d.update(data or {})
    n, m = problem_dims(problem, horizon)[:2]
    stem = os.path.join(directory, f"{problem}_{formulation}_{horizon}")
    opt = dict(delimiter=",", newline="\n")
    row = lambda x: np.reshape(np.asarray(x, dtype=np.float64), (1, -1))
    with open(stem + ".csv", "w") as f:
        for key in ("C_lb", "C_ub", "D_lb", "D_ub", "param"):
            np.savetxt(f, row(d[key]), **opt)
        np.savetxt(f, row(d.get("l1_reg", [])), **opt)
        np.savetxt(f, row([d.get("penalty_alm_split_csv", 0)]), **opt)  # generate_problem.py:68
    with open(stem + ".guess.csv", "w") as f:
        np.savetxt(f, [np.asarray(d["guess"], dtype=np.float64)], **opt)
        np.savetxt(f, [d.get("penalty_alm_split", m)], **opt)          # generate_problem.py:32,73
    return stem


def load_problem_files(path, formulation, horizon):
    """Read and check the sidecar files of "[<dir>/]<name>" (no device needed).  Returns a dict with
    has_csv, C_lb, C_ub, D_lb, D_ub, param, guess, penalty_alm_split; raises PantrError as the C ABI
    documents (missing guess file, malformed rows, dimensions that disagree with the problem)."""
    h = C.c_void_p()
    _check(lib().pantr_problem_data_load(path.encode(), formulation.encode(), horizon, C.byref(h)))
    try:
        n, m, nx = problem_dims(os.path.basename(path), horizon)[:3]
        out = dict(C_lb=np.zeros(n), C_ub=np.zeros(n), D_lb=np.zeros(m), D_ub=np.zeros(m),
                   param=np.zeros(nx), guess=np.zeros(n))
        has_csv, split = _i32(), _i32()
        _check(lib().pantr_problem_data_get(h, C.byref(has_csv), *[_ptr(out[k]) for k in
                                            ("C_lb", "C_ub", "D_lb", "D_ub", "param", "guess")],
                                            C.byref(split)))
        out["has_csv"], out["penalty_alm_split"] = bool(has_csv.value), split.value
        return out
    finally:
        lib().pantr_problem_data_destroy(h)


DLL_CHECK_FIELDS = ("psi", "grad_psi", "hess_psi_prod", "g", "yhat")


class CasadiDll:
    """A CasADi-generated problem library "<dir>/<name>_<formulation>_<N>.so" (what
    drivers/casadi-dll-wrapper.cpp:25-67 loads through alpaqa::CasADiProblem), opened on the host through
    the generated code's own C convention (host/casadi_dll.hpp).  A cross-checker of the device functions
    (PantrBatch.check_casadi_dll), never part of a solve.  No device needed."""

    def __init__(self, so_path):
        self._h = C.c_void_p()
        _check(lib().pantr_casadi_dll_open(os.fspath(so_path).encode(), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            lib().pantr_casadi_dll_close(self._h)
            self._h = None

    __del__ = close

    def function_info(self, name, cap=16):
        """dict(n_in, n_out, nnz_in, nnz_out, work=(sz_arg, sz_res, sz_iw, sz_w)) of a generated function."""
        n_in, n_out = _i32(), _i32()
        nnz_in, nnz_out, work = np.full(cap, -2, np.int64), np.full(cap, -2, np.int64), np.zeros(4, np.int64)
        _check(lib().pantr_casadi_dll_function_info(self._h, name.encode(), C.byref(n_in), C.byref(n_out),
                                                    _ptr(nnz_in), _ptr(nnz_out), cap, _ptr(work)))
        return dict(n_in=n_in.value, n_out=n_out.value, nnz_in=nnz_in[:n_in.value].tolist(),
                    nnz_out=nnz_out[:n_out.value].tolist(), work=tuple(int(w) for w in work))

    def call(self, name, args, out_sizes):
        """Evaluate `name` at `args` (arrays or scalars; None = zeros); returns the outputs as arrays."""
        ins = [None if a is None else np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64) for a in args]
        outs = [np.zeros(max(int(k), 1)) for k in out_sizes]
        pa = (C.c_void_p * max(len(ins), 1))(*[None if a is None or a.size == 0 else a.ctypes.data for a in ins])
        pr = (C.c_void_p * max(len(outs), 1))(*[o.ctypes.data for o in outs])
        _check(lib().pantr_casadi_dll_call(self._h, name.encode(), len(ins), pa, len(outs), pr))
        return [o[:int(k)] for o, k in zip(outs, out_sizes)]


class PantrBatch:
    """A problem (name, horizon) with device resources for up to max_instances solves."""

    def __init__(self, problem="hanging_chain", horizon=60, max_instances=1, device=0, opts=(),
                 files=None):
        """files=(path, formulation): load_benchmark_casadi_problem's data half - `path` =
        "[<dir>/]<name>" names the problem and its sidecar files, whose bounds, initial state and
        initial guess replace the compiled-in values (`problem` is ignored)."""
        self._h = C.c_void_p()
        if files is not None:
            path, formulation = files
            _check(lib().pantr_batch_create_from_files(path.encode(), formulation.encode(), horizon,
                                                       max_instances, device, C.byref(self._h)))
        else:
            _check(lib().pantr_batch_create(problem.encode(), horizon, max_instances, device,
                                            C.byref(self._h)))
        d = [_i32() for _ in range(6)]
        _check(lib().pantr_batch_dims(self._h, *[C.byref(x) for x in d]))
        self.n, self.m, self.nx, self.nu, self.nc, self.nc_N = [x.value for x in d]
        self.horizon = horizon
        self.max_instances = max_instances
        self.set_params(opts)

    def close(self):
        if getattr(self, "_h", None):
            lib().pantr_batch_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- options -------------------------------------------------------------------------
    def set_params(self, opts):
        for o in opts:
            _check(lib().pantr_batch_set_param(self._h, o.encode()))

    def get_param(self, key):
        v = _dbl()
        _check(lib().pantr_batch_get_param(self._h, key.encode(), C.byref(v)))
        return v.value

    def check_casadi_dll(self, so_path, samples=8):
        """Compare the generated library's psi / psi_grad_psi / hess_psi_prod / g with this handle's device
        functions on `samples` seeded points (pantr_batch_check_casadi_dll): dict quantity -> max relative
        error (None where the library or the problem has no such quantity).  The bar is 1e-10."""
        err = np.zeros(len(DLL_CHECK_FIELDS))
        _check(lib().pantr_batch_check_casadi_dll(self._h, os.fspath(so_path).encode(), samples, _ptr(err)))
        return {k: (None if e == -1.0 else float(e)) for k, e in zip(DLL_CHECK_FIELDS, err)}

    # ---- problem data --------------------------------------------------------------------
    def box_C(self):
        lb, ub = np.zeros(self.n), np.zeros(self.n)
        _check(lib().pantr_batch_get_box_C(self._h, _ptr(lb), _ptr(ub)))
        return lb, ub

    def box_D(self):
        lb, ub = np.zeros(self.m), np.zeros(self.m)
        _check(lib().pantr_batch_get_box_D(self._h, _ptr(lb), _ptr(ub)))
        return lb, ub

    def initial_guess(self):
        u = np.zeros(self.n)
        _check(lib().pantr_batch_get_initial_guess(self._h, _ptr(u)))
        return u

    def initial_state(self):
        x = np.zeros(self.nx)
        _check(lib().pantr_batch_get_initial_state(self._h, _ptr(x)))
        return x

    def penalty_alm_split(self):
        v = _i32()
        _check(lib().pantr_batch_get_penalty_alm_split(self._h, C.byref(v)))
        return v.value

    def kernel_name(self, B=None):
        """Device kernel a solve of B instances (default: the handle's maximum) launches, as rocprofv3 names it."""
        f = lib().pantr_batch_kernel_name
        f.restype, f.argtypes = C.c_char_p, [C.c_void_p, _i32]
        return f(self._h, self.max_instances if B is None else int(B)).decode()

    def footprint(self, B=None):
        """(LDS bytes per workgroup, HBM workspace bytes per instance) of the kernel a solve of B instances launches
        (default: the handle's maximum batch size; kernel_name(B) names that kernel)."""
        a, b = C.c_int64(), C.c_int64()
        if B is None:
            _check(lib().pantr_batch_footprint(self._h, C.byref(a), C.byref(b)))
        else:
            _check(lib().pantr_batch_footprint_for(self._h, B, C.byref(a), C.byref(b)))
        return a.value, b.value

    def forget_buffers(self):
        """After freeing or re-allocating buffers passed to solve_device: have the next call check them again."""
        _check(lib().pantr_batch_forget_buffers(self._h))

    # ---- hot path ------------------------------------------------------------------------
    def solve(self, x0, u0):
        """Solve B instances.  x0: (B, nx), u0: (B, n).  Returns (u, stats recarray)."""
        x0 = _f64(x0)
        B = x0.shape[0]
        x0 = _f64(x0, (B, self.nx))
        u = np.array(u0, dtype=np.float64, order="C", copy=True)
        if u.shape != (B, self.n):
            raise ValueError(f"expected u0 shape {(B, self.n)}, got {u.shape}")
        stats = np.zeros(B, dtype=STATS_DTYPE)
        if self.m > 0:
            raise ValueError("this problem has general constraints: use solve_al(x0, u0, y0)")
        _check(lib().pantr_batch_solve(self._h, B, _ptr(x0), _ptr(u), None, _ptr(stats)))
        return u, stats

    def solve_al(self, x0, u0, y0=None):
        """ALM solve of B instances of a problem with general constraints.  y0: (B, m) initial
        multipliers (zeros if None).  Returns (u, y, stats)."""
        x0 = _f64(x0)
        B = x0.shape[0]
        x0 = _f64(x0, (B, self.nx))
        u = np.array(u0, dtype=np.float64, order="C", copy=True)
        y = np.zeros((B, self.m)) if y0 is None else np.array(y0, dtype=np.float64, order="C", copy=True)
        if u.shape != (B, self.n) or y.shape != (B, self.m):
            raise ValueError("u0 must be (B, n) and y0 (B, m)")
        stats = np.zeros(B, dtype=STATS_DTYPE)
        _check(lib().pantr_batch_solve(self._h, B, _ptr(x0), _ptr(u), _ptr(y), _ptr(stats)))
        return u, y, stats

    def solve_device(self, B, d_x0, d_u, d_stats, stream=None, d_y=None):
        """Device-pointer solve (ints = device addresses), asynchronous on `stream`."""
        _check(lib().pantr_batch_solve_device(self._h, B, C.c_void_p(d_x0), C.c_void_p(d_u),
                                              C.c_void_p(d_y or 0), C.c_void_p(d_stats),
                                              C.c_void_p(stream or 0)))

    def last_kernel_ms(self):
        ms = C.c_float()
        _check(lib().pantr_batch_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def set_launch_order(self, order):
        if order is None:
            _check(lib().pantr_batch_set_launch_order(self._h, 0, None))
            return
        order = np.ascontiguousarray(order, dtype=np.int32)
        _check(lib().pantr_batch_set_launch_order(self._h, len(order), _ptr(order)))

    def last_launch_order(self, B=None):
        """(source, next_order): what the most recent launch's workgroup order came from ("index", "user",
        "previous solve") and, with B, the ranking that solve left for the next launch of B instances (None if none)."""
        src = _i32()
        _check(lib().pantr_batch_last_launch_order(self._h, 0, C.byref(src), None))
        nxt = None
        if B is not None:
            nxt = np.zeros(B, dtype=np.int32)
            if lib().pantr_batch_last_launch_order(self._h, B, None, _ptr(nxt)) != 0:
                nxt = None
        return ("index", "user", "previous solve")[src.value], nxt

    def set_trace(self, instances, cap):
        _check(lib().pantr_batch_set_trace(self._h, instances, cap))
        self._trace_shape = (instances, cap, len(TRACE_COLS))

    def get_trace(self):
        out = np.zeros(self._trace_shape)
        _check(lib().pantr_batch_get_trace(self._h, _ptr(out)))
        return out

    # ---- problem functions ---------------------------------------------------------------
    def _rows(self, x0, *vecs, width=None):
        """x0: (B, width); every further array (B, n) (or, as a (array, width) pair, (B, width)).
        Returns B and the validated contiguous fp64 arrays: the C side trusts these shapes."""
        x0 = _f64(x0)
        if x0.ndim != 2:
            raise ValueError(f"expected a 2-D batch (B, {width or self.nx}), got shape {x0.shape}")
        B = x0.shape[0]
        out = [_f64(x0, (B, self.nx if width is None else width))]
        for v in vecs:
            v, w = v if isinstance(v, tuple) else (v, self.n)
            out.append(_f64(v, (B, w)))
        return [B] + out

    def eval_psi(self, x0, u):
        B, x0, u = self._rows(x0, u)
        psi = np.zeros(B)
        _check(lib().pantr_batch_eval_psi(self._h, B, _ptr(x0), _ptr(u), _ptr(psi)))
        return psi

    def eval_psi_grad_psi(self, x0, u):
        B, x0, u = self._rows(x0, u)
        psi, grad = np.zeros(B), np.zeros((B, self.n))
        _check(lib().pantr_batch_eval_psi_grad_psi(self._h, B, _ptr(x0), _ptr(u), _ptr(psi),
                                                   _ptr(grad)))
        return psi, grad

    def eval_hess_psi_prod(self, x0, u, v, prefetch_depth=None):
        """Hv: records the second-order tape at u, then replays it for v (the solver's path).  prefetch_depth (tests,
        hanging chain): 1 | 2 stages ahead - the two forms of the product the solve kernels use."""
        if prefetch_depth is not None:
            lib().pantr_debug_set_eval_prefetch_depth(self._h, int(prefetch_depth))
        B, x0, u, v = self._rows(x0, u, v)
        Hv = np.zeros((B, self.n))
        _check(lib().pantr_batch_eval_hess_psi_prod(self._h, B, _ptr(x0), _ptr(u), _ptr(v), _ptr(Hv)))
        return Hv

    def eval_psi_grad_psi_al(self, x0, u, y, Sigma):
        """psi, grad psi and yhat with multipliers y and penalties Sigma (formulation.hpp:145-161)."""
        B, x0, u, y, Sigma = self._rows(x0, u, (y, self.m), (Sigma, self.m))
        psi, grad, yhat = np.zeros(B), np.zeros((B, self.n)), np.zeros((B, self.m))
        _check(lib().pantr_batch_eval_psi_grad_psi_al(self._h, B, _ptr(x0), _ptr(u), _ptr(y), _ptr(Sigma),
                                                      _ptr(psi), _ptr(grad), _ptr(yhat)))
        return psi, grad, yhat

    def eval_grad_L(self, x0, u, y):
        """grad_x L(x, y) = grad f + grad g' y (formulation.hpp:170-176), m > 0 problems."""
        B, x0, u, y = self._rows(x0, u, (y, self.m))
        grad = np.zeros((B, self.n))
        _check(lib().pantr_batch_eval_grad_L(self._h, B, _ptr(x0), _ptr(u), _ptr(y), _ptr(grad)))
        return grad

    def eval_g(self, x0, u):
        """g(u): the stage constraints along the trajectory, then the terminal ones (ss.py:39-40)."""
        B, x0, u = self._rows(x0, u)
        g = np.zeros((B, self.m))
        _check(lib().pantr_batch_eval_g(self._h, B, _ptr(x0), _ptr(u), _ptr(g)))
        return g

    def eval_hess_psi_prod_al(self, x0, u, y, Sigma, v):
        B, x0, u, y, Sigma, v = self._rows(x0, u, (y, self.m), (Sigma, self.m), v)
        Hv = np.zeros((B, self.n))
        _check(lib().pantr_batch_eval_hess_psi_prod_al(self._h, B, _ptr(x0), _ptr(u), _ptr(y), _ptr(Sigma),
                                                       _ptr(v), _ptr(Hv)))
        return Hv

    def eval_prox_grad_step(self, gamma, x, grad):
        B, x, grad = self._rows(x, grad, width=self.n)
        xh, p, s = np.zeros((B, self.n)), np.zeros((B, self.n)), np.zeros((B, 4))
        _check(lib().pantr_batch_eval_prox_grad_step(self._h, B, gamma, _ptr(x), _ptr(grad),
                                                     _ptr(xh), _ptr(p), _ptr(s)))
        return xh, p, s

    def simulate_dynamics(self, x, u):
        B, x, u = self._rows(x, (u, self.nu))
        xn = np.zeros((B, self.nx))
        _check(lib().pantr_batch_simulate_dynamics(self._h, B, _ptr(x), _ptr(u), _ptr(xn)))
        return xn

    def mpc(self, num_sim, warm, u_dist, num_dist=5, x_init=None, record=False):
        """B closed loops in lock-step on the device (do_mpc_experiment, mpc-driver.cpp:123-210).

        u_dist: (B, nu) disturbance per trajectory.  Returns states (num_sim+1, B, nx),
        inputs (num_sim, B, n), stats (num_sim, B); with record=True also a dict with the guesses
        every step was solved from (guess_u, guess_y) and the multipliers it returned."""
        u_dist = _f64(u_dist)
        if u_dist.ndim != 2 or u_dist.shape[1] != self.nu:
            raise ValueError(f"expected u_dist shape (B, {self.nu}), got {u_dist.shape}")
        B = u_dist.shape[0]
        states = np.zeros((num_sim + 1, B, self.nx))
        inputs = np.zeros((num_sim, B, self.n))
        stats = np.zeros((num_sim, B), dtype=STATS_DTYPE)
        xi = None if x_init is None else _f64(x_init, (B, self.nx))
        if not record:
            _check(lib().pantr_batch_mpc(self._h, B, num_sim, int(bool(warm)), num_dist,
                                         None if xi is None else _ptr(xi), _ptr(u_dist), _ptr(states),
                                         _ptr(inputs), _ptr(stats)))
            return states, inputs, stats
        rec = dict(guess_u=np.zeros((num_sim, B, self.n)), guess_y=np.zeros((num_sim, B, self.m)),
                   multipliers=np.zeros((num_sim, B, self.m)))
        _check(lib().pantr_batch_mpc_record(self._h, B, num_sim, int(bool(warm)), num_dist,
                                            None if xi is None else _ptr(xi), _ptr(u_dist), _ptr(states),
                                            _ptr(inputs), _ptr(stats), _ptr(rec["guess_u"]),
                                            _ptr(rec["guess_y"]), _ptr(rec["multipliers"])))
        return states, inputs, stats, rec


def device_count():
    return lib().pantr_device_count()


def version():
    return lib().pantr_version().decode()


def source_hash():
    """sha256 over everything the library is compiled from (csrc/*.h and csrc/*.hip in name order, then
    host/*.hpp in name order, then include/pantr_mi355x.h; the Makefile compiles the same digest into
    the library): ties a library, and a committed profile summary, to the code they were made from."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(PKG_DIR, "csrc")
    files = [os.path.join(d, n) for n in sorted(n for n in os.listdir(d) if n.endswith((".h", ".hip")))]
    d = os.path.join(PKG_DIR, "host")
    files += [os.path.join(d, n) for n in sorted(n for n in os.listdir(d) if n.endswith(".hpp"))]
    files.append(os.path.join(os.path.dirname(PKG_DIR), "include", "pantr_mi355x.h"))
    for path in files:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def copy_bandwidth(device=0, nbytes=2 << 30, reps=5):
    """Achievable HBM bandwidth (GB/s, read + written) from the library's 16 B/lane device copy."""
    v = _dbl()
    _check(lib().pantr_device_copy_bandwidth(device, nbytes, reps, C.byref(v)))
    return v.value


def is_product_build():
    """False for a library built with a switch that changes what it reports (pantr_version())."""
    return "+" not in version()


def algorithmic_bytes(stats, N, nx, nu, m=0):
    """trajectory-streaming byte model, summed over the records.

    B_alg = 8 [ n_psi (n + nx + N nx + 3m) + n_grad (2n + nx + 2 N nx + 4m)
              + n_Hv (3n + nx + 4 N nx + 4m) + n_cg 17 n + n_it 15 n ]
    with n_grad = #psi_grad_psi + #grad_psi + #grad_L.
    """
    n = nu * N
    s = stats
    n_psi = s["n_psi"].astype(np.int64).sum()
    n_grad = (s["n_psi_grad_psi"].astype(np.int64) + s["n_grad_psi"] + s["n_grad_L"]).sum()
    n_hv = s["n_hess_psi_prod"].astype(np.int64).sum()
    n_cg = s["cg_iterations"].astype(np.int64).sum()
    n_it = s["iterations"].astype(np.int64).sum()
    return 8 * (n_psi * (n + nx + N * nx + 3 * m) + n_grad * (2 * n + nx + 2 * N * nx + 4 * m) +
                n_hv * (3 * n + nx + 4 * N * nx + 4 * m) + n_cg * 17 * n + n_it * 15 * n)
