/* pantr_mi355x.h — C ABI of the MI355X-native batched PANTR solver.
 *
 * This is the drop-in boundary for the PANTR hot path of
 * kul-optec/pantr-cdc2023-experiments.  Every entry point names the reference
 * interface it replaces (file:line relative to the reference tree).  Plain
 * pointers and sizes only; no exceptions cross this boundary: every function
 * returns PANTR_OK or a negative error code and pantr_last_error() gives the
 * message (the reference throws C++ exceptions instead, mpc-driver.cpp:441-445).
 *
 * Layout conventions: fp64 everywhere (alpaqa_real_t = double,
 * problems/problems/formulation.hpp:12); batches are instance-major contiguous
 * rows: x0[B][nx], u[B][n], stats[B].  One handle per host thread.
 */
#ifndef PANTR_MI355X_H
#define PANTR_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PANTR_OK 0
#define PANTR_ERR_INVALID_ARGUMENT (-1) /* std::invalid_argument in the reference */
#define PANTR_ERR_OUT_OF_RANGE (-2)     /* std::out_of_range */
#define PANTR_ERR_NO_DEVICE (-3)        /* no usable gfx950 device / HIP failure */
#define PANTR_ERR_RUNTIME (-4)          /* std::runtime_error */

/* alpaqa::SolverStatus as printed by mpc-driver.cpp:218 and tested at :219 */
enum pantr_status {
    PANTR_STATUS_BUSY = 0,
    PANTR_STATUS_CONVERGED = 1,
    PANTR_STATUS_MAX_TIME = 2,
    PANTR_STATUS_MAX_ITER = 3,
    PANTR_STATUS_NOT_FINITE = 4,
    PANTR_STATUS_NO_PROGRESS = 5,
    PANTR_STATUS_INTERRUPTED = 6,
    PANTR_STATUS_EXCEPTION = 7
};

/* Per-instance result: the fields of alpaqa's PANTRStats/ALMStats the reference reads
 * (stats.status, stats.elapsed_time: mpc-driver.cpp:217-221) plus the evaluation counters
 * the reference logs through alpaqa::EvalCounter (drivers/results.hpp:63-87). */
typedef struct pantr_stats {
    double eps;          /* final stop-criterion value */
    double final_gamma;  /* final step size */
    double final_psi;    /* psi(xhat) at exit */
    double final_fbe;    /* forward-backward envelope at exit */
    double final_radius; /* final trust radius */
    double elapsed_s;    /* in-kernel solve time of this instance (constant-rate device clock) */
    int32_t status;      /* enum pantr_status */
    int32_t iterations;  /* PANTR iterations */
    int32_t cg_iterations;
    int32_t accepted, rejected, qub_backtracks;
    int32_t n_psi, n_grad_psi, n_psi_grad_psi, n_grad_L, n_hess_psi_prod, n_prox_grad_step;
    int32_t outer_iterations;           /* ALM iterations (1 when m = 0) */
    int32_t inner_convergence_failures; /* inner solves that did not reach their tolerance */
    /* n_hess_psi_prod counts the Hessian-vector products EVALUATED (what the byte model of the roofline
     * uses).  n_hess_psi_prod_calls is what alpaqa's EvalCounter would show (drivers/results.hpp:63-87,
     * summed by new-benchmarks-paper/benchmark_util.py:12-27): it also counts the products with q_K = 0
     * that the direction skips because their result is exactly zero. */
    int32_t n_hess_psi_prod_calls;
    /* two-wavefront kernels (speculation): requests whose result the solver wavefront stopped waiting for
     * (it evaluates the step itself and posts no further requests); 0 in a healthy run */
    int32_t spec_timeouts;
    double delta;                       /* ||Sigma^{-1}(yhat - y)||_inf at exit (0 when m = 0) */
    double penalty_norm;                /* ||Sigma||_2 / sqrt(m) at exit */
    /* method=panoc, strucpanoc (alpaqa::PANOCStats; qub_backtracks above = stepsize_backtracks) */
    int32_t linesearch_failures, linesearch_backtracks, lbfgs_failures, lbfgs_rejected;
    int32_t tau_1_accepted, count_tau;
    double sum_tau;
    /* seconds per phase, filled only by the profiling build of the library (`make phase`,
     * pantr_version() then ends in "+phase-timing"): direction (hess_psi_prod | L-BFGS apply),
     * psi_grad_psi at the candidate, psi(xhat), grad_L(xhat) + tape (| L-BFGS update).  Zero in
     * the product build. */
    double phase_s[4];
} pantr_stats_t;

typedef struct pantr_params pantr_params_t; /* solver option set */
typedef struct pantr_batch pantr_batch_t;   /* problem + device resources */
typedef struct pantr_problem_data pantr_problem_data_t; /* numeric data read from a problem's sidecar files */

/* ---- options: replaces alpaqa::params::set_params(obj, prefix, opts) ---------------------
 * (call sites mpc-driver.cpp:54,61,62,113) and the solver selection "method=<name>"
 * (mpc-driver.cpp:360-384: pantr, panoc, strucpanoc).  `kv` is one "prefix.key=value" string
 * with prefix solver. (PANTRParams / PANOCParams), dir. (NewtonTRDirectionParams /
 * [Structured]LBFGSDirectionParams), accel. (SteihaugCGParams / LBFGSParams) or alm.
 * (ALMParams).  Strings with another prefix or without a '.' are ignored, as
 * set_params ignores options addressed to other objects; an unknown key under a known
 * prefix or an unparsable value is PANTR_ERR_INVALID_ARGUMENT.  Durations accept "30s",
 * "500ms", "2min"; enums their alpaqa names ("ProjGradUnitNorm").
 * pantr_params_create applies the reference driver's defaults (mpc-driver.cpp:51-53,
 * 109-111) on top of alpaqa's.
 * Options of this implementation, spelled like "method=" (no prefix; the reference has no counterpart, results
 * are the same whatever their value): "speculation=auto|on|off" (method=pantr: a second wavefront per instance
 * evaluates ahead; auto = on for batches that leave the SIMDs idle), "spec_patience=<duration>" (how long a solver
 * wavefront waits for that helper before it goes on alone, default 1s; pantr_stats_t::spec_timeouts),
 * "residency=auto|<k>" (at most k workgroups per compute unit: a measurement aid), "launch_order=natural|previous|
 * auto" (pantr_batch_last_launch_order below). */
int pantr_params_create(pantr_params_t **out);
void pantr_params_destroy(pantr_params_t *p);
int pantr_params_set(pantr_params_t *p, const char *kv);
int pantr_params_get(const pantr_params_t *p, const char *key, double *value);

/* ---- problem: replaces load_benchmark_problem / load_benchmark_dl_problem ---------------
 * (drivers/casadi-dll-wrapper.cpp:25-67,69-129) for the problems this library carries as
 * compiled device code.  `problem` is the reference's problem name ("hanging_chain",
 * "quadcopter", "realistic_quadcopter"; a "dl:" prefix is accepted and dropped),
 * `horizon` the N of "<name>_<formulation>_<N>.so" (casadi-dll-wrapper.cpp:29-31).
 * `max_instances` sizes the device workspace; `device` is the HIP device ordinal. */
int pantr_batch_create(const char *problem, int32_t horizon, int32_t max_instances,
                       int32_t device, pantr_batch_t **out);
void pantr_batch_destroy(pantr_batch_t *b);

/* get_n/get_m (mpc-driver.cpp:131), get_nx/get_nu/get_nc/get_nc_N (casadi-dll-wrapper.cpp:109-113) */
int pantr_batch_dims(const pantr_batch_t *b, int32_t *n, int32_t *m, int32_t *nx, int32_t *nu,
                     int32_t *nc, int32_t *nc_N);
/* get_box_C() (mpc-driver.cpp:140; formulation.hpp:220-227): lb[n], ub[n] */
int pantr_batch_get_box_C(const pantr_batch_t *b, double *lb, double *ub);
/* get_box_D() (formulation.hpp:229-238): lb[m], ub[m] (nothing is written when m = 0) */
int pantr_batch_get_box_D(const pantr_batch_t *b, double *lb, double *ub);
/* "initial_guess" (casadi-dll-wrapper.cpp:100-106; hanging_chain.cpp:135): u[n] */
int pantr_batch_get_initial_guess(const pantr_batch_t *b, double *u);
/* "get_initial_state" (hanging_chain.cpp:154-155,121-124): x0[nx] */
int pantr_batch_get_initial_state(const pantr_batch_t *b, double *x0);

/* ---- numeric problem data: the data half of load_benchmark_casadi_problem ------------------
 * (drivers/casadi-dll-wrapper.cpp:25-67).  A generated problem "<dir>/<name>_<formulation>_<N>.so"
 * comes with "<same>.csv" (box C, box D, param = initial state, l1_reg, penalty_alm_split; written
 * through alpaqa.casadi_generator.write_casadi_problem_data, problems/generate_problem.py:62-69)
 * and "<same>.guess.csv" (row 1 initial guess, row 2 penalty_alm_split; generate_problem.py:71-73,
 * read at casadi-dll-wrapper.cpp:57-61).  The .so itself is host code and cannot run on the device:
 * the problem FUNCTIONS are this library's compiled device code for <name>; the DATA of the
 * sidecars replace the compiled-in values (bounds, initial state, initial guess).
 * `path` = "[<dir>/]<name>", `formulation` = "ss" | "ss2" | "ss2p" (casadi-dll-wrapper.cpp:29-32).
 * Errors: missing guess file -> PANTR_ERR_RUNTIME ("Failed to open guess file", :58-59); unparsable
 * file -> PANTR_ERR_RUNTIME; sizes that disagree with the compiled-in problem's n / m / nx, a box
 * that is not the stage box tiled over the horizon (formulation.hpp:220-238), an empty box, or
 * l1_reg != 0 -> PANTR_ERR_INVALID_ARGUMENT; unknown <name> -> PANTR_ERR_OUT_OF_RANGE.
 * pantr_problem_data_load needs no device (it reads and checks); rows a file leaves empty (and
 * everything, when only the guess file exists) fall back to the compiled-in values. */
int pantr_problem_data_load(const char *path, const char *formulation, int32_t horizon,
                            pantr_problem_data_t **out);
void pantr_problem_data_destroy(pantr_problem_data_t *d);
/* C_lb/C_ub/guess[n], D_lb/D_ub[m], param[nx]; any output pointer may be NULL */
int pantr_problem_data_get(const pantr_problem_data_t *d, int32_t *has_csv, double *C_lb, double *C_ub,
                           double *D_lb, double *D_ub, double *param, double *guess,
                           int32_t *penalty_alm_split);
/* pantr_batch_create for the problem named by `path`, with the sidecar data applied. */
int pantr_batch_create_from_files(const char *path, const char *formulation, int32_t horizon,
                                  int32_t max_instances, int32_t device, pantr_batch_t **out);
/* BenchmarkProblem::penalty_alm_split (casadi-dll-wrapper.hpp:30): read from the guess file, stored,
 * and - as in the reference driver - not used by the solve (0 unless created from files). */
int pantr_batch_get_penalty_alm_split(const pantr_batch_t *b, int32_t *split);
/* Dimensions and compiled-in data of a problem without a device (what generate_problem.py would
 * write for it): C_lb/C_ub/u_guess[n], D_lb/D_ub[m], x_init[nx]; any output pointer may be NULL. */
int pantr_problem_dims(const char *problem, int32_t horizon, int32_t *n, int32_t *m, int32_t *nx,
                       int32_t *nu, int32_t *nc, int32_t *nc_N);
int pantr_problem_defaults(const char *problem, int32_t horizon, double *C_lb, double *C_ub, double *D_lb,
                           double *D_ub, double *x_init, double *u_guess);

/* options of this handle's solver (same grammar as pantr_params_set) */
int pantr_batch_set_param(pantr_batch_t *b, const char *kv);
int pantr_batch_get_param(const pantr_batch_t *b, const char *key, double *value);

/* ---- the hot path: replaces stats = solver(problem, u, y) -------------------------------
 * (mpc-driver.cpp:217: ALMSolver<PANTRSolver<NewtonTRDirection>>::operator()) for B
 * independent instances.  x0[B][nx] is the problem parameter each instance is solved for
 * (the reference writes it through initial_state(), mpc-driver.cpp:126); u[B][n] holds the
 * initial guesses on entry and the solutions on exit; y[B][m] likewise (m = 0 for
 * hanging_chain: pass NULL).  Host pointers; copies in, one launch, copies out. */
int pantr_batch_solve(pantr_batch_t *b, int32_t B, const double *x0, double *u, double *y,
                      pantr_stats_t *stats);
/* Same with device pointers (inputs already resident in HBM), asynchronous on `stream`
 * (a hipStream_t, NULL = default stream).  d_y[B][m] may be NULL when m = 0; d_stats is device
 * memory, B records.  Every buffer must be device memory of this handle's GPU known to the HIP
 * runtime this library is bound to; otherwise PANTR_ERR_NO_DEVICE (this is what a process gets
 * that loaded the library before PyTorch's HIP runtime and then passes torch pointers). */
int pantr_batch_solve_device(pantr_batch_t *b, int32_t B, const double *d_x0, double *d_u, double *d_y,
                             pantr_stats_t *d_stats, void *stream);
/* Duration of the most recent solve kernel, measured with HIP events recorded on the stream
 * it was launched on (synchronises on the stop event). */
int pantr_batch_last_kernel_ms(pantr_batch_t *b, float *ms);
/* Launch order: order[B] lists instance indices in the order workgroups should start them
 * (longest-expected-first); NULL restores identity.  Used by the next solve calls. */
int pantr_batch_set_launch_order(pantr_batch_t *b, int32_t B, const int32_t *order);
/* Without such an order, option "launch_order=previous" (or "auto": where a launch has more workgroups than the chip
 * holds at once; the default is "natural") starts the instances longest first by the time each took in the solve
 * before it on this handle, ranked on the device behind every solve kernel - for batches that are solved again.  source (may be NULL): what the most recent launch used - 0 by index,
 * 1 pantr_batch_set_launch_order, 2 the solve before it; next_order (may be NULL): the B indices that solve left
 * behind for the next launch of B instances (PANTR_ERR_RUNTIME if it left none).  Synchronises the device. */
int pantr_batch_last_launch_order(pantr_batch_t *b, int32_t B, int32_t *source, int32_t *next_order);
/* Per-iteration trace of the first `instances` instances (cap rows of 11 doubles each:
 * eps, psi, ||p||, gamma, radius, rho, q_model, fbe, |J|, cg_its, accepted; with method=panoc,
 * strucpanoc: eps, psi, ||p||, gamma, tau, fbe, stored L-BFGS pairs, 4 unused) — what
 * solver.print_interval would print in the reference (mpc-driver.cpp:52). */
int pantr_batch_set_trace(pantr_batch_t *b, int32_t instances, int32_t cap);
int pantr_batch_get_trace(pantr_batch_t *b, double *out /* [instances][cap][11] */);

/* ---- problem functions on the device (parity tests of the individual rows) --------------
 * eval_psi / eval_psi_grad_psi / eval_hess_psi_prod: formulation.hpp:145-168;
 * eval_prox_grad_step: C signature problems/problems/lasso.cpp:72-84, box semantics
 * drivers/results.hpp:268-271; simulate_dynamics: formulation.hpp:258-261. Host pointers. */
int pantr_batch_eval_psi(pantr_batch_t *b, int32_t B, const double *x0, const double *u, double *psi);
int pantr_batch_eval_psi_grad_psi(pantr_batch_t *b, int32_t B, const double *x0, const double *u,
                                  double *psi, double *grad);
int pantr_batch_eval_hess_psi_prod(pantr_batch_t *b, int32_t B, const double *x0, const double *u,
                                   const double *v, double *Hv);
/* The same functions for problems with general constraints (m > 0), with the arguments of
 * eval_psi_grad_psi / eval_hess_psi_prod in formulation.hpp:145-168: multipliers y[B][m] and
 * penalties Sigma[B][m]; yhat[B][m] (may be NULL) receives Sigma (zeta - clamp(zeta, D)). */
int pantr_batch_eval_psi_grad_psi_al(pantr_batch_t *b, int32_t B, const double *x0, const double *u,
                                     const double *y, const double *Sigma, double *psi, double *grad,
                                     double *yhat);
int pantr_batch_eval_hess_psi_prod_al(pantr_batch_t *b, int32_t B, const double *x0, const double *u,
                                      const double *y, const double *Sigma, const double *v, double *Hv);
/* eval_grad_L (formulation.hpp:98,170-174): grad[B][n] = grad f(u) + grad g(u)' y with the
 * caller's multipliers y[B][m] (y may be NULL when m = 0, where L = f = psi). */
int pantr_batch_eval_grad_L(pantr_batch_t *b, int32_t B, const double *x0, const double *u, const double *y,
                            double *grad);
/* eval_g (formulation.hpp:70,127-129): g[B][m] = stage constraints c(x_0..x_{N-1}) then c_N(x_N)
 * (python/alpaqa_mpc_benchmarks/formulations/ss.py:39-40); nothing is written when m = 0. */
int pantr_batch_eval_g(pantr_batch_t *b, int32_t B, const double *x0, const double *u, double *g);
/* scalars[B][4] = { p'p, grad'p, ||Pi_C(x-grad)-x||_inf, |J(gamma)| } */
int pantr_batch_eval_prox_grad_step(pantr_batch_t *b, int32_t B, double gamma, const double *x,
                                    const double *grad, double *xhat, double *p, double *scalars);
int pantr_batch_simulate_dynamics(pantr_batch_t *b, int32_t B, const double *x, const double *u,
                                  double *x_next);

/* ---- closed loops: replaces do_mpc_experiment (mpc-driver.cpp:123-210) ---------------------
 * B independent MPC trajectories advanced in lock-step on the device (no host round trip between
 * steps): initial state x_init[B][nx] (NULL = the problem's), num_dist plant steps under the
 * disturbance u_dist[B][nu] (:153-154), an unrecorded first solve (:158), then num_sim times
 * { warm: shift the previous solution by nu (:168-170) | cold: the initial guess (:192-193) ;
 * solve ; record ; plant step with the first input block (:205) }.  Outputs (host):
 * states[num_sim+1][B][nx], inputs[num_sim][B][n] (the whole optimal u per step, as
 * results.hpp:227-251 stores it), stats[num_sim][B]. */
int pantr_batch_mpc(pantr_batch_t *b, int32_t B, int32_t num_sim, int32_t warm, int32_t num_dist,
                    const double *x_init, const double *u_dist, double *states, double *inputs,
                    pantr_stats_t *stats);

/* The same closed loops, also recording what a replay batch needs (any of the three may be NULL):
 * the guesses every step was solved from, guesses_u[num_sim][B][n] and guesses_y[num_sim][B][m]
 * (the "guesses" the reference driver keeps per step would be these), and the multipliers each solve
 * returned, multipliers[num_sim][B][m].  Solving step i again from (states[i], guesses_u[i],
 * guesses_y[i]) reproduces inputs[i] and multipliers[i]: that is the replay batching of the bench. */
int pantr_batch_mpc_record(pantr_batch_t *b, int32_t B, int32_t num_sim, int32_t warm, int32_t num_dist,
                           const double *x_init, const double *u_dist, double *states, double *inputs,
                           pantr_stats_t *stats, double *guesses_u, double *guesses_y, double *multipliers);

/* ---- the reference's generated problem library as a cross-checker ---------------------------
 * load_benchmark_casadi_problem (drivers/casadi-dll-wrapper.cpp:25-67) opens
 * "<dir>/<name>_<formulation>_<N>.so", the C code alpaqa.casadi_generator.generate_casadi_problem
 * emits (problems/generate_problem.py:55-61), through alpaqa::CasADiProblem / libcasadi.  The generated
 * C has a calling convention of its own that needs neither (int f(const double **arg, double **res,
 * long long *iw, double *w, int mem), f_work, f_n_in/out, f_sparsity_in/out; host/casadi_dll.hpp): these
 * entry points open such a file on the HOST and evaluate its functions - to compare them with the device
 * functions, never to solve with them.  No device is needed for open / function_info / call.
 * Errors: the file cannot be opened -> PANTR_ERR_RUNTIME; a function it does not export ->
 * PANTR_ERR_OUT_OF_RANGE (the reference: std::out_of_range, casadi-dll-wrapper.cpp:104-106); argument
 * counts or sizes that do not fit -> PANTR_ERR_INVALID_ARGUMENT. */
typedef struct pantr_casadi_dll pantr_casadi_dll_t;
int pantr_casadi_dll_open(const char *so_path, pantr_casadi_dll_t **out);
void pantr_casadi_dll_close(pantr_casadi_dll_t *d);
/* Counts, sizes (nonzeros per input / output, -1 where the library exports no pattern; at most `cap`
 * entries are written to each array) and work sizes {sz_arg, sz_res, sz_iw, sz_w} of function `name`. */
int pantr_casadi_dll_function_info(pantr_casadi_dll_t *d, const char *name, int32_t *n_in, int32_t *n_out,
                                   int64_t *nnz_in, int64_t *nnz_out, int32_t cap, int64_t work[4]);
/* One call: args[n_in] (NULL entry = zeros), res[n_out] (NULL entry = output not wanted). */
int pantr_casadi_dll_call(pantr_casadi_dll_t *d, const char *name, int32_t n_in, const double *const *args,
                          int32_t n_out, double *const *res);

/* Compare the generated library's psi, psi_grad_psi, hess_psi_prod and g with this handle's device
 * functions on `samples` seeded points (x = u inside box C around the initial guess, p = the initial
 * state perturbed, y, Sigma > 0 and v random, zl / zu = the handle's box D), inputs in the order
 * formulation.hpp:88-97 builds them: (x, p, y, Sigma, [s = 1,] zl, zu[, v]); whether psi and
 * psi_grad_psi take the scale s is read off their n_in.  max_rel_err[PANTR_DLL_CHECK_FIELDS] receives,
 * per quantity, max over the samples of ||dll - device||_inf / max(||dll||_inf, 1e-300):
 *   [0] psi (psi_grad_psi's value), [1] grad psi, [2] hess_psi_prod, [3] g, [4] yhat (psi's 2nd output);
 * -1 where the library has no such function (first-order formulations carry no hess_psi_prod) or the
 * problem no such quantity (m = 0: g, yhat).  The bar of BASELINE.json's north_star is 1e-10.
 * A library generated for other sizes (n, m, nx from its sparsity patterns) is refused
 * (PANTR_ERR_INVALID_ARGUMENT), not mis-fed. */
#define PANTR_DLL_CHECK_FIELDS 5
int pantr_batch_check_casadi_dll(pantr_batch_t *b, const char *so_path, int32_t samples, double *max_rel_err);

/* ---- misc -------------------------------------------------------------------------------- */
const char *pantr_last_error(void);
/* Library name and version; any build switch that changes what is computed or reported is
 * appended ("+phase-timing"), so a caller can refuse a non-product build. */
const char *pantr_version(void);
/* sha256 (hex) over the sources the library was built from (the headers and .hip files of csrc/, the
 * headers of host/, in name order, then this header); the Python binding compares it with the
 * tree's and refuses a stale library. */
const char *pantr_source_hash(void);
/* sizeof(pantr_stats_t) as compiled into the library: a binding checks its own record layout against it. */
size_t pantr_stats_size(void);
int pantr_device_count(void);
/* Achievable HBM bandwidth of `device`: a 16-bytes-per-lane device copy of `bytes` bytes (choose well
 * beyond the 256 MiB Infinity Cache), best of `reps` timed passes by HIP events, as read + written
 * GB/s.  The measured yardstick bench.py reports next to the 8 TB/s spec peak. */
int pantr_device_copy_bandwidth(int32_t device, int64_t bytes, int32_t reps, double *gb_per_s);
/* Name of the device kernel a solve of B instances on this handle launches (as rocprofv3 prints it, without
 * the namespace): which one depends on the problem, on `method=` and - method=pantr - on `speculation=`
 * and B (one or two wavefronts per instance).  Measurement code names its dominant kernel with it. */
const char *pantr_batch_kernel_name(const pantr_batch_t *b, int32_t B);
/* Bytes of LDS per workgroup and of HBM workspace per instance for this handle: of the kernel a solve of the
 * handle's maximum batch size launches (pantr_batch_footprint), or of B instances (pantr_batch_footprint_for) - a
 * smaller batch may run the two-wavefront kernel, which takes about twice the LDS (pantr_batch_kernel_name(b, B)
 * names the kernel the figure belongs to). */
int pantr_batch_footprint(const pantr_batch_t *b, int64_t *lds_bytes, int64_t *ws_bytes_per_instance);
int pantr_batch_footprint_for(const pantr_batch_t *b, int32_t B, int64_t *lds_bytes, int64_t *ws_bytes_per_instance);
/* pantr_batch_solve_device checks where its four buffers live (hipPointerGetAttributes) once per set of base
 * pointers and not again while a caller passes the same set - the queries stay off the launch path.  Contract: a
 * caller that frees or re-allocates any of those buffers calls this before its next solve (a failed solve forgets
 * the set by itself).  A handle is not thread-safe: one thread at a time. */
int pantr_batch_forget_buffers(pantr_batch_t *b);

#ifdef __cplusplus
}
#endif
#endif /* PANTR_MI355X_H */
