// benchmark-mpc-driver: the reference's MPC benchmark CLI (drivers/mpc-driver.cpp), "pantr"
// method, over the MI355X batched solver.
//
//   benchmark-mpc-driver [<path>/]<name> <horizon> <formulation> method=pantr|panoc|strucpanoc [<key>=<value>...]
//
// Same positional arguments, options and output as the reference (mpc-driver.cpp:39-43,
// 326-343): num_sim, num_dist, u_dist, warm, results_name, and the solver./dir./accel./alm.
// option groups.  The closed loop is the one of mpc-driver.cpp:123-210; every solve runs on the
// GPU (batch of one instance per MPC step, since step i+1 depends on step i).  Extension:
// `replay=true` re-solves all recorded steps of the finished closed loop as ONE batch and checks
// that the batched solutions equal the sequential ones (the replay batching of BASELINE.json
// configs[1]); `device=<k>` picks the GPU; `check_dll=true` compares the generated problem library
// next to <path> with the device functions before solving; `solver.print_interval=<k>` (mpc-driver.cpp:52)
// prints every k-th iterate of each solve from the device trace.
#include "alpaqa_compat.hpp"
#include "results.hpp"

#include <chrono>
#include <cmath>
#include <filesystem>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <numeric>
#include <typeinfo>

namespace fs = std::filesystem;
using alpaqa::crvec;
using alpaqa::length_t;
using alpaqa::real_t;
using alpaqa::rvec;
using alpaqa::vec;
using config_t = alpaqa::DefaultConfig;
using Problem  = alpaqa::TypeErasedProblem<config_t>;

static void print_usage(const char *a0) {
    std::cout << "Usage: " << a0
              << " [<path>/]<name> <horizon> <second-order> [method=<solver>] "
                 "[<key>=<value>...]\n";
}

/// mpc-driver.cpp:47-73
template <class InnerSolver>
auto make_inner_solver(std::span<const std::string_view> extra_opts) {
    typename InnerSolver::Params solver_param;
    solver_param.max_iter       = 50'000;
    solver_param.print_interval = 0;
    solver_param.stop_crit      = alpaqa::PANOCStopCrit::FPRNorm;
    alpaqa::params::set_params(solver_param, "solver", extra_opts);
    using Direction = typename InnerSolver::Direction;
    typename Direction::DirectionParams dir_param;
    typename Direction::AcceleratorParams accel_param;
    alpaqa::params::set_params(dir_param, "dir", extra_opts);
    alpaqa::params::set_params(accel_param, "accel", extra_opts);
    return InnerSolver{solver_param, Direction{{.accelerator = accel_param, .direction = dir_param}}};
}

/// mpc-driver.cpp:104-115
template <class InnerSolver>
auto make_solver(std::span<const std::string_view> extra_opts) {
    using ALMSolver = alpaqa::ALMSolver<InnerSolver>;
    typename ALMSolver::Params alm_param;
    alm_param.max_iter       = 200;
    alm_param.tolerance      = 1e-8;
    alm_param.dual_tolerance = 1e-8;
    alm_param.print_interval = 0;
    alpaqa::params::set_params(alm_param, "alm", extra_opts);
    return ALMSolver{alm_param, make_inner_solver<InnerSolver>(extra_opts)};
}

struct MPCResults {
    vec states, inputs, runtimes, guesses, y_guesses, multipliers;
};

/// mpc-driver.cpp:123-210 (single shooting)
template <class Solver>
MPCResults do_mpc_experiment(Problem &problem, Solver &solver, length_t N,
                             std::span<const std::string_view> extra_opts) {
    vec &x0       = problem.param; // mutable view of the problem parameter (:126)
    const auto nu = problem.nu, nx = problem.nx, nc = problem.nc, nc_N = problem.nc_N;
    if (static_cast<length_t>(x0.size()) != nx)
        throw std::logic_error("Invalid x0 dimension");
    const auto n = problem.get_n(), m = problem.get_m();
    if (n != nu * N)
        throw std::logic_error("only the single-shooting formulation is available");
    vec u_initial = problem.get_initial_guess();
    vec u         = u_initial;
    vec y(m, 0.0);
    const alpaqa::Box box_C = problem.get_box_C();
    vec u_dist(box_C.upperbound.begin(), box_C.upperbound.begin() + nu); // :140
    for (real_t &ud : u_dist)
        if (!std::isfinite(ud))
            ud = 0;
    alpaqa::params::set_params(u_dist, "u_dist", extra_opts);
    if (static_cast<length_t>(u_dist.size()) != nu)
        throw std::out_of_range("Invalid u_dist size");
    std::cout << "u_dist =";
    for (real_t v : u_dist)
        std::cout << ' ' << v;
    std::cout << '\n';
    length_t num_dist = 3;
    alpaqa::params::set_params(num_dist, "num_dist", extra_opts);
    length_t num_sim = N;
    alpaqa::params::set_params(num_sim, "num_sim", extra_opts);
    // Initial disturbance
    for (length_t i = 0; i < num_dist; ++i)
        x0 = problem.simulate(x0, u_dist);
    // Initial solve
    bool warm_start = true;
    alpaqa::params::set_params(warm_start, "warm", extra_opts);
    auto solve = [&](rvec u_, rvec y_) {
        auto stats = solver(problem, u_, y_);
        std::cerr << stats.status << '\n';
        real_t t = std::chrono::duration<real_t>(stats.elapsed_time).count();
        return stats.status == alpaqa::SolverStatus::Converged ? t : -t; // :219-221
    };
    solve(u, y);
    // Simulate
    x0 = problem.simulate(x0, crvec{u}.first(nu));
    problem.evaluations->reset();
    MPCResults res;
    res.runtimes.assign(num_sim, std::nan(""));
    res.states.resize(nx * (num_sim + 1));
    res.inputs.resize(n * num_sim);
    res.guesses.resize(n * num_sim);
    res.y_guesses.resize(m * num_sim);
    res.multipliers.resize(m * num_sim);
    for (length_t i = 0; i < num_sim; ++i) {
        if (warm_start) { // shift previous solution (:168-175)
            std::copy(u.begin() + nu, u.end(), u.begin());
            if (nc != nc_N)
                throw std::logic_error("invalid multiplier shift");
            if (m > nc_N)
                std::copy(y.begin() + nc_N, y.end(), y.begin());
        } else {
            u = u_initial;
            std::fill(y.begin(), y.end(), 0.0);
        }
        std::copy(u.begin(), u.end(), res.guesses.begin() + i * n);
        std::copy(y.begin(), y.end(), res.y_guesses.begin() + i * m);
        res.runtimes[i] = solve(u, y);
        std::copy(y.begin(), y.end(), res.multipliers.begin() + i * m);
        std::copy(x0.begin(), x0.end(), res.states.begin() + i * nx);
        std::copy(u.begin(), u.end(), res.inputs.begin() + i * n);
        x0 = problem.simulate(x0, crvec{u}.first(nu));
    }
    std::copy(x0.begin(), x0.end(), res.states.begin() + num_sim * nx);
    return res;
}

int main(int argc, char *argv[]) try {
    if (argc < 1)
        return -1;
    if (argc == 1)
        return print_usage(argv[0]), 0;
    if (argc < 4)
        return print_usage(argv[0]), -1;
    fs::path path{argv[1]};
    if (!path.has_parent_path())
        path = fs::canonical(fs::path(argv[0])).parent_path() / path;
    if (!path.has_parent_path() || !path.has_filename())
        return (std::cerr << "invalid problem path\n"), -1;
    length_t horizon             = std::stoi(argv[2]);
    std::string_view formulation = argv[3];
    std::vector<std::string_view> extra_opts;
    std::copy(argv + 4, argv + argc, std::back_inserter(extra_opts));

    std::string name = path.filename().string();
    if (formulation != "ss2p" && formulation != "ss" && formulation != "ss2")
        throw std::invalid_argument("formulation '" + std::string(formulation) +
                                    "' is not available (single shooting only: ss, ss2p)");
    length_t device = 0, replay_len = 0;
    bool replay = false;
    alpaqa::params::set_params(device, "device", extra_opts);
    alpaqa::params::set_params(replay, "replay", extra_opts);
    length_t num_sim = horizon;
    alpaqa::params::set_params(num_sim, "num_sim", extra_opts);
    replay_len = replay ? num_sim : 1;

    // Load the problem (casadi-dll-wrapper.cpp:25-67: "<name>_<formulation>_<N>.so").  The functions are
    // the library's device code for <name>; the numeric data of the generated problem's sidecar
    // files (bounds, initial state, initial guess) are used when they lie next to that path.
    fs::path full_path = path.parent_path() / (name + "_" + std::string(formulation) + "_" +
                                               std::to_string(horizon) + ".so");
    const int instances = static_cast<int>(std::max<length_t>(1, replay_len));
    Problem problem = fs::exists(fs::path(full_path).replace_extension(".guess.csv"))
                          ? Problem{Problem::FromFiles{}, path.string(), formulation, horizon, instances,
                                    static_cast<int>(device)}
                          : Problem{name, horizon, instances, static_cast<int>(device)};

    // check_dll=true: the reference's generated problem library `<dir>/<name>_<formulation>_<N>.so`
    // (casadi-dll-wrapper.cpp:29-31) is opened on the host through the generated code's own C convention and its
    // psi / psi_grad_psi / hess_psi_prod / g are compared with the device functions this driver solves with
    // (pantr_batch_check_casadi_dll; bar 1e-10, BASELINE.json north_star).  The library is never solved with.
    bool check_dll = false;
    alpaqa::params::set_params(check_dll, "check_dll", extra_opts);
    if (check_dll) {
        length_t samples = 8;
        alpaqa::params::set_params(samples, "check_dll_samples", extra_opts);
        double err[PANTR_DLL_CHECK_FIELDS];
        alpaqa::check(pantr_batch_check_casadi_dll(problem.get(), full_path.c_str(), static_cast<int32_t>(samples), err));
        static const char *names[PANTR_DLL_CHECK_FIELDS] = {"psi", "grad_psi", "hess_psi_prod", "g", "yhat"};
        bool ok = true;
        std::cout << "check_dll: " << full_path.string() << " against the device functions, " << samples
                  << " points, max relative difference\n";
        for (int i = 0; i < PANTR_DLL_CHECK_FIELDS; ++i) {
            std::cout << "check_dll:   " << names[i] << std::string(14 - std::string(names[i]).size(), ' ');
            if (err[i] == -1.0)
                std::cout << "(not in this library / problem)\n";
            else
                std::cout << err[i] << (err[i] <= 1e-10 ? "" : "   > 1e-10") << '\n';
            ok = ok && (err[i] == -1.0 || err[i] <= 1e-10);
        }
        std::cout << "check_dll: the generated functions and the device functions "
                  << (ok ? "agree to 1e-10" : "DIFFER") << std::endl;
        if (!ok)
            throw std::runtime_error("check_dll: the generated functions and the device functions DIFFER");
    }

    std::string_view solver_name = "fbetrust"; // mpc-driver.cpp:359
    alpaqa::params::set_params(solver_name, "method", extra_opts);

    // do_experiment (mpc-driver.cpp:280-326) + the optional batched replay
    auto run = [&](auto solver) {
             MPCResults mpc    = do_mpc_experiment(problem, solver, horizon, extra_opts);

             using namespace pantr::host;
             MPCBenchmarkResults results;
             results.problem_name = name;
             results.path         = path.parent_path().string();
             results.full_path    = full_path.string();
             results.formulation  = std::string(formulation);
             results.solver       = solver.get_name();
             results.horizon      = horizon;
             results.nvar         = problem.get_n();
             results.ncon         = problem.get_m();
             results.nx           = problem.nx;
             results.second_order = formulation.ends_with("2");
             results.evals        = *problem.evaluations;
             results.runtimes     = mpc.runtimes;
             results.states       = mpc.states;
             results.inputs       = mpc.inputs;
             results.options      = extra_opts;
             results.timestamp    = std::chrono::duration_cast<std::chrono::milliseconds>(
                                     std::chrono::system_clock::now().time_since_epoch())
                                     .count();
             print_results(std::cout, results);

             if (replay) { // all recorded steps as one batch: must reproduce the closed loop
                 const auto n = problem.get_n(), nx = problem.nx;
                 const length_t B = static_cast<length_t>(mpc.runtimes.size());
                 vec xs(mpc.guesses);
                 vec x0s(mpc.states.begin(), mpc.states.begin() + B * nx);
                 auto t0    = std::chrono::steady_clock::now();
                 vec ys(mpc.y_guesses);
                 auto stats = solver.solve_batch(problem, B, x0s, xs, ys);
                 auto dt    = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                 bool same  = std::equal(xs.begin(), xs.end(), mpc.inputs.begin()) &&
                             std::equal(ys.begin(), ys.end(), mpc.multipliers.begin());
                 length_t conv = 0;
                 for (auto &s : stats)
                     conv += s.status == alpaqa::SolverStatus::Converged;
                 std::cout << "replay: " << B << " steps as one batch in " << dt << " s ("
                           << B / dt << " steps/s), " << conv << " converged, solutions "
                           << (same ? "identical to" : "DIFFERENT from") << " the closed loop\n";
                 if (!same)
                     throw std::runtime_error("replay batch does not reproduce the closed loop");
                 (void)n;
             }

             auto suffix       = std::to_string(results.timestamp) + '_' + random_hex_string();
             auto results_name = "results_" + suffix;
             alpaqa::params::set_params(results_name, "results_name", extra_opts);
             std::cout << "results: " << suffix << std::endl;
             std::ofstream res_file{results_name + ".py"};
             if (res_file)
                 write_results(res_file, results);
             else
                 std::cerr << "Unable to open results file " << std::quoted(results_name + ".py") << std::endl;
    };
    std::map<std::string_view, std::function<void()>> solvers{
        {"pantr", [&] { // mpc-driver.cpp:364-370
             using Accelerator = alpaqa::NewtonTRDirection<config_t>;
             using InnerSolver = alpaqa::PANTRSolver<Accelerator>;
             run(make_solver<InnerSolver>(extra_opts));
         }},
        {"strucpanoc", [&] { // mpc-driver.cpp:371-377
             using Accelerator = alpaqa::StructuredLBFGSDirection<config_t>;
             using InnerSolver = alpaqa::PANOCSolver<Accelerator>;
             run(make_solver<InnerSolver>(extra_opts));
         }},
        {"panoc", [&] { // mpc-driver.cpp:378-384
             using Accelerator = alpaqa::LBFGSDirection<config_t>;
             using InnerSolver = alpaqa::PANOCSolver<Accelerator>;
             run(make_solver<InnerSolver>(extra_opts));
         }},
    };

    auto solver_it = solvers.find(solver_name);
    if (solver_it != solvers.end())
        solver_it->second();
    else
        throw std::invalid_argument("Unknown solver '" + std::string(solver_name) + "'\n" +
                                    "  Available solvers: 'pantr', 'strucpanoc', 'panoc'");
} catch (std::exception &e) {
    std::cerr << "Error: " << typeid(e).name() << ":\n  " << e.what() << std::endl;
    return -1;
}
