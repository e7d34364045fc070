// Results-file writer: a Python module `results = {...}` with the keys and literal syntax of
// the reference (drivers/results.hpp:227-251 for MPC runs, evaluation counters :63-87, literal
// syntax drivers/output.hpp:15-55), so new-benchmarks-paper/benchmark_util.py:4-9 and
// perfplot-cold-warm.py load the files unchanged.
#pragma once
#include "alpaqa_compat.hpp"

#include <cstdio>
#include <iomanip>
#include <ostream>
#include <random>
#include <span>
#include <string>
#include <string_view>
#include <vector>

namespace pantr::host {

inline std::string float_to_str(double v) {
    if (std::isnan(v))
        return "nan";
    if (std::isinf(v))
        return v > 0 ? "inf" : "-inf";
    char buf[40];
    std::snprintf(buf, sizeof buf, "%+.17e", v);
    return buf;
}

inline void python_literal(std::ostream &os, double v) { os << float_to_str(v); }
inline void python_literal(std::ostream &os, long long v) { os << v; }
inline void python_literal(std::ostream &os, unsigned v) { os << v; }
inline void python_literal(std::ostream &os, std::ptrdiff_t v) { os << v; }
inline void python_literal(std::ostream &os, bool v) { os << (v ? "True" : "False"); }
inline void python_literal(std::ostream &os, std::string_view v) { os << std::quoted(v); }
inline void python_literal(std::ostream &os, const std::string &v) { os << std::quoted(v); }
inline void python_literal(std::ostream &os, const std::vector<double> &v) {
    os << "np.array([";
    for (size_t i = 0; i < v.size(); ++i)
        os << (i ? ", " : "") << float_to_str(v[i]);
    os << "])";
}
/// Column-major matrix (rows x cols), printed as a list of rows like Eigen's print_python.
struct MatView {
    const std::vector<double> &data;
    std::ptrdiff_t rows, cols;
};
inline void python_literal(std::ostream &os, MatView m) {
    os << "np.array([";
    for (std::ptrdiff_t r = 0; r < m.rows; ++r) {
        os << (r ? ",\n [" : "[");
        for (std::ptrdiff_t c = 0; c < m.cols; ++c)
            os << (c ? ", " : "") << float_to_str(m.data[c * m.rows + r]);
        os << "]";
    }
    os << "])";
}
inline void python_literal(std::ostream &os, std::span<const std::string_view> v) {
    os << "[";
    for (const auto &el : v)
        os << std::quoted(el) << ", ";
    os << "]";
}

template <class A>
void dict_elem(std::ostream &os, std::string_view k, A &&a) {
    os << "    " << std::quoted(k) << ": ";
    python_literal(os, std::forward<A>(a));
    os << ",\n";
}

inline void write_evaluations(std::ostream &os, const alpaqa::EvalCounter &e) {
    os << std::quoted("evaluations") << ": {\n";
    dict_elem(os, "proj_diff_g", e.proj_diff_g);
    dict_elem(os, "proj_multipliers", e.proj_multipliers);
    dict_elem(os, "prox_grad_step", e.prox_grad_step);
    dict_elem(os, "f", e.f);
    dict_elem(os, "grad_f", e.grad_f);
    dict_elem(os, "f_grad_f", e.f_grad_f);
    dict_elem(os, "f_g", e.f_g);
    dict_elem(os, "grad_f_grad_g_prod", e.grad_f_grad_g_prod);
    dict_elem(os, "g", e.g);
    dict_elem(os, "grad_g_prod", e.grad_g_prod);
    dict_elem(os, "grad_gi", e.grad_gi);
    dict_elem(os, "grad_L", e.grad_L);
    dict_elem(os, "hess_L_prod", e.hess_L_prod);
    dict_elem(os, "hess_L", e.hess_L);
    dict_elem(os, "hess_ψ_prod", e.hess_psi_prod); // alpaqa's convention: calls (pantr_stats_t::n_hess_psi_prod_calls)
    dict_elem(os, "hess_ψ_prod_evaluated", e.hess_psi_prod_evaluated); // not an alpaqa key: products the device evaluated
    dict_elem(os, "hess_ψ", e.hess_psi);
    dict_elem(os, "ψ", e.psi);
    dict_elem(os, "grad_ψ", e.grad_psi);
    dict_elem(os, "ψ_grad_ψ", e.psi_grad_psi);
    os << "},\n";
}

struct MPCBenchmarkResults {
    std::string problem_name, path, full_path, formulation, solver;
    std::ptrdiff_t horizon = 0, nvar = 0, ncon = 0, nx = 0;
    bool second_order = false;
    alpaqa::EvalCounter evals;
    std::vector<double> runtimes; ///< seconds, negative when not converged (mpc-driver.cpp:219-221)
    std::vector<double> states;   ///< nx x (num_sim+1), column-major
    std::vector<double> inputs;   ///< n x num_sim, column-major
    std::span<const std::string_view> options;
    long long timestamp = 0;
};

inline void write_results(std::ostream &os, const MPCBenchmarkResults &r) {
    const std::ptrdiff_t num_sim = static_cast<std::ptrdiff_t>(r.runtimes.size());
    os << "from numpy import nan, inf\n"
          "import numpy as np\n"
          "__all__ = ['results']\n"
          "results = {\n";
    dict_elem(os, "opts", r.options);
    dict_elem(os, "time_utc_ms", r.timestamp);
    dict_elem(os, "runtimes", r.runtimes);
    dict_elem(os, "states", MatView{r.states, r.nx, num_sim + 1});
    dict_elem(os, "inputs", MatView{r.inputs, r.nvar, num_sim});
    dict_elem(os, "solver", r.solver);
    write_evaluations(os, r.evals);
    dict_elem(os, "problem", r.problem_name);
    dict_elem(os, "nvar", r.nvar);
    dict_elem(os, "ncon", r.ncon);
    dict_elem(os, "path", r.full_path);
    dict_elem(os, "horizon", r.horizon);
    dict_elem(os, "second_order", r.second_order);
    dict_elem(os, "formulation", r.formulation);
    os << "}\n";
}

inline void print_results(std::ostream &os, const MPCBenchmarkResults &r) {
    double time_s = 0;
    for (double t : r.runtimes)
        time_s += std::fabs(t);
    const auto &e = r.evals;
    os << '\n'
       << "prox_grad_step:" << std::setw(8) << e.prox_grad_step << "\n"
       << "        grad_L:" << std::setw(8) << e.grad_L << "\n"
       << "   hess_ψ_prod:" << std::setw(8) << e.hess_psi_prod << "\n"
       << "             ψ:" << std::setw(8) << e.psi << "\n"
       << "        grad_ψ:" << std::setw(8) << e.grad_psi << "\n"
       << "      ψ_grad_ψ:" << std::setw(8) << e.psi_grad_psi << "\n\n"
       << "solver:  " << r.solver << '\n'
       << "problem: " << r.problem_name << " " << r.horizon << " (from " << std::quoted(r.path) << ")\n"
       << "num var: " << r.nvar << '\n'
       << "num con: " << r.ncon << '\n'
       << "time: " << std::scientific << std::setprecision(3) << time_s << " s\n"
       << std::endl;
}

/// KKT residuals of (x, y), drivers/results.hpp:258-286: stationarity ||Pi_C(x - grad L) - x||_inf,
/// constraint violation ||g - Pi_D(g)||_inf, complementarity max |y_j v_j|.
struct KKTError {
    double stationarity, constr_violation, complementarity;
};
template <class Problem>
KKTError compute_kkt_error(const Problem &problem, std::span<const double> x, std::span<const double> y) {
    const size_t n = x.size(), m = y.size();
    std::vector<double> z(n), grad_Lx(n), work(n), g(m), v(m);
    problem.eval_grad_L(x, y, grad_Lx, work);
    problem.eval_prox_grad_step(1, x, grad_Lx, work, z);
    double stationarity = 0, constr_violation = 0, complementarity = 0;
    for (double zi : z)
        stationarity = std::fmax(stationarity, std::fabs(zi));
    problem.eval_g(x, g);
    problem.eval_proj_diff_g(g, v);
    for (size_t j = 0; j < m; ++j) {
        constr_violation = std::fmax(constr_violation, std::fabs(v[j]));
        complementarity  = std::fmax(complementarity, std::fabs(y[j] * v[j]));
    }
    return {stationarity, constr_violation, complementarity};
}

inline std::string random_hex_string() {
    std::random_device rd;
    char buf[9];
    std::snprintf(buf, sizeof buf, "%08x", static_cast<unsigned>(rd()));
    return buf;
}

} // namespace pantr::host
