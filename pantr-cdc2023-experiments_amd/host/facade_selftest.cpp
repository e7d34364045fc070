// facade-selftest: calls every TypeErasedProblem::eval_* of host/alpaqa_compat.hpp once and prints the
// results, one labelled row of %.17g values per output, for tests/test_gpu_driver.py to compare with
// the ctypes path (same C ABI underneath; this checks the C++ argument marshalling, in particular
// that y and Sigma reach the device for problems with general constraints).
//
//   facade-selftest <problem> <horizon> <input file>
// input file: five rows of comma-separated doubles: param (nx), x (n), y (m), Sigma (m), v (n); the
// rows of size m are empty lines when m = 0.
#include "alpaqa_compat.hpp"
#include "problem_data.hpp"
#include "results.hpp"

#include <cstdio>
#include <fstream>
#include <iostream>

using alpaqa::vec;

static void row(const char *label, std::span<const double> v) {
    std::printf("%s", label);
    for (double x : v)
        std::printf(" %.17g", x);
    std::printf("\n");
}

int main(int argc, char *argv[]) try {
    if (argc != 4) {
        std::cerr << "usage: " << argv[0] << " <problem> <horizon> <input file>\n";
        return 2;
    }
    alpaqa::TypeErasedProblem<> problem{argv[1], std::stoi(argv[2])};
    const auto n = problem.get_n(), m = problem.get_m();
    std::ifstream in(argv[3]);
    if (!in)
        throw std::runtime_error("cannot open the input file");
    vec x, y, Sigma, v;
    std::string err;
    for (vec *dst : {&problem.param, &x, &y, &Sigma, &v})
        if (!pantr::host::read_csv_row(in, *dst, err))
            throw std::runtime_error("input file: " + err);
    vec yhat(m), grad(n), grad_L(n), Hv(n), g(m), pd(m), xh(n), p(n), work_n(n), work_m(m);
    const double psi = problem.eval_psi(x, y, Sigma, yhat);
    row("psi", {{psi}});
    row("yhat", yhat);
    const double psi2 = problem.eval_psi_grad_psi(x, y, Sigma, grad, work_n, work_m);
    row("psi2", {{psi2}});
    row("grad_psi", grad);
    problem.eval_grad_L(x, y, grad_L, work_n);
    row("grad_L", grad_L);
    problem.eval_hess_psi_prod(x, y, Sigma, 1.0, v, Hv);
    row("hess_psi_prod", Hv);
    problem.eval_g(x, g);
    row("g", g);
    problem.eval_proj_diff_g(g, pd);
    row("proj_diff_g", pd);
    problem.eval_prox_grad_step(0.05, x, grad, xh, p);
    row("xhat", xh);
    row("p", p);
    row("simulate", problem.simulate(problem.param, std::span<const double>(x).first(problem.nu)));
    const auto C = problem.get_box_C();
    row("C_lb", C.lowerbound);
    row("C_ub", C.upperbound);
    const auto kkt = pantr::host::compute_kkt_error(problem, x, y);
    row("kkt", {{kkt.stationarity, kkt.constr_violation, kkt.complementarity}});
    return 0;
} catch (std::exception &e) {
    std::cerr << "Error: " << e.what() << std::endl;
    return 1;
}
