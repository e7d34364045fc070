// TEST INFRASTRUCTURE ONLY - a stand-in for a CasADi-generated problem library
// `<name>_<formulation>_<N>.so` (what problems/generate_problem.py:55-61 produces and
// drivers/casadi-dll-wrapper.cpp:25-67 loads), written to the generated code's C convention:
//     int f(const double **arg, double **res, long long *iw, double *w, int mem), f_work, f_n_in, f_n_out,
//     f_sparsity_in, f_sparsity_out, f_incref, f_decref, f_checkout, f_release
// with the function set and input order of problems/problems/formulation.hpp:62-101.
//
// The bodies wrap THIS PROJECT'S CPU ORACLE (oracle/*.hpp).  A stub therefore PINS NOTHING about parity
// with the reference: comparing the device functions with it only repeats the GPU = oracle tests.  What it
// tests is the loader and the checker (host/casadi_dll.hpp, pantr_batch_check_casadi_dll): symbol lookup,
// work-size queries, both sparsity forms, marshalling, and the ARGUMENT ORDER - the stub refuses a call
// whose scale is not 1 or whose zl / zu are not the problem's box D, and -DSTUB_SWAP builds a library that
// reads two of its arguments in the wrong order, which the checker must report as a difference.
//
// Build (tests/test_casadi_dll.py does it): g++ -O1 -ffp-contract=off -std=c++20 -shared -fPIC -I oracle
//     -DSTUB_PROBLEM=<0 hanging_chain | 1 quadcopter | 2 realistic_quadcopter> -DSTUB_N=<horizon>
//     [-DSTUB_WITH_SCALE=<0|1>: psi and psi_grad_psi take the scale s (formulation.hpp:88-92) or not]
//     [-DSTUB_FULL_PATTERNS: sparsity as {nrow, ncol, colind.., row..} instead of the dense flag form]
//     [-DSTUB_SWAP: hanging_chain reads (v, x) for (x, v) in hess_psi_prod; the others (Sigma, y) for (y, Sigma)]
#include "panoc_oracle.hpp"
#include "quad_models.hpp"

#include <cmath>
#include <vector>

#ifndef STUB_PROBLEM
#define STUB_PROBLEM 0
#endif
#ifndef STUB_N
#define STUB_N 5
#endif
#ifndef STUB_WITH_SCALE
#define STUB_WITH_SCALE 1
#endif

namespace {
using namespace oracle;
#if STUB_PROBLEM == 0
using P = SingleShooting<Chain>;
constexpr int NX = Chain::nx;
#elif STUB_PROBLEM == 1
using P = SingleShootingAL<Quadcopter>;
constexpr int NX = 9;
#else
using P = SingleShootingAL<RealisticQuadcopter>;
constexpr int NX = 9;
#endif
constexpr int HORIZON = STUB_N;
int n_() { return P(HORIZON).n(); }
#if STUB_PROBLEM == 0
int m_() { return 0; }
#else
int m_() { return P(HORIZON).m(); }
#endif

int refs = 0; // incref / decref bookkeeping (the test reads it back through stub_refcount)

/// A column vector's pattern in one of CasADi's two forms.
const long long *pattern(long long nrow) {
    static std::vector<std::vector<long long>> keep;
#ifdef STUB_FULL_PATTERNS
    std::vector<long long> sp{nrow, 1, 0, nrow};
    for (long long r = 0; r < nrow; ++r)
        sp.push_back(r);
#else
    std::vector<long long> sp{nrow, 1, 1};
#endif
    for (auto &k : keep)
        if (k == sp)
            return k.data();
    keep.reserve(64); // (the addresses handed out stay valid)
    keep.push_back(sp);
    return keep.back().data();
}

bool box_is_D(const double *zl, const double *zu) {
#if STUB_PROBLEM != 0
    P p(HORIZON);
    for (int j = 0; j < p.m(); ++j)
        if (!zl || !zu || zl[j] != p.d_lb(j) || zu[j] != p.d_ub(j))
            return false;
#else
    (void)zl, (void)zu;
#endif
    return true;
}

/// The problem at parameter p with multipliers y and penalties Sigma.
P problem(const double *p, const double *y, const double *Sig) {
    P pr(HORIZON);
    for (int l = 0; l < NX; ++l)
        pr.x0[l] = p ? p[l] : 0.0;
#if STUB_PROBLEM != 0
#ifdef STUB_SWAP
    pr.set_al(Sig, y);
#else
    pr.set_al(y, Sig);
#endif
#else
    (void)y, (void)Sig;
#endif
    return pr;
}
} // namespace

#define STUB_API(name, NIN, NOUT, ...)                                                                        \
    extern "C" long long name##_n_in(void) { return NIN; }                                                    \
    extern "C" long long name##_n_out(void) { return NOUT; }                                                  \
    extern "C" int name##_work(long long *a, long long *r, long long *iw, long long *w) {                     \
        *a = NIN + 2; *r = NOUT + 1; *iw = 3; *w = 7; /* (more than needed: the loader must honour them) */   \
        return 0;                                                                                             \
    }                                                                                                         \
    extern "C" void name##_incref(void) { ++refs; }                                                           \
    extern "C" void name##_decref(void) { --refs; }                                                           \
    extern "C" int name##_checkout(void) { return 0; }                                                        \
    extern "C" void name##_release(int) {}

extern "C" int stub_refcount(void) { return refs; }

// ---- f(x, p) -> f ; g(x, p) -> g ---------------------------------------------------------------------
STUB_API(f, 2, 1)
extern "C" const long long *f_sparsity_in(long long i) { return pattern(i == 0 ? n_() : NX); }
extern "C" const long long *f_sparsity_out(long long) { return pattern(1); }
extern "C" int f(const double **arg, double **res, long long *, double *, int) {
    std::vector<double> zero(std::max(m_(), 1), 0.0), one(std::max(m_(), 1), 1.0);
    P pr = problem(arg[1], zero.data(), one.data());
#if STUB_PROBLEM == 0
    if (res[0])
        res[0][0] = pr.eval_psi(arg[0]);
#else
    if (res[0])
        res[0][0] = pr.eval_f(arg[0]);
#endif
    return 0;
}
STUB_API(g, 2, 1)
extern "C" const long long *g_sparsity_in(long long i) { return pattern(i == 0 ? n_() : NX); }
extern "C" const long long *g_sparsity_out(long long) { return pattern(m_()); }
extern "C" int g(const double **arg, double **res, long long *, double *, int) {
#if STUB_PROBLEM != 0
    std::vector<double> zero(m_(), 0.0), one(m_(), 1.0);
    P pr = problem(arg[1], zero.data(), one.data());
    if (res[0])
        pr.eval_g(arg[0], res[0]);
#else
    (void)arg, (void)res;
#endif
    return 0;
}

// ---- psi, psi_grad_psi: (x, p, y, Sigma, [s,] zl, zu) ------------------------------------------------
#if STUB_WITH_SCALE
#define PSI_NIN 7
#define ZL 5
#define ZU 6
#else
#define PSI_NIN 6
#define ZL 4
#define ZU 5
#endif
static const long long *psi_like_in(long long i) {
    if (i == 0)
        return pattern(n_());
    if (i == 1)
        return pattern(NX);
    if (STUB_WITH_SCALE && i == 4)
        return pattern(1);
    return pattern(m_());
}
static bool psi_like_args_ok(const double **arg) {
    if (STUB_WITH_SCALE && (!arg[4] || arg[4][0] != 1.0))
        return false;
    return box_is_D(arg[ZL], arg[ZU]);
}
STUB_API(psi, PSI_NIN, 2)
extern "C" const long long *psi_sparsity_in(long long i) { return psi_like_in(i); }
extern "C" const long long *psi_sparsity_out(long long j) { return pattern(j == 0 ? 1 : m_()); }
extern "C" int psi(const double **arg, double **res, long long *, double *, int) {
    if (!psi_like_args_ok(arg))
        return 1;
    P pr = problem(arg[1], arg[2], arg[3]);
    std::vector<double> yhat(std::max(m_(), 1));
#if STUB_PROBLEM == 0
    const double v = pr.eval_psi(arg[0]);
#else
    const double v = pr.eval_psi(arg[0], yhat.data());
#endif
    if (res[0])
        res[0][0] = v;
    if (res[1])
        for (int j = 0; j < m_(); ++j)
            res[1][j] = yhat[j];
    return 0;
}
STUB_API(psi_grad_psi, PSI_NIN, 2)
extern "C" const long long *psi_grad_psi_sparsity_in(long long i) { return psi_like_in(i); }
extern "C" const long long *psi_grad_psi_sparsity_out(long long j) { return pattern(j == 0 ? 1 : n_()); }
extern "C" int psi_grad_psi(const double **arg, double **res, long long *, double *, int) {
    if (!psi_like_args_ok(arg))
        return 1;
    P pr = problem(arg[1], arg[2], arg[3]);
    std::vector<double> grad(n_());
    const double v = pr.eval_psi_grad_psi(arg[0], grad.data());
    if (res[0])
        res[0][0] = v;
    if (res[1])
        for (int i = 0; i < n_(); ++i)
            res[1][i] = grad[i];
    return 0;
}

// ---- hess_psi_prod(x, p, y, Sigma, s, zl, zu, v) -> Hv ---------------------------------------------
STUB_API(hess_psi_prod, 8, 1)
extern "C" const long long *hess_psi_prod_sparsity_in(long long i) {
    return pattern(i == 0 || i == 7 ? n_() : i == 1 ? NX : i == 4 ? 1 : m_());
}
extern "C" const long long *hess_psi_prod_sparsity_out(long long) { return pattern(n_()); }
extern "C" int hess_psi_prod(const double **arg, double **res, long long *, double *, int) {
    if (!arg[4] || arg[4][0] != 1.0 || !box_is_D(arg[5], arg[6]))
        return 1;
    P pr = problem(arg[1], arg[2], arg[3]);
    if (!res[0])
        return 0;
#if STUB_PROBLEM == 0 && defined(STUB_SWAP)
    pr.eval_hess_psi_prod(arg[7], arg[0], res[0]);
#else
    pr.eval_hess_psi_prod(arg[0], arg[7], res[0]);
#endif
    return 0;
}
