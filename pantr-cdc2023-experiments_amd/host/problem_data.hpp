// Numeric problem data next to a generated problem: the data half of the reference's problem-side
// boundary.
//
//   <dir>/<name>_<formulation>_<N>.csv        written by alpaqa.casadi_generator.write_casadi_problem_data
//                                             (call: problems/generate_problem.py:62-69, keyword order
//                                             C, D, param, l1_reg, penalty_alm_split), read back by
//                                             alpaqa::CasADiProblem when the .so is loaded
//                                             (drivers/casadi-dll-wrapper.cpp:37-54, param at :64)
//   <dir>/<name>_<formulation>_<N>.guess.csv  written by generate_problem.py:71-73 (row 1: initial
//                                             guess, row 2: penalty_alm_split), read by
//                                             drivers/casadi-dll-wrapper.cpp:57-61
//
// The writer of the first file lives in the un-vendored alpaqa submodule; its row layout is restated
// here from alpaqa's published casadi_generator (one numpy.savetxt row each, comma separated, "%.18e",
// "inf"/"-inf" for unbounded sides, an EMPTY line for an empty vector):
//     C lower | C upper | D lower | D upper | param | l1_reg | penalty_alm_split
// The function half of that boundary (the CasADi-generated .so itself, or a `dl:` plugin's
// alpaqa_problem_functions_t) is host code and cannot run on the device: the problem functions are
// compiled device code selected by <name>; this file's data are checked against that model and then
// used in its place (bounds, initial state, initial guess).
#pragma once
#include "params.hpp"

#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace pantr::host {

struct ProblemData {
    std::string name;     ///< problem name (file name part of the path)
    std::string stem;     ///< <dir>/<name>_<formulation>_<N>
    bool have_csv = false;
    std::vector<double> C_lb, C_ub, D_lb, D_ub, param, l1_reg, guess;
    int32_t penalty_alm_split_csv = 0; ///< last row of the .csv (generate_problem.py:68 writes 0)
    int32_t penalty_alm_split     = 0; ///< second row of the .guess.csv (generate_problem.py:32,73)
};

/// One CSV row: comma-separated doubles; an empty (or whitespace-only) line is an empty vector.
inline bool read_csv_row(std::istream &is, std::vector<double> &row, std::string &err) {
    row.clear();
    std::string line;
    if (!std::getline(is, line)) {
        err = "unexpected end of file";
        return false;
    }
    while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t'))
        line.pop_back();
    if (line.empty())
        return true;
    size_t pos = 0;
    while (true) {
        size_t comma = line.find(',', pos);
        std::string_view cell(line.data() + pos, (comma == std::string::npos ? line.size() : comma) - pos);
        while (!cell.empty() && cell.front() == ' ')
            cell.remove_prefix(1);
        double v;
        if (!parse_double(cell, v)) {
            err = "cannot parse '" + std::string(cell) + "' as a number";
            return false;
        }
        row.push_back(v);
        if (comma == std::string::npos)
            break;
        pos = comma + 1;
    }
    return true;
}

inline bool to_count(const std::vector<double> &row, int32_t &out) {
    if (row.size() != 1 || row[0] != std::floor(row[0]) || row[0] < 0 || row[0] > 2147483647.0)
        return false;
    out = static_cast<int32_t>(row[0]);
    return true;
}

enum class LoadStatus { Ok, NoGuessFile, Malformed };

/// Reads the sidecars of `<path>_<formulation>_<horizon>` (path = [<dir>/]<name>).  The guess file
/// is mandatory, as in the reference ("Failed to open guess file", casadi-dll-wrapper.cpp:58-59);
/// the .csv is optional (alpaqa loads it only when present).
inline LoadStatus load_problem_data(const std::string &path, const std::string &formulation, int32_t horizon,
                                    ProblemData &d, std::string &err) {
    const size_t slash = path.find_last_of('/');
    d.name = slash == std::string::npos ? path : path.substr(slash + 1);
    d.stem = path + "_" + formulation + "_" + std::to_string(horizon);
    std::ifstream guess(d.stem + ".guess.csv");
    if (!guess) {
        err = "Failed to open guess file '" + d.stem + ".guess.csv'";
        return LoadStatus::NoGuessFile;
    }
    std::vector<double> row;
    std::string e;
    if (!read_csv_row(guess, d.guess, e)) {
        err = d.stem + ".guess.csv, row 1 (initial guess): " + e;
        return LoadStatus::Malformed;
    }
    if (!read_csv_row(guess, row, e) || !to_count(row, d.penalty_alm_split)) {
        err = d.stem + ".guess.csv, row 2 (penalty_alm_split): " + (e.empty() ? "expected one non-negative integer" : e);
        return LoadStatus::Malformed;
    }
    std::ifstream csv(d.stem + ".csv");
    d.have_csv = static_cast<bool>(csv);
    if (d.have_csv) {
        struct {
            std::vector<double> *dst;
            const char *what;
        } rows[] = {{&d.C_lb, "C lower bound"}, {&d.C_ub, "C upper bound"}, {&d.D_lb, "D lower bound"},
                    {&d.D_ub, "D upper bound"}, {&d.param, "param"},        {&d.l1_reg, "l1_reg"}};
        int r = 1;
        for (auto &rw : rows) {
            if (!read_csv_row(csv, *rw.dst, e)) {
                err = d.stem + ".csv, row " + std::to_string(r) + " (" + rw.what + "): " + e;
                return LoadStatus::Malformed;
            }
            ++r;
        }
        if (!read_csv_row(csv, row, e) || (!row.empty() && !to_count(row, d.penalty_alm_split_csv))) {
            err = d.stem + ".csv, row 7 (penalty_alm_split): " + (e.empty() ? "expected one non-negative integer" : e);
            return LoadStatus::Malformed;
        }
    }
    return LoadStatus::Ok;
}

/// The compiled-in model the data are checked against.
struct ModelShape {
    int32_t n, m, nx, nu, nc, nc_N;
};

/// Dimension and structure checks: sizes n / m / nx (an empty row means "not given"), box C periodic
/// in nu and box D periodic in nc (formulation.hpp:220-238 tiles U and D; the device code keeps one
/// period), lower <= upper, no l1 regularisation, penalty_alm_split within [0, m].
inline bool check_problem_data(const ProblemData &d, const ModelShape &s, std::string &err) {
    auto size_is = [&](const std::vector<double> &v, int32_t want, const char *what) {
        if (v.empty() || static_cast<int32_t>(v.size()) == want)
            return true;
        err = d.stem + ": " + what + " has " + std::to_string(v.size()) + " entries, the compiled-in problem '" +
              d.name + "' needs " + std::to_string(want);
        return false;
    };
    if (!size_is(d.guess, s.n, "the initial guess") || d.guess.empty()) {
        if (err.empty())
            err = d.stem + ".guess.csv: the initial guess is empty, the compiled-in problem '" + d.name + "' needs " +
                  std::to_string(s.n) + " entries";
        return false;
    }
    if (!size_is(d.C_lb, s.n, "box C (lower)") || !size_is(d.C_ub, s.n, "box C (upper)") ||
        !size_is(d.D_lb, s.m, "box D (lower)") || !size_is(d.D_ub, s.m, "box D (upper)") ||
        !size_is(d.param, s.nx, "param (the initial state)"))
        return false;
    if (d.C_lb.size() != d.C_ub.size() || d.D_lb.size() != d.D_ub.size()) {
        err = d.stem + ".csv: a box needs both its lower and its upper bound";
        return false;
    }
    auto periodic = [&](const std::vector<double> &v, int32_t period, const char *what) {
        for (size_t i = 0; i < v.size(); ++i)
            if (!(v[i] == v[i % period])) {
                err = d.stem + ".csv: " + what + " is not the stage box tiled over the horizon (entry " +
                      std::to_string(i) + " differs from entry " + std::to_string(i % period) + ")";
                return false;
            }
        return true;
    };
    if (!periodic(d.C_lb, s.nu, "box C (lower)") || !periodic(d.C_ub, s.nu, "box C (upper)"))
        return false;
    if (s.nc > 0 && (!periodic(d.D_lb, s.nc, "box D (lower)") || !periodic(d.D_ub, s.nc, "box D (upper)")))
        return false;
    for (size_t i = 0; i < d.C_lb.size(); ++i)
        if (!(d.C_lb[i] <= d.C_ub[i])) {
            err = d.stem + ".csv: box C is empty at entry " + std::to_string(i);
            return false;
        }
    for (size_t i = 0; i < d.D_lb.size(); ++i)
        if (!(d.D_lb[i] <= d.D_ub[i])) {
            err = d.stem + ".csv: box D is empty at entry " + std::to_string(i);
            return false;
        }
    for (double v : d.l1_reg)
        if (v != 0) {
            err = d.stem + ".csv: l1 regularisation is not supported by the device solver";
            return false;
        }
    if (d.penalty_alm_split > s.m || d.penalty_alm_split_csv > s.m) {
        err = d.stem + ": penalty_alm_split exceeds the number of constraints m = " + std::to_string(s.m);
        return false;
    }
    return true;
}

} // namespace pantr::host
