// host_problem.cpp -- native host side of the solver hook: problem -> tableau(s) -> solution.
//
// C++ restatement (double-float) of what surrounds the hot path in the reference's default
// backend, so that a caller can hand over a parsed `problem` (src/problem.lisp:45-53) and get
// back what `simplex-solver` (src/simplex.lisp:506-542) returns for an LP, without any Lisp or
// Python in between:
//   mi355x_build_tableau      = build-tableau            (src/simplex.lisp:142-328), host only
//   mi355x_simplex_solver     = simplex-solver for LPs   (build -> n-solve-tableau on the GPU)
//   mi355x_solution_*         = tableau-objective-value / tableau-variable /
//                               tableau-reduced-cost      (src/simplex.lisp:74-120)
// The solution object keeps only what the read-back needs -- objective row, RHS column, basis,
// var-mapping -- so a 400 MB tableau is never downloaded.
// Variables are identified by their index in problem-vars.  Integer variables are declined
// (MI_UNSUPPORTED -> unsupported-constraint-error): branch-and-bound stays with the reference.
#include "../../include/mi355x_simplex.h"

extern "C" void mi355x_set_last_error_(const char *msg);   // simplex_capi.hip (thread-local message)
// simplex_capi.hip (same library, not exported): a compact handle whose stored rows are PRODUCED
// chunk by chunk into pinned staging buffers and copied while later rows are being assembled
extern "C" int mi355x_tab_create_compact_streamed_(mi355x_tab **out, int64_t rows, int64_t var_count, int64_t n_stored,
                                                   const int64_t *stored_cols, const int64_t *host_basis, int device,
                                                   void (*produce)(void *ctx, int64_t r0, int64_t r1, double *dst),
                                                   void *ctx, int n_workers);

// simplex_capi.hip (same library, not exported): cancel plumbing of a two-phase job
extern "C" void mi355x_tab_link_cancel_(mi355x_tab *a, mi355x_tab *b);
extern "C" void mi355x_tab_clear_cancel_(mi355x_tab *t);

#include <algorithm>
#include <chrono>
#include <map>
#include <tuple>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

enum MapKind { kPositive = 0, kNegative = 1, kSigned = 2 };   // var-mapping-entry, simplex.lisp:44-46

struct Mapping {
    int     kind = kPositive;
    int64_t col = 0;
    double  offset = 0.0;
};

struct Constraint {
    int op;                                   // 0: <=   1: >=   2: =
    std::vector<int64_t> var;
    std::vector<double>  coef;
    double rhs;
};

struct Bound {
    bool present = false;                     // an entry exists in problem-var-bounds
    bool has_lb = false, has_ub = false;
    double lb = 0.0, ub = 0.0;
};

}  // namespace

struct mi355x_problem {
    bool    is_max = true;
    int64_t n_vars = 0;
    std::vector<int64_t> obj_var;             // objective-func alist, in insertion order
    std::vector<double>  obj_coef;
    std::vector<Bound>   bounds;
    std::vector<char>    is_integer;
    std::vector<Constraint> constraints;
};

namespace {

struct HostTableau {
    int64_t rows = 0, cols = 0;               // rows = constraint-count + 1, cols = var-count + 1
    std::vector<double>  M;                   // row-major, tight
    std::vector<int64_t> basis;
    double &at(int64_t r, int64_t c) { return M[(size_t)r * cols + c]; }
};

struct Built {
    bool two_phase = false;
    HostTableau main_tab, art;
    std::vector<Mapping> map;
    int status = MI_OK;                       // MI_UNBOUNDED from the no-constraint special case
};

// build-tableau, src/simplex.lisp:142-328 (line numbers in the comments below)
Built build(const mi355x_problem &p)
{
    Built b;
    const int64_t n = p.n_vars;
    b.map.resize((size_t)n);
    // constraints in tableau order: the x <= ub rows pushed for doubly-bounded variables
    // (:198-203) come first, the problem's own rows follow -- by pointer, nothing is copied
    std::vector<Constraint> pushed;
    std::vector<const Constraint *> cons;

    if (p.constraints.empty()) {                                          // :153-186
        HostTableau &t = b.main_tab;
        t.rows = n + 1; t.cols = n + 1;
        t.M.assign((size_t)t.rows * t.cols, 0.0);
        t.basis.resize((size_t)n);
        std::vector<double> coef((size_t)n, 0.0);
        std::vector<char> has_coef((size_t)n, 0);
        for (size_t k = 0; k < p.obj_var.size(); ++k)
            if (!has_coef[(size_t)p.obj_var[k]]) { coef[(size_t)p.obj_var[k]] = p.obj_coef[k]; has_coef[(size_t)p.obj_var[k]] = 1; }
        double objective = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            const Bound &bd = p.bounds[(size_t)i];
            t.basis[(size_t)i] = i;
            t.at(i, i) = 1.0;
            const bool use_ub = ((0.0 <= coef[(size_t)i]) == p.is_max);
            if (use_ub) {
                if (!(bd.present && bd.has_ub)) { b.status = MI_UNBOUNDED; return b; }
                b.map[(size_t)i] = {kPositive, i, bd.ub};
                objective = objective + coef[(size_t)i] * bd.ub;
            } else {
                if (!(bd.present && bd.has_lb)) { b.status = MI_UNBOUNDED; return b; }
                b.map[(size_t)i] = {kPositive, i, bd.lb};
                objective = objective + coef[(size_t)i] * bd.lb;
            }
        }
        t.at(n, n) = objective;
        return b;
    }

    int64_t ncv = n, column = 0;                                           // :189-212
    for (int64_t v = 0; v < n; ++v) {
        const Bound &bd = p.bounds[(size_t)v];
        if (!bd.present) {
            b.map[(size_t)v] = {kPositive, column, 0.0};
        } else if (bd.has_lb && bd.has_ub) {
            Constraint c;
            if (0.0 <= bd.ub) { c.op = 0; c.rhs = bd.ub; } else { c.op = 1; c.rhs = -bd.ub; }
            c.var = {v}; c.coef = {1.0};
            pushed.insert(pushed.begin(), c);                              // (push ... constraints)
            b.map[(size_t)v] = {kPositive, column, bd.lb};
        } else if (bd.has_lb) {
            b.map[(size_t)v] = {kPositive, column, bd.lb};
        } else if (bd.has_ub) {
            b.map[(size_t)v] = {kNegative, column, bd.ub};
        } else {
            b.map[(size_t)v] = {kSigned, column, 0.0};
            ++column; ++ncv;
        }
        ++column;
    }

    for (const auto &c : pushed) cons.push_back(&c);
    for (const auto &c : p.constraints) cons.push_back(&c);
    const int64_t m = (int64_t)cons.size();                                // :214-221
    int64_t num_slack = 0;
    for (const Constraint *c : cons) if (c->op != 2) ++num_slack;
    const int64_t num_cols = ncv + num_slack + 1;
    HostTableau &t = b.main_tab;
    t.rows = m + 1; t.cols = num_cols;
    t.M.assign((size_t)t.rows * t.cols, 0.0);
    t.basis.assign((size_t)m, 0);
    std::vector<int64_t> art_rows;                                         // most recent first
    int64_t col_offset = 0;
    for (int64_t row = 0; row < m; ++row) {                                // :223-268
        const Constraint &c = *cons[(size_t)row];
        int op = c.op;
        t.at(row, num_cols - 1) = c.rhs;
        for (size_t k = 0; k < c.var.size(); ++k) {
            const Mapping &mp = b.map[(size_t)c.var[k]];
            const double coef = c.coef[k];
            if (mp.kind == kPositive) {
                t.at(row, mp.col) = coef;
                t.at(row, num_cols - 1) = t.at(row, num_cols - 1) - coef * mp.offset;
            } else if (mp.kind == kNegative) {
                t.at(row, mp.col) = -coef;
                t.at(row, num_cols - 1) = t.at(row, num_cols - 1) - coef * mp.offset;
            } else {
                t.at(row, mp.col) = coef;
                t.at(row, mp.col + 1) = -coef;
            }
        }
        if (t.at(row, num_cols - 1) < 0.0) {                               // :243-252
            for (int64_t cc = 0; cc < num_cols; ++cc) t.at(row, cc) = -t.at(row, cc);
            op = (op == 0) ? 1 : (op == 1) ? 0 : 2;
        }
        if (op == 0) {                                                     // :254-265
            t.at(row, ncv + col_offset) = 1.0;
            t.basis[(size_t)row] = ncv + col_offset;
            ++col_offset;
        } else if (op == 1) {
            art_rows.insert(art_rows.begin(), row);
            t.at(row, ncv + col_offset) = -1.0;
            t.basis[(size_t)row] = num_cols;
            ++col_offset;
        } else {
            art_rows.insert(art_rows.begin(), row);
            t.basis[(size_t)row] = num_cols;
        }
    }
    for (size_t k = 0; k < p.obj_var.size(); ++k) {                        // :270-283
        const Mapping &mp = b.map[(size_t)p.obj_var[k]];
        const double coef = p.obj_coef[k];
        if (mp.kind == kPositive) {
            t.at(m, mp.col) = -coef;
            t.at(m, num_cols - 1) = t.at(m, num_cols - 1) + coef * mp.offset;
        } else if (mp.kind == kNegative) {
            t.at(m, mp.col) = coef;
            t.at(m, num_cols - 1) = t.at(m, num_cols - 1) + coef * mp.offset;
        } else {
            t.at(m, mp.col) = -coef;
            t.at(m, mp.col + 1) = coef;
        }
    }
    if (art_rows.empty()) return b;

    b.two_phase = true;                                                    // :292-325
    const int64_t num_art = (int64_t)art_rows.size();
    const int64_t nac = num_cols + num_art;
    HostTableau &a = b.art;
    a.rows = m + 1; a.cols = nac;
    a.M.assign((size_t)a.rows * a.cols, 0.0);
    a.basis = t.basis;
    std::vector<char> is_art((size_t)m, 0);
    for (int64_t i = 0; i < num_art; ++i) {
        const int64_t row = art_rows[(size_t)i];
        is_art[(size_t)row] = 1;
        a.basis[(size_t)row] = num_cols - 1 + i;
        a.at(row, num_cols - 1 + i) = 1.0;
    }
    for (int64_t c = 0; c < num_cols - 1; ++c) {
        double s = 0.0;
        for (int64_t r = 0; r < m; ++r) {
            a.at(r, c) = t.at(r, c);
            if (is_art[(size_t)r]) s = s + a.at(r, c);
        }
        a.at(m, c) = s;
    }
    {
        double s = 0.0;
        for (int64_t r = 0; r < m; ++r) {
            a.at(r, nac - 1) = t.at(r, num_cols - 1);
            if (is_art[(size_t)r]) s = s + a.at(r, nac - 1);
        }
        a.at(m, nac - 1) = s;
    }
    return b;
}

int hfail(int code, const char *msg) { mi355x_set_last_error_(msg); return code; }

}  // namespace

// Single-phase problems (every row ends up a `<=` row: all slack columns basic, no artificial
// tableau) assembled straight into the compact form [structural columns | RHS] that the solve
// loop runs on -- the slack identity block is never materialised, on the host or on the device.
// Same arithmetic as build() entry by entry (rows are independent, so they are filled by a few
// host threads).  Returns false when the problem needs the general path.
struct CompactPlan {
    const mi355x_problem *p = nullptr;
    std::vector<Constraint> pushed;
    std::vector<const Constraint *> cons;
    std::vector<Mapping> map;
    std::vector<int64_t> basis;
    int64_t m = 0, ncv = 0;
};

static bool plan_compact(const mi355x_problem &p, CompactPlan &pl)
{
    const int64_t n = p.n_vars;
    if (p.constraints.empty()) return false;
    pl.p = &p;
    std::vector<Mapping> &map = pl.map;
    map.assign((size_t)n, Mapping());
    int64_t ncv = n, column = 0;                                           // :189-212
    for (int64_t v = 0; v < n; ++v) {
        const Bound &bd = p.bounds[(size_t)v];
        if (!bd.present) {
            map[(size_t)v] = {kPositive, column, 0.0};
        } else if (bd.has_lb && bd.has_ub) {
            Constraint c;
            if (0.0 <= bd.ub) { c.op = 0; c.rhs = bd.ub; } else { c.op = 1; c.rhs = -bd.ub; }
            c.var = {v}; c.coef = {1.0};
            pl.pushed.insert(pl.pushed.begin(), c);
            map[(size_t)v] = {kPositive, column, bd.lb};
        } else if (bd.has_lb) {
            map[(size_t)v] = {kPositive, column, bd.lb};
        } else if (bd.has_ub) {
            map[(size_t)v] = {kNegative, column, bd.ub};
        } else {
            map[(size_t)v] = {kSigned, column, 0.0};
            ++column; ++ncv;
        }
        ++column;
    }
    for (const auto &c : pl.pushed) pl.cons.push_back(&c);
    for (const auto &c : p.constraints) pl.cons.push_back(&c);
    const int64_t m = (int64_t)pl.cons.size();
    // does any row become >= or = (after the sign flip of a negative shifted RHS)?  Every
    // coefficient of the problem is read once for this (537 MB at config 3: 32 ms on one thread of
    // a 140 ms solve), so large problems are checked by a few threads, rows interleaved
    auto rows_stay_le = [&](int64_t first, int64_t step) {
        for (int64_t r = first; r < m; r += step) {
            const Constraint *c = pl.cons[(size_t)r];
            double rhs = c->rhs;
            for (size_t k = 0; k < c->var.size(); ++k) {
                const Mapping &mp = map[(size_t)c->var[k]];
                if (mp.kind != kSigned) rhs = rhs - c->coef[k] * mp.offset;
            }
            const int op = (rhs < 0.0) ? (c->op == 0 ? 1 : c->op == 1 ? 0 : 2) : c->op;
            if (op != 0) return false;
        }
        return true;
    };
    int64_t pairs = 0;
    for (const Constraint *c : pl.cons) pairs += (int64_t)c->var.size();
    const int nthr = pairs > (1 << 22) ? (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency())) : 1;
    if (nthr == 1) {
        if (!rows_stay_le(0, 1)) return false;
    } else {
        std::vector<char> ok((size_t)nthr, 1);
        std::vector<std::thread> pool;
        for (int k = 1; k < nthr; ++k) pool.emplace_back([&, k]() { ok[(size_t)k] = rows_stay_le(k, nthr) ? 1 : 0; });
        ok[0] = rows_stay_le(0, nthr) ? 1 : 0;
        for (auto &th : pool) th.join();
        for (char f : ok) if (!f) return false;
    }
    pl.m = m; pl.ncv = ncv;
    pl.basis.resize((size_t)m);
    for (int64_t r = 0; r < m; ++r) pl.basis[(size_t)r] = ncv + r;         // every row's slack column
    return true;
}

// rows [r0, r1) of [structural columns | RHS] (row m = the objective row) into dst, tightly packed;
// the same arithmetic as build() entry by entry -- rows are independent
static void produce_compact_rows(void *ctx, int64_t r0, int64_t r1, double *dst)
{
    const CompactPlan &pl = *static_cast<const CompactPlan *>(ctx);
    const int64_t ncv = pl.ncv, w = ncv + 1, m = pl.m;
    for (int64_t row = r0; row < r1; ++row) {
        double *out = dst + (size_t)(row - r0) * w;
        std::fill(out, out + w, 0.0);
        if (row == m) {                                                    // :270-283
            const mi355x_problem &p = *pl.p;
            for (size_t k = 0; k < p.obj_var.size(); ++k) {
                const Mapping &mp = pl.map[(size_t)p.obj_var[k]];
                const double coef = p.obj_coef[k];
                if (mp.kind == kPositive)      { out[mp.col] = -coef; out[ncv] = out[ncv] + coef * mp.offset; }
                else if (mp.kind == kNegative) { out[mp.col] = coef;  out[ncv] = out[ncv] + coef * mp.offset; }
                else                           { out[mp.col] = -coef; out[mp.col + 1] = coef; }
            }
            continue;
        }
        const Constraint &c = *pl.cons[(size_t)row];                       // :223-268
        out[ncv] = c.rhs;
        for (size_t k = 0; k < c.var.size(); ++k) {
            const Mapping &mp = pl.map[(size_t)c.var[k]];
            const double coef = c.coef[k];
            if (mp.kind == kPositive)      { out[mp.col] = coef;  out[ncv] = out[ncv] - coef * mp.offset; }
            else if (mp.kind == kNegative) { out[mp.col] = -coef; out[ncv] = out[ncv] - coef * mp.offset; }
            else                           { out[mp.col] = coef;  out[mp.col + 1] = -coef; }
        }
        if (out[ncv] < 0.0)
            for (int64_t cc = 0; cc < w; ++cc) out[cc] = -out[cc];
    }
}

struct mi355x_solution {
    int64_t rows = 0, cols = 0;
    std::vector<double>  last_row;            // objective row
    std::vector<double>  last_col;            // RHS column
    std::vector<int64_t> basis;
    std::vector<Mapping> map;
    int64_t n_pivots[2] = {0, 0};
};

struct mi355x_solve {
    enum State { kPhase1, kHandover, kPhase2, kDone };
    mi355x_tab *mt = nullptr, *at = nullptr;          // main / artificial tableau on the device
    mi355x_solution *sol = nullptr;                   // sized at begin, filled at finish
    State  state = kPhase2;
    int    status = MI_RUNNING;                       // the final status once state == kDone
    int    is_max = 1;
    double f = 1024.0;
    bool   timing = false;                            // MI355X_E2E_TIMING=1: where the time goes, on stderr
    ~mi355x_solve()
    {
        mi355x_tab_link_cancel_(at, nullptr);
        mi355x_tab_destroy(at);
        mi355x_tab_destroy(mt);
        delete sol;
    }
};

extern "C" {

int mi355x_problem_create(mi355x_problem **out, int is_max, int64_t n_vars)
{
    if (!out || n_vars < 1) return hfail(MI_BAD_ARG, "bad arguments");
    mi355x_problem *p = new (std::nothrow) mi355x_problem;
    if (!p) return hfail(MI_NO_MEMORY, "host allocation failed");
    p->is_max = is_max != 0;
    p->n_vars = n_vars;
    p->bounds.resize((size_t)n_vars);
    p->is_integer.assign((size_t)n_vars, 0);
    *out = p;
    return MI_OK;
}

void mi355x_problem_destroy(mi355x_problem *p) { delete p; }

static int check_vars(const mi355x_problem *p, const int64_t *var, int64_t nnz)
{
    for (int64_t k = 0; k < nnz; ++k)
        if (var[k] < 0 || var[k] >= p->n_vars) return 0;
    return 1;
}

int mi355x_problem_set_objective(mi355x_problem *p, const int64_t *var, const double *coef, int64_t nnz)
{
    if (!p || nnz < 0 || (nnz && (!var || !coef)) || !check_vars(p, var, nnz))
        return hfail(MI_BAD_ARG, "bad objective");
    p->obj_var.assign(var, var + nnz);
    p->obj_coef.assign(coef, coef + nnz);
    return MI_OK;
}

int mi355x_problem_set_bounds(mi355x_problem *p, int64_t var, int has_lb, double lb, int has_ub, double ub)
{
    if (!p || var < 0 || var >= p->n_vars) return hfail(MI_BAD_ARG, "bad variable index");
    if (has_lb && has_ub && ub < lb) return hfail(MI_BAD_ARG, "invalid bounds: upper < lower");   // invalid-bounds-error
    Bound &b = p->bounds[(size_t)var];
    b.present = true; b.has_lb = has_lb != 0; b.has_ub = has_ub != 0; b.lb = lb; b.ub = ub;
    return MI_OK;
}

int mi355x_problem_set_integer(mi355x_problem *p, int64_t var)
{
    if (!p || var < 0 || var >= p->n_vars) return hfail(MI_BAD_ARG, "bad variable index");
    p->is_integer[(size_t)var] = 1;
    return MI_OK;
}

int mi355x_problem_add_constraint(mi355x_problem *p, int op, const int64_t *var, const double *coef,
                                  int64_t nnz, double rhs)
{
    if (!p || nnz < 0 || (nnz && (!var || !coef)) || !check_vars(p, var, nnz))
        return hfail(MI_BAD_ARG, "bad constraint");
    if (op < 0 || op > 2) return hfail(MI_BAD_ARG, "not a valid constraint equation");   // parsing-error, :266
    Constraint c;
    c.op = op; c.rhs = rhs;
    c.var.assign(var, var + nnz);
    c.coef.assign(coef, coef + nnz);
    p->constraints.push_back(std::move(c));
    return MI_OK;
}

// The parsed problem as JSON (for inspection / tests): returns the length needed (excluding the
// terminating NUL); writes at most cap-1 characters.
int64_t mi355x_problem_to_json(const mi355x_problem *p, char *buf, int64_t cap)
{
    if (!p) return hfail(MI_BAD_ARG, "problem is NULL");
    std::string s = "{\"type\": \"";
    s += p->is_max ? "max" : "min";
    s += "\", \"n_vars\": " + std::to_string(p->n_vars) + ", \"objective\": [";
    char num[64];
    auto fmt = [&](double x) { snprintf(num, sizeof num, "%.17g", x); return std::string(num); };
    for (size_t k = 0; k < p->obj_var.size(); ++k)
        s += (k ? ", [" : "[") + std::to_string(p->obj_var[k]) + ", " + fmt(p->obj_coef[k]) + "]";
    s += "], \"integer\": [";
    bool first = true;
    for (int64_t v = 0; v < p->n_vars; ++v)
        if (p->is_integer[(size_t)v]) { s += (first ? "" : ", ") + std::to_string(v); first = false; }
    s += "], \"bounds\": [";
    first = true;
    for (int64_t v = 0; v < p->n_vars; ++v) {
        const Bound &b = p->bounds[(size_t)v];
        if (!b.present) continue;
        s += std::string(first ? "" : ", ") + "[" + std::to_string(v) + ", " +
             (b.has_lb ? fmt(b.lb) : "null") + ", " + (b.has_ub ? fmt(b.ub) : "null") + "]";
        first = false;
    }
    s += "], \"constraints\": [";
    static const char *ops[] = {"<=", ">=", "="};
    for (size_t k = 0; k < p->constraints.size(); ++k) {
        const Constraint &c = p->constraints[k];
        s += std::string(k ? ", " : "") + "[\"" + ops[c.op] + "\", [";
        for (size_t j = 0; j < c.var.size(); ++j)
            s += (j ? ", [" : "[") + std::to_string(c.var[j]) + ", " + fmt(c.coef[j]) + "]";
        s += "], " + fmt(c.rhs) + "]";
    }
    s += "]}";
    if (buf && cap > 0) {
        const size_t n = std::min<size_t>(s.size(), (size_t)cap - 1);
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return (int64_t)s.size();
}

int mi355x_build_tableau(const mi355x_problem *p, int which, int64_t *rows, int64_t *cols,
                         double *matrix, int64_t *basis, int *two_phase)
{
    if (!p) return hfail(MI_BAD_ARG, "problem is NULL");
    const Built b = build(*p);
    if (b.status != MI_OK) return b.status;
    if (two_phase) *two_phase = b.two_phase ? 1 : 0;
    if (which != 0 && !b.two_phase) return hfail(MI_BAD_ARG, "no artificial tableau for this problem");
    const HostTableau &t = which ? b.art : b.main_tab;
    if (rows) *rows = t.rows;
    if (cols) *cols = t.cols;
    if (matrix) std::memcpy(matrix, t.M.data(), t.M.size() * sizeof(double));
    if (basis && !t.basis.empty()) std::memcpy(basis, t.basis.data(), t.basis.size() * sizeof(int64_t));
    return MI_OK;
}

int mi355x_var_mapping(const mi355x_problem *p, int64_t var, int *kind, int64_t *col, double *offset)
{
    if (!p || var < 0 || var >= p->n_vars) return hfail(MI_BAD_ARG, "bad variable index");
    const Built b = build(*p);
    if (b.status != MI_OK) return b.status;
    if (kind) *kind = b.map[(size_t)var].kind;
    if (col) *col = b.map[(size_t)var].col;
    if (offset) *offset = b.map[(size_t)var].offset;
    return MI_OK;
}

// ---- simplex-solver for LPs as a resumable job: begin (build + upload), step (bounded chunks of
// n-solve-tableau across the phases), finish (light read-back).  mi355x_simplex_solver is the three
// in a row.
int mi355x_simplex_solver_begin(const mi355x_problem *p, double fp_tolerance, int device, mi355x_solve **out)
{
    if (!p || !out) return hfail(MI_BAD_ARG, "NULL argument");
    *out = nullptr;
    for (char f : p->is_integer)
        if (f) return hfail(MI_UNSUPPORTED, "integer constraints cannot be handled by the mi355x-simplex solver");
    std::unique_ptr<mi355x_solve> job(new (std::nothrow) mi355x_solve);
    std::unique_ptr<mi355x_solution> s(new (std::nothrow) mi355x_solution);
    if (!job || !s) return hfail(MI_NO_MEMORY, "host allocation failed");
    job->is_max = p->is_max ? 1 : 0;
    job->f = fp_tolerance;
    job->timing = getenv("MI355X_E2E_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    {   // single-phase problems: the compact form [structural columns | RHS] assembled by a few
        // host threads straight into pinned staging buffers and uploaded while later rows are still
        // being assembled (mi355x_tab_create_compact_streamed_)
        CompactPlan pl;
        if (plan_compact(*p, pl)) {
            const int64_t m = pl.m, rows = m + 1, ncv = pl.ncv, var_count = ncv + m;
            s->rows = rows; s->cols = var_count + 1;
            s->map = pl.map;
            s->last_row.resize((size_t)var_count + 1);
            s->last_col.resize((size_t)rows);
            s->basis.resize((size_t)m);
            std::vector<int64_t> stored((size_t)ncv);
            for (int64_t j = 0; j < ncv; ++j) stored[(size_t)j] = j;
            const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
            const int workers = (rows * (ncv + 1) > (1 << 22)) ? (int)hw : 1;
            const auto t1 = std::chrono::steady_clock::now();
            const int rc = mi355x_tab_create_compact_streamed_(&job->mt, rows, var_count, ncv, stored.data(), pl.basis.data(),
                                                               device, produce_compact_rows, &pl, workers);
            if (job->timing)
                fprintf(stderr, "mi355x_simplex_solver: plan %.1f ms, allocate + assemble + upload (%d workers) %.1f ms\n",
                        std::chrono::duration<double, std::milli>(t1 - t0).count(), workers,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
            if (rc != MI_OK) return rc;
            job->state = mi355x_solve::kPhase2;
            job->sol = s.release();
            *out = job.release();
            return MI_OK;
        }
    }
    Built b = build(*p);
    if (b.status != MI_OK) return b.status;
    HostTableau &t = b.main_tab;
    s->rows = t.rows; s->cols = t.cols;
    s->map = b.map;
    s->last_row.resize((size_t)t.cols);
    s->last_col.resize((size_t)t.rows);
    s->basis.resize((size_t)std::max<int64_t>(t.rows - 1, 0));
    int rc = mi355x_tab_create(&job->mt, t.rows, t.cols, t.M.data(), t.basis.empty() ? nullptr : t.basis.data(), device);
    if (rc == MI_OK && b.two_phase)
        rc = mi355x_tab_create(&job->at, b.art.rows, b.art.cols, b.art.M.data(), b.art.basis.data(), device);
    if (rc != MI_OK) return rc;                      // (the job's destructor releases what was created)
    if (b.two_phase) mi355x_tab_link_cancel_(job->at, job->mt);
    job->state = b.two_phase ? mi355x_solve::kPhase1 : mi355x_solve::kPhase2;
    job->sol = s.release();
    *out = job.release();
    return MI_OK;
}

int mi355x_simplex_solver_step(mi355x_solve *job, int64_t max_pivots, int64_t *n_pivots)
{
    if (n_pivots) *n_pivots = 0;
    if (!job) return hfail(MI_BAD_ARG, "job is NULL");
    if (max_pivots < 0) return hfail(MI_BAD_ARG, "max_pivots < 0");
    if (job->state == mi355x_solve::kDone) return job->status;
    int64_t done = 0;
    // whichever way this step ends, a cancel request aimed at it ends with it
    struct Clear { mi355x_solve *j; ~Clear() { mi355x_tab_clear_cancel_(j->at); mi355x_tab_clear_cancel_(j->mt); } } clear{job};
    auto finished = [&](int rc) { job->state = mi355x_solve::kDone; job->status = rc; if (n_pivots) *n_pivots = done; return rc; };
    auto paused = [&](int rc) { if (n_pivots) *n_pivots = done; return rc; };
    const auto t0 = std::chrono::steady_clock::now();
    if (job->state == mi355x_solve::kPhase1) {                              // simplex.lisp:403
        int64_t k = 0;
        const int rc = mi355x_tab_solve(job->at, /*is_max=*/0, job->f, max_pivots, &k);
        done += k; job->sol->n_pivots[0] += k;
        if (rc == MI_MAX_PIVOTS || rc == MI_CANCELLED) return paused(rc);
        if (rc != MI_OPTIMAL) return finished(rc);
        job->state = mi355x_solve::kHandover;
    }
    if (job->state == mi355x_solve::kHandover) {                            // simplex.lisp:405-451
        int64_t nd = 0;
        const int rc = mi355x_two_phase_handover(job->at, job->mt, job->f, &nd);
        done += nd; job->sol->n_pivots[0] += nd;
        if (rc != MI_OK) return finished(rc);
        job->state = mi355x_solve::kPhase2;
        if (max_pivots && done >= max_pivots) return paused(MI_MAX_PIVOTS);
    }
    int64_t k = 0;                                                          // simplex.lisp:452 / 453-461
    const int rc = mi355x_tab_solve(job->mt, job->is_max, job->f, max_pivots ? max_pivots - done : 0, &k);
    done += k; job->sol->n_pivots[1] += k;
    if (job->timing)
        fprintf(stderr, "mi355x_simplex_solver: step %.1f ms (%lld pivots, status %d)\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), (long long)done, rc);
    if (rc == MI_MAX_PIVOTS || rc == MI_CANCELLED) return paused(rc);
    return finished(rc);
}

int mi355x_simplex_solver_cancel(mi355x_solve *job)
{
    if (!job) return hfail(MI_BAD_ARG, "job is NULL");
    // (the two tableaux of a two-phase job look at each other's flag: one request reaches whichever
    // phase is running)
    return mi355x_tab_cancel(job->mt);
}

void mi355x_simplex_solver_abandon(mi355x_solve *job) { delete job; }

int mi355x_simplex_solver_finish(mi355x_solve *job, mi355x_solution **out)
{
    if (out) *out = nullptr;
    if (!job) return hfail(MI_BAD_ARG, "job is NULL");
    std::unique_ptr<mi355x_solve> owner(job);                               // consumed whatever happens
    if (!out) return hfail(MI_BAD_ARG, "out is NULL");
    if (job->state != mi355x_solve::kDone || job->status != MI_OPTIMAL)
        return hfail(MI_BAD_ARG, "the job has not ended with MI_OPTIMAL: there is no solution to read");
    mi355x_solution *s = job->sol;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = mi355x_tab_download(job->mt, nullptr, s->basis.empty() ? nullptr : s->basis.data(),
                                       s->last_row.data(), s->last_col.data());
    const auto t1 = std::chrono::steady_clock::now();
    if (rc != MI_OK) return rc;
    job->sol = nullptr;
    mi355x_tab_destroy(job->at); job->at = nullptr;
    mi355x_tab_destroy(job->mt); job->mt = nullptr;
    if (job->timing)
        fprintf(stderr, "mi355x_simplex_solver: read-back %.1f ms, destroy %.1f ms\n",
                std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    *out = s;
    return MI_OPTIMAL;
}

int mi355x_simplex_solver(const mi355x_problem *p, double fp_tolerance, int device,
                          mi355x_solution **out)
{
    if (!p || !out) return hfail(MI_BAD_ARG, "NULL argument");
    *out = nullptr;
    mi355x_solve *job = nullptr;
    int rc = mi355x_simplex_solver_begin(p, fp_tolerance, device, &job);
    if (rc != MI_OK) return rc;
    rc = mi355x_simplex_solver_step(job, 0, nullptr);
    if (rc != MI_OPTIMAL) { mi355x_simplex_solver_abandon(job); return rc; }
    return mi355x_simplex_solver_finish(job, out);
}

// ---- a LIST of problems (the glue's mi355x-solve-problems, natively) ---------------------------
// The hook takes one problem per call (src/solver.lisp:53-56); N small LPs solved one after the other
// leave the GPU almost empty (BASELINE config 4).  Here the library itself groups the members: same
// tableau shape and sense -> ONE multi-device batch (mi355x_multibatch_*; single-phase groups one batch,
// two-phase groups a pair of batches with the per-member step between the phases on the devices); a
// member alone in its group is an ordinary solver job.  Everything is stepped in bounded chunks, so the
// host language never sits in an unbounded foreign call.  Results are those of mi355x_simplex_solver on
// every member, bit for bit.
struct ManyUnit {
    std::vector<int64_t> members;                     // indices into the list
    mi355x_solve *single = nullptr;                   // a member alone in its group
    mi355x_multibatch *main_mb = nullptr, *art_mb = nullptr;
    int    is_max = 1;
    int    phase = 2;                                 // 1: phase 1 running (art_mb), 2: phase 2 / single phase, 3: done
    std::vector<int32_t> st1;                         // statuses phase 1 ended with (two-phase groups)
    std::vector<int32_t> between;                     // MI_OK or the member's final outcome after the hand-over
    std::vector<int64_t> np1, np2;
};
struct mi355x_solve_many {
    int64_t n = 0;
    double  f = 1024.0;
    std::vector<int32_t> status;                      // per member: MI_RUNNING until decided
    std::vector<std::unique_ptr<mi355x_solution>> sol;
    std::vector<ManyUnit> units;
    ~mi355x_solve_many()
    {
        for (ManyUnit &u : units) {
            delete u.single;
            mi355x_multibatch_destroy(u.art_mb);
            mi355x_multibatch_destroy(u.main_mb);
        }
    }
};

int mi355x_simplex_solver_many_begin(const mi355x_problem *const *problems, int64_t n, double fp_tolerance,
                                     int n_devices, const int *device_ids, mi355x_solve_many **out)
{
    if (!problems || n < 1 || !out || n_devices < 1) return hfail(MI_BAD_ARG, "bad arguments");
    *out = nullptr;
    for (int64_t k = 0; k < n; ++k) if (!problems[k]) return hfail(MI_BAD_ARG, "a problem of the list is NULL");
    std::unique_ptr<mi355x_solve_many> job(new (std::nothrow) mi355x_solve_many);
    if (!job) return hfail(MI_NO_MEMORY, "host allocation failed");
    job->n = n; job->f = fp_tolerance;
    job->status.assign((size_t)n, MI_RUNNING);
    job->sol.resize((size_t)n);
    std::vector<Built> built((size_t)n);
    // group key -> unit index
    struct Key { int two; int64_t r, c, mc; int is_max; bool operator<(const Key &o) const {
        return std::tie(two, r, c, mc, is_max) < std::tie(o.two, o.r, o.c, o.mc, o.is_max); } };
    std::map<Key, size_t> groups;
    for (int64_t k = 0; k < n; ++k) {
        const mi355x_problem &p = *problems[k];
        bool integer = false;
        for (char fl : p.is_integer) integer |= fl != 0;
        if (integer) { job->status[(size_t)k] = MI_UNSUPPORTED; continue; }
        built[(size_t)k] = build(p);
        Built &b = built[(size_t)k];
        if (b.status != MI_OK) { job->status[(size_t)k] = b.status; continue; }      // the unbounded no-constraint case
        if (b.main_tab.rows < 2) {                                                   // no constraint rows: nothing to batch
            groups[Key{2, k, 0, 0, 0}] = job->units.size();
            job->units.push_back(ManyUnit());
            job->units.back().members.push_back(k);
            continue;
        }
        const Key key = b.two_phase ? Key{1, b.art.rows, b.art.cols, b.main_tab.cols, p.is_max ? 1 : 0}
                                    : Key{0, b.main_tab.rows, b.main_tab.cols, 0, p.is_max ? 1 : 0};
        auto it = groups.find(key);
        if (it == groups.end()) { it = groups.emplace(key, job->units.size()).first; job->units.push_back(ManyUnit()); }
        job->units[it->second].members.push_back(k);
    }
    int64_t n_singles = 0;
    const int n_visible = std::max(1, mi355x_device_count());
    for (ManyUnit &u : job->units) {
        const int64_t g = (int64_t)u.members.size();
        const Built &b0 = built[(size_t)u.members[0]];
        u.is_max = problems[u.members[0]]->is_max ? 1 : 0;
        for (int64_t k : u.members) {
            std::unique_ptr<mi355x_solution> s(new (std::nothrow) mi355x_solution);
            if (!s) return hfail(MI_NO_MEMORY, "host allocation failed");
            const HostTableau &t = built[(size_t)k].main_tab;
            s->rows = t.rows; s->cols = t.cols; s->map = built[(size_t)k].map;
            s->last_row.resize((size_t)t.cols); s->last_col.resize((size_t)t.rows);
            s->basis.resize((size_t)std::max<int64_t>(t.rows - 1, 0));
            job->sol[(size_t)k] = std::move(s);
        }
        if (g == 1) {                                  // alone in its group: the one-problem job
            // (members alone in their group: dealt round-robin over the devices -- they all sat on the first one)
            const int dev = device_ids ? device_ids[n_singles % n_devices] : (int)(n_singles % std::min(n_devices, n_visible));
            ++n_singles;
            const int rc = mi355x_simplex_solver_begin(problems[u.members[0]], fp_tolerance, dev, &u.single);
            if (rc != MI_OK) return rc;
            continue;
        }
        auto pack = [&](bool art, mi355x_multibatch **mb) -> int {
            const HostTableau &t0 = art ? b0.art : b0.main_tab;
            std::vector<double>  M((size_t)(g * t0.rows * t0.cols));
            std::vector<int64_t> B((size_t)(g * (t0.rows - 1)));
            for (int64_t q = 0; q < g; ++q) {
                const Built &b = built[(size_t)u.members[(size_t)q]];
                const HostTableau &t = art ? b.art : b.main_tab;
                std::memcpy(M.data() + (size_t)q * t0.rows * t0.cols, t.M.data(), t.M.size() * sizeof(double));
                std::memcpy(B.data() + (size_t)q * (t0.rows - 1), t.basis.data(), t.basis.size() * sizeof(int64_t));
            }
            return mi355x_multibatch_create(mb, g, t0.rows, t0.cols, M.data(), B.data(), n_devices, device_ids);
        };
        int rc = pack(false, &u.main_mb);
        if (rc == MI_OK && b0.two_phase) { rc = pack(true, &u.art_mb); u.phase = 1; }
        if (rc != MI_OK) return rc;
        u.st1.assign((size_t)g, MI_RUNNING); u.between.assign((size_t)g, MI_OK);
        u.np1.assign((size_t)g, 0); u.np2.assign((size_t)g, 0);
    }
    *out = job.release();
    return MI_OK;
}

int mi355x_simplex_solver_many_step(mi355x_solve_many *job, int64_t max_pivots, int32_t *status)
{
    if (!job) return hfail(MI_BAD_ARG, "job is NULL");
    if (max_pivots < 0) return hfail(MI_BAD_ARG, "max_pivots < 0");
    bool running = false;
    // A unit that fails (a negative code: device error, out of memory ...) ends as that code in the status of
    // each of its members and the OTHER units go on (round-5 advisor finding: returning at the first failure
    // left them half-stepped and their members at MI_RUNNING, which callers read as "pivot cap reached").
    auto unit_failed = [&](ManyUnit &u, int rc) {
        for (int64_t k : u.members) job->status[(size_t)k] = rc;
        u.phase = 3;
    };
    for (ManyUnit &u : job->units) {
        if (u.phase == 3) continue;
        const int64_t g = (int64_t)u.members.size();
        if (u.single) {
            const int rc = mi355x_simplex_solver_step(u.single, max_pivots, nullptr);
            if (rc < 0) { unit_failed(u, rc); continue; }
            // (a cancelled step is a paused job, not a finished one: the next call carries on)
            if ((rc == MI_MAX_PIVOTS && max_pivots > 0) || rc == MI_CANCELLED) { running = true; continue; }
            job->status[(size_t)u.members[0]] = rc;
            u.phase = 3;
            continue;
        }
        std::vector<int32_t> st((size_t)g);
        std::vector<int64_t> np((size_t)g);
        if (u.phase == 1) {                                                  // simplex.lisp:403, all members
            const int rc = mi355x_multibatch_solve(u.art_mb, 0, job->f, max_pivots, st.data(), np.data());
            if (rc == MI_CANCELLED) { running = true; continue; }
            if (rc != MI_OK) { unit_failed(u, rc < 0 ? rc : MI_HIP_ERROR); continue; }
            bool more = false;
            for (int64_t q = 0; q < g; ++q) { u.np1[(size_t)q] += np[(size_t)q]; u.st1[(size_t)q] = st[(size_t)q]; more |= st[(size_t)q] == MI_MAX_PIVOTS; }
            if (more && max_pivots > 0) { running = true; continue; }
            std::vector<int64_t> nd((size_t)g, 0);                           // :405-451, per member on the devices
            const int hrc = mi355x_multibatch_two_phase_handover(u.art_mb, u.main_mb, job->f, u.st1.data(), u.between.data(), nd.data());
            if (hrc != MI_OK) { unit_failed(u, hrc < 0 ? hrc : MI_HIP_ERROR); continue; }
            for (int64_t q = 0; q < g; ++q) u.np1[(size_t)q] += nd[(size_t)q];
            u.phase = 2;
            if (max_pivots > 0) { running = true; continue; }                // phase 2 in the next call (the chunk is used up)
        }
        const int rc = mi355x_multibatch_solve(u.main_mb, u.is_max, job->f, max_pivots, st.data(), np.data());   // :452 / :453-461
        if (rc == MI_CANCELLED) { running = true; continue; }
        if (rc != MI_OK) { unit_failed(u, rc < 0 ? rc : MI_HIP_ERROR); continue; }
        bool more = false;
        for (int64_t q = 0; q < g; ++q) {
            u.np2[(size_t)q] += np[(size_t)q];
            const bool went = u.between[(size_t)q] == MI_OK;
            if (went && st[(size_t)q] == MI_MAX_PIVOTS && max_pivots > 0) { more = true; continue; }
            job->status[(size_t)u.members[(size_t)q]] = went ? st[(size_t)q] : u.between[(size_t)q];
        }
        if (more) running = true; else u.phase = 3;
    }
    if (status) for (int64_t k = 0; k < job->n; ++k) status[k] = job->status[(size_t)k];
    return running ? MI_MAX_PIVOTS : MI_OK;
}

int mi355x_simplex_solver_many_finish(mi355x_solve_many *job, int32_t *status, mi355x_solution **out)
{
    if (!job) return hfail(MI_BAD_ARG, "job is NULL");
    std::unique_ptr<mi355x_solve_many> owner(job);                          // consumed whatever happens
    if (!out) return hfail(MI_BAD_ARG, "out is NULL");
    for (int64_t k = 0; k < job->n; ++k) out[k] = nullptr;
    for (ManyUnit &u : job->units) {
        const int64_t g = (int64_t)u.members.size();
        if (u.single) {
            const int64_t k = u.members[0];
            if (u.phase == 3 && job->status[(size_t)k] == MI_OPTIMAL) {
                mi355x_solution *s = nullptr;
                const int rc = mi355x_simplex_solver_finish(u.single, &s);
                u.single = nullptr;                                          // consumed
                if (rc != MI_OPTIMAL) return rc;
                job->sol[(size_t)k].reset(s);
            }
            continue;
        }
        for (int64_t q = 0; q < g; ++q) {
            const int64_t k = u.members[(size_t)q];
            if (job->status[(size_t)k] != MI_OPTIMAL) continue;
            mi355x_solution &s = *job->sol[(size_t)k];
            const int rc = mi355x_multibatch_download(u.main_mb, q, nullptr, s.basis.empty() ? nullptr : s.basis.data(),
                                                      s.last_row.data(), s.last_col.data());
            if (rc != MI_OK) return rc;
            s.n_pivots[0] = u.np1[(size_t)q]; s.n_pivots[1] = u.np2[(size_t)q];
        }
    }
    for (int64_t k = 0; k < job->n; ++k) {
        if (status) status[k] = job->status[(size_t)k];
        if (job->status[(size_t)k] == MI_OPTIMAL) out[k] = job->sol[(size_t)k].release();
    }
    return MI_OK;
}

void mi355x_simplex_solver_many_abandon(mi355x_solve_many *job) { delete job; }

void mi355x_solution_destroy(mi355x_solution *s) { delete s; }

int mi355x_solution_objective_value(const mi355x_solution *s, double *out)
{
    if (!s || !out) return hfail(MI_BAD_ARG, "NULL argument");
    *out = s->last_row[(size_t)s->cols - 1];                                // simplex.lisp:74-78
    return MI_OK;
}

static double basic_value(const mi355x_solution *s, int64_t col)
{
    for (size_t i = 0; i < s->basis.size(); ++i)                            // `position`: first match
        if (s->basis[i] == col) return s->last_col[i];
    return 0.0;
}

int mi355x_solution_variable(const mi355x_solution *s, int64_t var, double *out)
{
    if (!s || !out) return hfail(MI_BAD_ARG, "NULL argument");
    if (var < 0 || var >= (int64_t)s->map.size()) return hfail(MI_BAD_ARG, "not a variable in the tableau");
    const Mapping &mp = s->map[(size_t)var];                                // simplex.lisp:81-107
    if (mp.kind == kPositive)      *out = mp.offset + basic_value(s, mp.col);
    else if (mp.kind == kNegative) *out = mp.offset + (-basic_value(s, mp.col));
    else                           *out = basic_value(s, mp.col) - basic_value(s, mp.col + 1);
    return MI_OK;
}

int mi355x_solution_reduced_cost(const mi355x_solution *s, int64_t var, double *out)
{
    if (!s || !out) return hfail(MI_BAD_ARG, "NULL argument");
    if (var < 0 || var >= (int64_t)s->map.size()) return hfail(MI_BAD_ARG, "not a variable in the tableau");
    const Mapping &mp = s->map[(size_t)var];                                // simplex.lisp:111-120
    if (mp.kind != kPositive) return hfail(MI_BAD_ARG, "variable has no lower bound");
    *out = s->last_row[(size_t)mp.col];
    return MI_OK;
}

int mi355x_solution_pivots(const mi355x_solution *s, int64_t *phase1, int64_t *phase2)
{
    if (!s) return hfail(MI_BAD_ARG, "NULL argument");
    if (phase1) *phase1 = s->n_pivots[0];
    if (phase2) *phase2 = s->n_pivots[1];
    return MI_OK;
}

}  // extern "C"
