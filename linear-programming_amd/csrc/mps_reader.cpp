// mps_reader.cpp -- fixed-width MPS reader producing a mi355x_problem.
//
// C++ counterpart of the reference's read-mps (src/external-formats.lisp:78-348), the data
// format on the input side of the solver hook: same fixed-column fields, same sections (ROWS,
// COLUMNS, RHS, RANGES, BOUNDS incl. the BV / LI / UI integer extensions, OBJSENSE, OBJNAME,
// ENDATA ends the problem so files can be embedded in other streams), same :read-case modes,
// the same :rhs-id selection (first RHS vector seen unless one is named), single-variable rows
// folded into bounds, rows with a negative right-hand side flipped.  Output is the parsed
// `problem` struct (src/problem.lisp:45-53) as a mi355x_problem, in double-float.
//
// Deliberate differences from the reference (each is a place where the Lisp code cannot have
// been meant as written, none is exercised by its tests):
//   * RANGES looks rows up by name (the reference interns the name as a symbol and then
//     searches a string-keyed table, :253-258, which can never match);
//   * (opt-in, MI_MPS_SINGLE_VARIABLE_ROWS_AS_MEANT) a single-variable `<=` row tightens the UPPER
//     bound and a `>=` row the LOWER bound, with the sense reversed for a negative coefficient.  By
//     default such rows are handled as the reference handles them (:312-323: lb-max into the
//     upper-bound slot, ub-min into the integer flag, the row after a folded one skipped);
//   * numbers are read with strtod (the reference's hand-written exponent parsing, :150-161,
//     discards the exponent); plain decimals such as 4.5 give the same double;
//   * variable order is file order (the reference iterates hash tables); constraints come out in file
//     order with the opt-in above, in the reference's push order (reverse of the rows) by default.
#include "../../include/mi355x_simplex.h"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

extern "C" void mi355x_set_last_error_(const char *msg);

namespace {

int mfail(int code, const std::string &msg) { mi355x_set_last_error_(msg.c_str()); return code; }
thread_local std::string g_mps_note;

std::string substring(const std::string &s, size_t a, size_t b)
{
    a = std::min(a, s.size()); b = std::min(b, s.size());
    return a < b ? s.substr(a, b - a) : std::string();
}

// field n of a data line, fixed columns (src/external-formats.lisp:100-104); n = 0: the line
std::string field(const std::string &line, int n)
{
    static const size_t start[] = {0, 1, 4, 14, 24, 39, 49}, end[] = {61, 3, 12, 22, 36, 47, 61};
    return substring(line, start[n], end[n]);
}

std::string trim(const std::string &s)
{
    size_t a = 0, b = s.size();
    while (a < b && s[a] == ' ') ++a;
    while (b > a && s[b - 1] == ' ') --b;
    return s.substr(a, b - a);
}

// :read-case handling of names (:108-127): 0 upcase, 1 downcase, 2 preserve, 3 invert
std::string name_case(std::string s, int mode)
{
    auto up = [](std::string x) { for (auto &c : x) c = (char)std::toupper((unsigned char)c); return x; };
    auto down = [](std::string x) { for (auto &c : x) c = (char)std::tolower((unsigned char)c); return x; };
    if (mode == 0) return up(s);
    if (mode == 1) return down(s);
    if (mode == 3) {
        // (every #'upper-case-p raw) is false for any non-letter, exactly as in the reference
        const bool all_up = !s.empty() && std::all_of(s.begin(), s.end(), [](char c) { return std::isupper((unsigned char)c); });
        const bool all_lo = !s.empty() && std::all_of(s.begin(), s.end(), [](char c) { return std::islower((unsigned char)c); });
        if (all_up) return down(s);
        if (all_lo) return up(s);
    }
    return s;
}

bool parse_number(const std::string &raw, double *out)
{
    const std::string t = trim(raw);
    if (t.empty()) return false;
    char *endp = nullptr;
    const double v = std::strtod(t.c_str(), &endp);
    if (endp == t.c_str()) return false;
    *out = v;
    return true;
}

struct Row {
    int type;                         // 0 <=, 1 >=, 2 =, 3 objective
    double rhs = 0.0;
    bool has_range = false;
    double range = 0.0;
    std::vector<int64_t> var;
    std::vector<double>  coef;
};

struct VarInfo {
    bool has_lb = true, has_ub = false;   // default (0 nil nil): lower bound 0, no upper bound
    double lb = 0.0, ub = 0.0;
    bool integer = false;
};

}  // namespace

struct mi355x_mps_names {                 // variable names in problem-vars order
    std::vector<std::string> vars;
    std::string objective;
};

static thread_local mi355x_mps_names g_names;

extern "C" {

int mi355x_problem_read_mps(const char *text, int64_t len, int default_is_max, const char *rhs_id_in,
                            int read_case, mi355x_problem **out)
{
    return mi355x_problem_read_mps_ex(text, len, default_is_max, rhs_id_in, read_case, 0, out);
}

int mi355x_problem_read_mps_ex(const char *text, int64_t len, int default_is_max, const char *rhs_id_in,
                               int read_case, int flags, mi355x_problem **out)
{
    g_mps_note.clear();
    if (!text || len < 0 || !out) return mfail(MI_BAD_ARG, "bad arguments");
    *out = nullptr;
    if (read_case < 0 || read_case > 3) return mfail(MI_BAD_ARG, "read_case must be 0..3");
    int is_max = default_is_max;          // 1 max, 0 min, -1: must come from OBJSENSE
    std::string rhs_id = rhs_id_in ? rhs_id_in : "";
    bool have_rhs_id = rhs_id_in != nullptr;
    std::string header, objective;
    std::vector<std::string> row_names;   // declaration order
    std::map<std::string, Row> rows;
    std::vector<std::string> var_names;
    std::map<std::string, int64_t> var_index;
    std::vector<VarInfo> vinfo;
    auto var_of = [&](const std::string &name) -> int64_t {
        auto it = var_index.find(name);
        if (it != var_index.end()) return it->second;
        var_index[name] = (int64_t)var_names.size();
        var_names.push_back(name);
        vinfo.push_back(VarInfo());
        return (int64_t)var_names.size() - 1;
    };
    auto add_coef = [&](const std::string &row, int64_t v, double c) -> bool {
        auto it = rows.find(row);
        if (it == rows.end()) return false;
        it->second.var.push_back(v);
        it->second.coef.push_back(c);
        return true;
    };

    const std::string all(text, (size_t)len);
    size_t pos = 0;
    bool ended = false;
    while (pos <= all.size() && !ended) {
        size_t nl = all.find('\n', pos);
        std::string line = all.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
        pos = nl == std::string::npos ? all.size() + 1 : nl + 1;
        while (!line.empty() && (line.back() == ' ' || line.back() == '\r')) line.pop_back();
        if (line.empty()) continue;
        if (line[0] != ' ') {                                             // header card, :164-176
            std::string card = substring(line, 0, 15);
            for (auto &c : card) c = (char)std::tolower((unsigned char)c);
            if (card[0] == '*') continue;                                 // comment
            if (card.compare(0, 4, "name") == 0 && (card.size() == 4 || card[4] == ' ')) { header.clear(); continue; }
            if (card == "endata") { ended = true; break; }
            header = card;
            continue;
        }
        double num = 0.0;
        if (header == "rows") {                                           // :179-192
            const std::string f1 = field(line, 1);
            const char tc = f1.empty() ? ' ' : (char)std::tolower((unsigned char)trim(f1).c_str()[0]);
            Row r;
            if (tc == 'n') r.type = 3; else if (tc == 'g') r.type = 1; else if (tc == 'l') r.type = 0;
            else if (tc == 'e') r.type = 2; else return mfail(MI_BAD_ARG, "unknown row type in: " + line);
            const std::string name = name_case(trim(field(line, 2)), read_case);
            if (r.type == 3 && objective.empty()) objective = name;       // first N row by default
            if (!rows.count(name)) row_names.push_back(name);
            rows[name] = r;
        } else if (header == "columns") {                                 // :194-206
            const int64_t v = var_of(name_case(trim(field(line, 2)), read_case));
            if (!parse_number(field(line, 4), &num)) return mfail(MI_BAD_ARG, "bad number in: " + line);
            if (!add_coef(name_case(trim(field(line, 3)), read_case), v, num))
                return mfail(MI_BAD_ARG, "undeclared row in: " + line);
            if (!field(line, 5).empty()) {
                if (!parse_number(field(line, 6), &num)) return mfail(MI_BAD_ARG, "bad number in: " + line);
                if (!add_coef(name_case(trim(field(line, 5)), read_case), v, num))
                    return mfail(MI_BAD_ARG, "undeclared row in: " + line);
            }
        } else if (header == "rhs" || header == "ranges") {               // :208-236
            const std::string id = name_case(trim(field(line, 2)), read_case);
            if (!have_rhs_id) { rhs_id = id; have_rhs_id = true; }
            if (header == "rhs" && id != rhs_id) continue;
            for (int k = 3; k <= 5; k += 2) {
                if (k == 5 && field(line, 5).empty()) break;
                auto it = rows.find(name_case(trim(field(line, k)), read_case));
                if (it == rows.end()) return mfail(MI_BAD_ARG, "undeclared row in: " + line);
                if (!parse_number(field(line, k + 1), &num)) return mfail(MI_BAD_ARG, "bad number in: " + line);
                if (header == "rhs") it->second.rhs = num;
                else { it->second.has_range = true; it->second.range = num; }
            }
        } else if (header == "bounds") {                                  // :238-268
            const int64_t v = var_of(name_case(trim(field(line, 3)), read_case));
            VarInfo &vi = vinfo[(size_t)v];
            std::string bt = trim(field(line, 1));
            for (auto &c : bt) c = (char)std::toupper((unsigned char)c);
            const bool needs_num = bt == "LO" || bt == "UP" || bt == "FX" || bt == "LI" || bt == "UI";
            if (needs_num && !parse_number(field(line, 4), &num)) return mfail(MI_BAD_ARG, "bad number in: " + line);
            if (bt == "LO")      { vi.has_lb = true; vi.lb = num; }
            else if (bt == "UP") { vi.has_ub = true; vi.ub = num; }
            else if (bt == "FX") { vi.has_lb = vi.has_ub = true; vi.lb = vi.ub = num; }
            else if (bt == "FR") { vi.has_lb = vi.has_ub = false; }
            else if (bt == "MI") { vi.has_lb = false; }
            else if (bt == "PL") { vi.has_ub = false; }
            else if (bt == "BV") { vi.has_lb = vi.has_ub = true; vi.lb = 0.0; vi.ub = 1.0; vi.integer = true; }
            else if (bt == "LI") { vi.has_lb = true; vi.lb = num; vi.integer = true; }
            else if (bt == "UI") { vi.has_ub = true; vi.ub = num; vi.integer = true; }
            else return mfail(MI_BAD_ARG, "\"" + bt + "\" is not a know bound type");
        } else if (header == "objsense") {                                // :270-282
            header.clear();
            std::string t = trim(field(line, 0));
            for (auto &c : t) c = (char)std::tolower((unsigned char)c);
            if (t == "max" || t == "maximizing") is_max = 1;
            else if (t == "min" || t == "minimizing") is_max = 0;
            else return mfail(MI_BAD_ARG, "\"" + t + "\" is not a know problem type");
        } else if (header == "objname") {                                 // :284-286
            header.clear();
            objective = name_case(trim(field(line, 0)), read_case);
        } else {
            return mfail(MI_BAD_ARG, "Unknown header-card " + header);
        }
    }
    if (is_max != 0 && is_max != 1) return mfail(MI_BAD_ARG, "No valid problem type was specified");
    if (!rows.count(objective)) return mfail(MI_BAD_ARG, "no objective row");

    // rows -> constraints (+ the second constraint a RANGES entry implies), :297-311
    std::vector<Row> cons;
    for (const auto &name : row_names) {
        const Row &r = rows[name];
        if (r.type == 3) continue;
        cons.push_back(r);
        if (r.has_range) {
            Row x = r;
            const double a = r.range < 0 ? -r.range : r.range;
            if (r.type == 0)      { x.type = 1; x.rhs = r.rhs - a; }
            else if (r.type == 1) { x.type = 0; x.rhs = r.rhs + a; }
            else if (r.range > 0) { x.type = 0; x.rhs = r.rhs + r.range; }
            else if (r.range < 0) { x.type = 1; x.rhs = r.rhs + r.range; }
            else continue;
            cons.push_back(x);
        }
    }
    // single-variable rows become bounds, negative right-hand sides are flipped, :312-335
    std::vector<Row> kept;
    if (flags & MI_MPS_SINGLE_VARIABLE_ROWS_AS_MEANT) {
        // opt-in: what such a row MEANS -- `<=` tightens the upper bound, `>=` the lower bound, the
        // sense flips for a negative coefficient, `=` fixes the variable; every row is looked at
        for (Row &c : cons) {
            if (c.var.size() == 1 && c.coef[0] != 0.0) {
                VarInfo &vi = vinfo[(size_t)c.var[0]];
                const double bound = c.rhs / c.coef[0];
                int op = c.type;
                if (c.coef[0] < 0 && op != 2) op = 1 - op;
                if (op == 0 || op == 2) { vi.ub = vi.has_ub ? std::min(vi.ub, bound) : bound; vi.has_ub = true; }
                if (op == 1 || op == 2) { vi.lb = vi.has_lb ? std::max(vi.lb, bound) : bound; vi.has_lb = true; }
                continue;
            }
            if (c.rhs < 0) {
                for (auto &x : c.coef) x = -x;
                c.rhs = -c.rhs;
                c.type = c.type == 0 ? 1 : c.type == 1 ? 0 : 2;
            }
            kept.push_back(c);
        }
    } else {
        // default: the reference's loop as written (:312-335).  `constraints` is the list PUSH left --
        // the rows in reverse (hash tables iterated in insertion order, as SBCL does), a RANGES
        // companion ahead of its row.  A single-variable row (info = (lb ub integer-flag), :313-318):
        //   <=   ub   := (lb-max ub bound)        nil -> bound, else the LARGER one; the sign of the
        //   >=   flag := (ub-min flag bound)      coefficient is not looked at; the integer flag becomes
        //   =    both                             a number, i.e. true: the variable turns integer
        // and is then spliced out by copying the NEXT cell over it (:320-321) -- the loop moves on to the
        // cell after that, so the constraint that followed a single-variable row is neither folded nor
        // has its negative right-hand side flipped; a single-variable row at the END of the list leaves
        // NIL in its place.
        std::vector<Row> L(cons.rbegin(), cons.rend());
        std::vector<char> flag_is_number(vinfo.size(), 0);
        std::string notes;                                     // what the reference's loop did to the rows' meaning (round-5 advisor finding)
        int n_notes = 0;
        auto note = [&](const std::string &what) { if (++n_notes <= 8) notes += (notes.empty() ? "" : "; ") + what; };
        size_t i = 0;
        while (i < L.size()) {
            Row &c = L[i];
            if (c.var.size() == 1) {
                const size_t v = (size_t)c.var[0];
                VarInfo &vi = vinfo[v];
                if (c.coef[0] == 0.0) return mfail(MI_BAD_ARG, "single-variable row with a zero coefficient: the reference divides by it (:315)");
                const double bound = c.rhs / c.coef[0];
                if (c.type == 0 || c.type == 2) {
                    if (vi.has_ub && bound > vi.ub) note("<= row on " + var_names[v] + " RAISES its upper bound (lb-max, :316)");
                    if (c.coef[0] < 0) note("the negative coefficient of the single-variable row on " + var_names[v] + " is not looked at");
                    vi.ub = vi.has_ub ? std::max(vi.ub, bound) : bound; vi.has_ub = true;
                }
                if (c.type == 1 || c.type == 2) {
                    if (vi.integer && !flag_is_number[v])
                        return mfail(MI_BAD_ARG, "a >= / = single-variable row on an integer variable: the reference calls (min t bound) (:317-318)");
                    if (!vi.integer) note(std::string(c.type == 1 ? ">=" : "=") + " row on " + var_names[v] + " turns it INTEGER (ub-min on the integer flag, :317-318)");
                    vi.integer = true;
                    flag_is_number[v] = 1;
                }
                if (i + 1 == L.size())
                    return mfail(MI_UNSUPPORTED, "a single-variable row ends the reference's constraint list: it leaves NIL among the "
                                                 "problem's constraints (:320-321), which no solver accepts");
                L.erase(L.begin() + (std::ptrdiff_t)i);           // the next cell's contents move here ...
                if (L[i].var.size() == 1 || L[i].rhs < 0)
                    note("the constraint behind the folded row on " + var_names[v] + " is stepped over (:320-321): neither folded nor sign-normalised");
                i += 1;                                            // ... and are stepped over
                continue;
            }
            if (c.rhs < 0) {
                for (auto &x : c.coef) x = -x;
                c.rhs = -c.rhs;
                c.type = c.type == 0 ? 1 : c.type == 1 ? 0 : 2;
            }
            i += 1;
        }
        kept = L;
        g_mps_note = notes.empty() ? "" : "mps note: " + notes + (n_notes > 8 ? " (+" + std::to_string(n_notes - 8) + " more)" : "") +
                                          " -- the reference's loop, src/external-formats.lisp:312-323; MI_MPS_SINGLE_VARIABLE_ROWS_AS_MEANT reads the rows as written";
    }

    mi355x_problem *p = nullptr;
    if (var_names.empty()) return mfail(MI_BAD_ARG, "no variables");
    int rc = mi355x_problem_create(&p, is_max, (int64_t)var_names.size());
    if (rc != MI_OK) return rc;
    const Row &obj = rows[objective];
    rc = mi355x_problem_set_objective(p, obj.var.data(), obj.coef.data(), (int64_t)obj.var.size());
    for (size_t v = 0; rc == MI_OK && v < vinfo.size(); ++v) {
        const VarInfo &vi = vinfo[v];
        if (vi.integer) rc = mi355x_problem_set_integer(p, (int64_t)v);
        // only non-default bounds get an entry (:341-344); validate-bounds = ub >= lb
        if (rc == MI_OK && (!(vi.has_lb && vi.lb == 0.0) || vi.has_ub))
            rc = mi355x_problem_set_bounds(p, (int64_t)v, vi.has_lb, vi.lb, vi.has_ub, vi.ub);
    }
    for (size_t k = 0; rc == MI_OK && k < kept.size(); ++k)
        rc = mi355x_problem_add_constraint(p, kept[k].type, kept[k].var.data(), kept[k].coef.data(),
                                           (int64_t)kept[k].var.size(), kept[k].rhs);
    if (rc != MI_OK) { mi355x_problem_destroy(p); return rc; }
    g_names.vars = var_names;
    g_names.objective = objective;
    *out = p;
    mi355x_set_last_error_(g_mps_note.c_str());                // (MI_OK with a note, or an empty string)
    return MI_OK;
}

/* names of the problem most recently read by THIS thread (problem-vars order) */
int64_t mi355x_mps_var_count(void) { return (int64_t)g_names.vars.size(); }
const char *mi355x_mps_var_name(int64_t i)
{
    return (i >= 0 && i < (int64_t)g_names.vars.size()) ? g_names.vars[(size_t)i].c_str() : "";
}
const char *mi355x_mps_objective_name(void) { return g_names.objective.c_str(); }

}  // extern "C"
