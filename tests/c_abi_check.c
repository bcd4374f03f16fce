/* Plain-C client of include/mi355x_simplex.h: proves the header is valid C (not just C++) and
 * that a C program links against libmi355x_simplex.so.  Exercises only host-side entry points,
 * so it also runs on a machine without a GPU, where every device entry point must fail loudly
 * with MI_NO_DEVICE.  Built and run by tests/test_capi_symbols.py. */
#include <stdio.h>
#include <string.h>
#include "mi355x_simplex.h"

int main(void)
{
    mi355x_problem *p = NULL;
    int64_t rows = 0, cols = 0;
    int two_phase = -1, rc;
    double M[3 * 6];
    int64_t basis[2];
    /* README.md:43-47 : max x + 4y + 3z, 2x + y <= 8, y + z <= 7 */
    int64_t ov[] = {0, 1, 2};  double oc[] = {1, 4, 3};
    int64_t v1[] = {0, 1};     double c1[] = {2, 1};
    int64_t v2[] = {1, 2};     double c2[] = {1, 1};
    const double expect[18] = {2, 1, 0, 1, 0, 8,  0, 1, 1, 0, 1, 7,  -1, -4, -3, 0, 0, 0};

    if (mi355x_abi_version() != MI355X_SIMPLEX_ABI_VERSION) return 1;
    if (mi355x_problem_create(&p, 1, 3) != MI_OK) return 2;
    if (mi355x_problem_set_objective(p, ov, oc, 3) != MI_OK) return 3;
    if (mi355x_problem_add_constraint(p, 0, v1, c1, 2, 8.0) != MI_OK) return 4;
    if (mi355x_problem_add_constraint(p, 0, v2, c2, 2, 7.0) != MI_OK) return 5;
    if (mi355x_build_tableau(p, 0, &rows, &cols, NULL, NULL, &two_phase) != MI_OK) return 6;
    if (rows != 3 || cols != 6 || two_phase != 0) return 7;
    if (mi355x_build_tableau(p, 0, NULL, NULL, M, basis, NULL) != MI_OK) return 8;
    if (memcmp(M, expect, sizeof expect) != 0 || basis[0] != 3 || basis[1] != 4) return 9;   /* t/simplex.lisp:60-72 */

    if (mi355x_device_count() == 0) {
        mi355x_tab *t = NULL;
        mi355x_solution *s = NULL;
        rc = mi355x_tab_create(&t, rows, cols, M, basis, 0);
        if (rc != MI_NO_DEVICE || t != NULL) return 10;
        if (mi355x_simplex_solver(p, 1024.0, 0, &s) != MI_NO_DEVICE || s != NULL) return 11;
        {   /* the resumable form fails the same way, and its argument checks need no device */
            mi355x_solve *job = NULL;
            if (mi355x_simplex_solver_begin(p, 1024.0, 0, &job) != MI_NO_DEVICE || job != NULL) return 61;
            if (mi355x_simplex_solver_step(NULL, 0, NULL) != MI_BAD_ARG || mi355x_simplex_solver_cancel(NULL) != MI_BAD_ARG ||
                mi355x_simplex_solver_finish(NULL, &s) != MI_BAD_ARG || s != NULL) return 62;
            if (mi355x_two_phase_handover(NULL, NULL, 1024.0, NULL) != MI_BAD_ARG) return 63;
            if (mi355x_simplex_solver_many_step(NULL, 0, NULL) != MI_BAD_ARG) return 86;
            mi355x_simplex_solver_many_abandon(NULL);
            mi355x_simplex_solver_abandon(NULL);
        }
        {   /* the multi-device entry points fail the same way (and never touch RCCL) */
            mi355x_colpart *cp = NULL;
            if (mi355x_colpart_create(&cp, rows, cols, M, basis, 8) != MI_NO_DEVICE || cp != NULL) return 14;
            if (mi355x_colpart_solve(NULL, 1, 1024.0, 0, NULL) != MI_BAD_ARG) return 15;
            if (mi355x_colpart_create_on(&cp, rows, cols, M, basis, 2, NULL) != MI_NO_DEVICE || cp != NULL) return 20;
        }
        {   /* ... and the batch handles */
            mi355x_multibatch *mb = NULL;
            if (mi355x_multibatch_create(&mb, 1, rows, cols, M, basis, 8, NULL) != MI_NO_DEVICE || mb != NULL) return 21;
            if (mi355x_batch_sync(NULL, NULL, NULL) != MI_BAD_ARG) return 22;
            if (mi355x_tab_cancel(NULL) != MI_BAD_ARG || mi355x_batch_cancel(NULL) != MI_BAD_ARG ||
                mi355x_multibatch_cancel(NULL) != MI_BAD_ARG || mi355x_colpart_cancel(NULL) != MI_BAD_ARG) return 36;
        }
        printf("no device: %s\n", mi355x_last_error());
    } else {
        mi355x_solution *s = NULL;
        double w = 0, x = 0;
        if (mi355x_simplex_solver(p, 1024.0, 0, &s) != MI_OPTIMAL) return 12;
        mi355x_solution_objective_value(s, &w);
        mi355x_solution_variable(s, 0, &x);
        if (w != 28.5 || x != 0.5) return 13;                                                /* README.md:58-62 */
        mi355x_solution_destroy(s);
        {   /* the Lisp glue's native route (solve-natively), call for call: begin, step in bounded
             * chunks (cap 1: MI_MAX_PIVOTS = "chunk used up, still running"), finish, the read-back
             * behind the four solution-* generics, destroy */
            mi355x_solve *job = NULL;
            mi355x_solution *s2 = NULL;
            int64_t k = 0, total = 0, p1 = -1, p2 = -1;
            double v = 0;
            int calls = 0;
            if (mi355x_simplex_solver_begin(p, 1024.0, 0, &job) != MI_OK || !job) return 64;
            while ((rc = mi355x_simplex_solver_step(job, 1, &k)) == MI_MAX_PIVOTS) { total += k; if (++calls > 16) return 65; }
            total += k;
            if (rc != MI_OPTIMAL || total != 2) return 66;                                   /* t/simplex.lisp:190: two pivots */
            if (mi355x_simplex_solver_finish(job, &s2) != MI_OPTIMAL || !s2) return 67;
            if (mi355x_solution_objective_value(s2, &v) != MI_OK || v != 28.5) return 68;
            if (mi355x_solution_variable(s2, 1, &v) != MI_OK || v != 7.0) return 69;
            if (mi355x_solution_reduced_cost(s2, 2, &v) != MI_OK || v != 0.5) return 70;       /* README.md:58-62 */
            if (mi355x_solution_variable(s2, 3, &v) != MI_BAD_ARG) return 71;                  /* "not a variable in the tableau" */
            if (mi355x_solution_pivots(s2, &p1, &p2) != MI_OK || p1 != 0 || p2 != 2) return 72;
            mi355x_solution_destroy(s2);
        }
        {   /* a LIST of problems as ONE job (the glue's :native :many): the README LP twice (one batch of
             * two) next to a two-phase problem alone in its group, stepped one pivot per call */
            mi355x_problem *q = NULL;
            const mi355x_problem *list[3];
            mi355x_solve_many *job = NULL;
            mi355x_solution *sols[3] = {NULL, NULL, NULL};
            int32_t st[3] = {-1, -1, -1};
            int64_t v3[] = {0, 1};  double c3[] = {1, 1};
            double v = 0;
            int calls = 0;
            if (mi355x_problem_create(&q, 1, 3) != MI_OK) return 80;
            mi355x_problem_set_objective(q, ov, oc, 3);
            mi355x_problem_add_constraint(q, 0, v1, c1, 2, 8.0);
            mi355x_problem_add_constraint(q, 0, v2, c2, 2, 7.0);
            mi355x_problem_add_constraint(q, 1, v3, c3, 2, 2.0);
            list[0] = p; list[1] = q; list[2] = p;
            if (mi355x_simplex_solver_many_begin(list, 3, 1024.0, 2, NULL, &job) != MI_OK || !job) return 81;
            while ((rc = mi355x_simplex_solver_many_step(job, 1, st)) == MI_MAX_PIVOTS) if (++calls > 64) return 82;
            if (rc != MI_OK || st[0] != MI_OPTIMAL || st[1] != MI_OPTIMAL || st[2] != MI_OPTIMAL) return 83;
            if (mi355x_simplex_solver_many_finish(job, st, sols) != MI_OK || !sols[0] || !sols[1] || !sols[2]) return 84;
            for (int k = 0; k < 3; ++k) {
                if (mi355x_solution_objective_value(sols[k], &v) != MI_OK || v != 28.5) return 85;
                mi355x_solution_destroy(sols[k]);
            }
            mi355x_problem_destroy(q);
        }
        {   /* the same tableau column-partitioned over 2 shards (logical shards on one GPU),
             * what the Lisp glue does for :devices 2 -- t/simplex.lisp:170-194: objective 57/2 */
            mi355x_colpart *cp = NULL;
            double last_col[3];
            int64_t b2[2], np = 0;
            if (mi355x_colpart_create(&cp, rows, cols, M, basis, 2) != MI_OK) return 16;
            if (mi355x_colpart_solve(cp, 1, 1024.0, 0, &np) != MI_OPTIMAL || np != 2) return 17;
            if (mi355x_colpart_download(cp, NULL, b2, NULL, last_col) != MI_OK) return 18;
            if (last_col[2] != 28.5 || b2[0] != 0 || b2[1] != 1) return 19;
            mi355x_colpart_destroy(cp);
        }
        {   /* two copies of the LP as one batch over 2 (logical) sub-batches, one call */
            mi355x_multibatch *mb = NULL;
            double MM[2 * 18], lc[3];
            int64_t bb[4], npv[2] = {0, 0};
            int32_t st[2] = {-1, -1};
            int nsub = 0, ndev = 0;
            memcpy(MM, M, sizeof M); memcpy(MM + 18, M, sizeof M);
            bb[0] = bb[2] = basis[0]; bb[1] = bb[3] = basis[1];
            if (mi355x_multibatch_create(&mb, 2, rows, cols, MM, bb, 2, NULL) != MI_OK) return 23;
            if (mi355x_multibatch_info(mb, &nsub, &ndev) != MI_OK || nsub != 2 || ndev != 1) return 24;
            if (mi355x_multibatch_solve(mb, 1, 1024.0, 0, st, npv) != MI_OK) return 25;
            if (st[0] != MI_OPTIMAL || st[1] != MI_OPTIMAL || npv[0] != 2 || npv[1] != 2) return 26;
            if (mi355x_multibatch_download(mb, 1, NULL, NULL, NULL, lc) != MI_OK || lc[2] != 28.5) return 27;
            mi355x_multibatch_destroy(mb);
        }
        {   /* a two-phase problem (x + y >= 2 next to the two <= rows) with its ARTIFICIAL tableau
             * column-partitioned: phase 1, hand-over and phase 2 on the partition; the main tableau
             * contributes its objective row only */
            mi355x_problem *q = NULL;
            int64_t v3[] = {0, 1};  double c3[] = {1, 1};
            int64_t ar = 0, ac = 0, mr = 0, mc = 0, npv[2] = {0, 0};
            double A[4 * 9], Mm[4 * 8], lc[4];
            int64_t ab[3], mb2[3];
            mi355x_colpart *art = NULL, *mn = NULL;
            int tp = 0;
            if (mi355x_problem_create(&q, 1, 3) != MI_OK) return 28;
            mi355x_problem_set_objective(q, ov, oc, 3);
            mi355x_problem_add_constraint(q, 0, v1, c1, 2, 8.0);
            mi355x_problem_add_constraint(q, 0, v2, c2, 2, 7.0);
            mi355x_problem_add_constraint(q, 1, v3, c3, 2, 2.0);
            if (mi355x_build_tableau(q, 1, &ar, &ac, NULL, NULL, &tp) != MI_OK || !tp || ar != 4 || ac > 9) return 29;
            if (mi355x_build_tableau(q, 1, NULL, NULL, A, ab, NULL) != MI_OK) return 30;
            if (mi355x_build_tableau(q, 0, &mr, &mc, NULL, NULL, NULL) != MI_OK || mr != 4 || mc > 8) return 31;
            if (mi355x_build_tableau(q, 0, NULL, NULL, Mm, mb2, NULL) != MI_OK) return 32;
            if (mi355x_colpart_create(&art, ar, ac, A, ab, 2) != MI_OK) return 33;
            if (mi355x_colpart_solve_two_phase(art, mc, Mm + (mr - 1) * mc, 1, 1024.0, npv, &mn) != MI_OPTIMAL || !mn) return 34;
            if (mi355x_colpart_download(mn, NULL, NULL, NULL, lc) != MI_OK || lc[3] != 28.5) return 35;
            mi355x_colpart_destroy(mn);
            mi355x_colpart_destroy(art);
            mi355x_problem_destroy(q);
        }
        {   /* the call sequence of the glue's mi355x-solve-problems on a mixed list: the single-phase
             * members as ONE batch solved in bounded chunks (cap 1 here: MI_MAX_PIVOTS = "chunk used up",
             * call again) with the light read-back per member; the two-phase member peeled off to
             * mi355x_solve_two_phase (t/simplex.lisp:239-275: x + y >= 2 ... objective 57/2 still) */
            mi355x_multibatch *mb = NULL;
            double MM[2 * 18], lr[6], lc[3];
            int64_t bb[4], npv[2], b1[2];
            int32_t st[2];
            int calls = 0, running = 1;
            memcpy(MM, M, sizeof M); memcpy(MM + 18, M, sizeof M);
            MM[18 + 12] = -3.0;                                   /* second member: max 3x + 4y + 3z */
            bb[0] = bb[2] = basis[0]; bb[1] = bb[3] = basis[1];
            if (mi355x_multibatch_create(&mb, 2, rows, cols, MM, bb, 1, NULL) != MI_OK) return 37;
            while (running) {
                if (mi355x_multibatch_solve(mb, 1, 1024.0, 1, st, npv) != MI_OK) return 38;
                running = st[0] == MI_MAX_PIVOTS || st[1] == MI_MAX_PIVOTS;
                if (++calls > 16) return 39;
            }
            if (st[0] != MI_OPTIMAL || st[1] != MI_OPTIMAL || calls != 3) return 40;         /* 2 and 3 pivots, one per call (optimality is seen before the cap) */
            if (mi355x_multibatch_download(mb, 0, NULL, b1, lr, lc) != MI_OK || lc[2] != 28.5 || lr[5] != 28.5) return 41;
            if (b1[0] != 0 || b1[1] != 1) return 42;
            if (mi355x_multibatch_download(mb, 1, NULL, b1, lr, lc) != MI_OK || lc[2] != 33.0) return 43;   /* x = 4, y = 0, z = 7: 12 + 21 */
            mi355x_multibatch_destroy(mb);
        }
        {
            mi355x_problem *q = NULL;
            int64_t v3[] = {0, 1};  double c3[] = {1, 1};
            int64_t ar = 0, ac = 0, mr = 0, mc = 0, npv[2] = {0, 0};
            double A[4 * 9], Mm[4 * 8], lc[4];
            int64_t ab[3], mb2[3];
            mi355x_tab *art = NULL, *mn = NULL;
            if (mi355x_problem_create(&q, 1, 3) != MI_OK) return 44;
            mi355x_problem_set_objective(q, ov, oc, 3);
            mi355x_problem_add_constraint(q, 0, v1, c1, 2, 8.0);
            mi355x_problem_add_constraint(q, 0, v2, c2, 2, 7.0);
            mi355x_problem_add_constraint(q, 1, v3, c3, 2, 2.0);
            if (mi355x_build_tableau(q, 1, &ar, &ac, NULL, NULL, NULL) != MI_OK) return 45;
            if (mi355x_build_tableau(q, 1, NULL, NULL, A, ab, NULL) != MI_OK) return 46;
            if (mi355x_build_tableau(q, 0, &mr, &mc, NULL, NULL, NULL) != MI_OK) return 47;
            if (mi355x_build_tableau(q, 0, NULL, NULL, Mm, mb2, NULL) != MI_OK) return 48;
            {   /* the same problem twice as a pair of batches: phase 1, hand-over and phase 2 member by
                 * member on the device (what mi355x-solve-problems does with two-phase members of one shape) */
                mi355x_multibatch *ba = NULL, *bm = NULL;
                double AA[2 * 4 * 9], MM2[2 * 4 * 8], lc2[4];
                int64_t ab2[6], mb3[6], np2[4] = {0, 0, 0, 0};
                int32_t st2[2] = {-1, -1};
                memcpy(AA, A, (size_t)(ar * ac) * sizeof(double)); memcpy(AA + ar * ac, A, (size_t)(ar * ac) * sizeof(double));
                memcpy(MM2, Mm, (size_t)(mr * mc) * sizeof(double)); memcpy(MM2 + mr * mc, Mm, (size_t)(mr * mc) * sizeof(double));
                memcpy(ab2, ab, 3 * sizeof(int64_t)); memcpy(ab2 + 3, ab, 3 * sizeof(int64_t));
                memcpy(mb3, mb2, 3 * sizeof(int64_t)); memcpy(mb3 + 3, mb2, 3 * sizeof(int64_t));
                if (mi355x_multibatch_create(&ba, 2, ar, ac, AA, ab2, 1, NULL) != MI_OK) return 56;
                if (mi355x_multibatch_create(&bm, 2, mr, mc, MM2, mb3, 1, NULL) != MI_OK) return 57;
                if (mi355x_multibatch_solve_two_phase(ba, bm, 1, 1024.0, st2, np2) != MI_OK) return 58;
                if (st2[0] != MI_OPTIMAL || st2[1] != MI_OPTIMAL || np2[0] != np2[2] || np2[1] != np2[3]) return 59;
                if (mi355x_multibatch_download(bm, 1, NULL, NULL, NULL, lc2) != MI_OK || lc2[3] != 28.5) return 60;
                mi355x_multibatch_destroy(bm);
                mi355x_multibatch_destroy(ba);
            }
            {   /* the glue's solve-two-phase-in-chunks: phase 1 in bounded calls, the hand-over on its
                 * own, phase 2 in bounded calls -- the pivots of the one-call form below */
                mi355x_tab *a2 = NULL, *m2 = NULL;
                int64_t k = 0, n1 = 0, n2 = 0, nd = -1;
                if (mi355x_tab_create(&a2, ar, ac, A, ab, 0) != MI_OK) return 73;
                if (mi355x_tab_create(&m2, mr, mc, Mm, mb2, 0) != MI_OK) return 74;
                while ((rc = mi355x_tab_solve(a2, 0, 1024.0, 1, &k)) == MI_MAX_PIVOTS) n1 += k;
                n1 += k;
                if (rc != MI_OPTIMAL) return 75;
                if (mi355x_two_phase_handover(a2, m2, 1024.0, &nd) != MI_OK || nd != 0) return 76;
                while ((rc = mi355x_tab_solve(m2, 1, 1024.0, 1, &k)) == MI_MAX_PIVOTS) n2 += k;
                n2 += k;
                if (rc != MI_OPTIMAL) return 77;
                if (mi355x_tab_download(m2, NULL, NULL, NULL, lc) != MI_OK || lc[3] != 28.5) return 78;
                npv[0] = n1; npv[1] = n2;
                mi355x_tab_destroy(m2);
                mi355x_tab_destroy(a2);
            }
            if (mi355x_tab_create(&art, ar, ac, A, ab, 0) != MI_OK) return 49;
            if (mi355x_tab_create(&mn, mr, mc, Mm, mb2, 0) != MI_OK) return 50;
            {
                int64_t n12[2] = {0, 0};
                if (mi355x_solve_two_phase(art, mn, 1, 1024.0, n12) != MI_OPTIMAL) return 51;
                if (n12[0] != npv[0] || n12[1] != npv[1]) return 79;
            }
            if (mi355x_tab_download(mn, NULL, NULL, NULL, lc) != MI_OK || lc[3] != 28.5) return 52;
            /* a cancel request with no solve in flight is aimed at the next solve: whole pivots, and
             * the request ends with that solve */
            if (mi355x_tab_cancel(mn) != MI_OK) return 53;
            {
                int64_t k = 0;
                const int r1 = mi355x_tab_solve(mn, 1, 1024.0, 0, &k);
                if (r1 != MI_OPTIMAL && r1 != MI_CANCELLED) return 54;
                if (mi355x_tab_solve(mn, 1, 1024.0, 0, &k) != MI_OPTIMAL || k != 0) return 55;
            }
            mi355x_tab_destroy(mn);
            mi355x_tab_destroy(art);
            mi355x_problem_destroy(q);
        }
        printf("solved on the GPU: w = %g, x = %g\n", w, x);
    }
    mi355x_problem_destroy(p);
    printf("c abi ok\n");
    return 0;
}
