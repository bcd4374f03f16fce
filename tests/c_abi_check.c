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
        {   /* the multi-device entry points fail the same way (and never touch RCCL) */
            mi355x_colpart *cp = NULL;
            if (mi355x_colpart_create(&cp, rows, cols, M, basis, 8) != MI_NO_DEVICE || cp != NULL) return 14;
            if (mi355x_colpart_solve(NULL, 1, 1024.0, 0, NULL) != MI_BAD_ARG) return 15;
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
        printf("solved on the GPU: w = %g, x = %g\n", w, x);
    }
    mi355x_problem_destroy(p);
    printf("c abi ok\n");
    return 0;
}
