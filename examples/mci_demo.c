/* mci_demo.c -- the C ABI (include/mci.h) used from plain C, no Python, no Julia:
 *     integrate((x, c) -> log(x[1]) / sqrt(x[1]); solver = :vegas, neval = 1e5)          (reference README.md:26)
 * build:  gcc -O2 -Iinclude examples/mci_demo.c -Lmcintegration.jl_amd/lib -lmci_hip -Wl,-rpath,'$ORIGIN/../mcintegration.jl_amd/lib' -lm
 * exit code: 0 ok, 7 no GPU (MCI_ERR_NO_DEVICE: there is no CPU fallback), other = failing status. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "mci.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != MCI_OK) {                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mci_last_error());     \
            return rc_;                                                          \
        }                                                                        \
    } while (0)

int main(int argc, char **argv) {
    const long neval = argc > 1 ? atol(argv[1]) : 100000;
    mci_ctx *ctx = NULL;
    CHECK(mci_ctx_create(0, &ctx));

    /* Configuration(var = Continuous(0.0, 1.0), dof = [[1]]) */
    mci_leaf_desc leaf = {MCI_CONTINUOUS, 0, 0.0, 1.0, 1000, 2.0, 1, NULL};
    const int32_t dof[1] = {1};
    mci_problem_desc desc = {1, &leaf, 1, 1, dof, NULL, NULL, NULL, NULL, 1};
    mci_problem *prob = NULL;
    CHECK(mci_problem_create(ctx, &desc, &prob));
    CHECK(mci_set_integrand_source(prob, "w[0] = log(x[0]) / sqrt(x[0]);", NULL, 0));

    enum { NITER = 10 };
    double iter_mean[NITER], iter_std[NITER], mean, stdev, chi2;
    mci_result res = {NITER, 1, iter_mean, iter_std, &mean, &stdev, &chi2, 0, 0.0, NULL};
    mci_integrate_args args = {MCI_VEGAS, neval, NITER, 16, -1, 1, 1.0, 1, 20240229u, 0, 0, 0.1, NULL};
    CHECK(mci_integrate(prob, &args, &res));

    for (int it = 0; it < NITER; ++it) printf("iter %2d  %.6f +- %.6f\n", it + 1, iter_mean[it], iter_std[it]);
    printf("Integral 1 = %.6f +- %.6f   (reduced chi2 = %.3g)   %.1f ms for %lld evaluations\n", mean, stdev, chi2,
           res.seconds * 1e3, (long long)res.neval);
    CHECK(mci_problem_destroy(prob));
    CHECK(mci_ctx_destroy(ctx));
    return fabs(mean + 4.0) < 7.0 * stdev ? 0 : 100; /* exact: -4 */
}
