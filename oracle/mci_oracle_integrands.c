/*
 * mci_oracle_integrands.c -- CPU ORACLE side of the integrand catalog (test infrastructure).
 *
 * Independent C restatements of the integrands the reference's tests/examples use; the product
 * carries its own HIP-source versions (mcintegration.jl_amd/catalog.py).  x is the flat draw
 * vector in the reference's draw order (pool, slot, leaf), 0-based.
 */
#include "mci_oracle.h"

#include <math.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ud[0] = D.  (2pi)^(-D/2) exp(-|x|^2/2): affine image of example/benchmark/vegas/benchmark4.jl:16-22 */
static void f_gaussian(const double *x, double *w, const double *ud) {
    int D = (int)ud[0];
    double r2 = 0.0;
    for (int d = 0; d < D; ++d) r2 += x[d] * x[d];
    w[0] = pow(2.0 * M_PI, -0.5 * D) * exp(-0.5 * r2);
}

/* example/benchmark/vegas/benchmark4.jl:16-22 verbatim semantics (4-D) */
static void f_gauss4_ref(const double *x, double *w, const double *ud) {
    (void)ud;
    double dx2 = 0.0;
    for (int d = 0; d < 4; ++d) dx2 += (x[d] - 0.5) * (x[d] - 0.5);
    w[0] = exp(-dx2 * 100.0) * 1013.2118364296088;
}

/* Genz product peak: ud = [D, a, u_0..u_{D-1}] ; prod 1/(a^-2 + (x_i-u_i)^2) */
static void f_genz_product_peak(const double *x, double *w, const double *ud) {
    int D = (int)ud[0];
    const double ia2 = 1.0 / (ud[1] * ud[1]);
    double q = 1.0; /* one division for the whole product: the same expression, in the same order, as catalog.genz_product_peak */
    for (int d = 0; d < D; ++d) {
        double t = x[d] - ud[2 + d];
        q *= ia2 + t * t;
    }
    w[0] = 1.0 / q;
}

/* test/montecarlo.jl:112-117 TestSingular1 */
static void f_log_over_sqrt(const double *x, double *w, const double *ud) {
    (void)ud;
    w[0] = log(x[0]) / sqrt(x[0]);
}

/* test/montecarlo.jl:4-9 Sphere1 */
static void f_sphere1(const double *x, double *w, const double *ud) {
    (void)ud;
    w[0] = (x[0] * x[0] + x[1] * x[1] < 1.0) ? 1.0 : 0.0;
}

/* test/montecarlo.jl:19-24 Sphere2 (two integrands, dof [[2],[3]]) */
static void f_sphere2(const double *x, double *w, const double *ud) {
    (void)ud;
    w[0] = (x[0] * x[0] + x[1] * x[1] < 1.0) ? 1.0 : 0.0;
    w[1] = (x[0] * x[0] + x[1] * x[1] + x[2] * x[2] < 1.0) ? 1.0 : 0.0;
}

/* test/montecarlo.jl:119-130 TestSingular2 */
static void f_singular2(const double *x, double *w, const double *ud) {
    (void)ud;
    w[0] = 1.0 / (1.0 - cos(x[0]) * cos(x[1]) * cos(x[2])) / (M_PI * M_PI * M_PI);
}

/* src/main.jl:64 docstring example */
static void f_x2y2(const double *x, double *w, const double *ud) {
    (void)ud;
    w[0] = x[0] * x[0] + x[1] * x[1];
}

/* test/montecarlo.jl:94-101 TestDiscrete: f(x) = x[1] */
static void f_discrete_id(const double *x, double *w, const double *ud) {
    (void)ud;
    w[0] = x[0];
}

/* test/montecarlo.jl:103-110 TestDiscrete2 */
static void f_one(const double *x, double *w, const double *ud) {
    (void)x;
    (void)ud;
    w[0] = 1.0;
}

/* test/montecarlo.jl:200-216 TestHyperSphere, ud[0] = N integrands, pool Continuous(-1,1), dof [[i+1]] */
static double volume_inverse(double d) {
    const double euler = 2.71828182845904523536028747135266249775724709369995957496696763;
    return pow(d / (2.0 * M_PI * euler), d / 2.0) * sqrt(d) * sqrt(M_PI);
}
static void f_hypersphere(const double *x, double *w, const double *ud) {
    int N = (int)ud[0];
    double _w = x[0] * x[0];
    for (int i = 1; i <= N; ++i) {
        _w += x[i] * x[i];
        w[i - 1] = _w < 1.0 ? volume_inverse((double)(i + 1)) : 0.0;
    }
}

/* example/bubble.jl:38-75.  ud = [kF, beta(scaled), me, spin, dim, Qsize, q_1..q_Qsize]
 * draws: x0=R, x1=theta, x2=phi, x3=T, x4=Ext (1-based integer stored as double) */
static double green(double tau, double omega, double beta) { /* example/bubble.jl:38-48 */
    if (tau >= 0.0)
        return omega > 0.0 ? exp(-omega * tau) / (1 + exp(-omega * beta)) : exp(omega * (beta - tau)) / (1 + exp(omega * beta));
    else
        return omega > 0.0 ? -exp(-omega * (tau + beta)) / (1 + exp(-omega * beta)) : -exp(-omega * tau) / (1 + exp(omega * beta));
}
static void f_bubble(const double *x, double *w, const double *ud) {
    double kF = ud[0], beta = ud[1], me = ud[2], spin = ud[3];
    int dim = (int)ud[4];
    double R = x[0], theta = x[1], phi = x[2], T = x[3];
    int extidx = (int)x[4];
    double r = R / (1 - R);                                     /* :56 */
    double k[3] = {r * sin(theta) * cos(phi), r * sin(theta) * sin(phi), r * cos(theta)}; /* :60 */
    double factor = 1.0 / pow(2.0 * M_PI, dim);                 /* :61 */
    factor *= r * r / ((1 - R) * (1 - R)) * sin(theta);         /* :62 */
    double q = ud[6 + (extidx - 1)];                            /* :66 extQ = [q,0,0] */
    double kq[3] = {k[0] + q, k[1], k[2]};                      /* :67 */
    double tau = T - 0.0;                                       /* :68 */
    double w1 = (k[0] * k[0] + k[1] * k[1] + k[2] * k[2] - kF * kF) / (2 * me);       /* :69 */
    double g1 = green(tau, w1, beta);                           /* :70 */
    double w2 = (kq[0] * kq[0] + kq[1] * kq[1] + kq[2] * kq[2] - kF * kF) / (2 * me); /* :71 */
    double g2 = green(-tau, w2, beta);                          /* :72 */
    w[0] = g1 * g2 * spin * factor;                             /* :74, n = 0 -> cos(0) = 1 */
}

/* BASELINE config 5 family: Ni nested unit Gaussians on a shared Continuous(0,1) pool,
 * integrand i covers the first dof_i = ud[1+i] coordinates; ud = [Ni, dof_0.., ] ;
 * f_i = prod_{d<dof_i} sqrt(200/pi)/erf(sqrt(50)/... ) is avoided: plain exp(-100 (x-1/2)^2)*sqrt(100/pi) per dim */
static void f_nested_gauss(const double *x, double *w, const double *ud) {
    int Ni = (int)ud[0];
    for (int i = 0; i < Ni; ++i) {
        int D = (int)ud[1 + i];
        double p = 1.0;
        for (int d = 0; d < D; ++d) p *= exp(-100.0 * (x[d] - 0.5) * (x[d] - 0.5)) * sqrt(100.0 / M_PI);
        w[i] = p;
    }
}

/* example/benchmark/cuba/benchmark.jl:35-46  t1..t11 on the unit cube */
static void f_cuba11(const double *x, double *w, const double *ud) {
    (void)ud;
    const double X = x[0], Y = x[1], Z = x[2];
    const double r2 = X * X + Y * Y + Z * Z;
    w[0] = sin(X) * cos(Y) * exp(Z);
    w[1] = 1.0 / ((X + Y) * (X + Y) + 0.003) * cos(Y) * exp(Z);
    w[2] = 1.0 / (3.75 - cos(M_PI * X) - cos(M_PI * Y) - cos(M_PI * Z));
    w[3] = fabs(r2 - 0.125);
    w[4] = exp(-r2);
    w[5] = 1.0 / (1.0 - X * Y * Z + 1e-10);
    w[6] = sqrt(fabs(X - Y - Z));
    w[7] = exp(-X * Y * Z);
    w[8] = X * X / (cos(X + Y + Z + 1.0) + 5.0);
    w[9] = (X > 0.5) ? 1.0 / sqrt(X * Y * Z + 1e-5) : sqrt(X * Y * Z);
    w[10] = (r2 < 1.0) ? 1.0 : 0.0;
}

mcio_integrand_fn mcio_builtin(const char *name) {
    static const struct { const char *n; mcio_integrand_fn f; } tab[] = {
        {"gaussian", f_gaussian}, {"gauss4_ref", f_gauss4_ref}, {"genz_product_peak", f_genz_product_peak},
        {"log_over_sqrt", f_log_over_sqrt}, {"sphere1", f_sphere1}, {"sphere2", f_sphere2},
        {"singular2", f_singular2}, {"x2y2", f_x2y2}, {"discrete_id", f_discrete_id}, {"one", f_one},
        {"hypersphere", f_hypersphere}, {"bubble", f_bubble}, {"nested_gauss", f_nested_gauss}, {"cuba11", f_cuba11},
    };
    for (unsigned i = 0; i < sizeof(tab) / sizeof(tab[0]); ++i)
        if (!strcmp(tab[i].n, name)) return tab[i].f;
    return 0;
}
