"""Canned integrands (HIP source) for the reference's tests, examples and the BASELINE configs.
The CPU oracle carries independent C restatements of the same functions (oracle/mci_oracle_integrands.c)."""
import math

from .integrand import Integrand


def gaussian(D):
    """(2 pi)^(-D/2) exp(-|x|^2/2): affine image of example/benchmark/vegas/benchmark4.jl:16-22 (BASELINE C2)."""
    body = """
    double r2 = 0.0;
    #pragma unroll
    for (int d = 0; d < %d; ++d) r2 += x[d] * x[d];
    w[0] = pow(2.0 * M_PI, -0.5 * %d) * exp(-0.5 * r2);""" % (D, D)
    return Integrand(body, [float(D)], "gaussian%d" % D)


def gauss4_ref():
    """example/benchmark/vegas/benchmark4.jl:16-22"""
    body = """
    double dx2 = 0.0;
    #pragma unroll
    for (int d = 0; d < 4; ++d) dx2 += (x[d] - 0.5) * (x[d] - 0.5);
    w[0] = exp(-dx2 * 100.0) * 1013.2118364296088;"""
    return Integrand(body, None, "gauss4_ref")


def genz_product_peak(D=32, a=5.0):
    """prod_i 1/(a^-2 + (x_i-u_i)^2), u_i = 0.3 + 0.4 i/(D-1)  (BASELINE C4); ud = [D, a, u...]"""
    u = [0.3 + 0.4 * i / (D - 1) for i in range(D)]
    # one division for the whole product (each factor lies in [a^-2, a^-2 + 1], so the product of D <= 64 of them stays far
    # inside the double range); the oracle's C twin evaluates the same expression in the same order
    body = """
    const double ia2 = 1.0 / (ud[1] * ud[1]);
    double q = 1.0;
    #pragma unroll
    for (int d = 0; d < %d; ++d) { const double t = x[d] - ud[2 + d]; q *= ia2 + t * t; }
    w[0] = 1.0 / q;""" % D
    return Integrand(body, [float(D), a] + u, "genz_product_peak%d" % D)


def log_over_sqrt():
    """test/montecarlo.jl:112-117 TestSingular1"""
    return Integrand("w[0] = log(x[0]) / sqrt(x[0]);", None, "log_over_sqrt")


def sphere1():
    """test/montecarlo.jl:4-9"""
    return Integrand("w[0] = (x[0] * x[0] + x[1] * x[1] < 1.0) ? 1.0 : 0.0;", None, "sphere1")


def sphere2():
    """test/montecarlo.jl:19-24 (two integrands, dof [[2],[3]])"""
    return Integrand("""
    w[0] = (x[0] * x[0] + x[1] * x[1] < 1.0) ? 1.0 : 0.0;
    w[1] = (x[0] * x[0] + x[1] * x[1] + x[2] * x[2] < 1.0) ? 1.0 : 0.0;""", None, "sphere2")


def singular2():
    """test/montecarlo.jl:119-130"""
    return Integrand("w[0] = 1.0 / (1.0 - cos(x[0]) * cos(x[1]) * cos(x[2])) / (M_PI * M_PI * M_PI);", None, "singular2")


def x2y2():
    """src/main.jl:64"""
    return Integrand("w[0] = x[0] * x[0] + x[1] * x[1];", None, "x2y2")


def discrete_id():
    """test/montecarlo.jl:94-101"""
    return Integrand("w[0] = x[0];", None, "discrete_id")


def one():
    """test/montecarlo.jl:103-110"""
    return Integrand("w[0] = 1.0;", None, "one")


def hypersphere(N=3):
    """test/montecarlo.jl:200-216"""
    body = """
    const double euler = 2.71828182845904523536028747135266249775724709369995957496696763;
    double _w = x[0] * x[0];
    #pragma unroll
    for (int i = 1; i <= %d; ++i) {
        _w += x[i] * x[i];
        const double d = (double)(i + 1);
        w[i - 1] = _w < 1.0 ? pow(d / (2.0 * M_PI * euler), d / 2.0) * sqrt(d) * sqrt(M_PI) : 0.0;
    }""" % N
    return Integrand(body, [float(N)], "hypersphere%d" % N)


def bubble_parameters(rs=1.0, beta=25.0, spin=2, Qsize=4, dim=3, me=0.5):
    """example/bubble.jl:10-22"""
    kF = (9 * math.pi / (2 * spin)) ** (1.0 / 3) / rs if dim == 3 else math.sqrt(4 / spin) / rs
    extQ = [1.5 * kF * i / (Qsize - 1) for i in range(Qsize)]
    return dict(kF=kF, beta=beta / (kF ** 2 / 2 / me), me=me, spin=spin, dim=dim, Qsize=Qsize, extQ=extQ)


def bubble(**kw):
    """example/bubble.jl:38-75; draws x0=R, x1=theta, x2=phi, x3=T, x4=Ext; ud = [kF, beta, me, spin, dim, Qsize, q...]"""
    p = bubble_parameters(**kw)
    body = """
    const double kF = ud[0], beta = ud[1], me = ud[2], spin = ud[3];
    const int dim = (int)ud[4];
    const double R = x[0], theta = x[1], phi = x[2], T = x[3];
    const int extidx = (int)x[4];
    const double r = R / (1 - R);
    const double k0 = r * sin(theta) * cos(phi), k1 = r * sin(theta) * sin(phi), k2 = r * cos(theta);
    double factor = 1.0 / pow(2.0 * M_PI, (double)dim);
    factor *= r * r / ((1 - R) * (1 - R)) * sin(theta);
    const double q = ud[6 + (extidx - 1)];
    const double kq0 = k0 + q;
    const double tau = T - 0.0;
    const double w1 = (k0 * k0 + k1 * k1 + k2 * k2 - kF * kF) / (2 * me);
    const double w2 = (kq0 * kq0 + k1 * k1 + k2 * k2 - kF * kF) / (2 * me);
    // green(tau, w1, beta), tau >= 0
    const double g1 = w1 > 0.0 ? exp(-w1 * tau) / (1 + exp(-w1 * beta)) : exp(w1 * (beta - tau)) / (1 + exp(w1 * beta));
    // green(-tau, w2, beta)
    double g2;
    if (-tau >= 0.0) g2 = w2 > 0.0 ? exp(-w2 * (-tau)) / (1 + exp(-w2 * beta)) : exp(w2 * (beta + tau)) / (1 + exp(w2 * beta));
    else g2 = w2 > 0.0 ? -exp(-w2 * (-tau + beta)) / (1 + exp(-w2 * beta)) : -exp(w2 * tau) / (1 + exp(w2 * beta));
    w[0] = g1 * g2 * spin * factor;"""
    ud = [p["kF"], p["beta"], p["me"], float(p["spin"]), float(p["dim"]), float(p["Qsize"])] + list(p["extQ"])
    return Integrand(body, ud, "bubble")


def bubble_fermik(**kw):
    """test/bubble_FermiK.jl:54-74: the same polarisation with the momentum as a FermiK variable:
    vars = (T, K, Ext) -> x = [tau, k_x, k_y, k_z, ext]"""
    p = bubble_parameters(**kw)
    body = """
    const double kF = ud[0], beta = ud[1], me = ud[2], spin = ud[3];
    const double tau = x[0];
    const double k0 = x[1], k1 = x[2], k2 = x[3];
    const int ext = (int)x[4] - 1;
    const double q = ud[6 + ext];
    const double w1 = (k0 * k0 + k1 * k1 + k2 * k2 - kF * kF) / (2.0 * me);
    const double kq0 = k0 + q;
    const double w2 = (kq0 * kq0 + k1 * k1 + k2 * k2 - kF * kF) / (2.0 * me);
    double g1, g2;
    if (tau >= 0.0) g1 = w1 > 0.0 ? exp(-w1 * tau) / (1 + exp(-w1 * beta)) : exp(w1 * (beta - tau)) / (1 + exp(w1 * beta));
    else g1 = w1 > 0.0 ? -exp(-w1 * (tau + beta)) / (1 + exp(-w1 * beta)) : -exp(-w1 * tau) / (1 + exp(w1 * beta));
    if (-tau >= 0.0) g2 = w2 > 0.0 ? exp(w2 * tau) / (1 + exp(-w2 * beta)) : exp(w2 * (beta + tau)) / (1 + exp(w2 * beta));
    else g2 = w2 > 0.0 ? -exp(-w2 * (-tau + beta)) / (1 + exp(-w2 * beta)) : -exp(w2 * tau) / (1 + exp(w2 * beta));
    w[0] = g1 * g2 * spin / (8.0 * M_PI * M_PI * M_PI);"""
    ud = [p["kF"], p["beta"], p["me"], float(p["spin"]), float(p["dim"]), float(p["Qsize"])] + list(p["extQ"])
    return Integrand(body, ud, "bubble_fermik")


def nested_gauss(dofs=(3, 6, 9, 12)):
    """BASELINE C5 family: integrand i = prod_{d<dof_i} sqrt(100/pi) exp(-100 (x_d-1/2)^2)"""
    lines = ["double p = 1.0;"]
    prev = 0
    for i, D in enumerate(dofs):
        lines.append("#pragma unroll\n    for (int d = %d; d < %d; ++d) p *= exp(-100.0 * (x[d] - 0.5) * (x[d] - 0.5)) * sqrt(100.0 / M_PI);" % (prev, D))
        lines.append("w[%d] = p;" % i)
        prev = D
    return Integrand("\n    ".join(lines), [float(len(dofs))] + [float(d) for d in dofs], "nested_gauss")


def cuba11():
    """The 11-component 3-D test set of example/benchmark/cuba/benchmark.jl:35-46 (t1..t11), dof = [[3]]*11:
    the only workload the reference publishes a wall time for (benchmark.jl:119-120, :146-147)."""
    body = """
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
    w[10] = (r2 < 1.0) ? 1.0 : 0.0;"""
    return Integrand(body, None, "cuba11")


BY_NAME = dict(log_over_sqrt=log_over_sqrt, sphere1=sphere1, sphere2=sphere2, singular2=singular2, x2y2=x2y2,
               discrete_id=discrete_id, one=one, gauss4_ref=gauss4_ref)
