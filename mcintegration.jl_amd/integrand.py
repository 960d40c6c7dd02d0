"""The integrand closure of the reference (vegas/montecarlo.jl:140-144) as HIP C++ source."""
import numpy as np


class Integrand:
    """body: HIP C++ statements with `const double* x` (flat draws, reference draw order: pool, slot, leaf;
    0-based), `double* w` (one output per integrand) and `const double* ud` (userdata) in scope.
    A body without `w[` is treated as an expression/return form: `return expr;` -> w[0] = expr."""

    def __init__(self, body, userdata=None, name=None):
        body = body.strip()
        if "w[" not in body:
            expr = body[len("return"):].strip().rstrip(";") if body.startswith("return") else body.rstrip(";")
            body = "w[0] = (%s);" % expr
        self.body = body
        self.userdata = np.ascontiguousarray([] if userdata is None else userdata, dtype=np.float64)
        self.name = name or "user"


class bin_by:
    """measure(vars, obs, weights, config): obs[i][Ext[1]] += weights[i]  (example/bubble.jl:81-84):
    observable bins are selected by the value of the Discrete pool `pool` (its first slot)."""

    def __init__(self, pool, slot=0, leaf=0):
        self.pool, self.slot, self.leaf = pool, slot, leaf


class Measure:
    """The `measure` callback (vegas/montecarlo.jl:156-161; mcmc/montecarlo.jl:166-169) as HIP C++ source.
    In scope: `x` (draws), `rw` (relative weights, one per integrand; (re, im) pairs for complex types), `ud`,
    `idx` (-1, or the current integrand of an mcmc chain) and `obs_add(k, v)`: obs.flat[k] += v, where the flat
    index runs over the `obs` kwarg in order (a complex entry takes two slots: re, im)."""

    def __init__(self, body):
        self.body = body.strip()


class HostIntegrand:
    """A Python closure as integrand -- the reference's `integrand(var, config)` (vegas/montecarlo.jl:140-144) run on
    the HOST, vectorised over the batch ("batch callback" slow path, include/mci.h mci_set_integrand_host):

        f(x, config) -> array[n] | tuple of arrays (one per integrand; complex arrays for type=complex)

    With one variable type `x[i]` is the vector of the i-th draw over the n samples of the batch (0-based; the
    reference's `x[i+1]`); with several, `x` is a tuple with one such array per variable type (a CompositeVar pool is
    indexed [leaf][slot] like the reference's, `x, y = cvar`; a FermiK pool [slot][component]).  solver="vegas": n = the samples of a launch, one call per launch; solver="vegasmc" (the reference's
    default) and "mcmc": n = the chains of a launch, one call per Markov step (the chains advance in lock step).

    indexed=True: the reference's `:mcmc` form `integrand(idx, var, config)` (mcmc/montecarlo.jl:34-36) --

        f(idx, x, config) -> array[m]

    is called once per integrand index that some chain needs, with `x` restricted to those m chains (idx is 0-based: the
    reference's idx - 1); include/mci.h mci_set_integrand_host_indexed.

    inplace=True: the reference's `inplace = true` form `integrand(var, weights, config)` (main.jl:26, vegas/montecarlo.jl:140-141,
    vegas_mc/updates.jl:67-70) --

        f(x, weights, config)          # weights[i] = ...   in place; what f returns is ignored

    `weights` is a writable [N, n] array (complex for type=complex), zero on entry: `weights[i] = expr` stores integrand i's values
    over the batch (0-based).  For real weights it is the library's own pinned buffer (mci_set_integrand_host hands the callback
    its output array: the C boundary is in-place already), so nothing is copied.

    integrate() picks the form the way the reference does -- by solver and the `inplace` keyword; wrapping a closure in
    HostIntegrand(fn, indexed=..., inplace=...) says it explicitly, and then any form works under every solver."""

    def __init__(self, fn, name=None, indexed=False, inplace=False):
        if indexed and inplace:
            raise ValueError("the :mcmc form integrand(idx, var, config) has no in-place variant (main.jl:26-28)")
        self.fn = fn
        self.indexed = bool(indexed)
        self.inplace = bool(inplace)
        self.name = name or getattr(fn, "__name__", "host")
        self.body = "/* host integrand %d%s */" % (id(fn), " indexed" if indexed else " inplace" if inplace else "")
        self.userdata = np.zeros(0)


class HostMeasure:
    """A Python closure as `measure` -- the reference's `measure(vars, obs, relative_weights, config)` (vegas/montecarlo.jl:156-161,
    vegas_mc/montecarlo.jl:224-227) run on the HOST, once per statistical block and vectorised over the block's records ("batch
    callback" slow path, include/mci.h mci_set_measure_host):

        m(x, obs, weights, config)     # obs[i] += ...   in place

    `x` as for HostIntegrand (x[i] = the vector of the i-th draw over the block's records), `weights[i]` the vector of integrand
    i's relative weights (complex for type=complex), `obs` a list shaped like the `obs` keyword (floats are 1-element arrays),
    zeroed for every block.  A record is a sample under solver="vegas" (weights zero for samples that `measurefreq` skips) and a
    measured step of one of the block's chains under "vegasmc" / "mcmc" (under "mcmc" only the integrand the chain sits on has a
    non-zero weight).

    indexed=True: the reference's `:mcmc` form `measure(idx, var, obs, relative_weight, config)` (mcmc/montecarlo.jl:166-169) --

        m(idx, x, obs, weight, config)

    is called once per integrand index (0-based) with the records that belong to it; include/mci.h mci_set_measure_host_indexed.
    Either form works under every solver."""

    def __init__(self, fn, name=None, indexed=False):
        self.fn = fn
        self.indexed = bool(indexed)
        self.name = name or getattr(fn, "__name__", "host_measure")
        self.body = "/* host measure %d%s */" % (id(fn), " indexed" if indexed else "")
