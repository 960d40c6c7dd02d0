"""Variable pools: Continuous / Discrete / CompositeVar  (reference src/distribution/variable.jl).

These are host-side descriptions; the live grids/distributions are device-resident inside an Engine
(read back through `.grid` / `.distribution` once a Configuration has been bound to one)."""
import numpy as np

MaxOrder = 16  # reference src/distribution/distribution.jl:59


class _Leaf:
    _engine = None
    _leaf_index = None


class ContinuousVar(_Leaf):
    """`Continuous(lower, upper, size=MaxOrder; offset=0, alpha=2.0, adapt=true, ninc=1000, grid=...)`
    reference variable.jl:137-153.  N = ninc-1 increments (999 by default)."""

    def __init__(self, lower, upper, size=MaxOrder, *, offset=0, alpha=2.0, adapt=True, ninc=1000, grid=None):
        assert offset + 1 < size                      # variable.jl:138
        assert upper > lower + 2 * np.finfo(float).eps  # variable.jl:140
        self.lower, self.upper = float(lower), float(upper)
        self.size = size + 1                          # one cache slot, variable.jl:139
        self.offset, self.alpha, self.adapt = int(offset), float(alpha), bool(adapt)
        self._grid0 = None if grid is None else np.ascontiguousarray(grid, dtype=np.float64)
        self.ninc = len(self._grid0) if grid is not None else int(ninc)

    @property
    def range(self):
        return self.upper - self.lower

    @property
    def grid(self):
        if self._engine is not None:
            return self._engine.grid(self._leaf_index)
        if self._grid0 is not None:
            return self._grid0.copy()
        return np.linspace(self.lower, self.upper, self.ninc)

    @property
    def histogram(self):
        """what the last iteration accumulated for this map (reference field `histogram`, variable.jl:95, :196-200): one entry per
        increment; the 1e-10 of clearStatistics (variable.jl:565) before anything has run"""
        if self._engine is not None:
            return np.array(self._engine.histogram(self._leaf_index))
        return np.full(self.ninc - 1, 1.0e-10)

    def __repr__(self):
        return "%s Continuous variable in [%g, %g).%s" % ("Adaptive" if self.adapt else "Nonadaptive", self.lower,
                                                          self.upper, " Learning rate = %g." % self.alpha if self.adapt else "")


class DiscreteVar(_Leaf):
    """`Discrete(lower, upper, size=MaxOrder; distribution=nothing, offset=0, alpha=2.0, adapt=true)`
    reference variable.jl:299-325."""

    def __init__(self, lower, upper, size=MaxOrder, *, distribution=None, offset=0, alpha=2.0, adapt=True):
        assert offset + 1 < size
        assert upper >= lower                          # variable.jl:304
        self.lower, self.upper = int(lower), int(upper)
        self.size = size + 1
        self.offset, self.alpha, self.adapt = int(offset), float(alpha), bool(adapt)
        if distribution is not None:
            distribution = np.ascontiguousarray(distribution, dtype=np.float64)
            assert np.all(distribution >= 0.0), "distribution should be all non-negative!"   # variable.jl:309
            assert len(distribution) == self.upper - self.lower + 1                          # variable.jl:310
        self._dist0 = distribution

    @property
    def distribution(self):
        if self._engine is not None:
            return self._engine.distribution(self._leaf_index)[0]
        k = self.upper - self.lower + 1
        d = np.ones(k) if self._dist0 is None else self._dist0
        return d / d.sum()

    @property
    def accumulation(self):
        if self._engine is not None:
            return self._engine.distribution(self._leaf_index)[1]
        return np.concatenate([[0.0], np.cumsum(self.distribution)])

    @property
    def histogram(self):
        """what the last iteration accumulated per value (reference field `histogram`, variable.jl:283, :362-367)"""
        if self._engine is not None:
            return np.array(self._engine.histogram(self._leaf_index))
        return np.full(self.upper - self.lower + 1, 1.0e-10)

    def __repr__(self):
        return "%s Discrete variable in [%d, ..., %d]." % ("Adaptive" if self.adapt else "Nonadaptive", self.lower, self.upper)


class CompositeVar:
    """`CompositeVar(vargs...; adapt=true, offset=0, size=MaxOrder)` reference variable.jl:415-427:
    a product of variables sampled together; the bundled leaves inherit adapt and offset."""

    def __init__(self, *vars, adapt=True, offset=0, size=MaxOrder):
        assert all(isinstance(v, (ContinuousVar, DiscreteVar)) for v in vars), "all arguments should variables"  # :416-417
        for v in vars:
            v.adapt, v.offset = bool(adapt), int(offset)  # :419-420
        self.vars = tuple(vars)
        self.adapt, self.offset, self.size = bool(adapt), int(offset), size

    def __len__(self):
        return len(self.vars)

    def __getitem__(self, i):
        return self.vars[i]

    def __iter__(self):
        return iter(self.vars)


class FermiK(_Leaf):
    """`FermiK(dim, kF, dk, maxK, size=MaxOrder; offset=0)` reference variable.jl:1-20: a dim-dimensional (2 | 3)
    momentum on the shell |k| in (kF - dk, kF + dk).  No adaptive map (train! is a no-op, :557); a slot contributes
    `dim` consecutive entries of the integrand's x.  solver=:mcmc only (test/bubble_FermiK.jl:2)."""

    def __init__(self, dim, kF, dk, maxK, size=MaxOrder, *, offset=0):
        assert offset + 1 < size                       # variable.jl:12
        assert dim in (2, 3)
        self.dim, self.kF, self.dk, self.maxK = int(dim), float(kF), float(dk), float(maxK)
        self.lower, self.upper = self.kF, self.dk      # how the C ABI carries them (include/mci.h)
        self.size = size + 1
        self.offset, self.alpha, self.adapt = int(offset), float(maxK), False

    def __repr__(self):
        return "%dD FermiK variable in [0, %g)." % (self.dim, self.maxK)   # variable.jl:22-27


def _is_bounds(x):
    return isinstance(x, (list, tuple)) and len(x) > 0 and isinstance(x[0], (list, tuple))


def Continuous(lower, upper=None, size=MaxOrder, **kw):
    """Continuous(lower, upper, ...) or Continuous([(lo, hi), ...], ...) -> CompositeVar (variable.jl:174-187;
    the multi-bound form always uses 1000-point grids, :177)."""
    if _is_bounds(lower):
        bounds = lower
        if upper is not None and not isinstance(upper, (list, tuple)):
            size = upper
        adapt, offset = kw.get("adapt", True), kw.get("offset", 0)
        grids = kw.get("grid", [None] * len(bounds))
        vs = [ContinuousVar(b[0], b[1], size, offset=offset, alpha=kw.get("alpha", 2.0), adapt=adapt, ninc=1000,
                            grid=grids[i]) for i, b in enumerate(bounds)]
        return CompositeVar(*vs, adapt=adapt, offset=offset, size=size)
    return ContinuousVar(lower, upper, size, **kw)


def Discrete(lower, upper=None, size=MaxOrder, **kw):
    """Discrete(lower, upper, ...), Discrete((lower, upper)) (variable.jl:326-328) or
    Discrete([(lo, hi), ...]) -> CompositeVar (variable.jl:342-353)."""
    if _is_bounds(lower):
        bounds = lower
        if upper is not None:
            size = upper
        adapt, offset = kw.get("adapt", True), kw.get("offset", 0)
        dists = kw.get("distribution", [None] * len(bounds))
        vs = [DiscreteVar(b[0], b[1], size, offset=offset, alpha=kw.get("alpha", 2.0), adapt=adapt,
                          distribution=dists[i]) for i, b in enumerate(bounds)]
        return CompositeVar(*vs, adapt=adapt, offset=offset, size=size)
    if isinstance(lower, (list, tuple)) and upper is None:
        return DiscreteVar(lower[0], lower[1], size, **kw)
    if isinstance(lower, (list, tuple)):
        return DiscreteVar(lower[0], lower[1], upper, **kw)
    return DiscreteVar(lower, upper, size, **kw)


def is_variable(v):
    """reference: Dist.is_variable (test/variable.jl:7-16)"""
    if isinstance(v, type):
        return v in (ContinuousVar, DiscreteVar, CompositeVar, FermiK)
    return isinstance(v, (ContinuousVar, DiscreteVar, CompositeVar, FermiK)) or v in (Continuous, Discrete)


def poolsize(v):
    return v.size


# ---- the reference's exported helpers of `train!` (src/utility/utility.jl:19; src/distribution/common.jl), for host code that wants them:
# the engine itself smooths, rescales and bisects inside its kernels (csrc/mci_train.h, mci_device.h) ----------------------------------
def locate(accumulation, p):
    """index i (1-based like the reference's) with accumulation[i] <= p < accumulation[i + 1]; an error outside (common.jl:8-36)"""
    a = np.asarray(accumulation, dtype=np.float64)
    if a[0] > p or a[-1] <= p:
        raise ValueError("%r is not in %r" % (p, accumulation))                       # common.jl:10-12
    return int(np.searchsorted(a, p, side="right"))


def smooth(dist, factor=6.0):
    """each entry averaged with its two neighbours, 1 : factor : 1; the ends (factor + 1) : 1 (common.jl:43-54)"""
    d = np.asarray(dist, dtype=np.float64)
    if len(d) <= 1:
        return d.copy()
    new = np.empty_like(d)
    new[0] = (d[0] * (factor + 1) + d[1]) / (factor + 2)
    new[-1] = (d[-1] * (factor + 1) + d[-2]) / (factor + 2)
    new[1:-1] = (d[:-2] + d[1:-1] * factor + d[2:]) / (factor + 2)
    return new


def rescale(dist, alpha=1.5):
    """normalise, then d -> (-(1 - d) / log d)^alpha where 0 < d <= 0.99999999; the result is NOT normalised again (common.jl:67-82)"""
    d = np.array(dist, dtype=np.float64)
    if len(d) == 1:
        return d
    assert np.all(d > 0), "distribution should be all positive and non-zero"          # common.jl:71
    d /= d.sum()
    m = (d > 0) & (d <= 0.99999999)
    d[m] = (-(1.0 - d[m]) / np.log(d[m])) ** alpha
    assert np.all(np.isfinite(d)), "distribution is not all finite"                   # common.jl:79
    return d
