"""`Configuration(; var, dof, obs, reweight, seed, userdata, ...)`  reference src/configuration.jl:105-194."""
import numpy as np

from .integrand import HostMeasure, Measure, bin_by
from .variables import CompositeVar, ContinuousVar, DiscreteVar, FermiK


def _normalize_dof(dof, nvar):
    """reference configuration.jl:134-151"""
    if isinstance(dof, (int, np.integer)):            # one integral with one variable   :134-136
        assert nvar == 1, "Only one type of variable is allowed when dof is an integer"
        return [[int(dof)]]
    if isinstance(dof, np.ndarray) and dof.ndim == 2:  # each column is a dof for one integral   :139-140
        return [[int(v) for v in dof[:, i]] for i in range(dof.shape[1])]
    dof = list(dof)
    if all(isinstance(d, (int, np.integer)) for d in dof):   # [2, 3] -> one integrand per element   :146-147
        return [[int(d)] for d in dof]
    if all(isinstance(d, (list, tuple, np.ndarray)) for d in dof):  # :142-145
        return [[int(v) for v in d] for d in dof]
    raise TypeError("Configuration.dof should be a Vector{Int} or Tuple{Int, ..., Int} or Vector{Vector{Int}} "
                    "or Vector{Tuple{Int, ..., Int}} to avoid mistakes.")   # :149


class Configuration:
    """Holds the problem description; `bind()` attaches the device-resident Engine state (trained grids
    survive across `integrate(...; config=res.config)` calls, docs/src/index.md:129)."""

    def __init__(self, var=None, dof=None, type=float, obs=None, reweight=None, seed=None, neighbor=None,
                 userdata=None, **kwargs):
        from .variables import Continuous
        if var is None:
            var = (Continuous(0.0, 1.0),)                                  # :106
        if isinstance(var, (ContinuousVar, DiscreteVar, CompositeVar, FermiK)):    # :116-117
            var = (var,)
        var = tuple(var)                                                   # :120-122
        assert all(isinstance(v, (ContinuousVar, DiscreteVar, CompositeVar, FermiK)) for v in var), \
            "All elements in var should be derived from the abstract type Variable"   # :119
        self.var = var
        nv = len(var)
        if dof is None:
            dof = [[1] * nv]                                               # :107
        self.dof = _normalize_dof(dof, nv)
        assert all(len(d) == nv for d in self.dof), "Each element of `dof` should have the same dimension as `var`"  # :166
        self.N = len(self.dof)
        assert self.N >= 1, "At least one integrand is required."        # :163
        self.maxdof = [max(d[v] for d in self.dof) for v in range(nv)]     # :229-236 (the dof=0 normalisation row never wins)
        for v, mx in zip(var, self.maxdof):                                # :156-160 resize pools to maxdof+2
            if mx + v.offset >= v.size - 2:
                v.size = mx + 2 + v.offset
                if isinstance(v, CompositeVar):
                    for leaf in v.vars:
                        leaf.size = v.size
        self.type = type
        self.ncomp = 2 if (type is complex or type in ("ComplexF64", np.complex128)) else 1   # :108
        if obs is None:
            obs = [type(0) if self.ncomp == 2 else 0.0] * self.N           # :109 zeros(type, N)
        assert len(obs) == self.N, "The number of observables should be equal to the number of integrands"  # :168
        # one statistics column per double: a complex entry is (re, im) (main.jl:279,284,302-305)
        self.obs_len = [int(np.size(o)) for o in obs]
        self.obs_nbin = [n * self.ncomp for n in self.obs_len]
        self.obs_is_array = [np.ndim(o) > 0 for o in obs]
        self.obs_shape = [tuple(np.shape(o)) if np.ndim(o) > 0 else (1,) for o in obs]     # what a measure closure indexes (an N-d observable keeps its axes)
        if reweight is None:
            reweight = np.ones(self.N + 1)                                 # :110
        reweight = np.asarray(reweight, dtype=np.float64)
        assert len(reweight) == self.N + 1, "Wrong reweight vector size! Note that the last element in reweight vector is for the normalization diagram."  # :174
        assert np.all(reweight > 0), "All reweight factors should be positive."   # :175
        self._reweight0 = reweight / reweight.sum()                        # :173
        self.seed = int(np.random.SeedSequence().entropy % 1000000) + 1 if seed is None else int(seed)  # :111
        self.userdata = userdata
        self.neighbor = neighbor
        self.norm = self.N + 1                                             # :177
        self.neval = 0
        self.normalization = 1.0e-10                                       # :179
        self.iterations_done = 0     # RNG stream offset for resumed runs
        self._engine = None
        self._engine_key = None
        self._pending_state = None
        # flat leaves
        self.leaves, self.leaf_pool = [], []
        for vi, v in enumerate(var):
            for leaf in (v.vars if isinstance(v, CompositeVar) else (v,)):
                self.leaves.append(leaf)
                self.leaf_pool.append(vi)

    # ---- resume across processes: trained grids / distributions / reweight <-> MCISTATE file ----
    def save(self, path):
        assert self._engine is not None, "nothing trained yet: run integrate(...) first"
        self._engine.save_state(path)

    def load(self, path):
        """restore a state written by save(); applied when the engine is (re)created by integrate(config=...)"""
        self._pending_state = str(path)
        if self._engine is not None:
            self._engine.load_state(path)
            self._pending_state = None
        return self

    def neighbor_lists(self):
        """`neighbor` kwarg normalised like _neighbor (configuration.jl:201-227) to 0-based lists per integrand
        (index N = normalisation); None = the default chain, built by the library."""
        nb = self.neighbor
        if nb is None:
            return None
        Nd = self.N + 1
        nb = list(nb)
        if nb and all(isinstance(e, tuple) and len(e) == 2 and all(isinstance(q, (int, np.integer)) for q in e) for e in nb):
            # Vector{Tuple{Int,Int}}: undirected 1-based edges (:213-221); neighbors(g, v) come back sorted
            adj = [set() for _ in range(Nd)]
            for a, b in nb:
                assert 1 <= a <= Nd and 1 <= b <= Nd, "neighbor edge (%d, %d) out of range" % (a, b)
                adj[a - 1].add(b - 1)
                adj[b - 1].add(a - 1)
            seen, stack = {0}, [0]
            while stack:
                for j in adj[stack.pop()]:
                    if j not in seen:
                        seen.add(j)
                        stack.append(j)
            assert len(seen) == Nd, "The neighbor graph is not connected."     # :220
            return [sorted(a) for a in adj]
        assert len(nb) == Nd, "%d elements are expected for neighbor=%s" % (Nd, nb)   # :226
        out = []
        for lst in nb:                                                       # Vector{Vector{Int}}, 1-based
            lst = [int(j) - 1 for j in lst]
            assert lst and all(0 <= j < Nd for j in lst)
            out.append(lst)
        return out

    # ---- derived layout -----------------------------------------------------------------------
    def draw_index(self, pool, slot=0, leaf=0):
        """flat position of (pool, slot, leaf) in the integrand's x[] (draw order: pool, slot, leaf)"""
        k = 0
        for vi, v in enumerate(self.var):
            nl = self.pool_width(vi)
            if vi == pool:
                assert slot < self.maxdof[vi] and leaf < nl
                return k + slot * nl + leaf
            k += self.maxdof[vi] * nl
        raise IndexError(pool)

    def pool_width(self, vi):
        """x entries per slot of pool vi: the number of leaves of a CompositeVar, the dimension of a FermiK, else 1"""
        v = self.var[vi]
        return len(v.vars) if isinstance(v, CompositeVar) else v.dim if isinstance(v, FermiK) else 1

    def pool_layout(self):
        """per variable type: (first flat draw, maxdof, x entries per slot, offset, is a CompositeVar, kind of every entry of a slot) --
        kinds: "c" a Continuous draw, "d" a Discrete one (an integer: the reference's Discrete pool holds Ints, variable.jl:283), "k" a
        FermiK momentum component.  What a closure's argument is built from (trace._argument, Engine._pool_views)."""
        out, k = [], 0
        for vi, v in enumerate(self.var):
            nl = self.pool_width(vi)
            leaves = list(v.vars) if isinstance(v, CompositeVar) else [v] * nl
            kinds = "".join("k" if isinstance(lf, FermiK) else "c" if hasattr(lf, "ninc") else "d" for lf in leaves)
            out.append((k, self.maxdof[vi], nl, int(getattr(v, "offset", 0) or 0), isinstance(v, CompositeVar), kinds))
            k += self.maxdof[vi] * nl
        return out

    @property
    def ndraw(self):
        return sum(self.maxdof[vi] * self.pool_width(vi) for vi in range(len(self.var)))

    def obs_bin_draw(self, measure):
        if measure is None:
            assert all(n == 1 for n in self.obs_len), \
                "the default measure can only handle observable as Vector with N scalar elements!"   # vegas/montecarlo.jl:104
            return [-1] * self.N
        if isinstance(measure, bin_by):
            assert self.ncomp == 1, "bin_by observables are real"
            k = self.draw_index(measure.pool, measure.slot, measure.leaf)
            return [k if n > 1 else -1 for n in self.obs_nbin]
        if isinstance(measure, (Measure, HostMeasure)):
            return [-1] * self.N
        raise TypeError("measure must be None, bin_by(pool), Measure(source) or a Python callable / HostMeasure (host slow path)")

    @property
    def reweight(self):
        if self._engine is not None:
            return self._engine.reweight()
        return self._reweight0.copy()

    def reset_seed(self, seed):
        """`reset_seed!(config, seed)` (configuration.jl:196-199): the seed of every stream drawn from now on, counted from its start again"""
        self.seed = int(seed)
        self.iterations_done = 0

    def _acceptance(self, which):
        nd = self.N + 1
        shape = (3, nd, max(nd, len(self.var)))
        if self._engine is not None and hasattr(self._engine, "acceptance"):
            return self._engine.acceptance()[which]
        return np.full(shape, 1.0e-8) if which == 0 else np.zeros(shape)

    @property
    def propose(self):
        """proposed updates of the last iteration, [update type, integrand, target] (configuration.jl:58, :185; 0-based)"""
        return self._acceptance(0)

    @property
    def accept(self):
        """accepted updates of the last iteration, shaped like `propose` (configuration.jl:59, :186)"""
        return self._acceptance(1)

    def __repr__(self):
        return "Configuration for %d integrands involves %d types of variables.\nNumber of variables for each integrand: %s.\n" % (
            self.N, len(self.var), self.dof)
