"""Test-only engine with the same driver-facing protocol as mcintegration_jl_amd.Engine, backed by the
CPU oracle.  It lets the N>1 host logic of integrate() (block partition, packed-buffer reduction,
identical train on every rank, Result) run on CPU with torch.distributed/gloo.  Never used by the product."""
import numpy as np

import mci_oracle as O
from mcintegration_jl_amd._lib import SOLVERS
from mcintegration_jl_amd.variables import ContinuousVar


class OracleEngine:
    def __init__(self, config, integrand, measure=None, device=0, **kw):
        self.config = config
        leaves = []
        for lf, pool in zip(config.leaves, config.leaf_pool):
            if isinstance(lf, ContinuousVar):
                leaves.append(dict(kind=0, pool=pool, lower=lf.lower, upper=lf.upper, npts=lf.ninc, alpha=lf.alpha,
                                   adapt=lf.adapt, grid=lf._grid0))
            else:
                leaves.append(dict(kind=1, pool=pool, lower=lf.lower, upper=lf.upper, alpha=lf.alpha, adapt=lf.adapt,
                                   distribution=lf._dist0))
        self.ocfg = O.Config(leaves, config.dof, obs_nbin=config.obs_nbin, obs_bin_draw=config.obs_bin_draw(measure))
        if kw.get("rng_bits"):
            self.ocfg.set_rng_bits(kw["rng_bits"])
        if kw.get("rng_rounds"):
            O.set_rng_rounds(kw["rng_rounds"])   # (process-wide in the oracle)
        if isinstance(integrand.name, int):
            self.fn = integrand.name
        else:
            try:
                self.fn = O.builtin(integrand.name)
            except KeyError:
                self.fn = O.builtin(integrand.name.rstrip("0123456789"))  # gaussian16 -> gaussian
        self.ud = integrand.userdata if len(integrand.userdata) else None
        self.nobs = self.ocfg.nobs
        self.packed_size = self.ocfg.packed_size()
        self._packed = np.zeros(self.packed_size)
        self.calls = []

    def set_reweight_goal(self, goal):
        self.ocfg.set_reweight_goal(goal)
        self._goal = None if goal is None else np.ascontiguousarray(goal, dtype=np.float64)

    def run(self, solver, nevalperblock, lo, hi, iteration, seed, measurefreq=1, nchain=0, thermal_ratio=0.1):
        solver = SOLVERS[solver]   # names or codes, like Engine.run
        self.calls.append((lo, hi, iteration))
        self.ocfg.set_thermal_ratio(thermal_ratio)
        self._packed = self.ocfg.iteration(int(solver), self.fn, self.ud, nevalperblock, lo, hi, iteration, seed,
                                           measurefreq=measurefreq, nchain=max(int(nchain), 1))

    def reduce(self):
        pass

    def get_packed(self):
        return self._packed.copy()

    def set_packed(self, a):
        self._packed = np.array(a, dtype=np.float64)

    def finish(self, solver, block_total, adapt=True, gamma=1.0, want_stats=True):
        solver = SOLVERS[solver]
        c = self.ocfg.c
        nstat = 2 * self.nobs + 2 + c.Ni + 1
        off = nstat
        for i in range(c.nleaf):  # reduced histograms -> config (bcastConfig!, configuration.jl:323-343)
            lf = c.leaf[i]
            np.ctypeslib.as_array(lf.hist, shape=(lf.nbin,))[:] = self._packed[off:off + lf.nbin]
            off += lf.nbin
        if int(solver) in (O.VEGASMC, O.MCMC):
            vis = self._packed[nstat - (c.Ni + 1):nstat]
            r = O.do_reweight(self.ocfg.reweight, vis, gamma, getattr(self, "_goal", None))
            np.ctypeslib.as_array(c.reweight, shape=(c.Ni + 1,))[:] = r
        if adapt:
            self.ocfg.train()
        return O.mean_std(self._packed[:self.nobs], self._packed[self.nobs:2 * self.nobs], block_total)

    def grid(self, i):
        return self.ocfg.grid(i)

    def set_grid(self, i, g):
        self.ocfg.set_grid(i, g)

    def distribution(self, i):
        return self.ocfg.distribution(i), self.ocfg.accumulation(i)

    def reweight(self):
        return self.ocfg.reweight
