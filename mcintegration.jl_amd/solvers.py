"""The reference's solver seam, `Vegas.montecarlo` / `VegasMC.montecarlo` / `MCMC.montecarlo` (src/vegas/montecarlo.jl:72-191,
src/vegas_mc/montecarlo.jl:112-241, src/mcmc/montecarlo.jl:72-184): ONE statistical block of `neval` evaluations on a Configuration,
which comes back with what `_block!` reads off it (src/main.jl:253-287) -- `observable`, `normalization`, `neval`, `visited`, `propose`,
`accept`, every variable's `histogram`.  `integrate()` is this, `block` times per iteration inside one launch, plus the merge and
`train!`; the seam is here for code that drives the solvers itself, the way the reference's own tests and older scripts do
(test/test2.jl:34)."""
import numpy as np

from ._lib import MCMC as _MCMC, VEGAS as _VEGAS, VEGASMC as _VEGASMC  # noqa: F401


class _Solver:
    def __init__(self, name, doc):
        self.name = name
        self.__doc__ = doc

    def __repr__(self):
        return "<solver :%s>" % self.name

    def montecarlo(self, config, integrand, neval, print=0, timer=None, debug=False, *, measure=None, measurefreq=1, inplace=False,
                   thermal_ratio=0.1, trace=None, device=None, nchain=0, engine_factory=None):
        """One block: clearStatistics! (configuration.jl:238-250), initialize! the pools, `neval` evaluations, and the block's sums
        left on `config` (returned).  Same positional and keyword arguments as the reference (`print`, `timer`, `debug` are accepted
        for the call's shape); `thermal_ratio` is :mcmc's.  Engine extras: `trace`, `device`, `nchain` (0: automatic; 1: the
        reference's one chain per block), `engine_factory` (test seam, like integrate's)."""
        from .integrate import TRACE_DEFAULT, _bind
        assert int(measurefreq) > 0                                                   # vegas/montecarlo.jl:77
        if self.name == "mcmc" and inplace:
            raise ValueError("MCMC.montecarlo has no inplace form (mcmc/montecarlo.jl:72-75)")
        eng = _bind(config, integrand, measure, self.name, inplace=inplace, trace=TRACE_DEFAULT if trace is None else trace, print=-1,
                    device=0 if device is None else device, engine_factory=engine_factory)
        eng.run(self.name, int(neval), 0, 1, config.iterations_done, config.seed, int(measurefreq), int(nchain), float(thermal_ratio))
        pk = eng.get_packed()
        config.iterations_done += 1            # (the next block draws from its own Philox stream, like the reference's advancing rng)
        if hasattr(eng, "check_status"):
            eng.check_status()                 # a block whose normalization is not positive raises here like main.jl:269-271
        nobs, N = eng.nobs, config.N
        norm = float(pk[2 * nobs])
        mean = pk[:nobs]                       # one block: its mean observable / normalization (main.jl:275-287)
        flat = mean * norm
        obs, off = [], 0
        for ln, nb, is_arr, shape in zip(config.obs_len, config.obs_nbin, config.obs_is_array, config.obs_shape):
            v = flat[off:off + nb]
            v = (v[0::2] + 1j * v[1::2]) if config.ncomp == 2 else v.copy()
            obs.append(v.reshape(shape) if is_arr else v[0])
            off += nb
        config.observable = obs
        config.normalization = norm
        config.neval = int(round(pk[2 * nobs + 1]))
        config.visited = pk[2 * nobs + 2: 2 * nobs + 2 + N + 1].copy()
        config._last_solver = self.name
        return config


Vegas = _Solver("vegas", "the :vegas solver (reference module Vegas, src/vegas/Vegas.jl)")
VegasMC = _Solver("vegasmc", "the :vegasmc solver (reference module VegasMC, src/vegas_mc/VegasMC.jl)")
MCMC = _Solver("mcmc", "the :mcmc solver (reference module MCMC, src/mcmc/MCMC.jl)")
