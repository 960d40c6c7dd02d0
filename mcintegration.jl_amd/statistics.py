"""`Result`, `average`, `report`  (reference src/statistics.jl) -- arithmetic done by the library's
host functions (mci_average / mci_mean_std)."""
import ctypes as C

import numpy as np

from ._lib import c_double_p, lib


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def mean_std(obs_sum, obs_sq, block):
    """_mean_std (main.jl:296-320)"""
    s = np.ascontiguousarray(np.atleast_1d(obs_sum), dtype=np.float64)
    q = np.ascontiguousarray(np.atleast_1d(obs_sq), dtype=np.float64)
    m, e = np.empty_like(s), np.empty_like(s)
    lib().mci_mean_std(_dp(s), _dp(q), len(s), int(block), _dp(m), _dp(e))
    return m, e


def average(iter_mean, iter_std, init=1, max=None):
    """average(history, idx; init, max) (statistics.jl:186-220) for one scalar series; 1-based init/max."""
    m = np.ascontiguousarray(iter_mean, dtype=np.float64)
    e = np.ascontiguousarray(iter_std, dtype=np.float64)
    if max is None:
        max = len(m)
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    lib().mci_average(_dp(m), _dp(e), 1, int(init), int(max), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def lineage_sums(block_mean, iter_std, init=1, max=None):
    """mci_lineage_sums: per observable, sum and sum of squares over the blocks of every block's weighted average over iterations
    init..max (1-based, the weights of `average`); block_mean[niter][nblocks][nobs], iter_std[niter][nobs]"""
    bm = np.ascontiguousarray(block_mean, dtype=np.float64)
    e = np.ascontiguousarray(iter_std, dtype=np.float64)
    niter, nb, nobs = bm.shape
    if max is None:
        max = niter
    s1, s2 = np.zeros(nobs), np.zeros(nobs)
    lib().mci_lineage_sums(_dp(bm), niter, nb, nobs, _dp(e), int(init), int(max), _dp(s1), _dp(s2))
    return s1, s2


def chain_estimator_bias(solver, steps_per_block, chains_per_block, nblocks, ncounted, propose, accept, ndraw):
    """What the ratio estimator of a chain solver's block costs when blocks are short (not in the reference; it is the reference's own
    chain that is being described).  A block mean is a RATIO of two sums over one correlated chain (main.jl:275-287): its bias is
    O(tau / N) for a block of N steps and autocorrelation time tau, its scatter O(1 / sqrt(N)).  Averaging B blocks over I counted
    iterations shrinks the scatter by sqrt(B I) and the bias not at all, so the bias of the final estimate in units of its error bar
    grows like  z = k tau sqrt(B I / N):  many short blocks are the bad direction, "fewer blocks" the reference's own knob (z ~ B at fixed neval).
      tau  = (ndraw + 1) (2 - a) / a     a = accepted / proposed updates of the last iteration (config.accept / config.propose): a step moves
                                         one of ndraw slots (or the integrand index), and keeps what it has with probability 1 - a
      k    = 0.6 (:mcmc), 0.12 (:vegasmc)   fitted to the measured mean deviation per run of one-chain-per-block calls, 16 seeds each
                                         (profiles/r06_bias.txt; tools/chain_bias_note.py): :mcmc on BASELINE configs[4] +6.8 at block = 256 /
                                         neval = 1e6 (z = 7.0), +5.1 at 64 / 1e5 (5.6), +2.3 at 16 / 1e4 (4.8); on the 2-D / 3-D spheres +3.8 (3.2),
                                         +2.7 (2.5), +1.35 (2.1); x^2 + y^2 +1.3 (1.3); :vegasmc -- every integrand evaluated at every step,
                                         numerator and normalisation strongly correlated -- a fifth of that: +0.4 .. +1.1 on configs[4]
    Returns dict(z, tau, times = N / tau, acceptance, steps_per_block, nblocks) for one chain per block, else None (a block of several
    chains averages over them first: the many-chain decomposition has its own floors, DESIGN "Chains")."""
    if solver not in ("vegasmc", "mcmc") or int(chains_per_block) != 1 or steps_per_block <= 0:
        return None
    pr, ac = float(np.sum(propose)), float(np.sum(accept))
    if not (pr > 0.0 and ac > 0.0):
        return None
    a = min(ac / pr, 1.0)
    tau = (int(ndraw) + 1) * (2.0 - a) / a
    z = (0.6 if solver == "mcmc" else 0.12) * tau * float(np.sqrt(max(int(nblocks) * max(int(ncounted), 1), 1) / float(steps_per_block)))
    return dict(z=z, tau=tau, times=float(steps_per_block) / tau, acceptance=a, steps_per_block=int(steps_per_block), nblocks=int(nblocks), solver=solver)


class Result:
    """statistics.jl:16-63.  mean/stdev/chi2 are lists with one entry per integrand: a float for scalar
    observables, an ndarray for array observables (like `obs=[zeros(4)]`).
    `block_mean` ([niter][local blocks][nobs]) with `correlated=True`: the iterations continued each other's chains (carried chains of
    the many-chain decomposition), so the error is the scatter of the blocks' weighted averages over the run (mci_lineage_sums) instead of
    statistics.jl:198, which assumes independent iterations.  A multi-rank run hands over the block means of ALL ranks' blocks
    (integrate() gathers them once, with every rank taking part), so a Result is plain data: `with_ignore` is local, talks to no
    communicator and keeps no engine alive."""

    def __init__(self, iter_mean, iter_std, config, ignore, neval=0, seconds=0.0, block_mean=None, correlated=False, block=None):
        self.iter_mean = np.asarray(iter_mean)   # [niter, nobs]
        self.iter_std = np.asarray(iter_std)
        self.config, self.ignore, self.neval, self.seconds = config, int(ignore), int(neval), seconds
        self.block_mean, self.correlated, self.block = block_mean, bool(correlated), block
        niter, nobs = self.iter_mean.shape
        flat = [average(self.iter_mean[:, o], self.iter_std[:, o], init=ignore + 1, max=niter) for o in range(nobs)]
        self._flat_mean = np.array([f[0] for f in flat])
        self._flat_std = np.array([f[1] for f in flat])
        self._flat_chi2 = np.array([f[2] for f in flat])
        if self.correlated and block_mean is not None and niter > ignore + 1:
            s1, s2 = lineage_sums(block_mean, self.iter_std, init=ignore + 1, max=niter)
            le = mean_std(s1, s2, block if block else np.asarray(block_mean).shape[1])[1]
            self._flat_std = np.where(le > 0.0, le, self._flat_std)   # (an identically-zero column keeps the reference's 1e-10-regularised error)
        # The reference combines the iterations with weights 1 / sigma_i^2 (statistics.jl:186-220).  On heavy-tailed integrands an
        # iteration's error estimate is correlated with its mean (a rare large sample raises both), and the weighted average is then
        # biased (log(x)/sqrt(x) under :vegasmc: +4.5 sigma pooled over 16 seeds, the plain mean +0.9; profiles/r04_validation_matrix.txt).
        # Reproduced, not corrected -- but said: `weighting_shift` = |weighted average - plain mean of the counted iterations| in units
        # of their combined error, per statistics column; report() prints a note where it exceeds 2.
        self.weighting_shift = np.zeros(nobs)
        counted = self.iter_mean[ignore:]
        # (not in the reference) the plain mean of the counted iterations and its scatter error, shaped like `mean` / `stdev`: what the
        # note of report() quotes -- unbiased where the 1/sigma_i^2 weights are not, noisier where they are fine; None below 3 iterations
        self.plain_mean = self.plain_stdev = None
        if counted.shape[0] >= 3:
            um, ue = counted.mean(0), counted.std(0, ddof=1) / np.sqrt(counted.shape[0])
            den = np.hypot(self._flat_std, ue)
            self.weighting_shift = np.where(den > 0.0, np.abs(self._flat_mean - um) / np.where(den > 0.0, den, 1.0), 0.0)
            self.plain_mean, self.plain_stdev = self._shape(um), self._shape(ue)
        # set by integrate() for a chain solver that ran the reference's one chain per block: chain_estimator_bias (report() prints a
        # note where the expected bias of the blocks' ratio estimator reaches 2 of the final error bars)
        self.chain_bias = None
        self.mean, self.stdev, self.chi2 = self._shape(self._flat_mean), self._shape(self._flat_std), self._shape(self._flat_chi2)
        self.iterations = [(self._shape(self.iter_mean[i]), self._shape(self.iter_std[i]), config) for i in range(niter)]

    def _shape(self, flat):
        """flat statistics columns -> one entry per integrand; complex types: re + im*1j, the standard deviation
        likewise (statistics.jl:207-214, main.jl:302-305)"""
        out, off = [], 0
        nc = getattr(self.config, "ncomp", 1)
        shapes = getattr(self.config, "obs_shape", None) or [None] * len(self.config.obs_nbin)
        for nb, isarr, shape in zip(self.config.obs_nbin, self.config.obs_is_array, shapes):
            v = np.array(flat[off:off + nb])
            if nc == 2:
                v = v[0::2] + 1j * v[1::2]
            if isarr and shape is not None and len(shape) > 1:
                v = v.reshape(shape)               # an N-d observable comes back with its axes (row-major, like the closure indexed it)
            out.append(v if isarr else (complex(v[0]) if nc == 2 else float(v[0])))
            off += nb
        return out

    def with_ignore(self, ignore):
        """Result(res, ignore) (statistics.jl:56-62)"""
        if ignore == self.ignore:
            return self
        r = Result(self.iter_mean, self.iter_std, self.config, ignore, self.neval, self.seconds, self.block_mean, self.correlated, self.block)
        r.chain_bias = self.chain_bias
        return r

    @property
    def chain_bias_note(self):
        """the sentence report() prints under the tables, or None"""
        b = self.chain_bias
        if not b or b["z"] < 2.0:
            return None
        return ("note: solver = :%s ran one chain per block of %d steps -- about %.0f autocorrelation times (acceptance %.2f) -- in %d blocks; a block mean "
                "is a ratio of two sums over that chain (main.jl:275-287), and its bias does not average out over blocks and iterations: expect the estimate "
                "about %.0f of its error bars off.  The reference's own knob: fewer blocks (block = %d gives %.1f)" % (
                    b["solver"], b["steps_per_block"], b["times"], b["acceptance"], b["nblocks"], b["z"], max(b["nblocks"] // 16, 1),
                    b["z"] / (b["nblocks"] / max(b["nblocks"] // 16, 1))))

    @property
    def dof(self):
        return (len(self.iterations) - (self.ignore + 1) + 1) - 1   # statistics.jl:65-68

    def __getitem__(self, idx):
        return self.mean[idx], self.stdev[idx], self.chi2[idx]

    def __repr__(self):
        lines = []
        for i in range(self.config.N):
            m, e, c2 = np.ravel(self.mean[i])[0], np.ravel(self.stdev[i])[0], np.ravel(self.chi2[i])[0]
            if self.dof == 0:
                lines.append("Integral %d = %s ± %s" % (i + 1, m, e))
            else:
                lines.append("Integral %d = %s ± %s   (reduced chi2 = %.3g)" % (i + 1, m, e, c2))
        return "\n".join(lines)


def _sig_digits(err):
    if err == 0 or not np.isfinite(err):
        return 0
    return max(0, 2 - int(np.floor(np.log10(abs(err)))))   # statistics.jl:74-79


def _tostring(m, e):
    if np.isfinite(m) and np.isfinite(e):
        nd = _sig_digits(e)
        return "%.*f ± %.*f" % (nd, m, nd, e)             # statistics.jl:87-96
    return "%s ± %s" % (m, e)


def report_config(config, io=None):
    """report(config) (configuration.jl:345-464): the ChangeIntegrand / ChangeVariable / SwapVariable acceptance tables per
    (integrand, target) from config.propose / config.accept, then visited and reweight.  Same rows, order and number formats
    as the reference (which prints every table whatever the solver: under :vegasmc only ChangeVariable row 1 is populated,
    vegas_mc/updates.jl:90; under :vegas nothing is)."""
    import sys
    import datetime
    from .variables import CompositeVar, ContinuousVar, DiscreteVar, FermiK
    io = io or sys.stdout
    eng = config._engine
    bar = "-" * 85
    Nd = config.N + 1
    print("", file=io)
    print("===========================  Configuration  =========================================", file=io)
    print(datetime.datetime.now().isoformat(sep="T", timespec="milliseconds"), file=io)
    print(bar, file=io)
    print("Integral num = %d, dof = %s, with variables:" % (config.N, config.dof), file=io)
    for vi, v in enumerate(config.var):
        print("%d. %r" % (vi + 1, v), file=io)
    print(bar, file=io)
    neval = max(config.neval, 1)
    if eng is not None and hasattr(eng, "acceptance"):
        pr, ac = eng.acceptance()
    else:
        m = max(Nd, len(config.var))
        pr, ac = np.full((3, Nd, m), 1.0e-8), np.zeros((3, Nd, m))                   # configuration.jl:185-186
    nbrs = config.neighbor_lists()
    if nbrs is None:                                                                 # the default graph, configuration.jl:203-208
        nbrs = [[i - 1, i + 1] for i in range(Nd)]
        nbrs[0] = [1] if Nd == 2 else [Nd - 1, 1]
        nbrs[Nd - 1] = [0]
        if Nd >= 3:
            nbrs[Nd - 2] = [Nd - 3]

    def row(label, u, i, j):
        print("%s %11.6f%% %11.6f%% %12.6f" % (label, pr[u, i, j] / neval * 100.0, ac[u, i, j] / neval * 100.0, ac[u, i, j] / pr[u, i, j]), file=io)

    print("%-20s %12s %12s %12s" % ("ChangeIntegrand", "Proposed", "Accepted", "Ratio  "), file=io)
    for n in nbrs[Nd - 1]:                                                           # :369-378
        row("Norm -> %2d:          " % (n + 1), 0, Nd - 1, n)
    for idx in range(Nd - 1):                                                        # :379-399
        for n in nbrs[idx]:
            if n == Nd - 1:
                row("  %d ->Norm:          " % (idx + 1), 0, idx, n)
            else:
                row("  %d -> %2d:           " % (idx + 1, n + 1), 0, idx, n)
    print(bar, file=io)
    for u, title in ((1, "ChangeVariable"), (2, "SwapVariable")):                    # :402-452
        print("%-20s %12s %12s %12s" % (title, "Proposed", "Accepted", "Ratio  "), file=io)
        for idx in range(Nd - 1):
            for vi, v in enumerate(config.var):
                typestr = "Continuous" if isinstance(v, ContinuousVar) else "Discrete" if isinstance(v, DiscreteVar) else \
                    "Composite" if isinstance(v, CompositeVar) else "FermiK" if isinstance(v, FermiK) else type(v).__name__
                row("  %2d / %-11s:  " % (idx + 1, typestr), u, idx, vi)
        print(bar, file=io)
    print("Integrand            Visited      ReWeight", file=io)
    vis = getattr(config, "visited", None)
    rw = config.reweight
    if vis is None:
        vis = np.zeros(config.N + 1)
    print("  Norm   :     %12d %12.6f" % (vis[-1], rw[-1]), file=io)
    for idx in range(config.N):
        print("  Order%2d:     %12d %12.6f" % (idx + 1, vis[idx], rw[idx]), file=io)
    print(bar, file=io)
    print("Integrand evaluation = %d\n" % config.neval, file=io)


def report(result, ignore=None, pick=0, name=None, verbose=0, io=None):
    """report(result) (statistics.jl:137-172): per-iteration table with the running weighted average;
    report(config) (configuration.jl:345-464) when given a Configuration."""
    import sys
    io = io or sys.stdout
    if not isinstance(result, Result):
        return report_config(result, io=io)
    ignore = result.ignore if ignore is None else ignore
    off = 0
    for i in range(result.config.N):
        col = off + pick
        off += result.config.obs_nbin[i]
        info = str(i + 1) if name is None else str(name[i])
        if verbose >= 0:
            bar = "-" * 127
            print("=" * 48 + "     Integral %s    " % info + "=" * 60, file=io)
            print("%6s                 %-32s                 %-32s %22s" % ("iter", "         integral", "        wgt average", "reduced chi2"), file=io)
            print(bar, file=io)
            for it in range(result.iter_mean.shape[0]):
                m0, e0 = result.iter_mean[it, col], result.iter_std[it, col]
                m, e, c2 = average(result.iter_mean[:, col], result.iter_std[:, col], init=ignore + 1, max=it + 1)
                label = "ignore" if it + 1 <= ignore else str(it + 1)
                print("%6s %36s %36s %16.4f" % (label, _tostring(m0, e0), _tostring(m, e), abs(c2)), file=io)
            print(bar, file=io)
            if getattr(result, "weighting_shift", None) is not None and result.weighting_shift[col] > 2.0:   # (not in the reference)
                print("  note: the weighted average lies %.1f sigma from the plain mean of the counted iterations (%s): weights 1/sigma_i^2 are "
                      "biased when an iteration's error estimate moves with its mean (heavy tails); more evaluations per iteration shrink both" % (
                          result.weighting_shift[col], _tostring(float(result.iter_mean[ignore:, col].mean()),
                                                                 float(result.iter_mean[ignore:, col].std(ddof=1) / np.sqrt(max(result.iter_mean.shape[0] - ignore, 1))))), file=io)
            if getattr(result, "correlated", False):   # (not in the reference: its iterations are independent)
                print("  the iterations continued each other's chains: block-lineage error of the average  %s%s" % (
                    _tostring(result._flat_mean[col], result._flat_std[col]),
                    "   (%d warm-up launches run again)" % result.warmup if getattr(result, "warmup", 0) else ""), file=io)
        else:
            print("Integral %s = %s ± %s" % (info, result._flat_mean[col], result._flat_std[col]), file=io)
    if getattr(result, "chain_bias_note", None):   # (not in the reference)
        print("  " + result.chain_bias_note, file=io)
