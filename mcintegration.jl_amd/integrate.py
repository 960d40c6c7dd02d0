"""`integrate(integrand; solver, config, neval, niter, block, ...)`  reference src/main.jl:71-218."""
import ctypes as C
import time

import numpy as np

from . import catalog
from ._lib import MCMC, SOLVERS, VEGAS, VEGASMC, lib
from .comm import LocalComm
from .configuration import Configuration
from .engine import Engine
from .integrand import HostIntegrand, HostMeasure, Integrand, Measure
from .statistics import Result, chain_estimator_bias, report
from .variables import Continuous, Discrete


def standardize_block(neval, nblock, nworker=1):
    """_standardize_block (main.jl:220-234)"""
    assert neval > nblock, "neval=%s should be larger than nblock = %s" % (neval, nblock)   # :222
    a, b = C.c_int64(), C.c_int64()
    lib().mci_standardize_block(int(neval), int(nblock), int(nworker), C.byref(a), C.byref(b))
    return a.value, b.value


def required_positionals(fn, fallback):
    """positional parameters of a closure that have NO default -- what decides between the reference's two callback forms
    (`integrand(var, config)` vegas/montecarlo.jl:140-144 | `integrand(idx, var, config)` mcmc/montecarlo.jl:34-36; likewise
    `measure`).  A defaulted or keyword-only parameter is the closure's own business; `fallback` for callables without a signature."""
    import inspect
    try:
        return len([q for q in inspect.signature(fn).parameters.values()
                    if q.kind in (q.POSITIONAL_ONLY, q.POSITIONAL_OR_KEYWORD) and q.default is q.empty])
    except (TypeError, ValueError):
        return fallback


def _positional_range(fn):
    """(required, most) positional arguments the closure takes; most = None with *args; None when it has no signature"""
    import inspect
    try:
        ps = list(inspect.signature(fn).parameters.values())
    except (TypeError, ValueError):
        return None
    pos = [q for q in ps if q.kind in (q.POSITIONAL_ONLY, q.POSITIONAL_OR_KEYWORD)]
    return len([q for q in pos if q.default is q.empty]), (None if any(q.kind == q.VAR_POSITIONAL for q in ps) else len(pos))


_INTEGRAND_CALL = {"plain": "integrand(var, config)", "inplace": "integrand(var, weights, config)", "indexed": "integrand(idx, var, config)"}
_MEASURE_CALL = {"plain": "measure(var, obs, relative_weights, config)", "indexed": "measure(idx, var, obs, relative_weight, config)"}


def callback_form(fn, solver, inplace=False, form=None, what="integrand"):
    """Which of the reference's callback forms a bare closure is called in -- decided like the reference decides it, by the SOLVER and
    the `inplace` keyword, never by counting parameters (main.jl:26-28, :38-40):

        solver = "mcmc"                -> integrand(idx, var, config)             mcmc/montecarlo.jl:34-36
        otherwise, inplace = True      -> integrand(var, weights, config)         vegas/montecarlo.jl:140-141, vegas_mc/updates.jl:67-70
        otherwise                      -> integrand(var, config)                  vegas/montecarlo.jl:142-143, vegas_mc/updates.jl:71-75
        measure: solver = "mcmc"       -> measure(idx, var, obs, relative_weight, config)   mcmc/montecarlo.jl:166-169
                 otherwise             -> measure(var, obs, relative_weights, config)       vegas/montecarlo.jl:156-161

    The closure's own parameter count is only a cross-check: one that cannot be called that way raises TypeError here (the
    reference's MethodError at the first call) instead of being read as another form.  `form` ("plain" | "inplace" | "indexed";
    integrate's integrand_form / measure_form keywords) overrides the rule: an engine extension -- every form runs under every solver."""
    calls = _INTEGRAND_CALL if what == "integrand" else _MEASURE_CALL
    if form is None:
        form = "indexed" if solver == "mcmc" else "inplace" if (inplace and what == "integrand") else "plain"
    elif form not in calls:
        raise ValueError("%s_form = %r: one of %s" % (what, form, sorted(calls)))
    want = calls[form].count(",") + 1
    rng = _positional_range(fn)
    if rng is not None and (rng[0] > want or (rng[1] is not None and rng[1] < want)):
        has = "%d" % rng[0] if rng[1] == rng[0] else "%d to %s" % (rng[0], "any number of" if rng[1] is None else rng[1])
        hint = ""
        if what == "integrand":
            hint = ("  The form follows the solver and the `inplace` keyword (reference src/main.jl:26-28): solver = \"vegas\" / \"vegasmc\" call "
                    "integrand(var, config), with inplace = True integrand(var, weights, config); solver = \"mcmc\" calls integrand(idx, var, config).  "
                    "integrand_form = \"plain\" | \"inplace\" | \"indexed\" forces a form under any solver.")
        else:
            hint = ("  The form follows the solver (reference src/main.jl:38-40): solver = \"mcmc\" calls measure(idx, var, obs, relative_weight, config), "
                    "the others measure(var, obs, relative_weights, config).  measure_form = \"plain\" | \"indexed\" forces a form under any solver.")
        raise TypeError("solver = %r%s calls %s -- %d arguments -- but the %s closure %s takes %s positional argument%s.%s"
                        % (solver, ", inplace = True" if (inplace and what == "integrand" and solver != "mcmc") else "", calls[form], want, what,
                           getattr(fn, "__name__", "?"), has, "" if has == "1" else "s", hint))
    return form


TRACE_DEFAULT = None   # what integrate(trace=None) means: None = trace Python closures where possible, silently; False = host callbacks


def _not_traced(what, err, trace, verbosity):
    """a closure that could not be written out as device source keeps the host callback path; say why when it was asked for"""
    msg = "%s not traced (%s): host callback path" % (what, err)
    if trace:
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)
    elif verbosity > 0:
        import builtins
        builtins.print(msg)


def _bind(config, integrand, measure, solver, *, inplace=False, trace=None, print=-1, device=0, engine_factory=None, rng_bits=52,
          rng_rounds=10, deterministic=False, integrand_form=None, measure_form=None):
    """The engine of `config` for this integrand and measure (created, or kept when nothing it was built for has changed): closures are
    put into the form the solver calls them in and traced where they can be; grids and reweight factors trained so far survive a change
    of integrand.  Shared by integrate() and the solver seam (Vegas / VegasMC / MCMC .montecarlo)."""
    if isinstance(integrand, str):
        integrand = Integrand(integrand, config.userdata)
    elif callable(integrand) and not isinstance(integrand, (Integrand, HostIntegrand)):
        # a Python closure, called in the form the reference's solver calls it in (callback_form; main.jl:26-28): traced into device
        # source where possible, else the host "batch callback" path (vegas: per launch, vegasmc / mcmc: per Markov step).
        # HostIntegrand(fn, indexed=..., inplace=...) / trace_integrand(...) say the form explicitly.
        form = callback_form(integrand, solver, inplace, integrand_form)
        traced = None
        if trace is None or trace:
            from .trace import TraceError, trace_integrand
            try:
                traced = trace_integrand(integrand, config, indexed=form == "indexed", inplace=form == "inplace")
            except TraceError as e:
                _not_traced("integrand", e, trace, print)
        integrand = traced if traced is not None else HostIntegrand(integrand, indexed=form == "indexed", inplace=form == "inplace")
    if callable(measure) and not isinstance(measure, (Measure, HostMeasure)) and not hasattr(measure, "pool"):
        # a Python closure as measure: the reference's :mcmc form measure(idx, var, obs, relative_weight, config) under solver = "mcmc"
        # (mcmc/montecarlo.jl:166-169), measure(var, obs, weights, config) under the others (vegas/montecarlo.jl:156-161)
        mindexed = callback_form(measure, solver, form=measure_form, what="measure") == "indexed"
        tmeasure = None
        if trace is None or trace:
            from .trace import TraceError, trace_measure
            try:
                tmeasure = trace_measure(measure, config, indexed=mindexed)
            except TraceError as e:
                _not_traced("measure", e, trace, print)
        measure = tmeasure if tmeasure is not None else HostMeasure(measure, indexed=mindexed)
    mkey = None if measure is None else measure.body if isinstance(measure, (Measure, HostMeasure)) else (measure.pool, measure.slot, measure.leaf)
    key = (integrand.body, tuple(integrand.userdata), mkey, device,
           repr(config.neighbor), int(rng_bits), int(rng_rounds), bool(deterministic))
    if config._engine is None or config._engine_key != key:
        # grids trained so far survive a change of integrand (`var = (res.config.var[1], ...)`, docs/src/index.md:129)
        # and so does the learned reweight (config.reweight lives across integrate calls, configuration.jl:50)
        old = config._engine
        saved, saved_rw = None, None
        if old is not None:
            saved = [(old.grid(i) if hasattr(lf, "ninc") else None if hasattr(lf, "kF") else old.distribution(i)[0])
                     for i, lf in enumerate(config.leaves)]
            saved_rw = old.reweight()
        eng = (engine_factory or Engine)(config, integrand, measure=measure, device=device, **({"rng_bits": rng_bits} if rng_bits != 52 else {}),
                                         **({"rng_rounds": rng_rounds} if rng_rounds != 10 else {}), **({"deterministic": True} if deterministic else {}))
        if saved is not None:
            for i, lf in enumerate(config.leaves):
                if saved[i] is not None:   # (a FermiK has nothing trained)
                    (eng.set_grid if hasattr(lf, "ninc") else eng.set_distribution)(i, saved[i])
            if hasattr(eng, "set_reweight"):
                eng.set_reweight(saved_rw)
            if hasattr(old, "close"):
                old.close()                # its device buffers go now, not at some later garbage collection
        config._engine, config._engine_key = eng, key
        if getattr(config, "_pending_state", None):
            eng.load_state(config._pending_state)
            config._pending_state = None
    return config._engine


def integrate(integrand, *, solver="vegasmc", config=None, neval=1e4, niter=10, block=16, verbose=-1, gamma=1.0,
              adapt=True, debug=False, reweight_goal=None, ignore=None, measure=None, measurefreq=1,
              thermal_ratio=0.1, inplace=False, parallel="nothread", print=-1, printio=None, timer=None,
              comm=None, device=None, nchain=0, engine_factory=None, rng_bits=52, rng_rounds=10, deterministic=False, trace=None,
              integrand_form=None, measure_form=None, **kwargs):
    """Same keywords as the reference (main.jl:71-90; unknown ones go to Configuration, :95-97).
    Extra, engine-specific keywords: `comm` (LocalComm | RcclComm | TorchDistComm), `device`, `nchain`
    (vegasmc chains per block; 0 = auto), `rng_bits` (52 | 32: opt-in cheaper uniform stream of solver="vegas", see
    mci_set_rng_bits), `rng_rounds` (10 | 7: opt-in Philox4x32-7 for every stream, mci_set_rng_rounds), `deterministic` (bit-identical
    results for a fixed seed like the reference's sequential loop, mci_set_deterministic), `trace` (None, the default, and True: a
    Python closure as integrand or measure is run once on symbolic draws and written out as device source -- trace.trace_integrand /
    trace_measure -- so that it runs inside the kernels like Julia's inlined closure does in the reference's loop; a closure that
    cannot be written out takes the host batch-callback path, silently with None, with a RuntimeWarning naming the reason with True;
    False: always the host path), `integrand_form` / `measure_form` (callback_form: by default a closure is called in the form the
    reference's solver calls it in -- `inplace` included -- and one whose parameters do not fit raises TypeError), `engine_factory`
    (test seam)."""
    if trace is None:
        trace = TRACE_DEFAULT
    if solver in (":vegas", ":vegasmc", ":mcmc"):
        solver = solver[1:]
    if solver not in SOLVERS:
        raise ValueError("Solver %s is not supported!" % solver)                      # main.jl:263
    print = max(print, verbose)                                                       # main.jl:93
    if config is None:
        config = Configuration(**kwargs)                                              # main.jl:95-97
    for mx, v in zip(config.maxdof, config.var):
        assert mx + 2 <= v.size, "maxdof should be less than the length of var"      # main.jl:99-101
    if ignore is None:
        ignore = 1 if adapt else 0                                                    # main.jl:82
    comm = comm or LocalComm()
    if device is None:   # an RcclComm is bound to one device: the engine has to live there (its all_reduce checks it)
        device = getattr(comm, "device", 0)
    neval = int(neval)
    nevalperblock, block = standardize_block(neval, block, comm.size)                 # main.jl:121
    assert block % comm.size == 0                                                     # main.jl:122
    per = block // comm.size
    lo, hi = per * comm.rank, per * (comm.rank + 1)

    eng = _bind(config, integrand, measure, solver, inplace=inplace, trace=trace, print=print, device=device, engine_factory=engine_factory,
                rng_bits=rng_bits, rng_rounds=rng_rounds, deterministic=deterministic, integrand_form=integrand_form, measure_form=measure_form)
    s = SOLVERS[solver]
    if hasattr(eng, "set_reweight_goal"):
        eng.set_reweight_goal(reweight_goal)                                          # main.jl:81, :334-337

    t0 = time.time()
    means, stds = [], []
    neval_done = 0
    warmup = 0                             # launches run again instead of being counted (automatic :mcmc chain lengths)
    neval_discarded = 0
    block_mean, correlated = None, False   # chain solvers: every block's mean of every iteration | the iterations continued each other's chains
    if type(comm) is LocalComm and engine_factory is None and hasattr(eng, "integrate") and getattr(eng, "comm_ranks", lambda: 0)() == 1:
        # one process: the whole loop runs inside the library (mci_integrate: the iterations are queued back to back on the
        # engine's stream and the statistics of all of them are read back once -- 22 us per launch-bound iteration instead of
        # 82 us with a host round trip per iteration, tools/call_overhead.py).  Same iterations, same numbers as the loop below.
        r = eng.integrate(s, nevalperblock * block, niter=niter, block=block, ignore=ignore, adapt=adapt, gamma=gamma,
                          measurefreq=measurefreq, seed=config.seed, nchain=nchain, first_iteration=config.iterations_done,
                          thermal_ratio=thermal_ratio, reweight_goal=reweight_goal)
        means, stds = list(r["iter_mean"]), list(r["iter_std"])
        block_mean, correlated, warmup = r.get("block_mean"), r.get("correlated", False), r.get("warmup", 0)
        neval_discarded = r.get("neval_discarded", 0)
        neval_done = nevalperblock * block * niter
        niter_loop = 0
        config.visited = r["visited"]    # config.visited of the last iteration (configuration.jl:46), for report(config)
    else:
        niter_loop = niter
        if s != VEGAS and hasattr(eng, "reset_block_log"):
            eng.reset_block_log()
    for it in range(niter_loop):                                                      # main.jl:142
        attempt = 0
        if hasattr(eng, "set_iteration_counted"):
            eng.set_iteration_counted(it >= ignore)                                   # (main.jl:82, :211; what mci_integrate tells its launches)
        while True:
            eng.run(s, nevalperblock, lo, hi, config.iterations_done + it + 16384 * attempt, config.seed, measurefreq, nchain, thermal_ratio)   # main.jl:152-166
            comm.all_reduce(eng)                                                      # main.jl:177-188
            fin_solver = s                                                            # doReweight! runs on the device (main.jl:183)
            m, e = eng.finish(fin_solver, block, adapt, gamma)                        # main.jl:190-203
            # warm-up of the automatic :mcmc chain length, like mci_integrate: an iteration whose chains were too short for the holds
            # they measured is run again (longer chains, new streams) instead of being counted, until the first one that is long enough
            if (s != MCMC or nchain > 0 or not hasattr(eng, "mcmc_launch_valid") or (it == 0 and ignore >= 1) or attempt >= 7
                    or config.iterations_done + it >= 16384 or eng.last_chain_launch()[0] <= 1):
                break
            valid, warm, _, _ = eng.mcmc_launch_valid()
            if valid or warm:
                break
            eng.discard_iteration()
            attempt += 1
            warmup += 1
            neval_discarded += nevalperblock * block
        means.append(m)
        stds.append(e)
        neval_done += nevalperblock * block
    if niter_loop and hasattr(eng, "set_iteration_counted"):
        eng.set_iteration_counted(False)
    config.iterations_done += niter
    config.neval = nevalperblock * block
    config._last_solver = solver
    if niter_loop and hasattr(eng, "get_packed"):   # config.visited of the last iteration (configuration.jl:46), for report(config)
        try:
            pk = eng.get_packed()
            config.visited = pk[2 * eng.nobs + 2: 2 * eng.nobs + 2 + config.N + 1].copy()
        except Exception:
            pass
    if niter_loop and s != VEGAS and hasattr(eng, "block_means"):
        block_mean, ncarried = eng.block_means(niter)
        correlated = ncarried > 0
    if comm.size > 1 and block_mean is not None:
        # every rank holds its own blocks' means: gathered ONCE, here, where all ranks take part (a sum of arrays that are zero outside the
        # rank's own blocks), so that the Result is plain data and Result.with_ignore never enters a collective
        full = np.zeros((np.asarray(block_mean).shape[0], block, np.asarray(block_mean).shape[2]))
        full[:, lo:hi, :] = block_mean
        block_mean = np.asarray(comm.sum_host(eng, full.ravel())).reshape(full.shape)
    res = Result(np.array(means), np.array(stds), config, ignore, neval=neval_done, seconds=time.time() - t0, block_mean=block_mean,
                 correlated=correlated, block=block)   # main.jl:211
    if s != VEGAS and hasattr(eng, "last_chain_launch") and hasattr(eng, "acceptance"):
        try:   # one chain per block: what its ratio estimator costs at this block length (statistics.chain_estimator_bias; a note of report())
            pr, ac = eng.acceptance()
            res.chain_bias = chain_estimator_bias(solver, nevalperblock, eng.last_chain_launch()[0], block, niter - ignore, pr, ac, eng.ndraw)
        except Exception:
            res.chain_bias = None
    res.warmup = warmup   # launches that were run again instead of being counted (automatic :mcmc chain lengths)
    res.neval_discarded = neval_discarded   # ... and their evaluations: spent (they trained the map), in neither res.neval nor the estimate
    if print >= 0:
        report(res, io=printio)                                                       # main.jl:212-213
    return res


def prefill_kernel_cache():
    """Compile (hiprtc, gfx950, no GPU needed) the sample-batch kernels of the BASELINE configs and of the
    test battery into the in-tree kernel cache, so that the GPU box starts from code objects."""
    import math
    L = math.sqrt(50.0)
    jobs = [
        (Configuration(var=Continuous(0.0, 1.0), dof=[[1]]), catalog.log_over_sqrt(), None),                      # C1
        (Configuration(var=Continuous(-L, L), dof=[[16]]), catalog.gaussian(16), None),                           # C2 shared pool
        (Configuration(var=Continuous([(-L, L)] * 16), dof=[[1]]), catalog.gaussian(16), None),                   # C2 16 grids
        (Configuration(var=Continuous([(0.0, 1.0)] * 32), dof=[[1]]), catalog.genz_product_peak(32), None),       # C4
    ]
    p = catalog.bubble_parameters()
    from .integrand import bin_by
    var = (Continuous(0.0, 1.0, alpha=3.0), Continuous(0.0, math.pi, alpha=3.0), Continuous(0.0, 2 * math.pi, alpha=3.0),
           Continuous(0.0, p["beta"], alpha=3.0), Discrete(1, 4, adapt=False))
    jobs.append((Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)]), catalog.bubble(), bin_by(4)))  # C3
    n = 0
    for cfg, f, meas in jobs:
        eng = Engine(cfg, f, measure=meas, device=-1)
        eng.compile("vegas")
        if meas is not None:
            eng.compile("vegasmc")  # C3 is a :vegasmc config
            eng.compile("vegasmc_lanes")   # ... whose launches of few chains (the example's neval = 1e6) give every chain a group of lanes
        eng.close()
        n += 1
    # C5: 4 integrands on a 12-D pool, :mcmc
    eng = Engine(Configuration(var=Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]]), catalog.nested_gauss(), device=-1)
    eng.compile("mcmc")
    eng.compile("mcmc_lanes")   # (the pilot-length first launch of a cold call runs few, long chains)
    eng.close()
    return n + 1
