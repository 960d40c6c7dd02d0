#!/usr/bin/env python3
"""bench.py -- headline benchmark: Msamples/s of the VEGAS sample-batch path on the 16-D Gaussian
(BASELINE.json configs[1]) on N MI355X, plus the MC estimate and its sigma.

    python bench.py --gpus N --steps K --warmup W            # N > 1: launches its own N ranks (one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                # ... or is launched as one of them

A "step" is one VEGAS iteration: neval samples per GPU drawn through the adaptive map, integrand
evaluated, observables + per-bin histogram accumulated, block statistics merged, one RCCL all-reduce of
the packed buffer (N > 1), grid refinement.  Warm-up iterations train the grid from uniform
(reference resume pattern, docs/src/index.md:129, test/bubble.jl:108-113); the K timed iterations
continue from the trained grid with adapt=true, ignore=0.  Weak scaling: per-GPU work is fixed
(neval_total = N * neval_per_gpu, block = N * 16); the reference fans its blocks out the same way inside
`integrate` (src/main.jl:113-122, :152-188).  Rank 0 prints ONE JSON line.

Without a GPU the launcher can still be exercised (`MCI_BENCH_ENGINE=module:factory`, used by
tests/test_bench_launcher.py with a test engine): gloo instead of RCCL, the line then carries "dry_run": true
and no throughput claim.
"""
import argparse
import importlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D = 16
L = math.sqrt(50.0)
EXACT = math.erf(L / math.sqrt(2.0)) ** D
B_ALG = 32 * D          # algorithmic bytes per sample: per dim 16 B of grid edges + 16 B histogram RMW
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
N_SIMD = 256 * 4        # 256 CUs x 4 SIMD-32
MICROBENCH = os.path.join(ROOT, "mcintegration.jl_amd", "lib", "issue_microbench")


def usable_cores():
    """host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(math.ceil(float(quota) / float(period)))))
    except Exception:
        pass
    return n


def cpu_baseline(trained_grid=None, seconds_target=12.0, nseeds=4):
    """The CPU oracle (oracle/, kind "port": the reference itself is Julia and cannot run here), threaded
    over blocks like parallel=:thread (src/main.jl:153-158) on the host cores this process may use, same 16-D
    Gaussian.  `nseeds` independent single-iteration runs continue from the grid the GPU trained (the reference's
    resume pattern, docs/src/index.md:129): every one of them is an unbiased estimate on the SAME map the GPU's timed
    iterations started from, so the GPU estimate can be tested against the set instead of against one draw."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mci_oracle as O
    cores = usable_cores()
    block = max(16, cores)
    cfg = O.Config([dict(kind=0, pool=0, lower=-L, upper=L)], [[D]])
    probe = 20000 * block
    t0 = time.time()
    cfg.integrate(O.VEGAS, "gaussian", [float(D)], neval=probe, niter=1, block=block, seed=1, nthreads=cores)
    rate = probe / max(time.time() - t0, 1e-6)
    neval = int(max(probe, min(rate * seconds_target / nseeds, 5e8)))
    per_seed, dt, done = [], 0.0, 0
    for s in range(nseeds):
        c = O.Config([dict(kind=0, pool=0, lower=-L, upper=L)], [[D]])
        if trained_grid is not None:
            c.set_grid(0, trained_grid)
        t0 = time.time()
        r = c.integrate(O.VEGAS, "gaussian", [float(D)], neval=neval, niter=1, block=block, seed=2 + s, nthreads=cores, ignore=0)
        dt += time.time() - t0
        done += (neval // block) * block
        per_seed.append([float(r["mean"][0]), float(r["stdev"][0])])
    w = [1.0 / (e * e) for _, e in per_seed]
    pooled = [sum(m * wi for (m, _), wi in zip(per_seed, w)) / sum(w), 1.0 / math.sqrt(sum(w))]
    return {"value": round(done / dt / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "estimate": pooled, "estimate_per_seed": per_seed,
            "sample": "oracle/mci_oracle.c (C restatement, OpenMP over blocks), 16-D Gaussian :vegas, %d independent runs (seeds 2..%d) of "
                      "%d samples x 1 iteration %s, block=%d, %.1f s in total; pooled estimate %.6f +- %.6f"
                      % (nseeds, 1 + nseeds, neval, "on the GPU-trained grid" if trained_grid is not None else "on the uniform grid",
                         block, dt, pooled[0], pooled[1])}


# ---------------------------------------------------------------------------------------------------------------
# launcher: `bench.py --gpus N` outside a torch.distributed launch starts its own N ranks
# ---------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """re-executes this script under torch.distributed.run with one rank per GPU (the driver's own launch line);
    the ranks inherit stdout, so rank 0's JSON line is this process's last line too"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC between the ranks' GPUs (RCCL)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------------
# roofline of the sample kernel: cycle-weighted issue rate of its own instruction mix (VALU and LDS pipes)
# ---------------------------------------------------------------------------------------------------------------
def measured_issue_costs():
    """issue cost (ns a SIMD is occupied per wave64 instruction, W waves competing) of the instruction forms the sample loop is
    made of, measured NOW on this GPU by tools/issue_microbench.hip (built in-tree by __graft_entry__.build()); the cheapest
    occupancy of each row is the pipe's cost.  Falls back to the committed table of the same tool (profiles/r02_issue_costs.json)."""
    rows, source = None, None
    if os.path.exists(MICROBENCH):
        try:
            out = subprocess.run([MICROBENCH, "2000", "roofline"], check=True, capture_output=True, text=True, timeout=300).stdout
            rows = [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]
            source = "measured in this run (tools/issue_microbench.hip)"
        except Exception as e:  # pragma: no cover
            print("[bench] issue microbenchmark failed: %s" % e, file=sys.stderr)
    if rows is None:
        path = os.path.join(ROOT, "profiles", "r02_issue_costs.json")
        if not os.path.exists(path):
            return None, None
        rows = json.load(open(path))["rows"]
        source = "profiles/r02_issue_costs.json (same tool, earlier run)"
    per_op = {}
    for r in rows:
        if "op" in r and r.get("waves_per_simd", 8) >= 4:
            # the slope between two launch lengths where the tool reports it (the fixed launch + tail time cancels), else the wall
            # figure (an upper bound on the cost)
            c = r.get("slope_ns_per_wave_inst_per_simd")
            if c is None or c <= 0:
                c = r["wall_ns_per_wave_inst_per_simd"]
            per_op.setdefault(r["op"], []).append(c)
    # 4 and 8 waves per SIMD both saturate a pipe: their mean (the minimum of two noisy slopes would be biased low)
    costs = {op: sum(v) / len(v) for op, v in per_op.items()}
    return costs, source


def issue_roofline(code_object, costs, samples_per_launch, kernel_ms, hist_copies=1, sclk_mhz=None):
    """{"valu": ..., "lds": ...}: the launch's wave-instructions on each pipe per second against the rate at which the chip
    can issue THAT mix (1024 SIMDs / mix-weighted mean issue cost).  The LDS rows are costs seen from one SIMD with all four
    SIMDs of the CU competing, so the same 1024 applies."""
    from mcintegration_jl_amd import isa_mix
    mix = isa_mix.loop_mix(code_object, "mci_vegas_batch", draws_per_sample=D)
    cyc = isa_mix.issue_cycles(mix, costs, hist_copies=hist_copies)
    trips = samples_per_launch / 64.0                       # one loop trip = one sample on each of a wave's 64 lanes
    out = {}
    for pipe in ("valu", "lds"):
        n_inst = sum(n for cls, (n, c) in cyc["per_class"].items() if cls.startswith(pipe))
        ach = n_inst * trips / (kernel_ms * 1e-3) / 1e9     # G wave-instructions/s
        peak = N_SIMD / (cyc[pipe] / n_inst) if n_inst else 0.0   # 1024 SIMDs / mean ns per instruction
        out[pipe] = {"achieved": round(ach, 2), "peak": round(peak, 2), "frac": round(ach / peak, 4) if peak else None,
                     "instructions_per_wave_sample": n_inst, "issue_ns_per_wave_sample": round(cyc[pipe], 1)}
    # the same mix at whole issue CYCLES per instruction class and the shader clock the timed launches measured themselves running at: a
    # fraction anyone can recompute from the instruction counts below -- sum(n * cycles) * wave-samples per SIMD / clock / kernel time.
    # valu_rounded_measured: the guide's 2 / 4 cycles, and 4 for the three-source / SGPR-source 32-bit forms as THIS repository's
    # microbenchmark measured them (rounded; isa_mix.DATASHEET_CYCLES -- not a published table); valu_flat_2cycle: every 32-bit form
    # at the guide's flat 2 cycles -- the pessimistic reading.  The truth lies between the two fractions.
    for key, table in (("valu_rounded_measured", None), ("valu_flat_2cycle", isa_mix.FLAT_CYCLES)):
        ds_cyc, ds_per = isa_mix.datasheet_valu_cycles(mix, table)
        out[key] = {"cycles_per_wave_sample": ds_cyc, "sclk_mhz": None if not sclk_mhz else round(sclk_mhz, 1),
                    "wave_samples_per_simd": round(trips / N_SIMD, 2),
                    "classes": {cls: {"n": n, "cycles_each": c} for cls, (n, c) in sorted(ds_per.items())}}
        if sclk_mhz:
            ns = ds_cyc / (sclk_mhz * 1e-3)                                  # ns per wave-sample on its SIMD
            out[key].update({"issue_ns_per_wave_sample": round(ns, 1), "frac": round(ns * 1e-6 * (trips / N_SIMD) / kernel_ms, 4)})
    out["mix"] = {cls: {"n": n, "ns_each": round(c, 3)} for cls, (n, c) in sorted(cyc["per_class"].items())}
    out["samples_per_loop_trip"] = mix["samples_per_trip"]
    out["resources"] = isa_mix.resources(code_object).get("mci_vegas_batch")
    return out


def recorded_traffic(code_object):
    """HBM bytes per launch from the PMC passes of profiles/collect.sh -- only if they profiled THIS code object"""
    import glob
    for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            pj = json.load(open(tf))
        except Exception:
            continue
        if pj.get("code_object") == os.path.basename(code_object):
            return pj.get("hbm_bytes_per_launch"), os.path.basename(tf), pj.get("sq_avg_per_launch")
    return None, None, None


# ---------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--passes", type=int, default=0, help="timed passes of K steps each; the median pass is `value` (0: at least 3, and as many as fill --min-seconds)")
    ap.add_argument("--min-seconds", type=float, default=5.0, help="with --passes 0: keep timing passes until this much wall time is covered")
    ap.add_argument("--neval-per-gpu", type=float, default=1e8)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): every GPU runs --neval-per-gpu samples and 16 blocks per iteration; strong: ONE problem of "
                         "--neval-per-gpu samples and 16 blocks per iteration (the reference's default block count, main.jl:74) is split "
                         "over the GPUs like main.jl:121-122 does (block rounded to a multiple of the rank count)")
    ap.add_argument("--seed", type=int, default=20240229)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rng-bits", type=int, default=52, choices=(52, 32),
                    help="52 (default): the faithful stream, 52 random mantissa bits per draw like rand(Float64); 32: the opt-in cheaper stream (mci_set_rng_bits)")
    ap.add_argument("--rng-rounds", type=int, default=10, choices=(10, 7),
                    help="10 (default): Philox4x32-10; 7: the opt-in cheaper generator (mci_set_rng_rounds), pinned on the Random123 vectors")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))

    # stdout carries the JSON line and nothing else: whatever the libraries print there on the way (RCCL's start-up banner, gloo's
    # connection notes) goes to stderr -- file descriptor 1 points at stderr until the line is written to the saved descriptor
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and rank == 0:
        print("[bench] --gpus %d but the launcher started %d ranks: running %d" % (a.gpus, world, world), file=sys.stderr)

    import numpy as np
    import mcintegration_jl_amd as mci
    # the JIT's compiler: the ROCm installation's hiprtc + comgr, pinned BEFORE PyTorch is imported (PyTorch bundles its own copies, another
    # compiler build: the first comgr loaded into a process serves everybody; mci.use_rocm_compiler).  The line says which one it was.
    mci.use_rocm_compiler()          # (comgr + hiprtc only: the HIP runtime of this process stays ONE -- PyTorch's, loaded next -- as before)
    import torch
    compiler = mci.compiler_id()     # (loads libmci_hip.so: after torch, so that it binds to the runtime torch brought)
    from mcintegration_jl_amd.comm import LocalComm, RcclComm, TorchDistComm

    # test seam: an engine factory "module:attr" replaces the HIP engine so that the launcher, the block partition and the
    # reduction can run without a GPU (gloo).  Such a line is marked dry_run and claims nothing.
    factory = None
    if os.environ.get("MCI_BENCH_ENGINE"):
        mod, attr = os.environ["MCI_BENCH_ENGINE"].split(":")
        factory = getattr(importlib.import_module(mod), attr)
    dry = factory is not None
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    if not dry:
        # one GPU per rank; a launcher that isolates the ranks (HIP_VISIBLE_DEVICES per rank) shows each of them a single device 0
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank

    # MCI_BENCH_FORCE_COMM=1: run the N > 1 code path (process group, RCCL bootstrap, per-iteration all-reduce inside the
    # timed loop) with a single rank -- the only way to exercise it on a 1-GPU box
    force_comm = world == 1 and os.environ.get("MCI_BENCH_FORCE_COMM", "0") != "0"
    multi = world > 1 or force_comm
    comm, comm_kind, dist = LocalComm(), "none", None
    if multi:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_comm:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if dry:
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=300))
            comm, comm_kind = TorchDistComm(), "gloo"
        elif os.environ.get("MCI_COMM", "rccl") == "gloo":
            # real engines, host-side reduction: lets N ranks share ONE GPU (RCCL refuses two ranks on a device), which is how
            # the whole N > 1 path of this script -- launcher, partition, timing, JSON -- runs on a 1-GPU box (tests/)
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=300))
            comm, comm_kind = TorchDistComm(), "gloo"
        else:
            # The job's control plane (rendezvous, barriers, the max-over-ranks of the pass time, the gathered rank records) is a gloo
            # group on the host: the ONLY RCCL communicator on a device is then the library's own, created from a 128-byte id that
            # rank 0 ships through that group -- no second (torch) RCCL instance shares the GPUs with the one on the data path.
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=300))
            comm_kind = os.environ.get("MCI_COMM", "rccl")
            if comm_kind == "rccl":
                # ncclAllReduce inside the library, on its stream.  Every rank must end up with the same reducer: agree on the
                # outcome before using it.
                err = None
                try:
                    comm = RcclComm.from_torch_distributed(local_rank)
                except Exception as e:  # pragma: no cover - needs a failing RCCL
                    err = e
                ok = torch.tensor([0 if err else 1], dtype=torch.int32)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) != 1:  # pragma: no cover
                    if rank == 0:
                        print("[bench] library RCCL bootstrap failed on some rank (%s); every rank uses torch.distributed's "
                              "RCCL all_reduce on the library's device buffer instead" % err, file=sys.stderr)
                    comm_kind = "torch"
            if comm_kind == "torch":
                # torch's RCCL group as the reducer (zero copy on the library's device buffer, on the library's stream)
                nccl_group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=300), device_id=torch.device("cuda", local_rank))
                comm = TorchDistComm(group=nccl_group, tensor_device=dev)

    def barrier():
        if multi:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    n_gpus = world
    if a.scaling == "strong":   # one fixed problem, split over the ranks (main.jl:121-122)
        neval = int(a.neval_per_gpu)
        _, block = mci.standardize_block(neval, 16, n_gpus)
        neval_gpu = neval // n_gpus
    else:
        neval_gpu = int(a.neval_per_gpu)
        block = 16 * n_gpus
        neval = neval_gpu * n_gpus
    cfg = mci.Configuration(var=mci.Continuous(-L, L), dof=[[D]], seed=a.seed)
    f = mci.catalog.gaussian(D)

    # ---- warm-up: JIT/cache load + W training iterations from the uniform grid (untimed) ----
    res_w = mci.integrate(f, config=cfg, solver="vegas", neval=neval, niter=max(a.warmup, 1), block=block, comm=comm,
                          device=local_rank, adapt=True, engine_factory=factory, rng_bits=a.rng_bits, rng_rounds=a.rng_rounds)
    eng = cfg._engine
    per = block // n_gpus
    lo, hi = per * rank, per * (rank + 1)
    nevalperblock = neval // block
    grid_after_warmup = eng.grid(0)   # handed to the CPU baseline: it continues from the same trained map

    # ---- timed region: `passes` x EXACTLY K iterations, no host synchronisation inside a pass ----
    # (by default the passes go on for ~5 s: a 40 ms burst is invisible to anything that samples the GPU from outside, and the
    # median of a hundred passes is a better number than the median of three)
    pass_dt, dry_stats = [], []
    coll0 = eng.comm_collectives()[0] if hasattr(eng, "comm_collectives") else None   # ncclAllReduce calls the library has issued so far
    if hasattr(eng, "reserve_iterations"):
        eng.reserve_iterations(a.steps * (a.passes if a.passes > 0 else 400))
    npass = a.passes if a.passes > 0 else 3
    ipass = 0
    while ipass < npass:
        ipass += 1
        it0 = cfg.iterations_done
        barrier()
        t0 = time.perf_counter()
        for it in range(a.steps):
            eng.run("vegas", nevalperblock, lo, hi, it0 + it, cfg.seed)
            comm.all_reduce(eng)
            r = eng.finish("vegas", block, adapt=True, want_stats=dry)
            if dry:
                dry_stats.append(r)
        barrier()
        dt = time.perf_counter() - t0
        if multi:
            t = torch.tensor([dt], dtype=torch.float64)   # (the control-plane group is gloo: host tensors)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        pass_dt.append(dt)
        cfg.iterations_done += a.steps
        if a.passes <= 0 and not dry and ipass == npass and sum(pass_dt) < a.min_seconds and npass < 400:
            npass += 1   # (dt is already the maximum over the ranks, so every rank takes the same decision)
    if not dry:
        eng.check_status()   # a launch that tripped a device-side error (normalization, histogram) must not print a number
    ntimed = a.steps * len(pass_dt)
    dt = sorted(pass_dt)[len(pass_dt) // 2]   # median pass
    coll = None
    if coll0 is not None and isinstance(comm, RcclComm):
        c1, last = eng.comm_collectives()
        coll = {"per_step": (c1 - coll0) / float(ntimed), "doubles_each": last}   # north star: ONE all-reduce per iteration

    # production estimate: every timed iteration, weighted average with ignore = 0
    # (k_train copied every iteration's statistics head into the device-side log; one D2H after the loop)
    nobs = eng.nobs
    if dry:
        means, stds = [float(m[0]) for m, _ in dry_stats], [float(e[0]) for _, e in dry_stats]
        neval_reduced = None
    else:
        log = eng.iteration_log(min(ntimed, 4096))   # the estimate pools the last <= 4096 timed iterations
        means, stds = [], []
        for row in log:
            m, e = mci.mean_std(row[:nobs], row[nobs:2 * nobs], block)
            means.append(m[0])
            stds.append(e[0])
        neval_reduced = float(log[-1][2 * nobs + 1])   # config.neval after the all-reduce: the samples of ALL ranks
    mean, err, chi2 = mci.average(means, stds, init=1, max=len(means))

    # what the communicator itself says about the job: the library's view (mci_comm_rank), every rank's block range
    lib_rank, lib_n = comm.library_ranks() if hasattr(comm, "library_ranks") else (rank, comm.size)
    # this rank's time inside the one exchange step, HIP events around the library's ncclAllReduce (wait for the slowest rank's sample
    # pass + the latency of an all-reduce of the packed buffer); None for reducers that do not run inside the library
    ar_ms = None
    if not dry and hasattr(eng, "comm_times_ms") and isinstance(comm, RcclComm):
        ct = eng.comm_times_ms(64)
        ar_ms = round(float(np.mean(ct)), 5) if len(ct) else None
    mine = {"rank": rank, "comm_rank": lib_rank, "comm_ranks": lib_n, "blocks": [lo, hi], "device": None if dry else local_rank,
            "allreduce_ms_avg": ar_ms}
    ranks = [mine]
    if multi:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)

    if rank == 0:
        value = a.steps * neval / dt / 1e6
        out = {
            "metric": "Msamples/sec (whole node), 16-D Gaussian :vegas",
            "value": round(value, 2),
            "unit": "Msamples/s",
            "n_gpus": n_gpus,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 16-D unit Gaussian on [-sqrt(50),sqrt(50)]^16, shared-pool "
                                   "Continuous (1 grid, 999 bins), :vegas, %s" % (
                                       "neval=%.0e per GPU per iteration, block=16 per GPU" % neval_gpu if a.scaling == "weak" else
                                       "ONE problem of neval=%.0e per iteration in %d blocks split over %d GPU(s)" % (neval, block, n_gpus)),
                       "neval_per_iteration": neval, "block": block, "rng_bits": a.rng_bits, "rng_rounds": a.rng_rounds},
            "timing": {"passes": len(pass_dt), "ms_per_step_per_pass": [round(x / a.steps * 1e3, 4) for x in pass_dt[:8]] + (["..."] if len(pass_dt) > 8 else []),
                       "ms_per_step_max": round(max(pass_dt) / a.steps * 1e3, 4), "timed_seconds": round(sum(pass_dt), 3),
                       "ms_per_step_min": round(min(pass_dt) / a.steps * 1e3, 4), "value_is": "median pass"},
            "comm": {"kind": comm_kind, "ranks": max(r["comm_ranks"] for r in ranks), "world_size": world,
                     "control_plane": "gloo" if multi else None, "payload_doubles": getattr(eng, "packed_size", None), "collectives": coll,
                     "per_rank": ranks, "neval_after_allreduce": neval_reduced},
            "estimate": {"mean": mean, "sigma": err, "chi2_dof": chi2, "exact": EXACT, "iterations": len(means),
                         "deviation_sigma": (mean - EXACT) / err if err > 0 else None,
                         "last_training_iteration": [float(res_w.iter_mean[-1, 0]), float(res_w.iter_std[-1, 0])]},
        }
        if dry:
            out["dry_run"] = True
            out["value"] = 0.0
            out["config"]["engine"] = os.environ["MCI_BENCH_ENGINE"]
        else:
            # per-launch HIP-event durations of the sampling kernel over the timed region (library stream)
            kms, wgs, threads = eng.kernel_times_ms(min(ntimed, 512))
            k_avg_ms = float(np.mean(kms))
            code_object = eng.code_object("vegas")
            out["config"].update({"kernel": "mci_vegas_batch", "workgroups": wgs, "threads": threads, "table_mode": eng.table_mode,
                                  "code_object": os.path.basename(code_object), "compiler": compiler,
                                  "histogram_copies": eng.histogram_copies() if hasattr(eng, "histogram_copies") else 1})
            spl = nevalperblock * per                                   # samples of one launch on one GPU
            achieved_hbm = B_ALG * spl / (k_avg_ms * 1e-3) / 1e9       # GB/s
            traffic, traffic_src, sq = recorded_traffic(code_object)
            hbm = {"bound": "hbm", "achieved": round(achieved_hbm, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(achieved_hbm / HBM_PEAK_GBS, 4), "bytes_per_sample": B_ALG, "traffic": traffic, "traffic_source": traffic_src,
                   "note": "SURVEY 8(d) model: 32*D algorithmic bytes per sample (grid edges + histogram RMW) / kernel time. NOT the binding "
                           "bound: the tables are LDS-resident by design, real HBM traffic (`traffic`, PMC) is ~1e-3 of it, so frac > 1"}
            costs, costs_src = measured_issue_costs()
            roof = {"bound": "valu+lds", "kernel_ms_avg": round(k_avg_ms, 4), "traffic": traffic, "hbm_model": hbm}
            # the shader clock of the timed region: every timed launch's first wave brackets its sample loop with s_memtime (shader
            # cycles) and s_memrealtime (constant rate); the chip clocks to its power budget under this loop, not to its 2.4 GHz peak
            clk = eng.kernel_clocks_mhz(min(ntimed, 512)) if hasattr(eng, "kernel_clocks_mhz") else []
            sclk = float(np.mean(clk)) if len(clk) else None
            roof["clock"] = {"sclk_mhz_avg": None if sclk is None else round(sclk, 1), "sclk_mhz_min": None if sclk is None else round(float(np.min(clk)), 1),
                             "sclk_mhz_max": None if sclk is None else round(float(np.max(clk)), 1), "launches": int(len(clk)),
                             "source": "s_memtime / s_memrealtime around the sample loop of workgroup 0's first wave, every timed launch (mci_kernel_clocks)"}
            if costs:
                ir = issue_roofline(code_object, costs, spl, k_avg_ms, hist_copies=out["config"].get("histogram_copies", 1), sclk_mhz=sclk)
                top = "valu" if ir["valu"]["frac"] >= ir["lds"]["frac"] else "lds"
                ds = ir["valu_rounded_measured"]
                # `frac`: the VALU pipe at whole issue cycles per class (measured, rounded) and the measured clock (recomputable from the line itself); the
                # self-calibrated reading (issue costs this run measured with its own microbenchmark) rides along as frac_self_calibrated
                if ds.get("frac") is not None:
                    peak_ds = N_SIMD / (ds["issue_ns_per_wave_sample"] / ir["valu"]["instructions_per_wave_sample"])
                    roof.update({"achieved": ir["valu"]["achieved"], "peak": round(peak_ds, 2), "unit": "G wave-instructions/s", "frac": ds["frac"],
                                 "binding_pipe": "valu", "frac_self_calibrated": ir[top]["frac"], "binding_pipe_self_calibrated": top})
                else:
                    roof.update({"achieved": ir[top]["achieved"], "peak": ir[top]["peak"], "unit": "G wave-instructions/s", "frac": ir[top]["frac"],
                                 "binding_pipe": top, "frac_self_calibrated": ir[top]["frac"]})
                roof.update({"valu_rounded_measured": ds, "valu_flat_2cycle": ir["valu_flat_2cycle"], "frac_flat_2cycle": ir["valu_flat_2cycle"].get("frac"),
                             "valu": ir["valu"], "lds": ir["lds"], "mix": ir["mix"], "resources": ir["resources"],
                             "costs_source": costs_src,
                             "note": "frac = VALU issue time of the sample loop at whole cycles per instruction class (valu_rounded_measured.classes: n x "
                                     "cycles per wave and sample; the guide's 2 / 4 cycles, and 4 for the 32-bit forms with three sources or an SGPR "
                                     "source as this repository's microbenchmark measured them -- rounded, not a published table) x wave-samples per "
                                     "SIMD / the shader clock the timed launches measured (clock.sclk_mhz_avg) / HIP-event kernel time; peak = 1024 "
                                     "SIMDs / (those cycles per instruction of this mix / that clock).  frac_flat_2cycle = the same with EVERY 32-bit "
                                     "form at the guide's flat 2 cycles (valu_flat_2cycle): the pessimistic reading, the truth lies between the two.  "
                                     "frac_self_calibrated = the same mix priced with the issue costs tools/issue_microbench.hip measured on this "
                                     "GPU in this run (valu / lds: separate pipes that overlap, the larger fraction binds).  The SURVEY 8(d) HBM "
                                     "model is kept in hbm_model"})
                if sq and sq.get("SQ_INSTS_VALU") and sq.get("SQ_WAVES"):
                    roof["pmc_check"] = {"SQ_INSTS_VALU_per_wave_sample": sq["SQ_INSTS_VALU"] / (spl / 64.0),
                                         "SQ_INSTS_LDS_per_wave_sample": sq.get("SQ_INSTS_LDS", 0) / (spl / 64.0), "source": traffic_src}
            else:
                roof.update({"achieved": None, "peak": None, "unit": "G wave-instructions/s", "frac": None,
                             "note": "issue-cost table unavailable (no microbenchmark binary, no profiles/r02_issue_costs.json)"})
            out["roofline"] = roof
            if not a.no_cpu_baseline and n_gpus == 1:
                cb = cpu_baseline(trained_grid=grid_after_warmup)
                out["cpu_baseline"] = cb
                cm, cs = cb["estimate"]
                # north star: "the estimate within 1 sigma of the CPU reference".  One difference of two unbiased estimates is
                # N(0,1) in units of its sigma whatever the sample sizes, so the GPU estimate is TESTED against the set of
                # independent CPU runs: chi2 = sum z_s^2 over the seeds, p-value from chi2(nseeds)
                zs = [(mean - m) / math.hypot(err, s) for m, s in cb["estimate_per_seed"]]
                c2 = sum(z * z for z in zs)
                try:
                    from scipy.stats import chi2 as _chi2
                    pval = float(_chi2.sf(c2, len(zs)))
                except Exception:  # pragma: no cover
                    pval = None
                out["estimate"]["vs_cpu_sigma"] = (mean - cm) / math.hypot(err, cs)
                out["estimate"]["vs_cpu"] = {"z_per_seed": [round(z, 3) for z in zs], "chi2": round(c2, 3), "dof": len(zs), "p_value": pval,
                                             "consistent": None if pval is None else bool(pval > 0.01)}
        import ctypes
        ctypes.CDLL(None).fflush(None)   # (the C stdio buffer of this process: RCCL's banner leaves through the redirected descriptor)
        print(json.dumps(out), file=json_out, flush=True)
    # orderly release while the HIP runtime is still up: problem, then stream + RCCL communicator, then torch's group
    if hasattr(eng, "close"):
        eng.close()
    cfg._engine = None
    mci.shutdown()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
        # multi-rank runs leave without the interpreter's teardown: the destruction order of torch, RCCL and the HIP
        # runtime at exit is not ours to control (seen once: glibc "double free" after the result was printed)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
