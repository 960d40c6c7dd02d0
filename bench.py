#!/usr/bin/env python3
"""bench.py -- headline benchmark: Msamples/s of the VEGAS sample-batch path on the 16-D Gaussian
(BASELINE.json configs[1]) on N MI355X, plus the MC estimate and its sigma.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one VEGAS iteration: neval samples per GPU drawn through the adaptive map, integrand
evaluated, observables + per-bin histogram accumulated, block statistics merged, one RCCL all-reduce of
the packed buffer (N > 1), grid refinement.  Warm-up iterations train the grid from uniform
(reference resume pattern, docs/src/index.md:129, test/bubble.jl:108-113); the K timed iterations
continue from the trained grid with adapt=true, ignore=0.  Weak scaling: per-GPU work is fixed
(neval_total = N * neval_per_gpu, block = N * 16).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D = 16
L = math.sqrt(50.0)
EXACT = math.erf(L / math.sqrt(2.0)) ** D
B_ALG = 32 * D          # algorithmic bytes per sample: per dim 16 B of grid edges + 16 B histogram RMW
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


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


def cpu_baseline(trained_grid=None, seconds_target=12.0):
    """The CPU oracle (oracle/, kind "port": the reference itself is Julia and cannot run here), threaded
    over blocks like parallel=:thread (src/main.jl:153-158) on the host cores this process may use, same 16-D
    Gaussian.  The timed run continues from the grid the GPU trained (the reference's resume pattern,
    docs/src/index.md:129), so that its estimate is sharp enough to compare the GPU's with."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mci_oracle as O
    cores = usable_cores()
    block = max(16, cores)
    cfg = O.Config([dict(kind=0, pool=0, lower=-L, upper=L)], [[D]])
    probe = 20000 * block
    t0 = time.time()
    cfg.integrate(O.VEGAS, "gaussian", [float(D)], neval=probe, niter=1, block=block, seed=1, nthreads=cores)
    rate = probe / max(time.time() - t0, 1e-6)
    neval = int(max(probe, min(rate * seconds_target / 3, 5e8)))
    if trained_grid is not None:
        cfg.set_grid(0, trained_grid)
    t0 = time.time()
    r = cfg.integrate(O.VEGAS, "gaussian", [float(D)], neval=neval, niter=3, block=block, seed=2, nthreads=cores,
                      ignore=0 if trained_grid is not None else 1)
    dt = time.time() - t0
    est = [float(r["mean"][0]), float(r["stdev"][0])]
    return {"value": round(3 * (neval // block) * block / dt / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "estimate": est,
            "sample": "oracle/mci_oracle.c (C restatement, OpenMP over blocks), 16-D Gaussian :vegas, %d samples x 3 iterations "
                      "%s, block=%d, %.1f s; estimate (weighted average) %.6f +- %.6f"
                      % (neval, "continuing from the GPU-trained grid" if trained_grid is not None else "from the uniform grid", block, dt, est[0], est[1])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--neval-per-gpu", type=float, default=1e8)
    ap.add_argument("--seed", type=int, default=20240229)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import numpy as np
    import torch
    import mcintegration_jl_amd as mci
    from mcintegration_jl_amd.comm import LocalComm, RcclComm, TorchDistComm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    comm_kind = "none"
    comm = LocalComm()
    # MCI_BENCH_FORCE_COMM=1: run the N > 1 code path (process group, RCCL bootstrap, per-iteration all-reduce inside the
    # timed loop) with a single rank -- the only way to exercise it on a 1-GPU box
    force_comm = world == 1 and os.environ.get("MCI_BENCH_FORCE_COMM", "0") != "0"
    if world > 1 or force_comm:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_comm:
            os.environ.setdefault("MASTER_PORT", "29577")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        comm_kind = os.environ.get("MCI_COMM", "rccl")
        if comm_kind == "rccl":
            # ncclAllReduce inside the library, on its stream.  The bootstrap runs under a watchdog: a rank that cannot
            # join within 120 s must not hang the job, and all ranks have to agree on the reducer they use.
            import threading
            box = {}

            def boot():
                try:
                    torch.cuda.set_device(local_rank)   # the current device is per thread; the id broadcast runs on it
                    box["comm"] = RcclComm.from_torch_distributed(local_rank)
                except Exception as e:  # pragma: no cover - exercised only on multi-GPU nodes
                    box["err"] = e
            th = threading.Thread(target=boot, daemon=True)
            if os.environ.get("MCI_BOOT_INLINE", "0") != "0":
                boot()
            else:
                th.start()
                th.join(120.0)
            ok = torch.tensor([1 if "comm" in box else 0], dtype=torch.int32, device="cuda:%d" % local_rank)
            if not th.is_alive():   # (a stuck bootstrap thread may sit inside a torch collective: do not issue another one)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                comm = box["comm"]
            else:  # pragma: no cover
                if rank == 0:
                    print("[bench] library RCCL bootstrap failed (%s); using torch.distributed all_reduce on the device buffer"
                          % box.get("err", "timeout"), file=sys.stderr)
                comm_kind = "torch"
        if comm_kind == "torch":
            comm = TorchDistComm(tensor_device="cuda:%d" % local_rank)  # zero copy, on the library's stream

    def barrier():
        if world > 1 or force_comm:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    n_gpus = world
    neval_gpu = int(a.neval_per_gpu)
    block = 16 * n_gpus
    neval = neval_gpu * n_gpus
    cfg = mci.Configuration(var=mci.Continuous(-L, L), dof=[[D]], seed=a.seed)
    f = mci.catalog.gaussian(D)

    # ---- warm-up: JIT/cache load + W training iterations from the uniform grid (untimed) ----
    res_w = mci.integrate(f, config=cfg, solver="vegas", neval=neval, niter=max(a.warmup, 1), block=block, comm=comm,
                          device=local_rank, adapt=True)
    eng = cfg._engine
    per = block // n_gpus
    lo, hi = per * rank, per * (rank + 1)
    nevalperblock = neval // block
    it0 = cfg.iterations_done
    grid_after_warmup = eng.grid(0)   # handed to the CPU baseline: it continues from the same trained map

    # ---- timed region: EXACTLY K iterations, no host synchronisation inside ----
    barrier()
    t0 = time.perf_counter()
    for it in range(a.steps):
        eng.run("vegas", nevalperblock, lo, hi, it0 + it, cfg.seed)
        comm.all_reduce(eng)
        eng.finish("vegas", block, adapt=True, want_stats=False)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1 or force_comm:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device="cuda:%d" % local_rank)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    cfg.iterations_done += a.steps

    # per-launch HIP-event durations of the sampling kernel over the timed region (library stream)
    kms, wgs, threads = eng.kernel_times_ms(a.steps)
    k_avg_ms = float(np.mean(kms))
    # production estimate: the K timed iterations, weighted average with ignore = 0
    # (k_train copied every iteration's statistics head into the device-side log; one D2H after the loop)
    log = eng.iteration_log(a.steps)
    means, stds = [], []
    for row in log:
        m, e = mci.mean_std(row[:eng.nobs], row[eng.nobs:2 * eng.nobs], block)
        means.append(m[0])
        stds.append(e[0])
    mean, err, chi2 = mci.average(means, stds, init=1, max=len(means))

    if rank == 0:
        value = a.steps * neval / dt / 1e6
        achieved = B_ALG * (nevalperblock * per) / (k_avg_ms * 1e-3) / 1e9  # GB/s, one launch on one GPU
        # PMC-derived per-launch figures of this same workload (profiles/collect.sh; separate --pmc passes)
        traffic, valu_insts = None, None
        import glob
        for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1:]:
            try:
                pj = json.load(open(tf))
                traffic = pj.get("hbm_bytes_per_launch")
                valu_insts = pj.get("sq_avg_per_launch", {}).get("SQ_INSTS_VALU")
            except Exception:
                pass
        out = {
            "metric": "Msamples/sec (whole node), 16-D Gaussian :vegas",
            "value": round(value, 2),
            "unit": "Msamples/s",
            "n_gpus": n_gpus,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 16-D unit Gaussian on [-sqrt(50),sqrt(50)]^16, shared-pool "
                                   "Continuous (1 grid, 999 bins), :vegas, neval=%.0e per GPU per iteration, block=16 per GPU" % neval_gpu,
                       "neval_per_iteration": neval, "block": block, "comm": comm_kind,
                       "kernel": "mci_vegas_batch", "workgroups": wgs, "threads": threads, "table_mode": eng.table_mode},
            "estimate": {"mean": mean, "sigma": err, "chi2_dof": chi2, "exact": EXACT,
                         "deviation_sigma": (mean - EXACT) / err if err > 0 else None,
                         "last_training_iteration": [res_w.iter_mean[-1, 0], res_w.iter_std[-1, 0]]},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel_ms_avg": round(k_avg_ms, 4), "bytes_per_sample": B_ALG,
                         "note": "achieved = algorithmic bytes (32*D B/sample: grid edges + histogram RMW) / HIP-event kernel time; "
                                 "the tables are LDS-resident by design, so real HBM traffic (traffic) is ~0 and frac may exceed 1: "
                                 "the kernel is fp64-VALU/Philox bound, see DESIGN.md"},
        }
        if valu_insts and neval_gpu == 10**8:
            # the bound the kernel actually runs against: one wave64 VALU instruction per 4 cycles on each of
            # 256 CUs x 4 SIMDs at 2.4 GHz (MI355X_MICROARCH.md); instruction count per launch from SQ_INSTS_VALU
            peak = 256 * 4 * 2.4e9 / 4 / 1e9
            ach = valu_insts / (k_avg_ms * 1e-3) / 1e9
            out["roofline_valu"] = {"bound": "valu-issue", "achieved": round(ach, 1), "peak": round(peak, 1),
                                    "unit": "G wave-instructions/s", "frac": round(ach / peak, 4),
                                    "insts_per_launch": valu_insts, "note": "SQ_INSTS_VALU (rocprofv3 --pmc, profiles/) / live HIP-event kernel time"}
        if not a.no_cpu_baseline and n_gpus == 1:
            out["cpu_baseline"] = cpu_baseline(trained_grid=grid_after_warmup)
            cm, cs = out["cpu_baseline"]["estimate"]
            # north star: "the estimate within 1 sigma of the CPU reference" -- the CPU run's estimate vs the GPU's
            out["estimate"]["vs_cpu_sigma"] = (mean - cm) / math.hypot(err, cs)
        import ctypes
        ctypes.CDLL(None).fflush(None)   # RCCL's start-up banner sits in the C stdio buffer: the JSON line stays the last line of stdout
        print(json.dumps(out), flush=True)
    # orderly release while the HIP runtime is still up: problem, then stream + RCCL communicator, then torch's group
    eng.close()
    cfg._engine = None
    mci.shutdown()
    if world > 1 or force_comm:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
        # multi-rank runs leave without the interpreter's teardown: the destruction order of torch, RCCL and the HIP
        # runtime at exit is not ours to control (seen once: glibc "double free" after the result was printed)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
