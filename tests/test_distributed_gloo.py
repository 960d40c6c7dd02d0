"""N>1 path on CPU: world_size-2 torch.distributed/gloo runs of the product's integrate() driver
(block partition main.jl:121-122,152-166; one summed packed buffer per iteration main.jl:177-188; identical
train on every rank) with the oracle-backed engine injected.  The 2-rank result must equal the 1-rank
result on the same global blocks up to the reassociation of one cross-rank sum."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, solver, q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mcintegration_jl_amd as mci
    from mcintegration_jl_amd.comm import TorchDistComm
    from oracle_engine import OracleEngine
    comm = TorchDistComm()
    res = mci.integrate(mci.catalog.sphere2(), var=mci.Continuous(0.0, 1.0), dof=[[2], [3]], solver=solver, neval=24000,
                        niter=4, block=8, seed=77, comm=comm, engine_factory=OracleEngine, nchain=4)
    eng = res.config._engine
    q.put((rank, res.iter_mean, res.iter_std, np.array(res.mean), np.array(res.stdev), eng.grid(0), eng.calls))
    dist.barrier()
    dist.destroy_process_group()


def _single(solver):
    import mcintegration_jl_amd as mci
    from oracle_engine import OracleEngine
    res = mci.integrate(mci.catalog.sphere2(), var=mci.Continuous(0.0, 1.0), dof=[[2], [3]], solver=solver, neval=24000,
                        niter=4, block=8, seed=77, engine_factory=OracleEngine, nchain=4)
    return res, res.config._engine


@pytest.mark.parametrize("solver", ["vegas", "vegasmc", "mcmc"])
def test_two_ranks_equal_one_rank(solver):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, solver, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=90) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref, eng = _single(solver)
    for rank, im, ie, m, s, grid, calls in outs:
        # every rank holds the same Result and the same trained grid (all-reduce, not reduce-to-root)
        if solver == "mcmc":
            # unit-weight histograms have empty bins, whose clearStatistics! offsets depend on the worker count (see
            # below): after the first train! the grids differ by ~1e-6 and a single flipped accept decision moves a chain
            # discretely -> iteration 1 (common grid) must agree exactly, the later ones statistically
            np.testing.assert_allclose(im[0], ref.iter_mean[0], rtol=1e-9)
            assert np.all(np.abs(im - ref.iter_mean) < 5 * np.hypot(ie, ref.iter_std) + 1e-12)
        else:
            np.testing.assert_allclose(im, ref.iter_mean, rtol=1e-9)
            np.testing.assert_allclose(ie, ref.iter_std, rtol=1e-6)
            np.testing.assert_allclose(m, ref.mean, rtol=1e-9)
        # every worker's summedConfig starts from clearStatistics! (1e-10 per bin, variable.jl:565) and the reduce sums
        # them (configuration.jl:271-279), so -- in the reference too -- EMPTY bins hold (nworker + nblock) * 1e-10:
        # only the unit-weight :mcmc histogram has empty bins at this size, and there the grid moves by ~1e-6
        np.testing.assert_allclose(grid, eng.grid(0), rtol=0, atol=1e-11 if solver != "mcmc" else 1e-3)
        # rank r ran global blocks [4r, 4r+4) in every iteration (main.jl:122: block % nprocs == 0)
        assert calls == [(4 * rank, 4 * rank + 4, it) for it in range(4)]
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    np.testing.assert_array_equal(outs[0][5], outs[1][5])


def test_block_count_is_rounded_to_a_multiple_of_the_worker_count():
    # _standardize_block with 3 workers: block 16 -> 15 (main.jl:225-227), nevalperblock = neval // 15
    from mcintegration_jl_amd.integrate import standardize_block
    assert standardize_block(30000, 16, 3) == (2000, 15)
    assert standardize_block(30000, 2, 8) == (3750, 8)


# ---------------------------------------------------------------------------------------------------------
# The same check with the REAL engine: two processes share GPU 0, each runs its half of the global blocks
# through the HIP kernels, the packed buffers are summed with gloo.  Philox indices are global, so the
# 2-rank run must reproduce the 1-rank run up to the reassociation of one cross-rank sum.
# ---------------------------------------------------------------------------------------------------------
def _gpu_worker(rank, world, port, solver, q):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mcintegration_jl_amd as mci
    from mcintegration_jl_amd.comm import TorchDistComm
    res = mci.integrate(mci.catalog.sphere2(), var=mci.Continuous(0.0, 1.0), dof=[[2], [3]], solver=solver, neval=64000,
                        niter=4, block=8, seed=77, comm=TorchDistComm(), nchain=4, device=0)
    q.put((rank, res.iter_mean, res.iter_std, res.config._engine.grid(0), np.asarray(res._flat_std), bool(res.correlated)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["vegas", "vegasmc", "mcmc"])
def test_two_ranks_on_one_gpu_equal_one_rank(solver):
    import mcintegration_jl_amd as mci
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, solver, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = mci.integrate(mci.catalog.sphere2(), var=mci.Continuous(0.0, 1.0), dof=[[2], [3]], solver=solver, neval=64000,
                        niter=4, block=8, seed=77, nchain=4, device=0)
    # the chain solvers ran many chains per block and carried them: both ranks report the block-lineage error, from the lineage sums of
    # ALL blocks (each rank's four, summed through the communicator: mci_lineage_sums + comm.sum_host) -- the one-rank run's
    assert [o[5] for o in outs] == [solver != "vegas"] * 2 and bool(ref.correlated) == (solver != "vegas")
    np.testing.assert_array_equal(outs[0][4], outs[1][4])
    if solver == "vegasmc":
        np.testing.assert_allclose(outs[0][4], ref._flat_std, rtol=1e-3)
    elif solver == "mcmc":
        assert np.all(outs[0][4] < 3 * ref._flat_std) and np.all(outs[0][4] > ref._flat_std / 3)
    for rank, im, ie, grid, _sd, _corr in outs:
        if solver == "mcmc":   # see test_two_ranks_equal_one_rank: exact on the common first grid, statistical afterwards
            np.testing.assert_allclose(im[0], ref.iter_mean[0], rtol=1e-9)
            assert np.all(np.abs(im - ref.iter_mean) < 5 * np.hypot(ie, ref.iter_std) + 1e-12)
            np.testing.assert_allclose(grid, ref.config._engine.grid(0), rtol=0, atol=1e-3)
            continue
        np.testing.assert_allclose(im, ref.iter_mean, rtol=1e-6)   # + the rounding amplification of 3 train! steps
        np.testing.assert_allclose(ie, ref.iter_std, rtol=1e-4)
        np.testing.assert_allclose(grid, ref.config._engine.grid(0), rtol=0, atol=1e-9)
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    np.testing.assert_array_equal(outs[0][3], outs[1][3])


def _gpu_worker_auto_mcmc(rank, world, port, q):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import math
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mcintegration_jl_amd as mci
    from mcintegration_jl_amd.comm import TorchDistComm
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]], seed=5)
    res = mci.integrate(mci.catalog.nested_gauss(), config=cfg, solver="mcmc", neval=4e6, niter=6, block=16, comm=TorchDistComm(), device=0)
    eng = cfg._engine
    q.put((rank, res.iter_mean, np.asarray(res._flat_mean), np.asarray(res._flat_std), eng.last_chain_launch()[0], int(res.warmup)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_size_their_mcmc_chains_from_the_same_summed_holding_times():
    """Automatic :mcmc chain lengths follow the holding times the launch before measured.  With several ranks every rank must derive the
    SAME length -- and take the same decision about running a warm-up iteration again -- so the 64 holding-time counts are summed over
    the ranks: they ride behind the tables in the ONE packed all-reduce of the iteration (mci_reduce_size doubles; an external reducer
    like TorchDistComm then calls mci_external_reduce_done).  Two processes on one GPU, gloo: the same chain count and the same number
    of warm-up repeats on both ranks, bit-identical iteration means, and the exact products of erf within 5 sigma."""
    import math
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker_auto_mcmc, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, im0, m0, s0, nchain0, wu0), (_, im1, m1, s1, nchain1, wu1) = outs
    assert nchain0 == nchain1 and nchain0 > 1 and wu0 == wu1, (nchain0, nchain1, wu0, wu1)
    np.testing.assert_array_equal(im0, im1)
    exact = np.array([math.erf(5.0) ** d for d in (3, 6, 9, 12)])
    assert np.all(np.abs(m0 - exact) < 5 * s0), (m0, s0)
