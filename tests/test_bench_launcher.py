"""bench.py's own N-rank launch path, in the build container: `python bench.py --gpus 2` (no torch.distributed launcher
around it) must start two ranks by itself, give each its half of the global blocks (src/main.jl:121-122, :152-166), sum the
packed buffers every iteration (src/main.jl:177-188) and print ONE JSON line that says so.  No GPU here: the HIP engine is
replaced by the oracle-backed test engine through bench.py's MCI_BENCH_ENGINE seam and the reducer is gloo, so the line is
a dry run (no throughput claim) -- what is under test is the launcher, the partition and the communicator bookkeeping."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _run(args, extra_env=None, timeout=300):
    env = dict(os.environ)
    env["MCI_BENCH_ENGINE"] = "oracle_engine:OracleEngine"
    env["PYTHONPATH"] = os.pathsep.join([HERE, os.path.join(ROOT, "oracle"), ROOT, env.get("PYTHONPATH", "")])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.stdout.strip().splitlines() == lines, p.stdout[-2000:]   # stdout is the JSON line and nothing else (library chatter goes to stderr)
    assert len(lines) == 1, p.stdout[-2000:]   # rank 0 prints ONE line
    return json.loads(lines[-1])


def test_gpus_2_launches_two_ranks_by_itself(oracle):
    out = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--passes", "2", "--neval-per-gpu", "32000", "--no-cpu-baseline"])
    assert out["dry_run"] is True and out["value"] == 0.0
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["block"] == 32 and out["config"]["neval_per_iteration"] == 64000
    comm = out["comm"]
    assert comm["kind"] == "gloo" and comm["ranks"] == 2 and comm["world_size"] == 2
    # two distinct block ranges that tile the global blocks
    assert [r["blocks"] for r in comm["per_rank"]] == [[0, 16], [16, 32]]
    assert [r["rank"] for r in comm["per_rank"]] == [0, 1]
    assert out["timing"]["passes"] == 2 and len(out["timing"]["ms_per_step_per_pass"]) == 2
    est = out["estimate"]
    assert est["iterations"] == 4
    assert est["sigma"] > 0 and 0 < est["mean"] < 10   # (6.4e4 samples in 16-D say nothing about the value: plumbing only)


def test_one_rank_line_matches_the_two_rank_estimate_on_the_same_global_blocks(oracle):
    """the union of streams does not depend on the rank count: N=1 with block=32 draws the same samples as N=2 with 16 each.
    bench.py fixes block = 16 per rank, so compare 2 ranks x 16000 with 1 rank x 32000 only through the launcher's fields."""
    one = _run(["--gpus", "1", "--steps", "1", "--warmup", "1", "--passes", "1", "--neval-per-gpu", "32000", "--no-cpu-baseline"])
    assert one["n_gpus"] == 1 and one["comm"]["kind"] == "none" and one["comm"]["ranks"] == 1
    assert one["comm"]["per_rank"][0]["blocks"] == [0, 16]


def test_launched_under_torch_distributed_run_it_does_not_spawn_again(oracle):
    """the driver's own launch line: WORLD_SIZE is set, so bench.py must run as ONE rank of the job"""
    env = dict(os.environ)
    env["MCI_BENCH_ENGINE"] = "oracle_engine:OracleEngine"
    env["PYTHONPATH"] = os.pathsep.join([HERE, os.path.join(ROOT, "oracle"), ROOT, env.get("PYTHONPATH", "")])
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--passes", "1",
           "--neval-per-gpu", "16000", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.stdout.strip().splitlines() == lines, p.stdout[-2000:]   # stdout is the JSON line and nothing else (library chatter goes to stderr)
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["comm"]["ranks"] == 2 and [r["blocks"] for r in out["comm"]["per_rank"]] == [[0, 16], [16, 32]]


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_through_the_real_engine():
    """the whole N = 2 path of bench.py with the HIP engine -- own launcher, block partition, per-iteration reduction inside the timed
    loop, max-over-ranks timing, roofline, JSON -- on a 1-GPU box: the two ranks share device 0 and reduce through gloo (RCCL itself
    refuses two ranks on one device; its single-rank path is test_library_rccl_single_rank_and_torch_reducer)"""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MCI_BENCH_ENGINE"):
        env.pop(k, None)
    env["MCI_COMM"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "3", "--passes", "2",
                        "--neval-per-gpu", "2e7", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.stdout.strip().splitlines() == lines, p.stdout[-2000:]   # stdout is the JSON line and nothing else (library chatter goes to stderr)
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert "dry_run" not in out and out["n_gpus"] == 2 and out["value"] > 1000.0
    assert out["comm"]["kind"] == "gloo" and [r["blocks"] for r in out["comm"]["per_rank"]] == [[0, 16], [16, 32]]
    assert out["comm"]["neval_after_allreduce"] == 4e7                  # config.neval after the reduction: both ranks' samples
    assert out["config"]["neval_per_iteration"] == 40000000 and out["roofline"]["bound"] == "valu+lds"
    est = out["estimate"]
    assert abs(est["mean"] - est["exact"]) < 6 * est["sigma"] and est["iterations"] == 6


@pytest.mark.gpu
@pytest.mark.parametrize("reducer", ["rccl", "torch"])
def test_forced_single_rank_communicator_goes_through_the_n_rank_code_path(reducer):
    """`MCI_BENCH_FORCE_COMM=1 bench.py --gpus 1`: everything an N > 1 job does except a second device -- gloo control plane
    (rendezvous, barriers, max-over-ranks of the pass time, gathered rank records), the library's own RCCL communicator created from
    an id shipped through that group (reducer "rccl": the only RCCL instance on the device) or torch's RCCL group on the library's
    device buffer and stream (reducer "torch": the fallback), ONE all-reduce of the packed buffer per iteration inside the timed
    loop (src/main.jl:177-188), and a JSON line that says what the communicator saw."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MCI_BENCH_ENGINE"):
        env.pop(k, None)
    env["MCI_BENCH_FORCE_COMM"] = "1"
    env["MCI_COMM"] = reducer
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "3", "--passes", "2",
                        "--neval-per-gpu", "2e7", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.stdout.strip().splitlines() == lines and len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    comm = out["comm"]
    assert comm["kind"] == reducer and comm["ranks"] == 1 and comm["world_size"] == 1 and comm["control_plane"] == "gloo"
    assert comm["neval_after_allreduce"] == 2e7 and comm["payload_doubles"] == 2 + 2 + 2 + 999 + 2 * 12
    assert comm["per_rank"][0]["blocks"] == [0, 16] and comm["per_rank"][0]["comm_ranks"] == 1
    if reducer == "rccl":   # HIP events around the library's ncclAllReduce (launches of >= 2^20 samples are timed)
        assert 0.0 < comm["per_rank"][0]["allreduce_ms_avg"] < 1.0, comm
    assert out["n_gpus"] == 1 and out["scaling"] == "weak" and out["value"] > 1000.0
    est = out["estimate"]
    assert abs(est["mean"] - est["exact"]) < 6 * est["sigma"] and est["iterations"] == 8


@pytest.mark.gpu
def test_strong_scaling_option_splits_one_problem():
    """--scaling strong: ONE problem (neval, 16 blocks) per iteration whatever the rank count, blocks rounded like main.jl:121-122;
    with one rank it is the same job as the weak line"""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MCI_BENCH_ENGINE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "3", "--passes", "1",
                        "--neval-per-gpu", "2e7", "--no-cpu-baseline", "--scaling", "strong"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-4000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "strong" and out["config"]["block"] == 16 and out["config"]["neval_per_iteration"] == 20000000
    assert out["comm"]["kind"] == "none" and out["value"] > 1000.0


def test_strong_scaling_block_partition_on_two_dry_ranks(oracle):
    """the strong option under the launcher with two (dry, gloo) ranks: 16 blocks in all, 8 per rank, neval split in two"""
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "1", "--passes", "1", "--neval-per-gpu", "32000", "--no-cpu-baseline", "--scaling", "strong"])
    assert out["scaling"] == "strong" and out["config"]["block"] == 16 and out["config"]["neval_per_iteration"] == 32000
    assert [r["blocks"] for r in out["comm"]["per_rank"]] == [[0, 8], [8, 16]]
