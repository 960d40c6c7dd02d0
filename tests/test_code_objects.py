"""The gfx950 code objects of the BASELINE configurations, inspected without a GPU: hiprtc cross-compiles them in the build
container (offline context), llvm-readelf gives their register / scratch budget, llvm-objdump the instruction mix of the sample
loop that bench.py's roofline is priced on (mcintegration.jl_amd/isa_mix.py)."""
import math
import os

import numpy as np
import pytest

import mcintegration_jl_amd as mci
from mcintegration_jl_amd import isa_mix

L = math.sqrt(50.0)
PI = math.pi


def _bubble():
    p = mci.catalog.bubble_parameters()
    var = (mci.Continuous(0.0, 1.0, alpha=3.0), mci.Continuous(0.0, PI, alpha=3.0), mci.Continuous(0.0, 2 * PI, alpha=3.0),
           mci.Continuous(0.0, p["beta"], alpha=3.0), mci.Discrete(1, 4, adapt=False))
    return mci.Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)])


# (name, Configuration, integrand, measure, solver, kernel, max VGPRs = the occupancy step the design relies on)
BASELINE = [
    ("c1", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1]]), mci.catalog.log_over_sqrt, None, "vegas", "mci_vegas_batch", 64),
    ("c2", lambda: mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]]), lambda: mci.catalog.gaussian(16), None, "vegas", "mci_vegas_batch", 128),
    # (the same Gaussian on 16 independent grids: histogram in the pass, 3 grids' edges cached, one 1024-thread workgroup per CU)
    ("c2_16grids", lambda: mci.Configuration(var=mci.Continuous([(-L, L)] * 16), dof=[[1]]), lambda: mci.catalog.gaussian(16), None, "vegas",
     "mci_vegas_batch", 168),
    ("c3", _bubble, mci.catalog.bubble, lambda: mci.bin_by(4), "vegasmc", "mci_vegasmc_chains", 256),
    # (one 1024-thread workgroup per CU owns its LDS: 4 waves/SIMD, so the budget is 128 registers -- the bins are packed as drawn and the
    # code object holds the measurefreq == 1 loop only)
    ("c4", lambda: mci.Configuration(var=mci.Continuous([(0.0, 1.0)] * 32), dof=[[1]]), lambda: mci.catalog.genz_product_peak(32), None, "vegas",
     "mci_vegas_batch", 128),
    ("c5", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]]), mci.catalog.nested_gauss, None, "mcmc",
     "mci_mcmc_chains", 512),
]


def _code_object(cfg, f, meas, solver):
    eng = mci.Engine(cfg(), f(), measure=meas() if meas else None, device=-1)   # offline context: compile only
    eng.compile(solver)
    path = eng.code_object(solver)
    eng.close()
    assert os.path.exists(path)
    return path


@pytest.mark.parametrize("name,cfg,f,meas,solver,kernel,max_vgpr", BASELINE, ids=[b[0] for b in BASELINE])
def test_baseline_kernels_do_not_spill(name, cfg, f, meas, solver, kernel, max_vgpr):
    """no VGPR spills, no scratch, and the register budget of the occupancy step each kernel is designed for
    (MI355X_MICROARCH.md: <= 128 VGPRs -> 4 waves/SIMD, <= 168 -> 3, <= 256 -> 2, <= 512 -> 1)"""
    res = isa_mix.resources(_code_object(cfg, f, meas, solver))
    assert kernel in res, res
    for k, r in res.items():
        assert r["vgpr_spill"] == 0 and r["scratch"] == 0, (name, k, r)
    assert res[kernel]["vgpr"] <= max_vgpr, (name, res[kernel])
    if name == "c2_16grids":
        assert res[kernel]["max_threads"] in (1024, 768)   # a rung of the 1024 / 768 / 512 ladder without scratch (the two measure the same: 2.46 ms per 1e8)
    if name == "c4":
        assert res["mci_vegas_tiles"]["vgpr"] <= 128  # the replay kernel shares the workgroup size (one workgroup per CU: its LDS tile)
        assert res[kernel]["max_threads"] == 1024     # plan A of the split-all pass: first rung of the ladder


@pytest.mark.parametrize("name,solver", [("c3", "vegasmc"), ("c5", "mcmc"), ("c5", "vegasmc"), ("c3", "mcmc")])
def test_kernels_with_several_lanes_per_chain_do_not_spill(name, solver):
    """csrc/mci_spec.h (a group of lanes steps one chain): launches of few chains, one wave per SIMD -- up to 512 registers (VGPRs + AGPRs),
    no scratch, and like every sample kernel no static LDS"""
    b = [x for x in BASELINE if x[0] == name][0]
    res = isa_mix.resources(_code_object(b[1], b[2], b[3], solver + "_lanes"))
    k = res["mci_%s_spec" % solver]
    # (no scratch memory; a handful of VGPRs parked in AGPRs -- vgpr_spill without scratch: v_accvgpr moves, the 12-D :vegasmc kernel shows 3
    # -- is what the unified 512-entry file of a one-wave-per-SIMD kernel is for)
    assert k["scratch"] == 0 and k["vgpr_spill"] <= 8 and k["vgpr"] <= 512 and k.get("lds", 0) == 0, k


@pytest.mark.parametrize("name", ["c1", "c2"])
def test_persistent_vegas_kernel_neither_spills_nor_declares_static_lds(name):
    """the persistent :vegas kernel (mci_set_persistent; sample loop + block merge + train! of one Continuous grid, 256 threads, plain
    layout): no scratch, at most 256 VGPRs (two waves per SIMD: a grid of <= 129 workgroups is co-resident many times over), and its LDS is
    all dynamic (the pair table is addressed from LDS address 0); layouts with several leaves have no such kernel"""
    b = [x for x in BASELINE if x[0] == name][0]
    res = isa_mix.resources(_code_object(b[1], b[2], None, "vegas_persistent"))
    k = res["mci_vegas_persist"]
    assert k["vgpr_spill"] == 0 and k["scratch"] == 0 and k["vgpr"] <= 256 and k["max_threads"] == 256, k
    assert k.get("lds", 0) == 0, k
    eng = mci.Engine(_bubble(), mci.catalog.bubble(), measure=mci.bin_by(4), device=-1)
    with pytest.raises(Exception, match="no persistent"):
        eng.compile("vegas_persistent")
    eng.close()


@pytest.mark.parametrize("alpha,ninc,trace", [(2.0, 1000, False), (1.0, 1000, False), (3.0, 1000, False), (2.5, 1000, False), (2.0, 1300, False),
                                              (2.5, 1300, False), (2.0, 1000, True)])
def test_every_preprocessor_variant_of_the_persistent_unit_compiles(alpha, ninc, trace, monkeypatch):
    """csrc/mci_train.h is compiled by hiprtc into the persistent :vegas kernel under switches the JIT sets from the problem
    (mci_jit.h): MCI_TRAIN_POWER = 1 | 2 | 3 (the exponent of the one grid as a product) | 4 (pow()), MCI_TRAIN_SHORT_SUMS (grids of at
    most 1024 increments: Julia's sum() is its @simd block alone) or the unrolled pairwise recursion, MCI_TRAIN_SCAN_ONLY +
    MCI_TRAIN_CONTINUOUS_ONLY (always, for this unit), and MCI_PERSIST_TRACE (tools/persist_trace.py, through MCI_JIT_FLAGS).  Every
    combination that can occur must compile for gfx950 without scratch or static LDS; the ahead-of-time kernels (k_train / k_finish:
    none of the switches) are compiled by build()."""
    if trace:
        monkeypatch.setenv("MCI_JIT_FLAGS", "-DMCI_PERSIST_TRACE=1")
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0, alpha=alpha, ninc=ninc), dof=[[2]])
    eng = mci.Engine(cfg, mci.catalog.x2y2(), device=-1)
    eng.compile("vegas_persistent")
    k = isa_mix.resources(eng.code_object("vegas_persistent"))["mci_vegas_persist"]
    eng.close()
    assert k["vgpr_spill"] == 0 and k["scratch"] == 0 and k.get("lds", 0) == 0 and k["max_threads"] == 256, (alpha, ninc, trace, k)


def test_split_all_pass_falls_back_to_512_threads_when_the_integrand_needs_every_draw_at_once():
    """32 independent grids with an integrand that cannot consume the draws as they come (it needs their mean first, then every draw
    and every intermediate again): at 768 threads (168 VGPRs) the sample pass would spill, so the library compiles plan B -- 512
    threads, 256 registers -- and still neither spills nor uses scratch"""
    body = """double m = 0.0; for (int i = 0; i < 32; ++i) m += x[i]; m *= 0.03125;
              double y[32], q = 0.0; for (int i = 0; i < 32; ++i) { y[i] = (x[i] - m) * (x[(i + 7) % 32] + m); q += y[i]; }
              double p = 1.0; for (int i = 0; i < 32; ++i) p *= 1.0 + (y[i] - q) * (x[31 - i] - y[(i + 13) % 32]); w[0] = p;"""
    res = isa_mix.resources(_code_object(lambda: mci.Configuration(var=mci.Continuous([(0.0, 1.0)] * 32), dof=[[1]]),
                                         lambda: mci.Integrand(body), None, "vegas"))
    k = res["mci_vegas_batch"]
    assert k["vgpr_spill"] == 0 and k["scratch"] == 0 and k["max_threads"] == 512 and 168 < k["vgpr"] <= 256, k


def test_c2_sample_loop_mix():
    """the 16-D Gaussian's sample loop: one ds_read_b128 + one ds_add_f64 per draw, Philox as v_mad_u64_u32 + v_bitop3_b32
    (no two-instruction xor3), straight-line body (no inner loops)"""
    mix = isa_mix.loop_mix(_code_object(*[b for b in BASELINE if b[0] == "c2"][0][1:5]), "mci_vegas_batch", draws_per_sample=16)
    m, c = mix["mnemonics"], mix["classes"]
    assert mix["inner_backward_branches"] == 0 and mix["samples_per_trip"] == 2     # the pipelined loop: two samples per trip
    assert m["v_lshlrev_b32_e32"] == 16 and m.get("v_mov_b32_e32", 0) <= 2          # table offsets as VOP2 shifts, no bin copies
    assert m["ds_read_b128"] == 16 and m["ds_add_f64"] == 16 and mix["pipes"]["lds"] == 32
    assert 120 <= m["v_mad_u64_u32"] <= 160 and 120 <= m["v_bitop3_b32"] <= 160     # 8 calls x 10 rounds x 2, minus shared first-round terms
    assert m.get("v_xor_b32_e32", 0) < 10 and m["v_alignbit_b32"] == 32              # u12(): two alignbits per draw
    assert 16 <= m["v_fract_f64_e32"] <= 17 and mix["pipes"].get("vmem", 0) == 0    # nothing leaves the CU inside the loop
    assert 480 <= mix["pipes"]["valu"] <= 600
    assert sum(c.values()) == mix["pipes"]["valu"] + mix["pipes"]["lds"] + mix["pipes"].get("salu", 0) + mix["pipes"].get("vmem", 0) + mix["pipes"].get("other", 0)
    cyc = isa_mix.issue_cycles(mix, {"v_xor_b32": 1.1, "v_alignbit_b32": 1.75, "v_bitop3_b32": 1.75, "v_mad_u64_u32": 1.87, "v_mul_lo_u32": 1.74,
                                     "v_lshrrev_b64": 1.8, "v_fma_f64": 1.77, "v_mul_f64": 1.76, "v_add_f64": 1.76, "v_fract_f64": 1.77,
                                     "v_cvt_i32_f64": 1.77, "v_rcp_f64": 6.8, "v_cmp_lt_f64": 1.8, "v_ldexp_f64": 1.76, "v_exp_f32": 3.5,
                                     "ds_read_b128 (random": 21.6, "ds_read_b64 (random": 11.7, "ds_add_f64 (random": 41.0})
    assert 800 < cyc["valu"] < 1100 and cyc["lds"] == pytest.approx(16 * 21.6 + 16 * 41.0)


def test_histogram_copies_follow_the_placement_rule_and_the_register_budget():
    """mci_device.h hslot: 8 interleaved histogram copies and two 512-thread workgroups per CU for the 16-D Gaussian (80.8 KB of LDS);
    the interleaved-copy row of the issue-cost table prices its ds_add_f64; a 1-D integrand keeps the plain layout (nothing to gain),
    and so do a kernel that needs more than 128 VGPRs (two 512-thread workgroups would not share a CU) and one that needs at most 80
    (it runs six or more waves per SIMD without the copies' LDS)"""
    c2 = [b for b in BASELINE if b[0] == "c2"][0]
    eng = mci.Engine(c2[1](), c2[2](), device=-1)
    assert eng.histogram_copies() == 8
    eng.compile("vegas")
    assert eng.histogram_copies() == 8
    res = isa_mix.resources(eng.code_object("vegas"))["mci_vegas_batch"]
    assert res["max_threads"] == 512 and res["vgpr"] <= 128 and res["scratch"] == 0
    mix = isa_mix.loop_mix(eng.code_object("vegas"), "mci_vegas_batch", draws_per_sample=16)
    costs = {"ds_read_b128 (random": 21.6, "ds_add_f64 (random bins, 999-bin table)": 41.0, "ds_add_f64 (random bins, 8 interleaved copies)": 26.9}
    assert isa_mix.issue_cycles(mix, costs, default_valu=1.8, hist_copies=8)["lds"] == pytest.approx(16 * 21.6 + 16 * 26.9)
    assert isa_mix.issue_cycles(mix, costs, default_valu=1.8)["lds"] == pytest.approx(16 * 21.6 + 16 * 41.0)
    eng.close()
    one = mci.Engine(mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1]]), mci.Integrand("w[0] = log(x[0]) / sqrt(x[0]);"), device=-1)
    assert one.histogram_copies() == 1
    one.close()
    # a light kernel (6 draws, 63 VGPRs) runs six or more waves per SIMD in 256-thread workgroups: the copies would cap it at four
    light = mci.Engine(mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[6]]),
                       mci.Integrand("double q = 0.0; for (int i = 0; i < 6; ++i) q += x[i] * x[i]; w[0] = q;"), device=-1)
    assert light.histogram_copies() == 8          # the rule's choice before the kernel exists
    light.compile("vegas")
    res = isa_mix.resources(light.code_object("vegas"))["mci_vegas_batch"]
    # (launch bound 512: light plain-layout kernels are compiled for the 512-thread workgroups their mid-size launches use -- it costs a
    # kernel of <= 128 registers nothing; big launches still run 256-thread workgroups, six or more waves per SIMD)
    assert res["vgpr"] <= 80 and light.histogram_copies() == 1 and res["max_threads"] == 512, res
    light.close()
    # C5 :vegas (12 draws on one grid, 4 integrands): the copies that fit next to its tables, two 512-thread workgroups per CU
    c5 = [b for b in BASELINE if b[0] == "c5"][0]
    mid = mci.Engine(c5[1](), c5[2](), device=-1)
    mid.compile("vegas")
    res = isa_mix.resources(mid.code_object("vegas"))["mci_vegas_batch"]
    assert 80 < res["vgpr"] <= 128 and mid.histogram_copies() == 4 and res["max_threads"] == 512 and res["scratch"] == 0, res
    mid.close()
    # an integrand that keeps every draw and every intermediate alive: more than 128 VGPRs -> the plain layout, 256 threads
    body = """double m = 0.0; for (int i = 0; i < 16; ++i) m += x[i]; m *= 0.0625;
              double y[16], q = 0.0; for (int i = 0; i < 16; ++i) { y[i] = sin(x[i] - m) * cos(x[(i + 7) % 16] + m); q += y[i]; }
              double p = 1.0; for (int i = 0; i < 16; ++i) p *= 1.0 + (y[i] - q) * exp(x[15 - i] - y[(i + 5) % 16]); w[0] = p;"""
    fat = mci.Engine(c2[1](), mci.Integrand(body), device=-1)
    assert fat.histogram_copies() == 8           # the rule's choice before the kernel exists
    fat.compile("vegas")
    res = isa_mix.resources(fat.code_object("vegas"))["mci_vegas_batch"]
    assert (fat.histogram_copies() == 1 and res["max_threads"] == 256) if res["vgpr"] > 128 else fat.histogram_copies() == 8, res
    assert res["vgpr"] > 128, "the test integrand no longer exceeds the register budget: make it fatter"
    fat.close()


def test_branch_targets_and_classes():
    assert isa_mix._branch_target(0x100, "s_cbranch_execnz", "65521") == 0x100 + 4 - 4 * 15
    assert isa_mix._branch_target(0x100, "s_branch", "3") == 0x110
    assert isa_mix.classify("v_bitop3_b32") == ("valu", "valu_bitop3")
    assert isa_mix.classify("v_bitop3_b32", "v45, v51, v60, s59 bitop3:0x96") == ("valu", "valu_bitop3")
    assert isa_mix.classify("v_bitop3_b32", "v45, v51, v60, v59 bitop3:0x96") == ("valu", "valu_bitop3_vgpr")
    assert isa_mix.classify("v_fma_f64") == ("valu", "valu_f64_fma")
    assert isa_mix.classify("v_lshl_add_u32") == ("valu", "valu_b32_3src")
    assert isa_mix.classify("v_xor_b32_e32") == ("valu", "valu_b32")
    assert isa_mix.classify("ds_add_f64") == ("lds", "lds_add_f64")
    assert isa_mix.classify("global_load_dwordx4") == ("vmem", "vmem")


@pytest.mark.parametrize("bits,rounds,threads,copies", [(52, 10, 512, 8), (52, 7, 512, 8), (32, 10, 512, 8), (32, 7, 1024, 16)])
def test_headline_kernel_of_every_stream_is_pipelined_and_does_not_spill(bits, rounds, threads, copies):
    """the 16-D Gaussian under the default stream and the three opt-in ones: the pipelined sample loop (two samples per trip), the
    histogram-copy plan the library picks for that stream, no scratch"""
    c2 = [b for b in BASELINE if b[0] == "c2"][0]
    eng = mci.Engine(c2[1](), c2[2](), device=-1, rng_bits=bits, rng_rounds=rounds)
    eng.compile("vegas")
    res = isa_mix.resources(eng.code_object("vegas"))["mci_vegas_batch"]
    mix = isa_mix.loop_mix(eng.code_object("vegas"), "mci_vegas_batch", draws_per_sample=16)
    assert res["scratch"] == 0 and res["vgpr_spill"] == 0 and res["vgpr"] <= 128 and res["lds"] == 0, res
    assert res["max_threads"] == threads and eng.histogram_copies() == copies
    assert mix["samples_per_trip"] == 2 and mix["mnemonics"]["ds_read_b128"] == 16 and mix["mnemonics"]["ds_add_f64"] == 16
    eng.close()


def test_jit_units_are_built_without_the_exec_mask_pass_that_miscompiled_one(tmp_path):
    """csrc/mci_jit.h: every JIT unit is compiled with -amdgpu-opt-exec-mask-pre-ra=0 -- si-optimize-exec-masking-pre-ra of ROCm 7.2's
    backend put the histogram adds of one campaign layout's :vegasmc group kernel into the wrong bins (profiles/r05_fuzz.txt; the GPU side
    of it: tests/test_hip_spec.py), and the pass is worth nothing on these kernels (profiles/r06_ablation.txt).  The pass lists of that
    layout's units (-opt-bisect-limit=-1: group kernel, lane-per-chain kernel, :vegas kernel) must not hold it."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AMD_COMGR_CACHE="0", MCI_JIT_FLAGS="-mllvm -opt-bisect-limit=-1", MCI_KERNEL_CACHE=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "repro_compile.py"), "vegasmc_lanes", "vegasmc", "vegas"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    log = r.stderr + r.stdout
    if "machine-scheduler on function (mci_vegasmc_spec)" not in log:
        pytest.skip("this toolchain does not list its passes under -opt-bisect-limit")
    for kernel in ("mci_vegasmc_spec", "mci_vegasmc_chains", "mci_vegas_batch"):
        assert "machine-scheduler on function (%s)" % kernel in log, kernel
        assert "si-optimize-exec-masking-pre-ra on function (%s)" % kernel not in log, kernel
        assert "si-optimize-exec-masking on function (%s)" % kernel in log, kernel    # (the post-RA pass of the default pipeline stays)


def test_kernel_cache_is_keyed_by_the_compiler(tmp_path, monkeypatch):
    """csrc/mci_jit.h: the cache key holds which compiler made the code object -- hiprtc's version, the files of libhiprtc and of
    libamd_comgr (the clang / LLVM that compiles), the target -- so that a ROCm upgrade or downgrade never serves another compiler's
    objects.  Two identities -> two file names for the same source; the same identity -> the cached file."""
    import ctypes as C
    from mcintegration_jl_amd._lib import lib
    monkeypatch.setenv("MCI_KERNEL_CACHE", str(tmp_path))
    buf = C.create_string_buffer(1024)
    lib().mci_debug_compiler_id(None, buf, len(buf))
    ident = buf.value.decode()
    assert ident.startswith("hiprtc ") and "libhiprtc.so" in ident and "libamd_comgr.so" in ident and ident.endswith("gfx950"), ident
    assert all(int(part.rsplit(":", 1)[1]) > 100000 for part in ident.split(" | ")[1:3]), ident      # (file sizes: the libraries were found)

    def obj():
        return _code_object(lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]]), mci.catalog.x2y2, None, "vegas")
    try:
        a = obj()
        assert os.path.dirname(a) == str(tmp_path) and obj() == a
        lib().mci_debug_compiler_id(b"hiprtc 7.3 | libhiprtc.so.7.3.70300:850000 | libamd_comgr.so.3.1.0:163000000 | gfx950", buf, len(buf))
        b = obj()
        lib().mci_debug_compiler_id(b"hiprtc 7.1 | libhiprtc.so.7.1.70100:840000 | libamd_comgr.so.2.9.0:161000000 | gfx950", buf, len(buf))
        c = obj()
        assert len({a, b, c}) == 3 and all(os.path.exists(q) for q in (a, b, c))
    finally:
        lib().mci_debug_compiler_id(b"", buf, len(buf))
    assert buf.value.decode() == ident and obj() == a


def test_the_compiler_in_the_cache_key_is_the_one_the_process_resolves(tmp_path):
    """hiprtc opens comgr -- the clang / LLVM inside -- by soname: the first matching copy loaded into a process serves everybody.  A
    process that imports PyTorch first compiles with the copies PyTorch bundles (another build: other code, another cache file);
    use_rocm_compiler() before `import torch` pins the ROCm installation's, and it stays pinned when torch comes in later: same identity,
    same file name and the SAME BYTES as a process that never saw torch."""
    import hashlib
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys, os, hashlib
sys.path.insert(0, %r)
order = sys.argv[1]
if order == "torch_first":
    import torch
import mcintegration_jl_amd as mci
pinned = mci.use_rocm_compiler()            # comgr + hiprtc only: no HIP runtime yet
assert pinned == (order != "torch_first")
if order == "mci_first":
    import torch
ident = mci.compiler_id()
if order != "plain":                        # ... and ONE HIP runtime in a process with torch: torch's, which libmci_hip.so binds to
    hip = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln})
    assert len(hip) == 1 and "/torch/" in hip[0], hip
eng = mci.Engine(mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]]), mci.catalog.x2y2(), device=-1)
eng.compile("vegas")
p = eng.code_object("vegas")
print("RESULT;%%s;%%s;%%s" %% (ident, os.path.basename(p), hashlib.md5(open(p, "rb").read()).hexdigest()))
""" % root
    out = {}
    for order in ("plain", "mci_first", "torch_first"):
        env = dict(os.environ, MCI_KERNEL_CACHE=str(tmp_path / order), AMD_COMGR_CACHE="0")
        r = subprocess.run([sys.executable, "-c", code, order], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        out[order] = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT;")][-1].split(";")[1:]
    assert out["mci_first"] == out["plain"], out                     # identity, file name, bytes
    assert "libamd_comgr.so." in out["plain"][0], out["plain"][0]    # (the installation's versioned file)
    if out["torch_first"][0] != out["plain"][0]:                     # this PyTorch bundles its own compiler: another key, never the same file
        assert out["torch_first"][1] != out["plain"][1], out
