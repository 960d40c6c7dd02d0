"""Static instruction mix of a sample-batch kernel's hot loop, read from its gfx950 code object.

Diagnostics for the roofline of bench.py (DESIGN.md "Measurement"): the VEGAS sample loop is bound by VALU issue and the
LDS pipe, not by HBM, so its ceiling is the cycle-weighted issue rate of ITS OWN instruction mix.  This module
disassembles a cached code object with llvm-objdump (no GPU needed), finds the sample loop -- the innermost backward
branch span holding the kernel's Philox multiplies -- and counts its instructions per issue class.  The classes carry the names of the
instruction forms `tools/issue_microbench.hip` measures on the box, so that mix x measured cycles = the bound.

The count is static: one trip of the loop = one sample per lane.  For the BASELINE :vegas kernels the loop body is
straight-line code (no inner loops; what remains under a forward branch is the ragged last trip), and the total agrees
with the dynamic SQ_INSTS_VALU / SQ_INSTS_LDS counters (profiles/).
"""
import os
import re
import subprocess

OBJDUMP = os.environ.get("MCI_LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
READELF = os.environ.get("MCI_LLVM_READELF", "/opt/rocm/lib/llvm/bin/llvm-readelf")

# issue classes -> the microbenchmark row whose cost they take (tools/issue_microbench.hip, kNames)
COST_KEY = {
    "valu_b32": "v_xor_b32",
    "valu_b32_3src": "v_alignbit_b32",
    "valu_bitop3": "v_bitop3_b32 (0x96 = xor3, one SGPR",
    "valu_bitop3_vgpr": "v_bitop3_b32 (three VGPR",
    "valu_mad_u64_u32": "v_mad_u64_u32",
    "valu_mul_u32": "v_mul_lo_u32",
    "valu_b64": "v_lshrrev_b64",
    "valu_f64_fma": "v_fma_f64",
    "valu_f64_mul": "v_mul_f64",
    "valu_f64_add": "v_add_f64",
    "valu_f64_fract": "v_fract_f64",
    "valu_f64_cvt": "v_cvt_i32_f64",
    "valu_f64_rcp": "v_rcp_f64",
    "valu_f64_cmp": "v_cmp_lt_f64",
    "valu_f64_ldexp": "v_ldexp_f64",
    "valu_trans_f32": "v_exp_f32",
    "lds_read_b128": "ds_read_b128",
    "lds_read_b64": "ds_read_b64 (random",
    "lds_add_f64": "ds_add_f64 (random",
    "lds_other": "ds_read_b64 (random",
}


# Issue cycles of a wave64 instruction on a CDNA4 SIMD, MEASURED AND ROUNDED -- not a published table: 32-bit VOP1/VOP2 forms 2 cycles
# and fp64 / the 64-bit integer forms 4 are the guide's figures (MI355X_MICROARCH.md), transcendentals at quarter rate 8, f64 rcp 16; the
# 4 cycles of v_mad_u64_u32 / v_mul_*_u32 and of every 32-bit form with three sources or an SGPR third source (63 instructions of the
# headline loop) come from THIS repository's microbenchmark (tools/issue_microbench.hip, profiles/r02_issue_costs.txt: 1.73-1.80 ns per
# wave-instruction and SIMD at ~2.3 GHz = 4.0-4.1 cycles) rounded to the cycle.  bench.py prices the loop's mix with these at the clock
# the kernel MEASURED itself running at (mci_kernel_clocks) -- roofline.valu_rounded_measured -- and, next to it, with the guide's flat
# 2 cycles for every 32-bit form (FLAT_CYCLES: roofline.valu_flat_2cycle), which is the pessimistic reading of the same launch.
DATASHEET_CYCLES = {
    "valu_b32": 2, "valu_bitop3_vgpr": 2, "valu_bitop3": 4, "valu_b32_3src": 4, "valu_mad_u64_u32": 4, "valu_mul_u32": 4, "valu_b64": 4,
    "valu_f64_fma": 4, "valu_f64_mul": 4, "valu_f64_add": 4, "valu_f64_fract": 4, "valu_f64_cvt": 4, "valu_f64_cmp": 4, "valu_f64_ldexp": 4,
    "valu_f64_rcp": 16, "valu_trans_f32": 8,
}


FLAT_CYCLES = dict(DATASHEET_CYCLES, valu_bitop3=2, valu_b32_3src=2)   # the guide's flat 2-cycle issue for EVERY 32-bit VALU form


def datasheet_valu_cycles(mix, table=None):
    """(cycles per wave and sample on the VALU pipe, {class: (n, cycles each)}) under DATASHEET_CYCLES (measured, rounded; the default)
    or another table (FLAT_CYCLES)"""
    table = DATASHEET_CYCLES if table is None else table
    per, tot = {}, 0.0
    for cls, n in mix["classes"].items():
        if cls.startswith("valu"):
            c = table.get(cls, 2)
            per[cls] = (n, c)
            tot += n * c
    return tot, per


# 32-bit forms with three source operands (VOP3-only): measured at the f64 rate, not at the VOP2 rate (tools/issue_microbench.hip)
_THREE_SOURCE_B32 = {"v_alignbit_b32", "v_lshl_add_u32", "v_add_lshl_u32", "v_and_or_b32", "v_lshl_or_b32", "v_or3_b32", "v_add3_u32",
                     "v_xad_u32", "v_bfe_u32", "v_bfe_i32", "v_bfi_b32", "v_perm_b32", "v_mad_u32_u24", "v_mad_i32_i24", "v_min3_u32",
                     "v_max3_u32", "v_med3_u32", "v_alignbyte_b32", "v_fma_f32", "v_mad_u32_u16", "v_sad_u32", "v_lerp_u8"}


def classify(mn, ops=""):
    """mnemonic (and operand string) -> (pipe, class)"""
    m = re.sub(r"_(e32|e64|sdwa|dpp)$", "", mn)
    if m.startswith("ds_"):
        if m == "ds_read_b128":
            return "lds", "lds_read_b128"
        if m in ("ds_read_b64", "ds_read2_b64", "ds_read2_b32", "ds_read_b32"):
            return "lds", "lds_read_b64"
        if m in ("ds_add_f64", "ds_add_rtn_f64"):
            return "lds", "lds_add_f64"
        return "lds", "lds_other"   # ds_write*, ds_bpermute, ds_swizzle ...
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem", "vmem"
    if m.startswith("s_"):
        return "salu", "salu"
    if not m.startswith("v_"):
        return "other", "other"
    if m == "v_bitop3_b32":
        # with all three sources in VGPRs the instruction issues at the VOP2 rate, with an SGPR source (a wave-uniform Philox round
        # key) at the three-source rate: two rows of the issue-cost table
        srcs = [o.strip() for o in ops.split(",")[1:4]]
        if len(srcs) == 3 and all(o.startswith("v") for o in srcs):
            return "valu", "valu_bitop3_vgpr"
        return "valu", "valu_bitop3"
    if m == "v_mad_u64_u32" or m == "v_mad_i64_i32":
        return "valu", "valu_mad_u64_u32"
    if m in ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mul_lo_i32"):
        return "valu", "valu_mul_u32"
    if m.endswith("_f64") or "_f64_" in m:
        if m.startswith(("v_fma_", "v_fmac_", "v_mad_", "v_div_fmas", "v_div_fixup")):
            return "valu", "valu_f64_fma"
        if m.startswith("v_mul_"):
            return "valu", "valu_f64_mul"
        if m.startswith(("v_add_", "v_sub_", "v_max_", "v_min_")):
            return "valu", "valu_f64_add"
        if m.startswith(("v_fract_", "v_floor_", "v_trunc_", "v_ceil_", "v_rndne_", "v_frexp_")):
            return "valu", "valu_f64_fract"
        if m.startswith("v_cvt_"):
            return "valu", "valu_f64_cvt"
        if m.startswith(("v_rcp_", "v_rsq_", "v_sqrt_")):
            return "valu", "valu_f64_rcp"
        if m.startswith(("v_cmp", "v_cmpx")):
            return "valu", "valu_f64_cmp"
        if m.startswith(("v_ldexp_", "v_div_scale")):
            return "valu", "valu_f64_ldexp"
        return "valu", "valu_f64_fma"
    if m.endswith(("_b64", "_u64", "_i64")) and not m.startswith(("v_cmp", "v_mov")):
        return "valu", "valu_b64"
    if m in _THREE_SOURCE_B32:
        return "valu", "valu_b32_3src"
    if m in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"):
        return "valu", "valu_trans_f32"
    return "valu", "valu_b32"


_LINE = re.compile(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")


def disassemble(path):
    """{kernel name: [(address, mnemonic, operands)]} of a gfx950 code object"""
    out = subprocess.run([OBJDUMP, "-d", path], check=True, capture_output=True, text=True).stdout
    kernels, cur = {}, None
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <([A-Za-z_0-9.$]+)>:", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        m = _LINE.match(line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return kernels


def _branch_target(addr, mn, ops):
    """byte address a s_branch / s_cbranch_* at `addr` jumps to (simm16 counts dwords from the next instruction)"""
    if not (mn == "s_branch" or mn.startswith("s_cbranch")):
        return None
    m = re.match(r"^(\d+)", ops.strip())
    if not m:
        return None
    off = int(m.group(1))
    if off >= 32768:
        off -= 65536
    return addr + 4 + 4 * off


def hot_loop(insts):
    """(first index, last index) of the sample loop: the SMALLEST backward-branch span that holds at least 25 % of the kernel's
    Philox multiplies (v_mad_u64_u32) -- every sample/chain-step loop draws its uniforms inside; block layout may put other,
    wider backward branches around it.  The :vegas kernels carry the loop twice, specialised on measurefreq == 1 (mci_device.h
    vegas_batch) and the pipelined form of it holds two samples per trip, followed by a straight-line tail: the smaller of the two
    loops is the measurefreq == 1 form, the one the BASELINE configurations run"""
    index = {a: i for i, (a, _, _) in enumerate(insts)}
    mads = [i for i, (_, mn, _) in enumerate(insts) if mn == "v_mad_u64_u32"]
    best = None
    for i, (a, mn, ops) in enumerate(insts):
        t = _branch_target(a, mn, ops)
        if t is None or t > a or t not in index:
            continue
        j = index[t]
        inside = sum(1 for m in mads if j <= m <= i)
        if mads and inside < 0.25 * len(mads):
            continue
        if best is None or i - j < best[1] - best[0]:
            best = (j, i)
    if best is None:
        raise ValueError("no loop holding the Philox multiplies found")
    return best


def loop_mix(path, kernel="mci_vegas_batch", draws_per_sample=None):
    """static per-SAMPLE instruction counts of the kernel's hot loop:
    {"classes": {class: n}, "pipes": {pipe: n}, "mnemonics": {mn: n}, "inner_backward_branches": k, "span": (lo, hi),
     "samples_per_trip": t}.  With draws_per_sample given, a loop body that holds several samples (the pipelined :vegas loop takes two
    per trip) is recognised by its v_cvt_i32_f64 count -- one per Continuous draw -- and every count is divided by t."""
    insts = disassemble(path)[kernel]
    lo, hi = hot_loop(insts)
    classes, pipes, mns, inner = {}, {}, {}, 0
    for i in range(lo, hi + 1):
        a, mn, ops = insts[i]
        t = _branch_target(a, mn, ops)
        if t is not None and t <= a and i != hi:
            inner += 1
        if mn in ("s_nop", "s_waitcnt", "s_endpgm", "s_barrier", "s_sleep"):
            continue
        pipe, cls = classify(mn, ops)
        classes[cls] = classes.get(cls, 0) + 1
        pipes[pipe] = pipes.get(pipe, 0) + 1
        mns[mn] = mns.get(mn, 0) + 1
    spt = 1
    if draws_per_sample:
        spt = max(1, int(round(sum(n for mn, n in mns.items() if mn.startswith("v_cvt_i32_f64")) / float(draws_per_sample) - 0.2)))
    if spt > 1:
        def per_sample(d):
            return {k: (v // spt if v % spt == 0 else v / float(spt)) for k, v in d.items()}
        classes, pipes, mns = per_sample(classes), per_sample(pipes), per_sample(mns)
    return {"kernel": kernel, "classes": classes, "pipes": pipes, "mnemonics": mns, "inner_backward_branches": inner,
            "span": (insts[lo][0], insts[hi][0]), "loop_instructions": hi - lo + 1, "kernel_instructions": len(insts),
            "samples_per_trip": spt}


def resources(path):
    """{kernel: {"vgpr": n, "sgpr": n, "vgpr_spill": n, "scratch": bytes, "lds": bytes, "max_threads": launch bound}} from the code object's metadata"""
    out = subprocess.run([READELF, "--notes", path], check=True, capture_output=True, text=True).stdout
    res, cur = {}, {}
    for line in out.splitlines():
        m = re.match(r"^\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "name", "agpr_count",
                 "max_flat_workgroup_size"):
            cur[k] = v
        if k == "wavefront_size":   # last key of a kernel entry
            if "name" in cur:
                res[cur["name"]] = {"vgpr": int(cur.get("vgpr_count", 0)), "sgpr": int(cur.get("sgpr_count", 0)),
                                    "vgpr_spill": int(cur.get("vgpr_spill_count", 0)),
                                    "scratch": int(cur.get("private_segment_fixed_size", 0)),
                                    "lds": int(cur.get("group_segment_fixed_size", 0)),
                                    "max_threads": int(cur.get("max_flat_workgroup_size", 0))}
            cur = {}
    return res


def issue_cycles(mix, costs, default_valu=None, hist_copies=1):
    """cycle-weighted cost of one loop trip per pipe.  costs: {microbench op name: cycles per wave-instruction}, matched
    by prefix through COST_KEY; hist_copies: the kernel's interleaved histogram copies (its ds_add_f64 are priced with that access
    pattern: `ds_add_f64 (random bins, N interleaved copies`).  Returns {"valu": cycles, "lds": cycles, "per_class": {class: (n, cycles each)}}"""
    def cost_of(cls):
        key = COST_KEY.get(cls)
        if cls == "lds_add_f64" and hist_copies > 1:
            key = "ds_add_f64 (random bins, %d interleaved copies" % hist_copies
        if key is not None:
            for name, c in costs.items():
                if name.startswith(key):
                    return c
            short = [name for name in costs if key.startswith(name)]   # a table that has only the bare instruction name
            if short:
                return costs[max(short, key=len)]
        return None
    per, tot = {}, {"valu": 0.0, "lds": 0.0}
    base = cost_of("valu_b32") if default_valu is None else default_valu
    for cls, n in mix["classes"].items():
        pipe = "lds" if cls.startswith("lds") else "valu" if cls.startswith("valu") else None
        if pipe is None:
            continue
        c = cost_of(cls)
        if c is None:
            c = base if pipe == "valu" else cost_of("lds_read_b64")
        per[cls] = (n, c)
        tot[pipe] += n * c
    tot["per_class"] = per
    return tot
