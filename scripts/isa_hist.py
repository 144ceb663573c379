#!/usr/bin/env python3
"""isa_hist.py — per-opcode VALU histogram of the hot loops of each gfx950 kernel, priced with the measured
issue costs of scripts/valu_calib.hip (profiles/round2_valu_calib.json).

  python3 scripts/isa_hist.py [--waves W] [--json out.json] [kernel-name-substring ...]

For every kernel of real-time-video-quality-analysis_amd/csrc/*.hip: compile to assembly (same flags as the
Makefile), split the kernel into basic blocks, take every INNERMOST loop (a backward branch whose span holds no
other backward-branch target), and for each report the VALU opcode mix and its mean issue cost

    mean_cost = sum(count[op] * cyc[op]) / sum(count[op])         [cycles per VALU wave-instruction per SIMD]

with cyc[op] the calibrated cycles per wave-instruction at W waves per SIMD.  Together with the DYNAMIC VALU
instruction count of a launch (rocprofv3 --pmc SQ_INSTS_VALU) this gives the launch's vector-ALU issue time:

    issue_fraction = SQ_INSTS_VALU * mean_cost / (kernel_cycles * 1024 SIMDs)

(measurement tool, not product code)
"""
import argparse
import collections
import glob
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "real-time-video-quality-analysis_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=fast", "--offload-arch=gfx950", "--cuda-device-only", "-S"]


def load_costs(waves):
    path = os.path.join(REPO, "profiles", "round2_valu_calib.json")
    ops = json.load(open(path))["ops"]
    cost = {}
    for name, per_w in ops.items():
        base = name.split()[0].split("(")[0]
        if "dependent" in name or "vcc written" in name or "+" in name or "(literal" in name or "(inline" in name \
                or "modifier" in name or "(sgpr src)" in name and base != "v_fma_f32":
            continue
        if base == "v_cndmask_b32" and "sgpr" not in name:
            continue  # VCC not freshly written by the VALU: a 19-cycle artefact compiled code does not show
        key = base + ("(sgpr)" if "sgpr src" in name else "")
        cost[key] = per_w[str(waves)]["cyc_per_wave_instr"]
    return cost


# opcodes not probed individually, priced as the probed opcode of the same encoding/data-path class
ALIAS = {
    "v_sub_f32": "v_add_f32", "v_subrev_f32": "v_add_f32", "v_mac_f32": "v_fmac_f32", "v_sub_u32": "v_add_u32",
    "v_subrev_u32": "v_add_u32", "v_min_f32": "v_max_f32", "v_min_i32": "v_max_i32", "v_max_u32": "v_min_u32",
    "v_min_u16": "v_min_u32", "v_max_u16": "v_min_u32", "v_min3_u32": "v_med3_i32", "v_max3_u32": "v_med3_i32",
    "v_max3_i32": "v_med3_i32", "v_min3_i32": "v_med3_i32", "v_med3_u32": "v_med3_i32", "v_min3_f32": "v_max3_f32",
    "v_med3_f32": "v_max3_f32", "v_lshlrev_b64": "v_mad_u64_u32", "v_lshrrev_b64": "v_mad_u64_u32",
    "v_ashrrev_i64": "v_mad_u64_u32", "v_mad_i32_i24": "v_mad_u32_u24", "v_mul_i32_i24": "v_mul_u32_u24",
    "v_mul_hi_u32": "v_mul_lo_u32", "v_mul_hi_i32": "v_mul_lo_u32", "v_mad_u32_u16": "v_mad_u32_u24",
    "v_bfe_i32": "v_bfe_u32", "v_pk_sub_u16": "v_pk_sub_i16", "v_pk_add_i16": "v_pk_add_u16", "v_pk_max_u16": "v_pk_max_i16",
    "v_pk_min_u16": "v_pk_max_i16", "v_pk_min_i16": "v_pk_max_i16", "v_pk_lshlrev_b16": "v_pk_add_u16",
    "v_pk_lshrrev_b16": "v_pk_add_u16", "v_pk_ashrrev_i16": "v_pk_add_u16", "v_pk_mad_i16": "v_pk_mad_u16",
    "v_addc_co_u32": "v_add_co_u32", "v_sub_co_u32": "v_add_co_u32", "v_subb_co_u32": "v_add_co_u32",
    "v_subrev_co_u32": "v_add_co_u32", "v_cvt_f32_i32": "v_cvt_f32_u32", "v_cvt_i32_f32": "v_cvt_u32_f32",
    "v_cvt_f32_ubyte1": "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte2": "v_cvt_f32_ubyte0", "v_cvt_f64_f32": "v_add_f64",
    "v_cvt_f32_f64": "v_add_f64", "v_cvt_f64_u32": "v_add_f64", "v_cvt_f64_i32": "v_add_f64", "v_mul_f64": "v_fma_f64",
    "v_not_b32": "v_mov_b32", "v_xad_u32": "v_add3_u32", "v_add_lshl_u32": "v_lshl_add_u32", "v_sub_u16": "v_add_u32",
    "v_add_u16": "v_add_u32", "v_mul_lo_u16": "v_mul_u32_u24", "v_mad_u16": "v_mad_u32_u24", "v_lshlrev_b16": "v_lshlrev_b32",
    "v_lshrrev_b16": "v_lshrrev_b32", "v_sqrt_f32": "v_rcp_f32", "v_rsq_f32": "v_rcp_f32", "v_log_f32": "v_rcp_f32",
    "v_exp_f32": "v_rcp_f32", "v_rcp_iflag_f32": "v_rcp_f32", "v_bcnt_u32_b32": "v_mad_u32_u24", "v_mbcnt_lo_u32_b32": "v_mad_u32_u24",
    "v_mbcnt_hi_u32_b32": "v_mad_u32_u24", "v_readfirstlane_b32": "v_mov_b32", "v_readlane_b32": "v_mov_b32",
    "v_writelane_b32": "v_mov_b32", "v_accvgpr_write_b32": "v_mov_b32", "v_accvgpr_read_b32": "v_mov_b32",
    "v_ldexp_f32": "v_max_f32", "v_fract_f32": "v_cvt_f32_u32", "v_floor_f32": "v_cvt_f32_u32", "v_rndne_f32": "v_cvt_f32_u32",
    "v_trunc_f32": "v_cvt_f32_u32", "v_sad_u16": "v_sad_u8", "v_msad_u8": "v_sad_u8", "v_pk_fma_f16": "v_pk_fma_f32",
    "v_div_fixup_f32": "v_max3_f32", "v_div_fmas_f32": "v_max3_f32", "v_div_scale_f32": "v_max3_f32", "v_mul_legacy_f32": "v_mul_f32",
    "v_lshl_or_b32": "v_lshl_or_b32", "v_alignbyte_b32": "v_alignbit_b32", "v_cvt_pk_u8_f32": "v_perm_b32", "v_swap_b32": "v_perm_b32",
}
CMP_RE = re.compile(r"^v_cmpx?_[a-z_]+_(f|i|u)(16|32|64)$")


def price(op, operands, cost, unknown):
    suffixes = ("_e32", "_e64", "_dpp", "_sdwa", "_e64_dpp")
    dpp = op.endswith("_dpp")
    for s in suffixes:
        if op.endswith(s):
            op = op[: -len(s)]
    if dpp:  # every DPP form issues at the half-rate class (probed: v_mov_b32 dpp row_shr:1)
        return cost.get("v_mov_b32 dpp", cost["v_pk_add_u16"])
    if CMP_RE.match(op) or op.startswith("v_cmp"):
        return cost["v_cmp_gt_u32"]
    if op == "v_cndmask_b32":
        return cost["v_cndmask_b32"] if "v_cndmask_b32" in cost else cost["v_perm_b32"]
    base = ALIAS.get(op, op)
    # a scalar-register (or literal) source on a plain fp32 FMA costs the slow form (calibrated)
    if base in ("v_fma_f32", "v_fmac_f32", "v_add_f32", "v_mul_f32", "v_add_u32", "v_and_b32", "v_or_b32", "v_xor_b32") and re.search(r"(^|[ ,])(s\d+|s\[\d+:\d+\]|vcc_lo|vcc_hi)(,|$)", operands):
        # an SGPR source halves the rate of the full-rate opcodes (literals and inline constants do not): calibrated
        return cost.get("v_fma_f32(sgpr)", cost["v_fma_f32"])
    if base in cost:
        return cost[base]
    unknown[op] += 1
    return cost["v_perm_b32"]  # unprobed: priced at the half-rate class


def kernels_of(asm):
    """yield (mangled name, list of lines) for every kernel body in an assembly file"""
    names = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", asm, re.M))
    lines = asm.splitlines()
    cur, body = None, []
    for ln in lines:
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", ln)
        if m and m.group(1) in names:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            body.append(ln)
            if re.match(r"^\s*s_endpgm", ln) and False:
                pass
            if ln.startswith(".Lfunc_end"):
                yield cur, body
                cur = None


def analyse(body, cost):
    # index instructions and labels
    instrs, labels = [], {}
    for ln in body:
        t = ln.strip()
        m = re.match(r"^(\.LBB[\w]+):", t)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        if not t or t.startswith((";", ".", "//")):
            continue
        parts = t.split(None, 1)
        instrs.append((parts[0], parts[1] if len(parts) > 1 else ""))
    loops = []
    for i, (op, args) in enumerate(instrs):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = args.strip()
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i))
    inner = [l for l in loops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in loops)]
    out = []
    unknown = collections.Counter()

    def region(a, b):
        hist = collections.Counter()
        cyc = 0.0
        n = salu = vmem = lds = 0
        for op, args in instrs[a:b + 1]:
            if op.startswith("v_") and not op.startswith("v_mfma"):
                base = op
                for s in ("_e64_dpp", "_e32", "_e64", "_dpp", "_sdwa"):
                    if base.endswith(s):
                        base = base[: -len(s)] + ("_dpp" if s.endswith("dpp") else "")
                        break
                hist[base] += 1
                cyc += price(op, args, cost, unknown)
                n += 1
            elif op.startswith("s_"):
                salu += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                vmem += 1
            elif op.startswith("ds_"):
                lds += 1
        return dict(valu=n, salu=salu, vmem=vmem, lds=lds, valu_issue_cycles=round(cyc, 1),
                    mean_cost=round(cyc / n, 3) if n else 0.0, top=hist.most_common(14))

    for a, b in sorted(inner):
        r = region(a, b)
        if r["valu"] >= 8:
            out.append(dict(span=[a, b], **r))
    whole = region(0, len(instrs) - 1)
    return dict(whole=whole, inner_loops=out, unknown=dict(unknown))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--waves", type=int, default=4, choices=[1, 2, 3, 4, 6, 8])
    ap.add_argument("--json", default="")
    ap.add_argument("filters", nargs="*")
    args = ap.parse_args()
    cost = load_costs(args.waves)
    res = {}
    for src in sorted(glob.glob(os.path.join(CSRC, "k_*.hip"))):
        asm = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", "-", src], capture_output=True, text=True, check=True).stdout
        for name, body in kernels_of(asm):
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            short = dem.split("(")[0].replace("void ", "").replace("vqa::", "")
            if args.filters and not any(f in short for f in args.filters):
                continue
            res[short] = analyse(body, cost)
            r = res[short]
            print("== %s   whole: %d VALU, mean %.2f cyc" % (short, r["whole"]["valu"], r["whole"]["mean_cost"]))
            for lp in sorted(r["inner_loops"], key=lambda x: -x["valu"])[:4]:
                print("   loop %s: %d VALU (%d SALU, %d VMEM, %d LDS)  issue %.0f cyc  mean %.2f  %s" % (
                    lp["span"], lp["valu"], lp["salu"], lp["vmem"], lp["lds"], lp["valu_issue_cycles"], lp["mean_cost"],
                    " ".join("%s:%d" % (k[2:] if k.startswith("v_") else k, v) for k, v in lp["top"][:10])))
            if r["unknown"]:
                print("   unpriced (taken as half-rate):", r["unknown"])
    if args.json:
        json.dump(dict(waves=args.waves, cost_table=cost, kernels=res), open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
