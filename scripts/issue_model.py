#!/usr/bin/env python3
"""issue_model.py — what fraction of a launch's duration the vector ALUs spend ISSUING, per kernel.

  python3 scripts/issue_model.py [round_tag, default round2]   ->  profiles/<tag>_issue_model.json

    issue_cycles   = SQ_INSTS_VALU (dynamic wave-instructions of the launch, rocprofv3 --pmc)
                     x mean issue cost of the kernel's hot loop   [cycles per wave-instruction per SIMD]
    issue_fraction = issue_cycles / (1024 SIMDs x shader cycles of the launch)

The mean issue cost prices the hot loop's per-opcode histogram (scripts/isa_hist.py, static ISA of the shipped
kernels) with the cycles-per-instruction MEASURED by scripts/valu_calib.hip at the kernel's waves/SIMD
(profiles/round2_valu_calib.json).  Shader cycles of the launch = GRBM_GUI_ACTIVE / 8 XCDs.  This replaces round 1's
"VALU busy" column, which was 4 x SQ_ACTIVE_INST_VALU / cycles and read exactly 4.00 cycles per instruction for every
kernel: that counter counts instructions in quad-cycle units, not occupancy.  (measurement tool, not product code)
"""
import csv
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "round2"

# kernel (as rocprofv3 names it, template arguments stripped) -> (isa_hist kernel key, waves/SIMD it runs at)
KERNELS = {
    "k_ssim_gauss": ("k_ssim_gauss<256, 2, 0>", 6),       # 80 VGPRs
    "k_dct8_march": ("k_dct8_march<true, true, false, false>", 3),  # 160 VGPRs
    "k_block_sad": ("k_block_sad", 4),                   # 122 VGPRs
    "k_canny_nms2": ("k_canny_nms2", 6),                 # 77 VGPRs
    "k_canny_hyst_all": ("k_canny_hyst_all", 6),
    "k_canny_hyst_list": ("k_canny_hyst_list", 6),
    "k_bgr2gray_hist": ("k_bgr2gray_hist<true>", 8),
}


def main():
    valu = json.load(open(os.path.join(REPO, "profiles", "%s_c3_valu.json" % tag)))["kernels"]
    stats = {}
    for r in csv.DictReader(open(os.path.join(REPO, "profiles", "%s_c3_kernel_stats.csv" % tag))):
        name = r["Name"].split("(")[0].replace("void ", "").replace("vqa::", "")
        # the counters are averaged over the full-batch launches only; MaxNs-side launches are those (a kernel that also
        # runs once per step on the single prev0 frame has a misleading AverageNs)
        avg, mx = float(r["AverageNs"]), float(r["MaxNs"])
        stats[name] = avg if avg > 0.6 * mx else None
    out = {"tag": tag, "workload": "c3 (256 x 1080p, full suite + PSNR/SSIM)", "simds": 1024,
           "method": __doc__.split("\n\n")[1], "kernels": {}}
    hist_cache = {}
    for short, (key, W) in KERNELS.items():
        if W not in hist_cache:
            path = "/tmp/isa_hist_w%d.json" % W
            subprocess.run([sys.executable, os.path.join(REPO, "scripts", "isa_hist.py"), "--waves", str(W), "--json", path],
                           check=True, capture_output=True)
            hist_cache[W] = json.load(open(path))["kernels"]
        k = hist_cache[W].get(key)
        dyn = next((v for n, v in valu.items() if n.split("<")[0] == short), None)
        if k is None or dyn is None:
            continue
        loops = sorted(k["inner_loops"], key=lambda x: -x["valu_issue_cycles"])  # the loop that costs the most issue time
        hot = loops[0] if loops else k["whole"]
        cost = hot["mean_cost"]
        # VALU-bound kernels spend their time in the hot loop; kernels with a large non-loop part use the whole-kernel mix
        whole_cost = k["whole"]["mean_cost"]
        insts, cycles = dyn["valu_insts_per_launch"], dyn["shader_cycles_per_launch"]
        ns = next((v for n, v in stats.items() if n.split("<")[0] == short), None)
        ent = {"waves_per_simd": W, "valu_insts_per_launch": insts, "shader_cycles_per_launch": cycles,
               "clock_GHz": round(cycles / ns, 3) if ns else None, "avg_launch_ms": round(ns / 1e6, 4) if ns else None,
               "launch_ms_from_cycles_at_2.1GHz": round(cycles / 2.1e6, 4),
               "hot_loop_valu": hot["valu"], "hot_loop_mean_issue_cycles": cost, "whole_kernel_mean_issue_cycles": whole_cost,
               "hot_loop_top_opcodes": hot["top"][:8],
               "issue_fraction_hot_mix": round(insts * cost / (1024.0 * cycles), 3),
               "issue_fraction_whole_mix": round(insts * whole_cost / (1024.0 * cycles), 3)}
        if short == "k_block_sad":
            # two regimes (QSAD loop at 16.5 cycles per instruction, everything else at ~3): count the QSADs exactly -
            # 256 per 64x16-pixel tile per wave, 67 x 30 tiles per 1080p frame, 256 frames per launch
            q = 256 * 67 * 30 * 256
            qc = hist_cache[W][key]["inner_loops"] and max(
                (l for l in hist_cache[W][key]["inner_loops"]), key=lambda l: l["valu_issue_cycles"])
            qsad_cost = 16.55  # v_qsad_pk_u16_u8 at 4 waves/SIMD (round2_valu_calib.json)
            other = (whole_cost * k["whole"]["valu"] - qsad_cost * 64) / max(k["whole"]["valu"] - 64, 1)
            ent["qsad_insts_per_launch"] = q
            ent["issue_fraction_exact_split"] = round((q * qsad_cost + (insts - q) * other) / (1024.0 * cycles), 3)
            ent["note"] = "issue_fraction_exact_split is the one to read: QSADs counted exactly, the rest priced at the non-QSAD mean (%.2f cycles)" % other
        out["kernels"][short] = ent
        print("%-20s W=%d  %.4g VALU inst  mean %.2f (whole %.2f) cyc  => issue fraction %.2f (%.2f)   %.3f ms @ %.2f GHz" % (
            short, W, insts, cost, whole_cost, ent["issue_fraction_hot_mix"], ent["issue_fraction_whole_mix"],
            ent["avg_launch_ms"] or 0, ent["clock_GHz"] or 0))
    json.dump(out, open(os.path.join(REPO, "profiles", "%s_issue_model.json" % tag), "w"), indent=1)


if __name__ == "__main__":
    main()
