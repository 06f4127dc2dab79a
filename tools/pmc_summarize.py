"""gpurun_out/r02/pmc_{gateup,down,o}.json (tools/pmc_parse.py output of tools/collect_r02.sh) -> profiles/r02_*_gemm_pmc.json:
the counters with their units, the gfx950 corrections of MI355X_MICROARCH.md's HBM section, and the derived per-launch figures
bench.py's roofline.traffic cites.  Usage: python tools/pmc_summarize.py [gpurun_out/r02] [profiles]"""
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r02"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"
tag = sys.argv[3] if len(sys.argv) > 3 else "r02"  # file prefix: profiles/<tag>_<shape>_gemm_pmc.json
M = 8192
SH = {
    "gateup": dict(N=16384, K=2048, desc="gate|up projection, fp16 operands, fused-RMSNorm row scale + SiLU*up epilogue",
                   alg=lambda: M * 2048 * 2 + 16384 * 2048 * 2 + M * 8192 * 2),
    "down": dict(N=2048, K=8192, desc="down projection, fp16 operands, residual added to the 16-bit residual stream in place + partial "
                                        "sums of squares (TCAVT_EPI_NORM_OUT, C == NULL), two-barrier deep-prefetch form",
                 alg=lambda: M * 8192 * 2 + 2048 * 8192 * 2 + 2 * M * 2048 * 2),
    "o": dict(N=2048, K=2048, desc="o projection, fp16 operands, residual added to the 16-bit residual stream in place + partial sums "
                                    "of squares (TCAVT_EPI_NORM_OUT, C == NULL)",
              alg=lambda: M * 2048 * 2 + 2048 * 2048 * 2 + 2 * M * 2048 * 2),
}
SH["attn"] = dict(N=0, K=0, desc="decoder attention (causal AND key-valid, GQA 32/8 heads x 64, B = 32, L = 256, fp16): one workgroup per "
                  "(sample, kv head), K and V^T staged once in LDS", alg=lambda: 83886080)
for sh, info in SH.items():
    if not os.path.exists(os.path.join(src, f"pmc_{sh}.json")):
        continue
    raw = json.load(open(os.path.join(src, f"pmc_{sh}.json")))
    (kname, c), = raw.items()
    us = c.pop("_avg_us_under_pmc")
    rd, wr = int(c["FETCH_SIZE"] * 1024 * 2), int(c["WRITE_SIZE"] * 1024)
    alg = info["alg"]()
    out = {
        "kernel": (f"{kname} = tcavt::gemm_bf16_w4_kernel (4-wave 256x256 kernel; {info['desc']})" if sh != "attn" else
                   f"{kname} = tcavt::attn_causal_gqa_kernel ({info['desc']})"),
        "shape": {"M": M, "N": info["N"], "K": info["K"]} if sh != "attn" else {"B": 32, "L": 256, "q_heads": 32, "kv_heads": 8, "head_dim": 64, "kv_len": "U{144..256}"},
        "command": f"tools/collect_{tag}.sh  ==  rocprofv3 --pmc <COUNTERS> --kernel-trace -- python3 tools/pmc_gemm.py 0 " + sh +
                   ", one pass per counter group (FETCH_SIZE and WRITE_SIZE each in a pass of their own), summarised by "
                   "tools/pmc_parse.py + tools/pmc_summarize.py",
        "units": "mean over launches of the per-instance counter value as rocprofv3 stores it: GRBM_GUI_ACTIVE per XCD (= kernel "
                 "cycles), SQ_* per shader-engine instance (32 instances; x32 = chip total), FETCH_SIZE / WRITE_SIZE in KiB chip total",
        "counters_mean_per_launch": c,
        "kernel_us_under_pmc": us,
        "corrections": "gfx950: FETCH_SIZE (KiB) reports half of the bytes of wide coalesced reads -> doubled (MI355X_MICROARCH.md, "
                       "HBM section); WRITE_SIZE (KiB) is exact for 16-byte streaming stores",
        "hbm_side_read_bytes": rd, "hbm_side_write_bytes": wr, "traffic_bytes_per_launch": rd + wr,
        "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round((rd + wr) / alg, 2),
        "mfma_busy_fraction_at_actual_clock": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 32), 3),
        "wait_over_active_inst": round(c["SQ_WAIT_INST_ANY"] / c["SQ_ACTIVE_INST_ANY"], 2),
        "l2_hit_rate": round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3),
        "note": "memory-side counters include Infinity-Cache hits: each XCD (private 4 MiB L2) re-fetches the operand panels its 32 "
                "CUs stream; profiled passes run at a lower clock than un-profiled launches (in situ: profiles/" + tag + "_bench_default.json)",
    }
    if sh == "attn":
        out["achieved_GBps_under_pmc"] = round(alg / (sum(us) / len(us) * 1e-6) / 1e9, 1)
        out["hbm_fraction_of_8TBps"] = round(out["achieved_GBps_under_pmc"] / 8000.0, 3)
    with open(os.path.join(dst, f"{tag}_{sh}_gemm_pmc.json" if sh != "attn" else f"{tag}_attention_pmc.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(sh, out["traffic_bytes_per_launch"], out["traffic_over_algorithmic"], out["mfma_busy_fraction_at_actual_clock"], out["l2_hit_rate"])
