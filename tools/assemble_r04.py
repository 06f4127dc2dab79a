"""Copy the summaries of tools/collect_r04.sh (gpurun_out/r04/) into profiles/r04_*: the variants file (one entry per bench
line / generation run) and the text summaries the judged numbers are read from.  Run here after the GPU call merged its output."""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r04")
DST = os.path.join(ROOT, "profiles")


def main():
    var = {}
    for name in ("bench_forward", "bench_nopipeline", "bench_bf16", "bench_feed_host", "bench_rccl1", "bench_lora", "bench_lora_full", "bench_lora_bf16",
                 "bench_gloo2", "generate_b4", "generate_b8", "generate_b16", "generate_b32", "generate_b8_row_major_activations", "generate_b8_greedy", "generate_b8_one_stage_sampler", "generate_b8_row_major_weights",
                 "generate_b32_row_major_weights", "generate_b32_row_major_activations"):
        p = os.path.join(SRC, name + ".json")
        if not os.path.exists(p):
            continue
        line = [ln for ln in open(p).read().splitlines() if ln.startswith("{")]
        if line:
            var[name] = json.loads(line[-1])
    extra = os.path.join(DST, "r04_variants_notes.json")
    if os.path.exists(extra):
        var.update(json.load(open(extra)))
    json.dump(var, open(os.path.join(DST, "r04_variants.json"), "w"), indent=1)
    for src, dst in (("lora_kernel_stats.csv", "r04_lora_trainable_kernel_stats.csv"), ("lora_timeline.txt", "r04_lora_trainable_timeline.txt"),
                     ("decode_step_breakdown.txt", "r04_decode_step_breakdown.txt"), ("config4_sweep.txt", "r04_config4_sweep.txt"),
                     ("config5_eval_k.txt", "r04_config5_eval_k.txt"), ("pipe_trace.txt", "r04_pipe_trace.txt"),
                     ("bench_sampler.txt", "r04_token_selection_microbench.txt")):
        if os.path.exists(os.path.join(SRC, src)):
            shutil.copyfile(os.path.join(SRC, src), os.path.join(DST, dst))
    print("variants:", ", ".join(f"{k} {v.get('ms_per_step', v.get('decode_ms_per_step'))}" for k, v in var.items()))


if __name__ == "__main__":
    main()
