"""bench.py's model-FLOP count (used for achieved_model_tflops) against SURVEY.md 8d's table."""
import importlib.util
import os


def _bench():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gflop_per_sample_matches_survey_table():
    from tcavt_amd import config

    b = _bench()
    cfg = config.PRESETS["llama32_1b"](seq_len=18, out_len=30, use_lora=True)
    assert abs(b.gflop_per_sample(cfg, 256) - 509.8) < 1.0          # total forward, L = 256, 18 -> 30
    nolora = config.PRESETS["llama32_1b"](seq_len=18, out_len=30, use_lora=False)
    assert abs((b.gflop_per_sample(cfg, 256) - b.gflop_per_sample(nolora, 256)) - 0.44) < 0.02  # LoRA r=8 on q, v
    # dense part linear in L, attention quadratic
    g128, g512 = b.gflop_per_sample(cfg, 128), b.gflop_per_sample(cfg, 512)
    assert 0.49 < g128 / b.gflop_per_sample(cfg, 256) < 0.51 and 2.0 < g512 / b.gflop_per_sample(cfg, 256) < 2.03
