"""fp16 RANGE on outlier activations (precision contract).

The reference runs a real Llama checkpoint in fp32 (scripts/train.py:427-431); real checkpoints carry a few hidden channels
whose activations sit orders of magnitude above the bulk of the residual stream.  The synthetic N(0, sigma) weights of the other
tests never do, so the range argument of the fp16 contract is exercised here on purpose with tcavt_amd.weights.plant_outliers:

  case 1  |stream| ~ 3e4  (inside fp16)            -> the plain 16-bit stream meets the 1e-3 bar, no flag
  case 2  |stream| ~ 2e5  (beyond fp16's 65504)    -> the plain contract raises the range flag (nothing is clamped silently);
          set_storage("auto") re-runs the flagged first pass down its ladder and keeps the first clean contract -- the
          16-bit stream held at 2^-k (tcavt_llama_stack_args.stream_scale) -- and the bar is met again; the fp32-stream
          contract (wide_stream: h != NULL, fp16 operands) needs the same scale for its 16-bit COPY and is checked as well.
All comparisons are against the oracle's fp32 graph (the reference's arithmetic)."""
import numpy as np
import pytest
import torch

from tests.util import batch_tensors, load_case, load_generation_case, rel_err

pytestmark = pytest.mark.gpu

CASE = "tiny_6_12_lora_ragged"


def _setup(dev, magnitude, case=CASE):
    from oracle import forward as O
    from tcavt_amd import model
    from tcavt_amd.weights import plant_outliers

    cfg, weights, fx = load_case(case)
    if magnitude:
        weights = plant_outliers(weights, cfg, magnitude)
    t = batch_tensors(fx)
    with torch.no_grad():
        ex = {}
        _, dec32 = O.model_forward(weights, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"], t["lane_polygon_len"],
                                   t["input_ids"], t["attention_mask"], y=t["target_traj"], norm_stat=t["norm_stat"],
                                   contract="fp32", extras=ex)
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
    g = {k: v.to(dev) for k, v in t.items()}
    return cfg, m, g, dec32, ex["final_hidden"]


def _fwd(m, g):
    with torch.no_grad():
        loss, dec = m(g["traj_emb"], g["vision_emb"], None, g["lane_polygon"], g["lane_polygon_len"], y=g["target_traj"],
                      norm_stat=g["norm_stat"], input_ids=g["input_ids"], attention_mask=g["attention_mask"], labels=g["labels"])
    torch.cuda.synchronize()
    return loss, dec


def test_outliers_inside_fp16_meet_the_bar_on_the_plain_stream(gpu):
    cfg, m, g, dec32, fh32 = _setup(gpu["device"], 3.0e4)
    _, dec = _fwd(m, g)
    m.mllm.check_flags()  # no range flag
    lw = m.mllm.llama_wrapper
    M = g["input_ids"].shape[0] * (cfg.q_num_query_tokens + g["input_ids"].shape[1])
    top = lw.norm_inputs(M, gpu["device"])[0].float().abs().max().item()
    e_dec, e_fh = rel_err(dec.cpu(), dec32), rel_err(m.last.final_hidden.float().cpu(), fh32)
    print(f"[range case 1] max |16-bit stream| {top:.3g}; decoded vs fp32 {e_dec:.2e}, final_hidden {e_fh:.2e}")
    assert 2.0e4 < top < 65504
    assert e_dec < 1e-3 and e_fh < 2e-3


def test_outliers_beyond_fp16_flag_the_plain_stream_and_auto_picks_a_contract_that_survives(gpu):
    cfg, m, g, dec32, fh32 = _setup(gpu["device"], 2.0e5)
    # (1) the plain contract: non-finite result AND the flag names where it started
    _, dec = _fwd(m, g)
    assert not torch.isfinite(dec).all()
    with pytest.raises(FloatingPointError) as ei:
        m.mllm.check_flags()
    assert "layer 0" in str(ei.value)
    # (2) auto: calibrated on the first forward, kept afterwards
    m.set_storage("auto")
    _, dec = _fwd(m, g)
    rc = m.range_contract
    print(f"[range case 2] auto trials (scale, fp32 stream, flag): {rc.trials}")
    assert rc.clean and rc.trials[0][2] != 0          # the plain pass was flagged ...
    assert rc.stream_scale == 2.0 ** -2 and not rc.wide_stream  # ... 2e5 / 4 fits: the smallest scale on the ladder that does
    m.mllm.check_flags()
    e_dec, e_fh = rel_err(dec.cpu(), dec32), rel_err(m.last.final_hidden.float().cpu(), fh32)
    print(f"[range case 2] 16-bit stream at 2^-2: decoded vs fp32 {e_dec:.2e}, final_hidden {e_fh:.2e}")
    assert e_dec < 1e-3 and e_fh < 2e-3
    _, dec2 = _fwd(m, g)  # no re-calibration, same result
    assert m.range_contract is rc and torch.equal(dec, dec2)
    # (3) the fp32-stream contract: its 16-bit copy overflows just the same without the scale, and passes with it
    m.set_storage(torch.float16, wide_stream=True)
    _fwd(m, g)
    with pytest.raises(FloatingPointError):
        m.mllm.check_flags()
    m.set_storage(torch.float16, stream_scale=2.0 ** -4, wide_stream=True)
    _, dec = _fwd(m, g)
    m.mllm.check_flags()
    e_dec, e_fh = rel_err(dec.cpu(), dec32), rel_err(m.last.final_hidden.float().cpu(), fh32)
    print(f"[range case 2] fp32 stream, 16-bit copy at 2^-4: decoded vs fp32 {e_dec:.2e}, final_hidden {e_fh:.2e}")
    assert e_dec < 1e-3 and e_fh < 2e-3


@pytest.mark.parametrize("case", ["tiny_6_12_lora_ragged", "tiny_18_30_nolora_ragged"])
@pytest.mark.parametrize("scale", [2.0 ** -2, 2.0 ** -8])
def test_a_scaled_stream_changes_nothing_on_a_model_that_needs_no_scale(gpu, case, scale):
    """Power-of-two scale: the 16-bit image holds the same significands (until values go subnormal), so the clean model's
    result moves by rounding noise only -- and auto keeps the plain contract for it."""
    cfg, m, g, dec32, fh32 = _setup(gpu["device"], 0.0, case)
    _, dec_plain = _fwd(m, g)
    m.set_storage(torch.float16, stream_scale=scale)
    _, dec_s = _fwd(m, g)
    m.mllm.check_flags()
    e = rel_err(dec_s.cpu(), dec_plain.cpu())
    print(f"[range] {case} stream at {scale:g}: decoded vs the plain stream {e:.2e}, vs fp32 {rel_err(dec_s.cpu(), dec32):.2e}")
    assert e < 3e-4 and rel_err(dec_s.cpu(), dec32) < 1e-3
    m.set_storage("auto")
    _fwd(m, g)
    assert m.range_contract.clean and m.range_contract.stream_scale == 1.0 and len(m.range_contract.trials) == 1


def test_generation_runs_on_the_scaled_stream(gpu):
    """Prefill + decode steps at stream_scale 2^-4: the greedy continuation of the reference-model fixture is unchanged
    (tiny_generation.npz: margins >= 0.32, far above what a re-scaled rounding can move)."""
    from tcavt_amd import model

    fx, cfg, w, t = load_generation_case()
    dev = gpu["device"]
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(w, device=dev).eval()
    m.set_storage(torch.float16, stream_scale=2.0 ** -4)
    n_new = int(fx["greedy_tokens"].shape[1])
    out = m.mllm.generate_batch(t["vision_emb"].to(dev), None, max_new_tokens=n_new, input_ids=t["input_ids"].to(dev),
                                attention_mask=t["attention_mask"].to(dev), do_sample=False, repetition_penalty=1.0,
                                no_repeat_ngram_size=0)
    torch.cuda.synchronize()
    m.mllm.check_flags()
    assert np.array_equal(out.cpu().numpy(), fx["greedy_tokens"])


def test_norm_out_epilogues_at_a_scale(gpu):
    """Kernel level (tcavt_gemm_args.norm_scale), both stream forms, tiled and skinny kernels: the 16-bit image is
    round(s * (acc + old)), the partial sums are those of the image, an fp32 stream stays unscaled."""
    import ctypes

    from tcavt_amd import capi, ops

    dev = gpu["device"]
    gen = torch.Generator(device="cpu").manual_seed(11)
    s = 2.0 ** -5
    for M, N, K, tile in ((512, 256, 256, 0), (8, 256, 256, 0), (512, 512, 256, 257)):  # 8-wave tiles, skinny form, 4-wave 256 x 256
        a = (torch.randn(M, K, generator=gen) * 0.5).to(dev).half()
        w = (torch.randn(N, K, generator=gen) * 0.1).to(dev).half()
        old = (torch.randn(M, N, generator=gen) * 3.0).to(dev)
        acc = a.float() @ w.float().t()
        npart = ops.norm_npart(M, N, K)
        # (a) 16-bit stream, in place: old image holds s * x
        h16 = (old * s).half()
        exp16 = (acc * s + h16.float()).half()
        part = torch.zeros(M, npart, device=dev)
        g = capi.GemmArgs()
        g.A, g.lda, g.W, g.ldw = a.data_ptr(), K, w.data_ptr(), K
        g.C, g.ldc, g.M, g.N, g.K = None, N, M, N, K
        g.out_dtype, g.in_dtype, g.tile = capi.F32, capi.F16, tile
        g.epilogue = capi.EPI_RESIDUAL | capi.EPI_NORM_OUT
        g.norm_h16, g.norm_part, g.norm_scale = h16.data_ptr(), part.data_ptr(), s
        capi.check(capi.lib().tcavt_gemm_bf16(ctypes.byref(g), capi.stream_ptr()), "gemm")
        torch.cuda.synchronize()
        d = (h16.float() - exp16.float()).abs().max().item()
        assert d <= 2.0 ** -10 * exp16.float().abs().max().item(), (M, d)   # (fp32 summation order: at most one ulp of the image)
        ss = part.sum(1)
        assert rel_err(ss.cpu(), (h16.float() ** 2).sum(1).cpu()) < 1e-5
        # (b) fp32 stream: C unscaled, copy scaled
        c = torch.zeros(M, N, device=dev)
        h16b = torch.zeros(M, N, device=dev, dtype=torch.float16)
        part.zero_()
        g.C, g.residual, g.ldr = c.data_ptr(), old.data_ptr(), N
        g.norm_h16 = h16b.data_ptr()
        capi.check(capi.lib().tcavt_gemm_bf16(ctypes.byref(g), capi.stream_ptr()), "gemm")
        torch.cuda.synchronize()
        assert rel_err(c.cpu(), (acc + old).cpu()) < 1e-5
        assert torch.equal(h16b, (c * s).half())
        assert rel_err(part.sum(1).cpu(), ((c * s) ** 2).sum(1).cpu()) < 1e-5
