"""Generate tests/golden/*.npz by running the REFERENCE's own modules (build container only).

Run:  python tests/golden/make_golden.py        (needs /root/reference; not needed on the GPU box)

What is executed from the reference (imported by path, never copied):
  * scripts/ablation_study_without_lora.py : MultiModalTrajectoryModel and every class under it
    (LanePolygonEncoder, BlipQFormer, LlamaMultiModal, LlamaWithCrossAttnPEFT,
    SelfAttentionBlock, LTSF_*, TransformerLTSF) -- line-for-line the classes of
    scripts/train.py:352-964 without the `peft` import.
  * scripts/baseline_cv.py : ConstantVelocityPredictor, build_dataset_from_tracks_sliding,
    MultiModalTrajectoryDataset, custom_collate_fn.
Shims (not reference code): `AutoModelForCausalLM.from_pretrained` is rebound, inside the
imported module's namespace only, to build a local random `LlamaForCausalLM(LlamaConfig)`
of the requested shape (no network fetch); the tokenizer to a stub.  LoRA (`peft` is not
installed) is a 6-line module implementing y = W x + (alpha/r) B A x on q_proj / v_proj
(targets: modify_scripts/modify_train.py:518), eval mode (dropout off).

Weights come from tcavt_amd.weights.make_weights(cfg, seed) and are loaded into the
reference modules with load_state_dict(strict=True); the .npz files therefore hold only
inputs, expected outputs and the (preset, seed) needed to regenerate the weights.
"""
import importlib.util
import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tcavt_amd import config as tconfig  # noqa: E402
from tcavt_amd import synth  # noqa: E402
from tcavt_amd.weights import make_weights, is_lora_key, LLAMA_PREFIX  # noqa: E402

REF = "/root/reference/scripts"


def _import(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _llama_factory(ll):
    from transformers import LlamaConfig, LlamaForCausalLM

    class Factory:
        @staticmethod
        def from_pretrained(name, **kw):
            cfg = LlamaConfig(
                hidden_size=ll.hidden, intermediate_size=ll.inter, num_hidden_layers=ll.layers,
                num_attention_heads=ll.n_q_heads, num_key_value_heads=ll.n_kv_heads, head_dim=ll.head_dim,
                vocab_size=ll.vocab, max_position_embeddings=131072, rms_norm_eps=ll.rms_eps,
                rope_theta=ll.rope_theta, tie_word_embeddings=True, attention_bias=False, mlp_bias=False,
                rope_scaling={"factor": ll.rope_factor, "high_freq_factor": ll.rope_high_freq_factor,
                              "low_freq_factor": ll.rope_low_freq_factor,
                              "original_max_position_embeddings": ll.rope_original_max_pos, "rope_type": "llama3"})
            return LlamaForCausalLM(cfg)

    return Factory


class _Tok:
    pad_token = "<pad>"
    eos_token = "<eos>"

    @staticmethod
    def from_pretrained(name, **kw):
        return _Tok()


class LoRALinear(nn.Module):
    """y = W x + (alpha/r) * B(A(dropout(x))); PEFT's LoRA layer restated (every adapted Linear owns its own
    lora_dropout module, so q_proj and v_proj draw independent masks in train mode)."""

    def __init__(self, base, A, B, scale, p_drop=0.0):
        super().__init__()
        self.base, self.A, self.B, self.scale = base, nn.Parameter(A), nn.Parameter(B), scale
        self.lora_dropout = nn.Dropout(p_drop)

    def forward(self, x):
        return self.base(x) + self.scale * ((self.lora_dropout(x) @ self.A.T) @ self.B.T)


def build_reference_model(ref, cfg, weights):
    ref.AutoModelForCausalLM = _llama_factory(cfg.llama)
    ref.AutoTokenizer = _Tok
    model = ref.MultiModalTrajectoryModel(
        seq_len=cfg.seq_len, out_len=cfg.out_len, individual=cfg.individual, feature_size=cfg.feature_size,
        d_model=cfg.d_model, lane_polygon_d_model=cfg.lane_polygon_d_model,
        lane_polygon_nhead=cfg.lane_polygon_nhead, lane_polygon_layers=cfg.lane_polygon_layers,
        max_polygon_points=cfg.max_polygon_points, use_post_mlp=cfg.use_post_mlp,
        post_mlp_hidden_dim=cfg.post_mlp_hidden_dim, base_model_name="local-random-llama",
        vision_dim=cfg.vision_dim, q_hidden_size=cfg.q_hidden_size, q_nhead=cfg.q_nhead,
        q_enc_layers=cfg.q_enc_layers, q_dec_layers=cfg.q_dec_layers,
        q_num_query_tokens=cfg.q_num_query_tokens, ltsf_nhead=cfg.ltsf_nhead, ltsf_dropout=cfg.ltsf_dropout)
    sd = {k: torch.from_numpy(v) for k, v in weights.items() if not is_lora_key(k)}
    model.load_state_dict(sd, strict=True)
    if cfg.use_lora:
        scale = cfg.lora_alpha / cfg.lora_r
        layers = model.mllm.llama_wrapper.llama_model.model.layers
        for i, layer in enumerate(layers):
            for proj in ("q_proj", "v_proj"):
                key = f"{LLAMA_PREFIX}layers.{i}.self_attn.{proj}."
                A = torch.from_numpy(weights[key + "lora_A.weight"])
                B = torch.from_numpy(weights[key + "lora_B.weight"])
                setattr(layer.self_attn, proj, LoRALinear(getattr(layer.self_attn, proj), A, B, scale, cfg.lora_dropout))
    model.eval()
    return model


CASES = [
    # name, preset, T, To, lora, B, text_len, ragged, empty_polygon_every, seed
    ("tiny_6_12_lora_ragged", "tiny", 6, 12, True, 3, 20, True, 3, 11),
    ("tiny_18_30_nolora_ragged", "tiny", 18, 30, False, 2, 24, True, 0, 12),
    ("tiny_6_30_lora_full", "tiny", 6, 30, True, 2, 16, False, 0, 13),
]


def run_model_case(ref, name, preset, T, To, lora, B, text_len, ragged, empty_every, seed):
    cfg = tconfig.PRESETS[preset](seq_len=T, out_len=To, use_lora=lora)
    weights = make_weights(cfg, seed)
    model = build_reference_model(ref, cfg, weights)
    batch = synth.make_batch(cfg, B, text_len=text_len, seed=seed, ragged=ragged, min_text=4,
                             empty_polygon_every=empty_every)
    t = {k: torch.from_numpy(v) for k, v in batch.items()}
    ns = [tuple(float(v) for v in row) for row in batch["norm_stat"]]
    pl = [int(v) for v in batch["lane_polygon_len"]]
    with torch.no_grad():
        poly_emb = model.lane_polygon_encoder(t["lane_polygon"], pl)
        img_tokens = model.mllm.qformer(t["vision_emb"])
        final_hidden, n_img = model.mllm(t["vision_emb"], None, input_ids=t["input_ids"],
                                         attention_mask=t["attention_mask"], labels=t["labels"])
        loss, decoded = model(t["traj_emb"], t["vision_emb"], None, t["lane_polygon"], pl, y=t["target_traj"],
                              norm_stat=ns, input_ids=t["input_ids"], attention_mask=t["attention_mask"],
                              labels=t["labels"])
        decoded_only = model(t["traj_emb"], t["vision_emb"], None, t["lane_polygon"], pl,
                             input_ids=t["input_ids"], attention_mask=t["attention_mask"])
    assert torch.equal(decoded, decoded_only) and n_img == cfg.q_num_query_tokens
    out = dict(batch)
    out.update({
        "preset": np.array(preset), "seed": np.array(seed), "use_lora": np.array(lora),
        "seq_len": np.array(T), "out_len": np.array(To),
        "exp_poly_emb": poly_emb.numpy(), "exp_img_tokens": img_tokens.numpy(),
        "exp_final_hidden": final_hidden.numpy(), "exp_decoded": decoded.numpy(),
        "exp_loss": np.array(loss.item(), np.float64),
    })
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"[golden] {name}: loss={loss.item():.6f} decoded[0,:,0]={decoded[0, :, 0].tolist()}")


def _sample(t, cap=512):
    """Full tensor when it has <= cap elements, otherwise an evenly strided sample of <= cap elements (the fixture stays
    small; tests draw the same sample with tcavt_amd-independent code: flat[::stride], stride = ceil(n / cap))."""
    flat = t.detach().reshape(-1)
    stride = -(-flat.numel() // cap)
    return flat[::stride].to(torch.float32).numpy().copy()


def run_train_case(ref, name, preset, T, To, lora, B, text_len, ragged, empty_every, seed):
    """Gradient / optimizer / dropout-site fixture of one model case (tests/golden/<name>_train.npz):
      * loss.backward() on the reference model in eval arithmetic (dropout off, deterministic), gradients of the
        train.py trainable set (everything outside mllm, scripts/train.py:1140-1145) and -- for the LoRA-trainable
        loop of modify_scripts/modify_train.py:512-528 -- of the adapter matrices and of the MLLM's front end (Q-Former,
        mllm.q_proj, modality embeddings: "front");
      * one torch.optim.AdamW(lr=5e-4, weight_decay=1e-4) step on the train.py set (scripts/train.py:1145,1182-1183);
      * the sequence of dropout calls of ONE train-mode forward (ddp_model.train(), scripts/train.py:1152):
        (kind, p, shape) of every F.dropout (nn.Dropout, attention weights of nn.MultiheadAttention's explicit path)
        and every scaled_dot_product_attention with dropout_p > 0 (the need_weights=False path of
        nn.Transformer*Layer); masks themselves are torch's RNG stream and are not portable, the SITES are."""
    import torch.nn.functional as F

    cfg = tconfig.PRESETS[preset](seq_len=T, out_len=To, use_lora=lora)
    weights = make_weights(cfg, seed)
    model = build_reference_model(ref, cfg, weights)
    batch = synth.make_batch(cfg, B, text_len=text_len, seed=seed, ragged=ragged, min_text=4,
                             empty_polygon_every=empty_every)
    t = {k: torch.from_numpy(v) for k, v in batch.items()}
    ns = [tuple(float(v) for v in row) for row in batch["norm_stat"]]
    pl = [int(v) for v in batch["lane_polygon_len"]]

    def fwd():
        return model(t["traj_emb"], t["vision_emb"], None, t["lane_polygon"], pl, y=t["target_traj"], norm_stat=ns,
                     input_ids=t["input_ids"], attention_mask=t["attention_mask"], labels=t["labels"])

    # ---- (1) gradients, eval arithmetic
    named = dict(model.named_parameters())
    trainable = [k for k in named if not k.startswith("mllm.")]
    # the MLLM's front end: trainable in modify_scripts/modify_train.py (:523-528 freeze only the non-LoRA Llama weights)
    front = [k for k in named if k.startswith(("mllm.qformer.", "mllm.q_proj.")) or
             k in ("mllm.vision_modality_embedding", "mllm.text_modality_embedding")]
    for k, p in named.items():
        p.requires_grad_(k in trainable or k in front or k.endswith((".A", ".B")))
    model.eval()
    loss, _ = fwd()
    loss.backward()
    out = {"preset": np.array(preset), "seed": np.array(seed), "use_lora": np.array(lora), "seq_len": np.array(T),
           "out_len": np.array(To), "loss": np.array(loss.item(), np.float64),
           "trainable": np.array(trainable)}
    for k in trainable:
        g = named[k].grad
        out["grad." + k] = _sample(g)
        out["gnorm." + k] = np.array(g.double().norm().item())
    out["front"] = np.array(front)
    for k in front:
        g = named[k].grad
        out["grad." + k] = _sample(g)
        out["gnorm." + k] = np.array(g.double().norm().item())
    if lora:
        layers = model.mllm.llama_wrapper.llama_model.model.layers
        for i, layer in enumerate(layers):
            for proj in ("q_proj", "v_proj"):
                m = getattr(layer.self_attn, proj)
                key = f"{LLAMA_PREFIX}layers.{i}.self_attn.{proj}."
                out["grad." + key + "lora_A.weight"] = m.A.grad.numpy().copy()
                out["grad." + key + "lora_B.weight"] = m.B.grad.numpy().copy()
    # ---- (2) one AdamW step on the train.py set
    opt = torch.optim.AdamW([named[k] for k in trainable], lr=5e-4, weight_decay=1e-4)
    opt.step()
    for k in trainable:
        out["adamw." + k] = _sample(named[k])
    # ---- (3) dropout sites of a train-mode forward
    sites = []
    real_dropout, real_sdpa = F.dropout, F.scaled_dot_product_attention

    def rec_dropout(input, p=0.5, training=True, inplace=False):
        if training and p > 0.0:
            sites.append((0, float(p), tuple(input.shape)))
        return real_dropout(input, p, training, inplace)

    def rec_sdpa(query, key, value, attn_mask=None, dropout_p=0.0, *a, **kw):
        if dropout_p > 0.0:
            sites.append((1, float(dropout_p), tuple(query.shape[:-1]) + (key.shape[-2],)))  # shape of the weights
        return real_sdpa(query, key, value, attn_mask, dropout_p, *a, **kw)

    F.dropout, F.scaled_dot_product_attention = rec_dropout, rec_sdpa
    try:
        model.train()
        torch.manual_seed(0)
        with torch.no_grad():
            fwd()
    finally:
        F.dropout, F.scaled_dot_product_attention = real_dropout, real_sdpa
        model.eval()
    out["drop_kind"] = np.array([s_[0] for s_ in sites], np.int32)
    out["drop_p"] = np.array([s_[1] for s_ in sites], np.float64)
    out["drop_numel"] = np.array([int(np.prod(s_[2])) for s_ in sites], np.int64)
    shp = np.zeros((len(sites), 4), np.int64)
    for i, s_ in enumerate(sites):
        shp[i, : len(s_[2])] = s_[2]
    out["drop_shape"] = shp
    np.savez_compressed(os.path.join(HERE, name + "_train.npz"), **out)
    print(f"[golden] {name}_train: loss={loss.item():.6f} {len(trainable)} gradient tensors, {len(sites)} dropout sites")
    for s_ in sites:
        print("   ", s_)


def run_generation_case(ref):
    """tests/golden/tiny_generation.npz -- the text-generation row (scripts/train.py:577-654):
      * greedy continuation of the tiny LoRA case's prompts by the reference model's own HF LlamaForCausalLM
        (`logits[:, -1]` of a full forward on the growing sequence; prefix = image tokens + valid prompt tokens built by
        the reference's Q-Former / q_proj / embedding modules as train.py:519-528 does; a generated token is embedded
        as a text token), with the top-1 / top-2 margin of every step;
      * transformers' own logits processors in generate()'s order -- RepetitionPenalty(1.2), NoRepeatNGram(3),
        Temperature(0.9), TopK(40), TopP(0.9) (train.py:628-642) -- applied to fixed random scores and histories."""
    from transformers.generation.logits_process import (NoRepeatNGramLogitsProcessor, RepetitionPenaltyLogitsProcessor,
                                                        TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper)

    from tests.util import scale_generation_weights

    name, preset, T, To, lora, B, text_len, ragged, empty_every, seed = CASES[0]
    cfg = tconfig.PRESETS[preset](seq_len=T, out_len=To, use_lora=lora)
    # Weight variant + prompts chosen (by a search with the oracle over batch seeds 100..259, both greedy modes) so that the PLAIN arg-max
    # continuation is not one token repeated: >= 7 distinct tokens in each sample's 12, smallest top-1 / top-2 margin 0.32 in both greedy modes
    # (logits have a standard deviation of ~5; the fp16 path's logit error is ~1e-2).
    text_mod_scale, out_gain, batch_seed = 0.1, 12.0, 253
    weights = scale_generation_weights(make_weights(cfg, seed), text_mod_scale, out_gain)
    model = build_reference_model(ref, cfg, weights)
    batch = synth.make_batch(cfg, B, text_len=text_len, seed=batch_seed, ragged=ragged, min_text=4, empty_polygon_every=empty_every)
    vision, ids, mask = (torch.from_numpy(batch[k]) for k in ("vision_emb", "input_ids", "attention_mask"))
    mm = model.mllm
    llama = mm.llama_wrapper.llama_model
    emb = llama.get_input_embeddings()
    N = 12
    toks = np.zeros((2, B, N), np.int64)       # [plain arg-max | with the reference's repetition penalty + 3-gram ban]
    margins = np.zeros((2, B, N), np.float32)
    rep, ngram = RepetitionPenaltyLogitsProcessor(1.2), NoRepeatNGramLogitsProcessor(3)
    with torch.no_grad():
        img = mm.q_proj(mm.qformer(vision)) + mm.vision_modality_embedding
        for mode in (0, 1):
            for b in range(B):
                n = int(mask[b].sum())
                seq = torch.cat([img[b:b + 1], emb(ids[b:b + 1, :n]) + mm.text_modality_embedding], dim=1)
                hist_ids = ids[b:b + 1, :n].clone()  # generate(input_ids=prompt_ids, ...): the processors see prompt + generated
                for i in range(N):
                    lg = llama(inputs_embeds=seq, attention_mask=torch.ones(1, seq.shape[1], dtype=torch.long)).logits[:, -1]
                    if mode == 1:
                        lg = ngram(hist_ids, rep(hist_ids, lg))
                    top = torch.topk(lg[0], 2)
                    toks[mode, b, i], margins[mode, b, i] = int(top.indices[0]), float(top.values[0] - top.values[1])
                    hist_ids = torch.cat([hist_ids, top.indices[:1][None]], dim=1)
                    seq = torch.cat([seq, emb(top.indices[:1])[None] + mm.text_modality_embedding], dim=1)
    # ---- processors on fixed scores
    g = torch.Generator().manual_seed(77)
    V, R = cfg.llama.vocab, 6
    scores = torch.randn(R, V, generator=g) * 3.0
    lens = [0, 1, 2, 5, 30, 60]
    hist = np.full((R, 64), -1, np.int64)
    processed, warped = np.zeros((R, V), np.float32), np.zeros((R, V), np.float32)
    for r_, n in enumerate(lens):
        h = torch.randint(0, 40, (n,), generator=g)  # a small alphabet: repeated tokens and repeated bigrams occur
        hist[r_, :n] = h.numpy()
        x = scores[r_:r_ + 1].clone()
        ids_row = h[None].to(torch.long)
        if n > 0:
            x = RepetitionPenaltyLogitsProcessor(1.2)(ids_row, x)
        x = NoRepeatNGramLogitsProcessor(3)(ids_row, x)
        processed[r_] = x[0].numpy()
        x = TemperatureLogitsWarper(0.9)(ids_row, x)
        x = TopKLogitsWarper(40)(ids_row, x)
        x = TopPLogitsWarper(0.9)(ids_row, x)
        warped[r_] = x[0].numpy()
    np.savez_compressed(os.path.join(HERE, "tiny_generation.npz"), case=np.array(name), preset=np.array(preset), seed=np.array(seed),
                        use_lora=np.array(lora), seq_len=np.array(T), out_len=np.array(To),
                        gen_text_mod_scale=np.array(text_mod_scale), gen_out_gain=np.array(out_gain),
                        gen_batch_seed=np.array(batch_seed), vision_emb=batch["vision_emb"], input_ids=batch["input_ids"],
                        attention_mask=batch["attention_mask"], greedy_tokens=toks[0],
                        greedy_margins=margins[0], greedy_proc_tokens=toks[1], greedy_proc_margins=margins[1],
                        scores=scores.numpy(), hist=hist, hist_len=np.array(lens, np.int32),
                        processed=processed, warped=warped)
    print(f"[golden] tiny_generation: greedy tokens {toks[0].tolist()} / with processors {toks[1].tolist()}, min margin "
          f"{margins[0].min():.3e} / {margins[1].min():.3e}; "
          f"kept candidates per row {[int(np.isfinite(w).sum()) for w in warped]}")


def builder_tracks():
    """The 64 synthetic tracks of config 1 in the form the reference's `train.py` builder expects them (vision features as
    torch tensors, train.py:187-193), with the edge cases its branches need: a track whose vision features end before its
    trajectory does (zero-padded and empty vision windows, :189-192), a track without vision features (:194-195), a track
    that names an A4 line (dropped by filter_context, :50), one without a lane (dropped, :142-144), one that jumps (dropped
    by is_trajectory_abnormal), one too short for a window (:152-153)."""
    tracks = synth.make_tracks(n_tracks=64, n_frames=400, seed=0)
    for t in tracks:
        t["vision_embeddings"] = torch.from_numpy(t["vision_embeddings"])
    tracks[3]["vision_embeddings"] = tracks[3]["vision_embeddings"][:150]   # 30 frames after [::5]
    del tracks[5]["vision_embeddings"]
    tracks[7]["context_str"] += "\nA4: a line that excludes the track."
    tracks[9]["context_str"] = "A1: no lane is named here."
    tracks[11]["raw_trajectory"] = tracks[11]["raw_trajectory"].copy()
    tracks[11]["raw_trajectory"][200:, 0] -= 400.0
    tracks[13]["raw_trajectory"] = tracks[13]["raw_trajectory"][:200]
    tracks[13]["vision_embeddings"] = tracks[13]["vision_embeddings"][:200]
    return tracks


def sanity_tracks():
    """Inputs for check_data_sanity: NaN, Inf, a coordinate beyond 1e6, a missing trajectory, a list instead of an array."""
    tracks = synth.make_tracks(n_tracks=12, n_frames=40, seed=3)
    for t in tracks:
        del t["vision_embeddings"]
    tracks[1]["raw_trajectory"] = tracks[1]["raw_trajectory"].copy()
    tracks[1]["raw_trajectory"][5, 1] = np.nan
    tracks[4]["raw_trajectory"] = tracks[4]["raw_trajectory"].copy()
    tracks[4]["raw_trajectory"][0, 0] = np.inf
    tracks[6]["raw_trajectory"] = tracks[6]["raw_trajectory"] * 1e4
    del tracks[8]["raw_trajectory"]
    tracks[10]["raw_trajectory"] = tracks[10]["raw_trajectory"].tolist()
    tracks[11]["raw_trajectory"] = tracks[11]["raw_trajectory"].astype(np.float64)
    tracks[11]["raw_trajectory"][3, 0] = 1e6 + 0.01   # rounds to 1e6 in float32: kept (the reference compares float32 values)
    return tracks


def _reference_function(path, name, namespace):
    """One top-level function of a reference script that cannot be imported whole (modify_train.py needs `peft`): its
    definition is compiled from the file where it lies and executed in `namespace`; nothing is copied."""
    import ast

    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), namespace)
    return namespace[name]


def run_builder_case(ref):
    """tests/golden/builder_64tracks.npz -- the `train.py` leg of the dataset builder (scripts/train.py:114-259,
    264-347; here the byte-identical copy in scripts/ablation_study_without_lora.py:112-330): vision windows with
    zero padding, prompt / answer tokenisation, labels = -100 on the prompt, truncation to max_length, collate.
    The tokenizer is tcavt_amd.synth.SyntheticTokenizer (the real one is a network fetch).  Three max_length values:
    512 (nothing truncated), 120 (prompt + answer cut), 40 (each segment cut, then the total).  Also check_data_sanity
    (modify_scripts/modify_train.py:26-49) on tracks with NaN / Inf / extreme / missing trajectories, and the wall time
    of the reference's builder on these tracks."""
    import time

    tok = synth.SyntheticTokenizer()
    T, To = 18, 30
    out = {}
    ref_time = None
    for ml in (512, 120, 40):
        tracks = builder_tracks()
        t0 = time.perf_counter()
        ins, outs = ref.build_dataset_from_tracks_sliding(tracks, seq_len=T, out_len=To, stride=6, max_step=50.0,
                                                          max_speed_diff=30.0, image_width=3840, image_height=2160,
                                                          downsample=5, tokenizer=tok, max_length=ml)
        dt = time.perf_counter() - t0
        if ml == 512:
            ref_time = dt
        n = len(ins)
        lens = np.array([len(s["input_ids"]) for s in ins], np.int64)
        width = int(lens.max())
        ids = np.zeros((n, width), np.int64)
        msk = np.zeros((n, width), np.int64)
        lab = np.full((n, width), -100, np.int64)
        for k, s in enumerate(ins):
            ids[k, :lens[k]], msk[k, :lens[k]], lab[k, :lens[k]] = s["input_ids"], s["attention_mask"], s["labels"]
        out[f"ml{ml}_len"], out[f"ml{ml}_ids"], out[f"ml{ml}_mask"], out[f"ml{ml}_labels"] = lens, ids, msk, lab
        if ml != 512:
            continue
        out["n_windows"] = np.array(n)
        out["track_id"] = np.array([s["track_id"] for s in ins])
        out["norm_stat"] = np.array([s["norm_stat"] for s in ins], np.float64)
        out["traj_in"] = np.stack([s["trajectory_embeddings"].numpy() for s in ins])
        out["traj_out"] = np.stack([o.numpy() for o in outs])
        out["vision_shape"] = np.array([tuple(s["vision_embeddings"].shape) for s in ins], np.int64)
        # vision windows: float64 sum and three probes per window for all of them, whole windows for a few
        out["vision_sum"] = np.array([float(s["vision_embeddings"].double().sum()) for s in ins])
        out["vision_probe"] = np.stack([s["vision_embeddings"].reshape(-1)[[0, 777 % s["vision_embeddings"].numel(), -1]].numpy()
                                        for s in ins])
        tids = list(out["track_id"])
        full_idx = sorted({0, n - 1, tids.index("syn0003"), len(tids) - 1 - tids[::-1].index("syn0003")})
        out["vision_full_idx"] = np.array(full_idx)
        out["vision_full"] = np.stack([ins[i]["vision_embeddings"].numpy() for i in full_idx])
        # dataset + collate on a ragged selection (different polygon lengths and text lengths)
        ds = ref.MultiModalTrajectoryDataset(ins, outs, max_polygon_points=16)   # 16 < every polygon: the truncation branch
        sel = list(range(0, n, max(1, n // 8)))[:8]
        coll = ref.custom_collate_fn([ds[i] for i in sel])
        out["coll_idx"] = np.array(sel)
        for k in ("traj_emb", "target_traj", "lane_polygon", "input_ids", "attention_mask", "labels"):
            out["coll_" + k] = coll[k].numpy()
        out["coll_lane_polygon_len"] = np.array(coll["lane_polygon_len"])
        out["coll_vision_sum"] = np.array(float(coll["vision_emb"].double().sum()))
    # ---- check_data_sanity of the LoRA-trainable script
    cds = _reference_function("/root/reference/modify_scripts/modify_train.py", "check_data_sanity", {"np": np})
    st = sanity_tracks()
    with redirect_stdout(io.StringIO()) as buf:
        kept = cds(st, max_coord_threshold=1e6)
    out["sanity_kept"] = np.array([next(i for i, t in enumerate(st) if t is d) for d in kept])
    out["sanity_printed"] = np.array(buf.getvalue().strip())
    out["ref_builder_seconds"] = np.array(ref_time)
    np.savez_compressed(os.path.join(HERE, "builder_64tracks.npz"), **out)
    print(f"[golden] builder_64tracks: windows={int(out['n_windows'])} text lengths {sorted(set(out['ml512_len'].tolist()))} / "
          f"{sorted(set(out['ml120_len'].tolist()))} / {sorted(set(out['ml40_len'].tolist()))}; sanity kept "
          f"{out['sanity_kept'].tolist()} ({out['sanity_printed']}); reference builder {ref_time:.3f} s")



def run_cv_case():
    """Config 1 (baseline_cv.py): dataset builder + collate + CV predictor + evaluate_cv's printed
    minADE/minFDE/minRMSE on 64 synthetic tracks (pickle written to a temp dir)."""
    import pickle
    import random
    import re
    import tempfile

    cv = _import(os.path.join(REF, "baseline_cv.py"), "ref_baseline_cv")
    tracks = synth.make_tracks(n_tracks=64, n_frames=400, seed=0)
    ins, outs = cv.build_dataset_from_tracks_sliding(tracks, seq_len=6, out_len=30, stride=6, max_step=50.0,
                                                     max_speed_diff=30.0, image_width=3840, image_height=2160,
                                                     downsample=5)
    ds = cv.TrajectoryDataset(ins, outs)
    idx = list(range(0, len(ds), max(1, len(ds) // 16)))[:16]
    coll = cv.custom_collate_fn([ds[i] for i in idx])
    model = cv.ConstantVelocityPredictor(6, 30)
    torch.manual_seed(0)
    with torch.no_grad():
        pred = model(coll["traj_emb"], num_candidates=10, noise_scale=0.1)  # (B,K,To,2)
    torch.manual_seed(0)
    noise = torch.stack([torch.randn(len(idx), 2) for _ in range(10)], dim=1)  # the draws it consumed

    # full evaluate_cv run (baseline_cv.py:280-360) on the same tracks; it only prints its result
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        pkl = os.path.join(tmp, "all_data.pkl")
        with open(pkl, "wb") as f:
            pickle.dump(tracks, f)
        os.chdir(tmp)
        try:
            random.seed(0)
            torch.manual_seed(0)
            buf = io.StringIO()
            with redirect_stdout(buf):
                cv.evaluate_cv({"all_data_pkl": pkl, "seq_len": 6, "out_len": 30, "batch_size": 16, "stride": 6,
                                "downsample": 5, "max_step": 50.0, "max_speed_diff": 30.0, "image_width": 3840,
                                "image_height": 2160})
        finally:
            os.chdir(cwd)
    m = re.search(r"minADE=([0-9.]+), minFDE=([0-9.]+), minRMSE=([0-9.]+)", buf.getvalue())
    printed = np.array([float(m.group(i)) for i in (1, 2, 3)])
    np.savez_compressed(
        os.path.join(HERE, "cv_64tracks.npz"),
        n_windows=np.array(len(ins)), sample_idx=np.array(idx),
        traj_emb=coll["traj_emb"].numpy(), target_traj=coll["target_traj"].numpy(),
        norm_stat=np.array(coll["norm_stat"], np.float64), cv_pred=pred.numpy(), cv_noise=noise.numpy(),
        all_norm_stat=np.array([s["norm_stat"] for s in ins], np.float64),
        all_track_id=np.array([s["track_id"] for s in ins]),
        all_poly_len=np.array([len(s["lane_polygon"]) for s in ins]),
        evaluate_cv_printed=printed,
    )
    print(f"[golden] cv_64tracks: windows={len(ins)} pred{tuple(pred.shape)} evaluate_cv={printed.tolist()}")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    with redirect_stdout(io.StringIO()):
        ref = _import(os.path.join(REF, "ablation_study_without_lora.py"), "ref_nolora")
    for case in CASES:
        run_model_case(ref, *case)
    for case in CASES[:2]:  # the ragged LoRA case and the ragged no-LoRA (train.py) case
        run_train_case(ref, *case)
    run_generation_case(ref)
    run_builder_case(ref)
    run_cv_case()


if __name__ == "__main__":
    main()
