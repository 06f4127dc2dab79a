"""The `train.py` leg of the dataset builder: vision windows + zero padding, prompt / answer tokenisation,
labels = -100 on the prompt, truncation to max_length, dataset + collate -- against tests/golden/builder_64tracks.npz,
which the REFERENCE's own builder produced (scripts/ablation_study_without_lora.py:112-330 = scripts/train.py:114-347) on
the same 64 synthetic tracks with tcavt_amd.synth.SyntheticTokenizer.  Integers bit-exact, floats bit-exact.
Also check_data_sanity (modify_scripts/modify_train.py:26-49) and generate_batch's marker cut-off (train.py:645-653)."""
import importlib.util
import io
import os
import time
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden", "builder_64tracks.npz")


def _mg():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)   # (imports nothing of the reference: that happens inside its run_* functions)
    return mod


def _build(max_length):
    from tcavt_amd import data, synth

    tracks = _mg().builder_tracks()
    return data.build_dataset_from_tracks_sliding(tracks, seq_len=18, out_len=30, stride=6, max_step=50.0, max_speed_diff=30.0,
                                                  image_width=3840, image_height=2160, downsample=5,
                                                  tokenizer=synth.SyntheticTokenizer(), max_length=max_length)


@pytest.mark.parametrize("ml", [512, 120, 40])
def test_tokenised_windows_match_reference_builder(ml):
    fx = np.load(GOLDEN)
    ins, outs = _build(ml)
    assert len(ins) == int(fx["n_windows"]) == len(outs)
    lens = fx[f"ml{ml}_len"]
    for k, s in enumerate(ins):
        n = int(lens[k])
        assert s["input_ids"].dtype == torch.long and s["input_ids"].shape == (n,)
        assert np.array_equal(s["input_ids"].numpy(), fx[f"ml{ml}_ids"][k, :n])
        assert np.array_equal(s["attention_mask"].numpy(), fx[f"ml{ml}_mask"][k, :n])
        assert np.array_equal(s["labels"].numpy(), fx[f"ml{ml}_labels"][k, :n])
    if ml == 40:   # each segment cut to 40, then the total: only prompt tokens survive, every label is -100
        assert all(bool((s["labels"] == -100).all()) for s in ins)
    if ml == 120:  # the answer is cut: some labels survive behind the 91-token prompt
        assert all(int((s["labels"] != -100).sum()) == 120 - 91 for s in ins)


def test_windows_vision_and_norm_stats_match_reference_builder():
    fx = np.load(GOLDEN)
    ins, outs = _build(512)
    assert [s["track_id"] for s in ins] == list(fx["track_id"])
    assert np.array_equal(np.array([s["norm_stat"] for s in ins], np.float64), fx["norm_stat"])
    assert np.array_equal(np.stack([s["trajectory_embeddings"].numpy() for s in ins]), fx["traj_in"])
    assert np.array_equal(np.stack([o.numpy() for o in outs]), fx["traj_out"])
    assert np.array_equal(np.array([tuple(s["vision_embeddings"].shape) for s in ins], np.int64), fx["vision_shape"])
    assert all(s["vision_embeddings"].dtype == torch.float32 for s in ins)
    assert np.array_equal(np.array([float(s["vision_embeddings"].double().sum()) for s in ins]), fx["vision_sum"])
    probe = np.stack([s["vision_embeddings"].reshape(-1)[[0, 777 % s["vision_embeddings"].numel(), -1]].numpy() for s in ins])
    assert np.array_equal(probe, fx["vision_probe"])
    for i, want in zip(fx["vision_full_idx"], fx["vision_full"]):
        assert np.array_equal(ins[int(i)]["vision_embeddings"].numpy(), want)
    # the track whose vision features end early: its last windows are zero-padded / all zero (train.py:189-192)
    last = [s for s in ins if s["track_id"] == "syn0003"][-1]["vision_embeddings"]
    assert float(last.abs().sum()) == 0.0
    # the track without vision features (train.py:194-195)
    assert all(tuple(s["vision_embeddings"].shape) == (18, 1) for s in ins if s["track_id"] == "syn0005")
    # dropped tracks (A4 line, no lane, a jump, too short)
    ids = {s["track_id"] for s in ins}
    assert not ids & {"syn0007", "syn0009", "syn0011", "syn0013"}


def test_dataset_and_collate_match_reference():
    from tcavt_amd import data

    fx = np.load(GOLDEN)
    ins, outs = _build(512)
    # (the selection mixes 512-d and 1-d vision windows only if the reference's torch.stack accepted it: it holds tracks
    #  with features only, as make_golden's selection does)
    ds = data.MultiModalTrajectoryDataset(ins, outs, max_polygon_points=16)
    batch = data.custom_collate_fn([ds[int(i)] for i in fx["coll_idx"]])
    for k in ("traj_emb", "target_traj", "lane_polygon", "input_ids", "attention_mask", "labels"):
        assert np.array_equal(batch[k].numpy(), fx["coll_" + k]), k
    assert batch["lane_polygon_len"] == fx["coll_lane_polygon_len"].tolist() == [16] * len(fx["coll_idx"])
    assert float(batch["vision_emb"].double().sum()) == float(fx["coll_vision_sum"])
    assert batch["input_ids"].dtype == torch.long and batch["labels"].dtype == torch.long


def test_check_data_sanity_matches_reference():
    from tcavt_amd import data

    fx = np.load(GOLDEN)
    st = _mg().sanity_tracks()
    buf = io.StringIO()
    with redirect_stdout(buf):
        kept = data.check_data_sanity(st, max_coord_threshold=1e6)
    assert [next(i for i, t in enumerate(st) if t is d) for d in kept] == fx["sanity_kept"].tolist()
    assert buf.getvalue().strip() == str(fx["sanity_printed"])
    assert data.check_data_sanity([], verbose=False) == []


def test_generate_batch_marker_cutoff():
    """train.py:645-653: text after the first 'No right-following vehicle.' is dropped."""
    from tcavt_amd.model import cut_generated_text, GENERATION_CUTOFF_MARKER

    m = GENERATION_CUTOFF_MARKER
    assert cut_generated_text(f"a b. {m} trailing junk {m} more") == f"a b. {m}"
    assert cut_generated_text("nothing to cut") == "nothing to cut"
    assert cut_generated_text(m) == m and cut_generated_text("") == ""


def test_builder_is_not_slower_than_the_reference_builder():
    """The vectorised builder (windows of a track at once, tokenisation once per track) against the reference builder's time
    on the same tracks, recorded in the fixture by make_golden.py on this container class."""
    fx = np.load(GOLDEN)
    _build(512)
    t0 = time.perf_counter()
    _build(512)
    dt = time.perf_counter() - t0
    assert dt < 3.0 * float(fx["ref_builder_seconds"]) + 0.5   # generous: CI machines differ; the measured ratio is in DESIGN
