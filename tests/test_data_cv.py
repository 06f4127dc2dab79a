"""BASELINE.json configs[0]: baseline_cv.py plumbing on 64 synthetic tracks, CPU only.

The fixture tests/golden/cv_64tracks.npz was produced by the reference's own builder / predictor /
evaluate_cv on tcavt_amd.synth.make_tracks(n_tracks=64, n_frames=400, seed=0).  This repo's restatement
(tcavt_amd.data, tcavt_amd.evaluate) must reproduce it: window count and every window's norm_stat exactly,
collated tensors bit-exact, the CV predictions bit-exact under the same seed, evaluate_cv's printed
minADE/minFDE/minRMSE to the 4 printed decimals."""
import os
import random

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cv_64tracks.npz")


def _build():
    from tcavt_amd import data, synth

    tracks = synth.make_tracks(n_tracks=64, n_frames=400, seed=0)
    ins, outs = data.build_dataset_from_tracks_sliding(tracks, seq_len=6, out_len=30, stride=6, max_step=50.0,
                                                       max_speed_diff=30.0, image_width=3840, image_height=2160,
                                                       downsample=5)
    return tracks, ins, outs


def test_builder_windows_and_norm_stats_match_reference():
    fx = np.load(GOLDEN)
    _, ins, outs = _build()
    assert len(ins) == int(fx["n_windows"]) == len(outs)
    got = np.array([s["norm_stat"] for s in ins], np.float64)
    assert np.array_equal(got, fx["all_norm_stat"])
    assert [s["track_id"] for s in ins] == list(fx["all_track_id"])
    assert [len(s["lane_polygon"]) for s in ins] == list(fx["all_poly_len"])


def test_collate_and_cv_predictor_bit_exact():
    from tcavt_amd import data, evaluate

    fx = np.load(GOLDEN)
    _, ins, outs = _build()
    ds = data.MultiModalTrajectoryDataset(ins, outs, max_polygon_points=64)
    batch = data.custom_collate_fn([ds[int(i)] for i in fx["sample_idx"]])
    assert np.array_equal(batch["traj_emb"].numpy(), fx["traj_emb"])
    assert np.array_equal(batch["target_traj"].numpy(), fx["target_traj"])
    assert np.array_equal(np.array(batch["norm_stat"], np.float64), fx["norm_stat"])
    assert batch["lane_polygon"].shape == (len(fx["sample_idx"]), 64, 2)
    assert batch["input_ids"].shape == (len(fx["sample_idx"]), 1)  # no tokenizer -> dummy ids (train.py:239-242)
    model = evaluate.ConstantVelocityPredictor(6, 30)
    torch.manual_seed(0)
    pred = model(batch["traj_emb"], num_candidates=10, noise_scale=0.1)
    assert pred.shape == (len(fx["sample_idx"]), 10, 30, 2)
    assert np.array_equal(pred.numpy(), fx["cv_pred"])


def test_evaluate_cv_reproduces_printed_metrics():
    from tcavt_amd import evaluate, synth

    fx = np.load(GOLDEN)
    tracks = synth.make_tracks(n_tracks=64, n_frames=400, seed=0)
    random.seed(0)
    torch.manual_seed(0)
    ade, fde, rmse, n = evaluate.evaluate_cv(tracks, seq_len=6, out_len=30, batch_size=16, stride=6, downsample=5)
    printed = fx["evaluate_cv_printed"]
    assert n > 0
    assert [round(ade, 4), round(fde, 4), round(rmse, 4)] == [float(v) for v in printed]


def test_filters():
    from tcavt_amd import data

    assert data.filter_context("A4: something") == (None, None)
    assert data.filter_context("   ") == ("No context provided", "R2L")
    assert data.filter_context("hello") == ("No valid context lines", "R2L")
    txt, d = data.filter_context("A1: moving left to right in lane A2\nB: x\nA3: y")
    assert d == "L2R" and txt == "A1: moving left to right in lane A2\nA3: y"
    assert data.parse_lane_from_context("in lane A3 now") == "3"
    assert data.parse_lane_from_context("lane safe") == "safe"
    assert data.parse_lane_from_context("lane A7") is None
    t = np.array([[100.0, 0.0], [90.0, 0.0], [95.0, 0.0]])
    assert data.is_trajectory_abnormal(t, "R2L") and not data.is_trajectory_abnormal(t[:2], "R2L")
    assert data.is_trajectory_abnormal(np.array([[0.0, 0.0], [60.0, 0.0]]), None)  # step > 50
    roi = {"Site C": {"A": {"1": [[0, 1], [2, 3]]}}}
    assert data.get_polygon_from_lane_roi(roi, "1").shape == (2, 2)
    assert data.get_polygon_from_lane_roi(roi, "2").shape == (0, 2)
    s = data.DistributedStridedSampler(10, 4, 1)
    assert list(s) == [1, 5, 9] and len(s) == 3
