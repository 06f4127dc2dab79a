"""Seeded synthetic inputs in the reference's batch layout ("NGSIM-shape").

The reference's dataset is a private drone-video pickle (scripts/train.py:1333); nothing
of it exists here.  These generators reproduce its *schema and value ranges*:

* ``make_tracks``  -- list of track dicts as ``build_dataset_from_tracks_sliding`` consumes
  them (scripts/train.py:129-157): raw pixel trajectories that survive its filters
  (step <= 50 px, |delta step| <= 30, x non-increasing for "R2L", window x-range >= 100 px
  after the [::5] down-sampling), per-frame 512-d vision features, a context string that
  names a lane, and a lane-ROI dict with 14-39 vertex polygons in the pixel range of
  scripts/graph.py:7-216 (x in [0,3839], y in [960,1215]).
* ``make_batch``   -- one collated batch exactly as ``custom_collate_fn`` returns it
  (scripts/train.py:301-347), directly, without going through tracks.
"""
import numpy as np

from .config import ModelConfig


def _lane_polygon(rng, n_vertices):
    """A thin quadrilateral-ish lane strip across the image, as graph.py's lanes are."""
    n_top = n_vertices // 2
    n_bot = n_vertices - n_top
    y0 = rng.uniform(965.0, 1150.0)
    height = rng.uniform(25.0, 60.0)
    xs_top = np.sort(rng.uniform(0.0, 3839.0, n_top))
    xs_bot = np.sort(rng.uniform(0.0, 3839.0, n_bot))[::-1]
    top = np.stack([xs_top, y0 + rng.normal(0, 2.0, n_top)], axis=1)
    bot = np.stack([xs_bot, y0 + height + rng.normal(0, 2.0, n_bot)], axis=1)
    return np.concatenate([top, bot], axis=0).astype(np.float32)


def _trajectory_px(rng, n_frames):
    """Right-to-left vehicle track in pixels: smooth speed, gentle lateral drift."""
    speed = rng.uniform(4.0, 7.5)  # px / frame  (x5 after down-sampling: 20-37 px / step)
    acc = rng.normal(0.0, 0.002, n_frames).cumsum()
    v = np.clip(speed + acc, 2.5, 9.0)
    x0 = rng.uniform(3000.0, 3800.0)
    x = x0 - np.cumsum(v)
    y = rng.uniform(990.0, 1180.0) + np.cumsum(rng.normal(0.0, 0.05, n_frames))
    return np.stack([x, y], axis=1).astype(np.float32)


def make_tracks(n_tracks=64, n_frames=400, vision_dim=512, seed=0):
    """Track dicts with the pickle schema of the reference (train.py:129-157)."""
    rng = np.random.default_rng([seed, 0x7AC])
    lanes = {str(i): _lane_polygon(rng, int(rng.integers(14, 40))).tolist() for i in (1, 2, 3)}
    lanes["safe"] = _lane_polygon(rng, 14).tolist()
    lane_roi = {"Site C": {"A": lanes}}
    tracks = []
    for t in range(n_tracks):
        lane = ["A1", "A2", "A3", "safe"][int(rng.integers(0, 4))]
        ctx = (f"A1: vehicle {t} drives in lane {lane} of Site C moving right to left.\n"
               f"A2: speed {rng.uniform(20, 80):.1f} km/h, heading {rng.uniform(170, 190):.1f} deg.\n"
               f"A3: no right-following vehicle.")
        tracks.append({
            "raw_trajectory": _trajectory_px(rng, n_frames),
            "vision_embeddings": rng.standard_normal((n_frames, vision_dim), dtype=np.float32),
            "context_str": ctx,
            "lane_roi": lane_roi,
            "track_id": f"syn{t:04d}",
        })
    return tracks


def make_batch(cfg: ModelConfig, batch: int, text_len: int = 240, seed: int = 0, ragged: bool = True,
               min_text: int = None, empty_polygon_every: int = 0):
    """One collated batch (numpy arrays; keys as custom_collate_fn, train.py:334-347).

    ragged=True draws per-sample text lengths in [min_text, text_len] and right-pads ids with 0,
    mask with 0 and labels with -100 (train.py:330-332).
    """
    rng = np.random.default_rng([seed, 0xBA7C4, batch, text_len])
    T, To = cfg.seq_len, cfg.out_len
    P = cfg.max_polygon_points
    traj = np.zeros((batch, 2, T), np.float32)
    target = np.zeros((batch, 2, To), np.float32)
    norm_stat = np.zeros((batch, 4), np.float32)
    polygon = np.zeros((batch, P, 2), np.float32)
    poly_len = np.zeros((batch,), np.int32)
    for b in range(batch):
        px = _trajectory_px(rng, (T + To) * 5)[::5]
        mn, mx = px.min(0), px.max(0)
        rx = max(mx[0] - mn[0], 1e-6)
        ry = mx[1] - mn[1]
        ry = ry if abs(ry) >= 1e-6 else 1.0
        nrm = (px - mn) / np.array([rx, ry], np.float32)
        traj[b] = nrm[:T].T
        target[b] = nrm[T:].T
        norm_stat[b] = (mn[0], mx[0], mn[1], mx[1])
        if empty_polygon_every and b % empty_polygon_every == empty_polygon_every - 1:
            continue
        nv = int(rng.integers(14, 40))
        polygon[b, :nv] = _lane_polygon(rng, nv)
        poly_len[b] = nv
    vision = rng.standard_normal((batch, T, cfg.vision_dim), dtype=np.float32)
    ids = rng.integers(0, cfg.llama.vocab, (batch, text_len), dtype=np.int64)
    mask = np.ones((batch, text_len), np.int64)
    labels = ids.copy()
    if ragged:
        lo = min_text if min_text is not None else max(1, text_len // 2)
        lens = rng.integers(lo, text_len + 1, batch)
        lens[0] = text_len  # at least one full row, as pad_sequence guarantees
        for b in range(batch):
            ids[b, lens[b]:] = 0
            mask[b, lens[b]:] = 0
            labels[b, lens[b]:] = -100
    return {
        "traj_emb": traj, "target_traj": target, "vision_emb": vision, "lane_polygon": polygon,
        "lane_polygon_len": poly_len, "norm_stat": norm_stat, "input_ids": ids, "attention_mask": mask,
        "labels": labels,
    }


def batch_to_samples(batch):
    """A collated batch (make_batch) taken apart into the per-sample dicts MultiModalTrajectoryDataset.__getitem__ returns
    (scripts/train.py:281-299): trajectories [T, 2], ids / mask / labels cut to the row's valid length -- so that
    data.custom_collate_fn(batch_to_samples(b)) reproduces b (used to feed a step from the host: bench.py --feed host)."""
    import torch

    out = []
    for i in range(batch["traj_emb"].shape[0]):
        n = max(1, int(batch["attention_mask"][i].sum()))
        out.append({
            "traj_emb": torch.from_numpy(np.ascontiguousarray(batch["traj_emb"][i].T)),
            "target_traj": torch.from_numpy(np.ascontiguousarray(batch["target_traj"][i].T)),
            "vision_emb": torch.from_numpy(batch["vision_emb"][i]),
            "lane_polygon": torch.from_numpy(batch["lane_polygon"][i]),
            "lane_polygon_len": int(batch["lane_polygon_len"][i]),
            "norm_stat": tuple(float(v) for v in batch["norm_stat"][i]),
            "context_str": f"A1: synthetic vehicle {i}.", "answer_str": "", "track_id": f"syn{i:04d}",
            "input_ids": torch.from_numpy(batch["input_ids"][i, :n].copy()),
            "attention_mask": torch.from_numpy(batch["attention_mask"][i, :n].copy()),
            "labels": torch.from_numpy(batch["labels"][i, :n].copy()),
        })
    return out


class SyntheticTokenizer:
    """Deterministic stand-in for the HF tokenizer the reference fetches by name (scripts/train.py:1056; no network here).

    Same call surface as the builder uses (train.py:214-229): ``tok(text, truncation=True, max_length=N,
    return_tensors="pt", add_special_tokens=False)`` -> ``{"input_ids": (1,n) int64, "attention_mask": (1,n) int64}``,
    plus ``decode(ids, skip_special_tokens=True)``, ``pad_token`` / ``eos_token`` and their ids.  Pieces are words,
    digit runs and single punctuation marks; a piece's id is a CRC of its bytes folded into [2, vocab) (0 = pad, 1 = eos),
    so ids are stable across processes and Python versions.  ``decode`` returns the pieces it has seen for the ids
    (unknown ids print as ``<id>``): enough for the marker cut-off of ``generate_batch`` to be exercised end to end."""

    pad_token, eos_token = "<pad>", "<eos>"
    pad_token_id, eos_token_id = 0, 1

    def __init__(self, vocab=128256):
        import re

        self.vocab = int(vocab)
        self._split = re.compile(r"[A-Za-z_]+|[0-9]+|\s+|[^\sA-Za-z0-9_]")
        self._seen = {}

    def piece_id(self, piece):
        import zlib

        i = 2 + zlib.crc32(piece.encode("utf-8")) % (self.vocab - 2)
        self._seen.setdefault(i, piece)
        return i

    def encode(self, text):
        return [self.piece_id(p) for p in self._split.findall(text) if not p.isspace()]

    def __call__(self, text, truncation=False, max_length=None, return_tensors=None, add_special_tokens=False,
                 padding=False, **kw):
        import torch

        if isinstance(text, (list, tuple)):
            # batch form (train.py:557, 590-598): right-padded to the longest row with pad_token_id, mask 0 on the padding
            rows = [self.encode(t) for t in text]
            if truncation and max_length is not None:
                rows = [r[:max_length] for r in rows]
            n = max((len(r) for r in rows), default=0)
            if not padding and any(len(r) != n for r in rows):
                raise ValueError("rows of different lengths need padding=True")
            ids = torch.full((len(rows), n), self.pad_token_id, dtype=torch.long)
            mask = torch.zeros((len(rows), n), dtype=torch.long)
            for i, r in enumerate(rows):
                ids[i, :len(r)] = torch.tensor(r, dtype=torch.long)
                mask[i, :len(r)] = 1
            return {"input_ids": ids, "attention_mask": mask}
        ids = self.encode(text)
        if truncation and max_length is not None:
            ids = ids[:max_length]
        t = torch.tensor([ids], dtype=torch.long).reshape(1, len(ids))
        return {"input_ids": t, "attention_mask": torch.ones_like(t)}

    def decode(self, ids, skip_special_tokens=True):
        out = []
        for i in (int(v) for v in ids):
            if skip_special_tokens and i in (self.pad_token_id, self.eos_token_id):
                continue
            out.append(self._seen.get(i, f"<{i}>"))
        # words are joined with blanks, punctuation attaches to the piece before it
        text = ""
        for p in out:
            text += p if (not text or (len(p) == 1 and not p.isalnum())) else " " + p
        return text

    @classmethod
    def from_pretrained(cls, name, **kw):
        return cls()
