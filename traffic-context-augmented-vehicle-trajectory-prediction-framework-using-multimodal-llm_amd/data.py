"""Host-side data path: track filtering, sliding windows, per-window normalisation, dataset, collate.

Mirror of the reference's L2/L3 layers (scripts/train.py:31-347; scripts/baseline_cv.py:25-183), which
define the batch layout the hot path consumes.  Same function names, arguments and
outputs; the implementation is this repo's own (window extraction and normalisation are vectorised
with numpy instead of per-window Python loops, tokenisation is done once per track because prompt and
answer depend on the track only, not on the window).

Pinned by tests/golden/cv_64tracks.npz (window count, every window's norm_stat, collated tensors),
which was produced by the reference's own builder on tcavt_amd.synth.make_tracks(seed=0).
"""
import random
import re

import numpy as np
import torch

_A456 = re.compile(r"^\s*A[4-6]\s*:")
_A123 = re.compile(r"^\s*A[1-3]\s*:")
_LANE = re.compile(r"lane\s+(A[1-3]|safe)")


def split_all_data(all_data, train_ratio=0.7, val_ratio=0.2, test_ratio=0.1):
    """In-place shuffle with the `random` module, then 70/20/10 (train.py:31-39)."""
    random.shuffle(all_data)
    n = len(all_data)
    a = int(n * train_ratio)
    b = a + int(n * val_ratio)
    return all_data[:a], all_data[a:b], all_data[b:]


def check_data_sanity(all_data, max_coord_threshold=1e6, verbose=True):
    """Tracks whose raw_trajectory is present, finite and within +-max_coord_threshold (as float32), in order
    (modify_scripts/modify_train.py:26-49, called in front of the split at :1061)."""
    clean = []
    for d in all_data:
        raw = d.get("raw_trajectory", None)
        if raw is None:
            continue
        raw = np.array(raw, dtype=np.float32)
        if not np.all(np.isfinite(raw)):
            continue
        if np.abs(raw).max() > max_coord_threshold:
            continue
        clean.append(d)
    if verbose:
        print(f"[check_data_sanity] clean_data: {len(clean)} / {len(all_data)}")
    return clean


def filter_context(context):
    """-> (kept A1..A3 lines, direction) or (None, None) when an A4..A6 line is present (train.py:44-65)."""
    if not context.strip():
        return "No context provided", "R2L"
    kept = []
    for line in context.splitlines():
        if _A456.match(line):
            return None, None
        if _A123.match(line):
            kept.append(line)
    if not kept:
        return "No valid context lines", "R2L"
    low = context.lower()
    direction = "L2R" if "left to right" in low else "R2L"
    return "\n".join(kept).strip(), direction


def parse_lane_from_context(context_str):
    m = _LANE.search(context_str)
    if not m:
        return None
    lane = m.group(1)
    return "safe" if lane == "safe" else lane[1:]


def get_polygon_from_lane_roi(lane_roi_dict, lane_str):
    if lane_str is None:
        return np.zeros((0, 2), dtype=np.float32)
    sub = lane_roi_dict.get("Site C", {}).get("A", {})
    if lane_str not in sub:
        return np.zeros((0, 2), dtype=np.float32)
    return np.array(sub[lane_str], dtype=np.float32)


def is_trajectory_abnormal(raw_traj, lane_label=None, max_step=50.0, max_speed_diff=30.0):
    """Step length, change of step length, and monotone x for the lane direction (train.py:89-108)."""
    if raw_traj.shape[0] < 2:
        return False
    steps = np.sqrt(np.sum((raw_traj[1:] - raw_traj[:-1]) ** 2, axis=-1))
    if np.any(steps > max_step) or np.any(np.abs(steps[1:] - steps[:-1]) > max_speed_diff):
        return True
    x = raw_traj[:, 0]
    if lane_label == "R2L":
        return bool(np.any(x[1:] > x[:-1]))
    if lane_label == "L2R":
        return bool(np.any(x[1:] < x[:-1]))
    return False


PROMPT_TEMPLATE = (
    "You are analyzing the ego vehicle with track_id={track_id}.\n"
    "Below is partial information about this ego vehicle and its surroundings.\n"
    "Use the provided data (<vision>) to create a comprehensive text describing:\n"
    "1) the ego vehicle's lane, site, and bounding box dimensions,\n"
    "2) velocity, acceleration, and heading info,\n"
    "3) neighbor vehicles,\n"
    "4) average speed in the area.\n\n"
    "Please provide your answer as a natural language paragraph.\n\n"
    "Answer:\n"
)


def _tokenize_track(tokenizer, prompt_text, answer_text, max_length):
    """Prompt + answer ids, mask, labels (-100 on the prompt), truncated to max_length (train.py:214-238)."""
    pe = tokenizer(prompt_text, truncation=True, max_length=max_length, return_tensors="pt", add_special_tokens=False)
    ae = tokenizer(answer_text, truncation=True, max_length=max_length, return_tensors="pt", add_special_tokens=False)
    ids = torch.cat([pe["input_ids"], ae["input_ids"]], dim=1)
    mask = torch.cat([pe["attention_mask"], ae["attention_mask"]], dim=1)
    labels = torch.full_like(ids, -100)
    n_prompt = pe["input_ids"].size(1)
    labels[:, n_prompt:] = ids[:, n_prompt:]
    return ids[0, :max_length], mask[0, :max_length], labels[0, :max_length]


def build_dataset_from_tracks_sliding(track_list, seq_len=30, out_len=60, stride=1, max_step=50.0, max_speed_diff=30.0,
                                      image_width=3840, image_height=1280, downsample=5, tokenizer=None,
                                      max_length=512):
    """Sliding windows over every track that survives the filters (train.py:114-259).

    Returns (inputs_list, outputs_list): per window a dict with trajectory_embeddings (T_in,2) normalised
    to the window's own min/max, vision_embeddings (T_in,512) when the track has them, context_str,
    answer_str, norm_stat (min_x,max_x,min_y,max_y), track_id, lane_polygon, input_ids/attention_mask/
    labels; and the normalised future (T_out,2).  Windows whose x-range is below 100 px are dropped.
    """
    inputs_list, outputs_list = [], []
    win = seq_len + out_len
    for item in track_list:
        raw = np.asarray(item["raw_trajectory"])[::downsample]
        vision = item.get("vision_embeddings", None)
        if vision is not None:
            vision = vision[::downsample]
        ctx = item.get("context_str", "")
        lane_roi = item.get("lane_roi", None)
        if lane_roi is None:
            continue
        filtered, direction = filter_context(ctx)
        if filtered is None:
            continue
        lane_str = parse_lane_from_context(ctx)
        if lane_str is None:
            continue
        polygon = get_polygon_from_lane_roi(lane_roi, lane_str)
        if is_trajectory_abnormal(raw, lane_label=direction, max_step=max_step, max_speed_diff=max_speed_diff):
            continue
        n = raw.shape[0]
        if n < win:
            continue
        track_id = item.get("track_id", "unknown")
        starts = np.arange(0, n - win + 1, stride)
        # all windows of the track at once: [n_windows, win, 2]
        idx = starts[:, None] + np.arange(win)[None, :]
        w = raw[idx]
        mn = w.min(axis=1)
        mx = w.max(axis=1)
        prompt_text = PROMPT_TEMPLATE.format(track_id=track_id)
        tok = _tokenize_track(tokenizer, prompt_text, ctx, max_length) if tokenizer is not None else None
        for k, start in enumerate(starts):
            min_x, max_x = float(mn[k, 0]), float(mx[k, 0])
            min_y, max_y = float(mn[k, 1]), float(mx[k, 1])
            rx, ry = max_x - min_x, max_y - min_y
            if rx < 100:
                continue
            if abs(rx) < 1e-6:
                rx = 1.0
            if abs(ry) < 1e-6:
                ry = 1.0
            norm = np.zeros((win, 2), dtype=np.float32)
            norm[:, 0] = (w[k, :, 0] - min_x) / rx
            norm[:, 1] = (w[k, :, 1] - min_y) / ry
            sample = {
                "trajectory_embeddings": torch.from_numpy(norm[:seq_len].copy()),
                "context_str": prompt_text,
                "answer_str": ctx,
                "norm_stat": (min_x, max_x, min_y, max_y),
                "track_id": track_id,
                "lane_polygon": polygon,
            }
            if vision is not None:
                v = vision[start:start + seq_len]
                v = torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v
                if v.shape[0] < seq_len:
                    v = torch.cat([v, torch.zeros(seq_len - v.shape[0], v.shape[1], dtype=v.dtype)], dim=0)
                sample["vision_embeddings"] = v.float()
            else:
                sample["vision_embeddings"] = torch.zeros(seq_len, 1, dtype=torch.float32)
            if tok is not None:
                sample["input_ids"], sample["attention_mask"], sample["labels"] = tok
            else:
                sample["input_ids"] = torch.zeros(1, dtype=torch.long)
                sample["attention_mask"] = torch.ones(1, dtype=torch.long)
                sample["labels"] = torch.zeros(1, dtype=torch.long)
            inputs_list.append(sample)
            outputs_list.append(torch.from_numpy(norm[seq_len:].copy()))
    return inputs_list, outputs_list


class MultiModalTrajectoryDataset(torch.utils.data.Dataset):
    """train.py:264-299: pads / truncates the lane polygon to max_polygon_points and reports its length."""

    def __init__(self, inputs_list, outputs_list, max_polygon_points=64):
        assert len(inputs_list) == len(outputs_list)
        self.inputs_list, self.outputs_list, self.max_polygon_points = inputs_list, outputs_list, max_polygon_points

    def __len__(self):
        return len(self.inputs_list)

    def __getitem__(self, idx):
        s = self.inputs_list[idx]
        out = {
            "traj_emb": s["trajectory_embeddings"], "vision_emb": s["vision_embeddings"], "context_str": s["context_str"],
            "answer_str": s["answer_str"], "norm_stat": s["norm_stat"], "target_traj": self.outputs_list[idx],
            "track_id": s.get("track_id", None), "input_ids": s["input_ids"], "attention_mask": s["attention_mask"],
            "labels": s["labels"],
        }
        polygon = s["lane_polygon"]
        n_p = polygon.shape[0] if polygon is not None else 0
        P = self.max_polygon_points
        padded = np.zeros((P, 2), dtype=np.float32)
        n = min(n_p, P)
        if n > 0:
            padded[:n] = polygon[:n]
        out["lane_polygon"] = torch.from_numpy(padded)
        out["lane_polygon_len"] = n
        return out


def custom_collate_fn(batch):
    """train.py:301-347: (B,2,T) trajectories, stacked vision/polygons, right-padded ids (0), mask (0),
    labels (-100); python lists for lengths / norm_stat / strings."""
    from torch.nn.utils.rnn import pad_sequence

    return {
        "traj_emb": torch.stack([b["traj_emb"].transpose(0, 1) for b in batch], dim=0),
        "target_traj": torch.stack([b["target_traj"].transpose(0, 1) for b in batch], dim=0),
        "vision_emb": torch.stack([b["vision_emb"] for b in batch], dim=0),
        "lane_polygon": torch.stack([b["lane_polygon"] for b in batch], dim=0),
        "lane_polygon_len": [b["lane_polygon_len"] for b in batch],
        "norm_stat": [b["norm_stat"] for b in batch],
        "context_str": [b["context_str"] for b in batch],
        "answer_str": [b["answer_str"] for b in batch],
        "track_id": [b["track_id"] for b in batch],
        "input_ids": pad_sequence([b["input_ids"] for b in batch], batch_first=True, padding_value=0),
        "attention_mask": pad_sequence([b["attention_mask"] for b in batch], batch_first=True, padding_value=0),
        "labels": pad_sequence([b["labels"] for b in batch], batch_first=True, padding_value=-100),
    }


class DistributedStridedSampler:
    """Index striding by rank without shuffling, padded to equal length (what DistributedSampler(shuffle=False)
    yields, train.py:1093,1271): rank r takes r, r+W, r+2W, ...."""

    def __init__(self, n, world, rank):
        self.n, self.world, self.rank = n, world, rank
        self.per_rank = (n + world - 1) // world

    def __iter__(self):
        idx = list(range(self.n))
        idx += idx[: self.per_rank * self.world - self.n]
        return iter(idx[self.rank::self.world])

    def __len__(self):
        return self.per_rank


class FedBatch(dict):
    """A collated batch whose tensors live on the device: the dict custom_collate_fn returns (tensor entries moved, lists of
    numbers as tensors, strings kept), plus `ready` -- the event of the copy that filled it -- and `slot`."""
    ready = None
    slot = -1


class DeviceFeeder:
    """Host -> device feeding of the training loop (scripts/train.py:1153-1166).

    The reference moves seven tensors of every batch with blocking ``.to(device)`` calls from pageable memory at the head of
    the step.  Here a batch goes through a ring of `slots` (>= 3) staging sets: pinned host buffers, then ONE copy kernel
    (tcavt_copy_batch: its lanes read the pinned buffers over the host link), and the step is told WHEN its inputs are ready
    (``FedBatch.ready``, an event) instead of waiting for them on the host: ``Trainer.step(..., inputs_ready=fed.ready)``.
    With the pipelined decoder (training.Trainer, frozen-MLLM variant) three batches are alive at once -- step i's head and
    backward, step i + 1's decoder, and the copy of batch i + 2 -- hence the ring.

    The copy is enqueued on the CALLER'S stream (the stream the steps are enqueued on): put(batch i + 1) is called before
    step i is enqueued, so on the card it runs after step i - 1's optimizer and before step i's head -- while the decoder of
    step i is busy on its own stream and the caller's stream would be idle anyway.  Stream order is then all the protection a
    slot needs (whatever read it three batches ago was enqueued, or joined, on the same stream before), and no sixth stream
    appears: the pipelined step keeps its cross-step overlap only while at most five are in use.  A
    `stream` may still be given (e.g. a loader thread's own); its copies then wait for ``release()`` of the slot's previous
    user on the HOST (a stream-side wait for an event that late would hold the stream's hardware queue).

    Host threads: keep torch's intra-op pool small in the feeding process (``torch.set_num_threads(2)``).  The collate and the
    staging copies are a few small CPU ops; at its default size the pool starts one thread per CPU the machine SHOWS, not per
    CPU the process may use, and under a container's CPU quota the spinning threads get the whole process frozen for the
    rest of the scheduler period: measured 90 ms stalls every few steps, 31-42 instead of 15.3 ms per step
    (profiles/r04_feed_host.txt)."""

    TENSOR_KEYS = ("traj_emb", "target_traj", "vision_emb", "lane_polygon", "input_ids", "attention_mask", "labels")
    LIST_KEYS = (("lane_polygon_len", torch.int32), ("norm_stat", torch.float32))

    def __init__(self, device, slots=3, stream=None):
        if slots < 3:
            raise ValueError("DeviceFeeder: at least three slots (two batches in flight + the one being copied)")
        self.device = torch.device(device)
        self.stream = stream  # None: the caller's current stream at put() time
        self._slots = [None] * slots
        self._free = [None] * slots
        self._count = 0
        self.bytes_per_batch = 0

    def _host_tensors(self, batch):
        out = {}
        for k in self.TENSOR_KEYS:
            out[k] = batch[k] if torch.is_tensor(batch[k]) else torch.as_tensor(batch[k])
        for k, dt in self.LIST_KEYS:
            v = batch[k]
            out[k] = (v if torch.is_tensor(v) else torch.tensor([list(r) for r in v] if k == "norm_stat" else list(v))).to(dt)
        return out

    def put(self, batch):
        """Stage one collated batch (host tensors / lists) and start its copy; returns the FedBatch at once."""
        from . import ops

        k = self._count % len(self._slots)
        self._count += 1
        host = self._host_tensors(batch)
        slot = self._slots[k]
        if slot is None or any(slot["pin"][n].shape != t.shape or slot["pin"][n].dtype != t.dtype for n, t in host.items()):
            if slot is not None:
                slot["ev"].synchronize()
                for t in slot["dev"].values():
                    t.record_stream(torch.cuda.current_stream(self.device))  # (a step enqueued earlier may still read it)
            slot = self._slots[k] = {
                "pin": {n: torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for n, t in host.items()},
                "dev": {n: torch.empty(t.shape, dtype=t.dtype, device=self.device) for n, t in host.items()}, "ev": None}
        if slot["ev"] is not None:
            slot["ev"].synchronize()  # the previous copy OUT of this pinned set (len(slots) batches ago: long done)
        if self.stream is not None and self._free[k] is not None:
            self._free[k].synchronize()  # own stream: HOST wait for the slot's previous user (see the class docstring)
        for n, t in host.items():
            slot["pin"][n].copy_(t)
        self.bytes_per_batch = sum(t.numel() * t.element_size() for t in host.values())
        st = self.stream if self.stream is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.stream(st):
            # ONE kernel launch that reads the pinned buffers over the host link (ops.copy_batch) instead of nine copy-engine transfers
            names = list(host)
            ops.copy_batch([slot["dev"][n] for n in names], [slot["pin"][n] for n in names])
            ev = torch.cuda.Event()
            ev.record(st)
        slot["ev"] = ev
        fed = FedBatch(slot["dev"])
        for n in ("context_str", "answer_str", "track_id"):
            if n in batch:
                fed[n] = batch[n]
        fed.ready, fed.slot = ev, k
        return fed

    def release(self, fed):
        """Call after the step that consumes `fed` has been enqueued (on the stream it was enqueued on)."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._free[fed.slot] = ev
