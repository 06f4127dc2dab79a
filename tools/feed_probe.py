"""Where the HOST spends a step that is fed from host memory (bench.py --feed host): per-phase wall time of
collate / DeviceFeeder.put / Trainer.step / release over a few steps at the bench shape.  Measurement only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    from tcavt_amd import capi, config, data, model, synth, training
    from tcavt_amd.weights import make_weights

    capi.init(0)
    if "--threads" in sys.argv:
        torch.set_num_threads(int(sys.argv[sys.argv.index("--threads") + 1]))
    print(f"torch intra-op threads: {torch.get_num_threads()}, os.cpu_count() = {os.cpu_count()}, "
          f"affinity = {len(os.sched_getaffinity(0))}")
    dev = torch.device("cuda", 0)
    cfg = config.PRESETS["llama32_1b"](seq_len=18, out_len=30, use_lora=True)
    B = 32
    with torch.device(dev):
        m = model.MultiModalTrajectoryModel.from_config(cfg)
    m.load_weights(make_weights(cfg, seed=1, backend="torch", device=dev))
    m.train(True)
    tr = training.Trainer(m, lr=5e-4, weight_decay=1e-4)
    host = [synth.batch_to_samples(synth.make_batch(cfg, B, text_len=240, seed=100 + j, ragged=True, min_text=128)) for j in range(4)]
    fd = data.DeviceFeeder(dev)
    t = {"collate": 0.0, "put": 0.0, "step": 0.0, "release": 0.0}
    detail = []

    def fetch(i):
        t0 = time.perf_counter()
        c = data.custom_collate_fn(host[i % 4])
        t1 = time.perf_counter()
        f = fd.put(c)
        t2 = time.perf_counter()
        t["collate"] += t1 - t0
        t["put"] += t2 - t1
        return f

    import cProfile
    import gc
    import pstats

    prof = cProfile.Profile() if "--cprofile" in sys.argv else None
    if "--no-gc" in sys.argv:
        gc.disable()
    gc_t = [0.0]
    gc_mark = [0.0]

    def gc_cb(phase, info):
        if phase == "start":
            gc_mark[0] = time.perf_counter()
        else:
            gc_t[0] += time.perf_counter() - gc_mark[0]
            if time.perf_counter() - gc_mark[0] > 0.005:
                print(f"  [gc] generation {info['generation']} took {(time.perf_counter() - gc_mark[0]) * 1e3:.1f} ms")
    gc.callbacks.append(gc_cb)
    cur = fetch(0)
    n = 24
    for i in range(n):
        if i == 8:
            torch.cuda.synchronize()
            for k in t:
                t[k] = 0.0
            gc_t[0] = 0.0
            w0 = time.perf_counter()
            if prof is not None:
                prof.enable()
        nxt = fetch(i + 1)
        t0 = time.perf_counter()
        tr.step(cur["traj_emb"], cur["vision_emb"], cur["lane_polygon"], cur["lane_polygon_len"], cur["target_traj"], cur["norm_stat"],
                cur["input_ids"], cur["attention_mask"], cur["labels"], next_vision_embs=nxt["vision_emb"], next_ready=nxt.ready,
                inputs_ready=cur.ready)
        t1 = time.perf_counter()
        fd.release(cur)
        t2 = time.perf_counter()
        t["step"] += t1 - t0
        t["release"] += t2 - t1
        detail.append((t1 - t0) * 1e3)
        cur = nxt
    if prof is not None:
        prof.disable()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - w0) / (n - 8) * 1e3
    print(f"python gc inside the timed steps: {gc_t[0] * 1e3:.1f} ms in all")
    if prof is not None:
        pstats.Stats(prof).sort_stats("tottime").print_stats(18)
    print(f"wall {wall:.2f} ms/step; host per step: " + ", ".join(f"{k} {v / (n - 8) * 1e3:.2f} ms" for k, v in t.items()))
    print("step() host ms:", " ".join(f"{d:.1f}" for d in detail))


if __name__ == "__main__":
    main()
