"""Timing experiment (not a product path): how much of the trainable head (LTSF forward, backward, AdamW) hides under the
frozen MLLM pass of the NEXT batch when that pass runs on its own stream.  The head consumes a cached MLLM result
(`_llm_cache`), so the numbers say what a cross-step pipeline could reach, not what the arithmetic is."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcavt_amd import capi, config, model, synth, training
from tcavt_amd.weights import make_weights

capi.init(0)
dev = torch.device("cuda:0")
cfg = config.PRESETS["llama32_1b"](seq_len=18, out_len=30, use_lora=True)
B = 32
with torch.device(dev):
    m = model.MultiModalTrajectoryModel.from_config(cfg)
m.load_weights(make_weights(cfg, seed=1, backend="torch", device=dev))
m.train(True)
b = synth.make_batch(cfg, B, text_len=240, seed=100, ragged=True, min_text=128)
g = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
tr = training.Trainer(m, lr=5e-4, weight_decay=1e-4)


def step(nv=None):
    return tr.step(g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"], g["target_traj"],
                   g["norm_stat"], g["input_ids"], g["attention_mask"], g["labels"], next_vision_embs=nv)


def timed(fn, n=30, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


with torch.no_grad():
    step()
    torch.cuda.synchronize()
    print(f"serial step with Q-Former prefetch: {timed(lambda: step(g['vision_emb'])):.3f} ms", flush=True)

    def mllm_only():
        m.mllm(g["vision_emb"], None, input_ids=g["input_ids"], attention_mask=g["attention_mask"], return_bf16=True)

    print(f"MLLM pass alone: {timed(mllm_only):.3f} ms", flush=True)
    m._llm_cache = (m.last.final_hidden, m.last.final_hidden_bf16)
    print(f"head alone (cached MLLM result): {timed(lambda: step()):.3f} ms", flush=True)

    prio = int(os.environ.get("EXP_D_PRIO", "0"))
    D = torch.cuda.Stream(device=dev, priority=prio)
    hp = int(os.environ.get("EXP_HEAD_PRIO", "0"))
    H = torch.cuda.Stream(device=dev, priority=hp) if hp else None
    if hp:  # head on high-priority streams: its launches go ahead of the decoder's pending workgroups
        from tcavt_amd import streams
        streams._POOL[("cuda", 0)] = [torch.cuda.Stream(device=dev, priority=hp) for _ in range(streams.N_SLOTS)]
        m._side = m.ltsf._kv_stream = None
        tr.bw._leaf_streams = tr.bw._poly_stream = None
        m.mllm._pf_stream = None
    state = {"done": None}

    def piped():
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        D.wait_event(ev)  # pass N+1 may start once step N-1 is over (its buffers are free)
        prev = state["done"]
        with torch.cuda.stream(D):
            mllm_only()
            d = torch.cuda.Event()
            d.record(D)
        state["done"] = d
        if H is not None:
            H.wait_stream(main)
            with torch.cuda.stream(H):
                if prev is not None:
                    H.wait_event(prev)
                step()
            main.wait_stream(H)
            return
        if prev is not None:
            main.wait_event(prev)  # head N needs pass N
        step()

    print(f"pipelined (MLLM of the next batch on its own stream, D prio {prio}, head prio {hp}): {timed(piped):.3f} ms", flush=True)

    # ---- the head (forward from cached MLLM result + loss + backward + AdamW) as a hipGraph: how much of its 3.9 ms is
    # launch / cross-stream dependency latency that a replay removes?
    if os.environ.get("EXP_HEAD_GRAPH", "1") == "1":
        m.pipeline_decoder = False
        graph, _ = tr.capture(g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"], g["target_traj"],
                              g["norm_stat"], g["input_ids"], g["attention_mask"], g["labels"])
        print(f"head alone as a hipGraph replay: {timed(graph.replay):.3f} ms", flush=True)

        def piped_graph():
            main = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            ev.record(main)
            D.wait_event(ev)
            prev = state["done"]
            with torch.cuda.stream(D):
                mllm_only()
                d = torch.cuda.Event()
                d.record(D)
            state["done"] = d
            if prev is not None:
                main.wait_event(prev)
            graph.replay()

        print(f"pipelined, head as a graph: {timed(piped_graph):.3f} ms", flush=True)

        def serial_graph():
            mllm_only()
            graph.replay()

        print(f"serial (MLLM pass incl. Q-Former, then head graph): {timed(serial_graph):.3f} ms", flush=True)
        tr.release_graph()
