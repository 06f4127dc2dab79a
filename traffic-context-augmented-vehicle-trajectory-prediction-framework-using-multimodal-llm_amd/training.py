"""Training step of scripts/train.py on the HIP path (train.py:1140-1145, 1168-1184).

    zero_grad -> forward (loss) -> backward -> [DDP gradient all-reduce] -> AdamW(lr 5e-4, wd 1e-4)

Trainable set = everything outside ``mllm`` (the MLLM is frozen after the DDP wrap, train.py:1141-1142):
lane-polygon encoder + TransformerLTSF, 18.1 M parameters at the Llama-3.2-1B shape.  They live in ONE
flat fp32 vector (backward.GradBook) so that the optimizer is a single fused kernel and the data-parallel
gradient exchange is a few large all-reduces over RCCL/xGMI on a side stream, launched as soon as a
contiguous bucket of the flat gradient is complete and overlapped with the rest of the backward
(reference: DistributedDataParallel's bucketed all-reduce, train.py:1127-1132; DDP averages gradients,
which is folded into the AdamW kernel as grad_scale = 1/world).
"""
import os

import torch
import torch.distributed as dist

from . import ops, streams
from .backward import Backward, GradBook
from .llm_backward import LoraBackward, lora_named_parameters
from .qformer_backward import QFormerBackward


def trainable_named_parameters(model):
    """Ordered by readiness in the backward: LTSF first (its head and cross-attention gradients are
    produced first), lane-polygon encoder last."""
    ltsf = [(n, p) for n, p in model.named_parameters() if n.startswith("ltsf.")]
    poly = [(n, p) for n, p in model.named_parameters() if n.startswith("lane_polygon_encoder.")]
    return ltsf + poly


class Trainer:
    """lora_trainable=False: scripts/train.py (whole MLLM frozen, :1140-1145).

    lora_trainable=True: the **LoRA-only subset** of modify_scripts/modify_train.py -- lora_A / lora_B of q_proj, v_proj
    train as well (:512-528; the backward walks through the frozen decoder layers, llm_backward.LoraBackward), gradients
    are clipped to max_grad_norm (:1192) and an update is skipped when the loss is not finite (:1190-1196, decided on
    the device).  NOT the whole of modify_train.py's trainable set: that script never freezes `mllm`, so its Q-Former,
    mllm.q_proj and the two modality embeddings (54 M parameters) keep requires_grad there; here the backward stops at
    the input of decoder layer 0 and those stay frozen ("LoRA-only subset").

    lora_trainable=True, train_mllm_front=True: the WHOLE trainable set of modify_train.py -- the backward continues below
    decoder layer 0 into the modality embeddings, mllm.q_proj and the Q-Former (qformer_backward.QFormerBackward).  The
    Q-Former then changes every step, so its pass cannot be prefetched under the previous step (next_vision_embs is ignored).

    Data parallel (world > 1): the constructor broadcasts rank 0's parameters to all ranks (what the reference's
    DistributedDataParallel wrap does at construction, train.py:1127-1132); gradients are SUM-all-reduced in buckets and
    turned into the mean before clipping (or inside AdamW when there is no clipping)."""

    def __init__(self, model, lr=5e-4, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8, process_group=None,
                 lora_trainable=False, max_grad_norm=None, skip_nonfinite=None, sync_initial_state=True,
                 check_flags_every=0, train_mllm_front=False):
        self.model = model
        dev = next(model.parameters()).device
        named = trainable_named_parameters(model)
        for p in model.mllm.parameters():  # train.py:1141-1142
            p.requires_grad_(False)
        self.lora_trainable = bool(lora_trainable)
        # (the LoRA-trainable variant runs on the model's storage type as well -- fp16 by default: the forward's tapes and,
        # under a device-chosen power-of-two scale, the gradients of the decoder backward are IEEE half; set_storage(bfloat16)
        # before constructing the Trainer keeps round 1's bf16 contract)
        self.max_grad_norm = max_grad_norm
        # modify_train.py:1190-1196 skips clip + step on a non-finite loss; train.py has no such test
        self.skip_nonfinite = self.lora_trainable if skip_nonfinite is None else bool(skip_nonfinite)
        self.check_flags_every = int(check_flags_every)  # > 0: host check of the input error flags every N steps (one sync)
        n_frozen_variant = len(named)
        if self.lora_trainable:
            if not model.mllm.llama_wrapper.use_lora:
                raise ValueError("Trainer(lora_trainable=True): the model has no LoRA adapters (use_lora=False)")
            lora = lora_named_parameters(model)
            for _, p in lora:
                p.requires_grad_(True)
            named = named + lora  # their gradients are the last to become ready
        self.train_mllm_front = bool(train_mllm_front)
        if self.train_mllm_front:
            if not self.lora_trainable:
                raise ValueError("Trainer(train_mllm_front=True) needs lora_trainable=True (the gradient reaches the MLLM's "
                                 "front end through the decoder's backward)")
            front = [(n, p) for n, p in model.named_parameters()
                     if n.startswith(("mllm.q_proj.", "mllm.qformer.")) or n in ("mllm.vision_modality_embedding",
                                                                                  "mllm.text_modality_embedding")]
            for _, p in front:
                p.requires_grad_(True)
            named = named + front  # ready last of all
        self.book = GradBook(named, dev)
        self.n_base = self.book.end_of(named[n_frozen_variant - 1][0])  # end of the train.py parameter set
        self.n_ltsf = self.book.end_of([n for n, _ in named if n.startswith("ltsf.")][-1])
        self.m = torch.zeros_like(self.book.params)
        self.v = torch.zeros_like(self.book.params)
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.step_count = 0
        self.bw = Backward(model, self.book)
        self.lbw = None
        if self.lora_trainable:
            self.bw.poly_after_chain = True  # (the decoder's backward follows on the caller's stream: backward.Backward._ltsf_stage)
            self.lbw = LoraBackward(model, self.book)
            model.mllm.llama_wrapper.save_for_backward = True
            model.ltsf.absorb_kv = False  # _lora_backward starts from dL/dk, dL/dv of the un-absorbed cross-attention
            self._lora_stacked = self._stacked_lora_views(lora)
        self.qbw = None
        if self.train_mllm_front:
            self.qbw = QFormerBackward(model, self.book, self.bw)
            self.lbw.input_grad = True
            model.mllm.qformer.save_for_backward = True
        model.lane_polygon_encoder.save_for_backward = True
        model.ltsf.save_for_backward = True
        # train.py's frozen MLLM: its pass reads nothing this step's backward / optimizer writes, so it runs on a stream
        # of its own and the next step's decoder overlaps this step's head, backward and AdamW (model.pipeline_decoder)
        model.pipeline_decoder = not self.lora_trainable and dev.type == "cuda"
        model.mllm.skip_f32_hidden = not self.lora_trainable  # the head consumes the 16-bit final hidden states only
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        # Stream budget.  The pipelined step (model.pipeline_decoder) keeps its overlap only while at most FIVE HIP streams are
        # in use: one process uses the caller's stream, the MLLM stream and three side channels; a sixth active stream -- measured
        # with a stand-in for RCCL's own stream -- made the MLLM stream idle 1.9 ms per step (15.6 -> 17.0 ms, i.e. the whole
        # overlap; 4 or 8 hardware queues alike).  Data-parallel ranks therefore (i) launch the all-reduce from the stream that
        # produced the bucket instead of a communication stream of their own (torch's process group runs it on its RCCL stream
        # and makes the launching stream wait: the exchange sits in the step's tail, which the next decoder hides), and (ii) fold
        # the side channels onto two streams (no measurable cost: 15.34 vs 15.34 ms).
        self._fake_dp = None  # tools only (TCAVT_FAKE_DP=1): one process with a stand-in for RCCL's stream
        # TCAVT_FORCE_DP=1 (tests, bench.py --rccl-self-test): a ONE-rank process group takes the data-parallel path -- the real
        # RCCL all-reduce per bucket on the real process-group stream, the reduced stream layout -- on the one GPU a box has
        self._force_dp = (self.world == 1 and os.environ.get("TCAVT_FORCE_DP", "0") == "1" and dist.is_available()
                          and dist.is_initialized())
        if dev.type == "cuda" and (self.world > 1 or self._force_dp or os.environ.get("TCAVT_FAKE_DP", "0") == "1"):
            streams.set_active_slots(2)
            model._side = model.ltsf._kv_stream = model.mllm._pf_stream = None  # (re-drawn from the pool on next use)
            if self.world == 1 and not self._force_dp:
                self._fake_dp = torch.cuda.Stream(device=dev)
        # Measurement only (bench.py's same-build single-rank comparison leg): local_steps() switches the gradient exchange off
        # for a few steps and re-synchronises the replicas afterwards -- without the exchange every rank steps on its own
        # gradients and the replicas drift apart, which nothing else would ever repair (only gradients are exchanged)
        self._exchange = True
        self.diag = None       # enable_diagnostics(): per-bucket all-reduce timings of the steps that follow
        self._ctl = torch.zeros(8, dtype=torch.int32, device=dev)  # device-side step counters of the gated optimizer
        if self.lbw is not None:  # the fp16 backward's scale backs off when an update was skipped for a non-finite norm (ctl[6])
            self.lbw.scale_backoff = self._ctl[6:7]
        self._last_loss = None
        self._skipped_seen = 0  # check_flags(): gated-optimizer skips already reported
        # hipGraph replay of the whole step (capture()): the optimizer's step count and the dropout epoch live on the device
        self.device_step = False
        self._finite = torch.zeros(1, dtype=torch.float32, device=dev)  # an always-finite "loss" for the un-gated loop
        self._epoch = None
        if self.world > 1 and sync_initial_state:
            self.broadcast_state()
        model.invalidate_prepared()

    def local_steps(self):
        """Context manager (measurement only): inside it the gradient exchange is skipped -- each rank steps on its own
        gradients, scaled by 1 / world as usual.  On exit rank 0's parameters and optimizer moments are broadcast again, so
        the data-parallel job continues from ONE state (what the ranks did inside is discarded everywhere but on rank 0)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            self._exchange = False
            try:
                yield self
            finally:
                self._exchange = True
                if self.world > 1:
                    torch.cuda.synchronize()
                    self.broadcast_state()
        return ctx()

    def broadcast_state(self, src=0):
        """Rank `src`'s model parameters (frozen ones included) and optimizer moments to every rank: ranks that built or
        loaded their weights differently would otherwise diverge silently, because only gradients are exchanged."""
        with torch.no_grad():
            dist.broadcast(self.book.params, src=src, group=self.pg)  # the trainable set (parameters are views into it)
            in_book = set(self.book.names)
            for n, p in self.model.named_parameters():
                if n not in in_book:
                    dist.broadcast(p.data, src=src, group=self.pg)
            dist.broadcast(self.m, src=src, group=self.pg)
            dist.broadcast(self.v, src=src, group=self.pg)
        self.model.invalidate_prepared()

    # ---- gradient exchange ------------------------------------------------------------------
    def _allreduce_bucket(self, lo, hi):
        """SUM all-reduce of grads[lo:hi], launched from the current stream (the one that completed the bucket); the mean is
        taken by the clip / AdamW kernels.  torch.distributed runs it on the process group's own RCCL stream, ordered after
        the current stream, and makes the current stream wait for it."""
        if (self.world == 1 and self._fake_dp is None and not self._force_dp) or not self._exchange:
            return
        view = self.book.grads[lo:hi]
        if self.diag is not None and self._fake_dp is None:
            # events on the LAUNCHING stream around the collective: torch runs it on the process group's RCCL stream and makes
            # this stream wait for it, so stop - start = how long the exchange held this stream (queueing included)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(view, group=self.pg)
            e1.record()
            self.diag.append((int(lo), int(hi), e0, e1))
            return
        if self._fake_dp is not None:  # tools: the stream choreography of an RCCL collective, an in-place kernel in its place
            cur = torch.cuda.current_stream()
            self._fake_dp.wait_stream(cur)
            with torch.cuda.stream(self._fake_dp):
                view.mul_(1.0)
            cur.wait_stream(self._fake_dp)
            return
        dist.all_reduce(view, group=self.pg)

    # ---- one optimisation step -----------------------------------------------------------------
    def forward_backward(self, x, vision_embs, lane_polygon_batch, lane_polygon_len, y, norm_stat, input_ids,
                         attention_mask, labels=None, next_vision_embs=None, inputs_ready=None, next_ready=None):
        """zero_grad + forward + backward (+ bucketed all-reduce); gradients end up in ``self.book.g``.
        next_vision_embs (optional): the next batch's vision embeddings, already resident -- its frozen Q-Former is
        enqueued on a side stream between this step's forward and backward (model.prefetch) and runs under this step's
        decoder; results are identical with or without it.
        inputs_ready: see MultiModalTrajectoryModel.forward (pipelined decoder of the frozen-MLLM variant: an event /
        True lets this step's decoder start under the previous step's head, backward and optimizer).  An event (the copy that
        uploaded the batch: data.DeviceFeeder) is also waited for by the caller's stream, which reads the head's inputs.
        next_ready: the same for next_vision_embs (the event of ITS upload), handed to the Q-Former prefetch."""
        m = self.model
        with torch.no_grad():
            self.book.grads.zero_()  # optimizer.zero_grad()
            if isinstance(inputs_ready, torch.cuda.Event):
                torch.cuda.current_stream().wait_event(inputs_ready)
            m.inputs_ready = inputs_ready
            loss, decoded = m(x, vision_embs, None, lane_polygon_batch, lane_polygon_len, y=y, norm_stat=norm_stat,
                              input_ids=input_ids, attention_mask=attention_mask, labels=labels)
            ns = norm_stat if torch.is_tensor(norm_stat) else torch.tensor([list(n) for n in norm_stat])
            ns = ns.to(device=x.device, dtype=torch.float32).contiguous()
            if next_vision_embs is not None and not self.train_mllm_front:
                m.prefetch(next_vision_embs, ready=next_ready)  # before the backward: its leaf work shares the prefetch stream's queue
            B, L = input_ids.shape[0], m.mllm.qformer.num_query_tokens + input_ids.shape[1]
            fh_b = m.last.final_hidden_bf16  # [B * L + 64 zeroed pad rows, H]
            self.bw.run(decoded, y.contiguous(), ns, x.contiguous(), m.last.poly_emb, fh_b, L,
                        after_ltsf=lambda: self._allreduce_bucket(0, self.n_ltsf))
            if self.lbw is not None:
                self._lora_backward(B, L)   # (the caller's chain continues; the lane-polygon encoder's backward runs beside it)
                self.bw.join()
                self._allreduce_bucket(self.n_ltsf, self.n_base)
                self._allreduce_bucket(self.n_base, self.book.total)
            else:
                self._allreduce_bucket(self.n_ltsf, self.n_base)
        self._last_loss = loss
        return loss, decoded

    def _stacked_lora_views(self, lora):
        """(A_q, B_q, A_v, B_v) of all layers as strided views of the flat parameter vector (lora_named_parameters puts
        the layers last-first, four matrices each, so consecutive layers are one constant stride apart)."""
        offs = [self.book.offsets[n] for n, _ in lora]
        nL = len(lora) // 4
        if nL == 1:
            stride = 0
        else:
            stride = offs[4][0] - offs[0][0]
            if any(offs[4 * i + k][0] - offs[k][0] != i * stride for i in range(nL) for k in range(4)):
                return None  # (not a uniform layout: refresh_lora falls back to per-layer copies)
        views = []
        for k in range(4):
            off, _, shape = offs[k]
            views.append(self.book.params.as_strided((nL,) + tuple(shape), (stride, shape[1], 1), off))
        return tuple(views)

    def _lora_backward(self, B, L):
        """Gradient of the decoder's final hidden states = what flows back through the cross-attention's key and value
        in-projections (k = fh W_k^T + b_k, v = fh W_v^T + b_v; the LTSF backward left dL/dk, dL/dv behind), then the
        walk through the frozen layers."""
        m, bw = self.model, self.bw
        H = m.llama_hidden_size
        ca = m.ltsf.decoder.cross_attn
        g_k = bw._buf("xa.g_k", (B * L + 64, H), torch.bfloat16)
        g_v = bw._buf("xa.g_v", (B * L + 64, H), torch.bfloat16)
        gfa = bw._buf("lora.gfa", (B * L, H), torch.bfloat16)
        gfb = bw._buf("lora.gfb", (B * L, H), torch.bfloat16)
        for W, g, out, tag in ((ca.in_proj_weight[H:2 * H], g_k, gfa, "k"), (ca.in_proj_weight[2 * H:], g_v, gfb, "v")):
            WT = bw._buf(f"lora.WT{tag}", (H, H), torch.bfloat16)
            ops.transpose_f32_bf16(W.detach(), WT, H, H, H)
            ops.gemm_bf16(g[: B * L], WT, out=out)
        g_h0 = self.lbw.run(gfa, gfb)
        if self.qbw is not None:
            self.qbw.run(g_h0, B, L)

    def clip_grad_norm_(self, max_norm, grad_scale=1.0):
        """torch.nn.utils.clip_grad_norm_(trainable, max_norm) (modify_train.py:1192) on the flat gradient vector,
        without a host synchronisation.  grad_scale is applied to the gradient first (1 / world: the SUM-all-reduced
        buckets become DDP's mean, which is what the reference clips)."""
        if getattr(self, "_clip_scratch", None) is None:
            self._clip_scratch = torch.zeros(1026, dtype=torch.float32, device=self.book.grads.device)
        ops.clip_grad_norm(self.book.grads, max_norm, self._clip_scratch, grad_scale=grad_scale)
        return self._clip_scratch[1025:1026]  # the norm (of the scaled gradient) before clipping (device scalar)

    def optimizer_step(self):
        m = self.model
        with torch.no_grad():
            grad_scale, norm = 1.0 / self.world, None
            if self.max_grad_norm is not None:
                norm = self.clip_grad_norm_(self.max_grad_norm, grad_scale=grad_scale)  # mean first, then clip
                grad_scale = 1.0
            self.step_count += 1
            if self.device_step and not self.skip_nonfinite:
                # train.py's loop has no finite-loss test: the gate sees a constant, only the device-side step counter
                # (bias corrections that follow the replay count) is used
                ops.adamw_gated(self.book.params, self.book.grads, self.m, self.v, self.lr, self.betas[0], self.betas[1],
                                self.eps, self.wd, self._finite, self._ctl, grad_scale=grad_scale, grad_norm=None)
            elif self.skip_nonfinite:
                if self._last_loss is None:
                    raise RuntimeError("optimizer_step(skip_nonfinite=True) needs the loss of forward_backward()")
                ops.adamw_gated(self.book.params, self.book.grads, self.m, self.v, self.lr, self.betas[0], self.betas[1],
                                self.eps, self.wd, self._last_loss.reshape(1), self._ctl, grad_scale=grad_scale,
                                grad_norm=norm)
            else:
                ops.adamw(self.book.params, self.book.grads, self.m, self.v, self.lr, self.betas[0], self.betas[1],
                          self.eps, self.wd, self.step_count, grad_scale=grad_scale)
            # bf16 shadows / stacked copies of the trainable weights are stale now
            m.ltsf._invalidate()
            if self.lora_trainable:
                m.mllm.llama_wrapper.refresh_lora(self._lora_stacked)
            if self.train_mllm_front:
                m.mllm.qformer._invalidate()
                m.mllm._invalidate()

    def prefetch(self, vision_embs, ready=None):
        """Optional: start the frozen Q-Former of the next batch underneath the step in flight (model.prefetch)."""
        self.model.prefetch(vision_embs, ready=ready)

    def capture(self, *args, **kw):
        """One whole step (zero_grad + forward + backward + AdamW) as a hipGraph: returns (graph, (loss, decoded)) with
        static result tensors; every graph.replay() is one step on the CURRENT contents of the argument tensors.
        Host-side per-step state moves to the device: the optimizer's step count (tcavt_adamw_gated's counter) and, in
        train mode, a dropout epoch that the first node of the graph advances and every mask generator adds to its seed
        (tcavt_set_dropout_epoch), so each replay draws fresh masks and its backward regenerates the same ones.
        Single process only (the gradient all-reduce is not captured); call release_graph() when done."""
        if self.world != 1:
            raise RuntimeError("Trainer.capture: data-parallel steps launch their all-reduces eagerly")
        if self.device_step is False and self.step_count > 0 and not self.skip_nonfinite:
            self._ctl[0] = self.step_count  # continue the bias-correction count of the eager steps taken so far
        self.device_step = True
        self.model.pipeline_decoder = False  # (one graph = one step: nothing to overlap across replays)
        if self.model.training:
            self._epoch = torch.zeros(1, dtype=torch.int64, device=self.book.grads.device)
            ops.set_dropout_epoch(self._epoch)
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):  # warm-up under the final configuration: allocations, packed weights, kernel attributes
            self.step(*args, **kw)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph):
                if self._epoch is not None:
                    ops.dropout_epoch_advance(self._epoch)
                out = self.step(*args, **kw)
        finally:
            # The captured kernels carry the epoch's address in their (by-value) arguments; the library-wide pointer is only
            # needed while launches are being recorded.  Clearing it here means no later eager launch reads the epoch, and a
            # Trainer that is dropped without release_graph() leaves no dangling pointer behind in the library.
            if self._epoch is not None:
                ops.set_dropout_epoch(None)
        graph._tcavt_epoch = self._epoch  # the graph's kernels read it on every replay: it lives as long as the graph
        return graph, out

    def release_graph(self):
        """Forget the captured step's device-side dropout epoch (the graph object keeps its own reference)."""
        self._epoch = None

    def optimizer_counters(self):
        """(applied, skipped) updates of the gated optimizer (one host sync)."""
        c = self._ctl.tolist()
        return (c[0], c[1]) if (self.skip_nonfinite or self.device_step) else (self.step_count, 0)

    def enable_diagnostics(self, on=True):
        """Data-parallel self-diagnosis (bench.py, world > 1): record HIP events around every bucket's all-reduce."""
        self.diag = [] if on else None

    def diagnostics(self):
        """-> per-bucket {first element, bytes, launches, mean / max ms the exchange held the launching stream} of the steps run
        since enable_diagnostics() (call after a device synchronisation), and the number of HIP streams this rank uses."""
        from . import streams as S

        buckets = {}
        for lo, hi, e0, e1 in (self.diag or []):
            b = buckets.setdefault((lo, hi), [])
            b.append(e0.elapsed_time(e1))
        m = self.model
        n_streams = 1 + (S._ACTIVE or S.N_SLOTS) + (1 if getattr(m, "pipeline_decoder", False) else 0) + (1 if (self.world > 1 or self._force_dp) else 0)
        return {
            "buckets": [{"first_element": lo, "bytes": 4 * (hi - lo), "launches": len(v), "mean_ms": round(sum(v) / len(v), 4),
                         "max_ms": round(max(v), 4)} for (lo, hi), v in sorted(buckets.items())],
            "hip_streams_in_use": n_streams,
            "hip_streams_note": "caller's stream + side-channel pool" + (" + MLLM stream" if getattr(m, "pipeline_decoder", False) else "")
                                + (" + the process group's RCCL stream" if (self.world > 1 or self._force_dp) else ""),
        }

    def check_flags(self):
        """Raise if any forward since the last check saw input_ids outside the vocabulary, a mask that is not a right-padded
        prefix or a 16-bit value outside its range (the kernels only SET the device flags; one host sync here) -- and WARN when the
        gated optimizer has skipped updates since the last check (modify_train.py:1190-1196 skips a step on a non-finite loss;
        here a non-finite gradient norm does the same, e.g. an fp16 gradient that left the half range under the step's
        power-of-two scale: every step skipped is a step not trained, and nothing else would say so)."""
        self.model.mllm.check_flags()
        if self.skip_nonfinite or self.device_step:
            applied, skipped = self.optimizer_counters()
            if skipped > self._skipped_seen:
                import warnings

                warnings.warn(f"Trainer: {skipped - self._skipped_seen} optimizer update(s) skipped since the last check "
                              f"({skipped} of {applied + skipped} in all): non-finite loss or gradient norm -- with fp16 storage try "
                              "model.set_storage(torch.bfloat16) (bf16 tapes and gradients: no range limit, 8 significant bits)",
                              RuntimeWarning, stacklevel=2)
                self._skipped_seen = skipped

    def step(self, *args, **kw):
        out = self.forward_backward(*args, **kw)
        self.optimizer_step()
        if self.check_flags_every > 0 and self.step_count % self.check_flags_every == 0:
            self.check_flags()
        return out
