"""Orchestrator with the reference's `CosyVoice3Model` surface
(CosyVoice/cosyvoice/cli/model.py:324-441): tts(**model_input) -> generator of
{'tts_speech': FloatTensor (1, S) on CPU}; plus `tts_batch`, the batched entry the
reference lacks (it is strictly batch 1: flow/flow.py:369).

All three stages run in libfy_cosy3 (HIP); this file only moves pointers around.
"""
from __future__ import annotations

import os
import threading
from typing import Dict, Iterable, List, Optional, Sequence

import torch

from ..flow import FlowEngine
from ..hift import HiftEngine
from ..llm import LlmEngine
from ..spec import ModelCfg, SAMPLE_RATE


class _Lane:
    """One set of single-threaded engine handles (+ the stream its calls run on; None = the caller's current stream)."""

    def __init__(self, llm, flow, hift, stream):
        self.llm, self.flow, self.hift, self.stream = llm, flow, hift, stream
        self.uniforms = None


class _LmAhead:
    """The LM of one stream=True generation on a host thread and HIP stream of its own (cli/model.py:104-129 llm_job): steps the lane's
    LM handle in slices until the generation ends and publishes (tokens kept, done) after each."""
    SLICE = 8                                               # fy_llm_step reads the stop flags every 8 steps anyway

    def __init__(self, model, ln, n0: int, first_need: int):
        import threading
        self.n, self.done, self.err, self._stop = n0, False, None, False
        self.cv = threading.Condition()
        if getattr(ln, "lm_stream", None) is None:
            ln.lm_stream = torch.cuda.Stream(device=model.device)
        dev, st = model.device, ln.lm_stream
        st.wait_stream(ln.stream if ln.stream is not None else torch.cuda.current_stream(dev))   # (begin has completed: it returned the first count)

        def job():
            try:
                with torch.inference_mode(), torch.cuda.device(dev), torch.cuda.stream(st):
                    n, done, k = self.n, False, max(1, first_need - n0)
                    while not done and not self._stop:
                        (n,), (done,) = ln.llm.step(k)
                        with self.cv:
                            self.n, self.done = n, done
                            self.cv.notify_all()
                        k = self.SLICE
            except BaseException as e:                      # surfaces in the consumer's thread
                with self.cv:
                    self.err, self.done = e, True
                    self.cv.notify_all()
        self.th = threading.Thread(target=job, name="fy-lm-ahead", daemon=True)
        self.th.start()

    def wait_for(self, need: int):
        with self.cv:
            self.cv.wait_for(lambda: self.n >= need or self.done)
            if self.err is not None:
                raise self.err
            return self.n, self.done

    def stop(self):
        self._stop = True
        self.th.join()


class CosyVoice3Model:
    def __init__(self, llm_weights: Dict[str, torch.Tensor], flow_weights: Dict[str, torch.Tensor],
                 hift_weights: Dict[str, torch.Tensor], cfg: ModelCfg = ModelCfg(), device: Optional[torch.device] = None,
                 max_batch: int = 8, max_text: int = 128, max_prompt_tokens: int = 800, max_tokens: int = 800,
                 rand_noise: Optional[torch.Tensor] = None, rand_ini: Optional[torch.Tensor] = None,
                 sine_noise: Optional[torch.Tensor] = None, fp16: bool = False, keep_llm_weights: bool = False, n_llm: int = 1,
                 sampler: str = "greedy", sampler_seed: int = 1986, lm_group: int = 1, concurrency: int = 1, flow_workers: int = 1,
                 flow_group: int = 1, cache_prompts: bool = True, incremental_stream: bool = True, llm_weight_planes: int = 0,
                 stream_lm_ahead: bool = True):
        if not torch.cuda.is_available():
            raise RuntimeError("fangyan_tts_amd needs an AMD GPU (ROCm); there is no CPU path")
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.cfg, self.fp16 = cfg, fp16
        # cache_prompts: keep the padded device copies of the prompts a call presented (tokens, mel, x-vector), keyed by the
        # identity and in-place version counter of the caller's tensors - a serving loop presents the same prompts again and
        # again.  The key cannot see a tensor rewritten behind torch's back (`.data` assignment, memory shared with numpy, a
        # kernel writing through data_ptr()): callers that do that, or that never repeat a prompt, pass False - every call then
        # pads and copies its prompts afresh.
        self.cache_prompts = bool(cache_prompts)
        # stream=True: the intermediate chunks push only their NEW mel rows through the DiT blocks, against the keys / values the
        # earlier chunks left per (Euler step, block) (csrc/flow.hip: FY_INCREMENTAL) - exact under the chunk mask, because the
        # reference's schedule ends every chunk on a mask boundary; the reference itself (and False here) re-runs the flow decoder
        # over everything so far for every chunk (cli/model.py:339-369).  The last call (no mask) is always a full pass.
        self.incremental_stream = bool(incremental_stream)
        # stream=True: the LM keeps generating on a stream and host thread of its own while the chunks are decoded (the reference's
        # llm_job thread, cli/model.py:104-129, 339-358); False: the LM is stepped on demand between two chunks, on the chunk's stream.
        self.stream_lm_ahead = bool(stream_lm_ahead) and os.environ.get("FY_STREAM_LM_AHEAD", "1") != "0"
        self.max_batch, self.max_tokens, self.max_prompt_tokens = max_batch, max_tokens, max_prompt_tokens
        max_frames = 2 * (max_tokens + max_prompt_tokens)
        # n_llm > 1: extra LM handles (own KV cache and workspace) so tts_pipeline can decode several batches at once;
        # one decode stream uses a few dozen workgroups per launch and is bound by launch latency, not by the chip
        # lm_group > 1: tts_pipeline decodes that many consecutive batches in ONE LM call (the weights stream once per
        # decode step for all of them and the launch count per batch drops), then runs flow + vocoder batch by batch
        self.lm_group = max(1, lm_group)
        # llm_weight_planes: how the LM stores its matrices (llm.py: LlmEngine) - 0 = two bf16 planes (w = hi + lo, ids track the fp32
        # reference on a general checkpoint) exactly when llm.pt is not bf16-representable, else one; 1 / 2 force it
        self.llms = [LlmEngine(llm_weights, cfg.llm, max_batch=max_batch * self.lm_group, max_ctx=2 + max_text + max_prompt_tokens + max_tokens,
                               device=self.device, keep_weights=keep_llm_weights and i == 0, weight_planes=llm_weight_planes)
                     for i in range(max(1, n_llm))]
        self.llm = self.llms[0]
        # LM handles decoding beside the flow decoder (tts_pipeline) use the one-launch-per-operation decode: its short
        # kernels interleave with the other streams, while the persistent step holds 152 CUs for its whole duration
        # (measured at batch 8: 86.6 ms per pipelined step against 102).  tts, tts_batch and stream=True use the persistent
        # step for up to 8 sequences: lowest latency.
        # tts_pipeline switches its LM handles to the per-operation path for its duration (below): a persistent grid needs
        # 152 CUs to itself, which it never has beside the flow stream or on a CU-masked stream.
        # flow_group > 1: tts_pipeline hands the flow decoder + vocoder that many consecutive batches in ONE call (their utterances
        # side by side in one ragged batch): a DiT product over more rows wastes less of its last round of tiles (DESIGN.md
        # section 10: 9.65 us per sequence and block at 16 sequences, 8.76 at 32)
        self.flow_group = max(1, flow_group)
        fb = max_batch * self.flow_group
        self.flow = FlowEngine(flow_weights, cfg.flow, max_batch=fb, max_frames=max_frames, device=self.device)
        self.hift = HiftEngine(hift_weights, cfg.hift, max_batch=fb, max_frames=2 * max_tokens, device=self.device)
        # The reference lets several threads enter tts() on one model object (a gRPC pool, runtime/python/grpc/server.py:68-69;
        # per-call state keyed by uuid, cli/model.py:330-333).  An engine handle is single-threaded, so concurrent calls each
        # take a LANE - an own (LM, flow, vocoder) handle set and stream; `concurrency` lanes exist (the reference's
        # trt_concurrent plays this role for its estimator contexts, cli/model.py:94-99).  With one lane calls take turns.
        import queue as _q
        self.lanes = [_Lane(self.llm, self.flow, self.hift, None)]
        for _ in range(1, max(1, concurrency)):
            self.lanes.append(_Lane(
                LlmEngine(llm_weights, cfg.llm, max_batch=max_batch, max_ctx=2 + max_text + max_prompt_tokens + max_tokens, device=self.device,
                          weight_planes=llm_weight_planes),
                FlowEngine(flow_weights, cfg.flow, max_batch=max_batch, max_frames=max_frames, device=self.device),
                HiftEngine(hift_weights, cfg.hift, max_batch=max_batch, max_frames=2 * max_tokens, device=self.device),
                torch.cuda.Stream(device=self.device)))
        self._free = _q.Queue()
        for ln in self.lanes:
            self._free.put(ln)
        # flow_workers > 1: tts_pipeline runs the flow decoder + vocoder of that many consecutive batches at once, each on its own
        # (flow, vocoder) handle pair and stream: one batch's launch tails, epilogue bursts and whole-tile round-up (a third of a DiT
        # product's time) overlap the other batch's K loops
        self._flow_sets = [_Lane(None, self.flow, self.hift, None)]
        for _ in range(1, max(1, flow_workers)):
            self._flow_sets.append(_Lane(None, FlowEngine(flow_weights, cfg.flow, max_batch=fb, max_frames=max_frames, device=self.device),
                                         HiftEngine(hift_weights, cfg.hift, max_batch=fb, max_frames=2 * max_tokens, device=self.device), None))
        # `model.llm.load_state_dict` (compare_inference.py:42 swaps a fine-tuned LM in) must reach every LM handle
        self.llm._peers = [e for e in self.llms[1:]] + [ln.llm for ln in self.lanes[1:]]
        self._count_mu = threading.Lock()
        # The reference draws these buffers once at construction and never stores them in a checkpoint
        # (flow_matching.py:199-200; generator.py:223-226); pass them in to reproduce a given instance.
        g = torch.Generator().manual_seed(0)
        self.rand_noise = (rand_noise if rand_noise is not None else torch.randn(1, 80, max_frames, generator=g)).to(self.device)
        self.rand_ini = (rand_ini if rand_ini is not None else torch.rand(1, 9, generator=g)).to(self.device)
        n = 2 * max_tokens * cfg.hift.upsample_total
        self.sine_noise = (sine_noise if sine_noise is not None else torch.rand(1, n, 9, generator=g)).to(self.device).contiguous()
        self.token_hop_len = 25
        # precision modes of the two back stages (include/fy_cosy3.h): 0 = bf16 MFMA operands (default); FY_PRECISE = fp32-class
        # (split operands, fp32 attention); for the vocoder also FY_DIRECT = exact fp32 convolutions.  Verification modes.
        self.flow_flags, self.hift_flags = 0, 0
        self.lock = threading.Lock()          # the engines' handles are single-threaded
        # "greedy": the deterministic rule of SURVEY 8 a4.  "ras": the reference's default repetition-aware sampling; its
        # multinomial draws come from fresh uniforms (one row per sequence slot) drawn before every LM call
        self.sampler, self.sampler_seed = sampler, sampler_seed
        self._uniforms = [torch.zeros(max_batch * self.lm_group, 4 * max_tokens + 256, device=self.device) for _ in self.llms] if sampler == "ras" else None
        self._n_batches = 0                   # batches decoded so far: batch k draws from a generator seeded (sampler_seed, k)
        self.max_text = max_text
        assert sampler in ("greedy", "ras")

    def _arm_lane_sampler(self, ln, k: int, nb: int):
        if self.sampler == "ras":
            if ln.uniforms is None:
                ln.uniforms = torch.zeros(self.max_batch, 4 * self.max_tokens + 256, device=self.device)
            g = torch.Generator(device=self.device).manual_seed((self.sampler_seed << 20) + int(k))
            ln.uniforms[:nb].uniform_(0.0, 1.0, generator=g)
            ln.llm.set_sampler("ras", ln.uniforms)

    def _arm_sampler(self, i: int = 0, batch_ids: Sequence[int] = (0,), sizes: Optional[Sequence[int]] = None):
        """Uniforms for the LM call of handle i that decodes the batches `batch_ids` (global batch counters): each batch's rows
        come from its own generator seeded (sampler_seed, batch counter), so what a batch draws does not depend on which
        handle, thread or call decodes it - tts_pipeline reproduces a sequence of tts_batch calls under "ras" too."""
        if self.sampler == "ras":
            u, o = self._uniforms[i], 0
            for j, k in enumerate(batch_ids):             # batch j of the call occupies the sequence slots o .. o + its size
                nb = self.max_batch if sizes is None else int(sizes[j])
                g = torch.Generator(device=self.device).manual_seed((self.sampler_seed << 20) + int(k))
                u[o: o + nb].uniform_(0.0, 1.0, generator=g)
                o += nb
            self.llms[i].set_sampler("ras", u)

    def _acquire_lane(self):
        """A free lane; waits while every lane is busy - unless THIS thread is what keeps them busy (a stream=True generator it
        has not exhausted or closed), which would wait forever: that is an error, said so."""
        import queue as _q
        me = threading.get_ident()
        holders = self.__dict__.setdefault("_lane_holders", {})
        while True:
            try:
                ln = self._free.get(timeout=0.2)
                break
            except _q.Empty:
                with self._count_mu:
                    mine = sum(1 for h in holders.values() if h == me)
                if mine == len(self.lanes):
                    raise RuntimeError("every lane of this model is held by this thread (a tts(stream=True) generator that was neither "
                                       "exhausted nor closed): finish or close() it first, or build the model with concurrency > 1")
        with self._count_mu:
            holders[id(ln)] = me
        return ln

    def _release_lane(self, ln):
        with self._count_mu:
            self.__dict__.setdefault("_lane_holders", {}).pop(id(ln), None)
        self._free.put(ln)

    def _lane_stream(self, ln):
        """The lane's stream as the current stream for the engine calls inside (lanes after the first have their own)."""
        import contextlib
        from .. import _lib as _l0s

        @contextlib.contextmanager
        def cm():
            if ln.stream is None:
                yield
            else:
                ln.stream.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(ln.stream):
                    yield
                _l0s.stream_wait(ln.stream)
        return cm()

    def _take_lane(self):
        import contextlib

        @contextlib.contextmanager
        def cm():
            ln = self._acquire_lane()
            try:
                with self._lane_stream(ln):
                    yield ln
            finally:
                self._release_lane(ln)
        return cm()

    def _take_all_lanes(self):
        import contextlib

        @contextlib.contextmanager
        def cm():
            got = []
            try:
                for _ in self.lanes:                           # the pipeline owns every handle while it runs
                    got.append(self._acquire_lane())
                yield
            finally:
                for ln in got:
                    self._release_lane(ln)
        return cm()

    def _next_batch_ids(self, n: int = 1):
        with self._count_mu:
            k = self._n_batches
            self._n_batches += n
        return k

    def _check_capacity(self, inputs, max_len):
        """The engines were sized at construction; say so up front instead of failing after the whole LM decode."""
        z = torch.zeros(1, 0, dtype=torch.int32)
        for b, d in enumerate(inputs):
            n_text = d["text"].reshape(-1).shape[0] + d.get("prompt_text", z).reshape(-1).shape[0]
            n_ps = d.get("llm_prompt_speech_token", z).reshape(-1).shape[0]
            n_fp = d["flow_prompt_speech_token"].reshape(-1).shape[0]
            cap = int(max_len[b]) if max_len is not None else 20 * d["text"].reshape(-1).shape[0]
            if n_text > self.max_text:
                raise ValueError(f"utterance {b}: {n_text} text + prompt-text ids, the model was built for max_text={self.max_text}")
            if max(n_ps, n_fp) > self.max_prompt_tokens:
                raise ValueError(f"utterance {b}: {max(n_ps, n_fp)} prompt speech tokens, the model was built for "
                                 f"max_prompt_tokens={self.max_prompt_tokens}")
            if cap > self.max_tokens:
                raise ValueError(f"utterance {b}: up to {cap} speech tokens (20 x the text length, llm.py:744), the model was built "
                                 f"for max_tokens={self.max_tokens}; pass max_len or build with a larger max_tokens")

    # ------------------------------------------------------------------ batched path
    @torch.inference_mode()
    def tts_batch(self, inputs: Sequence[Dict[str, torch.Tensor]], min_len: Optional[Sequence[int]] = None,
                  max_len: Optional[Sequence[int]] = None, speed: float = 1.0, keep_on_device: bool = False):
        """inputs: dicts with the reference's model_input keys (cli/frontend.py:168-213).
        Returns (wav (B, Smax) fp32, n_samples list, tokens list)."""
        B = len(inputs)
        assert 1 <= B <= self.max_batch
        z = torch.zeros(1, 0, dtype=torch.int32)
        text = [d["text"].reshape(-1).tolist() for d in inputs]
        ptext = [d.get("prompt_text", z).reshape(-1).tolist() for d in inputs]
        pspeech = [d.get("llm_prompt_speech_token", z).reshape(-1).tolist() for d in inputs]
        self._check_capacity(inputs, max_len)
        with self._take_lane() as ln:
            self._arm_lane_sampler(ln, self._next_batch_ids(), B)
            out, out_n, _ = ln.llm.generate(text, ptext, pspeech, min_len=min_len, max_len=max_len)
            n_tok = out_n.cpu().tolist()
            if min(n_tok) < 1:
                raise RuntimeError("the language model emitted no speech token for an utterance")
            wav, samples = self._token2wav(inputs, out, n_tok, speed, ln)
            res = wav if keep_on_device else wav.cpu()
        toks = [out[b, : n_tok[b]] for b in range(B)]
        return res, samples, toks

    # ------------------------------------------------------------------ pipelined batches
    @torch.inference_mode()
    def tts_pipeline(self, batches: Sequence[Sequence[Dict[str, torch.Tensor]]], min_len=None, max_len=None,
                     keep_on_device: bool = False, flow_cu_exclude: Optional[int] = None, lm_isolate: bool = False):
        """Consecutive batches, software-pipelined over HIP streams: the speech-token LM of the next batches
        (latency-bound, a few workgroups per launch; one stream per LM handle, `n_llm` of them) runs beside the
        flow decoder + vocoder of batch i (throughput-bound).  Yields (wav, n_samples, tokens) per batch, in
        order.  Same results as tts_batch.

        flow_cu_exclude: CUs the flow / vocoder stream may NOT use (hipExtStreamCreateWithCUMask); None = 0 = no mask.  Measured on
        MI355X at batch 8 (round 3, one LM handle decoding four batches per call): 60.6 ms per step with no mask, 70.6 / 74.5 / 75.1
        with 8 / 16 / 32 CUs kept clear, 82.7 with 80 - a masked stream is slower by itself and the LM's short kernels do not wait
        for whole CUs.  lm_isolate additionally confines the LM streams to the excluded CUs (complementary masks); measured slower
        at every split (round 2: 126-197 ms).
        Set FY_PIPE_TRACE=1 for per-stage wall times on stderr."""
        import os
        import queue
        import sys
        import threading as th
        import time
        from .. import _lib as _lib_w
        trace = bool(os.environ.get("FY_PIPE_TRACE"))        # per-stage wall times on stderr
        dev = self.device
        n_prod = len(self.llms)
        if flow_cu_exclude is None:
            flow_cu_exclude = 0
        n_fl = len(self._flow_sets)
        assert n_fl == 1 or not flow_cu_exclude, "a CU-masked flow stream exists once: use flow_workers=1 with flow_cu_exclude"
        pool = self._pipe_streams(n_fl + n_prod, flow_cu_exclude)
        flow_streams, lm_streams = pool[:n_fl], pool[n_fl:]
        qs = [queue.Queue(maxsize=2 * self.lm_group) for _ in range(n_prod)]
        z = torch.zeros(1, 0, dtype=torch.int32)
        stop, box = th.Event(), {}
        for bi, inputs in enumerate(batches):
            self._check_capacity(inputs, max_len[bi] if max_len is not None else None)

        def put(q, item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def producer(pi):
            llm, q = self.llms[pi], qs[pi]
            try:
                lm_stream = self._masked_stream(flow_cu_exclude, only=True, tag=pi) if (flow_cu_exclude > 0 and lm_isolate) else lm_streams[pi]
                with torch.cuda.device(dev), torch.cuda.stream(lm_stream):
                    G = self.lm_group
                    for g0 in range(pi * G, len(batches), n_prod * G):
                        if stop.is_set():
                            return
                        group = list(range(g0, min(g0 + G, len(batches))))
                        text, ptext, pspeech, mn, mx = [], [], [], [], []
                        for bi in group:
                            inputs = batches[bi]
                            text += [d["text"].reshape(-1).tolist() for d in inputs]
                            ptext += [d.get("prompt_text", z).reshape(-1).tolist() for d in inputs]
                            pspeech += [d.get("llm_prompt_speech_token", z).reshape(-1).tolist() for d in inputs]
                            mn += list(min_len[bi]) if min_len is not None else [int(len(d["text"].reshape(-1)) * 2) for d in inputs]
                            mx += list(max_len[bi]) if max_len is not None else [int(len(d["text"].reshape(-1)) * 20) for d in inputs]
                        self._arm_sampler(pi, [box["base"] + bi for bi in group], [len(batches[bi]) for bi in group])
                        t0 = time.perf_counter()
                        out, out_n, _ = llm.generate(text, ptext, pspeech, min_len=mn, max_len=mx)
                        _lib_w.stream_wait(lm_stream)             # the ids are complete (polled with sleeps: no core spins on the LM stream)
                        n_tok = out_n.cpu().tolist()
                        if trace:
                            print(f"[pipe] LM handle {pi} batches {group}: {1e3 * (time.perf_counter() - t0):.1f} ms", file=sys.stderr)
                        o = 0
                        for bi in group:
                            nb = len(batches[bi])
                            if not put(q, (batches[bi], out[o: o + nb], n_tok[o: o + nb])):
                                return
                            o += nb
            except BaseException as e:                             # surfaces in the consumer
                put(q, e)

        n_flow = len(self._flow_sets)
        work_q = [queue.Queue() for _ in range(n_flow)]
        res_q = [queue.Queue(maxsize=2 * self.flow_group) for _ in range(n_flow)]

        def take(q):
            while not stop.is_set():
                try:
                    return q.get(timeout=0.05)
                except queue.Empty:
                    pass
            return None

        def feeder():
            # ids arrive in batch order from the producers' queues; batch bi goes to flow worker bi % n_flow
            for bi in range(len(batches)):
                item = take(qs[(bi // self.lm_group) % n_prod])
                if item is None:
                    return
                work_q[bi % n_flow].put((bi, item, time.perf_counter()))
                if isinstance(item, BaseException):
                    return

        def worker(w):
            fs, st = self._flow_sets[w], flow_streams[w]
            try:
                with torch.cuda.device(dev):
                    mine = list(range(w, len(batches), n_flow))
                    FG = self.flow_group
                    for g0 in range(0, len(mine), FG):
                        parts = []
                        for bi in mine[g0: g0 + FG]:               # this worker's next flow_group batches, one ragged batch
                            got = take(work_q[w])
                            if got is None:
                                return
                            _, item, t1 = got
                            if isinstance(item, BaseException):
                                res_q[w].put(item)
                                return
                            if min(item[2]) < 1:
                                raise RuntimeError("the language model emitted no speech token for an utterance")
                            parts.append(item)
                        sizes = [len(p[0]) for p in parts]
                        inputs = [d for p in parts for d in p[0]]
                        n_tok = [n for p in parts for n in p[2]]
                        ld = max(p[1].shape[1] for p in parts)
                        out = parts[0][1] if len(parts) == 1 else torch.cat(
                            [torch.nn.functional.pad(p[1], (0, ld - p[1].shape[1])) for p in parts], dim=0)
                        bi = mine[g0]
                        # The worker waits for its batch before it takes the next one.  Measured against running ahead (the batch's
                        # completion as an event for the consumer's stream, the next batch enqueued behind it at once - possible
                        # since neither engine call synchronises the stream any more: their length tables ride in kernel
                        # arguments): 61.4 ms per step with the wait against 62.7 without (one worker), 59.4 against 59.6 (two).
                        # (Holes of 8-13 ms that rocprofv3 --kernel-trace shows on the flow queue between two batches stay when
                        # the launches are enqueued ahead: the profiler's stream placement, not the host - DESIGN.md section 10.)
                        with torch.cuda.stream(st):
                            wav, samples = self._token2wav(inputs, out, n_tok, 1.0, fs)
                            mel, frames = self.last_mel_of[id(fs)]
                            t2 = time.perf_counter()
                            if keep_on_device:
                                res, ev = wav, torch.cuda.Event()
                                ev.record(st)
                                _lib_w.stream_wait(st)
                            else:
                                res, ev = wav.cpu(), None
                        if trace:
                            print(f"[pipe] batch {bi} (flow worker {w}): launches enqueued {1e3 * (t2 - t1):.1f} ms after its ids arrived",
                                  file=sys.stderr)
                        o = 0
                        for nb in sizes:                           # handed out batch by batch, in order
                            r = (res[o: o + nb], samples[o: o + nb], [out[b, : n_tok[b]] for b in range(o, o + nb)], mel[o: o + nb],
                                 frames[o: o + nb], ev)
                            if not put(res_q[w], r):
                                return
                            o += nb
            except BaseException as e:                            # surfaces in the consumer
                res_q[w].put(e)

        threads = [th.Thread(target=producer, args=(i,), daemon=True) for i in range(n_prod)]
        threads += [th.Thread(target=feeder, daemon=True)] + [th.Thread(target=worker, args=(w,), daemon=True) for w in range(n_flow)]
        with self._take_all_lanes():
            was_persistent = [e.decode_mode for e in self.llms]
            # An LM call of 9 .. 32 sequences (lm_group batches) decodes with the few-CU persistent step (csrc/llm_decode32.hip: one launch
            # of 76 resident workgroups per token step): beside the flow stream a call takes 132 ms where the per-operation launches
            # took 266 (FY_PIPE_TRACE) - the LM is no longer what the pipeline waits for.  Decode mode 2 = ONLY that step: a call of up
            # to 8 sequences (a tail group, a warm-up) runs it too and never the 8-row persistent step, whose 152 workgroups would wait
            # for residency beside the flow streams.  FY_PIPE_LM_PERSISTENT32=0: always per-operation.
            import os as _os
            p32 = bool(int(_os.environ.get("FY_PIPE_LM_PERSISTENT32", "1")))
            for e in self.llms:
                e.set_decode_mode(2 if p32 else 0)
            # For the pipeline's duration the host threads that wait for the device - the LM producers (inside fy_llm_step and for
            # their ids), the flow workers - poll with sleeps instead of spinning (fy_set_host_wait, _lib.stream_wait): they wait
            # most of the time, and spinning they cost 2.7 cores per rank = 22 on an 8-rank host (measured 138 -> 52 ms of
            # process CPU per 50 ms step at the same step time).  tts / tts_batch keep hipStreamSynchronize: one thread, and a
            # 75-token generation looks at its stop flags ten times (+4 ms with 200 us sleeps).  FY_HOST_WAIT_SPIN=1: spin here too.
            _lib_w.check(_lib_w.lib().fy_set_host_wait(0 if _lib_w.host_wait_spin() else 1, int(_lib_w.HOST_WAIT_SLEEP_S * 1e6)))
            box["base"] = self._next_batch_ids(len(batches))
            for t in threads:
                t.start()
            try:
                for bi in range(len(batches)):
                    r = None
                    while r is None:
                        try:
                            r = res_q[bi % n_flow].get(timeout=0.5)
                        except queue.Empty:
                            if not any(t.is_alive() for t in threads[n_prod + 1:]):
                                # a worker may have put its last result and ended between the time-out and the liveness test
                                try:
                                    r = res_q[bi % n_flow].get_nowait()
                                except queue.Empty:
                                    raise RuntimeError("tts_pipeline: the flow workers ended without a result")
                    if isinstance(r, BaseException):
                        raise r
                    res, samples, toks, mel, frames, ev = r
                    if ev is not None:                                 # the caller's stream waits for the batch, not the host
                        torch.cuda.current_stream(dev).wait_event(ev)
                    self.last_mel, self.last_frames = mel, frames      # of the batch being handed out
                    # yielded on the caller's thread and stream: its own torch work (an all-gather, a copy) stays on its stream
                    yield res, samples, toks
            finally:
                # an abandoned or failed generator must not leave producers inside the single-threaded LM handles (or workers inside
                # the flow handles): tell them to stop, empty the queues so none stays parked in put(), and join them before the lock
                # is released
                stop.set()
                for t in threads:
                    while t.is_alive():
                        for q in qs + work_q + res_q:
                            try:
                                while True:
                                    q.get_nowait()
                            except queue.Empty:
                                pass
                        t.join(timeout=0.05)
                for e, p in zip(self.llms, was_persistent):
                    e.set_decode_mode(p)
                _lib_w.lib().fy_set_host_wait(0, int(_lib_w.HOST_WAIT_SLEEP_S * 1e6))

    def prepare_pipeline(self, flow_cu_exclude: Optional[int] = None):
        """Place the pipeline's streams now (otherwise the first tts_pipeline call does it, ~0.1-0.4 s)."""
        n_prod = len(self.llms)
        if flow_cu_exclude is None:
            flow_cu_exclude = 0
        self._pipe_streams(len(self._flow_sets) + n_prod, flow_cu_exclude)

    def _pipe_streams(self, n: int, flow_exclude: int = 0):
        """The streams of the pipeline (flow + vocoder first, then one per LM handle), chosen once per model so that they are
        served by different hardware pipes: two busy streams on one hardware queue or pipe take turns instead of overlapping
        (15-60 % of the pipelined step on MI355X), and which queue a stream gets depends on the process's history, so the
        pairs are measured (`fy_stream_overlap`, ~0.1 s) on a pool of 8 candidates."""
        import ctypes
        import itertools
        import os
        cache = self.__dict__.setdefault("_pipe_stream_sets", {})
        key = (n, flow_exclude)
        if key in cache:
            return cache[key]
        pool = self.__dict__.setdefault("_pipe_stream_pool", [])
        with torch.cuda.device(self.device):
            while len(pool) < max(8, n):
                st = torch.cuda.Stream(device=self.device)
                with torch.cuda.stream(st):
                    torch.zeros(1, device=self.device)          # first use binds the stream to its hardware queue
                st.synchronize()
                pool.append(st)
            # a CU-masked flow stream (flow_exclude > 0) is a stream of its own making: candidates for it go in front
            masked = [self._masked_stream(flow_exclude, tag=100 + t) for t in range(4)] if flow_exclude > 0 else []
            pool = masked + pool
            m = len(pool)
            ptrs = (ctypes.c_void_p * m)(*[s.cuda_stream for s in pool])
            ratio = (ctypes.c_float * (m * m))()
            from .. import _lib
            _lib.check(_lib.lib().fy_stream_overlap(ptrs, m, ratio))
        clash = lambda i, j: ratio[i * m + j] > 1.5
        best = None
        nm = len(masked)
        cands = ((f,) + c for f in range(nm) for c in itertools.combinations(range(nm, m), n - 1)) if nm else itertools.combinations(range(m), n)
        for c in cands:                                          # fewest clashing pairs, first such subset in pool order
            k = sum(clash(i, j) for i, j in itertools.combinations(c, 2))
            if best is None or k < best[0]:
                best = (k, c)
            if k == 0:
                break
        self.pipe_stream_clashes = best[0]                       # 0 unless the device offers fewer independent queues than streams
        cache[key] = [pool[i] for i in best[1]]
        if os.environ.get("FY_PIPE_TRACE"):
            import sys
            print(f"[pipe] streams {best[1]} of {m}, {best[0]} clashing pairs; overlap ratios of stream 0: "
                  f"{[round(ratio[j], 2) for j in range(m)]}", file=sys.stderr)
        return cache[key]

    def _masked_stream(self, exclude: int, only: bool = False, tag: int = 0):
        """A HIP stream whose kernels may not run on the first `exclude` CUs of the mask - or, with only=True, may run on
        nothing but those (cached per (value, only, tag))."""
        cache = self.__dict__.setdefault("_masked_streams", {})
        key = (exclude, only, tag)
        if key not in cache:
            import ctypes
            from .. import _lib
            n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
            words = (ctypes.c_uint32 * ((n_cu + 31) // 32))(*([0] * ((n_cu + 31) // 32)))
            for i in range(n_cu):
                inside = i < min(exclude, n_cu - 1)
                if inside == only:
                    words[i // 32] |= 1 << (i % 32)
            st = ctypes.c_void_p()
            with torch.cuda.device(self.device):
                # created by the HIP runtime libfy_cosy3 (and torch) are bound to, not by a second copy loaded by name
                rc = _lib.lib().fy_stream_create_masked(ctypes.byref(st), words, len(words))
            if rc != 0:                                      # no CU masking on this device / runtime: an ordinary stream
                cache[key] = torch.cuda.Stream(device=self.device)
                return cache[key]
            self.__dict__.setdefault("_masked_raw", []).append(st.value)
            cache[key] = torch.cuda.ExternalStream(st.value, device=self.device)
        return cache[key]

    def close(self):
        """Release what the engines do not own themselves (the CU-masked streams)."""
        from .. import _lib
        for raw in self.__dict__.pop("_masked_raw", []):
            _lib.lib().fy_stream_destroy(raw)
        self.__dict__.pop("_masked_streams", None)
        self.__dict__.pop("_pipe_stream_sets", None)

    def _prompt_tensors(self, inputs):
        """The flow decoder's per-utterance prompt (tokens, mel, x-vector) padded into batch tensors ON THE DEVICE, built once
        per set of prompt tensors (a serving loop and the benchmark present the same prompts again and again; keyed by the
        tensors' identities and in-place version counters; the cache keeps the tensors alive)."""
        cache = self.__dict__.setdefault("_prompt_cache", {})
        key = tuple((id(t), t._version) for d in inputs for t in (d["flow_prompt_speech_token"], d["prompt_speech_feat"], d["flow_embedding"]))
        hit = cache.get(key) if self.cache_prompts else None
        if hit is not None:
            return hit[1]
        B = len(inputs)
        fp = [d["flow_prompt_speech_token"].reshape(-1) for d in inputs]
        pf = [d["prompt_speech_feat"].reshape(-1, 80) for d in inputs]
        Pmax, PMmax = max(max(len(t) for t in fp), 1), max(max(f.shape[0] for f in pf), 1)
        ptok = torch.zeros(B, Pmax, dtype=torch.int32)
        pfeat = torch.zeros(B, PMmax, 80)
        for b in range(B):
            ptok[b, : len(fp[b])] = fp[b].to(torch.int32)
            pfeat[b, : pf[b].shape[0]] = pf[b]
        emb = torch.cat([d["flow_embedding"].reshape(1, -1) for d in inputs]).float()
        val = (ptok.to(self.device), [len(t) for t in fp], pfeat.to(self.device), [f.shape[0] for f in pf], emb.to(self.device))
        torch.cuda.current_stream(self.device).synchronize()     # the copies are complete before another stream may use them
        if not self.cache_prompts:
            return val
        with self._count_mu:                                      # tts_pipeline's flow workers share the cache
            if len(cache) >= 16:
                cache.pop(next(iter(cache)), None)
            cache[key] = ([(d["flow_prompt_speech_token"], d["prompt_speech_feat"], d["flow_embedding"]) for d in inputs], val)
        return val

    def _token2wav(self, inputs, out, n_tok, speed, ln=None):
        ln = ln or self.lanes[0]
        B = len(inputs)
        ptok, n_fp, pfeat, n_pf, emb = self._prompt_tensors(inputs)
        mel = ln.flow.inference(out, n_tok, ptok, n_fp, pfeat, n_pf, emb, self.rand_noise, flags=self.flow_flags)
        frames = [2 * n for n in n_tok]
        if speed != 1.0:                                    # cli/model.py:435-437
            assert B == 1, "speed change only supports a single utterance"
            mel = ln.flow.speed(mel, speed)
            frames = [mel.shape[2]]
        wav, _ = ln.hift.inference(mel, self.rand_ini, self.sine_noise, frames=frames, flags=self.hift_flags)
        self.last_mel, self.last_frames = mel, frames        # kept for parity tests / debugging
        self.__dict__.setdefault("last_mel_of", {})[id(ln)] = (mel, frames)       # per handle set: tts_pipeline's flow workers run side by side
        return wav, [f * self.cfg.hift.upsample_total for f in frames]

    # ------------------------------------------------------------------ stream=True (cli/model.py:339-369, 416-441)
    token_hop_len = 25                                          # cli/model.py:401: "must match training static_chunk_size"

    @torch.inference_mode()
    def _tts_stream(self, d: Dict[str, torch.Tensor], source_tokens: Optional[torch.Tensor] = None):
        """The reference's chunk schedule: a chunk is cut whenever `hop + pre_lookahead` tokens beyond the offset exist (the
        first hop is padded so prompt + hop is a multiple of 25); every chunk re-runs the flow decoder over all tokens so far
        (chunk attention mask, finalize=False) and the vocoder over the whole mel so far (finalize=False), and yields the
        samples beyond those already yielded; the last call is finalize=True without the chunk mask.  The reference polls
        a list its LM thread fills; here the LM is advanced on demand (`LlmEngine.begin` / `step`) by as many tokens as the
        next chunk still lacks, so the first chunk leaves after hop + look-ahead tokens, not after the whole utterance.  The
        chunk boundaries depend only on the token count, so the chunks are the reference's."""
        z = torch.zeros(1, 0, dtype=torch.int32)
        text = [d["text"].reshape(-1).tolist()]
        ptext = [d.get("prompt_text", z).reshape(-1).tolist()]
        pspeech = [d.get("llm_prompt_speech_token", z).reshape(-1).tolist()]
        fp = d["flow_prompt_speech_token"].reshape(1, -1).to(torch.int32)
        pf = d["prompt_speech_feat"].reshape(1, -1, 80).float()
        emb = d["flow_embedding"].reshape(1, -1).float()
        n_fp, n_pf = fp.shape[1], pf.shape[1]
        ptok = fp if n_fp else torch.zeros(1, 1, dtype=torch.int32)
        pfeat = pf if n_pf else torch.zeros(1, 1, 80)
        look, hop0 = self.cfg.flow.pre_lookahead, self.token_hop_len
        up = self.cfg.hift.upsample_total
        self._check_capacity([d], None)
        # The lane is held from the first chunk to the last (the LM handle carries the generation), but its stream is the
        # current stream only INSIDE the engine calls: the consumer's own torch work between two chunks stays on the
        # consumer's stream.  An abandoned generator releases the lane when it is closed or collected (GeneratorExit).
        ln = self._acquire_lane()
        ahead, mode_was = None, None
        try:
            with self._lane_stream(ln):
                if source_tokens is None:
                    self._arm_lane_sampler(ln, self._next_batch_ids(), 1)
                    out, _, _ = ln.llm.begin(text, ptext, pspeech)
                    (n,), (done,) = ln.llm.step(0)
                else:                                           # vc_job (cli/model.py:131-133): the token list is given, complete
                    out = source_tokens.reshape(1, -1).to(self.device, torch.int32)
                    n, done = out.shape[1], True
            pad = -(-n_fp // hop0) * hop0 - n_fp
            offset, speech_offset, mel_all = 0, 0, None
            ln.flow.stream_reset()                          # the incremental flow calls of this stream start from nothing
            # The LM runs ahead of the chunks on its own stream and thread (the reference's llm_job): it publishes (tokens kept, done)
            # after every slice of steps - a slice ends with the ids complete in memory (fy_llm_step waits for its stream), so the
            # chunk's stream may read out[:, :n] without an event.  The first slice is what the first chunk lacks.
            if source_tokens is None and self.stream_lm_ahead and not done:
                if os.environ.get("FY_STREAM_LM_MODE"):    # A/B: the LM's decode form beside the chunks (0 per-operation, 1, 2 few-CU)
                    mode_was = ln.llm.decode_mode
                    ln.llm.set_decode_mode(int(os.environ["FY_STREAM_LM_MODE"]))
                ahead = _LmAhead(self, ln, n, offset + hop0 + pad + look)

            def token2wav(n_in, offset, mel_all, speech_offset, streaming, finalize):
                mel = ln.flow.inference(out[:, :n_in], [n_in], ptok, [n_fp], pfeat, [n_pf], emb, self.rand_noise,
                                        streaming=streaming, finalize=finalize, flags=self.flow_flags,
                                        incremental=self.incremental_stream and streaming and not finalize)
                valid = 2 * (n_in if finalize else n_in - look)
                mel = mel[:, :, 2 * offset: valid]
                mel_all = mel if mel_all is None else torch.cat([mel_all, mel], dim=2)
                wav, _ = ln.hift.inference(mel_all.contiguous(), self.rand_ini, self.sine_noise, finalize=finalize, flags=self.hift_flags)
                end = up * (mel_all.shape[2] if finalize else mel_all.shape[2] - 8)
                return wav[:, speech_offset:end].cpu(), mel_all, end

            while True:
                need = offset + (hop0 + pad if offset == 0 else hop0) + look
                if ahead is not None:
                    n, done = ahead.wait_for(need)
                with self._lane_stream(ln):
                    while n < need and not done:            # the silent-token filter may drop tokens: ask again until enough
                        (n,), (done,) = ln.llm.step(need - n)
                    if n < need:
                        break
                    wav, mel_all, speech_offset = token2wav(need, offset, mel_all, speech_offset, True, False)
                offset = need - look
                yield {"tts_speech": wav}
            if n < 1:
                raise RuntimeError("the language model emitted no speech token for an utterance")
            with self._lane_stream(ln):
                wav, mel_all, speech_offset = token2wav(n, offset, mel_all, speech_offset, False, True)
            self.last_mel, self.last_frames = mel_all, [mel_all.shape[2]]
            yield {"tts_speech": wav}
        finally:
            if ahead is not None:
                ahead.stop()                                # (an abandoned generator: the LM thread ends after its current slice)
                if mode_was is not None:
                    ln.llm.set_decode_mode(mode_was)
            self._release_lane(ln)

    # ------------------------------------------------------------------ reference-shaped path
    def tts(self, text=torch.zeros(1, 0, dtype=torch.int32), flow_embedding=torch.zeros(0, 192), llm_embedding=torch.zeros(0, 192),
            prompt_text=torch.zeros(1, 0, dtype=torch.int32), llm_prompt_speech_token=torch.zeros(1, 0, dtype=torch.int32),
            flow_prompt_speech_token=torch.zeros(1, 0, dtype=torch.int32), prompt_speech_feat=torch.zeros(1, 0, 80),
            source_speech_token=torch.zeros(1, 0, dtype=torch.int32), stream=False, speed=1.0, **kwargs):
        vc = source_speech_token.shape[1] != 0                  # cli/model.py:334-337: vc_job instead of llm_job, no LM
        if not isinstance(text, torch.Tensor):
            # A text *generator* sends the reference to Qwen2LM.inference_bistream (cli/model.py:104-111), which reads
            # self.llm_embedding (llm/llm.py:545-546) - an attribute CosyVoice3LM.__init__ (llm.py:641-668) never creates, so
            # for this model family the reference's LM thread dies with exactly this error and no audio is produced.
            raise AttributeError("'CosyVoice3LM' object has no attribute 'llm_embedding' (streaming text input, "
                                 "inference_bistream, does not work for CosyVoice3 in the reference either)")
        if stream:
            assert speed == 1.0, "speed change only support non-stream inference mode"      # cli/model.py:436
            yield from self._tts_stream(dict(text=text, prompt_text=prompt_text, llm_prompt_speech_token=llm_prompt_speech_token,
                                             flow_prompt_speech_token=flow_prompt_speech_token,
                                             prompt_speech_feat=prompt_speech_feat, flow_embedding=flow_embedding),
                                        source_tokens=source_speech_token if vc else None)
            return
        if vc:
            inp = dict(flow_prompt_speech_token=flow_prompt_speech_token, prompt_speech_feat=prompt_speech_feat, flow_embedding=flow_embedding)
            with self._take_lane() as ln:
                out = source_speech_token.reshape(1, -1).to(self.device, torch.int32)
                wav, samples = self._token2wav([inp], out, [out.shape[1]], speed, ln)
                res = wav.cpu()[:, : samples[0]]
            yield {"tts_speech": res}
            return
        wav, samples, _ = self.tts_batch([dict(text=text, prompt_text=prompt_text, llm_prompt_speech_token=llm_prompt_speech_token,
                                               flow_prompt_speech_token=flow_prompt_speech_token,
                                               prompt_speech_feat=prompt_speech_feat, flow_embedding=flow_embedding)], speed=speed)
        yield {"tts_speech": wav[:, : samples[0]]}
