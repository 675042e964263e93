#!/usr/bin/env python3
"""Mint golden fixtures by running the REFERENCE's own modules.

Runs only in the build container (needs /root/reference); the GPU box and the
test-suite only ever read the .npz files this writes.  Nothing of the
reference is copied: the fixtures hold inputs' recipe names, output digests
and small output tensors.

    python tests/golden/mint_goldens.py [--only hift,flow,llm,e2e] [--full]

What is imported from /root/reference/CosyVoice/cosyvoice:
  hifigan/generator.py, hifigan/f0_predictor.py, flow/flow.py,
  flow/flow_matching.py, flow/DiT/dit.py, llm/llm.py, cli/model.py.
Packages those files import that are absent from this image get oracle-only
sys.modules stubs: an empty `torchaudio`; `omegaconf.DictConfig`
as an attribute dict; matcha's `BASECFM.__init__` (12 lines of attribute
setup, third_party/Matcha-TTS/matcha/models/components/flow_matching.py:12-30);
and x-transformers' RotaryEmbedding/apply_rotary_pos_emb.  The last stub
carries arithmetic (the package is not installed and not vendored), so it is
bound to oracle.flow.rope_freqs/apply_rope: DiT goldens are conditional on that
restatement (a15, parity unpinned for that sub-step).
"""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from _digest import digest, pack  # noqa: E402
from fangyan_tts_amd import synth  # noqa: E402
from fangyan_tts_amd.spec import FlowCfg, HiftCfg, LlmCfg, ModelCfg  # noqa: E402
from oracle import flow as oflow  # noqa: E402

REF = "/root/reference/CosyVoice"


def install_stubs():
    import transformers  # noqa: F401  (must be imported before the torchaudio stub)
    sys.path.insert(0, REF)
    sys.modules["torchaudio"] = types.ModuleType("torchaudio")
    oc = types.ModuleType("omegaconf")

    class DictConfig(dict):
        def __init__(self, content=None, **kw):
            super().__init__(content or {})
            self.__dict__ = self
    oc.DictConfig = DictConfig
    sys.modules["omegaconf"] = oc
    for n in ("matcha", "matcha.models", "matcha.models.components", "matcha.models.components.flow_matching"):
        sys.modules[n] = types.ModuleType(n)

    class BASECFM(torch.nn.Module):
        def __init__(self, n_feats, cfm_params, n_spks=1, spk_emb_dim=128):
            super().__init__()
            self.n_feats, self.n_spks, self.spk_emb_dim = n_feats, n_spks, spk_emb_dim
            self.solver = cfm_params.solver
            self.sigma_min = cfm_params.sigma_min if hasattr(cfm_params, "sigma_min") else 1e-4
            self.estimator = None
    sys.modules["matcha.models.components.flow_matching"].BASECFM = BASECFM

    class RotaryEmbedding(torch.nn.Module):
        def __init__(self, dim):
            super().__init__()
            self.dim = dim

        def forward_from_seq_len(self, n):
            return oflow.rope_freqs(n, self.dim), 1.0

    def apply_rotary_pos_emb(t, freqs, scale=1):
        return oflow.apply_rope(t, freqs) * scale if scale != 1 else oflow.apply_rope(t, freqs)
    xt = types.ModuleType("x_transformers")
    xtx = types.ModuleType("x_transformers.x_transformers")
    xtx.RotaryEmbedding, xtx.apply_rotary_pos_emb = RotaryEmbedding, apply_rotary_pos_emb
    sys.modules["x_transformers"], sys.modules["x_transformers.x_transformers"] = xt, xtx


def check_manifest(module, manifest, what):
    sd = module.state_dict()
    ref = {k: tuple(v.shape) for k, v in sd.items()}
    mine = {k: tuple(v) for k, v in manifest.items()}
    assert list(ref.keys()) == list(mine.keys()), (what, set(ref) ^ set(mine))
    assert ref == mine, what
    print(f"[manifest] {what}: {len(ref)} tensors, {sum(v.numel() for v in sd.values())} params - identical")


def load_synth(module, manifest):
    sd = synth.state_dict(manifest)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    module.eval()
    return sd


# ----------------------------------------------------------------------------- builders

def build_hift(cfg: HiftCfg):
    from cosyvoice.hifigan.f0_predictor import CausalConvRNNF0Predictor
    from cosyvoice.hifigan.generator import CausalHiFTGenerator
    m = CausalHiFTGenerator(
        in_channels=cfg.mel, base_channels=cfg.base, nb_harmonics=cfg.harmonics, sampling_rate=24000,
        nsf_alpha=cfg.nsf_alpha, nsf_sigma=cfg.nsf_sigma, nsf_voiced_threshold=cfg.voiced_thr,
        upsample_rates=list(cfg.ups), upsample_kernel_sizes=list(cfg.up_k),
        istft_params={"n_fft": cfg.n_fft, "hop_len": cfg.hop},
        resblock_kernel_sizes=list(cfg.rb_k), resblock_dilation_sizes=[list(cfg.rb_d)] * 3,
        source_resblock_kernel_sizes=list(cfg.src_rb_k), source_resblock_dilation_sizes=[list(cfg.rb_d)] * 3,
        lrelu_slope=cfg.lrelu, audio_limit=cfg.audio_limit, conv_pre_look_right=cfg.pre_look_right,
        f0_predictor=CausalConvRNNF0Predictor(1, cfg.mel, cfg.f0_ch))
    check_manifest(m, cfg.manifest(), "hift")
    load_synth(m, cfg.manifest())
    return m


def set_hift_noise(m, n_samples):
    m.m_source.l_sin_gen.rand_ini = torch.from_numpy(synth.hift_rand_ini())
    m.m_source.l_sin_gen.sine_waves = torch.from_numpy(synth.hift_sine_noise(n_samples))


def build_flow(cfg: FlowCfg):
    from omegaconf import DictConfig
    from cosyvoice.flow.DiT.dit import DiT
    from cosyvoice.flow.flow import CausalMaskedDiffWithDiT
    from cosyvoice.flow.flow_matching import CausalConditionalCFM
    from cosyvoice.transformer.upsample_encoder import PreLookaheadLayer
    est = DiT(dim=cfg.dim, depth=cfg.depth, heads=cfg.heads, dim_head=cfg.head_dim, ff_mult=cfg.ff_mult,
              mel_dim=cfg.mel, mu_dim=cfg.mel, spk_dim=cfg.mel, out_channels=cfg.mel,
              static_chunk_size=cfg.static_chunk, num_decoding_left_chunks=-1)
    cfm = CausalConditionalCFM(
        in_channels=240, n_spks=1, spk_emb_dim=80,
        cfm_params=DictConfig(content=dict(sigma_min=1e-6, solver="euler", t_scheduler="cosine",
                                           training_cfg_rate=0.2, inference_cfg_rate=cfg.cfg_rate,
                                           reg_loss_type="l1")),
        estimator=est)
    m = CausalMaskedDiffWithDiT(input_size=80, output_size=80, spk_embed_dim=cfg.spk_in, output_type="mel",
                                vocab_size=cfg.vocab, input_frame_rate=25, only_mask_loss=True,
                                token_mel_ratio=2, pre_lookahead_len=cfg.pre_lookahead,
                                pre_lookahead_layer=PreLookaheadLayer(80, cfg.pre_ch, cfg.pre_lookahead),
                                decoder=cfm)
    check_manifest(m, cfg.manifest(), "flow")
    load_synth(m, cfg.manifest())
    return m


class GreedySampler:
    """argmax injected as CosyVoice3LM.sampling.  sampling_ids (llm/llm.py:149-164)
    re-calls the sampler while the draw is an eos id and eos is forbidden; the
    second call for the same step answers with the argmax over real speech
    tokens - the greedy rule of SURVEY §8 a4."""

    def __init__(self, n_speech):
        self.n, self.last = n_speech, None

    def __call__(self, scores, decoded, sampling):
        key = (len(decoded), scores.data_ptr())
        if self.last == key:
            return int(torch.argmax(scores[: self.n]))
        self.last = key
        return int(torch.argmax(scores))


def build_llm(cfg: LlmCfg):
    from transformers import Qwen2Config, Qwen2ForCausalLM
    from cosyvoice.llm.llm import CosyVoice3LM, Qwen2Encoder
    enc = Qwen2Encoder.__new__(Qwen2Encoder)          # bypass from_pretrained(<absent dir>)
    torch.nn.Module.__init__(enc)
    qc = Qwen2Config(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter,
                     num_hidden_layers=cfg.layers, num_attention_heads=cfg.q_heads,
                     num_key_value_heads=cfg.kv_heads, max_position_embeddings=32768,
                     rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, tie_word_embeddings=True,
                     use_sliding_window=False)
    with torch.device("cpu"):
        enc.model = Qwen2ForCausalLM(qc)
    # The reference pins transformers==4.51.3 (CosyVoice/requirements.txt:38), where
    # an all-ones attention_mask is ignored for a cached decode step
    # (AttentionMaskConverter._ignore_causal_mask_sdpa) and the step attends the whole
    # cache.  forward_one_step (llm/llm.py:246-258) passes a mask as long as the NEW
    # tokens only; transformers 5.15 (this image) misreads that (1,1) mask against a
    # longer cache.  Dropping an all-ones mask on the third-party model restores the
    # pinned version's behaviour; the reference's own code is untouched.
    hf_forward = enc.model.forward

    def forward_4_51_semantics(*args, attention_mask=None, **kw):
        if attention_mask is not None and bool(attention_mask.all()):
            attention_mask = None
        return hf_forward(*args, attention_mask=attention_mask, **kw)
    enc.model.forward = forward_4_51_semantics
    m = CosyVoice3LM(cfg.hidden, cfg.hidden, cfg.speech_tokens, enc, sampling=GreedySampler(cfg.speech_tokens))
    check_manifest(m, cfg.manifest(), "llm")
    load_synth(m, cfg.manifest())
    return m


# ----------------------------------------------------------------------------- goldens

def synth_mel(name, frames):
    """log-mel-like prompt features: N(-5, 2^2) clipped to [-11.5, 2]."""
    return np.clip(synth.normal(name, (1, frames, 80), -5.0, 2.0), -11.5, 2.0)


def mint_hift(tag, cfg: HiftCfg, frames_list, out):
    m = build_hift(cfg)
    fx = {}
    with torch.inference_mode():
        for Fr in frames_list:
            set_hift_noise(m, Fr * cfg.upsample_total)
            mel = torch.from_numpy(synth.uniform(f"in.hift.mel.{Fr}", (1, 80, Fr), 0.0, 1.0))
            t0 = time.time()
            f0 = m.f0_predictor(mel)
            s = m.f0_upsamp(f0[:, None]).transpose(1, 2)
            s, _, _ = m.m_source(s)
            s = s.transpose(1, 2)
            wav, s2 = m.inference(mel)
            dt = time.time() - t0
            assert torch.equal(s, s2)
            print(f"[hift {tag}] F={Fr}: wav {tuple(wav.shape)} |wav|max {wav.abs().max():.3f} "
                  f"clamped {(wav.abs() >= 0.99).float().mean():.3f}  f0 min/mean/max "
                  f"{f0.min():.1f}/{f0.mean():.1f}/{f0.max():.1f} voiced {(f0 > 10).float().mean():.2f}  {dt:.2f}s")
            p = f"F{Fr}"
            fx.update(pack(p + ".f0", digest(f0)))
            fx.update(pack(p + ".source", digest(s)))
            fx.update(pack(p + ".wav", digest(wav)))
            if Fr <= 30:
                fx[p + ".wav_full"] = wav.numpy().astype(np.float32)
                fx[p + ".f0_full"] = f0.numpy().astype(np.float32)
            # per-stage taps through forward hooks on the reference modules
            taps = {}
            hooks = [m.conv_pre.register_forward_hook(lambda mod, i, o: taps.__setitem__("conv_pre", o)),
                     m.conv_post.register_forward_hook(lambda mod, i, o: taps.__setitem__("conv_post", o))]
            for i in range(3):
                hooks.append(m.resblocks[3 * i].register_forward_hook(
                    lambda mod, inp, o, i=i: taps.__setitem__(f"fuse{i}", inp[0])))
            m.decode(x=mel, s=s, finalize=True)
            for h in hooks:
                h.remove()
            for k, v in taps.items():
                fx.update(pack(f"{p}.{k}", digest(v)))
                print(f"    tap {k}: {tuple(v.shape)} std {v.std():.3f} absmax {v.abs().max():.2f}")
        # a single ResBlock per (C, k) on its own input
        for i in range(3):
            for j, k in enumerate(cfg.rb_k):
                C = cfg.stage_ch(i)
                x = torch.from_numpy(synth.normal(f"in.hift.rb.{i}.{j}", (1, C, 200)))
                y = m.resblocks[3 * i + j](x)
                fx.update(pack(f"rb{3 * i + j}", digest(y)))
    np.savez_compressed(os.path.join(out, f"hift_{tag}.npz"), **fx)


def dit_inputs(T, tag):
    x = synth.normal(f"in.dit.x.{tag}", (2, 80, T))
    mu = synth.normal(f"in.dit.mu.{tag}", (2, 80, T))
    cond = synth.normal(f"in.dit.cond.{tag}", (2, 80, T))
    spks = synth.normal(f"in.dit.spks.{tag}", (2, 80))
    t = np.array([0.3, 0.3], dtype=np.float32)
    return [torch.from_numpy(a) for a in (x, mu, cond, spks, t)]


def mint_flow(tag, cfg: FlowCfg, est_T, cfm_cases, out):
    m = build_flow(cfg)
    fx = {}
    with torch.inference_mode():
        for T in est_T:
            x, mu, cond, spks, t = dit_inputs(T, T)
            mask = torch.ones(2, 1, T)
            t0 = time.time()
            y = m.decoder.estimator(x, mask, mu, t, spks, cond, streaming=False)
            print(f"[flow {tag}] estimator T={T}: std {y.std():.3f} absmax {y.abs().max():.2f}  {time.time() - t0:.2f}s")
            fx.update(pack(f"est{T}", digest(y)))
            if cfg.dim <= 256 or T <= 16:
                fx[f"est{T}.full"] = y.numpy()
            ys = m.decoder.estimator(x, mask, mu, t, spks, cond, streaming=True)
            fx.update(pack(f"est{T}.stream", digest(ys)))
        for n, p_tok in cfm_cases:
            token = torch.from_numpy(synth.randint(f"in.flow.token.{n}", (1, n), 0, cfg.vocab))
            ptoken = torch.from_numpy(synth.randint(f"in.flow.ptoken.{p_tok}", (1, p_tok), 0, cfg.vocab))
            pfeat = torch.from_numpy(synth_mel(f"in.flow.pfeat.{p_tok}", 2 * p_tok))
            emb = torch.from_numpy(synth.normal("in.flow.spk", (1, cfg.spk_in)))
            T = 2 * (n + p_tok)
            m.decoder.rand_noise = torch.from_numpy(synth.flow_rand_noise(T))
            t0 = time.time()
            mel, _ = m.inference(token, torch.tensor([n]), ptoken, torch.tensor([p_tok]), pfeat,
                                 torch.tensor([2 * p_tok]), emb, streaming=False, finalize=True)
            print(f"[flow {tag}] cfm n={n} P={p_tok}: mel {tuple(mel.shape)} mean {mel.mean():.3f} std {mel.std():.3f} "
                  f"absmax {mel.abs().max():.2f}  {time.time() - t0:.2f}s")
            fx.update(pack(f"cfm{n}_{p_tok}", digest(mel)))
            if mel.numel() <= 8000:
                fx[f"cfm{n}_{p_tok}.full"] = mel.numpy()
    np.savez_compressed(os.path.join(out, f"flow_{tag}.npz"), **fx)


def llm_case(cfg: LlmCfg, n_text, n_prompt_text, p_tok, tag):
    text = synth.randint(f"in.llm.text.{tag}", (1, n_text), 0, min(cfg.vocab, 151643))
    ptext = synth.randint(f"in.llm.ptext.{tag}", (1, n_prompt_text), 0, min(cfg.vocab, 151643))
    ptok = synth.randint(f"in.llm.ptok.{tag}", (1, p_tok), 0, cfg.speech_tokens)
    return [torch.from_numpy(a) for a in (text, ptext, ptok)]


def mint_llm(tag, cfg: LlmCfg, cases, out, max_steps):
    m = build_llm(cfg)
    fx = {}
    for (n_text, n_ptext, p_tok) in cases:
        ctag = f"{n_text}_{n_ptext}_{p_tok}"
        text, ptext, ptok = llm_case(cfg, n_text, n_ptext, p_tok, ctag)
        logps = []
        dec = m.llm_decoder
        hook = dec.register_forward_hook(lambda mod, i, o: logps.append(o.log_softmax(dim=-1).squeeze(0).clone()))
        t0 = time.time()
        toks = []
        with torch.inference_mode():
            for tid in m.inference(text=text, text_len=torch.tensor([n_text], dtype=torch.int32),
                                   prompt_text=ptext, prompt_text_len=torch.tensor([n_ptext], dtype=torch.int32),
                                   prompt_speech_token=ptok, prompt_speech_token_len=torch.tensor([p_tok], dtype=torch.int32),
                                   embedding=torch.zeros(0, 192)):
                toks.append(int(tid))
                if len(toks) >= max_steps:
                    break
        hook.remove()
        gaps = [float((l.topk(2).values[0] - l.topk(2).values[1])) for l in logps]
        print(f"[llm {tag}] case {ctag}: {len(toks)} tokens (min_len {2 * n_text}) in {time.time() - t0:.1f}s; "
              f"min top-2 gap {min(gaps):.4f}; first ids {toks[:8]}")
        fx[f"c{ctag}.tokens"] = np.asarray(toks, dtype=np.int32)
        fx[f"c{ctag}.capped"] = np.asarray(len(toks) >= max_steps)
        for s in range(min(3, len(logps))):
            fx.update(pack(f"c{ctag}.logp{s}", digest(logps[s])))
        fx[f"c{ctag}.gap_min"] = np.float32(min(gaps))
    np.savez_compressed(os.path.join(out, f"llm_{tag}.npz"), **fx)


def mint_llm_ras(tag, cfg: LlmCfg, cases, out, max_steps):
    """The reference's DEFAULT sampler: ras_sampling (utils/common.py:137-166) inside sampling_ids (llm/llm.py:149-164), with the
    one thing no other implementation can follow - torch.multinomial's draws from the global generator - replaced by the
    inverse CDF at supplied uniforms (oracle.llm.inv_cdf), by patching Tensor.multinomial for the duration of the run."""
    from functools import partial
    from cosyvoice.utils.common import ras_sampling
    from oracle import llm as ollm
    m = build_llm(cfg)
    m.sampling = partial(ras_sampling, top_p=0.8, top_k=25, win_size=10, tau_r=0.1)          # cosyvoice3.yaml:32-36
    fx = {}
    orig = torch.Tensor.multinomial
    for (n_text, n_ptext, p_tok) in cases:
        ctag = f"{n_text}_{n_ptext}_{p_tok}"
        text, ptext, ptok = llm_case(cfg, n_text, n_ptext, p_tok, ctag)
        u = synth.uniform(f"in.llm.ras_u.{ctag}", (4096,), 0.0, 1.0)
        stream = ollm.UniformStream(u)
        draws = {"nucleus": 0, "full": 0}

        def fake_multinomial(self, num_samples, replacement=False, generator=None):
            assert num_samples == 1 and self.dim() == 1
            draws["full" if self.numel() > 32 else "nucleus"] += 1
            return torch.tensor([ollm.inv_cdf(self.detach().float().numpy(), stream.next())], dtype=torch.long)
        torch.Tensor.multinomial = fake_multinomial
        toks = []
        t0 = time.time()
        raised = ""
        try:
            with torch.inference_mode():
                for tid in m.inference(text=text, text_len=torch.tensor([n_text], dtype=torch.int32),
                                       prompt_text=ptext, prompt_text_len=torch.tensor([n_ptext], dtype=torch.int32),
                                       prompt_speech_token=ptok, prompt_speech_token_len=torch.tensor([p_tok], dtype=torch.int32),
                                       embedding=torch.zeros(0, 192)):
                    toks.append(int(tid))
                    if len(toks) >= max_steps:
                        break
        except RuntimeError as e:                 # sampling_ids gives up after 100 eos draws below min_len (llm.py:161-162)
            raised = str(e)
        finally:
            torch.Tensor.multinomial = orig
        fx[f"c{ctag}.raised"] = np.asarray(raised)
        print(f"[llm-ras {tag}] case {ctag}: raised {raised!r}; {len(toks)} tokens (min_len {2 * n_text}) in {time.time() - t0:.1f}s; draws {draws}, "
              f"{stream.pos} uniforms; first ids {toks[:8]}")
        fx[f"c{ctag}.tokens"] = np.asarray(toks, dtype=np.int32)
        fx[f"c{ctag}.capped"] = np.asarray(len(toks) >= max_steps)
        fx[f"c{ctag}.uniforms_used"] = np.asarray(stream.pos)
        fx[f"c{ctag}.full_draws"] = np.asarray(draws["full"])
    np.savez_compressed(os.path.join(out, f"llm_ras_{tag}.npz"), **fx)


def mint_e2e(tag, cfg: ModelCfg, cases, out):
    """CosyVoice3Model.tts(stream=False) of the reference: dict -> ids -> mel -> wav."""
    from cosyvoice.cli.model import CosyVoice3Model
    llm, flow, hift = build_llm(cfg.llm), build_flow(cfg.flow), build_hift(cfg.hift)
    model = CosyVoice3Model(llm, flow, hift)
    fx = {}
    for (n_text, n_ptext, p_llm, p_flow) in cases:
        ctag = f"{n_text}_{n_ptext}_{p_llm}_{p_flow}"
        text, ptext, ptok_llm = llm_case(cfg.llm, n_text, n_ptext, p_llm, ctag)
        ptok_flow = torch.from_numpy(synth.randint(f"in.flow.ptoken.{p_flow}", (1, p_flow), 0, cfg.flow.vocab))
        pfeat = torch.from_numpy(synth_mel(f"in.flow.pfeat.{p_flow}", 2 * p_flow))
        emb = torch.from_numpy(synth.normal("in.flow.spk", (1, cfg.flow.spk_in)))
        max_T = 2 * (p_flow + 20 * n_text)
        flow.decoder.rand_noise = torch.from_numpy(synth.flow_rand_noise(max_T))
        set_hift_noise(hift, 2 * 20 * n_text * 480)
        mels = []
        h = flow.register_forward_hook(lambda *a: None)
        h.remove()
        orig = flow.inference

        def tap(**kw):
            r = orig(**kw)
            mels.append(r[0].clone())
            return r
        flow.inference = tap
        toks_seen = {}
        orig_job = model.llm_job

        def job(text, prompt_text, llm_prompt_speech_token, llm_embedding, uuid):
            orig_job(text, prompt_text, llm_prompt_speech_token, llm_embedding, uuid)
            toks_seen["t"] = list(model.tts_speech_token_dict[uuid])
        model.llm_job = job
        t0 = time.time()
        outs = list(model.tts(text=text, flow_embedding=emb, llm_embedding=emb, prompt_text=ptext,
                              llm_prompt_speech_token=ptok_llm, flow_prompt_speech_token=ptok_flow,
                              prompt_speech_feat=pfeat, stream=False))
        flow.inference = orig
        model.llm_job = orig_job
        wav = outs[0]["tts_speech"]
        toks = toks_seen["t"]
        print(f"[e2e {tag}] case {ctag}: {len(toks)} tokens -> mel {tuple(mels[0].shape)} -> wav {tuple(wav.shape)} "
              f"in {time.time() - t0:.1f}s ({wav.shape[1] / 24000:.2f}s audio)")
        fx[f"c{ctag}.tokens"] = np.asarray(toks, dtype=np.int32)
        fx.update(pack(f"c{ctag}.mel", digest(mels[0])))
        fx.update(pack(f"c{ctag}.wav", digest(wav)))
    np.savez_compressed(os.path.join(out, f"e2e_{tag}.npz"), **fx)


def mint_stream(tag, cfg: ModelCfg, hift_frames, flow_case, e2e_cases, out):
    """The streaming forms: CausalHiFTGenerator.inference(finalize=False), CausalMaskedDiffWithDiT.inference(streaming=True,
    finalize=False) and CosyVoice3Model.tts(stream=True) (chunk by chunk)."""
    from cosyvoice.cli.model import CosyVoice3Model
    llm, flow, hift = build_llm(cfg.llm), build_flow(cfg.flow), build_hift(cfg.hift)
    fx = {}
    with torch.inference_mode():
        for Fr in hift_frames:
            set_hift_noise(hift, Fr * cfg.hift.upsample_total)
            mel = torch.from_numpy(synth.uniform(f"in.hift.mel.{Fr}", (1, 80, Fr), 0.0, 1.0))
            wav, s = hift.inference(mel, finalize=False)
            print(f"[stream {tag}] hift F={Fr} finalize=False: wav {tuple(wav.shape)} source {tuple(s.shape)}")
            fx.update(pack(f"hift.F{Fr}.wav", digest(wav)))
            fx.update(pack(f"hift.F{Fr}.source", digest(s)))
            if Fr <= 30:
                fx[f"hift.F{Fr}.wav_full"] = wav.numpy().astype(np.float32)
        n, p_tok = flow_case
        token = torch.from_numpy(synth.randint(f"in.flow.token.{n}", (1, n), 0, cfg.flow.vocab))
        ptoken = torch.from_numpy(synth.randint(f"in.flow.ptoken.{p_tok}", (1, p_tok), 0, cfg.flow.vocab))
        pfeat = torch.from_numpy(synth_mel(f"in.flow.pfeat.{p_tok}", 2 * p_tok))
        emb = torch.from_numpy(synth.normal("in.flow.spk", (1, cfg.flow.spk_in)))
        flow.decoder.rand_noise = torch.from_numpy(synth.flow_rand_noise(2 * (n + p_tok)))
        mel, _ = flow.inference(token, torch.tensor([n]), ptoken, torch.tensor([p_tok]), pfeat, torch.tensor([2 * p_tok]), emb,
                                streaming=True, finalize=False)
        print(f"[stream {tag}] flow n={n} P={p_tok} streaming, finalize=False: mel {tuple(mel.shape)}")
        fx.update(pack(f"flow.{n}_{p_tok}", digest(mel)))
        if mel.numel() <= 8000:
            fx[f"flow.{n}_{p_tok}.full"] = mel.numpy()
    model = CosyVoice3Model(llm, flow, hift)
    for (n_text, n_ptext, p_llm, p_flow) in e2e_cases:
        ctag = f"{n_text}_{n_ptext}_{p_llm}_{p_flow}"
        text, ptext, ptok_llm = llm_case(cfg.llm, n_text, n_ptext, p_llm, ctag)
        ptok_flow = torch.from_numpy(synth.randint(f"in.flow.ptoken.{p_flow}", (1, p_flow), 0, cfg.flow.vocab))
        pfeat = torch.from_numpy(synth_mel(f"in.flow.pfeat.{p_flow}", 2 * p_flow))
        emb = torch.from_numpy(synth.normal("in.flow.spk", (1, cfg.flow.spk_in)))
        flow.decoder.rand_noise = torch.from_numpy(synth.flow_rand_noise(2 * (p_flow + 20 * n_text)))
        set_hift_noise(hift, 2 * 20 * n_text * 480)
        toks_seen = {}
        orig_job = model.llm_job

        def job(text, prompt_text, llm_prompt_speech_token, llm_embedding, uuid):
            orig_job(text, prompt_text, llm_prompt_speech_token, llm_embedding, uuid)
            toks_seen["t"] = list(model.tts_speech_token_dict[uuid])
        model.llm_job = job
        t0 = time.time()
        outs = [o["tts_speech"] for o in model.tts(text=text, flow_embedding=emb, llm_embedding=emb, prompt_text=ptext,
                                                   llm_prompt_speech_token=ptok_llm, flow_prompt_speech_token=ptok_flow,
                                                   prompt_speech_feat=pfeat, stream=True)]
        model.llm_job = orig_job
        toks = toks_seen["t"]
        print(f"[stream {tag}] tts(stream=True) case {ctag}: {len(toks)} tokens -> chunks {[o.shape[1] for o in outs]} "
              f"in {time.time() - t0:.1f}s")
        fx[f"e2e.c{ctag}.tokens"] = np.asarray(toks, dtype=np.int32)
        fx[f"e2e.c{ctag}.chunk_samples"] = np.asarray([o.shape[1] for o in outs], dtype=np.int64)
        for i, o in enumerate(outs):
            fx.update(pack(f"e2e.c{ctag}.chunk{i}", digest(o)))
        fx.update(pack(f"e2e.c{ctag}.wav", digest(torch.cat(outs, dim=1))))
    np.savez_compressed(os.path.join(out, f"stream_{tag}.npz"), **fx)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="hift,flow,llm,e2e,stream,ras")
    ap.add_argument("--full", action="store_true", help="also mint the full-size (CosyVoice3-0.5B shape) fixtures")
    ap.add_argument("--sized", action="store_true", help="mint the fixtures at BASELINE.json's configuration sizes (*_sized.npz; minutes of CPU)")
    ap.add_argument("--fp32-weights", action="store_true",
                    help="mint the *_fp32w.npz fixtures: the same cases with synth's bf16 rounding of the matrices switched OFF - "
                         "general fp32 weights, as a real llm.pt / flow.pt / hift.pt holds them (cli/model.py:65-73)")
    ap.add_argument("--out", default=HERE)
    a = ap.parse_args()
    torch.manual_seed(0)
    install_stubs()
    only = set(a.only.split(","))
    tiny = ModelCfg.tiny()
    if a.fp32_weights:
        # every other fixture feeds the reference weights that happen to be bf16-exact, so the one quantisation
        # step a real checkpoint goes through in a bf16-storage engine is outside them.  These are the reference on unrounded weights.
        with synth.unrounded_weights():
            if "llm" in only:
                mint_llm("tiny_fp32w", tiny.llm, [(12, 8, 0), (10, 6, 30)], a.out, max_steps=400)
                mint_llm("full_fp32w", LlmCfg(), [(12, 8, 0), (14, 10, 40)], a.out, max_steps=60)
            if "flow" in only:
                mint_flow("tiny_fp32w", tiny.flow, [16, 150], [(20, 10)], a.out)
                mint_flow("full_fp32w", FlowCfg(), [16, 150], [(20, 0)], a.out)
            if "hift" in only:
                mint_hift("tiny_fp32w", tiny.hift, [30], a.out)
                mint_hift("full_fp32w", HiftCfg(), [30], a.out)
            if "e2e" in only:
                mint_e2e("tiny_fp32w", tiny, [(8, 6, 0, 12)], a.out)
                mint_e2e("sized_fp32w", ModelCfg(), [(14, 8, 0, 125)], a.out)
        return
    if a.sized:
        # SURVEY 8(c) G3 / G5 / G6 at the sizes BASELINE.json's configurations run at: the DiT at T = 512 (the top of
        # export_onnx.py's sweep) and T = 650 (config 3: 10 s prompt + 3 s), the 10-step mel at config 2's (75 tokens, 5 s
        # prompt) and config 3's (75, 10 s prompt) shapes, the LM behind a 250-token prompt (prefill of 296 rows), and the whole
        # tts() for config 1 (instruct: 8 + 14 text ids, 5 s prompt) and config 3 (zero-shot: 30 prompt-text ids, 250 + 250)
        if "flow" in only:
            mint_flow("sized", FlowCfg(), [512, 650], [(75, 125), (75, 250)], a.out)
        if "llm" in only:
            mint_llm("sized", LlmCfg(), [(14, 30, 250)], a.out, max_steps=40)
        if "e2e" in only:
            mint_e2e("sized", ModelCfg(), [(14, 8, 0, 125), (14, 30, 250, 250)], a.out)
        return
    if "hift" in only:
        mint_hift("tiny", tiny.hift, [12, 30], a.out)
        if a.full:
            mint_hift("full", HiftCfg(), [30, 150], a.out)
    if "flow" in only:
        mint_flow("tiny", tiny.flow, [16, 150], [(20, 10), (24, 0)], a.out)
        if a.full:
            mint_flow("full", FlowCfg(), [16, 150], [(20, 0), (16, 24)], a.out)
    if "llm" in only:
        mint_llm("tiny", tiny.llm, [(12, 8, 0), (10, 6, 30)], a.out, max_steps=400)
        if a.full:
            mint_llm("full", LlmCfg(), [(12, 8, 0), (14, 10, 40)], a.out, max_steps=60)
    if "e2e" in only:
        mint_e2e("tiny", tiny, [(8, 6, 0, 12), (6, 5, 20, 20)], a.out)
        if a.full:
            mint_e2e("full", ModelCfg(), [(8, 8, 0, 25)], a.out)
    if "ras" in only:
        mint_llm_ras("tiny", tiny.llm, [(12, 8, 0), (10, 6, 30), (16, 4, 10), (30, 5, 0)], a.out, max_steps=400)
        if a.full:
            mint_llm_ras("full", LlmCfg(), [(12, 8, 0)], a.out, max_steps=60)
    if "stream" in only:
        mint_stream("tiny", tiny, [30], (31, 10), [(40, 6, 0, 12)], a.out)
        if a.full:
            mint_stream("full", ModelCfg(), [30], (23, 10), [(8, 8, 0, 25)], a.out)


if __name__ == "__main__":
    main()
