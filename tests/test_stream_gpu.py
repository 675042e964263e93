"""GPU parity of the streaming forms: HiFT / flow with finalize=False and CosyVoice3Model.tts(stream=True),
against the fixtures minted from the reference (tests/golden/stream_*.npz) and the oracle.

Tolerances as for the non-streaming forms (tests/test_hift_gpu.py, test_flow_gpu.py, test_e2e_gpu.py): HiFT fp32 mode 2e-3 on
the waveform against the reference fixture, default bf16 mode 1.5e-2; flow mel max |err| <= 6e-2; streamed tokens and chunk
lengths exact; every chunk's samples within 1.5e-2 of the oracle vocoder run on the engine's own mel (the NSF source
integrates f0, so a comparison through a differently rounded mel decorrelates - see test_e2e_gpu.py); the first 10
frames of the stream within 3e-2 of the reference fixture itself.
"""
import numpy as np
import pytest
import torch

from _digest import check
from fangyan_tts_amd import synth
from fangyan_tts_amd.spec import FlowCfg, HiftCfg, ModelCfg
from gpu_util import golden, llm_case, maxerr, note, synth_mel

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_hift_chunk(tag):
    from fangyan_tts_amd.hift import HiftEngine
    f = golden(f"stream_{tag}.npz")
    if f is None:
        pytest.skip(f"stream_{tag}.npz not minted")
    cfg = ModelCfg.tiny().hift if tag == "tiny" else HiftCfg()
    sd = synth.state_dict_torch(cfg.manifest(), DEV)
    eng = HiftEngine(sd, cfg, max_batch=2, max_frames=40, device=DEV)
    Fr = 30
    mel = torch.from_numpy(synth.uniform(f"in.hift.mel.{Fr}", (1, 80, Fr), 0.0, 1.0)).to(DEV)
    ri = torch.from_numpy(synth.hift_rand_ini()).to(DEV)
    sn = torch.from_numpy(synth.hift_sine_noise(40 * 480)).to(DEV)
    for mode, flags, tol in (("direct", 2, 3e-5), ("bf16", 0, 2e-3)):                     # measured 1e-5 / 6e-4
        wav, src = eng.inference(mel, ri, sn, flags=flags, want_source=True, finalize=False)
        n = (Fr - 8) * 480
        e = float(np.abs(wav[0, :n].cpu().numpy() - f[f"hift.F{Fr}.wav_full"][0]).max())
        note("parity_stream.json", f"hift.{tag}.{mode}.wav_maxerr", e)
        assert e < tol
        assert float(wav[0, n:].abs().max()) == 0.0                      # the held-back frame is not written
        if flags == 2:
            check(src[:, :, : (Fr - 3) * 480].cpu(), f, f"hift.F{Fr}.source", 1e-3, 5e-3)
    # ragged batch: a chunk beside a longer one equals the chunk alone
    mel2 = torch.zeros(2, 80, 38, device=DEV)
    mel2[0, :, :Fr] = mel[0]
    mel2[1] = torch.from_numpy(synth.uniform("in.hift.mel.38", (1, 80, 38), 0.0, 1.0)).to(DEV)[0]
    w2, _ = eng.inference(mel2, ri, sn, frames=[Fr, 38], finalize=False)
    w1, _ = eng.inference(mel, ri, sn, finalize=False)
    assert torch.equal(w2[0, : (Fr - 8) * 480], w1[0, : (Fr - 8) * 480])


@pytest.mark.parametrize("tag,n,p_tok", [("tiny", 31, 10), ("full", 23, 10)])
def test_flow_chunk(tag, n, p_tok):
    from fangyan_tts_amd.flow import FlowEngine
    f = golden(f"stream_{tag}.npz")
    if f is None:
        pytest.skip(f"stream_{tag}.npz not minted")
    cfg = ModelCfg.tiny().flow if tag == "tiny" else FlowCfg()
    sd = synth.state_dict_torch(cfg.manifest(), DEV)
    eng = FlowEngine(sd, cfg, max_batch=1, max_frames=2 * (n + p_tok), device=DEV)
    token = torch.from_numpy(synth.randint(f"in.flow.token.{n}", (1, n), 0, cfg.vocab))
    ptoken = torch.from_numpy(synth.randint(f"in.flow.ptoken.{p_tok}", (1, p_tok), 0, cfg.vocab))
    pfeat = torch.from_numpy(synth_mel(f"in.flow.pfeat.{p_tok}", 2 * p_tok))
    emb = torch.from_numpy(synth.normal("in.flow.spk", (1, cfg.spk_in)))
    noise = torch.from_numpy(synth.flow_rand_noise(2 * (n + p_tok)))
    mel = eng.inference(token, [n], ptoken, [p_tok], pfeat, [2 * p_tok], emb, noise, streaming=True, finalize=False)
    valid = 2 * (n - cfg.pre_lookahead)
    assert float(mel[:, :, valid:].abs().max()) == 0.0
    check(mel[:, :, :valid].cpu(), f, f"flow.{n}_{p_tok}", 0.0, 5e-2)


@pytest.mark.parametrize("tag,case", [("tiny", (40, 6, 0, 12)), ("full", (8, 8, 0, 25))])
def test_tts_stream_against_reference(tag, case):
    from fangyan_tts_amd.cli.model import CosyVoice3Model
    from oracle import hift as ohift
    f = golden(f"stream_{tag}.npz")
    if f is None:
        pytest.skip(f"stream_{tag}.npz not minted")
    cfg = ModelCfg.tiny() if tag == "tiny" else ModelCfg()
    n_text, n_ptext, p_llm, p_flow = case
    ctag = f"{n_text}_{n_ptext}_{p_llm}_{p_flow}"
    sd = [synth.state_dict_torch(m.manifest(), DEV, skip=("lm_head",)) for m in (cfg.llm, cfg.flow, cfg.hift)]
    ri = torch.from_numpy(synth.hift_rand_ini())
    sn = torch.from_numpy(synth.hift_sine_noise(2 * 20 * n_text * 480))
    m = CosyVoice3Model(sd[0], sd[1], sd[2], cfg, device=DEV, max_batch=1, max_text=64, max_prompt_tokens=32, max_tokens=20 * n_text,
                        rand_noise=torch.from_numpy(synth.flow_rand_noise(2 * (p_flow + 20 * n_text))), rand_ini=ri, sine_noise=sn)
    t, pt, pk = llm_case(cfg.llm, n_text, n_ptext, p_llm, ctag)
    inp = {
        "text": torch.tensor([t], dtype=torch.int32), "prompt_text": torch.tensor([pt], dtype=torch.int32),
        "llm_prompt_speech_token": torch.tensor([pk], dtype=torch.int32).reshape(1, -1),
        "flow_prompt_speech_token": torch.from_numpy(synth.randint(f"in.flow.ptoken.{p_flow}", (1, p_flow), 0, 6561)),
        "prompt_speech_feat": torch.from_numpy(synth_mel(f"in.flow.pfeat.{p_flow}", 2 * p_flow)),
        "flow_embedding": torch.from_numpy(synth.normal("in.flow.spk", (1, 192))),
    }
    chunks = [o["tts_speech"] for o in m.tts(**inp, stream=True)]
    assert [c.shape[1] for c in chunks] == f[f"e2e.c{ctag}.chunk_samples"].tolist()
    assert all(c.device.type == "cpu" and c.shape[0] == 1 for c in chunks)
    # the token sequence behind the chunks: the reference's greedy ids (the chunk lengths above already depend on its length)
    wav_ns, samples, toks = m.tts_batch([inp])
    assert toks[0].cpu().tolist() == f[f"e2e.c{ctag}.tokens"].tolist()
    # every chunk against the oracle vocoder on the engine's own accumulated mel
    P = ohift.prepare({k: v.cpu().numpy() for k, v in sd[2].items()})
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_first = None
    for _ in m.tts(**inp, stream=True):                      # second pass (warm): when does the first chunk leave?
        t_first = t_first or time.perf_counter() - t0
    note("parity_stream.json", f"tts_stream.{tag}.first_chunk_ms_of_total_ms", [round(1e3 * t_first, 1), round(1e3 * (time.perf_counter() - t0), 1)])
    mel_all = m.last_mel.cpu()
    first = f[f"e2e.c{ctag}.chunk_samples"].tolist()
    off, worst = 0, 0.0
    for i, c in enumerate(chunks):
        last = i == len(chunks) - 1
        end = off + c.shape[1]
        Fk = end // 480 + (0 if last else 8)
        ref, _ = ohift.inference(mel_all[:, :, :Fk], P, cfg.hift, ri, sn, finalize=last)
        worst = max(worst, maxerr(c, ref[:, off:end]))
        off = end
    note("parity_stream.json", f"tts_stream.{tag}.chunk_wav_vs_oracle_vocoder_on_engine_mel", worst)
    assert worst < 2.5e-3                                   # measured 8e-4
    # and the first 10 frames of the stream against the reference fixture itself
    from _digest import sample_idx
    si = sample_idx(first[0])
    early = si < 4800
    got = chunks[0][0].numpy()[si][early]
    want = f[f"e2e.c{ctag}.chunk0.samples"][early]
    note("parity_stream.json", f"tts_stream.{tag}.first10frames_maxerr", float(np.abs(got - want).max()))
    assert np.abs(got - want).max() < 3e-2          # measured 1.9e-2 (full size), 1.9e-3 (reduced size)


def test_hift_chunked_equals_whole():
    """The reference's own self-consistency script (hifigan/generator.py:728-746) on the engine, full size: the vocoder run
    on growing prefixes (chunk 30 frames, 8 frames of context, finalize only on the last) reproduces the whole-utterance
    waveform - the model is causal, and the kernels' tiles are anchored at frame 0, so the arithmetic per sample is the same."""
    from fangyan_tts_amd.hift import HiftEngine
    cfg = HiftCfg()
    sd = synth.state_dict_torch(cfg.manifest(), DEV)
    max_len, chunk, ctx = 300, 30, 8
    eng = HiftEngine(sd, cfg, max_batch=1, max_frames=max_len, device=DEV)
    g = torch.Generator(device="cpu").manual_seed(0)
    mel = torch.rand(1, 80, max_len, generator=g).to(DEV)
    ri = torch.from_numpy(synth.hift_rand_ini()).to(DEV)
    sn = torch.from_numpy(synth.hift_sine_noise(max_len * 480)).to(DEV)
    for mode, flags, tol in (("direct", 2, 1e-5), ("bf16", 0, 1e-5)):
        gt, _ = eng.inference(mel, ri, sn, flags=flags)
        worst = 0.0
        for i in range(0, max_len, chunk):
            fin = i + chunk + ctx >= max_len
            n_in = min(i + chunk + ctx, max_len)
            w, _ = eng.inference(mel[:, :, :n_in].contiguous(), ri, sn, flags=flags, finalize=fin)
            n_out = (n_in if fin else n_in - 8) * 480
            worst = max(worst, maxerr(w[:, i * 480: n_out], gt[:, i * 480: n_out]))
        note("parity_stream.json", f"hift.chunked_vs_whole.{mode}", worst)
        assert worst <= tol, (mode, worst)


def test_flow_chunked_equals_whole():
    """The reference's flow self-consistency script (flow/flow.py:405-432), full size: with the chunk attention mask the mel
    of the tokens seen so far does not change when more tokens follow."""
    from fangyan_tts_amd.flow import FlowEngine
    cfg = FlowCfg()
    sd = synth.state_dict_torch(cfg.manifest(), DEV)
    chunk, ctx = 50, cfg.pre_lookahead                       # static_chunk_size is in mel frames (50); the script steps 50 TOKENS
    max_len = 4 * chunk
    eng = FlowEngine(sd, cfg, max_batch=1, max_frames=2 * (max_len + chunk), device=DEV)
    token = torch.from_numpy(synth.randint("in.flow.token.cons", (1, max_len), 0, cfg.vocab))
    ptoken = torch.from_numpy(synth.randint("in.flow.ptoken.cons", (1, chunk), 0, cfg.vocab))
    pfeat = torch.from_numpy(synth_mel("in.flow.pfeat.cons", 2 * chunk))
    emb = torch.from_numpy(synth.normal("in.flow.spk", (1, cfg.spk_in)))
    noise = torch.from_numpy(synth.flow_rand_noise(2 * (max_len + chunk)))
    gt = eng.inference(token, [max_len], ptoken, [chunk], pfeat, [2 * chunk], emb, noise, streaming=True, finalize=True)
    worst = 0.0
    for i in range(0, max_len, chunk):
        fin = i + chunk + ctx >= max_len
        n_in = min(i + chunk + ctx, max_len)
        m = eng.inference(token[:, :n_in].contiguous(), [n_in], ptoken, [chunk], pfeat, [2 * chunk], emb, noise, streaming=True, finalize=fin)
        n_out = 2 * (n_in if fin else n_in - ctx)
        worst = max(worst, maxerr(m[:, :, 2 * i: n_out], gt[:, :, 2 * i: n_out]))
    note("parity_stream.json", "flow.chunked_vs_whole", worst)
    assert worst <= 1e-5, worst                              # measured 0.0: rows and key tiles are anchored at frame 0


def test_flow_incremental_chunks_equal_full_recompute():
    """FY_INCREMENTAL (csrc/flow.hip): a streaming chunk that extends the previous one pushes only its NEW rows through the DiT blocks,
    against the keys / values kept per (Euler step, block).  With the reference's chunk schedule (25-token hops behind a prompt that
    is a multiple of 25 tokens, 3 look-ahead tokens: every call ends on a boundary of the 50-frame chunk mask) the earlier rows
    cannot see the new ones, so every chunk must equal the full recompute BIT FOR BIT (cli/model.py:339-369 recomputes; same values).
    Also: a second stream after stream_reset; a call that does not end on a mask boundary (computed in full, still equal)."""
    from fangyan_tts_amd.flow import FlowEngine
    cfg = FlowCfg()
    sd = synth.state_dict_torch(cfg.manifest(), DEV)
    hop, look, p_tok, n_max = 25, cfg.pre_lookahead, 50, 4 * 25 + 3
    eng = FlowEngine(sd, cfg, max_batch=1, max_frames=2 * (n_max + p_tok), device=DEV)
    ref = FlowEngine(sd, cfg, max_batch=1, max_frames=2 * (n_max + p_tok), device=DEV)
    noise = torch.from_numpy(synth.flow_rand_noise(2 * (n_max + p_tok)))
    emb = torch.from_numpy(synth.normal("in.flow.spk", (1, cfg.spk_in)))
    for stream in range(2):
        token = torch.from_numpy(synth.randint(f"in.flow.token.inc{stream}", (1, n_max), 0, cfg.vocab))
        ptoken = torch.from_numpy(synth.randint(f"in.flow.ptoken.inc{stream}", (1, p_tok), 0, cfg.vocab))
        pfeat = torch.from_numpy(synth_mel(f"in.flow.pfeat.inc{stream}", 2 * p_tok))
        eng.stream_reset()
        for k in range(1, 5):
            n_in = k * hop + look
            args = (token[:, :n_in].contiguous(), [n_in], ptoken, [p_tok], pfeat, [2 * p_tok], emb, noise)
            full = ref.inference(*args, streaming=True, finalize=False)
            inc = eng.inference(*args, streaming=True, finalize=False, incremental=True)
            valid = 2 * (n_in - look)
            assert torch.equal(inc[:, :, :valid], full[:, :, :valid]), (stream, k, maxerr(inc[:, :, :valid], full[:, :, :valid]))
            assert eng.stream_rows() == 2 * (p_tok + n_in - look)            # the call really was incremental: its rows are kept
    # a call that ends off the mask boundary is computed in full (and keeps nothing a later call could misuse)
    n_in = 2 * hop + look + 4
    args = (token[:, :n_in].contiguous(), [n_in], ptoken, [p_tok], pfeat, [2 * p_tok], emb, noise)
    assert torch.equal(eng.inference(*args, streaming=True, finalize=False, incremental=True), ref.inference(*args, streaming=True, finalize=False))
    assert eng.stream_rows() == 0
    n_in = 3 * hop + look
    args = (token[:, :n_in].contiguous(), [n_in], ptoken, [p_tok], pfeat, [2 * p_tok], emb, noise)
    assert torch.equal(eng.inference(*args, streaming=True, finalize=False, incremental=True), ref.inference(*args, streaming=True, finalize=False))
    # a caller that forgets stream_reset between two utterances: another prompt (here 25 tokens shorter), or fewer tokens than the stream
    # already holds, cannot be an extension - the call starts over instead of attending the previous utterance's keys and values
    assert eng.stream_rows() == 2 * (p_tok + n_in - look)
    p2 = p_tok - 25
    token2 = torch.from_numpy(synth.randint("in.flow.token.inc.other", (1, n_max), 0, cfg.vocab))
    for n_in in (2 * hop + look, 3 * hop + look, hop + look):
        args = (token2[:, :n_in].contiguous(), [n_in], ptoken[:, :p2].contiguous(), [p2], pfeat[:, :2 * p2].contiguous(), [2 * p2], emb, noise)
        valid = 2 * (n_in - look)
        inc = eng.inference(*args, streaming=True, finalize=False, incremental=True)
        full = ref.inference(*args, streaming=True, finalize=False)
        assert torch.equal(inc[:, :, :valid], full[:, :, :valid]), n_in
