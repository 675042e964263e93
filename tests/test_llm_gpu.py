"""GPU parity: the HIP speech-token LM through the C ABI against the fixtures minted from the
reference (real Qwen2ForCausalLM + CosyVoice3LM.inference) and the CPU oracle.

Token ids: bit-exact under greedy decode (weights bf16-representable, activations fp32).
log-probabilities of the first steps: |diff| <= 2e-3 (fp32 accumulation order).
"""
import numpy as np
import pytest
import torch

from _digest import check
from fangyan_tts_amd import synth
from fangyan_tts_amd.spec import LlmCfg
from gpu_util import golden, llm_case, note, to_dev

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def make(cfg, max_batch=4, max_ctx=512):
    from fangyan_tts_amd.llm import LlmEngine
    sd = synth.state_dict_torch(cfg.manifest(), DEV, skip=("lm_head",))
    return LlmEngine(sd, cfg, max_batch=max_batch, max_ctx=max_ctx)


# every test runs on both decode paths: the persistent one-launch token step (llm_decode.hip) and one launch per operation
@pytest.fixture(scope="module")
def _tiny():
    return make(LlmCfg.tiny())


@pytest.fixture(scope="module")
def _full():
    return make(LlmCfg(), max_batch=2, max_ctx=256)


def _mode(eng, persistent):
    eng.set_sampler("greedy")
    eng.set_decode_mode(persistent)
    assert eng.persistent == persistent, "the persistent decode kernel must be available for both test architectures"
    return eng


@pytest.fixture(params=[True, False], ids=["persistent", "per-op"])
def tiny(_tiny, request):
    return _mode(_tiny, request.param)


@pytest.fixture(params=[True, False], ids=["persistent", "per-op"])
def full(_full, request):
    return _mode(_full, request.param)


def run_cases(eng, f, cases, cap, tagname):
    cfg = eng.cfg
    texts, ptexts, ptoks, refs = [], [], [], []
    for c in cases:
        ctag = "%d_%d_%d" % c
        t, pt, pk = llm_case(cfg, *c, ctag)
        texts.append(t); ptexts.append(pt); ptoks.append(pk)
        refs.append(f[f"c{ctag}.tokens"].tolist())
    max_len = [cap if cap else int(len(t) * 20) for t in texts]
    out, out_n, raw_n = eng.generate(texts, ptexts, ptoks, max_len=max_len)
    out, out_n = out.cpu(), out_n.cpu().tolist()
    for b, c in enumerate(cases):
        got = out[b, : out_n[b]].tolist()
        ref = refs[b][: cap] if cap else refs[b]
        first_bad = next((i for i, (g, r) in enumerate(zip(got, ref)) if g != r), None)
        note("parity_llm.json", f"{tagname}.{c}.n", [len(got), len(ref), first_bad])
        assert got == ref, (c, first_bad, got[:10], ref[:10])
        ctag = "%d_%d_%d" % c
        for s in range(3):
            lp = eng.logp(s, len(cases))[b].cpu()
            check(lp, f, f"c{ctag}.logp{s}", 1e-3, 2e-3)


def test_tiny_tokens_bit_exact_batched(tiny):
    f = golden("llm_tiny.npz")
    assert f is not None
    run_cases(tiny, f, [(12, 8, 0), (10, 6, 30)], None, "tiny")


@pytest.mark.parametrize("size,n_rows", [("tiny", 12), ("tiny", 40), ("full", 20), ("full", 37)])
def test_many_rows_per_weight_pass(size, n_rows):
    """More than 8 sequences decode together on the per-operation path (what tts_pipeline's LM call over several steps'
    batches does): one weight pass serves 32 rows (gemv32.hip), 33+ rows take a second slice.  Every row must still emit the
    ids of the reference fixture for its case - the rows are independent, whatever slot and slice they sit in."""
    cfg = LlmCfg.tiny() if size == "tiny" else LlmCfg()
    f = golden(f"llm_{size}.npz")
    if f is None:
        pytest.skip(f"llm_{size}.npz not minted")
    base = [(12, 8, 0), (10, 6, 30)] if size == "tiny" else [(12, 8, 0), (14, 10, 40)]
    cap = None if size == "tiny" else 40
    eng = make(cfg, max_batch=n_rows, max_ctx=512 if size == "tiny" else 160)
    try:
        eng.set_decode_mode(False)
        cases = [base[(b * 7 // 3) % 2] for b in range(n_rows)]          # an irregular mix over the slots
        texts, ptexts, ptoks = (list(x) for x in zip(*[llm_case(cfg, *c, "%d_%d_%d" % c) for c in cases]))
        max_len = [cap if cap else int(len(t) * 20) for t in texts]
        out, out_n, _ = eng.generate(texts, ptexts, ptoks, max_len=max_len)
        out, out_n = out.cpu(), out_n.cpu().tolist()
        for b, c in enumerate(cases):
            ref = f["c%d_%d_%d.tokens" % c].tolist()
            ref = ref[:cap] if cap else ref
            assert out[b, : out_n[b]].tolist() == ref, (b, c)
    finally:
        eng.close()


@pytest.mark.parametrize("size,n_rows", [("tiny", 9), ("tiny", 20), ("tiny", 32), ("full", 12), ("full", 32)])
def test_persistent_step_for_up_to_32_rows(size, n_rows):
    """The few-CU persistent decode step (llm_decode32.hip: one launch per token step for 9 .. 32 sequences, 38 workgroups at full
    size) against the reference fixtures: every row emits its case's ids, whatever slot it sits in; and a generation that switches
    between this step and the per-operation launches mid-way (they share operands and layouts) emits the same ids."""
    cfg = LlmCfg.tiny() if size == "tiny" else LlmCfg()
    f = golden(f"llm_{size}.npz")
    if f is None:
        pytest.skip(f"llm_{size}.npz not minted")
    base = [(12, 8, 0), (10, 6, 30)] if size == "tiny" else [(12, 8, 0), (14, 10, 40)]
    cap = None if size == "tiny" else 40
    eng = make(cfg, max_batch=n_rows, max_ctx=512 if size == "tiny" else 160)
    try:
        eng.set_decode_mode(True)
        assert eng.persistent
        cases = [base[(b * 7 // 3) % 2] for b in range(n_rows)]
        texts, ptexts, ptoks = (list(x) for x in zip(*[llm_case(cfg, *c, "%d_%d_%d" % c) for c in cases]))
        max_len = [cap if cap else int(len(t) * 20) for t in texts]
        out, out_n, _ = eng.generate(texts, ptexts, ptoks, max_len=max_len)
        out, out_n = out.cpu().clone(), out_n.cpu().tolist()
        for b, c in enumerate(cases):
            ref = f["c%d_%d_%d.tokens" % c].tolist()
            ref = ref[:cap] if cap else ref
            assert out[b, : out_n[b]].tolist() == ref, (b, c)
        # the same generation in pieces, alternating between the persistent step and the per-operation launches
        out2, _, _ = eng.begin(texts, ptexts, ptoks, max_len=max_len)
        fin, k = [False], 0
        while not all(fin):
            eng.set_decode_mode(k % 2 == 0)
            n, fin = eng.step(5)
            k += 1
            assert k < 400
        assert n == out_n
        out2 = out2.cpu()
        for b in range(n_rows):
            assert out2[b, : n[b]].tolist() == out[b, : n[b]].tolist(), b
    finally:
        eng.close()


def test_eight_row_products_equal_the_32_row_products(monkeypatch):
    """FY_LLM_GEMV32=0 keeps the 8-row products of gemm.hip on the per-operation path (operands staged and split in every block,
    one weight pass per 8 rows): same ids and first log-probabilities as the 32-row products, tiny and full size."""
    for size, cases, cap in (("tiny", [(12, 8, 0), (10, 6, 30)], None), ("full", [(12, 8, 0), (14, 10, 40)], 24)):
        f = golden(f"llm_{size}.npz")
        if f is None:
            continue
        cfg = LlmCfg.tiny() if size == "tiny" else LlmCfg()
        monkeypatch.setenv("FY_LLM_GEMV32", "0")
        eng = make(cfg, max_batch=2, max_ctx=512 if size == "tiny" else 256)
        monkeypatch.delenv("FY_LLM_GEMV32")
        try:
            eng.set_decode_mode(False)
            run_cases(eng, f, cases, cap, f"{size}.gemv8")
        finally:
            eng.close()


def test_persistent_plan_recreated_after_a_generation():
    """A decode plan destroyed after a generation and a new one created (hipMalloc recycles small allocations and does not clear
    them): the new plan's flag array and "go" word are allocated and zeroed by decode_create, so its first launches hand off
    properly - ids equal the per-operation path's and the fixture's, several times over."""
    cfg = LlmCfg.tiny()
    f = golden("llm_tiny.npz")
    cases = [(12, 8, 0), (10, 6, 30)]
    for rep in range(4):
        eng = make(cfg)
        try:
            run_cases(_mode(eng, True), f, cases, None, f"tiny.recreated{rep}")
            if rep == 3:
                run_cases(_mode(eng, False), f, cases, None, "tiny.recreated.per-op")
        finally:
            eng.close()


def test_tiny_solo_equals_batched(tiny):
    f = golden("llm_tiny.npz")
    run_cases(tiny, f, [(10, 6, 30)], None, "tiny_solo")


def test_full_tokens_bit_exact(full):
    f = golden("llm_full.npz")
    if f is None:
        pytest.skip("llm_full.npz not minted")
    run_cases(full, f, [(12, 8, 0), (14, 10, 40)], 60, "full")


def test_full_size_behind_a_250_token_prompt(_full250, request):
    """BASELINE config 3's LM shape (zero-shot: 30 prompt-text ids + 14 text ids + 250 prompt speech tokens = a 296-row prefill)
    against the fixture minted from the reference's CosyVoice3LM.inference: ids exact, first log-probabilities, both paths."""
    f = golden("llm_sized.npz")
    if f is None:
        pytest.skip("llm_sized.npz not minted")
    for persistent in (True, False):
        run_cases(_mode(_full250, persistent), f, [(14, 30, 250)], 40, f"sized.{'persistent' if persistent else 'per-op'}")


def test_prefill_gemm_path_equals_the_8_row_products(monkeypatch):
    """The tiled exact-split GEMM prefill (used from ~300 rows on) and the 8-row products give the same ids and first
    log-probabilities, at the reduced size where both are forced on the same 2 x 40-row prefill."""
    cfg = LlmCfg.tiny()
    cases = [(12, 8, 0), (10, 6, 30)]
    texts, ptexts, ptoks = (list(x) for x in zip(*[llm_case(cfg, *c, "%d_%d_%d" % c) for c in cases]))
    res = []
    for rows in ("1", "100000"):
        monkeypatch.setenv("FY_LLM_PREFILL_GEMM_ROWS", rows)
        eng = make(cfg)
        out, out_n, _ = eng.generate(texts, ptexts, ptoks, max_len=[60, 60])
        res.append((out.cpu(), out_n.cpu(), eng.logp(0, 2).cpu()))
        eng.close()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert float((res[0][2] - res[1][2]).abs().max()) < 2e-3
    f = golden("llm_tiny.npz")
    assert res[0][0][0, : int(res[0][1][0])].tolist() == f["c12_8_0.tokens"].tolist()[: int(res[0][1][0])]


@pytest.fixture(scope="module")
def _full250():
    return make(LlmCfg(), max_batch=1, max_ctx=2 + 44 + 250 + 64)


def test_forced_length_and_min_len(tiny):
    """min_len = max_len = n (the benchmark's forced length) emits exactly n real speech tokens."""
    cfg = tiny.cfg
    t, pt, pk = llm_case(cfg, 10, 6, 30, "10_6_30")
    out, out_n, raw_n = tiny.generate([t, t], [pt, pt], [pk, []], min_len=[40, 33], max_len=[40, 33])
    assert raw_n.cpu().tolist() == [40, 33]
    o = out.cpu()
    assert int(o[0, :40].max()) < cfg.speech_tokens and int(o[1, :33].max()) < cfg.speech_tokens


def test_begin_step_equals_generate(tiny):
    """The generator form (fy_llm_begin + fy_llm_step in uneven pieces, as stream=True drives it) emits the ids of one
    fy_llm_generate call - against the reference's fixture, batched, greedy and repetition-aware sampling."""
    from oracle.llm import silent_filter
    cfg = tiny.cfg
    cases = [(12, 8, 0), (16, 4, 10)]
    texts, ptexts, ptoks = (list(x) for x in zip(*[llm_case(cfg, *c, "%d_%d_%d" % c) for c in cases]))
    cap = [min(20 * len(t), 400) for t in texts]
    for kind, fx in (("greedy", "llm_tiny.npz"), ("ras", "llm_ras_tiny.npz")):
        if kind == "ras":
            tiny.set_sampler("ras", _ras_rows(cases, tiny.max_batch))
        try:
            whole, whole_n, _ = tiny.generate(texts, ptexts, ptoks, max_len=cap)
            whole, whole_n = whole.cpu().clone(), whole_n.cpu().tolist()
            out, _, _ = tiny.begin(texts, ptexts, ptoks, max_len=cap)
            n, fin = tiny.step(0)
            assert not any(fin) and max(n) <= 1
            pieces = [1, 7, 2, 13, 5, 64]
            i = 0
            while not all(fin):
                n, fin = tiny.step(pieces[i % len(pieces)])
                i += 1
                assert i < 400
            assert n == whole_n, (kind, n, whole_n)
            out = out.cpu()
            for b in range(len(cases)):
                assert out[b, : n[b]].tolist() == whole[b, : n[b]].tolist(), (kind, b)
            f = golden(fx)
            c = cases[0]
            ref = silent_filter(f["c%d_%d_%d.tokens" % c].tolist())
            assert out[0, : n[0]].tolist() == ref
            n2, fin2 = tiny.step(5)                          # stepping a finished generation is a no-op
            assert n2 == n and all(fin2)
        finally:
            tiny.set_sampler("greedy")


def test_prefill_from_embeddings_equals_begin(tiny):
    """fy_llm_prefill - the embeddings-level entry shaped like the reference's vLLM hand-off (llm.py:482-510: the host passes
    `prompt_embeds` = lm_input (L, hidden)) - against fy_llm_begin on the same sequences: the host assembles lm_input =
    [speech_embedding[sos], embed_tokens(prompt_text + text), speech_embedding[task_id], speech_embedding(prompt_speech)]
    (llm.py:728-740) from the state dict, fp32 and bf16 (embed_tokens is bf16 in the engine, so fp32 rows carry the same values;
    bf16 rows round the fp32 speech embeddings, which the synthetic weights make bf16-exact) - identical ids, batched, and the
    reference fixture's ids for case 0."""
    from oracle.llm import silent_filter
    cfg = tiny.cfg
    cases = [(12, 8, 0), (16, 4, 10)]
    texts, ptexts, ptoks = (list(x) for x in zip(*[llm_case(cfg, *c, "%d_%d_%d" % c) for c in cases]))
    cap = [min(20 * len(t), 400) for t in texts]
    mn = [2 * len(t) for t in texts]
    whole, whole_n, _ = tiny.generate(texts, ptexts, ptoks, max_len=cap)
    whole, whole_n = whole.cpu().clone(), whole_n.cpu().tolist()
    sd = synth.state_dict_torch(cfg.manifest(), DEV, skip=("lm_head",))       # the deterministic synthetic weights the engine was built from
    emb_tok = sd["llm.model.model.embed_tokens.weight"].to(torch.bfloat16).float()       # the engine keeps embed_tokens in bf16
    emb_sp = sd["speech_embedding.weight"]
    for dt in (torch.float32, torch.bfloat16):
        rows = []
        for b in range(len(cases)):
            ids = torch.tensor(list(ptexts[b]) + list(texts[b]), dtype=torch.long, device=emb_tok.device)
            parts = [emb_sp[cfg.speech_tokens: cfg.speech_tokens + 1], emb_tok[ids], emb_sp[cfg.speech_tokens + 2: cfg.speech_tokens + 3]]
            if len(ptoks[b]):
                parts.append(emb_sp[torch.tensor(list(ptoks[b]), dtype=torch.long, device=emb_sp.device)])
            rows.append(torch.cat(parts, dim=0).to(dt))
        out, _, _ = tiny.prefill_embeds(rows, mn, cap)
        n, fin = tiny.step(0)
        while not all(fin):
            n, fin = tiny.step(50)
        assert n == whole_n, (dt, n, whole_n)
        out = out.cpu()
        for b in range(len(cases)):
            assert out[b, : n[b]].tolist() == whole[b, : n[b]].tolist(), (dt, b)
    f = golden("llm_tiny.npz")
    assert out[0, : n[0]].tolist() == silent_filter(f["c%d_%d_%d.tokens" % cases[0]].tolist())


def _ras_rows(cases, n_rows):
    u = np.zeros((n_rows, 4096), dtype=np.float32)
    for b, c in enumerate(cases):
        u[b] = synth.uniform("in.llm.ras_u.%d_%d_%d" % c, (4096,), 0.0, 1.0)
    return torch.from_numpy(u).to(DEV)


def test_ras_tokens_against_reference(tiny):
    """The reference's default sampler (ras_sampling inside sampling_ids) on the device, multinomial draws from the supplied
    uniforms: ids identical to the reference's own run (tests/golden/llm_ras_tiny.npz), batched; the reference's give-up
    RuntimeError surfaces with its message; greedy decoding is back afterwards."""
    from oracle.llm import silent_filter
    f = golden("llm_ras_tiny.npz")
    assert f is not None
    cfg = tiny.cfg
    cases = [(12, 8, 0), (16, 4, 10), (30, 5, 0)]
    texts, ptexts, ptoks = zip(*[llm_case(cfg, *c, "%d_%d_%d" % c) for c in cases])
    tiny.set_sampler("ras", _ras_rows(cases, tiny.max_batch))
    try:
        out, out_n, raw_n = tiny.generate(list(texts), list(ptexts), list(ptoks), max_len=[min(20 * len(t), 400) for t in texts])
        out, out_n = out.cpu(), out_n.cpu().tolist()
        for b, c in enumerate(cases):
            ref = silent_filter(f["c%d_%d_%d.tokens" % c].tolist())
            got = out[b, : out_n[b]].tolist()
            first_bad = next((i for i, (g, r) in enumerate(zip(got, ref)) if g != r), None)
            note("parity_llm.json", f"ras.tiny.{c}.n", [len(got), len(ref), first_bad])
            assert got == ref, (c, first_bad, got[:10], ref[:10])
        bad = (10, 6, 30)
        assert "max_trials" in str(f["c%d_%d_%d.raised" % bad])
        t, pt, pk = llm_case(cfg, *bad, "%d_%d_%d" % bad)
        tiny.set_sampler("ras", _ras_rows([bad], tiny.max_batch))
        with pytest.raises(RuntimeError, match="sampling reaches max_trials 100"):
            tiny.generate([t], [pt], [pk])
    finally:
        tiny.set_sampler("greedy")
    f0 = golden("llm_tiny.npz")
    run_cases(tiny, f0, [(12, 8, 0)], None, "tiny_after_ras")


def test_ras_full_size(full):
    f = golden("llm_ras_full.npz")
    if f is None:
        pytest.skip("llm_ras_full.npz not minted")
    from oracle.llm import silent_filter
    c = (12, 8, 0)
    t, pt, pk = llm_case(full.cfg, *c, "%d_%d_%d" % c)
    full.set_sampler("ras", _ras_rows([c], full.max_batch))
    try:
        if str(f["c%d_%d_%d.raised" % c]):
            with pytest.raises(RuntimeError, match="sampling reaches max_trials 100"):
                full.generate([t], [pt], [pk], max_len=[60])
        else:
            out, out_n, _ = full.generate([t], [pt], [pk], max_len=[60])
            ref = silent_filter(f["c%d_%d_%d.tokens" % c].tolist())[:60]
            assert out[0, : int(out_n[0])].cpu().tolist() == ref
    finally:
        full.set_sampler("greedy")


@pytest.mark.parametrize("M,N,K", [(800, 1152, 896), (296, 896, 4864), (37, 256, 256)])
def test_exact_split_products_ring_and_register_staged(M, N, K):
    """The LM prefill's GEMMs (round 4: three bf16 planes per stage on the LDS-DMA ring kernel, `gemm_exact3`; round 2's register-staged
    kernel is the fallback): with bf16 weights every product of the exact split x = hi + mid + lo is exact and the sums are fp32, so
    both forms sit within fp32 accumulation error of the float64 product - rows that do not fill a tile, K = 4864 and a wide
    dynamic range in A included - and agree with each other to the same bound."""
    from fangyan_tts_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(M * 7 + N)
    A = (torch.randn(M, K, generator=g) * torch.exp(3.0 * torch.randn(M, 1, generator=g))).to(DEV)
    W = torch.randn(N, K, generator=g).to(torch.bfloat16).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    ref = A.double() @ W.double().t() + bias.double()
    bound = (A.abs().double() @ W.abs().double().t()) * (K * 2.0 ** -24) + 1e-30          # fp32 accumulation of K exact products
    outs = []
    for ring in (1, 0):
        out = torch.full((M, N), float("nan"), device=DEV)
        planes = torch.empty(3 * M * K, dtype=torch.bfloat16, device=DEV)
        _lib.check(L.fy_debug_gemm_exact(A.data_ptr(), W.data_ptr(), M, N, K, bias.data_ptr(), out.data_ptr(), ring, planes.data_ptr(), None))
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        assert bool(((out.double() - ref).abs() <= bound).all()), (ring, float(((out.double() - ref).abs() / bound).max()))
        outs.append(out)
    assert bool(((outs[0].double() - outs[1].double()).abs() <= 2 * bound).all())
