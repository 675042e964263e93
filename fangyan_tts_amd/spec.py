"""Architecture constants and weight manifests of the CosyVoice3-0.5B hot path.

The numbers are the ones the reference constructs its modules with
(CosyVoice/examples/dialect/cosyvoice3/conf/cosyvoice3.yaml:8-100); the weight
names are the reference's state_dict keys (llm.pt / flow.pt / hift.pt), so a
real checkpoint and the synthetic generator (synth.py) feed the same loader.
``tests/golden/mint_goldens.py`` checks every manifest against the state_dict
of the reference modules, key by key and shape by shape.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

Manifest = Dict[str, Tuple[int, ...]]

SAMPLE_RATE = 24000          # cosyvoice3.yaml:8
TOKEN_MEL_RATIO = 2          # cosyvoice3.yaml:14
# FSQ silent / breath tokens, CosyVoice/cosyvoice/cli/model.py:414
SILENT_TOKENS = (1, 2, 28, 29, 55, 248, 494, 2241, 2242, 2322, 2323)
MAX_SILENT_RUN = 5           # cli/model.py:102


@dataclass(frozen=True)
class LlmCfg:
    """Qwen2 body + CosyVoice3LM heads (llm/llm.py:641-668)."""
    hidden: int = 896
    layers: int = 24
    q_heads: int = 14
    kv_heads: int = 2
    head_dim: int = 64
    inter: int = 4864
    vocab: int = 151936
    speech_tokens: int = 6561     # speech_token_size
    rms_eps: float = 1e-6
    rope_theta: float = 1e6

    @property
    def n_speech(self) -> int:    # llm_decoder / speech_embedding rows
        return self.speech_tokens + 200

    @property
    def sos(self) -> int:
        return self.speech_tokens

    @property
    def eos(self) -> int:
        return self.speech_tokens + 1

    @property
    def task_id(self) -> int:
        return self.speech_tokens + 2

    @staticmethod
    def tiny() -> "LlmCfg":
        return LlmCfg(hidden=256, layers=2, q_heads=4, kv_heads=2, inter=512, vocab=1000)

    def manifest(self) -> Manifest:
        h, kv = self.hidden, self.kv_heads * self.head_dim
        q = self.q_heads * self.head_dim
        m: Manifest = {"llm.model.model.embed_tokens.weight": (self.vocab, h)}
        for i in range(self.layers):
            p = f"llm.model.model.layers.{i}."
            m[p + "self_attn.q_proj.weight"] = (q, h)
            m[p + "self_attn.q_proj.bias"] = (q,)
            m[p + "self_attn.k_proj.weight"] = (kv, h)
            m[p + "self_attn.k_proj.bias"] = (kv,)
            m[p + "self_attn.v_proj.weight"] = (kv, h)
            m[p + "self_attn.v_proj.bias"] = (kv,)
            m[p + "self_attn.o_proj.weight"] = (h, q)
            m[p + "mlp.gate_proj.weight"] = (self.inter, h)
            m[p + "mlp.up_proj.weight"] = (self.inter, h)
            m[p + "mlp.down_proj.weight"] = (h, self.inter)
            m[p + "input_layernorm.weight"] = (h,)
            m[p + "post_attention_layernorm.weight"] = (h,)
        m["llm.model.model.norm.weight"] = (h,)
        # tied to embed_tokens and never used by inference
        m["llm.model.lm_head.weight"] = (self.vocab, h)
        m["llm_decoder.weight"] = (self.n_speech, h)
        m["speech_embedding.weight"] = (self.n_speech, h)
        return m


@dataclass(frozen=True)
class FlowCfg:
    """CausalMaskedDiffWithDiT + CausalConditionalCFM + DiT (cosyvoice3.yaml:38-75)."""
    mel: int = 80
    spk_in: int = 192
    vocab: int = 6561
    pre_ch: int = 1024            # PreLookaheadLayer channels
    pre_lookahead: int = 3
    dim: int = 1024
    depth: int = 22
    heads: int = 16
    head_dim: int = 64
    ff_mult: int = 2
    conv_pos_k: int = 31
    conv_pos_groups: int = 16
    n_timesteps: int = 10         # flow/flow.py:398
    cfg_rate: float = 0.7         # inference_cfg_rate
    static_chunk: int = 50        # chunk_size * token_mel_ratio
    noise_len: int = 50 * 300     # rand_noise, flow_matching.py:200

    @staticmethod
    def tiny() -> "FlowCfg":
        return FlowCfg(dim=256, depth=2, heads=4)

    def manifest(self) -> Manifest:
        d = self.dim
        m: Manifest = {
            "input_embedding.weight": (self.vocab, self.mel),
            "spk_embed_affine_layer.weight": (self.mel, self.spk_in),
            "spk_embed_affine_layer.bias": (self.mel,),
            "pre_lookahead_layer.conv1.weight": (self.pre_ch, self.mel, self.pre_lookahead + 1),
            "pre_lookahead_layer.conv1.bias": (self.pre_ch,),
            "pre_lookahead_layer.conv2.weight": (self.mel, self.pre_ch, 3),
            "pre_lookahead_layer.conv2.bias": (self.mel,),
        }
        e = "decoder.estimator."
        m[e + "time_embed.time_mlp.0.weight"] = (d, 256)
        m[e + "time_embed.time_mlp.0.bias"] = (d,)
        m[e + "time_embed.time_mlp.2.weight"] = (d, d)
        m[e + "time_embed.time_mlp.2.bias"] = (d,)
        m[e + "input_embed.proj.weight"] = (d, 4 * self.mel)
        m[e + "input_embed.proj.bias"] = (d,)
        for c in ("conv1", "conv2"):
            m[e + f"input_embed.conv_pos_embed.{c}.0.weight"] = (d, d // self.conv_pos_groups, self.conv_pos_k)
            m[e + f"input_embed.conv_pos_embed.{c}.0.bias"] = (d,)
        inner = self.heads * self.head_dim
        for i in range(self.depth):
            b = e + f"transformer_blocks.{i}."
            m[b + "attn_norm.linear.weight"] = (6 * d, d)
            m[b + "attn_norm.linear.bias"] = (6 * d,)
            for n in ("to_q", "to_k", "to_v"):
                m[b + f"attn.{n}.weight"] = (inner, d)
                m[b + f"attn.{n}.bias"] = (inner,)
            m[b + "attn.to_out.0.weight"] = (d, inner)
            m[b + "attn.to_out.0.bias"] = (d,)
            m[b + "ff.ff.0.0.weight"] = (d * self.ff_mult, d)
            m[b + "ff.ff.0.0.bias"] = (d * self.ff_mult,)
            m[b + "ff.ff.2.weight"] = (d, d * self.ff_mult)
            m[b + "ff.ff.2.bias"] = (d,)
        m[e + "norm_out.linear.weight"] = (2 * d, d)
        m[e + "norm_out.linear.bias"] = (2 * d,)
        m[e + "proj_out.weight"] = (self.mel, d)
        m[e + "proj_out.bias"] = (self.mel,)
        return m


@dataclass(frozen=True)
class HiftCfg:
    """CausalHiFTGenerator + CausalConvRNNF0Predictor (cosyvoice3.yaml:77-100)."""
    mel: int = 80
    base: int = 512
    harmonics: int = 8
    nsf_alpha: float = 0.1
    nsf_sigma: float = 0.003
    voiced_thr: float = 10.0
    ups: Tuple[int, ...] = (8, 5, 3)
    up_k: Tuple[int, ...] = (16, 11, 7)
    n_fft: int = 16
    hop: int = 4
    rb_k: Tuple[int, ...] = (3, 7, 11)
    rb_d: Tuple[int, ...] = (1, 3, 5)
    src_rb_k: Tuple[int, ...] = (7, 7, 11)
    lrelu: float = 0.1
    audio_limit: float = 0.99
    pre_look_right: int = 4
    f0_ch: int = 512
    noise_len: int = 300 * 24000  # SineGen2.sine_waves / SourceModuleHnNSF.uv

    @property
    def upsample_total(self) -> int:      # samples per mel frame = 480
        t = self.hop
        for u in self.ups:
            t *= u
        return t

    @property
    def stft_per_frame(self) -> int:      # STFT columns per mel frame = 120
        return self.upsample_total // self.hop

    def stage_ch(self, i: int) -> int:
        return self.base // (2 ** (i + 1))

    @staticmethod
    def tiny() -> "HiftCfg":
        return HiftCfg(base=128, f0_ch=64)

    def source_down(self, i: int) -> Tuple[int, int]:
        """(kernel, stride) of source_downs[i] (generator.py:642-652)."""
        rates = [1] + list(self.ups[::-1][:-1])
        cum = [1]
        for r in rates[1:]:
            cum.append(cum[-1] * r)
        u = cum[::-1][i]
        return (1, 1) if u == 1 else (u * 2, u)

    def manifest(self) -> Manifest:
        m: Manifest = {
            "m_source.l_linear.weight": (1, self.harmonics + 1),
            "m_source.l_linear.bias": (1,),
        }

        def wn(prefix: str, co: int, ci: int, k: int):
            m[prefix + ".bias"] = (co,)
            m[prefix + ".parametrizations.weight.original0"] = (co, 1, 1)
            m[prefix + ".parametrizations.weight.original1"] = (co, ci, k)

        def resblock(prefix: str, ch: int, k: int):
            for j in range(len(self.rb_d)):
                wn(f"{prefix}.convs1.{j}", ch, ch, k)
            for j in range(len(self.rb_d)):
                wn(f"{prefix}.convs2.{j}", ch, ch, k)
            for j in range(len(self.rb_d)):
                m[f"{prefix}.activations1.{j}.alpha"] = (ch,)
            for j in range(len(self.rb_d)):
                m[f"{prefix}.activations2.{j}.alpha"] = (ch,)

        wn("conv_pre", self.base, self.mel, self.pre_look_right + 1)
        for i, k in enumerate(self.up_k):
            wn(f"ups.{i}", self.stage_ch(i), self.base // (2 ** i), k)
        for i in range(len(self.ups)):
            k, _ = self.source_down(i)
            m[f"source_downs.{i}.weight"] = (self.stage_ch(i), self.n_fft + 2, k)
            m[f"source_downs.{i}.bias"] = (self.stage_ch(i),)
        for i in range(len(self.ups)):
            resblock(f"source_resblocks.{i}", self.stage_ch(i), self.src_rb_k[i])
        for i in range(len(self.ups)):
            for j, k in enumerate(self.rb_k):
                resblock(f"resblocks.{i * len(self.rb_k) + j}", self.stage_ch(i), k)
        wn("conv_post", self.n_fft + 2, self.stage_ch(len(self.ups) - 1), 7)
        wn("f0_predictor.condnet.0", self.f0_ch, self.mel, 4)
        for i in (2, 4, 6, 8):
            wn(f"f0_predictor.condnet.{i}", self.f0_ch, self.f0_ch, 3)
        m["f0_predictor.classifier.weight"] = (1, self.f0_ch)
        m["f0_predictor.classifier.bias"] = (1,)
        return m


@dataclass(frozen=True)
class ModelCfg:
    llm: LlmCfg = field(default_factory=LlmCfg)
    flow: FlowCfg = field(default_factory=FlowCfg)
    hift: HiftCfg = field(default_factory=HiftCfg)

    @staticmethod
    def tiny() -> "ModelCfg":
        return ModelCfg(LlmCfg.tiny(), FlowCfg.tiny(), HiftCfg.tiny())
