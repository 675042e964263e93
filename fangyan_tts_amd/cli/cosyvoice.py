"""The reference's user-facing surface, `cosyvoice.cli.cosyvoice.{AutoModel, CosyVoice3}`
(CosyVoice/cosyvoice/cli/cosyvoice.py:191-240), over the MI355X engines.

    model = AutoModel(model_dir=...)                 # compare_inference.py:31-34
    for out in model.inference_instruct2(text, instruct, prompt_wav, stream=False): out['tts_speech']
    model.model.llm.load_state_dict(sd, strict=False)   # compare_inference.py:42
    model.sample_rate

Model dir layout is the reference's: cosyvoice3.yaml, llm.pt, flow.pt, hift.pt (+ the frontend's tokenizer / ONNX
files).  Architecture sizes are read off the checkpoint tensors themselves; the yaml contributes only scalars.

The text / audio frontend is fangyan_tts_amd/cli/frontend.py (`CosyVoiceFrontEnd`: text normalisation + splitting, the
model_input dicts, the prompt mel on the GPU).  Built by default from <model_dir>/CosyVoice-BlankEN (Qwen tokenizer files)
and spk2info.pt; the two ONNX models of the reference's frontend (speech tokenizer, campplus x-vector) need onnxruntime,
which this image lacks: pass them as callables (`speech_tokenizer=`, `spk_embedder=`) or pass a whole `frontend=` object.
Everything from `model_input` on runs on the GPU engines.
"""
from __future__ import annotations

import logging
import os
import time
from typing import Dict, Generator, Optional

import torch
import yaml

from ..spec import FlowCfg, HiftCfg, LlmCfg, ModelCfg
from .model import CosyVoice3Model


class _TolerantLoader(yaml.SafeLoader):
    """cosyvoice3.yaml is HyperPyYAML (!new: / !name: / !ref / !apply tags); only its scalars are needed here."""


def _any_tag(loader, suffix, node):
    if isinstance(node, yaml.MappingNode):
        return loader.construct_mapping(node, deep=True)
    if isinstance(node, yaml.SequenceNode):
        return loader.construct_sequence(node, deep=True)
    return loader.construct_scalar(node)


_TolerantLoader.add_multi_constructor("!", _any_tag)


def read_yaml_scalars(path: str) -> Dict:
    with open(path, "r") as f:
        return yaml.load(f, Loader=_TolerantLoader) or {}


def _load_pt(path: str) -> Dict[str, torch.Tensor]:
    sd = torch.load(path, map_location="cpu", weights_only=True)
    # train checkpoints carry bookkeeping entries (compare_inference.py:39-41)
    return {k: v for k, v in sd.items() if isinstance(v, torch.Tensor) and not k.startswith(("epoch", "step"))}


def infer_cfg(llm_sd, flow_sd, hift_sd, conf: Dict) -> ModelCfg:
    """Architecture sizes from the tensors' shapes (the yaml does not hold the Qwen2 sizes at all)."""
    emb = llm_sd["llm.model.model.embed_tokens.weight"]
    n_layers = 1 + max(int(k.split(".")[4]) for k in llm_sd if k.startswith("llm.model.model.layers."))
    p0 = "llm.model.model.layers.0."
    llm = LlmCfg(hidden=emb.shape[1], layers=n_layers, q_heads=llm_sd[p0 + "self_attn.q_proj.weight"].shape[0] // 64,
                 kv_heads=llm_sd[p0 + "self_attn.k_proj.weight"].shape[0] // 64, inter=llm_sd[p0 + "mlp.gate_proj.weight"].shape[0],
                 vocab=emb.shape[0], speech_tokens=llm_sd["llm_decoder.weight"].shape[0] - 200)
    e = "decoder.estimator."
    dim = flow_sd[e + "proj_out.weight"].shape[1]
    depth = 1 + max(int(k.split(".")[3]) for k in flow_sd if k.startswith(e + "transformer_blocks."))
    fconf = ((conf.get("flow") or {}).get("decoder") or {})
    cfg_rate = float((((fconf.get("cfm_params") or {}).get("content") or {}).get("inference_cfg_rate", 0.7)))
    flow = FlowCfg(spk_in=flow_sd["spk_embed_affine_layer.weight"].shape[1], vocab=flow_sd["input_embedding.weight"].shape[0],
                   pre_ch=flow_sd["pre_lookahead_layer.conv1.weight"].shape[0],
                   pre_lookahead=flow_sd["pre_lookahead_layer.conv1.weight"].shape[2] - 1, dim=dim, depth=depth, heads=dim // 64,
                   ff_mult=flow_sd[e + "transformer_blocks.0.ff.ff.0.0.weight"].shape[0] // dim,
                   conv_pos_k=flow_sd[e + "input_embed.conv_pos_embed.conv1.0.weight"].shape[2],
                   conv_pos_groups=dim // flow_sd[e + "input_embed.conv_pos_embed.conv1.0.weight"].shape[1], cfg_rate=cfg_rate)
    hconf = conf.get("hift") or {}
    ups = tuple(hconf.get("upsample_rates", (8, 5, 3)))
    v = ".parametrizations.weight.original1"
    hift = HiftCfg(base=hift_sd["conv_pre" + v].shape[0], harmonics=hift_sd["m_source.l_linear.weight"].shape[1] - 1, ups=ups,
                   up_k=tuple(hift_sd[f"ups.{i}" + v].shape[2] for i in range(3)),
                   rb_k=tuple(hift_sd[f"resblocks.{j}.convs1.0" + v].shape[2] for j in range(3)),
                   src_rb_k=tuple(hift_sd[f"source_resblocks.{i}.convs1.0" + v].shape[2] for i in range(3)),
                   pre_look_right=hift_sd["conv_pre" + v].shape[2] - 1, f0_ch=hift_sd["f0_predictor.condnet.0" + v].shape[0],
                   nsf_alpha=float(hconf.get("nsf_alpha", 0.1)), nsf_sigma=float(hconf.get("nsf_sigma", 0.003)),
                   voiced_thr=float(hconf.get("nsf_voiced_threshold", 10)), lrelu=float(hconf.get("lrelu_slope", 0.1)),
                   audio_limit=float(hconf.get("audio_limit", 0.99)))
    return ModelCfg(llm, flow, hift)


class _LazyFrontEndError:
    """Stands in when the default frontend could not be built (no tokenizer files in the model dir): the reason is raised at the
    first call that needs the frontend, as the reference would fail in CosyVoiceFrontEnd.__init__."""
    spk2info: Dict = {}

    def __init__(self, err: BaseException):
        self._err = err

    def __getattr__(self, name):
        raise RuntimeError(f"the frontend could not be built: {self._err}; pass frontend=<object with text_normalize / "
                           "frontend_zero_shot / frontend_instruct2 / ...> or a tokenizer")


class CosyVoice3:
    def __init__(self, model_dir, load_trt=False, load_vllm=False, fp16=False, trt_concurrent=1, frontend=None, tokenizer=None,
                 speech_tokenizer=None, spk_embedder=None,
                 max_batch: int = 1, max_tokens: int = 1600, max_prompt_tokens: int = 750, max_text: int = 160, device=None,
                 sampler: str = "ras", concurrency: int = 1):
        """Capacities (the engines are sized once): max_text bounds prompt-text + text ids of a segment (text_normalize cuts
        segments at 80 tokens, cli/frontend.py:151), max_tokens the generated speech tokens (the LM's own bound is 20 x the
        text length, llm.py:744: 1600 for an 80-token segment), max_prompt_tokens the prompt (30 s = 750 tokens,
        cli/frontend.py:97).  concurrency > 1 builds that many independent engine sets so that many threads can be inside
        tts() at once (runtime/python/grpc/server.py:68-69); with 1, concurrent calls take turns."""
        self.model_dir, self.fp16 = model_dir, fp16
        if not os.path.exists(model_dir):
            raise ValueError("{} not found (no network here: the reference would call snapshot_download)".format(model_dir))
        hyper_yaml_path = "{}/cosyvoice3.yaml".format(model_dir)
        if not os.path.exists(hyper_yaml_path):
            raise ValueError("{} not found!".format(hyper_yaml_path))
        conf = read_yaml_scalars(hyper_yaml_path)
        if load_trt or load_vllm:
            logging.warning("load_trt / load_vllm select NVIDIA engines in the reference; ignored (the HIP engines always run)")
        self.sample_rate = int(conf.get("sample_rate", 24000))
        sds = [_load_pt("{}/{}.pt".format(model_dir, n)) for n in ("llm", "flow", "hift")]
        sds[2] = {k.replace("generator.", ""): v for k, v in sds[2].items()}          # cli/model.py:71
        self.cfg = infer_cfg(*sds, conf)
        assert self.cfg.llm.speech_tokens == self.cfg.flow.vocab, "llm.pt and flow.pt disagree on the speech vocabulary"
        dev = device or torch.device("cuda", torch.cuda.current_device())
        if frontend is None:
            # cli/cosyvoice.py:204-209: CosyVoiceFrontEnd(get_tokenizer, feat_extractor, campplus.onnx, speech_tokenizer_v3.onnx, spk2info.pt)
            from .frontend import CosyVoiceFrontEnd, load_qwen_tokenizer, onnx_prompt_models
            if speech_tokenizer is None and spk_embedder is None and dev.type == "cuda":
                # the reference's own two ONNX sessions when onnxruntime imports and the files are there (cli/frontend.py:41-46);
                # injected callables take precedence
                speech_tokenizer, spk_embedder = onnx_prompt_models(model_dir, dev)
            try:
                frontend = CosyVoiceFrontEnd(tokenizer if tokenizer is not None else load_qwen_tokenizer(model_dir),
                                             speech_tokenizer=speech_tokenizer, spk_embedder=spk_embedder,
                                             spk2info="{}/spk2info.pt".format(model_dir), device=dev)
            except (FileNotFoundError, OSError) as e:
                frontend = _LazyFrontEndError(e)
        self.frontend = frontend
        to_dev = lambda sd: {k: v.to(dev, torch.float32).contiguous() for k, v in sd.items() if "lm_head" not in k}
        self.model = CosyVoice3Model(to_dev(sds[0]), to_dev(sds[1]), to_dev(sds[2]), self.cfg, device=dev, max_batch=max_batch,
                                     max_tokens=max_tokens, max_prompt_tokens=max_prompt_tokens, max_text=max_text, fp16=fp16,
                                     keep_llm_weights=True, sampler=sampler, concurrency=concurrency)

    # ---- speaker bookkeeping (cli/cosyvoice.py:64-78) ------------------------------------------------
    def list_available_spks(self):
        return list(self.frontend.spk2info.keys())

    def add_zero_shot_spk(self, prompt_text, prompt_wav, zero_shot_spk_id):
        assert zero_shot_spk_id != "", "do not use empty zero_shot_spk_id"
        model_input = self.frontend.frontend_zero_shot("", prompt_text, prompt_wav, self.sample_rate, "")
        del model_input["text"]
        del model_input["text_len"]
        self.frontend.spk2info[zero_shot_spk_id] = model_input
        return True

    def save_spkinfo(self):
        torch.save(self.frontend.spk2info, "{}/spk2info.pt".format(self.model_dir))

    # ---- inference generators -----------------------------------------------------------------------------
    def _run(self, model_input, text, stream, speed) -> Generator[Dict[str, torch.Tensor], None, None]:
        start_time = time.time()
        logging.info("synthesis text {}".format(text))
        for model_output in self.model.tts(**model_input, stream=stream, speed=speed):
            speech_len = model_output["tts_speech"].shape[1] / self.sample_rate
            logging.info("yield speech len {}, rtf {}".format(speech_len, (time.time() - start_time) / speech_len))
            yield model_output
            start_time = time.time()

    def inference_sft(self, tts_text, spk_id, stream=False, speed=1.0, text_frontend=True):
        """cli/cosyvoice.py:80-89: a stored speaker embedding, no prompt audio (prompt lengths 0 on the flow side)."""
        for i in self.frontend.text_normalize(tts_text, split=True, text_frontend=text_frontend):
            model_input = self.frontend.frontend_sft(i, spk_id)
            yield from self._run(model_input, i, stream, speed)

    def inference_zero_shot(self, tts_text, prompt_text, prompt_wav, zero_shot_spk_id="", stream=False, speed=1.0, text_frontend=True):
        if "<|endofprompt|>" not in prompt_text + tts_text:
            logging.warning("<|endofprompt|> not found in CosyVoice3 inference, check your input text")
        prompt_text = self.frontend.text_normalize(prompt_text, split=False, text_frontend=text_frontend)
        for i in self.frontend.text_normalize(tts_text, split=True, text_frontend=text_frontend):
            model_input = self.frontend.frontend_zero_shot(i, prompt_text, prompt_wav, self.sample_rate, zero_shot_spk_id)
            yield from self._run(model_input, i, stream, speed)

    def inference_cross_lingual(self, tts_text, prompt_wav, zero_shot_spk_id="", stream=False, speed=1.0, text_frontend=True):
        for i in self.frontend.text_normalize(tts_text, split=True, text_frontend=text_frontend):
            model_input = self.frontend.frontend_cross_lingual(i, prompt_wav, self.sample_rate, zero_shot_spk_id)
            yield from self._run(model_input, i, stream, speed)

    def inference_instruct2(self, tts_text, instruct_text, prompt_wav, zero_shot_spk_id="", stream=False, speed=1.0, text_frontend=True):
        for i in self.frontend.text_normalize(tts_text, split=True, text_frontend=text_frontend):
            model_input = self.frontend.frontend_instruct2(i, instruct_text, prompt_wav, self.sample_rate, zero_shot_spk_id)
            yield from self._run(model_input, i, stream, speed)

    def inference_instruct(self, *a, **kw):
        raise AssertionError("inference_instruct is only implemented for CosyVoice!")           # cli/cosyvoice.py:118-119

    def inference_vc(self, source_wav, prompt_wav, stream=False, speed=1.0):
        """cli/cosyvoice.py:131-138: the source's speech tokens go straight to the flow decoder + vocoder (no LM)."""
        model_input = self.frontend.frontend_vc(source_wav, prompt_wav, self.sample_rate)
        yield from self._run(model_input, "<voice conversion>", stream, speed)


def AutoModel(**kwargs):
    """cli/cosyvoice.py:230-240: picks the class from the yaml present in model_dir; only CosyVoice3 is built."""
    model_dir = kwargs["model_dir"]
    if not os.path.exists(model_dir):
        raise ValueError("{} not found (no network here: the reference would call snapshot_download)".format(model_dir))
    if os.path.exists("{}/cosyvoice3.yaml".format(model_dir)):
        return CosyVoice3(**kwargs)
    if os.path.exists("{}/cosyvoice.yaml".format(model_dir)) or os.path.exists("{}/cosyvoice2.yaml".format(model_dir)):
        raise TypeError("CosyVoice 1/2 model dirs are outside this build (CosyVoice3 only)")
    raise TypeError("No valid model type found!")
