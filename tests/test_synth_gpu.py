"""The synthetic-weight generator on the device (fy_synth_uniform, csrc/runtime.hip) is bit-identical to the numpy form the
oracle and the reference fixtures were filled from (fangyan_tts_amd/synth.py) - every rule kind, and a whole state dict."""
import numpy as np
import pytest
import torch

from fangyan_tts_amd import synth
from fangyan_tts_amd.spec import ModelCfg

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("name,shape", [
    ("llm.model.model.layers.3.mlp.gate_proj.weight", (512, 256)),           # w: bf16-rounded uniform, gain by fan-in
    ("f0_predictor.condnet.0.parametrizations.weight.original1", (300, 80, 3)),   # wf32: not rounded
    ("speech_embedding.weight", (700, 128)),                                  # emb
    ("llm.model.model.layers.0.input_layernorm.weight", (70000,)),            # one: 1 + uniform
    ("decoder.estimator.transformer_blocks.1.ff.ff.0.0.bias", (66000,)),      # b
    ("resblocks.0.activations1.0.alpha", (65536,)),                           # range
])
def test_device_fill_equals_numpy(name, shape):
    want = synth.tensor(name, shape)
    got = synth._tensor_device(name, shape, DEV)
    assert got.shape == want.shape and got.dtype == torch.float32
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)), name


def test_state_dict_on_the_device_equals_numpy():
    cfg = ModelCfg.tiny()
    for m in (cfg.llm, cfg.flow, cfg.hift):
        man = m.manifest()
        a = synth.state_dict(man, skip=("lm_head",))
        b = synth.state_dict_torch(man, DEV, skip=("lm_head",))
        assert set(a) == set(b)
        for k in a:
            assert np.array_equal(b[k].cpu().numpy().view(np.uint32), a[k].view(np.uint32)), k
