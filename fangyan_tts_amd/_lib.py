"""ctypes binding of libfy_cosy3.so (include/fy_cosy3.h).

The product path has no CPU fallback: if the library is missing this raises,
and every call checks the status code and raises FyError with fy_last_error().
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfy_cosy3.so")


class FyError(RuntimeError):
    pass


class FyTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


class HiftConfig(C.Structure):
    _fields_ = [("mel", C.c_int32), ("base", C.c_int32), ("harmonics", C.c_int32), ("sampling_rate", C.c_int32),
                ("nsf_alpha", C.c_float), ("nsf_sigma", C.c_float), ("voiced_thr", C.c_float),
                ("ups", C.c_int32 * 3), ("up_k", C.c_int32 * 3), ("n_fft", C.c_int32), ("hop", C.c_int32),
                ("rb_k", C.c_int32 * 3), ("rb_d", C.c_int32 * 3), ("src_rb_k", C.c_int32 * 3),
                ("lrelu", C.c_float), ("audio_limit", C.c_float), ("pre_look_right", C.c_int32), ("f0_ch", C.c_int32)]


class FlowConfig(C.Structure):
    _fields_ = [("mel", C.c_int32), ("spk_in", C.c_int32), ("vocab", C.c_int32), ("pre_ch", C.c_int32),
                ("pre_lookahead", C.c_int32), ("dim", C.c_int32), ("depth", C.c_int32), ("heads", C.c_int32),
                ("head_dim", C.c_int32), ("ff_mult", C.c_int32), ("conv_pos_k", C.c_int32), ("conv_pos_groups", C.c_int32),
                ("n_timesteps", C.c_int32), ("cfg_rate", C.c_float), ("static_chunk", C.c_int32), ("t_span", C.c_float * 33)]


class LlmConfig(C.Structure):
    _fields_ = [("hidden", C.c_int32), ("layers", C.c_int32), ("q_heads", C.c_int32), ("kv_heads", C.c_int32),
                ("head_dim", C.c_int32), ("inter", C.c_int32), ("vocab", C.c_int32), ("speech_tokens", C.c_int32),
                ("rms_eps", C.c_float), ("rope_theta", C.c_float), ("weight_planes", C.c_int32)]


FY_PRECISE = 1
FY_DIRECT = 2
FY_STREAMING = 4
FY_NO_FINALIZE = 8
FY_INCREMENTAL = 16
FY_LLM_KEEP_LOGP = 4

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FyError(f"{LIB_PATH} is missing: build it with `python -m fangyan_tts_amd.build` "
                          "(there is no CPU fallback for the product path)")
        _lib = C.CDLL(LIB_PATH)
        _lib.fy_last_error.restype = C.c_char_p
        _declare(_lib)
    return _lib


def check(rc: int):
    if rc != 0:
        raise FyError(f"libfy_cosy3 error {rc}: {lib().fy_last_error().decode(errors='replace')}")


def _declare(L):
    vp, i32, u32, f32p, i32p = C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.POINTER(C.c_int32)
    L.fy_version.restype = C.c_int
    L.fy_prof_enable.argtypes = [C.c_int]
    L.fy_prof_only.argtypes = [C.c_char_p]
    L.fy_prof_only.restype = None
    L.fy_prof_enable.restype = None
    L.fy_prof_reset.restype = None
    L.fy_prof_get.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.fy_prof_union.argtypes = [C.c_char_p, C.POINTER(C.c_double)]
    L.fy_llm_set_sampler.argtypes = [vp, i32, f32p, C.c_int64, i32, C.c_float, i32, C.c_float]
    L.fy_hift_default_config.argtypes = [C.POINTER(HiftConfig)]
    L.fy_hift_default_config.restype = None
    L.fy_hift_create.argtypes = [C.POINTER(vp), C.POINTER(HiftConfig), C.POINTER(FyTensor), i32, i32, i32, vp]
    L.fy_hift_destroy.argtypes = [vp]
    L.fy_hift_destroy.restype = None
    L.fy_hift_infer.argtypes = [vp, f32p, i32p, i32, i32, f32p, f32p, f32p, f32p, u32, vp]
    L.fy_hift_f0.argtypes = [vp, f32p, i32p, i32, i32, f32p, vp]
    L.fy_hift_source.argtypes = [vp, f32p, i32p, i32, i32, f32p, f32p, f32p, vp]
    L.fy_hift_decode.argtypes = [vp, f32p, f32p, i32p, i32, i32, f32p, u32, vp]
    L.fy_hift_tap.argtypes = [vp, C.c_char_p, f32p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), vp]
    L.fy_hift_resblock.argtypes = [vp, i32, f32p, f32p, i32, i32, u32, vp]
    L.fy_flow_default_config.argtypes = [C.POINTER(FlowConfig)]
    L.fy_flow_default_config.restype = None
    L.fy_flow_create.argtypes = [C.POINTER(vp), C.POINTER(FlowConfig), C.POINTER(FyTensor), i32, i32, i32, vp]
    L.fy_flow_destroy.argtypes = [vp]
    L.fy_flow_destroy.restype = None
    L.fy_flow_infer.argtypes = [vp, vp, i32, i32p, vp, i32, i32p, f32p, i32, i32p, f32p, f32p, i32, i32, f32p, i32, u32, vp]
    L.fy_flow_stream_reset.argtypes = [vp]
    L.fy_flow_stream_rows.argtypes = [vp]
    L.fy_dit_estimator.argtypes = [vp, f32p, f32p, f32p, f32p, f32p, f32p, i32, i32, u32, vp]
    L.fy_llm_default_config.argtypes = [C.POINTER(LlmConfig)]
    L.fy_llm_default_config.restype = None
    L.fy_llm_create.argtypes = [C.POINTER(vp), C.POINTER(LlmConfig), C.POINTER(FyTensor), i32, i32, i32, vp]
    L.fy_llm_destroy.argtypes = [vp]
    L.fy_llm_destroy.restype = None
    L.fy_llm_generate.argtypes = [vp, i32p, i32p, i32p, i32p, i32p, i32p, i32, vp, i32, vp, vp, u32, vp]
    L.fy_stream_overlap.argtypes = [C.POINTER(C.c_void_p), i32, vp]
    L.fy_mel_speed.argtypes = [vp, i32, i32, vp, i32, vp]
    L.fy_llm_begin.argtypes = [vp, i32p, i32p, i32p, i32p, i32p, i32p, i32, vp, i32, vp]
    L.fy_llm_prefill.argtypes = [vp, vp, i32, i32p, i32p, i32p, i32, vp, i32, vp]
    L.fy_llm_step.argtypes = [vp, i32, vp, i32, vp, vp, i32p, vp]
    L.fy_llm_logp.argtypes = [vp, i32, f32p, vp]
    L.fy_prompt_mel_create.argtypes = [C.POINTER(vp), i32, vp]
    L.fy_prompt_mel_destroy.argtypes = [vp]
    L.fy_prompt_mel_destroy.restype = None
    L.fy_prompt_mel_frames.argtypes = [i32]
    L.fy_prompt_mel_run.argtypes = [vp, vp, i32, vp, i32, vp]
    L.fy_allgather_audio_scratch_floats.argtypes = [i32, i32, i32]
    L.fy_allgather_audio_scratch_floats.restype = C.c_size_t
    L.fy_allgather_audio.argtypes = [vp, i32, vp, C.c_int64, vp, i32, i32, i32, vp, vp, vp, vp]
    L.fy_audio_record_pack.argtypes = [vp, C.c_int64, i32p, i32, i32, i32, vp, vp]
    L.fy_synth_uniform.argtypes = [vp, C.c_int64, C.c_uint64, C.c_int64, C.c_double, C.c_double, u32, vp]
    L.fy_audio_feat_create.argtypes = [C.POINTER(vp), i32, vp]
    L.fy_audio_feat_destroy.argtypes = [vp]
    L.fy_audio_feat_destroy.restype = None
    L.fy_audio_feat_mels.argtypes = [vp]
    L.fy_audio_feat_frames.argtypes = [vp, C.c_int64]
    L.fy_audio_feat_run.argtypes = [vp, vp, C.c_int64, vp, i32, u32, vp]
    L.fy_stream_create_masked.argtypes = [C.POINTER(vp), C.POINTER(C.c_uint32), i32]
    L.fy_stream_destroy.argtypes = [vp]
    L.fy_llm_set_decode_mode.argtypes = [vp, i32]
    L.fy_llm_decode_mode.argtypes = [vp]
    L.fy_llm_weight_planes.argtypes = [vp]
    L.fy_set_host_wait.argtypes = [i32, i32]
    L.fy_flow_weight_planes.argtypes = [vp]
    L.fy_debug_decode_stamps.argtypes = [vp, C.POINTER(C.c_uint64), i32, vp]
    L.fy_debug_decode32_stamps.argtypes = [vp, C.POINTER(C.c_uint64), i32, vp]
    L.fy_debug_gemm_exact.argtypes = [vp, vp, i32, i32, i32, vp, vp, i32, vp, vp]
    L.fy_debug_gemm_bf16.argtypes = [vp, vp, i32, i32, i32, vp, vp, i32, i32, vp, i32, vp]


def tensor_table(weights):
    """dict name -> contiguous fp32 CUDA torch tensor  ->  (FyTensor array, keep-alive list)."""
    import torch
    arr = (FyTensor * len(weights))()
    keep = []
    for i, (name, t) in enumerate(weights.items()):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise FyError(f"weight {name}: need a contiguous fp32 tensor on the GPU")
        nb = name.encode()
        keep.append((nb, t))
        arr[i].name = nb
        arr[i].data = t.data_ptr()
        arr[i].ndim = max(t.dim(), 1)
        for k, s in enumerate(t.shape):
            arr[i].shape[k] = s
    return arr, keep


HOST_WAIT_SLEEP_S = 2e-4          # stream_wait's poll interval; fy_set_host_wait gets the same figure


def host_wait_spin() -> bool:
    return os.environ.get("FY_HOST_WAIT_SPIN", "0") == "1"


def stream_wait(stream):
    """Wait for everything enqueued on a torch stream WITHOUT spinning on a core: an event polled with short sleeps (what
    `stream.synchronize()` costs under HIP's default schedule is a whole core per waiting thread; a pipelined rank has 3-4)."""
    if host_wait_spin():
        stream.synchronize()
        return
    import time
    import torch
    ev = torch.cuda.Event()
    ev.record(stream)
    while not ev.query():
        time.sleep(HOST_WAIT_SLEEP_S)


def int_array(values):
    return (C.c_int32 * len(values))(*[int(v) for v in values])


def prof_union(name: str) -> float:
    """Milliseconds during which at least one profiled launch called `name` was running."""
    ms = C.c_double()
    check(lib().fy_prof_union(name.encode(), C.byref(ms)))
    return ms.value


def prof_get(name: str):
    """(total_ms, work, launches) of the profiled launches called `name`."""
    ms, work, n = C.c_double(), C.c_double(), C.c_int64()
    check(lib().fy_prof_get(name.encode(), C.byref(ms), C.byref(work), C.byref(n)))
    return ms.value, work.value, n.value
