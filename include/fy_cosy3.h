/* libfy_cosy3 - MI355X (gfx950) native CosyVoice3-0.5B inference path.
 *
 * Plain C ABI: opaque handles, caller-owned buffers, int status (0 ok, <0 error
 * with fy_last_error()), no exceptions across the boundary.  Every call takes
 * the HIP stream to run on (void* = hipStream_t, NULL = default stream).  A
 * handle is single-threaded; different handles may run concurrently.  All
 * pointers are DEVICE pointers unless a parameter says "host".
 *
 * The reference has no FFI of its own (it is Python); each entry point names
 * the Python interface of the reference it replaces:
 *   stage objects swapped inside CosyVoice3Model  - cosyvoice/cli/model.py:101-129, 416-441
 *   the TensorRT estimator hand-off (raw pointers)  - cosyvoice/flow/flow_matching.py:126-153
 *   the vLLM LM hand-off (prompt_embeds -> ids)    - cosyvoice/llm/llm.py:482-510
 * INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 */
#ifndef FY_COSY3_H
#define FY_COSY3_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* a named weight: fp32, contiguous, on the device, reference state_dict name and shape */
typedef struct fy_tensor {
    const char* name;
    const void* data;
    int32_t ndim;
    int64_t shape[4];
} fy_tensor;

const char* fy_last_error(void);
int fy_version(void);
/* How the library's host threads wait for a stream in its hot paths (fy_llm_step's look at the stop flags every 8 token steps):
 * mode 0 = hipStreamSynchronize (spins on a core under HIP's default schedule), mode 1 = hipStreamQuery polled with `sleep_us`
 * microseconds of sleep between polls - a waiting thread costs next to nothing, a wait ends at most sleep_us late.  A pipelined
 * rank has 3-4 threads that wait most of the time (the host-side cost of one process per GPU); the Python mirror
 * switches to mode 1 (200 us; FY_HOST_WAIT_SPIN=1 keeps mode 0) and waits for its own streams the same way.  Process-wide.        */
int fy_set_host_wait(int32_t mode, int32_t sleep_us);

/* ---- opt-in launch profiler: HIP events around the named kernels' launches, on their stream.
 * names: "gemm_bf16" (DiT / projection GEMMs; work = flops), "conv_mfma" (work = flops),
 * "gemv" (LLM decode products; work = weight bytes).  Adds two event records per launch while on. */
void fy_prof_enable(int on);
void fy_prof_only(const char* name);   /* record only launches of this name (null: all) - keeps the cost off the other streams */
void fy_prof_reset(void);
int fy_prof_get(const char* name, double* total_ms, double* work, int64_t* count);
int fy_prof_union(const char* name, double* union_ms);   /* time during which at least one recorded launch of this name was running (launches on several streams overlap) */

/* Which of `n` HIP streams can run side by side: ratio (host fp32, n x n) receives, per pair, the time two concurrent
 * chains of short dependent kernels take over the time of one chain - ~1 when the two streams are served by different
 * hardware pipes, ~2 when they share a hardware queue or a pipe (they then take turns).  Used by the host side to place
 * the concurrent LM / flow streams of the pipelined path (cli/model.py runs the LM in its own thread beside
 * token2wav, cli/model.py:101-129, 339-369); ~3 ms per pair.                                                           */
int fy_stream_overlap(void* const* streams, int32_t n, float* ratio);
/* A HIP stream restricted to the compute units whose bits are set in mask (word i / 32, bit i % 32), made by the runtime this
 * library is bound to; fy_stream_destroy releases it.  Used to keep the flow / vocoder stream off a few CUs so a single LM
 * stream beside it is not starved (cli/model.py's LM thread beside token2wav).                                            */
int fy_stream_create_masked(void** out, const uint32_t* mask, int32_t n_words);
int fy_stream_destroy(void* stream);

/* ---- flags ------------------------------------------------------------------ */
#define FY_PRECISE 1u /* split-bf16 (hi+lo) activations on the MFMA paths: fp32-class accuracy, 2x MFMA work */
#define FY_DIRECT 2u  /* HiFT / DiT position conv: run convolutions on the exact fp32 VALU kernel */
#define FY_STREAMING 4u /* flow: block-causal chunk attention mask (streaming=True in the reference) */
#define FY_INCREMENTAL 16u /* flow, with FY_STREAMING | FY_NO_FINALIZE and B = 1: this call extends the previous FY_INCREMENTAL call of the
                             handle (same prompt, the earlier tokens a prefix): only the new rows go through the DiT blocks, against the
                             keys / values of the earlier rows kept per (Euler step, block).  Exact under the chunk mask when both calls end
                             on a chunk boundary (the reference's schedule does).  After fy_flow_stream_reset, when the prompt lengths differ
                             from the previous call's or the token count shrank, everything is computed and kept for the next call; a call
                             whose length is not a whole number of chunks is computed whole and keeps NOTHING (the next call starts over).
                             Same results as without the flag.                                                                         */
#define FY_NO_FINALIZE 8u /* flow, HiFT: a streaming chunk (finalize=False in the reference): the last tokens / frames are
                            look-ahead context, not output (flow: pre_lookahead tokens; HiFT: 3 + 4 frames and 480 samples) */

/* ================================ HiFT vocoder ================================
 * replaces CausalHiFTGenerator.inference(speech_feat, finalize=True)
 *   cosyvoice/hifigan/generator.py:713-726 (called from cli/model.py:438)        */
typedef struct fy_hift fy_hift;

typedef struct fy_hift_config {          /* cosyvoice3.yaml:77-100; zero-initialise then fy_hift_default_config */
    int32_t mel, base, harmonics, sampling_rate;
    float nsf_alpha, nsf_sigma, voiced_thr;
    int32_t ups[3], up_k[3];
    int32_t n_fft, hop;
    int32_t rb_k[3], rb_d[3], src_rb_k[3];
    float lrelu, audio_limit;
    int32_t pre_look_right, f0_ch;
} fy_hift_config;

void fy_hift_default_config(fy_hift_config* cfg);

/* weights: the tensors of hift.pt under their state_dict names (weight-norm pairs
 * `...parametrizations.weight.original0/1` are folded here).  Borrowed only during the call. */
int fy_hift_create(fy_hift** out, const fy_hift_config* cfg, const fy_tensor* weights, int32_t n_weights,
                   int32_t max_batch, int32_t max_frames, void* stream);
void fy_hift_destroy(fy_hift* h);

/* mel: (B, mel, Fmax) fp32, the reference's layout; frames: host int32[B] valid frames per utterance;
 * rand_ini: (harmonics+1); sine_noise: (>= 480*Fmax, harmonics+1) - the reference's fixed
 * SineGen2.rand_ini / sine_waves buffers (generator.py:223-226), explicit here;
 * wav: (B, 480*Fmax); source (nullable): (B, 480*Fmax).                                     */
int fy_hift_infer(fy_hift* h, const float* mel, const int32_t* frames, int32_t B, int32_t Fmax,
                  const float* rand_ini, const float* sine_noise, float* wav, float* source,
                  uint32_t flags, void* stream);

/* stage entries (parity tests, microbenchmarks) */
int fy_hift_f0(fy_hift* h, const float* mel, const int32_t* frames, int32_t B, int32_t Fmax, float* f0 /* (B,Fmax) */, void* stream);
int fy_hift_source(fy_hift* h, const float* f0, const int32_t* frames, int32_t B, int32_t Fmax, const float* rand_ini,
                   const float* sine_noise, float* source /* (B,480*Fmax) */, void* stream);
int fy_hift_decode(fy_hift* h, const float* mel, const float* source, const int32_t* frames, int32_t B, int32_t Fmax,
                   float* wav, uint32_t flags, void* stream);
/* copy an internal channels-last tensor of the last call: names f0, source, s_stft, conv_pre, fuse0..2,
 * stage0..2, conv_post.  dst (device) receives rows*cols floats per utterance, B*rows*cols in all.       */
int fy_hift_tap(fy_hift* h, const char* name, float* dst, int64_t* rows, int64_t* cols, void* stream);
/* one ResBlock (generator.py:110-117) of the main stack, index 0..8, on x (B, L, C) channels-last, in -> out.
 * The microbenchmark entry for BASELINE config 5.                                                           */
int fy_hift_resblock(fy_hift* h, int32_t index, const float* x, float* y, int32_t B, int32_t L, uint32_t flags, void* stream);

/* ============================ flow-matching decoder ============================
 * replaces CausalMaskedDiffWithDiT.inference(token, ..., prompt_feat, embedding, streaming, finalize=True)
 *   cosyvoice/flow/flow.py:358-403 (called from cli/model.py:418-427)                                   */
typedef struct fy_flow fy_flow;

typedef struct fy_flow_config {          /* cosyvoice3.yaml:38-75 */
    int32_t mel, spk_in, vocab, pre_ch, pre_lookahead;
    int32_t dim, depth, heads, head_dim, ff_mult, conv_pos_k, conv_pos_groups;
    int32_t n_timesteps;                 /* 10, flow.py:398 */
    float cfg_rate;                      /* inference_cfg_rate 0.7 */
    int32_t static_chunk;                /* chunk_size * token_mel_ratio = 50 */
    float t_span[33];                    /* 1 - cos(linspace(0,1,n+1)*pi/2), flow_matching.py:223-225 */
} fy_flow_config;

void fy_flow_default_config(fy_flow_config* cfg);

/* weights: the tensors of flow.pt under their state_dict names. max_frames bounds prompt + generated mel frames. */
int fy_flow_create(fy_flow** out, const fy_flow_config* cfg, const fy_tensor* weights, int32_t n_weights,
                   int32_t max_batch, int32_t max_frames, void* stream);
void fy_flow_destroy(fy_flow* f);
/* 1: every estimator matrix of flow.pt was bf16-representable (one bf16 plane each, nothing lost).  2: a general fp32 checkpoint -
 * the handle also keeps the lo planes bf16(w - bf16(w)); FY_PRECISE multiplies by w = hi + lo (and runs the position convolutions
 * on the exact fp32 kernel), so it meets the reference's estimator-swap bar (export_onnx.py:109) on such weights too; the default
 * bf16 mode uses the hi plane only.  The step-dependent modulation vectors computed at create use both planes in either mode.     */
int fy_flow_weight_planes(const fy_flow* f);

/* token (B, tok_ld) int32, prompt_token (B, ptok_ld) int32, prompt_feat (B, pfeat_rows, 80) fp32,
 * embedding (B, 192) fp32 - device; n_token / n_prompt / n_pfeat - host int32[B];
 * rand_noise (80, noise_ld): the reference's fixed CausalConditionalCFM.rand_noise (flow_matching.py:199-200);
 * mel out (B, 80, mel_frames), utterance b fills frames [0, 2*n_token[b]).                                     */
int fy_flow_infer(fy_flow* f, const int32_t* token, int32_t tok_ld, const int32_t* n_token, const int32_t* prompt_token,
                  int32_t ptok_ld, const int32_t* n_prompt, const float* prompt_feat, int32_t pfeat_rows, const int32_t* n_pfeat,
                  const float* embedding, const float* rand_noise, int32_t noise_ld, int32_t B, float* mel, int32_t mel_frames,
                  uint32_t flags, void* stream);

/* forget what FY_INCREMENTAL calls have kept: the next one starts a new stream (call it when a stream=True generation begins) */
int fy_flow_stream_reset(fy_flow* f);
/* mel rows (prompt + generated frames) the incremental calls of the current stream have kept; 0 = nothing (the next call computes all) */
int fy_flow_stream_rows(const fy_flow* f);

/* replaces the estimator hand-off ConditionalCFM.forward_estimator uses for a non-nn.Module estimator
 *   cosyvoice/flow/flow_matching.py:126-153: contiguous x (B2,80,T), mask (B2,1,T), mu (B2,80,T), t (B2),
 *   spks (B2,80), cond (B2,80,T); the result overwrites x.                                                   */
int fy_dit_estimator(fy_flow* f, float* x, const float* mask, const float* mu, const float* t, const float* spks,
                     const float* cond, int32_t T, int32_t B2, uint32_t flags, void* stream);

/* speed != 1 (cli/model.py:435-437): F.interpolate(mel, size=int(F / speed), mode="linear") between the flow decoder and
 * the vocoder - torch's align_corners=False rule: out[i] blends the two frames around max(F_in / F_out * (i + 0.5) - 0.5, 0).
 * mel (rows, F_in) -> out (rows, F_out), fp32 device, rows = B * 80.                                                       */
int fy_mel_speed(const float* mel, int32_t rows, int32_t F_in, float* out, int32_t F_out, void* stream);

/* ============================ speech-token language model ============================
 * replaces CosyVoice3LM.inference(...) + CosyVoiceModel.llm_job's token filter
 *   cosyvoice/llm/llm.py:713-748, 511-525; cosyvoice/cli/model.py:101-129                   */
typedef struct fy_llm fy_llm;

typedef struct fy_llm_config {           /* Qwen2-0.5B body + CosyVoice3LM heads, llm/llm.py:641-668 */
    int32_t hidden, layers, q_heads, kv_heads, head_dim, inter, vocab, speech_tokens;
    float rms_eps, rope_theta;
    /* How the matrices are stored (the reference keeps llm.pt in fp32, cli/cosyvoice.py:193 fp16=False, cli/model.py:65-73):
     * 1 = one bf16 plane - exact only for bf16-representable weights; 2 = two bf16 planes w = hi + lo (values kept to 2^-17
     * relative, text embedding in fp32: token ids track the fp32 reference on a general checkpoint; twice the weight bytes per
     * token step, no 8-row persistent step); 0 = choose: 2 exactly when some matrix element is not bf16-representable.      */
    int32_t weight_planes;
} fy_llm_config;

void fy_llm_default_config(fy_llm_config* cfg);

/* weights: the tensors of llm.pt under their state_dict names (lm_head is not needed). max_ctx bounds
 * prefill + generated positions per sequence.                                                           */
int fy_llm_create(fy_llm** out, const fy_llm_config* cfg, const fy_tensor* weights, int32_t n_weights,
                  int32_t max_batch, int32_t max_ctx, void* stream);
void fy_llm_destroy(fy_llm* l);

/* Greedy generation for B sequences.  text ids = concat[prompt_text, text] per sequence (host int32,
 * packed back to back, n_text_all[b] each); prompt_speech (host int32, packed, n_prompt_speech[b] each);
 * min_len / max_len host int32[B] (llm.py:743-744: 2x / 20x the text-only length).
 * out_ids (B, out_ld) device int32 receives the emitted speech tokens after the silent-token filter
 * (cli/model.py:121-128); out_n (B) device int32 their count; raw_n (nullable, B) the count before the filter.
 * Greedy rule: argmax of log_softmax; while fewer than min_len tokens were emitted the
 * argmax runs over the real speech tokens only; an id >= speech_tokens ends the sequence.                   */
int fy_llm_generate(fy_llm* l, const int32_t* text_ids, const int32_t* n_text_all, const int32_t* prompt_speech,
                    const int32_t* n_prompt_speech, const int32_t* min_len, const int32_t* max_len, int32_t B,
                    int32_t* out_ids, int32_t out_ld, int32_t* out_n, int32_t* raw_n, uint32_t flags, void* stream);
/* The same generation in pieces, for stream=True (cli/model.py:339-369 polls the token list its LM thread fills;
 * CosyVoice3LM.inference is a generator, llm.py:511-525): fy_llm_begin = prefill + the first token; fy_llm_step = up to
 * n_steps further tokens of the sequences that have not ended (same out_ids / out_ld as in fy_llm_begin; out_n / raw_n as
 * above, counts so far); finished (nullable, host int32[B]) = 1 once a sequence met a stop token or its max_len.
 * fy_llm_generate == fy_llm_begin + fy_llm_step(max over max_len): the ids are identical however the steps are cut.        */
int fy_llm_begin(fy_llm* l, const int32_t* text_ids, const int32_t* n_text_all, const int32_t* prompt_speech,
                 const int32_t* n_prompt_speech, const int32_t* min_len, const int32_t* max_len, int32_t B,
                 int32_t* out_ids, int32_t out_ld, void* stream);
/* The embeddings-level entry - the shape of the reference's engine-level LM swap (its vLLM path, cosyvoice/llm/llm.py:482-510:
 * `add_request(uuid, {"prompt_embeds": lm_input (L, 896) bf16}, SamplingParams(stop_token_ids, min_tokens, max_tokens))`, then
 * `step()` per token; weights exported with embed_tokens := speech_embedding, cosyvoice/utils/file_utils.py:92-118).  The HOST has
 * assembled lm_input = [sos, embed_tokens(text), task_id, speech_embedding(prompt)] itself (llm.py:728-740); this call is
 * fy_llm_begin without the assembly: prefill of the given rows + the first token, and fy_llm_step continues it (every next input is
 * speech_embedding[id], gathered on the device).  embeds: device, the B sequences' rows packed back to back, (sum n_rows, hidden),
 * dtype 0 = fp32, 1 = bf16; n_rows / min_len / max_len: host int32[B]; sequence b occupies slot b of the handle until the next
 * fy_llm_begin / fy_llm_prefill.  Same ids as fy_llm_begin for the same rows (tests/test_llm_gpu.py).                          */
int fy_llm_prefill(fy_llm* l, const void* embeds, int32_t dtype, const int32_t* n_rows, const int32_t* min_len, const int32_t* max_len,
                   int32_t B, int32_t* out_ids, int32_t out_ld, void* stream);
int fy_llm_step(fy_llm* l, int32_t n_steps, int32_t* out_ids, int32_t out_ld, int32_t* out_n, int32_t* raw_n,
                int32_t* finished, void* stream);
/* Sampler used by the following fy_llm_generate calls.  kind 0: the greedy rule above.  kind 1: the reference's default,
 * repetition-aware sampling (utils/common.py:137-166 `ras_sampling` with top_p, top_k, win_size, tau_r = 0.8, 25, 10, 0.1 in
 * cosyvoice3.yaml) inside the ignore_eos retry loop of `sampling_ids` (llm/llm.py:149-164; > 100 retries -> FY_ERR_STATE with
 * the reference's message).  torch.multinomial's draws are replaced by the inverse CDF at caller-supplied uniforms:
 * `uniforms` is device fp32 (max_batch, n_uniforms), borrowed until the next call; sequence b consumes its row from the
 * start of every fy_llm_generate call, one value per multinomial the reference would draw.                               */
int fy_llm_set_sampler(fy_llm* l, int32_t kind, const float* uniforms, int64_t n_uniforms, int32_t top_k, float top_p,
                       int32_t win_size, float tau_r);
/* How fy_llm_step runs a token step (Qwen2Encoder.forward_one_step + llm_decoder, llm/llm.py:246-258, 518).
 * mode 1 (default when the architecture fits and the batch is <= 8): ONE persistent launch per token step - lowest latency,
 * but its workgroups hold 152 CUs while they wait on each other, so concurrent streams (a flow decoder beside it, other LM
 * handles) get the rest of the chip only; two such launches never overlap.  mode 0: one launch per operation (121 per step)
 * - slower alone, but its short kernels interleave with other streams' work.  Token ids are identical in both modes.
 * mode 2: only the FEW-CU persistent step (one launch per token step on 76 workgroups, for any batch <= 32 - what a pipeline that
 * runs flow decoders beside the LM wants: a call of <= 8 sequences then never takes the 152-workgroup step, whose grid would wait
 * for residency beside them); larger batches or handles without that step run one launch per operation.
 * fy_llm_decode_mode returns the mode in effect (0 when the persistent kernel is not available for this handle).        */
int fy_llm_set_decode_mode(fy_llm* l, int32_t mode);
int fy_llm_decode_mode(const fy_llm* l);
/* 1 or 2: how this handle stores its matrices (fy_llm_config.weight_planes resolved at create).                      */
int fy_llm_weight_planes(const fy_llm* l);
/* log_softmax of step `step` (0 = first generated token) of the last fy_llm_generate call, (B, speech_tokens+200).
 * Only the first FY_LLM_KEEP_LOGP steps are kept.                                                            */
#define FY_LLM_KEEP_LOGP 4
int fy_llm_logp(fy_llm* l, int32_t step, float* dst, void* stream);

/* ================================ frontend: prompt mel ================================
 * replaces matcha.utils.audio.mel_spectrogram as cosyvoice3.yaml:140-148 configures it (n_fft = win = 1920, hop 480, 80 mels,
 * fmin 0, fmax sr/2, center=False), third_party/Matcha-TTS/matcha/utils/audio.py:45-82, called by
 * CosyVoiceFrontEnd._extract_speech_feat on the 24 kHz prompt (cli/frontend.py:119-125).
 * wav: device fp32 (n_samples) in [-1, 1]; mel: device fp32 (frames, 80), frames = fy_prompt_mel_frames(n_samples)
 * (= n_samples / 480 when that divides).  The mel filterbank is librosa's Slaney form restated (librosa is not in the
 * image: that table is unpinned; the STFT is pinned against torch.stft).                                                   */
typedef struct fy_prompt_mel fy_prompt_mel;
int fy_prompt_mel_create(fy_prompt_mel** out, int32_t sample_rate, void* stream);
void fy_prompt_mel_destroy(fy_prompt_mel* p);
int fy_prompt_mel_frames(int32_t n_samples);
int fy_prompt_mel_run(fy_prompt_mel* p, const float* wav, int32_t n_samples, float* mel, int32_t frames, void* stream);

/* ================================ frontend: the 16 kHz features of the two ONNX models ================================
 * kind 0 replaces whisper.log_mel_spectrogram(speech, n_mels=128) of CosyVoiceFrontEnd._extract_speech_token
 *        (cli/frontend.py:94-108; the input of speech_tokenizer_v3.onnx): out (128, frames), frames = n_samples / 160.
 * kind 1 replaces torchaudio.compliance.kaldi.fbank(speech, num_mel_bins=80, dither=0, sample_frequency=16000) of
 *        _extract_spk_embedding (cli/frontend.py:110-117; the input of campplus.onnx): out (frames, 80),
 *        frames = 1 + (n_samples - 400) / 160; flags bit 0 also subtracts the mean over the frames (frontend.py:115).
 * wav: device fp32 (n_samples) at 16 kHz in [-1, 1].  whisper and torchaudio are not in the image: both algorithms are
 * restated from their published definitions (parity unpinned; the transforms are held to torch.stft / torch.fft by the tests). */
typedef struct fy_audio_feat fy_audio_feat;
int fy_audio_feat_create(fy_audio_feat** out, int32_t kind, void* stream);
void fy_audio_feat_destroy(fy_audio_feat* p);
int fy_audio_feat_mels(const fy_audio_feat* p);
int fy_audio_feat_frames(const fy_audio_feat* p, int64_t n_samples);
int fy_audio_feat_run(fy_audio_feat* p, const float* wav, int64_t n_samples, float* out, int32_t frames, uint32_t flags, void* stream);

/* ================================ multi-GPU: the all-gather of the finished audio ================================
 * north_star's one exchange: every rank's b <= b_max finished utterances (wav (b, wav_ld) fp32 on the device, valid
 * lengths n_samples (b) int32 on the device) to every rank in ONE fixed-size ncclAllGather on `stream`:
 * wav_all (world * b_max, s_max) in rank order, zero-padded; n_all (world, b_max + 1) int32 = count, lengths of each rank.
 * rccl_comm: the caller's ncclComm_t.  The library does not link RCCL - it calls the copy the process already holds (loaded with
 * global symbol visibility); FY_ERR_STATE when there is none.  scratch: fy_allgather_audio_scratch_floats(world, b_max, s_max)
 * floats on the device.  Python hosts use fangyan_tts_amd.parallel.gather_audio over torch.distributed instead.              */
size_t fy_allgather_audio_scratch_floats(int32_t world, int32_t b_max, int32_t s_max);
int fy_allgather_audio(void* rccl_comm, int32_t world, const float* wav, int64_t wav_ld, const int32_t* n_samples, int32_t b,
                       int32_t b_max, int32_t s_max, float* scratch, float* wav_all, int32_t* n_all, void* stream);

/* The record alone, for a host that gathers with its own collective (fangyan_tts_amd/parallel.py over torch.distributed): rec
 * (b_max * s_max + b_max + 1 floats on the device) = the b rows of wav zero-padded / cut to (b_max, s_max), then count and lengths
 * (min(n, s_max)) bit-cast to float.  n_samples_host: HOST int32[b]; they ride in the kernel arguments (no copy, no stream wait). */
int fy_audio_record_pack(const float* wav, int64_t wav_ld, const int32_t* n_samples_host, int32_t b, int32_t b_max, int32_t s_max,
                         float* rec, void* stream);

/* ================================ tooling: synthetic tensors ================================
 * No pretrained checkpoint is reachable offline: tests and bench.py fill the model from a counter-based generator
 * (fangyan_tts_amd/synth.py).  This is that generator's uniform draw on the device, bit-identical to the numpy form:
 * dst[i] = float(lo + (hi - lo) * u(seed, start + i)); flags bit 0 rounds to bf16-representable values, bit 1 adds 1.0f.  */
int fy_synth_uniform(float* dst, int64_t n, uint64_t seed, int64_t start, double lo, double hi, uint32_t flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif
