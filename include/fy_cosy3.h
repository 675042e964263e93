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

/* ---- flags ------------------------------------------------------------------ */
#define FY_PRECISE 1u /* split-bf16 (hi+lo) activations on the MFMA paths: fp32-class accuracy, 2x MFMA work */
#define FY_DIRECT 2u  /* HiFT: run every convolution on the exact fp32 VALU kernel */

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

#ifdef __cplusplus
}
#endif
#endif
