// Persistent decode step of the speech-token LM for up to 32 sequences on a FEW compute units (llm_decode32.hip): the 24 Qwen2
// layers + llm_decoder of one token step (Qwen2Encoder.forward_one_step + llm_decoder, CosyVoice/cosyvoice/llm/llm.py:246-258, 518)
// in ONE launch of G = inter / (16 TG) workgroups - 38 for CosyVoice3-0.5B - where the per-operation path (gemv32.hip) needs 122
// launches whose blocks come and go all over the chip beside the flow decoder.  Same operands and layouts as that path (the weights
// in gemv_pack's fragment order, the activations as gv32 A images), so a generation may switch between the two at any step.
#pragma once
#include "common.h"

struct Dec32Layer {
    const bf16_t *wqkv, *wo, *wgu, *wd;       // gemv_pack fragment order [N/32][K/16][64][8]; wgu rows (gate_i, up_i) interleaved
    const float *bqkv, *ln1, *ln2;
    float *Kc, *Vc;                           // this layer's cache [seq][Hk][max_ctx][64]
    // exact-weights mode (gemv32.h: Gv32Args::W_lo): the lo planes bf16(w - bf16(w)) in the same order; all null = one plane
    const bf16_t *wqkv_lo = nullptr, *wo_lo = nullptr, *wgu_lo = nullptr, *wd_lo = nullptr;
};

struct Dec32Shape {
    int H = 0, I = 0, Hq = 0, Hk = 0, layers = 0, NS = 0, max_ctx = 0, mb = 0;
    float eps = 0.f;
    int qkv() const { return (Hq + 2 * Hk) * 64; }
};

struct Dec32Plan;

bool decode32_supported(const Dec32Shape& s);
// layers: host array of s.layers entries (device pointers); copied to the device
// w_head_lo: the head's lo plane (exact-weights mode; then every layer's *_lo pointers must be set too), or null
int decode32_create(Dec32Plan** out, const Dec32Shape& s, const Dec32Layer* layers, const bf16_t* w_head, const float* norm_w, hipStream_t st,
                    const bf16_t* w_head_lo = nullptr);
void decode32_destroy(Dec32Plan* p);
int decode32_groups(const Dec32Plan* p);
// One token step for sequences 0 .. B-1 (8 < B <= 32 is what it is for; any 1 <= B <= 32 works).  In: h (fp32 [mb][H], the rows the
// sampler wrote), img_h / ssq (the gv32 A image of ln1[0] x h and its per-tile sums of squares, written by the sampler or the
// prefill); st = the handle's state block (row 0 = positions).  Scratch of the per-operation path is shared: qkv (fp32 [32][QKV]),
// img_ao.  Out: logits (fp32 [B][NS]); h / img_h / ssq are left as the last layer's output.
int decode32_step(Dec32Plan* p, int B, float* h, bf16_t* img_h, float* ssq, float* qkv, bf16_t* img_ao, const int* st_block,
                  const float* inv_freq, float* logits, hipStream_t stream);
// *out != 0 (valid after the stream has been synchronised): a grid hand-off timed out - not every workgroup was resident
int decode32_status(Dec32Plan* p, unsigned* out, hipStream_t stream);
// diagnostic: per-phase 100 MHz stamps of workgroup 0 in the last launch (n > 0 arms and copies, 0 disarms)
int decode32_stamps(Dec32Plan* p, unsigned long long* out, int n, hipStream_t stream);
