// Persistent decode step of the speech-token LM: the 24 Qwen2 layers + llm_decoder of ONE token step for up to
// 8 sequences in a single launch (llm_decode.hip).  Used by fy_llm_step when the architecture fits (DecodePlan::supported).
#pragma once
#include "common.h"

struct DecodeShape {
    int H = 0, I = 0, Hq = 0, Hk = 0, layers = 0, NS = 0, max_ctx = 0, mb = 0;
    float eps = 0.f;
    int qkv() const { return (Hq + 2 * Hk) * 64; }
};

// fp32 sources of one layer (device pointers, torch Linear layout [N][K]); wqkv = rows q | k | v, wgu = rows (gate_i, up_i) interleaved
struct DecodeLayerSrc {
    const float *wqkv, *wo, *wgu, *wd, *bqkv, *ln1, *ln2;
    float *Kc, *Vc;                      // this layer's cache: [seq][Hk][max_ctx][64] fp32
};

struct DecodePlan;

// nullptr (no error set) when the shape is not one the kernel is built for or FY_LLM_PERSISTENT=0
bool decode_supported(const DecodeShape& s);
int decode_create(DecodePlan** out, const DecodeShape& s, hipStream_t st);
// pack layer `i` (call once per layer, any order), then the head
int decode_pack_layer(DecodePlan* p, int i, const DecodeLayerSrc& src, hipStream_t st);
int decode_pack_head(DecodePlan* p, const float* w_head /*[NS][H]*/, const float* norm_w, hipStream_t st);
void decode_destroy(DecodePlan* p);
size_t decode_bytes(const DecodePlan* p);
// one token step for sequences 0..B-1: h (fp32 [mb][H], the rows sample_k wrote) -> logits (fp32 [B][NS]); st = the
// handle's state block (row 0 = positions); the caches are appended at those positions.
int decode_step(DecodePlan* p, int B, float* h, const int* st, const float* inv_freq, float* logits, hipStream_t stream);
// *out != 0 (valid after the stream has been synchronised): a grid-wide hand-off of some launch on this plan timed out
// (not every workgroup became resident within a second); the step's results are void
int decode_status(DecodePlan* p, unsigned* out, hipStream_t stream);
// diagnostic phase time stamps of the last launch (see llm_decode.hip)
int decode_stamps(DecodePlan* p, unsigned long long* out, int n, hipStream_t stream);
