#pragma once
#include "common.h"

// qkv: bf16 [nseq*Tmax][3*H*64] laid out [q | k | v]; out: bf16 [nseq*Tmax][H*64]; seq_len: device int[nseq].
// chunk > 0 adds the streaming block-causal mask key < (query/chunk + 1)*chunk.  q_begin > 0: only the rows from q_begin on are
// computed and written (the keys / values of the rows before them are read as they stand: incremental streaming).  interleaved:
// row t of sequence s is row nseq * t + s of qkv / out instead of s * Tmax + t.
int dit_attention(const bf16_t* qkv, bf16_t* out, const int* seq_len, int nseq, int Tmax, int H, int chunk, hipStream_t st, int q_begin = 0,
                  bool interleaved = false);

// the default kernel behind dit_attention (attn_dit.hip): row t of sequence s at row s * seq_rows + t * row_step
int dit_attention2(const bf16_t* qkv, bf16_t* out, const int* seq_len, int nseq, int Tmax, int H, int chunk, hipStream_t st, int q_begin, int seq_rows, int row_step);

// The split-operand form (fp32-class flow decoder): q, k, v and the output as x = hi + lo, two bf16 planes each, same layouts;
// three MFMAs per product keep the hi x hi, hi x lo and lo x hi terms, softmax in fp32.
int dit_attention_split(const bf16_t* qkv_hi, const bf16_t* qkv_lo, bf16_t* out_hi, bf16_t* out_lo, const int* seq_len, int nseq, int Tmax, int H, int chunk,
                        hipStream_t st);

// fp32 GQA attention over the KV cache, head_dim 64.  q: [R][q_ld]; K/V cache: [seq][Hk][max_ctx][64];
// row r attends positions 0..row_pos[r] of sequence row_seq[r].
int llm_attention(const float* q, int q_ld, const float* Kc, const float* Vc, const int* row_seq, const int* row_pos, float* out,
                  int o_ld, int R, int Hq, int Hk, int max_ctx, hipStream_t st);

// decode step: RoPE of q / new k, cache append and attention over 0..row_pos[r] in one launch (qkv raw, fp32)
int llm_attention_step(const float* qkv, float* Kc, float* Vc, const int* row_seq, const int* row_pos, const float* inv_freq,
                       float* out, int o_ld, int R, int Hq, int Hk, int max_ctx, hipStream_t st, bf16_t* img = nullptr);
