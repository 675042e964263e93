// Attention kernels.
//   dit_attention   : non-causal multi-head attention of the DiT estimator (16 heads x 64, bf16 in,
//                     fp32 online softmax, bf16 out) with a key-padding mask and the optional
//                     block-causal chunk mask of streaming inference
//                     (CosyVoice/cosyvoice/flow/DiT/modules.py:349-407; utils/mask.py:127-158).
//   llm_attention   : fp32 grouped-query causal attention over the KV cache, one wave per
//                     (query row, query head); serves prefill rows and decode steps alike
//                     (transformers Qwen2Attention as called from llm/llm.py:246-258).
#include "attn.h"

typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;

#define AT_D 64
#define AT_PITCH 72          // bf16 elements per LDS row (64 + 8 pad)

// qkv: bf16 [nseq*Tmax][3*H*64] as [q | k | v]; out: bf16 [nseq*Tmax][H*64]
__global__ __launch_bounds__(256) void dit_attention_k(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                       const int* __restrict__ seq_len, int Tmax, int H, int chunk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * AT_PITCH];
    __shared__ __attribute__((aligned(16))) bf16_t Vt[64 * AT_PITCH];          // [d][key]
    __shared__ __attribute__((aligned(16))) bf16_t Pw[4 * 16 * AT_PITCH];      // per wave [q][key]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int s = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64;
    const int len = seq_len[s];
    if (q0 >= len) return;
    const int ld = 3 * H * AT_D;
    const bf16_t* base = qkv + (long)s * Tmax * ld;
    const int g = lane >> 4, lc = lane & 15;
    const int qrow = q0 + wid * 16 + lc;                     // the row whose Q fragment this lane holds
    frag_ab qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        if (qrow < len) qf[ks] = *reinterpret_cast<const frag_ab*>(base + (long)qrow * ld + h * AT_D + ks * 32 + g * 8);
        else
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[ks][j] = (__bf16)0.f;
    }
    f32x4 o[4];
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) o[nd] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrow[4], lrow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { mrow[r] = -1e30f; lrow[r] = 0.f; }
    // keys this block may need: everything (non-causal) or up to the end of the last query's chunk
    int kend = len;
    if (chunk > 0) kend = min(len, ((min(q0 + 63, len - 1) / chunk) + 1) * chunk);
    bf16_t* pw = Pw + wid * 16 * AT_PITCH;
    // K/V tiles are fetched one tile ahead into registers: the loads of tile t+1 fly under the MFMAs of tile t
    uint4 kreg[2], vreg[2];
    auto load_kv = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int idx = tid + i * 256, key = idx >> 3, dc = (idx & 7) * 8;
            kreg[i] = make_uint4(0, 0, 0, 0);
            vreg[i] = kreg[i];
            if (k0 + key < len) {
                const bf16_t* p = base + (long)(k0 + key) * ld + h * AT_D + dc;
                kreg[i] = *reinterpret_cast<const uint4*>(p + H * AT_D);
                vreg[i] = *reinterpret_cast<const uint4*>(p + 2 * H * AT_D);
            }
        }
    };
    load_kv(0);
    for (int k0 = 0; k0 < kend; k0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int idx = tid + i * 256, key = idx >> 3, dc = (idx & 7) * 8;
            *reinterpret_cast<uint4*>(Ks + key * AT_PITCH + dc) = kreg[i];
            const bf16_t* ve = reinterpret_cast<const bf16_t*>(&vreg[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) Vt[(dc + j) * AT_PITCH + key] = ve[j];
        }
        __syncthreads();
        if (k0 + 64 < kend) load_kv(k0 + 64);
        // S = Q K^T for this wave's 16 rows x 64 keys: sc[nb][r] = S[row 4g+r][key 16nb+lc]
        f32x4 sc[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            sc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                frag_ab kf = *reinterpret_cast<const frag_ab*>(Ks + (nb * 16 + lc) * AT_PITCH + ks * 32 + g * 8);
                sc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], kf, sc[nb], 0, 0, 0);
            }
        }
        float tmax[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qr = q0 + wid * 16 + 4 * g + r;
            const int lim = chunk > 0 ? min(len, ((qr / chunk) + 1) * chunk) : len;
            float mx = -1e30f;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int key = k0 + nb * 16 + lc;
                float v = key < lim ? sc[nb][r] * scale : -1e30f;
                sc[nb][r] = v;
                mx = fmaxf(mx, v);
            }
#pragma unroll
            for (int of = 1; of < 16; of <<= 1) mx = fmaxf(mx, __shfl_xor(mx, of, 64));
            tmax[r] = mx;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float mnew = fmaxf(mrow[r], tmax[r]);
            const float alpha = __expf(mrow[r] - mnew);
            float ps = 0.f;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                float p = sc[nb][r] > -1e29f ? __expf(sc[nb][r] - mnew) : 0.f;
                ps += p;
                pw[(4 * g + r) * AT_PITCH + nb * 16 + lc] = f32_to_bf16(p);
            }
#pragma unroll
            for (int of = 1; of < 16; of <<= 1) ps += __shfl_xor(ps, of, 64);
            lrow[r] = lrow[r] * alpha + ps;
            mrow[r] = mnew;
#pragma unroll
            for (int nd = 0; nd < 4; ++nd) o[nd][r] *= alpha;
        }
        // the wave reads back only its own P rows; LDS ops of one wave complete in order
        __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag_ab pf = *reinterpret_cast<const frag_ab*>(pw + lc * AT_PITCH + ks * 32 + g * 8);
#pragma unroll
            for (int nd = 0; nd < 4; ++nd) {
                frag_ab vf = *reinterpret_cast<const frag_ab*>(Vt + (nd * 16 + lc) * AT_PITCH + ks * 32 + g * 8);
                o[nd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf, o[nd], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qr = q0 + wid * 16 + 4 * g + r;
        if (qr >= len) continue;
        const float inv = 1.f / lrow[r];
        bf16_t* op = out + ((long)s * Tmax + qr) * (H * AT_D) + h * AT_D;
#pragma unroll
        for (int nd = 0; nd < 4; ++nd) op[nd * 16 + lc] = f32_to_bf16(o[nd][r] * inv);
    }
}

int dit_attention(const bf16_t* qkv, bf16_t* out, const int* seq_len, int nseq, int Tmax, int H, int chunk, hipStream_t st) {
    FY_CHECK(qkv && out && seq_len && nseq >= 1 && Tmax >= 1 && H >= 1, FY_ERR_ARG, "dit_attention: bad arguments");
    dim3 grid(cdiv(Tmax, 64), H, nseq);
    hipLaunchKernelGGL(dit_attention_k, grid, dim3(256), 0, st, qkv, out, seq_len, Tmax, H, chunk, 0.125f);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// -----------------------------------------------------------------------------------------------
// q: fp32 [R][q_ld] (head hq at column hq*64); cache K/V: fp32 [seq][kvh][max_ctx][64];
// row r attends positions 0..row_pos[r] of sequence row_seq[r]; out: fp32 [R][o_ld]
__global__ __launch_bounds__(64) void llm_attention_k(const float* __restrict__ q, int q_ld, const float* __restrict__ Kc,
                                                      const float* __restrict__ Vc, const int* __restrict__ row_seq,
                                                      const int* __restrict__ row_pos, float* __restrict__ out, int o_ld,
                                                      int Hq, int Hk, int max_ctx, float scale) {
    extern __shared__ float sh[];             // [64] q + [max_ctx] scores
    float* qs = sh;
    float* sc = sh + 64;
    const int r = blockIdx.x, hq = blockIdx.y, lane = threadIdx.x;
    const int seq = row_seq[r], n = row_pos[r] + 1;
    const int hk = hq / (Hq / Hk);
    qs[lane] = q[(long)r * q_ld + hq * 64 + lane];
    __syncthreads();
    const float* Kb = Kc + ((long)seq * Hk + hk) * max_ctx * 64;
    const float* Vb = Vc + ((long)seq * Hk + hk) * max_ctx * 64;
    float mx = -1e30f;
    for (int j = lane; j < n; j += 64) {
        const float4* kr = reinterpret_cast<const float4*>(Kb + (long)j * 64);
        float s = 0.f;
#pragma unroll
        for (int d4 = 0; d4 < 16; ++d4) {
            float4 kv = kr[d4];
            s = fmaf(qs[d4 * 4 + 0], kv.x, s);
            s = fmaf(qs[d4 * 4 + 1], kv.y, s);
            s = fmaf(qs[d4 * 4 + 2], kv.z, s);
            s = fmaf(qs[d4 * 4 + 3], kv.w, s);
        }
        s *= scale;
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) {
        float p = expf(sc[j] - mx);
        sc[j] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    __syncthreads();
    float acc = 0.f;
    for (int j = 0; j < n; ++j) acc = fmaf(sc[j], Vb[(long)j * 64 + lane], acc);
    out[(long)r * o_ld + hq * 64 + lane] = acc / sum;
}

int llm_attention(const float* q, int q_ld, const float* Kc, const float* Vc, const int* row_seq, const int* row_pos, float* out,
                  int o_ld, int R, int Hq, int Hk, int max_ctx, hipStream_t st) {
    FY_CHECK(q && Kc && Vc && row_seq && row_pos && out && R >= 1 && Hq % Hk == 0, FY_ERR_ARG, "llm_attention: bad arguments");
    size_t lds = (64 + (size_t)max_ctx) * sizeof(float);
    FY_CHECK(lds <= 64 * 1024, FY_ERR_ARG, "llm_attention: context %d too long for the score buffer", max_ctx);
    hipLaunchKernelGGL(llm_attention_k, dim3(R, Hq), dim3(64), lds, st, q, q_ld, Kc, Vc, row_seq, row_pos, out, o_ld, Hq, Hk, max_ctx, 0.125f);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// -----------------------------------------------------------------------------------------------
// Decode step: split-half RoPE of this row's q and new k, the cache append, and attention over
// positions 0..pos in one launch.  qkv: fp32 [R][(Hq+2Hk)*64] raw projections of the current token;
// the new key/value never round-trip through the cache inside the step (each block rotates its own
// copy; the first query head of a kv group writes it for later steps).
__global__ __launch_bounds__(256) void llm_attention_step_k(const float* __restrict__ qkv, float* __restrict__ Kc, float* __restrict__ Vc,
                                                            const int* __restrict__ row_seq, const int* __restrict__ row_pos,
                                                            const float* __restrict__ inv_freq, float* __restrict__ out, int o_ld,
                                                            int Hq, int Hk, int max_ctx, float scale) {
    // 4 waves per (row, query head): every wave rotates q / the new k itself (64 lanes = 64 dims); the cached keys
    // are spread one per thread, the value rows in four contiguous quarters, partial sums meet in LDS.
    extern __shared__ float sh[];             // [64] q + [max_ctx] scores + [4][64] partial outputs + [8] reductions
    float* qs = sh;
    float* sc = sh + 64;
    float* part = sc + max_ctx;
    float* redm = part + 256;
    const int r = blockIdx.x, hq = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int seq = row_seq[r], pos = row_pos[r];
    const int grp = Hq / Hk, hk = hq / grp;
    const int ld = (Hq + 2 * Hk) * 64;
    const float* row = qkv + (long)r * ld;
    const float qv = row[hq * 64 + lane], kv = row[(Hq + hk) * 64 + lane], vv = row[(Hq + Hk + hk) * 64 + lane];
    const float ang = (float)pos * inv_freq[lane & 31];
    float sn, cs;
    sincosf(ang, &sn, &cs);
    const float qo = __shfl_xor(qv, 32, 64), ko = __shfl_xor(kv, 32, 64);
    const float qr = lane < 32 ? qv * cs - qo * sn : qv * cs + qo * sn;
    const float kr = lane < 32 ? kv * cs - ko * sn : kv * cs + ko * sn;
    float* Kb = Kc + ((long)seq * Hk + hk) * max_ctx * 64;
    float* Vb = Vc + ((long)seq * Hk + hk) * max_ctx * 64;
    if (wid == 0) {
        if (hq % grp == 0) {
            Kb[(long)pos * 64 + lane] = kr;
            Vb[(long)pos * 64 + lane] = vv;
        }
        qs[lane] = qr;
    }
    __syncthreads();
    const float s_new = wave_sum(qr * kr) * scale;
    float mx = s_new;
    for (int j = tid; j < pos; j += 256) {
        const float4* kp = reinterpret_cast<const float4*>(Kb + (long)j * 64);
        float4 k4[16];
#pragma unroll
        for (int d4 = 0; d4 < 16; ++d4) k4[d4] = kp[d4];
        float s = 0.f;
#pragma unroll
        for (int d4 = 0; d4 < 16; ++d4) {
            s = fmaf(qs[d4 * 4 + 0], k4[d4].x, s);
            s = fmaf(qs[d4 * 4 + 1], k4[d4].y, s);
            s = fmaf(qs[d4 * 4 + 2], k4[d4].z, s);
            s = fmaf(qs[d4 * 4 + 3], k4[d4].w, s);
        }
        s *= scale;
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if (lane == 0) redm[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    float sum = 0.f;
    for (int j = tid; j < pos; j += 256) {
        float p = expf(sc[j] - mx);
        sc[j] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) redm[4 + wid] = sum;
    __syncthreads();
    const float p_new = expf(s_new - mx);
    sum = ((redm[4] + redm[5]) + (redm[6] + redm[7])) + p_new;
    // values: wave w owns positions [w*q4, (w+1)*q4)
    const int q4 = (pos + 3) >> 2, j0 = wid * q4, j1 = min(pos, j0 + q4);
    float acc = 0.f;
    int j = j0;
    for (; j + 8 <= j1; j += 8) {             // eight independent row loads in flight per trip
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = Vb[(long)(j + u) * 64 + lane];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(sc[j + u], v[u], acc);
    }
    for (; j < j1; ++j) acc = fmaf(sc[j], Vb[(long)j * 64 + lane], acc);
    part[wid * 64 + lane] = acc;
    __syncthreads();
    if (wid == 0) {
        float o = ((part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane])) + p_new * vv;
        out[(long)r * o_ld + hq * 64 + lane] = o / sum;
    }
}

int llm_attention_step(const float* qkv, float* Kc, float* Vc, const int* row_seq, const int* row_pos, const float* inv_freq,
                       float* out, int o_ld, int R, int Hq, int Hk, int max_ctx, hipStream_t st) {
    FY_CHECK(qkv && Kc && Vc && row_seq && row_pos && inv_freq && out && R >= 1 && Hq % Hk == 0, FY_ERR_ARG, "llm_attention_step: bad arguments");
    size_t lds = (64 + (size_t)max_ctx + 256 + 8) * sizeof(float);
    FY_CHECK(lds <= 64 * 1024, FY_ERR_ARG, "llm_attention_step: context %d too long for the score buffer", max_ctx);
    hipLaunchKernelGGL(llm_attention_step_k, dim3(R, Hq), dim3(256), lds, st, qkv, Kc, Vc, row_seq, row_pos, inv_freq, out, o_ld, Hq, Hk,
                       max_ctx, 0.125f);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}
