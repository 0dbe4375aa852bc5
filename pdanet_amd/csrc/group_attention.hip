// group_attention.hip -- self-attention over the nsample tokens of one ball-query group.
//
// The PDA layer runs TransformerEncoderLayerPreNorm over sequences of length nsample (16 or 32)
// with a batch of B*npoint groups and 4 heads
// (/root/reference/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py:924-929,
// PointFormer.py:30-33: nn.MultiheadAttention, dropout 0, no mask).  Generic flash-attention
// kernels are built for long sequences; at S <= 32 they spend their time on tiling overhead
// (measured: 6.3 ms of a 58 ms training step, backward 2.3 ms for a 1 GB problem).
//
// Here ONE wave handles one (group, head) pair -- 32/S pairs when S < 32 -- entirely in registers:
//   * scores T = K Q^T on v_mfma_f32_32x32x2_f32 (fp32 in / fp32 accumulate, like the reference).
//     The orientation is chosen so that the accumulator holds T[j][i] with the QUERY i on the lane
//     and the keys j in the 16 registers x 2 lane halves: the softmax over keys is then a
//     per-lane reduction plus one cross-half exchange (no LDS, no cross-lane trees);
//   * the k index of an MFMA is free to permute as long as A and B agree, so lane half h simply
//     takes the contiguous half [h*hd/2, (h+1)*hd/2) of the head dimension: operands are loaded
//     with 16-byte loads straight from the (B*np, S, 3, H, hd) in_proj output;
//   * P^T (resp. dS^T) in accumulator layout is directly the B operand of O^T = V^T P^T
//     (resp. dQ^T = K^T dS^T): register t of lane half h holds key (t&3)+8(t>>2)+4h, the A operand
//     is fetched in that key order;
//   * the products that contract over the QUERY index (dV, dK) need P / dS transposed: one
//     33-float-stride LDS round trip per wave (conflict-free).
// Forward writes O (B*np, S, H*hd) and the log-sum-exp per query; backward recomputes P.
#include "pda_common.h"

namespace pda {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int HD>
struct GaPtrs {
    // element (b, s, which, head, d) of qkv (Bn, S, 3, H, HD)
    __device__ __forceinline__ static size_t qkv(int S, int H, size_t b, int s, int which, int head) {
        return (((b * S + s) * 3 + which) * H + head) * (size_t)HD;
    }
    __device__ __forceinline__ static size_t o(int S, int H, size_t b, int s, int head) {
        return ((b * S + s) * H + head) * (size_t)HD;
    }
};

// contiguous half-row operand: lane (c, h) gets x[h*HD/2 + t], t = 0..HD/2-1 (zeros when !ok)
template <int HD>
__device__ __forceinline__ void load_half_row(float (&dst)[HD / 2], const float* __restrict__ p, bool ok, int h) {
#pragma unroll
    for (int q = 0; q < HD / 8; ++q) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = *reinterpret_cast<const float4*>(p + h * (HD / 2) + 4 * q);
        dst[4 * q + 0] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w;
    }
}

// S: tokens per group (8, 16 or 32).  A wave covers G = 32/S groups of one head.
template <int S, int HD, bool BWD>
__global__ __launch_bounds__(256) void group_attention_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                               float* __restrict__ out, float* __restrict__ lse,
                                                               float* __restrict__ dqkv, int64_t nb, int H, float scale) {
    constexpr int G = 32 / S;
    constexpr int TRN = BWD ? 32 * 33 : 1;
    __shared__ float tr[4][2][TRN];  // per wave: P and dS transposes (backward only)
    const int w = wave_id(), lane = lane_id();
    const int c = lane & 31, h2 = lane >> 5;
    const int64_t task = (int64_t)blockIdx.x * 4 + w;      // (group-block, head)
    const int head = (int)(task % H);
    const int64_t b0 = (task / H) * G;                      // first group of this wave
    if (b0 >= nb) return;
    // my column / row as operand lane: group g = c / S, token s = c % S
    const int64_t bl = b0 + c / S;
    const int sl = c % S;
    const bool ok = bl < nb;
    const float* qrow = qkv + GaPtrs<HD>::qkv(S, H, ok ? bl : 0, sl, 0, head);
    const float* krow = qkv + GaPtrs<HD>::qkv(S, H, ok ? bl : 0, sl, 1, head);
    const float* vrow = qkv + GaPtrs<HD>::qkv(S, H, ok ? bl : 0, sl, 2, head);

    float kq[HD / 2], qq[HD / 2];
    load_half_row<HD>(kq, krow, ok, h2);  // A operand: rows j = keys
    load_half_row<HD>(qq, qrow, ok, h2);  // B operand: cols i = queries
    f32x16 T;
#pragma unroll
    for (int i = 0; i < 16; ++i) T[i] = 0.f;
#pragma unroll
    for (int t = 0; t < HD / 2; ++t) T = __builtin_amdgcn_mfma_f32_32x32x2f32(kq[t], qq[t], T, 0, 0, 0);

    // softmax over keys j (registers x lane halves) for my query column c; other groups' keys masked
    float lse_i;
    f32x16 P;
    if (!BWD) {
        float m = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool same = (acc_row(r, h2) / S) == (c / S);
            T[r] = same ? T[r] * scale : -__builtin_inff();
            m = fmaxf(m, T[r]);
        }
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { P[r] = __expf(T[r] - m); l += P[r]; }
        l += __shfl_xor(l, 32);
        const float inv = 1.f / l;
#pragma unroll
        for (int r = 0; r < 16; ++r) P[r] *= inv;
        lse_i = m + __logf(l);
        if (ok && h2 == 0) lse[(bl * H + head) * S + sl] = lse_i;
    } else {
        lse_i = ok ? lse[(bl * H + head) * S + sl] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool same = (acc_row(r, h2) / S) == (c / S);
            P[r] = same ? __expf(T[r] * scale - lse_i) : 0.f;
        }
    }

    // A operand fetched in accumulator key order: x[(group of column block), key j(t,h2)][dblk*32 + c]
    auto key_ptr = [&](int which, int t, const float* base) -> const float* {
        const int j = acc_row(t, h2);
        const int64_t bj = b0 + j / S;
        return (bj < nb) ? base + GaPtrs<HD>::qkv(S, H, bj, j % S, which, head) : nullptr;
    };

    if (!BWD) {
        // O^T[d][i] = sum_j V^T[d][j] P^T[j][i]
#pragma unroll
        for (int dblk = 0; dblk < HD / 32; ++dblk) {
            f32x16 O;
#pragma unroll
            for (int i = 0; i < 16; ++i) O[i] = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float* vp = key_ptr(2, t, qkv);
                const float a = vp ? vp[dblk * 32 + c] : 0.f;
                O = __builtin_amdgcn_mfma_f32_32x32x2f32(a, P[t], O, 0, 0, 0);
            }
            if (ok) {
                float* op = out + GaPtrs<HD>::o(S, H, bl, sl, head) + dblk * 32 + 4 * h2;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(op + 8 * q) = make_float4(O[4 * q], O[4 * q + 1], O[4 * q + 2], O[4 * q + 3]);
            }
        }
        return;
    } else {
        const float* dorow = dout + GaPtrs<HD>::o(S, H, ok ? bl : 0, sl, head);
        // dP^T[j][i] = sum_d V[j][d] dO^T[d][i]
        float vv[HD / 2], dd[HD / 2];
        load_half_row<HD>(vv, vrow, ok, h2);
        load_half_row<HD>(dd, dorow, ok, h2);
        f32x16 dP;
#pragma unroll
        for (int i = 0; i < 16; ++i) dP[i] = 0.f;
#pragma unroll
        for (int t = 0; t < HD / 2; ++t) dP = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[t], dd[t], dP, 0, 0, 0);
        float delta = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) delta += P[r] * dP[r];
        delta += __shfl_xor(delta, 32);
        f32x16 dS;  // dS^T[j][i], softmax scale folded in
#pragma unroll
        for (int r = 0; r < 16; ++r) dS[r] = P[r] * (dP[r] - delta) * scale;

        // transposes through LDS (stride 33: conflict-free both ways)
        float* tp = tr[w][0];
        float* ts = tr[w][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            tp[acc_row(r, h2) * 33 + c] = P[r];    // [j][i]
            ts[acc_row(r, h2) * 33 + c] = dS[r];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): wave-private LDS, no barrier needed
        float Pn[16], dSn[16];               // P[i(t,h2)][j = c], dS[i(t,h2)][j = c]
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            Pn[t] = tp[c * 33 + acc_row(t, h2)];
            dSn[t] = ts[c * 33 + acc_row(t, h2)];
        }
        auto qry_o_ptr = [&](int t) -> const float* {  // dO row of query i(t,h2)
            const int i = acc_row(t, h2);
            const int64_t bi = b0 + i / S;
            return (bi < nb) ? dout + GaPtrs<HD>::o(S, H, bi, i % S, head) : nullptr;
        };
        float* dq = dqkv + GaPtrs<HD>::qkv(S, H, ok ? bl : 0, sl, 0, head);
        float* dk = dqkv + GaPtrs<HD>::qkv(S, H, ok ? bl : 0, sl, 1, head);
        float* dv = dqkv + GaPtrs<HD>::qkv(S, H, ok ? bl : 0, sl, 2, head);
#pragma unroll
        for (int dblk = 0; dblk < HD / 32; ++dblk) {
            f32x16 aQ, aK, aV;
#pragma unroll
            for (int i = 0; i < 16; ++i) { aQ[i] = 0.f; aK[i] = 0.f; aV[i] = 0.f; }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float* kp = key_ptr(1, t, qkv);   // K[j(t)][d]      -> dQ^T = K^T dS^T
                const float* qp = key_ptr(0, t, qkv);   // Q[i(t)][d]      -> dK^T = Q^T dS
                const float* op = qry_o_ptr(t);         // dO[i(t)][d]     -> dV^T = dO^T P
                const float ak = kp ? kp[dblk * 32 + c] : 0.f;
                const float aq = qp ? qp[dblk * 32 + c] : 0.f;
                const float ao = op ? op[dblk * 32 + c] : 0.f;
                aQ = __builtin_amdgcn_mfma_f32_32x32x2f32(ak, dS[t], aQ, 0, 0, 0);
                aK = __builtin_amdgcn_mfma_f32_32x32x2f32(aq, dSn[t], aK, 0, 0, 0);
                aV = __builtin_amdgcn_mfma_f32_32x32x2f32(ao, Pn[t], aV, 0, 0, 0);
            }
            if (ok) {
                const int off = dblk * 32 + 4 * h2;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    *reinterpret_cast<float4*>(dq + off + 8 * q) = make_float4(aQ[4 * q], aQ[4 * q + 1], aQ[4 * q + 2], aQ[4 * q + 3]);
                    *reinterpret_cast<float4*>(dk + off + 8 * q) = make_float4(aK[4 * q], aK[4 * q + 1], aK[4 * q + 2], aK[4 * q + 3]);
                    *reinterpret_cast<float4*>(dv + off + 8 * q) = make_float4(aV[4 * q], aV[4 * q + 1], aV[4 * q + 2], aV[4 * q + 3]);
                }
            }
        }
    }
}

template <bool BWD>
static int launch_group_attention(const float* qkv, const float* dout, float* out, float* lse, float* dqkv,
                                  int64_t nb, int s, int h, int hd, hipStream_t stream, const char* what) {
    PDA_REQUIRE(nb >= 0 && h >= 1, "%s: bad size", what);
    if (nb == 0) return PDA_OK;
    PDA_REQUIRE(qkv && lse && (BWD ? (dout && dqkv) : (out != nullptr)), "%s: null pointer", what);
    const int G = 32 / (s > 0 ? s : 1);
    const float scale = 1.0f / sqrtf((float)hd);
    void (*kern)(const float*, const float*, float*, float*, float*, int64_t, int, float) = nullptr;
#define PDA_GA_CASE(SS, DD) if (s == SS && hd == DD) kern = group_attention_kernel<SS, DD, BWD>
    PDA_GA_CASE(32, 64); PDA_GA_CASE(16, 64); PDA_GA_CASE(8, 64);
    PDA_GA_CASE(32, 128); PDA_GA_CASE(16, 128); PDA_GA_CASE(8, 128);
    PDA_GA_CASE(32, 32); PDA_GA_CASE(16, 32); PDA_GA_CASE(8, 32);
#undef PDA_GA_CASE
    if (!kern) {
        set_error("%s: no kernel built for seq=%d head_dim=%d", what, s, hd);
        return PDA_ERR_UNSUPPORTED;
    }
    const int64_t tasks = divup64(nb, G) * h;
    const int64_t blocks = divup64(tasks, 4);
    PDA_REQUIRE(blocks < INT32_MAX, "%s: too many groups", what);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, stream, qkv, dout, out, lse, dqkv, nb, h, scale);
    return check_launch(what);
}

}  // namespace pda

PDA_API int pda_group_attention_fwd(const float* qkv, float* out, float* lse, int64_t num_groups, int seq,
                                    int heads, int head_dim, pda_stream_t stream) {
    return pda::launch_group_attention<false>(qkv, nullptr, out, lse, nullptr, num_groups, seq, heads, head_dim,
                                              (hipStream_t)stream, "pda_group_attention_fwd");
}

PDA_API int pda_group_attention_bwd(const float* qkv, const float* grad_out, const float* lse, float* grad_qkv,
                                    int64_t num_groups, int seq, int heads, int head_dim, pda_stream_t stream) {
    return pda::launch_group_attention<true>(qkv, grad_out, nullptr, const_cast<float*>(lse), grad_qkv, num_groups,
                                             seq, heads, head_dim, (hipStream_t)stream, "pda_group_attention_bwd");
}
