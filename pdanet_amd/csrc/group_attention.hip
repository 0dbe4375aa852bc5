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
//     takes the contiguous half [h*hd/2, (h+1)*hd/2) of the head dimension of its token; the
//     32-token tiles of the (B*np, S, 3, H, hd) in_proj output are read from HBM as whole rows and
//     turned into that lane = token layout through a wave-private LDS image (struct Tile);
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
    // element (row, which, head, d) of qkv (rows, 3, H, HD); row = b * S + s in the dense layout
    __device__ __forceinline__ static size_t qkv(int H, size_t row, int which, int head) {
        return ((row * 3 + which) * H + head) * (size_t)HD;
    }
    __device__ __forceinline__ static size_t o(int H, size_t row, int head) { return (row * H + head) * (size_t)HD; }
};

// A 32-token x HD tile moves between HBM and the MFMA operand layout through a wave-private LDS
// image: HBM side fully coalesced (each wave instruction covers 64/(HD/4) whole rows with 16-byte
// lanes), operand side lane = token.  Reading operands straight from HBM with lane = token uses
// 16 bytes of every 128-byte line per instruction and thrashes the 32 KB L1 (measured 2.1 TB/s).
// Row stride HD+4 floats: conflict-free for ds_write_b128 (8-lane groups) and ds_read_b128.
template <int HD>
struct Tile {
    static constexpr int LD = HD + 4;
    static constexpr int LPR = HD / 4;    // lanes per row
    static constexpr int RPI = 64 / LPR;  // rows per wave instruction
    static constexpr int NI = HD / 8;     // wave instructions per 32-row tile

    template <class RowPtr>
    __device__ __forceinline__ static void load(float4 (&raw)[NI], RowPtr row_ptr, int lane) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            bool valid;
            const auto* p = row_ptr(it * RPI + lane / LPR, valid);  // always dereferenceable (clamped)
            const float4 v = load4(p + 4 * (lane % LPR));
            raw[it] = valid ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // registers -> LDS image -> operand: lane (c, h) gets row c, columns h*HD/2 + t
    __device__ __forceinline__ static void to_operand(float (&dst)[HD / 2], const float4 (&raw)[NI], float* lds, int lane) {
#pragma unroll
        for (int it = 0; it < NI; ++it)
            *reinterpret_cast<float4*>(lds + (it * RPI + lane / LPR) * LD + 4 * (lane % LPR)) = raw[it];
        const float* src = lds + (lane & 31) * LD + (lane >> 5) * (HD / 2);
#pragma unroll
        for (int q = 0; q < HD / 8; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(src + 4 * q);
            dst[4 * q + 0] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w;
        }
    }
    // accumulator of X^T[d][token] (32 columns dblk*32..) -> LDS image rows = tokens
    __device__ __forceinline__ static void from_acc(float* lds, const f32x16& acc, int dblk, int lane) {
        float* dst = lds + (lane & 31) * LD + dblk * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(dst + 8 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
    template <class RowPtr>
    __device__ __forceinline__ static void store(const float* lds, RowPtr row_ptr, int lane) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int r = it * RPI + lane / LPR;
            auto* p = row_ptr(r);
            const float4 v = *reinterpret_cast<const float4*>(lds + r * LD + 4 * (lane % LPR));
            if (p) store4(p + 4 * (lane % LPR), v);
        }
    }
};

// S: tokens per group (8, 16 or 32).  A wave covers G = 32/S groups of one head.
// RAGGED (csrc/ragged.hip): the rows are COMPACT -- group b owns rows [off[b], off[b] + cnt[b]), cnt[b] <= S distinct
// tokens; slots >= cnt[b] are the repeats of token 0 the dense layout would hold.  They are not computed: their keys
// are masked and key 0 enters the softmax with weight S - cnt + 1 (score + log of it), which is exactly the sum over
// the repeated keys; their queries (identical to query 0) are not evaluated.  The tile logic is unchanged.
template <int S, int HD, bool BWD, typename TIO, bool RAGGED>
__global__ __launch_bounds__(256, 2) void group_attention_kernel(const TIO* __restrict__ qkv, const TIO* __restrict__ dout,
                                                               TIO* __restrict__ out, float* __restrict__ lse,
                                                               TIO* __restrict__ dqkv, int64_t nb, int H, float scale,
                                                               const int32_t* __restrict__ cnt, const int32_t* __restrict__ off) {
    constexpr int G = 32 / S;
    using TL = Tile<HD>;
    constexpr int TRN = 2 * 32 * 33;  // P and dS transposes (backward), stride 33
    constexpr int LDSW = (BWD && TRN > 32 * TL::LD) ? TRN : 32 * TL::LD;
    __shared__ __attribute__((aligned(16))) float lds_all[4][LDSW];  // wave-private images: no barriers
    const int w = wave_id(), lane = lane_id();
    float* lds = lds_all[w];
    const int c = lane & 31, h2 = lane >> 5;
    const int64_t task = (int64_t)blockIdx.x * 4 + w;      // (group-block, head)
    const int head = (int)(task % H);
    const int64_t b0 = (task / H) * G;                      // first group of this wave
    if (b0 >= nb) return;
    // RAGGED: tokens and first compact row of the wave's G groups (wave-uniform); a group past the end has 0 tokens
    int gn[G];
    int64_t go[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const bool gv = b0 + g < nb;
        gn[g] = RAGGED ? (gv ? cnt[b0 + g] : 0) : (gv ? S : 0);
        go[g] = RAGGED ? off[gv ? b0 + g : b0] : (b0 + g) * S;
    }
    auto g_n = [&](int g) -> int { int v = gn[0];
#pragma unroll
        for (int q = 1; q < G; ++q) v = g == q ? gn[q] : v;
        return v; };
    auto g_o = [&](int g) -> int64_t { int64_t v = go[0];
#pragma unroll
        for (int q = 1; q < G; ++q) v = g == q ? go[q] : v;
        return v; };
    // tile row r = (group r / S, slot r % S) -> its row in HBM; invalid rows read group b0's token 0 and are zeroed
    auto tile_row = [&](int r, bool& valid) -> int64_t {
        const int g = r / S, sl_ = r % S;
        valid = sl_ < g_n(g);
        return valid ? g_o(g) + sl_ : go[0];
    };
    // my column / row as operand lane: group g = c / S, token s = c % S
    const int64_t bl = b0 + c / S;
    const int sl = c % S;
    const bool ok = bl < nb && sl < g_n(c / S);
    // row r of the wave's 32-token tile
    // Loads never branch on validity (a divergent branch around a load serialises load -> wait ->
    // MFMA): rows of groups past the end read group b0's row instead and are zeroed by a select.
    auto in_row = [&](int which) {
        return [&, which](int r, bool& valid) -> const TIO* {
            return qkv + GaPtrs<HD>::qkv(H, (size_t)tile_row(r, valid), which, head);
        };
    };
    auto do_row = [&](int r, bool& valid) -> const TIO* {
        return dout + GaPtrs<HD>::o(H, (size_t)tile_row(r, valid), head);
    };
    // key j of the tile as seen by my query column: inside my group and a real (distinct) token?  bias = log of the
    // multiplicity of key 0 (RAGGED); a group past the end keeps key 0 so that its (unused) column stays finite
    auto key_ok = [&](int j) -> bool {
        return (j / S) == (c / S) && (j % S) < max(g_n(j / S), 1);
    };
    const float lnw = RAGGED ? __logf((float)(S - max(g_n(c / S), 1) + 1)) : 0.f;

    float kq[HD / 2], qq[HD / 2];
    {
        float4 rk[TL::NI], rq[TL::NI];
        TL::load(rk, in_row(1), lane);
        TL::load(rq, in_row(0), lane);
        TL::to_operand(kq, rk, lds, lane);  // A operand: rows j = keys
        TL::to_operand(qq, rq, lds, lane);  // B operand: cols i = queries
    }
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
            const int j = acc_row(r, h2);
            T[r] = key_ok(j) ? T[r] * scale + ((RAGGED && (j % S) == 0) ? lnw : 0.f) : -__builtin_inff();
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
            const int j = acc_row(r, h2);
            P[r] = key_ok(j) ? __expf(T[r] * scale + ((RAGGED && (j % S) == 0) ? lnw : 0.f) - lse_i) : 0.f;
        }
    }

    // A operand fetched in accumulator key order: x[key j(t,h2)][dblk*32 + c] (128-byte rows per half wave)
    auto key_val = [&](int which, int t, int col) -> float {
        bool valid;
        const int64_t row = tile_row(acc_row(t, h2), valid);
        const float v = load1(qkv + GaPtrs<HD>::qkv(H, (size_t)row, which, head) + col);
        return valid ? v : 0.f;
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
                O = __builtin_amdgcn_mfma_f32_32x32x2f32(key_val(2, t, dblk * 32 + c), P[t], O, 0, 0, 0);
            }
            TL::from_acc(lds, O, dblk, lane);
        }
        TL::store(lds, [&](int r) -> TIO* {
            bool valid;
            const int64_t row = tile_row(r, valid);
            return valid ? out + GaPtrs<HD>::o(H, (size_t)row, head) : nullptr;
        }, lane);
        return;
    } else {
        // dP^T[j][i] = sum_d V[j][d] dO^T[d][i]
        float vv[HD / 2], dd[HD / 2];
        {
            float4 rv[TL::NI], rd[TL::NI];
            TL::load(rv, in_row(2), lane);
            TL::load(rd, do_row, lane);
            TL::to_operand(vv, rv, lds, lane);
            TL::to_operand(dd, rd, lds, lane);
        }
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
        float* tp = lds;
        float* ts = lds + 32 * 33;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            tp[acc_row(r, h2) * 33 + c] = P[r];    // [j][i]
            ts[acc_row(r, h2) * 33 + c] = dS[r];
        }
        float Pn[16], dSn[16];               // P[i(t,h2)][j = c], dS[i(t,h2)][j = c]
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            Pn[t] = tp[c * 33 + acc_row(t, h2)];
            dSn[t] = ts[c * 33 + acc_row(t, h2)];
        }
        auto qry_o_val = [&](int t, int col) -> float {  // dO[query i(t,h2)][col]
            bool valid;
            const int64_t row = tile_row(acc_row(t, h2), valid);
            const float v = load1(dout + GaPtrs<HD>::o(H, (size_t)row, head) + col);
            return valid ? v : 0.f;
        };
        auto out_row = [&](int which) {
            return [&, which](int r) -> TIO* {
                bool valid;
                const int64_t row = tile_row(r, valid);
                return valid ? dqkv + GaPtrs<HD>::qkv(H, (size_t)row, which, head) : nullptr;
            };
        };
        // dQ^T = K^T dS^T
#pragma unroll
        for (int dblk = 0; dblk < HD / 32; ++dblk) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(key_val(1, t, dblk * 32 + c), dS[t], acc, 0, 0, 0);
            }
            TL::from_acc(lds, acc, dblk, lane);
        }
        TL::store(lds, out_row(0), lane);
        // dK^T = Q^T dS
#pragma unroll
        for (int dblk = 0; dblk < HD / 32; ++dblk) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(key_val(0, t, dblk * 32 + c), dSn[t], acc, 0, 0, 0);
            }
            TL::from_acc(lds, acc, dblk, lane);
        }
        TL::store(lds, out_row(1), lane);
        // dV^T = dO^T P
#pragma unroll
        for (int dblk = 0; dblk < HD / 32; ++dblk) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qry_o_val(t, dblk * 32 + c), Pn[t], acc, 0, 0, 0);
            }
            TL::from_acc(lds, acc, dblk, lane);
        }
        TL::store(lds, out_row(2), lane);
    }
}

template <bool BWD, typename TIO, bool RAGGED = false>
static int launch_group_attention(const TIO* qkv, const TIO* dout, TIO* out, float* lse, TIO* dqkv,
                                  int64_t nb, int s, int h, int hd, hipStream_t stream, const char* what,
                                  const int32_t* cnt = nullptr, const int32_t* off = nullptr) {
    PDA_REQUIRE(nb >= 0 && h >= 1, "%s: bad size", what);
    if (nb == 0) return PDA_OK;
    PDA_REQUIRE(qkv && lse && (BWD ? (dout && dqkv) : (out != nullptr)), "%s: null pointer", what);
    PDA_REQUIRE(!RAGGED || (cnt && off), "%s: null pointer", what);
    const int G = 32 / (s > 0 ? s : 1);
    const float scale = 1.0f / sqrtf((float)hd);
    void (*kern)(const TIO*, const TIO*, TIO*, float*, TIO*, int64_t, int, float, const int32_t*, const int32_t*) = nullptr;
#define PDA_GA_CASE(SS, DD) if (s == SS && hd == DD) kern = group_attention_kernel<SS, DD, BWD, TIO, RAGGED>
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
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, stream, qkv, dout, out, lse, dqkv, nb, h, scale, cnt, off);
    return check_launch(what);
}

}  // namespace pda

PDA_API int pda_group_attention_fwd(const float* qkv, float* out, float* lse, int64_t num_groups, int seq,
                                    int heads, int head_dim, pda_stream_t stream) {
    return pda::launch_group_attention<false, float>(qkv, nullptr, out, lse, nullptr, num_groups, seq, heads, head_dim,
                                              (hipStream_t)stream, "pda_group_attention_fwd");
}

PDA_API int pda_group_attention_bwd(const float* qkv, const float* grad_out, const float* lse, float* grad_qkv,
                                    int64_t num_groups, int seq, int heads, int head_dim, pda_stream_t stream) {
    return pda::launch_group_attention<true, float>(qkv, grad_out, nullptr, const_cast<float*>(lse), grad_qkv, num_groups,
                                             seq, heads, head_dim, (hipStream_t)stream, "pda_group_attention_bwd");
}

PDA_API int pda_group_attention_fwd_bf16(const uint16_t* qkv, uint16_t* out, float* lse, int64_t num_groups, int seq,
                                         int heads, int head_dim, pda_stream_t stream) {
    return pda::launch_group_attention<false, pda::bf16_t>(qkv, nullptr, out, lse, nullptr, num_groups, seq, heads, head_dim,
                                                           (hipStream_t)stream, "pda_group_attention_fwd_bf16");
}

PDA_API int pda_group_attention_bwd_bf16(const uint16_t* qkv, const uint16_t* grad_out, const float* lse, uint16_t* grad_qkv,
                                         int64_t num_groups, int seq, int heads, int head_dim, pda_stream_t stream) {
    return pda::launch_group_attention<true, pda::bf16_t>(qkv, grad_out, nullptr, const_cast<float*>(lse), grad_qkv, num_groups,
                                                          seq, heads, head_dim, (hipStream_t)stream, "pda_group_attention_bwd_bf16");
}

// Compact ("ragged") rows: qkv (U, 3, H, hd), out / grad (U, H * hd), lse (groups, H, seq); see csrc/ragged.hip.
PDA_API int pda_group_attention_ragged_fwd(const float* qkv, const int32_t* cnt, const int32_t* off, float* out, float* lse,
                                           int64_t num_groups, int seq, int heads, int head_dim, pda_stream_t stream) {
    return pda::launch_group_attention<false, float, true>(qkv, nullptr, out, lse, nullptr, num_groups, seq, heads, head_dim,
                                                           (hipStream_t)stream, "pda_group_attention_ragged_fwd", cnt, off);
}

PDA_API int pda_group_attention_ragged_bwd(const float* qkv, const float* grad_out, const float* lse, const int32_t* cnt,
                                           const int32_t* off, float* grad_qkv, int64_t num_groups, int seq, int heads,
                                           int head_dim, pda_stream_t stream) {
    return pda::launch_group_attention<true, float, true>(qkv, grad_out, nullptr, const_cast<float*>(lse), grad_qkv, num_groups, seq,
                                                          heads, head_dim, (hipStream_t)stream, "pda_group_attention_ragged_bwd", cnt, off);
}
