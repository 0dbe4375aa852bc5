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

    // row_off(r, valid): element offset of tile row r from `base` (wave-uniform pointer, 32-bit lane offsets: one address
    // register per load instead of two, and no 64-bit lane arithmetic); always dereferenceable (clamped)
    template <class TIO, class RowOff>
    __device__ __forceinline__ static void load(float4 (&raw)[NI], const TIO* base, RowOff row_off, int lane) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            bool valid;
            const unsigned o = row_off(it * RPI + lane / LPR, valid) + 4u * (unsigned)(lane % LPR);
            const float4 v = load4(base + o);
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
    template <class TIO, class RowOff>
    __device__ __forceinline__ static void store(const float* lds, TIO* base, RowOff row_off, int lane) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int r = it * RPI + lane / LPR;
            bool valid;
            const unsigned o = row_off(r, valid) + 4u * (unsigned)(lane % LPR);
            const float4 v = *reinterpret_cast<const float4*>(lds + r * LD + 4 * (lane % LPR));
            if (valid) store4(base + o, v);
        }
    }
};

// One 32-row tile of one head: rows [base, base + rows) of the (compact or dense) token axis, holding whole groups.
// Per lane (operand column / query c = lane & 31): lo = first tile row of its group, n = tokens of its group, gid = the
// group; a lane past `rows` has lo = c, n = 1 (its own zeroed row as the only key: the unused column stays finite and
// contributes zeros to every contraction over queries).
// RAGGED (csrc/ragged.hip): a group's rows are its cnt <= S DISTINCT tokens; the S - cnt repeats of token 0 the dense
// layout would hold are not computed: key 0 enters the softmax with weight S - cnt + 1 (score + log of it), which is
// exactly the sum over the repeated keys; their queries (identical to query 0) are not evaluated.
template <int S, int HD, bool BWD, typename TIO, bool RAGGED>
__device__ __forceinline__ void group_attention_tile(const TIO* __restrict__ qkv, const TIO* __restrict__ dout,
                                                     TIO* __restrict__ out, float* __restrict__ lse, TIO* __restrict__ dqkv,
                                                     int H, int head, float scale, float* lds, int lane, int64_t base, int rows,
                                                     int lo, int n, int64_t gid) {
    using TL = Tile<HD>;
    const int c = lane & 31, h2 = lane >> 5;
    const bool ok = c < rows;
    const int sl = c - lo;
    // the tile's first row in HBM (wave-uniform pointers) and element offsets of tile row r from it; rows past the end read
    // row `base` instead and are zeroed by a select (loads never branch on validity: a divergent branch around a load
    // serialises load -> wait -> MFMA)
    const TIO* qb = qkv + GaPtrs<HD>::qkv(H, (size_t)base, 0, head);
    const TIO* dob = BWD ? dout + GaPtrs<HD>::o(H, (size_t)base, head) : nullptr;
    const unsigned OS = (unsigned)(H * HD), RS = 3u * OS;   // elements per out / qkv row
    auto in_row = [&](int which) {
        return [&, which](int r, bool& valid) -> unsigned {
            valid = r < rows;
            return (valid ? (unsigned)r * RS : 0u) + (unsigned)which * OS;
        };
    };
    auto o_row = [&](int r, bool& valid) -> unsigned {
        valid = r < rows;
        return valid ? (unsigned)r * OS : 0u;
    };
    // key j of the tile as seen by my query column: a token of my group?
    auto key_ok = [&](int j) -> bool { return j >= lo && j < lo + n; };
    const float lnw = (RAGGED && ok) ? __logf((float)(S - n + 1)) : 0.f;   // log of the multiplicity of key 0

    float kq[HD / 2], qq[HD / 2];
    {
        float4 rk[TL::NI], rq[TL::NI];
        TL::load(rk, qb, in_row(1), lane);
        TL::load(rq, qb, in_row(0), lane);
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
            T[r] = key_ok(j) ? T[r] * scale + ((RAGGED && j == lo) ? lnw : 0.f) : -__builtin_inff();
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
        if (ok && h2 == 0) lse[(gid * H + head) * S + sl] = lse_i;
    } else {
        lse_i = ok ? lse[(gid * H + head) * S + sl] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = acc_row(r, h2);
            P[r] = key_ok(j) ? __expf(T[r] * scale + ((RAGGED && j == lo) ? lnw : 0.f) - lse_i) : 0.f;
        }
    }

    // A operand fetched in accumulator key order: x[key j(t,h2)][dblk*32 + c] (128-byte rows per half wave)
    auto key_val = [&](int which, int t, int col) -> float {
        bool valid;
        const unsigned o = in_row(which)(acc_row(t, h2), valid) + (unsigned)col;
        const float v = load1(qb + o);
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
        TL::store(lds, out + GaPtrs<HD>::o(H, (size_t)base, head), o_row, lane);
    } else {
        // dP^T[j][i] = sum_d V[j][d] dO^T[d][i]
        float vv[HD / 2], dd[HD / 2];
        {
            float4 rv[TL::NI], rd[TL::NI];
            TL::load(rv, qb, in_row(2), lane);
            TL::load(rd, dob, o_row, lane);
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
            const unsigned o = o_row(acc_row(t, h2), valid) + (unsigned)col;
            const float v = load1(dob + o);
            return valid ? v : 0.f;
        };
        TIO* dqb = dqkv + GaPtrs<HD>::qkv(H, (size_t)base, 0, head);
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
        TL::store(lds, dqb, in_row(0), lane);
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
        TL::store(lds, dqb, in_row(1), lane);
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
        TL::store(lds, dqb, in_row(2), lane);
    }
}

// Groups one wave owns (kb): dense, the 32/S groups of one tile; RAGGED, kb consecutive groups whose distinct tokens it
// packs greedily into as few 32-row tiles as they need (round 3: a group of 32 slots holds ~10 distinct tokens on the
// ONCE scenes, so one group per tile left two thirds of every MFMA and of every load slot empty).  The launcher picks kb
// from the mean count: the groups that fill ONE tile on average (a wave that walks several tiles runs them back to back
// with nothing overlapping its loads; separate waves overlap each other: measured, a second tile per wave costs 10-20 %),
// more -- up to GA_BLOCK_ROWS distinct tokens -- only while at least GA_MIN_WAVES waves remain.
constexpr int GA_BLOCK_ROWS = 128;
constexpr int64_t GA_MIN_WAVES = 8192;

// S: tokens per group (8, 16 or 32).  RAGGED: the rows are COMPACT -- group b owns rows [off[b], off[b] + cnt[b]).
template <int S, int HD, bool BWD, typename TIO, bool RAGGED>
__global__ __launch_bounds__(256, 2) void group_attention_kernel(const TIO* __restrict__ qkv, const TIO* __restrict__ dout,
                                                               TIO* __restrict__ out, float* __restrict__ lse,
                                                               TIO* __restrict__ dqkv, int64_t nb, int H, float scale,
                                                               const int32_t* __restrict__ cnt, const int32_t* __restrict__ off,
                                                               int kb) {
    const int KB = RAGGED ? kb : 32 / S;
    using TL = Tile<HD>;
    constexpr int TRN = 2 * 32 * 33;  // P and dS transposes (backward), stride 33
    constexpr int LDSW = (BWD && TRN > 32 * TL::LD) ? TRN : 32 * TL::LD;
    __shared__ __attribute__((aligned(16))) float lds_all[4][LDSW];  // wave-private images: no barriers
    const int w = wave_id(), lane = lane_id();
    float* lds = lds_all[w];
    const int c = lane & 31;
    const int64_t task = (int64_t)blockIdx.x * 4 + w;      // (group block, head)
    const int head = (int)(task % H);
    const int64_t b0 = (task / H) * KB;                     // first group of this wave
    if (b0 >= nb) return;
    if (!RAGGED) {
        const int rows = (int)min((int64_t)KB, nb - b0) * S;
        group_attention_tile<S, HD, BWD, TIO, false>(qkv, dout, out, lse, dqkv, H, head, scale, lds, lane, b0 * S, rows,
                                                     c < rows ? (c / S) * S : c, c < rows ? S : 1, b0 + c / S);
    } else {
        const int64_t bend = min(nb, b0 + (int64_t)KB);
        int64_t g = b0;
#pragma nounroll
        while (g < bend) {                                  // every pass takes its first group (cn <= 32, off[g] == base): the loop ends
            // the lane number is re-defined opaquely in every pass: the ~250 per-lane offsets the tile derives from it are loop
            // invariants otherwise, and hoisted out of the loop they stay live across it (HD = 128 backward: 492 bytes of scratch)
            int lv = lane;
            asm volatile("" : "+v"(lv));
            const int c = lv & 31;
            const int64_t base = off[g];
            int rows = 0, lo = c, n = 1;
            int64_t gid = g;
            do {                                            // wave-uniform: the groups that still fit into this tile
                const int cn = min(max(cnt[g], 0), S);
                if (rows + cn > 32 || off[g] != base + rows) break;
                if (c >= rows && c < rows + cn) { lo = rows; n = cn; gid = g; }
                rows += cn;
                ++g;
            } while (g < bend);
            if (rows > 0)
                group_attention_tile<S, HD, BWD, TIO, true>(qkv, dout, out, lse, dqkv, H, head, scale, lds, lv, base, rows, lo,
                                                            n, gid);
        }
    }
}

template <bool BWD, typename TIO, bool RAGGED = false>
static int launch_group_attention(const TIO* qkv, const TIO* dout, TIO* out, float* lse, TIO* dqkv,
                                  int64_t nb, int s, int h, int hd, hipStream_t stream, const char* what,
                                  const int32_t* cnt = nullptr, const int32_t* off = nullptr, int64_t tokens = 0) {
    PDA_REQUIRE(nb >= 0 && h >= 1 && s >= 1 && s <= 32 && tokens >= 0, "%s: bad size", what);
    if (nb == 0) return PDA_OK;
    PDA_REQUIRE(qkv && lse && (BWD ? (dout && dqkv) : (out != nullptr)), "%s: null pointer", what);
    PDA_REQUIRE(!RAGGED || (cnt && off), "%s: null pointer", what);
    int G = 32 / s;                                           // groups per wave (group_attention_kernel)
    if (RAGGED) {
        const int64_t mean = std::max<int64_t>(1, std::min<int64_t>(s, divup64(tokens, nb)));
        const int64_t cap = std::max<int64_t>(1, nb * h / GA_MIN_WAVES);
        G = (int)std::max<int64_t>(32 / mean, std::min<int64_t>(GA_BLOCK_ROWS / mean, cap));
        G = std::max(1, std::min(G, 64));
    }
    const float scale = 1.0f / sqrtf((float)hd);
    void (*kern)(const TIO*, const TIO*, TIO*, float*, TIO*, int64_t, int, float, const int32_t*, const int32_t*, int) = nullptr;
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
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, stream, qkv, dout, out, lse, dqkv, nb, h, scale, cnt, off, G);
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
                                           int64_t tokens, int64_t num_groups, int seq, int heads, int head_dim,
                                           pda_stream_t stream) {
    return pda::launch_group_attention<false, float, true>(qkv, nullptr, out, lse, nullptr, num_groups, seq, heads, head_dim,
                                                           (hipStream_t)stream, "pda_group_attention_ragged_fwd", cnt, off, tokens);
}

PDA_API int pda_group_attention_ragged_bwd(const float* qkv, const float* grad_out, const float* lse, const int32_t* cnt,
                                           const int32_t* off, float* grad_qkv, int64_t tokens, int64_t num_groups, int seq,
                                           int heads, int head_dim, pda_stream_t stream) {
    return pda::launch_group_attention<true, float, true>(qkv, grad_out, nullptr, const_cast<float*>(lse), grad_qkv, num_groups, seq,
                                                          heads, head_dim, (hipStream_t)stream, "pda_group_attention_ragged_bwd", cnt, off,
                                                          tokens);
}
