// sa_xyz_grad.hip -- the COORDINATE columns of the first layer of a vanilla set-abstraction scale in the backward pass
// (include/pda_train.h).  z1 = [xyz[idx] - new_xyz | features[idx]] W1^T with W1 (c1, 3 + C) (QueryAndGroup + the first
// 1x1 convolution, pointnet2_utils.py:671-704 + pointnet2_modules.py:1657).  Its backward splits by column block:
//   * the C feature columns are a plain (tokens x C) problem: weight gradient on csrc/wgrad.hip, input gradient on
//     csrc/gemm_split.hip, both at full matrix-core width (C = 256);
//   * the 3 coordinate columns made the whole thing a 259-wide problem that went to the library.  Here they are one
//     streaming pass over grad_z1 (thread = output channel, tokens in order): dW1[:, 0:3] += g (x) (xyz[idx] - centre)
//     and, per centre, grad_new_xyz = -(sum over its samples of g) W1[:, 0:3] -- the centre enters every sample with -1.
// The pass reads grad_z1 once (tokens * c1 * 4 bytes) and nothing else of size.
#include "pda_common.h"

namespace pda {

constexpr int XG_BLOCKS = 512;
constexpr int XG_THREADS = 256;
constexpr int XG_MAXC = 4;      // channels per thread: c1 <= 1024

template <int CPT>
__global__ __launch_bounds__(XG_THREADS) void sa_xyz_grad_kernel(const float* __restrict__ g, const float* __restrict__ xyz,
                                                                 const float* __restrict__ new_xyz, const int32_t* __restrict__ idx,
                                                                 const float* __restrict__ w, int ldw, float* __restrict__ partial,
                                                                 float* __restrict__ grad_new, int n, int m, int ns, int c1, int64_t groups) {
    __shared__ float red[XG_THREADS / 64][3];
    const int tid = threadIdx.x;
    float accw[CPT][3], wx[CPT][3];
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int o = tid + q * XG_THREADS;
#pragma unroll
        for (int d = 0; d < 3; ++d) { accw[q][d] = 0.f; wx[q][d] = o < c1 ? w[(size_t)o * ldw + d] : 0.f; }
    }
    for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int bs = (int)(grp / m);
        const float* ct = new_xyz + grp * 3;
        const float cx = ct[0], cy = ct[1], cz = ct[2];
        const float* pts = xyz + (size_t)bs * n * 3;
        const int32_t* ids = idx + grp * ns;
        const float* grow = g + grp * ns * (int64_t)c1;
        float gs[CPT];
#pragma unroll
        for (int q = 0; q < CPT; ++q) gs[q] = 0.f;
        for (int s0 = 0; s0 < ns; s0 += 8) {
            float gv[8][CPT], dx[8], dy[8], dz[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = s0 + u < ns ? s0 + u : ns - 1;
                const float* pt = pts + (size_t)ids[s] * 3;
                dx[u] = pt[0] - cx; dy[u] = pt[1] - cy; dz[u] = pt[2] - cz;          // pointnet2_utils.py:692
#pragma unroll
                for (int q = 0; q < CPT; ++q) {
                    const int o = tid + q * XG_THREADS;
                    gv[u][q] = (s0 + u < ns && o < c1) ? grow[(int64_t)s * c1 + o] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < CPT; ++q) {
                    accw[q][0] = __builtin_fmaf(gv[u][q], dx[u], accw[q][0]);
                    accw[q][1] = __builtin_fmaf(gv[u][q], dy[u], accw[q][1]);
                    accw[q][2] = __builtin_fmaf(gv[u][q], dz[u], accw[q][2]);
                    gs[q] += gv[u][q];
                }
        }
        if (grad_new) {
            float v[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < CPT; ++q)
#pragma unroll
                for (int d = 0; d < 3; ++d) v[d] = __builtin_fmaf(gs[q], wx[q][d], v[d]);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) v[d] += __shfl_xor(v[d], o);
            }
            __syncthreads();                     // the previous group's result has been read
            if (lane_id() == 0) { red[wave_id()][0] = v[0]; red[wave_id()][1] = v[1]; red[wave_id()][2] = v[2]; }
            __syncthreads();
            if (tid < 3) grad_new[grp * 3 + tid] = -((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
        }
    }
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int o = tid + q * XG_THREADS;
        if (o < c1) {
            float* p = partial + ((size_t)blockIdx.x * c1 + o) * 3;
            p[0] = accw[q][0]; p[1] = accw[q][1]; p[2] = accw[q][2];
        }
    }
}

// 16 elements x 16 slices of the workgroup partials per block, slices summed in fixed order
__global__ __launch_bounds__(256) void sa_xyz_grad_reduce_kernel(const float* __restrict__ partial, int nblocks, int c1, float* __restrict__ dw, int lddw) {
    __shared__ double red[16][16];
    const int el = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    const bool ok = e < c1 * 3;
    double a = 0.0;
    if (ok)
        for (int k = part; k < nblocks; k += 16) a += (double)partial[(size_t)k * c1 * 3 + e];
    red[part][el] = a;
    __syncthreads();
    if (part == 0 && ok) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][el];
        dw[(size_t)(e / 3) * lddw + e % 3] = (float)t;
    }
}

// ---- forward: the first layer of a wide SA scale WITHOUT a per-token contraction ------------------------------------------
// z1[token] = W1 [xyz[idx] - centre | f[idx]] = (W_f f)[idx] + W_xyz (xyz[idx] - centre): a linear layer commutes with the
// gather.  The feature part is computed once per POINT (P = F W_f^T: b*n rows instead of b*m*nsample -- 4096 instead of
// 229 376 at ONCE layer 5: 0.5 GFLOP instead of 30) and gathered as rows here; the coordinate part is three FMAs per
// output on the centred coordinates themselves (no cancellation between large absolute coordinates).
// thread = 4 consecutive channels of one token; P rows (c1 floats) are L2-resident (4 MB), z1 leaves as 16-byte stores.
__global__ __launch_bounds__(256) void sa_point_gather_kernel(const float* __restrict__ prow, const float* __restrict__ xyz,
                                                              const float* __restrict__ new_xyz, const int32_t* __restrict__ idx,
                                                              const float* __restrict__ w, int ldw, const float* __restrict__ bias,
                                                              int relu, float* __restrict__ z, int n, int m, int ns, int c1,
                                                              int64_t tokens) {
    const int cq = c1 >> 2;                                         // channel quads per token
    // cq divides 256 (c1 in {128, 256, 512, 1024}: the launcher checks): a thread keeps ONE channel quad for the whole walk, so
    // its 12 coordinate weights are loaded once (per token they would be 12 loads at a 1 KB stride across the wave)
    const int q = threadIdx.x % cq;
    float wx[4], wy[4], wz[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float* wr = w + (size_t)(4 * q + k) * ldw;             // W1 row of this channel: columns 0..2 are the coordinates'
        wx[k] = wr[0]; wy[k] = wr[1]; wz[k] = wr[2];
    }
    float bq[4] = {0.f, 0.f, 0.f, 0.f};                             // inference: the folded BatchNorm shift, then ReLU
    if (bias) { const float4 b4 = *reinterpret_cast<const float4*>(bias + 4 * q); bq[0] = b4.x; bq[1] = b4.y; bq[2] = b4.z; bq[3] = b4.w; }
    const int tpb = 256 / cq;                                        // tokens per workgroup step
    for (int64_t tok = (int64_t)blockIdx.x * tpb + threadIdx.x / cq; tok < tokens; tok += (int64_t)gridDim.x * tpb) {
        const int64_t grp = tok / ns;
        const int bs = (int)(grp / m);
        const int id = idx[tok];
        const float* pt = xyz + ((size_t)bs * n + id) * 3;
        const float* ct = new_xyz + grp * 3;
        const float dx = pt[0] - ct[0], dy = pt[1] - ct[1], dz = pt[2] - ct[2];      // pointnet2_utils.py:692
        const float4 p4 = *reinterpret_cast<const float4*>(prow + ((size_t)bs * n + id) * c1 + 4 * q);
        const float pv[4] = {p4.x, p4.y, p4.z, p4.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[k] = __builtin_fmaf(wz[k], dz, __builtin_fmaf(wy[k], dy, __builtin_fmaf(wx[k], dx, pv[k]))) + bq[k];
            o[k] = relu ? fmaxf(o[k], 0.f) : o[k];
        }
        *reinterpret_cast<float4*>(z + tok * c1 + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace pda

PDA_API int pda_sa_point_gather(const float* point_rows, const float* xyz, const float* new_xyz, const int32_t* idx, const float* w, int ldw,
                                const float* bias, int relu, float* z, int b, int n, int m, int ns, int c1, pda_stream_t stream) {
    using namespace pda;
    PDA_REQUIRE(b >= 0 && n >= 1 && m >= 0 && ns >= 1 && c1 >= 4 && (c1 & 3) == 0 && ldw >= 3, "pda_sa_point_gather: bad size");
    const int64_t tokens = (int64_t)b * m * ns;
    if (tokens == 0) return PDA_OK;
    PDA_REQUIRE(point_rows && xyz && new_xyz && idx && w && z && (((uintptr_t)point_rows | (uintptr_t)z | (uintptr_t)bias) & 15) == 0,
                "pda_sa_point_gather: null or misaligned pointer");
    if (256 % (c1 >> 2) != 0) {
        set_error("pda_sa_point_gather: c1 = %d (c1 / 4 must divide 256)", c1);
        return PDA_ERR_UNSUPPORTED;
    }
    const int64_t total = tokens * (c1 >> 2);
    const int blocks = (int)(divup64(total, 256) < 8192 ? divup64(total, 256) : 8192);
    hipLaunchKernelGGL(sa_point_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, point_rows, xyz, new_xyz, idx, w, ldw, bias, relu, z,
                       n, m, ns, c1, tokens);
    return check_launch("pda_sa_point_gather");
}

PDA_API int64_t pda_sa_xyz_grad_scratch_bytes(int c1) { return c1 > 0 ? (int64_t)pda::XG_BLOCKS * c1 * 3 * sizeof(float) : 0; }

PDA_API int pda_sa_xyz_grad(const float* grad_z1, const float* xyz, const float* new_xyz, const int32_t* idx, const float* w, int ldw,
                            float* dw, int lddw, float* grad_new_xyz, void* scratch, int b, int n, int m, int ns, int c1,
                            pda_stream_t stream) {
    using namespace pda;
    PDA_REQUIRE(b >= 0 && n >= 1 && m >= 0 && ns >= 1 && c1 >= 1 && ldw >= 3 && lddw >= 3, "pda_sa_xyz_grad: bad size");
    const int64_t groups = (int64_t)b * m;
    if (groups == 0) return PDA_OK;
    PDA_REQUIRE(grad_z1 && xyz && new_xyz && idx && w && dw && scratch, "pda_sa_xyz_grad: null pointer");
    if (c1 > XG_THREADS * XG_MAXC) {
        set_error("pda_sa_xyz_grad: c1 = %d > %d", c1, XG_THREADS * XG_MAXC);
        return PDA_ERR_UNSUPPORTED;
    }
    const hipStream_t s = (hipStream_t)stream;
    const int blocks = (int)(groups < XG_BLOCKS ? groups : XG_BLOCKS);
    float* partial = (float*)scratch;
    const int cpt = divup(c1, XG_THREADS);
#define PDA_XG(CPT) hipLaunchKernelGGL((sa_xyz_grad_kernel<CPT>), dim3(blocks), dim3(XG_THREADS), 0, s, grad_z1, xyz, new_xyz, idx, w, ldw, \
                                       partial, grad_new_xyz, n, m, ns, c1, groups)
    if (cpt == 1) PDA_XG(1); else if (cpt == 2) PDA_XG(2); else PDA_XG(4);
#undef PDA_XG
    hipLaunchKernelGGL(sa_xyz_grad_reduce_kernel, dim3(divup(c1 * 3, 16)), dim3(256), 0, s, (const float*)partial, blocks, c1, dw, lddw);
    return check_launch("pda_sa_xyz_grad");
}
