// head_loss.hip -- the loss terms of the IA-SSD head, each as ONE launch that produces the term AND its gradient.
//
// The reference computes every term as a chain of elementwise torch operators over a few thousand rows
// (/root/reference/pcdet/models/dense_heads/IASSD_head.py:525-735, :1239-1321; pcdet/utils/loss_utils.py:75-194, :340-363);
// restated in torch that is ~700 launches of 2-3 us per iteration, forward and backward (2.3 ms of device time for a few
// hundred kFLOP).  The tensors are tiny -- B x 1024 centres, B x 4096 / 2048 instance-aware points -- so one workgroup per term
// does the whole job: a first pass over the rows for the normalisers (positive counts, per-instance means), a second pass for
// the per-row loss and d(term)/d(prediction), block reductions in double.  The backward pass of the autograd node is one
// multiplication of the stored gradient with the incoming scalar.
//   head_cls_loss_kernel      WeightedClassificationLoss (sigmoid cross-entropy with soft one-hot targets), centre and
//                             instance-aware classification (IASSD_head.py:637-664, :668-735)
//   head_centerness_kernel    generate_center_ness_mask (:795-817)
//   head_box_loss_kernel      get_center_box_binori_layer_loss (:1239-1281)
//   head_vote_loss_kernel     get_contextual_vote_loss (:525-548) and _ver2 (:579-619)
//   head_corner_loss_kernel   get_corner_layer_loss (:1307-1321) through PointResidual_BinOri_Coder.decode_torch
//                             (box_coder_utils.py:266-319) and boxes_to_corners_3d (box_utils.py:28-53)
#include "pda_common.h"

namespace pda {

constexpr int HL_THREADS = 1024;

// sum over the workgroup, result in every thread; red: 17 doubles of LDS
__device__ __forceinline__ double hl_block_sum(double v, double* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    __syncthreads();                     // red may still be read from the previous reduction
    if (lane == 0) red[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int i = 0; i < HL_THREADS / 64; ++i) s += red[i];
        red[16] = s;
    }
    __syncthreads();
    return red[16];
}

__device__ __forceinline__ float hl_smooth_l1(float d, float beta, float& dd) {      // value and derivative wrt d
    const float n = fabsf(d);
    if (beta < 1e-5f) { dd = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); return n; }
    if (n < beta) { dd = d / beta; return 0.5f * n * n / beta; }
    dd = d > 0.f ? 1.f : -1.f;
    return n - 0.5f * beta;
}

// loss = scale * sum_rows w_row * mean_c bce(x_rc, t_rc),  w_row = [label >= 0] / max(#(label > 0), 1),
// t_rc = [label == c + 1] * soft_row;  out = {loss, #positives};  grad has the layout of preds (row_stride columns).
__global__ __launch_bounds__(HL_THREADS) void head_cls_loss_kernel(const float* __restrict__ preds, int row_stride, int col0, int C,
                                                                   const int64_t* __restrict__ labels, const float* __restrict__ soft,
                                                                   int64_t n, float scale, float* __restrict__ out,
                                                                   float* __restrict__ grad) {
    __shared__ double red[17];
    double cnt = 0;
    for (int64_t r = threadIdx.x; r < n; r += HL_THREADS) cnt += labels[r] > 0 ? 1.0 : 0.0;
    const double npos = hl_block_sum(cnt, red);
    const float wnorm = 1.f / (float)(npos > 1.0 ? npos : 1.0);
    double acc = 0;
    for (int64_t r = threadIdx.x; r < n; r += HL_THREADS) {
        const int64_t lab = labels[r];
        const float w = lab >= 0 ? wnorm : 0.f;
        const float s = soft ? soft[r] : 1.f;
        const float* x = preds + r * row_stride;
        float* g = grad + r * row_stride;
        for (int c = 0; c < col0; ++c) g[c] = 0.f;
        for (int c = col0 + C; c < row_stride; ++c) g[c] = 0.f;
        float rowloss = 0.f;
        for (int c = 0; c < C; ++c) {
            const float v = x[col0 + c];
            const float t = lab == c + 1 ? s : 0.f;
            const float e = expf(-fabsf(v));
            rowloss += fmaxf(v, 0.f) - v * t + log1pf(e);
            const float sig = v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
            g[col0 + c] = scale * w * (sig - t) / (float)C;
        }
        acc += (double)(w * rowloss) / C;
    }
    const double total = hl_block_sum(acc, red);
    if (threadIdx.x == 0) { out[0] = scale * (float)total; out[1] = (float)npos; }
}

// cube root of prod_axes min(d-, d+) / max(d-, d+) of a positive centre inside its box (clamped at 1e-6), 0 elsewhere
__global__ __launch_bounds__(256) void head_centerness_kernel(const float* __restrict__ centers, const float* __restrict__ gt,
                                                              const int64_t* __restrict__ labels, float* __restrict__ out, int64_t n) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float* c = centers + p * 4 + 1;
    const float* g = gt + p * 8;
    const float dx = c[0] - g[0], dy = c[1] - g[1], dz = c[2] - g[2];
    const float a = -g[6], ca = cosf(a), sa = sinf(a);
    const float o[3] = {dx * ca + dy * (-sa), dx * sa + dy * ca, dz};
    float prod = 1.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float m = g[3 + k] * 0.5f;
        const float d0 = m - o[k], d1 = m + o[k];          // margin - off, -(-margin - off)
        prod *= fminf(d0, d1) / fmaxf(d0, d1);
    }
    out[p] = labels[p] > 0 ? powf(fmaxf(prod, 1e-6f), 1.f / 3.f) : 0.f;
}

// out = {total, xyzwhl, ori_bin * dir_weight, ori_res}; grad (n, 6 + 2 nb)
__global__ __launch_bounds__(HL_THREADS) void head_box_loss_kernel(const float* __restrict__ preds, const float* __restrict__ labels,
                                                                   const int64_t* __restrict__ cls_labels, const float* __restrict__ code_w,
                                                                   float beta, int nb, float dir_w, float box_w, int64_t n,
                                                                   float* __restrict__ out, float* __restrict__ grad) {
    __shared__ double red[17];
    const int W = 6 + 2 * nb;
    double cnt = 0, sres = 0;
    for (int64_t r = threadIdx.x; r < n; r += HL_THREADS) {
        cnt += cls_labels[r] > 0 ? 1.0 : 0.0;
        // smooth_l1(res, lab_res) with reduction 'mean' over ALL rows (beta 1), as the reference writes it (:1268-1269)
        const int id = (int)labels[r * 8 + 6];
        float dd;
        sres += hl_smooth_l1(preds[r * W + 6 + nb + id] - labels[r * 8 + 7], 1.f, dd);
    }
    const double npos = hl_block_sum(cnt, red);
    const double res_mean = hl_block_sum(sres, red) / (double)(n > 0 ? n : 1);
    const float wpos = 1.f / (float)(npos > 1.0 ? npos : 1.0);
    const float sumw = npos > 0.0 ? 1.f : 0.f;               // sum of the weights = npos / max(npos, 1)
    double a_xyz = 0, a_cls = 0;
    for (int64_t r = threadIdx.x; r < n; r += HL_THREADS) {
        const float w = cls_labels[r] > 0 ? wpos : 0.f;
        const float* x = preds + r * W;
        const float* l = labels + r * 8;
        float* g = grad + r * W;
        float rl = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float cw = code_w ? code_w[k] : 1.f;
            // NaN targets follow the input (loss_utils.py:166).  Bit test: -fno-honor-nans lets the compiler fold l != l
            const float t = (__float_as_uint(l[k]) & 0x7fffffffu) > 0x7f800000u ? x[k] : l[k];
            float dd;
            rl += hl_smooth_l1((x[k] - t) * cw, beta, dd);
            g[k] = box_w * w * dd * cw;
        }
        a_xyz += (double)(w * rl);
        // cross entropy over the nb heading bins
        const int id = (int)l[6];
        float mx = x[6];
        for (int k = 1; k < nb; ++k) mx = fmaxf(mx, x[6 + k]);
        float se = 0.f;
        for (int k = 0; k < nb; ++k) se += expf(x[6 + k] - mx);
        const float lse = mx + logf(se);
        a_cls += (double)(w * (lse - x[6 + id]));
        for (int k = 0; k < nb; ++k) g[6 + k] = box_w * dir_w * w * (expf(x[6 + k] - lse) - (k == id ? 1.f : 0.f));
        // residual of the labelled bin: d(mean over all rows) * sum(w)
        float dd;
        hl_smooth_l1(x[6 + nb + id] - l[7], 1.f, dd);
        for (int k = 0; k < nb; ++k) g[6 + nb + k] = k == id ? box_w * sumw * dd / (float)n : 0.f;
    }
    const double xyz = hl_block_sum(a_xyz, red);
    const double cls = hl_block_sum(a_cls, red) * dir_w;
    if (threadIdx.x == 0) {
        const double res = res_mean * sumw;
        out[0] = (float)((xyz + res + cls) * box_w); out[1] = (float)xyz; out[2] = (float)cls; out[3] = (float)res;
    }
}

// mode 0 (LOSS_VOTE_TYPE none): key = class label of the point; loss = weight * mean over the classes present of
//   sum_{label == c} sum_xyz smooth_l1(vote - centre) / (3 n_c).
// mode 1 (ver2): key = box index of the point (-1: none), segment = scene * S + key; per instance
//   [sum sl1(vote, box centre) + 0.5 sum sl1(vote, mean vote of the instance)] / #points, mean over the instances present.
// origin, offsets: (n, 4) with the coordinates in columns 1..3; grad (n, 4), column 0 zero.
constexpr int HL_MAX_SEG = 2048;
__global__ __launch_bounds__(HL_THREADS) void head_vote_loss_kernel(int mode, const float* __restrict__ origin, const float* __restrict__ offsets,
                                                                    const int64_t* __restrict__ key, const float* __restrict__ gt, int B,
                                                                    int S, int num_class, float weight, int64_t n, float* __restrict__ out,
                                                                    float* __restrict__ grad) {
    __shared__ double red[17];
    __shared__ float cnt[HL_MAX_SEG], m3[HL_MAX_SEG][3], a3[HL_MAX_SEG][3], lsum[HL_MAX_SEG];
    const int nseg = mode == 0 ? num_class + 1 : B * S;
    for (int s = threadIdx.x; s < nseg; s += HL_THREADS) {
        cnt[s] = 0.f; lsum[s] = 0.f;
        for (int k = 0; k < 3; ++k) { m3[s][k] = 0.f; a3[s][k] = 0.f; }
    }
    __syncthreads();
    const int64_t per_scene = n / (B > 0 ? B : 1);
    auto seg_of = [&](int64_t r) -> int {
        const int64_t k = key[r];
        if (mode == 0) return (k >= 1 && k <= num_class) ? (int)k : -1;
        return k >= 0 ? (int)(r / per_scene) * S + (int)k : -1;
    };
    for (int64_t r = threadIdx.x; r < n; r += HL_THREADS) {
        const int s = seg_of(r);
        if (s < 0) continue;
        atomicAdd(&cnt[s], 1.f);
        if (mode == 1)
            for (int k = 0; k < 3; ++k) atomicAdd(&m3[s][k], origin[r * 4 + 1 + k] + offsets[r * 4 + 1 + k]);
    }
    __syncthreads();
    if (mode == 1)
        for (int s = threadIdx.x; s < nseg; s += HL_THREADS)
            for (int k = 0; k < 3; ++k) m3[s][k] = m3[s][k] / fmaxf(cnt[s], 1.f);
    __syncthreads();
    double present = 0;
    for (int s = threadIdx.x; s < nseg; s += HL_THREADS) present += cnt[s] > 0.f ? 1.0 : 0.0;
    const double npresent = hl_block_sum(present, red);
    const float inv_present = 1.f / (float)(npresent > 1.0 ? npresent : 1.0);
    // pass 2: per-point loss into its segment, and (ver2) the sums of the mean-term derivatives per segment
    for (int64_t r = threadIdx.x; r < n; r += HL_THREADS) {
        const int s = seg_of(r);
        if (s < 0) continue;
        float l = 0.f;
        for (int k = 0; k < 3; ++k) {
            const float p = origin[r * 4 + 1 + k] + offsets[r * 4 + 1 + k];
            float dd;
            l += hl_smooth_l1(p - gt[r * 8 + k], 1.f, dd);
            if (mode == 1) {
                float dm;
                l += 0.5f * hl_smooth_l1(p - m3[s][k], 1.f, dm);
                atomicAdd(&a3[s][k], dm);
            }
        }
        atomicAdd(&lsum[s], l);
    }
    __syncthreads();
    // loss
    double acc = 0;
    for (int s = threadIdx.x; s < nseg; s += HL_THREADS)
        if (cnt[s] > 0.f) acc += (double)lsum[s] / (double)(cnt[s] * (mode == 0 ? 3.f : 1.f));
    const double total = hl_block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = mode == 0 ? (float)(total / npresent * weight) : (float)(total * inv_present * weight);
    // gradient wrt the offsets (the votes are origin + offset)
    for (int64_t r = threadIdx.x; r < n; r += HL_THREADS) {
        const int s = seg_of(r);
        float* g = grad + r * 4;
        g[0] = 0.f;
        if (s < 0) { g[1] = g[2] = g[3] = 0.f; continue; }
        const float base = weight / (mode == 0 ? (float)npresent * 3.f * cnt[s] : cnt[s]) * (mode == 0 ? 1.f : inv_present);
        for (int k = 0; k < 3; ++k) {
            const float p = origin[r * 4 + 1 + k] + offsets[r * 4 + 1 + k];
            float dd, dm = 0.f;
            hl_smooth_l1(p - gt[r * 8 + k], 1.f, dd);
            float d = dd;
            if (mode == 1) {
                hl_smooth_l1(p - m3[s][k], 1.f, dm);
                d += 0.5f * (dm - a3[s][k] / cnt[s]);          // the mean moves with every vote of the instance
            }
            g[1 + k] = base * d;
        }
    }
}

// Corner loss of the positive centres: decode the predicted box (mean-size anchors of the PREDICTED class, heading = arg-max
// bin + its residual), 8 corners against the box's and its heading-flipped twin's, smooth-L1 (beta 1) of the smaller corner
// distance, mean over the corners, mean over the positives (0 / 0 = NaN without positives, as in the reference).
// grad_box (n, 6 + 2 nb), grad_ctr (n, 4) (the decoded centre is the vote centre plus a residual).
__global__ __launch_bounds__(HL_THREADS) void head_corner_loss_kernel(const float* __restrict__ box_preds, const float* __restrict__ centers,
                                                                      const float* __restrict__ cls_preds, int C, const float* __restrict__ gt,
                                                                      const int64_t* __restrict__ cls_labels, const float* __restrict__ mean_size,
                                                                      int nb, float weight, int64_t n, float* __restrict__ out,
                                                                      float* __restrict__ grad_box, float* __restrict__ grad_ctr) {
    __shared__ double red[17];
    const int W = 6 + 2 * nb;
    const float PI = 3.14159265358979323846f;
    const float inter = 2.f * PI / (float)nb;
    double cnt = 0;
    for (int64_t r = threadIdx.x; r < n; r += HL_THREADS) cnt += cls_labels[r] > 0 ? 1.0 : 0.0;
    const double npos = hl_block_sum(cnt, red);
    const float wpos = weight / (float)npos;                   // inf without positives: never multiplied by a positive row
    double acc = 0;
    const float tmpl[8][3] = {{1, 1, -1}, {1, -1, -1}, {-1, -1, -1}, {-1, 1, -1}, {1, 1, 1}, {1, -1, 1}, {-1, -1, 1}, {-1, 1, 1}};
    for (int64_t r = threadIdx.x; r < n; r += HL_THREADS) {
        float* gb = grad_box + r * W;
        float* gc = grad_ctr + r * 4;
        for (int k = 0; k < W; ++k) gb[k] = 0.f;
        gc[0] = gc[1] = gc[2] = gc[3] = 0.f;
        if (!(cls_labels[r] > 0)) continue;
        const float* e = box_preds + r * W;
        // predicted class -> anchor
        int pc = 0;
        for (int c = 1; c < C; ++c) pc = cls_preds[r * C + c] > cls_preds[r * C + pc] ? c : pc;
        float dxa = 1.f, dya = 1.f, dza = 1.f, diag = 1.f;
        if (mean_size) { dxa = mean_size[pc * 3]; dya = mean_size[pc * 3 + 1]; dza = mean_size[pc * 3 + 2]; diag = sqrtf(dxa * dxa + dya * dya); }
        const float xa = centers[r * 4 + 1], ya = centers[r * 4 + 2], za = centers[r * 4 + 3];
        const float xg = mean_size ? e[0] * diag + xa : e[0] + xa, yg = mean_size ? e[1] * diag + ya : e[1] + ya,
                    zg = mean_size ? e[2] * dza + za : e[2] + za;
        const float dxg = expf(e[3]) * dxa, dyg = expf(e[4]) * dya, dzg = expf(e[5]) * dza;
        int bin = 0;
        for (int k = 1; k < nb; ++k) bin = e[6 + k] > e[6 + bin] ? k : bin;
        const float ry = ((float)bin * inter - PI + inter * 0.5f) + e[6 + nb + bin] * (inter * 0.5f);
        const float* g = gt + r * 8;
        const float cp = cosf(ry), sp = sinf(ry), cg = cosf(g[6]), sg = sinf(g[6]), cf = cosf(g[6] + PI), sf = sinf(g[6] + PI);
        float d_x = 0.f, d_y = 0.f, d_z = 0.f, d_dx = 0.f, d_dy = 0.f, d_dz = 0.f, d_ry = 0.f, lsum = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float tx = tmpl[k][0] * 0.5f, ty = tmpl[k][1] * 0.5f, tz = tmpl[k][2] * 0.5f;
            const float px = tx * dxg, py = ty * dyg;
            const float rx = px * cp - py * sp, rry = px * sp + py * cp;          // (px, py) @ [[c, s], [-s, c]]
            const float pcx = rx + xg, pcy = rry + yg, pcz = tz * dzg + zg;
            const float qx = tx * g[3], qy = ty * g[4];
            const float gx = qx * cg - qy * sg + g[0], gy = qx * sg + qy * cg + g[1], gz = tz * g[5] + g[2];
            const float fx = qx * cf - qy * sf + g[0], fy = qx * sf + qy * cf + g[1];
            const float ax = pcx - gx, ay = pcy - gy, az = pcz - gz;
            const float bx = pcx - fx, by = pcy - fy;
            const float d0 = sqrtf(ax * ax + ay * ay + az * az), d1 = sqrtf(bx * bx + by * by + az * az);
            const bool first = d0 <= d1;                       // torch.min: ties give the gradient to ... either; distances equal
            const float d = first ? d0 : d1;
            float dd;
            lsum += hl_smooth_l1(d, 1.f, dd);
            const float inv = d > 0.f ? dd / d : 0.f;
            const float ux = (first ? ax : bx) * inv, uy = (first ? ay : by) * inv, uz = az * inv;     // d(smooth-l1)/d(corner)
            d_x += ux; d_y += uy; d_z += uz;
            d_dx += ux * tx * cp + uy * tx * sp;
            d_dy += -ux * ty * sp + uy * ty * cp;
            d_dz += uz * tz;
            d_ry += -ux * rry + uy * rx;
        }
        acc += (double)(lsum * 0.125f);
        const float s = wpos * 0.125f;
        gc[1] = s * d_x; gc[2] = s * d_y; gc[3] = s * d_z;
        gb[0] = s * d_x * (mean_size ? diag : 1.f); gb[1] = s * d_y * (mean_size ? diag : 1.f); gb[2] = s * d_z * (mean_size ? dza : 1.f);
        gb[3] = s * d_dx * dxg; gb[4] = s * d_dy * dyg; gb[5] = s * d_dz * dzg;
        gb[6 + nb + bin] = s * d_ry * (inter * 0.5f);
    }
    const double total = hl_block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = (float)(total / npos) * weight;
}

}  // namespace pda

PDA_API int pda_head_cls_loss(const float* preds, int row_stride, int col0, int num_class, const int64_t* labels, const float* soft,
                              int64_t n, float scale, float* out2, float* grad, pda_stream_t stream) {
    PDA_REQUIRE(n >= 0 && num_class >= 1 && col0 >= 0 && col0 + num_class <= row_stride, "pda_head_cls_loss: bad shape");
    PDA_REQUIRE(preds && labels && out2 && grad, "pda_head_cls_loss: null pointer");
    hipLaunchKernelGGL(pda::head_cls_loss_kernel, dim3(1), dim3(pda::HL_THREADS), 0, (hipStream_t)stream, preds, row_stride, col0, num_class,
                       labels, soft, n, scale, out2, grad);
    return pda::check_launch("pda_head_cls_loss");
}

PDA_API int pda_head_centerness(const float* centers, const float* gt, const int64_t* labels, float* out, int64_t n, pda_stream_t stream) {
    PDA_REQUIRE(n >= 0, "pda_head_centerness: n=%lld", (long long)n);
    if (n == 0) return PDA_OK;
    PDA_REQUIRE(centers && gt && labels && out, "pda_head_centerness: null pointer");
    hipLaunchKernelGGL(pda::head_centerness_kernel, dim3((unsigned)pda::divup64(n, 256)), dim3(256), 0, (hipStream_t)stream, centers, gt, labels,
                       out, n);
    return pda::check_launch("pda_head_centerness");
}

PDA_API int pda_head_box_loss(const float* preds, const float* labels, const int64_t* cls_labels, const float* code_weights, float beta,
                              int bins, float dir_weight, float box_weight, int64_t n, float* out4, float* grad, pda_stream_t stream) {
    PDA_REQUIRE(n >= 1 && bins >= 1 && bins <= 64, "pda_head_box_loss: n=%lld bins=%d", (long long)n, bins);
    PDA_REQUIRE(preds && labels && cls_labels && out4 && grad, "pda_head_box_loss: null pointer");
    hipLaunchKernelGGL(pda::head_box_loss_kernel, dim3(1), dim3(pda::HL_THREADS), 0, (hipStream_t)stream, preds, labels, cls_labels,
                       code_weights, beta, bins, dir_weight, box_weight, n, out4, grad);
    return pda::check_launch("pda_head_box_loss");
}

PDA_API int pda_head_vote_loss(int mode, const float* origin, const float* offsets, const int64_t* key, const float* gt, int b, int boxes,
                               int num_class, float weight, int64_t n, float* out1, float* grad, pda_stream_t stream) {
    PDA_REQUIRE(n >= 1 && b >= 1 && n % b == 0 && (mode == 0 || mode == 1), "pda_head_vote_loss: n=%lld b=%d mode=%d", (long long)n, b, mode);
    if ((mode == 0 ? num_class + 1 : b * boxes) > pda::HL_MAX_SEG) {
        pda::set_error("pda_head_vote_loss: %d segments exceed %d", mode == 0 ? num_class + 1 : b * boxes, pda::HL_MAX_SEG);
        return PDA_ERR_UNSUPPORTED;
    }
    PDA_REQUIRE(origin && offsets && key && gt && out1 && grad, "pda_head_vote_loss: null pointer");
    hipLaunchKernelGGL(pda::head_vote_loss_kernel, dim3(1), dim3(pda::HL_THREADS), 0, (hipStream_t)stream, mode, origin, offsets, key, gt, b,
                       boxes, num_class, weight, n, out1, grad);
    return pda::check_launch("pda_head_vote_loss");
}

PDA_API int pda_head_corner_loss(const float* box_preds, const float* centers, const float* cls_preds, int num_class, const float* gt,
                                 const int64_t* cls_labels, const float* mean_size, int bins, float weight, int64_t n, float* out1,
                                 float* grad_box, float* grad_centers, pda_stream_t stream) {
    PDA_REQUIRE(n >= 1 && bins >= 1 && bins <= 64 && num_class >= 1, "pda_head_corner_loss: n=%lld bins=%d", (long long)n, bins);
    PDA_REQUIRE(box_preds && centers && cls_preds && gt && cls_labels && out1 && grad_box && grad_centers, "pda_head_corner_loss: null pointer");
    hipLaunchKernelGGL(pda::head_corner_loss_kernel, dim3(1), dim3(pda::HL_THREADS), 0, (hipStream_t)stream, box_preds, centers, cls_preds,
                       num_class, gt, cls_labels, mean_size, bins, weight, n, out1, grad_box, grad_centers);
    return pda::check_launch("pda_head_corner_loss");
}
