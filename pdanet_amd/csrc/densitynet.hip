// densitynet.hip -- DensityNet (pointnet2_modules.py:958-981) in training mode as 4 + 5 launches
// (include/pda_train.h): y = relu(bn3(w3 . relu(bn2(W2 relu(bn1(w1 x + b1)) + b2)) + b3)) for a SCALAR input x per
// (group, neighbour) token, hidden widths 16 and 8, batch statistics over all tokens.
//
// Through torch this is 3 x (K<=16 GEMM + bias, BatchNorm statistics / finalise / apply) forward and about 30
// more launches backward per scale, each a few microseconds of latency-bound work on a (tokens, <=16) tensor.
// The input is 4 bytes per token, so nothing is materialised here: every pass re-reads x and recomputes the
// earlier layers in registers.  Forward: P1 sum x, x^2 (layer 1 is affine in x, its batch statistics follow
// from mean / variance of x); P2 statistics of layer 2's pre-activation; P3 of layer 3's; P4 writes y.
// Backward (the input has no gradient: it is a function of coordinates only): B1..B4 walk the layers from the
// last to the first, each pass producing the BatchNorm reduction sums the next one needs plus the weight
// gradients that have become computable; B5 adds up the last partials.  Per-block partial sums in double,
// added in fixed order by every block of the following pass (deterministic, no extra finalise launches).
#include "pda_common.h"

namespace pda {

constexpr int DN_H1 = 16, DN_H2 = 8;
constexpr int DN_BLOCKS = 128;   // few, fat blocks: every pass starts by adding up the previous pass's per-block partials
// parameter block (floats): w1[16] b1[16] g1[16] be1[16] W2[8][16] b2[8] g2[8] be2[8] w3[8] b3 g3 be3
constexpr int DN_W1 = 0, DN_B1 = 16, DN_G1 = 32, DN_BE1 = 48, DN_W2 = 64, DN_B2 = 192, DN_G2 = 200, DN_BE2 = 208,
              DN_W3 = 216, DN_B3 = 224, DN_G3 = 225, DN_BE3 = 226, DN_NPARAM = 227;
// statistics block (floats, written by the passes): mx, varx, a1[16] (= w1 g1 / sqrt(w1^2 varx + eps)),
// mean2[8] inv2[8] mean3 inv3
// (means are kept as hi + lo float pairs: a mean rounded to float shifts every normalised value by the same
// ~1e-8, and the backward sums, which cancel to O(eps), pick up n times that)
constexpr int DN_MX = 0, DN_VX = 1, DN_A1 = 2, DN_M2 = 18, DN_I2 = 26, DN_M3 = 34, DN_I3 = 35, DN_MXL = 36, DN_M2L = 37, DN_M3L = 45,
              DN_NSTAT = 46;
// widest partial record of a pass (B3): dW2[128] db2[8] s1[16] s1x[16]
constexpr int DN_MAXP = 168;

struct DnState {
    float p[DN_NPARAM];
};

// Which tokens a pass walks.  Dense (rowmap == nullptr): token t of n, every one on its own.  Unique (csrc/ragged.hip):
// x / y / dy keep their dense (groups, ns) layout, but ball_query's repeats of a group's first neighbour (slots cnt..ns-1)
// hold the same x as slot 0 and are not evaluated: the passes walk the *n_unique distinct slots rowmap[0..), slot 0
// standing for roww = ns - cnt + 1 tokens.  Statistics are those of all n dense tokens (weighted sums); forward writes a
// group's repeat slots along with its slot 0; backward takes a distinct slot's gradient as the SUM over its copies (every
// BatchNorm-backward term that is per dense token, i.e. the two batch means, enters w times).  Results equal the dense
// passes' up to the order of the float additions.
struct DnRows {
    const int32_t* rowmap;
    const float* roww;
    const int32_t* n_unique;   // device memory (off[groups] of the plan): no host read
    int ns;
};
__device__ __forceinline__ int64_t dn_rows(const DnRows& R, int64_t n) { return R.rowmap ? (int64_t)*R.n_unique : n; }

// block-wide sum of `cnt` per-thread doubles -> partial[blockIdx][k]
// (a thread sees at most n / 65536 + 1 tokens, so its own running sums stay in float; everything across
// lanes, waves and blocks is added in double)
template <int CNT>
__device__ __forceinline__ void dn_block_partials(const float (&v)[CNT], double* __restrict__ partial) {
    __shared__ double red[4][CNT];
    const int lane = lane_id(), w = wave_id();
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
        double a = (double)v[k];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0) red[w][k] = a;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < CNT; k += 256) partial[(size_t)blockIdx.x * DN_MAXP + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
}

// sums of up to three earlier passes at once: thread k < c1 + c2 + c3 owns one value (fixed order over the blocks)
__device__ __forceinline__ void dn_sum_partials3(const double* __restrict__ p1, int c1, double* o1, const double* __restrict__ p2, int c2,
                                                 double* o2, const double* __restrict__ p3, int c3, double* o3, int nblocks) {
    __syncthreads();
    int k = threadIdx.x;
    const double* src = nullptr;
    double* dst = nullptr;
    if (k < c1) { src = p1; dst = o1; }
    else if ((k -= c1) < c2) { src = p2; dst = o2; }
    else if ((k -= c2) < c3) { src = p3; dst = o3; }
    if (src) {
        double a = 0;
        for (int b0 = 0; b0 < nblocks; b0 += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = b0 + u < nblocks ? src[(size_t)(b0 + u) * DN_MAXP + k] : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) a += v[u];
        }
        dst[k] = a;
    }
    __syncthreads();
}

// sum over the blocks of the previous pass, fixed order; result in LDS `out[0..cnt)` for all threads
__device__ __forceinline__ void dn_sum_partials(const double* __restrict__ partial, int nblocks, int cnt, double* out) {
    __syncthreads();
    for (int k = threadIdx.x; k < cnt; k += 256) {
        double a = 0;
        for (int b0 = 0; b0 < nblocks; b0 += 16) {       // 16 independent loads per round (a serial chain of
            double v[16];                                 // nblocks dependent L2 round trips costs ~0.5 us each)
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = b0 + u < nblocks ? partial[(size_t)(b0 + u) * DN_MAXP + k] : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) a += v[u];
        }
        out[k] = a;
    }
    __syncthreads();
}

struct DnFwd {  // forward values of one token
    float xh;                 // x - mean_x
    float h1[DN_H1];
    float z2h[DN_H2];         // normalised layer-2 pre-activation
    float h2[DN_H2];
    float z3h, y;
};

__device__ __forceinline__ void dn_layer1(const float* __restrict__ prm, const float* __restrict__ st, float x, DnFwd& f) {
    f.xh = (x - st[DN_MX]) - st[DN_MXL];
#pragma unroll
    for (int c = 0; c < DN_H1; ++c) f.h1[c] = fmaxf(st[DN_A1 + c] * f.xh + prm[DN_BE1 + c], 0.f);
}
__device__ __forceinline__ void dn_z2(const float* __restrict__ prm, const DnFwd& f, float (&z2)[DN_H2]) {
#pragma unroll
    for (int j = 0; j < DN_H2; ++j) {
        float a = prm[DN_B2 + j];
#pragma unroll
        for (int c = 0; c < DN_H1; ++c) a += prm[DN_W2 + j * DN_H1 + c] * f.h1[c];
        z2[j] = a;
    }
}
__device__ __forceinline__ float dn_layer2(const float* __restrict__ prm, const float* __restrict__ st, DnFwd& f) {
    float z2[DN_H2];
    dn_z2(prm, f, z2);
    float z3 = prm[DN_B3];
#pragma unroll
    for (int j = 0; j < DN_H2; ++j) {
        f.z2h[j] = ((z2[j] - st[DN_M2 + j]) - st[DN_M2L + j]) * st[DN_I2 + j];
        f.h2[j] = fmaxf(f.z2h[j] * prm[DN_G2 + j] + prm[DN_BE2 + j], 0.f);
        z3 += prm[DN_W3 + j] * f.h2[j];
    }
    return z3;
}

__device__ __forceinline__ void dn_running(float* rm, float* rv, int c, double mean, double var, int64_t n, float momentum) {
    if (!rm) return;
    const double unbiased = n > 1 ? var * ((double)n / (double)(n - 1)) : var;
    rm[c] = (float)((1.0 - momentum) * rm[c] + momentum * mean);
    rv[c] = (float)((1.0 - momentum) * rv[c] + momentum * unbiased);
}

// PASS: 1 = sum x, x^2; 2 = layer-2 statistics; 3 = layer-3 statistics; 4 = output
template <int PASS>
__device__ __forceinline__ void densitynet_fwd_pass(const float* __restrict__ x, const float* __restrict__ prm_g,
                                                             float* __restrict__ stats, double* __restrict__ part_in,
                                                             double* __restrict__ part_out, float* __restrict__ y, int64_t n,
                                                             float eps, float momentum, float* rm1, float* rv1, float* rm2,
                                                             float* rv2, float* rm3, float* rv3, int nblocks, DnRows R) {
    __shared__ float prm[DN_NPARAM];
    __shared__ float st[DN_NSTAT];
    __shared__ double sums[DN_MAXP];
    for (int k = threadIdx.x; k < DN_NPARAM; k += 256) prm[k] = prm_g[k];
    if (PASS >= 3) for (int k = threadIdx.x; k < DN_NSTAT; k += 256) st[k] = stats[k];   // earlier passes' results
    __syncthreads();
    // finalise the previous pass's reduction (every block redundantly, block 0 publishes)
    if (PASS == 2) {
        dn_sum_partials(part_in, nblocks, 2, sums);
        const double mx = sums[0] / (double)n;
        double vx = sums[1] / (double)n - mx * mx;
        vx = vx < 0 ? 0 : vx;
        if (threadIdx.x < DN_H1) {
            const int c = threadIdx.x;
            const double w = prm[DN_W1 + c];
            const double var1 = w * w * vx;
            st[DN_A1 + c] = (float)(w * (double)prm[DN_G1 + c] / sqrt(var1 + (double)eps));
            if (blockIdx.x == 0) dn_running(rm1, rv1, c, w * mx + (double)prm[DN_B1 + c], var1, n, momentum);
        }
        if (threadIdx.x == 0) { st[DN_MX] = (float)mx; st[DN_MXL] = (float)(mx - (double)(float)mx); st[DN_VX] = (float)vx; }
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x < DN_A1 + DN_H1) stats[threadIdx.x] = st[threadIdx.x];
        if (blockIdx.x == 0 && threadIdx.x == 0) stats[DN_MXL] = st[DN_MXL];
    } else if (PASS == 3) {
        dn_sum_partials(part_in, nblocks, 2 * DN_H2, sums);
        if (threadIdx.x < DN_H2) {
            const int j = threadIdx.x;
            const double m = sums[j] / (double)n;
            double v = sums[DN_H2 + j] / (double)n - m * m;
            v = v < 0 ? 0 : v;
            st[DN_M2 + j] = (float)m;
            st[DN_M2L + j] = (float)(m - (double)(float)m);
            st[DN_I2 + j] = (float)(1.0 / sqrt(v + (double)eps));
            if (blockIdx.x == 0) {
                dn_running(rm2, rv2, j, m, v, n, momentum);
                stats[DN_M2 + j] = st[DN_M2 + j]; stats[DN_M2L + j] = st[DN_M2L + j]; stats[DN_I2 + j] = st[DN_I2 + j];
            }
        }
        __syncthreads();
    } else if (PASS == 4) {
        dn_sum_partials(part_in, nblocks, 2, sums);
        if (threadIdx.x == 0) {
            const double m = sums[0] / (double)n;
            double v = sums[1] / (double)n - m * m;
            v = v < 0 ? 0 : v;
            st[DN_M3] = (float)m;
            st[DN_M3L] = (float)(m - (double)(float)m);
            st[DN_I3] = (float)(1.0 / sqrt(v + (double)eps));
            if (blockIdx.x == 0) { dn_running(rm3, rv3, 0, m, v, n, momentum); stats[DN_M3] = st[DN_M3]; stats[DN_M3L] = st[DN_M3L]; stats[DN_I3] = st[DN_I3]; }
        }
        __syncthreads();
    }
    constexpr int CNT = PASS == 2 ? 2 * DN_H2 : 2;
    float acc[CNT];
#pragma unroll
    for (int k = 0; k < CNT; ++k) acc[k] = 0;
    const int64_t rows = dn_rows(R, n);
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < rows; t += (int64_t)nblocks * 256) {
        const int64_t e = R.rowmap ? R.rowmap[t] : t;
        const float w = R.rowmap ? R.roww[t] : 1.f;
        const float xv = x[e];
        if (PASS == 1) {
            acc[0] += w * xv; acc[1] += w * (xv * xv);
        } else {
            DnFwd f;
            dn_layer1(prm, st, xv, f);
            if (PASS == 2) {
                float z2[DN_H2];
                dn_z2(prm, f, z2);
#pragma unroll
                for (int j = 0; j < DN_H2; ++j) { acc[j] += w * z2[j]; acc[DN_H2 + j] += w * (z2[j] * z2[j]); }
            } else {
                const float z3 = dn_layer2(prm, st, f);
                if (PASS == 3) { acc[0] += w * z3; acc[1] += w * (z3 * z3); }
                else {
                    const float yv = fmaxf(((z3 - st[DN_M3]) - st[DN_M3L]) * st[DN_I3] * prm[DN_G3] + prm[DN_BE3], 0.f);
                    y[e] = yv;
                    const int rep = (int)w - 1;                       // slot 0 of a short list: its repeats sit at the group's end
                    for (int q = 0; q < rep; ++q) y[e + R.ns - rep + q] = yv;
                }
            }
        }
    }
    if (PASS != 4) dn_block_partials<CNT>(acc, part_out);
}

// Backward passes.  Partial layouts (doubles):
//  B1 out: [0] sum dyh3, [1] sum dyh3*z3h
//  B2 out: dW3[8] @0, db3 @8, s2[8] @9 (sum dyh2), s2z[8] @17 (sum dyh2*z2h)
//  B3 out: dW2[128] @0, db2[8] @128, s1[16] @136, s1x[16] @152 (sum dyh1 * x1h with x1h = normalised layer-1 input)
//  B4 out: dW1[16] @0, db1[16] @16
template <int PASS>
__device__ __forceinline__ void densitynet_bwd_pass(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ prm_g, const float* __restrict__ stats,
                                                             const double* __restrict__ p1, const double* __restrict__ p2,
                                                             const double* __restrict__ p3, double* __restrict__ part_out,
                                                             int64_t n, float eps, int nblocks, DnRows R) {
    __shared__ float prm[DN_NPARAM];
    __shared__ float st[DN_NSTAT];
    __shared__ double sums[DN_MAXP];
    // the BatchNorm-backward means stay in double: rounded to float their error enters every token's dz with the
    // same sign and the weight-gradient sums (ill-conditioned: BN makes them nearly cancel) inherit n times that
    __shared__ double m3[2], m2[2 * DN_H2], m1[2 * DN_H1];
    __shared__ float inv1[DN_H1];
    for (int k = threadIdx.x; k < DN_NPARAM; k += 256) prm[k] = prm_g[k];
    for (int k = threadIdx.x; k < DN_NSTAT; k += 256) st[k] = stats[k];
    __syncthreads();
    {
        __shared__ double s1[2], s2[25];
        dn_sum_partials3(p1, PASS >= 2 ? 2 : 0, s1, p2, PASS >= 3 ? 25 : 0, s2, p3, PASS >= 4 ? DN_MAXP : 0, sums, nblocks);
        if (PASS >= 2 && threadIdx.x < 2) m3[threadIdx.x] = s1[threadIdx.x] / (double)n;
        if (PASS >= 3 && threadIdx.x < 2 * DN_H2) m2[threadIdx.x] = s2[9 + threadIdx.x] / (double)n;
        if (PASS >= 4 && threadIdx.x < 2 * DN_H1) m1[threadIdx.x] = sums[136 + threadIdx.x] / (double)n;
    }
    if (threadIdx.x < DN_H1) {
        // 1 / sqrt(var1 + eps) of layer 1 = a1 / (w1 g1) is ill-conditioned for tiny weights: recompute from var_x
        const double w = prm[DN_W1 + threadIdx.x];
        inv1[threadIdx.x] = (float)(1.0 / sqrt(w * w * (double)st[DN_VX] + (double)eps));
    }
    __syncthreads();
    const int64_t rows = dn_rows(R, n);
    // token t of the walk: its dense slot, multiplicity, and the gradient of all its copies
    auto operands = [&](int64_t t, int64_t& e, float& w, float& g) {
        e = R.rowmap ? R.rowmap[t] : t;
        w = R.rowmap ? R.roww[t] : 1.f;
        g = dy[e];
        const int rep = (int)w - 1;
        for (int q = 0; q < rep; ++q) g += dy[e + R.ns - rep + q];
    };
    if (PASS == 3) {
        // dW2 is an 8 x 16 outer-product sum: 128 running sums per token-owning thread would not fit in registers.
        // Tokens are staged 256 at a time in LDS (h1[16], dz2[8], dyh1[16], xh) and thread k < 168 owns ONE of the
        // 168 sums of this pass, walking the staged tokens (LDS broadcast reads).
        __shared__ float stage[256][DN_H1 + DN_H2 + DN_H1 + 1];
        double mine = 0;
        const int k = threadIdx.x;
        for (int64_t t0 = (int64_t)blockIdx.x * 256; t0 < rows; t0 += (int64_t)nblocks * 256) {
            const int64_t t = t0 + threadIdx.x;
            float* sp = stage[threadIdx.x];
            if (t < rows) {
                int64_t e;
                float w, gy;
                operands(t, e, w, gy);
                const double wd = (double)w;
                DnFwd f;
                dn_layer1(prm, st, x[e], f);
                const float z3 = dn_layer2(prm, st, f);
                f.z3h = ((z3 - st[DN_M3]) - st[DN_M3L]) * st[DN_I3];
                const float dyh3 = (f.z3h * prm[DN_G3] + prm[DN_BE3]) > 0.f ? gy : 0.f;
                const float dz3 = prm[DN_G3] * st[DN_I3] * (float)((double)dyh3 - wd * m3[0] - wd * ((double)f.z3h * m3[1]));
                float dz2[DN_H2];
#pragma unroll
                for (int j = 0; j < DN_H2; ++j) {
                    const float dyh2 = f.h2[j] > 0.f ? prm[DN_W3 + j] * dz3 : 0.f;
                    dz2[j] = prm[DN_G2 + j] * st[DN_I2 + j] * (float)((double)dyh2 - wd * m2[j] - wd * ((double)f.z2h[j] * m2[DN_H2 + j]));
                    sp[DN_H1 + j] = dz2[j];
                }
#pragma unroll
                for (int c = 0; c < DN_H1; ++c) {
                    float a = 0.f;
#pragma unroll
                    for (int j = 0; j < DN_H2; ++j) a += prm[DN_W2 + j * DN_H1 + c] * dz2[j];
                    sp[c] = f.h1[c];
                    sp[DN_H1 + DN_H2 + c] = f.h1[c] > 0.f ? a : 0.f;
                }
                sp[DN_H1 + DN_H2 + DN_H1] = f.xh;
            } else {
#pragma unroll
                for (int q = 0; q < DN_H1 + DN_H2 + DN_H1 + 1; ++q) sp[q] = 0.f;
            }
            __syncthreads();
            if (k < DN_MAXP) {
                // double: these sums become the BatchNorm-backward means of the next pass (see m1/m2/m3 above)
                // 8 tokens per step with independent LDS reads (a dependent chain of 256 LDS round trips otherwise)
                double a = 0;
                int ia, ib;            // a token contributes stage[q][ia] * (ib >= 0 ? stage[q][ib] : 1)
                float sc = 1.f;
                if (k < 128) { ia = DN_H1 + (k >> 4); ib = k & 15; }
                else if (k < 136) { ia = DN_H1 + (k - 128); ib = -1; }
                else if (k < 152) { ia = DN_H1 + DN_H2 + (k - 136); ib = -1; }
                else { const int c = k - 152; ia = DN_H1 + DN_H2 + c; ib = DN_H1 + DN_H2 + DN_H1; sc = prm[DN_W1 + c] * inv1[c]; }
                for (int q0 = 0; q0 < 256; q0 += 8) {
                    float va[8], vb[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { va[u] = stage[q0 + u][ia]; vb[u] = ib >= 0 ? stage[q0 + u][ib] : 1.f; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) a += (double)(va[u] * (sc * vb[u]));
                }
                mine += a;
            }
            __syncthreads();
        }
        if (k < DN_MAXP) part_out[(size_t)blockIdx.x * DN_MAXP + k] = mine;
        return;
    }
    constexpr int CNT = PASS == 1 ? 2 : (PASS == 2 ? 25 : 2 * DN_H1);
    float acc[CNT];
#pragma unroll
    for (int k = 0; k < CNT; ++k) acc[k] = 0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < rows; t += (int64_t)nblocks * 256) {
        int64_t e;
        float w, gy;
        operands(t, e, w, gy);
        const double wd = (double)w;
        const float xv = x[e];
        DnFwd f;
        dn_layer1(prm, st, xv, f);
        const float z3 = dn_layer2(prm, st, f);
        f.z3h = ((z3 - st[DN_M3]) - st[DN_M3L]) * st[DN_I3];
        const float yv = f.z3h * prm[DN_G3] + prm[DN_BE3];
        const float dyh3 = yv > 0.f ? gy : 0.f;
        if (PASS == 1) { acc[0] += dyh3; acc[1] += dyh3 * f.z3h; continue; }
        const float dz3 = prm[DN_G3] * st[DN_I3] * (float)((double)dyh3 - wd * m3[0] - wd * ((double)f.z3h * m3[1]));
        float dyh2[DN_H2];
#pragma unroll
        for (int j = 0; j < DN_H2; ++j) dyh2[j] = f.h2[j] > 0.f ? prm[DN_W3 + j] * dz3 : 0.f;
        if (PASS == 2) {
#pragma unroll
            for (int j = 0; j < DN_H2; ++j) { acc[j] += dz3 * f.h2[j]; acc[9 + j] += dyh2[j]; acc[17 + j] += dyh2[j] * f.z2h[j]; }
            acc[8] += dz3;
            continue;
        }
        // PASS == 4
        float dz2[DN_H2];
#pragma unroll
        for (int j = 0; j < DN_H2; ++j) dz2[j] = prm[DN_G2 + j] * st[DN_I2 + j] * (float)((double)dyh2[j] - wd * m2[j] - wd * ((double)f.z2h[j] * m2[DN_H2 + j]));
#pragma unroll
        for (int c = 0; c < DN_H1; ++c) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < DN_H2; ++j) a += prm[DN_W2 + j * DN_H1 + c] * dz2[j];
            const float dyh1 = f.h1[c] > 0.f ? a : 0.f;
            const float x1h = prm[DN_W1 + c] * f.xh * inv1[c];          // normalised layer-1 pre-activation
            const float dz1 = prm[DN_G1 + c] * inv1[c] * (float)((double)dyh1 - wd * m1[c] - wd * ((double)x1h * m1[DN_H1 + c]));
            acc[c] += dz1 * xv;
            acc[DN_H1 + c] += dz1;
        }
    }
    dn_block_partials<CNT>(acc, part_out);
}

// B5: all parameter gradients, in the layout of the parameter block
__device__ __forceinline__ void densitynet_grads_pass(const double* __restrict__ p1, const double* __restrict__ p2,
                                                               const double* __restrict__ p3, const double* __restrict__ p4,
                                                               float* __restrict__ grads, int nblocks) {
    __shared__ double sums[DN_MAXP];
    dn_sum_partials(p1, nblocks, 2, sums);
    if (threadIdx.x == 0) { grads[DN_BE3] = (float)sums[0]; grads[DN_G3] = (float)sums[1]; }
    dn_sum_partials(p2, nblocks, 25, sums);
    if (threadIdx.x < DN_H2) {
        grads[DN_W3 + threadIdx.x] = (float)sums[threadIdx.x];
        grads[DN_BE2 + threadIdx.x] = (float)sums[9 + threadIdx.x];
        grads[DN_G2 + threadIdx.x] = (float)sums[17 + threadIdx.x];
    }
    if (threadIdx.x == 0) grads[DN_B3] = (float)sums[8];
    dn_sum_partials(p3, nblocks, DN_MAXP, sums);
    if (threadIdx.x < DN_H2 * DN_H1) grads[DN_W2 + threadIdx.x] = (float)sums[threadIdx.x];
    if (threadIdx.x < DN_H2) grads[DN_B2 + threadIdx.x] = (float)sums[128 + threadIdx.x];
    if (threadIdx.x < DN_H1) {
        grads[DN_BE1 + threadIdx.x] = (float)sums[136 + threadIdx.x];
        grads[DN_G1 + threadIdx.x] = (float)sums[152 + threadIdx.x];
    }
    dn_sum_partials(p4, nblocks, 2 * DN_H1, sums);
    if (threadIdx.x < DN_H1) {
        grads[DN_W1 + threadIdx.x] = (float)sums[threadIdx.x];
        grads[DN_B1 + threadIdx.x] = (float)sums[DN_H1 + threadIdx.x];
    }
}

// ---- launches: up to PDA_DENSITYNET_MAX_SCALES independent problems per launch (blockIdx.y) -------------------------------
// The two scales of a PDA layer run the same nine passes on different tensors and parameters; each pass is ~10 us of
// dependency latency on 64-128 workgroups, so two problems in one launch cost what one does.
struct DnBatch {
    pda_densitynet_scale_t s[PDA_DENSITYNET_MAX_SCALES];
    int nblocks[PDA_DENSITYNET_MAX_SCALES];    // every problem keeps the grid it has on its own: same blocks, same sums, same bits
};
__device__ __forceinline__ DnRows dn_rows_of(const pda_densitynet_scale_t& S) {
    return DnRows{S.rowmap, S.row_weight, S.n_unique, S.nsample};
}
__device__ __forceinline__ double* dn_part(const pda_densitynet_scale_t& S, int k) {
    return (double*)S.scratch + (size_t)k * DN_BLOCKS * DN_MAXP;
}

template <int PASS>
__global__ __launch_bounds__(256) void densitynet_fwd_kernel(const DnBatch B) {
    const pda_densitynet_scale_t& S = B.s[blockIdx.y];
    const int nblocks = B.nblocks[blockIdx.y];
    if ((int)blockIdx.x >= nblocks) return;
    // partial buffers alternate: P1 -> a, P2 reads a writes b, P3 reads b writes a, P4 reads a
    double* in = PASS == 1 ? nullptr : dn_part(S, PASS == 3 ? 1 : 0);
    double* out = PASS == 4 ? nullptr : dn_part(S, PASS == 2 ? 1 : 0);
    densitynet_fwd_pass<PASS>(S.x, S.params, S.stats, in, out, S.y, S.n, S.eps, S.momentum, S.running[0], S.running[1], S.running[2],
                              S.running[3], S.running[4], S.running[5], nblocks, dn_rows_of(S));
}

template <int PASS>
__global__ __launch_bounds__(256) void densitynet_bwd_kernel(const DnBatch B) {
    const pda_densitynet_scale_t& S = B.s[blockIdx.y];
    const int nblocks = B.nblocks[blockIdx.y];
    if ((int)blockIdx.x >= nblocks) return;
    densitynet_bwd_pass<PASS>(S.x, S.grad_y, S.params, S.stats, dn_part(S, 0), dn_part(S, 1), dn_part(S, 2), dn_part(S, PASS - 1), S.n,
                              S.eps, nblocks, dn_rows_of(S));
}

__global__ __launch_bounds__(256) void densitynet_grads_kernel(const DnBatch B) {
    const pda_densitynet_scale_t& S = B.s[blockIdx.y];
    const int nblocks = B.nblocks[blockIdx.y];
    densitynet_grads_pass(dn_part(S, 0), dn_part(S, 1), dn_part(S, 2), dn_part(S, 3), S.grad_params, nblocks);
}

// Unique rows: the count is on the device (blocks behind it add zeros) and 5-10 x smaller than n; every pass opens with each
// block adding up the previous pass's per-block partials, 16 loads per round.  Measured over the step's four scales (36
// launches): 128 blocks 0.573 ms, 64 0.547, 32 0.690 (the staged outer-product pass of the backward wants the blocks), 16 1.07.
static int dn_grid(int64_t n, const DnRows& R) {
    const int64_t b = divup64(n, 256);
    const int cap = R.rowmap ? DN_BLOCKS / 2 : DN_BLOCKS;
    return (int)(b < cap ? b : cap);
}

}  // namespace pda

PDA_API int pda_densitynet_param_count(void) { return pda::DN_NPARAM; }
PDA_API int64_t pda_densitynet_scratch_bytes(void) {
    return (int64_t)4 * pda::DN_BLOCKS * pda::DN_MAXP * (int64_t)sizeof(double) + pda::DN_NSTAT * (int64_t)sizeof(float);
}

namespace pda {

// one launch set for all scales: the grid's x extent is the largest problem's, a smaller one's spare blocks leave at once
static int densitynet_launch(const pda_densitynet_scale_t* scales, int nscales, bool backward, hipStream_t st, const char* what) {
    PDA_REQUIRE(scales && nscales >= 1 && nscales <= PDA_DENSITYNET_MAX_SCALES, "%s: 1..%d scales", what, PDA_DENSITYNET_MAX_SCALES);
    DnBatch B{};
    int grid = 1;
    for (int i = 0; i < nscales; ++i) {
        const pda_densitynet_scale_t& S = scales[i];
        PDA_REQUIRE(S.n >= 1, "%s: n = %lld", what, (long long)S.n);
        PDA_REQUIRE((S.rowmap == nullptr) == (S.row_weight == nullptr) && (S.rowmap == nullptr) == (S.n_unique == nullptr),
                    "%s: rowmap, row_weight and n_unique come together", what);
        PDA_REQUIRE(!S.rowmap || (S.nsample >= 1 && S.n % S.nsample == 0), "%s: n = %lld is not groups x nsample = %d", what,
                    (long long)S.n, S.nsample);
        PDA_REQUIRE(S.x && S.params && S.stats && S.scratch && (backward ? (S.grad_y && S.grad_params) : (S.y != nullptr)),
                    "%s: null pointer", what);
        bool any = false, all = true;
        for (int k = 0; k < 6; ++k) { any = any || S.running[k]; all = all && S.running[k]; }
        PDA_REQUIRE(backward || any == all, "%s: the six running statistics come together", what);
        B.s[i] = S;
        const int g = dn_grid(S.n, DnRows{S.rowmap, S.row_weight, S.n_unique, S.nsample});
        B.nblocks[i] = g;
        grid = g > grid ? g : grid;
    }
    const dim3 g3((unsigned)grid, (unsigned)nscales), b3(256);
    if (!backward) {
        hipLaunchKernelGGL(densitynet_fwd_kernel<1>, g3, b3, 0, st, B);
        hipLaunchKernelGGL(densitynet_fwd_kernel<2>, g3, b3, 0, st, B);
        hipLaunchKernelGGL(densitynet_fwd_kernel<3>, g3, b3, 0, st, B);
        hipLaunchKernelGGL(densitynet_fwd_kernel<4>, g3, b3, 0, st, B);
    } else {
        hipLaunchKernelGGL(densitynet_bwd_kernel<1>, g3, b3, 0, st, B);
        hipLaunchKernelGGL(densitynet_bwd_kernel<2>, g3, b3, 0, st, B);
        hipLaunchKernelGGL(densitynet_bwd_kernel<3>, g3, b3, 0, st, B);
        hipLaunchKernelGGL(densitynet_bwd_kernel<4>, g3, b3, 0, st, B);
        hipLaunchKernelGGL(densitynet_grads_kernel, dim3(1, (unsigned)nscales), b3, 0, st, B);
    }
    return check_launch(what);
}

static pda_densitynet_scale_t dn_one(const float* x, const float* grad_y, const float* params, float* y, float* stats, void* scratch,
                                     float* grad_params, int64_t n, float eps, float momentum) {
    pda_densitynet_scale_t S{};
    S.x = x; S.grad_y = grad_y; S.params = params; S.y = y; S.stats = stats; S.scratch = scratch; S.grad_params = grad_params;
    S.n = n; S.nsample = 1; S.eps = eps; S.momentum = momentum;
    return S;
}

}  // namespace pda

PDA_API int pda_densitynet_fwd_multi(const pda_densitynet_scale_t* scales, int nscales, pda_stream_t stream) {
    return pda::densitynet_launch(scales, nscales, false, (hipStream_t)stream, "pda_densitynet_fwd_multi");
}

PDA_API int pda_densitynet_bwd_multi(const pda_densitynet_scale_t* scales, int nscales, pda_stream_t stream) {
    return pda::densitynet_launch(scales, nscales, true, (hipStream_t)stream, "pda_densitynet_bwd_multi");
}

PDA_API int pda_densitynet_fwd(const float* x, const float* params, float* y, float* stats, void* scratch, float* running_mean1,
                               float* running_var1, float* running_mean2, float* running_var2, float* running_mean3,
                               float* running_var3, int64_t n, float eps, float momentum, pda_stream_t stream) {
    pda_densitynet_scale_t S = pda::dn_one(x, nullptr, params, y, stats, scratch, nullptr, n, eps, momentum);
    float* r[6] = {running_mean1, running_var1, running_mean2, running_var2, running_mean3, running_var3};
    for (int k = 0; k < 6; ++k) S.running[k] = r[k];
    return pda::densitynet_launch(&S, 1, false, (hipStream_t)stream, "pda_densitynet_fwd");
}

PDA_API int pda_densitynet_bwd(const float* x, const float* grad_y, const float* params, const float* stats, float* grad_params,
                               void* scratch, int64_t n, float eps, pda_stream_t stream) {
    pda_densitynet_scale_t S = pda::dn_one(x, grad_y, params, nullptr, const_cast<float*>(stats), scratch, grad_params, n, eps, 0.f);
    return pda::densitynet_launch(&S, 1, true, (hipStream_t)stream, "pda_densitynet_bwd");
}

PDA_API int pda_densitynet_fwd_unique(const float* x, const float* params, float* y, float* stats, void* scratch,
                                      float* running_mean1, float* running_var1, float* running_mean2, float* running_var2,
                                      float* running_mean3, float* running_var3, int64_t n, const int32_t* rowmap,
                                      const float* row_weight, const int32_t* n_unique, int nsample, float eps, float momentum,
                                      pda_stream_t stream) {
    PDA_REQUIRE(rowmap && row_weight && n_unique, "pda_densitynet_fwd_unique: null pointer");
    pda_densitynet_scale_t S = pda::dn_one(x, nullptr, params, y, stats, scratch, nullptr, n, eps, momentum);
    float* r[6] = {running_mean1, running_var1, running_mean2, running_var2, running_mean3, running_var3};
    for (int k = 0; k < 6; ++k) S.running[k] = r[k];
    S.rowmap = rowmap; S.row_weight = row_weight; S.n_unique = n_unique; S.nsample = nsample;
    return pda::densitynet_launch(&S, 1, false, (hipStream_t)stream, "pda_densitynet_fwd_unique");
}

PDA_API int pda_densitynet_bwd_unique(const float* x, const float* grad_y, const float* params, const float* stats,
                                      float* grad_params, void* scratch, int64_t n, const int32_t* rowmap, const float* row_weight,
                                      const int32_t* n_unique, int nsample, float eps, pda_stream_t stream) {
    PDA_REQUIRE(rowmap && row_weight && n_unique, "pda_densitynet_bwd_unique: null pointer");
    pda_densitynet_scale_t S = pda::dn_one(x, grad_y, params, nullptr, const_cast<float*>(stats), scratch, grad_params, n, eps, 0.f);
    S.rowmap = rowmap; S.row_weight = row_weight; S.n_unique = n_unique; S.nsample = nsample;
    return pda::densitynet_launch(&S, 1, true, (hipStream_t)stream, "pda_densitynet_bwd_unique");
}

// ---- inference: BatchNorm folded into the convolutions (running statistics), the three layers in one launch -------------
// folded: w1[16] b1[16] W2[8][16] b2[8] w3[8] b3 (pointnet2_modules._folded_conv_bn per layer)
namespace pda {
constexpr int DN_EVAL_NPARAM = 2 * DN_H1 + DN_H2 * DN_H1 + DN_H2 + DN_H2 + 1;

__global__ __launch_bounds__(256) void densitynet_eval_kernel(const float* __restrict__ x, const float* __restrict__ folded,
                                                              float* __restrict__ y, int64_t n) {
    __shared__ float prm[DN_EVAL_NPARAM];
    for (int k = threadIdx.x; k < DN_EVAL_NPARAM; k += 256) prm[k] = folded[k];
    __syncthreads();
    const float *w1 = prm, *b1 = prm + DN_H1, *W2 = prm + 2 * DN_H1, *b2 = W2 + DN_H2 * DN_H1, *w3 = b2 + DN_H2;
    const float b3 = w3[DN_H2];
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        const float xv = x[t];
        float h1[DN_H1];
#pragma unroll
        for (int c = 0; c < DN_H1; ++c) h1[c] = fmaxf(w1[c] * xv + b1[c], 0.f);
        float z3 = b3;
#pragma unroll
        for (int j = 0; j < DN_H2; ++j) {
            float a = b2[j];
#pragma unroll
            for (int c = 0; c < DN_H1; ++c) a += W2[j * DN_H1 + c] * h1[c];
            z3 += w3[j] * fmaxf(a, 0.f);
        }
        y[t] = fmaxf(z3, 0.f);
    }
}
}  // namespace pda

PDA_API int pda_densitynet_eval_param_count(void) { return pda::DN_EVAL_NPARAM; }

PDA_API int pda_densitynet_eval(const float* x, const float* folded, float* y, int64_t n, pda_stream_t stream) {
    PDA_REQUIRE(n >= 0, "pda_densitynet_eval: n = %lld", (long long)n);
    if (n == 0) return PDA_OK;
    PDA_REQUIRE(x && folded && y, "pda_densitynet_eval: null pointer");
    const int64_t b = pda::divup64(n, 256);
    hipLaunchKernelGGL(pda::densitynet_eval_kernel, dim3((unsigned)(b < 2048 ? b : 2048)), dim3(256), 0, (hipStream_t)stream, x, folded, y, n);
    return pda::check_launch("pda_densitynet_eval");
}

// ---- PDA grouper geometry -------------------------------------------------------------------------------
// QueryAndGroup_alone_grouped_density_directional (pointnet2_utils.py:590-607) + the relative-position
// assembly of the PDA layer (pointnet2_modules.py:905-913) + PointConvDensitySetAbstraction's per-group max
// normalisation (:1000-1001), for the point-major layout: from xyz (B,N,3), centres (B,M,3), idx (B,M,ns):
//   rppe (B,M,ns,12) = [centre, neighbour, centre - neighbour, (neighbour - centre) / r]
//   dscale (B,M,ns)  = density / max over the group, density = exp(-|d|^2 / (2 r^2)) / (2.5 r)
// One thread per (centre, neighbour); the group maximum is a butterfly over the ns lanes of the group.
// Through torch this is 12 launches over (tokens, 1..12) tensors.  Coordinates carry no gradient.
namespace pda {

__global__ __launch_bounds__(256) void pda_geometry_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                           const int* __restrict__ idx, float* __restrict__ rppe,
                                                           float* __restrict__ dscale, int n, int m, int ns, int64_t total,
                                                           float r) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (b, centre, neighbour) flattened; ns | 64 | 256
    const bool live = e < total;
    const int64_t ec = live ? e : total - 1;
    const int64_t bm = ec / ns;
    const int64_t b = bm / m;
    const int k = idx[ec];
    const float* p = xyz + ((size_t)b * n + k) * 3;
    const float* c = new_xyz + (size_t)bm * 3;
    const float px = p[0], py = p[1], pz = p[2], cx = c[0], cy = c[1], cz = c[2];
    const float dx = px - cx, dy = py - cy, dz = pz - cz;
    const float dist = sqrtf(dx * dx + dy * dy + dz * dz);               // torch.norm
    const float dens = expf(-(dist * dist) / (2.f * r * r)) / (2.5f * r);
    float mx = dens;
    for (int o = ns >> 1; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (!live) return;
    dscale[e] = dens / mx;
    float4* out = reinterpret_cast<float4*>(rppe + (size_t)e * 12);
    out[0] = make_float4(cx, cy, cz, px);
    out[1] = make_float4(py, pz, -dx, -dy);
    out[2] = make_float4(-dz, dx / r, dy / r, dz / r);
}

}  // namespace pda

PDA_API int pda_pda_geometry(const float* xyz, const float* new_xyz, const int32_t* idx, float* rppe, float* dscale, int b, int n,
                             int m, int nsample, float radius, pda_stream_t stream) {
    PDA_REQUIRE(b >= 0 && n >= 1 && m >= 0 && nsample >= 1 && nsample <= 64 && (nsample & (nsample - 1)) == 0,
                "pda_pda_geometry: b=%d n=%d m=%d nsample=%d (nsample: power of two <= 64)", b, n, m, nsample);
    const int64_t total = (int64_t)b * m * nsample;
    if (total == 0) return PDA_OK;
    PDA_REQUIRE(xyz && new_xyz && idx && rppe && dscale, "pda_pda_geometry: null pointer");
    PDA_REQUIRE(((uintptr_t)rppe & 15) == 0, "pda_pda_geometry: rppe must be 16-byte aligned");
    hipLaunchKernelGGL(pda::pda_geometry_kernel, dim3((unsigned)pda::divup64(total, 256)), dim3(256), 0, (hipStream_t)stream, xyz,
                       new_xyz, idx, rppe, dscale, n, m, nsample, total, radius);
    return pda::check_launch("pda_pda_geometry");
}
