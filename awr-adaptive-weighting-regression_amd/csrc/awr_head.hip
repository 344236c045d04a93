// AWR head, GT dense map, Huber losses, Adam/SGD -- the HBM-bound part of the hot path.
//
// Every kernel here streams its operands exactly once with 16-byte loads (4 consecutive pixels of
// one feature-map row per lane), keeps the whole per-(image,joint) reduction in registers / LDS
// and writes the result once.  Compiled with -ffp-contract=off: the GT map has hard thresholds
// (heat-map >= 0, depth < 0.99), so its arithmetic is kept operation-for-operation identical to
// the reference's (util/feature_tool.py:29-35) to avoid mask flips from fused multiply-adds.
#include <math.h>
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>

#include "awr_common.h"

namespace awr {

static thread_local char g_err[512] = "ok";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

constexpr float kBeta = 30.0f;     // util/feature_tool.py:60
constexpr float kDepthBg = 0.99f;  // util/feature_tool.py:35, :57

// pixel-centre coordinate in [-1,1]; util/feature_tool.py:23-24, :50-51 (exact in fp32 for F = 2^k)
__device__ __forceinline__ float grid_coord(int i, int F) { return 2.0f * ((float)i + 0.5f) / (float)F - 1.0f; }

__device__ __forceinline__ float ks_dis(float ks, float h) { return ks - h * ks; }      // util/feature_tool.py:61

struct Px4 {  // 4 consecutive pixels of one feature row
    float d[4], cx[4], cy;
};

__device__ __forceinline__ Px4 load_px4(const float* __restrict__ img, int b, int p0, int F, int H) {
    Px4 r;
    const int rs = H / F;
    const int y = p0 / F, x = p0 - y * F;
    const float* row = img + ((int64_t)b * H + (int64_t)y * rs) * H;
    if (rs == 2) {  // the common case: 8 consecutive floats, take the even ones
        float4 a = ld4(row + 2 * x), c = ld4(row + 2 * x + 4);
        r.d[0] = a.x; r.d[1] = a.z; r.d[2] = c.x; r.d[3] = c.z;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) r.d[i] = row[(x + i) * rs];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) r.cx[i] = grid_coord(x + i, F);
    r.cy = grid_coord(y, F);
    return r;
}

__device__ __forceinline__ float comp(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// ------------------------------------------------------------------------------------------
// head forward: one workgroup per (image, joint); online softmax over all P pixels.
// ------------------------------------------------------------------------------------------
struct SoftAcc {
    float m, s, a0, a1, a2;
};
__device__ __forceinline__ SoftAcc combine(const SoftAcc& x, const SoftAcc& y) {
    SoftAcc r;
    r.m = fmaxf(x.m, y.m);
    const float fx = (x.m == -INFINITY) ? 0.f : expf(x.m - r.m);
    const float fy = (y.m == -INFINITY) ? 0.f : expf(y.m - r.m);
    r.s = x.s * fx + y.s * fy;
    r.a0 = x.a0 * fx + y.a0 * fy;
    r.a1 = x.a1 * fx + y.a1 * fy;
    r.a2 = x.a2 * fx + y.a2 * fy;
    return r;
}

__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ offset, const float* __restrict__ img,
                                                       int J, int F, int H, float ks, float* __restrict__ jt,
                                                       float* __restrict__ stat) {
    const int bj = blockIdx.x, b = bj / J, j = bj - b * J;
    const int P = F * F;
    const float* vec = offset + ((int64_t)b * 4 * J + 3 * j) * P;
    const float* ht = offset + ((int64_t)b * 4 * J + 3 * J + j) * P;
    SoftAcc acc = {-INFINITY, 0.f, 0.f, 0.f, 0.f};
    for (int p0 = threadIdx.x * 4; p0 < P; p0 += 256 * 4) {
        const Px4 px = load_px4(img, b, p0, F, H);
        const float4 h4 = ld4(ht + p0), v0 = ld4(vec + p0), v1 = ld4(vec + P + p0), v2 = ld4(vec + 2 * P + p0);
        float h[4], l[4];
        float cm = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float mk = px.d[i] < kDepthBg ? 1.f : 0.f;
            h[i] = comp(h4, i) * mk;  // masked pixels keep logit 0 (feature_tool.py:59-60)
            l[i] = h[i] * kBeta;
            cm = fmaxf(cm, l[i]);
        }
        if (cm > acc.m) {
            const float sc = (acc.m == -INFINITY) ? 0.f : expf(acc.m - cm);
            acc.s *= sc; acc.a0 *= sc; acc.a1 *= sc; acc.a2 *= sc;
            acc.m = cm;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float mk = px.d[i] < kDepthBg ? 1.f : 0.f;
            const float e = expf(l[i] - acc.m);
            const float dis = ks - h[i] * ks;  // feature_tool.py:61
            acc.s += e;
            acc.a0 += (comp(v0, i) * mk * dis + px.cx[i]) * e;
            acc.a1 += (comp(v1, i) * mk * dis + px.cy) * e;
            acc.a2 += (comp(v2, i) * mk * dis + px.d[i]) * e;
        }
    }
    // wave reduction with the softmax-merge operator, then across the 4 waves through LDS
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        SoftAcc other;
        other.m = __shfl_xor(acc.m, o, 64);
        other.s = __shfl_xor(acc.s, o, 64);
        other.a0 = __shfl_xor(acc.a0, o, 64);
        other.a1 = __shfl_xor(acc.a1, o, 64);
        other.a2 = __shfl_xor(acc.a2, o, 64);
        acc = combine(acc, other);
    }
    __shared__ SoftAcc part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        SoftAcc t = combine(combine(part[0], part[1]), combine(part[2], part[3]));
        float* o = jt + (int64_t)bj * 3;
        o[0] = t.a0 / t.s;
        o[1] = t.a1 / t.s;
        o[2] = t.a2 / t.s;
        if (stat) {
            stat[2 * bj] = t.m;
            stat[2 * bj + 1] = t.s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// head backward: pure streaming given the saved (max, sum-exp) and the forward output.
//   d/dvec[c,p] = g_c * w_p * dis_p * m_p
//   d/dht[p]    = m_p * sum_c g_c * ( -ks * w_p * vec_c,p * m_p + 30 * w_p * (val_c,p - out_c) )
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ offset, const float* __restrict__ img,
                                                       const float* __restrict__ jt, const float* __restrict__ stat,
                                                       const float* __restrict__ g_jt, int J, int F, int H, float ks,
                                                       float* __restrict__ g_offset, int accumulate) {
    const int bj = blockIdx.y, b = bj / J, j = bj - b * J;
    const int P = F * F;
    const int p0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (p0 >= P) return;
    const int64_t voff = ((int64_t)b * 4 * J + 3 * j) * P + p0, hoff = ((int64_t)b * 4 * J + 3 * J + j) * P + p0;
    const Px4 px = load_px4(img, b, p0, F, H);
    const float4 h4 = ld4(offset + hoff), v0 = ld4(offset + voff), v1 = ld4(offset + voff + P), v2 = ld4(offset + voff + 2 * P);
    const float mx = stat[2 * bj], inv_s = 1.0f / stat[2 * bj + 1];
    const float g0 = g_jt[bj * 3], g1 = g_jt[bj * 3 + 1], g2 = g_jt[bj * 3 + 2];
    const float o0 = jt[bj * 3], o1 = jt[bj * 3 + 1], o2 = jt[bj * 3 + 2];
    float r0[4], r1[4], r2[4], rh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float mk = px.d[i] < kDepthBg ? 1.f : 0.f;
        const float h = comp(h4, i) * mk;
        const float w = expf(h * kBeta - mx) * inv_s;
        const float dis = ks - h * ks;
        const float a0 = comp(v0, i) * mk, a1 = comp(v1, i) * mk, a2 = comp(v2, i) * mk;
        const float wd = w * dis * mk;
        r0[i] = g0 * wd;
        r1[i] = g1 * wd;
        r2[i] = g2 * wd;
        const float t0 = -ks * a0 + kBeta * (a0 * dis + px.cx[i] - o0);
        const float t1 = -ks * a1 + kBeta * (a1 * dis + px.cy - o1);
        const float t2 = -ks * a2 + kBeta * (a2 * dis + px.d[i] - o2);
        rh[i] = mk * w * (g0 * t0 + g1 * t1 + g2 * t2);
    }
    float4 q0 = make_float4(r0[0], r0[1], r0[2], r0[3]), q1 = make_float4(r1[0], r1[1], r1[2], r1[3]);
    float4 q2 = make_float4(r2[0], r2[1], r2[2], r2[3]), qh = make_float4(rh[0], rh[1], rh[2], rh[3]);
    if (accumulate) {
        const float4 e0 = ld4(g_offset + voff), e1 = ld4(g_offset + voff + P), e2 = ld4(g_offset + voff + 2 * P), eh = ld4(g_offset + hoff);
        q0.x += e0.x; q0.y += e0.y; q0.z += e0.z; q0.w += e0.w;
        q1.x += e1.x; q1.y += e1.y; q1.z += e1.z; q1.w += e1.w;
        q2.x += e2.x; q2.y += e2.y; q2.z += e2.z; q2.w += e2.w;
        qh.x += eh.x; qh.y += eh.y; qh.z += eh.z; qh.w += eh.w;
    }
    st4(g_offset + voff, q0);
    st4(g_offset + voff + P, q1);
    st4(g_offset + voff + 2 * P, q2);
    st4(g_offset + hoff, qh);
}

// ------------------------------------------------------------------------------------------
// GT dense map for 4 pixels of joint j (util/feature_tool.py:29-39), operation for operation.
// ------------------------------------------------------------------------------------------
struct Gt4 {
    float u0[4], u1[4], u2[4], hm[4];
};
__device__ __forceinline__ Gt4 gt_map4(const Px4& px, float j0, float j1, float j2, float ks) {
    Gt4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float o0 = j0 - px.cx[i], o1 = j1 - px.cy, o2 = j2 - px.d[i];
        const float dist = sqrtf(((o0 * o0 + o1 * o1) + o2 * o2) + 1e-8f);
        const float hm = (ks - dist) / ks;
        const float mk = (hm >= 0.f ? 1.f : 0.f) * (px.d[i] < kDepthBg ? 1.f : 0.f);
        r.u0[i] = o0 / dist * mk;
        r.u1[i] = o1 / dist * mk;
        r.u2[i] = o2 / dist * mk;
        r.hm[i] = hm * mk;
    }
    return r;
}

__global__ __launch_bounds__(256) void joint2offset_kernel(const float* __restrict__ jt_gt, const float* __restrict__ img, int J,
                                                           int F, int H, float ks, float* __restrict__ out) {
    const int bj = blockIdx.y, b = bj / J, j = bj - b * J;
    const int P = F * F;
    const int p0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (p0 >= P) return;
    const Px4 px = load_px4(img, b, p0, F, H);
    const Gt4 g = gt_map4(px, jt_gt[bj * 3], jt_gt[bj * 3 + 1], jt_gt[bj * 3 + 2], ks);
    const int64_t voff = ((int64_t)b * 4 * J + 3 * j) * P + p0, hoff = ((int64_t)b * 4 * J + 3 * J + j) * P + p0;
    st4(out + voff, make_float4(g.u0[0], g.u0[1], g.u0[2], g.u0[3]));
    st4(out + voff + P, make_float4(g.u1[0], g.u1[1], g.u1[2], g.u1[3]));
    st4(out + voff + 2 * P, make_float4(g.u2[0], g.u2[1], g.u2[2], g.u2[3]));
    st4(out + hoff, make_float4(g.hm[0], g.hm[1], g.hm[2], g.hm[3]));
}

// Huber pieces (model/loss.py:8-25): value and d/dz
__device__ __forceinline__ float huber_val(float z, float delta) {
    const float a = fabsf(z);
    return a < delta ? 0.5f * z * z : delta * (a - 0.5f * delta);
}
__device__ __forceinline__ float huber_grad(float z, float delta) { return fminf(fmaxf(z, -delta), delta); }

// Loss accumulators.  Default: fp64 atomic adds of the block partials.  Deterministic mode (awr_set_deterministic): the SAME 8 bytes
// hold a signed 2^-50 fixed-point number and the partials are added with INTEGER atomics -- associative, so the sum does not depend
// on the order in which the workgroups finish.  Range [0, 2048) (losses are O(1) and below; larger / NaN values are flagged, not wrapped), resolution 9e-16 per partial: far below
// the fp32 loss that is reported.  awr_loss_finalize converts back.
constexpr double LOSS_FIX = 1125899906842624.0;      // 2^50 (the encoding is chosen by awr_get_deterministic() at launch time: do not
                                                      // toggle the mode between accumulating into `acc` and awr_loss_finalize)

__device__ __forceinline__ void block_accumulate(double local, double scale, double* acc, int fixed_point) {
    __shared__ double part[4];
    local = wave_sum(local);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double v = (part[0] + part[1] + part[2] + part[3]) * scale;
        if (fixed_point) {
            // Huber partials are >= 0, so the running integer stays far below bit 61 for any finite loss < 2048.  A NaN partial sets
            // bit 61, a partial that does not fit (an exploding loss) bit 62: loss_finalize_kernel maps them back to NaN / +inf instead
            // of letting the integer wrap into a plausible finite number.
            unsigned long long* a = reinterpret_cast<unsigned long long*>(acc);
            if (v != v) atomicOr(a, 1ULL << 61);
            else if (!(v < 2048.0)) atomicOr(a, 1ULL << 62);
            else atomicAdd(a, (unsigned long long)__double2ll_rn(v * LOSS_FIX));
        } else {
            atomicAdd(acc, v);
        }
    }
}

// fused GT map + dense Huber forward/backward: reads pred + depth once, writes the gradient once.
__global__ __launch_bounds__(256) void dense_loss_kernel(const float* __restrict__ pred, const float* __restrict__ jt_gt,
                                                         const float* __restrict__ img, int J, int F, int H, float ks, float delta,
                                                         float gscale, double lscale, double* __restrict__ acc,
                                                         float* __restrict__ g_offset, int accumulate, int fixed_point) {
    const int bj = blockIdx.y, b = bj / J, j = bj - b * J;
    const int P = F * F;
    const int p0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    double local = 0.0;
    if (p0 < P) {
        const Px4 px = load_px4(img, b, p0, F, H);
        const Gt4 g = gt_map4(px, jt_gt[bj * 3], jt_gt[bj * 3 + 1], jt_gt[bj * 3 + 2], ks);
        const int64_t voff = ((int64_t)b * 4 * J + 3 * j) * P + p0, hoff = ((int64_t)b * 4 * J + 3 * J + j) * P + p0;
        const int64_t offs[4] = {voff, voff + P, voff + 2 * P, hoff};
        const float* gts[4] = {g.u0, g.u1, g.u2, g.hm};
        float lsum = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 pv = ld4(pred + offs[c]);
            float gr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float z = comp(pv, i) - gts[c][i];
                lsum += huber_val(z, delta);
                gr[i] = huber_grad(z, delta) * gscale;
            }
            if (g_offset) {
                float4 q = make_float4(gr[0], gr[1], gr[2], gr[3]);
                if (accumulate) {
                    const float4 e = ld4(g_offset + offs[c]);
                    q.x += e.x; q.y += e.y; q.z += e.z; q.w += e.w;
                }
                st4(g_offset + offs[c], q);
            }
        }
        local = (double)lsum;
    }
    block_accumulate(local, lscale, acc, fixed_point);
}

__global__ __launch_bounds__(256) void huber_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t n, float delta,
                                                    float gscale, double lscale, double* __restrict__ acc, float* __restrict__ gx,
                                                    int accumulate, int fixed_point) {
    double local = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float z = x[i] - y[i];
        local += (double)huber_val(z, delta);
        if (gx) {
            const float g = huber_grad(z, delta) * gscale;
            gx[i] = accumulate ? gx[i] + g : g;
        }
    }
    block_accumulate(local, lscale, acc, fixed_point);
}

// ------------------------------------------------------------------------------------------
// NHWC forms (round 3): head forward, fused GT map + dense Huber, head backward on the backbone's own layout.
// ------------------------------------------------------------------------------------------
// The head GEMM leaves the dense map as (B, P, Cp) rows -- Cp = 4J rounded up to 32, channel 3j + c = offset component c of joint
// j, channel 3J + j = its heat map (util/feature_tool.py:46-48) -- and the backward wants its gradient in the same layout; the NCHW
// kernels above forced a transpose pass in each direction (2 x 29 us, 235 MB per batch-64 step).  Here a workgroup walks a run of
// 64-pixel tiles of one image: the tile (64 x Cp floats, one contiguous piece of HBM) is requested one tile ahead with 16-byte loads
// (4-16 in flight per thread), parked in LDS, and thread (pixel slot, joint) reads its four values from there -- every element of the
// map is fetched once, whatever the per-joint reduction needs.  MODE bits: 1 = online-softmax partials of offset2joint_softmax
// (util/feature_tool.py:41-65) per (image, chunk, joint), merged by head_finish_kernel; 2 = GT map + dense Huber value and gradient
// (util/feature_tool.py:12-39, model/loss.py:8-25), written in place into the tile and streamed out as the NHWC gradient;
// 4 = the head's backward (closed form, see head_bwd_kernel) added onto that gradient.  One launch does 1|2 (coord_weight == 0: the
// reference default) -- the dense map is read ONCE per step -- or 1, then 2|4 once the joints exist.
constexpr int TPX = 64;      // pixels per tile

// exp(x) for x <= 0 (softmax terms after the max subtraction): two-term range reduction with explicit FMAs (x log2 e = n + r,
// |r| <= 1/2, log2 e split into a high and a low part so that r keeps full precision for |x| up to several hundred), the hardware
// exp2 on r (1 ulp there) and an exact scaling by 2^n -- 7 instructions against ~18 for the library expf (which also guards overflow
// and positive arguments), ~2 ulp.  The NHWC kernels are VALU-bound (ISA count: ~300 instructions per (pixel, joint)), not HBM-bound.
__device__ __forceinline__ float exp_nonpos(float x) {
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-08f;
    const float n = rintf(x * L2E_HI);
    float r = __builtin_fmaf(x, L2E_HI, -n);
    r = __builtin_fmaf(x, L2E_LO, r);
    return ldexpf(__builtin_amdgcn_exp2f(r), (int)n);
}

struct nhwc_args {
    const float* pred;       // (B, P, Cp)
    const float* img;        // (B, 1, H, H)
    const float* jt_gt;      // (B, J, 3)     MODE & 2
    const float* jt;         // (B, J, 3)     MODE & 4: the head's output
    const float* stat;       // (B, J, 2)     MODE & 4: softmax max / sum
    const float* g_jt;       // (B, J, 3)     MODE & 4: upstream gradient
    float* partial;          // (B, chunks, J, 5)   MODE & 1
    float* grad;             // (B, P, Cp)    MODE & 6
    double* acc;             // dense-loss accumulator   MODE & 2
    int J, F, H, Cp, tiles_per_wg, lgF;      // lgF: log2(F) when F is a power of two (every map of both backbones), else -1
    float ks, delta, gscale;
    double lscale;
    int fixed_point;
};

template <int JS, int MODE, int DEPTH, bool POW2>
__global__ __launch_bounds__(256) void dense_nhwc_kernel(const nhwc_args a) {
    constexpr int NS = 256 / JS;          // pixel slots per pass
    constexpr int PPT = TPX / NS;         // pixels per thread per tile
    constexpr int NLMAX = JS / 4;         // float4 loads per thread per tile: 64 * Cp / 4 / 256 = Cp / 16 <= 4 JS / 16
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int pitch = a.Cp + 4;
    float* const tile = smem;                    // [TPX][pitch]
    float* const dtile = smem + TPX * pitch;     // [TPX] depth of the tile's pixels
    const int tid = threadIdx.x, j = tid % JS, slot = tid / JS;
    const int b = blockIdx.y, chunk = blockIdx.x, J = a.J, F = a.F, P = F * F;
    const bool active = j < J;
    const int rs = a.H / F;
    const int NL = a.Cp >> 4, C4 = a.Cp >> 2;
    const int tile0 = chunk * a.tiles_per_wg;
    int ntile = P / TPX - tile0;
    if (ntile > a.tiles_per_wg) ntile = a.tiles_per_wg;
    const float* src = a.pred + ((int64_t)b * P + (int64_t)tile0 * TPX) * a.Cp;
    float* dst = (MODE & 6) ? a.grad + ((int64_t)b * P + (int64_t)tile0 * TPX) * a.Cp : nullptr;
    const float* dimg = a.img + (int64_t)b * a.H * a.H;

    // per-(image, joint) constants
    float gj0 = 0.f, gj1 = 0.f, gj2 = 0.f;                 // GT joint
    float mx = 0.f, inv_s = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f;
    if (active) {
        const int bj = b * J + j;
        if (MODE & 2) { gj0 = a.jt_gt[bj * 3]; gj1 = a.jt_gt[bj * 3 + 1]; gj2 = a.jt_gt[bj * 3 + 2]; }
        if (MODE & 4) {
            mx = a.stat[2 * bj]; inv_s = 1.0f / a.stat[2 * bj + 1];
            g0 = a.g_jt[bj * 3]; g1 = a.g_jt[bj * 3 + 1]; g2 = a.g_jt[bj * 3 + 2];
            o0 = a.jt[bj * 3]; o1 = a.jt[bj * 3 + 1]; o2 = a.jt[bj * 3 + 2];
        }
    }
    SoftAcc sa = {-INFINITY, 0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;
    const float inv_f = 1.0f / (float)F;

    // DEPTH register sets: tile t + DEPTH is requested while tile t is being worked on (DEPTH = 2 doubles the bytes a workgroup keeps
    // in flight for 16 more registers)
    float4 r[DEPTH][NLMAX];
    float rd[DEPTH];
    auto request = [&](int t, float4 (&rr)[NLMAX], float& rdd) {       // tile t of this workgroup: 64 * Cp contiguous floats + the 64 depth samples
        const float* s = src + (int64_t)t * TPX * a.Cp;
#pragma unroll
        for (int i = 0; i < NLMAX; ++i)
            if (i < NL) rr[i] = ld4(s + ((int64_t)(tid + 256 * i) << 2));
        if (tid < TPX) {
            const int p = (tile0 + t) * TPX + tid, y = p / F, x = p - y * F;
            rdd = dimg[(int64_t)y * rs * a.H + x * rs];
        }
    };
    auto body = [&](int t, float4 (&rr)[NLMAX], float& rdd) {
#pragma unroll
        for (int i = 0; i < NLMAX; ++i)
            if (i < NL) {
                const int e = tid + 256 * i, px = e / C4, c4 = e - px * C4;
                st4(tile + px * pitch + 4 * c4, rr[i]);
            }
        if (tid < TPX) dtile[tid] = rdd;
        __syncthreads();
        if (t + DEPTH < ntile) request(t + DEPTH, rr, rdd);
        if (active) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int px = slot + NS * k;
                const int p = (tile0 + t) * TPX + px;
                int x, y;
                float cx, cy;
                if (POW2) {            // shifts, and 1/F is exact: (2 i + 1) * (1/F) - 1 == 2 (i + 0.5) / F - 1 bit for bit
                    y = p >> a.lgF; x = p & (F - 1);
                    cx = (float)(2 * x + 1) * inv_f - 1.0f; cy = (float)(2 * y + 1) * inv_f - 1.0f;
                } else {
                    y = p / F; x = p - y * F;
                    cx = grid_coord(x, F); cy = grid_coord(y, F);
                }
                const float d = dtile[px];
                float* row = tile + px * pitch;
                const float v0 = row[3 * j], v1 = row[3 * j + 1], v2 = row[3 * j + 2], hraw = row[3 * J + j];
                const float mk = d < kDepthBg ? 1.f : 0.f;
                const float h = hraw * mk;             // masked pixels keep logit 0 (feature_tool.py:59-60)
                const float dis = ks_dis(a.ks, h);
                if (MODE & 1) {
                    const float l = h * kBeta;
                    if (l > sa.m) {
                        const float sc = (sa.m == -INFINITY) ? 0.f : exp_nonpos(sa.m - l);
                        sa.s *= sc; sa.a0 *= sc; sa.a1 *= sc; sa.a2 *= sc;
                        sa.m = l;
                    }
                    const float e = exp_nonpos(l - sa.m);
                    sa.s += e;
                    sa.a0 += (v0 * mk * dis + cx) * e;
                    sa.a1 += (v1 * mk * dis + cy) * e;
                    sa.a2 += (v2 * mk * dis + d) * e;
                }
                if (MODE & 6) {
                    float q0 = 0.f, q1 = 0.f, q2 = 0.f, qh = 0.f;
                    if (MODE & 2) {       // util/feature_tool.py:29-39, operation for operation (gt_map4)
                        // The GT map is zero wherever its mask is (background pixels; pixels farther than ks from the joint): there
                        // x - (+-0) == x exactly, so the correctly rounded sqrt / divides (~150 VALU instructions per (pixel, joint),
                        // what made the NCHW kernel compute-bound at 2.2 TB/s) only run where the mask can be 1.
                        float z0 = v0, z1 = v1, z2 = v2, zh = hraw;
                        if (mk != 0.f) {
                            const float e0 = gj0 - cx, e1 = gj1 - cy, e2 = gj2 - d;
                            const float dist = sqrtf(((e0 * e0 + e1 * e1) + e2 * e2) + 1e-8f);
                            const float hm = (a.ks - dist) / a.ks;
                            if (hm >= 0.f) { z0 = v0 - e0 / dist; z1 = v1 - e1 / dist; z2 = v2 - e2 / dist; zh = hraw - hm; }
                        }
                        lsum += huber_val(z0, a.delta); lsum += huber_val(z1, a.delta); lsum += huber_val(z2, a.delta); lsum += huber_val(zh, a.delta);
                        q0 = huber_grad(z0, a.delta) * a.gscale; q1 = huber_grad(z1, a.delta) * a.gscale;
                        q2 = huber_grad(z2, a.delta) * a.gscale; qh = huber_grad(zh, a.delta) * a.gscale;
                    }
                    if (MODE & 4) {       // head_bwd_kernel's closed form
                        const float w = expf(h * kBeta - mx) * inv_s;
                        const float a0 = v0 * mk, a1 = v1 * mk, a2 = v2 * mk;
                        const float wd = w * dis * mk;
                        const float t0 = -a.ks * a0 + kBeta * (a0 * dis + cx - o0);
                        const float t1 = -a.ks * a1 + kBeta * (a1 * dis + cy - o1);
                        const float t2 = -a.ks * a2 + kBeta * (a2 * dis + d - o2);
                        q0 += g0 * wd; q1 += g1 * wd; q2 += g2 * wd;
                        qh += mk * w * (g0 * t0 + g1 * t1 + g2 * t2);
                    }
                    row[3 * j] = q0; row[3 * j + 1] = q1; row[3 * j + 2] = q2; row[3 * J + j] = qh;
                }
            }
        }
        if (MODE & 6) {       // the tile now holds the gradient (padding channels: the map's own zeros): stream it out
            __syncthreads();
            float* o = dst + (int64_t)t * TPX * a.Cp;
#pragma unroll
            for (int i = 0; i < NLMAX; ++i)
                if (i < NL) {
                    const int e = tid + 256 * i, px = e / C4, c4 = e - px * C4;
                    st4(o + ((int64_t)e << 2), ld4(tile + px * pitch + 4 * c4));
                }
        }
        __syncthreads();
    };
#pragma unroll
    for (int q = 0; q < DEPTH; ++q)
        if (q < ntile) request(q, r[q], rd[q]);
    for (int t = 0; t < ntile; t += DEPTH) {
#pragma unroll
        for (int q = 0; q < DEPTH; ++q)
            if (t + q < ntile) body(t + q, r[q], rd[q]);
    }
    if (MODE & 1) {
        // lanes j, j + JS, ... of a wave hold partials of the same joint: fold them, then the four waves through LDS
#pragma unroll
        for (int o = 32; o >= JS; o >>= 1) {
            SoftAcc other;
            other.m = __shfl_xor(sa.m, o, 64); other.s = __shfl_xor(sa.s, o, 64);
            other.a0 = __shfl_xor(sa.a0, o, 64); other.a1 = __shfl_xor(sa.a1, o, 64); other.a2 = __shfl_xor(sa.a2, o, 64);
            sa = combine(sa, other);
        }
        SoftAcc* part = reinterpret_cast<SoftAcc*>(smem);      // [4][JS] (the tile is dead)
        const int lane = tid & 63, wave = tid >> 6;
        if (lane < JS) part[wave * JS + lane] = sa;
        __syncthreads();
        if (tid < JS && tid < J) {
            SoftAcc tacc = part[tid];
            if (JS < 64) {
#pragma unroll
                for (int w = 1; w < 4; ++w) tacc = combine(tacc, part[w * JS + tid]);
            } else {
                tacc = combine(combine(part[tid], part[JS + tid]), combine(part[2 * JS + tid], part[3 * JS + tid]));
            }
            float* o = a.partial + (((int64_t)b * gridDim.x + chunk) * J + tid) * 5;
            o[0] = tacc.m; o[1] = tacc.s; o[2] = tacc.a0; o[3] = tacc.a1; o[4] = tacc.a2;
        }
        __syncthreads();
    }
    if (MODE & 2) block_accumulate((double)lsum, a.lscale, a.acc, a.fixed_point);
}

// merge the per-chunk softmax partials -> joints (B, J, 3) + (max, sum) for the backward; optionally the coordinate Huber loss
// (train.py:125: crit(jt_uvd_pred, jt_uvd_gt)) and its gradient w.r.t. the joints.  A 32-lane half-wave per (image, joint): lane c takes
// chunks c, c + 32, ... and the halves fold with the softmax-merge operator (a thread per joint walking the chunks one after the other
// took 16 us -- as long as the streaming pass itself).
__global__ __launch_bounds__(256) void head_finish_kernel(const float* __restrict__ partial, int chunks, int J, int BJ, const float* __restrict__ jt_gt,
                                                          float delta, float gscale, double lscale, float* __restrict__ jt, float* __restrict__ stat,
                                                          float* __restrict__ g_jt, double* __restrict__ acc, int fixed_point) {
    const int i = blockIdx.x * 8 + (threadIdx.x >> 5), c0 = threadIdx.x & 31;
    double local = 0.0;
    SoftAcc t = {-INFINITY, 0.f, 0.f, 0.f, 0.f};
    if (i < BJ) {
        const int b = i / J, j = i - b * J;
        for (int c = c0; c < chunks; c += 32) {
            const float* q = partial + (((int64_t)b * chunks + c) * J + j) * 5;
            SoftAcc o = {q[0], q[1], q[2], q[3], q[4]};
            t = combine(t, o);
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        SoftAcc other;
        other.m = __shfl_xor(t.m, o, 64); other.s = __shfl_xor(t.s, o, 64);
        other.a0 = __shfl_xor(t.a0, o, 64); other.a1 = __shfl_xor(t.a1, o, 64); other.a2 = __shfl_xor(t.a2, o, 64);
        t = combine(t, other);
    }
    if (i < BJ && c0 == 0) {
        const float out[3] = {t.a0 / t.s, t.a1 / t.s, t.a2 / t.s};
#pragma unroll
        for (int c = 0; c < 3; ++c) jt[i * 3 + c] = out[c];
        if (stat) { stat[2 * i] = t.m; stat[2 * i + 1] = t.s; }
        if (acc) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float z = out[c] - jt_gt[i * 3 + c];
                local += (double)huber_val(z, delta);
                if (g_jt) g_jt[i * 3 + c] = huber_grad(z, delta) * gscale;
            }
        }
    }
    if (acc) block_accumulate(local, lscale, acc, fixed_point);
}

__global__ void zero_f64_kernel(double* p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}
__global__ void loss_finalize_kernel(double* acc, int n, float* out, int fixed_point, int reset) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < n; ++i) {
            double v = acc[i];
            if (fixed_point) {
                const unsigned long long q = reinterpret_cast<const unsigned long long*>(acc)[i];
                v = (q & (1ULL << 61)) ? (double)NAN : (q >> 62) ? (double)INFINITY : (double)(long long)q / LOSS_FIX;
            }
            out[i] = (float)v;
            t += v;
            if (reset) acc[i] = 0.0;      // (all-zero bits in both encodings) the next step accumulates from zero without a fill launch
        }
        out[n] = (float)t;
    }
}

// ------------------------------------------------------------------------------------------
// optimisers over flat arenas (torch.optim.Adam / SGD single-tensor update rules)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n4, int64_t n, float one_minus_b1, float b2,
                                                   float one_minus_b2, float eps, float wd, float step_size, float bc2_sqrt,
                                                   float gscale) {
    const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 < n4) {
        float4 pp = ld4(p + i4 * 4), gg = ld4(g + i4 * 4), mm = ld4(m + i4 * 4), vv = ld4(v + i4 * 4);
        float* pa = &pp.x; float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gr = ga[k] * gscale;
            if (wd != 0.f) gr = gr + wd * pa[k];
            ma[k] = ma[k] + (gr - ma[k]) * one_minus_b1;      // exp_avg.lerp_(grad, 1-beta1)
            va[k] = va[k] * b2 + one_minus_b2 * (gr * gr);    // exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
            const float denom = sqrtf(va[k]) / bc2_sqrt + eps;
            pa[k] = pa[k] - step_size * (ma[k] / denom);      // param.addcdiv_(exp_avg, denom, -step_size)
        }
        st4(p + i4 * 4, pp); st4(m + i4 * 4, mm); st4(v + i4 * 4, vv);
    }
    // scalar tail (n not a multiple of 4)
    if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) {
        const int64_t i = n4 * 4 + threadIdx.x;
        float gr = g[i] * gscale;
        if (wd != 0.f) gr = gr + wd * p[i];
        const float mk = m[i] + (gr - m[i]) * one_minus_b1;
        const float vk = v[i] * b2 + one_minus_b2 * (gr * gr);
        m[i] = mk; v[i] = vk;
        p[i] = p[i] - step_size * (mk / (sqrtf(vk) / bc2_sqrt + eps));
    }
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, int64_t n,
                                                  float lr, float mom, float wd, int first, float gscale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float gr = g[i] * gscale;
    if (wd != 0.f) gr = gr + wd * p[i];
    const float b = first ? gr : buf[i] * mom + gr;
    buf[i] = b;
    p[i] = p[i] - lr * b;
}

}  // namespace awr

using namespace awr;

template <int MODE>
static int launch_dense_nhwc(const nhwc_args& a, int B, int chunks, hipStream_t st) {
    const dim3 grid((unsigned)chunks, (unsigned)B);
    const size_t lds = (size_t)(TPX * (a.Cp + 4) + TPX) * sizeof(float);
    static const int depth = []() { const char* e = getenv("AWR_NHWC_DEPTH"); return e ? atoi(e) : 1; }();      // tuning hook (measured: 2 is no faster, profiles/r03_summary.md)
#define AWR_NHWC_LAUNCH(js, dp)                                                                                           \
    do {                                                                                                                   \
        if (a.lgF >= 0) hipLaunchKernelGGL((dense_nhwc_kernel<js, MODE, dp, true>), grid, dim3(256), lds, st, a);          \
        else hipLaunchKernelGGL((dense_nhwc_kernel<js, MODE, dp, false>), grid, dim3(256), lds, st, a);                    \
    } while (0)
    if (a.J <= 16 && depth == 2) AWR_NHWC_LAUNCH(16, 2);
    else if (a.J <= 16) AWR_NHWC_LAUNCH(16, 1);
    else if (a.J <= 32) AWR_NHWC_LAUNCH(32, 1);
    else AWR_NHWC_LAUNCH(64, 1);
#undef AWR_NHWC_LAUNCH
    return check_launch("dense_nhwc_kernel");
}

extern "C" {

int awr_version(void) { return 100; }
const char* awr_last_error(void) { return awr::g_err; }

int awr_device_info(int* n_cu, int* clock_mhz, char* name, int name_len) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        set_error("hipGetDeviceProperties failed");
        return AWR_ERR_HIP;
    }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = prop.clockRate / 1000;
    if (name && name_len > 0) {
        strncpy(name, prop.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    return AWR_OK;
}

static int check_head_dims(int B, int J, int F, int H) {
    AWR_REQUIRE(B > 0 && J > 0 && F > 0 && H > 0, "head: B,J,F,H must be positive (got %d,%d,%d,%d)", B, J, F, H);
    AWR_REQUIRE(F % 4 == 0 && H % F == 0, "head: need F %% 4 == 0 and H %% F == 0 (F=%d, H=%d)", F, H);
    AWR_REQUIRE((H / F) != 2 || H % 8 == 0, "head: H must be a multiple of 8");
    return AWR_OK;
}

int awr_head_forward(const float* offset, const float* img, int B, int J, int F, int H, float ks, float* jt, float* stat,
                     void* stream) {
    if (int e = check_head_dims(B, J, F, H)) return e;
    AWR_REQUIRE(offset && img && jt, "head_forward: null pointer");
    hipLaunchKernelGGL(head_fwd_kernel, dim3(B * J), dim3(256), 0, as_stream(stream), offset, img, J, F, H, ks, jt, stat);
    return check_launch("head_fwd_kernel");
}

int awr_head_backward(const float* offset, const float* img, const float* jt, const float* stat, const float* g_jt, int B, int J,
                      int F, int H, float ks, float* g_offset, int accumulate, void* stream) {
    if (int e = check_head_dims(B, J, F, H)) return e;
    AWR_REQUIRE(offset && img && jt && stat && g_jt && g_offset, "head_backward: null pointer");
    const int P = F * F;
    hipLaunchKernelGGL(head_bwd_kernel, dim3((P / 4 + 255) / 256, B * J), dim3(256), 0, as_stream(stream), offset, img, jt, stat,
                       g_jt, J, F, H, ks, g_offset, accumulate);
    return check_launch("head_bwd_kernel");
}

int awr_joint2offset(const float* jt_gt, const float* img, int B, int J, int F, int H, float ks, float* out, void* stream) {
    if (int e = check_head_dims(B, J, F, H)) return e;
    AWR_REQUIRE(jt_gt && img && out, "joint2offset: null pointer");
    const int P = F * F;
    hipLaunchKernelGGL(joint2offset_kernel, dim3((P / 4 + 255) / 256, B * J), dim3(256), 0, as_stream(stream), jt_gt, img, J, F, H,
                       ks, out);
    return check_launch("joint2offset_kernel");
}

int awr_dense_loss(const float* offset_pred, const float* jt_gt, const float* img, int B, int J, int F, int H, float ks, float delta,
                   float weight, double* acc, float* g_offset, int accumulate, void* stream) {
    if (int e = check_head_dims(B, J, F, H)) return e;
    AWR_REQUIRE(offset_pred && jt_gt && img && acc, "dense_loss: null pointer");
    const int P = F * F;
    const double n = (double)B * 4.0 * J * P;
    hipLaunchKernelGGL(dense_loss_kernel, dim3((P / 4 + 255) / 256, B * J), dim3(256), 0, as_stream(stream), offset_pred, jt_gt, img,
                       J, F, H, ks, delta, (float)((double)weight / n), (double)weight / n, acc, g_offset, accumulate, awr_get_deterministic());
    return check_launch("dense_loss_kernel");
}

// ---- NHWC forms ------------------------------------------------------------------------------------------------------
static int log2_exact(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return (1 << s) == v ? s : -1;
}

static int nhwc_geometry(int B, int J, int F, int H, int Cp, int* chunks, int* tiles_per_wg) {
    if (int e = check_head_dims(B, J, F, H)) return e;
    AWR_REQUIRE(J <= 64, "head (NHWC): at most 64 joints (got %d)", J);
    // (64 (Cp + 4) + 64) floats of dynamic LDS per workgroup must stay below the 64 KB a launch gets without opting in: Cp <= 224, i.e. J <= 56;
    // wider maps take the NCHW kernels (awr_plan_set_nhwc_boundary refuses, the engines fall back)
    AWR_REQUIRE(Cp >= 4 * J && Cp % 32 == 0 && Cp <= 224, "head (NHWC): Cp=%d must be a multiple of 32 in [4J, 224]", Cp);
    AWR_REQUIRE(Cp <= 4 * (J <= 16 ? 16 : J <= 32 ? 32 : 64), "head (NHWC): Cp=%d too wide for %d joints (expected 4J rounded up to 32)", Cp, J);
    const int P = F * F;
    AWR_REQUIRE(P % TPX == 0, "head (NHWC): F*F must be a multiple of %d", TPX);
    const int tiles = P / TPX;
    // every tile after a workgroup's first is requested while the previous one is being worked on: as many tiles per workgroup as
    // still leave ~4 workgroups per CU (1024 per launch), at most 16
    static const int want_wgs = []() { const char* e = getenv("AWR_NHWC_WGS"); return e ? atoi(e) : 1024; }();      // tuning hook
    int per = 1;
    while (per < 16 && per * 2 <= tiles && (int64_t)B * (tiles / (per * 2)) >= want_wgs) per *= 2;
    *tiles_per_wg = per;
    *chunks = (tiles + per - 1) / per;
    return AWR_OK;
}

int64_t awr_head_nhwc_scratch(int B, int J, int F) {
    if (B <= 0 || J <= 0 || F <= 0) return 0;
    return (int64_t)B * ((int64_t)F * F / TPX + 1) * J * 5;      // floats: one partial per (image, tile, joint) at most
}

int awr_head_forward_nhwc(const float* pred, int Cp, const float* img, int B, int J, int F, int H, float ks, float* scratch, float* jt,
                          float* stat, void* stream) {
    AWR_REQUIRE(pred && img && scratch && jt, "head_forward_nhwc: null pointer");
    int chunks, per;
    if (int e = nhwc_geometry(B, J, F, H, Cp, &chunks, &per)) return e;
    nhwc_args a;
    memset(&a, 0, sizeof a);
    a.pred = pred; a.img = img; a.partial = scratch; a.J = J; a.F = F; a.H = H; a.Cp = Cp; a.tiles_per_wg = per; a.ks = ks; a.lgF = log2_exact(F);
    hipStream_t st = as_stream(stream);
    if (int e = launch_dense_nhwc<1>(a, B, chunks, st)) return e;
    hipLaunchKernelGGL(head_finish_kernel, dim3((B * J + 7) / 8), dim3(256), 0, st, scratch, chunks, J, B * J, nullptr, 0.f, 0.f, 0.0, jt, stat,
                       nullptr, nullptr, 0);
    return check_launch("head_finish_kernel");
}

int awr_head_loss_step_nhwc(const float* pred, int Cp, const float* img, const float* jt_gt, int B, int J, int F, int H, float ks, float delta,
                            float coord_weight, float dense_weight, float* scratch, float* jt, float* stat, float* g_jt, double* acc, float* grad,
                            void* stream) {
    AWR_REQUIRE(pred && img && jt_gt && scratch && jt && stat && acc && grad, "head_loss_step_nhwc: null pointer");
    AWR_REQUIRE(coord_weight == 0.f || g_jt, "head_loss_step_nhwc: a coordinate loss needs the g_jt buffer");
    int chunks, per;
    if (int e = nhwc_geometry(B, J, F, H, Cp, &chunks, &per)) return e;
    const int P = F * F;
    const double nd = (double)B * 4.0 * J * P, nc = (double)B * J * 3.0;
    const int fixed = awr_get_deterministic();
    nhwc_args a;
    memset(&a, 0, sizeof a);
    a.pred = pred; a.img = img; a.jt_gt = jt_gt; a.partial = scratch; a.grad = grad; a.acc = acc + 1;
    a.J = J; a.F = F; a.H = H; a.Cp = Cp; a.tiles_per_wg = per; a.ks = ks; a.delta = delta; a.lgF = log2_exact(F);
    a.gscale = (float)((double)dense_weight / nd); a.lscale = (double)dense_weight / nd; a.fixed_point = fixed;
    hipStream_t st = as_stream(stream);
    const bool coord = coord_weight != 0.f;
    // coord_weight == 0 (config.py:41, the reference default): ONE pass over the map does the softmax partials, the dense loss and its
    // gradient; otherwise the joints have to exist before the head's backward can run: partials -> finish -> dense loss + head backward
    if (int e = coord ? launch_dense_nhwc<1>(a, B, chunks, st) : launch_dense_nhwc<3>(a, B, chunks, st)) return e;
    hipLaunchKernelGGL(head_finish_kernel, dim3((B * J + 7) / 8), dim3(256), 0, st, scratch, chunks, J, B * J, jt_gt, delta,
                       (float)((double)coord_weight / nc), (double)coord_weight / nc, jt, stat, coord ? g_jt : nullptr, acc, fixed);
    if (int e = check_launch("head_finish_kernel")) return e;
    if (coord) {
        a.jt = jt; a.stat = stat; a.g_jt = g_jt;
        if (int e = launch_dense_nhwc<6>(a, B, chunks, st)) return e;
    }
    return AWR_OK;
}

int awr_huber(const float* x, const float* y, int64_t n, float delta, float weight, double* acc, float* gx, int accumulate,
              void* stream) {
    AWR_REQUIRE(x && y && acc && n > 0, "huber: bad arguments");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(huber_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, y, n, delta,
                       (float)((double)weight / (double)n), (double)weight / (double)n, acc, gx, accumulate, awr_get_deterministic());
    return check_launch("huber_kernel");
}

int awr_zero_f64(double* p, int64_t n, void* stream) {
    AWR_REQUIRE(p && n > 0, "zero_f64: bad arguments");
    hipLaunchKernelGGL(zero_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), p, n);
    return check_launch("zero_f64_kernel");
}

int awr_loss_finalize(const double* acc, int n, float* out, void* stream) {
    AWR_REQUIRE(acc && out && n > 0 && n <= 16, "loss_finalize: bad arguments");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, as_stream(stream), const_cast<double*>(acc), n, out, awr_get_deterministic(), 0);
    return check_launch("loss_finalize_kernel");
}

int awr_loss_finalize_reset(double* acc, int n, float* out, void* stream) {
    AWR_REQUIRE(acc && out && n > 0 && n <= 16, "loss_finalize_reset: bad arguments");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, as_stream(stream), acc, n, out, awr_get_deterministic(), 1);
    return check_launch("loss_finalize_kernel");
}

int awr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int64_t step, float grad_scale, void* stream) {
    AWR_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
    AWR_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: arenas must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const int64_t n4 = n / 4;
    const int64_t blocks = (n4 + 255) / 256 > 0 ? (n4 + 255) / 256 : 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p, g, m, v, n4, n,
                       (float)(1.0 - (double)beta1), beta2, (float)(1.0 - (double)beta2), eps, weight_decay, (float)((double)lr / bc1),
                       (float)sqrt(bc2), grad_scale);
    return check_launch("adam_kernel");
}

int awr_sgd_step(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float weight_decay, int64_t step,
                 float grad_scale, void* stream) {
    AWR_REQUIRE(p && g && buf && n > 0 && step >= 1, "sgd_step: bad arguments");
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), p, g, buf, n, lr, momentum,
                       weight_decay, step == 1 ? 1 : 0, grad_scale);
    return check_launch("sgd_kernel");
}

}  // extern "C"
