// Fused ResNet18-deconv stem for gfx950 (model/resnet_deconv.py:31-36, :118-121):
//
//     conv 5x5 pad 2 (1 -> 64 channels, no bias) -> BatchNorm -> ReLU -> MaxPool(3, 2, 1)
//
// The stem's full-resolution output (64 channels at HxW: 4 MB per 128x128 image, 268 MB at batch 64) used to be
// written once and re-read six times per train step (im2col GEMM out, BN statistics, max-pool, pool backward, BN
// backward reduce + apply, weight gradient).  A 5x5 conv of a ONE-channel image costs 25 MACs per output -- cheaper to
// recompute from the 64 KB image than to move: the kernels below never materialise that tensor.
//
//   forward   stem_stats_kernel    conv -> per-channel sum / sum of squares (BatchNorm batch statistics)
//             stem_pool_kernel     conv -> scale/shift -> ReLU -> 3x3/2 max-pool (+ argmax) : writes the POOLED map only
//   backward  stem_bwd_kernel<0>   conv again -> ReLU mask, pool-backward gather -> sum g, sum g*xhat (BN backward sums)
//             stem_bwd_kernel<1>   conv again -> dY = BN backward -> dW[64][25] += dY^T * im2col(image)
//
// The Hourglass stem (model/hourglass.py:112: conv 5x5 WITH bias -> BatchNorm -> ReLU, no pooling) uses the same kernels in their
// "dense" form: stem_conv_kernel writes relu(bn(conv + bias)) at full resolution (the only tensor the next layer needs), the
// backward kernels take the dense gradient of that tensor instead of scattering pooled gradients, and recompute the conv
// output for the ReLU mask, xhat and the weight gradient (the un-normalised conv output is never stored either).
//
// The conv runs on the FP32 matrix pipe (v_mfma_f32_32x32x2_f32, exact f32): a workgroup stages its image patch in LDS,
// the A operand of K-step kk is ONE ds_read_b32 per lane (pixel row of the lane + the tap offset of k = 2 kk + lane/32),
// the B operand (the 64x25 filter bank, 13 K-steps of two taps) lives in 13 registers for the whole kernel.  In the
// weight-gradient kernel the accumulator registers of the conv (lane = channel, register = pixel) ARE, register for
// register, the A operand of the second GEMM (dY^T x im2col): no transposition, no LDS round trip.
#include "awr_common.h"

namespace awr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ST_T = 25;         // taps
constexpr int ST_TB = 26;        // columns of the weight-gradient result: 25 taps + the bias gradient (a tap whose image value is 1)
constexpr int ST_KS = 13;        // MFMA K-steps (2 taps each; tap 25 carries a zero weight)
constexpr int ST_PW = 48;        // patch pitch of the 16x16-pixel kernels: the two 16-pixel runs of a half-wave sit 16 banks apart
constexpr int ST_PH = 20;        // patch rows / used columns (16 + 2 * 2)
constexpr int SP_PW = 49;        // patch pitch of the pooling kernel (21 used columns).  Its 32 conv pixels per half-wave are a run of the 9 x 17 pixel block, i.e.
                                 // two or three patch rows of <= 17 consecutive floats: a pitch of 17 (mod 32) puts consecutive rows on disjoint banks.  (Round 1-3: 24 --
                                 // rows r and r + 1 shared 9 banks, every one of the 13 A-operand reads per tile was 2-way conflicted: 4.18 M conflict cycles per launch,
                                 // 0.23 M with 49; the kernel's time did not move -- 79 us either way -- profiles/r04_stem_lds_conflicts.txt.)

// MFMA 32x32 C/D layout: lane = column (l & 31), register r of half-wave h = row (r & 3) + 8 (r >> 2) + 4 h
__device__ __forceinline__ int mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct stem_lane {
    float w[ST_KS];      // filter taps k = 2 kk + h of channel ch0 + (lane & 31)
    int koff[ST_KS];     // patch offset of tap k
};

template <int PW>
__device__ __forceinline__ void stem_lane_init(stem_lane& s, const float* __restrict__ w, int ch, int h) {
#pragma unroll
    for (int kk = 0; kk < ST_KS; ++kk) {
        const int k = 2 * kk + h;
        s.w[kk] = k < ST_T ? w[ch * ST_T + k] : 0.f;
        s.koff[kk] = k < ST_T ? (k / 5) * PW + (k % 5) : 0;
    }
}

// y[32 pixels][32 channels] of one tile: pixel row of this lane starts at patch[base]
__device__ __forceinline__ f32x16 stem_conv_tile(const float* patch, int base, const stem_lane& s, float bias = 0.f) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias;
#pragma unroll
    for (int kk = 0; kk < ST_KS; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(patch[base + s.koff[kk]], s.w[kk], acc, 0, 0, 0);
    return acc;
}

// 20x20 image patch under a 16x16 tile of conv pixels (zero outside the image = the conv's padding), in two halves so that a
// multi-tile workgroup can request the next tile's patch before it computes the current one (256 threads: two elements each)
struct patch_regs { float v[2]; };
__device__ __forceinline__ patch_regs stem_fetch_patch16(const float* __restrict__ img, int b, int y0, int x0, int H, int W) {
    patch_regs r;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = threadIdx.x + 256 * k;
        const int py = i / ST_PH, px = i - py * ST_PH;
        const int gy = y0 - 2 + py, gx = x0 - 2 + px;
        r.v[k] = (i < ST_PH * ST_PH && gy >= 0 && gy < H && gx >= 0 && gx < W) ? img[((size_t)b * H + gy) * W + gx] : 0.f;
    }
    return r;
}
__device__ __forceinline__ void stem_commit_patch16(float* patch, const patch_regs& r) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < ST_PH * ST_PH) patch[(i / ST_PH) * ST_PW + (i % ST_PH)] = r.v[k];
    }
}
__device__ __forceinline__ void stem_load_patch16(float* patch, const float* __restrict__ img, int b, int y0, int x0, int H, int W) {
    for (int i = threadIdx.x; i < ST_PH * ST_PH; i += blockDim.x) {
        const int py = i / ST_PH, px = i - py * ST_PH;
        const int gy = y0 - 2 + py, gx = x0 - 2 + px;
        patch[py * ST_PW + px] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? img[((size_t)b * H + gy) * W + gx] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// BatchNorm batch statistics of the (never stored) conv output
// ------------------------------------------------------------------------------------------
// grid (groups of tiles_per_wg tiles of 16x16 pixels, channel half, image); 256 threads = 4 waves x 2 tiles of 32 pixels x 32
// channels.  A workgroup walks several tiles: the filter bank is loaded once and the next tile's patch is requested before the
// current tile's MFMAs (a one-tile workgroup spends most of its life waiting for its 13 + 2 loads).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void stem_stats_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ bias,
                                                         int H, int W, int tiles_per_wg, int nslots, double* __restrict__ stats) {
    __shared__ float patch[ST_PH * ST_PW];
    __shared__ double red[4][2][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, h = lane >> 5;
    const int tiles_x = W >> 4;
    const int b = blockIdx.z, ch0 = blockIdx.y * 32;
    const int tile0 = blockIdx.x * tiles_per_wg;
    patch_regs nxt = stem_fetch_patch16(img, b, (tile0 / tiles_x) * 16, (tile0 % tiles_x) * 16, H, W);
    stem_lane sl;
    stem_lane_init<ST_PW>(sl, w, ch0 + m, h);
    const float bs = bias ? bias[ch0 + m] : 0.f;
    // sums shifted by the lane's first value (a constant background gives a nearly constant channel: x^2 would round the same way
    // in every term and the variance would drown; csrc/awr_conv.hip: gemm_epilogue), back to the plain sums in fp64 per lane
    float s1 = 0.f, s2 = 0.f, c0 = 0.f;
    for (int tt = 0; tt < tiles_per_wg; ++tt) {
        if (tt) __syncthreads();           // the previous tile's readers are done with the patch
        stem_commit_patch16(patch, nxt);
        if (tt + 1 < tiles_per_wg) {
            const int tile = tile0 + tt + 1;
            nxt = stem_fetch_patch16(img, b, (tile / tiles_x) * 16, (tile % tiles_x) * 16, H, W);
        }
        __syncthreads();
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const int q = 32 * (wave * 2 + ti) + m;
            const f32x16 acc = stem_conv_tile(patch, (q >> 4) * ST_PW + (q & 15), sl, bs);
            if (tt == 0 && ti == 0) c0 = acc[0];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = acc[r] - c0;
                s1 += d;
                s2 += d * d;
            }
        }
    }
    const double n = 32.0 * tiles_per_wg;
    double t1 = (double)s1 + n * (double)c0, t2 = (double)s2 + 2.0 * (double)c0 * (double)s1 + n * (double)c0 * (double)c0;
    t1 += __shfl_xor(t1, 32, 64);
    t2 += __shfl_xor(t2, 32, 64);
    if (h == 0) {
        red[wave][0][m] = t1;
        red[wave][1][m] = t2;
    }
    __syncthreads();
    if (tid < 64) {
        const int st = tid >> 5, c = tid & 31;
        const double v = red[0][st][c] + red[1][st][c] + red[2][st][c] + red[3][st][c];
        const int slot = (blockIdx.x + blockIdx.z * gridDim.x) % nslots;
        atomicAdd(stats + ((size_t)slot * 2 + st) * 64 + ch0 + c, v);
    }
}

// ------------------------------------------------------------------------------------------
// conv (+ bias) -> affine (BatchNorm) -> [ReLU]: writes the full-resolution NHWC map (Hourglass stem)
// ------------------------------------------------------------------------------------------
// grid (tiles of 16x16 pixels, channel half, image).  A store instruction covers two pixels x 32 channels = two full 128-byte lines.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void stem_conv_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ bias,
                                                        const float* __restrict__ scale, const float* __restrict__ shift, int relu, int H, int W,
                                                        float* __restrict__ out) {
    __shared__ float patch[ST_PH * ST_PW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, h = lane >> 5;
    const int tiles_x = W >> 4, ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int b = blockIdx.z, ch0 = blockIdx.y * 32;
    stem_load_patch16(patch, img, b, ty * 16, tx * 16, H, W);
    stem_lane sl;
    stem_lane_init<ST_PW>(sl, w, ch0 + m, h);
    const float bs = bias ? bias[ch0 + m] : 0.f, sc = scale[ch0 + m], sh = shift[ch0 + m];
    const float lo = relu ? 0.f : -INFINITY;
    __syncthreads();
    float* const o0 = out + (((size_t)b * H + ty * 16) * W + tx * 16) * 64 + ch0 + m;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const int i = wave * 2 + ti, q = 32 * i + m;
        const f32x16 acc = stem_conv_tile(patch, (q >> 4) * ST_PW + (q & 15), sl, bs);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = 32 * i + mfma_row(r, h);
            o0[((size_t)(p >> 4) * W + (p & 15)) * 64] = fmaxf(acc[r] * sc + sh, lo);
        }
    }
}

// ------------------------------------------------------------------------------------------
// conv -> affine (BatchNorm) -> ReLU -> MaxPool(3,2,1): writes the pooled NHWC map (+ argmax)
// ------------------------------------------------------------------------------------------
// grid (tiles of 4 rows x 8 columns of POOLED pixels, channel half, image); 320 threads = 5 waves x one tile of 32 conv pixels:
// the 9 x 17 conv pixels under the tile (153, padded to 160) go through LDS as act[pixel][32 channels] (20 KB: six workgroups
// per CU keep the load -> conv -> pool -> store chains of different tiles overlapped), then 256 threads pool them.
constexpr int SP_RH = 9, SP_RW = 17, SP_PY = 4, SP_PX = 8;      // conv rows / columns under a tile; pooled tile shape

__global__ __launch_bounds__(320) void stem_pool_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ shift, int H, int W,
                                                        int tiles_per_wg, float* __restrict__ pooled, uint8_t* __restrict__ argmax) {
    __shared__ float patch[(SP_RH + 4) * SP_PW];
    // act[pixel][32 channels] as TWO PLANES of 16 channels, the second shifted by 16 floats (round 4).  The pooling threads read 16 bytes of
    // pixel 2 px + dx: with one 32-float row per pixel, the sixteen lanes that one LDS cycle serves -- (px0, c 0-3), (px1, c 4-7), (px2, c 4-7),
    // (px3, c 0-3) for lanes {0-3, 12-15, 20-27} -- fell on 8 distinct 16-byte slots (px stride = 64 floats = 0 mod the 16 slots): every
    // window read 2-way conflicted (PMC: 4.2 M conflict cycles per launch).  Per plane a pixel is 4 slots, px advances 8, and the shifted
    // second plane takes the other half: sixteen distinct slots; the conv's 4-byte stores (32 consecutive channels per pixel) stay
    // conflict-free (plane 0 and plane 1 of one pixel sit 16 banks apart).
    constexpr int SP_PLANE = 160 * 16 + 16;
    __shared__ __attribute__((aligned(16))) float act[2 * SP_PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, h = lane >> 5;
    const int Ho = H >> 1, Wo = W >> 1;
    const int tiles_x = Wo / SP_PX;
    const int b = blockIdx.z, ch0 = blockIdx.y * 32;
    // this thread's patch element (273 of the 320 threads hold one); the next tile's is requested before the current tile's MFMAs
    const int ppy = tid / (SP_RW + 4), ppx = tid - ppy * (SP_RW + 4);
    const bool pact = tid < (SP_RH + 4) * (SP_RW + 4);
    auto fetch = [&](int tile) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int gy = 2 * ty * SP_PY - 3 + ppy, gx = 2 * tx * SP_PX - 3 + ppx;
        return (pact && gy >= 0 && gy < H && gx >= 0 && gx < W) ? img[((size_t)b * H + gy) * W + gx] : 0.f;
    };
    const int tile0 = blockIdx.x * tiles_per_wg;
    float nxt = fetch(tile0);
    stem_lane sl;
    stem_lane_init<SP_PW>(sl, w, ch0 + m, h);
    const float sc = scale[ch0 + m], sh = shift[ch0 + m];
    int q = 32 * wave + m;
    q = q < SP_RH * SP_RW ? q : SP_RH * SP_RW - 1;       // rows 153..159 of the last tile: recomputed, never pooled
    const int qy = q / SP_RW, qx = q - qy * SP_RW;
    for (int tt = 0; tt < tiles_per_wg; ++tt) {
        const int tile = tile0 + tt;
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int oy0 = ty * SP_PY, ox0 = tx * SP_PX;
        if (tt) __syncthreads();           // the previous tile's pooling threads are done with act / patch
        if (pact) patch[ppy * SP_PW + ppx] = nxt;
        if (tt + 1 < tiles_per_wg) nxt = fetch(tile + 1);
        __syncthreads();
        {
            const f32x16 acc = stem_conv_tile(patch, qy * SP_PW + qx, sl);
#pragma unroll
            for (int r = 0; r < 16; ++r) act[(m >> 4) * SP_PLANE + (32 * wave + mfma_row(r, h)) * 16 + (m & 15)] = fmaxf(acc[r] * sc + sh, 0.f);
        }
        __syncthreads();
        if (tid < SP_PY * SP_PX * 8) {
            const int c4 = tid & 7, pp = tid >> 3, py = pp >> 3, px = pp & 7;
            float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            int am[4] = {0, 0, 0, 0};
            // window row dy <-> conv row 2 oy - 1 + dy: only row / column -1 can fall outside (H, W even), torch pads with -inf
            for (int dy = (oy0 + py == 0) ? 1 : 0; dy < 3; ++dy)
                for (int dx = (ox0 + px == 0) ? 1 : 0; dx < 3; ++dx) {
                    const float4 v = ld4(&act[(c4 >> 2) * SP_PLANE + ((2 * py + dy) * SP_RW + 2 * px + dx) * 16 + 4 * (c4 & 3)]);
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (vv[c] > mx[c]) {      // first maximum in scan order wins (ATen)
                            mx[c] = vv[c];
                            am[c] = dy * 3 + dx;
                        }
                }
            const size_t oidx = (((size_t)b * Ho + oy0 + py) * Wo + ox0 + px) * 64 + ch0 + 4 * c4;
            st4(pooled + oidx, make_float4(mx[0], mx[1], mx[2], mx[3]));
            if (argmax) *reinterpret_cast<uchar4*>(argmax + oidx) = make_uchar4((uint8_t)am[0], (uint8_t)am[1], (uint8_t)am[2], (uint8_t)am[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward: pool-backward gather + ReLU mask + BatchNorm backward, on the recomputed conv output
// ------------------------------------------------------------------------------------------
// WGRAD = 0: sums[slot][0][c] += sum g, sums[slot][1][c] += sum g * xhat          (g = d loss / d bn-output, masked)
// WGRAD = 1: dY = gi (g - k1 - xhat k2);  dw[slot][c][tap] += sum_pixels dY[pixel][c] * image[pixel + tap]
// Max-pool backward as a SCATTER: every pooled window hands its gradient to the one pixel its argmax names, so the 9 x 9
// windows that touch a 16x16 tile are routed into an LDS tile gt[pixel][channel] -- 81 (window, channel) items per channel
// instead of testing, for each of the 256 pixels, the up to four windows that contain it.  Windows of equal row / column
// parity never share a pixel (3 wide, 4 apart), so the four parity classes are added one after the other without atomics:
// the sum order is fixed (deterministic).
// DENSE (Hourglass stem, no pooling): dpool is the gradient of the full-resolution activation, (B,H,W,64) NHWC; each lane loads the
// gradient of its accumulator elements directly (two pixels x 32 channels = two 128-byte lines per load instruction).
// grid (groups of tiles_per_wg tiles of 16x16 pixels, channel half, image); 256 threads = 4 waves x 2 tiles of 32 pixels
template <int WGRAD, bool DENSE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void stem_bwd_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ bias,
                                                       const float* __restrict__ coef4, const float* __restrict__ bcoef,
                                                       const float* __restrict__ dpool, const uint8_t* __restrict__ argmax, int H, int W,
                                                       int tiles_per_wg, int nslots, double* __restrict__ sums, float* __restrict__ dw) {
    __shared__ float patch[ST_PH * ST_PW];
    __shared__ __attribute__((aligned(16))) float gt[DENSE ? 4 : 256 * 32];       // pooled gradient routed to the tile's pixels [pixel][channel]
    __shared__ float red[WGRAD ? 4 * 1024 : 4 * 2 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, h = lane >> 5;
    const int Ho = H >> 1, Wo = W >> 1;
    const int tiles_x = W >> 4;
    const int b = blockIdx.z, ch0 = blockIdx.y * 32;
    stem_lane sl;
    stem_lane_init<ST_PW>(sl, w, ch0 + m, h);
    const float sc = coef4[ch0 + m], sh = coef4[64 + ch0 + m], mu = coef4[128 + ch0 + m], is = coef4[192 + ch0 + m];
    const float bs = bias ? bias[ch0 + m] : 0.f;
    float k1 = 0.f, k2 = 0.f, gi = 0.f;
    if (WGRAD) { k1 = bcoef[ch0 + m]; k2 = bcoef[64 + ch0 + m]; gi = bcoef[128 + ch0 + m]; }
    const int tm = m < ST_T ? m : 0;
    const int toff = (tm / 5) * ST_PW + (tm % 5);          // second GEMM: this lane's column = tap m; column 25 = the bias gradient
    const float bcol = m == ST_T ? 1.f : 0.f;              // (its "image" is 1 everywhere), columns 26..31 are zero
    // scatter items of this thread, one per parity class (cy, cx): window (cy + 2 a, cx + 2 b), channels 4 c4 .. 4 c4 + 3
    const int c4 = tid & 7, wi = tid >> 3;
    float s1 = 0.f, s2 = 0.f;
    f32x16 wacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) wacc[r] = 0.f;

    for (int tt = 0; tt < tiles_per_wg; ++tt) {
        const int tile = blockIdx.x * tiles_per_wg + tt;
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int y0 = ty * 16, x0 = tx * 16;
        if constexpr (!DENSE) {
            float4 itd[4];
            uchar4 ita[4];
            int itp[4];                     // window (wy << 4 | wx) of the item, -1 = none
#pragma unroll
            for (int cls = 0; cls < 4; ++cls) {
                const int cy = cls >> 1, cx = cls & 1, nx = cx ? 4 : 5, ny = cy ? 4 : 5;
                const int wy = cy + 2 * (wi / nx), wx = cx + 2 * (wi % nx);
                const int oy = (y0 >> 1) + wy, ox = (x0 >> 1) + wx;
                itp[cls] = -1;
                if (wi < nx * ny && oy < Ho && ox < Wo) {
                    const size_t o = (((size_t)b * Ho + oy) * Wo + ox) * 64 + ch0 + 4 * c4;
                    itd[cls] = ld4(dpool + o);
                    ita[cls] = *reinterpret_cast<const uchar4*>(argmax + o);
                    itp[cls] = (wy << 4) | wx;
                }
            }
            if (tt) __syncthreads();           // the previous tile's readers are done with the LDS arrays
            stem_load_patch16(patch, img, b, y0, x0, H, W);
#pragma unroll
            for (int i = 0; i < 8; ++i) st4(&gt[(tid + 256 * i) * 4], make_float4(0, 0, 0, 0));
            __syncthreads();
#pragma unroll
            for (int cls = 0; cls < 4; ++cls) {
                if (itp[cls] >= 0) {
                    const int ly0 = 2 * (itp[cls] >> 4) - 1, lx0 = 2 * (itp[cls] & 15) - 1;       // the window's top-left pixel (may be -1)
                    const float dv[4] = {itd[cls].x, itd[cls].y, itd[cls].z, itd[cls].w};
                    const int av[4] = {ita[cls].x, ita[cls].y, ita[cls].z, ita[cls].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ky = av[j] >= 6 ? 2 : (av[j] >= 3 ? 1 : 0), kx = av[j] - 3 * ky;
                        const int ly = ly0 + ky, lx = lx0 + kx;
                        if ((unsigned)ly < 16u && (unsigned)lx < 16u) gt[(ly * 16 + lx) * 32 + 4 * c4 + j] += dv[j];
                    }
                }
                __syncthreads();
            }
        } else {
            if (tt) __syncthreads();
            stem_load_patch16(patch, img, b, y0, x0, H, W);
            __syncthreads();
        }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const int i = wave * 2 + ti;
            const int q = 32 * i + m;
            float gv[16];
            if constexpr (DENSE) {      // requested before the conv's MFMAs: their latency hides under them
                const float* g0 = dpool + (((size_t)b * H + y0) * W + x0) * 64 + ch0 + m;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = 32 * i + mfma_row(r, h);
                    gv[r] = g0[((size_t)(p >> 4) * W + (p & 15)) * 64];
                }
            }
            f32x16 acc = stem_conv_tile(patch, (q >> 4) * ST_PW + (q & 15), sl, bs);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float g = DENSE ? gv[r] : gt[(32 * i + mfma_row(r, h)) * 32 + m];
                const float y = acc[r];
                g = y * sc + sh > 0.f ? g : 0.f;
                const float xh = (y - mu) * is;
                if (WGRAD) {
                    acc[r] = gi * (g - k1 - xh * k2);
                } else {
                    s1 += g;
                    s2 += g * xh;
                }
            }
            if (WGRAD) {
                // dW[channel][tap] += sum over the tile's 32 pixels: K-step r contracts the pixels of register r (one per half-wave):
                // pixel row 2 i + (r >> 3), column 8 ((r >> 2) & 1) + (r & 3) + 4 h of the 16x16 tile
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pb = (2 * i + (r >> 3)) * ST_PW + 8 * ((r >> 2) & 1) + (r & 3) + 4 * h;
                    wacc = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[r], m < ST_T ? patch[pb + toff] : bcol, wacc, 0, 0, 0);
                }
            }
        }
    }
    if (WGRAD) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = wacc[r];
        __syncthreads();
        const int slot = (blockIdx.x + blockIdx.z * gridDim.x) % nslots;
        for (int e = tid; e < 1024; e += 256) {
            const int r = e >> 6, l = e & 63, t = l & 31;
            if (t < ST_TB) {
                const float v = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
                atomicAdd(dw + ((size_t)slot * 64 + ch0 + mfma_row(r, l >> 5)) * ST_TB + t, v);
            }
        }
    } else {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (h == 0) {
            red[(wave * 2 + 0) * 32 + m] = s1;
            red[(wave * 2 + 1) * 32 + m] = s2;
        }
        __syncthreads();
        if (tid < 64) {
            const int st = tid >> 5, c = tid & 31;
            const double v = (double)red[(0 * 2 + st) * 32 + c] + (double)red[(1 * 2 + st) * 32 + c] + (double)red[(2 * 2 + st) * 32 + c] +
                             (double)red[(3 * 2 + st) * 32 + c];
            const int slot = (blockIdx.x + blockIdx.z * gridDim.x) % nslots;
            atomicAdd(sums + ((size_t)slot * 2 + st) * 64 + ch0 + c, v);
        }
    }
}

// grad[c][tap] (and gbias[c] = column 25) = sum over the slot copies; re-arms the accumulator for the next step
__global__ void stem_dw_finalize_kernel(float* __restrict__ dw_slots, int nslots, float* __restrict__ grad, float* __restrict__ gbias) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * ST_TB) return;
    float s = 0.f;
    for (int k = 0; k < nslots; ++k) {
        s += dw_slots[k * 64 * ST_TB + i];
        dw_slots[k * 64 * ST_TB + i] = 0.f;
    }
    const int c = i / ST_TB, t = i - c * ST_TB;
    if (t < ST_T) grad[c * ST_T + t] = s;
    else if (gbias) gbias[c] = s;
}

}  // namespace awr

using namespace awr;

extern "C" {

#define AWR_STEM_GEOMETRY(name)                                                                                              \
    AWR_REQUIRE(B > 0 && H >= 16 && W >= 16 && H % 16 == 0 && W % 16 == 0, name ": H=%d, W=%d must be positive multiples of 16", H, W); \
    AWR_REQUIRE((int64_t)B * H * W < (1LL << 31), name ": batch too large")

// tiles a workgroup walks: the largest divisor of the tile count that is <= want (power of two)
static int stem_tiles_per_wg(int tiles, int want) {
    int tpw = want;
    while (tiles % tpw) tpw >>= 1;
    return tpw;
}

// nslots (0 = AWR_STAT_SLOTS): slot copies of the accumulator the kernel may spread over; awr_stem_slots() of them give every
// workgroup its own copy (deterministic mode)
int awr_stem_slots(int B, int H, int W, int* stats_slots, int* wgrad_slots) {
    AWR_REQUIRE(stats_slots && wgrad_slots, "stem_slots: null pointer");
    AWR_STEM_GEOMETRY("stem_slots");
    const int tiles = (H / 16) * (W / 16);
    int tpw = 8;
    while (tiles % tpw) tpw >>= 1;
    *stats_slots = tiles * B;
    *wgrad_slots = tiles / tpw * B;
    return AWR_OK;
}

int awr_stem_stats(const float* img, const float* w, const float* bias, int B, int H, int W, double* stats, int nslots, void* stream) {
    AWR_REQUIRE(img && w && stats && nslots >= 0, "stem_stats: null pointer");
    AWR_STEM_GEOMETRY("stem_stats");
    const int tiles = (H / 16) * (W / 16), tpw = stem_tiles_per_wg(tiles, 4);
    hipLaunchKernelGGL(stem_stats_kernel, dim3(tiles / tpw, 2, B), dim3(256), 0, as_stream(stream), img, w, bias, H, W, tpw,
                       nslots ? nslots : AWR_STAT_SLOTS, stats);
    return check_launch("stem_stats_kernel");
}

int awr_stem_conv(const float* img, const float* w, const float* bias, const float* scale, const float* shift, int relu, int B, int H, int W,
                  float* out, void* stream) {
    AWR_REQUIRE(img && w && scale && shift && out, "stem_conv: null pointer");
    AWR_STEM_GEOMETRY("stem_conv");
    hipLaunchKernelGGL(stem_conv_kernel, dim3((H / 16) * (W / 16), 2, B), dim3(256), 0, as_stream(stream), img, w, bias, scale, shift, relu, H, W, out);
    return check_launch("stem_conv_kernel");
}

int awr_stem_pool(const float* img, const float* w, const float* scale, const float* shift, int B, int H, int W, float* pooled,
                  uint8_t* argmax, void* stream) {
    AWR_REQUIRE(img && w && scale && shift && pooled, "stem_pool: null pointer");
    AWR_STEM_GEOMETRY("stem_pool");
    const int tiles = (H / 2 / SP_PY) * (W / 2 / SP_PX), tpw = stem_tiles_per_wg(tiles, 4);
    hipLaunchKernelGGL(stem_pool_kernel, dim3(tiles / tpw, 2, B), dim3(320), 0, as_stream(stream), img, w, scale, shift, H, W, tpw, pooled, argmax);
    return check_launch("stem_pool_kernel");
}

int awr_stem_bwd_reduce(const float* img, const float* w, const float* bias, const float* coef4, const float* dg, const uint8_t* argmax, int B,
                        int H, int W, double* sums, int nslots, void* stream) {
    AWR_REQUIRE(img && w && coef4 && dg && sums && nslots >= 0, "stem_bwd_reduce: null pointer");
    AWR_STEM_GEOMETRY("stem_bwd_reduce");
    const int tiles = (H / 16) * (W / 16), tpw = stem_tiles_per_wg(tiles, 4);
    const dim3 grid(tiles / tpw, 2, B);
    if (argmax)
        hipLaunchKernelGGL((stem_bwd_kernel<0, false>), grid, dim3(256), 0, as_stream(stream), img, w, bias, coef4, (const float*)nullptr, dg, argmax, H, W,
                           tpw, nslots ? nslots : AWR_STAT_SLOTS, sums, (float*)nullptr);
    else
        hipLaunchKernelGGL((stem_bwd_kernel<0, true>), grid, dim3(256), 0, as_stream(stream), img, w, bias, coef4, (const float*)nullptr, dg, argmax, H, W,
                           tpw, nslots ? nslots : AWR_STAT_SLOTS, sums, (float*)nullptr);
    return check_launch("stem_bwd_kernel<0>");
}

int awr_stem_bwd_wgrad(const float* img, const float* w, const float* bias, const float* coef4, const float* bwd_coef, const float* dg,
                       const uint8_t* argmax, int B, int H, int W, float* dw_slots, float* grad, float* gbias, int nslots, void* stream) {
    AWR_REQUIRE(img && w && coef4 && bwd_coef && dg && dw_slots && grad && nslots >= 0, "stem_bwd_wgrad: null pointer");
    if (nslots == 0) nslots = AWR_STAT_SLOTS;
    AWR_STEM_GEOMETRY("stem_bwd_wgrad");
    const int tiles = (H / 16) * (W / 16);
    int tpw = 8;                                   // tiles per workgroup: fewer, longer workgroups = fewer atomics on the 64x26 result
    while (tiles % tpw) tpw >>= 1;
    const dim3 grid(tiles / tpw, 2, B);
    if (argmax)
        hipLaunchKernelGGL((stem_bwd_kernel<1, false>), grid, dim3(256), 0, as_stream(stream), img, w, bias, coef4, bwd_coef, dg, argmax, H, W, tpw, nslots,
                           (double*)nullptr, dw_slots);
    else
        hipLaunchKernelGGL((stem_bwd_kernel<1, true>), grid, dim3(256), 0, as_stream(stream), img, w, bias, coef4, bwd_coef, dg, argmax, H, W, tpw, nslots,
                           (double*)nullptr, dw_slots);
    if (int e = check_launch("stem_bwd_kernel<1>")) return e;
    hipLaunchKernelGGL(stem_dw_finalize_kernel, dim3((64 * ST_TB + 255) / 256), dim3(256), 0, as_stream(stream), dw_slots, nslots, grad, gbias);
    return check_launch("stem_dw_finalize_kernel");
}

}  // extern "C"
