// NYU data path on the device: the image work of the reference's per-sample loader -- Loader.crop,
// Loader.augment (recrop / rotate) and Loader.normalize (dataloader/loader.py:19-51, :75-179) -- batched, one workgroup per
// sample, bit-identical to the numpy restatement in awr_amd/nyu_data.py.
//
// Everything here is HBM-bound gather work (a 128 x 128 crop reads <= one window of a 480 x 640 uint16 frame and writes 64 KB):
// no MFMA, no GEMM reshaping.  What matters is (1) the crop tile lives in LDS between the crop and the warp, so the only HBM
// traffic of a sample is its frame window in and its normalised image out, (2) rows of the window are read by neighbouring lanes,
// (3) the arithmetic is the reference's: coordinates in IEEE double with separate multiplies and adds (this file is compiled with
// -ffp-contract=off), OpenCV's 1/32-pixel fixed point for the bilinear taps, float32 tap blending in numpy's association order.
#include <math.h>

#include "awr_common.h"

namespace awr {

constexpr int NYU_THREADS = 1024;
constexpr int INTER_BITS = 5, AB_BITS = 10;       // OpenCV imgwarp: INTER_TAB_SIZE = 32, affine increments in 2^-10 pixel

// ---- Loader.crop, one pixel (loader.py:19-51 via nyu_data.crop / bounds2crop / resize_nearest) ------------------------------
template <typename T>
__device__ __forceinline__ float crop_pixel(const T* __restrict__ frame, int fh, int fw, const awr_nyu_sample& s, int y, int x) {
    const int j = y - s.oy, i = x - s.ox;
    if (j < 0 || j >= s.rh || i < 0 || i >= s.rw) return 0.f;                  // res = zeros(dsize) outside the pasted window
    // resizeNN: min(floor(dst * ifx), src - 1) in doubles
    const int ys = min((int)floor((double)j * s.ify), s.ch - 1);
    const int xs = min((int)floor((double)i * s.ifx), s.cw - 1);
    const int fv = s.vstart + ys, fu = s.ustart + xs;
    float v = 0.f;                                                               // np.pad(..., constant_values = 0)
    if (fv >= 0 && fv < fh && fu >= 0 && fu < fw) v = (float)frame[(int64_t)fv * fw + fu];
    // cube clamp: float32 pixels against float64 bounds compare in float64; the assignment rounds zstart to float32
    const double d = (double)v;
    if (v != 0.f) {
        if (d < s.zstart) v = (float)s.zstart;
        else if (d > s.zend) v = 0.f;
    }
    return v;
}

// ---- cv2 INTER_LINEAR taps at 1/32-pixel fixed-point coordinates (nyu_data._bilinear_q5) ------------------------------------
template <typename Src>
__device__ __forceinline__ float bilinear_q5(const Src& src, int h, int w, int64_t X, int64_t Y, float border) {
    const int64_t sx = X >> INTER_BITS, sy = Y >> INTER_BITS;
    const float fx = (float)(X & 31) / 32.f, fy = (float)(Y & 31) / 32.f;
    const bool y0 = sy >= 0 && sy < h, y1 = sy + 1 >= 0 && sy + 1 < h;
    const bool x0 = sx >= 0 && sx < w, x1 = sx + 1 >= 0 && sx + 1 < w;
    const float t00 = (y0 && x0) ? src((int)sy, (int)sx) : border;
    const float t01 = (y0 && x1) ? src((int)sy, (int)sx + 1) : border;
    const float t10 = (y1 && x0) ? src((int)sy + 1, (int)sx) : border;
    const float t11 = (y1 && x1) ? src((int)sy + 1, (int)sx + 1) : border;
    const float top = t00 * (1.f - fx) + t01 * fx;
    const float bot = t10 * (1.f - fx) + t11 * fx;
    return top * (1.f - fy) + bot * fy;
}

// destination pixel -> fixed-point source coordinates
__device__ __forceinline__ void persp_coords(const double* m, int y, int x, int64_t& X, int64_t& Y) {
    const double xs = (double)x, ys = (double)y;
    double W = m[6] * xs + m[7] * ys + m[8];
    W = (W != 0.0) ? 32.0 / W : 0.0;
    const double fX = fmin(fmax((m[0] * xs + m[1] * ys + m[2]) * W, -2147483648.0), 2147483647.0);
    const double fY = fmin(fmax((m[3] * xs + m[4] * ys + m[5]) * W, -2147483648.0), 2147483647.0);
    X = (int64_t)rint(fX);
    Y = (int64_t)rint(fY);
}
__device__ __forceinline__ void affine_coords(const double* m, int y, int x, int64_t& X, int64_t& Y) {
    const double xs = (double)x, ys = (double)y, sc = (double)(1 << AB_BITS);
    const int64_t rd = (1 << AB_BITS) / 32 / 2;
    const int64_t ad = (int64_t)rint(m[0] * xs * sc), bd = (int64_t)rint(m[3] * xs * sc);
    const int64_t X0 = (int64_t)rint((m[1] * ys + m[2]) * sc) + rd, Y0 = (int64_t)rint((m[4] * ys + m[5]) * sc) + rd;
    X = (X0 + ad) >> (AB_BITS - INTER_BITS);
    Y = (Y0 + bd) >> (AB_BITS - INTER_BITS);
}

template <typename Src>
__device__ __forceinline__ float warp_pixel(const Src& src, int h, int w, const double* m, int op, int y, int x, float border) {
    int64_t X, Y;
    if (op == AWR_NYU_PERSPECTIVE) persp_coords(m, y, x, X, Y); else affine_coords(m, y, x, X, Y);
    return bilinear_q5(src, h, w, X, Y, border);
}

// ---- Loader.augment's image half after the resampler: fringe clean-up + cube clamp (recrop, loader.py:127-135) ---------------
__device__ __forceinline__ float recrop_cleanup(float v, float minpos, const awr_nyu_sample& s) {
    const float nv = minpos - 1.0f;                     // np.min(img[img > 0]) - 1, float32
    if (v < nv) v = 0.f;
    const double d = (double)v;
    if (v != 0.f) {
        if (d < s.zstart2) v = (float)s.zstart2;
        else if (d > s.zend2) v = 0.f;
    }
    return v;
}

// ---- Loader.normalize, one pixel (loader.py:88-101) -----------------------------------------------------------------------------
__device__ __forceinline__ float normalize_pixel(float v, float depth_max, const awr_nyu_sample& s) {
    const float farf = (float)s.far;
    if (v == depth_max) v = farf;
    if (v == 0.f) v = farf;
    if (s.norm32) {
        float f = fminf(fmaxf(v, (float)s.lo), farf);
        f = f - (float)s.center_z;
        return f / (float)s.half;
    }
    double d = fmin(fmax((double)v, s.lo), s.far);
    d = d - s.center_z;
    d = d / s.half;
    return (float)d;
}

struct GlobalSrc {
    const float* p; int w;
    __device__ __forceinline__ float operator()(int y, int x) const { return p[(int64_t)y * w + x]; }
};
struct LdsSrc {
    const float* p; int w;
    __device__ __forceinline__ float operator()(int y, int x) const { return p[y * w + x]; }
};

// max / smallest positive value over the workgroup; red: 2 * (NYU_THREADS / 64) floats of LDS
__device__ __forceinline__ void block_max_minpos(float& mx, float& mn, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        mn = fminf(mn, __shfl_xor(mn, o, 64));
    }
    const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wv] = mx; red[nw + wv] = mn; }
    __syncthreads();
    mx = red[0]; mn = red[nw];
    for (int i = 1; i < nw; ++i) { mx = fmaxf(mx, red[i]); mn = fminf(mn, red[nw + i]); }
}

template <typename T>
__global__ __launch_bounds__(NYU_THREADS) void nyu_crop_kernel(const T* __restrict__ frames, int fh, int fw,
                                                               const awr_nyu_sample* __restrict__ S, int dsize, float* __restrict__ crop,
                                                               float* __restrict__ stats) {
    __shared__ float red[2 * NYU_THREADS / 64];
    const awr_nyu_sample s = S[blockIdx.x];
    const T* frame = frames + s.frame * (int64_t)fh * fw;
    const int n = dsize * dsize;
    float mx = -INFINITY, mn = INFINITY;
    for (int p = threadIdx.x; p < n; p += NYU_THREADS) {
        const float v = crop_pixel(frame, fh, fw, s, p / dsize, p % dsize);
        crop[(int64_t)blockIdx.x * n + p] = v;
        mx = fmaxf(mx, v);
        if (v > 0.f) mn = fminf(mn, v);
    }
    block_max_minpos(mx, mn, red);
    if (threadIdx.x == 0) { stats[2 * blockIdx.x] = mx; stats[2 * blockIdx.x + 1] = mn; }
}

__global__ __launch_bounds__(256) void nyu_warp_kernel(const float* __restrict__ src, int sh, int sw, const double* __restrict__ m, int op,
                                                       float border, int dh, int dw, float* __restrict__ dst) {
    const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (p >= dh * dw) return;
    const GlobalSrc g{src + (int64_t)b * sh * sw, sw};
    dst[(int64_t)b * dh * dw + p] = warp_pixel(g, sh, sw, m + 9 * b, op, p / dw, p % dw, border);
}

__global__ __launch_bounds__(256) void nyu_normalize_kernel(const float* __restrict__ img, const float* __restrict__ depth_max,
                                                            const awr_nyu_sample* __restrict__ S, int64_t n, float* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= n) return;
    out[b * n + p] = normalize_pixel(img[b * n + p], depth_max[b], S[b]);
}

__global__ __launch_bounds__(256) void nyu_augment_kernel(const float* __restrict__ crop, const float* __restrict__ stats,
                                                          const awr_nyu_sample* __restrict__ S, int dsize, float* __restrict__ out,
                                                          int* __restrict__ status) {
    const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    const int n = dsize * dsize;
    if (p >= n) return;
    const awr_nyu_sample& s = S[b];
    const float mx = stats[2 * b], mn = stats[2 * b + 1];
    const GlobalSrc g{crop + (int64_t)b * n, dsize};
    float v;
    if (s.op == AWR_NYU_NONE) v = g.p[p];
    else {
        v = warp_pixel(g, dsize, dsize, s.m, s.op, p / dsize, p % dsize, 0.f);
        if (s.op == AWR_NYU_PERSPECTIVE) v = recrop_cleanup(v, mn, s);
    }
    out[(int64_t)b * n + p] = normalize_pixel(v, mx, s);
    if (p == 0 && status) status[b] = (s.op == AWR_NYU_PERSPECTIVE && !(mn < INFINITY)) ? 1 : 0;
}

// The production kernel: crop -> LDS tile -> (warp + clean-up) -> normalize -> HBM, one workgroup per sample.
template <typename T>
__global__ __launch_bounds__(NYU_THREADS) void nyu_batch_kernel(const T* __restrict__ frames, int fh, int fw,
                                                                const awr_nyu_sample* __restrict__ S, int dsize, float* __restrict__ out,
                                                                int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n = dsize * dsize;
    float* tile = lds;
    float* red = lds + n;
    const awr_nyu_sample s = S[blockIdx.x];
    const T* frame = frames + s.frame * (int64_t)fh * fw;
    float mx = -INFINITY, mn = INFINITY;
    for (int p = threadIdx.x; p < n; p += NYU_THREADS) {
        const float v = crop_pixel(frame, fh, fw, s, p / dsize, p % dsize);
        tile[p] = v;
        mx = fmaxf(mx, v);
        if (v > 0.f) mn = fminf(mn, v);
    }
    block_max_minpos(mx, mn, red);          // its barrier also publishes the tile
    float* o = out + (int64_t)blockIdx.x * n;
    const LdsSrc t{tile, dsize};
    if (s.op == AWR_NYU_NONE) {
        for (int p = threadIdx.x; p < n; p += NYU_THREADS) o[p] = normalize_pixel(tile[p], mx, s);
    } else if (s.op == AWR_NYU_PERSPECTIVE) {
        for (int p = threadIdx.x; p < n; p += NYU_THREADS) {
            const float v = recrop_cleanup(warp_pixel(t, dsize, dsize, s.m, AWR_NYU_PERSPECTIVE, p / dsize, p % dsize, 0.f), mn, s);
            o[p] = normalize_pixel(v, mx, s);
        }
    } else {
        for (int p = threadIdx.x; p < n; p += NYU_THREADS)
            o[p] = normalize_pixel(warp_pixel(t, dsize, dsize, s.m, AWR_NYU_AFFINE, p / dsize, p % dsize, 0.f), mx, s);
    }
    if (threadIdx.x == 0 && status) status[blockIdx.x] = (s.op == AWR_NYU_PERSPECTIVE && !(mn < INFINITY)) ? 1 : 0;
}

constexpr int NYU_LDS_MAX = 160 * 1024;

static int check_common(const void* frames, int frame_type, int fh, int fw, const awr_nyu_sample* samples, int B, int dsize) {
    AWR_REQUIRE(frames && samples, "awr_nyu: NULL frame store / sample table");
    AWR_REQUIRE(frame_type == AWR_NYU_U16 || frame_type == AWR_NYU_F32, "awr_nyu: frame_type must be AWR_NYU_U16 or AWR_NYU_F32 (got %d)", frame_type);
    AWR_REQUIRE(fh > 0 && fw > 0 && B > 0 && dsize > 0 && dsize <= 4096, "awr_nyu: bad sizes (frame %d x %d, B = %d, dsize = %d)", fh, fw, B, dsize);
    return AWR_OK;
}

}  // namespace awr

using namespace awr;

extern "C" {

int awr_nyu_crop(const void* frames, int frame_type, int fh, int fw, const awr_nyu_sample* samples, int B, int dsize, float* crop,
                 float* stats, void* stream) {
    if (int rc = check_common(frames, frame_type, fh, fw, samples, B, dsize)) return rc;
    AWR_REQUIRE(crop && stats, "awr_nyu_crop: NULL output");
    if (frame_type == AWR_NYU_U16)
        nyu_crop_kernel<uint16_t><<<B, NYU_THREADS, 0, as_stream(stream)>>>((const uint16_t*)frames, fh, fw, samples, dsize, crop, stats);
    else
        nyu_crop_kernel<float><<<B, NYU_THREADS, 0, as_stream(stream)>>>((const float*)frames, fh, fw, samples, dsize, crop, stats);
    return check_launch("awr_nyu_crop");
}

int awr_nyu_warp(const float* src, int sh, int sw, const double* m, int op, float border, int B, int dh, int dw, float* dst, void* stream) {
    AWR_REQUIRE(src && m && dst, "awr_nyu_warp: NULL pointer");
    AWR_REQUIRE(op == AWR_NYU_PERSPECTIVE || op == AWR_NYU_AFFINE, "awr_nyu_warp: op must be AWR_NYU_PERSPECTIVE or AWR_NYU_AFFINE (got %d)", op);
    AWR_REQUIRE(sh > 0 && sw > 0 && dh > 0 && dw > 0 && B > 0 && B <= 65535, "awr_nyu_warp: bad sizes");
    nyu_warp_kernel<<<dim3((dh * dw + 255) / 256, B), 256, 0, as_stream(stream)>>>(src, sh, sw, m, op, border, dh, dw, dst);
    return check_launch("awr_nyu_warp");
}

int awr_nyu_normalize(const float* img, const float* depth_max, const awr_nyu_sample* samples, int B, int64_t n, float* out, void* stream) {
    AWR_REQUIRE(img && depth_max && samples && out, "awr_nyu_normalize: NULL pointer");
    AWR_REQUIRE(B > 0 && B <= 65535 && n > 0, "awr_nyu_normalize: bad sizes");
    nyu_normalize_kernel<<<dim3((unsigned)((n + 255) / 256), B), 256, 0, as_stream(stream)>>>(img, depth_max, samples, n, out);
    return check_launch("awr_nyu_normalize");
}

int awr_nyu_augment(const float* crop, const float* stats, const awr_nyu_sample* samples, int B, int dsize, float* out, int* status,
                    void* stream) {
    AWR_REQUIRE(crop && stats && samples && out, "awr_nyu_augment: NULL pointer");
    AWR_REQUIRE(B > 0 && B <= 65535 && dsize > 0 && dsize <= 4096, "awr_nyu_augment: bad sizes");
    nyu_augment_kernel<<<dim3((dsize * dsize + 255) / 256, B), 256, 0, as_stream(stream)>>>(crop, stats, samples, dsize, out, status);
    return check_launch("awr_nyu_augment");
}

int awr_nyu_batch(const void* frames, int frame_type, int fh, int fw, const awr_nyu_sample* samples, int B, int dsize, float* out,
                  int* status, float* scratch, void* stream) {
    if (int rc = check_common(frames, frame_type, fh, fw, samples, B, dsize)) return rc;
    AWR_REQUIRE(out, "awr_nyu_batch: NULL output");
    const size_t lds = ((size_t)dsize * dsize + 2 * NYU_THREADS / 64) * sizeof(float);
    if (lds > (size_t)NYU_LDS_MAX) {
        AWR_REQUIRE(scratch, "awr_nyu_batch: a %d x %d crop does not fit in LDS; pass scratch (B * (dsize * dsize + 2) floats)", dsize, dsize);
        float* stats = scratch + (int64_t)B * dsize * dsize;
        if (int rc = awr_nyu_crop(frames, frame_type, fh, fw, samples, B, dsize, scratch, stats, stream)) return rc;
        return awr_nyu_augment(scratch, stats, samples, B, dsize, out, status, stream);
    }
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e1 = hipFuncSetAttribute((const void*)nyu_batch_kernel<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, NYU_LDS_MAX);
        hipError_t e2 = hipFuncSetAttribute((const void*)nyu_batch_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, NYU_LDS_MAX);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            set_error("awr_nyu_batch: hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
            return AWR_ERR_HIP;
        }
        attr_done = true;
    }
    if (frame_type == AWR_NYU_U16)
        nyu_batch_kernel<uint16_t><<<B, NYU_THREADS, lds, as_stream(stream)>>>((const uint16_t*)frames, fh, fw, samples, dsize, out, status);
    else
        nyu_batch_kernel<float><<<B, NYU_THREADS, lds, as_stream(stream)>>>((const float*)frames, fh, fw, samples, dsize, out, status);
    return check_launch("awr_nyu_batch");
}

}  // extern "C"
