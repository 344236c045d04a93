// Round-6 study: fused Winograd F(2x2, 3x3) on the FP32 matrix pipe for stride-1 3x3 convolutions (the forward and,
// with mirrored weights, the data gradient of model/resnet_deconv.py:139-142,:161-165 BasicBlock convs and model/hourglass.py:35 Residual
// conv2).  2.25x fewer multiplies than the direct implicit GEMM: per 2x2 output patch and channel pair 16 products instead of 36.
//
//   U[pos][c][n] = (G g G^T)[pos]           weights, transformed once per optimiser step (wino_weight_kernel)
//   V[pos][patch][c] = (B^T d B)[pos]       the 4x4 input window of the patch, transformed while it is staged into LDS
//   M[pos][patch][n] = sum_c V * U          16 independent GEMMs (one per position) on v_mfma_f32_32x32x2_f32
//   Y[patch] = A^T M A                      2x2 outputs, after an exchange of the accumulators through LDS
//
// Three generations live here (profiles/r06_winograd.txt has the numbers): wino_fwd_kernel (v1: every patch gathers its own 4x4 window from global
// memory -- kept for the measurements), wino2_fwd_kernel<.., 32> (v2: the RAW input tile of a 2-D block of 64 patches goes through LDS once per K stage;
// 8 waves x 2 positions, two workgroups per CU) and wino2_fwd_kernel<.., 64> (v2 wide: 64 output channels per workgroup, 16 waves x 1 position, one
// workgroup per CU -- 174 / 188 algorithmic TFLOP/s on 128 -> 128 @ 64 x 64 / 256 -> 256 @ 16 x 16).  Plans use v2 (awr_wino_conv) for the forward and,
// with the mirrored transform, for the data gradients of eligible layers (awr_set_conv_winograd); tests/test_wino_gpu.py, tools/microbench_wino.py.
#include <stdlib.h>
#include <string.h>

#include "awr_common.h"

namespace awr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int W_TP = 64, W_TN = 32;

struct wino_args {
    const float* in;      // (B, H, W, C) NHWC
    const float* U;       // [16][C][N]
    const float* bias;    // [N] or null
    float* out;           // (B, H, W, N)
    int B, H, W, C, N, relu;
};

// w: OIHW (N, C, 3, 3); mirror != 0: the data-gradient form (taps mirrored, roles of N and C swapped by the caller's indexing)
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, int N, int C, int Npad, int Cpad, int mirror, float* __restrict__ U) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cpad * Npad) return;
    const int c = idx / Npad, n = idx % Npad;
    float g[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float v = 0.f;
            if (n < N && c < C) v = mirror ? w[((int64_t)c * N + n) * 9 + (2 - i) * 3 + (2 - j)] : w[((int64_t)n * C + c) * 9 + i * 3 + j];
            g[i][j] = v;
        }
    float t[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        t[0][j] = g[0][j];
        t[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
        t[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
        t[3][j] = g[2][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float u0 = t[i][0], u1 = 0.5f * (t[i][0] + t[i][1] + t[i][2]), u2 = 0.5f * (t[i][0] - t[i][1] + t[i][2]), u3 = t[i][2];
        U[((int64_t)(i * 4 + 0) * Cpad + c) * Npad + n] = u0;
        U[((int64_t)(i * 4 + 1) * Cpad + c) * Npad + n] = u1;
        U[((int64_t)(i * 4 + 2) * Cpad + c) * Npad + n] = u2;
        U[((int64_t)(i * 4 + 3) * Cpad + c) * Npad + n] = u3;
    }
}

template <int KB, int NW, int PROBE = 0>      // NW waves per workgroup, 16 / NW positions per wave; PROBE (timing only, wrong results): 1 = no input
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 4 : 2))) void wino_fwd_kernel(const wino_args a) {
    constexpr int NT = 64 * NW, QP = 16 / NW;
    constexpr int VROW = KB + 1;                       // odd pitch: conflict-free A-fragment reads
    constexpr int V_FLOATS = 16 * W_TP * VROW, U_FLOATS = 16 * KB * W_TN;
    constexpr int X_FLOATS = 16 * 32 * 32;             // epilogue exchange: 16 positions x 32 patches x 32 channels
    constexpr int LDS_FLOATS = (V_FLOATS + U_FLOATS) > X_FLOATS ? (V_FLOATS + U_FLOATS) : X_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float* const Vs = lds;
    float* const Us = lds + V_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int PH = a.H >> 1, PW = a.W >> 1;
    const int64_t P_total = (int64_t)a.B * PH * PW;
    const int64_t p0 = (int64_t)blockIdx.x * W_TP;
    const int n0 = blockIdx.y * W_TN;

    // ---- input-transform items of this thread: channel tid % KB of patches tid / KB + (256 / KB) * pass ----
    constexpr int PPP = NT / KB < W_TP ? NT / KB : W_TP, PASSES = W_TP / PPP;     // patches per pass, passes per stage
    const bool titem = tid < PPP * KB;                     // (KB = 4 with 512 threads: half of them have no transform item)
    const int ch = tid % KB, psub = tid / KB;
    int64_t base[PASSES];
    unsigned vmask[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int64_t p = p0 + psub + PPP * ps;
        vmask[ps] = 0;
        base[ps] = 0;
        if (p < P_total) {
            const int b = (int)(p / (PH * PW)), rem = (int)(p % (PH * PW)), py = rem / PW, px = rem % PW;
            const int y0 = 2 * py - 1, x0 = 2 * px - 1;
            base[ps] = (((int64_t)b * a.H + y0) * a.W + x0) * a.C;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int y = y0 + (t >> 2), x = x0 + (t & 3);
                if (y >= 0 && y < a.H && x >= 0 && x < a.W) vmask[ps] |= 1u << t;
            }
        }
    }
    float d[PASSES][16];
    constexpr int UQ = U_FLOATS / 4 / NT > 0 ? U_FLOATS / 4 / NT : 1;      // float4 weight loads per thread and stage
    const bool uitem = tid < U_FLOATS / 4;
    float4 ureg[UQ];

    auto load_stage = [&](int k0) {
        if (PROBE >= 3 && k0) return;                       // probe: nothing travels after the first stage
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            if (PROBE == 1 && k0) break;                    // probe: the input window is loaded once
#pragma unroll
            for (int t = 0; t < 16; ++t)
                d[ps][t] = (titem && ((vmask[ps] >> t) & 1u)) ? a.in[base[ps] + ((int64_t)(t >> 2) * a.W + (t & 3)) * a.C + k0 + ch] : 0.f;
        }
        if (PROBE == 2 && k0) return;                       // probe: the weights are loaded once
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
            const int f = tid + NT * q, n4 = f % (W_TN / 4), k = (f / (W_TN / 4)) % KB, pos = f / ((W_TN / 4) * KB);
            if (uitem) ureg[q] = ld4(a.U + ((int64_t)pos * a.C + k0 + k) * a.N + n0 + 4 * n4);
        }
    };
    auto store_stage = [&]() {
        if (PROBE == 4) return;                             // probe: no transform, no LDS writes either (MFMA + fragment reads + barriers only)
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            if (!titem) break;
            float t[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[0][j] = d[ps][0 + j] - d[ps][8 + j];
                t[1][j] = d[ps][4 + j] + d[ps][8 + j];
                t[2][j] = d[ps][8 + j] - d[ps][4 + j];
                t[3][j] = d[ps][4 + j] - d[ps][12 + j];
            }
            float* v = Vs + (psub + PPP * ps) * VROW + ch;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[(i * 4 + 0) * W_TP * VROW] = t[i][0] - t[i][2];
                v[(i * 4 + 1) * W_TP * VROW] = t[i][1] + t[i][2];
                v[(i * 4 + 2) * W_TP * VROW] = t[i][2] - t[i][1];
                v[(i * 4 + 3) * W_TP * VROW] = t[i][1] - t[i][3];
            }
        }
#pragma unroll
        for (int q = 0; q < UQ; ++q)
            if (uitem) st4(Us + (tid + NT * q) * 4, ureg[q]);
    };

    f32x16 acc[QP][2];
#pragma unroll
    for (int q = 0; q < QP; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][h][r] = 0.f;

    load_stage(0);
    if (PROBE != 4) store_stage();
    __syncthreads();
    for (int k0 = 0; k0 < a.C; k0 += KB) {
        const bool more = k0 + KB < a.C;
        if (more) load_stage(k0 + KB);                 // the next stage's operands travel while this stage multiplies
#pragma unroll
        for (int s = 0; s < KB / 2; ++s)
#pragma unroll
            for (int q = 0; q < QP; ++q) {
                const int pos = QP * wave + q;
                const float b = Us[(pos * KB + 2 * s + half) * W_TN + l31];
                const float a0 = Vs[(pos * W_TP + l31) * VROW + 2 * s + half];
                const float a1 = Vs[(pos * W_TP + 32 + l31) * VROW + 2 * s + half];
                acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[q][0], 0, 0, 0);
                acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[q][1], 0, 0, 0);
            }
        __syncthreads();                               // everyone is done reading the stage
        if (more) {
            store_stage();
            __syncthreads();
        }
    }

    // ---- epilogue: exchange through LDS (32 patches at a time), output transform, bias / ReLU, 16-byte stores ----
    // item = (patch of the half, channel quad, output row): 512 items per half
    float* const X = lds;
    const int n4 = tid & 7, pi = (tid >> 3) & 31;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h) __syncthreads();
#pragma unroll
        for (int q = 0; q < QP; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) X[((QP * wave + q) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[q][h][r];
        __syncthreads();
        const int64_t p = p0 + 32 * h + pi;
        for (int rr = tid >> 8; rr < 2; rr += NT / 256) {          // output row of the patch (both rows for a 256-thread workgroup)
            if (p >= P_total || n0 + 4 * n4 >= a.N) break;
            float4 s[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 m1 = ld4(X + ((4 + j) * 32 + pi) * 32 + 4 * n4), m2 = ld4(X + ((8 + j) * 32 + pi) * 32 + 4 * n4);
                const float4 m03 = ld4(X + ((rr ? 12 + j : j) * 32 + pi) * 32 + 4 * n4);
                if (rr == 0) s[j] = make_float4(m03.x + m1.x + m2.x, m03.y + m1.y + m2.y, m03.z + m1.z + m2.z, m03.w + m1.w + m2.w);
                else s[j] = make_float4(m1.x - m2.x - m03.x, m1.y - m2.y - m03.y, m1.z - m2.z - m03.z, m1.w - m2.w - m03.w);
            }
            float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias) bs = ld4(a.bias + n0 + 4 * n4);
            float4 y0 = make_float4(s[0].x + s[1].x + s[2].x + bs.x, s[0].y + s[1].y + s[2].y + bs.y, s[0].z + s[1].z + s[2].z + bs.z, s[0].w + s[1].w + s[2].w + bs.w);
            float4 y1 = make_float4(s[1].x - s[2].x - s[3].x + bs.x, s[1].y - s[2].y - s[3].y + bs.y, s[1].z - s[2].z - s[3].z + bs.z, s[1].w - s[2].w - s[3].w + bs.w);
            if (a.relu) {
                y0.x = fmaxf(y0.x, 0.f); y0.y = fmaxf(y0.y, 0.f); y0.z = fmaxf(y0.z, 0.f); y0.w = fmaxf(y0.w, 0.f);
                y1.x = fmaxf(y1.x, 0.f); y1.y = fmaxf(y1.y, 0.f); y1.z = fmaxf(y1.z, 0.f); y1.w = fmaxf(y1.w, 0.f);
            }
            const int b = (int)(p / (PH * PW)), rem = (int)(p % (PH * PW)), py = rem / PW, px = rem % PW;
            float* o = a.out + (((int64_t)b * a.H + 2 * py + rr) * a.W + 2 * px) * a.N + n0 + 4 * n4;
            st4(o, y0);
            st4(o + a.N, y1);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// v2: the RAW input tile goes through LDS.  v1's probes (profiles/r06_winograd.txt): its MFMA loop alone runs at 135 TF executed (303
// algorithmic), but the 4x4 windows of neighbouring patches overlap -- every input pixel was requested four times, 32 bytes at a time, and those
// requests cost 44 % of the launch.  Here a tile is a 2-D block of patches of ONE image (or whole small images), its input region is loaded
// ONCE per stage with 16-byte requests into LDS, and the transform reads its windows from there.
// ---------------------------------------------------------------------------------------------------------------------------------------
struct wino2_args {
    const float* in;
    const float* U;
    const float* bias;
    const float* in_scale;   // optional fused input affine (+ ReLU): the previous BatchNorm, applied to valid pixels only (padding stays zero)
    const float* in_shift;
    const float* out_scale;  // optional per-output-channel affine after the bias (a folded eval-mode BatchNorm): (acc + bias) * scale + shift, then + res, then ReLU
    const float* out_shift;
    float* out;
    double* stats;           // optional: per-channel sum / sum of squares of the stored output, [nslots][2][N] (awr_bn_finalize's layout)
    // data-gradient epilogue (awr_conv_args.res / bnr_y / bnr_coef / bnr_act of the direct kernel): out = mask(acc + res), stats += (sum g, sum g * xhat)
    const float* res;
    const float* bnr_y;
    const float* bnr_coef;   // [scale | shift | mean | invstd][N]
    const float* bnr_act;
    int B, H, W, C, N, relu, relu_in, nslots;
    int PRt, PCt, nimg;      // tile: nimg images x PRt x PCt patches = 64
};

// RITEMS: 16-byte raw-tile requests per thread and stage; TN: output channels per workgroup (32: 8 waves x 2 positions, two workgroups per CU;
// 64: 16 waves x 1 position, one workgroup per CU -- every transformed window feeds twice as many multiplies)
template <int KB, int RITEMS, int TN = 32>
__global__ __launch_bounds__(TN == 64 ? 1024 : 512) __attribute__((amdgpu_waves_per_eu(4))) void wino2_fwd_kernel(const wino2_args a) {
    // (a k-pair-interleaved layout with 8-byte LDS accesses -- Vs[pos][k / 2][patch][2], Raw[pixel][KB + 2], 256 transform items of two
    // channels -- was built and measured SLOWER: 568 vs 528 us on 128 -> 128 @ 64 x 64 x 64; profiles/r06_winograd.txt)
    constexpr int NT = TN == 64 ? 1024 : 512, QP = TN == 64 ? 1 : 2, NTT = TN / 32;      // threads, positions per wave, 32-channel tiles per position
    constexpr int W_TN = TN;
    constexpr int VROW = KB + 1, RP = KB + 1;          // odd pitches: conflict-free A-fragment reads, 2-way window reads
    constexpr int V_FLOATS = 16 * W_TP * VROW, U_FLOATS = 16 * KB * W_TN;
    constexpr int RAW_MAX_PX = 576;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Vs = lds;
    float* const Us = lds + V_FLOATS;
    float* const Raw = lds + V_FLOATS + U_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int PH = a.H >> 1, PW = a.W >> 1;
    const int RH = 2 * a.PRt + 2, RW = 2 * a.PCt + 2;
    // 1-D grid, XCD-aware: the hardware deals consecutive workgroups round-robin over the 8 XCDs (each with its own L2); the remap gives every XCD
    // a contiguous run of logical ids, and the channel tile varies fastest -- the N / 32 workgroups that read the SAME input tile run on one XCD,
    // back to back, so the tile comes from HBM once
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int ntiles_n = a.N / W_TN, tile = logical / ntiles_n;
    const int n0 = (logical % ntiles_n) * W_TN;
    // tile -> (first image, first patch row, first patch column)
    const int tiles_x = PW / a.PCt, tiles_y = PH / a.PRt;
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, tb = tile / (tiles_x * tiles_y);
    const int img0 = tb * a.nimg, pr0 = ty * a.PRt, pc0 = tx * a.PCt;

    // ---- raw-tile items of this thread (fixed over the K loop): pixel of the region x channel quad ----
    constexpr int QUADS = KB / 4;
    static_assert(RAW_MAX_PX * QUADS <= 3 * NT, "raw tile");
    const int nraw = a.nimg * RH * RW * QUADS;
    int roff[RITEMS];              // element offsets (the host checks that the tensor has < 2^31 elements); < 0: padding / outside the batch
    int rlds[RITEMS];
#pragma unroll
    for (int r = 0; r < RITEMS; ++r) {
        const int i = tid + NT * r, px = i / QUADS, quad = i % QUADS;
        rlds[r] = i < nraw ? px * RP + 4 * quad : -1;
        const int il = px / (RH * RW), rr = (px / RW) % RH, cc = px % RW;
        const int b = img0 + il, y = 2 * pr0 - 1 + rr, x = 2 * pc0 - 1 + cc;
        const bool ok = i < nraw && b < a.B && y >= 0 && y < a.H && x >= 0 && x < a.W;
        roff[r] = ok ? ((b * a.H + y) * a.W + x) * a.C + 4 * quad : -1;
    }
    // ---- transform item: patch tid / KB, channel tid % KB ----
    const bool titem = tid < W_TP * KB;                                      // (512 items: the wide form's upper half of the workgroup has none)
    const int ch = tid % KB, lp = (tid / KB) % W_TP;
    const int lil = lp / (a.PRt * a.PCt), lpr = (lp / a.PCt) % a.PRt, lpc = lp % a.PCt;
    const int tbase = ((lil * RH + 2 * lpr) * RW + 2 * lpc) * RP + ch;
    constexpr int UQ = U_FLOATS / 4 / NT;
    float4 rreg[RITEMS], ureg[UQ];

    auto load_stage = [&](int k0) {
        // (the channel quad of a thread's raw items is the same for all of them: NT is a multiple of QUADS -> one scale / shift pair per stage)
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.in_scale) {
            sc = ld4(a.in_scale + k0 + 4 * (tid % QUADS));
            sh = ld4(a.in_shift + k0 + 4 * (tid % QUADS));
        }
#pragma unroll
        for (int r = 0; r < RITEMS; ++r) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (roff[r] >= 0) {
                v = ld4(a.in + roff[r] + k0);
                if (a.in_scale) v = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
                if (a.relu_in) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            }
            rreg[r] = v;
        }
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
            const int f = tid + NT * q, n4 = f % (W_TN / 4), k = (f / (W_TN / 4)) % KB, pos = f / ((W_TN / 4) * KB);
            ureg[q] = ld4(a.U + ((int64_t)pos * a.C + k0 + k) * a.N + n0 + 4 * n4);
        }
    };
    auto store_raw = [&]() {
#pragma unroll
        for (int r = 0; r < RITEMS; ++r)
            if (rlds[r] >= 0) {
                float* p = Raw + rlds[r];
                p[0] = rreg[r].x; p[1] = rreg[r].y; p[2] = rreg[r].z; p[3] = rreg[r].w;
            }
    };
    auto store_u = [&]() {
#pragma unroll
        for (int q = 0; q < UQ; ++q) st4(Us + (tid + NT * q) * 4, ureg[q]);
    };
    auto transform = [&]() {
        if (!titem) return;
        float d[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) d[t] = Raw[tbase + ((t >> 2) * RW + (t & 3)) * RP];
        float t4[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t4[0][j] = d[0 + j] - d[8 + j];
            t4[1][j] = d[4 + j] + d[8 + j];
            t4[2][j] = d[8 + j] - d[4 + j];
            t4[3][j] = d[4 + j] - d[12 + j];
        }
        float* v = Vs + lp * VROW + ch;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[(i * 4 + 0) * W_TP * VROW] = t4[i][0] - t4[i][2];
            v[(i * 4 + 1) * W_TP * VROW] = t4[i][1] + t4[i][2];
            v[(i * 4 + 2) * W_TP * VROW] = t4[i][2] - t4[i][1];
            v[(i * 4 + 3) * W_TP * VROW] = t4[i][1] - t4[i][3];
        }
    };

    f32x16 acc[QP][2][NTT];
#pragma unroll
    for (int q = 0; q < QP; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int jt = 0; jt < NTT; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][h][jt][r] = 0.f;

    load_stage(0);
    store_raw();
    store_u();
    __syncthreads();
    transform();
    __syncthreads();
    for (int k0 = 0; k0 < a.C; k0 += KB) {
        const bool more = k0 + KB < a.C;
        if (more) load_stage(k0 + KB);
#pragma unroll
        for (int s = 0; s < KB / 2; ++s)
#pragma unroll
            for (int q = 0; q < QP; ++q) {
                const int pos = QP * wave + q;
                const float a0 = Vs[(pos * W_TP + l31) * VROW + 2 * s + half];
                const float a1 = Vs[(pos * W_TP + 32 + l31) * VROW + 2 * s + half];
#pragma unroll
                for (int jt = 0; jt < NTT; ++jt) {
                    const float b = Us[(pos * KB + 2 * s + half) * W_TN + 32 * jt + l31];
                    acc[q][0][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[q][0][jt], 0, 0, 0);
                    acc[q][1][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[q][1][jt], 0, 0, 0);
                }
            }
        if (more) store_raw();            // (nobody reads Raw now: the previous transform finished before the previous barrier)
        __syncthreads();                  // everyone is done reading Vs / Us
        if (more) {
            store_u();
            transform();
            __syncthreads();
        }
    }

    // ---- epilogue: exchange through LDS (32 patches at a time), output transform, bias / ReLU / statistics, 16-byte stores ----
    float* const X = lds;
    const int n4 = tid & 7, pi = (tid >> 3) & 31, rr = (tid >> 8) & 1;
    const bool oitem = tid < 512;                      // 32 patches x 8 channel quads x 2 output rows per exchange round
    // statistics in double from the first product on: sum x^2 of a nearly constant channel (std << |mean|) loses its variance to the rounding of x^2
    // in fp32 (the direct kernel sums x - c for the same reason, csrc/awr_conv_kernels.inc)
    double ssum[NTT][4], ssq[NTT][4];
#pragma unroll
    for (int jt = 0; jt < NTT; ++jt)
#pragma unroll
        for (int e = 0; e < 4; ++e) ssum[jt][e] = ssq[jt][e] = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int jt = 0; jt < NTT; ++jt) {
        if (h || jt) __syncthreads();
#pragma unroll
        for (int q = 0; q < QP; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) X[((QP * wave + q) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[q][h][jt][r];
        __syncthreads();
        const int p = 32 * h + pi;
        const int il = p / (a.PRt * a.PCt), pr = (p / a.PCt) % a.PRt, pc = p % a.PCt;
        const int b = img0 + il;
        const int nq = n0 + 32 * jt + 4 * n4;          // first of this thread's four output channels
        if (oitem && b < a.B) {
            float4 s[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 m1 = ld4(X + ((4 + j) * 32 + pi) * 32 + 4 * n4), m2 = ld4(X + ((8 + j) * 32 + pi) * 32 + 4 * n4);
                const float4 m03 = ld4(X + ((rr ? 12 + j : j) * 32 + pi) * 32 + 4 * n4);
                if (rr == 0) s[j] = make_float4(m03.x + m1.x + m2.x, m03.y + m1.y + m2.y, m03.z + m1.z + m2.z, m03.w + m1.w + m2.w);
                else s[j] = make_float4(m1.x - m2.x - m03.x, m1.y - m2.y - m03.y, m1.z - m2.z - m03.z, m1.w - m2.w - m03.w);
            }
            float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias) bs = ld4(a.bias + nq);
            float4 y0 = make_float4(s[0].x + s[1].x + s[2].x + bs.x, s[0].y + s[1].y + s[2].y + bs.y, s[0].z + s[1].z + s[2].z + bs.z, s[0].w + s[1].w + s[2].w + bs.w);
            float4 y1 = make_float4(s[1].x - s[2].x - s[3].x + bs.x, s[1].y - s[2].y - s[3].y + bs.y, s[1].z - s[2].z - s[3].z + bs.z, s[1].w - s[2].w - s[3].w + bs.w);
            if (a.out_scale) {          // folded eval-mode BatchNorm (inference plans)
                const float4 osc = ld4(a.out_scale + nq), osh = ld4(a.out_shift + nq);
                y0 = make_float4(y0.x * osc.x + osh.x, y0.y * osc.y + osh.y, y0.z * osc.z + osh.z, y0.w * osc.w + osh.w);
                y1 = make_float4(y1.x * osc.x + osh.x, y1.y * osc.y + osh.y, y1.z * osc.z + osh.z, y1.w * osc.w + osh.w);
            }
            const int64_t ooff = (((int64_t)b * a.H + 2 * (pr0 + pr) + rr) * a.W + 2 * (pc0 + pc)) * a.N + nq;
            if (a.res) {
                const float4 r0 = ld4(a.res + ooff), r1 = ld4(a.res + ooff + a.N);
                y0.x += r0.x; y0.y += r0.y; y0.z += r0.z; y0.w += r0.w;
                y1.x += r1.x; y1.y += r1.y; y1.z += r1.z; y1.w += r1.w;
            }
            if (a.bnr_y) {
                // the value is the gradient w.r.t. relu(bn(y) [+ residual]): mask it with the re-derived ReLU (or the stored activation's sign) and reduce
                // sum g, sum g * xhat for that BatchNorm's backward -- the EM 3 / 4 epilogues of the direct kernel (csrc/awr_conv_kernels.inc)
                const int nn = nq;
                const float4 ksc = ld4(a.bnr_coef + nn), ksh = ld4(a.bnr_coef + a.N + nn), kmu = ld4(a.bnr_coef + 2 * a.N + nn), kis = ld4(a.bnr_coef + 3 * a.N + nn);
                const float4 q0 = ld4(a.bnr_y + ooff), q1 = ld4(a.bnr_y + ooff + a.N);
                if (a.bnr_act) {
                    const float4 a0 = ld4(a.bnr_act + ooff), a1 = ld4(a.bnr_act + ooff + a.N);
                    y0.x = a0.x > 0.f ? y0.x : 0.f; y0.y = a0.y > 0.f ? y0.y : 0.f; y0.z = a0.z > 0.f ? y0.z : 0.f; y0.w = a0.w > 0.f ? y0.w : 0.f;
                    y1.x = a1.x > 0.f ? y1.x : 0.f; y1.y = a1.y > 0.f ? y1.y : 0.f; y1.z = a1.z > 0.f ? y1.z : 0.f; y1.w = a1.w > 0.f ? y1.w : 0.f;
                } else {
                    y0.x = q0.x * ksc.x + ksh.x > 0.f ? y0.x : 0.f; y0.y = q0.y * ksc.y + ksh.y > 0.f ? y0.y : 0.f;
                    y0.z = q0.z * ksc.z + ksh.z > 0.f ? y0.z : 0.f; y0.w = q0.w * ksc.w + ksh.w > 0.f ? y0.w : 0.f;
                    y1.x = q1.x * ksc.x + ksh.x > 0.f ? y1.x : 0.f; y1.y = q1.y * ksc.y + ksh.y > 0.f ? y1.y : 0.f;
                    y1.z = q1.z * ksc.z + ksh.z > 0.f ? y1.z : 0.f; y1.w = q1.w * ksc.w + ksh.w > 0.f ? y1.w : 0.f;
                }
                const float* g0 = &y0.x;
                const float* g1 = &y1.x;
                const float* t0 = &q0.x;
                const float* t1 = &q1.x;
                const float* mu = &kmu.x;
                const float* is = &kis.x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ssum[jt][e] += (double)g0[e] + (double)g1[e];
                    ssq[jt][e] += (double)(g0[e] * ((t0[e] - mu[e]) * is[e])) + (double)(g1[e] * ((t1[e] - mu[e]) * is[e]));
                }
            }
            if (a.relu) {
                y0.x = fmaxf(y0.x, 0.f); y0.y = fmaxf(y0.y, 0.f); y0.z = fmaxf(y0.z, 0.f); y0.w = fmaxf(y0.w, 0.f);
                y1.x = fmaxf(y1.x, 0.f); y1.y = fmaxf(y1.y, 0.f); y1.z = fmaxf(y1.z, 0.f); y1.w = fmaxf(y1.w, 0.f);
            }
            if (a.stats && !a.bnr_y) {
                const float* p0 = &y0.x;
                const float* p1 = &y1.x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double u = (double)p0[e], v = (double)p1[e];
                    ssum[jt][e] += u + v;
                    ssq[jt][e] += u * u + v * v;
                }
            }
            float* o = a.out + ooff;
            st4(o, y0);
            st4(o + a.N, y1);
        }
    }
    if (a.stats) {
        // lanes with the same channel quad (tid & 7) of one wave hold 8 patches: reduce over them in the wave, combine the eight participating waves through
        // LDS, then ONE fp64 atomic per channel and statistic (the exchange tile is dead: reuse it)
        __syncthreads();
        double* red = reinterpret_cast<double*>(lds);      // [wave 0..7][jt][stat][channel 0..31]
        if (oitem) {
#pragma unroll
            for (int jt = 0; jt < NTT; ++jt) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ssum[jt][e] += __shfl_xor(ssum[jt][e], o, 64);
                        ssq[jt][e] += __shfl_xor(ssq[jt][e], o, 64);
                    }
                if (lane < 8) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        red[((wave * NTT + jt) * 2 + 0) * 32 + 4 * n4 + e] = ssum[jt][e];
                        red[((wave * NTT + jt) * 2 + 1) * 32 + 4 * n4 + e] = ssq[jt][e];
                    }
                }
            }
        }
        __syncthreads();
        if (tid < NTT * 2 * 32) {
            const int c = tid & 31, stt = (tid >> 5) & 1, jt = tid >> 6;
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += red[((w * NTT + jt) * 2 + stt) * 32 + c];
            const int slot = tile % a.nslots;
            atomicAdd(a.stats + (int64_t)slot * 2 * a.N + (int64_t)stt * a.N + n0 + 32 * jt + c, v);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// Weight gradient in the Winograd domain:  dg = G^T [ sum_patches (B^T d B) (.) (A dY A^T) ] G   (the transpose of the forward's three steps).
//   V[pos][patch][c] = B^T d B of the 4x4 input window (the forward's transform), W[pos][patch][n] = A dY A^T of the patch's 2x2 output gradients,
//   dU[pos][c][n] = sum_patches V * W: 16 GEMMs with K = patches -- 2.25x fewer multiplies than the nine taps of the direct weight gradient.
// A workgroup owns a 64 x 64 (c, n) tile of ALL 16 positions and a contiguous range of patches (split-K); per stage it takes a 2 x 4 block of patches,
// every thread transforms one input window (16 loads, 32 adds) and one patch of output gradients (4 loads, 12 adds) straight from global memory (64
// consecutive channels per wave = 256-byte requests; the window overlap is served by L1 / L2) into a double-buffered pair of LDS operand tiles.  Every split
// stores its dU tile (no atomics: wino_wgrad_reduce_kernel sums the copies in order, applies G^T . G and writes the packed [N][9][C] gradient the direct
// kernel would have written); the bias gradient is sum_patches W[1][1].
// ---------------------------------------------------------------------------------------------------------------------------------------
struct wino_wgrad_args {
    const float* x;          // (B, H, W, C) NHWC: the convolution's input ...
    const float* dy;         // (B, H, W, N): ... and the gradient of its output
    const float* x_scale;    // optional fused affine (+ ReLU) of the input (a BatchNorm that was never materialised); padding stays zero
    const float* x_shift;
    float* part;             // [S][16][C][N]
    float* bpart;            // [S][N] or null
    int B, H, W, C, N, x_relu;
    int lg_bpr, lg_bpi;      // log2 of the 2x4 patch blocks per block row / per image
    int nblocks, S;
};

constexpr int WG_KT = 8;     // patches per stage

// Eight waves, two positions per wave (eight 32x32 accumulators), two waves per SIMD with 256 registers each.  (A first form -- 16 waves x one position,
// the transform of a stage in front of its MFMAs, operands requested one stage ahead -- ran at 459 us on 128 -> 128 @ 64 x 64 x 64 where this one takes
// 4xx: all waves of a SIMD ran their transform phase together, right behind the barrier, with the matrix pipe idle; profiles/r06_winograd.txt.)
//   - the transform of stage i + 1 is cut into four slices, one behind the eight MFMAs of each k-step of stage i, in program order -- the vector work of a
//     wave issues while its own MFMAs occupy the matrix pipe;
//   - two operand register sets: a set is refilled (stage i + 3) as soon as its last slice is consumed, a full stage before its next use.
// AFF / RELU: the input sits behind a fused per-channel affine / ReLU (compile-time: a runtime flag costs a select per element on top of the operation)
template <bool AFF, bool RELU>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void wino_wgrad_kernel(const wino_wgrad_args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Vs = lds;                              // [2][16][8][64]
    float* const Ws = lds + 2 * 16 * WG_KT * 64;        // [2][16][8][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int nct = a.C >> 6, nnt = a.N >> 6;
    const int split = logical / (nct * nnt), c0 = ((logical / nnt) % nct) << 6, n0 = (logical % nnt) << 6;
    // whole PAIRS of stages per split (the block count is even: maps of at least 8 x 8): the loop body below is two stages, nothing conditional in it
    const int npairs = a.nblocks >> 1;
    const int g_lo = 2 * (int)((int64_t)npairs * split / a.S), g_hi = 2 * (int)((int64_t)npairs * (split + 1) / a.S);
    const int PH = a.H >> 1;
    // the patch of a thread is its WAVE's: everything about its position is wave-uniform -- told to the compiler (readfirstlane), the address arithmetic of
    // the twenty operand requests of a stage runs on the scalar unit and the requests take the (scalar base + lane offset) form
    const int t = __builtin_amdgcn_readfirstlane(wave), tr = t >> 2, tc = t & 3;
    float sc = 1.f, sh = 0.f;
    if (AFF) { sc = a.x_scale[c0 + lane]; sh = a.x_shift[c0 + lane]; }

    struct Regs { float v[16]; float w[4]; unsigned mask; };
    auto load = [&](int g, Regs& q) {
        const int b = g >> a.lg_bpi, rem = g & ((1 << a.lg_bpi) - 1), pr = 2 * (rem >> a.lg_bpr) + tr, pc = 4 * (rem & ((1 << a.lg_bpr) - 1)) + tc;
        // borders: clamp the address (always a valid element), zero the value at transform time -- sixteen unconditional loads, no branches
        const bool r0 = pr > 0, r3 = pr < PH - 1, q0 = pc > 0, q3 = 2 * pc + 2 < a.W;
        const int rowoff[4] = {r0 ? -a.W : 0, 0, a.W, r3 ? 2 * a.W : a.W}, coloff[4] = {q0 ? -1 : 0, 0, 1, q3 ? 2 : 1};
        // four scalar row bases x four per-lane column offsets (sixteen scalar address pairs cost a tenth of the wave's issue slots: the counters said so)
        const float* px = a.x + ((int64_t)(b * a.H + 2 * pr) * a.W + 2 * pc) * a.C + c0;
        int vcol[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) vcol[j] = lane + coloff[j] * a.C;
        q.mask = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* prow = px + rowoff[i] * a.C;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = (i == 0 ? r0 : i == 3 ? r3 : true) && (j == 0 ? q0 : j == 3 ? q3 : true);
                q.v[i * 4 + j] = prow[vcol[j]];
                q.mask |= ok ? 1u << (i * 4 + j) : 0u;
            }
        }
        const float* pg = a.dy + ((int64_t)(b * a.H + 2 * pr) * a.W + 2 * pc) * a.N + n0;
        const float* pg1 = pg + a.W * a.N;
        q.w[0] = pg[lane];
        q.w[1] = pg[lane + a.N];
        q.w[2] = pg1[lane];
        q.w[3] = pg1[lane + a.N];
    };
    float bsum = 0.f;
    float d[16], tv[4][4], tw[4][2];
    auto in_row = [&](const Regs& q, int r) {          // row r of the window: affine, ReLU, zero padding
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = q.v[r * 4 + j];
            if (AFF) v = v * sc + sh;
            if (RELU) asm("v_max_f32 %0, 0, %1" : "=v"(v) : "v"(v));      // (fmaxf costs a canonicalising v_max on top; the value is a finite activation)
            d[r * 4 + j] = ((q.mask >> (r * 4 + j)) & 1u) ? v : 0.f;
        }
    };
    // slice s of the transform of one stage (s = the row of the 4 x 4 transformed tiles it stores)
    auto slice = [&](const Regs& q, int buf, int s, bool live = true) {
        if (s == 0) {
            in_row(q, 0);
            in_row(q, 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) tv[0][j] = d[0 + j] - d[8 + j];
            tw[0][0] = q.w[0];          tw[0][1] = q.w[1];
            tw[1][0] = q.w[0] + q.w[2]; tw[1][1] = q.w[1] + q.w[3];
            tw[2][0] = q.w[0] - q.w[2]; tw[2][1] = q.w[1] - q.w[3];
            tw[3][0] = -q.w[2];         tw[3][1] = -q.w[3];
            bsum += live ? tw[1][0] + tw[1][1] : 0.f;
        } else if (s == 1) {
            in_row(q, 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tv[1][j] = d[4 + j] + d[8 + j];
                tv[2][j] = d[8 + j] - d[4 + j];
            }
        } else if (s == 3) {
            in_row(q, 3);
#pragma unroll
            for (int j = 0; j < 4; ++j) tv[3][j] = d[4 + j] - d[12 + j];
        }
        float* ov = Vs + buf * (16 * WG_KT * 64) + t * 64 + lane;
        float* ow = Ws + buf * (16 * WG_KT * 64) + t * 64 + lane;
        ov[(s * 4 + 0) * (WG_KT * 64)] = tv[s][0] - tv[s][2];
        ov[(s * 4 + 1) * (WG_KT * 64)] = tv[s][1] + tv[s][2];
        ov[(s * 4 + 2) * (WG_KT * 64)] = tv[s][2] - tv[s][1];
        ov[(s * 4 + 3) * (WG_KT * 64)] = tv[s][1] - tv[s][3];
        ow[(s * 4 + 0) * (WG_KT * 64)] = tw[s][0];
        ow[(s * 4 + 1) * (WG_KT * 64)] = tw[s][0] + tw[s][1];
        ow[(s * 4 + 2) * (WG_KT * 64)] = tw[s][0] - tw[s][1];
        ow[(s * 4 + 3) * (WG_KT * 64)] = -tw[s][1];
    };

    f32x16 acc[2][2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][i][j][r] = 0.f;

    const int nst = g_hi - g_lo;           // even, >= 2 (the host caps the split count at the number of stage pairs)
    const int g_last = g_hi - 1;
    Regs A, B;
    load(g_lo, A);
    load(g_lo + 1, B);
#pragma unroll
    for (int s = 0; s < 4; ++s) slice(A, 0, s);
    load(min(g_lo + 2, g_last), A);
    __syncthreads();
    // stage i multiplies out of buffer `buf` while the slices of stage i + 1 (operands in q) go into the other one; q is refilled with stage i + 3
    // (past the end the operand requests are clamped to the last stage and its slices go into the buffer nobody reads any more: the waits the compiler
    // derives stay "everything but the twenty newest requests" -- with a conditional request in the loop it falls back to vmcnt(0) at every use)
    auto stage = [&](int i, int buf, Regs& q) {
        // the operand fragments of k-step s + 1 are requested before the MFMAs of k-step s (two register sets): with the schedule pinned per k-step, reading
        // them where they are used puts an LDS round trip in front of every group of eight MFMAs
        float fa[2][2][2], fb[2][2][2];
        auto frags = [&](int s, int par) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float* va = Vs + (buf * 16 + 2 * wave + p) * (WG_KT * 64) + half * 64 + l31 + s * 128;
                const float* wb = Ws + (buf * 16 + 2 * wave + p) * (WG_KT * 64) + half * 64 + l31 + s * 128;
                fa[par][p][0] = va[0]; fa[par][p][1] = va[32];
                fb[par][p][0] = wb[0]; fb[par][p][1] = wb[32];
            }
        };
        frags(0, 0);
#pragma unroll
        for (int s = 0; s < WG_KT / 2; ++s) {
            if (s + 1 < WG_KT / 2) frags(s + 1, (s + 1) & 1);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float a0 = fa[s & 1][p][0], a1 = fa[s & 1][p][1], b0 = fb[s & 1][p][0], b1 = fb[s & 1][p][1];
                acc[p][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[p][0][0], 0, 0, 0);
                acc[p][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[p][0][1], 0, 0, 0);
                acc[p][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[p][1][0], 0, 0, 0);
                acc[p][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[p][1][1], 0, 0, 0);
            }
            slice(q, buf ^ 1, s, i + 1 < nst);
            __builtin_amdgcn_sched_barrier(0);      // (the scheduler would hoist the NEXT stage's slices up here, and with them the wait for the newest requests)
        }
        load(min(g_lo + i + 3, g_last), q);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int i = 0; i < nst; i += 2) {
        stage(i, 0, B);
        stage(i + 1, 1, A);
    }

#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float* o = a.part + ((int64_t)(split * 16 + 2 * wave + q) * a.C + c0) * a.N + n0 + l31;
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[(int64_t)(32 * ci + (r & 3) + 8 * (r >> 2) + 4 * half) * a.N + 32 * ni] = acc[q][ci][ni][r];
    }
    if (a.bpart && c0 == 0) {
        float* red = lds;
        red[t * 64 + lane] = bsum;
        __syncthreads();
        if (tid < 64) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) v += red[k * 64 + tid];
            a.bpart[(int64_t)split * a.N + n0 + tid] = v;
        }
    }
}

// grid (N / 16, C), 1024 threads = 4 split groups x 16 positions x 16 output channels: sum the split copies (a fixed order: eight independent partial
// sums per group -- the requests of a thread overlap instead of queueing behind one another -- then the groups in order), dg = G^T dU G -> R[n][tap][c]
// (+ bias gradient).  (First form: 256 threads, one serial chain over all copies per thread: 108 us average in the ResNet18 step's trace, as long as the
// matrix kernel it follows on the single-tile layers with 256 copies.)
__global__ __launch_bounds__(1024) void wino_wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bpart, int S, int C, int N,
                                                                 float* __restrict__ R, int ld, float* __restrict__ bias_out) {
    __shared__ float grp[4][16][17];
    __shared__ float dU[16][17];
    const int tid = threadIdx.x & 255, g = threadIdx.x >> 8, pos = tid >> 4, ni = tid & 15, c = blockIdx.y, n0 = blockIdx.x * 16;
    const float* p = part + ((int64_t)pos * C + c) * N + n0 + ni;
    const int64_t sstride = (int64_t)16 * C * N;
    const int s_lo = (int)((int64_t)S * g / 4), s_hi = (int)((int64_t)S * (g + 1) / 4);
    float v8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int sp = s_lo;
    for (; sp + 8 <= s_hi; sp += 8)
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] += p[(sp + u) * sstride];
    for (int u = 0; sp < s_hi; ++sp, ++u) v8[u] += p[sp * sstride];
    grp[g][pos][ni] = ((v8[0] + v8[1]) + (v8[2] + v8[3])) + ((v8[4] + v8[5]) + (v8[6] + v8[7]));
    __syncthreads();
    if (g == 0) dU[pos][ni] = (grp[0][pos][ni] + grp[1][pos][ni]) + (grp[2][pos][ni] + grp[3][pos][ni]);
    __syncthreads();
    if (threadIdx.x < 144) {
        const int tap = tid >> 4, i = tap / 3, j = tap % 3;
        // rows of G^T: (1, .5, .5, 0), (0, .5, -.5, 0), (0, .5, .5, 1)
        const float gi[3][4] = {{1.f, 0.5f, 0.5f, 0.f}, {0.f, 0.5f, -0.5f, 0.f}, {0.f, 0.5f, 0.5f, 1.f}};
        float r = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float rowv = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) rowv += dU[u * 4 + w][ni] * gi[j][w];
            r += gi[i][u] * rowv;
        }
        R[((int64_t)(n0 + ni) * 9 + tap) * ld + c] = r;
    }
    if (bias_out && bpart && c == 0 && threadIdx.x < 16) {
        float b = 0.f;
        for (int sb = 0; sb < S; ++sb) b += bpart[(int64_t)sb * N + n0 + threadIdx.x];
        bias_out[n0 + threadIdx.x] = b;
    }
}

}  // namespace awr

using namespace awr;

extern "C" {

int awr_wino_weights(const float* w, int N, int C, int Npad, int Cpad, int mirror, float* U, void* stream) {
    AWR_REQUIRE(w && U && N > 0 && C > 0 && Npad >= N && Npad % 32 == 0 && Cpad >= C && Cpad % 8 == 0,
                "wino_weights: N=%d C=%d Npad=%d Cpad=%d (Npad: a multiple of 32 >= N, Cpad: a multiple of 8 >= C)", N, C, Npad, Cpad);
    hipLaunchKernelGGL(wino_weight_kernel, dim3((Cpad * Npad + 255) / 256), dim3(256), 0, as_stream(stream), w, N, C, Npad, Cpad, mirror, U);
    return check_launch("wino_weight_kernel");
}

static int g_winograd = []() { const char* e = getenv("AWR_WINOGRAD"); return e ? atoi(e) : 0; }();

int awr_set_conv_winograd(int on) {
    AWR_REQUIRE(on >= 0 && on < 16, "conv_winograd: 0 (direct implicit GEMM everywhere), 1 (Winograd F(2x2, 3x3) forward where it is eligible), 2 (forward, data "
                                    "and weight gradient), 3 (forward and weight gradient); + 4 (tests: wherever the kernels can run, whatever the launch size); "
                                    "+ 8 (A/B: never the 64-channel tile form)");
    g_winograd = on;
    return AWR_OK;
}

int awr_get_conv_winograd(void) { return g_winograd; }

int awr_wino_eligible(int B, int H, int W, int C, int N) {
    const int minside = (g_winograd & 4) ? 4 : 8;      // (8 x 8 maps: only where the batch still gives 256 workgroups -- ResNet18 layer4 at batch 128 and up)
    if (H < minside || W < minside || (H & (H - 1)) || (W & (W - 1)) || C % 8 || N % 32) return 0;
    if ((int64_t)B * H * W * C >= (1LL << 31)) return 0;
    const int PH = H / 2, PW = W / 2, PCt = PW < 32 ? PW : 32, PRt = PH < W_TP / PCt ? PH : W_TP / PCt, nimg = W_TP / (PRt * PCt);
    const int64_t wgs = (int64_t)(PW / PCt) * (PH / PRt) * ((B + nimg - 1) / nimg) * (N / W_TN);
    return (g_winograd & 4) || wgs >= 256;      // (fewer workgroups than half the chip's slots: the direct kernel's smaller tiles win, profiles/r06_winograd.txt)
}

int awr_wino_conv3x3(const float* in, const float* U, const float* bias, float* out, int B, int H, int W, int C, int N, int relu, int kb, void* stream) {
    AWR_REQUIRE(in && U && out, "wino_conv3x3: NULL pointer");
    AWR_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "wino_conv3x3: even map sizes only (H=%d, W=%d)", H, W);
    AWR_REQUIRE(C % 8 == 0 && N % 32 == 0, "wino_conv3x3: C %% 8 == 0 and N %% 32 == 0 (C=%d, N=%d)", C, N);
    const int64_t P = (int64_t)B * (H / 2) * (W / 2);
    wino_args a{in, U, bias, out, B, H, W, C, N, relu};
    const dim3 grid((unsigned)((P + W_TP - 1) / W_TP), N / W_TN);
    // kb: channels per LDS stage (4 / 8); + 100: the 256-thread form (4 positions per wave) instead of the 512-thread one (2 per wave)
    if (kb == 104) hipLaunchKernelGGL((wino_fwd_kernel<4, 4>), grid, dim3(256), 0, as_stream(stream), a);
    else if (kb == 108) hipLaunchKernelGGL((wino_fwd_kernel<8, 4>), grid, dim3(256), 0, as_stream(stream), a);
    else if (kb == 4) hipLaunchKernelGGL((wino_fwd_kernel<4, 8>), grid, dim3(512), 0, as_stream(stream), a);
    else if (kb == 1008) hipLaunchKernelGGL((wino_fwd_kernel<8, 8, 1>), grid, dim3(512), 0, as_stream(stream), a);
    else if (kb == 2008) hipLaunchKernelGGL((wino_fwd_kernel<8, 8, 2>), grid, dim3(512), 0, as_stream(stream), a);
    else if (kb == 3008) hipLaunchKernelGGL((wino_fwd_kernel<8, 8, 3>), grid, dim3(512), 0, as_stream(stream), a);
    else if (kb == 4008) hipLaunchKernelGGL((wino_fwd_kernel<8, 8, 4>), grid, dim3(512), 0, as_stream(stream), a);
    else hipLaunchKernelGGL((wino_fwd_kernel<8, 8>), grid, dim3(512), 0, as_stream(stream), a);
    return check_launch("wino_fwd_kernel");
}


int awr_wino_conv(const awr_wino_args* p, void* stream) {
    AWR_REQUIRE(p && p->in && p->U && p->out, "wino_conv: NULL pointer");
    const int B = p->B, H = p->H, W = p->W, C = p->C, N = p->N;
    AWR_REQUIRE(B > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "wino_conv: even map sizes only (H=%d, W=%d)", H, W);
    AWR_REQUIRE(C % 8 == 0 && N % 32 == 0, "wino_conv: C %% 8 == 0 and N %% 32 == 0 (C=%d, N=%d)", C, N);
    AWR_REQUIRE((int64_t)B * H * W * C < (1LL << 31), "wino_conv: the input tensor must have fewer than 2^31 elements (32-bit offsets)");
    AWR_REQUIRE((p->in_scale == nullptr) == (p->in_shift == nullptr), "wino_conv: in_scale / in_shift come together");
    AWR_REQUIRE(!p->stats || p->nslots > 0, "wino_conv: statistics need nslots > 0");
    AWR_REQUIRE(!p->bnr_y || (p->bnr_coef && p->stats && !p->bias && !p->relu), "wino_conv: the BatchNorm-backward reduction needs bnr_coef and stats, and excludes bias / ReLU");
    AWR_REQUIRE(!p->bnr_act || p->bnr_y, "wino_conv: bnr_act without bnr_y");
    const int PH = H / 2, PW = W / 2;
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    AWR_REQUIRE(pow2(PH) && pow2(PW), "wino_conv: power-of-two maps only (H=%d, W=%d)", H, W);
    AWR_REQUIRE((p->out_scale == nullptr) == (p->out_shift == nullptr), "wino_conv: out_scale / out_shift come together");
    AWR_REQUIRE(!p->out_scale || (!p->stats && !p->bnr_y), "wino_conv: the output affine (folded eval-mode BatchNorm) excludes statistics / the BatchNorm-backward reduction");
    wino2_args a{p->in, p->U, p->bias, p->in_scale, p->in_shift, p->out_scale, p->out_shift, p->out, p->stats, p->res, p->bnr_y, p->bnr_coef, p->bnr_act,
                 B, H, W, C, N, p->relu, p->relu_in, p->nslots, 0, 0, 0};
    a.PCt = PW < 32 ? PW : 32;
    a.PRt = PH < W_TP / a.PCt ? PH : W_TP / a.PCt;
    a.nimg = W_TP / (a.PRt * a.PCt);
    AWR_REQUIRE(a.nimg * (2 * a.PRt + 2) * (2 * a.PCt + 2) <= 576, "wino_conv: raw tile of %d x %d x %d patches exceeds the LDS region", a.nimg, a.PRt, a.PCt);
    const int tiles = (PW / a.PCt) * (PH / a.PRt) * ((B + a.nimg - 1) / a.nimg);
    constexpr int KB = 8;
    // the wide form (64 output channels per workgroup, 16 waves, one workgroup per CU): every transformed window feeds twice the multiplies -- 174 instead of
    // 163 algorithmic TF on 128 -> 128 @ 64 x 64, 188 instead of 162 on 256 -> 256 @ 16 x 16 (profiles/r06_winograd.txt); needs N % 64 == 0 and enough
    // workgroups to fill the chip at one per CU
    const bool wide = !(g_winograd & 8) && N % 64 == 0 && ((g_winograd & 4) || (int64_t)tiles * (N / 64) >= 256);
    const int TNsel = wide ? 64 : W_TN;
    const size_t lds_v = (size_t)(16 * W_TP * (KB + 1) + 16 * KB * TNsel + 576 * (KB + 1)) * 4, lds_x = 16 * 32 * 32 * 4;
    const size_t lds = lds_v > lds_x ? lds_v : lds_x;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)wino2_fwd_kernel<KB, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wino2_fwd_kernel<KB, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wino2_fwd_kernel<KB, 2, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) { set_error("wino_conv: hipFuncSetAttribute: %s", hipGetErrorString(e)); return AWR_ERR_HIP; }
        attr_done = true;
    }
    if (wide)
        hipLaunchKernelGGL((wino2_fwd_kernel<KB, 2, 64>), dim3(tiles * (N / 64)), dim3(1024), lds, as_stream(stream), a);
    else if (a.nimg * (2 * a.PRt + 2) * (2 * a.PCt + 2) * (KB / 4) <= 2 * 512)
        hipLaunchKernelGGL((wino2_fwd_kernel<KB, 2>), dim3(tiles * (N / W_TN)), dim3(512), lds, as_stream(stream), a);
    else
        hipLaunchKernelGGL((wino2_fwd_kernel<KB, 3>), dim3(tiles * (N / W_TN)), dim3(512), lds, as_stream(stream), a);
    return check_launch("wino2_fwd_kernel");
}

int awr_wino2_conv3x3(const float* in, const float* U, const float* bias, const float* in_scale, const float* in_shift, int relu_in, float* out,
                      double* stats, int nslots, int B, int H, int W, int C, int N, int relu, void* stream) {
    awr_wino_args p;
    memset(&p, 0, sizeof p);
    p.in = in; p.U = U; p.bias = bias; p.in_scale = in_scale; p.in_shift = in_shift; p.out = out; p.stats = stats;
    p.B = B; p.H = H; p.W = W; p.C = C; p.N = N; p.relu = relu; p.relu_in = relu_in; p.nslots = nslots;
    return awr_wino_conv(&p, stream);
}

// The data gradient of a stride-1 3x3 convolution as described by the direct kernel's argument block (plans): whatever of its epilogue this kernel
// implements runs as Winograd (U = the MIRRORED transform, awr_wino_weights(..., mirror = 1)), anything else falls back to awr_conv_gemm.
int awr_wino_dgrad_supported(const awr_conv_args* d) {
    if (!d) return 0;
    const bool plain = !d->in_scale && !d->relu_in && !d->bias && !d->out_scale && !d->relu_out && !d->in2 && !d->w2 && !d->partial && !d->in_bnb_y && !d->bnr2_y &&
                       !d->in_split && !d->pool_out && (!d->res || d->res == d->out) && (!d->bnr_y || (d->bnr_coef && d->stats)) && (d->bnr_y || !d->stats);
    return plain ? 1 : 0;
}

int awr_wino_dgrad_or_direct(const awr_conv_args* d, const float* U, void* stream) {
    AWR_REQUIRE(d && U, "wino_dgrad: NULL pointer");
    if (!awr_wino_dgrad_supported(d)) return awr_conv_gemm(d, stream);
    awr_wino_args p;
    memset(&p, 0, sizeof p);
    p.in = d->in; p.U = U; p.out = d->out; p.res = d->res; p.bnr_y = d->bnr_y; p.bnr_coef = d->bnr_coef; p.bnr_act = d->bnr_act; p.stats = d->stats;
    p.nslots = d->stats ? (d->stat_slots > 0 ? d->stat_slots : AWR_STAT_SLOTS) : 0;
    p.B = d->B; p.H = d->Hin; p.W = d->Win; p.C = d->Cin; p.N = d->N;
    return awr_wino_conv(&p, stream);
}


// Weight gradient of a stride-1 3x3 convolution in the Winograd domain (see wino_wgrad_kernel).  R[N][9][ld] (ld >= C) and bias_grad[N] are ASSIGNED.
// scratch: awr_wino_wgrad_scratch() floats.  Deterministic (ordered sums over the split copies).
static int wino_wgrad_splits(int B, int H, int W, int C, int N) {
    // ONE round of workgroups, one per CU (a second round costs a second 256 KB copy per CU and buys nothing: 433 -> 401 us on 128 -> 128 @ 64 x 64 x 64)
    const int nblocks = B * (H / 4) * (W / 8), tiles = (C / 64) * (N / 64);
    int S = (256 + tiles - 1) / tiles;
    if (S > nblocks / 2) S = nblocks / 2;      // every split owns at least one PAIR of stages (the loop body of wino_wgrad_kernel)
    return S < 1 ? 1 : S;
}

int awr_wino_wgrad_eligible(int B, int H, int W, int C, int N) {
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    if (C % 64 || N % 64 || !pow2(H) || !pow2(W) || H < 8 || W < 8) return 0;
    if ((int64_t)B * H * W * (C > N ? C : N) >= (1LL << 31)) return 0;
    if (g_winograd & 4) return 1;
    // every split stores a 256 KB copy of its tile (67 MB per launch whatever the layer): it takes a K loop of 8 stages to pay for it
    // (profiles/r06_winograd.txt: 64 -> 64 @ 32 x 32 x 64 = 8 stages per split: 1.32x the direct kernel; 256 -> 256 @ 8 x 8 x 64: 1.06-1.17x)
    const int stages = B * (H / 4) * (W / 8) / wino_wgrad_splits(B, H, W, C, N);
    return stages >= 8;
}

int64_t awr_wino_wgrad_scratch(int B, int H, int W, int C, int N) {
    const int S = wino_wgrad_splits(B, H, W, C, N);
    return (int64_t)S * 16 * C * N + (int64_t)S * N;
}

int awr_wino_wgrad(const float* x, const float* dy, const float* x_scale, const float* x_shift, int x_relu, int B, int H, int W, int C, int N,
                   float* scratch, float* R, int ld, float* bias_grad, void* stream) {
    AWR_REQUIRE(x && dy && scratch && R, "wino_wgrad: NULL pointer");
    AWR_REQUIRE((x_scale == nullptr) == (x_shift == nullptr), "wino_wgrad: x_scale / x_shift come together");
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    AWR_REQUIRE(B > 0 && C > 0 && N > 0 && C % 64 == 0 && N % 64 == 0 && pow2(H) && pow2(W) && H >= 8 && W >= 8 && ld >= C,
                "wino_wgrad: C and N multiples of 64, power-of-two maps of at least 8 x 8 (B=%d H=%d W=%d C=%d N=%d ld=%d)", B, H, W, C, N, ld);
    AWR_REQUIRE((int64_t)B * H * W * (C > N ? C : N) < (1LL << 31), "wino_wgrad: tensors of fewer than 2^31 elements (32-bit offsets)");
    wino_wgrad_args a;
    memset(&a, 0, sizeof a);
    a.x = x; a.dy = dy; a.x_scale = x_scale; a.x_shift = x_shift; a.x_relu = x_relu;
    a.B = B; a.H = H; a.W = W; a.C = C; a.N = N;
    a.S = wino_wgrad_splits(B, H, W, C, N);
    a.nblocks = B * (H / 4) * (W / 8);
    a.lg_bpr = __builtin_ctz(W / 8);
    a.lg_bpi = __builtin_ctz((H / 4) * (W / 8));
    a.part = scratch;
    a.bpart = bias_grad ? scratch + (int64_t)a.S * 16 * C * N : nullptr;
    const size_t lds = (size_t)4 * 16 * WG_KT * 64 * 4;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)wino_wgrad_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wino_wgrad_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wino_wgrad_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wino_wgrad_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) { set_error("wino_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e)); return AWR_ERR_HIP; }
        attr_done = true;
    }
    const dim3 grid(a.S * (C / 64) * (N / 64));
    if (x_scale && x_relu) hipLaunchKernelGGL((wino_wgrad_kernel<true, true>), grid, dim3(512), lds, as_stream(stream), a);
    else if (x_scale) hipLaunchKernelGGL((wino_wgrad_kernel<true, false>), grid, dim3(512), lds, as_stream(stream), a);
    else if (x_relu) hipLaunchKernelGGL((wino_wgrad_kernel<false, true>), grid, dim3(512), lds, as_stream(stream), a);
    else hipLaunchKernelGGL((wino_wgrad_kernel<false, false>), grid, dim3(512), lds, as_stream(stream), a);
    int rc = check_launch("wino_wgrad_kernel");
    if (rc != AWR_OK) return rc;
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(N / 16, C), dim3(1024), 0, as_stream(stream), a.part, a.bpart, a.S, C, N, R, ld, bias_grad);
    return check_launch("wino_wgrad_reduce_kernel");
}

}  // extern "C"
