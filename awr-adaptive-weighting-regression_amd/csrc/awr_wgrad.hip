// Weight gradients of the convolution family: host-side planning / dispatch of the four kernels (conv_wgrad_kernel, conv_wgrad_dma_kernel,
// conv_wgrad_row_kernel, conv_wgrad_taps_kernel, conv_wgrad_split_kernel in csrc/awr_conv_kernels.inc).
#include <string.h>

#include "awr_conv_kernels.inc"
#include "awr_conv_modes.h"

using namespace awr;

template <int TM, int TN, int KP>
static void launch_wgrad_dma_t(const awr_wgrad_args* a, bool dreg, bool greg, dim3 grid, hipStream_t st, int chunk, int wshift, int hshift) {
    if (dreg && greg) hipLaunchKernelGGL((conv_wgrad_dma_kernel<TM, TN, KP, true, true>), grid, dim3(256), 0, st, *a, chunk, wshift, hshift);
    else if (dreg) hipLaunchKernelGGL((conv_wgrad_dma_kernel<TM, TN, KP, true, false>), grid, dim3(256), 0, st, *a, chunk, wshift, hshift);
    else if (greg) hipLaunchKernelGGL((conv_wgrad_dma_kernel<TM, TN, KP, false, true>), grid, dim3(256), 0, st, *a, chunk, wshift, hshift);
    else hipLaunchKernelGGL((conv_wgrad_dma_kernel<TM, TN, KP, false, false>), grid, dim3(256), 0, st, *a, chunk, wshift, hshift);
}
static void launch_wgrad_dma(const awr_wgrad_args* a, int TM, int TN, int kp, bool dreg, bool greg, dim3 grid, hipStream_t st, int chunk, int wshift, int hshift) {
#define AWR_WD(tm, tn)                                                                                       \
    do {                                                                                                     \
        if (kp == 32) launch_wgrad_dma_t<tm, tn, 32>(a, dreg, greg, grid, st, chunk, wshift, hshift);        \
        else launch_wgrad_dma_t<tm, tn, 16>(a, dreg, greg, grid, st, chunk, wshift, hshift);                 \
    } while (0)
    if (TM == 2 && TN == 2) AWR_WD(2, 2);
    else if (TM == 2 && TN == 1) AWR_WD(2, 1);
    else if (TM == 1 && TN == 2) AWR_WD(1, 2);
    else AWR_WD(1, 1);
#undef AWR_WD
}

extern "C" {

// geometry served by the one-wave-per-tap kernel: 3x3 (stride 1 / 2) and 4x4 stride-2 filters whose taps are listed row-major
// from the top-left one, power-of-two D maps that a PH x 8 patch tiles
static int wgrad_taps_patch_rows(const awr_wgrad_args* a) {
    const int ks = a->T == 9 ? 3 : a->T == 16 ? 4 : 0;
    if (!ks || (ks == 4 && a->sg != 2) || (a->sg != 1 && a->sg != 2)) return 0;
    for (int t = 0; t < a->T; ++t)
        if (a->dy[t] != a->dy[0] + t / ks || a->dx[t] != a->dx[0] + t % ks) return 0;
    const int ph = a->sg == 1 ? 4 : 2;
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    if (!pow2(a->Hd) || !pow2(a->Wd) || a->Wd < 8 || a->Hd < ph) return 0;
    return ph;
}

// launch geometry of one weight-gradient problem: algorithm, tile, split-K depth.  Shared by the launch and by
// awr_conv_wgrad_splits (a deterministic-mode caller sizes its per-chunk copies of R with it)
struct wgrad_launch {
    int row_pw;         // > 0: one-workgroup-per-kernel-row kernel (3x3 stride 1), segments of this many pixels (8 | 16)
    int taps_ph;        // > 0: one-wave-per-tap kernel with this patch height
    int TM, TN, tiles;
    int64_t nsplit, chunk;
    int pc_log, pr_log;
};

static bool wgrad_row_ok(const awr_wgrad_args* a) {
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    const int64_t M = (int64_t)a->B * a->Hd * a->Wd;
    bool ok = wg_products() == 1 && g_staging != 0 && a->T == 9 && a->sg == 1 && a->Hd == a->Hg && a->Wd == a->Wg && pow2(a->Wd) && pow2(a->Hd) && a->Wd >= 8 &&
              a->Hd >= 2 && !a->d_scale && M < (1 << 24) && a->Cd * 4 < (1 << 24) && a->Cg * 4 < (1 << 24);
    for (int t = 0; ok && t < 9; ++t) ok = a->dy[t] == t / 3 - 1 && a->dx[t] == t % 3 - 1;
    return ok;
}

static int wgrad_plan(const awr_wgrad_args* a, wgrad_launch* w) {
    AWR_REQUIRE(a->Cd % 4 == 0 && a->Cg % 4 == 0 && a->Cd > 0 && a->Cg > 0, "conv_wgrad: channel counts must be multiples of 4");
    AWR_REQUIRE(a->T >= 1 && a->T <= 16 && a->ld >= a->Cg && a->sg >= 1, "conv_wgrad: bad geometry");
    AWR_REQUIRE((a->d_scale == nullptr) == (a->d_shift == nullptr) && (a->g_scale == nullptr) == (a->g_shift == nullptr),
                "conv_wgrad: scale/shift must come in pairs");
    AWR_REQUIRE(a->algo >= 0 && a->algo <= 3, "conv_wgrad: algo must be 0 (automatic), 1 (workgroup per tap), 2 (wave per tap) or 3 (workgroup per kernel row)");
    AWR_REQUIRE(a->split_stride >= 0 && (a->split_stride == 0 || a->max_split >= 1), "conv_wgrad: split_stride > 0 (deterministic K-chunk copies) needs max_split >= 1");
    const int64_t M = (int64_t)a->B * a->Hd * a->Wd;
    AWR_REQUIRE(M > 0 && M < (1LL << 31), "conv_wgrad: bad pixel count");
    AWR_REQUIRE(M * a->Cd * 4 < (1LL << 32) && (int64_t)a->B * a->Hg * a->Wg * a->Cg * 4 < (1LL << 32),
                "conv_wgrad: tensors must stay below 4 GB (32-bit buffer offsets)");
    static const int env_algo = []() { const char* e = getenv("AWR_WGRAD_ALGO"); return e ? atoi(e) : 0; }();      // study hook
    const int ph = wg_products() == 1 ? wgrad_taps_patch_rows(a) : 0;
    const int algo = a->algo ? a->algo : (env_algo ? env_algo : 1);
    AWR_REQUIRE(a->algo != 2 || ph, "conv_wgrad: algo 2 (wave per tap) does not serve this geometry / product mode");
    w->taps_ph = 0;
    w->row_pw = 0;
    {   // one workgroup per kernel row: 3x3, stride 1, same-size power-of-two maps, taps row-major from the top-left one, D plain
        const bool ok = wgrad_row_ok(a);
        // default (algo 0): wherever the geometry allows -- isolated launches 121-136 TF against 105-127 for the best per-tap geometry with a
        // plain gathered operand, 114-127 against 98-124 with the fused BatchNorm loader (since that became an in-LDS pass;
        // profiles/r04_loop_exits.txt).  AWR_WGRAD_ROW=0: never; an explicit algo wins, and the plan autotuner times algo 3 against the
        // per-tap geometries per launch.
        static const int env_row = env_int("AWR_WGRAD_ROW", -1);
        const bool want_row = env_row != 0;
        AWR_REQUIRE(a->algo != 3 || ok, "conv_wgrad: algo 3 (workgroup per kernel row) serves 3x3 stride-1 filters on power-of-two maps >= 8 wide in the FP32-MFMA mode");
        if (ok && (a->algo == 3 || (a->algo == 0 && want_row))) {
            w->row_pw = a->Wd >= 16 ? 16 : 8;
            const int64_t nstage = M / 16;
            w->tiles = ((a->Cd + 63) / 64) * ((a->Cg + 63) / 64) * 3;
            // The workgroups of a launch are equally long, so the count that fits ONE generation of resident workgroups wins (five or six per
            // CU by registers; isolated launches: 768 / 1024 / 1280 workgroups within 2 TF of each other, 1026 with the round's first build --
            // two generations, the second almost empty -- 106 against 114-119; profiles/r04_microbench_wgrad_row.txt): floor, not ceil
            const int want = a->target_blocks > 0 ? a->target_blocks : 1024;
            int64_t nsplit = want / w->tiles;
            if (nsplit > nstage / 8) nsplit = nstage / 8;            // at least 8 stages (128 pixels) per workgroup
            if (a->split_stride && nsplit > a->max_split) nsplit = a->max_split;
            if (nsplit < 1) nsplit = 1;
            w->chunk = (nstage + nsplit - 1) / nsplit;
            w->nsplit = (nstage + w->chunk - 1) / w->chunk;
            return AWR_OK;
        }
    }
    if (ph && algo == 2) {
        auto log2i = [](int v) { int s = 0; while ((1 << s) < v) ++s; return s; };
        w->taps_ph = ph;
        w->pc_log = log2i(a->Wd / 8);
        w->pr_log = log2i(a->Hd / ph);
        const int64_t npatch = (int64_t)a->B << (w->pc_log + w->pr_log);
        w->tiles = ((a->Cd + 63) / 64) * ((a->Cg + 63) / 64);
        const int want = a->target_blocks > 0 ? a->target_blocks : 256;
        int64_t nsplit = (want + w->tiles - 1) / w->tiles;
        if (nsplit > npatch / 4) nsplit = npatch / 4;            // at least 4 K-slices per workgroup
        if (a->split_stride && nsplit > a->max_split) nsplit = a->max_split;
        if (nsplit < 1) nsplit = 1;
        w->chunk = (npatch + nsplit - 1) / nsplit;
        w->nsplit = (npatch + w->chunk - 1) / w->chunk;
        return AWR_OK;
    }
    // measured (tools/microbench_gemm.py, AWR_WGRAD_BLOCKS sweep): 64x64 tiles with ~3072 workgroups win on the small
    // feature maps; the 128x64 (cd x cg) tile with ~2048 workgroups wins once there are >= 128K pixels to contract.
    int TM = (a->Cd > 64 && M >= 131072) ? 2 : 1, TN = 1;
    if (a->tile_m) {
        AWR_REQUIRE((a->tile_m == 1 || a->tile_m == 2) && (a->tile_n == 1 || a->tile_n == 2), "conv_wgrad: tile_m/tile_n must be 1 or 2");
        TM = a->tile_m;
        TN = a->tile_n;
    }
    if (g_force_tm) { TM = g_force_tm; TN = g_force_tn; }
    w->TM = TM;
    w->TN = TN;
    w->tiles = ((a->Cd + 64 * TM - 1) / (64 * TM)) * ((a->Cg + 64 * TN - 1) / (64 * TN)) * a->T;
    static const int target_blocks = []() { const char* e = getenv("AWR_WGRAD_BLOCKS"); return e ? atoi(e) : 0; }();   // tuning hook
    const int want_blocks = a->target_blocks > 0 ? a->target_blocks : target_blocks ? target_blocks : (TM == 2 ? 2048 : 3072);
    int64_t nsplit = (want_blocks + w->tiles - 1) / w->tiles;
    const int64_t max_split = (M + 8 * BK - 1) / (8 * BK);   // at least 256 pixels per workgroup
    if (nsplit > max_split) nsplit = max_split;
    if (a->split_stride && nsplit > a->max_split) nsplit = a->max_split;
    if (nsplit < 1) nsplit = 1;
    int64_t chunk = (M + nsplit - 1) / nsplit;
    chunk = (chunk + 63) / 64 * 64;
    w->chunk = chunk;
    w->nsplit = (M + chunk - 1) / chunk;
    return AWR_OK;
}

int awr_conv_wgrad_algo_ok(const awr_wgrad_args* a, int algo) {
    if (!a) return 0;
    if (algo == 0 || algo == 1) return 1;
    if (algo == 2) return wg_products() == 1 && wgrad_taps_patch_rows(a) > 0;
    if (algo == 3) return wgrad_row_ok(a) ? 1 : 0;
    return 0;
}

int awr_conv_wgrad_splits(const awr_wgrad_args* a, int* nsplit) {
    AWR_REQUIRE(a && nsplit, "conv_wgrad_splits: null pointer");
    const int64_t d_img = (int64_t)a->Hd * a->Wd * a->Cd, g_img = (int64_t)a->Hg * a->Wg * a->Cg;
    int nchunk = 1;
    while ((d_img * (a->B / nchunk) * 4 >= (1LL << 32) || g_img * (a->B / nchunk) * 4 >= (1LL << 32)) && a->B % (nchunk * 2) == 0) nchunk *= 2;
    awr_wgrad_args b = *a;
    b.B = a->B / nchunk;
    wgrad_launch w;
    if (int e = wgrad_plan(&b, &w)) return e;
    *nsplit = (int)w.nsplit * nchunk;          // batch chunks (tensors above 4 GB) take consecutive ranges of copies
    return AWR_OK;
}

static int conv_wgrad_one(const awr_wgrad_args* a, void* stream) {
    wgrad_launch w;
    if (int e = wgrad_plan(a, &w)) return e;
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)w.tiles, (unsigned)w.nsplit);
    if (w.row_pw) {
        int wlog = 0, hlog = 0;
        while ((1 << wlog) < a->Wd) ++wlog;
        while ((1 << hlog) < a->Hd) ++hlog;
        if (w.row_pw == 16) {
            if (a->g_scale) hipLaunchKernelGGL((conv_wgrad_row_kernel<16, true>), grid, dim3(256), 0, st, *a, (int)w.chunk, wlog, hlog);
            else hipLaunchKernelGGL((conv_wgrad_row_kernel<16, false>), grid, dim3(256), 0, st, *a, (int)w.chunk, wlog, hlog);
        } else {
            if (a->g_scale) hipLaunchKernelGGL((conv_wgrad_row_kernel<8, true>), grid, dim3(256), 0, st, *a, (int)w.chunk, wlog, hlog);
            else hipLaunchKernelGGL((conv_wgrad_row_kernel<8, false>), grid, dim3(256), 0, st, *a, (int)w.chunk, wlog, hlog);
        }
        return check_launch("conv_wgrad_row_kernel");
    }
    if (w.taps_ph) {
        if (a->T == 16) hipLaunchKernelGGL((conv_wgrad_taps_kernel<2, 4, 2>), grid, dim3(1024), 0, st, *a, (int)w.chunk, w.pc_log, w.pr_log);
        else if (a->sg == 2) hipLaunchKernelGGL((conv_wgrad_taps_kernel<2, 3, 2>), grid, dim3(768), 0, st, *a, (int)w.chunk, w.pc_log, w.pr_log);
        else hipLaunchKernelGGL((conv_wgrad_taps_kernel<1, 3, 4>), grid, dim3(768), 0, st, *a, (int)w.chunk, w.pc_log, w.pr_log);
        return check_launch("conv_wgrad_taps_kernel");
    }
    const int TM = w.TM, TN = w.TN;
    const int64_t chunk = w.chunk;
    auto log2i = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; };
    int wshift = log2i(a->Wd), hshift = log2i(a->Hd);
    if (wshift < 0 || hshift < 0) wshift = hshift = -1;
    // FP32 kernel: multiply-free slice addressing when G's map sides are powers of two as well and both tensors hold < 2^24 pixels
    // (packed into the hshift argument: bits 0-5 log2 Hd, bit 6 the flag, bits 8-13 log2 Wg, bits 16-21 log2 Hg)
    int hshift_f32 = hshift;
    {
        const int gws = log2i(a->Wg), ghs = log2i(a->Hg);
        static const bool no_fast = getenv("AWR_WGRAD_SLOW_ADDR") != nullptr;      // same-box A/B hook
        if (!no_fast && wshift >= 0 && gws >= 0 && ghs >= 0 && (a->sg == 1 || a->sg == 2) && (int64_t)a->B * a->Hd * a->Wd < (1 << 24) &&
            (int64_t)a->B * a->Hg * a->Wg < (1 << 24) && a->Cd * 4 < (1 << 24) && a->Cg * 4 < (1 << 24))
            hshift_f32 = hshift | 64 | (gws << 8) | (ghs << 16);
    }
    // LDS-DMA staging (conv_wgrad_dma_kernel): the default whenever BOTH operands are plain (AWR_WGRAD_DMA: 0 = never, 1 = always, unset = that
    // rule; AWR_WGRAD_KP = stage depth in pixels, 16 | 32).  Since its pipelined loop lost its exits (162 -> 100 / 96 -> 69 / 50 -> 36 registers)
    // isolated launches with plain operands gain 5-7 % on every layer shape (profiles/r04_microbench_wgrad_dma.txt) and the ResNet18 step 0.8 %;
    // an operand that still goes through registers (fused BatchNorm loader, bias-gradient column sums) gains nothing in isolation and the
    // Hourglass step (nearly all of whose weight gradients have one) nothing either.
    static const int wdma = env_int("AWR_WGRAD_DMA", -1);
    static const int wkp = env_int("AWR_WGRAD_KP", 0);
    if (wdma && g_staging && wg_products() == 1 && hshift_f32 >= 64) {
        const bool dreg = a->d_scale != nullptr || a->d_colsum != nullptr, greg = a->g_scale != nullptr;
        if (wdma > 0 || (!dreg && !greg)) {
            const int kp = wkp ? wkp : ((TM == 1 && TN == 1) ? 32 : 16);
            launch_wgrad_dma(a, TM, TN, kp, dreg, greg, grid, st, (int)chunk, wshift, hshift_f32);
            return check_launch("conv_wgrad_dma_kernel");
        }
    }
#define AWR_LAUNCH_WGRAD(tm, tn)                                                                                                          \
    do {                                                                                                                                  \
        if (wg_products() == 6) hipLaunchKernelGGL((conv_wgrad_split_kernel<tm, tn, 6>), grid, dim3(256), 0, st, *a, (int)chunk, wshift, hshift);       \
        else if (hshift_f32 >= 64) hipLaunchKernelGGL((conv_wgrad_kernel<tm, tn, true>), grid, dim3(256), 0, st, *a, (int)chunk, wshift, hshift_f32); \
        else hipLaunchKernelGGL((conv_wgrad_kernel<tm, tn>), grid, dim3(256), 0, st, *a, (int)chunk, wshift, hshift_f32);                            \
    } while (0)
    if (TM == 2 && TN == 2) AWR_LAUNCH_WGRAD(2, 2);
    else if (TM == 2 && TN == 1) AWR_LAUNCH_WGRAD(2, 1);
    else if (TM == 1 && TN == 2) AWR_LAUNCH_WGRAD(1, 2);
    else AWR_LAUNCH_WGRAD(1, 1);
#undef AWR_LAUNCH_WGRAD
    return check_launch("conv_wgrad_kernel");
}

// Deterministic mode (split_stride > 0): the caller sums ALL max_split K-chunk copies.  A launch may write fewer than it did when the caller sized them --
// the kernel choice follows the process-wide product / staging modes at LAUNCH time (awr_set_gemm_products / awr_set_gemm_staging after the plan was
// built) -- so the copies this launch leaves untouched are zero-filled here: the sum stays the gradient whatever the modes were switched to.
static int wgrad_clear_unwritten(const awr_wgrad_args* a, int written, void* stream) {
    AWR_REQUIRE(written <= a->max_split, "conv_wgrad: the launch writes %d K-chunk copies, the caller allocated %d", written, a->max_split);
    if (written == a->max_split) return AWR_OK;
    const size_t n = (size_t)(a->max_split - written);
    if (hipMemsetAsync(a->R + (int64_t)written * a->split_stride, 0, n * (size_t)a->split_stride * sizeof(float), as_stream(stream)) != hipSuccess ||
        (a->d_colsum && hipMemsetAsync(a->d_colsum + (int64_t)written * a->Cd, 0, n * (size_t)a->Cd * sizeof(float), as_stream(stream)) != hipSuccess)) {
        set_error("conv_wgrad: clearing the unwritten K-chunk copies failed");
        return AWR_ERR_HIP;
    }
    return AWR_OK;
}

int awr_conv_wgrad(const awr_wgrad_args* a, void* stream) {
    AWR_REQUIRE(a && a->D && a->G && a->R, "conv_wgrad: null pointer");
    const int64_t d_img = (int64_t)a->Hd * a->Wd * a->Cd, g_img = (int64_t)a->Hg * a->Wg * a->Cg;
    int nchunk = 1;
    while ((d_img * (a->B / nchunk) * 4 >= (1LL << 32) || g_img * (a->B / nchunk) * 4 >= (1LL << 32)) && a->B % (nchunk * 2) == 0) nchunk *= 2;
    if (a->split_stride) {
        int written = 0;
        if (int e = awr_conv_wgrad_splits(a, &written)) return e;
        if (int e = wgrad_clear_unwritten(a, written, stream)) return e;
    }
    if (nchunk == 1) return conv_wgrad_one(a, stream);
    for (int c = 0; c < nchunk; ++c) {       // split-K over batch chunks: partial sums accumulate in R
        awr_wgrad_args b = *a;
        b.B = a->B / nchunk;
        b.D = a->D + d_img * b.B * c;
        b.G = a->G + g_img * b.B * c;
        if (a->split_stride) {            // deterministic mode: every batch chunk writes its own range of K-chunk copies
            wgrad_launch w;
            if (int e = wgrad_plan(&b, &w)) return e;
            b.R = a->R + (int64_t)c * w.nsplit * a->split_stride;
            if (a->d_colsum) b.d_colsum = a->d_colsum + (int64_t)c * w.nsplit * a->Cd;
        }
        if (int e = conv_wgrad_one(&b, stream)) return e;
    }
    return AWR_OK;
}

}  // extern "C"
