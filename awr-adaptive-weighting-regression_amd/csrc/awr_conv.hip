// Convolution family, main translation unit: process-wide modes, the forward / data-gradient dispatch (awr_conv_gemm), the register-staged
// GEMM, split-K and fused-pair launches.  Kernels: csrc/awr_conv_kernels.inc; the LDS-DMA GEMM instantiations: csrc/awr_gemm_t*.hip;
// weight gradients: csrc/awr_wgrad.hip.
#include <string.h>

#define AWR_CONV_MAIN_TU
#include "awr_conv_kernels.inc"
#include "awr_conv_modes.h"

namespace awr {
// (plain functions, not lambdas, for the initialisers: hipcc 7.2 initialised a second namespace-scope `static int g = []() { ... }();` of one
// translation unit with the FIRST lambda's body -- DESIGN.md 5, side finding of round 3)
int g_force_tm = 0, g_force_tn = 0, g_products = env_int("AWR_GEMM_PRODUCTS", 1);
// Split-operand mode, weight gradients.  AWR_WGRAD_SPLIT=0 (round-5 study, profiles/r05_split_mode_studies.txt) runs them on the FP32-MFMA kernels instead
// (kernel row, LDS-DMA per tap: exact fp32 products as well, and FASTER in isolation than the split-operand kernel whose staged rows are transposed and
// cut in registers) -- and the step gets SLOWER: 10.77-10.91 -> 11.39-11.46 ms (a weight gradient runs beside the bf16 data-gradient chain; an FP32-MFMA
// launch holds the matrix pipe three times as long per product).  Default: the split-operand weight gradient.
int g_wgrad_split = env_int("AWR_WGRAD_SPLIT", 1);
int g_staging = env_int("AWR_DMA", 2);
int g_accum = env_int("AWR_ACCUM", 2);      // auto
int g_accum_auto_k = 576;       // accum = 2 (auto): launches whose K extent reaches this many terms accumulate blocked ...
int g_accum_auto_dgrad = 0;      // ... forward launches only (0) or data gradients too (1)
// launch-path knobs, read ONCE (a getenv per launch is neither cheap nor safe against a concurrent setenv); tests and same-box A/Bs flip them
// through awr_debug_set_knob
int g_knob_deep = env_int("AWR_DEEP", 1), g_knob_deep_1x1 = env_int("AWR_DEEP_1X1", 0), g_knob_fast_stats = env_int("AWR_FAST_STATS", 1);
// the LDS-DMA GEMM, one instantiation per workgroup-tile shape (csrc/awr_gemm_t*.hip)
template <int TM, int TN>
void launch_dma_tile(const awr_conv_args* a, dim3 grid, hipStream_t st, int mode, int aff, bool epre, int em);
extern template void launch_dma_tile<1, 1>(const awr_conv_args*, dim3, hipStream_t, int, int, bool, int);
extern template void launch_dma_tile<1, 2>(const awr_conv_args*, dim3, hipStream_t, int, int, bool, int);
extern template void launch_dma_tile<2, 1>(const awr_conv_args*, dim3, hipStream_t, int, int, bool, int);
extern template void launch_dma_tile<2, 2>(const awr_conv_args*, dim3, hipStream_t, int, int, bool, int);
}  // namespace awr

using namespace awr;

static void launch_dma(const awr_conv_args* a, int TM, int TN, dim3 grid, hipStream_t st, int mode, int aff, bool epre, int em) {
    if (TM == 2 && TN == 2) launch_dma_tile<2, 2>(a, grid, st, mode, aff, epre, em);
    else if (TM == 2 && TN == 1) launch_dma_tile<2, 1>(a, grid, st, mode, aff, epre, em);
    else if (TM == 1 && TN == 2) launch_dma_tile<1, 2>(a, grid, st, mode, aff, epre, em);
    else launch_dma_tile<1, 1>(a, grid, st, mode, aff, epre, em);
}

extern "C" {

int awr_debug_force_tile(int tm, int tn) {
    AWR_REQUIRE((tm == 0 && tn == 0) || ((tm == 1 || tm == 2) && (tn == 1 || tn == 2)), "force_tile: tm,tn must be 0,0 or in {1,2}");
    g_force_tm = tm;
    g_force_tn = tn;
    return AWR_OK;
}

int awr_debug_set_knob(const char* name, int value) {
    AWR_REQUIRE(name, "debug_set_knob: NULL name");
    int* k = !strcmp(name, "deep") ? &g_knob_deep : !strcmp(name, "deep_1x1") ? &g_knob_deep_1x1 : !strcmp(name, "fast_stats") ? &g_knob_fast_stats : nullptr;
    AWR_REQUIRE(k, "debug_set_knob: unknown knob '%s' (deep, deep_1x1, fast_stats)", name);
    *k = value;
    return AWR_OK;
}

int awr_set_gemm_products(int n) {
    AWR_REQUIRE(n == 1 || n == 6, "gemm_products: 1 (f32 MFMA) or 6 (3-way bf16 split, 6 bf16 MFMA products per fp32 product)");
    g_products = n;
    return AWR_OK;
}

int awr_get_gemm_products(void) { return g_products; }

int awr_get_wgrad_products(void) { return wg_products(); }

int awr_set_gemm_staging(int mode) {
#ifdef AWR_DMA_STUDY
    AWR_REQUIRE(mode >= 0 && mode <= 3, "gemm_staging: 0 (registers), 1 / 2 / 3 (LDS-DMA study variants)");
#else
    AWR_REQUIRE(mode == 0 || mode == 2, "gemm_staging: 0 (global -> registers -> LDS) or 2 (LDS-DMA, the default)");
#endif
    g_staging = mode;
    return AWR_OK;
}

int awr_get_gemm_staging(void) { return g_staging; }

int awr_set_gemm_accum(int mode) {
    AWR_REQUIRE(mode >= 0 && mode <= 2, "gemm_accum: 0 (ordered), 1 (blocked: restart every 128 k) or 2 (auto: blocked where the K extent is long)");
    g_accum = mode;
    return AWR_OK;
}

int awr_get_gemm_accum(void) { return g_accum; }

int awr_set_gemm_accum_auto(int min_k, int dgrad) {
    AWR_REQUIRE(min_k >= 256, "gemm_accum_auto: the threshold is a K extent >= 256 (shorter extents are one block anyway)");
    g_accum_auto_k = min_k;
    g_accum_auto_dgrad = dgrad != 0;
    return AWR_OK;
}

int awr_get_gemm_accum_auto(int* min_k, int* dgrad) {
    if (min_k) *min_k = g_accum_auto_k;
    if (dgrad) *dgrad = g_accum_auto_dgrad;
    return AWR_OK;
}

int awr_resolve_gemm_accum(int k_extent, int kind) {
    if (g_accum != 2) return g_accum;
    if (kind != AWR_GEMM_FORWARD && !(kind == AWR_GEMM_DGRAD && g_accum_auto_dgrad)) return 0;
    return (g_products == 1 && g_staging != 0 && k_extent >= g_accum_auto_k) ? 1 : 0;
}

static int conv_gemm_one(const awr_conv_args* a, void* stream);

// Tensors above 4 GB (Hourglass stem-resolution maps at batch 128) exceed the 32-bit buffer offsets of the kernels: the
// batch is processed in power-of-two chunks, each an independent launch on the same stream (images are independent rows
// of the GEMM; the BatchNorm statistic atomics simply accumulate across chunks).
int awr_conv_gemm(const awr_conv_args* a, void* stream) {
    AWR_REQUIRE(a && a->in && a->w && a->out, "conv_gemm: null pointer");
    const int64_t in_img = (int64_t)a->Hin * a->Win * a->Cin, out_img = (int64_t)a->Hout * a->Wout * a->N;
    const int in_b = a->in_split ? 6 : 4;      // bytes per input element the kernel addresses (pre-cut image: three bf16 pieces)
    int nchunk = 1;
    while ((in_img * (a->B / nchunk) * in_b >= (1LL << 32) || out_img * (a->B / nchunk) * 4 >= (1LL << 32)) && a->B % (nchunk * 2) == 0) nchunk *= 2;
    if (nchunk == 1) return conv_gemm_one(a, stream);
    for (int c = 0; c < nchunk; ++c) {
        awr_conv_args b = *a;
        b.B = a->B / nchunk;
        b.in = a->in + in_img * b.B * c;
        if (a->in_split) b.in_split = static_cast<const char*>(a->in_split) + in_img * b.B * c * 6;
        if (a->pool_out) b.pool_out = a->pool_out + out_img / 4 * b.B * c;
        b.out = a->out + out_img * b.B * c;
        if (a->res) b.res = a->res + out_img * b.B * c;
        if (a->bnr_y) b.bnr_y = a->bnr_y + out_img * b.B * c;
        if (a->bnr_act) b.bnr_act = a->bnr_act + out_img * b.B * c;
        if (a->bnr2_y) b.bnr2_y = a->bnr2_y + out_img * b.B * c;
        if (a->in2 && a->w2) {       // fused pair: `in` holds Cin channels (advanced above), `in2` the N1x extra channels of the second GEMM
            b.in2 = a->in2 + (int64_t)a->Hin * a->Win * a->N1x * b.B * c;
        } else if (a->in2) {         // two-tensor K extent: Cin1 channels in `in`, the rest in `in2`
            b.in = a->in + (int64_t)a->Hin * a->Win * a->Cin1 * b.B * c;
            b.in2 = a->in2 + (int64_t)a->Hin * a->Win * (a->Cin - a->Cin1) * b.B * c;
        }
        if (a->in_bnb_y) b.in_bnb_y = a->in_bnb_y + in_img * b.B * c;
        if (a->stat_slots > 0) b.stat_slot_base = a->stat_slot_base + c * (a->stat_slots / nchunk);
        if (int e = conv_gemm_one(&b, stream)) return e;
    }
    return AWR_OK;
}

#ifdef AWR_STUDY      // study form (round 5, measured slower on every shape: profiles/r05_half_batch_wavefront.txt); not in the default library
// One of `nparts` equal batch parts of the launch `a` describes (images [part * B / nparts, (part + 1) * B / nparts)): images are independent rows of
// the GEMM, so the parts are independent launches that may be issued at different points of a stream (the half-batch BatchNorm-backward wavefront of
// the training plans: the data gradient of half A runs while half B's d(y) is still being written).  Deterministic mode: part p's workgroups take the
// statistics slots from p * stat_slots / nparts on.
int awr_conv_gemm_part(const awr_conv_args* a, int nparts, int part, void* stream) {
    AWR_REQUIRE(a && nparts >= 1 && part >= 0 && part < nparts && a->B % nparts == 0 && !a->w2 && !(a->partial && a->split_max > 1),
                "conv_gemm_part: part %d of %d of a batch of %d (no fused pair / split-K)", part, nparts, a ? a->B : 0);
    if (nparts == 1) return awr_conv_gemm(a, stream);
    const int64_t in_img = (int64_t)a->Hin * a->Win * a->Cin, out_img = (int64_t)a->Hout * a->Wout * a->N;
    awr_conv_args b = *a;
    b.B = a->B / nparts;
    const int64_t ioff = in_img * b.B * part, ooff = out_img * b.B * part;
    b.in = a->in + ioff;
    if (a->in_split) b.in_split = static_cast<const char*>(a->in_split) + ioff * 6;
    b.out = a->out + ooff;
    if (a->res) b.res = a->res + ooff;
    if (a->bnr_y) b.bnr_y = a->bnr_y + ooff;
    if (a->bnr_act) b.bnr_act = a->bnr_act + ooff;
    if (a->bnr2_y) b.bnr2_y = a->bnr2_y + ooff;
    if (a->in2) {         // two-tensor K extent: Cin1 channels in `in`, the rest in `in2`
        b.in = a->in + (int64_t)a->Hin * a->Win * a->Cin1 * b.B * part;
        b.in2 = a->in2 + (int64_t)a->Hin * a->Win * (a->Cin - a->Cin1) * b.B * part;
    }
    if (a->in_bnb_y) b.in_bnb_y = a->in_bnb_y + ioff;
    if (a->stat_slots > 0) b.stat_slot_base = a->stat_slot_base + part * (a->stat_slots / nparts);      // (slot = (base + workgroup) % stat_slots, as in the > 4 GB chunk loop)
    return awr_conv_gemm(&b, stream);
}
#endif

static int conv_gemm_one(const awr_conv_args* a_in, void* stream) {
    // output-store policy (awr_conv_args.out_nt): 0 = automatic -> streaming (`buffer_store ... nt`, the epilogue's operand loads too) when the output tensor is
    // at least as large as the 256 MB Infinity Cache -- its consumer fetches it from HBM either way -- and the K extent is short (<= 512: the launches whose
    // 32 KB tile per workgroup follows 4 ... 32 stages of operand traffic; isolated 1x1 launches +5-12 %, long-K launches unmoved).  Measured per step
    // (profiles/r05_nt_policy.txt): config 5 288 -> 283 ms, ResNet18 batch 256 and config 3 -0.3 %, batch-64 steps within noise; with EVERY store streaming
    // ResNet18 loses 0.5 % (its 67-134 MB activations otherwise reach their consumer from the cache).  AWR_NT_MIN_MB (0 = never) / AWR_NT_MAX_K: the A/B knobs
    awr_conv_args a_res = *a_in;
    const awr_conv_args* a = &a_res;
    AWR_REQUIRE(a_in->out_nt >= 0 && a_in->out_nt <= 2, "conv_gemm: out_nt=%d", a_in->out_nt);
    if (a_res.out_nt == 0) {
        static const int64_t min_mb = env_int("AWR_NT_MIN_MB", 256), max_k = env_int("AWR_NT_MAX_K", 512);
        const int64_t obytes = (int64_t)a_res.B * a_res.Hout * a_res.Wout * a_res.N * 4, kext = (int64_t)a_res.Cin * a_res.ph[0].ntaps;
        a_res.out_nt = (min_mb > 0 && obytes >= (min_mb << 20) && kext <= max_k) ? 2 : 1;
    }
    AWR_REQUIRE(a->Cin > 0 && a->Cin % BK == 0, "conv_gemm: Cin=%d must be a positive multiple of %d", a->Cin, BK);
    AWR_REQUIRE(a->accum == 0 || a->accum == 1, "conv_gemm: accum=%d", a->accum);
    AWR_REQUIRE(!a->in_bnb_y || (a->in_bnb_coef && !a->in_scale && !a->relu_in && !a->in2 && !a->w2 && g_products == 1 && g_staging != 0 &&
                                 a->Cin <= AFF_MAXC && !(a->partial && a->split_k > 1)),
                "conv_gemm: an un-materialised BatchNorm-backward input (in_bnb_y) needs in_bnb_coef, the FP32-MFMA mode with LDS-DMA staging, Cin <= %d, "
                "and no other input arithmetic / second tensor / fused pair / split-K", AFF_MAXC);
    AWR_REQUIRE(a->nphase >= 1 && a->nphase <= 4, "conv_gemm: nphase=%d", a->nphase);
    AWR_REQUIRE(g_products == 1 || a->w_split, "conv_gemm: the %d-product mode needs the split image of the weights (w_split)", g_products);
    AWR_REQUIRE(a->B > 0 && a->Hq > 0 && a->Wq > 0 && a->N > 0 && a->T > 0 && a->so >= 1 && a->si >= 1, "conv_gemm: bad geometry");
    AWR_REQUIRE(a->N % 4 == 0, "conv_gemm: N=%d must be a multiple of 4 (16-byte output rows)", a->N);
    AWR_REQUIRE(!a->bnr_y || (a->bnr_coef && a->stats && (!a->res || (a->bnr_act && a->res == a->out))),
                "conv_gemm: fused BN-backward reduction needs coef + stats; accumulating (res) only in place and with bnr_act");
    AWR_REQUIRE(!a->bnr_act || a->bnr_y, "conv_gemm: bnr_act without bnr_y");
    AWR_REQUIRE(!a->bnr_y || (!a->bias && !a->out_scale), "conv_gemm: the fused BatchNorm-backward reduction belongs to a data gradient: no bias / output affine");
    // epilogue forms no kernel variant implements (they used to be downgraded silently by the LDS-DMA dispatch)
    AWR_REQUIRE(!a->in_bnb_y || !a->stats || a->bnr_y, "conv_gemm: an un-materialised BatchNorm-backward input (in_bnb_y) has no statistics-only epilogue");
    AWR_REQUIRE(!(a->in2 && !a->w2 && a->bnr_y), "conv_gemm: the two-tensor K extent (in2) has no fused BatchNorm-backward reduction epilogue");
    // blocked accumulation is a property of the LDS-DMA kernel: fail instead of returning ordered results under the parity flag
    AWR_REQUIRE(a->accum == 0 || (g_products == 1 && g_staging != 0 && !a->w2 && !(a->partial && (a->split_k > 1 || a->split_max > 1))),
                "conv_gemm: accum = 1 (blocked accumulation) needs the FP32-MFMA mode with LDS-DMA staging and no fused pair / split-K scratch");
    AWR_REQUIRE(!a->bnr2_y || (a->bnr_y && a->bnr2_coef && a->stats2), "conv_gemm: a second fused reduction (bnr2_y) needs bnr_y, bnr2_coef and stats2");
    AWR_REQUIRE((a->in_scale == nullptr) == (a->in_shift == nullptr), "conv_gemm: in_scale/in_shift must come together");
    AWR_REQUIRE((a->out_scale == nullptr) == (a->out_shift == nullptr), "conv_gemm: out_scale/out_shift must come together");
    AWR_REQUIRE(!a->pool_out || a->w2, "conv_gemm: pool_out belongs to the fused pair (w2)");
    AWR_REQUIRE(!a->in2 || a->w2 || (g_products == 1 && a->nphase == 1 && a->ph[0].ntaps == 1 && a->T == 1 && a->Cin1 > 0 && a->Cin1 < a->Cin && a->Cin1 % BK == 0),
                "conv_gemm: a second input tensor needs the FP32-MFMA mode, one tap and 0 < Cin1 < Cin, Cin1 %% 32 == 0");
    for (int p = 0; p < a->nphase; ++p) {
        AWR_REQUIRE(a->ph[p].ntaps >= 1 && a->ph[p].ntaps <= 16, "conv_gemm: phase %d has %d taps", p, a->ph[p].ntaps);
        for (int t = 0; t < a->ph[p].ntaps; ++t) AWR_REQUIRE((a->ph[p].tap[t] >> 16) >= 0 && (a->ph[p].tap[t] >> 16) < a->T, "conv_gemm: tap index out of range");
    }
    const int64_t M = (int64_t)a->B * a->Hq * a->Wq;
    AWR_REQUIRE(M < (1LL << 31), "conv_gemm: too many output pixels");
    if (a->w2) {      // two convolutions back to back (FUSE2): the tile's N extent is the first conv's channel count
        AWR_REQUIRE(g_products == 1 && (a->N1 == 128 || a->N1 == 64) && a->N == 2 * a->N1 && a->so == 1 && a->nphase == 1 && !a->stats && !a->bnr_y &&
                        !a->partial && a->split_k <= 1 && a->N1x >= 0 && a->N1x % BK == 0 && (a->N1x == 0) == (a->in2 == nullptr) && !(a->N1x && a->res),
                    "conv_gemm: the fused pair (w2) needs the FP32-MFMA mode, 64 or 128 intermediate channels, N == 2 N1, a stride-1 output, "
                    "no stats / bnr_y / split-K, and either a residual or a second input (N1=%d, N=%d, N1x=%d)", a->N1, a->N, a->N1x);
        AWR_REQUIRE((int64_t)a->B * a->Hout * a->Wout * a->N * 4 < (1LL << 32) && (int64_t)a->B * a->Hin * a->Win * a->Cin * 4 < (1LL << 32) &&
                        M * (int64_t)a->N1x * 4 < (1LL << 32),
                    "conv_gemm: tensors must stay below 4 GB (32-bit buffer offsets)");
        const dim3 grid2((unsigned)((M + (a->N1 == 64 && a->tile_m == 2 ? 127 : 63)) / (a->N1 == 64 && a->tile_m == 2 ? 128 : 64)), 1);
        // first GEMM on LDS-DMA staging (round 5) whenever its input needs no arithmetic; AWR_DMA=0 / AWR_FUSE2_DMA=0: the register-staged pair
        static const int pair_dma = env_int("AWR_FUSE2_DMA", 1);
        if (a->pool_out) {      // the pair also writes the 2x2 max-pool of its output: 2D workgroup tiles (two image rows x BM / 2 columns)
            const int bm = (a->N1 == 64 && a->tile_m == 2) ? 128 : 64;
            AWR_REQUIRE(pair_dma && g_staging != 0 && !a->in_scale && !a->relu_in && a->Hq % 2 == 0 && a->Wq % (bm / 2) == 0 && a->si == 1,
                        "conv_gemm: pool_out needs the LDS-DMA pair, a plain input, an even map height and a width that is a multiple of %d", bm / 2);
            if (a->N1 == 128) hipLaunchKernelGGL((conv_gemm_dma_pair_kernel<1, 2, true>), grid2, dim3(256), 0, as_stream(stream), *a);
            else if (a->tile_m == 2) hipLaunchKernelGGL((conv_gemm_dma_pair_kernel<2, 1, true>), grid2, dim3(256), 0, as_stream(stream), *a);
            else hipLaunchKernelGGL((conv_gemm_dma_pair_kernel<1, 1, true>), grid2, dim3(256), 0, as_stream(stream), *a);
            return check_launch("conv_gemm_dma_pair_kernel<pool>");
        }
        if (pair_dma && g_staging != 0 && !a->in_scale && !a->relu_in) {
            if (a->N1 == 128) hipLaunchKernelGGL((conv_gemm_dma_pair_kernel<1, 2>), grid2, dim3(256), 0, as_stream(stream), *a);
            else if (a->tile_m == 2) hipLaunchKernelGGL((conv_gemm_dma_pair_kernel<2, 1>), grid2, dim3(256), 0, as_stream(stream), *a);
            else hipLaunchKernelGGL((conv_gemm_dma_pair_kernel<1, 1>), grid2, dim3(256), 0, as_stream(stream), *a);
            return check_launch("conv_gemm_dma_pair_kernel");
        }
        if (a->N1 == 128) hipLaunchKernelGGL((conv_gemm_kernel<1, 2, 0, false, false, false, true>), grid2, dim3(256), 0, as_stream(stream), *a);
        else if (a->tile_m == 2) hipLaunchKernelGGL((conv_gemm_kernel<2, 1, 0, false, false, false, true>), grid2, dim3(256), 0, as_stream(stream), *a);
        else hipLaunchKernelGGL((conv_gemm_kernel<1, 1, 0, false, false, false, true>), grid2, dim3(256), 0, as_stream(stream), *a);
        return check_launch("conv_gemm_kernel<fused pair>");
    }
    AWR_REQUIRE((int64_t)a->B * a->Hin * a->Win * a->Cin * 4 < (1LL << 32) && (int64_t)a->B * a->Hout * a->Wout * a->N * 4 < (1LL << 32),
                "conv_gemm: tensors must stay below 4 GB (32-bit buffer offsets)");
    // Tile choice (measured, tools/microbench_gemm.py): 64-row tiles win on every ResNet18/Hourglass layer shape --
    // 3-4 workgroups per CU de-synchronise prologue/epilogue bubbles that two lock-stepped 128x128 workgroups
    // expose, and the extra L2 traffic is free at FP32-MFMA rates.  128 columns when N allows and the grid stays
    // >= 2 workgroups per CU, else 64x64.
    auto blocks = [&](int tm, int tn) { return ((M + 64 * tm - 1) / (64 * tm)) * ((a->N + 64 * tn - 1) / (64 * tn)) * a->nphase; };
    int TM = 1, TN = (a->N > 64 && blocks(1, 2) >= 512) ? 2 : 1;
    if (a->Cin * a->ph[0].ntaps <= 64) TM = 2;     // one or two K-slices (the im2col'd stem): store-bound, amortise the epilogue
    if (a->tile_m) {
        AWR_REQUIRE((a->tile_m == 1 || a->tile_m == 2) && (a->tile_n == 1 || a->tile_n == 2), "conv_gemm: tile_m/tile_n must be 1 or 2");
        TM = a->tile_m;
        TN = a->tile_n;
    }
    if (g_force_tm) { TM = g_force_tm; TN = g_force_tn; }
    hipStream_t st = as_stream(stream);
    // split-K: few workgroups with a long K loop (low-batch inference: a layer4 conv at batch 4 is 32 workgroups x 144 slices)
    int S = 1;
    if (a->partial && a->split_max > 1 && g_products == 1 && !a->in2 && !a->stats && !a->bnr_y) {
        int minsteps = a->ph[0].ntaps;
        for (int p = 1; p < a->nphase; ++p) minsteps = a->ph[p].ntaps < minsteps ? a->ph[p].ntaps : minsteps;
        minsteps *= a->Cin / BK;
        if (a->split_k > 0) {
            S = a->split_k;
        } else {      // heuristic: fill ~2 workgroups per CU, keep >= 8 slices per range
            const int64_t nb = blocks(TM, TN);
            while (S * 2 <= a->split_max && nb * S * 2 <= 512 && minsteps / (S * 2) >= 8) S *= 2;
        }
        AWR_REQUIRE(S >= 1 && S <= a->split_max, "conv_gemm: split_k=%d exceeds split_max=%d", S, a->split_max);
    } else {      // (the split-operand mode ignores a split-K request: a plan keeps its scratch and depth across mode switches)
        AWR_REQUIRE(a->split_k <= 1 || (a->partial && a->split_max > 1 && !a->in2 && !a->stats && !a->bnr_y),
                    "conv_gemm: split_k needs `partial` scratch and no stats / bnr_y / in2");
    }
    if (S > 1) {
        const dim3 grid((unsigned)(blocks(TM, TN) / a->nphase), a->nphase, S);
        if (TM == 2 && TN == 2) hipLaunchKernelGGL((conv_gemm_kernel<2, 2, 0, false, false, true>), grid, dim3(256), 0, st, *a);
        else if (TM == 2 && TN == 1) hipLaunchKernelGGL((conv_gemm_kernel<2, 1, 0, false, false, true>), grid, dim3(256), 0, st, *a);
        else if (TM == 1 && TN == 2) hipLaunchKernelGGL((conv_gemm_kernel<1, 2, 0, false, false, true>), grid, dim3(256), 0, st, *a);
        else hipLaunchKernelGGL((conv_gemm_kernel<1, 1, 0, false, false, true>), grid, dim3(256), 0, st, *a);
        if (int e = check_launch("conv_gemm_kernel<split>")) return e;
        const int64_t numel = (int64_t)a->B * a->Hout * a->Wout * a->N, n4 = numel / 4;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, a->partial, S, numel, a->bias, a->out_scale, a->out_shift,
                           a->res, a->relu_out, n4, a->N / 4, a->out);
        return check_launch("splitk_reduce_kernel");
    }
    const dim3 grid((unsigned)(blocks(TM, TN) / a->nphase), a->nphase);
    const bool aff = a->in_scale != nullptr || a->relu_in;
#ifdef AWR_STUDY      // study form (round 5: 15-40 % slower than the in-register cut, profiles/r05_split_mode_studies.txt); not in the default library
    // split-operand mode with a PRE-CUT activation image (in_split): both operands by LDS-DMA (conv_gemm_sdma_body)
    static const int sdma_on = env_int("AWR_SPLIT_DMA", 1);      // 0: never (the same-box A/B hook back to the in-kernel cut)
    if (g_products == 6 && a->in_split && sdma_on) {
        AWR_REQUIRE(!aff && !a->in2 && !a->in_bnb_y && !(a->partial && a->split_k > 1),
                    "conv_gemm: a pre-cut activation image (in_split) excludes input arithmetic (in_scale / relu_in), a second tensor and split-K");
        AWR_REQUIRE((int64_t)a->B * a->Hin * a->Win * a->Cin * 6 < (1LL << 32), "conv_gemm: the pre-cut image must stay below 4 GB (32-bit buffer offsets)");
        const int em = a->bnr_y ? ((a->bnr_act || a->res || a->bnr2_y) ? 4 : 3) : a->stats ? 2 : 1;
#define AWR_SDMA_EM(tm, tn)                                                                                              \
        do {                                                                                                             \
            if (em == 4) hipLaunchKernelGGL((conv_gemm_sdma_kernel<tm, tn, 4>), grid, dim3(256), 0, st, *a);             \
            else if (em == 3) hipLaunchKernelGGL((conv_gemm_sdma_kernel<tm, tn, 3>), grid, dim3(256), 0, st, *a);        \
            else if (em == 2) hipLaunchKernelGGL((conv_gemm_sdma_kernel<tm, tn, 2>), grid, dim3(256), 0, st, *a);        \
            else hipLaunchKernelGGL((conv_gemm_sdma_kernel<tm, tn, 1>), grid, dim3(256), 0, st, *a);                     \
        } while (0)
        if (TM == 2 && TN == 2) AWR_SDMA_EM(2, 2);
        else if (TM == 2 && TN == 1) AWR_SDMA_EM(2, 1);
        else if (TM == 1 && TN == 2) AWR_SDMA_EM(1, 2);
        else AWR_SDMA_EM(1, 1);
#undef AWR_SDMA_EM
        return check_launch("conv_gemm_sdma_kernel");
    }
#else
    AWR_REQUIRE(!a->in_split, "conv_gemm: pre-cut activation images (in_split) are a study form: build the library with -DAWR_STUDY");
#endif
    // short K loops (<= 8 slices) whose epilogue reads exactly one operand tensor: that tensor's rows are requested ahead (EPRE)
    static const bool no_epre = getenv("AWR_NO_EPRE") != nullptr;      // same-box A/B hook
    const bool epre = !no_epre && g_products == 1 && !a->in2 && a->nphase == 1 && a->ph[0].ntaps * (a->Cin / BK) <= 8 &&
                      ((a->res != nullptr) != (a->bnr_y != nullptr)) && !a->bnr_act && !a->bnr2_y;
#define AWR_LAUNCH_GEMM(tm, tn)                                                                          \
    do {                                                                                                 \
        if (epre) hipLaunchKernelGGL((conv_gemm_kernel<tm, tn, 0, false, false, false, false, true>), grid, dim3(256), 0, st, *a);  \
        else if (a->in2) hipLaunchKernelGGL((conv_gemm_kernel<tm, tn, 0, false, true>), grid, dim3(256), 0, st, *a);             \
        else if (g_products == 6 && aff) hipLaunchKernelGGL((conv_gemm_kernel<tm, tn, 6, true>), grid, dim3(256), 0, st, *a);   \
        else if (g_products == 6) hipLaunchKernelGGL((conv_gemm_kernel<tm, tn, 6, false>), grid, dim3(256), 0, st, *a);    \
        else hipLaunchKernelGGL((conv_gemm_kernel<tm, tn, 0, false>), grid, dim3(256), 0, st, *a);                         \
    } while (0)
    // LDS-DMA staging (conv_gemm_dma_body) is the default of the FP32-MFMA mode: 16-float stages, double-buffered.  awr_set_gemm_staging(0) /
    // AWR_DMA=0 is the same-box A/B hook back to the register-staged kernel; builds with -DAWR_DMA_STUDY also carry 1 = 32-float stages x 2 and
    // 3 = 32-float stage x 1.
    const int dma_mode = g_staging;
    if (dma_mode && g_products == 1 && ((!aff && !a->in_bnb_y) || (a->in2 ? a->Cin1 : a->Cin) <= AFF_MAXC)) {
        const int em = a->bnr_y ? ((a->bnr_act || a->res || a->bnr2_y) ? 4 : 3) : a->stats ? 2 : 1;
        // single-tap launches whose tap is (0, 0) never read outside the image: the fused input affine without its padding selects
        bool pad_free = true;
        for (int p = 0; p < a->nphase && pad_free; ++p) pad_free = a->ph[p].ntaps == 1 && (a->ph[p].tap[0] & 0xffff) == 0;
        launch_dma(a, TM, TN, grid, st, dma_mode, a->in_bnb_y ? 2 : aff ? (pad_free ? 3 : 1) : 0, epre && !a->in_bnb_y, em);
        return check_launch("conv_gemm_dma_kernel");
    }
    static const bool occ6 = getenv("AWR_NO_OCC6") == nullptr;
    if (TM == 1 && TN == 1 && occ6 && g_products == 1 && !a->in2 && !epre) hipLaunchKernelGGL(conv_gemm_kernel_11_occ6, grid, dim3(256), 0, st, *a);
    else if (TM == 2 && TN == 2) AWR_LAUNCH_GEMM(2, 2);
    else if (TM == 2 && TN == 1) AWR_LAUNCH_GEMM(2, 1);
    else if (TM == 1 && TN == 2) AWR_LAUNCH_GEMM(1, 2);
    else AWR_LAUNCH_GEMM(1, 1);
#undef AWR_LAUNCH_GEMM
    return check_launch("conv_gemm_kernel");
}

}  // extern "C"
