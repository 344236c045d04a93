/*
 * awr_hip.h -- C ABI of libawr_hip.so: the MI355X (gfx950) implementation of the AWR hot path.
 *
 * The reference (Elody-07/AWR-Adaptive-Weighting-Regression) has no FFI layer: its hot path is
 * plain Python over torch ops.  Each entry point below therefore cites the reference Python
 * interface it replaces (file:line under the reference tree).  The Python host side in
 * awr-adaptive-weighting-regression_amd/ binds these with ctypes and mirrors the reference's
 * call surface (get_deconv_net, PoseNet, FeatureModule, My_SmoothL1Loss, Trainer step).
 *
 * Conventions
 *   - every function returns 0 on success, a negative code otherwise; awr_last_error() returns a
 *     thread-local description.  Nothing throws across the ABI.
 *   - all pointers are DEVICE pointers to fp32 unless stated; the caller owns every buffer.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).  One call = kernels enqueued
 *     on that stream; no call synchronises the device.
 *   - activations inside the backbone are NHWC; the reference-facing tensors (depth image, dense
 *     offset map, joints) keep the reference's NCHW / (B,J,3) layouts.
 */
#ifndef AWR_HIP_H
#define AWR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AWR_OK 0
#define AWR_ERR_ARG (-1)
#define AWR_ERR_HIP (-2)
#define AWR_ERR_UNSUPPORTED (-3)

/* BatchNorm statistic accumulators are [AWR_STAT_SLOTS][2][C] doubles: producers spread their atomics over
 * the slots (less L2 serialisation), awr_bn_finalize sums the slots and zeroes them.  Every entry point that touches
 * such an accumulator takes `nslots` (0 = AWR_STAT_SLOTS): the number of copies the caller allocated. */
#define AWR_STAT_SLOTS 16
/* upper bound of the workgroup count of awr_channel_stats / awr_bn_bwd_reduce launches */
#define AWR_REDUCE_MAX_BLOCKS 1024

/* Deterministic mode (process-wide flag, default 0 or $AWR_DETERMINISTIC): the library itself only stores it; callers
 * that see it set give every producer workgroup its OWN accumulator copy, which makes all results independent of the
 * order in which workgroups finish:
 *   - statistic accumulators ([nslots][2][C] doubles): pass nslots >= the producer's workgroup count (a single add onto a
 *     zeroed copy is exact); the finalize kernels sum the copies in a fixed order;
 *   - weight gradients: awr_wgrad_args.split_stride > 0 makes K-chunk y STORE its partial tile at R + y*split_stride
 *     (and its bias column sums at d_colsum + y*Cd) instead of atomically adding into R; awr_conv_wgrad_splits tells how
 *     many copies a launch writes and awr_unpack_job.slots sums them in order.
 * Cost: see DESIGN.md / profiles. */
int awr_set_deterministic(int on);
int awr_get_deterministic(void);

int awr_version(void);
const char* awr_last_error(void);
/* number of compute units / device name of the current device (diagnostics for bench.py) */
int awr_device_info(int* n_cu, int* clock_mhz, char* name, int name_len);

/* ------------------------------------------------------------------------------------------
 * AWR head.  offset: (B,4J,F,F) NCHW, channels [0,3J) unit offsets (joint-major, xyz-minor),
 * [3J,4J) closeness heat maps.  img: (B,1,H,H) normalised depth; the head samples img[b,0,y*H/F,
 * x*H/F] (nearest F.interpolate).  jt: (B,J,3).
 * -----------------------------------------------------------------------------------------*/

/* replaces FeatureModule.offset2joint_softmax (util/feature_tool.py:41-65).
 * stat (optional, B*J*2): per (b,j) softmax max-logit and sum-exp, consumed by the backward. */
int awr_head_forward(const float* offset, const float* img, int B, int J, int F, int H, float ks,
                     float* jt, float* stat, void* stream);

/* replaces autograd's backward of offset2joint_softmax (implicit in train.py:130).
 * g_offset (B,4J,F,F) = d(sum(jt*g_jt))/d(offset); accumulate!=0 adds into g_offset. */
int awr_head_backward(const float* offset, const float* img, const float* jt, const float* stat,
                      const float* g_jt, int B, int J, int F, int H, float ks, float* g_offset,
                      int accumulate, void* stream);

/* replaces FeatureModule.joint2offset (util/feature_tool.py:12-39): GT dense map (B,4J,F,F). */
int awr_joint2offset(const float* jt_gt, const float* img, int B, int J, int F, int H, float ks,
                     float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Losses.  My_SmoothL1Loss (model/loss.py:8-25) == Huber(delta) averaged over all elements.
 * Partial sums are accumulated in a device double (`acc`); the caller zeroes it (awr_zero_f64)
 * and reads it through awr_loss_finalize.  (In deterministic mode the 8 bytes hold a 2^-50 fixed-point
 * integer and the partials are added with integer atomics: order-independent.  Only
 * awr_loss_finalize may interpret `acc`.)
 * -----------------------------------------------------------------------------------------*/

/* acc[0] += weight * mean(huber(x-y)); gx (optional) = weight * clamp(x-y,+-delta)/n
 * (accumulate!=0 adds into gx).  Replaces criterion(x, y) + its autograd (train.py:125-130). */
int awr_huber(const float* x, const float* y, int64_t n, float delta, float weight, double* acc,
              float* gx, int accumulate, void* stream);

/* Fused GT-map synthesis + dense Huber forward + backward: never materialises the GT map.
 * Equivalent to weight*criterion(offset_pred, FM.joint2offset(jt_gt, img, ks, F)) and its
 * gradient w.r.t. offset_pred (train.py:113, :126, :130).  g_offset optional. */
int awr_dense_loss(const float* offset_pred, const float* jt_gt, const float* img, int B, int J,
                   int F, int H, float ks, float delta, float weight, double* acc, float* g_offset,
                   int accumulate, void* stream);

/* NHWC forms (round 3) for a host that keeps the dense map in the backbone's own layout: pred / grad are (B, F*F, Cp) rows, Cp = 4J
 * rounded up to 32 (channel 3j+c = offset component c of joint j, 3J+j = its heat map; padding channels zero), as awr_plan_head_nhwc
 * hands them out -- no NCHW transposes on either side of the loss.  scratch: awr_head_nhwc_scratch(B, J, F) floats.
 * awr_head_forward_nhwc = FeatureModule.offset2joint_softmax (util/feature_tool.py:41-65).
 * awr_head_loss_step_nhwc = train.py:118-127 in one call: joints + (max, sum) statistics, the coordinate Huber loss (acc[0] +=
 * coord_weight * mean) and, when coord_weight != 0, its gradient through the head; the fused GT map + dense Huber loss (acc[1] +=
 * dense_weight * mean; util/feature_tool.py:12-39, model/loss.py:8-25); the total gradient w.r.t. the dense map written (not
 * accumulated) to grad.  coord_weight == 0 reads the map once (one pass + a merge kernel), otherwise twice.  acc: two doubles the call
 * ADDS onto; they must be zero on entry -- zeroed once by the caller (awr_zero_f64), afterwards by awr_loss_finalize_reset, which reads
 * the losses out and re-arms the accumulator in the same launch (a step that fails between the two calls must zero acc itself). */
int64_t awr_head_nhwc_scratch(int B, int J, int F);
int awr_head_forward_nhwc(const float* pred, int Cp, const float* img, int B, int J, int F, int H, float ks, float* scratch,
                          float* jt, float* stat /* optional */, void* stream);
int awr_head_loss_step_nhwc(const float* pred, int Cp, const float* img, const float* jt_gt, int B, int J, int F, int H, float ks,
                            float delta, float coord_weight, float dense_weight, float* scratch, float* jt, float* stat,
                            float* g_jt /* needed when coord_weight != 0 */, double* acc, float* grad, void* stream);
int awr_zero_f64(double* p, int64_t n, void* stream);
/* out[i] = (float)acc[i] for i<n, out[n] = sum -- e.g. {coord, dense, total} */
int awr_loss_finalize(const double* acc, int n, float* out, void* stream);
/* the same, and acc[0..n) is left zeroed: a step loop needs no awr_zero_f64 launch between steps */
int awr_loss_finalize_reset(double* acc, int n, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimiser: torch.optim.Adam defaults semantics (train.py:66-67, :131) over flat arenas.
 * g is multiplied by grad_scale first (1/world_size after the RCCL sum all-reduce).
 * -----------------------------------------------------------------------------------------*/
int awr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                  void* stream);
/* SGD with momentum (train.py:68-69): buf = mom*buf + g (buf = g on step 1); p -= lr*buf */
int awr_sgd_step(float* p, const float* g, float* buf, int64_t n, float lr, float momentum,
                 float weight_decay, int64_t step, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backbone building blocks (replace the torch.nn modules used by model/resnet_deconv.py:19-215
 * and model/hourglass.py:6-165 and their autograd).  Activations NHWC fp32.
 * -----------------------------------------------------------------------------------------*/

/* Weight repack between the checkpoint layout W[d0][d1][T] (Conv2d: O,I,kh*kw; ConvTranspose2d:
 * I,O,kh*kw) and the GEMM layout P[n][T][ld] (K-contiguous rows, zero padded: n < n_pad rows,
 * ld >= inner extent).  transpose==0: P[d0][t][d1] = W[d0][d1][t]; transpose!=0: P[d1][t][d0]. */
int awr_pack_weight(const float* w, int d0, int d1, int T, int transpose, int n_pad, int ld,
                    float* packed, void* stream);
/* grad[d0][d1][t] (+)= P[d0][t][d1] (P has row length ld): wgrad GEMM output -> checkpoint layout */
int awr_unpack_wgrad(const float* packed, int d0, int d1, int T, int ld, float* grad, int accumulate,
                     void* stream);

/* Batched forms of the two calls above: ONE launch repacks / scatters every layer of a network, one workgroup
 * per row (pack: packed row of T*ld floats; unpack: gradient row of d1*T floats).  The descriptor tables live
 * in DEVICE memory (built once per plan); `first` is the running ROW offset of each job (`rows` rows per pack
 * job, d0 rows per unpack job), `total_rows` their sum = the number of workgroups. */
typedef struct awr_pack_job {
    const float* src;
    float* dst;
    void* split;          /* optional: split image of dst (awr_split_weight layout), written in the same pass */
    int d0, d1, T, transpose, rows, ld;
    int64_t first;
    int cols;             /* 0 = ld.  < ld: only columns [0, cols) of each ld-float line are written (two jobs fill one buffer side by side) */
    int reserved;
} awr_pack_job;
typedef struct awr_unpack_job {
    const float* packed;
    float* grad;
    int d0, d1, T, ld;
    int64_t first;
    int slots, slot_stride; /* slots > 1: the value is the sum of `slots` copies of the packed buffer, slot_stride floats apart */
} awr_unpack_job;
/* Split image of a packed weight buffer of n fp32 elements (n % 32 == 0) for the 6- / 9-product modes of awr_conv_gemm:
 * every 32-element K-slice becomes 192 bytes [h: 32 bf16 | m: 32 bf16 | l: 32 bf16] with x == h + m + l exactly
 * (truncating 8+8+8-bit cut of the fp32 significand).  Weights are split once per optimiser step instead of once per
 * workgroup that stages them. */
int awr_split_weight(const float* packed, void* split, int64_t n, void* stream);
int awr_pack_weights_batched(const awr_pack_job* jobs_dev, int njobs, int64_t total_rows, void* stream);
int awr_unpack_wgrads_batched(const awr_unpack_job* jobs_dev, int njobs, int64_t total_rows, void* stream);

/* Geometry of one implicit-GEMM convolution-like gather:
 *   out[b, qy*so+py, qx*so+px, n] = sum_{tap in phase} sum_c in[b, qy*si+dy, qx*si+dx, c] * P[n][wt][c]
 * conv kxk stride s pad p      : so=1, si=s, one phase, taps (dy,dx,wt)=(ky-p,kx-p,ky*k+kx)
 * transposed conv k4 s2 p1     : so=2, si=1, four phases (py,px), taps with (py+p-ky) even,
 *                                dy=(py+p-ky)/2 (likewise x), wt=ky*k+kx
 * Tensors must stay below 4 GB (the kernels use 32-bit buffer offsets with hardware bounds checking).
 * Built by the host (awr-adaptive-weighting-regression_amd/ops.py: ConvSpec.fwd_problem / dgrad_problem / wgrad_problem). */
typedef struct awr_phase {
    int py, px, ntaps;
    int32_t tap[16];   /* (dy & 0xff) | (dx & 0xff) << 8 | wt << 16 : one scalar load per K-slice */
} awr_phase;

typedef struct awr_conv_args {
    const float* in;        /* (B,Hin,Win,Cin) */
    const float* w;         /* packed P[n_pad][T][Cin] */
    float* out;             /* (B,Hout,Wout,N) */
    const float* in_scale;  /* optional per-input-channel affine applied while loading ...     */
    const float* in_shift;  /* ... in-bounds pixels only (padding stays 0)                       */
    const float* bias;      /* optional per-output-channel bias                                  */
    const float* out_scale; /* optional per-output-channel affine after bias (folded eval BN)    */
    const float* out_shift;
    const float* res;       /* optional tensor added element-wise, same shape as out            */
    double* stats;          /* optional [stat_slots][2][N]: += sum and sum of squares of the stored
                               value (taken after bias/affine/res, before relu_out)              */
    const float* bnr_y;     /* optional fused BatchNorm-backward reduction (data-gradient launches): the result v (after `res`) */
    const float* bnr_coef;  /* is d(loss)/d(relu(bn(y)));  with coef = [scale|shift|mean|invstd][N] the kernel stores */
                            /* g = v * (y*scale+shift > 0) and accumulates sum g, sum g*(y-mean)*invstd into `stats` */
    int B, Hin, Win, Cin;
    int Hq, Wq;             /* per-phase output grid */
    int Hout, Wout, N;
    int so, si, T;
    int relu_in, relu_out;
    int nphase;
    int tile_m, tile_n;     /* workgroup tile in units of 64 rows / 64 columns ({1,2} each); 0,0 = built-in heuristic.
                               Static plans autotune this per launch (engine.Plan.autotune). */
    awr_phase ph[4];
    const void* w_split;    /* split image of w (awr_split_weight); required when awr_get_gemm_products() != 1 */
    int stat_slots;         /* slot copies in `stats` (0 = AWR_STAT_SLOTS); workgroup i adds into copy (stat_slot_base + i) % stat_slots.
                               ceil(M/64)*ceil(N/64)*nphase copies (the 64x64 tile's workgroup count) = one per workgroup */
    int stat_slot_base;
    const float* in2;       /* optional second input (B,Hin,Win,Cin-Cin1) of a single-tap launch: K = [Cin1 channels of `in` | the channels of
                               `in2`], P rows [n][Cin] hold the two weight rows side by side; in_scale / in_shift / relu_in apply to `in` only.
                               out = W_a.in + W_x.in2: the hourglass residual's conv3 + skip_layer in one launch (FP32-MFMA mode only) */
    int Cin1;
    float* partial;         /* optional scratch of split_max * B*Hout*Wout*N floats: enables split-K (small launches whose few workgroups
                               would each walk a long K loop -- low-batch inference): blockIdx.z takes a contiguous range of the K
                               slices and stores its raw partial tile, a second kernel sums the copies in order and applies the
                               epilogue (bias, folded BN, residual, ReLU).  FP32-MFMA mode, no `stats` / `bnr_y` / `in2` */
    int split_k;            /* K ranges (1 = off, 0 = heuristic from the workgroup count and K depth), <= split_max */
    int split_max;
    const float* bnr_act;   /* with bnr_y: take the ReLU mask from this stored activation (act > 0) instead of re-deriving it from y -- the
                               BatchNorm whose output had a residual added before the ReLU (resnet_deconv.py:74-78).  `res` may then be
                               set as well: the launch is the LAST producer of the gradient, adds its tile onto the earlier
                               contributions, masks, reduces and stores the masked gradient in place */
    const float* bnr2_y;    /* with bnr_y: a SECOND BatchNorm whose output was added to the first one's before the ReLU (the ResNet downsample */
    const float* bnr2_coef; /* projection: a = relu(bn2(y) + bn_ds(y2))) shares the masked gradient g: also accumulate sum g, */
    double* stats2;         /* sum g*(y2-mean2)*invstd2 into stats2 (same slot geometry as `stats`); coef2 = [scale|shift|mean|invstd][N] */
    const float* w2;        /* optional: TWO convolutions in one launch (inference; FP32-MFMA mode).  The conv described above has N1 = 128 */
    const float* bias2;     /* or 64 output channels: bias / out_scale / out_shift / relu_out apply to THAT intermediate, which never leaves */
    int N1;                 /* the chip; a 1x1 conv with the packed weights w2 [N][1][N1 + N1x] follows, its epilogue takes bias2 and `res`; */
    int N1x;                /* `out` and N = 2 N1 describe the second conv's output.  N1x > 0: the second conv's K extent continues with N1x */
                            /* channels of `in2` at the same pixel (no `res` then).  hourglass.py:44-59: conv2 -> bn3 -> ReLU -> conv3 + skip */
                            /* (identity skip = `res`; skip conv = [W3 | Wskip] over [intermediate | block input]) */
    const float* in_bnb_y;  /* optional (data gradients): `in` is g = d(loss)/d(bn(y)) (ReLU mask already applied) of a BatchNorm whose backward is NOT */
    const float* in_bnb_coef; /* materialised; the GEMM input is d(y) = a1 g + a2 (y - mean) + a3 per channel with y = in_bnb_y (same shape as `in`) and */
                            /* coef = [a1 | a2 | a3 | mean][Cin] (awr_bn_bwd_finalize_lin), formed while the operand is fed to the matrix pipe; padding */
                            /* taps stay zero.  Replaces the awr_bn_bwd_apply pass between two dependent data-gradient GEMMs.  LDS-DMA staging only */
    int accum;              /* 0 = one k-ordered accumulation chain over the whole K extent (v_mfma_f32_32x32x2_f32 after v_mfma ...); 1 = BLOCKED: every */
                            /* 128 k the running sum is folded into a second accumulator set and restarted (chains of 128 + K / 128 terms, the */
                            /* rounding behaviour of oneDNN's blocked kernels that the reference's CPU numbers come from).  LDS-DMA staging, K > 256 */
    const void* in_split;   /* optional (split-operand mode): the PRE-CUT image of `in` -- [pixel][Cin / 32][h | m | l][32] bf16, 6 bytes per element, written */
                            /* by awr_split_act (or a producer's epilogue): both operands then travel global -> LDS by DMA and the K loop holds no cutting */
                            /* arithmetic.  Excludes in_scale / relu_in (the producer applies them before it cuts), in2, in_bnb_y, split-K */
    float* pool_out;        /* optional, fused pair only (w2): also write MaxPool2d(2, 2) of the pair's output, (B, Hout / 2, Wout / 2, N) -- the workgroup */
                            /* tiles become 2D patches (two image rows x 32 / 64 columns) so that every window meets in the epilogue; needs an even map */
                            /* height and a width that is a multiple of the patch width.  model/hourglass.py:65-70: every level's input feeds up1 AND a pool */
    int out_nt;             /* cache policy of the output stores and of the epilogue's operand loads (res / bnr_y / bnr_act): 0 = automatic, 1 = cached, */
                            /* 2 = streaming (`buffer_store ... nt`: a short-K launch's 32 KB tile per workgroup does not evict the operand lines the K loops */
                            /* of its neighbours still want -- DESIGN.md 4.2) */
} awr_conv_args;

/* conv / transposed conv forward and data-gradient (all are the same gather-GEMM).
 * Replaces nn.Conv2d / nn.ConvTranspose2d forward and their dgrad. */
int awr_conv_gemm(const awr_conv_args* a, void* stream);
#ifdef AWR_STUDY
/* STUDY BUILDS ONLY (-DAWR_STUDY; the default library does not export it): one of `nparts` equal batch parts of that launch (B % nparts == 0) --
 * the half-batch BatchNorm-backward wavefront of round 5 (AWR_HALF_BNB_MIN_ROWS), measured slower on every shape */
int awr_conv_gemm_part(const awr_conv_args* a, int nparts, int part, void* stream);
#endif
/* test hook: force the (TM,TN) in {1,2}^2 workgroup tile of awr_conv_gemm / awr_conv_wgrad
 * (0,0 = automatic choice).  Not for production use. */
int awr_debug_force_tile(int tm, int tn);
/* How awr_conv_gemm / awr_conv_wgrad form their fp32 products (process-wide; default 1, or $AWR_GEMM_PRODUCTS):
 *   1 = v_mfma_f32_32x32x2_f32 on fp32 operands (bit-equal to an fmaf chain);
 *   6 = every operand is cut EXACTLY into three bf16 pieces (x == h + m + l, 8+8+8 significand bits) and each fp32 product
 *       is formed from six of its nine bf16 x bf16 partial products on v_mfma_f32_32x32x16_bf16 -- each partial product
 *       exact, accumulated in fp32; the three dropped ones (m*l, l*m, l*l) weigh <= 2^-23 of the product, i.e. less than
 *       the rounding of the fp32 accumulation that both modes share.  16x faster matrix pipe, 6 passes: see DESIGN.md.
 * Replaces nothing in the reference (torch's conv precision is whatever cuDNN / oneDNN pick; cuDNN defaults to TF32). */
int awr_set_gemm_products(int n);
int awr_get_gemm_products(void);
#ifdef AWR_STUDY
/* STUDY BUILDS ONLY (-DAWR_STUDY; the default library exports neither this nor the kernel that reads awr_conv_args.in_split: 15-40 % slower than
 * the in-register cut, profiles/r05_split_mode_studies.txt).  Split image of an activation tensor x (npix, C), C % 32 == 0: every element [relu](x * scale[c] + shift[c]) (scale / shift optional) cut EXACTLY
 * into three bf16 pieces, element idx -> shorts (idx / 32) * 96 + idx % 32 + {0, 32, 64} (the format of awr_split_weight).  What awr_conv_args.in_split
 * reads: a BatchNorm + ReLU output that the FP32 mode never materialises is written ONCE here instead of being cut per (tap, column tile) in the GEMM. */
int awr_split_act(const float* x, const float* scale, const float* shift, int relu, int64_t npix, int C, void* split, void* stream);
#endif
/* the product mode awr_conv_wgrad runs in: awr_get_gemm_products(), unless $AWR_WGRAD_SPLIT=0 sends the split mode's weight gradients to the FP32-MFMA
 * kernels (a measured-slower study arm: profiles/r05_split_mode_studies.txt) */
int awr_get_wgrad_products(void);
/* How the FP32-MFMA forward / data-gradient GEMM stages its operands (process-wide; default 2, or $AWR_DMA):
 *   2 = LDS-DMA: `buffer_load_dwordx4 ... lds` straight into swizzled, unpadded LDS rows, 16-float stages, double-buffered
 *       (weights always; activations whenever no fused input affine / ReLU has to touch them on the way in);
 *   0 = global -> registers -> ds_write -> LDS (rounds 1-3; kept as the same-box A/B reference and for split-K / fused pairs).
 * Results are bit-identical between the two (same k order).  Replaces nothing in the reference. */
/* Launch-path study knobs (cached at load from $AWR_DEEP / $AWR_DEEP_1X1 / $AWR_FAST_STATS): "deep" = deep pipeline for small launches (1),
 * "deep_1x1" = ... for every single-tap launch (0), "fast_stats" = BatchNorm statistics summed from the accumulators (1).  Results are
 * bit-identical either way except the statistics' summation order ("fast_stats"); tests and same-box A/Bs only. */
int awr_debug_set_knob(const char* name, int value);
int awr_set_gemm_staging(int mode);
int awr_get_gemm_staging(void);
/* Accumulation order of the forward / data-gradient GEMMs of plans created from now on (process-wide; default 2, or $AWR_ACCUM):
 * 0 = ordered, 1 = blocked (awr_conv_args.accum), 2 = AUTO (the default): a plain FORWARD launch of a TRAINING plan (FP32-MFMA, LDS-DMA
 * staging, no fused pair / split-K: what the blocked kernel exists for) is blocked when its K extent (taps x input channels of its longest
 * phase) reaches min_k terms (default 576: every 3x3 convolution from 64 channels up and the transposed convolutions), everything else --
 * data gradients, evaluation plans -- is ordered.  Why this split: the joints a network returns are a function of its forward GEMMs only;
 * training-mode BatchNorm divides by batch statistics, which is where a long ordered chain's rounding error (it grows with the chain) gets
 * amplified, while eval-mode plans measure no difference (1.851e-4 mm from the oracle either way); and blocking costs a second accumulator
 * set (occupancy 4 -> 2-3 waves per SIMD on the 128x128 tile), +4.5 % on the ResNet18 step when every launch is blocked, +1.3 % for the
 * training forward alone (profiles/r06_accum_modes.txt) -- for which the joints of the two-image training-mode fixtures land CLOSER to
 * float64 than the fp32 oracle's own.  awr_set_gemm_accum_auto(min_k, dgrad): the threshold, and whether data-gradient launches follow the
 * same rule (default no).  Blocked = every launch the blocked kernel exists for: the parity mode (InferEngine(parity=True),
 * TrainEngine(accum="blocked")).  awr_resolve_gemm_accum: what a launch of that K extent and kind gets under the current mode (what plan
 * builders store in awr_conv_args.accum, which itself only takes 0 / 1). */
#define AWR_GEMM_OTHER 0        /* evaluation-plan forward, fused pairs, anything else */
#define AWR_GEMM_FORWARD 1      /* forward launch of a training plan */
#define AWR_GEMM_DGRAD 2        /* data gradient */
int awr_set_gemm_accum(int mode);
int awr_get_gemm_accum(void);
int awr_set_gemm_accum_auto(int min_k, int dgrad);
int awr_get_gemm_accum_auto(int* min_k, int* dgrad);
int awr_resolve_gemm_accum(int k_extent, int kind);

/* weight gradient:  R[cd][t][cg] += sum_m D[m][cd] * G[pix(m,t)][cg]
 * D: dense operand (B,Hd,Wd,Cd); G: gathered operand (B,Hg,Wg,Cg) read at (y*sg+dy[t], x*sg+dx[t]).
 * conv wgrad: D=dY, G=X, (dy,dx)=(ky-p,kx-p), sg=stride -> R == packed [Cout][T][Cin].
 * deconv wgrad: D=X, G=dY, sg=2 -> R == [Cin][T][Cout].  R (row length ld>=Cg) must be zeroed by
 * the caller; split-K partial sums are combined with fp32 atomics. */
typedef struct awr_wgrad_args {
    const float* D;
    const float* G;
    float* R;
    const float* d_scale;   /* optional per-channel affine (+ReLU) applied to D / G while staging: the operand is */
    const float* d_shift;   /* then relu(t*scale+shift) of the stored tensor, i.e. a BatchNorm+ReLU output that  */
    const float* g_scale;   /* was never materialised (zero padding of the gather stays zero)                    */
    const float* g_shift;
    float* d_colsum;        /* optional [AWR_STAT_SLOTS][Cd]: slot copies whose SUM += column sums of D over all pixels (for conv
                               wgrad, D = dY, this IS the bias gradient -- it falls out of the slices the kernel stages anyway);
                               zeroed by the caller.  Slots: hundreds of workgroups add to the same Cd addresses */
    int d_relu, g_relu;
    int B, Hd, Wd, Cd, Hg, Wg, Cg, sg, T, ld;
    int tile_m, tile_n;     /* (cd, cg) tile in units of 64; 0,0 = heuristic */
    int target_blocks;      /* split-K: aim for this many workgroups; 0 = heuristic */
    int algo;               /* 0 = automatic; 1 = one workgroup per (tap, channel tile, pixel chunk); 2 = one WAVE per tap: a
                               workgroup owns a 64x64 channel tile for all taps and stages D / halo'd G patches once
                               (3x3 stride 1/2 and 4x4 stride-2 filters on power-of-two maps); 3 = one workgroup per KERNEL ROW: the three
                               taps of a row share the staged D pixels and one halo'd G row segment, three accumulators per wave, operands by
                               LDS-DMA (3x3 stride-1 filters, power-of-two maps at least 8 wide, D without a fused affine) */
    int8_t dy[16], dx[16];
    int64_t split_stride;   /* 0: split-K partial sums are combined with atomics in R.  > 0 (deterministic mode): K-chunk y stores its
                               tile at R + y*split_stride floats and its column sums at d_colsum + y*Cd; nothing needs zeroing */
    int max_split;          /* with split_stride: number of copies the caller allocated (caps the split-K depth) */
} awr_wgrad_args;
int awr_conv_wgrad(const awr_wgrad_args* a, void* stream);
/* 1 if awr_wgrad_args.algo = `algo` serves this geometry in the current product / staging mode (what a tuner may try), else 0 */
int awr_conv_wgrad_algo_ok(const awr_wgrad_args* a, int algo);
/* number of K-chunk copies the launch described by `a` writes (split_stride mode) */
int awr_conv_wgrad_splits(const awr_wgrad_args* a, int* nsplit);

/* Fused stems: conv 5x5 pad 2 of the one-channel depth image (1 -> 64 channels) -> BatchNorm -> ReLU [-> MaxPool(3,2,1)], their
 * autograd and the BatchNorm's running-stat update.  ResNet18-deconv (model/resnet_deconv.py:31-36, :118-121): no conv bias, pooled;
 * stacked hourglass (model/hourglass.py:112, Conv(1, 64, 5, 1, bn=True, relu=True)): conv bias, no pooling.  The un-normalised
 * full-resolution conv output is NEVER written: every kernel recomputes it from the image on the FP32 matrix pipe.  img (B,1,H,W),
 * H and W multiples of 16; w = the conv weight in checkpoint layout [64][25]; bias [64] or NULL; all maps NHWC.
 *   awr_stem_stats       stats[nslots][2][64] += per-channel sum / sum of squares of conv + bias
 *                        (feed awr_bn_finalize with count = B*H*W)
 *   awr_stem_pool        pooled (B,H/2,W/2,64) = maxpool(relu(conv * scale + shift)); argmax (optional, training) = window code
 *                        ky*3+kx of the first maximum, the format awr_maxpool_fwd writes
 *   awr_stem_conv        out (B,H,W,64) = [relu]((conv + bias) * scale + shift)
 *   awr_stem_bwd_reduce  sums[nslots][2][64] += sum g, sum g*xhat with g = relu' * dg; coef4 = [scale|shift|mean|invstd][64] of the
 *                        forward (feed awr_bn_bwd_finalize).  argmax != NULL: dg is the gradient of the POOLED map, routed through
 *                        the max-pool backward; argmax == NULL: dg is the dense gradient of awr_stem_conv's output
 *   awr_stem_bwd_wgrad   grad[64][25] = sum_pixels dY * image taps, gbias[64] (optional) = sum_pixels dY, dY = BatchNorm backward of
 *                        g with bwd_coef = [mean g | mean g*xhat | gamma*invstd][64]; dw_slots: nslots*64*26 floats of scratch,
 *                        zero before the first call (the call re-arms it) */
int awr_stem_stats(const float* img, const float* w, const float* bias, int B, int H, int W, double* stats, int nslots,
                   void* stream);
int awr_stem_pool(const float* img, const float* w, const float* scale, const float* shift, int B, int H, int W,
                  float* pooled, uint8_t* argmax, void* stream);
int awr_stem_conv(const float* img, const float* w, const float* bias, const float* scale, const float* shift, int relu,
                  int B, int H, int W, float* out, void* stream);
int awr_stem_bwd_reduce(const float* img, const float* w, const float* bias, const float* coef4, const float* dg,
                        const uint8_t* argmax, int B, int H, int W, double* sums, int nslots, void* stream);
int awr_stem_bwd_wgrad(const float* img, const float* w, const float* bias, const float* coef4, const float* bwd_coef,
                       const float* dg, const uint8_t* argmax, int B, int H, int W, float* dw_slots, float* grad,
                       float* gbias, int nslots, void* stream);
/* workgroup counts of awr_stem_stats / awr_stem_bwd_reduce (stats_slots) and of awr_stem_bwd_wgrad (wgrad_slots, also the
 * number of 64*26-float copies in dw_slots): that many slot copies = one per workgroup.  nslots = 0 means AWR_STAT_SLOTS. */
int awr_stem_slots(int B, int H, int W, int* stats_slots, int* wgrad_slots);

/* 5x5 stem (Cin=1): im2col of the depth image into (B,H,W,32) rows (25 taps + 7 zeros) so the
 * stem conv and its wgrad run on the same MFMA GEMMs (resnet_deconv.py:32, hourglass.py:112). */
int awr_stem_im2col(const float* img, int B, int H, int W, float* cols, void* stream);

/* BatchNorm2d (training): stats -> (scale, shift, mean, invstd) + running-stat update
 * (momentum, unbiased running var); zeroes `stats` afterwards.  nn.BatchNorm2d forward, train. */
int awr_bn_finalize(double* stats, int C, int64_t count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps,
                    float* scale, float* shift, float* mean, float* invstd, int nslots, void* stream);
/* eval-mode fold: scale = gamma/sqrt(var+eps), shift = beta - mean*scale */
int awr_bn_fold_eval(int C, const float* gamma, const float* beta, const float* running_mean,
                     const float* running_var, float eps, float* scale, float* shift, void* stream);
/* per-channel sum / sum of squares of an NHWC tensor (for BNs whose input is not a conv output);
 * stats is [AWR_STAT_SLOTS][2][C] like the conv epilogue's */
int awr_channel_stats(const float* x, int64_t npix, int C, double* stats, int nslots, void* stream);
/* awr_maxpool_fwd / awr_upsample2_add that ALSO accumulate the per-channel sum and sum of squares of the tensor they write (the slot layout of
 * awr_channel_stats): the BatchNorm that follows needs no statistics pass of its own over that tensor (model/hourglass.py:62-88: every pooled /
 * up-sampled-and-added map enters a pre-activation residual) */
int awr_maxpool_fwd_stats(const float* x, const float* in_scale, const float* in_shift, int in_relu, int B, int H, int W, int C, int k, int s, int p,
                          float* out, uint8_t* argmax, double* stats, int nslots, void* stream);
int awr_upsample2_add_stats(const float* up1, const float* low, int B, int Hl, int Wl, int C, float* out, double* stats, int nslots, void* stream);
/* out = [relu]( x*scale[c] + shift[c] [+ res] ) */
int awr_bn_apply(const float* x, const float* scale, const float* shift, const float* res, int relu,
                 float* out, int64_t npix, int C, void* stream);
/* backward pass 1: g = dout * relu_mask ; sums[slot][0][c] += sum g ; sums[slot][1][c] += sum g*xhat
 * (sums is [AWR_STAT_SLOTS][2][C] doubles, zero before the first use; pass 2 re-arms it).
 * relu_mask: (act > 0) if act is given; (y*mask_scale+mask_shift > 0) if mask_scale/shift are given (the
 * activation is re-derived from y with the forward's scale/shift: no read of the activation tensor); else 1. */
int awr_bn_bwd_reduce(const float* dout, const float* act, const float* y, const float* mean,
                      const float* invstd, const float* mask_scale, const float* mask_shift, int64_t npix,
                      int C, double* sums, int nslots, void* stream);
/* backward pass 2: dy = gamma*invstd*(g - sum_g/n - xhat*sum_gx/n) [+ dy_add]; optional g_out = g
 * (residual branch); dgamma/dbeta (+)= sums (accumulate); zeroes sums afterwards.  dy may alias
 * dy_add or dout.  coef: 3*C floats of scratch (per-channel coefficients collapsed from the slots). */
int awr_bn_bwd_apply(const float* dout, const float* act, const float* y, const float* mean,
                     const float* invstd, const float* gamma, const float* mask_scale,
                     const float* mask_shift, double* sums, float* coef, int64_t npix, int C,
                     float* dy, const float* dy_add, float* g_out, float* dgamma, float* dbeta,
                     int accumulate, int nslots, void* stream);
/* first half of awr_bn_bwd_apply on its own: collapse the slot-spread sums into coef = [mean g | mean g*xhat |
 * gamma*invstd][C], emit dgamma / dbeta, zero the sums (for fused consumers such as awr_stem_bwd_wgrad) */
int awr_bn_bwd_finalize(double* sums, int C, int64_t count, const float* gamma, const float* invstd, float* coef,
                        float* dgamma, float* dbeta, int accumulate, int nslots, void* stream);
/* awr_bn_bwd_finalize that ALSO writes lin4 = [a1 | a2 | a3 | mean][C] with d(y) = a1 g + a2 (y - mean) + a3: the form a consumer GEMM evaluates
 * on the fly (awr_conv_args.in_bnb_y / in_bnb_coef) when d(y) is not written to HBM on the critical chain.  Same reference lines as above. */
int awr_bn_bwd_finalize_lin(double* sums, int C, int64_t count, const float* gamma, const float* mean, const float* invstd, float* coef,
                            float* lin4, float* dgamma, float* dbeta, int accumulate, int nslots, void* stream);
/* second half of awr_bn_bwd_apply on its own: d(y) from coef (awr_bn_bwd_finalize / _lin), no reduction, no parameter gradients */
int awr_bn_bwd_apply_only(const float* dout, const float* act, const float* y, const float* mean, const float* invstd,
                          const float* mask_scale, const float* mask_shift, const float* coef, int64_t npix, int C, float* dy,
                          const float* dy_add, float* g_out, void* stream);
/* plain ReLU backward / mask: g = dout * (act > 0) */
int awr_relu_bwd(const float* dout, const float* act, float* g, int64_t n, void* stream);
/* out = a + b (n elements); out may alias a */
int awr_add(const float* a, const float* b, float* out, int64_t n, void* stream);
/* per-channel bias gradient: db[c] (+)= sum over pixels dy[pix][c] */
int awr_bias_grad(const float* dy, int64_t npix, int C, float* db, int accumulate, void* stream);

/* MaxPool2d(k,s,p) NHWC forward (+ uint8 argmax) and backward; nn.MaxPool2d(3,2,1)/(2,2).
 * in_scale/in_shift (optional): pool relu(x*scale+shift) instead of x (fused BatchNorm+ReLU input). */
int awr_maxpool_fwd(const float* x, const float* in_scale, const float* in_shift, int in_relu, int B, int H,
                    int W, int C, int k, int s, int p, float* out, uint8_t* argmax, void* stream);
int awr_maxpool_bwd(const float* dout, const uint8_t* argmax, int B, int H, int W, int C, int k,
                    int s, int p, float* dx, int accumulate, void* stream);
/* hourglass.py:88: out = up1 + nearest_upsample2(low)  and its backward (dlow = 2x2 sums of dout) */
int awr_upsample2_add(const float* up1, const float* low, int B, int Hl, int Wl, int C, float* out,
                      void* stream);
int awr_upsample2_bwd(const float* dout, int B, int Hl, int Wl, int C, float* dlow, int accumulate,
                      void* stream);

/* layout bridges at the reference boundary: NHWC (C padded to Cp) <-> NCHW (C channels) */
int awr_nhwc_to_nchw(const float* in, int B, int P, int Cp, int C, float* out, void* stream);
int awr_nchw_to_nhwc(const float* in, int B, int P, int Cp, int C, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Network-level API: the two backbones as natively built and replayed static plans.
 * Replaces get_deconv_net(18, J, downsample) / PoseNet('hourglass_<n>', J) construction, forward and
 * autograd backward (model/resnet_deconv.py:8-16,:19-136,:145-174; model/hourglass.py:6-165;
 * train.py:52-58, :116-130).  The Python host keeps only the nn.Module shell (state_dict views,
 * autograd hook, optimiser plumbing).
 *
 * Ownership: the caller owns the parameter / gradient / BatchNorm-buffer arenas (awr_net_bind) and
 * the boundary tensors handed to awr_plan_create (depth batch, NCHW dense maps and their gradients);
 * the library owns packed weights, activations, gradients and scratch (freed by *_destroy).  Calls
 * on one net / plan come from one host thread.
 * -----------------------------------------------------------------------------------------*/
typedef struct awr_net awr_net;
typedef struct awr_plan awr_plan;

/* kind 0: ResNet-deconv -- `nstack` carries the depth: 18 (also 0 / 1; BasicBlock) or 50 / 101 / 152 (Bottleneck), resnet_deconv.py:9-13;
 * kind 1: stacked hourglass (downsample ignored, feature size H/2).
 * Creates the checkpoint layout only: no device memory is touched until awr_net_bind. */
int awr_net_create(int kind, int nstack, int J, int downsample, awr_net** out);
int awr_net_destroy(awr_net* net);
/* state_dict entries (parameters, running stats, counters) in the reference's order; arena sizes in floats.
 * [0, n_active) of the parameter arena receives gradients; the tail holds hourglass skip_layers that never run */
int awr_net_sizes(const awr_net* net, int64_t* n_tensors, int64_t* n_params, int64_t* n_active,
                  int64_t* n_buffers, int* n_counters, int* nstage);
/* entry i: torch key, kind (0 conv weight OIHW, 1 transposed-conv weight IOHW, 2 conv bias, 3 BN weight, 4 BN bias,
 * 5 running_mean, 6 running_var, 7 num_batches_tracked), shape, offset in floats into the parameter arena (kinds 0-4)
 * or the buffer arena (5, 6), counter index (7); unused = never receives a gradient */
int awr_net_tensor_info(const awr_net* net, int64_t i, const char** key, int* kind, int* ndim,
                        int64_t shape[4], int64_t* offset, int* unused);
/* attach 16-byte aligned device arenas (n_params, n_params, n_buffers floats); destroys the net's plans */
int awr_net_bind(awr_net* net, float* params, float* grads, float* buffers);

/* One static plan for (B, H, mode).  img (B,1,H,H); outs[stage] (B,4J,F,F) NCHW dense maps; grad_outs[stage] their
 * gradients (training plans; zero the stages you do not supervise).  supervised_mask: bit s = stage s receives a
 * gradient (hourglass training supervises the last stage only, train.py:116-121); bn_repeat: BatchNorm momentum is
 * applied that many times per forward (the reference runs `stacks` forwards per iteration); n_buckets > 1: the backward
 * hands out gradient-arena ranges through the bucket callback as soon as they are final. */
int awr_plan_create(awr_net* net, int B, int H, int training, unsigned supervised_mask, int bn_repeat,
                    int n_buckets, float* img, float* const* outs, float* const* grad_outs, awr_plan** out);
int awr_plan_destroy(awr_plan* plan);
int awr_plan_info(const awr_plan* plan, int64_t* bytes, int* deterministic, int* n_fwd, int* n_bwd,
                  int* n_buckets, int* n_gemm, int* n_bn);
/* forward and data-gradient launches of the plan that run as Winograd F(2x2, 3x3) (awr_set_conv_winograd): how many, and their ALGORITHMIC multiply-adds per replay
 * (the matrix pipe executes 16 / 36 of them) */
int awr_plan_winograd(const awr_plan* plan, int* n, double* macs);
int awr_plan_bucket(const awr_plan* plan, int i, int64_t* lo, int64_t* hi, int* ready_op);
/* op i of the forward (list 0) / backward (list 1) launch list: name, algorithmic MACs (GEMM-family launches),
 * flags bit 0 = weight gradient that may run on a side stream, bit 1 = conv / stem family (the launches a roofline is quoted for) */
int awr_plan_op(const awr_plan* plan, int list, int i, const char** name, double* macs, int* flags);
/* parity / debugging introspection: activation tensor i of the plan in build order (i past the end returns AWR_ERR_ARG): name
 * ("<layer>.out" = a conv's raw output, "<bn>.act" = a materialised BatchNorm(+ReLU) output, "<bn>.act(lazy)" = one that is never
 * written, ...), NHWC dims {B,H,W,C}, the value buffer and, after a backward replay, its gradient buffer (NULL if none; identity skips
 * alias the buffer of the tensor they come from).  lazy != 0: the handle stands for buf * scale[c] + shift[c] (2: with a ReLU on top) of
 * ANOTHER tensor's buffer, applied by the consumers' loaders; lz_scale / lz_shift are those per-channel coefficients. */
int awr_plan_tensor(const awr_plan* plan, int i, const char** name, int dims[4], float** buf, float** grad, int* lazy,
                    const float** lz_scale, const float** lz_shift);
/* NHWC boundary (round 3): stage s's dense map as the head GEMM leaves it, (B, F*F, Cp) rows, and -- training plans, supervised stages
 * -- the buffer the backward reads its gradient from (NULL when that stage's gradient has a second producer and must go through
 * grad_outs).  awr_plan_set_nhwc_boundary(plan, 1): the plan stops transposing to / from the NCHW boundary tensors (outs / grad_outs are
 * then neither written nor read): the caller runs awr_head_forward_nhwc / awr_head_loss_step_nhwc on these buffers between
 * awr_plan_forward and awr_plan_backward.  Refused (AWR_ERR_UNSUPPORTED) when a supervised stage has no NHWC gradient buffer. */
int awr_plan_head_nhwc(const awr_plan* plan, int stage, const float** pred, float** grad, int* Cp);
int awr_plan_set_nhwc_boundary(awr_plan* plan, int on);
/* n_side extra HIP streams (weight gradients in the backward, forked branches in the forward); comm != 0 adds the
 * stream buckets are handed to */
int awr_plan_set_streams(awr_plan* plan, int n_side, int comm);      /* n_side in 0..4 */
typedef void (*awr_bucket_cb)(void* user, int64_t lo, int64_t hi, void* stream);
/* cb(user, lo, hi, stream): grads[lo, hi) is final in `stream` order -- start its all-reduce there */
int awr_plan_set_bucket_callback(awr_plan* plan, awr_bucket_cb cb, void* user);
/* The library's side / branch streams come from ONE process-wide pool, ordered by a probe (a 100 us spin kernel on two streams at once):
 * the first n_independent streams share a hardware queue neither with the null stream nor with each other -- HIP multiplexes streams onto
 * a few hardware queues and streams on one queue serialise, so which streams a plan gets decides whether its side streams overlap anything.
 * Built on first use (a few milliseconds, synchronises the device once). */
int awr_stream_pool_info(int* n_streams, int* n_independent);
/* repack every conv weight (and re-fold eval BatchNorms) from the arena; forward; backward */
int awr_plan_refresh_weights(awr_plan* plan, void* stream);
int awr_plan_forward(awr_plan* plan, void* stream);
int awr_plan_backward(awr_plan* plan, void* stream);
/* serial replay with a HIP-event pair around every launch: ms[i] per op, 0 for fills / copies / markers (synchronises the stream) */
int awr_plan_run_timed(awr_plan* plan, int list, void* stream, float* ms);
/* time the tile / split-K candidates of every GEMM launch in place (no-op in deterministic mode); read / preset choices
 * (target_blocks: weight gradients = workgroup target of the split over pixels; conv launches with `partial` scratch = split-K depth) */
int awr_plan_autotune(awr_plan* plan, int reps, void* stream);
int awr_plan_gemm(const awr_plan* plan, int i, const char** name, int* tile_m, int* tile_n, int* target_blocks,
                  float* us, int* tuned);
int awr_plan_set_gemm(awr_plan* plan, int i, int tile_m, int tile_n, int target_blocks, float us);
/* ... and the algorithm of a weight-gradient launch (awr_wgrad_args.algo: 0 automatic, 1 workgroup per tap, 2 wave per tap, 3 workgroup
 * per kernel row; always 0 for conv launches): what a tuning cache stores beside the tile.  set: AWR_ERR_ARG if the launch cannot run it. */
int awr_plan_gemm_algo(const awr_plan* plan, int i, int* algo);
int awr_plan_set_gemm_algo(awr_plan* plan, int i, int algo);

/* ------------------------------------------------------------------------------------------
 * Data-parallel API (net-new w.r.t. the reference, which hard-wires one GPU: train.py:29,:233).
 * One process per GPU.  The library opens librccl.so at run time (dlopen; AWR_RCCL_LIB overrides the path) -- libawr_hip.so has
 * no link-time dependency on RCCL and a host that exchanges gradients itself (awr_plan_set_bucket_callback) never loads it.
 *   rank 0:      awr_dp_unique_id(id)            -> ship the 128 bytes to the other ranks out of band (file, socket, MPI, ...)
 *   every rank:  awr_dp_init(rank, world, id, &dp)   (collective; the communicator lives on the CURRENT device)
 *                awr_dp_broadcast(dp, params, n, 0, stream) / (..., bn_buffers, ...): identical replicas
 *                awr_plan_set_dp(plan, dp): awr_plan_backward all-reduces (SUM) every gradient bucket as soon as it is final -- on the
 *                communicator's own stream, behind the bucket's scatter, overlapping the rest of the backward -- and returns with the
 *                caller's stream ordered after all of them; pass grad_scale = 1/world to awr_adam_step / awr_sgd_step.
 * BatchNorm statistics stay rank-local (what stock DDP does).  awr_dp_allreduce / awr_dp_broadcast order themselves behind `stream`
 * and run on the communicator's stream; awr_dp_wait(dp, stream) orders `stream` behind everything issued so far.
 * -----------------------------------------------------------------------------------------*/
#define AWR_DP_ID_BYTES 128
typedef struct awr_dp awr_dp;
int awr_dp_available(int* version, const char** path);          /* AWR_OK if librccl.so and its symbols were found */
int awr_dp_unique_id(void* id128);
int awr_dp_init(int rank, int world, const void* id128, awr_dp** out);
int awr_dp_destroy(awr_dp* dp);
int awr_dp_info(const awr_dp* dp, int* rank, int* world, int* device);
int awr_dp_allreduce(awr_dp* dp, float* buf, int64_t n, void* stream);               /* in place, SUM */
int awr_dp_broadcast(awr_dp* dp, float* buf, int64_t n, int root, void* stream);     /* in place */
int awr_dp_wait(awr_dp* dp, void* stream);
/* dp != NULL: the plan's gradient buckets (n_buckets of awr_plan_create; 1 = one exchange after the backward) go through `dp`;
 * NULL detaches.  Replaces the bucket callback while set.  Lifetime: the plan keeps the pointer, not a reference -- detach (NULL) before
 * awr_dp_destroy; a backward that finds its communicator destroyed fails with AWR_ERR_ARG instead of calling into it. */
int awr_plan_set_dp(awr_plan* plan, awr_dp* dp);

/* ------------------------------------------------------------------------------------------
 * Winograd F(2x2, 3x3) on the FP32 matrix pipe (round 6, csrc/awr_wino.hip): stride-1 pad-1 3x3 convolutions with 2.25x fewer multiplies than the
 * direct implicit GEMM -- nn.Conv2d(C, N, 3, 1, 1) of model/resnet_deconv.py:139-142,:161-165 / model/hourglass.py:35.
 *   U[pos][c][n] = (G g G^T)[pos]            awr_wino_weights: from the checkpoint tensor (OIHW), once per optimiser step; mirror != 0 = the
 *                                            data-gradient form (N = the layer's Cin, C = its Cout)
 *   V = B^T d B per 4x4 input window, M[pos] = sum_c V U (16 independent GEMMs on v_mfma_f32_32x32x2_f32), Y = A^T M A per 2x2 output patch
 * awr_wino2_conv3x3: out (B, H, W, N) = conv([relu](in * in_scale + in_shift)) [+ bias] [ReLU], NHWC; the raw input tile of a workgroup (a 2-D
 * block of 64 patches of one image, or several small images) is staged in LDS once per K stage, padding stays zero under the fused input affine;
 * `stats` (optional) += per-channel sum / sum of squares of the stored output ([nslots][2][N] doubles, awr_bn_finalize's layout).  Power-of-two map
 * sizes >= 4, C % 8 == 0, N % 32 == 0, fewer than 2^31 input elements.  Results are NOT bit-compatible with awr_conv_gemm: Winograd's
 * rounding differs -- measured 0.3-0.6x the direct kernel's error against float64 (chains of Cin terms instead of 9 Cin).
 * awr_wino_conv3x3 is the first form (windows gathered from global memory; kb = channels per stage, + 100 = 256-thread workgroups), kept for the
 * measurements in profiles/r06_winograd.txt.
 * Plans: awr_set_conv_winograd(1 | 2) (process-wide, captured when a plan is built, default $AWR_WINOGRAD or 0; 2 = also the data and weight
 * gradients, + 4 = ignore the launch-size rules: tests, + 8 = never the 64-channel tile form: A/B) makes plan builders run the FORWARD
 * of every eligible layer (awr_wino_eligible: maps >= 8 x 8, at least 256 workgroups; epilogue = bias / ReLU / statistics) through
 * awr_wino2_conv3x3, the DATA GRADIENT of those layers (mirrored transform, accumulate / BatchNorm-backward-reduction epilogues) through
 * awr_wino_dgrad_or_direct, and the WEIGHT GRADIENT of the layers awr_wino_wgrad_eligible names through awr_wino_wgrad (below).  Inference plans take the
 * forward form with the folded eval-mode BatchNorm and a residual add in the epilogue (awr_wino_args.out_scale / out_shift / res); a Hourglass residual then runs
 * conv2 as Winograd followed by the direct conv3 (+ skip) GEMM instead of the fused two-GEMM launch (awr_conv_args.w2).
 * -----------------------------------------------------------------------------------------*/
int awr_wino_weights(const float* w, int N, int C, int Npad, int Cpad, int mirror, float* U, void* stream);
int awr_wino_conv3x3(const float* in, const float* U, const float* bias, float* out, int B, int H, int W, int C, int N, int relu, int kb, void* stream);
int awr_wino2_conv3x3(const float* in, const float* U, const float* bias, const float* in_scale, const float* in_shift, int relu_in, float* out,
                      double* stats, int nslots, int B, int H, int W, int C, int N, int relu, void* stream);
/* the same kernel through an argument block, with the data-gradient epilogue forms of awr_conv_args (res = accumulate in place, bnr_y / bnr_coef /
 * bnr_act = mask with the re-derived ReLU and reduce sum g, sum g * xhat into `stats` for the BatchNorm backward) */
typedef struct awr_wino_args {
    const float *in, *U, *bias, *in_scale, *in_shift;
    float* out;
    double* stats;
    const float *res, *bnr_y, *bnr_coef, *bnr_act;
    int B, H, W, C, N, relu, relu_in, nslots;
    const float *out_scale, *out_shift;      /* optional per-output-channel affine after the bias (a folded eval-mode BatchNorm: inference plans):
                                                out = [relu]((acc + bias) * out_scale + out_shift [+ res]); `res` may then be any (B,H,W,N) tensor */
} awr_wino_args;
int awr_wino_conv(const awr_wino_args* a, void* stream);
/* plans: the data gradient of a stride-1 3x3 convolution described by the DIRECT kernel's argument block `d` -- as Winograd (U = awr_wino_weights(...,
 * mirror = 1)) when this kernel implements everything `d` asks for, through awr_conv_gemm(d) otherwise (decided at launch: plan builders complete
 * `d` after they have created the launch) */
int awr_wino_dgrad_or_direct(const awr_conv_args* d, const float* U, void* stream);
int awr_wino_dgrad_supported(const awr_conv_args* d);
int awr_set_conv_winograd(int on);
int awr_get_conv_winograd(void);
int awr_wino_eligible(int B, int H, int W, int C, int N);
/* Weight gradient of a stride-1 3x3 convolution in the Winograd domain: dg = G^T [sum_patches (B^T d B) (.) (A dY A^T)] G -- 16 GEMMs over the patches,
 * 2.25x fewer multiplies than the nine taps of awr_conv_wgrad.  x (B,H,W,C) is the convolution's input (optionally behind a fused per-channel affine
 * + ReLU: a BatchNorm output that was never written), dy (B,H,W,N) the gradient of its output.  ASSIGNS R[N][9][ld] (the packed layout awr_conv_wgrad
 * accumulates into for D = dY, G = X; ld >= C) and, if not NULL, bias_grad[N] = sum of dy over the pixels.  `scratch`: awr_wino_wgrad_scratch(...)
 * floats (one copy of the transformed-domain tile per K split; summed in a fixed order: deterministic).  C, N multiples of 64, power-of-two maps >= 8 x 8.
 * awr_wino_wgrad_eligible: 1 where a plan should use it (every split keeps a long enough K loop to pay for the copy it stores). */
int awr_wino_wgrad(const float* x, const float* dy, const float* x_scale, const float* x_shift, int x_relu, int B, int H, int W, int C, int N,
                   float* scratch, float* R, int ld, float* bias_grad, void* stream);
int64_t awr_wino_wgrad_scratch(int B, int H, int W, int C, int N);
int awr_wino_wgrad_eligible(int B, int H, int W, int C, int N);

/* ------------------------------------------------------------------------------------------
 * NYU data path on the device (csrc/awr_nyu.hip).  Replaces the IMAGE work of the reference's per-sample loader --
 * Loader.crop (dataloader/loader.py:19-51: center2bounds window, bounds2crop zero padding + cube clamp :190-208, cv2.resize
 * INTER_NEAREST, paste), Loader.augment (:75-86) with recrop = cv2.warpPerspective + fringe clean-up + cube clamp (:123-137) or
 * cv2.warpAffine (:140-160), and Loader.normalize (:88-101) -- for a whole batch per launch.  The host keeps what is tiny and
 * sequential: the RandomState draws, the 3x3 matrices in float64 / float32 exactly as numpy computes them, the label arithmetic
 * (awr_amd.nyu_device builds one awr_nyu_sample per image).  Results are bit-identical to awr_amd.nyu_data (the numpy restatement,
 * itself bit-exact against the reference everywhere except cv2's three resamplers, which no cv2-produced vector pins: README).
 *   frames: the decoded depth frames of the dataset, resident in HBM ([n_frames][fh][fw], uint16 millimetres = G*256 + B of the
 *           PNG, dataloader/nyu_loader.py:71-74; or float32); a sample addresses its frame by index.
 *   All per-pixel coordinate arithmetic is IEEE double / float without contraction (the file is built with -ffp-contract=off).
 * -----------------------------------------------------------------------------------------*/
#define AWR_NYU_NONE 0
#define AWR_NYU_PERSPECTIVE 1       /* translate / scale: recrop through a homography (loader.py:103-137, :163-179) */
#define AWR_NYU_AFFINE 2            /* rotate (loader.py:140-160) */
#define AWR_NYU_U16 0
#define AWR_NYU_F32 1
typedef struct awr_nyu_sample {
    int64_t frame;              /* row of the frame store */
    int32_t ustart, vstart;     /* crop window origin in the frame; may be negative (zero padding, loader.py:196-200) */
    int32_t cw, ch;             /* window extent uend - ustart, vend - vstart */
    int32_t rw, rh;             /* extent after cv2.resize(..., INTER_NEAREST) (loader.py:37-40) */
    int32_t ox, oy;             /* where the resized window is pasted into the dsize x dsize crop (loader.py:43-47) */
    double ifx, ify;            /* resizeNN's inverse scale factors 1. / (rw / cw), 1. / (rh / ch) (two divisions, in this order) */
    double zstart, zend;        /* depth range of the cube: below -> zstart, above -> 0, zero stays zero (loader.py:202-206) */
    int32_t op;                 /* AWR_NYU_NONE / _PERSPECTIVE / _AFFINE */
    int32_t norm32;             /* normalisation arithmetic: 0 = float64 (numpy's promotion for a float64 centre or cube), 1 = float32 */
    double m[9];                /* destination -> source map: _PERSPECTIVE inv(M_new . inv(M)) row-major; _AFFINE the inverted 2x3 in m[0..5] */
    double zstart2, zend2;      /* cube clamp of the recrop (loader.py:131-135) */
    double lo, far, center_z, half;   /* normalize: x in {depth_max, 0} -> far; clip(lo, far); (x - center_z) / half (loader.py:88-101) */
} awr_nyu_sample;

/* Loader.crop (loader.py:19-51) for B samples: crop (B, dsize, dsize) float32; stats (B, 2) float32 = {max of the crop (augment's
 * depth_max, loader.py:76), smallest positive value (recrop's nv_val + 1, loader.py:116; +inf if none)}.  One workgroup per sample. */
int awr_nyu_crop(const void* frames, int frame_type, int fh, int fw, const awr_nyu_sample* samples, int B, int dsize, float* crop,
                 float* stats, void* stream);
/* The resamplers alone, cv2.warpPerspective / cv2.warpAffine with INTER_LINEAR + BORDER_CONSTANT (OpenCV's fixed-point rule: source
 * coordinates in 1/32 pixel, 10-bit affine increments): dst (B, dh, dw) from src (B, sh, sw); m (B, 9) doubles as awr_nyu_sample.m. */
int awr_nyu_warp(const float* src, int sh, int sw, const double* m, int op, float border, int B, int dh, int dw, float* dst, void* stream);
/* Loader.normalize (loader.py:88-101) alone: out[b] = normalize(depth_max[b], img[b], centre, cube) over n pixels per sample */
int awr_nyu_normalize(const float* img, const float* depth_max, const awr_nyu_sample* samples, int B, int64_t n, float* out, void* stream);
/* Loader.augment's image half (loader.py:75-86) from materialised crops: warp per `op`, fringe clean-up against stats, cube clamp,
 * normalize -> out (B, 1, dsize, dsize).  status[b] (optional) = 1 where a recrop found no positive pixel (the reference raises). */
int awr_nyu_augment(const float* crop, const float* stats, const awr_nyu_sample* samples, int B, int dsize, float* out, int* status,
                    void* stream);
/* The production entry: crop + augment + normalize in ONE launch, one workgroup per sample, the crop staged in LDS (never written to
 * HBM) when dsize * dsize floats fit (dsize <= 160); larger crops go through `scratch` (B * (dsize * dsize + 2) floats, else may be
 * NULL) with the two kernels above.  Test-time samples (nyu_loader.py:59-60) are op = AWR_NYU_NONE. */
int awr_nyu_batch(const void* frames, int frame_type, int fh, int fw, const awr_nyu_sample* samples, int B, int dsize, float* out,
                  int* status, float* scratch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AWR_HIP_H */
