/*
 * diga_hip.h -- C-ABI of libdiga_hip.so: the MI355X (gfx950) kernels behind DIGA's backward-free
 * test-time-adaptation forward path.
 *
 * The reference (Waybaba/DIGA) is 100 % Python: it has no FFI/operator layer, its "kernels" are
 * ATen ops reached from eager PyTorch.  Each entry point below therefore cites the reference
 * *Python call site* it replaces ("ttda" = src/models/ttda_module.py, "adaptseg" =
 * src/models/components/adaptseg.py).  The host-side mirror of the reference's class API
 * (diga_amd/ttda_module.py: DIGA, DeepLabv2AdaptSeg, SIFABatchNorm2dTrainable, SegmentationMetric)
 * binds these with ctypes -- see INTEGRATION.md.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer on the current HIP device unless its name ends in _host;
 *   - `stream` is a hipStream_t passed as void*; every call only ENQUEUES work on it;
 *   - no entry point allocates, frees or synchronises (safe under stream capture) -- except the SET-UP calls of the one-shot
 *     exchange (diga_xgmi_create / _connect / _destroy allocate and map its buffer) and diga_xgmi_check (synchronises);
 *   - return value: 0 on success, negative DIGA_E* on error (never throws);
 *     diga_last_error_string() describes the last error of the calling thread;
 *   - activations are NHWC ("pixels x channels") bf16 (raw uint16_t bit patterns), statistics,
 *     prototypes, logits and probabilities are fp32, labels are int32 internally.
 */
#ifndef DIGA_HIP_H
#define DIGA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIGA_OK 0
#define DIGA_EINVAL (-1)   /* bad argument / unsupported shape */
#define DIGA_ELAUNCH (-2)  /* HIP launch error */

#define DIGA_MAX_TAPS 18
#define DIGA_NUM_CLASSES_MAX 32

int diga_version(void);
const char* diga_last_error_string(void);

/* ------------------------------------------------------------------------------------------
 * Convolutions (adaptseg:63-74 Bottleneck convs, adaptseg:111,119-121 head; nn.Conv2d call sites
 * adaptseg:85,89,93,97,178).  Implicit GEMM on bf16 MFMA with fp32 accumulation:
 *   out[m, co] = sum_{tap, ci} in[n, oh*stride+dy[tap], ow*stride+dx[tap], ci] * wgt[co, tap, ci]
 * with m = (n, oh, ow) flattened; out-of-image taps read as zero (== zero padding).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t B, H, W, Cin;        /* input  NHWC bf16; Cin % 64 == 0 */
  int32_t Ho, Wo, Cout;        /* output NHWC; Cout % 8 == 0 for bf16 output */
  int32_t stride;              /* 1 or 2 (stride sits on the first 1x1 of a block, adaptseg:63) */
  int32_t ntaps;               /* 1 (1x1), 9 (3x3 dilated) or 18 (fused 2-branch head) */
  int16_t dy[DIGA_MAX_TAPS];   /* tap offsets in input pixels, e.g. (kh-1)*dilation */
  int16_t dx[DIGA_MAX_TAPS];
  int32_t cout_pad;            /* rows of wgt (>= Cout, multiple of the N tile); padding rows are zero */
  int32_t splitk;              /* >= 1; > 1 only with out_f32 */
  int32_t out_f32;             /* 0: out = bf16 [M][Cout];  1: out = fp32 [splitk][M][cout_pad] partials */
  int32_t tile;                /* 0 = auto; else tile-config id (csrc/conv_igemm.hip kTiles; 91 = the weights-stationary 1x1
                                  kernel: stride 1, Cin 64 / 128 / 256, Cout % 128 == 0, bf16 output + statistics; 92 = the same kernel for
                                  Cin 512 with 64-column slices; 94 / 95 / 96 = measured variants of 91 (two passes with 12 waves; the epilogue pipelined inside the
                                  wave, 8 / 4 waves), same shapes, kept for A/B runs; 93 = the A-stationary form of diga_conv_igemm_bf16_fused_in -- all refused with
                                  an error for anything else).  Outputs and statistics do not depend on the id. */
  int32_t nplanes;             /* 0 / 1: bf16 operands (the product path).  2 / 3: verification mode -- `in` is nplanes bf16
                                * planes [nplanes][B*H*W][Cin] whose SUM is the fp32 activation (hi = bf16(x), lo = bf16(x - hi),
                                * ...), `wgt` likewise [nplanes][cout_pad][K]; the plane products (a, b) with a + b < nplanes are
                                * accumulated into the same fp32 accumulators (3 terms: "bf16x3", ~2^-17 relative; 6 terms:
                                * "bf16x6", fp32-grade).  Requires out_f32 and splitk == 1. */
} diga_conv_desc;

/* Number of statistics units diga_conv_igemm_bf16 writes for `desc`: B * ceil(Ho*Wo / 32).  A unit is
 * 32 consecutive output rows of ONE sample (M tiles never straddle samples), independent of the tile shape. */
int diga_conv_stat_units(const diga_conv_desc* desc);

/* in: bf16 NHWC; wgt: bf16 [cout_pad][Cin/64][ntaps][64] (K order = channel chunk, tap, channel); out: see desc.out_f32
 * (with out_f32 and splitk == 1 a stat_slab may be given too: fp32 output + statistics, the verification mode's form);
 * stat_slab: NULL, or fp32 [stat_units][2 (sum, sumsq)][Cout] of *shifted* values (x - stat_shift[c])
 * taken from the fp32 accumulators, sample-major; stat_shift: fp32 [Cout] (the source running_mean:
 * the shifted-data variance algorithm stays accurate when mean >> std).  Which samples form the
 * reference's half_first groups is decided in diga_bn_finalize, not here. */
int diga_conv_igemm_bf16(const diga_conv_desc* desc, const void* in, const void* wgt, void* out,
                         float* stat_slab, const float* stat_shift, void* stream);
/* The same convolution with the DAM finalize folded into the launch (SIFABatchNorm2d.forward's statistics and lambda-mix,
 * ttda:313-343, for the case where EVERY SAMPLE is normalised with its own statistics: batch 1, or the frames of a batch-1
 * stream in flight): the last tile to finish of each (sample, N tile) group reduces that sample's slab columns in the fixed
 * order of diga_bn_finalize(per_sample = 1) and writes scale / shift [B][Cout] -- bit-identical to
 * diga_conv_igemm_bf16 + diga_bn_finalize, one launch less per BN layer.
 * ticket: uint32 [diga_conv_dam_ticket_words(desc)] device words, ZERO before the first launch; every launch leaves them
 * zero.  Launches that share a ticket buffer must not overlap (one buffer per stream). */
int diga_conv_dam_ticket_words(const diga_conv_desc* desc);
int diga_conv_igemm_bf16_dam(const diga_conv_desc* desc, const void* in, const void* wgt, void* out, float* stat_slab,
                             const float* running_mean, const float* running_var, const float* weight,
                             const float* bias, float lam, float eps, float* scale, float* shift,
                             uint32_t* ticket, void* stream);
/* The same convolution (bf16 output + statistics) whose INPUT is the producer's RAW output: the DAM apply + ReLU of the previous
 * BN layer runs on the operand path, A = bf16(relu(raw_in*in_scale[c] + in_shift[c] [+ in_res])), with the arithmetic and the
 * single bf16 rounding of diga_bn_apply -- results are bit-identical to diga_bn_apply followed by diga_conv_igemm_bf16, but
 * the normalised activation is never written and read back ("conv -> bn -> relu -> conv", adaptseg:85-93).
 * in_res / act_out (both or neither; 1x1 stride-1 convs): the residual form relu(bn3(raw3) + x) that opens the NEXT bottleneck
 * (adaptseg:94-100 followed by adaptseg:85): in_res = the previous block's input x, and A -- the new block input, which the
 * next residual needs -- is also written to act_out (bf16 [B*H*W][Cin]).
 * in_scale / in_shift: fp32 [Cin], or [B][Cin] with nsamples_affine = B (per-sample statistics, see diga_bn_finalize). */
int diga_conv_igemm_bf16_fused_in(const diga_conv_desc* desc, const void* raw_in, const float* in_scale,
                                  const float* in_shift, const void* in_res, void* act_out, int nsamples_affine,
                                  const void* wgt, void* out, float* stat_slab, const float* stat_shift,
                                  void* stream);
/* out == NULL with a stat_slab: statistics only, nothing is stored.  Together with the entry point below this
 * replaces "conv3 -> bn3 -> += residual -> relu" (adaptseg:93-100) by two passes over the same GEMM: pass 1 takes
 * the DAM statistics, diga_bn_finalize makes the affine, pass 2 recomputes the conv and writes
 *   out = relu?( acc*scale[c] + shift[c] + res )  (bf16)
 * straight from the fp32 accumulators, so the widest tensor of a bottleneck is never written raw and read back
 * (the 1x1 conv3 is cheap to recompute: 4.4 of a layer3 bottleneck's 18.7 GFLOP, against 6 of its 8 bytes per
 * element of memory traffic).  res: bf16 [M][Cout] already-normalised residual; nsamples_affine: 0 = one affine [C],
 * B = per-sample affines [B][C] (see diga_bn_finalize). */
int diga_conv_igemm_bf16_apply(const diga_conv_desc* desc, const void* in, const void* wgt, void* out,
                               const float* scale, const float* shift, const void* res, int relu,
                               int nsamples_affine, void* stream);

/* Stem (adaptseg:133,178): 7x7 stride 2 pad 3, 3 -> 64, fp32 NCHW input [B,3,H,W] -> raw bf16 NHWC
 * [B,Ho,Wo,64] + statistics slab [units][2][64] (sample-major, units/B per sample).
 * wgt: fp32 [64][3][7][7] (the reference layout).  Returns the number of slab units when x == NULL.
 * Arithmetic: inputs and weights are split into bf16 hi + lo and multiplied on the matrix pipes in three terms
 * (hi.lo + lo.hi + hi.hi, fp32 accumulation: 2^-17 relative per product, far inside the bf16 rounding of the output;
 * diga_stem_conv7x7s2_f32 below is the fp32 vector form in the reference's (c, kh, kw) order).  x must be finite and
 * below bf16's range (an image is): the K padding of the matrix form multiplies neighbouring pixels by zero weights. */
int diga_stem_conv7x7s2(const float* x, const float* wgt, void* out, float* stat_slab,
                        const float* stat_shift, int B, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------
 * DAM batch norm (SIFABatchNorm2d.forward eval branch, ttda:299-345).
 * ------------------------------------------------------------------------------------------ */
/* Reduce a statistics slab in fixed unit order (deterministic) and produce the per-channel affine
 *   mean_g = K + S1_g/n_g,  var_g = S2_g/n_g - (S1_g/n_g)^2            (g = A, B; biased var)
 *   mean_cur = n_b > 0 ? (mean_A + mean_B)/2 : mean_A   (same for var)   ttda:315-320
 *   mean = lam*running_mean + (1-lam)*mean_cur, var likewise (1-lam in fp32)   ttda:335-336
 *   scale = weight / sqrt(var + eps),  shift = bias - mean*scale        ttda:341-343
 * slab: fp32 [nsamples*units_per_sample][2][C], sample-major.
 * per_sample == 0: the reference's batch rule (half_first): group A = the first a_samples samples,
 *   group B = the rest; n_a / n_b = element counts of the groups; scale/shift fp32 [C].
 * per_sample != 0: every sample is normalised with its own statistics, i.e. nsamples independent
 *   batch-1 frames of a stream ride in one launch; n_a = elements per sample, n_b = 0;
 *   scale/shift fp32 [nsamples][C].  Sample s's result is bit-identical to a batch-1 call. */
int diga_bn_finalize(const float* slab, int units_per_sample, int nsamples, int a_samples,
                     int per_sample, int C, double n_a, double n_b, const float* running_mean,
                     const float* running_var, const float* weight, const float* bias, float lam,
                     float eps, float* scale, float* shift, void* stream);

/* Multi-GPU form of the same statistic: every rank sums ITS slab over its units into
 * sums[2 groups][2][C] in fp64 (group A = its first a_samples samples: 1 on rank 0, 0 elsewhere), the
 * ranks all-reduce(sum) that buffer, and diga_bn_finalize_sums turns the global sums into scale/shift
 * with the arithmetic of diga_bn_finalize (n_a / n_b = global element counts).  fp64 keeps the
 * exchange exact enough that a sharded batch reproduces the single-process batch bit for bit in
 * practice (sums of exactly the same fp32 unit values, rounded to fp32 only after the division).
 * per_sample != 0 (a stream of global batches, k in flight): local sample j belongs to global batch j; its
 * sums go to set j, sums[nsamples][2][2][C], as group A if a_samples > 0 (rank 0) else as group B;
 * diga_bn_finalize_sums(nsets = k) then writes scale/shift [k][C]. */
int diga_bn_reduce_slab(const float* slab, int units_per_sample, int nsamples, int a_samples,
                        int per_sample, int C, double* sums, void* stream);
int diga_bn_finalize_sums(const double* sums, int nsets, int C, double n_a, double n_b,
                          const float* running_mean, const float* running_var, const float* weight,
                          const float* bias, float lam, float eps, float* scale, float* shift,
                          void* stream);

/* y = relu?( raw*scale + shift  [+ res_raw*res_scale + res_shift | + res] ) -> bf16 [M][C]
 * (adaptseg:86-87,90-91,94,96-100).  raw, res: bf16 [M][C]; res_scale == NULL means `res` is an
 * already-normalised activation; res == NULL means no residual.  nsamples_affine == 0: one affine
 * [C] for all rows; > 0: the M rows are that many equal samples and scale/shift (and res_scale/
 * res_shift) are [nsamples_affine][C] (per-sample statistics, see diga_bn_finalize). */
int diga_bn_apply(const void* raw, const float* scale, const float* shift, const void* res,
                  const float* res_scale, const float* res_shift, void* out, int64_t M, int C,
                  int relu, int nsamples_affine, void* stream);

/* diga_bn_finalize followed by diga_bn_apply (no residual / normalised residual) in ONE launch (ttda:313-345 +
 * adaptseg:86-87,90-91,94-100): for small passes (one frame per step), where both kernels sit at their launch floor.
 * Every block repeats the fixed-order slab reduction of its 64 channels, so scale / shift (also written out, as
 * diga_bn_finalize does: [C], or [nsamples][C] with per_sample) and `out` are bit-identical to the two-launch form.
 * M = all rows; per_sample: M is nsamples equal samples, each with its own statistics. */
int diga_bn_finalize_apply(const float* slab, int units_per_sample, int nsamples, int a_samples,
                           int per_sample, int C, double n_a, double n_b, const float* running_mean,
                           const float* running_var, const float* weight, const float* bias, float lam,
                           float eps, float* scale, float* shift, const void* raw, const void* res, void* out,
                           int64_t M, int relu, void* stream);

/* The stand-alone module form (SIFABatchNorm2dTrainable.forward on an fp32 NCHW tensor, ttda:299-345):
 * per-channel shifted sums with 64-lane wavefront-shuffle reductions -> slab [2 groups][2][C] for
 * diga_bn_finalize(units_per_sample 1, nsamples 2, a_samples 1) (group A = samples [0, split_b),
 * group B = the rest), then y = x*scale + shift. */
int diga_bn_stats_nchw_f32(const float* x, int B, int C, int HW, int split_b, const float* stat_shift,
                           float* slab, void* stream);
int diga_bn_apply_nchw_f32(const float* x, const float* scale, const float* shift, float* out, int B,
                           int C, int HW, void* stream);

/* bn1 + ReLU + MaxPool2d(3, stride 2, pad 1, ceil_mode=True) fused (adaptseg:179-181,139):
 * raw bf16 NHWC [B,H,W,C] -> bf16 NHWC [B,Ho,Wo,C], Ho = ceil((H-1)/2)+1 (PyTorch ceil rule).
 * per_sample != 0: scale/shift are [B][C]. */
int diga_bn_relu_maxpool(const void* raw, const float* scale, const float* shift, void* out, int B,
                         int H, int W, int C, int Ho, int Wo, int per_sample, void* stream);

/* ------------------------------------------------------------------------------------------
 * SAM (ttda:1662-1805) and fusion (ttda:1644)
 * ------------------------------------------------------------------------------------------ */
/* Sum split-K head partials + bias -> logits fp32 NCHW [B,C,h,w]; softmax/argmax/threshold ->
 * pseudo-labels int32 [B*h*w] (255 = not confident; strict '<', ttda:1675) (ttda:1662-1676).
 * partials: fp32 [splitk][M][cpad] (M = B*h*w) or NULL when logits_in is given (NCHW fp32).
 * label_sample_stride: distance in elements between the label rows of consecutive samples
 * (0 = hw, dense); a stream of batch-1 frames keeps [pseudo | gt] per sample, i.e. stride 2*hw. */
int diga_head_finish_pseudo_label(const float* partials, int splitk, int cpad, const float* bias_sum,
                                  const float* logits_in, float* logits_out, int32_t* labels,
                                  int B, int C, int hw, float threshold, int64_t label_sample_stride,
                                  void* stream);

/* The same head as ONE dense GEMM + tap gather: z = fp32 [B*h*w][zstride] with z[m][t*C + c] = feat[m] . W_t[c]
 * (a 1x1 diga_conv_igemm_bf16 with Cout = ntaps*C, out_f32, that reads the feature map once instead of ntaps
 * times); logits[m][c] = bias_sum[c] + sum over the taps t whose pixel m + (dy[t], dx[t]) lies inside the image of
 * z[that pixel][t*C + c] (adaptseg:119-121, zero padding); then the pseudo-label rule above.
 * dy_host / dx_host: HOST arrays of ntaps tap offsets in feature pixels. */
int diga_head_tapsum_pseudo_label(const float* z, int zstride, int ntaps, const int16_t* dy_host,
                                  const int16_t* dx_host, const float* bias_sum, float* logits_out,
                                  int32_t* labels, int B, int C, int h, int w, float threshold,
                                  int64_t label_sample_stride, void* stream);

/* F.interpolate(y.float(), size=(h,w), mode='nearest').long() (ttda:1695): src = floor(dst*in/out)
 * computed as the reference does (fp32 scale).  y: int64 (y_is_i64) or int32 [B,H,W] -> int32 [B,h,w];
 * out_sample_stride as label_sample_stride above (0 = dense). */
int diga_nearest_down_labels(const void* y, int y_is_i64, int32_t* out, int B, int H, int W, int h,
                             int w, int64_t out_sample_stride, void* stream);

/* cal_proto (ttda:1788-1805) for up to two label sets in one call.  Stable (pixel-order) per-class
 * index lists, then chunked register-accumulated row sums -- deterministic.
 * feat: bf16 [P][CH]; labels: int32 [nsets][P]; work: scratch, diga_proto_pool_work_bytes() bytes;
 * sums: fp32 [nsets][C][CH]; counts: int32 [nsets][C]. */
int64_t diga_proto_pool_work_bytes(int nsets, int64_t P, int C, int CH);
int diga_proto_pool(const void* feat, const int32_t* labels, int nsets, int64_t P, int C, int CH,
                    void* work, float* sums, int32_t* counts, void* stream);
/* The same for nframes independent frames in one set of launches (a stream of batch-1 frames in flight: pooling does
 * not depend on the running prototypes, so all frames are pooled before the per-frame EMA / classifier chain).
 * feat: bf16 [nframes][P][CH]; labels: int32 [nframes][nsets][P]; sums: fp32 [nframes][nsets][C][CH]; counts: int32
 * [nframes][nsets][C]; every frame's result is bit-identical to its own diga_proto_pool call. */
int64_t diga_proto_pool_frames_work_bytes(int nframes, int nsets, int64_t P, int C, int CH);
int diga_proto_pool_frames(const void* feat, const int32_t* labels, int nframes, int nsets, int64_t P, int C,
                           int CH, void* work, float* sums, int32_t* counts, void* stream);

/* Instance prototypes + running-prototype seed / EMA (ttda:1700-1705, 1736-1765).
 * sums/counts: as produced by diga_proto_pool with set 0 = pseudo-labels, set 1 = ground truth
 * (all-reduced over ranks in the sharded path).  state_flag: int32[1] device word, 0 before the
 * first frame; set to 1 by this call.  one_minus_rho / rho are passed separately because the
 * reference forms them in double precision (ttda:1764). */
int diga_proto_update(const float* sums, const int32_t* counts, int C, int CH, float* inst_proto,
                      float* inst_flag, float* run_proto, float* run_flag, int32_t* state_flag,
                      float one_minus_rho, float rho, void* stream);

/* The same chain for nframes consecutive batch-1 frames of a stream in one launch pair (the frames' sums / counts come
 * from diga_proto_pool_frames): frame f's instance prototypes and the running prototypes AFTER frame f's update --
 * what frame f's classifier reads -- are stored per frame; run_proto / run_flag / state_flag carry the state in and out.
 * sums fp32 [nframes][2][C][CH], counts int32 [nframes][2][C]; inst_protos, run_protos fp32 [nframes][C][CH];
 * inst_flags, run_flags fp32 [nframes][C].  Every frame's values are bit-identical to nframes diga_proto_update calls. */
int diga_proto_update_frames(const float* sums, const int32_t* counts, int nframes, int C, int CH,
                             float* inst_protos, float* inst_flags, float* run_protos, float* run_flags,
                             float* run_proto, float* run_flag, int32_t* state_flag, float one_minus_rho,
                             float rho, void* stream);

/* cal_pred_of_proto x2 + proto blend + softmax fusion (ttda:1707-1709,1712-1734,1644):
 *   out = fl*(pl*P_run + pl1*P_inst) + fl1*softmax(logits),  P_x[c] = exp(-||proto_x[c]-f||)*flag/sum
 * 0/0 -> NaN is reproduced.  feat bf16 [P][CH] (CH % 64 == 0); logits fp32 NCHW [B,C,hw]; out fp32 NCHW
 * [B,C,hw]; work: scratch of diga_proto_classify_work_bytes(P = B*hw, CH) bytes (prototype image + partial squared distances). */
int64_t diga_proto_classify_work_bytes(int64_t P, int CH);
int diga_proto_classify_fuse(const void* feat, const float* logits, const float* inst_proto,
                             const float* inst_flag, const float* run_proto, const float* run_flag,
                             float* out, void* work, int B, int C, int hw, int CH, float tau, float pl,
                             float pl1, float fl, float fl1, void* stream);
/* The frames-in-flight form: sample f is classified against ITS prototypes and flags (the per-frame arrays of
 * diga_proto_update_frames); work: diga_proto_classify_frames_work_bytes(nframes, P = nframes*hw, CH) bytes. */
int64_t diga_proto_classify_frames_work_bytes(int nsets, int64_t P, int CH);
int diga_proto_classify_fuse_frames(const void* feat, const float* logits, const float* inst_protos,
                                    const float* inst_flags, const float* run_protos, const float* run_flags,
                                    float* out, void* work, int nframes, int C, int hw, int CH, float tau,
                                    float pl, float pl1, float fl, float fl1, void* stream);

/* diga_proto_classify_fuse (batch form, one prototype set) for callers that CANNOT ALLOCATE device memory -- code under
 * HIP-graph capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  The
 * workspace (prototype image + partial squared distances) is a file-scope device array inside the library, 20.5 MiB per
 * GPU the library runs on, held until the process ends; the call allocates and frees nothing.
 * Limits: B*hw <= DIGA_NOALLOC_MAX_PIXELS and CH <= DIGA_NOALLOC_MAX_CH, else DIGA_EINVAL with a message.
 * The first call on a device resolves the array's device address (hipGetSymbolAddress; make that call outside stream
 * capture).  All calls on one GPU share the array: calls that overlap on one GPU (two streams, two threads) race --
 * serialise them.  Results are bit-identical to diga_proto_classify_fuse. */
#define DIGA_NOALLOC_MAX_PIXELS 16384
#define DIGA_NOALLOC_MAX_CH 2048
int diga_proto_classify_fuse_noalloc(const void* feat, const float* logits, const float* inst_proto,
                                     const float* inst_flag, const float* run_proto, const float* run_flag,
                                     float* out, int B, int C, int hw, int CH, float tau, float pl, float pl1,
                                     float fl, float fl1, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optional SAM extension (BASELINE.json north_star (2)).  NO counterpart in the reference: DIGA as
 * published has no connected-component step, the default path never calls these, and with respect to the
 * reference their results are "parity unpinned".  Specification (the build's own), pinned against
 * scipy.ndimage.label (tests/golden/fn_ccl.npz):
 * ------------------------------------------------------------------------------------------ */
/* Connected components of the per-class masks of a label map, union-find with canonical labels.
 * labels: int32, sample b at labels + b*label_sample_stride (0 = h*w, dense), h*w values each; a value in
 *   [0, C) puts the pixel in that class's mask, anything else (255 = ignore) is background.
 * connectivity: 4 or 8.
 * inst: int32 [B][h*w]: 0 = background, else the number of the pixel's component inside ITS class's mask, counted
 *   from 1 in raster order of the components' first pixels  (== scipy.ndimage.label(labels[b] == c, structure)).
 * ncomp: int32 [B][C] components per class.  work: diga_ccl_work_bytes() bytes of scratch. */
int64_t diga_ccl_work_bytes(int B, int h, int w, int C);
int diga_ccl_label(const int32_t* labels, int64_t label_sample_stride, int B, int h, int w, int C,
                   int connectivity, int32_t* inst, int32_t* ncomp, void* work, void* stream);

/* Instance-masked prototype pooling: the mean feature row of every component found by diga_ccl_label.
 * feat: bf16 [B*hw][CH] (CH % 8 == 0); labels / inst / ncomp as above.  Row g of the instance table is component
 * (sample b, class c, number i) with g = (number of components before (b, c) in (sample, class) order) + i - 1.
 * protos: fp32 [max_inst][CH] (rows >= min(total, max_inst) are not written); counts (pixels), inst_class,
 * inst_sample: int32 [max_inst] (0 / -1 / -1 in unused rows); n_total: int32 [1] = number of components found --
 * when it exceeds max_inst the surplus rows were dropped.  Stable pixel-order lists and fixed-order sums:
 * deterministic, no floating-point atomics.  work: diga_instance_pool_work_bytes() bytes. */
int64_t diga_instance_pool_work_bytes(int B, int64_t hw, int C, int CH, int max_inst);
int diga_instance_pool(const void* feat, const int32_t* labels, int64_t label_sample_stride,
                       const int32_t* inst, const int32_t* ncomp, int B, int64_t hw, int C, int CH,
                       int max_inst, float* protos, int32_t* counts, int32_t* inst_class,
                       int32_t* inst_sample, int32_t* n_total, void* work, void* stream);

/* ------------------------------------------------------------------------------------------
 * Tail (ttda:1610, 158-173): bilinear upsample (align_corners=False) + argmax + confusion matrix.
 * probs fp32 NCHW [B,C,h,w]; target int64/int32 [B,H,W]; pred uint8 [B,H,W] (may be NULL);
 * hist int64 [C*C] += bincount(C*target+pred) * B for 0 <= target < C; probs_up (may be NULL)
 * fp32 NCHW [B,C,H,W]. */
int diga_upsample_argmax_hist(const float* probs, const void* target, int target_is_i64, uint8_t* pred,
                              int64_t* hist, float* probs_up, int B, int C, int h, int w, int H, int W,
                              void* stream);
/* SegmentationMetric.update called on full-resolution probabilities (ttda:178-179, 168-173): preds.argmax(1) + confusion counts
 * (x B), no interpolation -- a NaN pixel (the 0/0 branch, ttda:1733) stays one pixel. */
int diga_argmax_hist(const float* probs, const void* target, int target_is_i64, uint8_t* pred, int64_t* hist,
                     int B, int C, int H, int W, void* stream);
/* nframes batch-1 metric updates (a stream of frames in flight) in one launch: the same as nframes calls with
 * B = 1 -- the counts are integers, so the sum does not depend on the order -- i.e. hist += bincount * 1 per frame. */
int diga_upsample_argmax_hist_frames(const float* probs, const void* target, int target_is_i64, uint8_t* pred,
                                     int64_t* hist, float* probs_up, int nframes, int C, int h, int w, int H,
                                     int W, void* stream);

/* ------------------------------------------------------------------------------------------
 * One-shot all-reduce over direct peer writes (no counterpart in the reference, which has no distributed code).
 * For the sharded path's small, latency-bound exchanges (per BN layer [2][2][C] fp64 sums, per step the prototype sums):
 * every rank writes its contribution into a slot of every peer's buffer (hipIpc-mapped, one hop over the direct xGMI link),
 * raises a flag, waits (bounded) for the peers' flags on its own buffer and sums the slots in rank order -- one launch,
 * deterministic, identical bits on every rank, capturable in a HIP graph.
 * Set-up (these three allocate / map memory; the exchange itself does not):
 *   diga_xgmi_create   -> local buffer + its 64-byte hipIpc handle; exchange the handles of all ranks out of band
 *   diga_xgmi_connect  -> handles: world x 64 bytes in rank order (the own entry is ignored)
 *   diga_xgmi_destroy
 * diga_xgmi_allreduce: in-place sum over ranks of `count` values at `data` (device, 8-byte aligned); dtype 0 = fp64,
 *   1 = fp32, 2 = int32 (4-byte types: count even); count*size <= max_elems*8.  Every rank must issue the same sequence of
 *   calls (same counts) on ONE stream per communicator (two streams = two communicators).
 * Ordering: the buffer is uncached device memory and every access to it is a system-scope atomic, so "payload before flag" /
 *   "flag before payload" are this wave's own `s_waitcnt vmcnt(0)` between them; no cache-wide write-back / invalidate is issued
 *   (a buffer that could not be allocated uncached, or DIGA_XGMI_FENCES=1 in the environment: system-scope fences instead).
 * Waits are bounded by wall-clock time (default 30 s, diga_xgmi_set_timeout_ms): when a peer has not arrived by then the launch
 *   ends, POISONS its payload (NaN for fp64 / fp32, INT32_MIN for int32, NaN scale / shift) and raises a sticky error word;
 *   diga_xgmi_check (synchronises the device) turns that word into DIGA_ELAUNCH.  A communicator that has failed is dead.
 * diga_bn_exchange_finalize: the exchange FUSED into the DAM finalize -- one launch per BN layer instead of
 *   diga_bn_reduce_slab + diga_xgmi_allreduce + diga_bn_finalize_sums (same arguments as the first and the last; same fixed
 *   summation orders, so scale / shift are bit-identical to the three-launch form): every (64-channel block, set) block reduces
 *   its part of the local slab and writes its fp64 group sums into every rank's buffer; the last block of a channel block
 *   raises the flag, waits for the peers, sums in rank order and writes scale / shift [nsets][C] (ttda:313-343 on the GLOBAL
 *   batch).  Limits: C % 64 == 0, C <= 2048, at most 16 sets (per_sample: nsamples <= 16).
 * ------------------------------------------------------------------------------------------ */
int64_t diga_xgmi_buffer_bytes(int world, int64_t max_elems);
int diga_xgmi_create(int world, int rank, int64_t max_elems, void** comm_out, void* ipc_handle_out64);
int diga_xgmi_connect(void* comm, const void* handles_world_x64);
int diga_xgmi_destroy(void* comm);
int diga_xgmi_allreduce(void* comm, void* data, int64_t count, int dtype, void* stream);
int diga_xgmi_check(void* comm);
int diga_xgmi_set_timeout_ms(void* comm, int timeout_ms);
int diga_bn_exchange_finalize(void* comm, const float* slab, int units_per_sample, int nsamples, int a_samples,
                              int per_sample, int C, double n_a, double n_b, const float* running_mean,
                              const float* running_var, const float* weight, const float* bias, float lam, float eps,
                              float* scale, float* shift, void* stream);

/* ------------------------------------------------------------------------------------------
 * Input pipeline: the arithmetic the reference's datasets apply to a decoded frame.
 * ------------------------------------------------------------------------------------------ */
/* BaseDataset.preprocess (src/datamodules/seg/base_dataset.py:45-52): img HWC RGB (uint8 or the
 * reference's float32) [B,H,W,3] -> out fp32 CHW [B,3,H,W]; swap_rb != 0: channel c of the output is
 * input channel 2-c (RGB -> BGR) minus mean[c]; swap_rb == 0: the BDD100K / Mapillary form
 * (bdd100k.py:290: mean subtracted from unswapped RGB).  The subtraction runs in fp64 and is rounded
 * to fp32 once, as numpy does for `float32_array -= float64_array`. */
int diga_preprocess_image(const void* img, int img_is_u8, float* out, int B, int H, int W, double mean0,
                          double mean1, double mean2, int swap_rb, void* stream);

/* CityscapesDataSet.map_labels (cityscapes.py:50-51): out[i] = lut[(int64)ids[i]] with numpy's negative
 * indexing; ids: uint8 (ids_kind 0), int32 (1) or the reference's float32 (2); lut: int32 [nlut] on the
 * device; out int64 (out_is_i64) or int32; ids outside the table -> 255 (numpy would raise). */
int diga_map_labels(const void* ids, int ids_kind, const int32_t* lut, int nlut, void* out, int out_is_i64,
                    int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * fp32 forms: (a) the verification mode (fp32-exact end-to-end parity run: the reference is fp32 end to end,
 * ttda:1634-1645; activations travel as fp32 + bf16 operand planes, see diga_conv_desc.nplanes) and (b) the helper API
 * called with arbitrary fp32 feature tensors (cal_proto / cal_pred_of_proto, ttda:1712-1805, compute in fp32).
 * ------------------------------------------------------------------------------------------ */
/* diga_stem_conv7x7s2 with the raw output kept in fp32 NHWC [B,Ho,Wo,64]. */
int diga_stem_conv7x7s2_f32(const float* x, const float* wgt, float* out, float* stat_slab,
                            const float* stat_shift, int B, int H, int W, void* stream);
/* diga_bn_apply on fp32 data: raw, res fp32 [M][C]; y -> out_f32 (fp32 [M][C], may be NULL) and/or out_planes (bf16
 * [nplanes][M][C], may be NULL): plane 0 = bf16(y), plane q = bf16(y - planes before it) (adaptseg:86-100).
 * nsamples_affine as diga_bn_apply (> 0: per-sample affines [nsamples][C]: frames of a stream in flight). */
int diga_bn_apply_f32(const float* raw, const float* scale, const float* shift, const float* res,
                      const float* res_scale, const float* res_shift, float* out_f32, void* out_planes,
                      int nplanes, int64_t M, int C, int relu, int nsamples_affine, void* stream);
/* diga_bn_relu_maxpool on fp32 data (adaptseg:179-181), outputs as diga_bn_apply_f32; per_sample as diga_bn_relu_maxpool. */
int diga_bn_relu_maxpool_f32(const float* raw, const float* scale, const float* shift, float* out_f32,
                             void* out_planes, int nplanes, int B, int H, int W, int C, int Ho, int Wo, int per_sample,
                             void* stream);
/* x fp32 [n] -> bf16 planes [nplanes][n] whose sum is x (weights / inputs of the verification mode). */
int diga_split_planes_f32(const float* x, void* planes, int nplanes, int64_t n, void* stream);
/* diga_proto_pool / diga_proto_classify_fuse with fp32 features [P][CH] (same work sizes, same arithmetic order). */
int diga_proto_pool_f32(const float* feat, const int32_t* labels, int nsets, int64_t P, int C, int CH,
                        void* work, float* sums, int32_t* counts, void* stream);
int diga_proto_classify_fuse_f32(const float* feat, const float* logits, const float* inst_proto,
                                 const float* inst_flag, const float* run_proto, const float* run_flag,
                                 float* out, void* work, int B, int C, int hw, int CH, float tau, float pl,
                                 float pl1, float fl, float fl1, void* stream);
int diga_nchw_f32_to_nhwc_f32(const float* in, float* out, int B, int HW, int C, void* stream);
int diga_nhwc_f32_to_nchw_f32(const float* in, float* out, int B, int HW, int C, void* stream);

/* Layout helpers for the Python-visible API (net(x, feat=True) returns NCHW fp32, adaptseg:188). */
int diga_nhwc_bf16_to_nchw_f32(const void* in, float* out, int B, int HW, int C, void* stream);
int diga_nchw_f32_to_nhwc_bf16(const float* in, void* out, int B, int HW, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIGA_HIP_H */
