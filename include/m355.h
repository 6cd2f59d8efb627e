/*
 * m355.h -- C-ABI of libm355.so, the MI355X (gfx950) hot path of 2dimageto3dmodel_amd.
 *
 * The reference (NikolaZubic/2dimageto3dmodel) is pure Python: its "FFI" for this path is the
 * nn.Module / function surface listed in SURVEY.md section 8b.  Every entry point below replaces the
 * body of one of those Python functions (file:line relative to /root/reference/code); the Python
 * classes of the same name in 2dimageto3dmodel_amd/ bind them through ctypes.
 *
 * Conventions
 *   - every function returns 0 on success, a negative m355_status otherwise; m355_last_error()
 *     returns a thread-local description.  Nothing throws across the boundary.
 *   - no allocation, no ownership transfer: all pointers are caller-owned DEVICE pointers
 *     (torch tensors' data_ptr()), fp32 contiguous unless stated; workspaces are sized by the
 *     *_ws_bytes queries.
 *   - asynchronous on the hipStream_t passed as `stream` (void* here so the header needs no HIP).
 *   - re-entrant for distinct streams/devices; no global mutable state besides the error string.
 */
#ifndef M355_H
#define M355_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    M355_OK = 0,
    M355_ERR_BAD_ARG = -1,      /* null pointer / non-positive size / unsupported size */
    M355_ERR_UNSUPPORTED = -2,  /* combination not handled by this kernel (caller picks another path) */
    M355_ERR_LAUNCH = -3        /* HIP launch failure (message holds hipGetErrorString) */
} m355_status;

const char *m355_last_error(void);
/* name of the kernel family the calling thread's last m355_conv2d_* call dispatched to (profiling aid) */
const char *m355_last_kernel(void);
int m355_abi_version(void);   /* 2: deterministic reductions (round 4): workspaces on cproj_bwd / head_tail_bwd / mesh_flat_fwd, gather tables on the mesh backward, m355_conv2d_wgrad_det; 3 (round 5): m355_act_bytes; 4 (round 6): m355_conv_plan.w_dgrad_row_elems / .wgrad_ws_ordered, m355_ipc_* */
/* Bytes of an ACTIVATION element ("bf16" in the comments below) in this build of the library: 2 = bf16, the product; 4 = fp32, the
 * EXACT build (lib/libm355_exact.so, compiled from the same sources with -DM355_EXACT; SURVEY.md 8c "an fp32-accumulate exact mode
 * for 1e-4 checks").  Same entry points, same argument meaning; the GAN path's activation tensors, conv operands and weight views
 * are then fp32 and the convolutions run on plain fp32 kernels with fp64 accumulation (csrc/conv_exact.hip) -- what the module-level
 * parity tests hold against the reference's fp32 CPU results (models/gan.py, utils/losses.py, main.py:431-447,588-589,691-723). */
int m355_act_bytes(void);

/* flags for the projection entry points */
#define M355_FIXED_WEIGHTS 1   /* w0 = 1-(g-floor g) instead of the literal 1-g-floor g (trilinear_interpolation.py:66) */
#define M355_TAPS_FROM_SIGMA 2 /* `taps` points at the device scalar sigma; taps are built in-kernel (smooth_voxels.py:24-31) */
#define M355_TRUE_GAUSSIAN 4   /* exp(-x^2/2s^2) instead of the literal exp(+x^2/2s^2) (smooth_voxels.py:29) */
#define M355_DET_SPLAT 8       /* m355_proj_render_fwd/_bwd (21 explicit taps): the occupancy splat accumulates in 64-bit fixed-point
                                  LDS cells -- bit-reproducible whatever the voxel collisions; one workgroup per CU instead of four */

/* ---- P1+P2  CameraUtilities.transformation_3d_coord_to_camera_coord
 *      camera/coordinate_system_transformation.py:20-39 (+ quaternions/points_quaternions.py:41-81,
 *      quaternions/operations.py:68-97,120-136); the reference passes fov 1.875, distance 2.0.
 *      pc[B,N,3], q[B,4] -> cam[B,N,3] in (z,y,x) order, BIT-EXACT with the torch-CPU reference. */
int m355_proj_transform_fwd(const float *pc, const float *q, float *cam, int B, int N, float fov, float dist,
                            void *stream);

/* ---- P1 alone  PointsQuaternionsRotator.rotate_points, quaternions/points_quaternions.py:41-81 (with
 *      QuaternionOperations.quaternion_multiplication / quaternion_conjugate, quaternions/operations.py:68-97,120-136):
 *      out[b,n,:] = vector part of q (x) (0,p) (x) q*  (inverse = 0, :72-75)  or  q* (x) (0,p) (x) q  (inverse = 1, :67-70)
 *      with q = F.normalize(rotation); pc/out [B,N,3], q [B,4]; BIT-EXACT with the torch-CPU reference.
 *      _bwd: dout [B,N,3] -> dpc [B,N,3], dq [B,4] (through the normalisation). */
int m355_quat_rotate_fwd(const float *pc, const float *q, float *out, int B, int N, int inverse, void *stream);
int m355_quat_rotate_bwd(const float *pc, const float *q, const float *dout, float *dpc, float *dq, int B, int N, int inverse,
                         void *stream);

/*      The same transform fused with the per-tile binning the fused renderer consumes (one launch).
 *      A tile is a TH x TW block of rays (silhouette pixels); m355_proj_ntiles(S) tiles cover the S x S image.
 *      pc/q both NULL: cam is an INPUT and only the binning runs.
 *      raykey (nullable, [B,N] int32): (floor g1 << 16 | floor g2), or -1 when the point fails the in-bounds
 *      test of trilinear_interpolation.py:24 -- the "index-exact bins" contract, exposed for the parity tests.
 *      tile_start[B, ntiles+1] int32: exclusive prefix of the per-tile record counts.
 *      tile_pts[B, 4N, 4] fp32: records (c0,c1,c2, bitcast(n)); a point is filed under every tile its 2x2 ray
 *      footprint touches (1, 2 or 4 tiles). */
int m355_proj_ntiles(int S);
int m355_proj_bin_fwd(const float *pc, const float *q, float *cam, int32_t *raykey, int32_t *tile_start,
                      float *tile_pts, int B, int N, int S, float fov, float dist, void *stream);

/*      backward: dcam[B,N,nslots,3] (slots summed in order) -> dpc[B,N,3], dq[B,4].
 *      mask_oob != 0: points failing the in-bounds test get zero gradient and their slots are not read.
 *      dscale_part[B,nparts] -> dscale[B] (deterministic second-stage reduce; both nullable). */
int m355_proj_transform_bwd(const float *pc, const float *q, const float *dcam, int nslots, int mask_oob,
                            float *dpc, float *dq, const float *dscale_part, int nparts, float *dscale, int B,
                            int N, float fov, float dist, void *stream);

/* ---- P4 taps  VoxelsSmooth.separate_kernels (utils/smooth_voxels.py:14-42): sigma (device scalar) -> taps[ntaps],
 *      x = -ntaps//2+1 .. ntaps//2, exp(+x^2/(2 sigma^2)) / sum unless M355_TRUE_GAUSSIAN. */
int m355_smooth_taps(const float *sigma, int ntaps, int flags, float *taps, void *stream);

/* ---- P3..P6 fused  EffectiveLossFunction.forward  (utils/effective_loss_function.py:58-81 with shims S0/S1):
 *      trilinear splat (utils/trilinear_interpolation.py:37-74) -> depth-axis smoothing
 *      (utils/smooth_voxels.py:44-84, literal behaviour: only the last = depth kernel survives) ->
 *      scale+clamp -> termination_probs (elf:18-56) -> sum over depth, flip y (elf:81).
 *      The S^3 occupancy volume lives in LDS tiles only and never touches HBM.
 *      tile_start/tile_pts from m355_proj_bin_fwd, scale[B] nullable, taps[ntaps] (odd, <= 63; 21 takes the
 *      tuned kernel), proj[B,S,S].  S <= 512. */
int m355_proj_render_fwd(const int32_t *tile_start, const float *tile_pts, const float *scale, const float *taps,
                         int ntaps, float *proj, int B, int N, int S, int flags, void *stream);

/*      backward: dproj[B,S,S] * gmul -> dcam_slots[B,N,4,3] (one slot per ray (j,k) of the point, written
 *      exactly once for in-bounds points, never for the others), dscale_part[B, m355_proj_ntiles(S)]. */
int m355_proj_render_bwd(const int32_t *tile_start, const float *tile_pts, const float *scale, const float *taps,
                         int ntaps, const float *dproj, float gmul, float *dcam_slots, float *dscale_part, int B,
                         int N, int S, int flags, void *stream);

/* ---- P3 dense  TrilinearInterpolation.trilinear_interpolation (utils/trilinear_interpolation.py:62-74):
 *      tile lists (m355_proj_bin_fwd with pc = q = NULL bins an existing cam) -> vol[B,S,S,S] = clamp(splat,0,1);
 *      raw (nullable) receives the un-clamped sums. */
int m355_trilinear_fwd(const int32_t *tile_start, const float *tile_pts, float *vol, float *raw, int B, int N, int S,
                       int flags, void *stream);
/*      dvol[B,S,S,S] -> dcam[B,N,3] (overwritten; the clamp mask is recomputed from the re-splatted sums) */
int m355_trilinear_bwd(const int32_t *tile_start, const float *tile_pts, const float *dvol, float *dcam, int B, int N,
                       int S, int flags, void *stream);

/* ---- P4 dense  VoxelsSmooth.smooth (utils/smooth_voxels.py:44-84), one axis per call: axis 0 depth, 1 y, 2 x;
 *      out = conv1d(in, taps, zero pad ntaps/2) [* scale[b], clamp(0,1) when scale != NULL]; transpose != 0 applies
 *      the adjoint of the convolution (backward). */
int m355_smooth_axis(const float *in, float *out, const float *taps, int ntaps, int axis, const float *scale,
                     int transpose, int B, int S, void *stream);
/*      backward of the scale/clamp epilogue: pre = conv output before scaling; dout -> dpre, dscale[B] */
int m355_scale_clamp_bwd(const float *pre, const float *scale, const float *dout, float *dpre, float *dscale, int B,
                         size_t per_sample, void *stream);

/* ---- P5 dense  EffectiveLossFunction.termination_probs (elf:18-56): vol[B,D,H,W] -> T[B,D+1,H,W] */
int m355_termination_fwd(const float *vol, float *T, int B, int D, int H, int W, float eps, void *stream);
int m355_termination_bwd(const float *vol, const float *dT, float *dvol, int B, int D, int H, int W, float eps,
                         void *stream);

/* ---- P7/P8  SupervisedLoss.forward (models/supervised_part.py:68-72) and the per-cloud SSE used by
 *      UnsupervisedLoss.forward (models/unsupervised_part.py:108-126):
 *      mask[B/mask_repeat,Hin,Win] (row b/mask_repeat serves cloud b: batch_repetition.py:6-19)
 *      -bilinear 1/2, align_corners- m[B,S,S]; diff = proj - m; sse[b] = sum diff^2;
 *      total = sum_b sse[b].  ws >= m355_sil_loss_ws_bytes(B,S). */
size_t m355_sil_loss_ws_bytes(int B, int S);
int m355_sil_loss_fwd(const float *proj, const float *mask, int Hin, int Win, int mask_repeat, float *diff,
                      float *sse, float *total, void *ws, int B, int S, void *stream);

/* ---- Chamfer nearest neighbour (BASELINE configs[4]).  NEW capability: the reference contains no Chamfer code
 *      (SURVEY.md 0.3), parity is against the brute-force oracle only.
 *      a[B,N,3], b[B,M,3] -> dist[B,N] = min_j |a_i-b_j|^2, idx[B,N] = argmin (lowest j on ties); the squared distance is
 *      the fused chain fma(dz, dz, fma(dy, dy, dx * dx)) on both sides (oracle/p_oracle.c orc_chamfer_nn): bit-exact. */
int m355_chamfer_nn_fwd(const float *a, const float *b, float *dist, int32_t *idx, int B, int N, int M,
                        void *stream);
/*      the same with a workspace: when B * ceil(N / 256) < 256 query blocks would leave most of the chip idle (B = 1, 16384
 *      points: 64), the TARGET sweep is split over workgroups as well: every slice stores its (distance bits << 32 | chunk)
 *      key per query into its own row of ws, a second launch takes the minimum over the rows -- the lowest-index tie rule is
 *      that key's order; no atomics, no fill.  ws >= m355_chamfer_nn_ws_bytes(B,N,M) (0: not needed, ws may be NULL). */
size_t m355_chamfer_nn_ws_bytes(int B, int N, int M);
int m355_chamfer_nn_fwd_ws(const float *a, const float *b, float *dist, int32_t *idx, int B, int N, int M, void *ws,
                           void *stream);

/* ---- G  conv2d of the GAN stacks (models/gan.py:57-65,163-177,294-302,359,364 -> F.conv2d) as a bf16 MFMA
 *      implicit GEMM with fp32 accumulation.  Activations are NHWC bf16, weights bf16 views built by
 *      m355_conv2d_weight_prep from the fp32 [Cout][Cin][kh][kw] parameter.
 *      The pads the reference materialises in front of the convs and the nearest x2 upsample are folded into
 *      the loader: pad_w_mode 0 zero, 1 replicate (F.pad replicate, gan.py:329), 2 circular (circpad,
 *      rendering/utils.py:60-64); H is always zero padded (Conv2d padding=(p,0), gan.py:294).
 *      Cin must be a multiple of 8 (callers zero-pad the channels); dy carries m355_conv2d_dy_channels(Cout)
 *      channels (8 when Cout <= 8, else Cout rounded up to 32; the extra channels are zero). */
typedef struct {
    int N, H, W, Cin;   /* stored input (before the optional upsample) */
    int Cout, kh, kw;
    int stride;         /* 1 or 2 */
    int pad_h, pad_w;
    int pad_w_mode;
    int upsample;       /* 0 or 1: taps see the nearest x2 upsampled input (gan.py:319) */
} m355_conv_desc;

int m355_conv2d_out_hw(const m355_conv_desc *d, int *Ho, int *Wo);

/* Everything a caller has to know about a layer BEFORE it launches -- output extent, buffer sizes, and which of the optional forms
 * (bit masks, fused activation backward, fused statistics, split-K forward, workspace / deterministic weight gradient) this build
 * of the library runs the shape in -- in ONE call (round 5; the individual queries below answer the same questions and remain).  The
 * selection rules live in the library only: a binding asks for the plan once per descriptor and follows it. */
typedef struct m355_conv_plan {
    int Ho, Wo;                   /* m355_conv2d_out_hw */
    int dy_channels;              /* m355_conv2d_dy_channels(Cout) */
    int act_bytes;                /* m355_act_bytes */
    size_t w_fwd_elems, w_dgrad_elems;   /* m355_conv2d_weight_elems(d, 0 / 1), 2-byte elements */
    int fwd_bits_ok, dgrad_bits_ok;      /* m355_conv2d_maskbits_ok(d, 0 / 1) */
    int dgrad_mask_ok;            /* m355_conv2d_dgrad_mask_ok */
    int fwd_stats_rows;           /* m355_conv2d_fwd_stats_rows (0: no fused statistics) */
    int fwd_ws_stats_rows;        /* m355_conv2d_fwd_ws_stats_rows */
    int wgrad_fuses_dbias;        /* m355_conv2d_wgrad_fuses_dbias */
    size_t fwd_ws_bytes;          /* m355_conv2d_fwd_ws_bytes (0: no split-K forward) */
    size_t dgrad_ws_bytes;        /* m355_conv2d_dgrad_ws_bytes */
    size_t wgrad_ws_bytes;        /* m355_conv2d_wgrad_ws_bytes (0: use m355_conv2d_wgrad / _acc / _det) */
    size_t wgrad_det_ws_bytes;    /* m355_conv2d_wgrad_det_ws_bytes */
    double exec_ratio;            /* m355_conv2d_exec_ratio */
    int w_dgrad_row_elems;        /* row length (elements) of the stride-1 dgrad view [rows][taps * dy_channels, padded]: what a kernel that
                                   * reads that view itself (m355_cproj_bwd_conv5's Kp) must be given; 0 for stride-2 / sub-pixel views */
    int wgrad_ws_ordered;         /* 1: m355_conv2d_wgrad_ws adds per-workgroup partial rows in order -- the same bits on every run, in every
                                   * mode (a binding takes it in deterministic mode too); 0: no workspace form, or zeroed cells + atomics */
} m355_conv_plan;
int m355_conv2d_plan(const m355_conv_desc *d, m355_conv_plan *plan);
/*      adjoint of the nearest x2 upsample (gan.py:319) on NHWC bf16: g[N,2H,2W,C] -> dx[N,H,W,C] (2x2 block sums) */
int m355_fold2x2(const void *g, void *dx, int N, int H, int W, int C, void *stream);
int m355_conv2d_dy_channels(int cout);
/*      executed / algorithmic multiply-accumulates of the layer (forward, dgrad, workspace / deterministic wgrad): 4/9 where an
 *      upsample + 3x3 layer (models/gan.py:319,386-404 in front of ResBlockUp.conv1 :294,309) runs in the SUB-PIXEL form -- four 2x2
 *      class convs of the stored tensor with pre-summed weights instead of nine taps on the upsampled one -- else 1 */
double m355_conv2d_exec_ratio(const m355_conv_desc *d);
/*      elements (bf16) of the weight views: which 0 forward [ceil64(Cout)][ceil32(kh*kw*Cin)], 1 dgrad */
size_t m355_conv2d_weight_elems(const m355_conv_desc *d, int which);
/*      w_oihw is the fp32 parameter [Cout][cin_w][kh][kw]; channels cin_w..Cin-1 of the views are zero. */
/*      sigma (nullable, 1 float on the device): the views hold w / sigma[0] -- the spectral-norm division
 *      (torch.nn.utils.spectral_norm: weight = weight_orig / sigma) folded into the bf16 conversion. */
int m355_conv2d_weight_prep(const m355_conv_desc *d, const float *w_oihw, int cin_w, const float *sigma, void *w_fwd,
                            void *w_dgrad, void *stream);

/* ... and the views of ALL layers of a network in one launch (the per-layer calls were 87 launches per GAN cycle):
 * fill one host entry per layer (m355_weight_prep_entry_bytes() bytes each; same arguments as above, DEVICE pointers;
 * returns the layer's largest view in elements or -1), copy the table to the device once, run it every step. */
size_t m355_weight_prep_entry_bytes(void);
long long m355_weight_prep_fill_entry(const m355_conv_desc *d, const float *w_oihw, int cin_w, const float *sigma,
                                      void *w_fwd, void *w_dgrad, void *entry_host);
int m355_weight_prep_batched(const void *table_dev, int n_layers, long long max_elems, void *stream);
/* The same with the "regular" layers (every view's rows and channels multiples of 32, no K padding, <= 16 taps) on a kernel that
 * transposes 32 x 32 (Cout, Cin) tiles through LDS instead of gathering: m355_weight_prep_entry_tiles(entry) = the tiles that
 * layer needs there (0: it stays on the gather kernel); max_tiles9 / max_tiles16 = the largest such value among the layers with
 * kh*kw <= 9 / 10..16 taps (two instantiations: LDS per tile), max_elems = the largest view of the layers that stay (0: none). */
int m355_weight_prep_entry_tiles(const void *entry_host);
int m355_weight_prep_batched_tiled(const void *table_dev, int n_layers, long long max_elems, int max_tiles9, int max_tiles16,
                                   void *stream);
/*      y: bf16 NHWC [N,Ho,Wo,Cout] or (y_f32_nchw) fp32 [N,Cout,Ho,Wo]; epilogue: + bias[Cout] (nullable),
 *      LeakyReLU(lrelu_slope) (1.0 = identity). */
int m355_conv2d_fwd(const m355_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y,
                    int y_f32_nchw, float lrelu_slope, void *stream);
/*      dy[N,Ho,Wo,dy_channels(Cout)] bf16 -> dx[N,H,W,Cin] bf16 (autograd of F.conv2d w.r.t. its input, through the
 *      pad / upsample).  ws >= m355_conv2d_dgrad_ws_bytes(d). */
size_t m355_conv2d_dgrad_ws_bytes(const m355_conv_desc *d);
/*      mask_x (nullable; same shape as dx, i.e. this conv's input x): dx *= (mask_x > 0 ? 1 : mask_slope) in the
 *      epilogue -- the backward of the LeakyReLU that produced x (fused conv+LeakyReLU layer below), so that layer's
 *      gradient arrives already masked.  Only with the direct form (no upsample, zero/circular W pad). */
int m355_conv2d_dgrad(const m355_conv_desc *d, const void *dy, const void *w_dgrad, void *dx, void *ws,
                      const void *mask_x, float mask_slope, void *stream);

/*      the same for a layer whose input channels beyond the first `lead` are constants of the model (TextureDiscriminator.conv1,
 *      models/gan.py:204-213: 4 image channels + the 4 positional planes of gan.py:9-20): only dx[..., 0 .. lead-1] is specified (the
 *      other channels hold zeros or the true gradient, whichever the dispatched kernel produces); no activation mask */
int m355_conv2d_dgrad_lead(const m355_conv_desc *d, const void *dy, const void *w_dgrad, void *dx, void *ws, int lead, void *stream);

/* Bit-packed activation masks between a conv+LeakyReLU forward and its (only) consumer's dgrad: 1 bit per element
 * instead of re-reading the 2-byte activation (D.conv2's dgrad read 1 GB just to test signs).  Layout
 * [N,H,W,C/64,2] uint32 in the register order of the epilogues (opaque: produce with _fwd_bits, consume with
 * _dgrad_bits).  m355_conv2d_maskbits_ok(d, 0): this layer's forward can write them; (d, 1): this layer's dgrad can
 * read them for its input.  Mirrors the autograd pair F.conv2d -> F.leaky_relu of models/gan.py:92-94,210-213. */
int m355_conv2d_maskbits_ok(const m355_conv_desc *d, int role);
/*      1 when m355_conv2d_dgrad accepts mask_x for this layer (the direct forms, and the replicate-padded 5x5 heads) */
int m355_conv2d_dgrad_mask_ok(const m355_conv_desc *d);
int m355_conv2d_fwd_bits(const m355_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y,
                         float lrelu_slope, void *mask_bits_out, void *stream);
/*      forward of a conv whose output feeds batch-norm statistics (the generator's ResBlockUp convs in front of the
 *      ConditionalBatchNorm2d layers, /root/reference/code/models/gan.py:264-312): besides y, every persistent workgroup writes
 *      the sums of its fp32 results and of their squares, part[rows][2][Cout] fp32 with rows =
 *      m355_conv2d_fwd_stats_rows(d) -- the layout m355_bn_finalize reduces -- so the statistics cost no pass over y.
 *      rows == 0: this shape has no fused statistics (run m355_bn_stats_partial on y).  No activation epilogue. */
int m355_conv2d_fwd_stats_rows(const m355_conv_desc *d);
/*      SMALL layers (fewer 128 x 128 output tiles than CUs: the generator's 8x4 / 16x8 stages, gan.py:294-302 at :386-391) as
 *      split-K: m355_conv2d_fwd_ws_bytes(d) > 0 -> the K slices' fp32 accumulators go to `ws`, one finishing pass adds them in
 *      slice order, applies bias / LeakyReLU, stores bf16 NHWC and (stats_part != NULL) writes the batch-norm partial sums
 *      part[m355_conv2d_fwd_ws_stats_rows(d)][2][Cout] -- the pass REPLACES m355_bn_stats_partial on these layers, which is
 *      what makes the split pay (36 + 9 -> 33 us on the 512 -> 512 convs at 8x4); the Python side takes this path only for convs
 *      that feed a batch norm. */
size_t m355_conv2d_fwd_ws_bytes(const m355_conv_desc *d);
int m355_conv2d_fwd_ws_stats_rows(const m355_conv_desc *d);
int m355_conv2d_fwd_ws(const m355_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y, float lrelu_slope,
                       void *ws, float *stats_part, void *stream);
int m355_conv2d_fwd_stats(const m355_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y, float *part,
                          void *stream);
int m355_conv2d_dgrad_bits(const m355_conv_desc *d, const void *dy, const void *w_dgrad, void *dx, void *ws,
                           const void *mask_bits, float mask_slope, void *stream);
/*      x[N,H,W,Cin], dy[N,Ho,Wo,dy_channels(Cout)] -> dw fp32 [Cout][kh][kw][Cin] (overwritten; split-K partial tiles are
 *      combined with fp32 atomics, so the last bits depend on arrival order). */
/*      dbias (nullable, [Cout] fp32): column sums of dy = the bias gradient, accumulated by the workgroups that stage
 *      the dy tiles anyway; only where m355_conv2d_wgrad_fuses_dbias(d) != 0. */
int m355_conv2d_wgrad_fuses_dbias(const m355_conv_desc *d);
int m355_conv2d_wgrad(const m355_conv_desc *d, const void *x, const void *dy, float *dw, float *dbias, void *stream);
/*      the same with dw / dbias += (no zero fill: the caller zeroed them, e.g. one memset for all layers of a backward pass) */
int m355_conv2d_wgrad_acc(const m355_conv_desc *d, const void *x, const void *dy, float *dw, float *dbias, void *stream);
/*      DETERMINISTIC form of the same: every wgrad kernel splits the pixel axis over workgroups and adds the partial tiles with
 *      atomics -- fp32 atomics give run-to-run differences in the last bits (the reference's CPU path is bit-deterministic,
 *      SURVEY 8c; code/main.py:691-723).  Here the SAME kernels accumulate the partials as 64-bit fixed-point integers
 *      (three per cell, on the grids 1, 2^-50 and 2^-100: integer addition is associative, so the order the workgroups finish
 *      in cannot matter, and the triple holds the exact sum of the partial tiles whatever their magnitude; a partial touches one
 *      of the three in the common case -- csrc/conv_dma.h wg_accum) in `ws`
 *      (m355_conv2d_wgrad_det_ws_bytes(d) bytes, zeroed by the call) and one pass converts to fp32 with one rounding.  dw and
 *      dbias are OVERWRITTEN.  A non-finite partial (or one beyond 1e8) poisons the result with NaN, nothing is hidden. */
/*      The 8-input-channel layer (D.conv1, gan.py:163) with a workspace: every workgroup stores its partial tile and a second
 *      launch adds them in workgroup order -- no atomics, deterministic in every mode; dw / dbias are OVERWRITTEN.
 *      _ws_bytes == 0: the layer has no such form (the <= 8-output-channel heads measured slower with it). */
size_t m355_conv2d_wgrad_ws_bytes(const m355_conv_desc *d);
int m355_conv2d_wgrad_ws(const m355_conv_desc *d, const void *x, const void *dy, float *dw, float *dbias, void *ws, void *stream);
size_t m355_conv2d_wgrad_det_ws_bytes(const m355_conv_desc *d);
int m355_conv2d_wgrad_det(const m355_conv_desc *d, const void *x, const void *dy, void *ws, float *dw, float *dbias, void *stream);

/* ---- G3 / G-bwd  normalisation + conditional affine + LeakyReLU on NHWC bf16 (models/gan.py:264-286, 306-312).
 *      All reductions are two-stage and deterministic; ws >= m355_chan_reduce_ws_bytes(pixels per group, groups,
 *      values per channel, C).  C must be a multiple of 8 with C/8 dividing 256. */
size_t m355_chan_reduce_ws_bytes(size_t pixels_per_group, int groups, int nvals, int C);
/*      x[P][C] -> sums[2][C] = (sum, sum of squares): the batch statistics of BatchNorm2d / SynchronizedBatchNorm2d */
int m355_bn_stats(const void *x, float *sums, void *ws, size_t P, int C, void *stream);
/*      x[P][C] -> sums[C] (bias gradient) */
int m355_chan_sum(const void *x, float *sums, void *ws, size_t P, int C, void *stream);
/*      y = LeakyReLU_slope(x * a[n,c] + b[n,c]) [+ res];  x,y,res [N][HW][C];  a = rstd*(1+gamma), b = beta - mean*a;
 *      res (nullable): the residual branch of ResBlockUp (gan.py:312) added in the same pass; res_w > 0: res is stored
 *      at half resolution [N][H/2][res_w/2][C] and read through the nearest x2 upsample (res_w = full-res width);
 *      out_slope != 1: y = LeakyReLU_out_slope(that) -- the activation Generator.forward applies to a block's output in
 *      front of conv_final / conv_mesh (gan.py:406,410), whose backward the consuming conv's dgrad applies (mask_x) */
int m355_affine_act_fwd(const void *x, const float *a, const float *b, const void *res, int res_w, void *y, int N, int HW,
                        int C, float slope, float out_slope, void *stream);
/*      dz = dy * LeakyReLU'(x*a+b);  sums[N][2][C] = (sum_hw dz, sum_hw dz*x) */
int m355_affine_act_bwd_reduce(const void *dy, const void *x, const float *a, const float *b, float *sums, void *ws,
                               int N, int HW, int C, float slope, void *stream);
/*      dx = dz * A[n,c] + x * Bc[c] + Cc[c]   (the batch-norm backward collapses into these three coefficients) */
int m355_affine_act_bwd_apply(const void *dy, const void *x, const float *a, const float *b, const float *A,
                              const float *Bc, const float *Cc, void *dx, int N, int HW, int C, float slope,
                              void *stream);
/*      backward of a conv epilogue LeakyReLU (discriminators, gan.py:92-94,210-213): g = dy * (y > 0 ? 1 : slope),
 *      dbias[C] = sum over pixels of g */
int m355_lrelu_bwd(const void *dy, const void *y, void *g, float *dbias, void *ws, size_t P, int C, float slope,
                   void *stream);

/*      first stages only (the second stage is fused into m355_bn_finalize / m355_bn_bwd_finalize):
 *      part[m355_chan_reduce_nblk(P, C)][2][C] resp. part[N][m355_chan_reduce_nblk(HW, C)][2][C] */
int m355_chan_reduce_nblk(size_t pixels_per_group, int C);
int m355_bn_stats_partial(const void *x, float *part, size_t P, int C, void *stream);
/*      SynchronizedBatchNorm2d (code/sync_batchnorm/batchnorm.py:110-131 sends (sum, ssum, size) to the master): the message of
 *      the one all-reduce that replaces it, vec[2C + 1] = [ column sums of part[nblk][2][C] | pixel count ] */
int m355_bn_sync_pack(const float *part, int nblk, int C, float count, float *vec, void *stream);
int m355_affine_act_bwd_partial(const void *dy, const void *x, const float *a, const float *b, float *part, int N, int HW,
                                int C, float slope, void *stream);

/* ---- G5  discriminator input: [N,C,H,W] fp32 image (C <= 8) ++ P constant planes pos[P][H][W] (positional encoding,
 *      gan.py:9-20,204-209) -> NHWC bf16 with 8 channels (zero beyond C+P); and the backward w.r.t. the image. */
int m355_pack_nhwc8(const float *x_nchw, const float *pos, void *out_nhwc8, int N, int C, int P, int H, int W, void *stream);
int m355_unpack_nhwc8(const void *g_nhwc8, float *dx_nchw, int N, int C, int H, int W, void *stream);

/* ---- G9  spectral normalisation (torch.nn.utils.spectral_norm, one power iteration per training forward:
 *      v = normalize(W^T u), u = normalize(W v), sigma = u.(W v); gan.py:57-65,163-177,294-302) for ALL layers of
 *      a network in three launches (two in eval mode).  `table` is a device-resident array of L entries; scratch =
 *      m355_sn_scratch_words(L, max_rows, max_cols) 4-byte words (no initialisation needed): per-workgroup partial sums of the
 *      two norms, added in a fixed order by the next launch -- sigma / u / v have the same bits on every run (no floating-point
 *      atomics).  u, v are updated in place when training, u_snap / v_snap (nullable)
 *      receive the values this forward used (what autograd needs in the backward). */
typedef struct {
    const float *w;          /* weight_orig viewed as [rows][cols] = [Cout][Cin*kh*kw] */
    float *u, *v;            /* [rows], [cols] power-iteration state */
    float *t, *s;            /* scratch [cols], [rows] */
    float *u_snap, *v_snap;  /* [rows], [cols] or NULL */
    int rows, cols;
} m355_sn_layer;
size_t m355_sn_scratch_words(int L, int max_rows, int max_cols);
int m355_sn_power_iter(const m355_sn_layer *table, int L, int max_rows, int max_cols, float *scratch, float *sigma,
                       int training, float eps, void *stream);
/*      weight gradient epilogue: g = dL/dW_sn as the wgrad kernel leaves it, [Cout][kh][kw][CinP] ->
 *      dw [Cout][Cin][kh][kw] = g / sigma - (<g, w_orig> / sigma^2) u v^T   (sigma NULL: plain re-layout).
 *      part: scratch of 256 floats. */
int m355_sn_wgrad_finish(const float *g_khwc, const float *w_orig, const float *u, const float *v, const float *sigma,
                         float *part, float *dw, int Cout, int Cin, int CinP, int kh, int kw, void *stream);
/*      the same for up to M355_SNFIN_MAX layers of one backward pass in two launches (the per-layer form costs two launches of
 *      ~7 us per nn.Conv2d: 76 per training cycle).  entries_host: HOST array, copied into the kernel arguments (device
 *      pointers inside); part: Cout floats of scratch per entry with sigma; (CinP + 1) * kh * kw <= 12832. */
#define M355_SNFIN_MAX 24
typedef struct {
    const float *g_khwc, *w_orig, *u, *v, *sigma;
    float *part, *dw;
    int Cout, Cin, CinP, kh, kw;
    int pad_;
} m355_snfin_entry;
int m355_sn_wgrad_finish_batched(const m355_snfin_entry *entries_host, int L, void *stream);

/* ---- G3  conditional batch norm coefficient algebra (gan.py:264-286) in one launch each way.
 *      forward:  part[nblk][2][C] (sum, sum of squares over `count` pixels) -> mean[C], rstd[C] (biased variance,
 *      eps inside the sqrt), running stats (momentum, unbiased variance; nullable),
 *      a[n,c] = rstd (1 + gamma[n,c]), b[n,c] = beta[n,c] - mean a[n,c];  gamma/beta rows are gstride floats apart.
 *      backward: part[N][nblk][2][C] (sum dz, sum dz x) -> dgamma, dbeta [N][C], A[N][C], Bc[C], Cc[C] with
 *      dx = dz A + x Bc + Cc. */
int m355_bn_finalize(const float *part, int nblk, float count, const float *count_dev, const float *gamma, const float *beta,
                     int gstride, int N, int C, float eps, float momentum, float *running_mean, float *running_var, float *mean,
                     float *rstd, float *a, float *b, void *stream);
int m355_bn_bwd_finalize(const float *part, int nblk, float count, const float *gamma, int gstride, int N, int C,
                         const float *mean, const float *rstd, int batch_stats, float *dgamma, float *dbeta, float *A,
                         float *Bc, float *Cc, float *m_out, void *stream);
/*      SyncBN (one process per GPU): m_out != NULL makes m355_bn_bwd_finalize emit the local moment sums
 *      m[2][C] = (sum_n (1+gamma) s1, sum_n (1+gamma) dgamma) instead of Bc / Cc; after their all-reduce (RCCL)
 *      this turns them into Bc, Cc with the global pixel count.  The forward side needs no extra entry point:
 *      all-reduce the summed partials and call m355_bn_finalize with nblk = 1.
 *      count_dev (both entry points): when non-NULL the pixel count is read from this DEVICE scalar instead of `count`
 *      -- the all-reduced [sum | sumsq | count] vector's last element, so ranks with ragged shards need no host sync
 *      (code/sync_batchnorm/batchnorm.py:110-131 sums the real sizes the same way). */
int m355_bn_bwd_coeffs(const float *m, float count, const float *count_dev, const float *mean, const float *rstd, int C,
                       float *Bc, float *Cc, void *stream);

/* ---- input / output glue of the GAN stacks (csrc/gan_io.hip); all tensors fp32 NCHW unless stated
 *      ModelWrapper.forward, code/main.py:493,503-507: X = cat(fake * alpha, alpha) [N,4,H,W]; with `real` non-NULL the
 *      batch concatenation with cat(real, alpha) in the same pass: X [2N,4,H,W].  _bwd: dfake = dX[:N,:3] * alpha. */
int m355_mask_cat_fwd(const float *fake, const float *real, const float *alpha, float *X, int N, int H, int W, void *stream);
int m355_mask_cat_bwd(const float *dX, const float *alpha, float *dfake, int N, int H, int W, void *stream);
/*      TextureDiscriminator / MeshDiscriminator.forward up to conv1, code/models/gan.py:79-99,192-211:
 *      out[m,y,x,:] (NHWC bf16, CP = 8 | 16 channels, zero filled) = avg_pool2d(x, f)[m,:,y,x] ++ extra[m,:,y,x] ++ pos[:,y,x];
 *      mask (nullable) [M,H/(f g),W/(f g)] = avg_pool2d(avg_pool2d(x, f)[:, mask_chan], g).  m355_pool_pack_ok: shapes
 *      the kernel takes (C <= 4, f in 1,2,4,8,16, pooled H and W multiples of 16, g in 4,8,16). */
int m355_pool_pack_ok(int C, int H, int W, int f, int E, int P, int g);
int m355_pool_pack_fwd(const float *x, int M, int C, int H, int W, int f, const float *extra, int E, const float *pos, int P,
                       void *out, int CP, float *mask, int mask_chan, int g, void *stream);
/*      adjoint with respect to x for up to three packed tensors of the same x (dh_k [M,H/f_k,W/f_k,cp_k] bf16, NULL ends
 *      the list): dx[m,c,y,x] = sum_k dh_k[m,y/f_k,x/f_k,c] / f_k^2;  m355_unpack_range: channels c0..c0+E-1 of an NHWC
 *      bf16 tensor -> [M,E,HW] fp32 (the gradient of `extra`) */
int m355_pool_unpack_bwd(const void *dh0, int f0, int cp0, const void *dh1, int f1, int cp1, const void *dh2, int f2, int cp2,
                         float *dx, int M, int C, int H, int W, void *stream);
int m355_unpack_range(const void *g, float *out, int M, int HW, int CP, int c0, int E, void *stream);
/*      main.py:493,503-507 and gan.py:79-99,192-211 in ONE pass: m355_pool_pack_fwd on x = m355_mask_cat_fwd(fake, real, alpha)
 *      without x ever being written (C = 4, mask_chan = 3; samples [0,Nf) are cat(fake * alpha, alpha), samples [Nf,M) -- M is Nf
 *      or 2 Nf -- are cat(real, alpha)); _bwd: dfake [N,3,H,W] = m355_mask_cat_bwd(m355_pool_unpack_bwd(...)).  Same bits as the
 *      two-call forms (the product is rounded to fp32 before the pooling sum). */
int m355_pool_pack_parts_fwd(const float *fake, const float *real, const float *alpha, int Nf, int M, int H, int W, int f,
                             const float *extra, int E, const float *pos, int P, void *out, int CP, float *mask, int g, void *stream);
int m355_pool_unpack_parts_bwd(const void *dh0, int f0, int cp0, const void *dh1, int f1, int cp1, const void *dh2, int f2, int cp2,
                               const float *alpha, float *dfake, int N, int H, int W, void *stream);
/*      Generator.forward after conv_final / conv_mesh, code/models/gan.py:407-419: flags M355_HT_TANH (tanh_),
 *      M355_HT_POLES (adjust_poles, rendering/utils.py:21-26), M355_HT_SYMM (symmetrize_texture, rendering/utils.py:15-18:
 *      out width 2W).  y [N,C<=3,H,W] -> out.  _bwd: dout, out -> g [N,H,W,8] bf16 (the layout m355_conv2d_dgrad / _wgrad
 *      expect of a 3-channel head's dy) and dbias[C] (overwritten; ws = M355_HEAD_TAIL_WS_FLOATS floats of scratch: the
 *      workgroups' partial sums, added in workgroup order -- same bits on every run). */
#define M355_HT_TANH 1
#define M355_HT_POLES 2
#define M355_HT_SYMM 4
#define M355_HEAD_TAIL_WS_FLOATS 8192
int m355_head_tail_fwd(const float *y, float *out, int N, int C, int H, int W, int flags, void *stream);
int m355_head_tail_bwd(const float *dout, const float *out, void *g_nhwc8, float *dbias, float *ws, int N, int C, int H, int W,
                       int flags, void *stream);
/*      GANLoss('hinge') over the K <= 3 discriminator outputs, code/utils/losses.py:49-120, with divide_pred
 *      (code/main.py:414-422) done by index: p[k] / m[k] (HOST arrays of K device pointers; m or m[k] NULL = unmasked)
 *      are [B,hw[k]]; samples [0,split) are the "fake" half (slot 0), [split,B) the "real" half (slot 1).
 *      mode 1 (discriminator): loss2[0] = loss vs target False on the fake half, loss2[1] = loss vs target True on the real
 *      half; mode 0 (generator, split = B): loss2[0] = -mean.  w (nullable): per-discriminator weights (main.py:486-489).
 *      msum [2,K,B]: [0] per-sample mask sums kept for the backward, [1] scratch (the per-(discriminator, sample) loss terms,
 *      which a second launch adds in (k, b) order: same bits on every run); _bwd: gl2[2] incoming gradients -> dp[k] [B,hw[k]]. */
int m355_hinge_fwd(int K, const float *const *p, const float *const *m, const int *hw, const float *w, int B, int split, int mode,
                   float *loss2, float *msum, void *stream);
int m355_hinge_bwd(int K, const float *const *p, const float *const *m, const int *hw, const float *w, int B, int split, int mode,
                   const float *gl2, const float *msum, float *const *dp, void *stream);

/* ---- the DIB-R rasteriser and fragment shader behind Renderer.forward (code/rendering/renderer.py:39-77,
 *      code/rendering/fragment_shader.py:6-37); csrc/dibr_raster.hip.
 *      m355_dibr_rasterize_fwd replaces kaolin.graphics.dib_renderer.rasterizer.linear_rasterizer(height, width,
 *      points3d_bxfx9, points2d_bxfx6, normalz_bxfx1, vertex_attr_bxfx3d) as called at renderer.py:62-69 (Kaolin at the
 *      commit of code/rendering/monkey_patches.py:4 is not available here: published DIB-R algorithm, kaolin's defaults
 *      expand = 0.02, knum = 30, delta = 7000; PARITY UNPINNED, oracle = oracle/raster_ref.py):
 *        imfeat [B,H,W,D]  attributes interpolated with the barycentric weights of the closest front face covering the
 *                          pixel centre (zero elsewhere);   improb [B,H,W]  soft silhouette probability;
 *        imidx [B,H,W] int32 covering face or -1, imwei [B,H,W,3] its weights (kept for the backward);
 *        ws: m355_dibr_ws_bytes(B, F) bytes, written by _fwd and read again by _bwd.
 *      _bwd: d imfeat, d improb -> d points2d [B,F,6], d attr [B,F,3D] (points3d only selects the face: no gradient). */
size_t m355_dibr_ws_bytes(int B, int F);
int m355_dibr_rasterize_fwd(int height, int width, const float *points3d_bxfx9, const float *points2d_bxfx6,
                            const float *normalz_bxfx1, const float *attr_bxfx3d, int B, int F, int D, float expand, int knum,
                            float delta, void *ws, float *imfeat, float *improb, int32_t *imidx, float *imwei, void *stream);
int m355_dibr_rasterize_bwd(int height, int width, const float *points3d_bxfx9, const float *points2d_bxfx6,
                            const float *attr_bxfx3d, int B, int F, int D, int knum, float delta, const void *ws,
                            const float *improb, const int32_t *imidx, const float *imwei, const float *dimfeat,
                            const float *dimprob, float *dpoints2d_bxfx6, float *dattr_bxfx3d, void *stream);
/*      _bwd_det (deterministic mode): the same gradients, but every sum over pixels -- per tile in LDS, across tiles in memory -- is
 *      taken in integers (a contribution as a pair of 64-bit cells, resolution 2^-48; fix_ws = m355_dibr_rasterize_bwd_det_ws_bytes(B, F,
 *      D) bytes, zeroed here) and converted once: bit-identical from run to run.  A non-finite or >= 1e9 contribution turns the
 *      outputs into NaN. */
size_t m355_dibr_rasterize_bwd_det_ws_bytes(int B, int F, int D);
int m355_dibr_rasterize_bwd_det(int height, int width, const float *points3d_bxfx9, const float *points2d_bxfx6,
                                const float *attr_bxfx3d, int B, int F, int D, int knum, float delta, const void *ws,
                                const float *improb, const int32_t *imidx, const float *imwei, const float *dimfeat,
                                const float *dimprob, void *fix_ws, float *dpoints2d_bxfx6, float *dattr_bxfx3d, void *stream);
/*      fragmentshader(imtexcoord, texture, improb, filtering='bilinear', background_image) of fragment_shader.py:22-37:
 *      uvm [B,H,W,3] = (u, v, hard mask) as the rasteriser writes them; color = grid_sample(texture, (uv*2-1)*(1,-1),
 *      bilinear, align_corners=True) * hard, or lerp(background, that, hard).  _bwd: d color -> d uvm, d texture (nullable),
 *      d background (nullable). */
int m355_dibr_shade_fwd(const float *uvm_bxhxwx3, const float *texture_bx3xthxtw, const float *background_bxhxwx3,
                        float *color_bxhxwx3, int B, int H, int W, int TH, int TW, void *stream);
int m355_dibr_shade_bwd(const float *uvm_bxhxwx3, const float *texture_bx3xthxtw, const float *background_bxhxwx3,
                        const float *dcolor_bxhxwx3, float *duvm_bxhxwx3, float *dtexture_bx3xthxtw, float *dbackground_bxhxwx3,
                        int B, int H, int W, int TH, int TW, void *stream);
/*      _bwd_det: the texel gradient (many pixels per texel) summed in integers as above; fix_ws =
 *      m355_dibr_shade_bwd_det_ws_bytes(B, TH, TW) bytes (only read when d texture is requested) */
size_t m355_dibr_shade_bwd_det_ws_bytes(int B, int TH, int TW);
int m355_dibr_shade_bwd_det(const float *uvm_bxhxwx3, const float *texture_bx3xthxtw, const float *background_bxhxwx3,
                            const float *dcolor_bxhxwx3, void *fix_ws, float *duvm_bxhxwx3, float *dtexture_bx3xthxtw,
                            float *dbackground_bxhxwx3, int B, int H, int W, int TH, int TW, void *stream);

/* Projection discriminator (code/models/gan.py:104-116, 216-228): out[n,p] = sum_c feat[n,p,c] * emb[n,c] on the NHWC bf16
 * feature map (C a power of two in 8..2048); backward: dfeat = g * emb (bf16, overwritten), demb[n,c] = sum_p g * feat.
 * mask_slope != 1: feat is a fused conv + LeakyReLU(mask_slope) output and dfeat is multiplied by that activation's derivative
 * (feat > 0 ? 1 : mask_slope), as m355_conv2d_dgrad does with mask_x for the other consumer of the same tensor. */
int m355_cproj_fwd(const void *feat, const float *emb, float *out /*[N,HW]*/, int N, int HW, int C, void *stream);
size_t m355_cproj_bwd_ws_floats(int N, int HW, int C);   /* scratch of _bwd (per-workgroup shares of demb, added in order); 0: none */
int m355_cproj_bwd(const void *feat, const float *emb, const float *g /*[N,HW]*/, void *dfeat, float *demb /*[N,C]*/,
                   float *ws, int N, int HW, int C, float mask_slope, void *stream);
/*      the whole tail of a discriminator in one backward pass (round 5): feat has two consumers, the 5x5 one-channel logit conv and the
 *      projection term; dfeat = mask(feat) * (g[n,p] * emb[n,c] + the logit conv's input gradient), i.e. m355_cproj_bwd + the masked
 *      m355_conv2d_dgrad of that conv + the addition of the two, without the two extra passes over feat.  dy5 [N,H,W] fp32 = the logit
 *      gradient, w_dgrad / Kp = the logit conv's dgrad view and its row length; pad_w_mode 0 or 2. */
int m355_cproj_bwd_conv5_ok(int H, int W, int C);
size_t m355_cproj_bwd_conv5_ws_floats(int N, int H, int W, int C);
int m355_cproj_bwd_conv5(const void *feat, const float *emb, const float *g, const float *dy5, const void *w_dgrad, int Kp,
                         void *dfeat, float *demb, float *ws, int N, int H, int W, int C, float mask_slope, int pad_w_mode,
                         void *stream);

/* ---- mesh-template deformation, face normals, flat (smoothness) loss -- code/main.py:697-699 ----
 * Replaces MeshTemplate.get_vertex_positions / deform / compute_normals (code/rendering/mesh_template.py:106-149) and
 * loss_flat (code/utils/losses.py:5-17); the reference-side binding is in INTEGRATION.md.  All fp32, row-major.
 *   dmap  [B,3,H,W]  displacement map (the generator's mesh head, NCHW)
 *   uv    [S,2]      grid_sample coordinates of the S source vertices in the W-padded map (host: mesh.py, per W)
 *   tgm   [S,3,3]    tangent frames (normal, tangent, bitangent) of the source vertices
 *   base  [V,3]      template vertices;  src [V] int32 source index of vertex v;  xsign [V] x-factor (-1 mirrored, 0 on
 *                    the symmetry plane, +1 otherwise)
 *   symmetric: 1 = circular pad by one column each side (mesh_template.py:166), 0 = one wrapped column (:169)
 *   faces [F,3] int32, ff [F,3] int32 face-face adjacency (kaolin `mesh.ff`)
 * The backward entry points are GATHERS over static CSR tables the host builds once per template (mesh.py; all int32):
 *   tex_ptr [H*W+1], tex_vtx [E], tex_w [E]   per map texel: the (vertex, bilinear weight) pairs whose grid_sample taps land on
 *                                              it (pad columns folded onto their source column), vertices ascending
 *   vtx_ptr [V+1], vtx_fc [3F]                per vertex: 4 * face + corner of its incident face corners, ascending
 *   rev_ptr [F+1], rev_idx [3F]               per face f: the faces g with f in ff[g] (the reverse of ff), ascending
 * Every output element is written by exactly one thread, summing in table order: no atomics, no zero fill, and the same bits
 * on every run (the reference's CPU path is deterministic).  _bwd entry points overwrite their gradient outputs;
 * mesh_flat_fwd's ws is B floats of scratch (per-sample partial sums, added in index order). */
int m355_mesh_vertices_fwd(const float *dmap, const float *uv, const float *tgm, const float *base, const int *src,
                           const float *xsign, float *pos /*[B,V,3]*/, int B, int V, int H, int W, int symmetric, void *stream);
int m355_mesh_vertices_bwd(const float *dpos, const float *tgm, const int *src, const float *xsign, const int *tex_ptr,
                           const int *tex_vtx, const float *tex_w, float *ddmap /*[B,3,H,W]*/, int B, int V, int H, int W,
                           void *stream);
int m355_mesh_normals_fwd(const float *pos, const int *faces, float *normals /*[B,F,3]*/, int B, int V, int F, void *stream);
int m355_mesh_normals_bwd(const float *pos, const int *faces, const float *dnormals, const int *vtx_ptr, const int *vtx_fc,
                          float *dpos, int B, int V, int F, void *stream);
int m355_mesh_flat_fwd(const float *normals, const int *ff, float *loss /*[1]*/, float *ws /*[B]*/, int B, int F, void *stream);
int m355_mesh_flat_bwd(const float *normals, const int *ff, const int *rev_ptr, const int *rev_idx, const float *gloss /*[1]*/,
                       float *dnormals, int B, int F, void *stream);

/* ---- small-message all-reduce over peer-mapped device memory (round 6; csrc/ipc_exchange.hip) -----------------------------------
 * The SyncBN statistics exchange of the data-parallel GAN path -- [sum | sum of squares | count] forward, two moment sums backward,
 * <= 4097 floats, 56 dependent messages per training cycle -- as ONE kernel launch on the compute stream instead of a collective-
 * library call.  Replaces the master / slave pipes of /root/reference/code/sync_batchnorm/batchnorm.py:110-131 and comm.py:18-133
 * (what the default build routes through an RCCL all-reduce).  Protocol: every rank owns one fine-grained device region, exports it
 * (hipIpcGetMemHandle, 64 bytes, through any channel), maps the peers' (m355_ipc_open); a message publishes the local vector with a
 * release-stored sequence flag, acquire-spins on the peers' flags, adds the W vectors in RANK order (same additions in the same order
 * on every rank: bit-equal results, deterministic).  Messages are matched per CHANNEL (= call site: one SyncBN layer's forward or backward;
 * different call sites may execute in different orders on different ranks when branches run on two streams); the per-channel sequence
 * numbers live in the region, so a launch replays from a hipGraph.  Every wait is bounded (timeout_ms, default 2000): a missing peer raises a bit in *status and the kernel leaves. */
#define M355_IPC_MAX_RANKS 16
#define M355_IPC_CHANNELS 64
size_t m355_ipc_region_bytes(void);
int m355_ipc_max_floats(void);
int m355_ipc_channels(void);
int m355_ipc_alloc(void **region, void *handle64);
int m355_ipc_open(const void *handle64, void **region);
int m355_ipc_close(void *region);
int m355_ipc_free(void *region);
int m355_ipc_allreduce(float *inout, int n, void *const *regions /*[world], host array of device pointers*/, int rank, int world,
                       int channel, unsigned *status /*device word*/, int timeout_ms, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* M355_H */
