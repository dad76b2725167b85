/*
 * lighthand_hip.h -- C ABI of liblighthand_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the heatmap-regression hot path of leejeongho3214/LightHand.
 * The reference has no FFI of its own: all arithmetic on this
 * path is done by PyTorch/cuDNN underneath plain Python callables.  Each entry point
 * below therefore cites the reference call site whose device work it replaces
 * (paths relative to the reference root); the Python mirror of those callables lives
 * in lighthand_amd/ and reaches this library through ctypes (lighthand_amd/_lib.py).
 *
 * Conventions
 *  - the caller owns every buffer (device pointers unless stated); the library never
 *    allocates, frees or retains device memory;
 *  - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*);
 *    no entry point synchronises, so every call is legal inside hipGraph capture;
 *  - return 0 on success, a negative lh_status otherwise; lh_last_error() gives text;
 *  - activations are NHWC, `dtype` selects their element type (weights packs use the
 *    same type; BatchNorm vectors, loss, gradients of weights and Adam state are fp32);
 *  - weight tensors at the boundary keep the reference/PyTorch layouts (OIHW for
 *    Conv2d, [C_in, C_out, kH, kW] for ConvTranspose2d); the pack functions build the
 *    device-side K-major images the MFMA kernels consume.
 */
#ifndef LIGHTHAND_HIP_H
#define LIGHTHAND_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { LH_F32 = 0, LH_BF16 = 1, LH_F16 = 2 } lh_dtype;

typedef enum {
    LH_OK = 0,
    LH_ERR_ARG = -1,      /* bad shape / dtype / alignment */
    LH_ERR_HIP = -2,      /* a HIP runtime call failed     */
    LH_ERR_UNSUPPORTED = -3
} lh_status;

int lh_version(void);
const char* lh_last_error(void);
/* number of bytes of one activation element for `dtype` */
int lh_dtype_size(int dtype);

/* ------------------------------------------------------------------ layout transforms
 * images.cuda() -> model(images): src/utils/method.py:165-167.  NCHW fp32 image batch ->
 * zero-padded NHWC4 image in the run dtype ([N][H+2*pad][Wp][4], channel 3 = 0), the
 * input format of the stem convolution kernel. */
int lh_image_to_nhwc4(const float* nchw, void* out, int n, int h, int w, int pad, int wp,
                      int dtype, void* stream);
/* Fused input pipeline (the CPU DataLoader work of src/tools/dataset.py:128-159 moved onto the device): uint8 HWC
 * [n][hs][ws][3] -> /255 -> bilinear resize to h x w (half-pixel centres, no antialias) -> (x - mean)/std (HOST
 * float[3] arrays) -> zero-padded NHWC4 in the run dtype, i.e. the stem's input format. */
int lh_image_u8_to_nhwc4(const unsigned char* hwc, void* out, int n, int hs, int ws, int h, int w, int pad, int wp,
                         const float* mean3, const float* std3, int dtype, void* stream);
/* The same with torchvision's ColorJitter(brightness, contrast, saturation, hue) between Resize and Normalize -- the
 * training transform of src/tools/dataset.py:134-146.  The random draw (ColorJitter.get_params) stays with the caller:
 * factors_dev fp32 [n][4] = per-image brightness / contrast / saturation / hue factors, order_dev int32 [n][4] = the op
 * order (op ids 0..3, a negative id skips: the reference jitters only a fraction of its samples).  workspace (device,
 * lh_image_jitter_workspace_bytes(n)) holds the fp64 strip sums of the per-image grey mean the contrast op needs. */
size_t lh_image_jitter_workspace_bytes(int n);
int lh_image_u8_jitter_to_nhwc4(const unsigned char* hwc, void* out, int n, int hs, int ws, int h, int w, int pad, int wp,
                                const float* mean3, const float* std3, const float* factors_dev, const int* order_dev,
                                void* workspace, int dtype, void* stream);
/* NHWC (run dtype) -> NCHW fp32 heatmaps (what model(images) returns, pose_resnet.py:246)
 * and the inverse for the incoming gradient. c_stride = channel stride of the NHWC side. */
int lh_nhwc_to_nchw_f32(const void* nhwc, float* nchw, int n, int h, int w, int c, int c_stride,
                        int dtype, void* stream);
int lh_nchw_f32_to_nhwc(const float* nchw, void* nhwc, int n, int h, int w, int c, int c_stride,
                        int dtype, void* stream);

/* ------------------------------------------------------------------ weight packs
 * Build the K-major device image [Cout_pad128][ntaps][Kpad] of one tap subset of a
 * weight tensor.  Logical weight element (o, i, r, s) is read from
 * w[o*so + i*si + r*sr + s*ss] (fp32, reference layout via strides), taps are listed as
 * (r, s) pairs; rows o >= n_out and columns i >= n_in are zero.  Returns packed bytes
 * through *bytes when out == NULL (size query). */
int lh_pack_weight(const float* w, void* out, size_t* bytes, int n_out, int n_in,
                   long so, long si, long sr, long ss, int ntaps, const int* taps_rs,
                   int dtype, void* stream);

/* The same for every pack of a model in one launch: `items` is a DEVICE array of descriptors. */
typedef struct {
    const float* w;
    void* out;
    int n_out, n_in, ntaps, pad_;
    long so, si, sr, ss;
    signed char r[64];
    signed char s[64];
} lh_pack_item;
/* chunk tables (device): chunk j rebuilds elements [chunk_start[j], +lh_pack_chunk_elems()) of pack chunk_item[j] */
int lh_pack_chunk_elems(void);
int lh_pack_weights_multi(const lh_pack_item* items_dev, const int* chunk_item_dev, const long* chunk_start_dev,
                          int n_chunks, int dtype, void* stream);

/* Regular weight tensors w[d0][d1][kH*kW] (Conv2d: d0 = C_out, d1 = C_in; ConvTranspose2d: d0 = C_in, d1 = C_out):
 * every pack of every such tensor in one launch, through 32 x 32 x taps LDS tiles (coalesced reads, 64-byte writes).
 * A pack is [rows pad128][ntaps][kpad] with rows = d1 (row_is_d1) or d0; taps[] index kH*kW positions (r*kW + s).
 * Padding of the pack images must be zero beforehand (it is never written).  chunk tables (device int32): chunk j
 * handles tile (t0[j], t1[j]) (units of 32) of tensor conv[j]. */
typedef struct {
    void* out;
    int row_is_d1, ntaps, kpad, pad_;
    int taps[16];
} lh_pack_out;
typedef struct {
    const float* w;
    int d0, d1, rs, npacks;
    lh_pack_out packs[5];
} lh_pack_conv;
int lh_pack_weights_tiled(const lh_pack_conv* convs_dev, const int* chunk_conv_dev, const int* chunk_t0_dev,
                          const int* chunk_t1_dev, int n_chunks, int max_rs, int dtype, void* stream);

/* ------------------------------------------------------------------ convolutions
 * nn.Conv2d(..., bias=False) + the head conv with bias: pose_resnet.py:23-26,66-72,152,
 * 169-175,181-182; pose_hrnet.py:22-25,65-71,145-149,200-204,218-222,230-234,282-286,
 * 323-329,344-348,363-365,378-381.  nn.ConvTranspose2d(k=4,s=2,p=1): pose_resnet.py:219-227.
 *
 * One descriptor covers forward, data-gradient and transposed forms: an output pixel
 * grid (ho, wo) per image, a list of taps (dh, dw) that select the input pixel
 * (a*sh + dh, b*sw + dw) and the weight-pack slice, and an affine placement of the
 * output pixel (a*osh + ooh, b*osw + oow) inside an [OH][OW] image (used by the four
 * sub-pixel phases of stride-2 transposed convolutions). */
typedef struct {
    int n, hi, wi;            /* input batch / image size in pixels                  */
    int in_pix_stride;        /* elements between adjacent input pixels              */
    int k_run;                /* valid contiguous elements per tap                   */
    int ho, wo;               /* output pixel grid of this launch                    */
    int sh, sw;               /* input sampling stride                               */
    int cout;                 /* valid output channels                               */
    int OH, OW;               /* full output image size                              */
    int osh, osw, ooh, oow;   /* output pixel = (a*osh + ooh, b*osw + oow)           */
    int out_pix_stride;       /* elements between adjacent output pixels             */
    int ntaps;
    int relu;                 /* apply max(.,0) before the store                     */
    signed char dh[64];
    signed char dw[64];
    /* Kernel configuration chosen by the caller (all zero = the library's static default).  cfg[0..4] = convolution
     * kernel: tile output channels, tile pixels, ring depth, K bytes per ring stage, reserved (0); ring depth 1 selects
     * the persistent pointwise kernel (1x1, dense output, K <= 512, 16-bit types): channels of the resident weight
     * panel, pixels a wave takes per step, 1, padded K of the panel
     * -- one of lh_igemm_candidates().  cfg[5..7] = weight-gradient kernel (lh_wgrad*): tile output channels, tile
     * input channels, pixel splits -- one of lh_wgrad_candidates().  Results do not depend on cfg[0..4]; the
     * weight gradient's fp32 summation order depends on the split count (deterministic for a given cfg). */
    int cfg[8];
} lh_igemm_desc;

/* out[pixel][co] = relu?( (sum_taps sum_k in[pix(tap)][k] * wpack[co][tap][k] + bias[co]) * scale[co] + shift[co]
 *                          + addend[pixel][co] )      (bias / scale+shift / addend optional; relu from the descriptor).
 * scale/shift fold an eval-mode BatchNorm (and with addend + relu a whole residual-unit tail) into the epilogue.
 * addend_mask (optional, dense outputs only): the relu_mask bits lh_fuse_fwd stored for an activation; element e of a
 * 16-byte chunk of the addend is added only where its bit is set.  Lets a data-gradient launch add the identity-shortcut
 * gradient of a residual tail, dout * (out > 0), straight from dout (loss.backward() through `out += residual; relu`,
 * pose_resnet.py:96-97) instead of reading a copy lh_fuse_bwd would have to write first.
 * stats (optional): fp32 [gridM][2][cout] per-block column sums / sums of squares of the stored values,
 * consumed by lh_bn_finalize.  addend may alias out. */
int lh_igemm(const lh_igemm_desc* d, const void* in, const void* wpack, void* out,
             const void* addend, const void* addend_mask, const float* bias, const float* scale, const float* shift,
             float* stats, int dtype, void* stream);
/* A data-gradient launch whose OUTPUT is the gradient of an activation a = relu(BN(x)) (x = the forward convolution's raw
 * output, BN in training mode: pose_resnet.py:60-66 and what loss.backward() does with them) can do the first half of that
 * BatchNorm's backward pass on the tile it holds: it stores g = dout * (x * scale + shift > 0) -- the ReLU-gated gradient
 * -- instead of dout, and writes the per-block partial sums { sum g, sum g * (x - mean) * invstd } per channel into `partial`
 * (fp32 [rows][2][cout], rows = lh_igemm_stats_rows: the layout of the forward statistics).  lh_fuse_bwd then takes them
 * (lh_fuse_bwd_desc.pre_partial / pre_rows) and skips its own reduce pass over dout and x: one read of dout, one launch
 * less per BatchNorm.  x has the layout of this launch's output (same pixel stride).
 * mask (optional; dense outputs): the activation is a residual tail a = relu(BN(x) + r) (`out += residual; relu`,
 * pose_resnet.py:96-97) -- its sign does not follow from x alone, it is read from the relu_mask bits lh_fuse_fwd stored for the tail
 * (scale / shift are then unused and may be NULL).  x2 (optional, with mask; pointwise kernel only): r is a projection shortcut
 * BN2(x2) (`residual = self.downsample(x)`, pose_resnet.py:93-94) -- partial2 takes { sum g, sum g * (x2 - mean2) * invstd2 }.
 * `partial` / `partial2` have lh_igemm_gated_rows(d, dtype, nterms = 1 | 2) rows.
 * Tiled LDS-DMA configurations (lh_igemm_config: ring depth 2..9 and the dense-wave forms) and the persistent pointwise kernel
 * (ring depth 1); LH_ERR_UNSUPPORTED otherwise. */
typedef struct {
    const void* x;
    const float* mean;
    const float* invstd;
    const float* scale;
    const float* shift;
    float* partial;
    const void* mask;
    const void* x2;
    const float* mean2;
    const float* invstd2;
    float* partial2;
} lh_bn_bwd_gate;
int lh_igemm_gated_rows(const lh_igemm_desc* d, int dtype, int nterms);
int lh_igemm_gated(const lh_igemm_desc* d, const void* in, const void* wpack, void* out, const void* addend, const void* addend_mask,
                   const lh_bn_bwd_gate* gate, int dtype, void* stream);
/* Phase batching: 2..4 lh_igemm launches that share input, output tensor, sizes and epilogue and differ only in weight
 * pack, tap list and output placement (ooh, oow) -- the sub-pixel phases of a 4x4/s2 transposed convolution
 * (pose_resnet.py:194-232) or of a stride-2 convolution's data gradient -- as ONE grid.  Phases may have zero taps
 * (they store addend / bias only).  stats: nphase * lh_igemm_phases_rows(..) rows, phase p's rows at p * rows.
 * Requires the LDS-DMA kernel (16-byte aligned rows, regular tap grids); returns LH_ERR_* otherwise. */
int lh_igemm_phases_rows(const lh_igemm_desc* const* descs, int nphase, int dtype);
int lh_igemm_phases(const lh_igemm_desc* const* descs, int nphase, const void* in, const void* const* wpacks,
                    void* out, const void* addend, const void* addend_mask, const float* bias, const float* scale,
                    const float* shift, float* stats, int dtype, void* stream);
/* Inference head (pose_resnet.py:245-246 with eval-mode BatchNorm: x = relu(bn(deconv(x))); x = final_layer(x)): the
 * sub-pixel phases of the LAST transposed convolution with the folded BatchNorm affine and ReLU, and the 1x1 final layer
 * applied to every output tile while it is still in LDS -- the C-channel activation never reaches HBM, only the fp32
 * NCHW heat-map [n][n_out][OH][OW] is written.  Needs the 256 x 256 tile (descs[*]->cfg), cout <= 256, a 16-bit type.
 * head.w: K-major rows [>= 32][w_row_bytes] of the 1x1 weight in the run precision (rows >= n_out zero: the pack
 * lh_pack_weight makes for that convolution), head.bias: fp32 [n_out] or NULL. */
typedef struct {
    const void* w;
    size_t w_row_bytes;
    const float* bias;
    float* out;
    int n_out;
} lh_head;
int lh_igemm_phases_head(const lh_igemm_desc* const* descs, int nphase, const void* in, const void* const* wpacks,
                         const float* scale, const float* shift, const lh_head* head, int dtype, void* stream);
/* tile (output channels x pixels) the dispatcher picks for this descriptor: names the kernel
 * instantiation a launch uses -- igemm_ring_kernel<T, bm, bp, .., depth, kbytes> when *ring != 0 (LDS-DMA ring
 * path, 16-byte aligned pixel rows; *ring = kbytes*10 + depth) else igemm_kernel<T, bm, bp, ..>. */
int lh_igemm_tile(const lh_igemm_desc* d, int dtype, int* bm, int* bp, int* ring);
/* The configuration a launch of this descriptor runs with (cfg5 = tile channels, tile pixels, ring depth, K bytes per
 * stage, 0; depth 0 = the register-staged kernel), and the configurations compiled in that fit it
 * (5 ints each, returns the count; 0 when the form only runs on the register-staged kernel): the plan times them on
 * the real buffers and writes the winner into lh_igemm_desc.cfg (lighthand_amd/engine.py, Plan._tune). */
int lh_igemm_config(const lh_igemm_desc* d, int dtype, int* cfg5);
int lh_igemm_candidates(const lh_igemm_desc* d, int dtype, int* cfgs, int max);
/* Weight gradient of a convolution whose kernel rows are separate "row taps" (the C_in = 3 stem on NHWC4: a kernel row =
 * one contiguous run of k pixels x 4 channels, pose_resnet.py:152) with ALL rows in one pass: d describes ONE tap (the
 * first kernel row) and has k_run = rows * run; gradient input index i = row * run + k is read at input row ih + row,
 * element k of the run.  dy is read once per input tile instead of once per kernel row.  Slab / fold as for lh_wgrad with
 * the same descriptor (ntaps = 1, n_in = d->k_run).  LDS-DMA kernel only (16-bit types, 16-byte aligned runs). */
int lh_wgrad_rowfold(const lh_igemm_desc* d, int rows, const void* x, const void* dy, int dy_pix_stride,
                     int n_out, float* slab, int dtype, void* stream);
int lh_wgrad_tile(const lh_igemm_desc* d, int n_out, int n_in, int dtype, int* bo, int* bi, int* nsplit, int* ring);
/* Launch plans of the LDS-DMA weight-gradient kernel that fit this descriptor: 5 ints each = { tile output channels,
 * tile input channels, cfg[7] encoding (splits | stage rows << 16 | ring depth << 24), workgroups, slab MiB }.  Writing
 * the first three into lh_igemm_desc.cfg[5..7] selects the plan for lh_wgrad / lh_wgrad_slab_bytes / lh_wgrad_reduce
 * (0 when only the register-staged kernel applies: fp32, unaligned pixel rows). */
int lh_wgrad_candidates(const lh_igemm_desc* d, int n_out, int n_in, int dtype, int* out, int max);
/* n independent lh_igemm launches (one entry = the arguments of lh_igemm) as ONE grid: every descriptor's cfg must name the
 * same TILED configuration (ring depth >= 2, a 4-wave tile, 16-bit types); the same layer position of HRNet's parallel
 * branches (pose_hrnet.py:139-185).  Statistics slabs, addends, epilogue vectors are per problem. */
typedef struct {
    const lh_igemm_desc* d; const void* in; const void* wpack; void* out; const void* addend; const void* addend_mask;
    const float* bias; const float* scale; const float* shift; float* stats;
} lh_igemm_call;
int lh_igemm_multi(const lh_igemm_call* calls, int n, int dtype, void* stream);
/* rows of the stats slab lh_igemm writes for this descriptor (= number of pixel tiles) */
int lh_igemm_stats_rows(const lh_igemm_desc* d, int dtype);

/* Weight gradient (loss.backward(), src/utils/method.py:182):
 *   dW[tap][o][i] = sum_pixels dy[pixel][o] * x[pix(tap)][i]
 * with the same tap/gather description (x is the gathered operand, dy is dense
 * [n*ho*wo][dy_pix_stride]).  Partial sums go to `slab` (fp32, lh_wgrad_slab_bytes), then
 * lh_wgrad_reduce folds the split-K slabs into the reference-layout gradient
 * grad[o*so + i*si + r*sr + s*ss] (accumulate != 0 adds to what is there). */
size_t lh_wgrad_slab_bytes(const lh_igemm_desc* d, int n_out, int n_in, int dtype);
int lh_wgrad(const lh_igemm_desc* d, const void* x, const void* dy, int dy_pix_stride,
             int n_out, int n_in, float* slab, int dtype, void* stream);
int lh_wgrad_reduce(const lh_igemm_desc* d, const float* slab, float* grad, int n_out, int n_in,
                    long so, long si, long sr, long ss, const int* taps_rs, int accumulate,
                    int dtype, void* stream);
/* Both steps as one call: lh_wgrad (rows <= 1) or lh_wgrad_rowfold (rows > 1) into `workspace`
 * (lh_wgrad_workspace_bytes, device; launches that run one after another on a stream may share it), then lh_wgrad_reduce
 * into `grad`. */
size_t lh_wgrad_workspace_bytes(const lh_igemm_desc* d, int n_out, int n_in, int dtype);
/* n independent lh_wgrad_fused calls (one entry = its arguments; every problem needs its OWN workspace) as one weight-gradient
 * launch + one fold launch where tile / stage / ring depth agree (cfg[5..7]; 4-wave tiles), one by one otherwise. */
typedef struct {
    const lh_igemm_desc* d; int rows; const void* x; const void* dy; int dy_pix_stride, n_out, n_in; void* workspace; float* grad;
    long so, si, sr, ss; const int* taps_rs; int accumulate;
} lh_wgrad_call;
int lh_wgrad_fused_multi(const lh_wgrad_call* calls, int n, int dtype, void* stream);
int lh_wgrad_fused(const lh_igemm_desc* d, int rows, const void* x, const void* dy, int dy_pix_stride, int n_out, int n_in,
                   void* workspace, float* grad, long so, long si, long sr, long ss, const int* taps_rs, int accumulate,
                   int dtype, void* stream);
/* Table launches -- the deferred weight gradients of a whole stage (loss.backward(), src/utils/method.py:182, through the blocks of
 * one layerN of pose_resnet.py:61-99,177-192 or the branches of one HighResolutionModule, pose_hrnet.py:247-265) as ONE grid of the
 * LDS-DMA weight-gradient kernel plus at most ONE fold grid, whatever their number and shapes.  lh_wgrad_table_build (host only, once
 * per plan: every pointer of a plan is static) turns n lh_wgrad_call entries (16-bit types, rows <= 1; their `workspace` fields are
 * ignored) into a table blob the caller uploads to device memory: argument blocks + work-item lists.  cfg4 = { tile output channels,
 * tile input channels, pixel rows per ring stage, ring depth } (one of the compiled-in configurations, see lh_wgrad_candidates);
 * EVERY PROBLEM GETS ITS OWN PIXEL-SPLIT COUNT: work items of about `target_stages` ring stages each (0 = automatic: split-free when
 * the tiles alone fill the machine twice, else about four rounds of its workgroup slots), ordered longest first.  A problem with one
 * split, one tap and a gradient that is dense in [o][i] is written by the kernel itself (no slab, no fold entry).  The sums are a
 * deterministic function of (calls, cfg4, target_stages).  Call with host_blob = NULL to size the blob and the shared fp32 slab
 * workspace (info->table_bytes, info->workspace_bytes; runs without a device), then with both buffers.  lh_wgrad_table_run replays it:
 * `table` = the device copy of the blob. */
typedef struct {
    int bo, bi, kps, depth;              /* kernel configuration */
    int n_problems, n_fold;              /* problems in the table, problems that need the fold launch */
    int n_items, n_fold_items;           /* workgroups of the weight-gradient launch / of the fold launch (0: no fold launch) */
    int fold_lds, target_stages, nsplit_max;
    int run_parts;                       /* lh_wgrad_table_run: 0 = both launches, 1 = the weight-gradient grid only, 2 = the fold grid only (timing) */
    size_t off_items, off_fold_args, off_fold_items, table_bytes, workspace_bytes;
} lh_wgrad_table_info;
int lh_wgrad_table_build(const lh_wgrad_call* calls, int n, int dtype, const int* cfg4, int target_stages, void* workspace,
                         void* host_blob, size_t host_bytes, lh_wgrad_table_info* info);
int lh_wgrad_table_run(const void* table, const lh_wgrad_table_info* info, int dtype, void* stream);

/* ------------------------------------------------------------------ BatchNorm / fused elementwise
 * nn.BatchNorm2d(C, momentum=0.1): pose_resnet.py:19,35,... ; nn.ReLU / residual add:
 * pose_resnet.py:55-56,96-97; nn.Upsample(nearest): pose_hrnet.py:207. */

/* Column sums / sums of squares of a dense [m][c] activation -> stats slab [rows][2][c]. */
int lh_bn_stats(const void* y, int m, int c, float* stats, int* rows_out, int dtype, void* stream);
int lh_bn_stats_rows(int m, int c);
/* Bytes a stats slab of `rows` rows and `c` channels must have (slab + fp64 fold scratch). */
size_t lh_bn_stats_slab_bytes(int rows, int c);
/* Fold a stats slab: batch mean / biased var -> scale = gamma*rsqrt(var+eps),
 * shift = beta - mean*scale, saved mean / invstd; running stats updated with momentum and
 * the unbiased variance, num_batches_tracked (int64) += 1 (all device pointers). */
int lh_bn_finalize(const float* stats, int rows, int count, int c, const float* gamma,
                   const float* beta, float* running_mean, float* running_var,
                   long long* num_batches_tracked, float momentum, float eps, float* scale,
                   float* shift, float* save_mean, float* save_invstd, void* stream);
/* n independent BatchNorm layers (one entry = the arguments of lh_bn_finalize) as one launch; see lh_fuse_fwd_multi. */
typedef struct {
    const float* stats; int rows, count, c; const float* gamma; const float* beta; float* running_mean; float* running_var;
    long long* num_batches_tracked; float momentum, eps; float* scale; float* shift; float* save_mean; float* save_invstd;
} lh_bn_finalize_call;
int lh_bn_finalize_multi(const lh_bn_finalize_call* calls, int n, void* stream);
/* Eval mode: scale/shift from running statistics. */
int lh_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, int c, float* scale, float* shift,
                      void* stream);

/* out = relu?( sum_t term_t ),  term_t = up_t( x_t * scale_t + shift_t )   (scale_t NULL = identity).
 * Up to 4 terms; up_t is nearest-neighbour upsampling by 2^log2up of a [n][h>>l][w>>l][c] map. */
typedef struct {
    const void* x[4];
    const float* scale[4];
    const float* shift[4];
    int log2up[4];
    int nterms;
    int relu;
    void* relu_mask;            /* optional output: one byte per 16-byte chunk of `out`, bit e = (element e > 0); lets the
                                 * backward pass read n*h*w*c/8 mask bytes instead of the whole stored activation */
    const lh_bn_finalize_call* fin[4];
                                /* optional, per term: the BatchNorm finalize of that term (batch statistics -> scale / shift /
                                 * saved mean / invstd / running statistics: what lh_bn_finalize does) has NOT run yet and is part
                                 * of this call.  Small tensors fold the statistics slab inside the elementwise launch itself
                                 * (one launch instead of two on the dependency chain of every BatchNorm); otherwise the call
                                 * launches the finalize first.  fin[t]->scale / ->shift must equal scale[t] / shift[t].
                                 * ZERO-INITIALISE the descriptor (memset / = {0}): a caller compiled against the round-3 layout, or
                                 * one that fills the fields one by one, would otherwise pass garbage pointers here. */
    const void* l2_touch;       /* optional (round 5): l2_touch_bytes of device memory the NEXT launch on the stream reads first -- the
                                 * weight pack of the convolution that consumes `out`.  The streaming kernels read one word of every
                                 * 128-byte line of it in each XCD right before they end, so the convolution finds it in L2
                                 * (speed only; ignored by the kernels that have no such tail).  NULL / 0 = off. */
    size_t l2_touch_bytes;
} lh_fuse_desc;
int lh_fuse_fwd(const lh_fuse_desc* d, void* out, int n, int h, int w, int c, int dtype, void* stream);
/* Multi-problem forms (pose_hrnet.py:139-185, 247-265: the same layer position of the 2-4 parallel branches of a
 * HighResolutionModule): n INDEPENDENT calls -- one entry = the arguments of the single call -- run as one launch per
 * kernel instead of n (the argument blocks travel in the kernel-argument segment, four problems per launch).  Calls
 * that do not plan the same kernels (e.g. an upsampled term beside plain ones) run one by one: always legal, same results. */
typedef struct { const lh_fuse_desc* d; void* out; int n, h, w, c; } lh_fuse_fwd_call;
int lh_fuse_fwd_multi(const lh_fuse_fwd_call* calls, int n, int dtype, void* stream);

/* Backward of lh_fuse_fwd, two launches per BN term:
 *  reduce: sums[t] = { sum g_t, sum g_t * xhat_t } with g = dout * (out > 0), g_t = g summed over
 *          each upsampling cell;  apply: dx_t = scale_t * (g_t - mean(g_t) - xhat_t * mean(g_t xhat_t));
 *          identity terms get dx_t = g_t.  dgamma = sum g xhat, dbeta = sum g are written to
 *          dgamma/dbeta (fp32 [c]).  accumulate[t] != 0 adds into dx[t]. */
typedef struct {
    const void* dout;
    const void* out;            /* forward result (ReLU mask); NULL when relu == 0 */
    const void* x[4];           /* raw BN inputs (NULL for identity terms)          */
    const float* scale[4];
    const float* shift[4];      /* forward shift vectors (lets a single-BN-term ReLU mask be recomputed from x) */
    const float* save_mean[4];
    const float* save_invstd[4];
    void* dx[4];
    float* dgamma[4];
    float* dbeta[4];
    int log2up[4];
    int accumulate[4];
    int nterms;
    int relu;
    const void* relu_mask;      /* mask bits written by lh_fuse_fwd; when set, `out` is not read and may be NULL */
    int strips_cap;             /* 0 = default (512): upper bound on the strips of the streaming reduce pass = rows of the
                                 * partial-sum slab; 256 is the measured choice for nodes that share lh_fuse_bwd_multi launches */
    const float* pre_partial;   /* a ReLU node with ONE BN term -- alone, or beside one identity term (a residual tail) -- or with TWO BN
                                 * terms (a tail with a projection shortcut; pre_partial2 then holds term 1's sums) whose dout was written
                                 * by lh_igemm_gated: dout is already the gated gradient and these are its partial sums
                                 * [pre_rows][2][c] -- no reduce pass, no mask */
    int pre_rows;
    const void* l2_touch;       /* optional (round 5), as lh_fuse_desc.l2_touch: the last apply pass of the call warms these bytes in L2 */
    size_t l2_touch_bytes;
    const float* pre_partial2;  /* with pre_partial and two BN terms: the partial sums of term 1 (pre_partial: term 0), same rows */
} lh_fuse_bwd_desc;
size_t lh_fuse_bwd_workspace_bytes(int n, int h, int w, int c);
int lh_fuse_bwd(const lh_fuse_bwd_desc* d, int n, int h, int w, int c, void* workspace,
                int dtype, void* stream);
/* n independent nodes, each with its OWN workspace: their reduce / coefficient-fold / apply kernels as three launches. */
typedef struct { const lh_fuse_bwd_desc* d; int n, h, w, c; void* workspace; } lh_fuse_bwd_call;
int lh_fuse_bwd_multi(const lh_fuse_bwd_call* calls, int n, int dtype, void* stream);

/* The inference stem of SimpleBaseline-ResNet as ONE launch: maxpool(relu(bn1(conv1(x)))) -- conv1 = nn.Conv2d(3, 64, 7, 2, 3),
 * bn1 in eval mode (running statistics folded into scale / shift), nn.MaxPool2d(3, 2, 1): src/modeling/simplebaseline/
 * pose_resnet.py:151-156 and the first four lines of PoseResNet.forward.  img: the zero-padded NHWC4 image the engine's input
 * transforms write ([n][hp][wp][4], image pixel (y, x) at (y + 3, x + 3)); wpack: the stem's weight pack (64 channels, one tap
 * per kernel ROW, K run = 8 pixels x 4 channels); out = [n][ph][pw][64] with ph = (conv_h - 1) / 2 + 1.  relu must be 1 (the
 * pool's padding is realised as zeros).  16-bit types.  Bit-identical to lh_igemm + lh_maxpool3x3s2_fwd on the same operands. */
int lh_stem_pool(const void* img, int n, int hp, int wp, const void* wpack, const float* bias, const float* scale,
                 const float* shift, void* out, int conv_h, int conv_w, int relu, int dtype, void* stream);

/* The inference BOTTLENECK of the first ResNet stage as ONE launch (round 5):
 *   out = relu( bn3(conv3( relu(bn2(conv2( relu(bn1(conv1(x))) ))) )) + residual )
 * conv1 = 1x1 (cin -> 64), conv2 = 3x3 / stride 1 / pad 1 (64 -> 64), conv3 = 1x1 (64 -> 256), the BatchNorms in eval mode folded
 * into per-channel scale / shift (lh_bn_eval_affine): Bottleneck.forward, src/modeling/simplebaseline/pose_resnet.py:61-99, for the
 * blocks with stride 1 (pytorch and caffe style alike).  x = [n][h][w][cin], residual and out = [n][h][w][256] (residual = x for an
 * identity shortcut, the projection's output otherwise; out may alias neither).  w1 / w2 / w3: the weight packs lh_pack_weight makes
 * for those three convolutions (K-major rows, taps in (r, s) order).  16-bit types.  Only x and the residual are read and out is
 * written -- the two 64-channel intermediates stay in LDS.  Bit-identical to the three lh_igemm launches with the same folds. */
typedef struct { int n, h, w, cin, mid, cout; } lh_bottleneck_desc;
int lh_bottleneck_infer(const lh_bottleneck_desc* d, const void* x, const void* w1, const void* w2, const void* w3,
                        const float* s1, const float* b1, const float* s2, const float* b2, const float* s3, const float* b3,
                        const void* residual, void* out, int dtype, void* stream);

/* The TRAINING stem: conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False) alone (pose_resnet.py:151-152; bn1 runs on batch statistics),
 * same operands as lh_stem_pool (padded NHWC4 image, the stem's weight pack), on the same direct kernel form: out =
 * [n][conv_h][conv_w][64] raw convolution output (bit-identical to lh_igemm on that pack), stats = fp32
 * [lh_stem_conv_rows(n, conv_h, conv_w)][2][64] per-workgroup sums / sums of squares of the stored values for lh_bn_finalize.
 * 16-bit types. */
int lh_stem_conv_rows(int n, int conv_h, int conv_w);
int lh_stem_conv(const void* img, int n, int hp, int wp, const void* wpack, void* out, float* stats, int conv_h, int conv_w,
                 int dtype, void* stream);

/* nn.MaxPool2d(3, 2, 1): pose_resnet.py:156.  idx (uint8 [n][ho][wo][c]) keeps the window
 * position (first maximum in scan order, NaN propagates) for the backward pass; NULL when no backward pass follows. */
int lh_maxpool3x3s2_fwd(const void* x, void* out, unsigned char* idx, int n, int h, int w, int c,
                        int dtype, void* stream);
int lh_maxpool3x3s2_bwd(const void* dout, const unsigned char* idx, void* dx, int n, int h, int w,
                        int c, int dtype, void* stream);
/* lh_maxpool3x3s2_bwd whose output dx is the gradient of a = relu(BN(gate->x)) (the training stem: the pool follows bn1 + relu):
 * stores the ReLU-gated gradient and one row of BatchNorm-backward partial sums per workgroup (gate->partial: fp32
 * [lh_maxpool3x3s2_bwd_gated_rows][2][c]), for lh_fuse_bwd's pre_partial -- see lh_igemm_gated.  16-bit types. */
int lh_maxpool3x3s2_bwd_gated_rows(int n, int h, int w, int c, int dtype);
int lh_maxpool3x3s2_bwd_gated(const void* dout, const unsigned char* idx, void* dx, const lh_bn_bwd_gate* gate, int n, int h, int w,
                              int c, int dtype, void* stream);
/* maxpool(relu(bn(x))) of the training stem (pose_resnet.py:153-156: bn1, relu, maxpool) as ONE pass: x is the RAW BatchNorm
 * input, scale / shift the training-mode affine lh_bn_finalize derived from the batch statistics; every tap is
 * relu(x * scale + shift) rounded to the run precision -- the value lh_fuse_fwd would have stored -- so out and idx are
 * bit-identical to lh_fuse_fwd followed by lh_maxpool3x3s2_fwd, and the activation between them (the largest of the
 * network) is never written: the backward pass does not need it (lh_maxpool3x3s2_bwd works from idx, lh_fuse_bwd
 * recomputes the ReLU mask from x). */
int lh_bn_relu_maxpool3x3s2_fwd(const void* x, const float* scale, const float* shift, void* out, unsigned char* idx, int n,
                                int h, int w, int c, int dtype, void* stream);

/* ------------------------------------------------------------------ heatmap target / loss / decode */
/* CustomDataset.generate_target: src/tools/dataset.py:165-212.  joints fp32 [b][j][jstride]
 * (x, y in input pixels) -> fp32 [b][j][size][size].  `patch` is the (2*radius+1)^2 fp32
 * Gaussian the host evaluates exactly as the reference does (numpy float32 exp), so that the
 * placed values are bit-identical to the reference's on the same host. */
int lh_gaussian_target(const float* joints, int jstride, const float* patch, int radius,
                       float* target, int b, int j, int size, void* stream);
/* GenerateHeatmap.__call__: src/utils/dataset_loader.py:22-53 (the alternate renderer the reference's dataset classes
 * keep beside generate_target).  points fp32 [b][j][pstride] ALREADY in heat-map coordinates -> fp32 [b][j][res][res];
 * `patch` = the (6*sigma+3)^2 Gaussian the host evaluates in float64 like the reference and rounds to fp32 (what the
 * reference's float32 map stores); sigma = res / 64 must be an integer (the reference uses res = 64). */
int lh_gaussian_target_alt(const float* points, int pstride, const float* patch, int sigma,
                           float* target, int b, int j, int res, void* stream);
/* JointsMSELoss(use_target_weight=False): src/utils/loss.py:306-325.  loss (fp32 scalar on
 * device) = 0.5*mean((p-g)^2); grad (optional) = (p-g) * grad_scale/(numel). workspace >=
 * lh_mse_workspace_bytes(numel). */
size_t lh_mse_workspace_bytes(long numel);
int lh_mse_heatmap(const float* pred, const float* target, long numel, float* loss, float* grad,
                   const float* grad_scale, void* workspace, void* stream);
/* get_max_preds: src/utils/loss.py:327-355 (+ the x4 of src/utils/method.py:157,176-178 via
 * `scale`).  heatmaps fp32 NCHW [b*j][h*w] -> preds fp32 [b*j][2], maxvals fp32 [b*j],
 * idx int32 [b*j] (first-occurrence arg-max, NaN counts as maximum). */
int lh_heatmap_argmax(const float* heatmaps, int bj, int h, int w, float scale, float* preds,
                      float* maxvals, int* idx, void* stream);
/* Opt-in quarter-pixel refinement of lh_heatmap_argmax's result (NOT in the reference: its config carries the unused
 * switch TEST.POST_PROCESS, src/modeling/simplebaseline/config.py:109; this is the published SimpleBaseline rule
 * coord += 0.25 * sign(hm[+1] - hm[-1]) per axis for peaks strictly inside the map).  idx / maxvals / preds are
 * lh_heatmap_argmax's outputs (same scale); preds is updated in place. */
/* Opt-in soft-arg-max decode (the project brief names it; the reference decodes with the hard arg-max above, so this is an
 * extension without a reference oracle): preds[b*j] = sum_p softmax(beta * hm)[p] * (x_p, y_p) * scale. */
int lh_heatmap_soft_argmax(const float* heatmaps, int bj, int h, int w, float beta, float scale, float* preds,
                           void* stream);
int lh_heatmap_refine(const float* heatmaps, const int* idx, const float* maxvals, int bj, int h, int w,
                      float scale, float* preds, void* stream);

/* Validation metrics of Runner.run (src/utils/method.py:243-250) on the device: per sample, wrong[b] = number of
 * joints with error / bbox-diagonal(gt) > T (PCK_2d_loss 'proportion', src/utils/loss.py:116-148) and epe[b] = sum of
 * the errors of joints 1..J-2 (EPE_train's joint range, src/utils/loss.py:50-67).  pred fp32 [b][j][2],
 * gt fp32 [b][j][gt_stride]. */
int lh_keypoint_metrics(const float* pred, const float* gt, int gt_stride, int b, int j, float T, int* wrong,
                        float* epe, void* stream);

/* ------------------------------------------------------------------ optimiser
 * torch.optim.Adam(lr, betas, eps, weight_decay=0).step(): src/tools/train.py:45-48,
 * src/utils/method.py:183.  One launch over a flat fp32 arena (plus a one-thread
 * tick kernel).  hyper (device, fp64[4]) = { lr, beta1, beta2, eps }; step (device int32) is
 * incremented on the device, so a captured hipGraph replays correctly; derived (device fp32[8])
 * is scratch for the bias-corrected step size evaluated in fp64 like the reference's Python. */
int lh_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long numel,
                 const double* hyper, int* step, float* derived, float grad_scale, void* stream);
/* The two halves of lh_adam_step, for an update applied SLICE BY SLICE while the backward pass still runs (a gradient
 * bucket's parameters are updated as soon as the bucket is final / all-reduced; torch.optim.Adam's update is elementwise,
 * so the result is bit-identical to one lh_adam_step over the whole range): lh_adam_tick once per iteration (step += 1,
 * derived = bias-corrected step size ...), then lh_adam_apply per slice -- pointers advanced to the slice, which must
 * start on a 16-byte boundary. */
int lh_adam_tick(const double* hyper, int* step, float* derived, void* stream);
int lh_adam_apply(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long numel, const float* derived,
                  float grad_scale, void* stream);

/* Bias gradient of the head's 1x1 convolution (pose_resnet.py:169-175; loss.backward()): out[c] = sum over n, h, w of an
 * NCHW fp32 gradient.  fp64 partials in a fixed order (deterministic).  workspace >= lh_channel_sum_workspace_bytes(c). */
size_t lh_channel_sum_workspace_bytes(int c);
int lh_channel_sum_nchw(const float* x, int n, int c, int hw, float* out, void* workspace, void* stream);

/* Bias gradient of a convolution / transposed convolution WITH bias inside the network (DECONV_WITH_BIAS:
 * src/modeling/simplebaseline/pose_resnet.py:149,227; what autograd's loss.backward() adds for nn.ConvTranspose2d(bias=True)):
 * out[ch] = sum over `pixels` rows of an NHWC gradient of the run precision, channels [0, c) of rows of pix_stride
 * elements (16-byte aligned rows).  fp64 partials in a fixed order (deterministic); same workspace size. */
int lh_channel_sum_nhwc(const void* x, long pixels, int c, int pix_stride, float* out, void* workspace, int dtype, void* stream);

/* PCK curve of pred_eval (src/utils/argparser.py:326-388) on the device: counts[t] += visible joints (gt[..][2] == 1) whose
 * error (pixel distance; divided by bb[sample] when bb != NULL, the 'pckb' mode) is < thr[t]; *nvis += visible joints;
 * diff_row[s] = sum of pixel errors over ALL joints of sample s.  float64 arithmetic like the NumPy original; counts /
 * nvis ACCUMULATE (zero them first) with integer atomics, so the result is exact and ranks can be summed by one small
 * all-reduce.  AUC = trapz(100*counts/nvis, thr) / trapz(1, thr) on the host (lighthand_amd.metrics.auc_from_counts). */
int lh_pck_curve(const float* pred, const float* gt, int gt_stride, const float* bb, int n, int j, const double* thr,
                 int nthr, unsigned long long* counts, unsigned long long* nvis, double* diff_row, void* stream);

/* ------------------------------------------------------------------ data-parallel gradient exchange
 * The reference trains on one device (no DistributedDataParallel anywhere: src/utils/comm.py:15-32 only queries rank /
 * world size for logging and checkpoint gating); the data-parallel path is this engine's addition.  One
 * communicator per process (= per GPU) over RCCL / xGMI: rank 0 creates a 128-byte id (lh_comm_unique_id) that the host
 * side hands to every rank (any channel: torch.distributed's store, a file); lh_comm_allreduce_sum sums one gradient
 * bucket in place across the ranks, asynchronously on `stream` (legal inside hipGraph capture).  dtype: LH_F32 for exact
 * sums, LH_BF16 for half the bytes per link.  The library binds to the RCCL already loaded in the process. */
typedef struct lh_comm lh_comm;
int lh_comm_unique_id(void* id128);
int lh_comm_init(lh_comm** comm, int rank, int nranks, const void* id128);
int lh_comm_allreduce_sum(lh_comm* comm, void* buf, size_t count, int dtype, void* stream);
/* The pieces of a DIRECT gradient exchange (reduce-scatter + all-gather with all seven xGMI peers at once, 2 x bytes / N
 * per link instead of a ring's 2 (N - 1) / N x bytes over one), all stream-ordered and capturable like the all-reduce, so that the whole
 * data-parallel step stays ONE hipGraph (parallel.LhComm.direct_sum_):
 *   lh_comm_alltoall            recv chunk r <- rank r's send chunk `rank` (grouped point-to-point exchange, `count` elements per chunk);
 *                               then lh_sum_chunks adds the N received chunks in RANK order -- every rank ends with the same bits;
 *   lh_comm_allgather           recv chunk r <- rank r's `count` elements (recv + rank * count == send: in place);
 *   lh_comm_reduce_scatter_sum  RCCL's own reduce-scatter (recv <- the sum of every rank's send chunk `rank`), for comparison.
 * lh_comm_size: the rank / rank count the communicator was created with. */
int lh_comm_reduce_scatter_sum(lh_comm* comm, const void* send, void* recv, size_t count, int dtype, void* stream);
int lh_comm_allgather(lh_comm* comm, const void* send, void* recv, size_t count, int dtype, void* stream);
int lh_comm_alltoall(lh_comm* comm, const void* send, void* recv, size_t count, int dtype, void* stream);
int lh_comm_size(const lh_comm* comm, int* rank, int* nranks);
int lh_comm_destroy(lh_comm* comm);

/* dst[i0][i1][i2][i3] = src[i0][i1][i2][i3], fp32, element strides on both sides (shape4 / strides are HOST arrays read at
 * call time).  The layout shuffles of a step that the reference leaves to tensor views: the stem weight and its gradient
 * between [O][3][k][k] (pose_resnet.py:151) and the padded NHWC4 staging, the head-gradient crop, bias padding. */
int lh_copy_strided_f32(float* dst, const float* src, const int* shape4, const long* dst_strides4, const long* src_strides4,
                        void* stream);


/* Staging of a gradient bucket that travels as bfloat16 (parallel.GradSync(compress="bf16"); the reference has no
 * multi-GPU path): to_f32 = 0 rounds the fp32 arena slice `f32` into the bf16 buffer `b16` (nearest even),
 * to_f32 = 1 widens `b16` back into `f32`.  n elements, any alignment. */
int lh_cast_f32_bf16(float* f32, void* b16, long n, int to_f32, void* stream);

/* The local reduction of the DIRECT gradient exchange (parallel.GradSync(algo="direct"): all-to-all + this + all-gather = reduce-scatter
 * and all-gather with every xGMI peer at once): out[i] = sum over r < rows of in[r * len + i], accumulated in fp32 in row
 * (= rank) order and rounded once; dtype LH_F32 or LH_BF16; `out` may be the first row of `in`. */
int lh_sum_chunks(const void* in, void* out, int rows, long len, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LIGHTHAND_HIP_H */
