/*
 * niw.h -- C ABI of libniw_hip.so: the MI355X (gfx950) implementation of the volumetric
 * rendering hot path of sfchng/neural_invertible_warp.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer to caller-allocated, contiguous memory (fp32 unless
 *     stated; int64 for pixel indices); the library never allocates persistent memory and
 *     never synchronises; workspaces are sized with the *_floats() queries and passed in;
 *   - `stream` is a hipStream_t passed as void* (the caller's current stream);
 *   - return value 0 = success, negative = niw_status; the message of the last failure on
 *     the calling thread is returned by niw_last_error_string(); nothing throws;
 *   - outputs are fully overwritten unless the parameter is documented as "accumulated";
 *   - threads: every entry point may be called from any host thread.  The two that use a library-owned second stream
 *     (niw_mlp_bwd_dw from 96,000 samples, niw_train_step with overlap) share ONE such stream and its fork / join events per
 *     device; they hold a per-device lock from the fork to the join, so two threads (or two trainers on different streams) take
 *     turns through that part of the call rather than re-recording each other's events -- and a call that fails behind the
 *     fork still joins the stream before it returns (a capture in progress stays valid).
 *
 * Each declaration names the reference interface (file:line in the reference repository)
 * whose arithmetic it replaces.  The Python mirror of the reference's classes lives in
 * neural_invertible_warp_amd/ and binds these symbols with ctypes.
 */
#ifndef NIW_H_
#define NIW_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* niw_stream_t;

enum niw_status {
    NIW_OK = 0,
    NIW_ERR_INVALID_ARG = -1,   /* shape / enum / null-pointer check failed on the host */
    NIW_ERR_UNSUPPORTED = -2,   /* architecture constant outside what the kernels are built for */
    NIW_ERR_LAUNCH = -3         /* hipGetLastError() after a launch */
};

int niw_version(void);
const char* niw_last_error_string(void);

/* ------------------------------------------------------------------ field MLP constants
 * The kernels are specialised for the architecture used by all reference configs
 * (options/nerf_llff_repr.yaml:3-10, options/nerf_inn_llff.yaml:3-10):
 *   8 x 256 feature layers, skip at layer 4, 128-wide colour layer, L_3D = 10, L_view = 4. */
#define NIW_L3D 10
#define NIW_LVIEW 4
#define NIW_ENC_SLOTS 64          /* 3 + 6*L_3D = 63 input features, padded, in MFMA slot order */
#define NIW_VENC_SLOTS 32         /* 3 + 6*L_view = 27 */
#define NIW_NERF_PARAM_FLOATS 530052   /* 527,872 weights + 2,180 biases, state-dict order */

/* rows of the saved-activation / saved-gradient workspaces: NIW_*_ROWS * Mpad floats each, opaque to the caller (internally a
 * feature-major [rows][Mpad] matrix stored as a quad-row image [rows / 4][Mpad][4]; the raw-density row and the mask records plain) */
#define NIW_SAVE_ROWS 2346        /* enc 64 | h1..h7 7*256 | feat 256 | venc 32 | hr 128 | sigma_raw 1 | rgb(unused) 1 | ReLU bit masks 72 (9 KiB per 32 samples) */
#define NIW_GRAD_ROWS 2336        /* dY0..dY6 7*256 | dY7 288 (row 256 = d sigma_raw) | dYrgb0 128 | dYrgb1 32 | stash 64+32 */

enum niw_density_activ { NIW_ACT_RELU = 0, NIW_ACT_SOFTPLUS = 1 };

/* Arithmetic of the field-MLP entry points (the `precision` argument; SURVEY section 8(b) "flags: exact-fp32 / bf16").
 *   NIW_PREC_FP32    v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fmaf chain.  The default everywhere, the mode of every headline number
 *                    and of every parity claim at the reference's fp32 tolerance.
 *   NIW_PREC_BF16X3  opt-in: every operand carried as two bf16 planes (16 significand bits), a product formed as
 *                    hi*hi + hi*mid + mid*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (3/16 of the matrix-pipe time).
 *   NIW_PREC_BF16    opt-in: the leading bf16 plane only (1/16 of the matrix-pipe time; SURVEY 8(c)'s bf16 tolerance class).
 *                    Its three kernels keep the saved activations / gradients (`save`, `gradws`) as bf16 quad rows inside the same
 *                    caller-provided buffers (half the bytes; the layout is private to the mode): forward, dX and dW of one
 *                    step must be called with the SAME precision.
 * A packed-weight image belongs to ONE precision class: fp32 images (niw_mlp_pack_weights*) for NIW_PREC_FP32, the split image of
 * niw_mlp_pack_weights_prec for the two bf16 modes (one image serves both). */
enum niw_precision { NIW_PREC_FP32 = 0, NIW_PREC_BF16X3 = 1, NIW_PREC_BF16 = 2 };

/* Mpad: number of (ray,sample) rows rounded up to the 128-row workgroup tile. */
int64_t niw_mlp_padded_rows(int64_t n_rays, int n_samples);
/* floats of the packed-weight buffer written by niw_mlp_pack_weights */
int64_t niw_mlp_packed_floats(void);
/* bytes of the packed-weight image of a precision class (NIW_PREC_FP32: 4 * niw_mlp_packed_floats()), and its builder */
int64_t niw_mlp_packed_bytes(int precision);
int niw_mlp_pack_weights_prec(const float* params, int precision, void* packed, niw_stream_t stream);
/* floats of the split-M partial-sum workspace used by niw_mlp_bwd */
int64_t niw_mlp_bwd_workspace_floats(int64_t n_rays, int n_samples);

/* Re-orders the nn.Linear weights of NeRF (model/nerf.py:373-400; `params` = the 20 tensors
 * mlp_feat.{0..7}.{weight,bias}, mlp_rgb.{0,1}.{weight,bias} concatenated in that order,
 * NIW_NERF_PARAM_FLOATS floats) into MFMA A-fragment order for the forward and the
 * transposed order for the backward.  Call after every optimizer step. */
int niw_mlp_pack_weights(const float* params, float* packed, niw_stream_t stream);
/* The same re-ordering through a gather table: niw_mlp_pack_index fills `index` [niw_mlp_packed_floats()] once (it depends
 * only on the architecture constants; -1 = zero padding), niw_mlp_pack_weights_indexed then is a plain gather (~4x faster
 * than decoding the fragment layout per element; a training step packs two networks). */
int niw_mlp_pack_index(int32_t* index, niw_stream_t stream);
int niw_mlp_pack_weights_indexed(const float* params, const int32_t* index, float* packed, niw_stream_t stream);

/* NeRF.forward_samples (model/nerf.py:449-456) = get_3D_points_from_depth (camera.py:517-521)
 * + F.normalize + NeRF.forward (model/nerf.py:416-447) incl. positional_encoding
 * (model/nerf.py:476-483) with the BARF c2f band weights (model/barf_inn_llff.py:427-442).
 *   center, ray  [n_rays,3]; depth [n_rays,n_samples]
 *   band_w3d[10], band_wview[4]  HOST arrays, copied into the launch (NULL = all ones)
 *   band_dev     DEVICE array of 14 floats {w3d[10], wview[4]} or NULL; when given it overrides the host arrays and is
 *                read by the kernel at run time, so a captured HIP graph replays with whatever the buffer then holds
 *   noise        [n_rays*n_samples] or NULL (density_noise_reg * randn, nerf.py:428-429)
 *   rgb [n_rays,n_samples,3], sigma [n_rays,n_samples]  (outputs)
 *   save         [NIW_SAVE_ROWS, Mpad] or NULL; non-NULL = training mode (activations kept
 *                for niw_mlp_bwd: fp32 quad rows under NIW_PREC_FP32 / NIW_PREC_BF16X3, bf16 half-pitch quad rows inside the
 *                same buffer under NIW_PREC_BF16 -- the backward entry points must be called with the forward's precision).
 *   precision    enum niw_precision; `packed` must be the image of that class */
int niw_mlp_fwd(const float* packed, const float* center, const float* ray,
                const float* depth, const float* noise, int64_t n_rays, int n_samples,
                const float* band_w3d, const float* band_wview, const float* band_dev, int density_activ, int precision,
                float* rgb, float* sigma, float* save, niw_stream_t stream);

/* Backward of niw_mlp_fwd (autograd of model/nerf.py:416-456).
 *   d_rgb [n_rays,n_samples,3], d_sigma [n_rays,n_samples]: incoming gradients
 *   save: the buffer niw_mlp_fwd filled; gradws [NIW_GRAD_ROWS, Mpad] scratch;
 *   partial: niw_mlp_bwd_workspace_floats() scratch
 *   d_params [NIW_NERF_PARAM_FLOATS]: OVERWRITTEN with dL/dparams (state-dict order)
 *   d_center, d_ray [n_rays,3] or both NULL: OVERWRITTEN with the gradients w.r.t. the rays -- per-sample terms summed per
 *   ray in a fixed order, bit-reproducible from run to run (two of the three routes of SURVEY section 8a: sample points
 *   and view directions; the ray-length route belongs to niw_composite_bwd). */
int niw_mlp_bwd(const float* packed, const float* center, const float* ray,
                const float* depth, int64_t n_rays, int n_samples, int density_activ, int precision,
                const float* rgb, const float* d_rgb, const float* d_sigma,
                const float* save, float* gradws, float* partial,
                float* d_params, float* d_center, float* d_ray, niw_stream_t stream);

/* The two passes of niw_mlp_bwd as separate entry points (niw_mlp_bwd = dx then dw):
 *   niw_mlp_bwd_dx: the register-chained dX chain; writes every dY into gradws, then d_center / d_ray (a second, small launch);
 *   niw_mlp_bwd_dw: dW = dY . X^T (split-M NT GEMMs on the fp32 MFMA path) + deterministic reduction.  In exact mode from 96,000
 *     samples its two matrix-vector pieces (density row, colour rows) run on a library-owned second stream, forked from and joined to
 *     `stream` by events inside the call (capturable; the stream is created at the first such call or by niw_train_step_prepare(),
 *     which must therefore come before a stream capture that contains the call).  `stream`-ordered like every other entry point. */
int niw_mlp_bwd_dx(const float* packed, const float* center, const float* ray, const float* depth,
                   int64_t n_rays, int n_samples, int density_activ, int precision,
                   const float* rgb, const float* d_rgb, const float* d_sigma,
                   const float* save, float* gradws, float* d_center, float* d_ray, niw_stream_t stream);
int niw_mlp_bwd_dw(const float* save, const float* gradws, int64_t n_rays, int n_samples, int precision, float* partial,
                   float* d_params, niw_stream_t stream);

/* ------------------------------------------------------------------ compositing
 * NeRF.composite (model/nerf.py:458-474).  ray [n_rays,3], rgb_s [n_rays,S,3],
 * sigma_s, depth_s [n_rays,S] -> rgb [n_rays,3], depth, opacity [n_rays], prob [n_rays,S]
 * (prob may be NULL).  bg: background colour added as bg*(1-opacity) when has_bg != 0
 * (opt.nerf.setbg_opaque, nerf.py:472-473). */
int niw_composite_fwd(const float* ray, const float* rgb_s, const float* sigma_s, const float* depth_s,
                      int64_t n_rays, int n_samples, int has_bg, float bg,
                      float* rgb, float* depth, float* opacity, float* prob, niw_stream_t stream);

/* Closed-form backward.  d_prob may be NULL.  Outputs: d_rgb_s
 * [n_rays,S,3], d_sigma_s [n_rays,S], d_ray [n_rays,3] (all overwritten). */
int niw_composite_bwd(const float* ray, const float* rgb_s, const float* sigma_s, const float* depth_s,
                      int64_t n_rays, int n_samples, int has_bg, float bg,
                      const float* d_rgb, const float* d_depth, const float* d_opacity, const float* d_prob,
                      float* d_rgb_s, float* d_sigma_s, float* d_ray, niw_stream_t stream);

/* Training form of the three calls a train iteration makes around the photometric loss (Graph.render's composite call,
 * model/nerf.py:305 / nerf_inn_llff.py:602, then compute_loss's MSE, nerf_inn_llff.py:555-559 with base.py:209-211, then autograd of
 * both): niw_composite_fwd + niw_mse_fwd_bwd + niw_composite_bwd of one ray batch as ONE launch.  The ray's residual against its pixel
 * is local to the ray (d rgb = grad_scale * 2 (rgb - pixel) / n_norm), so the backward runs on the registers of the forward; the
 * loss VALUE is the only reduction: every ray leaves its three residuals in resid [n_rays,3], and niw_mse_from_residuals sums their
 * squares in niw_mse_fwd_bwd's order -- outputs, gradients and loss are bit-identical to the three calls.
 *   image [n_views,3,hw], ray_idx [n_rays_per_view] or NULL, first_ray, n_norm, grad_scale: as niw_mse_fwd_bwd (below);
 *   rgb, depth, opacity, prob (may be NULL): as niw_composite_fwd;  d_rgb [n_rays,3] (may be NULL): what niw_mse_fwd_bwd writes;
 *   d_rgb_s, d_sigma_s, d_ray: as niw_composite_bwd.  No background colour, no incoming depth / opacity / prob gradients (the train
 *   iteration has none).  NIW_ERR_UNSUPPORTED, nothing launched, unless n_samples % 4 == 0, n_samples <= 256 and rows are 16-byte
 *   aligned: make the three calls then. */
int niw_composite_mse_train(const float* ray, const float* rgb_s, const float* sigma_s, const float* depth_s, int64_t n_rays, int n_samples,
                            const float* image, const int64_t* ray_idx, int n_views, int64_t n_rays_per_view, int64_t hw, int64_t first_ray,
                            double n_norm, float grad_scale, float* rgb, float* depth, float* opacity, float* prob, float* resid, float* d_rgb,
                            float* d_rgb_s, float* d_sigma_s, float* d_ray, niw_stream_t stream);

/* loss[0] = sum(resid^2) / n_norm over resid [n_rays,3] (overwritten; one workgroup, fixed order = niw_mse_fwd_bwd's). */
int niw_mse_from_residuals(const float* resid, int64_t n_rays, double n_norm, float* loss, niw_stream_t stream);

/* ------------------------------------------------------------------ sampling
 * Graph.sample_depth (model/nerf.py:334-344).  u [n_rays,S] stratified draws or NULL (0.5).
 * The range travels as doubles: the reference forms (depth_max - depth_min) from the yaml's Python floats in double precision and
 * only then multiplies the fp32 tensor by it (5.2 - 1.2 is 4.0 that way, 3.9999998 when both ends are rounded to fp32 first); a range
 * read from an fp32 tensor (DTU: var.depth_range) is exact in double either way. */
int niw_sample_stratified(const float* u, int64_t n_rays, int n_samples, double depth_min, double depth_max,
                          int inverse, float* depth, niw_stream_t stream);

/* The same with the stratified draw made inside the kernel (the reference draws torch.rand on the device at this point,
 * model/nerf.py:337): u = Philox4x32-10(key = seed, counter = (sample index / 4, draw)) -> 24-bit uniform in [0,1).  `draw` numbers
 * the draw (the training iteration: a resumed run continues the same stream); a non-NULL draw_dev (one uint64 in device memory)
 * overrides it at run time (captured-graph replays).  u_out (optional, [n_rays,S]) receives the draws, so that a checker can
 * reproduce the depths with niw_sample_stratified / the oracle. */
int niw_sample_stratified_rng(uint64_t seed, uint64_t draw, const uint64_t* draw_dev, int64_t n_rays, int n_samples,
                              double depth_min, double depth_max, int inverse, float* depth, float* u_out, niw_stream_t stream);

/* Density noise of the train-mode field forward (reference model/nerf.py:428-429: `density += torch.randn_like(density) * density_noise_reg`)
 * drawn on the device: out[i] = scale * z_i, z ~ N(0, 1) by Box-Muller over the Philox4x32-10 stream of niw_sample_stratified_rng (counter =
 * (i / 4, draw), key = seed; draw_dev as there).  The result is what niw_mlp_fwd takes as `noise`.  A pure function of (seed, draw, i). */
int niw_normal_rng(uint64_t seed, uint64_t draw, const uint64_t* draw_dev, int64_t n, float scale, float* out, niw_stream_t stream);

/* Graph.sample_depth_from_pdf (model/nerf.py:346-365) followed by the cat + ascending sort of
 * Graph.render (model/nerf.py:313-315).  pdf [n_rays,S], depth_coarse [n_rays,S];
 * unif [Sf] = mid-points of linspace(0,1,Sf+1) (nerf.py:352-353) and bins [S+1] =
 * linspace(depth_min,depth_max,S+1) (nerf.py:356) are the two per-config constant tables, built
 * once by the host mirror with the same torch.linspace calls as the reference.
 * Outputs: depth_fine [n_rays,Sf] (may be NULL), depth_merged [n_rays,S+Sf] ascending.
 * S+Sf <= 1024. */
int niw_sample_pdf_merge(const float* pdf, const float* depth_coarse, const float* unif, const float* bins,
                         int64_t n_rays, int n_samples, int n_fine,
                         float* depth_fine, float* depth_merged, niw_stream_t stream);

/* ------------------------------------------------------------------ ray generation
 * mode 0: camera.get_unwarped_center_and_ray (camera.py:359-390): out_a = center (zeros, or
 *         the camera centre under pose_init), out_b = grid point K^-1 [x+.5, y+.5, 1] (moved to
 *         world by pose_init when given).
 * mode 1: camera.get_center_and_ray (camera.py:419-443): out_a = centre, out_b = ray
 *         (pose is world->camera and is inverted as camera.py:89-95 does).
 * intr [B,3,3]; pose [B,3,4] or NULL; ray_idx [R] int64 pixel ids (y*W+x), or NULL: the R consecutive pixels
 * first_pixel .. first_pixel+R-1 (a slice of a full-image render, model/nerf.py:321-332; 0 and R = H*W: the whole image);
 * outputs [B,R,3]. */
int niw_raygen(const float* intr, const float* pose, const int64_t* ray_idx, int64_t first_pixel, int n_views,
               int64_t n_rays_per_view, int H, int W, int mode, float* out_a, float* out_b, niw_stream_t stream);

/* The random pixel subset of a training step: `torch.randperm(H*W)[:n]` of the reference (model/nerf_inn_llff.py:510,
 * model/nerf.py:256) as ONE launch without a sort: out[i] = P(first + i * stride), i < n, where P is a keyed pseudo-random
 * PERMUTATION of [0, n_pixels) (4-round Feistel network on ceil(log2 n_pixels) bits, cycle-walked into range), so the
 * indices are distinct and every pixel is equally likely, as with randperm.  The key is hash(seed, draw); `draw` comes by
 * value or, when draw_dev != NULL, from that device word (a captured HIP graph then draws a fresh subset on every replay
 * after the host bumped the word).  first / stride select a rank's share idx[rank::world] of the common permutation. */
int niw_draw_ray_idx(int64_t n_pixels, int64_t n, uint64_t seed, uint64_t draw, const uint64_t* draw_dev, int64_t first,
                     int64_t stride, int64_t* out, niw_stream_t stream);

/* camera.convert_NDC (camera.py:523-540); center, ray [B,R,3] in/out buffers distinct. */
int niw_convert_ndc(const float* center, const float* ray, const float* intr, int n_views, int64_t n_rays_per_view,
                    float near, float* center_ndc, float* ray_ndc, niw_stream_t stream);

/* Reverse pass of niw_convert_ndc (round 6): gradients of the camera-frame centre / ray from those of the NDC centre / ray (either may be
 * NULL = zero) -- autograd of camera.py:523-540 for rays that carry a gradient (warped rays in training, a refined pose at test time).
 * center, ray: the FORWARD's inputs.  d_center / d_ray [n_views,n_rays_per_view,3] are overwritten. */
int niw_convert_ndc_bwd(const float* center, const float* ray, const float* intr, int n_views, int64_t n_rays_per_view, float near,
                        const float* d_center_ndc, const float* d_ray_ndc, float* d_center, float* d_ray, niw_stream_t stream);

/* ------------------------------------------------------------------ gradient-free render of a pixel range, one call
 * Graph.render under torch.no_grad() (model/nerf.py:293-319): rays of the pixels first_pixel .. first_pixel+n_pixels-1 of every
 * view (camera.get_center_and_ray, camera.py:419-443) -> convert_NDC when `ndc` (camera.py:523-540) -> sample_depth
 * (nerf.py:334-344) -> NeRF.forward_samples + composite (nerf.py:449-474) -> when n_fine > 0: sample_depth_from_pdf, cat, sort,
 * the fine network and its composite (nerf.py:310-318).  With first_pixel = 0 and n_pixels = H*W it is the whole
 * Graph.render_by_slices loop (nerf.py:321-332) in one call; the slice size of the reference only bounded its memory and has no
 * effect on any value.  Results are bit-identical to calling the stages above one by one.
 *
 * niw_render_desc is a HOST struct (the only host pointer of this call besides the two band arrays inside it); every pointer
 * member is a device pointer unless stated. */
typedef struct niw_render_desc {
    const float* intr;          /* [n_views,3,3] */
    const float* pose;          /* [n_views,3,4] world->camera */
    int32_t n_views, H, W;
    int32_t ndc;                /* != 0: LLFF normalised device coordinates */
    int64_t first_pixel, n_pixels;
    float ndc_near;
    double depth_min, depth_max; /* opt.nerf.depth.range (DTU: var.depth_range[0]); doubles: see niw_sample_stratified */
    int32_t inverse_depth;      /* opt.nerf.depth.param == "inverse" */
    int32_t n_samples, n_fine;  /* n_fine = 0: single pass */
    int32_t density_activ;      /* enum niw_density_activ */
    int32_t precision;          /* enum niw_precision: arithmetic of the field MLP; packed / packed_fine must be images of that class */
    int32_t has_bg;             /* opt.nerf.setbg_opaque */
    float bg;
    const float* u;             /* [n_views*n_pixels, n_samples] stratified draws, or NULL: interval mid-points */
    const float* unif;          /* inverse-CDF tables of niw_sample_pdf_merge (n_fine > 0) */
    const float* bins;
    const float* packed;        /* packed weights of the (coarse) network, niw_mlp_pack_weights */
    const float* packed_fine;   /* fine network (n_fine > 0) */
    const float* band_w3d;      /* HOST [10] or NULL */
    const float* band_wview;    /* HOST [4] or NULL */
    const float* band_dev;      /* device [14] or NULL (see niw_mlp_fwd) */
    const float* band_w3d_fine; /* the same three for the fine network (each may be NULL) */
    const float* band_wview_fine;
    const float* band_dev_fine;
} niw_render_desc;

int64_t niw_render_fwd_workspace_floats(int n_views, int64_t n_pixels, int n_samples, int n_fine);
/* rgb [n_views,n_pixels,3], depth, opacity [n_views,n_pixels]; the *_fine outputs are written when n_fine > 0 (else may be NULL).
 * workspace: niw_render_fwd_workspace_floats() floats, 16-byte aligned. */
int niw_render_fwd(const niw_render_desc* desc, float* workspace, float* rgb, float* depth, float* opacity,
                   float* rgb_fine, float* depth_fine, float* opacity_fine, niw_stream_t stream);

/* ------------------------------------------------------------------ NVP invertible warp
 * DeformNetwork.forward / .inverse (model/nvp/nvp_ndr.py:365-468, 471-567), per-point part.
 * The per-view / per-parameter preprocessing (weight norm g*v/|v| nvp_ndr.py:291-292, code
 * projection lin_c(code)+code :381, and the latent half of the first layers) is folded by the
 * host mirror into:
 *   w_emb   [3 blocks][ part a: 128 rows of 28 | part b: 128 rows of 16 ]   effective first-layer weights: 26 / 13 embedding
 *           columns per row, padded to 16-byte multiples (pad columns: written as zero, never read)
 *   view_b  [n_views][3][2][128]                            W[:,emb:] . code_b + bias   (per view)
 *   w_head  [3][ a: 1x128 + 1 | b: 3x128 + 3 ]              second-layer weights and biases
 * pts [n_views, n_pts, 3]; chan_w[6] (host) per-band window or NULL; index_window[6] (host) or NULL: the
 * reference's dim-1 slicing of model/nvp/embedder.py:47 on 4-D input -- window value i scales ALL
 * channels of the points (2i+1)d .. (2i+3)d-1 (d = 2 for the 2-D embedding, 1 for the 1-D one), passed by value
 * with the launch; window_dev: DEVICE array of 12 floats {chan_w[6], index_window[6]} or NULL -- when given it overrides the
 * two host arrays and is read by the kernel at run time (HIP-graph replays), with use_index_window saying whether the index
 * window applies; pt_scale_a / pt_scale_b [n_pts] (device) optional additional per-point scales.
 * inverse != 0 evaluates .inverse.  xin_save [n_views,n_pts,3,3] or NULL: the input point of each of the three coupling blocks,
 * kept for niw_warp_bwd (forward warp only), which otherwise recomputes them -- a third of its work. */
#define NIW_WARP_WEMB_FLOATS (3 * (128 * 28 + 128 * 16))
#define NIW_WARP_WHEAD_FLOATS (3 * (128 + 1 + 3 * 128 + 3))
#define NIW_WARP_PARAM_FLOATS 165900    /* DeformNetwork parameters, flat in parameters() order (see niw_warp_prep.hip) */

/* Operand preparation of the warp and its backward (reference nvp_ndr.py:291-292 weight norm, :381 code
 * projection, :416-420 / :433-437 latent half of the first layers).
 *   params [NIW_WARP_PARAM_FLOATS] flat parameters; code [n_views,128] (warp_latent.weight)
 *   -> w_emb [NIW_WARP_WEMB_FLOATS], view_b [n_views,3,2,128], w_head [NIW_WARP_WHEAD_FLOATS]
 * backward: d_w_emb, d_view_b, d_w_head -> d_params [NIW_WARP_PARAM_FLOATS] (overwritten), d_code [n_views,128].
 * workspace: caller-allocated scratch of niw_warp_prep_{fwd,bwd}_workspace_floats(n_views) floats (contents need not
 * survive between the two calls).  n_views <= 64. */
int64_t niw_warp_prep_fwd_workspace_floats(int n_views);
int64_t niw_warp_prep_bwd_workspace_floats(int n_views);
int niw_warp_prep_fwd(const float* params, const float* code, int n_views, float* workspace, float* w_emb, float* view_b,
                      float* w_head, niw_stream_t stream);
int niw_warp_prep_bwd(const float* params, const float* code, int n_views, const float* d_w_emb, const float* d_view_b,
                      const float* d_w_head, float* workspace, float* d_params, float* d_code, niw_stream_t stream);
int niw_warp_fwd(const float* w_emb, const float* view_b, const float* w_head, const float* pts,
                 int n_views, int64_t n_pts, const float* chan_w, const float* index_window, const float* window_dev,
                 int use_index_window, const float* pt_scale_a, const float* pt_scale_b, int inverse, float* out,
                 float* xin_save, niw_stream_t stream);

/* Backward of the forward warp.  d_out [n_views,n_pts,3] -> d_w_emb, d_view_b, d_w_head (same
 * shapes as the inputs, overwritten) and d_pts [n_views,n_pts,3] (may be NULL).
 * xin_saved: what niw_warp_fwd wrote to xin_save for the same operands, or NULL.
 * workspace: niw_warp_bwd_workspace_floats() floats of scratch.  n_views <= 64 per call. */
int64_t niw_warp_bwd_workspace_floats(int n_views, int64_t n_pts);
int niw_warp_bwd(const float* w_emb, const float* view_b, const float* w_head, const float* pts,
                 int n_views, int64_t n_pts, const float* chan_w, const float* index_window, const float* window_dev,
                 int use_index_window, const float* pt_scale_a, const float* pt_scale_b, const float* xin_saved, const float* d_out,
                 float* workspace, float* d_w_emb, float* d_view_b, float* d_w_head, float* d_pts, niw_stream_t stream);

/* ------------------------------------------------------------------ global-alignment loss
 * Rotation of the rigid registration (Kabsch with reflection fix) behind `roma.rigid_points_registration`
 * (reference model/nerf_inn_llff.py:569, model/pose_models/inn.py:100): for n moment matrices M [n,3,3] = sum (y - ym)(x - xm)^T
 * returns R [n,3,3] = U diag(1,1,det(U V^T)) V^T, plus what the backward needs: Us = U diag(1,1,d) [n,3,3], V [n,3,3],
 * S [n,3] = (s0, s1, d s2).  Backward: dR [n,3,3] -> dM [n,3,3].  No host synchronisation. */
int niw_kabsch_rotation_fwd(const float* M, int n, float* R, float* Us, float* V, float* S, niw_stream_t stream);
int niw_kabsch_rotation_bwd(const float* Us, const float* V, const float* S, const float* dR, int n, float* dM, niw_stream_t stream);

/* The whole alignment term fused (model/nerf_inn_llff.py:563-572, model/nerf_inn_dtu.py:410-414):
 *   niw_align_moments: target (warped points x), source (un-warped points y) [n_views,n_points,3] -> moments [n_views,16] DOUBLE:
 *                      n, sum x[3], sum y[3], sum y x^T[9].  Under ray sharding the ranks all-reduce this buffer.
 *   niw_align_solve:   moments -> poses [n_views,3,4] = [R|t] minimising sum |R x + t - y|^2 (Kabsch with reflection fix).
 *   niw_align_loss:    loss[0] = sum |x - R^T (y - t)|^2 / n_norm (overwritten; n_norm = 3 * n_views * n_points of the GLOBAL
 *                      batch) and d_target [n_views,n_points,3] = 2 (x - R^T (y - t)) / n_norm (may be NULL).  The loss is
 *                      stationary in (R, t), so this IS the total derivative (csrc/niw_align.hip). */
int niw_align_moments(const float* target, const float* source, int n_views, int64_t n_points, double* moments, niw_stream_t stream);
int niw_align_solve(const double* moments, int n_views, float* poses, niw_stream_t stream);
int niw_align_loss(const float* target, const float* source, const float* poses, int n_views, int64_t n_points, double n_norm,
                   float* loss, float* d_target, niw_stream_t stream);

/* ------------------------------------------------------------------ loss and optimizer
 * Graph.compute_loss photometric part + MSE_loss (model/nerf_inn_llff.py:548-559,
 * model/base.py:209-211): gathers image[b,:,ray_idx] ([B,3,H*W] layout), writes
 * loss[0] = mean((rgb - image)^2) * 1 and d_rgb = scale * 2 (rgb - image) / (3*B*R_norm).
 * n_norm: element count used for the mean (= 3*B*R of the GLOBAL batch under ray sharding).
 * first_ray, n_rays: rgb / d_rgb hold the rays [first_ray, first_ray + n_rays) of the flattened (view-major) [B][R] ray list -- one
 * rank's contiguous share under ray sharding; n_rays <= 0 = the whole batch ([B,R,3]).
 * loss[0] is OVERWRITTEN (one workgroup, fixed-order reduction: bit-reproducible, no float atomics, no zero-fill). */
int niw_mse_fwd_bwd(const float* rgb, const float* image, const int64_t* ray_idx, int n_views,
                    int64_t n_rays_per_view, int64_t hw, int64_t first_ray, int64_t n_rays, double n_norm, float grad_scale,
                    float* loss, float* d_rgb, niw_stream_t stream);

/* torch.optim.Adam step (model/nerf.py:34-38 uses it with default betas/eps) on a flat buffer.
 * hyper_dev: DEVICE array {lr / (1 - beta1^step), sqrt(1 - beta2^step)} or NULL; when given it replaces the two values
 * formed from lr / step on the host and is read at run time (HIP-graph replays). */
int niw_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  double lr, double beta1, double beta2, double eps, int step, const float* hyper_dev, niw_stream_t stream);

/* Adam over several flat buffers in ONE launch (the engine's optimizer groups: NeRF, fine NeRF, warp network, latent table;
 * reference model/nerf.py:34-38, model/barf_inn_llff.py:84-104 build one torch.optim.Adam per pair of groups and step them one
 * after the other -- per element the same arithmetic as niw_adam_step).  `groups` is a HOST array of n_groups (<= 8) entries;
 * hyper_dev: DEVICE array [n_groups][2] = {lr / (1 - beta1^step), sqrt(1 - beta2^step)} per group or NULL (then formed from
 * lr / step of each entry on the host).  A group with n = 0 is skipped. */
typedef struct niw_adam_group {
    float* param;             /* device */
    const float* grad;        /* device */
    float* exp_avg;           /* device */
    float* exp_avg_sq;        /* device */
    int64_t n;
    double lr;
    int32_t step;             /* 1-based */
    int32_t reserved;
} niw_adam_group;
int niw_adam_step_multi(const niw_adam_group* groups, int n_groups, double beta1, double beta2, double eps, const float* hyper_dev,
                        niw_stream_t stream);

/* ------------------------------------------------------------------ one INN train iteration, forward and backward, one call
 * The body of the reference's train step between zero_grad and the optimizers (model/nerf_inn_llff.py:493-573 Graph.forward +
 * compute_loss, model/barf_inn_llff.py:305-364 get_pose, model/base.py:130-142 summarize_loss, and the backward pass autograd runs
 * through all of it; the DTU family: model/nerf_inn_dtu.py:371-415, model/pose_models/inn.py:63-102):
 *
 *   pixel draw -> camera-frame grid / centre points per view (-> world by the initial poses, DTU) -> NVP warp of [grid ; centre]
 *   -> rays -> stratified depths -> field MLP -> compositing [-> inverse-CDF resampling, merge, fine field MLP, compositing]
 *   -> photometric loss(es) + rigid registration of the warped onto the un-warped points and the alignment loss
 *   -> loss.all = sum_k 10^w_k loss_k -> backward of every stage -> parameter gradients of the field network(s), the warp
 *   network and the latent table, written to the caller's four buffers (overwritten; the engine's all-reduce / Adam bucket).
 *
 * The entry point only SEQUENCES the library's own stages (the entry points above, same kernels, same launch shapes) on the
 * caller's stream into ONE caller-provided workspace, plus three small kernels of its own for what the reference does as tensor
 * glue (stack / split of the point sets, the sum of the gradient routes into the warp, the weighted loss total).  No allocation,
 * no synchronisation, no host read-back; capturable into a HIP graph (the per-step scalars then come from band_dev / window_dev /
 * draw_dev).  The reference's Python `Graph.forward / compute_loss` interface stays available on the mirror classes (autograd over
 * the per-stage entry points); this call is what the engine runs per iteration.
 *
 * Ray sharding (one process per GPU): the rank handles the whole views [view0, view1) in the warp and the alignment term (it
 * counts the alignment loss of the views [own0, own1)) and the rays [ray_lo, ray_hi) of the flattened view-major
 * [n_views][rays_per_view] list in everything per (ray, sample); unsharded: view0 = own0 = 0, view1 = own1 = n_views, ray_lo = 0,
 * ray_hi = n_views * rays_per_view.  Losses are normalised by the GLOBAL element counts, so per-rank losses and gradients SUM to
 * the unsharded step.
 *
 * niw_train_desc is a HOST struct; pointer members are device pointers unless stated. */
enum niw_train_stage {          /* the call executes the stages [stage_begin, stage_end); a timing harness calls them one by one */
    NIW_STAGE_FRONT = 0,        /* one launch: pixel draw, un-warped points [grid ; centre] per view, stratified depths, weight images */
    NIW_STAGE_WARP_FWD,         /* operand preparation, warp, rays = grid - centre */
    NIW_STAGE_MLP_FWD,          /* coarse field */
    NIW_STAGE_COMPOSITE_FWD,
    NIW_STAGE_RESAMPLE,         /* inverse-CDF depths merged with the coarse ones (n_fine > 0) */
    NIW_STAGE_MLP_FWD_FINE,
    NIW_STAGE_COMPOSITE_FWD_FINE,
    NIW_STAGE_LOSS,             /* photometric loss(es), registration, alignment loss, weighted total */
    NIW_STAGE_COMPOSITE_BWD_FINE,
    NIW_STAGE_MLP_BWD_DX_FINE,
    NIW_STAGE_MLP_BWD_DW_FINE,
    NIW_STAGE_COMPOSITE_BWD,
    NIW_STAGE_MLP_BWD_DX,
    NIW_STAGE_MLP_BWD_DW,
    NIW_STAGE_WARP_BWD,         /* sum of the gradient routes into the warped points, warp backward, operand-preparation backward */
    NIW_STAGE_END
};

typedef struct niw_train_desc {
    /* the resident batch */
    const float* image;         /* [n_views,3,H*W] */
    const float* intr;          /* [n_views,3,3] */
    const float* pose_init;     /* [n_views,3,4] world->camera initial poses (DTU, camera.py:382-384) or NULL (LLFF) */
    int32_t n_views, H, W;
    int32_t view0, view1, own0, own1;      /* ray sharding, see above */
    int32_t stratified;         /* opt.nerf.sample_stratified: Philox draws (niw_sample_stratified_rng); 0: interval mid-points */
    int64_t rays_per_view;      /* of the GLOBAL batch: rand_rays // n_views */
    int64_t ray_lo, ray_hi;
    /* draws: pixel subset niw_draw_ray_idx(pixel_seed, draw), depths niw_sample_stratified_rng(depth_seed, draw) */
    uint64_t pixel_seed, depth_seed, draw;
    const uint64_t* draw_dev;   /* device word overriding `draw` at run time, or NULL */
    /* sampling */
    int32_t n_samples, n_fine;  /* n_fine = 0: no fine pass */
    int32_t inverse_depth;      /* opt.nerf.depth.param == "inverse" */
    int32_t density_activ;      /* enum niw_density_activ */
    double depth_min, depth_max;
    const float* unif;          /* inverse-CDF tables of niw_sample_pdf_merge (n_fine > 0) */
    const float* bins;
    /* field networks: flat parameters in state-dict order (NIW_NERF_PARAM_FLOATS each) */
    const float* nerf_params;
    const float* nerf_fine_params;   /* n_fine > 0 */
    const int32_t* pack_index;  /* table of niw_mlp_pack_index (fp32 images) or NULL (the image is decoded per element) */
    int32_t precision;          /* enum niw_precision */
    int32_t use_index_window;   /* with window_dev: whether the index window applies (niw_warp_fwd) */
    const float* band_w3d;      /* HOST [10] or NULL: c2f band weights of both networks (they share `progress`) */
    const float* band_wview;    /* HOST [4] or NULL */
    const float* band_dev;      /* device [14] or NULL */
    /* warp */
    const float* warp_params;   /* [NIW_WARP_PARAM_FLOATS] */
    const float* latent;        /* [n_views,128] the whole table (rows view0..view1 are used) */
    const float* chan_w;        /* HOST [6] or NULL */
    const float* index_window;  /* HOST [6] or NULL */
    const float* window_dev;    /* device [12] or NULL */
    /* loss: weights 10^w of model/base.py:130-142; a NEGATIVE weight = the term is absent (yaml weight `null`) */
    float w_render, w_render_fine, w_align;
    int32_t always_register;    /* DTU: the registration runs (and refreshes `poses`) also without the alignment term (inn.py:96-102) */
    double mse_norm;            /* element count of the photometric mean: 3 * n_views * rays_per_view of the GLOBAL batch */
    /* outputs */
    float* loss;                /* [4] = {render, render_fine, global_alignment, all}; absent terms 0 */
    float* d_nerf;              /* [NIW_NERF_PARAM_FLOATS] */
    float* d_nerf_fine;         /* n_fine > 0 */
    float* d_warp;              /* [NIW_WARP_PARAM_FLOATS] */
    float* d_latent;            /* [n_views,128]; rows outside [view0, view1) are zeroed */
    float* poses;               /* [n_views,12] registered [R|t] per view (global_rigid / pose_global); rows view0..view1 refreshed; or NULL */
    float* rgb;                 /* optional [n_rays,3] rendered colours of the share (NULL: kept in the workspace only) */
    float* rgb_fine;            /* optional */
    /* != 0 (and the call spans all stages): the small stages that do not depend on each other run on a second, library-owned
     * stream beside the field-MLP kernels -- the registration and alignment loss beside the field forward, the whole warp backward
     * beside the coarse network's dW GEMMs -- forked from and joined back into `stream`
     * with events (capturable: the branches become parallel branches of a HIP graph).  Same kernels, same numbers.  0: every
     * launch on `stream`, in stage order. */
    int32_t overlap;
    int32_t reserved;
    /* Optional hipEvent_t (as void*), or NULL: recorded on `stream` right behind the launch that completes d_nerf_fine -- the first
     * optimizer group whose gradients are final, a third of the way into the backward.  A data-parallel caller makes its communication
     * stream wait for it and exchanges that group while the coarse network's and the warp's backward are still running (engine.py). */
    void* fine_grads_ready;
    /* ---- round 6: the VANILLA model's iteration (reference model/nerf.py:251-288: Graph.forward + compute_loss in train mode on the
     * ground-truth poses; BASELINE configs[0], options/nerf_llff_repr.yaml).  warp_params == NULL selects it: rays of the cameras
     * `pose_init` (then REQUIRED: world->camera [n_views,3,4]; niw_raygen mode 1), no warp, no registration / alignment term
     * (w_align must be negative), no ray gradients (the dX chain stops above layer 0, SURVEY section 8(a) "Gradient routes");
     * latent / d_warp / d_latent / poses are not touched and may be NULL.  With a warp, `density_noise` and `ndc` apply likewise:
     * the warped rays are re-parametrised behind the warp and the summed gradient routes go back through niw_convert_ndc_bwd before
     * they reach the warp (model/nerf_inn_llff.py:627-641 under autograd). */
    float density_noise;        /* opt.nerf.density_noise_reg: N(0, density_noise^2) added to the raw density of every sample in BOTH passes
                                   (model/nerf.py:428-429), drawn by niw_normal_rng(noise_seed [+ 1 for the fine pass], draw); 0: none */
    int32_t ndc;                /* opt.camera.ndc: rays re-parametrised by niw_convert_ndc (camera.py:523-540) */
    uint64_t noise_seed;
    float ndc_near;             /* near plane of the NDC re-parametrisation (the reference passes 1) */
    int32_t has_bg;             /* opt.nerf.setbg_opaque: rgb += bg (1 - opacity) in both compositing passes (model/nerf.py:470-472); the passes
                                   then run as niw_composite_fwd / niw_mse_fwd_bwd / niw_composite_bwd (the one-launch form has no background) */
    float bg;                   /* opt.data.bgcolor */
    int32_t reserved3;
} niw_train_desc;

/* Optional: create the library's own streams (niw_train_desc.overlap; niw_mlp_bwd_dw's second stream) for the current device NOW instead
 * of at the first call that uses them.  Streams share a small number of hardware queues in creation order; a process that also runs RCCL / other stream users calls
 * this first, so that the second stream does not end up sharing a hardware queue with the stream it is meant to run beside. */
int niw_train_step_prepare(void);
/* floats of the workspace (256-byte aligned); <= 0 with niw_last_error_string() set when the descriptor is not supported */
int64_t niw_train_step_workspace_floats(const niw_train_desc* desc);
int niw_train_step(const niw_train_desc* desc, float* workspace, int stage_begin, int stage_end, niw_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NIW_H_ */
