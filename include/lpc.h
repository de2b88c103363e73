/*
 * lpc.h -- C ABI of the MI355X-native iterative deconvolution engine
 *          ("lpc" = LenslessPiCam hot path).
 *
 * The reference (LCAV/LenslessPiCam) is pure Python and has NO FFI for this path: its
 * "plugin API" is the Python class interface ReconstructionAlgorithm.set_data()/apply()
 * (lensless/recon/recon.py:179-605) implemented by ADMM (lensless/recon/admm.py:24-338),
 * GradientDescent / NesterovGradientDescent / FISTA (lensless/recon/gd.py:62-241) on top
 * of RealFFTConvolve2D (lensless/recon/rfft_convolve.py:26-223).  This header is the
 * boundary a maintainer would bind (ctypes stub in INTEGRATION.md); every entry point names
 * the reference method whose arithmetic it replaces.
 *
 * Conventions
 *  - plain C, no torch types.  Every pointer named dev_* is a DEVICE pointer (HBM) that the
 *    library only borrows for the duration of the call; images are lpc_real, channels-last,
 *    exactly the reference's layouts: psf (D,H,W,C), data (B,H,W,C), image estimate
 *    (B,D,H,W,C) for the gradient-descent family and (B,D,Hp,Wp,C) for ADMM (which iterates
 *    on the padded frame, admm.py:101 pad=False).
 *  - every function returns 0 on success, non-zero on failure; lpc_last_error() then holds
 *    a message (thread-local).  Nothing throws across the boundary.
 *  - one handle <-> one stream at a time; a handle is not thread-safe, distinct handles are
 *    independent.  stream is a hipStream_t passed as void* (NULL = default stream).
 *  - the library REQUIRES a HIP device; there is no CPU fallback.
 */
#ifndef LPC_H_
#define LPC_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lpc_engine* lpc_handle;

/* The library is built twice from the same sources: liblpc.so computes in float32 (the reference's
 * default dtype), liblpc_f64.so (-DLPC_DOUBLE) in float64 (dtype="float64", lensless/utils/io.py:645-674).
 * Every image / state buffer of a library is an array of its lpc_real. */
#ifdef LPC_DOUBLE
typedef double lpc_real;
#else
typedef float lpc_real;
#endif

enum lpc_algo {
  LPC_ALGO_CONV = 0,     /* operator only: RealFFTConvolve2D            rfft_convolve.py:26  */
  LPC_ALGO_ADMM = 1,     /* ADMM, TV prior + non-negativity             admm.py:24           */
  LPC_ALGO_GD = 2,       /* projected gradient descent                  gd.py:62             */
  LPC_ALGO_NESTEROV = 3, /* + Nesterov momentum                         gd.py:143            */
  LPC_ALGO_FISTA = 4     /* + FISTA momentum                            gd.py:191            */
};

enum lpc_norm { LPC_NORM_BACKWARD = 0, LPC_NORM_ORTHO = 1, LPC_NORM_FORWARD = 2 };

typedef struct lpc_config {
  int algo;                /* enum lpc_algo                                                   */
  int height, width;       /* un-padded H, W of PSF and data                                  */
  int channels;            /* C: 1 or 3                          recon.py:255-256             */
  int depth;               /* D >= 1.  ADMM with D > 1 = D independent planes sharing the     */
                           /* measurement (the reference refuses)  */
  int batch;               /* B >= 1 measurements sharing the PSF (additive to the reference) */
  int norm;                /* enum lpc_norm of the PSF spectrum  rfft_convolve.py:27,121      */
  int pad;                 /* LPC_ALGO_CONV only: operator pads/crops (1) or works on the     */
                           /* padded frame (0)                   rfft_convolve.py:133-176     */
  double mu1, mu2, mu3, tau; /* ADMM                              admm.py:39-42                */
  double lip_fact;         /* GD family step factor              gd.py:67,107-112             */
  double nesterov_mu, nesterov_p; /* gd.py:153,178-181                                         */
  double fista_tk;         /* gd.py:200,227-233                                               */
  const char* options;     /* NULL, or "key=value,key=value": launch-plan choices (below).  Read  */
                           /* by lpc_create only; the string is not kept.                         */
} lpc_config;

/* Launch-plan options (lpc_config.options; the environment variable LPC_OPTIONS, same syntax, supplies process-wide
 * defaults that a handle's own string overrides).  None changes a result beyond float round-off; none is needed in
 * production -- they exist so that tests and A/B measurements can select a launch plan without touching the process
 * environment.  lpc_plan_info() reports the plan a handle ended up with.
 * Which kernels a handle gets
 *   no_static=1        run-time FFT plans only (no plan module is looked for)
 *   jit=0              never compile a plan module: use what is on disk (default 1: a frame shape whose module is
 *                      missing is compiled with hipcc on first use, ~3 s, and kept in <libdir>/modules or ~/.cache)
 *   jit_min_points=N   padded frames with fewer than N points keep the run-time plans (default 65536)
 *   module_dir=PATH    first place modules are looked for / written;  compiler=PATH  the hipcc to use
 *   module_max=N module_loaded_max=N    plan-module files kept per directory this library writes to (256, least
 *                      recently used removed first); modules kept loaded once no handle uses them (64)
 * Forcing a plan the chooser takes at other sizes (how the tests run every kernel family on small frames)
 *   rows_half=0|1      paired rows / one real row per half-length transform (default: by width)
 *   col_t=N tile_budget=N split_n2=N passa_t=N     column tiling: image columns per tile, LDS points per tile, forced
 *                      length of the fused middle transform, columns per pass-A tile
 *   mid_seq=0|1        ADMM single-pass middle one spectrum at a time / side by side (default: by batch size);
 *   mid_pre=0          ... one at a time: the second tile's loads behind the first transform (default: both up front)
 *   prow_nt128=0|1     short paired rows on 256 / 128 threads (default: by batch size)
 *   g_plane=0|1        ADMM middles read |PsiT Psi| as a row term + a column term when it separates (0) / from its plane
 *                      (1); default: the terms when the plane exceeds 8 MB
 *   mid_swz=0|1        side-by-side middle: adjacent half-line column tiles on one XCD (default: when a tile row is < 128
 *                      bytes and the spectra are not in pair lines)
 *   spec_lay=0         ADMM work spectra as plain rows (default: paired rows + a single-pass middle of 8-column tiles keep
 *                      them in PAIR LINES -- rows 2p, 2p + 1 x 8 columns share one 128-byte line, lpc_kernels.h: spec_col)
 *   row_rad=16.16.8    radices of the compile-time row plan instead of the chooser's (one module)
 * The structure of an ADMM iteration (each setting is an older, complete form of the same arithmetic)
 *   xi_full=1 hv_full=1   without the sensor-window structure of xi / of the H V row transforms
 *   mid_pc=0           sequential middle on pair-line spectra: H, |G| and the phase factors loaded and combined per element
 *                      (default 1: precombined once per PSF and step sizes into one 16-byte + one 4-byte load per element)
 *   k1_half=0          duals stored plain between the iterations of one call (default 1: half-applied, the tiled kernel
 *                      then does not read V_old: 9R -> 8R)
 *   k1_rows=0          keep the tiled TV / W kernel (default 1: paired rows of one or two quads per lane -- padded widths up
 *                      to 2048 -- take that half of the image-domain work as well: three launches per iteration, r_sp never
 *                      stored; rows of up to 1024 points then run on 256 lanes at every batch size)
 *   k1_group=N         ... on launches of more than 8192 row blocks: runs of N consecutive blocks per XCD (default 16;
 *                      0: launch order)
 * Block orders (permutations)
 *   rev_order=BITS     which ADMM kernels walk their grids backwards (1 tiled kernel, 2 / 4 forward / inverse pass A,
 *                      8 LDS middle; default 9);  gd_rev=BITS (gradient-descent family / operator: 1 residual rows, 2
 *                      update rows, 4 register middle; default: all three when a work spectrum exceeds the 256-MB
 *                      memory-side cache)
 * Gradient-descent family
 *   gd_v2=0            the first form of the two fused row kernels (default 1: the second form, lpc_gd_v2_kernels.h,
 *                      wherever the row plan has one radix and the window offset / width are even)
 *   gd_no_fuse_fwd=1   update without the next iteration's forward rows
 * (Variants that were built, measured and not adopted -- persistent row workgroups with LDS-DMA prefetch, whole-column
 * single-launch transforms, the xor / i + i/16 LDS layouts, staggered first generations, padded row pitches, the half-line
 * tile pairing of the sequential middle -- left the sources in round 6; their measurements are in profiles/HISTORY.md.)
 * Path-valued options (module_dir, compiler) are unescaped before use: %XX (two hex digits) stands for the byte XX, so the
 * separators of this string can appear in a path -- "%2C" is ',', "%3B" ';', "%20" ' ' -- and a literal '%' that is
 * followed by two hex digits must itself be written "%25".
 * An unknown key makes lpc_create fail. */

/* ---- life cycle ------------------------------------------------------------------ */
/* replaces the constructors (recon.py:203-329, admm.py:35-135, gd.py:67-92) minus the PSF */
int lpc_create(const lpc_config* cfg, lpc_handle* out);
int lpc_destroy(lpc_handle h);
/* Plan modules.  The compile-time-plan kernels of one frame shape live in a small shared object next to the library
 * (lenslesspicam_amd/csrc/lpc_plan.h); lpc_create loads it, compiling it first when missing.  lpc_plan_module reports
 * the key of the module lpc_create(cfg) would use ("" = none: run-time plans) and, with build != 0, compiles it now if
 * it is not on disk -- no device needed, so a build step can pre-build the shapes it knows (build.py does, for
 * BASELINE.json's).  key_buf may be NULL. */
int lpc_plan_module(const lpc_config* cfg, int build, char* key_buf, size_t n);
const char* lpc_last_error(void);
const char* lpc_backend(void);        /* "hip-gfx950" for the product library */
const char* lpc_real_name(void);      /* "float32" or "float64": the arithmetic type of this build */

/* padded frame chosen by the engine: next 5-smooth length >= 2*dim-1 (rfft_convolve.py:110-117) */
int lpc_padded_shape(lpc_handle h, int* Hp, int* Wp, int* start_h, int* start_w);

/* ---- operator: RealFFTConvolve2D.set_psf / convolve / deconvolve ------------------- */
/* dev_psf: (D,H,W,C).  Computes the PSF spectrum (and for ADMM R_divmat, for the GD family
 * the step alpha and the default initial estimate).  rfft_convolve.py:102-131,
 * admm.py:186-193, gd.py:94-126.  Implies lpc_reset(). */
int lpc_set_psf(lpc_handle h, const lpc_real* dev_psf, void* stream);

/* out = ifftshift(irfft2(rfft2(pad?(x)) * H or conj(H))) cropped if cfg.pad.
 * x: (n, D, Hx, Wx, x_channels), out: (n, D, Hx, Wx, C) with (Hx,Wx) = (H,W) if cfg.pad else (Hp,Wp);
 * n <= cfg.batch.  x_channels is C, or 1 (a grayscale input against an RGB PSF broadcasts over the channels like
 * the reference's `vpad[...] = v`, rfft_convolve.py:96-99); anything else is refused (it would be read out of bounds).
 * adjoint = 0: convolve (rfft_convolve.py:133-176); 1: deconvolve (:178-223). */
int lpc_convolve(lpc_handle h, const lpc_real* dev_x, lpc_real* dev_out, int n, int x_channels, int adjoint,
                 void* stream);

/* convolve / deconvolve with return_fft=True (rfft_convolve.py:148-150,161-163,193-195,206-208): the spectrum
 * rfft2(pad?(x)) * H (adjoint: * conj(H)) the reference returns instead of transforming back.  dev_out: complex values
 * as (re, im) pairs of lpc_real, (n, D, Hp, Wp/2+1, C), natural frequency order like rfft2's.  LPC_ALGO_CONV handles. */
int lpc_convolve_spectrum(lpc_handle h, const lpc_real* dev_x, lpc_real* dev_out, int n, int x_channels, int adjoint,
                          void* stream);

/* ---- solver state: set_data / _set_initial_estimate / reset ------------------------ */
/* dev_data: (B,H,W,data_channels) with B == cfg.batch; data_channels is C, or 1 (broadcast over the PSF's channels,
 * as `self._convolver._pad(self._data)` does, admm.py:253 / `- self._data`, gd.py:129).  recon.py:352-381 */
int lpc_set_data(lpc_handle h, const lpc_real* dev_data, int data_channels, void* stream);
/* dev_est: image-estimate shape (see top) or NULL to clear.  Takes effect at the next
 * lpc_reset(), like recon.py:383-413. */
int lpc_set_initial_estimate(lpc_handle h, const lpc_real* dev_est, void* stream);
/* admm.py:150-230 / gd.py:94-126,178-181,227-233 */
int lpc_reset(lpc_handle h, void* stream);
/* Nesterov: overrides (p, mu) like NesterovGradientDescent.reset(p, mu) gd.py:178-181;
 * FISTA: overrides tk like FISTA.reset(tk) gd.py:227-233.  Call after lpc_reset. */
int lpc_set_momentum(lpc_handle h, double p, double mu, double tk);

/* Unrolled ADMM (lensless/recon/unrolled_admm.py:133-234): iteration i (counted from the last reset)
 * uses mu1[i], mu2[i], mu3[i], tau[i]; the last entry is held beyond n.  n <= 0 clears the schedule and
 * returns to the constructor's constants. */
int lpc_set_admm_schedule(lpc_handle h, int n, const double* mu1, const double* mu2, const double* mu3,
                          const double* tau);

/* Unrolled FISTA (lensless/recon/unrolled_fista.py:60-106): iteration i uses the step alpha[i*C + c] and
 * the momentum factor coef[i] = (t_i - 1) / t_{i+1}; x_k starts as the initial image.  alpha, coef are
 * HOST arrays of n*C and n floats.  n <= 0 returns to the plain FISTA recursion. */
int lpc_set_fista_schedule(lpc_handle h, int n, const lpc_real* alpha, const lpc_real* coef, void* stream);

/* ---- the hot loop: `for i in range(n_iter): self._update(i)`  recon.py:575-576 ------ */
/* exactly n_iter iterations, asynchronous on `stream`; no early exit exists on this path */
int lpc_iterate(lpc_handle h, int n_iter, void* stream);

/* Plug-and-play hook (gradient-descent family): ONE iteration split where the reference calls
 * `self._form_image()` = `self._proj(self._image_est[, noise_level])` inside `_update` (gd.py:132-140,183-188,
 * 235-241; `proj=` argument gd.py:67, external denoiser gd.py:89-92).  lpc_iterate_begin runs the fused gradient /
 * momentum half and leaves the UNPROJECTED estimate readable as lpc_get_state("image_est"); the caller applies its
 * projection or denoiser to that (B,D,H,W,C) array and passes the result to lpc_iterate_end, which completes the
 * iteration (FISTA: the extrapolation and the t_k recursion).  Mixing with lpc_iterate between the two is an error. */
int lpc_iterate_begin(lpc_handle h, void* stream);
int lpc_iterate_end(lpc_handle h, const lpc_real* dev_projected, void* stream);

/* Plug-and-play ADMM (admm.py:126-133,235-243,266-275,300-311): ONE iteration split at the U-update, where the
 * reference calls the external denoiser.  lpc_admm_pnp_begin writes the denoiser's input as (B,D,Hp,Wp,C):
 * U + eta/mu2 if use_dual, else the image estimate; the caller runs its denoiser and hands U (same shape) to
 * lpc_admm_pnp_end, which does the X / W updates, the spectral image update and the dual updates exactly as the
 * reference's branch is written (U and eta image-shaped, Psi^T = identity; use_dual: r_k = (mu3 W - rho) + mu2 U - eta,
 * else r_k = mu2 U + H^T(mu1 X - xi)).  A handle runs either these or lpc_iterate between two resets, not both.
 * lpc_get_state then serves "U","eta","X","W","xi","rho","forward_out","image_est" as (B,D,Hp,Wp,C). */
int lpc_admm_pnp_begin(lpc_handle h, int use_dual, lpc_real* dev_denoiser_in, void* stream);
int lpc_admm_pnp_end(lpc_handle h, int use_dual, const lpc_real* dev_U, void* stream);

/* ADMM with a caller-supplied sparsifying operator `psi / psi_adj / psi_gram` (admm.py:44-46,104-120): the operator
 * cannot be fused, so the caller keeps U, eta and Psi(V) (their shape is the operator's) and runs Psi, Psi^T and the
 * soft-threshold itself; the engine does everything else of `_update` (admm.py:252-329) in one call:
 *   lpc_set_psi_gram   |psi_gram(padded_shape)| (real, (Hp, Wc), NATURAL frequency order, one plane shared by all
 *                      channels) replaces the finite-difference gram in R_divmat (admm.py:186-190).  Call after
 *                      lpc_set_psf (which restores the default gram).  Synchronises `stream` once: the engine checks
 *                      on the device whether the plane is a row term + a column term (the finite-difference gram is)
 *                      and, for planes above 8 MB, lets its fused middles read the two vectors instead (option g_plane).
 *   lpc_admm_psi_step  dev_psit = Psi^T(mu2 U - eta) as (B,D,Hp,Wp,C): X and W updates, r_k = (mu3 W - rho) + dev_psit +
 *                      H^T(mu1 X - xi), the spectral image update, the xi and rho updates.  The new estimate is
 *                      lpc_get_state("image_est"); the caller then updates Psi(V) and eta (admm.py:302-308).
 * A handle runs either these or lpc_iterate between two resets. */
int lpc_set_psi_gram(lpc_handle h, const lpc_real* dev_gabs, void* stream);
int lpc_admm_psi_step(lpc_handle h, const lpc_real* dev_psit, void* stream);

/* _form_image(): ADMM crop + clamp (admm.py:331-338), GD family projection (gd.py:136-140).
 * dev_out: (B,D,H,W,C).  Like the reference, the ADMM clamp is an in-place side effect on the image
 * estimate: it is visible to the W-update of the following iterations (and to "image_est"). */
int lpc_form_image(lpc_handle h, lpc_real* dev_out, void* stream);

/* inspection (tests, warm starts).  name: "image_est" (solver state shape), and for ADMM
 * "X","xi","rho","forward_out","W" as (B,D,Hp,Wp,C), "U","eta" as (B,D,Hp,Wp,C,2)
 * [values as the reference holds them after the same number of iterations].
 * GD family: "alpha" writes C floats. */
int lpc_get_state(lpc_handle h, const char* name, lpc_real* dev_out, void* stream);

/* ---- evaluation reductions on the device (no host synchronisation of the results) -------------- */
/* ReconstructionAlgorithm.reconstruction_error (recon.py:607-653):
 *   out[b] = sum_{d,h,w,c} (N(H x)[b,d] - y[b])^2 / (D*H*W*C),  N(z) = (z - min z) / max(z - min z) over (H,W,C)
 * (N = identity when normalize == 0).  dev_pred: (B,D,H,W,C), e.g. what lpc_form_image wrote; dev_data:
 * (B,H,W,C) or NULL = the frame given to lpc_set_data; dev_out: B values.  Works on every handle kind. */
int lpc_reconstruction_error(lpc_handle h, const lpc_real* dev_pred, const lpc_real* dev_data, int normalize,
                             lpc_real* dev_out, void* stream);
/* mse() and psnr() of lensless/eval/metric.py:119-172 for n_items image pairs of n values each:
 * dev_out[2*i] = mean((t/max t - e/max e)^2), dev_out[2*i+1] = 10 log10(R^2 / mse) with R = 1 if min t >= 0 else 2
 * (skimage's data-range rule for float images); normalize == 0 skips the division by the maxima.
 * Handle-free and asynchronous on `stream` (scratch comes from the device's stream-ordered memory pool). */
int lpc_image_metrics(const lpc_real* dev_true, const lpc_real* dev_est, long n, int n_items, int normalize,
                      lpc_real* dev_out, void* stream);

/* ---- raw-frame preparation on the device: the step in front of lpc_set_psf / lpc_set_data -------------
 * Restates lensless/utils/io.py load_image (:157-196), load_psf (:283-375) and their chaining in load_data
 * (:462-552), minus file decoding, Bayer demosaicing and resizing.  Handle-free, asynchronous on `stream` (scratch
 * comes from the device's stream-ordered memory pool); results stay on the device. */
enum lpc_raw_type { LPC_RAW_U8 = 0, LPC_RAW_U16 = 1, LPC_RAW_F32 = 2, LPC_RAW_F64 = 3 };
typedef struct lpc_prep_config {
  int raw_type;          /* enum lpc_raw_type of the raw buffer                                        */
  int height, width;     /* of the raw buffer == of the result (no resizing)                            */
  int channels;          /* of the raw buffer, channels-last: 1 or 3                                    */
  int flip_ud, flip_lr;  /* io.py:160-166 (load_image's flip = both)                                    */
  int bgr_input;         /* 3 channels arrive as BGR (io.py:153-154)                                    */
  int gray;              /* rgb2gray AFTER normalisation, weights 0.299/0.587/0.114 (io.py:550-552)      */
  int normalize;         /* frames: divide by the frame's maximum (io.py:196-197)                        */
  int single_psf;        /* PSF: sum the colour channels into one (io.py:355-364)                        */
  int out_channels;      /* PSF with single_psf: replicate it over 1 or 3 channels (io.py:553-559)       */
  int bg_pix0, bg_pix1;  /* PSF: background level = mean of [p0:p1, p0:p1] per channel (io.py:331-350);  */
                         /* p1 <= p0: no background estimation (bg_pix=None)                             */
} lpc_prep_config;
/* n raw frames (n,H,W,C) -> dev_out (n,H,W,C') lpc_real, C' = 1 if gray else C.  dev_bg: C background levels on the
 * device (what lpc_preprocess_psf wrote: fractions of full scale, re-scaled by get_max_val of each integer frame,
 * image.py:251-278; values > 1 are taken as pixel units) or NULL.  Background removal clips at 0 (io.py:175). */
int lpc_preprocess_frames(const lpc_prep_config* cfg, const void* dev_raw, int n, const lpc_real* dev_bg,
                          lpc_real* dev_out, void* stream);
/* raw PSF stack (D,H,W,C) -> dev_psf_out (D,H,W,C') with unit l2 norm; dev_bg_out (may be NULL): the C background
 * levels divided by the stack's full-scale value (io.py:368). */
int lpc_preprocess_psf(const lpc_prep_config* cfg, const void* dev_raw, int depth, lpc_real* dev_psf_out,
                       lpc_real* dev_bg_out, void* stream);

/* resize (lensless/utils/image.py:28-80, the torch branch): anti-aliased bilinear resampling of n channels-last images
 * (n,H,W,C) -> (n,Hout,Wout,C), last axis first, result clipped to the input's [min, max].  This is what
 * load_psf(downsample=...) / load_data's "resize the frame to the PSF" do through torchvision's
 * Resize(size, antialias=True) = torch.nn.functional.interpolate(mode="bilinear", antialias=True).  Handle-free,
 * asynchronous on `stream`. */
int lpc_resize_aa(const lpc_real* dev_in, int n, int H, int W, int C, int Hout, int Wout, lpc_real* dev_out,
                  void* stream);

/* ---- measurement support (bench.py roofline leg) ----------------------------------- */
enum lpc_kernel_id {
  LPC_K_SPATIAL = 0,   /* fused prox/update kernel (ADMM) / fused update (GD family)  */
  LPC_K_ROW_FWD = 1,
  LPC_K_COL_A_FWD = 2,
  LPC_K_COL_MID = 3,
  LPC_K_COL_A_INV = 4,
  LPC_K_ROW_INV = 5,
  LPC_K_COUNT = 6
};
/* on = 1: every launch of the hot loop is bracketed by hipEvents on its stream; on = 2 << k (or a sum of such terms):
 * only the launches of kernel id k -- two event records per launch cost a 12-MP iteration 1.4 % (ADMM, 6 launches) to
 * 3.4 % (FISTA, 8 launches), so a timed region brackets the one kernel it reports; on = 0: off.  Resets the counts. */
int lpc_profile_enable(lpc_handle h, int on);
/* average milliseconds per launch and launch counts since the last enable; arrays of LPC_K_COUNT */
int lpc_profile_read(lpc_handle h, double* avg_ms, long* launches);
/* algorithmic HBM bytes one launch of kernel k moves */
int lpc_kernel_bytes(lpc_handle h, int kernel_id, double* bytes);
/* SURVEY.md section 8(d)'s algorithmic bytes of ONE solver iteration of all frames of the handle (the yardstick the
 * whole iteration is scored against): ADMM 19R + R0 + 13.5S, Nesterov / FISTA 8R0 + 14S, vanilla GD 6R0 + 14S, with
 * R = padded real array, S = half spectrum, R0 = un-padded array (all planes).  Independent of how many passes the
 * engine really makes. */
int lpc_model_bytes(lpc_handle h, double* bytes);
/* one-line description of the launch plan the handle chose (row scheme, column split, which passes run on compile-time
 * plans, whether the ADMM image-domain kernel is fused into the rows) -- for logs and tests */
int lpc_plan_info(lpc_handle h, char* buf, size_t n);
/* bytes of HBM the handle owns */
int lpc_workspace_bytes(lpc_handle h, size_t* bytes);

#ifdef __cplusplus
}
#endif
#endif /* LPC_H_ */
