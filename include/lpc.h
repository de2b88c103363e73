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
} lpc_config;

/* ---- life cycle ------------------------------------------------------------------ */
/* replaces the constructors (recon.py:203-329, admm.py:35-135, gd.py:67-92) minus the PSF */
int lpc_create(const lpc_config* cfg, lpc_handle* out);
int lpc_destroy(lpc_handle h);
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
 * x, out: (n, D, Hx, Wx, C) with (Hx,Wx) = (H,W) if cfg.pad else (Hp,Wp); n <= cfg.batch.
 * adjoint = 0: convolve (rfft_convolve.py:133-176); 1: deconvolve (:178-223). */
int lpc_convolve(lpc_handle h, const lpc_real* dev_x, lpc_real* dev_out, int n, int adjoint, void* stream);

/* ---- solver state: set_data / _set_initial_estimate / reset ------------------------ */
/* dev_data: (B,H,W,C) with B == cfg.batch.  recon.py:352-381 */
int lpc_set_data(lpc_handle h, const lpc_real* dev_data, void* stream);
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

/* _form_image(): ADMM crop + clamp (admm.py:331-338), GD family projection (gd.py:136-140).
 * dev_out: (B,D,H,W,C).  Like the reference, the ADMM clamp is an in-place side effect on the image
 * estimate: it is visible to the W-update of the following iterations (and to "image_est"). */
int lpc_form_image(lpc_handle h, lpc_real* dev_out, void* stream);

/* inspection (tests, warm starts).  name: "image_est" (solver state shape), and for ADMM
 * "X","xi","rho","forward_out","W" as (B,D,Hp,Wp,C), "U","eta" as (B,D,Hp,Wp,C,2)
 * [values as the reference holds them after the same number of iterations].
 * GD family: "alpha" writes C floats. */
int lpc_get_state(lpc_handle h, const char* name, lpc_real* dev_out, void* stream);

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
/* when on, every launch of the hot loop is bracketed by hipEvents on its stream */
int lpc_profile_enable(lpc_handle h, int on);
/* average milliseconds per launch and launch counts since the last enable; arrays of LPC_K_COUNT */
int lpc_profile_read(lpc_handle h, double* avg_ms, long* launches);
/* algorithmic HBM bytes one launch of kernel k moves */
int lpc_kernel_bytes(lpc_handle h, int kernel_id, double* bytes);
/* bytes of HBM the handle owns */
int lpc_workspace_bytes(lpc_handle h, size_t* bytes);

#ifdef __cplusplus
}
#endif
#endif /* LPC_H_ */
