// lpc_prep_kernels.h -- raw-frame preparation on the device, the step in front of set_data
// (SURVEY.md section 8f row N3).  Restates, minus file decoding / Bayer / resizing:
//   load_image  lensless/utils/io.py:157-196   flips, background removal, clip, normalise by the frame maximum
//   load_psf    lensless/utils/io.py:283-375   background level = mean of a corner window, clip, single_psf, / ||.||_2
//   load_data   lensless/utils/io.py:462-552   PSF background (fraction of full scale) re-scaled to the frame's
//                                              bit depth (get_max_val, image.py:251-278), optional rgb2gray
// All of it is streaming work on uint8 / uint16 / float input: one reduction pass over the raw pixels
// (per-channel maxima, or the window mean and the energy for a PSF) and one pass that writes the float image;
// the normalisers are derived on the device, so a capture goes camera buffer -> solver without touching the host.
#pragma once
#include "lpc_metric_kernels.h"

struct PrepGeom {
  int H, W, Cin, Cout;   // Cout = 1 when gray (or single_psf without channel repeat), else Cin
  int flip_ud, flip_lr;  // source row / column reversed
  int rev;               // channel order reversed on load (BGR input)
  int gray;              // rgb2gray AFTER normalisation (io.py:550-552)
  int raw_type;          // 0 u8, 1 u16, 2 f32, 3 f64
  int p0, p1;            // PSF background window [p0:p1, p0:p1]
  int single;            // PSF: sum the channels (io.py:357-361)
  int normalize;
};

static __device__ __forceinline__ real raw_at(const void* LPC_RESTRICT raw, int type, long i) {
  switch (type) {
    case 0: return (real)((const unsigned char*)raw)[i];
    case 1: return (real)((const unsigned short*)raw)[i];
    case 2: return (real)((const float*)raw)[i];
    default: return (real)((const double*)raw)[i];
  }
}

// index of (image n, output row r, output col c, channel ch) in the raw channels-last array
static __device__ __forceinline__ long raw_index(const PrepGeom& g, long n, int r, int c, int ch) {
  const int sr = g.flip_ud ? g.H - 1 - r : r;
  const int sc = g.flip_lr ? g.W - 1 - c : c;
  const int sch = g.rev ? g.Cin - 1 - ch : ch;
  return ((n * g.H + sr) * g.W + sc) * g.Cin + sch;
}

// partial[(n*Cin + ch)*nblk + blk] = max of raw channel ch of image n
template <int NT>
__global__ __launch_bounds__(NT) void k_prep_chanmax(PrepGeom g, const void* LPC_RESTRICT raw,
                                                      real* LPC_RESTRICT partial) {
  LPC_DYN_SMEM(smem);
  real* scratch = (real*)smem;
  const int tid = threadIdx.x;
  const long n = blockIdx.y / g.Cin;
  const int ch = blockIdx.y % g.Cin;
  const long npx = (long)g.H * g.W;
  real mx = -INFINITY, mn = INFINITY;
  for (long e = (long)blockIdx.x * NT + tid; e < npx; e += (long)gridDim.x * NT) {
    const real a = raw_at(raw, g.raw_type, (n * npx + e) * g.Cin + (g.rev ? g.Cin - 1 - ch : ch));
    mx = rmax(mx, a); mn = rmin(mn, a);
  }
  block_minmax<NT>(mx, mn, scratch, tid);
  if (tid == 0) partial[(long)blockIdx.y * gridDim.x + blockIdx.x] = mx;
}

// get_max_val (image.py:251-278): full-scale value of an integer image from its largest pixel
static __device__ __forceinline__ real full_scale(real raw_max) {
  const long m = (long)raw_max;
  int nbits = 0;
  while ((1L << nbits) < m) ++nbits;                       // ceil(log2(m))
  if (nbits != 8 && nbits != 10 && nbits != 12 && nbits != 16)
    nbits = nbits < 8 ? 8 : nbits < 10 ? 10 : nbits < 12 ? 12 : 16;
  return (real)((1L << nbits) - 1);
}

// per frame: the background in pixel units and the normaliser.  par[n] = (bg_0, bg_1, bg_2, 1/unused, max)
//   bg.max() <= 1 and integer frame  ->  bg *= get_max_val(frame)          (io.py:170-172)
//   max = max_c clip(max_c(raw) - bg_c, 0)      (the subtraction and the clip are monotone)
__global__ void k_prep_frame_params(PrepGeom g, const real* LPC_RESTRICT partial, int nblk,
                                    const real* LPC_RESTRICT bg, int nframes, real* LPC_RESTRICT par) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nframes) return;
  real cmax[3] = {(real)0., (real)0., (real)0.};
  real all = -INFINITY;
  for (int ch = 0; ch < g.Cin; ++ch) {
    real mx = -INFINITY;
    for (int b = 0; b < nblk; ++b) mx = rmax(mx, partial[((long)n * g.Cin + ch) * nblk + b]);
    cmax[ch] = mx;
    all = rmax(all, mx);
  }
  real bgs[3] = {(real)0., (real)0., (real)0.};
  if (bg) {
    real bmax = -INFINITY;
    for (int ch = 0; ch < g.Cin; ++ch) bmax = rmax(bmax, bg[ch]);
    const real sc = (bmax <= (real)1. && g.raw_type < 2) ? full_scale(all) : (real)1.;
    for (int ch = 0; ch < g.Cin; ++ch) bgs[ch] = bg[ch] * sc;
  }
  real nm = -INFINITY;
  for (int ch = 0; ch < g.Cin; ++ch) {
    real v = cmax[ch] - bgs[ch];
    if (bg) v = rmax(v, (real)0.);
    nm = rmax(nm, v);
  }
  par[5 * n + 0] = bgs[0]; par[5 * n + 1] = bgs[1]; par[5 * n + 2] = bgs[2];
  par[5 * n + 3] = bg ? (real)1. : (real)0.;
  par[5 * n + 4] = g.normalize ? nm : (real)1.;
}

// the float frame: flips, background, clip, / max, optional gray (weights 0.299 / 0.587 / 0.114 in double, image.py:205-219)
template <int NT>
__global__ __launch_bounds__(NT) void k_prep_frame(PrepGeom g, const void* LPC_RESTRICT raw,
                                                    const real* LPC_RESTRICT par, real* LPC_RESTRICT out) {
  const long n = blockIdx.y;
  const long npx = (long)g.H * g.W;
  const real b0 = par[5 * n], b1 = par[5 * n + 1], b2 = par[5 * n + 2], nm = par[5 * n + 4];
  const bool has_bg = par[5 * n + 3] != (real)0.;
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < npx; e += (long)gridDim.x * NT) {
    const int r = (int)(e / g.W), c = (int)(e - (long)r * g.W);
    real v[3];
    for (int ch = 0; ch < g.Cin; ++ch) {
      real a = raw_at(raw, g.raw_type, raw_index(g, n, r, c, ch));
      if (has_bg) a = rmax(a - (ch == 0 ? b0 : ch == 1 ? b1 : b2), (real)0.);
      v[ch] = g.normalize ? a / nm : a;
    }
    if (g.gray && g.Cin == 3) {
      out[n * npx + e] = (real)((double)v[0] * 0.299 + (double)v[1] * 0.587 + (double)v[2] * 0.114);
    } else {
      for (int ch = 0; ch < g.Cin; ++ch) out[(n * npx + e) * g.Cin + ch] = v[ch];
    }
  }
}

// ---- PSF ---------------------------------------------------------------------------------------------
// one block per channel: bgv[ch] = mean over all depths of the window [p0:p1, p0:p1] of the (flipped) stack
template <int NT>
__global__ __launch_bounds__(NT) void k_prep_psf_bg(PrepGeom g, const void* LPC_RESTRICT raw, int depth,
                                                     real* LPC_RESTRICT bgv) {
  LPC_DYN_SMEM(smem);
  double* scratch = (double*)smem;
  const int ch = blockIdx.x, tid = threadIdx.x;
  const int wn = g.p1 - g.p0;
  const long cnt = (long)depth * wn * wn;
  double acc = 0.0;
  for (long e = tid; e < cnt; e += NT) {
    const long d = e / ((long)wn * wn);
    const int rem = (int)(e - d * wn * wn);
    const int r = g.p0 + rem / wn, c = g.p0 + rem % wn;
    acc += (double)raw_at(raw, g.raw_type, raw_index(g, d, r, c, ch));
  }
  acc = block_sum<NT>(acc, scratch, tid);
  if (tid == 0) bgv[ch] = (real)(acc / (double)cnt);
}

// value of PSF pixel (d, r, c, ch) after background removal and clip
static __device__ __forceinline__ real psf_px(const PrepGeom& g, const void* LPC_RESTRICT raw,
                                                const real* LPC_RESTRICT bgv, bool has_bg, long d, int r, int c,
                                                int ch) {
  real a = raw_at(raw, g.raw_type, raw_index(g, d, r, c, ch));
  if (has_bg) a = rmax(a - bgv[ch], (real)0.);
  return a;
}

// partial sums of squares of the cleaned PSF (channels summed first when single_psf) and raw maxima
template <int NT>
__global__ __launch_bounds__(NT) void k_prep_psf_energy(PrepGeom g, const void* LPC_RESTRICT raw, int depth,
                                                         const real* LPC_RESTRICT bgv, int has_bg,
                                                         double* LPC_RESTRICT psum, real* LPC_RESTRICT pmax) {
  LPC_DYN_SMEM(smem);
  double* scratch = (double*)smem;
  real* scratch_r = (real*)(scratch + NT);
  const int tid = threadIdx.x;
  const long npx = (long)depth * g.H * g.W;
  double acc = 0.0;
  real mx = -INFINITY, mn = INFINITY;
  for (long e = (long)blockIdx.x * NT + tid; e < npx; e += (long)gridDim.x * NT) {
    const long d = e / ((long)g.H * g.W);
    const long rem = e - d * g.H * g.W;
    const int r = (int)(rem / g.W), c = (int)(rem - (long)r * g.W);
    real s = (real)0.;
    for (int ch = 0; ch < g.Cin; ++ch) {
      const real raw_v = raw_at(raw, g.raw_type, raw_index(g, d, r, c, ch));
      mx = rmax(mx, raw_v); mn = rmin(mn, raw_v);
      const real a = has_bg ? rmax(raw_v - bgv[ch], (real)0.) : raw_v;
      if (g.single) s += a; else acc += (double)a * (double)a;
    }
    if (g.single) acc += (double)s * (double)s;
  }
  acc = block_sum<NT>(acc, scratch, tid);
  __syncthreads();
  block_minmax<NT>(mx, mn, scratch_r, tid);
  if (tid == 0) { psum[blockIdx.x] = acc; pmax[blockIdx.x] = mx; }
}

// nrm[0] = ||psf||_2 ; bg_out[ch] = bg[ch] / get_max_val(raw)     (io.py:366-369)
__global__ void k_prep_psf_finish(PrepGeom g, const double* LPC_RESTRICT psum, const real* LPC_RESTRICT pmax,
                                  int nblk, const real* LPC_RESTRICT bgv, int has_bg, real* LPC_RESTRICT nrm,
                                  real* LPC_RESTRICT bg_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  real mx = -INFINITY;
  for (int b = 0; b < nblk; ++b) { s += psum[b]; mx = rmax(mx, pmax[b]); }
  nrm[0] = (real)sqrt(s);
  if (bg_out) {
    const real fs = g.raw_type < 2 ? full_scale(mx) : (real)1.;
    for (int ch = 0; ch < g.Cin; ++ch) bg_out[ch] = has_bg ? bgv[ch] / fs : (real)0.;
  }
}

// the float PSF (D,H,W,Cout): cleaned / ||.||_2, single_psf replicated over `rep` channels, optional gray
template <int NT>
__global__ __launch_bounds__(NT) void k_prep_psf(PrepGeom g, const void* LPC_RESTRICT raw, int depth,
                                                  const real* LPC_RESTRICT bgv, int has_bg,
                                                  const real* LPC_RESTRICT nrm, int rep, real* LPC_RESTRICT out) {
  const long npx = (long)depth * g.H * g.W;
  const real nm = nrm[0];
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < npx; e += (long)gridDim.x * NT) {
    const long d = e / ((long)g.H * g.W);
    const long rem = e - d * g.H * g.W;
    const int r = (int)(rem / g.W), c = (int)(rem - (long)r * g.W);
    real v[3] = {(real)0., (real)0., (real)0.};
    if (g.single) {
      real s = (real)0.;
      for (int ch = 0; ch < g.Cin; ++ch) s += psf_px(g, raw, bgv, has_bg != 0, d, r, c, ch);
      s = s / nm;
      for (int ch = 0; ch < rep; ++ch) v[ch] = s;
    } else {
      for (int ch = 0; ch < g.Cin; ++ch) v[ch] = psf_px(g, raw, bgv, has_bg != 0, d, r, c, ch) / nm;
    }
    const int nch = g.single ? rep : g.Cin;
    if (g.gray && nch == 3) {
      out[e] = (real)((double)v[0] * 0.299 + (double)v[1] * 0.587 + (double)v[2] * 0.114);
    } else {
      for (int ch = 0; ch < nch; ++ch) out[e * nch + ch] = v[ch];
    }
  }
}

// ---- resize: lensless/utils/image.py:28-80 (torch branch) ----------------------------------------------------
// Anti-aliased bilinear resampling, separable, last spatial axis first -- what torchvision's
// Resize(size, antialias=True) computes through torch.nn.functional.interpolate (aten UpSampleKernel.cpp; restated
// and pinned in oracle/preprocess_oracle.py).  One axis per launch: the array is viewed as (outer, L_in, inner) and
// written as (outer, L_out, inner); each thread owns one output value and forms its <= 2*ceil(support)+1 normalised
// triangle weights on the fly (downsample 4: 9 taps).  `rng` (max, min of the INPUT) != NULL: clip on the way out
// (image.py:80).
template <int NT>
__global__ __launch_bounds__(NT) void k_resize_aa_axis(const real* LPC_RESTRICT src, real* LPC_RESTRICT dst, long outer,
                                                        int Lin, int Lout, long inner, const real* LPC_RESTRICT rng) {
  const long total = outer * Lout * inner;
  const real scale = (real)Lin / (real)Lout;
  const real support = scale >= (real)1. ? scale : (real)1.;
  const real inv = scale >= (real)1. ? (real)1. / scale : (real)1.;
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
    const long in_ = e % inner;
    const long t = e / inner;
    const int i = (int)(t % Lout);
    const long o = t / Lout;
    const real center = scale * ((real)i + (real)0.5);
    int xmin = (int)(center - support + (real)0.5);
    xmin = xmin < 0 ? 0 : xmin;
    int xmax = (int)(center + support + (real)0.5);
    xmax = xmax > Lin ? Lin : xmax;
    real tot = (real)0.;
    for (int j = xmin; j < xmax; ++j) tot += rmax((real)0., (real)1. - rabs(((real)j - center + (real)0.5) * inv));
    const real* sp = src + (o * Lin) * inner + in_;
    real acc = (real)0.;
    for (int j = xmin; j < xmax; ++j) {
      const real w = rmax((real)0., (real)1. - rabs(((real)j - center + (real)0.5) * inv)) / tot;
      acc = j == xmin ? sp[(long)j * inner] * w : acc + sp[(long)j * inner] * w;
    }
    if (rng) acc = rmin(rmax(acc, rng[1]), rng[0]);
    dst[e] = acc;
  }
}
