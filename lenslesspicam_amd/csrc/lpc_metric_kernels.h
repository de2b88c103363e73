// lpc_metric_kernels.h -- on-device evaluation reductions:
//   ReconstructionAlgorithm.reconstruction_error   lensless/recon/recon.py:607-653
//   mse / psnr                                      lensless/eval/metric.py:119-172
// Memory-bound single passes; squares are accumulated in double (the reference sums float32 with
// torch's pairwise tree, skimage averages in float64), results never leave the device.
#pragma once
#include "lpc_gd_kernels.h"

template <int NT>
static __device__ __forceinline__ double block_sum(double v, double* scratch, int tid) {
#if !defined(LPC_SIMT_EMU)
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);   // 64-lane wavefront
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  if (tid == 0)
    for (int w = 1; w < NT / 64; ++w) v += scratch[w];
#else
  scratch[tid] = v;
  __syncthreads();
  if (tid == 0)
    for (int w = 1; w < NT; ++w) v += scratch[w];
#endif
  return v;
}

// (min, max - min) of item (b, d) over its C planes: the normalisation of recon.py:640-645
// (amin / amax over (H, W, C); max(z - min) == max(z) - min because the subtraction is monotone)
__global__ void k_item_range(const real* LPC_RESTRICT partial, int nblk, int C, int nitems,
                             real* LPC_RESTRICT rng) {
  const int it = blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= nitems) return;
  real mx = -INFINITY, mn = INFINITY;
  for (int c = 0; c < C; ++c)
    for (int b = 0; b < nblk; ++b) {
      const long i = 2 * ((long)(it * C + c) * nblk + b);
      mx = rmax(mx, partial[i]);
      mn = rmin(mn, partial[i + 1]);
    }
  rng[2 * it] = mn;
  rng[2 * it + 1] = mx - mn;
}

// partial[pl][blk] = sum over the block's pixels of (N(hx) - y)^2 for un-padded plane pl = (b*D + d)*C + c.
// y: planar [b*C + c][H][W] (the frame kept by lpc_set_data) or channels-last (B,H,W,C) (y_hwc).
template <int NT>
__global__ __launch_bounds__(NT) void k_sqerr(PlaneGeom g, const real* LPC_RESTRICT hx,
                                               const real* LPC_RESTRICT y, int y_hwc,
                                               const real* LPC_RESTRICT rng, double* LPC_RESTRICT partial) {
  LPC_DYN_SMEM(smem);
  double* scratch = (double*)smem;
  const int tid = threadIdx.x;
  const long pl = blockIdx.y;
  const long item = pl / g.C;                         // (b, d)
  const int c = (int)(pl % g.C);
  const long b = pl / g.DC;
  real mn = (real)0., den = (real)1.;
  if (rng) { mn = rng[2 * item]; den = rng[2 * item + 1]; }
  double acc = 0.0;
  for (long e = (long)blockIdx.x * NT + tid; e < g.uplane; e += (long)gridDim.x * NT) {
    real v = hx[pl * g.uplane + e];
    if (rng) v = (v - mn) / den;
    const real yv = y_hwc ? y[(b * g.uplane + e) * g.C + c] : y[(b * g.C + c) * g.uplane + e];
    const real d = v - yv;
    acc += (double)d * (double)d;
  }
  acc = block_sum<NT>(acc, scratch, tid);
  if (tid == 0) partial[pl * gridDim.x + blockIdx.x] = acc;
}

// out[b] = sum over the D*C planes of batch item b / npix        (recon.py:650)
__global__ void k_sqerr_finish(const double* LPC_RESTRICT partial, int nblk, int DC, int B, double npix,
                               real* LPC_RESTRICT out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s = 0.0;
  for (long i = (long)b * DC * nblk; i < (long)(b + 1) * DC * nblk; ++i) s += partial[i];
  out[b] = (real)(s / npix);
}

// ---- mse / psnr between image pairs (flat items of n values each) ---------------------------------------
// (max, min) partials of item `blockIdx.y` of a flat array
template <int NT>
__global__ __launch_bounds__(NT) void k_flat_minmax(const real* LPC_RESTRICT x, long n,
                                                     real* LPC_RESTRICT partial) {
  LPC_DYN_SMEM(smem);
  real* scratch = (real*)smem;
  const int tid = threadIdx.x;
  const long it = blockIdx.y;
  real mx = -INFINITY, mn = INFINITY;
  for (long e = (long)blockIdx.x * NT + tid; e < n; e += (long)gridDim.x * NT) {
    const real a = x[it * n + e];
    mx = rmax(mx, a); mn = rmin(mn, a);
  }
  block_minmax<NT>(mx, mn, scratch, tid);
  if (tid == 0) {
    partial[2 * (it * gridDim.x + blockIdx.x)] = mx;
    partial[2 * (it * gridDim.x + blockIdx.x) + 1] = mn;
  }
}

// rng[it] = (max, min) of the item
__global__ void k_flat_range(const real* LPC_RESTRICT partial, int nblk, int nitems, real* LPC_RESTRICT rng) {
  const int it = blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= nitems) return;
  real mx = -INFINITY, mn = INFINITY;
  for (int b = 0; b < nblk; ++b) {
    mx = rmax(mx, partial[2 * ((long)it * nblk + b)]);
    mn = rmin(mn, partial[2 * ((long)it * nblk + b) + 1]);
  }
  rng[2 * it] = mx;
  rng[2 * it + 1] = mn;
}

// partial[it][blk] = sum (t/tmax - e/emax)^2   (metric.py:136-141: both images divided by their own maximum)
template <int NT>
__global__ __launch_bounds__(NT) void k_pair_sqdiff(const real* LPC_RESTRICT t, const real* LPC_RESTRICT x, long n,
                                                     const real* LPC_RESTRICT rng_t,
                                                     const real* LPC_RESTRICT rng_x, int normalize,
                                                     double* LPC_RESTRICT partial) {
  LPC_DYN_SMEM(smem);
  double* scratch = (double*)smem;
  const int tid = threadIdx.x;
  const long it = blockIdx.y;
  const real tm = normalize ? rng_t[2 * it] : (real)1., xm = normalize ? rng_x[2 * it] : (real)1.;
  double acc = 0.0;
  for (long e = (long)blockIdx.x * NT + tid; e < n; e += (long)gridDim.x * NT) {
    const real d = t[it * n + e] / tm - x[it * n + e] / xm;
    acc += (double)d * (double)d;
  }
  acc = block_sum<NT>(acc, scratch, tid);
  if (tid == 0) partial[it * gridDim.x + blockIdx.x] = acc;
}

// out[it] = (mse, psnr).  psnr = 10 log10(R^2 / mse) with skimage's rule for float images
// (peak_signal_noise_ratio, data_range=None): R = 1 if min(true) >= 0 else 2.
__global__ void k_pair_finish(const double* LPC_RESTRICT partial, int nblk, int nitems, long n,
                              const real* LPC_RESTRICT rng_t, int normalize, real* LPC_RESTRICT out) {
  const int it = blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= nitems) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += partial[(long)it * nblk + b];
  const double mse = s / (double)n;
  const real tmin = normalize ? rng_t[2 * it + 1] / rng_t[2 * it] : rng_t[2 * it + 1];
  const double R = tmin >= (real)0. ? 1.0 : 2.0;
  out[2 * it] = (real)mse;
  out[2 * it + 1] = (real)(10.0 * log10(R * R / mse));
}
