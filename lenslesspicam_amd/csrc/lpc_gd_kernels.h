// lpc_gd_kernels.h -- kernels of the projected-gradient family (vanilla / Nesterov / FISTA)
// and the small reductions of the set-up path.
//
// One GD iteration = grad = H^T (H x - y) followed by a momentum update and the
// non-negativity projection (lensless/recon/gd.py:128-134,183-188,235-241).  The state is
// UN-padded; only spectra are padded.  Pad, crop, ifftshift, "- y", the momentum arithmetic
// and the projection are all folded into the row passes:
//   k_rfwd_rows        x rows (pad on load)                         -> spectrum rows
//   [column passes, * H fused in the middle]
//   k_rinv_gd_mid      spectrum rows -> irfft -> shift+crop -> - y -> pad -> rfft -> spectrum rows
//   [column passes, * conj(H) fused in the middle]
//   k_rinv_gd_update   spectrum rows -> irfft -> shift+crop = grad -> fused update of x (+ momentum)
#pragma once
#include "lpc_kernels.h"

// ---- inverse rows -> residual -> forward rows, all inside the workgroup -------------------------
// R2: plan_inv is the inverse-row plan with the radix-2 stage first (fused into the tangling, un-skewed
// tile); the forward transform keeps the skewed tile (SK) and folds its last radix-2 stage into the untangling.
template <int NT, int EMAX, int SK, bool R2>
__global__ __launch_bounds__(NT) void k_rinv_gd_mid(PlaneGeom g, Fft1dPlan plan, Fft1dPlan plan_inv,
                                                     const real2* LPC_RESTRICT Sin,
                                                     real2* LPC_RESTRICT Sout,
                                                     const real* LPC_RESTRICT Y) {
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int tid = LPC_TID(NT);
  const int u0 = 2 * blockIdx.x, u1 = u0 + 1;
  const long pl = blockIdx.y;
  const bool v1 = u1 < g.H;
  const int hh = g.Hp / 2, hw = g.Wp / 2;
  const int sr0 = wrap_add(g.sh + u0, hh, g.Hp);
  const int sr1 = wrap_add(g.sh + (v1 ? u1 : u0), hh, g.Hp);
  const real2* sp = Sin + pl * g.cplane;
  constexpr int SKI = R2 ? 0 : SK;    // tile layout of the inverse transform's result
  if (R2) tangle_r2_load<NT>(s, g.Wp, sp + (long)sr0 * g.cpitch, sp + (long)sr1 * g.cpitch, v1, tid);
  else tangle_load<NT, EMAX, SK>(s, g.Wp, g.Wc, sp + (long)sr0 * g.cpitch, sp + (long)sr1 * g.cpitch, v1, tid);
  __syncthreads();
  fft_tile<NT, EMAX, true, SKI, true>(s, R2 ? plan_inv : plan, 1, make_fastdiv_dev1(), tid, LdsNatural{}, LdsNatural{},
                                      NoFix{}, R2 ? 1 : 0, 0);
  // residual, re-padded: sample i of the new row = (i in window) ? conv[(i + Wp/2) mod Wp] - y[i - sw] : 0,
  // evaluated on the fly as the source of the forward transform's first stage
  const int dpl = (int)(pl / g.DC) * g.C + (int)(pl % g.C);
  const real* y0 = Y + (long)dpl * g.uplane + (long)u0 * g.W;
  const real* y1 = y0 + g.W;
  auto resid = [&](int i, int) {
    const int c = i - g.sw;
    if (c < 0 || c >= g.W) return make_real2((real)0., (real)0.);
    const real2 z = s[lds_slot<SKI>(wrap_add(i, hw, g.Wp))];
    return make_real2(z.x - y0[c], v1 ? z.y - y1[c] : (real)0.);
  };
  fft_tile<NT, EMAX, false, SK, true, true>(s, plan, 1, make_fastdiv_dev1(), tid, resid, LdsNatural{}, NoFix{}, 0,
                                            R2 ? 1 : 0);
  real2* o = Sout + pl * g.cplane + (long)(g.sh + u0) * g.cpitch;
  if (R2) untangle_r2_store<NT, SK>(s, g.Wp, plan.tw, o, o + g.cpitch, v1, tid);
  else untangle_store<NT, SK>(s, g.Wp, g.Wc, o, o + g.cpitch, v1, tid);
}

// ---- inverse rows -> gradient -> fused update --------------------------------------------
struct GdScalars {
  int kind;        // 0 vanilla, 1 nesterov, 2 fista
  real mu;        // nesterov: float32(mu)
  real negmu;     // nesterov: float32(-mu)
  real onepmu;    // nesterov: float32(1 + mu)
  real coef;      // fista: float32((t_k - 1) / t_{k+1})
  int first;       // fista: x_k aliases the iterate during the first update (gd.py:233,236)
  int split;       // 1: stop in front of the projection (plug-and-play hook, lpc_iterate_begin / _end)
};

// One element of the three _update()s as a function of values (x = iterate, pp = the auxiliary state, gr = gradient):
// returns the value to store in X (the point the next iteration's forward model is evaluated at) and sets aux_out to
// the value AUX takes where the variant writes it (gd_aux_access says whether it reads / writes AUX at all).
// KIND >= 0: the variant is known at compile time (the half-row kernels branch on p.kind ONCE and run a straight-line
// body: with the uniform branches inside every element the loads of x and of the auxiliary state could not be issued
// ahead of one another -- 5 700 static SALU instructions in the update kernel); KIND < 0: read p.kind.
// SPLIT: 0 / 1 when the caller knows p.split at compile time (the update kernels with fused forward rows never run split)
template <int KIND = -1, int SPLIT = -1>
static __device__ __forceinline__ real gd_update_val(real x, real pp, real gr, real al, const GdScalars& p,
                                                      real& aux_out) {
  const int kind = KIND >= 0 ? KIND : p.kind;
  aux_out = pp;
  if (SPLIT >= 0 ? SPLIT != 0 : p.split != 0) {   // everything up to `self._form_image()` of the three _update()s; k_gd_post finishes
    if (kind == 1) {
      const real pn = p.mu * pp - al * gr;
      aux_out = pn;
      return x + (p.negmu * pp + p.onepmu * pn);
    }
    const real xs = x - al * gr;
    aux_out = xs;              // stored only when kind == 2 && first: x_k aliases the iterate before the first projection
    return xs;
  }
  if (kind == 0) return rmax(x - al * gr, (real)0.);   // gd.py:132-134
  if (kind == 1) {                                      // gd.py:183-188
    const real pn = p.mu * pp - al * gr;
    const real xn = x + (p.negmu * pp + p.onepmu * pn);
    aux_out = pn;
    return rmax(xn, (real)0.);
  }
  const real x1 = x - al * gr;                            // gd.py:235-241
  const real xk = rmax(x1, (real)0.);
  const real xp = p.first ? x1 : pp;
  aux_out = xk;
  return xk + p.coef * (xk - xp);
}
template <int KIND = -1, int SPLIT = -1>
static __device__ __forceinline__ void gd_aux_access(const GdScalars& p, bool& rd, bool& wr) {
  const int kind = KIND >= 0 ? KIND : p.kind;
  const bool split = SPLIT >= 0 ? SPLIT != 0 : p.split != 0;
  rd = kind == 1 || (kind == 2 && !split && !p.first);
  wr = kind == 1 || (kind == 2 && (!split || p.first));
}
template <int KIND = -1>
static __device__ __forceinline__ real gd_update_one(real* LPC_RESTRICT X, real* LPC_RESTRICT AUX, long o,
                                                      real gr, real al, const GdScalars& p) {
  bool rd, wr;
  gd_aux_access<KIND>(p, rd, wr);
  real an;
  const real xs = gd_update_val<KIND>(X[o], rd ? AUX[o] : (real)0., gr, al, p, an);
  if (wr) AUX[o] = an;
  X[o] = xs;
  return xs;
}
// two neighbouring columns at once (o even: 8-byte accesses; the half-row kernels use it when the window offset,
// the frame width and Wp / 2 are all even, so that gradient samples 2i and 2i + 1 are one aligned pair of x)
template <int KIND = -1>
static __device__ __forceinline__ real2 gd_update_pair(real* LPC_RESTRICT X, real* LPC_RESTRICT AUX, long o,
                                                        real2 gr, real al, const GdScalars& p) {
  bool rd, wr;
  gd_aux_access<KIND>(p, rd, wr);
  const real2 x = *(const real2*)(X + o);
  const real2 pp = rd ? *(const real2*)(AUX + o) : make_real2((real)0., (real)0.);
  real2 an, xs;
  xs.x = gd_update_val<KIND>(x.x, pp.x, gr.x, al, p, an.x);
  xs.y = gd_update_val<KIND>(x.y, pp.y, gr.y, al, p, an.y);
  if (wr) *(real2*)(AUX + o) = an;
  *(real2*)(X + o) = xs;
  return xs;
}

template <int NT, int EMAX, int SK, bool R2>
__global__ __launch_bounds__(NT) void k_rinv_gd_update(PlaneGeom g, Fft1dPlan plan,
                                                        const real2* LPC_RESTRICT Sin,
                                                        real* LPC_RESTRICT X, real* LPC_RESTRICT AUX,
                                                        const real* LPC_RESTRICT alpha, GdScalars p) {
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int tid = LPC_TID(NT);
  const int u0 = 2 * blockIdx.x, u1 = u0 + 1;
  const long pl = blockIdx.y;
  const bool v1 = u1 < g.H;
  const int hh = g.Hp / 2, hw = g.Wp / 2;
  const int sr0 = wrap_add(g.sh + u0, hh, g.Hp);
  const int sr1 = wrap_add(g.sh + (v1 ? u1 : u0), hh, g.Hp);
  const real2* sp = Sin + pl * g.cplane;
  if (R2) tangle_r2_load<NT>(s, g.Wp, sp + (long)sr0 * g.cpitch, sp + (long)sr1 * g.cpitch, v1, tid);
  else tangle_load<NT, EMAX, SK>(s, g.Wp, g.Wc, sp + (long)sr0 * g.cpitch, sp + (long)sr1 * g.cpitch, v1, tid);
  __syncthreads();
  const real al = alpha[pl % g.C];
  const long base = pl * g.uplane + (long)u0 * g.W;
  // the drain of the inverse transform hands each gradient sample straight to the fused update (shift + crop)
  auto upd = [&](int i, int, real2 z) {
    const int c = shifted_col(i, hw, g.sw, g.Wp);
    if (c < g.W) {
      gd_update_one(X, AUX, base + c, z.x, al, p);
      if (v1) gd_update_one(X, AUX, base + g.W + c, z.y, al, p);
    }
  };
  fft_tile<NT, EMAX, true, SK, true, false, true>(s, plan, 1, make_fastdiv_dev1(), tid, LdsNatural{}, upd, NoFix{},
                                                  R2 ? 1 : 0, 0);
}

// ---- the same two kernels with one real row per half-length transform (wide frames, see k_rfwd_half) ----
template <int NT, int EMAX, int SK, class PL = Fft1dPlan>
__global__ __launch_bounds__(NT) void k_rinv_gd_mid_half(PlaneGeom g, PL plan, const real2* LPC_RESTRICT twW,
                                                          const real2* LPC_RESTRICT Sin,
                                                          real2* LPC_RESTRICT Sout,
                                                          const real* LPC_RESTRICT Y) {
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int tid = LPC_TID(NT), u = (int)LPC_BX(g);
  const long pl = LPC_BY(g);
  const int hh = g.Hp / 2, hw = g.Wp / 2, M = g.Wp >> 1;
  const int sr = wrap_add(g.sh + u, hh, g.Hp);
  tangle_half_load<NT, EMAX, SK>(s, M, twW, Sin + pl * g.cplane + (long)sr * g.cpitch, tid);
  __syncthreads();
  fft_tile<NT, EMAX, true, SK, true>(s, plan, 1, make_fastdiv_dev1(), tid, LdsNatural{}, LdsNatural{});
  // slot j now holds (conv[2j], conv[2j+1]); residual sample m = (m in window) ? conv[(m + Wp/2) mod Wp] - y[m - sw] : 0
  const int dpl = (int)(pl / g.DC) * g.C + (int)(pl % g.C);
  const real* y = Y + (long)dpl * g.uplane + (long)u * g.W;
  auto sample = [&](int m) {
    const int c = m - g.sw;
    if (c < 0 || c >= g.W) return (real)0.;
    const int q = wrap_add(m, hw, g.Wp);
    const real2 z = s[lds_slot<SK>(q >> 1)];
    return ((q & 1) ? z.y : z.x) - y[c];
  };
  const bool pair = ((g.sw | g.W | hw) & 1) == 0;   // samples 2i, 2i + 1 = one aligned pair of y and one LDS slot
  auto resid = [&](int i, int) {
    if (pair) {
      const int c = 2 * i - g.sw;
      if (c < 0 || c >= g.W) return make_real2((real)0., (real)0.);
      const real2 z = s[lds_slot<SK>(wrap_add(2 * i, hw, g.Wp) >> 1)];
      const real2 yy = *(const real2*)(y + c);
      return make_real2(z.x - yy.x, z.y - yy.y);
    }
    return make_real2(sample(2 * i), sample(2 * i + 1));
  };
  fft_tile<NT, EMAX, false, SK, true, true>(s, plan, 1, make_fastdiv_dev1(), tid, resid, LdsNatural{});
  untangle_half_store<NT, SK>(s, M, twW, Sout + pl * g.cplane + (long)(g.sh + u) * g.cpitch, tid);
}

template <int NT, int EMAX, int SK, class PL = Fft1dPlan>
__global__ __launch_bounds__(NT) void k_rinv_gd_update_half(PlaneGeom g, PL plan,
                                                             const real2* LPC_RESTRICT twW,
                                                             const real2* LPC_RESTRICT Sin, real* LPC_RESTRICT X,
                                                             real* LPC_RESTRICT AUX, const real* LPC_RESTRICT alpha,
                                                             GdScalars p) {
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int tid = LPC_TID(NT), u = (int)LPC_BX(g);
  const long pl = LPC_BY(g);
  const int hh = g.Hp / 2, hw = g.Wp / 2;
  const int sr = wrap_add(g.sh + u, hh, g.Hp);
  tangle_half_load<NT, EMAX, SK>(s, g.Wp >> 1, twW, Sin + pl * g.cplane + (long)sr * g.cpitch, tid);
  __syncthreads();
  const real al = alpha[pl % g.C];
  const long base = pl * g.uplane + (long)u * g.W;
  const bool pair = ((g.sw | g.W | hw) & 1) == 0;
  auto run = [&](auto kind_tag) {           // one branch on the variant, then a straight-line body (gd_update_val)
    constexpr int KIND = decltype(kind_tag)::value;
    auto upd = [&](int i, int, real2 z) {   // gradient samples 2i, 2i+1 -> shift + crop -> fused update
      const int c0 = shifted_col(2 * i, hw, g.sw, g.Wp);
      if (pair) {
        if (c0 < g.W) gd_update_pair<KIND>(X, AUX, base + c0, z, al, p);
        return;
      }
      if (c0 < g.W) gd_update_one<KIND>(X, AUX, base + c0, z.x, al, p);
      const int c1 = shifted_col(2 * i + 1, hw, g.sw, g.Wp);
      if (c1 < g.W) gd_update_one<KIND>(X, AUX, base + c1, z.y, al, p);
    };
    fft_tile<NT, EMAX, true, SK, true, false, true>(s, plan, 1, make_fastdiv_dev1(), tid, LdsNatural{}, upd);
  };
  if (p.kind == 2) run(std::integral_constant<int, 2>{});
  else if (p.kind == 1) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 0>{});
}

// ---- the same update with the NEXT iteration's forward rows fused behind it -------------------------------------
// irfft row -> shift + crop = gradient -> fused update of x (+ momentum, projection) -> the updated row, re-padded, goes
// straight back through the forward row transform: the next iteration finds the row spectra of its iterate already in
// `Sout` and skips its forward row pass (one launch and one read of x less per iteration).  Same structure as
// k_rinv_gd_mid_half: the update is the source functor of the forward transform's first stage.
template <int NT, int EMAX, int SK, class PL = Fft1dPlan>
__global__ __launch_bounds__(NT) void k_rinv_gd_update_fwd_half(PlaneGeom g, PL plan, const real2* LPC_RESTRICT twW,
                                                                 const real2* LPC_RESTRICT Sin,
                                                                 real2* LPC_RESTRICT Sout, real* LPC_RESTRICT X,
                                                                 real* LPC_RESTRICT AUX,
                                                                 const real* LPC_RESTRICT alpha, GdScalars p) {
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int tid = LPC_TID(NT), u = (int)LPC_BX(g);
  const long pl = LPC_BY(g);
  const int hh = g.Hp / 2, hw = g.Wp / 2, M = g.Wp >> 1;
  const int sr = wrap_add(g.sh + u, hh, g.Hp);
  tangle_half_load<NT, EMAX, SK>(s, M, twW, Sin + pl * g.cplane + (long)sr * g.cpitch, tid);
  __syncthreads();
  fft_tile<NT, EMAX, true, SK, true>(s, plan, 1, make_fastdiv_dev1(), tid, LdsNatural{}, LdsNatural{});
  // slot j now holds gradient samples (2j, 2j+1) before the shift; padded sample m of the NEW row =
  // (m in window) ? updated x[m - sw] : 0, where x[c] takes gradient sample (m + Wp/2) mod Wp
  const real al = alpha[pl % g.C];
  const long base = pl * g.uplane + (long)u * g.W;
  const bool pair = ((g.sw | g.W | hw) & 1) == 0;
  auto run = [&](auto kind_tag) {           // one branch on the variant, then a straight-line body (gd_update_val)
    constexpr int KIND = decltype(kind_tag)::value;
    auto sample = [&](int m) {
      const int c = m - g.sw;
      if (c < 0 || c >= g.W) return (real)0.;
      const int q = wrap_add(m, hw, g.Wp);
      const real2 z = s[lds_slot<SK>(q >> 1)];
      return gd_update_one<KIND>(X, AUX, base + c, (q & 1) ? z.y : z.x, al, p);
    };
    auto newrow = [&](int i, int) {
      if (pair) {
        const int c = 2 * i - g.sw;
        if (c < 0 || c >= g.W) return make_real2((real)0., (real)0.);
        const real2 z = s[lds_slot<SK>(wrap_add(2 * i, hw, g.Wp) >> 1)];
        return gd_update_pair<KIND>(X, AUX, base + c, z, al, p);
      }
      return make_real2(sample(2 * i), sample(2 * i + 1));
    };
    fft_tile<NT, EMAX, false, SK, true, true>(s, plan, 1, make_fastdiv_dev1(), tid, newrow, LdsNatural{});
  };
  if (p.kind == 2) run(std::integral_constant<int, 2>{});
  else if (p.kind == 1) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 0>{});
  untangle_half_store<NT, SK>(s, M, twW, Sout + pl * g.cplane + (long)(g.sh + u) * g.cpitch, tid);
}

// second half of a split iteration: PROJ = proj(image_est) as the caller computed it, channels-last (n,H,W,C)
//   vanilla / nesterov: x = PROJ                          gd.py:134,188
//   fista: x_k = PROJ; x = x_k + coef (x_k - x_{k-1})     gd.py:236-241
template <int NT>
__global__ __launch_bounds__(NT) void k_gd_post(PlaneGeom g, const real* LPC_RESTRICT PROJ, real* LPC_RESTRICT X,
                                                 real* LPC_RESTRICT AUX, int kind, real coef) {
  const long pl = blockIdx.y;
  const long img = pl / g.C;
  const int ch = (int)(pl % g.C);
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < g.uplane; e += (long)gridDim.x * NT) {
    const real xk = PROJ[(img * g.uplane + e) * g.C + ch];
    const long o = pl * g.uplane + e;
    if (kind == 2) {
      X[o] = xk + coef * (xk - AUX[o]);
      AUX[o] = xk;
    } else {
      X[o] = xk;
    }
  }
}

// ---- reductions (set-up only): per-plane max/min with wavefront shuffles ---------------------
template <int NT>
static __device__ __forceinline__ void block_minmax(real& mx, real& mn, real* scratch, int tid) {
#if !defined(LPC_SIMT_EMU)
  for (int off = 32; off > 0; off >>= 1) {  // 64-lane wavefront
    mx = rmax(mx, __shfl_down(mx, off, 64));
    mn = rmin(mn, __shfl_down(mn, off, 64));
  }
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 0) { scratch[2 * wave] = mx; scratch[2 * wave + 1] = mn; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < NT / 64; ++w) { mx = rmax(mx, scratch[2 * w]); mn = rmin(mn, scratch[2 * w + 1]); }
  }
#else
  scratch[2 * tid] = mx; scratch[2 * tid + 1] = mn;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < NT; ++w) { mx = rmax(mx, scratch[2 * w]); mn = rmin(mn, scratch[2 * w + 1]); }
  }
#endif
}

// mode 0: values are |H* H| of a spectrum plane (pitch cpitch, Wc valid columns);
// mode 1: values are an un-padded image plane.  Writes (max, min) per (plane, block).
template <int NT>
__global__ __launch_bounds__(NT) void k_plane_minmax(PlaneGeom g, const real2* LPC_RESTRICT Hs,
                                                      const real* LPC_RESTRICT plane, int mode,
                                                      real* LPC_RESTRICT partial) {
  LPC_DYN_SMEM(smem);
  real* scratch = (real*)smem;
  const int tid = LPC_TID(NT);
  const long pl = blockIdx.y;
  real mx = -INFINITY, mn = INFINITY;
  if (mode == 0) {
    const long n = (long)g.Hp * g.Wc;
    for (long e = (long)blockIdx.x * NT + tid; e < n; e += (long)gridDim.x * NT) {
      const int r = (int)(e / g.Wc), c = (int)(e - (long)r * g.Wc);
      const real2 h = Hs[pl * g.cplane + (long)r * g.cpitch + c];
      const real a = h.x * h.x + h.y * h.y;
      mx = rmax(mx, a); mn = rmin(mn, a);
    }
  } else {
    const long n = g.uplane;
    for (long e = (long)blockIdx.x * NT + tid; e < n; e += (long)gridDim.x * NT) {
      const real a = plane[pl * g.uplane + e];
      mx = rmax(mx, a); mn = rmin(mn, a);
    }
  }
  block_minmax<NT>(mx, mn, scratch, tid);
  if (tid == 0) {
    partial[2 * (pl * gridDim.x + blockIdx.x)] = mx;
    partial[2 * (pl * gridDim.x + blockIdx.x) + 1] = mn;
  }
}

// ---- is a real spectrum plane G[r][c] the sum of a row term and a column term?  (ADMM set-up, ColPass::ga) -----
// ga[r] = G[r][0], gb[c] = G[0][c] - G[0][0]
static __global__ void k_gsep_extract(const real* LPC_RESTRICT G, int Hp, int Wc, long cpitch, real* LPC_RESTRICT ga,
                                      real* LPC_RESTRICT gb) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < Hp) ga[e] = G[e * cpitch];
  if (e < cpitch) gb[e] = e < Wc ? G[e] - G[0] : (real)0.;
}
// partial[2 b] = max |G - ga - gb|, partial[2 b + 1] = -max |G| over the block's share of the plane
template <int NT>
__global__ __launch_bounds__(NT) void k_gsep_check(const real* LPC_RESTRICT G, int Hp, int Wc, long cpitch,
                                                    const real* LPC_RESTRICT ga, const real* LPC_RESTRICT gb,
                                                    real* LPC_RESTRICT partial) {
  LPC_DYN_SMEM(smem);
  real* scratch = (real*)smem;
  const int tid = LPC_TID(NT);
  real mx = (real)0., mn = (real)0.;
  const long n = (long)Hp * Wc;
  for (long e = (long)blockIdx.x * NT + tid; e < n; e += (long)gridDim.x * NT) {
    const int r = (int)(e / Wc), c = (int)(e - (long)r * Wc);
    const real v = G[(long)r * cpitch + c];
    mx = rmax(mx, rabs(v - (ga[r] + gb[c])));
    mn = rmin(mn, -rabs(v));
  }
  block_minmax<NT>(mx, mn, scratch, tid);
  if (tid == 0) { partial[2 * blockIdx.x] = mx; partial[2 * blockIdx.x + 1] = mn; }
}

// final per-channel combine over depth planes and blocks (gd.py:100-112 flatten (D,H,W) per channel):
// mode 0: out[c] = lip_fact / max;  mode 1: out[c] = (max + min) / 2
static __global__ void k_channel_finish(const real* LPC_RESTRICT partial, int nblk, int D, int C, int mode, real lip,
                                 real* LPC_RESTRICT out) {
  const int c = threadIdx.x;
  if (c >= C) return;
  real mx = -INFINITY, mn = INFINITY;
  for (int d = 0; d < D; ++d)
    for (int b = 0; b < nblk; ++b) {
      const long i = 2 * ((long)(d * C + c) * nblk + b);
      mx = rmax(mx, partial[i]);
      mn = rmin(mn, partial[i + 1]);
    }
  out[c] = mode == 0 ? lip / mx : (mx + mn) / 2;
}

// x[plane][...] = val[plane % C]
template <int NT>
__global__ __launch_bounds__(NT) void k_fill_per_channel(real* LPC_RESTRICT x, long plane_elems, int C,
                                                          const real* LPC_RESTRICT val) {
  const long pl = blockIdx.y;
  const real v = val[pl % C];
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < plane_elems; e += (long)gridDim.x * NT)
    x[pl * plane_elems + e] = v;
}

// two planar arrays (component 0 / 1) -> channels-last with a trailing axis of 2
template <int NT>
__global__ __launch_bounds__(NT) void k_planar2_to_hwc2(const real* LPC_RESTRICT a0, const real* LPC_RESTRICT a1,
                                                         real* LPC_RESTRICT dst, int rows, int cols, int C,
                                                         int pitch, long splane) {
  const long n = (long)rows * cols * C;
  const long img = blockIdx.y;
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < n; e += (long)gridDim.x * NT) {
    const int c = (int)(e % C);
    const long rc = e / C;
    const int col = (int)(rc % cols);
    const int row = (int)(rc / cols);
    const long so = (img * C + c) * splane + (long)row * pitch + col;
    dst[(img * n + e) * 2 + 0] = a0[so];
    dst[(img * n + e) * 2 + 1] = a1[so];
  }
}
