// lpc_rows.cpp -- launches of every row pass (see lpc_engine.h for the split of the library)
#include "lpc_engine.h"

// ------------------------------------------------------------- 2-D transform pieces --
// forward rows of ONE real source (pairs of rows) into spectrum S (planes = nplanes)
int rows_fwd_single(Engine* e, const RealSrc& src, real2* S, int nplanes, int kid) {
  const PlaneGeom& g = e->g;
  if (e->mod && e->mod->rows_fwd_single) return e->mod->rows_fwd_single(e, &src, S, nplanes, kid);
  if (e->rows_half)
    return dispatch_row(g.Wp / 2, e->planWh.skew_ok, false, [&](auto NT, auto EM, auto SK, auto) {
      constexpr int nt = decltype(NT)::value, em = decltype(EM)::value;
      constexpr bool sk = decltype(SK)::value;
      return launch_k(e, kid, k_rfwd_rows_half<nt, em, sk>, dim3(src.nrows, nplanes), nt,
                      LPC_ROW_SMEM_BYTES(g.Wp / 2, sk), g, e->planWh, e->planW.tw, src, S);
    });
  const int nblk = (src.nrows + 1) / 2;
  return dispatch_row(g.Wp, e->planW.skew_ok, e->rows_r2, [&](auto NT, auto EM, auto SK, auto R2) {
    constexpr int nt = decltype(NT)::value, em = decltype(EM)::value;
    constexpr bool sk = decltype(SK)::value, r2 = decltype(R2)::value;
    return launch_k(e, kid, k_rfwd_rows<nt, em, sk, r2>, dim3(nblk, nplanes), nt, LPC_ROW_SMEM_BYTES(g.Wp, sk), g,
                    e->planW, src, S);
  });
}

int rows_inv_single(Engine* e, const real2* S, const RealDst& dst, int nplanes, int kid) {
  const PlaneGeom& g = e->g;
  if (e->mod && e->mod->rows_inv_single) return e->mod->rows_inv_single(e, S, &dst, nplanes, kid);
  if (e->rows_half)
    return dispatch_row(g.Wp / 2, e->planWh.skew_ok, false, [&](auto NT, auto EM, auto SK, auto) {
      constexpr int nt = decltype(NT)::value, em = decltype(EM)::value;
      constexpr bool sk = decltype(SK)::value;
      return launch_k(e, kid, k_rinv_rows_half<nt, em, sk>, dim3(dst.nrows, nplanes), nt,
                      LPC_ROW_SMEM_BYTES(g.Wp / 2, sk), g, e->planWh, e->planW.tw, S, dst);
    });
  const int nblk = (dst.nrows + 1) / 2;
  const Fft1dPlan& pinv = e->rows_r2 ? e->planWi : e->planW;
  return dispatch_row(g.Wp, pinv.skew_ok, e->rows_r2, [&](auto NT, auto EM, auto SK, auto R2) {
    constexpr int nt = decltype(NT)::value, em = decltype(EM)::value;
    constexpr bool sk = decltype(SK)::value, r2 = decltype(R2)::value;
    return launch_k(e, kid, k_rinv_rows<nt, em, sk, r2>, dim3(nblk, nplanes), nt, LPC_ROW_SMEM_BYTES(g.Wp, sk), g,
                    pinv, S, dst);
  });
}

// ---- ADMM: rows of r_sp and a (e->Rsp, e->Aarr) -> the two work spectra --------------------------------------
int admm_rows_fwd(Engine* e) {
  const PlaneGeom& g = e->g;
  real2* SA = e->S;
  real2* SB = e->S + (size_t)e->P * g.cplane;
  if (e->mod && e->mod->admm_rows_fwd) return e->mod->admm_rows_fwd(e);
  if (e->rows_half)
    return dispatch_row(g.Wp / 2, e->planWh.skew_ok, false, [&](auto NTc, auto EM, auto SK, auto) {
      constexpr int nt = decltype(NTc)::value, em = decltype(EM)::value;
      constexpr bool sk = decltype(SK)::value;
      return launch_k(e, LPC_K_ROW_FWD, k_rfwd_half<nt, em, sk>, dim3(2 * g.Hp, e->P), nt,
                      LPC_ROW_SMEM_BYTES(g.Wp / 2, sk), g, e->planWh, e->planW.tw, (const real*)e->Rsp,
                      (const real*)e->Aarr, SA, SB);
    });
  return dispatch_row(g.Wp, e->planW.skew_ok, e->rows_r2, [&](auto NTc, auto EM, auto SK, auto R2) {
    constexpr int nt = decltype(NTc)::value, em = decltype(EM)::value;
    constexpr bool sk = decltype(SK)::value, r2 = decltype(R2)::value;
    return launch_k(e, LPC_K_ROW_FWD, k_rfwd_arrays<nt, em, sk, r2>, dim3(paired_rows_grid(g, false), e->P), nt,
                    LPC_ROW_SMEM_BYTES(g.Wp, sk), g, e->planW, (const real*)e->Rsp, (const real*)e->Aarr, SA, SB);
  });
}

// ---- ADMM: forward rows of r_sp (stored) and of a = mu1 X - xi' (computed in the kernel from xi, HV, HV_old, y) ------
// compile-time plans only: k_rfwd_half_x (half-length rows) / k_rfwd_arrays_x (paired rows)
int admm_rows_fwd_x(Engine* e, const AdmmScalars& sc, const K1Rows* k1) {
  if (e->mod && e->mod->admm_rows_fwd_x) return e->mod->admm_rows_fwd_x(e, &sc, k1);
  return fail("internal: the X-half row kernel lives in the plan module");
}

// ---- ADMM: the two work spectra -> V and H V (padded, no shift) ------------------------------------------------
int admm_rows_inv(Engine* e, real* Vout, real* HVout, bool skip_hv_outside) {
  if (e->mod && e->mod->admm_rows_inv) return e->mod->admm_rows_inv(e, Vout, HVout, skip_hv_outside ? 1 : 0);
  if (skip_hv_outside) return fail("internal: skipping H V rows needs the plan module's row kernels");
  const PlaneGeom& g = e->g;
  const int irows = paired_rows_grid(g, false), hrows = 2 * g.Hp, wo = 0;
  real2* SA = e->S;
  real2* SB = e->S + (size_t)e->P * g.cplane;
  const Fft1dPlan& pinv = e->rows_r2 ? e->planWi : e->planW;
  if (e->rows_half)
    return dispatch_row(g.Wp / 2, e->planWh.skew_ok, false, [&](auto NTc, auto EM, auto SK, auto) {
      constexpr int nt = decltype(NTc)::value, em = decltype(EM)::value;
      constexpr bool sk = decltype(SK)::value;
      return launch_k(e, LPC_K_ROW_INV, k_rinv_half<nt, em, sk>, dim3(hrows, e->P), nt,
                      LPC_ROW_SMEM_BYTES(g.Wp / 2, sk), g, e->planWh, e->planW.tw, (const real2*)SA,
                      (const real2*)SB, Vout, HVout, skip_hv_outside ? 1 : 0);
    });
  return dispatch_row(g.Wp, pinv.skew_ok, e->rows_r2, [&](auto NTc, auto EM, auto SK, auto R2) {
    constexpr int nt = decltype(NTc)::value, em = decltype(EM)::value;
    constexpr bool sk = decltype(SK)::value, r2 = decltype(R2)::value;
    return launch_k(e, LPC_K_ROW_INV, k_rinv_arrays<nt, em, sk, r2>, dim3(irows, e->P), nt,
                    LPC_ROW_SMEM_BYTES(g.Wp, sk), g, pinv, (const real2*)SA, (const real2*)SB, Vout, HVout, wo);
  });
}
