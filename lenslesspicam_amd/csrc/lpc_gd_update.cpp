// lpc_gd_update.cpp -- launches of the gradient-descent update rows (own translation unit: the device compiler works on
// the units in parallel, and this kernel family is the longest to compile)
#include "lpc_engine.h"
#include "lpc_gd_kernels.h"

// spectrum rows of the gradient (e->S2) -> irfft -> shift + crop -> fused momentum / projection update of x
int gd_rows_update(Engine* e, const GdScalars& sc, const real* alpha) {
  const PlaneGeom& g = e->g;
  const int nblk = (g.H + 1) / 2;
  const Fft1dPlan& pinv = e->rows_r2 ? e->planWi : e->planW;
  if (e->mod && e->mod->gd_rows_update) return e->mod->gd_rows_update(e, &sc, alpha);
  if (e->rows_half)
    return dispatch_row(g.Wp / 2, e->planWh.skew_ok, false, [&](auto NTc, auto EM, auto SK, auto) {
      constexpr int nt = decltype(NTc)::value, em = decltype(EM)::value;
      constexpr bool sk = decltype(SK)::value;
      return launch_k(e, LPC_K_SPATIAL, k_rinv_gd_update_half<nt, em, sk>, dim3(g.H, e->P), nt,
                      LPC_ROW_SMEM_BYTES(g.Wp / 2, sk), g, e->planWh, e->planW.tw, (const real2*)e->S2, e->gx,
                      e->gaux, alpha, sc);
    });
  return dispatch_row(g.Wp, pinv.skew_ok, e->rows_r2, [&](auto NTc, auto EM, auto SK, auto R2) {
    constexpr int nt = decltype(NTc)::value, em = decltype(EM)::value;
    constexpr bool sk = decltype(SK)::value, r2 = decltype(R2)::value;
    return launch_k(e, LPC_K_SPATIAL, k_rinv_gd_update<nt, em, sk, r2>, dim3(nblk, e->P), nt,
                    LPC_ROW_SMEM_BYTES(g.Wp, sk), g, pinv, (const real2*)e->S2, e->gx, e->gaux, alpha, sc);
  });
}

// the same + the forward row transform of the updated rows (e->S2 -> x, e->S): compile-time half-row plans only
int gd_rows_update_fwd(Engine* e, const GdScalars& sc, const real* alpha) {
  if (e->mod && e->mod->gd_rows_update_fwd) return e->mod->gd_rows_update_fwd(e, &sc, alpha);
  return fail("internal: the update kernel with fused forward rows lives in the plan module");
}
