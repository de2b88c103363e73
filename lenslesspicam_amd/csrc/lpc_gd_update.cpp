// lpc_gd_update.cpp -- launches of the gradient-descent update rows (own translation unit: the device compiler works on
// the units in parallel, and this kernel family is the longest to compile)
#include "lpc_engine.h"
#include "lpc_gd_kernels.h"

// spectrum rows of the gradient (e->S2) -> irfft -> shift + crop -> fused momentum / projection update of x
int gd_rows_update(Engine* e, const GdScalars& sc, const real* alpha) {
  const PlaneGeom& g = e->g;
  const int nblk = (g.H + 1) / 2;
  const Fft1dPlan& pinv = e->rows_r2 ? e->planWi : e->planW;
  if (e->rows_half && e->static_rows)
    return with_row_shape(e, [&](auto SHc) {
      using SH = decltype(SHc);
      return with_sk(e->static_sk, [&](auto SKc) {
      constexpr bool sk = decltype(SKc)::value;
        return launch_k(e, LPC_K_SPATIAL, k_rinv_gd_update_half<SH::nt, SH::em, sk, SPlanArg<typename SH::plan>>, dim3(g.H, e->P), SH::nt,
                        LPC_ROW_SMEM_BYTES(SH::plan::n, sk), g, splan_arg<typename SH::plan>(e->planWh), e->planW.tw,
                      (const real2*)e->S2, e->gx, e->gaux, alpha, sc);
      });
    });
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
