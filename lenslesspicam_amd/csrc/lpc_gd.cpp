// lpc_gd.cpp -- launches of the gradient-descent family's fused row kernels (see lpc_engine.h for the split)
#include "lpc_engine.h"
#include "lpc_gd_kernels.h"

// spectrum rows of H x (e->S) -> irfft -> shift + crop -> - y -> re-pad -> rfft -> spectrum rows (e->S2)
int gd_rows_mid(Engine* e) {
  const PlaneGeom& g = e->g;
  const int nblk = (g.H + 1) / 2;
  if (e->mod && e->mod->gd_rows_mid) return e->mod->gd_rows_mid(e);
  if (e->rows_half)
    return dispatch_row(g.Wp / 2, e->planWh.skew_ok, false, [&](auto NTc, auto EM, auto SK, auto) {
      constexpr int nt = decltype(NTc)::value, em = decltype(EM)::value;
      constexpr bool sk = decltype(SK)::value;
      return launch_k(e, LPC_K_ROW_INV, k_rinv_gd_mid_half<nt, em, sk>, dim3(g.H, e->P), nt,
                      LPC_ROW_SMEM_BYTES(g.Wp / 2, sk), g, e->planWh, e->planW.tw, (const real2*)e->S, e->S2,
                      (const real*)e->Y);
    });
  return dispatch_row(g.Wp, e->planW.skew_ok, e->rows_r2, [&](auto NTc, auto EM, auto SK, auto R2) {
    constexpr int nt = decltype(NTc)::value, em = decltype(EM)::value;
    constexpr bool sk = decltype(SK)::value, r2 = decltype(R2)::value;
    return launch_k(e, LPC_K_ROW_INV, k_rinv_gd_mid<nt, em, sk, r2>, dim3(nblk, e->P), nt,
                    LPC_ROW_SMEM_BYTES(g.Wp, sk), g, e->planW, e->rows_r2 ? e->planWi : e->planW,
                    (const real2*)e->S, e->S2, (const real*)e->Y);
  });
}
