// lpc_gd_update_p0.cpp -- gradient-descent update rows, two real rows per complex transform, without the radix-2 stage
// folded into the Hermitian tangling (see lpc_gd_update.cpp for why this is its own translation unit)
#include "lpc_engine.h"
#include "lpc_gd_kernels.h"

int gd_rows_update_paired_plain(Engine* e, const GdScalars& sc, const real* alpha) {
  const PlaneGeom& g = e->g;
  const int nblk = (g.H + 1) / 2;
  const Fft1dPlan& pinv = e->planW;
  return dispatch_cfg(g.Wp, [&](auto NTc, auto EM) {
    constexpr int nt = decltype(NTc)::value, em = decltype(EM)::value;
    if (pinv.skew_ok)
      return launch_k(e, LPC_K_SPATIAL, k_rinv_gd_update<nt, em, true, false>, dim3(nblk, e->P), nt,
                      LPC_ROW_SMEM_BYTES(g.Wp, true), g, pinv, (const real2*)e->S2, e->gx, e->gaux, alpha, sc);
    return launch_k(e, LPC_K_SPATIAL, k_rinv_gd_update<nt, em, false, false>, dim3(nblk, e->P), nt,
                    LPC_ROW_SMEM_BYTES(g.Wp, false), g, pinv, (const real2*)e->S2, e->gx, e->gaux, alpha, sc);
  });
}
