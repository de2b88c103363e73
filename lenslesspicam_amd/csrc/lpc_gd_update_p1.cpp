// lpc_gd_update_p1.cpp -- gradient-descent update rows, two real rows per complex transform, with the radix-2 stage
// folded into the Hermitian tangling (see lpc_gd_update.cpp for why this is its own translation unit)
#include "lpc_engine.h"
#include "lpc_gd_kernels.h"

int gd_rows_update_paired_r2(Engine* e, const GdScalars& sc, const real* alpha) {
  const PlaneGeom& g = e->g;
  const int nblk = (g.H + 1) / 2;
  const Fft1dPlan& pinv = e->planWi;
  return dispatch_cfg(g.Wp, [&](auto NTc, auto EM) {
    constexpr int nt = decltype(NTc)::value, em = decltype(EM)::value;
    // (the inverse plan with the radix-2 stage first never keeps the LDS skew affine: planWi.skew_ok == 0)
    return launch_k(e, LPC_K_SPATIAL, k_rinv_gd_update<nt, em, false, true>, dim3(nblk, e->P), nt,
                    LPC_ROW_SMEM_BYTES(g.Wp, false), g, pinv, (const real2*)e->S2, e->gx, e->gaux, alpha, sc);
  });
}
