// lpc_gd_update_fwd.cpp -- launches of the update rows with the next iteration's forward rows fused behind them
#include "lpc_engine.h"
#include "lpc_gd_kernels.h"

// the same + the forward row transform of the updated rows (e->S2 -> x, e->S); compile-time half-row plans only
int gd_rows_update_fwd(Engine* e, const GdScalars& sc, const real* alpha) {
  const PlaneGeom& g = e->g;
  return with_row_shape(e, [&](auto SHc) {
    using SH = decltype(SHc);
    return with_sk(e->static_sk, [&](auto SKc) {
      constexpr bool sk = decltype(SKc)::value;
      return launch_k(e, LPC_K_SPATIAL, k_rinv_gd_update_fwd_half<SH::nt, SH::em, sk, SPlanArg<typename SH::plan>>,
                      dim3(g.H, e->P), SH::nt, LPC_ROW_SMEM_BYTES(SH::plan::n, sk), g,
                      splan_arg<typename SH::plan>(e->planWh), e->planW.tw, (const real2*)e->S2, e->S, e->gx, e->gaux,
                      alpha, sc);
    });
  });
}
