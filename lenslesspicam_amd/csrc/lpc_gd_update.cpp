// lpc_gd_update.cpp -- launches of the gradient-descent update rows.  This kernel family is the longest to compile (the
// fused momentum / projection update behind an inverse row transform, one instantiation per workgroup shape), so it is
// spread over three translation units that the device compiler works on in parallel: this one (one real row per
// half-length transform) and lpc_gd_update_p0.cpp / lpc_gd_update_p1.cpp (paired rows without / with the folded radix-2
// stage).
#include "lpc_engine.h"
#include "lpc_gd_kernels.h"

int gd_rows_update_paired_r2(Engine* e, const GdScalars& sc, const real* alpha);   // lpc_gd_update_p1.cpp
int gd_rows_update_paired_plain(Engine* e, const GdScalars& sc, const real* alpha);   // lpc_gd_update_p0.cpp

// spectrum rows of the gradient (e->S2) -> irfft -> shift + crop -> fused momentum / projection update of x
int gd_rows_update(Engine* e, const GdScalars& sc, const real* alpha) {
  const PlaneGeom& g = e->g;
  if (e->mod && e->mod->gd_rows_update) return e->mod->gd_rows_update(e, &sc, alpha);
  if (e->rows_half)
    return dispatch_row(g.Wp / 2, e->planWh.skew_ok, false, [&](auto NTc, auto EM, auto SK, auto) {
      constexpr int nt = decltype(NTc)::value, em = decltype(EM)::value;
      constexpr bool sk = decltype(SK)::value;
      return launch_k(e, LPC_K_SPATIAL, k_rinv_gd_update_half<nt, em, sk>, dim3(g.H, e->P), nt,
                      LPC_ROW_SMEM_BYTES(g.Wp / 2, sk), g, e->planWh, e->planW.tw, (const real2*)e->S2, e->gx,
                      e->gaux, alpha, sc);
    });
  return e->rows_r2 ? gd_rows_update_paired_r2(e, sc, alpha) : gd_rows_update_paired_plain(e, sc, alpha);
}

// the same + the forward row transform of the updated rows (e->S2 -> x, e->S): compile-time half-row plans only
int gd_rows_update_fwd(Engine* e, const GdScalars& sc, const real* alpha) {
  if (e->mod && e->mod->gd_rows_update_fwd) return e->mod->gd_rows_update_fwd(e, &sc, alpha);
  return fail("internal: the update kernel with fused forward rows lives in the plan module");
}
