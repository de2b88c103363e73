// lpc_module.cpp -- ONE plan module: the compile-time-plan instantiations (lpc_sfft.h) of the hot-loop kernels for one
// frame shape, plus their launchers.  Compiled on its own into <libdir>/modules/lpcmod_<key>.so with the -D flags of
// plan_spec_defines() (lpc_plan.h): by build.py for BASELINE.json's shapes, by lpc_create() itself (lpc_jit.cpp) for any
// other shape on first use.  The core library loads it with dlopen and calls the launchers through the LpcModule table;
// a launcher is the static branch of the core's own launch code (lpc_rows.cpp / lpc_cols.cpp / lpc_gd.cpp) with every
// plan choice turned into a constant.  Nothing here is shape-specific source: the shape arrives as macros.
#include "lpc_engine.h"
#include "lpc_gd_kernels.h"
#include "lpc_gd_v2_kernels.h"

#ifndef LPC_MOD_MID_PRE
#define LPC_MOD_MID_PRE 0
#endif
#ifndef LPC_MOD_SLAY
#define LPC_MOD_SLAY 0
#endif
#ifndef LPC_MOD_MID_PC
#define LPC_MOD_MID_PC 0
#endif
#ifndef LPC_MOD_FAMILY
#error "lpc_module.cpp is compiled with the flags of plan_spec_defines() (lpc_plan.h)"
#endif

static inline PlaneGeom geom_rev(const Engine* e, bool rev) {   // the launch's copy of the geometry (PlaneGeom::rev)
  PlaneGeom g = e->g;
  g.rev = rev ? 1 : 0;
  return g;
}

// ============================================================================== rows ==
#if LPC_MOD_ROW_KIND != 0
typedef SPlan<LPC_MOD_ROW_RAD> RowP;
#ifndef LPC_MOD_TW_LANE
#define LPC_MOD_TW_LANE 1      // row kernels: radix-8 / -16 stage twiddles from the lane-ordered table (lpc_sfft.h); 0: gathered
#endif
typedef SPlanArg<RowP, LPC_MOD_TW_LANE != 0> RowPA;
static constexpr int RNT = LPC_MOD_ROW_NT, REM = LPC_MOD_ROW_EM;
static constexpr int RSK = LPC_MOD_ROW_SK;       // LDS layout of the row tile: LPC_LAY_NONE / LPC_LAY_SKEW8 (lpc_fft.h)
static_assert(RSK != LPC_LAY_SKEW8 || RowP::skew_ok(), "this row plan does not keep the LDS skew affine");
static const size_t kRowSmem = LPC_ROW_SMEM_BYTES(RowP::n, RSK);
#endif

#if LPC_MOD_ROW_KIND == LPC_ROWS_HALF
static RowPA row_arg(const Engine* e) { return splan_arg<RowP, LPC_MOD_TW_LANE != 0>(e->planWh, e->tws_row); }

static int m_rows_fwd_single(Engine* e, const RealSrc* src, real2* S, int nplanes, int kid) {
  return launch_k(e, kid, k_rfwd_rows_half<RNT, REM, RSK, RowPA>, dim3(src->nrows, nplanes), RNT, kRowSmem, e->g,
                  row_arg(e), e->planW.tw, *src, S);
}
static int m_rows_inv_single(Engine* e, const real2* S, const RealDst* dst, int nplanes, int kid) {
  return launch_k(e, kid, k_rinv_rows_half<RNT, REM, RSK, RowPA>, dim3(dst->nrows, nplanes), RNT, kRowSmem, e->g,
                  row_arg(e), e->planW.tw, S, *dst);
}
#if LPC_MOD_FAMILY == LPC_FAM_ADMM
static int m_admm_rows_fwd(Engine* e) {
  const PlaneGeom& g = e->g;
  real2* SA = e->S;
  real2* SB = e->S + (size_t)e->P * g.cplane;
  return launch_k(e, LPC_K_ROW_FWD, k_rfwd_half<RNT, REM, RSK, RowPA>, dim3(2 * g.Hp, e->P), RNT, kRowSmem, g,
                  row_arg(e), e->planW.tw, (const real*)e->Rsp, (const real*)e->Aarr, SA, SB);
}
static int m_admm_rows_inv(Engine* e, real* Vout, real* HVout, int skip_hv_outside) {
  const PlaneGeom& g = e->g;
  real2* SA = e->S;
  real2* SB = e->S + (size_t)e->P * g.cplane;
  const int hrows = skip_hv_outside ? g.Hp + g.H : 2 * g.Hp;
  return launch_k(e, LPC_K_ROW_INV, k_rinv_half<RNT, REM, RSK, RowPA>, dim3(hrows, e->P), RNT, kRowSmem,
                  e->g, row_arg(e),
                  e->planW.tw, (const real2*)SA, (const real2*)SB, Vout, HVout, skip_hv_outside ? 1 : 0);
}
#if LPC_MOD_ROW_X
static int m_admm_rows_fwd_x(Engine* e, const AdmmScalars* sc, const K1Rows* k1) {
  if (k1) return fail("internal: half-length rows do not hold the TV / W half");
  const PlaneGeom& g = e->g;
  real2* SA = e->S;
  real2* SB = e->S + (size_t)e->P * g.cplane;
  return launch_k(e, LPC_K_ROW_FWD, k_rfwd_half_x<RNT, REM, RSK, RowPA>, dim3(2 * g.Hp, e->P), RNT, kRowSmem,
                  e->g, *sc,
                  row_arg(e), (const real2*)e->planW.tw, (const real*)e->Rsp, (const real*)e->HVb[e->hcur],
                  (const real*)e->HVb[e->hcur ^ 1], e->xi, (const real*)e->Y, SA, SB);
}
#endif
#else   // gradient-descent family
// the second form of the fused rows keeps its tile in the NATURAL layout whatever the module's other row kernels use:
// immediate LDS offsets and the fewest registers (no scratch, five workgroups per CU for the residual rows); its bank
// conflicts cost nothing measurable -- these kernels wait on memory, not on LDS (profiles/r05_notes.md)
#ifndef LPC_MOD_V2_SK
#define LPC_MOD_V2_SK LPC_LAY_NONE
#endif
static constexpr int V2SK = LPC_MOD_V2_SK;
static const size_t kV2Smem = LPC_ROW_SMEM_BYTES(RowP::n, V2SK);
static int m_gd_rows_mid(Engine* e) {
  const PlaneGeom& g = e->g;
#ifndef LPC_DOUBLE
  if constexpr (GdV2<RowP>::ok) {
    if (e->gd_v2)     // second form (lpc_gd_v2_kernels.h): one-radix plan, M / R lanes per row
      return launch_k(e, LPC_K_ROW_INV, k_gd_resid_v2<GdV2<RowP>::NB, V2SK, RowPA>, dim3(g.H, e->P), GdV2<RowP>::NB, kV2Smem,
                      geom_rev(e, e->opt.gd_rev & 1), row_arg(e), e->planW.tw, (const real2*)e->S, e->S2,
                      (const real*)e->Y, make_fastdiv((unsigned)g.DC), make_fastdiv((unsigned)g.C));
  }
#endif
  return launch_k(e, LPC_K_ROW_INV, k_rinv_gd_mid_half<RNT, REM, RSK, RowPA>, dim3(g.H, e->P), RNT, kRowSmem,
                  geom_rev(e, e->opt.gd_rev & 1),
                  row_arg(e), e->planW.tw, (const real2*)e->S, e->S2, (const real*)e->Y);
}
static int m_gd_rows_update(Engine* e, const GdScalars* sc, const real* alpha) {
  const PlaneGeom& g = e->g;
  return launch_k(e, LPC_K_SPATIAL, k_rinv_gd_update_half<RNT, REM, RSK, RowPA>, dim3(g.H, e->P), RNT, kRowSmem,
                  geom_rev(e, e->opt.gd_rev & 2),
                  row_arg(e), e->planW.tw, (const real2*)e->S2, e->gx, e->gaux, alpha, *sc);
}
static int m_gd_rows_update_fwd(Engine* e, const GdScalars* sc, const real* alpha) {
  const PlaneGeom& g = e->g;
#ifndef LPC_DOUBLE
  if constexpr (GdV2<RowP>::ok) {
    if (e->gd_v2) {
      auto go = [&](auto kernel) {
        return launch_k(e, LPC_K_SPATIAL, kernel, dim3(g.H, e->P), GdV2<RowP>::NB, kV2Smem, geom_rev(e, e->opt.gd_rev & 2),
                        row_arg(e), e->planW.tw, (const real2*)e->S2, e->S, e->gx, e->gaux, alpha, *sc,
                        make_fastdiv((unsigned)g.C));
      };
      constexpr int NB = GdV2<RowP>::NB;
      if (sc->kind == 2) return sc->first ? go(k_gd_update_fwd_v2<NB, V2SK, RowPA, 2, 1>) : go(k_gd_update_fwd_v2<NB, V2SK, RowPA, 2, 0>);
      if (sc->kind == 1) return go(k_gd_update_fwd_v2<NB, V2SK, RowPA, 1, 0>);
      return go(k_gd_update_fwd_v2<NB, V2SK, RowPA, 0, 0>);
    }
  }
#endif
  return launch_k(e, LPC_K_SPATIAL, k_rinv_gd_update_fwd_half<RNT, REM, RSK, RowPA>, dim3(g.H, e->P), RNT, kRowSmem,
                  geom_rev(e, e->opt.gd_rev & 2),
                  row_arg(e), e->planW.tw, (const real2*)e->S2, e->S, e->gx, e->gaux, alpha, *sc);
}
#endif
#endif   // half rows

#if LPC_MOD_ROW_KIND == LPC_ROWS_PAIRED   // ADMM only: two real rows per complex transform of length Wp
static RowPA row_arg(const Engine* e) { return splan_arg<RowP, LPC_MOD_TW_LANE != 0>(e->planW, e->tws_row); }

static int m_admm_rows_fwd(Engine* e) {
  const PlaneGeom& g = e->g;
  real2* SA = e->S;
  real2* SB = e->S + (size_t)e->P * g.cplane;
  return launch_k(e, LPC_K_ROW_FWD, k_rfwd_arrays<RNT, REM, RSK, false, RowPA, LPC_MOD_SLAY>, dim3(paired_rows_grid(g, false), e->P), RNT, kRowSmem, g,
                  row_arg(e), (const real*)e->Rsp, (const real*)e->Aarr, SA, SB);
}
static int m_admm_rows_inv(Engine* e, real* Vout, real* HVout, int skip_hv_outside) {
  const PlaneGeom& g = e->g;
  real2* SA = e->S;
  real2* SB = e->S + (size_t)e->P * g.cplane;
  const int irows = paired_rows_grid(g, skip_hv_outside != 0);
  return launch_k(e, LPC_K_ROW_INV, k_rinv_arrays<RNT, REM, RSK, false, RowPA, LPC_MOD_SLAY>, dim3(irows, e->P), RNT, kRowSmem,
                  e->g,
                  row_arg(e), (const real2*)SA, (const real2*)SB, Vout, HVout, skip_hv_outside ? 1 : 0);
}
#if LPC_MOD_ROW_X
// quads per lane and row with which the TV / W half can ride along: 1 or 2 (padded widths up to 8 x the lanes), else 0
constexpr int kK1Quads = (RowP::n >> 2) <= RNT ? 1 : ((RowP::n >> 2) <= 2 * RNT ? 2 : 0);
constexpr bool kK1Rows = kK1Quads != 0;
static int m_admm_rows_fwd_x(Engine* e, const AdmmScalars* sc, const K1Rows* k1) {
  const PlaneGeom& g = e->g;
  real2* SA = e->S;
  real2* SB = e->S + (size_t)e->P * g.cplane;
  // sc->skipa: `a` on the rows of the sensor window alone
  const int xrows = paired_rows_grid(g, sc->skipa != 0);
  if (k1) {
    if constexpr (kK1Rows)
      return launch_k(e, LPC_K_ROW_FWD, k_rfwd_arrays_x<RNT, REM, RSK, RowPA, true, LPC_MOD_SLAY>, dim3(xrows, e->P), RNT, kRowSmem,
                      e->g, *sc,
                      row_arg(e), (const real*)e->Rsp, (const real*)e->HVb[e->hcur], (const real*)e->HVb[e->hcur ^ 1],
                      e->xi, (const real*)e->Y, SA, SB, *k1);
    return fail("internal: this module's rows do not hold the TV / W half");
  }
  return launch_k(e, LPC_K_ROW_FWD, k_rfwd_arrays_x<RNT, REM, RSK, RowPA, false, LPC_MOD_SLAY>, dim3(xrows, e->P), RNT, kRowSmem,
                  e->g, *sc,
                  row_arg(e), (const real*)e->Rsp, (const real*)e->HVb[e->hcur], (const real*)e->HVb[e->hcur ^ 1],
                  e->xi, (const real*)e->Y, SA, SB, K1Rows{});
}
#endif
#endif   // paired rows

// ============================================================================ pass A ==
#if LPC_MOD_PASSA
typedef SPlan<LPC_MOD_PASSA_RAD> PassAP;
typedef SPlanArg<PassAP> PassAPA;
// cp: the engine's pass-A descriptor with this call's mode, zero rows and scale already set (cols_passA)
static int m_cols_passA(Engine* e, const ColPass* cp, real2* S, int nplanes, int inverse, int kid) {
  constexpr int NT = LPC_MOD_PASSA_NT, EM = LPC_MOD_PASSA_EM, T = LPC_MOD_PASSA_T;
  const dim3 grid(cp->G * cp->ntile_c, nplanes);
  const size_t smem = (size_t)PassAP::n * (T + 2) * sizeof(real2);   // tile + the plan's and the four-step twiddles
  const PassAPA pa = splan_arg<PassAP>(e->planA);
  if (inverse) return launch_k(e, kid, k_cols<NT, EM, true, PassAPA, T, true>, grid, NT, smem, e->g, pa, *cp, S);
  return launch_k(e, kid, k_cols<NT, EM, false, PassAPA, T, true>, grid, NT, smem, e->g, pa, *cp, S);
}
#endif

// ============================================================ ADMM fused middle (LDS) ==
#if LPC_MOD_MID_KIND != 0
typedef SPlan<LPC_MOD_MID_RAD> MidP;
typedef SPlanArg<MidP> MidPA;
static int m_admm_mid(Engine* e, const ColPass* cp, const AdmmScalars* sc, real sb_outside_scale) {
  constexpr int NT = LPC_MOD_MID_NT, EM = LPC_MOD_MID_EM, T = LPC_MOD_MID_T;
  const PlaneGeom& g = e->g;
  real2* SA = e->S;
  real2* SB = e->S + (size_t)e->P * g.cplane;
  const MidPA pa = splan_arg<MidP>(e->planB);
  const real rscale = (real)1.0 / ((real)g.Hp * (real)g.Wp);
#if LPC_MOD_MID_KIND == LPC_MID_SEQ     // single-pass columns, one spectrum at a time through T columns
  // (LDS: the tile + the plan's twiddle table behind it)
  constexpr int PC = LPC_MOD_SLAY != 0 ? LPC_MOD_MID_PC : 0;      // precombined point-wise constants (k_mid_consts)
  return launch_k(e, LPC_K_COL_MID, k_cols_mid_admm_seq<NT, EM, MidPA, T, LPC_MOD_MID_MINW, LPC_MOD_MID_PRE != 0, LPC_MOD_SLAY, PC>,
                  dim3(cp->ntile_c * e->P), NT, (size_t)MidP::n * (T + 1) * sizeof(real2), g, pa, *cp, SA, SB,
                  PC ? (const real2*)e->midc : (const real2*)(LPC_MOD_SLAY ? e->Hs_t : e->Hs),
                  PC ? (const real*)e->midrd : (const real*)(LPC_MOD_SLAY ? e->Gabs_t : e->Gabs), (const real2*)e->phr, (const real2*)e->phc, sc->mu1,
                  sc->mu2, sc->mu3, rscale, sb_outside_scale);
#else                                   // both spectra side by side: [N][2 T]
  const FastDiv t2 = make_fastdiv((unsigned)(2 * T));
  return launch_k(e, LPC_K_COL_MID, k_cols_mid_admm<NT, EM, MidPA, 2 * T, true, LPC_MOD_SLAY>, dim3(cp->G * cp->ntile_c, e->P), NT,
                  (size_t)MidP::n * (2 * T + 1) * sizeof(real2), g, pa, *cp, SA, SB, (const real2*)(LPC_MOD_SLAY ? e->Hs_t : e->Hs),
                  (const real*)(LPC_MOD_SLAY ? e->Gabs_t : e->Gabs), (const real2*)e->phr, (const real2*)e->phc, t2, sc->mu1, sc->mu2, sc->mu3,
                  rscale, sb_outside_scale);
#endif
}
#endif

// ================================================================================ table ==
#if defined(LPC_STAMP)
// timing builds (lpc_rt.h: LPC_STAMP; tools/stamp_timeline.py): the stamps of the last launch of every stamped kernel
extern "C" int lpc_module_stamps(unsigned long long* dst, size_t bytes) {
  if (bytes > sizeof(lpc_stamp_buf)) bytes = sizeof(lpc_stamp_buf);
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(lpc_stamp_buf), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif
extern "C" int lpc_module_init(LpcModule* m, size_t engine_size, const char* src_fp) {
  if (!m || engine_size != sizeof(Engine) || !src_fp || std::strcmp(src_fp, LPC_SRC_FP) != 0) return 1;
  std::memset((void*)m, 0, sizeof(*m));
#if LPC_MOD_ROW_KIND == LPC_ROWS_HALF
  m->rows_fwd_single = m_rows_fwd_single;
  m->rows_inv_single = m_rows_inv_single;
#endif
#if LPC_MOD_ROW_KIND != 0 && LPC_MOD_FAMILY == LPC_FAM_ADMM
  m->admm_rows_fwd = m_admm_rows_fwd;
  m->admm_rows_inv = m_admm_rows_inv;
#if LPC_MOD_ROW_X
  m->admm_rows_fwd_x = m_admm_rows_fwd_x;
#if LPC_MOD_ROW_KIND == LPC_ROWS_PAIRED
  m->k1_rows = kK1Quads;
#endif
#endif
#endif
#if LPC_MOD_ROW_KIND == LPC_ROWS_HALF && LPC_MOD_FAMILY == LPC_FAM_GD
#ifndef LPC_DOUBLE
  m->gd_v2 = GdV2<RowP>::ok ? 1 : 0;
#endif
  m->gd_rows_mid = m_gd_rows_mid;
  m->gd_rows_update = m_gd_rows_update;
  m->gd_rows_update_fwd = m_gd_rows_update_fwd;
#endif
#if LPC_MOD_PASSA
  m->cols_passA = m_cols_passA;
#endif
#if LPC_MOD_MID_KIND != 0
  m->admm_mid = m_admm_mid;
#endif
  m->slay = LPC_MOD_SLAY;
#if LPC_MOD_MID_KIND == LPC_MID_SEQ
  m->mid_pc = LPC_MOD_SLAY != 0 ? LPC_MOD_MID_PC : 0;
#endif
  return 0;
}
