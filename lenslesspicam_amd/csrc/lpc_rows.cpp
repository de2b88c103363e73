// lpc_rows.cpp -- launches of every row pass (see lpc_engine.h for the split of the library)
#include "lpc_engine.h"

// ------------------------------------------------------------- 2-D transform pieces --
// forward rows of ONE real source (pairs of rows) into spectrum S (planes = nplanes)
int rows_fwd_single(Engine* e, const RealSrc& src, real2* S, int nplanes, int kid) {
  const PlaneGeom& g = e->g;
  if (e->rows_half && e->static_rows)
    return with_row_shape(e, [&](auto SHc) {
      using SH = decltype(SHc);
      return with_sk(e->static_sk, [&](auto SKc) {
      constexpr bool sk = decltype(SKc)::value;
        return launch_k(e, kid, k_rfwd_rows_half<SH::nt, SH::em, sk, SPlanArg<typename SH::plan>>, dim3(src.nrows, nplanes), SH::nt,
                        LPC_ROW_SMEM_BYTES(SH::plan::n, sk), g, splan_arg<typename SH::plan>(e->planWh), e->planW.tw, src, S);
      });
    });
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
  if (e->rows_half && e->static_rows)
    return with_row_shape(e, [&](auto SHc) {
      using SH = decltype(SHc);
      return with_sk(e->static_sk, [&](auto SKc) {
      constexpr bool sk = decltype(SKc)::value;
        return launch_k(e, kid, k_rinv_rows_half<SH::nt, SH::em, sk, SPlanArg<typename SH::plan>>, dim3(dst.nrows, nplanes), SH::nt,
                        LPC_ROW_SMEM_BYTES(SH::plan::n, sk), g, splan_arg<typename SH::plan>(e->planWh), e->planW.tw, S, dst);
      });
    });
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
  if (e->rows_half && e->static_rows)
    return with_row_shape(e, [&](auto SHc) {
      using SH = decltype(SHc);
      return with_sk(e->static_sk, [&](auto SKc) {
      constexpr bool sk = decltype(SKc)::value;
        return launch_k(e, LPC_K_ROW_FWD, k_rfwd_half<SH::nt, SH::em, sk, SPlanArg<typename SH::plan>>, dim3(2 * g.Hp, e->P), SH::nt,
                        LPC_ROW_SMEM_BYTES(SH::plan::n, sk), g, splan_arg<typename SH::plan>(e->planWh), e->planW.tw,
                      (const real*)e->Rsp, (const real*)e->Aarr, SA, SB);
      });
    });
  if (e->rows_half)
    return dispatch_row(g.Wp / 2, e->planWh.skew_ok, false, [&](auto NTc, auto EM, auto SK, auto) {
      constexpr int nt = decltype(NTc)::value, em = decltype(EM)::value;
      constexpr bool sk = decltype(SK)::value;
      return launch_k(e, LPC_K_ROW_FWD, k_rfwd_half<nt, em, sk>, dim3(2 * g.Hp, e->P), nt,
                      LPC_ROW_SMEM_BYTES(g.Wp / 2, sk), g, e->planWh, e->planW.tw, (const real*)e->Rsp,
                      (const real*)e->Aarr, SA, SB);
    });
  if (e->static_prow == 2048)    // 760 x 1014 frames: paired rows of 2048 points = 256 threads x 8
    return launch_k(e, LPC_K_ROW_FWD, k_rfwd_arrays<256, 8, true, false, SPlanArg<RowPlan2048>>, dim3(g.Hp, e->P), 256,
                    LPC_ROW_SMEM_BYTES(2048, true), g, splan_arg<RowPlan2048>(e->planW), (const real*)e->Rsp,
                    (const real*)e->Aarr, SA, SB);
  if (e->static_prow == 960)     // C1 / C4: paired rows of 960 points = 256 threads x 4 (3.75)
    return launch_k(e, LPC_K_ROW_FWD, k_rfwd_arrays<256, 4, true, false, SPlanArg<RowPlan960>>, dim3(g.Hp, e->P), 256,
                    LPC_ROW_SMEM_BYTES(960, true), g, splan_arg<RowPlan960>(e->planW), (const real*)e->Rsp,
                    (const real*)e->Aarr, SA, SB);
  return dispatch_row(g.Wp, e->planW.skew_ok, e->rows_r2, [&](auto NTc, auto EM, auto SK, auto R2) {
    constexpr int nt = decltype(NTc)::value, em = decltype(EM)::value;
    constexpr bool sk = decltype(SK)::value, r2 = decltype(R2)::value;
    return launch_k(e, LPC_K_ROW_FWD, k_rfwd_arrays<nt, em, sk, r2>, dim3(g.Hp, e->P), nt,
                    LPC_ROW_SMEM_BYTES(g.Wp, sk), g, e->planW, (const real*)e->Rsp, (const real*)e->Aarr, SA, SB);
  });
}

// ---- ADMM: forward rows of r_sp (stored) and of a = mu1 X - xi' (computed here from xi, HV, HV_old, y) -------------
int admm_rows_fwd_x(Engine* e, const AdmmScalars& sc) {
#ifdef LPC_DOUBLE
  (void)sc;
  return fail("internal: the X-half row kernel is float32-only");
#else
  const PlaneGeom& g = e->g;
  real2* SA = e->S;
  real2* SB = e->S + (size_t)e->P * g.cplane;
  if (!e->rows_half) {     // paired rows (static_prow): 960 = 256 threads x 4, 2048 = 256 x 8
    // sc.skipa: the window rows as usual + the rows of r_sp outside it two per transform (k_rfwd_arrays_x)
    const int xrows = sc.skipa ? g.H + outside_pair_count(g) : g.Hp;
    auto go = [&](auto plan_tag, auto em_tag) {
      using P = decltype(plan_tag);
      constexpr int em = decltype(em_tag)::value;
      return launch_k(e, LPC_K_ROW_FWD, k_rfwd_arrays_x<256, em, true, SPlanArg<P>>, dim3(xrows, e->P), 256,
                      LPC_ROW_SMEM_BYTES(P::n, true), g, sc, splan_arg<P>(e->planW), (const real*)e->Rsp,
                      (const real*)e->HVb[e->hcur], (const real*)e->HVb[e->hcur ^ 1], e->xi, (const real*)e->Y, SA, SB);
    };
    if (e->static_prow == 960 && e->prow_nt128)
      return launch_k(e, LPC_K_ROW_FWD, k_rfwd_arrays_x<128, 8, true, SPlanArg<RowPlan960>>, dim3(xrows, e->P), 128,
                      LPC_ROW_SMEM_BYTES(960, true), g, sc, splan_arg<RowPlan960>(e->planW), (const real*)e->Rsp,
                      (const real*)e->HVb[e->hcur], (const real*)e->HVb[e->hcur ^ 1], e->xi, (const real*)e->Y, SA, SB);
    if (e->static_prow == 960) return go(RowPlan960{}, std::integral_constant<int, 4>{});
    if (e->static_prow == 2048) return go(RowPlan2048{}, std::integral_constant<int, 8>{});
    return fail("internal: no static paired-row plan");
  }
  return with_row_shape(e, [&](auto SHc) {
    using SH = decltype(SHc);
    using PA = SPlanArg<typename SH::plan>;
    return with_sk(e->static_sk, [&](auto SKc) {
      constexpr bool sk = decltype(SKc)::value;
      return launch_k(e, LPC_K_ROW_FWD, k_admm_rows_fused<SH::nt, SH::em, sk, 1, 1, PA, false>, dim3(2 * g.Hp, e->P),
                      SH::nt, LPC_ROW_SMEM_BYTES(SH::plan::n, sk), g, sc, splan_arg<typename SH::plan>(e->planWh),
                      (const real2*)e->planW.tw, (const real*)nullptr, (const real*)e->Rsp,
                      (const real*)e->HVb[e->hcur], (const real*)e->HVb[e->hcur ^ 1], e->xi, (const real*)nullptr,
                      (const real*)nullptr, (real*)nullptr, (real*)nullptr, (real*)nullptr, (const real*)e->Y, SA, SB);
    });
  });
#endif
}

// ---- ADMM: the image-domain kernel fused into the forward rows (float32, half-length rows, Wp % 4 == 0) -------
int admm_rows_fused(Engine* e, const AdmmScalars& sc, const real* Vc, const real* Vo) {
#ifdef LPC_DOUBLE
  (void)sc; (void)Vc; (void)Vo;
  return fail("internal: the fused ADMM rows are float32-only");
#else
  const PlaneGeom& g = e->g;
  real2* SA = e->S;
  real2* SB = e->S + (size_t)e->P * g.cplane;
  auto launch = [&](auto kernel, int nt, size_t smem, auto plan_arg) {
    return launch_k(e, LPC_K_SPATIAL, kernel, dim3(2 * g.Hp, e->P), nt, smem, g, sc, plan_arg,
                    (const real2*)e->planW.tw, Vc, Vo, (const real*)e->HVb[e->hcur], (const real*)e->HVb[e->hcur ^ 1],
                    e->xi, (const real*)e->eta0[e->ecur], (const real*)e->eta1[e->ecur], e->eta0[e->ecur ^ 1],
                    e->eta1[e->ecur ^ 1], e->rho, (const real*)e->Y, SA, SB);
  };
  if (e->static_rows)
    return with_row_shape(e, [&](auto SHc) {
      using SH = decltype(SHc);
      using PA = SPlanArg<typename SH::plan>;
      return with_sk(e->static_sk, [&](auto SKc) {
        constexpr bool sk = decltype(SKc)::value;
        const size_t smem = LPC_ROW_SMEM_BYTES(SH::plan::n, sk);
        const PA pa = splan_arg<typename SH::plan>(e->planWh);
#ifdef LPC_DEBUG_KNOBS   // phase-timing experiments of profiles/r02_notes.md (results are garbage)
        if (std::getenv("LPC_DEBUG_FUSED_NOFFT")) return launch(k_admm_rows_fused<SH::nt, SH::em, sk, 1, 2, PA>, SH::nt, smem, pa);
        if (std::getenv("LPC_DEBUG_FUSED_NOSPATIAL")) return launch(k_admm_rows_fused<SH::nt, SH::em, sk, 1, 3, PA>, SH::nt, smem, pa);
#endif
        return launch(k_admm_rows_fused<SH::nt, SH::em, sk, 1, 1, PA>, SH::nt, smem, pa);
      });
    });
  return dispatch_row(g.Wp / 2, e->planWh.skew_ok, false, [&](auto NTc, auto EM, auto SK, auto) {
    constexpr int nt = decltype(NTc)::value, em = decltype(EM)::value;
    constexpr bool sk = decltype(SK)::value;
    return launch(k_admm_rows_fused<nt, em, sk>, nt, LPC_ROW_SMEM_BYTES(g.Wp / 2, sk), e->planWh);
  });
#endif
}

// ---- ADMM: the two work spectra -> V and H V (padded, no shift) ------------------------------------------------
int admm_rows_inv(Engine* e, real* Vout, real* HVout, bool skip_hv_outside) {
  const PlaneGeom& g = e->g;
  // paired rows, skip_hv_outside: the window rows as usual + the rows of V outside it two per transform (k_rinv_arrays)
  const int wo = (skip_hv_outside && !e->rows_half) ? 1 : 0;
  const int irows = wo ? g.H + outside_pair_count(g) : g.Hp;
  const int hrows = skip_hv_outside ? g.Hp + g.H : 2 * g.Hp;      // half-length rows: k_rinv_half
  real2* SA = e->S;
  real2* SB = e->S + (size_t)e->P * g.cplane;
  const Fft1dPlan& pinv = e->rows_r2 ? e->planWi : e->planW;
  if (e->rows_half && e->static_rows)
    return with_row_shape(e, [&](auto SHc) {
      using SH = decltype(SHc);
      return with_sk(e->static_sk, [&](auto SKc) {
      constexpr bool sk = decltype(SKc)::value;
        return launch_k(e, LPC_K_ROW_INV, k_rinv_half<SH::nt, SH::em, sk, SPlanArg<typename SH::plan>>, dim3(hrows, e->P), SH::nt,
                        LPC_ROW_SMEM_BYTES(SH::plan::n, sk), g, splan_arg<typename SH::plan>(e->planWh), e->planW.tw,
                      (const real2*)SA, (const real2*)SB, Vout, HVout, skip_hv_outside ? 1 : 0);
      });
    });
  if (e->rows_half)
    return dispatch_row(g.Wp / 2, e->planWh.skew_ok, false, [&](auto NTc, auto EM, auto SK, auto) {
      constexpr int nt = decltype(NTc)::value, em = decltype(EM)::value;
      constexpr bool sk = decltype(SK)::value;
      return launch_k(e, LPC_K_ROW_INV, k_rinv_half<nt, em, sk>, dim3(hrows, e->P), nt,
                      LPC_ROW_SMEM_BYTES(g.Wp / 2, sk), g, e->planWh, e->planW.tw, (const real2*)SA,
                      (const real2*)SB, Vout, HVout, skip_hv_outside ? 1 : 0);
    });
  if (e->static_prow == 2048)
    return launch_k(e, LPC_K_ROW_INV, k_rinv_arrays<256, 8, true, false, SPlanArg<RowPlan2048>>, dim3(irows, e->P), 256,
                    LPC_ROW_SMEM_BYTES(2048, true), g, splan_arg<RowPlan2048>(e->planW), (const real2*)SA, (const real2*)SB,
                    Vout, HVout, wo);
  if (e->static_prow == 960 && e->prow_nt128)
    return launch_k(e, LPC_K_ROW_INV, k_rinv_arrays<128, 8, true, false, SPlanArg<RowPlan960>>, dim3(irows, e->P), 128,
                    LPC_ROW_SMEM_BYTES(960, true), g, splan_arg<RowPlan960>(e->planW), (const real2*)SA, (const real2*)SB,
                    Vout, HVout, wo);
  if (e->static_prow == 960)
    return launch_k(e, LPC_K_ROW_INV, k_rinv_arrays<256, 4, true, false, SPlanArg<RowPlan960>>, dim3(irows, e->P), 256,
                    LPC_ROW_SMEM_BYTES(960, true), g, splan_arg<RowPlan960>(e->planW), (const real2*)SA, (const real2*)SB,
                    Vout, HVout, wo);
  return dispatch_row(g.Wp, pinv.skew_ok, e->rows_r2, [&](auto NTc, auto EM, auto SK, auto R2) {
    constexpr int nt = decltype(NTc)::value, em = decltype(EM)::value;
    constexpr bool sk = decltype(SK)::value, r2 = decltype(R2)::value;
    return launch_k(e, LPC_K_ROW_INV, k_rinv_arrays<nt, em, sk, r2>, dim3(irows, e->P), nt,
                    LPC_ROW_SMEM_BYTES(g.Wp, sk), g, pinv, (const real2*)SA, (const real2*)SB, Vout, HVout, wo);
  });
}
