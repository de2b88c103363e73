// lpc_cols.cpp -- launches of every column pass (see lpc_engine.h for the split of the library)
#include "lpc_engine.h"

// column pass A (only when split) over nplanes planes; inverse => conj twiddles before FFT
int cols_passA(Engine* e, real2* S, int nplanes, bool inverse, int zr0, int zr1, int kid,
                      bool crop_rows_only, real sb_outside_scale) {
  if (e->N1 == 1) return 0;
  const PlaneGeom& g = e->g;
  ColPass cp = e->passA;
  cp.tw_mode = inverse ? 2 : 1;
  cp.zr0 = zr0; cp.zr1 = zr1;
  if (!inverse && sb_outside_scale != (real)0.) {   // ADMM work spectra: planes [P, 2P) = SB, rows outside the window
    cp.sc_plane0 = e->P; cp.sc_r0 = g.sh; cp.sc_r1 = g.sh + g.H; cp.sc = sb_outside_scale;
  }
  if (inverse && crop_rows_only) {   // the row pass that follows reads spectrum rows (sh + u + Hp/2) mod Hp, u < H
    cp.need0 = (g.sh + g.Hp / 2) % g.Hp;
    cp.needn = g.H;
  }
  const dim3 grid(cp.G * cp.ntile_c, nplanes);
  // compile-time plans: 32 columns per tile (512 threads: 128 x 32 = 512 x 8 points, 90 x 32 <= 512 x 6, 64 x 32 = 512 x 4),
  // or 16 (256 threads) when the engine kept the narrow tile (narrow frames, tuning knobs)
  auto static_passA = [&](auto plan_tag) {
    using P = decltype(plan_tag);
    const SPlanArg<P> pa = splan_arg<P>(e->planA);
    static const bool twl = getenv("LPC_NO_TW_LDS") == nullptr;   // plan + four-step twiddles staged in LDS
    if (cp.T == 32) {
      const size_t smem = (size_t)P::n * (32 + (twl ? 2 : 0)) * sizeof(real2);
      if (twl) {
        if (inverse) return launch_k(e, kid, k_cols<512, 8, true, SPlanArg<P>, 32, true>, grid, 512, smem, g, pa, cp, S);
        return launch_k(e, kid, k_cols<512, 8, false, SPlanArg<P>, 32, true>, grid, 512, smem, g, pa, cp, S);
      }
      if (inverse) return launch_k(e, kid, k_cols<512, 8, true, SPlanArg<P>, 32>, grid, 512, smem, g, pa, cp, S);
      return launch_k(e, kid, k_cols<512, 8, false, SPlanArg<P>, 32>, grid, 512, smem, g, pa, cp, S);
    }
    const size_t smem = (size_t)P::n * (16 + (twl ? 2 : 0)) * sizeof(real2);
    if (twl) {
      if (inverse) return launch_k(e, kid, k_cols<256, 8, true, SPlanArg<P>, 16, true>, grid, 256, smem, g, pa, cp, S);
      return launch_k(e, kid, k_cols<256, 8, false, SPlanArg<P>, 16, true>, grid, 256, smem, g, pa, cp, S);
    }
    if (inverse) return launch_k(e, kid, k_cols<256, 8, true, SPlanArg<P>, 16>, grid, 256, smem, g, pa, cp, S);
    return launch_k(e, kid, k_cols<256, 8, false, SPlanArg<P>, 16>, grid, 256, smem, g, pa, cp, S);
  };
  if (e->static_passA == 128) return static_passA(ColPlan128{});
  if (e->static_passA == 90) return static_passA(ColPlan90{});
  if (e->static_passA == 64) return static_passA(ColPlan64{});
  return dispatch_cfg(cp.N * cp.T, [&](auto NT, auto EM) {
    constexpr int nt = decltype(NT)::value, em = decltype(EM)::value;
    const size_t smem = (size_t)cp.N * cp.T * sizeof(real2);
    if (inverse) return launch_k(e, kid, k_cols<nt, em, true>, grid, nt, smem, g, e->planA, cp, S);
    return launch_k(e, kid, k_cols<nt, em, false>, grid, nt, smem, g, e->planA, cp, S);
  });
}

// plain forward pass B (setup transforms only)
int cols_passB_fwd(Engine* e, real2* S, int nplanes, int zr0, int zr1) {
  const PlaneGeom& g = e->g;
  ColPass cp = e->passB;
  cp.tw_mode = 0;
  cp.zr0 = zr0; cp.zr1 = zr1;
  const dim3 grid(cp.G * cp.ntile_c, nplanes);
  return dispatch_cfg(cp.N * cp.T, [&](auto NT, auto EM) {
    constexpr int nt = decltype(NT)::value, em = decltype(EM)::value;
    return launch_k(e, -1, k_cols<nt, em, false>, grid, nt, (size_t)cp.N * cp.T * sizeof(real2), g, e->planB,
                    cp, S);
  });
}

// middle of a convolution on S (nplanes): [A] -> B fwd * H * B inv -> [A inv]
int conv_middle(Engine* e, real2* S, int nplanes, bool adjoint, int zr0, int zr1,
                       bool crop_rows_only) {
  const PlaneGeom& g = e->g;
  const bool split = e->N1 > 1;
  if (split) LPC_OK(cols_passA(e, S, nplanes, false, zr0, zr1, LPC_K_COL_A_FWD));
  ColPass cp = e->passB;
  cp.zr0 = split ? 0 : zr0;
  cp.zr1 = split ? g.Hp : zr1;
  const dim3 grid(cp.G * cp.ntile_c, nplanes);
  const real hscale = (real)1.0 / ((real)g.Hp * (real)g.Wp);
  // one lane = one whole pass-B column transform in registers, for the lengths choose_split produces most
  auto reg_mid = [&](auto kernel) {
    const dim3 rgrid((g.Wc + 63) / 64, cp.G, nplanes);
    return launch_k(e, LPC_K_COL_MID, kernel, rgrid, 64, 0, g, e->planB, cp, S, (const real2*)e->Hs,
                    adjoint ? 1 : 0, hscale, e->Ppsf);
  };
  const int regN = (split && e->mid_reg) ? cp.N : 0;
  if (regN == 48) { LPC_OK(reg_mid(k_cols_mid_mul_reg<8, 6>)); }
  else if (regN == 40) { LPC_OK(reg_mid(k_cols_mid_mul_reg<8, 5>)); }
  else if (regN == 36) { LPC_OK(reg_mid(k_cols_mid_mul_reg<6, 6>)); }
  else if (regN == 32) { LPC_OK(reg_mid(k_cols_mid_mul_reg<8, 4>)); }
  else if (regN == 30) { LPC_OK(reg_mid(k_cols_mid_mul_reg<6, 5>)); }
  else if (regN == 24) { LPC_OK(reg_mid(k_cols_mid_mul_reg<8, 3>)); }
  else
  LPC_OK(dispatch_cfg(cp.N * cp.T, [&](auto NT, auto EM) {
    constexpr int nt = decltype(NT)::value, em = decltype(EM)::value;
    return launch_k(e, LPC_K_COL_MID, k_cols_mid_mul<nt, em>, grid, nt, (size_t)cp.N * cp.T * sizeof(real2), g,
                    e->planB, cp, S, (const real2*)e->Hs, adjoint ? 1 : 0, hscale, e->Ppsf);
  }));
  if (split) LPC_OK(cols_passA(e, S, nplanes, true, 0, g.Hp, LPC_K_COL_A_INV, crop_rows_only));
  return 0;
}

// ---- ADMM: [pass A] -> fused middle (V-hat, H V-hat) -> [inverse pass A] on the two work spectra ----------------
int admm_cols(Engine* e, const AdmmScalars& sc) {
  const PlaneGeom& g = e->g;
  real2* SA = e->S;
  real2* SB = e->S + (size_t)e->P * g.cplane;
  const bool split = e->N1 > 1;
  // sc.skipa: the rows of SB outside the sensor window were not re-transformed, they still hold what the last inverse
  // row pass consumed = rfft(HV row) / Wp; a = mu1 HV there
  if (split) LPC_OK(cols_passA(e, e->S, 2 * e->P, false, 0, g.Hp, LPC_K_COL_A_FWD, false,
                               sc.skipa ? sc.mu1 * (real)g.Wp : (real)0.));
  {
    ColPass cp = e->passB;
    const dim3 grid(cp.G * cp.ntile_c, e->P);
    const FastDiv t2 = make_fastdiv((unsigned)(2 * cp.T));
    auto reg_mid = [&](auto kernel) {
      const dim3 rgrid((g.Wc + 63) / 64, cp.G, e->P);
      return launch_k(e, LPC_K_COL_MID, kernel, rgrid, 64, 0, g, e->planB, cp, SA, SB, (const real2*)e->Hs,
                      (const real*)e->Gabs, (const real2*)e->phr, (const real2*)e->phc, sc.mu1, sc.mu2, sc.mu3,
                      (real)1.0 / ((real)g.Hp * (real)g.Wp));
    };
    const int regN = (split && e->mid_reg && sizeof(real) == 4) ? cp.N : 0;
    // two arrays per lane: only short pass-B transforms fit the register file.  Measured at 12 MP
    // (profiles/r01b_notes.md): 24 points 0.89 ms and 32 points 0.83 ms beat the LDS middle (0.99 / 0.92 ms) but
    // need a 256- / 192-point pass A that costs more than it saves; 48 points is 1.62 ms (AGPR traffic).
    if (regN == 24) { LPC_OK(reg_mid(k_cols_mid_admm_reg<8, 3>)); }
    else if (e->static_mid == 541) {   // C1 / C4, one spectrum at a time: 540 points x 16 columns = 512 threads x 17
      // three stages 6.10.9 inside a 128-register budget: TWO workgroups per CU (2 x 69 KiB of LDS) overlap one
      // another's loads and barriers -- 0.650 ms per launch at 64 frames against 0.84 ms for the two-stage 30.18
      // plan (184 registers, one workgroup per CU) and 0.95 ms for 6.6.5.3 (profiles/r02_notes.md section 4)
      constexpr int minw = sizeof(real) == 4 ? 4 : 1;
      const real sbsc = sc.skipa ? sc.mu1 * (real)g.Wp : (real)0.;   // AdmmScalars::skipa: rows of SB kept from the last inverse rows
      static const bool twlds = getenv("LPC_NO_TW_LDS") == nullptr;
      if (twlds)
      LPC_OK(launch_k(e, LPC_K_COL_MID, k_cols_mid_admm_seq<512, 18, SPlanArg<ColPlan540Seq>, 16, minw, true>,
                      dim3(cp.ntile_c, e->P), 512, (size_t)540 * 17 * sizeof(real2), g,
                      splan_arg<ColPlan540Seq>(e->planB), cp, SA, SB, (const real2*)e->Hs, (const real*)e->Gabs,
                      (const real2*)e->phr, (const real2*)e->phc, sc.mu1, sc.mu2, sc.mu3,
                      (real)1.0 / ((real)g.Hp * (real)g.Wp), sbsc));
      else
      LPC_OK(launch_k(e, LPC_K_COL_MID, k_cols_mid_admm_seq<512, 18, SPlanArg<ColPlan540Seq>, 16, minw>,
                      dim3(cp.ntile_c, e->P), 512, (size_t)540 * 16 * sizeof(real2), g,
                      splan_arg<ColPlan540Seq>(e->planB), cp, SA, SB, (const real2*)e->Hs, (const real*)e->Gabs,
                      (const real2*)e->phr, (const real2*)e->phc, sc.mu1, sc.mu2, sc.mu3,
                      (real)1.0 / ((real)g.Hp * (real)g.Wp), sbsc));
    }
    else if (e->static_mid == 540) {   // C1 / C4: 540 points x 2 x 8 tile columns = 8640 points = 512 threads x 17
      LPC_OK(launch_k(e, LPC_K_COL_MID, k_cols_mid_admm<512, 18, SPlanArg<ColPlan540>, 16, true>, grid, 512,
                      (size_t)540 * 17 * sizeof(real2), g, splan_arg<ColPlan540>(e->planB), cp, SA, SB,
                      (const real2*)e->Hs, (const real*)e->Gabs, (const real2*)e->phr, (const real2*)e->phc, t2,
                      sc.mu1, sc.mu2, sc.mu3, (real)1.0 / ((real)g.Hp * (real)g.Wp)));
    }
    else if (e->static_mid == 48) {   // 48-point middle, 2 x 16 tile columns: 1536 points = 256 threads x 6
      static const bool twl = getenv("LPC_NO_TW_LDS") == nullptr;
      if (twl)
      LPC_OK(launch_k(e, LPC_K_COL_MID, k_cols_mid_admm<256, 8, SPlanArg<ColPlan48>, 32, true>, grid, 256,
                      (size_t)48 * 33 * sizeof(real2), g, splan_arg<ColPlan48>(e->planB), cp, SA, SB,
                      (const real2*)e->Hs, (const real*)e->Gabs, (const real2*)e->phr, (const real2*)e->phc, t2,
                      sc.mu1, sc.mu2, sc.mu3, (real)1.0 / ((real)g.Hp * (real)g.Wp)));
      else
      LPC_OK(launch_k(e, LPC_K_COL_MID, k_cols_mid_admm<256, 8, SPlanArg<ColPlan48>, 32>, grid, 256,
                      (size_t)48 * 32 * sizeof(real2), g, splan_arg<ColPlan48>(e->planB), cp, SA, SB,
                      (const real2*)e->Hs, (const real*)e->Gabs, (const real2*)e->phr, (const real2*)e->phc, t2,
                      sc.mu1, sc.mu2, sc.mu3, (real)1.0 / ((real)g.Hp * (real)g.Wp)));
    }
    else if (cp.N * cp.T * 2 > 8192 && cp.N * cp.T * 2 <= 9216) {
      // just above 8192 points (C1 / C4: 540 rows x 8 columns x 2 arrays = 8640): 512 threads x 18 points keeps
      // TWO workgroups per CU inside the 128-VGPR budget; 1024 x 16 is one 16-wave workgroup per CU in lock-step
      // at every barrier (C4: middle 1.435 -> 1.331 ms, 17.5k -> 18.0k frame-it/s)
      LPC_OK(launch_k(e, LPC_K_COL_MID, k_cols_mid_admm<512, 18>, grid, 512, (size_t)cp.N * cp.T * 2 * sizeof(real2),
                      g, e->planB, cp, SA, SB, (const real2*)e->Hs, (const real*)e->Gabs, (const real2*)e->phr,
                      (const real2*)e->phc, t2, sc.mu1, sc.mu2, sc.mu3, (real)1.0 / ((real)g.Hp * (real)g.Wp)));
    } else
    LPC_OK(dispatch_cfg(cp.N * cp.T * 2, [&](auto NTc, auto EM) {
      constexpr int nt = decltype(NTc)::value, em = decltype(EM)::value;
      return launch_k(e, LPC_K_COL_MID, k_cols_mid_admm<nt, em>, grid, nt,
                      (size_t)cp.N * cp.T * 2 * sizeof(real2), g, e->planB, cp, SA, SB, (const real2*)e->Hs,
                      (const real*)e->Gabs, (const real2*)e->phr, (const real2*)e->phc, t2, sc.mu1, sc.mu2,
                      sc.mu3, (real)1.0 / ((real)g.Hp * (real)g.Wp));
    }));
  }
  if (split) LPC_OK(cols_passA(e, e->S, 2 * e->P, true, 0, g.Hp, LPC_K_COL_A_INV));
  return 0;
}
