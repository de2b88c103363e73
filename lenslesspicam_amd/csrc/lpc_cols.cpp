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
  cp.rev = (e->opt.rev_order & (inverse ? 4 : 2)) ? 1 : 0;
  if (!inverse && sb_outside_scale != (real)0.) {   // ADMM work spectra: planes [P, 2P) = SB, rows outside the window
    cp.sc_plane0 = e->P; cp.sc_r0 = g.sh; cp.sc_r1 = g.sh + g.H; cp.sc = sb_outside_scale;
  }
  if (inverse && crop_rows_only) {   // the row pass that follows reads spectrum rows (sh + u + Hp/2) mod Hp, u < H
    cp.need0 = (g.sh + g.Hp / 2) % g.Hp;
    cp.needn = g.H;
  }
  if (e->mod && e->mod->cols_passA) return e->mod->cols_passA(e, &cp, S, nplanes, inverse ? 1 : 0, kid);
  const dim3 grid(cp.G * cp.ntile_c, nplanes);
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
    PlaneGeom gl = g;
    gl.rev = (e->opt.gd_rev & 4) ? 1 : 0;       // PlaneGeom::rev
    return launch_k(e, LPC_K_COL_MID, kernel, rgrid, 64, 0, gl, e->planB, cp, S, (const real2*)e->Hs,
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
    // pairs of column tiles on one XCD: measured (profiles/r03_notes.md) -6 % on the 5-iteration C1 call, whose 8-column
    // tiles read half cache lines (middle 0.0278 -> 0.0228 ms); at 12 MP (16 columns = whole lines) it removes a third of
    // the middle's excess HBM reads (2.44 -> 2.28 GB against 1.91 GB asked for) but runs 3 % slower -- off there
    cp.ga = e->g_sep ? e->Ga : nullptr;
    cp.gb = e->g_sep ? e->Gb : nullptr;
    cp.rev = (e->opt.rev_order & 8) ? 1 : 0;
    cp.swz = e->opt.mid_swz >= 0 ? e->opt.mid_swz : ((size_t)cp.T * sizeof(real2) < 128 && !g.slay ? 1 : 0);
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
    else if (e->mod && e->mod->admm_mid) {   // compile-time plan in LDS: both spectra side by side, or one at a time
      if (e->midc && !(e->midc_valid && e->midc_par[0] == (double)sc.mu1 && e->midc_par[1] == (double)sc.mu2 &&
                       e->midc_par[2] == (double)sc.mu3)) {      // k_mid_consts: once per (PSF, step sizes)
        const long n = (long)((g.Hp + 1) & ~1) * g.cpitch;
        auto consts = [&](auto kernel) {
          return launch_k(e, -1, kernel, grid1d(n, 256, e->Ppsf), 256, 0, (const real2*)e->Hs_t, (const real*)e->Gabs_t,
                          cp.ga, cp.gb, (const real2*)e->phr, (const real2*)e->phc, g.Hp, g.Wc, g.cpitch, g.cplane, sc.mu1,
                          sc.mu2, sc.mu3, (real)1.0 / ((real)g.Hp * (real)g.Wp), e->midc, e->midrd);
        };
        if (e->mod->mid_pc == 2) LPC_OK(consts(k_mid_consts<256, true>));
        else LPC_OK(consts(k_mid_consts<256, false>));
        e->midc_par[0] = (double)sc.mu1; e->midc_par[1] = (double)sc.mu2; e->midc_par[2] = (double)sc.mu3;
        e->midc_valid = true;
      }
      LPC_OK(e->mod->admm_mid(e, &cp, &sc, (sc.skipa && !split) ? sc.mu1 * (real)g.Wp : (real)0.));
    }
    else if (cp.N * cp.T * 2 > 8192 && cp.N * cp.T * 2 <= 9216) {
      // just above 8192 points (C1 / C4: 540 rows x 8 columns x 2 arrays = 8640): 512 threads x 18 points keeps
      // TWO workgroups per CU inside the 128-VGPR budget; 1024 x 16 is one 16-wave workgroup per CU in lock-step
      // at every barrier (C4: middle 1.435 -> 1.331 ms, 17.5k -> 18.0k frame-it/s)
      LPC_OK(launch_k(e, LPC_K_COL_MID, k_cols_mid_admm<512, 18>, grid, 512, (size_t)cp.N * cp.T * 2 * sizeof(real2),
                      g, e->planB, cp, SA, SB, (const real2*)e->Hs, (const real*)e->Gabs, (const real2*)e->phr,
                      (const real2*)e->phc, t2, sc.mu1, sc.mu2, sc.mu3, (real)1.0 / ((real)g.Hp * (real)g.Wp), (real)0.));
    } else
    LPC_OK(dispatch_cfg(cp.N * cp.T * 2, [&](auto NTc, auto EM) {
      constexpr int nt = decltype(NTc)::value, em = decltype(EM)::value;
      return launch_k(e, LPC_K_COL_MID, k_cols_mid_admm<nt, em>, grid, nt,
                      (size_t)cp.N * cp.T * 2 * sizeof(real2), g, e->planB, cp, SA, SB, (const real2*)e->Hs,
                      (const real*)e->Gabs, (const real2*)e->phr, (const real2*)e->phc, t2, sc.mu1, sc.mu2,
                      sc.mu3, (real)1.0 / ((real)g.Hp * (real)g.Wp), (real)0.);
    }));
  }
  if (split) LPC_OK(cols_passA(e, e->S, 2 * e->P, true, 0, g.Hp, LPC_K_COL_A_INV));
  return 0;
}
