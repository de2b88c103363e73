// lpc_gd_v2_kernels.h -- the gradient-descent family's two fused row kernels, second form (round 5; compile-time plans,
// one real row per half-length transform, float32).
//
//   k_gd_resid_v2        spectrum row -> irfft -> shift + crop -> - y -> re-pad -> rfft -> spectrum row
//   k_gd_update_fwd_v2   spectrum row -> irfft -> shift + crop = gradient -> fused update of x (+ momentum, projection)
//                        -> the updated row, re-padded -> rfft -> spectrum row of the next iteration's H x
// (gd.py:128-134,183-188,235-241; rfft_convolve.py:145-170,190-216 -- the same dataflow as k_rinv_gd_mid_half /
// k_rinv_gd_update_fwd_half in lpc_gd_kernels.h, which stay the kernels of every other geometry.)
//
// What the first form spent its time on, read off its assembly (profiles/r05_notes.md section 1): the source functor of
// the forward transform's first stage sat behind per-element branches, so each of a lane's 16 loads of y (of x and the
// auxiliary state in the update kernel) was followed by `s_waitcnt vmcnt(0)` -- sixteen HBM latencies in a row on the
// critical path of every row; 1371 of 2258 VALU instructions were index arithmetic; a 64-bit scalar division (the data
// plane of a state plane) ran between the two transforms; and half of the 512 lanes had no butterfly in any stage.
//
// Here a workgroup is M / R lanes (4096-point rows: 256 lanes x 16 points, 128 VGPRs, four workgroups per CU = the same
// sixteen working waves per CU as before without the sixteen idle ones) and every lane owns butterfly j of every stage:
//   * the first inverse stage takes its inputs straight from global memory: lane j needs Z[k], k = j + (M/R) m, and the
//     Hermitian tangling needs X[k] and X[M - k] -- both are loaded by the lane itself (the mirror is the coalesced
//     descending run of lane M/R - j; every spectrum element is read by two lanes of the workgroup, once from HBM);
//   * the LAST inverse stage leaves lane j with samples (pairs) j + (M/R) m -- and the ifftshift by Wp / 2 samples = M / 2
//     pairs = R / 2 butterfly strides maps pair (j, m) to (j, m + R/2 mod R): the lane that produced a sample is the lane
//     whose first FORWARD butterfly consumes it.  Residual / update happen in registers; the tile makes no trip through
//     LDS between the two transforms (one write, one read and two barriers less per row);
//   * every global load of a phase is issued before the first use (unconditional, clamped addresses): y right after the
//     first inverse stage, so that two stages hide its latency; x / aux in front of the last inverse stage;
//   * M is a compile-time constant (the plan), so is everything derived from it; the data plane comes from two
//     reciprocal multiplications (FastDiv) prepared by the launcher.
// LDS trips per row: 5 writes + 5 reads of the tile instead of 7 + 7; barriers 9 instead of 13.
// Needs: every radix of the plan equal (R = 8 | 16: 4096 = 16.16.16, 512 = 8.8.8), M / R a multiple of 64, and the
// `pair` geometry (window offset and frame width even: 8-byte accesses to y / x) -- the launcher checks.
#pragma once
#include "lpc_gd_kernels.h"

#ifndef LPC_DOUBLE

// Timing-only knock-outs (a plan module compiled by hand with -DLPC_V2_KNOCK_MASK=n and loaded through option module_dir,
// tools/knock_modules.py; the results are garbage by construction): which resource do these kernels wait for?
// bit 0: no butterflies / twiddle products; bit 1: no LDS traffic between the stages; bit 2: no y / x / aux accesses;
// bit 3: no barriers.  A knocked-out operation sits behind a predicate that is false at run time but opaque to the
// compiler, so that everything feeding it and depending on it stays in the instruction stream.
#ifndef LPC_V2_KNOCK_MASK
#define LPC_V2_KNOCK_MASK 0
#endif
static __device__ __forceinline__ bool v2_live(int bit) { return !(LPC_V2_KNOCK_MASK & bit) || lpc_opaque(0) != 0; }
template <int R, bool INV>
static __device__ __forceinline__ void v2_dft(real2* v) {
  if (v2_live(1)) Dft<R, INV>::run(v);
}
static __device__ __forceinline__ void v2_lds_st(real2* s, int slot, real2 v) {
  if (v2_live(2)) s[slot] = v;
}
static __device__ __forceinline__ real2 v2_lds_ld(const real2* s, int slot) {
  return v2_live(2) ? s[slot] : make_real2((real)slot, (real)1.);
}
static __device__ __forceinline__ void v2_barrier() {
  if (v2_live(8)) __syncthreads();
}

template <class P>
struct GdV2 {
  static constexpr int M = P::n, R = P::radix(0), NB = P::n / P::radix(0), L = P::nst;
  static constexpr bool uniform() {
    for (int st = 0; st < P::nst; ++st)
      if (P::radix(st) != P::radix(0)) return false;
    return true;
  }
  static constexpr bool ok = uniform() && (R == 8 || R == 16) && L >= 2 && NB % 64 == 0 && NB <= 1024;
};

// The tangling twiddles of a lane, bins j + (M / R) m, m < R / 2 (the upper half is a swap away: w^(k + M/2) = -i w^k).
// TC: ONE table entry, the others as products with exp(-2 pi i m / (2 R)) (M / R bins of a length-2M table are 1 / (2 R) of
// a turn apart) -- the products are free here (removing every butterfly of these kernels does not change their time), the
// loads are not (profiles/r05_notes.md section 2); !TC: the R / 2 table entries themselves.
#ifndef LPC_V2_TC_RESID
#define LPC_V2_TC_RESID 1
#endif
#ifndef LPC_V2_TC_UPDATE
#define LPC_V2_TC_UPDATE 1
#endif
template <int N2R> struct V2Rot;
template <> struct V2Rot<32> { static constexpr WPair w[8] = {{(real)1.00000000000000000000e+00, (real)-0.00000000000000000000e+00}, {(real)9.80785280403230430579e-01, (real)-1.95090322016128248084e-01}, {(real)9.23879532511286738483e-01, (real)-3.82683432365089781779e-01}, {(real)8.31469612302545235671e-01, (real)-5.55570233019602177649e-01}, {(real)7.07106781186547572737e-01, (real)-7.07106781186547461715e-01}, {(real)5.55570233019602288671e-01, (real)-8.31469612302545235671e-01}, {(real)3.82683432365089837290e-01, (real)-9.23879532511286738483e-01}, {(real)1.95090322016128331351e-01, (real)-9.80785280403230430579e-01}}; };
template <> struct V2Rot<16> { static constexpr WPair w[4] = {{(real)1.00000000000000000000e+00, (real)-0.00000000000000000000e+00}, {(real)9.23879532511286738483e-01, (real)-3.82683432365089781779e-01}, {(real)7.07106781186547572737e-01, (real)-7.07106781186547461715e-01}, {(real)3.82683432365089837290e-01, (real)-9.23879532511286738483e-01}}; };
template <class P, bool TC>
static __device__ __forceinline__ void v2_tangle_twiddles(const real2* LPC_RESTRICT twW, int j, real2* tw) {
  constexpr int R = GdV2<P>::R, NB = GdV2<P>::NB;
  if (TC) {
    const real2 t0 = twW[j];
    tw[0] = t0;
#pragma unroll
    for (int m = 1; m < R / 2; ++m) tw[m] = cmul(t0, make_real2(V2Rot<2 * R>::w[m].re, V2Rot<2 * R>::w[m].im));
  } else {
#pragma unroll
    for (int m = 0; m < R / 2; ++m) tw[m] = twW[j + NB * m];
  }
}

// X[k], X[M - k] of one half-spectrum row -> Z (irfft semantics, see tangle_half_load) -> first inverse stage
// (radix R, no twiddles) -> tile.  No trailing barrier.
template <class P, int SK, bool TC>
static __device__ __forceinline__ void v2_load_tangle_first(real2* s, const real2* LPC_RESTRICT in,
                                                            const real2* LPC_RESTRICT twW, int j) {
  constexpr int M = GdV2<P>::M, R = GdV2<P>::R, NB = GdV2<P>::NB;
  real2 zk[R], zm[R], tw[R / 2];
#pragma unroll
  for (int m = 0; m < R; ++m) zk[m] = in[j + NB * m];
#pragma unroll
  for (int m = 0; m < R; ++m) zm[m] = in[M - j - NB * m];
  v2_tangle_twiddles<P, TC>(twW, j, tw);
  real2 v[R];
#pragma unroll
  for (int m = 0; m < R; ++m) {
    // (w^(k + M/2) = -i w^k, w = exp(-2 pi i / Wp): the upper half of a lane's twiddles is a swap away from the lower)
    const real2 t = m < R / 2 ? tw[m] : cmul_mi(tw[m - R / 2]);
    real2 a = zk[m], b = zm[m];
    if (m == 0) {   // k == 0: the imaginary parts of the DC and Nyquist bins are ignored
      a.y = j == 0 ? (real)0. : a.y;
      b.y = j == 0 ? (real)0. : b.y;
    }
    const real2 e = make_real2(a.x + b.x, a.y - b.y);
    const real2 d = make_real2(a.x - b.x, a.y + b.y);
    const real2 od = cmul_conj(d, t);
    v[m] = make_real2(e.x - od.y, e.y + od.x);
  }
  v2_dft<R, true>(v);
#pragma unroll
  for (int m = 0; m < R; ++m) v2_lds_st(s, lds_slot<SK>(j * R + m), v[m]);
}

// ---- stage twiddles, loaded ONE STAGE AHEAD ------------------------------------------------------------------------
// twiddle_mul() loads the base powers w, w^2, w^4 (, w^8) of its butterfly inside the stage, i.e. behind the barrier that
// opens it: an L2 round trip on the critical path of every twiddled stage (four per row), with four waves per SIMD to
// hide it -- the counters show the waves of these kernels waiting on vector memory for 45 % of their lifetime
// (profiles/r05_notes.md).  Here the base powers of stage ST + 1 are requested while stage ST still computes.
// v2_tw_apply is twiddle_mul's arithmetic on the loaded values: same products in the same order.
template <class P, int ST>
static __device__ __forceinline__ void v2_tw_load(const real2* LPC_RESTRICT tws, int j, real2* w) {
  constexpr int R = GdV2<P>::R, NB = P::n / R;
  // two 16-byte loads, contiguous across the wave, from the lane-ordered table (lpc_sfft.h: SPlan::tws_off; the values
  // are the table entries twiddle_mul() gathers).  An opaque copy of the lane index: the inverse and the forward transform
  // request the same entries, and the compiler would rather keep the addresses alive -- spilled -- than recompute them.
  const real2* b = tws + P::tws_off(ST) + 2 * lpc_opaque(j);
  const real4_t lo = *(const real4_t*)b, hi = *(const real4_t*)(b + 2 * NB);
  w[0] = make_real2(lo.x, lo.y); w[1] = make_real2(lo.z, lo.w);
  w[2] = make_real2(hi.x, hi.y);
  if (R == 16) w[3] = make_real2(hi.z, hi.w);
}
template <int R, bool INV>
static __device__ __forceinline__ void v2_tw_apply(real2* v, const real2* wb) {
  if (!v2_live(1)) { v[1].x += wb[0].x + wb[1].x + wb[2].x + wb[R == 16 ? 3 : 0].x; return; }
  real2 w[16];
  w[1] = wb[0]; w[2] = wb[1]; w[4] = wb[2];
  if (R == 16) w[8] = wb[3];
  w[3] = cmul(w[1], w[2]); w[5] = cmul(w[1], w[4]); w[6] = cmul(w[2], w[4]); w[7] = cmul(w[3], w[4]);
  if (R == 16) {
#pragma unroll
    for (int m = 9; m < 16; ++m) w[m] = cmul(w[m - 8], w[8]);
  }
#pragma unroll
  for (int m = 1; m < R; ++m) v[m] = INV ? cmul_conj(v[m], w[m]) : cmul(v[m], w[m]);
}

// stage ST (0 < ST < L - 1) in place in the tile, twiddles in wb; ends with a barrier (sfft_stage with one butterfly per
// lane)
template <class P, int ST, int SK, bool INV>
static __device__ __forceinline__ void v2_mid_stage(real2* s, int j, const real2* wb) {
  constexpr int R = GdV2<P>::R, NB = GdV2<P>::NB, NS = P::ns(ST);
  real2 v[R];
#pragma unroll
  for (int m = 0; m < R; ++m) v[m] = v2_lds_ld(s, lds_slot<SK>(j + NB * m));
  v2_tw_apply<R, INV>(v, wb);
  v2_dft<R, INV>(v);
  const int oi = (j / NS) * NS * R + j % NS;
  v2_barrier();
#pragma unroll
  for (int m = 0; m < R; ++m) v2_lds_st(s, lds_slot<SK>(oi + m * NS), v[m]);
  v2_barrier();
}

// stages 1 .. L - 2 of a transform, in the tile.  On entry wb holds the base twiddles of stage 1, on exit those of
// stage L - 1: each stage requests its successor's before its own arithmetic.
template <class P, int ST, int SK, bool INV>
static __device__ __forceinline__ void v2_mid_one(real2* s, const real2* LPC_RESTRICT tw, int j, real2* wb) {
  real2 wn[4];
  v2_tw_load<P, ST + 1>(tw, j, wn);
  v2_mid_stage<P, ST, SK, INV>(s, j, wb);
#pragma unroll
  for (int i = 0; i < 4; ++i) wb[i] = wn[i];
}
template <class P, int SK, bool INV, int... I>
static __device__ __forceinline__ void v2_mid_chain(real2* s, const real2* LPC_RESTRICT tw, int j, real2* wb,
                                                    std::integer_sequence<int, I...>) {
  (v2_mid_one<P, 1 + I, SK, INV>(s, tw, j, wb), ...);
}
// last stage (ST = L - 1: NS = NB, butterfly j): inputs from the tile, outputs (elements j + NB m) stay in v[].  NEXT: the
// base twiddles of stage 1 of the transform that follows are requested in front of the arithmetic and returned in wb.
template <class P, int SK, bool INV, bool NEXT>
static __device__ __forceinline__ void v2_final_stage(const real2* s, const real2* LPC_RESTRICT tw, int j, real2* wb,
                                                      real2* v) {
  constexpr int R = GdV2<P>::R, NB = GdV2<P>::NB;
  real2 wn[4];
  if (NEXT) v2_tw_load<P, 1>(tw, j, wn);
#pragma unroll
  for (int m = 0; m < R; ++m) v[m] = v2_lds_ld(s, lds_slot<SK>(j + NB * m));
  v2_tw_apply<R, INV>(v, wb);
  v2_dft<R, INV>(v);
  if (NEXT) {
#pragma unroll
    for (int i = 0; i < 4; ++i) wb[i] = wn[i];
  }
}

// first forward stage (radix R, no twiddles) from registers -> tile; trailing barrier
template <class P, int SK>
static __device__ __forceinline__ void v2_first_fwd(real2* s, int j, real2* r) {
  constexpr int R = GdV2<P>::R;
  v2_dft<R, false>(r);
#pragma unroll
  for (int m = 0; m < R; ++m) v2_lds_st(s, lds_slot<SK>(j * R + m), r[m]);
  v2_barrier();
}

// X = FFT_M(z) (elements j + NB m in x[]) -> half spectrum of the real row (see untangle_half_store) -> o[0 .. M].
// Precondition: every lane is done reading the tile.
template <class P, int SK, bool TC>
static __device__ __forceinline__ void v2_untangle_store(real2* s, const real2* LPC_RESTRICT twW, int j, const real2* x,
                                                         real2* LPC_RESTRICT o) {
  constexpr int M = GdV2<P>::M, R = GdV2<P>::R, NB = GdV2<P>::NB;
  // (the addresses below are those of the tangling at the top of the kernel: recomputed from an opaque copy of the lane
  // index, a few integer instructions, instead of kept alive -- i.e. spilled -- across both transforms)
  j = lpc_opaque(j);
  real2 tw[R / 2];
  v2_tangle_twiddles<P, TC>(twW, j, tw);
#pragma unroll
  for (int m = 0; m < R; ++m) v2_lds_st(s, lds_slot<SK>(j + NB * m), x[m]);
  v2_barrier();
  real2 xm[R];
#pragma unroll
  for (int m = 0; m < R; ++m) {
    const int km = (m == 0 && j == 0) ? 0 : M - j - NB * m;    // the DC bin pairs with itself
    xm[m] = v2_lds_ld(s, lds_slot<SK>(km));
  }
#pragma unroll
  for (int m = 0; m < R; ++m) {
    const real2 zk = x[m], zm = xm[m];
    const real ex = (real)0.5 * (zk.x + zm.x), ey = (real)0.5 * (zk.y - zm.y);
    const real2 od = make_real2((real)0.5 * (zk.y + zm.y), (real)-0.5 * (zk.x - zm.x));
    const real2 wo = cmul(m < R / 2 ? tw[m] : cmul_mi(tw[m - R / 2]), od);
    o[j + NB * m] = make_real2(ex + wo.x, ey + wo.y);
    if (m == 0 && j == 0) o[M] = make_real2(ex - wo.x, wo.y - ey);    // the Nyquist bin
  }
}

// data plane of state plane pl: (pl / DC) * C + pl % C
static __device__ __forceinline__ int v2_data_plane(unsigned pl, FastDiv fdc, FastDiv fc, int C) {
  const unsigned q = fd_div(pl, fdc);
  return (int)(q * (unsigned)C + (pl - fd_div(pl, fc) * fc.d));
}

// (five waves per SIMD = at most 96 VGPRs: with the natural tile, 32 KB, five residual workgroups share a CU; same box,
// 0.1678 -> 0.1596 ms per launch at 12 MP, profiles/r05o_tc.log)
#ifndef LPC_V2_RESID_MINW
#define LPC_V2_RESID_MINW 5
#endif
template <int NT, int SK, class PL>
__global__ __launch_bounds__(NT, LPC_V2_RESID_MINW) void k_gd_resid_v2(PlaneGeom g, PL plan, const real2* LPC_RESTRICT twW,
                                                               const real2* LPC_RESTRICT Sin,
                                                               real2* LPC_RESTRICT Sout, const real* LPC_RESTRICT Y,
                                                               FastDiv fdc, FastDiv fc) {
  using P = typename PL::plan;
  constexpr int M = GdV2<P>::M, R = GdV2<P>::R, NB = GdV2<P>::NB, L = GdV2<P>::L;
  static_assert(GdV2<P>::ok && NT == NB, "k_gd_resid_v2: one butterfly per lane and stage");
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int j = LPC_TID(NT), u = (int)LPC_BX(g);
  const unsigned pl = LPC_BY(g);
  const int sr = wrap_add(g.sh + u, g.Hp / 2, g.Hp);
  real2 wb[4];
  v2_tw_load<P, 1>(plan.tws, j, wb);
  v2_load_tangle_first<P, SK, LPC_V2_TC_RESID != 0>(s, Sin + (long)pl * g.cplane + (long)sr * g.cpitch, twW, j);
  v2_barrier();
  v2_mid_chain<P, SK, true>(s, plan.tws, j, wb, std::make_integer_sequence<int, L - 2>{});
  // the measurement row, in flight across the last inverse stage (across two stages it costs the registers that the
  // twiddle prefetch needs): padded pair i = j + NB m covers columns 2 i - sw, 2 i - sw + 1 of the frame; outside the
  // window the loads return zero (lpc_make_rsrc)
  const lpc_rsrc yr = lpc_make_rsrc(Y + (long)v2_data_plane(pl, fdc, fc, g.C) * g.uplane + (long)u * g.W,
                                    (unsigned)g.W * (unsigned)sizeof(real));
  real2 yy[R];
#pragma unroll
  for (int m = 0; m < R; ++m)
    yy[m] = !v2_live(4) ? make_real2((real)m, (real)j) : lpc_buf_load2(yr, lpc_opaque((2 * (j + NB * m) - g.sw) * (int)sizeof(real)));
  real2 v[R], r[R];
  v2_final_stage<P, SK, true, true>(s, plan.tws, j, wb, v);
  // conv pair (j, m) is pair (j, m + R/2 mod R) of the shifted row: residual inside the window, zero outside
#pragma unroll
  for (int m = 0; m < R; ++m) {
    const real2 z = v[(m + R / 2) % R];
    r[m] = (unsigned)(2 * (j + NB * m) - g.sw) < (unsigned)g.W ? make_real2(z.x - yy[m].x, z.y - yy[m].y)
                                                                : make_real2((real)0., (real)0.);
  }
  v2_barrier();
  v2_first_fwd<P, SK>(s, j, r);
  v2_mid_chain<P, SK, false>(s, plan.tws, j, wb, std::make_integer_sequence<int, L - 2>{});
  v2_final_stage<P, SK, false, false>(s, plan.tws, j, wb, v);
  v2_barrier();
  v2_untangle_store<P, SK, LPC_V2_TC_RESID != 0>(s, twW, j, v, Sout + (long)pl * g.cplane + (long)(g.sh + u) * g.cpitch);
}

// KIND (0 vanilla, 1 Nesterov, 2 FISTA) and FIRST (FISTA's first update, where x_k aliases the iterate: gd.py:233,236) are
// template arguments -- the launcher picks the instantiation: with the variant and `first` as run-time flags inside one
// kernel the loads of the auxiliary state sat behind a branch (a `s_waitcnt vmcnt(0)` right behind them) and the three
// variants' registers added up to 250 bytes of scratch per lane.
template <int NT, int SK, class PL, int KIND, int FIRST>
__global__ __launch_bounds__(NT, 4) void k_gd_update_fwd_v2(PlaneGeom g, PL plan, const real2* LPC_RESTRICT twW,
                                                            const real2* LPC_RESTRICT Sin, real2* LPC_RESTRICT Sout,
                                                            real* LPC_RESTRICT X, real* LPC_RESTRICT AUX,
                                                            const real* LPC_RESTRICT alpha, GdScalars pin, FastDiv fc) {
  using P = typename PL::plan;
  constexpr int M = GdV2<P>::M, R = GdV2<P>::R, NB = GdV2<P>::NB, L = GdV2<P>::L, H = R / 2;
  static_assert(GdV2<P>::ok && NT == NB, "k_gd_update_fwd_v2: one butterfly per lane and stage");
  constexpr bool rd = KIND == 1 || (KIND == 2 && !FIRST), wr = KIND != 0;     // gd_aux_access, split == 0
  GdScalars p = pin;
  p.kind = KIND; p.first = FIRST; p.split = 0;
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int j = LPC_TID(NT), u = (int)LPC_BX(g);
  const unsigned pl = LPC_BY(g);
  const int sr = wrap_add(g.sh + u, g.Hp / 2, g.Hp);
  real2 wb[4];
  v2_tw_load<P, 1>(plan.tws, j, wb);
  v2_load_tangle_first<P, SK, LPC_V2_TC_UPDATE != 0>(s, Sin + (long)pl * g.cplane + (long)sr * g.cpitch, twW, j);
  v2_barrier();
  v2_mid_chain<P, SK, true>(s, plan.tws, j, wb, std::make_integer_sequence<int, L - 2>{});
  const real al = alpha[pl - fd_div(pl, fc) * fc.d];
  // the rows of x and of the auxiliary state as range-checked buffers: loads outside the window return zero, stores
  // outside it are dropped (lpc_make_rsrc); padded pair i = j + NB m covers columns 2 i - sw, 2 i - sw + 1
  const unsigned rowb = (unsigned)g.W * (unsigned)sizeof(real);
  const lpc_rsrc xr = lpc_make_rsrc(X + (long)pl * g.uplane + (long)u * g.W, rowb);
  const lpc_rsrc ar = lpc_make_rsrc(AUX + (long)pl * g.uplane + (long)u * g.W, rowb);
  real2 v[R], r[R];
  // Two halves of R / 2 pairs, so that x, the auxiliary state and the butterfly's registers never add up: the first
  // half's rows travel across the last inverse stage, the second half's across the first half's arithmetic.
  // (byte offsets are recomputed where they are used -- one integer operation each -- instead of kept across the butterfly)
  auto offs = [&](int m) { return lpc_opaque((2 * (j + NB * m) - g.sw) * (int)sizeof(real)); };
  auto loads = [&](int m0, real2* xx, real2* aa) {
#pragma unroll
    for (int m = 0; m < H; ++m) xx[m] = !v2_live(4) ? make_real2((real)m, (real)j) : lpc_buf_load2(xr, offs(m0 + m));
#pragma unroll
    for (int m = 0; m < H; ++m) aa[m] = (rd && v2_live(4)) ? lpc_buf_load2(ar, offs(m0 + m)) : make_real2((real)0., (real)0.);
  };
  // gradient pair (j, m + R/2 mod R) belongs to padded pair (j, m): update inside the window, zero outside
  auto update = [&](int m0, const real2* xx, const real2* aa) {
#pragma unroll
    for (int m = 0; m < H; ++m) {
      const real2 gr = v[(m0 + m + H) % R];
      real2 an, xs;
      xs.x = gd_update_val<KIND, 0>(xx[m].x, aa[m].x, gr.x, al, p, an.x);
      xs.y = gd_update_val<KIND, 0>(xx[m].y, aa[m].y, gr.y, al, p, an.y);
      const int off = offs(m0 + m);
      if (wr && v2_live(4)) lpc_buf_store2(ar, off, an);
      if (v2_live(4)) lpc_buf_store2(xr, off, xs);
      r[m0 + m] = (unsigned)off < rowb ? xs : make_real2((real)0., (real)0.);
    }
  };
  real2 x0[H], a0[H], x1[H], a1[H];
  loads(0, x0, a0);
  v2_final_stage<P, SK, true, true>(s, plan.tws, j, wb, v);
  LPC_SCHED_FENCE();
  loads(H, x1, a1);
  LPC_SCHED_FENCE();
  update(0, x0, a0);
  update(H, x1, a1);
  v2_barrier();
  v2_first_fwd<P, SK>(s, j, r);
  v2_mid_chain<P, SK, false>(s, plan.tws, j, wb, std::make_integer_sequence<int, L - 2>{});
  v2_final_stage<P, SK, false, false>(s, plan.tws, j, wb, v);
  v2_barrier();
  v2_untangle_store<P, SK, LPC_V2_TC_UPDATE != 0>(s, twW, j, v, Sout + (long)pl * g.cplane + (long)(g.sh + u) * g.cpitch);
}

#endif   // !LPC_DOUBLE
