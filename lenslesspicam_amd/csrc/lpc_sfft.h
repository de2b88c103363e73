// lpc_sfft.h -- the workgroup FFT of lpc_fft.h with the PLAN AS A TYPE.
//
// lpc_fft.h runs any 5-smooth length from a plan passed by value at launch time: radix per stage from a `switch`,
// index arithmetic through run-time strides and reciprocal multiplications, tail guards on every loop, and a plan
// structure that alone takes ~60 SGPRs.  A kernel instantiated on `SPlanArg<SPlan<radices...>>` instead of `Fft1dPlan`
// gets fft_tile() -- same name, same contract, overloaded on the plan type -- expanded into straight-line stages; the
// instantiations a frame shape needs are compiled per shape into a plan module (lpc_plan.h, lpc_module.cpp):
//   * length, radix, stride, twiddle step, tile width and thread count are constants: butterfly / element indices are
//     shifts and masks, LDS offsets are immediates, no tail guards when the work divides by the workgroup;
//   * the stage loop is unrolled at compile time (no radix switch, so registers are allocated for the radices that
//     are really used);
//   * the kernel argument shrinks to the twiddle-table pointer.
// Kernels stay single-source: they are templates on the plan argument type and call fft_tile(s, plan, ...) either way.
#pragma once
#include "lpc_fft.h"
#include <utility>

template <int... RS>
struct SPlan {
  static constexpr int nst = (int)sizeof...(RS);
  static constexpr int n = (1 * ... * RS);
  static __host__ __device__ constexpr int radix(int st) {
    const int r[] = {RS...};
    return r[st];
  }
  static __host__ __device__ constexpr int ns(int st) {   // product of the radices of the stages before st
    int v = 1;
    for (int i = 0; i < st; ++i) v *= radix(i);
    return v;
  }
  // same rule as plan_from_radices() in lpc_engine.cpp: the i + i/8 skew stays affine in every stage
  static __host__ __device__ constexpr bool skew_ok() {
    for (int st = 0; st < nst; ++st) {
      if ((n / radix(st)) % 8 != 0) return false;
      if (!(ns(st) % 8 == 0 || (ns(st) == 1 && radix(st) % 8 == 0))) return false;
    }
    return true;
  }
  // Stage twiddles in LANE ORDER (round 5; SPlanArg::tws): for every stage st >= 1 and butterfly j < n / radix(st) the base
  // powers w^q, w^2q, w^4q, w^8q (q = (j % ns(st)) * n / (ns(st) radix(st)); radices 8 and 16) that twiddle_mul() gathers
  // from the n-entry table, stored [st][h][j] as pairs {w^q, w^2q} (h = 0) and {w^4q, w^8q} (h = 1): a lane's two 16-byte
  // loads per stage are contiguous across the wave (8 cache lines per instruction) where the gathers tw[q], tw[2q], tw[4q],
  // tw[8q] touch up to 32 lines each -- measured on the row kernels' access pattern, 32 such gathers per lane cost as
  // much as the rows' HBM traffic itself (tools/probe/row_pattern.hip, profiles/r05_notes.md).
  // (other radices: the R - 1 powers w^(q m) themselves, [m - 1][j])
  static __host__ __device__ constexpr int tws_block(int st) {    // real2 entries of stage st's block
    return ((radix(st) == 8 || radix(st) == 16) ? 4 : radix(st) - 1) * (n / radix(st));
  }
  static __host__ __device__ constexpr int tws_off(int st) {      // offset of stage st's block, in real2
    int o = 0;
    for (int i = 1; i < st; ++i) o += tws_block(i);
    return o;
  }
  static constexpr int tws_size = tws_off(nst);
  static bool matches(const Fft1dPlan& p) {             // host: is this the plan build_plan() made?
    if (p.n != n || p.nst != nst) return false;
    for (int st = 0; st < nst; ++st)
      if (p.radix[st] != radix(st)) return false;
    return true;
  }
};

// what a kernel receives instead of an Fft1dPlan: only the twiddle table travels at run time
// LANE: the kernel takes the radix-8 / -16 stage twiddles of its row tiles (BT == 1) from `tws`, the lane-ordered table --
// a property of the TYPE (the row kernels of a plan module), so that a kernel holds one of the two code paths, not both
// (with a run-time choice between them the 512-lane row kernels grew from 64 to 122 VGPRs: two workgroups per CU, +35 %)
template <class P, bool LANE = false>
struct SPlanArg {
  using plan = P;
  static constexpr int n = P::n;
  static constexpr bool lane_tw = LANE;
  const real2* tw;   // exp(-2 pi i q / n), n entries (the table of the Fft1dPlan this replaces)
  const real2* tws;  // LANE: the stage twiddles in lane order (SPlan::tws_off)
};
template <class P, bool LANE = false>
static inline SPlanArg<P, LANE> splan_arg(const Fft1dPlan& p, const real2* tws = nullptr) {
  SPlanArg<P, LANE> a;
  a.tw = p.tw;
  a.tws = tws;
  return a;
}

// The twiddle table of a SHORT transform moves into LDS behind the tile (P::n entries, written by every lane of the
// workgroup): each lane reads the same few powers in every transform of its kernel, and as global loads those were
// most of a column kernel's memory instructions (C4's 540-point middle: 136 of 204 loads; 0.651 -> 0.590 ms).  The
// first twiddle is read after the tile fill's barrier (stage 0 of a Stockham pass has none), so no extra barrier.
template <int NT, class P, bool LANE>
static __device__ __forceinline__ SPlanArg<P, LANE> twiddles_to_lds(SPlanArg<P, LANE> pa, real2* dst, int tid) {
  for (int q = tid; q < P::n; q += NT) dst[q] = pa.tw[q];
  pa.tw = dst;
  return pa;
}
template <int NT>
static __device__ __forceinline__ Fft1dPlan twiddles_to_lds(const Fft1dPlan& p, real2*, int) { return p; }
// The same copy in two halves around `mid()`: the table's loads are ISSUED (no branch: lanes past the table read entry 0),
// mid() issues whatever else the caller wants in flight, then the values go to LDS.  In straight-line code the compiler
// waits with a count, not for everything -- twiddles_to_lds() in front of a kernel's tile loads cost every workgroup one
// memory latency before its first tile load went out (profiles/r05_notes.md section 5).  The caller points pa.tw at dst.
template <int NT, class P, bool LANE, class Mid>
static __device__ __forceinline__ void twiddles_to_lds_around(const SPlanArg<P, LANE>& src, real2* dst, int tid, Mid mid) {
  constexpr int TN = (P::n + NT - 1) / NT;
  real2 v[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int q = tid + t * NT;
    v[t] = src.tw[q < P::n ? q : 0];
  }
  mid();
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int q = tid + t * NT;
    if (q < P::n) dst[q] = v[t];
  }
}

template <class T> struct is_static_plan : std::false_type {};
template <class P, bool LANE> struct is_static_plan<SPlanArg<P, LANE>> : std::true_type {};

// ---- one stage, everything but tid and the pointers known at compile time ------------------------------------
// LANE (SPlanArg::lane_tw): row tiles (BT == 1) take the twiddles of radix-8 / -16 stages from tws, the lane-ordered table
template <class P, int ST, int NT, int BT, bool INV, int SKEW, bool LANE = false>
static __device__ __forceinline__ void sfft_stage(real2* s, const real2* LPC_RESTRICT tw, int tid,
                                                   const real2* LPC_RESTRICT tws = nullptr) {
  constexpr int R = P::radix(ST), N = P::n, NS = P::ns(ST);
  constexpr int NB = N / R, NWORK = NB * BT, MAXB = (NWORK + NT - 1) / NT;
  constexpr bool GUARD = (NWORK % NT) != 0;
  constexpr int IST = NB * BT, OST = NS * BT;
  constexpr int RS = lds_stride<SKEW>(IST), WS = lds_stride<SKEW>(OST);
  constexpr bool RAFF = lds_affine<SKEW>(IST), WAFF = lds_affine<SKEW>(OST);
  constexpr int TWSTEP = N / (NS * R);
  real2 v[MAXB][R];
  int obase[MAXB];          // element index of the butterfly's first output (WAFF: already its slot)
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    const int w = tid + b * NT;
    obase[b] = -1;
    if (!GUARD || w < NWORK) {
      const int j = w / BT, c = w % BT;
      const int jq = j / NS, k = j % NS;
      const int rb = lds_slot<SKEW>(w);
#pragma unroll
      for (int m = 0; m < R; ++m) v[b][m] = RAFF ? s[rb + m * RS] : s[lds_slot<SKEW>(w + m * IST)];
      if constexpr (LANE && NS > 1 && BT == 1) twiddle_mul_lane<R, INV>(v[b], tws + P::tws_off(ST), NB, j);
      else if (NS > 1) twiddle_mul<R, INV>(v[b], tw, k * TWSTEP);
      Dft<R, INV>::run(v[b]);
      const int oi = (jq * NS * R + k) * BT + c;
      obase[b] = (WAFF || SKEW == LPC_LAY_SKEW8) ? lds_slot<SKEW>(oi) : oi;
    }
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    if (!GUARD || obase[b] >= 0) {
      if (SKEW == LPC_LAY_SKEW8 && NS == 1) {
#pragma unroll
        for (int m = 0; m < R; ++m) s[obase[b] + m + (m >> 3)] = v[b][m];
      } else if (WAFF || SKEW == LPC_LAY_SKEW8) {
#pragma unroll
        for (int m = 0; m < R; ++m) s[obase[b] + m * WS] = v[b][m];
      } else {
#pragma unroll
        for (int m = 0; m < R; ++m) s[lds_slot<SKEW>(obase[b] + m * OST)] = v[b][m];
      }
    }
  }
  __syncthreads();
}

template <class P, int NT, int BT, bool INV, int SKEW, int FIRST, bool LANE = false, int... I>
static __device__ __forceinline__ void sfft_stages(real2* s, const real2* LPC_RESTRICT tw, int tid,
                                                    std::integer_sequence<int, I...>, const real2* LPC_RESTRICT tws = nullptr) {
  (sfft_stage<P, FIRST + I, NT, BT, INV, SKEW, LANE>(s, tw, tid, tws), ...);
}

// first stage fused into the tile fill (see fft_first_stage_fused)
template <class P, int NT, int BT, bool INV, int SKEW, bool SRC_LDS, class Src, class Fix>
static __device__ __forceinline__ void sfft_first_fused(real2* s, int tid, Src& src, Fix& fix) {
  constexpr int R = P::radix(0), N = P::n;
  constexpr int NB = N / R, NWORK = NB * BT, MAXB = (NWORK + NT - 1) / NT;
  constexpr bool GUARD = (NWORK % NT) != 0;
  real2 v[MAXB][R];
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    const int w = tid + b * NT;
    if (!GUARD || w < NWORK) {
      const int j = w / BT, c = w % BT;
#pragma unroll
      for (int m = 0; m < R; ++m) v[b][m] = src(j + m * NB, c);
    }
  }
  if (SRC_LDS) __syncthreads();
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    const int w = tid + b * NT;
    if (!GUARD || w < NWORK) {
      const int j = w / BT, c = w % BT;
#pragma unroll
      for (int m = 0; m < R; ++m) v[b][m] = fix(j + m * NB, c, v[b][m]);
      Dft<R, INV>::run(v[b]);
      const int ob = lds_slot<SKEW>(j * R * BT + c);
      if (SKEW == LPC_LAY_SKEW8) {
#pragma unroll
        for (int m = 0; m < R; ++m) s[ob + m + (m >> 3)] = v[b][m];
      } else {
#pragma unroll
        for (int m = 0; m < R; ++m) s[ob + m * BT] = v[b][m];
      }
    }
  }
  __syncthreads();
}

// last stage fused into the drain (see fft_last_stage_fused)
template <class P, int NT, int BT, bool INV, int SKEW, bool LANE = false, class Dst>
static __device__ __forceinline__ void sfft_last_fused(real2* s, const real2* LPC_RESTRICT tw, int tid, Dst& dst,
                                                        const real2* LPC_RESTRICT tws = nullptr) {
  constexpr int ST = P::nst - 1;
  constexpr int R = P::radix(ST), N = P::n, NS = P::ns(ST);
  constexpr int NB = N / R, NWORK = NB * BT, MAXB = (NWORK + NT - 1) / NT;
  constexpr bool GUARD = (NWORK % NT) != 0;
  constexpr int IST = NB * BT, RS = lds_stride<SKEW>(IST);
  constexpr bool RAFF = lds_affine<SKEW>(IST);
  constexpr int TWSTEP = N / (NS * R);
  real2 v[MAXB][R];
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    const int w = tid + b * NT;
    if (!GUARD || w < NWORK) {
      const int rb = lds_slot<SKEW>(w);
#pragma unroll
      for (int m = 0; m < R; ++m) v[b][m] = RAFF ? s[rb + m * RS] : s[lds_slot<SKEW>(w + m * IST)];
    }
  }
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    const int w = tid + b * NT;
    if (!GUARD || w < NWORK) {
      const int j = w / BT, c = w % BT;
      const int jq = j / NS, k = j % NS;
      if constexpr (LANE && NS > 1 && BT == 1) twiddle_mul_lane<R, INV>(v[b], tws + P::tws_off(ST), NB, j);
      else if (NS > 1) twiddle_mul<R, INV>(v[b], tw, k * TWSTEP);
      Dft<R, INV>::run(v[b]);
      const int oi = jq * NS * R + k;
#pragma unroll
      for (int m = 0; m < R; ++m) dst(oi + m * NS, c, v[b][m]);
    }
  }
}

// ---- fft_tile, static-plan overload: same template parameters and call shape as the run-time one ----------------
// EMAX must be n * BT / NT rounded up (the kernels' launch tables guarantee it); BT is passed as a run-time value for
// source compatibility but MUST equal the compile-time SBT the kernel was instantiated for.
// hook(): called once, right behind the issue of the tile loads (loads of constants a later step needs queue up BEHIND
// the data the first stage waits for instead of in front of it)
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};
template <int NT, int EMAX, bool INV, int SKEW, bool SRC_LDS, bool FUSE1 = false, bool FUSEL = false, int SBT = 1,
          class P, bool LANE, class Src, class Dst, class Fix = NoFix, class Hook = NoHook>
static __device__ __forceinline__ void fft_tile(real2* s, const SPlanArg<P, LANE>& pa, int /*BT*/, FastDiv /*btdiv*/, int tid,
                                                 Src src, Dst dst, Fix fix = Fix(), Hook hook = Hook()) {
  constexpr int BT = SBT;
  constexpr int NELEM = P::n * BT;
  constexpr int EM = (NELEM + NT - 1) / NT;
  constexpr bool GUARD = (NELEM % NT) != 0;
  static_assert(EM <= EMAX, "static plan: the tile does not fit the workgroup shape");
  constexpr bool src_lds = std::is_same<Src, LdsNatural>::value, dst_lds = std::is_same<Dst, LdsNatural>::value;
  constexpr bool fuse1 = FUSE1 && !src_lds && P::nst >= 1;
  constexpr bool fusel = FUSEL && !dst_lds && (P::nst - (fuse1 ? 1 : 0)) >= 1;
  const real2* tw = pa.tw;
  if constexpr (fuse1) {
    // (a hook that writes LDS -- the twiddle copy of the fused middles -- would race with the stages below: there is no
    // barrier between it and their first read)
    static_assert(std::is_same<Hook, NoHook>::value, "FUSE1 and a hook do not combine");
    sfft_first_fused<P, NT, BT, INV, SKEW, SRC_LDS>(s, tid, src, fix);
    hook();
  } else if constexpr (!src_lds) {
    real2 v[EM];
#pragma unroll
    for (int k = 0; k < EM; ++k) {
      const int e = tid + k * NT;
      if (!GUARD || e < NELEM) v[k] = src(e / BT, e % BT);
    }
    hook();
    if (SRC_LDS) __syncthreads();
#pragma unroll
    for (int k = 0; k < EM; ++k) {
      const int e = tid + k * NT;
      if (!GUARD || e < NELEM) {
        if constexpr (std::is_same<Fix, NoFix>::value) s[lds_slot<SKEW>(e)] = v[k];
        else s[lds_slot<SKEW>(e)] = fix(e / BT, e % BT, v[k]);
      }
    }
    __syncthreads();
  }
  constexpr int FIRST = fuse1 ? 1 : 0;
  constexpr int NMID = P::nst - FIRST - (fusel ? 1 : 0);
  sfft_stages<P, NT, BT, INV, SKEW, FIRST, LANE>(s, tw, tid, std::make_integer_sequence<int, (NMID > 0 ? NMID : 0)>{}, pa.tws);
  if constexpr (fusel) {
    sfft_last_fused<P, NT, BT, INV, SKEW, LANE>(s, tw, tid, dst, pa.tws);
  } else if constexpr (!dst_lds) {
#pragma unroll
    for (int k = 0; k < EM; ++k) {
      const int e = tid + k * NT;
      if (!GUARD || e < NELEM) dst(e / BT, e % BT, s[lds_slot<SKEW>(e)]);
    }
  }
}
