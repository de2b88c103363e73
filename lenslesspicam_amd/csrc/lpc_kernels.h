// lpc_kernels.h -- device kernels of the MI355X deconvolution engine.
//
// Memory layout in HBM (every element is `real`: float in liblpc.so, double in liblpc_f64.so):
//   * images are PLANAR: plane q = (b*D + d)*C + c, rows contiguous.  The reference keeps
//     channels innermost (stride-3 FFTs); planar turns B, D and C into one batch index.
//   * padded real plane:   [Hp][rpitch]  (rpitch >= Wp)
//   * half spectrum plane: [Hp][cpitch]  real2 (cpitch >= Wc = Wp/2+1, 16-element padded so
//     every tile row is a whole number of 128-byte lines)
//   * un-padded plane:     [H][W]
// Column (H-axis) transforms longer than LDS allows are split four-step style,
// Hp = N1*N2: pass A = length-N1 FFTs over rows {n1*N2 + n2} (+ twiddle), pass B =
// length-N2 FFTs over the contiguous row block {k1*N2 + n2}.  Frequencies therefore stay
// in a PERMUTED row order (row k1*N2+k2 holds frequency k1 + N1*k2); every spectral
// constant (H, |PsiT Psi|, phase tables) is generated in the same order, so no transpose
// or reordering pass ever touches HBM.
#pragma once
#include "lpc_sfft.h"

struct PlaneGeom {
  int H, W;        // un-padded spatial size
  int Hp, Wp, Wc;  // padded rows / cols, half-spectrum cols
  int sh, sw;      // origin of the sensor window inside the padded frame
  int rpitch;      // floats per padded real row
  int cpitch;      // real2 per spectrum row
  long rplane;     // floats per padded real plane
  long cplane;     // real2 per spectrum plane
  long uplane;     // floats per un-padded plane
  int DC;          // D*C: number of PSF planes (state plane q uses PSF plane q % DC)
  int C;           // channels (data plane of state plane q = (q / DC) * C + q % C)
  int rev;         // per launch (the launcher sets it on its copy): the row kernels that honour it hand their workgroups
                   // out from the last (plane, row) to the first (see ColPass::rev)
  int slay;        // layout of the ADMM work spectra (and of the copies of H / |G| the fused middle reads): 0 rows of cpitch
                   // elements; 1 PAIR LINES (spec_col below) -- kernels take it as a template argument, the host and
                   // paired_rows_of read this copy
};
// ---- pair-line layout of a half spectrum (paired rows + 8-column middles: DiffuserCam-sized frames) --------------------
// The single-pass fused middle of a 540-row frame holds 8 image columns per tile: 64-byte row segments, half a cache line
// per access, and a line whose two halves are written by different workgroups is evicted half-dirty in between (the other
// half is fetched and the whole line written back: 1.19 x the algorithmic traffic at C4, profiles/r05_notes.md section 3).
// Here rows (2p, 2p + 1) x columns [8c, 8c + 8) share ONE 128-byte line:
//     element (r, k)  at  (r >> 1) * 2 cpitch + (k >> 3) * 16 + (r & 1) * 8 + (k & 7)
// -- a tile row pair is a full line for the middle (16 consecutive lanes), and the paired row kernels, which hold rows
// 2p and 2p + 1 of one array in one transform, still write / read whole lines (their two stores per bin are the two
// halves of the same line).  The base of a row pair is r0 * cpitch as before (r0 even: paired_rows_of aligns the
// window's pairs), SL = 0 is the plain layout.
template <int SL>
static __host__ __device__ __forceinline__ int spec_col(int k) { return SL ? k + (k & ~7) : k; }

// block coordinates of a kernel that honours PlaneGeom::rev
#define LPC_BX(g) ((g).rev ? gridDim.x - 1u - blockIdx.x : blockIdx.x)
#define LPC_BY(g) ((g).rev ? gridDim.y - 1u - blockIdx.y : blockIdx.y)
#define LPC_BZ(g) ((g).rev ? gridDim.z - 1u - blockIdx.z : blockIdx.z)

static __device__ __forceinline__ int wrap_add(int i, int d, int n) {  // (i + d) mod n for |d| <= n
  int r = i + d;
  if (r >= n) r -= n;
  if (r < 0) r += n;
  return r;
}

// four adjacent reals (16 bytes in float32; the float64 build moves two 16-byte halves)
#ifdef LPC_DOUBLE
struct alignas(16) real4 { double x, y, z, w; };     // two 16-byte halves per lane
#else
typedef float4 real4;
#endif
static __host__ __device__ __forceinline__ real4 make_real4(real x, real y, real z, real w) {
  real4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r;
}
static __device__ __forceinline__ real4 ld4(const real* p) { return *reinterpret_cast<const real4*>(p); }
static __device__ __forceinline__ void st4(real* p, real4 v) { *reinterpret_cast<real4*>(p) = v; }

// ===================================================================== row passes ==
// Two real rows ride through ONE complex FFT of length Wp (re = row A, im = row B) and are
// separated afterwards by Hermitian symmetry; works for even and odd Wp alike.
// The first FFT stage pulls its inputs straight from global memory (source lambda) and the last
// stage pushes its outputs straight out (sink lambda); LDS only carries the tile between stages
// and the Hermitian (un)tangling, which pairs bins k and Wp-k held by different lanes.
#define LPC_ROW_SMEM_BYTES(Wp, skew) ((size_t)lds_slots_of((Wp), (int)(skew)) * sizeof(real2))

// s[] holds Z = FFT(a + i b) in natural order; writes A[k], B[k] for k in [0, Wc)
// (SL = 1: outA = the pair's base, outB = base + 8, see spec_col; validA: row A exists -- the first pair of a window that
// starts on an odd row holds the row above it, which is not stored)
template <int NT, int SK, int SL = 0>
static __device__ __forceinline__ void untangle_store(const real2* s, int Wp, int Wc, real2* outA,
                                                       real2* outB, bool validB, int tid, bool validA = true) {
  for (int k = tid; k < Wc; k += NT) {
    real2 zk = s[lds_slot<SK>(k)];
    real2 zn = s[lds_slot<SK>(k == 0 ? 0 : Wp - k)];
    if (validA) outA[spec_col<SL>(k)] = make_real2((real)0.5 * (zk.x + zn.x), (real)0.5 * (zk.y - zn.y));
    if (validB) outB[spec_col<SL>(k)] = make_real2((real)0.5 * (zk.y + zn.y), -(real)0.5 * (zk.x - zn.x));
  }
}

// builds Z[k] = A[k] + i B[k] over the full length from two half spectra (irfft semantics:
// imaginary parts of the DC and Nyquist bins are ignored).  All loads are issued before the
// first LDS write (unrolled to the compile-time bound) so they overlap in flight.
template <int NT, int EMAX, int SK, int SL = 0>
static __device__ __forceinline__ void tangle_load(real2* s, int Wp, int Wc, const real2* inA,
                                                    const real2* inB, bool validB, int tid, bool validA = true) {
  constexpr int EH = EMAX / 2 + 1;
  if constexpr (SL == 1) {
    // pair lines: bins k, k + 1 (k even) of one row are 16 adjacent, 16-byte-aligned bytes: ONE load per row and lane instead
    // of two (C4's inverse rows 0.276 -> 0.268 ms, same box, two trees; the same for the stores of the forward rows bought
    // nothing).  The load of a row's last pair may reach bin Wc in the padding of the row pitch -- never used.
    constexpr int EP = (EH + 1) / 2;
    real4 a4[EP], b4[EP];
    const real4 z4 = make_real4((real)0., (real)0., (real)0., (real)0.);
#pragma unroll
    for (int q = 0; q < EP; ++q) {
      const int k = 2 * (tid + q * NT);
      a4[q] = b4[q] = z4;
      if (k < Wc) {
        if (validA) a4[q] = ld4((const real*)(inA + spec_col<SL>(k)));
        if (validB) b4[q] = ld4((const real*)(inB + spec_col<SL>(k)));
      }
    }
#pragma unroll
    for (int q = 0; q < EP; ++q) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k = 2 * (tid + q * NT) + u;
        if (k < Wc) {
          real2 av = u ? make_real2(a4[q].z, a4[q].w) : make_real2(a4[q].x, a4[q].y);
          real2 bv = u ? make_real2(b4[q].z, b4[q].w) : make_real2(b4[q].x, b4[q].y);
          const bool selfconj = (k == 0) || (2 * k == Wp);
          if (selfconj) { av.y = (real)0.; bv.y = (real)0.; }
          s[lds_slot<SK>(k)] = make_real2(av.x - bv.y, av.y + bv.x);
          if (!selfconj) s[lds_slot<SK>(Wp - k)] = make_real2(av.x + bv.y, bv.x - av.y);
        }
      }
    }
    return;
  }
  real2 a[EH], b[EH];
#pragma unroll
  for (int q = 0; q < EH; ++q) {
    const int k = tid + q * NT;
    a[q] = make_real2((real)0., (real)0.);
    b[q] = make_real2((real)0., (real)0.);
    if (k < Wc) {
      if (validA) a[q] = inA[spec_col<SL>(k)];
      if (validB) b[q] = inB[spec_col<SL>(k)];
    }
  }
#pragma unroll
  for (int q = 0; q < EH; ++q) {
    const int k = tid + q * NT;
    if (k < Wc) {
      real2 av = a[q], bv = b[q];
      const bool selfconj = (k == 0) || (2 * k == Wp);
      if (selfconj) { av.y = (real)0.; bv.y = (real)0.; }
      s[lds_slot<SK>(k)] = make_real2(av.x - bv.y, av.y + bv.x);
      if (!selfconj) s[lds_slot<SK>(Wp - k)] = make_real2(av.x + bv.y, bv.x - av.y);
    }
  }
}

// Radix-2 stage folded into the Hermitian (un)tangling (row lengths whose plan ends in a radix-2 stage,
// e.g. 8192 = 8*8*8*8*2).  With nb = Wp/2, the lane that owns butterflies j and nb-j of that stage holds
// Z[j], Z[nb+j], Z[nb-j], Z[Wp-j] -- exactly what the (un)tangling of bins j and nb-j needs -- so the stage
// never makes its own trip through LDS.
//
// forward: s[] holds the tile BEFORE the last (radix-2, ns = nb) stage; writes A[k], B[k], k in [0, nb]
template <int NT, int SK>
static __device__ __forceinline__ void untangle_r2_store(const real2* s, int Wp, const real2* LPC_RESTRICT tw,
                                                          real2* outA, real2* outB, bool validB, int tid) {
  const int nb = Wp >> 1;
  auto emit = [&](int k, real2 zk, real2 zn) {     // zk = Z[k], zn = Z[Wp - k]
    outA[k] = make_real2((real)0.5 * (zk.x + zn.x), (real)0.5 * (zk.y - zn.y));
    if (validB) outB[k] = make_real2((real)0.5 * (zk.y + zn.y), (real)-0.5 * (zk.x - zn.x));
  };
  for (int j = tid; j <= nb / 2; j += NT) {
    const int jm = nb - j;                          // mirror butterfly (== nb for j == 0: no such butterfly)
    const real2 u0 = s[lds_slot<SK>(j)];
    const real2 v0 = j ? cmul(s[lds_slot<SK>(j + nb)], tw[j]) : s[lds_slot<SK>(nb)];
    const real2 zj = cadd(u0, v0), zjn = csub(u0, v0);            // Z[j], Z[j + nb]
    if (j == 0) {
      emit(0, zj, zj);                                           // DC pairs with itself
      emit(nb, zjn, zjn);                                        // so does the Nyquist bin
    } else if (j == jm) {
      emit(j, zj, zjn);                                          // Wp - j == j + nb
    } else {
      const real2 u1 = s[lds_slot<SK>(jm)];
      const real2 v1 = cmul(s[lds_slot<SK>(jm + nb)], tw[jm]);
      const real2 zm = cadd(u1, v1), zmn = csub(u1, v1);         // Z[nb - j], Z[Wp - j]
      emit(j, zj, zmn);
      emit(jm, zm, zjn);                                         // Wp - (nb - j) == nb + j
    }
  }
}

// inverse: builds Z from the half spectra and applies the FIRST (radix-2, ns = 1, twiddle-free) stage of the
// inverse plan: writes y[2j] = Z[j] + Z[j+nb], y[2j+1] = Z[j] - Z[j+nb] into the (un-skewed) LDS tile.
// Every spectrum element is loaded exactly once.
template <int NT>
static __device__ __forceinline__ void tangle_r2_load(real2* s, int Wp, const real2* inA, const real2* inB,
                                                       bool validB, int tid) {
  const int nb = Wp >> 1;
  const real2 zero = make_real2((real)0, (real)0);
  for (int j = tid; j <= nb / 2; j += NT) {
    const int jm = nb - j;
    real2 a0 = inA[j], b0 = validB ? inB[j] : zero;
    real2 a1 = inA[jm], b1 = validB ? inB[jm] : zero;               // j == 0: the Nyquist bin
    if (j == 0) { a0.y = (real)0; b0.y = (real)0; a1.y = (real)0; b1.y = (real)0; }
    // Z[k] = A[k] + i B[k];  Z[Wp - k] = conj(A[k]) + i conj(B[k])
    const real2 zj = make_real2(a0.x - b0.y, a0.y + b0.x);          // Z[j]
    const real2 zm = make_real2(a1.x - b1.y, a1.y + b1.x);          // Z[nb - j]   (j == 0: Z[nb])
    const real2 zjn = make_real2(a1.x + b1.y, b1.x - a1.y);         // Z[j + nb]  = mirror of bin nb - j
    const real2 zmn = make_real2(a0.x + b0.y, b0.x - a0.y);         // Z[Wp - j]  = mirror of bin j
    if (j == 0) {
      s[0] = cadd(zj, zm);
      s[1] = csub(zj, zm);
    } else {
      s[2 * j] = cadd(zj, zjn);
      s[2 * j + 1] = csub(zj, zjn);
      if (j != jm) {
        s[2 * jm] = cadd(zm, zmn);
        s[2 * jm + 1] = csub(zm, zmn);
      }
    }
  }
}

// ---- ADMM, paired rows: which two rows a workgroup transforms ------------------------------------------------------
// Two real rows share one complex transform of length Wp -- always two rows of the SAME array (rows r, r + 1 of r_sp; of
// `a`; of V; of H V).  Round 3 paired row r of one array with row r of the other (r_sp with a, V with H V): the smaller
// signal then inherits eps x |larger signal| of rounding noise through the Hermitian separation, which the reference
// does not have (it transforms every array on its own) -- 2e-5 instead of 4e-7 after six iterations with norm="forward"
// or a PSF scaled to l2 = 1e-3, where a and H V sit 3-6 decades below r_sp and V (tests/test_norm_scale.py).  Two
// adjacent rows of one array are of one magnitude, like the rows of the reference's own 2-D transform.
// Array 0 always has all its rows transformed: pairs (2j, 2j + 1), j < nA = ceil(Hp / 2).  Array 1 either likewise
// (nB = nA) or on the rows of the sensor window alone: pairs (sh + 2j, sh + 2j + 1), j < nB = ceil(H / 2)
// (AdmmScalars::skipa / skiphv).  grid.x = nA + nB; the first 2 nB blocks alternate between the arrays (block b runs on
// XCD b % 8, flipped every eighth block so that every XCD gets both kinds), the rest are array 0.
// Pair-line spectra (PlaneGeom::slay) need r0 even: the window's pairs then start at sh & ~1, and a pair that straddles the
// window's edge has one row that is neither formed nor stored (first / second).
struct PairedRows { int arr, r0; bool first, second; };
static inline int paired_rows_count(int rows) { return (rows + 1) >> 1; }
static __host__ __device__ __forceinline__ int paired_rows_window(const PlaneGeom& g) {
  return g.slay ? ((g.sh + g.H + 1) >> 1) - (g.sh >> 1) : (g.H + 1) >> 1;
}
static inline int paired_rows_grid(const PlaneGeom& g, bool window_b) {
  return paired_rows_count(g.Hp) + (window_b ? paired_rows_window(g) : paired_rows_count(g.Hp));
}
static __device__ __forceinline__ PairedRows paired_rows_of(const PlaneGeom& g, unsigned bx, bool window_b) {
  const int nB = window_b ? paired_rows_window(g) : (g.Hp + 1) >> 1;
  PairedRows r;
  int idx;
  if ((int)bx < 2 * nB) { r.arr = (int)((bx ^ (bx >> 3)) & 1u); idx = (int)(bx >> 1); }
  else { r.arr = 0; idx = (int)bx - nB; }
  r.first = true;
  if (r.arr == 1 && window_b) {
    r.r0 = (g.slay ? g.sh & ~1 : g.sh) + 2 * idx;
    r.first = r.r0 >= g.sh;
    r.second = r.r0 + 1 < g.sh + g.H;
  }
  else { r.r0 = 2 * idx; r.second = r.r0 + 1 < g.Hp; }
  return r;
}

// ---- forward, ADMM: rows (r, r + 1) of array A -> SA, of array B -> SB ------------
template <int NT, int EMAX, int SK, bool R2, class PL = Fft1dPlan, int SL = 0>
__global__ __launch_bounds__(NT) void k_rfwd_arrays(PlaneGeom g, PL plan,
                                                     const real* LPC_RESTRICT A,
                                                     const real* LPC_RESTRICT B,
                                                     real2* LPC_RESTRICT SA,
                                                     real2* LPC_RESTRICT SB) {
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int tid = LPC_TID(NT);
  const long pl = blockIdx.y;
  const PairedRows pr = paired_rows_of(g, blockIdx.x, false);
  const real* a = (pr.arr ? B : A) + pl * g.rplane + (long)pr.r0 * g.rpitch;
  const real* b = a + g.rpitch;
  const bool v1 = pr.second;
  auto src = [&](int i, int) { return make_real2(a[i], v1 ? b[i] : (real)0.); };
  if constexpr (is_static_plan<PL>::value)     // compile-time plans: no radix-2 folding (R2 == false)
    fft_tile<NT, EMAX, false, SK, false, true>(s, plan, 1, make_fastdiv_dev1(), tid, src, LdsNatural{});
  else
    fft_tile<NT, EMAX, false, SK, false, true>(s, plan, 1, make_fastdiv_dev1(), tid, src, LdsNatural{}, NoFix{}, 0,
                                               R2 ? 1 : 0);
  real2* oa = (pr.arr ? SB : SA) + pl * g.cplane + (long)pr.r0 * g.cpitch;
  real2* ob = oa + (SL ? 8 : g.cpitch);
  static_assert(!(R2 && SL), "pair-line spectra: compile-time plans only");
  if (R2) untangle_r2_store<NT, SK>(s, g.Wp, plan.tw, oa, ob, v1, tid);
  else untangle_store<NT, SK, SL>(s, g.Wp, g.Wc, oa, ob, v1, tid);
}

// ---- ADMM rows, one real row per HALF-length complex transform ---------------------------------------
// Even / odd packing: z[j] = x[2j] + i x[2j+1], Z = FFT_M(z), M = Wp/2, then
//   X[k] = E + w^k O,  X[M-k] = conj(E - w^k O),   E = (Z[k] + conj Z[M-k])/2,  O = -i (Z[k] - conj Z[M-k])/2,
// w = exp(-2 pi i / Wp).  Same arithmetic per row as pairing two arrays in one length-Wp transform, but the tile
// is half as large (4 workgroups of 256 threads per CU instead of 2 of 512 at Wp = 8192): measured on MI355X the
// row passes are bound by compute that two workgroups per CU cannot hide (profiles/r01b_notes.md).
// blockIdx.x = 2*row + array.  `plan` has length M, `twW` is the length-Wp table.  Needs Wp even.
// s[] holds Z = FFT_M(z) in natural order; writes X[0 .. M] (M = Wp/2) to o
template <int NT, int SK>
static __device__ __forceinline__ void untangle_half_store(const real2* s, int M, const real2* LPC_RESTRICT twW,
                                                            real2* LPC_RESTRICT o, int tid) {
  for (int k = tid; k <= M / 2; k += NT) {
    const int km = M - k;
    const real2 zk = s[lds_slot<SK>(k)];
    const real2 zm = s[lds_slot<SK>(k == 0 ? 0 : km)];
    const real ex = (real)0.5 * (zk.x + zm.x), ey = (real)0.5 * (zk.y - zm.y);
    const real2 od = make_real2((real)0.5 * (zk.y + zm.y), (real)-0.5 * (zk.x - zm.x));   // O
    const real2 wo = cmul(twW[k], od);
    o[k] = make_real2(ex + wo.x, ey + wo.y);
    if (k != km) o[km] = make_real2(ex - wo.x, wo.y - ey);
  }
}

template <int NT, int EMAX, int SK, class PL = Fft1dPlan>
__global__ __launch_bounds__(NT) void k_rfwd_half(PlaneGeom g, PL plan, const real2* LPC_RESTRICT twW,
                                                   const real* LPC_RESTRICT A, const real* LPC_RESTRICT B,
                                                   real2* LPC_RESTRICT SA, real2* LPC_RESTRICT SB) {
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int tid = LPC_TID(NT), row = blockIdx.x >> 1, arr = blockIdx.x & 1;
  const long pl = blockIdx.y;
  const real2* a2 = (const real2*)((arr ? B : A) + pl * g.rplane + (long)row * g.rpitch);
  auto src = [&](int i, int) { return a2[i]; };
  fft_tile<NT, EMAX, false, SK, false, true>(s, plan, 1, make_fastdiv_dev1(), tid, src, LdsNatural{});
  untangle_half_store<NT, SK>(s, g.Wp >> 1, twW, (arr ? SB : SA) + pl * g.cplane + (long)row * g.cpitch, tid);
}

// inverse: Z[k] = E' + i O',  Z[M-k] = conj(E') + i conj(O'),  E' = X[k] + conj X[M-k],
// O' = (X[k] - conj X[M-k]) conj(w^k); the unnormalised inverse FFT_M of Z is (x[2j], x[2j+1]).
// irfft semantics: the imaginary parts of the DC and Nyquist bins are ignored.
// builds Z (natural order, length M) in LDS from one half-spectrum row; ends WITHOUT a barrier
template <int NT, int EMAX, int SK>
static __device__ __forceinline__ void tangle_half_load(real2* s, int M, const real2* LPC_RESTRICT twW,
                                                         const real2* LPC_RESTRICT in, int tid) {
  constexpr int EH = EMAX / 2 + 1;
  real2 xk[EH], xm[EH];
#pragma unroll
  for (int q = 0; q < EH; ++q) {          // every spectrum element is loaded once; all loads before the LDS writes
    const int k = tid + q * NT;
    xk[q] = make_real2((real)0., (real)0.);
    xm[q] = xk[q];
    if (k <= M / 2) { xk[q] = in[k]; xm[q] = in[M - k]; }
  }
#pragma unroll
  for (int q = 0; q < EH; ++q) {
    const int k = tid + q * NT;
    if (k <= M / 2) {
      real2 a = xk[q], b = xm[q];
      if (k == 0) { a.y = (real)0.; b.y = (real)0.; }
      const real2 e = make_real2(a.x + b.x, a.y - b.y);            // E'
      const real2 d = make_real2(a.x - b.x, a.y + b.y);            // X[k] - conj X[M-k]
      const real2 od = cmul_conj(d, twW[k]);                       // O'
      s[lds_slot<SK>(k)] = make_real2(e.x - od.y, e.y + od.x);
      if (k != 0 && k != M - k) s[lds_slot<SK>(M - k)] = make_real2(e.x + od.y, od.x - e.y);
    }
  }
}

template <int NT, int EMAX, int SK, class PL = Fft1dPlan>
__global__ __launch_bounds__(NT) void k_rinv_half(PlaneGeom g, PL plan, const real2* LPC_RESTRICT twW,
                                                   const real2* LPC_RESTRICT SA, const real2* LPC_RESTRICT SB,
                                                   real* LPC_RESTRICT A, real* LPC_RESTRICT B, int skip_b_outside) {
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  // block b runs on XCD b % 8: the array flips every fourth row so that each XCD transforms rows of both (the rows of
  // B outside the sensor window may be skipped: with `arr = b & 1` only the odd XCDs would have less to do: 0.49 -> 0.47 ms, r02ak)
  const int tid = LPC_TID(NT);
  const unsigned bx = LPC_BX(g);
  int row = bx >> 1, arr = (bx ^ (bx >> 3)) & 1;
  const long pl = LPC_BY(g);
  // AdmmScalars::skiphv: grid.x = 2 H + (Hp - H) -- both arrays on the rows of the sensor window, then A (= V) alone on
  // the rows above and below it (no empty workgroups: each would still claim its LDS and a launch slot)
  if (skip_b_outside) {
    if ((int)bx < 2 * g.H) row += g.sh;
    else {
      const int q = (int)bx - 2 * g.H;
      row = q < g.sh ? q : q + g.H;
      arr = 0;
    }
  }
  tangle_half_load<NT, EMAX, SK>(s, g.Wp >> 1, twW, (arr ? SB : SA) + pl * g.cplane + (long)row * g.cpitch, tid);
  __syncthreads();
  real2* o2 = (real2*)((arr ? B : A) + pl * g.rplane + (long)row * g.rpitch);
  auto out = [&](int i, int, real2 v) { o2[i] = v; };
  fft_tile<NT, EMAX, true, SK, true, false, true>(s, plan, 1, make_fastdiv_dev1(), tid, LdsNatural{}, out);
}

// ---- forward, generic: rows (2b, 2b+1) of ONE real source -> spectrum rows ------------
struct RealSrc {
  const real* base;
  long plane_stride;  // floats
  int pitch;          // floats per row
  int nrows;          // rows >= nrows are implicit zeros
  int ncols, col0;    // column c of the padded frame maps to base[...][c-col0] if in [col0,col0+ncols)
  int out_row0;       // source row r lands in spectrum row out_row0 + r
};

template <int NT, int EMAX, int SK, bool R2>
__global__ __launch_bounds__(NT) void k_rfwd_rows(PlaneGeom g, Fft1dPlan plan, RealSrc src,
                                                   real2* LPC_RESTRICT S) {
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int tid = LPC_TID(NT);
  const int r0 = 2 * blockIdx.x, r1 = r0 + 1;
  const long pl = blockIdx.y;
  const bool v1 = r1 < src.nrows;
  const real* a = src.base + pl * src.plane_stride + (long)r0 * src.pitch;
  const real* b = src.base + pl * src.plane_stride + (long)r1 * src.pitch;
  auto in = [&](int i, int) {   // pad on load
    const int c = i - src.col0;
    const bool ok = (c >= 0) && (c < src.ncols);
    return make_real2(ok ? a[c] : (real)0., (ok && v1) ? b[c] : (real)0.);
  };
  fft_tile<NT, EMAX, false, SK, false, true>(s, plan, 1, make_fastdiv_dev1(), tid, in, LdsNatural{}, NoFix{}, 0,
                                             R2 ? 1 : 0);
  real2* o = S + pl * g.cplane + (long)(src.out_row0 + r0) * g.cpitch;
  if (R2) untangle_r2_store<NT, SK>(s, g.Wp, plan.tw, o, o + g.cpitch, v1, tid);
  else untangle_store<NT, SK>(s, g.Wp, g.Wc, o, o + g.cpitch, v1, tid);
}

// one real row per half-length transform (see k_rfwd_half), generic source with pad on load
template <int NT, int EMAX, int SK, class PL = Fft1dPlan>
__global__ __launch_bounds__(NT) void k_rfwd_rows_half(PlaneGeom g, PL plan, const real2* LPC_RESTRICT twW,
                                                        RealSrc src, real2* LPC_RESTRICT S) {
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int tid = LPC_TID(NT), r = blockIdx.x;
  const long pl = blockIdx.y;
  const real* a = src.base + pl * src.plane_stride + (long)r * src.pitch;
  auto in = [&](int i, int) {   // z[i] = (x[2i], x[2i+1]), x zero outside [col0, col0 + ncols)
    const int c = 2 * i - src.col0;
    return make_real2((c >= 0 && c < src.ncols) ? a[c] : (real)0.,
                      (c + 1 >= 0 && c + 1 < src.ncols) ? a[c + 1] : (real)0.);
  };
  fft_tile<NT, EMAX, false, SK, false, true>(s, plan, 1, make_fastdiv_dev1(), tid, in, LdsNatural{});
  untangle_half_store<NT, SK>(s, g.Wp >> 1, twW, S + pl * g.cplane + (long)(src.out_row0 + r) * g.cpitch, tid);
}

// ---- inverse, ADMM: spectra SA, SB -> real arrays A, B (no shift, padded), two rows of one array per transform -------
// R2: `plan` is the inverse-row plan whose FIRST stage is the radix-2 one (fused into the tangling); SK is
// false in that case (the skew is not affine for ns = 2).
// window_only (AdmmScalars::skiphv): B (= H V) is produced on the rows of the sensor window alone (paired_rows_of).
template <int NT, int EMAX, int SK, bool R2, class PL = Fft1dPlan, int SL = 0>
__global__ __launch_bounds__(NT) void k_rinv_arrays(PlaneGeom g, PL plan,
                                                     const real2* LPC_RESTRICT SA,
                                                     const real2* LPC_RESTRICT SB,
                                                     real* LPC_RESTRICT A, real* LPC_RESTRICT B, int window_only) {
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int tid = LPC_TID(NT);
  LPC_STAMP_BEGIN(3);
  const long pl = LPC_BY(g);
  const PairedRows pr = paired_rows_of(g, LPC_BX(g), window_only != 0);
  const bool va = pr.first, vb = pr.second;
  const real2* ia = (pr.arr ? SB : SA) + pl * g.cplane + (long)pr.r0 * g.cpitch;
  const real2* ib = ia + (SL ? 8 : g.cpitch);
  static_assert(!(R2 && SL), "pair-line spectra: compile-time plans only");
  if (R2) tangle_r2_load<NT>(s, g.Wp, ia, ib, vb, tid);
  else tangle_load<NT, EMAX, SK, SL>(s, g.Wp, g.Wc, ia, ib, vb, tid, va);
  __syncthreads();
  real* a = (pr.arr ? B : A) + pl * g.rplane + (long)pr.r0 * g.rpitch;
  real* b = a + g.rpitch;
  auto out = [&](int i, int, real2 v) { if (va) a[i] = v.x; if (vb) b[i] = v.y; };
  if constexpr (is_static_plan<PL>::value)
    fft_tile<NT, EMAX, true, SK, true, false, true>(s, plan, 1, make_fastdiv_dev1(), tid, LdsNatural{}, out);
  else
    fft_tile<NT, EMAX, true, SK, true, false, true>(s, plan, 1, make_fastdiv_dev1(), tid, LdsNatural{}, out, NoFix{},
                                                    R2 ? 1 : 0, 0);
  LPC_STAMP_END();
}

// ---- inverse, generic: spectrum rows -> ONE real sink with ifftshift (+ crop) -----------
// Output row i of the shifted frame comes from spectrum row (i + Hp/2) mod Hp and output
// column c from FFT sample (c + Wp/2) mod Wp (fft.ifftshift is a roll by -(n//2)).  That is pure
// index arithmetic in the sink of the last FFT stage: exact, and no extra pass over HBM.
struct RealDst {
  real* base;
  long plane_stride;
  int pitch;
  int nrows;       // number of output rows (Hp if not cropping, H if cropping)
  int row0, col0;  // output (r, c) = shifted-frame (row0 + r, col0 + c); (0,0) when not cropping
  int ncols;       // Wp or W
};

// FFT sample i of a row lands in output column (i - Wp/2 - col0) mod Wp (if < ncols)
static __device__ __forceinline__ int shifted_col(int i, int hw, int col0, int Wp) {
  int c = i - hw - col0;
  if (c < 0) c += Wp;
  if (c < 0) c += Wp;
  return c;
}

template <int NT, int EMAX, int SK, bool R2>
__global__ __launch_bounds__(NT) void k_rinv_rows(PlaneGeom g, Fft1dPlan plan,
                                                   const real2* LPC_RESTRICT S, RealDst dst) {
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int tid = LPC_TID(NT);
  const int r0 = 2 * blockIdx.x, r1 = r0 + 1;
  const long pl = blockIdx.y;
  const bool v1 = r1 < dst.nrows;
  const int hh = g.Hp / 2, hw = g.Wp / 2;
  const int sr0 = wrap_add(dst.row0 + r0, hh, g.Hp);
  const int sr1 = wrap_add(dst.row0 + (v1 ? r1 : r0), hh, g.Hp);
  if (R2) tangle_r2_load<NT>(s, g.Wp, S + pl * g.cplane + (long)sr0 * g.cpitch,
                             S + pl * g.cplane + (long)sr1 * g.cpitch, v1, tid);
  else tangle_load<NT, EMAX, SK>(s, g.Wp, g.Wc, S + pl * g.cplane + (long)sr0 * g.cpitch,
                                 S + pl * g.cplane + (long)sr1 * g.cpitch, v1, tid);
  __syncthreads();
  real* a = dst.base + pl * dst.plane_stride + (long)r0 * dst.pitch;
  real* b = dst.base + pl * dst.plane_stride + (long)r1 * dst.pitch;
  auto out = [&](int i, int, real2 v) {
    const int c = shifted_col(i, hw, dst.col0, g.Wp);
    if (c < dst.ncols) {
      a[c] = v.x;
      if (v1) b[c] = v.y;
    }
  };
  fft_tile<NT, EMAX, true, SK, true, false, true>(s, plan, 1, make_fastdiv_dev1(), tid, LdsNatural{}, out, NoFix{},
                                                  R2 ? 1 : 0, 0);
}

// one real row per half-length transform, generic sink with ifftshift (+ crop)
template <int NT, int EMAX, int SK, class PL = Fft1dPlan>
__global__ __launch_bounds__(NT) void k_rinv_rows_half(PlaneGeom g, PL plan, const real2* LPC_RESTRICT twW,
                                                        const real2* LPC_RESTRICT S, RealDst dst) {
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int tid = LPC_TID(NT), r = blockIdx.x;
  const long pl = blockIdx.y;
  const int hh = g.Hp / 2, hw = g.Wp / 2;
  const int sr = wrap_add(dst.row0 + r, hh, g.Hp);
  tangle_half_load<NT, EMAX, SK>(s, g.Wp >> 1, twW, S + pl * g.cplane + (long)sr * g.cpitch, tid);
  __syncthreads();
  real* a = dst.base + pl * dst.plane_stride + (long)r * dst.pitch;
  auto out = [&](int i, int, real2 v) {     // samples 2i and 2i+1 of the row
    const int c0 = shifted_col(2 * i, hw, dst.col0, g.Wp);
    if (c0 < dst.ncols) a[c0] = v.x;
    const int c1 = shifted_col(2 * i + 1, hw, dst.col0, g.Wp);
    if (c1 < dst.ncols) a[c1] = v.y;
  };
  fft_tile<NT, EMAX, true, SK, true, false, true>(s, plan, 1, make_fastdiv_dev1(), tid, LdsNatural{}, out);
}

#ifndef LPC_MID_FUSE1
#define LPC_MID_FUSE1 false  // measured: no gain on the fused middle (profiles/r01b_notes.md)
#endif
#ifndef LPC_MID_CONSTS_LATE
#define LPC_MID_CONSTS_LATE 1
#endif
#ifndef LPC_COLS_FUSEL
#define LPC_COLS_FUSEL true
#endif
// ================================================================== column passes ==
struct ColPass {
  int N;            // transform length of this pass
  int G;            // groups along H (Hp / N)
  int istride;      // rows between consecutive transform elements
  int gstride;      // rows between consecutive groups:  row(g,i) = g*gstride + i*istride
  int T;            // image columns per tile
  int ntile_c;      // ceil(Wc / T)
  int tw_mode;      // 0: none; 1: multiply result k by twH[g*k] (forward pass A);
                    // 2: multiply input k by conj(twH[g*k]) (inverse pass A)
  int zr0, zr1;     // forward only: rows outside [zr0,zr1) are implicit zeros on load
  int need0, needn; // inverse only: rows r with ((r - need0) mod Hp) >= needn are never read again
                    // (they fall outside the crop window after ifftshift) and are not stored
  const real2* twH;  // exp(-2 pi i q / Hp), q in [0, Hp)
  // forward pass A of the ADMM work spectra: planes >= sc_plane0 (the spectrum of `a`) are multiplied by `sc` on load
  // in rows outside [sc_r0, sc_r1) -- rows whose forward row transform was skipped because it is mu1 * Wp times the
  // row spectrum the last inverse row pass consumed (AdmmScalars::skipa).  sc_plane0 = INT_MAX: off.
  int sc_plane0, sc_r0, sc_r1;
  real sc;
  FastDiv tdiv;     // fast divide by T
  FastDiv tcdiv;    // fast divide by ntile_c
  // ADMM LDS middle: hand the workgroups out so that two adjacent column tiles run on the SAME XCD (block b runs on XCD
  // b % 8): within every 16 consecutive blocks, block 8 s + x takes tile 2 x + s.  With 8-column tiles (64-byte row
  // segments: single-pass columns of one DiffuserCam-sized frame) the two tiles share every cache line, and on one L2 the
  // second one's loads are hits.
  int swz;
  // ADMM middles: |PsiT Psi| = ga[spectrum row] + gb[column] (null: read it from the plane).  The reference's gram
  // separates; read as a plane its 64-byte tile rows are fetched as whole lines once per colour plane -- 0.5 GB of the
  // 12-MP middle's 3.65 GB (profiles/r03_notes.md section 17)
  const real* ga;
  const real* gb;
  // walk the grid backwards (planes and blocks): a pass that starts where the previous kernel finished finds the last
  // ~256 MB that one wrote still in the memory-side cache (MI355X: 256 MB Infinity Cache in front of HBM)
  int rev;
};

// element at a 32-bit byte offset from a workgroup-uniform base (SGPR base + VGPR offset addressing)
template <class T>
static __device__ __forceinline__ T ld_off(const T* LPC_RESTRICT ubase, unsigned byte_off) {
  return *(const T*)((const char*)ubase + byte_off);
}
template <class T>
static __device__ __forceinline__ void st_off(T* LPC_RESTRICT ubase, unsigned byte_off, T v) {
  *(T*)((char*)ubase + byte_off) = v;
}

// plain pass over ONE spectrum array, in place (global -> registers -> [LDS] -> registers -> global)
// PL / SBT: run-time plan (SBT unused), or a compile-time plan with SBT == cp.T columns per tile (lpc_sfft.h)
template <int NT, int EMAX, bool INV, class PL = Fft1dPlan, int SBT = 0, bool TWLDS = false>
__global__ __launch_bounds__(NT) void k_cols(PlaneGeom g, PL plan, ColPass cp,
                                              real2* LPC_RESTRICT S) {
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int tid = LPC_TID(NT);
  const unsigned bx = cp.rev ? gridDim.x - 1u - blockIdx.x : blockIdx.x;     // ColPass::rev
  const unsigned by = cp.rev ? gridDim.y - 1u - blockIdx.y : blockIdx.y;
  const int grp = (int)fd_div(bx, cp.tcdiv);
  const int c0 = ((int)bx - grp * cp.ntile_c) * cp.T;
  real2* base = S + (long)by * g.cplane + (long)grp * cp.gstride * g.cpitch + c0;
  const long rstep = (long)cp.istride * g.cpitch;
  // compile-time plans address the tile with 32-bit byte offsets from the workgroup-uniform base, row index x row step
  // as a 24-bit product (choose_plan() admits them only when the step is below 2^24 bytes): one full-rate v_mad_u32_u24
  // per access instead of a quarter-rate 64-bit multiply-add and a 64-bit shift-add
  constexpr bool O32 = is_static_plan<PL>::value;
  constexpr unsigned c8 = (unsigned)sizeof(real2);
  const unsigned r8 = (unsigned)rstep * c8;
  const int row0 = grp * cp.gstride;
  auto in = [&](int i, int c) {
    real2 x = make_real2((real)0., (real)0.);
    const int row = row0 + i * cp.istride;
    if (c0 + c < g.Wc && (INV || (row >= cp.zr0 && row < cp.zr1)))
      x = O32 ? ld_off(base, mul24((unsigned)i, r8) + (unsigned)c * c8) : base[i * rstep + c];
    // (a select + an unconditional product instead of this branch measured slower: pass A 0.47 -> 0.50 ms, r02am)
    if (!INV && (int)by >= cp.sc_plane0 && (row < cp.sc_r0 || row >= cp.sc_r1)) x = cscale(x, cp.sc);
    return x;
  };
  // compile-time plans (short transforms): the plan's twiddles and this group's four-step twiddles
  // twH[grp * i] live in LDS behind the tile (the host adds 2 n entries to the launch's LDS size)
  constexpr bool TWL = is_static_plan<PL>::value && TWLDS;
  const real2* tws = nullptr;
  if constexpr (TWL) {
    real2* t0 = s + PL::n * SBT;
    plan = twiddles_to_lds<NT>(plan, t0, tid);
    if ((!INV && cp.tw_mode == 1) || (INV && cp.tw_mode == 2)) {
      for (int i = tid; i < PL::n; i += NT) t0[PL::n + i] = cp.twH[grp * i];
      tws = t0 + PL::n;
    }
  }
  auto untwiddle = [&](int i, int, real2 x) {   // inverse pass A: conj four-step twiddle on the way in
    return (INV && cp.tw_mode == 2) ? cmul_conj(x, TWL ? tws[i] : cp.twH[grp * i]) : x;
  };
  auto out = [&](int i, int c, real2 x) {
    if (c0 + c < g.Wc) {
      if (!INV && cp.tw_mode == 1) x = cmul(x, TWL ? tws[i] : cp.twH[grp * i]);
      if (INV && cp.needn < g.Hp) {
        int d = row0 + i * cp.istride - cp.need0;
        d = d < 0 ? d + g.Hp : d;
        if (d >= cp.needn) return;
      }
      if (O32) st_off(base, mul24((unsigned)i, r8) + (unsigned)c * c8, x);
      else base[i * rstep + c] = x;
    }
  };
  if constexpr (is_static_plan<PL>::value) {
    // (SRC_LDS = TWL: the barrier between the tile's loads and `untwiddle` makes the staged twiddles visible)
    if (INV) fft_tile<NT, EMAX, INV, false, TWL, true, LPC_COLS_FUSEL, SBT>(s, plan, cp.T, cp.tdiv, tid, in, out, untwiddle);
    else fft_tile<NT, EMAX, INV, false, false, false, LPC_COLS_FUSEL, SBT>(s, plan, cp.T, cp.tdiv, tid, in, out);
  } else {
    if (INV) fft_tile<NT, EMAX, INV, false, false, true, LPC_COLS_FUSEL>(s, plan, cp.T, cp.tdiv, tid, in, out, untwiddle);
    else fft_tile<NT, EMAX, INV, false, false, false, LPC_COLS_FUSEL>(s, plan, cp.T, cp.tdiv, tid, in, out);
  }
}

// fused middle of a convolution: forward pass B -> multiply by the PSF spectrum (or its
// conjugate) -> inverse pass B, one trip through HBM.  hscale folds 1/(Hp*Wp).
template <int NT, int EMAX>
__global__ __launch_bounds__(NT) void k_cols_mid_mul(PlaneGeom g, Fft1dPlan plan, ColPass cp,
                                                      real2* LPC_RESTRICT S,
                                                      const real2* LPC_RESTRICT Hs, int conjH,
                                                      real hscale, int psf_planes) {
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int tid = LPC_TID(NT);
  const int T = cp.T;
  const int grp = (int)fd_div(blockIdx.x, cp.tcdiv);
  const int c0 = ((int)blockIdx.x - grp * cp.ntile_c) * T;
  const long rowoff = ((long)grp * cp.gstride) * g.cpitch + c0;
  real2* base = S + (long)blockIdx.y * g.cplane + rowoff;
  const real2* hb = Hs + (long)((int)blockIdx.y % psf_planes) * g.cplane + rowoff;
  const int nelem = cp.N * T;
  const long rstep = (long)cp.istride * g.cpitch;
  const int row0 = grp * cp.gstride;
  real2 h[EMAX];
#pragma unroll
  for (int k = 0; k < EMAX; ++k) {   // PSF spectrum tile: in flight during the forward transform
    const int e = tid + k * NT;
    h[k] = make_real2((real)0., (real)0.);
    if (e < nelem) {
      const int i = (int)fd_div((unsigned)e, cp.tdiv);
      const int j = e - i * T;
      if (c0 + j < g.Wc) h[k] = hb[i * rstep + j];
    }
  }
  auto in = [&](int i, int c) {
    const int row = row0 + i * cp.istride;
    return (c0 + c < g.Wc && row >= cp.zr0 && row < cp.zr1) ? base[i * rstep + c] : make_real2((real)0., (real)0.);
  };
  fft_tile<NT, EMAX, false, false, false, LPC_MID_FUSE1>(s, plan, T, cp.tdiv, tid, in, LdsNatural{});
#pragma unroll
  for (int k = 0; k < EMAX; ++k) {
    const int e = tid + k * NT;
    if (e < nelem) {
      real2 x = s[e];
      x = conjH ? cmul_conj(x, h[k]) : cmul(x, h[k]);
      s[e] = cscale(x, hscale);
    }
  }
  __syncthreads();
  auto out = [&](int i, int c, real2 x) {
    if (c0 + c < g.Wc) base[i * rstep + c] = x;
  };
  fft_tile<NT, EMAX, true, false, true, false, LPC_COLS_FUSEL>(s, plan, T, cp.tdiv, tid, LdsNatural{}, out);
}

// ---- register-resident middle (split column passes, short pass-B transforms) ---------------------------
// When the fused middle transform is short (N = R1*R2 <= 64, e.g. 48 = 8*6 at 6144 rows) ONE LANE can hold a whole
// column transform: N complex values in VGPRs, two-factor Cooley-Tukey with compile-time indices -- no LDS, no
// barriers, no index arithmetic, twiddles at constant table offsets (scalar loads).  A wave covers 64 adjacent
// columns, so every row access is one 512-byte segment.  Measured on MI355X (profiles/r01b_notes.md) the LDS
// version of this kernel spent a third of its time on un-hidden butterflies.
//
// forward: x[j1*R2 + j2] = element n = j1*R2 + j2 (natural)  ->  x[k1*R2 + k2] = X[k1 + R1*k2]
template <int R1, int R2>
static __device__ __forceinline__ void reg_fft_fwd(real2* x, const real2* LPC_RESTRICT tw) {
  constexpr int N = R1 * R2;
#pragma unroll
  for (int j2 = 0; j2 < R2; ++j2) {
    real2 v[R1];
#pragma unroll
    for (int j1 = 0; j1 < R1; ++j1) v[j1] = x[j1 * R2 + j2];
    Dft<R1, false>::run(v);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) x[k1 * R2 + j2] = (k1 * j2) ? cmul(v[k1], tw[(k1 * j2) % N]) : v[k1];
  }
#pragma unroll
  for (int k1 = 0; k1 < R1; ++k1) Dft<R2, false>::run(x + k1 * R2);
}
// inverse (unnormalised): x[k1*R2 + k2] = X[k1 + R1*k2] (what reg_fft_fwd leaves)  ->  x[s] = element s (natural)
template <int R1, int R2>
static __device__ __forceinline__ void reg_fft_inv(real2* x, const real2* LPC_RESTRICT tw) {
  constexpr int N = R1 * R2;
#pragma unroll
  for (int k1 = 0; k1 < R1; ++k1) {           // length-R2 transforms over k2, then the twiddle w^(m1*k1)
    Dft<R2, true>::run(x + k1 * R2);
#pragma unroll
    for (int m1 = 0; m1 < R2; ++m1)
      if (m1 * k1) x[k1 * R2 + m1] = cmul_conj(x[k1 * R2 + m1], tw[(m1 * k1) % N]);
  }
#pragma unroll
  for (int m1 = 0; m1 < R2; ++m1) {           // length-R1 transforms over k1: output m1 + R2*m2 at slot m2*R2 + m1
    real2 v[R1];
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) v[k1] = x[k1 * R2 + m1];
    Dft<R1, true>::run(v);
#pragma unroll
    for (int m2 = 0; m2 < R1; ++m2) x[m2 * R2 + m1] = v[m2];
  }
}

// keeps the instruction scheduler from hoisting every load of a later phase to the top (which costs more
// registers than the lane has: measured 996 bytes of scratch per lane without the fences)
#if defined(LPC_SIMT_EMU)
#define LPC_SCHED_FENCE() ((void)0)
#else
#define LPC_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
// 1 / d for the ADMM middles' R_divmat (admm.py:186-190; d = mu1 |H|^2 + mu2 |G| + mu3 > 0, far from the overflow and
// denormal ranges): the hardware reciprocal (1 ulp) + one Newton step = 3 VALU operations, against ~10 for the compiler's
// IEEE division (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup) -- 170 of the sequential middle's 3100.
static __device__ __forceinline__ real recip_pos(real d) {
#if defined(LPC_SIMT_EMU) || defined(LPC_DOUBLE)
  return (real)1.0 / d;
#else
  const float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
#endif
}
// fused middle of a convolution, register-resident (same contract as k_cols_mid_mul; split passes only:
// cp.istride == 1, every row valid).  grid = (ceil(Wc/64), groups, planes), 64 threads.
template <int R1, int R2>
__global__ __launch_bounds__(64) void k_cols_mid_mul_reg(PlaneGeom g, Fft1dPlan plan, ColPass cp,
                                                          real2* LPC_RESTRICT S, const real2* LPC_RESTRICT Hs,
                                                          int conjH, real hscale, int psf_planes) {
  constexpr int N = R1 * R2;
  const int col = (int)LPC_BX(g) * 64 + (int)threadIdx.x;
  if (col >= g.Wc) return;
  // plain 64-bit addresses on purpose: with wave-uniform bases + 32-bit offsets this kernel needs 58 instead of
  // 214 AGPRs but runs 65 % slower (0.62 vs 0.38 ms at 12 MP) -- the hoisted address arithmetic is what lets
  // all 2N loads of a lane be in flight at once
  const long rowoff = (long)LPC_BY(g) * cp.gstride * g.cpitch + col;
  real2* base = S + (long)LPC_BZ(g) * g.cplane + rowoff;
  const real2* hb = Hs + (long)((int)LPC_BZ(g) % psf_planes) * g.cplane + rowoff;
  real2 x[N], h[N];
#pragma unroll
  for (int n = 0; n < N; ++n) x[n] = base[(long)n * g.cpitch];
#pragma unroll
  for (int n = 0; n < N; ++n) h[n] = hb[(long)n * g.cpitch];
  reg_fft_fwd<R1, R2>(x, plan.tw);
  const real hs = conjH ? -hscale : hscale;
#pragma unroll
  for (int k1 = 0; k1 < R1; ++k1)
#pragma unroll
    for (int k2 = 0; k2 < R2; ++k2) {
      const real2 hh = make_real2(h[k1 + R1 * k2].x * hscale, h[k1 + R1 * k2].y * hs);   // H or conj(H), scaled
      x[k1 * R2 + k2] = cmul(x[k1 * R2 + k2], hh);
    }
  reg_fft_inv<R1, R2>(x, plan.tw);
#pragma unroll
  for (int n = 0; n < N; ++n) base[(long)n * g.cpitch] = x[n];
}

// fused middle of one ADMM iteration, register-resident (same contract as k_cols_mid_admm below; split passes
// only, SHORT pass-B transforms: the lane holds 2 x N complex values).  One lane owns column c of BOTH spectra,
// one after the other: a = FFT(SB column) is turned into t = s conj(H) a in place, then r = FFT(SA column),
// Vh = Rdiv (r + t) and HVh = s H Vh overwrite the two register arrays, which are inverse-transformed and stored.
// The row phases are wave-uniform (scalar loads), H is simply read twice.
// grid = (ceil(Wc/64), groups, planes), 64 threads.
template <int R1, int R2>
__global__ __launch_bounds__(64) void k_cols_mid_admm_reg(PlaneGeom g, Fft1dPlan plan, ColPass cp,
                                                           real2* LPC_RESTRICT SA, real2* LPC_RESTRICT SB,
                                                           const real2* LPC_RESTRICT Hs,
                                                           const real* LPC_RESTRICT Gabs,
                                                           const real2* LPC_RESTRICT phr,
                                                           const real2* LPC_RESTRICT phc, real mu1, real mu2,
                                                           real mu3, real rscale) {
  constexpr int N = R1 * R2;
  const int col = (int)blockIdx.x * 64 + (int)threadIdx.x;
  if (col >= g.Wc) return;
  const int row0 = (int)blockIdx.y * cp.gstride;
  // wave-uniform bases (SGPR pairs) + 32-bit per-lane byte offsets: global_load ... v_off, s[base]
  const long urow = (long)row0 * g.cpitch;
  real2* ba = SA + (long)blockIdx.z * g.cplane + urow;
  real2* bb = SB + (long)blockIdx.z * g.cplane + urow;
  const real2* hb = Hs + (long)((int)blockIdx.z % g.DC) * g.cplane + urow;
  const real* gb = Gabs + urow;
  const bool gsep = cp.ga != nullptr;
  const real* gar = gsep ? cp.ga + row0 : Gabs;          // wave-uniform row terms (scalar loads, like the row phases)
  const real gcol = gsep ? cp.gb[col] : (real)0.;
  const real2* pr = phr + row0;
  const real2 pc = phc[col];
  const unsigned c8 = (unsigned)col * (unsigned)sizeof(real2), c4 = (unsigned)col * (unsigned)sizeof(real);
  const unsigned p8 = (unsigned)g.cpitch * (unsigned)sizeof(real2), p4 = (unsigned)g.cpitch * (unsigned)sizeof(real);
  real2 a[N], r[N];
#pragma unroll
  for (int n = 0; n < N; ++n) a[n] = ld_off(bb, c8 + (unsigned)n * p8);
#pragma unroll
  for (int n = 0; n < N; ++n) r[n] = ld_off(ba, c8 + (unsigned)n * p8);
  LPC_SCHED_FENCE();
  reg_fft_fwd<R1, R2>(a, plan.tw);
  LPC_SCHED_FENCE();
#pragma unroll
  for (int k1 = 0; k1 < R1; ++k1) {
    LPC_SCHED_FENCE();
#pragma unroll
    for (int k2 = 0; k2 < R2; ++k2) {          // slot k1*R2 + k2 holds frequency k = k1 + R1*k2
      const int k = k1 + R1 * k2;
      a[k1 * R2 + k2] = cmul(cmul_conj(a[k1 * R2 + k2], ld_off(hb, c8 + (unsigned)k * p8)), cmul(pr[k], pc));
    }
  }
  LPC_SCHED_FENCE();
  reg_fft_fwd<R1, R2>(r, plan.tw);
  LPC_SCHED_FENCE();
#pragma unroll
  for (int k1 = 0; k1 < R1; ++k1) {
    LPC_SCHED_FENCE();
#pragma unroll
    for (int k2 = 0; k2 < R2; ++k2) {
      const int k = k1 + R1 * k2, sl = k1 * R2 + k2;
      const real2 hh = ld_off(hb, c8 + (unsigned)k * p8);
      const real gk = gsep ? gar[k] + gcol : ld_off(gb, c4 + (unsigned)k * p4);
      const real rdiv = rscale * recip_pos(mu1 * rabs(hh.x * hh.x + hh.y * hh.y) + mu2 * gk + mu3);
      const real2 vh = cscale(cadd(r[sl], a[sl]), rdiv);
      r[sl] = vh;
      a[sl] = cmul(cmul(vh, hh), cmul(pr[k], pc));
    }
  }
  LPC_SCHED_FENCE();
  reg_fft_inv<R1, R2>(r, plan.tw);
#pragma unroll
  for (int n = 0; n < N; ++n) st_off(ba, c8 + (unsigned)n * p8, r[n]);
  LPC_SCHED_FENCE();
  reg_fft_inv<R1, R2>(a, plan.tw);
#pragma unroll
  for (int n = 0; n < N; ++n) st_off(bb, c8 + (unsigned)n * p8, a[n]);
}

// fused middle of one ADMM iteration (4-FFT form).  In: SA = rows+colsA transform of
// r_sp, SB = same of a = mu1 X - xi.  After forward pass B:
//   Vh  = Rdiv * (Rh + s * conj(H) * Ah)        (Rdiv formed in-kernel, includes 1/(Hp*Wp))
//   HVh = s * H * Vh                             (s = spectral phase of ifftshift)
// then inverse pass B; SA <- Vh path, SB <- HVh path.
// Tile: [N][2T] -- columns 0..T-1 belong to SA, T..2T-1 to SB.
// PL / SBT2: run-time plan, or a compile-time plan with SBT2 == 2 * cp.T tile columns (both arrays)
// SL = 1 (single-pass columns, T == 8, compile-time plans): the work spectra and the copies of H / |G| passed in are in
// the pair-line layout (spec_col): the tile's rows (2p, 2p + 1) of either array are one 128-byte line.
template <int NT, int EMAX, class PL = Fft1dPlan, int SBT2 = 0, bool TWLDS = false, int SL = 0>
__global__ __launch_bounds__(NT) void k_cols_mid_admm(PlaneGeom g, PL plan, ColPass cp,
                                                       real2* LPC_RESTRICT SA,
                                                       real2* LPC_RESTRICT SB,
                                                       const real2* LPC_RESTRICT Hs,
                                                       const real* LPC_RESTRICT Gabs,
                                                       const real2* LPC_RESTRICT phr,
                                                       const real2* LPC_RESTRICT phc,
                                                       FastDiv t2div, real mu1, real mu2, real mu3,
                                                       real rscale, real sb_outside_scale) {
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int tid = LPC_TID(NT);
  LPC_STAMP_BEGIN(2);
  // (compile-time plans: the tile width is a constant -- e / T, e % T are shifts, not reciprocal multiplies)
  const int T = is_static_plan<PL>::value ? SBT2 / 2 : cp.T, T2 = 2 * T;
  unsigned bid = cp.rev ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
  const unsigned by = cp.rev ? gridDim.y - 1u - blockIdx.y : blockIdx.y;
  if (cp.swz && bid < (gridDim.x & ~15u)) bid = (bid & ~15u) + ((bid & 7u) << 1) + ((bid >> 3) & 1u);   // ColPass::swz
  const int grp = (int)fd_div(bid, cp.tcdiv);
  const int c0 = ((int)bid - grp * cp.ntile_c) * T;
  static_assert(!SL || (is_static_plan<PL>::value && SBT2 == 16), "pair lines: 8 columns per array, compile-time plans");
  const long rowoff = ((long)grp * cp.gstride) * g.cpitch + (SL ? 2 * c0 : c0);      // (SL: single pass, grp == 0)
  real2* ba = SA + (long)by * g.cplane + rowoff;
  real2* bb = SB + (long)by * g.cplane + rowoff;
  const int pp = (int)by % g.DC;
  const real2* hb = Hs + (long)pp * g.cplane + rowoff;
  const real* rb = Gabs + rowoff;  // |PsiT Psi| spectrum: one plane, the same for every channel
  const int npair = cp.N * T;
  const long rstep = (long)cp.istride * g.cpitch;
  constexpr int EP = (EMAX + 1) / 2;
  constexpr bool O32 = is_static_plan<PL>::value;      // 32-bit byte offsets, see k_cols
  constexpr unsigned c8 = (unsigned)sizeof(real2), c4 = (unsigned)sizeof(real);
  const unsigned r8 = (unsigned)rstep * c8, r4 = (unsigned)rstep * c4;
  // byte offsets of (row i, column j of the array's 8): see k_cols_mid_admm_seq
  auto off8 = [=](int i, int j) {
    const unsigned e8 = ((unsigned)i << 3) | (unsigned)j;
    return SL ? mul24(e8 >> 4, 2u * r8) + ((e8 & 15u) << 3) : mul24((unsigned)i, r8) + (unsigned)j * c8;
  };
  auto off4 = [=](int i, int j) {
    const unsigned e8 = ((unsigned)i << 3) | (unsigned)j;
    return SL ? mul24(e8 >> 4, 2u * r4) + ((e8 & 15u) << 2) : mul24((unsigned)i, r4) + (unsigned)j * c4;
  };
  real2 h[EP];
  real rd[EP];
  // spectral constants: in flight during the forward FFT.  Compile-time plans issue them BEHIND the tile loads (hook of
  // fft_tile): the memory counter is in order, and in front of them they delayed the first stage by their own latency --
  // one small frame is a chain of latencies (profiles/r05_notes.md section 5)
  // Branch-free: loads behind `if (e < npair)` / `if (c0 + j < Wc)` each got a `s_waitcnt vmcnt(0)` of their own (five
  // latencies one after the other per lane at C1).  Lanes without an element load element 0, columns past the frame's edge
  // the frame's last column; neither value is used.
  const int jmax = g.Wc - 1 - c0;
  auto consts_of = [&](auto terms_c) {
    constexpr bool TERMS = decltype(terms_c)::value;
#pragma unroll
    for (int k = 0; k < EP; ++k) {
      const int e = tid + k * NT, ec = e < npair ? e : 0;
      const int i = (is_static_plan<PL>::value ? ec / T : (int)fd_div((unsigned)ec, cp.tdiv));
      const int j = ec - i * T, jc = j < jmax ? j : jmax;
      if (O32) h[k] = ld_off(hb, off8(i, jc));
      else h[k] = hb[i * rstep + jc];
      if (TERMS) rd[k] = cp.ga[grp * cp.gstride + i * cp.istride] + cp.gb[c0 + jc];      // rd: |G| for now
      else if (O32) rd[k] = ld_off(rb, off4(i, jc));
      else rd[k] = rb[i * rstep + jc];
    }
  };
  auto consts = [&]() {
    if (cp.ga) consts_of(std::true_type{});
    else consts_of(std::false_type{});
  };
  if constexpr (!(is_static_plan<PL>::value && LPC_MID_CONSTS_LATE)) consts();
  // sb_outside_scale != 0 (AdmmScalars::skipa, single-pass columns only): the rows of SB outside the sensor window were
  // not re-transformed; they hold rfft(H V row) / Wp from the last inverse row pass, and a = mu1 H V there
  const real sb_k = sb_outside_scale != (real)0. ? sb_outside_scale : (real)1.;
  // The tile loads carry neither a branch nor arithmetic (a column past the frame's edge is a copy of the frame's last
  // column: an independent transform that is never stored); the scale of SB's rows is applied where the value goes into
  // LDS (`fix` of fft_tile) -- with the product behind the load the one guarded element of a lane waited for ALL loads.
  auto in = [&](int i, int c) {
    const int j = c < T ? c : c - T, jc = j < jmax ? j : jmax;
    return O32 ? ld_off(c < T ? ba : bb, off8(i, jc)) : (c < T ? ba : bb)[i * rstep + jc];
  };
  auto in_fix = [&](int i, int c, real2 x) {
    return cscale(x, (c >= T && (unsigned)(i - g.sh) >= (unsigned)g.H) ? sb_k : (real)1.);
  };
  if constexpr (is_static_plan<PL>::value) {
    if constexpr (LPC_MID_CONSTS_LATE) {
      // behind the tile loads: the twiddle table's loads (-> LDS, lpc_sfft.h), then the constants
      const PL gplan = plan;
      if constexpr (TWLDS) plan.tw = s + PL::n * SBT2;
      real2* twl = s + PL::n * SBT2;
      auto hook = [&]() {
        if constexpr (TWLDS) twiddles_to_lds_around<NT>(gplan, twl, tid, consts);
        else consts();
      };
      fft_tile<NT, EMAX, false, false, false, LPC_MID_FUSE1, false, SBT2>(s, plan, T2, t2div, tid, in, LdsNatural{}, in_fix, hook);
    } else {
      if constexpr (TWLDS) plan = twiddles_to_lds<NT>(plan, s + PL::n * SBT2, tid);
      fft_tile<NT, EMAX, false, false, false, LPC_MID_FUSE1, false, SBT2>(s, plan, T2, t2div, tid, in, LdsNatural{}, in_fix);
    }
  } else
    fft_tile<NT, EMAX, false, false, false, LPC_MID_FUSE1>(s, plan, T2, t2div, tid, in, LdsNatural{}, in_fix);
  // (branch-free like the constants, only the two LDS stores are conditional; all phase factors are requested before the
  // first is used: loaded where they were used, every element waited for its own pair)
  real2 pr[EP], pc[EP];
#pragma unroll
  for (int k = 0; k < EP; ++k) {
    const int e = tid + k * NT, ec = e < npair ? e : 0;
    const int i = (is_static_plan<PL>::value ? ec / T : (int)fd_div((unsigned)ec, cp.tdiv));
    const int j = ec - i * T, jc = j < jmax ? j : jmax;
    pr[k] = phr[grp * cp.gstride + i * cp.istride];
    pc[k] = phc[c0 + jc];
  }
#pragma unroll
  for (int k = 0; k < EP; ++k) {
    const int e = tid + k * NT, ec = e < npair ? e : 0;
    const int i = (is_static_plan<PL>::value ? ec / T : (int)fd_div((unsigned)ec, cp.tdiv));
    const int j = ec - i * T, jc = j < jmax ? j : jmax;
    const real2 hh = h[k];
    // R_divmat = 1 / (mu1 |H* H| + mu2 |PsiT Psi| + mu3)  (admm.py:186-190), formed on the fly so
    // that per-iteration step sizes cost nothing; rscale folds the inverse FFT's 1/(Hp*Wp)
    const real rdiv = rscale * recip_pos(mu1 * rabs(hh.x * hh.x + hh.y * hh.y) + mu2 * rd[k] + mu3);
    const real2 ph = cmul(pr[k], pc[k]);
    const real2 rh = s[i * T2 + jc];
    const real2 ah = s[i * T2 + T + jc];
    real2 t = cmul(cmul_conj(ah, hh), ph);          // s * conj(H) * Ah
    real2 vh = cscale(cadd(rh, t), rdiv);
    real2 hv = cmul(cmul(vh, hh), ph);
    if (e < npair && j <= jmax) {
      s[i * T2 + j] = vh;
      s[i * T2 + T + j] = hv;
    }
  }
  __syncthreads();
  auto out = [&](int i, int c, real2 x) {
    const int j = c < T ? c : c - T;
    if (c0 + j < g.Wc) {
      if (O32) st_off(c < T ? ba : bb, off8(i, j), x);
      else (c < T ? ba : bb)[i * rstep + j] = x;
    }
  };
  if constexpr (is_static_plan<PL>::value)
    fft_tile<NT, EMAX, true, false, true, false, LPC_COLS_FUSEL, SBT2>(s, plan, T2, t2div, tid, LdsNatural{}, out);
  else
    fft_tile<NT, EMAX, true, false, true, false, LPC_COLS_FUSEL>(s, plan, T2, t2div, tid, LdsNatural{}, out);
  LPC_STAMP_END();
}

// ---- the point-wise step's constants of the pair-line sequential middle, precombined ------------------------------------
// Step 3 of k_cols_mid_admm_seq needs, per tile element, H, |G|, the row and the column phase: four loads and a
// reciprocal per element, nine elements per lane, each waiting for its own loads -- a third of a workgroup's lifetime at C4
// (profiles/r05_notes.md section 5).  They depend on the PSF plane and the step sizes only, not on the frame: one small
// kernel forms  c1 = conj(H) ph,  c2 = H ph  (ph = phr[row] phc[column])  and  rd = rscale / (mu1 |H|^2 + mu2 |G| + mu3)
// once per (PSF, step sizes) -- every iteration of an unrolled schedule, never again otherwise -- in the pair-line
// layout; the step then is one 16-byte and one 4-byte load and three products per element.
struct alignas(16) MidConst { real2 c1, c2; };     // (k_mid_consts: behind the kernel that reads them)

// ---- the same fused middle, ONE ARRAY AT A TIME through a tile of T image columns (single-pass columns only) ------
// k_cols_mid_admm keeps both spectra in one [N][2T] tile; when a whole column is long (540 rows for the
// DiffuserCam-sized frames of C1 / C4) that tile allows only T = 8 columns per array -- 64-byte row segments, half a
// cache line per access.  Here the transform of `a` is parked in registers (N*T/NT values per lane) while the same
// LDS tile transforms `r_sp`: either T = 16 columns per workgroup in the same 69 KiB (every row access a whole 128-byte
// line, half as many workgroups), or -- the default since round 3 -- T = 8 columns in 39 KiB on 256 lanes: four
// workgroups per CU, each barrier over 4 waves instead of 8, and the frames-fastest block order below puts the two
// tiles that share a cache line on the same XCD 1.5 MB apart (C4: 0.608 -> 0.550 ms per launch, r03_notes.md section 15).
// H and |G| are shared by all frames of a batch (L2-resident) and are loaded where they are used.  Compile-time plans
// only (SBT == cp.T).
// SL = 1: the work spectra AND the copies of H / |G| passed in are in the pair-line layout (spec_col; T == 8): a tile's
// rows (2p, 2p + 1) are one 128-byte line.
// PC (with SL): Hs / Gabs are the precombined constants of k_mid_consts (1: MidConst planes, 2: c1 planes -- real phases --
// and the rd planes), phr / phc are not read.
template <int NT, int EMAX, class PL, int SBT, int MINW = 1, bool PRE = false, int SL = 0, int PC = 0>
__global__ __launch_bounds__(NT, MINW) void k_cols_mid_admm_seq(PlaneGeom g, PL plan, ColPass cp, real2* LPC_RESTRICT SA,
                                                           real2* LPC_RESTRICT SB, const real2* LPC_RESTRICT Hs,
                                                           const real* LPC_RESTRICT Gabs, const real2* LPC_RESTRICT phr,
                                                           const real2* LPC_RESTRICT phc, real mu1, real mu2, real mu3,
                                                           real rscale, real sb_outside_scale) {
  static_assert(!PC || SL, "precombined constants live in the pair-line layout");
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  // (plain threadIdx.x, not LPC_TID: knowing tid < NT the optimiser keeps all 17 row indices of a lane -- as 64-bit
  // offsets for ga[] / phr[] -- alive from the tile loads to the point-wise step and spills them: 144 bytes of scratch,
  // C4's middle 0.484 -> 0.797 ms, profiles/r04g_ab.log)
  const int tid = threadIdx.x;
  constexpr int T = SBT, N = PL::n, NELEM = N * T, EM = (NELEM + NT - 1) / NT;
  static_assert(EM <= EMAX, "tile does not fit the workgroup shape");
  LPC_STAMP_BEGIN(2);
  // 1-D grid of (column tiles x planes) workgroups, FRAMES FASTEST: all frames of a batch share H and |G|, and block b
  // runs on XCD b % 8 -- consecutive blocks are the same column tile of the same PSF plane in different frames, so each
  // XCD's L2 fetches that tile of H / |G| from HBM once and serves it to the frames it owns.  (With the tile index
  // fastest, the 64 frames of C4 re-read H 64 times: PMC traffic 2.29 GB per launch against 1.60 GB algorithmic.)
  const int nfr = (int)(gridDim.x / (unsigned)(cp.ntile_c * g.DC));       // frames
  const unsigned bx = cp.rev ? gridDim.x - 1u - blockIdx.x : blockIdx.x;   // ColPass::rev
  const int fr = (int)(bx % (unsigned)nfr), rest = (int)(bx / (unsigned)nfr);
  const int tile = rest % cp.ntile_c, pp = rest / cp.ntile_c;             // column tile, PSF plane
  const long pl = (long)fr * g.DC + pp;
  const int c0 = tile * T;
  static_assert(!SL || T == 8, "pair lines hold 8 columns of two rows");
  const int cb = SL ? 2 * c0 : c0;                  // first element of the tile's column chunk within a row (pair)
  real2* ba = SA + pl * g.cplane + cb;
  real2* bb = SB + pl * g.cplane + cb;
  const real2* hb = Hs + (PC == 1 ? 2 : 1) * ((long)pp * g.cplane + cb);       // (PC == 1: 16-byte elements)
  const real* rb = Gabs + (PC ? (long)pp * g.cplane : 0) + cb;
  // 32-bit byte offsets from workgroup-uniform bases (a plane is < 4 GB): one v_mad_u32 per access instead of a
  // 64-bit multiply-add (quarter rate) + 64-bit shift-add
  const unsigned r8 = (unsigned)g.cpitch * (unsigned)sizeof(real2), r4 = (unsigned)g.cpitch * (unsigned)sizeof(real);
  constexpr unsigned c8 = (unsigned)sizeof(real2), c4 = (unsigned)sizeof(real);
  // byte offset of tile element (row i, column j) of an array of 8- / 4-byte elements.  Pair lines: with e = 8 i + j (the
  // element's index in the [N][8] tile) the pair is e >> 4 (2 cpitch elements each) and the place in its line e & 15 --
  // the same two operations as the plain layout's i * pitch + j
  auto off8 = [=](int i, int j) {
    const unsigned e8 = ((unsigned)i << 3) | (unsigned)j;
    return SL ? mul24(e8 >> 4, 2u * r8) + ((e8 & 15u) << 3) : mul24((unsigned)i, r8) + (unsigned)j * c8;
  };
  auto off4 = [=](int i, int j) {
    const unsigned e8 = ((unsigned)i << 3) | (unsigned)j;
    return SL ? mul24(e8 >> 4, 2u * r4) + ((e8 & 15u) << 2) : mul24((unsigned)i, r4) + (unsigned)j * c4;
  };
  // A lane's elements in tile order, e = tid + k NT, keep their column j0 and advance RSTEP rows per round: every tile-order
  // access is (uniform base + k * rstep elements) + ONE lane-constant byte offset -- no address arithmetic, no address
  // registers per access (under the 64-VGPR bound of four workgroups per CU a per-access v_mad spilled: 44-88 bytes of
  // scratch with the pair-line offsets, the 8-frame shard's middle 0.068 -> 0.089 ms)
  static_assert(NT % T == 0 && (!SL || (NT / T) % 2 == 0), "a lane keeps its column (and its row parity) from round to round");
  constexpr int RSTEP = NT / T, KFULL = NELEM / NT;   // (KFULL: rounds in which every lane has an element)
  const int i0 = tid / T, j0 = tid % T;
  const unsigned l8 = off8(i0, j0), l4 = off4(i0, j0);
  const long rstep = (long)RSTEP * g.cpitch;
  // sb_outside_scale != 0 (AdmmScalars::skipa): the rows of SB outside the sensor window were not re-transformed; they
  // hold rfft(H V row) / Wp from the last inverse row pass, and a = mu1 H V there
  const real sb_k = sb_outside_scale != (real)0. ? sb_outside_scale : (real)1.;
  const int wc = g.Wc - c0, sh = g.sh, hwin = g.H;
  // Loads are UNCONDITIONAL and carry no arithmetic: the columns of a tile are independent transforms, so a column past
  // the frame's edge may hold whatever the row padding holds (cpitch is a multiple of 16 >= every tile width: the
  // address is always inside the plane) -- it is never stored; the tail lanes of the last round load the round's first
  // element.  The scale of SB's rows is applied where the value goes into LDS.
  static_assert(16 % T == 0, "tile columns must stay inside the padded row pitch");
  auto tile_ld = [=](const real2* base, int k) {
    return ld_off(base + k * rstep, (k < KFULL || tid + k * NT < NELEM) ? l8 : 0u);
  };
  auto fixB = [=](int i, real2 x) {
    return cscale(x, (unsigned)(i - sh) >= (unsigned)hwin ? sb_k : (real)1.);   // one compare, one select, one product
  };
  // the twiddle table moves into LDS behind the tile (lpc_sfft.h twiddles_to_lds)
  plan = twiddles_to_lds<NT>(plan, s + NELEM, tid);
  real2 a[EM];
  {
    // PRE: both tiles' loads are issued up front, `a` first, r_sp right behind it: the transform of `a` then runs while
    // the tile of r_sp is still in flight (the wait for `a` is a vmcnt(EM), not a vmcnt(0)), instead of every workgroup
    // sitting out two full load latencies.  The registers that later park Ah hold the r_sp tile until then.  Otherwise
    // (launches of few workgroups per CU) r_sp's loads go out behind the transform of `a`.
    real2 pb[EM], pa[EM];
#pragma unroll
    for (int k = 0; k < EM; ++k) pb[k] = tile_ld(bb, k);
    if constexpr (PRE) {
#pragma unroll
      for (int k = 0; k < EM; ++k) pa[k] = tile_ld(ba, k);
    }
#pragma unroll
    for (int k = 0; k < EM; ++k)
      if (k < KFULL || tid + k * NT < NELEM) s[tid + k * NT] = fixB(i0 + k * RSTEP, pb[k]);
    __syncthreads();
    // 1. Ah = FFT(a), parked in registers in tile order
    fft_tile<NT, EMAX, false, false, false, false, false, SBT>(s, plan, T, cp.tdiv, tid, LdsNatural{}, LdsNatural{});
#pragma unroll
    for (int k = 0; k < EM; ++k) {
      a[k] = make_real2((real)0., (real)0.);
      if (k < KFULL || tid + k * NT < NELEM) a[k] = s[tid + k * NT];
    }
    if constexpr (!PRE) {
#pragma unroll
      for (int k = 0; k < EM; ++k) pa[k] = tile_ld(ba, k);
    }
    __syncthreads();
    // 2. Rh = FFT(r_sp) in the same tile
#pragma unroll
    for (int k = 0; k < EM; ++k)
      if (k < KFULL || tid + k * NT < NELEM) s[tid + k * NT] = pa[k];
    __syncthreads();
    fft_tile<NT, EMAX, false, false, false, false, false, SBT>(s, plan, T, cp.tdiv, tid, LdsNatural{}, LdsNatural{});
  }
  // 3. Vh = Rdiv (Rh + s conj(H) Ah) -> tile;  HVh = s H Vh -> the registers that held Ah
#pragma unroll
  for (int k = 0; k < EM; ++k) {
    if (k < KFULL || tid + k * NT < NELEM) {
      if (j0 < wc) {
        const int e = tid + k * NT, i = i0 + k * RSTEP;
        if constexpr (PC == 1) {
          const MidConst mc = ld_off((const MidConst*)hb + k * rstep, 2u * l8);
          const real rdiv = ld_off(rb + k * rstep, l4);
          const real2 vh = cscale(cadd(s[e], cmul(a[k], mc.c1)), rdiv);
          s[e] = vh;
          a[k] = cmul(vh, mc.c2);
        } else if constexpr (PC == 2) {
          const real2 c1 = ld_off(hb + k * rstep, l8);
          const real rdiv = ld_off(rb + k * rstep, l4);
          const real2 vh = cscale(cadd(s[e], cmul(a[k], c1)), rdiv);
          s[e] = vh;
          a[k] = cmul_conj(vh, c1);
        } else {
          const real2 hh = ld_off(hb + k * rstep, l8);
          const real gk = cp.ga ? cp.ga[i] + cp.gb[c0 + j0] : ld_off(rb + k * rstep, l4);
          const real rdiv = rscale * recip_pos(mu1 * rabs(hh.x * hh.x + hh.y * hh.y) + mu2 * gk + mu3);
          const real2 ph = cmul(phr[i], phc[c0 + j0]);
          const real2 t = cmul(cmul_conj(a[k], hh), ph);
          const real2 vh = cscale(cadd(s[e], t), rdiv);
          s[e] = vh;
          a[k] = cmul(cmul(vh, hh), ph);
        }
      }
    }
  }
  __syncthreads();
  // 4. V-hat back through the inverse transform, straight to SA
  auto outA = [=](int i, int j, real2 x) { if (j < wc) st_off(ba, off8(i, j), x); };
  fft_tile<NT, EMAX, true, false, true, false, LPC_COLS_FUSEL, SBT>(s, plan, T, cp.tdiv, tid, LdsNatural{}, outA);
  __syncthreads();
  // 5. H V-hat: registers -> tile -> inverse transform -> SB
#pragma unroll
  for (int k = 0; k < EM; ++k) {
    const int e = tid + k * NT;
    if (e < NELEM) s[e] = a[k];
  }
  __syncthreads();
  auto outB = [=](int i, int j, real2 x) { if (j < wc) st_off(bb, off8(i, j), x); };
  fft_tile<NT, EMAX, true, false, true, false, LPC_COLS_FUSEL, SBT>(s, plan, T, cp.tdiv, tid, LdsNatural{}, outB);
  LPC_STAMP_END();
}

// REALPH: the phases are +-1 (even padded sizes: ifftshift = (-1)^k per axis), c2 = conj(c1): only c1 is stored (8 bytes)
template <int NT, bool REALPH>
__global__ __launch_bounds__(NT) void k_mid_consts(const real2* LPC_RESTRICT Hs_t, const real* LPC_RESTRICT Gabs_t,
                                                    const real* LPC_RESTRICT ga, const real* LPC_RESTRICT gb,
                                                    const real2* LPC_RESTRICT phr, const real2* LPC_RESTRICT phc, int Hp,
                                                    int Wc, int cpitch, long cplane, real mu1, real mu2, real mu3,
                                                    real rscale, void* LPC_RESTRICT Cv, real* LPC_RESTRICT RD) {
  const long n = (long)((Hp + 1) & ~1) * cpitch;          // pair-line positions of one plane
  const long pl = blockIdx.y;
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < n; e += (long)gridDim.x * NT) {
    const long pair = e / (2 * cpitch);
    const int w = (int)(e - pair * 2 * cpitch);
    const int i = 2 * (int)pair + ((w >> 3) & 1), j = ((w >> 4) << 3) + (w & 7);
    MidConst mc;
    mc.c1 = mc.c2 = make_real2((real)0., (real)0.);
    real rd = (real)0.;
    if (i < Hp && j < Wc) {
      const real2 hh = Hs_t[pl * cplane + e];
      const real gk = ga ? ga[i] + gb[j] : Gabs_t[e];
      const real2 ph = cmul(phr[i], phc[j]);
      mc.c1 = cmul(make_real2(hh.x, -hh.y), ph);
      mc.c2 = cmul(hh, ph);
      rd = rscale * recip_pos(mu1 * rabs(hh.x * hh.x + hh.y * hh.y) + mu2 * gk + mu3);
    }
    if (REALPH) ((real2*)Cv)[pl * cplane + e] = mc.c1;
    else ((MidConst*)Cv)[pl * cplane + e] = mc;
    RD[pl * cplane + e] = rd;
  }
}

// ============================================================ ADMM spatial kernel ==
// K1: everything of one ADMM iteration that lives in the image domain, in ONE pass:
//   (pending) dual updates of the previous iteration  xi, eta, rho
//   U  = soft(Psi V + eta/mu2, tau/mu2)     X = M (xi + mu1 HV + pad(y))     W = max(rho/mu3 + V, 0)
//   r_sp = (mu3 W - rho) + Psi^T(mu2 U - eta)           a = mu1 X - xi
// U and W are never stored: the previous iteration's U, W are recomputed from V_old (kept by
// ping-ponging the V buffer), which replaces 3 reads + 3 writes of padded arrays by 1 read.
// V / V_old tiles (+1 halo) and q = mu2 U - eta (+1 row / +1 col) are staged in LDS.
// eta is read at halo pixels owned by neighbouring workgroups, so its update is written to a
// second buffer (ping-pong, no extra traffic); every other array is touched only at owned pixels.
struct AdmmScalars {
  real mu1, mu2, mu3;  // this iteration's step sizes
  real thr;            // (real)(tau / mu2)
  real m_in, m_out;    // X_divmat inside / outside the sensor window
  int first;            // 1: no pending dual update (first iteration after reset)
  // the PREVIOUS iteration's values: its dual updates are still pending and its U, W are recomputed.
  // Equal to the current ones for plain ADMM; differ for unrolled ADMM (unrolled_admm.py:171-211)
  real mu1p, mu2p, mu3p, thrp;
  real m_in_p, m_out_p;  // X_divmat of the previous iteration (its X is recomputed, never stored)
  // correctly rounded reciprocals of the step sizes the kernels divide by (see div_by)
  real r_mu2, r_mu3, r_mu2p, r_mu3p;
  // Outside the sensor window the data term has no measurement: X_divmat = 1/mu1 there (admm.py:193), so
  //   X = xi/mu1 + HV,   a = mu1 X - xi = mu1 HV,   xi' = xi + mu1 (HV' - X) = mu1 (HV' - HV)
  // -- neither `a` nor the next xi depends on the stored xi.  xiw: the X half of the forward rows works from HV alone
  // outside the window (3/4 of the padded frame: no xi read, no HV_old read, no xi write); xi_store: this is the last
  // iteration of the lpc_iterate() call, write xi = mu1p (HV - HV_old) there as well so that every read-out between
  // calls (lpc_get_state, plug-and-play entries) finds the whole array valid.
  int xiw, xi_store;
  // Rows that lie outside the sensor window altogether carry a = mu1 HV, and HV of such a row is the (unnormalised)
  // inverse row transform of the spectrum row SB[r] that the last inverse row pass read: rfft(a row) = mu1 Wp SB[r].
  // skipa: the forward row blocks of `a` outside the window do nothing (SB[r] is still in place from the previous
  // iteration of this call; forward pass A applies the factor, ColPass::sc); skiphv: the inverse row blocks of HV
  // outside the window do nothing (no later iteration of this call reads those rows of HV -- the engine keeps the last
  // three iterations of a call complete so that xi, X and HV read out whole).
  int skipa, skiphv;
  // The reference clamps the image estimate IN PLACE whenever _form_image runs (admm.py:331-338: negative entries of
  // the sensor window become 0), and only the W-update ever sees that clamped copy (everything else works from the
  // cached Psi V / H V).  The clamp is a pure function of V, so no copy is kept: clamp_cur / clamp_old say that the
  // W-update of this / of the previous iteration saw clamp(V) / clamp(V_old).
  int clamp_cur, clamp_old;
  int rev;    // the tiled kernel walks its grid backwards (see ColPass::rev)
  // Half-applied duals between the iterations of ONE lpc_iterate() call (k_admm_spatial_v4, option k1_half):
  //   eta~ = eta - mu2 U,   rho~ = rho - mu3 W          (stored by an iteration with half_out)
  //   eta' = eta~ + mu2p Psi V_new,   rho' = rho~ + mu3p V_new      (the pending update, read by one with half_in)
  // -- algebraically eta + mu2p (Psi V_new - U), the reference's update (admm.py:300-311), but U and W of the previous
  // iteration need not be RECOMPUTED from V_old, so the kernel does not read V_old: 9 R -> 8 R of HBM traffic per launch.
  // The first iteration of a call reads, and the last one writes, the plain duals: every other entry point (read-outs,
  // plug-and-play, a caller's psi) sees the state of round 3.  Rounding differs from the reference's order of operations
  // by one ulp of the dual per iteration (float64 build: identical to 1e-16).
  int half_in, half_out;
};
// the estimate as the W-update saw it
static __device__ __forceinline__ real w_sees(real v, bool clamped, bool inside) {
  return (clamped && inside && v < (real)0.) ? (real)0. : v;
}

// x / d for a wave-uniform divisor d whose reciprocal r = RN(1 / d) was rounded on the host: one Newton step on
// q0 = RN(x r) with the exact residual (Markstein's sequence), q = RN(q0 + RN(x - d q0) r), is the correctly
// rounded quotient -- what the reference's `eta / mu2` computes -- in 3 VALU operations; the compiler's IEEE
// division expands to ~10 (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup), which made up 47 % of the
// image-domain kernel's instruction stream (34 divisions per 4 pixels).  No scaling / fix-up: the operands are image
// values and duals far away from the overflow and denormal ranges.
// (The emulator runs the same sequence -- std::fma is exact on the host too -- so the CPU suite checks these three
// operations, not a plain division; -DLPC_EMU_PLAIN_DIV brings `x / d` back for a comparison.)
static __device__ __forceinline__ real div_by(real x, real d, real r) {
#if defined(LPC_SIMT_EMU) && defined(LPC_EMU_PLAIN_DIV)
  (void)r;
  return x / d;
#else
  const real q = x * r;
#ifdef LPC_DOUBLE
  const real e = fma(-d, q, x);
  return fma(e, r, q);
#else
  const real e = fmaf(-d, q, x);
  return fmaf(e, r, q);
#endif
#endif
}

static __device__ __forceinline__ real soft_thresh_dev(real a, real thr) {   // sign(a) max(|a| - thr, 0), admm.py:341-346
  const real m = rmax(rabs(a) - thr, (real)0.);
#if defined(LPC_SIMT_EMU) && defined(LPC_EMU_PLAIN_DIV)
  return a > (real)0. ? m : (a < (real)0. ? -m : (real)0.);
#elif defined(LPC_DOUBLE)
  return copysign(m, a);     // a == 0 gives m == 0: the sign of that zero never reaches a result
#else
  return copysignf(m, a);
#endif
}

template <int TH, int TW, int NT>
__global__ __launch_bounds__(NT) void k_admm_spatial(PlaneGeom g, AdmmScalars p,
                                                      const real* LPC_RESTRICT V,
                                                      const real* LPC_RESTRICT Vold,
                                                      const real* LPC_RESTRICT HV,
                                                      const real* LPC_RESTRICT HVold, real* LPC_RESTRICT xi,
                                                      const real* LPC_RESTRICT eta0,
                                                      const real* LPC_RESTRICT eta1,
                                                      real* LPC_RESTRICT eta0_out,
                                                      real* LPC_RESTRICT eta1_out,
                                                      real* LPC_RESTRICT rho,
                                                      const real* LPC_RESTRICT Y,
                                                      real* LPC_RESTRICT Rsp, real* LPC_RESTRICT Aout,
                                                      unsigned tiles_x) {
  LPC_DYN_SMEM(smem);
  constexpr int VW = TW + 2, VH = TH + 2;
  real* sV = (real*)smem;                 // [VH][VW], local (ly+1, lx+1)
  real* sO = sV + VH * VW;                 // same for V_old
  real* sQ0 = sO + VH * VW;                // [TH+1][TW]
  real* sQ1 = sQ0 + (TH + 1) * TW;         // [TH][TW+1]
  const int tid = LPC_TID(NT);
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch rule, used for speed
  // only).  Give each XCD a contiguous band of tiles so that the halo lines shared by neighbouring
  // tiles are re-read from the SAME L2 instead of from the fabric.
  const unsigned nblk = gridDim.x, b = blockIdx.x;
  const unsigned q = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
  const unsigned tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const unsigned ty_ = tile / tiles_x;
  const int r0 = (int)ty_ * TH, c0 = (int)(tile - ty_ * tiles_x) * TW;
  const long pl = blockIdx.y;
  const long poff = pl * g.rplane;
  const real* v = V + poff;
  const real* vo = Vold + poff;

  // ---- stage V, V_old (+halo, circular) ----
  for (int e = tid; e < VH * VW; e += NT) {
    const int ly = e / VW, lx = e - ly * VW;
    int gr = r0 + ly - 1, gc = c0 + lx - 1;
    gr = gr < 0 ? gr + g.Hp : gr; gr = gr >= g.Hp ? gr - g.Hp : gr; gr = gr >= g.Hp ? gr % g.Hp : gr;
    gc = gc < 0 ? gc + g.Wp : gc; gc = gc >= g.Wp ? gc - g.Wp : gc; gc = gc >= g.Wp ? gc % g.Wp : gc;
    const long o = (long)gr * g.rpitch + gc;
    sV[e] = v[o];
    sO[e] = p.first ? (real)0. : vo[o];
  }
  __syncthreads();

  // ---- q0 over [0,TH] x [0,TW), q1 over [0,TH) x [0,TW]; eta' stored for owned pixels ----
  for (int e = tid; e < (TH + 1) * (TW + 1); e += NT) {
    const int ly = e / (TW + 1), lx = e - ly * (TW + 1);
    int gr = r0 + ly, gc = c0 + lx;
    const bool own = (ly < TH) && (lx < TW) && (gr < g.Hp) && (gc < g.Wp);
    gr = gr >= g.Hp ? gr % g.Hp : gr;
    gc = gc >= g.Wp ? gc % g.Wp : gc;
    const long o = poff + (long)gr * g.rpitch + gc;
    const int li = (ly + 1) * VW + (lx + 1);
    const real vc = sV[li], oc = sO[li];
    if (lx < TW) {  // component 0 (row difference)
      real e0 = eta0[o];
      const real psi = sV[li - VW] - vc;
      if (!p.first) {
        const real psio = sO[li - VW] - oc;
        const real uo = soft_thresh_dev(psio + div_by(e0, p.mu2p, p.r_mu2p), p.thrp);
        e0 = e0 + p.mu2p * (psi - uo);
      }
      const real un = soft_thresh_dev(psi + div_by(e0, p.mu2, p.r_mu2), p.thr);
      sQ0[ly * TW + lx] = p.mu2 * un - e0;
      if (own) eta0_out[o] = e0;
    }
    if (ly < TH) {  // component 1 (column difference)
      real e1 = eta1[o];
      const real psi = sV[li - 1] - vc;
      if (!p.first) {
        const real psio = sO[li - 1] - oc;
        const real uo = soft_thresh_dev(psio + div_by(e1, p.mu2p, p.r_mu2p), p.thrp);
        e1 = e1 + p.mu2p * (psi - uo);
      }
      const real un = soft_thresh_dev(psi + div_by(e1, p.mu2, p.r_mu2), p.thr);
      sQ1[ly * (TW + 1) + lx] = p.mu2 * un - e1;
      if (own) eta1_out[o] = e1;
    }
  }
  __syncthreads();

  // ---- owned pixels: xi, rho, X, W, r_sp, a ----
  const int dpl = (int)(pl / g.DC) * g.C + (int)(pl % g.C);
  const real* y = Y + (long)dpl * g.uplane;
  for (int e = tid; e < TH * TW; e += NT) {
    const int ly = e / TW, lx = e - ly * TW;
    const int gr = r0 + ly, gc = c0 + lx;
    if (gr >= g.Hp || gc >= g.Wp) continue;
    const long o = poff + (long)gr * g.rpitch + gc;
    const int li = (ly + 1) * VW + (lx + 1);
    const real vc = sV[li];
    const real hv = HV[o];
    real xiv = xi[o], rhov = rho[o];
    const bool inside = (gr >= g.sh) && (gr < g.sh + g.H) && (gc >= g.sw) && (gc < g.sw + g.W);
    const real yv = inside ? y[(long)(gr - g.sh) * g.W + (gc - g.sw)] : (real)0;
    if (!p.first) {
      // X of the previous iteration, recomputed bit-for-bit from what it was computed from (admm.py:252-254)
      const real xo = (inside ? p.m_in_p : p.m_out_p) * (xiv + p.mu1p * HVold[o] + yv);
      xiv = xiv + p.mu1p * (hv - xo);
      const real wo = rmax(div_by(rhov, p.mu3p, p.r_mu3p) + w_sees(sO[li], p.clamp_old, inside), (real)0);
      rhov = rhov + p.mu3p * (vc - wo);
    }
    const real xn = (inside ? p.m_in : p.m_out) * (xiv + p.mu1 * hv + yv);
    const real wn = rmax(div_by(rhov, p.mu3, p.r_mu3) + w_sees(vc, p.clamp_cur, inside), (real)0);
    const real d1 = sQ0[(ly + 1) * TW + lx] - sQ0[ly * TW + lx];
    const real d2 = sQ1[ly * (TW + 1) + lx + 1] - sQ1[ly * (TW + 1) + lx];
    xi[o] = xiv;
    rho[o] = rhov;
    Rsp[o] = (p.mu3 * wn - rhov) + (d1 + d2);
    Aout[o] = p.mu1 * xn - xiv;
  }
}

// ---- K1, 16-byte-lane version (padded width a multiple of 4) -----------------------------------
// Same arithmetic as k_admm_spatial, re-shaped for HBM: every lane moves four pixels (float4: a wave covers 1 KiB of
// one image row per array; the float64 build moves two 16-byte halves), tiles are TH rows x 256 columns, only V / V_old
// are staged in LDS (+1 halo, circular); q = mu2 U - eta of the lower / right neighbour is RECOMPUTED in registers from
// the LDS tile and one extra (L1/L2-resident) load of eta instead of being exchanged through LDS, so there is
// one barrier and 21 KiB of LDS per workgroup (7 workgroups per CU).

// eta' and q for one pixel and one difference direction
static __device__ __forceinline__ void tv_component(const AdmmScalars& p, real vc, real vn, real oc, real on,
                                                     real eta, real& eta_new, real& q) {
  const real psi = vn - vc;                       // finite_diff: roll(+1) - x   (admm.py:349-359)
  if (!p.first) {
    if (p.half_in) {
      eta = eta + p.mu2p * psi;                    // stored: eta - mu2p U_old (AdmmScalars::half_in)
    } else {
      const real uo = soft_thresh_dev((on - oc) + div_by(eta, p.mu2p, p.r_mu2p), p.thrp);
      eta = eta + p.mu2p * (psi - uo);             // pending eta update of the previous iteration
    }
  }
  const real un = soft_thresh_dev(psi + div_by(eta, p.mu2, p.r_mu2), p.thr);
  q = p.mu2 * un - eta;
  eta_new = p.half_out ? -q : eta;
}

// XHALF == false: the TV / W half only (eta, rho, r_sp); xi and a = mu1 X - xi are then produced by the forward row
// kernel itself (k_rfwd_half_x / k_rfwd_arrays_x), which needs nothing but its own row for them.
// HIN: the duals arrive half-applied (AdmmScalars::half_in): no V_old tile is staged or read.
template <int TH, int NT, bool XHALF = true, bool HIN = false>
__global__ __launch_bounds__(NT) void k_admm_spatial_v4(PlaneGeom g, AdmmScalars p,
                                                         const real* LPC_RESTRICT V,
                                                         const real* LPC_RESTRICT Vold,
                                                         const real* LPC_RESTRICT HV,
                                                         const real* LPC_RESTRICT HVold, real* LPC_RESTRICT xi,
                                                         const real* LPC_RESTRICT eta0,
                                                         const real* LPC_RESTRICT eta1,
                                                         real* LPC_RESTRICT eta0_out,
                                                         real* LPC_RESTRICT eta1_out,
                                                         real* LPC_RESTRICT rho,
                                                         const real* LPC_RESTRICT Y,
                                                         real* LPC_RESTRICT Rsp, real* LPC_RESTRICT Aout,
                                                         unsigned tiles_x) {
  LPC_DYN_SMEM(smem);
  constexpr int TW = 256, LP = TW + 8;          // LDS row: [3] = col -1, [4..259] = cols 0..255, [260] = col 256
  constexpr int VH = TH + 2;
  real* sV = (real*)smem;                     // [VH][LP]
  real* sO = sV + VH * LP;
  const int tid = LPC_TID(NT);
  const unsigned nblk = gridDim.x, bid = p.rev ? gridDim.x - 1u - blockIdx.x : blockIdx.x;   // XCD-aware tile order (see k_admm_spatial)
  const unsigned qd = nblk >> 3, rm = nblk & 7, xcd = bid & 7, idx = bid >> 3;
  const unsigned tile = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
  const unsigned ty_ = tile / tiles_x;
  const int r0 = (int)ty_ * TH, c0 = (int)(tile - ty_ * tiles_x) * TW;
  const long pl = p.rev ? gridDim.y - 1u - blockIdx.y : blockIdx.y;
  const long poff = pl * g.rplane;
  const real* v = V + poff;
  const real* vo = Vold + poff;
  // circular neighbours: -1 -> n - 1 and n -> 0 are the only wraps a VALID pixel ever needs (its neighbours lie in
  // [-1, n]); indices further out belong to the overhang of the last tile, whose values no valid pixel reads -- they are
  // clamped to a safe address.  (A general `% n` here was 8 integer divisions per lane: a third of the kernel's VALU.)
  auto wrap = [](int x, int n) { x = x < 0 ? x + n : x; x = x >= n ? x - n : x; return x < n ? x : n - 1; };
  auto wrap_r = [&](int r) { return wrap(r, g.Hp); };
  auto wrap_c = [&](int c) { return wrap(c, g.Wp); };
  auto wrap_c4 = [&](int c) { c = c >= g.Wp ? c - g.Wp : c; return c < g.Wp ? c : g.Wp - 4; };   // quads: c, Wp multiples of 4

  // ---- stage V, V_old: body as real4, the two halo columns as scalars ----
  for (int e = tid; e < VH * (TW / 4); e += NT) {
    const int ly = e / (TW / 4), l4 = e - ly * (TW / 4);
    const long o = (long)wrap_r(r0 + ly - 1) * g.rpitch + wrap_c4(c0 + 4 * l4);
    st4(sV + ly * LP + 4 + 4 * l4, ld4(v + o));
    if (!HIN) st4(sO + ly * LP + 4 + 4 * l4, p.first ? make_real4((real)0., (real)0., (real)0., (real)0.) : ld4(vo + o));
  }
  for (int e = tid; e < VH * 2; e += NT) {
    const int ly = e >> 1, side = e & 1;
    const long o = (long)wrap_r(r0 + ly - 1) * g.rpitch + wrap_c(side ? c0 + TW : c0 - 1);
    sV[ly * LP + (side ? 4 + TW : 3)] = v[o];
    if (!HIN) sO[ly * LP + (side ? 4 + TW : 3)] = p.first ? (real)0. : vo[o];
  }
  __syncthreads();

  const int dpl = (int)(pl / g.DC) * g.C + (int)(pl % g.C);
  const real* y = Y + (long)dpl * g.uplane;
  const int lane = tid & 63, wv = tid >> 6;
  const int gc = c0 + 4 * lane;
#pragma unroll
  for (int rr = 0; rr < TH / (NT / 64); ++rr) {
    const int ly = wv + rr * (NT / 64);
    const int gr = r0 + ly;
    if (gr >= g.Hp || gc >= g.Wp) continue;
    const long o = poff + (long)gr * g.rpitch + gc;
    const long o_dn = poff + (long)wrap_r(gr + 1) * g.rpitch + gc;        // eta0 of the row below
    const long o_rt = poff + (long)gr * g.rpitch + wrap_c(gc + 4);        // eta1 of the pixel right of the quad
    // global loads first (all independent)
    const real4 z4 = make_real4((real)0., (real)0., (real)0., (real)0.);
    const real4 hv4 = XHALF ? ld4(HV + o) : z4, xi4 = XHALF ? ld4(xi + o) : z4, rho4 = ld4(rho + o);
    const real4 e04 = ld4(eta0 + o), e14 = ld4(eta1 + o), e0d4 = ld4(eta0 + o_dn);
    const real e1r = eta1[o_rt];
    real4 ho4 = z4;
    if (XHALF && !p.first) ho4 = ld4(HVold + o);      // previous H V: lets the previous X be recomputed instead of stored
    // LDS neighbourhood: rows ly-1, ly, ly+1 of the quad, plus the pixel left and right of it
    const real* rowm = sV + ly * LP + 4 + 4 * lane;        // global row gr-1  (local ly)
    const real* rowc = rowm + LP;                          // gr
    const real* rowp = rowc + LP;                          // gr+1
    const real* orm = sO + ly * LP + 4 + 4 * lane;
    const real* orc = orm + LP;
    const real* orp = orc + LP;
    const real4 vm4 = ld4(rowm), vc4 = ld4(rowc), vp4 = ld4(rowp);
    const real4 zo4 = make_real4((real)0., (real)0., (real)0., (real)0.);
    const real4 om4 = HIN ? zo4 : ld4(orm), oc4 = HIN ? zo4 : ld4(orc), op4 = HIN ? zo4 : ld4(orp);
    const real vl = rowc[-1], vr = rowc[4], ol = HIN ? (real)0. : orc[-1], orr = HIN ? (real)0. : orc[4];
    const real vcs[6] = {vl, vc4.x, vc4.y, vc4.z, vc4.w, vr};       // cols gc-1 .. gc+4 of row gr
    const real ocs[6] = {ol, oc4.x, oc4.y, oc4.z, oc4.w, orr};
    const real vms[4] = {vm4.x, vm4.y, vm4.z, vm4.w}, vps[4] = {vp4.x, vp4.y, vp4.z, vp4.w};
    const real oms[4] = {om4.x, om4.y, om4.z, om4.w}, ops[4] = {op4.x, op4.y, op4.z, op4.w};
    const real hvs[4] = {hv4.x, hv4.y, hv4.z, hv4.w}, xis[4] = {xi4.x, xi4.y, xi4.z, xi4.w};
    const real rhs[4] = {rho4.x, rho4.y, rho4.z, rho4.w}, hos[4] = {ho4.x, ho4.y, ho4.z, ho4.w};
    const real e0s[4] = {e04.x, e04.y, e04.z, e04.w}, e0ds[4] = {e0d4.x, e0d4.y, e0d4.z, e0d4.w};
    const real e1s[5] = {e14.x, e14.y, e14.z, e14.w, e1r};
    real q1[5], e1n[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)   // column-difference component at cols gc .. gc+4 (the 5th only for q)
      tv_component(p, vcs[i + 1], vcs[i], ocs[i + 1], ocs[i], e1s[i], e1n[i], q1[i]);
    real xin[4], e0n[4], rhn[4], rs[4], as[4];
    const bool row_in = (gr >= g.sh) && (gr < g.sh + g.H);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      real q0c, q0d, dummy;
      tv_component(p, vcs[i + 1], vms[i], ocs[i + 1], oms[i], e0s[i], e0n[i], q0c);   // this pixel
      tv_component(p, vps[i], vcs[i + 1], ops[i], ocs[i + 1], e0ds[i], dummy, q0d);   // the pixel below
      const real vc = vcs[i + 1], hv = hvs[i];
      real xiv = xis[i], rhov = rhs[i];
      const int cc = gc + i;
      const bool inside = row_in && (cc >= g.sw) && (cc < g.sw + g.W);
      const real yv = (XHALF && inside) ? y[(long)(gr - g.sh) * g.W + (cc - g.sw)] : (real)0.;
      if (!p.first) {
        if (XHALF) {
          const real xo = (inside ? p.m_in_p : p.m_out_p) * (xiv + p.mu1p * hos[i] + yv);   // previous X
          xiv = xiv + p.mu1p * (hv - xo);
        }
        if (HIN || p.half_in) {
          rhov = rhov + p.mu3p * vc;                // stored: rho - mu3p W_old (AdmmScalars::half_in)
        } else {
          const real wo = rmax(div_by(rhov, p.mu3p, p.r_mu3p) + w_sees(ocs[i + 1], p.clamp_old, inside), (real)0.);
          rhov = rhov + p.mu3p * (vc - wo);
        }
      }
      const real xnew = (inside ? p.m_in : p.m_out) * (xiv + p.mu1 * hv + yv);
      const real wn = rmax(div_by(rhov, p.mu3, p.r_mu3) + w_sees(vc, p.clamp_cur, inside), (real)0.);
      const real d1 = q0d - q0c;
      const real d2 = q1[i + 1] - q1[i];
      xin[i] = xiv; rhn[i] = p.half_out ? rhov - p.mu3 * wn : rhov;
      rs[i] = (p.mu3 * wn - rhov) + (d1 + d2);
      as[i] = p.mu1 * xnew - xiv;
    }
    if (XHALF) st4(xi + o, make_real4(xin[0], xin[1], xin[2], xin[3]));
    st4(rho + o, make_real4(rhn[0], rhn[1], rhn[2], rhn[3]));
    st4(eta0_out + o, make_real4(e0n[0], e0n[1], e0n[2], e0n[3]));
    st4(eta1_out + o, make_real4(e1n[0], e1n[1], e1n[2], e1n[3]));
    st4(Rsp + o, make_real4(rs[0], rs[1], rs[2], rs[3]));
    if (XHALF) st4(Aout + o, make_real4(as[0], as[1], as[2], as[3]));
  }
}

// ---- the X half of the image-domain work on four adjacent pixels of one row (k_rfwd_half_x, k_rfwd_arrays_x) ---------
// Loads first (xhalf_load: up to four 16-byte loads, all independent -- callers issue the loads of every quad they own
// before the arithmetic of the first), then the statements of the X part of k_admm_spatial_v4 (xhalf_apply).
struct XQuad { real4 hv, xi, ho; real ys[4]; bool ins[4]; bool skip; };
// row_in: the row lies inside the sensor window; y: that row of the measurement (dereferenced only when row_in);
// y4: window offset and width allow 16-byte loads of y
static __device__ __forceinline__ XQuad xhalf_load(const PlaneGeom& g, const AdmmScalars& p, const real* LPC_RESTRICT HV,
                                                   const real* LPC_RESTRICT HVold, const real* xi,
                                                   const real* LPC_RESTRICT y, long o_row, bool row_in, bool y4, int q) {
  XQuad c;
  const int gc = 4 * q;
  // the whole quad lies outside the sensor window: HV is all it needs (see AdmmScalars::xiw)
  c.skip = p.xiw && !(row_in && gc + 4 > g.sw && gc < g.sw + g.W);
  c.hv = ld4(HV + o_row + gc);
  c.xi = c.ho = make_real4((real)0., (real)0., (real)0., (real)0.);
  if (!c.skip) c.xi = ld4(xi + o_row + gc);
  if (!p.first && (!c.skip || p.xi_store)) c.ho = ld4(HVold + o_row + gc);
#pragma unroll
  for (int i = 0; i < 4; ++i) { c.ys[i] = (real)0.; c.ins[i] = false; }
  if (row_in) {
    if (y4) {
      if (gc >= g.sw && gc < g.sw + g.W) {
        const real4 yv = ld4(y + (gc - g.sw));
        c.ys[0] = yv.x; c.ys[1] = yv.y; c.ys[2] = yv.z; c.ys[3] = yv.w;
        c.ins[0] = c.ins[1] = c.ins[2] = c.ins[3] = true;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cc = gc + i;
        c.ins[i] = (cc >= g.sw) && (cc < g.sw + g.W);
        if (c.ins[i]) c.ys[i] = y[cc - g.sw];
      }
    }
  }
  return c;
}
// xin: the updated xi (stored by the caller unless c.skip && !p.xi_store); as: a = mu1 X - xi'
static __device__ __forceinline__ void xhalf_apply(const AdmmScalars& p, const XQuad& c, real xin[4], real as[4]) {
  const real hvs[4] = {c.hv.x, c.hv.y, c.hv.z, c.hv.w}, xis[4] = {c.xi.x, c.xi.y, c.xi.z, c.xi.w};
  const real hos[4] = {c.ho.x, c.ho.y, c.ho.z, c.ho.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    real xiv = xis[i];
    const real hv = hvs[i], yv = c.ys[i];
    if (p.xiw && !c.ins[i]) {      // outside the window: a = mu1 HV, xi = mu1p (HV - HV_old) (stored on request only)
      xin[i] = p.first ? (real)0. : p.mu1p * (hv - hos[i]);
      as[i] = p.mu1 * hv;
      continue;
    }
    if (!p.first) {
      const real xo = (c.ins[i] ? p.m_in_p : p.m_out_p) * (xiv + p.mu1p * hos[i] + yv);   // previous X
      xiv = xiv + p.mu1p * (hv - xo);
    }
    const real xnew = (c.ins[i] ? p.m_in : p.m_out) * (xiv + p.mu1 * hv + yv);
    xin[i] = xiv;
    as[i] = p.mu1 * xnew - xiv;
  }
}

// ---- forward rows of r_sp and a, with the X half of the image-domain work (wide frames: one real row per half-length
// transform) -------------------------------------------------------------------------------------------------------
// The image-domain work of an ADMM iteration separates cleanly:
//   r_sp = (mu3 W - rho') + Psi^T(mu2 U - eta')  depends on V, V_old, eta, rho only and needs neighbours (TV stencil),
//   a    = mu1 X - xi'                           depends on xi, HV, HV_old, y only and needs nothing but its own pixel.
// The tiled kernel (k_admm_spatial_v4<.., XHALF = false>) keeps the stencil half at its own occupancy and writes r_sp;
// here block (row, 0) transforms that stored row exactly like k_rfwd_half, and block (row, 1) COMPUTES the row of `a`
// (and the pending xi update) from xi, HV, HV_old, y in registers and transforms it: `a` never exists in HBM and the
// tiled kernel no longer touches xi, HV, HV_old, y (-2R per iteration).  Same arithmetic, statement for statement, as
// the X part of k_admm_spatial_v4.  grid = (2 * Hp, planes); `plan` has length Wp/2, `twW` is the length-Wp table.
// (Three ways to split the image-domain work were built and measured, profiles/r02_notes.md: the stand-alone kernel,
// everything inside the forward rows -- its stencil half then runs at the row kernel's 4 workgroups per CU, no faster
// than the pair once the rows run on compile-time plans -- and this one, a win on every box and shape.)
template <int NT, int EMAX, int SK, class PL = Fft1dPlan>
__global__ __launch_bounds__(NT) void k_rfwd_half_x(PlaneGeom g, AdmmScalars p, PL plan, const real2* LPC_RESTRICT twW,
                                                     const real* LPC_RESTRICT Rsp, const real* LPC_RESTRICT HV,
                                                     const real* LPC_RESTRICT HVold, real* LPC_RESTRICT xi,
                                                     const real* LPC_RESTRICT Y, real2* LPC_RESTRICT SA,
                                                     real2* LPC_RESTRICT SB) {
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int tid = LPC_TID(NT);
  // No row needs a neighbour, and rows inside the sensor window cost more than rows outside (AdmmScalars::xiw): bands of
  // rows per XCD would leave the XCDs that hold the window rows working while the others idle, so rows go round-robin
  // over the XCDs (block b runs on XCD b % 8: the even XCDs transform the rows of r_sp, the odd ones form and transform
  // the rows of `a` -- about twice the work per row, on half of the rows once those outside the window are skipped.
  // Flipping the pair every fourth row to mix both kinds on every XCD measured SLOWER: 0.47 -> 0.60 ms, r02ak; a compact
  // grid without the empty blocks of skipped `a` rows measured the same: 0.498 / 0.480 vs 0.499 / 0.481 ms, r02an)
  const unsigned bx = LPC_BX(g);
  const int gr = (int)(bx >> 1), arr = (int)(bx & 1);
  const long pl = LPC_BY(g);
  const long poff = pl * g.rplane;
  const long o_row = poff + (long)gr * g.rpitch;
  const int n4 = g.Wp >> 2;
  if (arr == 1 && p.skipa && (gr < g.sh || gr >= g.sh + g.H)) return;   // AdmmScalars::skipa (uniform per block)
  if (arr == 0) {      // the stored row of r_sp: first stage fused into the fill, like k_rfwd_half
    const real2* a2 = (const real2*)(Rsp + o_row);
    auto src = [&](int i, int) { return a2[i]; };
    fft_tile<NT, EMAX, false, SK, false, true>(s, plan, 1, make_fastdiv_dev1(), tid, src, LdsNatural{});
    untangle_half_store<NT, SK>(s, g.Wp >> 1, twW, SA + pl * g.cplane + (long)gr * g.cpitch, tid);
    return;
  }
  {
    const int dpl = (int)(pl / g.DC) * g.C + (int)(pl % g.C);
    const bool row_in = (gr >= g.sh) && (gr < g.sh + g.H);
    const real* y = Y + (long)dpl * g.uplane + (long)(gr - g.sh) * g.W;     // dereferenced only when row_in
    const bool y4 = ((g.sw | g.W) & 3) == 0;                                 // window and pitch allow real4 loads of y
    XQuad nxt = xhalf_load(g, p, HV, HVold, xi, y, o_row, row_in, y4, tid < n4 ? tid : 0);
#pragma unroll 1
    for (int q = tid; q < n4; q += NT) {
      const int gc = 4 * q;
      const XQuad c = nxt;
      if (q + NT < n4) nxt = xhalf_load(g, p, HV, HVold, xi, y, o_row, row_in, y4, q + NT);
      real xin[4], as[4];
      xhalf_apply(p, c, xin, as);
      if (!c.skip || p.xi_store) st4(xi + o_row + gc, make_real4(xin[0], xin[1], xin[2], xin[3]));
      s[lds_slot<SK>(2 * q)] = make_real2(as[0], as[1]);
      s[lds_slot<SK>(2 * q + 1)] = make_real2(as[2], as[3]);
    }
  }
  __syncthreads();
  fft_tile<NT, EMAX, false, SK, false>(s, plan, 1, make_fastdiv_dev1(), tid, LdsNatural{}, LdsNatural{});
  untangle_half_store<NT, SK>(s, g.Wp >> 1, twW, SB + pl * g.cplane + (long)gr * g.cpitch, tid);
}

// ---- the TV / W half of the image-domain work on the quads of TWO adjacent rows (k_rfwd_arrays_x<.., K1 = true>) ----
// Small frames are a chain of dependent launches, each a chain of memory latencies (profiles/r05_notes.md section 5):
// here the forward row block of r_sp rows (r0, r0 + 1) forms them itself -- the statements of k_admm_spatial_v4<..,
// XHALF = false>, quad by quad, with the stencil's neighbours read straight from global memory (L2-resident at these
// sizes) instead of a staged tile -- so that an iteration is three launches instead of four and r_sp never exists in
// memory.  v, vo, eta*, rho: plane bases; q: the lane's quad (columns 4 q .. 4 q + 3); second: row r0 + 1 exists; the
// rows go into the tile s as z = r_sp[r0] + i r_sp[r0 + 1].  Rows wrap circularly like the tiled kernel's; eta is read at
// rows owned by other blocks, hence eta*_out (ping-pong).
template <int SK>
static __device__ __forceinline__ void k1_two_rows(const PlaneGeom& g, const AdmmScalars& p, const real* LPC_RESTRICT v,
                                                   const real* LPC_RESTRICT vo, const real* LPC_RESTRICT eta0,
                                                   const real* LPC_RESTRICT eta1, real* LPC_RESTRICT eta0_out,
                                                   real* LPC_RESTRICT eta1_out, real* rho, int r0, bool second, int q,
                                                   real2* s) {
  constexpr unsigned eb = (unsigned)sizeof(real);
  const int gc = 4 * q;
  // 32-bit byte offsets from the plane bases (a plane is < 4 GB); circular column / row neighbours (a row that does not
  // exist is never used: `second`)
  const unsigned bq = (unsigned)gc * eb;
  const unsigned bl = (unsigned)(gc == 0 ? g.Wp - 1 : gc - 1) * eb, br = (unsigned)(gc + 4 >= g.Wp ? 0 : gc + 4) * eb;
  const unsigned rp = (unsigned)g.rpitch * eb;
  unsigned ro[4];
  ro[0] = (unsigned)(r0 == 0 ? g.Hp - 1 : r0 - 1) * rp;
  ro[1] = (unsigned)r0 * rp;
  ro[2] = (unsigned)(r0 + 1 >= g.Hp ? r0 + 1 - g.Hp : r0 + 1) * rp;
  ro[3] = (unsigned)(r0 + 2 >= g.Hp ? r0 + 2 - g.Hp : r0 + 2) * rp;
  auto L4 = [](const real* b, unsigned off) { return *(const real4*)((const char*)b + off); };
  auto L1 = [](const real* b, unsigned off) { return *(const real*)((const char*)b + off); };
  auto S4 = [](real* b, unsigned off, real4 x) { *(real4*)((char*)b + off) = x; };
  const bool need_old = !p.first && !p.half_in;
  // every load of both rows first -- except V_old's, which only a call's first iteration without a reset (or k1_half=0)
  // reads: those are requested row by row below (their registers would otherwise cost every launch a workgroup per CU)
  real4 v4[4], e04[3], e14[2], rh4[2];
  real vl[2], vr[2], e1r[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) v4[k] = L4(v, ro[k] + bq);
#pragma unroll
  for (int k = 0; k < 2; ++k) { vl[k] = L1(v, ro[1 + k] + bl); vr[k] = L1(v, ro[1 + k] + br); }
#pragma unroll
  for (int k = 0; k < 3; ++k) e04[k] = L4(eta0, ro[1 + k] + bq);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    e14[k] = L4(eta1, ro[1 + k] + bq);
    e1r[k] = L1(eta1, ro[1 + k] + br);
    rh4[k] = L4(rho, ro[1 + k] + bq);
  }
#pragma unroll
  for (int row = 0; row < 2; ++row) {
    if (row == 1 && !second) {
#pragma unroll
      for (int i = 0; i < 4; ++i) s[lds_slot<SK>(gc + i)].y = (real)0.;
      break;
    }
    const int gr = r0 + row;
    const real4 z4 = make_real4((real)0., (real)0., (real)0., (real)0.);
    real4 om4 = z4, oc4 = z4, op4 = z4;
    real ol = (real)0., orr = (real)0.;
    if (need_old) {
      om4 = L4(vo, ro[row] + bq); oc4 = L4(vo, ro[row + 1] + bq); op4 = L4(vo, ro[row + 2] + bq);
      ol = L1(vo, ro[row + 1] + bl); orr = L1(vo, ro[row + 1] + br);
    }
    const real4 vm4 = v4[row], vc4 = v4[row + 1], vp4 = v4[row + 2];
    const real vcs[6] = {vl[row], vc4.x, vc4.y, vc4.z, vc4.w, vr[row]};       // cols gc-1 .. gc+4 of row gr
    const real ocs[6] = {ol, oc4.x, oc4.y, oc4.z, oc4.w, orr};
    const real vms[4] = {vm4.x, vm4.y, vm4.z, vm4.w}, vps[4] = {vp4.x, vp4.y, vp4.z, vp4.w};
    const real oms[4] = {om4.x, om4.y, om4.z, om4.w}, ops[4] = {op4.x, op4.y, op4.z, op4.w};
    const real4 rho4 = rh4[row], e0c4 = e04[row], e0d4 = e04[row + 1], e1c4 = e14[row];
    const real rhs[4] = {rho4.x, rho4.y, rho4.z, rho4.w};
    const real e0s[4] = {e0c4.x, e0c4.y, e0c4.z, e0c4.w}, e0ds[4] = {e0d4.x, e0d4.y, e0d4.z, e0d4.w};
    const real e1s[5] = {e1c4.x, e1c4.y, e1c4.z, e1c4.w, e1r[row]};
    real q1[5], e1n[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)   // column-difference component at cols gc .. gc+4 (the 5th only for q)
      tv_component(p, vcs[i + 1], vcs[i], ocs[i + 1], ocs[i], e1s[i], e1n[i], q1[i]);
    real e0n[4], rhn[4];
    const bool row_in = (gr >= g.sh) && (gr < g.sh + g.H);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      real q0c, q0d, dummy;
      tv_component(p, vcs[i + 1], vms[i], ocs[i + 1], oms[i], e0s[i], e0n[i], q0c);   // this pixel
      tv_component(p, vps[i], vcs[i + 1], ops[i], ocs[i + 1], e0ds[i], dummy, q0d);   // the pixel below
      const real vc = vcs[i + 1];
      real rhov = rhs[i];
      const int cc = gc + i;
      const bool inside = row_in && (cc >= g.sw) && (cc < g.sw + g.W);
      if (!p.first) {
        if (p.half_in) {
          rhov = rhov + p.mu3p * vc;                // stored: rho - mu3p W_old (AdmmScalars::half_in)
        } else {
          const real wo = rmax(div_by(rhov, p.mu3p, p.r_mu3p) + w_sees(ocs[i + 1], p.clamp_old, inside), (real)0.);
          rhov = rhov + p.mu3p * (vc - wo);
        }
      }
      const real wn = rmax(div_by(rhov, p.mu3, p.r_mu3) + w_sees(vc, p.clamp_cur, inside), (real)0.);
      const real d1 = q0d - q0c;
      const real d2 = q1[i + 1] - q1[i];
      rhn[i] = p.half_out ? rhov - p.mu3 * wn : rhov;
      const real rs = (p.mu3 * wn - rhov) + (d1 + d2);
      // z = r_sp[r0] + i r_sp[r0 + 1]: the tile entry's real / imaginary part
      if (row == 0) s[lds_slot<SK>(gc + i)].x = rs;
      else s[lds_slot<SK>(gc + i)].y = rs;
    }
    const unsigned o = ro[1 + row] + bq;
    S4(rho, o, make_real4(rhn[0], rhn[1], rhn[2], rhn[3]));
    S4(eta0_out, o, make_real4(e0n[0], e0n[1], e0n[2], e0n[3]));
    S4(eta1_out, o, make_real4(e1n[0], e1n[1], e1n[2], e1n[3]));
  }
}

// ---- paired forward rows with the X half computed on the fly (narrow frames: C1 / C4, 760 x 1014) ---------------
// Same split of the image-domain work as k_rfwd_half_x, for frames whose rows ride in pairs (paired_rows_of: two rows of
// ONE array per transform).  Blocks of array 0 transform two stored rows of r_sp (k_admm_spatial_v4<.., XHALF = false>
// wrote them); blocks of array 1 form two rows of a = mu1 X - xi' element by element from xi, HV, HV_old and y inside
// the source functor of the first FFT stage (which also stores xi').  p.skipa: `a` on the rows of the sensor window
// alone -- outside it a = mu1 HV needs no transform, SB keeps the row spectra the last inverse row pass read and the
// fused middle rescales them (AdmmScalars::skipa).  Compile-time plans only.
#ifndef LPC_K1ROWS_MINW
#define LPC_K1ROWS_MINW 1
#endif
#ifndef LPC_RFWDX_MINW
#define LPC_RFWDX_MINW 1
#endif
// K1 (rows of at most 4 NT columns): the blocks of array 0 form their two rows of r_sp themselves (k1_two_rows: V,
// V_old, eta, rho in; eta', rho' out) instead of reading what the tiled kernel stored -- that kernel is not launched.
struct K1Rows {
  const real *V, *Vold, *eta0, *eta1;
  real *eta0_out, *eta1_out, *rho;
  int xcd_order;      // hand the row blocks out XCD by XCD: -1 an eighth of the launch each (small launches), G > 0 runs of G
                      // blocks (see the kernel), 0 launch order
};
template <int NT, int EMAX, int SK, class PL, bool K1 = false, int SL = 0>
__global__ __launch_bounds__(NT, K1 ? LPC_K1ROWS_MINW : NT == 256 ? LPC_RFWDX_MINW : 1) void k_rfwd_arrays_x(PlaneGeom g, AdmmScalars p, PL plan, const real* LPC_RESTRICT Rsp,
                                                       const real* LPC_RESTRICT HV, const real* LPC_RESTRICT HVold,
                                                       real* LPC_RESTRICT xi, const real* LPC_RESTRICT Y,
                                                       real2* LPC_RESTRICT SA, real2* LPC_RESTRICT SB, K1Rows k1) {
  LPC_DYN_SMEM(smem);
  real2* s = (real2*)smem;
  const int tid = LPC_TID(NT);
  LPC_STAMP_BEGIN(1);
  unsigned bx = LPC_BX(g), by = LPC_BY(g);
  if (K1 && k1.xcd_order) {
    // XCD-aware order (workgroup w runs on XCD w % 8, each with its own L2): the blocks of r_sp read V at rows r0 - 1 and
    // r0 + 2 and eta at row r0 + 2 -- rows that belong to the neighbouring row pairs, which in launch order sit on OTHER
    // XCDs.  Here every XCD walks a contiguous eighth of the (plane, row pair) space.  The host asks for it on small
    // launches only (same box, trees: C1 0.217 -> 0.212 ms, 380 x 507 0.281 -> 0.269 ms per 5 iterations; 8 frames
    // unchanged; C4's 64 frames 0.815 -> 0.855 ms per launch -- profiles/r05zb_xcd_order_trees.log)
    const unsigned gx = gridDim.x, total = gx * gridDim.y, lin = bx + gx * by;
    unsigned l2 = lin;
    if (k1.xcd_order < 0) {
      const unsigned qd = total >> 3, rm = total & 7u, xcd = lin & 7u, idx = lin >> 3;
      l2 = (xcd < rm ? xcd * (qd + 1u) : rm * (qd + 1u) + (xcd - rm) * qd) + idx;
    } else {
      // large launches: runs of G consecutive row blocks per XCD inside spans of 8 G blocks -- the eight XCDs stay within
      // 8 G rows of one another (one stream through every array, not eight), and only one block in G has its
      // neighbours' rows on another XCD
      const unsigned G = (unsigned)k1.xcd_order, span = 8u * G, full = total - total % span;
      if (lin < full) {
        const unsigned c = lin / span, w = lin - c * span;
        l2 = c * span + (w & 7u) * G + (w >> 3);
      }
    }
    by = l2 / gx;
    bx = l2 - by * gx;
  }
  const long pl = by;
  const PairedRows pr = paired_rows_of(g, bx, p.skipa != 0);
  const bool v1 = pr.second;
  const long o_row = pl * g.rplane + (long)pr.r0 * g.rpitch;
  if (K1 && pr.arr == 0) {
    if constexpr (K1) {
      static_assert((PL::n >> 2) <= 2 * NT, "one or two quads per lane and row");
      const long po = pl * g.rplane;
#pragma unroll
      for (int q = tid; q < (PL::n >> 2); q += NT)
        k1_two_rows<SK>(g, p, k1.V + po, k1.Vold + po, k1.eta0 + po, k1.eta1 + po, k1.eta0_out + po, k1.eta1_out + po,
                        k1.rho + po, pr.r0, v1, q, s);
      __syncthreads();
      fft_tile<NT, EMAX, false, SK, false>(s, plan, 1, make_fastdiv_dev1(), tid, LdsNatural{}, LdsNatural{});
      real2* o0 = SA + pl * g.cplane + (long)pr.r0 * g.cpitch;
      untangle_store<NT, SK, SL>(s, g.Wp, g.Wc, o0, o0 + (SL ? 8 : g.cpitch), v1, tid);
      LPC_STAMP_END();
    }
    return;
  }
  if (pr.arr == 0) {
    const real* ra = Rsp + o_row;
    const real* rb = ra + g.rpitch;
    auto two = [&](int i, int) { return make_real2(ra[i], v1 ? rb[i] : (real)0.); };
    fft_tile<NT, EMAX, false, SK, false, true>(s, plan, 1, make_fastdiv_dev1(), tid, two, LdsNatural{});
    real2* o0 = SA + pl * g.cplane + (long)pr.r0 * g.cpitch;
    untangle_store<NT, SK, SL>(s, g.Wp, g.Wc, o0, o0 + (SL ? 8 : g.cpitch), v1, tid);
    LPC_STAMP_END();
    return;
  }
  // two rows of a = mu1 X - xi' (xhalf_load / xhalf_apply): every 16-byte load of both rows is in flight before the
  // first statement that needs one, the pair of rows goes into the tile as z = a_r0 + i a_r1, then the transform runs
  // from LDS.  (The first form of this kernel formed `a` inside the source functor of the first FFT stage -- one load,
  // one wait, one store at a time behind four data-dependent branches per element; with two rows per block that chain
  // was the whole kernel: C1 0.272 -> 0.293 ms per 5 iterations, r04a/ab_pairing.log.)
  const int dpl = (int)(pl / g.DC) * g.C + (int)(pl % g.C);
  const int r1 = pr.r0 + 1, n4 = g.Wp >> 2;
  const bool v0 = pr.first;     // (false: the row above a window that starts on an odd row -- read like any row outside
                                // the window, neither stored nor transformed)
  const bool in0 = (pr.r0 >= g.sh) && (pr.r0 < g.sh + g.H), in1 = v1 && (r1 >= g.sh) && (r1 < g.sh + g.H);
  const real* y0 = Y + (long)dpl * g.uplane + (long)(pr.r0 - g.sh) * g.W;    // dereferenced only inside the window
  const real* y1 = y0 + g.W;
  const long o_row1 = o_row + g.rpitch;
  const bool y4 = ((g.sw | g.W) & 3) == 0;
  // (compile-time plans: Wp == PL::n.  One quad per lane when the row fits the workgroup -- no loop, no second set of
  // quads in flight: 88 -> VGPRs of the one-trip form, and with them the number of workgroups a CU holds; at C1 1620
  // workgroups are launched and only 1280 of the two-set form were resident, profiles/r05_notes.md section 5)
  constexpr int N4C = PL::n >> 2;
  if constexpr (N4C <= NT) {
    if (tid < N4C) {
      const int gc = 4 * tid;
      const XQuad c0 = xhalf_load(g, p, HV, HVold, xi, y0, o_row, in0, y4, tid);
      const XQuad c1 = xhalf_load(g, p, HV, HVold, xi, y1, v1 ? o_row1 : o_row, in1, y4, tid);
      real x0[4], a0[4], x1[4], a1[4];
      xhalf_apply(p, c0, x0, a0);
      xhalf_apply(p, c1, x1, a1);
      if (v0 && (!c0.skip || p.xi_store)) st4(xi + o_row + gc, make_real4(x0[0], x0[1], x0[2], x0[3]));
      if (v1 && (!c1.skip || p.xi_store)) st4(xi + o_row1 + gc, make_real4(x1[0], x1[1], x1[2], x1[3]));
#pragma unroll
      for (int i = 0; i < 4; ++i) s[lds_slot<SK>(gc + i)] = make_real2(v0 ? a0[i] : (real)0., v1 ? a1[i] : (real)0.);
    }
  } else {
    const int q0 = tid < n4 ? tid : 0;
    XQuad n0 = xhalf_load(g, p, HV, HVold, xi, y0, o_row, in0, y4, q0);
    XQuad n1 = xhalf_load(g, p, HV, HVold, xi, y1, v1 ? o_row1 : o_row, in1, y4, q0);
#pragma unroll 1
    for (int q = tid; q < n4; q += NT) {
      const int gc = 4 * q;
      const XQuad c0 = n0, c1 = n1;
      if (q + NT < n4) {
        n0 = xhalf_load(g, p, HV, HVold, xi, y0, o_row, in0, y4, q + NT);
        n1 = xhalf_load(g, p, HV, HVold, xi, y1, v1 ? o_row1 : o_row, in1, y4, q + NT);
      }
      real x0[4], a0[4], x1[4], a1[4];
      xhalf_apply(p, c0, x0, a0);
      xhalf_apply(p, c1, x1, a1);
      if (v0 && (!c0.skip || p.xi_store)) st4(xi + o_row + gc, make_real4(x0[0], x0[1], x0[2], x0[3]));
      if (v1 && (!c1.skip || p.xi_store)) st4(xi + o_row1 + gc, make_real4(x1[0], x1[1], x1[2], x1[3]));
#pragma unroll
      for (int i = 0; i < 4; ++i) s[lds_slot<SK>(gc + i)] = make_real2(v0 ? a0[i] : (real)0., v1 ? a1[i] : (real)0.);
    }
  }
  __syncthreads();
  fft_tile<NT, EMAX, false, SK, false>(s, plan, 1, make_fastdiv_dev1(), tid, LdsNatural{}, LdsNatural{});
  real2* o1 = SB + pl * g.cplane + (long)pr.r0 * g.cpitch;
  untangle_store<NT, SK, SL>(s, g.Wp, g.Wc, o1, o1 + (SL ? 8 : g.cpitch), v1, tid, v0);
  LPC_STAMP_END();
}

// ---- plug-and-play ADMM: the U-prox is an external denoiser (admm.py:126-133,235-243,266-275,300-311) ----------
// Explicit state (U and eta are image-shaped, Psi^T is the identity), plain streaming kernels around the caller's
// function; the reference's branch is reproduced as written, including what its conditional expression drops:
//   use_dual:  r_k = (mu3 W - rho) + mu2 U - eta                 (no data term)
//   else:      r_k = mu2 U + H^T (mu1 X - xi)                    (no W term)
template <int NT>
__global__ __launch_bounds__(NT) void k_pnp_input(real* LPC_RESTRICT out, const real* LPC_RESTRICT U,
                                                   const real* LPC_RESTRICT eta, real mu2, long n) {
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < n; e += (long)gridDim.x * NT) out[e] = U[e] + eta[e] / mu2;
}

// X, W of this iteration (kept for the dual updates) and the two inputs of the spectral step
template <int NT>
__global__ __launch_bounds__(NT) void k_pnp_pre(PlaneGeom g, AdmmScalars p, int dual, const real* LPC_RESTRICT V,
                                                 const real* LPC_RESTRICT HV, const real* LPC_RESTRICT xi,
                                                 const real* LPC_RESTRICT rho, const real* LPC_RESTRICT U,
                                                 const real* LPC_RESTRICT eta, const real* LPC_RESTRICT Y,
                                                 real* LPC_RESTRICT X, real* LPC_RESTRICT W, real* LPC_RESTRICT Rsp,
                                                 real* LPC_RESTRICT Aout) {
  const long n = (long)g.Hp * g.Wp;
  const long pl = blockIdx.y;
  const int dpl = (int)(pl / g.DC) * g.C + (int)(pl % g.C);
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < n; e += (long)gridDim.x * NT) {
    const int r = (int)(e / g.Wp), c = (int)(e - (long)r * g.Wp);
    const long o = pl * g.rplane + (long)r * g.rpitch + c;
    const bool inside = (r >= g.sh) && (r < g.sh + g.H) && (c >= g.sw) && (c < g.sw + g.W);
    const real yv = inside ? Y[(long)dpl * g.uplane + (long)(r - g.sh) * g.W + (c - g.sw)] : (real)0.;
    const real x = (inside ? p.m_in : p.m_out) * (xi[o] + p.mu1 * HV[o] + yv);   // admm.py:252-254
    const real w = rmax(div_by(rho[o], p.mu3, p.r_mu3) + V[o], (real)0.);                        // admm.py:256-262
    X[o] = x;
    W[o] = w;
    if (dual == 2) {   // caller-supplied prior (admm.py:104-120, 277-281): U holds Psi^T(mu2 U - eta), computed by the caller
      Rsp[o] = (p.mu3 * w - rho[o]) + U[o];
      Aout[o] = p.mu1 * x - xi[o];
    } else if (dual) {
      Rsp[o] = (p.mu3 * w - rho[o]) + p.mu2 * U[o] - eta[o];
      Aout[o] = (real)0.;
    } else {
      Rsp[o] = p.mu2 * U[o];
      Aout[o] = p.mu1 * x - xi[o];
    }
  }
}

// dual updates with the new image estimate (admm.py:296-311)
template <int NT>
__global__ __launch_bounds__(NT) void k_pnp_post(PlaneGeom g, AdmmScalars p, int dual, const real* LPC_RESTRICT Vn,
                                                  const real* LPC_RESTRICT HVn, const real* LPC_RESTRICT X,
                                                  const real* LPC_RESTRICT W, const real* LPC_RESTRICT U,
                                                  real* LPC_RESTRICT xi, real* LPC_RESTRICT eta,
                                                  real* LPC_RESTRICT rho) {
  const long n = (long)g.Hp * g.Wp;
  const long pl = blockIdx.y;
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < n; e += (long)gridDim.x * NT) {
    const int r = (int)(e / g.Wp), c = (int)(e - (long)r * g.Wp);
    const long o = pl * g.rplane + (long)r * g.rpitch + c;
    xi[o] = xi[o] + p.mu1 * (HVn[o] - X[o]);
    if (dual) eta[o] = eta[o] + p.mu2 * (Vn[o] - U[o]);
    rho[o] = rho[o] + p.mu3 * (Vn[o] - W[o]);
  }
}

// materialise U, W and the flushed duals for inspection (tests / get_state); no state change
template <int NT>
__global__ __launch_bounds__(NT) void k_admm_flush(PlaneGeom g, AdmmScalars p,
                                                    const real* LPC_RESTRICT V,
                                                    const real* LPC_RESTRICT Vold,
                                                    const real* LPC_RESTRICT HV,
                                                    const real* LPC_RESTRICT HVold,
                                                    const real* LPC_RESTRICT Y,
                                                    const real* LPC_RESTRICT xi,
                                                    const real* LPC_RESTRICT eta0,
                                                    const real* LPC_RESTRICT eta1,
                                                    const real* LPC_RESTRICT rho, real* LPC_RESTRICT out0,
                                                    real* LPC_RESTRICT out1, int which0, int which1) {
  // quantities: 0 xi', 1 eta0', 2 eta1', 3 rho', 4 U0, 5 U1, 6 W, 7 X; out0 receives `which0`, out1 (if not null)
  // `which1` -- the caller asks for what it reads out, the scratch is two padded arrays the loop leaves idle
  const long n = (long)g.Hp * g.Wp;
  const long pl = blockIdx.y;
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < n; e += (long)gridDim.x * NT) {
    const int r = (int)(e / g.Wp), c = (int)(e - (long)r * g.Wp);
    const long o = pl * g.rplane + (long)r * g.rpitch + c;
    const long ou = pl * g.rplane + (long)wrap_add(r, -1, g.Hp) * g.rpitch + c;
    const long ol = pl * g.rplane + (long)r * g.rpitch + wrap_add(c, -1, g.Wp);
    real xiv = xi[o], e0 = eta0[o], e1 = eta1[o], rh = rho[o];
    real u0 = (real)0., u1 = (real)0., w = (real)0., x = (real)0.;
    if (!p.first) {
      const bool inside = (r >= g.sh) && (r < g.sh + g.H) && (c >= g.sw) && (c < g.sw + g.W);
      const int dpl = (int)(pl / g.DC) * g.C + (int)(pl % g.C);
      const real yv = inside ? Y[(long)dpl * g.uplane + (long)(r - g.sh) * g.W + (c - g.sw)] : (real)0.;
      x = (inside ? p.m_in_p : p.m_out_p) * (xiv + p.mu1p * HVold[o] + yv);   // X of the last iteration
      const real oc = Vold[o], vc = V[o];
      u0 = soft_thresh_dev((Vold[ou] - oc) + div_by(e0, p.mu2p, p.r_mu2p), p.thrp);
      u1 = soft_thresh_dev((Vold[ol] - oc) + div_by(e1, p.mu2p, p.r_mu2p), p.thrp);
      w = rmax(div_by(rh, p.mu3p, p.r_mu3p) + w_sees(oc, p.clamp_old, inside), (real)0.);
      xiv = xiv + p.mu1p * (HV[o] - x);
      e0 = e0 + p.mu2p * ((V[ou] - vc) - u0);
      e1 = e1 + p.mu2p * ((V[ol] - vc) - u1);
      rh = rh + p.mu3p * (vc - w);
    }
    const real val[8] = {xiv, e0, e1, rh, u0, u1, w, x};
    out0[o] = val[which0];
    if (out1) out1[o] = val[which1];
  }
}

// ========================================================= layout / setup kernels ==
// channels-last (n, rows, cols, C) <-> planar (n*C planes)[rows][pitch]
// Csrc: channels of the source, C or 1 (a one-channel source is broadcast over the C planes, the way the
// reference's `vpad[...] = v` / `rfft2(x) * H` broadcast a grayscale input against an RGB PSF)
template <int NT>
__global__ __launch_bounds__(NT) void k_hwc_to_planar(const real* LPC_RESTRICT src, real* LPC_RESTRICT dst,
                                                       int rows, int cols, int C, int pitch, long dplane, int Csrc) {
  const long n = (long)rows * cols * C;
  const long nsrc = (long)rows * cols * Csrc;
  const long img = blockIdx.y;
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < n; e += (long)gridDim.x * NT) {
    const int c = (int)(e % C);
    const long rc = e / C;
    const int col = (int)(rc % cols);
    const int row = (int)(rc / cols);
    dst[(img * C + c) * dplane + (long)row * pitch + col] = src[img * nsrc + (Csrc == C ? e : rc)];
  }
}

// planar -> channels-last with optional crop window and clamp (>= 0).  clamp_src: also write
// the clamped value back into the planar source (the reference's ADMM._form_image clamps
// its state IN PLACE, admm.py:331-338).
template <int NT>
__global__ __launch_bounds__(NT) void k_planar_to_hwc(real* LPC_RESTRICT src, real* LPC_RESTRICT dst,
                                                       int rows, int cols, int C, int pitch, long splane,
                                                       int row0, int col0, int clamp, int clamp_src, int wr0, int wr1,
                                                       int wc0, int wc1) {
  const long n = (long)rows * cols * C;
  const long img = blockIdx.y;
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < n; e += (long)gridDim.x * NT) {
    const int c = (int)(e % C);
    const long rc = e / C;
    const int col = (int)(rc % cols);
    const int row = (int)(rc / cols);
    const long so = (img * C + c) * splane + (long)(row0 + row) * pitch + (col0 + col);
    real v = src[so];
    if (clamp && v < (real)0. && row >= wr0 && row < wr1 && col >= wc0 && col < wc1) {   // [wr0,wr1) x [wc0,wc1): where
      v = (real)0.;
      if (clamp_src) src[so] = (real)0.;
    }
    dst[img * n + e] = v;
  }
}

// the in-place clamp of ADMM._form_image on the sensor window (plug-and-play ADMM keeps its image estimate in one
// explicit array; also applied to the stored initial estimate, see lpc_form_image)
template <int NT>
__global__ __launch_bounds__(NT) void k_clamp_window_inplace(PlaneGeom g, real* x) {
  const long n = (long)g.H * g.W;
  const long pl = blockIdx.y;
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < n; e += (long)gridDim.x * NT) {
    const int r = (int)(e / g.W), c = (int)(e - (long)r * g.W);
    const long o = pl * g.rplane + (long)(g.sh + r) * g.rpitch + (g.sw + c);
    if (x[o] < (real)0.) x[o] = (real)0.;
  }
}

// |G| of the TV gram spectrum (admm.py:188 takes torch.abs of it), one plane
template <int NT>
__global__ __launch_bounds__(NT) void k_abs_complex(const real2* LPC_RESTRICT Gs, real* LPC_RESTRICT out, long n) {
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < n; e += (long)gridDim.x * NT) {
    const real2 gg = Gs[e];
    out[e] = rsqrt_of(gg.x * gg.x + gg.y * gg.y);
  }
}

// a real spectrum plane in natural row order [Hp][Wc] -> the engine's stored (four-step) row order [Hp][cpitch]:
// stored row p = k1*N2 + k2 holds frequency k1 + N1*k2
template <int NT>
__global__ __launch_bounds__(NT) void k_permute_spectrum_rows(const real* LPC_RESTRICT nat, real* LPC_RESTRICT out,
                                                               int Hp, int Wc, int cpitch, int N1, int N2) {
  const long n = (long)Hp * Wc;
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < n; e += (long)gridDim.x * NT) {
    const int prow = (int)(e / Wc), c = (int)(e - (long)prow * Wc);
    const int k = (prow / N2) + N1 * (prow % N2);
    out[(long)prow * cpitch + c] = nat[(long)k * Wc + c];
  }
}

// spectrum planes [Hp][cpitch] -> the pair-line layout (spec_col): the copies of H and |G| an 8-column fused middle reads
template <int NT, class Tp>
__global__ __launch_bounds__(NT) void k_to_pair_lines(const Tp* LPC_RESTRICT src, Tp* LPC_RESTRICT dst, int Hp, int cpitch,
                                                       long cplane) {
  const long n = (long)Hp * cpitch;
  const long pl = blockIdx.y;
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < n; e += (long)gridDim.x * NT) {
    const int i = (int)(e / cpitch), j = (int)(e - (long)i * cpitch);
    dst[pl * cplane + (long)(i >> 1) * 2 * cpitch + spec_col<1>(j) + (i & 1) * 8] = src[pl * cplane + e];
  }
}

// return_fft=True (rfft_convolve.py:148-150,193-195): the engine's work spectrum (planar, permuted row order: stored row
// p = k1*N2 + k2 holds frequency k1 + N1*k2) times H or conj(H) -> the reference's layout (img, Hp, Wc, C) complex,
// natural frequency order, channels last.  grid = (blocks over Hp*Wc*C, images).
template <int NT>
__global__ __launch_bounds__(NT) void k_spectrum_mul_to_hwc(PlaneGeom g, const real2* LPC_RESTRICT S,
                                                             const real2* LPC_RESTRICT Hs, int conjH,
                                                             real2* LPC_RESTRICT out, int N1, int N2) {
  const long n = (long)g.Hp * g.Wc * g.C;
  const long img = blockIdx.y;
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < n; e += (long)gridDim.x * NT) {
    const int c = (int)(e % g.C);
    const long kk = e / g.C;
    const int kc = (int)(kk % g.Wc), kr = (int)(kk / g.Wc);
    const int p = (kr % N1) * N2 + kr / N1;
    const long q = img * g.C + c;
    const long off = (long)p * g.cpitch + kc;
    const real2 v = S[q * g.cplane + off], h = Hs[(q % g.DC) * g.cplane + off];
    out[img * n + e] = conjH ? cmul_conj(v, h) : cmul(v, h);
  }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_scale_complex(real2* LPC_RESTRICT S, long n, real sc) {
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < n; e += (long)gridDim.x * NT) {
    S[e] = cscale(S[e], sc);
  }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_fill(real* LPC_RESTRICT p, long n, real v) {
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < n; e += (long)gridDim.x * NT) p[e] = v;
}
