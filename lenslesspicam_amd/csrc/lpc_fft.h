// lpc_fft.h -- batched mixed-radix (2,3,4,5,8) Stockham FFT held entirely in LDS.
//
// Design (MI355X-first, not a rocFFT/cuFFT translation):
//  * one workgroup owns a tile of BT transforms of length n laid out in LDS as
//    s[i*BT + b] (element i of transform b) -- for column passes b runs over adjacent
//    image columns, so consecutive lanes touch consecutive float2 (ds_read/write_b64,
//    conflict-free) and the global loads that fill the tile are 128-byte segments;
//  * every stage is "all lanes read their butterflies into VGPRs -> barrier -> all lanes
//    write" so the tile is transformed in place (no ping-pong buffer => half the LDS,
//    twice the workgroups per CU);
//  * the autosort (Stockham) indexing leaves each 1-D transform in natural order, which
//    the real<->half-spectrum untangling of the row passes needs;
//  * twiddles come from one double-precision-generated table per length (exp(-2 pi i q/n)),
//    L1/L2 resident.
#pragma once
#include "lpc_rt.h"

#define LPC_MAX_STAGES 12

struct Fft1dPlan {
  int n;                       // transform length (5-smooth)
  int nst;                     // number of stages
  int radix[LPC_MAX_STAGES];   // radix of stage s
  int ns[LPC_MAX_STAGES];      // product of the radices of the stages before s
  int twstep[LPC_MAX_STAGES];  // n / (ns*radix)
  FastDiv nsdiv[LPC_MAX_STAGES];
  const float2* tw;            // device table, n entries: exp(-2 pi i q / n)
};

// multiply by -i (forward) / +i (inverse)
template <bool INV>
static __device__ __forceinline__ float2 rot90(float2 a) {
  return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

template <int R, bool INV>
struct Dft;

template <bool INV>
struct Dft<2, INV> {
  static __device__ __forceinline__ void run(float2* v) {
    float2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  }
};

template <bool INV>
struct Dft<4, INV> {
  static __device__ __forceinline__ void run(float2* v) {
    float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
    float2 t2 = cadd(v[1], v[3]), t3 = rot90<INV>(csub(v[1], v[3]));
    v[0] = cadd(t0, t2);
    v[1] = cadd(t1, t3);
    v[2] = csub(t0, t2);
    v[3] = csub(t1, t3);
  }
};

template <bool INV>
struct Dft<3, INV> {
  static __device__ __forceinline__ void run(float2* v) {
    const float S = 0.86602540378443864676f;  // sin(2 pi / 3)
    float2 t = cadd(v[1], v[2]);
    float2 m = make_float2(v[0].x - 0.5f * t.x, v[0].y - 0.5f * t.y);
    float2 d = cscale(csub(v[1], v[2]), S);
    v[0] = cadd(v[0], t);
    // forward: X1 = m - i d, X2 = m + i d ; inverse swaps them
    float2 p = make_float2(m.x + d.y, m.y - d.x);
    float2 q = make_float2(m.x - d.y, m.y + d.x);
    v[1] = INV ? q : p;
    v[2] = INV ? p : q;
  }
};

template <bool INV>
struct Dft<5, INV> {
  static __device__ __forceinline__ void run(float2* v) {
    const float C1 = 0.30901699437494742410f;   // cos(2 pi/5)
    const float C2 = -0.80901699437494742410f;  // cos(4 pi/5)
    const float S1 = 0.95105651629515357212f;   // sin(2 pi/5)
    const float S2 = 0.58778525229247312917f;   // sin(4 pi/5)
    float2 a0 = v[0];
    float2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
    float2 t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    float2 m1 = make_float2(a0.x + C1 * t1.x + C2 * t2.x, a0.y + C1 * t1.y + C2 * t2.y);
    float2 m2 = make_float2(a0.x + C2 * t1.x + C1 * t2.x, a0.y + C2 * t1.y + C1 * t2.y);
    float2 n1 = make_float2(S1 * t3.x + S2 * t4.x, S1 * t3.y + S2 * t4.y);
    float2 n2 = make_float2(S2 * t3.x - S1 * t4.x, S2 * t3.y - S1 * t4.y);
    v[0] = make_float2(a0.x + t1.x + t2.x, a0.y + t1.y + t2.y);
    // forward: X1 = m1 - i n1, X4 = m1 + i n1, X2 = m2 - i n2, X3 = m2 + i n2
    float2 x1 = make_float2(m1.x + n1.y, m1.y - n1.x);
    float2 x4 = make_float2(m1.x - n1.y, m1.y + n1.x);
    float2 x2 = make_float2(m2.x + n2.y, m2.y - n2.x);
    float2 x3 = make_float2(m2.x - n2.y, m2.y + n2.x);
    v[1] = INV ? x4 : x1;
    v[4] = INV ? x1 : x4;
    v[2] = INV ? x3 : x2;
    v[3] = INV ? x2 : x3;
  }
};

template <bool INV>
struct Dft<8, INV> {
  static __device__ __forceinline__ void run(float2* v) {
    const float C = 0.70710678118654752440f;
    float2 b0 = cadd(v[0], v[4]), b4 = csub(v[0], v[4]);
    float2 b1 = cadd(v[1], v[5]), b5 = csub(v[1], v[5]);
    float2 b2 = cadd(v[2], v[6]), b6 = csub(v[2], v[6]);
    float2 b3 = cadd(v[3], v[7]), b7 = csub(v[3], v[7]);
    // b5 *= w8, b6 *= w8^2, b7 *= w8^3   (w8 = exp(-+ i pi/4))
    b5 = INV ? make_float2(C * (b5.x - b5.y), C * (b5.x + b5.y))
             : make_float2(C * (b5.x + b5.y), C * (b5.y - b5.x));
    b6 = rot90<INV>(b6);
    b7 = INV ? make_float2(-C * (b7.x + b7.y), C * (b7.x - b7.y))
             : make_float2(C * (b7.y - b7.x), -C * (b7.x + b7.y));
    float2 e[4] = {b0, b1, b2, b3};
    float2 o[4] = {b4, b5, b6, b7};
    Dft<4, INV>::run(e);
    Dft<4, INV>::run(o);
    v[0] = e[0]; v[1] = o[0]; v[2] = e[1]; v[3] = o[1];
    v[4] = e[2]; v[5] = o[2]; v[6] = e[3]; v[7] = o[3];
  }
};

// One Stockham stage over a tile of BT transforms.  Ends with a barrier.
template <int R, int NT, int EMAX, bool INV>
static __device__ __forceinline__ void fft_stage(float2* s, int n, int BT, FastDiv btdiv, int ns,
                                                  FastDiv nsdiv, int twstep,
                                                  const float2* LPC_RESTRICT tw, int tid) {
  constexpr int MAXB = (EMAX + R - 1) / R;
  const int nb = n / R;
  const int nwork = nb * BT;
  const int istride = nb * BT;   // LDS distance between the R inputs of one butterfly
  const int ostride = ns * BT;   // LDS distance between its R outputs
  float2 v[MAXB][R];
  int obase[MAXB];
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    const int w = tid + b * NT;
    obase[b] = -1;
    if (w < nwork) {
      const int j = (int)fd_div((unsigned)w, btdiv);
      const int c = w - j * BT;
      const int jq = (int)fd_div((unsigned)j, nsdiv);
      const int k = j - jq * ns;
#pragma unroll
      for (int m = 0; m < R; ++m) v[b][m] = s[w + m * istride];
      if (ns > 1) {
        const int q1 = k * twstep;
#pragma unroll
        for (int m = 1; m < R; ++m) {
          float2 t = tw[q1 * m];
          v[b][m] = INV ? cmul_conj(v[b][m], t) : cmul(v[b][m], t);
        }
      }
      Dft<R, INV>::run(v[b]);
      obase[b] = (jq * ns * R + k) * BT + c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    if (obase[b] >= 0) {
#pragma unroll
      for (int m = 0; m < R; ++m) s[obase[b] + m * ostride] = v[b][m];
    }
  }
  __syncthreads();
}

// In-place FFT of BT interleaved transforms of length plan.n held in LDS.
// Precondition: the tile is fully written and a barrier has been passed.
// Postcondition: result in natural order, barrier passed.  Unnormalised in both directions.
template <int NT, int EMAX, bool INV>
static __device__ __forceinline__ void lds_fft(float2* s, const Fft1dPlan& p, int BT, FastDiv btdiv,
                                                int tid) {
  for (int st = 0; st < p.nst; ++st) {
    const int ns = p.ns[st];
    const FastDiv nd = p.nsdiv[st];
    const int ts = p.twstep[st];
    switch (p.radix[st]) {
      case 8: fft_stage<8, NT, EMAX, INV>(s, p.n, BT, btdiv, ns, nd, ts, p.tw, tid); break;
      case 4: fft_stage<4, NT, EMAX, INV>(s, p.n, BT, btdiv, ns, nd, ts, p.tw, tid); break;
      case 2: fft_stage<2, NT, EMAX, INV>(s, p.n, BT, btdiv, ns, nd, ts, p.tw, tid); break;
      case 3: fft_stage<3, NT, EMAX, INV>(s, p.n, BT, btdiv, ns, nd, ts, p.tw, tid); break;
      default: fft_stage<5, NT, EMAX, INV>(s, p.n, BT, btdiv, ns, nd, ts, p.tw, tid); break;
    }
  }
}
