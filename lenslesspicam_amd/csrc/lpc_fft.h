// lpc_fft.h -- batched mixed-radix (8,6,5,4,3,2; 16 optional) Stockham FFT for one workgroup.
//
// Design (MI355X-first, not a rocFFT/cuFFT translation):
//  * one workgroup owns a tile of BT transforms of length n.  Between stages the tile lives in
//    LDS as s[i*BT + b] (element i of transform b) -- for column passes b runs over adjacent
//    image columns, so consecutive lanes touch consecutive real2 (ds_read/write_b64,
//    conflict-free) and the global accesses that fill / drain the tile are 128-byte segments;
//  * a tile is filled from a caller-supplied SOURCE functor (global loads, padding, residuals)
//    with all of a thread's loads issued back-to-back into VGPRs before the first LDS write, and
//    drained into a SINK functor (global stores, crop/shift, fused solver updates);
//  * every stage is "all lanes read their butterflies into VGPRs -> barrier -> all lanes
//    write", i.e. in place: no ping-pong buffer, half the LDS, twice the workgroups per CU;
//  * measured on MI355X (profiles/r01b_notes.md): more points per thread beats more waves per
//    workgroup, but radix 16 costs ~50 VGPRs in every kernel (the radix switch is allocated for
//    its fattest arm) and loses more occupancy than it saves in barriers -> radix <= 8 by default;
//  * the autosort (Stockham) indexing leaves each 1-D transform in natural order, which the
//    real<->half-spectrum untangling of the row passes needs;
//  * twiddles come from one double-precision-generated table per length (exp(-2 pi i q/n)); radix
//    8 / 16 butterflies load w^1, w^2, w^4 (, w^8) and form the other powers by <= 2 products.
#pragma once
#include "lpc_rt.h"
#include <type_traits>

#define LPC_MAX_STAGES 12

struct Fft1dPlan {
  int n;                       // transform length (5-smooth)
  int nst;                     // number of stages
  int radix[LPC_MAX_STAGES];   // radix of stage s
  int ns[LPC_MAX_STAGES];      // product of the radices of the stages before s
  int twstep[LPC_MAX_STAGES];  // n / (ns*radix)
  FastDiv nsdiv[LPC_MAX_STAGES];
  const real2* tw;            // device table, n entries: exp(-2 pi i q / n)
  int skew_ok;                 // row mode: the i + i/8 LDS skew stays affine in every stage
};

// multiply by -i (forward) / +i (inverse)
template <bool INV>
static __device__ __forceinline__ real2 rot90(real2 a) {
  return INV ? cmul_pi(a) : cmul_mi(a);
}
// multiply by exp(-+ 2 pi i q / 16) for the constants a radix-16 butterfly needs
template <bool INV, int Q>
static __device__ __forceinline__ real2 mul_w16(real2 a) {
  constexpr real C8 = (real)0.70710678118654752440;   // cos(pi/4)
  constexpr real C1 = (real)0.92387953251128675613;   // cos(pi/8)
  constexpr real S1 = (real)0.38268343236508977173;   // sin(pi/8)
  constexpr real wr = Q == 1 ? C1 : Q == 2 ? C8 : Q == 3 ? S1 : Q == 6 ? -C8 : -C1;   // Q in {1,2,3,6,9}
  constexpr real wf = Q == 1 ? -S1 : Q == 2 ? -C8 : Q == 3 ? -C1 : Q == 6 ? -C8 : S1;  // forward imag part
  constexpr real wi = INV ? -wf : wf;
  return cmul(a, make_real2(wr, wi));
}

template <int R, bool INV>
struct Dft;

template <bool INV>
struct Dft<2, INV> {
  static __device__ __forceinline__ void run(real2* v) {
    real2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  }
};

template <bool INV>
struct Dft<4, INV> {
  static __device__ __forceinline__ void run(real2* v) {
    real2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
    real2 t2 = cadd(v[1], v[3]), t3 = rot90<INV>(csub(v[1], v[3]));
    v[0] = cadd(t0, t2);
    v[1] = cadd(t1, t3);
    v[2] = csub(t0, t2);
    v[3] = csub(t1, t3);
  }
};

template <bool INV>
struct Dft<3, INV> {
  static __device__ __forceinline__ void run(real2* v) {
    const real S = (real)0.86602540378443864676;  // sin(2 pi / 3)
    real2 t = cadd(v[1], v[2]);
    real2 m = caxpy(t, -(real)0.5, v[0]);
    real2 d = cmul_mi(cscale(csub(v[1], v[2]), S));   // -i d
    v[0] = cadd(v[0], t);
    // forward: X1 = m - i d, X2 = m + i d ; inverse swaps them
    real2 p = cadd(m, d);
    real2 q = csub(m, d);
    v[1] = INV ? q : p;
    v[2] = INV ? p : q;
  }
};

template <bool INV>
struct Dft<5, INV> {
  static __device__ __forceinline__ void run(real2* v) {
    const real C1 = (real)0.30901699437494742410;   // cos(2 pi/5)
    const real C2 = -(real)0.80901699437494742410;  // cos(4 pi/5)
    const real S1 = (real)0.95105651629515357212;   // sin(2 pi/5)
    const real S2 = (real)0.58778525229247312917;   // sin(4 pi/5)
    real2 a0 = v[0];
    real2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
    real2 t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    real2 m1 = caxpy(t2, C2, caxpy(t1, C1, a0));
    real2 m2 = caxpy(t2, C1, caxpy(t1, C2, a0));
    real2 n1 = cmul_mi(caxpy(t4, S2, cscale(t3, S1)));    // -i n1
    real2 n2 = cmul_mi(caxpy(t4, -S1, cscale(t3, S2)));   // -i n2
    v[0] = cadd(cadd(a0, t1), t2);
    // forward: X1 = m1 - i n1, X4 = m1 + i n1, X2 = m2 - i n2, X3 = m2 + i n2
    real2 x1 = cadd(m1, n1);
    real2 x4 = csub(m1, n1);
    real2 x2 = cadd(m2, n2);
    real2 x3 = csub(m2, n2);
    v[1] = INV ? x4 : x1;
    v[4] = INV ? x1 : x4;
    v[2] = INV ? x3 : x2;
    v[3] = INV ? x2 : x3;
  }
};

template <bool INV>
struct Dft<6, INV> {  // 6 = 2 x 3 (decimation in time over the even / odd inputs)
  static __device__ __forceinline__ void run(real2* v) {
    const real S = (real)0.86602540378443864676;
    real2 e[3] = {v[0], v[2], v[4]};
    real2 o[3] = {v[1], v[3], v[5]};
    Dft<3, INV>::run(e);
    Dft<3, INV>::run(o);
    // w6^1 = (1/2, -+ S), w6^2 = (-1/2, -+ S)
    const real si = INV ? S : -S;
    real2 o1 = caxpy(cmul_pi(o[1]), si, cscale(o[1], (real)0.5));    // o * (1/2 + i si)
    real2 o2 = caxpy(cmul_pi(o[2]), si, cscale(o[2], -(real)0.5));   // o * (-1/2 + i si)
    v[0] = cadd(e[0], o[0]); v[3] = csub(e[0], o[0]);
    v[1] = cadd(e[1], o1);   v[4] = csub(e[1], o1);
    v[2] = cadd(e[2], o2);   v[5] = csub(e[2], o2);
  }
};

template <bool INV>
struct Dft<8, INV> {
  static __device__ __forceinline__ void run(real2* v) {
    const real C = (real)0.70710678118654752440;
    real2 b0 = cadd(v[0], v[4]), b4 = csub(v[0], v[4]);
    real2 b1 = cadd(v[1], v[5]), b5 = csub(v[1], v[5]);
    real2 b2 = cadd(v[2], v[6]), b6 = csub(v[2], v[6]);
    real2 b3 = cadd(v[3], v[7]), b7 = csub(v[3], v[7]);
    // b5 *= w8, b6 *= w8^2, b7 *= w8^3   (w8 = exp(-+ i pi/4))
    b5 = cscale(cadd(b5, rot90<INV>(b5)), C);
    b6 = rot90<INV>(b6);
    b7 = cscale(csub(rot90<INV>(b7), b7), C);
    real2 e[4] = {b0, b1, b2, b3};
    real2 o[4] = {b4, b5, b6, b7};
    Dft<4, INV>::run(e);
    Dft<4, INV>::run(o);
    v[0] = e[0]; v[1] = o[0]; v[2] = e[1]; v[3] = o[1];
    v[4] = e[2]; v[5] = o[2]; v[6] = e[3]; v[7] = o[3];
  }
};

template <bool INV>
struct Dft<16, INV> {  // 16 = 4 x 4: X[k1 + 4 k2] = sum_n2 w16^(n2 k1) [sum_n1 x[4 n1 + n2] w4^(n1 k1)] w4^(n2 k2)
  static __device__ __forceinline__ void run(real2* v) {
    real2 y[4][4];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
      real2 t[4] = {v[n2], v[4 + n2], v[8 + n2], v[12 + n2]};
      Dft<4, INV>::run(t);
#pragma unroll
      for (int k1 = 0; k1 < 4; ++k1) y[n2][k1] = t[k1];
    }
    // twiddles w16^(n2*k1)
    y[1][1] = mul_w16<INV, 1>(y[1][1]); y[1][2] = mul_w16<INV, 2>(y[1][2]); y[1][3] = mul_w16<INV, 3>(y[1][3]);
    y[2][1] = mul_w16<INV, 2>(y[2][1]); y[2][2] = rot90<INV>(y[2][2]);      y[2][3] = mul_w16<INV, 6>(y[2][3]);
    y[3][1] = mul_w16<INV, 3>(y[3][1]); y[3][2] = mul_w16<INV, 6>(y[3][2]); y[3][3] = mul_w16<INV, 9>(y[3][3]);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
      real2 t[4] = {y[0][k1], y[1][k1], y[2][k1], y[3][k1]};
      Dft<4, INV>::run(t);
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) v[k1 + 4 * k2] = t[k2];
    }
  }
};


// ---- fat butterflies for the 540-point columns of DiffuserCam-sized frames: 540 = 30 x 18 in TWO stages ----------
// A radix-30 (= 6 x 5) and a radix-18 (= 6 x 3) butterfly held in registers, each a two-factor Cooley-Tukey with
// compile-time inner twiddles: the 540-point transform makes ONE trip through LDS between its two stages instead of
// three (plan 6.6.5.3), with half the barriers.  Natural order in, natural order out, both directions.
struct WPair { real re, im; };
template <int N> struct WTab;
template <> struct WTab<30> { static constexpr WPair w[30] = {{(real)1.00000000000000000000e+00, (real)-0.00000000000000000000e+00}, {(real)9.78147600733805688833e-01, (real)-2.07911690817759314820e-01}, {(real)9.13545457642600866599e-01, (real)-4.06736643075800152758e-01}, {(real)8.09016994374947451263e-01, (real)-5.87785252292473137103e-01}, {(real)6.69130606358858237570e-01, (real)-7.43144825477394133095e-01}, {(real)5.00000000000000111022e-01, (real)-8.66025403784438596588e-01}, {(real)3.09016994374947451263e-01, (real)-9.51056516295153531182e-01}, {(real)1.04528463267653456970e-01, (real)-9.94521895368273289861e-01}, {(real)-1.04528463267653332069e-01, (real)-9.94521895368273400884e-01}, {(real)-3.09016994374947340241e-01, (real)-9.51056516295153642204e-01}, {(real)-4.99999999999999777955e-01, (real)-8.66025403784438707611e-01}, {(real)-6.69130606358857904503e-01, (real)-7.43144825477394466162e-01}, {(real)-8.09016994374947340241e-01, (real)-5.87785252292473248126e-01}, {(real)-9.13545457642600977621e-01, (real)-4.06736643075800041736e-01}, {(real)-9.78147600733805688833e-01, (real)-2.07911690817759314820e-01}, {(real)-1.00000000000000000000e+00, (real)-5.66553889764797961539e-16}, {(real)-9.78147600733805688833e-01, (real)2.07911690817759065020e-01}, {(real)-9.13545457642600866599e-01, (real)4.06736643075800208269e-01}, {(real)-8.09016994374947562285e-01, (real)5.87785252292473026081e-01}, {(real)-6.69130606358858459615e-01, (real)7.43144825477394022073e-01}, {(real)-5.00000000000000444089e-01, (real)8.66025403784438374544e-01}, {(real)-3.09016994374947562285e-01, (real)9.51056516295153531182e-01}, {(real)-1.04528463267654234126e-01, (real)9.94521895368273289861e-01}, {(real)1.04528463267652985125e-01, (real)9.94521895368273400884e-01}, {(real)3.09016994374947229218e-01, (real)9.51056516295153642204e-01}, {(real)5.00000000000000111022e-01, (real)8.66025403784438596588e-01}, {(real)6.69130606358858459615e-01, (real)7.43144825477394022073e-01}, {(real)8.09016994374947340241e-01, (real)5.87785252292473359148e-01}, {(real)9.13545457642600977621e-01, (real)4.06736643075800152758e-01}, {(real)9.78147600733805688833e-01, (real)2.07911690817758981753e-01}}; };
template <> struct WTab<18> { static constexpr WPair w[18] = {{(real)1.00000000000000000000e+00, (real)-0.00000000000000000000e+00}, {(real)9.39692620785908427905e-01, (real)-3.42020143325668712908e-01}, {(real)7.66044443118978013452e-01, (real)-6.42787609686539251896e-01}, {(real)5.00000000000000111022e-01, (real)-8.66025403784438596588e-01}, {(real)1.73648177666930414453e-01, (real)-9.84807753012208020316e-01}, {(real)-1.73648177666930303431e-01, (real)-9.84807753012208020316e-01}, {(real)-4.99999999999999777955e-01, (real)-8.66025403784438707611e-01}, {(real)-7.66044443118977902429e-01, (real)-6.42787609686539473941e-01}, {(real)-9.39692620785908316883e-01, (real)-3.42020143325668879442e-01}, {(real)-1.00000000000000000000e+00, (real)-1.22464679914735320717e-16}, {(real)-9.39692620785908427905e-01, (real)3.42020143325668657397e-01}, {(real)-7.66044443118978346519e-01, (real)6.42787609686538918830e-01}, {(real)-5.00000000000000444089e-01, (real)8.66025403784438374544e-01}, {(real)-1.73648177666930331187e-01, (real)9.84807753012208020316e-01}, {(real)1.73648177666929970364e-01, (real)9.84807753012208131338e-01}, {(real)4.99999999999999333866e-01, (real)8.66025403784439040678e-01}, {(real)7.66044443118977791407e-01, (real)6.42787609686539584963e-01}, {(real)9.39692620785908427905e-01, (real)3.42020143325668601886e-01}}; };

template <> struct WTab<9> { static constexpr WPair w[9] = {{(real)1.00000000000000000000e+00, (real)-0.00000000000000000000e+00}, {(real)7.66044443118978013452e-01, (real)-6.42787609686539251896e-01}, {(real)1.73648177666930414453e-01, (real)-9.84807753012208020316e-01}, {(real)-4.99999999999999777955e-01, (real)-8.66025403784438707611e-01}, {(real)-9.39692620785908316883e-01, (real)-3.42020143325668879442e-01}, {(real)-9.39692620785908427905e-01, (real)3.42020143325668657397e-01}, {(real)-5.00000000000000444089e-01, (real)8.66025403784438374544e-01}, {(real)1.73648177666929970364e-01, (real)9.84807753012208131338e-01}, {(real)7.66044443118977791407e-01, (real)6.42787609686539584963e-01}}; };
template <> struct WTab<10> { static constexpr WPair w[10] = {{(real)1.00000000000000000000e+00, (real)-0.00000000000000000000e+00}, {(real)8.09016994374947451263e-01, (real)-5.87785252292473137103e-01}, {(real)3.09016994374947451263e-01, (real)-9.51056516295153531182e-01}, {(real)-3.09016994374947340241e-01, (real)-9.51056516295153642204e-01}, {(real)-8.09016994374947340241e-01, (real)-5.87785252292473248126e-01}, {(real)-1.00000000000000000000e+00, (real)-1.22464679914735320717e-16}, {(real)-8.09016994374947562285e-01, (real)5.87785252292473026081e-01}, {(real)-3.09016994374947562285e-01, (real)9.51056516295153531182e-01}, {(real)3.09016994374947229218e-01, (real)9.51056516295153642204e-01}, {(real)8.09016994374947340241e-01, (real)5.87785252292473359148e-01}}; };

template <> struct WTab<12> { static constexpr WPair w[12] = {{(real)1.00000000000000000000e+00, (real)-0.00000000000000000000e+00}, {(real)8.66025403784438707611e-01, (real)-4.99999999999999944489e-01}, {(real)5.00000000000000111022e-01, (real)-8.66025403784438596588e-01}, {(real)6.12323399573676603587e-17, (real)-1.00000000000000000000e+00}, {(real)-4.99999999999999777955e-01, (real)-8.66025403784438707611e-01}, {(real)-8.66025403784438707611e-01, (real)-4.99999999999999944489e-01}, {(real)-1.00000000000000000000e+00, (real)-1.22464679914735320717e-16}, {(real)-8.66025403784438818633e-01, (real)4.99999999999999722444e-01}, {(real)-5.00000000000000444089e-01, (real)8.66025403784438374544e-01}, {(real)-1.83697019872102968750e-16, (real)1.00000000000000000000e+00}, {(real)5.00000000000000111022e-01, (real)8.66025403784438596588e-01}, {(real)8.66025403784438374544e-01, (real)5.00000000000000444089e-01}}; };
template <> struct WTab<15> { static constexpr WPair w[15] = {{(real)1.00000000000000000000e+00, (real)-0.00000000000000000000e+00}, {(real)9.13545457642600866599e-01, (real)-4.06736643075800152758e-01}, {(real)6.69130606358858237570e-01, (real)-7.43144825477394133095e-01}, {(real)3.09016994374947451263e-01, (real)-9.51056516295153531182e-01}, {(real)-1.04528463267653332069e-01, (real)-9.94521895368273400884e-01}, {(real)-4.99999999999999777955e-01, (real)-8.66025403784438707611e-01}, {(real)-8.09016994374947340241e-01, (real)-5.87785252292473248126e-01}, {(real)-9.78147600733805688833e-01, (real)-2.07911690817759314820e-01}, {(real)-9.78147600733805688833e-01, (real)2.07911690817759065020e-01}, {(real)-8.09016994374947562285e-01, (real)5.87785252292473026081e-01}, {(real)-5.00000000000000444089e-01, (real)8.66025403784438374544e-01}, {(real)-1.04528463267654234126e-01, (real)9.94521895368273289861e-01}, {(real)3.09016994374947229218e-01, (real)9.51056516295153642204e-01}, {(real)6.69130606358858459615e-01, (real)7.43144825477394022073e-01}, {(real)9.13545457642600977621e-01, (real)4.06736643075800152758e-01}}; };
template <> struct WTab<20> { static constexpr WPair w[20] = {{(real)1.00000000000000000000e+00, (real)-0.00000000000000000000e+00}, {(real)9.51056516295153531182e-01, (real)-3.09016994374947395752e-01}, {(real)8.09016994374947451263e-01, (real)-5.87785252292473137103e-01}, {(real)5.87785252292473137103e-01, (real)-8.09016994374947451263e-01}, {(real)3.09016994374947451263e-01, (real)-9.51056516295153531182e-01}, {(real)6.12323399573676603587e-17, (real)-1.00000000000000000000e+00}, {(real)-3.09016994374947340241e-01, (real)-9.51056516295153642204e-01}, {(real)-5.87785252292473026081e-01, (real)-8.09016994374947451263e-01}, {(real)-8.09016994374947340241e-01, (real)-5.87785252292473248126e-01}, {(real)-9.51056516295153531182e-01, (real)-3.09016994374947506774e-01}, {(real)-1.00000000000000000000e+00, (real)-1.22464679914735320717e-16}, {(real)-9.51056516295153753227e-01, (real)3.09016994374946896151e-01}, {(real)-8.09016994374947562285e-01, (real)5.87785252292473026081e-01}, {(real)-5.87785252292473248126e-01, (real)8.09016994374947340241e-01}, {(real)-3.09016994374947562285e-01, (real)9.51056516295153531182e-01}, {(real)-1.83697019872102968750e-16, (real)1.00000000000000000000e+00}, {(real)3.09016994374947229218e-01, (real)9.51056516295153642204e-01}, {(real)5.87785252292472915059e-01, (real)8.09016994374947562285e-01}, {(real)8.09016994374947340241e-01, (real)5.87785252292473359148e-01}, {(real)9.51056516295153531182e-01, (real)3.09016994374947617796e-01}}; };
template <> struct WTab<24> { static constexpr WPair w[24] = {{(real)1.00000000000000000000e+00, (real)-0.00000000000000000000e+00}, {(real)9.65925826289068312214e-01, (real)-2.58819045102520739476e-01}, {(real)8.66025403784438707611e-01, (real)-4.99999999999999944489e-01}, {(real)7.07106781186547572737e-01, (real)-7.07106781186547461715e-01}, {(real)5.00000000000000111022e-01, (real)-8.66025403784438596588e-01}, {(real)2.58819045102520739476e-01, (real)-9.65925826289068312214e-01}, {(real)6.12323399573676603587e-17, (real)-1.00000000000000000000e+00}, {(real)-2.58819045102520628454e-01, (real)-9.65925826289068312214e-01}, {(real)-4.99999999999999777955e-01, (real)-8.66025403784438707611e-01}, {(real)-7.07106781186547461715e-01, (real)-7.07106781186547572737e-01}, {(real)-8.66025403784438707611e-01, (real)-4.99999999999999944489e-01}, {(real)-9.65925826289068201191e-01, (real)-2.58819045102521017032e-01}, {(real)-1.00000000000000000000e+00, (real)-1.22464679914735320717e-16}, {(real)-9.65925826289068312214e-01, (real)2.58819045102520794988e-01}, {(real)-8.66025403784438818633e-01, (real)4.99999999999999722444e-01}, {(real)-7.07106781186547905804e-01, (real)7.07106781186547128648e-01}, {(real)-5.00000000000000444089e-01, (real)8.66025403784438374544e-01}, {(real)-2.58819045102520628454e-01, (real)9.65925826289068312214e-01}, {(real)-1.83697019872102968750e-16, (real)1.00000000000000000000e+00}, {(real)2.58819045102520295387e-01, (real)9.65925826289068423236e-01}, {(real)5.00000000000000111022e-01, (real)8.66025403784438596588e-01}, {(real)7.07106781186547350693e-01, (real)7.07106781186547683760e-01}, {(real)8.66025403784438374544e-01, (real)5.00000000000000444089e-01}, {(real)9.65925826289068090169e-01, (real)2.58819045102521572144e-01}}; };
// v[n], n = j1*R2 + j2  ->  v[k], k = k1 + R1*k2  (X[k1 + R1 k2] = sum_j2 w_N^(j2 k1) [sum_j1 x[j1 R2 + j2] w_R1^(j1 k1)] w_R2^(j2 k2))
template <int R1, int R2, bool INV>
static __device__ __forceinline__ void dft_two_factor(real2* v) {
  constexpr int N = R1 * R2;
  real2 y[R2][R1];
#pragma unroll
  for (int j2 = 0; j2 < R2; ++j2) {
    real2 t[R1];
#pragma unroll
    for (int j1 = 0; j1 < R1; ++j1) t[j1] = v[j1 * R2 + j2];
    Dft<R1, INV>::run(t);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
      const int q = (k1 * j2) % N;
      const real2 w = make_real2(WTab<N>::w[q].re, INV ? -WTab<N>::w[q].im : WTab<N>::w[q].im);
      y[j2][k1] = q ? cmul(t[k1], w) : t[k1];
    }
  }
#pragma unroll
  for (int k1 = 0; k1 < R1; ++k1) {
    real2 t[R2];
#pragma unroll
    for (int j2 = 0; j2 < R2; ++j2) t[j2] = y[j2][k1];
    Dft<R2, INV>::run(t);
#pragma unroll
    for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = t[k2];
  }
}
template <bool INV> struct Dft<30, INV> { static __device__ __forceinline__ void run(real2* v) { dft_two_factor<6, 5, INV>(v); } };
template <bool INV> struct Dft<18, INV> { static __device__ __forceinline__ void run(real2* v) { dft_two_factor<6, 3, INV>(v); } };
template <bool INV> struct Dft<10, INV> { static __device__ __forceinline__ void run(real2* v) { dft_two_factor<5, 2, INV>(v); } };
template <bool INV> struct Dft<9, INV> { static __device__ __forceinline__ void run(real2* v) { dft_two_factor<3, 3, INV>(v); } };
// (radices the plan chooser does not use on its own: reachable through the options row_rad / passa_rad / mid_rad, for
// three-stage plans of lengths like 1920 = 16.15.8 or 960 = 16.12.5 -- measured in profiles/r03_notes.md section 13)
template <bool INV> struct Dft<12, INV> { static __device__ __forceinline__ void run(real2* v) { dft_two_factor<4, 3, INV>(v); } };
template <bool INV> struct Dft<15, INV> { static __device__ __forceinline__ void run(real2* v) { dft_two_factor<5, 3, INV>(v); } };
template <bool INV> struct Dft<20, INV> { static __device__ __forceinline__ void run(real2* v) { dft_two_factor<5, 4, INV>(v); } };
template <bool INV> struct Dft<24, INV> { static __device__ __forceinline__ void run(real2* v) { dft_two_factor<8, 3, INV>(v); } };

// v[m] *= w^m (m = 1..R-1), w = exp(-+ 2 pi i q1 / n) = tw[q1] (conjugated for the inverse).
template <int R, bool INV>
static __device__ __forceinline__ void twiddle_mul(real2* v, const real2* LPC_RESTRICT tw, int q1) {
  if constexpr (R != 8 && R != 16) {  // exact table entries for every power
#pragma unroll
    for (int m = 1; m < R; ++m) {
      real2 t = tw[q1 * m];
      v[m] = INV ? cmul_conj(v[m], t) : cmul(v[m], t);
    }
  } else {
    real2 w[16];
    w[1] = tw[q1]; w[2] = tw[2 * q1]; w[4] = tw[4 * q1];
    if (R == 16) w[8] = tw[8 * q1];
    w[3] = cmul(w[1], w[2]); w[5] = cmul(w[1], w[4]); w[6] = cmul(w[2], w[4]); w[7] = cmul(w[3], w[4]);
    if (R == 16) {
#pragma unroll
      for (int m = 9; m < 16; ++m) w[m] = cmul(w[m - 8], w[8]);
    }
#pragma unroll
    for (int m = 1; m < R; ++m) v[m] = INV ? cmul_conj(v[m], w[m]) : cmul(v[m], w[m]);
  }
}

// The same products from base powers that arrive as two 16-byte loads, contiguous across the lanes of a wave: `blk` is the
// stage's block of a lane-ordered table (lpc_sfft.h: SPlan::tws_off), nb its butterflies, j this lane's butterfly.
template <int R, bool INV>
static __device__ __forceinline__ void twiddle_mul_lane(real2* v, const real2* LPC_RESTRICT blk, int nb, int j) {
  if constexpr (R == 8 || R == 16) {
    const real4_t lo = *(const real4_t*)(blk + 2 * j), hi = *(const real4_t*)(blk + 2 * (nb + j));
    real2 w[16];
    w[1] = make_real2(lo.x, lo.y); w[2] = make_real2(lo.z, lo.w); w[4] = make_real2(hi.x, hi.y);
    if (R == 16) w[8] = make_real2(hi.z, hi.w);
    w[3] = cmul(w[1], w[2]); w[5] = cmul(w[1], w[4]); w[6] = cmul(w[2], w[4]); w[7] = cmul(w[3], w[4]);
    if (R == 16) {
#pragma unroll
      for (int m = 9; m < 16; ++m) w[m] = cmul(w[m - 8], w[8]);
    }
#pragma unroll
    for (int m = 1; m < R; ++m) v[m] = INV ? cmul_conj(v[m], w[m]) : cmul(v[m], w[m]);
  } else {      // every power w^(q m), m = 1 .. R - 1, as its own table row [m - 1][j] (the entries twiddle_mul gathers)
    real2 t[R];
#pragma unroll
    for (int m = 1; m < R; ++m) t[m] = blk[(m - 1) * nb + j];
#pragma unroll
    for (int m = 1; m < R; ++m) v[m] = INV ? cmul_conj(v[m], t[m]) : cmul(v[m], t[m]);
  }
}

// Row mode (one transform per workgroup, BT == 1) may store element i at LDS slot i + i/8: the
// early Stockham stages write with a lane stride of R real2 (an 8- or 16-way conflict on the
// 32 x 4-byte banks of ds_write_b64).  Only used when it stays AFFINE inside every stage
// (plan.skew_ok, checked on the host) so that a butterfly's R accesses are base + m*stride'.
// LDS layout of a row tile (the template parameter named SK / SKEW everywhere): where element i of the tile lives.
//   0  natural
//   1  i + i/8  (rounds 1-3): removes the 8- / 16-way conflicts of the first stage's stride-R stores, but every
//      CONTIGUOUS access then spans 36 slots per 32 lanes and wraps onto its own banks -- ds_read_b64 2 -> 4 array cycles,
//      ds_write_b64 4 -> 8: half of all LDS cycles of the row kernels were conflicts (SQ_LDS_BANK_CONFLICT /
//      SQ_LDS_IDX_ACTIVE = 0.44-0.50 in profiles/r03fin_c2_counters.md; tools/lds_model.py reproduces the factor)
#define LPC_LAY_NONE 0
#define LPC_LAY_SKEW8 1
template <int SKEW>
static __device__ __forceinline__ int lds_slot(int i) {
  return SKEW == LPC_LAY_SKEW8 ? i + (i >> 3) : i;
}
// slot(i + m * stride) == slot(i) + m * lds_stride(stride) for every i, m?  (then a stage addresses its R elements with
// immediates from one base)
template <int SKEW>
static __host__ __device__ constexpr bool lds_affine(int stride) {
  return SKEW == LPC_LAY_SKEW8 ? stride % 8 == 0 : true;
}
template <int SKEW>
static __host__ __device__ constexpr int lds_stride(int stride) {
  return SKEW == LPC_LAY_SKEW8 ? stride + (stride >> 3) : stride;
}
static __host__ __device__ __forceinline__ int lds_slots_skewed(int n) { return n + (n >> 3) + 1; }
static __host__ __device__ __forceinline__ int lds_slots_of(int n, int layout) {
  return layout == LPC_LAY_SKEW8 ? lds_slots_skewed(n) : n;
}

// One Stockham stage over a tile of BT transforms held in LDS (in place).  Ends with a barrier.
template <int R, int NT, int EMAX, bool INV, int SKEW>
static __device__ __forceinline__ void fft_stage(real2* s, int n, int BT, FastDiv btdiv, int ns,
                                                  FastDiv nsdiv, int twstep,
                                                  const real2* LPC_RESTRICT tw, int tid) {
  constexpr int MAXB = (EMAX + R - 1) / R;
  const int nb = n / R;
  const int nwork = nb * BT;
  const int istride = nb * BT;   // tile distance between the R inputs of one butterfly
  const int ostride = ns * BT;   // tile distance between its R outputs
  const int rs = SKEW ? istride + (istride >> 3) : istride;   // same, in (skewed) LDS slots
  const int ws = SKEW ? ostride + (ostride >> 3) : ostride;
  real2 v[MAXB][R];
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
      const int rb = lds_slot<SKEW>(w);
#pragma unroll
      for (int m = 0; m < R; ++m) v[b][m] = s[rb + m * rs];
      if (ns > 1) twiddle_mul<R, INV>(v[b], tw, k * twstep);
      Dft<R, INV>::run(v[b]);
      obase[b] = lds_slot<SKEW>((jq * ns * R + k) * BT + c);
    }
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    if (obase[b] >= 0) {
      if (SKEW && ns == 1) {  // R is a multiple of 8 here (plan.skew_ok): constant offsets
#pragma unroll
        for (int m = 0; m < R; ++m) s[obase[b] + m + (m >> 3)] = v[b][m];
      } else {
#pragma unroll
        for (int m = 0; m < R; ++m) s[obase[b] + m * ws] = v[b][m];
      }
    }
  }
  __syncthreads();
}

// In-place FFT of BT interleaved transforms of length plan.n held in LDS (natural order,
// element (i, c) at slot(i*BT + c)).  Precondition: tile written and a barrier passed.
// Postcondition: result in natural order, barrier passed.  Unnormalised in both directions.
template <int NT, int EMAX, bool INV, int SKEW = false>
static __device__ __forceinline__ void lds_fft(real2* s, const Fft1dPlan& p, int BT, FastDiv btdiv,
                                                int tid, int first_stage = 0, int skip_last = 0) {
  for (int st = first_stage; st < p.nst - skip_last; ++st) {
    const int ns = p.ns[st];
    const FastDiv nd = p.nsdiv[st];
    const int ts = p.twstep[st];
#define LPC_STAGE(R) fft_stage<R, NT, EMAX, INV, SKEW>(s, p.n, BT, btdiv, ns, nd, ts, p.tw, tid)
    switch (p.radix[st]) {
#ifdef LPC_ENABLE_R16  // measured slower on MI355X: +50 VGPRs in every kernel (profiles/r01b_notes.md)
      case 16: LPC_STAGE(16); break;
#endif
      case 8: LPC_STAGE(8); break;
      case 6: LPC_STAGE(6); break;
      case 5: LPC_STAGE(5); break;
      case 4: LPC_STAGE(4); break;
      case 3: LPC_STAGE(3); break;
      default: LPC_STAGE(2); break;
    }
#undef LPC_STAGE
  }
}

// Tag for "the tile already is / shall stay in LDS in natural order" as source or sink of fft_tile.
struct LdsNatural {};

// Tile transform with pluggable source and sink:
//   src(i, c) -> real2   element i of transform c (global loads, padding, residuals, ...)
//   dst(i, c, v)          receives output element i of transform c (natural order)
// All of a thread's src() calls are issued before the first LDS write (loops unrolled to the
// compile-time bound EMAX): with a run-time trip count the compiler emits load / wait / ds_write
// per element and the kernel serialises on HBM latency (measured: +25 % on the column passes).
// SRC_LDS: src itself reads the LDS tile, so a barrier separates its reads from the refill.
// First Stockham stage (ns = 1: no twiddles) fused into the tile fill: every thread loads exactly the
// R inputs of its own butterflies (element j + m*n/R, still coalesced across lanes), transforms them
// in the staging registers and writes the stage OUTPUT to LDS -- one LDS round trip and two barriers
// less per transform.  Register need = the EMAX staging registers the plain fill uses anyway.
template <int R, int NT, int EMAX, bool INV, int SKEW, bool SRC_LDS, class Src, class Fix>
static __device__ __forceinline__ void fft_first_stage_fused(real2* s, int n, int BT, FastDiv btdiv, int tid,
                                                              Src& src, Fix& fix) {
  constexpr int MAXB = (EMAX + R - 1) / R;
  const int nb = n / R;
  const int nwork = nb * BT;
  real2 v[MAXB][R];
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    const int w = tid + b * NT;
    if (w < nwork) {
      const int j = (int)fd_div((unsigned)w, btdiv);
      const int c = w - j * BT;
#pragma unroll
      for (int m = 0; m < R; ++m) v[b][m] = src(j + m * nb, c);
    }
  }
  if (SRC_LDS) __syncthreads();
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    const int w = tid + b * NT;
    if (w < nwork) {
      const int j = (int)fd_div((unsigned)w, btdiv);
      const int c = w - j * BT;
#pragma unroll
      for (int m = 0; m < R; ++m) v[b][m] = fix(j + m * nb, c, v[b][m]);
      Dft<R, INV>::run(v[b]);
      const int ob = lds_slot<SKEW>(j * R * BT + c);
      if (SKEW) {  // row mode, BT == 1: element j*R + m (R % 8 == 0 by plan.skew_ok)
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

// Last Stockham stage fused into the drain: outputs go from the butterfly registers straight to the sink
// (coalesced: consecutive lanes hold consecutive output elements) -- again one LDS round trip and two
// barriers less.  The tile must have passed a barrier after the previous stage's writes.
template <int R, int NT, int EMAX, bool INV, int SKEW, class Dst>
static __device__ __forceinline__ void fft_last_stage_fused(real2* s, int n, int BT, FastDiv btdiv, int ns,
                                                             FastDiv nsdiv, int twstep,
                                                             const real2* LPC_RESTRICT tw, int tid, Dst& dst) {
  constexpr int MAXB = (EMAX + R - 1) / R;
  const int nb = n / R;
  const int nwork = nb * BT;
  const int istride = nb * BT;
  const int rs = SKEW ? istride + (istride >> 3) : istride;
  real2 v[MAXB][R];
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    const int w = tid + b * NT;
    if (w < nwork) {
      const int rb = lds_slot<SKEW>(w);
#pragma unroll
      for (int m = 0; m < R; ++m) v[b][m] = s[rb + m * rs];
    }
  }
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    const int w = tid + b * NT;
    if (w < nwork) {
      const int j = (int)fd_div((unsigned)w, btdiv);
      const int c = w - j * BT;
      const int jq = (int)fd_div((unsigned)j, nsdiv);
      const int k = j - jq * ns;
      if (ns > 1) twiddle_mul<R, INV>(v[b], tw, k * twstep);
      Dft<R, INV>::run(v[b]);
      const int oi = jq * ns * R + k;
#pragma unroll
      for (int m = 0; m < R; ++m) dst(oi + m * ns, c, v[b][m]);
    }
  }
}

struct NoFix {
  __device__ __forceinline__ real2 operator()(int, int, real2 v) const { return v; }
};

// fix(i, c, v): optional per-element transform applied AFTER the batched loads have landed (e.g. the
// four-step twiddle of the inverse pass A): its own table gathers then do not delay the tile loads.
// FUSE1: fuse the first stage into the fill (fft_first_stage_fused).  A per-call-site choice, measured
// on MI355X (profiles/r01b_notes.md): it pays for the forward row pass (-10 %) and the inverse column
// pass A (-13 %), costs +25 % on the forward pass A, and compiling BOTH paths slows the fused middle.
// FUSEL: fuse the last stage into the drain (fft_last_stage_fused); same per-call-site rule.
template <int NT, int EMAX, bool INV, int SKEW, bool SRC_LDS, bool FUSE1 = false, bool FUSEL = false, class Src,
          class Dst, class Fix = NoFix>
// skip_first / skip_last: the caller has fused that many stages at the front / back itself (row kernels
// fold a radix-2 stage into the Hermitian tangling): only stages [skip_first, nst - skip_last) run here.
static __device__ __forceinline__ void fft_tile(real2* s, const Fft1dPlan& p, int BT, FastDiv btdiv,
                                                 int tid, Src src, Dst dst, Fix fix = Fix(), int skip_first = 0,
                                                 int skip_last = 0) {
  const int nelem = p.n * BT;
  int first_stage = skip_first;
  if constexpr (FUSE1 && !std::is_same<Src, LdsNatural>::value) {
    if (p.nst >= 1) {
#define LPC_FUSED(R) fft_first_stage_fused<R, NT, EMAX, INV, SKEW, SRC_LDS>(s, p.n, BT, btdiv, tid, src, fix)
      switch (p.radix[0]) {
        case 8: LPC_FUSED(8); break;
        case 6: LPC_FUSED(6); break;
        case 5: LPC_FUSED(5); break;
        case 4: LPC_FUSED(4); break;
        case 3: LPC_FUSED(3); break;
        default: LPC_FUSED(2); break;
      }
#undef LPC_FUSED
      first_stage = 1;
    }
  }
  if (first_stage == 0)
  if constexpr (!std::is_same<Src, LdsNatural>::value) {
    real2 v[EMAX];
#pragma unroll
    for (int k = 0; k < EMAX; ++k) {
      const int e = tid + k * NT;
      if (e < nelem) {
        const int i = (int)fd_div((unsigned)e, btdiv);
        v[k] = src(i, e - i * BT);
      }
    }
    if (SRC_LDS) __syncthreads();
#pragma unroll
    for (int k = 0; k < EMAX; ++k) {
      const int e = tid + k * NT;
      if (e < nelem) {
        if constexpr (std::is_same<Fix, NoFix>::value) {
          s[lds_slot<SKEW>(e)] = v[k];
        } else {
          const int i = (int)fd_div((unsigned)e, btdiv);
          s[lds_slot<SKEW>(e)] = fix(i, e - i * BT, v[k]);
        }
      }
    }
    __syncthreads();
  }
  if constexpr (FUSEL && !std::is_same<Dst, LdsNatural>::value) {
    if (p.nst - first_stage >= 1 && skip_last == 0) {
      lds_fft<NT, EMAX, INV, SKEW>(s, p, BT, btdiv, tid, first_stage, 1);
      const int st = p.nst - 1;
#define LPC_FUSEDL(R) \
  fft_last_stage_fused<R, NT, EMAX, INV, SKEW>(s, p.n, BT, btdiv, p.ns[st], p.nsdiv[st], p.twstep[st], p.tw, tid, dst)
      switch (p.radix[st]) {
        case 8: LPC_FUSEDL(8); break;
        case 6: LPC_FUSEDL(6); break;
        case 5: LPC_FUSEDL(5); break;
        case 4: LPC_FUSEDL(4); break;
        case 3: LPC_FUSEDL(3); break;
        default: LPC_FUSEDL(2); break;
      }
#undef LPC_FUSEDL
      return;
    }
  }
  lds_fft<NT, EMAX, INV, SKEW>(s, p, BT, btdiv, tid, first_stage, skip_last);
  if constexpr (!std::is_same<Dst, LdsNatural>::value) {
#pragma unroll
    for (int k = 0; k < EMAX; ++k) {
      const int e = tid + k * NT;
      if (e < nelem) {
        const int i = (int)fd_div((unsigned)e, btdiv);
        dst(i, e - i * BT, s[lds_slot<SKEW>(e)]);
      }
    }
  }
}
