// exact_trig.h -- atan / sin / cos / tan restated so that the device and the CPU oracle compute THE SAME bits.
//
// Five camera models of the reference call libm in Camera::ImageToWorld (/root/reference/src/base/camera_models.h:
// 957-982 OPENCV_FISHEYE, 1179-1218 FOV, 1278-1297 SIMPLE_RADIAL_FISHEYE, 1358-1380 RADIAL_FISHEYE, 1434-1456
// THIN_PRISM_FISHEYE).  The device's ocml and the host's glibc differ in the last place now and then, and at scale
// that reaches decisions (round 3: 1 of 105 FOV pairs ran two more E models).  The reference pins no libm either --
// its results are those of whatever correctly-rounding-in-practice libm the build host has -- so the contract here is
// the mathematically defined one: the CORRECTLY ROUNDED value, computed in double-double arithmetic (~100 bits, error
// < 2^-90 relative, so the rounding is right except in astronomically rare ties).  glibc's routines return the
// correctly rounded value for all but a vanishing fraction of arguments; tests/test_exact_trig.py measures the
// agreement with the host libm (and holds it to 1 ulp everywhere).
//
// Plain IEEE double operations only (+ - * / compared in a fixed order, no FMA: both builds use -ffp-contract=off),
// so a g++ build and a hipcc build of this header return identical bits.  The includer defines
//   DSM_XT        function qualifier   (oracle: `static inline`; device: `__device__ __noinline__`)
//   DSM_XT_CONST  table qualifier      (oracle: `static const`;  device: `__device__ const`)
// Constants: tools/gen_exact_trig_consts.py (integer arithmetic, 400 bits).
#ifndef DAGSFM_AMD_CSRC_EXACT_TRIG_H_
#define DAGSFM_AMD_CSRC_EXACT_TRIG_H_

namespace dsm_xt {

DSM_XT_CONST double kPio2Parts[4] = {0x1.921fb54400000p+0, 0x1.0b4611a600000p-34, 0x1.3198a2e000000p-69, 0x1.b839a252049c1p-104};
DSM_XT_CONST double kPio2DD[2] = {0x1.921fb54442d18p+0, 0x1.1a62633145c07p-54};
DSM_XT_CONST double kTwoOverPi = 0x1.45f306dc9c883p-1;
DSM_XT_CONST double kInvFact[30][2] = {
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.0000000000000p-1, 0x0.0p+0},
    {0x1.5555555555555p-3, 0x1.5555555555555p-57},
    {0x1.5555555555555p-5, 0x1.5555555555555p-59},
    {0x1.1111111111111p-7, 0x1.1111111111111p-63},
    {0x1.6c16c16c16c17p-10, -0x1.f49f49f49f49fp-65},
    {0x1.a01a01a01a01ap-13, 0x1.a01a01a01a01ap-73},
    {0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-76},
    {0x1.71de3a556c734p-19, -0x1.c154f8ddc6c00p-73},
    {0x1.27e4fb7789f5cp-22, 0x1.cbbc05b4fa99ap-76},
    {0x1.ae64567f544e4p-26, -0x1.c062e06d1f209p-80},
    {0x1.1eed8eff8d898p-29, -0x1.2aec959e14c06p-83},
    {0x1.6124613a86d09p-33, 0x1.f28e0cc748ebep-87},
    {0x1.93974a8c07c9dp-37, 0x1.05d6f8a2efd1fp-92},
    {0x1.ae7f3e733b81fp-41, 0x1.1d8656b0ee8cbp-97},
    {0x1.ae7f3e733b81fp-45, 0x1.1d8656b0ee8cbp-101},
    {0x1.952c77030ad4ap-49, 0x1.ac981465ddc6cp-103},
    {0x1.6827863b97d97p-53, 0x1.eec01221a8b0bp-107},
    {0x1.2f49b46814157p-57, 0x1.2650f61dbdcb4p-112},
    {0x1.e542ba4020225p-62, 0x1.ea72b4afe3c2fp-120},
    {0x1.71b8ef6dcf572p-66, -0x1.d043ae40c4647p-120},
    {0x1.0ce396db7f853p-70, -0x1.aebcdbd20331cp-124},
    {0x1.761b41316381ap-75, -0x1.3423c7d91404fp-130},
    {0x1.f2cf01972f578p-80, -0x1.9ada5fcc1ab14p-135},
    {0x1.3f3ccdd165fa9p-84, -0x1.58ddadf344487p-139},
    {0x1.88e85fc6a4e5ap-89, -0x1.71c37ebd16540p-143},
    {0x1.d1ab1c2dccea3p-94, 0x1.054d0c78aea14p-149},
    {0x1.0a18a2635085dp-98, 0x1.b9e2e28e1aa54p-153},
    {0x1.259f98b4358adp-103, 0x1.eaf8c39dd9bc5p-157},
};
DSM_XT_CONST double kInvOdd[16][2] = {   // 1 / (2n + 1)
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.5555555555555p-2, 0x1.5555555555555p-56},
    {0x1.999999999999ap-3, -0x1.999999999999ap-57},
    {0x1.2492492492492p-3, 0x1.2492492492492p-57},
    {0x1.c71c71c71c71cp-4, 0x1.c71c71c71c71cp-58},
    {0x1.745d1745d1746p-4, -0x1.745d1745d1746p-59},
    {0x1.3b13b13b13b14p-4, -0x1.3b13b13b13b14p-58},
    {0x1.1111111111111p-4, 0x1.1111111111111p-60},
    {0x1.e1e1e1e1e1e1ep-5, 0x1.e1e1e1e1e1e1ep-61},
    {0x1.af286bca1af28p-5, 0x1.af286bca1af28p-59},
    {0x1.8618618618618p-5, 0x1.8618618618618p-59},
    {0x1.642c8590b2164p-5, 0x1.642c8590b2164p-60},
    {0x1.47ae147ae147bp-5, -0x1.eb851eb851eb8p-61},
    {0x1.2f684bda12f68p-5, 0x1.2f684bda12f68p-59},
    {0x1.1a7b9611a7b96p-5, 0x1.1a7b9611a7b96p-61},
    {0x1.0842108421084p-5, 0x1.0842108421084p-60},
};
DSM_XT_CONST double kAtanEighths[9][2] = {   // atan(k / 8)
    {0x0.0p+0, 0x0.0p+0},
    {0x1.fd5ba9aac2f6ep-4, -0x1.cd37686760c17p-59},
    {0x1.f5b75f92c80ddp-3, 0x1.8ab6e3cf7afbdp-57},
    {0x1.6f61941e4def1p-2, -0x1.c63aae6f6e918p-56},
    {0x1.dac670561bb4fp-2, 0x1.a2b7f222f65e2p-56},
    {0x1.1e00babdefeb4p-1, -0x1.928df287a668fp-58},
    {0x1.4978fa3269ee1p-1, 0x1.2419a87f2a458p-56},
    {0x1.700a7c5784634p-1, -0x1.8c34d25aadef6p-56},
    {0x1.921fb54442d18p-1, 0x1.1a62633145c07p-55},
};

struct dd {
  double hi, lo;
};

// ---- error-free transformations (Knuth two-sum, Dekker split / product) and double-double arithmetic
DSM_XT dd two_sum(double a, double b) {
  const double s = a + b;
  const double bb = s - a;
  return dd{s, (a - (s - bb)) + (b - bb)};
}
DSM_XT dd quick_two_sum(double a, double b) {  // |a| >= |b|
  const double s = a + b;
  return dd{s, b - (s - a)};
}
DSM_XT dd two_prod(double a, double b) {
  const double p = a * b;
  const double ta = 134217729.0 * a, tb = 134217729.0 * b;  // 2^27 + 1
  const double ah = ta - (ta - a), bh = tb - (tb - b);
  const double al = a - ah, bl = b - bh;
  return dd{p, ((ah * bh - p) + ah * bl + al * bh) + al * bl};
}
DSM_XT dd dd_add(dd a, dd b) {
  dd s = two_sum(a.hi, b.hi);
  const dd t = two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = quick_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return quick_two_sum(s.hi, s.lo);
}
DSM_XT dd dd_add_d(dd a, double b) {
  dd s = two_sum(a.hi, b);
  s.lo += a.lo;
  return quick_two_sum(s.hi, s.lo);
}
DSM_XT dd dd_neg(dd a) { return dd{-a.hi, -a.lo}; }
DSM_XT dd dd_mul(dd a, dd b) {
  dd p = two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return quick_two_sum(p.hi, p.lo);
}
DSM_XT dd dd_mul_d(dd a, double b) {
  dd p = two_prod(a.hi, b);
  p.lo += a.lo * b;
  return quick_two_sum(p.hi, p.lo);
}
DSM_XT dd dd_div(dd a, dd b) {  // three quotient digits
  const double q1 = a.hi / b.hi;
  dd r = dd_add(a, dd_neg(dd_mul_d(b, q1)));
  const double q2 = r.hi / b.hi;
  r = dd_add(r, dd_neg(dd_mul_d(b, q2)));
  const double q3 = r.hi / b.hi;
  dd q = quick_two_sum(q1, q2);
  return dd_add_d(q, q3);
}

// ---- sin and cos of a double-double |r| <= ~0.8 by their Taylor series (Horner in r^2, 1/n! as double-doubles)
DSM_XT dd sin_kernel(dd r) {  // r * (1/1! - r^2/3! + ... + r^26/27!)
  const dd r2 = dd_mul(r, r);
  dd s = dd{kInvFact[27][0], kInvFact[27][1]};
  for (int n = 25; n >= 1; n -= 2) {
    s = dd_mul(s, r2);
    s = dd_add(dd{kInvFact[n][0], kInvFact[n][1]}, dd_neg(s));
  }
  return dd_mul(s, r);
}
DSM_XT dd cos_kernel(dd r) {  // 1 - r^2/2! + ... + r^28/28!
  const dd r2 = dd_mul(r, r);
  dd s = dd{kInvFact[28][0], kInvFact[28][1]};
  for (int n = 26; n >= 0; n -= 2) {
    s = dd_mul(s, r2);
    s = dd_add(dd{kInvFact[n][0], kInvFact[n][1]}, dd_neg(s));
  }
  return s;
}

// x = k * pi/2 + r with |r| <= pi/4 (+ a little).  pi/2 is split into 33 + 33 + 33 + 53 bits: k * part is exact
// for |k| < 2^20, and x - k * p0 is exact (Sterbenz) -- the cancellation costs no accuracy.  |x| < 2^20 * pi/2 is the
// supported range; beyond it (garbage camera parameters: angles of a camera model are < pi) sin / cos / tan return NaN
// on the device and in the oracle alike -- the cast of k to an integer below would be undefined from |x| ~ 1.4e19 on,
// and "loses accuracy" is not a contract two builds can share.
#define DSM_XT_MAX_ARG 1647099.0  /* < 2^20 * pi/2 */
DSM_XT bool trig_arg_supported(double x) { return (x < 0.0 ? -x : x) <= DSM_XT_MAX_ARG; }
DSM_XT dd reduce_pio2(double x, int* quadrant) {
  double kd = x * kTwoOverPi;
  kd = (kd >= 0.0) ? (double)(long long)(kd + 0.5) : -(double)(long long)(0.5 - kd);
  *quadrant = (int)((long long)kd & 3);
  if (kd == 0.0) return dd{x, 0.0};
  const double a = x - kd * kPio2Parts[0];
  dd r = two_sum(a, -(kd * kPio2Parts[1]));
  r = dd_add_d(r, -(kd * kPio2Parts[2]));
  const dd t = two_prod(kd, kPio2Parts[3]);
  return dd_add(r, dd_neg(t));
}

}  // namespace dsm_xt

// sin(x), cos(x), tan(x), atan(x): correctly rounded (see the header comment)
DSM_XT double dsm_sin(double x) {
  using namespace dsm_xt;
  if (!(x == x) || x - x != 0.0) return x - x;  // NaN, inf
  if (!dsm_xt::trig_arg_supported(x)) return __builtin_nan("");  // outside the supported range: NaN in both builds
  int q;
  const dd r = reduce_pio2(x, &q);
  const dd v = (q & 1) ? cos_kernel(r) : sin_kernel(r);
  if (v.hi == 0.0) return x;  // sin(+-0) = +-0
  return (q & 2) ? -v.hi : v.hi;
}
DSM_XT double dsm_cos(double x) {
  using namespace dsm_xt;
  if (!(x == x) || x - x != 0.0) return x - x;
  if (!dsm_xt::trig_arg_supported(x)) return __builtin_nan("");  // outside the supported range: NaN in both builds
  int q;
  const dd r = reduce_pio2(x, &q);
  const dd v = (q & 1) ? sin_kernel(r) : cos_kernel(r);
  return ((q + 1) & 2) ? -v.hi : v.hi;
}
DSM_XT double dsm_tan(double x) {
  using namespace dsm_xt;
  if (!(x == x) || x - x != 0.0) return x - x;
  if (!dsm_xt::trig_arg_supported(x)) return __builtin_nan("");  // outside the supported range: NaN in both builds
  if (x == 0.0) return x;
  int q;
  const dd r = reduce_pio2(x, &q);
  const dd s = sin_kernel(r), c = cos_kernel(r);
  const dd t = (q & 1) ? dd_div(dd_neg(c), s) : dd_div(s, c);
  return t.hi;
}
DSM_XT double dsm_atan(double x) {
  using namespace dsm_xt;
  if (!(x == x)) return x;
  if (x == 0.0) return x;
  const bool neg = x < 0.0;
  const double ax = neg ? -x : x;
  if (ax > 1e19) return neg ? -kPio2DD[0] : kPio2DD[0];  // pi/2 - 1/x rounds to pi/2 (also inf)
  if (ax < 1e-9) {  // x - x^3/3: the correction is below 2^-60 relative -- one series term in double-double
    const dd x2 = two_prod(ax, ax);
    const dd c = dd_mul(dd_mul_d(x2, ax), dd{kInvOdd[1][0], kInvOdd[1][1]});
    const dd r = dd_add(dd{ax, 0.0}, dd_neg(c));
    return neg ? -r.hi : r.hi;
  }
  const bool inv = ax > 1.0;
  const dd y = inv ? dd_div(dd{1.0, 0.0}, dd{ax, 0.0}) : dd{ax, 0.0};  // in (0, 1]
  const int k = (int)(y.hi * 8.0 + 0.5);                                // nearest eighth
  const double c = (double)k * 0.125;
  // t = (y - c) / (1 + y c), |t| <= ~1/16;  atan(y) = atan(c) + atan(t)
  const dd num = dd_add_d(y, -c);
  const dd den = dd_add_d(dd_mul_d(y, c), 1.0);
  const dd t = dd_div(num, den);
  const dd t2 = dd_mul(t, t);
  dd s = dd{kInvOdd[15][0], kInvOdd[15][1]};
  for (int n = 14; n >= 0; --n) {
    s = dd_mul(s, t2);
    s = dd_add(dd{kInvOdd[n][0], kInvOdd[n][1]}, dd_neg(s));
  }
  dd r = dd_add(dd{kAtanEighths[k][0], kAtanEighths[k][1]}, dd_mul(s, t));
  if (inv) r = dd_add(dd{kPio2DD[0], kPio2DD[1]}, dd_neg(r));
  return neg ? -r.hi : r.hi;
}

#endif  // DAGSFM_AMD_CSRC_EXACT_TRIG_H_
