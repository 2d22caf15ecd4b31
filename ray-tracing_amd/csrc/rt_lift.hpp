// rt_lift.hpp -- the matrix-core form of the conservative sphere filter (pass A of the closest-hit search).
//
// The sign test of the vector filters,  q = (o.dn - c.dn)^2 - (|o - c|^2 - r^2 - slack),  is bilinear in a lifted
// ray vector R(o, dn) and a lifted sphere vector S(c, r):
//
//     q = [ (o.dn)^2 - |o|^2 (1-k) ] * 1  +  sum_a 2 (o_a - (o.dn) dn_a) * c_a  +  sum_{a<=b} dn_a dn_b * m_ab c_a c_b
//         +  1 * ( -(|c|^2 (1-k) - r^2 (1+k) - k_abs) ),                          m_aa = 1, m_ab = 2 otherwise,
//
// eleven products, so a 32 x 32 tile of (sphere, ray) pairs is one small matrix product and runs on the matrix
// cores (v_mfma_f32_32x32x16_f16).  Each factor is carried as an FP16 pair hi + lo (22 significant bits) and each
// product as hi*hi + hi*lo + lo*hi, which fills K = 32:
//
//     slot  0, 1      E_hi, E_lo      x  1
//     slot  2, 3, 4   1               x  NKQ_hi, NKQ_mid, NKQ_lo
//     slot  5 + 3t .. rho_hi, rho_hi, rho_lo  x  sigma_hi, sigma_lo, sigma_hi        t = 0..8
//                     rho   = Wx Wy Wz Pxx Pyy Pzz Pxy Pxz Pyz   (ray side)
//                     sigma = cx cy cz cx^2 cy^2 cz^2 2cxcy 2cxcz 2cycz   (sphere side)
//
// Coordinates are first translated by T and scaled by sigma = 2^e (both per scene, chosen on the host) so that
// the bulk of the scene sits well inside the FP16 range.  A sphere whose lifted values would overflow FP16, or
// whose own share of the slack would more than double its radius^2, is not filtered at all: its record makes q a
// large positive constant, so it is always handed to the exact pass B.  Padding records make q large negative.
//
// Error budget (u = 2^-24, S = |o'|^2 + |c'|^2 in scaled units, M = sum of |terms| <= ~2.8 S + r^2):
//   FP16 pairs: 4u per factor and 4u for the dropped lo*lo product -> <= 12u M, plus 2^-25 absolute per
//   subnormal lo part (<= ~2^-21 in total against factors <= 1, the rest proportional to S);
//   matrix-core accumulation: measured <= 4.4u M per 16-term instruction, two chained -> <= ~9u M;
//   ray-side roundings of E, W, P: <= u M, of o.dn: <= 4u |o'||o' - c'|;
//   reference side: its discriminant b*b - a*c >= 0 implies b^2/a >= |oc|^2 - r^2 - 17u |oc|^2 exactly.
// Total <= ~100u S against the slack of kMfmaSlack (S + r^2) = 256u (S + r^2) plus kMfmaAbsSlack; the kDirScale
// inflation of dn adds more.  rt_selftest_filter measures the guarantee on adversarial pairs.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_HD __host__ __device__ inline

namespace rt {

constexpr float kMfmaSlack = 0x1p-16f;
constexpr double kMfmaAbsSlack = 0x1p-18;   // in scaled units
constexpr double kHalfLimit = 60000.0;      // lifted magnitudes are kept below this (FP16 max is 65504)
constexpr int kLiftSlots = 32;

struct LiftFrame {  // scene translation and power-of-two scale of the lifted coordinates
    float tx, ty, tz, sigma;
};

RT_HD void split2(double v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)(float)v;
    lo = (_Float16)(float)(v - (double)(float)hi);
}

// A record that makes q = +kHalfLimit (always a candidate) or -kHalfLimit (never one) for every ray.
RT_HD void lift_constant(bool candidate, _Float16 s[kLiftSlots]) {
    for (int i = 0; i < kLiftSlots; ++i) s[i] = (_Float16)0.0f;
    s[2] = (_Float16)(float)(candidate ? kHalfLimit : -kHalfLimit);
}

// Sphere side.  Returns false (and the always-candidate record) when the sphere cannot be filtered.
// slack is kMfmaSlack except in the self-test, which probes the margin with smaller values.
RT_HD bool lift_sphere(double cx, double cy, double cz, double r, const LiftFrame &F, double slack, _Float16 s[kLiftSlots]) {
    const double g = (double)F.sigma;
    const double x = (cx - (double)F.tx) * g, y = (cy - (double)F.ty) * g, z = (cz - (double)F.tz) * g, rr = r * g;
    const double cc = x * x + y * y + z * z, r2 = rr * rr;
    const double sig[9] = {x, y, z, x * x, y * y, z * z, 2.0 * x * y, 2.0 * x * z, 2.0 * y * z};
    const double kq = cc * (1.0 - slack) - r2 * (1.0 + slack) - kMfmaAbsSlack;
    bool ok = r2 > 0.0 && cc == cc && r2 == r2 && !(kq > kHalfLimit) && !(kq < -kHalfLimit) && slack * (cc + r2) + kMfmaAbsSlack <= r2;
    for (int t = 0; t < 9; ++t) ok = ok && !(sig[t] > kHalfLimit) && !(sig[t] < -kHalfLimit);
    if (!ok) {
        lift_constant(true, s);
        return false;
    }
    s[0] = (_Float16)1.0f;
    s[1] = (_Float16)1.0f;
    // -kq in three parts; the remainder after them is below 2^-33 |kq| + 2^-25
    double v = -kq;
    for (int i = 2; i < 5; ++i) {
        s[i] = (_Float16)(float)v;
        v -= (double)(float)s[i];
    }
    for (int t = 0; t < 9; ++t) {
        _Float16 hi, lo;
        split2(sig[t], hi, lo);
        s[5 + 3 * t] = hi;
        s[6 + 3 * t] = lo;
        s[7 + 3 * t] = hi;
    }
    return true;
}

// Where slot k of the sphere in position p (0..63) of a 64-sphere block lives in the A-operand array, which is
// laid out [32-sphere group][K half][lane] x 8 halves so that every lane fetches its operand with one 16-byte
// read.  The matrix instruction returns rows 8a + 4h + b (a, b = 0..3) to lane half h; spheres are assigned to
// rows so that after the two halves exchange their words, bit 31 - j of word w is sphere 32 w + j of the block.
RT_HD size_t lift_index(int block, int p, int k) {
    const int h = p >> 5, g = (p >> 4) & 1, kk = p & 15;
    const int row = 8 * (kk >> 2) + 4 * h + (kk & 3);
    const int khalf = k >> 4, part = (k >> 3) & 1;
    return ((((size_t)(2 * block + g) * 2 + khalf) * 64) + (size_t)(part * 32 + row)) * 8 + (size_t)(k & 7);
}

}  // namespace rt
