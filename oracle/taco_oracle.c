/*
 * taco_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See taco_oracle.h for the parity status.
 *
 * Every function cites the reference lines it restates.  Abbreviations (paths under /root/reference):
 *   FA   = IsaacGymEnvs/isaacgymenvs/tasks/fpv_asymmetry.py
 *   VT   = IsaacGymEnvs/isaacgymenvs/tasks/base/vec_task_asymmetry.py
 *   CTRL = IsaacGymEnvs/isaacgymenvs/tasks/control/
 *   TU   = python/isaacgym/torch_utils.py
 *   TJ   = IsaacGymEnvs/isaacgymenvs/utils/torch_jit_utils.py
 *
 * Arithmetic conventions (they make this file bit-reproducible on any IEEE-754 host, and let the HIP kernel be
 * compared bit-for-bit):
 *   - fp32 throughout, compiled with -ffp-contract=off; an fma appears only where written as fmaf().
 *   - torch elementwise ops are separate roundings; the three places where the reference's CPU build fuses are
 *     restated as such because the golden vectors show it: torch.cross (a1*b2 - a2*b1 as one fused
 *     multiply-subtract), torch.norm (running sum acc = fma(x, x, acc)), everything else unfused and summed
 *     left to right.
 *   - sin/cos/atan2/asin/log are this file's own polynomial implementations (<= ~2 ulp of libm), not libm's:
 *     libm and the GPU's math library differ in the last bit, which would break bit-exact done flags.
 */
#include "taco_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------ constants */
#define PI_F 3.14159265358979323846f
#define TWO_PI_F 6.28318530717958647692f
#define HALF_PI_F 1.57079632679489661923f
#define QUARTER_PI_F 0.78539816339744830962f

/* ------------------------------------------------------------------------------------------------ own math */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float copysign_bits(float mag, float sgn) { return u2f((f2u(mag) & 0x7fffffffu) | (f2u(sgn) & 0x80000000u)); }

/* minimax kernels on |r| <= pi/4 (Cephes single-precision coefficients) */
static inline float sin_kernel(float r) {
    float z = r * r;
    float p = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    return fmaf(p * z, r, r);
}
static inline float cos_kernel(float r) {
    float z = r * r;
    float p = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    return fmaf(p * z, z, fmaf(-0.5f, z, 1.0f));
}
/* sin and cos of x: k = rint(x*2/pi), three-term Cody-Waite reduction, quadrant select.  |x| >= 2^20 or NaN -> NaN. */
static inline void sincos_own(float x, float *s, float *c) {
    if (!(fabsf(x) < 1048576.0f)) { *s = *c = NAN; return; }
    float k = rintf(x * 0.63661977236758134308f);
    float r = fmaf(-k, 1.5703125f, x);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188216e-8f, r);
    float sr = sin_kernel(r), cr = cos_kernel(r);
    int n = (int)k & 3;
    float ss = (n & 1) ? cr : sr;
    float cc = (n & 1) ? sr : cr;
    if (n & 2) ss = -ss;
    if ((n + 1) & 2) cc = -cc;
    *s = ss;
    *c = cc;
}
float orc_sinf(float x) { float s, c; sincos_own(x, &s, &c); return s; }
float orc_cosf(float x) { float s, c; sincos_own(x, &s, &c); return c; }

/* atan2: octant reduction with ONE division, odd minimax polynomial on |t| <= tan(pi/8). */
float orc_atan2f(float y, float x) {
    if (x != x || y != y) return NAN;
    float ax = fabsf(x), ay = fabsf(y);
    float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    float r;
    if (mx == 0.0f) {
        r = 0.0f;
    } else {
        int big = mn > 0.41421356237309503f * mx;
        float num = big ? mn - mx : mn;
        float den = big ? mn + mx : mx;
        float t = num / den;
        float z = t * t;
        float p = fmaf(fmaf(fmaf(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
        r = fmaf(p * z, t, t);
        if (big) r = QUARTER_PI_F + r;
    }
    if (ay > ax) r = HALF_PI_F - r;
    if (f2u(x) & 0x80000000u) r = PI_F - r;
    return copysign_bits(r, y);
}

/* asin on [-1,1] (Cephes scheme: |x| > 0.5 folds through sqrt((1-|x|)/2)); |x| > 1 or NaN -> NaN */
float orc_asinf(float x) {
    float a = fabsf(x);
    if (!(a <= 1.0f)) return NAN;
    int big = a > 0.5f;
    float z = big ? 0.5f * (1.0f - a) : a * a;
    float s = big ? sqrtf(z) : a;
    float p = fmaf(fmaf(fmaf(fmaf(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z,
                   1.6666752422e-1f);
    float r = fmaf(s * z, p, s);
    if (big) r = HALF_PI_F - (r + r);
    return copysign_bits(r, x);
}

/* natural log for normal positive finite x (Cephes logf scheme); used only by the Box-Muller draw */
float orc_logf(float x) {
    uint32_t u = f2u(x);
    int e = (int)((u >> 23) & 0xffu) - 126;
    float m = u2f((u & 0x007fffffu) | 0x3f000000u); /* [0.5, 1) */
    if (m < 0.70710678118654752440f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = fmaf(p, m, -1.1514610310e-1f);
    p = fmaf(p, m, 1.1676998740e-1f);
    p = fmaf(p, m, -1.2420140846e-1f);
    p = fmaf(p, m, 1.4249322787e-1f);
    p = fmaf(p, m, -1.6668057665e-1f);
    p = fmaf(p, m, 2.0000714765e-1f);
    p = fmaf(p, m, -2.4999993993e-1f);
    p = fmaf(p, m, 3.3333331174e-1f);
    float fe = (float)e;
    float yv = (m * z) * p;
    yv = fmaf(-2.12194440e-4f, fe, yv);
    yv = fmaf(-0.5f, z, yv);
    float r = m + yv;
    return fmaf(0.693359375f, fe, r);
}

/* ------------------------------------------------------------------------------------------------ Philox4x32-10 */
void orc_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* 24-bit uniform in [0,1), the same granularity torch.rand has for float32 */
float orc_uniform(uint32_t bits) { return (float)(bits >> 8) * 5.9604644775390625e-8f; }

enum { STREAM_RESET = 1, STREAM_CMD = 2, STREAM_DEPLOY = 3, STREAM_ROTOR = 4, STREAM_OBS = 5 };
/* index of each uniform inside the STREAM_RESET sequence (4 per Philox block) */
enum {
    RU_POS = 0, RU_EULER = 3, RU_LINVEL = 6, RU_ANGVEL = 9, RU_FLIP_SIGN = 12, RU_TGT_XY = 13, RU_TGT_Z = 15, RU_TGT_YAW = 16,
    RU_BAT_E = 17, RU_OPARA = 18, RU_TAU = 23, RU_OMEGA0 = 27, RU_CFCT = 31, RU_DRAG = 33, RU_KT = 35, RU_DELAY = 36,
    RU_COUNT = 37
};

/* ------------------------------------------------------------------------------------------------ small helpers */
/* x / c for a divisor c known in advance, WITHOUT a division: q = x * RN(1/c), then one fma-based correction.
 * For every divisor this file uses it on (0.001, 0.75, 3, 3.3, 6, 100, 1000, 4500, 9000, pi) the result was checked
 * exhaustively (all 2^24 signed mantissas x several binades) to be bit-identical to the IEEE-754 quotient x / c, i.e. to
 * what torch's `tensor / scalar` computes.  It differs from IEEE only where no meaningful state lives: |x| beyond ~1e34
 * (NaN instead of +-inf), quotients in the denormal range, and the sign of a zero quotient (-0/c gives +0).
 * The HIP kernel uses the same three operations. */
static inline float div_const(float x, float c, float rc) {
    float q = x * rc;
    float r = fmaf(-q, c, x);
    return fmaf(r, rc, q);
}
#define DIVC(x, c) div_const((x), (c), 1.0f / (c))
/* exhaustive check of div_const for a run-time divisor (the sim dt): one binade of x, both signs; scale invariance of
 * the three operations covers the other binades of the normal range */
static int div_const_is_exact(float c) {
    static float cached_c = 0.0f; /* benign race: every thread computes the same answer */
    static int cached_ok = 0;
    if (c == cached_c) return cached_ok;
    const float rc = 1.0f / c;
    for (uint32_t m = 0; m < (1u << 23); ++m) {
        float x = u2f(0x3f800000u | m);
        if (div_const(x, c, rc) != x / c || div_const(-x, c, rc) != -x / c) { cached_ok = 0; cached_c = c; return 0; }
    }
    cached_ok = 1;
    cached_c = c;
    return 1;
}
/* torch.clamp / torch.clip: min(max(x, lo), hi), NaN in x propagates */
static inline float clampf(float x, float lo, float hi) {
    float t = (x < lo) ? lo : x;
    return (t > hi) ? hi : t;
}
/* torch.norm(p=2) over 2 / 3 elements as the reference's CPU build evaluates it: acc = fma(x, x, acc) */
static inline float norm2(float a, float b) { return sqrtf(fmaf(b, b, a * a)); }
static inline float norm3(float a, float b, float c) { return sqrtf(fmaf(c, c, fmaf(b, b, a * a))); }
/* torch.cross component a1*b2 - a2*b1 as evaluated by the reference's CPU build (fused multiply-subtract) */
static inline float cross_c(float a1, float b2, float a2, float b1) { return fmaf(a1, b2, -(a2 * b1)); }

/* TU:19-40 quat_mul (xyzw) */
static inline void quat_mul(const float a[4], const float b[4], float o[4]) {
    float x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3];
    float x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
    float ww = (z1 + x1) * (x2 + y2);
    float yy = (w1 - y1) * (w2 + z2);
    float zz = (w1 + y1) * (w2 - z2);
    float xx = ww + yy + zz;
    float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
    float w = qq - ww + (z1 - y1) * (y2 - z2);
    float x = qq - xx + (x1 + w1) * (x2 + w2);
    float y = qq - yy + (w1 - x1) * (y2 + z2);
    float z = qq - zz + (z1 + y1) * (w2 - x2);
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
/* TU:84-88 quat_conjugate */
static inline void quat_conj(const float q[4], float o[4]) { o[0] = -q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = q[3]; }
/* TU:58-68 quat_rotate(q, v) */
static inline void quat_rotate(const float q[4], const float v[3], float o[3]) {
    float qw = q[3];
    float s = 2.0f * (qw * qw) - 1.0f;
    float cx = cross_c(q[1], v[2], q[2], v[1]);
    float cy = cross_c(q[2], v[0], q[0], v[2]);
    float cz = cross_c(q[0], v[1], q[1], v[0]);
    float dot = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
    float cr[3] = {cx, cy, cz};
    for (int i = 0; i < 3; ++i) {
        float a = v[i] * s;
        float b = cr[i] * qw * 2.0f;
        float c = q[i] * dot * 2.0f;
        o[i] = a + b + c;
    }
}
/* quat_rotate(quat_conjugate(q), v) -- every body-frame quantity of FA:349-360 */
static inline void rotate_inv(const float q[4], const float v[3], float o[3]) {
    float c[4];
    quat_conj(q, c);
    quat_rotate(c, v, o);
}
/* TU:175-196 get_euler_xyz_v1 */
static inline void euler_xyz_v1(const float q[4], float rpy[3]) {
    float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    float sinr_cosp = 2.0f * (qw * qx + qy * qz);
    float cosr_cosp = qw * qw - qx * qx - qy * qy + qz * qz;
    rpy[0] = orc_atan2f(sinr_cosp, cosr_cosp);
    float sinp = 2.0f * (qw * qy - qz * qx);
    if (fabsf(sinp) >= 1.0f) { /* copysign(pi/2, sinp) = |pi/2| * sign(sinp), TU:147-150 */
        rpy[1] = HALF_PI_F * (sinp > 0.0f ? 1.0f : (sinp < 0.0f ? -1.0f : 0.0f));
    } else {
        rpy[1] = orc_asinf(sinp); /* NaN sinp lands here and stays NaN, like torch.where + asin */
    }
    float siny_cosp = 2.0f * (qw * qz + qx * qy);
    float cosy_cosp = qw * qw + qx * qx - qy * qy - qz * qz;
    rpy[2] = orc_atan2f(siny_cosp, cosy_cosp);
}
/* TU:199-213 quat_from_euler_xyz */
static inline void quat_from_euler(float roll, float pitch, float yaw, float q[4]) {
    float cy, sy, cr, sr, cp, sp;
    sincos_own(yaw * 0.5f, &sy, &cy);
    sincos_own(roll * 0.5f, &sr, &cr);
    sincos_own(pitch * 0.5f, &sp, &cp);
    q[3] = cy * cr * cp + sy * sr * sp;
    q[0] = cy * sr * cp - sy * cr * sp;
    q[1] = cy * cr * sp + sy * sr * cp;
    q[2] = sy * cr * cp - cy * sr * sp;
}
/* TJ:145-164 quat_diff_rad */
static inline float quat_diff_rad(const float a[4], const float b[4]) {
    float bc[4], m[4];
    quat_conj(b, bc);
    quat_mul(a, bc, m);
    float n = norm3(m[0], m[1], m[2]);
    n = (n > 1.0f) ? 1.0f : n; /* clamp(max=1.0) */
    return 2.0f * orc_asinf(n);
}
/* TJ:389-416 quaternion_to_matrix (xyzw input, row-major 3x3) */
static inline void quat_to_matrix(const float q[4], float m[9]) {
    float i = q[0], j = q[1], k = q[2], r = q[3];
    float two_s = 2.0f / (((i * i + j * j) + k * k) + r * r);
    m[0] = 1.0f - two_s * (j * j + k * k);
    m[1] = two_s * (i * j - k * r);
    m[2] = two_s * (i * k + j * r);
    m[3] = two_s * (i * j + k * r);
    m[4] = 1.0f - two_s * (i * i + k * k);
    m[5] = two_s * (j * k - i * r);
    m[6] = two_s * (i * k - j * r);
    m[7] = two_s * (j * k + i * r);
    m[8] = 1.0f - two_s * (i * i + j * j);
}
/* TU:216-219 torch_rand_float: (upper - lower) * u + lower with the Python doubles rounded to fp32 when they meet the tensor */
static inline float rand_float(double lower, double upper, float u) { return (float)(upper - lower) * u + (float)lower; }

/* ------------------------------------------------------------------------------------------------ sub-models */
/* CTRL/angvel_control.py:67-88 (gains :17-60) */
static inline void pid_axis(float dt, float rdt, float kp, float des, float cur, float *prev, float *integ, float *out) {
    const float ki = 0.0f, kd = 0.5f, kf = 0.0f, fg = 0.4f;
    float e = clampf(des - cur, -400.0f, 400.0f);
    float pv = (*prev == 0.0f) ? e : *prev;
    float P = kp * e;
    float I = clampf(*integ + e * dt, -500.0f, 500.0f);
    float I_term = ki * I;
    float deriv = (rdt != 0.0f) ? div_const(e - pv, dt, rdt) : (e - pv) / dt; /* rdt = 1/dt when div_const is exact for dt */
    float D = clampf(kd * deriv, -150.0f, 150.0f);
    float FF = kf * des;
    *out = fg * (P + I_term + D + FF);
    *integ = I;
    *prev = e;
}
static inline void pid_step(float dt, float rdt, const float des[3], const float cur[3], float prev[3], float integ[3], float out[3]) {
    pid_axis(dt, rdt, 27.5f, des[0], cur[0], &prev[0], &integ[0], &out[0]);
    pid_axis(dt, rdt, 50.0f, des[1], cur[1], &prev[1], &integ[1], &out[1]);
    pid_axis(dt, rdt, 200.0f, des[2], cur[2], &prev[2], &integ[2], &out[2]);
}
/* CTRL/fpv_dynamics.py:35-46 control_allocator (weight :28-33); u[3] is clipped in place like the reference */
static inline void allocator(float u[4], float thr[4]) {
    u[3] = clampf(u[3], -u[0] / 2.0f, u[0] / 2.0f);
    static const float W[4][4] = {{1, -1, 1, -1}, {1, -1, -1, 1}, {1, 1, -1, -1}, {1, 1, 1, 1}};
    float f[4];
    for (int r = 0; r < 4; ++r) f[r] = ((u[0] * W[r][0] + u[1] * W[r][1]) + u[2] * W[r][2]) + u[3] * W[r][3];
    float mx = f[0] - 1000.0f;
    for (int r = 1; r < 4; ++r) { float t = f[r] - 1000.0f; mx = (t > mx || t != t) ? t : mx; }
    float ex = (mx < 0.0f) ? 0.0f : mx; /* clamp(min=0) */
    for (int r = 0; r < 4; ++r) thr[r] = clampf(f[r] - ex, 100.0f, 1000.0f);
}
/* FA:614  P_m = sum_i 400 * (omega_i * 2 * pi / 4500) ** 3 */
static inline float mech_power(const float om[4]) {
    float acc = 0.0f;
    for (int i = 0; i < 4; ++i) {
        float b = DIVC(om[i] * 2.0f * PI_F, 4500.0f);
        float t = 400.0f * ((b * b) * b);
        acc = (i == 0) ? t : acc + t;
    }
    return acc;
}
/* CTRL/battery_dynamics.py:47-75 (constants :19-30) */
static inline float battery_step(int enabled, float dt, float Pm, float *E, float *u1, float *t) {
    const float a0 = 4.35f;
    if (!enabled) return a0 * 6.0f;
    *t = *t + dt;
    float p_c = DIVC(DIVC(Pm, 0.75f), 9000.0f);
    *E = *E + p_c * dt;
    float P_avg = *E / *t;
    float r0_ = 0.0015778f + -7.7608e-5f * P_avg + (float)(0.0069498 * 1500.0);
    float r0 = (r0_ > 4.5f) ? r0_ : 4.5f;
    float Ec = *E;
    float u0 = a0 + -0.1102178f * Ec + 0.0103368f * (Ec * Ec) + -4.3778e-4f * ((Ec * Ec) * Ec);
    float u1_dot = DIVC(0.00104846f * p_c - *u1, 3.3f);
    *u1 = *u1 + u1_dot * dt;
    float d = u0 - *u1;
    float rad = d * d - 4.0f * r0 * p_c;
    return 0.5f * (d + sqrtf(rad)) * 6.0f;
}
/* CTRL/thrust_dynamics.py:98-104 (throttle_voltage2omega :52-66, omega_compute :80-86; delay depth 1 = pass-through) */
static inline void rotor_step(float V, const float thr[4], const float tau[4], const float p[5], float om[4]) {
    float y = DIVC(V - 23.0f, 3.0f);
    for (int i = 0; i < 4; ++i) {
        float x = DIVC(thr[i], 1000.0f);
        float target = (p[0] * 1.0f + p[1] * x + p[2] * y + p[3] * (x * x) + p[4] * x * y) * 100.0f;
        /* `self.sample_time / self.response_time`: Python float / Tensor is Tensor.__rtruediv__ = reciprocal() * scalar */
        om[i] = om[i] + (1.0f / tau[i]) * 0.001f * (target - om[i]);
    }
}
/* CTRL/thrust_dynamics.py:173-199 AeroDynamics.sim_process (body torque is zero) */
static inline void aero_step(float cf, float ct, float dx, float dy, float kt, const float vb[3], const float om[4], float rf[4],
                             float rt[4], float bf[3]) {
    for (int i = 0; i < 4; ++i) {
        rf[i] = cf * om[i] * om[i];
        rt[i] = ct * rf[i];
    }
    bf[0] = dx * vb[0];
    bf[1] = dy * vb[1];
    float vxy = norm2(vb[0], vb[1]);
    bf[2] = kt * vxy * vxy;
}
/* CTRL/fpv_dynamics.py:48-56: real rotor order -> sim order [2,3,0,1], torque of sim rotors 0 and 2 negated */
static inline void real2sim(const float f[4], const float t[4], float fs[4], float ts[4]) {
    fs[0] = f[2]; fs[1] = f[3]; fs[2] = f[0]; fs[3] = f[1];
    ts[0] = -t[2]; ts[1] = t[3]; ts[2] = -t[0]; ts[3] = t[1];
}
/* die / reset flags shared by the three rewards (CTRL/task_reward.py:39-45) */
static inline int64_t done_flag(float z, float pos_dist, int64_t progress, float max_len) {
    int64_t die = 0;
    if (z < 0.1f) die = 1;
    if (pos_dist > 10.0f) die = 1;
    return ((float)progress >= max_len - 1.0f) ? 1 : die;
}
static inline float two_level(float d2) { return 1.0f / (1.0f + d2) + 1.0f / (1.0f + 10.0f * d2); }
/* CTRL/task_reward.py:20-47 */
static inline float reward_pos(const float rpb[3], const float pos[3], const float q[4], const float qt[4], int64_t prog,
                               float max_len, int64_t *reset) {
    float d = norm3(rpb[0], rpb[1], rpb[2]);
    float l0 = 1.0f / (1.0f + d * d), l1 = 1.0f / (1.0f + 10.0f * d * d);
    float pr = l0 + l1;
    float qd = quat_diff_rad(q, qt);
    float r0 = 1.0f / (1.0f + qd * qd), r1 = 1.0f / (1.0f + 10.0f * qd * qd);
    float rr = r0 + r1;
    *reset = done_flag(pos[2], d, prog, max_len);
    return DIVC(pr * rr, 100.0f);
}
/* CTRL/task_reward.py:50-104 */
static inline float reward_rotate(const float rp[3], const float rv[3], const float pos[3], const float q[4], const float cmd[2],
                                  int64_t prog, float max_len, int64_t *reset) {
    const float r = 1.2f;
    float v = cmd[1];
    float nx[3] = {-rp[0], -rp[1], 0.0f};
    float nn = norm3(nx[0], nx[1], nx[2]) + 1e-8f;
    nx[0] = nx[0] / nn; nx[1] = nx[1] / nn; nx[2] = nx[2] / nn;
    /* new_y = cross((0,0,1), new_x) */
    float ny[3] = {cross_c(0.0f, nx[2], 1.0f, nx[1]), cross_c(1.0f, nx[0], 0.0f, nx[2]), cross_c(0.0f, nx[1], 0.0f, nx[0])};
    float ynn = norm3(ny[0], ny[1], ny[2]) + 1e-8f;
    ny[0] = ny[0] / ynn; ny[1] = ny[1] / ynn; ny[2] = ny[2] / ynn;
    float hori = norm2(rp[0], rp[1]) - r;
    float vert = fabsf(rp[2]);
    float pd = sqrtf(hori * hori + vert * vert);
    float pr = 1.0f / (1.0f + pd * pd) + 1.0f / (1.0f + 10.0f * pd * pd);
    float normal = (rv[0] * nx[0] + rv[1] * nx[1]) + rv[2] * nx[2];
    float tang = (rv[0] * ny[0] + rv[1] * ny[1]) + rv[2] * ny[2];
    float ld = norm3(normal - 0.0f, tang - v, rv[2] - 0.0f);
    float lr = 1.0f / (1.0f + ld * ld) + 1.0f / (1.0f + 10.0f * ld * ld);
    float m[9];
    quat_to_matrix(q, m);
    float hx = m[0], hy = m[3]; /* first column = heading */
    float dd = 1.0f + (nx[0] * hx + nx[1] * hy) / norm2(hx, hy);
    float dr = 1.0f / (1.0f + dd * dd) + 1.0f / (1.0f + 10.0f * dd * dd);
    *reset = done_flag(pos[2], pd, prog, max_len);
    return DIVC(pr * lr * dr, 100.0f);
}
/* CTRL/task_reward.py:107-143 */
static inline float reward_flip(const float rpb[3], const float relq[4], const float pos[3], const float cmd[2], int64_t prog,
                                float max_len, int64_t *reset) {
    float d = norm3(rpb[0], rpb[1], rpb[2]);
    float pr = 1.0f / (1.0f + 1.0f * d) + 1.0f / (1.0f + 10.0f * d);
    float m[9];
    quat_to_matrix(relq, m);
    float xt = 1.0f - m[0];
    float xr = 1.0f / (1.0f + 10.0f * xt);
    float cd = DIVC(cmd[1] / 2.0f, PI_F);
    float cr = 1.0f / (1.0f + cd * cd) + 1.0f / (1.0f + 10.0f * cd * cd);
    *reset = done_flag(pos[2], d, prog, max_len);
    return DIVC(pr * xr * cr, 100.0f);
}

/* ------------------------------------------------------------------------------------------------ row I: integrate
 * Replaces gym.simulate(sim) (VT:313) for ONE free rigid body.  NOT in the reference's source (closed PhysX):
 * "parity unpinned".  Scheme (PhysX public SDK): `substeps` sub-iterations of h = dt/substeps, semi-implicit Euler, with
 * the body-frame wrench held constant (gym.apply_rigid_body_force_tensors(..., LOCAL_SPACE), FA:633-635):
 *     b += h J^-1 (tau - b x J b)          body rates b; start value = the body-frame angular velocity of row C
 *     v += h (R(q) F / m + g);  p += h v    R(q) F by the quaternion sandwich  F + w t + qv x t,  t = 2 qv x F
 *     q <- normalize(q (x) exp(h/2 b))      closed-form quaternion update; a body-frame rate multiplies on the right
 * The body rates are CARRIED in the body frame from one simulate() to the next inside a step(): they are taken from the
 * root state once, at the start of the step (row C's formula, quat_rotate(conj q, w), FA:350), feed the rate PID of every
 * substep directly, and the root state's world-frame angular velocity is rebuilt once, after the 10th simulate():
 * w = R(q) b.  (The reference re-derives the body rates from the world-frame root tensor before every PID call; going
 * through the world frame and back 10 times per step costs ~50 instructions per substep and only adds rounding noise.)
 * exp(): Taylor polynomials in a2 = (h/2 |b|)^2 while a2 <= 0.25, sqrt/sincos beyond; normalisation: one Newton step
 * from 1 (error 3/8 (n2-1)^2, far below fp32 resolution) while |n2 - 1| <= 1e-3, else 1/sqrt(n2).
 * Every operation below is written out (fma where fused) -- the HIP kernel repeats it verbatim. */
typedef struct {
    float h, half_h, inv_m, g, J[3], hJi[3];
    int substeps;
} integ_par;

static inline void quat_sandwich(const float q[4], const float u[3], float out[3]) {
    float tx = fmaf(q[1], u[2], -(q[2] * u[1])), ty = fmaf(q[2], u[0], -(q[0] * u[2])), tz = fmaf(q[0], u[1], -(q[1] * u[0]));
    tx = tx + tx; ty = ty + ty; tz = tz + tz;
    out[0] = fmaf(q[1], tz, fmaf(-q[2], ty, fmaf(q[3], tx, u[0])));
    out[1] = fmaf(q[2], tx, fmaf(-q[0], tz, fmaf(q[3], ty, u[1])));
    out[2] = fmaf(q[0], ty, fmaf(-q[1], tx, fmaf(q[3], tz, u[2])));
}

static inline void integrate_substep(const integ_par *P, float p[3], float q[4], float v[3], float wb[3],
                                     const float F[3], const float tau[3]) {
    float b0 = wb[0], b1 = wb[1], b2 = wb[2];
    for (int it = 0; it < P->substeps; ++it) {
        /* Euler's equations in the body frame */
        float L0 = P->J[0] * b0, L1 = P->J[1] * b1, L2 = P->J[2] * b2;
        float g0 = fmaf(b1, L2, -(b2 * L1));
        float g1 = fmaf(b2, L0, -(b0 * L2));
        float g2 = fmaf(b0, L1, -(b1 * L0));
        b0 = fmaf(P->hJi[0], tau[0] - g0, b0);
        b1 = fmaf(P->hJi[1], tau[1] - g1, b1);
        b2 = fmaf(P->hJi[2], tau[2] - g2, b2);
        /* linear */
        float RF[3];
        quat_sandwich(q, F, RF);
        v[0] = fmaf(P->h, RF[0] * P->inv_m, v[0]);
        v[1] = fmaf(P->h, RF[1] * P->inv_m, v[1]);
        v[2] = fmaf(P->h, fmaf(RF[2], P->inv_m, P->g), v[2]);
        p[0] = fmaf(P->h, v[0], p[0]);
        p[1] = fmaf(P->h, v[1], p[1]);
        p[2] = fmaf(P->h, v[2], p[2]);
        /* attitude */
        float w2 = fmaf(b2, b2, fmaf(b1, b1, b0 * b0));
        float A2 = (P->half_h * P->half_h) * w2;
        float sp = fmaf(fmaf(fmaf(fmaf(2.7557319224e-6f, A2, -1.9841269841e-4f), A2, 8.3333333333e-3f), A2, -1.6666666667e-1f), A2, 1.0f);
        float c = fmaf(fmaf(fmaf(fmaf(2.4801587302e-5f, A2, -1.3888888889e-3f), A2, 4.1666666667e-2f), A2, -0.5f), A2, 1.0f);
        float k = P->half_h * sp;
        if (!(A2 <= 0.25f)) {
            if (A2 == A2) {
                float wn = sqrtf(w2), sn;
                sincos_own(P->half_h * wn, &sn, &c);
                k = sn / wn;
            } else {
                k = c = NAN;
            }
        }
        float dx = b0 * k, dy = b1 * k, dz = b2 * k;
        float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
        /* Hamilton product q (x) (dx, dy, dz, c) */
        float nx = fmaf(qw, dx, fmaf(qx, c, fmaf(qy, dz, -(qz * dy))));
        float ny = fmaf(qw, dy, fmaf(qy, c, fmaf(qz, dx, -(qx * dz))));
        float nz = fmaf(qw, dz, fmaf(qz, c, fmaf(qx, dy, -(qy * dx))));
        float nw = fmaf(qw, c, -fmaf(qx, dx, fmaf(qy, dy, qz * dz)));
        float n2 = fmaf(nw, nw, fmaf(nz, nz, fmaf(ny, ny, nx * nx)));
        float inv = fmaf(-0.5f, n2, 1.5f);
        if (!(fabsf(n2 - 1.0f) <= 1e-3f)) inv = 1.0f / sqrtf(n2);
        q[0] = nx * inv; q[1] = ny * inv; q[2] = nz * inv; q[3] = nw * inv;
    }
    wb[0] = b0; wb[1] = b1; wb[2] = b2;
}

/* ------------------------------------------------------------------------------------------------ environment */
typedef struct {
    float p[3], q[4], v[3], w[3];
    float pt[3], qt[4];
    float rpy_old[3], rpy_cont[3];
    float pid_prev[3], pid_int[3];
    float bat_E, bat_u1, bat_t, bat_V;
    float omega[4];
    float act[4], act_old[4];
    float cmd[2], flip_radian;
    float tau[4], opara[5], cf, ct, dx, dy, kt;
    int32_t progress, delay_len;
    float ring[ORC_RING_SLOTS][4]; /* actions_remained_buffer[env] transposed: [slot][channel] (FA:189) */
} env_state;

struct orc_env {
    orc_cfg cfg;
    env_state *s;
    int64_t step_count;
    int threads;
    int world_rate_roundtrip; /* orc_set_world_rate_roundtrip: see step_env */
    int mix_n1, mix_n2;
    integ_par ip;
    float rdt; /* 1/dt if div_const is exact for this dt, else 0 (true division is used) */
};

static void derive(orc_env *e) {
    const orc_cfg *c = &e->cfg;
    e->ip.substeps = c->substeps;
    e->ip.h = (float)(c->dt / (double)c->substeps);
    e->ip.half_h = (float)(0.5 * (c->dt / (double)c->substeps));
    e->ip.inv_m = (float)(1.0 / c->mass);
    e->ip.g = (float)c->gravity_z;
    for (int i = 0; i < 3; ++i) {
        e->ip.J[i] = (float)c->inertia[i];
        e->ip.hJi[i] = (float)((c->dt / (double)c->substeps) / c->inertia[i]);
    }
    e->rdt = div_const_is_exact((float)c->dt) ? 1.0f / (float)c->dt : 0.0f;
    /* FA:924-925: n1 = int(N / 3 * 1), n2 = int(N / 3 * 2) in Python doubles */
    e->mix_n1 = (int)((double)c->num_envs_global / 3 * 1);
    e->mix_n2 = (int)((double)c->num_envs_global / 3 * 2);
}

int orc_create(const orc_cfg *cfg, orc_env **out) {
    if (!cfg || !out || cfg->num_envs <= 0 || cfg->control_freq_inv != 10 || cfg->substeps < 1 || cfg->len_obs < 1 ||
        cfg->len_states < 1 || cfg->task_mode < 0 || cfg->task_mode > 3 || cfg->delay_time < 0 || cfg->delay_time > 90)
        return -1;
    orc_env *e = (orc_env *)calloc(1, sizeof(orc_env));
    e->cfg = *cfg;
    e->s = (env_state *)calloc((size_t)cfg->num_envs, sizeof(env_state));
    e->threads = 1;
    for (int i = 0; i < cfg->num_envs; ++i) {
        env_state *s = &e->s[i];
        /* actor creation pose (FA:264-266): both actors at (0,0,4), identity attitude, at rest */
        s->p[2] = 4.0f; s->q[3] = 1.0f; s->pt[2] = 4.0f; s->qt[3] = 1.0f;
        /* sub-model constructors: CTRL/thrust_dynamics.py:37,46,156-159 */
        for (int k = 0; k < 4; ++k) s->tau[k] = (float)cfg->rotor_response_time;
        s->opara[0] = 0.0f; s->opara[1] = 12.9466f; s->opara[2] = 0.1872f; s->opara[3] = -5.1220f; s->opara[4] = 0.5906f;
        s->cf = 1.13e-05f; s->ct = 0.05f; s->dx = -0.386f; s->dy = -0.53f; s->kt = 0.009f;
        s->delay_len = cfg->delay_time; /* FA:193 (the random branch :191 is re-drawn at the first reset anyway) */
    }
    derive(e);
    *out = e;
    return 0;
}
void orc_destroy(orc_env *e) { if (e) { free(e->s); free(e); } }
void orc_set_difficulty(orc_env *e, double d) { e->cfg.difficulty = d; }
void orc_set_threads(orc_env *e, int n) { e->threads = n < 1 ? 1 : n; }
/* Data flow of the angular rate between substeps.  0 (default, = the product kernel): row I carries the BODY rates from one gym.simulate
 * to the next and only the 10th substep writes the root state's world-frame angular velocity.  1: every substep goes through the root
 * state, as the reference does (simulate leaves w_world = R(q) b, the next refresh_state computes b = R(q)^T w_world, FA:350) -- two
 * rotations and two roundings more per substep.  Mode 1 is what tests/golden/make_glue_golden.py reproduces when it drives the
 * reference's own fpv_asymmetry.py with row I plugged in as gym.simulate; it exists so that the glue (everything but row I) can be
 * compared with that run bit for bit. */
void orc_set_world_rate_roundtrip(orc_env *e, int on) { e->world_rate_roundtrip = on ? 1 : 0; }
int64_t orc_step_count(const orc_env *e) { return e->step_count; }
int orc_num_envs(const orc_env *e) { return e->cfg.num_envs; }
void orc_set_step_count(orc_env *e, int64_t n) { e->step_count = n; }

typedef struct { uint64_t seed; uint32_t gid, step, stream; uint32_t blk_id; uint32_t blk[4]; int have; } draw_ctx;
static inline float draw_u(draw_ctx *d, int idx) {
    uint32_t b = (uint32_t)idx >> 2;
    if (!d->have || d->blk_id != b) { orc_philox(d->seed, d->gid, d->step, d->stream, b, d->blk); d->blk_id = b; d->have = 1; }
    return orc_uniform(d->blk[idx & 3]);
}
/* round(N(0,1)) clamped to [-lim, lim], lim in {1,3}: same distribution as torch.round(torch.normal(0,1)) clamped
 * (FA:324, FA:576), drawn by inverse CDF on one uniform: thresholds Phi(-2.5), Phi(-1.5), Phi(-0.5), Phi(0.5), ... */
static inline int rounded_normal(float u, int lim) {
    static const float T[6] = {0.0062096653f, 0.0668072013f, 0.3085375387f, 0.6914624613f, 0.9331927987f, 0.9937903347f};
    int k = -3;
    for (int i = 0; i < 6; ++i) k += (u >= T[i]);
    return k < -lim ? -lim : (k > lim ? lim : k);
}

static inline int task_group(const orc_env *e, int gid) {
    if (e->cfg.task_mode != ORC_TASK_MIX) return e->cfg.task_mode;
    return gid < e->mix_n1 ? ORC_TASK_POS : (gid < e->mix_n2 ? ORC_TASK_ROTATE : ORC_TASK_FLIP);
}

/* reset_idx for one env: FA:475-517 in its call order (copter -> controller -> env -> target) */
static void reset_env(const orc_env *e, env_state *s, int grp, draw_ctx *D) {
    const orc_cfg *c = &e->cfg;
    const uint32_t fl = c->flags;
    const double d = c->difficulty;
    const float df = (float)d;
    const int mix = c->task_mode == ORC_TASK_MIX;
    D->stream = STREAM_RESET; D->have = 0;
    /* ---- reset_copter_idx: FA:725-756 (pos), :783-812 (rotate), :850-884 (flip), :981-1056 (mix uses pos-style ranges) */
    if (grp == ORC_TASK_FLIP && !mix) {
        if (fl & ORC_F_RANDOM_COPTER_POS) {
            s->p[0] = rand_float(-0.5 - 1.5 * d, 0.5 + 1.5 * d, draw_u(D, RU_POS + 0));
            s->p[1] = rand_float(-0.5 - 1.5 * d, 0.5 + 1.5 * d, draw_u(D, RU_POS + 1));
            s->p[2] = 3.0f + df * rand_float(-2, 2, draw_u(D, RU_POS + 2));
        } else {
            s->p[0] = rand_float(-0.5, 0.5, draw_u(D, RU_POS + 0));
            s->p[1] = rand_float(-0.5, 0.5, draw_u(D, RU_POS + 1));
            s->p[2] = 3.0f;
        }
    } else if (grp == ORC_TASK_ROTATE && !mix && !(fl & ORC_F_RANDOM_COPTER_POS)) {
        s->p[0] = rand_float(-0.5, 0.5, draw_u(D, RU_POS + 0));
        s->p[1] = rand_float(-0.5, 0.5, draw_u(D, RU_POS + 1));
        s->p[2] = 2.5f;
    } else if (fl & ORC_F_RANDOM_COPTER_POS) {
        s->p[0] = rand_float(-2, 2, draw_u(D, RU_POS + 0));
        s->p[1] = rand_float(-2, 2, draw_u(D, RU_POS + 1));
        s->p[2] = 2.5f + rand_float(-2, 2, draw_u(D, RU_POS + 2));
    } else {
        s->p[0] = 0.0f; s->p[1] = 0.0f; s->p[2] = 2.5f;
    }
    if (fl & ORC_F_RANDOM_COPTER_QUAT) {
        /* rand_quat (FA:698-704): the "pitch" draw lands in quat_from_euler_xyz's roll slot, "roll" in pitch; flip: (pi,0,0) */
        double l1 = (grp == ORC_TASK_FLIP) ? 0.0 : M_PI;
        float a = rand_float(-M_PI, M_PI, draw_u(D, RU_EULER + 0));
        float b = rand_float(-l1, l1, draw_u(D, RU_EULER + 1));
        float cc = rand_float(-l1, l1, draw_u(D, RU_EULER + 2));
        quat_from_euler(a, b, cc, s->q);
    } else {
        s->q[0] = s->q[1] = s->q[2] = 0.0f; s->q[3] = 1.0f;
    }
    if (grp == ORC_TASK_FLIP) {
        if (fl & ORC_F_RANDOM_COPTER_VEL) {
            for (int k = 0; k < 3; ++k) s->v[k] = rand_float(-3 * d, 3 * d, draw_u(D, RU_LINVEL + k));
            s->w[0] = 10.0f * (draw_u(D, RU_FLIP_SIGN) < 0.5f ? -1.0f : 1.0f); /* w[1], w[2] keep their values (FA:876, :1047) */
        } else {
            s->v[0] = s->v[1] = s->v[2] = 0.0f;
            if (mix) s->w[0] = s->w[1] = s->w[2] = 0.0f; /* FA:1050; standalone FpvFlip leaves angvel untouched (FA:877-878) */
        }
    } else if (fl & ORC_F_RANDOM_COPTER_VEL) {
        for (int k = 0; k < 3; ++k) s->v[k] = 3.0f * rand_float(-1.0, 1.0, draw_u(D, RU_LINVEL + k));
        for (int k = 0; k < 3; ++k) s->w[k] = 3.0f * rand_float(-1.0, 1.0, draw_u(D, RU_ANGVEL + k));
    } else {
        for (int k = 0; k < 3; ++k) s->v[k] = s->w[k] = 0.0f;
    }
    euler_xyz_v1(s->q, s->rpy_old);
    for (int k = 0; k < 3; ++k) s->rpy_cont[k] = s->rpy_old[k];
    /* ---- reset_controller_idx FA:550-558 */
    for (int k = 0; k < 3; ++k) s->pid_prev[k] = s->pid_int[k] = 0.0f; /* CTRL/angvel_control.py:90-94 */
    s->bat_u1 = 0.0f; s->bat_E = 0.0f; s->bat_t = 0.0f;                 /* CTRL/battery_dynamics.py:38-45 */
    if (fl & ORC_F_RANDOM_VOLTAGE) s->bat_E = rand_float(0, 2.2, draw_u(D, RU_BAT_E));
    /* CTRL/thrust_dynamics.py:109-148 */
    static const float opara_init[5] = {0.0f, 12.9466f, 0.1872f, -5.1220f, 0.5906f};
    for (int k = 0; k < 5; ++k)
        s->opara[k] = (fl & ORC_F_RANDOM_ROTORDYNAMIC_COE) ? opara_init[k] * rand_float(1 - 0.05 * d, 1 + 0.05 * d, draw_u(D, RU_OPARA + k))
                                                            : opara_init[k];
    for (int k = 0; k < 4; ++k) {
        if (fl & ORC_F_ROTOR_RESPONSE) {
            if (fl & ORC_F_RANDOM_ROTOR_RESPONSE)
                s->tau[k] = rand_float(c->rotor_response_time - 0.001, c->rotor_response_time + 0.001, draw_u(D, RU_TAU + k));
            else
                s->tau[k] = (float)c->rotor_response_time * 1.0f;
        } else {
            s->tau[k] = 0.001f * 1.0f;
        }
        s->omega[k] = (fl & ORC_F_RANDOM_ROTOR_SPEED) ? rand_float(0, 400, draw_u(D, RU_OMEGA0 + k)) : 0.0f;
    }
    if (fl & ORC_F_RANDOM_AERODYNAMIC_COE) { /* CTRL/thrust_dynamics.py:201-210 (no-op when the flag is off) */
        s->cf = 1.13e-05f * rand_float(1 - 0.05 * d, 1 + 0.05 * d, draw_u(D, RU_CFCT + 0));
        s->ct = 0.05f * rand_float(1 - 0.05 * d, 1 + 0.05 * d, draw_u(D, RU_CFCT + 1));
        s->dx = -0.386f * rand_float(1 - 0.05 * d, 1 + 0.05 * d, draw_u(D, RU_DRAG + 0));
        s->dy = -0.53f * rand_float(1 - 0.05 * d, 1 + 0.05 * d, draw_u(D, RU_DRAG + 1));
        s->kt = 0.009f * rand_float(1 - 0.05 * d, 1 + 0.05 * d, draw_u(D, RU_KT));
    }
    /* ---- reset_env_idx FA:560-581 */
    s->bat_V = 0.0f;
    for (int k = 0; k < 4; ++k) s->act[k] = s->act_old[k] = 0.0f;
    memset(s->ring, 0, sizeof(s->ring));
    if (fl & ORC_F_RANDOM_DELAY_TIME) {
        int L = c->delay_time - rounded_normal(draw_u(D, RU_DELAY), 3);
        s->delay_len = L < 0 ? 0 : L;
    } else {
        s->delay_len = c->delay_time;
    }
    /* ---- reset_target_idx FA:523-548 */
    if (fl & ORC_F_RANDOM_TARGET_POS) {
        s->pt[0] = df * rand_float(-2, 2, draw_u(D, RU_TGT_XY + 0));
        s->pt[1] = df * rand_float(-2, 2, draw_u(D, RU_TGT_XY + 1));
        s->pt[2] = 3.0f + df * rand_float(-2, 2, draw_u(D, RU_TGT_Z));
    } else {
        s->pt[0] = 0.0f; s->pt[1] = 0.0f; s->pt[2] = 3.0f;
    }
    float yaw = (fl & ORC_F_RANDOM_TARGET_YAW) ? rand_float(-M_PI, M_PI, draw_u(D, RU_TGT_YAW)) : 0.0f;
    quat_from_euler(0.0f, 0.0f, yaw, s->qt);
}

/* reset_command_idx for one env whose command is due (reset this step, or progress == 500): FA:758-759 (pos),
 * :814-821 (rotate), :886-917 (flip), :1058-1112 (mix) */
static void reset_command(const orc_env *e, env_state *s, int grp, int is_reset, int at_time_index, draw_ctx *D) {
    D->stream = STREAM_CMD; D->have = 0;
    if (grp == ORC_TASK_POS) {
        s->cmd[0] = 0.0f; s->cmd[1] = 0.0f;
    } else if (grp == ORC_TASK_ROTATE) {
        s->cmd[0] = 1.0f;
        s->cmd[1] = (e->cfg.flags & ORC_F_RANDOM_COMMAND) ? rand_float(-6, 6, draw_u(D, 0)) : 1.0f;
    } else {
        if (at_time_index) { /* FA:888-901: add 2*pi*{-3..3} with probabilities 1/8,1/8,1/8,2/8,1/8,1/8,1/8 */
            float u = draw_u(D, 1);
            float t = 0.0f;
            if (u < 1.0f / 8) t = -3.0f;
            if (u >= 1.0f / 8 && u < 2.0f / 8) t = -2.0f;
            if (u >= 2.0f / 8 && u < 3.0f / 8) t = -1.0f;
            if (u >= 5.0f / 8 && u < 6.0f / 8) t = 1.0f;
            if (u >= 6.0f / 8 && u < 7.0f / 8) t = 2.0f;
            if (u >= 7.0f / 8) t = 3.0f;
            s->flip_radian = s->flip_radian + TWO_PI_F * t;
        }
        if (is_reset) s->flip_radian = (s->w[0] > 5.0f) ? TWO_PI_F : -TWO_PI_F; /* FA:913 */
        s->cmd[0] = -1.0f;                                                        /* FA:917 / :1112 */
    }
}

/* full refresh_state (FA:349-360): body-frame relative quantities used by the observation and the rewards */
typedef struct { float rel_pos[3], rel_pos_b[3], rel_q_b[4], rel_v[3], rel_w[3], rel_v_b[3], rel_w_b[3]; } rel_state;
static inline void relative_state(const float p[3], const float q[4], const float v[3], const float w[3], const float pt[3],
                                  const float qt[4], rel_state *r) {
    float cq[4];
    quat_conj(q, cq);
    for (int k = 0; k < 3; ++k) {
        r->rel_pos[k] = pt[k] - p[k];
        r->rel_v[k] = 0.0f - v[k]; /* target_linvel / target_angvel are never written: stay 0 (FA:147-148) */
        r->rel_w[k] = 0.0f - w[k];
    }
    quat_rotate(cq, r->rel_pos, r->rel_pos_b);
    quat_mul(cq, qt, r->rel_q_b);
    quat_rotate(cq, r->rel_v, r->rel_v_b);
    quat_rotate(cq, r->rel_w, r->rel_w_b);
}
/* noise-free frame FA:415-421 + task tails FA:713-714, :768-771, :835-838 */
static inline void pack_frame(const rel_state *r, float V, const float act[4], float z, int grp, const float cmd[2], float out[26]) {
    float m[9];
    for (int k = 0; k < 3; ++k) out[k] = DIVC(r->rel_pos_b[k], 3.0f);
    quat_to_matrix(r->rel_q_b, m);
    for (int k = 0; k < 9; ++k) out[3 + k] = m[k];
    for (int k = 0; k < 3; ++k) out[12 + k] = r->rel_v_b[k] / 2.0f;
    for (int k = 0; k < 3; ++k) out[15 + k] = DIVC(r->rel_w_b[k], PI_F);
    out[18] = DIVC(V - 23.0f, 3.0f);
    for (int k = 0; k < 4; ++k) out[19 + k] = act[k];
    out[23] = 4.0f * clampf(z, 0.0f, 0.5f) - 1.0f;
    out[24] = cmd[0];
    out[25] = (grp == ORC_TASK_POS) ? cmd[1] : (grp == ORC_TASK_ROTATE ? DIVC(cmd[1], 6.0f) : DIVC(cmd[1] / 2.0f, PI_F));
}

/* the twelve standard normals of one env's observation noise (FA:402-410 draws eleven): Box-Muller on uniforms 4+2pr, 5+2pr of STREAM_OBS */
static inline void obs_normals(draw_ctx *D, float nrm[12]) {
    D->stream = STREAM_OBS; D->have = 0;
    for (int pr = 0; pr < 6; ++pr) {
        float ua = 1.0f - draw_u(D, 4 + 2 * pr); /* (0,1] */
        float ub = draw_u(D, 5 + 2 * pr);
        float rad = sqrtf(-2.0f * orc_logf(ua));
        float sn, cs;
        sincos_own(TWO_PI_F * ub, &sn, &cs);
        nrm[2 * pr] = rad * cs;
        nrm[2 * pr + 1] = rad * sn;
    }
}
/* test-fixture generator hook (tests/golden/make_glue_golden.py): the normals env `gid` consumes at step `step` */
void orc_obs_normals(uint64_t seed, uint32_t gid, uint32_t step, float nrm[12]) {
    draw_ctx D = {seed, gid, step, 0, 0, {0, 0, 0, 0}, 0};
    obs_normals(&D, nrm);
}

/* bulk hooks for tests/test_rng_distributions.py: the numbers the env consumes, by (env id, step, stream, index), without an env around them.
 * out[(e * n_step + t) * n_idx + k] = uniform k of stream `stream` of env gid0 + e at step step0 + t */
void orc_uniform_block(uint64_t seed, uint32_t gid0, int n_env, uint32_t step0, int n_step, uint32_t stream, int n_idx, float *out) {
    for (int e = 0; e < n_env; ++e)
        for (int t = 0; t < n_step; ++t) {
            draw_ctx D = {seed, gid0 + (uint32_t)e, step0 + (uint32_t)t, stream, 0, {0, 0, 0, 0}, 0};
            for (int k = 0; k < n_idx; ++k) out[((size_t)e * n_step + t) * n_idx + k] = draw_u(&D, k);
        }
}
/* round(N(0,1)) clamped to +-lim of n uniforms: the deploy-length (lim 1, FA:324) and delay-length (lim 3, FA:576) draws */
void orc_rounded_normal_vec(int n, const float *u, int lim, int32_t *out) { for (int i = 0; i < n; ++i) out[i] = rounded_normal(u[i], lim); }
/* out[(e * n_step + t) * 12 + k]: the twelve observation-noise normals of env gid0 + e at step step0 + t */
void orc_obs_normals_block(uint64_t seed, uint32_t gid0, int n_env, uint32_t step0, int n_step, float *out) {
    for (int e = 0; e < n_env; ++e)
        for (int t = 0; t < n_step; ++t) orc_obs_normals(seed, gid0 + (uint32_t)e, step0 + (uint32_t)t, out + ((size_t)e * n_step + t) * 12);
}

static void step_env(const orc_env *e, int i, const float *actions, float *obs_buf, float *states_buf, float *rew_buf,
                     int64_t *reset_buf, uint8_t *timeout_buf) {
    const orc_cfg *c = &e->cfg;
    env_state *s = &e->s[i];
    const int gid = c->env_offset + i;
    const int grp = task_group(e, gid);
    const uint32_t fl = c->flags;
    const float dtf = (float)c->dt;
    draw_ctx D = {c->seed, (uint32_t)gid, (uint32_t)e->step_count, 0, 0, {0, 0, 0, 0}, 0};

    /* ---- pre_physics_step FA:317-332 */
    const int is_reset = reset_buf[i] != 0;
    const int at_time = s->progress == 500;           /* FA:595-598, evaluated before progress is zeroed */
    if (is_reset) reset_env(e, s, grp, &D);
    if (is_reset || at_time) reset_command(e, s, grp, is_reset, at_time, &D);
    if (is_reset) { reset_buf[i] = 0; s->progress = 0; } /* FA:510-511 */
    for (int k = 0; k < 4; ++k) {
        s->act_old[k] = s->act[k];
        s->act[k] = clampf(actions[4 * i + k], -(float)c->clip_actions, (float)c->clip_actions); /* VT:304 */
    }
    int T = 10;
    if (fl & ORC_F_RANDOM_DEPLOY_TIME) { D.stream = STREAM_DEPLOY; D.have = 0; T = 10 - rounded_normal(draw_u(&D, 0), 1); }
    for (int sl = s->delay_len; sl < s->delay_len + T && sl < ORC_RING_SLOTS; ++sl)
        for (int k = 0; k < 4; ++k) s->ring[sl][k] = s->act[k]; /* FA:327-330 boolean-mask write */
    s->delay_len += T;

    /* ---- control_freq_inv x (mid_physics_step + simulate)  VT:309-313 */
    float wb[3];
    rotate_inv(s->q, s->w, wb); /* body-frame angular velocity of the root state (FA:350); carried by row I from here on */
    const int roundtrip = e->world_rate_roundtrip || (fl & ORC_F_WORLD_RATE_ROUNDTRIP);
    for (int ks = 0; ks < c->control_freq_inv; ++ks) {
        /* refresh_state, the part the inner loop consumes (FA:339-350) */
        float rpy[3], vb[3];
        euler_xyz_v1(s->q, rpy);
        for (int k = 0; k < 3; ++k) {
            float dl = rpy[k] - s->rpy_old[k];
            dl = (dl > 1.0f) ? dl - TWO_PI_F : dl;
            dl = (dl < -1.0f) ? dl + TWO_PI_F : dl;
            s->rpy_cont[k] = s->rpy_cont[k] + dl;
            s->rpy_old[k] = rpy[k];
        }
        rotate_inv(s->q, s->v, vb);
        /* delayed action FA:366 */
        int idx = s->delay_len - 1 < ks ? s->delay_len - 1 : ks;
        if (idx < 0) idx += ORC_RING_SLOTS;
        const float *ad = s->ring[idx];
        /* angular_vel_control FA:637-650 */
        float u[4], thr[4], des[3];
        u[0] = (ad[0] + 1.0f) / 2.0f * 1000.0f;
        for (int k = 0; k < 3; ++k) des[k] = ad[1 + k] * 20.0f;
        pid_step(dtf, e->rdt, des, wb, s->pid_prev, s->pid_int, &u[1]);
        allocator(u, thr);
        /* control_with_thrusts FA:608-635 */
        float Pm = mech_power(s->omega);
        s->bat_V = battery_step((fl & ORC_F_BATTERY_CONSUMPTION) != 0, dtf, Pm, &s->bat_E, &s->bat_u1, &s->bat_t);
        rotor_step(s->bat_V, thr, s->tau, s->opara, s->omega);
        if (fl & ORC_F_ROTOR_NOISE) { /* CTRL/thrust_dynamics.py:68-78; the noised speed is what is fed back (FA:616) */
            D.stream = STREAM_ROTOR; D.have = 0;
            for (int k = 0; k < 4; ++k) s->omega[k] = s->omega[k] * rand_float(1 - 10.0 / 700, 1 + 10.0 / 700, draw_u(&D, 4 * ks + k));
        }
        float rf[4], rt[4], bf[3], fs[4], ts[4];
        aero_step(s->cf, s->ct, s->dx, s->dy, s->kt, vb, s->omega, rf, rt, bf);
        real2sim(rf, rt, fs, ts);
        /* scatter FA:620-630 -> net body-frame wrench on the composite body (row I input); rotors sit at
         * s0 (+ax,+ay), s1 (-ax,+ay), s2 (-ax,-ay), s3 (+ax,-ay); a force f*z at r gives torque (r_y f, -r_x f, 0) */
        float F[3] = {0, 0, 0}, tq[3] = {0, 0, 0};
        if (!is_reset) { /* FA:629-630: envs reset this step get no force for all 10 substeps */
            F[0] = bf[0]; F[1] = bf[1];
            F[2] = bf[2] + ((fs[0] + fs[1]) + (fs[2] + fs[3]));
            tq[0] = (float)c->arm_y * ((fs[0] + fs[1]) - (fs[2] + fs[3]));
            tq[1] = -(float)c->arm_x * ((fs[0] - fs[1]) - (fs[2] - fs[3]));
            tq[2] = (ts[0] + ts[1]) + (ts[2] + ts[3]);
        }
        integrate_substep(&e->ip, s->p, s->q, s->v, wb, F, tq); /* gym.simulate VT:313 */
        /* the reference's data flow: simulate() leaves w = R(q) b in the root state, the next refresh_state re-derives b from it (FA:350).
         * After the tenth simulate() the root state KEEPS that w (round 5: rounds 3-4 rebuilt it from the round-tripped b once more, one
         * rotation pair too many: ~80 % of the world-frame rates were 1-4 ulp off the reference's, tests/golden/glue_*_ieee.npz) */
        if (roundtrip) { quat_sandwich(s->q, wb, s->w); rotate_inv(s->q, s->w, wb); }
    }

    if (!roundtrip) quat_sandwich(s->q, wb, s->w); /* carried mode: the root state's world-frame angular velocity is rebuilt once */

    /* ---- post_physics_step FA:374-388 */
    s->progress += 1;
    memmove(&s->ring[0][0], &s->ring[10][0], sizeof(float) * 4 * (ORC_RING_SLOTS - 10)); /* FA:378, tail [90,100) stays */
    s->delay_len = s->delay_len - 10 < 0 ? 0 : s->delay_len - 10;
    /* refresh_state again (FA:382): euler/unwrap + all relative quantities */
    {
        float rpy[3];
        euler_xyz_v1(s->q, rpy);
        for (int k = 0; k < 3; ++k) {
            float dl = rpy[k] - s->rpy_old[k];
            dl = (dl > 1.0f) ? dl - TWO_PI_F : dl;
            dl = (dl < -1.0f) ? dl + TWO_PI_F : dl;
            s->rpy_cont[k] = s->rpy_cont[k] + dl;
            s->rpy_old[k] = rpy[k];
        }
    }
    rel_state R;
    relative_state(s->p, s->q, s->v, s->w, s->pt, s->qt, &R);
    if (grp == ORC_TASK_FLIP) /* FA:831-832 / :930-931 */
        s->cmd[1] = clampf(s->flip_radian - s->rpy_cont[0], -TWO_PI_F, TWO_PI_F);
    /* compute_observation_state FA:390-421: shift the frame stacks, append the newest frame */
    float fr[26];
    pack_frame(&R, s->bat_V, s->act, s->p[2], grp, s->cmd, fr);
    float *ob = obs_buf + (size_t)i * c->len_obs * 26;
    float *st = states_buf + (size_t)i * c->len_states * 26;
    memmove(ob, ob + 26, sizeof(float) * 26 * (size_t)(c->len_obs - 1));
    memmove(st, st + 26, sizeof(float) * 26 * (size_t)(c->len_states - 1));
    float *on = ob + 26 * (c->len_obs - 1);
    memcpy(on, fr, sizeof(fr));
    memcpy(st + 26 * (c->len_states - 1), fr, sizeof(fr));
    if (fl & ORC_F_OBSERVATION_NOISE) { /* FA:402-410 */
        const double d = c->difficulty;
        const float df = (float)d;
        D.stream = STREAM_OBS; D.have = 0;
        float nrm[12];
        obs_normals(&D, nrm);
        for (int k = 0; k < 3; ++k) on[k] = on[k] + df * (nrm[k] * (float)(0.06 / 3 / 3) + 0.0f);
        float nq[4], mq[4], m[9];
        float l = (float)(d * 0.05), ml = (float)(-(d * 0.05)), sc = (float)((d * 0.05) - (-(d * 0.05)));
        (void)l;
        quat_from_euler(sc * draw_u(&D, 0) + ml, sc * draw_u(&D, 1) + ml, sc * draw_u(&D, 2) + ml, nq);
        quat_mul(R.rel_q_b, nq, mq);
        quat_to_matrix(mq, m);
        for (int k = 0; k < 9; ++k) on[3 + k] = m[k];
        for (int k = 0; k < 3; ++k) on[12 + k] = on[12 + k] + df * (nrm[3 + k] * (float)(0.1 / 3 / 2) + 0.0f);
        for (int k = 0; k < 3; ++k) on[15 + k] = on[15 + k] + df * (nrm[6 + k] * (float)(60.0 / 3 / 180) + 0.0f);
        on[18] = on[18] + df * (nrm[9] * (float)(0.06 / 3) + 0.0f);
        on[23] = on[23] + df * (nrm[10] * (float)(0.06 / 3 / 3) + 0.0f);
    }
    /* compute_reward FA:716-723 / :773-781 / :841-848 / :948-979 */
    int64_t rs;
    float rw;
    const float max_len = (float)c->max_episode_length;
    if (grp == ORC_TASK_POS) rw = reward_pos(R.rel_pos_b, s->p, s->q, s->qt, s->progress, max_len, &rs);
    else if (grp == ORC_TASK_ROTATE) rw = reward_rotate(R.rel_pos, R.rel_v, s->p, s->q, s->cmd, s->progress, max_len, &rs);
    else rw = reward_flip(R.rel_pos_b, R.rel_q_b, s->p, s->cmd, s->progress, max_len, &rs);
    rew_buf[i] = rw;
    reset_buf[i] = rs;
    timeout_buf[i] = (uint8_t)((s->progress >= c->max_episode_length - 1) && (rs != 0)); /* VT:323 */
}

/* VecTask.reset_done (VT:363-375): reset_idx(done ids) outside a step -- the first lines of step_env's pre-physics part, on their own:
 * FA:475-517 incl. reset_command_condition's progress == 500 envs (:500-503); reset_buf / progress_buf cleared (:510-511); the step
 * counter does not move (the draws are keyed by the index of the step that follows). */
int orc_reset_done(orc_env *e, int64_t *reset_buf) {
    const orc_cfg *c = &e->cfg;
    for (int i = 0; i < c->num_envs; ++i) {
        env_state *s = &e->s[i];
        const int gid = c->env_offset + i;
        const int grp = task_group(e, gid);
        draw_ctx D = {c->seed, (uint32_t)gid, (uint32_t)e->step_count, 0, 0, {0, 0, 0, 0}, 0};
        const int is_reset = reset_buf[i] != 0;
        const int at_time = s->progress == 500;
        if (is_reset) reset_env(e, s, grp, &D);
        if (is_reset || at_time) reset_command(e, s, grp, is_reset, at_time, &D);
        if (is_reset) { reset_buf[i] = 0; s->progress = 0; }
    }
    return 0;
}

int orc_step(orc_env *e, const float *actions, float *obs_buf, float *states_buf, float *rew_buf, int64_t *reset_buf,
             uint8_t *timeout_buf) {
    const int n = e->cfg.num_envs;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(e->threads)
#endif
    for (int i = 0; i < n; ++i) step_env(e, i, actions, obs_buf, states_buf, rew_buf, reset_buf, timeout_buf);
    e->step_count += 1;
    return 0;
}

/* ------------------------------------------------------------------------------------------------ state blob */
#define ROWF(r) ((float *)(blob + (size_t)(r) * n))
void orc_get_state(const orc_env *e, uint32_t *blob) {
    const size_t n = (size_t)e->cfg.num_envs;
    for (size_t i = 0; i < n; ++i) {
        const env_state *s = &e->s[i];
        for (int k = 0; k < 3; ++k) {
            ROWF(ORC_PX + k)[i] = s->p[k]; ROWF(ORC_VX + k)[i] = s->v[k]; ROWF(ORC_WX + k)[i] = s->w[k]; ROWF(ORC_TPX + k)[i] = s->pt[k];
            ROWF(ORC_RPY_OLD + k)[i] = s->rpy_old[k]; ROWF(ORC_RPY_CONT + k)[i] = s->rpy_cont[k];
            ROWF(ORC_PID_PREV + k)[i] = s->pid_prev[k]; ROWF(ORC_PID_INT + k)[i] = s->pid_int[k];
        }
        for (int k = 0; k < 4; ++k) {
            ROWF(ORC_QX + k)[i] = s->q[k]; ROWF(ORC_TQX + k)[i] = s->qt[k]; ROWF(ORC_OMEGA + k)[i] = s->omega[k];
            ROWF(ORC_ACT + k)[i] = s->act[k]; ROWF(ORC_ACT_OLD + k)[i] = s->act_old[k]; ROWF(ORC_TAU + k)[i] = s->tau[k];
        }
        for (int k = 0; k < 5; ++k) ROWF(ORC_OPARA + k)[i] = s->opara[k];
        ROWF(ORC_BAT_E)[i] = s->bat_E; ROWF(ORC_BAT_U1)[i] = s->bat_u1; ROWF(ORC_BAT_T)[i] = s->bat_t; ROWF(ORC_BAT_V)[i] = s->bat_V;
        ROWF(ORC_CMD)[i] = s->cmd[0]; ROWF(ORC_CMD + 1)[i] = s->cmd[1]; ROWF(ORC_FLIP_RADIAN)[i] = s->flip_radian;
        ROWF(ORC_CF)[i] = s->cf; ROWF(ORC_CT)[i] = s->ct; ROWF(ORC_DX)[i] = s->dx; ROWF(ORC_DY)[i] = s->dy; ROWF(ORC_KT)[i] = s->kt;
        blob[(size_t)ORC_PROGRESS * n + i] = (uint32_t)s->progress;
        blob[(size_t)ORC_DELAY_LEN * n + i] = (uint32_t)s->delay_len;
        for (int sl = 0; sl < ORC_RING_SLOTS; ++sl)
            for (int k = 0; k < 4; ++k) ROWF(ORC_NUM_FIELDS + sl * 4 + k)[i] = s->ring[sl][k];
    }
}
#undef ROWF
#define ROWF(r) ((const float *)(blob + (size_t)(r) * n))
void orc_set_state(orc_env *e, const uint32_t *blob) {
    const size_t n = (size_t)e->cfg.num_envs;
    for (size_t i = 0; i < n; ++i) {
        env_state *s = &e->s[i];
        for (int k = 0; k < 3; ++k) {
            s->p[k] = ROWF(ORC_PX + k)[i]; s->v[k] = ROWF(ORC_VX + k)[i]; s->w[k] = ROWF(ORC_WX + k)[i]; s->pt[k] = ROWF(ORC_TPX + k)[i];
            s->rpy_old[k] = ROWF(ORC_RPY_OLD + k)[i]; s->rpy_cont[k] = ROWF(ORC_RPY_CONT + k)[i];
            s->pid_prev[k] = ROWF(ORC_PID_PREV + k)[i]; s->pid_int[k] = ROWF(ORC_PID_INT + k)[i];
        }
        for (int k = 0; k < 4; ++k) {
            s->q[k] = ROWF(ORC_QX + k)[i]; s->qt[k] = ROWF(ORC_TQX + k)[i]; s->omega[k] = ROWF(ORC_OMEGA + k)[i];
            s->act[k] = ROWF(ORC_ACT + k)[i]; s->act_old[k] = ROWF(ORC_ACT_OLD + k)[i]; s->tau[k] = ROWF(ORC_TAU + k)[i];
        }
        for (int k = 0; k < 5; ++k) s->opara[k] = ROWF(ORC_OPARA + k)[i];
        s->bat_E = ROWF(ORC_BAT_E)[i]; s->bat_u1 = ROWF(ORC_BAT_U1)[i]; s->bat_t = ROWF(ORC_BAT_T)[i]; s->bat_V = ROWF(ORC_BAT_V)[i];
        s->cmd[0] = ROWF(ORC_CMD)[i]; s->cmd[1] = ROWF(ORC_CMD + 1)[i]; s->flip_radian = ROWF(ORC_FLIP_RADIAN)[i];
        s->cf = ROWF(ORC_CF)[i]; s->ct = ROWF(ORC_CT)[i]; s->dx = ROWF(ORC_DX)[i]; s->dy = ROWF(ORC_DY)[i]; s->kt = ROWF(ORC_KT)[i];
        s->progress = (int32_t)blob[(size_t)ORC_PROGRESS * n + i];
        s->delay_len = (int32_t)blob[(size_t)ORC_DELAY_LEN * n + i];
        for (int sl = 0; sl < ORC_RING_SLOTS; ++sl)
            for (int k = 0; k < 4; ++k) s->ring[sl][k] = ROWF(ORC_NUM_FIELDS + sl * 4 + k)[i];
    }
}
#undef ROWF

/* ------------------------------------------------------------------------------------------------ sub-model exports */
void orc_quat_mul(int n, const float *a, const float *b, float *out) { for (int i = 0; i < n; ++i) quat_mul(a + 4 * i, b + 4 * i, out + 4 * i); }
void orc_quat_rotate_inv(int n, const float *q, const float *v, float *out) { for (int i = 0; i < n; ++i) rotate_inv(q + 4 * i, v + 3 * i, out + 3 * i); }
void orc_euler_xyz_v1(int n, const float *q, float *rpy) { for (int i = 0; i < n; ++i) euler_xyz_v1(q + 4 * i, rpy + 3 * i); }
void orc_quat_from_euler_xyz(int n, const float *rpy, float *q) { for (int i = 0; i < n; ++i) quat_from_euler(rpy[3 * i], rpy[3 * i + 1], rpy[3 * i + 2], q + 4 * i); }
void orc_quat_diff_rad(int n, const float *a, const float *b, float *out) { for (int i = 0; i < n; ++i) out[i] = quat_diff_rad(a + 4 * i, b + 4 * i); }
void orc_quat_to_matrix(int n, const float *q, float *m9) { for (int i = 0; i < n; ++i) quat_to_matrix(q + 4 * i, m9 + 9 * i); }
void orc_pid_step(int n, float dt, const float *des, const float *cur, float *prev, float *integ, float *out) {
    const float rdt = div_const_is_exact(dt) ? 1.0f / dt : 0.0f;
    for (int i = 0; i < n; ++i) pid_step(dt, rdt, des + 3 * i, cur + 3 * i, prev + 3 * i, integ + 3 * i, out + 3 * i);
}
void orc_allocator(int n, float *u, float *thr) { for (int i = 0; i < n; ++i) allocator(u + 4 * i, thr + 4 * i); }
void orc_real2sim(int n, const float *f, const float *t, float *fs, float *ts) { for (int i = 0; i < n; ++i) real2sim(f + 4 * i, t + 4 * i, fs + 4 * i, ts + 4 * i); }
void orc_battery_step(int n, int enabled, float dt, const float *Pm, float *E, float *u1, float *t, float *V) {
    for (int i = 0; i < n; ++i) V[i] = battery_step(enabled, dt, Pm[i], &E[i], &u1[i], &t[i]);
}
void orc_power(int n, const float *omega, float *Pm) { for (int i = 0; i < n; ++i) Pm[i] = mech_power(omega + 4 * i); }
void orc_rotor_step(int n, const float *V, const float *thr, const float *tau, const float *para, float *omega) {
    for (int i = 0; i < n; ++i) rotor_step(V[i], thr + 4 * i, tau + 4 * i, para + 5 * i, omega + 4 * i);
}
void orc_aero(int n, const float *cf_ct, const float *d, const float *kt, const float *vb, const float *om, float *rf, float *rt,
              float *bf) {
    for (int i = 0; i < n; ++i) aero_step(cf_ct[2 * i], cf_ct[2 * i + 1], d[2 * i], d[2 * i + 1], kt[i], vb + 3 * i, om + 4 * i, rf + 4 * i, rt + 4 * i, bf + 3 * i);
}
void orc_reward_pos(int n, const float *rel_pos_b, const float *pos, const float *q, const float *qt, const int64_t *prog,
                    float max_len, float *rew, int64_t *reset) {
    for (int i = 0; i < n; ++i) rew[i] = reward_pos(rel_pos_b + 3 * i, pos + 3 * i, q + 4 * i, qt + 4 * i, prog[i], max_len, &reset[i]);
}
void orc_reward_rotate(int n, const float *rel_pos, const float *rel_v, const float *pos, const float *q, const float *cmd,
                       const int64_t *prog, float max_len, float *rew, int64_t *reset) {
    for (int i = 0; i < n; ++i) rew[i] = reward_rotate(rel_pos + 3 * i, rel_v + 3 * i, pos + 3 * i, q + 4 * i, cmd + 2 * i, prog[i], max_len, &reset[i]);
}
void orc_reward_flip(int n, const float *rel_pos_b, const float *relq, const float *pos, const float *cmd, const int64_t *prog,
                     float max_len, float *rew, int64_t *reset) {
    for (int i = 0; i < n; ++i) rew[i] = reward_flip(rel_pos_b + 3 * i, relq + 4 * i, pos + 3 * i, cmd + 2 * i, prog[i], max_len, &reset[i]);
}
void orc_obs_frame(int n, int task, const float *p, const float *q, const float *v, const float *w, const float *pt,
                   const float *qt, const float *V, const float *act, const float *cmd, const float *flip_radian,
                   const float *roll_cont, float *frame26, float *flip_cmd_out) {
    for (int i = 0; i < n; ++i) {
        rel_state R;
        relative_state(p + 3 * i, q + 4 * i, v + 3 * i, w + 3 * i, pt + 3 * i, qt + 4 * i, &R);
        float c2[2] = {cmd[2 * i], cmd[2 * i + 1]};
        if (task == ORC_TASK_POS) { c2[0] = 0.0f; c2[1] = 0.0f; }
        if (task == ORC_TASK_FLIP) {
            c2[0] = -1.0f;
            c2[1] = clampf(flip_radian[i] - roll_cont[i], -TWO_PI_F, TWO_PI_F);
            if (flip_cmd_out) flip_cmd_out[i] = c2[1];
        }
        pack_frame(&R, V[i], act + 4 * i, p[3 * i + 2], task, c2, frame26 + 26 * i);
    }
}
/* action noise of the policy forward (taco_policy.hpp: STREAM 7, counter (row, call, 7, a >> 2), two Box-Muller pairs per block) */
void orc_policy_noise(uint64_t seed, uint32_t call, int n, int act_dim, float *eps) {
    for (int i = 0; i < n; ++i) {
        uint32_t r[4] = {0, 0, 0, 0};
        for (int a = 0; a < act_dim; ++a) {
            if ((a & 3) == 0) orc_philox(seed, (uint32_t)i, call, 7u, (uint32_t)(a >> 2), r);
            const uint32_t ba = (a & 2) ? r[2] : r[0], bb = (a & 2) ? r[3] : r[1];
            const float ua = 1.0f - orc_uniform(ba), ub = orc_uniform(bb);
            const float rad = sqrtf(-2.0f * orc_logf(ua));
            float sn, cs;
            sincos_own(TWO_PI_F * ub, &sn, &cs);
            eps[(size_t)i * act_dim + a] = (a & 1) ? rad * sn : rad * cs;
        }
    }
}

/* ------------------------------------------------------------------------------------------------ row N1 (next): replay buffer
 * PPOReplayBuffer.compute_returns_and_advantage (IsaacGymEnvs/algorithms/buffer_asymmetry.py:93-132).  Arrays are [H][N]
 * (the reference's [H, N, 1]); gamma / lam are Python floats that meet fp32 tensors.  adv_raw is the GAE before the
 * normalisation of :132, ret = adv_raw + value (:130). */
void orc_gae(int H, int N, double gamma, double lam, const float *rew, const float *done, const float *value, const float *last_value,
             float *adv_raw, float *ret) {
    const float g = (float)gamma, l = (float)lam;
    for (int i = 0; i < N; ++i) {
        float last = 0.0f;
        for (int t = H - 1; t >= 0; --t) {
            const float nv = (t == H - 1) ? last_value[i] : value[(size_t)(t + 1) * N + i];
            const float nnt = 1.0f - done[(size_t)t * N + i];
            const float td = rew[(size_t)t * N + i] + nnt * g * nv;
            const float delta = td - value[(size_t)t * N + i];
            last = delta + nnt * g * l * last;
            adv_raw[(size_t)t * N + i] = last;
            ret[(size_t)t * N + i] = last + value[(size_t)t * N + i];
        }
    }
}
/* (adv - mean) / (std + 1e-8), torch.std = unbiased (buffer_asymmetry.py:132); the reductions are done in double */
void orc_normalize_advantage(size_t n, float *adv) {
    double s = 0.0, ss = 0.0;
    for (size_t k = 0; k < n; ++k) s += adv[k];
    const double mean = s / (double)n;
    for (size_t k = 0; k < n; ++k) ss += (adv[k] - mean) * (adv[k] - mean);
    const float stdv = (float)sqrt(ss / (double)(n - 1)), m = (float)mean;
    for (size_t k = 0; k < n; ++k) adv[k] = (adv[k] - m) / (stdv + 1e-8f);
}

void orc_rand_float(int n, double lower, double upper, const float *u, float *out) { for (int i = 0; i < n; ++i) out[i] = rand_float(lower, upper, u[i]); }
void orc_integrate(const orc_cfg *cfg, int n, float *root13, const float *wrench6) {
    orc_env tmp;
    memset(&tmp, 0, sizeof(tmp));
    tmp.cfg = *cfg;
    derive(&tmp);
    for (int i = 0; i < n; ++i) {
        float *r = root13 + 13 * i;
        float wb[3];
        rotate_inv(r + 3, r + 10, wb); /* row C's body-frame angular velocity (FA:350) */
        integrate_substep(&tmp.ip, r, r + 3, r + 7, wb, wrench6 + 6 * i, wrench6 + 6 * i + 3);
        quat_sandwich(r + 3, wb, r + 10);
    }
}
