// Wavefront-level arithmetic for gfx950 (CDNA4): one 64-lane wavefront cooperates on one
// MPC instance.  Cross-lane data movement is ds_bpermute (LDS crossbar, no LDS memory);
// reductions are xor-butterflies so that every lane ends with the same bits.
//
// Built with -ffp-contract=off: the order of operations written here is the order that
// runs.  DESIGN.md ("Arithmetic specification") states that order.
#pragma once
#include <hip/hip_runtime.h>

namespace pantr {

constexpr int WAVE = 64;

/// sqrt and division for operands well inside the normal range (no zeros, infinities, subnormals;
/// exponents far from the ends): the compiler's correctly rounded fp64 expansions (v_rsq_f64 /
/// v_rcp_f64 seeds, Goldschmidt / Newton steps in fma) without their range handling - the
/// v_ldexp scaling and class test around the square root (8 of 18 instructions), v_div_scale,
/// v_div_fmas and v_div_fixup around the division (4 of 11).  On such operands the dropped
/// instructions are identities, so the results are the correctly rounded ones, bit for bit those of
/// __builtin_sqrt and '/' (and of the oracle's).  Outside that range (the idle lanes of the chain
/// sweeps divide by zero) the result is NaN or infinity; the callers never use those lanes.
__device__ __forceinline__ double sqrt_inrange(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    double r = __builtin_fma(-h, g, 0.5);
    g        = __builtin_fma(g, r, g);
    h        = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g        = __builtin_fma(d, h, g);
    d        = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
__device__ __forceinline__ double div_inrange(double a, double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y        = __builtin_fma(y, e, y);
    e        = __builtin_fma(-b, y, 1.0);
    y        = __builtin_fma(y, e, y);
    double q = a * y;
    double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}

// ---- first-order dual number (value, tangent) -------------------------------------------
struct dual {
    double v, t;
};
__device__ __forceinline__ dual operator+(dual a, dual b) { return {a.v + b.v, a.t + b.t}; }
__device__ __forceinline__ dual operator-(dual a, dual b) { return {a.v - b.v, a.t - b.t}; }
__device__ __forceinline__ dual operator*(dual a, dual b) {
    return {a.v * b.v, __builtin_fma(a.t, b.v, a.v * b.t)};
}
__device__ __forceinline__ dual operator*(double c, dual a) { return {c * a.v, c * a.t}; }
__device__ __forceinline__ dual operator+(dual a, double c) { return {a.v + c, a.t}; }
__device__ __forceinline__ dual operator-(dual a, double c) { return {a.v - c, a.t}; }
__device__ __forceinline__ dual operator-(double c, dual a) { return {c - a.v, 0.0 - a.t}; }
/// c / b = q with tangent b.t * k, k = (-q) / b.v: divisions act on primal values only, which
/// is what lets the tape path record (q, k) once and replay tangents without dividing.
__device__ __forceinline__ dual operator/(double c, dual b) {
    double q = div_inrange(c, b.v);
    double k = div_inrange(-q, b.v);
    return {q, b.t * k};
}
/// a / b = q with tangent (a.t - q b.t) / b.v
__device__ __forceinline__ dual operator/(dual a, dual b) {
    double q = div_inrange(a.v, b.v);
    return {q, div_inrange(a.t - q * b.t, b.v)};
}
/// a / b of the models' stage functions: the in-range division (same results; its reciprocal
/// iterations do not wait for the numerator, which shortens the dependent chain of the attitude integration)
__device__ __forceinline__ double qdiv(double a, double b) { return div_inrange(a, b); }
__device__ __forceinline__ dual qdiv(dual a, dual b) { return a / b; }
__device__ __forceinline__ double qrecip(double b) { return div_inrange(1.0, b); }
__device__ __forceinline__ dual qrecip(dual b) { return 1.0 / b; }
__device__ __forceinline__ dual operator-(dual a) { return {-a.v, -a.t}; }
/// c k + s in one rounding (the Runge-Kutta combinations of the scalar models)
__device__ __forceinline__ double madd(double c, double k, double s) { return __builtin_fma(c, k, s); }
__device__ __forceinline__ dual madd(double c, dual k, dual s) {
    return {__builtin_fma(c, k.v, s.v), __builtin_fma(c, k.t, s.t)};
}
/// clamp pieces with the tangent of the selected branch
__device__ __forceinline__ double tmin(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double tmax(double a, double b) { return a > b ? a : b; }
__device__ __forceinline__ dual tmin(dual a, double b) { return a.v < b ? a : dual{b, 0.0}; }
__device__ __forceinline__ dual tmax(dual a, double b) { return a.v > b ? a : dual{b, 0.0}; }

/// Deterministic sin/cos: 2-term Cody-Waite reduction by pi/2 and the fdlibm kernel polynomials
/// with explicit fma; the oracle runs the same operations, so trigonometric values agree bit for
/// bit between CPU and GPU (the device math library and libm do not agree with each other).
__device__ __forceinline__ void sincos_det(double x, double &s, double &c) {
    const double kd = __builtin_rint(x * 6.36619772367581382433e-01); // x * 2/pi, ties to even
    double r        = __builtin_fma(-kd, 1.57079632673412561417e+00, x);
    r               = __builtin_fma(-kd, 6.07710050650619224932e-11, r);
    const double z  = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps        = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps        = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps        = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps        = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double sr = __builtin_fma(r * z, ps, r);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc        = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc        = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc        = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc        = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double cr = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
    // quadrant q = kd mod 4: s = sr, cr, -sr, -cr and c = cr, -sr, -cr, sr for q = 0..3, as one swap on
    // the low bit of q and one sign-bit flip each (the same bits as the selects, in a third of the instructions)
    const int q = static_cast<int>(kd); // |x| < 3e9 (64-bit conversions are emulated on the GPU)
    const bool sw = q & 1;
    const double s_ = sw ? cr : sr, c_ = sw ? sr : cr;
    s = __hiloint2double(__double2hiint(s_) ^ ((q & 2) << 30), __double2loint(s_));
    c = __hiloint2double(__double2hiint(c_) ^ (((q + 1) & 2) << 30), __double2loint(c_));
}
__device__ __forceinline__ void sincos_t(double x, double &s, double &c) { sincos_det(x, s, c); }
__device__ __forceinline__ void sincos_t(dual x, dual &s, dual &c) {
    double sv, cv;
    sincos_det(x.v, sv, cv);
    s = {sv, cv * x.t};
    c = {cv, -(sv * x.t)};
}

/// sqrt(a) = r with tangent a.t * h, h = 0.5 / r
__device__ __forceinline__ dual wsqrt(dual a) {
    double r = __builtin_sqrt(a.v);
    double h = 0.5 / r;
    return {r, a.t * h};
}
__device__ __forceinline__ double wsqrt(double a) { return __builtin_sqrt(a); }

template <class T>
__device__ __forceinline__ T make_const(double c);
template <>
__device__ __forceinline__ double make_const<double>(double c) { return c; }
template <>
__device__ __forceinline__ dual make_const<dual>(double c) { return {c, 0.0}; }

__device__ __forceinline__ double select(bool c, double a, double b) { return c ? a : b; }
__device__ __forceinline__ dual select(bool c, dual a, dual b) {
    return {c ? a.v : b.v, c ? a.t : b.t};
}

// ---- cross-lane ----------------------------------------------------------------------------
/// Value of `v` held by lane `src` (src taken modulo 64 by the hardware).
__device__ __forceinline__ double lane_read(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo     = __builtin_amdgcn_ds_bpermute(src << 2, lo);
    hi     = __builtin_amdgcn_ds_bpermute(src << 2, hi);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ dual lane_read(dual v, int src) {
    return {lane_read(v.v, src), lane_read(v.t, src)};
}

// DPP row shifts (VALU-speed cross-lane moves inside a 16-lane row; a lane whose source falls
// outside the row reads 0).  row_from_prev: lane i gets lane i-1; row_from_next: lane i gets i+1.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo     = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi     = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
/// The same with a row mask: lanes of the 16-lane rows outside ROW_MASK (bit r = row r) read 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_rows(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo     = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, true);
    hi     = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, true);
    return __hiloint2double(hi, lo);
}
/// Inclusive running sum over the 64 lanes, lane i <- x_0 + ... + x_i, the additions grouped as the
/// oracle's prefix_scan() groups them: Kogge-Stone at distances 1, 2, 4, 8 inside the 16-lane rows
/// (row_shr DPP moves; a lane without a source adds +0.0), then rows 1 and 3 add the total of the row
/// before them (row_bcast:15), then rows 2 and 3 add the total of the lower half (row_bcast:31).  All
/// lanes must be active.
__device__ __forceinline__ double wave_running_sum(double x) {
    x = x + dpp_move<0x111>(x);
    x = x + dpp_move<0x112>(x);
    x = x + dpp_move<0x114>(x);
    x = x + dpp_move<0x118>(x);
    x = x + dpp_move_rows<0x142, 0xA>(x);
    x = x + dpp_move_rows<0x143, 0xC>(x);
    return x;
}
/// The same inside every 16-lane row by itself: the first four levels of wave_running_sum() - for lanes 0..15 the
/// very additions of wave_running_sum() when lanes 16..63 hold zeros (its last two levels then add nothing to row 0).
__device__ __forceinline__ double row_running_sum(double x) {
    x = x + dpp_move<0x111>(x);
    x = x + dpp_move<0x112>(x);
    x = x + dpp_move<0x114>(x);
    x = x + dpp_move<0x118>(x);
    return x;
}
/// Like dpp_move_rows, but a lane without a source (outside ROW_MASK, or shifted in from outside its
/// row) keeps `old`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_or(double old, double v) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
/// An affine map of a pair, y -> A y + b (oracle: SingleShootingAL::Affine2, same arithmetic).
struct Affine2 {
    double a00, a01, a10, a11, b0, b1;
    __device__ __forceinline__ void apply(double y0, double y1, double &o0, double &o1) const {
        o0 = __builtin_fma(a01, y1, __builtin_fma(a00, y0, b0));
        o1 = __builtin_fma(a11, y1, __builtin_fma(a10, y0, b1));
    }
};
__device__ __forceinline__ Affine2 affine_identity() { return {1.0, 0.0, 0.0, 1.0, 0.0, 0.0}; }
/// later o earlier
__device__ __forceinline__ Affine2 affine_compose(const Affine2 &l, const Affine2 &e) {
    Affine2 c;
    c.a00 = __builtin_fma(l.a01, e.a10, l.a00 * e.a00);
    c.a01 = __builtin_fma(l.a01, e.a11, l.a00 * e.a01);
    c.a10 = __builtin_fma(l.a11, e.a10, l.a10 * e.a00);
    c.a11 = __builtin_fma(l.a11, e.a11, l.a10 * e.a01);
    c.b0  = __builtin_fma(l.a01, e.b1, __builtin_fma(l.a00, e.b0, l.b0));
    c.b1  = __builtin_fma(l.a11, e.b1, __builtin_fma(l.a10, e.b0, l.b1));
    return c;
}
/// Inclusive composition over the 64 lanes, lane i <- m_i o ... o m_0, grouped as the oracle's
/// affine_scan() groups it (the levels of wave_running_sum; a lane without a partner composes with
/// the identity).  All lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Affine2 affine_partner(const Affine2 &m) {
    return {dpp_move_or<CTRL, ROW_MASK>(1.0, m.a00), dpp_move_or<CTRL, ROW_MASK>(0.0, m.a01),
            dpp_move_or<CTRL, ROW_MASK>(0.0, m.a10), dpp_move_or<CTRL, ROW_MASK>(1.0, m.a11),
            dpp_move_or<CTRL, ROW_MASK>(0.0, m.b0),  dpp_move_or<CTRL, ROW_MASK>(0.0, m.b1)};
}
__device__ __forceinline__ Affine2 wave_affine_scan(Affine2 m) {
    m = affine_compose(m, affine_partner<0x111, 0xF>(m));
    m = affine_compose(m, affine_partner<0x112, 0xF>(m));
    m = affine_compose(m, affine_partner<0x114, 0xF>(m));
    m = affine_compose(m, affine_partner<0x118, 0xF>(m));
    m = affine_compose(m, affine_partner<0x142, 0xA>(m));
    m = affine_compose(m, affine_partner<0x143, 0xC>(m));
    return m;
}
/// The value of the lane below (lane 0: +0.0): wave_shr:1.
__device__ __forceinline__ double lane_below(double v) { return dpp_move<0x138>(v); }
/// The value lane `src` (wave-uniform) holds, on every lane.
__device__ __forceinline__ double lane_value(double v, int src) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_from_prev(double v) { return dpp_move<0x111>(v); } // row_shr:1
__device__ __forceinline__ double row_from_next(double v) { return dpp_move<0x101>(v); } // row_shl:1
__device__ __forceinline__ dual row_from_prev(dual v) { return {row_from_prev(v.v), row_from_prev(v.t)}; }
__device__ __forceinline__ dual row_from_next(dual v) { return {row_from_next(v.v), row_from_next(v.t)}; }

/// 64-lane xor-butterfly all-reduce, masks 32, 16, 8, 4, 2, 1 (a + partner at every level, so all
/// lanes finish with bit-identical sums), without the LDS crossbar: v_permlane32_swap and
/// v_permlane16_swap (gfx950) across rows, DPP moves inside a 16-lane row.  After a permlane swap
/// of a value with itself the two results hold "own" and "partner" in one order or the other on
/// every row, and own + partner is commutative.  ~190 cycles against 477 for six ds_bpermute
/// levels (scripts/microbench/wave_sum_test.hip).  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ double dpp_partner(double v) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
/// {a, b} = {own, value of the lane 32 (SWAP32) or 16 lanes away} on every lane.
template <bool SWAP32>
__device__ __forceinline__ void row_partner(double v, double &a, double &b) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    if constexpr (SWAP32) {
        auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        a = __hiloint2double(h[0], l[0]);
        b = __hiloint2double(h[1], l[1]);
    } else {
        auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        a = __hiloint2double(h[0], l[0]);
        b = __hiloint2double(h[1], l[1]);
    }
}
/// Rows 0, 1, 2 of the wavefront hold x0, x1, x2 at the same position of their 16 lanes: all three, on every row
/// (v_permlane16_swap pairs rows (0,1) and (2,3), v_permlane32_swap rows (0,2) and (1,3); row 3's value is dropped).
__device__ __forceinline__ void rows_gather3(double v, double &c0, double &c1, double &c2) {
    double even, odd, c3;
    row_partner<false>(v, even, odd); // rows 0,1: (row 0, row 1); rows 2,3: (row 2, row 3)
    row_partner<true>(even, c0, c2);  // row 0's and row 2's value, on all rows
    row_partner<true>(odd, c1, c3);   // row 1's (and row 3's)
}
/// Value of lane ^ 4: four lanes up for lanes with bit 2 clear, four lanes down for the others.
__device__ __forceinline__ double xor4_partner(double v) {
    double up = dpp_partner<0x104>(v), down = dpp_partner<0x114>(v); // row_shl:4, row_shr:4
    return (threadIdx.x & 4) ? down : up;
}
__device__ __forceinline__ double wave_sum(double v) {
    double a, b;
    row_partner<true>(v, a, b);
    v = a + b;
    row_partner<false>(v, a, b);
    v = a + b;
    v = v + dpp_partner<0x128>(v); // row_ror:8 = lane ^ 8
    v = v + xor4_partner(v);
    v = v + dpp_partner<0x4E>(v); // quad_perm [2,3,0,1]
    v = v + dpp_partner<0xB1>(v); // quad_perm [1,0,3,2]
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
    double a, b;
    row_partner<true>(v, a, b);
    v = fmax(a, b);
    row_partner<false>(v, a, b);
    v = fmax(a, b);
    v = fmax(v, dpp_partner<0x128>(v));
    v = fmax(v, xor4_partner(v));
    v = fmax(v, dpp_partner<0x4E>(v));
    return fmax(v, dpp_partner<0xB1>(v));
}

/// Order this wavefront's LDS/global accesses: what any lane wrote before is visible to every
/// lane after (the workgroup is a single wavefront, so no s_barrier is involved).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

/// The same among the lanes of ONE wavefront only (nothing another wavefront reads is being published): a
/// wavefront's LDS accesses execute in order and so do its vector-memory operations, so nothing is waited for -
/// the fences keep the compiler from moving accesses across.
__device__ __forceinline__ void wave_sync_self() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/// LDS-DMA (gfx950 global_load_lds_dwordx4): every lane fetches 16 bytes from its own global
/// address and the wavefront's 1 KiB lands contiguously at the wave-uniform LDS address
/// `lds_row` (+ lane*16), without passing through VGPRs.  Completion is tracked by vmcnt.
/// OFFSET: the instruction's immediate offset, added to BOTH addresses (scripts/microbench/lds_dma_offset_test.hip):
/// the consecutive 1 KiB rows of a stage go with one address register pair.
template <int OFFSET = 0>
__device__ __forceinline__ void lds_dma_row(const void *gsrc_lane, void *lds_row) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc_lane,
                                     (__attribute__((address_space(3))) void *)lds_row, 16, OFFSET, 0);
}
/// ROWS consecutive 1 KiB rows (ROWS <= 4) from one per-lane global address to one LDS address, as one group of
/// global_load_lds_dwordx4 with immediate offsets 0, 1024, ... and M0 written ONCE: the builtin form writes M0
/// anew before every transfer (two scalar instructions each), and a lone wavefront pays for every instruction
/// it issues.  The transfers are invisible to the compiler's s_waitcnt bookkeeping - which only makes the waits
/// it inserts for its own loads more conservative (vmcnt counts every operation) - and to its record of the LDS
/// objects LDS-DMA writes, from which it would otherwise put a full s_waitcnt vmcnt(0) in front of some reads of
/// the landing buffers (it does not see a sweep's explicit vmcnt(n) waits): a sweep using this waits for its rows
/// explicitly (WaveProblem::hess_prod_tape_d2).  lds_row must be wave-uniform.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm" // (M0 on the clobber list: the compiler writes M0 right before each of its own uses)
template <int ROWS>
__device__ __forceinline__ void lds_dma_rows(const void *gsrc_lane, void *lds_row) {
    static_assert(ROWS >= 1 && ROWS <= 4, "immediate offsets up to 3072");
    const int dst = __builtin_amdgcn_readfirstlane((int)(uintptr_t)(__attribute__((address_space(3))) char *)lds_row);
    if constexpr (ROWS == 1)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc_lane), "s"(dst) : "memory", "m0");
    else if constexpr (ROWS == 2)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024"
                     ::"v"(gsrc_lane), "s"(dst) : "memory", "m0");
    else if constexpr (ROWS == 3)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, off offset:2048" ::"v"(gsrc_lane), "s"(dst) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                     ::"v"(gsrc_lane), "s"(dst) : "memory", "m0");
}
/// The 4-byte form in the same manner: 256 bytes per wavefront.
__device__ __forceinline__ void lds_dma_256(const void *gsrc_lane, void *lds_row) {
    const int dst = __builtin_amdgcn_readfirstlane((int)(uintptr_t)(__attribute__((address_space(3))) char *)lds_row);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc_lane), "s"(dst) : "memory", "m0");
}
#pragma clang diagnostic pop
/// Wait until this wavefront's outstanding global loads/stores and LDS-DMA have completed.
__device__ __forceinline__ void wait_vmem() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

/// The value lane SRC holds, as a wave-uniform (scalar-register) double: two v_readlane_b32.
template <int SRC>
__device__ __forceinline__ double read_lane(double v) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), SRC);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), SRC);
    return __hiloint2double(hi, lo);
}

/// The value lane SRC of this lane's 16-lane row holds: one v_mov_b64_dpp row_newbcast (the only DPP
/// form the 64-bit ALU has), against two v_readlane_b32 + the scalar-to-vector hop of read_lane().
template <int SRC>
__device__ __forceinline__ double row_bcast(double v) {
    static_assert(SRC >= 0 && SRC < 16, "row_newbcast selects a lane of the 16-lane row");
    long long r = __builtin_amdgcn_update_dpp(0ll, __builtin_bit_cast(long long, v), 0x150 + SRC, 0xF, 0xF, true);
    return __builtin_bit_cast(double, r);
}

/// Move a wave-uniform double into scalar registers (lets the compiler branch on SCC).
__device__ __forceinline__ double uniform(double v) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

} // namespace pantr
