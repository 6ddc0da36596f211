// kernels_col.hpp -- "lane per column" transport kernels (the production path of every tracer but T and S).
//
// Same physics as kernels_fct.hpp (FCT adv_flux, isoflux, explicit update,
// invtri; reference lines cited there), laid out for the CDNA4 wavefront:
//
//   one lane  = one (i, r) ocean column of one tracer; the lane marches down k = 1..km keeping
//               the k-1,k,k+1 window of both time levels in registers
//   one wave  = 64 lanes taken from a host-built LANE MAP (uvic_gpu.hip: build_col_lanes): the
//               ocean columns of the slab row by row, so that lanes are adjacent columns of a row
//               wherever the sea is; land columns and the polar caps get no lanes at all
//   pass A: x-neighbours come from the adjacent lanes by DPP whole-wave shifts, so every run of
//               adjacent ocean columns carries two halo lanes on each side (not owned: they
//               compute, their results are not stored); a wave may hold pieces of several runs
//               and rows.  Rows r-1, r+1, r+2 come from the lane's own coalesced loads.  Longitude
//               wraps cyclically.  The four waves of a workgroup are four tracers of the same lanes
//               and share what does not depend on the tracer through LDS.
//   pass B: no neighbour exchange at all: the lanes are the ocean columns and nothing else.
//
// The isopycnal flux terms are linear in the tracer with coefficients that do not
// depend on the tracer: `ai_coef_cell` folds Ai * slope (and metric factors, masks,
// background diffusivities) into 19 per-face coefficients ONCE per step, so the
// 24 fp64 divisions per cell update of the reference formulation disappear from
// the per-tracer work.  This re-associates floating-point products:
//     reference  ((Ai*dT)*drodx)/(drodz+eps)      here  (Ai*drodx/(drodz+eps))*dT
// so results agree with the reference to rounding (tested: <= 1e-13 relative after one step,
// <= 1e-12 after 20 and 100 steps, tests/test_gpu_fast.py, test_gpu_drift100.py), not bit for bit.
// T and S, on whose bits every convective adjustment is decided, do not come this way in the
// production step: kernels_colx.hpp keeps them in the reference's order of operations.
//
// Pass A (`colfct_wave`): low-order fluxes, t_lo, limiter ratios, limited x and z
//   fluxes, all diffusive fluxes -> S = the explicit tendency except the y
//   advection and the source term; and t_lo and the y ratios of the row to the north, with which
//   it forms the FINAL limited flux through its north face.
// Pass B (`colupd_wave`): ADV_Ty from the final fluxes of rows r and r-1, explicit update,
//   tridiagonal solve -> t(tau+1).
// Fluxes are carried as HALF of the reference's (which stores 2 x flux, fdift.h): the upstream flux
// v*(a+b) + |v|*(a-b) is 2*v*a for v >= 0 and 2*v*b otherwise, i.e. one product and a select instead of
// four operations, the centred part v*(a+b) is v times the face mean the limiter needs anyway, and
// the metric factors lose their 1/2 (cstdxtr for cstdxt2r, dztr for dzt2r).
#ifndef UVIC_KERNELS_COL_HPP
#define UVIC_KERNELS_COL_HPP

#include "kernels_fct.hpp"

namespace uvic {

// slots of the coefficient buffer.  Two slots share one 16-byte element per cell (a lane fetches both with
// one global_load_dwordx4): the buffer is (2, imt, km, jmt, CF_PAIRS).  The north-face slots come first
// because pass A also reads them for row r-1.
enum {
  CF_AN = 0, CF_CN = 1,   // north face:  A, C[jq + 2*kr]
  CF_AE = 5, CF_CE = 6,   // east face:   A, C[ip + 2*kr]
  CF_BV = 10, CF_CBX = 11, CF_CBY = 15,  // bottom face
  CF_COUNT = 19,
  // the total advective velocities, packed by isopyc_adv_cell and isopyc_column (or k_tot_vel) for the pass-A waves that share a level's pairs through LDS:
  CF_VE = 20, CF_VN = 21,   // through the east and north faces of the cell
  CF_VB = 22, CF_VS = 23,   // through the bottom face of level k (adv_vbt at k = km: tracer.F:1065) and the north face of row j-1
  CF_PAIRS = 12, CF_DPAIRS = 10   // all pair planes; those of the diffusive coefficients
};
#define CF_IDX(slot, q, N3) ((((size_t)((slot) / 2) * (N3)) + (q)) * 2 + ((slot) % 2))

// ---------------------------------------------------------------------------
// ai_east / ai_north / ai_bottom (isopyc.F:559-921) and the coefficient folding in one pass over the
// cell: every slope drod?/(drodz+eps) is formed once and serves the taper of Ai, K11/K22/K33 and the
// folded coefficient Ai*slope (the separate passes divide twice by the same denominator and move the
// 16 Ai planes through memory).  Column-kernel path only; Ai_* are not stored.  i = 2..imt-1,
// k = 1..km, j = 1..jmt-1.
// ---------------------------------------------------------------------------
#if defined(__HIPCC__) && !defined(UV_NO_CONTRACT)
#define UV_FAST_FP _Pragma("clang fp contract(fast)")
#else
#define UV_FAST_FP
#endif
// Nothing here is contracted: Ai, K11, K22, K33 come out bit-identical to isopyc_ai_cell (the T,S passes and diff_cbt =
// background + K33 depend on them to the bit); the folded coefficients are products only.
// `store_ai`: also store the sixteen Ai planes (the exact T,S kernels read them, kernels_colx.hpp).
// A land cell (37 % of the grid) is left at once: every product of it carries its mask and is zero, the buffers are
// zero-filled when they are allocated and again when kmt changes (uvic_gpu.hip: iso_set_alloc, make_tmask), and nothing
// reads the one unmasked slot (the vertical-diffusion coefficient) of a land cell: t(tau-1) is zero on both sides of
// the face it belongs to.  The nineteen coefficients leave in ten 16-byte stores.
// x / y with the reciprocal of y formed once for several quotients: the compiler's own fp64 division sequence
// (v_rcp_f64, two Newton steps; quotient, residual, one correction) split in two -- the same bits as x / y for operands
// away from the ends of the exponent range, which slopes and density differences are (kernels_colx.hpp uses the same
// split).  The host build divides.
UVIC_DEV double uv_rcp2(double y) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(y);
  r = __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
  return __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
#else
  return 1.0 / y;
#endif
}
UVIC_DEV double uv_div2(double x, double y, double r) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double q = x * r;
  return __builtin_fma(__builtin_fma(-y, q, x), r, q);
#else
  (void)r;
  return x / y;
#endif
}
UVIC_DEV void ai_coef_cell(const uvic_ctx &c, double *cf, int i, int k, int j, int store_ai = 0) {
  UV_DIMS(c);
  const size_t q = X3(i, k, j);
  if (TMASK(i, k, j) == 0.0) return;
  const double sc = 1.0 / (c.slmxr * c.dtxsqr[k - 1]);
  const double dzt4r = 0.5 * c.dzt2r[k - 1];
  double cfv[2 * CF_DPAIRS];
  for (int p = 0; p < 2 * CF_DPAIRS; ++p) cfv[p] = 0.0;
  const bool inner = j >= 2 && j <= jmt - 1;   // rows whose east and bottom faces are needed
#define IDX(ii) X3(ii, k, j)
  if (j >= 2) {  // east face
    const double mm = TMASK(i, k, j) * TMASK(i + 1, k, j);
    const double Ai0 = .5 * (c.fisop[XFIS(i, j, k)] + c.fisop[XFIS(i + 1, j, k)]) * c.ahisop + c.addisop[q];
    double sumz = 0.0;
    for (int kr = 0; kr <= 1; ++kr)
      for (int ip = 0; ip <= 1; ++ip) {
        const double sl = drodxe(i, k, j, ip) / (drodze(i, k, j, ip, kr) + UV_EPSLN);
        const double sxe = dabs(sl);
        double a;
        if (sxe > sc) {
          const double r = sc / (sxe + UV_EPSLN);
          a = Ai0 * TMASK(i, k, j) * TMASK(i + 1, k, j) * (r * r);
        } else {
          a = Ai0 * TMASK(i, k, j) * TMASK(i + 1, k, j);
        }
        if (store_ai) { double *p = c.Ai_ez + (size_t)(ip + 2 * kr) * N3; UV_CYC_STORE(p, IDX, i, a); }
        sumz = sumz + c.dzw[k - 1 + kr] * a;
        cfv[CF_CE + ip + 2 * kr] = -dzt4r * (a * sl);
      }
    const double k11 = dzt4r * sumz;
    UV_CYC_STORE(c.K11, IDX, i, k11);
    const double cstdxur = c.cstr[j - 1] * c.dxur[i - 1];
    cfv[CF_AE] = (c.diff_cet * cstdxur + k11 * cstdxur) * mm;
  }
  {  // north face
    const double mm = TMASK(i, k, j) * TMASK(i, k, j + 1);
    const double Ai0 = 0.5 * (c.fisop[XFIS(i, j, k)] + c.fisop[XFIS(i, j + 1, k)]) * c.ahisop;
    const double csu_dzt4r = c.csu[j - 1] * dzt4r;
    double sumz = 0.0;
    for (int kr = 0; kr <= 1; ++kr)
      for (int jq = 0; jq <= 1; ++jq) {
        const double sl = drodyn(i, k, j, jq) / (drodzn(i, k, j, jq, kr) + UV_EPSLN);
        const double syn = dabs(sl);
        double a;
        if (syn > sc) {
          const double r = sc / (syn + UV_EPSLN);
          a = Ai0 * TMASK(i, k, j) * TMASK(i, k, j + 1) * (r * r);
        } else {
          a = Ai0 * TMASK(i, k, j) * TMASK(i, k, j + 1);
        }
        if (store_ai) { double *p = c.Ai_nz + (size_t)(jq + 2 * kr) * N3; UV_CYC_STORE(p, IDX, i, a); }
        sumz = sumz + c.dzw[k - 1 + kr] * a;
        cfv[CF_CN + jq + 2 * kr] = -csu_dzt4r * (a * sl);
      }
    const double k22 = dzt4r * sumz;
    UV_CYC_STORE(c.K22, IDX, i, k22);
    cfv[CF_AN] = (c.diff_cnt * c.csu_dyur[j - 1] + k22 * c.csu_dyur[j - 1]) * mm;
  }
  if (j >= 2 && k <= km - 1) {  // bottom face
    const double Ai0 = 0.5 * (c.fisop[XFIS(i, j, k + 1)] + c.fisop[XFIS(i, j, k)]) * c.ahisop;
    // the eight slopes of the bottom face have two denominators between them: one reciprocal each
    const double zb[2] = {drodzb(i, k, j, 0) + UV_EPSLN, drodzb(i, k, j, 1) + UV_EPSLN};
    const double rzb[2] = {uv_rcp2(zb[0]), uv_rcp2(zb[1])};
    double sumx = 0.0;
    for (int ip = 0; ip <= 1; ++ip)
      for (int kr = 0; kr <= 1; ++kr) {
        const double sl = uv_div2(drodxb(i, k, j, ip, kr), zb[kr], rzb[kr]);
        const double sxb = dabs(sl);
        double a;
        if (sxb > sc) {
          const double r = sc / (sxb + UV_EPSLN);
          a = Ai0 * TMASK(i, k + 1, j) * (r * r);
        } else {
          a = Ai0 * TMASK(i, k + 1, j);
        }
        if (store_ai) { double *p = c.Ai_bx + (size_t)(ip + 2 * kr) * N3; UV_CYC_STORE(p, IDX, i, a); }
        sumx = sumx + c.dxu[i - 1 + ip - 1] * a * (sxb * sxb);
        cfv[CF_CBX + ip + 2 * kr] = -c.dxt4r[i - 1] * (a * c.cstr[j - 1] * sl);
      }
    double sumy = 0.0;
    for (int jq = 0; jq <= 1; ++jq) {
      const double facty = c.csu[j - 1 + jq - 1] * c.dyu[j - 1 + jq - 1];
      for (int kr = 0; kr <= 1; ++kr) {
        const double sl = uv_div2(drodyb(i, k, j, jq, kr), zb[kr], rzb[kr]);
        const double syb = dabs(sl);
        double a;
        if (syb > sc) {
          const double r = sc / (syb + UV_EPSLN);
          a = Ai0 * TMASK(i, k + 1, j) * (r * r);
        } else {
          a = Ai0 * TMASK(i, k + 1, j);
        }
        if (store_ai) { double *p = c.Ai_by + (size_t)(jq + 2 * kr) * N3; UV_CYC_STORE(p, IDX, i, a); }
        sumy = sumy + facty * a * (syb * syb);
        cfv[CF_CBY + jq + 2 * kr] = -c.dyt4r[j - 1] * c.cstr[j - 1] * (a * c.csu[j - 1 + jq - 1] * sl);
      }
    }
    const double k33 = c.dxt4r[i - 1] * sumx + c.dyt4r[j - 1] * c.cstr[j - 1] * sumy;
    UV_CYC_STORE(c.K33, IDX, i, k33);
    // diff_cbt = background + K33 is what isopyc_column stores for this cell; a given diff_cbt (host or device
    // vmixc) is read instead -- when vmixc runs on the device after isopyc, coef_bv_cell renews this slot
    const double dcb = c.diff_cbt_given ? c.diff_cbt[q] : c.diff_cbt_bg[q] + k33;
    cfv[CF_BV] = dcb * c.dzwr[k] * (1.0 - c.aidif);
  }
#undef IDX
  if (inner) {
    double *cfa = (double *)__builtin_assume_aligned(cf, 16);
    for (int p = 0; p < CF_DPAIRS; ++p) {
      cfa[2 * ((size_t)p * N3 + q)] = cfv[2 * p];
      cfa[2 * ((size_t)p * N3 + q) + 1] = cfv[2 * p + 1];
    }
  } else {   // row 1 (and a last row, which the kernel does not visit): the north-face slots alone
    for (int p = CF_AN; p < CF_AN + 5; ++p) cf[CF_IDX(p, q, N3)] = cfv[p];
  }
}
// the vertical-diffusion coefficient alone (after a device vmixc has rewritten diff_cbt)
UVIC_DEV void coef_bv_cell(const uvic_ctx &c, double *cf, int i, int k, int j) {
  UV_DIMS(c);
  const size_t q = X3(i, k, j);
  if (j >= 2 && j <= jmt - 1 && k <= km - 1) cf[CF_IDX(CF_BV, q, N3)] = c.diff_cbt[q] * c.dzwr[k] * (1.0 - c.aidif);
}

#if defined(__HIPCC__)
// The column kernels are tolerance-tested (1e-13 one step), not bit-exact: mul+add pairs may fuse here.  The
// library is built -ffp-contract=off for the exact kernels (kernels_fct.hpp, kernels_isopyc.hpp, kernels_colx.hpp); the
// pragma holds until the matching contract(off) at the end of this block.
__device__ __forceinline__ double add_nc(double a, double b) { return a + b; }
__device__ __forceinline__ double sub_nc(double a, double b) { return a - b; }
#pragma clang fp contract(fast)
// The work list of a pass: `lanes` holds one code per lane, 64 per wave: column i (bits 0-11), row r (bits 12-23),
// bit 24 = owned (the lane stores what it computes), built on the host from kmt (uvic_gpu.hip: build_col_lanes).
// Lanes that are not owned hold a valid (i, r) all the same, so that every address stays inside its buffer.
struct ColGrid {
  const int *lanes;
  int nwaves, total;           // total = nwaves * nt_local work items (one wave each)
  int *zero_word;              // a counter the NEXT kernel on the stream wants cleared (spares a memset node), or null
};
#define COL_LANE_I(code) ((code) & 0xfff)
#define COL_LANE_R(code) (((code) >> 12) & 0xfff)
#define COL_LANE_OWNED(code) (((code) >> 24) & 1)

// Neighbour exchange by DPP whole-wave shifts (gfx9 `wave_shr:1` / `wave_shl:1`): one
// v_mov_b32_dpp per dword at VALU latency instead of an LDS round trip (ds_bpermute) --
// several of these sit on the dependency chain of every level.  The lane without a
// source (lane 0 / lane 63) reads zero (bound_ctrl, so no copy of the old value is needed);
// those lanes are halo.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  const int lo = dpp_i<CTRL>(__double2loint(v)), hi = dpp_i<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
#define DPP_WAVE_SHL1 0x130 /* lane i <- lane i+1 */
#define DPP_WAVE_SHR1 0x138 /* lane i <- lane i-1 */
__device__ __forceinline__ double shfl_w(double v) { return dpp_d<DPP_WAVE_SHR1>(v); }  // value of lane-1 (west)
__device__ __forceinline__ double shfl_e(double v) { return dpp_d<DPP_WAVE_SHL1>(v); }  // value of lane+1 (east)
// The 1-D grid metrics are written at upload time only: read them through the constant address
// space so that a wave-uniform index becomes a scalar load (s_load, own counter) rather than a
// vector load whose wait would also drain the level-ahead prefetch.
__device__ __forceinline__ double kload(const double *p, int idx) {
  return ((const __attribute__((address_space(4))) double *)p)[idx];
}
// A per-level metric table (km <= 64 entries) held one entry per lane and broadcast with v_readlane:
// no memory latency on the level-to-level dependency chain (a scalar load there costs ~200 cycles,
// twice that when its pointer has to be re-read from the kernel arguments first).
struct LaneTable {
  int lo, hi;
  __device__ __forceinline__ void load(const double *p, int n) {
    const int l = threadIdx.x & 63;
    const double v = (l < n) ? kload(p, l) : 0.0;
    lo = __double2loint(v); hi = __double2hiint(v);
  }
  __device__ __forceinline__ double at(int idx) const {   // idx wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(hi, idx), __builtin_amdgcn_readlane(lo, idx));
  }
};
// Buffer addressing (`buffer_load_dwordx2 v, v_off, s[rsrc], s_off offen`): address = descriptor base + a
// wave-uniform byte offset in ONE scalar register + the lane's 32-bit byte offset, formed by the memory
// pipeline.  The flat form costs a 64-bit VALU add per load and two scalar adds per pointer.  num_records
// bounds the lane offset, so a lane that strays reads zero instead of faulting.
typedef unsigned uv2 __attribute__((ext_vector_type(2)));
typedef unsigned uv4 __attribute__((ext_vector_type(4)));
typedef double dv2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t brsrc;
__device__ __forceinline__ brsrc mkbuf(const void *p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)(bytes > 0x7fffffffu ? 0x7fffffffu : bytes), 0x00020000);
}
__device__ __forceinline__ double bld(brsrc r, unsigned voff, int soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ double2 bld2(brsrc r, unsigned voff, int soff) {
  const dv2 v = __builtin_bit_cast(dv2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  return make_double2(v.x, v.y);
}
__device__ __forceinline__ void bst(brsrc r, unsigned voff, int soff, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uv2, v), r, voff, soff, 0);
}
// v_max_f64 / v_min_f64: one instruction instead of compare + two selects (operands are never NaN here)
__device__ __forceinline__ double fmx(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ double fmn(double a, double b) { return __builtin_fmin(a, b); }
// HALF of the upstream flux v*(a+b) + |v|*(a-b) of adv_flx:501-514: v*a for v >= 0, v*b otherwise (equal to rounding)
__device__ __forceinline__ double hup(double v, double a, double b) { return v * ((v >= 0.0) ? a : b); }
// x / y for the limiter ratios.  x is finite, y = P + epsln lies in [1e-20, ~1e6]: the range handling of the IEEE
// division sequence (two v_div_scale, v_div_fmas, v_div_fixup) is dead weight here.  v_rcp_f64 (relative error
// 2^-23), ONE Newton step (2^-46) and one correction of the quotient (error (2^-46)^2, below the final rounding):
// 6 instructions instead of 12, result within 1 ulp of x / y.
__device__ __forceinline__ double div_pos(double x, double y) {
  double r = __builtin_amdgcn_rcp(y);
  r = __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
  const double q = x * r;
  return __builtin_fma(__builtin_fma(-y, q, x), r, q);   // (without this correction the ratios carry 2^-46 and the step gains 0.7 %: not taken)
}
// Zalesak's limiter for one cell (tracer_adv_flx.F:672-690), WITHOUT the clamp min(1, .): the face limiter is
// min(R+ of one cell, R- of the other) = min(1, q+, q-), so the clamp is taken once per face (`limited`) instead of
// twice per cell.  flxlft, flxrgt are half fluxes and `scale` carries the factor 2.
__device__ __forceinline__ void fct_ratio(double fxa, double fxb, double tlo, double scale, double flxlft, double flxrgt,
                                          double mask, double &qp, double &qm) {
  const double trmax = fmx(fmx(fxa, fxb), tlo), trmin = fmn(fmn(fxa, fxb), tlo);
  const double pplus = scale * (fmx(0.0, flxlft) - fmn(0.0, flxrgt));
  const double pminus = scale * (fmx(0.0, flxrgt) - fmn(0.0, flxlft));
  qp = div_pos(mask * (trmax - tlo), pplus + UV_EPSLN);
  qm = div_pos(mask * (tlo - trmin), pminus + UV_EPSLN);
}
// 0.5*((cpos+cneg)*f + (cpos-cneg)*|f|) of adv_flx:703-705 is cpos*f for f >= 0 and cneg*f otherwise, with
// cpos = min(R+ downstream, R- upstream): (qp_dn, qm_up) are the unclamped ratios for f >= 0, (qp_up, qm_dn) for f < 0
__device__ __forceinline__ double limited(double qp_dn, double qm_up, double qp_up, double qm_dn, double f) {
  const bool pos = f >= 0.0;
  return fmn(1.0, fmn(pos ? qp_dn : qp_up, pos ? qm_up : qm_dn)) * f;
}

// ===========================================================================
// pass A: one sweep down the column, one tracer per wave
//
// BULK (the launch of the tracers other than T and S; the four waves of a workgroup are four tracers of the SAME lanes):
//   * what does not depend on the tracer comes through LDS: per level 10 pair planes of folded coefficients of row r, 3 of
//     row r-1 (its north-face ones), the (tot_e, tot_n) and (tot_b, tot_n of row j-1) pairs of row r and of row r+1: 17
//     slots of 64 x 16 B.  Each wave brings a quarter of the NEXT level's slots with `buffer_load ... lds` (no registers),
//     one workgroup barrier per level, every wave reads all 17 with ds_read_b128.
//   * the wave's own loads (t of four rows at both time levels, 8 per level) are issued one level ahead into a second
//     register set: neither wave of a SIMD waits for memory inside a level.
// !BULK (T and S when every tracer goes through the column kernels, `set_exact(2)`): every load is the wave's own.
//
// The wave also forms t_lo and the y-limiter ratios of the row to its NORTH (from t of rows r+1, r+2 and that row's
// velocities) and with them the FINAL, limited advective flux through its north face, which it stores (8 bytes per cell
// update): pass B reads that flux of rows r and r-1 and nothing else of the y direction.
// ===========================================================================
#define COL_SHARE_SLOTS (CF_PAIRS + 5)   /* row r: all pairs; row r-1: the north-face ones; row r+1: the velocity pairs */
template <bool BULK>
__device__ __forceinline__ void colfct_wave(const uvic_ctx &c, const double *__restrict__ cf, double *__restrict__ S,
                                            int code, int n1, bool live, double *lds = nullptr, int wv = 0) {
  constexpr int SLOTS = COL_SHARE_SLOTS;
  UV_DIMS(c);
  const int i = COL_LANE_I(code), r = COL_LANE_R(code);
  const bool owned = COL_LANE_OWNED(code) != 0 && live;
  // per-level metrics, one entry per lane, broadcast by v_readlane (no memory latency in the march)
  LaneTable t_dtxcel, t_dztr;
  t_dtxcel.load(c.dtxcel, km); t_dztr.load(c.dztr, km);
  const int kz = c.kmt[X2(i, r)], kz_s = c.kmt[X2(i, r - 1)], kz_n = c.kmt[X2(i, r + 1)];
  // across the seam between two runs of the lane map the neighbour is not the x-neighbour: only halo lanes look there
  const int kz_w = dpp_i<DPP_WAVE_SHR1>(kz), kz_e = dpp_i<DPP_WAVE_SHL1>(kz);
  const double cstdxtr = c.cstr[r - 1] * c.dxtr[i - 1], cstdytr = c.cstdytr[r - 1];
  const bool south_wall = (r - 1 == 1);   // no antidiffusive flux through the face to row 1 (adv_flx: jstrt)
  const double c2dtts = c.c2dtts;
  // the row to the north (N = r+1 <= jmt) and the one beyond it (the reference clamps: jp2 = min(j+2, jmt), adv_flx:555)
  const int rnn = imin(r + 2, jmt);
  const int kz_nn = c.kmt[X2(i, rnn)];
  const double cstdxtr_N = c.cstr[r] * c.dxtr[i - 1], cstdytr_N = c.cstdytr[r];
  // addresses = buffer descriptor + wave-uniform byte offset (one scalar register: level and row shift) + the lane's
  // 32-bit offset of its (i, r) column (pointing at row r-1, so that the scalar offset is never negative: the hardware adds it unsigned)
  const int rowstride = imt * km;
  const unsigned lb = (unsigned)((r - 2) * rowstride + (i - 1)) * 8u;   // level 1 of column i of row r-1, cell fields
  const unsigned lb2 = lb * 2u;
  const unsigned lbf = (unsigned)((r - 1) * imt * (km + 1) + (i - 1)) * 8u;   // face 0 of the lane's column, face fields
  const unsigned lb_nn = (unsigned)((rnn - 1) * rowstride + (i - 1)) * 8u;     // level 1 of column i of row min(r+2, jmt)
  const unsigned lbf_N = lbf + (unsigned)(imt * (km + 1)) * 8u;               // face 0 of the column to the north
  const brsrc b_te = mkbuf(c.tot_e, N3 * 8), b_tn = mkbuf(c.tot_n, N3 * 8);
  const brsrc b_tb = mkbuf(c.tot_b, NF * 8), b_vb = mkbuf(c.adv_vbt, NF * 8);
  const brsrc b_cf = mkbuf(cf, N3 * 16 * CF_PAIRS);
  const size_t nloc = (size_t)(n1 - 1 - c.n0);
  const brsrc b_tm = mkbuf(c.t_taum1 + (size_t)(n1 - 1) * N3, N3 * 8), b_tt = mkbuf(c.t_tau + (size_t)(n1 - 1) * N3, N3 * 8);
  const brsrc b_S = mkbuf(S + nloc * N3, N3 * 8), b_fy = mkbuf(c.fny + nloc * N3, N3 * 8);
  const double stf = c.stf[X2(i, r) + (size_t)(n1 - 1) * imt * jmt] * (1.0 - c.aidif);
  const double btf = c.btf[X2(i, r) + (size_t)(n1 - 1) * imt * jmt] * (1.0 - c.aidif);
#define OC(k, dj) (((((k)-1) * imt + ((dj) + 1) * rowstride)) * 8)   /* byte offset of level k of row r+dj from the lane's offset */
#define CFP(pair, k, dj) bld2(b_cf, lb2, ((int)(pair) * (int)N3 + ((k)-1) * imt + ((dj) + 1) * rowstride) * 16)
#define OF(kf) (((kf) * imt) * 8)
  const double vb0 = bld(b_vb, lbf, OF(0));
  // state carried from level to level
  double mc1 = bld(b_tm, lb, OC(1, 0)), ms1 = bld(b_tm, lb, OC(1, -1)), mn1 = bld(b_tm, lb, OC(1, 1));   // level s of t(tau-1): centre, south, north
  double tc1 = bld(b_tt, lb, OC(1, 0));                    // level s of t(tau)
  double hb_up = vb0 * mc1, hab_up = hb_up;               // half low-order and raw antidiffusive flux through the face above level s (adv_flx:543, 617)
  double hbfin_up = vb0 * tc1;                             // half of the FINAL advective flux through that face (top: tracer.F:1063)
  double hbN_up = bld(b_vb, lbf_N, OF(0)) * mn1;           // half low-order flux through the face above level s of the column to the north
  double mz_up = 0.0;                                      // mean of t(tau) across the face above level s
  double dfb_up = stf, dfbi_up = 0.0;                      // diffusive fluxes through the face above level s
  double qzp_prev = 0.0, qzm_prev = 0.0, spart_prev = 0.0, mk_prev = 0.0;
  // differences that level s+1 needs again are handed down instead of being formed (and shuffled) twice:
  // T(s)-T(s+1) of the own, east, south and north columns; T(i+1)-T(i) and T(i)-T(i-1) at level s+1
  double me_next = shfl_e(mc1), dz_c = 0.0, dz_e = 0.0, dz_s = 0.0, dz_n = 0.0;
  double dx_next = me_next - mc1, dxw_next = shfl_w(dx_next);
  // everything level s reads from memory
  struct LvlIn {
    double ve, vn, vs, vb, veN, vnN, vbN;
    double cfc[2 * CF_DPAIRS], cfs[6];   // !BULK: folded coefficients of row r and the north-face ones (slots 0..4) of row r-1
    double mc2, ms2, mn2, tc2, t_s, t_n, m_nn, t_nn;
  };
  auto load_in = [&](LvlIn &L, int s) {
    const int sp = (s == km) ? km : s + 1;
    if (!BULK) {
      L.ve = bld(b_te, lb, OC(s, 0)); L.vn = bld(b_tn, lb, OC(s, 0)); L.vs = bld(b_tn, lb, OC(s, -1));
      L.vb = (s < km) ? bld(b_tb, lbf, OF(s)) : bld(b_vb, lbf, OF(km));
      L.veN = bld(b_te, lb, OC(s, 1)); L.vnN = bld(b_tn, lb, OC(s, 1));
      L.vbN = (s < km) ? bld(b_tb, lbf_N, OF(s)) : 0.0;
      _Pragma("unroll") for (int p = 0; p < CF_DPAIRS; ++p) {
        const double2 v = CFP(p, s, 0);
        L.cfc[2 * p] = v.x; L.cfc[2 * p + 1] = v.y;
      }
      _Pragma("unroll") for (int p = 0; p < 3; ++p) {
        const double2 v = CFP(p, s, -1);
        L.cfs[2 * p] = v.x; L.cfs[2 * p + 1] = v.y;
      }
    }
    L.mc2 = bld(b_tm, lb, OC(sp, 0)); L.ms2 = bld(b_tm, lb, OC(sp, -1)); L.mn2 = bld(b_tm, lb, OC(sp, 1));
    L.tc2 = bld(b_tt, lb, OC(sp, 0)); L.t_s = bld(b_tt, lb, OC(s, -1)); L.t_n = bld(b_tt, lb, OC(s, 1));
    L.m_nn = bld(b_tm, lb_nn, OC(s, -1)); L.t_nn = bld(b_tt, lb_nn, OC(s, -1));   // (lb_nn points at the row itself)
  };
  auto level = [&](const LvlIn &L, int s) {
    const bool last = (s == km);
    // ---- what the tracers share: velocities, folded coefficients, masks, metrics ------------------------
    double ve, vn, vs, vb, veN, vnN, vbN;
    const double2 *sl = BULK ? (const double2 *)lds + (size_t)(s & 1) * SLOTS * 64 + (threadIdx.x & 63) : nullptr;
    if (BULK) {
      const double2 a = sl[(CF_VE / 2) * 64], b = sl[(CF_VB / 2) * 64];
      ve = a.x; vn = a.y; vb = b.x; vs = b.y;
      const double2 aN = sl[(CF_PAIRS + 3) * 64], bN = sl[(CF_PAIRS + 4) * 64];
      veN = aN.x; vnN = aN.y; vbN = last ? 0.0 : bN.x;
    } else {
      ve = L.ve; vn = L.vn; vs = L.vs; vb = L.vb; veN = L.veN; vnN = L.vnN; vbN = L.vbN;
    }
    const double ddztr = t_dztr.at(s - 1);
    const double ddztr_up = (s >= 2) ? t_dztr.at(s - 2) : 0.0;
    const double mk = (s <= kz) ? 1.0 : 0.0;
    const double twodt_mk = c2dtts * t_dtxcel.at(s - 1) * mk;
    // wet neighbour -> its face value, land -> t_lo (adv_flx:640-668 blends with the 0/1 mask: the same value)
    const bool wet_w = s <= kz_w, wet_e = s <= kz_e, wet_s = s <= kz_s, wet_n = s <= kz_n;
    const bool wet_up = s - 1 >= 1 && s - 1 <= kz, wet_dn = s + 1 <= kz;
    const double mc2 = L.mc2, ms2 = L.ms2, mn2 = L.mn2;
    const double m_c = mc1;
    const double mc2_e = shfl_e(mc2);
    // =================== advective part (adv_flx) =====================================================
    const double tc2 = L.tc2, t_s = L.t_s, t_n = L.t_n;
    const double tt_c = tc1;
    const double m_e = me_next, tt_e = shfl_e(tt_c);
    // face means of t(tau): the centred flux is v times them, the limiter takes them as the neighbours' values
    const double me = 0.5 * (tt_c + tt_e), mw = shfl_w(me);
    const double mn_ = 0.5 * (tt_c + t_n), ms_ = 0.5 * (t_s + tt_c);
    const double mz_dn = 0.5 * (tt_c + tc2);
    // ---- low order and raw antidiffusive fluxes, halved (adv_flx:500-619) ----------
    const double he = hup(ve, m_c, m_e);
    const double hae = ve * me - he;
    const double he_w = shfl_w(he), hae_w = shfl_w(hae);
    const double hn = hup(vn, m_c, mn1), hs = hup(vs, ms1, m_c);
    double hb = 0.0, hab = 0.0;
    if (!last) {
      hb = hup(vb, mc2, m_c);
      hab = vb * mz_dn - hb * mk;
    }
    const double advx = (he - he_w) * cstdxtr, advy = (hn - hs) * cstdytr;
    const double advz = (hb_up - hb) * ddztr;
    const double tlo = m_c - twodt_mk * (advx + advy + advz);
    // ---- limiter ratios (unclamped) ---------------------------------------------
    double qxp, qxm, qyp, qym, qzp, qzm;
    fct_ratio(wet_w ? mw : tlo, wet_e ? me : tlo, tlo, c2dtts * cstdxtr, hae_w, hae, mk, qxp, qxm);
    const double han = vn * mn_ - hn;
    {
      const double has = south_wall ? 0.0 : vs * ms_ - hs;
      fct_ratio(wet_s ? ms_ : tlo, wet_n ? mn_ : tlo, tlo, c2dtts * cstdytr, has, han, mk, qyp, qym);
    }
    fct_ratio(wet_up ? mz_up : tlo, (!last && wet_dn) ? mz_dn : tlo, tlo, c2dtts * ddztr, hab, hab_up, mk, qzp, qzm);
    {
      // ---- the row to the north: low-order fluxes, t_lo, y-limiter ratios (the same formulas one row up) ----------
      const double m_nn = L.m_nn, t_nn = L.t_nn;
      const double m_N = mn1;
      const double mkN = (s <= kz_n) ? 1.0 : 0.0;
      const double heN = hup(veN, m_N, shfl_e(m_N));
      const double hnN = hup(vnN, m_N, m_nn);
      double hbN = 0.0;
      if (!last) hbN = hup(vbN, mn2, m_N);
      const double advN = (heN - shfl_w(heN)) * cstdxtr_N + (hnN - hn) * cstdytr_N + (hbN_up - hbN) * ddztr;
      const double tloN = m_N - c2dtts * t_dtxcel.at(s - 1) * mkN * advN;
      const double mnn_ = 0.5 * (t_n + t_nn);
      const double hann = vnN * mnn_ - hnN;
      double qypN, qymN;
      fct_ratio(mk != 0.0 ? mn_ : tloN, (s <= kz_nn) ? mnn_ : tloN, tloN, c2dtts * cstdytr_N, han, hann, mkN, qypN, qymN);
      // ---- the limited flux through the north face, final (adv_flx:770-783, 994-996) --------------------------------
      const double hn_fin = (limited(qypN, qym, qyp, qymN, han) + hn) * mk;
      if (owned) bst(b_fy, lb, OC(s, 0), hn_fin);
      hbN_up = hbN;
    }
    // ---- limited x flux and its divergence (adv_flx:695-711, 989-992) ---------------
    const double qxp_e = shfl_e(qxp), qxm_e = shfl_e(qxm);
    const double hefin = limited(qxp_e, qxm, qxp, qxm_e, hae) + he;
    double spart = -((hefin - shfl_w(hefin)) * cstdxtr);     // -ADV_Tx; what this level adds to S but for the z advection (finalised one level later)
    // ---- finalise level s-1: limited z flux through the face between s-1 and s (adv_flx:857-887, 994-999)
    double sfin_prev = 0.0;
    if (s >= 2) {
      const double hbfin = (limited(qzp_prev, qzm, qzp, qzm_prev, hab_up) + hb_up) * mk_prev;
      const double ADV_Tz = (hbfin_up - hbfin) * ddztr_up;
      sfin_prev = sub_nc(spart_prev, ADV_Tz);
      hbfin_up = hbfin;
    }
    double adz_last = 0.0;
    if (last) adz_last = (hbfin_up - 0.5 * (vb * tt_c)) * ddztr;   // bottom face of the column (tracer.F:1065); vb holds adv_vbt there
    qzp_prev = qzp; qzm_prev = qzm;
    hb_up = hb; hab_up = hab; mz_up = mz_dn;
    me_next = mc2_e;
    tc1 = tc2;
    // =================== diffusive part (coefficients folded by ai_coef_cell) ===========================
    double cfc_l[2 * CF_PAIRS], cfs_l[6];
    if (BULK) {
      _Pragma("unroll") for (int p = 0; p < CF_DPAIRS; ++p) {
        const double2 v = sl[p * 64];
        cfc_l[2 * p] = v.x; cfc_l[2 * p + 1] = v.y;
      }
      _Pragma("unroll") for (int p = 0; p < 3; ++p) {
        const double2 v = sl[(CF_PAIRS + p) * 64];
        cfs_l[2 * p] = v.x; cfs_l[2 * p + 1] = v.y;
      }
    }
    const double *cfc = BULK ? cfc_l : L.cfc, *cfs = BULK ? cfs_l : L.cfs;
    const double dz_up = dz_c, dz_dn = (!last) ? m_c - mc2 : 0.0;          // own column (dz_up = dz_dn of the level above, 0 at the top)
    const double dze_up = dz_e, dze_dn = shfl_e(dz_dn);                      // east column
    const double dzs_up = dz_s, dzs_dn = (!last) ? ms1 - ms2 : 0.0;        // south row
    const double dzn_up = dz_n, dzn_dn = (!last) ? mn1 - mn2 : 0.0;        // north row
    const double dx_c = dx_next, dx_d = mc2_e - mc2;                         // T(i+1)-T(i) at levels s, s+1
    const double dxw_c = dxw_next, dxw_d = shfl_w(dx_d);                     // T(i)-T(i-1)
    const double dfe = cfc[CF_AE] * dx_c + cfc[CF_CE + 0] * dz_up + cfc[CF_CE + 1] * dze_up + cfc[CF_CE + 2] * dz_dn +
                       cfc[CF_CE + 3] * dze_dn;
    const double DIFF_Tx = (dfe - shfl_w(dfe)) * cstdxtr;
    const double dfn_n = cfc[CF_AN] * (mn1 - m_c) + cfc[CF_CN + 0] * dz_up + cfc[CF_CN + 1] * dzn_up + cfc[CF_CN + 2] * dz_dn +
                         cfc[CF_CN + 3] * dzn_dn;
    const double dfn_s = cfs[0] * (m_c - ms1) + cfs[1] * dzs_up + cfs[2] * dz_up + cfs[3] * dzs_dn + cfs[4] * dz_dn;
    const double DIFF_Ty = (dfn_n - dfn_s) * cstdytr;
    double dfb = 0.0, dfbi = 0.0;  // through the face below level s
    if (!last) {
      dfb = cfc[CF_BV] * (m_c - mc2);
      dfbi = cfc[CF_CBX + 0] * dxw_c + cfc[CF_CBX + 1] * dx_c + cfc[CF_CBX + 2] * dxw_d + cfc[CF_CBX + 3] * dx_d +
             cfc[CF_CBY + 0] * (m_c - ms1) + cfc[CF_CBY + 1] * (mn1 - m_c) + cfc[CF_CBY + 2] * (mc2 - ms2) +
             cfc[CF_CBY + 3] * (mn2 - mc2);
    }
    if (s == kz) dfb = btf;  // bottom boundary condition of the explicit vertical flux (tracer.F:1060-1062)
    if (kz == 0 && s == 1) dfb_up = btf;
    const double DIFF_Tz = (dfb_up - dfb) * ddztr + (dfbi_up - dfbi) * ddztr;
    spart = add_nc(spart, DIFF_Tx + DIFF_Ty + DIFF_Tz);
    dfb_up = dfb; dfbi_up = dfbi;
    dx_next = dx_d; dxw_next = dxw_d;
    dz_c = dz_dn; dz_e = dze_dn; dz_s = dzs_dn; dz_n = dzn_dn;
    // ---- stores of S -----------------------------------------------------------------------
    if (s >= 2 && owned) bst(b_S, lb, OC(s - 1, 0), sfin_prev);
    if (last && owned) bst(b_S, lb, OC(km, 0), sub_nc(spart, adz_last));
    spart_prev = spart;
    mk_prev = mk;
    mc1 = mc2; ms1 = ms2; mn1 = mn2;
  };
  if (BULK) {
    // this wave's share of the pair slots of level s: slots wv, wv+4, ... of the 17 (row r: 0..11, row r-1: 12..14, row r+1: 15, 16)
    auto bring = [&](int s) {
      typedef __attribute__((address_space(3))) void *ldsp;
      _Pragma("unroll") for (int q = 0; q < (SLOTS + 3) / 4; ++q) {
        const int p = wv + 4 * q;
        if (p < SLOTS) {
          double2 *dst = (double2 *)lds + ((size_t)(s & 1) * SLOTS + p) * 64;
          const int pair = p < CF_PAIRS ? p : (p < CF_PAIRS + 3 ? p - CF_PAIRS : p - (CF_PAIRS + 3) + CF_VE / 2);
          const int dj = p < CF_PAIRS ? 0 : (p < CF_PAIRS + 3 ? -1 : 1);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(b_cf, (ldsp)dst, 16, lb2, ((int)(pair) * (int)N3 + (s - 1) * imt + (dj + 1) * rowstride) * 16, 0, 0);
        }
      }
    };
    // No branch around a load (the compiler's wait counts stay exact on straight-line code only): the level index is
    // clamped instead (the last pair re-reads level km) and an odd last level is peeled.  Before a level: its pairs have
    // landed (this wave's share: vmcnt; the others': the barrier) and every wave has finished reading the other buffer.
    LvlIn A, B;
    bring(1);
    load_in(A, 1);
    int s = 1;
    for (; s + 1 <= km; s += 2) {
      __builtin_amdgcn_s_waitcnt(0x0f70);   /* vmcnt(0), lgkmcnt and expcnt left alone */
      __builtin_amdgcn_s_barrier();
      bring(s + 1);
      load_in(B, s + 1);
      level(A, s);
      __builtin_amdgcn_s_waitcnt(0x0f70);
      __builtin_amdgcn_s_barrier();
      bring(imin(s + 2, km));          // (an odd km: the pairs of the peeled last level; an even one: level km once more, unused)
      load_in(A, imin(s + 2, km));
      level(B, s + 1);
    }
    if (s == km) {
      __builtin_amdgcn_s_waitcnt(0x0f70);
      __builtin_amdgcn_s_barrier();
      level(A, km);
    }
  } else {
    for (int s = 1; s <= km; ++s) {
      LvlIn L;
      load_in(L, s);
      level(L, s);
    }
  }
#undef CFP
#undef OC
#undef OF
}

// ===========================================================================
// pass B: y advection from the final fluxes pass A left, explicit update, implicit vertical diffusion (invtri.F)
// `ework` is the wave's LDS scratch: e(k) and z(k) of the Thomas recurrence, each (km+1) x 64 doubles laid out
// [k][lane], so that the forward sweep writes t(tau+1) nowhere and the back substitution stores it once.
// keep_lds: the back substitution also leaves t(tau+1) of the column in LDS (zwork[k][lane], k = 1..km), for a caller
// that goes on with it (T and S through the column kernels: the convective walk of the same workgroup)
// ===========================================================================
__device__ __forceinline__ void colupd_wave(const uvic_ctx &c, const double *__restrict__ S, double *ework, int code, int n1,
                                            bool keep_lds = false) {
  UV_DIMS(c);
  const int lane = threadIdx.x;
  const int i = COL_LANE_I(code), r = COL_LANE_R(code);
  LaneTable t_dtxcel, t_dztur, t_dztlr, t_dztr;   // filled while every lane is still active
  t_dtxcel.load(c.dtxcel, c.km); t_dztur.load(c.dztur, c.km); t_dztlr.load(c.dztlr, c.km); t_dztr.load(c.dztr, c.km);
  // the lanes are ocean columns; land keeps the zeros it was given once (uvic_gpu.hip: land_clean)
  if (!COL_LANE_OWNED(code)) return;   // padding of the last wave
  const size_t nloc = (size_t)(n1 - 1 - c.n0);
  double *tp = c.t_taup1 + (size_t)(n1 - 1) * N3;
  const double *Sn = S + nloc * N3;
  // the source term is read here, not in pass A: with MOBI computed one step ahead on the side
  // stream only this pass has to wait for it
  const double *source = 0;
  if (c.src && c.itrc[n1 - 1] != 0) source = c.src + (size_t)(c.itrc[n1 - 1] - 1) * N3;
  const int kz = c.kmt[X2(i, r)];
  const double cstdytr = c.cstdytr[r - 1];
  const int rowstride = imt * km;
  // the lane's column in row r-1; the scalar offset shifts level and row and is never negative (the hardware adds it unsigned)
  const unsigned lb = (unsigned)((r - 2) * rowstride + (i - 1)) * 8u;
#define OC(k, dj) (((((k)-1) * imt + ((dj) + 1) * rowstride)) * 8)
  const double topbc = c.stf[X2(i, r) + (size_t)(n1 - 1) * imt * jmt], botbc = c.btf[X2(i, r) + (size_t)(n1 - 1) * imt * jmt];
  const double aidif = c.aidif, eps = 1.e-30;
  const int kb = imax(2, kz);
  double bet = 0.0, zprev = 0.0, cprev = 0.0;
  const int ic = (i == 2) ? imt : ((i == imt - 1) ? 1 : 0);
  double *zwork = ework + (size_t)(km + 1) * 64;
  // source term: always loaded (from S when the tracer has none) and selected afterwards, so that the
  // number of loads in flight is the same on every path and the waits stay exact
  const bool has_src = source != 0;
  const brsrc b_tm = mkbuf(c.t_taum1 + (size_t)(n1 - 1) * N3, N3 * 8), b_fy = mkbuf(c.fny + nloc * N3, N3 * 8);
  const brsrc b_S = mkbuf(Sn, N3 * 8), b_src = mkbuf(has_src ? source : Sn, N3 * 8), b_dcb = mkbuf(c.diff_cbt, N3 * 8), b_tp = mkbuf(tp, N3 * 8);
  // The sweep is one dependent chain per column (Thomas recurrence): what a level reads is fetched one
  // level ahead into the other of two register sets so that the chain never waits for memory.
  struct Lvl {
    double m_c, fy0, fys, sn, src, dcb;
  };
  auto load_level = [&](Lvl &L, int k) {
    L.m_c = bld(b_tm, lb, OC(k, 0));
    L.fy0 = bld(b_fy, lb, OC(k, 0)); L.fys = bld(b_fy, lb, OC(k, -1));   // the final (half) fluxes through the north faces of rows r, r-1
    L.sn = bld(b_S, lb, OC(k, 0));
    L.src = bld(b_src, lb, OC(k, 0));
    L.dcb = bld(b_dcb, lb, OC(k, 0));
  };
  double dcb_up = 0.0;   // diff_cbt of the level above (the reference reads level max(1,k-1); at k=1 its factor is zeroed)
  auto level = [&](const Lvl &L, int k) {
    const double m_c = L.m_c;
    const double mk = (k <= kz) ? 1.0 : 0.0;
    const double ADV_Ty = (L.fy0 - L.fys) * cstdytr;
    const double tdt = c.c2dtts * t_dtxcel.at(k - 1);
    const double z = m_c + tdt * (L.sn - ADV_Ty + (has_src ? L.src : 0.0)) * mk;
    // Thomas forward sweep, invtri.F:57-100
    const int kp1 = imin(k + 1, km);
    const double factu = t_dztur.at(k - 1) * tdt * aidif, factl = t_dztlr.at(k - 1) * tdt * aidif;
    double a = -((k == 1) ? L.dcb : dcb_up) * factu * mk;
    double cc = -L.dcb * factl * ((kp1 <= kz) ? 1.0 : 0.0);
    double f = z * mk;
    if (k == 1) a = 0.0;
    if (k == km) cc = 0.0;
    const double b = 1.0 - a - cc;
    if (k == 1) f = z + topbc * tdt * t_dztr.at(0) * aidif * mk;
    if (k == kb) f = z - botbc * tdt * t_dztr.at(k - 1) * aidif * mk;
    double znew;
    if (k == 1) {
      bet = div_pos(mk, b + eps);   // the pivots of this diagonally dominant system are >= 1 (a, c <= 0, b = 1 - a - c)
      znew = f * bet;
    } else {
      const double e = cprev * bet;
      ework[(size_t)k * 64 + lane] = e;
      bet = div_pos(mk, b - a * e + eps);
      znew = (f - a * zprev) * bet;
    }
    zwork[(size_t)k * 64 + lane] = znew;
    zprev = znew;
    cprev = cc;
    dcb_up = L.dcb;
  };
  {
    // no branch around a load: the compiler's wait counts stay exact only on straight-line code, so the
    // level index is clamped instead (the last pair re-reads level km) and an odd last level is peeled
    Lvl A, B;
    load_level(A, 1);
    int k = 1;
    for (; k + 1 <= km; k += 2) {
      load_level(B, k + 1);
      level(A, k);
      load_level(A, imin(k + 2, km));
      level(B, k + 1);
    }
    if (k == km) level(A, km);
  }
  // back substitution, invtri.F:104-110, and the cyclic images (tracer.F:1153-1155)
  double znext = zprev;
  bst(b_tp, lb, OC(km, 0), znext);
  if (ic) tp[X3(ic, km, r)] = znext;
  for (int k = km - 1; k >= 1; --k) {
    const double zk = zwork[(size_t)k * 64 + lane] - ework[(size_t)(k + 1) * 64 + lane] * znext;
    bst(b_tp, lb, OC(k, 0), zk);
    if (ic) tp[X3(ic, k, r)] = zk;
    if (keep_lds) zwork[(size_t)k * 64 + lane] = zk;
    znext = zk;
  }
}
#undef OC
#pragma clang fp contract(off)
#endif  // __HIPCC__

}  // namespace uvic
#endif
