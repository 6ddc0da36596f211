// kernels_col.hpp -- "lane per column" transport kernels (the production path).
//
// Same physics as kernels_fct.hpp (FCT adv_flux, isoflux, explicit update,
// invtri; reference lines cited there), laid out for the CDNA4 wavefront:
//
//   one lane  = one (i, r) ocean column of one tracer; the lane marches down k = 1..km keeping
//               the k-1,k,k+1 window of both time levels in registers
//   one wave  = 64 lanes taken from a host-built LANE MAP (uvic_gpu.hip: build_col_lanes): the
//               ocean columns of the slab row by row, so that lanes are adjacent columns of a row
//               wherever the sea is; land columns, the polar caps and the padding of a fixed
//               60-column segmentation get no lanes at all
//   pass A: x-neighbours come from the adjacent lanes by DPP whole-wave shifts, so every run of
//               adjacent ocean columns carries two halo lanes on each side (not owned: they
//               compute, their results are not stored); a wave may hold pieces of several runs
//               and rows.  Rows r-1, r+1 come from the lane's own coalesced loads; no LDS, no
//               barriers.  Longitude wraps cyclically.
//   pass B: no neighbour exchange at all: the lanes are the ocean columns and nothing else.
//
// The isopycnal flux terms are linear in the tracer with coefficients that do not
// depend on the tracer: `ai_coef_cell` folds Ai * slope (and metric factors, masks,
// background diffusivities) into 19 per-face coefficients ONCE per step, so the
// 24 fp64 divisions per cell update of the reference formulation disappear from
// the per-tracer work.  This re-associates floating-point products:
//     reference  ((Ai*dT)*drodx)/(drodz+eps)      here  (Ai*drodx/(drodz+eps))*dT
// so results agree with the reference to rounding (tested: <= 1e-12 relative
// after 20 and 100 steps, tests/test_gpu_fast.py), not bit for bit.  The
// bit-exact formulation stays available (kernels_fct.hpp, UVIC_EXACT=1).
//
// Pass A (`colfct_wave`): low-order fluxes, t_lo, limiter ratios, limited x and z
//   fluxes, all diffusive fluxes -> S = the explicit tendency except the y
//   advection and the source term, and the y-limiter ratios R+-Y.
// Pass B (`colupd_wave`): limited y fluxes from R+-Y(r-1..r+1), explicit update,
//   tridiagonal solve -> t(tau+1).
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
    double sumx = 0.0;
    for (int ip = 0; ip <= 1; ++ip)
      for (int kr = 0; kr <= 1; ++kr) {
        const double sl = drodxb(i, k, j, ip, kr) / (drodzb(i, k, j, kr) + UV_EPSLN);
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
        const double sl = drodyb(i, k, j, jq, kr) / (drodzb(i, k, j, kr) + UV_EPSLN);
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
// library is built -ffp-contract=off for the exact kernels (kernels_fct.hpp, kernels_isopyc.hpp); the pragma
// holds until the matching contract(off) at the end of this block.  -DUV_NO_CONTRACT: measurement only.
// Sums that join the two halves of pass A (advective, diffusive) never fuse with the products they add up: the one-sweep
// and the two-sweep form of the pass then give the same bits (the halves meet in registers in one, in memory in the other).
__device__ __forceinline__ double add_nc(double a, double b) { return a + b; }
__device__ __forceinline__ double sub_nc(double a, double b) { return a - b; }
#ifndef UV_NO_CONTRACT
#pragma clang fp contract(fast)
#endif
// The work list of a pass: `lanes` holds one code per lane, 64 per wave: column i (bits 0-11), row r (bits 12-23),
// bit 24 = owned (the lane stores what it computes), built on the host from kmt (uvic_gpu.hip: build_col_lanes).
// Lanes that are not owned hold a valid (i, r) all the same, so that every address stays inside its buffer.
struct ColGrid {
  const int *lanes;
  int nwaves, total;           // total = nwaves * nt_local work items (one wave each)
  int fuse_convect;            // pass B: replay the convective mixing found by the T,S walk before t(tau+1) is stored
  int *zero_word;              // pass B of T,S: a counter the NEXT kernel on the stream wants cleared (spares a memset node)
};
#define COL_LANE_I(code) ((code) & 0xfff)
#define COL_LANE_R(code) (((code) >> 12) & 0xfff)
#define COL_LANE_OWNED(code) (((code) >> 24) & 1)
#ifndef COLUPD_WAVES
#define COLUPD_WAVES 1
#endif //  // waves per workgroup of pass B: 2 x (km+1) x 512 B of LDS each (20 KB at km = 19), so that workgroups still fit beside a MOBI team (90 KB) on a CU

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
// pipeline.  The flat form costs a 64-bit VALU add per load (v_lshl_add_u64: 25 of ~400 VALU instructions
// per level of pass A) and two scalar adds per pointer.  num_records bounds the lane offset, so a lane
// that strays reads zero instead of faulting.
typedef unsigned uv2 __attribute__((ext_vector_type(2)));
typedef unsigned uv4 __attribute__((ext_vector_type(4)));
typedef double dv2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t brsrc;
__device__ __forceinline__ brsrc mkbuf(const void *p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)(bytes > 0x7fffffffu ? 0x7fffffffu : bytes), 0x00020000);
}
#ifdef UV_ABL_NOLOAD   // ablation (timing only, results meaningless): no vector memory loads, operands made up from the offsets
__device__ __forceinline__ double bld(brsrc, unsigned voff, int soff) { return (double)(int)(voff + (unsigned)soff) * 1e-9 + 1.0; }
__device__ __forceinline__ double2 bld2(brsrc, unsigned voff, int soff) {
  const double v = (double)(int)(voff + (unsigned)soff) * 1e-9;
  return make_double2(v, v + 1e-3);
}
#else
__device__ __forceinline__ double bld(brsrc r, unsigned voff, int soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ double2 bld2(brsrc r, unsigned voff, int soff) {
  const dv2 v = __builtin_bit_cast(dv2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  return make_double2(v.x, v.y);
}
#endif
__device__ __forceinline__ void bst(brsrc r, unsigned voff, int soff, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uv2, v), r, voff, soff, 0);
}
__device__ __forceinline__ void bst2(brsrc r, unsigned voff, int soff, double a, double b) {
  dv2 v; v.x = a; v.y = b;
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uv4, v), r, voff, soff, 0);
}
// v_max_f64 / v_min_f64: one instruction instead of compare + two selects (operands are never NaN here)
__device__ __forceinline__ double fmx(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ double fmn(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ double upstream(double v, double a, double b) { return v * (a + b) + dabs(v) * (a - b); }
// 0.5*((cpos+cneg)*f + (cpos-cneg)*|f|) of adv_flx:703-705 is cpos*f for f >= 0 and cneg*f otherwise: compare,
// select, one multiplication instead of six operations (equal to rounding, not bit for bit)
__device__ __forceinline__ double limited(double cpos, double cneg, double f) { return ((f >= 0.0) ? cpos : cneg) * f; }
// x / y for the limiter ratios.  x is finite, y = P + epsln lies in [1e-20, ~1e6]: the range handling of the IEEE
// division sequence (two v_div_scale, v_div_fmas, v_div_fixup) is dead weight here.  v_rcp_f64 (relative error
// 2^-23), ONE Newton step (2^-46) and one correction of the quotient (error (2^-46)^2, below the final rounding):
// 6 instructions instead of 12, result within 1 ulp of x / y.
__device__ __forceinline__ double div_pos(double x, double y) {
  double r = __builtin_amdgcn_rcp(y);
  r = __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
  const double q = x * r;
  return __builtin_fma(__builtin_fma(-y, q, x), r, q);
}
// R+ and R- of Zalesak's limiter for one cell (tracer_adv_flx.F:672-690)
__device__ __forceinline__ void fct_ratio(double fxa, double fxb, double tlo, double scale, double flxlft, double flxrgt,
                                          double mask, double &rp, double &rm) {
  const double trmax = fmx(fmx(fxa, fxb), tlo), trmin = fmn(fmn(fxa, fxb), tlo);
  const double pplus = scale * (fmx(0.0, flxlft) - fmn(0.0, flxrgt));
  const double pminus = scale * (fmx(0.0, flxrgt) - fmn(0.0, flxlft));
  rp = fmn(1., div_pos(mask * (trmax - tlo), pplus + UV_EPSLN));
  rm = fmn(1., div_pos(mask * (tlo - trmin), pminus + UV_EPSLN));
}

// pass A order: tracer (group) index fastest, so the four waves of a workgroup work on four tracer groups of the same
// lanes and share their coefficient lines in L1.  Each wave takes NTR consecutive tracers of the launch; when nt_local
// is not a multiple of NTR the last group repeats its first tracer as a stand-in (live = false).
template <int NTR>
__device__ __forceinline__ bool col_decode(const uvic_ctx &c, const ColGrid &g, int item, int &code, int (&n1)[NTR], bool (&live)[NTR]) {
  if (item >= g.total) return false;
  const int ngroups = (c.nt_local + NTR - 1) / NTR;
  const int grp = item % ngroups;
  _Pragma("unroll") for (int q = 0; q < NTR; ++q) {
    const int nl = grp * NTR + q;
    live[q] = nl < c.nt_local;
    n1[q] = c.n0 + (live[q] ? nl : grp * NTR) + 1;
  }
  code = g.lanes[(size_t)(item / ngroups) * 64 + threadIdx.x];
  return true;
}
// pass B order: the waves of one tracer next to each other (rows ascending), so that rows r-1, r, r+1 of t, R+-Y
// that a wave reads are the centre rows of its neighbours and come from L1/L2 instead of being fetched three times
__device__ __forceinline__ bool col_decode_rows(const uvic_ctx &c, const ColGrid &g, int item, int &code, int &n1) {
  if (item >= g.total) return false;
  n1 = c.n0 + item / g.nwaves + 1;
  code = g.lanes[(size_t)(item % g.nwaves) * 64 + threadIdx.x];
  return true;
}
// ===========================================================================
// pass A: one sweep down the column, NTR tracers per lane
//
// NTR = 2 puts two tracers through the same lanes: the 13 coefficient pairs, four velocities, kmt, the masks and
// every scalar of a level are fetched and formed once for both, the two tracers are two independent dependency
// chains for the issue logic, and the launch needs half as many waves (at ~250 VGPRs two of them share a SIMD:
// 2048 slots hold all 15 x 133 waves of the nt = 30 case at once, where the one-tracer form needs a second,
// nearly empty round of its 160-VGPR waves).  live[q] = false: tracer q is a stand-in (odd tracer count), its
// results are not stored.
// ===========================================================================
//
// PART splits the pass into two sweeps with half the state each, so that four waves instead of three fit on a SIMD
// (a wave alone issues one fp64 instruction per ~9 cycles, two READY waves are needed to saturate the VALU, and the
// memory wait of every level takes one of three out of the race) and all waves of an nt = 30 launch are resident at once:
//   PART_DIF (first):  the diffusive fluxes with the folded coefficients -> S = DIFF_Tx + DIFF_Ty + DIFF_Tz
//   PART_ADV (second): low-order fluxes, t_lo, limiter ratios, limited x and z fluxes -> R+-Y and S -= ADV_Tx + ADV_Tz
//   PART_ALL: both in one sweep
enum { PART_ALL = 0, PART_ADV = 1, PART_DIF = 2 };
// AHEAD: what a level reads is fetched one level ahead into the other of two register sets (+72 VGPRs).  For the bulk launch
// this ties with none (three waves on a SIMD cover each other's memory waits); for the T,S launch -- a few hundred waves
// others wait for, among thousands of waves that keep the memory system busy -- every level's wait is otherwise exposed.
// SHARE: the four waves of a workgroup work on four tracers of the SAME lanes, and the 13 coefficient pairs of a level (two
// thirds of what a wave fetches through the vector memory pipe, which is what bounds the pass: 1.5 GB per step at 64 B per
// clock and CU) are the same for all four.  Each wave brings a quarter of the next level's pairs into LDS (buffer_load ...
// lds: no registers), one workgroup barrier per level, and every wave reads all 13 from LDS (256 B per clock).
// `lds` = 2 x 13 x 64 double2 per workgroup, `wv` = the wave's number in it.
#define COL_SHARE_SLOTS(YFIN) (CF_PAIRS + 3 + ((YFIN) ? 2 : 0))   /* row r: all pairs; row r-1: the north-face ones; YFIN: the velocity pairs of row r+1 */
// YFIN: the wave also forms t_lo and the y-limiter ratios of the row to its NORTH (from t of rows r+1, r+2 and that row's
// velocities: five more loads and ~60 more instructions per level) and with them the FINAL, limited advective flux through
// its north face, which it stores (8 bytes per cell update) instead of the ratio pair R+-Y (16).  Pass B then reads that flux
// of rows r and r-1 -- not R+-Y of three rows, t(tau-1) and t(tau) of three rows and the velocities to form the fluxes again:
// 48 bytes per cell update instead of 136, for the pass that is bound by memory traffic.
template <int NTR, int PART, bool AHEAD = false, bool SHARE = false, bool YFIN = false>
__device__ __forceinline__ void colfct_wave(const uvic_ctx &c, const double *__restrict__ cf, double *__restrict__ S,
                                            int code, const int (&n1)[NTR], const bool (&live)[NTR], double *lds = nullptr, int wv = 0) {
  constexpr bool ADV = PART != PART_DIF, DIF = PART != PART_ADV;
  constexpr int SLOTS = COL_SHARE_SLOTS(YFIN);
  static_assert(!YFIN || PART == PART_ALL, "the final y flux is formed by the one-sweep pass");
  UV_DIMS(c);
  const int i = COL_LANE_I(code), r = COL_LANE_R(code);
  const bool owned = COL_LANE_OWNED(code) != 0;
  // per-level metrics, one entry per lane, broadcast by v_readlane (no memory latency in the march)
  LaneTable t_dzt2r, t_dtxcel, t_dztr;
  t_dzt2r.load(c.dzt2r, km); t_dtxcel.load(c.dtxcel, km); t_dztr.load(c.dztr, km);
  const int kz = c.kmt[X2(i, r)], kz_s = c.kmt[X2(i, r - 1)], kz_n = c.kmt[X2(i, r + 1)];
  // across the seam between two runs of the lane map the neighbour is not the x-neighbour: only halo lanes look there
  const int kz_w = dpp_i<DPP_WAVE_SHR1>(kz), kz_e = dpp_i<DPP_WAVE_SHL1>(kz);
  const double cstr_r = c.cstr[r - 1];
  const double cstdxt2r = cstr_r * c.dxtr[i - 1] * 0.5, cstdxtr = cstr_r * c.dxtr[i - 1];
  const double cstdyt2r = c.cstdyt2r[r - 1], cstdytr = c.cstdytr[r - 1];
  const bool south_wall = (r - 1 == 1);   // no antidiffusive flux through the face to row 1 (adv_flx: jstrt)
  const double c2dtts = c.c2dtts;
  // YFIN: the row to the north (N = r+1 <= jmt) and the one beyond it (the reference clamps: jp2 = min(j+2, jmt), adv_flx:555)
  const int rnn = imin(r + 2, jmt);
  const int kz_nn = YFIN ? c.kmt[X2(i, rnn)] : 0;
  const double cstdxt2r_N = YFIN ? c.cstr[r] * c.dxtr[i - 1] * 0.5 : 0.0, cstdyt2r_N = YFIN ? c.cstdyt2r[r] : 0.0;
  // addresses = buffer descriptor + wave-uniform byte offset (one scalar register: level and row shift) + the lane's
  // 32-bit offset of its (i, r) column
  const int rowstride = imt * km;
  // (the lane offset points at row r-1, so that the scalar offset is never negative: the hardware adds it unsigned)
  const unsigned lb = (unsigned)((r - 2) * rowstride + (i - 1)) * 8u;   // level 1 of column i of row r-1, cell fields
  const unsigned lb2 = lb * 2u;
  const unsigned lbf = (unsigned)((r - 1) * imt * (km + 1) + (i - 1)) * 8u;   // face 0 of the lane's column, face fields
  const unsigned lb_nn = (unsigned)((rnn - 1) * rowstride + (i - 1)) * 8u;     // level 1 of column i of row min(r+2, jmt)
  const unsigned lbf_N = lbf + (unsigned)(imt * (km + 1)) * 8u;               // face 0 of the column to the north
  const brsrc b_te = mkbuf(c.tot_e, N3 * 8), b_tn = mkbuf(c.tot_n, N3 * 8);
  const brsrc b_tb = mkbuf(c.tot_b, NF * 8), b_vb = mkbuf(c.adv_vbt, NF * 8);
  const brsrc b_cf = mkbuf(cf, N3 * 16 * CF_PAIRS);
  brsrc b_tm[NTR], b_tt[NTR], b_S[NTR], b_R[NTR];
  double stf[NTR], btf[NTR];
  _Pragma("unroll") for (int q = 0; q < NTR; ++q) {
    const size_t nloc = (size_t)(n1[q] - 1 - c.n0);
    b_tm[q] = mkbuf(c.t_taum1 + (size_t)(n1[q] - 1) * N3, N3 * 8);
    b_tt[q] = mkbuf(c.t_tau + (size_t)(n1[q] - 1) * N3, N3 * 8);
    b_S[q] = mkbuf(S + nloc * N3, N3 * 8);
    b_R[q] = mkbuf(c.Rpm + nloc * N3 * 2, N3 * 16);   // (R+, R-) of the y limiter, one 16-byte element per cell
    stf[q] = c.stf[X2(i, r) + (size_t)(n1[q] - 1) * imt * jmt] * (1.0 - c.aidif);
    btf[q] = c.btf[X2(i, r) + (size_t)(n1[q] - 1) * imt * jmt] * (1.0 - c.aidif);
  }
#define OC(k, dj) (((((k)-1) * imt + ((dj) + 1) * rowstride)) * 8)   /* byte offset of level k of row r+dj from the lane's offset */
#define LDQ(b, k, dj) bld(b[q], lb, OC(k, dj))
#define LD(b, k, dj) bld(b_##b, lb, OC(k, dj))
#define CFP(pair, k, dj) bld2(b_cf, lb2, ((int)(pair) * (int)N3 + ((k)-1) * imt + ((dj) + 1) * rowstride) * 16)
#define OF(kf) (((kf) * imt) * 8)
#define FORQ _Pragma("unroll") for (int q = 0; q < NTR; ++q)
  // surface faces
  const double vb0 = bld(b_vb, lbf, OF(0));
  // state carried from level to level, per tracer
  double mc1[NTR], ms1[NTR], mn1[NTR];   // level s of t(tau-1) centre/south/north
  double tc0[NTR], tc1[NTR];             // levels s-1 and s of t(tau)
  double fblo_up[NTR], afb_up[NTR];      // low-order and raw antidiffusive flux through the face above level s (adv_flx:617)
  double dfb_up[NTR], dfbi_up[NTR];      // diffusive fluxes through the face above level s
  double rzp_prev[NTR], rzm_prev[NTR], spart_prev[NTR];
  double sfin_prev[NTR], sfin_last[NTR], sdif_last[NTR];
  // differences that level s+1 needs again are handed down instead of being formed (and shuffled) twice:
  // T(s)-T(s+1) of the own, east, south and north columns; T(i+1)-T(i) and T(i)-T(i-1) at level s+1
  double me_next[NTR], dz_c[NTR], dz_e[NTR], dz_s[NTR], dz_n[NTR], dx_next[NTR], dxw_next[NTR];
  double fbfin_up[NTR];                  // FINAL advective flux through the face above the level being finalised (top: tracer.F:1063)
  double fbloN_up[NTR];                  // YFIN: low-order flux through the face above level s of the column to the north
  double mk_prev = 0.0;
  FORQ {
    mc1[q] = LDQ(b_tm, 1, 0); ms1[q] = LDQ(b_tm, 1, -1); mn1[q] = LDQ(b_tm, 1, 1);
    tc0[q] = tc1[q] = LDQ(b_tt, 1, 0);
    fblo_up[q] = vb0 * 2.0 * mc1[q];
    afb_up[q] = fblo_up[q];
    dfb_up[q] = stf[q]; dfbi_up[q] = 0.0;
    rzp_prev[q] = rzm_prev[q] = spart_prev[q] = sfin_prev[q] = sfin_last[q] = sdif_last[q] = 0.0;
    me_next[q] = shfl_e(mc1[q]);
    dz_c[q] = dz_e[q] = dz_s[q] = dz_n[q] = 0.0;
    dx_next[q] = me_next[q] - mc1[q]; dxw_next[q] = shfl_w(dx_next[q]);
    fbfin_up[q] = vb0 * (tc1[q] + tc1[q]);
    fbloN_up[q] = YFIN ? bld(b_vb, lbf_N, OF(0)) * 2.0 * mn1[q] : 0.0;
  }
  // everything level s reads from memory
  struct LvlIn {
    double ve, vn, vs, vb;
    double cfc[2 * CF_PAIRS], cfs[6];   // folded coefficients of row r and the north-face ones (slots 0..4) of row r-1
    double mc2[NTR], ms2[NTR], mn2[NTR], tc2[NTR], t_s[NTR], t_n[NTR];
    double m_nn[NTR], t_nn[NTR];   // YFIN: t(tau-1), t(tau) of level s two rows to the north
  };
  auto load_in = [&](LvlIn &L, int s) {
    const int sp = (s == km) ? km : s + 1;
    L.ve = L.vn = L.vs = L.vb = 0.0;
    if (ADV && !SHARE) {   // (SHARE: the velocity pairs come through LDS with the coefficients)
      L.ve = LD(te, s, 0); L.vn = LD(tn, s, 0); L.vs = LD(tn, s, -1);
      L.vb = (s < km) ? bld(b_tb, lbf, OF(s)) : bld(b_vb, lbf, OF(km));
    }
    if (DIF && SHARE) {
      // (read from LDS where they are used, in the diffusive half of level(): they need not occupy registers before)
    } else if (DIF) {
      _Pragma("unroll") for (int p = 0; p < CF_DPAIRS; ++p) {
        const double2 v = CFP(p, s, 0);
        L.cfc[2 * p] = v.x; L.cfc[2 * p + 1] = v.y;
      }
      _Pragma("unroll") for (int p = 0; p < 3; ++p) {
        const double2 v = CFP(p, s, -1);
        L.cfs[2 * p] = v.x; L.cfs[2 * p + 1] = v.y;
      }
    }
    FORQ {
      L.mc2[q] = LDQ(b_tm, sp, 0); L.ms2[q] = LDQ(b_tm, sp, -1); L.mn2[q] = LDQ(b_tm, sp, 1);
      L.tc2[q] = L.t_s[q] = L.t_n[q] = 0.0;
      if (ADV) { L.tc2[q] = LDQ(b_tt, sp, 0); L.t_s[q] = LDQ(b_tt, s, -1); L.t_n[q] = LDQ(b_tt, s, 1); }
      L.m_nn[q] = L.t_nn[q] = 0.0;
      if (YFIN) { L.m_nn[q] = bld(b_tm[q], lb_nn, OC(s, -1)); L.t_nn[q] = bld(b_tt[q], lb_nn, OC(s, -1)); }   // (lb_nn points at the row itself)
    }
  };
  auto level = [&](const LvlIn &L, int s) {
    const bool last = (s == km);
    // ---- what the tracers share: velocities, folded coefficients, masks, metrics ------------------------
    double ve = L.ve, vn = L.vn, vs = L.vs, vb = L.vb, veN = 0.0, vnN = 0.0, vbN = 0.0;
    const double2 *sl = SHARE ? (const double2 *)lds + (size_t)(s & 1) * SLOTS * 64 + (threadIdx.x & 63) : nullptr;
    if (SHARE && ADV) {
      const double2 a = sl[(CF_VE / 2) * 64], b = sl[(CF_VB / 2) * 64];
      ve = a.x; vn = a.y; vb = b.x; vs = b.y;
      if (YFIN) {
        const double2 aN = sl[(CF_PAIRS + 3) * 64], bN = sl[(CF_PAIRS + 4) * 64];
        veN = aN.x; vnN = aN.y; vbN = bN.x;
      }
    }
    double cfc_l[2 * CF_PAIRS], cfs_l[6];
    const double *cfc = SHARE ? cfc_l : L.cfc, *cfs = SHARE ? cfs_l : L.cfs;
    const double dzt2r_s = t_dzt2r.at(s - 1), ddztr = t_dztr.at(s - 1);
    const double dzt2r_up = (s >= 2) ? t_dzt2r.at(s - 2) : 0.0;
    const double twodt = c2dtts * t_dtxcel.at(s - 1);
    const double mk = (s <= kz) ? 1.0 : 0.0;
    // wet neighbour -> its face value, land -> t_lo (adv_flx:640-668 blends with the 0/1 mask: the same value)
    const bool wet_w = s <= kz_w, wet_e = s <= kz_e, wet_s = s <= kz_s, wet_n = s <= kz_n;
    const bool wet_up = s - 1 >= 1 && s - 1 <= kz, wet_dn = s + 1 <= kz;
    const double avb = dabs(vb);
    // YFIN without SHARE (the T,S launch): the velocities of the row to the north, asked for at the head of the level
    if (YFIN && !SHARE) {
      veN = bld(b_te, lb, OC(s, 1)); vnN = bld(b_tn, lb, OC(s, 1));
      vbN = (s < km) ? bld(b_tb, lbf_N, OF(s)) : 0.0;
    }
    FORQ {
      const double mc2 = L.mc2[q], ms2 = L.ms2[q], mn2 = L.mn2[q];
      const double m_c = mc1[q];
      const double mc2_e = shfl_e(mc2);
      double spart = 0.0;     // what this sweep adds to S of level s, but for the z advection (finalised one level later)
      // =================== advective part (adv_flx) =====================================================
      if (ADV) {
        const double tc2 = L.tc2[q], t_s = L.t_s[q], t_n = L.t_n[q];
        // the diffusive sweep has stored its share of S already: fetched here, used when level s-1 is finalised
        double sdif_prev = 0.0;
        if (PART == PART_ADV) sdif_prev = bld(b_S[q], lb, OC(s >= 2 ? s - 1 : 1, 0));
        if (PART == PART_ADV && last) sdif_last[q] = bld(b_S[q], lb, OC(km, 0));
        const double tt_c = tc1[q];
        const double m_e = me_next[q], tt_e = shfl_e(tt_c), tt_w = shfl_w(tt_c);
        // ---- low order and raw antidiffusive fluxes (adv_flx:500-619) ----------
        const double felo = upstream(ve, m_c, m_e);
        const double afe = ve * (tt_c + tt_e) - felo;
        const double felo_w = shfl_w(felo), afe_w = shfl_w(afe);
        const double fnlo_n = upstream(vn, m_c, mn1[q]), fnlo_s = upstream(vs, ms1[q], m_c);
        double fblo = 0.0, afb = 0.0;
        if (!last) {
          fblo = vb * (mc2 + m_c) + avb * (mc2 - m_c);
          afb = vb * (tt_c + tc2) - fblo * mk;
        }
        const double advx = (felo - felo_w) * cstdxt2r, advy = (fnlo_n - fnlo_s) * cstdyt2r;
        const double advz = (fblo_up[q] - fblo) * dzt2r_s;
        const double tlo = m_c - twodt * (advx + advy + advz) * mk;
        // ---- limiter ratios ---------------------------------------------------------
        double rxp, rxm, ryp, rym, rzp, rzm;
        {
          const double mw = 0.5 * (tt_w + tt_c), me = 0.5 * (tt_c + tt_e);
          fct_ratio(wet_w ? mw : tlo, wet_e ? me : tlo, tlo, c2dtts * cstdxt2r, afe_w, afe, mk, rxp, rxm);
        }
        {
          const double afn_n = vn * (tt_c + t_n) - fnlo_n;
          const double afn_s = south_wall ? 0.0 : vs * (t_s + tt_c) - fnlo_s;
          fct_ratio(wet_s ? 0.5 * (t_s + tt_c) : tlo, wet_n ? 0.5 * (tt_c + t_n) : tlo, tlo, c2dtts * cstdyt2r, afn_s, afn_n, mk,
                    ryp, rym);
        }
        {
          const double fxa = wet_up ? 0.5 * (tc0[q] + tt_c) : tlo;
          const double fxb = (!last && wet_dn) ? 0.5 * (tt_c + tc2) : tlo;
          fct_ratio(fxa, fxb, tlo, c2dtts * dzt2r_s, afb, afb_up[q], mk, rzp, rzm);
        }
        if (YFIN) {
          // ---- the row to the north: low-order fluxes, t_lo, y-limiter ratios (the same formulas one row up) ----------
          const double m_nn = L.m_nn[q], t_nn = L.t_nn[q];
          const double m_N = mn1[q];
          const double mkN = (s <= kz_n) ? 1.0 : 0.0;
          const double feloN = upstream(veN, m_N, shfl_e(m_N));
          const double fnloN_n = upstream(vnN, m_N, m_nn);
          double fbloN = 0.0;
          if (!last) fbloN = vbN * (mn2 + m_N) + dabs(vbN) * (mn2 - m_N);
          const double advN = (feloN - shfl_w(feloN)) * cstdxt2r_N + (fnloN_n - fnlo_n) * cstdyt2r_N + (fbloN_up[q] - fbloN) * dzt2r_s;
          const double tloN = m_N - twodt * advN * mkN;
          const double afn_nn = vnN * (t_n + t_nn) - fnloN_n;
          double rypN, rymN;
          fct_ratio(mk != 0.0 ? 0.5 * (tt_c + t_n) : tloN, (s <= kz_nn) ? 0.5 * (t_n + t_nn) : tloN, tloN, c2dtts * cstdyt2r_N,
                    vn * (tt_c + t_n) - fnlo_n, afn_nn, mkN, rypN, rymN);
          // ---- the limited flux through the north face, final (adv_flx:770-783, 994-996) --------------------------------
          const double afn_n = vn * (tt_c + t_n) - fnlo_n;
          const double fn_fin = (limited(fmn(rypN, rym), fmn(ryp, rymN), afn_n) + fnlo_n) * mk;
          if (owned && live[q]) bst(b_R[q], lb, OC(s, 0), fn_fin);
          fbloN_up[q] = fbloN;
        } else if (owned && live[q]) bst2(b_R[q], lb2, OC(s, 0) * 2, ryp, rym);
        // ---- limited x flux and its divergence (adv_flx:695-711, 989-992) ---------------
        const double rxp_e = shfl_e(rxp), rxm_e = shfl_e(rxm);
        const double fefin = limited(fmn(rxp_e, rxm), fmn(rxp, rxm_e), afe) + felo;
        const double ADV_Tx = (fefin - shfl_w(fefin)) * cstdxt2r;
        spart = -ADV_Tx;
        // ---- finalise level s-1: limited z flux through the face between s-1 and s (adv_flx:857-887, 994-999)
        if (s >= 2) {
          const double fbfin = (limited(fmn(rzp_prev[q], rzm), fmn(rzp, rzm_prev[q]), afb_up[q]) + fblo_up[q]) * mk_prev;
          const double ADV_Tz = (fbfin_up[q] - fbfin) * dzt2r_up;
          sfin_prev[q] = sub_nc(add_nc(spart_prev[q], sdif_prev), ADV_Tz);
          fbfin_up[q] = fbfin;
        }
        if (last) {  // bottom face of the column (tracer.F:1065); vb holds adv_vbt there
          const double fbfin = vb * tt_c;
          sfin_last[q] = (fbfin_up[q] - fbfin) * dzt2r_s;   // ADV_Tz of the bottom level
        }
        rzp_prev[q] = rzp; rzm_prev[q] = rzm;
        fblo_up[q] = fblo; afb_up[q] = afb;
        me_next[q] = mc2_e;
        tc0[q] = tc1[q]; tc1[q] = tc2;
      }
      // =================== diffusive part (coefficients folded by ai_coef_cell) ===========================
      if (DIF) {
        if (SHARE && q == 0) {
          _Pragma("unroll") for (int p = 0; p < CF_DPAIRS; ++p) {
            const double2 v = sl[p * 64];
            cfc_l[2 * p] = v.x; cfc_l[2 * p + 1] = v.y;
          }
          _Pragma("unroll") for (int p = 0; p < 3; ++p) {
            const double2 v = sl[(CF_PAIRS + p) * 64];
            cfs_l[2 * p] = v.x; cfs_l[2 * p + 1] = v.y;
          }
        }
        const double dz_up = dz_c[q], dz_dn = (!last) ? m_c - mc2 : 0.0;          // own column (dz_up = dz_dn of the level above, 0 at the top)
        const double dze_up = dz_e[q], dze_dn = shfl_e(dz_dn);                      // east column
        const double dzs_up = dz_s[q], dzs_dn = (!last) ? ms1[q] - ms2 : 0.0;      // south row
        const double dzn_up = dz_n[q], dzn_dn = (!last) ? mn1[q] - mn2 : 0.0;      // north row
        const double dx_c = dx_next[q], dx_d = mc2_e - mc2;                         // T(i+1)-T(i) at levels s, s+1
        const double dxw_c = dxw_next[q], dxw_d = shfl_w(dx_d);                     // T(i)-T(i-1)
        const double dfe = cfc[CF_AE] * dx_c + cfc[CF_CE + 0] * dz_up + cfc[CF_CE + 1] * dze_up + cfc[CF_CE + 2] * dz_dn +
                           cfc[CF_CE + 3] * dze_dn;
        const double DIFF_Tx = (dfe - shfl_w(dfe)) * cstdxtr;
        const double dfn_n = cfc[CF_AN] * (mn1[q] - m_c) + cfc[CF_CN + 0] * dz_up + cfc[CF_CN + 1] * dzn_up + cfc[CF_CN + 2] * dz_dn +
                             cfc[CF_CN + 3] * dzn_dn;
        const double dfn_s = cfs[0] * (m_c - ms1[q]) + cfs[1] * dzs_up + cfs[2] * dz_up + cfs[3] * dzs_dn + cfs[4] * dz_dn;
        const double DIFF_Ty = (dfn_n - dfn_s) * cstdytr;
        double dfb = 0.0, dfbi = 0.0;  // through the face below level s
        if (!last) {
          dfb = cfc[CF_BV] * (m_c - mc2);
          dfbi = cfc[CF_CBX + 0] * dxw_c + cfc[CF_CBX + 1] * dx_c + cfc[CF_CBX + 2] * dxw_d + cfc[CF_CBX + 3] * dx_d +
                 cfc[CF_CBY + 0] * (m_c - ms1[q]) + cfc[CF_CBY + 1] * (mn1[q] - m_c) + cfc[CF_CBY + 2] * (mc2 - ms2) +
                 cfc[CF_CBY + 3] * (mn2 - mc2);
        }
        if (s == kz) dfb = btf[q];  // bottom boundary condition of the explicit vertical flux (tracer.F:1060-1062)
        if (kz == 0 && s == 1) dfb_up[q] = btf[q];
        const double DIFF_Tz = (dfb_up[q] - dfb) * ddztr + (dfbi_up[q] - dfbi) * ddztr;
        const double dsum = DIFF_Tx + DIFF_Ty + DIFF_Tz;
        spart = (PART == PART_DIF) ? dsum : add_nc(spart, dsum);
        dfb_up[q] = dfb; dfbi_up[q] = dfbi;
        dx_next[q] = dx_d; dxw_next[q] = dxw_d;
        dz_c[q] = dz_dn; dz_e[q] = dze_dn; dz_s[q] = dzs_dn; dz_n[q] = dzn_dn;
      }
      // ---- stores of S -----------------------------------------------------------------------
      if (PART == PART_DIF) {
        if (owned && live[q]) bst(b_S[q], lb, OC(s, 0), spart);
      } else {
        if (s >= 2 && owned && live[q]) bst(b_S[q], lb, OC(s - 1, 0), sfin_prev[q]);
        if (last && owned && live[q]) bst(b_S[q], lb, OC(km, 0), sub_nc(add_nc(spart, sdif_last[q]), sfin_last[q]));
        spart_prev[q] = spart;
      }
      mc1[q] = mc2; ms1[q] = ms2; mn1[q] = mn2;
    }
    mk_prev = mk;
  };
  if (AHEAD && SHARE) {
    // both: the pairs through LDS one level ahead, and what the wave loads for itself one level ahead into a second register set
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
    LvlIn A, B;
    bring(1);
    load_in(A, 1);
    int s = 1;
    for (; s + 1 <= km; s += 2) {
      __builtin_amdgcn_s_waitcnt(0x0f70);
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
  } else if (AHEAD) {
    // no branch around a load (the compiler's wait counts stay exact on straight-line code only): the level index is
    // clamped instead (the last pair re-reads level km) and an odd last level is peeled
    LvlIn A, B;
    load_in(A, 1);
    int s = 1;
    for (; s + 1 <= km; s += 2) {
      load_in(B, s + 1);
      level(A, s);
      load_in(A, imin(s + 2, km));
      level(B, s + 1);
    }
    if (s == km) level(A, km);
  } else if (SHARE) {
    // this wave's share of the coefficient pairs of level s: slots wv, wv+4, ... of the 13 (row r: 0..9, row r-1: 10..12)
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
    bring(1);
    for (int s = 1; s <= km; ++s) {
      // the level's pairs have landed (this wave's share: vmcnt; the others': the barrier), and every wave has finished
      // reading the other buffer (it did so in level s-1, before it came here)
      __builtin_amdgcn_s_waitcnt(0x0f70);   /* vmcnt(0), lgkmcnt and expcnt left alone */
      __builtin_amdgcn_s_barrier();
      if (s < km) bring(s + 1);
      LvlIn L;
      load_in(L, s);
      level(L, s);
    }
  } else {
    for (int s = 1; s <= km; ++s) {
      LvlIn L;
      load_in(L, s);
      level(L, s);
    }
  }
#undef FORQ
#undef LDQ
#undef LD
#undef CFP
#undef OC
#undef OF
}

// ===========================================================================
// pass B: y advection, explicit update, implicit vertical diffusion (invtri.F)
// `ework` is the wave's LDS scratch: e(k) and z(k) of the Thomas recurrence, each (km+1) x 64 doubles laid out
// [k][lane], so that the forward sweep writes t(tau+1) nowhere and the back substitution stores it once
// ===========================================================================
//
// ZG: the forward sweep parks z(k) in t(tau+1) itself (global memory, read back by the same lane) instead of LDS, so that a
// wave needs 10 KB of LDS for e(k) instead of 20: sixteen waves fit on a CU instead of eight, and seven instead of three
// beside a MOBI team (90 KB).  Not with the fused convective replay, which works on the column in LDS.
// YFIN: pass A has left the final advective flux through the north face of every row (see colfct_wave): the pass reads that
// of rows r and r-1 and none of what it would otherwise need to form them.
template <bool ZG, bool YFIN = false>
// keep_lds: the back substitution also leaves t(tau+1) of the column in LDS (zwork[k][lane], k = 1..km), for a caller
// that goes on with it (the T,S launch: the convective walk of the same workgroup)
__device__ __forceinline__ void colupd_wave(const uvic_ctx &c, const double *__restrict__ S, double *ework, int code, int n1,
                                            int fuse_convect, bool keep_lds = false) {
  UV_DIMS(c);
  const int lane = threadIdx.x;
  const int i = COL_LANE_I(code), r = COL_LANE_R(code);
  LaneTable t_dtxcel, t_dztur, t_dztlr, t_dztr;   // filled while every lane is still active
  t_dtxcel.load(c.dtxcel, c.km); t_dztur.load(c.dztur, c.km); t_dztlr.load(c.dztlr, c.km); t_dztr.load(c.dztr, c.km);
  // the lanes are ocean columns; land keeps the zeros it was given once (uvic_gpu.hip: land_clean)
  if (!COL_LANE_OWNED(code)) return;   // padding of the last wave
  const size_t nloc = (size_t)(n1 - 1 - c.n0);
  const double *tm = c.t_taum1 + (size_t)(n1 - 1) * N3;
  const double *tt = c.t_tau + (size_t)(n1 - 1) * N3;
  double *tp = c.t_taup1 + (size_t)(n1 - 1) * N3;
  const double *Rpm = c.Rpm + nloc * N3 * 2;
  const double *Sn = S + nloc * N3;
  // the source term is read here, not in pass A: with MOBI computed one step ahead on the side
  // stream only this pass has to wait for it
  const double *source = 0;
  if (c.src && c.itrc[n1 - 1] != 0) source = c.src + (size_t)(c.itrc[n1 - 1] - 1) * N3;
  const int kz = c.kmt[X2(i, r)], kz_s = c.kmt[X2(i, r - 1)];
  const double cstdyt2r = c.cstdyt2r[r - 1];
  const bool south_wall = (r - 1 == 1);
  const int rowstride = imt * km;
  // the lane's column in row r-1; the scalar offset shifts level and row and is never negative (the hardware adds it unsigned)
  const unsigned lb = (unsigned)((r - 2) * rowstride + (i - 1)) * 8u, lb2 = lb * 2u;
#define OC(k, dj) (((((k)-1) * imt + ((dj) + 1) * rowstride)) * 8)
#define AT(b, k, dj) bld(b, lb, OC(k, dj))
#define RPM(k, dj) bld2(b_R, lb2, OC(k, dj) * 2)
  const double topbc = c.stf[X2(i, r) + (size_t)(n1 - 1) * imt * jmt], botbc = c.btf[X2(i, r) + (size_t)(n1 - 1) * imt * jmt];
  const double aidif = c.aidif, eps = 1.e-30;
  const int kb = imax(2, kz);
  double bet = 0.0, zprev = 0.0, cprev = 0.0;
  const int ic = (i == 2) ? imt : ((i == imt - 1) ? 1 : 0);
  double *zwork = ework + (size_t)(km + 1) * 64;
  // source term: always loaded (from S when the tracer has none) and selected afterwards, so that the
  // number of loads in flight is the same on every path and the waits stay exact
  const bool has_src = source != 0;
  const double *srcp = has_src ? source : Sn;
  const brsrc b_tm = mkbuf(tm, N3 * 8), b_tt = mkbuf(tt, N3 * 8), b_tn = mkbuf(c.tot_n, N3 * 8), b_R = mkbuf(Rpm, N3 * 16);
  const brsrc b_S = mkbuf(Sn, N3 * 8), b_src = mkbuf(srcp, N3 * 8), b_dcb = mkbuf(c.diff_cbt, N3 * 8), b_tp = mkbuf(tp, N3 * 8);
  // The sweep is one dependent chain per column (Thomas recurrence): what a level reads is fetched one
  // level ahead into the other of two register sets so that the chain never waits for memory.
  struct Lvl {
    double m_c, m_s, m_n, t_c, t_s, t_n, vn, vs, rp0, rm0, rps, rms, rpn, rmn, sn, src, dcb;
  };
  auto load_level = [&](Lvl &L, int k) {
    if (YFIN) {
      L.m_c = AT(b_tm, k, 0);
      L.rp0 = bld(b_R, lb, OC(k, 0)); L.rps = bld(b_R, lb, OC(k, -1));   // the final fluxes through the north faces of rows r, r-1
      L.sn = AT(b_S, k, 0);
      L.src = AT(b_src, k, 0);
      L.dcb = AT(b_dcb, k, 0);
      return;
    }
    L.m_c = AT(b_tm, k, 0); L.m_s = AT(b_tm, k, -1); L.m_n = AT(b_tm, k, 1);
    L.t_c = AT(b_tt, k, 0); L.t_s = AT(b_tt, k, -1); L.t_n = AT(b_tt, k, 1);
    L.vn = AT(b_tn, k, 0); L.vs = AT(b_tn, k, -1);
    {
      const double2 r0 = RPM(k, 0), rs = RPM(k, -1), rn = RPM(k, 1);
      L.rp0 = r0.x; L.rm0 = r0.y; L.rps = rs.x; L.rms = rs.y; L.rpn = rn.x; L.rmn = rn.y;
    }
    L.sn = AT(b_S, k, 0);
    L.src = AT(b_src, k, 0);
    L.dcb = AT(b_dcb, k, 0);
  };
  double dcb_up = 0.0;   // diff_cbt of the level above (the reference reads level max(1,k-1); at k=1 its factor is zeroed)
  auto level = [&](const Lvl &L, int k) {
    const double m_c = L.m_c;
    const double mk = (k <= kz) ? 1.0 : 0.0;
    double ADV_Ty;
    if (YFIN) {
      ADV_Ty = (L.rp0 - L.rps) * cstdyt2r;
    } else {
      const double t_c = L.t_c, mk_s = (k <= kz_s) ? 1.0 : 0.0;
      const double lo_n = upstream(L.vn, m_c, L.m_n), lo_s = upstream(L.vs, L.m_s, m_c);
      const double f_n = L.vn * (t_c + L.t_n) - lo_n;
      const double f_s = south_wall ? 0.0 : L.vs * (L.t_s + t_c) - lo_s;
      const double fn_n = (limited(fmn(L.rpn, L.rm0), fmn(L.rp0, L.rmn), f_n) + lo_n) * mk;
      const double fn_s = (limited(fmn(L.rp0, L.rms), fmn(L.rps, L.rm0), f_s) + lo_s) * mk_s;
      ADV_Ty = (fn_n - fn_s) * cstdyt2r;
    }
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
    if (ZG) bst(b_tp, lb, OC(k, 0), znew);
    else zwork[(size_t)k * 64 + lane] = znew;
    zprev = znew;
    cprev = cc;
    dcb_up = L.dcb;
  };
#ifdef UV_COL_TIMING
  const long long tq0 = clock64();
#endif
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
#ifdef UV_COL_TIMING
  const long long tq1 = clock64();
#endif
  // Convection fused in (the T,S walk of this step has run already, on the side stream): when a column of the
  // wave has a mixed segment, the back substitution leaves t(tau+1) in LDS, the segments are replayed in the
  // order they were found (convect.F:257-271: tsm = sum t*dztxcl over kt..kb, in that order, / zsm) and the column
  // is stored once.  Same arithmetic as convect_apply_cell, which this replaces.
  const int ncv = (fuse_convect && !ZG) ? c.cv_nseg[X2(i, r)] : 0;
  if (!ZG && __builtin_amdgcn_ballot_w64(ncv > 0) != 0) {
    double zn = zprev;
    for (int k = km - 1; k >= 1; --k) {
      const double zk = zwork[(size_t)k * 64 + lane] - ework[(size_t)(k + 1) * 64 + lane] * zn;
      zwork[(size_t)k * 64 + lane] = zk;
      zn = zk;
    }
    for (int sg = 1; sg <= ncv; ++sg) {
#pragma clang fp contract(off)   /* the same bits as convect_apply_cell, whichever of the two runs (tests/test_gpu_fast.py) */
      const int kt = c.cv_kt[X3(i, sg, r)], kb = c.cv_kb[X3(i, sg, r)];
      const double zsm = c.cv_z[X3(i, sg, r)];
      double tsm3 = 0.0;
      for (int k = kt; k <= kb; ++k) tsm3 = tsm3 + zwork[(size_t)k * 64 + lane] * c.dztxcl[k - 1];
      const double tmx3 = tsm3 / zsm;
      for (int k = kt; k <= kb; ++k) zwork[(size_t)k * 64 + lane] = tmx3;
    }
    for (int k = 1; k <= km; ++k) {
      const double zk = zwork[(size_t)k * 64 + lane];
      bst(b_tp, lb, OC(k, 0), zk);
      if (ic) tp[X3(ic, k, r)] = zk;
    }
    return;
  }
  // back substitution, invtri.F:104-110, and the cyclic images (tracer.F:1153-1155)
  double znext = zprev;
  if (!ZG) bst(b_tp, lb, OC(km, 0), znext);
  if (ic) tp[X3(ic, km, r)] = znext;
  if (ZG) {
    // z(k) comes back from t(tau+1), six levels at a time ahead of the recurrence that consumes them
    for (int k0 = km - 1; k0 >= 1; k0 -= 6) {
      double zq[6];
      _Pragma("unroll") for (int u = 0; u < 6; ++u) zq[u] = bld(b_tp, lb, OC(imax(k0 - u, 1), 0));
      _Pragma("unroll") for (int u = 0; u < 6; ++u) {
        const int k = k0 - u;
        if (k >= 1) {
          const double zk = zq[u] - ework[(size_t)(k + 1) * 64 + lane] * znext;
          bst(b_tp, lb, OC(k, 0), zk);
          if (ic) tp[X3(ic, k, r)] = zk;
          znext = zk;
        }
      }
    }
  } else
  for (int k = km - 1; k >= 1; --k) {
    const double zk = zwork[(size_t)k * 64 + lane] - ework[(size_t)(k + 1) * 64 + lane] * znext;
    bst(b_tp, lb, OC(k, 0), zk);
    if (ic) tp[X3(ic, k, r)] = zk;
    if (keep_lds) zwork[(size_t)k * 64 + lane] = zk;
    znext = zk;
  }
#ifdef UV_COL_TIMING
  if (lane == 0 && (blockIdx.x % 301) == 0 && threadIdx.y == 0)
    printf("colupd blk %d r %d n %d: start %lld forward %lld backsub %lld\n", blockIdx.x, r, n1, tq0, tq1 - tq0, clock64() - tq1);
#endif
}
#undef AT
#undef OC
#undef RPM
#pragma clang fp contract(off)
#endif  // __HIPCC__

}  // namespace uvic
#endif
