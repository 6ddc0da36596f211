// uvic_gpu.hip -- gfx950 kernels' launch wrappers and the C ABI of include/uvic_gpu.h.
//
// GPU code only: there is no CPU path in this library.  Every entry point
// returns a non-zero status (and sets uvic_gpu_last_error) when a HIP call
// fails, so that the Fortran shim can `stop '=>tracer (gpu)'` like the
// reference does on its own errors (updates/09/source/mom/tracer.F:1250).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/uvic_gpu.h"
#include "kernels_col.hpp"
#include "kernels_colx.hpp"
#include "kernels_prep.hpp"
#include "kernels_filter.hpp"
#include "kernels_clinic.hpp"
#include "filter_host.hpp"
#include "kernels_fct.hpp"
#include "kernels_isopyc.hpp"
#include "kernels_mobi.hpp"
#include "kernels_mobi_gen.hpp"
#include "kernels_mobi_gt.hpp"
#include "uvic_ctx.h"

using namespace uvic;
// Measurement switches read from the environment exist in the experiments build only (-DUVIC_EXPERIMENTS, tools/): the
// shipped library reads UVIC_EXACT (the Fortran overlay's one switch) and nothing else; tests and tools
// reach its cross-check paths through uvic_gpu_set_option.
static const char *uv_env(const char *name) {
#ifdef UVIC_EXPERIMENTS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}
#define UV_STR_(x) #x
#define UV_STR(x) UV_STR_(x)

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
// XCD-aware work-item order: the dispatcher deals workgroups round-robin over
// the 8 XCDs (MI355X_MICROARCH.md "Workgroup dispatch"), so block b and b+8
// share an L2.  Remap so that each XCD walks a CONTIGUOUS range of the logical
// work list; neighbouring tiles (same row, next tracer / next row, same tracer)
// then hit the same L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int b, int total) {
  const int per = (total + 7) / 8;
  return (b % 8) * per + b / 8;
}

// decode a flat cell id into (i,k,j), i fastest
#define CELL_DECODE(c)                                        \
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; \
  const int i = (int)(gid % (c).imt) + 1;                     \
  const int k = (int)((gid / (c).imt) % (c).km) + 1;          \
  const int j = (int)(gid / ((long long)(c).imt * (c).km)) + 1

// latitude-slab runs (uvic_gpu_set_shard js..je): the T,S-derived fields are needed two rows beyond the slab
// (SURVEY.md §8e); rows further out are left alone
#define SLAB_OUT(c, j) ((j) < (c).js - 2 || (j) > (c).je + 2)
__global__ void __launch_bounds__(256) k_isopyc_elements(const uvic_ctx c) {
  if (c.prio & 4) __builtin_amdgcn_s_setprio(3);   // short kernel of a latency chain: win issue arbitration over the bulk passes
  CELL_DECODE(c);
  if (j > c.jmt || i < 2 || i > c.imt - 1 || SLAB_OUT(c, j)) return;
  isopyc_elements_cell(c, i, k, j);
}
__global__ void __launch_bounds__(256) k_isopyc_ai(const uvic_ctx c) {
  CELL_DECODE(c);
  if (j > c.jmt - 1 || i < 2 || i > c.imt - 1 || SLAB_OUT(c, j)) return;
  isopyc_ai_cell(c, i, k, j);
}
__global__ void __launch_bounds__(256) k_isopyc_adv(const uvic_ctx c, double *cf) {
  if (c.prio & 4) __builtin_amdgcn_s_setprio(3);   // short kernel of a latency chain: win issue arbitration over the bulk passes
  CELL_DECODE(c);
  if (j > c.jmt - 1 || SLAB_OUT(c, j)) return;
  isopyc_adv_cell(c, i, k, j, cf + CF_IDX(CF_VE, 0, (size_t)c.imt * c.km * c.jmt));
}
// isopyc_column (kernels_isopyc.hpp) with eight threads per column.  One thread per column walks the levels in batches and
// pays a memory round trip per batch: ~30 us however few columns there are, on the chain every step waits for.  Here the
// eight threads of a column fetch the operands of all levels at once and form the divergence terms (phase 1), one of them
// adds them up top-down in the reference's order through LDS (phase 2: the only sequential part, no memory access in it),
// and all eight store (phase 3).  Same operations in the same order per element: bit-identical to isopyc_column.
#define ISO_COL_PARTS 8
__global__ void __launch_bounds__(64 * ISO_COL_PARTS) k_isopyc_column(const uvic_ctx c, double *cf) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if (c.prio & 4) __builtin_amdgcn_s_setprio(3);   // short kernel of a latency chain: win issue arbitration over the bulk passes
  UV_DIMS(c);
  const int x = threadIdx.x, y = threadIdx.y;
  const int gid = blockIdx.x * 64 + x;
  const int i = gid % imt + 1, j = gid / imt + 1;
  const bool live = !(j < 2 || j > jmt - 1 || i < 2 || i > imt - 1 || SLAB_OUT(c, j));
  double *ld = lds + x, *lvb = lds + (size_t)(km + 1) * 64 + x, *lvs = lds + (size_t)2 * (km + 1) * 64 + x;
  double *vbs = cf + CF_IDX(CF_VB, 0, N3);
  if (live) {
#pragma unroll 2
    for (int k = y; k <= km; k += ISO_COL_PARTS) {
      double d = 0.0, vs = 0.0;
      if (k >= 1 && k <= km - 1)
        d = c.dzt[k - 1] * c.cstr[j - 1] *
            ((c.adv_vetiso[X3(i, k, j)] - c.adv_vetiso[X3(i - 1, k, j)]) * c.dxtr[i - 1] +
             (c.adv_vntiso[X3(i, k, j)] - c.adv_vntiso[X3(i, k, j - 1)]) * c.dytr[j - 1]);
      if (k >= 1) vs = c.tot_n[X3(i, k, j - 1)];
      ld[(size_t)k * 64] = d;
      lvb[(size_t)k * 64] = c.adv_vbt[XF(i, k, j)];
      lvs[(size_t)k * 64] = vs;
      if (!c.diff_cbt_given && k >= 1) c.diff_cbt[X3(i, k, j)] = c.diff_cbt_bg[X3(i, k, j)] + c.K33[X3(i, k, j)];
    }
  }
  __syncthreads();
  if (live && y == 0) {
    const int kz = c.kmt[X2(i, j)];
    double run = 0.0;
    for (int k = 0; k <= km; ++k) {
      double v = 0.0;
      if (k >= 1 && k <= km - 1) {
        run = ld[(size_t)k * 64] + run;
        v = run;
      }
      if (k == kz) v = 0.0;
      ld[(size_t)k * 64] = v;
    }
  }
  __syncthreads();
  if (!live) return;
#define IDXF(ii) XF(ii, k, j)
  for (int k = y; k <= km; k += ISO_COL_PARTS) {
    const double v = ld[(size_t)k * 64], tb = lvb[(size_t)k * 64] + v;
    UV_CYC_STORE(c.adv_vbtiso, IDXF, i, v);
    UV_CYC_STORE(c.tot_b, IDXF, i, tb);
    if (k >= 1) {   // the pair plane of pass A (kernels_col.hpp: CF_VB; at k = km v is zero: adv_vbt itself, tracer.F:1065)
      vbs[2 * X3(i, k, j)] = tb;
      vbs[2 * X3(i, k, j) + 1] = lvs[(size_t)k * 64];
    }
  }
#undef IDXF
}

// one workgroup per (row, local tracer, longitude chunk)
struct TileGrid {
  int r0, nrows, nchunk, total;
  int *zero_word;   // a counter the NEXT kernel on the stream wants cleared (spares a memset node), or null
};
__device__ __forceinline__ bool tile_decode(const uvic_ctx &c, const TileGrid &g, int &row, int &n1, int &chunk) {
  const int L = xcd_remap(blockIdx.x, g.total);
  if (L >= g.total) return false;
  chunk = L % g.nchunk;
  const int rest = L / g.nchunk;
  n1 = c.n0 + rest % c.nt_local + 1;
  row = g.r0 + rest / c.nt_local;
  return true;
}
__global__ void __launch_bounds__(1024) k_fct_rows(const uvic_ctx c, const TileGrid g) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if (c.prio & 2) __builtin_amdgcn_s_setprio(3);   // the T,S passes: a latency chain others wait for
  int row, n1, chunk;
  if (!tile_decode(c, g, row, n1, chunk)) return;
  GpuEnv env;
  fct_rows_block(env, c, n1, row, chunk, g.nchunk, lds);
}
__global__ void __launch_bounds__(1024) k_update_rows(const uvic_ctx c, const TileGrid g) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if (g.zero_word && blockIdx.x == 0 && threadIdx.x == 0) *g.zero_word = 0;
  if (c.prio & 2) __builtin_amdgcn_s_setprio(3);   // the T,S passes: a latency chain others wait for
  int row, n1, chunk;
  if (!tile_decode(c, g, row, n1, chunk)) return;
  GpuEnv env;
  update_rows_block(env, c, n1, row, chunk, g.nchunk, lds);
}
// ---- lane-per-column production path (kernels_col.hpp) ------------------------------
__global__ void __launch_bounds__(256) k_adv_vel_hor(const uvic_ctx c) {
  CELL_DECODE(c);
  if (j > c.jmt) return;
  adv_vel_hor_cell(c, i, k, j);
}
__global__ void __launch_bounds__(64) k_adv_vel_vert(const uvic_ctx c) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = gid % c.imt + 1, j = gid / c.imt + 1;
  if (j < 2 || j > c.jmt || i < 2 || i > c.imt - 1) return;
  adv_vel_vert_column(c, i, j);
}
__global__ void __launch_bounds__(256) k_vmixc(const uvic_ctx c) {
  CELL_DECODE(c);
  if (j < 2 || j > c.jmt - 1 || i < 2 || i > c.imt - 1) return;
  vmixc_cell(c, i, k, j);
}
// What a step whose T,S-derived fields were computed ahead has to do once its inputs have arrived from the host
// (uvic_gpu_overlay_inputs without adv_vbt): adv_vbt by continuity (adv_vel_vert_column), the total velocities
// (k_tot_vel) and the vertical-diffusion coefficient from the step's diff_cbt (coef_bv_cell) -- three short kernels of
// the T,S chain, each ~10 us of launch latency plus a walk down the column -- as one cell-parallel kernel: every cell sums
// the divergences of the levels above it itself, in the order of the downward integration (the same bits, ~km/2 extra
// cached loads per cell instead of a serial walk); i = 1 and imt sum their cyclic images.
__global__ void __launch_bounds__(256) k_inputs_cell(const uvic_ctx c, double *cf) {
  if (c.prio & 4) __builtin_amdgcn_s_setprio(3);
  CELL_DECODE(c);
  if (j > c.jmt) return;
  const int imt = c.imt, km = c.km, jmt = c.jmt;
  const size_t N3 = (size_t)imt * km * jmt;
  const long long rowstride = (long long)imt * km;
  const int is = (i == 1) ? imt - 1 : ((i == imt) ? 2 : i);
  double *vbt = const_cast<double *>(c.adv_vbt);
  const size_t f0 = (size_t)(i - 1) + (size_t)imt * ((size_t)(km + 1) * (j - 1));   // face 0 of the column
  const size_t q = (size_t)gid, fq = f0 + (size_t)imt * k;
  double acc;
  if (j >= 2) {
    const double dx = c.dxtr[is - 1], dy = c.dytr[j - 1], cs = c.cstr[j - 1];
    size_t qs = (size_t)(is - 1) + (size_t)imt * ((size_t)km * (j - 1));
    acc = 0.0;
    for (int kk = 1; kk <= k; ++kk, qs += imt) {
      const double div = ((c.adv_vet[qs] - c.adv_vet[qs - 1]) * dx + (c.adv_vnt[qs] - c.adv_vnt[qs - rowstride]) * dy) * cs * c.dzt[kk - 1];
      acc = div + acc;
    }
    vbt[fq] = acc;
    if (k == 1) { vbt[f0] = 0.0; c.tot_b[f0] = 0.0 + c.adv_vbtiso[f0]; }
  } else {   // row 1: adv_vel leaves adv_vbt alone there
    acc = vbt[fq];
    if (k == 1) c.tot_b[f0] = vbt[f0] + c.adv_vbtiso[f0];
  }
  const double e = c.adv_vet[q] + c.adv_vetiso[q], n = c.adv_vnt[q] + c.adv_vntiso[q];
  c.tot_e[q] = e;
  c.tot_n[q] = n;
  c.tot_b[fq] = acc + c.adv_vbtiso[fq];
  cf[CF_IDX(CF_VE, q, N3)] = e;
  cf[CF_IDX(CF_VN, q, N3)] = n;
  cf[CF_IDX(CF_VB, q, N3)] = (k < km) ? acc + c.adv_vbtiso[fq] : acc;
  cf[CF_IDX(CF_VS, q, N3)] = (j >= 2) ? c.adv_vnt[q - rowstride] + c.adv_vntiso[q - rowstride] : 0.0;
  if (j >= 2 && j <= jmt - 1 && i >= 2 && i <= imt - 1 && !SLAB_OUT(c, j)) {
    const double dcb = c.vmix_dev ? vmixc_cell(c, i, k, j) : c.diff_cbt[q];   // (vmixc on the device: uvic_gpu_overlay_inputs)
    if (k <= km - 1) cf[CF_IDX(CF_BV, q, N3)] = dcb * c.dzwr[k] * (1.0 - c.aidif);
  }
}
// MOBI's four forcing planes of an ocean segment (light, ice cover, ice and snow thickness) straight out of the caller's
// page-locked arrays: 4 x imt*jmt doubles at the head of the chain that needs them -- no copy engine, no event between streams
struct Pull4 { const double *src[4]; double *dst[4]; };
__global__ void __launch_bounds__(256) k_pull4(const Pull4 p, int n) {
  const int q = blockIdx.y;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) p.dst[q][e] = p.src[q][e];
}
#if defined(UVIC_EXPERIMENTS)
// stress test of the schedule (not in the shipped library): holds the stream it is launched on for `us` microseconds
__global__ void k_stall(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
#endif
// ---- time-average steps (O_time_averages + O_save_convection, timavgperts): what `tracer` accumulates inside its own loops
// The convection diagnostics of convct2 (source/mom/convect.F:183-191, 279-283, 295-301): pe(i,j) is ONE running sum -- the
// potential energy of the column before convection added level by level, then the same after convection subtracted level by
// level, then divided by c2dtts -- so the two halves run before and after the walk on the same plane, every interior
// column (land included: the reference's residue of rounding is reproduced, not assumed zero); totalk and vdepth come
// from the segments the walk recorded.  diag: (imt, jmt, 3) = totalk, vdepth, pe.
__global__ void __launch_bounds__(64) k_conv_pe(const uvic_ctx c, const double *zt, double grav, double *diag, int phase) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int ni = c.imt - 2, nrows = c.je - c.js + 1;
  if (gid >= ni * nrows) return;
  const int i = gid % ni + 2, j = c.js + gid / ni;
  UV_DIMS(c);
  const size_t ij = X2(i, j), N2 = (size_t)imt * jmt;
  const double *ts = c.t_taup1;
  double pe = phase == 0 ? 0.0 : diag[2 * N2 + ij];
  for (int k = 1; k <= km; ++k) {
    const double ru = eos_dens(c.c, km, ts[X3(i, k, j)] - c.to[k - 1], ts[X3(i, k, j) + N3] - c.so[k - 1], k);
    const double term = grav * zt[k - 1] * ru * c.dztxcl[k - 1];
    pe = phase == 0 ? pe + term : pe - term;
  }
  if (phase == 0) { diag[2 * N2 + ij] = pe; return; }
  diag[2 * N2 + ij] = pe / c.c2dtts;
  double totalk = 0.0, vdepth = 0.0;
  const int nseg = c.cv_nseg[ij];
  for (int s = 1; s <= nseg; ++s) {
    const int kt = c.cv_kt[X3(i, s, j)], kb = c.cv_kb[X3(i, s, j)];
    totalk = totalk + (double)(kb - kt + 1);
    if (kt == 1) vdepth = c.zw[kb - 1];
  }
  diag[ij] = totalk;
  diag[N2 + ij] = vdepth;
}
// delta 14C of the final t(tau+1) as a field (tracer.F:1329-1340): what ta_dc14 accumulates
__global__ void __launch_bounds__(256) k_dc14_field(const uvic_ctx c, int ic14, int idic, double rc14std, double *out) {
  CELL_DECODE(c);
  if (j > c.jmt) return;
  const size_t N3 = (size_t)c.imt * c.km * c.jmt, q = (size_t)gid;
  const double rrc14std = 1000. / rc14std;
  out[q] = (rrc14std * c.t_taup1[(size_t)(ic14 - 1) * N3 + q] / (c.t_taup1[(size_t)(idic - 1) * N3 + q] + UV_EPSLN) - 1000.) * c.tmask[q];
}
// ---- baroclinic momentum step (kernels_clinic.hpp) -----------------------------------
#define COL_DECODE(m)                                                 \
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;              \
  const int i = gid % (m).imt + 1, j = gid / (m).imt + 1
__global__ void __launch_bounds__(256) k_state(const uvic_mom_ctx m) {
  CELL_DECODE(m);
  if (j < 2 || j > m.jmt) return;
  state_cell(m, i, k, j);
}
__global__ void __launch_bounds__(64) k_clinic_gradp(const uvic_mom_ctx m) {
  COL_DECODE(m);
  if (j < m.js || j > m.je || i < 2 || i > m.imt - 1) return;
  clinic_gradp_column(m, i, j);
}
__global__ void __launch_bounds__(256) k_clinic_tend(const uvic_mom_ctx m) {
  CELL_DECODE(m);
  if (j < m.js || j > m.je || i < 2 || i > m.imt - 1) return;
  clinic_tend_cell(m, i, k, j);
}
// sbc_flags bit 0: also isbcu/asbcu of this column (they read u(tau) only, so their place in the sequence is free)
__global__ void __launch_bounds__(64) k_clinic_finish(const uvic_mom_ctx m, int sbc_flags, double rts) {
  COL_DECODE(m);
  if (j < m.js || j > m.je || i < 2 || i > m.imt - 1) return;
  clinic_finish_column(m, i, j);
  if (sbc_flags & 1) clinic_sbcu_cell(m, i, j, (sbc_flags >> 1) & 3, rts);
}
__global__ void __launch_bounds__(64) k_add_ext_mode(const uvic_mom_ctx m, const double *psi, double *u1, double *u2) {
  COL_DECODE(m);
  if (j < 1 || j > m.jmt - 1 || i < 2 || i > m.imt - 1) return;
  add_ext_mode_column(m, i, j, psi, u1, u2);
}
// polar filter of u(tau+1): one workgroup per strip of a level of a row (both components)
__global__ void __launch_bounds__(1024) k_filuv(const uvic_mom_ctx m, const FilterItem *items, const double *mats, const double *spsin,
                                                const double *spcos, int nitems) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if ((int)blockIdx.x >= nitems) return;
  const FilterItem it = items[blockIdx.x];
  if (it.j < m.js || it.j > m.je) return;
  GpuEnv env;
  filuv_block(env, m.imt, m.km, it, mats, spsin, spcos, m.up1, m.up2, lds);
}
__global__ void __launch_bounds__(64) k_filuv_mean(const uvic_mom_ctx m, const int *rows, int nrows) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = gid % m.imt + 1, r = gid / m.imt;
  if (r >= nrows || i < 2 || i > m.imt - 1) return;
  const int j = rows[r];
  if (j < m.js || j > m.je) return;
  filuv_mean_column(m.imt, m.km, i, j, m.kmu, m.hr, m.dzt, m.up1, m.up2);
}
__global__ void __launch_bounds__(256) k_ai_coef(const uvic_ctx c, double *cf, int store_ai) {
  if (c.prio & 4) __builtin_amdgcn_s_setprio(3);   // short kernel of a latency chain: win issue arbitration over the bulk passes
  CELL_DECODE(c);
  if (j > c.jmt - 1 || i < 2 || i > c.imt - 1 || SLAB_OUT(c, j)) return;
  ai_coef_cell(c, cf, i, k, j, store_ai);
}
__global__ void __launch_bounds__(256) k_coef_bv(const uvic_ctx c, double *cf) {
  CELL_DECODE(c);
  if (j > c.jmt - 1 || i < 2 || i > c.imt - 1 || SLAB_OUT(c, j)) return;
  coef_bv_cell(c, cf, i, k, j);
}
// The total advective velocities adv_v?t + adv_v?tiso are formed by the isopyc kernels.  When those ran a step ahead and the
// host has since uploaded this step's adv_vet/vnt/vbt (the Fortran overlay does, every step), the sums are formed again
// from the new velocities and the GM velocities computed ahead: the same additions, element by element.
__global__ void __launch_bounds__(256) k_tot_vel(const uvic_ctx c, double *cf) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long N3 = (long long)c.imt * c.km * c.jmt, NF = (long long)c.imt * (c.km + 1) * c.jmt;
  if (gid < N3) {
    const double e = c.adv_vet[gid] + c.adv_vetiso[gid], n = c.adv_vnt[gid] + c.adv_vntiso[gid];
    c.tot_e[gid] = e;
    c.tot_n[gid] = n;
    // ... and their packed copies for pass A (isopyc_adv_cell, isopyc_column)
    const long long rowstride = (long long)c.imt * c.km;
    const int j = (int)(gid / rowstride) + 1, k = (int)(gid % rowstride) / c.imt + 1, i = (int)(gid % c.imt) + 1;
    const size_t fq = (size_t)(i - 1) + (size_t)c.imt * ((size_t)k + (size_t)(c.km + 1) * (j - 1));   // face k of the column
    cf[CF_IDX(CF_VE, gid, N3)] = e;
    cf[CF_IDX(CF_VN, gid, N3)] = n;
    cf[CF_IDX(CF_VB, gid, N3)] = (k < c.km) ? c.adv_vbt[fq] + c.adv_vbtiso[fq] : c.adv_vbt[fq];
    cf[CF_IDX(CF_VS, gid, N3)] = (j >= 2) ? c.adv_vnt[gid - rowstride] + c.adv_vntiso[gid - rowstride] : 0.0;
  }
  if (gid < NF) c.tot_b[gid] = c.adv_vbt[gid] + c.adv_vbtiso[gid];
}
// pass A of the bulk launch: the four waves of a workgroup are four tracers of the same 64 lanes and share the coefficient and
// velocity pairs of a level through LDS (g.total counts waves = waves of the lane map x the tracer count rounded up to a
// multiple of four; a wave beyond the launch's tracers stands in for the first one, brings its share of the pairs and
// stores nothing)
__global__ void __launch_bounds__(256) k_colfct(const uvic_ctx c, const double *cf, double *S, const ColGrid g) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int nblk = g.total / 4;
  const int blk = xcd_remap(blockIdx.x, nblk);
  if (blk >= nblk) return;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int ngrp = (c.nt_local + 3) / 4;              // workgroups per wave of the lane map
  const int nl = (blk % ngrp) * 4 + wv;
  const int code = g.lanes[(size_t)(blk / ngrp) * 64 + threadIdx.x];
  colfct_wave<true>(c, cf, S, code, c.n0 + (nl < c.nt_local ? nl : 0) + 1, nl < c.nt_local, lds, wv);
}
// pass A with every load the wave's own (T and S when they go through the column kernels, set_exact(2)): work item =
// (wave of the lane map, tracer), tracers fastest
__global__ void __launch_bounds__(256) k_colfct_ts(const uvic_ctx c, const double *cf, double *S, const ColGrid g) {
  if (g.zero_word && blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0) *g.zero_word = 0;   // (a counter a later kernel of the stream wants cleared)
  const int nblk = (g.total + 3) / 4;
  const int blk = xcd_remap(blockIdx.x, nblk);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int item = blk * 4 + wv;
  if (blk >= nblk || item >= g.total) return;
  const int code = g.lanes[(size_t)(item / c.nt_local) * 64 + threadIdx.x];
  if (c.prio & 2) __builtin_amdgcn_s_setprio(3);   // the short T,S passes: a latency chain others wait for
  colfct_wave<false>(c, cf, S, code, c.n0 + item % c.nt_local + 1, true);
}
// pass B: one wave per (wave of the pass-B lane map, tracer), the waves of one tracer next to each other (rows ascending), so
// that rows r-1 and r of the final fluxes a wave reads are its neighbour's centre row
__global__ void __launch_bounds__(64) k_colupd(const uvic_ctx c, const double *S, const ColGrid g) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int blk = xcd_remap(blockIdx.x, g.total);
  if (blk >= g.total) return;
  const int code = g.lanes[(size_t)(blk % g.nwaves) * 64 + threadIdx.x];
  if (c.prio & 2) __builtin_amdgcn_s_setprio(3);
  colupd_wave(c, S, lds, code, c.n0 + blk / g.nwaves + 1);
}
// Pass B of T and S and the convective walk in one launch: a workgroup is two waves, T and S of the same 64 ocean columns;
// each solves its column (t(tau+1) stored, and kept in its LDS), then the first wave walks the columns as convect_ts_column
// does -- from LDS, no reload.  One kernel boundary less on the chain every step waits for.
__global__ void __launch_bounds__(128) k_colupd_conv_ts(const uvic_ctx c, const double *S, const ColGrid g, int *cvl) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int nblk = g.nwaves;
  const int blk = xcd_remap(blockIdx.x, nblk);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const size_t per_wave = (size_t)2 * (c.km + 1) * 64;
  double *tab = lds + 2 * per_wave;
  for (int q = threadIdx.y * 64 + threadIdx.x; q < 12 * c.km; q += 128) {   // the per-level tables of the walk
    const int a = q / c.km, k = q % c.km;
    tab[q] = a < 9 ? c.c[q] : (a == 9 ? c.to[k] : (a == 10 ? c.so[k] : c.dztxcl[k]));
  }
  if (c.prio & 2) __builtin_amdgcn_s_setprio(3);
  int code = 0;
  if (blk < nblk) {
    code = g.lanes[(size_t)blk * 64 + threadIdx.x];
    colupd_wave(c, S, lds + (size_t)wv * per_wave, code, c.n0 + wv + 1, true);
  }
  __syncthreads();
  if (blk >= nblk || wv != 0 || !COL_LANE_OWNED(code)) return;
  const int i = COL_LANE_I(code), j = COL_LANE_R(code);
  double *zT = lds + (size_t)(c.km + 1) * 64, *zS = lds + per_wave + (size_t)(c.km + 1) * 64;   // zwork[k][lane], k = 1..km
  convect_ts_column(c, i, j, zT + 64 + threadIdx.x, zS + 64 + threadIdx.x, 64, tab, true);
  const int wid = (i - 1) + c.imt * (j - 1);
  if (cvl && c.cv_nseg[wid] > 0) cvl[1 + atomicAdd(cvl, 1)] = wid;
}
// ---- T and S in the reference's own order of operations (kernels_colx.hpp) --------------------------------
// a workgroup = four waves on the same 64 lanes of the pass-A lane map: (T, advective), (S, advective), (T, diffusive), (S, diffusive)
__global__ void __launch_bounds__(256) k_colx_fct(const uvic_ctx c, const ColxOut o, const ColGrid g) {
  if (g.zero_word && blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0) *g.zero_word = 0;   // (a counter a later kernel of the stream wants cleared)
  const int blk = xcd_remap(blockIdx.x, g.nwaves);
  if (blk >= g.nwaves) return;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int tr = wv & 1;
  if (tr >= c.nt_local) return;
  const int code = g.lanes[(size_t)blk * 64 + threadIdx.x];
  if (c.prio & 2) __builtin_amdgcn_s_setprio(3);
  if (wv < 2) colx_adv_wave(c, o, code, c.n0 + tr + 1, tr);
  else colx_dif_wave(c, o, code, c.n0 + tr + 1, tr);
}
// pass B of the local tracers among T and S, one wave per (wave of the pass-B lane map, tracer)
__global__ void __launch_bounds__(64) k_colx_upd(const uvic_ctx c, const ColxOut o, const ColGrid g) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int blk = xcd_remap(blockIdx.x, g.total);
  if (blk >= g.total) return;
  const int tr = blk / g.nwaves;
  const int code = g.lanes[(size_t)(blk % g.nwaves) * 64 + threadIdx.x];
  colx_upd_wave(c, o, lds, code, c.n0 + tr + 1, tr);
}
// ... and, for T and S together, with the convective walk in the same launch (as k_colupd_conv_ts)
__global__ void __launch_bounds__(128) k_colx_upd_conv(const uvic_ctx c, const ColxOut o, const ColGrid g, int *cvl) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int nblk = g.nwaves;
  const int blk = xcd_remap(blockIdx.x, nblk);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const size_t per_wave = (size_t)2 * (c.km + 1) * 64;
  double *tab = lds + 2 * per_wave;
  for (int q = threadIdx.y * 64 + threadIdx.x; q < 12 * c.km; q += 128) {   // the per-level tables of the walk
    const int a = q / c.km, k = q % c.km;
    tab[q] = a < 9 ? c.c[q] : (a == 9 ? c.to[k] : (a == 10 ? c.so[k] : c.dztxcl[k]));
  }
  if (c.prio & 2) __builtin_amdgcn_s_setprio(3);
  int code = 0;
  if (blk < nblk) {
    code = g.lanes[(size_t)blk * 64 + threadIdx.x];
    colx_upd_wave(c, o, lds + (size_t)wv * per_wave, code, c.n0 + wv + 1, wv);
  }
  __syncthreads();
  if (blk >= nblk || wv != 0 || !COL_LANE_OWNED(code)) return;
  const int i = COL_LANE_I(code), j = COL_LANE_R(code);
  double *zT = lds + (size_t)(c.km + 1) * 64, *zS = lds + per_wave + (size_t)(c.km + 1) * 64;   // zwork[k][lane], k = 1..km
  convect_ts_column(c, i, j, zT + 64 + threadIdx.x, zS + 64 + threadIdx.x, 64, tab, true);
  const int wid = (i - 1) + c.imt * (j - 1);
  if (cvl && c.cv_nseg[wid] > 0) cvl[1 + atomicAdd(cvl, 1)] = wid;
}
__global__ void __launch_bounds__(128) k_convect(const uvic_ctx c) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = gid % c.imt + 1, j = gid / c.imt + 1;
  if (j < c.js || j > c.je || i < 2 || i > c.imt - 1) return;
  convect_column(c, i, j);
}
// convection in two passes (kernels_fct.hpp): T,S walk with the column staged in LDS, then replay
// Column-local kernels (convection, MOBI) run over the ocean columns only: `ij` lists the columns with
// kmt > 0 among i = 2..imt-1, j = 2..jmt-1 as (i-1) + imt*(j-1), row by row, so that a latitude slab is a
// contiguous range [first, first+count) and neighbouring lanes still read neighbouring addresses.  Land
// columns are never touched (their sources stay zero, see src_clean()).
struct WetCols {
  const int *ij;
  int first, count;
};
#define WET_DECODE(w, cc)                          \
  const int wid_ = (w).ij[(w).first + (cc)];       \
  const int i = wid_ % c.imt + 1, j = wid_ / c.imt + 1
__global__ void __launch_bounds__(64) k_convect_ts(const uvic_ctx c, const WetCols w, int *cvl) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (c.prio & 2) __builtin_amdgcn_s_setprio(3);
  // the per-level tables of the walk go to LDS first (12 km doubles for the 64 columns of the workgroup)
  double *tab = lds + (size_t)2 * c.km * 64;
  for (int q = threadIdx.x; q < 12 * c.km; q += 64) {
    const int a = q / c.km, k = q % c.km;
    tab[q] = a < 9 ? c.c[q] : (a == 9 ? c.to[k] : (a == 10 ? c.so[k] : c.dztxcl[k]));
  }
  __syncthreads();
  if (gid >= w.count) return;
  WET_DECODE(w, gid);
  convect_ts_column(c, i, j, lds + threadIdx.x, lds + (size_t)c.km * 64 + threadIdx.x, 64, tab);
  if (cvl && c.cv_nseg[wid_] > 0) cvl[1 + atomicAdd(cvl, 1)] = wid_;
}
// The columns in which the walk mixed something, as a list: cvl[0] counts them, cvl[1..] holds their ids ((i-1) + imt*(j-1)).
// Few columns convect in a step, and convect_apply over the list is a handful of waves instead of one thread per
// (ocean column, tracer) that finds nothing to do.
// (A handful of waves on the main stream between pass B and the next pass A: what counts is its chain of dependent memory
// round trips -- each costs microseconds while the other streams keep the memory system busy.  The list entry is fetched
// beside the count, a segment's levels in batches of eight instead of one round trip per level: same sums in the same
// order as convect_apply_cell.)
__global__ void __launch_bounds__(256) k_convect_apply_list(const uvic_ctx c, const int *cvl, int maxcol) {
  UV_DIMS(c);
  const int ntr = c.nt - 2;
  const double *dz = c.dztxcl;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;; q += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(q / ntr), n = (int)(q % ntr) + 3;   // tracers fastest: a wave shares its columns' segment records
    const int ncol = cvl[0];
    const int wid = cvl[1 + (col < maxcol ? col : maxcol - 1)];   // (fetched beside the count, not after it)
    if (col >= ncol) break;
    const int i = wid % imt + 1, j = wid / imt + 1;
    const int nseg = c.cv_nseg[X2(i, j)];
    double *t = c.t_taup1 + (size_t)(n - 1) * N3;
    for (int s = 1; s <= nseg; ++s) {
      const int kt = c.cv_kt[X3(i, s, j)], kb = c.cv_kb[X3(i, s, j)];
      const double zsm = c.cv_z[X3(i, s, j)];
      double tsm3 = 0.0;
      for (int k0 = kt; k0 <= kb; k0 += 8) {
        double v[8];
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {
          const int k = k0 + u <= kb ? k0 + u : kb;
          v[u] = t[X3(i, k, j)] * dz[k - 1];
        }
        _Pragma("unroll") for (int u = 0; u < 8; ++u)
          if (k0 + u <= kb) tsm3 = tsm3 + v[u];
      }
      const double tmx3 = tsm3 / zsm;
      for (int k = kt; k <= kb; ++k) t[X3(i, k, j)] = tmx3;
    }
    const int ic = (i == 2) ? imt : ((i == imt - 1) ? 1 : 0);
    if (ic && nseg > 0)
      for (int k0 = 1; k0 <= km; k0 += 8) {
        double v[8];
        _Pragma("unroll") for (int u = 0; u < 8; ++u) v[u] = t[X3(i, (k0 + u <= km ? k0 + u : km), j)];
        _Pragma("unroll") for (int u = 0; u < 8; ++u)
          if (k0 + u <= km) t[X3(ic, k0 + u, j)] = v[u];
      }
  }
}
__global__ void __launch_bounds__(256) k_convect_apply(const uvic_ctx c, const WetCols w) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = gid / w.count + 3;
  if (n > c.nt) return;
  WET_DECODE(w, gid % w.count);
  convect_apply_cell(c, i, j, n);
}
// cell-parallel MOBI passes: thread = (ocean column, level), columns fastest
__global__ void __launch_bounds__(128) k_mobi_pre(const uvic_ctx c, const mobi_dev m, const WetCols w) {
  if (c.prio & 4) __builtin_amdgcn_s_setprio(3);   // short kernel of a latency chain: win issue arbitration over the bulk passes
  // two threads per cell: the first half of the grid does the carbonate chemistry, the second the rest (whole waves each)
  const int half = (int)((gridDim.x + 1) / 2);
  const int part = blockIdx.x < half ? 1 : 2;
  const int gid = (blockIdx.x - (part == 2 ? half : 0)) * blockDim.x + threadIdx.x;
  const int k = gid / w.count + 1;
  if (k > c.km) return;
  WET_DECODE(w, gid % w.count);
  mobi_pre_cell(c, m, i, k, j, part);
}
__global__ void __launch_bounds__(128) k_mobi_post(const uvic_ctx c, const mobi_dev m, const WetCols w) {
  if (c.prio & 4) __builtin_amdgcn_s_setprio(3);   // short kernel of a latency chain: win issue arbitration over the bulk passes
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = gid / w.count + 1;
  if (k > c.km) return;
  WET_DECODE(w, gid % w.count);
  mobi_post_cell(c, m, i, k, j);
}
// polar Fourier filter: one workgroup per (strip of a level of a row, local tracer)
__global__ void __launch_bounds__(1024) k_filt(const uvic_ctx c, const FilterItem *items, const double *mats, int nitems) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int item = blockIdx.x % nitems, n1 = c.n0 + blockIdx.x / nitems + 1;
  const FilterItem it = items[item];
  if (it.j < c.js || it.j > c.je) return;     // latitude-slab runs filter their own rows only
  GpuEnv env;
  filt_block(env, c, it, n1, mats, lds);
}
// team form: four waves (one per SIMD of a CU) share 64 columns, each with its own compile-time
// role, see kernels_mobi.hpp
template <int R>
struct GpuTeam {
  static constexpr bool team = true;
  static constexpr int role = R;
  int lane;
  double *xs;
  unsigned xc;
#ifdef UV_MOBI_TIMING
  long long tq[8];
#endif
  __device__ __forceinline__ void sync() const { __syncthreads(); }
};
template <int R>
__device__ __forceinline__ void mobi_team_role(const uvic_ctx &c, const mobi_dev &m, double *lds, int i, int j, bool live, int kmax) {
  GpuTeam<R> T{(int)threadIdx.x, lds, 0u};
#ifdef UV_MOBI_TIMING
  for (int q = 0; q < 8; ++q) T.tq[q] = 0;
  const long long tk0 = clock64();
#endif
  mobi_column_body(T, c, m, i, j, live, kmax);
#ifdef UV_MOBI_TIMING
  if (threadIdx.x == 0 && (blockIdx.x == 60 || blockIdx.x == 100))
    printf("blk %d role %d kmax %d total %lld | role %lld put %lld bar %lld get %lld shared+own %lld put %lld bar %lld get %lld\n", blockIdx.x, R, kmax,
           clock64() - tk0, T.tq[0], T.tq[1], T.tq[2], T.tq[3], T.tq[4], T.tq[5], T.tq[6], T.tq[7]);
#endif
}
#ifndef UV_TEAM_WAVES_PER_EU
#define UV_TEAM_WAVES_PER_EU 3
#endif
#if defined(UV_TEAM_NUM_VGPR)
#define UV_TEAM_OCC __attribute__((amdgpu_num_vgpr(UV_TEAM_NUM_VGPR)))
#elif UV_TEAM_WAVES_PER_EU > 0
#define UV_TEAM_OCC __attribute__((amdgpu_waves_per_eu(UV_TEAM_WAVES_PER_EU, UV_TEAM_WAVES_PER_EU)))
#else
#define UV_TEAM_OCC
#endif
// register budget: the team's waves are resident for the whole MOBI pass on a side stream, so what
// matters is how many transport waves still fit beside one of them on a SIMD
__global__ void __launch_bounds__(256) UV_TEAM_OCC k_mobi_team(const uvic_ctx c, const mobi_dev m, const WetCols w) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int gid = blockIdx.x * 64 + threadIdx.x;
  const bool live = gid < w.count;
  WET_DECODE(w, live ? gid : 0);   // lanes beyond the list walk the first column and store nothing
  int kmax = live ? c.kmt[(size_t)(i - 1) + (size_t)c.imt * (j - 1)] : 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) kmax = max(kmax, __shfl_xor(kmax, off));
  if (!(c.prio & 1)) __builtin_amdgcn_s_setprio(2);
#ifdef UV_CLOCK_PROBE
  // diagnostic build: shader clock held during this (long) kernel = d(s_memtime) / d(s_memrealtime) x 100 MHz
  const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  switch (threadIdx.y) {   // wave-uniform: every wave runs the code specialised for its role
    case 0: mobi_team_role<0>(c, m, lds, i, j, live, kmax); break;
    case 1: mobi_team_role<1>(c, m, lds, i, j, live, kmax); break;
    case 2: mobi_team_role<2>(c, m, lds, i, j, live, kmax); break;
    default: mobi_team_role<3>(c, m, lds, i, j, live, kmax); break;
  }
#ifdef UV_CLOCK_PROBE
  if (threadIdx.x == 0 && threadIdx.y == 0 && (blockIdx.x == 7 || blockIdx.x == 70)) {
    const unsigned long long ck1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
    printf("clock probe blk %d: %llu shader cycles in %llu ticks of 10 ns = %.0f MHz\n", blockIdx.x, ck1 - ck0, rt1 - rt0,
           (double)(ck1 - ck0) / (double)(rt1 - rt0) * 100.0);
  }
#endif
}
// any other option set (kernels_mobi_gen.hpp): one thread per ocean column, the reference's three loops.  One kernel
// per option set of SURVEY.md 2c (own register allocation each), one with the flags at run time for anything else.
__global__ void __launch_bounds__(128) k_mobi_gen_pre(const uvic_ctx c, const mobi_dev m, const WetCols w) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = gid / w.count + 1;
  if (k > c.km) return;
  WET_DECODE(w, gid % w.count);
  mobig_pre_cell(c, m, i, k, j);
}
template <int TN15, int TC13, int TCACO3, int TSIL>
__global__ void __launch_bounds__(64) k_mobi_gen(const uvic_ctx c, const mobi_dev m, const WetCols w) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= w.count) return;
  WET_DECODE(w, gid);
  mobig_column<TN15, TC13, TCACO3, TSIL>(c, m, i, j);
}
// ... and the sets with prognostic CaCO3 (F, run/mk.in's) in the team form (kernels_mobi_gt.hpp)
template <int R, int TN15, int TC13, int TCACO3, int TSIL>
__device__ __forceinline__ void mobigt_team_role(const uvic_ctx &c, const mobi_dev &m, double *lds, int i, int j, bool live, int kmax) {
  GpuTeam<R> T{(int)threadIdx.x, lds, 0u};
  mobigt_column<GpuTeam<R>, TN15, TC13, TCACO3, TSIL>(T, c, m, i, j, live, kmax);
}
// (register budget: the nt = 37 set carries 32 pools; at set C's 168 registers it spills 92 of them, at 256 none, and a
// 256-register team wave still shares its SIMD with one pass-A wave)
template <int TN15, int TC13, int TCACO3, int TSIL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_mobi_gteam(const uvic_ctx c, const mobi_dev m, const WetCols w) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int gid = blockIdx.x * 64 + threadIdx.x;
  const bool live = gid < w.count;
  WET_DECODE(w, live ? gid : 0);   // lanes beyond the list walk the first column and store nothing
  int kmax = live ? c.kmt[(size_t)(i - 1) + (size_t)c.imt * (j - 1)] : 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) kmax = max(kmax, __shfl_xor(kmax, off));
  if (!(c.prio & 1)) __builtin_amdgcn_s_setprio(2);
  switch (threadIdx.y) {   // wave-uniform: every wave runs the code specialised for its role
    case 0: mobigt_team_role<0, TN15, TC13, TCACO3, TSIL>(c, m, lds, i, j, live, kmax); break;
    case 1: mobigt_team_role<1, TN15, TC13, TCACO3, TSIL>(c, m, lds, i, j, live, kmax); break;
    case 2: mobigt_team_role<2, TN15, TC13, TCACO3, TSIL>(c, m, lds, i, j, live, kmax); break;
    default: mobigt_team_role<3, TN15, TC13, TCACO3, TSIL>(c, m, lds, i, j, live, kmax); break;
  }
}
template <int TN15, int TC13, int TCACO3, int TSIL>
__global__ void __launch_bounds__(128) k_mobi_gpost(const uvic_ctx c, const mobi_dev m, const WetCols w) {
  if (c.prio & 4) __builtin_amdgcn_s_setprio(3);
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = gid / w.count + 1;
  if (k > c.km) return;
  WET_DECODE(w, gid % w.count);
  mobigt_post_cell<TN15, TC13, TCACO3, TSIL>(c, m, i, k, j);
}
__global__ void __launch_bounds__(64) k_mobi(const uvic_ctx c, const mobi_dev m, const WetCols w) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= w.count) return;
  WET_DECODE(w, gid);
  // few, long, latency-bound waves: let them win issue arbitration over the streaming kernels
  // that share their SIMDs when the sources are computed one step ahead on the side stream
  __builtin_amdgcn_s_setprio(3);
  mobi_column_kernel(c, m, i, j);
}

// ---- the time-step integrals of O_time_step_monitor (diagt1, u09/mom/tracer.F:1516-1537; :1329-1353; clinic.F:616-630) ----
// With tsiint = tsiper (the shipped run/control.in) every ocean step is one of these: the sums are formed here so that the
// step stays on the device.  One thread per (row, level, tracer) adds along i in the reference's order (a running sum is
// not associative: the same order gives the same bits).  tp = t(tau+1) BEFORE convection (diagt1 is called at tracer.F:1161).
// acc: tbar, travar, dtabs, each (0:km, nt, jmt) as source/common/diag.h declares them.
// Eight rows (k, j, n) per wave.  Row by row the 64 lanes fetch 64 consecutive columns at once (coalesced) and form the
// three terms of each column in parallel, into an LDS tile; then lane r adds up row r of the tile column by column -- the
// sequential sum of diagt1 (tracer.F:1516-1537), to the bit -- so a row costs one load instruction per 64 columns instead
// of 64 scattered ones and the serial part runs eight rows abreast (eight, not sixteen or four: 12 KB of LDS per wave
// leave enough waves per CU to cover the load latency; 120 / 85 / 68 us for 16 / 8 / 4 rows, but the T,S launch likes 8).
#define TSI_ROWS 8
__global__ void __launch_bounds__(64) k_tsi_rows(const uvic_ctx c, double *acc) {
  __shared__ double t3[TSI_ROWS][65], t1[TSI_ROWS][65], t2[TSI_ROWS][65];
  __shared__ double f_w[TSI_ROWS], f_x[TSI_ROWS];        // per row: dzt(k) and cst(j)*dyt(j); r2dt/dtxcel(k)
  __shared__ long long base_t[TSI_ROWS], base_m[TSI_ROWS];   // per row: offset of the row in t (tracer n) and in tmask; -1: no row
  const int lane = threadIdx.x;
  const int nrows = c.je - c.js + 1;
  const int total = c.km * nrows * c.nt_local, g0 = blockIdx.x * TSI_ROWS;
  const size_t N3 = (size_t)c.imt * c.km * c.jmt;
  int k = 0, j = 0, n = 0;
  if (lane < TSI_ROWS) {
    const int gid = g0 + lane;
    long long bt = -1, bm = -1;
    double fw = 0.0, fx = 0.0;
    if (gid < total) {
      k = gid % c.km + 1; j = c.js + (gid / c.km) % nrows; n = c.n0 + gid / (c.km * nrows) + 1;
      bm = (long long)c.imt * ((long long)(k - 1) + (long long)c.km * (j - 1));
      bt = (long long)(n - 1) * (long long)N3 + bm;
      fw = c.dzt[k - 1]; fx = (1.0 / c.c2dtts) / c.dtxcel[k - 1];
      f_x[lane] = fx;
    }
    base_t[lane] = bt; base_m[lane] = bm; f_w[lane] = fw;
    t3[lane][64] = (gid < total) ? c.cst[j - 1] * c.dyt[j - 1] : 0.0;   // (the spare column of the tile: cosdyt of the row)
  }
  __syncthreads();
  double s_bar = 0.0, s_var = 0.0, s_abs = 0.0;
  for (int i0 = 1; i0 < c.imt - 1; i0 += 64) {   // i = 2..imt-1
    const int i = i0 + lane;
    const bool in = i < c.imt - 1;
    const double dx = in ? c.dxt[i] : 0.0;
    for (int r = 0; r < TSI_ROWS; ++r) {
      const long long bt = base_t[r];
      double a3 = 0.0, a1 = 0.0, a2 = 0.0;
      if (bt >= 0 && in) {
        const double a = c.t_tau[bt + i], w = f_w[r] * dx * t3[r][64] * c.tmask[base_m[r] + i];
        a3 = a * w; a1 = a * a * w; a2 = dabs(c.t_taup1[bt + i] - c.t_taum1[bt + i]) * w * f_x[r];
      }
      t3[r][lane] = a3; t1[r][lane] = a1; t2[r][lane] = a2;
    }
    __syncthreads();
    if (lane < TSI_ROWS) {
      const int cnt = c.imt - 1 - i0 < 64 ? c.imt - 1 - i0 : 64;
      for (int q = 0; q < cnt; ++q) { s_bar = s_bar + t3[lane][q]; s_var = s_var + t1[lane][q]; s_abs = s_abs + t2[lane][q]; }
    }
    __syncthreads();
  }
  if (lane < TSI_ROWS && g0 + lane < total) {
    const size_t q = (size_t)k + (size_t)(c.km + 1) * ((size_t)(n - 1) + (size_t)c.nt * (j - 1)), NA = (size_t)(c.km + 1) * c.nt * c.jmt;
    acc[q] = s_bar; acc[NA + q] = s_var; acc[2 * NA + q] = s_abs;
  }
}
// delta 14C of t(tau+1) (tracer.F:1329-1353): the reference keeps one running sum over the whole grid; here one partial sum
// per row (k, j) -- a wave per row, lanes over the columns, a fixed tree over the lanes -- which uvic_gpu_tsi_read adds up
// in the order of the rows (agreement with the reference to rounding, run to run to the bit).
__global__ void __launch_bounds__(256) k_tsi_dc14(const uvic_ctx c, int ic14, int idic, double rc14std, double *rows) {
  const int gid = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
  const int nrows = c.je - c.js + 1;
  if (gid >= c.km * nrows) return;
  const int k = gid % c.km + 1, j = c.js + gid / c.km;
  const size_t N3 = (size_t)c.imt * c.km * c.jmt;
  const double rrc14std = 1000. / rc14std, fyz = c.cst[j - 1] * c.dyt[j - 1] * c.dzt[k - 1];
  const size_t row = (size_t)c.imt * ((size_t)(k - 1) + (size_t)c.km * (j - 1));
  const double *t14 = c.t_taup1 + (size_t)(ic14 - 1) * N3 + row, *tdic = c.t_taup1 + (size_t)(idic - 1) * N3 + row, *msk = c.tmask + row;
  double sum = 0.0;
  for (int i = 1 + lane; i < c.imt - 1; i += 64) {
    const double dc14 = (rrc14std * t14[i] / (tdic[i] + UV_EPSLN) - 1000.) * msk[i];
    sum = sum + dc14 * c.dxt[i] * fyz * msk[i];
  }
  for (int off = 32; off > 0; off >>= 1) sum = sum + __shfl_down(sum, off, 64);
  if (lane == 0) rows[(size_t)(k - 1) + (size_t)c.km * (j - 1)] = sum;
}
// Kinetic energy of the time-step monitor (clinic.F:616-630), one sum per row (k, j) over both components and the columns in
// the reference's order; eight rows per wave through an LDS tile, as k_tsi_rows.
__global__ void __launch_bounds__(64) k_tsi_ektot(const uvic_ctx c, const double *u1, const double *u2, double rho0, double *out) {
  __shared__ double tile[TSI_ROWS][65];
  const int lane = threadIdx.x;
  const int nrows = c.je - c.js + 1, total = c.km * nrows, g0 = blockIdx.x * TSI_ROWS;
  const int gid = g0 + (lane < TSI_ROWS ? lane : 0);
  const int k = gid % c.km + 1, j = c.js + gid / c.km;    // (the row lane `lane` sums, if it is one)
  double sum = 0.0;
  for (int n = 0; n < 2; ++n) {
    const double *u = n ? u2 : u1;
    for (int i0 = 1; i0 < c.imt - 1; i0 += 64) {   // i = 2..imt-1
      const int i = i0 + lane;
      for (int r = 0; r < TSI_ROWS; ++r) {
        const int g = g0 + r;
        double v = 0.0;
        if (g < total && i < c.imt - 1) {
          const int kr = g % c.km + 1, jr = c.js + g / c.km;
          const double fx = rho0 * 0.5 * c.csu[jr - 1] * c.dyu[jr - 1], fxz = fx * c.dzt[kr - 1];
          const double x = u[(size_t)c.imt * ((size_t)(kr - 1) + (size_t)c.km * (jr - 1)) + i];
          v = x * x * (fxz * c.dxu[i]);
        }
        tile[r][lane] = v;
      }
      __syncthreads();
      if (lane < TSI_ROWS) {
        const int cnt = c.imt - 1 - i0 < 64 ? c.imt - 1 - i0 : 64;
        for (int q = 0; q < cnt; ++q) sum = sum + tile[lane][q];
      }
      __syncthreads();
    }
  }
  if (lane < TSI_ROWS && g0 + lane < total) out[(size_t)k + (size_t)(c.km + 1) * (j - 1)] = sum;
}
// Pass B runs over the ocean columns only; t(tau+1) is zero on land (the update is masked, tracer.F:1109-1130).  This
// kernel clears the land columns of rows js..je of every local tracer (and the cyclic images of land columns 2 and
// imt-1) ONCE per buffer: nothing on the device writes there afterwards, see land_clean().
__global__ void __launch_bounds__(256) k_land_zero(const uvic_ctx c, double *tp) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nrows = c.je - c.js + 1;
  const long long per = (long long)c.imt * c.km * nrows;
  if (gid >= per * c.nt_local) return;
  const int n = c.n0 + (int)(gid / per);
  const long long q = gid % per;
  const int i = (int)(q % c.imt) + 1, k = (int)((q / c.imt) % c.km) + 1, j = c.js + (int)(q / ((long long)c.imt * c.km));
  const int isrc = (i == 1) ? c.imt - 1 : ((i == c.imt) ? 2 : i);
  if (c.kmt[(size_t)(isrc - 1) + (size_t)c.imt * (j - 1)] > 0) return;
  tp[(size_t)n * c.imt * c.km * c.jmt + (size_t)(i - 1) + (size_t)c.imt * ((size_t)(k - 1) + (size_t)c.km * (j - 1))] = 0.0;
}

// latitude-slab halo: `nrow` rows j0.. of every tracer of `t` <-> a staging buffer laid out (nt, nrow, imt*km);
// dir 0 packs (t -> staging), 1 unpacks (staging -> t).  One launch moves what the exchange sends or receives on a side.
__global__ void __launch_bounds__(256) k_halo_rows(double *t, double *stage, int rowlen, int jmt, int nt, int j0, int nrow, int dir) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long per = (long long)nrow * rowlen;
  if (gid >= per * nt) return;
  const int n = (int)(gid / per);
  const long long q = gid % per;   // (row - j0) * rowlen + offset within the row: rows are adjacent in t as well
  double *cell = t + ((long long)n * jmt + (j0 - 1)) * rowlen + q;
  if (dir == 0) stage[gid] = *cell; else *cell = stage[gid];
}

// -- direct push between the ranks of a node (uvic_gpu_push_*).  Every rank owns a receive window and a
// row of arrival counters in uncached device memory; its peers map both through hipIpc and write them with their own
// kernels, so the data crosses xGMI once, on the link between the two ranks, with no collective library in the path.
#define UVIC_PUSH_MAX 64
struct PushTargets {
  int n;
  double *dst[8];
  unsigned long long *flag[8];
};
// the rank's contiguous slice of t(tau+1), to up to 8 peers at once (blockIdx.y = peer: every link is driven together)
__global__ void __launch_bounds__(256) k_push_slice(const double2 *src, const PushTargets tg, long long n2) {
  double2 *dst = (double2 *)tg.dst[blockIdx.y];
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n2; q += (long long)gridDim.x * blockDim.x) dst[q] = src[q];
}
// ... a kernel boundary later (the payload has left the chip), the peers' arrival counters
__global__ void k_push_raise(const PushTargets tg, unsigned long long seq) {
  if ((int)threadIdx.x < tg.n) __hip_atomic_store(tg.flag[threadIdx.x], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// wait until the counters named by `mask` (bit r: rank r or side r) have reached `seq`; gives up after `ticks` of the
// 100 MHz wall clock and says so in *err, so that a lost peer ends in an error at the next sync, not in a hung GPU
__global__ void k_push_wait(const unsigned long long *flags, unsigned long long mask, unsigned long long seq, long long ticks, int *err) {
  const int r = threadIdx.x;
  if (!((mask >> r) & 1ull)) return;
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(flags + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
    if (wall_clock64() - t0 > ticks) { *err = r + 1; return; }
    __builtin_amdgcn_s_sleep(32);
  }
}
// the slices of the other ranks, window -> t(tau+1) (same layout: slice r at r * per)
__global__ void __launch_bounds__(256) k_push_take(double2 *t, const double2 *win, long long per2, int world, int rank) {
  const long long n2 = per2 * (world - 1);
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n2; q += (long long)gridDim.x * blockDim.x) {
    const long long g = q < per2 * rank ? q : q + per2;
    t[g] = win[g];
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(const char *what, hipError_t e, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s: %s (uvic_gpu.hip:%d)", what, hipGetErrorString(e), line);
  g_err = buf;
  return 1;
}
static int fail_msg(const std::string &m) {
  g_err = m;
  return 2;
}
#define HIPCHK(x)                                   \
  do {                                              \
    hipError_t _e = (x);                            \
    if (_e != hipSuccess) return fail(#x, _e, __LINE__); \
  } while (0)

enum Kind { K_IMT, K_JMT, K_KM, K_KMP1, K_KM9, K_NT, K_S, K_C, K_F };
struct FieldDesc {
  const char *name;
  Kind kind;
  int extra;  // 1, 2, 4, or -1 = nt, -2 = nsrc, -3 = km
  bool is_int;
};
static const FieldDesc FIELDS[UVIC_F_COUNT] = {
    {"dxt", K_IMT, 1, false}, {"dxtr", K_IMT, 1, false}, {"dxu", K_IMT, 1, false}, {"dxur", K_IMT, 1, false},
    {"dxt4r", K_IMT, 1, false},
    {"dyt", K_JMT, 1, false}, {"dytr", K_JMT, 1, false}, {"dyu", K_JMT, 1, false}, {"dyur", K_JMT, 1, false},
    {"dyt4r", K_JMT, 1, false},
    {"cst", K_JMT, 1, false}, {"cstr", K_JMT, 1, false}, {"csu", K_JMT, 1, false}, {"cstdytr", K_JMT, 1, false},
    {"cstdyt2r", K_JMT, 1, false}, {"csu_dyur", K_JMT, 1, false},
    {"dzt", K_KM, 1, false}, {"dztr", K_KM, 1, false}, {"dzt2r", K_KM, 1, false}, {"dztur", K_KM, 1, false},
    {"dztlr", K_KM, 1, false},
    {"dzw", K_KMP1, 1, false}, {"dzwr", K_KMP1, 1, false},
    {"dtxcel", K_KM, 1, false}, {"dtxsqr", K_KM, 1, false}, {"dztxcl", K_KM, 1, false},
    {"to", K_KM, 1, false}, {"so", K_KM, 1, false}, {"c", K_KM9, 1, false},
    {"kmt", K_S, 1, true},
    {"fisop", K_S, -3, false},
    {"addisop", K_C, 1, false},
    {"t_taum1", K_C, -1, false}, {"t_tau", K_C, -1, false}, {"t_taup1", K_C, -1, false},
    {"adv_vet", K_C, 1, false}, {"adv_vnt", K_C, 1, false}, {"adv_vbt", K_F, 1, false},
    {"diff_cbt_bg", K_C, 1, false},
    {"stf", K_S, -1, false}, {"btf", K_S, -1, false},
    {"src", K_C, -2, false},
    {"itrc", K_NT, 1, true},
    {"alphai", K_C, 1, false}, {"betai", K_C, 1, false}, {"ddxt", K_C, 2, false}, {"ddyt", K_C, 2, false},
    {"ddzt", K_F, 2, false},
    {"Ai_ez", K_C, 4, false}, {"Ai_nz", K_C, 4, false}, {"Ai_bx", K_C, 4, false}, {"Ai_by", K_C, 4, false},
    {"K11", K_C, 1, false}, {"K22", K_C, 1, false}, {"K33", K_C, 1, false},
    {"adv_vetiso", K_C, 1, false}, {"adv_vntiso", K_C, 1, false}, {"adv_vbtiso", K_F, 1, false},
    {"diff_cbt", K_C, 1, false},
    {"u1", K_C, 1, false}, {"u2", K_C, 1, false}, {"dxt2r", K_IMT, 1, false}, {"dyt2r", K_JMT, 1, false},
    {"zw", K_KM, 1, false}, {"tlat", K_S, 1, false},
    {"edrm2", K_C, 1, false}, {"edrs2", K_C, 1, false}, {"edrk1", K_C, 1, false}, {"edro1", K_C, 1, false},
    {"rho", K_C, 1, false}, {"um1", K_C, 1, false}, {"um2", K_C, 1, false}, {"up1", K_C, 1, false}, {"up2", K_C, 1, false},
    {"zu", K_S, 2, false}, {"grad_p", K_C, 2, false}, {"smf", K_S, 2, false}, {"kmu", K_S, 1, true}, {"hr", K_S, 1, false},
    {"cori", K_S, 2, false}, {"visc_ceu", K_C, 1, false}, {"amc_north", K_C, 1, false}, {"amc_south", K_C, 1, false},
    {"dxu2r", K_IMT, 1, false}, {"dxmetr", K_IMT, 1, false}, {"duw", K_IMT, 1, false}, {"due", K_IMT, 1, false},
    {"dyu2r", K_JMT, 1, false}, {"dyu4r", K_JMT, 1, false}, {"csur", K_JMT, 1, false}, {"dus", K_JMT, 1, false},
    {"dun", K_JMT, 1, false}, {"csudyu2r", K_JMT, 1, false}, {"advmet", K_JMT, 2, false}, {"am3", K_JMT, 1, false},
    {"am4", K_JMT, 2, false},
    {"sbc_gu", K_S, 1, false}, {"sbc_gv", K_S, 1, false}, {"sbc_su", K_S, 1, false}, {"sbc_sv", K_S, 1, false},
    {"spsin", K_IMT, 1, false}, {"spcos", K_IMT, 1, false}, {"phi", K_JMT, 1, false},
    {"psi", K_S, 2, false},
};

struct KernelStat {
  const char *name;
  double ms;
  int calls;
};

struct uvic_gpu {
  uvic_dims d;
  int device;
  hipStream_t stream;
  void *buf[UVIC_F_COUNT];
  double *work[8];  // tot_e, tot_n, tot_b, adv_x (also S of the column path), adv_z, RpY, RmY
  int *cv_int[3];   // convection segments: nseg, kt, kb
  int *cv_int2[3];  // the second set of the segment arrays and of cv_z (see cv_lists)
  double *cv_z2;
  int *cv_list;     // [0] number of columns the T,S walk of this step mixed, [1..] their ids (k_convect_list)
  int *cv_lists[2]; // ... two of them, taken in turn (uvic_gpu_rotate): the next step's T,S passes may begin before this
                    // step's k_convect_apply_list has read its list (the resident overlay's T,S chain, launch_transport)
  double *cv_z;
  bool exact_convect;  // single-kernel convct2 (debug: UVIC_CONVECT_ONEPASS=1)
  double *coef;     // folded isopycnal coefficients, CF_PAIRS pair planes (kernels_col.hpp)
  double *fny;      // column kernels: half of the final limited flux through the north face of every cell, per tracer
  bool exact;       // bit-exact row kernels (kernels_fct.hpp) instead of the column path
  bool ts_rows;     // ... through the row kernels of kernels_fct.hpp instead of the exact column kernels (cross-check: set_exact(3))
  bool ts_exact;    // production default: T and S (whose bits decide every convective adjustment) go through the bit-exact kernels, the other tracers through the column kernels
  bool mixing_next_guard = false;
  // ocean columns, row by row (WetCols): device list, and where each row starts in it (host, size jmt+2)
  int *wet_dev;
  std::vector<int> wet_row_start;
  // lane maps of the column passes (ColGrid) for the current slab, rebuilt when kmt or the slab changes
  std::vector<int> kmt_host;
  int *lanes_dev;            // pass A map (nwaves_a * 64 codes) followed by the pass B map
  int nwaves_a, nwaves_b;
  bool lanes_dirty;
  // t buffers whose land columns (rows js..je) are known to hold zeros: pass B does not touch land, see land_clean()
  std::vector<void *> land_zeroed;
  // latitude-slab halo staging (uvic_gpu_halo_*): send south/north, receive south/north, each UVIC_HALO rows of every tracer
  double *halo[4];
  // direct push (uvic_gpu_push_*): the receive window [2 parities][slots][slot_elems], the arrival counters [slots],
  // and what the peers have let this rank map of theirs
  struct {
    int world = 0, rank = 0, mode = 0;
    double *window = nullptr;
    unsigned long long *flags = nullptr;
    size_t slot_elems = 0;
    int slots = 0;
    double *peer_window[UVIC_PUSH_MAX] = {};
    unsigned long long *peer_flags[UVIC_PUSH_MAX] = {};
    bool mapped[UVIC_PUSH_MAX] = {};     // opened through hipIpc (to be closed), as opposed to the rank's own
    unsigned long long seq = 0;
    int *err = nullptr;                  // pinned host word the wait kernel writes when it gives up
    double wait_ms = 2000.0;
  } push;
  // source buffers known to hold zeros on land (MOBI writes ocean columns only): see src_clean()
  std::vector<void *> src_zeroed;
  uvic_ctx ctx;
  mobi_dev mobi;
  mobi_store mobi_st;
  uvic_mobi_options opt_staged;   // uvic_gpu_mobi_options_flat: taken by the next uvic_gpu_set_mobi_flat
  bool have_opt_staged;
  int mobi_key;   // n15 | c13<<1 | caco3<<2 | silicon<<3 of the set bound by uvic_gpu_set_mobi_opt
  bool have_mobi;
  bool have_vmix;   // uvic_gpu_set_vmix_params was called
  double *vmix_tab = nullptr;   // (km*km + km): uvic_ctx.vmix_e, vmix_d
  bool vmix_tab_ready = false;
  std::vector<void *> pinned;   // host ranges page-locked through uvic_gpu_pin_host
  hipStream_t side_mom;     // uvic_gpu_momentum_async: state + clinic beside the tracer step
  hipEvent_t ev_mom_in, ev_mom_done;
  bool mom_pending;
  int tmm_ncols;    // > 0: a column-batch handle (uvic_gpu_tmm_create); the batch is row 2, columns 2..ncols+1
  // baroclinic momentum step (uvic_gpu_state / uvic_gpu_clinic)
  uvic_clinic_params clinic_p;
  bool have_clinic;
  FilterItem *fltu_items;   // polar filter of the velocities (uvic_gpu_set_filter_u): strips, operators, filtered rows
  double *fltu_mats;
  int *fltu_rows;
  int fltu_nitems, fltu_threads, fltu_nrows;
  // polar filter (uvic_gpu_set_filter): strips and operators, built once
  FilterItem *flt_items;
  double *flt_mats;
  int flt_nitems, flt_threads;
  double mobi_dtnpzd;
  bool mobi_team;   // four-wave team kernel (default) or one thread per column (set_option "mobi_team" 0)
  bool mobi_generic;   // set_option "mobi_generic": option set C through the general column kernel as well (cross-check)
  // one-step-ahead source terms on side streams (uvic_gpu_prefetch_sources): two of them, taken in turn, so that
  // the MOBI chain of step n+2 (pre -> team -> post) may start while that of step n+1 is still running
  hipStream_t side_m[2];
  int mobi_flip;            // which of the two the next prefetch uses
  bool mobi_two_streams;    // UVIC_MOBI_STREAMS=1 keeps every chain on the first (measurement)
  // events of the look-ahead chains alternate, because the chain of step n+1 is queued (and records its event)
  // before step n waits for the chain of step n: `_pending` = recorded during this step, `_ready` = what this step waits for
  int ev_flip;
  hipEvent_t ev_src_ready, ev_src_pending;
  hipEvent_t ev_step_end[2], ev_end_ready, ev_end_pending;   // end of a step's own work (step_end)
  bool end_ready, end_pending;
  // isopyc one step ahead on a second side stream (uvic_gpu_prefetch_isopyc): alternate set of its products
  hipStream_t side2;
  // (T and S finish pass B and the convective T,S walk on side2 as well, beside pass B of the other tracers: the
  // device offers four hardware queues, and a fifth stream would share one with another and wait behind its barriers)
  bool iso_waited;    // this step's T,S-derived fields came from a look-ahead chain (iso_set[iso_cur].ev)
  bool unmix_at_rotate;   // uvic_gpu_step_lookahead ran a forward step: uvic_gpu_rotate ends the aliasing
  bool step_begun;    // ev_step_begin of the current step is recorded (uvic_gpu_rotate ends the step)
  hipStream_t side_ts; // the T,S passes: an alias of side2
  hipEvent_t ev_fct_done, ev_ts_done;
  bool ts_ahead;      // this step's convect_ts was already issued on side2
  // resident overlay (uvic_gpu_overlay_step): uploads without host synchronisation, T,S of t(tau+1) sent to the host as soon
  // as they are final, the surface boundary condition sums of set_sbc kept on the device
  bool host_sync;               // 0: uploads are queued on the main stream and not waited for
  double *ts_host;              // where T,S of this step's t(tau+1) go (null: nowhere); set for one step by overlay_step
  hipEvent_t ev_ts_host;
  bool ts_host_queued;
  // the step's inputs from the host (uvic_gpu_overlay_inputs): three device copies of the velocities and fluxes, filled in
  // turn by copies on streams of their own (the caller has T,S of step n back while the other tracers of step n-1 may
  // still be in their pass B: the copy step n+1 fills is that of step n-2); a step waits for the group of copies it
  // needs, where it needs it
  struct {
    double *set[3][5] = {};       // adv_vet, adv_vnt, adv_vbt, stf, btf
    int cur = 0;
    bool used = true;             // a step has been queued on the current set
    hipStream_t st[2] = {nullptr, nullptr};
    int nstreams = 2;
    hipEvent_t ev_first = nullptr, ev_rest = nullptr, ev_link[2] = {nullptr, nullptr};
    bool first_pending = false, rest_pending = false;   // queued, and the main stream has not been told to wait for them yet
    bool waited = false;          // this step's main stream began with a wait for them (the T,S stream starts behind it)
    bool derive_vbt = false;      // adv_vbt was not sent: formed from adv_vet, adv_vnt on the device (adv_vel.F:98-127)
    // MOBI's light, ice and snow fields of the segment (uvic_gpu_set_mobi_step): fetched from the caller's page-locked
    // arrays by the first MOBI chain that follows (forcing_pull), which the other chains then wait for (forcing_sent)
    hipEvent_t ev_forcing = nullptr;
    const double *forcing_src[4] = {};
    bool forcing_pull = false, forcing_sent = false, forcing_inflight = false;
    bool rest_inflight = false;   // copies the caller's arrays must outlast: overlay_step returns behind them
    // the velocities of the coming step formed on the device from u (uvic_gpu_overlay_momentum) into the copy the next
    // uvic_gpu_overlay_inputs then completes; the step's streams start behind ev_vel
    hipEvent_t ev_vel = nullptr;
    bool vel_pending = false;
    double *ektot_dev = nullptr;
  } in;
  // the event behind which the MOBI sources of a step are complete (its chain has read t(tau-1), T and S included), by step
  // parity: the T,S passes of the NEXT step write that buffer (launch_transport)
  hipEvent_t ev_mobi_of[2];
  long long mobi_ev_step[2];
  hipEvent_t ev_src_inline2[2];   // sources of the current step computed on a MOBI side stream (launch_mobi), by step parity
  long long end_step_of[2];       // the step that recorded ev_step_end[q]
  bool close_step = false;        // uvic_gpu_set_option "close_step"
  bool prep_deferred;             // launch_isopyc left k_inputs_cell (and the wait for the inputs) to launch_transport
  int sbc_count;                // tracers whose surface level is accumulated
  int *sbc_tracer;              // device: their 1-based tracer numbers
  double *sbc_acc;              // device (imt, jmt, sbc_count)
  // time-step integrals (O_time_step_monitor) formed on the device on the steps the caller names (uvic_gpu_set_tsi)
  bool tsi_step;                // this step
  int tsi_ic14, tsi_idic;       // tracer numbers of 14C and DIC (0: no delta-14C sum)
  double *tsi_acc;              // tbar, travar, dtabs, each (0:km, nt, jmt); then (km, jmt) row sums of delta 14C
  // time-average steps (uvic_gpu_set_tavg): convection diagnostics (imt, jmt, 3), the delta-14C field, depths of the T points
  bool tavg_step = false;
  double tavg_grav = 0.0;
  double *tavg_zt = nullptr, *tavg_diag = nullptr, *tavg_dc14 = nullptr;
  int tavg_ic14 = 0, tavg_idic = 0;
  double *tsi_host = nullptr;   // the same, page-locked: filled behind every time-step-monitor step
  hipEvent_t ev_tsi = nullptr;
  bool tsi_inflight = false;
  // what the look-ahead MOBI chain assumed about the step it computed for (checked when that step starts)
  double src_relyr, src_co2ccn, src_c2dtts;
  bool serial;        // uvic_gpu_profile: everything on the main stream, one kernel after the other
  bool ts_no_src;     // itrc(1) = itrc(2) = 0: T and S have no source term (known from the upload of itrc)
  // The T,S-derived fields (mixing tensor, GM velocities, folded coefficients, diff_cbt) exist in three sets: step m
  // uses set m % 3, so the chain of step m+1 or m+2 can be written while step m still reads its own.  h->buf[], work[0..2]
  // and coef are views of the set in use (use_iso_set).
  struct IsoSet {
    void *f[16];
    double *work[3], *coef;
    hipEvent_t ev;          // recorded behind the chain that filled the set
    hipStream_t st;         // ... on this stream
    long long for_step;     // the step whose fields a look-ahead chain put there, -1: none
    bool vel_stale;         // adv_vet/vnt/vbt were uploaded after the chain had formed the total velocities from them
    bool allocated;
  } iso_set[3];
  int iso_cur;
  long long step_no;        // counts uvic_gpu_rotate
  hipEvent_t ev_ts_final;   // T and S of this step's t(tau+1) are final (after convection and the polar filter)
  bool ts_final_valid;
  hipEvent_t ev_ts_filt = nullptr;   // ... filtered on the T,S stream, behind their walk
  bool ts_filtered = false;          // launch_convect's filter then leaves T and S out
  hipEvent_t ev_step_begin, ev_src_next[2];
  hipEvent_t ev_begin_cur;      // the event that stands for this step's begin: ev_step_begin, or the previous step's end event
  bool idle_until_next;         // the caller has promised that nothing follows the step on the main stream before the next one
  long long ts_waited_begin;    // the step whose begin event the T,S stream has waited for already
  void *src_alt;
  bool prefetch_pending, src_from_prefetch, mixing;
  int nchunk, fct_threads, upd_threads;
  size_t fct_lds, upd_lds;
  // profiling
  bool profiling;
  std::vector<hipEvent_t> ev[5];            // [0] main stream, [1] MOBI, [2] isopyc, [3] T,S side streams, [4] MOBI (odd steps)
  std::vector<const char *> ev_names[5];
  std::vector<hipEvent_t> ev_pool;
};

static int64_t plane(const uvic_dims &d, Kind k) {
  switch (k) {
    case K_IMT: return d.imt;
    case K_JMT: return d.jmt;
    case K_KM: return d.km;
    case K_KMP1: return d.km + 1;
    case K_KM9: return (int64_t)d.km * 9;
    case K_NT: return d.nt;
    case K_S: return (int64_t)d.imt * d.jmt;
    case K_C: return (int64_t)d.imt * d.km * d.jmt;
    case K_F: return (int64_t)d.imt * (d.km + 1) * d.jmt;
  }
  return 0;
}
static int64_t extra_of(const uvic_dims &d, int e) {
  if (e == -1) return d.nt;
  if (e == -2) return d.nsrc > 0 ? d.nsrc : 1;
  if (e == -3) return d.km;
  return e;
}
static int64_t field_elems(const uvic_dims &d, int f) { return plane(d, FIELDS[f].kind) * extra_of(d, FIELDS[f].extra); }
static size_t elem_size(int f) { return FIELDS[f].is_int ? 4 : 8; }

extern "C" const char *uvic_gpu_last_error(void) { return g_err.c_str(); }
extern "C" int uvic_gpu_abi_version(void) { return 11; }   // 11: uvic_gpu_set_tavg/_tavg_read, uvic_gpu_overlay_inputs, uvic_gpu_overlay_velocities, uvic_gpu_overlay_momentum, uvic_gpu_push_*; 10: uvic_gpu_set_option, set_exact modes 2 and 3 (T and S bit-exact by default); 9: uvic_gpu_unpin_host; 8: uvic_gpu_momentum_async/_wait; 7: uvic_gpu_rotate_u, uvic_gpu_add_ext_mode; 6: uvic_gpu_tmm_*; 5: uvic_gpu_state, uvic_gpu_clinic

static void bind_ctx(uvic_gpu *h) {
  uvic_ctx &c = h->ctx;
  const uvic_dims &d = h->d;
  c.imt = d.imt; c.jmt = d.jmt; c.km = d.km; c.nt = d.nt; c.nsrc = d.nsrc;
#define B(field, F) c.field = (decltype(c.field))h->buf[F]
  B(dxt, UVIC_F_DXT); B(dxtr, UVIC_F_DXTR); B(dxu, UVIC_F_DXU); B(dxur, UVIC_F_DXUR); B(dxt4r, UVIC_F_DXT4R);
  B(dyt, UVIC_F_DYT); B(dytr, UVIC_F_DYTR); B(dyu, UVIC_F_DYU); B(dyur, UVIC_F_DYUR); B(dyt4r, UVIC_F_DYT4R);
  B(cst, UVIC_F_CST); B(cstr, UVIC_F_CSTR); B(csu, UVIC_F_CSU); B(cstdytr, UVIC_F_CSTDYTR);
  B(cstdyt2r, UVIC_F_CSTDYT2R); B(csu_dyur, UVIC_F_CSU_DYUR);
  B(dzt, UVIC_F_DZT); B(dztr, UVIC_F_DZTR); B(dzt2r, UVIC_F_DZT2R); B(dztur, UVIC_F_DZTUR); B(dztlr, UVIC_F_DZTLR);
  B(dzw, UVIC_F_DZW); B(dzwr, UVIC_F_DZWR); B(dtxcel, UVIC_F_DTXCEL); B(dtxsqr, UVIC_F_DTXSQR);
  B(dztxcl, UVIC_F_DZTXCL); B(to, UVIC_F_TO); B(so, UVIC_F_SO); B(c, UVIC_F_C);
  B(kmt, UVIC_F_KMT); B(fisop, UVIC_F_FISOP); B(addisop, UVIC_F_ADDISOP);
  B(t_taum1, UVIC_F_T_TAUM1); B(t_tau, UVIC_F_T_TAU); B(t_taup1, UVIC_F_T_TAUP1);
  B(adv_vet, UVIC_F_ADV_VET); B(adv_vnt, UVIC_F_ADV_VNT); B(adv_vbt, UVIC_F_ADV_VBT);
  B(diff_cbt_bg, UVIC_F_DIFF_CBT_BG); B(diff_cbt, UVIC_F_DIFF_CBT);
  B(stf, UVIC_F_STF); B(btf, UVIC_F_BTF); B(itrc, UVIC_F_ITRC);
  c.src = d.nsrc > 0 ? (const double *)h->buf[UVIC_F_SRC] : nullptr;
  B(alphai, UVIC_F_ALPHAI); B(betai, UVIC_F_BETAI); B(ddxt, UVIC_F_DDXT); B(ddyt, UVIC_F_DDYT); B(ddzt, UVIC_F_DDZT);
  B(Ai_ez, UVIC_F_AI_EZ); B(Ai_nz, UVIC_F_AI_NZ); B(Ai_bx, UVIC_F_AI_BX); B(Ai_by, UVIC_F_AI_BY);
  B(K11, UVIC_F_K11); B(K22, UVIC_F_K22); B(K33, UVIC_F_K33);
  B(adv_vetiso, UVIC_F_ADV_VETISO); B(adv_vntiso, UVIC_F_ADV_VNTISO); B(adv_vbtiso, UVIC_F_ADV_VBTISO);
  B(u1, UVIC_F_U1); B(u2, UVIC_F_U2); B(dxt2r, UVIC_F_DXT2R); B(dyt2r, UVIC_F_DYT2R); B(zw, UVIC_F_ZW); B(tlat, UVIC_F_TLAT);
  B(edrm2, UVIC_F_EDRM2); B(edrs2, UVIC_F_EDRS2); B(edrk1, UVIC_F_EDRK1); B(edro1, UVIC_F_EDRO1);
#undef B
  if (h->mixing) c.t_taum1 = c.t_tau;  // forward step: both slots hold tau (updates/09/source/mom/loadmw.F:107-111)
  c.tot_e = h->work[0]; c.tot_n = h->work[1]; c.tot_b = h->work[2];
  c.adv_x = h->work[3]; c.adv_z = h->work[4]; c.RpY = h->work[5]; c.RmY = h->work[6];
  c.fny = h->fny;
}

// the t(:,:,:,:,-1:1) slots rotate by pointer; tmask is derived from kmt on upload
// -- the three sets of T,S-derived fields ---------------------------------------------
static const int ISO_FIELDS[16] = {UVIC_F_ALPHAI, UVIC_F_BETAI, UVIC_F_DDXT, UVIC_F_DDYT, UVIC_F_DDZT, UVIC_F_AI_EZ, UVIC_F_AI_NZ,
                                   UVIC_F_AI_BX, UVIC_F_AI_BY, UVIC_F_K11, UVIC_F_K22, UVIC_F_K33, UVIC_F_ADV_VETISO,
                                   UVIC_F_ADV_VNTISO, UVIC_F_ADV_VBTISO, UVIC_F_DIFF_CBT};
static const int IN_FIELDS[5] = {UVIC_F_ADV_VET, UVIC_F_ADV_VNT, UVIC_F_ADV_VBT, UVIC_F_STF, UVIC_F_BTF};
static int make_tmask(uvic_gpu *h);
static void iso_set_adopt(uvic_gpu *h);
static void iso_set_release(uvic_gpu *h);
static int use_iso_set(uvic_gpu *h, int s);

// tile geometry of the row kernels (kernels_fct.hpp): the fewest longitude chunks whose tile fits the LDS budget, or `nchunk_forced`
static int set_tile_geometry(uvic_gpu *h, int nchunk_forced) {
  const uvic_dims *dims = &h->d;
  int budget_kb = 150;
  if (const char *e = uv_env("UVIC_LDS_BUDGET_KB")) budget_kb = atoi(e);
  h->fct_threads = 1024;
  if (const char *e = uv_env("UVIC_FCT_THREADS")) h->fct_threads = atoi(e);
  h->upd_threads = 512;
  if (const char *e = uv_env("UVIC_UPD_THREADS")) h->upd_threads = atoi(e);
  int nchunk = 1;
  for (;; ++nchunk) {
    const int per = (dims->imt - 2 + nchunk - 1) / nchunk;
    const int W = per + 4;
    const size_t need = ((size_t)W * dims->km * 6 + (size_t)W * (dims->km + 1) * 2) * 8;
    if (need <= (size_t)budget_kb * 1024 || per <= 8) break;
  }
  if (nchunk_forced > 0) nchunk = nchunk_forced;
  h->nchunk = nchunk;
  const int per = (dims->imt - 2 + nchunk - 1) / nchunk;
  const int W = per + 4;
  h->fct_lds = ((size_t)W * dims->km * 6 + (size_t)W * (dims->km + 1) * 2) * 8;
  h->upd_lds = ((size_t)W * dims->km * 3 + (size_t)W * (dims->km + 1) * 2) * 8;
  if (h->fct_lds > 160 * 1024) return fail_msg("row kernels: the tile does not fit LDS with this many longitude chunks");
  HIPCHK(hipFuncSetAttribute((const void *)k_fct_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->fct_lds));
  HIPCHK(hipFuncSetAttribute((const void *)k_update_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->upd_lds));
  return 0;
}
extern "C" int uvic_gpu_sync(uvic_gpu *h);
// cross-check and tuning switches for tests and tools (the defaults are what the library is measured with)
extern "C" int uvic_gpu_set_option(uvic_gpu *h, const char *name, int value) {
  if (!h || !name) return fail_msg("uvic_gpu_set_option: null argument");
  if (int rc = uvic_gpu_sync(h)) return rc;
  const std::string n(name);
  if (n == "nchunk") return set_tile_geometry(h, value);           // longitude chunks of the row kernels (0: automatic)
  if (n == "mobi_generic") { h->mobi_generic = value != 0; return 0; }   // option set C through the general MOBI kernel (before set_mobi_opt)
  if (n == "mobi_team") { h->mobi_team = value != 0; return 0; }         // 0: one thread per column instead of four-wave teams
  if (n == "convect_onepass") { h->exact_convect = value != 0; return 0; }   // convct2 as one kernel over all tracers
  if (n == "mobi_streams") { h->mobi_two_streams = value != 1 && h->side_m[1] != h->side_m[0]; return 0; }   // 1: every MOBI chain on the first side stream
  if (n == "close_step") { h->close_step = value != 0; return 0; }   // diagnosis: no chain crosses a step boundary (see uvic_gpu_step_lookahead_at)
  if (n == "upload_streams") {   // copy streams of uvic_gpu_overlay_inputs (before its first call)
    if (h->in.st[0]) return fail_msg("uvic_gpu_set_option: upload_streams after the first uvic_gpu_overlay_inputs");
    h->in.nstreams = value == 1 ? 1 : 2; return 0;
  }
  if (n == "push_wait_ms") { h->push.wait_ms = value > 0 ? value : 2000.0; return 0; }   // how long an exchange waits for a peer before it reports it lost
  return fail_msg("uvic_gpu_set_option: unknown option " + n);
}
// The step runs on four streams and the resident overlay adds two for its copies; the HIP runtime maps streams onto four
// hardware queues unless told otherwise, and a copy stream that shares a queue with a MOBI chain has its event markers
// held up behind a 0.3 ms kernel (0.15 ms per overlay call, measured).  Ask for more queues while the runtime has not
// read its settings yet (its first call in the process: the Fortran driver's case); a value the user has set stands.
static void want_hw_queues() {
  static bool done = false;
  if (done) return;
  done = true;
  (void)setenv("GPU_MAX_HW_QUEUES", "8", 0);
}
extern "C" int uvic_gpu_create(uvic_gpu **out, const uvic_dims *dims, int device) {
  if (!out || !dims) return fail_msg("uvic_gpu_create: null argument");
  want_hw_queues();
  if (dims->imt < 6 || dims->jmt < 6 || dims->km < 2 || dims->nt < 2)
    return fail_msg("uvic_gpu_create: dimensions too small (need imt,jmt >= 6, km >= 2, nt >= 2)");
  if (dims->km > 64) return fail_msg("uvic_gpu_create: km > 64 not supported (per-level metrics are held one per lane of a wave)");
  HIPCHK(hipSetDevice(device));
  uvic_gpu *h = new uvic_gpu();
  h->d = *dims;
  h->device = device;
  h->profiling = false; h->prep_deferred = false;
  h->have_mobi = false;
  h->have_vmix = false;
  h->tmm_ncols = 0;
  h->side_mom = nullptr; h->ev_mom_in = h->ev_mom_done = nullptr; h->mom_pending = false;
  h->have_clinic = false;
  h->fltu_items = nullptr; h->fltu_mats = nullptr; h->fltu_rows = nullptr; h->fltu_nitems = h->fltu_threads = h->fltu_nrows = 0;
  h->flt_items = nullptr; h->flt_mats = nullptr; h->flt_nitems = 0; h->flt_threads = 0;
  h->wet_dev = nullptr;
  h->lanes_dev = nullptr; h->nwaves_a = h->nwaves_b = 0; h->lanes_dirty = true;
  for (int q = 0; q < 4; ++q) h->halo[q] = nullptr;
  h->wet_row_start.assign((size_t)dims->jmt + 2, 0);
  memset(&h->mobi_st, 0, sizeof h->mobi_st);
  memset(&h->opt_staged, 0, sizeof h->opt_staged);
  h->have_opt_staged = false;
  memset(&h->mobi, 0, sizeof h->mobi);
  // The main stream carries the bulk passes (pass A and B of the nt-2 tracers): their waves fill every SIMD for most of
  // a step, and the short kernels of the latency chains on the side streams (isopyc, MOBI pre/post, the T,S passes) then
  // wait a whole wave lifetime for a free slot.  UVIC_MAIN_CUS=N (a multiple of 8, < 256) keeps the bulk passes off
  // 256-N compute units (the mask bits go round the 8 XCDs), which stay free for the side streams.
  {
    int main_cus = 0;
    if (const char *e = uv_env("UVIC_MAIN_CUS")) main_cus = atoi(e);
    if (main_cus >= 8 && main_cus < 256) {
      uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int b = 0; b < main_cus; ++b) mask[b / 32] |= 1u << (b % 32);
      HIPCHK(hipExtStreamCreateWithCUMask(&h->stream, 8, mask));
    } else {
      HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    }
  }
  h->mobi_flip = 0;
  h->mobi_two_streams = true;
  if (const char *e = uv_env("UVIC_MOBI_STREAMS")) h->mobi_two_streams = atoi(e) != 1;
  HIPCHK(hipStreamCreateWithFlags(&h->side_m[0], hipStreamNonBlocking));
  if (h->mobi_two_streams) HIPCHK(hipStreamCreateWithFlags(&h->side_m[1], hipStreamNonBlocking));
  else h->side_m[1] = h->side_m[0];
  HIPCHK(hipStreamCreateWithFlags(&h->side2, hipStreamNonBlocking));
  h->side_ts = h->side2;   // a stream of their own did not pay: the device has four hardware queues
  HIPCHK(hipEventCreateWithFlags(&h->ev_fct_done, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&h->ev_ts_done, hipEventDisableTiming));
  h->host_sync = true; h->ts_host = nullptr; h->ts_host_queued = false;
  HIPCHK(hipEventCreateWithFlags(&h->ev_ts_host, hipEventDisableTiming));
  h->sbc_count = 0; h->sbc_tracer = nullptr; h->sbc_acc = nullptr;
  h->tsi_step = false; h->tsi_acc = nullptr; h->tsi_ic14 = h->tsi_idic = 0;
  h->src_relyr = h->src_co2ccn = 0.0;
  h->ts_ahead = false; h->serial = false; h->ts_no_src = false; h->step_begun = false; h->iso_waited = false; h->unmix_at_rotate = false;
  for (int q = 0; q < 3; ++q) {
    memset(&h->iso_set[q], 0, sizeof h->iso_set[q]);
    HIPCHK(hipEventCreateWithFlags(&h->iso_set[q].ev, hipEventDisableTiming));
    h->iso_set[q].st = nullptr;
    h->iso_set[q].for_step = -1;
  }
  h->iso_cur = 0; h->step_no = 0; h->ts_final_valid = false;
  h->ev_flip = 0;
  HIPCHK(hipEventCreateWithFlags(&h->ev_step_begin, hipEventDisableTiming));
  h->ev_begin_cur = h->ev_step_begin;
  h->idle_until_next = false;
  h->ts_waited_begin = -1;
  for (int q = 0; q < 2; ++q) HIPCHK(hipEventCreateWithFlags(&h->ev_src_next[q], hipEventDisableTiming));
  for (int q = 0; q < 2; ++q) {
    HIPCHK(hipEventCreateWithFlags(&h->ev_src_inline2[q], hipEventDisableTiming));
    h->mobi_ev_step[q] = h->end_step_of[q] = -1; h->ev_mobi_of[q] = nullptr;
  }
  h->ev_src_ready = h->ev_src_pending = h->ev_src_next[0];
  for (int q = 0; q < 2; ++q) HIPCHK(hipEventCreateWithFlags(&h->ev_step_end[q], hipEventDisableTiming));
  h->ev_end_ready = h->ev_end_pending = h->ev_step_end[0];
  h->end_ready = h->end_pending = false;
  h->src_alt = nullptr;
  h->mobi_team = true;
  h->mobi_generic = false;
  if (const char *e = uv_env("UVIC_MOBI_TEAM")) h->mobi_team = atoi(e) != 0;
  h->prefetch_pending = h->src_from_prefetch = h->mixing = false;
  for (int f = 0; f < UVIC_F_COUNT; ++f) {
    const size_t bytes = (size_t)field_elems(h->d, f) * elem_size(f);
    HIPCHK(hipMalloc(&h->buf[f], bytes));
    HIPCHK(hipMemset(h->buf[f], 0, bytes));
  }
  const size_t N3 = (size_t)dims->imt * dims->km * dims->jmt, NF = (size_t)dims->imt * (dims->km + 1) * dims->jmt;
  const size_t wsz[7] = {N3, N3, NF, N3 * dims->nt, N3 * dims->nt, N3 * dims->nt, N3 * dims->nt};
  for (int w = 0; w < 7; ++w) {
    HIPCHK(hipMalloc((void **)&h->work[w], wsz[w] * 8));
    HIPCHK(hipMemset(h->work[w], 0, wsz[w] * 8));
  }
  {
    const size_t NS = (size_t)dims->imt * dims->jmt;
    const size_t isz[3] = {NS, N3, N3};
    for (int q = 0; q < 3; ++q) {
      HIPCHK(hipMalloc((void **)&h->cv_int[q], isz[q] * 4));
      HIPCHK(hipMemset(h->cv_int[q], 0, isz[q] * 4));
      HIPCHK(hipMalloc((void **)&h->cv_int2[q], isz[q] * 4));
      HIPCHK(hipMemset(h->cv_int2[q], 0, isz[q] * 4));
    }
    for (int q = 0; q < 2; ++q) {
      HIPCHK(hipMalloc((void **)&h->cv_lists[q], (NS + 1) * 4));
      HIPCHK(hipMemset(h->cv_lists[q], 0, (NS + 1) * 4));
    }
    h->cv_list = h->cv_lists[0];
    HIPCHK(hipMalloc((void **)&h->cv_z, N3 * 8));
    HIPCHK(hipMemset(h->cv_z, 0, N3 * 8));
    HIPCHK(hipMalloc((void **)&h->cv_z2, N3 * 8));
    HIPCHK(hipMemset(h->cv_z2, 0, N3 * 8));
    h->exact_convect = false;
    if (const char *e = uv_env("UVIC_CONVECT_ONEPASS")) h->exact_convect = atoi(e) != 0;
  }
  HIPCHK(hipMalloc((void **)&h->fny, N3 * 8 * (size_t)dims->nt));
  HIPCHK(hipMemset(h->fny, 0, N3 * 8 * (size_t)dims->nt));
  HIPCHK(hipMalloc((void **)&h->coef, N3 * 16 * CF_PAIRS));
  HIPCHK(hipMemset(h->coef, 0, N3 * 16 * CF_PAIRS));
  iso_set_adopt(h);   // what was just allocated is set 0
  h->exact = false;
  h->ts_exact = true;
  h->ts_rows = false;
  if (const char *e = getenv("UVIC_EXACT")) { h->exact = atoi(e) == 1; h->ts_exact = atoi(e) != 2; }
  // tmask lives in its own buffer (derived data)
  double *tmask;
  HIPCHK(hipMalloc((void **)&tmask, N3 * 8));
  HIPCHK(hipMemset(tmask, 0, N3 * 8));
  memset(&h->ctx, 0, sizeof h->ctx);
  h->ctx.tmask = tmask;
  h->ctx.cv_nseg = h->cv_int[0]; h->ctx.cv_kt = h->cv_int[1]; h->ctx.cv_kb = h->cv_int[2]; h->ctx.cv_z = h->cv_z;
  bind_ctx(h);
  h->ctx.n0 = 0; h->ctx.nt_local = dims->nt; h->ctx.js = 2; h->ctx.je = dims->jmt - 1;
  h->ctx.c2dtts = 0.0; h->ctx.aidif = 0.5;
  h->ctx.no_landskip = uv_env("UVIC_NO_LANDSKIP") ? 1 : 0;
  h->ctx.prio = uv_env("UVIC_TEAM_PRIO0") ? 1 : 0;
  if (const char *e = uv_env("UVIC_SMALL_PRIO")) h->ctx.prio |= atoi(e) ? 4 : 0;
  if (int rc = set_tile_geometry(h, 0)) return rc;
  {
    const size_t colupd_lds = (size_t)2 * (h->d.km + 1) * 64 * 8;
    if (colupd_lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)k_colupd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)colupd_lds));
  }
  // Touch every stream once now: a HIP stream gets its hardware queue when it is first used, and the four of this
  // library must get the device's four queues before anything else in the process (RCCL's own streams, created with the
  // communicator) takes one.  Measured on a 12-row slab: 0.21 ms per step this way, 0.34 ms when the communicator was
  // created first.
  {
    hipStream_t sts[4] = {h->stream, h->side_m[0], h->side_m[1], h->side2};
    for (hipStream_t st : sts) HIPCHK(hipMemsetAsync(tmask, 0, 8, st));
  }
  HIPCHK(hipDeviceSynchronize());
  *out = h;
  return 0;
}

static void push_release(uvic_gpu *h);
extern "C" int uvic_gpu_destroy(uvic_gpu *h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  (void)uvic_gpu_sync(h);   // every stream, the momentum side stream included: it may still be copying into a pinned host range
  iso_set_release(h);   // the three sets of T,S-derived fields; clears their views in buf[], work[0..2], coef
  if (h->in.set[1][0]) for (int q = 0; q < 5; ++q) h->buf[IN_FIELDS[q]] = h->in.set[0][q];   // sets 1 and 2 are freed below
  for (int f = 0; f < UVIC_F_COUNT; ++f) (void)hipFree(h->buf[f]);
  for (int w = 0; w < 7; ++w) (void)hipFree(h->work[w]);
  (void)hipFree((void *)h->ctx.tmask);
  (void)hipFree(h->fny);
  (void)hipFree(h->wet_dev);
  (void)hipFree(h->lanes_dev);
  for (int q = 0; q < 4; ++q) (void)hipFree(h->halo[q]);
  push_release(h);
  if (h->in.st[0]) {
    for (int q = 0; q < 5; ++q) { (void)hipFree(h->in.set[1][q]); (void)hipFree(h->in.set[2][q]); }   // (set[0] are the buffers of buf[], freed above)
    for (int q = 0; q < 2; ++q) { if (h->in.st[q]) (void)hipStreamDestroy(h->in.st[q]); (void)hipEventDestroy(h->in.ev_link[q]); }
    (void)hipEventDestroy(h->in.ev_first); (void)hipEventDestroy(h->in.ev_rest);
  }
  if (h->in.ev_forcing) (void)hipEventDestroy(h->in.ev_forcing);
  if (h->in.ev_vel) (void)hipEventDestroy(h->in.ev_vel);
  (void)hipFree(h->in.ektot_dev);
  (void)hipFree(h->flt_items);
  (void)hipFree(h->flt_mats);
  (void)hipFree(h->fltu_items);
  (void)hipFree(h->fltu_mats);
  (void)hipFree(h->fltu_rows);
  for (int q = 0; q < 3; ++q) { (void)hipFree(h->cv_int[q]); (void)hipFree(h->cv_int2[q]); }
  (void)hipFree(h->cv_z); (void)hipFree(h->cv_z2);
  for (int q = 0; q < 2; ++q) (void)hipFree(h->cv_lists[q]);
  (void)hipFree(h->sbc_tracer);
  (void)hipFree(h->sbc_acc);
  (void)hipFree(h->vmix_tab);
  (void)hipFree(h->tsi_acc);
  if (h->tsi_host) { (void)hipHostFree(h->tsi_host); (void)hipEventDestroy(h->ev_tsi); }
  (void)hipFree(h->tavg_zt); (void)hipFree(h->tavg_diag); (void)hipFree(h->tavg_dc14);
  (void)hipEventDestroy(h->ev_ts_host);
  if (h->mobi_st.params) {
    (void)hipFree(h->mobi_st.params);
    (void)hipFree(h->mobi_st.work);
    (void)hipFree(h->mobi_st.opts);
    for (int q = 0; q < 2; ++q) (void)hipFree(h->mobi_st.work_side[q]);
    for (int q = 0; q < 8; ++q) (void)hipFree(h->mobi_st.f[q]);
  }
  for (auto e : h->ev_pool) (void)hipEventDestroy(e);
  if (h->src_alt) (void)hipFree(h->src_alt);
  (void)hipEventDestroy(h->ev_step_begin);
  for (int q = 0; q < 2; ++q) { (void)hipEventDestroy(h->ev_src_next[q]); (void)hipEventDestroy(h->ev_step_end[q]); }
  for (int q = 0; q < 2; ++q) (void)hipEventDestroy(h->ev_src_inline2[q]);
  for (void *q : h->pinned) (void)hipHostUnregister(q);
  h->pinned.clear();
  if (h->side_mom) { (void)hipStreamSynchronize(h->side_mom); (void)hipStreamDestroy(h->side_mom); (void)hipEventDestroy(h->ev_mom_in); (void)hipEventDestroy(h->ev_mom_done); }
  (void)hipStreamDestroy(h->side_m[0]);
  if (h->side_m[1] != h->side_m[0]) (void)hipStreamDestroy(h->side_m[1]);
  if (h->side_ts != h->side2) (void)hipStreamDestroy(h->side_ts);
  (void)hipStreamDestroy(h->side2);
  (void)hipEventDestroy(h->ev_fct_done);
  (void)hipEventDestroy(h->ev_ts_done);
  if (h->ev_ts_filt) (void)hipEventDestroy(h->ev_ts_filt);
  for (int q = 0; q < 3; ++q) (void)hipEventDestroy(h->iso_set[q].ev);
  (void)hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

extern "C" int64_t uvic_gpu_field_elems(uvic_gpu *h, int field) {
  if (!h || field < 0 || field >= UVIC_F_COUNT) return -1;
  return field_elems(h->d, field);
}
extern "C" void *uvic_gpu_field_devptr(uvic_gpu *h, int field) {
  if (!h || field < 0 || field >= UVIC_F_COUNT) return nullptr;
  return h->buf[field];
}
extern "C" void *uvic_gpu_stream(uvic_gpu *h) { return h ? (void *)h->stream : nullptr; }

__global__ void k_make_tmask(const int *kmt, double *tmask, int imt, int km, int jmt) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)imt * km * jmt;
  if (gid >= n) return;
  const int i = (int)(gid % imt), k = (int)((gid / imt) % km), j = (int)(gid / ((long long)imt * km));
  // updates/09/source/mom/loadmw.F:60-78
  tmask[gid] = (kmt[i + (size_t)imt * j] >= k + 1) ? 1.0 : 0.0;
}
static int make_tmask(uvic_gpu *h) {
  const long long n = (long long)h->d.imt * h->d.km * h->d.jmt;
  hipLaunchKernelGGL(k_make_tmask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->ctx.kmt,
                     (double *)h->ctx.tmask, h->d.imt, h->d.km, h->d.jmt);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  // the ocean columns (WetCols) follow kmt
  const int imt = h->d.imt, jmt = h->d.jmt;
  std::vector<int> kmt((size_t)imt * jmt), wet;
  HIPCHK(hipMemcpy(kmt.data(), h->buf[UVIC_F_KMT], kmt.size() * 4, hipMemcpyDeviceToHost));
  h->kmt_host = kmt;
  h->lanes_dirty = true;
  h->land_zeroed.clear();
  h->wet_row_start.assign((size_t)jmt + 2, 0);
  for (int j = 1; j <= jmt; ++j) {
    h->wet_row_start[j] = (int)wet.size();
    if (j >= 2 && j <= jmt - 1)
      for (int i = 2; i <= imt - 1; ++i)
        if (kmt[(size_t)(i - 1) + (size_t)imt * (j - 1)] > 0) wet.push_back((i - 1) + imt * (j - 1));
  }
  h->wet_row_start[jmt + 1] = (int)wet.size();
  for (int q = 0; q < 2; ++q) HIPCHK(hipStreamSynchronize(h->side_m[q]));   // a look-ahead chain may still walk the old list
  HIPCHK(hipStreamSynchronize(h->side2));
  h->prefetch_pending = h->src_from_prefetch = false;
  for (int q = 0; q < 3; ++q) h->iso_set[q].for_step = -1;
  // what was ocean may be land now: ai_coef_cell leaves land cells alone and counts on zeros there
  for (int q = 0; q < 3; ++q) {
    uvic_gpu::IsoSet &S = h->iso_set[q];
    if (!S.allocated) continue;
    for (int f = 5; f < 12; ++f) HIPCHK(hipMemsetAsync(S.f[f], 0, (size_t)field_elems(h->d, ISO_FIELDS[f]) * 8, h->stream));   // Ai_*, K11, K22, K33
    HIPCHK(hipMemsetAsync(S.coef, 0, (size_t)imt * h->d.km * jmt * 16 * CF_PAIRS, h->stream));
  }
  (void)hipFree(h->wet_dev);
  h->wet_dev = nullptr;
  HIPCHK(hipMalloc((void **)&h->wet_dev, (wet.size() + 1) * 4));
  if (!wet.empty()) HIPCHK(hipMemcpy(h->wet_dev, wet.data(), wet.size() * 4, hipMemcpyHostToDevice));
  h->src_zeroed.clear();   // what was ocean may be land now
  // pass A stores the flux of ocean cells only
  HIPCHK(hipMemsetAsync(h->fny, 0, (size_t)imt * h->d.km * jmt * 8 * (size_t)h->d.nt, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
// Lane maps of the column passes for the slab js..je (kernels_col.hpp: ColGrid).
//   pass A covers one row beyond the slab on each side.  In every row the ocean columns form runs (cyclic in longitude;
//   runs closer than five columns are joined, marching over the land between them costs less than the four halo lanes
//   of a cut); a run is laid into the waves with two halo lanes on each side, and cut -- with halo lanes again -- where
//   a wave ends.
//   pass B: the ocean columns of the slab and nothing else, row by row.
static int build_col_lanes(uvic_gpu *h) {
  if (!h->lanes_dirty) return 0;
  const int imt = h->d.imt, jmt = h->d.jmt, nx = imt - 2;
  const uvic_ctx &c = h->ctx;
  if (imt > 0xfff || jmt > 0xfff) return fail_msg("column kernels: imt, jmt <= 4095 (lane codes hold 12 bits each)");
  const bool have_kmt = h->kmt_host.size() == (size_t)imt * jmt;
  auto wet = [&](int x, int r) {   // x = 0..nx-1 <-> column x+2
    return c.no_landskip || !have_kmt || h->kmt_host[(size_t)(x + 1) + (size_t)imt * (r - 1)] > 0;
  };
  auto code_of = [&](int x, int r, int owned) { return (((x % nx + nx) % nx) + 2) | (r << 12) | (owned << 24); };
  std::vector<int> la, lb;
  // (with the final y flux formed in pass A, the pass needs rows js-1..je: the flux through the north face of row je is
  // formed there from t of rows up to je+2, which the 2-row halo holds; else R+-Y of row je+1 is needed as well)
  const int ra0 = c.js - 1 < 2 ? 2 : c.js - 1, ra1 = std::min(c.je, jmt - 1);
  for (int r = ra0; r <= ra1; ++r) {
    // runs of this row as (start x, length), cyclic
    std::vector<std::pair<int, int>> runs;
    int nwet = 0;
    for (int x = 0; x < nx; ++x) nwet += wet(x, r) ? 1 : 0;
    if (nwet == 0) continue;
    if (nwet == nx) runs.push_back({0, nx});
    else {
      int x0 = 0;
      while (wet(x0, r)) ++x0;            // a land column to start the scan from
      for (int t = 1; t < nx;) {          // t counts columns after x0; a run cannot pass x0 (land)
        if (!wet((x0 + t) % nx, r)) { ++t; continue; }
        int len = 0;
        while (t + len < nx && wet((x0 + t + len) % nx, r)) ++len;
        runs.push_back({(x0 + t) % nx, len});
        t += len;
      }
      // join runs whose gap is at most 4 columns (the last with the first as well: cyclic), unless that closes the circle
      bool joined = true;
      while (joined && runs.size() > 1) {
        joined = false;
        for (size_t q = 0; q < runs.size(); ++q) {
          const size_t qn = (q + 1) % runs.size();
          const int end = runs[q].first + runs[q].second;                       // first column after run q
          const int gap = ((runs[qn].first - end) % nx + nx) % nx;
          if (gap <= 4 && runs[q].second + gap + runs[qn].second < nx) {
            runs[q].second += gap + runs[qn].second;
            runs.erase(runs.begin() + (long)qn);
            joined = true;
            break;
          }
        }
      }
    }
    for (auto &run : runs) {
      int x = run.first, left = run.second;
      while (left > 0) {
        int space = 64 - (int)(la.size() % 64);
        if (space < 5) {                       // not even one owned lane fits: pad the wave
          for (int q = 0; q < space; ++q) la.push_back(code_of(x, r, 0));
          space = 64;
        }
        const int own = std::min(left, space - 4);
        la.push_back(code_of(x - 2, r, 0)); la.push_back(code_of(x - 1, r, 0));
        for (int q = 0; q < own; ++q) la.push_back(code_of(x + q, r, 1));
        la.push_back(code_of(x + own, r, 0)); la.push_back(code_of(x + own + 1, r, 0));
        x += own; left -= own;
      }
    }
  }
  while (la.size() % 64) la.push_back(la.empty() ? code_of(0, ra0, 0) : (la.back() & ~(1 << 24)));
  for (int r = c.js; r <= c.je; ++r)
    for (int x = 0; x < nx; ++x)
      if (wet(x, r)) lb.push_back(code_of(x, r, 1));
  while (lb.size() % 64) lb.push_back(lb.empty() ? code_of(0, c.js, 0) : (lb.back() & ~(1 << 24)));
  (void)hipFree(h->lanes_dev);
  h->lanes_dev = nullptr;
  HIPCHK(hipMalloc((void **)&h->lanes_dev, (la.size() + lb.size() + 64) * 4));
  if (!la.empty()) HIPCHK(hipMemcpy(h->lanes_dev, la.data(), la.size() * 4, hipMemcpyHostToDevice));
  if (!lb.empty()) HIPCHK(hipMemcpy(h->lanes_dev + la.size(), lb.data(), lb.size() * 4, hipMemcpyHostToDevice));
  h->nwaves_a = (int)(la.size() / 64); h->nwaves_b = (int)(lb.size() / 64);
  h->lanes_dirty = false;
  return 0;
}
// Pass B leaves land alone: a buffer about to receive t(tau+1) gets its land columns cleared once (first use, or first use
// since kmt, the slab or the buffer's contents came from the host); the cleared buffers are remembered by address (the
// three time levels rotate by pointer)
static int land_clean(uvic_gpu *h, const uvic_ctx &c, hipStream_t st) {
  void *tp = (void *)c.t_taup1;
  for (void *q : h->land_zeroed)
    if (q == tp) return 0;
  const long long n = (long long)c.imt * c.km * (c.je - c.js + 1) * c.nt_local;
  if (n > 0) hipLaunchKernelGGL(k_land_zero, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c, c.t_taup1);
  HIPCHK(hipGetLastError());
  if (c.n0 == 0 && c.nt_local == c.nt) h->land_zeroed.push_back(tp);   // a partial launch (T,S alone) does not vouch for the buffer
  return 0;
}
// the ocean columns of rows js..je
static WetCols wet_range(const uvic_gpu *h, int js, int je) {
  WetCols w;
  w.ij = h->wet_dev;
  w.first = h->wet_row_start[js];
  w.count = h->wet_row_start[je + 1] - w.first;
  return w;
}
// MOBI writes the sources of ocean columns only; a buffer it is about to fill for the first time (or the first
// time since kmt or src came from the host) is cleared on the same stream before
static int src_clean(uvic_gpu *h, void *src, hipStream_t st) {
  for (void *p : h->src_zeroed)
    if (p == src) return 0;
  HIPCHK(hipMemsetAsync(src, 0, (size_t)field_elems(h->d, UVIC_F_SRC) * 8, st));
  h->src_zeroed.push_back(src);
  return 0;
}

// a transfer touches fields a momentum step on its own stream (uvic_gpu_momentum_async) may still read or write: wait for it
static int mom_host_join(uvic_gpu *h) {
  if (h->mom_pending) {
    HIPCHK(hipEventSynchronize(h->ev_mom_done));
    h->mom_pending = false;
  }
  return 0;
}
// new state (a time level of t) or new sources from the host: whatever the side streams computed ahead from the old ones
// is void, and they must have finished before the buffers change under them
static int state_from_host(uvic_gpu *h, int field) {
  if (field != UVIC_F_T_TAUM1 && field != UVIC_F_T_TAU && field != UVIC_F_T_TAUP1 && field != UVIC_F_SRC) return 0;
  for (int q = 0; q < 2; ++q) HIPCHK(hipStreamSynchronize(h->side_m[q]));
  HIPCHK(hipStreamSynchronize(h->side2));
  HIPCHK(hipStreamSynchronize(h->side_ts));
  h->prefetch_pending = h->src_from_prefetch = false;
  for (int q = 0; q < 3; ++q) h->iso_set[q].for_step = -1;
  h->ts_final_valid = false;
  h->end_ready = h->end_pending = false;
  return 0;
}
// the host wrote into a time level of t: its land columns are the host's business again (land_clean)
static void velocity_touched(uvic_gpu *h, int field) {
  if (field == UVIC_F_ZW) h->vmix_tab_ready = false;
  if (field == UVIC_F_DIFF_CBT) h->ctx.vmix_dev = 0;   // the caller brought this step's diff_cbt itself
  if (field != UVIC_F_ADV_VET && field != UVIC_F_ADV_VNT && field != UVIC_F_ADV_VBT) return;
  for (int q = 0; q < 3; ++q)
    if (h->iso_set[q].for_step >= 0) h->iso_set[q].vel_stale = true;
}
static void land_touched(uvic_gpu *h, int field) {
  if (field != UVIC_F_T_TAUM1 && field != UVIC_F_T_TAU && field != UVIC_F_T_TAUP1) return;
  auto &v = h->land_zeroed;
  v.erase(std::remove(v.begin(), v.end(), h->buf[field]), v.end());
}
extern "C" int uvic_gpu_upload(uvic_gpu *h, int field, const void *host, int64_t offset, int64_t count) {
  if (h) h->idle_until_next = false;   // something is queued on the main stream between two steps
  if (!h || !host) return fail_msg("uvic_gpu_upload: null argument");
  if (field < 0 || field >= UVIC_F_COUNT) return fail_msg("uvic_gpu_upload: bad field id");
  const int64_t n = field_elems(h->d, field);
  if (offset < 0 || count < 0 || offset + count > n) return fail_msg(std::string("uvic_gpu_upload: range outside field ") + FIELDS[field].name);
  HIPCHK(hipSetDevice(h->device));
  if (field != UVIC_F_PSI)      // (the stream function is not read by a momentum step running beside the main stream)
    if (int rc = mom_host_join(h)) return rc;
  const size_t es = elem_size(field);
  if (int rc = state_from_host(h, field)) return rc;
  HIPCHK(hipMemcpyAsync((char *)h->buf[field] + offset * es, host, count * es, hipMemcpyHostToDevice, h->stream));
  // (the time levels of t are not among what the resident overlay re-fills every step: always waited for)
  const bool t_field = field == UVIC_F_T_TAUM1 || field == UVIC_F_T_TAU || field == UVIC_F_T_TAUP1;
  if (h->host_sync || field == UVIC_F_KMT || t_field) HIPCHK(hipStreamSynchronize(h->stream));
  if (field == UVIC_F_ITRC && offset == 0 && count >= 2) {
    const int32_t *it = (const int32_t *)host;
    h->ts_no_src = it[0] == 0 && it[1] == 0;
  }
  if (field == UVIC_F_SRC) h->src_zeroed.clear();
  land_touched(h, field);
  velocity_touched(h, field);
  if (field == UVIC_F_KMT) return make_tmask(h);
  return 0;
}
extern "C" int uvic_gpu_download(uvic_gpu *h, int field, void *host, int64_t offset, int64_t count) {
  if (!h || !host) return fail_msg("uvic_gpu_download: null argument");
  if (int rc = mom_host_join(h)) return rc;
  if (field < 0 || field >= UVIC_F_COUNT) return fail_msg("uvic_gpu_download: bad field id");
  const int64_t n = field_elems(h->d, field);
  if (offset < 0 || count < 0 || offset + count > n) return fail_msg(std::string("uvic_gpu_download: range outside field ") + FIELDS[field].name);
  HIPCHK(hipSetDevice(h->device));
  const size_t es = elem_size(field);
  HIPCHK(hipMemcpyAsync(host, (const char *)h->buf[field] + offset * es, count * es, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

static double g_xfer_ms = 0.0;   // UVIC_OVL_TIMING: host time spent inside the transfer entry points since the last report
#if defined(UVIC_EXPERIMENTS)
// ... and where the host is, in wall time since the first overlay entry point of the step, when it submits the parts of a call
static std::vector<std::pair<const char *, std::chrono::steady_clock::time_point>> g_stamps;
static void ovl_stamp(const char *what) {
  static const bool on = uv_env("UVIC_OVL_TIMING") != nullptr;
  if (on) g_stamps.emplace_back(what, std::chrono::steady_clock::now());
}
static void ovl_stamps_print() {
  if (g_stamps.empty()) return;
  fprintf(stderr, "overlay host stamps (us):");
  for (auto &q : g_stamps) fprintf(stderr, " %s %.0f", q.first, std::chrono::duration<double, std::micro>(q.second - g_stamps[0].second).count());
  fprintf(stderr, "\n");
  g_stamps.clear();
}
#else
static inline void ovl_stamp(const char *) {}
static inline void ovl_stamps_print() {}
#endif
struct XferTimer {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  ~XferTimer() { g_xfer_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
static int rows_xfer(uvic_gpu *h, int field, double *host, int jlo, int jhi, bool up) {
  XferTimer tm_;
  if (!h || !host) return fail_msg("uvic_gpu_*_rows: null argument");
  if (int rc = mom_host_join(h)) return rc;
  if (field < 0 || field >= UVIC_F_COUNT || FIELDS[field].is_int) return fail_msg("uvic_gpu_*_rows: bad field id");
  const Kind kd = FIELDS[field].kind;
  if (kd != K_S && kd != K_C && kd != K_F) return fail_msg("uvic_gpu_*_rows: field has no row dimension");
  if (jlo < 1 || jhi > h->d.jmt || jlo > jhi) return fail_msg("uvic_gpu_*_rows: row range outside 1..jmt");
  const int64_t rowlen = plane(h->d, kd) / h->d.jmt;
  const int64_t ex = extra_of(h->d, FIELDS[field].extra);
  const int64_t nrows = jhi - jlo + 1;
  HIPCHK(hipSetDevice(h->device));
  if (up && field == UVIC_F_SRC) h->src_zeroed.clear();
  if (up) {
    if (int rc = state_from_host(h, field)) return rc;
    land_touched(h, field); velocity_touched(h, field);
  }
  if (jlo == 1 && jhi == h->d.jmt && ex > 1) {   // every row: the planes of all `extra` entries are one contiguous range
    if (up)                                       // (one copy instead of nt of them: each costs ~10 us of stream time)
      HIPCHK(hipMemcpyAsync(h->buf[field], host, (size_t)ex * plane(h->d, kd) * 8, hipMemcpyHostToDevice, h->stream));
    else
      HIPCHK(hipMemcpyAsync(host, h->buf[field], (size_t)ex * plane(h->d, kd) * 8, hipMemcpyDeviceToHost, h->stream));
  } else
  for (int64_t e = 0; e < ex; ++e) {
    char *dev = (char *)h->buf[field] + (e * plane(h->d, kd) + (int64_t)(jlo - 1) * rowlen) * 8;
    char *hst = (char *)host + e * nrows * rowlen * 8;
    if (up)
      HIPCHK(hipMemcpyAsync(dev, hst, nrows * rowlen * 8, hipMemcpyHostToDevice, h->stream));
    else
      HIPCHK(hipMemcpyAsync(hst, dev, nrows * rowlen * 8, hipMemcpyDeviceToHost, h->stream));
  }
  const bool t_field = field == UVIC_F_T_TAUM1 || field == UVIC_F_T_TAU || field == UVIC_F_T_TAUP1;
  if (h->host_sync || !up || t_field) HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int uvic_gpu_upload_rows(uvic_gpu *h, int field, const double *host, int jlo, int jhi) {
  if (h) h->idle_until_next = false;   // something is queued on the main stream between two steps
  return rows_xfer(h, field, (double *)host, jlo, jhi, true);
}
extern "C" int uvic_gpu_download_rows(uvic_gpu *h, int field, double *host, int jlo, int jhi) {
  return rows_xfer(h, field, host, jlo, jhi, false);
}

// one level of one tracer (or of a plain cell field: n = 1) as an (imt, jmt) plane: what set_sbc wants of t(tau+1)
// when the state stays on the device (u09/mom/set_sbc.F:36-72 reads t(i,1,j,n,taup1) only)
extern "C" int uvic_gpu_download_level(uvic_gpu *h, int field, int n, int k, double *host) {
  if (!h || !host) return fail_msg("uvic_gpu_download_level: null argument");
  if (field < 0 || field >= UVIC_F_COUNT || FIELDS[field].is_int || FIELDS[field].kind != K_C)
    return fail_msg("uvic_gpu_download_level: field is not a cell field (imt,km,jmt[,n])");
  const int64_t ex = extra_of(h->d, FIELDS[field].extra);
  if (n < 0 || n > ex || k < 1 || k > h->d.km) return fail_msg("uvic_gpu_download_level: n or k out of range");
  HIPCHK(hipSetDevice(h->device));
  // rows of one level lie imt*km apart, within a tracer and from the last row of one tracer to the first of the next:
  // one strided copy serves one tracer (jmt rows) or all of them (n = 0: jmt*ex rows -> host (imt, jmt, ex))
  const size_t N3 = (size_t)h->d.imt * h->d.km * h->d.jmt;
  const char *src = (const char *)h->buf[field] + ((size_t)(n > 0 ? n - 1 : 0) * N3 + (size_t)(k - 1) * h->d.imt) * 8;
  const size_t rows = (size_t)h->d.jmt * (n > 0 ? 1 : (size_t)ex);
  HIPCHK(hipMemcpy2DAsync(host, (size_t)h->d.imt * 8, src, (size_t)h->d.imt * h->d.km * 8, (size_t)h->d.imt * 8, rows,
                          hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int uvic_gpu_set_params(uvic_gpu *h, const uvic_params *p) {
  if (!h || !p) return fail_msg("uvic_gpu_set_params: null argument");
  h->ctx.c2dtts = p->c2dtts; h->ctx.aidif = p->aidif;
  h->ctx.diff_cet = p->diff_cet; h->ctx.diff_cnt = p->diff_cnt;
  h->ctx.slmxr = p->slmxr; h->ctx.ahisop = p->ahisop; h->ctx.athkdf = p->athkdf;
  h->ctx.diff_cbt_given = p->diff_cbt_has_k33 ? 1 : 0;
  return 0;
}
extern "C" int uvic_gpu_sync(uvic_gpu *h);
extern "C" int uvic_gpu_set_exact(uvic_gpu *h, int exact) {
  if (!h) return fail_msg("uvic_gpu_set_exact: null handle");
  if (int rc = uvic_gpu_sync(h)) return rc;   // (a look-ahead chain may be writing what the other arithmetic recomputes in line)
  h->exact = exact == 1;       // 1: every tracer through the bit-exact kernels
  h->ts_exact = exact != 2;    // 0 (default): T and S exact, the others through the column kernels; 2: every tracer through the column kernels
  h->ts_rows = exact == 3;     // 3: as 0 with T and S through the row kernels (cross-check of the exact column kernels)
  for (int q = 0; q < 3; ++q) h->iso_set[q].for_step = -1;   // a look-ahead chain formed its products for the other arithmetic
  return 0;
}
extern "C" int uvic_gpu_set_shard(uvic_gpu *h, int n0, int nt_local, int js, int je) {
  if (!h) return fail_msg("uvic_gpu_set_shard: null handle");
  if (n0 < 0 || nt_local < 0 || n0 + nt_local > h->d.nt) return fail_msg("uvic_gpu_set_shard: tracer range outside 1..nt");
  if (js < 2 || je > h->d.jmt - 1 || js > je) return fail_msg("uvic_gpu_set_shard: row range outside 2..jmt-1");
  if (js != h->ctx.js || je != h->ctx.je) { h->lanes_dirty = true; h->land_zeroed.clear(); }
  h->ctx.n0 = n0; h->ctx.nt_local = nt_local; h->ctx.js = js; h->ctx.je = je;
  return 0;
}

static void iso_set_adopt(uvic_gpu *h) {   // uvic_gpu_create: the buffers just allocated are set 0
  uvic_gpu::IsoSet &S = h->iso_set[0];
  for (int q = 0; q < 16; ++q) S.f[q] = h->buf[ISO_FIELDS[q]];
  for (int q = 0; q < 3; ++q) S.work[q] = h->work[q];
  S.coef = h->coef;
  S.allocated = true;
  h->iso_cur = 0;
}
static void iso_set_release(uvic_gpu *h) {
  for (int s = 0; s < 3; ++s) {
    uvic_gpu::IsoSet &S = h->iso_set[s];
    if (!S.allocated) continue;
    for (int q = 0; q < 16; ++q) (void)hipFree(S.f[q]);
    for (int q = 0; q < 3; ++q) (void)hipFree(S.work[q]);
    (void)hipFree(S.coef);
    S.allocated = false;
  }
  for (int q = 0; q < 16; ++q) h->buf[ISO_FIELDS[q]] = nullptr;
  for (int q = 0; q < 3; ++q) h->work[q] = nullptr;
  h->coef = nullptr;
}
static int iso_set_alloc(uvic_gpu *h, int s, hipStream_t st) {   // zero-filled like the first one
  uvic_gpu::IsoSet &S = h->iso_set[s];
  if (S.allocated) return 0;
  const uvic_dims &d = h->d;
  for (int q = 0; q < 16; ++q) {
    const size_t bytes = (size_t)field_elems(d, ISO_FIELDS[q]) * 8;
    HIPCHK(hipMalloc(&S.f[q], bytes));
    HIPCHK(hipMemsetAsync(S.f[q], 0, bytes, st));
  }
  const size_t N3 = (size_t)d.imt * d.km * d.jmt, NF = (size_t)d.imt * (d.km + 1) * d.jmt;
  const size_t wsz[3] = {N3, N3, NF};
  for (int q = 0; q < 3; ++q) {
    HIPCHK(hipMalloc((void **)&S.work[q], wsz[q] * 8));
    HIPCHK(hipMemsetAsync(S.work[q], 0, wsz[q] * 8, st));
  }
  HIPCHK(hipMalloc((void **)&S.coef, N3 * 16 * CF_PAIRS));
  HIPCHK(hipMemsetAsync(S.coef, 0, N3 * 16 * CF_PAIRS, st));
  HIPCHK(hipStreamSynchronize(st));   // once per set: whichever stream writes or reads it first finds the zeros
  S.allocated = true;
  return 0;
}
// point buf[], work[0..2], coef and the context at set s
static int use_iso_set(uvic_gpu *h, int s) {
  if (h->iso_cur == s) return 0;
  if (int rc = iso_set_alloc(h, s, h->stream)) return rc;
  const uvic_gpu::IsoSet &S = h->iso_set[s];
  for (int q = 0; q < 16; ++q) h->buf[ISO_FIELDS[q]] = S.f[q];
  for (int q = 0; q < 3; ++q) h->work[q] = S.work[q];
  h->coef = S.coef;
  h->iso_cur = s;
  bind_ctx(h);
  return 0;
}

// -- launch helpers ------------------------------------------------------------
static void mark_on(uvic_gpu *h, const char *name, int sid) {
  if (!h->profiling) return;
  size_t used = 0;
  for (int q = 0; q < 5; ++q) used += h->ev[q].size();
  if (used >= h->ev_pool.size()) {
    hipEvent_t e;
    (void)hipEventCreate(&e);
    h->ev_pool.push_back(e);
  }
  hipEvent_t e = h->ev_pool[used];
  hipStream_t sts[5] = {h->stream, h->side_m[0], h->side2, h->side_ts, h->side_m[1]};
  (void)hipEventRecord(e, sts[sid]);
  h->ev[sid].push_back(e);
  h->ev_names[sid].push_back(name);
}
static void mark(uvic_gpu *h, const char *name) { mark_on(h, name, 0); }
static unsigned cell_blocks(const uvic_gpu *h, int bs) {
  const long long n = (long long)h->d.imt * h->d.km * h->d.jmt;
  return (unsigned)((n + bs - 1) / bs);
}
static unsigned col_blocks(const uvic_gpu *h, int bs) {
  const long long n = (long long)h->d.imt * h->d.jmt;
  return (unsigned)((n + bs - 1) / bs);
}

// the T,S-derived fields of a step (mixing tensor, GM velocities, folded coefficients): ctx `c` says what is read
// (t_taum1) and where the products go; `sid` 0 = main stream, 2 = the isopyc side stream
static int launch_conv_pe(uvic_gpu *h, const uvic_ctx &c, hipStream_t st, int phase);
static int inputs_next_copy(uvic_gpu *h);
static int inputs_first(uvic_gpu *h, bool vbt_follows = false);
static int inputs_rest(uvic_gpu *h);
static int launch_isopyc_on(uvic_gpu *h, const uvic_ctx &c, double *coef, hipStream_t st, int sid) {
  mark_on(h, "begin", sid);
  hipLaunchKernelGGL(k_isopyc_elements, dim3(cell_blocks(h, 256)), dim3(256), 0, st, c);
  mark_on(h, "isopyc_elements", sid);
  if (h->exact)
    hipLaunchKernelGGL(k_isopyc_ai, dim3(cell_blocks(h, 256)), dim3(256), 0, st, c);
  else   // column-kernel path: mixing tensor and folded coefficients in one pass (Ai_* stay in registers)
    hipLaunchKernelGGL(k_ai_coef, dim3(cell_blocks(h, 256)), dim3(256), 0, st, c, coef, h->ts_exact ? 1 : 0);
  mark_on(h, "isopyc_ai", sid);
  hipLaunchKernelGGL(k_isopyc_adv, dim3(cell_blocks(h, 256)), dim3(256), 0, st, c, coef);
  mark_on(h, "isopyc_adv", sid);
  hipLaunchKernelGGL(k_isopyc_column, dim3(col_blocks(h, 64)), dim3(64, ISO_COL_PARTS), (size_t)3 * (h->d.km + 1) * 64 * 8, st, c, coef);
  mark_on(h, "isopyc_column", sid);
  HIPCHK(hipGetLastError());
  return 0;
}
static int launch_isopyc(uvic_gpu *h, bool may_defer = false) {
  h->iso_waited = false;
  h->prep_deferred = false;
  // velocities and diff_cbt queued by uvic_gpu_overlay_inputs: when the fields were computed ahead, one kernel forms
  // adv_vbt, the total velocities and the vertical-diffusion coefficient from them
  const int ahead_set = (int)(h->step_no % 3);
  const bool fuse = h->in.first_pending && h->in.derive_vbt && !h->exact && h->ctx.diff_cbt_given && !h->mixing &&
                    h->iso_set[ahead_set].for_step == h->step_no && h->iso_set[ahead_set].vel_stale;
  if (fuse && may_defer) h->in.used = true;   // (the wait goes to the stream that runs k_inputs_cell)
  else if (int rc = inputs_first(h, fuse)) return rc;
  const int set = (int)(h->step_no % 3);
  if (int rc = use_iso_set(h, set)) return rc;
  if (h->iso_set[set].for_step == h->step_no && h->mixing) {
    // computed ahead for a leapfrog step, but this is a forward step (t(tau-1) := t(tau)): void; redo it behind the chain
    HIPCHK(hipStreamWaitEvent(h->stream, h->iso_set[set].ev, 0));
    h->iso_set[set].for_step = -1;
  }
  if (h->iso_set[set].for_step == h->step_no) {   // computed ahead on a side stream (uvic_gpu_prefetch_isopyc)
    HIPCHK(hipStreamWaitEvent(h->stream, h->iso_set[set].ev, 0));
    h->iso_set[set].for_step = -1;
    h->iso_waited = true;
    if (fuse) {
      h->iso_set[set].vel_stale = false;
      h->iso_waited = false;
      if (may_defer) { h->prep_deferred = true; return 0; }   // launch_transport puts it at the head of the T,S stream
      hipLaunchKernelGGL(k_inputs_cell, dim3(cell_blocks(h, 256)), dim3(256), 0, h->stream, h->ctx, h->coef);
      mark(h, "inputs_cell");
      HIPCHK(hipGetLastError());
      return 0;
    }
    if (h->iso_set[set].vel_stale) {   // this step's velocities arrived after the chain ran
      const long long nf = (long long)h->d.imt * (h->d.km + 1) * h->d.jmt;
      hipLaunchKernelGGL(k_tot_vel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, h->stream, h->ctx, h->coef);
      mark(h, "tot_vel");
      h->iso_set[set].vel_stale = false;
      h->iso_waited = false;   // the T,S stream reads them too: it must not start before this
    }
    if (h->ctx.vmix_dev) {   // ... the step's own diff_cbt is formed here, from the fields the chain left
      hipLaunchKernelGGL(k_vmixc, dim3(cell_blocks(h, 256)), dim3(256), 0, h->stream, h->ctx);
      mark(h, "vmixc");
      h->iso_waited = false;
    }
    if (h->ctx.diff_cbt_given && !h->exact) {   // ... and so did the step's own diff_cbt
      hipLaunchKernelGGL(k_coef_bv, dim3(cell_blocks(h, 256)), dim3(256), 0, h->stream, h->ctx, h->coef);
      mark(h, "coef_bv");
      h->iso_waited = false;
    }
    HIPCHK(hipGetLastError());
    return 0;
  }
  if (int rc = launch_isopyc_on(h, h->ctx, h->coef, h->stream, 0)) return rc;
  if (h->ctx.vmix_dev) {   // vmixc follows isopyc (mom.F:340-347): diff_cbt, then the coefficient folded from it
    hipLaunchKernelGGL(k_vmixc, dim3(cell_blocks(h, 256)), dim3(256), 0, h->stream, h->ctx);
    mark(h, "vmixc");
    if (!h->exact) {
      hipLaunchKernelGGL(k_coef_bv, dim3(cell_blocks(h, 256)), dim3(256), 0, h->stream, h->ctx, h->coef);
      mark(h, "coef_bv");
    }
    HIPCHK(hipGetLastError());
  }
  return 0;
}
// the time-step integrals of the tracers of `c` on stream `st`: after their pass B, before convection (diagt1, tracer.F:1161)
static int launch_tsi_rows(uvic_gpu *h, const uvic_ctx &c, hipStream_t st, int sid) {
  if (!h->tsi_step || c.nt_local <= 0) return 0;
  const long long n = (long long)c.km * (c.je - c.js + 1) * c.nt_local;   // rows: TSI_ROWS to a wave
  hipLaunchKernelGGL(k_tsi_rows, dim3((unsigned)((n + TSI_ROWS - 1) / TSI_ROWS)), dim3(64), 0, st, c, h->tsi_acc);
  mark_on(h, "tsi_rows", sid);
  HIPCHK(hipGetLastError());
  return 0;
}
// the bit-exact row kernels (kernels_fct.hpp) for the tracers of `c` (n0, nt_local) on stream `st`; marks go to list `sid`
static int launch_rows(uvic_gpu *h, const uvic_ctx &c, hipStream_t st, int sid, int *zero_word, const char *name_a, const char *name_b) {
  if (c.nt_local <= 0) return 0;
  TileGrid g1, g2;
  g1.r0 = c.js - 1 < 2 ? 2 : c.js - 1;
  const int r1 = c.je + 1 > c.jmt - 1 ? c.jmt - 1 : c.je + 1;
  g1.nrows = r1 - g1.r0 + 1;
  g1.nchunk = h->nchunk;
  g1.total = g1.nrows * c.nt_local * h->nchunk;
  g1.zero_word = nullptr;
  g2.r0 = c.js; g2.nrows = c.je - c.js + 1; g2.nchunk = h->nchunk;
  g2.total = g2.nrows * c.nt_local * h->nchunk;
  g2.zero_word = zero_word;
  hipLaunchKernelGGL(k_fct_rows, dim3((unsigned)(((g1.total + 7) / 8) * 8)), dim3(h->fct_threads), h->fct_lds, st, c, g1);
  mark_on(h, name_a, sid);
  if (sid == 0 && h->src_from_prefetch) {  // the FCT kernel does not read the sources; only the update does
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_src_ready, 0));
    h->src_from_prefetch = false;
  }
  hipLaunchKernelGGL(k_update_rows, dim3((unsigned)(((g2.total + 7) / 8) * 8)), dim3(h->upd_threads), h->upd_lds, st, c, g2);
  mark_on(h, name_b, sid);
  HIPCHK(hipGetLastError());
  return 0;
}
// T and S (the local tracers among them: c.n0 < 2, c.nt_local <= 2) through the exact column kernels on stream `st`.
// `walk`: pass B and the convective T,S walk in one launch (both tracers local, the whole `tracer` step)
static int launch_colx(uvic_gpu *h, const uvic_ctx &c, const ColGrid &a, const ColGrid &b, hipStream_t st, int sid, bool walk,
                       hipEvent_t update_after = nullptr) {
  if (c.nt_local <= 0) return 0;
  ColxOut o;
  o.adv_x = h->work[3]; o.adv_z = h->work[4]; o.fn = h->work[5]; o.dif = h->work[6];
  ColGrid ga = a, gb = b;
  ga.zero_word = walk ? h->cv_list : nullptr;
  gb.total = gb.nwaves * c.nt_local;
  if (ga.nwaves > 0) hipLaunchKernelGGL(k_colx_fct, dim3((unsigned)(((ga.nwaves + 7) / 8) * 8)), dim3(64, 4), 0, st, c, o, ga);
  mark_on(h, "colx_fct_ts", sid);
  if (sid == 0 && h->src_from_prefetch) {  // (T or S with a source term: only the update reads it)
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_src_ready, 0));
    h->src_from_prefetch = false;
  }
  if (update_after) HIPCHK(hipStreamWaitEvent(st, update_after, 0));   // what still reads the buffer t(tau+1) goes to
  if (gb.nwaves > 0) {
    if (walk) {
      const size_t lds_b = ((size_t)2 * 2 * (c.km + 1) * 64 + (size_t)12 * c.km) * 8;
      hipLaunchKernelGGL(k_colx_upd_conv, dim3((unsigned)(((gb.nwaves + 7) / 8) * 8)), dim3(64, 2), lds_b, st, c, o, gb, h->cv_list);
    } else {
      hipLaunchKernelGGL(k_colx_upd, dim3((unsigned)(((gb.total + 7) / 8) * 8)), dim3(64), (size_t)2 * (c.km + 1) * 64 * 8, st, c, o, gb);
    }
  }
  mark_on(h, walk ? "colx_upd_conv_ts" : "colx_upd_ts", sid);
  HIPCHK(hipGetLastError());
  return 0;
}
// `convect_follows`: the caller runs convct2 right after (the whole `tracer` step): T,S may go first and the replay be fused
static int launch_transport(uvic_gpu *h, bool convect_follows) {
  const uvic_ctx &c = h->ctx;
  if (c.nt_local <= 0) return 0;
  if (c.c2dtts == 0.0) return fail_msg("uvic_gpu_transport: c2dtts not set (uvic_gpu_set_params)");
  // T and S first, on the side stream: they have no source terms, and their t(tau+1) is all the convective walk needs.  Both
  // passes and the walk run there while the main stream works on the other tracers; the mixed ranges the walk lists are
  // replayed on those by k_convect_apply_list afterwards (launch_convect).  Not under tracer sharding, where convection
  // follows the exchange.
  const bool split = !h->exact && convect_follows && !h->serial && !h->exact_convect && c.n0 == 0 && c.nt_local == c.nt && c.nt > 2 && h->ts_no_src;
  bool ts_stream_ready = false;
  hipEvent_t ts_update_after = nullptr;
  if (h->prep_deferred) {
    // the inputs have been queued beside the main stream and the T,S-derived fields were computed ahead: what is left to
    // form from them (k_inputs_cell) goes to the head of the stream whose chain the host waits for -- one hop between
    // streams less on it (a hop costs ~25 us); the other stream follows by an event
    h->prep_deferred = false;
    hipStream_t st = split && h->step_begun ? h->side_ts : h->stream;
    const int sid = st == h->stream ? 0 : 3;
    if (st != h->stream) {
      // What T and S of this step must not overtake.  For a caller that waits for them (the resident overlay) the other
      // tracers of the PREVIOUS step are not among it: their passes read none of what the T,S chain writes -- its planes of
      // t(tau+1) and of the work arrays, this step's set of T,S-derived fields, its copy of the inputs and its set of the
      // convective walk's arrays, all of which were last read two or three steps ago -- except the MOBI chain of the
      // previous step, which reads T and S of what is now the t(tau+1) buffer.  So: the end of the step before the previous
      // one and that chain, when both are known; the previous step's end otherwise.
      const int par = (int)(h->step_no & 1);
      const bool mobi_known = !h->have_mobi || h->mobi_ev_step[par ^ 1] == h->step_no - 1;
      // (and every time level has had its land columns cleared: land_clean runs on the main stream)
      const bool relaxed = h->ts_host && h->flt_nitems == 0 && !h->tsi_step && !h->mixing && mobi_known &&
                           h->end_step_of[h->ev_flip] == h->step_no - 2 && h->land_zeroed.size() >= 3;
      if (relaxed) {
        HIPCHK(hipStreamWaitEvent(st, h->ev_step_end[h->ev_flip], 0));
        if (h->have_mobi) {   // (only the update writes that buffer: pass A may run beside the chain)
          if (h->ts_exact && !h->ts_rows) ts_update_after = h->ev_mobi_of[par ^ 1];
          else HIPCHK(hipStreamWaitEvent(st, h->ev_mobi_of[par ^ 1], 0));
        }
      } else {
        HIPCHK(hipStreamWaitEvent(st, h->ev_begin_cur, 0));
        h->ts_waited_begin = h->step_no;
      }
      if (h->iso_set[h->iso_cur].st != st) HIPCHK(hipStreamWaitEvent(st, h->iso_set[h->iso_cur].ev, 0));
    }
    HIPCHK(hipStreamWaitEvent(st, h->in.ev_first, 0));
    if (h->in.vel_pending) { HIPCHK(hipStreamWaitEvent(st, h->in.ev_vel, 0)); h->in.vel_pending = false; }
    h->in.first_pending = false; h->in.waited = true;
    uvic_ctx cp = c;
    cp.prio |= 4;
    mark_on(h, "begin", sid);
    hipLaunchKernelGGL(k_inputs_cell, dim3(cell_blocks(h, 256)), dim3(256), 0, st, cp, h->coef);
    mark_on(h, "inputs_cell", sid);
    ovl_stamp("inputs_cell");
    if (st != h->stream) {
      HIPCHK(hipEventRecord(h->ev_fct_done, st));
      HIPCHK(hipStreamWaitEvent(h->stream, h->ev_fct_done, 0));
      ts_stream_ready = true;
    }
  }
  if (int rc = inputs_first(h)) return rc;
  if (h->exact) {
    if (int rc = inputs_rest(h)) return rc;
    mark(h, "begin");
    if (int rc = launch_rows(h, c, h->stream, 0, nullptr, "fct_rows", "update_rows")) return rc;
    return launch_tsi_rows(h, c, h->stream, 0);
  }
  // lane-per-column path
  if (int rc = build_col_lanes(h)) return rc;
  ColGrid a, b;
  a.lanes = h->lanes_dev; a.nwaves = h->nwaves_a; a.total = a.nwaves * c.nt_local;
  b.lanes = h->lanes_dev + (size_t)h->nwaves_a * 64; b.nwaves = h->nwaves_b; b.total = b.nwaves * c.nt_local;
  a.zero_word = b.zero_word = nullptr;
  double *S = h->work[3];
  const size_t N3 = (size_t)c.imt * c.km * c.jmt;
  const size_t upd_lds = (size_t)2 * (c.km + 1) * 64 * 8;
  const double *cf = (const double *)h->coef;
  auto blocks8 = [](int n) { return (unsigned)(((n + 7) / 8) * 8); };
  // pass A of the bulk launch (the four waves of a workgroup: four tracers of the same lanes; g.total counts waves)
  auto launch_a = [&](const uvic_ctx &cc, ColGrid g, const double *Sg, hipStream_t st) {
    g.total = g.nwaves * ((cc.nt_local + 3) / 4) * 4;
    if (g.total <= 0) return;
    hipLaunchKernelGGL(k_colfct, dim3(blocks8(g.total / 4)), dim3(64, 4), (size_t)2 * COL_SHARE_SLOTS * 64 * 16, st, cc, cf, (double *)Sg, g);
  };
  auto launch_b = [&](const uvic_ctx &cc, const ColGrid &g, const double *Sg, hipStream_t st) {
    if (g.total <= 0) return;
    hipLaunchKernelGGL(k_colupd, dim3(blocks8(g.total)), dim3(64), upd_lds, st, cc, Sg, g);
  };
  // the tracers of `cc` (T and/or S) through the column kernels on stream `st` (set_exact(2)); `walk`: pass B and the
  // convective T,S walk in one launch
  auto launch_ts_columns = [&](const uvic_ctx &cc, hipStream_t st, int sid, bool walk) {
    ColGrid ga = a, gb = b;
    ga.total = ga.nwaves * cc.nt_local; gb.total = gb.nwaves * cc.nt_local;
    ga.zero_word = walk ? h->cv_list : nullptr;
    if (ga.total > 0) hipLaunchKernelGGL(k_colfct_ts, dim3(blocks8((ga.total + 3) / 4)), dim3(64, 4), 0, st, cc, cf, S, ga);
    mark_on(h, "colfct_ts", sid);
    if (walk) {
      const size_t lds_b = ((size_t)2 * 2 * (c.km + 1) * 64 + (size_t)12 * c.km) * 8;
      hipLaunchKernelGGL(k_colupd_conv_ts, dim3(blocks8(gb.nwaves)), dim3(64, 2), lds_b, st, cc, (const double *)S, gb, h->cv_list);
      mark_on(h, "colupd_conv_ts", sid);
    } else {
      launch_b(cc, gb, (const double *)S, st);
      mark_on(h, "colupd_ts", sid);
    }
  };
  if (split) {
    // the side stream may start when the previous step is complete and this step's T,S-derived fields are: both
    // have events already when the fields came from the look-ahead chain (no extra packet on the main stream)
    if (ts_stream_ready) {
      // (k_inputs_cell ran at its head, behind the same events)
    } else if (h->iso_waited && h->step_begun && !h->in.waited) {
      HIPCHK(hipStreamWaitEvent(h->side_ts, h->ev_begin_cur, 0));
      h->ts_waited_begin = h->step_no;
      if (h->iso_set[h->iso_cur].st != h->side_ts)   // (the chain ran on this very stream: nothing to wait for, and a wait packet costs ~6 us)
        HIPCHK(hipStreamWaitEvent(h->side_ts, h->iso_set[h->iso_cur].ev, 0));
    } else {
      HIPCHK(hipEventRecord(h->ev_fct_done, h->stream));
      HIPCHK(hipStreamWaitEvent(h->side_ts, h->ev_fct_done, 0));
    }
    uvic_ctx cts = c;
    cts.nt_local = 2;
    cts.prio |= 2;
    mark_on(h, "begin", 3);
    const WetCols w = wet_range(h, c.js, c.je);
    // (a time-step-monitor step wants t(tau+1) of T and S before convection: pass B, the sums, then the walk on its own)
    // (and so does a time-average step: the convection diagnostics sum over the column before and after the walk)
    const bool walk_fused = w.count > 0 && !h->tsi_step && !h->tavg_step && !(h->ts_exact && h->ts_rows);
    if (h->ts_exact && h->ts_rows) {   // cross-check: the row kernels of kernels_fct.hpp
      if (int rc = launch_rows(h, cts, h->side_ts, 3, nullptr, "fct_rows_ts", "update_rows_ts")) return rc;
    } else if (h->ts_exact) {
      // T and S in the reference's own order of operations (kernels_colx.hpp): every convective adjustment is decided on
      // their bits (convect.F:189-255), and a density comparison of rounding size flips on a 1-ulp difference
      if (int rc = launch_colx(h, cts, a, b, h->side_ts, 3, walk_fused, ts_update_after)) return rc;
    } else {
      launch_ts_columns(cts, h->side_ts, 3, walk_fused);
    }
    if (!walk_fused) {
      if (int rc = launch_tsi_rows(h, cts, h->side_ts, 3)) return rc;
      if (int rc = launch_conv_pe(h, cts, h->side_ts, 0)) return rc;
      HIPCHK(hipMemsetAsync(h->cv_list, 0, 4, h->side_ts));
      if (w.count > 0)
        hipLaunchKernelGGL(k_convect_ts, dim3((unsigned)((w.count + 63) / 64)), dim3(64), ((size_t)2 * h->d.km * 64 + (size_t)12 * h->d.km) * 8, h->side_ts, cts, w, h->cv_list);
      mark_on(h, "convect_ts", 3);
      if (int rc = launch_conv_pe(h, cts, h->side_ts, 1)) return rc;
    }
    HIPCHK(hipEventRecord(h->ev_ts_done, h->side_ts));
    if (h->flt_nitems > 0) {
      // the polar filter of T and S (tracer.F:1245, after convection) here, behind their walk: they are final -- and on
      // their way to a host that waits for them -- while the other tracers are still in their passes
      uvic_ctx cf = cts;
      cf.prio = c.prio;
      hipLaunchKernelGGL(k_filt, dim3((unsigned)h->flt_nitems * 2u), dim3(h->flt_threads), (size_t)(2 * h->flt_threads + 4) * 8, h->side_ts, cf,
                         (const FilterItem *)h->flt_items, (const double *)h->flt_mats, h->flt_nitems);
      mark_on(h, "filt_ts", 3);
      if (!h->ev_ts_filt) HIPCHK(hipEventCreateWithFlags(&h->ev_ts_filt, hipEventDisableTiming));
      HIPCHK(hipEventRecord(h->ev_ts_filt, h->side_ts));
      h->ts_filtered = true;
    }
    if (h->ts_host) {   // the resident overlay wants T,S of t(tau+1) as soon as they are final
      HIPCHK(hipMemcpyAsync(h->ts_host, c.t_taup1, (size_t)2 * c.imt * c.km * c.jmt * 8, hipMemcpyDeviceToHost, h->side_ts));
      HIPCHK(hipEventRecord(h->ev_ts_host, h->side_ts));
      h->ts_host_queued = true;
      ovl_stamp("ts_d2h");
    }
    h->ts_ahead = true;
    h->ev_ts_final = h->ts_filtered ? h->ev_ts_filt : h->ev_ts_done;
    h->ts_final_valid = true;
    // the other tracers on the main stream: work arrays are indexed from the group's first tracer
    uvic_ctx cr = c;
    cr.n0 = 2; cr.nt_local = c.nt - 2;
    cr.fny = c.fny + 2 * N3;
    ColGrid br = b;
    br.total = br.nwaves * cr.nt_local;
    if (int rc = land_clean(h, c, h->stream)) return rc;
    if (int rc = inputs_rest(h)) return rc;   // the fluxes of the other tracers (pass A applies them at the surface and the bottom)
    mark(h, "begin");
    launch_a(cr, a, S + 2 * N3, h->stream);
    mark(h, "colfct");
    if (h->src_from_prefetch) {
      HIPCHK(hipStreamWaitEvent(h->stream, h->ev_src_ready, 0));
      h->src_from_prefetch = false;
    }
#if defined(UVIC_EXPERIMENTS)
    if (const char *e = uv_env("UVIC_STALL_MAIN_US"))   // stress: the other tracers fall far behind T and S
      hipLaunchKernelGGL(k_stall, dim3(1), dim3(1), 0, h->stream, (long long)atoi(e) * 100);
#endif
    launch_b(cr, br, (const double *)(S + 2 * N3), h->stream);
    if (h->tsi_step) { mark(h, "colupd"); if (int rc = launch_tsi_rows(h, cr, h->stream, 0)) return rc; }
  } else {
    if (int rc = land_clean(h, c, h->stream)) return rc;
    if (int rc = inputs_rest(h)) return rc;
    mark(h, "begin");
    // the local tracers among T and S take their own kernels first (bit-exact in the production step), the others the bulk passes
    const int n_ts = c.n0 < 2 ? std::min(c.n0 + c.nt_local, 2) - c.n0 : 0;
    if (n_ts > 0) {
      uvic_ctx cts = c;
      cts.nt_local = n_ts;
      if (h->ts_exact && h->ts_rows) { if (int rc = launch_rows(h, cts, h->stream, 0, nullptr, "fct_rows_ts", "update_rows_ts")) return rc; }
      else if (h->ts_exact) { if (int rc = launch_colx(h, cts, a, b, h->stream, 0, false)) return rc; }
      else {
        if (h->src_from_prefetch) {   // (T or S with a source term: pass B reads it)
          HIPCHK(hipStreamWaitEvent(h->stream, h->ev_src_ready, 0));
          h->src_from_prefetch = false;
        }
        launch_ts_columns(cts, h->stream, 0, false);
      }
    }
    uvic_ctx cr = c;
    cr.n0 = c.n0 + n_ts; cr.nt_local = c.nt_local - n_ts;
    cr.fny = c.fny + (size_t)n_ts * N3;
    ColGrid br = b;
    br.total = br.nwaves * cr.nt_local;
    launch_a(cr, a, S + (size_t)n_ts * N3, h->stream);
    mark(h, "colfct");
    if (h->src_from_prefetch) {
      HIPCHK(hipStreamWaitEvent(h->stream, h->ev_src_ready, 0));
      h->src_from_prefetch = false;
    }
    launch_b(cr, br, (const double *)(S + (size_t)n_ts * N3), h->stream);
    if (h->tsi_step) { mark(h, "colupd"); if (int rc = launch_tsi_rows(h, c, h->stream, 0)) return rc; }
  }
  mark(h, "colupd");
  HIPCHK(hipGetLastError());
  return 0;
}
// the two halves of the convection diagnostics of a time-average step, around the T,S walk on stream `st`
static int launch_conv_pe(uvic_gpu *h, const uvic_ctx &c, hipStream_t st, int phase) {
  if (!h->tavg_step) return 0;
  const int n = (c.imt - 2) * (c.je - c.js + 1);
  hipLaunchKernelGGL(k_conv_pe, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, c, (const double *)h->tavg_zt, h->tavg_grav, h->tavg_diag, phase);
  HIPCHK(hipGetLastError());
  return 0;
}
static int launch_convect(uvic_gpu *h) {
  mark(h, "begin");
  if (h->exact_convect) {
    if (int rc = launch_conv_pe(h, h->ctx, h->stream, 0)) return rc;
    hipLaunchKernelGGL(k_convect, dim3(col_blocks(h, 128)), dim3(128), 0, h->stream, h->ctx);
    mark(h, "convect");
    if (int rc = launch_conv_pe(h, h->ctx, h->stream, 1)) return rc;
  } else {
    const WetCols w = wet_range(h, h->ctx.js, h->ctx.je);
    const bool fused = h->ts_ahead;   // the T,S walk has run on the side stream (launch_transport): its mixed ranges are replayed here
    h->ts_ahead = false;
    if (fused) {
      HIPCHK(hipStreamWaitEvent(h->stream, h->ev_ts_done, 0));
      if (w.count > 0 && h->d.nt > 2)   // over the list of columns the walk mixed (few): 64 workgroups stride over it
        hipLaunchKernelGGL(k_convect_apply_list, dim3(64), dim3(256), 0, h->stream, h->ctx, (const int *)h->cv_list, w.count);
      mark(h, "convect_apply");
    }
    if (!fused) {
      if (int rc = launch_conv_pe(h, h->ctx, h->stream, 0)) return rc;
      if (w.count > 0)
        hipLaunchKernelGGL(k_convect_ts, dim3((unsigned)((w.count + 63) / 64)), dim3(64), (size_t)2 * h->d.km * 64 * 8 + (size_t)12 * h->d.km * 8, h->stream, h->ctx, w, (int *)nullptr);
      mark(h, "convect_ts");
      if (int rc = launch_conv_pe(h, h->ctx, h->stream, 1)) return rc;
    }
    if (h->d.nt > 2 && !fused) {
      const long long n = (long long)w.count * (h->d.nt - 2);
      if (n > 0) hipLaunchKernelGGL(k_convect_apply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->ctx, w);
      mark(h, "convect_apply");
    }
  }
  if (h->flt_nitems > 0) {   // filt follows convection inside `tracer` (tracer.F:1245); like convection it
    uvic_ctx cf = h->ctx;    // runs on every tracer (under tracer sharding: replicated, after the exchange)
    cf.n0 = 0; cf.nt_local = h->d.nt;
    if (h->ts_filtered) {    // T and S had theirs on the T,S stream (launch_transport); the step ends behind it
      HIPCHK(hipStreamWaitEvent(h->stream, h->ev_ts_filt, 0));
      cf.n0 = 2; cf.nt_local = h->d.nt - 2;
      h->ts_filtered = false;
    }
    if (cf.nt_local > 0)
    hipLaunchKernelGGL(k_filt, dim3((unsigned)h->flt_nitems * (unsigned)cf.nt_local), dim3(h->flt_threads),
                       (size_t)(2 * h->flt_threads + 4) * 8, h->stream, cf, (const FilterItem *)h->flt_items,
                       (const double *)h->flt_mats, h->flt_nitems);
    mark(h, "filt");
  }
  if (h->tavg_step && h->tavg_ic14 > 0 && h->tavg_idic > 0) {   // the delta-14C field of the final t(tau+1), tracer.F:1329-1340
    hipLaunchKernelGGL(k_dc14_field, dim3(cell_blocks(h, 256)), dim3(256), 0, h->stream, h->ctx, h->tavg_ic14, h->tavg_idic, UV_RC14STD, h->tavg_dc14);
    mark(h, "dc14_field");
  }
  if (h->tsi_step && h->tsi_ic14 > 0 && h->tsi_idic > 0) {   // delta 14C of the final t(tau+1), tracer.F:1329-1353
    const size_t NA = (size_t)(h->d.km + 1) * h->d.nt * h->d.jmt;
    const int nrows = h->d.km * (h->ctx.je - h->ctx.js + 1);
    hipLaunchKernelGGL(k_tsi_dc14, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, h->stream, h->ctx, h->tsi_ic14, h->tsi_idic, UV_RC14STD,
                       h->tsi_acc + 3 * NA);
    mark(h, "tsi_dc14");
  }
  if (h->tsi_step && h->tsi_host) {
    // the integrals go to page-locked memory behind the step's last kernel (the T,S rows were summed on the T,S stream,
    // which the main stream has waited for before it replayed the convective ranges): uvic_gpu_tsi_read waits for this copy
    // alone, not for the look-ahead chains
    const size_t NA = (size_t)(h->d.km + 1) * h->d.nt * h->d.jmt;
    if (!h->ts_ahead) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_ts_done, 0));
    HIPCHK(hipMemcpyAsync(h->tsi_host, h->tsi_acc, (3 * NA + (size_t)h->d.km * h->d.jmt) * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipEventRecord(h->ev_tsi, h->stream));
    h->tsi_inflight = true;
  }
  HIPCHK(hipGetLastError());
  return 0;
}
// per-step scalars, updates/09/source/mom/tracer.F:311-343
static int mobi_step_scalars(uvic_gpu *h, double c2dtts, double relyr, mobi_step &S) {
  if (c2dtts == 0.0) return fail_msg("uvic_gpu_mobi: c2dtts not set (uvic_gpu_set_params)");
  S.nbio = (int)(c2dtts / h->mobi_dtnpzd);
  if (S.nbio < 1) return fail_msg("uvic_gpu_mobi: c2dtts/dtnpzd < 1");
  S.dtbio = c2dtts / S.nbio;
  S.rdtts = 1. / c2dtts;
  S.rnbio = 1. / S.nbio;
  const double yrtime = fmod(relyr, 1.);
  S.month = 12;
  for (int m = 1; m <= 12; ++m)
    if (yrtime <= m / 12.) { S.month = m; break; }
  S.declin = sin((fmod(relyr, 1.) - 0.22) * 2. * h->mobi.pi) * 0.4;
  return 0;
}
// the three MOBI passes over the ocean columns of the slab, on stream `st` (profile list `sid`)
static int launch_mobi_on(uvic_gpu *h, const uvic_ctx &c, const mobi_dev &m, hipStream_t st, int sid) {
  const WetCols w = wet_range(h, c.js, c.je);
  if (int rc = src_clean(h, (void *)c.src, st)) return rc;
  mark_on(h, "begin", sid);
  const unsigned cells = (unsigned)(((long long)w.count * c.km + 127) / 128), cols = (unsigned)((w.count + 63) / 64);
  if (m.O) {   // an option set other than C: the general column kernel does all of it
    if (w.count > 0) {
      hipLaunchKernelGGL(k_mobi_gen_pre, dim3(cells), dim3(128), 0, st, c, m, w);   // carbonate chemistry, cell-parallel
      mark_on(h, "mobi_pre", sid);
      if (h->mobi_team && (h->mobi_key == 4 || h->mobi_key == 15)) {   // sets F and run/mk.in's: four-wave teams, then the cell pass
        const size_t lds = UV_MOBIGT_LDS_DOUBLES * 8;
        if (h->mobi_key == 4) {
          hipLaunchKernelGGL((k_mobi_gteam<0, 0, 1, 0>), dim3(cols), dim3(64, 4), lds, st, c, m, w);
          mark_on(h, "mobi", sid);
          hipLaunchKernelGGL((k_mobi_gpost<0, 0, 1, 0>), dim3(cells), dim3(128), 0, st, c, m, w);
        } else {
          hipLaunchKernelGGL((k_mobi_gteam<1, 1, 1, 1>), dim3(cols), dim3(64, 4), lds, st, c, m, w);
          mark_on(h, "mobi", sid);
          hipLaunchKernelGGL((k_mobi_gpost<1, 1, 1, 1>), dim3(cells), dim3(128), 0, st, c, m, w);
        }
        mark_on(h, "mobi_post", sid);
        HIPCHK(hipGetLastError());
        return 0;
      }
      if (h->mobi_key == 4) hipLaunchKernelGGL((k_mobi_gen<0, 0, 1, 0>), dim3(cols), dim3(64), 0, st, c, m, w);         // set F
      else if (h->mobi_key == 15) hipLaunchKernelGGL((k_mobi_gen<1, 1, 1, 1>), dim3(cols), dim3(64), 0, st, c, m, w);   // run/mk.in's set
      else if (h->mobi_key == 3) hipLaunchKernelGGL((k_mobi_gen<1, 1, 0, 0>), dim3(cols), dim3(64), 0, st, c, m, w);    // set C (cross-check)
      else hipLaunchKernelGGL((k_mobi_gen<-1, -1, -1, -1>), dim3(cols), dim3(64), 0, st, c, m, w);
    }
    mark_on(h, "mobi_gen", sid);
    HIPCHK(hipGetLastError());
    return 0;
  }
  if (w.count > 0) hipLaunchKernelGGL(k_mobi_pre, dim3(2 * cells), dim3(128), 0, st, c, m, w);   // two threads per cell
  mark_on(h, "mobi_pre", sid);
  if (w.count > 0) {
    if (h->mobi_team)
      hipLaunchKernelGGL(k_mobi_team, dim3(cols), dim3(64, 4), UV_MOBI_LDS_DOUBLES * 8, st, c, m, w);
    else
      hipLaunchKernelGGL(k_mobi, dim3(cols), dim3(64), 0, st, c, m, w);
  }
  mark_on(h, "mobi", sid);
  if (w.count > 0) hipLaunchKernelGGL(k_mobi_post, dim3(cells), dim3(128), 0, st, c, m, w);
  mark_on(h, "mobi_post", sid);
  HIPCHK(hipGetLastError());
  return 0;
}
// at the head of a MOBI chain on stream `st`: the segment's forcing fields are there, or on their way on another stream
static int forcing_on(uvic_gpu *h, hipStream_t st) {
  auto &I = h->in;
  if (I.forcing_pull) {
    Pull4 p;
    for (int q = 0; q < 4; ++q) { p.src[q] = I.forcing_src[q]; p.dst[q] = (double *)h->mobi_st.f[1 + q]; }
    const int n = h->d.imt * h->d.jmt;
    hipLaunchKernelGGL(k_pull4, dim3((unsigned)((n + 255) / 256), 4), dim3(256), 0, st, p, n);
    HIPCHK(hipEventRecord(I.ev_forcing, st));
    I.forcing_pull = false; I.forcing_sent = true; I.forcing_inflight = true;
    return 0;
  }
  if (I.forcing_sent) HIPCHK(hipStreamWaitEvent(st, I.ev_forcing, 0));
  return 0;
}
static int launch_mobi(uvic_gpu *h) {
  if (!h->have_mobi) return 0;
  const int par = (int)(h->step_no & 1);
  h->mobi_ev_step[par] = -1;
  if (h->src_from_prefetch) {   // computed one step ahead on the side stream; launch_transport waits for it
    // The chain assumed a leapfrog step with c2dtts_next and named a clock and a CO2 value.  The bit-exact path takes its
    // sources only if all of that came true to the bit; the production path also when the clock it was promised differs
    // by rounding (a caller that extrapolates relyr), as long as the month of the dust field is the same.
    auto month_of = [](double relyr) { const double y = fmod(relyr, 1.); int mo = 12; for (int q = 1; q <= 12; ++q) if (y <= q / 12.) { mo = q; break; } return mo; };
    const bool clock_ok = h->src_relyr == h->mobi.relyr ||
                          (!h->exact && fabs(h->src_relyr - h->mobi.relyr) <= 1e-9 && month_of(h->src_relyr) == month_of(h->mobi.relyr));
    if (!h->mixing && h->src_c2dtts == h->ctx.c2dtts && clock_ok && h->src_co2ccn == h->mobi.co2ccn) {
      h->ev_mobi_of[par] = h->ev_src_ready; h->mobi_ev_step[par] = h->step_no;
      return 0;
    }
    // otherwise its sources are void: recompute in line, behind the chain (it wrote the buffer this step reads)
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_src_ready, 0));
    h->src_from_prefetch = false;
  }
  if (int rc = mobi_step_scalars(h, h->ctx.c2dtts, h->mobi.relyr, h->mobi.S)) return rc;
  if (h->step_begun && !h->serial) {
    // Not computed ahead (the first step of an ocean segment, a forward step): on a MOBI side stream all the same.  Neither
    // pass A nor the T,S passes read the sources; pass B of the other tracers waits for them as it does for a chain that
    // ran ahead.  The buffer was last read by a pass B before this step's begin.
    // (the chain that computes the NEXT step's sources, queued later in this step, takes the same stream: side by side the
    // two would share the chip and this one, which pass B is waiting for, would take 100 us longer)
    const int q = h->mobi_flip;
    hipStream_t st = h->side_m[q];
    HIPCHK(hipStreamWaitEvent(st, h->ev_begin_cur, 0));
    if (int rc = forcing_on(h, st)) return rc;
    if (int rc = launch_mobi_on(h, h->ctx, h->mobi, st, q ? 4 : 1)) return rc;
    HIPCHK(hipEventRecord(h->ev_src_inline2[par], st));
    h->ev_src_ready = h->ev_src_inline2[par];
    h->src_from_prefetch = true;
    h->ev_mobi_of[par] = h->ev_src_ready; h->mobi_ev_step[par] = h->step_no;
    return 0;
  }
  if (int rc = forcing_on(h, h->stream)) return rc;
  return launch_mobi_on(h, h->ctx, h->mobi, h->stream, 0);
}

// -- producers of the shared inputs (kernels_prep.hpp) ---------------------------------
static int mom_join(uvic_gpu *h);
static int launch_adv_vel(uvic_gpu *h) {
  if (int rc = mom_join(h)) return rc;     // a momentum step beside the main stream still reads the old ones
  velocity_touched(h, UVIC_F_ADV_VET);   // a look-ahead chain formed its total velocities from the old ones: redo them
  mark(h, "begin");
  hipLaunchKernelGGL(k_adv_vel_hor, dim3(cell_blocks(h, 256)), dim3(256), 0, h->stream, h->ctx);
  mark(h, "adv_vel_hor");
  hipLaunchKernelGGL(k_adv_vel_vert, dim3(col_blocks(h, 64)), dim3(64), 0, h->stream, h->ctx);
  mark(h, "adv_vel_vert");
  HIPCHK(hipGetLastError());
  return 0;
}
// the exponentials of vmixc.F:103-105 for every level pair, by the host's exp (synchronises the main stream once)
static int vmix_tables(uvic_gpu *h) {
  if (h->vmix_tab_ready) return 0;
  const int km = h->d.km;
  std::vector<double> zw(km), tab((size_t)km * km + km, 0.0);
  HIPCHK(hipMemcpyAsync(zw.data(), h->buf[UVIC_F_ZW], (size_t)km * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  const double zetar = h->ctx.zetar;
  for (int k = 1; k <= km; ++k)
    for (int k1 = k + 1; k1 <= km; ++k1) {
      const double hab = zw[k - 1] - zw[k1 - 1];
      tab[(size_t)(k - 1) * km + k1 - 1] = exp(hab * zetar);
    }
  for (int k1 = 1; k1 <= km; ++k1) tab[(size_t)km * km + k1 - 1] = 1 - exp(-zetar * zw[k1 - 1]);
  if (!h->vmix_tab) HIPCHK(hipMalloc((void **)&h->vmix_tab, tab.size() * 8));
  HIPCHK(hipMemcpyAsync(h->vmix_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->ctx.vmix_e = h->vmix_tab;
  h->ctx.vmix_d = h->vmix_tab + (size_t)km * km;
  h->vmix_tab_ready = true;
  return 0;
}
static int launch_vmixc(uvic_gpu *h) {
  if (!h->have_vmix) return fail_msg("uvic_gpu_vmixc: call uvic_gpu_set_vmix_params first");
  if (!h->ctx.diff_cbt_given) return fail_msg("uvic_gpu_vmixc: set uvic_params.diff_cbt_has_k33 = 1 (isopyc must leave diff_cbt to vmixc)");
  if (int rc = vmix_tables(h)) return rc;
  mark(h, "begin");
  hipLaunchKernelGGL(k_vmixc, dim3(cell_blocks(h, 256)), dim3(256), 0, h->stream, h->ctx);
  mark(h, "vmixc");
  if (!h->exact) {   // the folded vertical-diffusion coefficient follows the new diff_cbt
    hipLaunchKernelGGL(k_coef_bv, dim3(cell_blocks(h, 256)), dim3(256), 0, h->stream, h->ctx, h->coef);
    mark(h, "coef_bv");
  }
  HIPCHK(hipGetLastError());
  return 0;
}
// polar Fourier filter of t(tau+1): strips and operators from kmt and the grid already uploaded
extern "C" int uvic_gpu_set_filter(uvic_gpu *h, double pi, int jfrst, int jft0, int jft1, int jft2, int lsegf) {
  if (!h) return fail_msg("null handle");
  HIPCHK(hipSetDevice(h->device));
  (void)hipFree(h->flt_items); h->flt_items = nullptr;
  (void)hipFree(h->flt_mats); h->flt_mats = nullptr;
  h->flt_nitems = 0;
  if (jfrst > h->d.jmt) return 0;          // `jrow .lt. jfrst` for every row: filter off
  if (lsegf < 1) return fail_msg("uvic_gpu_set_filter: lsegf < 1");
  const uvic_dims &d = h->d;
  std::vector<int> kmt((size_t)d.imt * d.jmt);
  std::vector<double> cst(d.jmt), cstr(d.jmt);
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(kmt.data(), h->buf[UVIC_F_KMT], kmt.size() * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(cst.data(), h->buf[UVIC_F_CST], cst.size() * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(cstr.data(), h->buf[UVIC_F_CSTR], cstr.size() * 8, hipMemcpyDeviceToHost));
  FilterSetup fs;
  std::string err;
  if (int rc = filter_build(d.imt, d.jmt, d.km, kmt.data(), cst.data(), cstr.data(), pi, jfrst, jft0, jft1, jft2, lsegf, fs, err)) {
    g_err = err;
    return rc;
  }
  if (fs.items.empty()) return 0;
  int maxim = 0;
  for (auto &it : fs.items) maxim = it.im > maxim ? it.im : maxim;
  if (maxim > 1024) return fail_msg("uvic_gpu_set_filter: strips longer than 1024 columns are not supported");
  h->flt_threads = ((maxim + 63) / 64) * 64;
  HIPCHK(hipMalloc((void **)&h->flt_items, fs.items.size() * sizeof(FilterItem)));
  HIPCHK(hipMemcpy(h->flt_items, fs.items.data(), fs.items.size() * sizeof(FilterItem), hipMemcpyHostToDevice));
  const size_t nm = fs.mats.empty() ? 1 : fs.mats.size();
  HIPCHK(hipMalloc((void **)&h->flt_mats, nm * 8));
  if (!fs.mats.empty()) HIPCHK(hipMemcpy(h->flt_mats, fs.mats.data(), fs.mats.size() * 8, hipMemcpyHostToDevice));
  h->flt_nitems = (int)fs.items.size();
  return 0;
}

extern "C" int uvic_gpu_adv_vel(uvic_gpu *h) {
  if (!h) return fail_msg("null handle");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = launch_adv_vel(h)) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int uvic_gpu_adv_vel_async(uvic_gpu *h) {
  if (!h) return fail_msg("null handle");
  HIPCHK(hipSetDevice(h->device));
  return launch_adv_vel(h);
}
extern "C" int uvic_gpu_set_vmix_params(uvic_gpu *h, const uvic_vmix_params *p) {
  if (!h || !p) return fail_msg("uvic_gpu_set_vmix_params: null argument");
  h->ctx.kappa_h = p->kappa_h; h->ctx.zetar = p->zetar; h->ctx.ogamma = p->ogamma; h->ctx.gravrho0r = p->gravrho0r;
  h->have_vmix = true;
  h->vmix_tab_ready = false;
  return 0;
}
extern "C" int uvic_gpu_vmixc(uvic_gpu *h) {
  if (!h) return fail_msg("null handle");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = launch_vmixc(h)) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

// -- O_TMM column-batch source operator: host-side packing around the calls above ------------------------------
extern "C" int uvic_gpu_tmm_create(uvic_gpu **out, int ncols, int km, int nt, int nsrc, int ntnpzd, int device) {
  if (ncols < 4) return fail_msg("uvic_gpu_tmm_create: a batch needs at least 4 columns");
  uvic_dims d;
  d.imt = ncols + 2; d.jmt = 6; d.km = km; d.nt = nt; d.nsrc = nsrc; d.ntnpzd = ntnpzd;
  if (int rc = uvic_gpu_create(out, &d, device)) return rc;
  (*out)->tmm_ncols = ncols;
  return 0;
}
// per-column array with `x` values per column, (ncols, x) -> the (imt, jmt, x) array of the handle's grid, row 2
static std::vector<double> tmm_pad(const uvic_gpu *h, const double *cols, int x) {
  const int imt = h->d.imt, jmt = h->d.jmt, nc = h->tmm_ncols;
  std::vector<double> a((size_t)imt * jmt * x, 0.0);
  if (cols)
    for (int q = 0; q < x; ++q)
      for (int c = 0; c < nc; ++c) a[(size_t)(c + 1) + (size_t)imt * (1 + (size_t)jmt * q)] = cols[(size_t)c + (size_t)nc * q];
  return a;
}
extern "C" int uvic_gpu_tmm_set_mobi(uvic_gpu *h, const int32_t *kmt, const uvic_mobi_params *p, const uvic_mobi_options *o,
                                     const uvic_mobi_forcing *f) {
  if (!h || !kmt || !p || !f) return fail_msg("uvic_gpu_tmm_set_mobi: null argument");
  if (h->tmm_ncols <= 0) return fail_msg("uvic_gpu_tmm_set_mobi: not a column-batch handle (uvic_gpu_tmm_create)");
  const int imt = h->d.imt, jmt = h->d.jmt, km = h->d.km, nc = h->tmm_ncols;
  std::vector<int32_t> k2((size_t)imt * jmt, 0);
  for (int c = 0; c < nc; ++c) {
    if (kmt[c] < 0 || kmt[c] > km) return fail_msg("uvic_gpu_tmm_set_mobi: kmt out of range");
    k2[(size_t)(c + 1) + (size_t)imt] = kmt[c];
  }
  if (int rc = uvic_gpu_upload(h, UVIC_F_KMT, k2.data(), 0, (int64_t)k2.size())) return rc;
  std::vector<int32_t> itrc((size_t)h->d.nt, 0);   // sources are read where they lie: no tracer takes one here
  if (int rc = uvic_gpu_upload(h, UVIC_F_ITRC, itrc.data(), 0, (int64_t)itrc.size())) return rc;
  const std::vector<double> tlat = tmm_pad(h, f->tlat, 1), dnswr = tmm_pad(h, f->dnswr, 1), aice = tmm_pad(h, f->aice, 1),
                            hice = tmm_pad(h, f->hice, 1), hsno = tmm_pad(h, f->hsno, 1), sgb = tmm_pad(h, f->sg_bathy, km),
                            dep = tmm_pad(h, f->fe_atmdep, 12), hyd = tmm_pad(h, f->fe_hydr, km);
  uvic_mobi_forcing g = *f;
  g.tlat = tlat.data(); g.dnswr = dnswr.data(); g.aice = aice.data(); g.hice = hice.data(); g.hsno = hsno.data();
  g.sg_bathy = sgb.data(); g.fe_atmdep = dep.data(); g.fe_hydr = hyd.data();
  return o ? uvic_gpu_set_mobi_opt(h, p, o, &g) : uvic_gpu_set_mobi(h, p, &g);
}
extern "C" int uvic_gpu_tmm_sources(uvic_gpu *h, double c2dtts, double relyr, double co2ccn, const double *t_taum1, const double *dnswr,
                                    const double *aice, const double *hice, const double *hsno, double *src) {
  if (!h || !t_taum1 || !src) return fail_msg("uvic_gpu_tmm_sources: null argument");
  if (h->tmm_ncols <= 0) return fail_msg("uvic_gpu_tmm_sources: not a column-batch handle (uvic_gpu_tmm_create)");
  if (!h->have_mobi) return fail_msg("uvic_gpu_tmm_sources: call uvic_gpu_tmm_set_mobi first");
  const int imt = h->d.imt, km = h->d.km, nt = h->d.nt, nsrc = h->d.nsrc, nc = h->tmm_ncols;
  // t(1:ncols,:,1,:,taum1) -> row 2 of the handle's grid, one column of padding either side
  std::vector<double> st((size_t)imt * km * (nt > nsrc ? nt : nsrc), 0.0);
  for (int n = 0; n < nt; ++n)
    for (int k = 0; k < km; ++k)
      memcpy(&st[(size_t)1 + (size_t)imt * (k + (size_t)km * n)], t_taum1 + (size_t)nc * (k + (size_t)km * n), (size_t)nc * 8);
  if (int rc = uvic_gpu_upload_rows(h, UVIC_F_T_TAUM1, st.data(), 2, 2)) return rc;
  h->ctx.c2dtts = c2dtts;
  if (dnswr || aice || hice || hsno) {
    if (!(dnswr && aice && hice && hsno)) return fail_msg("uvic_gpu_tmm_sources: give all four forcing fields or none");
    const std::vector<double> a = tmm_pad(h, dnswr, 1), b = tmm_pad(h, aice, 1), c = tmm_pad(h, hice, 1), d = tmm_pad(h, hsno, 1);
    if (int rc = uvic_gpu_set_mobi_step(h, relyr, co2ccn, a.data(), b.data(), c.data(), d.data())) return rc;
  } else if (int rc = uvic_gpu_set_mobi_step(h, relyr, co2ccn, nullptr, nullptr, nullptr, nullptr)) {
    return rc;
  }
  HIPCHK(hipSetDevice(h->device));
  if (int rc = launch_mobi(h)) return rc;
  if (int rc = uvic_gpu_download_rows(h, UVIC_F_SRC, st.data(), 2, 2)) return rc;
  for (int n = 0; n < nsrc; ++n)
    for (int k = 0; k < km; ++k)
      memcpy(src + (size_t)nc * (k + (size_t)km * n), &st[(size_t)1 + (size_t)imt * (k + (size_t)km * n)], (size_t)nc * 8);
  return 0;
}

// -- baroclinic momentum step ---------------------------------------------------------------
static uvic_mom_ctx mom_ctx(uvic_gpu *h) {
  uvic_mom_ctx m;
  memset(&m, 0, sizeof m);
  const uvic_dims &d = h->d;
  m.imt = d.imt; m.jmt = d.jmt; m.km = d.km;
  m.js = h->ctx.js; m.je = h->ctx.je;
  m.c2dtuv = h->clinic_p.c2dtuv; m.grav_rho0r = h->clinic_p.grav * h->clinic_p.rho0r;
  m.kappa_m = h->clinic_p.kappa_m; m.cdbot = h->clinic_p.cdbot;
#define B(field, F) m.field = (decltype(m.field))h->buf[F]
  B(dxur, UVIC_F_DXUR); B(dxu2r, UVIC_F_DXU2R); B(dxtr, UVIC_F_DXTR); B(dxmetr, UVIC_F_DXMETR); B(duw, UVIC_F_DUW); B(due, UVIC_F_DUE);
  B(dyur, UVIC_F_DYUR); B(dyu2r, UVIC_F_DYU2R); B(dyu4r, UVIC_F_DYU4R); B(dytr, UVIC_F_DYTR); B(csur, UVIC_F_CSUR); B(cst, UVIC_F_CST);
  B(dus, UVIC_F_DUS); B(dun, UVIC_F_DUN); B(csudyu2r, UVIC_F_CSUDYU2R);
  B(advmet, UVIC_F_ADVMET); B(am3, UVIC_F_AM3); B(am4, UVIC_F_AM4);
  B(dzt, UVIC_F_DZT); B(dztr, UVIC_F_DZTR); B(dzt2r, UVIC_F_DZT2R); B(dzw, UVIC_F_DZW); B(dzwr, UVIC_F_DZWR);
  B(to, UVIC_F_TO); B(so, UVIC_F_SO); B(c, UVIC_F_C);
  B(kmt, UVIC_F_KMT); B(kmu, UVIC_F_KMU); B(hr, UVIC_F_HR); B(cori, UVIC_F_CORI);
  B(visc_ceu, UVIC_F_VISC_CEU); B(amc_north, UVIC_F_AMC_NORTH); B(amc_south, UVIC_F_AMC_SOUTH);
  B(adv_vet, UVIC_F_ADV_VET); B(adv_vnt, UVIC_F_ADV_VNT); B(adv_vbt, UVIC_F_ADV_VBT);
  B(smf, UVIC_F_SMF); B(rho, UVIC_F_RHO);
  B(ut1, UVIC_F_U1); B(ut2, UVIC_F_U2); B(um1, UVIC_F_UM1); B(um2, UVIC_F_UM2); B(up1, UVIC_F_UP1); B(up2, UVIC_F_UP2);
  B(zu, UVIC_F_ZU); B(grad_p, UVIC_F_GRAD_P);
  B(sbc_gu, UVIC_F_SBC_GU); B(sbc_gv, UVIC_F_SBC_GV); B(sbc_su, UVIC_F_SBC_SU); B(sbc_sv, UVIC_F_SBC_SV);
#undef B
  // T and S of t(tau): the slots rotate by pointer, the tracer context knows the current one
  m.t_tau = h->ctx.t_tau;
  m.s_tau = h->ctx.t_tau + (size_t)d.imt * d.km * d.jmt;
  return m;
}
// `st`: the main stream (marks for the profile) or the momentum stream (uvic_gpu_momentum_async)
static int launch_state(uvic_gpu *h, hipStream_t st) {
  if (h->ctx.n0 != 0) return fail_msg("uvic_gpu_state: this rank's tracer shard does not hold T and S");
  const uvic_mom_ctx m = mom_ctx(h);
  const bool mk = st == h->stream;
  if (mk) mark(h, "begin");
  hipLaunchKernelGGL(k_state, dim3(cell_blocks(h, 256)), dim3(256), 0, st, m);
  if (mk) mark(h, "state");
  HIPCHK(hipGetLastError());
  return 0;
}
static int launch_filuv(uvic_gpu *h, const uvic_mom_ctx &m, hipStream_t st);
static int launch_clinic(uvic_gpu *h, int sbc_flags, double rts, hipStream_t st) {
  if (!h->have_clinic) return fail_msg("uvic_gpu_clinic: call uvic_gpu_set_clinic_params first");
  const uvic_mom_ctx m = mom_ctx(h);
  const bool mk = st == h->stream;
  if (mk) mark(h, "begin");
  hipLaunchKernelGGL(k_clinic_gradp, dim3(col_blocks(h, 64)), dim3(64), 0, st, m);
  if (mk) mark(h, "clinic_gradp");
  hipLaunchKernelGGL(k_clinic_tend, dim3(cell_blocks(h, 256)), dim3(256), 0, st, m);
  if (mk) mark(h, "clinic_tend");
  hipLaunchKernelGGL(k_clinic_finish, dim3(col_blocks(h, 64)), dim3(64), 0, st, m, sbc_flags, rts);
  if (mk) mark(h, "clinic_finish");
  if (int rc = launch_filuv(h, m, st)) return rc;
  HIPCHK(hipGetLastError());
  return 0;
}
static int launch_filuv(uvic_gpu *h, const uvic_mom_ctx &m, hipStream_t st) {
  if (h->fltu_nitems == 0) return 0;
  const bool mk = st == h->stream;
  // 2*H threads per strip (H = strip length rounded up to a wave): the two components are filtered side by side
  hipLaunchKernelGGL(k_filuv, dim3((unsigned)h->fltu_nitems), dim3(2 * h->fltu_threads), (size_t)(4 * h->fltu_threads + 8) * 8, st, m,
                     h->fltu_items, h->fltu_mats, (const double *)h->buf[UVIC_F_SPSIN], (const double *)h->buf[UVIC_F_SPCOS],
                     h->fltu_nitems);
  if (mk) mark(h, "filuv");
  const unsigned n = (unsigned)h->fltu_nrows * (unsigned)h->d.imt;
  hipLaunchKernelGGL(k_filuv_mean, dim3((n + 63) / 64), dim3(64), 0, st, m, h->fltu_rows, h->fltu_nrows);
  if (mk) mark(h, "filuv_mean");
  return 0;
}
// what the main stream queues next on u or the advective velocities waits for a momentum step still running beside it
static int mom_join(uvic_gpu *h) {
  if (h->mom_pending) {
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_mom_done, 0));
    h->mom_pending = false;
  }
  return 0;
}
extern "C" int uvic_gpu_rotate_u(uvic_gpu *h) {
  if (!h) return fail_msg("null handle");
  HIPCHK(hipSetDevice(h->device));
  // (pointer rotation only: kernels already queued hold the old pointers in their arguments)
  void *m1 = h->buf[UVIC_F_UM1], *m2 = h->buf[UVIC_F_UM2];
  h->buf[UVIC_F_UM1] = h->buf[UVIC_F_U1]; h->buf[UVIC_F_UM2] = h->buf[UVIC_F_U2];
  h->buf[UVIC_F_U1] = h->buf[UVIC_F_UP1]; h->buf[UVIC_F_U2] = h->buf[UVIC_F_UP2];
  h->buf[UVIC_F_UP1] = m1; h->buf[UVIC_F_UP2] = m2;
  bind_ctx(h);
  return 0;
}
extern "C" int uvic_gpu_add_ext_mode(uvic_gpu *h, int level) {
  if (!h) return fail_msg("null handle");
  if (level != 0 && level != -1) return fail_msg("uvic_gpu_add_ext_mode: level is 0 (tau) or -1 (tau-1)");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = mom_join(h)) return rc;
  const uvic_mom_ctx m = mom_ctx(h);
  const double *psi = (const double *)h->buf[UVIC_F_PSI] + (level == 0 ? 0 : (size_t)h->d.imt * h->d.jmt);
  double *u1 = (double *)h->buf[level == 0 ? UVIC_F_U1 : UVIC_F_UM1], *u2 = (double *)h->buf[level == 0 ? UVIC_F_U2 : UVIC_F_UM2];
  mark(h, "begin");
  hipLaunchKernelGGL(k_add_ext_mode, dim3(col_blocks(h, 64)), dim3(64), 0, h->stream, m, psi, u1, u2);
  mark(h, "add_ext_mode");
  HIPCHK(hipGetLastError());
  if (h->host_sync) HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int uvic_gpu_set_clinic_params(uvic_gpu *h, const uvic_clinic_params *p) {
  if (!h || !p) return fail_msg("uvic_gpu_set_clinic_params: null argument");
  h->clinic_p = *p;
  h->have_clinic = true;
  return 0;
}
extern "C" int uvic_gpu_state(uvic_gpu *h) {
  if (!h) return fail_msg("null handle");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = mom_join(h)) return rc;
  if (int rc = launch_state(h, h->stream)) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int uvic_gpu_state_async(uvic_gpu *h) {
  if (!h) return fail_msg("null handle");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = mom_join(h)) return rc;
  return launch_state(h, h->stream);
}
extern "C" int uvic_gpu_clinic_async(uvic_gpu *h, int sbc_flags, double rts) {
  if (!h) return fail_msg("null handle");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = mom_join(h)) return rc;
  return launch_clinic(h, sbc_flags, rts, h->stream);
}
// state + clinic of this time step on a stream of their own, beside the tracer step the caller queues next on the main
// stream: `clinic` reads rho (T,S at tau), u and the advective velocities, none of which the tracer step writes.  They start
// behind what the main stream holds now (add_ext_mode, adv_vel); zu is copied to `zu_host` (imt,jmt,2; may be null) and
// uvic_gpu_momentum_wait returns when it has arrived -- long before the tracer step ends, so that the host's `tropic` runs
// beside it.  The next uvic_gpu_add_ext_mode / uvic_gpu_adv_vel / state / clinic on the main stream wait for it by event.
static int momentum_stream(uvic_gpu *h) {
  if (!h->side_mom) {
    HIPCHK(hipStreamCreateWithFlags(&h->side_mom, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&h->ev_mom_in, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_mom_done, hipEventDisableTiming));
  }
  return 0;
}
extern "C" int uvic_gpu_momentum_async(uvic_gpu *h, int sbc_flags, double rts, double *zu_host) {
  if (!h) return fail_msg("null handle");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = momentum_stream(h)) return rc;
  if (int rc = mom_join(h)) return rc;
  HIPCHK(hipEventRecord(h->ev_mom_in, h->stream));
  HIPCHK(hipStreamWaitEvent(h->side_mom, h->ev_mom_in, 0));
  if (int rc = launch_state(h, h->side_mom)) return rc;
  if (int rc = launch_clinic(h, sbc_flags, rts, h->side_mom)) return rc;
  if (zu_host)
    HIPCHK(hipMemcpyAsync(zu_host, h->buf[UVIC_F_ZU], (size_t)2 * h->d.imt * h->d.jmt * 8, hipMemcpyDeviceToHost, h->side_mom));
  HIPCHK(hipEventRecord(h->ev_mom_done, h->side_mom));
  h->mom_pending = true;
  return 0;
}
// For a caller that keeps u on the device (the resident overlays tracer_gpu.F + clinic_gpu.F), on the momentum stream,
// beside whatever the main stream still does; nothing is waited for until uvic_gpu_momentum_wait.
//
// uvic_gpu_overlay_velocities: the start of a leapfrog step, before `tracer` -- what loadmw does to u with the memory
// window wide open (u09/mom/loadmw.F:86-99): the time levels rotate, the external mode of psi(,,1) is added to u(tau) (of
// psi(,,2) to u(tau-1) as well if `ext_taum1`: the first time step); then adv_vel (source/mom/adv_vel.F:63-131) into the
// copy of the step's inputs the tracer step will read: the uvic_gpu_overlay_inputs that follows leaves adv_vet, adv_vnt out.
extern "C" int uvic_gpu_overlay_velocities(uvic_gpu *h, int ext_taum1, const double *psi) {
  if (!h || !psi) return fail_msg("uvic_gpu_overlay_velocities: null argument");
  ovl_stamp("vel_in");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = momentum_stream(h)) return rc;
  if (int rc = mom_host_join(h)) return rc;
  const uvic_dims &d = h->d;
  const size_t N2 = (size_t)d.imt * d.jmt;
  hipStream_t st = h->side_mom;
  if (int rc = uvic_gpu_rotate_u(h)) return rc;
  HIPCHK(hipMemcpyAsync(h->buf[UVIC_F_PSI], psi, 2 * N2 * 8, hipMemcpyHostToDevice, st));
  // the copy of the inputs the coming tracer step reads (it was the copy of the step three back: nothing reads it any more)
  if (int rc = inputs_next_copy(h)) return rc;
  const uvic_mom_ctx m = mom_ctx(h);
  for (int level = 0; level >= (ext_taum1 ? -1 : 0); --level) {
    const double *ps = (const double *)h->buf[UVIC_F_PSI] + (level == 0 ? 0 : N2);
    double *u1 = (double *)h->buf[level == 0 ? UVIC_F_U1 : UVIC_F_UM1], *u2 = (double *)h->buf[level == 0 ? UVIC_F_U2 : UVIC_F_UM2];
    hipLaunchKernelGGL(k_add_ext_mode, dim3(col_blocks(h, 64)), dim3(64), 0, st, m, ps, u1, u2);
  }
  hipLaunchKernelGGL(k_adv_vel_hor, dim3(cell_blocks(h, 256)), dim3(256), 0, st, h->ctx);
  hipLaunchKernelGGL(k_adv_vel_vert, dim3(col_blocks(h, 64)), dim3(64), 0, st, h->ctx);
  HIPCHK(hipEventRecord(h->in.ev_vel, st));
  h->in.vel_pending = true;
  velocity_touched(h, UVIC_F_ADV_VET);
  HIPCHK(hipGetLastError());
  return 0;
}
// uvic_gpu_overlay_momentum: the momentum row itself, where the reference calls `clinic` (after `tracer`, mom.F:389-395) --
// the time-step monitor's kinetic energy of u(tau) (clinic.F:616-630; ektot (0:km, jmt)) if ektot_host is given; `state`
// (loadmw.F:154) from T,S of the level the step began with as t(tau) -- t_level 0: UVIC_F_T_TAU, -1: UVIC_F_T_TAUM1 (the
// tracer step of this time step has rotated the levels already) -- unless rho_host (imt,km,2:jmt) is given; `clinic` with
// sbc_flags/rts as uvic_gpu_clinic; zu (imt,jmt,2) to zu_host.  `fresh`: u(tau), u(tau-1) have just been uploaded on the
// main stream.  u(tau+1) stays on the device (UVIC_F_UP1/UP2).
// smf, rho_host, zu_host, ektot_host: page-locked, left alone until uvic_gpu_momentum_wait returns.
extern "C" int uvic_gpu_overlay_momentum(uvic_gpu *h, int fresh, int t_level, int sbc_flags, double rts, double rho0, const double *smf,
                                         const double *rho_host, double *zu_host, double *ektot_host) {
  if (!h || !smf || !zu_host) return fail_msg("uvic_gpu_overlay_momentum: null argument");
  if (!h->have_clinic) return fail_msg("uvic_gpu_overlay_momentum: call uvic_gpu_set_clinic_params first");
  if (t_level != 0 && t_level != -1) return fail_msg("uvic_gpu_overlay_momentum: t_level is 0 or -1");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = momentum_stream(h)) return rc;
  if (int rc = mom_host_join(h)) return rc;
  const uvic_dims &d = h->d;
  const size_t N2 = (size_t)d.imt * d.jmt, N3 = N2 * d.km, row = (size_t)d.imt * d.km;
  hipStream_t st = h->side_mom;
  if (fresh) {   // the levels (and whatever else the caller uploaded) came up on the main stream just now
    HIPCHK(hipEventRecord(h->ev_mom_in, h->stream));
    HIPCHK(hipStreamWaitEvent(st, h->ev_mom_in, 0));
  }
  HIPCHK(hipMemcpyAsync(h->buf[UVIC_F_SMF], smf, 2 * N2 * 8, hipMemcpyHostToDevice, st));
  if (rho_host) HIPCHK(hipMemcpyAsync((char *)h->buf[UVIC_F_RHO] + row * 8, rho_host, (N3 - row) * 8, hipMemcpyHostToDevice, st));
  if (ektot_host) {
    const size_t n = (size_t)(d.km + 1) * d.jmt;
    if (!h->in.ektot_dev) HIPCHK(hipMalloc((void **)&h->in.ektot_dev, n * 8));
    HIPCHK(hipMemsetAsync(h->in.ektot_dev, 0, n * 8, st));
    const int work = d.km * (h->ctx.je - h->ctx.js + 1);
    hipLaunchKernelGGL(k_tsi_ektot, dim3((unsigned)((work + TSI_ROWS - 1) / TSI_ROWS)), dim3(64), 0, st, h->ctx, (const double *)h->buf[UVIC_F_U1],
                       (const double *)h->buf[UVIC_F_U2], rho0, h->in.ektot_dev);
    HIPCHK(hipMemcpyAsync(ektot_host, h->in.ektot_dev, n * 8, hipMemcpyDeviceToHost, st));
  }
  if (!rho_host) {
    uvic_mom_ctx m = mom_ctx(h);
    if (t_level == -1) { m.t_tau = h->ctx.t_taum1; m.s_tau = h->ctx.t_taum1 + N3; }
    hipLaunchKernelGGL(k_state, dim3(cell_blocks(h, 256)), dim3(256), 0, st, m);
  }
  if (int rc = launch_clinic(h, sbc_flags, rts, st)) return rc;
  HIPCHK(hipMemcpyAsync(zu_host, h->buf[UVIC_F_ZU], 2 * N2 * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipEventRecord(h->ev_mom_done, st));
  h->mom_pending = true;
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int uvic_gpu_momentum_wait(uvic_gpu *h) {
  if (!h) return fail_msg("null handle");
  if (h->side_mom) HIPCHK(hipEventSynchronize(h->ev_mom_done));
  h->mom_pending = false;
  return 0;
}
extern "C" int uvic_gpu_clinic(uvic_gpu *h, int sbc_flags, double rts) {
  if (!h) return fail_msg("null handle");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = mom_join(h)) return rc;
  if (int rc = launch_clinic(h, sbc_flags, rts, h->stream)) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
// polar Fourier filter of u(tau+1): strips of kmu and operators from the fields already uploaded (UVIC_F_KMU, CSU, CSUR, PHI)
extern "C" int uvic_gpu_set_filter_u(uvic_gpu *h, double pi, int jfrst, int jfu0, int jfu1, int jfu2, int lsegf) {
  if (!h) return fail_msg("null handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  (void)hipFree(h->fltu_items); h->fltu_items = nullptr;
  (void)hipFree(h->fltu_mats); h->fltu_mats = nullptr;
  (void)hipFree(h->fltu_rows); h->fltu_rows = nullptr;
  h->fltu_nitems = h->fltu_nrows = 0;
  if (jfrst > h->d.jmt) return 0;          // `jrow .lt. jfrst` for every row: filter off
  if (lsegf < 1) return fail_msg("uvic_gpu_set_filter_u: lsegf < 1");
  const uvic_dims &d = h->d;
  std::vector<int> kmu((size_t)d.imt * d.jmt);
  std::vector<double> csu(d.jmt), csur(d.jmt), phi(d.jmt);
  HIPCHK(hipMemcpy(kmu.data(), h->buf[UVIC_F_KMU], kmu.size() * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(csu.data(), h->buf[UVIC_F_CSU], csu.size() * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(csur.data(), h->buf[UVIC_F_CSUR], csur.size() * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(phi.data(), h->buf[UVIC_F_PHI], phi.size() * 8, hipMemcpyDeviceToHost));
  FilterSetup fs;
  std::vector<int> rows;
  std::string err;
  if (int rc = filter_build_u(d.imt, d.jmt, d.km, kmu.data(), csu.data(), csur.data(), phi.data(), pi, jfrst, jfu0, jfu1, jfu2, lsegf,
                              fs, rows, err)) {
    g_err = err;
    return rc;
  }
  if (fs.items.empty()) return 0;
  int maxim = 0;
  for (auto &it : fs.items) maxim = it.im > maxim ? it.im : maxim;
  if (maxim > 512) return fail_msg("uvic_gpu_set_filter_u: strips longer than 512 columns are not supported");
  h->fltu_threads = ((maxim + 63) / 64) * 64;
  HIPCHK(hipMalloc((void **)&h->fltu_items, fs.items.size() * sizeof(FilterItem)));
  HIPCHK(hipMemcpy(h->fltu_items, fs.items.data(), fs.items.size() * sizeof(FilterItem), hipMemcpyHostToDevice));
  const size_t nm = fs.mats.empty() ? 1 : fs.mats.size();
  HIPCHK(hipMalloc((void **)&h->fltu_mats, nm * 8));
  if (!fs.mats.empty()) HIPCHK(hipMemcpy(h->fltu_mats, fs.mats.data(), fs.mats.size() * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMalloc((void **)&h->fltu_rows, rows.size() * 4));
  HIPCHK(hipMemcpy(h->fltu_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
  h->fltu_nitems = (int)fs.items.size();
  h->fltu_nrows = (int)rows.size();
  return 0;
}

extern "C" int uvic_gpu_isopyc(uvic_gpu *h) {
  if (!h) return fail_msg("null handle");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = launch_isopyc(h)) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int uvic_gpu_transport(uvic_gpu *h) {
  if (!h) return fail_msg("null handle");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = launch_transport(h, false)) return rc;   // transport alone: t(tau+1) before convct2
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int uvic_gpu_convect(uvic_gpu *h) {
  if (!h) return fail_msg("null handle");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = launch_convect(h)) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
static int launch_tracer(uvic_gpu *h) {
  if (int rc = launch_mobi(h)) return rc;
  if (int rc = launch_transport(h, true)) return rc;
  if (int rc = launch_convect(h)) return rc;
  return 0;
}
extern "C" int uvic_gpu_tracer(uvic_gpu *h) {
  if (!h) return fail_msg("null handle");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = launch_tracer(h)) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
// The point on the main stream where the previous step is complete: what the look-ahead chains of this step wait
// for.  Recorded once per step by whichever of step_async / step_pre_async / prefetch_* comes first.
static int step_begin(uvic_gpu *h) {
  if (h->step_begun) return 0;
  if (h->idle_until_next && h->end_ready) {
    // nothing was queued on the main stream since the previous step's end event: it is this step's begin (one marker
    // packet less on the stream whose chain is the step: ~6 us)
    h->ev_begin_cur = h->ev_end_ready;
  } else {
    HIPCHK(hipEventRecord(h->ev_step_begin, h->stream));
    h->ev_begin_cur = h->ev_step_begin;
  }
  h->idle_until_next = false;
  h->step_begun = true;
  return 0;
}
// asynchronous variants used by the time loop of bench.py: no host sync
// The point on the main stream where this step's own rows of t(tau+1) are complete -- before a latitude-slab caller
// queues the halo exchange.  The MOBI chain of the step after next is column-local and waits for this, not for the
// exchange (on a 12-row slab the chain is the critical path and the exchange costs 0.06 ms).
static int step_end(uvic_gpu *h) {
  h->in.waited = false;
  h->ev_end_pending = h->ev_step_end[h->ev_flip];
  HIPCHK(hipEventRecord(h->ev_end_pending, h->stream));
  h->end_step_of[h->ev_flip] = h->step_no;
  h->end_pending = true;
  if (!h->ts_final_valid) { h->ev_ts_final = h->ev_end_pending; h->ts_final_valid = true; }
  return 0;
}
extern "C" int uvic_gpu_step_async(uvic_gpu *h) {
  if (!h) return fail_msg("null handle");
  if (int rc = step_begin(h)) return rc;
  if (int rc = launch_isopyc(h, true)) return rc;
  if (int rc = launch_tracer(h)) return rc;
  return step_end(h);
}
extern "C" int uvic_gpu_prefetch_sources(uvic_gpu *h, double c2dtts_next);
extern "C" int uvic_gpu_prefetch_sources_at(uvic_gpu *h, double c2dtts_next, double relyr_next, double co2ccn_next);
extern "C" int uvic_gpu_prefetch_isopyc(uvic_gpu *h);
static int prefetch_isopyc_ahead(uvic_gpu *h, int ahead);
extern "C" int uvic_gpu_set_mixing(uvic_gpu *h, int on);
extern "C" int uvic_gpu_step_lookahead_at(uvic_gpu *h, double c2dtts, int mixing, int mobi_ahead, double c2dtts_next, double relyr_next,
                                          double co2ccn_next, int iso_ahead);
// One call per time step for a device-resident loop: what the Python TimeLoop does with six (at 0.2 ms per step of a
// small slab the host's share counts).  mixing: forward step (t(tau-1) := t(tau), c2dtts = dtts); mobi_ahead, iso_ahead:
// start the look-ahead chains of the NEXT step (only when that one is a leapfrog step with c2dtts_next and, for MOBI,
// keeps this step's surface forcing).  The caller then exchanges halo rows if it has neighbours, and calls
// uvic_gpu_rotate (which also ends a forward step's aliasing).
// The `_at` form names relyr and co2ccn of the next step (the reference advances its clock every ocean step and takes the
// month of the dust field and the declination from it, tracer.F:311-338): the look-ahead MOBI chain computes with them,
// and the next step discards the chain's sources if it is then given other values (uvic_gpu_set_mobi_step).
extern "C" int uvic_gpu_step_lookahead(uvic_gpu *h, double c2dtts, int mixing, int mobi_ahead, double c2dtts_next, int iso_ahead) {
  if (!h) return fail_msg("null handle");
  return uvic_gpu_step_lookahead_at(h, c2dtts, mixing, mobi_ahead, c2dtts_next, h->mobi.relyr, h->mobi.co2ccn, iso_ahead);
}
extern "C" int uvic_gpu_step_lookahead_at(uvic_gpu *h, double c2dtts, int mixing, int mobi_ahead, double c2dtts_next, double relyr_next,
                                          double co2ccn_next, int iso_ahead) {
  if (!h) return fail_msg("null handle");
  if (int rc = uvic_gpu_set_mixing(h, mixing)) return rc;
  h->ctx.c2dtts = c2dtts;
  if (int rc = uvic_gpu_step_async(h)) return rc;
  if (mobi_ahead && h->have_mobi)
    if (int rc = uvic_gpu_prefetch_sources_at(h, c2dtts_next, relyr_next, co2ccn_next)) return rc;
  // bit 0: the next step is a leapfrog step, bit 1: the step after next is (and no halo exchange follows)
  if (iso_ahead & 1)
    if (int rc = prefetch_isopyc_ahead(h, 1)) return rc;
  if ((iso_ahead & 2) && !h->ctx.diff_cbt_given)
    if (int rc = prefetch_isopyc_ahead(h, 2)) return rc;
  if (h->close_step) {
    // diagnosis (uvic_gpu_set_option "close_step"): every chain queued in this step ends before the step does -- what a
    // step captured as one HIP graph would impose, since a graph launch ends before the next one in its stream begins
    if (h->prefetch_pending) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_src_pending, 0));
    for (int q = 0; q < 3; ++q)
      if (h->iso_set[q].for_step > h->step_no) HIPCHK(hipStreamWaitEvent(h->stream, h->iso_set[q].ev, 0));
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_ts_done, 0));
  }
  h->unmix_at_rotate = mixing != 0;
  static const bool fewer = uv_env("UVIC_MAIN_ALIAS") && atoi(uv_env("UVIC_MAIN_ALIAS")) != 0;
  h->idle_until_next = fewer && (iso_ahead & 4) != 0;   // bit 2: the caller queues nothing on the main stream before the next step
  return 0;
}
// MOBI sources of the NEXT step from t(tau) (= next step's t(tau-1) on a leapfrog step) on
// the side stream, overlapped with this step's transport.  Call before (preferred: its kernels are then queued ahead of this step's side-stream work) or after uvic_gpu_step_async
// and before uvic_gpu_rotate; only valid when the next step is a leapfrog step.
extern "C" int uvic_gpu_prefetch_sources(uvic_gpu *h, double c2dtts_next) {
  if (!h) return fail_msg("null handle");
  return uvic_gpu_prefetch_sources_at(h, c2dtts_next, h->mobi.relyr, h->mobi.co2ccn);   // a clock that stands still
}
extern "C" int uvic_gpu_prefetch_sources_at(uvic_gpu *h, double c2dtts_next, double relyr_next, double co2ccn_next) {
  if (!h) return fail_msg("null handle");
  if (!h->have_mobi) return 0;
  const size_t bytes = (size_t)field_elems(h->d, UVIC_F_SRC) * 8;
  if (!h->src_alt) {
    HIPCHK(hipMalloc(&h->src_alt, bytes));
  }
  uvic_ctx c = h->ctx;
  mobi_dev m = h->mobi;
  c.t_taum1 = h->ctx.t_tau;
  const int q = h->mobi_flip;          // this chain's stream, work planes and event; the next prefetch takes the other set
  if (h->mobi_two_streams) h->mobi_flip ^= 1;
  hipStream_t st = h->side_m[q];
  const int sid = q ? 4 : 1;
  mobi_set_work(&m, h->mobi_st.work_side[q], h->d.imt, h->d.jmt, h->d.km);
  c.src = (const double *)h->src_alt;
  c.c2dtts = c2dtts_next;
  m.relyr = relyr_next; m.co2ccn = co2ccn_next;   // tracer.F:311-338 takes month and declination from the step's own relyr
  h->src_relyr = relyr_next; h->src_co2ccn = co2ccn_next; h->src_c2dtts = c2dtts_next;
  if (int rc = mobi_step_scalars(h, c2dtts_next, relyr_next, m.S)) return rc;
  if (int rc = step_begin(h)) return rc;
  // src_alt was read last by pass B of the previous step (before ev_step_begin); the other chain writes the other buffer
  // (the end of the previous step's own work is enough: MOBI is column-local, the halo rows do not matter to it)
  HIPCHK(hipStreamWaitEvent(st, h->end_ready ? h->ev_end_ready : h->ev_begin_cur, 0));
  if (int rc = forcing_on(h, st)) return rc;
  if (int rc = launch_mobi_on(h, c, m, st, sid)) return rc;
  h->ev_src_pending = h->ev_src_next[h->ev_flip];
  HIPCHK(hipEventRecord(h->ev_src_pending, st));
  h->prefetch_pending = true;
  return 0;
}
// The T,S-derived fields of a LATER step on a side stream, overlapped with this step.  isopyc of step m reads T,S at tau-1 of
// that step, on a leapfrog step m the result of step m-2:
//   ahead = 1: for the next step, from this step's t(tau); on the T,S stream, behind this step's T,S passes;
//   ahead = 2: for the step after next, from this step's t(tau+1) as soon as T and S of it are final (ev_ts_final: after the
//              convective walk on the T,S stream, or -- with the polar filter on -- at the end of the step); on the MOBI stream
//              whose chain (the sources of THIS step) has ended by then, so that neither the T,S passes of the next step nor
//              anything else the main stream waits for queues up behind it.  Not for a latitude slab (t(tau+1) of the halo
//              rows arrives with the exchange).
// Call after uvic_gpu_step_async and before uvic_gpu_rotate; only valid when the target step is a leapfrog step.
static int prefetch_isopyc_ahead(uvic_gpu *h, int ahead) {
  // (with an uploaded diff_cbt -- host vmixc, diff_cbt_has_k33 = 1 -- the chain leaves diff_cbt alone and the folded
  // vertical-diffusion coefficient is renewed by k_coef_bv once that step's diff_cbt is there: launch_isopyc)
  HIPCHK(hipSetDevice(h->device));
  const long long target = h->step_no + ahead;
  const int set = (int)(target % 3), cur = h->iso_cur;
  if (h->iso_set[set].for_step == target) return 0;   // there already
  hipStream_t st = h->side2;
  int sid = 2;
  if (ahead == 2) {
    if (!h->ts_final_valid) return 0;                  // nothing says when T,S are final: leave it to ahead = 1 of the next step
    st = h->side_m[h->mobi_flip];
    sid = h->mobi_flip ? 4 : 1;
  }
  if (int rc = iso_set_alloc(h, set, st)) return rc;
  // a context that writes set `set` and reads T,S of the level that will be tau-1 then
  if (int rc = use_iso_set(h, set)) return rc;
  uvic_ctx c = h->ctx;
  double *coef = h->coef;
  if (int rc = use_iso_set(h, cur)) return rc;
  c.t_taum1 = (ahead == 2) ? (const double *)h->ctx.t_taup1 : h->ctx.t_tau;
  if (int rc = step_begin(h)) return rc;
  // the set was last read by step target-3; whatever this step still does with t does not touch what the chain reads
  if (ahead == 2) HIPCHK(hipStreamWaitEvent(st, h->ev_ts_final, 0));
  else if (!(st == h->side_ts && h->ts_waited_begin == h->step_no))   // (this stream has waited for the step's begin already)
    HIPCHK(hipStreamWaitEvent(st, h->ev_begin_cur, 0));
  if (int rc = launch_isopyc_on(h, c, coef, st, sid)) return rc;
  HIPCHK(hipEventRecord(h->iso_set[set].ev, st));
  h->iso_set[set].st = st;
  h->iso_set[set].for_step = target;
  h->iso_set[set].vel_stale = false;
  return 0;
}
extern "C" int uvic_gpu_prefetch_isopyc(uvic_gpu *h) {
  if (!h) return fail_msg("null handle");
  return prefetch_isopyc_ahead(h, 1);
}
// forward (mixing) step: t(tau-1) aliases t(tau) until switched off again
extern "C" int uvic_gpu_set_mixing(uvic_gpu *h, int on) {
  if (!h) return fail_msg("null handle");
  h->mixing = on != 0;
  bind_ctx(h);
  return 0;
}
// sharded time loop: everything before the exchange of t(tau+1) ...
extern "C" int uvic_gpu_step_pre_async(uvic_gpu *h) {
  if (!h) return fail_msg("null handle");
  if (int rc = step_begin(h)) return rc;
  if (int rc = launch_isopyc(h, true)) return rc;
  if (int rc = launch_mobi(h)) return rc;
  return launch_transport(h, false);
}
// ... and convection (all tracers, needs T,S of every column) after it
extern "C" int uvic_gpu_convect_async(uvic_gpu *h) {
  if (!h) return fail_msg("null handle");
  if (int rc = launch_convect(h)) return rc;
  return step_end(h);
}
// -- latitude-slab halo rows of t(tau+1): staging buffers the exchange sends from and receives into
#define UVIC_HALO 2
extern "C" int64_t uvic_gpu_halo_elems(uvic_gpu *h) {
  return h ? (int64_t)UVIC_HALO * h->d.imt * h->d.km * h->d.nt : -1;
}
extern "C" void *uvic_gpu_halo_buffer(uvic_gpu *h, int which) {   // 0 send south, 1 send north, 2 receive south, 3 receive north
  if (!h || which < 0 || which > 3) return nullptr;
  if (!h->halo[which]) {
    if (hipSetDevice(h->device) != hipSuccess) return nullptr;
    const size_t bytes = (size_t)uvic_gpu_halo_elems(h) * 8;
    if (hipMalloc((void **)&h->halo[which], bytes) != hipSuccess) return nullptr;
    (void)hipMemsetAsync(h->halo[which], 0, bytes, h->stream);
  }
  return h->halo[which];
}
static int halo_move(uvic_gpu *h, int which, int j0, int dir) {
  double *stage = (double *)uvic_gpu_halo_buffer(h, which);
  if (!stage) return fail_msg("uvic_gpu_halo: no staging buffer");
  if (j0 < 1 || j0 + UVIC_HALO - 1 > h->d.jmt) return fail_msg("uvic_gpu_halo: rows outside 1..jmt");
  const int rowlen = h->d.imt * h->d.km;
  const long long n = (long long)UVIC_HALO * rowlen * h->d.nt;
  hipLaunchKernelGGL(k_halo_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (double *)h->buf[UVIC_F_T_TAUP1], stage,
                     rowlen, h->d.jmt, h->d.nt, j0, UVIC_HALO, dir);
  HIPCHK(hipGetLastError());
  return 0;
}
// pack the outermost owned rows of t(tau+1) (the slab of uvic_gpu_set_shard) into the send buffers, on the main stream
extern "C" int uvic_gpu_halo_pack(uvic_gpu *h, int south, int north) {
  if (h) h->idle_until_next = false;   // something is queued on the main stream between two steps
  if (!h) return fail_msg("null handle");
  if (south) if (int rc = halo_move(h, 0, h->ctx.js, 0)) return rc;
  if (north) if (int rc = halo_move(h, 1, h->ctx.je - UVIC_HALO + 1, 0)) return rc;
  return 0;
}
// ... and the received rows into the halo rows beyond the slab
extern "C" int uvic_gpu_halo_unpack(uvic_gpu *h, int south, int north) {
  if (h) h->idle_until_next = false;   // something is queued on the main stream between two steps
  if (!h) return fail_msg("null handle");
  if (south) if (int rc = halo_move(h, 2, h->ctx.js - UVIC_HALO, 1)) return rc;
  if (north) if (int rc = halo_move(h, 3, h->ctx.je + 1, 1)) return rc;
  return 0;
}
// -- direct push: the exchange of t(tau+1) after a step without a collective library.  Setup, once:
//   uvic_gpu_push_setup(world, rank, mode)   mode 1 tracer shards (every rank's slice to every other rank),
//                                            mode 2 latitude slabs (UVIC_HALO edge rows to the two neighbours)
//   uvic_gpu_push_export(handles)            2 x 64 bytes (hipIpcMemHandle_t of the window and of the counters) to hand to the peers
//   uvic_gpu_push_open(peer, handles)        what `peer` exported (its own rank: no mapping, the rank's own window)
// then per step uvic_gpu_push_exchange() on the main stream, where the RCCL collective would stand.
static_assert(sizeof(hipIpcMemHandle_t) == 64, "uvic_gpu_push_export hands out 2 x 64 bytes");
static void push_release(uvic_gpu *h) {
  for (int r = 0; r < UVIC_PUSH_MAX; ++r) {
    if (h->push.mapped[r]) { (void)hipIpcCloseMemHandle(h->push.peer_window[r]); (void)hipIpcCloseMemHandle(h->push.peer_flags[r]); }
    h->push.mapped[r] = false; h->push.peer_window[r] = nullptr; h->push.peer_flags[r] = nullptr;
  }
  (void)hipFree(h->push.window); (void)hipFree(h->push.flags); (void)hipHostFree(h->push.err);
  h->push.window = nullptr; h->push.flags = nullptr; h->push.err = nullptr; h->push.mode = 0; h->push.seq = 0;
}
extern "C" int uvic_gpu_push_setup(uvic_gpu *h, int world, int rank, int mode) {
  if (!h) return fail_msg("null handle");
  if (world < 1 || world > UVIC_PUSH_MAX || rank < 0 || rank >= world) return fail_msg("uvic_gpu_push_setup: rank outside the world, or more than 64 ranks");
  if (mode != 1 && mode != 2) return fail_msg("uvic_gpu_push_setup: mode is 1 (tracer shards) or 2 (latitude slabs)");
  if (mode == 1 && h->d.nt % world) return fail_msg("uvic_gpu_push_setup: tracer shards need nt padded to a multiple of the world size");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = uvic_gpu_sync(h)) return rc;
  push_release(h);
  h->push.world = world; h->push.rank = rank; h->push.mode = mode;
  h->push.slots = mode == 1 ? world : 2;
  h->push.slot_elems = mode == 1 ? (size_t)h->d.imt * h->d.km * h->d.jmt * (h->d.nt / world) : (size_t)uvic_gpu_halo_elems(h);
  // uncached: written by another device (or another process), read here one kernel boundary after the counter says so
  const size_t wbytes = 2 * h->push.slots * h->push.slot_elems * 8, fbytes = UVIC_PUSH_MAX * sizeof(unsigned long long);
  HIPCHK(hipExtMallocWithFlags((void **)&h->push.window, wbytes, hipDeviceMallocUncached));
  HIPCHK(hipExtMallocWithFlags((void **)&h->push.flags, fbytes, hipDeviceMallocUncached));
  HIPCHK(hipMemset(h->push.window, 0, wbytes));
  HIPCHK(hipMemset(h->push.flags, 0, fbytes));
  HIPCHK(hipHostMalloc((void **)&h->push.err, sizeof(int), hipHostMallocDefault));
  *h->push.err = 0;
  return 0;
}
extern "C" int uvic_gpu_push_export(uvic_gpu *h, void *handles) {
  if (!h || !handles) return fail_msg("uvic_gpu_push_export: null argument");
  if (!h->push.mode) return fail_msg("uvic_gpu_push_export: uvic_gpu_push_setup first");
  hipIpcMemHandle_t hw, hf;
  HIPCHK(hipIpcGetMemHandle(&hw, h->push.window));
  HIPCHK(hipIpcGetMemHandle(&hf, h->push.flags));
  memcpy(handles, &hw, 64);
  memcpy((char *)handles + 64, &hf, 64);
  return 0;
}
extern "C" int uvic_gpu_push_open(uvic_gpu *h, int peer, const void *handles) {
  if (!h) return fail_msg("null handle");
  if (!h->push.mode) return fail_msg("uvic_gpu_push_open: uvic_gpu_push_setup first");
  if (peer < 0 || peer >= h->push.world) return fail_msg("uvic_gpu_push_open: peer outside the world");
  if (h->push.peer_window[peer]) return fail_msg("uvic_gpu_push_open: peer already open");
  if (peer == h->push.rank) {   // a rank that is its own neighbour (cyclic tests at world size 1): no mapping
    h->push.peer_window[peer] = h->push.window; h->push.peer_flags[peer] = h->push.flags;
    return 0;
  }
  if (!handles) return fail_msg("uvic_gpu_push_open: null handles");
  HIPCHK(hipSetDevice(h->device));
  hipIpcMemHandle_t hw, hf;
  memcpy(&hw, handles, 64);
  memcpy(&hf, (const char *)handles + 64, 64);
  void *w = nullptr, *f = nullptr;
  HIPCHK(hipIpcOpenMemHandle(&w, hw, hipIpcMemLazyEnablePeerAccess));
  if (hipError_t e = hipIpcOpenMemHandle(&f, hf, hipIpcMemLazyEnablePeerAccess)) { (void)hipIpcCloseMemHandle(w); return fail("hipIpcOpenMemHandle", e, __LINE__); }
  h->push.peer_window[peer] = (double *)w; h->push.peer_flags[peer] = (unsigned long long *)f; h->push.mapped[peer] = true;
  return 0;
}
// One exchange, queued on the main stream: push to the peers' windows (parity seq & 1: a peer may run one exchange
// ahead of this rank, never two, because its next push waits for this rank's), raise their counters, wait for this
// rank's own, take what has arrived.  south/north: the neighbour ranks of a slab (-1: none); ignored for tracer shards.
extern "C" int uvic_gpu_push_exchange(uvic_gpu *h, int south, int north) {
  if (!h) return fail_msg("null handle");
  h->idle_until_next = false;   // something is queued on the main stream between two steps
  auto &P = h->push;
  if (!P.mode) return fail_msg("uvic_gpu_push_exchange: uvic_gpu_push_setup first");
  const unsigned long long seq = ++P.seq;
  const size_t half = (size_t)P.slots * P.slot_elems, par = (seq & 1) * half;
  const long long ticks = (long long)(P.wait_ms * 1e5);
  double *tp = (double *)h->buf[UVIC_F_T_TAUP1];
  unsigned long long mask = 0;
  if (P.mode == 2) {
    const int peer[2] = {south, north};
    const int rowlen = h->d.imt * h->d.km;
    const long long n = (long long)UVIC_HALO * rowlen * h->d.nt;
    PushTargets tg; tg.n = 0;
    for (int side = 0; side < 2; ++side) {
      const int p = peer[side];
      if (p < 0) continue;
      if (p >= P.world || !P.peer_window[p]) return fail_msg("uvic_gpu_push_exchange: neighbour not open");
      // the rows this rank sends south arrive in the neighbour's slot 1 ("from the north"), and the other way round
      double *dst = P.peer_window[p] + par + (size_t)(1 - side) * P.slot_elems;
      const int j0 = side == 0 ? h->ctx.js : h->ctx.je - UVIC_HALO + 1;
      hipLaunchKernelGGL(k_halo_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, tp, dst, rowlen, h->d.jmt, h->d.nt, j0, UVIC_HALO, 0);
      tg.flag[tg.n++] = P.peer_flags[p] + (1 - side);
      mask |= 1ull << side;
    }
    if (!tg.n) return 0;
    hipLaunchKernelGGL(k_push_raise, dim3(1), dim3(64), 0, h->stream, tg, seq);
    hipLaunchKernelGGL(k_push_wait, dim3(1), dim3(64), 0, h->stream, P.flags, mask, seq, ticks, P.err);
    for (int side = 0; side < 2; ++side) {
      if (peer[side] < 0) continue;
      const int j0 = side == 0 ? h->ctx.js - UVIC_HALO : h->ctx.je + 1;
      if (j0 < 1 || j0 + UVIC_HALO - 1 > h->d.jmt) return fail_msg("uvic_gpu_push_exchange: halo rows outside 1..jmt");
      hipLaunchKernelGGL(k_halo_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, tp, P.window + par + (size_t)side * P.slot_elems,
                         rowlen, h->d.jmt, h->d.nt, j0, UVIC_HALO, 1);
    }
  } else {
    if (P.world == 1) return 0;
    const long long per2 = (long long)(P.slot_elems / 2);
    if (P.slot_elems & 1) return fail_msg("uvic_gpu_push_exchange: odd slice length");
    const double2 *mine = (const double2 *)(tp + (size_t)P.rank * P.slot_elems);
    for (int p0 = 0; p0 < P.world; p0 += 8) {
      PushTargets tg; tg.n = 0;
      for (int p = p0; p < P.world && p < p0 + 8; ++p) {
        if (p == P.rank) continue;
        if (!P.peer_window[p]) return fail_msg("uvic_gpu_push_exchange: peer not open");
        tg.dst[tg.n] = P.peer_window[p] + par + (size_t)P.rank * P.slot_elems;
        tg.flag[tg.n++] = P.peer_flags[p] + P.rank;
        mask |= 1ull << p;
      }
      if (!tg.n) continue;
      const unsigned nb = (unsigned)std::min<long long>((per2 + 255) / 256, 2048 / tg.n + 1);
      hipLaunchKernelGGL(k_push_slice, dim3(nb, tg.n), dim3(256), 0, h->stream, mine, tg, per2);
      hipLaunchKernelGGL(k_push_raise, dim3(1), dim3(64), 0, h->stream, tg, seq);
    }
    hipLaunchKernelGGL(k_push_wait, dim3(1), dim3(64), 0, h->stream, P.flags, mask, seq, ticks, P.err);
    const long long n2 = per2 * (P.world - 1);
    hipLaunchKernelGGL(k_push_take, dim3((unsigned)std::min<long long>((n2 + 255) / 256, 4096)), dim3(256), 0, h->stream, (double2 *)tp,
                       (const double2 *)(P.window + par), per2, P.world, P.rank);
  }
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int uvic_gpu_rotate(uvic_gpu *h) {
  if (!h) return fail_msg("null handle");
  h->step_begun = false;
  if (h->unmix_at_rotate) { h->mixing = false; h->unmix_at_rotate = false; }
  h->ev_flip ^= 1;
  h->end_ready = h->end_pending;
  h->ev_end_ready = h->ev_end_pending;
  h->end_pending = false;
  void *m1 = h->buf[UVIC_F_T_TAUM1], *t0 = h->buf[UVIC_F_T_TAU], *p1 = h->buf[UVIC_F_T_TAUP1];
  h->buf[UVIC_F_T_TAUM1] = t0;
  h->buf[UVIC_F_T_TAU] = p1;
  h->buf[UVIC_F_T_TAUP1] = m1;
  if (h->prefetch_pending) {  // the side stream filled the other source buffer for the step that starts now
    void *s0 = h->buf[UVIC_F_SRC];
    h->buf[UVIC_F_SRC] = h->src_alt;
    h->src_alt = s0;
    h->prefetch_pending = false;
    h->src_from_prefetch = true;
    h->ev_src_ready = h->ev_src_pending;
  }
  h->tsi_step = false;
  h->tavg_step = false;
  if (h->cv_lists[0]) {   // the convective walk's arrays of the step that starts now
    const int odd = (int)((h->step_no + 1) & 1);
    h->cv_list = h->cv_lists[odd];
    int **ci = odd ? h->cv_int2 : h->cv_int;
    h->ctx.cv_nseg = ci[0]; h->ctx.cv_kt = ci[1]; h->ctx.cv_kb = ci[2]; h->ctx.cv_z = odd ? h->cv_z2 : h->cv_z;
  }
  h->step_no += 1;      // the T,S-derived fields of the new step live in set step_no % 3: current from now on, so that
  h->ts_final_valid = false;   // a diff_cbt uploaded for that step lands in it
  if (int rc = use_iso_set(h, (int)(h->step_no % 3))) return rc;
  bind_ctx(h);
  return 0;
}
extern "C" int uvic_gpu_sync(uvic_gpu *h) {
  if (!h) return fail_msg("null handle");
  if (h->side_mom) HIPCHK(hipStreamSynchronize(h->side_mom));
  for (int q = 0; q < 2; ++q) HIPCHK(hipStreamSynchronize(h->side_m[q]));
  HIPCHK(hipStreamSynchronize(h->side2));
  HIPCHK(hipStreamSynchronize(h->side_ts));
  for (int q = 0; q < 2; ++q) if (h->in.st[q]) HIPCHK(hipStreamSynchronize(h->in.st[q]));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->push.err && *h->push.err) {
    const int who = *h->push.err - 1;
    *h->push.err = 0;
    return fail_msg("uvic_gpu_push_exchange: nothing arrived from " + (h->push.mode == 2 ? std::string(who ? "the north" : "the south") : "rank " + std::to_string(who)) +
                    " within the waiting time: t(tau+1) is incomplete");
  }
  return 0;
}

static void profile_reset(uvic_gpu *h) {
  for (int q = 0; q < 5; ++q) { h->ev[q].clear(); h->ev_names[q].clear(); }
}
// mean duration per kernel name from the recorded events (consecutive events of one stream)
static int profile_collect(uvic_gpu *h, int max_kernels, const char **names, double *mean_ms, int *nkernels) {
  for (int q = 0; q < 2; ++q) HIPCHK(hipStreamSynchronize(h->side_m[q]));
  HIPCHK(hipStreamSynchronize(h->side2));
  HIPCHK(hipStreamSynchronize(h->side_ts));
  HIPCHK(hipStreamSynchronize(h->stream));
  std::vector<KernelStat> st;
  for (int q = 0; q < 5; ++q)
    for (size_t e = 1; e < h->ev[q].size(); ++e) {
      if (strcmp(h->ev_names[q][e], "begin") == 0) continue;
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, h->ev[q][e - 1], h->ev[q][e]));
      bool found = false;
      for (auto &s : st)
        if (strcmp(s.name, h->ev_names[q][e]) == 0) {
          s.ms += ms; s.calls++; found = true;
        }
      if (!found) st.push_back({h->ev_names[q][e], (double)ms, 1});
    }
  int n = 0;
  for (auto &s : st) {
    if (n >= max_kernels) break;
    names[n] = s.name;
    mean_ms[n] = s.ms / s.calls;
    ++n;
  }
  *nkernels = n;
  profile_reset(h);
  return 0;
}
extern "C" int uvic_gpu_profile(uvic_gpu *h, int nrep, int max_kernels, const char **names, double *mean_ms, int *nkernels) {
  if (!h || !names || !mean_ms || !nkernels) return fail_msg("uvic_gpu_profile: null argument");
  HIPCHK(hipSetDevice(h->device));
  profile_reset(h);
  h->profiling = true;
  h->serial = true;
  int rc = 0;
  for (int r = 0; r < nrep && !rc; ++r) {
    rc = launch_isopyc(h);
    if (!rc) rc = launch_tracer(h);
  }
  h->profiling = false;
  h->serial = false;
  if (rc) return rc;
  return profile_collect(h, max_kernels, names, mean_ms, nkernels);
}
// live form: events are recorded around every kernel of the caller's own time loop (both streams)
// from uvic_gpu_profile_live(h, 1) until uvic_gpu_profile_read, which returns the means and stops
extern "C" int uvic_gpu_profile_live(uvic_gpu *h, int on) {
  if (!h) return fail_msg("null handle");
  HIPCHK(hipSetDevice(h->device));
  profile_reset(h);
  h->profiling = on != 0;
  return 0;
}
extern "C" int uvic_gpu_profile_read(uvic_gpu *h, int max_kernels, const char **names, double *mean_ms, int *nkernels) {
  if (!h || !names || !mean_ms || !nkernels) return fail_msg("uvic_gpu_profile_read: null argument");
  HIPCHK(hipSetDevice(h->device));
  h->profiling = false;
  return profile_collect(h, max_kernels, names, mean_ms, nkernels);
}

// -- MOBI parameters -------------------------------------------------------------
extern "C" int uvic_gpu_set_mobi(uvic_gpu *h, const uvic_mobi_params *p, const uvic_mobi_forcing *f) {
  if (!h || !p || !f) return fail_msg("uvic_gpu_set_mobi: null argument");
  if (p->nsrc != h->d.nsrc || p->ntnpzd != h->d.ntnpzd) return fail_msg("uvic_gpu_set_mobi: nsrc/ntnpzd differ from uvic_gpu_create");
  if (p->dtnpzd <= 0.0) return fail_msg("uvic_gpu_set_mobi: dtnpzd must be positive");
  HIPCHK(hipSetDevice(h->device));
  int rc = mobi_bind(h->d.imt, h->d.jmt, h->d.km, p, f, &h->mobi, &h->mobi_st, h->stream, g_err);
  if (rc) return rc;
  h->mobi_dtnpzd = p->dtnpzd;
  h->have_mobi = true;
  return 0;
}
extern "C" int uvic_gpu_set_mobi_opt(uvic_gpu *h, const uvic_mobi_params *p, const uvic_mobi_options *o, const uvic_mobi_forcing *f) {
  if (!h || !p || !o || !f) return fail_msg("uvic_gpu_set_mobi_opt: null argument");
  if (o->n15 && o->c13 && !o->caco3 && !o->silicon && !h->mobi_generic) return uvic_gpu_set_mobi(h, p, f);   // option set C: its own kernels
  if (p->nsrc != h->d.nsrc || p->ntnpzd != h->d.ntnpzd) return fail_msg("uvic_gpu_set_mobi_opt: nsrc/ntnpzd differ from uvic_gpu_create");
  if (p->dtnpzd <= 0.0) return fail_msg("uvic_gpu_set_mobi_opt: dtnpzd must be positive");
  if (p->ntnpzd > UV_MOBI_MAXT) return fail_msg("uvic_gpu_set_mobi_opt: ntnpzd > 40");
  // what the general kernel indexes with must be there: every pool of the set has a column position, a tracer and a slot
  static const int always[] = {X_po4, X_phyt, X_phyt_phos, X_zoop, X_detr, X_detr_phos, X_dic, X_dop, X_no3, X_don, X_diaz, X_dfe, X_detrfe};
  static const int n15s[] = {X_din15, X_don15, X_phytn15, X_zoopn15, X_detrn15, X_diazn15};
  static const int c13s[] = {X_dic13, X_phytc13, X_zoopc13, X_detrc13, X_doc13, X_diazc13};
  std::vector<int> need(always, always + sizeof always / sizeof *always);
  if (o->n15) need.insert(need.end(), n15s, n15s + 6);
  if (o->c13) need.insert(need.end(), c13s, c13s + 6);
  if (o->caco3) need.push_back(X_caco3);
  if (o->silicon) { need.push_back(X_diat); need.push_back(X_sil); need.push_back(X_opl); }
  if (o->silicon && o->n15) need.push_back(X_diatn15);
  if (o->silicon && o->c13) need.push_back(X_diatc13);
  if (o->caco3 && o->c13) need.push_back(X_caco3c13);
  for (int x : need) {
    const int m = o->im[x], s = o->is[x];
    if (m < 1 || m > p->ntnpzd || s < 1 || s > p->nsrc) return fail_msg("uvic_gpu_set_mobi_opt: a MOBI tracer of this option set has no column position or source slot");
    if (p->tracer_of_mobi[m - 1] < 1 || p->tracer_of_mobi[m - 1] > h->d.nt) return fail_msg("uvic_gpu_set_mobi_opt: tracer_of_mobi out of range");
  }
  if (o->is_alk < 1 || o->is_alk > p->nsrc || o->is_o2 < 1 || o->is_o2 > p->nsrc || o->is_c14 < 0 || o->is_c14 > p->nsrc)
    return fail_msg("uvic_gpu_set_mobi_opt: source slot of alk, o2 or c14 out of range");
  if (p->itemp < 1 || p->isalt < 1 || p->idic < 1 || p->ialk < 1 || p->io2 < 1 || (o->is_c14 > 0 && p->ic14 < 1))
    return fail_msg("uvic_gpu_set_mobi_opt: temp, salt, dic, alk and o2 are needed (without O_mobi_alk the reference itself is undefined: ialk = 0)");
  HIPCHK(hipSetDevice(h->device));
  int rc = mobi_bind(h->d.imt, h->d.jmt, h->d.km, p, f, &h->mobi, &h->mobi_st, h->stream, g_err, o);
  if (rc) return rc;
  h->mobi_key = (o->n15 != 0) | ((o->c13 != 0) << 1) | ((o->caco3 != 0) << 2) | ((o->silicon != 0) << 3);
  h->mobi_dtnpzd = p->dtnpzd;
  h->have_mobi = true;
  return 0;
}
extern "C" int uvic_gpu_set_mobi_flat(uvic_gpu *h, int km, int ntnpzd, int nsrc, const int32_t *idx, const int32_t *tracer_of_mobi,
                                      const int32_t *slot_of_mobi, const int32_t *itr, const double *scal, const double *prof,
                                      const double *fsc, const double *tlat, const double *dnswr, const double *aice,
                                      const double *hice, const double *hsno, const double *sg_bathy,
                                      const double *fe_atmdep, const double *fe_hydr) {
  if (!idx || !tracer_of_mobi || !slot_of_mobi || !itr || !scal || !prof || !fsc) return fail_msg("uvic_gpu_set_mobi_flat: null argument");
  if (ntnpzd < 1 || ntnpzd > 40 || km < 1 || km > 64) return fail_msg("uvic_gpu_set_mobi_flat: ntnpzd or km out of range");
  uvic_mobi_params P;
  memset(&P, 0, sizeof P);
  P.km = km; P.ntnpzd = ntnpzd; P.nsrc = nsrc;
  memcpy(&P.im, idx, sizeof(uvic_mobi_index));
  memcpy(&P.is, idx + 28, sizeof(uvic_mobi_index));
  for (int m = 0; m < ntnpzd; ++m) { P.tracer_of_mobi[m] = tracer_of_mobi[m]; P.slot_of_mobi[m] = slot_of_mobi[m]; }
  P.itemp = itr[0]; P.isalt = itr[1]; P.idic = itr[2]; P.ialk = itr[3]; P.io2 = itr[4]; P.ic14 = itr[5];
  P.dtnpzd = scal[0];
  const size_t nscal = (size_t)(&P.capr - &P.kw) + 1;   // contiguous doubles kw .. capr
  memcpy(&P.kw, scal + 1, nscal * sizeof(double));
  double *arr[7] = {P.wd, P.ztt, P.rcak, P.rcab, P.zt, P.dzt, P.dztr};
  for (int a = 0; a < 7; ++a) memcpy(arr[a], prof + (size_t)a * km, (size_t)km * sizeof(double));
  uvic_mobi_forcing F;
  F.pi = fsc[0]; F.radian = fsc[1]; F.relyr = fsc[2]; F.co2ccn = fsc[3];
  F.tlat = tlat; F.dnswr = dnswr; F.aice = aice; F.hice = hice; F.hsno = hsno;
  F.sg_bathy = sg_bathy; F.fe_atmdep = fe_atmdep; F.fe_hydr = fe_hydr;
  if (h && h->have_opt_staged) return uvic_gpu_set_mobi_opt(h, &P, &h->opt_staged, &F);
  return uvic_gpu_set_mobi(h, &P, &F);
}
extern "C" int uvic_gpu_mobi_options_flat(uvic_gpu *h, const int32_t *flags, const int32_t *im, const int32_t *is, const int32_t *isx,
                                          const double *oscal, const double *wc, const double *wo, int km) {
  if (!h || !flags || !im || !is || !isx || !oscal) return fail_msg("uvic_gpu_mobi_options_flat: null argument");
  if (km < 1 || km > 64) return fail_msg("uvic_gpu_mobi_options_flat: km out of range");
  uvic_mobi_options &O = h->opt_staged;
  memset(&O, 0, sizeof O);
  O.n15 = flags[0]; O.c13 = flags[1]; O.caco3 = flags[2]; O.silicon = flags[3];
  for (int q = 0; q < UVIC_MOBI_NX; ++q) { O.im[q] = im[q]; O.is[q] = is[q]; }
  O.is_alk = isx[0]; O.is_o2 = isx[1]; O.is_c14 = isx[2];
  memcpy(&O.kc_c, oscal, (size_t)((&O.opl_disk0 - &O.kc_c) + 1) * sizeof(double));
  if (O.caco3) { if (!wc) return fail_msg("uvic_gpu_mobi_options_flat: wc missing"); memcpy(O.wc, wc, (size_t)km * 8); }
  if (O.silicon) { if (!wo) return fail_msg("uvic_gpu_mobi_options_flat: wo missing"); memcpy(O.wo, wo, (size_t)km * 8); }
  h->have_opt_staged = true;
  return 0;
}
// the part of the MOBI forcing that changes from step to step (tracer.F:355-390 reads dnswr, aice, hice, hsno of the
// current step; relyr selects the month of the dust field and the declination; co2ccn the atmospheric CO2)
extern "C" int uvic_gpu_set_mobi_step(uvic_gpu *h, double relyr, double co2ccn, const double *dnswr, const double *aice,
                                      const double *hice, const double *hsno) {
  if (h) h->idle_until_next = false;   // something is queued on the main stream between two steps
  if (!h) return fail_msg("uvic_gpu_set_mobi_step: null argument");
  if (!h->have_mobi) return fail_msg("uvic_gpu_set_mobi_step: call uvic_gpu_set_mobi first");
  HIPCHK(hipSetDevice(h->device));
  const double *src[4] = {dnswr, aice, hice, hsno};
  if (dnswr || aice || hice || hsno) {   // (all null: the forcing fields stay, only the clock and the CO2 move on)
    if (!dnswr || !aice || !hice || !hsno) return fail_msg("uvic_gpu_set_mobi_step: give all four forcing fields or none");
    // a look-ahead chain still running reads the old fields, and its sources are not those of the new forcing
    for (int q = 0; q < 2; ++q) HIPCHK(hipStreamSynchronize(h->side_m[q]));
    h->prefetch_pending = false;
    if (h->src_from_prefetch) {
      HIPCHK(hipStreamWaitEvent(h->stream, h->ev_src_ready, 0));
      h->src_from_prefetch = false;
    }
    const size_t bytes = (size_t)h->d.imt * h->d.jmt * 8;
    // page-locked arrays of a caller that does not wait for its uploads (the resident overlay): the first MOBI chain that
    // follows fetches them itself (forcing_on); the chains that read the old fields have ended (synchronised above, the
    // in-line ones with their step).  The arrays must stay as they are until the step's uvic_gpu_overlay_step returns.
    void *dp[4] = {nullptr, nullptr, nullptr, nullptr};
    bool mapped = !h->host_sync;
    for (int q = 0; q < 4 && mapped; ++q)
      if (hipHostGetDevicePointer(&dp[q], (void *)src[q], 0) != hipSuccess) { (void)hipGetLastError(); mapped = false; }
    if (mapped) {
      if (!h->in.ev_forcing) HIPCHK(hipEventCreateWithFlags(&h->in.ev_forcing, hipEventDisableTiming));
      for (int q = 0; q < 4; ++q) h->in.forcing_src[q] = (const double *)dp[q];
      h->in.forcing_pull = true; h->in.forcing_sent = false;
    } else {
      h->in.forcing_pull = false;
      for (int q = 0; q < 4; ++q) HIPCHK(hipMemcpyAsync(h->mobi_st.f[1 + q], src[q], bytes, hipMemcpyHostToDevice, h->stream));
      if (h->host_sync) HIPCHK(hipStreamSynchronize(h->stream));
      else {   // MOBI chains run on side streams: they wait for these copies
        if (!h->in.ev_forcing) HIPCHK(hipEventCreateWithFlags(&h->in.ev_forcing, hipEventDisableTiming));
        HIPCHK(hipEventRecord(h->in.ev_forcing, h->stream));
        h->in.forcing_sent = true;
      }
    }
  }
  h->mobi.relyr = relyr;
  h->mobi.co2ccn = co2ccn;
  return 0;
}
// Page-lock a host range the caller will upload from or download into every step (COMMON blocks live as long as the
// process): transfers then run at the link's rate instead of through the runtime's staging buffer.  The memory stays the
// caller's; the registration is this handle's: uvic_gpu_destroy (or uvic_gpu_unpin_host) ends it, so that
// a range the caller frees afterwards does not stay page-locked under whatever the allocator puts there next.
extern "C" int uvic_gpu_pin_host(uvic_gpu *h, void *ptr, int64_t bytes) {
  if (!h || !ptr || bytes <= 0) return fail_msg("uvic_gpu_pin_host: null argument");
  HIPCHK(hipSetDevice(h->device));
  const hipError_t e = hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault);
  if (e == hipErrorHostMemoryAlreadyRegistered) { (void)hipGetLastError(); return 0; }
  if (e != hipSuccess) return fail("hipHostRegister", e, __LINE__);
  h->pinned.push_back(ptr);
  return 0;
}
extern "C" int uvic_gpu_unpin_host(uvic_gpu *h, void *ptr) {
  if (!h || !ptr) return fail_msg("uvic_gpu_unpin_host: null argument");
  auto it = std::find(h->pinned.begin(), h->pinned.end(), ptr);
  if (it == h->pinned.end()) return 0;
  HIPCHK(hipSetDevice(h->device));
  if (int rc = uvic_gpu_sync(h)) return rc;
  (void)hipHostUnregister(ptr);
  h->pinned.erase(it);
  return 0;
}
extern "C" int uvic_gpu_set_host_sync(uvic_gpu *h, int on) {
  if (!h) return fail_msg("null handle");
  h->host_sync = on != 0;
  return 0;
}
// -- surface boundary condition sums (u09/mom/set_sbc.F:36-72) kept on the device ---------------------------------------
// set_sbc adds t(i,1,j,n,taup1) to sbc(i,j,isbc) every step, zeroes sbc at a segment's first step and averages it at
// the last (ocean points only).  With the state resident the host would need the surface level of every such tracer
// each step for nothing but this sum: the device keeps the sums (same additions in the same order) and the host fetches
// them once per segment.
__global__ void __launch_bounds__(256) k_sbc_accumulate(const uvic_ctx c, const int *tracers, double *acc, int count, int zero_first) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long ns = (long long)c.imt * c.jmt;
  if (gid >= ns * count) return;
  const int q = (int)(gid / ns), ij = (int)(gid % ns);
  const int i = ij % c.imt, j = ij / c.imt;
  if (i < 1 || i > c.imt - 2 || j < 1 || j > c.jmt - 2) return;   // set_sbc.F:40-72 runs over i = 2..imt-1 and the rows of the window
  double a = acc[gid];
  if (zero_first && c.kmt[ij] != 0) a = 0.0;
  const size_t N3 = (size_t)c.imt * c.km * c.jmt;
  acc[gid] = a + c.t_taup1[(size_t)(tracers[q] - 1) * N3 + (size_t)i + (size_t)c.imt * c.km * j];
}
// The step that follows is a time-step-monitor step (tsiperts, source/common/switch.F:458-459): form tbar, travar, dtabs
// (diagt1, u09/mom/tracer.F:1516-1537) and, when ic14 and idic are tracer numbers, the volume sum of delta 14C (:1329-1353)
// on the device.  Holds until the next uvic_gpu_rotate.
extern "C" int uvic_gpu_set_tsi(uvic_gpu *h, int on, int ic14, int idic) {
  if (!h) return fail_msg("uvic_gpu_set_tsi: null handle");
  if (ic14 < 0 || ic14 > h->d.nt || idic < 0 || idic > h->d.nt) return fail_msg("uvic_gpu_set_tsi: tracer number outside 1..nt");
  HIPCHK(hipSetDevice(h->device));
  if (on && !h->tsi_acc) {
    const size_t bytes = ((size_t)3 * (h->d.km + 1) * h->d.nt * h->d.jmt + (size_t)h->d.km * h->d.jmt) * 8;
    HIPCHK(hipMalloc((void **)&h->tsi_acc, bytes));
    HIPCHK(hipMemset(h->tsi_acc, 0, bytes));
    HIPCHK(hipHostMalloc((void **)&h->tsi_host, bytes, hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&h->ev_tsi, hipEventDisableTiming));
  }
  h->tsi_step = on != 0;
  h->tsi_ic14 = ic14; h->tsi_idic = idic;
  return 0;
}
// ... and fetch them once the step is complete (waits for it): tbar, travar, dtabs as source/common/diag.h declares them,
// (0:km, nt, jmt) each (rows and levels the step did not compute are zero, as diagi leaves them); dc14bar = the sum of the
// rows' sums, rows ascending.  Call before the uvic_gpu_rotate that ends the step, or right after uvic_gpu_overlay_step.
// -- time-average steps (O_time_averages, timavgperts; u09/mom/tracer.F:1211-1222, 1355-1364 with O_save_convection and
// O_carbon_14): the step named by uvic_gpu_set_tavg also forms the convection diagnostics totalk, vdepth, pe of convct2
// (source/mom/convect.F:183-301) and the delta-14C field; uvic_gpu_tavg_read fetches them after the step (before the
// rotation that ends it, or right after uvic_gpu_overlay_step).  grav, zt(km): pconst / coord.h.  ic14/idic = 0: no 14C.
extern "C" int uvic_gpu_set_tavg(uvic_gpu *h, int on, double grav, const double *zt, int ic14, int idic) {
  if (!h) return fail_msg("uvic_gpu_set_tavg: null handle");
  if (ic14 < 0 || ic14 > h->d.nt || idic < 0 || idic > h->d.nt) return fail_msg("uvic_gpu_set_tavg: tracer number outside 1..nt");
  HIPCHK(hipSetDevice(h->device));
  if (on) {
    if (!zt) return fail_msg("uvic_gpu_set_tavg: zt missing");
    const size_t N2 = (size_t)h->d.imt * h->d.jmt, N3 = N2 * h->d.km;
    if (!h->tavg_zt) {
      HIPCHK(hipMalloc((void **)&h->tavg_zt, (size_t)h->d.km * 8));
      HIPCHK(hipMalloc((void **)&h->tavg_diag, 3 * N2 * 8));
      HIPCHK(hipMemset(h->tavg_diag, 0, 3 * N2 * 8));
    }
    if (ic14 > 0 && idic > 0 && !h->tavg_dc14) {
      HIPCHK(hipMalloc((void **)&h->tavg_dc14, N3 * 8));
      HIPCHK(hipMemset(h->tavg_dc14, 0, N3 * 8));
    }
    HIPCHK(hipMemcpyAsync(h->tavg_zt, zt, (size_t)h->d.km * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->tavg_grav = grav;
  }
  h->tavg_step = on != 0;
  h->tavg_ic14 = ic14; h->tavg_idic = idic;
  return 0;
}
extern "C" int uvic_gpu_tavg_read(uvic_gpu *h, double *totalk, double *vdepth, double *pe, double *dc14) {
  if (!h || !totalk || !vdepth || !pe) return fail_msg("uvic_gpu_tavg_read: null argument");
  if (!h->tavg_diag) return fail_msg("uvic_gpu_tavg_read: no time-average step has run (uvic_gpu_set_tavg)");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = uvic_gpu_sync(h)) return rc;
  const size_t N2 = (size_t)h->d.imt * h->d.jmt, N3 = N2 * h->d.km;
  HIPCHK(hipMemcpy(totalk, h->tavg_diag, N2 * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(vdepth, h->tavg_diag + N2, N2 * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(pe, h->tavg_diag + 2 * N2, N2 * 8, hipMemcpyDeviceToHost));
  if (dc14) {
    if (!h->tavg_dc14) return fail_msg("uvic_gpu_tavg_read: no delta-14C field (uvic_gpu_set_tavg without ic14, idic)");
    HIPCHK(hipMemcpy(dc14, h->tavg_dc14, N3 * 8, hipMemcpyDeviceToHost));
  }
  return 0;
}
extern "C" int uvic_gpu_tsi_read(uvic_gpu *h, double *tbar, double *travar, double *dtabs, double *dc14bar) {
  if (!h || !tbar || !travar || !dtabs) return fail_msg("uvic_gpu_tsi_read: null argument");
  if (!h->tsi_acc) return fail_msg("uvic_gpu_tsi_read: no time-step-monitor step has run (uvic_gpu_set_tsi)");
  HIPCHK(hipSetDevice(h->device));
  const size_t NA = (size_t)(h->d.km + 1) * h->d.nt * h->d.jmt, NR = (size_t)h->d.km * h->d.jmt;
  if (h->tsi_inflight) {   // the step's own copy (launch_convect)
    HIPCHK(hipEventSynchronize(h->ev_tsi));
    h->tsi_inflight = false;
  } else {
    if (int rc = uvic_gpu_sync(h)) return rc;
    HIPCHK(hipMemcpy(h->tsi_host, h->tsi_acc, (3 * NA + NR) * 8, hipMemcpyDeviceToHost));
  }
  memcpy(tbar, h->tsi_host, NA * 8);
  memcpy(travar, h->tsi_host + NA, NA * 8);
  memcpy(dtabs, h->tsi_host + 2 * NA, NA * 8);
  if (dc14bar) {
    const double *rows = h->tsi_host + 3 * NA;
    double sum = 0.0;
    for (int j = h->ctx.js; j <= h->ctx.je; ++j)
      for (int k = 1; k <= h->d.km; ++k) sum = sum + rows[(size_t)(k - 1) + (size_t)h->d.km * (j - 1)];
    *dc14bar = sum;
  }
  return 0;
}
// ektot of O_time_step_monitor (clinic.F:616-630) from the u(tau) on the device (UVIC_F_U1, UVIC_F_U2), (0:km, jmt) as
// source/common/diag.h declares it; rows and levels outside the window are zero.  Synchronous.
extern "C" int uvic_gpu_tsi_ektot(uvic_gpu *h, double rho0, double *ektot) {
  if (!h || !ektot) return fail_msg("uvic_gpu_tsi_ektot: null argument");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = mom_join(h)) return rc;
  const size_t n = (size_t)(h->d.km + 1) * h->d.jmt;
  double *dev;
  HIPCHK(hipMalloc((void **)&dev, n * 8));
  HIPCHK(hipMemsetAsync(dev, 0, n * 8, h->stream));
  const int work = h->d.km * (h->ctx.je - h->ctx.js + 1);
  hipLaunchKernelGGL(k_tsi_ektot, dim3((unsigned)((work + TSI_ROWS - 1) / TSI_ROWS)), dim3(64), 0, h->stream, h->ctx, (const double *)h->buf[UVIC_F_U1],
                     (const double *)h->buf[UVIC_F_U2], rho0, dev);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(ektot, dev, n * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipFree(dev));
  return 0;
}
extern "C" int uvic_gpu_sbc_config(uvic_gpu *h, int count, const int32_t *tracers) {
  if (!h || count < 0 || (count > 0 && !tracers)) return fail_msg("uvic_gpu_sbc_config: bad argument");
  HIPCHK(hipSetDevice(h->device));
  for (int q = 0; q < count; ++q)
    if (tracers[q] < 1 || tracers[q] > h->d.nt) return fail_msg("uvic_gpu_sbc_config: tracer number outside 1..nt");
  (void)hipFree(h->sbc_tracer); (void)hipFree(h->sbc_acc);
  h->sbc_tracer = nullptr; h->sbc_acc = nullptr; h->sbc_count = count;
  if (count == 0) return 0;
  const size_t n = (size_t)h->d.imt * h->d.jmt * count;
  HIPCHK(hipMalloc((void **)&h->sbc_tracer, (size_t)count * 4));
  HIPCHK(hipMemcpy(h->sbc_tracer, tracers, (size_t)count * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMalloc((void **)&h->sbc_acc, n * 8));
  HIPCHK(hipMemset(h->sbc_acc, 0, n * 8));
  return 0;
}
// host (imt, jmt, count) <-> the sums; synchronises the main stream
extern "C" int uvic_gpu_sbc_transfer(uvic_gpu *h, double *host, int upload) {
  if (!h || !host) return fail_msg("uvic_gpu_sbc_transfer: null argument");
  if (h->sbc_count == 0) return 0;
  HIPCHK(hipSetDevice(h->device));
  const size_t bytes = (size_t)h->d.imt * h->d.jmt * h->sbc_count * 8;
  if (upload) HIPCHK(hipMemcpyAsync(h->sbc_acc, host, bytes, hipMemcpyHostToDevice, h->stream));
  else HIPCHK(hipMemcpyAsync(host, h->sbc_acc, bytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
// One call of the resident Fortran overlay: the step and its look-ahead chains are queued, T and S of t(tau+1) travel to
// `ts_host` (imt, km, jmt, 2) as soon as they are final, the surface sums are updated behind the step, the time levels
// rotate -- and the call returns when T,S are on the host, while pass B of the other tracers may still be running
// (the next call queues behind it).
// the device copy of the step's inputs that the coming step will read: the one after the current one if a step has been
// queued on that (three copies, made on the first call as images of the fields a plain upload fills)
static int inputs_next_copy(uvic_gpu *h) {
  auto &I = h->in;
  const uvic_dims &d = h->d;
  const size_t N3 = (size_t)d.imt * d.km * d.jmt, NF = (size_t)d.imt * (d.km + 1) * d.jmt, N2 = (size_t)d.imt * d.jmt;
  const size_t bytes[5] = {N3 * 8, N3 * 8, NF * 8, N2 * d.nt * 8, N2 * d.nt * 8};
  if (!I.st[0]) {   // (rows no upload covers keep their values)
    for (int q = 0; q < 5; ++q) {
      I.set[0][q] = (double *)h->buf[IN_FIELDS[q]];
      for (int z = 1; z < 3; ++z) {
        HIPCHK(hipMalloc((void **)&I.set[z][q], bytes[q]));
        HIPCHK(hipMemcpyAsync(I.set[z][q], I.set[0][q], bytes[q], hipMemcpyDeviceToDevice, h->stream));
      }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int q = 0; q < I.nstreams; ++q) HIPCHK(hipStreamCreateWithFlags(&I.st[q], hipStreamNonBlocking));
    if (I.nstreams == 1) I.st[1] = nullptr;
    for (int q = 0; q < 2; ++q) HIPCHK(hipEventCreateWithFlags(&I.ev_link[q], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&I.ev_first, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&I.ev_rest, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&I.ev_vel, hipEventDisableTiming));
    I.cur = 0; I.used = true;
  }
  if (I.used) {
    I.cur = (I.cur + 1) % 3; I.used = false;
    for (int q = 0; q < 5; ++q) h->buf[IN_FIELDS[q]] = I.set[I.cur][q];
    bind_ctx(h);
  }
  return 0;
}
// The step's inputs from the host in one call (the resident overlay: tracer_gpu.F), replacing six uvic_gpu_upload_rows.
// Host arrays as the memory window holds them: adv_vet (imt,km,jsmw:jmt), adv_vnt (imt,km,1:jmt), adv_vbt
// (imt,0:km,jsmw:jmt) or null, diff_cbt (imt,km,jsmw:jemw), stf and btf (imt,jmt,nt); page-locked (uvic_gpu_pin_host).
// The copies do not go through the main stream: they run on two streams of their own into the device copy the previous
// step does not read, T and S first -- the horizontal velocities, diff_cbt and the T,S planes of the fluxes, after which
// the T,S passes may start -- then the fluxes of the other tracers, which the bulk pass A waits for.  A null adv_vbt is
// formed on the device from adv_vet and adv_vnt, as adv_vel.F:98-127 does on the host (rigid lid: zero at the surface).
extern "C" int uvic_gpu_overlay_inputs(uvic_gpu *h, int jsmw, int jemw, const double *adv_vet, const double *adv_vnt, const double *adv_vbt,
                                       const double *diff_cbt, const double *stf, const double *btf) {
  if (!h || !stf || !btf || (!adv_vet) != (!adv_vnt)) return fail_msg("uvic_gpu_overlay_inputs: null argument");
  if (!diff_cbt) {   // vmixc on the device
    if (!h->have_vmix) return fail_msg("uvic_gpu_overlay_inputs: no diff_cbt given and no uvic_gpu_set_vmix_params for the device to form it");
    if (!h->ctx.diff_cbt_given) return fail_msg("uvic_gpu_overlay_inputs: no diff_cbt given: set uvic_params.diff_cbt_has_k33 = 1 (vmixc forms all of it)");
    if (int rc = vmix_tables(h)) return rc;
  }
  h->ctx.vmix_dev = diff_cbt ? 0 : 1;
  ovl_stamp("inputs_in");
  if (!adv_vet && !(h->in.vel_pending && !h->in.used))
    return fail_msg("uvic_gpu_overlay_inputs: no velocities given and none formed on the device for this step (uvic_gpu_overlay_velocities)");
  const uvic_dims &d = h->d;
  if (jsmw < 1 || jsmw > 2 || jemw < jsmw || jemw > d.jmt) return fail_msg("uvic_gpu_overlay_inputs: window rows outside 1..jmt");
  h->idle_until_next = false;
  HIPCHK(hipSetDevice(h->device));
  if (int rc = mom_host_join(h)) return rc;
  auto &I = h->in;
  const size_t N3 = (size_t)d.imt * d.km * d.jmt, NF = (size_t)d.imt * (d.km + 1) * d.jmt, N2 = (size_t)d.imt * d.jmt;
  if (int rc = inputs_next_copy(h)) return rc;
  hipStream_t sa = I.st[0], sb = I.st[1] ? I.st[1] : I.st[0];
  const size_t row = (size_t)d.imt * d.km * 8, rowf = (size_t)d.imt * (d.km + 1) * 8;
  double **dev = I.set[I.cur];
  auto up = [&](hipStream_t st, void *dst, const void *src, size_t n) { return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, st); };
  // (two copies at a time move 59 GB/s where one moves 47; each copy costs ~10 us of latency on its stream: the two
  // long ones on one stream, the third long one and the two short ones on the other)
  if (adv_vet) {
    HIPCHK(up(sa, (char *)dev[0] + (size_t)(jsmw - 1) * row, adv_vet, (size_t)(d.jmt - jsmw + 1) * row));
    HIPCHK(up(sb, dev[1], adv_vnt, (size_t)d.jmt * row));
    I.vel_pending = false;
  }
  if (diff_cbt) HIPCHK(up(sa, (char *)h->buf[UVIC_F_DIFF_CBT] + (size_t)(jsmw - 1) * row, diff_cbt, (size_t)(jemw - jsmw + 1) * row));
  HIPCHK(up(sb, dev[3], stf, 2 * N2 * 8));
  HIPCHK(up(sb, dev[4], btf, 2 * N2 * 8));
  if (adv_vbt && adv_vet) HIPCHK(up(sb, (char *)dev[2] + (size_t)(jsmw - 1) * rowf, adv_vbt, (size_t)(d.jmt - jsmw + 1) * rowf));
  if (sb != sa) { HIPCHK(hipEventRecord(I.ev_link[0], sb)); HIPCHK(hipStreamWaitEvent(sa, I.ev_link[0], 0)); }
  HIPCHK(hipEventRecord(I.ev_first, sa));
  if (d.nt > 2) {
    HIPCHK(up(sa, dev[3] + 2 * N2, stf + 2 * N2, (size_t)(d.nt - 2) * N2 * 8));
    HIPCHK(up(sb, dev[4] + 2 * N2, btf + 2 * N2, (size_t)(d.nt - 2) * N2 * 8));
    if (sb != sa) { HIPCHK(hipEventRecord(I.ev_link[1], sb)); HIPCHK(hipStreamWaitEvent(sa, I.ev_link[1], 0)); }
  }
  HIPCHK(hipEventRecord(I.ev_rest, sa));
  I.first_pending = I.rest_pending = true;
  I.rest_inflight = true;
  I.derive_vbt = adv_vbt == nullptr || adv_vet == nullptr;
  velocity_touched(h, UVIC_F_ADV_VET);
  ovl_stamp("inputs_out");
  return 0;
}
// the main stream waits for the inputs a step is about to read (every other stream of the step starts behind the main one)
static int inputs_first(uvic_gpu *h, bool vbt_follows) {
  h->in.used = true;
  if (!h->in.first_pending) return 0;
  HIPCHK(hipStreamWaitEvent(h->stream, h->in.ev_first, 0));
  if (h->in.vel_pending) { HIPCHK(hipStreamWaitEvent(h->stream, h->in.ev_vel, 0)); h->in.vel_pending = false; }
  h->in.first_pending = false;
  h->in.waited = true;
  if (h->in.derive_vbt && !vbt_follows) {
    hipLaunchKernelGGL(k_adv_vel_vert, dim3(col_blocks(h, 64)), dim3(64), 0, h->stream, h->ctx);
    mark(h, "adv_vel_vert");
  }
  return 0;
}
static int inputs_rest(uvic_gpu *h) {
  if (!h->in.rest_pending) return 0;
  HIPCHK(hipStreamWaitEvent(h->stream, h->in.ev_rest, 0));
  h->in.rest_pending = false;
  return 0;
}
extern "C" int uvic_gpu_overlay_step(uvic_gpu *h, const uvic_overlay_step *s, double *ts_host) {
  if (!h || !s) return fail_msg("uvic_gpu_overlay_step: null argument");
  HIPCHK(hipSetDevice(h->device));
  static const bool timing = uv_env("UVIC_OVL_TIMING") != nullptr;   // diagnosis: host wall time of the call's parts
  const auto tq0 = std::chrono::steady_clock::now();
  ovl_stamp("step_in");
  h->ts_host = ts_host;
  h->ts_host_queued = false;
  if (int rc = uvic_gpu_step_lookahead_at(h, s->c2dtts, s->mixing, s->mobi_ahead, s->c2dtts_next, s->relyr_next, s->co2ccn_next, s->iso_ahead))
    return rc;
  h->ts_host = nullptr;
  const auto tq1 = std::chrono::steady_clock::now();
  ovl_stamp("queued");
  if (h->sbc_count > 0 && (s->sbc_accumulate || s->sbc_zero)) {
    const long long n = (long long)h->d.imt * h->d.jmt * h->sbc_count;
    hipLaunchKernelGGL(k_sbc_accumulate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->ctx, (const int *)h->sbc_tracer,
                       h->sbc_acc, h->sbc_count, s->sbc_zero);
    HIPCHK(hipGetLastError());
  }
  if (ts_host) {
    if (h->ts_host_queued) {
      HIPCHK(hipEventSynchronize(h->ev_ts_host));
    } else {   // no T,S passes of their own in this configuration: behind the whole step
      HIPCHK(hipMemcpyAsync(ts_host, h->ctx.t_taup1, (size_t)2 * h->d.imt * h->d.km * h->d.jmt * 8, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
    }
  }
  // the caller's arrays are its own again when this returns: the last copies out of them and MOBI's fetch of its forcing
  // fields ended long before T and S came back, but nothing said so yet
  if (h->in.rest_inflight) { HIPCHK(hipEventSynchronize(h->in.ev_rest)); h->in.rest_inflight = false; }
  if (h->in.forcing_inflight) { HIPCHK(hipEventSynchronize(h->in.ev_forcing)); h->in.forcing_inflight = false; }
  if (timing) {
    ovl_stamp("returned");
    ovl_stamps_print();
    const auto tq2 = std::chrono::steady_clock::now();
    fprintf(stderr, "overlay_step: queueing %.3f ms, wait for T,S %.3f ms; row transfers before it %.3f ms\n",
            std::chrono::duration<double, std::milli>(tq1 - tq0).count(), std::chrono::duration<double, std::milli>(tq2 - tq1).count(), g_xfer_ms);
    g_xfer_ms = 0.0;
  }
  return uvic_gpu_rotate(h);
}
extern "C" int uvic_gpu_mobi(uvic_gpu *h) {
  if (!h) return fail_msg("null handle");
  if (!h->have_mobi) return fail_msg("uvic_gpu_mobi: call uvic_gpu_set_mobi first");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = launch_mobi(h)) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
