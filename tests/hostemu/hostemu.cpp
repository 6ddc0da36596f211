// tests/hostemu/hostemu.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Compiles the package's kernel sources (uvic2.9_amd/csrc/kernels_*.hpp) for the
// host with `HostEnv` (kenv.hpp): every barrier-delimited phase of a workgroup
// routine is run for tid = 0..nthreads-1 in a loop.  The build container has
// no GPU; this lets `pytest -m "not gpu"` check the kernel LOGIC (indexing, tile
// halos, cyclic wrap, evaluation order) bit-for-bit against the oracle.  The
// product never loads this library and has no CPU path (uvic2.9_amd/capi.py
// fails loudly when libuvic_gpu.so or a GPU is missing).
#include <pthread.h>

#include <cstdlib>
#include <thread>
#include <vector>

#include "../../uvic2.9_amd/csrc/kernels_fct.hpp"
#include "../../uvic2.9_amd/csrc/kernels_isopyc.hpp"
#include "../../uvic2.9_amd/csrc/kernels_mobi.hpp"
#include "../../uvic2.9_amd/csrc/kernels_prep.hpp"
#include "../../uvic2.9_amd/csrc/kernels_clinic.hpp"
#include "../../uvic2.9_amd/csrc/kernels_filter.hpp"
#include "../../uvic2.9_amd/csrc/filter_host.hpp"

using namespace uvic;

extern "C" int emu_ctx_size(void) { return (int)sizeof(uvic_ctx); }

extern "C" void emu_isopyc(const uvic_ctx *cp) {
  const uvic_ctx &c = *cp;
  // latitude-slab runs: the same row window as the GPU kernels (two rows beyond the slab)
  const int lo = c.js - 2, hi = c.je + 2;
#define IN(j) ((j) >= lo && (j) <= hi)
  for (int j = 1; j <= c.jmt; ++j)
    for (int k = 1; k <= c.km; ++k)
      for (int i = 2; i <= c.imt - 1; ++i) if (IN(j)) isopyc_elements_cell(c, i, k, j);
  for (int j = 1; j <= c.jmt - 1; ++j)
    for (int k = 1; k <= c.km; ++k)
      for (int i = 2; i <= c.imt - 1; ++i) if (IN(j)) isopyc_ai_cell(c, i, k, j);
  for (int j = 1; j <= c.jmt - 1; ++j)
    for (int k = 1; k <= c.km; ++k)
      for (int i = 1; i <= c.imt; ++i) if (IN(j)) isopyc_adv_cell(c, i, k, j);
  for (int j = 2; j <= c.jmt - 1; ++j)
    for (int i = 2; i <= c.imt - 1; ++i) if (IN(j)) isopyc_column(c, i, j);
#undef IN
}

extern "C" void emu_transport(const uvic_ctx *cp, int nchunk, int nthreads) {
  const uvic_ctx &c = *cp;
  const int per = (c.imt - 2 + nchunk - 1) / nchunk, W = per + 4;
  std::vector<double> lds((size_t)W * c.km * 6 + (size_t)W * (c.km + 1) * 2, -7.0e33);
  HostEnv env{nthreads};
  const int r0 = c.js - 1 < 2 ? 2 : c.js - 1, r1 = c.je + 1 > c.jmt - 1 ? c.jmt - 1 : c.je + 1;
  for (int r = r0; r <= r1; ++r)
    for (int n = c.n0 + 1; n <= c.n0 + c.nt_local; ++n)
      for (int ch = 0; ch < nchunk; ++ch) {
        std::fill(lds.begin(), lds.end(), -7.0e33);   // poison: reads of unwritten LDS show up
        fct_rows_block(env, c, n, r, ch, nchunk, lds.data());
      }
  for (int j = c.js; j <= c.je; ++j)
    for (int n = c.n0 + 1; n <= c.n0 + c.nt_local; ++n)
      for (int ch = 0; ch < nchunk; ++ch) {
        std::fill(lds.begin(), lds.end(), -7.0e33);
        update_rows_block(env, c, n, j, ch, nchunk, lds.data());
      }
}

extern "C" void emu_convect(const uvic_ctx *cp) {
  const uvic_ctx &c = *cp;
  for (int j = c.js; j <= c.je; ++j)
    for (int i = 2; i <= c.imt - 1; ++i) convect_column(c, i, j);
}

// the two-pass form used on the GPU
extern "C" void emu_convect_twopass(const uvic_ctx *cp) {
  const uvic_ctx &c = *cp;
  std::vector<double> col(2 * (size_t)c.km);
  for (int j = c.js; j <= c.je; ++j)
    for (int i = 2; i <= c.imt - 1; ++i) convect_ts_column(c, i, j, col.data(), col.data() + c.km, 1);
  for (int n = 3; n <= c.nt; ++n)
    for (int j = c.js; j <= c.je; ++j)
      for (int i = 2; i <= c.imt - 1; ++i) convect_apply_cell(c, i, j, n);
}

// producers of the shared inputs (kernels_prep.hpp)
extern "C" void emu_adv_vel(const uvic_ctx *cp) {
  const uvic_ctx &c = *cp;
  for (int j = 1; j <= c.jmt; ++j)
    for (int k = 1; k <= c.km; ++k)
      for (int i = 1; i <= c.imt; ++i) adv_vel_hor_cell(c, i, k, j);
  for (int j = 2; j <= c.jmt; ++j)
    for (int i = 2; i <= c.imt - 1; ++i) adv_vel_vert_column(c, i, j);
}
extern "C" void emu_vmixc(const uvic_ctx *cp) {
  const uvic_ctx &c = *cp;
  for (int j = 2; j <= c.jmt - 1; ++j)
    for (int k = 1; k <= c.km; ++k)
      for (int i = 2; i <= c.imt - 1; ++i) vmixc_cell(c, i, k, j);
}

// polar filter: the library's host set-up + the workgroup routine under HostEnv
extern "C" int emu_filt(const uvic_ctx *cp, double pi, int jfrst, int jft0, int jft1, int jft2, int lsegf, int nthreads) {
  const uvic_ctx &c = *cp;
  FilterSetup fs;
  std::string err;
  if (filter_build(c.imt, c.jmt, c.km, c.kmt, c.cst, c.cstr, pi, jfrst, jft0, jft1, jft2, lsegf, fs, err)) return -1;
  std::vector<double> lds((size_t)2 * nthreads + 4);
  HostEnv env{nthreads};
  for (int n = c.n0 + 1; n <= c.n0 + c.nt_local; ++n)
    for (const FilterItem &it : fs.items) {
      if (it.j < c.js || it.j > c.je) continue;
      if (it.im > nthreads) return -2;
      std::fill(lds.begin(), lds.end(), -7.0e33);
      filt_block(env, c, it, n, fs.mats.data(), lds.data());
    }
  return (int)fs.items.size();
}

// 0: the carbonate solve evaluates the reference's expression (bit-exact against the oracle); 1: the form the
// device uses by default, with shared reciprocals (agrees to rounding)
static int g_carb_shared = 0;
extern "C" void emu_set_carb_shared(int on) { g_carb_shared = on; }
// MOBI column kernel on the host: same source as the GPU kernel, libm instead of ocml
extern "C" void emu_mobi(const uvic_ctx *cp, const uvic_mobi_params *P, const uvic_mobi_forcing *F) {
  const uvic_ctx &c = *cp;
  mobi_dev M;
  M.P = P;
  M.carb_shared = g_carb_shared;
  M.tlat = F->tlat; M.dnswr = F->dnswr; M.aice = F->aice; M.hice = F->hice; M.hsno = F->hsno;
  M.sg_bathy = F->sg_bathy; M.fe_atmdep = F->fe_atmdep; M.fe_hydr = F->fe_hydr;
  M.pi = F->pi; M.radian = F->radian; M.relyr = F->relyr; M.co2ccn = F->co2ccn;
  mobi_step &S = M.S;
  S.nbio = (int)(c.c2dtts / P->dtnpzd);
  S.dtbio = c.c2dtts / S.nbio;
  S.rdtts = 1. / c.c2dtts;
  S.rnbio = 1. / S.nbio;
  const double yrtime = fmod(F->relyr, 1.);
  S.month = 12;
  for (int m = 1; m <= 12; ++m)
    if (yrtime <= m / 12.) { S.month = m; break; }
  S.declin = sin((fmod(F->relyr, 1.) - 0.22) * 2. * F->pi) * 0.4;
  std::vector<double> work(mobi_work_doubles(c.imt, c.jmt, c.km), 0.0);
  mobi_set_work(&M, work.data(), c.imt, c.jmt, c.km);
  for (int j = c.js; j <= c.je; ++j)
    for (int k = 1; k <= c.km; ++k)
      for (int i = 2; i <= c.imt - 1; ++i) mobi_pre_cell(c, M, i, k, j);
  for (int j = c.js; j <= c.je; ++j)
    for (int i = 2; i <= c.imt - 1; ++i) mobi_column_kernel(c, M, i, j);
  for (int j = c.js; j <= c.je; ++j)
    for (int k = 1; k <= c.km; ++k)
      for (int i = 2; i <= c.imt - 1; ++i) mobi_post_cell(c, M, i, k, j);
}


// ---- MOBI column kernel in TEAM form: four "waves" x 64 lanes per team, real threads and a
// pthread barrier stand in for the workgroup (state lives in registers across barriers, so the
// phase-loop emulation of HostEnv does not apply here)
template <int R>
struct HostTeam {
  static constexpr bool team = true;
  static constexpr int role = R;
  int lane;
  double *xs;
  unsigned xc;
  pthread_barrier_t *bar;
  void sync() { pthread_barrier_wait(bar); }
};

extern "C" void emu_mobi_team(const uvic_ctx *cp, const uvic_mobi_params *P, const uvic_mobi_forcing *F) {
  const uvic_ctx &c = *cp;
  mobi_dev M;
  M.P = P;
  M.carb_shared = g_carb_shared;
  M.tlat = F->tlat; M.dnswr = F->dnswr; M.aice = F->aice; M.hice = F->hice; M.hsno = F->hsno;
  M.sg_bathy = F->sg_bathy; M.fe_atmdep = F->fe_atmdep; M.fe_hydr = F->fe_hydr;
  M.pi = F->pi; M.radian = F->radian; M.relyr = F->relyr; M.co2ccn = F->co2ccn;
  mobi_step &S = M.S;
  S.nbio = (int)(c.c2dtts / P->dtnpzd);
  S.dtbio = c.c2dtts / S.nbio;
  S.rdtts = 1. / c.c2dtts;
  S.rnbio = 1. / S.nbio;
  const double yrtime = fmod(F->relyr, 1.);
  S.month = 12;
  for (int m = 1; m <= 12; ++m)
    if (yrtime <= m / 12.) { S.month = m; break; }
  S.declin = sin((fmod(F->relyr, 1.) - 0.22) * 2. * F->pi) * 0.4;
  std::vector<double> work(mobi_work_doubles(c.imt, c.jmt, c.km), 0.0);
  mobi_set_work(&M, work.data(), c.imt, c.jmt, c.km);
  for (int j = c.js; j <= c.je; ++j)
    for (int k = 1; k <= c.km; ++k)
      for (int i = 2; i <= c.imt - 1; ++i) mobi_pre_cell(c, M, i, k, j);
  const int ncol = c.imt * c.jmt;
  for (int g0 = 0; g0 < ncol; g0 += 64) {
    auto decode = [&](int lane, int &i, int &j) {
      const int gid = g0 + lane;
      i = gid % c.imt + 1; j = gid / c.imt + 1;
      return gid < ncol && j >= c.js && j <= c.je && i >= 2 && i <= c.imt - 1;
    };
    int kmax = 0;
    for (int lane = 0; lane < 64; ++lane) {
      int i, j;
      if (decode(lane, i, j)) kmax = std::max(kmax, c.kmt[(size_t)(i - 1) + (size_t)c.imt * (j - 1)]);
    }
    if (kmax == 0) continue;  // nothing but land
    std::vector<double> xs(UV_MOBI_LDS_DOUBLES, -3.0e33);
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, 256);
    std::vector<std::thread> th;
    for (int wave = 0; wave < 4; ++wave)
      for (int lane = 0; lane < 64; ++lane)
        th.emplace_back([&, wave, lane] {
          int i, j;
          const bool live = decode(lane, i, j);
          if (!live) { i = 2; j = c.js; }
          switch (wave) {
            case 0: { HostTeam<0> T{lane, xs.data(), 0u, &bar}; mobi_column_body(T, c, M, i, j, live, kmax); break; }
            case 1: { HostTeam<1> T{lane, xs.data(), 0u, &bar}; mobi_column_body(T, c, M, i, j, live, kmax); break; }
            case 2: { HostTeam<2> T{lane, xs.data(), 0u, &bar}; mobi_column_body(T, c, M, i, j, live, kmax); break; }
            default: { HostTeam<3> T{lane, xs.data(), 0u, &bar}; mobi_column_body(T, c, M, i, j, live, kmax); break; }
          }
        });
    for (auto &t : th) t.join();
    pthread_barrier_destroy(&bar);
  }
  for (int j = c.js; j <= c.je; ++j)
    for (int k = 1; k <= c.km; ++k)
      for (int i = 2; i <= c.imt - 1; ++i) mobi_post_cell(c, M, i, k, j);
}

// baroclinic momentum step (kernels_clinic.hpp): the launch order of uvic_gpu_state / uvic_gpu_clinic
extern "C" int emu_mom_ctx_size(void) { return (int)sizeof(uvic_mom_ctx); }
extern "C" void emu_state(const uvic_mom_ctx *mp) {
  const uvic_mom_ctx &m = *mp;
  for (int j = 2; j <= m.jmt; ++j)
    for (int k = 1; k <= m.km; ++k)
      for (int i = 1; i <= m.imt; ++i) state_cell(m, i, k, j);
}
extern "C" void emu_clinic(const uvic_mom_ctx *mp) {
  const uvic_mom_ctx &m = *mp;
  for (int j = m.js; j <= m.je; ++j)
    for (int i = 2; i <= m.imt - 1; ++i) clinic_gradp_column(m, i, j);
  for (int j = m.js; j <= m.je; ++j)
    for (int k = 1; k <= m.km; ++k)
      for (int i = 2; i <= m.imt - 1; ++i) clinic_tend_cell(m, i, k, j);
  for (int j = m.js; j <= m.je; ++j)
    for (int i = 2; i <= m.imt - 1; ++i) clinic_finish_column(m, i, j);
}
extern "C" void emu_add_ext_mode(const uvic_mom_ctx *mp, const double *psi, double *u1, double *u2) {
  const uvic_mom_ctx &m = *mp;
  for (int j = 1; j <= m.jmt - 1; ++j)
    for (int i = 2; i <= m.imt - 1; ++i) add_ext_mode_column(m, i, j, psi, u1, u2);
}
extern "C" void emu_sbcu(const uvic_mom_ctx *mp, int flags, double rts) {
  const uvic_mom_ctx &m = *mp;
  for (int j = m.js; j <= m.je; ++j)
    for (int i = 2; i <= m.imt - 1; ++i) clinic_sbcu_cell(m, i, j, flags, rts);
}

// polar filter of the velocities: the host-side set-up of the library and its two kernels
extern "C" int emu_filuv(const uvic_mom_ctx *mp, const double *csu, const double *phi, const double *spsin, const double *spcos, double pi,
                         int jfrst, int jfu0, int jfu1, int jfu2, int lsegf, int nthreads) {
  const uvic_mom_ctx &m = *mp;
  FilterSetup fs;
  std::vector<int> rows;
  std::string err;
  if (int rc = filter_build_u(m.imt, m.jmt, m.km, m.kmu, csu, m.csur, phi, pi, jfrst, jfu0, jfu1, jfu2, lsegf, fs, rows, err)) return rc;
  std::vector<double> lds((size_t)4 * nthreads + 8);
  HostEnv env{2 * nthreads};
  for (auto &it : fs.items) {
    if (it.im > nthreads) return 3;
    std::fill(lds.begin(), lds.end(), -7.0e33);
    filuv_block(env, m.imt, m.km, it, fs.mats.data(), spsin, spcos, m.up1, m.up2, lds.data());
  }
  for (int j : rows)
    for (int i = 2; i <= m.imt - 1; ++i) filuv_mean_column(m.imt, m.km, i, j, m.kmu, m.hr, m.dzt, m.up1, m.up2);
  return 0;
}
