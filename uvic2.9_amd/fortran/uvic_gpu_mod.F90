!=======================================================================
!     uvic_gpu_mod -- ISO_C_BINDING interface of include/uvic_gpu.h
!
!     Bound by the overlay tracer_gpu.F.  Every dummy is either a C
!     scalar by value or the base address of a COMMON array (assumed
!     size), exactly what the C side declares: no descriptors cross the
!     boundary.  Field ids are the enumerators of `enum uvic_field`.
!=======================================================================
      module uvic_gpu_mod
      use iso_c_binding
      implicit none

      type, bind(C) :: uvic_dims
        integer(c_int32_t) :: imt, jmt, km, nt, nsrc, ntnpzd
      end type uvic_dims

!     one resident step (include/uvic_gpu.h: uvic_overlay_step)
      type, bind(C) :: uvic_overlay_step
        real(c_double) :: c2dtts, c2dtts_next, relyr_next, co2ccn_next
        integer(c_int32_t) :: mixing, mobi_ahead, iso_ahead, sbc_zero
        integer(c_int32_t) :: sbc_accumulate, pad
      end type uvic_overlay_step

      type, bind(C) :: uvic_params
        real(c_double) :: c2dtts, aidif, diff_cet, diff_cnt
        real(c_double) :: slmxr, ahisop, athkdf
        integer(c_int32_t) :: diff_cbt_has_k33, pad
      end type uvic_params

!     enum uvic_field (include/uvic_gpu.h) -- keep the order
      integer(c_int), parameter :: F_DXT=0, F_DXTR=1, F_DXU=2, F_DXUR=3
      integer(c_int), parameter :: F_DXT4R=4, F_DYT=5, F_DYTR=6, F_DYU=7
      integer(c_int), parameter :: F_DYUR=8, F_DYT4R=9, F_CST=10
      integer(c_int), parameter :: F_CSTR=11, F_CSU=12, F_CSTDYTR=13
      integer(c_int), parameter :: F_CSTDYT2R=14, F_CSU_DYUR=15
      integer(c_int), parameter :: F_DZT=16, F_DZTR=17, F_DZT2R=18
      integer(c_int), parameter :: F_DZTUR=19, F_DZTLR=20, F_DZW=21
      integer(c_int), parameter :: F_DZWR=22, F_DTXCEL=23, F_DTXSQR=24
      integer(c_int), parameter :: F_DZTXCL=25, F_TO=26, F_SO=27, F_C=28
      integer(c_int), parameter :: F_KMT=29, F_FISOP=30, F_ADDISOP=31
      integer(c_int), parameter :: F_T_TAUM1=32, F_T_TAU=33
      integer(c_int), parameter :: F_T_TAUP1=34, F_ADV_VET=35
      integer(c_int), parameter :: F_ADV_VNT=36, F_ADV_VBT=37
      integer(c_int), parameter :: F_DIFF_CBT_BG=38, F_STF=39, F_BTF=40
      integer(c_int), parameter :: F_SRC=41, F_ITRC=42
      integer(c_int), parameter :: F_DIFF_CBT=58
!     adv_vel on the device (uvic_gpu_overlay_velocities)
      integer(c_int), parameter :: F_DXT2R=61, F_DYT2R=62
!     depth of the T-cell bottoms: vdepth of the convection diagnostics (uvic_gpu_set_tavg)
      integer(c_int), parameter :: F_ZW=63
!     vmixc on the device (UVIC_RESIDENT=3: mixing_gpu.F): latitudes and tidal energy dissipation rates (tidal_kv.h)
      integer(c_int), parameter :: F_TLAT=64, F_EDRM2=65, F_EDRS2=66
      integer(c_int), parameter :: F_EDRK1=67, F_EDRO1=68
!     baroclinic momentum step (clinic_gpu.F)
      integer(c_int), parameter :: F_U1=59, F_U2=60, F_RHO=69
      integer(c_int), parameter :: F_UM1=70, F_UM2=71, F_UP1=72, F_UP2=73
      integer(c_int), parameter :: F_ZU=74, F_GRAD_P=75, F_SMF=76
      integer(c_int), parameter :: F_KMU=77, F_HR=78, F_CORI=79
      integer(c_int), parameter :: F_VISC_CEU=80, F_AMC_NORTH=81
      integer(c_int), parameter :: F_AMC_SOUTH=82, F_DXU2R=83
      integer(c_int), parameter :: F_DXMETR=84, F_DUW=85, F_DUE=86
      integer(c_int), parameter :: F_DYU2R=87, F_DYU4R=88, F_CSUR=89
      integer(c_int), parameter :: F_DUS=90, F_DUN=91, F_CSUDYU2R=92
      integer(c_int), parameter :: F_ADVMET=93, F_AM3=94, F_AM4=95
      integer(c_int), parameter :: F_SBC_GU=96, F_SBC_GV=97, F_SBC_SU=98
      integer(c_int), parameter :: F_SBC_SV=99, F_SPSIN=100, F_SPCOS=101
      integer(c_int), parameter :: F_PHI=102, F_PSI=103

!     scalars of vmixc (include/uvic_gpu.h: uvic_vmix_params)
      type, bind(C) :: uvic_vmix_params
        real(c_double) :: kappa_h, zetar, ogamma, gravrho0r
      end type uvic_vmix_params

!     scalars of clinic (include/uvic_gpu.h: uvic_clinic_params)
      type, bind(C) :: uvic_clinic_params
        real(c_double) :: c2dtuv, grav, rho0r, kappa_m, cdbot
      end type uvic_clinic_params

      interface
        function uvic_gpu_create(h, dims, device) bind(C,name='uvic_gpu_create') result(rc)
          import
          type(c_ptr) :: h
          type(uvic_dims) :: dims
          integer(c_int), value :: device
          integer(c_int) :: rc
        end function
        function uvic_gpu_destroy(h) bind(C,name='uvic_gpu_destroy') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int) :: rc
        end function
        function uvic_gpu_upload(h, field, host, offset, count) bind(C,name='uvic_gpu_upload') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int), value :: field
          type(*) :: host(*)
          integer(c_int64_t), value :: offset, count
          integer(c_int) :: rc
        end function
        function uvic_gpu_download(h, field, host, offset, count) bind(C,name='uvic_gpu_download') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int), value :: field
          type(*) :: host(*)
          integer(c_int64_t), value :: offset, count
          integer(c_int) :: rc
        end function
        function uvic_gpu_upload_rows(h, field, host, jlo, jhi) bind(C,name='uvic_gpu_upload_rows') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int), value :: field, jlo, jhi
          real(c_double) :: host(*)
          integer(c_int) :: rc
        end function
        function uvic_gpu_download_rows(h, field, host, jlo, jhi) bind(C,name='uvic_gpu_download_rows') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int), value :: field, jlo, jhi
          real(c_double) :: host(*)
          integer(c_int) :: rc
        end function
        function uvic_gpu_set_params(h, p) bind(C,name='uvic_gpu_set_params') result(rc)
          import
          type(c_ptr), value :: h
          type(uvic_params) :: p
          integer(c_int) :: rc
        end function
        function uvic_gpu_set_mobi_flat(h, km, ntnpzd, nsrc, idx, tom, som, itr, scal, prof, fsc, tlat, dnswr,   &
     &      aice, hice, hsno, sgb, fe_atmdep, fe_hydr) bind(C,name='uvic_gpu_set_mobi_flat') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int), value :: km, ntnpzd, nsrc
          integer(c_int32_t) :: idx(*), tom(*), som(*), itr(*)
          real(c_double) :: scal(*), prof(*), fsc(*), tlat(*), dnswr(*), aice(*), hice(*), hsno(*)
          real(c_double) :: sgb(*), fe_atmdep(*), fe_hydr(*)
          integer(c_int) :: rc
        end function
        function uvic_gpu_mobi_options_flat(h, flags, im, is, isx, oscal, wc, wo, km)                              &
     &      bind(C,name='uvic_gpu_mobi_options_flat') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int32_t) :: flags(*), im(*), is(*), isx(*)
          real(c_double) :: oscal(*), wc(*), wo(*)
          integer(c_int), value :: km
          integer(c_int) :: rc
        end function
        function uvic_gpu_set_filter(h, pi, jfrst, jft0, jft1, jft2, lsegf) bind(C,name='uvic_gpu_set_filter') result(rc)
          import
          type(c_ptr), value :: h
          real(c_double), value :: pi
          integer(c_int), value :: jfrst, jft0, jft1, jft2, lsegf
          integer(c_int) :: rc
        end function
        function uvic_gpu_isopyc(h) bind(C,name='uvic_gpu_isopyc') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int) :: rc
        end function
        function uvic_gpu_tracer(h) bind(C,name='uvic_gpu_tracer') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int) :: rc
        end function
        function uvic_gpu_download_level(h, field, n, k, host) bind(C,name='uvic_gpu_download_level') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int), value :: field, n, k
          real(c_double) :: host(*)
          integer(c_int) :: rc
        end function
        function uvic_gpu_set_mobi_step(h, relyr, co2ccn, dnswr, aice, hice, hsno)                             &
     &      bind(C,name='uvic_gpu_set_mobi_step') result(rc)
          import
          type(c_ptr), value :: h
          real(c_double), value :: relyr, co2ccn
          real(c_double) :: dnswr(*), aice(*), hice(*), hsno(*)
          integer(c_int) :: rc
        end function
        function uvic_gpu_pin_host(h, host, bytes) bind(C,name='uvic_gpu_pin_host') result(rc)
          import
          type(c_ptr), value :: h
          type(*) :: host(*)
          integer(c_int64_t), value :: bytes
          integer(c_int) :: rc
        end function
        function uvic_gpu_step_lookahead(h, c2dtts, mixing, mobi_ahead, c2dtts_next, iso_ahead)              &
     &      bind(C,name='uvic_gpu_step_lookahead') result(rc)
          import
          type(c_ptr), value :: h
          real(c_double), value :: c2dtts, c2dtts_next
          integer(c_int), value :: mixing, mobi_ahead, iso_ahead
          integer(c_int) :: rc
        end function
        function uvic_gpu_set_mixing(h, on) bind(C,name='uvic_gpu_set_mixing') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int), value :: on
          integer(c_int) :: rc
        end function
        function uvic_gpu_rotate(h) bind(C,name='uvic_gpu_rotate') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int) :: rc
        end function
        function uvic_gpu_set_host_sync(h, on) bind(C,name='uvic_gpu_set_host_sync') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int), value :: on
          integer(c_int) :: rc
        end function
        function uvic_gpu_sbc_config(h, count, tracers) bind(C,name='uvic_gpu_sbc_config') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int), value :: count
          integer(c_int32_t) :: tracers(*)
          integer(c_int) :: rc
        end function
        function uvic_gpu_sbc_transfer(h, host, upload) bind(C,name='uvic_gpu_sbc_transfer') result(rc)
          import
          type(c_ptr), value :: h
          real(c_double) :: host(*)
          integer(c_int), value :: upload
          integer(c_int) :: rc
        end function
        function uvic_gpu_set_tsi(h, on, ic14, idic) bind(C,name='uvic_gpu_set_tsi') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int), value :: on, ic14, idic
          integer(c_int) :: rc
        end function
        function uvic_gpu_tsi_read(h, tbar, travar, dtabs, dc14bar) bind(C,name='uvic_gpu_tsi_read') result(rc)
          import
          type(c_ptr), value :: h
          real(c_double) :: tbar(*), travar(*), dtabs(*), dc14bar
          integer(c_int) :: rc
        end function
        function uvic_gpu_tsi_ektot(h, rho0, ektot) bind(C,name='uvic_gpu_tsi_ektot') result(rc)
          import
          type(c_ptr), value :: h
          real(c_double), value :: rho0
          real(c_double) :: ektot(*)
          integer(c_int) :: rc
        end function
        function uvic_gpu_overlay_inputs(h, jsmw, jemw, adv_vet, adv_vnt, adv_vbt, diff_cbt, stf, btf)       &
     &      bind(C,name='uvic_gpu_overlay_inputs') result(rc)
!         adv_vbt = c_null_ptr: formed on the device from adv_vet, adv_vnt (rigid lid)
          import
          type(c_ptr), value :: h
          integer(c_int), value :: jsmw, jemw
          real(c_double) :: adv_vet(*), adv_vnt(*), diff_cbt(*), stf(*), btf(*)
          type(c_ptr), value :: adv_vbt
          integer(c_int) :: rc
        end function
        function uvic_gpu_overlay_inputs_dev(h, jsmw, jemw, p1, p2, p3, diff_cbt, stf, btf)                   &
     &      bind(C,name='uvic_gpu_overlay_inputs') result(rc)
!         the same entry point with no velocities (p1 = p2 = p3 = c_null_ptr): uvic_gpu_overlay_velocities made them
          import
          type(c_ptr), value :: h
          integer(c_int), value :: jsmw, jemw
          type(c_ptr), value :: p1, p2, p3
          real(c_double) :: diff_cbt(*), stf(*), btf(*)
          integer(c_int) :: rc
        end function
        function uvic_gpu_overlay_inputs_p(h, jsmw, jemw, p1, p2, p3, pcbt, stf, btf)                        &
     &      bind(C,name='uvic_gpu_overlay_inputs') result(rc)
!         the same entry point by addresses: pcbt = c_null_ptr has the device form diff_cbt itself (vmixc there)
          import
          type(c_ptr), value :: h
          integer(c_int), value :: jsmw, jemw
          type(c_ptr), value :: p1, p2, p3, pcbt
          real(c_double) :: stf(*), btf(*)
          integer(c_int) :: rc
        end function
        function uvic_gpu_set_vmix_params(h, p) bind(C,name='uvic_gpu_set_vmix_params') result(rc)
          import
          type(c_ptr), value :: h
          type(uvic_vmix_params) :: p
          integer(c_int) :: rc
        end function
        function uvic_gpu_overlay_inputs_vbt(h, jsmw, jemw, adv_vet, adv_vnt, adv_vbt, diff_cbt, stf, btf)   &
     &      bind(C,name='uvic_gpu_overlay_inputs') result(rc)
!         the same entry point with adv_vbt sent as well (free surface: not zero at the top)
          import
          type(c_ptr), value :: h
          integer(c_int), value :: jsmw, jemw
          real(c_double) :: adv_vet(*), adv_vnt(*), adv_vbt(*), diff_cbt(*), stf(*), btf(*)
          integer(c_int) :: rc
        end function
        function uvic_gpu_set_tavg(h, on, grav, zt, ic14, idic) bind(C,name='uvic_gpu_set_tavg') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int), value :: on, ic14, idic
          real(c_double), value :: grav
          real(c_double) :: zt(*)
          integer(c_int) :: rc
        end function
        function uvic_gpu_tavg_read(h, totalk, vdepth, pe, dc14) bind(C,name='uvic_gpu_tavg_read') result(rc)
          import
          type(c_ptr), value :: h
          real(c_double) :: totalk(*), vdepth(*), pe(*)
          type(c_ptr), value :: dc14
          integer(c_int) :: rc
        end function
        function uvic_gpu_overlay_velocities(h, ext_taum1, psi) bind(C,name='uvic_gpu_overlay_velocities') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int), value :: ext_taum1
          real(c_double) :: psi(*)
          integer(c_int) :: rc
        end function
        function uvic_gpu_overlay_momentum(h, fresh, t_level, sbc_flags, rts, rho0, smf, rho_host, zu_host, ektot_host)   &
     &      bind(C,name='uvic_gpu_overlay_momentum') result(rc)
!         rho_host, ektot_host: c_loc of the array or c_null_ptr
          import
          type(c_ptr), value :: h
          integer(c_int), value :: fresh, t_level, sbc_flags
          real(c_double), value :: rts, rho0
          real(c_double) :: smf(*), zu_host(*)
          type(c_ptr), value :: rho_host, ektot_host
          integer(c_int) :: rc
        end function
        function uvic_gpu_momentum_wait(h) bind(C,name='uvic_gpu_momentum_wait') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int) :: rc
        end function
        function uvic_gpu_rotate_u(h) bind(C,name='uvic_gpu_rotate_u') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int) :: rc
        end function
        function uvic_gpu_add_ext_mode(h, level) bind(C,name='uvic_gpu_add_ext_mode') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int), value :: level
          integer(c_int) :: rc
        end function
        function uvic_gpu_overlay_step(h, s, ts_host) bind(C,name='uvic_gpu_overlay_step') result(rc)
          import
          type(c_ptr), value :: h
          type(uvic_overlay_step) :: s
          real(c_double) :: ts_host(*)
          integer(c_int) :: rc
        end function
        function uvic_gpu_set_mobi_clock(h, relyr, co2ccn, p1, p2, p3, p4)                                    &
     &      bind(C,name='uvic_gpu_set_mobi_step') result(rc)
!         uvic_gpu_set_mobi_step with null forcing fields: only relyr and co2ccn move on
          import
          type(c_ptr), value :: h
          real(c_double), value :: relyr, co2ccn
          type(c_ptr), value :: p1, p2, p3, p4
          integer(c_int) :: rc
        end function
        function uvic_gpu_set_clinic_params(h, p) bind(C,name='uvic_gpu_set_clinic_params') result(rc)
          import
          type(c_ptr), value :: h
          type(uvic_clinic_params) :: p
          integer(c_int) :: rc
        end function
        function uvic_gpu_clinic(h, sbc_flags, rts) bind(C,name='uvic_gpu_clinic') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int), value :: sbc_flags
          real(c_double), value :: rts
          integer(c_int) :: rc
        end function
        function uvic_gpu_set_filter_u(h, pi, jfrst, jfu0, jfu1, jfu2, lsegf) bind(C,name='uvic_gpu_set_filter_u') result(rc)
          import
          type(c_ptr), value :: h
          real(c_double), value :: pi
          integer(c_int), value :: jfrst, jfu0, jfu1, jfu2, lsegf
          integer(c_int) :: rc
        end function
        function uvic_gpu_sync(h) bind(C,name='uvic_gpu_sync') result(rc)
          import
          type(c_ptr), value :: h
          integer(c_int) :: rc
        end function
        function uvic_gpu_last_error() bind(C,name='uvic_gpu_last_error') result(msg)
          import
          type(c_ptr) :: msg
        end function
      end interface

      type(c_ptr), save :: uvic_handle = c_null_ptr
!     what the overlays have set up on the device instance (module state, so that tracer_gpu_close can start over):
!     static fields of `tracer` / of `clinic` uploaded, MOBI bound, the filter rows the operators were built for
      logical, save :: uvic_tracer_ready = .false., uvic_clinic_ready = .false., uvic_mobi_set = .false.
      integer, save :: uvic_flt_rows(4) = (/ -1, -1, -1, -1 /), uvic_fltu_rows(4) = (/ -1, -1, -1, -1 /)
!     time step (itt) whose adv_vet/adv_vnt/adv_vbt the tracer overlay has put on the device: clinic_gpu.F, called
!     later in the same step (source/mom/mom.F:389-395), need not send them again
      integer, save :: uvic_adv_itt = -1
!     resident mode (environment UVIC_RESIDENT=1, read at the first call of the overlay): every tracer stays on
!     the device from step to step; uvic_dev_state says that the device holds t(tau-1), t(tau) of the coming step
      logical, save :: uvic_resident = .false.
      logical, save :: uvic_dev_state = .false.
!     ... and so do the velocities (UVIC_RESIDENT=2: uvic_resident_u) once clinic_gpu.F has made a time step on the
!     device: uvic_u_dev says that the device
!     holds the internal mode of u(tau+1) of step uvic_u_itt (and u(tau), u(tau-1) of that step, external mode included);
!     uvic_u_rot_itt is the step for which the device has rotated the levels and added the external mode (done by the
!     first overlay called in a step), uvic_u_host_itt the step for which the host's u(tau), u(tau-1) are what the
!     device holds; uvic_vel_dev_itt the step whose adv_v?t were formed on the device
      logical, save :: uvic_resident_u = .false.
!     ... and so do isopyc and vmixc (UVIC_RESIDENT=3: uvic_resident_mix; mixing_gpu.F): uvic_mix_skip_itt is the step
!     on which the host's two routines were left out, so that `tracer` sends no diff_cbt and the device forms its own
!     the atmosphere and ice fields MOBI reads change at a segment's first step; if that step went to the reference routine the
!     device has not seen them yet
      logical, save :: uvic_forcing_stale = .false.
      logical, save :: uvic_resident_mix = .false.
      integer, save :: uvic_mix_skip_itt = -1
!     ... and loadmw's `state` (rho of t(tau) for clinic, loadmw.F:154): left out on the step uvic_state_skip_itt; whoever
!     turns out to need the host's rho after all (clinic_gpu.F on the steps it hands to clinic_cpu or takes from the host's
!     arrays) calls uvic_host_rho first, which has the reference's routine run then (mixing_gpu.F: uvic_state_replay)
      integer, save :: uvic_state_skip_itt = -1
      procedure(), pointer, save :: uvic_state_replay_p => null()
      logical, save :: uvic_u_dev = .false.
      integer, save :: uvic_u_itt = -1, uvic_u_rot_itt = -1, uvic_u_host_itt = -1, uvic_vel_dev_itt = -1
!     the step after which the device holds the running sums of isbcu/asbcu (clinic.F:729-895)
      integer, save :: uvic_sbcu_itt = -1
      real(c_double), allocatable, target, save :: uvic_zu(:,:,:)
!     resident mode keeps the surface sums of set_sbc on the device for these tracers (n >= 3 with trsbcindex(n) /= 0)
      integer, save :: uvic_nsbc = 0
      integer(c_int32_t), allocatable, save :: uvic_sbc_tracer(:)
      real(c_double), allocatable, save :: uvic_sbc_plane(:,:,:)

      contains

      subroutine uvic_host_rho(itt)
!       the host's rho is about to be read: form it now if loadmw's `state` was left out on this step
        integer, intent(in) :: itt
        if (uvic_state_skip_itt .eq. itt .and. associated(uvic_state_replay_p)) then
          call uvic_state_replay_p()
        endif
        uvic_state_skip_itt = -1
      end subroutine

      function uvic_addr(a) result(p)
!       the address of a COMMON array (which has no TARGET attribute of its own) for an optional argument of the C ABI
        real(c_double), target :: a(*)
        type(c_ptr) :: p
        p = c_loc(a)
      end function

      subroutine uvic_check(rc, where)
        integer(c_int), intent(in) :: rc
        character(*), intent(in) :: where
        character(kind=c_char), pointer :: msg(:)
        integer :: n
        if (rc .eq. 0) return
        call c_f_pointer(uvic_gpu_last_error(), msg, [512])
        n = 1
        do while (n .lt. 512 .and. msg(n) .ne. c_null_char)
          n = n + 1
        enddo
        write (*,'(4a)') '=> Error in ', where, ': ', msg(1:n-1)
!       the reference's own error convention, updates/09/source/mom/tracer.F:1250
        stop '=>tracer (gpu)'
      end subroutine uvic_check

      end module uvic_gpu_mod
