! Fortran caller of the C ABI (include/atlas_amd.h), bound the way atlas_f binds atlas__Trans__* (interface blocks with
! bind(C), handles as type(c_ptr): src/atlas_f/trans/atlas_Trans_module.F90:156-177 for the constructor, the pointer calls of
! src/atlas/trans/detail/TransInterface.h:74-79 for the transforms).  Checks, as src/tests/trans/test_transgeneral.cc:829-839 does
! (rel-RMS 1e-13): unit spectral coefficients -> closed-form spherical harmonics on every point of F32 and O32; the IFS-style call
! invtrans(nb_scalar, sp, nb_vordiv, vor, div, gp) on a solid-body rotation with a scalar beside it; atlas_HaloExchange's
! setup / execute / execute_adjoint with the strides and extents atlas_f passes for field(nvar, nnodes).
!   test_trans_f --host-only    sizes, grids, error reporting (no GPU)
!   test_trans_f                + the transforms on the MI355X
! Built and run by tests/test_fortran_api.py (amdflang).
module atlas_amd_c_binding
  use, intrinsic :: iso_c_binding
  implicit none
  interface
    function atlas_amd__device_count() bind(C, name="atlas_amd__device_count") result(n)
      import :: c_int
      integer(c_int) :: n
    end function
    function atlas_amd__last_error() bind(C, name="atlas_amd__last_error") result(msg)
      import :: c_ptr
      type(c_ptr) :: msg
    end function
    function atlas_amd__Grid__new_gaussian(name) bind(C, name="atlas_amd__Grid__new_gaussian") result(grid)
      import :: c_ptr, c_char
      character(kind=c_char), dimension(*), intent(in) :: name
      type(c_ptr) :: grid
    end function
    subroutine atlas_amd__Grid__delete(grid) bind(C, name="atlas_amd__Grid__delete")
      import :: c_ptr
      type(c_ptr), value :: grid
    end subroutine
    function atlas_amd__Grid__ny(grid) bind(C, name="atlas_amd__Grid__ny") result(ny)
      import :: c_ptr, c_int
      type(c_ptr), value :: grid
      integer(c_int) :: ny
    end function
    function atlas_amd__Grid__size(grid) bind(C, name="atlas_amd__Grid__size") result(n)
      import :: c_ptr, c_int64_t
      type(c_ptr), value :: grid
      integer(c_int64_t) :: n
    end function
    function atlas_amd__Grid__nx(grid, nx) bind(C, name="atlas_amd__Grid__nx") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: grid
      integer(c_int), dimension(*) :: nx
      integer(c_int) :: rc
    end function
    function atlas_amd__Grid__y(grid, lat_deg) bind(C, name="atlas_amd__Grid__y") result(rc)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: grid
      real(c_double), dimension(*) :: lat_deg
      integer(c_int) :: rc
    end function
    function atlas_amd__Trans__new(grid, truncation) bind(C, name="atlas_amd__Trans__new") result(trans)
      import :: c_ptr, c_int
      type(c_ptr), value :: grid
      integer(c_int), value :: truncation
      type(c_ptr) :: trans
    end function
    function atlas_amd__Trans__new_config(grid, truncation, config, cache, cache_size) &
        & bind(C, name="atlas_amd__Trans__new_config") result(trans)
      import :: c_ptr, c_int, c_char, c_size_t
      type(c_ptr), value :: grid
      integer(c_int), value :: truncation
      character(kind=c_char), dimension(*), intent(in) :: config
      type(c_ptr), value :: cache
      integer(c_size_t), value :: cache_size
      type(c_ptr) :: trans
    end function
    subroutine atlas_amd__Trans__delete(trans) bind(C, name="atlas_amd__Trans__delete")
      import :: c_ptr
      type(c_ptr), value :: trans
    end subroutine
    function atlas_amd__Trans__truncation(trans) bind(C, name="atlas_amd__Trans__truncation") result(t)
      import :: c_ptr, c_int
      type(c_ptr), value :: trans
      integer(c_int) :: t
    end function
    function atlas_amd__Trans__nb_gridpoints(trans) bind(C, name="atlas_amd__Trans__nb_gridpoints") result(n)
      import :: c_ptr, c_int64_t
      type(c_ptr), value :: trans
      integer(c_int64_t) :: n
    end function
    function atlas_amd__Trans__nb_spectral_coefficients(trans) &
        & bind(C, name="atlas_amd__Trans__nb_spectral_coefficients") result(n)
      import :: c_ptr, c_int64_t
      type(c_ptr), value :: trans
      integer(c_int64_t) :: n
    end function
    function atlas_amd__Trans__invtrans_scalar(trans, nb_fields, scalar_spectra, scalar_fields) &
        & bind(C, name="atlas_amd__Trans__invtrans_scalar") result(rc)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: trans
      integer(c_int), value :: nb_fields
      real(c_double), dimension(*), intent(in) :: scalar_spectra
      real(c_double), dimension(*) :: scalar_fields
      integer(c_int) :: rc
    end function
    function atlas_amd__Trans__invtrans(trans, nb_scalar_fields, scalar_spectra, nb_vordiv_fields, vorticity_spectra, &
        & divergence_spectra, gp_fields) bind(C, name="atlas_amd__Trans__invtrans") result(rc)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: trans
      integer(c_int), value :: nb_scalar_fields, nb_vordiv_fields
      real(c_double), dimension(*), intent(in) :: scalar_spectra, vorticity_spectra, divergence_spectra
      real(c_double), dimension(*) :: gp_fields
      integer(c_int) :: rc
    end function
    function atlas_amd__Trans__invtrans_vordiv2wind(trans, nb_fields, vorticity_spectra, divergence_spectra, wind_fields) &
        & bind(C, name="atlas_amd__Trans__invtrans_vordiv2wind") result(rc)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: trans
      integer(c_int), value :: nb_fields
      real(c_double), dimension(*), intent(in) :: vorticity_spectra, divergence_spectra
      real(c_double), dimension(*) :: wind_fields
      integer(c_int) :: rc
    end function
    function atlas_amd__HaloExchange__new() bind(C, name="atlas_amd__HaloExchange__new") result(hx)
      import :: c_ptr
      type(c_ptr) :: hx
    end function
    subroutine atlas_amd__HaloExchange__delete(hx) bind(C, name="atlas_amd__HaloExchange__delete")
      import :: c_ptr
      type(c_ptr), value :: hx
    end subroutine
    function atlas_amd__HaloExchange__setup(hx, part, remote_idx, base, parsize) &
        & bind(C, name="atlas_amd__HaloExchange__setup") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: hx
      integer(c_int), dimension(*), intent(in) :: part, remote_idx
      integer(c_int), value :: base, parsize
      integer(c_int) :: rc
    end function
    function atlas_amd__HaloExchange__execute_strided_double(hx, field, var_strides, var_extents, var_rank) &
        & bind(C, name="atlas_amd__HaloExchange__execute_strided_double") result(rc)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: hx
      real(c_double), dimension(*) :: field
      integer(c_int), dimension(*), intent(in) :: var_strides, var_extents
      integer(c_int), value :: var_rank
      integer(c_int) :: rc
    end function
    function atlas_amd__HaloExchange__execute_adjoint_strided_double(hx, field, var_strides, var_extents, var_rank) &
        & bind(C, name="atlas_amd__HaloExchange__execute_adjoint_strided_double") result(rc)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: hx
      real(c_double), dimension(*) :: field
      integer(c_int), dimension(*), intent(in) :: var_strides, var_extents
      integer(c_int), value :: var_rank
      integer(c_int) :: rc
    end function
  end interface
contains
  function c_str(s) result(z)   ! null-terminated copy, as fckit's c_str
    character(len=*), intent(in) :: s
    character(kind=c_char), dimension(len_trim(s) + 1) :: z
    integer :: i
    do i = 1, len_trim(s)
      z(i) = s(i:i)
    end do
    z(len_trim(s) + 1) = c_null_char
  end function
end module atlas_amd_c_binding

program test_trans_f
  use, intrinsic :: iso_c_binding
  use atlas_amd_c_binding
  implicit none
  integer :: failures, before, iarg
  logical :: host_only
  character(len=64) :: arg
  real(c_double), parameter :: pi = 3.14159265358979323846264338327950288_c_double

  failures  = 0
  host_only = .false.
  do iarg = 1, command_argument_count()
    call get_command_argument(iarg, arg)
    if (trim(arg) == "--host-only") host_only = .true.
  end do

  before = failures
  call case_grids_and_sizes()
  call report("grids_and_sizes", before)
  before = failures
  call case_errors_are_reported()
  call report("errors_are_reported", before)
  if (.not. host_only) then
    if (atlas_amd__device_count() < 1) then
      print '(a)', "no HIP device"
      stop 2
    end if
    before = failures
    call case_invtrans_analytic("F32")
    call report("invtrans_analytic_F32", before)
    before = failures
    call case_invtrans_analytic("O32")
    call report("invtrans_analytic_O32", before)
    before = failures
    call case_invtrans_vordiv_with_scalar()
    call report("invtrans_vordiv_with_scalar", before)
    before = failures
    call case_halo_exchange()
    call report("halo_exchange", before)
  end if
  print '(i0,a)', failures, " failure(s)"
  if (failures /= 0) stop 1

contains

  subroutine expect(cond, what)
    logical, intent(in) :: cond
    character(len=*), intent(in) :: what
    if (.not. cond) then
      print '(2a)', "FAILED ", what
      failures = failures + 1
    end if
  end subroutine

  subroutine report(name, failures_before)
    character(len=*), intent(in) :: name
    integer, intent(in) :: failures_before
    if (failures == failures_before) then
      print '(2a)', "ok     ", name
    else
      print '(2a)', "FAILED ", name
    end if
  end subroutine

  ! normalised associated Legendre functions (1/2 * integral of P^2 over mu = 1: LegendrePolynomials.cc:29), closed forms
  function legendre_closed_form(n, m, phi) result(p)
    integer, intent(in) :: n, m
    real(c_double), intent(in) :: phi
    real(c_double) :: p, s, c
    s = sin(phi)
    c = cos(phi)
    p = 0
    select case (10 * n + m)
    case (0)
      p = 1
    case (10)
      p = sqrt(3._c_double) * s
    case (11)
      p = sqrt(1.5_c_double) * c
    case (20)
      p = sqrt(5._c_double) * (1.5_c_double * s * s - 0.5_c_double)
    case (21)
      p = sqrt(7.5_c_double) * s * c
    case (22)
      p = sqrt(15._c_double / 8._c_double) * c * c
    case (32)
      p = sqrt(105._c_double / 8._c_double) * c * c * s
    case (33)
      p = sqrt(35._c_double / 16._c_double) * c * c * c
    end select
  end function

  subroutine case_grids_and_sizes()
    type(c_ptr) :: grid
    integer(c_int), allocatable :: nx(:)
    real(c_double), allocatable :: y(:)
    integer(c_int) :: rc
    grid = atlas_amd__Grid__new_gaussian(c_str("O32"))
    call expect(c_associated(grid), "Grid O32")
    if (.not. c_associated(grid)) return
    call expect(atlas_amd__Grid__ny(grid) == 64, "O32 has 64 rows")
    call expect(atlas_amd__Grid__size(grid) == 5248_c_int64_t, "O32 has 5248 points")
    allocate (nx(64), y(64))
    rc = atlas_amd__Grid__nx(grid, nx)
    call expect(rc == 0 .and. nx(1) == 20 .and. nx(32) == 144 .and. nx(64) == 20, "O32 rows have 20 + 4 j points")
    rc = atlas_amd__Grid__y(grid, y)
    call expect(rc == 0 .and. y(1) > 87._c_double .and. abs(y(1) + y(64)) < 1e-12_c_double, "O32 latitudes north to south")
    call atlas_amd__Grid__delete(grid)
  end subroutine

  subroutine case_errors_are_reported()
    type(c_ptr) :: grid
    grid = atlas_amd__Grid__new_gaussian(c_str("Q7"))   ! no such grid: a null handle and a message, no abort
    call expect(.not. c_associated(grid), "unknown grid name gives a null handle")
    call expect(c_associated(atlas_amd__last_error()), "last_error is set")
  end subroutine

  subroutine case_invtrans_analytic(gridname)
    character(len=*), intent(in) :: gridname
    integer, parameter :: T = 31
    integer, parameter :: nm(2, 8) = reshape([0, 0, 1, 0, 1, 1, 2, 0, 2, 1, 2, 2, 3, 2, 3, 3], [2, 8])
    type(c_ptr) :: grid, trans
    integer(c_int), allocatable :: nx(:)
    real(c_double), allocatable :: y(:), sp(:), gp(:)
    integer :: case_nm(16), case_imag(16), nf, c, f, n, m, i, j, ny
    integer(c_int64_t) :: npts, nspec, pos, p
    integer(c_int) :: rc
    real(c_double) :: num, den, ref, pl, lam
    character(len=96) :: what

    grid = atlas_amd__Grid__new_gaussian(c_str(gridname))
    call expect(c_associated(grid), "grid "//gridname)
    if (.not. c_associated(grid)) return
    ! atlas_Trans(grid, nsmax, config) with config%set("type", "local")
    trans = atlas_amd__Trans__new_config(grid, int(T, c_int), c_str("type=local"), c_null_ptr, 0_c_size_t)
    call expect(c_associated(trans), "Trans "//gridname)
    if (.not. c_associated(trans)) return
    npts  = atlas_amd__Trans__nb_gridpoints(trans)
    nspec = atlas_amd__Trans__nb_spectral_coefficients(trans)
    call expect(atlas_amd__Trans__truncation(trans) == T, "truncation")
    call expect(nspec == int((T + 1) * (T + 2), c_int64_t), "nb_spectral_coefficients = (T+1)(T+2)")
    call expect(npts == atlas_amd__Grid__size(grid), "nb_gridpoints = grid size")
    ny = atlas_amd__Grid__ny(grid)
    allocate (nx(ny), y(ny))
    rc = atlas_amd__Grid__nx(grid, nx)
    rc = atlas_amd__Grid__y(grid, y)

    nf = 0   ! every case as a field of ONE call: real part, and the imaginary part where m > 0
    do c = 1, 8
      nf = nf + 1
      case_nm(nf) = c
      case_imag(nf) = 0
      if (nm(2, c) > 0) then
        nf = nf + 1
        case_nm(nf) = c
        case_imag(nf) = 1
      end if
    end do
    allocate (sp(nspec * nf), gp(npts * nf))
    sp = 0
    gp = -999
    do f = 1, nf   ! C layout sp[(2 pos + imag) nf + fld], zero based
      n   = nm(1, case_nm(f))
      m   = nm(2, case_nm(f))
      pos = int((2 * T + 3 - m) * m / 2 + (n - m), c_int64_t)
      sp((2 * pos + case_imag(f)) * nf + f) = 1
    end do
    rc = atlas_amd__Trans__invtrans_scalar(trans, int(nf, c_int), sp, gp)
    call expect(rc == 0, "invtrans_scalar returns 0")
    do f = 1, nf
      n   = nm(1, case_nm(f))
      m   = nm(2, case_nm(f))
      num = 0
      den = 0
      p   = int(f - 1, c_int64_t) * npts
      do j = 1, ny
        pl = legendre_closed_form(n, m, y(j) * pi / 180)
        do i = 0, nx(j) - 1
          lam = 2 * pi * i / nx(j)
          if (case_imag(f) == 0) then
            ref = pl * cos(m * lam) * merge(2._c_double, 1._c_double, m > 0)
          else
            ref = -2 * pl * sin(m * lam)
          end if
          p   = p + 1
          num = num + (gp(p) - ref)**2
          den = den + ref**2
        end do
      end do
      write (what, '(a,a,a,i0,a,i0,a,i0,a,es10.3)') "analytic ", gridname, " n=", n, " m=", m, " imag=", case_imag(f), &
        & " rel-rms ", sqrt(num / den)
      call expect(sqrt(num / den) < 1e-13_c_double, trim(what))   ! tolerance of test_transgeneral.cc:829-839
    end do
    call atlas_amd__Trans__delete(trans)
    call atlas_amd__Grid__delete(grid)
  end subroutine

  ! the call IFS-style callers make: scalars and vor/div pairs in one invtrans, gp = [u fields][v fields][scalar fields]
  ! (TransLocal.cc:1567-1596).  Solid-body rotation: vorticity 2 omega sin(lat) = coefficient (n=1, m=0) 2 omega / sqrt(3),
  ! u = omega a cos(lat), v = 0; the scalar beside it is the harmonic (2, 1).
  subroutine case_invtrans_vordiv_with_scalar()
    integer, parameter :: T = 31
    real(c_double), parameter :: a = 6371229._c_double, omega = 1e-5_c_double
    type(c_ptr) :: grid, trans
    integer(c_int), allocatable :: nx(:)
    real(c_double), allocatable :: y(:), sp(:), vor(:), div(:), gp(:), wind(:)
    integer(c_int64_t) :: npts, nspec, p, pos
    integer(c_int) :: rc
    integer :: i, j, ny
    real(c_double) :: err_u, err_v, err_s, err_w, lam

    grid  = atlas_amd__Grid__new_gaussian(c_str("O32"))
    trans = atlas_amd__Trans__new(grid, int(T, c_int))
    call expect(c_associated(trans), "Trans O32")
    if (.not. c_associated(trans)) return
    npts  = atlas_amd__Trans__nb_gridpoints(trans)
    nspec = atlas_amd__Trans__nb_spectral_coefficients(trans)
    ny    = atlas_amd__Grid__ny(grid)
    allocate (nx(ny), y(ny), sp(nspec), vor(nspec), div(nspec), gp(3 * npts), wind(2 * npts))
    rc  = atlas_amd__Grid__nx(grid, nx)
    rc  = atlas_amd__Grid__y(grid, y)
    sp  = 0
    vor = 0
    div = 0
    gp  = -999
    vor(2 * 1 + 1) = 2 * omega / sqrt(3._c_double)            ! pos(n=1, m=0) = 1, real part
    pos = int((2 * T + 3 - 1) * 1 / 2 + (2 - 1), c_int64_t)   ! pos(n=2, m=1)
    sp(2 * pos + 1) = 1
    rc = atlas_amd__Trans__invtrans(trans, 1_c_int, sp, 1_c_int, vor, div, gp)
    call expect(rc == 0, "invtrans returns 0")
    rc = atlas_amd__Trans__invtrans_vordiv2wind(trans, 1_c_int, vor, div, wind)
    call expect(rc == 0, "invtrans_vordiv2wind returns 0")
    err_u = 0
    err_v = 0
    err_s = 0
    err_w = 0
    p     = 0
    do j = 1, ny
      do i = 0, nx(j) - 1
        lam   = 2 * pi * i / nx(j)
        p     = p + 1
        err_u = max(err_u, abs(gp(p) - omega * a * cos(y(j) * pi / 180)))
        err_v = max(err_v, abs(gp(npts + p)))
        err_s = max(err_s, abs(gp(2 * npts + p) - 2 * legendre_closed_form(2, 1, y(j) * pi / 180) * cos(lam)))
        err_w = max(err_w, abs(wind(p) - gp(p)), abs(wind(npts + p) - gp(npts + p)))
      end do
    end do
    call expect(err_u < 1e-10_c_double * omega * a, "u = omega a cos(lat)")
    call expect(err_v < 1e-10_c_double * omega * a, "v = 0")
    call expect(err_s < 1e-12_c_double, "the scalar beside the wind pair")
    call expect(err_w == 0, "invtrans_vordiv2wind gives the same winds")
    call atlas_amd__Trans__delete(trans)
    call atlas_amd__Grid__delete(grid)
  end subroutine

  ! atlas_HaloExchange as atlas_f drives it (src/atlas_f/parallel/atlas_HaloExchange_module.fypp:88-120): setup(part, remote_idx)
  ! with base 1; execute on field(nvar, nnodes) with strides (stride of dim 2, stride of dim 1), extents (1, nvar), rank 2
  ! (HaloExchange.cc:195-209 builds the shape [parsize, 1, nvar] from that).  One process: ghost nodes are duplicates of owned
  ! nodes of the same process (periodic points).  The adjoint adds the halo values onto their owners and zeroes the halo.
  subroutine case_halo_exchange()
    integer, parameter :: nnodes = 12, nowned = 8, nvar = 3
    integer(c_int) :: part(nnodes), remote_idx(nnodes), rc
    real(c_double) :: field(nvar, nnodes), scalar(nnodes), expected(nvar, nnodes)
    type(c_ptr) :: hx
    integer :: i, v
    part = 0
    do i = 1, nnodes
      remote_idx(i) = i
    end do
    remote_idx(nowned + 1:nnodes) = [3, 1, 8, 5]
    hx = atlas_amd__HaloExchange__new()
    call expect(c_associated(hx), "HaloExchange handle")
    rc = atlas_amd__HaloExchange__setup(hx, part, remote_idx, 1_c_int, int(nnodes, c_int))
    call expect(rc == 0, "HaloExchange setup")
    do i = 1, nnodes
      do v = 1, nvar
        field(v, i) = merge(100._c_double * i + v, -1._c_double, i <= nowned)
      end do
      scalar(i) = merge(real(i, c_double), -1._c_double, i <= nowned)
    end do
    rc = atlas_amd__HaloExchange__execute_strided_double(hx, field, [int(nvar, c_int), 1_c_int], &
      & [1_c_int, int(nvar, c_int)], 2_c_int)
    call expect(rc == 0, "execute rank 2")
    rc = atlas_amd__HaloExchange__execute_strided_double(hx, scalar, [1_c_int], [1_c_int], 1_c_int)
    call expect(rc == 0, "execute rank 1")
    do i = 1, nnodes
      do v = 1, nvar
        call expect(field(v, i) == 100._c_double * remote_idx(i) + v, "halo values of field(nvar, nnodes)")
      end do
      call expect(scalar(i) == real(remote_idx(i), c_double), "halo values of a scalar field")
    end do
    ! adjoint: owners collect their duplicates, halo zeroed (HaloExchange.h:227-306)
    field = 1
    expected = 1
    expected(:, nowned + 1:nnodes) = 0
    do i = nowned + 1, nnodes
      expected(:, remote_idx(i)) = expected(:, remote_idx(i)) + 1
    end do
    rc = atlas_amd__HaloExchange__execute_adjoint_strided_double(hx, field, [int(nvar, c_int), 1_c_int], &
      & [1_c_int, int(nvar, c_int)], 2_c_int)
    call expect(rc == 0, "execute_adjoint rank 2")
    call expect(all(field == expected), "adjoint: owners collect, halo zeroed")
    call atlas_amd__HaloExchange__delete(hx)
  end subroutine

end program test_trans_f
