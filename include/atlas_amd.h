/*
 * atlas_amd -- C ABI of the MI355X-native TransLocal inverse spherical-harmonics transform and HaloExchange.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ / torch types.  Every entry point names the
 * reference (ecmwf/atlas 0.44.1) interface it replaces.  INTEGRATION.md shows the adapter a maintainer adds on
 * the Atlas side (a TransImpl subclass registered with TransBuilderGrid, and a HaloExchange shim).
 *
 * Conventions
 *   - functions returning int return 0 on success, non-zero on error; atlas_amd__last_error() gives the message
 *     (the reference throws eckit::Exception through its extern "C" layer, TransInterface.cc:43-283; a C ABI
 *     cannot, so the adapter re-throws).  What the reference asserts (ATLAS_ASSERT: null handles, negative field counts, an
 *     array missing for a call that announces fields of its kind) is such an error here, never a crash; getters on a null
 *     handle return -1 / NULL.  Zero fields of a kind need no array of that kind (TransLocal.cc:1486-1490 passes nullptr).
 *   - "_device" variants take device pointers and are asynchronous on the object's HIP stream;
 *     the others take host pointers and are synchronous.
 *   - layouts are exactly those of the reference (SURVEY.md section 8 "Layout cheat-sheet"):
 *       spectra   sp[(2*pos(m,n) + imag)*nf + fld],  pos(m,n) = (2T+3-m)*m/2 + (n-m)     (TransLocal.cc:970-987)
 *       gridpoint gp[fld*npts + rowoffset(jlat) + jlon]                                   (TransLocal.cc:1132,1187)
 */
#ifndef ATLAS_AMD_H
#define ATLAS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct atlas_amd_Grid atlas_amd_Grid;
typedef struct atlas_amd_Trans atlas_amd_Trans;
typedef struct atlas_amd_HaloExchange atlas_amd_HaloExchange;
typedef struct atlas_amd_StructuredColumns atlas_amd_StructuredColumns;

/* ---------------------------------------------------------------------------------------------------------------
 * errors / library info
 * ------------------------------------------------------------------------------------------------------------- */
const char* atlas_amd__last_error(void);
/* thread-local diagnostics that are not errors, set (or cleared) by atlas_amd__Trans__new_config -- the only call that has any so
 * far -- and left untouched by every other call: e.g. the TransLocal option
 * keys atlas_amd__Trans__new_config accepted and ignored (fft, matrix_multiply, precompute, warning, write_fft, ...;
 * src/atlas/option/TransOptions.cc:38-74, TransLocal.cc:61-110) */
const char* atlas_amd__last_note(void);
const char* atlas_amd__version(void);
/* Environment hygiene (a library loaded into Atlas inherits the caller's environment).  Every ATLAS_AMD_* switch the library reads
 * is listed in ONE table (csrc/env.cpp; INTEGRATION.md section 8 is generated from it).  atlas_amd__set_ignore_env(1) -- what the
 * adapter plugin calls when it is loaded -- makes every switch read as unset for the rest of the process (0: honour them again);
 * ATLAS_AMD_IGNORE_ENV=1 in the environment does the same.  Development switches exist only in dev / experiments builds.
 * atlas_amd__effective_config writes one line per switch, "NAME<TAB>class<TAB>value in effect<TAB>source<TAB>default<TAB>what",
 * source = default | env | ignored | compiled out, into buf (NUL-terminated, truncated to `capacity`) and returns the number of
 * bytes the whole text needs (call with capacity 0 to size the buffer).  No reference counterpart. */
int atlas_amd__set_ignore_env(int on);
long long atlas_amd__effective_config(char* buf, long long capacity);
/* number of visible HIP devices (0: the transform cannot run; there is no CPU fallback) */
int atlas_amd__device_count(void);
/* stream ordering helper for callers that own their device arrays on another HIP stream: all work submitted to
 * `waiting_stream` after this call starts after the work submitted to `signalling_stream` before it (event record +
 * hipStreamWaitEvent, no host synchronisation).  Trans / HaloExchange objects run on their own non-blocking stream
 * (atlas_amd__Trans__stream / atlas_amd__HaloExchange__stream). */
int atlas_amd__stream_wait_stream(void* waiting_stream, void* signalling_stream);
/* device memory for callers without the HIP headers (C / Fortran drivers of the device-pointer entry points): plain
 * hipMalloc / hipFree / hipMemcpy (synchronous) / hipDeviceSynchronize / hipSetDevice */
void* atlas_amd__device_malloc(size_t bytes);
int atlas_amd__device_free(void* ptr);
int atlas_amd__device_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
int atlas_amd__device_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);
int atlas_amd__device_synchronize(void);
int atlas_amd__set_device(int device);
/* measurement aid (bench.py): TFLOP/s this device sustains on v_mfma_f64_16x16x4_f64 alone, best of `repeats` kernels of
 * about target_ms each on the current device's default stream (4 wavefronts per SIMD, random operands) */
int atlas_amd__diag_mfma_f64_rate(double target_ms, int repeats, double* tflops_out);
/* the same for v_mfma_f32_16x16x4_f32, the instruction of the fp32 variant's Legendre stage [r4] */
int atlas_amd__diag_mfma_f32_rate(double target_ms, int repeats, double* tflops_out);
/* measurement aid: average milliseconds of the field-major [nb_fields][npts] -> point-major [npts][nb_fields] transposition that
 * atlas_amd__Trans__invtrans_distributed_many_halo runs in front of the halo exchange (device pointers; default stream) */
int atlas_amd__diag_gp_to_field(const double* gp_dev, double* field_dev, long long npts, int nb_fields, int repeats,
                                double* ms_out);

/* ---------------------------------------------------------------------------------------------------------------
 * Grid description.  Replaces the `const Grid::Implementation*` argument of atlas__Trans__new
 * (src/atlas/trans/detail/TransInterface.h:52-54): the adapter passes ny, nx(j), y(j) of its StructuredGrid.
 * ------------------------------------------------------------------------------------------------------------- */
/* "F<N>" regular Gaussian / "O<N>" octahedral Gaussian (src/atlas/grid/detail/grid/Gaussian.cc:111-177) */
atlas_amd_Grid* atlas_amd__Grid__new_gaussian(const char* name);
/* any global structured grid: ny latitudes (degrees, north to south), nx[j] points per latitude, xmin = 0 */
atlas_amd_Grid* atlas_amd__Grid__new_structured(int ny, const int nx[], const double lat_deg[]);
void atlas_amd__Grid__delete(atlas_amd_Grid* g);
int atlas_amd__Grid__ny(const atlas_amd_Grid* g);
int atlas_amd__Grid__nxmax(const atlas_amd_Grid* g);
int64_t atlas_amd__Grid__size(const atlas_amd_Grid* g);
int atlas_amd__Grid__regular(const atlas_amd_Grid* g);
int atlas_amd__Grid__nx(const atlas_amd_Grid* g, int nx_out[]);
int atlas_amd__Grid__y(const atlas_amd_Grid* g, double lat_deg_out[]);
/* Gaussian latitudes north pole -> south pole, 2N values
 * (src/atlas/grid/detail/spacing/gaussian/Latitudes.cc:41-67) */
int atlas_amd__gaussian_latitudes_npole_spole(int N, double lats_out[]);

/* ---------------------------------------------------------------------------------------------------------------
 * Trans.  Replaces atlas__Trans__* (src/atlas/trans/detail/TransInterface.h:44-107) for type "local".
 * ------------------------------------------------------------------------------------------------------------- */
/* atlas__Trans__new(grid, truncation)                                        TransInterface.h:52 */
atlas_amd_Trans* atlas_amd__Trans__new(const atlas_amd_Grid* grid, int truncation);
/* atlas__Trans__new_config(grid, truncation, config).  `config` is "key=value;key=value" with keys
 *   profile=0|1          record HIP events around the two stages (atlas_amd__Trans__timings)
 *   nparts=P;part=p      multi-GPU decomposition: this object owns wavenumbers m%P==p and latitude band p
 *   shard=m|band|mirror  m (default): Legendre stage on the owned wavenumbers, Fourier stage on the owned band; the
 *                        caller transposes in between (stage API below).  band: both stages on the owned latitude
 *                        band, no exchange: the *_device invtrans entry points then return the band's grid points.
 *                        mirror: both stages on a northern band of rows AND its mirror image in the south
 *                        (atlas_amd__Trans__mirror_rows): no exchange and the hemisphere symmetry of the Legendre stage
 *                        is kept; the output holds the northern rows, then the southern rows (both north to south)
 *   rows=j0:j1           zonal-band crop: transform only latitude rows j0..j1-1 of the (global) grid -- the nested
 *                        regional case of TransLocal (TransLocal.cc:394-470) for domains that keep whole rows;
 *                        nb_gridpoints and the output arrays then cover these rows only
 *   type=local|mi355x    accepted for atlas option::type compatibility
 *   fft=OFF|FFTW|pocketfft, matrix_multiply=..., precompute=..., warning=..., write_fft=..., read_fft=..., write_legendre=...,
 *   read_legendre=..., export_legendre=..., global=..., split_y=..., nproma=..., flt=..., scalar_derivatives=..., ...
 *                        TransLocal's own option keys (option/TransOptions.cc:38-74; read at TransLocal.cc:61-110,326-335):
 *                        accepted and ignored -- FFT and GEMM are this library's kernels, the tables are always precomputed,
 *                        cache files are the caller's (legendre_cache argument; adapter/TransMI355X.cc handles write_legendre)
 *                        -- each with an entry in atlas_amd__last_note(); an fft value the reference rejects is rejected
 *   tables=host|device   where the Legendre table is computed when no cache is given: on the host (OpenMP, then
 *                        uploaded) or on the device from O(T^2) host-prepared inputs (bit-identical; default: the
 *                        environment variable ATLAS_AMD_TABLES, else device: 0.9 s against 6.7 s at TL1279)
 * legendre_cache / size: optional Legendre cache blob in TransLocal's file layout (TransLocal.cc:608-614), or NULL */
atlas_amd_Trans* atlas_amd__Trans__new_config(const atlas_amd_Grid* grid, int truncation, const char* config,
                                              const void* legendre_cache, size_t legendre_cache_size);
void atlas_amd__Trans__delete(atlas_amd_Trans* t);                           /* atlas__Trans__delete      :55 */
int atlas_amd__Trans__truncation(const atlas_amd_Trans* t);                  /* atlas__Trans__truncation  :99 */
int64_t atlas_amd__Trans__nb_gridpoints(const atlas_amd_Trans* t);           /* local band (== global for nparts=1) */
/* Targets that are NOT a crop of a global grid: a regular longitude-latitude grid with arbitrary latitudes (degrees, in the
 * order of its rows) and equally spaced longitudes west + i * dlon, i < nlon -- TransLocal's "no_nest" branch
 * (src/atlas/trans/local/TransLocal.cc:394-406: no hemisphere symmetry of the target, no Fourier truncation towards the poles,
 * :535-557 Legendre polynomials at the grid's own latitudes, :719-738,1139-1148 Fourier stage as a matrix product).  Scalar
 * fields; grid points gp[lon + nlon * (lat + nlat * field)] as the reference; spectra as atlas__Trans__invtrans_scalar.  The
 * device-pointer variant is asynchronous on atlas_amd__RegionalTrans__stream. */
typedef struct atlas_amd_RegionalTrans atlas_amd_RegionalTrans;
atlas_amd_RegionalTrans* atlas_amd__RegionalTrans__new(int nlon, double west, double dlon, int nlat, const double lats[],
                                                       int truncation);
/* unstructured target: npts points (lon, lat) in degrees -- TransLocal's unstructured path (TransLocal.cc:741-790, 1200-1420):
 * Legendre polynomials at every point's latitude, the Fourier sum evaluated point by point; grid points gp[point + npts *
 * field]; u and v of the vor/div path divided by the cosine of the point's latitude.  Points at the poles are evaluated at
 * +-89.9999999 degrees like the structured paths. */
atlas_amd_RegionalTrans* atlas_amd__RegionalTrans__new_unstructured(int npts, const double lons[], const double lats[],
                                                                    int truncation);
void atlas_amd__RegionalTrans__delete(atlas_amd_RegionalTrans* t);
int64_t atlas_amd__RegionalTrans__nb_gridpoints(const atlas_amd_RegionalTrans* t);
int atlas_amd__RegionalTrans__invtrans_scalar(atlas_amd_RegionalTrans* t, int nb_fields, const double scalar_spectra[],
                                              double gp_fields[]);
int atlas_amd__RegionalTrans__invtrans_scalar_device(atlas_amd_RegionalTrans* t, int nb_fields, const double* sp_dev,
                                                     double* gp_dev);
/* as atlas__Trans__invtrans: gp = [u fields][v fields][scalar fields] (TransLocal.cc:1523-1597) */
int atlas_amd__RegionalTrans__invtrans_vordiv(atlas_amd_RegionalTrans* t, int nb_scalar_fields, const double scalar_spectra[],
                                              int nb_vordiv_fields, const double vorticity_spectra[],
                                              const double divergence_spectra[], double gp_fields[]);
int atlas_amd__RegionalTrans__synchronize(atlas_amd_RegionalTrans* t);
void* atlas_amd__RegionalTrans__stream(const atlas_amd_RegionalTrans* t);

/* RectangularDomain crop of a global structured grid (atlas::Grid(grid, domain), src/atlas/grid/detail/grid/Structured.cc:
 * 390-560): the rows [row_begin, row_end) whose latitude lies in [south, north] and per row the run of count[r] points
 * starting at global index first_index[r] (wrapping around) whose longitude, normalised to [west, west + 360), lies in
 * [west, east]; bounds inclusive with the reference's tolerance of 1e-6 degrees.  first_index / count may be NULL (rows
 * only), else they hold at least `capacity` >= row_end - row_begin entries.  A Trans for the crop: config key
 * "domain=west,east,south,north" of atlas_amd__Trans__new_config (trans::Trans(grid, domain, truncation),
 * TransLocal.cc:394-470); its grid points are the windows row by row. */
int atlas_amd__Grid__crop_to_domain(const atlas_amd_Grid* grid, double west, double east, double south, double north,
                                    int* row_begin, int* row_end, int first_index[], int count[], int capacity);
int64_t atlas_amd__Trans__nb_gridpoints_global(const atlas_amd_Trans* t);
int64_t atlas_amd__Trans__nb_spectral_coefficients(const atlas_amd_Trans* t); /* (T+1)(T+2), TransLocal.h:90 */

/* atlas__Trans__invtrans_scalar(t, nb_fields, sp, gp)                         TransInterface.h:77 */
int atlas_amd__Trans__invtrans_scalar(atlas_amd_Trans* t, int nb_fields, const double scalar_spectra[],
                                      double scalar_fields[]);
/* atlas__Trans__invtrans(t, nb_scalar, sp, nb_vordiv, vor, div, gp, config)   TransInterface.h:74
 * gp holds [u fields][v fields][scalar fields] (TransLocal.cc:1567-1596) */
int atlas_amd__Trans__invtrans(atlas_amd_Trans* t, int nb_scalar_fields, const double scalar_spectra[],
                               int nb_vordiv_fields, const double vorticity_spectra[],
                               const double divergence_spectra[], double gp_fields[]);
/* atlas__Trans__invtrans_vordiv2wind(t, nb_fields, vor, div, wind)            TransInterface.h:79 */
int atlas_amd__Trans__invtrans_vordiv2wind(atlas_amd_Trans* t, int nb_fields, const double vorticity_spectra[],
                                           const double divergence_spectra[], double wind_fields[]);
/* device-pointer variants (asynchronous on the Trans stream) */
int atlas_amd__Trans__invtrans_scalar_device(atlas_amd_Trans* t, int nb_fields, const double* scalar_spectra_dev,
                                             double* scalar_fields_dev);
int atlas_amd__Trans__invtrans_device(atlas_amd_Trans* t, int nb_scalar_fields, const double* scalar_spectra_dev,
                                      int nb_vordiv_fields, const double* vorticity_spectra_dev,
                                      const double* divergence_spectra_dev, double* gp_fields_dev);
/* fp32 variant of invtrans_scalar on device arrays (an extension: TransLocal is double only; BASELINE config C5).
 * float spectra and grid points in the layouts above; Legendre stage on v_mfma_f32_16x16x4_f32 with the table converted
 * to float, Fourier stage with fp32 loads/stores around fp64 arithmetic.  Expect ~1e-6 relative accuracy. */
int atlas_amd__Trans__invtrans_scalar_device_f32(atlas_amd_Trans* t, int nb_fields, const float* scalar_spectra,
                                                 float* gp_fields);
int atlas_amd__Trans__invtrans_scalar_f32(atlas_amd_Trans* t, int nb_fields, const float scalar_spectra[],
                                          float gp_fields[]); /* host pointers */
/* [r5] the vor/div call of the fp32 variant on device arrays: atlas__Trans__invtrans (TransInterface.h:74-79; TransLocal.cc:1523-1597)
 * with float spectra and grid points; U, V spectra formed in double and stored as float; gp = [u fields][v fields][scalar fields] */
int atlas_amd__Trans__invtrans_device_f32(atlas_amd_Trans* t, int nb_scalar_fields, const float* scalar_spectra_dev,
                                          int nb_vordiv_fields, const float* vorticity_spectra_dev,
                                          const float* divergence_spectra_dev, float* gp_fields_dev);

/* direct transforms and adjoints: not implemented by TransLocal either (TransLocal.cc:848-857,899-927,1599-1685);
 * these return an error whose message starts with "Not implemented" */
int atlas_amd__Trans__dirtrans_scalar(atlas_amd_Trans* t, int nb_fields, const double scalar_fields[],
                                      double scalar_spectra[]);
int atlas_amd__Trans__dirtrans_wind2vordiv(atlas_amd_Trans* t, int nb_fields, const double wind_fields[],
                                           double vorticity_spectra[], double divergence_spectra[]);
int atlas_amd__Trans__invtrans_adj_scalar(atlas_amd_Trans* t, int nb_fields, const double gp_fields[],
                                          double scalar_spectra[]);
/* atlas__Trans__invtrans_adj / __invtrans_vordiv2wind_adj                    TransInterface.h:61-68 (not implemented) */
int atlas_amd__Trans__invtrans_adj(atlas_amd_Trans* t, int nb_scalar_fields, const double gp_fields[],
                                   int nb_vordiv_fields, double vorticity_spectra[], double divergence_spectra[],
                                   double scalar_spectra[]);
int atlas_amd__Trans__invtrans_vordiv2wind_adj(atlas_amd_Trans* t, int nb_fields, const double wind_fields[],
                                               double vorticity_spectra[], double divergence_spectra[]);

/* Backend registry of atlas::trans::Trans (TransInterface.h:46-48, Trans.cc:37-48): this library provides the "local"
 * implementation, also reachable under the name "mi355x".  __backend returns a malloc'ed copy (free with free()). */
int atlas_amd__Trans__has_backend(const char* backend);
int atlas_amd__Trans__set_backend(const char* backend);
int atlas_amd__Trans__backend(char** backend, size_t* size);
/* atlas__Trans__grid (TransInterface.h:103): the grid the object was built with (borrowed) */
const atlas_amd_Grid* atlas_amd__Trans__grid(const atlas_amd_Trans* t);
/* atlas__Trans__handle (TransInterface.h:98): TransImpl::handle() is ATLAS_NOTIMPLEMENTED for every backend but the IFS
 * one (TransImpl.cc:20-22), TransLocal included: returns an error whose message starts with "Not implemented" */
int atlas_amd__Trans__handle(const atlas_amd_Trans* t, int* handle);
/* atlas__Trans__spectral (TransInterface.h:104): the spectral function space of the transform, functionspace::Spectral
 * (truncation) (TransLocal.cc:809-814); borrowed, lives as long as the Trans */
typedef struct atlas_amd_Spectral atlas_amd_Spectral;
const atlas_amd_Spectral* atlas_amd__Trans__spectral(const atlas_amd_Trans* t);
int atlas_amd__Spectral__truncation(const atlas_amd_Spectral* s);
int64_t atlas_amd__Spectral__nb_spectral_coefficients(const atlas_amd_Spectral* s);          /* (T+1)(T+2), Spectral.h:184 */
int64_t atlas_amd__Spectral__nb_spectral_coefficients_global(const atlas_amd_Spectral* s);

/* Field / FieldSet overloads (TransInterface.h:73-97, TransLocal.cc:818-897).  A field is described by its host data
 * pointer and C-order shape (what array::make_view sees); only rank-1 fields are supported, as in TransLocal, plus the
 * rank-2 wind field of invtrans_vordiv2wind_field in either (2, npts) or (npts, 2) shape.  NB the (npts, 2) branch
 * reproduces TransLocal's gp_transpose call literally (TransLocal.cc:861-867,888-893): out[f*npts + g] = tmp[g*2 + f]
 * with tmp the (2, npts) result. */
typedef struct atlas_amd_Field {
    double* data;
    int rank;
    long shape[2];
} atlas_amd_Field;
int atlas_amd__Trans__invtrans_field(atlas_amd_Trans* t, const atlas_amd_Field* spfield, atlas_amd_Field* gpfield);
int atlas_amd__Trans__invtrans_fieldset(atlas_amd_Trans* t, const atlas_amd_Field* spfields, int nb_spfields,
                                        atlas_amd_Field* gpfields, int nb_gpfields);
int atlas_amd__Trans__invtrans_vordiv2wind_field(atlas_amd_Trans* t, const atlas_amd_Field* spvor,
                                                 const atlas_amd_Field* spdiv, atlas_amd_Field* gpwind);
/* not implemented by TransLocal: error "Not implemented" */
int atlas_amd__Trans__invtrans_grad_field(atlas_amd_Trans* t, const atlas_amd_Field* spfield, atlas_amd_Field* gradfield);
int atlas_amd__Trans__invtrans_adj_field(atlas_amd_Trans* t, const atlas_amd_Field* gpfield, atlas_amd_Field* spfield);
int atlas_amd__Trans__invtrans_adj_fieldset(atlas_amd_Trans* t, const atlas_amd_Field* gpfields, int nb_gpfields,
                                            atlas_amd_Field* spfields, int nb_spfields);
int atlas_amd__Trans__invtrans_grad_adj_field(atlas_amd_Trans* t, const atlas_amd_Field* gpfield,
                                              atlas_amd_Field* spfield);
int atlas_amd__Trans__invtrans_vordiv2wind_adj_field(atlas_amd_Trans* t, const atlas_amd_Field* gpwind,
                                                     atlas_amd_Field* spvor, atlas_amd_Field* spdiv);
int atlas_amd__Trans__dirtrans_field(atlas_amd_Trans* t, const atlas_amd_Field* gpfield, atlas_amd_Field* spfield);
int atlas_amd__Trans__dirtrans_fieldset(atlas_amd_Trans* t, const atlas_amd_Field* gpfields, int nb_gpfields,
                                        atlas_amd_Field* spfields, int nb_spfields);
int atlas_amd__Trans__dirtrans_wind2vordiv_field(atlas_amd_Trans* t, const atlas_amd_Field* gpwind,
                                                 atlas_amd_Field* spvor, atlas_amd_Field* spdiv);

/* VorDivToUV (the sibling factory TransLocal uses, src/atlas/trans/VorDivToUV.h:36-133; "local" implementation
 * VorDivToUVLocal.cc:62-189): spectral vorticity / divergence -> spectral U = u cos(lat), V = v cos(lat), all four arrays in
 * the spectral layout of invtrans with `nb_fields` fields and truncation `truncation`; nb_coeff must be
 * (truncation+1)*(truncation+2).  Host pointers; the _device variant takes device pointers and is asynchronous on
 * `hip_stream` (NULL: the default stream). */
int atlas_amd__VorDivToUV__execute(int truncation, int nb_coeff, int nb_fields, const double vorticity[],
                                   const double divergence[], double U[], double V[]);
int atlas_amd__VorDivToUV__execute_device(int truncation, int nb_coeff, int nb_fields, const double* vorticity,
                                          const double* divergence, double* U, double* V, void* hip_stream);

/* stream control */
void* atlas_amd__Trans__stream(atlas_amd_Trans* t);              /* hipStream_t */
int atlas_amd__Trans__set_stream(atlas_amd_Trans* t, void* hip_stream);
int atlas_amd__Trans__synchronize(atlas_amd_Trans* t);

/* Legendre cache, byte-compatible with TransLocal's write_legendre / LegendreCache blobs
 * (TransLocal.cc:638-647, src/atlas/trans/Cache.h:98-136) */
size_t atlas_amd__Trans__legendre_cache_size(const atlas_amd_Trans* t);
int atlas_amd__Trans__legendre_cache_export(const atlas_amd_Trans* t, void* buffer, size_t size);

/* shard=mirror: out = {b0, b1}: the object transforms rows [b0, b1) and [ny-b1, ny-b0) of the grid */
int atlas_amd__Trans__mirror_rows(const atlas_amd_Trans* t, int out[2]);
/* the row boundaries shard=mirror uses for nparts parts (nparts+1 values over rows 0 .. ny/2; host only) */
int atlas_amd__mirror_bands(const atlas_amd_Grid* grid, int nparts, int bands_out[]);
/* the latitude bands of the distributed transform for nparts ranks (nparts+1 row boundaries; Atlas BandsDistribution rule:
 * a row belongs to the part of its first point) -- what atlas_amd__Trans__bands returns, without a device (host only) */
int atlas_amd__latitude_bands(const atlas_amd_Grid* grid, int truncation, int nparts, int bands_out[]);
/* host-only test hook: nlat0[T+1] and the Fourier truncation of every row, for the grid (caps_rows = 0) or for its two
 * polar caps of caps_rows latitudes each taken as a grid inside the full one (2*caps_rows rows) */
int atlas_amd__trans_geometry_probe(const atlas_amd_Grid* grid, int truncation, int caps_rows, int nlat0_out[],
                                    int row_mmax_out[]);
/* the tile-blocked Legendre table as it sits in device memory (legendre_table_bytes / 8 doubles): test hook for the
 * device generation of the table (config key tables=device|host) */
int atlas_amd__Trans__legendre_table_download(const atlas_amd_Trans* t, double* out, size_t size_doubles);

/* the two stages separately: multi-GPU drivers and stage-level parity tests.
 * Fourier intermediate layout: F[(lat*m_cnt + m/nparts)*RP + 2*fld + imag], RP = fourier_row_pitch */
int atlas_amd__Trans__fourier_row_pitch(const atlas_amd_Trans* t, int nb_fields);
int64_t atlas_amd__Trans__fourier_size(const atlas_amd_Trans* t, int nb_fields); /* doubles */
int atlas_amd__Trans__owned_wavenumbers(const atlas_amd_Trans* t);
int atlas_amd__Trans__bands(const atlas_amd_Trans* t, int bands_out[] /* nparts+1 */);
int atlas_amd__Trans__legendre_device(atlas_amd_Trans* t, int truncation_in, int nb_fields,
                                      const double* spectra_dev, double* fourier_dev);
int atlas_amd__Trans__fourier_device(atlas_amd_Trans* t, int nb_fields, int nb_vordiv_fields,
                                     const double* const part_base_dev[], const int part_cnt[], double* gp_dev);

/* introspection used by tests */
int atlas_amd__Trans__nlat0(const atlas_amd_Trans* t, int nlat0_out[] /* T+1 */);
/* the Fourier kernel every latitude row of the grid takes: out[3 j] = method (csrc/fft_plan.h: FftMethod), out[3 j + 1] = transform
 * length M of the row's plan (Bluestein: the convolution length; direct rows: n/2; native rows: 10 x first radix + stages), out[3 j + 2] = kernel
 * (0 run-time shaped, 1 specialised Bluestein, 2 specialised direct, 3 dense-stage experiment, 4 native mixed radix).  Parity tests
 * take one northern and one southern row of every (method, M, kernel) that is launched (TransLocal.cc:1155-1196 treats all rows alike) */
int atlas_amd__Trans__fft_row_classes(const atlas_amd_Trans* t, int out[] /* 3 * nlats */);
/* kernel launches of one Fourier stage: out = {launches (one per row class), of which: the fused launch of the coarse Bluestein
 * classes (0 / 1), launches of native rows with two fields per workgroup}.  The A/B switches ATLAS_AMD_FFT_COARSE_FUSED /
 * ATLAS_AMD_FFT_NATIVE_FPJ are read when the object is built; the bitwise tests assert through this that the other path ran */
int atlas_amd__Trans__fourier_launch_plan(const atlas_amd_Trans* t, int out[3]);
double atlas_amd__Trans__legendre_flops(const atlas_amd_Trans* t, int nb_fields);
int64_t atlas_amd__Trans__legendre_table_bytes(const atlas_amd_Trans* t);
/* accumulated kernel times from HIP events on the Trans stream (profile=1):
 * out = {legendre_ms, legendre_calls, fourier_ms, fourier_calls}; reset != 0 clears the accumulators */
int atlas_amd__Trans__timings(atlas_amd_Trans* t, double out[4], int reset);
/* the same for the stage only the vor/div calls have -- extend_truncation + vd2uv + field interleave in one kernel
 * (TransLocal.cc:1496-1581, VorDivToUVLocal.cc:62-184): out = {prepare_ms, prepare_calls} */
int atlas_amd__Trans__timings_vordiv(atlas_amd_Trans* t, double out[2], int reset);
int atlas_amd__Trans__set_profile(atlas_amd_Trans* t, int on);
/* dev profiling of the FFT kernel: if out != NULL read the 64 per-phase shader-clock accumulators (slots 0..31
 * Bluestein rows, 32..63 direct rows), then enable (and zero) or disable the accumulation */
int atlas_amd__Trans__fft_phase_profile(atlas_amd_Trans* t, int enable, unsigned long long out[64]);
/* dev builds of the library (-DAA_FFT_TRACE) only: per-wavefront trace of the specialised Fourier kernel (hardware id and
 * shader clock at the phase boundaries, 8 words per wavefront).  out == NULL: allocate and zero `words` words (0 frees);
 * else copy the first `words` words out.  A normal build accepts the call and records nothing. */
int atlas_amd__Trans__fft_trace(atlas_amd_Trans* t, unsigned long long words, unsigned long long* out);

/* host-only helpers exposed for CPU tests of the host logic (no GPU needed) */
int atlas_amd__fourier_truncation(int truncation, int nx, int nxmax, int ndgl, double lat_rad, int fullgrid);
/* Legendre tables in the reference layout for (grid, truncation): sizes via *_size, then fill */
int atlas_amd__legendre_reference_tables(const atlas_amd_Grid* grid, int truncation, double* leg_sym,
                                         size_t size_sym, double* leg_asym, size_t size_asym);
int atlas_amd__legendre_reference_sizes(const atlas_amd_Grid* grid, int truncation, size_t* size_sym,
                                        size_t* size_asym);
/* host run of the device generator of the Legendre table (legendre_gen_core.h, the code of legendre_gen_kernel.hip)
 * against the host generator, for the decomposition (nparts, part, by_band): number of table entries that differ in
 * any bit.  Test hook only. */
int atlas_amd__legendre_gen_host_selfcheck(const atlas_amd_Grid* grid, int truncation, int nparts, int part,
                                           int by_band, long long* table_doubles, long long* mismatches);
/* run ONE row of the c2r transform on the host with the kernel's own phase code (fft_core.h); modes: n/2+1
 * interleaved complex values; out: n reals.  Test hook only -- the product never computes on the CPU. */
int atlas_amd__fft_host_row(int n, const double* modes, int mmax, double* out);
/* same through the generic (run-time shape) phase code even where a compile-time specialised instance exists */
int atlas_amd__fft_host_row_generic(int n, const double* modes, int mmax, double* out);
/* the same with the native mixed-radix rows (csrc/fft_native.h; opt-in, ATLAS_AMD_FFT_NATIVE=1) forced on -- an error if the
 * length has no native plan -- / off [r4] */
int atlas_amd__fft_host_row_native(int n, const double* modes, int mmax, double* out);
int atlas_amd__fft_host_row_bluestein(int n, const double* modes, int mmax, double* out);
/* the plan of row length n (native != 0: with the native rows enabled): out = {method (csrc/fft_plan.h: FftMethod), transform
 * length M, LDS elements, number of stages, radix[0..7] (DIF order), specialised instance (0 / 1), LDS pitch of a native row's
 * top-level blocks, entries of the native tables, 0} */
int atlas_amd__fft_plan_info(int n, int native, int out[16]);
/* same with the dense-stage ("hybrid") plan where the row length admits one: h = n/2 = A*B, A the product of the prime
 * factors > 5 of h (A <= 257), B {2,3,5}-smooth; other lengths take their usual plan */
int atlas_amd__fft_host_row_hybrid(int n, const double* modes, int mmax, double* out);
/* same with the coarse row classes Trans plans for small reduced grids (Bluestein rows of length 256 / 512 / 1024 / 2048 for
 * every even n whose 2 (n/2) - 1 fits; csrc/fft_plan.h: PlanOptions::coarse_classes) */
int atlas_amd__fft_host_row_coarse(int n, const double* modes, int mmax, double* out);

/* atlas::trans::LegendreCacheCreator, type "local" (src/atlas/trans/LegendreCacheCreator.h:30-111,
 * local/LegendreCacheCreatorLocal.cc:66-165): uid = "local-T<T>-GaussianN<N>|L-ny<ny>|S-ny<ny>|grid-<md5>-OPT<md5>" (expected
 * strings: src/tests/trans/test_trans.cc:600-696), estimate = T^3/2*8 bytes.  create() is
 * atlas_amd__Trans__legendre_cache_export on a Trans of the same (grid, truncation). */
int atlas_amd__LegendreCacheCreator__uid(const atlas_amd_Grid* grid, int truncation, int flt, char* out, size_t capacity);
int64_t atlas_amd__LegendreCacheCreator__estimate(int truncation);
int atlas_amd__LegendreCacheCreator__supported(const atlas_amd_Grid* grid);

/* grid::Partitioner("equal_regions", N) for structured grids, Atlas's default (EqualRegionsPartitioner.cc:70-347,443-605):
 * eq_caps = Leopardi's zones north -> south (regions per zone, colatitude of each zone's southern edge; expected values of
 * src/tests/mesh/test_rgg.cc:103-165); partition_out[npts] = part of every grid point in global order, the explicit
 * grid::Distribution atlas_amd__StructuredColumns__new_distribution accepts
 * (src/tests/functionspace/test_structuredcolumns.cc:87-106: O8 on 5 parts). */
int atlas_amd__eq_caps(int nb_regions, int capacity, int regions_per_zone[], double zone_colatitudes[], int* nb_zones);
int atlas_amd__equal_regions_partition(const atlas_amd_Grid* grid, int nb_parts, int partition_out[]);

/* ---------------------------------------------------------------------------------------------------------------
 * Communicators: the inter-GPU transport of the library (grouped point-to-point exchanges of device buffers).
 * Replaces the eckit::mpi communicator behind parallel::HaloExchange (iReceive / iSend per peer,
 * src/atlas/parallel/HaloExchange.h:191-219,333-369; allToAll / allToAllv of the setup, HaloExchange.cc:118,156).
 *   rccl  : one process per GPU, ncclSend / ncclRecv groups over xGMI.  One rank calls get_unique_id and the caller's
 *           control plane (MPI_Bcast in Atlas) hands the bytes to the others; every rank then calls new_rccl with the
 *           HIP device it will use already current.
 *   local : N ranks inside one process (one host thread per rank, same device): rendezvous + device copies.  For tests
 *           and single-process multi-rank drivers. */
typedef struct atlas_amd_Comm atlas_amd_Comm;
typedef struct atlas_amd_CommHub atlas_amd_CommHub;
int atlas_amd__Comm__unique_id_bytes(void);                 /* 128 */
int atlas_amd__Comm__get_unique_id(void* out);
atlas_amd_Comm* atlas_amd__Comm__new_rccl(const void* unique_id, int nranks, int rank);
atlas_amd_CommHub* atlas_amd__CommHub__new(int nranks);
void atlas_amd__CommHub__delete(atlas_amd_CommHub* hub);
atlas_amd_Comm* atlas_amd__Comm__new_local(atlas_amd_CommHub* hub, int rank);
void atlas_amd__Comm__delete(atlas_amd_Comm* comm);
int atlas_amd__Comm__size(const atlas_amd_Comm* comm);
int atlas_amd__Comm__rank(const atlas_amd_Comm* comm);
const char* atlas_amd__Comm__kind(const atlas_amd_Comm* comm);   /* "rccl" | "local" */
int atlas_amd__Comm__barrier(atlas_amd_Comm* comm);
/* one grouped exchange of device buffers, asynchronous on `stream`; between two ranks the k-th send of one side is
 * matched with the k-th receive of the other (same size) */
int atlas_amd__Comm__exchange(atlas_amd_Comm* comm, int nsend, const int send_peer[], void* const send_ptr[],
                              const size_t send_bytes[], int nrecv, const int recv_peer[], void* const recv_ptr[],
                              const size_t recv_bytes[], void* stream);

/* Distributed inverse transform (one rank per GPU): `t` made with config "nparts=<P> part=<p> shard=m" for the
 * communicator's (size, rank).  Legendre stage on the rank's wavenumbers (m % P == p), m -> latitude transposition of the
 * Fourier intermediate over the communicator (messages of at most 512 MiB), Fourier stage on the rank's latitude band
 * (Atlas BandsDistribution rule; atlas_amd__Trans__bands).  sp_dev: the full spectra (replicated, as TransLocal's callers
 * hold them); gp_dev: nb_fields * atlas_amd__Trans__nb_gridpoints(t) values, the rank's band in StructuredColumns owned
 * order.  Asynchronous on the Trans stream; the exchange runs on a second stream.  _many pipelines several transforms:
 * the exchange of transform i overlaps the Legendre stage of i+1 and the Fourier stage of i-1. */
int atlas_amd__Trans__invtrans_distributed(atlas_amd_Trans* t, atlas_amd_Comm* comm, int nb_fields, const double* sp_dev,
                                           double* gp_dev);
int atlas_amd__Trans__invtrans_distributed_many(atlas_amd_Trans* t, atlas_amd_Comm* comm, int ntransforms, int nb_fields,
                                                const double* const* sp_dev, double* const* gp_dev);
/* [r3] the same with the input scattered by zonal wavenumber: sp_shard_dev[i] holds only the wavenumbers this rank
 * owns (m % P == p), the block of every such m in the inner layout of atlas__Trans__invtrans_scalar -- (n = m..T) x (re, im) x
 * nb_fields -- back to back in increasing m: 1/P of the replicated array.  atlas_amd__Trans__spectral_shard gives the layout:
 * moff_out[m] (T+1 entries, may be NULL) = offset of m's block in doubles PER FIELD (multiply by nb_fields), -1 if m is not
 * owned; *size_per_field = doubles per field of the whole shard. */
int atlas_amd__Trans__invtrans_distributed_sharded(atlas_amd_Trans* t, atlas_amd_Comm* comm, int ntransforms, int nb_fields,
                                                   const double* const* sp_shard_dev, double* const* gp_dev);
int atlas_amd__Trans__spectral_shard(const atlas_amd_Trans* t, long long moff_out[], long long* size_per_field);
/* invtrans_distributed_many + per transform, on the library's communication stream -- i.e. beside the Legendre stage of
 * the transforms that follow: the band's grid points transposed into the owned part of field_dev[i], a StructuredColumns
 * field [size_halo][nb_fields] of doubles on the partition that owns this rank's latitude band (the row_bands distribution
 * of the same number of parts), and that field's halo exchange between the ranks (parallel::HaloExchange::execute,
 * HaloExchange.h:191-219).  hx must have been set up with atlas_amd__HaloExchange__setup_comm on the same communicator.
 * Consecutive transforms need distinct gp_dev / field_dev buffers (reuse with period two is ordered by the pipeline).
 * Asynchronous: complete when the Trans stream is (atlas_amd__Trans__synchronize). */
int atlas_amd__Trans__invtrans_distributed_many_halo(atlas_amd_Trans* t, atlas_amd_Comm* c, int ntransforms, int nb_fields,
                                                     const double* const* sp_dev, double* const* gp_dev,
                                                     atlas_amd_HaloExchange* hx, double* const* field_dev);
/* measurement aid (tools/scaling_model.py): the pack kernel of this rank of a wavenumber-sharded Trans (nparts, part, shard=m), alone on
 * the device: ms per launch and bytes packed per launch.  No communicator involved. */
int atlas_amd__Trans__pack_probe(atlas_amd_Trans* t, int nb_fields, int reps, double* ms, long long* bytes);
/* ... and the rank's Fourier stage as the distributed transform runs it: on its latitude band, reading the packed runs of all
 * `nparts` sources (zeros) through per-row offsets and the piece table; ms per stage */
int atlas_amd__Trans__fourier_packed_probe(atlas_amd_Trans* t, int nb_fields, int reps, double* ms);
/* largest message of the transposition (default 512 MiB).  COLLECTIVE over `comm`: every rank calls it, with the same value -- both
 * ends of a pair cut their runs alike; the ranks compare the value inside the call (a mismatch is an error on every rank); takes effect
 * at the next transform. */
int atlas_amd__Trans__set_max_message_bytes(atlas_amd_Trans* t, atlas_amd_Comm* comm, long long bytes);
/* measurement aid of the distributed transform (Trans made with profile=1): accumulated since the last reset, this rank:
 * out[0] pack kernel ms, out[1] exchange ms (send / receive group on the communication stream, incl. waiting for the peers),
 * out[2] transforms counted; per transform: out[3] bytes sent to other ranks, out[4] bytes received from other ranks,
 * out[5] bytes to the busiest peer, out[6] peers with data; out[7] reserved.  (No reference counterpart: TransLocal is
 * single-process, TransLocal.cc:338-340.) */
int atlas_amd__Trans__timings_distributed(atlas_amd_Trans* t, atlas_amd_Comm* comm, double out[8], int reset);
/* [r3] the messages the distributed transform sends (test hook, host only): rank `part` packs, for every latitude row, the
 * wavenumbers m <= row_mmax[row] it owns (m % nparts == part) with `cols` = 2 * nb_fields doubles each -- no dead
 * wavenumbers above the row's Fourier truncation, no pitch padding -- and the rows of band q, one contiguous run, go to
 * rank q.  Offsets in doubles into the rank's packed send buffer / its receive buffer; totals = {doubles sent (all
 * destinations, itself included), doubles received}. */
int atlas_amd__packed_transpose_messages(int nlats, const int row_mmax[], int cols, int nparts, int part, const int bands[],
                                         long long max_message_elems, int capacity, int* peer, long long* send_begin,
                                         long long* send_end, long long* recv_begin, long long* recv_end, int* count,
                                         long long totals[2]);
/* the slab form of the transposition (rows x owned wavenumbers x RP, what atlas_amd/dist_torch.py sends over
 * torch.distributed; test hook): offsets in doubles into the rank's intermediate and into its receive buffer */
int atlas_amd__transpose_messages(int truncation, int RP, int nparts, int part, const int bands[], long long max_message_elems,
                                  int capacity, int* peer, long long* send_begin, long long* send_end, long long* recv_begin,
                                  long long* recv_end, int* count);

/* ---------------------------------------------------------------------------------------------------------------
 * HaloExchange.  Replaces atlas__HaloExchange__* (src/atlas/parallel/HaloExchange.h:429-456).
 * dtype codes: 0 int, 1 long, 2 float, 3 double (the four types Atlas instantiates, detail/Packer.cc:71-95).
 * ------------------------------------------------------------------------------------------------------------- */
atlas_amd_HaloExchange* atlas_amd__HaloExchange__new(void);                 /* atlas__HaloExchange__new    :430 */
void atlas_amd__HaloExchange__delete(atlas_amd_HaloExchange* h);            /* atlas__HaloExchange__delete :431 */
/* atlas__HaloExchange__setup(This, part, remote_idx, base, size)  :432 -- one process (periodic / pole duplicates are
 * exchanged with the same rank) */
int atlas_amd__HaloExchange__setup(atlas_amd_HaloExchange* h, const int part[], const int remote_idx[], int base,
                                   int size);
/* HaloExchange::setup(part, remote_idx, base, parsize, halo_begin)  (HaloExchange.cc:70-72) */
int atlas_amd__HaloExchange__setup_halo_begin(atlas_amd_HaloExchange* h, const int part[], const int remote_idx[],
                                              int base, int size, int halo_begin);
/* multi-process setup in two phases around the caller's allToAll(recvcounts -> sendcounts) and
 * allToAllv(send_requests -> recv_requests)  (HaloExchange.cc:118,156-159): begin computes recvcounts, recvmap and
 * send_requests; finish stores sendcounts and sendmap */
int atlas_amd__HaloExchange__setup_begin(atlas_amd_HaloExchange* h, int nproc, int myproc, const int part[],
                                         const int remote_idx[], int base, int size, int halo_begin);
/* same with part / remote_idx in device memory: ghost list built by wavefront-ballot compaction */
int atlas_amd__HaloExchange__setup_begin_device(atlas_amd_HaloExchange* h, int nproc, int myproc,
                                                const int* part_dev, const int* remote_idx_dev, int base, int size,
                                                int halo_begin);
int atlas_amd__HaloExchange__setup_finish(atlas_amd_HaloExchange* h, const int sendcounts[],
                                          const int recv_requests[]);
int atlas_amd__HaloExchange__nproc(const atlas_amd_HaloExchange* h);
int atlas_amd__HaloExchange__sendcnt(const atlas_amd_HaloExchange* h);
int atlas_amd__HaloExchange__recvcnt(const atlas_amd_HaloExchange* h);
/* what: "sendcounts" | "recvcounts" | "senddispls" | "recvdispls" (nproc ints), "sendmap" (sendcnt),
 *       "recvmap" | "send_requests" (recvcnt) */
int atlas_amd__HaloExchange__get(const atlas_amd_HaloExchange* h, const char* what, int out[]);
/* atlas__HaloExchange__execute_strided_<T>(This, field, var_strides, var_shape, var_rank)   :433-440
 * host arrays, one process; parallel dimension slowest with stride var_shape[0]*var_strides[0] */
int atlas_amd__HaloExchange__execute_strided_int(atlas_amd_HaloExchange* h, int field[], const int var_strides[],
                                                 const int var_shape[], int var_rank);
int atlas_amd__HaloExchange__execute_strided_long(atlas_amd_HaloExchange* h, long field[], const int var_strides[],
                                                  const int var_shape[], int var_rank);
int atlas_amd__HaloExchange__execute_strided_float(atlas_amd_HaloExchange* h, float field[], const int var_strides[],
                                                   const int var_shape[], int var_rank);
int atlas_amd__HaloExchange__execute_strided_double(atlas_amd_HaloExchange* h, double field[],
                                                    const int var_strides[], const int var_shape[], int var_rank);
/* atlas__HaloExchange__execute_adjoint_strided_<T>   :445-452 */
int atlas_amd__HaloExchange__execute_adjoint_strided_int(atlas_amd_HaloExchange* h, int field[],
                                                         const int var_strides[], const int var_shape[],
                                                         int var_rank);
int atlas_amd__HaloExchange__execute_adjoint_strided_long(atlas_amd_HaloExchange* h, long field[],
                                                          const int var_strides[], const int var_shape[],
                                                          int var_rank);
int atlas_amd__HaloExchange__execute_adjoint_strided_float(atlas_amd_HaloExchange* h, float field[],
                                                           const int var_strides[], const int var_shape[],
                                                           int var_rank);
int atlas_amd__HaloExchange__execute_adjoint_strided_double(atlas_amd_HaloExchange* h, double field[],
                                                            const int var_strides[], const int var_shape[],
                                                            int var_rank);
/* atlas__HaloExchange__execute[_adjoint]_{int,float,double}(This, field, var_rank) (:441-443 / :453-455) are declared
 * but never defined in the reference (no shape is passed).  Provided here for the one case that is well defined without a
 * shape, var_rank == 0 (one value per node, field[size]); any other var_rank is an error. */
int atlas_amd__HaloExchange__execute_int(atlas_amd_HaloExchange* h, int field[], int var_rank);
int atlas_amd__HaloExchange__execute_float(atlas_amd_HaloExchange* h, float field[], int var_rank);
int atlas_amd__HaloExchange__execute_double(atlas_amd_HaloExchange* h, double field[], int var_rank);
int atlas_amd__HaloExchange__execute_adjoint_int(atlas_amd_HaloExchange* h, int field[], int var_rank);
int atlas_amd__HaloExchange__execute_adjoint_float(atlas_amd_HaloExchange* h, float field[], int var_rank);
int atlas_amd__HaloExchange__execute_adjoint_double(atlas_amd_HaloExchange* h, double field[], int var_rank);
/* the reference setup and a complete exchange between the ranks of a communicator (HaloExchange.cc:78-172,
 * HaloExchange.h:191-219 / :227-290): pack kernel -> grouped send/recv per peer -> unpack kernel, asynchronous on the
 * object's stream (a transform on another stream overlaps it).  field_dev: device pointer, described as in field_op. */
int atlas_amd__HaloExchange__setup_comm(atlas_amd_HaloExchange* h, atlas_amd_Comm* comm, const int part[],
                                        const int remote_idx[], int base, int size, int halo_begin);
int atlas_amd__HaloExchange__execute_comm(atlas_amd_HaloExchange* h, atlas_amd_Comm* comm, int dtype, void* field_dev,
                                          int rank, const int shape[], const long long strides[], int parallel_dim,
                                          int adjoint);
/* general form of HaloExchange::execute<T,RANK,ParallelDim> / the pack and unpack stages (HaloExchange.h:151-290):
 * op 0 execute, 1 execute_adjoint (both: one process), 2 pack(sendmap), 3 unpack(recvmap), 4 pack_adjoint(recvmap),
 * 5 unpack_adjoint(+= at sendmap), 6 zero_halos.  Buffers hold sendcnt*var_size (ops 2,5) or recvcnt*var_size
 * (ops 3,4) elements.  on_device != 0: device pointers, asynchronous on the object's stream. */
int atlas_amd__HaloExchange__field_op(atlas_amd_HaloExchange* h, int op, int dtype, void* field, int rank,
                                      const int shape[], const long long strides[], int parallel_dim, void* buffer,
                                      int on_device);
void* atlas_amd__HaloExchange__stream(atlas_amd_HaloExchange* h);
int atlas_amd__HaloExchange__set_stream(atlas_amd_HaloExchange* h, void* hip_stream);
int atlas_amd__HaloExchange__synchronize(atlas_amd_HaloExchange* h);

/* ---------------------------------------------------------------------------------------------------------------
 * functionspace::StructuredColumns (global structured grids, band distributions): halo index construction
 * (src/atlas/functionspace/detail/StructuredColumns_setup.cc:88-663, _create_remote_index.cc:37-255) and the
 * halo-exchange dispatch of atlas__FunctionSpace__halo_exchange_field (FunctionSpaceInterface.h:40-43 ->
 * StructuredColumns.cc:811-911).  blocksize: 1 = "equal_bands", nx = "regular_bands", 0 = "row_bands" (whole rows, each
 * with the equal_bands part of its first point: the decomposition atlas_amd__Trans__bands returns)
 * (src/atlas/grid/detail/distribution/BandsDistribution.h:32-34).  Indices are 0-based.
 * ------------------------------------------------------------------------------------------------------------- */
atlas_amd_StructuredColumns* atlas_amd__StructuredColumns__new(const atlas_amd_Grid* grid, int halo,
                                                               int periodic_points, int nparts, int part,
                                                               int blocksize);
/* the same for an explicit grid::Distribution: partition[g] of every grid point g in global order (what Atlas's
 * equal_regions / checkerboard / ... partitioners produce; StructuredColumns_setup.cc:141 distribution.partition(c)).
 * As in the reference the points a part owns in one row must be one contiguous i-range. */
atlas_amd_StructuredColumns* atlas_amd__StructuredColumns__new_distribution(const atlas_amd_Grid* grid, int halo,
                                                                            int periodic_points, int nparts, int part,
                                                                            const int* partition, long long npts);
void atlas_amd__StructuredColumns__delete(atlas_amd_StructuredColumns* fs);
int atlas_amd__StructuredColumns__size_owned(const atlas_amd_StructuredColumns* fs);
int atlas_amd__StructuredColumns__size_halo(const atlas_amd_StructuredColumns* fs);
/* out = {j_begin, j_end, j_begin_halo, j_end_halo} */
int atlas_amd__StructuredColumns__bounds(const atlas_amd_StructuredColumns* fs, int out[4]);
/* out = {i_begin(j), i_end(j), i_begin_halo(j), i_end_halo(j)} */
int atlas_amd__StructuredColumns__row_bounds(const atlas_amd_StructuredColumns* fs, int j, int out[4]);
int atlas_amd__StructuredColumns__index(const atlas_amd_StructuredColumns* fs, int i, int j, int* out);
/* what: "partition" | "ghost" | "index_i" | "index_j" | "remote_idx" (size_halo ints) | "pole_row_nodes" */
int atlas_amd__StructuredColumns__get_int(const atlas_amd_StructuredColumns* fs, const char* what, int out[]);
int atlas_amd__StructuredColumns__nb_pole_row_nodes(const atlas_amd_StructuredColumns* fs);
int atlas_amd__StructuredColumns__global_index(const atlas_amd_StructuredColumns* fs, int64_t out[]); /* 1-based */
int atlas_amd__StructuredColumns__xy(const atlas_amd_StructuredColumns* fs, double out[]);           /* [n][2] */
/* HaloExchange::setup(partition, remote_index, base, sizeHalo, sizeOwned) (StructuredColumns.cc:145-148): complete
 * setup for nparts == 1, local phase (atlas_amd__HaloExchange__setup_begin) otherwise */
int atlas_amd__StructuredColumns__setup_halo_exchange(const atlas_amd_StructuredColumns* fs,
                                                      atlas_amd_HaloExchange* hx, int nparts, int part);
/* FixupHaloForVectors (StructuredColumns.cc:732-808): negate components 0,1 of field(n,[k,]var) in the halo rows
 * beyond the poles; strides in elements; device pointer, asynchronous on hip_stream */
int atlas_amd__StructuredColumns__fixup_halo_for_vectors(atlas_amd_StructuredColumns* fs, int dtype, void* field_dev,
                                                         int levels, long long stride_n, long long stride_k,
                                                         long long stride_v, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
