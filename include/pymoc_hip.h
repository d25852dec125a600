/*
 * pymoc_hip.h -- C-ABI of libpymoc_hip.so, the MI355X (gfx950) engine behind the
 * PyMOC timestep() path.
 *
 * The reference (pymoc 0.0.1rc5) has no FFI of its own: its boundary is the Python
 * object API of four classes.  Every entry point below therefore names the Python
 * method (reference file:line, relative to the reference checkout) whose arithmetic
 * it replaces; the ctypes binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  Pointers inside the pm_* batch
 *     structs are DEVICE pointers obtained from pm_malloc(); everything else
 *     (the struct itself, host buffers of pm_memcpy_*) lives in host memory.
 *   - all floating point data is IEEE fp64; index/flag data is int32.
 *   - batches are "ensemble-major": a field of a batch of n independent members is
 *     one dense array [n][nlev], level fastest.
 *   - every function returns PM_OK (0) or an error code; pm_last_error() gives the
 *     text.  Kernel launches are asynchronous on the given stream (NULL = the
 *     library's default stream); pm_stream_sync / pm_memcpy_d2h are sync points.
 *   - no host pointer is retained after a call returns.
 */
#ifndef PYMOC_HIP_H
#define PYMOC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_OK 0
#define PM_EINVAL 1     /* bad argument (shape, NULL pointer, unsupported size)  */
#define PM_EHIP 2       /* a HIP runtime call failed                              */
#define PM_ENCCL 3      /* an RCCL call failed / librccl could not be loaded      */
#define PM_ENODEV 4     /* no HIP device visible                                  */

typedef void *pm_stream_t;
typedef void *pm_event_t;
typedef void *pm_comm_t;
typedef void *pm_graph_t;

/* ------------------------------------------------------------------ runtime */
const char *pm_version(void);
const char *pm_last_error(void);
int pm_device_count(int *count);
int pm_set_device(int device);
int pm_device_info(char *name, size_t name_len, int *compute_units,
                   size_t *hbm_bytes, int *clock_mhz);

int pm_malloc(void **dptr, size_t bytes);
int pm_free(void *dptr);
int pm_memset(void *dptr, int value, size_t bytes, pm_stream_t stream);
int pm_memcpy_h2d(void *dst, const void *src, size_t bytes, pm_stream_t stream);
int pm_memcpy_d2h(void *dst, const void *src, size_t bytes, pm_stream_t stream);
int pm_memcpy_d2d(void *dst, const void *src, size_t bytes, pm_stream_t stream);
/* Page-locked host memory and a device-to-host copy that does NOT synchronise: the copy is
 * ordered on `stream`, the caller reads `dst` after pm_stream_sync / pm_event_sync.  `dst` must
 * come from pm_host_alloc and stay alive until then (the one exception to "no host pointer is
 * retained": the caller owns the buffer and the sync point).  Used by the diagnostic time
 * series, which the reference fills every Diag_iters steps
 * (examples/run_JansenNadeau_2018.py:192-198, 218-226) and reads after the loop (:268-272).   */
int pm_host_alloc(void **hptr, size_t bytes);
int pm_host_free(void *hptr);
int pm_memcpy_d2h_async(void *dst_pinned, const void *src, size_t bytes, pm_stream_t stream);

/* Row gather on the device: for every item k < nitems (<= PM_PACK_MAX_ITEMS) and row r < nrows
 *   items[k].dst[r][0:nlev] = items[k].src[ sel ? sel[r] : r ][0:nlev]
 * (src rows `src_stride` doubles apart, dst rows dense), ONE launch.  `sel` is a device array
 * of int32 row numbers or NULL.  This is how a diagnostic sample -- the reference's
 * `AMOC_save[:, k] = AMOC.Psi`, ... of run_JansenNadeau_2018.py:219-225 -- is appended to the
 * device-resident time series (selected members only), and how the per-rank send buffer of the
 * diagnostic exchange is packed.                                                             */
#define PM_PACK_MAX_ITEMS 8
typedef struct pm_row_copy {
  const double *src;   /* [.][src_stride] */
  double *dst;         /* [nrows][nlev]   */
  int32_t nlev;
  int32_t src_stride;
} pm_row_copy;
int pm_rows_pack(const pm_row_copy *items, int32_t nitems, const int32_t *sel, int32_t nrows,
                 pm_stream_t stream);

int pm_stream_create(pm_stream_t *stream);
/* priority > 0: the device's highest stream priority (its workgroups are dispatched before those
   of normal streams when both have work pending), 0: as pm_stream_create */
int pm_stream_create_priority(pm_stream_t *stream, int priority);
int pm_stream_destroy(pm_stream_t stream);
int pm_stream_sync(pm_stream_t stream);
int pm_device_sync(void);

int pm_event_create(pm_event_t *event);
int pm_event_destroy(pm_event_t event);
int pm_event_record(pm_event_t event, pm_stream_t stream);
int pm_event_sync(pm_event_t event);
/* work submitted to `stream` after this call waits for `event` (hipStreamWaitEvent): lets two
 * independent launches -- Psi_SO.solve and Psi_Thermwind of one overturning update -- run side
 * by side on two streams and join before the columns step                                    */
int pm_stream_wait_event(pm_stream_t stream, pm_event_t event);
int pm_event_elapsed_ms(pm_event_t start, pm_event_t stop, float *ms);

/* hipGraph capture of a launch sequence issued on `stream` */
int pm_graph_begin_capture(pm_stream_t stream);
int pm_graph_end_capture(pm_stream_t stream, pm_graph_t *graph);
int pm_graph_launch(pm_graph_t graph, pm_stream_t stream);
int pm_graph_destroy(pm_graph_t graph);

/* ------------------------------------------------------------------ Column
 * Replaces pymoc.modules.Column time stepping:
 *   Column.convect       src/pymoc/modules/column.py:251-271
 *   Column.vertadvdiff   src/pymoc/modules/column.py:210-249
 *   Column.horadv        src/pymoc/modules/column.py:288-313
 *   Column.timestep      src/pymoc/modules/column.py:315-348  (order: convect,
 *                        vertadvdiff, horadv)
 * A batch is ncols independent columns on ONE shared vertical grid z[nz].
 */
#define PM_COL_DO_CONV 1   /* per-column flag: timestep(do_conv=True)             */
#define PM_COL_BZBOT 2     /* per-column flag: bzbot is not None (column.py:232)  */
#define PM_COL_STATIC_IN_RANGE 4 /* per-column HINT: z, kappa, Area, d(A kappa)/dz of every
                              coefficient set and bs / bbot / bzbot / N2min are finite and zero
                              or of magnitude in [2^-200, 2^200] (Area and the grid spacings not
                              zero).  The column kernels divide by static denominators with an
                              exact 4-instruction sequence that is IEEE-identical inside that
                              window; they test their operands and take IEEE division otherwise.
                              With the hint the one-step streaming kernel tests only what
                              changes from launch to launch (b, wA); without it, everything.   */
#define PM_COL_UNIFORM_AREA 8 /* per-column HINT: Area(z) is constant in z (every reference script);
                              the column kernels (G = 64) then read area[col][0] only.  Not
                              verified (that would be the read it saves): a wrong hint gives the
                              result for Area = area[col][0].                                   */

#define PM_OP_CONVECT 1      /* run convect() on columns flagged PM_COL_DO_CONV   */
#define PM_OP_VERTADVDIFF 2  /* run vertadvdiff(wA, dt, do_conv=flag)             */
#define PM_OP_HORADV 4       /* run horadv(vdx_in, b_in, dt)                      */
#define PM_OP_TIMESTEP 7     /* what Column.timestep does (horadv iff vdx given)  */
#define PM_OP_WEFF 8         /* modifier: the `wA` argument holds weff = wA - d(A kappa)/dz of
                                the coefficient set in use (pm_column_weff), which is all the step
                                needs of the two (column.py:241); d(A kappa)/dz is then not read.
                                wA is static between two overturning updates in every reference
                                driver, so weff is too.                                          */
#define PM_OP_WA_PSI 32      /* modifier for two-column ensembles (rows [0, ncols/2) basin, [ncols/2,
                                ncols) north): the `wA` argument holds the isopycnal overturning
                                Psi_iso [ncols][nz] in Sv and `b_in` (horadv's slot, no horadv with
                                this modifier) Psi_SO [ncols/2][nz] or NULL; the kernel forms the
                                drivers' forcing itself, wA_basin = (Psi_iso - Psi_SO) * 1e6,
                                wA_north = -Psi_iso * 1e6 (example_twocol_plusSO.py:105-106) --
                                what pm_thermwind_update's wA1 / wA2 outputs hold, without the
                                thermal wind having to wait for Psi_SO.  Launches of >= 3 steps. */
#define PM_OP_WA_TWOBASIN 64 /* modifier for the two-basin driver's three-column ensembles (rows [0, n)
                                Atlantic, [n, 2n) north, [2n, 3n) Pacific, n = ncols / 3): `wA` holds the
                                AMOC's isopycnal overturnings [2n][nz] (rows [0, n) on the Atlantic
                                column, [n, 2n) on the northern one), `vdx_in` (horadv's slot: no horadv
                                with this modifier) the zonal overturning's [2n][nz] (Atlantic, Pacific)
                                and `b_in` the two sectors' Psi_SO [2n][nz] (Atlantic, Pacific), all
                                in Sv; the kernel forms the driver's forcing itself,
                                  wA_Atl = (iso_A + zon_A - SO_A) * 1e6,  wA_north = -iso_N * 1e6,
                                  wA_Pac = (-zon_P - SO_P) * 1e6   (twobasin_NadeauJansen.py:103-105)
                                -- pm_twobasin_forcing's operations, without that launch.  Launches
                                of >= 3 steps.                                                     */
#define PM_OP_CONTRACTED 16  /* modifier, OPT-IN tolerance mode: launches of >= 3 plain timesteps
                                (one wave per column, nz <= 256) use the contracted update
                                b_i += cu_i (b_{i+1}-b_i) + cl_i (b_i-b_{i-1}) with per-launch
                                coefficients instead of the reference's operation order (3
                                instead of 21 instructions per level).  Agrees with the reference
                                to rounding (<= 1e-12 relative over BASELINE's runs), not bit for
                                bit; without the flag every result is bit-identical to NumPy.    */

#define PM_COLS_ALL_UNIFORM_AREA 1 /* pm_columns.reserved, batch-wide HINT: EVERY column carries
                              PM_COL_UNIFORM_AREA.  With it, PM_OP_WEFF and kappa_base /
                              kappa_profile a one-step launch on a large batch keeps only b and
                              weff of a column in flight (24 nz B per column-step)               */

#define PM_COLS_DIV3_PROVEN 2 /* pm_columns.reserved, batch-wide HINT (with PM_COLS_ALL_UNIFORM_AREA): the
                              caller has established with pm_div3_proven() that EVERY static
                              denominator of the batch's column step -- the grid spacings z[i+1] -
                              z[i], the centred spacings 0.5 ((z[i+1] - z[i]) + (z[i] - z[i-1])) and
                              every column's Area -- admits the 3-instruction exact quotient
                              (below).  Fused launches then divide in 3 instead of 4 instructions;
                              results are bit-identical (the quotient is the IEEE one).           */

typedef struct pm_columns {
  int32_t ncols;         /* independent columns in the batch                      */
  int32_t nz;            /* levels per column (2 <= nz <= 1024)                   */
  int32_t nsel;          /* coefficient sets per column (1, or 2 for JN2018)      */
  int32_t reserved;      /* batch-wide hint bits (PM_COLS_*), else 0              */
  const double *z;       /* [nz]              shared grid, ascending              */
  double *b;             /* [ncols][nz]       buoyancy, updated in place          */
  const double *kappa;   /* [nsel][ncols][nz] kappa(z)                            */
  const double *area;    /* [ncols][nz]       Area(z)                             */
  const double *dAkappa; /* [nsel][ncols][nz] np.gradient(Area*kappa, z)          */
  const double *bs;      /* [ncols] surface buoyancy                              */
  const double *bbot;    /* [ncols] bottom buoyancy (used unless PM_COL_BZBOT)    */
  const double *bzbot;   /* [ncols] bottom stratification (may be NULL)           */
  const double *N2min;   /* [ncols] convective-adjustment stratification          */
  const int32_t *flags;  /* [ncols] PM_COL_* bits                                 */
  const int32_t *ksel;   /* [ncols] coefficient set in use (NULL -> set 0)        */
  int32_t *nonfinite;    /* [ncols] out: 1 if b holds a non-finite value at the
                            end of the call (NULL -> not reported)                */
  const double *kappa_base;    /* [ncols] or NULL */
  const double *kappa_profile; /* [nz] or NULL.  Both given (nsel = 1): a HINT that
                            kappa[col][i] == kappa_base[col] + kappa_profile[i] BIT FOR BIT (one
                            fp64 addition) -- a parameter sweep over a background diffusivity, the
                            kappa sweep of BASELINE's ensembles.  Launches that stream the
                            coefficient arrays (one or two steps per launch on a large batch) then
                            FORM kappa instead of reading it: with PM_OP_WEFF 24 nz B per
                            column-step (b and weff in, b out: SURVEY 8d's algorithmic bytes) instead
                            of 32 nz.  The caller vouches for the identity (the Python ColumnBatch
                            verifies it on the host arrays); other launches read kappa as ever.  */
} pm_columns;

/* nsteps repetitions of the selected ops with wA (and vdx_in/b_in) held fixed, the
 * whole loop fused in one launch with b resident in registers.
 *   wA      [ncols][nz]  m^3/s
 *   vdx_in  [ncols][nz]  or NULL; b_in [ncols][nz] required when vdx_in given
 *   lanes_per_col  16, 32 or 64 lanes cooperate on one column; 0 = choose        */
int pm_column_steps(const pm_columns *cols, const double *wA, const double *vdx_in,
                    const double *b_in, double dt, int32_t nsteps, int32_t ops,
                    int32_t lanes_per_col, pm_stream_t stream);

/* weff[ncols][nz] = wA - d(A kappa)/dz of each column's coefficient set in use (column.py:241),
 * for pm_column_steps(..., ops | PM_OP_WEFF).                                                 */
int pm_column_weff(const pm_columns *cols, const double *wA, double *weff, pm_stream_t stream);

/* The work decomposition pm_column_steps would use for a batch of this shape: lanes
 * cooperating on one column (`lanes_per_col` = 0 lets the library choose) and levels held
 * per lane; reporting only (bench.py names the kernel instantiation with it).            */
int pm_column_kernel_shape(int32_t ncols, int32_t nz, int32_t lanes_per_col,
                           int32_t *lanes, int32_t *levels_per_lane);
/* Name of the kernel instantiation a pm_column_steps call of this shape launches (reporting
 * only): "k_column_steps<G,P,DIV,PLAIN>" or, for 1-2 steps on a large ensemble,
 * "k_column_stream<P>".  has_horadv: vdx_in given.                                       */
int pm_column_kernel_name(int32_t ncols, int32_t nz, int32_t lanes_per_col, int32_t nsteps,
                          int32_t ops, int32_t has_horadv, char *name, size_t name_len);

/* ------------------------------------------------------------------ Psi_Thermwind
 * Replaces pymoc.modules.Psi_Thermwind for n independent members on one grid z[nz]:
 *   Psi_Thermwind.solve  src/pymoc/modules/psi_thermwind.py:125-135  (PM_TW_SOLVE)
 *   Psi_Thermwind.Psib   src/pymoc/modules/psi_thermwind.py:137-185  (PM_TW_PSIB)
 *   Psi_Thermwind.Psibz  src/pymoc/modules/psi_thermwind.py:187-208  (PM_TW_PSIBZ,
 *                        needs PM_TW_PSIB in the same call)
 * and, optionally, the drivers' coupling to the columns (examples/example_twocol.py
 * :87-88, example_twocol_plusSO.py:105-106): wA1 = (psibz1 - Psi_SO)*1e6 (Psi_SO NULL
 * -> psibz1*1e6), wA2 = -psibz2*1e6.  Without PM_TW_SOLVE, Psi is an input.          */
#define PM_TW_SOLVE 1
#define PM_TW_PSIB 2
#define PM_TW_PSIBZ 4
#define PM_TW_WA_PSI 8 /* with PM_TW_SOLVE: wA1 = Psi*1e6 (example_timestepping.py:75) */

typedef struct pm_thermwind {
  int32_t n;            /* members                                               */
  int32_t nz;           /* levels (2 <= nz <= 1024)                              */
  int32_t nb;           /* isopycnal classes of Psib (reference default 500)     */
  int32_t reserved;
  const double *z;      /* [nz]    shared grid                                   */
  const double *b1;     /* [n][nz] buoyancy of the basin column                  */
  const double *b2;     /* [n][nz] buoyancy of the northern column               */
  const double *f;      /* [n]     Coriolis parameter                            */
  double *Psi;          /* [n][nz] overturning, Sv (out with PM_TW_SOLVE, else in)*/
  double *bgrid;        /* [n][nb] out, may be NULL                              */
  double *psib;         /* [n][nb] out, may be NULL.  With bgrid AND psib NULL and PM_TW_PSIBZ
                           a member whose upstream cells lie in chain order sums only the
                           classes Psibz's interpolations read (70-135 of 500 on BASELINE's
                           ensembles): psibz1 / psibz2 / wA1 / wA2 are the same bits either way */
  double *psibz1;       /* [n][nz] out, may be NULL                              */
  double *psibz2;       /* [n][nz] out, may be NULL                              */
  const double *Psi_SO; /* [n][nz] in, may be NULL                               */
  double *wA1;          /* [n][nz] out, may be NULL                              */
  double *wA2;          /* [n][nz] out, may be NULL                              */
  const double *b1_mid; /* [n][nz] b1 at the midpoints z[k] + (z[k+1]-z[k])/2 of the     */
  const double *b2_mid; /*         intervals, for CALLABLE profiles (solve_bvp evaluates */
                        /*         them there); NULL = linear between the levels        */
  double *dPsi;         /* [n][nz] out with PM_TW_SOLVE, may be NULL: d(Psi 1e6)/dz at the  */
                        /*         levels (the second component of solve_bvp's solution)    */
} pm_thermwind;

int pm_thermwind_update(const pm_thermwind *tw, int32_t ops, pm_stream_t stream);

/* solve_bvp's rms residual of every interval of ONE member's mesh x[m] (scipy 1.15.3
 * _bvp.py:estimate_rms_residuals: 5-point Lobatto rule on the C1 cubic spline of (y, f)) for the
 * thermal-wind problem y0' = y1, y1' = g(x): y0[m] = Psi (in Sv, as stored) and y1[m] = dPsi from
 * pm_thermwind_update on the mesh, g[m] = (b2 - b1)/f at the nodes, g_lob[2][m-1] at the two
 * inner Lobatto points x_mid +- h sqrt(3/7)/2 of every interval (CALLABLE profiles: sampled by
 * the host, which owns solve_bvp's mesh loop: pymoc_amd/modules/psi_thermwind.py).  rms[m-1].  */
int pm_thermwind_residuals(int32_t m, const double *x, const double *y0, const double *y1,
                           const double *g, const double *g_lob, double *rms,
                           pm_stream_t stream);

/* ------------------------------------------------------------------ Psi_SO
 * Replaces pymoc.modules.Psi_SO for n independent members on shared grids z[nz], y[ny]:
 *   Psi_SO.ys          src/pymoc/modules/psi_SO.py:106-140
 *   Psi_SO.calc_Ekman  src/pymoc/modules/psi_SO.py:218-243   (PM_SO_OP_EKMAN)
 *   Psi_SO.calc_GM     src/pymoc/modules/psi_SO.py:277-331   (PM_SO_OP_GM; without
 *                      PM_SO_OP_EKMAN it reads Psi_Ek as the caller's self.Psi_Ek)
 *   Psi_SO.solve       src/pymoc/modules/psi_SO.py:333-354   (PM_SO_OP_SOLVE)
 * Scalar options of the constructor are shared by the batch; KGM and tau vary per member. */
#define PM_SO_HAS_C 1          /* c is not None: F2010 boundary-value smoother       */
#define PM_SO_BVP_WITH_EK 2
#define PM_SO_HAS_HSILL 4
#define PM_SO_HAS_HEK 8
#define PM_SO_HAS_HTAPERTOP 16
#define PM_SO_HAS_HTAPERBOT 32
#define PM_SO_TAU_ARRAY 64     /* tau is [n][ny] on y instead of one scalar per member */

#define PM_SO_OP_EKMAN 1
#define PM_SO_OP_GM 2
#define PM_SO_OP_SOLVE 3

typedef struct pm_psi_so {
  int32_t n, nz, ny, flags;
  int32_t bvp_refine;   /* GM BVP mesh: 0 / -1 = follow scipy solve_bvp's own adaptive mesh
                           (1e-14 from the reference); R > 0 = fixed R-fold refinement
                           of the column grid (2-3x faster, ~1e-6 from the reference)   */
  int32_t reserved;
  const double *z;      /* [nz] */
  const double *y;      /* [ny] */
  const double *b;      /* [n][nz] basin buoyancy at the northern end of the channel */
  const double *bs;     /* [n][ny] surface buoyancy                                  */
  const double *tau;    /* [n] or, with PM_SO_TAU_ARRAY, [n][ny]                     */
  const double *KGM;    /* [n] */
  double f, rho, L, smax, c, Hsill, HEk, Htapertop, Htaperbot;
  double *Psi;          /* [n][nz] out, Sv                                           */
  double *Psi_Ek;       /* [n][nz] out, Sv (in when PM_SO_OP_GM alone)               */
  double *Psi_GM;       /* [n][nz] out, Sv                                           */
  double *Ek_raw;       /* [n][nz] out, m^3/s: calc_Ekman() return value (may be NULL)*/
  double *GM_raw;       /* [n][nz] out, m^3/s: calc_GM() return value (may be NULL)  */
  double *ys;           /* [n][nz] out: outcrop latitude of b[i] (may be NULL)        */
  int32_t *status;      /* [n] out: bit0 bs not monotone north of its minimum (several
                           roots: ys then follows brentq's own iteration, like the
                           reference), bit1 non-finite Psi, bit2 NaN in bs, bit3 the adaptive GM
                           mesh would exceed solve_bvp's own max_nodes = 1000 (it stops
                           there too; the solution of the last mesh is returned).  For
                           nz <= 128 the first launch follows meshes up to 256 nodes and
                           a follow-up launch redoes the members it flagged, which needs
                           `status` (without it bit 3 semantics cannot be repaired and
                           the 256-node solution stands) (may be NULL)                   */
  const double *ys_in;      /* [n][nz] optional: outcrop latitudes computed by the caller   */
  const double *tau_ave_in; /* [n][nz] optional: wind stress averaged from ys to y[-1].
                               For CALLABLE bs / tau, which only the host can evaluate: the
                               drop-in class root-finds and averages exactly like the
                               reference (psi_SO.py:106-140, :238-240) and hands the results
                               in; both NULL = computed on the device from bs[n][ny], tau  */
} pm_psi_so;

int pm_psi_so_update(const pm_psi_so *so, int32_t ops, pm_stream_t stream);

/* ------------------------------------------------------------------ SO_ML
 * Replaces pymoc.modules.SO_ML.timestep / advdiff (src/pymoc/modules/SO_ML.py:198-303,
 * with set_boundary_conditions :77-98, calc_advective_tendency :100-134 and the
 * Crank-Nicolson calc_implicit_diffusion :136-196 as a Thomas sweep) for n members on a
 * shared uniform grid y[ny]; the basin profiles live on z with nz levels.               */
typedef struct pm_so_ml {
  int32_t n, nz, ny, reserved;
  const double *y;         /* [ny] uniform meridional grid                            */
  double *bs;              /* [n][ny] mixed-layer buoyancy, updated in place           */
  double *Psi_s;           /* [n][ny] out: overturning at the surface, Sv (may be NULL) */
  const double *b_basin;   /* [n][nz] buoyancy of the adjoining basin                  */
  const double *Psi_b;     /* [n][nz] SO overturning on the basin's levels, Sv         */
  const double *surflux;   /* [n][ny] */
  const double *rest_mask; /* [n][ny] */
  const double *b_rest;    /* [n][ny] */
  double Ks, h, L, v_pist;
  int32_t *status;         /* [n] out: 1 where the reference raises IndexError (Psi_b all
                              zero / no upwelling level; state left untouched), 2 non-finite
                              result, 16 (pm_jn2018_steps only) a wrong
                              PM_JN_UNIFORM_AREA hint, 32 (pm_jn2018_steps' uniform-Area
                              kernel only) an operand of the column steps -- state, forcing,
                              coefficients, grid, dt -- outside the window [2^-200, 2^200] in
                              which the kernel's 4-instruction division is IEEE-identical: the
                              member's steps from there on were taken by the call's follow-up
                              launch with true divisions (the reference's arithmetic for any
                              operand, like pm_column_steps' own fallback); informational.
                              Bits 8.. are scratch of that hand-over and are cleared by it
                              (may be NULL: pm_jn2018_steps then uses its general kernel)   */
} pm_so_ml;

int pm_so_ml_step(const pm_so_ml *ml, double dt, pm_stream_t stream);

/* Per-step bottom boundary condition and bottom-boundary-layer diffusivity selection of the
 * Jansen & Nadeau driver, examples/run_JansenNadeau_2018.py:233-254.  Columns are stored
 * basin rows [0, n) then north rows [n, 2n); coefficient set 0 = kappa, 1 = kappaeff.    */
typedef struct pm_jn2018_bc {
  int32_t n, nz, ny, reserved;
  const double *Psi_SO;    /* [n][nz] PsiSO.Psi                                        */
  const double *Psi_res_b; /* [n][nz] AMOC.Psibz()[0]                                  */
  const double *Psi_res_n; /* [n][nz] AMOC.Psibz()[1]                                  */
  const double *b_basin;   /* [n][nz] */
  const double *b_north;   /* [n][nz] */
  const double *bs_SO;     /* [n][ny] channel.bs                                       */
  double *bbot;            /* [2n] in/out: Column.bbot of basin / north columns        */
  int32_t *ksel;           /* [2n] in/out: coefficient set of basin / north columns    */
} pm_jn2018_bc;

int pm_jn2018_bc_switch(const pm_jn2018_bc *bc, pm_stream_t stream);

/* Fused Jansen & Nadeau time loop (examples/run_JansenNadeau_2018.py:228-261): nsteps x
 * [bottom-BC switch -> basin.timestep(do_conv) -> north.timestep(do_conv) ->
 * channel.timestep] per member in ONE launch, with wA / Psi_SO / Psibz held fixed (they only
 * change at MOC updates, :204-217).  Bit-identical to issuing pm_jn2018_bc_switch +
 * pm_column_steps + pm_so_ml_step per step.  `cols` holds 2n columns (basin rows [0,n),
 * north rows [n,2n), nsel = 2, bbot / ksel updated in place); `ml.b_basin` / `ml.Psi_b` are
 * ignored (the basin column and Psi_SO are used).  nz <= 256.                          */
#define PM_JN_UNIFORM_AREA 1 /* hint: Area(z) of every column is constant in z (true for every
                                reference script); a kernel variant then keeps it in scalar
                                registers.  A wrong hint is detected: the launch fails the
                                members' status with bit 4 (16) and leaves them untouched.  */
#define PM_JN_CONTRACTED 2   /* OPT-IN tolerance mode of the uniform-Area kernel: the two columns
                                step in the contracted form of PM_OP_CONTRACTED (agreement with
                                the reference to rounding, not bit for bit; the mixed layer and
                                the bottom-BC switch keep the reference's operation order)     */
#define PM_JN_SHARED_COEF 4  /* hint of the uniform-Area kernel: the kappa / dAkappa / Area rows of
                                every basin column equal basin column 0's and those of every
                                northern column equal northern column 0's (a parameter sweep
                                over forcing and boundary values, as run_JansenNadeau_2018.py's
                                parameters are: one kappa(z) for all members).  The kernel then
                                reads only rows 0 and n -- they stay in L2 -- instead of six rows
                                per member from HBM.  The CALLER vouches for it (verifying it on
                                the device would read the rows it is meant to save); the Python
                                driver compares the host arrays.                                */
#define PM_JN_SPLIT_LANES 8  /* OPT-IN lane layout of the uniform-Area kernel (round 5): the two
                                columns of a member step together, basin on lanes 0..31 and
                                north on lanes 32..63 with ceil(nz/32) levels per lane, instead
                                of one after the other on 64 lanes each.  Same arithmetic per
                                level, bit-identical results; 65 <= nz <= 128 or 193 <= nz <= 224,
                                other shapes ignore the hint.                                  */
#define PM_JN_DIV3_PROVEN 16 /* hint of the uniform-Area kernel: pm_div3_proven and pm_recip_check hold
                              for every static denominator of the fused loop -- the grid spacings and
                              centred spacings of z, both columns' Area of every member, and the
                              mixed layer's h, L and y[1] - y[0]: its quotients then take 3 instead
                              of 4 instructions (bit-identical)                                    */
typedef struct pm_jn2018 {
  int32_t n, hints, reserved1, reserved2;
  pm_columns cols;
  const double *wA;        /* [2n][nz] */
  const double *Psi_SO;    /* [n][nz]  */
  const double *Psi_res_b; /* [n][nz]  */
  const double *Psi_res_n; /* [n][nz]  */
  pm_so_ml ml;
} pm_jn2018;

int pm_jn2018_steps(const pm_jn2018 *jn, double dt, int32_t nsteps, pm_stream_t stream);

/* ------------------------------------------------------------------ whole coupled runs
 * The reference's coupled drivers are loops "every MOC_up_iters steps refresh the overturning
 * diagnostics, then step" (examples/example_twocol.py:85-96, run_JansenNadeau_2018.py:201-261).
 * pm_twocol_run / pm_jn2018_run carry every member through MANY such intervals in ONE launch:
 * a wavefront owns a member and runs the diagnostic phase and the stepping phase of each
 * interval back to back (members never interact, so no launch boundary is needed between
 * them).  The phases are the device functions of pm_psi_so_update, pm_thermwind_update and
 * pm_jn2018_steps / pm_column_steps, so the result is bit-identical to issuing those calls in
 * the drivers' order.  Schedule of one launch:
 *     n_first steps;  then n_updates x [refresh the diagnostics; steps], where the last block
 *     steps n_last (possibly 0) and the others m_steps.
 * A driver ends a launch where it wants to look at the state (diagnostic output, RCCL gather). */
typedef struct pm_run_schedule {
  int32_t n_first, n_updates, m_steps, n_last;
} pm_run_schedule;

/* example_twocol.py:85-96: per member a basin column (row m of `cols`) and a northern column
 * (row n + m), Psi_Thermwind solve / Psib / Psibz between them (`tw`: b1 = cols.b, b2 = cols.b +
 * n nz; wA1 = wA, wA2 = wA + n nz; psibz1 / psibz2 / Psi are outputs; Psi_SO NULL), forcing
 * wA[2n][nz] (in for the n_first steps, rewritten by every refresh).  Requires: Area constant in
 * z, no bzbot, nz <= 256 with (7 nz' + 16 max(...)) doubles of LDS <= 160 KB (checked).
 * status[n] (must be zeroed by the caller; bits are OR-ed in): 2 non-finite state, 16 columns this
 * kernel cannot step (left untouched), 32 some interval stepped with IEEE divisions because an
 * operand was outside the exact-division window (same guard as pm_column_steps).            */
typedef struct pm_twocol_loop {
  pm_columns cols;
  pm_thermwind tw;
  const double *wA;
  double dt;
  pm_run_schedule sched;
  int32_t *status;
} pm_twocol_loop;
int pm_twocol_run(const pm_twocol_loop *run, pm_stream_t stream);

/* run_JansenNadeau_2018.py:201-261: per interval PsiSO.solve (`so`: b = the basin rows of jn.cols,
 * bs = jn.ml.bs, Psi = jn.Psi_SO), AMOC.solve / Psibz (`tw`: b1 / b2 = the basin / northern rows,
 * Psi_SO = jn.Psi_SO, wA1 / wA2 = jn.wA rows, psibz1 / psibz2 = jn.Psi_res_b / Psi_res_n), then the
 * fused step loop of pm_jn2018_steps.  `jn` as for pm_jn2018_steps with PM_JN_UNIFORM_AREA
 * (ny <= 64, 4 <= nz <= 256); `so` without the boundary-value smoother (c = None).
 * jn.ml.status as for pm_jn2018_steps, but OR-ed over the intervals (zero it first).        */
typedef struct pm_jn2018_loop {
  pm_jn2018 jn;
  pm_thermwind tw;
  pm_psi_so so;
  double dt;
  pm_run_schedule sched;
} pm_jn2018_loop;
int pm_jn2018_run(const pm_jn2018_loop *run, pm_stream_t stream);

/* One MOC update of the Jansen & Nadeau driver (run_JansenNadeau_2018.py:206-217) in ONE launch:
 * pm_psi_so_update(so, PM_SO_OP_SOLVE) followed by pm_thermwind_update(tw, tw_ops) for the same
 * members (same device functions, bit-identical results; `tw` may read so->Psi as its Psi_SO).
 * `so` without the boundary-value smoother (c = None), so->n == tw->n, so->nz == tw->nz <= 256.
 * Other shapes: PM_EINVAL -- issue the two calls.                                              */
int pm_so_tw_update(const pm_psi_so *so, const pm_thermwind *tw, int32_t tw_ops,
                    pm_stream_t stream);

/* LDS bytes per block (16 members) the phases of a run kernel need for this shape: kind 0 =
 * pm_twocol_run, 1 = pm_jn2018_run (ny: points of the mixed layer).  A launch is refused
 * (PM_EINVAL) above 160 KB; a driver asks first and keeps its launch sequence otherwise.
 * 0 = shape not supported at all.                                                           */
int pm_run_lds_bytes(int32_t kind, int32_t nz, int32_t nb, int32_t ny, size_t *bytes);

/* Column forcing of the two-column drivers as an array (what PM_OP_WA_PSI forms inside the
 * column kernel; for launches of fewer than 3 steps): wA[0:n] = (Psi_iso[0:n] - Psi_SO) * 1e6
 * (Psi_SO may be NULL), wA[n:2n] = -Psi_iso[n:2n] * 1e6; Psi_iso, wA [2n][nz], Psi_SO [n][nz]. */
int pm_twocol_forcing(int32_t n, int32_t nz, const double *Psi_iso, const double *Psi_SO,
                      double *wA, pm_stream_t stream);

/* Column forcing of the two-basin driver, examples/twobasin_NadeauJansen.py:103-105:
 *   wA_Atl = (Psi_iso_Atl + Psi_zonal_Atl - SO_Atl.Psi)*1e6
 *   wAN    = -Psi_iso_N*1e6
 *   wA_Pac = (-Psi_zonal_Pac - SO_Pac.Psi)*1e6          all arrays [n][nz]            */
int pm_twobasin_forcing(int32_t n, int32_t nz, const double *Psi_iso_Atl,
                        const double *Psi_zonal_Atl, const double *SO_Atl,
                        const double *Psi_iso_N, const double *Psi_zonal_Pac,
                        const double *SO_Pac, double *wA_Atl, double *wAN, double *wA_Pac,
                        pm_stream_t stream);

/* ---------------------------------------------------- equilibrium column
 * Column.solve_equi (src/pymoc/modules/column.py:187-208 with ode :161-164 and bc :124-159):
 * the steady advective-diffusive profile, which the reference obtains from
 * scipy.integrate.solve_bvp.  pm_column_equi_pass is ONE mesh pass of that solver for n
 * members, each on its own mesh: the exact solution of the 4th-order collocation system
 * (the ODE is linear), the rms residual estimate of every interval and the number of nodes
 * solve_bvp would insert.  The caller refines the flagged meshes and calls again until nadd
 * is 0 (pymoc_amd/equilibrium.py does, following solve_bvp's loop).
 * Point sets of a mesh x[m]: 0 = the m nodes; 1 = the m-1 interval middles x[i] + h/2;
 * 2 / 3 = the Lobatto points middle +- (h/2) sqrt(3/7).                                  */
typedef struct pm_column_equi {
  int32_t n, nz, mmax, reserved;
  const int32_t *m;       /* [n] mesh nodes of each member, 2 <= m <= mmax <= 1024        */
  const int32_t *active;  /* [n] 0 = skip this member (may be NULL: all active)           */
  const double *x;        /* [n][mmax] ascending mesh, contains every level of z          */
  const double *Ak;       /* [4][n][mmax] Column.Akappa at the four point sets            */
  const double *dAk;      /* [4][n][mmax] Column.dAkappa_dz at the four point sets        */
  const double *wA;       /* [4][n][mmax] wA at the four point sets, or NULL to use wA_z  */
  const double *wA_z;     /* [n][nz] wA on the column grid (np.interp'ed on the device)   */
  const double *z;        /* [nz] column grid (needed with wA_z)                          */
  const double *bs;       /* [n] */
  const double *bbot;     /* [n] */
  const double *bzbot;    /* [n] or NULL */
  const int32_t *flags;   /* [n] PM_COL_BZBOT: b'(-H) = bzbot replaces b(-H) = bbot (or NULL) */
  const int32_t *zidx;    /* [n][nz] position of column level k in the member's mesh      */
  double tol;             /* solve_bvp's tol (the reference uses the default 1e-3)        */
  double *y;              /* [n][2][mmax] out: b and db/dz on the mesh (may be NULL)      */
  double *rms;            /* [n][mmax] out: rms residual of every interval (may be NULL)  */
  int32_t *nadd;          /* [n] out: nodes solve_bvp would insert; 0 = converged (or NULL) */
  double *b;              /* [n][nz] out: Column.b on the column grid (may be NULL)       */
  double *bz;             /* [n][nz] out: Column.bz (may be NULL)                         */
} pm_column_equi;

int pm_column_equi_pass(const pm_column_equi *eq, pm_stream_t stream);

/* Equi_Column.solve (src/pymoc/modules/equi_column.py:408-435; ode :349-406, bc :286-347,
 * alpha :231-249, bz :251-284, non-dimensional profiles :116-185): the equilibrium overturning
 * of a basin as a 4th-order boundary-value problem on z* in [-1, 0], with the cell depth H
 * either given or an unknown parameter, which the reference solves with
 * scipy.integrate.solve_bvp.  pm_equi_column_newton is ONE mesh iteration of solve_bvp for n
 * members (each on its own mesh): solve_newton (forward-difference Jacobians, damped Newton,
 * <= 8 iterations / 4 Jacobians), the rms residual estimate of every interval, the number of
 * nodes solve_bvp would insert, and the largest boundary residual.  The caller owns
 * solve_bvp's outer loop (pymoc_amd/equi_column.py): insert nodes, transfer the solution to
 * the new mesh with the C1 cubic spline of (y, yp), call again.
 * Profiles are scalars or samples on the model grid zg (np.interp'ed on the device).      */
#define PM_EQ_HFREE 1       /* H is an unknown parameter (p in/out), else p = H fixed      */
#define PM_EQ_HAS_BBOT 2    /* bottom condition d2Psi(-1) = b_bot/H, else d3Psi(-1) = -bz(H) */
#define PM_EQ_KAPPA_ARRAY 4 /* kappa_z / dkappa_z tables are used instead of the scalar kappa */
#define PM_EQ_PSI_ARRAY 8   /* psi_z table is used (otherwise psi_so = 0)                  */
typedef struct pm_equi_column {
  int32_t n, nzg, mmax, reserved;
  const int32_t *m;       /* [n] mesh nodes of each member, 3 <= m <= mmax                */
  const int32_t *active;  /* [n] 0 = skip this member (may be NULL)                       */
  const double *x;        /* [n][mmax] mesh on [-1, 0]                                    */
  double *y;              /* [n][4][mmax] in: initial guess; out: final Newton iterate    */
  double *yp;             /* [n][4][mmax] out: ode(x, y) at the nodes (spline derivatives) */
  double *p;              /* [n] in/out: H (initial guess -> solution with PM_EQ_HFREE)   */
  const double *f;        /* [n] Coriolis parameter                                       */
  const double *A;        /* [n] basin area                                               */
  const double *bs;       /* [n] -b_s/f^2  (Equi_Column.bs)                               */
  const double *bb;       /* [n] -b_bot/f^2 (Equi_Column.b_bot) with PM_EQ_HAS_BBOT, else B_int */
  const double *kappa;    /* [n] scalar diffusivity (ignored with PM_EQ_KAPPA_ARRAY)      */
  const int32_t *flags;   /* [n] PM_EQ_* */
  const double *zg;       /* [nzg] the model's z grid for array profiles (NULL if unused) */
  const double *kappa_z;  /* [n][nzg] kappa on zg                                         */
  const double *dkappa_z; /* [n][nzg] np.gradient(kappa, zg)                              */
  const double *psi_z;    /* [n][nzg] psi_so on zg                                        */
  double tol;             /* solve_bvp's tol = bc_tol (the reference uses the default 1e-3) */
  double *rms;            /* [n][mmax] out: rms residual of every interval (may be NULL)  */
  int32_t *nadd;          /* [n] out: nodes solve_bvp would insert (may be NULL)          */
  int32_t *status;        /* [n] out: 2 = singular Jacobian, else 0 (may be NULL)         */
  int32_t *niter;         /* [n] out: Newton iterations taken (may be NULL)               */
  double *info;           /* [n][2] out: max rms residual, max |bc residual| (may be NULL) */
  double *scratch;        /* [n][pm_equi_column_scratch_doubles(mmax)] workspace          */
} pm_equi_column;

size_t pm_equi_column_scratch_doubles(int32_t mmax);
int pm_equi_column_newton(const pm_equi_column *eq, pm_stream_t stream);

/* out[i] = alpha*x[i] + beta*y[i] (two products, one sum, in that order): the relaxation
 * b1 <- 0.8*b1 + 0.2*basin.b of examples/example_iteration.py:67.                         */
int pm_axpby(size_t count, double alpha, const double *x, double beta, const double *y,
             double *out, pm_stream_t stream);

/* ------------------------------------------------------------------ RCCL
 * One process per GPU.  The ensemble is sharded by member, stepping needs no
 * communication; the only exchange is the gather of per-member output at diagnostic
 * time (the reference has no distributed path: SURVEY.md section 5/8e).  librccl.so is
 * dlopen()ed on first use, so single-GPU runs never load it.
 * Bootstrap: rank 0 calls pm_comm_unique_id and ships the 128 bytes to the other ranks
 * by any host channel; then every rank calls pm_comm_init.                          */
#define PM_COMM_ID_BYTES 128
int pm_comm_unique_id(void *id128);
int pm_comm_init(pm_comm_t *comm, int32_t nranks, int32_t rank, const void *id128);
int pm_comm_destroy(pm_comm_t comm);
/* recv[nranks][count] <- send[count] of every rank (fp64, device pointers) */
int pm_comm_allgather(pm_comm_t comm, const void *send, void *recv, size_t count,
                      pm_stream_t stream);
/* gather to ONE rank: recv[nranks][count] on `root` <- send[count] of every rank (point-to-point
 * ncclSend / ncclRecv in one group; the root's own block is a device copy).  `recv` is only
 * read on the root and may be NULL elsewhere.  The diagnostic exchange needs no more than this:
 * one process writes the output (the reference's script is that process).  With nranks = 1 the
 * root sends to itself through RCCL when `self_loop` != 0 (plumbing rehearsal), else copies.   */
int pm_comm_gather_root(pm_comm_t comm, const void *send, void *recv, size_t count, int32_t root,
                        int32_t self_loop, pm_stream_t stream);
/* elementwise max over ranks (fp64, device pointers; in place allowed) */
int pm_comm_allreduce_max(pm_comm_t comm, const void *send, void *recv, size_t count,
                          pm_stream_t stream);
/* all ranks have reached this point AND their `stream` work before it is done */
int pm_comm_barrier(pm_comm_t comm, pm_stream_t stream);

/* debug/test: the kernels' two divisions by a precomputed reciprocal (div_by_recip from RN(1/d),
 * div_by_recip2 from the double-double reciprocal; both must be correctly rounded) against IEEE `/` on
 * `4*per_thread*256*blocks` random operand pairs (exponents within +-emax, 3/8 of the
 * mantissas at the all-ones / all-zeros / half-way edges); bitwise mismatches counted. */
int pm_selftest_fastdiv(uint64_t seed, int32_t blocks, int32_t per_thread, int32_t emax,
                        uint64_t *tested, uint64_t *mismatches);

/* Host function (no device work): *proven = 1 when, for EVERY one of the n denominators, the
 * 3-instruction quotient  y = RN(1/d); q0 = RN(a y); r = fma(-d, q0, a); q = fma(r, y, q0)  is the
 * correctly rounded a / d for every numerator a (finite operands whose quotient and residual stay
 * normal).  Only numerators whose quotient lies within 3 * 2^-53 ulp of a rounding boundary could
 * fail; for a given d they are the <= ~12 solutions of a congruence on the integer mantissas,
 * which the function enumerates and runs through the very sequence (pymoc_hip.hip: div3_proof).
 * Zero, subnormal and non-finite denominators are "not proven".  `candidates` (may be NULL)
 * receives the number of numerators tested.  What a caller establishes before it sets
 * PM_COLS_DIV3_PROVEN.  Replaces nothing of the reference (which divides in NumPy); it licenses
 * a cheaper instruction sequence for column.py:235-247's three divisions.                       */
int pm_div3_proven(const double *d, int64_t n, int32_t *proven, int64_t *candidates);

/* The proof above takes y = RN(1/d); the kernels form y on the DEVICE, whose fp64 `/` is not
 * correctly rounded in every case (see pm_selftest_div3).  *ok = 1 when the device's 1.0 / d[i]
 * equals the host's IEEE quotient bit for bit for all n host-side denominators (one small launch;
 * the kernels' prologues evaluate the same expression).  A caller sets PM_COLS_DIV3_PROVEN only
 * when both pm_div3_proven and pm_recip_check hold for the batch's denominators.               */
int pm_recip_check(const double *d, int64_t n, int32_t *ok);

/* debug/test: the DEVICE's 3-instruction quotient against the HOST's IEEE `/` on the candidate
 * numerators of `ndenoms` random denominators (uniform mantissas, mantissas next to 1 and 2,
 * mantissas with trailing zeros; numerators of both signs, rescaled) plus two arbitrary numerators
 * each: `mismatches` must be 0.  `unproven` = denominators the host proof rejected (their
 * candidates are tested all the same).  `device_div_off` = pairs on which the device's own
 * `a / d` differs from the host's quotient: NOT zero on gfx950 -- about 1e-4 of these
 * near-midpoint quotients come out one ulp off (its final correction does not use a correctly
 * rounded reciprocal); for an arbitrary operand pair that is a ~1e-19 event.                    */
int pm_selftest_div3(uint64_t seed, int32_t ndenoms, uint64_t *tested, uint64_t *mismatches,
                     uint64_t *unproven, uint64_t *device_div_off,
                     double *one_bad_pair /* {a, d} of a mismatch, or NULL */);

/* debug/test: the wave scans of the GM boundary-value solve (DPP row steps + row joins: element
 * prefix / suffix scans, the affine suffix scan, the node-count prefix sum) against the same
 * compositions done serially, for `nhas` (1..64) occupied lanes: max_rel3 = largest relative
 * deviations {prefix, suffix, affine} (association differs: ~1e-14), sum_mismatches = lanes whose
 * integer prefix sum differs (must be 0). */
int pm_selftest_so_scans(int32_t nhas, uint64_t seed, double *max_rel3, int32_t *sum_mismatches);

/* debug/test: lane-shift primitive self check (DPP wave shifts vs ds_bpermute) */
int pm_selftest_lane_shift(int32_t *mismatches);

#ifdef __cplusplus
}
#endif
#endif /* PYMOC_HIP_H */
