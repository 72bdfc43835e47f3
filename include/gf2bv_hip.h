/*
 * gf2bv_hip.h -- C ABI of libgf2bv_hip.so, the MI355X (gfx950) GF(2) solver that sits where
 * gf2bv's `_internal.c` calls M4RI.
 *
 * Plain C: pointers, sizes, opaque handles.  No torch / HIP types in any signature (a HIP
 * stream or device pointer crosses as `void*`).  Reference interface each entry point
 * replaces is cited as file:line relative to maple3142/gf2bv.
 *
 * Matrix layout everywhere ("augmented words"): row-major uint64, `stride` words per row,
 * column c of A = bit (c % 64) of word (c / 64), the right-hand side b is column `cols`
 * (so a row needs ceil((cols+1)/64) words).  This is M4RI's mzd_t bit order
 * (_internal.c:398-426 fills it with mzd_write_bit) and int.to_bytes(...,'little').
 *
 * Solution vectors: ceil(cols/64) words, bit j = variable j (_internal.c:32-39).
 *
 * Threading: every call owns its device buffers and stream; no global mutable state except
 * the last-error string, which is thread-local, and a thread-safe pool that recycles idle
 * device buffers, streams and events between calls (small buffers up to 512 MiB in total,
 * plus up to six large working buffers per device, at most a sixth of the device's memory and
 * never more than 48 GiB; GF2BV_KEEP_BIG=0 in the environment keeps none; gf2bv_pool_trim()
 * returns all of them to the device, and an allocation the device refuses -- the library's own or
 * gf2bv_device_alloc -- frees them and is repeated once).  Matches the reference releasing the GIL around the
 * whole factor/solve/kernel section (_internal.c:429-492).
 */
#ifndef GF2BV_HIP_H
#define GF2BV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GF2BV_OK            0
#define GF2BV_ERR_ARG       1   /* bad argument (the binding maps these to ValueError/TypeError) */
#define GF2BV_ERR_NODEVICE  2   /* no usable gfx950 device: the product path fails loudly, no CPU fallback */
#define GF2BV_ERR_HIP       3   /* a HIP runtime call failed, see gf2bv_last_error() */
#define GF2BV_ERR_NOMEM     4

#define GF2BV_MODE_SINGLE        0   /* _internal.h:25 SOLVE_MODE_SINGLE       */
#define GF2BV_MODE_AFFINE_SPACE  1   /* _internal.h:26 SOLVE_MODE_AFFINE_SPACE */

#define GF2BV_STATUS_SOLVED        0
#define GF2BV_STATUS_INCONSISTENT  1   /* _internal.c:440-446 -> Python None */

typedef struct gf2bv_result gf2bv_result;     /* owns origin / basis / pivots of one solve */

typedef struct gf2bv_stats {
	int64_t rows, cols, stride_words;
	int64_t rank, dimension;
	int32_t status;
	int32_t n_panels;          /* 64-column panels processed                                  */
	int32_t n_sweeps;          /* bulk-update passes over the trailing matrix that did work    */
	int32_t panels_per_sweep;  /* G: 64-column panels fused into one pass                      */
	int32_t tables_per_sweep;  /* G*T: grease tables applied per row per pass                  */
	int32_t table_bits;        /* k: index bits of the widest table                            */
	int32_t tile_words;        /* 64-bit words of one row segment handled by a lane group      */
	int32_t gang_systems;      /* systems that shared this solve's kernel launches (1: a solve of its own)    */
	double  sweep_words;       /* sum over sweeps of rows_swept x active_words  (unit of work) */
	double  row_xors;          /* sum over sweeps of rows_swept x T                            */
	float   ms_pack;           /* digits/words -> device matrix (H2D + pack kernel)            */
	float   ms_eliminate;      /* forward elimination, all panels (HIP events)                 */
	float   ms_sweep;          /* time inside the bulk-update kernels (HIP events): the sum over passes; launches that run side by side
	                            * (the two halves of an outer pass of the two-level elimination, the inner updates beside them) count once */
	float   ms_backsub;        /* consistency check + back-substitution + kernel basis         */
	float   ms_export;         /* D2H of origin / basis                                        */
	float   ms_total;          /* host wall clock of the whole call                            */
	int32_t search_handovers;  /* panels whose first search unit stopped waiting for the rest of its launch and
	                              left publishing to the last arriver (co-residency is not assumed)           */
	int32_t fast_blocks;       /* blocks of 4 panels factorised by the one-launch dense block search (k_block_fast)     */
	/* round 3 (two-level elimination): sweep_words above counts one word per word and BLOCK applied (the unit of the
	 * roofline, whatever kernel applied it); an outer pass applies K blocks per trip through HBM, so what the bulk kernels
	 * actually read + wrote is less: */
	double  hbm_words;         /* 64-bit words the bulk-update launches moved (each read once and written once)       */
	int32_t bulk_launches;     /* k_update16 + k_update16k launches that had pivots to apply                          */
	int32_t outer_blocks;      /* blocks applied through outer passes (0: one-level schedule)                         */
	int32_t handover_retries;  /* 1: a stream hand-over gate expired on the device and this is the result of the SECOND attempt,
	                              made with events (the device stays on events for this process); 0 normally        */
	int32_t small_path;        /* 1: solved by the one-launch kernel for systems that fit the LDS of one workgroup (k_small_solve; the
	                              phase times and sweep counters above are then 0 except ms_total); GF2BV_SMALL=0 disables that path */
} gf2bv_stats;

/* ---- library / device ------------------------------------------------------------------ */
int         gf2bv_version(void);
const char *gf2bv_build_id(void);              /* content hash of the HIP sources this binary was compiled from (build.py) */
int         gf2bv_device_count(void);          /* 0 when no HIP device is visible */
const char *gf2bv_last_error(void);            /* thread-local, never NULL */

/* ---- solve: replaces the body of m4ri_solve, gf2bv/_internal.c:398-489 -------------------- */

/* Equations as raw CPython `int` digit arrays (the binding memcpy's ob_digit, the device
 * pack kernel does what _internal.c:403-426 does bit by bit under the GIL):
 * row r occupies digits[digit_off[r] .. digit_off[r+1]), little-endian digits of
 * `bits_per_digit` payload bits (30 on 64-bit CPython) in uint32; sign already dropped;
 * bit 0 = affine term, bit k = coefficient of variable k-1; bits above `cols` ignored.
 * digit_off[0] must be 0 (offsets are relative to `digits`); GF2BV_ERR_ARG otherwise.
 * Requires rows >= cols > 0 (_internal.c:372-395). */
int gf2bv_solve_digits(const uint32_t *digits, const int64_t *digit_off, int bits_per_digit,
                       int64_t rows, int64_t cols, int mode, int device, gf2bv_result **out);

/* Same system, already packed as augmented words in HOST memory (not modified). */
int gf2bv_solve_words(const uint64_t *aug, int64_t rows, int64_t cols, int64_t stride_words,
                      int mode, int device, gf2bv_result **out);

/* Same, matrix resident in DEVICE memory (row-major augmented words, left untouched: the solver
 * first copies it into its own tile-major working layout, one extra pass over the matrix).
 * d_aug must be 16-byte aligned and stride_words even.  `stream` is a HIP stream handle or NULL: the solve is
 * enqueued on it (so it is ordered after whatever produced the matrix there) and the call returns when
 * the result is on the host.
 * `time_kernels` != 0 brackets every bulk-update launch with HIP events (fills ms_sweep). */
int gf2bv_solve_device(void *d_aug, int64_t rows, int64_t cols, int64_t stride_words,
                       int mode, int device, void *stream, int time_kernels, gf2bv_result **out);

/* Batch of `nsys` independent equal-shape systems resident on one device
 * (system s starts at d_aug + s*sys_stride_words words); out[0..nsys) receives handles.
 * The systems run in lock-step "gangs": one set of kernel launches eliminates a whole gang
 * (grid dimension y = system), so the latency-bound panel path is paid once per gang and the bulk
 * updates of all its systems fill the chip; results are identical to nsys separate calls.  In the
 * stats of a gang member ms_eliminate is the gang's elimination time, and ms_sweep (with
 * `time_kernels` != 0) the time inside the GANG's bulk-update launches (one launch serves all its
 * systems), not a per-system share.
 * Ordering: the gangs run on the library's own streams; they start after everything that was
 * enqueued on `stream` (a HIP stream handle, NULL = the null stream) when the call is made -- the stream
 * that produced the matrices -- and the call returns when all systems are solved.
 * Independent systems are the unit that bench.py shards across GPUs (BASELINE configs[3]). */
int gf2bv_solve_batch_device(void *d_aug, int64_t nsys, int64_t sys_stride_words,
                             int64_t rows, int64_t cols, int64_t stride_words,
                             int mode, int device, void *stream, int time_kernels, gf2bv_result **out);

/* Batch of `nsys` independent equal-shape systems given as digit arrays (see gf2bv_solve_digits):
 * row r of system s is entry s*rows + r of digit_off (nsys*rows + 1 entries).  One upload, lock-step
 * gangs as in gf2bv_solve_batch_device.  This is what a batched m4ri_solve binds
 * (gf2bv_amd._internal.m4ri_solve_many). */
int gf2bv_solve_batch_digits(const uint32_t *digits, const int64_t *digit_off, int bits_per_digit,
                             int64_t nsys, int64_t rows, int64_t cols, int mode, int device,
                             gf2bv_result **out);

/* The same batch over SEVERAL devices of the node, from one process and without any collective: entry k of
 * devices[0..ndevices) takes the k-th contiguous share of the systems (floor(nsys*k/ndevices) ..) on its own host thread,
 * uploads only its share of the digits and runs it as lock-step gangs there; out[] is in input order and identical to
 * what the one-device call returns.  A device may be listed more than once (its shares then run as concurrent gangs).
 * The reference solves one system per m4ri_solve call (gf2bv/_internal.c:359-502): independent systems -- one per
 * output bit / per instance in the recovery examples -- are the natural shard unit; this is what
 * m4ri_solve_many(..., devices) binds: devices=None = the module's default device (as m4ri_solve: a process pinned to one
 * GPU stays on it), devices="all" = every visible device, or an explicit list. */
int gf2bv_solve_batch_digits_multi(const uint32_t *digits, const int64_t *digit_off, int bits_per_digit,
                                   int64_t nsys, int64_t rows, int64_t cols, int mode,
                                   const int *devices, int ndevices, gf2bv_result **out);

/* ---- result accessors (AffineSpace getters, gf2bv/_internal.c:206-240) -------------------- */
int     gf2bv_result_status(const gf2bv_result *r);      /* GF2BV_STATUS_* */
int64_t gf2bv_result_rank(const gf2bv_result *r);
int64_t gf2bv_result_dimension(const gf2bv_result *r);   /* cols - rank (mode 1), basis rows */
int64_t gf2bv_result_words(const gf2bv_result *r);       /* ceil(cols/64) */
/* origin: the solution with every free variable 0 (_internal.c:438-455) */
int     gf2bv_result_origin(const gf2bv_result *r, uint64_t *out_words);
/* basis: dimension x words, M4RI kernel order (_internal.c:309-357, :475-489) */
int     gf2bv_result_basis(const gf2bv_result *r, uint64_t *out_words);
/* pivot columns c_0 < c_1 < ... (column rank profile), `rank` entries */
int     gf2bv_result_pivots(const gf2bv_result *r, int32_t *out);
int     gf2bv_result_stats(const gf2bv_result *r, gf2bv_stats *out);
void    gf2bv_result_free(gf2bv_result *r);

/* ---- AffineSpace arithmetic on host word vectors (tiny; gf2bv/_internal.c:242-273, :101-122) */
/* out = origin ^ XOR_{i: bit i of selector words set} basis[i]   (AffineSpace.get) */
void gf2bv_space_combine(const uint64_t *origin, const uint64_t *basis, int64_t dimension,
                         int64_t words, const uint64_t *selector, int64_t selector_words,
                         uint64_t *out);

/* ---- column-slab solve: ONE system over the GPUs of a node (replaces the single _mzd_pluq call of
 * gf2bv/_internal.c:431-433 for systems one GPU should not solve alone) --------------------------------------------
 * One process per GPU.  Rank r of `world` owns the column tiles t with t % world == r (gf2bv_slab_tiles(cols) tiles
 * of 8 words).  The caller owns the working matrix (d_work: gf2bv_slab_work_words(rows, cols) uint64, tile-major: tile
 * t is the contiguous slab d_work[t * slab_words .. (t+1) * slab_words), slab_words = work_words / tiles) and ALL the
 * communication:
 *     for b in range(gf2bv_slab_blocks(h)):
 *         if gf2bv_slab_owner(h, b) == rank: gf2bv_slab_factor(h, b, payload)    # panel path of the block, records out
 *         broadcast(payload, src = gf2bv_slab_owner(h, b))                       # ONE collective per block (RCCL over xGMI)
 *         gf2bv_slab_apply(h, b, payload)                                        # TRSM + bulk update of the own tiles
 *     gf2bv_slab_finish_local(h); gather every rank's tiles of d_work on rank 0; gf2bv_slab_solve(h, &result) there
 * payload: gf2bv_slab_payload_bytes(h) bytes of device memory (block records + rows x 32 bytes of multipliers).
 * Results are bit-identical to gf2bv_solve_device on the same matrix (the elimination is the same; only who applies
 * it to which columns differs).  d_aug: the full row-major system on every rank (left untouched). */
typedef struct gf2bv_slab gf2bv_slab;
int64_t gf2bv_slab_work_words(int64_t rows, int64_t cols);
int64_t gf2bv_slab_tiles(int64_t cols);
int     gf2bv_slab_open(void *d_aug, int64_t rows, int64_t cols, int64_t stride_words, void *d_work, int64_t work_words,
                        int world, int rank, int device, gf2bv_slab **out);
int64_t gf2bv_slab_blocks(const gf2bv_slab *h);
int     gf2bv_slab_owner(const gf2bv_slab *h, int block);
int64_t gf2bv_slab_payload_bytes(const gf2bv_slab *h);
int     gf2bv_slab_factor(gf2bv_slab *h, int block, void *d_payload);
int     gf2bv_slab_apply(gf2bv_slab *h, int block, const void *d_payload);
/* The same two steps ordered against the caller's collective stream instead of the host (`stream`: the HIP stream the
 * broadcast of d_payload is enqueued on).  The caller alternates TWO payload buffers: block b uses buffer b & 1 on every
 * rank.  _factor_on exports the records with one launch on the library's panel stream, makes `stream` wait for it and
 * returns at once: a broadcast enqueued on `stream` afterwards sends finished records, and the panel stream never waits
 * for the collective.  _apply_on imports what `stream` holds when it is called (non-owners), and orders `stream` behind the
 * import of the PREVIOUS block (whose buffer the next collective will write).  No call waits for the device: the broadcast
 * of block b + 1 overlaps the bulk update of block b on every rank.  At world size 1 nothing is exported at all.
 * (gf2bv_amd/slab.py runs this form.) */
int     gf2bv_slab_factor_on(gf2bv_slab *h, int block, void *d_payload, void *stream);
int     gf2bv_slab_apply_on(gf2bv_slab *h, int block, const void *d_payload, void *stream);
int     gf2bv_slab_finish_local(gf2bv_slab *h);
int     gf2bv_slab_solve(gf2bv_slab *h, gf2bv_result **out);
void    gf2bv_slab_close(gf2bv_slab *h);

/* ---- AffineSpace on the device: bulk enumeration (gf2bv/_internal.c:101-122 Gray walk, :63-91 binary walk) ---- */
/* The reference yields one element per call: one row XOR and one int export each.  Here a whole range of elements is
 * materialised by one kernel: element g of the walk = origin ^ XOR of basis[i] over the set bits i of code(g), with
 * code(g) = g ^ (g >> 1) for gray != 0 (AffineSpaceIterator, dimension <= 64) and code(g) = g otherwise
 * (AffineSpaceIteratorSlow, AffineSpace.get).  gf2bv_space_open uploads origin (words) and basis (dimension x words)
 * once; gf2bv_space_enumerate writes elements first .. first+count-1 (count x words uint64) to out_words (host
 * memory) -- or, with out_words == NULL, leaves them in the handle's own pinned buffer, valid until the next
 * enumerate / close (gf2bv_space_buffer): the iterators of the CPython binding slice their ints straight out of it. */
typedef struct gf2bv_space gf2bv_space;
int  gf2bv_space_open(const uint64_t *origin, const uint64_t *basis, int64_t dimension, int64_t words, int device,
                      gf2bv_space **out);
int  gf2bv_space_enumerate(gf2bv_space *space, uint64_t first, int64_t count, int gray, uint64_t *out_words);
const uint64_t *gf2bv_space_buffer(const gf2bv_space *space);
void gf2bv_space_close(gf2bv_space *space);

/* ---- synthetic systems + independent residual check (bench / tests) ----------------------- */
/* word w of row r = mix64(mix64(seed) ^ ((r<<20)|w)); planted solution = pseudo-row 0xFFFFF;
 * RHS = <row, planted>.  Writes rows x stride_words words at d_aug. */
/* (asynchronous: the generator kernel is enqueued on `stream` and the call returns) */
int gf2bv_synth_device(void *d_aug, int64_t rows, int64_t cols, int64_t stride_words,
                       uint64_t seed, int device, void *stream);
/* counts rows i with <A_i, x> != b_i on an (untouched) device matrix; x in host memory */
int gf2bv_residual_device(const void *d_aug, int64_t rows, int64_t cols, int64_t stride_words,
                          const uint64_t *x_words, int device, void *stream, int64_t *bad_rows);

/* Practical HBM ceilings of this device, measured with plain streaming kernels on a scratch buffer of
 * `bytes` bytes (use >= 1 GiB, well past the 256 MiB Infinity Cache): rmw_gbs = in-place 16-byte
 * read-XOR-write (the access pattern of the bulk update; read + written bytes per second),
 * read_gbs = read-only.  Reported by bench.py next to the 8 TB/s spec peak. */
int gf2bv_stream_ceiling_device(int device, int64_t bytes, double *rmw_gbs, double *read_gbs);

/* Shader clock of this device under an LDS-bound load (every CU issuing ds_read_b128 back to back for ~5 ms), from the
 * shader-clock and the 100 MHz real-time counters inside the kernel, and the LDS bytes per clock and CU that load reached
 * (MI355X_MICROARCH.md: 256 B/clk/CU for ds_read_b128).  bench.py prices the table lookups of the K-fused outer pass
 * (k_update16k, LDS-bound) against CUs x 256 B x this clock. */
int gf2bv_lds_clock_device(int device, double *shader_mhz, double *lds_bytes_per_clk_cu);

/* Registers per lane and static LDS bytes of the kernels that must fit on a compute unit TOGETHER -- the panel path of
 * block b+1 runs beside the bulk update of block b, and a panel kernel that does not fit next to an update workgroup
 * (two wavefronts per SIMD, 128 KiB of tables) silently waits for one to retire: out[0..1] the default bulk-update
 * instance, then k_block_fast, k_narrow_all, k_prio_window, k_panel_step (registers, LDS each; n >= 10).  A test holds the
 * budget: registers <= 512 - 2 x round_up(update's, 8), LDS <= 160 KiB - update's.  With n >= 13: out[10..12] = registers,
 * LDS and SCRATCH bytes per lane of k_update16k, the outer pass of the two-level elimination (it keeps 16 row segments per
 * lane in registers: scratch must be 0).  With n >= 15: out[13..14] = registers, LDS of k_block_fast_narrow.  With n >= 17:
 * out[15..16] = registers, LDS of k_block_sparse<256, 4> (the sparse block search, first pool size: beside the bulk update like
 * every panel kernel). */
int gf2bv_kernel_resources(int device, int32_t *out, int n);

/* plain device buffer helpers so a host language without a HIP binding can stage data.  gf2bv_device_alloc: when the device
 * is out of memory the pool's idle buffers are freed and the allocation is repeated once. */
int gf2bv_device_alloc(int device, int64_t bytes, void **d_ptr);
int gf2bv_device_free(int device, void *d_ptr);
int gf2bv_device_upload(int device, void *d_dst, const void *h_src, int64_t bytes);
int gf2bv_device_download(int device, void *h_dst, const void *d_src, int64_t bytes);


/* Page-locked host staging for bindings that assemble their input on the host (the CPython shim gathers every equation's digit
 * array -- gf2bv/_internal.c:403-426 walks them bit by bit instead -- into ONE buffer before gf2bv_solve_digits): the
 * host-to-device copy out of such a buffer is a single DMA, and the buffers are recycled between calls (up to four idle ones,
 * GF2BV_HOST_POOL_MB MiB in all, default 4608; gf2bv_host_pool_trim frees the idle ones and returns their bytes).  Any host
 * pointer remains valid input for every entry point; this is an optimisation, not a requirement. */
int  gf2bv_host_alloc(int64_t bytes, void **h_ptr);
void gf2bv_host_free(void *h_ptr);
int64_t gf2bv_host_pool_trim(void);

/* ---- the buffer pool (see "Threading" at the top) ----------------------------------------------- */
/* Frees every IDLE buffer the pool keeps on `device` (large working buffers and the small-buffer cache; nothing a running
 * solve holds).  Returns the bytes given back to the device, -1 without a usable device.  A caller that shares the GPU
 * (a tensor framework, another library) calls this between jobs; the library calls it itself when hipMalloc reports out of memory. */
int64_t gf2bv_pool_trim(int device);
/* Bytes of idle buffers the pool currently keeps on `device` (what gf2bv_pool_trim would free). */
int64_t gf2bv_pool_idle_bytes(int device);
/* The gang size gf2bv_solve_batch_* would choose for `nsys` systems of rows x cols with `free_bytes` of device memory free: a pure
 * function, no device is touched (bench.py --dry-run-ranks: every rank's plan of the multi-GPU batch job, testable without GPUs). */
int64_t gf2bv_plan_gang(int64_t nsys, int64_t rows, int64_t cols, int64_t free_bytes);

#ifdef __cplusplus
}
#endif
#endif /* GF2BV_HIP_H */
