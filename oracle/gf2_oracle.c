/*
 * gf2_oracle.c -- CPU restatement of gf2bv's solve path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing under gf2bv_amd/ (the product) may import, link or call this file.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and there
 * only as the checker / the timed CPU baseline -- never as the thing shipped.
 *
 * What it restates (reference = maple3142/gf2bv, file:line relative to /root/reference):
 *   gf2bv/_internal.c:398-426   matrix assembly: bit 0 of each equation -> B, bit k -> A[.,k-1]
 *                               (here: the caller hands a row-major 64-bit-word augmented
 *                               matrix, column c = word c/64 bit c%64, RHS in column `cols`)
 *   gf2bv/_internal.c:431-433   r = _mzd_pluq(A,P,Q,0)          -> rank + column rank profile
 *   gf2bv/_internal.c:438-455   _mzd_pluq_solve_left(...)      -> consistency + particular
 *                               solution with every free variable = 0
 *   gf2bv/_internal.c:309-357   _mzd_kernel_left_pluq          -> right-kernel basis, ordered
 *                               by the Q transposition sequence
 *   gf2bv/_internal.c:475-489   transpose(ker)                  -> basis rows
 *
 * The arithmetic itself lives in M4RI (third-party, release 20260122 pinned by the
 * reference's setup.py:14-17), which is neither vendored in /root/reference nor
 * installed in this image.  The restatement therefore follows M4RI's published PLUQ
 * contract:
 *   S1  pivot columns c_0<c_1<... = the column rank profile of A
 *   S2  b not in colspace(A)  => no solution
 *   S3  origin = the solution with all non-pivot variables 0
 *   S4  order=[0..cols); for i<r swap(order[i],order[c_i]); free=order[r:];
 *       basis[i] = kernel vector with x[free[i]]=1, other free vars 0
 * Results of S1-S4 are functions of the RREF of [A|b], which is unique, so any correct
 * elimination order gives identical bits.
 *
 * PARITY PIN STATUS: unique-solution cases are pinned by the reference's own
 * known-answer test examples/mt.py:21-22,38 (tests/test_oracle_golden.py).  For
 * rank-deficient systems (choice of origin, basis order) no test in the reference pins
 * M4RI's behaviour and M4RI could not be run here: "parity unpinned" at that boundary.
 *
 * Two independent eliminations are provided and cross-checked in tests:
 *   algo 0  textbook Gauss-Jordan, one column at a time (obviously correct, slow)
 *   algo 1  Method-of-Four-Russians (M4RM-style: 64-column panels, 8-bit grease tables,
 *           column-tiled so the tables stay in cache, OpenMP over rows) -- this is the
 *           "port" CPU baseline that bench.py times next to the GPU.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;

typedef struct {
	int64_t rows, cols, cw;   /* cw = ceil(cols/64) words per solution vector */
	int status;               /* 0 = solvable, 1 = inconsistent */
	int64_t rank, dim;
	int32_t *pivcols;         /* rank entries */
	u64 *origin;              /* cw words */
	u64 *basis;               /* dim x cw words (mode 1 only) */
	/* work counters of the last algo-1 run (for the CPU baseline) */
	double row_xors;          /* sum over sweeps of rows_swept * tables_applied */
	double sweep_words;       /* sum over sweeps of rows_swept * active_words */
} gf2o_result;

static inline int getbit(const u64 *row, int64_t c) { return (int)((row[c >> 6] >> (c & 63)) & 1); }
static inline void setbit(u64 *row, int64_t c) { row[c >> 6] |= (u64)1 << (c & 63); }

/* ---------------------------------------------------------------------------------- */
/* algo 0: plain Gauss-Jordan to RREF, leftmost-column pivoting (column rank profile). */
static int64_t rref_plain(u64 *M, int64_t rows, int64_t cols, int64_t stride, int32_t *pivcols)
{
	int64_t r = 0;
	int64_t wt = (cols + 1 + 63) / 64;
	for (int64_t c = 0; c < cols && r < rows; c++) {
		int64_t piv = -1;
		for (int64_t i = r; i < rows; i++)
			if (getbit(M + i * stride, c)) { piv = i; break; }
		if (piv < 0) continue;
		if (piv != r)
			for (int64_t w = 0; w < wt; w++) {
				u64 t = M[piv * stride + w]; M[piv * stride + w] = M[r * stride + w]; M[r * stride + w] = t;
			}
		const u64 *pr = M + r * stride;
		for (int64_t i = 0; i < rows; i++) {
			if (i == r) continue;
			u64 *ri = M + i * stride;
			if (!getbit(ri, c)) continue;
			for (int64_t w = c >> 6; w < wt; w++) ri[w] ^= pr[w];
		}
		pivcols[r++] = (int32_t)c;
	}
	return r;
}

/* ---------------------------------------------------------------------------------- */
/* algo 1: M4RM-style.  Panel = the (up to) 64 columns of one word.  Per panel:
 *   1. scan word j of the not-yet-pivot rows, keep an echelon XOR-basis keyed by lowest
 *      set bit, remember which row supplied each basis vector and which earlier
 *      suppliers were folded into it;
 *   2. fully reduce the basis (each vector keeps exactly its own pivot bit among pivot
 *      bits), turn the recorded combinations into full-width pivot rows, move them to
 *      rows [r, r+p) sorted by pivot column;
 *   3. every other row: m = row[j] & pivot_mask, row ^= XOR_t table_t[byte t of m]
 *      (Gauss-Jordan: rows above are swept too, so the final matrix is the RREF). */
#define TILE_WORDS 64

static void xor_words(u64 *__restrict dst, const u64 *__restrict src, int64_t n)
{
	for (int64_t i = 0; i < n; i++) dst[i] ^= src[i];
}

static int64_t rref_m4rm(u64 *M, int64_t rows, int64_t cols, int64_t stride, int32_t *pivcols,
                         double *row_xors, double *sweep_words)
{
	int64_t wt = (cols + 1 + 63) / 64;       /* words that carry data (incl. RHS column) */
	int64_t npanels = (cols + 63) / 64;
	int64_t r = 0;
	u64 *tmp = (u64 *)malloc((size_t)64 * wt * sizeof(u64));
	u64 *mult = (u64 *)malloc((size_t)rows * sizeof(u64));
	u64 *tab = (u64 *)malloc((size_t)((wt + TILE_WORDS - 1) / TILE_WORDS) * 8 * 256 * TILE_WORDS * sizeof(u64));
	double rx = 0, sw = 0;

	for (int64_t j = 0; j < npanels && r < rows; j++) {
		u64 colmask = (cols - 64 * j >= 64) ? ~(u64)0 : (((u64)1 << (cols - 64 * j)) - 1);
		u64 bw[64], bc[64];      /* basis vector / combination mask, indexed by leading bit */
		int64_t slot_row[64];
		u64 have = 0;
		int nslots = 0;
		int full = __builtin_popcountll(colmask);
		for (int64_t i = r; i < rows && nslots < full; i++) {
			u64 w = M[i * stride + j] & colmask, c = 0;
			while (w) {
				int b = __builtin_ctzll(w);
				if (!((have >> b) & 1)) break;
				w ^= bw[b]; c ^= bc[b];
			}
			if (!w) continue;
			int b = __builtin_ctzll(w);
			bw[b] = w; bc[b] = c | ((u64)1 << nslots);
			slot_row[nslots++] = i;
			have |= (u64)1 << b;
		}
		int p = nslots;
		if (!p) continue;
		/* full reduction, highest pivot bit first */
		for (int b = 63; b >= 0; b--) {
			if (!((have >> b) & 1)) continue;
			for (int a = 0; a < 64; a++)
				if (a != b && ((have >> a) & 1) && ((bw[a] >> b) & 1)) { bw[a] ^= bw[b]; bc[a] ^= bc[b]; }
		}
		/* build the p reduced pivot rows (trailing words only) in tmp, sorted by pivot bit */
		int k = 0;
		for (int b = 0; b < 64; b++) {
			if (!((have >> b) & 1)) continue;
			u64 *t = tmp + (size_t)k * wt;
			memset(t + j, 0, (size_t)(wt - j) * sizeof(u64));
			for (int s = 0; s < p; s++)
				if ((bc[b] >> s) & 1) xor_words(t + j, M + slot_row[s] * stride + j, wt - j);
			pivcols[r + k] = (int32_t)(64 * j + b);
			k++;
		}
		/* move displaced rows out of [r, r+p), then drop the pivot rows in */
		{
			char occ[64]; memset(occ, 0, sizeof occ);
			int64_t vac[64]; int nv = 0;
			for (int s = 0; s < p; s++) {
				if (slot_row[s] < r + p) occ[slot_row[s] - r] = 1; else vac[nv++] = slot_row[s];
			}
			int v = 0;
			for (int q = 0; q < p; q++)
				if (!occ[q]) { memcpy(M + vac[v] * stride, M + (r + q) * stride, (size_t)wt * sizeof(u64)); v++; }
			for (int q = 0; q < p; q++) {
				/* words < j of a row that was still active are all zero */
				memset(M + (r + q) * stride, 0, (size_t)j * sizeof(u64));
				memcpy(M + (r + q) * stride + j, tmp + (size_t)q * wt + j, (size_t)(wt - j) * sizeof(u64));
			}
		}
		/* multipliers (snapshot before the sweep rewrites word j) */
		for (int64_t i = 0; i < rows; i++)
			mult[i] = (i >= r && i < r + p) ? 0 : (M[i * stride + j] & have);
		/* sweep.  Tables: 8 groups x 256 entries over all active words, built tile by tile (TILE_WORDS
		 * words) so a thread works on one cache-sized piece; then ONE parallel loop over the rows, each
		 * row XORs its 8 table entries tile by tile.  Two parallel regions per panel. */
		int64_t aw = wt - j;                              /* active words */
		int64_t ntiles = (aw + TILE_WORDS - 1) / TILE_WORDS;
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(static)
#endif
		for (int64_t t = 0; t < ntiles; t++) {
			for (int g = 0; g < 8; g++) {
				int64_t w0 = j + t * TILE_WORDS;
				int64_t tw = (wt - w0 < TILE_WORDS) ? (wt - w0) : TILE_WORDS;
				u64 *T = tab + ((size_t)t * 8 + g) * 256 * TILE_WORDS;
				memset(T, 0, (size_t)tw * sizeof(u64));
				for (int l = 0; l < 8; l++) {
					int b = 8 * g + l;
					const u64 *src = NULL;
					if ((have >> b) & 1) {
						int kk = __builtin_popcountll(have & (((u64)1 << b) - 1));
						src = M + (r + kk) * stride + w0;
					}
					for (int idx = 0; idx < (1 << l); idx++) {
						u64 *dst = T + (size_t)(idx | (1 << l)) * TILE_WORDS;
						const u64 *lo = T + (size_t)idx * TILE_WORDS;
						if (src) for (int64_t w = 0; w < tw; w++) dst[w] = lo[w] ^ src[w];
						else     for (int64_t w = 0; w < tw; w++) dst[w] = lo[w];
					}
				}
			}
		}
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (rows * aw >= 32768)
#endif
		for (int64_t i = 0; i < rows; i++) {
			u64 m = mult[i];
			if (!m) continue;
			for (int64_t t = 0; t < ntiles; t++) {
				int64_t w0 = j + t * TILE_WORDS;
				int64_t tw = (wt - w0 < TILE_WORDS) ? (wt - w0) : TILE_WORDS;
				u64 *row = M + i * stride + w0;
				for (int g = 0; g < 8; g++) {
					unsigned idx = (unsigned)((m >> (8 * g)) & 255);
					if (idx) xor_words(row, tab + (((size_t)t * 8 + g) * 256 + idx) * TILE_WORDS, tw);
				}
			}
		}
		rx += (double)(rows - p) * 8.0;
		sw += (double)(rows - p) * (double)(wt - j);
		r += p;
	}
	free(tmp); free(mult); free(tab);
	if (row_xors) *row_xors = rx;
	if (sweep_words) *sweep_words = sw;
	return r;
}

/* ---------------------------------------------------------------------------------- */
/* From the RREF of [A|b]: S2 consistency, S3 origin, S4 ordered kernel basis.         */
static void finish(gf2o_result *R, const u64 *M, int64_t stride, int mode)
{
	int64_t rows = R->rows, cols = R->cols, cw = R->cw, r = R->rank;
	R->status = 0;
	for (int64_t i = r; i < rows; i++)
		if (getbit(M + i * stride, cols)) { R->status = 1; break; }
	R->dim = cols - r;
	R->origin = (u64 *)calloc((size_t)(cw ? cw : 1), sizeof(u64));
	R->basis = NULL;
	if (R->status) return;
	for (int64_t k = 0; k < r; k++)
		if (getbit(M + k * stride, cols)) setbit(R->origin, R->pivcols[k]);
	if (mode != 1) return;
	/* S4: replay the Q transpositions (gf2bv/_internal.c:348 mzd_apply_p_left_trans(R,Q)
	 * undoes M4RI's pivot-column compression, which swaps column i with column c_i for
	 * i = 0..r-1 in that order). */
	int64_t *order = (int64_t *)malloc((size_t)cols * sizeof(int64_t));
	for (int64_t i = 0; i < cols; i++) order[i] = i;
	for (int64_t i = 0; i < r; i++) {
		int64_t c = R->pivcols[i], t = order[i];
		order[i] = order[c]; order[c] = t;
	}
	R->basis = (u64 *)calloc((size_t)(R->dim ? R->dim : 1) * (size_t)(cw ? cw : 1), sizeof(u64));
	for (int64_t t = 0; t < R->dim; t++) {
		int64_t f = order[r + t];
		u64 *v = R->basis + t * cw;
		setbit(v, f);
		for (int64_t k = 0; k < r; k++)
			if (getbit(M + k * stride, f)) setbit(v, R->pivcols[k]);
	}
	free(order);
}

/* aug: rows x stride words, not modified.  algo 0 = plain, 1 = M4RM.  mode as in
 * gf2bv/_internal.h:25-26 (0 = single solution, 1 = affine space). */
gf2o_result *gf2o_solve(const u64 *aug, int64_t rows, int64_t cols, int64_t stride, int mode, int algo)
{
	int64_t wt = (cols + 1 + 63) / 64;
	if (rows < 0 || cols <= 0 || stride < wt) return NULL;
	gf2o_result *R = (gf2o_result *)calloc(1, sizeof *R);
	R->rows = rows; R->cols = cols; R->cw = (cols + 63) / 64;
	u64 *M = (u64 *)malloc((size_t)(rows ? rows : 1) * (size_t)wt * sizeof(u64));
	u64 tailmask = ((cols + 1) & 63) ? (((u64)1 << ((cols + 1) & 63)) - 1) : ~(u64)0;
	for (int64_t i = 0; i < rows; i++) {
		memcpy(M + i * wt, aug + i * stride, (size_t)wt * sizeof(u64));
		M[i * wt + wt - 1] &= tailmask;    /* bits above column `cols` are ignored (_internal.c:414) */
	}
	int64_t maxr = rows < cols ? rows : cols;
	R->pivcols = (int32_t *)malloc((size_t)(maxr ? maxr : 1) * sizeof(int32_t));
	if (algo == 0) R->rank = rref_plain(M, rows, cols, wt, R->pivcols);
	else           R->rank = rref_m4rm(M, rows, cols, wt, R->pivcols, &R->row_xors, &R->sweep_words);
	finish(R, M, wt, mode);
	free(M);
	return R;
}

int     gf2o_status(const gf2o_result *R) { return R->status; }
int64_t gf2o_rank(const gf2o_result *R) { return R->rank; }
int64_t gf2o_dim(const gf2o_result *R) { return R->dim; }
double  gf2o_row_xors(const gf2o_result *R) { return R->row_xors; }
double  gf2o_sweep_words(const gf2o_result *R) { return R->sweep_words; }
void gf2o_pivcols(const gf2o_result *R, int32_t *out) { memcpy(out, R->pivcols, (size_t)R->rank * sizeof(int32_t)); }
void gf2o_origin(const gf2o_result *R, u64 *out) { memcpy(out, R->origin, (size_t)R->cw * sizeof(u64)); }
void gf2o_basis(const gf2o_result *R, u64 *out)
{
	if (R->basis && R->dim) memcpy(out, R->basis, (size_t)R->dim * (size_t)R->cw * sizeof(u64));
}
void gf2o_free(gf2o_result *R)
{
	if (!R) return;
	free(R->pivcols); free(R->origin); free(R->basis); free(R);
}
int gf2o_max_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}
void gf2o_set_threads(int n)
{
#ifdef _OPENMP
	if (n > 0) omp_set_num_threads(n);
#else
	(void)n;
#endif
}

/* ---------------------------------------------------------------------------------- */
/* Synthetic dense systems (DESIGN.md "synthetic generator"):     */
/* word w of row r = mix64(mix64(seed) ^ ((r << 20) | w)); planted x* lives in          */
/* pseudo-row 0xFFFFF; RHS bit = <row, x*>.  Same definition as the HIP generator       */
/* kernel.  (The seed is hashed first: XOR-ing a raw small seed into the key would only  */
/* permute word columns, so neighbouring seeds would give column-permuted copies.)       */
static inline u64 mix64(u64 x)
{
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}
u64 gf2o_synth_word(u64 seed, int64_t r, int64_t w) { return mix64(mix64(seed) ^ (((u64)r << 20) | (u64)w)); }

void gf2o_gen_synthetic(u64 *aug, int64_t rows, int64_t cols, int64_t stride, u64 seed)
{
	int64_t cw = (cols + 63) / 64;
	u64 lastmask = (cols & 63) ? (((u64)1 << (cols & 63)) - 1) : ~(u64)0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
	for (int64_t r = 0; r < rows; r++) {
		u64 *row = aug + r * stride;
		u64 par = 0;
		memset(row, 0, (size_t)stride * sizeof(u64));
		for (int64_t w = 0; w < cw; w++) {
			u64 a = gf2o_synth_word(seed, r, w);
			u64 x = gf2o_synth_word(seed, 0xFFFFF, w);
			if (w == cw - 1) { a &= lastmask; x &= lastmask; }
			row[w] = a;
			par ^= a & x;
		}
		if (__builtin_popcountll(par) & 1) setbit(row, cols);
	}
}
void gf2o_planted_solution(u64 *x, int64_t cols, u64 seed)
{
	int64_t cw = (cols + 63) / 64;
	u64 lastmask = (cols & 63) ? (((u64)1 << (cols & 63)) - 1) : ~(u64)0;
	for (int64_t w = 0; w < cw; w++) {
		x[w] = gf2o_synth_word(seed, 0xFFFFF, w);
		if (w == cw - 1) x[w] &= lastmask;
	}
}

/* number of rows i with <A_i, x> != b_i on the ORIGINAL system (independent check) */
int64_t gf2o_check_solution(const u64 *aug, int64_t rows, int64_t cols, int64_t stride, const u64 *x)
{
	int64_t cw = (cols + 63) / 64, bad = 0;
	u64 lastmask = (cols & 63) ? (((u64)1 << (cols & 63)) - 1) : ~(u64)0;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : bad) schedule(static)
#endif
	for (int64_t i = 0; i < rows; i++) {
		const u64 *row = aug + i * stride;
		u64 par = 0;
		for (int64_t w = 0; w < cw; w++) {
			u64 a = row[w];
			if (w == cw - 1) a &= lastmask;
			par ^= a & x[w];
		}
		int lhs = __builtin_popcountll(par) & 1;
		if (lhs != getbit(row, cols)) bad++;
	}
	return bad;
}
