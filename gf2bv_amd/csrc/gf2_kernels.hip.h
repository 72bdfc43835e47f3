// gf2_kernels.hip.h -- CDNA4 (gfx950) device code of the GF(2) solver.
//
// Replaces the M4RI routines gf2bv reaches from gf2bv/_internal.c (reference file:line):
//   mzd_write_bit loop           _internal.c:403-426   -> k_pack_digits
//   _mzd_pluq                    _internal.c:431-433   -> k_panel_scan / k_panel_select /
//                                                         k_pivot_apply / k_gather_mult / k_sweep
//   _mzd_pluq_solve_left         _internal.c:438-447   -> k_check_rhs + k_extract_y + k_sweep(above)
//   _mzd_kernel_left_pluq        _internal.c:309-357   -> same back-substitution with the free
//                                                         columns as extra right-hand sides
//   mzd_transpose / export       _internal.c:450,486   -> k_scatter_solution
//
// Everything is XOR / AND / shift / ctz / popcount on 64-bit words: HBM-bound integer work,
// no MFMA.  Wavefront = 64 lanes; a 64-column panel is one matrix word, so "one pivot bit
// per lane" and "ballot over 64 candidate rows" both map 1:1 onto a wavefront.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;   // matches HIP's 64-bit atomics and __ballot
typedef long long i64;

// One record per 64-column panel, written by k_panel_select.
struct PanelRec {
	int start;      // first pivot row of this panel (= rank before the panel)
	int p;          // pivots found in this panel (0..64)
	u64 mask;       // pivot bits inside the panel word (bit b <-> column 64*j+b)
};

// Per-solve device state.  Lives in device memory; the host never reads it mid-solve, so a
// whole elimination is one uninterrupted stream of launches.
struct SolveState {
	int rank;            // pivots found so far
	int inconsistent;    // set by k_check_rhs
	int nd;              // rows displaced out of [start, start+p) by the current panel
	int pad;
	int slot_row[64];    // row that supplied basis slot s (discovery order)
	u64 comb[64];        // comb[k]: slots XORed together to form sorted pivot row k
	int disp_from[64];
	int disp_to[64];
};

__device__ __forceinline__ u64 readlane64(u64 v, int l)
{
	unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l);
	unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
	return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ int ctz64(u64 v) { return __ffsll((long long)v) - 1; }
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u64 lanemask_lt(int lane) { return lane ? (~0ull >> (64 - lane)) : 0ull; }

// ------------------------------------------------------------------------------------------
// Matrix assembly: CPython digits -> augmented words (replaces _internal.c:403-426).
// One thread per output word.  Row r's int occupies digits[off[r]..off[r+1]); bit 0 is the
// affine term (-> column `cols`), bit k the coefficient of variable k-1 (-> column k-1).
__global__ void k_pack_digits(const uint32_t *__restrict__ digits, const i64 *__restrict__ off,
                              int bpd, i64 rows, i64 cols, i64 stride, u64 *__restrict__ M)
{
	i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
	i64 r = g / stride, w = g % stride;
	if (r >= rows) return;
	const uint32_t *d = digits + off[r];
	i64 nd = off[r + 1] - off[r];
	u64 val = 0;
	i64 c0 = w * 64;                         // first column of this word
	if (c0 < cols) {
		i64 p0 = c0 + 1;                     // int bit of column c0
		i64 di = p0 / bpd;
		int sh = (int)(p0 % bpd);
		int filled = 0;
		while (filled < 64 && di < nd) {
			u64 piece = (u64)(d[di] >> sh);
			val |= piece << filled;
			filled += bpd - sh;
			sh = 0;
			di++;
		}
		if (cols - c0 < 64) val &= (1ull << (cols - c0)) - 1;   // bits above `cols` are ignored
	}
	if (w == (cols >> 6) && nd > 0) val |= (u64)(d[0] & 1u) << (cols & 63);
	M[r * stride + w] = val;
}

// Host words (tight stride) -> padded device layout happens with a 2D memcpy; nothing to do here.

// ------------------------------------------------------------------------------------------
// Panel factorisation, step A.  Each wavefront ("unit") scans its slice of the still-active
// rows [rank, rows), word j only, and keeps an echelon XOR-basis keyed by lowest set bit:
// lane b owns the basis vector whose leading (lowest) bit is b.  64 candidate words are
// tested per step with __ballot; a unit stops as soon as its basis is full.  Output: the
// rows that supplied each unit's basis vectors (<= 64 per unit).  The union over units spans
// the same space as all active rows, which is all step B needs.
__global__ void __launch_bounds__(256)
k_panel_scan(const u64 *__restrict__ M, i64 stride, i64 rows, int j, u64 colmask,
             const SolveState *__restrict__ st, int *__restrict__ cand_cnt,
             int *__restrict__ cand_rows, int units)
{
	const int lane = threadIdx.x & 63;
	const int u = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	if (u >= units) return;
	const i64 r0 = st->rank;
	const i64 n = rows - r0;
	if (n <= 0) { if (lane == 0) cand_cnt[u] = 0; return; }
	i64 per = (n + units - 1) / units;
	per = (per + 63) & ~(i64)63;
	const i64 lo = r0 + (i64)u * per;
	const i64 hi = (lo + per < rows) ? lo + per : rows;
	const int full = __popcll(colmask);

	u64 bw = 0;            // my basis vector (lane = leading bit), 0 if none yet
	int brow = -1;
	u64 have = 0;          // wave-uniform: leading bits present
	for (i64 base = lo; base < hi && __popcll(have) < full; base += 256) {
		// 4 chunks of 64 rows in flight (word j of consecutive rows is a strided gather)
		u64 wq[4]; i64 iq[4];
#pragma unroll
		for (int q = 0; q < 4; q++) {
			iq[q] = base + 64 * q + lane;
			wq[q] = (iq[q] < hi) ? (M[iq[q] * stride + j] & colmask) : 0ull;
		}
#pragma unroll
		for (int q = 0; q < 4; q++) {
			u64 w = wq[q];
			u64 hv = have;
			while (hv) {                               // reduce by the current basis, ascending
				int b = uniform(ctz64(hv)); hv &= hv - 1;
				u64 v = readlane64(bw, b);
				if ((w >> b) & 1) w ^= v;
			}
			while (true) {                             // insert survivors one at a time
				u64 m = __ballot(w != 0);
				if (!m) break;
				int L = uniform(ctz64(m));
				u64 v = readlane64(w, L);
				int row = __builtin_amdgcn_readlane((int)iq[q], L);
				int b = uniform(ctz64(v));
				if (lane == b) { bw = v; brow = row; }
				have |= 1ull << b;
				if ((w >> b) & 1) w ^= v;              // lane L itself becomes 0
			}
			if (__popcll(have) >= full) break;
		}
	}
	if ((have >> lane) & 1) cand_rows[u * 64 + __popcll(have & lanemask_lt(lane))] = brow;
	if (lane == 0) cand_cnt[u] = __popcll(have);
}

// Panel factorisation, step B (one wavefront).  Re-runs the same basis construction over the
// candidate rows, this time remembering for every basis vector which candidate rows were
// folded into it (a 64-bit mask over discovery "slots").  Then reduces the basis completely
// (every vector keeps exactly its own pivot bit among the pivot bits), so that afterwards a
// row's multiplier is simply `word & pivot_mask` -- no sequential dependency between the 64
// columns of a panel.  Publishes: pivot mask, pivot columns, the slot rows, the per-pivot
// combination masks, and the row moves that bring the pivot rows to [rank, rank+p).
__global__ void __launch_bounds__(64)
k_panel_select(const u64 *__restrict__ M, i64 stride, i64 rows, int j, u64 colmask,
               SolveState *__restrict__ st, PanelRec *__restrict__ panels,
               int *__restrict__ pivcol, const int *__restrict__ cand_cnt,
               const int *__restrict__ cand_rows, int units)
{
	__shared__ int occ[64];
	const int lane = threadIdx.x;
	const int r0 = st->rank;
	const int full = __popcll(colmask);
	u64 bw = 0, bc = 0;
	int srow = -1;
	u64 have = 0;
	int nslots = 0;
	if ((i64)r0 < rows) {
		for (int u = 0; u < units && nslots < full; u++) {
			const int cnt = cand_cnt[u];
			if (cnt == 0) continue;
			int i = (lane < cnt) ? cand_rows[u * 64 + lane] : -1;
			u64 w = (i >= 0) ? (M[(i64)i * stride + j] & colmask) : 0ull;
			u64 c = 0;
			u64 hv = have;
			while (hv) {
				int b = uniform(ctz64(hv)); hv &= hv - 1;
				u64 v = readlane64(bw, b), vc = readlane64(bc, b);
				if ((w >> b) & 1) { w ^= v; c ^= vc; }
			}
			while (nslots < full) {
				u64 m = __ballot(w != 0);
				if (!m) break;
				int L = uniform(ctz64(m));
				u64 v = readlane64(w, L);
				u64 vc = readlane64(c, L) | (1ull << nslots);
				int row = __builtin_amdgcn_readlane(i, L);
				int b = uniform(ctz64(v));
				if (lane == b) { bw = v; bc = vc; }
				if (lane == nslots) srow = row;
				have |= 1ull << b;
				nslots++;
				if ((w >> b) & 1) { w ^= v; c ^= vc; }
			}
		}
	}
	// full reduction, highest pivot bit first
	{
		u64 hv = have;
		while (hv) {
			int b = uniform(63 - __clzll((long long)hv)); hv &= ~(1ull << b);
			u64 v = readlane64(bw, b), vc = readlane64(bc, b);
			if (lane != b && ((have >> lane) & 1) && ((bw >> b) & 1)) { bw ^= v; bc ^= vc; }
		}
	}
	const int p = nslots;
	if ((have >> lane) & 1) {
		int k = __popcll(have & lanemask_lt(lane));
		st->comb[k] = bc;
		pivcol[r0 + k] = 64 * j + lane;
	}
	if (lane < p) st->slot_row[lane] = srow;
	// row moves: pivot rows go to [r0, r0+p); rows sitting there that are not sources move
	// into the holes left by sources that lived below r0+p.
	occ[lane] = 0;
	__syncthreads();
	if (lane < p && srow < r0 + p) occ[srow - r0] = 1;
	__syncthreads();
	{
		bool is_d = (lane < p) && !occ[lane];
		u64 md = __ballot(is_d);
		if (is_d) st->disp_from[__popcll(md & lanemask_lt(lane))] = r0 + lane;
		bool is_v = (lane < p) && (srow >= r0 + p);
		u64 mv = __ballot(is_v);
		if (is_v) st->disp_to[__popcll(mv & lanemask_lt(lane))] = srow;
		if (lane == 0) {
			st->nd = __popcll(md);
			st->rank = r0 + p;
			panels[j].start = r0;
			panels[j].p = p;
			panels[j].mask = have;
		}
	}
}

// Panel factorisation, step C.  One workgroup per 128-byte column tile: stages the <= 64
// source rows and the displaced rows of its tile in LDS, then writes (a) displaced rows into
// the vacated positions and (b) the fully reduced pivot rows into [start, start+p).  All
// reads of a tile precede all writes of that tile, and tiles are disjoint across workgroups.
template <int TW>
__global__ void __launch_bounds__(256)
k_pivot_apply(u64 *__restrict__ M, i64 stride, int j, int tile0,
              const SolveState *__restrict__ st, const PanelRec *__restrict__ panels)
{
	__shared__ u64 S[64 * TW];
	__shared__ u64 D[64 * TW];
	const int p = panels[j].p;
	if (p == 0) return;
	const int r0 = panels[j].start;
	const int nd = st->nd;
	const i64 w0 = (i64)(tile0 + blockIdx.x) * TW;
	for (int idx = threadIdx.x; idx < p * TW; idx += 256) {
		int s = idx / TW, w = idx % TW;
		S[idx] = M[(i64)st->slot_row[s] * stride + w0 + w];
	}
	for (int idx = threadIdx.x; idx < nd * TW; idx += 256) {
		int q = idx / TW, w = idx % TW;
		D[idx] = M[(i64)st->disp_from[q] * stride + w0 + w];
	}
	__syncthreads();
	for (int idx = threadIdx.x; idx < nd * TW; idx += 256) {
		int q = idx / TW, w = idx % TW;
		M[(i64)st->disp_to[q] * stride + w0 + w] = D[idx];
	}
	for (int idx = threadIdx.x; idx < p * TW; idx += 256) {
		int k = idx / TW, w = idx % TW;
		u64 c = st->comb[k], acc = 0;
		while (c) { int s = ctz64(c); c &= c - 1; acc ^= S[s * TW + w]; }
		M[(i64)(r0 + k) * stride + w0 + w] = acc;
	}
}

// Multiplier snapshot: mult[i] = word j of row i restricted to the panel's pivot bits.
// Taken before the sweep rewrites word j (the sweep of the tile that contains word j would
// otherwise race with the other tiles' reads).  This is the panel's column of "L".
//   above == 0 : rows [start+p, rows)   (forward elimination)
//   above == 1 : rows [0, start)        (back-substitution, multipliers read from U)
__global__ void __launch_bounds__(256)
k_gather_mult(const u64 *__restrict__ M, i64 stride, i64 rows, int j,
              const PanelRec *__restrict__ rec, int above, u64 *__restrict__ mult)
{
	const int p = rec->p;
	if (p == 0) return;
	const i64 lo = above ? 0 : (i64)rec->start + p;
	const i64 hi = above ? (i64)rec->start : rows;
	const u64 mask = rec->mask;
	for (i64 i = lo + (i64)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (i64)gridDim.x * blockDim.x)
		mult[i] = M[i * stride + j] & mask;
}

// ------------------------------------------------------------------------------------------
// The sweep: row[i][tile] ^= XOR_t table_t[ bits [K*t, K*t+K) of mult[i] ].
//
// Grease tables ("Method of the Four Russians") for one 64-pivot panel and one column tile
// of TW words live in LDS: T = ceil(64/K) tables of 2^K entries (the last one smaller), each
// entry TW*8 bytes.  Bits of the panel word that are not pivots index a zero row, so the
// table index is a plain bit-field of the multiplier -- no pext.
//
// Lane mapping: a row segment (TW words) is covered by LPR = TW/2 consecutive lanes, 16 bytes
// each, so one global_load_dwordx4 / ds_read_b128 / global_store_dwordx4 per lane per table;
// a wavefront covers 64/LPR rows.  Lanes of one row read consecutive 16-byte slots of the
// SAME table entry (conflict-free); HBM accesses are whole 128-byte (TW=16) segments.
// Rows whose multiplier is 0 are neither loaded nor stored (sparse systems).
template <int K, int TW>
struct SweepCfg {
	static constexpr int T = (64 + K - 1) / K;
	static constexpr int LASTBITS = 64 - K * (T - 1);
	static constexpr int ENTRIES = (T - 1) * (1 << K) + (1 << LASTBITS);
	static constexpr int LPR = TW / 2;                 // lanes per row segment (16 B per lane)
	static constexpr int LDS_BYTES = ENTRIES * TW * 8;
};

template <int K, int TW, int NT>
__global__ void __launch_bounds__(NT)
k_sweep(u64 *__restrict__ M, i64 stride, i64 rows_total, const PanelRec *__restrict__ rec,
        int above, const u64 *__restrict__ mult, int tile0, int ntiles, int rows_per_block)
{
	typedef SweepCfg<K, TW> C;
	extern __shared__ __attribute__((aligned(16))) uint4 tab[];
	const int p = rec->p;
	if (p == 0) return;
	const int start = rec->start;
	const u64 pmask = rec->mask;
	const i64 lo = above ? 0 : (i64)start + p;
	const i64 hi = above ? (i64)start : rows_total;
	const int ct = blockIdx.x % ntiles;
	const i64 rb = blockIdx.x / ntiles;
	const i64 rbeg = lo + rb * rows_per_block;
	if (rbeg >= hi) return;
	const i64 rend = (rbeg + rows_per_block < hi) ? rbeg + rows_per_block : hi;
	const i64 w0 = (i64)(tile0 + ct) * TW;
	const int lr = threadIdx.x % C::LPR;
	const int rr = threadIdx.x / C::LPR;
	constexpr int RPP = NT / C::LPR;                   // rows per pass of the whole workgroup

	// ---- build the tables for this (panel, tile) ----
	// stage 1: entries whose index has bits only in the low half or only in the high half of
	// the table's bit-field come straight from (L2-resident) pivot rows; stage 2: the rest is
	// low_part ^ high_part, one LDS pass.
	const uint4 *Mq = reinterpret_cast<const uint4 *>(M);
	for (int g = rr; g < C::ENTRIES; g += RPP) {
		int t = g >> K; if (t > C::T - 1) t = C::T - 1;
		const int idx = g - (t << K);
		const int kt = (t == C::T - 1) ? C::LASTBITS : K;
		const int kl = kt >> 1;
		const int lomask = (1 << kl) - 1;
		if ((idx & lomask) && (idx & ~lomask)) continue;
		uint4 acc = make_uint4(0, 0, 0, 0);
		int bits = idx;
		while (bits) {
			int l = __ffs(bits) - 1; bits &= bits - 1;
			int b = t * K + l;
			if ((pmask >> b) & 1) {
				i64 prow = (i64)start + __popcll(pmask & ((1ull << b) - 1));
				uint4 v = Mq[(prow * stride + w0) / 2 + lr];
				acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
			}
		}
		tab[g * C::LPR + lr] = acc;
	}
	__syncthreads();
	for (int g = rr; g < C::ENTRIES; g += RPP) {
		int t = g >> K; if (t > C::T - 1) t = C::T - 1;
		const int idx = g - (t << K);
		const int kt = (t == C::T - 1) ? C::LASTBITS : K;
		const int kl = kt >> 1;
		const int lomask = (1 << kl) - 1;
		if (!((idx & lomask) && (idx & ~lomask))) continue;
		const int base = g - idx;
		uint4 a = tab[(base + (idx & lomask)) * C::LPR + lr];
		uint4 b = tab[(base + (idx & ~lomask)) * C::LPR + lr];
		a.x ^= b.x; a.y ^= b.y; a.z ^= b.z; a.w ^= b.w;
		tab[g * C::LPR + lr] = a;
	}
	__syncthreads();

	// ---- stream the rows ----
	uint4 *Mw = reinterpret_cast<uint4 *>(M);
	constexpr int U = 4;
	for (i64 base = rbeg; base < rend; base += (i64)RPP * U) {
		u64 m[U];
		uint4 d[U];
		i64 q[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			i64 row = base + (i64)u * RPP + rr;
			m[u] = (row < rend) ? mult[row] : 0ull;
			q[u] = (row * stride + w0) / 2 + lr;
		}
#pragma unroll
		for (int u = 0; u < U; u++)
			if (m[u]) d[u] = Mw[q[u]];
#pragma unroll
		for (int u = 0; u < U; u++) {
			if (!m[u]) continue;
			uint4 acc = d[u];
#pragma unroll
			for (int t = 0; t < C::T; t++) {
				const int kt = (t == C::T - 1) ? C::LASTBITS : K;
				const unsigned idx = (unsigned)(m[u] >> (K * t)) & ((1u << kt) - 1);
				const uint4 v = tab[((t << K) + idx) * C::LPR + lr];
				acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
			}
			Mw[q[u]] = acc;
		}
	}
}

// ------------------------------------------------------------------------------------------
// After forward elimination rows >= rank are zero in A; the system is consistent iff their
// RHS bits are zero too (the check inside _mzd_pluq_solve_left, _internal.c:440).
__global__ void __launch_bounds__(256)
k_check_rhs(const u64 *__restrict__ M, i64 stride, i64 rows, i64 cols, SolveState *__restrict__ st)
{
	const i64 r = st->rank;
	int bad = 0;
	for (i64 i = r + (i64)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (i64)gridDim.x * blockDim.x)
		bad |= (int)((M[i * stride + (cols >> 6)] >> (cols & 63)) & 1);
	if (__ballot(bad) && (threadIdx.x & 63) == 0) st->inconsistent = 1;
}

// Y[k][t] = U[k][ycols[t]] for k < rank: the right-hand sides of the back-substitution (the
// RHS column, plus the free columns when a kernel basis is wanted).  One wavefront packs 64
// columns of one row with a single ballot.
__global__ void __launch_bounds__(256)
k_extract_y(const u64 *__restrict__ M, i64 stride, const SolveState *__restrict__ st,
            const int *__restrict__ ycols, int ny, u64 *__restrict__ Y, i64 ys)
{
	const int lane = threadIdx.x & 63;
	const i64 wave = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const int nyw = (ny + 63) >> 6;
	const i64 k = wave / nyw;
	const int tw = (int)(wave % nyw);
	if (k >= st->rank) return;
	const int t = tw * 64 + lane;
	int bit = 0;
	if (t < ny) { int c = ycols[t]; bit = (int)((M[k * stride + (c >> 6)] >> (c & 63)) & 1); }
	u64 w = __ballot(bit);
	if (lane == 0) Y[k * ys + tw] = w;
}

// out[t][pivcol[k]] = Y[k][t]: pivot-variable part of the origin (t = ny-1) and of every
// kernel vector (t < ny-1).  out is ny x cw words, zero-initialised.
__global__ void __launch_bounds__(256)
k_scatter_solution(const u64 *__restrict__ Y, i64 ys, const SolveState *__restrict__ st,
                   const int *__restrict__ pivcol, int ny, u64 *__restrict__ out, i64 cw)
{
	const int nyw = (ny + 63) >> 6;
	const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
	const i64 k = g / nyw;
	const int tw = (int)(g % nyw);
	if (k >= st->rank) return;
	u64 y = Y[k * ys + tw];
	const int c = pivcol[k];
	while (y) {
		int b = ctz64(y); y &= y - 1;
		i64 t = (i64)tw * 64 + b;
		if (t < ny) atomicOr(&out[t * cw + (c >> 6)], 1ull << (c & 63));
	}
}

// ------------------------------------------------------------------------------------------
// Synthetic dense systems + independent residual check (bench / tests).
__device__ __forceinline__ u64 mix64(u64 x)
{
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}

// one wavefront per row: pass 1 accumulates <row, planted>, pass 2 writes the row
__global__ void __launch_bounds__(256)
k_synth(u64 *__restrict__ M, i64 rows, i64 cols, i64 stride, u64 seed)
{
	const int lane = threadIdx.x & 63;
	const i64 r = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (r >= rows) return;
	const i64 cw = (cols + 63) >> 6;
	const u64 lastmask = (cols & 63) ? ((1ull << (cols & 63)) - 1) : ~0ull;
	seed = mix64(seed);                       // hashed seed: neighbouring seeds give unrelated matrices
	u64 par = 0;
	for (i64 w = lane; w < cw; w += 64) {
		u64 a = mix64(seed ^ (((u64)r << 20) | (u64)w));
		u64 x = mix64(seed ^ ((0xFFFFFull << 20) | (u64)w));
		if (w == cw - 1) { a &= lastmask; x &= lastmask; }
		par ^= a & x;
	}
	const int rhs = __popcll(__ballot(__popcll(par) & 1)) & 1;
	for (i64 w = lane; w < stride; w += 64) {
		u64 a = 0;
		if (w < cw) {
			a = mix64(seed ^ (((u64)r << 20) | (u64)w));
			if (w == cw - 1) a &= lastmask;
		}
		if (w == (cols >> 6) && rhs) a |= 1ull << (cols & 63);
		M[r * stride + w] = a;
	}
}

// one wavefront per row: bad += (<A_i, x> != b_i)
__global__ void __launch_bounds__(256)
k_residual(const u64 *__restrict__ M, i64 rows, i64 cols, i64 stride, const u64 *__restrict__ x,
           u64 *__restrict__ bad)
{
	const int lane = threadIdx.x & 63;
	const i64 r = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (r >= rows) return;
	const i64 cw = (cols + 63) >> 6;
	const u64 lastmask = (cols & 63) ? ((1ull << (cols & 63)) - 1) : ~0ull;
	u64 par = 0;
	for (i64 w = lane; w < cw; w += 64) {
		u64 a = M[r * stride + w];
		if (w == cw - 1) a &= lastmask;
		par ^= a & x[w];
	}
	const int lhs = __popcll(__ballot(__popcll(par) & 1)) & 1;
	const int rhs = (int)((M[r * stride + (cols >> 6)] >> (cols & 63)) & 1);
	if (lane == 0 && lhs != rhs) atomicAdd(bad, 1ull);
}
