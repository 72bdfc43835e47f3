// ROUND 6: moved out of the product (gf2bv_amd/csrc/gf2_kernels.hip.h) together with its host side (three_level_host.inc).
// The three-level elimination was built and measured in round 5, bit-exact and SLOWER than the two-level default at every size
// measured (262144^2: 1.24-1.52 s against 1.05-1.07 s; profiles/r05_three_level.txt, r05_strassen.txt), so it no longer ships in
// libgf2bv_hip.so.  What stays usable: tools/microbench_strassen.hip, which includes this header after the product's kernel header.
#pragma once
// ==========================================================================================
// THREE-LEVEL ELIMINATION (round 5): super-panels, the Schur update as ONE GF(2) matrix product
// ==========================================================================================
// M4RI's _mzd_pluq (gf2bv/_internal.c:431-433) is block-recursive: the trailing matrix takes a whole column half at a time as a
// matrix product (mzd_addmul: Strassen-Winograd over an M4RM base case).  The two-level elimination above sends every outer
// panel of K <= 12 blocks through the whole trailing matrix: rows make one HBM trip per K blocks, each trip pays its load / store
// phase, and nothing sub-cubic can be done with an inner dimension of 3072.  Here SP consecutive outer panels form a SUPER-PANEL
// (D = SP x K x 256 columns, ~30000): inside it the two-level elimination runs unchanged but confined to the super-panel's own
// column tiles; right of it
//   (1) the super-panel's D pivot rows are brought up to date panel by panel (the REPLAY: k_outer_apply + k_update16k restricted
//       to rows that die later in the same super-panel -- the triangular solve U12 = L11^-1 A12, D^2/2 x columns of work),
//   (2) their segments are gathered into a compact B (k_gather_b), the multipliers of rows that died inside the super-panel are
//       cleared (k_zero_dead_mults: what is left is exactly L21),
//   (3) ONE product C ^= A x B updates every alive row: A = the per-row multipliers of the super-panel's blocks as the panel path
//       stored them, C = the matrix tiles right of the super-panel.  Its base case k_mul16k is the outer pass's table code with the
//       row segments held in registers through ALL blocks of the product (5.8 TB/s of sweep-words isolated against 4.9 for an
//       outer pass of 12 blocks inside a solve), and above it the host runs Strassen-Winograd levels (tools/microbench_strassen.hip,
//       profiles/r05_strassen.txt: 0.87 / 0.82 of the classical time with one / two levels at an inner dimension of 32768).
// Operand views (uint4 = 16-byte units): C tile t, row r at p[t * ts + r]; A block k, row r at p[k * bs + 2 r .. + 1] (stored
// multiplier form, midx / mult_stored: quadrant splits keep row offsets at multiples of 64); B block k, tile t, pivot i
// (= 64 panel + pivot bit) at p[k * bs + 256 t + i].
struct MulC { uint4 *p; i64 ts; };
struct MulA { const uint4 *p; i64 bs; };
struct MulB { const uint4 *p; i64 bs; };

// C (R rows x ntiles) ^= A x B over nb blocks (ZERO: C = A x B, C is not read).  One item = (tile, chunk of SEG x 512 rows); rows
// past R in the last chunk are read (the buffers carry a chunk of slack) and never stored.
template <int SEG, bool ZERO>
__global__ void __launch_bounds__(512)
k_mul16k(MulC C, i64 R, int ntiles, MulA A, MulB B, int nb)
{
	constexpr int NT = 512, NW = 8;
	__shared__ __attribute__((aligned(256))) uint4 tab[2 * 256 * 16];      // 128 KiB at LDS address 0
	__shared__ uint4 stage[GF2_GMAX * 64];
	if ((unsigned)(size_t)tab != 0u) __builtin_trap();
	const int lane = threadIdx.x & 63;
	const unsigned ulane = (unsigned)lane;
	const int wvu = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	unsigned KC[6];
	{
		const int rq_lo = lane & 7, rq_hi = (lane >> 3) & 1;
#pragma unroll
		for (int v = 0; v < 6; v++) {
			unsigned k = 1u << 24;
#pragma unroll
			for (int b = 0; b < 3; b++) {
				const int s = 3 * v + b;
				if (s < 16) k |= (unsigned)(16 * (8 * ((s >> 3) ^ rq_hi) + (((s & 7) + rq_lo) & 7))) << (8 * b);
			}
			asm volatile("" : "+v"(k));
			KC[v] = k;
		}
	}
	constexpr i64 CH = (i64)SEG * NT;
	const i64 nch = (R + CH - 1) / CH;
	const i64 items = nch * ntiles;
	for (i64 it = blockIdx.x; it < items; it += gridDim.x) {
		uint4 *Mw = C.p + (it / nch) * C.ts;
		const i64 rb0 = (it % nch) * CH + (i64)wvu * 64;
		uint4 *Mrow = Mw + rb0;
		uint4 d[SEG];
#pragma unroll
		for (int j = 0; j < SEG; j++) d[j] = ZERO ? make_uint4(0, 0, 0, 0) : (Mrow + j * (NW * 64))[ulane];
		const uint4 *Bt = B.p + (it / nch) * 256;
		uint4 staged = make_uint4(0, 0, 0, 0);
		if (threadIdx.x < GF2_GMAX * 64) staged = Bt[threadIdx.x];
#pragma unroll 1
		for (int k = 0; k < nb; k++) {
			__syncthreads();
			if (threadIdx.x < GF2_GMAX * 64) stage[threadIdx.x] = staged;
			__syncthreads();
			for (int e = threadIdx.x; e < 2 * 31 * 16; e += NT) {
				const int sub = e & 15, q = (e >> 4) % 31, grp = (e >> 4) / 31;
				const int idx = q <= 15 ? q : (q - 15) << 4;
				const uint4 *st = stage + (2 * grp + (sub >> 3)) * 64 + 8 * (sub & 7);
				uint4 acc = make_uint4(0, 0, 0, 0);
				int bits = idx;
				while (bits) { const int l = __ffs(bits) - 1; bits &= bits - 1; acc = xor4(acc, st[l]); }
				tab[grp * 4096 + idx * 16 + sub] = acc;
			}
			__syncthreads();
			for (int e = threadIdx.x; e < 2 * 225 * 16; e += NT) {
				const int sub = e & 15, q = (e >> 4) % 225, grp = (e >> 4) / 225;
				const int lo = 1 + q % 15, hi = (1 + q / 15) << 4;
				uint4 *tb = tab + grp * 4096 + sub;
				tb[(lo | hi) * 16] = xor4(tb[lo * 16], tb[hi * 16]);
			}
			if (k + 1 < nb && threadIdx.x < GF2_GMAX * 64) staged = Bt[(i64)(k + 1) * B.bs + threadIdx.x];
			__syncthreads();
			uint4 m0[2], m1[2];
			const uint4 *mrow = A.p + (i64)k * A.bs + rb0 * 2;
			auto loadm = [&](int j, int slot) {
				const uint4 *mr = mrow + (j < SEG ? j : SEG - 1) * (NW * 64 * 2);
				m0[slot] = mr[2 * ulane]; m1[slot] = mr[2 * ulane + 1];
			};
			auto issue = [&](u32x4 *v, const uint4 &a0, const uint4 &a1, int r) {
				const unsigned mw[8] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w };
				const int grp = r >> 1, hf = r & 1;
#pragma unroll
				for (int q = 0; q < 8; q++) {
					const int s = 8 * hf + q;
					const unsigned sel = (unsigned)(s % 3) | ((4u + (unsigned)(q & 3)) << 8) | ((grp ? 3u : 12u) << 16) | (12u << 24);
					const unsigned at = __builtin_amdgcn_perm(mw[2 * (2 * grp + hf) + (q >> 2)], KC[s / 3], sel);
					v[q] = *(lds_u4_ptr)(size_t)at;
				}
			};
			auto fold = [&](uint4 &acc, const u32x4 *v) {
#pragma unroll
				for (int h = 0; h < 4; h++) {
					acc.x = __builtin_amdgcn_bitop3_b32(acc.x, v[2 * h].x, v[2 * h + 1].x, 0x96);
					acc.y = __builtin_amdgcn_bitop3_b32(acc.y, v[2 * h].y, v[2 * h + 1].y, 0x96);
					acc.z = __builtin_amdgcn_bitop3_b32(acc.z, v[2 * h].z, v[2 * h + 1].z, 0x96);
					acc.w = __builtin_amdgcn_bitop3_b32(acc.w, v[2 * h].w, v[2 * h + 1].w, 0x96);
				}
			};
			u32x4 va[8], vb[8];
			loadm(0, 0); loadm(1, 1);
			issue(va, m0[0], m1[0], 0);
#pragma unroll
			for (int j = 0; j < SEG; j++) {
				const int c = j & 1;
				issue(vb, m0[c], m1[c], 1); fold(d[j], va);
				issue(va, m0[c], m1[c], 2); fold(d[j], vb);
				issue(vb, m0[c], m1[c], 3); fold(d[j], va);
				const uint4 n0 = m0[c ^ 1], n1 = m1[c ^ 1];
				loadm(j + 2, c);
				issue(va, n0, n1, 0); fold(d[j], vb);
			}
		}
#pragma unroll
		for (int j = 0; j < SEG; j++)
			if (rb0 + j * (NW * 64) + lane < R) (Mrow + j * (NW * 64))[ulane] = d[j];
	}
}

// X = Y ^ Z (^ W): the additions of the Strassen-Winograd levels on any operand type, element i of outer slice o at
// p[o * stride + i] -- ONE 16-byte element per thread with the non-temporal hint, the fastest stream form of this chip
__global__ void __launch_bounds__(256)
k_xor16(uint4 *X, i64 xs, const uint4 *Y, i64 ys, const uint4 *Z, i64 zs, const uint4 *W, i64 ws, i64 inner)
{
	const i64 i = (i64)blockIdx.x * 256 + threadIdx.x, o = blockIdx.y;
	if (i >= inner) return;
	u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(Y + o * ys + i));
	const u32x4 b = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(Z + o * zs + i));
	a ^= b;
	if (W) a ^= __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(W + o * ws + i));
	__builtin_nontemporal_store(a, reinterpret_cast<u32x4 *>(X + o * xs + i));
}
__global__ void __launch_bounds__(256)
k_zero16(uint4 *X, i64 xs, i64 inner)
{
	const i64 i = (i64)blockIdx.x * 256 + threadIdx.x, o = blockIdx.y;
	if (i < inner) X[o * xs + i] = make_uint4(0, 0, 0, 0);
}

// B of a super-panel: the final pivot rows of its blocks on the tiles [tile_begin, tile_begin + gridDim.x), compact.  lists: the row
// lists of its outer panels (k_outer_prow), panel pi of the super-panel in list slot (slot0 + pi) % nlist; a block without pivots
// (or a pivot bit without pivot) gives zeros.  grid (tiles, blocks).
__global__ void __launch_bounds__(256)
k_gather_b(const u64 *__restrict__ M, i64 srows, int tile_begin, const int *__restrict__ lists, int slot0, int nlist, int K,
           uint4 *__restrict__ Bc, i64 bs)
{
	const int k = blockIdx.y, pi = k / K, kl = k % K;
	const int *gprow = lists + (size_t)((slot0 + pi) % nlist) * GF2_OUTER_LISTS;
	const int pr = gprow[kl * 256 + threadIdx.x];
	const uint4 *Mw = reinterpret_cast<const uint4 *>(M) + ((i64)tile_begin + blockIdx.x) * srows;
	Bc[(i64)k * bs + (i64)blockIdx.x * 256 + threadIdx.x] = pr >= 0 ? Mw[pr] : make_uint4(0, 0, 0, 0);
}

// Rows that became pivot sources inside the super-panel (j_begin <= died < j_end) recorded multipliers for the blocks before their
// own while they were alive: the replay has used them, the product must not (their rows are final): cleared in all nb sets.
__global__ void __launch_bounds__(256)
k_zero_dead_mults(const int *__restrict__ died, i64 rows, i64 row_begin, int j_begin, int j_end, uint4 *__restrict__ mult, i64 bs, int nb)
{
	const i64 r = row_begin + (i64)blockIdx.x * 256 + threadIdx.x;
	if (r >= rows) return;
	const int d = died[r];
	if (d < j_begin || d >= j_end) return;
	for (int k = 0; k < nb; k++) {
		mult[(i64)k * bs + 2 * r] = make_uint4(0, 0, 0, 0);
		mult[(i64)k * bs + 2 * r + 1] = make_uint4(0, 0, 0, 0);
	}
}

