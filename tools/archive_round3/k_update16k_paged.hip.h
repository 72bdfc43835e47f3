// ARCHIVE (round 3, not built into the product): the outer-pass kernel with PAGE-PHASED table building -- the lookups of
// one 64 KiB page of the tables run for all 16 segments while the other page is rebuilt for its next use, so that no table
// build stands between the blocks.  Bit-exact (two-level stress tests + parity with GF2BV_TWO_LEVEL=8), but:
//   isolated (tools/microbench_update16k.hip, 131072 x 1024 tiles): K=4 4.45, K=8 5.30 TB/s-eq -- the plain kernel: 4.80 / 5.34;
//   255 VGPRs instead of 217: two of its wavefronts fill a SIMD and the inner path's kernels no longer fit beside it --
//   262144^2 1.65 s instead of 1.29 s.
// The experiment switches below gave the reason (profiles/r03_kloop.txt): with the multipliers taken from registers AND no
// table build at all the kernel reaches 6.28 TB/s-eq at K=8 -- the 32 ds_read_b128 per segment and block alone take 8.4 us of
// the ~10 us a block costs per item (4.9 clocks per wavefront instruction, ~80 % of the LDS rate): the kernel is bound by the
// LDS lookups, and hiding the build behind them only trades one LDS user for another.
// Drop-in for k_update16k of gf2_kernels.hip.h (same arguments); needs `stg` instead of `stage` in its shared declarations.
#ifndef GF2_K16K_EXP
#define GF2_K16K_EXP 0      // (tools/microbench_update16k.hip: 1 = multipliers from registers, 2 = no table build)
#endif
template <int V> struct IntC { static constexpr int value = V; };
template <int SEG>
__global__ void __launch_bounds__(512)
k_update16k(u64 *__restrict__ M, i64 rows, i64 srows, int nblk, const int *__restrict__ gprow,
            const u64 *__restrict__ mult, i64 set_words, int set0, int nsets,
            const int *__restrict__ blk_first, const int *__restrict__ died, int j_end, int tile_begin, int ntiles)
{
	constexpr int NT = 512, NW = 8;
	static_assert(GF2_KMAX * GF2_GMAX * 64 % 512 == 0, "k_outer_apply: whole pivots per lane");
	__shared__ __attribute__((aligned(256))) uint4 tab[2 * 256 * 16];      // 128 KiB, must sit at LDS address 0 (checked below)
	__shared__ uint4 stg[2 * 128];                                         // rows a page is built from: [page][panel in page][pivot bit]
	__shared__ int prow[GF2_KMAX * GF2_GMAX * 64];                        // [block][panel][pivot bit] -> physical row, -1 if none
	__shared__ int anyb[GF2_KMAX];
	if ((unsigned)(size_t)tab != 0u) __builtin_trap();
	const int lane = threadIdx.x & 63;
	const unsigned ulane = (unsigned)lane;
	const int wvu = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));       // wave-uniform by construction: keep it scalar
	for (int t = threadIdx.x; t < nblk * GF2_GMAX * 64; t += NT) prow[t] = gprow[t];
	if (threadIdx.x < GF2_KMAX) anyb[threadIdx.x] = gprow[GF2_KMAX * GF2_GMAX * 64 + threadIdx.x];
	__syncthreads();
	int first_blk = -1, last_blk = -1;
	for (int k = 0; k < nblk; k++) if (anyb[k]) { if (first_blk < 0) first_blk = k; last_blk = k; }
	if (first_blk < 0) return;
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
	const i64 rlo = (i64)(*blk_first) & ~(i64)63;      // rows below were pivots before this panel began
	const i64 R64 = (rows + 63) & ~(i64)63;
	constexpr i64 CH = (i64)SEG * NT;
	const i64 nch = (R64 - rlo + CH - 1) / CH;
	const i64 items = nch * ntiles;
	const uint4 *mbase = reinterpret_cast<const uint4 *>(mult);
	i64 it = blockIdx.x;
	if (it >= items) return;
	// Every address below = a wave-uniform 64-bit base (scalar registers) + a 32-BIT lane offset (batch j of a wavefront =
	// rows base + 512 j + lane): nothing per batch lives in a 64-bit VGPR pair -- with per-lane 64-bit row indices, clamped to
	// the row range, the compiler kept 16 address pairs per stream across the block loop and spilled up to 1700 registers.
	// Instead of clamping, batches past the padded row range (only in a tile's last chunk) simply run: they read up to
	// 8192 rows into the next tile / the next multiplier set (the solver leaves that much slack behind the matrix and
	// the multiplier sets), take garbage and are never stored -- a lane stores only if its row is alive.
	// tile-major items: neighbouring workgroups take neighbouring row chunks of ONE tile.
	auto item_rows = [&](i64 item) { return rlo + (item % nch) * CH + (i64)wvu * 64; };
	auto item_tile = [&](i64 item) { return reinterpret_cast<uint4 *>(M) + ((i64)tile_begin + item / nch) * srows; };
	for (; it < items; it += gridDim.x) {
		uint4 *Mw = item_tile(it);
		const i64 rb0 = item_rows(it);
		uint4 *Mrow = Mw + rb0;
		uint4 d[SEG];
#pragma unroll
		for (int j = 0; j < SEG; j++) d[j] = (Mrow + j * (NW * 64))[ulane];
		unsigned alive = 0;
#pragma unroll
		for (int j = 0; j < SEG; j++) {
			const i64 rl = rb0 + j * (NW * 64) + lane;
			const int dd = died[rl < rows ? rl : rows - 1];
			// alive BEHIND this panel: never a pivot source, or one of a later panel (the next panel may already be under way)
			if (rl < rows && dd >= j_end) alive |= 1u << j;
		}
		// Half-blocks: page p of the tables (64 KiB) holds the byte fields of panels 2p, 2p + 1 of a block and goes with 16 B
		// of the 32 B multipliers.  A step = the lookups of ONE page for all SEG segments, and next to them the other page is
		// rebuilt for its next use (page 1 of this block during the page-0 step, page 0 of the next block during the page-1
		// step): table building (~1/4 of a block's time when it ran between the blocks) disappears behind the lookups.
		// Blocks without pivots inside [first_blk, last_blk] (rank-deficient tails only) run with all-zero tables.
		if (threadIdx.x < 128) {
			const int p0 = prow[first_blk * 256 + threadIdx.x], p1 = prow[first_blk * 256 + 128 + threadIdx.x];
			const uint4 s0 = p0 >= 0 ? Mw[p0] : make_uint4(0, 0, 0, 0), s1 = p1 >= 0 ? Mw[p1] : make_uint4(0, 0, 0, 0);
			stg[threadIdx.x] = s0; stg[128 + threadIdx.x] = s1;
		}
		__syncthreads();
		auto build0 = [&](int page) {                   // entries with bits in one nibble only: one per thread
			const int e = threadIdx.x;
			if (e < 31 * 16) {
				const int sub = e & 15, q = e >> 4;
				const int idx = q <= 15 ? q : (q - 15) << 4;
				const uint4 *st = stg + page * 128 + (sub >> 3) * 64 + 8 * (sub & 7);
				uint4 acc = make_uint4(0, 0, 0, 0);
				int bits = idx;
				while (bits) { const int l = __ffs(bits) - 1; bits &= bits - 1; acc = xor4(acc, st[l]); }
				tab[page * 4096 + idx * 16 + sub] = acc;
			}
		};
		auto build1 = [&](int page) {                   // mixed = low-nibble entry ^ high-nibble entry
			for (int e = threadIdx.x; e < 225 * 16; e += NT) {
				const int sub = e & 15, q = e >> 4;
				const int lo = 1 + q % 15, hi = (1 + q / 15) << 4;
				uint4 *tb = tab + page * 4096 + sub;
				tb[(lo | hi) * 16] = xor4(tb[lo * 16], tb[hi * 16]);
			}
		};
		build0(0);
		__syncthreads();
		build1(0);
		__syncthreads();
		auto issue = [&](u32x4 *v, const uint4 &a, int page, int hf) {
			const unsigned mw[4] = { a.x, a.y, a.z, a.w };
#pragma unroll
			for (int q = 0; q < 8; q++) {
				const int s = 8 * hf + q;
				const unsigned sel = (unsigned)(s % 3) | ((4u + (unsigned)(q & 3)) << 8) | ((page ? 3u : 12u) << 16) | (12u << 24);
				const unsigned at = __builtin_amdgcn_perm(mw[2 * hf + (q >> 2)], KC[s / 3], sel);
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
		const uint4 *mitem = mbase + rb0 * 2;
#pragma unroll 1
		for (int k = first_blk; k <= last_blk; k++) {
			const uint4 *mrow = mitem + (i64)((set0 + k) % nsets) * (set_words / 2);
			const bool more = k < last_blk;                 // (uniform)
			auto step = [&](auto PAGE) {
				constexpr int page = decltype(PAGE)::value;
				// rows the NEXT step builds from: (k + 1, page) -- requested now, put into stg[page] at the end of this step
				uint4 staged = make_uint4(0, 0, 0, 0);
				if (more && threadIdx.x < 128) { const int pr = prow[(k + 1) * 256 + page * 128 + threadIdx.x]; if (pr >= 0) staged = Mw[pr]; }
				const bool building = page == 0 || more;    // this step builds page ^ 1: of this block (page 0) / the next (page 1)
				auto half = [&](auto HALF) {
					constexpr int hf2 = decltype(HALF)::value;
					constexpr int HS = SEG / 2, jb = hf2 * HS;
					uint4 m[2];
					u32x4 va[8], vb[8];
					#if GF2_K16K_EXP == 1 || GF2_K16K_EXP == 3
					auto loadm = [&](int j, int slot) { m[slot] = make_uint4(d[j].x * 0x9E3779B9u, d[j].y, d[j].z ^ j, d[j].w); };
#else
					auto loadm = [&](int j, int slot) { m[slot] = (mrow + j * (NW * 64 * 2))[2 * ulane + page]; };
#endif
					loadm(jb, 0); loadm(jb + 1, 1);
					#if GF2_K16K_EXP != 2 && GF2_K16K_EXP != 3
					if (building) { if (hf2 == 0) build0(page ^ 1); else build1(page ^ 1); }
#endif
					issue(va, m[0], page, 0);
#pragma unroll
					for (int jj = 0; jj < HS; jj++) {
						const int j = jb + jj, c = jj & 1;
						issue(vb, m[c], page, 1); fold(d[j], va);
						if (jj + 1 < HS) {
							const uint4 n = m[c ^ 1];
							if (jj + 2 < HS) loadm(j + 2, c);
							issue(va, n, page, 0);
						}
						fold(d[j], vb);
						// pin the folds HERE: the steps are separate basic blocks (the conditional table build / stage write between
						// them), and the compiler otherwise sinks every fold of the k loop to its last block -- keeping all
						// ~1000 looked-up values of a block alive across it (2000 spilled registers)
						asm volatile("" : "+v"(d[j].x), "+v"(d[j].y), "+v"(d[j].z), "+v"(d[j].w));
					}
					if (hf2 == 1 && more && threadIdx.x < 128) stg[page * 128 + threadIdx.x] = staged;
					__syncthreads();
				};
				half(IntC<0>{}); half(IntC<1>{});
			};
			step(IntC<0>{}); step(IntC<1>{});
		}
		// (Requesting the next item's segments while the last block is applied -- each into the registers of the segment just
		// stored -- was built and measured: the two copies of the lookup loop cost ~80 spilled registers and the kernel ran
		// 12 % slower, 3.87 against 4.38 TB/s for K = 4; tools/microbench_update16k.hip.)
#pragma unroll
		for (int j = 0; j < SEG; j++)
			if ((alive >> j) & 1) (Mrow + j * (NW * 64))[ulane] = d[j];
	}
}

