// ARCHIVE (not compiled by build()): the bulk update of round 1 -- 64- / 128-byte column tiles (GF2_TW = 8 / 16), four lanes per row
// segment, T = 12 five/six-bit tables per panel -- as it stood in the product header until round 3 (there under `#if GF2_TW != 2`).
// The product layout is GF2_TW = 2 (k_update16); the last commit in which the whole solver builds with -DGF2_TW=8 for A/B runs is
// 0554a43 (round-2 final).  Its helpers (Fields<T>, rot_fields_rt, rowq, UpdateCfg) are kept at the top of this file.
// Rows whose multipliers are all 0 (dead rows, the block's own sources, sparse rows) are not written back.
template <int G, int T>
struct UpdateCfg {
	static constexpr int TW = GF2_TW;
	static constexpr int LPR = GF2_LPR;
	static constexpr int SLOTS = Fields<T>::SLOTS;                  // 256-byte slots per panel
	static constexpr int LDS_BYTES = G * SLOTS * 256 + G * 64 * 4;
};

// (register budget of a 1024-thread workgroup -- 128 VGPRs -- also when launched with fewer threads: a 768-thread
// instance then leaves a quarter of every SIMD's register file to the panel kernels)
template <int G, int T, int NT>
__global__ void __launch_bounds__(1024)
k_update(u64 *__restrict__ M, i64 rows, i64 srows, int j0, int gb, int wlo,
         const PanelRec *__restrict__ panels, const PanelAux *__restrict__ aux,
         const u64 *__restrict__ multset, const int *__restrict__ blk_first,
         int tile_begin, int ntiles, int world, int wrank, int nw_lo, int nw_hi, SysStride ss)
{
	// Words [nw_lo, nw_hi) -- the next block's window -- are never WRITTEN here: the panel stream owns them
	// (k_prio_window has carried them into Wb, and the next block's panel steps store its pivot rows there
	// while this launch is still running).
	{
		const i64 ao = blockIdx.y * ss.arena_bytes;
		M += blockIdx.y * ss.m_words;
		panels = sys_at(panels, ao); aux = sys_at(aux, ao); multset = sys_at(multset, ao); blk_first = sys_at(blk_first, ao);
	}
	typedef UpdateCfg<G, T> C;
	typedef Fields<T> F;
	constexpr int TW = C::TW, LPR = C::LPR, SLOTS = C::SLOTS, IL = F::IL;
	// STATIC shared memory (up to 129 KiB; gfx950 has 160 KiB per CU): the tables then start at LDS address 0
	// known to the compiler, and a lookup address is (field << 8 | lane constant) with nothing to add --
	// two VALU instructions per lookup (shift, v_and_or) instead of three
	__shared__ __attribute__((aligned(256))) uint4 tab[G * SLOTS * 16];
	__shared__ int prow[G * 64];                    // [G][64] physical row of pivot bit, -1 if none
	__shared__ uint4 stage[G * 64 * LPR];           // the tile's segment of every pivot row, [panel][pivot bit][lane] (zero: no pivot)
	const int lr = threadIdx.x % LPR;
	const int rr = threadIdx.x / LPR;
	constexpr int RPP = NT / LPR;

#ifdef GF2_STEP_PROBE
	const bool uprobe = j0 == gf2_probe_j0 && blockIdx.y == 0 && threadIdx.x == 0 && blockIdx.x < GF2_PROBE_WGS;
	unsigned long long up_t0 = 0, up_tab = 0, up_spans = 0;
	if (uprobe) { up_t0 = wall_clock64(); gf2_probe_upd[blockIdx.x][0] = up_t0; }
#endif
	int anyp = 0;
	for (int g = 0; g < gb; g++) anyp |= panels[j0 + g].p;
	if (!anyp) return;                          // a block without pivots changes nothing
	// Work = (tile, alive row) pairs, tile-major: ntiles x R of them.  Every workgroup takes ONE contiguous
	// span of that line -- equal spans, so the launch has no tail of half-empty rounds whatever ntiles is
	// (129 tiles x 8 row ranges were 4.03 rounds of 256 workgroups) -- and rebuilds its tables when the span
	// crosses into the next tile (1 + span/R builds per workgroup).  Span starts are multiples of 1024 rows from
	// a multiple of 8, so a lane's rowq is a constant (rows just below the bound are dead: zero multipliers).
	const i64 rlo = (i64)(*blk_first) & ~(i64)7;
	constexpr int ALIGN = RPP * 4;
	const i64 R = (rows - rlo + ALIGN - 1) / ALIGN * ALIGN;
	const i64 total = (i64)ntiles * R;
	i64 chunk = (total + gridDim.x - 1) / gridDim.x;
	chunk = (chunk + ALIGN - 1) / ALIGN * ALIGN;
	i64 pos = (i64)blockIdx.x * chunk;
	const i64 pend = (pos + chunk < total) ? pos + chunk : total;
	for (bool first_span = true; pos < pend; first_span = false) {
	const int ct = (int)(pos / R);
	const i64 r0 = pos - (i64)ct * R;
	const i64 span = (R - r0 < pend - pos) ? R - r0 : pend - pos;
	pos += span;
	const i64 tile = owned_item(ct, tile_begin, GF2_OWN_LOG - GF2_TW_LOG, world, wrank);     // (column-slab solve: the tiles this rank owns)
	const i64 w0 = tile * TW;
	const i64 rbeg = rlo + r0;
	if (rbeg >= rows) continue;                 // padding at the end of a tile's line
	const i64 rend = (rbeg + span < rows) ? rbeg + span : rows;
	if (!first_span) __syncthreads();           // the previous span's rows are done with the tables
#ifdef GF2_STEP_PROBE
	unsigned long long up_a = 0;
	if (uprobe) { up_a = wall_clock64(); if (up_spans == 1) gf2_probe_upd[blockIdx.x][5] = up_a; }
#endif
	// ---- tables ----
	if (first_span) {                           // which physical row holds pivot bit b of panel g: the same for every tile
		for (int t = threadIdx.x; t < gb * 64; t += NT) {
			const int g = t >> 6, b = t & 63;
			const PanelRec rec = panels[j0 + g];
			prow[t] = ((rec.mask >> b) & 1) ? aux[j0 + g].slot_row[__popcll(rec.mask & ((1ull << b) - 1))] : -1;
		}
		__syncthreads();
	}
	// words below wlo belong to windows the panel path owns: their table slots stay zero
	const uint4 keep = make_uint4((w0 + 2 * lr >= wlo) ? ~0u : 0u, (w0 + 2 * lr >= wlo) ? ~0u : 0u,
	                              (w0 + 2 * lr + 1 >= wlo) ? ~0u : 0u, (w0 + 2 * lr + 1 >= wlo) ? ~0u : 0u);
	const uint4 *Mq = reinterpret_cast<const uint4 *>(M) + tile * srows * LPR;    // this tile's slab, LPR x uint4 per row
#ifdef GF2_STEP_PROBE
	if (uprobe && blockIdx.x == 8 && first_span) gf2_probe_wave[4][0] = wall_clock64();
#endif
	// The pivot rows' segments come in with ONE load per thread, all in flight together, and the entries are then
	// combined from LDS.  (A workgroup whose span crosses into the next tile rebuilds its tables while the other
	// 255 keep HBM saturated: entries that fetched their <= 3 rows one after the other paid the loaded memory
	// latency 24 times in a row -- measured +325 us on a 420 us pass for those workgroups, i.e. for the launch.)
	for (int t = threadIdx.x; t < gb * 64 * LPR; t += NT) {
		const int pr = prow[t / LPR];
		uint4 v = Mq[(i64)(pr >= 0 ? pr : 0) * LPR + lr];
		if (pr < 0) v = make_uint4(0, 0, 0, 0);
		v.x &= keep.x; v.y &= keep.y; v.z &= keep.z; v.w &= keep.w;
		stage[t] = v;
	}
	__syncthreads();
#ifdef GF2_STEP_PROBE
	if (uprobe && blockIdx.x == 8 && first_span) gf2_probe_wave[4][1] = wall_clock64();
#endif
	// A table entry is the XOR of the pivot rows selected by its index.  Pass 0: the PURE entries -- index bits only in
	// the low half or only in the high half of the field -- straight from the staged rows (<= 3 of them); pass 1: the
	// mixed ones = low ^ high.  Each pass enumerates exactly its own entries, and what an item is (group, table,
	// index) depends on the thread alone, not on the panel: decoded once, used for every panel.  (The first version
	// walked all 2048 entries in both passes and decoded each from scratch: 21 iterations of index arithmetic for
	// 8 us of the 9-13 us a build took; tools/probe_step.py prints the phases.)
	for (int pass = 0; pass < 2; pass++) {
		int per_panel = 0;                              // entries of this pass per panel
#pragma unroll
		for (int m = 0; m < F::NG; m++) {
			const int w = F::width(IL * m), kl = w >> 1, nlo = (1 << kl) - 1, nhi = (1 << (w - kl)) - 1;
			per_panel += IL * (pass == 0 ? 1 + nlo + nhi : nlo * nhi);
		}
		for (int it = threadIdx.x; it < per_panel * LPR; it += NT) {
			int e = it / LPR;                           // (lane lr = it % LPR = threadIdx.x % LPR)
			int off = 0, lo = 0, hi = 0, sh = 0;        // tab index of the entry inside a panel, index halves, field position
#pragma unroll
			for (int m = 0; m < F::NG; m++) {
				const int w = F::width(IL * m), kl = w >> 1, nlo = (1 << kl) - 1, nhi = (1 << (w - kl)) - 1;
				const int cnt = IL * (pass == 0 ? 1 + nlo + nhi : nlo * nhi);
				if (e >= 0 && e < cnt) {
					const int k = e / IL, part = e % IL;
					if (pass == 0) { lo = k <= nlo ? k : 0; hi = k <= nlo ? 0 : (k - nlo) << kl; }
					else { lo = 1 + k % nlo; hi = (1 + k / nlo) << kl; }
					off = (F::groupoff(m) + (lo | hi)) * 16 + part * LPR + lr;
					sh = F::shift(IL * m + part);
				}
				e -= cnt;                               // (negative once found)
			}
			const int idx = lo | hi;
			for (int g = 0; g < gb; g++) {
				if (pass == 0) {
					uint4 acc = make_uint4(0, 0, 0, 0);
					int bits = idx;
					while (bits) {
						const int l = __ffs(bits) - 1; bits &= bits - 1;
						acc = xor4(acc, stage[(g * 64 + sh + l) * LPR + lr]);
					}
					tab[g * SLOTS * 16 + off] = acc;
				} else {
					const int base = g * SLOTS * 16 + off - idx * 16;
					tab[base + idx * 16] = xor4(tab[base + lo * 16], tab[base + hi * 16]);
				}
			}
		}
		__syncthreads();
#ifdef GF2_STEP_PROBE
		if (uprobe && blockIdx.x == 8 && first_span) gf2_probe_wave[4][2 + pass] = wall_clock64();
#endif
	}

#ifdef GF2_STEP_PROBE
	if (uprobe) { const unsigned long long now = wall_clock64(); up_tab += now - up_a; if (up_spans++ == 0) gf2_probe_upd[blockIdx.x][1] = now; }
#endif
	// ---- stream the rows ----
	// Per lane: U rows per half-batch; the global loads (multipliers + data) of half-batch h+1 are issued
	// before half-batch h is computed and stored, so every wavefront always has HBM requests in flight
	// while it works through its LDS lookups.  Table reads go out 8 at a time before the first XOR;
	// three-input XORs (v_bitop3) fold two entries at once.
	uint4 *Mw = reinterpret_cast<uint4 *>(M) + tile * srows * LPR;
	const int q = rowq(rr);                         // rbeg and the row steps are multiples of 8
	// byte offset inside a 256-byte slot at step s (< 256), plus the 64-KiB page of the group (the ds_read
	// immediate holds 16 bits)
	constexpr int PAGES = (G * SLOTS * 256 + 65535) / 65536;
	unsigned cbyte[PAGES][IL];
#pragma unroll
	for (int pg = 0; pg < PAGES; pg++)
#pragma unroll
		for (int sidx = 0; sidx < IL; sidx++) {
			unsigned c = ((unsigned)(((q + sidx) % IL) * LPR + lr) * 16u) | ((unsigned)pg << 16);
			// opaque to the optimiser: otherwise it peels the page bit off again and spends v_and + v_add per lookup
			asm volatile("" : "+v"(c));
			cbyte[pg][sidx] = c;
		}
	const char *tabb = reinterpret_cast<const char *>(tab);
	constexpr int U = (G >= 4) ? 1 : GF2_UROWS;      // rows per lane per half-batch (register budget: 128 VGPRs at 1024 threads)
	constexpr int GPB = 8 / IL;                     // groups per batch -> 8 table reads in flight per lane
	struct Half { u64 m[U][G]; uint4 d[U]; int qx[U]; bool on[U]; };
	// FAST (compile-time tag): the half-batch lies entirely inside the row range, all G panels are present and
	// there is no window to deposit.  Then every global load and the store are UNCONDITIONAL -- no control flow
	// around vector-memory instructions -- which is what lets the compiler wait with vmcnt(N > 0): with loads
	// under per-lane or per-launch conditions it cannot count what is outstanding, falls back to vmcnt(0) right
	// after issuing the prefetch, and the software pipeline degenerates to load -> wait -> compute.
	typedef std::integral_constant<bool, true> FastT;
	typedef std::integral_constant<bool, false> SafeT;
	// rows of a half-batch: base + u * rstride + roff (static split: rstride = RPP, roff = rr; dynamic: 16, lane's row)
	auto load_half = [&](auto tag, Half &H, i64 base, int rstride, int roff) {
		constexpr bool FAST = decltype(tag)::value;
#pragma unroll
		for (int u = 0; u < U; u++) {
			const i64 row = base + (i64)u * rstride + roff;
#ifdef GF2_MB_L2               /* tools/microbench_update.hip: keep the row data L2-resident to time the table work alone */
			H.qx[u] = (int)((row & 1023) * LPR + lr);
#else
			H.qx[u] = (int)(row * LPR + lr);
#endif
		}
#pragma unroll
		for (int u = 0; u < U; u++) {
			const i64 row = base + (i64)u * rstride + roff;
			u64 any = 0;
#pragma unroll
			for (int g = 0; g < G; g++) {
				u64 v;
				if (FAST) v = multset[(i64)g * rows + row];
				else v = (row < rend && g < gb) ? multset[(i64)g * rows + row] : 0ull;
				any |= v;
				H.m[u][g] = v;
			}
			H.on[u] = any != 0;
		}
		// the data load does NOT wait for the multipliers (no dependent second memory round trip):
		// every row of the range is fetched; rows that turn out to have zero multipliers are written back
		// unchanged (FAST) or not at all
#pragma unroll
		for (int u = 0; u < U; u++)
			if (FAST || base + (i64)u * rstride + roff < rend) H.d[u] = Mw[H.qx[u]];
	};
	// this lane's two words against the no-write range (only the tile that holds the next window is affected)
	const bool nw0 = (int)(w0 + 2 * lr) >= nw_lo && (int)(w0 + 2 * lr) < nw_hi;
	const bool nw1 = (int)(w0 + 2 * lr + 1) >= nw_lo && (int)(w0 + 2 * lr + 1) < nw_hi;
	const bool nw_tile = (int)w0 < nw_hi && (int)(w0 + TW) > nw_lo;        // uniform per workgroup
	const bool nw_lanes = ((nw_lo | nw_hi) & 1) == 0;                       // the range covers whole lanes (always for G = 4)
	const int qdummy = (int)((srows - 1) * LPR + lr);                       // padding row of the slab (rows < srows - 1)
	auto compute_half = [&](auto tag, Half &H, i64 base) {
		constexpr bool FAST = decltype(tag)::value;
#pragma unroll
		for (int u = 0; u < U; u++) {
			if (!FAST && !H.on[u]) continue;
			uint4 acc = H.d[u];
			if (!FAST || H.on[u]) {
#ifndef GF2_MB_NOLOOKUP        /* tools/microbench_update.hip: time the HBM stream without the table work */
#pragma unroll
			for (int g = 0; g < G; g++) {
				if (!FAST && g >= gb) break;        // tables of absent panels were never built (uniform branch)
				const unsigned mlo = (unsigned)H.m[u][g], mhi = (unsigned)(H.m[u][g] >> 32);
#pragma unroll
				for (int m0 = 0; m0 < F::NG; m0 += GPB) {
					uint4 v[8];
#pragma unroll
					for (int h = 0; h < GPB; h++) {
#pragma unroll
						for (int sidx = 0; sidx < IL; sidx++) {
							const int gm = m0 + h;
							if (gm >= F::NG) continue;
							// byte offset inside the group = field * 256 + (slot part, lane) * 16: the field is moved to
							// bit 8 with ONE shift (or alignbit across the 32-bit boundary), then masked and merged with the
							// lane constant in ONE three-operand op (v_bitop3 / v_and_or); the group base is a compile-time
							// immediate of the ds_read
							const int sh = F::shift(IL * gm + sidx), wd = F::width(IL * gm);
							const unsigned fm = ((1u << wd) - 1) << 8;
							unsigned x;
							if (sh >= 32) x = (sh - 32 >= 8) ? (mhi >> (sh - 40)) : (mhi << (40 - sh));
							else if (sh + wd <= 32) x = (sh >= 8) ? (mlo >> (sh - 8)) : (mlo << (8 - sh));
							else x = __builtin_amdgcn_alignbit(mhi, mlo, sh - 8);      // sh >= 8 whenever a field straddles
							const int goff = (g * SLOTS + F::groupoff(gm)) * 256;     // byte offset of the group (constant after unrolling)
							const unsigned at = (x & fm) | cbyte[goff >> 16][sidx];    // one v_and_or_b32
							v[h * IL + sidx] = *reinterpret_cast<const uint4 *>(tabb + (goff & 0xffff) + at);
						}
					}
					const int nv = ((F::NG - m0 < GPB) ? (F::NG - m0) : GPB) * IL;     // entries actually read (even; folds after unrolling)
#pragma unroll
					for (int h = 0; h < 4; h++) {
						if (h >= nv / 2) break;
						acc.x = __builtin_amdgcn_bitop3_b32(acc.x, v[2 * h].x, v[2 * h + 1].x, 0x96);
						acc.y = __builtin_amdgcn_bitop3_b32(acc.y, v[2 * h].y, v[2 * h + 1].y, 0x96);
						acc.z = __builtin_amdgcn_bitop3_b32(acc.z, v[2 * h].z, v[2 * h + 1].z, 0x96);
						acc.w = __builtin_amdgcn_bitop3_b32(acc.w, v[2 * h].w, v[2 * h + 1].w, 0x96);
					}
				}
			}
#endif
			}
			// FAST: a lane whose two words belong to the next window (whole lanes: the range is lane-aligned there)
			// stores to the slab's padding row instead -- a select on the index, no control flow around the store
			if (FAST) Mw[nw0 ? qdummy : H.qx[u]] = acc;
			else if (!(nw0 | nw1)) Mw[H.qx[u]] = acc;
			else {                                      // the window's tile: 8-byte stores of the words that may be written
				u64 *dst = reinterpret_cast<u64 *>(Mw + H.qx[u]);
				if (!nw0) dst[0] = ((u64)acc.y << 32) | acc.x;
				if (!nw1) dst[1] = ((u64)acc.w << 32) | acc.z;
			}
		}
	};
	constexpr i64 STEP = (i64)RPP * U;
	Half A, B;
	i64 base = rbeg;
	load_half(SafeT(), A, base, RPP, rr);
	if (gb == G && (!nw_tile || nw_lanes))              // full blocks: the branch-free pipeline
		for (; base + 3 * STEP <= rend; base += 2 * STEP) {
			load_half(FastT(), B, base + STEP, RPP, rr);
			compute_half(FastT(), A, base);
			load_half(FastT(), A, base + 2 * STEP, RPP, rr);
			compute_half(FastT(), B, base + STEP);
		}
	// (Handing the rows out dynamically, 16 at a time per wavefront from an LDS counter, makes the four wavefronts of
	// a SIMD finish together -- with this static split they finish at 515 / 570 / 655 / 750 us of a 750 us pass,
	// oldest first -- but the pass is not shorter: the SIMD is issue-bound whatever the number of waves left.
	// Same chip, same run: 262144^2 -1 %, 131072^2 +-0, 65536^2 +7 % (the early finishers make room for the panel
	// kernels).  Not kept.)
	for (; base < rend; base += 2 * STEP) {                 // the range's tail, the window's tile, partial blocks
		load_half(SafeT(), B, base + STEP, RPP, rr);    // rows >= rend load nothing (on = false)
		compute_half(SafeT(), A, base);
		load_half(SafeT(), A, base + 2 * STEP, RPP, rr);
		compute_half(SafeT(), B, base + STEP);
	}
	}       // spans
#ifdef GF2_STEP_PROBE
	if (j0 == gf2_probe_j0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && (blockIdx.x & 63) == 8 && blockIdx.x < 256)
		gf2_probe_wave[blockIdx.x >> 6][threadIdx.x >> 6] = wall_clock64();
	if (uprobe) { gf2_probe_upd[blockIdx.x][2] = wall_clock64(); gf2_probe_upd[blockIdx.x][3] = up_spans; gf2_probe_upd[blockIdx.x][4] = up_tab; }
#endif
}

