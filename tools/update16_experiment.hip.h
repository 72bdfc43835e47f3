// EXPERIMENT (not part of the product): the bulk update with 16-byte column tiles and byte bit-fields.
// Included by tools/microbench_update16.hip only.  Result on MI355X (1 GiB pass, 131072 rows x 512 tiles, G = 4):
//   table work alone 0.32-0.35 ms, HBM stream WITHOUT the multiplier loads 0.433 ms (5.0 TB/s, the chip's in-place
//   read-XOR-write ceiling), stream WITH them 0.52 ms, whole kernel 0.52-0.53 ms = 4.1 TB/s -- against 0.537 ms =
//   3.99 TB/s for the product's k_update<4,12,768> on the same box.  A 16-byte tile needs 32 bytes of multipliers per
//   16 bytes of row data: those loads hit in L2 (FETCH_SIZE shows no extra fabric traffic) but every GiB of them costs
//   ~0.045 ms, so the kernel is bound by L2 -> CU traffic instead of LDS + VALU issue; the +3 % does not pay for a
//   change of the working layout.  See DESIGN.md section 4.
#pragma once
// ------------------------------------------------------------------------------------------
// Bulk update, second form: 16-BYTE column tiles and BYTE bit-fields.
//
// The LDS holds G x T x 2^k x E bytes of tables (G panels, T = 64/k fields of k bits, entries of E bytes =
// the tile width); the lookups a row segment costs are G x T whatever E is.  k_update above spends its
// 128 KiB on E = 64: T = 12 fields of 5-6 bits, 48 lookups of 64 B per 64-byte segment -- 24 bytes of LDS
// reads per byte of HBM traffic, and the LDS + VALU issue is what binds it.  Here E = 16:
// a tile is TWO words wide, a lane owns a whole row segment, k = 8: 32 tables of 256 entries = 128 KiB, 32
// lookups of 16 B per 16-byte segment -- 16 bytes of LDS per HBM byte -- and the address of a lookup is ONE
// v_perm_b32 (field byte -> bits 8..15, lane constant -> bits 0..7, 64-KiB page -> bit 16).
//
// LDS layout: two groups (pages) of 16 tables = panels {0,1} and {2,3}; slot `idx` of a group = 256 B =
// [entry idx of table 0 | ... | table 15] = all 64 banks; table 8 * (panel & 1) + byte.  ds_read_b128 is
// serviced 16 lanes at a time whose rows (consecutive lanes = consecutive rows) have 16 different values of
// rq = row & 15; at step s = 8 * s_hi + s_lo of a group a row reads table 8 * (s_hi ^ rq_hi) + ((s_lo + rq_lo) & 7)
// (rq_lo = rq & 7, rq_hi = rq >> 3): a bijection of rq for every s, so every read touches each bank once.
// The panel path stores the multipliers so that this needs no per-lookup work: mult4[row][j] (32 B per row,
// two 16-byte loads) = rotr64(multiplier of panel j ^ rq_hi, 8 * rq_lo) -- see mult_slot / mult_rot.
__host__ __device__ __forceinline__ int mult_slot(int g, i64 row) { return g ^ (int)((row >> 3) & 1); }
__host__ __device__ __forceinline__ u64 mult_rot(u64 m, i64 row)
{
	const int sh = 8 * (int)(row & 7);
	return sh ? ((m >> sh) | (m << (64 - sh))) : m;
}
__host__ __device__ __forceinline__ u64 mult_unrot(u64 m, i64 row)
{
	const int sh = 8 * (int)(row & 7);
	return sh ? ((m << sh) | (m >> (64 - sh))) : m;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4 *lds_u4_ptr;

// HALF: the tile's first word belongs to the next block's window (the panel stream owns it): only the second
// word is stored (one tile of one block per solve at most, its own small launch).
template <int NT, bool HALF, int DEPTH, bool PIPE>
__global__ void __launch_bounds__(NT)
k_update16(u64 *__restrict__ M, i64 rows, i64 srows, int j0, int gb, int wlo,
           const PanelRec *__restrict__ panels, const PanelAux *__restrict__ aux,
           const u64 *__restrict__ mult4, const int *__restrict__ blk_first,
           int tile_begin, int ntiles, SysStride ss)
{
	{
		const i64 ao = blockIdx.y * ss.arena_bytes;
		M += blockIdx.y * ss.m_words;
		panels = sys_at(panels, ao); aux = sys_at(aux, ao); mult4 = sys_at(mult4, ao); blk_first = sys_at(blk_first, ao);
	}
	constexpr int NW = NT / 64;
	__shared__ __attribute__((aligned(256))) uint4 tab[2 * 256 * 16];      // 128 KiB, must sit at LDS address 0 (checked below)
	__shared__ uint4 stage[GF2_GMAX * 64];          // the tile's segment of every pivot row, [panel][pivot bit] (zero: no pivot)
	__shared__ int prow[GF2_GMAX * 64];             // physical row of pivot bit, -1 if none
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	if ((unsigned)(size_t)tab != 0u) __builtin_trap();      // lookup addresses are absolute (folds away when the compiler placed it at 0)

	int anyp = 0;
	for (int g = 0; g < gb; g++) anyp |= panels[j0 + g].p;
	if (!anyp) return;
	// Work = (tile, row) pairs, tile-major, rows from the alive bound (rounded down to a wavefront's 64) to the
	// padded end; every workgroup takes one contiguous span (see k_update) in steps of NW x 64 rows.
	const i64 rlo = (i64)(*blk_first) & ~(i64)63;
	const i64 R64 = (rows + 63) & ~(i64)63;          // the slab and the multiplier array are padded to this
	constexpr int ALIGN = NW * 64;
	const i64 R = (R64 - rlo + ALIGN - 1) / ALIGN * ALIGN;
	const i64 total = (i64)ntiles * R;
	i64 chunk = (total + gridDim.x - 1) / gridDim.x;
	chunk = (chunk + ALIGN - 1) / ALIGN * ALIGN;
	i64 pos = (i64)blockIdx.x * chunk;
	const i64 pend = (pos + chunk < total) ? pos + chunk : total;

	// lane constants: byte s % 3 of KC[s / 3] = 16 * (table read at step s), byte 3 = 1 (the page bit of group 1)
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
			asm volatile("" : "+v"(k));             // keep them in registers as built
			KC[v] = k;
		}
	}

	for (bool first_span = true; pos < pend; first_span = false) {
		const int ct = (int)(pos / R);
		const i64 r0 = pos - (i64)ct * R;
		const i64 span = (R - r0 < pend - pos) ? R - r0 : pend - pos;
		pos += span;
		const i64 tile = tile_begin + ct;
		const i64 rbeg = rlo + r0;
		if (rbeg >= R64) continue;
		const i64 rend = (rbeg + span < R64) ? rbeg + span : R64;
		if (!first_span) __syncthreads();           // the previous span's rows are done with the tables
		// ---- tables ----
		if (first_span) {
			for (int t = threadIdx.x; t < GF2_GMAX * 64; t += NT) {
				const int g = t >> 6, b = t & 63;
				int pr = -1;
				if (g < gb) {
					const PanelRec rec = panels[j0 + g];
					if ((rec.mask >> b) & 1) pr = aux[j0 + g].slot_row[__popcll(rec.mask & ((1ull << b) - 1))];
				}
				prow[t] = pr;
			}
			__syncthreads();
		}
		const uint4 *Mq = reinterpret_cast<const uint4 *>(M) + tile * srows;       // this tile's slab, one uint4 per row
		if (threadIdx.x < GF2_GMAX * 64) {
			const int pr = prow[threadIdx.x];
			uint4 v = Mq[pr >= 0 ? pr : 0];
			// words left of wlo belong to windows the panel path owns: their table bits stay zero
			const bool k0 = pr >= 0 && 2 * tile >= wlo, k1 = pr >= 0 && 2 * tile + 1 >= wlo;
			if (!k0) { v.x = 0; v.y = 0; }
			if (!k1) { v.z = 0; v.w = 0; }
			stage[threadIdx.x] = v;
		}
		__syncthreads();
		// pass 0: the entries whose index has bits in one nibble only, straight from the staged rows (<= 4 of them)
		for (int it = threadIdx.x; it < 2 * 31 * 16; it += NT) {
			const int sub = it & 15, e = (it >> 4) % 31, grp = (it >> 4) / 31;
			const int idx = e <= 15 ? e : (e - 15) << 4;
			const uint4 *st = stage + (2 * grp + (sub >> 3)) * 64 + 8 * (sub & 7);
			uint4 acc = make_uint4(0, 0, 0, 0);
			int bits = idx;
			while (bits) {
				const int l = __ffs(bits) - 1; bits &= bits - 1;
				acc = xor4(acc, st[l]);
			}
			tab[grp * 4096 + idx * 16 + sub] = acc;
		}
		__syncthreads();
		// pass 1: the mixed ones = low-nibble entry ^ high-nibble entry
		for (int it = threadIdx.x; it < 2 * 225 * 16; it += NT) {
			const int sub = it & 15, k = (it >> 4) % 225, grp = (it >> 4) / 225;
			const int lo = 1 + k % 15, hi = (1 + k / 15) << 4;
			uint4 *tb = tab + grp * 4096 + sub;
			tb[(lo | hi) * 16] = xor4(tb[lo * 16], tb[hi * 16]);
		}
		__syncthreads();

		// ---- stream the rows ----
		// A wavefront takes 64 consecutive rows (1 KiB of the slab, 2 KiB of multipliers) per batch; the loads of
		// batch i+1 are in flight while batch i does its 32 lookups.  No control flow around vector-memory
		// instructions (the compiler then waits with vmcnt(N > 0), see k_update): a wave knows its batch count up
		// front and its last prefetch re-reads its last batch.
		uint4 *Mw = reinterpret_cast<uint4 *>(M) + tile * srows;
		const uint4 *mq = reinterpret_cast<const uint4 *>(mult4);
		const i64 nsteps = (rend - rbeg + ALIGN - 1) / ALIGN;
		// batches of this wave: rows rbeg + (i * NW + wv) * 64 + lane, i < nb
		i64 nb = nsteps;
		if (rbeg + ((nsteps - 1) * NW + wv) * 64 >= rend) nb--;
		struct Bt { uint4 d, m0, m1; i64 row; };
		auto load = [&](Bt &H, i64 i) {
			const i64 ic = i < nb ? i : nb - 1;         // (a prefetch past the end re-reads the last batch: no control flow around loads)
			const i64 row = rbeg + (ic * NW + wv) * 64 + lane;
			H.row = row;
#if defined(GF2_MB_M0ONLY)
			H.m0 = mq[row]; H.m1 = H.m0;
#elif defined(GF2_MB_NTMULT)
			{ typedef unsigned v4u __attribute__((ext_vector_type(4)));
			  const v4u a = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(mq + row * 2)), b = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(mq + row * 2 + 1));
			  H.m0 = make_uint4(a.x, a.y, a.z, a.w); H.m1 = make_uint4(b.x, b.y, b.z, b.w); }
#elif defined(GF2_MB_MULTSMALL)
			H.m0 = mq[(row & 1023) * 2]; H.m1 = mq[(row & 1023) * 2 + 1];      // multipliers from a 32 KiB window (L1/L2-hot)
#elif defined(GF2_MULT_SPLIT)
			H.m0 = mq[row]; H.m1 = mq[R64 + row];      // two arrays of 16 bytes per row: both loads contiguous
#else
			H.m0 = mq[row * 2]; H.m1 = mq[row * 2 + 1];
#endif
#ifdef GF2_MB_L2               /* tools/microbench_update16.hip: keep the row data L2-resident to time the table work alone */
			H.d = Mw[row & 4095];
#else
			H.d = Mw[row];
#endif
		};
		// round r = 0..3 of a batch: the 8 lookups of (group r >> 1, half r & 1)
		auto issue = [&](u32x4 *v, const Bt &H, int r) {
			const unsigned mw[8] = { H.m0.x, H.m0.y, H.m0.z, H.m0.w, H.m1.x, H.m1.y, H.m1.z, H.m1.w };
			const int grp = r >> 1, hf = r & 1;
#pragma unroll
			for (int k = 0; k < 8; k++) {
				const int s = 8 * hf + k;
				// {byte 0: lane constant of step s, byte 1: the field, byte 2: page, byte 3: 0} in one v_perm_b32
				// (selector codes 0-3 = bytes of the second source, 4-7 = bytes of the first, 12 = 0x00)
				const unsigned sel = (unsigned)(s % 3) | ((4u + (unsigned)(k & 3)) << 8) | ((grp ? 3u : 12u) << 16) | (12u << 24);
				const unsigned at = __builtin_amdgcn_perm(mw[2 * (2 * grp + hf) + (k >> 2)], KC[s / 3], sel);
				v[k] = *(lds_u4_ptr)(size_t)at;
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
		auto store = [&](const Bt &H, const uint4 &acc) {
#ifdef GF2_MB_L2
			const i64 q = H.row & 4095;
#else
			const i64 q = H.row;
#endif
			if (HALF) reinterpret_cast<u64 *>(Mw + q)[1] = ((u64)acc.w << 32) | acc.z;
			else Mw[q] = acc;
		};
		if (nb > 0) {
			// DEPTH batches in flight per wave (loads DEPTH-1 batches ahead); PIPE: the lookups of round r+1 -- also
			// across the batch boundary -- are issued before the XORs of round r, so the LDS always has this wave's
			// next 8 reads queued
			Bt H[DEPTH];
#pragma unroll
			for (int d = 0; d < DEPTH - 1; d++) load(H[d], d);
			u32x4 va[8], vb[8];
#ifndef GF2_MB_NOLOOKUP
			if (PIPE) issue(va, H[0], 0);
#endif
			for (i64 i = 0; i < nb; i += DEPTH) {
#pragma unroll
				for (int d = 0; d < DEPTH; d++) {
					if (i + d >= nb) break;              // wave-uniform
					load(H[(d + DEPTH - 1) % DEPTH], i + d + DEPTH - 1);
					Bt &C = H[d];
					uint4 acc = C.d;
#ifdef GF2_MB_NOLOOKUP         /* tools/microbench_update16.hip: time the HBM stream (multiplier loads included) without the table work */
					acc.x ^= C.m0.x ^ C.m0.y ^ C.m0.z ^ C.m0.w; acc.y ^= C.m1.x ^ C.m1.y ^ C.m1.z ^ C.m1.w;
#else
					if (PIPE) {
						issue(vb, C, 1); fold(acc, va);
						issue(va, C, 2); fold(acc, vb);
						issue(vb, C, 3); fold(acc, va);
						issue(va, H[(d + 1) % DEPTH], 0); fold(acc, vb);      // (past the end: a harmless extra round)
					} else {
#pragma unroll
						for (int r = 0; r < 4; r++) { issue(va, C, r); fold(acc, va); }
					}
#endif
					store(C, acc);
				}
			}
		}
	}       // spans
}

