// Costing of the "fewer memory-side bytes per pivot" bulk update BEFORE touching the solver.
//
// The table method's LDS budget fixes (pivots per pass) x (bytes a lane owns) = 4096: the product kernel k_update16 sits at
// 256 pivots x 16 B (32 tables of 256 entries of 16 B = 128 KiB); this file times the neighbouring point
//   512 pivots x 8 B: 64 tables of 256 entries of 8 B = 128 KiB, 64 ds_read_b64 lookups per 8-byte row segment,
// i.e. HALF the HBM round trips per pivot, the same LDS bytes per pivot, TWICE the multiplier bytes per pivot (64 B of
// multipliers per 8 B of row data, from L2).  Synthetic tables and multipliers (the table build is not part of the
// question), result checked against a host recomputation on a sample of rows.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_update8x512.hip -o /tmp/mbu8 && /tmp/mbu8 [rows] [tiles] [wgs]
//   variants: -DMB_NOLOOKUP (stream + multiplier loads only)   -DMB_L2 (table work only: row data L2-resident)
//             -DMB_NOMULT (lookups by a register-resident multiplier: no multiplier loads)
// "tiles" are 8-byte (one-word) column tiles: rows x tiles x 8 B of matrix; default 131072 x 1024 = 1 GiB.
//
// LDS layout: two pages of 32 tables; slot idx of a page = [entry idx of table 0 | ... | table 31] = 256 B = all 64
// banks.  ds_read_b64 is serviced 32 lanes at a time (MI355X_MICROARCH.md, LDS table); lane with rq = row & 31 reads
// table (s + rq) & 31 at step s: a bijection of rq for every s, so each half-wave touches every bank pair once.
// The multipliers are stored rotated by rq bytes within each 32-byte page group, as the product stores its 8-byte ones.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned long long u64;
typedef long long i64;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const u32x2 *lds_u2_ptr;

template <int NT, int DEPTH>
__global__ void __launch_bounds__(NT)
k_update8(u64 *__restrict__ M, i64 rows, i64 srows, int ntiles, const uint4 *__restrict__ mult /* [rows][4] = 64 B per row */,
          const u64 *__restrict__ tabsrc /* [2][256][32] */)
{
	constexpr int NW = NT / 64;
	__shared__ __attribute__((aligned(256))) u64 tab[2 * 256 * 32];       // 128 KiB at LDS address 0
	if ((unsigned)(size_t)tab != 0u) __builtin_trap();
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	for (int i = threadIdx.x; i < 2 * 256 * 32; i += NT) tab[i] = tabsrc[i];
	__syncthreads();
	// lane constants: byte b < 3 of KC[v] = 8 * ((3 v + b + rq) & 31), byte 3 = 1 (the 64-KiB page bit of group 1)
	unsigned KC[11];
#pragma unroll
	for (int v = 0; v < 11; v++) {
		unsigned k = 1u << 24;
#pragma unroll
		for (int b = 0; b < 3; b++) k |= (unsigned)(8 * ((3 * v + b + (lane & 31)) & 31)) << (8 * b);
		asm volatile("" : "+v"(k));
		KC[v] = k;
	}
	constexpr int ALIGN = NW * 64;
	const i64 R = (rows + ALIGN - 1) / ALIGN * ALIGN;
	const i64 total = (i64)ntiles * R;
	i64 chunk = (total + gridDim.x - 1) / gridDim.x;
	chunk = (chunk + ALIGN - 1) / ALIGN * ALIGN;
	i64 pos = (i64)blockIdx.x * chunk;
	const i64 pend = (pos + chunk < total) ? pos + chunk : total;
	while (pos < pend) {
		const int tile = (int)(pos / R);
		const i64 r0 = pos - (i64)tile * R;
		const i64 span = (R - r0 < pend - pos) ? R - r0 : pend - pos;
		pos += span;
		const i64 rbeg = r0, rend = (rbeg + span < rows) ? rbeg + span : rows;
		if (rbeg >= rows) continue;
		u64 *Mw = M + (i64)tile * srows;
		const i64 nsteps = (rend - rbeg + ALIGN - 1) / ALIGN;
		i64 nb = nsteps;
		if (rbeg + ((nsteps - 1) * NW + wv) * 64 >= rend) nb--;
		struct Bt { u64 d; uint4 m[4]; i64 row; };
		auto load = [&](Bt &H, i64 i) {
			const i64 ic = i < nb ? i : nb - 1;
			i64 row = rbeg + (ic * NW + wv) * 64 + lane;
			if (row >= rows) row = rows - 1;
			H.row = row;
#ifndef MB_NOMULT
#pragma unroll
			for (int q = 0; q < 4; q++) H.m[q] = mult[row * 4 + q];
#else
			H.m[0] = make_uint4((unsigned)row, (unsigned)row * 3u, (unsigned)row * 5u, (unsigned)row * 7u); H.m[1] = H.m[0]; H.m[2] = H.m[0]; H.m[3] = H.m[0];
#endif
#ifdef MB_L2
			H.d = Mw[row & 4095];
#else
			H.d = Mw[row];
#endif
		};
		// round r = 0..7: lookups of steps 8 (r & 3) .. +7 of page r >> 2
		auto issue = [&](u32x2 *v, const Bt &H, int r) {
			const int grp = r >> 2, q = r & 3;
			const uint4 mm = H.m[2 * grp + (q >> 1)];
			const unsigned mw[2] = { (q & 1) ? mm.z : mm.x, (q & 1) ? mm.w : mm.y };
#pragma unroll
			for (int k = 0; k < 8; k++) {
				const int s = 8 * q + k;
				// {byte 0: lane constant of step s, byte 1: the field byte, byte 2: page, byte 3: 0}
				// selector codes 0-3 = bytes of src1 (second operand), 4-7 = bytes of src0, 12 = 0x00
				const unsigned sel = (unsigned)(s % 3) | ((4u + (unsigned)(k & 3)) << 8) | ((grp ? 3u : 12u) << 16) | (12u << 24);
				const unsigned at = __builtin_amdgcn_perm(mw[k >> 2], KC[s / 3], sel);
				v[k] = *(lds_u2_ptr)(size_t)at;
			}
		};
		auto fold = [&](uint2 &acc, const u32x2 *v) {
#pragma unroll
			for (int h = 0; h < 4; h++) {
				acc.x = __builtin_amdgcn_bitop3_b32(acc.x, v[2 * h].x, v[2 * h + 1].x, 0x96);
				acc.y = __builtin_amdgcn_bitop3_b32(acc.y, v[2 * h].y, v[2 * h + 1].y, 0x96);
			}
		};
		if (nb > 0) {
			Bt H[DEPTH];
#pragma unroll
			for (int d = 0; d < DEPTH - 1; d++) load(H[d], d);
			u32x2 va[8], vb[8];
#ifndef MB_NOLOOKUP
			issue(va, H[0], 0);
#endif
			for (i64 i = 0; i < nb; i += DEPTH) {
#pragma unroll
				for (int d = 0; d < DEPTH; d++) {
					if (i + d >= nb) break;
					load(H[(d + DEPTH - 1) % DEPTH], i + d + DEPTH - 1);
					Bt &C = H[d];
					uint2 acc = make_uint2((unsigned)C.d, (unsigned)(C.d >> 32));
#ifdef MB_NOLOOKUP
#pragma unroll
					for (int q = 0; q < 4; q++) { acc.x ^= C.m[q].x ^ C.m[q].z; acc.y ^= C.m[q].y ^ C.m[q].w; }
#else
					issue(vb, C, 1); fold(acc, va);
					issue(va, C, 2); fold(acc, vb);
					issue(vb, C, 3); fold(acc, va);
					issue(va, C, 4); fold(acc, vb);
					issue(vb, C, 5); fold(acc, va);
					issue(va, C, 6); fold(acc, vb);
					issue(vb, C, 7); fold(acc, va);
					issue(va, H[(d + 1) % DEPTH], 0); fold(acc, vb);
#endif
#ifdef MB_L2
					Mw[C.row & 4095] = ((u64)acc.y << 32) | acc.x;
#else
					Mw[C.row] = ((u64)acc.y << 32) | acc.x;
#endif
				}
			}
		}
	}
}

static u64 rng_state = 88172645463325252ull;
static u64 rnd() { u64 &x = rng_state; x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; }

template <int NT, int DEPTH>
void run(const char *name, u64 *M, i64 rows, i64 srows, int ntiles, uint4 *mult, u64 *tabsrc, int wgs)
{
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto launch = [&] { k_update8<NT, DEPTH><<<dim3(wgs), dim3(NT), 0>>>(M, rows, srows, ntiles, mult, tabsrc); };
	launch(); CK(hipDeviceSynchronize());
	const int reps = getenv("MB_REPS") ? atoi(getenv("MB_REPS")) : 5;
	CK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
	const double bytes = 2.0 * (double)rows * ntiles * 8;
	printf("%-9s NT=%4d D=%d wgs=%4d: %.3f ms per 512-pivot pass  %.2f TB/s per pass (r+w)  = %.2f TB/s in 256-pivot sweep-words\n",
	       name, NT, DEPTH, wgs, ms, bytes / ms / 1e9, 2 * bytes / ms / 1e9);
}

int main(int argc, char **argv)
{
	const i64 rows = argc > 1 ? atol(argv[1]) : 131072;
	const int ntiles = argc > 2 ? atoi(argv[2]) : 1024;
	const int wgs = argc > 3 ? atoi(argv[3]) : 256;
	const i64 srows = (rows + 63) / 64 * 64 + 2;
	u64 *M, *tabsrc; uint4 *mult;
	CK(hipMalloc(&M, (size_t)ntiles * srows * 8));
	CK(hipMalloc(&mult, (size_t)rows * 64));
	CK(hipMalloc(&tabsrc, 2 * 256 * 32 * 8));
	// tables: T[t][v], t < 64, v < 256, linear in v (as real Four-Russians tables are): T[t][v] = XOR of gen[t][bit] over the bits of v
	std::vector<u64> gen(64 * 8), T(64 * 256), htab(2 * 256 * 32);
	for (auto &g : gen) g = rnd();
	for (int t = 0; t < 64; t++) for (int v = 0; v < 256; v++) { u64 e = 0; for (int b = 0; b < 8; b++) if (v >> b & 1) e ^= gen[t * 8 + b]; T[t * 256 + v] = e; }
	for (int t = 0; t < 64; t++) for (int v = 0; v < 256; v++) htab[(size_t)(t >> 5) * 8192 + v * 32 + (t & 31)] = T[t * 256 + v];
	CK(hipMemcpy(tabsrc, htab.data(), htab.size() * 8, hipMemcpyHostToDevice));
	// multipliers: plain[row][64 bytes]; stored: byte s of page g = plain byte 32 g + ((s + rq) & 31)
	std::vector<uint8_t> plain((size_t)rows * 64), stored((size_t)rows * 64);
	for (size_t i = 0; i < plain.size(); i += 8) { u64 v = rnd(); memcpy(&plain[i], &v, 8); }
	for (i64 r = 0; r < rows; r++) for (int g = 0; g < 2; g++) for (int s = 0; s < 32; s++)
		stored[(size_t)r * 64 + 32 * g + s] = plain[(size_t)r * 64 + 32 * g + ((s + (int)(r & 31)) & 31)];
	CK(hipMemcpy(mult, stored.data(), stored.size(), hipMemcpyHostToDevice));
#if !defined(MB_NOLOOKUP) && !defined(MB_L2) && !defined(MB_NOMULT)
	const char *name = "full";
	{
		const i64 crow = std::min<i64>(rows, 20000); const int ct = std::min(ntiles, 5);
		const i64 csr = (crow + 63) / 64 * 64 + 2;
		std::vector<u64> h0((size_t)ct * csr), h1(h0.size());
		for (auto &v : h0) v = rnd();
		CK(hipMemcpy(M, h0.data(), h0.size() * 8, hipMemcpyHostToDevice));
		k_update8<512, 3><<<dim3(7), dim3(512), 0>>>(M, crow, csr, ct, mult, tabsrc);
		CK(hipDeviceSynchronize());
		CK(hipMemcpy(h1.data(), M, h1.size() * 8, hipMemcpyDeviceToHost));
		i64 bad = 0, checked = 0;
		for (i64 r = 0; r < crow; r += (r < 2048 ? 1 : 37)) for (int t = 0; t < ct; t++) {
			u64 e = h0[(size_t)t * csr + r];
			for (int q = 0; q < 64; q++) e ^= T[q * 256 + plain[(size_t)r * 64 + q]];
			checked++; if (e != h1[(size_t)t * csr + r]) { if (bad < 5) printf("  MISMATCH row %lld tile %d\n", (long long)r, t); bad++; }
		}
		printf("correctness: %lld words checked, %lld wrong\n", (long long)checked, (long long)bad);
	}
#elif defined(MB_NOLOOKUP)
	const char *name = "nolookup";
#elif defined(MB_L2)
	const char *name = "l2data";
#else
	const char *name = "nomult";
#endif
	CK(hipMemset(M, 0x5a, (size_t)ntiles * srows * 8));
	run<1024, 2>(name, M, rows, srows, ntiles, mult, tabsrc, wgs);
	run<768, 2>(name, M, rows, srows, ntiles, mult, tabsrc, wgs);
	run<512, 2>(name, M, rows, srows, ntiles, mult, tabsrc, wgs);
	run<1024, 3>(name, M, rows, srows, ntiles, mult, tabsrc, wgs);
	run<768, 3>(name, M, rows, srows, ntiles, mult, tabsrc, wgs);
	run<512, 3>(name, M, rows, srows, ntiles, mult, tabsrc, wgs);
	run<512, 4>(name, M, rows, srows, ntiles, mult, tabsrc, wgs);
	return 0;
}
