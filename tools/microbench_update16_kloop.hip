// K-fusion of the bulk update, costed in isolation.
//
// Product: one pass over HBM per block of 256 pivots (k_update16: 16-byte row segment in, 32 table lookups, out).
// Here a workgroup keeps S x 512 row segments of ONE column tile in registers (S = 8 / 12 / 16 -> 4096 / 6144 / 8192 rows,
// 64 - 128 KiB: the register file is the only on-chip store left, the LDS holds the tables) across K consecutive blocks:
// per block it rebuilds the 32 byte-field tables from that block's 256 pivot-row segments (prefetched during the previous
// block) and does the 32 lookups per segment with that block's 32 B of multipliers (streamed from L2); the rows are loaded
// and stored ONCE per K blocks.  Memory-side traffic per segment update: 32 B of multipliers + 32 B / K of row data
// (product: 32 + 32), LDS traffic unchanged, but one table build per S x 512 segments instead of one per ~100 000.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_update16_kloop.hip -o /tmp/mbk && /tmp/mbk [rows] [tiles]
// Synthetic pivot rows and multipliers; the result is checked against a host recomputation on a sample of rows.
// Reported: time per launch, and the rate in the product's unit (256-pivot sweep-words: 16 B x K per segment) so that it
// reads against k_update16's 4.0 - 4.1 TB/s per pass on the same box (tools/microbench_update16.hip).
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
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4 *lds_u4_ptr;

__host__ __device__ inline int mult_slot(int g, i64 row) { return g ^ (int)((row >> 3) & 1); }
__host__ __device__ inline u64 mult_rot(u64 m, i64 row) { const int sh = 8 * (int)(row & 7); return sh ? ((m >> sh) | (m << (64 - sh))) : m; }
__device__ __forceinline__ uint4 xor4(uint4 a, uint4 b) { return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }

// M: tile-major, [tile][row] uint4.  piv: [block][tile][256] uint4 (the block's pivot-row segments of the tile, panel-major).
// mult: [block][row][2] uint4 in the product's stored form (slot g ^ rq_hi, bytes rotated by rq_lo).
template <int S, int K>
__global__ void __launch_bounds__(512)
k_kloop(uint4 *__restrict__ M, i64 rows, int ntiles, const uint4 *__restrict__ piv, const uint4 *__restrict__ mult, int order)
{
	constexpr int NT = 512, NW = 8;
	__shared__ __attribute__((aligned(256))) uint4 tab[2 * 256 * 16];      // 128 KiB at LDS address 0
	__shared__ uint4 stage[256];
	if ((unsigned)(size_t)tab != 0u) __builtin_trap();
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	unsigned KC[6];
	{
		const int rq_lo = lane & 7, rq_hi = (lane >> 3) & 1;
#pragma unroll
		for (int v = 0; v < 6; v++) {
			unsigned k = 1u << 24;
#pragma unroll
			for (int b = 0; b < 3; b++) { const int s = 3 * v + b; if (s < 16) k |= (unsigned)(16 * (8 * ((s >> 3) ^ rq_hi) + (((s & 7) + rq_lo) & 7))) << (8 * b); }
			asm volatile("" : "+v"(k));
			KC[v] = k;
		}
	}
	constexpr i64 CH = (i64)S * NT;                     // rows per item
	const i64 nch = rows / CH;                          // (rows is a multiple of CH in this benchmark)
	const i64 items = (i64)ntiles * nch;
	for (i64 it = blockIdx.x; it < items; it += gridDim.x) {
		// order 0: tile-major (neighbouring workgroups: neighbouring row chunks of one tile); 1: chunk-major (the same rows of
		// neighbouring tiles: the workgroups share their multipliers, but sit rows x 16 B apart in HBM)
		const int tile = order ? (int)(it % ntiles) : (int)(it / nch);
		const i64 r0 = (order ? it / ntiles : it % nch) * CH;
		uint4 *Mw = M + (i64)tile * rows;
		uint4 d[S];
#pragma unroll
		for (int j = 0; j < S; j++) d[j] = Mw[r0 + ((i64)j * NW + wv) * 64 + lane];
		uint4 staged = make_uint4(0, 0, 0, 0);
		if (threadIdx.x < 256) staged = piv[((i64)0 * ntiles + tile) * 256 + threadIdx.x];
#pragma unroll 1
		for (int k = 0; k < K; k++) {
			__syncthreads();                            // the previous block's lookups are done with the tables
			if (threadIdx.x < 256) stage[threadIdx.x] = staged;
			__syncthreads();
			for (int e = threadIdx.x; e < 2 * 31 * 16; e += NT) {            // entries with bits in one nibble only
				const int sub = e & 15, q = (e >> 4) % 31, grp = (e >> 4) / 31;
				const int idx = q <= 15 ? q : (q - 15) << 4;
				const uint4 *st = stage + (2 * grp + (sub >> 3)) * 64 + 8 * (sub & 7);
				uint4 acc = make_uint4(0, 0, 0, 0);
				int bits = idx;
				while (bits) { const int l = __ffs(bits) - 1; bits &= bits - 1; acc = xor4(acc, st[l]); }
				tab[grp * 4096 + idx * 16 + sub] = acc;
			}
			__syncthreads();
			for (int e = threadIdx.x; e < 2 * 225 * 16; e += NT) {           // mixed = low-nibble entry ^ high-nibble entry
				const int sub = e & 15, q = (e >> 4) % 225, grp = (e >> 4) / 225;
				const int lo = 1 + q % 15, hi = (1 + q / 15) << 4;
				uint4 *tb = tab + grp * 4096 + sub;
				tb[(lo | hi) * 16] = xor4(tb[lo * 16], tb[hi * 16]);
			}
			if (k + 1 < K && threadIdx.x < 256) staged = piv[((i64)(k + 1) * ntiles + tile) * 256 + threadIdx.x];
			__syncthreads();
			const uint4 *mq = mult + (i64)k * rows * 2;
			uint4 m0[2], m1[2];
			auto loadm = [&](int j, int slot) {
				const int jc = j < S ? j : S - 1;
				const i64 row = r0 + ((i64)jc * NW + wv) * 64 + lane;
				m0[slot] = mq[row * 2]; m1[slot] = mq[row * 2 + 1];
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
			for (int j = 0; j < S; j++) {
				const int c = j & 1;
				issue(vb, m0[c], m1[c], 1); fold(d[j], va);
				issue(va, m0[c], m1[c], 2); fold(d[j], vb);
				issue(vb, m0[c], m1[c], 3); fold(d[j], va);
				const uint4 n0 = m0[c ^ 1], n1 = m1[c ^ 1];
				loadm(j + 2, c);                        // (past the end: re-reads the last rows' multipliers, unused)
				issue(va, n0, n1, 0); fold(d[j], vb);   // (past the end: a harmless extra round)
			}
		}
#pragma unroll
		for (int j = 0; j < S; j++) Mw[r0 + ((i64)j * NW + wv) * 64 + lane] = d[j];
	}
}

static u64 rng_state = 88172645463325252ull;
static u64 rnd() { u64 &x = rng_state; x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; }

template <int S, int K>
void run(uint4 *M, i64 rows, int ntiles, uint4 *piv, uint4 *mult, bool check, const std::vector<u64> &hp, const std::vector<u64> &plain)
{
	const i64 CH = (i64)S * 512;
	const i64 rr = rows / CH * CH;
	if (rr <= 0) return;
	if (check) {
		const int ct = std::min(ntiles, 3);
		std::vector<u64> h0((size_t)ct * rr * 2), h1(h0.size());
		for (auto &v : h0) v = rnd();
		CK(hipMemcpy(M, h0.data(), h0.size() * 8, hipMemcpyHostToDevice));
		// (piv is indexed with the launch's ntiles: the check launches with the full tile count but only compares ct tiles)
		k_kloop<S, K><<<dim3(97), dim3(512)>>>(M, rr, ntiles, piv, mult, 0);
		CK(hipDeviceSynchronize());
		// the launch above walked every tile with row stride rr: compare the first ct tiles
		CK(hipMemcpy(h1.data(), M, h1.size() * 8, hipMemcpyDeviceToHost));
		i64 bad = 0, checked = 0;
		for (i64 r = 0; r < rr; r += (r < 1024 ? 1 : 61)) for (int t = 0; t < ct; t++) for (int w = 0; w < 2; w++) {
			u64 e = h0[((size_t)t * rr + r) * 2 + w];
			for (int k = 0; k < K; k++) for (int g = 0; g < 4; g++) {
				u64 m = plain[((size_t)k * 4 + g) * rows + r];
				while (m) { const int b = __builtin_ctzll(m); m &= m - 1; e ^= hp[((((size_t)k * ntiles + t) * 256) + 64 * g + b) * 2 + w]; }
			}
			checked++; if (e != h1[((size_t)t * rr + r) * 2 + w]) { if (bad < 3) printf("  MISMATCH S=%d K=%d row %lld tile %d word %d\n", S, K, (long long)r, t, w); bad++; }
		}
		printf("S=%2d K=%d correctness: %lld words checked, %lld wrong\n", S, K, (long long)checked, (long long)bad);
	}
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const int order = getenv("MB_ORDER") ? atoi(getenv("MB_ORDER")) : 0;
	auto launch = [&] { k_kloop<S, K><<<dim3(256), dim3(512)>>>(M, rr, ntiles, piv, mult, order); };
	CK(hipMemset(M, 0x5a, (size_t)rr * ntiles * 16));          // the same contents for every variant
	launch(); launch(); CK(hipDeviceSynchronize());
	const int reps = getenv("MB_REPS") ? atoi(getenv("MB_REPS")) : 10;
	CK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
	const double seg_bytes = (double)rr * ntiles * 16;
	printf("S=%2d (%5lld rows per build) K=%d: %.3f ms per launch = %.3f ms per 256-pivot block   %.2f TB/s in 256-pivot sweep-words   (HBM side: %.2f TB/s r+w)\n",
	       S, (long long)CH, K, ms, ms / K, 2.0 * K * seg_bytes / ms / 1e9, 2.0 * seg_bytes / ms / 1e9);
}

int main(int argc, char **argv)
{
	const i64 rows = argc > 1 ? atol(argv[1]) : 98304;          // multiple of 4096, 6144 and 8192
	const int ntiles = argc > 2 ? atoi(argv[2]) : 512;
	constexpr int KMAX = 8;
	uint4 *M, *piv, *mult;
	CK(hipMalloc(&M, (size_t)ntiles * rows * 16));
	CK(hipMalloc(&piv, (size_t)KMAX * ntiles * 256 * 16));
	CK(hipMalloc(&mult, (size_t)KMAX * rows * 32));
	std::vector<u64> hp((size_t)KMAX * ntiles * 256 * 2);
	for (auto &v : hp) v = rnd();
	CK(hipMemcpy(piv, hp.data(), hp.size() * 8, hipMemcpyHostToDevice));
	std::vector<u64> plain((size_t)KMAX * 4 * rows), hm((size_t)KMAX * rows * 4);
	for (auto &v : plain) v = rnd();
	for (int k = 0; k < KMAX; k++) for (i64 r = 0; r < rows; r++) for (int g = 0; g < 4; g++)
		hm[((size_t)k * rows + r) * 4 + mult_slot(g, r)] = mult_rot(plain[((size_t)k * 4 + g) * rows + r], r);
	CK(hipMemcpy(mult, hm.data(), hm.size() * 8, hipMemcpyHostToDevice));
	printf("# K-loop bulk update: %lld rows x %d tiles of 16 B = %.2f GiB; multipliers %.1f MiB per block\n", (long long)rows, ntiles,
	       (double)rows * ntiles * 16 / 1073741824.0, (double)rows * 32 / 1048576.0);
	const bool chk = !getenv("MB_NOCHECK");
	if (getenv("MB_QUICK")) {                                  // one configuration (size / order sweeps)
		run<16, 4>(M, rows, ntiles, piv, mult, false, hp, plain);
		run<16, 8>(M, rows, ntiles, piv, mult, false, hp, plain);
		return 0;
	}
	for (int round = 0; round < 2; round++) {                  // twice: the second round shows what order / clocks do to the figures
		run<8, 1>(M, rows, ntiles, piv, mult, chk && !round, hp, plain);
		run<8, 2>(M, rows, ntiles, piv, mult, false, hp, plain);
		run<8, 4>(M, rows, ntiles, piv, mult, chk && !round, hp, plain);
		run<8, 8>(M, rows, ntiles, piv, mult, false, hp, plain);
		run<12, 2>(M, rows, ntiles, piv, mult, false, hp, plain);
		run<12, 4>(M, rows, ntiles, piv, mult, chk && !round, hp, plain);
		run<12, 8>(M, rows, ntiles, piv, mult, false, hp, plain);
		run<16, 1>(M, rows, ntiles, piv, mult, false, hp, plain);
		run<16, 2>(M, rows, ntiles, piv, mult, false, hp, plain);
		run<16, 3>(M, rows, ntiles, piv, mult, false, hp, plain);
		run<16, 4>(M, rows, ntiles, piv, mult, chk && !round, hp, plain);
		run<16, 6>(M, rows, ntiles, piv, mult, false, hp, plain);
		run<16, 8>(M, rows, ntiles, piv, mult, false, hp, plain);
	}
	return 0;
}
