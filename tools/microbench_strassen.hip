// Costing of the reference's own lever for the large Schur updates ("Missing 1"): M4RI's _mzd_pluq
// (gf2bv/_internal.c:431-433) is block-recursive and multiplies with Strassen-Winograd over an M4RM base case
// (mzd_addmul -> _mzd_addmul_even -> M4RM).  Here: the GF(2) product  C (R x cols) ^= A (R x 256 nb) . B (256 nb x cols)
// in the solver's own operand forms --
//   C  tile-major 16-byte row segments (tile t, row r at C[t * ts + r]),
//   A  per-row multipliers, 32 B per row and block of 256 pivots, in the STORED form of the panel path (midx / mult_stored),
//   B  compact pivot rows [block][tile][256] x 16 B (what k_block_trsm leaves in Pc),
// with the table / lookup code of the outer pass k_update16k as the base case (rows held in registers across ALL nb blocks:
// C is read once and written once per base-case product), and 0 ... L levels of Strassen-Winograd above it, the additions
// as plain one-access-per-thread HBM streams.  The result of every level is compared word for word with the classical one.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_strassen.hip -o /tmp/mbs && /tmp/mbs [R] [tiles] [nb] [max level]
// defaults: R = 262144 rows, 1024 tiles (131072 columns), nb = 512 blocks (131072 pivots): C(N x N/2) ^= A(N x N/2) . B(N/2 x N/2), N = 262144
#include "../gf2bv_amd/csrc/gf2_kernels.hip.h"
#include "three_level/gf2_three_level.hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// operand views and kernels: the solver's own (gf2_kernels.hip.h, "THREE-LEVEL ELIMINATION"): MulC / MulA / MulB, k_mul16k, k_xor16
typedef MulC CV;
typedef MulA AV;
typedef MulB BV;
#define k_xor k_xor16

static double g_add_bytes = 0, g_mul_words = 0;
static int g_base_launches = 0;
static void xor_any(uint4 *X, i64 xs, const uint4 *Y, i64 ys, const uint4 *Z, i64 zs, const uint4 *W, i64 ws, i64 inner, i64 outer)
{
	k_xor<<<dim3((unsigned)((inner + 255) / 256), (unsigned)outer), dim3(256)>>>(X, xs, Y, ys, Z, zs, W, ws, inner);
	g_add_bytes += (double)inner * outer * 16 * (W ? 4 : 3);
}

struct Temps { uint4 *S, *T, *U, *V; };          // per level: A-quadrant, B-quadrant, two C-quadrants
static std::vector<Temps> g_tmp;
static int g_wgs_cap = 0;

static void mul(CV C, i64 R, int T, AV A, BV B, int nb, int level, bool zero, int depth = 0)
{
	constexpr i64 CH = (i64)GF2_KSEG * 512;
	if (level == 0 || R % (2 * CH) || (T & 1) || (nb & 1) || T < 2 || nb < 2) {
		const i64 items = R / CH * T;
		const unsigned wgs = (unsigned)(g_wgs_cap > 0 ? std::min<i64>(items, g_wgs_cap) : items);
		if (zero) k_mul16k<GF2_KSEG, true><<<dim3(wgs), dim3(512)>>>(C, R, T, A, B, nb);
		else k_mul16k<GF2_KSEG, false><<<dim3(wgs), dim3(512)>>>(C, R, T, A, B, nb);
		g_mul_words += (double)R * T * 2 * nb;
		g_base_launches++;
		return;
	}
	const Temps &t = g_tmp[depth];
	const i64 R2 = R / 2; const int T2 = T / 2, n2 = nb / 2;
	const CV C11{ C.p, C.ts }, C12{ C.p + (i64)T2 * C.ts, C.ts }, C21{ C.p + R2, C.ts }, C22{ C.p + (i64)T2 * C.ts + R2, C.ts };
	const AV A11{ A.p, A.bs }, A12{ A.p + (i64)n2 * A.bs, A.bs }, A21{ A.p + R2 * 2, A.bs }, A22{ A.p + (i64)n2 * A.bs + R2 * 2, A.bs };
	const BV B11{ B.p, B.bs }, B12{ B.p + (i64)T2 * 256, B.bs }, B21{ B.p + (i64)n2 * B.bs, B.bs }, B22{ B.p + (i64)n2 * B.bs + (i64)T2 * 256, B.bs };
	const AV S{ t.S, R2 * 2 }; const BV Tq{ t.T, (i64)T2 * 256 }; const CV U{ t.U, R2 }, V{ t.V, R2 };
	auto xa = [&](const AV &y, const AV &z) { xor_any(t.S, S.bs, y.p, y.bs, z.p, z.bs, nullptr, 0, R2 * 2, n2); };
	auto xb = [&](const BV &y, const BV &z) { xor_any(t.T, Tq.bs, y.p, y.bs, z.p, z.bs, nullptr, 0, (i64)T2 * 256, n2); };
	auto xc = [&](const CV &x, const CV &y, const CV &z, const CV *w) { xor_any(x.p, x.ts, y.p, y.ts, z.p, z.ts, w ? w->p : nullptr, w ? w->ts : 0, R2, T2); };
	if (zero) {      // (overwrite at an inner level: cleared, then accumulated into -- one write + one read of C more than a true overwrite)
		for (int tt = 0; tt < T; tt++) CK(hipMemsetAsync(C.p + (i64)tt * C.ts, 0, (size_t)R * 16, 0));
		g_add_bytes += (double)R * T * 16;
	}
	xa(A21, A22); xb(B12, B11);                        // S1, T1
	mul(V, R2, T2, S, Tq, n2, level - 1, true, depth + 1);        // V = P5
	xa(S, A11); xb(B22, Tq);                           // S2, T2
	mul(U, R2, T2, A11, B11, n2, level - 1, true, depth + 1);     // U = P1
	xc(C11, C11, U, nullptr);
	mul(C11, R2, T2, A12, B21, n2, level - 1, false, depth + 1);  // C11 += P2
	mul(U, R2, T2, S, Tq, n2, level - 1, false, depth + 1);       // U = P1 + P6
	xc(C12, C12, U, &V);
	xa(A12, S);                                        // S4
	mul(C12, R2, T2, S, B22, n2, level - 1, false, depth + 1);    // C12 += P3
	xb(Tq, B21);                                       // T4
	mul(C21, R2, T2, A22, Tq, n2, level - 1, false, depth + 1);   // C21 += P4
	xa(A11, A21); xb(B22, B12);                        // S3, T3
	mul(U, R2, T2, S, Tq, n2, level - 1, false, depth + 1);       // U = P1 + P6 + P7
	xc(C21, C21, U, nullptr);
	xc(C22, C22, U, &V);
}

__global__ void k_fill(u64 *p, i64 n, u64 seed)
{
	const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = mix64(seed ^ mix64((u64)i));
}
__global__ void k_diff(const u64 *a, const u64 *b, i64 n, unsigned long long *cnt)
{
	const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && a[i] != b[i]) atomicAdd(cnt, 1ull);
}

int main(int argc, char **argv)
{
	const i64 R = argc > 1 ? atol(argv[1]) : 262144;
	const int T = argc > 2 ? atoi(argv[2]) : 1024;
	const int nb = argc > 3 ? atoi(argv[3]) : 512;
	const int maxlev = argc > 4 ? atoi(argv[4]) : 3;
	if (getenv("MB_WGS")) g_wgs_cap = atoi(getenv("MB_WGS"));
	constexpr i64 CH = (i64)GF2_KSEG * 512;
	if (R % CH) { printf("R must be a multiple of %lld\n", (long long)CH); return 1; }
	const size_t cbytes = (size_t)R * T * 16, abytes = (size_t)nb * R * 32, bbytes = (size_t)nb * T * 256 * 16;
	uint4 *C0, *C1, *Cin, *A, *B;
	CK(hipMalloc(&C0, cbytes)); CK(hipMalloc(&C1, cbytes)); CK(hipMalloc(&Cin, cbytes)); CK(hipMalloc(&A, abytes)); CK(hipMalloc(&B, bbytes));
	auto fill = [&](void *p, size_t bytes, u64 seed) { const i64 n = (i64)(bytes / 8); k_fill<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>((u64 *)p, n, seed); };
	fill(Cin, cbytes, 1); fill(A, abytes, 2); fill(B, bbytes, 3);
	g_tmp.resize(maxlev + 1);
	size_t tmp_bytes = 0;
	{
		i64 r = R; int t = T, n = nb;
		for (int l = 0; l < maxlev; l++) {          // depth l: quadrants of the operands halved l + 1 times
			r /= 2; t /= 2; n /= 2;
			const size_t sa = (size_t)n * r * 32, sb = (size_t)n * t * 256 * 16, sc = (size_t)r * t * 16;
			CK(hipMalloc(&g_tmp[l].S, sa)); CK(hipMalloc(&g_tmp[l].T, sb)); CK(hipMalloc(&g_tmp[l].U, sc)); CK(hipMalloc(&g_tmp[l].V, sc));
			tmp_bytes += sa + sb + 2 * sc;
		}
	}
	CK(hipDeviceSynchronize());
	printf("# C (%lld x %d bits, %.2f GiB) ^= A (%lld x %d, %.2f GiB) . B (%d x %d, %.2f GiB); base case: k_update16k's table code, %d row segments per lane; temporaries for %d levels: %.2f GiB\n",
	       (long long)R, T * 128, cbytes / 1073741824.0, (long long)R, nb * 256, abytes / 1073741824.0, nb * 256, T * 128, bbytes / 1073741824.0, GF2_KSEG, maxlev, tmp_bytes / 1073741824.0);
	// host spot check of the base case on the first rows of tile 0 (the bilinear form the lookups implement: pivot (panel g, bit b) of
	// block k selects B[k][tile][64 g + b]; a row's plain multiplier of panel g is its stored word (g ^ rq_hi) rotated left by 8 (row & 7))
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	unsigned long long *dcnt; CK(hipMalloc(&dcnt, 8));
	double t_classic = 0;
	for (int lev = 0; lev <= maxlev; lev++) {
		uint4 *Cx = lev == 0 ? C0 : C1;
		float best = 1e30f;
		const int reps = lev == 0 ? 2 : 2;
		for (int rep = 0; rep < reps; rep++) {
			CK(hipMemcpy(Cx, Cin, cbytes, hipMemcpyDeviceToDevice));
			g_add_bytes = 0; g_mul_words = 0; g_base_launches = 0;
			CK(hipDeviceSynchronize());
			CK(hipEventRecord(e0));
			mul(CV{ Cx, R }, R, T, AV{ A, R * 2 }, BV{ B, (i64)T * 256 }, nb, lev, false);
			CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
			CK(hipGetLastError());
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			best = std::min(best, ms);
		}
		unsigned long long bad = 0;
		if (lev > 0) {
			CK(hipMemset(dcnt, 0, 8));
			const i64 n = (i64)(cbytes / 8);
			k_diff<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>((const u64 *)C0, (const u64 *)C1, n, dcnt);
			CK(hipMemcpy(&bad, dcnt, 8, hipMemcpyDeviceToHost));
		} else t_classic = best;
		const double classic_words = (double)R * T * 2 * nb;
		printf("level %d: %9.3f ms  (%.3f of classical)  %6.2f TB/s in classical 256-pivot sweep-words; base-case launches %4d doing %.4f of the classical lookups, additions move %7.2f GB; differing words vs classical: %llu\n",
		       lev, best, best / t_classic, classic_words * 16 / best / 1e9, g_base_launches, g_mul_words / classic_words, g_add_bytes / 1e9, bad);
		fflush(stdout);
	}
	if (getenv("MB_HOSTCHECK")) {
		// classical result of a few rows against a host recomputation
		const int rows_chk = 48;
		std::vector<u64> hc((size_t)rows_chk * 2), hin((size_t)rows_chk * 2), ha((size_t)nb * rows_chk * 4), hb((size_t)nb * 256 * 2);
		CK(hipMemcpy(hc.data(), C0, hc.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hin.data(), Cin, hin.size() * 8, hipMemcpyDeviceToHost));
		for (int k = 0; k < nb; k++) {
			CK(hipMemcpy(&ha[(size_t)k * rows_chk * 4], (const u64 *)A + (size_t)k * R * 4, (size_t)rows_chk * 32, hipMemcpyDeviceToHost));
			CK(hipMemcpy(&hb[(size_t)k * 512], (const u64 *)B + (size_t)k * T * 512, 4096, hipMemcpyDeviceToHost));
		}
		int wrong = 0;
		for (int r = 0; r < rows_chk; r++) {
			u64 x0 = hin[2 * r], x1 = hin[2 * r + 1];
			for (int k = 0; k < nb; k++)
				for (int g = 0; g < 4; g++) {
					const u64 st = ha[((size_t)k * rows_chk + r) * 4 + (g ^ ((r >> 3) & 1))];
					const int sh = 8 * (r & 7);
					const u64 m = sh ? ((st << sh) | (st >> (64 - sh))) : st;
					for (int b = 0; b < 64; b++) if ((m >> b) & 1) { x0 ^= hb[((size_t)k * 256 + 64 * g + b) * 2]; x1 ^= hb[((size_t)k * 256 + 64 * g + b) * 2 + 1]; }
				}
			if (x0 != hc[2 * r] || x1 != hc[2 * r + 1]) wrong++;
		}
		printf("host check of the classical product, %d rows of tile 0: %d wrong\n", rows_chk, wrong);
	}
	return 0;
}
