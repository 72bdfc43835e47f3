// The PRODUCT outer-pass kernel (k_update16k of gf2_kernels.hip.h) in isolation, on synthetic records: is the in-solve rate
// (3.9-4.3 TB/s of 256-pivot sweep-words at 131072^2) the kernel's or its surroundings'?  Companion of
// tools/microbench_update16_kloop.hip (the stripped-down form the design was costed with: 5.0-5.7 at the same size).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_update16k.hip -o /tmp/mbk2 && /tmp/mbk2 [rows] [tiles] [K]
#include "../gf2bv_amd/csrc/gf2_kernels.hip.h"
// variants of the workgroup shape (round 5, late): -DMB_SEG=8 -DMB_NT=1024 -DMB_RB=4 = 16 wavefronts x 8 segments, read batches of 4
#ifdef MB_WIDE        /* -DMB_WIDE: the shipped default kernel, k_update16k_wide */
#define MB_SEG GF2_WSEG
#define MB_NT 1024
#define MB_RB 2
#endif
#ifndef MB_SEG
#define MB_SEG GF2_KSEG
#endif
#ifndef MB_NT
#define MB_NT 512
#endif
#ifndef MB_NB
#define MB_NB 2
#endif
#ifndef MB_RB
#define MB_RB 8
#endif
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
static u64 rng_state = 88172645463325252ull;
static u64 rnd() { u64 &x = rng_state; x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; }
int main(int argc, char **argv)
{
	const i64 rows = argc > 1 ? atol(argv[1]) : 131072;
	const int ntiles = argc > 2 ? atoi(argv[2]) : 1024;
	const int K = argc > 3 ? atoi(argv[3]) : 4;
	const bool aligned = getenv("MB_ALIGNED") != nullptr;          // slab stride = rows (every 1 KiB wave access aligned) instead of rows + 2
	const i64 R64 = (rows + 63) / 64 * 64, srows = aligned ? R64 : R64 + 2;
	const int npan = K * GF2_GMAX;
	u64 *M, *mult; PanelRec *panels; PanelAux *aux; int *died, *blkf;
	const size_t slack = (size_t)MB_SEG * MB_NT * 32;        // the kernel does not clamp a tile's last chunk (the solver leaves this slack too)
	CK(hipMalloc(&M, (size_t)ntiles * srows * 16 + slack)); CK(hipMemset(M, 0x5a, (size_t)ntiles * srows * 16 + slack));
	const i64 set_words = (i64)GF2_GMAX * mult_rows(rows);
	std::vector<u64> hm((size_t)K * set_words);
	for (auto &v : hm) v = rnd();
	for (int k = 0; k < K; k++) for (i64 r = 0; r < 256 * K; r++) for (int g = 0; g < 4; g++) hm[(size_t)k * set_words + r * 4 + g] = 0;   // pivot rows: no multipliers
	CK(hipMalloc(&mult, hm.size() * 8 + slack)); CK(hipMemcpy(mult, hm.data(), hm.size() * 8, hipMemcpyHostToDevice));
	std::vector<PanelRec> hp(npan); std::vector<PanelAux> ha(npan);
	for (int q = 0; q < npan; q++) { hp[q].start = 64 * q; hp[q].p = 64; hp[q].mask = ~0ull; for (int s = 0; s < 64; s++) { ha[q].slot_row[s] = 64 * q + s; ha[q].comb[s] = 1ull << s; } }
	CK(hipMalloc(&panels, sizeof(PanelRec) * npan)); CK(hipMemcpy(panels, hp.data(), sizeof(PanelRec) * npan, hipMemcpyHostToDevice));
	CK(hipMalloc(&aux, sizeof(PanelAux) * npan)); CK(hipMemcpy(aux, ha.data(), sizeof(PanelAux) * npan, hipMemcpyHostToDevice));
	std::vector<int> hd(rows, GF2_NEVER);
	for (int r = 0; r < 64 * npan; r++) hd[r] = r / 64;
	CK(hipMalloc(&died, rows * 4)); CK(hipMemcpy(died, hd.data(), rows * 4, hipMemcpyHostToDevice));
	int first = 0; CK(hipMalloc(&blkf, 4)); CK(hipMemcpy(blkf, &first, 4, hipMemcpyHostToDevice));
	int *gprow; CK(hipMalloc(&gprow, sizeof(int) * GF2_OUTER_LISTS));
	k_outer_prow<<<dim3(1), dim3(256)>>>(0, K, panels, aux, gprow, SysStride{0, 0});
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	// default: one workgroup per item (as the solver launches it); MB_WGS=256: persistent workgroups
	const i64 nitems = (R64 + (i64)MB_SEG * MB_NT - 1) / ((i64)MB_SEG * MB_NT) * ntiles;
	const int wgs = getenv("MB_WGS") ? atoi(getenv("MB_WGS")) : (int)nitems;
	const int order = getenv("MB_ORDER") ? atoi(getenv("MB_ORDER")) : 0;      // != 0: chunk-major items (the solver's order)
#ifdef MB_WIDE        /* the shipped default: k_update16k_wide (GF2_WSEG x 1024 rows per item, budget of 120 registers) */
	auto launch = [&] { k_update16k_wide<<<dim3(wgs), dim3(1024)>>>(M, rows, srows, K, gprow, mult, set_words, 0, K, blkf, died, npan, 0, ntiles, SysStride{0, 0}, order != 0); };
#else
	auto launch = [&] { k_update16k<MB_SEG, MB_NT, MB_RB, MB_NB><<<dim3(wgs), dim3(MB_NT)>>>(M, rows, srows, K, gprow, mult, set_words, 0, K, blkf, died, npan, 0, ntiles, SysStride{0, 0}, order != 0); };
#endif
	launch(); launch(); CK(hipDeviceSynchronize());
	const int reps = 6;
	CK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
	const double bytes = (double)(rows - 64 * npan) * ntiles * 16;
	printf("k_update16k<%d,%d,%d> rows %lld tiles %d K %d wgs %d %s: %.3f ms per launch = %.3f ms per GiB and block   %.2f TB/s in 256-pivot sweep-words\n", MB_SEG, MB_NT, MB_RB,
	       (long long)rows, ntiles, K, wgs, aligned ? "aligned" : "srows=rows+2", ms, ms / K / (bytes / 1073741824.0), 2.0 * K * bytes / ms / 1e9);
	return 0;
}
