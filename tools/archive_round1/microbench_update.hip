// Isolated timing of the bulk-update kernel (k_update) with synthetic multipliers / pivot rows:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DGF2_MB_NOLOOKUP | -DGF2_MB_L2] tools/microbench_update.hip -o /tmp/mbu && /tmp/mbu
// full build: real kernel;  NOLOOKUP: HBM stream only;  L2: table work only (row data stays in L2).
#define GF2_TW 8        /* the 64-byte-tile kernel of round 1 (the product builds with GF2_TW = 2 since round 2) */
#include "../gf2bv_amd/csrc/gf2_kernels.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int G, int T, int NT>
void run(const char *name, u64 *M, i64 rows, i64 srows, int ntiles, PanelRec *panels, PanelAux *aux, u64 *mult, int *blkf, int nsplit)
{
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto launch = [&] { k_update<G, T, NT><<<dim3(256), dim3(NT), 0>>>(M, rows, srows, 0, G, 0, panels, aux, mult, blkf, 0, ntiles, 1, 0, 0, SysStride{0, 0}); };
	launch(); CK(hipDeviceSynchronize());
	const int reps = getenv("MB_REPS") ? atoi(getenv("MB_REPS")) : 5;      // MB_REPS=400: sustained load (clocks settle)
	CK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
	double bytes = 2.0 * (double)(rows - 256) * ntiles * GF2_TW * 8;
	printf("%-10s G=%d T=%2d NT=%4d nsplit=%3d: %.3f ms  %.2f TB/s per pass  (x%d = %.2f TB/s single-panel equivalent)\n", name, G, T, NT, nsplit, ms,
	       bytes / ms / 1e9, G, G * bytes / ms / 1e9);
#ifdef GF2_STEP_PROBE      /* per-workgroup durations of one more launch (-DGF2_STEP_PROBE) */
	{
		int zero = 0; CK(hipMemcpyToSymbol(HIP_SYMBOL(gf2_probe_j0), &zero, sizeof(int)));
		launch(); CK(hipDeviceSynchronize());
		static unsigned long long h[GF2_PROBE_WGS][6];
		CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(gf2_probe_upd), sizeof(h)));
		unsigned long long t0 = ~0ull; for (int k = 0; k < 256; k++) if (h[k][0] < t0) t0 = h[k][0];
		printf("   WG end (us) by index, every 8th:");
		for (int k = 0; k < 256; k += (getenv("MB_ALL") ? 1 : 8)) printf(" %.0f", (h[k][2] - t0) / 100.0);
		printf("\n   second span starts at (us), every 8th:");
		for (int k = 0; k < 256; k += (getenv("MB_ALL") ? 1 : 8)) printf(" %.0f", h[k][5] ? (h[k][5] - t0) / 100.0 : 0.0);
		static unsigned long long hw[5][16];
		CK(hipMemcpyFromSymbol(hw, HIP_SYMBOL(gf2_probe_wave), sizeof(hw)));
		for (int q = 0; q < 4; q++) { printf("\n   wavefront ends of WG %d:", 8 + 64 * q); for (int w = 0; w < 16; w++) printf(" %.0f", (hw[q][w] - t0) / 100.0); }
		double mean = 0, mx = 0; for (int k = 0; k < 256; k++) { double e = (h[k][2] - t0) / 100.0; mean += e / 256; if (e > mx) mx = e; }
		printf("\n   mean end %.0f us, last %.0f us\n", mean, mx);
		int off = -1; CK(hipMemcpyToSymbol(HIP_SYMBOL(gf2_probe_j0), &off, sizeof(int)));
	}
#endif
}

int main(int argc, char **argv)
{
	const i64 rows = argc > 1 ? atol(argv[1]) : 131072;
	const int ntiles = argc > 2 ? atoi(argv[2]) : 1024 / GF2_TW;     // 1 KiB of row width by default
	const i64 srows = (rows + 63) / 64 * 64 + 2;
	u64 *M, *mult; PanelRec *panels; PanelAux *aux; int *blkf;
	CK(hipMalloc(&M, (size_t)ntiles * srows * GF2_TW * 8)); CK(hipMemset(M, 0x5a, (size_t)ntiles * srows * GF2_TW * 8));
	CK(hipMalloc(&mult, (size_t)GF2_GMAX * rows * 8));
	std::vector<u64> hm((size_t)GF2_GMAX * rows);
	u64 x = 88172645463325252ull;
	for (auto &v : hm) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x; }
	for (int g = 0; g < GF2_GMAX; g++) for (int r = 0; r < 256; r++) hm[(size_t)g * rows + r] = 0;     // pivot rows: not updated
	CK(hipMemcpy(mult, hm.data(), hm.size() * 8, hipMemcpyHostToDevice));
	std::vector<PanelRec> hp(GF2_GMAX); std::vector<PanelAux> ha(GF2_GMAX);
	for (int g = 0; g < GF2_GMAX; g++) { hp[g].start = 64 * g; hp[g].p = 64; hp[g].mask = ~0ull; for (int k = 0; k < 64; k++) { ha[g].slot_row[k] = 64 * g + k; ha[g].comb[k] = 1ull << k; } }
	CK(hipMalloc(&panels, sizeof(PanelRec) * GF2_GMAX)); CK(hipMemcpy(panels, hp.data(), sizeof(PanelRec) * GF2_GMAX, hipMemcpyHostToDevice));
	CK(hipMalloc(&aux, sizeof(PanelAux) * GF2_GMAX)); CK(hipMemcpy(aux, ha.data(), sizeof(PanelAux) * GF2_GMAX, hipMemcpyHostToDevice));
	int first = 256; CK(hipMalloc(&blkf, 4)); CK(hipMemcpy(blkf, &first, 4, hipMemcpyHostToDevice));
#ifdef GF2_MB_NOLOOKUP
	const char *name = "nolookup";
#elif defined(GF2_MB_L2)
	const char *name = "l2data";
#else
	const char *name = "full";
#endif
	for (int ns : {16}) {
#if GF2_TW == 8
#ifndef GF2_STEP_PROBE
		run<1, 16, 1024>(name, M, rows, srows, ntiles, panels, aux, mult, blkf, ns);
		run<4, 16, 1024>(name, M, rows, srows, ntiles, panels, aux, mult, blkf, ns);
		run<1, 12, 1024>(name, M, rows, srows, ntiles, panels, aux, mult, blkf, ns);
		run<2, 12, 1024>(name, M, rows, srows, ntiles, panels, aux, mult, blkf, ns);
		run<3, 12, 1024>(name, M, rows, srows, ntiles, panels, aux, mult, blkf, ns);
#endif
		run<4, 12, 1024>(name, M, rows, srows, ntiles, panels, aux, mult, blkf, ns);
		run<4, 12, 768>(name, M, rows, srows, ntiles, panels, aux, mult, blkf, ns);
#ifndef GF2_STEP_PROBE
		run<1, 8, 1024>(name, M, rows, srows, ntiles, panels, aux, mult, blkf, ns);
#endif
#else
		run<1, 16, 1024>(name, M, rows, srows, ntiles, panels, aux, mult, blkf, ns);
		run<2, 16, 1024>(name, M, rows, srows, ntiles, panels, aux, mult, blkf, ns);
		run<3, 16, 1024>(name, M, rows, srows, ntiles, panels, aux, mult, blkf, ns);
		run<4, 16, 1024>(name, M, rows, srows, ntiles, panels, aux, mult, blkf, ns);
#endif
	}
	return 0;
}
