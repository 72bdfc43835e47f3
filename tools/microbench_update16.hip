// Isolated timing + correctness check of the 16-byte-tile bulk update (k_update16, the GF2_TW = 2 build) with synthetic multipliers / pivot rows:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DGF2_MB_NOLOOKUP | -DGF2_MB_L2] tools/microbench_update16.hip -o /tmp/mbu16 && /tmp/mbu16 [rows] [ntiles] [wgs]
// full build: real kernel (checked against a host recomputation on a sample of rows);  NOLOOKUP: HBM stream only;
// L2: table work only (row data stays in L2).
#define GF2_TW 2
#include "../gf2bv_amd/csrc/gf2_kernels.hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static u64 rng_state = 88172645463325252ull;
static u64 rnd() { u64 &x = rng_state; x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; }

template <int NT, int DEPTH, bool PIPE>
void run(const char *name, u64 *M, i64 rows, i64 srows, int ntiles, PanelRec *panels, PanelAux *aux, u64 *mult4, int *blkf, int wgs)
{
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto launch = [&] { k_update16<NT, false, DEPTH, PIPE, NT><<<dim3(wgs), dim3(NT), 0>>>(M, rows, srows, 0, GF2_GMAX, 0, panels, aux, mult4, blkf, 0, ntiles, 1, 0, nullptr, SysStride{0, 0}, 0); };
	launch(); CK(hipDeviceSynchronize());
	const int reps = getenv("MB_REPS") ? atoi(getenv("MB_REPS")) : 5;      // MB_REPS=400: sustained load (clocks settle)
	CK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
	double bytes = 2.0 * (double)(rows - 256) * ntiles * 16;
	printf("%-10s NT=%4d D=%d P=%d wgs=%4d: %.3f ms  %.2f TB/s per pass  (x4 = %.2f TB/s single-panel equivalent)\n", name, NT, DEPTH, (int)PIPE, wgs, ms,
	       bytes / ms / 1e9, 4 * bytes / ms / 1e9);
}

int main(int argc, char **argv)
{
	const i64 rows = argc > 1 ? atol(argv[1]) : 131072;
	const int ntiles = argc > 2 ? atoi(argv[2]) : 512;               // 8 KiB of row width by default: 1 GiB
	const int wgs = argc > 3 ? atoi(argv[3]) : 256;
	const i64 srows = (rows + 63) / 64 * 64 + (getenv("MB_PAD") ? atol(getenv("MB_PAD")) : 64);      // MB_PAD: rows of padding behind a tile (x 16 B): does the tile stride matter to the HBM channels?
	const i64 R64 = (rows + 63) / 64 * 64;
	u64 *M, *mult4; PanelRec *panels; PanelAux *aux; int *blkf;
	const size_t mwords = (size_t)ntiles * srows * 2;
	CK(hipMalloc(&M, mwords * 8));
	CK(hipMalloc(&mult4, (size_t)4 * R64 * 8));
	// plain multipliers m[g][row] -> stored form
	std::vector<u64> plain((size_t)4 * rows), hm((size_t)4 * R64, 0);
	for (auto &v : plain) v = rnd();
	for (int g = 0; g < 4; g++) for (int r = 0; r < 256; r++) plain[(size_t)g * rows + r] = 0;     // pivot rows: not updated
	for (i64 r = 0; r < rows; r++)
		for (int g = 0; g < 4; g++) {
			hm[(size_t)r * 4 + mult_slot(g, r)] = mult_rot(plain[(size_t)g * rows + r], r);
		}
	CK(hipMemcpy(mult4, hm.data(), hm.size() * 8, hipMemcpyHostToDevice));
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
	{   // correctness: one launch on a small random matrix, a sample of rows recomputed on the host
		const i64 crow = std::min<i64>(rows, 20000);
		const int ctiles = std::min(ntiles, 5);
		const i64 csr = (crow + 63) / 64 * 64 + 2;
		std::vector<u64> h0((size_t)ctiles * csr * 2), h1(h0.size());
		for (auto &v : h0) v = rnd();
		CK(hipMemcpy(M, h0.data(), h0.size() * 8, hipMemcpyHostToDevice));
		k_update16<768, false, 3, true, 768><<<dim3(7), dim3(768), 0>>>(M, crow, csr, 0, GF2_GMAX, 0, panels, aux, mult4, blkf, 0, ctiles, 1, 0, nullptr, SysStride{0, 0}, 0);
		CK(hipDeviceSynchronize());
		CK(hipMemcpy(h1.data(), M, h1.size() * 8, hipMemcpyDeviceToHost));
		i64 bad = 0, checked = 0;
		for (i64 r = 0; r < crow; r += (r < 2048 ? 1 : 37))
			for (int t = 0; t < ctiles; t++)
				for (int w = 0; w < 2; w++) {
					u64 e = h0[((size_t)t * csr + r) * 2 + w];
					if (r >= 256)
						for (int g = 0; g < 4; g++) {
							u64 m = plain[(size_t)g * rows + r];
							while (m) { int b = __builtin_ctzll(m); m &= m - 1; e ^= h0[((size_t)t * csr + 64 * g + b) * 2 + w]; }
						}
					checked++;
					if (e != h1[((size_t)t * csr + r) * 2 + w]) { if (bad < 5) printf("  MISMATCH row %lld tile %d word %d\n", (long long)r, t, w); bad++; }
				}
		printf("correctness: %lld words checked, %lld wrong\n", (long long)checked, (long long)bad);
	}
#endif
	CK(hipMemset(M, 0x5a, mwords * 8));
const char *only = getenv("MB_ONLY");          // e.g. MB_ONLY=768,2,1 : one variant (for rocprofv3 --pmc)
#define RUN(NT, D, P) do { char tag[32]; snprintf(tag, sizeof tag, "%d,%d,%d", NT, D, (int)P); if (!only || !strcmp(only, tag)) run<NT, D, P>(name, M, rows, srows, ntiles, panels, aux, mult4, blkf, wgs); } while (0)
	RUN(1024, 2, false); RUN(768, 2, false); RUN(512, 2, false);
	RUN(1024, 3, false); RUN(768, 3, false); RUN(512, 3, false);
	RUN(1024, 2, true); RUN(768, 2, true); RUN(512, 2, true);
	RUN(768, 3, true); RUN(512, 3, true); RUN(512, 4, true);
	return 0;
}
