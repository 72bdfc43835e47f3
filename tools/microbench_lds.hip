// LDS read-bandwidth ceilings on the GPU box for the access shapes of k_update:
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_lds.hip -o gpurun_out/mb_lds && gpurun_out/mb_lds
// Every workgroup (1024 threads, 128 KiB of LDS, one per CU) issues ds_read_b128 in batches of 8 with
//   lin : lane l reads 16 B at (l*16 + k*1024): the textbook conflict-free stream
//   e64 : groups of 4 lanes read a random 64-byte entry; the 4 entries of a 16-lane service group sit in
//         different quarters of the bank row (k_update, 64-byte tiles)
//   e128: groups of 8 lanes read a random 128-byte entry, 2 entries per service group in different halves
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(1024) k_lds(unsigned *sink, int iters, unsigned seed)
{
	__shared__ __attribute__((aligned(256))) uint4 tab[8192];
	for (int i = threadIdx.x; i < 8192; i += 1024) tab[i] = make_uint4(i, i * 3, i * 5, i * 7);
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const unsigned x = seed ^ (blockIdx.x * 40503u);
	uint4 acc = make_uint4(0, 0, 0, 0);
	const char *tb = reinterpret_cast<const char *>(tab);
	// lane constants
	unsigned c64, c128;
	{
		const int row = lane >> 2, lr = lane & 3;            // e64: 4 lanes per row, rowq as in k_update
		const int q = (row >> 1) & 3;
		c64 = (unsigned)(q * 4 + lr) * 16u;
		const int row8 = lane >> 3, l8 = lane & 7;           // e128: 8 lanes per row, halves alternate
		c128 = (unsigned)((row8 & 1) * 8 + l8) * 16u;
	}
	unsigned fixed[8];                                 // MODE 3 / 4: the e64 / e128 pattern with addresses computed once
#pragma unroll
	for (int k = 0; k < 8; k++) {
		const unsigned rowid = (MODE == 3) ? (threadIdx.x >> 2) : (threadIdx.x >> 3);
		unsigned r = (rowid + x) * 2654435761u + (unsigned)k * 40503u;
		r ^= r >> 13;
		fixed[k] = ((r & 511u) << 8) | ((MODE == 3) ? c64 : c128);
	}
	for (int it = 0; it < iters; it++) {
		uint4 v[8];
#pragma unroll
		for (int k = 0; k < 8; k++) {
			unsigned at;
			if (MODE == 0) at = (unsigned)lane * 16u + (unsigned)((it * 8 + k) & 127) * 1024u;
			else if (MODE >= 3) at = fixed[k] ^ ((unsigned)(it & 1) << 16);     // toggles the 64-KiB page: one VALU op per read
			else {
				// per-row pseudo-random slot: all lanes of a row compute the same value (no cross-lane traffic)
				const unsigned rowid = (MODE == 1) ? (threadIdx.x >> 2) : (threadIdx.x >> 3);
				unsigned r = (rowid + x) * 2654435761u + (unsigned)(it * 8 + k) * 40503u;
				r ^= r >> 13;
				at = ((r & 511u) << 8) | ((MODE == 1) ? c64 : c128);
			}
			v[k] = *reinterpret_cast<const uint4 *>(tb + at);
		}
#pragma unroll
		for (int k = 0; k < 8; k++) { acc.x ^= v[k].x; acc.y ^= v[k].y; acc.z ^= v[k].z; acc.w ^= v[k].w; }
	}
	if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = acc.x;
}

template <int MODE>
void run(const char *name, unsigned *sink, int wgs)
{
	const int iters = 4096;
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	k_lds<MODE><<<wgs, 1024>>>(sink, 16, 1u); CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0)); k_lds<MODE><<<wgs, 1024>>>(sink, iters, 7u); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	const double bytes = (double)wgs * 1024 * 16.0 * 8 * iters;
	printf("%-5s %4d workgroups: %8.3f ms  %8.1f TB/s  (%.1f B/clk/CU at 2.4 GHz)\n", name, wgs, ms, bytes / ms / 1e9,
	       bytes / ms / 1e-3 / wgs / 2.4e9);
}

int main()
{
	unsigned *sink; CK(hipMalloc(&sink, 4));
	for (int wgs : { 256, 512 }) {
		run<0>("lin", sink, wgs);
		run<1>("e64", sink, wgs);
		run<2>("e128", sink, wgs);
		run<3>("f64", sink, wgs);
		run<4>("f128", sink, wgs);
	}
	return 0;
}
