// What does this chip stream?  Independent reference points for the roofline discussion:
//   (1) hipMemcpyDtoD of 4 GiB (the runtime's own copy path: blit kernel / SDMA, nothing of this repo),
//   (2) BabelStream-form Copy / Mul / Add / Triad (one double per thread, 1024-thread workgroups, arrays of 2 GiB each --
//       the form and the byte accounting of the public BabelStream benchmark: Copy/Mul 2 arrays, Add/Triad 3),
//   (3) the same Copy / Triad with non-temporal loads and stores (__builtin_nontemporal_*),
//   (4) 16-byte-per-thread copy (float4 form, the shape MI355X_MICROARCH.md quotes 6.29 TB/s for),
//   (5) in-place read-XOR-write (the access pattern of the bulk update), 16 B per thread, plain and non-temporal,
//   (6) read-only and write-only streams.
// All on one box in one run; every figure is bytes moved (read + written) / time, best and median of REPS launches.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_ceiling.hip -o /tmp/sc && /tmp/sc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define TB 1024
__global__ void __launch_bounds__(TB) bs_copy(const double *a, double *c) { const size_t i = (size_t)blockDim.x * blockIdx.x + threadIdx.x; c[i] = a[i]; }
__global__ void __launch_bounds__(TB) bs_mul(double *b, const double *c) { const size_t i = (size_t)blockDim.x * blockIdx.x + threadIdx.x; b[i] = 0.4 * c[i]; }
__global__ void __launch_bounds__(TB) bs_add(const double *a, const double *b, double *c) { const size_t i = (size_t)blockDim.x * blockIdx.x + threadIdx.x; c[i] = a[i] + b[i]; }
__global__ void __launch_bounds__(TB) bs_triad(double *a, const double *b, const double *c) { const size_t i = (size_t)blockDim.x * blockIdx.x + threadIdx.x; a[i] = b[i] + 0.4 * c[i]; }
__global__ void __launch_bounds__(TB) nt_copy(const double *a, double *c) { const size_t i = (size_t)blockDim.x * blockIdx.x + threadIdx.x; __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), c + i); }
__global__ void __launch_bounds__(TB) nt_triad(double *a, const double *b, const double *c) { const size_t i = (size_t)blockDim.x * blockIdx.x + threadIdx.x; __builtin_nontemporal_store(__builtin_nontemporal_load(b + i) + 0.4 * __builtin_nontemporal_load(c + i), a + i); }

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(TB) f4_copy(const f4 *a, f4 *c) { const size_t i = (size_t)blockDim.x * blockIdx.x + threadIdx.x; c[i] = a[i]; }
__global__ void __launch_bounds__(TB) f4_copy_nt(const f4 *a, f4 *c) { const size_t i = (size_t)blockDim.x * blockIdx.x + threadIdx.x; __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), c + i); }
__global__ void __launch_bounds__(256) f4_copy_gs(const f4 *a, f4 *c, size_t n) { for (size_t i = (size_t)blockDim.x * blockIdx.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c[i] = a[i]; }
__global__ void __launch_bounds__(TB) u4_rmw(u4 *a, unsigned k) { const size_t i = (size_t)blockDim.x * blockIdx.x + threadIdx.x; a[i] = a[i] ^ k; }
__global__ void __launch_bounds__(TB) u4_rmw_nt(u4 *a, unsigned k) { const size_t i = (size_t)blockDim.x * blockIdx.x + threadIdx.x; __builtin_nontemporal_store(__builtin_nontemporal_load(a + i) ^ k, a + i); }
__global__ void __launch_bounds__(256) u4_rmw_gs(u4 *a, size_t n, unsigned k) { for (size_t i = (size_t)blockDim.x * blockIdx.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = a[i] ^ k; }
// every workgroup owns one contiguous range (the bulk update's partition), 4 x 16 B per lane in flight
__global__ void __launch_bounds__(TB) u4_rmw_range(u4 *a, size_t per_wg, unsigned k)
{
	u4 *p = a + (size_t)blockIdx.x * per_wg;
	for (size_t i = threadIdx.x; i < per_wg; i += 4 * TB) {
		u4 v[4];
#pragma unroll
		for (int u = 0; u < 4; u++) v[u] = p[i + (size_t)u * TB];
#pragma unroll
		for (int u = 0; u < 4; u++) p[i + (size_t)u * TB] = v[u] ^ k;
	}
}
// the same with non-temporal loads (NL) and / or stores (NS)
template <bool NL, bool NS>
__global__ void __launch_bounds__(TB) u4_rmw_range_nt(u4 *a, size_t per_wg, unsigned k)
{
	u4 *p = a + (size_t)blockIdx.x * per_wg;
	for (size_t i = threadIdx.x; i < per_wg; i += 4 * TB) {
		u4 v[4];
#pragma unroll
		for (int u = 0; u < 4; u++) v[u] = NL ? __builtin_nontemporal_load(p + i + (size_t)u * TB) : p[i + (size_t)u * TB];
#pragma unroll
		for (int u = 0; u < 4; u++) { if (NS) __builtin_nontemporal_store(v[u] ^ k, p + i + (size_t)u * TB); else p[i + (size_t)u * TB] = v[u] ^ k; }
	}
}
// ... and with a second, cache-resident read stream of twice the bytes beside it (the bulk update reads 32 B of
// multipliers, an 8 MiB array shared by all workgroups, per 16 B of row data)
template <bool NL, bool NS>
__global__ void __launch_bounds__(TB) u4_rmw_range_side(u4 *a, size_t per_wg, const u4 *side, size_t side_mask, unsigned k)
{
	u4 *p = a + (size_t)blockIdx.x * per_wg;
	for (size_t i = threadIdx.x; i < per_wg; i += 2 * TB) {
		u4 v[2], m[4];
#pragma unroll
		for (int u = 0; u < 2; u++) {
			v[u] = NL ? __builtin_nontemporal_load(p + i + (size_t)u * TB) : p[i + (size_t)u * TB];
			m[2 * u] = side[(2 * (i + (size_t)u * TB)) & side_mask]; m[2 * u + 1] = side[(2 * (i + (size_t)u * TB) + 1) & side_mask];
		}
#pragma unroll
		for (int u = 0; u < 2; u++) { const u4 r = v[u] ^ m[2 * u] ^ m[2 * u + 1] ^ k; if (NS) __builtin_nontemporal_store(r, p + i + (size_t)u * TB); else p[i + (size_t)u * TB] = r; }
	}
}
__global__ void __launch_bounds__(256) u4_read_gs(const u4 *a, size_t n, unsigned *out) { unsigned acc = 0; for (size_t i = (size_t)blockDim.x * blockIdx.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { u4 v = a[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; } if (acc == 0x12345) out[0] = acc; }
__global__ void __launch_bounds__(TB) u4_write(u4 *a, unsigned k) { const size_t i = (size_t)blockDim.x * blockIdx.x + threadIdx.x; a[i] = (u4){k, k, k, k}; }

template <typename F> void timeit(const char *name, double bytes, F f, int reps = 20)
{
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	f(); f(); CK(hipDeviceSynchronize());
	std::vector<float> t(reps);
	for (int r = 0; r < reps; r++) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t[r], e0, e1)); }
	std::sort(t.begin(), t.end());
	printf("%-44s %8.3f ms best %8.3f ms median   %6.2f TB/s best %6.2f TB/s median\n", name, t[0], t[reps / 2], bytes / t[0] / 1e9, bytes / t[reps / 2] / 1e9);
	fflush(stdout);
}

int main(int argc, char **argv)
{
	const size_t bytes = (argc > 1 ? (size_t)atol(argv[1]) : 2048) << 20;     // per array, MiB (default 2 GiB)
	hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
	printf("device: %s  CUs %d  memory clock %d kHz  bus %d bit  -> spec %.2f TB/s;  arrays of %.2f GiB\n", pr.name, pr.multiProcessorCount, pr.memoryClockRate,
	       pr.memoryBusWidth, 2.0 * pr.memoryClockRate * 1e3 * pr.memoryBusWidth / 8 / 1e12, bytes / 1073741824.0);
	char *A, *B, *C; unsigned *o;
	CK(hipMalloc(&A, 2 * bytes)); CK(hipMalloc(&B, 2 * bytes)); CK(hipMalloc(&C, bytes)); CK(hipMalloc(&o, 64));
	CK(hipMemset(A, 1, 2 * bytes)); CK(hipMemset(B, 2, 2 * bytes)); CK(hipMemset(C, 3, bytes));
	const size_t nd = bytes / 8, n4 = bytes / 16;
	double *a = (double *)A, *b = (double *)B, *c = (double *)C;
	printf("-- (1) runtime copy\n");
	timeit("hipMemcpyDtoD 4 GiB", 2.0 * 2 * bytes, [&] { CK(hipMemcpyDtoD((hipDeviceptr_t)B, (hipDeviceptr_t)A, 2 * bytes)); }, 10);
	timeit("hipMemcpyAsync D2D 4 GiB (null stream)", 2.0 * 2 * bytes, [&] { CK(hipMemcpyAsync(B, A, 2 * bytes, hipMemcpyDeviceToDevice, 0)); }, 10);
	printf("-- (2) BabelStream form: one double per thread, %d-thread workgroups\n", TB);
	timeit("Copy  c = a", 2.0 * bytes, [&] { bs_copy<<<nd / TB, TB>>>(a, c); });
	timeit("Mul   b = s c", 2.0 * bytes, [&] { bs_mul<<<nd / TB, TB>>>(b, c); });
	timeit("Add   c = a + b", 3.0 * bytes, [&] { bs_add<<<nd / TB, TB>>>(a, b, c); });
	timeit("Triad a = b + s c", 3.0 * bytes, [&] { bs_triad<<<nd / TB, TB>>>(a, b, c); });
	printf("-- (3) non-temporal loads and stores\n");
	timeit("Copy  nt", 2.0 * bytes, [&] { nt_copy<<<nd / TB, TB>>>(a, c); });
	timeit("Triad nt", 3.0 * bytes, [&] { nt_triad<<<nd / TB, TB>>>(a, b, c); });
	printf("-- (4) 16 B per thread copy\n");
	timeit("float4 copy, one per thread", 2.0 * bytes, [&] { f4_copy<<<n4 / TB, TB>>>((f4 *)A, (f4 *)C); });
	timeit("float4 copy nt, one per thread", 2.0 * bytes, [&] { f4_copy_nt<<<n4 / TB, TB>>>((f4 *)A, (f4 *)C); });
	for (int blocks : {2048, 8192, 32768})
		{ char nm[64]; snprintf(nm, sizeof nm, "float4 copy grid-stride %d x 256", blocks); timeit(nm, 2.0 * bytes, [&] { f4_copy_gs<<<blocks, 256>>>((f4 *)A, (f4 *)C, n4); }); }
	printf("-- (5) in-place read-XOR-write, 16 B per thread\n");
	timeit("rmw one per thread", 2.0 * bytes, [&] { u4_rmw<<<n4 / TB, TB>>>((u4 *)A, 5); });
	timeit("rmw nt one per thread", 2.0 * bytes, [&] { u4_rmw_nt<<<n4 / TB, TB>>>((u4 *)A, 5); });
	for (int blocks : {2048, 8192})
		{ char nm[64]; snprintf(nm, sizeof nm, "rmw grid-stride %d x 256", blocks); timeit(nm, 2.0 * bytes, [&] { u4_rmw_gs<<<blocks, 256>>>((u4 *)A, n4, 5); }); }
	for (int wgs : {256, 512, 1024})
		{ char nm[64]; snprintf(nm, sizeof nm, "rmw contiguous range per workgroup, %d wgs", wgs); timeit(nm, 2.0 * bytes, [&] { u4_rmw_range<<<wgs, TB>>>((u4 *)A, n4 / wgs, 5); }); }
	for (int wgs : {256, 1024}) {
		char nm[80];
		snprintf(nm, sizeof nm, "rmw range, nt loads + nt stores, %d wgs", wgs); timeit(nm, 2.0 * bytes, [&] { u4_rmw_range_nt<true, true><<<wgs, TB>>>((u4 *)A, n4 / wgs, 5); });
		snprintf(nm, sizeof nm, "rmw range, nt loads only, %d wgs", wgs); timeit(nm, 2.0 * bytes, [&] { u4_rmw_range_nt<true, false><<<wgs, TB>>>((u4 *)A, n4 / wgs, 5); });
		snprintf(nm, sizeof nm, "rmw range, nt stores only, %d wgs", wgs); timeit(nm, 2.0 * bytes, [&] { u4_rmw_range_nt<false, true><<<wgs, TB>>>((u4 *)A, n4 / wgs, 5); });
	}
	printf("-- (5b) the same range form beside a cache-resident read stream of 2x the bytes (8 MiB array), rate = row bytes only\n");
	for (int wgs : {256}) {
		char nm[80];
		const size_t smask = (8u << 20) / 16 - 1;
		snprintf(nm, sizeof nm, "rmw range + side stream, plain, %d wgs", wgs); timeit(nm, 2.0 * bytes, [&] { u4_rmw_range_side<false, false><<<wgs, TB>>>((u4 *)A, n4 / wgs, (const u4 *)B, smask, 5); });
		snprintf(nm, sizeof nm, "rmw range + side stream, nt rows, %d wgs", wgs); timeit(nm, 2.0 * bytes, [&] { u4_rmw_range_side<true, true><<<wgs, TB>>>((u4 *)A, n4 / wgs, (const u4 *)B, smask, 5); });
		snprintf(nm, sizeof nm, "rmw range + side stream, nt stores, %d wgs", wgs); timeit(nm, 2.0 * bytes, [&] { u4_rmw_range_side<false, true><<<wgs, TB>>>((u4 *)A, n4 / wgs, (const u4 *)B, smask, 5); });
	}
	printf("-- (6) one direction only\n");
	timeit("read  grid-stride 8192 x 256", 1.0 * bytes, [&] { u4_read_gs<<<8192, 256>>>((u4 *)A, n4, o); });
	timeit("write one per thread", 1.0 * bytes, [&] { u4_write<<<n4 / TB, TB>>>((u4 *)A, 7); });
	timeit("hipMemsetAsync", 1.0 * bytes, [&] { CK(hipMemsetAsync(A, 0, bytes, 0)); }, 10);
	return 0;
}
