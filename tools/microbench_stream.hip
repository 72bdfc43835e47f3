// Streaming micro-benchmarks on the GPU box: what HBM rate do the access patterns of k_update reach
// without any LDS work?  hipcc --offload-arch=gfx950 -O3 tools/microbench_stream.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_copy(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_inplace(uint4 *__restrict__ a, size_t n, unsigned c)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		uint4 v = a[i]; v.x ^= c; v.y ^= c; v.z ^= c; v.w ^= c; a[i] = v;
	}
}
// each workgroup owns a contiguous range (like k_update: one column tile, a row range), U chunks per lane in flight
template <int U>
__global__ void __launch_bounds__(1024) k_range(uint4 *__restrict__ a, size_t per_wg, unsigned c)
{
	uint4 *p = a + (size_t)blockIdx.x * per_wg;
	for (size_t i = threadIdx.x; i < per_wg; i += (size_t)1024 * U) {
		uint4 v[U];
#pragma unroll
		for (int u = 0; u < U; u++) if (i + (size_t)u * 1024 < per_wg) v[u] = p[i + (size_t)u * 1024];
#pragma unroll
		for (int u = 0; u < U; u++) if (i + (size_t)u * 1024 < per_wg) { v[u].x ^= c; v[u].y ^= c; v[u].z ^= c; v[u].w ^= c; p[i + (size_t)u * 1024] = v[u]; }
	}
}
// read-only and write-only
__global__ void k_read(const uint4 *__restrict__ a, size_t n, unsigned *out)
{
	unsigned acc = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = a[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
	if (acc == 0x12345) out[0] = acc;
}
__global__ void k_write(uint4 *__restrict__ a, size_t n, unsigned c)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = make_uint4(c, c, c, c);
}

template <typename F> float timeit(F f, int reps = 5)
{
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	f(); CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

int main(int argc, char **argv)
{
	// default 2 GiB working set (>> 256 MiB infinity cache); argv[1] = MiB (e.g. 64: resident in the infinity cache)
	const size_t bytes = argc > 1 ? (size_t)atol(argv[1]) << 20 : (size_t)2 << 30;
	const size_t n = bytes / 16;
	uint4 *a, *b; unsigned *o;
	CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 64));
	CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
	for (int blocks : {2048, 8192}) {
		float t = timeit([&] { k_copy<<<blocks, 256>>>(a, b, n); });
		printf("copy      blocks=%5d x256: %.3f ms  %.2f TB/s (r+w)\n", blocks, t, 2.0 * bytes / t / 1e9);
		t = timeit([&] { k_inplace<<<blocks, 256>>>(a, n, 7); });
		printf("inplace   blocks=%5d x256: %.3f ms  %.2f TB/s (r+w)\n", blocks, t, 2.0 * bytes / t / 1e9);
		t = timeit([&] { k_read<<<blocks, 256>>>(a, n, o); });
		printf("read      blocks=%5d x256: %.3f ms  %.2f TB/s\n", blocks, t, 1.0 * bytes / t / 1e9);
		t = timeit([&] { k_write<<<blocks, 256>>>(a, n, 9); });
		printf("write     blocks=%5d x256: %.3f ms  %.2f TB/s\n", blocks, t, 1.0 * bytes / t / 1e9);
	}
	for (int wgs : {256, 512, 1024, 2048, 4096}) {
		size_t per = n / wgs;
		float t = timeit([&] { k_range<1><<<wgs, 1024>>>(a, per, 3); });
		printf("range U=1 wgs=%5d x1024: %.3f ms  %.2f TB/s\n", wgs, t, 2.0 * bytes / t / 1e9);
		t = timeit([&] { k_range<2><<<wgs, 1024>>>(a, per, 3); });
		printf("range U=2 wgs=%5d x1024: %.3f ms  %.2f TB/s\n", wgs, t, 2.0 * bytes / t / 1e9);
		t = timeit([&] { k_range<4><<<wgs, 1024>>>(a, per, 3); });
		printf("range U=4 wgs=%5d x1024: %.3f ms  %.2f TB/s\n", wgs, t, 2.0 * bytes / t / 1e9);
	}
	return 0;
}
