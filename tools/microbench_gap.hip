// What does a stream hand-over cost?  Gap between the last instruction of kernel K1 and the first of K2 (100 MHz wall clock
// stamps taken inside the kernels) for the ways the solver chains its launches:
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_gap.hip -o /tmp/mbgap && /tmp/mbgap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1);} } while (0)
typedef unsigned long long u64;
__global__ void k1(u64 *stamp, int *flag, int spin_us)
{
	const u64 t0 = wall_clock64();
	while (wall_clock64() - t0 < (u64)spin_us * 100) { }
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		stamp[0] = wall_clock64();
		if (flag) __hip_atomic_store(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
	}
}
__global__ void k2(u64 *stamp, int *flag)
{
	if (flag) while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) { }
	if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
}
int main()
{
	hipStream_t s1, s2; int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
	CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo));
	u64 *stamp; int *flag; CK(hipMalloc(&stamp, 64)); CK(hipMalloc(&flag, 4));
	hipEvent_t old, e, et; CK(hipEventCreateWithFlags(&old, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); CK(hipEventCreate(&et));
	CK(hipEventRecord(old, s2)); CK(hipDeviceSynchronize());
	const char *names[] = { "same stream, back to back", "K1 carries a completion event (ext launch)", "hipEventRecord between", "wait on a long-complete event between",
	                        "completion event + wait on old event", "cross-stream: K1 completion event, s2 waits", "cross-stream: hipEventRecord, s2 waits",
	                        "cross-stream: flag in memory, K2 spins", "K1 carries a TIMING completion event", "cross-stream: wait enqueued, then K1 launched (K2 queue idle)",
	                        "K1; wait on an event of the other stream that completes while K1 runs; K2", "same, K1 also carries a completion event",
	                        "cross-stream: K1, hipStreamWriteValue32 | hipStreamWaitValue32, K2", "cross-stream: K1, hipStreamWriteValue32 | K2 spins",
	                        "cross-stream: K1 stores the flag | hipStreamWaitValue32, K2", "same stream: K1, WriteValue32, K2 (cost of the write op)" };
	for (int wg = 1; wg <= 256; wg *= 256)
	for (int v = 0; v < 16; v++) {
		std::vector<double> gaps;
		for (int rep = 0; rep < 25; rep++) {
			CK(hipMemsetAsync(flag, 0, 4, s1)); CK(hipStreamSynchronize(s1));
			switch (v) {
			case 0: k1<<<wg, 256, 0, s1>>>(stamp, nullptr, 30); k2<<<wg, 256, 0, s1>>>(stamp, nullptr); break;
			case 1: hipExtLaunchKernelGGL(k1, dim3(wg), dim3(256), 0, s1, nullptr, e, 0, stamp, (int *)nullptr, 30); k2<<<wg, 256, 0, s1>>>(stamp, nullptr); break;
			case 2: k1<<<wg, 256, 0, s1>>>(stamp, nullptr, 30); CK(hipEventRecord(e, s1)); k2<<<wg, 256, 0, s1>>>(stamp, nullptr); break;
			case 3: k1<<<wg, 256, 0, s1>>>(stamp, nullptr, 30); CK(hipStreamWaitEvent(s1, old, 0)); k2<<<wg, 256, 0, s1>>>(stamp, nullptr); break;
			case 4: hipExtLaunchKernelGGL(k1, dim3(wg), dim3(256), 0, s1, nullptr, e, 0, stamp, (int *)nullptr, 30); CK(hipStreamWaitEvent(s1, old, 0)); k2<<<wg, 256, 0, s1>>>(stamp, nullptr); break;
			case 5: hipExtLaunchKernelGGL(k1, dim3(wg), dim3(256), 0, s1, nullptr, e, 0, stamp, (int *)nullptr, 30); CK(hipStreamWaitEvent(s2, e, 0)); k2<<<wg, 256, 0, s2>>>(stamp, nullptr); break;
			case 6: k1<<<wg, 256, 0, s1>>>(stamp, nullptr, 30); CK(hipEventRecord(e, s1)); CK(hipStreamWaitEvent(s2, e, 0)); k2<<<wg, 256, 0, s2>>>(stamp, nullptr); break;
			case 7: k1<<<wg, 256, 0, s1>>>(stamp, flag, 30); k2<<<1, 64, 0, s2>>>(stamp, flag); break;
			case 8: hipExtLaunchKernelGGL(k1, dim3(wg), dim3(256), 0, s1, nullptr, et, 0, stamp, (int *)nullptr, 30); k2<<<wg, 256, 0, s1>>>(stamp, nullptr); break;
			case 9: hipExtLaunchKernelGGL(k1, dim3(wg), dim3(256), 0, s1, nullptr, e, 0, stamp, (int *)nullptr, 200); CK(hipStreamWaitEvent(s2, e, 0)); k2<<<wg, 256, 0, s2>>>(stamp, nullptr); break;
			case 10: hipExtLaunchKernelGGL(k1, dim3(wg), dim3(256), 0, s2, nullptr, e, 0, stamp + 2, (int *)nullptr, 60); k1<<<wg, 256, 0, s1>>>(stamp, nullptr, 300); CK(hipStreamWaitEvent(s1, e, 0)); k2<<<wg, 256, 0, s1>>>(stamp, nullptr); break;
			case 12: k1<<<wg, 256, 0, s1>>>(stamp, nullptr, 30); CK(hipStreamWriteValue32(s1, flag, 1, 0)); CK(hipStreamWaitValue32(s2, flag, 1, hipStreamWaitValueGte, 0xffffffffu)); k2<<<wg, 256, 0, s2>>>(stamp, nullptr); break;
			case 13: k1<<<wg, 256, 0, s1>>>(stamp, nullptr, 30); CK(hipStreamWriteValue32(s1, flag, 1, 0)); k2<<<1, 64, 0, s2>>>(stamp, flag); break;
			case 14: k1<<<wg, 256, 0, s1>>>(stamp, flag, 30); CK(hipStreamWaitValue32(s2, flag, 1, hipStreamWaitValueGte, 0xffffffffu)); k2<<<wg, 256, 0, s2>>>(stamp, nullptr); break;
			case 15: k1<<<wg, 256, 0, s1>>>(stamp, nullptr, 30); CK(hipStreamWriteValue32(s1, flag, 1, 0)); k2<<<wg, 256, 0, s1>>>(stamp, nullptr); break;
			case 11: hipExtLaunchKernelGGL(k1, dim3(wg), dim3(256), 0, s2, nullptr, e, 0, stamp + 2, (int *)nullptr, 60); hipExtLaunchKernelGGL(k1, dim3(wg), dim3(256), 0, s1, nullptr, et, 0, stamp, (int *)nullptr, 300); CK(hipStreamWaitEvent(s1, e, 0)); k2<<<wg, 256, 0, s1>>>(stamp, nullptr); break;
			}
			CK(hipDeviceSynchronize());
			u64 h[2]; CK(hipMemcpy(h, stamp, 16, hipMemcpyDeviceToHost));
			gaps.push_back(((double)h[1] - (double)h[0]) / 100.0);
		}
		std::sort(gaps.begin(), gaps.end());
		printf("wgs %3d  %-62s  gap min %6.1f  median %6.1f  max %6.1f us\n", wg, names[v], gaps[0], gaps[gaps.size() / 2], gaps.back());
	}
	return 0;
}
